#pragma once
#include <ocs2_centroidal_model/CentroidalModelPinocchioMapping.h>
