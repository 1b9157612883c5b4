#pragma once
#include <ocs2_centroidal_model/CentroidalModelInfo.h>
