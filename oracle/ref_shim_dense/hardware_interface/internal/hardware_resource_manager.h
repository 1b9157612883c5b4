#pragma once
#include <hardware_interface/joint_state_interface.h>
