// TEST INFRASTRUCTURE (oracle/_ref build only).
#pragma once
#include <hardware_interface/joint_state_interface.h>
