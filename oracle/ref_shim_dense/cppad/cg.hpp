// TEST INFRASTRUCTURE (oracle/_ref build only).  legged_interface/common/utils.h includes <cppad/cg.hpp> without using it.
#pragma once
