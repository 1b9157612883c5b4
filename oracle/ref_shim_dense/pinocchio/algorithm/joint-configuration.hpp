#pragma once
#include <pinocchio/shim.hpp>
