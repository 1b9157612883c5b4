// TEST STAND-IN: the registration macro must at least name two complete, related types.
#pragma once
#include <type_traits>
#define PLUGINLIB_EXPORT_CLASS(cls, base) static_assert(std::is_base_of<base, cls>::value, "plugin class must derive from its base");
