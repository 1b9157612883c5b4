// TEST INFRASTRUCTURE (oracle/_ref build only).  sqp::Settings / loadSettings as LeggedInterface.cpp:94-99 calls them: the solver
// settings are read by OCS2 code that is not here, so this keeps only where they were asked for (file, block name).
#pragma once
#include <string>
namespace ocs2 {
namespace sqp {
struct Settings { std::string file, block; };
inline Settings loadSettings(const std::string& file, const std::string& block = "sqp", bool = true) { return Settings{file, block}; }
}  // namespace sqp
}  // namespace ocs2
