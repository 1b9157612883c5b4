// TEST INFRASTRUCTURE (oracle/_ref build only).  mpc::Settings / loadSettings as LeggedInterface.cpp:94-99 calls them: the solver
// settings are read by OCS2 code that is not here, so this keeps only where they were asked for (file, block name).
#pragma once
#include <string>
namespace ocs2 {
namespace mpc {
struct Settings { std::string file, block; };
inline Settings loadSettings(const std::string& file, const std::string& block = "mpc", bool = true) { return Settings{file, block}; }
}  // namespace mpc
}  // namespace ocs2
