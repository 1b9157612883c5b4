// TEST INFRASTRUCTURE (oracle/_ref build only).  ddp::Settings / loadSettings as LeggedInterface.cpp:94-99 calls them: the solver
// settings are read by OCS2 code that is not here, so this keeps only where they were asked for (file, block name).
#pragma once
#include <string>
namespace ocs2 {
namespace ddp {
struct Settings { std::string file, block; };
inline Settings loadSettings(const std::string& file, const std::string& block = "ddp", bool = true) { return Settings{file, block}; }
}  // namespace ddp
}  // namespace ocs2
