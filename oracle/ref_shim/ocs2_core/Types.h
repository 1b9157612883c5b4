// TEST INFRASTRUCTURE (oracle/_ref build only). Stand-in for OCS2's <ocs2_core/Types.h>, which is not vendored in the
// reference. The reference's CubicSpline / MultiCubicSpline sources use exactly one thing from it — the typedef
// `ocs2::scalar_t` (= double in OCS2) — plus <vector> / <cassert> that the real header pulls in transitively.
#pragma once
#include <cassert>
#include <cstddef>
#include <string>
#include <vector>
namespace ocs2 {
using scalar_t = double;
}
