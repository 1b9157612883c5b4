// TEST INFRASTRUCTURE (oracle/_ref build only). Stand-in for OCS2's <ocs2_core/misc/Lookup.h> [OCS2-knowledge: published
// definitions]: index i of a sorted time array such that t_{i-1} < t <= t_i (std::lower_bound), and the interval variant
// (index - 1, except that t == front belongs to interval 0).
#pragma once
#include <algorithm>
#include <vector>
#include <ocs2_core/Types.h>
namespace ocs2 {
namespace lookup {
template <typename SCALAR = scalar_t>
int findIndexInTimeArray(const std::vector<SCALAR>& timeArray, SCALAR time) {
  return static_cast<int>(std::lower_bound(timeArray.begin(), timeArray.end(), time) - timeArray.begin());
}
template <typename SCALAR = scalar_t>
int findIntervalInTimeArray(const std::vector<SCALAR>& timeArray, SCALAR time) {
  if (!timeArray.empty() && !(time != timeArray.front())) return 0;
  return findIndexInTimeArray(timeArray, time) - 1;
}
}  // namespace lookup
}  // namespace ocs2
