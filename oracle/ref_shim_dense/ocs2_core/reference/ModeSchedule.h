// TEST INFRASTRUCTURE (oracle/_ref build only). Stand-in for OCS2's <ocs2_core/reference/ModeSchedule.h> [OCS2-knowledge:
// published definition]: event times, mode sequence (one more entry), modeAtTime = modeSequence[findIndexInTimeArray].
#pragma once
#include <ostream>
#include <ocs2_core/Types.h>
#include <ocs2_core/misc/Lookup.h>
namespace ocs2 {
struct ModeSchedule {
  ModeSchedule() : ModeSchedule(std::vector<scalar_t>{}, std::vector<size_t>{0}) {}
  ModeSchedule(std::vector<scalar_t> eventTimesInput, std::vector<size_t> modeSequenceInput)
      : eventTimes(std::move(eventTimesInput)), modeSequence(std::move(modeSequenceInput)) {
    assert(!modeSequence.empty());
    assert(eventTimes.size() + 1 == modeSequence.size());
  }
  size_t modeAtTime(scalar_t time) const { return modeSequence[size_t(lookup::findIndexInTimeArray(eventTimes, time))]; }
  std::vector<scalar_t> eventTimes;
  std::vector<size_t> modeSequence;
};
inline std::ostream& operator<<(std::ostream& os, const ModeSchedule& m) {
  os << "event times: {";
  for (scalar_t t : m.eventTimes) os << t << ", ";
  os << "}, mode sequence: {";
  for (size_t v : m.modeSequence) os << v << ", ";
  return os << "}\n";
}
}  // namespace ocs2
