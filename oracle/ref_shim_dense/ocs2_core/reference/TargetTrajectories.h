// TEST INFRASTRUCTURE (oracle/_ref build only). Stand-in for OCS2's <ocs2_core/reference/TargetTrajectories.h>
// [OCS2-knowledge: published definition]: three parallel arrays and getDesiredState = piecewise-linear interpolation of the
// state array, held constant outside [front, back] (OCS2 LinearInterpolation::timeSegment: alpha = (t_{i+1} - t) /
// (t_{i+1} - t_i), value alpha x_i + (1 - alpha) x_{i+1}).
#pragma once
#include <stdexcept>
#include <ocs2_core/Types.h>
#include <ocs2_core/misc/Lookup.h>
namespace ocs2 {
struct TargetTrajectories {
  explicit TargetTrajectories(size_t size = 0) : timeTrajectory(size), stateTrajectory(size), inputTrajectory(size) {}
  TargetTrajectories(scalar_array_t desiredTimeTrajectory, vector_array_t desiredStateTrajectory,
                     vector_array_t desiredInputTrajectory = vector_array_t())
      : timeTrajectory(std::move(desiredTimeTrajectory)), stateTrajectory(std::move(desiredStateTrajectory)),
        inputTrajectory(std::move(desiredInputTrajectory)) {}
  bool empty() const { return timeTrajectory.empty() || stateTrajectory.empty(); }
  size_t size() const { return timeTrajectory.size(); }
  vector_t getDesiredState(scalar_t time) const {
    if (empty()) throw std::runtime_error("[TargetTrajectories] TargetTrajectories is empty!");
    return interpolate(time, stateTrajectory);
  }
  vector_t getDesiredInput(scalar_t time) const {
    if (timeTrajectory.empty() || inputTrajectory.empty()) throw std::runtime_error("[TargetTrajectories] TargetTrajectories is empty!");
    return interpolate(time, inputTrajectory);
  }
  scalar_array_t timeTrajectory;
  vector_array_t stateTrajectory;
  vector_array_t inputTrajectory;

 private:
  vector_t interpolate(scalar_t time, const vector_array_t& data) const {
    const auto& ta = timeTrajectory;
    if (ta.size() <= 1) return data[0];
    const int index = lookup::findIntervalInTimeArray(ta, time);
    const int lastInterval = static_cast<int>(ta.size()) - 1;
    int i;
    scalar_t alpha;
    if (index < 0) { i = 0; alpha = 1.0; }
    else if (index >= lastInterval) { i = std::max(lastInterval - 1, 0); alpha = 0.0; }
    else { i = index; alpha = (ta[size_t(i) + 1] - time) / (ta[size_t(i) + 1] - ta[size_t(i)]); }
    return alpha * data[size_t(i)] + (scalar_t(1.0) - alpha) * data[size_t(i) + 1];
  }
};
}  // namespace ocs2
