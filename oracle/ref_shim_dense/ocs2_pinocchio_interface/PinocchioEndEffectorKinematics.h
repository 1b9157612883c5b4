// TEST INFRASTRUCTURE (oracle/_ref build only).  Stand-in for OCS2's PinocchioEndEffectorKinematics: positions / velocities of
// the four contact frames of the interface last set, taken from the feed (ref_feed.h).
#pragma once
#include <vector>
#include <ocs2_core/Types.h>
#include <ocs2_pinocchio_interface/PinocchioInterface.h>
namespace ocs2 {
class PinocchioEndEffectorKinematics {
 public:
  using vector3_t = Eigen::Matrix<scalar_t, 3, 1>;
  PinocchioEndEffectorKinematics() = default;
  template <class Mapping, class Names> PinocchioEndEffectorKinematics(const PinocchioInterface& i, const Mapping&, const Names&) : iface_(&i) {}
  PinocchioEndEffectorKinematics* clone() const { return new PinocchioEndEffectorKinematics(*this); }
  void setPinocchioInterface(const PinocchioInterface& i) { iface_ = &i; }
  std::vector<vector3_t> getPosition(const vector_t&) const { return get(ref_feed::feed().role[iface_->getData().role].ee_pos); }
  std::vector<vector3_t> getVelocity(const vector_t&, const vector_t&) const { return get(ref_feed::feed().role[iface_->getData().role].ee_vel); }
 private:
  static std::vector<vector3_t> get(const double* p) {
    std::vector<vector3_t> out;
    for (int i = 0; i < 4; ++i) out.emplace_back(p[3 * i], p[3 * i + 1], p[3 * i + 2]);
    return out;
  }
  const PinocchioInterface* iface_ = nullptr;
};
}  // namespace ocs2
