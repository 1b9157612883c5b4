// TEST INFRASTRUCTURE (oracle/_ref build only).  Visualisation stand-in (shadows the reference header, which needs tf / RViz).
#pragma once
#include <ros/ros.h>
#include <ocs2_mpc/MPC_MRT_Interface.h>
#include <ocs2_pinocchio_interface/PinocchioEndEffectorKinematics.h>
#include <legged_interface/foot_planner/SwingTrajectoryPlanner.h>
namespace legged {
class LeggedRobotVisualizer {
 public:
  LeggedRobotVisualizer(ocs2::PinocchioInterface, ocs2::CentroidalModelInfo, const ocs2::PinocchioEndEffectorKinematics&, ros::NodeHandle&) {}
  template <class P> void update(const ocs2::SystemObservation&, const ocs2::PrimalSolution&, const ocs2::CommandData&, const P&) {}
};
}  // namespace legged
