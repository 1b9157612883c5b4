// TEST INFRASTRUCTURE (oracle/_ref build only).  Visualisation stand-in (shadows the reference header, which needs RViz message types).
#pragma once
#include <ros/ros.h>
#include <ocs2_mpc/MPC_MRT_Interface.h>
#include <ocs2_centroidal_model/CentroidalModelPinocchioMapping.h>
namespace ocs2 { class PinocchioGeometryInterface {}; }
namespace legged {
class LeggedSelfCollisionVisualization {
 public:
  LeggedSelfCollisionVisualization(ocs2::PinocchioInterface, ocs2::PinocchioGeometryInterface, const ocs2::CentroidalModelPinocchioMapping&, ros::NodeHandle&) {}
  void update(const ocs2::SystemObservation&) {}
};
}  // namespace legged
