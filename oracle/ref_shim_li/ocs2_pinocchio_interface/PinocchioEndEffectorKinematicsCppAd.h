// TEST INFRASTRUCTURE (oracle/_ref build only).  The CppAD end-effector kinematics as LeggedInterface::getEeKinematicsPtr constructs
// them (LeggedInterface.cpp:393-412): a holder of the constructor arguments; the constraints built on it are never evaluated here.
#pragma once
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>
#include <ocs2_centroidal_model/CentroidalModelPinocchioMapping.h>
#include <ocs2_pinocchio_interface/PinocchioInterface.h>
#include <ocs2_robotic_tools/end_effector/EndEffectorKinematics.h>
namespace ocs2 {
using PinocchioInterfaceCppAd = PinocchioInterface;
using CentroidalModelPinocchioMappingCppAd = CentroidalModelPinocchioMapping;
class PinocchioEndEffectorKinematicsCppAd final : public EndEffectorKinematics<scalar_t> {
 public:
  using update_pinocchio_interface_callback = std::function<void(const ad_vector_t&, PinocchioInterfaceCppAd&)>;
  PinocchioEndEffectorKinematicsCppAd(const PinocchioInterface&, const CentroidalModelPinocchioMappingCppAd&, std::vector<std::string> endEffectorIds,
                                      size_t stateDim, size_t inputDim, update_pinocchio_interface_callback, const std::string& modelName,
                                      const std::string& modelFolder = "/tmp/ocs2", bool recompileLibraries = true, bool verbose = false)
      : ids_(std::move(endEffectorIds)), stateDim(stateDim), inputDim(inputDim), modelName(modelName), modelFolder(modelFolder),
        recompile(recompileLibraries), verbose(verbose) {}
  PinocchioEndEffectorKinematicsCppAd* clone() const override { return new PinocchioEndEffectorKinematicsCppAd(*this); }
  const std::vector<std::string>& getIds() const override { return ids_; }
  std::vector<vector3_t> getPosition(const vector_t&) const override { throw std::runtime_error("kinematics stand-in: no values"); }
  std::vector<vector3_t> getVelocity(const vector_t&, const vector_t&) const override { throw std::runtime_error("kinematics stand-in: no values"); }
  std::vector<VectorFunctionLinearApproximation> getPositionLinearApproximation(const vector_t&) const override { throw std::runtime_error("kinematics stand-in: no values"); }
  std::vector<VectorFunctionLinearApproximation> getVelocityLinearApproximation(const vector_t&, const vector_t&) const override { throw std::runtime_error("kinematics stand-in: no values"); }
  std::vector<std::string> ids_;
  size_t stateDim, inputDim;
  std::string modelName, modelFolder;
  bool recompile, verbose;
};
}  // namespace ocs2
