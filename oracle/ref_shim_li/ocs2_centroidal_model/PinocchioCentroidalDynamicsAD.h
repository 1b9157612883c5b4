// TEST INFRASTRUCTURE (oracle/_ref build only).  The CppAD-generated centroidal dynamics are not evaluated in this build: the class
// records how the reference constructs it (legged_interface/src/dynamics/LeggedRobotDynamicsAD.cpp) and throws if asked for values.
#pragma once
#include <stdexcept>
#include <string>
#include <ocs2_centroidal_model/CentroidalModelInfo.h>
#include <ocs2_pinocchio_interface/PinocchioInterface.h>
namespace ocs2 {
class PinocchioCentroidalDynamicsAD {
 public:
  PinocchioCentroidalDynamicsAD(const PinocchioInterface&, const CentroidalModelInfo& info, const std::string& modelName,
                                const std::string& modelFolder = "/tmp/ocs2", bool recompileLibraries = true, bool verbose = false)
      : modelName(modelName), modelFolder(modelFolder), recompile(recompileLibraries), verbose(verbose), stateDim(info.stateDim) {}
  vector_t getValue(scalar_t, const vector_t&, const vector_t&) const { throw std::runtime_error("PinocchioCentroidalDynamicsAD stand-in: no values"); }
  VectorFunctionLinearApproximation getLinearApproximation(scalar_t, const vector_t&, const vector_t&) const {
    throw std::runtime_error("PinocchioCentroidalDynamicsAD stand-in: no values");
  }
  std::string modelName, modelFolder;
  bool recompile, verbose;
  size_t stateDim;
};
}  // namespace ocs2
