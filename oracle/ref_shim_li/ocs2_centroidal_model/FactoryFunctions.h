// TEST INFRASTRUCTURE (oracle/_ref build only).  [OCS2-knowledge: centroidal_model factory functions as LeggedInterface::setupModel
// calls them (LeggedInterface.cpp:186-201).]  The URDF is not parsed here: the rigid-body model (joint limits, mass, the kinematics
// behind the pinocchio stand-in) is the hb_model the generator feeds (ref_li_feed::model()); `centroidalModelType` and
// `defaultJointState` are read from the files the reference names.
#pragma once
#include <stdexcept>
#include <string>
#include <vector>
#include <ocs2_centroidal_model/CentroidalModelInfo.h>
#include <ocs2_core/misc/LoadData.h>
#include <ocs2_pinocchio_interface/PinocchioInterface.h>
namespace ref_li_feed {
inline const hb_model*& model() { static const hb_model* m = nullptr; return m; }
struct Calls { std::string urdf; std::vector<std::string> jointNames, contacts3, contacts6; };
inline Calls& calls() { static Calls c; return c; }
}  // namespace ref_li_feed
namespace ocs2 {
namespace centroidal_model {
inline PinocchioInterface createPinocchioInterface(const std::string& urdfFile, const std::vector<std::string>& jointNames) {
  const hb_model* hb = ref_li_feed::model();
  if (!hb) throw std::runtime_error("createPinocchioInterface stand-in: no model fed");
  ref_li_feed::calls().urdf = urdfFile;
  ref_li_feed::calls().jointNames = jointNames;
  PinocchioInterface pi;
  pinocchio::Model& m = pi.mutableModel();
  m.hb = hb;
  m.nq = m.nv = 6 + HB_NJ;
  m.lowerPositionLimit.setZero(m.nq);
  m.upperPositionLimit.setZero(m.nq);
  m.velocityLimit.setZero(m.nv);
  for (int j = 0; j < HB_NJ; ++j) {
    m.lowerPositionLimit(6 + j) = hb->q_lower[j];
    m.upperPositionLimit(6 + j) = hb->q_upper[j];
    m.velocityLimit(6 + j) = hb->qd_limit[j];
  }
  return pi;
}
inline CentroidalModelType loadCentroidalType(const std::string& taskFile, const std::string& field = "centroidalModelType") {
  size_t t = 0;
  loadData::loadCppDataType(taskFile, field, t);
  return static_cast<CentroidalModelType>(t);
}
inline vector_t loadDefaultJointState(size_t numJointState, const std::string& referenceFile, const std::string& field = "defaultJointState") {
  vector_t v = vector_t::Zero(int(numJointState));
  loadData::loadEigenMatrix(referenceFile, field, v);
  return v;
}
inline CentroidalModelInfo createCentroidalModelInfo(PinocchioInterface& interface, const CentroidalModelType& type, const vector_t& nominalJointAngles,
                                                     const std::vector<std::string>& threeDofContactNames, const std::vector<std::string>& sixDofContactNames) {
  ref_li_feed::calls().contacts3 = threeDofContactNames;
  ref_li_feed::calls().contacts6 = sixDofContactNames;
  interface.mutableModel().frame_names = threeDofContactNames;   // frame id = contact index, as everywhere in the pinocchio stand-in
  CentroidalModelInfo info;
  info.centroidalModelType = type;
  info.numThreeDofContacts = threeDofContactNames.size();
  info.numSixDofContacts = sixDofContactNames.size();
  info.endEffectorFrameIndices.clear();
  for (size_t i = 0; i < threeDofContactNames.size(); ++i) info.endEffectorFrameIndices.push_back(i);
  info.generalizedCoordinatesNum = size_t(interface.getModel().nq);
  info.actuatedDofNum = info.generalizedCoordinatesNum - 6;
  info.stateDim = info.generalizedCoordinatesNum + 6;
  info.inputDim = info.actuatedDofNum + 3 * info.numThreeDofContacts + 6 * info.numSixDofContacts;
  info.robotMass = 0.0;
  for (int b = 0; b < HB_NBODY; ++b) info.robotMass += interface.getModel().hb->mass[b];
  info.qPinocchioNominal = vector_t::Zero(int(info.generalizedCoordinatesNum));
  for (int j = 0; j < int(info.actuatedDofNum); ++j) info.qPinocchioNominal(6 + j) = nominalJointAngles(j);
  return info;
}
}  // namespace centroidal_model
}  // namespace ocs2
