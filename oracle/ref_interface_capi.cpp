// TEST INFRASTRUCTURE — not shipped, not measured (oracle/_ref/libref_interface.so; only tests/golden/make_ref_interface.py loads it).
// The reference's own problem set-up compiled where it lies and EXECUTED on the reference's own task.info / reference.info:
//   legged_interface/src/LeggedInterface.cpp (whole: constructor :50-100, setupOptimalControlProblem :105-162 and every helper below
//   them), src/common/ModelSettings.cpp, src/gait/ModeSequenceTemplate.cpp, src/dynamics/LeggedRobotDynamicsAD.cpp, the constraint /
//   cost / initializer / reference-manager sources it instantiates.
// What comes out is the optimal control problem AS THE REFERENCE ASSEMBLES IT: the named terms of every collection in the order they
// are added, and the parameters each was built with (Q, R = blkdiag(R_force, J' R_task J) at the initial state, friction-cone
// configuration and barrier, the limit barriers and their bounds, the zero-velocity / swing-reference constraint configurations,
// model settings, gait schedule, swing-planner settings).  The OCS2 classes behind it are holders (ref_shim_li/): nothing is solved.
// The rigid-body model behind pinocchio (joint limits, mass, contact-point Jacobians) is the hb_model the generator feeds.
#include <pinocchio/fwd.hpp>
#include <cstdio>
#include <cstring>
#include <sstream>
#include <string>
// (the reference keeps the configurations it builds in private members without getters; this translation unit only reads them)
#define private public
#define protected public
#include "legged_interface/LeggedInterface.h"
#include "legged_interface/LeggedRobotPreComputation.h"
#include "legged_interface/constraint/FrictionConeConstraint.h"
#include "legged_interface/constraint/LeggedSelfCollisionConstraint.h"
#include "legged_interface/constraint/NormalVelocityConstraintCppAd.h"
#include "legged_interface/constraint/XYReferenceConstraintCppAd.h"
#include "legged_interface/constraint/ZeroForceConstraint.h"
#include "legged_interface/constraint/ZeroVelocityConstraintCppAd.h"
#include "legged_interface/cost/LeggedRobotQuadraticTrackingCost.h"
#include "legged_interface/dynamics/LeggedRobotDynamicsAD.h"
#include "legged_interface/initialization/LeggedRobotInitializer.h"
#undef private
#undef protected
#include <ocs2_core/constraint/LinearStateInputConstraint.h>
#include <ocs2_core/penalties/penalties/DoubleSidedPenalty.h>
#include <ocs2_core/soft_constraint/StateInputSoftConstraint.h>
#include <ocs2_core/soft_constraint/StateSoftConstraint.h>
#include <ocs2_centroidal_model/FactoryFunctions.h>
#include <ros/ros.h>

namespace {
using namespace ocs2;
using namespace ocs2::legged_robot;
struct J {
  std::ostringstream os;
  J() { os.precision(17); }
};
void num(J& j, double v) { j.os << v; }
void str(J& j, const std::string& s) { j.os << '"' << s << '"'; }
template <class M> void mat(J& j, const M& m) {
  j.os << '[';
  for (int r = 0; r < int(m.rows()); ++r) {
    j.os << (r ? "," : "") << '[';
    for (int c = 0; c < int(m.cols()); ++c) { j.os << (c ? "," : ""); num(j, m(r, c)); }
    j.os << ']';
  }
  j.os << ']';
}
template <class V> void vec(J& j, const V& v) {
  j.os << '[';
  for (int i = 0; i < int(v.size()); ++i) { j.os << (i ? "," : ""); num(j, v(i)); }
  j.os << ']';
}
template <class T> void list(J& j, const std::vector<T>& v) {
  j.os << '[';
  for (size_t i = 0; i < v.size(); ++i) j.os << (i ? "," : "") << v[i];
  j.os << ']';
}
void strlist(J& j, const std::vector<std::string>& v) {
  j.os << '[';
  for (size_t i = 0; i < v.size(); ++i) { j.os << (i ? "," : ""); str(j, v[i]); }
  j.os << ']';
}
void penalty(J& j, const PenaltyBase* p) {
  j.os << "{\"type\":"; str(j, p->name());
  if (const auto* q = dynamic_cast<const QuadraticPenalty*>(p)) { j.os << ",\"scale\":"; num(j, q->scale); }
  if (const auto* r = dynamic_cast<const RelaxedBarrierPenalty*>(p)) { j.os << ",\"mu\":"; num(j, r->config.mu); j.os << ",\"delta\":"; num(j, r->config.delta); }
  if (const auto* d = dynamic_cast<const DoubleSidedPenalty*>(p)) {
    j.os << ",\"lower\":"; num(j, d->lowerBound); j.os << ",\"upper\":"; num(j, d->upperBound); j.os << ",\"penalty\":"; penalty(j, d->penalty.get());
  }
  j.os << '}';
}
void ee_config(J& j, const EndEffectorLinearConstraint& c) {
  j.os << "{\"rows\":" << c.numConstraints_ << ",\"b\":"; vec(j, c.config_.b);
  j.os << ",\"Ax\":"; mat(j, c.config_.Ax); j.os << ",\"Av\":"; mat(j, c.config_.Av);
  j.os << ",\"end_effectors\":"; strlist(j, c.endEffectorKinematicsPtr_->getIds()); j.os << '}';
}
void constraint(J& j, const StateInputConstraint* c) {
  j.os << "{\"order\":" << (c->getOrder() == ConstraintOrder::Linear ? "\"Linear\"" : "\"Quadratic\"");
  if (const auto* f = dynamic_cast<const FrictionConeConstraint*>(c)) {
    j.os << ",\"type\":\"FrictionConeConstraint\",\"contact\":" << f->contactPointIndex_ << ",\"frictionCoefficient\":"; num(j, f->config_.frictionCoefficient);
    j.os << ",\"regularization\":"; num(j, f->config_.regularization); j.os << ",\"gripperForce\":"; num(j, f->config_.gripperForce);
    j.os << ",\"hessianDiagonalShift\":"; num(j, f->config_.hessianDiagonalShift);
  } else if (const auto* z = dynamic_cast<const ZeroForceConstraint*>(c)) {
    j.os << ",\"type\":\"ZeroForceConstraint\",\"contact\":" << z->contactPointIndex_;
  } else if (const auto* v = dynamic_cast<const ZeroVelocityConstraintCppAd*>(c)) {
    j.os << ",\"type\":\"ZeroVelocityConstraintCppAd\",\"contact\":" << v->contactPointIndex_ << ",\"config\":"; ee_config(j, *v->eeLinearConstraintPtr_);
  } else if (const auto* n = dynamic_cast<const NormalVelocityConstraintCppAd*>(c)) {
    j.os << ",\"type\":\"NormalVelocityConstraintCppAd\",\"contact\":" << n->contactPointIndex_ << ",\"config\":"; ee_config(j, *n->eeLinearConstraintPtr_);
  } else if (const auto* x = dynamic_cast<const XYReferenceConstraintCppAd*>(c)) {
    j.os << ",\"type\":\"XYReferenceConstraintCppAd\",\"contact\":" << x->contactPointIndex_ << ",\"config\":"; ee_config(j, *x->eeLinearConstraintPtr_);
  } else if (const auto* l = dynamic_cast<const LinearStateInputConstraint*>(c)) {
    j.os << ",\"type\":\"LinearStateInputConstraint\",\"e\":"; vec(j, l->e); j.os << ",\"C\":"; mat(j, l->C); j.os << ",\"D\":"; mat(j, l->D);
  } else {
    j.os << ",\"type\":\"?\"";
  }
  j.os << '}';
}
void cost(J& j, const StateInputCost* c) {
  if (const auto* q = dynamic_cast<const LeggedRobotStateInputQuadraticCost*>(c)) {
    j.os << "{\"type\":\"LeggedRobotStateInputQuadraticCost\",\"Q\":"; mat(j, q->weightQ()); j.os << ",\"R\":"; mat(j, q->weightR()); j.os << '}';
  } else if (const auto* s = dynamic_cast<const StateInputSoftConstraint*>(c)) {
    j.os << "{\"type\":\"StateInputSoftConstraint\",\"per_row\":" << (s->per_row ? "true" : "false") << ",\"constraint\":"; constraint(j, s->constraint.get());
    j.os << ",\"penalties\":[";
    for (size_t i = 0; i < s->penalties.size(); ++i) { j.os << (i ? "," : ""); penalty(j, s->penalties[i].get()); }
    j.os << "]}";
  } else {
    j.os << "{\"type\":\"?\"}";
  }
}
template <class C, class F> void collection(J& j, const char* name, const C& col, F item) {
  j.os << '"' << name << "\":[";
  for (size_t i = 0; i < col.terms.size(); ++i) {
    j.os << (i ? "," : "") << "{\"name\":"; str(j, col.terms[i].first); j.os << ",\"term\":"; item(j, col.terms[i].second.get()); j.os << '}';
  }
  j.os << ']';
}
}  // namespace

extern "C" {
// Runs LeggedInterface(task, urdf, reference) + setupOptimalControlProblem and writes the assembled problem as JSON into `out`
// (capacity `cap`).  Returns the length written, -1 if it does not fit, -2 on an exception (message in `out`).
int refli_run(const hb_model* model, const char* task, const char* urdf, const char* reference, char* out, int cap) {
  try {
    ref_li_feed::model() = model;
    ::ros::ref_shim::string_params()["/referenceFile"] = reference;  // what the launch file puts on the parameter server (SwitchedModelReferenceManager.cpp:105-106)
    legged::LeggedInterface li(task, urdf, reference);
    li.setupOptimalControlProblem(task, urdf, reference, false);
    const OptimalControlProblem& p = li.getOptimalControlProblem();
    J j;
    j.os << '{';
    collection(j, "cost", *p.costPtr, cost); j.os << ',';
    collection(j, "softConstraint", *p.softConstraintPtr, cost); j.os << ',';
    collection(j, "equalityConstraint", *p.equalityConstraintPtr, constraint); j.os << ',';
    collection(j, "inequalityConstraint", *p.inequalityConstraintPtr, constraint); j.os << ',';
    collection(j, "stateSoftConstraint", *p.stateSoftConstraintPtr, [](J& jj, const StateCost* c) {
      const auto* s = dynamic_cast<const StateSoftConstraint*>(c);
      const auto* sc = s ? dynamic_cast<const legged::LeggedSelfCollisionConstraint*>(s->constraint.get()) : nullptr;
      jj.os << "{\"type\":" << (sc ? "\"LeggedSelfCollisionConstraint\"" : "\"?\"");
      if (sc) {
        jj.os << ",\"minimumDistance\":"; num(jj, sc->minimumDistance);
        jj.os << ",\"num_pairs\":" << sc->geometry.getNumCollisionPairs() << ",\"link_pairs\":[";
        for (size_t i = 0; i < sc->geometry.linkPairs.size(); ++i) { jj.os << (i ? "," : "") << '['; str(jj, sc->geometry.linkPairs[i].first); jj.os << ','; str(jj, sc->geometry.linkPairs[i].second); jj.os << ']'; }
        jj.os << "],\"penalty\":"; penalty(jj, s->penalty.get());
      }
      jj.os << '}';
    });
    const auto* dyn = dynamic_cast<const LeggedRobotDynamicsAD*>(p.dynamicsPtr.get());
    j.os << ",\"dynamics\":{\"type\":" << (dyn ? "\"LeggedRobotDynamicsAD\"" : "\"?\"");
    if (dyn) { j.os << ",\"modelName\":"; str(j, dyn->pinocchioCentroidalDynamicsAd_.modelName); }
    j.os << "},\"preComputation\":" << (dynamic_cast<const LeggedRobotPreComputation*>(p.preComputationPtr.get()) ? "\"LeggedRobotPreComputation\"" : "\"?\"");
    j.os << ",\"initializer\":" << (dynamic_cast<const LeggedRobotInitializer*>(&li.getInitializer()) ? "\"LeggedRobotInitializer\"" : "\"?\"");
    j.os << ",\"rollout\":" << (dynamic_cast<const TimeTriggeredRollout*>(&li.getRollout()) ? "\"TimeTriggeredRollout\"" : "\"?\"");
    const ModelSettings& ms = li.modelSettings();
    j.os << ",\"modelSettings\":{\"positionErrorGain\":"; num(j, ms.positionErrorGain); j.os << ",\"phaseTransitionStanceTime\":"; num(j, ms.phaseTransitionStanceTime);
    j.os << ",\"jointNames\":"; strlist(j, ms.jointNames); j.os << ",\"contactNames3DoF\":"; strlist(j, ms.contactNames3DoF);
    j.os << ",\"contactNames6DoF\":"; strlist(j, ms.contactNames6DoF); j.os << '}';
    const CentroidalModelInfo& info = li.getCentroidalModelInfo();
    j.os << ",\"centroidalModelInfo\":{\"type\":" << int(info.centroidalModelType) << ",\"stateDim\":" << info.stateDim << ",\"inputDim\":" << info.inputDim
         << ",\"actuatedDofNum\":" << info.actuatedDofNum << ",\"numThreeDofContacts\":" << info.numThreeDofContacts << ",\"robotMass\":"; num(j, info.robotMass);
    j.os << ",\"qPinocchioNominal\":"; vec(j, info.qPinocchioNominal); j.os << '}';
    j.os << ",\"initialState\":"; vec(j, li.getInitialState());
    j.os << ",\"settings_blocks\":{\"mpc\":"; str(j, li.mpcSettings().block); j.os << ",\"ddp\":"; str(j, li.ddpSettings().block); j.os << ",\"sqp\":"; str(j, li.sqpSettings().block);
    j.os << ",\"ipm\":"; str(j, li.ipmSettings().block); j.os << ",\"rollout\":"; str(j, li.rolloutSettings().block); j.os << '}';
    // the gait schedule the reference manager starts from: mode schedule over [0, 3] and the template behind it
    const auto rm = li.getSwitchedModelReferenceManagerPtr();
    const ModeSchedule sched = rm->getGaitSchedule()->getModeSchedule(0.0, 3.0);
    j.os << ",\"modeSchedule_0_3\":{\"eventTimes\":"; list(j, sched.eventTimes); j.os << ",\"modeSequence\":"; list(j, sched.modeSequence); j.os << '}';
    const SwingTrajectoryPlanner::Config& sc = rm->getSwingTrajectoryPlanner()->config_;
    j.os << ",\"swingConfig\":{\"liftOffVelocity\":"; num(j, sc.liftOffVelocity); j.os << ",\"touchDownVelocity\":"; num(j, sc.touchDownVelocity);
    j.os << ",\"swingHeight\":"; num(j, sc.swingHeight); j.os << ",\"swingTimeScale\":"; num(j, sc.swingTimeScale); j.os << '}';
    j.os << ",\"factory_calls\":{\"urdf\":"; str(j, ref_li_feed::calls().urdf); j.os << ",\"jointNames\":"; strlist(j, ref_li_feed::calls().jointNames);
    j.os << ",\"contacts3\":"; strlist(j, ref_li_feed::calls().contacts3); j.os << "}}";
    const std::string s = j.os.str();
    if (int(s.size()) + 1 > cap) return -1;
    std::memcpy(out, s.c_str(), s.size() + 1);
    return int(s.size());
  } catch (const std::exception& e) {
    std::snprintf(out, size_t(cap), "%s", e.what());
    return -2;
  }
}
}
