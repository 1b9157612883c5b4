// hunter_hip.hpp — header-only C++14 host adapter over the C ABI (hunter_hip.h).
//
// The reference's host code for this path is C++ (ROS1 / OCS2 / Eigen).  None of those are needed here: this
// header mirrors the *operator interface* the reference's control loop talks to, with the same names, argument
// order and error behaviour, over plain std::vector<double>:
//
//   legged::WbcBase::update(stateDesired, inputDesired, rbdStateMeasured, mode, period)
//                                   legged_wbc/include/legged_wbc/WbcBase.h:43-44, WeightedWbc.cpp:18-66
//   ocs2::MPC_MRT_Interface::{setCurrentObservation, advanceMpc, updatePolicy, evaluatePolicy, resetMpcNode}
//                                   call sites legged_controllers/src/LeggedController.cpp:144-159,406,464
//   LeggedController::update        legged_controllers/src/LeggedController.cpp:137-185 (policy + WBC part)
//
// Errors: the C ABI never throws and returns negative hb_status codes; the reference's C++ throws
// std::runtime_error / std::invalid_argument (e.g. LeggedInterface.cpp:68-76) — this adapter converts.
// A WBC QP that does not converge is NOT an exception in the reference: WeightedWbc::update prints
// "ERROR: WeightWBC Not Solved!!!" and returns the previous solution (WeightedWbc.cpp:57-65); same here.
//
// One Context = one GPU = `batch` independent robot instances.  batch == 1 reproduces the reference's shapes.
#pragma once

#include <cmath>
#include <cstdio>
#include <cstring>
#include <deque>
#include <numeric>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "hunter_hip.h"
#include "hunter_ingest.hpp"
#include "hunter_lcm.h"

namespace hunter_hip {

using scalar_t = double;
using vector_t = std::vector<double>;

class Error : public std::runtime_error {
 public:
  Error(int32_t status, const std::string& what) : std::runtime_error(what), status_(status) {}
  int32_t status() const { return status_; }

 private:
  int32_t status_;
};

// ocs2::SystemObservation restricted to what the path reads (LeggedController.cpp:280-349).
struct SystemObservation {
  scalar_t time = 0.0;
  vector_t state = vector_t(HB_NX, 0.0);
  vector_t input = vector_t(HB_NU, 0.0);
  size_t mode = 3;
};

// Node tables of one batch (what SwitchedModelReferenceManager::modifyReferences + the OCS2 time discretisation
// hand to the solver, see hb_mpc_set_references).  Strides are max_nodes (+1 for t).
struct ReferenceTables {
  std::vector<int32_t> nNodes;  // [batch]
  vector_t t;                   // [batch][maxNodes + 1]
  std::vector<int32_t> mode;    // [batch][maxNodes]
  vector_t xRef;                // [batch][maxNodes][22]
  vector_t swingRef;            // [batch][maxNodes][4][6]
};

// The packaged parameter image (data/hunter_params.bin, written by tools/make_hunter_params.py; format HB02, hunter_ingest.hpp):
// the flattened URDF + task.info / reference.info the reference reads at LeggedController::init.  hunter_ingest.hpp also reads
// the reference's files themselves (loadParameters).
inline void loadPackagedParameters(const std::string& path, hb_model& model, hb_config& config) {
  const Parameters p = loadParametersBlob(path);
  model = p.model;
  config = p.config;
}

class Context {
 public:
  Context(const hb_model& model, const hb_config& config, int batch, int maxNodes, int device = 0)
      : batch_(batch), maxNodes_(maxNodes) {
    hb_ctx* raw = nullptr;
    const int32_t rc = hb_create(&model, &config, batch, maxNodes, device, &raw);
    if (rc != HB_OK) throw Error(rc, std::string("[hunter_hip] hb_create failed: ") + hb_last_error(nullptr));
    ctx_.reset(raw, [](hb_ctx* p) { hb_destroy(p); });
  }
  hb_ctx* get() const { return ctx_.get(); }
  int batch() const { return batch_; }
  int maxNodes() const { return maxNodes_; }
  void check(int32_t rc, const char* what) const {
    if (rc != HB_OK) throw Error(rc, std::string("[hunter_hip] ") + what + " failed: " + hb_last_error(ctx_.get()));
  }
  hb_stats stats() const {
    hb_stats st;
    check(hb_get_stats(ctx_.get(), &st), "hb_get_stats");
    return st;
  }

 private:
  std::shared_ptr<hb_ctx> ctx_;
  int batch_, maxNodes_;
};

// ---- MPC side: the calls LeggedController makes on ocs2::MPC_MRT_Interface --------------------------------------
class MpcMrtInterface {
 public:
  explicit MpcMrtInterface(Context ctx) : ctx_(std::move(ctx)), x0_(size_t(ctx_.batch()) * HB_NX, 0.0) {}

  // reference manager output for the next solve (SwitchedModelReferenceManager::modifyReferences)
  void setReferences(const ReferenceTables& r, int instBegin = 0) {
    const int count = int(r.nNodes.size());
    const size_t n = size_t(count), N = size_t(ctx_.maxNodes());
    if (r.t.size() != n * (N + 1) || r.mode.size() != n * N || r.xRef.size() != n * N * HB_NX ||
        r.swingRef.size() != n * N * HB_NC * HB_SWING_REF)
      throw std::invalid_argument("[hunter_hip] reference tables do not match (batch, maxNodes)");
    ctx_.check(hb_mpc_set_references(ctx_.get(), instBegin, count, r.nNodes.data(), r.t.data(), r.mode.data(), r.xRef.data(),
                                     r.swingRef.data()),
               "hb_mpc_set_references");
  }
  // MPC_MRT_Interface::resetMpcNode (LeggedController.cpp:464): cold start from the given states
  void resetMpcNode(const vector_t& initialStates) {
    requireSize(initialStates, size_t(ctx_.batch()) * HB_NX, "resetMpcNode");
    ctx_.check(hb_mpc_reset(ctx_.get(), initialStates.data()), "hb_mpc_reset");
    x0_ = initialStates;
  }
  // MPC_MRT_Interface::setCurrentObservation (LeggedController.cpp:144); one observation per instance
  void setCurrentObservation(const std::vector<SystemObservation>& obs) {
    if (int(obs.size()) != ctx_.batch()) throw std::invalid_argument("[hunter_hip] one observation per instance expected");
    for (size_t i = 0; i < obs.size(); ++i) {
      requireSize(obs[i].state, HB_NX, "setCurrentObservation");
      std::memcpy(&x0_[i * HB_NX], obs[i].state.data(), HB_NX * sizeof(double));
    }
  }
  void setCurrentObservation(const SystemObservation& obs) { setCurrentObservation(std::vector<SystemObservation>{obs}); }
  // MPC_MRT_Interface::advanceMpc (LeggedController.cpp:406): one SQP solve from the current observation on the library's MPC
  // stream; this (MPC) thread then WAITS for it — advanceMpc is a blocking call in the reference as well — and only then hands
  // the finished policy over.  The order matters for the control thread: hb_mpc_publish makes the WBC stream wait for the copies
  // it enqueues on the MPC stream, so published BEHIND an in-flight solve it would make the next 500 Hz hb_wbc_update wait for
  // that whole solve; published after hb_mpc_get_status (which synchronises the MPC stream) the control thread waits for five
  // short device-to-device copies at most and keeps evaluating the previous policy until then, like LeggedController::update
  // with MPC_MRT_Interface::updatePolicy.  hb_mpc_publish is an MPC-THREAD call (include/hunter_hip.h): it must be ordered with
  // the table updates and the warm start that precede the solve on this thread.
  void advanceMpc() {
    ctx_.check(hb_mpc_solve(ctx_.get(), x0_.data()), "hb_mpc_solve");
    mpcStatus_.resize(size_t(ctx_.batch()));
    ctx_.check(hb_mpc_get_status(ctx_.get(), mpcStatus_.data()), "hb_mpc_get_status");   // returns when the solve has finished
    ctx_.check(hb_mpc_publish(ctx_.get()), "hb_mpc_publish");
  }
  // per-instance status word of the last advanceMpc (HB_INST_OK / HB_INST_MAXITER / HB_INST_NAN)
  const std::vector<int32_t>& mpcStatus() const { return mpcStatus_; }
  // MPC_MRT_Interface::updatePolicy (LeggedController.cpp:154): kept for call-site parity; the hand-over already happened in
  // advanceMpc on the MPC thread, nothing is left to do on the control thread
  void updatePolicy() {}
  // PrimalSolution of the last solve (LeggedController.cpp:269, visualisation)
  void getSolution(vector_t& stateTrajectory, vector_t& inputTrajectory) const {
    const size_t B = size_t(ctx_.batch()), N = size_t(ctx_.maxNodes());
    stateTrajectory.assign(B * (N + 1) * HB_NX, 0.0);
    inputTrajectory.assign(B * N * HB_NU, 0.0);
    ctx_.check(hb_mpc_get_solution(ctx_.get(), 0, ctx_.batch(), stateTrajectory.data(), inputTrajectory.data()), "hb_mpc_get_solution");
  }
  // OCS2 PerformanceIndex of the last iteration per instance: [merit, dynamics SSE, equality SSE, accepted step size]
  vector_t getPerformanceIndices() const {
    vector_t perf(size_t(ctx_.batch()) * 4, 0.0);
    ctx_.check(hb_mpc_get_performance(ctx_.get(), perf.data()), "hb_mpc_get_performance");
    return perf;
  }
  const Context& context() const { return ctx_; }

 private:
  static void requireSize(const vector_t& v, size_t n, const char* who) {
    if (v.size() != n) throw std::invalid_argument(std::string("[hunter_hip] ") + who + ": wrong vector size");
  }
  Context ctx_;
  vector_t x0_;
  std::vector<int32_t> mpcStatus_;
};

// ---- WBC side: legged::WbcBase / WeightedWbc / HierarchicalWbc ----------------------------------------------------
// Which of the two the context runs is hb_config::wbc_type (LeggedController.cpp:85 constructs WeightedWbc).
class Wbc {
 public:
  explicit Wbc(Context ctx)
      : ctx_(std::move(ctx)), last_(size_t(ctx_.batch()) * HB_NWBC, 0.0), status_(size_t(ctx_.batch()), 0),
        stance_(size_t(ctx_.batch()), 0) {}

  void setStanceMode(bool stanceMode) { std::fill(stance_.begin(), stance_.end(), stanceMode ? 1 : 0); }  // WbcBase.h:73-76

  // WbcBase::update for every instance of the batch: x = [qdd(16) F(12) tau(10)] per instance.
  // stateDesired / inputDesired [batch][22], rbdStateMeasured [batch][32], mode [batch].
  const vector_t& update(const vector_t& stateDesired, const vector_t& inputDesired, const vector_t& rbdStateMeasured,
                         const std::vector<int32_t>& mode, scalar_t period) {
    const size_t B = size_t(ctx_.batch());
    if (stateDesired.size() != B * HB_NX || inputDesired.size() != B * HB_NU || rbdStateMeasured.size() != B * HB_NRBD ||
        mode.size() != B)
      throw std::invalid_argument("[hunter_hip] Wbc::update: wrong vector size");
    ctx_.check(hb_wbc_update_direct(ctx_.get(), stateDesired.data(), inputDesired.data(), rbdStateMeasured.data(), mode.data(),
                                    stance_.data(), period, last_.data(), status_.data()),
               "hb_wbc_update_direct");
    reportUnsolved();
    return last_;
  }
  // the reference's single-robot signature (WbcBase.h:43-44)
  vector_t update(const vector_t& stateDesired, const vector_t& inputDesired, const vector_t& rbdStateMeasured, size_t mode,
                  scalar_t period) {
    if (ctx_.batch() != 1) throw std::invalid_argument("[hunter_hip] single-instance update on a batched context");
    return update(stateDesired, inputDesired, rbdStateMeasured, std::vector<int32_t>{int32_t(mode)}, period);
  }
  const std::vector<int32_t>& status() const { return status_; }
  const Context& context() const { return ctx_; }

 protected:
  void reportUnsolved() const {
    // instances that hit the iteration limit keep their previous solution on the device (WeightedWbc.cpp:57-65)
    for (size_t i = 0; i < status_.size(); ++i)
      if (status_[i] != HB_INST_OK) {
        std::cout << "ERROR: WeightWBC Not Solved!!! (instance " << i << ", status " << status_[i] << ")" << std::endl;
        break;
      }
  }
  Context ctx_;
  vector_t last_;
  std::vector<int32_t> status_, stance_;
};

// ---- reference side: ocs2::legged_robot::GaitSchedule + SwitchedModelReferenceManager ------------------------------
// The gait scheduler is a handful of integers and event times per robot; it stays on the host, in C++ like the
// reference's (legged_interface/src/gait/GaitSchedule.cpp:57-161).  Everything it feeds — targets, time grid, swing
// planner, joint-reference IK — runs on the device (hb_refgen_*).
struct ModeSequenceTemplate {       // legged_interface/include/legged_interface/gait/ModeSequenceTemplate.h
  std::vector<scalar_t> switchingTimes;  // N + 1 phase boundaries of one period
  std::vector<int32_t> modeSequence;     // N modes
};
struct ModeSchedule {               // ocs2::ModeSchedule
  std::vector<scalar_t> eventTimes;
  std::vector<int32_t> modeSequence{3};
  int32_t modeAtTime(scalar_t t) const {  // lower_bound, like ModeSchedule::modeAtTime
    size_t i = 0;
    while (i < eventTimes.size() && eventTimes[i] < t) ++i;
    return modeSequence[i];
  }
};
class GaitSchedule {
 public:
  GaitSchedule(ModeSchedule initModeSchedule, ModeSequenceTemplate initModeSequenceTemplate, scalar_t phaseTransitionStanceTime)
      : modeSchedule_(std::move(initModeSchedule)), modeSequenceTemplate_(std::move(initModeSequenceTemplate)),
        phaseTransitionStanceTime_(phaseTransitionStanceTime) {}
  // GaitSchedule::insertModeSequenceTemplate (GaitSchedule.cpp:57-89)
  void insertModeSequenceTemplate(const ModeSequenceTemplate& modeSequenceTemplate, scalar_t startTime, scalar_t finalTime) {
    modeSequenceTemplate_ = modeSequenceTemplate;
    auto& ev = modeSchedule_.eventTimes;
    auto& md = modeSchedule_.modeSequence;
    size_t idx = 0;
    while (idx < ev.size() && ev[idx] < startTime) ++idx;  // lower_bound
    if (idx < ev.size()) {
      ev.erase(ev.begin() + idx, ev.end());
      md.erase(md.begin() + idx + 1, md.end());
    }
    scalar_t pts = phaseTransitionStanceTime_;
    if (!md.empty() && md.back() == kStance) pts = 0.0;
    if (pts > 0.0) {
      ev.push_back(startTime);
      md.push_back(kStance);
    }
    tileModeSequenceTemplate(startTime + pts, finalTime);
  }
  // GaitSchedule::getModeSchedule (GaitSchedule.cpp:94-120): drops the past, tiles the template up to the upper bound
  ModeSchedule getModeSchedule(scalar_t lowerBoundTime, scalar_t upperBoundTime) {
    auto& ev = modeSchedule_.eventTimes;
    auto& md = modeSchedule_.modeSequence;
    size_t idx = 0;
    while (idx < ev.size() && ev[idx] < lowerBoundTime) ++idx;
    if (idx > 0) {
      ev.erase(ev.begin(), ev.begin() + (idx - 1));
      md.erase(md.begin(), md.begin() + (idx - 1));
      md.front() = kStance;
    }
    const scalar_t start = ev.empty() ? lowerBoundTime : ev.back();
    if (!ev.empty()) ev.pop_back();
    md.pop_back();
    tileModeSequenceTemplate(start, upperBoundTime);
    return modeSchedule_;
  }

 private:
  enum : int32_t { kStance = 3 };  // STANCE (MotionPhaseDefinition.h:55-95)
  // GaitSchedule::tileModeSequenceTemplate (GaitSchedule.cpp:125-161)
  void tileModeSequenceTemplate(scalar_t startTime, scalar_t finalTime) {
    auto& ev = modeSchedule_.eventTimes;
    auto& md = modeSchedule_.modeSequence;
    const auto& tp = modeSequenceTemplate_;
    if (tp.modeSequence.empty()) return;
    if (!ev.empty() && startTime <= ev.back())
      throw std::runtime_error("The initial time for template-tiling is not greater than the last event time.");
    ev.push_back(startTime);
    while (ev.back() < finalTime)
      for (size_t i = 0; i < tp.modeSequence.size(); ++i) {
        md.push_back(tp.modeSequence[i]);
        const scalar_t deltaTime = tp.switchingTimes[i + 1] - tp.switchingTimes[i];  // the difference first: the reference's rounding
        ev.push_back(ev.back() + deltaTime);
      }
    md.push_back(kStance);
  }
  ModeSchedule modeSchedule_;
  ModeSequenceTemplate modeSequenceTemplate_;
  scalar_t phaseTransitionStanceTime_;
};

// The per-callback rate limiter of the /cmd_vel subscriber (lastVel_ / changeLimit_,
// legged_controllers/include/legged_controllers/TargetTrajectoriesPublisher.h:97-119): every message moves the filtered
// command by at most (0.1, 0.05, -, 0.3) towards the request; linear z is forced to zero.  One filter per instance.
class CmdVelFilter {
 public:
  // request (vx, vy, yaw rate) -> filtered [vx, vy, 0, yaw rate] (what cmdVelToTargetTrajectories and the device take)
  const scalar_t* operator()(scalar_t vx, scalar_t vy, scalar_t yawRate) {
    step(0, vx, 0.1);
    step(1, vy, 0.05);
    last_[2] = 0.0;
    step(3, yawRate, 0.3);
    return last_;
  }
  const scalar_t* last() const { return last_; }

 private:
  void step(int k, scalar_t want, scalar_t limit) {
    scalar_t d = want - last_[k];
    d = d > 0 ? std::fmin(d, limit) : std::fmax(d, -limit);
    last_[k] += d;
  }
  scalar_t last_[4] = {0.0, 0.0, 0.0, 0.0};
};

// The two templates hard-coded next to the reference manager (SwitchedModelReferenceManager.cpp:55-61).
inline ModeSequenceTemplate stanceTemplate() { return ModeSequenceTemplate{{0.0, 0.5}, {3}}; }
inline ModeSequenceTemplate trotTemplate() { return ModeSequenceTemplate{{0.0, 0.3, 0.6}, {2, 1}}; }

// gaitLevel_ / velAbsHistory_ of one instance: SwitchedModelReferenceManager::{calculateVelAbs, walkGait,
// findInsertModeSequenceTemplateTimer} (SwitchedModelReferenceManager.cpp:173-249).  `update` is what modifyReferences does
// between getModeSchedule and the swing-planner update; a template it inserts takes effect from the NEXT getModeSchedule.
class GaitSelector {
 public:
  int level() const { return level_; }
  scalar_t velAvg() const { return velAvg_; }
  // cmdVel [vx vy vz yawRate] (filtered), observation state x[22], the schedule window this MPC call got, initTime / finalTime
  void update(const scalar_t* cmdVel, const scalar_t* x, const ModeSchedule& window, scalar_t initTime, scalar_t finalTime,
              GaitSchedule& gaitSchedule) {
    // stateTrajectory[0] of cmdVelToTargetTrajectories (TargetTrajectoriesPublisher.cpp:102-130): the command rotated by the
    // observed ZYX angles with the 0.06 dead band (x, ELSE y); its pose knot keeps the yaw and zeroes pitch / roll
    scalar_t R[9];
    rotZyx(x[9], x[10], x[11], R);
    scalar_t v[3] = {R[0] * cmdVel[0] + R[1] * cmdVel[1] + R[2] * cmdVel[2], R[3] * cmdVel[0] + R[4] * cmdVel[1] + R[5] * cmdVel[2],
                     R[6] * cmdVel[0] + R[7] * cmdVel[1] + R[8] * cmdVel[2]};
    if (std::fabs(v[0]) < 0.06) v[0] = 0.0;
    else if (std::fabs(v[1]) < 0.06) v[1] = 0.0;
    // calculateVelAbs (:229-249): command rotated by the first target's angles (yaw, 0, 0), z zeroed, yaw rate / 3, averaged
    // with the first target's momentum entries [v, 0] treated the same way; 50-sample history
    rotZyx(x[9], 0.0, 0.0, R);
    const scalar_t c0 = R[0] * cmdVel[0] + R[1] * cmdVel[1] + R[2] * cmdVel[2], c1 = R[3] * cmdVel[0] + R[4] * cmdVel[1] + R[5] * cmdVel[2];
    const scalar_t m[4] = {0.5 * c0 + 0.5 * v[0], 0.5 * c1 + 0.5 * v[1], 0.0, 0.5 * (cmdVel[3] / 3.0) + 0.5 * (0.0 / 3.0)};
    history_.push_front(std::sqrt(m[0] * m[0] + m[1] * m[1] + m[2] * m[2] + m[3] * m[3]));
    while (history_.size() > 50) history_.pop_back();
    velAvg_ = std::accumulate(history_.begin(), history_.end(), 0.0) / scalar_t(history_.size());
    // walkGait (:185-217)
    int want = level_;
    if (velAvg_ <= 0.02) want = 0;
    else if (velAvg_ > 0.03 && velAvg_ < 0.4) want = 1;
    else if (velAvg_ >= 0.4) want = 3;
    if (want == level_) return;
    level_ = want;
    if (want == 3) return;  // "flying trot": the reference only prints (:206-214)
    size_t id = 0;           // findInsertModeSequenceTemplateTimer: first event of the window >= initTime
    while (id < window.eventTimes.size() && window.eventTimes[id] < initTime) ++id;
    if (id >= window.eventTimes.size()) return;  // (the reference reads past the end here)
    gaitSchedule.insertModeSequenceTemplate(want == 0 ? stanceTemplate() : trotTemplate(), window.eventTimes[id], finalTime);
  }

 private:
  static void rotZyx(scalar_t z, scalar_t y, scalar_t xr, scalar_t* R) {
    const scalar_t cz = std::cos(z), sz = std::sin(z), cy = std::cos(y), sy = std::sin(y), cx = std::cos(xr), sx = std::sin(xr);
    R[0] = cz * cy; R[1] = cz * sy * sx - sz * cx; R[2] = cz * sy * cx + sz * sx;
    R[3] = sz * cy; R[4] = sz * sy * sx + cz * cx; R[5] = sz * sy * cx - cz * sx;
    R[6] = -sy;     R[7] = cy * sx;                R[8] = cy * cx;
  }
  int level_ = 0;
  scalar_t velAvg_ = 0.0;
  std::deque<scalar_t> history_;
};

// SwitchedModelReferenceManager::modifyReferences (SwitchedModelReferenceManager.cpp:136-171) for the whole batch: the
// pre-solver hook the MPC thread runs before every advanceMpc().  One GaitSchedule (+ gait selector) per instance.
class ReferenceManager {
 public:
  ReferenceManager(Context ctx, const hb_refgen_config& settings, std::vector<GaitSchedule> gaitSchedules)
      : ctx_(std::move(ctx)), gaits_(std::move(gaitSchedules)), selectors_(gaits_.size()) {
    if (int(gaits_.size()) != ctx_.batch()) throw std::invalid_argument("[hunter_hip] one GaitSchedule per instance expected");
    ctx_.check(hb_refgen_reset(ctx_.get(), &settings, nullptr), "hb_refgen_reset");
  }
  GaitSchedule& gaitSchedule(int instance) { return gaits_.at(size_t(instance)); }
  GaitSelector& gaitSelector(int instance) { return selectors_.at(size_t(instance)); }
  // gaitType_ == 0 of the reference (the default, /gait_type topic): the gait of every instance follows its averaged command
  // speed (walkGait).  It needs the observation on the host, i.e. preSolverRun must be given `observation`.
  void setWalkGaitSelection(bool on) { walkGait_ = on; }
  // initTime [batch], cmdVel [batch][4] = (vx, vy, vz, yaw rate); observation [batch][22] or nullptr = the resident one
  void preSolverRun(const vector_t& initTime, scalar_t timeHorizon, const vector_t& cmdVel, const vector_t* observation = nullptr) {
    const size_t B = size_t(ctx_.batch());
    if (initTime.size() != B || cmdVel.size() != B * 4 || (observation && observation->size() != B * HB_NX))
      throw std::invalid_argument("[hunter_hip] ReferenceManager::preSolverRun: wrong vector size");
    std::vector<int32_t> nEvents(B), modes(B * (HB_MAX_EVENTS + 1), 3);
    vector_t events(B * HB_MAX_EVENTS, 0.0);
    for (size_t i = 0; i < B; ++i) {
      // the reference asks for [t - T, t + 2T] (SwitchedModelReferenceManager.cpp:147)
      const ModeSchedule ms = gaits_[i].getModeSchedule(initTime[i] - timeHorizon, initTime[i] + 2.0 * timeHorizon);
      if (ms.eventTimes.size() > size_t(HB_MAX_EVENTS)) throw std::invalid_argument("[hunter_hip] mode schedule longer than HB_MAX_EVENTS");
      if (walkGait_) {
        if (!observation) throw std::invalid_argument("[hunter_hip] walk-gait selection needs the observation on the host");
        selectors_[i].update(cmdVel.data() + 4 * i, observation->data() + HB_NX * i, ms, initTime[i], initTime[i] + timeHorizon, gaits_[i]);
      }
      nEvents[i] = int32_t(ms.eventTimes.size());
      for (size_t e = 0; e < ms.eventTimes.size(); ++e) events[i * HB_MAX_EVENTS + e] = ms.eventTimes[e];
      for (size_t e = 0; e < ms.modeSequence.size(); ++e) modes[i * (HB_MAX_EVENTS + 1) + e] = ms.modeSequence[e];
    }
    ctx_.check(hb_refgen_set_schedule(ctx_.get(), 0, ctx_.batch(), nEvents.data(), events.data(), modes.data()), "hb_refgen_set_schedule");
    status_.resize(B);
    ctx_.check(hb_refgen_update(ctx_.get(), initTime.data(), timeHorizon, observation ? observation->data() : nullptr, cmdVel.data(),
                                status_.data()),
               "hb_refgen_update");
    for (size_t i = 0; i < B; ++i)
      if (status_[i] != 0) throw std::runtime_error("[hunter_hip] reference generation failed for instance " + std::to_string(i));
  }

 private:
  Context ctx_;
  std::vector<GaitSchedule> gaits_;
  std::vector<GaitSelector> selectors_;
  bool walkGait_ = false;
  std::vector<int32_t> status_;
};

// ---- estimator side: legged::KalmanFilterEstimate / StateEstimateBase ---------------------------------------------
// updateJointStates / updateContact / updateImu / update(time, period) (LeggedController.cpp:323-327) in one call per
// tick for the whole batch; returns rbdState (what WbcBase::update takes) and fills the centroidal observation states
// (LeggedController.cpp:331-334).
class KalmanFilterEstimate {
 public:
  KalmanFilterEstimate(Context ctx, const hb_estimator_config& settings, const vector_t* xHat0 = nullptr) : ctx_(std::move(ctx)) {
    if (xHat0 && xHat0->size() != size_t(ctx_.batch()) * 18) throw std::invalid_argument("[hunter_hip] xHat0: wrong vector size");
    ctx_.check(hb_estimator_reset(ctx_.get(), &settings, xHat0 ? xHat0->data() : nullptr), "hb_estimator_reset");
  }
  // quat [batch][4] (x y z w), angularVelLocal / linearAccelLocal [batch][3], jointPos / jointVel [batch][10],
  // contactFlag [batch][4].  toResident feeds the device-resident inputs of hb_step_resident.
  const vector_t& update(scalar_t period, const vector_t& quat, const vector_t& angularVelLocal, const vector_t& linearAccelLocal,
                         const vector_t& jointPos, const vector_t& jointVel, const std::vector<int32_t>& contactFlag,
                         bool toResident = false) {
    const size_t B = size_t(ctx_.batch());
    if (quat.size() != B * 4 || angularVelLocal.size() != B * 3 || linearAccelLocal.size() != B * 3 || jointPos.size() != B * HB_NJ ||
        jointVel.size() != B * HB_NJ || contactFlag.size() != B * HB_NC)
      throw std::invalid_argument("[hunter_hip] KalmanFilterEstimate::update: wrong vector size");
    rbdState_.resize(B * HB_NRBD);
    observationState_.resize(B * HB_NX);
    ctx_.check(hb_estimator_update(ctx_.get(), period, quat.data(), angularVelLocal.data(), linearAccelLocal.data(), jointPos.data(),
                                   jointVel.data(), contactFlag.data(), toResident ? 1 : 0, rbdState_.data(), observationState_.data()),
               "hb_estimator_update");
    return rbdState_;
  }
  const vector_t& observationState() const { return observationState_; }  // currentObservation_.state per instance
  // StateEstimateBase::setCmdTorque + estContactForce (StateEstimateBase.cpp:130-206; LeggedController.cpp:344-345): the momentum
  // observer on the rbd state the last update() left on the device.  cmdTorque [batch][10] = the measured joint efforts.
  void estContactForce(scalar_t period, const vector_t& cmdTorque) {
    const size_t B = size_t(ctx_.batch());
    if (cmdTorque.size() != B * HB_NJ) throw std::invalid_argument("[hunter_hip] KalmanFilterEstimate::estContactForce: wrong vector size");
    estDisturbanceTorque_.resize(B * HB_NV);
    estContactForce_.resize(B * 16);
    ctx_.check(hb_estimator_contact_force(ctx_.get(), period, nullptr, cmdTorque.data(), estDisturbanceTorque_.data(), estContactForce_.data()),
               "hb_estimator_contact_force");
  }
  const vector_t& getEstContactForce() const { return estContactForce_; }            // StateEstimateBase.h:87-90
  const vector_t& getEstDisturbanceTorque() const { return estDisturbanceTorque_; }  // StateEstimateBase.h:91-94

 private:
  Context ctx_;
  vector_t rbdState_, observationState_, estContactForce_, estDisturbanceTorque_;
};

// ---- LCM bridge: legged::LeggedMujocoSim::read / write (legged_examples/legged_mujoco/src/LeggedMujocoSim.cpp:28-62) ----
// read():  the LOWSTATE payloads of the batch (336 bytes each, as received from LCM) -> estimator -> rbdState;
// write(): joint command law on the WBC result still on the device -> LOWCMD payloads (496 bytes each) to publish.
// Packing / unpacking runs on the device (include/hunter_lcm.h); liblcm itself is only needed for the transport.
class LcmBridge {
 public:
  LcmBridge(Context ctx, const hb_estimator_config& settings, const hb_joint_gains& gains, const vector_t* xHat0 = nullptr)
      : ctx_(std::move(ctx)), gains_(gains) {
    ctx_.check(hb_estimator_reset(ctx_.get(), &settings, xHat0 ? xHat0->data() : nullptr), "hb_estimator_reset");
  }
  const vector_t& read(scalar_t period, const std::vector<uint8_t>& lowState, const std::vector<int32_t>& contactFlag,
                       bool toResident = false) {
    const size_t B = size_t(ctx_.batch());
    if (lowState.size() != B * HB_LCM_LOW_STATE_BYTES || contactFlag.size() != B * HB_NC)
      throw std::invalid_argument("[hunter_hip] LcmBridge::read: wrong buffer size");
    rbdState_.resize(B * HB_NRBD);
    observationState_.resize(B * HB_NX);
    stamps_.resize(B);
    ctx_.check(hb_estimator_update_lcm(ctx_.get(), period, lowState.data(), contactFlag.data(), toResident ? 1 : 0, rbdState_.data(),
                                       observationState_.data(), stamps_.data()),
               "hb_estimator_update_lcm");
    return rbdState_;
  }
  const std::vector<uint8_t>& write(scalar_t period, int64_t timestampNs) {
    lowCmd_.resize(size_t(ctx_.batch()) * HB_LCM_LOW_CMD_BYTES);
    ctx_.check(hb_joint_command_lcm(ctx_.get(), &gains_, period, timestampNs, lowCmd_.data()), "hb_joint_command_lcm");
    return lowCmd_;
  }
  const vector_t& observationState() const { return observationState_; }
  const std::vector<int64_t>& timestamps() const { return stamps_; }

 private:
  Context ctx_;
  hb_joint_gains gains_;
  vector_t rbdState_, observationState_;
  std::vector<int64_t> stamps_;
  std::vector<uint8_t> lowCmd_;
};

// ---- the hot part of LeggedController::update (LeggedController.cpp:151-185) --------------------------------------
//   updatePolicy(); evaluatePolicy(t, x, optimizedState, optimizedInput, plannedMode); wbc_->update(...)
// evaluatePolicy runs on the device and feeds the WBC without a host round trip; its outputs are returned because the
// joint PD law that follows (:187-257) consumes optimizedState / optimizedInput.
struct ControlOutput {
  vector_t x;               // [batch][38] WBC solution
  vector_t optimizedState;  // [batch][22]
  vector_t optimizedInput;  // [batch][22]
  std::vector<int32_t> plannedMode, status;
};

inline void controllerUpdate(MpcMrtInterface& mpcMrt, const vector_t& time, const vector_t& rbdStateMeasured,
                             const std::vector<int32_t>* walkFlag, scalar_t period, ControlOutput& out) {
  const Context& c = mpcMrt.context();
  const size_t B = size_t(c.batch());
  if (time.size() != B || rbdStateMeasured.size() != B * HB_NRBD || (walkFlag && walkFlag->size() != B))
    throw std::invalid_argument("[hunter_hip] controllerUpdate: wrong vector size");
  out.x.resize(B * HB_NWBC);
  out.optimizedState.resize(B * HB_NX);
  out.optimizedInput.resize(B * HB_NU);
  out.plannedMode.resize(B);
  out.status.resize(B);
  mpcMrt.updatePolicy();
  c.check(hb_wbc_update(c.get(), time.data(), rbdStateMeasured.data(), walkFlag ? walkFlag->data() : nullptr, period, out.x.data(),
                        out.optimizedState.data(), out.optimizedInput.data(), out.plannedMode.data(), out.status.data()),
          "hb_wbc_update");
}

// ---- one batch over several GPUs (BASELINE north_star: "independent robot instances shard across the 8 GPUs of one
// node") -----------------------------------------------------------------------------------------------------------------------------
// Instances are independent: device g owns the contiguous range shardRange(batch, G, g) — the split of the Python harness
// (hunter_bipedal_control_amd/sharding.py) — with its own Context, and every call runs the shards side by side, one host thread per
// device for the duration of the call (a context's MPC-side calls come from one thread at a time, as the ABI asks).  Nothing is exchanged
// between devices; results are gathered into the caller's arrays in instance order, so a sharded run returns bit for bit what one
// context of the whole batch returns (tests/cpp/sharded_test.cpp).  `devices` may name a device more than once (two shards on one GPU).
class ShardedSolver {
 public:
  // rank r of `world` owns [begin, end): sizes differ by at most one, earlier ranks take the remainder
  static std::pair<int, int> shardRange(int total, int world, int rank) {
    const int base = total / world, rem = total % world;
    const int begin = rank * base + std::min(rank, rem);
    return {begin, begin + base + (rank < rem ? 1 : 0)};
  }
  ShardedSolver(const hb_model& model, const hb_config& config, int batch, int maxNodes, const std::vector<int>& devices)
      : batch_(batch), maxNodes_(maxNodes) {
    if (devices.empty() || batch < int(devices.size())) throw std::invalid_argument("[hunter_hip] ShardedSolver: at least one instance per device");
    for (size_t g = 0; g < devices.size(); ++g) {
      const std::pair<int, int> r = shardRange(batch, int(devices.size()), int(g));
      begin_.push_back(r.first);
      mpc_.emplace_back(Context(model, config, r.second - r.first, maxNodes, devices[g]));
      wbc_.emplace_back(mpc_.back().context());
    }
    begin_.push_back(batch);
  }
  int shards() const { return int(mpc_.size()); }
  int batch() const { return batch_; }
  int shardBegin(int g) const { return begin_[size_t(g)]; }
  MpcMrtInterface& shard(int g) { return mpc_[size_t(g)]; }

  void setReferences(const ReferenceTables& r) {
    const size_t N = size_t(maxNodes_);
    if (r.nNodes.size() != size_t(batch_)) throw std::invalid_argument("[hunter_hip] ShardedSolver::setReferences: tables of the whole batch expected");
    forEachShard([&](int g, size_t b, size_t n) {
      ReferenceTables s;
      s.nNodes.assign(r.nNodes.begin() + b, r.nNodes.begin() + b + n);
      s.t.assign(r.t.begin() + b * (N + 1), r.t.begin() + (b + n) * (N + 1));
      s.mode.assign(r.mode.begin() + b * N, r.mode.begin() + (b + n) * N);
      s.xRef.assign(r.xRef.begin() + b * N * HB_NX, r.xRef.begin() + (b + n) * N * HB_NX);
      s.swingRef.assign(r.swingRef.begin() + b * N * HB_NC * HB_SWING_REF, r.swingRef.begin() + (b + n) * N * HB_NC * HB_SWING_REF);
      mpc_[size_t(g)].setReferences(s);
    });
  }
  void resetMpcNode(const vector_t& initialStates) {
    requireSize(initialStates, size_t(batch_) * HB_NX, "resetMpcNode");
    forEachShard([&](int g, size_t b, size_t n) { mpc_[size_t(g)].resetMpcNode(slice(initialStates, b, n, HB_NX)); });
  }
  // states [batch][22]: the observation every shard's next advanceMpc starts from
  void setCurrentObservation(const vector_t& states) {
    requireSize(states, size_t(batch_) * HB_NX, "setCurrentObservation");
    forEachShard([&](int g, size_t b, size_t n) {
      std::vector<SystemObservation> obs(n);
      for (size_t i = 0; i < n; ++i) obs[i].state.assign(states.begin() + (b + i) * HB_NX, states.begin() + (b + i + 1) * HB_NX);
      mpc_[size_t(g)].setCurrentObservation(obs);
    });
  }
  // MPC_MRT_Interface::advanceMpc on every device at once; returns the per-instance status words in instance order
  const std::vector<int32_t>& advanceMpc() {
    mpcStatus_.assign(size_t(batch_), 0);
    forEachShard([&](int g, size_t b, size_t) {
      mpc_[size_t(g)].advanceMpc();
      const std::vector<int32_t>& st = mpc_[size_t(g)].mpcStatus();
      std::copy(st.begin(), st.end(), mpcStatus_.begin() + b);
    });
    return mpcStatus_;
  }
  void getSolution(vector_t& stateTrajectory, vector_t& inputTrajectory) {
    const size_t N = size_t(maxNodes_);
    stateTrajectory.assign(size_t(batch_) * (N + 1) * HB_NX, 0.0);
    inputTrajectory.assign(size_t(batch_) * N * HB_NU, 0.0);
    forEachShard([&](int g, size_t b, size_t) {
      vector_t xs, us;
      mpc_[size_t(g)].getSolution(xs, us);
      std::copy(xs.begin(), xs.end(), stateTrajectory.begin() + b * (N + 1) * HB_NX);
      std::copy(us.begin(), us.end(), inputTrajectory.begin() + b * N * HB_NU);
    });
  }
  vector_t getPerformanceIndices() {
    vector_t perf(size_t(batch_) * 4, 0.0);
    forEachShard([&](int g, size_t b, size_t) {
      const vector_t p = mpc_[size_t(g)].getPerformanceIndices();
      std::copy(p.begin(), p.end(), perf.begin() + b * 4);
    });
    return perf;
  }
  // the hot part of LeggedController::update for the whole batch (controllerUpdate per shard); outputs in instance order
  void controllerUpdate(const vector_t& time, const vector_t& rbdStateMeasured, const std::vector<int32_t>* walkFlag, scalar_t period,
                        ControlOutput& out) {
    const size_t B = size_t(batch_);
    if (time.size() != B || rbdStateMeasured.size() != B * HB_NRBD || (walkFlag && walkFlag->size() != B))
      throw std::invalid_argument("[hunter_hip] ShardedSolver::controllerUpdate: wrong vector size");
    out.x.assign(B * HB_NWBC, 0.0);
    out.optimizedState.assign(B * HB_NX, 0.0);
    out.optimizedInput.assign(B * HB_NU, 0.0);
    out.plannedMode.assign(B, 0);
    out.status.assign(B, 0);
    forEachShard([&](int g, size_t b, size_t n) {
      ControlOutput o;
      std::vector<int32_t> wf;
      if (walkFlag) wf.assign(walkFlag->begin() + b, walkFlag->begin() + b + n);
      hunter_hip::controllerUpdate(mpc_[size_t(g)], slice(time, b, n, 1), slice(rbdStateMeasured, b, n, HB_NRBD), walkFlag ? &wf : nullptr, period, o);
      std::copy(o.x.begin(), o.x.end(), out.x.begin() + b * HB_NWBC);
      std::copy(o.optimizedState.begin(), o.optimizedState.end(), out.optimizedState.begin() + b * HB_NX);
      std::copy(o.optimizedInput.begin(), o.optimizedInput.end(), out.optimizedInput.begin() + b * HB_NU);
      std::copy(o.plannedMode.begin(), o.plannedMode.end(), out.plannedMode.begin() + b);
      std::copy(o.status.begin(), o.status.end(), out.status.begin() + b);
    });
  }

 private:
  static vector_t slice(const vector_t& v, size_t b, size_t n, size_t w) { return vector_t(v.begin() + b * w, v.begin() + (b + n) * w); }
  static void requireSize(const vector_t& v, size_t n, const char* who) {
    if (v.size() != n) throw std::invalid_argument(std::string("[hunter_hip] ShardedSolver::") + who + ": wrong vector size");
  }
  // f(shard, first instance, instance count) on one host thread per shard; the first exception any of them threw is rethrown here
  template <class F>
  void forEachShard(F f) {
    std::vector<std::exception_ptr> err(mpc_.size());
    std::vector<std::thread> th;
    for (size_t g = 0; g < mpc_.size(); ++g)
      th.emplace_back([&, g] {
        try {
          f(int(g), size_t(begin_[g]), size_t(begin_[g + 1] - begin_[g]));
        } catch (...) {
          err[g] = std::current_exception();
        }
      });
    for (std::thread& t : th) t.join();
    for (const std::exception_ptr& e : err)
      if (e) std::rethrow_exception(e);
  }
  int batch_, maxNodes_;
  std::vector<int> begin_;
  std::vector<MpcMrtInterface> mpc_;
  std::vector<Wbc> wbc_;
  std::vector<int32_t> mpcStatus_;
};

}  // namespace hunter_hip
