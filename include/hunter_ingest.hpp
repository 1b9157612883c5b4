// C++ ingest of the reference's input files — what LeggedController::init reads through LeggedInterface (legged_interface/src/
// LeggedInterface.cpp:55-96: task.info, hunter.urdf, reference.info), WbcBase::loadTasksSetting (legged_wbc/src/WbcBase.cpp:352-411),
// KalmanFilterEstimate::loadSettings (legged_estimation/src/LinearKalmanFilter.cpp:317-335) and SwingTrajectoryPlanner's
// loadSwingTrajectorySettings (SwingTrajectoryPlanner.cpp:537-568) — flattened into the plain structs of include/hunter_hip.h.
// Header-only C++14, no third-party dependency (the reference uses boost::property_tree, urdfdom and pinocchio for this).
// The Python side of the package does the same in hunter_bipedal_control_amd/ingest.py + abi.py; tests/test_cpp_ingest.py holds the
// two to each other byte for byte, and to the packaged data/hunter_params.bin.
//
// URDF subset: revolute / fixed joints and <inertial>; every rpy of hunter.urdf is zero (checked).  Fixed children (imu_link,
// leg_*_f{1,2}_link) are merged into their movable ancestor the way pinocchio's URDF parser does.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "hunter_hip.h"
#include "hunter_info.hpp"

namespace hunter_hip {

struct ModeTemplateData {             // a gait of gait.info / the defaultModeSequenceTemplate of reference.info
  std::vector<double> switchingTimes;
  std::vector<int32_t> modes;
};
struct Parameters {
  hb_model model{};
  hb_config config{};
  hb_estimator_config estimator{};
  hb_refgen_config refgen{};
  hb_joint_gains gains{};
  double timeHorizon = 0.0, mpcFrequency = 0.0, phaseTransitionStanceTime = 0.0;
  std::vector<double> initialEventTimes;          // reference.info initialModeSchedule
  std::vector<int32_t> initialModes;
  ModeTemplateData defaultTemplate;               // reference.info defaultModeSequenceTemplate
  std::map<std::string, ModeTemplateData> gaits;  // gait.info (optional)
};

namespace ingest_detail {
inline int modeNumber(const std::string& s) {     // MotionPhaseDefinition.h:66-95
  if (s == "FLY") return 0;
  if (s == "R") return 1;
  if (s == "L") return 2;
  if (s == "STANCE") return 3;
  throw std::invalid_argument("unknown mode name '" + s + "'");
}
struct V3 { double x = 0, y = 0, z = 0; };
inline V3 vec3(const std::string& s) {
  V3 v;
  if (std::sscanf(s.c_str(), "%lf %lf %lf", &v.x, &v.y, &v.z) != 3) throw std::invalid_argument("URDF: bad vector '" + s + "'");
  return v;
}
// attribute `key` of the tag text `tag` ("" if absent)
inline std::string attr(const std::string& tag, const std::string& key) {
  size_t at = 0;
  while ((at = tag.find(key + "=", at)) != std::string::npos) {
    if (at > 0 && (std::isalnum(static_cast<unsigned char>(tag[at - 1])) || tag[at - 1] == '_')) { at += key.size(); continue; }
    const size_t q0 = tag.find_first_of("\"'", at);
    if (q0 == std::string::npos) return "";
    const size_t q1 = tag.find(tag[q0], q0 + 1);
    return tag.substr(q0 + 1, q1 - q0 - 1);
  }
  return "";
}
struct Body { double m = 0; V3 c; double I[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}; };
struct Joint { std::string name, type, parent, child; V3 origin, axis; double lower = 0, upper = 0, effort = 0, velocity = 0; bool has_type = false; };
// combine two rigid bodies expressed in the same frame (inertia about own COM) — ingest.py::_merge, same operation order
inline Body merge(const Body& a, const V3& ca, const Body& b, const V3& cb) {
  Body o;
  o.m = a.m + b.m;
  o.c.x = (a.m * ca.x + b.m * cb.x) / o.m;
  o.c.y = (a.m * ca.y + b.m * cb.y) / o.m;
  o.c.z = (a.m * ca.z + b.m * cb.z) / o.m;
  auto shift = [](double mm, const V3& r, double S[3][3]) {
    const double rr[3] = {r.x, r.y, r.z};
    const double d = rr[0] * rr[0] + rr[1] * rr[1] + rr[2] * rr[2];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) S[i][j] = mm * (d * (i == j ? 1.0 : 0.0) - rr[i] * rr[j]);
  };
  double S1[3][3], S2[3][3];
  shift(a.m, V3{ca.x - o.c.x, ca.y - o.c.y, ca.z - o.c.z}, S1);
  shift(b.m, V3{cb.x - o.c.x, cb.y - o.c.y, cb.z - o.c.z}, S2);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) o.I[i][j] = ((a.I[i][j] + S1[i][j]) + b.I[i][j]) + S2[i][j];
  return o;
}
}  // namespace ingest_detail

// hunter.urdf -> the flat 11-body model (base + 10 links, fixed children merged)
inline hb_model readUrdf(const std::string& path) {
  using namespace ingest_detail;
  const std::string text = read_text_file(path);
  std::map<std::string, Body> links;
  std::vector<Joint> joints;
  // tag scanner: <link ...> ... </link>, <joint ...> ... </joint> at the top level; <transmission> blocks are skipped
  std::string curLink, section;
  Joint curJoint;
  bool inJoint = false, inInertial = false;
  int skipDepth = 0;
  size_t pos = 0;
  while ((pos = text.find('<', pos)) != std::string::npos) {
    const size_t end = text.find('>', pos);
    if (end == std::string::npos) break;
    std::string tag = text.substr(pos + 1, end - pos - 1);
    pos = end + 1;
    if (tag.empty() || tag[0] == '?' || tag[0] == '!') {
      if (tag.compare(0, 3, "!--") == 0 && tag.size() >= 2 && tag.compare(tag.size() - 2, 2, "--") != 0) {  // comment with '>' inside
        const size_t ce = text.find("-->", pos);
        pos = ce == std::string::npos ? text.size() : ce + 3;
      }
      continue;
    }
    const bool closing = tag[0] == '/', selfclosing = tag.back() == '/';
    const size_t n0 = closing ? 1 : 0;
    size_t n1 = n0;
    while (n1 < tag.size() && !std::isspace(static_cast<unsigned char>(tag[n1])) && tag[n1] != '/') ++n1;
    const std::string name = tag.substr(n0, n1 - n0);
    if (skipDepth > 0 || name == "transmission" || name == "gazebo") {  // blocks with their own <joint> children
      if (name == "transmission" || name == "gazebo") skipDepth += closing ? -1 : (selfclosing ? 0 : 1);
      continue;
    }
    if (closing) {
      if (name == "link") curLink.clear();
      else if (name == "joint" && inJoint) { if (curJoint.has_type) joints.push_back(curJoint); inJoint = false; }
      else if (name == "inertial") inInertial = false;
      continue;
    }
    if (name == "link" && !inJoint) {
      curLink = attr(tag, "name");
      links[curLink];
      if (selfclosing) curLink.clear();
    } else if (name == "joint" && curLink.empty()) {
      curJoint = Joint();
      curJoint.name = attr(tag, "name");
      curJoint.type = attr(tag, "type");
      curJoint.has_type = !curJoint.type.empty();
      inJoint = !selfclosing;
    } else if (!curLink.empty()) {
      if (name == "inertial") inInertial = !selfclosing;
      else if (inInertial && name == "origin") {
        if (!attr(tag, "xyz").empty()) links[curLink].c = vec3(attr(tag, "xyz"));
        if (!attr(tag, "rpy").empty()) { const V3 r = vec3(attr(tag, "rpy")); if (r.x != 0 || r.y != 0 || r.z != 0) throw std::invalid_argument("URDF: rotated inertial frames unsupported"); }
      } else if (inInertial && name == "mass") links[curLink].m = std::strtod(attr(tag, "value").c_str(), nullptr);
      else if (inInertial && name == "inertia") {
        auto g = [&tag](const char* k) { const std::string s = attr(tag, k); return s.empty() ? 0.0 : std::strtod(s.c_str(), nullptr); };
        Body& b = links[curLink];
        b.I[0][0] = g("ixx"); b.I[0][1] = b.I[1][0] = g("ixy"); b.I[0][2] = b.I[2][0] = g("ixz");
        b.I[1][1] = g("iyy"); b.I[1][2] = b.I[2][1] = g("iyz"); b.I[2][2] = g("izz");
      }
    } else if (inJoint) {
      if (name == "parent") curJoint.parent = attr(tag, "link");
      else if (name == "child") curJoint.child = attr(tag, "link");
      else if (name == "origin") {
        if (!attr(tag, "xyz").empty()) curJoint.origin = vec3(attr(tag, "xyz"));
        if (!attr(tag, "rpy").empty()) { const V3 r = vec3(attr(tag, "rpy")); if (r.x != 0 || r.y != 0 || r.z != 0) throw std::invalid_argument("URDF: rotated joint frames unsupported"); }
      } else if (name == "axis") curJoint.axis = vec3(attr(tag, "xyz"));
      else if (name == "limit") {
        auto g = [&tag](const char* k) { const std::string s = attr(tag, k); return s.empty() ? 0.0 : std::strtod(s.c_str(), nullptr); };
        curJoint.lower = g("lower"); curJoint.upper = g("upper"); curJoint.effort = g("effort"); curJoint.velocity = g("velocity");
      }
    }
  }
  static const char* JOINT_NAMES[HB_NJ] = {"leg_l1_joint", "leg_l2_joint", "leg_l3_joint", "leg_l4_joint", "leg_l5_joint",
                                           "leg_r1_joint", "leg_r2_joint", "leg_r3_joint", "leg_r4_joint", "leg_r5_joint"};
  static const char* CONTACT_NAMES[HB_NC] = {"leg_l_f1_link", "leg_r_f1_link", "leg_l_f2_link", "leg_r_f2_link"};  // ModelSettings.h:62
  auto findJoint = [&joints](const std::string& n) -> const Joint& {
    for (const Joint& j : joints)
      if (j.name == n) return j;
    throw std::invalid_argument("URDF: joint '" + n + "' not found");
  };
  std::vector<std::string> bodyLinks{"base_link"};
  for (const char* jn : JOINT_NAMES) bodyLinks.push_back(findJoint(jn).child);
  std::map<std::string, int> bodyIndex;
  for (size_t i = 0; i < bodyLinks.size(); ++i) bodyIndex[bodyLinks[i]] = int(i);
  std::vector<Body> bodies;
  for (const std::string& n : bodyLinks) {
    if (!links.count(n)) throw std::invalid_argument("URDF: link '" + n + "' not found");
    bodies.push_back(links[n]);
  }
  std::map<std::string, std::pair<int, V3>> frames;   // link -> (body, offset in the body frame)
  for (const auto& kv : bodyIndex) frames[kv.first] = {kv.second, V3{}};
  std::vector<const Joint*> pending;
  for (const Joint& j : joints)
    if (j.type == "fixed") pending.push_back(&j);
  while (!pending.empty()) {
    bool progressed = false;
    for (size_t k = 0; k < pending.size();) {
      const Joint& j = *pending[k];
      auto it = frames.find(j.parent);
      if (it == frames.end()) { ++k; continue; }
      const int b = it->second.first;
      const V3 off{it->second.second.x + j.origin.x, it->second.second.y + j.origin.y, it->second.second.z + j.origin.z};
      frames[j.child] = {b, off};
      const Body& ch = links[j.child];
      if (ch.m > 0) {
        const V3 cc{off.x + ch.c.x, off.y + ch.c.y, off.z + ch.c.z};
        bodies[size_t(b)] = merge(bodies[size_t(b)], bodies[size_t(b)].c, ch, cc);
      }
      pending.erase(pending.begin() + long(k));
      progressed = true;
    }
    if (!progressed) throw std::invalid_argument("URDF: dangling fixed joint");
  }
  hb_model m{};
  for (int j = 0; j < HB_NJ; ++j) {
    const Joint& jt = findJoint(JOINT_NAMES[j]);
    m.parent[j] = bodyIndex.at(jt.parent);
    m.joint_origin[j][0] = jt.origin.x; m.joint_origin[j][1] = jt.origin.y; m.joint_origin[j][2] = jt.origin.z;
    m.joint_axis[j][0] = jt.axis.x; m.joint_axis[j][1] = jt.axis.y; m.joint_axis[j][2] = jt.axis.z;
    m.q_lower[j] = jt.lower; m.q_upper[j] = jt.upper; m.qd_limit[j] = jt.velocity; m.effort[j] = jt.effort;
  }
  for (int b = 0; b < HB_NBODY; ++b) {
    const Body& bd = bodies[size_t(b)];
    m.mass[b] = bd.m;
    m.com[b][0] = bd.c.x; m.com[b][1] = bd.c.y; m.com[b][2] = bd.c.z;
    m.inertia[b][0] = bd.I[0][0]; m.inertia[b][1] = bd.I[0][1]; m.inertia[b][2] = bd.I[0][2];
    m.inertia[b][3] = bd.I[1][1]; m.inertia[b][4] = bd.I[1][2]; m.inertia[b][5] = bd.I[2][2];
  }
  for (int i = 0; i < HB_NC; ++i) {
    const auto it = frames.find(CONTACT_NAMES[i]);
    if (it == frames.end()) throw std::invalid_argument(std::string("URDF: contact frame '") + CONTACT_NAMES[i] + "' not found");
    m.contact_body[i] = it->second.first;
    m.contact_offset[i][0] = it->second.second.x; m.contact_offset[i][1] = it->second.second.y; m.contact_offset[i][2] = it->second.second.z;
  }
  m.gravity = 9.81;
  return m;
}

// task.info + reference.info (+ gait.info) -> hb_config and the settings of the estimator / reference generation / joint gains
inline void readConfig(const std::string& taskFile, const std::string& referenceFile, const std::string& gaitFile, Parameters& p) {
  const InfoNode task = read_info_file(taskFile), ref = read_info_file(referenceFile);
  hb_config& c = p.config;
  std::memset(&c, 0, sizeof(c));
  c.dt = task.number("sqp.dt");
  c.sqp_iterations = int32_t(task.number("sqp.sqpIteration"));
  c.wbc_type = 0;
  c.g_max = task.number("sqp.g_max");
  c.g_min = task.number("sqp.g_min");
  c.alpha_decay = 0.5; c.alpha_min = 1e-4; c.gamma_c = 1e-6; c.armijo_factor = 1e-4;   // OCS2 FilterLinesearch defaults
  const std::vector<double> Q = task.matrix("Q", 22, 22), R = task.matrix("R", 24, 24), x0 = task.matrix("initialState", 22, 1);
  for (int i = 0; i < HB_NX; ++i) { c.Q_diag[i] = Q[size_t(i) * 22 + size_t(i)]; c.initial_state[i] = x0[size_t(i)]; }
  for (int i = 0; i < 24; ++i) c.R_task_diag[i] = R[size_t(i) * 24 + size_t(i)];
  c.friction_mu = task.number("frictionConeSoftConstraint.frictionCoefficient");
  c.friction_reg = 25.0; c.friction_gripper = 0.0; c.friction_hess_shift = 1e-6;           // FrictionConeConstraint.h:77-83
  c.friction_barrier_mu = task.number("frictionConeSoftConstraint.mu");
  c.friction_barrier_delta = task.number("frictionConeSoftConstraint.delta");
  c.soft_swing_weight = task.number("softSwingTraj.weight");
  c.pos_limit_barrier[0] = 1.0; c.pos_limit_barrier[1] = 0.1;                               // LeggedInterface.cpp:337-339,352
  c.vel_limit_barrier[0] = 1.0; c.vel_limit_barrier[1] = 0.1;
  c.force_limit_barrier[0] = 0.1; c.force_limit_barrier[1] = 1.0;
  c.force_limit[0] = 0.0; c.force_limit[1] = 350.0;
  c.position_error_gain = task.number("model_settings.positionErrorGain");
  c.zero_vel_z_gain = 3.0; c.zero_vel_z_offset = -0.06;                                     // LeggedInterface.cpp:436-444
  c.xy_ref_gain = 3.0;                                                                      // LeggedRobotPreComputation.cpp:113-116
  const std::vector<double> tl = task.matrix("torqueLimitsTask", 5, 1);
  for (int i = 0; i < 5; ++i) c.torque_limits[i] = tl[size_t(i)];
  c.wbc_friction_mu = task.number("frictionConeTask.frictionCoefficient");
  c.swing_kp = task.number("swingLegTask.kp"); c.swing_kd = task.number("swingLegTask.kd");
  c.base_height_kp = task.number("baseHeightTask.kp"); c.base_height_kd = task.number("baseHeightTask.kd");
  c.base_angular_kp = task.number("baseAngularTask.kp"); c.base_angular_kd = task.number("baseAngularTask.kd");
  c.weight_swing_leg = task.number("weight.swingLeg"); c.weight_base_accel = task.number("weight.baseAccel");
  c.weight_contact_force = task.number("weight.contactForce");
  c.wbc_eps_reg = 1e-8;   // the regularised-minimiser rule (why not qpOASES's 5e3 * EPS, with measurements)
  c.wbc_max_iter = 120;
  const std::vector<double> dj = ref.matrix("defaultJointState", 10, 1);
  for (int j = 0; j < HB_NJ; ++j) c.default_joint_state[j] = dj[size_t(j)];
  c.delta_tol = task.number("sqp.deltaTol");
  c.wbc_reg_steps = 1;    // qpOASES setToMPC(): numRegularisationSteps = 1 (WeightedWbc.cpp:47-48, HoQp.cpp:175-176)
  p.timeHorizon = task.number("mpc.timeHorizon");
  p.mpcFrequency = task.number("mpc.mpcDesiredFrequency");
  p.phaseTransitionStanceTime = task.number("model_settings.phaseTransitionStanceTime");
  // state estimator (task.info kalmanFilter block; defaults LinearKalmanFilter.h:50-56)
  hb_estimator_config& e = p.estimator;
  e.foot_radius = task.number("kalmanFilter.footRadius", 0.02);
  e.imu_process_noise_position = task.number("kalmanFilter.imuProcessNoisePosition", 0.02);
  e.imu_process_noise_velocity = task.number("kalmanFilter.imuProcessNoiseVelocity", 0.02);
  e.foot_process_noise_position = task.number("kalmanFilter.footProcessNoisePosition", 0.002);
  e.foot_sensor_noise_position = task.number("kalmanFilter.footSensorNoisePosition", 0.005);
  e.foot_sensor_noise_velocity = task.number("kalmanFilter.footSensorNoiseVelocity", 0.1);
  e.foot_height_sensor_noise = task.number("kalmanFilter.footHeightSensorNoise", 0.01);
  // contactForceEsimation block (sic), StateEstimateBase::loadSettings (StateEstimateBase.cpp:365-377)
  e.contact_force_cutoff_frequency = task.number("contactForceEsimation.cutoffFrequency", 250.0);
  e.contact_threshold = task.number("contactForceEsimation.contactThreshold", 75.0);
  // reference generation (swing_trajectory_config; the loader key is next_position_z, SwingTrajectoryPlanner.cpp:560)
  hb_refgen_config& g = p.refgen;
  std::memset(&g, 0, sizeof(g));
  g.dt = c.dt;
  g.com_height = ref.number("comHeight");
  g.next_position_z = task.number("swing_trajectory_config.next_position_z", 0.02);
  g.swing_height = task.number("swing_trajectory_config.swingHeight");
  g.swing_time_scale = task.number("swing_trajectory_config.swingTimeScale");
  const double bx1 = task.number("swing_trajectory_config.feet_bias_x1"), bx2 = task.number("swing_trajectory_config.feet_bias_x2");
  const double by = task.number("swing_trajectory_config.feet_bias_y"), bz = task.number("swing_trajectory_config.feet_bias_z");
  const double bias[HB_NC][3] = {{bx1, by, bz}, {bx1, -by, bz}, {bx2, by, bz}, {bx2, -by, bz}};
  for (int i = 0; i < HB_NC; ++i)
    for (int a = 0; a < 3; ++a) g.feet_bias[i][a] = bias[i][a];
  for (int j = 0; j < HB_NJ; ++j) g.default_joints[j] = c.default_joint_state[j];
  g.joint_ik = 1;
  // joint gains: dynamic_reconfigure defaults of legged_controllers/cfg/Tutorials.cfg:6-16
  hb_joint_gains& k = p.gains;
  k.kp_big_stance = 40.0; k.kp_big_swing = 30.0; k.kd_big = 2.0; k.kp_small_stance = 30.0; k.kp_small_swing = 20.0; k.kd_small = 2.0;
  k.kd_feet = 0.01; k.kp_position = 10.0; k.kd_position = 3.0;
  // mode schedules (reference.info:21-46)
  p.initialEventTimes.clear(); p.initialModes.clear();
  for (const std::string& s : ref.list("initialModeSchedule.eventTimes")) p.initialEventTimes.push_back(std::strtod(s.c_str(), nullptr));
  for (const std::string& s : ref.list("initialModeSchedule.modeSequence")) p.initialModes.push_back(ingest_detail::modeNumber(s));
  p.defaultTemplate = ModeTemplateData();
  for (const std::string& s : ref.list("defaultModeSequenceTemplate.switchingTimes")) p.defaultTemplate.switchingTimes.push_back(std::strtod(s.c_str(), nullptr));
  for (const std::string& s : ref.list("defaultModeSequenceTemplate.modeSequence")) p.defaultTemplate.modes.push_back(ingest_detail::modeNumber(s));
  p.gaits.clear();
  if (!gaitFile.empty()) {
    const InfoNode gait = read_info_file(gaitFile);
    for (const std::string& name : gait.list("list")) {
      ModeTemplateData t;
      for (const std::string& s : gait.list(name + ".switchingTimes")) t.switchingTimes.push_back(std::strtod(s.c_str(), nullptr));
      for (const std::string& s : gait.list(name + ".modeSequence")) t.modes.push_back(ingest_detail::modeNumber(s));
      p.gaits[name] = t;
    }
  }
}

inline Parameters loadParameters(const std::string& taskFile, const std::string& urdfFile, const std::string& referenceFile,
                                 const std::string& gaitFile = "") {
  Parameters p;
  p.model = readUrdf(urdfFile);
  readConfig(taskFile, referenceFile, gaitFile, p);
  return p;
}

// ---- binary image (data/hunter_params.bin, version 2) ------------------------------------------------------------------------
//   u32 magic "HB02", sizeof(hb_model), sizeof(hb_config), sizeof(hb_estimator_config), sizeof(hb_refgen_config),
//   sizeof(hb_joint_gains), n_initial_events, n_template_times; then the five structs; then f64 timeHorizon, mpcFrequency,
//   phaseTransitionStanceTime; initial event times (f64) and modes (i32, n + 1); template switching times (f64) and modes (i32, n - 1).
constexpr uint32_t PARAMS_MAGIC_V2 = 0x48423032u;
inline void writeParametersBlob(const Parameters& p, const std::string& path) {
  std::FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) throw std::invalid_argument("[hunter_hip] cannot write " + path);
  const uint32_t head[8] = {PARAMS_MAGIC_V2, uint32_t(sizeof(hb_model)), uint32_t(sizeof(hb_config)), uint32_t(sizeof(hb_estimator_config)),
                            uint32_t(sizeof(hb_refgen_config)), uint32_t(sizeof(hb_joint_gains)), uint32_t(p.initialEventTimes.size()),
                            uint32_t(p.defaultTemplate.switchingTimes.size())};
  std::fwrite(head, sizeof(head), 1, f);
  std::fwrite(&p.model, sizeof(hb_model), 1, f);
  std::fwrite(&p.config, sizeof(hb_config), 1, f);
  std::fwrite(&p.estimator, sizeof(hb_estimator_config), 1, f);
  std::fwrite(&p.refgen, sizeof(hb_refgen_config), 1, f);
  std::fwrite(&p.gains, sizeof(hb_joint_gains), 1, f);
  const double tail[3] = {p.timeHorizon, p.mpcFrequency, p.phaseTransitionStanceTime};
  std::fwrite(tail, sizeof(tail), 1, f);
  std::fwrite(p.initialEventTimes.data(), 8, p.initialEventTimes.size(), f);
  std::fwrite(p.initialModes.data(), 4, p.initialModes.size(), f);
  std::fwrite(p.defaultTemplate.switchingTimes.data(), 8, p.defaultTemplate.switchingTimes.size(), f);
  std::fwrite(p.defaultTemplate.modes.data(), 4, p.defaultTemplate.modes.size(), f);
  std::fclose(f);
}
inline Parameters loadParametersBlob(const std::string& path) {
  std::FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) throw std::invalid_argument("[hunter_hip] parameter file not found: " + path);
  Parameters p;
  uint32_t head[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  bool ok = std::fread(head, sizeof(head), 1, f) == 1 && head[0] == PARAMS_MAGIC_V2 && head[1] == sizeof(hb_model) && head[2] == sizeof(hb_config) &&
            head[3] == sizeof(hb_estimator_config) && head[4] == sizeof(hb_refgen_config) && head[5] == sizeof(hb_joint_gains) && head[6] < 4096 &&
            head[7] < 4096;
  ok = ok && std::fread(&p.model, sizeof(hb_model), 1, f) == 1 && std::fread(&p.config, sizeof(hb_config), 1, f) == 1 &&
       std::fread(&p.estimator, sizeof(hb_estimator_config), 1, f) == 1 && std::fread(&p.refgen, sizeof(hb_refgen_config), 1, f) == 1 &&
       std::fread(&p.gains, sizeof(hb_joint_gains), 1, f) == 1;
  double tail[3] = {0, 0, 0};
  ok = ok && std::fread(tail, sizeof(tail), 1, f) == 1;
  if (ok) {
    p.timeHorizon = tail[0]; p.mpcFrequency = tail[1]; p.phaseTransitionStanceTime = tail[2];
    p.initialEventTimes.resize(head[6]);
    p.initialModes.resize(head[6] + 1);
    p.defaultTemplate.switchingTimes.resize(head[7]);
    p.defaultTemplate.modes.resize(head[7] > 0 ? head[7] - 1 : 0);
    ok = (head[6] == 0 || std::fread(p.initialEventTimes.data(), 8, head[6], f) == head[6]) &&
         std::fread(p.initialModes.data(), 4, p.initialModes.size(), f) == p.initialModes.size() &&
         (head[7] == 0 || std::fread(p.defaultTemplate.switchingTimes.data(), 8, head[7], f) == head[7]) &&
         (p.defaultTemplate.modes.empty() || std::fread(p.defaultTemplate.modes.data(), 4, p.defaultTemplate.modes.size(), f) == p.defaultTemplate.modes.size());
  }
  std::fclose(f);
  if (!ok) throw std::invalid_argument("[hunter_hip] parameter file does not match this ABI (HB02): " + path);
  return p;
}

}  // namespace hunter_hip
