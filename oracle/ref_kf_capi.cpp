// TEST INFRASTRUCTURE (oracle/_ref build only; oracle/Makefile target ref).  C entry points over the reference's OWN state
// estimator, compiled in place: legged_estimation/src/{LinearKalmanFilter, StateEstimateBase}.cpp.  Stand-ins
// (oracle/ref_shim_dense/): the dense Eigen subset, roscpp / tf2 / realtime_tools / nav_msgs plumbing (publishers that send
// nothing), pinocchio forward kinematics as no-ops and PinocchioEndEffectorKinematics handing out the foot positions /
// velocities FED IN by the caller (the generator computes them with the CPU oracle for the q, v the filter builds: base at the
// origin, measured orientation, joint angles / rates — LinearKalmanFilter.cpp:88-104).  The filter arithmetic that runs —
// updateImu / updateJointStates / updateContact, the 18-state / 28-measurement predict + correct, noise scheduling by contact,
// the covariance reset rule, rbdState packing — is the reference's.  tests/golden/make_ref_kf.py writes tests/golden/ref_kf.npz.
#include <cstring>
#include <deque>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include <Eigen/Dense>
#include <boost/property_tree/ptree.hpp>
#include <ros/ros.h>

#define private public   // xHat_, p_ are private members of KalmanFilterEstimate; the golden vectors read them
#define protected public
#include <legged_estimation/LinearKalmanFilter.h>
#undef private
#undef protected

using namespace legged;

namespace {
struct Handle {
  ocs2::PinocchioInterface iface;
  ocs2::CentroidalModelInfo info;
  ocs2::PinocchioEndEffectorKinematics ee;
  std::unique_ptr<KalmanFilterEstimate> kf;
  double t = 0.0;
};
}  // namespace

extern "C" {

void* refkf_create(const char* task_file) {
  auto* h = new Handle();
  h->kf.reset(new KalmanFilterEstimate(h->iface, h->info, h->ee));
  if (task_file && task_file[0]) h->kf->loadSettings(task_file, false);
  return h;
}
void refkf_destroy(void* h) { delete static_cast<Handle*>(h); }

// sensor side of LeggedController::updateStateEstimation (LeggedController.cpp:280-330): joint states, contact flags, IMU.
// Returns in rbd_after_imu the rbdState after these calls (orientation / angular velocity / joints filled), which is what
// KalmanFilterEstimate::update builds its pinocchio q, v from.
void refkf_set_sensors(void* hv, const double* quat_wxyz, const double* ang_vel_local, const double* lin_acc_local, const double* joint_pos,
                       const double* joint_vel, const int* contact, double* rbd_after_imu) {
  Handle& h = *static_cast<Handle*>(hv);
  vector_t jp(10), jv(10);
  for (int i = 0; i < 10; ++i) { jp(i) = joint_pos[i]; jv(i) = joint_vel[i]; }
  contact_flag_t cf;
  for (int i = 0; i < 4; ++i) cf[size_t(i)] = contact[i] != 0;
  h.kf->updateJointStates(jp, jv);
  h.kf->updateContact(cf);
  const Eigen::Quaternion<scalar_t> q(quat_wxyz[0], quat_wxyz[1], quat_wxyz[2], quat_wxyz[3]);
  const vector3_t w(ang_vel_local[0], ang_vel_local[1], ang_vel_local[2]), a(lin_acc_local[0], lin_acc_local[1], lin_acc_local[2]);
  const matrix3_t z = matrix3_t::Zero();
  h.kf->updateImu(q, w, a, z, z, z);
  for (int i = 0; i < 32; ++i) rbd_after_imu[i] = h.kf->rbdState_(i);
}

// KalmanFilterEstimate::update(time, period) with the foot kinematics of the current q, v fed in (ee_pos / ee_vel 4 x 3).
void refkf_update(void* hv, double dt, const double* ee_pos, const double* ee_vel, double* rbd, double* xhat, double* P) {
  Handle& h = *static_cast<Handle*>(hv);
  ref_feed::Feed& f = ref_feed::feed();
  f.role[0].ee_pos = ee_pos;
  f.role[0].ee_vel = ee_vel;
  h.t += dt;
  const vector_t r = h.kf->update(ros::Time(h.t), ros::Duration(dt));
  for (int i = 0; i < 32; ++i) rbd[i] = r(i);
  for (int i = 0; i < 18; ++i) xhat[i] = h.kf->xHat_(i);
  for (int i = 0; i < 18; ++i)
    for (int j = 0; j < 18; ++j) P[i * 18 + j] = h.kf->p_(i, j);
}

void refkf_settings(void* hv, double* out7) {
  Handle& h = *static_cast<Handle*>(hv);
  const double v[7] = {h.kf->footRadius_, h.kf->imuProcessNoisePosition_, h.kf->imuProcessNoiseVelocity_, h.kf->footProcessNoisePosition_,
                       h.kf->footSensorNoisePosition_, h.kf->footSensorNoiseVelocity_, h.kf->footHeightSensorNoise_};
  std::memcpy(out7, v, sizeof v);
}

// StateEstimateBase::loadSettings (StateEstimateBase.cpp:365-377): the contactForceEsimation block of task.info
void refkf_load_contact_force_settings(void* hv, const char* task_file, double* out2) {
  Handle& h = *static_cast<Handle*>(hv);
  h.kf->StateEstimateBase::loadSettings(task_file, false);
  out2[0] = h.kf->cutoffFrequency_;
  out2[1] = h.kf->contactThreshold_;
}
// setCmdTorque + estContactForce (LeggedController.cpp:344-345; StateEstimateBase.cpp:130-206) on the rbd state the object holds (the one
// the last refkf_update left).  The pinocchio results it reads are FED: M (16 x 16), g, C'v with its v, the angular rows of the 6-D
// Jacobians of contact frames 0 / 1, the linear rows of the contact Jacobians (12 x 16) — all from the oracle's contact_force_rbd at the
// q, v estContactForce builds.  Out: estDisturbancetorque_ (16), estContactforce_ (16), pSCgZinvlast_ (16).
void refkf_contact_force(void* hv, double dt, const double* tau, const double* M, const double* g, const double* CTv, const double* v,
                         const double* Jlin, const double* Jang, double* dist, double* cf, double* z) {
  Handle& h = *static_cast<Handle*>(hv);
  ref_feed::Rbd& f = ref_feed::feed().role[0];
  f.M = M; f.g = g; f.CTv = CTv; f.v = v; f.J = Jlin; f.Jang = Jang;
  vector_t t(10);
  for (int i = 0; i < 10; ++i) t(i) = tau[i];
  h.kf->setCmdTorque(t);
  h.kf->estContactForce(ros::Duration(dt));
  for (int i = 0; i < 16; ++i) { dist[i] = h.kf->estDisturbancetorque_(i); cf[i] = h.kf->estContactforce_(i); z[i] = h.kf->pSCgZinvlast_(i); }
  f.g = nullptr; f.CTv = nullptr; f.v = nullptr; f.Jang = nullptr;
}

// quatToZyx of StateEstimateBase.h:153-166 (a template in the reference's header)
void refkf_quat_to_zyx(const double* quat_wxyz, double* zyx) {
  const Eigen::Quaternion<scalar_t> q(quat_wxyz[0], quat_wxyz[1], quat_wxyz[2], quat_wxyz[3]);
  const vector3_t z = quatToZyx(q);
  for (int i = 0; i < 3; ++i) zyx[i] = z(i);
}

}  // extern "C"
