// TEST INFRASTRUCTURE (oracle/_ref build only; see oracle/Makefile target ref).  C entry points over the reference's OWN
// whole-body-controller sources, compiled in place from /root/reference:
//   legged_wbc/src/{WbcBase, WeightedWbc, HierarchicalWbc, HoQp}.cpp + include/legged_wbc/Task.h
// What the reference does not vendor is stood in for under oracle/ref_shim_dense/: a dense Eigen subset, the pinocchio / OCS2
// entry points (which hand out rigid-body quantities FED IN by the caller, ref_feed.h — the generator computes them with the
// CPU oracle), boost::property_tree over the product's INFO reader, and qpOASES::QProblem delegating to the oracle's QP.
// Every line of task-row arithmetic, gain handling, stacking, weighting, cascade formulation and null-space projection that
// runs is the reference's.  tests/golden/make_ref_wbc.py writes tests/golden/ref_wbc.npz from this library.
#include <cstring>
#include <memory>
#include <string>

#include <qpOASES.hpp>

#include <legged_wbc/HierarchicalWbc.h>
#include <legged_wbc/HoQp.h>
#include <legged_wbc/WeightedWbc.h>

namespace {
using namespace legged;

// the protected task builders, re-exported
struct WeightedProbe : WeightedWbc {
  using WeightedWbc::WeightedWbc;
  using WbcBase::formulateBaseAccelTask;
  using WbcBase::formulateContactForceTask;
  using WbcBase::formulateFloatingBaseEomTask;
  using WbcBase::formulateFrictionConeTask;
  using WbcBase::formulateNoContactMotionTask;
  using WbcBase::formulateSwingLegTask;
  using WbcBase::formulateTorqueLimitsTask;
  using WeightedWbc::formulateConstraints;
  using WeightedWbc::formulateStanceBaseAccelTask;
  using WeightedWbc::formulateWeightedTasks;
  void baseUpdate(const vector_t& x, const vector_t& u, const vector_t& rbd, size_t mode) { WbcBase::update(x, u, rbd, mode, 0.002); }
};

struct Handle {
  ocs2::PinocchioInterface iface_w, iface_h;  // one source interface per controller: copies are numbered per source (PinocchioInterface.h)
  ocs2::CentroidalModelInfo info;
  ocs2::PinocchioEndEffectorKinematics ee;
  std::unique_ptr<WeightedProbe> weighted;
  std::unique_ptr<HierarchicalWbc> hier;
};

void set_feed(const double* const* meas /*8: M nle J dJ Jb dJb ee_pos ee_vel*/, const double* des_ee_pos, const double* des_ee_vel,
              const double* base_pose, const double* base_vel, const double* base_acc) {
  ref_feed::Feed& f = ref_feed::feed();
  ref_feed::Rbd& m = f.role[0];
  m.M = meas[0]; m.nle = meas[1]; m.J = meas[2]; m.dJ = meas[3]; m.Jb = meas[4]; m.dJb = meas[5]; m.ee_pos = meas[6]; m.ee_vel = meas[7];
  f.role[1] = ref_feed::Rbd();
  f.role[1].ee_pos = des_ee_pos;
  f.role[1].ee_vel = des_ee_vel;
  f.base_pose_des = base_pose; f.base_vel_des = base_vel; f.base_acc_des = base_acc;
}
vector_t vec(const double* p, int n) {
  vector_t v(n);
  for (int i = 0; i < n; ++i) v(i) = p[i];
  return v;
}
void put_task(const Task& t, double* A, double* b, int* mA, double* D, double* f, int* mD) {
  const int n = 38;
  *mA = t.a_.rows();
  *mD = t.d_.rows();
  for (int i = 0; i < t.a_.rows(); ++i) {
    for (int j = 0; j < n; ++j) A[i * n + j] = t.a_(i, j);
    b[i] = t.b_(i);
  }
  for (int i = 0; i < t.d_.rows(); ++i) {
    for (int j = 0; j < n; ++j) D[i * n + j] = t.d_(i, j);
    f[i] = t.f_(i);
  }
}
}  // namespace

extern "C" {

void* refwbc_create(const char* task_file) {
  auto* h = new Handle();
  h->weighted.reset(new WeightedProbe(h->iface_w, h->info, h->ee));
  h->weighted->loadTasksSetting(task_file, false);
  h->hier.reset(new HierarchicalWbc(h->iface_h, h->info, h->ee));
  h->hier->loadTasksSetting(task_file, false);
  return h;
}
void refwbc_destroy(void* h) { delete static_cast<Handle*>(h); }

// which: 0 EoM, 1 torque limits, 2 friction cone, 3 no-contact motion, 4 base accel, 5 swing leg, 6 contact force,
//        7 stance base accel, 8 WeightedWbc constraints, 9 WeightedWbc weighted tasks (stance flag applies)
int refwbc_task(void* hv, const double* const* meas, const double* des_ee_pos, const double* des_ee_vel, const double* base_pose,
                const double* base_vel, const double* base_acc, const double* x_des, const double* u_des, const double* rbd, int mode,
                int stance, int which, double* A, double* b, int* mA, double* D, double* f, int* mD) {
  Handle& h = *static_cast<Handle*>(hv);
  set_feed(meas, des_ee_pos, des_ee_vel, base_pose, base_vel, base_acc);
  const vector_t x = vec(x_des, 22), u = vec(u_des, 22), r = vec(rbd, 32);
  WeightedProbe& w = *h.weighted;
  w.setStanceMode(stance != 0);
  w.baseUpdate(x, u, r, size_t(mode));
  Task t;
  switch (which) {
    case 0: t = w.formulateFloatingBaseEomTask(); break;
    case 1: t = w.formulateTorqueLimitsTask(); break;
    case 2: t = w.formulateFrictionConeTask(); break;
    case 3: t = w.formulateNoContactMotionTask(); break;
    case 4: t = w.formulateBaseAccelTask(x, u, 0.002); break;
    case 5: t = w.formulateSwingLegTask(); break;
    case 6: t = w.formulateContactForceTask(u); break;
    case 7: t = w.formulateStanceBaseAccelTask(x, u, 0.002); break;
    case 8: t = w.formulateConstraints(); break;
    case 9: t = w.formulateWeightedTasks(x, u, 0.002); break;
    default: return -1;
  }
  put_task(t, A, b, mA, D, f, mD);
  return 0;
}

// kind 0: WeightedWbc::update, 1: HierarchicalWbc::update.  sol[38].
int refwbc_update(void* hv, int kind, const double* const* meas, const double* des_ee_pos, const double* des_ee_vel,
                  const double* base_pose, const double* base_vel, const double* base_acc, const double* x_des, const double* u_des,
                  const double* rbd, int mode, int stance, double* sol) {
  Handle& h = *static_cast<Handle*>(hv);
  set_feed(meas, des_ee_pos, des_ee_vel, base_pose, base_vel, base_acc);
  const vector_t x = vec(x_des, 22), u = vec(u_des, 22), r = vec(rbd, 32);
  vector_t s;
  if (kind == 0) {
    h.weighted->setStanceMode(stance != 0);
    s = h.weighted->update(x, u, r, size_t(mode), 0.002);
  } else {
    s = h.hier->update(x, u, r, size_t(mode), 0.002);
  }
  for (int i = 0; i < 38; ++i) sol[i] = s(i);
  return 0;
}

// The reference's HoQp cascade on plain dense tasks (L levels, highest priority first, n decision variables; rows
// concatenated level by level): solution x[n], stacked slack of all levels, and the stacked null-space basis of the LAST level
// (Z, n x nz row-major) for invariant checks.
int ref_hoqp(int n, int L, const int* mA, const double* A, const double* b, const int* mD, const double* D, const double* f, double* x,
             double* slack, int* n_slack, double* Z, int* nz) {
  std::shared_ptr<HoQp> prev;
  size_t oa = 0, od = 0;
  for (int l = 0; l < L; ++l) {
    matrix_t a(mA[l], n), d(mD[l], n);
    vector_t bb(mA[l]), ff(mD[l]);
    for (int i = 0; i < mA[l]; ++i) {
      for (int j = 0; j < n; ++j) a(i, j) = A[(oa + size_t(i)) * size_t(n) + size_t(j)];
      bb(i) = b[oa + size_t(i)];
    }
    for (int i = 0; i < mD[l]; ++i) {
      for (int j = 0; j < n; ++j) d(i, j) = D[(od + size_t(i)) * size_t(n) + size_t(j)];
      ff(i) = f[od + size_t(i)];
    }
    oa += size_t(mA[l]);
    od += size_t(mD[l]);
    Task t(mA[l] ? a : matrix_t(), mA[l] ? bb : vector_t(), mD[l] ? d : matrix_t(), mD[l] ? ff : vector_t());
    prev = prev ? std::make_shared<HoQp>(t, prev) : std::make_shared<HoQp>(t);
  }
  const vector_t sol = prev->getSolutions();
  for (int i = 0; i < n; ++i) x[i] = sol(i);
  const vector_t sl = prev->getStackedSlackSolutions();
  *n_slack = sl.rows();
  for (int i = 0; i < sl.rows(); ++i) slack[i] = sl(i);
  const matrix_t Zm = prev->getStackedZMatrix();
  *nz = Zm.cols();
  for (int i = 0; i < Zm.rows(); ++i)
    for (int j = 0; j < Zm.cols(); ++j) Z[i * Zm.cols() + j] = Zm(i, j);
  return 0;
}

void ref_set_qp_eps(double eps) { qpOASES::shim_eps() = eps; }
void ref_set_qp_reg_steps(int n) { qpOASES::shim_reg_steps() = n; }
// number of QProblem::init calls that failed since the last call of this function (HoQp.cpp ignores the return value, :180-182)
int ref_qp_failures() { const int n = qpOASES::shim_failures(); qpOASES::shim_failures() = 0; return n; }

}  // extern "C"
