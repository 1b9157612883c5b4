/*
 * hunter_hip.h — C ABI of the MI355X-native batched NMPC + WBC solver for the EC-hunter80 biped.
 *
 * This is the drop-in boundary for the hot path of bridgedp/hunter_bipedal_control.
 * Every entry point names the reference interface it replaces (paths relative to the reference root).
 * Plain pointers and sizes only; no C++/torch types.  All functions return 0 on success and a
 * negative hb_status on failure (never throw); hb_last_error() gives the message.
 *
 * Conventions
 *   MPC state  x[22] = [h_lin/m (3), h_ang/m (3), base pos (3), base ZYX euler (3), joints l1..l5 r1..r5 (10)]
 *   MPC input  u[22] = [F(L_f1) F(R_f1) F(L_f2) F(R_f2) (world frame, 12), joint velocities (10)]
 *   rbd state  [32]  = [zyx(3), pos(3), q_j(10), omega_world(3), v_lin(3), qd_j(10)]
 *                      (legged_estimation/src/StateEstimateBase.cpp:73-106)
 *   WBC output [38]  = [qdd(16) | F(12) | tau(10)]            (legged_wbc/src/WbcBase.cpp:40)
 *   modes: 0 FLY, 1 R (feet 1,3 closed), 2 L (feet 0,2 closed), 3 STANCE
 *                      (legged_interface/include/legged_interface/gait/MotionPhaseDefinition.h:55-95)
 *   All matrices passed through this ABI are row-major doubles.
 *
 * Batch layout: instance-major.  Host buffers are caller-owned; device buffers are library-owned.
 * One hb_ctx drives one GPU (one process per GPU; ranks shard the batch).
 */
#ifndef HUNTER_HIP_H
#define HUNTER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HB_NX 22
#define HB_NU 22
#define HB_NV 16   /* generalized coordinates: base pos(3) + zyx(3) + joints(10) */
#define HB_NJ 10
#define HB_NC 4    /* 3-DoF point contacts */
#define HB_NBODY 11 /* base + 10 links (fixed URDF children merged into their parent) */
#define HB_NWBC 38
#define HB_NRBD 32
#define HB_SWING_REF 6 /* per foot and node: pos xyz, vel xyz of the swing reference */

typedef enum hb_status {
  HB_OK = 0,
  HB_ERR_ARG = -1,       /* bad argument (null pointer, range) */
  HB_ERR_DEVICE = -2,    /* HIP runtime error */
  HB_ERR_STATE = -3,     /* call order (e.g. solve before references were set) */
  HB_ERR_NO_GPU = -4     /* no gfx950 device visible: the product path never falls back to a CPU */
} hb_status;

/* Per-instance solver status words written by the device (SURVEY.md §5 "failure detection"). */
#define HB_INST_OK 0
#define HB_INST_MAXITER 1   /* WBC QP hit its working-set-change limit: previous solution reused
                               (legged_wbc/src/WeightedWbc.cpp:57-65); MPC: the line search reached alpha_min without
                               an acceptable step (a search that stops on deltaTol — converged — is HB_INST_OK) */
#define HB_INST_INFEASIBLE 2
#define HB_INST_NAN 3       /* non-finite value / non-positive Riccati pivot */

/* Rigid-body model, replaces PinocchioInterface built from the URDF
 * (legged_interface/src/LeggedInterface.cpp:188-200).  Body 0 is base_link, bodies 1..5 the left leg
 * links, 6..10 the right leg links; joint j connects body parent[j] to body j+1. */
typedef struct hb_model {
  int32_t parent[HB_NJ];
  double joint_origin[HB_NJ][3];   /* joint frame origin in the parent body frame (all URDF rpy are 0) */
  double joint_axis[HB_NJ][3];
  double q_lower[HB_NJ], q_upper[HB_NJ], qd_limit[HB_NJ], effort[HB_NJ];
  double mass[HB_NBODY];
  double com[HB_NBODY][3];         /* body frame */
  double inertia[HB_NBODY][6];     /* about the body COM, body axes: xx xy xz yy yz zz */
  int32_t contact_body[HB_NC];     /* order L_f1, R_f1, L_f2, R_f2 (ModelSettings.h:62) */
  double contact_offset[HB_NC][3];
  double gravity;                  /* 9.81 (legged_interface/include/legged_interface/common/utils.h:82) */
} hb_model;

/* Flattened task.info / reference.info (legged_controllers/config/hunter). */
typedef struct hb_config {
  /* sqp block, task.info:79-96 */
  double dt;
  int32_t sqp_iterations;
  int32_t wbc_type;                /* 0 WeightedWbc (LeggedController.cpp:85), 1 HierarchicalWbc */
  double g_max, g_min;             /* filter line search thresholds */
  double alpha_decay, alpha_min, gamma_c, armijo_factor; /* OCS2 FilterLinesearch defaults 0.5 1e-4 1e-6 1e-4.  Any decay in (0, 1): the
                                      backtracking step sizes decay^k >= alpha_min are evaluated 16 at a time, window after window */
  /* cost, task.info:186-253 + LeggedInterface.cpp:263-312 */
  double Q_diag[HB_NX];
  double R_task_diag[24];          /* 12 contact-force weights, 12 foot-velocity (task space) weights */
  double initial_state[HB_NX];     /* configuration at which the task-space R is pulled back to joint space */
  /* soft constraints */
  double friction_mu, friction_reg, friction_gripper, friction_hess_shift; /* FrictionConeConstraint.h:77-83 */
  double friction_barrier_mu, friction_barrier_delta;                      /* task.info:255-262 */
  double soft_swing_weight;        /* task.info:265-268 */
  double pos_limit_barrier[2], vel_limit_barrier[2], force_limit_barrier[2]; /* (mu,delta) LeggedInterface.cpp:337-339 */
  double force_limit[2];           /* [0,350] LeggedInterface.cpp:352 */
  /* equality constraints */
  double position_error_gain;      /* swing normal-velocity constraint, task.info:10 */
  double zero_vel_z_gain, zero_vel_z_offset; /* Ax(2,2)=3, b(2)=-0.06, LeggedInterface.cpp:436-444 */
  double xy_ref_gain;              /* 3, LeggedRobotPreComputation.cpp:113-116 */
  /* WBC, task.info:289-333 */
  double torque_limits[5];
  double wbc_friction_mu;
  double swing_kp, swing_kd, base_height_kp, base_height_kd, base_angular_kp, base_angular_kd;
  double weight_swing_leg, weight_base_accel, weight_contact_force;
  double wbc_eps_reg;              /* Tikhonov term of the regularised-minimiser rule */
  int32_t wbc_max_iter;            /* working-set-change limit; reference nWSR = 20 (WeightedWbc.cpp:50) */
  int32_t reserved;                /* 0.  Tests and tuning only: 101 / 104 force the one- / four-wavefront backward sweep, 111 / 114 the row /
                                      wave form of the forward sweep, 120 + s trips of 2^s nodes per wavefront in the LQ kernel, 130 + L trips of L <= 16 nodes, 129 the one-node kernel (the library
                                      picks all three by the number of instances in flight; the forms are bit-identical); other values are phase-by-phase exits of the profiling build (-DHB_ABLATE) */
  double default_joint_state[HB_NJ]; /* reference.info:7-19 */
  double delta_tol;                /* sqp.deltaTol, task.info:84: the line search gives up (no step, as at alpha_min) once
                                      alpha |dx| and alpha |du| — l2 norms over the whole trajectory — are both below it
                                      ([OCS2-knowledge] SqpSolver::takeStep "escape early"); 0 disables */
  int32_t wbc_reg_steps;           /* regularisation steps after the eps-regularised WBC solve: qpOASES Options::setToMPC() leaves
                                      numRegularisationSteps = 1 (WeightedWbc.cpp:47-48, HoQp.cpp:175-176 [qpOASES-knowledge]).  Each step is
                                      one proximal-point step x <- argmin f(x) + eps/2 |x - x_prev|^2 on the final working set; 1 removes
                                      the first-order-in-eps bias of the regularised minimiser.  0 = plain Tikhonov point */
  int32_t wbc_eps_mode;            /* 0: the Tikhonov term is wbc_eps_reg for every problem (default).  1 (WeightedWbc only): per problem,
                                      eps = |H|_F * 1e3 * DBL_EPSILON with H = A_w' A_w — what qpOASES 3.2's regulariseHessian adds to the
                                      diagonal (epsRegularisation = 1e3 EPS, Options::setToMPC; [qpOASES-knowledge]); wbc_eps_reg
                                      is then unused.  Other values and mode 1 with wbc_type = 1 are rejected by hb_create */
} hb_config;
#define HB_WBC_REG_STEPS_MAX 8     /* hb_create rejects wbc_reg_steps outside [0, HB_WBC_REG_STEPS_MAX] */

typedef struct hb_ctx hb_ctx;

/* Aggregate per-phase device time of the last hb_mpc_solve / hb_wbc_update (HIP events on the
 * library's own streams), replacing the reference's mpcTimer_/wbcTimer_ (LeggedController.cpp:359-366). */
typedef struct hb_stats {
  double ms_lq, ms_riccati_bwd, ms_riccati_fwd, ms_linesearch, ms_mpc_total;
  double ms_wbc;
  int64_t n_mpc_solves, n_wbc_solves;
  int32_t n_status[4];             /* histogram of the last WBC status words */
} hb_stats;

/* ---- lifetime -------------------------------------------------------------------------------
 * Replaces LeggedController::init -> setupLeggedInterface/setupMpc/setupMrt + WeightedWbc ctor +
 * loadTasksSetting (LeggedController.cpp:41-88,376-431).  `batch` robot instances live on HIP
 * device `device`; every instance has at most `max_nodes` shooting intervals. */
int32_t hb_create(const hb_model* model, const hb_config* config, int32_t batch, int32_t max_nodes,
                  int32_t device, hb_ctx** out);
void hb_destroy(hb_ctx* ctx);
/* Message of the last call that FAILED ON THE CALLING THREAD (errno-like, thread-local): the reference drives one solver from
 * two threads (control thread: hb_wbc_update / hb_joint_command / estimator; MPC thread: hb_refgen_update / hb_mpc_solve /
 * hb_mpc_publish — LeggedController.cpp:396-421), which this library supports for exactly that split. */
const char* hb_last_error(const hb_ctx* ctx); /* ctx may be NULL: message of the failed hb_create */

/* ---- references ------------------------------------------------------------------------------
 * Node tables produced by the reference manager before each solve, replacing what
 * SwitchedModelReferenceManager::modifyReferences (SwitchedModelReferenceManager.cpp:136-171) and the
 * OCS2 time discretisation hand to the SQP solver:
 *   n_nodes[i]            number of shooting intervals N_i <= max_nodes
 *   t[i][k], k<=N_i       node times (event times are grid nodes; interval k uses mode[i][k])
 *   mode[i][k], k<N_i     contact mode of interval k (ModeSchedule::modeAtTime just after t_k)
 *   x_ref[i][k][22]       TargetTrajectories::getDesiredState(t_k)
 *   swing_ref[i][k][4][6] SwingTrajectoryPlanner::get{X,Y,Z}{position,velocity}Constraint(foot, t_k)
 * Host arrays are strided by max_nodes(+1) as declared in hb_create. */
int32_t hb_mpc_set_references(hb_ctx* ctx, int32_t inst_begin, int32_t inst_count, const int32_t* n_nodes,
                              const double* t, const int32_t* mode, const double* x_ref,
                              const double* swing_ref);

/* Cold start: x_k = x0, u_k = weight compensation of mode_k (LeggedRobotInitializer.cpp:67-77). */
int32_t hb_mpc_reset(hb_ctx* ctx, const double* x0 /*[batch][22], or NULL: the device-resident observation*/);
/* Cold start of the instances with mask[i] != 0 only — the per-instance form of LeggedController::resetMPC / resetMpcNode
 * (LeggedController.cpp:460-465), e.g. after hb_mpc_get_status reported HB_INST_NAN for them.  x0 may be NULL (the
 * device-resident observation); otherwise only the masked rows of x0 [batch][22] are read. */
int32_t hb_mpc_reset_masked(hb_ctx* ctx, const uint8_t* mask /*[batch]*/, const double* x0);
/* Per-instance status word of the last MPC call (hb_inst_status): HB_INST_NAN = non-positive Riccati pivot or a non-finite
 * value (the step was not taken, the iterate is the previous one: the batch counterpart of the exception path of the MPC
 * thread, LeggedController.cpp:413-418), HB_INST_MAXITER = the filter line search rejected every step size. */
int32_t hb_mpc_get_status(hb_ctx* ctx, int32_t* status /*[batch]*/);
/* Warm start from caller-provided trajectories (x [batch][max_nodes+1][22], u [batch][max_nodes][22]) on the CURRENT tables.
 * Between MPC calls the library warm-starts by itself: when the node tables changed (hb_mpc_set_references /
 * hb_refgen_update) the next hb_mpc_solve first interpolates the previous iterate onto the new node times and falls back to
 * the initializer beyond the previous horizon (OCS2 SqpSolver::initializeStateInputTrajectories). */
int32_t hb_mpc_set_trajectory(hb_ctx* ctx, const double* x, const double* u);

/* ---- MPC -------------------------------------------------------------------------------------
 * One MPC call = config.sqp_iterations SQP iterations (LQ approximation, constraint projection,
 * backward/forward Riccati, filter line search) for every instance, from measured state x0.
 * Replaces MPC_MRT_Interface::advanceMpc -> SqpSolver::runImpl (LeggedController.cpp:406).
 * Asynchronous on the library's MPC stream. x0 is a host pointer [batch][22] or NULL to keep the
 * device-resident x0 (previous call). */
int32_t hb_mpc_solve(hb_ctx* ctx, const double* x0);
/* Make the last solution the active policy (MPC_MRT_Interface::updatePolicy, LeggedController.cpp:154).  Enqueue-only: five
 * device-to-device copies on the MPC stream, which the WBC stream then waits for.  Real-time note for the two-thread split: called
 * while the solve is still in flight, the copies — and with them the control thread's next hb_wbc_update — queue behind the whole
 * solve.  A caller whose control tick must never wait for the solver publishes AFTER the solve has completed: hb_mpc_solve,
 * hb_mpc_get_status (synchronises the MPC stream), hb_mpc_publish, all on the MPC thread (hunter_hip.hpp MpcMrtInterface::advanceMpc
 * does exactly that); the control thread keeps evaluating the previous policy until then, as the reference does. */
int32_t hb_mpc_publish(hb_ctx* ctx);
/* Copy trajectories to the host (PrimalSolution, LeggedController.cpp:269); any pointer may be NULL.
 * x [count][max_nodes+1][22], u [count][max_nodes][22]. Synchronises the MPC stream. */
int32_t hb_mpc_get_solution(hb_ctx* ctx, int32_t inst_begin, int32_t inst_count, double* x, double* u);
/* Per-instance performance index of the accepted step: [merit, dynamics SSE, equality SSE, step size]. */
int32_t hb_mpc_get_performance(hb_ctx* ctx, double* perf /*[batch][4]*/);

/* ---- WBC -------------------------------------------------------------------------------------
 * Evaluates the published policy at t_now (MPC_MRT_Interface::evaluatePolicy, LeggedController.cpp:155)
 * unless walk_flag[i]==0, in which case the stand-still target of LeggedController.cpp:161-173 is used;
 * then runs WbcBase::update + WeightedWbc::update (legged_wbc/src/WeightedWbc.cpp:18-66) or
 * HierarchicalWbc::update (legged_wbc/src/HierarchicalWbc.cpp:18-30) for every instance.
 * Host in: t_now[batch], rbd[batch][32] (both NULL: the device-resident time / rbd), walk_flag[batch] (NULL = all
 * walking).
 * Host out (any may be NULL): sol[batch][38], x_des[batch][22], u_des[batch][22], planned_mode[batch],
 * status[batch].  Synchronous with respect to the WBC stream when an output pointer is given. */
int32_t hb_wbc_update(hb_ctx* ctx, const double* t_now, const double* rbd, const int32_t* walk_flag,
                      double dt, double* sol, double* x_des, double* u_des, int32_t* planned_mode,
                      int32_t* status);
/* Direct form of WbcBase::update(stateDesired, inputDesired, rbdStateMeasured, mode, period)
 * (legged_wbc/include/legged_wbc/WbcBase.h:43-44), batched; host pointers. */
int32_t hb_wbc_update_direct(hb_ctx* ctx, const double* x_des, const double* u_des, const double* rbd,
                             const int32_t* mode, const int32_t* stance_flag, double dt, double* sol,
                             int32_t* status);

/* ---- joint command law ------------------------------------------------------------------------
 * The per-joint command of LeggedController::update after the WBC (LeggedController.cpp:186-257, the
 * loadControllerFlag_ branch), evaluated on the results of the last hb_wbc_update / hb_wbc_update_direct /
 * hb_step_resident that are still on the device:
 *   posDes = joints(optimizedState) + 0.5 qdd_wbc dt^2,  velDes = jointVel(optimizedInput) + qdd_wbc dt   (:186-191)
 *   (kp, kd) by joint: hip roll / yaw (0,1,5,6) small gains, ankle (4,9) small kp with kd_feet, others big gains;
 *   stance or swing kp by the planned contact flag of the leg (:223-245);  feed-forward = WBC torque
 *   torque = ff + kp (posDes - q) + kd (velDes - qd)                                                       (:252-256)
 * Limit protection (:196-208): a measured joint position (rbd) more than 0.02 rad outside the urdf limits latches the
 * instance's emergency stop — only while its controller is loaded — and from that joint on, and on every later call, the
 * command is setCommand(0, 0, 0, 1, 0) (:245-248).  Unloaded controller (:209-221): MPC joint targets with kp_position /
 * kd_position (kd_feet on the ankle joints 4, 9), no feed-forward.
 * Outputs (any may be NULL) are [batch][10]. Gains default to legged_controllers/cfg/Tutorials.cfg:6-16. */
typedef struct hb_joint_gains {
  double kp_big_stance, kp_big_swing, kd_big, kp_small_stance, kp_small_swing, kd_small, kd_feet;
  double kp_position, kd_position;   /* unloaded-controller branch (Tutorials.cfg:6-7) */
} hb_joint_gains;
/* Per-instance controller flags of the joint command law, device-resident: loadControllerFlag_ (default 1 = loaded; the
 * reference starts at 0 until /load_controller, LeggedController.cpp:489-493) and the emergency-stop latch
 * (emergencyStopFlag_, also set by the /emergency_stop topic, :477-481).  Either array may be NULL (left as is). */
int32_t hb_joint_set_flags(hb_ctx* ctx, const int32_t* controller_loaded /*[batch]*/, const int32_t* emergency_stop /*[batch]*/);
int32_t hb_joint_get_emergency_stop(hb_ctx* ctx, int32_t* emergency_stop /*[batch]*/);
int32_t hb_joint_command(hb_ctx* ctx, const hb_joint_gains* gains, double dt, double* pos_des, double* vel_des,
                         double* kp, double* kd, double* tau_ff, double* torque);

/* ---- plant stub for closed-loop rollouts ---------------------------------------------------
 * The reference closes its loop through Gazebo / MuJoCo (legged_gazebo/src/LeggedHWSim.cpp:166-192,
 * mujoco/src/main.cc:247).  This stub integrates M(q) vdot + nle = S' tau + Jc' lambda with the contact points of the
 * commanded mode pinned by acceleration-level constraints (Baumgarte gain `baumgarte`, damped normal equations with
 * relative damping `eps`), semi-implicit Euler; it does NOT enforce unilateral contact or friction limits.
 * Coordinates: q = [pos, zyx, joints], v = [v_lin (world), ZYX rates, joint rates]. */
int32_t hb_plant_reset(hb_ctx* ctx, const double* q0 /*[batch][16]*/, const double* v0 /*[batch][16] or NULL*/,
                       double baumgarte, double eps);
/* Advance every instance by dt in `substeps` substeps.  tau[batch][10] / contact[batch][4] may be NULL: the torque of the
 * last hb_joint_command and the planned contact flags of the last WBC call, still on the device, are used.
 * to_resident != 0 repacks the new state as rbd + MPC observation into the resident inputs and advances the resident time
 * by dt, so that hb_refgen_update(x_now = NULL), hb_mpc_solve(NULL), hb_wbc_update(t_now = NULL, rbd = NULL),
 * hb_joint_command and hb_plant_step(NULL, NULL) close the loop without a host round trip. */
int32_t hb_plant_step(hb_ctx* ctx, const double* tau, const int32_t* contact, double dt, int32_t substeps,
                      int32_t to_resident);
/* State to the host (any may be NULL): q[batch][16], v[batch][16], rbd[batch][32], lambda[batch][12] (last contact
 * forces), vdot[batch][16] (last acceleration). */
int32_t hb_plant_get_state(hb_ctx* ctx, double* q, double* v, double* rbd, double* lambda, double* vdot);

/* ---- device-resident stepping (bench / rollouts; inputs already in HBM) ------------------------
 * hb_step_resident runs hb_mpc_solve(NULL) + hb_mpc_publish + WBC on device-resident t_now/rbd that
 * were uploaded once with hb_set_resident_inputs; nothing crosses PCIe. */
int32_t hb_set_resident_inputs(hb_ctx* ctx, const double* x0, const double* t_now, const double* rbd,
                               const int32_t* walk_flag);
/* The device-resident controller time alone ([batch], host).  hb_plant_step(to_resident) advances it by itself; an estimator
 * with to_resident (which replaces the resident observation and rbd state but knows no clock) leaves it to the caller.
 * Enqueue-only (pinned staging, no device synchronisation). */
int32_t hb_set_resident_time(hb_ctx* ctx, const double* t_now);
int32_t hb_step_resident(hb_ctx* ctx, double dt);
/* One whole tick on the resident state, enqueue-only: hb_set_resident_time(t_now) + hb_estimator_update(dt_est, sensors,
 * to_resident) + hb_refgen_update(t_now, horizon, x_now = the estimate, cmd_vel) + hb_step_resident(dt_wbc) — what
 * LeggedController::update and its MPC thread do per MPC period (LeggedController.cpp:137-185, 396-412) for the whole batch.  With
 * instance ranges (hb_set_chunks > 1) every range runs its slice of ALL of it on its own stream, tick after tick, without a
 * whole-batch stage between two steps; results are identical to the four calls.  Host arrays as in hb_estimator_update /
 * hb_refgen_update; they are the caller's again on return (pinned staging). */
int32_t hb_tick_resident(hb_ctx* ctx, double dt_est, const double* quat, const double* ang_vel_local, const double* lin_acc_local,
                         const double* joint_pos, const double* joint_vel, const int32_t* contact_flag, const double* t_now,
                         double horizon, const double* cmd_vel, double dt_wbc);
/* Optional: a device-resident cyclic sequence of measured states x0_seq[n_seq][batch][22]; step k of
 * hb_step_resident starts its MPC solve from x0_seq[k % n_seq] (emulates the estimator feeding a new state each
 * MPC call, LeggedController.cpp:141-144).  n_seq = 0 disables it. */
int32_t hb_set_resident_x0_sequence(hb_ctx* ctx, int32_t n_seq, const double* x0_seq);
int32_t hb_get_wbc_solution(hb_ctx* ctx, double* sol /*[batch][38]*/, int32_t* status /*[batch]*/);
/* Active-set iterations of the last WBC solve of every instance (constraint additions + drops of the dual active-set
 * method; the role of nWSR at WeightedWbc.cpp:51-55).  iters: [batch]. */
int32_t hb_get_wbc_iterations(hb_ctx* ctx, int32_t* iters /*[batch]*/);
/* Pipelining of hb_step_resident: the batch is cut into n_chunks (1..8) instance ranges, each a linear
 * MPC -> publish -> WBC sequence on its own HIP stream so that the per-instance sweeps of one range overlap the
 * per-node kernels of another.  Results are identical for every n_chunks; hb_get_stats phase times are only
 * recorded with n_chunks = 1 (the default).
 * THREADING: hb_tick_resident and hb_step_resident with n_chunks > 1 are SINGLE-THREAD entry points — they share the pinned
 * staging rings and the lazy range join with the enqueue-only calls (hb_set_resident_time, hb_estimator_update, hb_refgen_update),
 * which are not locked.  The two-thread split of hb_last_error's note (MPC thread / control thread) applies to n_chunks = 1 and
 * the non-resident entry points only; do not mix it with chunked stepping on one context. */
int32_t hb_set_chunks(hb_ctx* ctx, int32_t n_chunks);

/* ---- state estimator (the step immediately before the path every tick) ------------------
 * Batched KalmanFilterEstimate::update (legged_estimation/src/LinearKalmanFilter.cpp:72-184): 18-state
 * [base pos, base vel, 4 foot positions] / 28-measurement linear Kalman filter on leg kinematics, preceded by the
 * sensor packing of StateEstimateBase::{updateJointStates, updateImu} (StateEstimateBase.cpp:73-106) and followed by
 * the centroidal-state conversion + yaw unwrapping of LeggedController::updateStateEstimation
 * (LeggedController.cpp:331-334).  The filter state (xHat[18], P[18][18], last yaw) is device-resident per
 * instance.  ROS topics / tf (updateFromTopic) and the contact-force estimator are not part of this entry point. */
typedef struct hb_estimator_config {  /* task.info kalmanFilter block (:336-345); LinearKalmanFilter.h:50-56 */
  double foot_radius;
  double imu_process_noise_position, imu_process_noise_velocity, foot_process_noise_position;
  double foot_sensor_noise_position, foot_sensor_noise_velocity, foot_height_sensor_noise;
  /* task.info contactForceEsimation block (:347-351), StateEstimateBase::loadSettings (StateEstimateBase.cpp:365-377) */
  double contact_force_cutoff_frequency;   /* lambda of the momentum observer's low pass (hb_estimator_contact_force) */
  double contact_threshold;                /* normal force above which estContactState would call a leg "in contact" (:206-226) */
} hb_estimator_config;
/* (Re)initialise: xHat = x_hat0 (or zeros if NULL), P = 100 I, last yaw = 0 (LinearKalmanFilter.cpp:31-60). */
int32_t hb_estimator_reset(hb_ctx* ctx, const hb_estimator_config* cfg, const double* x_hat0 /*[batch][18] or NULL*/);
/* One filter step for every instance.  Host in: quat[batch][4] (x y z w), ang_vel_local[batch][3],
 * lin_acc_local[batch][3], joint_pos[batch][10], joint_vel[batch][10], contact_flag[batch][4] (contact order
 * L_f1 R_f1 L_f2 R_f2).  Host out (either may be NULL): rbd[batch][32] (the vector WbcBase::update takes),
 * x_state[batch][22] (the MPC observation state).  to_resident != 0 additionally stores both into the
 * device-resident inputs of hb_step_resident, so that estimate -> MPC -> WBC never leaves the GPU.
 * With rbd == NULL and x_state == NULL the call is ENQUEUE-ONLY (sensor arrays through pinned staging, no device
 * synchronisation; see hb_refgen_update). */
int32_t hb_estimator_update(hb_ctx* ctx, double dt, const double* quat, const double* ang_vel_local,
                            const double* lin_acc_local, const double* joint_pos, const double* joint_vel,
                            const int32_t* contact_flag, int32_t to_resident, double* rbd, double* x_state);
/* StateEstimateBase::setCmdTorque + estContactForce (legged_estimation/src/StateEstimateBase.cpp:130-206), which
 * LeggedController::updateStateEstimation runs every tick behind the filter update (LeggedController.cpp:344-345): a
 * generalised-momentum observer for the disturbance torque (low pass exp(-cutoff dt) from hb_estimator_config, state per
 * instance on the device, zeroed by hb_estimator_reset) and, per leg, the minimum-norm wrench at its first contact frame
 * (L_f1 / R_f1).  rbd [batch][32] or NULL = the state the last hb_estimator_update left on the device;
 * joint_torque [batch][10] = the measured joint efforts.  Out (either may be NULL): est_disturbance_torque [batch][16]
 * (estDisturbancetorque_), est_contact_force [batch][16] = [wrench leg 0 (force 3, moment 3) | wrench leg 1 | |F0| |F1| |
 * |W0| |W1|] (estContactforce_).  In the reference nothing reads these values (estContactState, their only reader, is
 * never called); they are provided for the same diagnostics. */
int32_t hb_estimator_contact_force(hb_ctx* ctx, double dt, const double* rbd, const double* joint_torque,
                                   double* est_disturbance_torque, double* est_contact_force);
/* Filter state to the host (either may be NULL): x_hat[batch][18], P[batch][18][18]. */
int32_t hb_estimator_get_filter(hb_ctx* ctx, double* x_hat, double* P);

/* ---- reference generation on the device ---------------------------------------------------
 * What SwitchedModelReferenceManager::modifyReferences (SwitchedModelReferenceManager.cpp:136-171) produces before
 * every MPC call, for the whole batch, written straight into the node tables hb_mpc_set_references would upload:
 * 2-knot target from cmd_vel (TargetTrajectoriesPublisher.h:101-131), event-clipped shooting grid, swing planner
 * (footholds: SwingTrajectoryPlanner::calNextFootPos; x/y/z multi-node cubic splines: genSwingTrajs,
 * SwingTrajectoryPlanner.cpp:164-358).  The gait scheduler (GaitSchedule.cpp:57-161, a few integers and event times per
 * instance) stays on the host: its output, the mode schedule, is an input.  With joint_ik the targets are resampled
 * every 0.15 s and their joint part replaced by the inverse kinematics of the planned foot positions
 * (calculateJointRef), warm-started knot to knot. */
#define HB_MAX_EVENTS 64
typedef struct hb_refgen_config {  /* reference.info comHeight / defaultJointState, task.info swing_trajectory_config */
  double dt, com_height, next_position_z, swing_height, swing_time_scale;
  double feet_bias[HB_NC][3];      /* (feet_bias_x1|x2, +-feet_bias_y, feet_bias_z) in contact order */
  double default_joints[HB_NJ];
  int32_t joint_ik;                /* 1: per-knot joint reference by inverse kinematics (calculateJointRef,
                                      SwitchedModelReferenceManager.cpp:251-300; InverseKinematics.cpp:20-231), 0: defaultJointState */
  int32_t reserved;
} hb_refgen_config;
/* Planner state: latest_stance[batch][4][3] (SwingTrajectoryPlanner::latestStanceposition_), or NULL to take the
 * current foot positions at the first hb_refgen_update. */
int32_t hb_refgen_reset(hb_ctx* ctx, const hb_refgen_config* cfg, const double* latest_stance);
/* Mode schedule of instances [inst_begin, inst_begin + inst_count): n_events[i] <= HB_MAX_EVENTS event times
 * (strictly increasing) and n_events[i] + 1 modes; arrays strided by HB_MAX_EVENTS / HB_MAX_EVENTS + 1. */
int32_t hb_refgen_set_schedule(hb_ctx* ctx, int32_t inst_begin, int32_t inst_count, const int32_t* n_events,
                               const double* event_times, const int32_t* modes);
/* Generate the references of every instance for the horizon [t0[i], t0[i] + horizon].  x_now[batch][22] is the
 * observation (NULL: the device-resident x0, e.g. the estimator's output); cmd_vel[batch][4] = (vx, vy, vz, yaw rate).
 * status[batch] (may be NULL): 0 ok, 1 a swing phase runs out of the schedule, 2 grid longer than max_nodes.
 * With status == NULL the call is ENQUEUE-ONLY: the host arrays are copied into library-owned pinned staging (they are the
 * caller's again on return), no device synchronisation takes place, and the status words are read later with
 * hb_refgen_get_status — a driver that feeds a batch every tick can then run a few ticks ahead of the device. */
int32_t hb_refgen_update(hb_ctx* ctx, const double* t0, double horizon, const double* x_now, const double* cmd_vel,
                         int32_t* status);
/* Status words of the last hb_refgen_update (synchronises). */
int32_t hb_refgen_get_status(hb_ctx* ctx, int32_t* status /*[batch]*/);
/* Node tables back to the host (any pointer may be NULL); layouts as in hb_mpc_set_references. */
int32_t hb_mpc_get_references(hb_ctx* ctx, int32_t inst_begin, int32_t inst_count, int32_t* n_nodes, double* t,
                              int32_t* mode, double* x_ref, double* swing_ref);

/* ---- misc ------------------------------------------------------------------------------------ */
int32_t hb_sync(hb_ctx* ctx);
int32_t hb_get_stats(hb_ctx* ctx, hb_stats* out);
/* Joint-space input cost R (22x22) built at init from R_task_diag (LeggedInterface.cpp:263-290). */
int32_t hb_get_input_cost(const hb_ctx* ctx, double* R /*[22][22]*/);
/* Library/ABI version: major*10000 + minor*100 + patch. */
int32_t hb_version(void);

/* ---- unit-level entry points used by the parity tests (each is one HIP kernel launch) ----------- */
/* Centroidal flow map and its Jacobians (PinocchioCentroidalDynamicsAD, LeggedRobotDynamicsAD.cpp:57-70). */
int32_t hb_eval_flow_map(hb_ctx* ctx, int32_t n, const double* x, const double* u, double* f /*[n][22]*/,
                         double* dfdx /*[n][22][22] or NULL*/, double* dfdu /*[n][22][22] or NULL*/);
/* Foot positions / velocities (PinocchioEndEffectorKinematicsCppAd, EndEffectorLinearConstraint.cpp:105-128). */
int32_t hb_eval_foot_kinematics(hb_ctx* ctx, int32_t n, const double* x, const double* u,
                                double* pos /*[n][4][3]*/, double* vel /*[n][4][3]*/);
/* Rigid-body quantities of WbcBase::updateMeasured (WbcBase.cpp:70-120): M[16][16], nle[16], J[12][16], dJv[12]. */
int32_t hb_eval_rbd(hb_ctx* ctx, int32_t n, const double* rbd, double* M, double* nle, double* J, double* dJv);
/* MPC observation state from the rbd state: x = [A(q) v / m, base pose, joints]
 * (CentroidalModelRbdConversions::computeCentroidalStateFromRbdModel, call site LeggedController.cpp:332); no yaw
 * unwrapping. */
int32_t hb_centroidal_state_from_rbd(hb_ctx* ctx, int32_t n, const double* rbd /*[n][32]*/, double* x /*[n][22]*/);
/* Solve a batch of equality-free LQ problems (n <= batch, N <= max_nodes, nu <= 12 inputs per stage) with the
 * Riccati backward kernel (HPIPM's role).  Stage data row-major: A[n][N][22][22], B[n][N][22][nu],
 * b[n][N][22], Q[n][N][22][22], R[n][N][nu][nu], P[n][N][nu][22], q[n][N][22], r[n][N][nu], dx0[n][22].
 * Clobbers the MPC reference tables of the context (call hb_mpc_set_references again afterwards). */
int32_t hb_riccati_solve(hb_ctx* ctx, int32_t n, int32_t N, int32_t nu, const double* A, const double* B,
                         const double* b, const double* Q, const double* R, const double* P, const double* q,
                         const double* r, const double* dx0, double* dx /*[n][N+1][22]*/, double* du /*[n][N][nu]*/);
/* Generic hierarchical QP cascade on small dense tasks (HoQp.cpp:21-198; HoQp.h:24-89): n_problems independent stacks of
 * n_levels <= 3 tasks {A x = b in the least-squares sense, D x <= f with slack} on n_vars <= 8 variables, m_eq[l] / m_in[l]
 * <= 8 rows per level (the same shape for every problem).  Blocks are padded: A, D [n_problems][3][8][8] row-major, b, f
 * [n_problems][3][8].  Outputs: x [n_problems][3][8] = solution after each level (HoQp::getSolutions of that level),
 * slack [n_problems][3][8] = the level's own slack (HoQp::getStackedSlackSolutions tail), status [n_problems] = hb_inst_status.
 * The device counterpart of the reference's unit test legged_wbc/test/HoQp_test.cpp:18-55; runs the building blocks of the
 * HierarchicalWbc kernel (hb_config.wbc_type = 1) on plain matrices. */
int32_t hb_hoqp_solve(hb_ctx* ctx, int32_t n_problems, int32_t n_vars, int32_t n_levels, const int32_t* m_eq, const int32_t* m_in,
                      const double* A, const double* b, const double* D, const double* f, double* x, double* slack, int32_t* status);
/* Diagnostics of the chunked hb_step_resident: out4 = [graph launches, directly enqueued chunk steps, forks from the library streams,
 * graph captures] since hb_create. */
int32_t hb_debug_chunk_counters(hb_ctx* ctx, int64_t* out4);
/* out2 = [graph captures / instantiations that FAILED since hb_create, 1 if this context has therefore given up on graphs and steps
 * its ranges with direct launches (until the next hb_set_chunks)].  A failed capture is not retried on every step. */
int32_t hb_debug_graph_state(hb_ctx* ctx, int64_t* out2);
/* n independent inverse-kinematics problems of the joint-reference generator (InverseKinematics::computeIK(init_q, leg, pos, R_des),
 * legged_interface/src/foot_planner/InverseKinematics.cpp:36-231): q16[n][16] = [base pos, zyx, joints] start configurations,
 * leg[n] in {0 left, 1 right}, des_pos[n][3] target of contact f1 of the leg, R_des[n][9] row-major desired foot rotation;
 * out5[n][5] = the leg's joint angles.  Runs the lane-cooperative device routine that hb_refgen_update uses per knot; it is
 * the entry the reference-compiled golden vectors (tests/golden/ref_ik.json) are checked through. */
int32_t hb_ik_solve(hb_ctx* ctx, int32_t n, const double* q16, const int32_t* leg, const double* des_pos, const double* R_des, double* out5);
/* QP step of the last SQP iteration (before the line search scaled it): dx[batch][max_nodes+1][22],
 * du[batch][max_nodes][22]; either may be NULL. */
int32_t hb_mpc_get_step(hb_ctx* ctx, double* dx, double* du);

#ifdef __cplusplus
}
#endif
#endif /* HUNTER_HIP_H */
