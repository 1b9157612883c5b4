"""Closed-loop batched rollouts through the C ABI: plant -> observation -> references -> MPC ->
policy -> WBC -> joint command -> plant, the loop of LeggedController::update (legged_controllers/src/LeggedController.cpp:
137-278) and its MPC thread (:396-412), for every instance of a HunterSolver.

Everything between the plant's state and the joint torque runs on the device (reference generation, SQP, policy
evaluation, WBC, joint command law); the plant stub integrates either on the device (ResidentLoop, hb_plant_step) or on the host (DeviceLoop, with a plant object the
caller injects — the tests pass the numpy twin oracle/plant.py)  with the device's rigid-body
terms.  The observation is the plant's true state (no estimator noise); `hb_estimator_update` can be put in its place.
"""
from __future__ import annotations

import numpy as np

from . import abi, gait
from . import solver as _solver
from .gait import schedule_window  # noqa: F401  (re-exported: the tests address it as rollout.schedule_window)


def standing_configuration(params: dict, batch: int, solver) -> np.ndarray:
    """q[B][16]: initialState of task.info with the base lowered so that the mean contact-point height is zero (contact
    points from the device kinematics, hb_eval_foot_kinematics)."""
    x0 = np.array(params["config"]["initial_state"], dtype=float)
    q = np.zeros((batch, 16))
    q[:, 0:3], q[:, 3:6], q[:, 6:] = x0[6:9], x0[9:12], x0[12:]
    feet = solver.eval_foot_kinematics(x0[None, :], np.zeros((1, 22)))[0]
    q[:, 2] -= np.asarray(feet).reshape(4, 3)[:, 2].mean()
    return q


class DeviceLoop:
    """One HunterSolver driven in closed loop.  `gait` per instance (names of gait.info), `cmd_vel` [B][4]."""

    def __init__(self, solver, params: dict, gaits, cmd_vel, n_intervals: int = 100, mpc_every: int = 8, dt: float = 0.002,
                 t_gait_start: float = 0.3, joint_ik: bool = True, use_estimator: bool = False, use_lcm: bool = False,
                 plant_factory=None):
        """use_lcm (with use_estimator): sensors and commands cross the controller boundary as LCM wire images — the
        plant side encodes low_state_t / applies the PD + feed-forward law of a decoded low_cmd_t like the MuJoCo bridge
        (mujoco/src/main.cc), the controller side uses hb_estimator_update_lcm / hb_joint_command_lcm."""
        assert not (use_lcm and not use_estimator)
        if plant_factory is None:
            raise ValueError("DeviceLoop integrates the plant on the host: pass plant_factory(rbd_fn, foot_fn, q0) "
                             "(tests: oracle.plant.Plant); ResidentLoop keeps the plant on the device")
        self.use_lcm = use_lcm
        self.tau_applied = np.zeros((solver.B, 10))
        self.s, self.params = solver, params
        self.B = solver.B
        c = params["config"]
        self.horizon = n_intervals * c["dt"]
        self.dt, self.mpc_every = dt, mpc_every
        self.cmd = np.ascontiguousarray(cmd_vel, dtype=float).reshape(self.B, 4)
        self.gains = abi.make_joint_gains()
        self.t, self.tick = 0.0, 0
        solver.refgen_reset(abi.make_refgen_config(params, joint_ik=joint_ik))
        # GaitSchedule output per instance; each MPC call hands the device the window [t - 1, t + horizon + 1.5] of it
        # (the reference asks its gait schedule for [t - T, t + 2T], SwitchedModelReferenceManager.cpp:147)
        self.schedules = [gait.gait_schedule(params, g, t_gait_start, 1.0e3 if g == "stance" else 60.0) for g in gaits]
        zeros_u = np.zeros((1, 22))

        def foot_fn(q):
            x = np.zeros((q.shape[0], 22))
            x[:, 6:9], x[:, 9:12], x[:, 12:] = q[:, 0:3], q[:, 3:6], q[:, 6:]
            return solver.eval_foot_kinematics(x, np.repeat(zeros_u, q.shape[0], axis=0))[0]

        self.plant = plant_factory(lambda rbd: solver.eval_rbd(rbd), foot_fn, standing_configuration(params, self.B, solver))
        self.started = False
        self.last = {}
        # use_estimator: the observation comes from hb_estimator_update fed with the plant's ideal IMU, joint encoders and
        # the commanded contact flags (LeggedController::updateStateEstimation) instead of the plant's true state
        self.use_estimator = use_estimator
        self.contact = np.ones((self.B, 4), dtype=np.int32)
        if use_estimator:
            q = self.plant.q
            xh0 = np.zeros((self.B, 18))
            xh0[:, 0:3] = q[:, 0:3]
            feet = self.plant.foot_fn(q)
            xh0[:, 6:18] = feet.reshape(self.B, 12)
            solver.estimator_reset(abi.make_estimator_config(params), xh0)

    def step(self):
        s, B = self.s, self.B
        if self.use_estimator:
            quat, w_loc, a_loc = self.plant.imu()
            if self.use_lcm:   # LcmInterface.cpp:54-70: quaternion (w x y z), gyroscope, accelerometer, joint pos / vel / torque
                fields = np.concatenate([quat[:, 3:4], quat[:, 0:3], w_loc, a_loc, self.plant.q[:, 6:], self.plant.v[:, 6:],
                                         self.tau_applied], axis=1)
                wire = _solver.lcm_encode(_solver.LCM_LOW_STATE, int(round(self.t * 1e9)), fields)
                rbd, x_obs, _ = s.estimator_update_lcm(self.dt, wire, self.contact)
            else:
                rbd, x_obs = s.estimator_update(self.dt, quat, w_loc, a_loc, self.plant.q[:, 6:], self.plant.v[:, 6:], self.contact)
        else:
            rbd = self.plant.rbd()
            x_obs = s.centroidal_state_from_rbd(rbd)
        if self.tick % self.mpc_every == 0:                       # MPC thread: references, one SQP iteration, publish
            s.refgen_set_schedule([schedule_window(ms, self.t - 1.0, self.t + self.horizon + 1.5) for ms in self.schedules])
            status = s.refgen_update(np.full(B, self.t), self.horizon, x_obs, self.cmd)
            if status.max() != 0:
                raise RuntimeError(f"reference generation failed: {status}")
            if not self.started:
                s.reset(x_obs)                                    # cold start (LeggedRobotInitializer)
                self.started = True
            s.mpc_solve(x_obs)
            s.publish()
        out = s.wbc_update(np.full(B, self.t), rbd, dt=self.dt)   # control thread: policy, WBC, joint command
        if self.use_lcm:
            wire = s.joint_command_lcm(self.gains, self.dt, int(round(self.t * 1e9)))
            _, f = _solver.lcm_decode(_solver.LCM_LOW_CMD, wire)      # plant side: PD + feed-forward on the decoded command
            cmd = dict(pos_des=f[:, 0:10], vel_des=f[:, 10:20], tau_ff=f[:, 30:40], kp=f[:, 40:50], kd=f[:, 50:60])
            cmd["torque"] = cmd["tau_ff"] + cmd["kp"] * (cmd["pos_des"] - self.plant.q[:, 6:]) + cmd["kd"] * (cmd["vel_des"] - self.plant.v[:, 6:])
            self.tau_applied = cmd["torque"]
        else:
            cmd = s.joint_command(self.gains, self.dt)
        contact = np.array([gait.mode_to_contact_flags(int(m)) for m in out["mode"]])
        self.plant.step(cmd["torque"], contact, self.dt)
        self.contact = contact.astype(np.int32)
        self.t += self.dt
        self.tick += 1
        self.last = dict(out=out, cmd=cmd, contact=contact, x_obs=x_obs)
        return self.plant.q, self.plant.v


class ResidentLoop:
    """The same loop with the plant stub on the device too (hb_plant_step): nothing but the reference-generation time
    stamps, the commands and the mode-schedule windows crosses PCIe."""

    def __init__(self, solver, params: dict, gaits, cmd_vel, n_intervals: int = 100, mpc_every: int = 8, dt: float = 0.002,
                 t_gait_start: float = 0.3, joint_ik: bool = True, substeps: int = 4, static_schedule_until: float = 0.0):
        """static_schedule_until > 0: the mode schedules are uploaded once for [-1, static_schedule_until] (at most
        HB_MAX_EVENTS events) instead of a sliding window per MPC call — no per-call host work for large batches."""
        self.s, self.params, self.B = solver, params, solver.B
        self.horizon = n_intervals * params["config"]["dt"]
        self.dt, self.mpc_every, self.substeps = dt, mpc_every, substeps
        self.cmd = np.ascontiguousarray(cmd_vel, dtype=float).reshape(self.B, 4)
        self.gains = abi.make_joint_gains()
        self.t, self.tick = 0.0, 0
        solver.refgen_reset(abi.make_refgen_config(params, joint_ik=joint_ik))
        self.schedules = [gait.gait_schedule(params, g, t_gait_start, 1.0e3 if g == "stance" else 60.0) for g in gaits]
        q0 = standing_configuration(params, self.B, solver)
        solver.plant_reset(q0)
        # resident observation of the initial state
        rbd = np.zeros((self.B, 32))
        rbd[:, 0:3], rbd[:, 3:6], rbd[:, 6:16] = q0[:, 3:6], q0[:, 0:3], q0[:, 6:]
        solver.set_resident_inputs(solver.centroidal_state_from_rbd(rbd), np.zeros(self.B), rbd)
        self.started = False
        self.static = static_schedule_until > 0.0
        if self.static:
            solver.refgen_set_schedule([schedule_window(ms, -1.0, static_schedule_until) for ms in self.schedules])

    def step(self):
        s = self.s
        if self.tick % self.mpc_every == 0:
            if not self.static:
                s.refgen_set_schedule([schedule_window(ms, self.t - 1.0, self.t + self.horizon + 1.5) for ms in self.schedules])
            status = s.refgen_update(np.full(self.B, self.t), self.horizon, None, self.cmd)
            if status.max() != 0:
                raise RuntimeError(f"reference generation failed: {status}")
            if not self.started:
                s.reset_resident()
                self.started = True
            s.mpc_solve(None)
            s.publish()
        s.wbc_update_resident(self.dt)
        s.joint_command_resident(self.gains, self.dt)
        s.plant_step(None, None, self.dt, self.substeps, to_resident=True)
        self.t += self.dt
        self.tick += 1
