"""TEST INFRASTRUCTURE: the closed loop of hunter_bipedal_control_amd/rollout.py with every controller stage taken from
the CPU oracle and the host reference manager (refgen.py) — the independent twin of DeviceLoop."""
import numpy as np

from hunter_bipedal_control_amd import abi
from oracle import refgen
from oracle.plant import Plant


def standing_configuration(params, batch):
    """q[B][16]: initialState of task.info, base lowered so that the mean contact-point height is zero (oracle kinematics)."""
    x0 = np.array(params["config"]["initial_state"], dtype=float)
    q = np.zeros((batch, 16))
    q[:, 0:3], q[:, 3:6], q[:, 6:] = x0[6:9], x0[9:12], x0[12:]
    q[:, 2] -= refgen.foot_positions(params["model"], x0)[:, 2].mean()
    return q


def warm_shift(params, prev_refs, xs, us, refs):
    """The previous call's iterate brought onto the new node tables (device: k_warm_shift; OCS2
    SqpSolver::initializeStateInputTrajectories): states interpolated linearly in time, inputs interpolated at the interval
    start but HELD across a mode switch of the previous solution and on its last interval; beyond the previous horizon the
    state is carried on and the input is the weight compensation of the interval's mode."""
    n, npv = int(refs["n_nodes"][0]), int(prev_refs["n_nodes"][0])
    t, tp, mp = refs["t"][0], prev_refs["t"][0], prev_refs["mode"][0]
    m = sum(params["model"]["mass"])
    xn, un = np.zeros_like(xs), np.zeros_like(us)
    for k in range(n + 1):
        tk = t[k]
        i = 0
        while i + 1 < npv and tp[i + 1] <= tk:
            i += 1
        if tk >= tp[npv]:
            xn[0, k] = xs[0, npv]
        elif tk <= tp[0]:
            xn[0, k] = xs[0, 0]
        else:
            a = (tk - tp[i]) / (tp[i + 1] - tp[i])
            xn[0, k] = (1.0 - a) * xs[0, i] + a * xs[0, i + 1]
        if k < n:
            if tk >= tp[npv]:
                cf = refgen.mode_to_contact_flags(int(refs["mode"][0, k]))
                u = np.zeros(22)
                for c in range(4):
                    if cf[c]:
                        u[3 * c + 2] = m * 9.81 / sum(cf)
                un[0, k] = u
            elif tk <= tp[0]:
                un[0, k] = us[0, 0]
            elif i + 1 >= npv or mp[i + 1] != mp[i]:
                un[0, k] = us[0, i]
            else:
                a = (tk - tp[i]) / (tp[i + 1] - tp[i])
                un[0, k] = (1.0 - a) * us[0, i] + a * us[0, i + 1]
    return xn, un


class OracleLoop:
    def __init__(self, oracle, params, gait, cmd_vel, n_intervals=100, mpc_every=8, dt=0.002, t_gait_start=0.3):
        self.o, self.params, self.gait, self.cmd = oracle, params, gait, tuple(cmd_vel)
        self.N, self.mpc_every, self.dt, self.t_gait_start = n_intervals, mpc_every, dt, t_gait_start
        self.horizon = n_intervals * params["config"]["dt"]
        self.gains = abi.make_joint_gains()
        model = params["model"]

        def foot_fn(q):
            out = np.zeros((q.shape[0], 4, 3))
            for i in range(q.shape[0]):
                x = np.zeros(22)
                x[6:9], x[9:12], x[12:] = q[i, 0:3], q[i, 3:6], q[i, 6:]
                out[i] = refgen.foot_positions(model, x)
            return out

        self.plant = Plant(lambda rbd: oracle.rbd(rbd), foot_fn, standing_configuration(params, 1))
        self.planner = None
        self.sched = refgen.gait_schedule(params, gait, t_gait_start, 1.0e3 if gait == "stance" else 40.0)
        self.t, self.tick = 0.0, 0
        self.xs = self.us = None
        self.pol = None
        self.last = {}

    def _references(self, x_obs):
        c = self.params["config"]
        nmax = self.N + 8
        targets = refgen.cmd_vel_targets(self.t, x_obs, self.cmd, self.horizon, c["com_height"], c["default_joint_state"])
        if self.planner is None:
            self.planner = refgen.SwingTrajectoryPlanner(c["swing"])
            self.planner.latest_stance = [f.copy() for f in refgen.foot_positions(self.params["model"], x_obs)]
        self.planner.body_vel_cmd = np.array([self.cmd[0], self.cmd[1], self.cmd[2], self.cmd[3], 0.0, 0.0])
        self.planner.current_feet = list(refgen.foot_positions(self.params["model"], x_obs))
        self.planner.update(self.sched, targets, self.t)
        targets = refgen.joint_reference_ik(self.params, targets, self.planner, self.t, self.t + self.horizon, x_obs)
        return refgen.stack_tables([refgen.build_node_tables(self.t, self.horizon, c["dt"], self.sched, targets, self.planner, nmax)])

    def step(self):
        o, dt = self.o, self.dt
        rbd = self.plant.rbd()
        x_obs = o.centroidal_state_from_rbd(rbd)
        if self.tick % self.mpc_every == 0:
            refs = self._references(x_obs[0])
            nmax = refs["mode"].shape[1]
            if self.xs is None:
                self.xs, self.us = np.zeros((1, nmax + 1, 22)), np.zeros((1, nmax, 22))
                n = refs["n_nodes"][0]
                self.xs[0, :n + 1], self.us[0, :n] = o.cold_start(refs["mode"][0, :n], x_obs[0])
            else:
                self.xs, self.us = warm_shift(self.params, self.pol[0], self.xs, self.us, refs)
            o.mpc_solve(refs, x_obs, self.xs, self.us, iters=1)
            self.pol = (refs, self.xs.copy(), self.us.copy())
        refs, px, pu = self.pol
        tt, n = refs["t"][0], refs["n_nodes"][0]
        k = 0
        while k < n - 1 and self.t >= tt[k + 1]:
            k += 1
        a = (self.t - tt[k]) / (tt[k + 1] - tt[k])
        xd = (1 - a) * px[0, k] + a * px[0, k + 1]
        ud = (1 - a) * pu[0, k] + a * pu[0, min(k + 1, n - 1)]
        md = int(refs["mode"][0, k])
        sol, st, _ = o.wbc_update(xd[None], ud[None], rbd, np.array([md], dtype=np.int32), stance_flag=np.zeros(1, dtype=np.int32))
        qdd, tau = sol[0, 6:16], sol[0, 28:38]
        cf = refgen.mode_to_contact_flags(md)
        g = self.gains
        kp, kd = np.zeros(10), np.zeros(10)
        for j in range(10):
            c = cf[j // 5]
            if j in (0, 1, 5, 6):
                kp[j], kd[j] = (g.kp_small_stance if c else g.kp_small_swing), g.kd_small
            elif j in (4, 9):
                kp[j], kd[j] = (g.kp_small_stance if c else g.kp_small_swing), g.kd_feet
            else:
                kp[j], kd[j] = (g.kp_big_stance if c else g.kp_big_swing), g.kd_big
        pos, vel = xd[12:] + 0.5 * qdd * dt * dt, ud[12:] + qdd * dt
        torque = tau + kp * (pos - rbd[0, 6:16]) + kd * (vel - rbd[0, 22:32])
        self.plant.step(torque[None], np.array([cf]), dt)
        self.t += dt
        self.tick += 1
        self.last = dict(sol=sol, status=st, mode=md, torque=torque)
        return self.plant.q, self.plant.v
