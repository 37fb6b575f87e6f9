"""TEST INFRASTRUCTURE -- CPU oracle for the ungar_amd hot path.  NOT part of the product.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (ungar_amd/) never does and fails loudly if its HIP library is missing.

What it is: an independent FP64 restatement (numpy/torch, written separately from the C++ node
models in ungar_amd/csrc/models/nodes.hpp) of the per-shooting-node functions the reference inlines
into its whole-horizon tapes, with Jacobians obtained from torch.autograd (an AD implementation that
shares nothing with the product's tape engine):

  quadrotor_node   /root/reference/example/mpc/quadrotor.example.cpp:126-190
  rc_car_node      /root/reference/example/mpc/rc_car.example.cpp:131-185
  srbd_node        /root/reference/example/mpc/quadruped.example.cpp:148-203
  helpers          /root/reference/include/ungar/utils/utils.hpp:731-749 (ApproximateNorm / -ExponentialMap),
                   Eigen 3.4.0 quaternion product and quaternion-times-vector formulas
  aba              /root/reference/include/ungar/rbd/quantities/generalized_accelerations.hpp:42-43
                   (= pinocchio::aba, Pinocchio v2.7.0 -- NOT in /root/reference, fetched by
                   external/config/pinocchio/CMakeLists.txt.in:15; restated from the published
                   Featherstone ABA in Pinocchio's conventions)
  anymal_node      ABA + the Lie-group semi-implicit Euler step of quadruped.example.cpp:197-200
                   (the full-body node function is defined by this project)

PARITY STATUS: **derivative half parity unpinned** against the real reference -- CppAD,
CppADCodeGen and Pinocchio are third-party, absent from /root/reference and not installable
(SURVEY.md §8(c)).  What pins this oracle instead (tests/test_oracle.py):
  * the reference's closed-form known answers (test/autodiff/function.test.cpp:70-89, 120-131);
  * ApproximateExponentialMap vs the exact exponential map (function.test.cpp:40-58);
  * second-order finite differences, the reference's own self-check (function.hpp:285-325);
  * for ABA: M(q) ddq + h(q, v) = tau with M, h from an independent RNEA written here, free-fall
    and total-mass checks, model dimensions nq=19 / nv=18 (test/rbd/robot.test.cpp:103-106).
The layout half IS pinned: tests/golden/layout_*.txt come from the reference's own headers
(oracle/ref_layout/build_ref.sh).
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field

import numpy as np
import torch

torch.set_default_dtype(torch.float64)
EPS = float(np.finfo(np.float64).eps)

DIMS = {  # name: (nx, nu, nw, np)
    "quadrotor": (13, 4, 0, 20),
    "rc_car": (6, 2, 0, 15),
    "srbd": (13, 24, 4, 6),
    "anymal": (37, 12, 0, 1),
    "srbd_ineq": (13, 24, 4, 14),  # 12 outputs per knot (inequality rows), not a dynamics node
    "quadrotor_ineq": (13, 4, 0, 1),  # 8 outputs: rotor-speed bounds
    "rc_car_ineq": (6, 2, 0, 0),  # 3 outputs: input bounds, minimum forward velocity
    "srbd_feet": (13, 24, 0, 0),  # 12 outputs: world foot positions
}


# ----------------------------------------------------------------------------- helpers
def cross(a, b):
    return torch.stack((a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]))


def quat_rotate(q, v):
    """Eigen QuaternionBase::_transformVector: v + w*t + u x t, t = 2 (u x v); q = (x, y, z, w)."""
    u = q[:3]
    t = 2.0 * cross(u, v)
    return v + q[3] * t + cross(u, t)


def quat_mul(a, b):
    """Eigen quaternion product, xyzw storage."""
    ax, ay, az, aw = a[0], a[1], a[2], a[3]
    bx, by, bz, bw = b[0], b[1], b[2], b[3]
    return torch.stack((aw * bx + ax * bw + ay * bz - az * by,
                        aw * by + ay * bw + az * bx - ax * bz,
                        aw * bz + az * bw + ax * by - ay * bx,
                        aw * bw - ax * bx - ay * by - az * bz))


def approximate_norm(v):
    """utils.hpp:731-736."""
    return torch.sqrt((v * v).sum() + EPS)


def approximate_exponential_map(v):
    """utils.hpp:738-749."""
    n = approximate_norm(v)
    return torch.cat((v * torch.sin(0.5 * n) / n, torch.cos(0.5 * n).reshape(1)))


def exact_exponential_map(v):
    """utils.hpp:700-729 (the branchy exact map; only used to pin the approximate one)."""
    n = float(np.linalg.norm(v))
    if n == 0.0:
        return np.array([0.0, 0.0, 0.0, 1.0])
    return np.concatenate((np.asarray(v) * math.sin(0.5 * n) / n, [math.cos(0.5 * n)]))


# ----------------------------------------------------------------------------- node functions
def quadrotor_node(x, u, w, p):
    dt, m, moi = p[0], p[1], p[2:5]
    g0, b, d = p[17], p[18], p[19]
    pos, q, pdot, om = x[0:3], x[3:7], x[7:10], x[10:13]
    ez = torch.tensor([0.0, 0.0, 1.0])
    sum_f = torch.zeros(3)
    sum_m = torch.zeros(3)
    sum_d = torch.zeros(3)
    for i in range(4):
        thrust = b * u[i] ** 2 * ez
        sum_f = sum_f + thrust
        sum_m = sum_m + cross(p[5 + 3 * i:8 + 3 * i], thrust)
        sum_d = sum_d + d * u[i] ** 2 * ez * (-1.0) ** i
    pdotdot = (quat_rotate(q, sum_f) - m * g0 * ez) / m
    omdot = (1.0 / moi) * (sum_m + sum_d - cross(om, moi * om))
    pdot_n = pdot + dt * pdotdot
    om_n = om + dt * omdot
    pos_n = pos + dt * pdot_n
    q_n = quat_mul(q, approximate_exponential_map(dt * om_n))
    return torch.cat((pos_n, q_n, pdot_n, om_n))


def rc_car_node(x, u, w, p):
    dt, m, moi, lf, lr = p[0], p[1], p[2], p[3], p[4]
    Bf, Cf, Df, Br, Cr, Dr = p[5], p[6], p[7], p[8], p[9], p[10]
    Cm1, Cm2, Cr0, Cr2 = p[11], p[12], p[13], p[14]
    px, py, phi, vx, vy, om = x[0], x[1], x[2], x[3], x[4], x[5]
    d, delta = u[0], u[1]
    alphaf = -torch.atan((om * lf + vy) / (vx + EPS)) + delta
    alphar = torch.atan((om * lr - vy) / (vx + EPS))
    Ffy = Df * torch.sin(Cf * torch.atan(Bf * alphaf))
    Fry = Dr * torch.sin(Cr * torch.atan(Br * alphar))
    Frx = (Cm1 - Cm2 * vx) * d - Cr0 - Cr2 * vx ** 2
    vxdot = (Frx - Ffy * torch.sin(delta) + m * vy * om) / m
    vydot = (Fry + Ffy * torch.cos(delta) - m * vx * om) / m
    omdot = (Ffy * lf * torch.cos(delta) - Fry * lr) / moi
    vx_n, vy_n, om_n = vx + dt * vxdot, vy + dt * vydot, om + dt * omdot
    return torch.stack((px + dt * (vx_n * torch.cos(phi) - vy_n * torch.sin(phi)),
                        py + dt * (vx_n * torch.sin(phi) + vy_n * torch.cos(phi)),
                        phi + dt * om_n, vx_n, vy_n, om_n))


def srbd_node(x, u, w, p):
    dt, m, moi, g0 = p[0], p[1], p[2:5], p[5]
    pos, q, pdot, om = x[0:3], x[3:7], x[7:10], x[10:13]
    pdotdot = -g0 * torch.tensor([0.0, 0.0, 1.0])
    omdot = -cross(om, moi * om)
    for i in range(4):
        f, r, s = u[6 * i:6 * i + 3], u[6 * i + 3:6 * i + 6], w[i]
        pdotdot = pdotdot + s * f / m
        omdot = omdot + s * cross(r, quat_rotate(q, f))
    omdot = omdot / moi
    pdot_n = pdot + dt * pdotdot
    om_n = om + dt * omdot
    pos_n = pos + dt * pdot_n
    q_n = quat_mul(q, approximate_exponential_map(dt * om_n))
    return torch.cat((pos_n, q_n, pdot_n, om_n))


def quadruped_whole_horizon(z, par, N=30):
    """The reference's quadruped OCP AS WRITTEN, whole horizon (example/mpc/quadruped.example.cpp:209-304): objective value and the 883 equality rows
    [x_0 - x_m; x_(k+1) - f(x_k, u_k); foot-contact rows] as torch expressions of the decision variables z = (X, U) (1123) and the parameters par = (P, Rho) (948).
    Variable layout :57-139: x = (position 3, orientation xyzw 4, linear velocity 3, body angular velocity 3), u = 4 x (ground reaction force 3, body-frame foot position 3),
    p_k = (reference state 13, 4 x (reference contact state 1, body-frame reference foot position 3)), Rho = (step size, mass, moi diagonal 3, 4 x hip position 3, leg length,
    gravity, friction coefficient, measured state 13, 4 x (measured contact state 1, measured foot position 3))."""
    nx, nu, npk = 13, 24, 29
    X = z[:(N + 1) * nx].reshape(N + 1, nx)
    U = z[(N + 1) * nx:].reshape(N, nu)
    P = par[:(N + 1) * npk].reshape(N + 1, npk)
    rho = par[(N + 1) * npk:]
    dt, mass, moi = rho[0], rho[1], rho[2:5]
    g0 = rho[18]
    xm = rho[20:33]
    measured = rho[33:49].reshape(4, 4)  # (contact state, foot position) per leg
    node_p = torch.cat((dt.reshape(1), mass.reshape(1), moi, g0.reshape(1)))
    weights = torch.tensor([0.1, 0.1, 10.0], dtype=torch.float64)
    value = torch.zeros((), dtype=torch.float64)
    for k in range(N + 1):  # :215-245
        ref = P[k, :13]
        q, qr = X[k, 3:7], ref[3:7]
        value = value + ((weights * (X[k, 0:3] - ref[0:3])) ** 2).sum() + torch.minimum(((q - qr) ** 2).sum(), ((q + qr) ** 2).sum()) \
            + ((X[k, 7:10] - ref[7:10]) ** 2).sum() + ((X[k, 10:13] - ref[10:13]) ** 2).sum()
        if k != N:
            for i in range(4):
                f, r = U[k, 6 * i:6 * i + 3], U[k, 6 * i + 3:6 * i + 6]
                r_ref = P[k, 13 + 4 * i + 1:13 + 4 * i + 4]
                value = value + ((r - r_ref) ** 2).sum() + 1e-8 * (f ** 2).sum()
    rows = [X[0] - xm]  # :262-265
    for k in range(N):  # :268-276
        contact = torch.stack([P[k, 13 + 4 * i] for i in range(4)])
        rows.append(X[k + 1] - srbd_node(X[k], U[k], contact, node_p))
    for k in range(N):  # :278-304
        for i in range(4):
            s = P[k, 13 + 4 * i]
            s_prev = P[k - 1, 13 + 4 * i] if k else measured[i, 0]
            foot = X[k, 0:3] + quat_rotate(X[k, 3:7], U[k, 6 * i + 3:6 * i + 6])
            foot_prev = X[k - 1, 0:3] + quat_rotate(X[k - 1, 3:7], U[k - 1, 6 * i + 3:6 * i + 6]) if k else measured[i, 1:4]
            rows.append(((1.0 - s_prev) * s * foot[2]).reshape(1))
            rows.append(s_prev * s * (foot - foot_prev))
    return value, torch.cat(rows)


def quadruped_whole_horizon_inequalities(z, par, N=30):
    """The 360 inequality rows h <= 0 of the reference's quadruped OCP, whole horizon (example/mpc/quadruped.example.cpp:312-338): per knot k < N and leg i
    [-s f_z, s |f_xy|~ - mu f_z, s |r - hip_i|~ - leg_length] with s the REFERENCE contact state of (k, i) and |.|~ Utils::ApproximateNorm.  Same layouts as
    quadruped_whole_horizon."""
    nx, nu, npk = 13, 24, 29
    U = z[(N + 1) * nx:].reshape(N, nu)
    P = par[:(N + 1) * npk].reshape(N + 1, npk)
    rho = par[(N + 1) * npk:]
    hips, leg_length, mu = rho[5:17].reshape(4, 3), rho[17], rho[19]
    rows = []
    for k in range(N):
        for i in range(4):
            s = P[k, 13 + 4 * i]
            f, r = U[k, 6 * i:6 * i + 3], U[k, 6 * i + 3:6 * i + 6]
            rows += [-s * f[2], s * approximate_norm(f[:2]) - mu * f[2], s * approximate_norm(r - hips[i]) - leg_length]
    return torch.stack(rows)


def rc_car_whole_horizon(z, par, N=30):
    """The reference's RC-car OCP AS WRITTEN, whole horizon (example/mpc/rc_car.example.cpp:191-285): objective value, the 186 equality rows
    [x_0 - x_m; x_(k+1) - f(x_k, u_k)] and the 90 inequality rows [|d_k| - 15, |delta_k| - 15, 0.3 - v_x,k] as torch expressions of z = (X, U) (246) and
    par = (15 car parameters, 31 reference positions, measured state) (83).  Layout :52-122: x = (position 2, yaw, body linear velocity 2, yaw rate), u = (duty cycle,
    steering angle)."""
    nx, nu = 6, 2
    X = z[:(N + 1) * nx].reshape(N + 1, nx)
    U = z[(N + 1) * nx:].reshape(N, nu)
    car = par[:15]
    ref = par[15:15 + 2 * (N + 1)].reshape(N + 1, 2)
    xm = par[15 + 2 * (N + 1):15 + 2 * (N + 1) + nx]
    value = torch.zeros((), dtype=torch.float64)
    for k in range(N):  # :204-222
        value = value + ((X[k, 0:2] - ref[k]) ** 2).sum() + 1e-6 * (U[k] ** 2).sum()
        if k:
            value = value + 1e-6 * ((U[k] - U[k - 1]) ** 2).sum()
    value = value + ((X[N, 0:2] - ref[N]) ** 2).sum()  # :224-226
    eq = [X[0] - xm] + [X[k + 1] - rc_car_node(X[k], U[k], None, car) for k in range(N)]  # :243-263
    ineq = []
    for k in range(N):  # :271-282
        ineq += [torch.abs(U[k, 0]) - 15.0, torch.abs(U[k, 1]) - 15.0, 0.3 - X[k, 3]]
    return value, torch.cat(eq), torch.stack(ineq)


# ----------------------------------------------------------------------------- rigid-body model
def _rpy(r, p, y):
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def _skew(v):
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


def _spatial_inertia(mass, com, Ic):
    """6x6, (linear, angular) ordering, about the frame origin."""
    cx = _skew(com)
    Y = np.zeros((6, 6))
    Y[:3, :3] = mass * np.eye(3)
    Y[:3, 3:] = -mass * cx
    Y[3:, :3] = mass * cx
    Y[3:, 3:] = Ic - mass * cx @ cx
    return Y


def _force_xform(R, p):
    X = np.zeros((6, 6))
    X[:3, :3] = R
    X[3:, 3:] = R
    X[3:, :3] = _skew(p) @ R
    return X


@dataclass
class OJoint:
    name: str
    parent: int
    R: np.ndarray
    p: np.ndarray
    axis: np.ndarray | None  # None => free flyer
    Y: np.ndarray
    iq: int = 0
    iv: int = 0


@dataclass
class OModel:
    joints: list = field(default_factory=list)
    nq: int = 0
    nv: int = 0
    gravity: np.ndarray = field(default_factory=lambda: np.array([0.0, 0.0, -9.81]))


def load_robot(path: str) -> OModel:
    """Reads the flat '.robot' text (tools/urdf_to_robot.py) and builds the free-flyer model with
    fixed joints lumped, children visited depth-first in joint-name order (Pinocchio/urdfdom)."""
    links, joints = {}, {}
    with open(path) as f:
        for line in f:
            t = line.split()
            if not t or t[0].startswith("#"):
                continue
            if t[0] == "link":
                v = [float(s) for s in t[3:]]
                links[t[1]] = dict(has=int(t[2]), m=v[0], c=np.array(v[1:4]), rpy=v[4:7],
                                   I=np.array([[v[7], v[8], v[9]], [v[8], v[10], v[11]], [v[9], v[11], v[12]]]))
            elif t[0] == "joint":
                v = [float(s) for s in t[5:]]
                joints[t[1]] = dict(type=t[2], parent=t[3], child=t[4], xyz=np.array(v[0:3]), rpy=v[3:6], axis=np.array(v[6:9]))
    children = {j["child"] for j in joints.values()}
    (root,) = [l for l in links if l not in children]

    def link_Y(name):
        L = links[name]
        if not L["has"]:
            return np.zeros((6, 6))
        R = _rpy(*L["rpy"])
        return _spatial_inertia(L["m"], L["c"], R @ L["I"] @ R.T)

    model = OModel()
    model.joints.append(OJoint("universe", 0, np.eye(3), np.zeros(3), None, np.zeros((6, 6))))
    model.joints.append(OJoint("root_joint", 0, np.eye(3), np.zeros(3), None, link_Y(root)))
    model.frames = [(root, 1, np.eye(3), np.zeros(3))]  # (link name, supporting joint, R, p in that joint's frame)

    def visit(link, support, R, p):
        for name in sorted(joints):  # ASCII order, as std::map<std::string,...>
            j = joints[name]
            if j["parent"] != link:
                continue
            Rj = R @ _rpy(*j["rpy"])
            pj = p + R @ j["xyz"]
            if j["type"] == "fixed":
                X = _force_xform(Rj, pj)
                model.joints[support].Y = model.joints[support].Y + X @ link_Y(j["child"]) @ X.T
                model.frames.append((j["child"], support, Rj, pj))
                visit(j["child"], support, Rj, pj)
            else:
                model.joints.append(OJoint(name, support, Rj, pj, j["axis"], link_Y(j["child"])))
                model.frames.append((j["child"], len(model.joints) - 1, np.eye(3), np.zeros(3)))
                visit(j["child"], len(model.joints) - 1, np.eye(3), np.zeros(3))

    visit(root, 1, np.eye(3), np.zeros(3))
    iq = iv = 0
    for k, j in enumerate(model.joints):
        j.iq, j.iv = iq, iv
        if k == 0:
            continue
        iq += 7 if j.axis is None else 1
        iv += 6 if j.axis is None else 1
    model.nq, model.nv = iq, iv
    return model


_ANYMAL = None


def anymal_model() -> OModel:
    global _ANYMAL
    if _ANYMAL is None:
        here = os.path.dirname(os.path.abspath(__file__))
        _ANYMAL = load_robot(os.path.join(here, "..", "ungar_amd", "data", "anymal_b.robot"))
    return _ANYMAL


def _quat_to_rot(q):
    """Eigen toRotationMatrix, q = (x, y, z, w), no normalisation."""
    x, y, z, w = q[0], q[1], q[2], q[3]
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return torch.stack((torch.stack((1 - (tyy + tzz), txy - twz, txz + twy)),
                        torch.stack((txy + twz, 1 - (txx + tzz), tyz - twx)),
                        torch.stack((txz - twy, tyz + twx, 1 - (txx + tyy)))))


def _axis_rot(axis, angle):
    a = torch.as_tensor(axis)
    K = torch.as_tensor(_skew(axis))
    return torch.eye(3) + torch.sin(angle) * K + (1 - torch.cos(angle)) * (K @ K) + 0.0 * a.sum()


def _kinematics(model, q, v):
    """liMi (R, p) and joint-local spatial velocity contributions for every joint."""
    Rs, ps, vjs, Ss = [None], [None], [None], [None]
    for j in model.joints[1:]:
        if j.axis is None:
            Rj = _quat_to_rot(q[j.iq + 3:j.iq + 7])
            pj = q[j.iq:j.iq + 3]
            vj = v[j.iv:j.iv + 6]
            S = torch.eye(6)
        else:
            Rj = _axis_rot(j.axis, q[j.iq])
            pj = torch.zeros(3)
            S = torch.cat((torch.zeros(3), torch.as_tensor(j.axis))).reshape(6, 1)
            vj = S[:, 0] * v[j.iv]
        Rs.append(torch.as_tensor(j.R) @ Rj)
        ps.append(torch.as_tensor(j.p) + torch.as_tensor(j.R) @ pj)
        vjs.append(vj)
        Ss.append(S)
    return Rs, ps, vjs, Ss


def _act_inv_motion(R, p, m):
    return torch.cat((R.T @ (m[:3] - cross(p, m[3:])), R.T @ m[3:]))


def _act_force(R, p, f):
    l = R @ f[:3]
    return torch.cat((l, R @ f[3:] + cross(p, l)))


def _cross_motion(v, m):
    return torch.cat((cross(v[3:], m[:3]) + cross(v[:3], m[3:]), cross(v[3:], m[3:])))


def _cross_force(v, f):
    return torch.cat((cross(v[3:], f[:3]), cross(v[3:], f[3:]) + cross(v[:3], f[:3])))


def _force_xform_t(R, p):
    px = torch.stack((torch.stack((torch.zeros(()), -p[2], p[1])),
                      torch.stack((p[2], torch.zeros(()), -p[0])),
                      torch.stack((-p[1], p[0], torch.zeros(())))))
    top = torch.cat((R, torch.zeros(3, 3)), dim=1)
    bot = torch.cat((px @ R, R), dim=1)
    return torch.cat((top, bot), dim=0)


def frame_placements(model: OModel, q):
    """World placement (R, p) of every frame of the model for configuration q (forward kinematics)."""
    Rs, ps, _, _ = _kinematics(model, q, torch.zeros(model.nv, dtype=torch.float64))
    oR, op = [None] * len(model.joints), [None] * len(model.joints)
    for i in range(1, len(model.joints)):
        par = model.joints[i].parent
        if par == 0:
            oR[i], op[i] = Rs[i], ps[i]
        else:
            oR[i], op[i] = oR[par] @ Rs[i], op[par] + oR[par] @ ps[i]
    out = {}
    for name, joint, R, p in model.frames:
        out[name] = (oR[joint] @ torch.as_tensor(R), op[joint] + oR[joint] @ torch.as_tensor(p))
    return out


def centroidal_momentum(model: OModel, q, v):
    """[linear; angular] momentum about the centre of mass, world axes, by spatial algebra: every body's
    momentum Y_i v_i is carried to the world origin with the force transform of its world placement, summed,
    and shifted to the centre of mass (angular_G = angular_O - com x linear)."""
    Rs, ps, vjs, _ = _kinematics(model, q, v)
    n = len(model.joints)
    oR, op, vel = [None] * n, [None] * n, [torch.zeros(6, dtype=torch.float64)] + [None] * (n - 1)
    h = torch.zeros(6, dtype=torch.float64)
    mass, first = 0.0, torch.zeros(3, dtype=torch.float64)
    for i in range(1, n):
        j = model.joints[i]
        vel[i] = vjs[i] + (_act_inv_motion(Rs[i], ps[i], vel[j.parent]) if j.parent > 0 else 0.0)
        oR[i], op[i] = (Rs[i], ps[i]) if j.parent == 0 else (oR[j.parent] @ Rs[i], op[j.parent] + oR[j.parent] @ ps[i])
        Y = torch.as_tensor(j.Y)
        h = h + _act_force(oR[i], op[i], Y @ vel[i])
        m_i = Y[0, 0]
        c_i = torch.stack((Y[5, 1], Y[3, 2], Y[4, 0])) / m_i if m_i > 0 else torch.zeros(3, dtype=torch.float64)  # m c from the m c^ block
        mass = mass + m_i
        first = first + m_i * (op[i] + oR[i] @ c_i)
    com = first / mass
    return torch.cat((h[:3], h[3:] - torch.linalg.cross(com, h[:3])))


def composite_inertia_about_com(model: OModel, q):
    """Rotational inertia (3 x 3, world axes) of the robot frozen at q about its centre of mass: every body's
    spatial inertia is carried to the world origin (X Y X^T with the force transform of its world placement),
    summed, and the parallel-axis term of the total mass at the centre of mass is removed."""
    Rs, ps, _, _ = _kinematics(model, q, torch.zeros(model.nv, dtype=torch.float64))
    n = len(model.joints)
    oR, op = [None] * n, [None] * n
    total = torch.zeros((6, 6), dtype=torch.float64)
    for i in range(1, n):
        par = model.joints[i].parent
        oR[i], op[i] = (Rs[i], ps[i]) if par == 0 else (oR[par] @ Rs[i], op[par] + oR[par] @ ps[i])
        X = torch.as_tensor(_force_xform(oR[i].numpy(), op[i].numpy()))
        total = total + X @ torch.as_tensor(model.joints[i].Y) @ X.T
    mass = total[0, 0]
    com = torch.stack((total[5, 1], total[3, 2], total[4, 0])) / mass
    return total[3:, 3:] - mass * (com.dot(com) * torch.eye(3, dtype=torch.float64) - torch.outer(com, com))


def aba(model: OModel, q, v, tau):
    """Featherstone's articulated-body algorithm in Pinocchio's formulation (three passes)."""
    n = len(model.joints)
    Rs, ps, vjs, Ss = _kinematics(model, q, v)
    vel, acc, f, Y = [torch.zeros(6)] + [None] * (n - 1), [None] * n, [None] * n, [None] * n
    for i in range(1, n):
        j = model.joints[i]
        vel[i] = vjs[i] + (_act_inv_motion(Rs[i], ps[i], vel[j.parent]) if j.parent > 0 else 0.0)
        acc[i] = _cross_motion(vel[i], vjs[i])
        Y[i] = torch.as_tensor(j.Y)
        f[i] = _cross_force(vel[i], Y[i] @ vel[i])
    u = [None] * n
    U, Dinv, UDinv = [None] * n, [None] * n, [None] * n
    for i in range(n - 1, 0, -1):
        j = model.joints[i]
        nv = 6 if j.axis is None else 1
        S = Ss[i]
        u[i] = tau[j.iv:j.iv + nv] - S.T @ f[i]
        U[i] = Y[i] @ S
        D = S.T @ U[i]
        Dinv[i] = torch.linalg.inv(D)
        UDinv[i] = U[i] @ Dinv[i]
        if j.parent > 0:
            Ia = Y[i] - UDinv[i] @ U[i].T
            pa = f[i] + Ia @ acc[i] + UDinv[i] @ u[i]
            X = _force_xform_t(Rs[i], ps[i])
            Y[j.parent] = Y[j.parent] + X @ Ia @ X.T
            f[j.parent] = f[j.parent] + _act_force(Rs[i], ps[i], pa)
    acc[0] = torch.cat((-torch.as_tensor(model.gravity), torch.zeros(3)))
    ddq = [None] * n
    for i in range(1, n):
        j = model.joints[i]
        acc[i] = acc[i] + _act_inv_motion(Rs[i], ps[i], acc[j.parent])
        ddq[i] = Dinv[i] @ u[i] - UDinv[i].T @ acc[i]
        acc[i] = acc[i] + Ss[i] @ ddq[i]
    return torch.cat(ddq[1:])


def rnea(model: OModel, q, v, a, gravity=True):
    """Recursive Newton-Euler inverse dynamics (independent cross-check of aba)."""
    n = len(model.joints)
    Rs, ps, vjs, Ss = _kinematics(model, q, v)
    g = torch.as_tensor(model.gravity) if gravity else torch.zeros(3)
    vel, acc, f = [torch.zeros(6)] + [None] * (n - 1), [torch.cat((-g, torch.zeros(3)))] + [None] * (n - 1), [None] * n
    for i in range(1, n):
        j = model.joints[i]
        nv = 6 if j.axis is None else 1
        vel[i] = vjs[i] + (_act_inv_motion(Rs[i], ps[i], vel[j.parent]) if j.parent > 0 else 0.0)
        acc[i] = _act_inv_motion(Rs[i], ps[i], acc[j.parent]) + Ss[i] @ a[j.iv:j.iv + nv] + _cross_motion(vel[i], vjs[i])
        Yi = torch.as_tensor(j.Y)
        f[i] = Yi @ acc[i] + _cross_force(vel[i], Yi @ vel[i])
    tau = [None] * n
    for i in range(n - 1, 0, -1):
        j = model.joints[i]
        tau[i] = Ss[i].T @ f[i]
        if j.parent > 0:
            f[j.parent] = f[j.parent] + _act_force(Rs[i], ps[i], f[i])
    return torch.cat(tau[1:])


# ----------------------------------------------------------------------------- rigid-body quantities as node functions
RBD_DIMS = {  # name: (nx, nu, ny) of the batched quantity models of SURVEY.md section 8(f) N4
    "anymal_rnea": (37, 18, 18), "anymal_crba": (19, 0, 324), "anymal_minv": (19, 0, 324), "anymal_feet": (19, 0, 48), "anymal_centroidal": (37, 0, 6),
}
FOOT_FRAMES = ("LF_FOOT", "LH_FOOT", "RF_FOOT", "RH_FOOT")  # test/rbd/robot.test.cpp:49-52


def inertia_matrix(model: OModel, q):
    """M(q) column by column from inverse dynamics: M e_j = RNEA(q, 0, e_j) without gravity (an identity of the
    equations of motion, not the composite-rigid-body recursion the product uses)."""
    zero = torch.zeros(model.nv, dtype=torch.float64)
    eye = torch.eye(model.nv, dtype=torch.float64)
    return torch.stack([rnea(model, q, zero, eye[j], gravity=False) for j in range(model.nv)], dim=1)


def rbd_quantity(name: str, x, u=None):
    """y = f(x, u) of the quantity model `name` for ONE configuration (torch vectors in, torch vector out)."""
    model = anymal_model()
    if name == "anymal_rnea":  # rbd/quantities/joint_torques.hpp:42-43
        return rnea(model, x[:19], x[19:], u)
    if name == "anymal_crba":  # joint_space_inertia_matrix.hpp:42-43
        return inertia_matrix(model, x).reshape(-1)
    if name == "anymal_minv":  # joint_space_inertia_matrix_inverse.hpp:42-43
        return torch.linalg.inv(inertia_matrix(model, x)).reshape(-1)
    if name == "anymal_feet":  # frames.hpp:42-43
        placements = frame_placements(model, x)
        return torch.cat([torch.cat((placements[f][1], placements[f][0].reshape(-1))) for f in FOOT_FRAMES])
    if name == "anymal_centroidal":  # centroidal_momentum.hpp:42-43
        return centroidal_momentum(model, x[:19], x[19:])
    raise KeyError(name)


def rbd_quantity_jacobian(name: str, x, u=None):
    """(y, d y / d (x, u)) with torch.autograd."""
    nx = RBD_DIMS[name][0]
    z = torch.cat((x, u)) if u is not None else x
    g = lambda zz: rbd_quantity(name, zz[:nx], zz[nx:] if u is not None else None)  # noqa: E731
    J = torch.autograd.functional.jacobian(g, z, vectorize=False)
    with torch.no_grad():
        return g(z), J


def synthetic_rbd_inputs(name: str, count: int, seed: int = 0):
    """Seeded (x, u) rows for the quantity models: q = [p, unit quaternion, joints], v, a ~ U(-1, 1)."""
    rng = np.random.default_rng(0x5EED0000 + seed)
    nx, nu, _ = RBD_DIMS[name]
    quat = rng.normal(size=(count, 4))
    quat /= np.linalg.norm(quat, axis=1, keepdims=True)
    q = np.concatenate((rng.uniform(-1, 1, (count, 3)), quat, rng.uniform(-1, 1, (count, 12))), axis=1)
    x = q if nx == 19 else np.concatenate((q, rng.uniform(-1, 1, (count, 18))), axis=1)
    return x, rng.uniform(-1, 1, (count, nu))


def anymal_node(x, u, w, p):
    model = anymal_model()
    nq, nv = model.nq, model.nv
    dt = p[0]
    q, v = x[:nq], x[nq:nq + nv]
    tau = torch.cat((torch.zeros(6), u))
    a = aba(model, q, v, tau)
    v_n = v + dt * a
    quat = q[3:7]
    pos_n = q[0:3] + dt * quat_rotate(quat, v_n[0:3])
    quat_n = quat_mul(quat, approximate_exponential_map(dt * v_n[3:6]))
    qj_n = q[7:] + dt * v_n[6:]
    return torch.cat((pos_n, quat_n, qj_n, v_n))


def srbd_ineq_node(x, u, w, p):
    """Inequality rows h <= 0 of one knot of the quadruped OCP (quadruped.example.cpp:321-335): per leg
    -s f_z, s |f_xy|~ - mu f_z, s |r - hip|~ - leg_length; p = [mu, 4 x hip(3), leg_length]."""
    rows = []
    for i in range(4):
        f, r = u[6 * i:6 * i + 3], u[6 * i + 3:6 * i + 6]
        hip = p[1 + 3 * i:4 + 3 * i]
        s = w[i]
        rows += [-s * f[2], s * approximate_norm(f[:2]) - p[0] * f[2], s * approximate_norm(r - hip) - p[13]]
    return torch.stack(rows)


def quadrotor_ineq_node(x, u, w, p):
    """quadrotor.example.cpp:280-288: per rotor [r - r_max, -r]; p = [max_rotor_speed]."""
    return torch.stack([t for i in range(4) for t in (u[i] - p[0], -u[i])])


def rc_car_ineq_node(x, u, w, p):
    """rc_car.example.cpp:271-282: [|d| - 15, |delta| - 15, 0.3 - v_x]."""
    return torch.stack((torch.abs(u[0]) - 15.0, torch.abs(u[1]) - 15.0, 0.3 - x[3]))


def srbd_feet_node(x, u, w, p):
    """quadruped.example.cpp:288-291: pFoot_i = p + q * r_i (the node-local part of the foot-contact equality rows)."""
    return torch.cat([x[0:3] + quat_rotate(x[3:7], u[6 * i + 3:6 * i + 6]) for i in range(4)])


NODES = {"quadrotor": quadrotor_node, "rc_car": rc_car_node, "srbd": srbd_node, "anymal": anymal_node, "srbd_ineq": srbd_ineq_node,
         "quadrotor_ineq": quadrotor_ineq_node, "rc_car_ineq": rc_car_ineq_node, "srbd_feet": srbd_feet_node}


# ----------------------------------------------------------------------------- evaluation API
# ----------------------------------------------------------------------------- scalar stage cost
def quadrotor_cost(x, u, p):
    """Per-knot stage cost of example/mpc/quadrotor.example.cpp:196-236 (input-rate term excluded: it couples
    consecutive knots): p = [p_ref(3), q_ref(4), v_ref(3), omega_ref(3)]."""
    track = ((x[0:3] - p[0:3]) ** 2).sum()
    track = track + torch.minimum(((x[3:7] - p[3:7]) ** 2).sum(), ((x[3:7] + p[3:7]) ** 2).sum())
    track = track + ((x[7:13] - p[7:13]) ** 2).sum()
    return track + 1e-6 * (u ** 2).sum()


def srbd_cost(x, u, p):
    """Per-knot stage cost of example/mpc/quadruped.example.cpp:215-245:
    p = [p_ref(3), q_ref(4), v_ref(3), omega_ref(3), 4 x b_reference_foot_position(3)], u = 4 x [f(3), r(3)]."""
    weight = torch.tensor([0.1, 0.1, 10.0], dtype=torch.float64)
    value = ((weight * (x[0:3] - p[0:3])) ** 2).sum()
    value = value + torch.minimum(((x[3:7] - p[3:7]) ** 2).sum(), ((x[3:7] + p[3:7]) ** 2).sum())
    value = value + ((x[7:13] - p[7:13]) ** 2).sum()
    for leg in range(4):
        f, r = u[6 * leg:6 * leg + 3], u[6 * leg + 3:6 * leg + 6]
        value = value + ((r - p[13 + 3 * leg:16 + 3 * leg]) ** 2).sum() + 1e-8 * (f ** 2).sum()
    return value


def rc_car_cost(x, u, p):
    """Per-knot stage cost of example/mpc/rc_car.example.cpp:204-222 (input-variation term excluded): p = [reference_position(2)]."""
    return ((x[0:2] - p[0:2]) ** 2).sum() + 1e-6 * (u ** 2).sum()


def anymal_cost(x, u, p):
    """Stage cost of the full-body quadruped (ungar_amd's own: the reference has no full-body OCP; restated here from its definition in
    DESIGN.md section 4.9, not from the product's source): p = [x_ref(37), w_position, w_orientation, w_joints, w_velocity, w_torque],
    orientation through the sign-invariant min(|q - q_ref|^2, |q + q_ref|^2) of quadrotor.example.cpp:206-212."""
    ref, (wp, wq, wj, wv, wt) = p[:37], p[37:42]
    quat = torch.minimum(((x[3:7] - ref[3:7]) ** 2).sum(), ((x[3:7] + ref[3:7]) ** 2).sum())
    return (wp * ((x[0:3] - ref[0:3]) ** 2).sum() + wq * quat + wj * ((x[7:19] - ref[7:19]) ** 2).sum() + wv * ((x[19:37] - ref[19:37]) ** 2).sum() +
            wt * (u ** 2).sum())


COSTS = {"quadrotor_cost": (quadrotor_cost, 13, 4, 13), "srbd_cost": (srbd_cost, 13, 24, 25), "rc_car_cost": (rc_car_cost, 6, 2, 2),
         "anymal_cost": (anymal_cost, 37, 12, 42)}  # fn, nx, nu, np


def cost_value_gradient_hessian(x, u, p, name="quadrotor_cost"):
    """(y, gradient, dense Hessian) w.r.t. z = (x, u) for a batch of numpy inputs."""
    cost, nx = COSTS[name][0], COSTS[name][1]
    X, U, P = (torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64) for a in (x, u, p))
    ys, gs, hs = [], [], []
    for b in range(X.shape[0]):
        z = torch.cat((X[b], U[b])).requires_grad_(True)
        fn = lambda zz: cost(zz[:nx], zz[nx:], P[b])  # noqa: E731
        y = fn(z)
        (g,) = torch.autograd.grad(y, z, create_graph=False)
        ys.append(y.detach())
        gs.append(g)
        hs.append(torch.autograd.functional.hessian(fn, z.detach()))
    return torch.stack(ys).numpy(), torch.stack(gs).numpy(), torch.stack(hs).numpy()


def synthetic_cost_inputs(count: int, seed: int = 0, name: str = "quadrotor_cost"):
    """States/inputs as for the dynamics node, references = perturbed states (some with flipped quaternion
    sign, so that both branches of the min are exercised); srbd_cost: plus perturbed footholds."""
    x, u, _, _ = synthetic_inputs({"quadrotor_cost": "quadrotor", "srbd_cost": "srbd", "rc_car_cost": "rc_car", "anymal_cost": "anymal"}[name], count, seed)
    rng = np.random.default_rng(0xC057 + seed)
    ref = x + rng.normal(scale=0.3, size=x.shape)
    if name == "rc_car_cost":
        return x, u, ref[:, :2].copy()
    if name == "anymal_cost":
        ref[:, 3:7] /= np.linalg.norm(ref[:, 3:7], axis=1, keepdims=True)
        ref[::2, 3:7] *= -1.0
        weights = rng.uniform(0.1, 10.0, size=(count, 5)) * np.array([1.0, 1.0, 1.0, 0.1, 1e-3])
        return x, u, np.concatenate((ref, weights), axis=1)
    ref[:, 3:7] /= np.linalg.norm(ref[:, 3:7], axis=1, keepdims=True)
    ref[::2, 3:7] *= -1.0
    if name == "srbd_cost":
        feet = np.concatenate([u[:, 6 * leg + 3:6 * leg + 6] for leg in range(4)], axis=1)
        ref = np.concatenate((ref, feet + rng.normal(scale=0.05, size=feet.shape)), axis=1)
    return x, u, ref


def node_value(name: str, x, u, w, p) -> np.ndarray:
    """f for a batch: x (B,nx), u (B,nu), w (B,nw), p (B,np) numpy -> (B,nx)."""
    fn = NODES[name]
    X, U, W, P = (torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64) for a in (x, u, w, p))
    with torch.no_grad():
        return torch.stack([fn(X[b], U[b], W[b], P[b]) for b in range(X.shape[0])]).numpy()


def node_jacobian(name: str, x, u, w, p):
    """(f, J) for a batch; J is the dense (B, nx, nx+nu) block [A | B] = d f / d (x, u)."""
    fn = NODES[name]
    nx = DIMS[name][0]
    X, U, W, P = (torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64) for a in (x, u, w, p))
    fs, js = [], []
    for b in range(X.shape[0]):
        z = torch.cat((X[b], U[b]))
        g = lambda zz: fn(zz[:nx], zz[nx:], W[b], P[b])  # noqa: E731
        js.append(torch.autograd.functional.jacobian(g, z, vectorize=False))
        with torch.no_grad():
            fs.append(g(z))
    return torch.stack(fs).numpy(), torch.stack(js).numpy()


# ----------------------------------------------------------------------------- synthetic inputs
def default_params(name: str) -> np.ndarray:
    """Per-instance parameter block p with the reference's values."""
    if name == "quadrotor":  # quadrotor.example.cpp:326-343
        props = [[0.2, 0.2, 0.0], [-0.2, 0.2, 0.0], [-0.2, -0.2, 0.0], [0.2, -0.2, 0.0]]
        return np.array([1.0 / 30.0, 1.5, 3e-2, 3e-2, 3e-2] + [c for pp in props for c in pp] + [9.80665, 0.015, 0.1])
    if name == "rc_car":  # rc_car.example.cpp:320-337
        return np.array([1.0 / 30.0, 0.041, 27.8e-6, 0.029, 0.033, 2.579, 1.2, 0.192, 3.3852, 1.2691, 0.1737, 0.287, 0.0545,
                         0.0518, 0.00035])
    if name == "srbd":  # quadruped.example.cpp:378-392
        return np.array([1.0 / 30.0, 25.0, 0.048125, 0.093125, 0.055625, 9.80665])
    if name == "anymal":
        return np.array([1.0 / 20.0])
    if name == "srbd_ineq":  # quadruped.example.cpp:384-392
        return np.array([0.7, 0.2, 0.15, -0.1, 0.2, -0.15, -0.1, -0.2, 0.15, -0.1, -0.2, -0.15, -0.1, 0.42])
    if name == "quadrotor_ineq":  # max_rotor_speed: twice the hover speed (quadrotor.example.cpp:356-358)
        return np.array([2.0 * math.sqrt(1.5 * 9.80665 / (4 * 0.015))])
    if name in ("rc_car_ineq", "srbd_feet"):
        return np.zeros(0)
    raise KeyError(name)


def node_jacobian_batched(name: str, x, u, w, p):
    """The SAME node functions as node_jacobian, differentiated in forward mode and vectorised over the batch (torch.func.vmap(jacfwd)): one pass over the
    operations for all nodes instead of one reverse sweep per node and output row -- 512 ANYmal nodes in about a second instead of minutes, which is what
    lets the tests compare a spread SAMPLE of every full-size launch with this independent oracle (tests/test_tiles.py, tests/test_gpu_parity.py).  Agrees with
    node_jacobian to rounding (tests/test_oracle.py)."""
    from torch.func import jacfwd, vmap
    fn = NODES[name]
    nx = DIMS[name][0]
    X, U, W, P = (torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64) for a in (x, u, w, p))

    def g(z, wb, pb):
        return fn(z[:nx], z[nx:], wb, pb)

    Z = torch.cat((X, U), dim=1)
    with torch.no_grad():
        f = vmap(g)(Z, W, P)
    return f.numpy(), vmap(jacfwd(g))(Z, W, P).numpy()


def synthetic_inputs(name: str, count: int, seed: int = 0):
    """Deterministic random (x, u, w, p) in the ranges of SURVEY.md §8(d)."""
    rng = np.random.default_rng(0x5EED0000 + seed)
    nx, nu, nw, npar = DIMS[name]
    p = np.tile(default_params(name), (count, 1))
    w = np.zeros((count, nw))
    if name == "quadrotor":
        q = rng.normal(size=(count, 4))
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        hover = math.sqrt(p[0, 1] * p[0, 17] / (4 * p[0, 18]))
        x = np.concatenate((rng.uniform(-2, 2, (count, 3)), q, rng.uniform(-1, 1, (count, 3)), rng.uniform(-1, 1, (count, 3))), axis=1)
        u = rng.uniform(0.5, 1.5, (count, 4)) * hover
    elif name == "rc_car":
        x = np.concatenate((rng.uniform(-1, 1, (count, 2)), rng.uniform(-math.pi, math.pi, (count, 1)), rng.uniform(0.5, 2.0, (count, 1)),
                            rng.uniform(-0.3, 0.3, (count, 1)), rng.uniform(-2, 2, (count, 1))), axis=1)
        u = np.concatenate((rng.uniform(-1, 1, (count, 1)), rng.uniform(-0.3, 0.3, (count, 1))), axis=1)
    elif name == "srbd":
        q = rng.normal(size=(count, 4))
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        x = np.concatenate((rng.uniform(-2, 2, (count, 3)), q, rng.uniform(-1, 1, (count, 3)), rng.uniform(-1, 1, (count, 3))), axis=1)
        mg4 = p[0, 1] * p[0, 5] / 4
        hips = np.array([[0.2, 0.15, -0.1], [0.2, -0.15, -0.1], [-0.2, 0.15, -0.1], [-0.2, -0.15, -0.1]])
        u = np.zeros((count, 24))
        for i in range(4):
            u[:, 6 * i:6 * i + 3] = mg4 * np.concatenate((rng.uniform(-0.2, 0.2, (count, 2)), rng.uniform(0.5, 1.5, (count, 1))), axis=1)
            u[:, 6 * i + 3:6 * i + 6] = hips[i] + rng.uniform(-0.1, 0.1, (count, 3)) - np.array([0, 0, 0.3])
        w = (rng.uniform(size=(count, 4)) < 0.5).astype(np.float64)
    elif name == "anymal":
        q = rng.normal(size=(count, 4))
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        x = np.concatenate((rng.uniform(-1, 1, (count, 3)), q, rng.uniform(-1, 1, (count, 12)), rng.uniform(-1, 1, (count, 18))), axis=1)
        u = rng.uniform(-20, 20, (count, 12))
    elif name == "srbd_ineq":  # states/forces/footholds as for the srbd dynamics node; some constraints violated
        x, u, w, _ = synthetic_inputs("srbd", count, seed)
        u[::3, 2::6] *= -0.1  # pulling contact forces: unilateral and friction rows active
    elif name == "quadrotor_ineq":  # rotor speeds on both sides of [0, r_max]
        x, u, _, _ = synthetic_inputs("quadrotor", count, seed)
        u[::4] *= 2.5
        u[1::4, 0] *= -0.2
    elif name == "rc_car_ineq":  # inputs on both sides of the bounds and of zero (the kink of |.|), slow and fast cars
        x, u, _, _ = synthetic_inputs("rc_car", count, seed)
        u *= 20.0
        x[::3, 3] *= 0.1
    elif name == "srbd_feet":
        x, u, _, _ = synthetic_inputs("srbd", count, seed)
    else:
        raise KeyError(name)
    return x, u, w, p
