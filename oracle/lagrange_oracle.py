"""TEST INFRASTRUCTURE -- third, formulation-independent oracle for the rigid-body half of the hot path (SURVEY.md
section 8(a) row A7, section 8(c)).

The product computes forward dynamics with Featherstone's articulated-body algorithm and differentiates it through
RNEA / CRBA (spatial 6-D algebra); the first oracle (ungar_oracle.py) restates ABA in the same algebra and reads the same
`anymal_b.robot` file -- a common-mode risk.  This oracle shares NEITHER:

  * data: `read_urdf()` is a second, independent reader of the reference's URDF
    (data/robots/anymal_b_description/robots/anymal.urdf) that keeps all 23 links un-lumped; its output is committed
    as tests/golden/anymal_urdf_values.json (numbers only).  `read_robot_file()` reads ungar_amd/data/anymal_b.robot
    into the same structure so that the two sources can be compared with each other;
  * formulation: Lagrange's equations from energies.  Positions and rotation matrices of every link are plain
    functions of a local chart xi of the configuration manifold around (p0, R0, theta0),
        p = p0 + R0 xi_lin,   R = R0 Exp(xi_ang),   theta = theta0 + xi_j,
    the mass matrix is  M(xi) = sum_links m Jc^T Jc + Jw^T (R I R^T) Jw  with the point / rotation Jacobians obtained by
    automatic differentiation (torch.func) of those positions and rotations, the potential is  V = sum m g z_com, and
        M xi'' + (dM/dt) xi' - dT/dxi + dV/dxi = Q,      T = 1/2 xi'^T M(xi) xi'
    is solved for xi'' at xi = 0, where xi' equals the body-frame velocity v of Pinocchio's conventions
    (free-flyer: linear and angular velocity in the BASE frame) and
        d(v_lin)/dt = xi''_lin - omega x v_lin,    d(omega)/dt = xi''_ang,    d(theta')/dt = xi''_j .
    No spatial vectors, no Pluecker transforms, no recursion over the tree beyond composing homogeneous transforms.

Conventions (test/rbd/robot.test.cpp:44-52): q = [p(3), quaternion xyzw(4), 12 joint angles in the
order LF, LH, RF, RH x (HAA, HFE, KFE)], v = [base linear(3), base angular(3), 12 joint rates], gravity (0, 0, -9.81).
"""
from __future__ import annotations

import json
import math
import os
import xml.etree.ElementTree as ET

import numpy as np
import torch

GRAVITY = 9.81  # Pinocchio's default model gravity (0, 0, -9.81)
LEGS = ("LF", "LH", "RF", "RH")  # joint order of the reference's tests (robot.test.cpp:44-47)
JOINT_ORDER = tuple(f"{leg}_{j}" for leg in LEGS for j in ("HAA", "HFE", "KFE"))
HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(os.path.dirname(HERE), "tests", "golden", "anymal_urdf_values.json")
ROBOT_FILE = os.path.join(os.path.dirname(HERE), "ungar_amd", "data", "anymal_b.robot")


# ------------------------------------------------------------------------------------------------ data sources
def _floats(text, n=3, default=0.0):
    return [float(t) for t in text.split()] if text else [default] * n


def read_urdf(path: str) -> dict:
    """Second reader of the URDF: every link with its inertial block, every joint with origin / axis / type."""
    root = ET.parse(path).getroot()
    links, joints = {}, []
    for link in root.findall("link"):
        inertial = link.find("inertial")
        entry = {"mass": 0.0, "com": [0.0] * 3, "com_rpy": [0.0] * 3, "inertia": [0.0] * 6}
        if inertial is not None:
            origin = inertial.find("origin")
            if origin is not None:
                entry["com"] = _floats(origin.get("xyz"))
                entry["com_rpy"] = _floats(origin.get("rpy"))
            entry["mass"] = float(inertial.find("mass").get("value"))
            i = inertial.find("inertia")
            entry["inertia"] = [float(i.get(k)) for k in ("ixx", "ixy", "ixz", "iyy", "iyz", "izz")]
        links[link.get("name")] = entry
    for joint in root.findall("joint"):
        origin, axis = joint.find("origin"), joint.find("axis")
        joints.append({"name": joint.get("name"), "type": joint.get("type"), "parent": joint.find("parent").get("link"),
                       "child": joint.find("child").get("link"),
                       "xyz": _floats(origin.get("xyz")) if origin is not None else [0.0] * 3,
                       "rpy": _floats(origin.get("rpy")) if origin is not None else [0.0] * 3,
                       "axis": _floats(axis.get("xyz")) if axis is not None else [1.0, 0.0, 0.0]})
    return {"source": os.path.basename(path), "links": links, "joints": joints}


def read_robot_file(path: str = ROBOT_FILE) -> dict:
    """ungar_amd/data/anymal_b.robot (what the product loads) into the same structure."""
    links, joints = {}, []
    with open(path) as fh:
        for line in fh:
            t = line.split()
            if not t or t[0].startswith("#"):
                continue
            if t[0] == "link":
                v = [float(s) for s in t[3:]]
                links[t[1]] = {"mass": v[0], "com": v[1:4], "com_rpy": v[4:7], "inertia": v[7:13]}
            elif t[0] == "joint":
                v = [float(s) for s in t[5:]]
                joints.append({"name": t[1], "type": t[2], "parent": t[3], "child": t[4], "xyz": v[0:3], "rpy": v[3:6], "axis": v[6:9]})
    return {"source": os.path.basename(path), "links": links, "joints": joints}


def load_fixture(path: str = FIXTURE) -> dict:
    with open(path) as fh:
        return json.load(fh)


# ------------------------------------------------------------------------------------------------ kinematics
def _rpy(r, p, y):
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    return torch.tensor([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                         [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                         [-sp, cp * sr, cp * cr]], dtype=torch.float64)


def _hat(w):
    z = torch.zeros((), dtype=torch.float64)
    return torch.stack((torch.stack((z, -w[2], w[1])), torch.stack((w[2], z, -w[0])), torch.stack((-w[1], w[0], z))))


def _exp_so3(phi):
    """Rodrigues' formula with the coefficients expanded in theta^2 (smooth at 0; exact to O(theta^8): the chart is only
    ever evaluated and differentiated at xi = 0)."""
    t2 = phi.dot(phi)
    a = 1.0 - t2 / 6.0 + t2 * t2 / 120.0 - t2 ** 3 / 5040.0
    b = 0.5 - t2 / 24.0 + t2 * t2 / 720.0 - t2 ** 3 / 40320.0
    K = _hat(phi)
    return torch.eye(3, dtype=torch.float64) + a * K + b * (K @ K)


def _axis_angle(axis, angle):
    """exact rotation about a fixed unit axis"""
    K = _hat(torch.as_tensor(axis, dtype=torch.float64))
    return torch.eye(3, dtype=torch.float64) + torch.sin(angle) * K + (1.0 - torch.cos(angle)) * (K @ K)


def quat_to_rot(q):
    x, y, z, w = q
    return torch.stack((torch.stack((1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y))),
                        torch.stack((2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x))),
                        torch.stack((2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)))))


class LagrangeModel:
    """The kinematic tree as a list of (link, parent link, joint) in topological order, plus per-link inertial data."""

    def __init__(self, data: dict, root: str = "base"):
        self.links = data["links"]
        by_parent = {}
        for j in data["joints"]:
            by_parent.setdefault(j["parent"], []).append(j)
        self.order = []  # (child, parent, joint dict, actuated index or None)
        stack = [root]
        while stack:
            parent = stack.pop()
            for j in by_parent.get(parent, []):
                idx = JOINT_ORDER.index(j["name"]) if j["type"] in ("revolute", "continuous") else None
                self.order.append((j["child"], parent, j, idx))
                stack.append(j["child"])
        self.root = root
        self.names = [root] + [c for c, _, _, _ in self.order]
        self.mass = torch.tensor([self.links[n]["mass"] for n in self.names], dtype=torch.float64)
        self.com = torch.tensor([self.links[n]["com"] for n in self.names], dtype=torch.float64)
        inertia = []
        for n in self.names:
            ixx, ixy, ixz, iyy, iyz, izz = self.links[n]["inertia"]
            RI = _rpy(*self.links[n]["com_rpy"])
            inertia.append(RI @ torch.tensor([[ixx, ixy, ixz], [ixy, iyy, iyz], [ixz, iyz, izz]], dtype=torch.float64) @ RI.T)
        self.inertia = torch.stack(inertia)  # about the centre of mass, in link axes
        assert sorted(i for _, _, _, i in self.order if i is not None) == list(range(12)), "12 actuated joints expected"

    def total_mass(self) -> float:
        return float(self.mass.sum())

    def kinematics(self, xi, p0, R0, theta0):
        """(centre-of-mass positions [L,3], link rotations [L,3,3], link origins [L,3]) in the world frame as functions of xi."""
        R = {self.root: R0 @ _exp_so3(xi[3:6])}
        o = {self.root: p0 + R0 @ xi[0:3]}
        for child, parent, j, idx in self.order:
            Rj = _rpy(*j["rpy"])
            if idx is not None:
                Rj = Rj @ _axis_angle(j["axis"], theta0[idx] + xi[6 + idx])
            o[child] = o[parent] + R[parent] @ torch.tensor(j["xyz"], dtype=torch.float64)
            R[child] = R[parent] @ Rj
        Rs = torch.stack([R[n] for n in self.names])
        os_ = torch.stack([o[n] for n in self.names])
        return os_ + torch.einsum("lij,lj->li", Rs, self.com), Rs, os_

    def mass_matrix(self, xi, p0, R0, theta0):
        f = lambda z: self.kinematics(z, p0, R0, theta0)[:2]  # noqa: E731
        Jc, JR = torch.func.jacfwd(f)(xi)  # [L,3,18], [L,3,3,18]
        _, R = f(xi)
        W = torch.einsum("lijk,lmj->limk", JR, R)  # (dR/dxi_k) R^T: skew-symmetric, its axial vector is the angular-velocity Jacobian
        Jw = torch.stack((W[:, 2, 1, :], W[:, 0, 2, :], W[:, 1, 0, :]), dim=1)
        Iw = torch.einsum("lij,ljk,lmk->lim", R, self.inertia, R)
        return torch.einsum("l,lik,lim->km", self.mass, Jc, Jc) + torch.einsum("lik,lij,ljm->km", Jw, Iw, Jw)

    def potential(self, xi, p0, R0, theta0, gravity):
        """V = - sum m g . c  for the uniform field `gravity` expressed in the inertial frame of the chart."""
        c, _, _ = self.kinematics(xi, p0, R0, theta0)
        return -(self.mass * (c @ gravity)).sum()

    # ------------------------------------------------------------------------------------------- dynamics
    def forward_dynamics(self, q, v, tau):
        """dv/dt (18) for q (19), v (18), generalised force tau (18: base wrench in the base frame, 12 joint torques)."""
        q, v, tau = (torch.as_tensor(a, dtype=torch.float64) for a in (q, v, tau))
        # Inertial frame of the chart := the base frame at xi = 0 (the equations of motion do not depend on that choice);
        # the stored quaternion then enters exactly as in Pinocchio's free-flyer, through the gravity field seen from the
        # base, R(quat)^T g, with R the polynomial of Eigen's toRotationMatrix -- also OFF the unit sphere, which fixes
        # the radial quaternion column of the Jacobian the same way the reference's tape does.
        p0, R0, theta0 = torch.zeros(3, dtype=torch.float64), torch.eye(3, dtype=torch.float64), q[7:19]
        gravity = quat_to_rot(q[3:7]).T @ torch.tensor([0.0, 0.0, -GRAVITY], dtype=torch.float64)
        xi0 = torch.zeros(18, dtype=torch.float64)
        M = self.mass_matrix(xi0, p0, R0, theta0)
        dM = torch.func.jacfwd(lambda z: self.mass_matrix(z, p0, R0, theta0))(xi0)  # [18,18,18]: dM[i,j,k] = d M_ij / d xi_k
        dV = torch.func.jacrev(lambda z: self.potential(z, p0, R0, theta0, gravity))(xi0)
        mdot_v = torch.einsum("ijk,k,j->i", dM, v, v)
        dT = 0.5 * torch.einsum("jki,j,k->i", dM, v, v)
        xidd = torch.linalg.solve(M, tau - mdot_v + dT - dV)
        return torch.cat((xidd[0:3] - torch.linalg.cross(v[3:6], v[0:3]), xidd[3:]))

    def inverse_dynamics_terms(self, q, v):
        """(M, h) with M a + h = tau in Pinocchio's velocity coordinates: h = nonlinear effects (Coriolis + gravity)."""
        zero = torch.zeros(18, dtype=torch.float64)
        a0 = self.forward_dynamics(q, v, zero)
        q_t = torch.as_tensor(q, dtype=torch.float64)
        M = self.mass_matrix(zero, torch.zeros(3, dtype=torch.float64), torch.eye(3, dtype=torch.float64), q_t[7:19])
        return M, -(M @ a0)

    def frame_positions(self, q, names):
        q = torch.as_tensor(q, dtype=torch.float64)
        _, _, o = self.kinematics(torch.zeros(18, dtype=torch.float64), q[0:3], quat_to_rot(q[3:7]), q[7:19])
        return torch.stack([o[self.names.index(n)] for n in names])


# ------------------------------------------------------------------------------------------------ node function
def _quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return torch.stack((aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz, aw * bz + az * bw + ax * by - ay * bx,
                        aw * bw - ax * bx - ay * by - az * bz))


def anymal_node(model: LagrangeModel, x, u, dt: float):
    """The ANYmal shooting-node function of DESIGN.md section 4.3 (Lie-group semi-implicit Euler of
    example/mpc/quadruped.example.cpp:197-200) on top of the Lagrangian forward dynamics."""
    x, u = torch.as_tensor(x, dtype=torch.float64), torch.as_tensor(u, dtype=torch.float64)
    q, v = x[:19], x[19:]
    tau = torch.cat((torch.zeros(6, dtype=torch.float64), u))
    vn = v + dt * model.forward_dynamics(q, v, tau)
    R = quat_to_rot(q[3:7])
    pn = q[0:3] + dt * (R @ vn[0:3])
    w = dt * vn[3:6]
    n = torch.sqrt(w.dot(w) + torch.finfo(torch.float64).eps)  # Utils::ApproximateNorm (utils.hpp:731-736)
    dq = torch.cat((w * torch.sin(0.5 * n) / n, torch.cos(0.5 * n).reshape(1)))  # ApproximateExponentialMap (utils.hpp:738-749)
    return torch.cat((pn, _quat_mul(q[3:7], dq), q[7:19] + dt * vn[6:18], vn))


def node_jacobian(model: LagrangeModel, x, u, dt: float):
    """EXACT Jacobian of the node function w.r.t. the 49 stored coordinates of (x, u): forward-mode automatic
    differentiation straight through the Lagrangian forward dynamics (a third level of torch.func nesting)."""
    z = torch.cat((torch.as_tensor(x, dtype=torch.float64), torch.as_tensor(u, dtype=torch.float64)))
    return torch.func.jacfwd(lambda t: anymal_node(model, t[:37], t[37:], dt))(z)


def node_jacobian_fd(model: LagrangeModel, x, u, dt: float, h: float = 1e-6):
    """Central differences of the node function w.r.t. (x, u) in the 49 stored coordinates (quaternion entries perturbed
    independently, exactly what the product's [A|B] block differentiates)."""
    z = np.concatenate((np.asarray(x, dtype=np.float64), np.asarray(u, dtype=np.float64)))
    J = np.zeros((37, 49))
    for j in range(49):
        zp, zm = z.copy(), z.copy()
        zp[j] += h
        zm[j] -= h
        J[:, j] = (anymal_node(model, zp[:37], zp[37:], dt).numpy() - anymal_node(model, zm[:37], zm[37:], dt).numpy()) / (2 * h)
    return J


if __name__ == "__main__":  # regenerate the URDF-values fixture (needs the reference tree)
    ref = os.environ.get("UNGAR_REFERENCE", "/root/reference")
    data = read_urdf(os.path.join(ref, "data", "robots", "anymal_b_description", "robots", "anymal.urdf"))
    with open(FIXTURE, "w") as fh:
        json.dump(data, fh, indent=1)
    print("wrote", FIXTURE, "mass", LagrangeModel(data).total_mass())
