#!/usr/bin/env python3
"""Generates tests/golden/node_{quadrotor,rc_car,anymal}_edge.npz (same keys as node_<model>.npz): shooting nodes that sit ON the edge cases of the reference's helper functions, with the oracle's (f, dense J)
(oracle/ungar_oracle.py: torch.float64 autograd over an independent restatement of the node lambdas; nothing from /root/reference is read).
  quadrotor  * body angular velocity 0 and four equal rotor speeds: omega+ = 0 EXACTLY, so Utils::ApproximateExponentialMap is evaluated at the zero vector
               (utils.hpp:731-749; the reference pins that point for the bare helper, test/autodiff/function.test.cpp:40-58)
             * a stored quaternion that is NOT of unit length (|q| = 1.3, 0.7): the Lie-group integrator does not normalise (quadrotor.example.cpp:184-187)
  rc_car     * v_x at the lower bound of the synthetic range (0.5) and at the minimum-velocity bound of the OCP (0.3), with v_y = omega = 0 (slip-angle
               arguments of atan exactly 0, rc_car.example.cpp:158-161)
  anymal     * the robot at rest (v = 0, tau = 0: omega+ of the base comes from gravity alone) with the joints at 0, and a non-unit base quaternion
Run from the repo root:  python tests/golden/make_edge_cases.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ungar_oracle as O  # noqa: E402

out = {}
# ---- quadrotor: x = (p 3, q xyzw 4, v 3, omega 3), u = 4 rotor speeds
x, u, w, p = O.synthetic_inputs("quadrotor", 8, seed=11)
x[0:4, 10:13] = 0.0
u[0:4] = u[0:4, :1]            # equal rotor speeds: no net torque
x[0, 3:7] = [0.0, 0.0, 0.0, 1.0]  # ... also at the identity orientation
x[4:6, 3:7] *= 1.3
x[6:8, 3:7] *= 0.7
f, J = O.node_jacobian("quadrotor", x, u, w, p)
quat_next = f[0:4, 3:7]
assert np.abs(np.linalg.norm(quat_next, axis=1) - np.linalg.norm(x[0:4, 3:7], axis=1)).max() < 1e-12  # omega+ = 0: the orientation does not move
assert abs(np.linalg.norm(x[4, 3:7]) - 1.3) < 1e-12
out.update(quadrotor_x=x, quadrotor_u=u, quadrotor_w=w, quadrotor_p=p, quadrotor_f=f, quadrotor_J=J)
# ---- rc_car: x = (p 2, phi, v 2, omega), u = (d, delta)
x, u, w, p = O.synthetic_inputs("rc_car", 8, seed=11)
x[0:4, 3] = [0.5, 0.5, 0.3, 0.3]
x[0:4, 4:6] = 0.0
u[1, :] = 0.0
u[3, :] = 0.0
f, J = O.node_jacobian("rc_car", x, u, w, p)
assert np.isfinite(J).all()
out.update(rc_car_x=x, rc_car_u=u, rc_car_w=w, rc_car_p=p, rc_car_f=f, rc_car_J=J)
# ---- anymal: x = (p 3, q xyzw 4, joints 12, v 18), u = 12 torques
x, u, w, p = O.synthetic_inputs("anymal", 4, seed=11)
x[0:2, 19:37] = 0.0
u[0:2] = 0.0
x[0, 7:19] = 0.0
x[0, 3:7] = [0.0, 0.0, 0.0, 1.0]
x[2, 3:7] *= 1.3
x[3, 3:7] *= 0.7
f, J = O.node_jacobian("anymal", x, u, w, p)
assert np.isfinite(J).all()
out.update(anymal_x=x, anymal_u=u, anymal_w=w, anymal_p=p, anymal_f=f, anymal_J=J)
for name in ("quadrotor", "rc_car", "anymal"):
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"node_{name}_edge.npz"), **{k: out[f"{name}_{k}"] for k in ("x", "u", "w", "p", "f", "J")})
    print(name, out[name + "_x"].shape[0], "nodes, max|J|", np.abs(out[name + "_J"]).max())
