#!/usr/bin/env python3
"""Generates tests/golden/node_<model>.npz: seeded inputs and the oracle's (f, dense J) for each
built-in shooting-node model.  The expected values come from oracle/ungar_oracle.py (torch.float64
autograd over an independent restatement of the reference's node lambdas) -- nothing from
/root/reference is read.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ungar_oracle as O  # noqa: E402

COUNTS = {"quadrotor": 24, "rc_car": 32, "srbd": 16, "anymal": 8, "srbd_ineq": 24, "quadrotor_ineq": 16, "rc_car_ineq": 24, "srbd_feet": 16}
ONLY = sys.argv[1:]  # optional: regenerate these fixtures only

for name, count in COUNTS.items():
    if ONLY and name not in ONLY:
        continue
    x, u, w, p = O.synthetic_inputs(name, count, seed=7)
    f, J = O.node_jacobian(name, x, u, w, p)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"node_{name}.npz"), x=x, u=u, w=w, p=p, f=f, J=J)
    print(name, count, "max|J|", np.abs(J).max())

for cost, fname in (("quadrotor_cost", "cost_quadrotor.npz"), ("srbd_cost", "cost_srbd.npz"), ("rc_car_cost", "cost_rc_car.npz"), ("anymal_cost", "cost_anymal.npz")):
    if ONLY and cost not in ONLY:
        continue
    x, u, ref = O.synthetic_cost_inputs(32, seed=7, name=cost)
    y, g, H = O.cost_value_gradient_hessian(x, u, ref, name=cost)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", fname), x=x, u=u, p=ref, y=y, g=g, H=H)
    print(cost, 32, "max|H|", np.abs(H).max())
