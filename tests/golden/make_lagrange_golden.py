#!/usr/bin/env python3
"""Generates tests/golden/anymal_lagrange.npz: seeded ANYmal node inputs with value, forward-dynamics accelerations and the
EXACT Jacobian from the Lagrangian oracle (oracle/lagrange_oracle.py: energies + torch.func automatic differentiation, robot
data from the second URDF reader's fixture tests/golden/anymal_urdf_values.json).  Nothing here shares a formula or a data
file with the product's ABA / RNEA / CRBA path.  Run from the repo root:  python tests/golden/make_lagrange_golden.py
(regenerate the URDF values first with `python oracle/lagrange_oracle.py` where the reference tree is present)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import lagrange_oracle as L  # noqa: E402

COUNT, DT = 10, 1.0 / 20.0
rng = np.random.default_rng(0x1A62A)
quat = rng.normal(size=(COUNT, 4))
quat /= np.linalg.norm(quat, axis=1, keepdims=True)
x = np.concatenate((rng.uniform(-1, 1, (COUNT, 3)), quat, rng.uniform(-1, 1, (COUNT, 12)), rng.uniform(-1, 1, (COUNT, 18))), axis=1)
u = rng.uniform(-20, 20, (COUNT, 12))
x[0, 19:] = 0.0  # at rest
u[1] = 0.0       # unactuated
model = L.LagrangeModel(L.load_fixture())
f, J, a = [], [], []
for i in range(COUNT):
    f.append(L.anymal_node(model, x[i], u[i], DT).numpy())
    J.append(L.node_jacobian(model, x[i], u[i], DT).numpy())
    a.append(model.forward_dynamics(x[i, :19], x[i, 19:], np.concatenate((np.zeros(6), u[i]))).numpy())
    print(i, "max|J|", np.abs(J[-1]).max(), "max|a|", np.abs(a[-1]).max(), flush=True)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "anymal_lagrange.npz"), x=x, u=u, w=np.zeros((COUNT, 0)), p=np.full((COUNT, 1), DT), f=np.array(f),
                    J=np.array(J), a=np.array(a))
