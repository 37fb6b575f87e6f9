#!/usr/bin/env python3
"""Generates tests/golden/rbd_<model>.npz for the batched rigid-body quantity models: seeded
inputs, values and Jacobians from the oracle (oracle/ungar_oracle.py: RNEA / RNEA-column mass matrix / forward kinematics /
spatial-momentum sum with torch.autograd -- none of it the product's CRBA, U D U^T or tape code).  The 324 x 19 Jacobian of
the mass matrix takes minutes per sample with reverse-mode autograd, hence a committed fixture instead of a live oracle.
Run from the repo root:  python tests/golden/make_rbd_golden.py [model ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ungar_oracle as O  # noqa: E402

COUNTS = {"anymal_rnea": 6, "anymal_crba": 3, "anymal_minv": 4, "anymal_feet": 6, "anymal_centroidal": 6}
for name in (sys.argv[1:] or list(COUNTS)):
    count = COUNTS[name]
    nx, nu, ny = O.RBD_DIMS[name]
    x, u = O.synthetic_rbd_inputs(name, count, seed=21)
    ys, Js = [], []
    for i in range(count):
        xi, ui = torch.as_tensor(x[i]), (torch.as_tensor(u[i]) if nu else None)
        if name == "anymal_minv":  # value-only model on the device
            ys.append(O.rbd_quantity(name, xi, ui).numpy())
            Js.append(np.zeros((ny, nx + nu)))
        else:
            y, J = O.rbd_quantity_jacobian(name, xi, ui)
            ys.append(y.numpy())
            Js.append(J.numpy())
        print(name, i, "max|y|", np.abs(ys[-1]).max(), "max|J|", np.abs(Js[-1]).max(), flush=True)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"rbd_{name}.npz"), x=x, u=u, y=np.array(ys), J=np.array(Js))
