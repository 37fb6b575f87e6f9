#!/usr/bin/env python3
"""Generates tests/golden/node_anymal_256.npz: 256 seeded ANYmal shooting nodes (inputs, f, dense [A|B]) from the INDEPENDENT oracle (oracle/ungar_oracle.py: torch.float64
autograd over a spatial-algebra restatement of ABA in Pinocchio's conventions and of the node's semi-implicit Euler step) -- the headline kernel against a derivation that shares
no code with the product, at more than tens of nodes.  Generated in the build container; the GPU box only loads the file.
Run from the repo root:  python tests/golden/make_anymal_many.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ungar_oracle as O  # noqa: E402

COUNT = 256
t0 = time.time()
x, u, w, p = O.synthetic_inputs("anymal", COUNT, seed=2024)
f, J = O.node_jacobian("anymal", x, u, w, p)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "node_anymal_256.npz"), x=x, u=u, w=w, p=p, f=f, J=J)
print("anymal", COUNT, "nodes, max|J|", np.abs(J).max(), f"{time.time() - t0:.1f} s")
