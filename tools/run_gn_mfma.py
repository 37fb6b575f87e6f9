#!/usr/bin/env python3
"""A few launches of the matrix-core Gauss-Newton contraction on the config-4 shapes (for rocprofv3 counter passes: tools/gpu_pmc_mfma.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ungar_amd  # noqa: E402
from ungar_amd.sharding import unit_fastest  # noqa: E402

rows, cols, count = 37, 49, 81920
J = unit_fastest(rows * cols, count, torch)
J.copy_(torch.randn((rows * cols, count), device="cuda", dtype=torch.float64))
d = unit_fastest(rows, count, torch)
d.copy_(torch.rand((rows, count), device="cuda", dtype=torch.float64))
G = torch.zeros((count, cols, cols), dtype=torch.float64, device="cuda")
for _ in range(5):
    ungar_amd.gn_hessian_unit_fastest(J, d, G, rows, cols, count)
torch.cuda.synchronize()
