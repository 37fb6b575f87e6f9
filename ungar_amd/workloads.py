"""Benchmark / test workloads of the batched node evaluation: sizes, the reference examples' parameter
values, and deterministic synthetic inputs generated ON THE DEVICE.

Product-side host logic: nothing here touches oracle/ (the oracle keeps its own, independently written copy of
the parameter values -- tests/test_workloads.py checks that the two agree).
"""
from __future__ import annotations

import math

import numpy as np

# name -> (nx, nu, nw, np): state, input, per-node parameters, per-instance parameters of the built-in node models
DIMS = {
    "quadrotor": (13, 4, 0, 20),
    "rc_car": (6, 2, 0, 15),
    "srbd": (13, 24, 4, 6),
    "anymal": (37, 12, 0, 1),
}

# BASELINE.json configs: name -> (model, horizon N, default instances per GPU)
WORKLOADS = {
    "anymal": ("anymal", 20, 4096),         # configs[3] (1 GPU) / configs[4] (65 536 instances over 8 GPUs)
    "quadrotor": ("quadrotor", 128, 4096),  # configs[1]
    "rc_car": ("rc_car", 200, 16384),       # configs[2]
    "srbd": ("srbd", 30, 4096),             # the reference's own quadruped example, scaled to a batch
}
CONFIG5_TOTAL_BATCH = 65536  # BASELINE.json configs[4]: the fixed batch that is partitioned over the GPUs of a node


def algorithmic_bytes(nx: int, nu: int, jac_entries: int | None = None) -> int:
    """read (x, u), write f and the Jacobian block (dense nx x (nx+nu) unless `jac_entries`), FP64."""
    return 8 * ((nx + nu) + nx + (nx * (nx + nu) if jac_entries is None else jac_entries))


def default_params(name: str) -> np.ndarray:
    """Per-instance parameter block p with the values of the reference's examples."""
    if name == "quadrotor":  # example/mpc/quadrotor.example.cpp:326-343: dt, mass, inertia diag, 4 propeller positions, g, b, k
        arms = [[0.2, 0.2, 0.0], [-0.2, 0.2, 0.0], [-0.2, -0.2, 0.0], [0.2, -0.2, 0.0]]
        return np.array([1.0 / 30.0, 1.5, 3e-2, 3e-2, 3e-2] + [c for a in arms for c in a] + [9.80665, 0.015, 0.1])
    if name == "rc_car":  # example/mpc/rc_car.example.cpp:320-337
        return np.array([1.0 / 30.0, 0.041, 27.8e-6, 0.029, 0.033, 2.579, 1.2, 0.192, 3.3852, 1.2691, 0.1737, 0.287, 0.0545, 0.0518,
                         0.00035])
    if name == "srbd":  # example/mpc/quadruped.example.cpp:378-392: dt, mass, inertia diag, g
        return np.array([1.0 / 30.0, 25.0, 0.048125, 0.093125, 0.055625, 9.80665])
    if name == "anymal":  # dt = 1 / N of BASELINE config 4
        return np.array([1.0 / 20.0])
    if name == "srbd_ineq":  # example/mpc/quadruped.example.cpp:384-392: friction coefficient, four hip offsets, maximum leg extension
        return np.array([0.7, 0.2, 0.15, -0.1, 0.2, -0.15, -0.1, -0.2, 0.15, -0.1, -0.2, -0.15, -0.1, 0.42])
    if name == "quadrotor_ineq":  # maximum rotor speed: twice the hover speed (example/mpc/quadrotor.example.cpp:356-358)
        return np.array([2.0 * np.sqrt(1.5 * 9.80665 / (4 * 0.015))])
    raise KeyError(name)


def synth_device_inputs(name: str, count: int, seed: int, torch, device="cuda", begin: int = 0, end: int | None = None):
    """Deterministic node inputs (x, u, w, p) in the unit-fastest device layout (elements, nodes), value ranges per
    SURVEY.md §8(d).  The random stream is a function of (seed, count) only: a rank that owns the node range
    [begin, end) of a `count`-node batch gets exactly the columns a single GPU would see for those nodes, so whole-batch
    checksums do not depend on how the batch is partitioned."""
    end = count if end is None else end
    gen = torch.Generator(device=device)
    gen.manual_seed(0x5EED0000 + seed)

    def r(n, lo, hi):
        return (torch.rand((n, count), generator=gen, device=device, dtype=torch.float64) * (hi - lo) + lo)[:, begin:end]

    def unit_quaternions():
        q = torch.randn((4, count), generator=gen, device=device, dtype=torch.float64)[:, begin:end]
        return q / q.norm(dim=0, keepdim=True)

    p = torch.as_tensor(default_params(name), device=device)
    w = None
    if name == "anymal":  # base position, unit quaternion (xyzw), 12 joint angles, 18 velocities; 12 joint torques
        x = torch.cat((r(3, -1, 1), unit_quaternions(), r(12, -1, 1), r(18, -1, 1)))
        u = r(12, -20, 20)
    elif name == "quadrotor":
        hover = math.sqrt(1.5 * 9.80665 / (4 * 0.015))  # rotor speed at hover, quadrotor.example.cpp:356-358
        x = torch.cat((r(3, -2, 2), unit_quaternions(), r(3, -1, 1), r(3, -1, 1)))
        u = r(4, 0.5, 1.5) * hover
    elif name == "rc_car":
        x = torch.cat((r(2, -1, 1), r(1, -math.pi, math.pi), r(1, 0.5, 2.0), r(1, -0.3, 0.3), r(1, -2, 2)))
        u = torch.cat((r(1, -1, 1), r(1, -0.3, 0.3)))
    elif name == "srbd":
        x = torch.cat((r(3, -2, 2), unit_quaternions(), r(3, -1, 1), r(3, -1, 1)))
        u = r(24, -1, 1).clone()
        u[2::6] = 25.0 * 9.80665 / 4 * (u[2::6] * 0.5 + 1.0)
        w = (r(4, 0, 1) < 0.5).to(torch.float64).contiguous()
    else:
        raise KeyError(name)
    return x.contiguous(), u.contiguous(), w, p
