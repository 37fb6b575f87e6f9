#!/usr/bin/env python3
"""Device Riccati solve (default route: the register-resident one-wavefront kernels of ocp_riccati_wave.hip where instantiated) against a plain numpy
restatement of the recursion of ocp_riccati.hpp on a sample of the instances, and its time per launch.  usage: check_riccati_wave.py [batch] [NXxNU ...]"""
import json
import sys

import numpy as np
import torch

import os  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ungar_amd.sqp import riccati_solve  # noqa: E402


def reference(AB, b, W, w, dx0, WN, wN, reg):
    N, nx, n = AB.shape
    nu = n - nx
    P, p = np.triu(WN) + np.triu(WN, 1).T + reg * np.eye(nx), wN.copy()
    Ks, ks = [None] * N, [None] * N
    for k in range(N - 1, -1, -1):
        Wk = np.triu(W[k]) + np.triu(W[k], 1).T + reg * np.eye(n)
        H = Wk + AB[k].T @ P @ AB[k]
        h = w[k] + AB[k].T @ (P @ b[k] + p)
        R = H[nx:, nx:]
        K = -np.linalg.solve(R, H[nx:, :nx])
        kff = -np.linalg.solve(R, h[nx:])
        Ks[k], ks[k] = K, kff
        P = H[:nx, :nx] + H[:nx, nx:] @ K
        P = 0.5 * (P + P.T)
        p = h[:nx] + H[:nx, nx:] @ kff
    dX, dU = [dx0], []
    for k in range(N):
        du = Ks[k] @ dX[-1] + ks[k]
        dU.append(du)
        dX.append(AB[k] @ np.concatenate([dX[-1], du]) + b[k])
    return np.array(dX), np.array(dU)


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    only = sys.argv[2:]
    g = torch.Generator(device="cuda").manual_seed(3)
    worst = 0.0
    for nx, nu, N in [(37, 12, 20), (25, 24, 30), (13, 24, 30), (17, 4, 30), (13, 4, 30)]:
        if only and f"{nx}x{nu}" not in only:
            continue
        n = nx + nu
        r = lambda *s: torch.randn(*s, generator=g, device="cuda", dtype=torch.float64)  # noqa: E731
        AB = 0.3 * r(batch, N, nx, n)
        AB[:, :, :, :nx] += torch.eye(nx, device="cuda", dtype=torch.float64)
        L = r(batch, N, n, n)
        W = torch.triu(0.1 * L @ L.transpose(-1, -2)).contiguous()
        LN = r(batch, nx, nx)
        WN = torch.triu(LN @ LN.transpose(-1, -2)).contiguous()
        b, w, wN, dx0 = 0.1 * r(batch, N, nx), r(batch, N, n), r(batch, nx), r(batch, nx)
        dX, dU, status = riccati_solve(nx, nu, N, batch, AB, b, W, w, dx0, WN, wN)
        torch.cuda.synchronize()
        assert int(status.abs().max()) == 0, "a reduced Hessian was reported indefinite"
        err = 0.0
        for i in sorted({0, 1, batch // 2, batch - 1}):
            c = lambda t: t[i].cpu().numpy()  # noqa: E731
            rX, rU = reference(c(AB), c(b), c(W), c(w), c(dx0), c(WN), c(wN), 1e-6)
            err = max(err, np.abs(c(dX) - rX).max() / np.abs(rX).max(), np.abs(c(dU) - rU).max() / np.abs(rU).max())
        times = []
        for _ in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            riccati_solve(nx, nu, N, batch, AB, b, W, w, dx0, WN, wN)
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1))
        print(json.dumps({"nx": nx, "nu": nu, "N": N, "batch": batch, "relative_error_vs_numpy": err, "ms_min": min(times), "ms_median": sorted(times)[4]}), flush=True)
        worst = max(worst, err)
    assert worst < 1e-9, worst
    print("OK")


if __name__ == "__main__":
    main()
