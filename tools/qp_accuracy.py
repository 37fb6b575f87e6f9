"""Diagnostics: how accurately do the two QP solvers solve the SAME quadratic subproblem?

Reads the dumps of `build/batched_quadruped_test <folder> <batch> <compared> <dump folder>` (QP data as assembled on the device, the
batched Riccati step, the facade's whole-KKT step), builds the dense KKT system of the stage-form QP, solves it in float64 and refines
it with residuals in extended precision (numpy longdouble, 64-bit mantissa) until the correction stalls -- a reference solution several
digits beyond what either float64 solver can deliver -- and prints the forward errors of both steps against it, per variable class."""
import glob
import sys

import numpy as np


def load(path):
    with open(path) as fh:
        N, nz, nu, ne, nc = map(int, fh.readline().split())
        vals = np.array([float(t) for t in fh.read().split()])
    nd, nx = nz + nu, nz - nc
    sizes = [N * nz * nd, N * nz, (N + 1) * nd * nd, (N + 1) * nd, N * ne * nd, (N + 1) * ne, nz, (N + 1) * nz, N * nu, (N + 1) * nx + N * nu]
    parts, at = [], 0
    for s in sizes:
        parts.append(vals[at:at + s])
        at += s
    assert at == len(vals)
    AB, b, W, w, E, e, dz0, dZ, dU, d = parts
    return dict(N=N, nz=nz, nu=nu, ne=ne, nc=nc, nx=nx, AB=AB.reshape(N, nz, nd), b=b.reshape(N, nz), W=W.reshape(N + 1, nd, nd), w=w.reshape(N + 1, nd),
                E=E.reshape(N, ne, nd), e=e.reshape(N + 1, ne), dz0=dz0, dZ=dZ.reshape(N + 1, nz), dU=dU.reshape(N, nu), d=d)


def kkt(q):
    N, nz, nu = q["N"], q["nz"], q["nu"]
    nvar = (N + 1) * nz + N * nu
    zs = lambda k: np.arange(k * nz, (k + 1) * nz)  # noqa: E731
    us = lambda k: np.arange((N + 1) * nz + k * nu, (N + 1) * nz + (k + 1) * nu)  # noqa: E731
    H, g = np.zeros((nvar, nvar)), np.zeros(nvar)
    for k in range(N):
        Wk = np.triu(q["W"][k])
        Wk = Wk + np.triu(Wk, 1).T
        idx = np.r_[zs(k), us(k)]
        H[np.ix_(idx, idx)] += Wk
        g[idx] += q["w"][k]
    WN = np.triu(q["W"][N][:nz, :nz])
    H[np.ix_(zs(N), zs(N))] += WN + np.triu(WN, 1).T
    g[zs(N)] += q["w"][N][:nz]
    rows, rhs = [], []
    for i in range(nz):
        r = np.zeros(nvar)
        r[i] = 1.0
        rows.append(r)
        rhs.append(q["dz0"][i])
    for k in range(N):
        for i in range(nz):
            r = np.zeros(nvar)
            r[zs(k + 1)[i]] = 1.0
            r[np.r_[zs(k), us(k)]] -= q["AB"][k][i]
            rows.append(r)
            rhs.append(q["b"][k][i])
        for j in range(q["ne"]):
            if np.any(q["E"][k][j] != 0.0):
                r = np.zeros(nvar)
                r[np.r_[zs(k), us(k)]] = q["E"][k][j]
                rows.append(r)
                rhs.append(-q["e"][k][j])
    A, c = np.array(rows), np.array(rhs)
    m = A.shape[0]
    K = np.block([[H, A.T], [A, np.zeros((m, m))]])
    return K, np.r_[-g, c], nvar


def main():
    for path in sorted(glob.glob(sys.argv[1] + "/qp_it*_inst*.txt")):
        q = load(path)
        K, rhs, nvar = kkt(q)
        x = np.linalg.solve(K, rhs)
        Kl, rl = K.astype(np.longdouble), rhs.astype(np.longdouble)
        xl = x.astype(np.longdouble)
        for it in range(12):
            res = rl - Kl @ xl
            corr = np.linalg.solve(K, res.astype(np.float64))
            xl = xl + corr.astype(np.longdouble)
            if np.abs(corr).max() <= 1e-17 * np.abs(x).max():
                break
        truth = xl[:nvar].astype(np.float64)
        N, nz, nu, nc, nx = q["N"], q["nz"], q["nu"], q["nc"], q["nx"]
        tZ, tU = truth[:(N + 1) * nz].reshape(N + 1, nz), truth[(N + 1) * nz:].reshape(N, nu)
        fX, fU = q["d"][:(N + 1) * nx].reshape(N + 1, nx), q["d"][(N + 1) * nx:].reshape(N, nu)
        scale = max(np.abs(tZ[:, nc:]).max(), np.abs(tU).max())
        cond = np.linalg.cond(K)
        def report(name, X, U):
            ex, eu = np.abs(X - tZ[:, nc:]).max(), np.abs(U - tU).max()
            force = np.abs(U - tU).reshape(N, 4, 6)[:, :, :3].max()
            foot = np.abs(U - tU).reshape(N, 4, 6)[:, :, 3:].max() if nu == 24 else float("nan")
            return f"{name}: states {ex / scale:.2e}  inputs {eu / scale:.2e} (forces {force / scale:.2e}, footholds {foot / scale:.2e})"
        print(f"{path.split('/')[-1]}: cond(KKT) {cond:.2e}, |d|max {scale:.3g}, refinement steps {it + 1}, numpy float64 solve error {np.abs(x[:nvar] - truth).max() / scale:.2e}")
        print("   ", report("batched Riccati", q["dZ"][:, nc:], q["dU"]))
        print("   ", report("facade KKT     ", fX, fU))


if __name__ == "__main__":
    main()
