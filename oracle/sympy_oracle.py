"""TEST INFRASTRUCTURE -- closed-form (symbolic) Jacobians of the three reference shooting-node functions.

The torch oracle (ungar_oracle.py) and the product's tape engine are both automatic-differentiation systems evaluating
restatements of the reference's node lambdas.  This module adds a derivative source of a different kind: each node function is
written ONCE MORE as a sympy expression, operation for operation after the reference
(example/mpc/quadrotor.example.cpp:126-190, example/mpc/rc_car.example.cpp:131-185, example/mpc/quadruped.example.cpp:148-203,
helpers include/ungar/utils/utils.hpp:731-749 and Eigen 3.4's quaternion formulas), differentiated symbolically
(`Matrix.jacobian`), and evaluated in floating point by `lambdify`.  tests/golden/sympy_<model>.npz holds seeded inputs with
values, closed-form Jacobians and the structural non-zero pattern of the symbolic Jacobian (the counts SURVEY.md section 8(a)
A6 quotes -- 118 / 32 / 238 -- came from this kind of probe).  Regenerate with `python oracle/sympy_oracle.py`.
"""
from __future__ import annotations

import os

import numpy as np
import sympy as sp

EPS = sp.Float(np.finfo(np.float64).eps)  # Eigen::NumTraits<double>::epsilon()
HERE = os.path.dirname(os.path.abspath(__file__))
DIMS = {"quadrotor": (13, 4, 0, 20), "rc_car": (6, 2, 0, 15), "srbd": (13, 24, 4, 6)}  # nx, nu, nw, np


def _cross(a, b):
    return sp.Matrix([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]])


def _rotate(q, v):
    """Eigen::QuaternionBase::_transformVector: v + w t + u x t with t = 2 (u x v); q = (x, y, z, w)."""
    u = sp.Matrix(q[:3])
    t = 2 * _cross(u, v)
    return v + q[3] * t + _cross(u, t)


def _quat_mul(a, b):
    """Eigen's quaternion product, (x, y, z, w) storage."""
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return sp.Matrix([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz, aw * bz + az * bw + ax * by - ay * bx,
                      aw * bw - ax * bx - ay * by - az * bz])


def _approximate_exponential_map(v):
    """Utils::ApproximateExponentialMap (utils.hpp:738-749) on Utils::ApproximateNorm (utils.hpp:731-736)."""
    n = sp.sqrt(v.dot(v) + EPS)
    return sp.Matrix(list(v * sp.sin(n / 2) / n) + [sp.cos(n / 2)])


def _lie_euler(pos, q, vel, om, dt, vel_dot, om_dot):
    """Lie-group semi-implicit Euler of the quadrotor / quadruped examples (quadrotor.example.cpp:184-187)."""
    vel_n = vel + dt * vel_dot
    om_n = om + dt * om_dot
    pos_n = pos + dt * vel_n
    q_n = _quat_mul(q, _approximate_exponential_map(dt * om_n))
    return sp.Matrix(list(pos_n) + list(q_n) + list(vel_n) + list(om_n))


def quadrotor(x, u, w, p):
    dt, m = p[0], p[1]
    moi = sp.Matrix(p[2:5])
    g0, b, d = p[17], p[18], p[19]
    pos, q, vel, om = sp.Matrix(x[0:3]), list(x[3:7]), sp.Matrix(x[7:10]), sp.Matrix(x[10:13])
    ez = sp.Matrix([0, 0, 1])
    sum_f, sum_m, sum_d = sp.zeros(3, 1), sp.zeros(3, 1), sp.zeros(3, 1)
    for i in range(4):
        thrust = b * u[i] ** 2 * ez
        sum_f += thrust
        sum_m += _cross(sp.Matrix(p[5 + 3 * i:8 + 3 * i]), thrust)
        sum_d += d * u[i] ** 2 * ez * (-1) ** i
    vel_dot = (_rotate(q, sum_f) - m * g0 * ez) / m
    om_dot = sp.Matrix([(sum_m[k] + sum_d[k] - _cross(om, sp.Matrix([moi[j] * om[j] for j in range(3)]))[k]) / moi[k] for k in range(3)])
    return _lie_euler(pos, q, vel, om, dt, vel_dot, om_dot)


def rc_car(x, u, w, p):
    dt, m, moi, lf, lr, Bf, Cf, Df, Br, Cr, Dr, Cm1, Cm2, Cr0, Cr2 = p
    px, py, phi, vx, vy, om = x
    d, delta = u
    alphaf = -sp.atan((om * lf + vy) / (vx + EPS)) + delta
    alphar = sp.atan((om * lr - vy) / (vx + EPS))
    Ffy = Df * sp.sin(Cf * sp.atan(Bf * alphaf))
    Fry = Dr * sp.sin(Cr * sp.atan(Br * alphar))
    Frx = (Cm1 - Cm2 * vx) * d - Cr0 - Cr2 * vx ** 2
    vx_n = vx + dt * (Frx - Ffy * sp.sin(delta) + m * vy * om) / m
    vy_n = vy + dt * (Fry + Ffy * sp.cos(delta) - m * vx * om) / m
    om_n = om + dt * (Ffy * lf * sp.cos(delta) - Fry * lr) / moi
    return sp.Matrix([px + dt * (vx_n * sp.cos(phi) - vy_n * sp.sin(phi)), py + dt * (vx_n * sp.sin(phi) + vy_n * sp.cos(phi)), phi + dt * om_n, vx_n, vy_n, om_n])


def srbd(x, u, w, p):
    dt, m = p[0], p[1]
    moi = sp.Matrix(p[2:5])
    g0 = p[5]
    pos, q, vel, om = sp.Matrix(x[0:3]), list(x[3:7]), sp.Matrix(x[7:10]), sp.Matrix(x[10:13])
    vel_dot = sp.Matrix([0, 0, -g0])
    om_dot = -_cross(om, sp.Matrix([moi[j] * om[j] for j in range(3)]))
    for i in range(4):
        f, r, s = sp.Matrix(u[6 * i:6 * i + 3]), sp.Matrix(u[6 * i + 3:6 * i + 6]), w[i]
        vel_dot += s * f / m
        om_dot += s * _cross(r, _rotate(q, f))
    om_dot = sp.Matrix([om_dot[k] / moi[k] for k in range(3)])
    return _lie_euler(pos, q, vel, om, dt, vel_dot, om_dot)


NODES = {"quadrotor": quadrotor, "rc_car": rc_car, "srbd": srbd}


def build(name):
    """(value function, Jacobian function, structural pattern) -- callables take (x, u, w, p) numpy rows."""
    nx, nu, nw, npar = DIMS[name]
    x, u, w, p = (sp.symbols(f"{s}0:{n}", real=True) if n else () for s, n in (("x", nx), ("u", nu), ("w", nw), ("p", npar)))
    f = NODES[name](x, u, w, p)
    J = f.jacobian(sp.Matrix(list(x) + list(u)))
    pattern = np.array([[0 if J[r, c] == 0 else 1 for c in range(nx + nu)] for r in range(nx)], dtype=np.int8)
    args = list(x) + list(u) + list(w) + list(p)
    fv = sp.lambdify(args, f, modules="numpy", cse=True)
    Jv = sp.lambdify(args, J, modules="numpy", cse=True)

    def call(fn, shape):
        def run(xr, ur, wr, pr):
            return np.asarray(fn(*xr, *ur, *(wr if nw else ()), *pr), dtype=np.float64).reshape(shape)
        return run

    return call(fv, (nx,)), call(Jv, (nx, nx + nu)), pattern


if __name__ == "__main__":
    import sys
    sys.path.insert(0, os.path.dirname(HERE))
    from oracle import ungar_oracle as O
    for name, count in (("quadrotor", 16), ("rc_car", 24), ("srbd", 12)):
        value, jacobian, pattern = build(name)
        x, u, w, p = O.synthetic_inputs(name, count, seed=41)
        f = np.array([value(x[i], u[i], w[i], p[i]) for i in range(count)])
        J = np.array([jacobian(x[i], u[i], w[i], p[i]) for i in range(count)])
        np.savez_compressed(os.path.join(os.path.dirname(HERE), "tests", "golden", f"sympy_{name}.npz"), x=x, u=u, w=w, p=p, f=f, J=J, pattern=pattern)
        print(name, "structural nnz", int(pattern.sum()), "max|J|", np.abs(J).max())
