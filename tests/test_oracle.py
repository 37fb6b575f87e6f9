"""Pins the CPU oracle (oracle/ungar_oracle.py) -- see its header for why the derivative half is
'parity unpinned' against the real reference and what pins it instead."""
import numpy as np
import pytest
import torch

from oracle import ungar_oracle as O

MODELS = ("quadrotor", "rc_car", "srbd", "anymal")


def test_closed_form_known_answers_of_the_reference():
    """test/autodiff/function.test.cpp:70-89, 120-131 through the oracle's AD (torch.autograd)."""
    rng = np.random.default_rng(0)
    for _ in range(32):
        x = torch.tensor(rng.uniform(-1, 1, 4))
        p = torch.tensor(rng.uniform(-1, 1))
        f = lambda z: torch.stack((p * (z * z).sum(), 2 * z[0] ** 2))  # noqa: E731
        J = torch.autograd.functional.jacobian(f, x)
        want = torch.zeros(2, 4)
        want[0] = 2 * p * x
        want[1, 0] = 4 * x[0]
        assert torch.allclose(J, want, atol=1e-14)
        H = torch.autograd.functional.hessian(lambda z: p * (z * z).sum(), x)
        assert torch.allclose(H, 2 * p * torch.eye(4), atol=1e-14)


def test_approximate_exponential_map_against_exact():
    """function.test.cpp:40-58: the recorded map is the approximate one; it equals the exact map to
    ~sqrt(eps) and is smooth at 0."""
    rng = np.random.default_rng(1)
    for _ in range(256):
        v = rng.uniform(-1, 1, 3)
        a = O.approximate_exponential_map(torch.tensor(v)).numpy()
        assert np.abs(a - O.exact_exponential_map(v)).max() < 1e-7
    z = O.approximate_exponential_map(torch.zeros(3)).numpy()
    assert np.abs(z - [0, 0, 0, 1]).max() < 1e-7
    J0 = torch.autograd.functional.jacobian(O.approximate_exponential_map, torch.zeros(3)).numpy()
    assert np.abs(J0[:3] - 0.5 * np.eye(3)).max() < 1e-7 and np.abs(J0[3]).max() < 1e-7


@pytest.mark.parametrize("name", MODELS)
def test_jacobian_matches_second_order_finite_differences(name):
    """The reference's own self-check (function.hpp:285-325), at a far tighter tolerance."""
    count = 3 if name == "anymal" else 8
    x, u, w, p = O.synthetic_inputs(name, count, seed=3)
    f, J = O.node_jacobian(name, x, u, w, p)
    nx = x.shape[1]
    z = np.concatenate((x, u), axis=1)
    eps = 1e-6
    for j in range(z.shape[1]):
        zp, zm = z.copy(), z.copy()
        zp[:, j] += eps
        zm[:, j] -= eps
        fd = (O.node_value(name, zp[:, :nx], zp[:, nx:], w, p) - O.node_value(name, zm[:, :nx], zm[:, nx:], w, p)) / (2 * eps)
        assert np.abs(fd - J[:, :, j]).max() < 1e-6 * max(1.0, np.abs(J).max())
    assert np.abs(f - O.node_value(name, x, u, w, p)).max() == 0.0


@pytest.mark.parametrize("name", MODELS)
def test_golden_fixtures_are_the_oracle(repo_root, name):
    g = np.load(f"{repo_root}/tests/golden/node_{name}.npz")
    sl = slice(0, 2 if name == "anymal" else 6)
    f, J = O.node_jacobian(name, g["x"][sl], g["u"][sl], g["w"][sl], g["p"][sl])
    assert np.array_equal(f, g["f"][sl]) and np.array_equal(J, g["J"][sl])


@pytest.mark.parametrize("name", ["quadrotor", "rc_car", "srbd", "anymal"])
def test_batched_evaluation_of_the_oracle_is_the_oracle(repo_root, name):
    """node_jacobian_batched (forward mode, vectorised over the batch: what the spread samples of the full-size launches use) against the committed fixtures of
    node_jacobian (reverse mode, node by node): the same node functions, two differentiation routes."""
    g = np.load(f"{repo_root}/tests/golden/node_{name}.npz")
    f, J = O.node_jacobian_batched(name, g["x"], g["u"], g["w"], g["p"])
    assert np.abs(f - g["f"]).max() <= 1e-12 * max(1.0, np.abs(g["f"]).max())
    assert np.abs(J - g["J"]).max() <= 1e-12 * np.abs(g["J"]).max()


def test_anymal_model_matches_reference_dimensions():
    """test/rbd/robot.test.cpp:103-106 (nq=19, nv=18), SURVEY.md Appendix D (13 moving joints,
    total mass 30.475 kg, leg order LF, LH, RF, RH)."""
    m = O.anymal_model()
    assert (m.nq, m.nv, len(m.joints)) == (19, 18, 14)
    assert [j.name for j in m.joints[2:]] == [f"{leg}_{jt}" for leg in ("LF", "LH", "RF", "RH") for jt in ("HAA", "HFE", "KFE")]
    assert abs(sum(j.Y[0, 0] for j in m.joints) - 30.475) < 1e-3
    assert all(np.allclose(j.axis, [1, 0, 0]) for j in m.joints[2::3]) and all(np.allclose(j.axis, [0, 1, 0]) for j in m.joints[3::3])


def test_aba_satisfies_inverse_dynamics_and_free_fall():
    """ABA checked against an independently written RNEA: tau_rnea(q, v, aba(q, v, tau)) == tau;
    and with v = 0, tau = 0 the robot free-falls: base acceleration = R^T g, joint accelerations 0."""
    m = O.anymal_model()
    rng = np.random.default_rng(5)
    for _ in range(5):
        quat = rng.normal(size=4)
        quat /= np.linalg.norm(quat)
        q = torch.tensor(np.concatenate((rng.uniform(-1, 1, 3), quat, rng.uniform(-1, 1, 12))))
        v = torch.tensor(rng.uniform(-1, 1, 18))
        tau = torch.tensor(rng.uniform(-20, 20, 18))
        a = O.aba(m, q, v, tau)
        assert torch.allclose(O.rnea(m, q, v, a), tau, atol=1e-9)
        a0 = O.aba(m, q, torch.zeros(18), torch.zeros(18))
        R = O._quat_to_rot(q[3:7])
        assert torch.allclose(a0[:3], R.T @ torch.tensor([0.0, 0.0, -9.81]), atol=1e-10)
        assert a0[3:].abs().max() < 1e-9
    # the mass matrix implied by RNEA is symmetric positive definite
    q = torch.tensor(np.concatenate(([0, 0, 0.5], [0, 0, 0, 1.0], 0.3 * np.ones(12))))
    M = torch.stack([O.rnea(m, q, torch.zeros(18), torch.eye(18)[i], gravity=False) for i in range(18)], dim=1)
    assert torch.allclose(M, M.T, atol=1e-10) and torch.linalg.eigvalsh(M).min() > 0
