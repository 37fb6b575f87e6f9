"""Host logic of the bench workloads (ungar_amd/workloads.py): sizes and parameter values agree with the oracle's
independently written copy, the algorithmic-byte figures are SURVEY.md section 8(d)'s, and the synthetic inputs of a
node range do not depend on how the batch is partitioned over ranks."""
import numpy as np
import torch

from oracle import ungar_oracle as O
from ungar_amd import workloads as W
from ungar_amd.sharding import shard_range


def test_dims_and_parameters_match_the_oracle():
    for name, dims in W.DIMS.items():
        assert tuple(O.DIMS[name]) == dims
        np.testing.assert_array_equal(W.default_params(name), O.default_params(name))
        assert len(W.default_params(name)) == dims[3]
    for name in ("srbd_ineq", "quadrotor_ineq"):  # parameter blocks of the inequality nodes used by the SQP benches
        np.testing.assert_allclose(W.default_params(name), O.default_params(name), rtol=1e-15)


def test_algorithmic_bytes_are_the_survey_figures():
    got = {n: W.algorithmic_bytes(d[0], d[1]) for n, d in W.DIMS.items()}
    assert got == {"quadrotor": 2008, "rc_car": 496, "srbd": 4248, "anymal": 15192}
    assert W.algorithmic_bytes(13, 4, 118) == 1184 and W.algorithmic_bytes(6, 2, 32) == 368 and W.algorithmic_bytes(13, 24, 238) == 2304


def test_synthetic_inputs_do_not_depend_on_the_partition():
    for name, (_, N, _) in W.WORKLOADS.items():
        total, world = 12, 3
        whole = W.synth_device_inputs(name, total * N, 0, torch, device="cpu")
        parts = []
        for rank in range(world):
            b, e = shard_range(total, world, rank)
            parts.append(W.synth_device_inputs(name, total * N, 0, torch, device="cpu", begin=b * N, end=e * N))
        for k in range(3):  # x, u, w
            if whole[k] is None:
                continue
            assert torch.equal(whole[k], torch.cat([p[k] for p in parts], dim=1))
        x = whole[0]
        if name != "rc_car":
            assert torch.allclose(x[3:7].norm(dim=0), torch.ones(total * N, dtype=torch.float64))  # unit quaternions (xyzw)
