"""Multi-GPU path on CPU: gloo, world_size 2 (the GPU box runs the same code over RCCL)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ungar_amd.sharding import reduce_sums, reduce_timing, shard_range


def test_shard_ranges_partition_the_batch():
    for total in (0, 1, 7, 4096, 65536, 65537):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(65536, 8, 3) == (24576, 32768)  # BASELINE config 5
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b, e = shard_range(4097, world, rank)
    # each rank "evaluates" its shard: a checksum over instance ids stands in for the outputs
    local = torch.arange(b, e, dtype=torch.float64)
    elapsed, evals = reduce_timing(0.010 * (rank + 1), (e - b) * 20, dist)
    s, nodes = reduce_sums([float(local.sum()), (e - b) * 20], dist)  # checksum-of-checksums and node count: SUM over ranks
    q.put((rank, elapsed, evals, s, nodes))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_timing_reduction_and_coverage():
    world = 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, elapsed, evals, checksum, nodes in out:
        assert elapsed == pytest.approx(0.020)        # MAX over ranks
        assert evals == 4097 * 20                     # SUM over ranks: every instance exactly once
        assert checksum == pytest.approx(4096 * 4097 / 2)
        assert nodes == 4097 * 20


def test_reductions_are_identities_on_one_rank():
    assert reduce_timing(0.5, 7) == (0.5, 7)
    assert reduce_sums([1.5, 2]) == [1.5, 2.0]


def test_padded_stride_properties():
    """Element stride of unit-fastest operands (sharding.padded_stride): whole 128-byte segments, never smaller than the node
    count, segment count 3 mod 4 (so never a multiple of a large power of two), idempotent enough to be cheap (< 128 nodes of padding)."""
    from ungar_amd.sharding import padded_stride
    for nodes in list(range(0, 200)) + [4096 * 20, 8192 * 20, 4096 * 128, 16384 * 200, 65536 * 20, 81937]:
        st = padded_stride(nodes)
        assert st >= nodes and st % 16 == 0 and (st // 16) % 4 == 3 and st - nodes < 128
    assert padded_stride(4096 * 20) == 81968
    with pytest.raises(ValueError):
        padded_stride(-1)


def test_bench_launches_its_own_ranks(repo_root, monkeypatch):
    """`python bench.py --gpus N` started directly (no WORLD_SIZE) re-executes itself under torch.distributed.run on 127.0.0.1 with one
    process per GPU; under a launcher it checks --gpus against WORLD_SIZE instead of guessing."""
    import importlib.util
    import subprocess
    import sys
    spec = importlib.util.spec_from_file_location("ungar_bench", os.path.join(repo_root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cmd = bench.launcher_command(4, ["--gpus", "4", "--steps", "3"], port=29999)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    assert cmd[-5] == os.path.join(repo_root, "bench.py") and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    free = bench.launcher_command(2, [])
    assert 1024 < int(free[free.index("--master-port") + 1]) < 65536
    calls = []
    monkeypatch.setattr(subprocess, "call", lambda c, **kw: calls.append((c, kw)) or 0)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as stop:
        bench.main()
    assert stop.value.code == 0 and len(calls) == 1 and "--nproc-per-node=2" in calls[0][0] and calls[0][0][-6:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]
    # under a launcher that disagrees with --gpus: refuse (no silent single-rank run reported as N GPUs)
    monkeypatch.setenv("WORLD_SIZE", "1")
    with pytest.raises(SystemExit) as stop:
        bench.main()
    assert "WORLD_SIZE=1" in str(stop.value.code)
    assert bench.reduce_min_max(3.5) == (3.5, 3.5)


def _run_bench(repo_root, args, env, timeout=600):
    import subprocess
    import sys
    return subprocess.run([sys.executable, os.path.join(repo_root, "bench.py"), *args], cwd=repo_root, capture_output=True, text=True, timeout=timeout,
                          env={**os.environ, "HSA_ENABLE_IPC_MODE_LEGACY": "0", **env})


@pytest.mark.gpu
def test_bench_runs_under_rccl_at_world_size_one(repo_root):
    """SURVEY.md section 8(e) on ONE GPU: `bench.py --gpus 1` under init_process_group("nccl", world_size=1, device_id=...) -- RCCL initialisation,
    the MAX / SUM / MIN all-reduces on DEVICE tensors, the barriers of the timed region and the teardown all execute; the line reports what the
    process group saw.  (The 2/4/8-GPU runs are the driver's; this is their first contact with RCCL.)"""
    import json
    out = _run_bench(repo_root, ["--gpus", "1", "--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--no-sub-results", "--prewarm-seconds", "0.1"],
                     {"UNGAR_BENCH_FORCE_DIST": "1", "UNGAR_BENCH_BACKEND": "nccl"})
    assert out.returncode == 0, out.stderr[-3000:]
    assert out.stdout.strip().splitlines()[-1].startswith("{"), out.stdout[-2000:]  # the JSON line is the last one, whatever the native libraries print
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["backend"] == "nccl" and d["ranks_seen"] == 1 and d["n_gpus"] == 1
    assert d["config"]["nodes_per_step"] == 4096 * 20 and abs(d["checksum"] - 5033491.53798481) < 1e-3
    assert d["per_rank_evals_per_s"]["min"] == d["per_rank_evals_per_s"]["max"] > 1e7
    assert abs(d["value"] - d["config"]["nodes_per_step"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]


@pytest.mark.gpu
def test_two_rccl_ranks_on_one_device_fail_cleanly(repo_root):
    """`python bench.py --gpus 2` with the RCCL backend where ONE device is visible: every rank exits non-zero with an explanation before any
    communicator is built (two RCCL ranks on one device would otherwise fail deep inside RCCL or wait for each other); no hang."""
    import torch as _torch
    if _torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with a single visible GPU")
    out = _run_bench(repo_root, ["--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], {"UNGAR_BENCH_BACKEND": "nccl", "UNGAR_BENCH_DIST_TIMEOUT": "60"}, timeout=300)
    assert out.returncode != 0
    assert "needs one GPU per rank" in out.stderr and "WORLD_SIZE=2" in out.stderr


def test_forced_process_group_of_one_rank_runs_the_collectives():
    """The reductions no longer short-cut an initialised group of one rank (gloo here, RCCL in the GPU test above)."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        assert reduce_timing(0.25, 11, dist) == (0.25, 11)
        assert reduce_sums([1.5, 2], dist) == [1.5, 2.0]
    finally:
        dist.destroy_process_group()
        for k in ("MASTER_ADDR", "MASTER_PORT", "RANK", "WORLD_SIZE"):
            os.environ.pop(k, None)
