#!/usr/bin/env python3
"""bench.py -- shooting-node Jacobian evaluations per second on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic input: for every (instance, knot)
of `batch` independent NMPC instances with horizon N, evaluate x+ = f(x,u) and the dense
[A|B] = df/d(x,u) block.  Default workload = BASELINE.json configs[3]: ANYmal-class quadruped
(nx=37, nu=12), N=20, batch=4096 per GPU.  Inputs are generated on the device before the timed
region; outputs stay resident.  N>1: one process per GPU (torch.distributed / RCCL), the batch axis
is sharded, no data-path collective; only the timing is reduced (MAX over ranks).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {  # name -> (model, N, default batch per GPU)
    "anymal": ("anymal", 20, 4096),        # configs[3] / [4]
    "quadrotor": ("quadrotor", 128, 4096),  # configs[1]
    "rc_car": ("rc_car", 200, 16384),       # configs[2]
    "srbd": ("srbd", 30, 4096),
}
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes(nx, nu):
    """read (x,u), write f and the dense nx x (nx+nu) block, FP64."""
    return 8 * ((nx + nu) + nx + nx * (nx + nu))


def synth_device_inputs(name, count, seed, torch):
    """Deterministic synthetic node inputs in the unit-fastest device layout (elements, count),
    ranges per SURVEY.md §8(d)."""
    from oracle import ungar_oracle as O  # only for the parameter VALUES of the reference examples
    nx, nu, nw, npar = O.DIMS[name]
    gen = torch.Generator(device="cuda")
    gen.manual_seed(0x5EED0000 + seed)
    r = lambda n, lo, hi: torch.rand((n, count), generator=gen, device="cuda", dtype=torch.float64) * (hi - lo) + lo  # noqa: E731
    quat = torch.randn((4, count), generator=gen, device="cuda", dtype=torch.float64)
    quat = quat / quat.norm(dim=0, keepdim=True)
    p = torch.as_tensor(O.default_params(name), device="cuda")
    w = None
    if name == "anymal":
        x = torch.cat((r(3, -1, 1), quat, r(12, -1, 1), r(18, -1, 1)))
        u = r(12, -20, 20)
    elif name == "quadrotor":
        hover = float(np.sqrt(1.5 * 9.80665 / (4 * 0.015)))
        x = torch.cat((r(3, -2, 2), quat, r(3, -1, 1), r(3, -1, 1)))
        u = r(4, 0.5, 1.5) * hover
    elif name == "rc_car":
        x = torch.cat((r(2, -1, 1), r(1, -np.pi, np.pi), r(1, 0.5, 2.0), r(1, -0.3, 0.3), r(1, -2, 2)))
        u = torch.cat((r(1, -1, 1), r(1, -0.3, 0.3)))
    else:  # srbd
        x = torch.cat((r(3, -2, 2), quat, r(3, -1, 1), r(3, -1, 1)))
        u = r(24, -1, 1)
        u[2::6] = 25.0 * 9.80665 / 4 * (u[2::6] * 0.5 + 1.0)
        w = (r(4, 0, 1) < 0.5).to(torch.float64)
    return x.contiguous(), u.contiguous(), w, p


def cpu_baseline(name, seconds=12.0):
    """Times the oracle's generated-C Jacobian (stand-in for the reference's CppADCodeGen C; see
    oracle/build_oracle.py) single-threaded, one instance per call -- the reference's execution
    model (function.hpp:216-230) -- on a bounded sample of the same workload."""
    from oracle import build_oracle
    from oracle import ungar_oracle as O
    tag, flags = "portable", build_oracle.PORTABLE_FLAGS
    lib = ctypes.CDLL(build_oracle.build(tag))  # rebuilt only when its generated sources are newer
    nx, nu, nw, npar = O.DIMS[name]
    sample = 2048
    x, u, w, p = O.synthetic_inputs(name, sample, seed=99)
    w = w if nw else np.zeros((sample, 1))
    nnz = ctypes.c_int.in_dll(lib, f"{name}_jac_nnz").value
    fn = getattr(lib, f"{name}_sparse_jacobian")
    dp = ctypes.POINTER(ctypes.c_double)
    fn.argtypes = [dp] * 6
    fn.restype = None
    f, jac = np.zeros(nx), np.zeros(nnz)
    ptr = lambda a: a.ctypes.data_as(dp)  # noqa: E731
    args = [(ptr(x[i]), ptr(u[i]), ptr(w[i]), ptr(p[i]), ptr(f), ptr(jac)) for i in range(sample)]
    for a in args[:64]:
        fn(*a)
    # ctypes call overhead (~1 us) is included; it is small against an ANYmal evaluation and is
    # stated in the sample description for the small models.
    evals, t0 = 0, time.perf_counter()
    while True:
        for a in args:
            fn(*a)
        evals += sample
        dt = time.perf_counter() - t0
        if dt >= seconds:
            break
    single = evals / dt
    # (b) the whole sample spread over all host cores: one thread per core, each evaluating its contiguous share
    # in ONE foreign call (C loop of oracle/_gen/batch_loops.c; ctypes releases the GIL) -- SURVEY.md section 8(d)
    import threading
    cores = min(64, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))  # bounded: the box may be CPU-throttled
    loop = getattr(lib, f"{name}_sparse_jacobian_batch")
    loop.argtypes = [dp] * 6 + [ctypes.c_long, ctypes.c_long, ctypes.c_long]
    loop.restype = None
    fo, jo = np.zeros((sample, nx)), np.zeros((sample, nnz))
    share = sample // cores  # nodes per thread; every thread sweeps its share `reps` times
    reps = max(1, int(single * 1.5 / max(1, share)))  # ~1.5 s of single-thread work per thread
    threads = [threading.Thread(target=loop, args=(ptr(x), ptr(u), ptr(w), ptr(p), ptr(fo), ptr(jo), t * share, (t + 1) * share, reps)) for t in range(cores)]
    t1 = time.perf_counter()
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    dt_all = time.perf_counter() - t1
    counts = [share * cores * reps]
    return {"value": single, "unit": "node Jacobian evals/s", "cores": 1, "kind": "port",
            "sample": f"{evals} single-instance calls of the tape-generated C Jacobian ({name}, sparse nnz={nnz}) over {sample} seeded nodes, "
                      f"gcc {' '.join(flags)}, 1 thread, {dt:.1f} s, via ctypes",
            "all_cores": {"value": sum(counts) / dt_all, "cores": cores, "seconds": dt_all}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="anymal", choices=sorted(WORKLOADS))
    ap.add_argument("--batch-per-gpu", type=int, default=None)
    ap.add_argument("--model", default=None, help="kernel variant of the workload's model (e.g. anymal_ad, anymal_reg)")
    ap.add_argument("--layout", default="soa", choices=["soa"])
    ap.add_argument("--jacobian", default="dense", choices=["dense", "sparse"],
                    help="dense [A|B] block (BASELINE metric, default) or the CSR value array of Function::Jacobian (function.hpp:216-230)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    import torch
    import ungar_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    device_index = local_rank % max(1, torch.cuda.device_count())  # one rank per GPU on a real node; the modulo only matters in the
    torch.cuda.set_device(device_index)                           # single-GPU dry run of the multi-rank control flow (gloo) below
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("UNGAR_BENCH_BACKEND", "nccl")  # nccl == RCCL on ROCm; "gloo" only for dry runs
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)

    model_name, N, default_batch = WORKLOADS[args.workload]
    batch = args.batch_per_gpu or default_batch
    count = batch * N  # nodes evaluated by THIS rank per step (batch axis sharded across ranks)
    kernel_model = args.model or model_name
    m = ungar_amd.NodeModel(kernel_model)
    nx, nu, ncols = m.nx, m.nu, m.nx + m.nu
    x, u, w, p = synth_device_inputs(model_name, count, seed=rank, torch=torch)
    f = torch.empty((nx, count), dtype=torch.float64, device="cuda")
    nnz = len(m.jacobian_sparsity()[0])
    jac_len = nx * ncols if args.jacobian == "dense" else nnz
    J = torch.empty((jac_len, count), dtype=torch.float64, device="cuda")
    Op = ungar_amd.Operand
    ops = (count, Op.soa(x, count, N), Op.soa(u, count, N), None if w is None else Op.soa(w, count, N), Op.per_instance(p, m.np, shared=True),
           Op.soa(f, count, N), Op.soa(J, count, N))
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        (m.dense_jacobian if args.jacobian == "dense" else m.sparse_jacobian)(*ops, knots=N, stream=stream)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        starts[i].record()  # same stream the kernel is launched on
        step()
        ends[i].record()
    fence()
    elapsed = time.perf_counter() - t0
    from ungar_amd.sharding import reduce_timing
    reduce_device = "cuda" if dist is None or dist.get_backend() == "nccl" else "cpu"
    elapsed, total_evals = reduce_timing(elapsed, count * args.steps, dist, reduce_device)  # MAX time, SUM evals over ranks
    kernel_ms = float(np.mean([s.elapsed_time(e) for s, e in zip(starts, ends)]))
    assert torch.isfinite(f).all() and torch.isfinite(J).all()

    if rank == 0:
        bytes_per_eval = algorithmic_bytes(nx, nu) if args.jacobian == "dense" else 8 * ((nx + nu) + nx + nnz)
        achieved = count * bytes_per_eval / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as fh:
                traffic = json.load(fh).get(f"{args.workload}:{batch}") if args.jacobian == "dense" and kernel_model == model_name else None
        out = {
            "metric": "shooting-node Jacobian evals/sec",
            "value": total_evals / elapsed,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{args.workload} shooting-node value + {'dense [A|B]' if args.jacobian == 'dense' else f'sparse (nnz={nnz})'} Jacobian, nx={nx} nu={nu}, N={N}, batch={batch}/GPU "
                                   f"({count} nodes/GPU/step), unit-fastest (SoA) device layout",
                       "horizon": N, "batch_per_gpu": batch, "nodes_per_step": count * world, "kernel_variant": kernel_model,
                       "parallelism": f"batch-sharded x{world}, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel_ms": kernel_ms, "algorithmic_bytes_per_eval": bytes_per_eval,
                         "kernel": f"NodeKernel<{kernel_model}, {args.jacobian} Jacobian>"},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(model_name, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
