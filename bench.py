#!/usr/bin/env python3
"""bench.py -- shooting-node Jacobian evaluations per second on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic input: for every (instance, knot) of a batch of
independent NMPC instances with horizon N, evaluate x+ = f(x,u) and the dense [A|B] = df/d(x,u) block.

  --gpus 1 (default)  BASELINE.json configs[3]: ANYmal-class quadruped (nx=37, nu=12), N=20, batch=4096.
  --gpus G > 1        BASELINE.json configs[4]: the FIXED batch of 65 536 instances partitioned over the G ranks
                      (contiguous split of the instance axis, ungar_amd.sharding.shard_range; "scaling": "strong").
                      --batch-per-gpu B keeps B instances on every rank instead ("scaling": "weak").
One process per GPU (torch.distributed; backend nccl == RCCL).  The path has no exchange step: the
only collectives are MAX of the elapsed time and SUM of the evaluation count and of the output checksum.

Inputs are generated on the device before the timed region; outputs stay resident.  Rank 0 prints ONE JSON line with
`roofline` (dominant kernel, HIP events on the launch stream), `sub_results` (configs[1] quadrotor N=128 x 4096 and
configs[2] rc_car N=200 x 16384, 1 GPU) and `cpu_baseline` (the oracle's tape-generated C on the host cores).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import statistics
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


# ------------------------------------------------------------------------------------------------ CPU baseline
class NativeOracleBuild:
    """Compiles the oracle's generated C with the reference's JIT flags (-O3 -g -march=native -mtune=native -ffast-math,
    function.hpp:610-611) ON THIS BOX, in a background thread started before the GPU measurement; falls back to the
    prebuilt -march=x86-64-v3 library if the compiler is missing or does not finish within `patience` seconds."""

    def __init__(self, models):
        self.models, self.path, self.error = tuple(models), None, None
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    def _run(self):
        try:
            from oracle import build_oracle
            self.path = build_oracle.build("native", models=self.models)
        except Exception as exc:  # noqa: BLE001 -- any failure means "use the portable build"
            self.error = repr(exc)

    def library(self, patience: float):
        from oracle import build_oracle
        self.thread.join(timeout=patience)
        if self.path:
            return self.path, "native", build_oracle.REFERENCE_FLAGS
        return build_oracle.build("portable"), "portable", build_oracle.PORTABLE_FLAGS


def cpu_baseline(name, native: NativeOracleBuild | None, seconds=10.0, patience=240.0, light=False):
    """Times the oracle's generated-C Jacobian (stand-in for the reference's CppADCodeGen C; oracle/build_oracle.py):
    (a) single-threaded, one instance per call -- the reference's execution model (function.hpp:216-230) -- median of 5
    sweeps over a bounded sample, for the structured program AND for the taped-ABA program (the route the reference
    itself takes: record ABA, let the AD tool differentiate it); (b) the same sample over all host cores."""
    from oracle import build_oracle
    from oracle import ungar_oracle as O
    if native is not None:
        path, tag, flags = native.library(patience)
    else:
        path, tag, flags = build_oracle.build("portable"), "portable", build_oracle.PORTABLE_FLAGS
    lib = ctypes.CDLL(path)
    nx, nu, nw, npar = O.DIMS[name]
    sample = 2048
    x, u, w, p = O.synthetic_inputs(name, sample, seed=99)
    w = w if nw else np.zeros((sample, 1))
    dp = ctypes.POINTER(ctypes.c_double)
    ptr = lambda a: a.ctypes.data_as(dp)  # noqa: E731

    def single_thread(model):
        """median over 5 sweeps of evals/s; every sweep runs whole passes over the sample for ~seconds/5."""
        if not hasattr(lib, f"{model}_sparse_jacobian_batch"):
            return None
        nnz = ctypes.c_int.in_dll(lib, f"{model}_jac_nnz").value
        loop = getattr(lib, f"{model}_sparse_jacobian_batch")  # C loop around the single-instance function: no per-call Python overhead
        loop.argtypes = [dp] * 6 + [ctypes.c_long, ctypes.c_long, ctypes.c_long]
        loop.restype = None
        fo, jo = np.zeros((sample, nx)), np.zeros((sample, nnz))
        t0 = time.perf_counter()
        loop(ptr(x), ptr(u), ptr(w), ptr(p), ptr(fo), ptr(jo), 0, sample, 1)  # warm-up sweep, also sizes the repetitions
        per_pass = max(time.perf_counter() - t0, 1e-6)
        reps = max(1, int(seconds / 5 / per_pass))
        rates = []
        for _ in range(5):
            t0 = time.perf_counter()
            loop(ptr(x), ptr(u), ptr(w), ptr(p), ptr(fo), ptr(jo), 0, sample, reps)
            rates.append(sample * reps / (time.perf_counter() - t0))
        return {"value": statistics.median(rates), "sweeps": rates, "evals_per_sweep": sample * reps, "nnz": nnz}

    structured = single_thread(name)
    if light:  # the secondary workloads (configs 0-2): one thread, one figure
        return {"value": structured["value"], "unit": "node Jacobian evals/s", "cores": 1, "kind": "port",
                "sample": f"median of 5 sweeps x {structured['evals_per_sweep']} single-instance evaluations of the tape-generated C Jacobian ({name}, sparse nnz={structured['nnz']}) "
                          f"over {sample} seeded nodes, gcc {' '.join(flags)} ({'compiled on this box' if tag == 'native' else 'prebuilt portable library'}), 1 thread",
                "sweeps": structured["sweeps"]}
    taped = single_thread(name + "_ad") if name == "anymal" else None
    # (b) all host cores: one thread per core, each sweeping its contiguous share in ONE foreign call (ctypes releases the GIL)
    cores = min(64, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))  # bounded: the box may be CPU-throttled
    nnz = structured["nnz"]
    loop = getattr(lib, f"{name}_sparse_jacobian_batch")
    fo, jo = np.zeros((sample, nx)), np.zeros((sample, nnz))
    share = max(1, sample // cores)
    reps = max(1, int(structured["value"] * 1.5 / share))  # ~1.5 s of single-thread work per thread
    threads = [threading.Thread(target=loop, args=(ptr(x), ptr(u), ptr(w), ptr(p), ptr(fo), ptr(jo), t * share, (t + 1) * share, reps))
               for t in range(min(cores, sample))]
    t1 = time.perf_counter()
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    dt_all = time.perf_counter() - t1
    out = {"value": structured["value"], "unit": "node Jacobian evals/s", "cores": 1, "kind": "port",
           "sample": f"median of 5 sweeps x {structured['evals_per_sweep']} single-instance evaluations of the tape-generated C Jacobian "
                     f"({name}, structured program, sparse nnz={nnz}) over {sample} seeded nodes, gcc {' '.join(flags)} "
                     f"({'compiled on this box' if tag == 'native' else 'prebuilt portable library: native build unavailable'}), 1 thread",
           "sweeps": structured["sweeps"],
           "all_cores": {"value": share * len(threads) * reps / dt_all, "cores": len(threads), "seconds": dt_all}}
    out["genuine_reference"] = genuine_reference_timing(min(seconds, 5.0))
    if taped:
        out["taped_aba"] = {"value": taped["value"], "sweeps": taped["sweeps"], "nnz": taped["nnz"],
                            "note": "same node function, derivatives by taping ABA (test/rbd/robot.test.cpp:124-135) -- the reference's own route"}
    return out


def genuine_reference_timing(seconds):
    """BASELINE.md section 3.2 / SURVEY.md section 8(d): where the reference's own stack is installed (<cppad/cg.hpp>), build and time the
    REAL Ungar::Autodiff::Function (oracle/ref_timing/); otherwise say so.  In this image the stack is absent."""
    import subprocess
    script = os.path.join(ROOT, "oracle", "ref_timing", "build_reference_timing.sh")
    try:
        probe = subprocess.run(["bash", script], capture_output=True, text=True, timeout=600)
        if probe.returncode != 0:
            return (probe.stdout.strip().splitlines() or ["unavailable"])[-1]
        run = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "time_reference_function"), str(seconds)], capture_output=True, text=True, timeout=120 + 4 * seconds)
        return json.loads(run.stdout.strip().splitlines()[-1]) if run.returncode == 0 else "unavailable: the genuine-reference program failed at run time"
    except Exception as exc:  # noqa: BLE001
        return f"unavailable: {exc!r}"


# ------------------------------------------------------------------------------------------------ GPU measurement
def measure(torch, ungar_amd, workload, instances, total_instances, begin, steps, warmup, jacobian="dense", kernel_model=None, fence=None,
            seed=0, tile=None, pad_stride=True, prewarm_s=0.0, layout="soa"):
    """Times `steps` passes over the node range of instances [begin, begin + instances) of a `total_instances` batch.  The shard
    is stored as [tile][element][node of tile] (ungar_amd.sharding.tile_ranges) and a pass is one launch per tile on one stream.
    Returns per-rank figures (elapsed seconds on the host clock, mean launch duration from HIP events on the launch stream,
    nodes per step, output checksum)."""
    from ungar_amd import workloads as W
    from ungar_amd.sharding import DEFAULT_TILE_INSTANCES, tile_ranges, unit_fastest
    model_name, N, _ = W.WORKLOADS[workload]
    m = ungar_amd.NodeModel(kernel_model or model_name)
    nx, nu, ncols = m.nx, m.nu, m.nx + m.nu
    count = instances * N
    x, u, w, p = W.synth_device_inputs(model_name, total_instances * N, seed, torch, begin=begin * N, end=(begin + instances) * N)
    nnz = m.jac_nnz
    jac_len = nx * ncols if jacobian == "dense" else nnz
    Op = ungar_amd.Operand
    launches, outputs = [], []
    wave_tiles = layout == "tiles"  # dense block as register images of the wavefronts (include/ungar_amd.h: ungar_tile_layout); inputs and f stay unit-fastest
    if wave_tiles and jacobian != "dense":
        raise SystemExit("bench.py: the wave-tile layout holds the dense block")
    for b, e in tile_ranges(instances, tile or DEFAULT_TILE_INSTANCES):
        tn = (e - b) * N
        sl = slice(b * N, e * N)
        def operand(src_rows, src=None):  # (elements, tn) view whose element stride is padded off the power-of-two channel stride
            t = unit_fastest(src_rows, tn, torch) if pad_stride else torch.empty((src_rows, tn), dtype=torch.float64, device="cuda")
            if src is not None:
                t.copy_(src)
            return t
        xt, ut, wt = operand(nx, x[:, sl]), operand(nu, u[:, sl]), None if w is None else operand(w.shape[0], w[:, sl])
        f = operand(nx)
        # (wave tiles: zero-filled once, so that the padding of the last band -- never written -- does not enter the checksum)
        J = torch.zeros((m.tile_doubles(tn),), dtype=torch.float64, device="cuda") if wave_tiles else operand(jac_len)
        outputs.append((f, J))
        es = f.stride(0)
        launches.append((tn, Op.soa(xt, es, N), Op.soa(ut, es, N), None if wt is None else Op.soa(wt, es, N), Op.per_instance(p, m.np, shared=True),
                         Op.soa(f, es, N), J if wave_tiles else Op.soa(J, es, N)))
    del x, u, w
    stream = torch.cuda.current_stream().cuda_stream
    call = m.dense_jacobian_tiles if wave_tiles else m.dense_jacobian if jacobian == "dense" else m.sparse_jacobian

    def step():
        for ops in launches:
            call(*ops, knots=N, stream=stream)

    fence = fence or torch.cuda.synchronize
    # Untimed pre-warm: the clock governor needs a few hundred milliseconds of sustained load to reach the steady-state clocks this
    # FP64-latency-bound kernel runs at (measured: 0.262-0.266 ms per launch right after start, 0.253-0.255 ms after 0.3 s of the
    # same launches, profiles/archive/r02e_prewarm.log).  It precedes the W warm-up steps; the timed region is exactly `steps` steps.
    t_end = time.perf_counter() + prewarm_s
    while time.perf_counter() < t_end:
        for _ in range(50):
            step()
        torch.cuda.synchronize()
    for _ in range(warmup):
        step()
    fence()
    # One pair of HIP events around the timed region, on the stream the kernels are launched on: the launch duration is (region / launches).  (Until round 5 every
    # step had its own pair: the event packets between the launches kept the command processor from preparing the next dispatch during the current one and read
    # 2-4 % above rocprofv3's per-kernel average of the same command -- the contract asks the two to agree.)
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    start.record()
    for i in range(steps):
        step()
    end.record()
    fence()
    elapsed = time.perf_counter() - t0
    step_ms = float(start.elapsed_time(end)) / steps
    assert all(bool(torch.isfinite(f).all()) and bool(torch.isfinite(J).all()) for f, J in outputs)
    checksum = float(sum(f.sum().item() + J.sum().item() for f, J in outputs))  # summed over ranks afterwards: independent of the partition (up to rounding)
    bytes_per_eval = W.algorithmic_bytes(nx, nu, None if jacobian == "dense" else nnz)
    return {"elapsed": elapsed, "kernel_ms": step_ms / len(launches), "step_kernel_ms": step_ms, "launches_per_step": len(launches), "count": count,
            "nodes_per_launch": count / len(launches), "checksum": checksum, "bytes_per_eval": bytes_per_eval, "nx": nx, "nu": nu, "nnz": nnz, "N": N,
            "kernel_model": kernel_model or model_name, "layout": layout}


def roofline(r, jacobian, traffic=None, traffic_source=None):
    achieved = r["nodes_per_launch"] * r["bytes_per_eval"] / (r["kernel_ms"] * 1e-3) / 1e9  # algorithmic bytes per launch / launch duration
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_source": traffic_source, "kernel_ms": r["kernel_ms"], "launches_per_step": r["launches_per_step"], "nodes_per_launch": r["nodes_per_launch"],
            "algorithmic_bytes_per_eval": r["bytes_per_eval"],
            "kernel": f"NodeKernel<{r['kernel_model']}, {jacobian} Jacobian, {'wave-tile' if r.get('layout') == 'tiles' else 'unit-fastest'} operand>"}


def box_store_ceilings(timeout=60):
    """What THIS box's memory system takes to WRITE the bytes of one headline launch (81 920 nodes x 1813 entries x 8 B = 1.19 GB), measured in this run so that
    box-to-box spread of the headline is attributable: `store_only_ms` = a store-only kernel with the node kernel's store instructions, lane layout and band
    interleave (tools/store_ceiling_tiles.hip: band16_nt_64, one wavefront per SIMD), `memset_ms` = hipMemsetAsync over the same bytes."""
    import subprocess
    exe = os.path.join(ROOT, "tools", "_bin", "store_ceiling_tiles")
    if not os.path.exists(exe):
        return {"unavailable": "tools/_bin/store_ceiling_tiles not built (ungar_amd._build.build_tools)"}
    out = {}
    for key, variant in (("store_only_ms", "band16_nt_64"), ("memset_ms", "memset")):
        try:
            text = subprocess.run([exe, variant, "100"], capture_output=True, text=True, timeout=timeout).stdout
            out[key] = float(text.split()[1])
        except Exception as exc:  # noqa: BLE001 -- a context figure: never fails the line
            out[key] = None
            out.setdefault("errors", []).append(f"{variant}: {exc!r}"[:200])
    return out


def free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def gn_chain(torch, ungar_amd, reps=30):
    """BASELINE configs[3], second half: ANYmal node Jacobians (unit-fastest J) -> Gauss-Newton term upper(J^T diag(d) J) (soft_sqp.hpp:257-264 per node), with the
    contraction on the FP64 matrix cores (ungar_gn_hessian_upper_unit_fastest, v_mfma_f64_16x16x4_f64) and on the FP64 vector ALU (ungar_gn_hessian_upper_tiles) side by
    side; HIP events on the launch stream, algorithmic bytes = read x, u; write f, J; read J, d; write the upper triangle."""
    from ungar_amd import workloads as W
    from ungar_amd.sharding import unit_fastest
    rows, cols, N, batch = 37, 49, 20, 4096
    count = N * batch
    m = ungar_amd.NodeModel("anymal")
    x0, u0, _, p = W.synth_device_inputs("anymal", count, 0, torch)
    Op = ungar_amd.Operand
    def operand(elements, src=None):
        t = unit_fastest(elements, count, torch)
        if src is not None:
            t.copy_(src)
        return t
    x, u, f, J = operand(m.nx, x0), operand(m.nu, u0), operand(rows), operand(rows * cols)
    d = operand(rows, torch.rand((rows, count), device="cuda", dtype=torch.float64))
    G_valu = operand(cols * cols)
    G_mfma = torch.zeros((count, cols, cols), dtype=torch.float64, device="cuda")
    es = J.stride(0)
    ops = (count, Op.soa(x, es, N), Op.soa(u, es, N), None, Op.per_instance(p, 1, shared=True), Op.soa(f, es, N), Op.soa(J, es, N))
    jac = lambda: m.dense_jacobian(*ops, knots=N)  # noqa: E731
    valu = lambda: ungar_amd.gn_hessian_tiles(J, d, G_valu, rows, cols, count, True)  # noqa: E731
    mfma = lambda: ungar_amd.gn_hessian_unit_fastest(J, d, G_mfma, rows, cols, count)  # noqa: E731
    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / reps
    upper = cols * (cols + 1) // 2
    jac_bytes, gn_bytes = count * W.algorithmic_bytes(m.nx, m.nu, None), count * 8 * (rows * cols + rows + upper)
    t_end = time.perf_counter() + 0.3  # steady-state clocks, as for the headline measurement
    while time.perf_counter() < t_end:
        for _ in range(20):
            jac()
            valu()
        torch.cuda.synchronize()
    t_jac, t_valu, t_mfma = timeit(jac), timeit(valu), timeit(mfma)
    t_chain_valu, t_chain_mfma = timeit(lambda: (jac(), valu())), timeit(lambda: (jac(), mfma()))
    iu = torch.triu_indices(cols, cols, device="cuda")
    agree = float((G_mfma[:, iu[0], iu[1]].T - G_valu.reshape(cols, cols, count)[iu[0], iu[1]]).abs().max() / G_mfma.abs().max())
    flops = count * 2 * rows * upper  # useful multiply-adds of the upper triangle
    counters, source = None, None
    cpath = os.path.join(ROOT, "profiles", "mfma_counters.json")
    if os.path.exists(cpath):
        with open(cpath) as fh:
            table = json.load(fh)
        counters, source = table.get("gn_hessian_unit_fastest"), table.get("_source")
    def leg(ms, byts):
        return {"ms": ms, "algorithmic_GB": byts / 1e9, "frac_of_8TBs": byts / ms / 1e6 / HBM_PEAK_GBS}
    return {"workload": "anymal node Jacobians -> upper(J^T diag(d) J) per node, N=20 batch=4096 (81920 nodes), unit-fastest J", "reps": reps,
            "node_jacobian": leg(t_jac, jac_bytes),
            "contraction_mfma": {**leg(t_mfma, gn_bytes), "useful_TFLOPs": flops / t_mfma / 1e9, "frac_of_fp64_matrix_peak_78.6": flops / t_mfma / 1e9 / 78.6,
                                 "kernel": "ungar_gn_hessian_upper_unit_fastest (v_mfma_f64_16x16x4_f64)", "mfma_counters": counters, "mfma_counters_source": source},
            "contraction_valu": {**leg(t_valu, gn_bytes), "useful_TFLOPs": flops / t_valu / 1e9, "kernel": "ungar_gn_hessian_upper_tiles (FP64 vector ALU, LDS-DMA row streaming)"},
            "chain_mfma": leg(t_chain_mfma, jac_bytes + gn_bytes), "chain_valu": leg(t_chain_valu, jac_bytes + gn_bytes),
            "max_rel_difference_between_the_two_contractions": agree}


def sqp_iterations(batch=4096, timeout=420):
    """One batched soft-SQP iteration of each reference OCP AS WRITTEN (quadrotor.example.cpp:196-291, rc_car.example.cpp:191-285, quadruped.example.cpp:209-338)
    through the C++20 driver Ungar::BatchedSoftSQPOptimizer: the prebuilt host programs of tests/cpp (build/batched_*_test, `<folder> <batch> 0` = no facade
    comparison) time five iterations after two; their stage functions are compiled at run time on first use (not part of the iteration)."""
    import re
    import subprocess
    import tempfile
    out = {}
    folder = os.environ.get("UNGAR_BENCH_CODEGEN") or os.path.join(tempfile.gettempdir(), "ungar_bench_codegen")  # model cache of the run-time stage functions
    for problem in ("quadrotor", "rc_car", "quadruped"):
        exe = os.path.join(ROOT, "build", f"batched_{problem}_test")
        if not os.path.exists(exe):
            out[problem] = {"skipped": f"{exe} missing (python -c 'import __graft_entry__ as g; g.build()')"}
            continue
        t0 = time.perf_counter()
        try:
            r = subprocess.run([exe, os.path.join(folder, f"batched_{problem}"), str(batch), "0"], capture_output=True, text=True, timeout=timeout)
            m = re.search(r"timing: ([0-9.]+) ms per SQP iteration of (\d+) instances", r.stdout)
            if r.returncode == 0 and m:
                ms = float(m.group(1))
                out[problem] = {"ms_per_iteration": ms, "instances": int(m.group(2)), "instances_per_s": int(m.group(2)) / ms * 1e3, "wall_s_including_jit": time.perf_counter() - t0,
                                "kernel_split": f"profiles/r06z_batched_{problem}_kernel_stats.csv"}
                try:
                    out[problem]["cpu_baseline"] = sqp_cpu_stand_in(problem)
                except Exception as e:  # noqa: BLE001 -- the stand-in must never cost the GPU figure
                    out[problem]["cpu_baseline"] = {"failed": repr(e)[:300]}
            else:
                out[problem] = {"failed": (r.stdout + r.stderr)[-400:]}
        except subprocess.TimeoutExpired:
            out[problem] = {"failed": f"timeout after {timeout} s"}
    return out


def sqp_cpu_stand_in(problem, seconds=3.0):
    """A CPU figure beside `sqp_iterations`: one soft-SQP iteration of ONE instance on one host core -- the reference's execution model (one instance, one thread,
    soft_sqp.hpp:143-158) -- composed of its three dominant parts, each timed here on the box's host:
      N stage Jacobians      the oracle's tape-generated C for the node dynamics (single-instance calls, as in `cpu_baseline`);
      the QP                 the Riccati recursion of the batched SQP compiled for the host (oracle/build_oracle.py: build_qp_host; stage equality rows inside the recursion
                             for the quadruped) -- an EXACT structure-exploiting solve where the reference runs OSQP's ADMM iterations on the whole-horizon KKT system;
      2 x N stage values     the first line-search stage (two candidate steps; the reference evaluates candidates one by one until one is accepted).
    Cost / barrier terms and the assembly of the QP data are left out (quadratic forms: small next to the parts above), so the figure is a LOWER bound of the port's time.
    kind "port": none of this is the reference's own code (CppADCodeGen C + OSQP are absent from the image)."""
    from oracle import build_oracle
    from oracle import ungar_oracle as O
    model, nz, nu, ne, N = {"quadrotor": ("quadrotor", 17, 4, 0, 30), "rc_car": ("rc_car", 8, 2, 0, 30), "quadruped": ("srbd", 25, 24, 16, 30)}[problem]
    lib = ctypes.CDLL(build_oracle.build("portable"))
    nx, nu_m, nw, _ = O.DIMS[model]
    sample = 512
    x, u, w, p = O.synthetic_inputs(model, sample, seed=5)
    w = w if nw else np.zeros((sample, 1))
    dp = ctypes.POINTER(ctypes.c_double)
    ptr = lambda a: a.ctypes.data_as(dp)  # noqa: E731
    nnz = ctypes.c_int.in_dll(lib, f"{model}_jac_nnz").value
    fo, jo = np.zeros((sample, nx)), np.zeros((sample, nnz))

    def rate(fn, args):
        fn.restype = None
        t0 = time.perf_counter()
        fn(*args, 0, sample, 1)
        reps = max(1, int(seconds / 3 / max(time.perf_counter() - t0, 1e-6)))
        t0 = time.perf_counter()
        fn(*args, 0, sample, reps)
        return sample * reps / (time.perf_counter() - t0)

    jac = getattr(lib, f"{model}_sparse_jacobian_batch")
    jac.argtypes = [dp] * 6 + [ctypes.c_long] * 3
    val = getattr(lib, f"{model}_forward_zero_batch")
    val.argtypes = [dp] * 5 + [ctypes.c_long] * 3
    jac_rate, val_rate = rate(jac, (ptr(x), ptr(u), ptr(w), ptr(p), ptr(fo), ptr(jo))), rate(val, (ptr(x), ptr(u), ptr(w), ptr(p), ptr(fo)))
    # the QP of one instance: random well-posed data of the problem's stage sizes
    qp = ctypes.CDLL(build_oracle.build_qp_host())
    rng = np.random.default_rng(5)
    batch, n = 16, nz + nu
    AB = 0.3 * rng.standard_normal((batch, N, nz, n))
    AB[:, :, :, :nz] += np.eye(nz)
    L = rng.standard_normal((batch, N, n, n))
    W = np.ascontiguousarray(np.triu(0.1 * L @ L.transpose(0, 1, 3, 2) + np.eye(n)))
    LN = rng.standard_normal((batch, nz, nz))
    WN = np.ascontiguousarray(np.triu(LN @ LN.transpose(0, 2, 1) + np.eye(nz)))
    b, wv, wN, dx0 = 0.1 * rng.standard_normal((batch, N, nz)), rng.standard_normal((batch, N, n)), rng.standard_normal((batch, nz)), rng.standard_normal((batch, nz))
    dX, dU, st = np.zeros((batch, N + 1, nz)), np.zeros((batch, N, nu)), np.zeros(batch, dtype=np.int32)
    ip = st.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
    if ne:
        E = np.zeros((batch, N, ne, n))
        E[:, :, np.arange(ne), nz + np.arange(ne)] = 1.0  # every row pins one input
        E[:, :, :, :nz] = 0.1 * rng.standard_normal((batch, N, ne, nz))
        ev = 0.1 * rng.standard_normal((batch, N + 1, ne))
        call = lambda: qp.riccati_host_solve_eq(0, nz, nu, ne, N, ctypes.c_longlong(batch), ptr(AB), ptr(b), ptr(W), ptr(wv), ptr(WN), 0, ptr(wN), ptr(dx0), ptr(E), ptr(ev),  # noqa: E731
                                                ctypes.c_double(1e-6), ptr(dX), ptr(dU), ip)
    else:
        call = lambda: qp.riccati_host_solve(nz, nu, N, ctypes.c_longlong(batch), ptr(AB), ptr(b), ptr(W), ptr(wv), ptr(WN), ptr(wN), ptr(dx0), ctypes.c_double(1e-6), ptr(dX),  # noqa: E731
                                             ptr(dU), ip)
    call()
    t0, solves = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds / 3:
        call()
        solves += batch
    qp_s = (time.perf_counter() - t0) / solves
    per_instance = N / jac_rate + qp_s + 2 * N / val_rate
    cores = min(64, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))

    # MEASURED on all cores: one thread per core, each running whole instance-iterations (N stage Jacobians, one QP, 2 N stage values) on buffers of its own for
    # seconds / 3 -- the foreign calls release the GIL, an instance-iteration is three of them
    M = 16  # instances per foreign call (batch of the host QP = 16 above): the Python glue between calls then costs nothing next to them

    def worker(tid, deadline, done):
        rs = np.random.default_rng(100 + tid)
        nodes = M * N
        pick = rs.integers(0, sample, nodes)
        xs, us, ws, ps = (np.ascontiguousarray(a[pick]) for a in (x, u, w, p))
        f1, j1 = np.zeros((nodes, nx)), np.zeros((nodes, nnz))
        AB1, b1, W1, w1, WN1, wN1, d1 = (a.copy() for a in (AB, b, W, wv, WN, wN, dx0))
        dX1, dU1, st1 = np.zeros((M, N + 1, nz)), np.zeros((M, N, nu)), np.zeros(M, dtype=np.int32)
        ip1 = st1.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
        if ne:
            E1, ev1 = E.copy(), ev.copy()
        count = 0
        while time.perf_counter() < deadline:
            jac(ptr(xs), ptr(us), ptr(ws), ptr(ps), ptr(f1), ptr(j1), 0, nodes, 1)
            if ne:
                qp.riccati_host_solve_eq(0, nz, nu, ne, N, ctypes.c_longlong(M), ptr(AB1), ptr(b1), ptr(W1), ptr(w1), ptr(WN1), 0, ptr(wN1), ptr(d1), ptr(E1), ptr(ev1),
                                         ctypes.c_double(1e-6), ptr(dX1), ptr(dU1), ip1)
            else:
                qp.riccati_host_solve(nz, nu, N, ctypes.c_longlong(M), ptr(AB1), ptr(b1), ptr(W1), ptr(w1), ptr(WN1), ptr(wN1), ptr(d1), ctypes.c_double(1e-6), ptr(dX1), ptr(dU1), ip1)
            val(ptr(xs), ptr(us), ptr(ws), ptr(ps), ptr(f1), 0, nodes, 2)
            count += M
        done[tid] = count

    done = [0] * cores
    span = max(0.5, seconds / 3)
    t0 = time.perf_counter()
    threads = [threading.Thread(target=worker, args=(t, t0 + span, done)) for t in range(cores)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    all_rate = sum(done) / (time.perf_counter() - t0)
    return {"ms_per_iteration_of_one_instance": per_instance * 1e3, "instances_per_s": 1.0 / per_instance, "cores": 1, "kind": "port",
            "all_cores": {"instances_per_s": all_rate, "cores": cores, "measured": f"{sum(done)} instance-iterations by {cores} threads in {span:.1f} s ({M} instances per foreign call)"},
            "parts_ms": {"stage_jacobians": N / jac_rate * 1e3, "qp_solve": qp_s * 1e3, "line_search_values": 2 * N / val_rate * 1e3},
            "sample": f"{model} node C (gcc, portable flags) over {sample} seeded nodes; host Riccati {nz} + {nu}" + (f", {ne} equality rows" if ne else "") + f", N = {N}, {solves} solves; "
                      "cost / barrier / assembly not included (lower bound); the reference's own CppADCodeGen C + OSQP are absent from the image"}


def facade_single_instance_latency(timeout=300):
    """BASELINE configs[0]'s execution model -- ONE instance per call from host memory (function.hpp:216-230) -- through the facade's Function::Jacobian on the quadrotor
    node (37 inputs -> 118 values): H2D copy, batch-1 launch, D2H copy, synchronise.  PCIe-inclusive; never part of `value`."""
    import re
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "build", "function_test")
    if not os.path.exists(exe):
        return {"skipped": f"{exe} missing"}
    try:
        r = subprocess.run([exe, os.path.join(os.environ.get("UNGAR_BENCH_CODEGEN") or os.path.join(tempfile.gettempdir(), "ungar_bench_codegen"), "latency"), "latency"], capture_output=True, text=True, timeout=timeout)
        m = re.search(r"HOST_CALL_LATENCY_US ([0-9.]+)", r.stdout)
        if r.returncode == 0 and m:
            us = float(m.group(1))
            return {"us_per_call": us, "evals_per_s": 1e6 / us, "what": "Ungar::Autodiff::Function::Jacobian, quadrotor node, single instance, host memory in and out (PCIe-inclusive); node-sized functions are served by a wavefront resident on the device between calls (include/ungar_amd.h: ungar_function_eval_host)"}
        return {"failed": (r.stdout + r.stderr)[-300:]}
    except subprocess.TimeoutExpired:
        return {"failed": f"timeout after {timeout} s"}


def launcher_command(gpus: int, argv: list[str], port: int | None = None) -> list[str]:
    """The one-process-per-GPU launch of this file: `python bench.py --gpus N` run directly (no WORLD_SIZE in the environment)
    re-executes itself under torch.distributed.run on 127.0.0.1 -- the same command the driver uses."""
    if port is None:
        port = free_port()
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
            os.path.abspath(__file__), *argv]


def reduce_min_max(value: float, dist=None, device=None) -> tuple[float, float]:
    """(min, max) of a per-rank figure over the ranks."""
    if dist is None or not dist.is_initialized():
        return value, value
    import torch
    lo = torch.tensor([value], dtype=torch.float64, device=device)
    hi = lo.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return float(lo.item()), float(hi.item())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="anymal", choices=["anymal", "quadrotor", "rc_car", "srbd"])
    ap.add_argument("--total-batch", type=int, default=None,
                    help="instances of the whole job, partitioned over the ranks (default: the workload's single-GPU batch for --gpus 1, "
                         "65536 = BASELINE config 5 for the anymal workload on --gpus > 1)")
    ap.add_argument("--batch-per-gpu", type=int, default=None, help="weak-scaling variant: this many instances on EVERY rank")
    ap.add_argument("--tile-instances", type=int, default=None,
                    help="instances per tile of the [tile][element][node] device layout (default ungar_amd.sharding.DEFAULT_TILE_INSTANCES = 8192)")
    ap.add_argument("--prewarm-seconds", type=float, default=0.5,
                    help="untimed launches of the same step BEFORE the --warmup steps, so that the timed steps run at steady-state clocks (0 disables)")
    ap.add_argument("--no-stride-pad", action="store_true",
                    help="A/B switch: element stride of the unit-fastest operands = nodes of the tile (consecutive elements a multiple of 2^17 bytes apart "
                         "for these batch sizes: one memory channel per wavefront) instead of ungar_amd.sharding.padded_stride")
    ap.add_argument("--model", default=None, help="kernel variant of the workload's model (e.g. anymal_ad, anymal_reg)")
    ap.add_argument("--layout", default=None, choices=["soa", "tiles"],
                    help="operand layout of the dense block: `tiles` = register images of the wavefronts in bands (default where the model has a tile program: "
                         "the anymal workload's dense block), `soa` = unit-fastest [entry][node]")
    ap.add_argument("--jacobian", default="dense", choices=["dense", "sparse"],
                    help="dense [A|B] block (BASELINE metric, default) or the CSR value array of Function::Jacobian (function.hpp:216-230)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sub-results", action="store_true")
    ap.add_argument("--no-extended-sub-results", action="store_true", help="skip the Gauss-Newton chain, the batched SQP iterations and the single-instance host-call latency")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started directly: become the launcher of one rank per GPU (rank 0 prints the JSON line, which passes through)
        import subprocess
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.call(launcher_command(args.gpus, sys.argv[1:]), env=env))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # UNGAR_BENCH_FORCE_DIST=1: build the process group even for ONE rank, so that RCCL initialisation, the device-tensor all-reduces
    # (MAX / SUM / MIN), the barrier and the teardown can be exercised on a single-GPU box (tests/test_sharding.py, -m gpu)
    force_dist = os.environ.get("UNGAR_BENCH_FORCE_DIST") == "1"
    # the CPU baseline belongs to the single-GPU line (rank 0, N = 1); multi-rank lines say so instead of dropping it silently
    want_cpu = not args.no_cpu_baseline and world == 1 and rank == 0
    from ungar_amd import workloads as W
    model_name = W.WORKLOADS[args.workload][0]
    # compiles while the GPU runs (UNGAR_BENCH_PORTABLE_ORACLE=1: time the prebuilt portable library instead -- the contract test's choice, a native build of the
    # generated C takes a minute of host time; the JSON line says which one was timed)
    native = None  # (started AFTER the headline measurement below: four gcc -O3 jobs on multi-MB files must not share the box with the one driver-timed number)

    import torch
    import ungar_amd
    from ungar_amd.sharding import reduce_sums, reduce_timing, shard_range

    device_index = local_rank % max(1, torch.cuda.device_count())  # one rank per GPU on a real node; the modulo only matters in the
    torch.cuda.set_device(device_index)                           # single-GPU dry run of the multi-rank control flow (gloo) below
    dist = None
    if world > 1 or force_dist:
        import datetime
        import torch.distributed as dist
        backend = os.environ.get("UNGAR_BENCH_BACKEND", "nccl")  # nccl == RCCL on ROCm; "gloo" only for dry runs
        if backend == "nccl" and world > torch.cuda.device_count():
            # two RCCL ranks on one device either fail deep inside the communicator set-up or wait for each other: refuse up front
            raise SystemExit(f"bench.py: the RCCL (nccl) backend needs one GPU per rank -- WORLD_SIZE={world} but {torch.cuda.device_count()} device(s) visible "
                             f"(rank {rank}); use UNGAR_BENCH_BACKEND=gloo for a dry run of the multi-rank control flow on fewer devices")
        if world == 1:  # single forced rank: no launcher set the rendezvous variables
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        limit = datetime.timedelta(seconds=float(os.environ.get("UNGAR_BENCH_DIST_TIMEOUT", "180")))  # a rendezvous that cannot complete fails instead of hanging
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index), timeout=limit)
        else:
            dist.init_process_group(backend, timeout=limit)

    _, N, default_batch = W.WORKLOADS[args.workload]
    if args.batch_per_gpu is not None:  # weak scaling: fixed work per rank
        total, scaling = args.batch_per_gpu * world, "weak"
        begin, end = rank * args.batch_per_gpu, (rank + 1) * args.batch_per_gpu
    else:  # strong scaling: a fixed batch partitioned over the ranks
        total = args.total_batch or (default_batch if world == 1 else (W.CONFIG5_TOTAL_BATCH if args.workload == "anymal" else default_batch * 8))
        scaling = "strong"
        begin, end = shard_range(total, world, rank)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    layout = args.layout or ("tiles" if args.workload == "anymal" and args.jacobian == "dense" and args.model in (None, "anymal") else "soa")
    r = measure(torch, ungar_amd, args.workload, end - begin, total, begin, args.steps, args.warmup, args.jacobian, args.model, fence, tile=args.tile_instances, pad_stride=not args.no_stride_pad, prewarm_s=args.prewarm_seconds, layout=layout)
    # the host compiles the CPU baseline's C with the reference's JIT flags from here on, next to the secondary GPU measurements (UNGAR_BENCH_PORTABLE_ORACLE=1: time
    # the prebuilt portable library instead -- the contract test's choice, a native build of the generated C takes a minute of host time; the JSON line says which)
    if want_cpu and os.environ.get("UNGAR_BENCH_PORTABLE_ORACLE") != "1":
        native = NativeOracleBuild((model_name, "anymal_ad", "quadrotor", "rc_car") if model_name == "anymal" else (model_name,))
    reduce_device = "cuda" if dist is None or dist.get_backend() == "nccl" else "cpu"
    elapsed, total_evals = reduce_timing(r["elapsed"], r["count"] * args.steps, dist, reduce_device)  # MAX time, SUM evals over ranks
    checksum, nodes_per_step = reduce_sums([r["checksum"], float(r["count"])], dist, reduce_device)  # SUM over ranks
    rate_min, rate_max = reduce_min_max(r["count"] * args.steps / r["elapsed"], dist, reduce_device)  # per-rank evals/s
    kernel_min, kernel_max = reduce_min_max(r["kernel_ms"], dist, reduce_device)
    ranks_seen = dist.get_world_size() if dist is not None else 1  # what the process group reports, not what the environment asked for

    if rank == 0:
        traffic = traffic_source = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and args.jacobian == "dense" and r["kernel_model"] == model_name:
            with open(tpath) as fh:
                table = json.load(fh)
            traffic = table.get(f"{args.workload}:{int(r['nodes_per_launch']) // N}" + (":tiles" if layout == "tiles" else ""))  # keyed by the instances of one launch (and the layout)
            traffic_source = table.get("_source") if traffic is not None else None  # NOT measured in this run: see profiles/
        out = {
            "metric": "shooting-node Jacobian evals/sec",
            "value": total_evals / elapsed,
            "unit": "evals/s",
            "n_gpus": world,
            "ranks_seen": ranks_seen,
            "backend": (dist.get_backend() if dist is not None else None),
            "per_rank_evals_per_s": {"min": rate_min, "max": rate_max},
            "per_rank_kernel_ms": {"min": kernel_min, "max": kernel_max},
            "steps": args.steps,
            "warmup": args.warmup,
            "prewarm_s": args.prewarm_seconds,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{args.workload} shooting-node value + {'dense [A|B]' if args.jacobian == 'dense' else 'sparse (nnz=%d)' % r['nnz']} Jacobian, "
                                   f"nx={r['nx']} nu={r['nu']}, N={N}, batch={total} instances over {world} GPU(s) "
                                   f"({end - begin} instances = {r['count']} nodes on rank 0 per step), "
                                   + ("dense block in the wave-tile layout (register images of the wavefronts, bands of 64 tiles: include/ungar_amd.h), x / u / f unit-fastest, " if layout == "tiles" else "unit-fastest (SoA) device layout, ")
                                   + f"one launch per <= {args.tile_instances or 8192} instances, element stride of the unit-fastest operands "
                                   + ("= nodes of the launch" if args.no_stride_pad else "padded off the 2^17-byte channel stride (sharding.padded_stride)"),
                       "horizon": N, "total_batch": total, "batch_rank0": end - begin, "nodes_per_step": int(nodes_per_step), "kernel_variant": r["kernel_model"], "layout": layout,
                       "parallelism": f"instance axis partitioned x{world} (shard_range), no data-path collective"},
            "checksum": checksum,
            "roofline": roofline(r, args.jacobian, traffic, traffic_source),
        }
        if world == 1 and args.workload == "anymal":
            ceilings = box_store_ceilings()
            out["roofline"]["box_store_only_ms"], out["roofline"]["box_memset_ms"] = ceilings.get("store_only_ms"), ceilings.get("memset_ms")
            if "unavailable" in ceilings or "errors" in ceilings:
                out["roofline"]["box_ceilings_note"] = ceilings.get("unavailable") or ceilings.get("errors")
        if world == 1 and not args.no_sub_results and args.workload == "anymal":
            subs = []
            for wl in ("quadrotor", "rc_car"):  # BASELINE.json configs[1], configs[2]
                _, n_sub, b_sub = W.WORKLOADS[wl]
                s = measure(torch, ungar_amd, wl, b_sub, b_sub, 0, max(20, args.steps // 2), max(2, args.warmup // 2), pad_stride=not args.no_stride_pad, prewarm_s=min(args.prewarm_seconds, 0.2))
                rl = roofline(s, "dense")
                subs.append({"workload": f"{wl} N={n_sub} batch={b_sub} dense [A|B]", "value": s["count"] * max(20, args.steps // 2) / s["elapsed"],
                             "unit": "evals/s", "kernel_ms": s["kernel_ms"], "roofline_frac": rl["frac"], "achieved_GBps": rl["achieved"],
                             "algorithmic_bytes_per_eval": s["bytes_per_eval"]})
            out["sub_results"] = subs
            if not args.no_extended_sub_results:
                out["gn_chain"] = gn_chain(torch, ungar_amd)
                out["sqp_iterations"] = sqp_iterations()
                out["single_instance_host_call"] = facade_single_instance_latency()
        if want_cpu:
            out["cpu_baseline"] = cpu_baseline(model_name, native, args.cpu_seconds)
            if args.workload == "anymal" and not args.no_sub_results:  # configs[0..2]: the reference's execution model on the secondary workloads
                out["cpu_baseline"]["sub_results"] = {wl: cpu_baseline(wl, native, min(args.cpu_seconds, 3.0), light=True) for wl in ("quadrotor", "rc_car")}
        elif world > 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = {"skipped": "timed on rank 0 of the single-GPU run only (python bench.py --gpus 1): host cores are shared by the ranks here"}
        line = json.dumps(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line is the LAST line of stdout: native libraries (libdrm under RCCL / the HIP runtime) write notices through C stdio, which is fully
        # buffered on a pipe and would otherwise be flushed after Python's own output, at exit
        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(line, flush=True)


if __name__ == "__main__":
    main()
