"""bench.py's output contract (the driver parses this line): one JSON object on the last stdout line with the metric of BASELINE.json,
whole-job value, exact step / warm-up echo, `roofline` (HBM bound, achieved = algorithmic bytes / measured launch time, frac = achieved / peak)
and, on one GPU, `cpu_baseline`.  Run with the driver's own arguments (`--gpus 1 --steps 20 --warmup 5`)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_bench_line_has_the_contract_fields(repo_root, shared_codegen):
    # (UNGAR_BENCH_PORTABLE_ORACLE skips the minute-long native build of the CPU baseline; UNGAR_BENCH_CODEGEN: the SQP legs find the stage functions the batched
    # SQP tests of this session compiled -- a cold bench run compiles them itself, ~60 s of its ~130 s)
    out = subprocess.run([sys.executable, os.path.join(repo_root, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--cpu-seconds", "2"], cwd=repo_root,
                         capture_output=True, text=True, timeout=900, env={**os.environ, "UNGAR_BENCH_PORTABLE_ORACLE": "1", "UNGAR_BENCH_CODEGEN": str(shared_codegen.root)})
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["metric"].startswith("shooting-node Jacobian evals/sec") and d["unit"] == "evals/s" and d["higher_is_better"] is True
    assert (d["n_gpus"], d["steps"], d["warmup"]) == (1, 20, 5) and d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["scaling"] in ("strong", "weak") and "workload" in d["config"] and "model" not in d["config"]
    assert d["config"]["total_batch"] == 4096 and d["config"]["nodes_per_step"] == 4096 * 20
    # value is the whole-job rate over the timed region; ms_per_step is that region / steps
    assert abs(d["value"] - d["config"]["nodes_per_step"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["algorithmic_bytes_per_eval"] == 15192
    assert abs(r["achieved"] - r["nodes_per_launch"] * 15192 / (r["kernel_ms"] * 1e-3) / 1e9) <= 1e-6 * r["achieved"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.3 < r["frac"] < 1.0
    assert r["kernel_ms"] <= d["ms_per_step"] * 1.001, "the kernel cannot take longer than the step that contains it"
    # the headline launch writes the dense block in the wave-tile layout, and the line carries the box's own store ceilings measured in the same run
    assert d["config"]["layout"] == "tiles" and "wave-tile" in r["kernel"]
    assert 0.1 < r["box_memset_ms"] < r["box_store_only_ms"] < r["kernel_ms"] * 1.05, (r["box_memset_ms"], r["box_store_only_ms"], r["kernel_ms"])
    assert r["traffic"] is None or (0.9 < r["traffic"] / (r["nodes_per_launch"] * 15192) < 1.2 and "profiles/" in r["traffic_source"])
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] == 1 and c["value"] > 1e4 and "sample" in c
    assert d["value"] / c["value"] > 100, "one GPU against one CPU core"
    assert abs(d["checksum"] - 5033491.53798481) < 1e-3, "synthetic inputs and results are deterministic"
    assert len(d["sub_results"]) == 2 and all(0.3 < s["roofline_frac"] < 1.0 for s in d["sub_results"])
    # BASELINE configs[3], second half: node Jacobians -> upper(J^T diag(d) J), matrix-core and vector contraction side by side (soft_sqp.hpp:257-264)
    g = d["gn_chain"]
    for leg in ("node_jacobian", "contraction_mfma", "contraction_valu", "chain_mfma", "chain_valu"):
        assert g[leg]["ms"] > 0 and 0.05 < g[leg]["frac_of_8TBs"] < 1.0 and g[leg]["algorithmic_GB"] > 0.5
    assert g["max_rel_difference_between_the_two_contractions"] < 1e-12 and 0.0 < g["contraction_mfma"]["frac_of_fp64_matrix_peak_78.6"] < 1.0
    assert g["contraction_mfma"]["mfma_counters"] is None or (g["contraction_mfma"]["mfma_counters"]["cycles_per_mfma"] == 64.0 and "profiles/" in g["contraction_mfma"]["mfma_counters_source"])
    # one batched SQP iteration of each reference OCP as written, through the C++ driver
    q = d["sqp_iterations"]
    assert set(q) == {"quadrotor", "rc_car", "quadruped"} and all(0.05 < q[k]["ms_per_iteration"] < 100.0 and q[k]["instances"] == 4096 for k in q), q
    assert all(q[k]["cpu_baseline"]["all_cores"]["instances_per_s"] > 0 and "threads" in q[k]["cpu_baseline"]["all_cores"]["measured"] for k in q), "the all-core CPU figure is measured"
    # BASELINE configs[0]: the reference's execution model (one instance per call) through the facade, and its CPU stand-ins for the secondary workloads
    assert 1.0 < d["single_instance_host_call"]["us_per_call"] < 5000.0
    assert set(c["sub_results"]) == {"quadrotor", "rc_car"} and all(v["cores"] == 1 and v["value"] > 1e5 for v in c["sub_results"].values())
