"""Batched soft SQP from C++ on the reference's OCPs AS WRITTEN.

`build/batched_{quadrotor,rc_car,quadruped}_test` (tests/cpp/batched_*_test.cpp) are plain C++20 host programs over
the C ABI.  Each states one of the reference's THREE MPC problems twice through the facade -- whole-horizon, as
example/mpc/quadrotor.example.cpp:196-291 / rc_car.example.cpp:191-285 / quadruped.example.cpp:209-338 do (523 / 246 / 1123 decision
variables, input-rate coupling, terminal tracking, 480 foot-contact equality rows), and in stage form with the cross-knot quantity carried in the stage state -- and advances
  * >= 1024 perturbed instances (random references, measured states, rotor bounds active or not; random gaits for the quadruped) with
    Ungar::BatchedSoftSQPOptimizer: stage derivatives, QP assembly, Riccati recursion with stage equality rows, stacked backtracking
    search, all on the device;
  * a sample of them with the facade's Ungar::SoftSQPOptimizer: sparse L D L^T of the whole KKT system of the QP the reference hands
    to OSQP (soft_sqp.hpp:143-158) and the host line search (backtracking_line_search.hpp:116-151).
The search direction, the accepted step size and the iterate must agree to 1e-9 after the first AND after the second iteration."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("problem,batch,compared", [("quadrotor", 1024, 8), ("rc_car", 1024, 8), ("quadruped", 1024, 8)])
def test_batched_sqp_equals_the_whole_horizon_facade(repo_root, tmp_path, problem, batch, compared):
    exe = os.path.join(repo_root, "build", f"batched_{problem}_test")
    assert os.path.exists(exe), f"{exe} missing: run __graft_entry__.build()"
    r = subprocess.run([exe, str(tmp_path / "codegen"), str(batch), str(compared)], capture_output=True, text=True, timeout=1500)
    print(r.stdout[-6000:], r.stderr[-2000:])
    assert r.returncode == 0 and f"PASS batched {problem} SQP (batch {batch}, {compared} compared)" in r.stdout
    lines = [l for l in r.stdout.splitlines() if l.startswith("iteration") and "instances accepted" in l]
    assert len(lines) == 2
    for line in lines:
        moved, total = map(int, re.search(r"(\d+) of (\d+) instances accepted", line).groups())
        assert total == batch and moved >= batch // 2  # the comparison is not vacuous: steps are taken


def test_assembly_kernel_sections_give_the_same_bits(repo_root, tmp_path):
    """The shooting assembly kernel runs stage nodes with few equality rows (the quadruped: 16 x 50 tableau) through wavefront-specialised
    sections -- Gauss-Jordan elimination in the registers of one wavefront, barrier terms without workgroup barriers -- and
    everything else, or everything under UNGAR_AMD_ASSEMBLE_GENERIC=1, through the generic sections.  Same pivot rule, same arithmetic:
    the assembled QP data (AB, b, W, w, reduced equality rows and residuals, printed with 17 digits) and both steps of the compared
    instances must agree to the last bit, over two SQP iterations."""
    exe = os.path.join(repo_root, "build", "batched_quadruped_test")
    assert os.path.exists(exe), f"{exe} missing: run __graft_entry__.build()"
    dumps = {}
    for mode in ("specialised", "generic"):
        env = dict(os.environ)
        env.pop("UNGAR_AMD_ASSEMBLE_GENERIC", None)
        if mode == "generic":
            env["UNGAR_AMD_ASSEMBLE_GENERIC"] = "1"
        folder = tmp_path / mode
        folder.mkdir()
        r = subprocess.run([exe, str(tmp_path / "codegen"), "256", "3", str(folder)], capture_output=True, text=True, timeout=1500, env=env)
        print(r.stdout[-2000:], r.stderr[-1000:])
        assert r.returncode == 0 and "PASS batched quadruped SQP (batch 256, 3 compared)" in r.stdout
        dumps[mode] = {f.name: f.read_bytes() for f in sorted(folder.iterdir())}
    assert len(dumps["specialised"]) >= 6 and dumps["specialised"].keys() == dumps["generic"].keys()  # 3 instances x 2 iterations
    for name, data in dumps["specialised"].items():
        assert len(data) > 100_000 and data == dumps["generic"][name], f"{name}: the two code paths disagree"
