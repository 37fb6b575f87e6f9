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
