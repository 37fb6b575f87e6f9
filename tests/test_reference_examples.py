"""The reference's OWN example programs (example/mpc/*.cpp), compiled UNCHANGED from where they lie
against ungar_amd's headers and library (oracle/ref_examples/build_examples.sh -> oracle/_ref/*_example):
the API-surface check of BASELINE.json's north star, and an end-to-end run of the SQP caller (SURVEY.md
section 8(f) N1/N2) on top of the device derivative engine.  The binaries are built where the reference is
present and travel to the GPU box; nothing here reads /root/reference at run time."""
import os
import re
import subprocess

import pytest

LINE = re.compile(r"t = ([\d.]+), obj = ([-\d.e+]+), eqs = ([-\d.e+]+), ineqs = ([-\d.e+]+).*?z ref = ([-\d.]+), z = ([-\d.]+), yaw ref = ([-\d.]+), yaw = ([-\d.]+)")


# "" = built on the facade's own minimal algebra; "_eigen" = the same sources built on the REAL Eigen 3.4 the reference bundles
# (-DUNGAR_AMD_USE_SYSTEM_EIGEN): the facade's types are then Eigen's own, as in an existing Ungar installation.
VARIANTS = ["", "_eigen"]


def _run(repo_root, name, tmp_path, timeout, variant=""):
    exe = os.path.join(repo_root, "oracle", "_ref", f"{name}_example{variant}")
    if not os.path.exists(exe):
        pytest.skip(f"{exe} not built (needs the reference sources at build time)")
    out = subprocess.run([exe], cwd=tmp_path, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("variant", VARIANTS)
def test_quadrotor_example_tracks_its_reference(repo_root, tmp_path, variant):
    """quadrotor.example.cpp PART IV: 10 s of receding-horizon control; after the mission start the
    quadrotor must follow the sinusoidal height reference and the yaw ramp, dynamics constraints closed."""
    rows = [tuple(map(float, m.groups())) for m in map(LINE.search, _run(repo_root, "quadrotor", tmp_path, 1500, variant).splitlines()) if m]
    assert len(rows) >= 290, "one log line per control step is expected"
    late = [r for r in rows if r[0] > 6.0]
    assert max(abs(r[2]) for r in late) < 1e-3, "dynamics equality constraints must be satisfied by the SQP iterates"
    assert max(r[3] for r in late) < 1e-2, "rotor-speed bounds violated"
    # height tracking: the MPC previews the reference, so the error stays a fraction of the 1 m amplitude
    assert max(abs(r[4] - r[5]) for r in late) < 0.15
    assert min(r[5] for r in late) < 3.4 and max(r[5] for r in late) > 4.6, "the quadrotor must actually follow the +-1 m sinusoid"


RC_LINE = re.compile(r"t = ([\d.]+), obj = ([-\d.e+]+), eqs = ([-\d.e+]+), ineqs = ([-\d.e+]+), p ref = \[([-\d.]+), ([-\d.]+)\], p = \[([-\d.]+), ([-\d.]+)\]")


@pytest.mark.gpu
@pytest.mark.parametrize("variant", VARIANTS)
def test_rc_car_example_tracks_its_reference(repo_root, tmp_path, variant):
    """rc_car.example.cpp: the car must converge onto the reference path and stay on it."""
    rows = [tuple(map(float, m.groups())) for m in map(RC_LINE.search, _run(repo_root, "rc_car", tmp_path, 600, variant).splitlines()) if m]
    assert len(rows) >= 290
    late = [r for r in rows if r[0] > 2.0]
    assert max(abs(r[2]) for r in late) < 1e-3
    assert max(r[3] for r in late) < 1e-2
    assert max(abs(r[4] - r[6]) + abs(r[5] - r[7]) for r in late) < 0.05


@pytest.mark.parametrize("variant", VARIANTS)
def test_variable_map_example_runs_on_the_host(repo_root, tmp_path, variant):
    """example/variable_map.example.cpp is pure layout (compile-time static_asserts on indices and on the
    view TYPES returned by VariableMap::Get, then setZero / setOnes / setLinSpaced through the views)."""
    out = " ".join(_run(repo_root, "variable_map", tmp_path, 60, variant).split())  # real Eigen prints a column vector one coefficient per line
    assert "u0 = 0 0 0 0" in out and "u1 = 1 1 1 1" in out and "uN-1 = 2 4 6 8" in out


@pytest.mark.parametrize("variant", VARIANTS)
def test_variable_example_runs_on_the_host(repo_root, tmp_path, variant):
    """example/variable.example.cpp: the variable hierarchy of the single-rigid-body model, ~40 compile-time static_asserts on sizes /
    indices / `At<"name">` lookups, then `ForEach` over the sub-variables of `x`, each logged as a Boost.Hana struct with the reference's
    compact presentation `{:c}` (built against the real Hana the reference bundles)."""
    out = _run(repo_root, "variable", tmp_path, 60, variant)
    for line in ("{ x, 0, 13, 'vector' }", "{ position, 0, 3, 'vector' }", "{ orientation, 3, 4, 'quaternion' }", "{ linear_velocity, 7, 3, 'vector' }",
                 "{ angular_velocity, 10, 3, 'vector' }"):
        assert line in out, out[-1500:]


@pytest.mark.gpu
@pytest.mark.parametrize("variant", VARIANTS)
def test_function_example_self_checks(repo_root, tmp_path, variant):
    """example/autodiff/function.example.cpp asserts TestJacobian / TestHessian (finite differences) itself."""
    _run(repo_root, "function", tmp_path, 600, variant)


QP_LINE = re.compile(r"t = ([\d.]+), obj = ([-\d.e+]+), eqs = ([-\d.e+]+), ineqs = ([-\d.e+]+) \(\d+\), z ref = ([-\d.]+), z = ([-\d.]+), "
                     r"yaw ref = ([-\d.]+), yaw = ([-\d.]+), grf z = ([-\d.]+), ([-\d.]+), ([-\d.]+), ([-\d.]+)")


@pytest.mark.gpu
@pytest.mark.parametrize("variant", VARIANTS)
def test_quadruped_example_trots_and_tracks(repo_root, shared_codegen, variant):
    """quadruped.example.cpp (single-rigid-body quadruped, friction cones, contact schedule): the base must
    keep its height and follow the yaw ramp while exactly one diagonal leg pair carries the weight."""
    import math
    out = _run(repo_root, "quadruped", shared_codegen("quadruped_example"), 1500, variant)  # (one working directory for both algebra variants: the second finds the first one's compiled models)
    rows = [tuple(map(float, m.groups())) for m in map(QP_LINE.search, out.splitlines()) if m]
    assert len(rows) >= 290
    late = [r for r in rows if r[0] > 5.0]
    assert max(abs(r[2]) for r in late) < 0.1 and max(r[3] for r in late) < 1e-2
    assert max(abs(r[4] - r[5]) for r in late) < 0.02
    for r in late:
        d = abs(r[6] - r[7]) % math.pi  # the logged yaw follows Eigen's eulerAngles convention (folded into [0, pi])
        assert min(d, math.pi - d) < 0.05
        swing = [f == 0.0 for f in r[8:12]]
        assert swing in ([False, True, False, True], [True, False, True, False]), "trot: one diagonal pair in swing"
        assert sum(r[8:12]) > 150.0  # the stance pair carries the ~25 kg body


def test_rbd_quantity_example_runs_on_the_host(repo_root, tmp_path):
    """example/rbd/quantity.example.cpp (a user-defined quantity with its own evaluator / getter over pinocchio::forwardKinematics / updateFramePlacement, the feet
    of ANYmal B addressed by frame INDEX 12 / 22 / 32 / 42 and by name: both must give the same pose -- the frame numbering of the reference's model builder):
    builds unchanged on the real Eigen and runs without a GPU."""
    exe = os.path.join(repo_root, "oracle", "_ref", "quantity_example_eigen")
    if not os.path.exists(exe):
        pytest.skip(f"{exe} not built (needs the reference sources at build time)")
    out = subprocess.run([exe], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and all(f"Pose of the foot ('{leg}_FOOT')" in out.stdout for leg in ("LF", "LH", "RF", "RH")), (out.stdout + out.stderr)[-2000:]


@pytest.mark.gpu
def test_rbd_robot_example_checks_itself(repo_root, tmp_path):
    """example/rbd/robot.example.cpp: centre-of-mass acceleration, composite inertia, forward dynamics on the host, then the centroidal momentum as an
    Autodiff::Function (recorded through Robot<ad_scalar_t>, evaluated on the GPU) compared with the host value by the example itself (UNGAR_ASSERT(ok))."""
    exe = os.path.join(repo_root, "oracle", "_ref", "robot_example_eigen")
    if not os.path.exists(exe):
        pytest.skip(f"{exe} not built (needs the reference sources at build time)")
    out = subprocess.run([exe], cwd=tmp_path, capture_output=True, text=True, timeout=900, env={**os.environ, "UNGAR_CODEGEN_FOLDER": str(tmp_path)})
    text = out.stdout + out.stderr
    assert out.returncode == 0 and "Autodiff Jacobian (6 leftmost columns)" in out.stdout and "mismatch" not in text, text[-3000:]
    a = [float(v) for v in out.stdout.split("Autodiff function:")[1].split("\n")[1].split()]
    b = [float(v) for v in out.stdout.split("Ground truth:")[1].split("\n")[1].split()]
    assert len(a) == 6 and max(abs(x - y) for x, y in zip(a, b)) <= 1e-9 * max(1.0, max(abs(y) for y in b))
