"""CPU tests of the tape engine and of the helpers model lambdas call while recording (SURVEY.md section 8(a) rows A10,
A12), run through a TEST-ONLY interpreter of the expression DAG (tests/cpp/tape_interpreter.hpp; the product itself has
no CPU evaluation path):
  build/tape_test     derivative transforms: absolute-zero reverse sweeps below conditionals, forward == reverse
  build/helpers_test  Min / Sign / Abs / SmoothMin / SmoothAbs / Pow / Sqrt / ApproximateNorm values + derivatives vs closed
                      forms; AD-safe quaternion inverse / normalize(d) / slerp (autodiff/support/quaternion.hpp:34-192)
and the two operations the reference forbids on recorded scalars must not compile (support/quaternion.hpp:120-129, 194-222)."""
import os
import subprocess

import pytest


def _build_and_run(repo_root, name, extra=()):
    exe = os.path.join(repo_root, "build", name)
    src = os.path.join(repo_root, "tests", "cpp", f"{name}.cpp")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    inc = os.path.join(repo_root, "ungar_amd", "include")
    deps = [src, os.path.join(repo_root, "tests", "cpp", "tape_interpreter.hpp"), os.path.join(inc, "ungar", "linalg.hpp"),
            os.path.join(inc, "ungar", "utils", "utils.hpp"), os.path.join(repo_root, "ungar_amd", "csrc", "tape", "derive.hpp"),
            os.path.join(repo_root, "ungar_amd", "csrc", "tape", "graph.hpp"), os.path.join(repo_root, "ungar_amd", "csrc", "tape", "scalar.hpp")]
    if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
        subprocess.run(["g++", "-std=c++20", "-O1", "-I", inc, "-I", os.path.join(repo_root, "tests", "cpp"), *extra, "-o", exe, src], check=True, timeout=600)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-3000:]


def test_derivative_transforms_on_the_host_interpreter(repo_root):
    _build_and_run(repo_root, "tape_test")


def test_scalar_and_quaternion_helpers(repo_root):
    _build_and_run(repo_root, "helpers_test")


@pytest.mark.parametrize("snippet,message", [
    ("Ungar::Quaternionad q; q.setFromTwoVectors(Ungar::Vector3ad{}, Ungar::Vector3ad{});", "from two vectors is not implemented"),
    ("Ungar::Quaternionad q; Eigen::DenseMatrix<Ungar::ad_scalar_t> R(3, 3); q = R;", "from rotation matrices with scalar type 'ad_scalar_t' is not implemented"),
    ("Ungar::Vector3ad v; (void)Ungar::Utils::ExponentialMap(v);", "ExponentialMap is not implemented for AD scalars"),
])
def test_operations_unsupported_on_recorded_scalars_do_not_compile(repo_root, tmp_path, snippet, message):
    src = tmp_path / "unsupported.cpp"
    src.write_text('#include "ungar/utils/utils.hpp"\nint main() { ' + snippet + " return 0; }\n")
    r = subprocess.run(["g++", "-std=c++20", "-fsyntax-only", "-I", os.path.join(repo_root, "ungar_amd", "include"), str(src)], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0 and message in r.stderr, r.stderr[-2000:]
    ok = tmp_path / "supported.cpp"  # the same operations on real scalars compile
    ok.write_text('#include "ungar/utils/utils.hpp"\nint main() { ' + snippet.replace("ad_scalar_t", "real_t").replace("ad", "r") + " return 0; }\n")
    r = subprocess.run(["g++", "-std=c++20", "-fsyntax-only", "-I", os.path.join(repo_root, "ungar_amd", "include"), str(ok)], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]


def test_facade_on_the_real_eigen(repo_root):
    """UNGAR_AMD_USE_SYSTEM_EIGEN (ungar/linalg.hpp): the facade's vector / quaternion / sparse types are the REAL Eigen 3.4's --
    what an existing Ungar installation has in every translation unit.  The reference bundles Eigen as a zip; where it was
    present at build time, the helper test and the optimisation-layer test were also built on it and must pass unchanged
    (the reference's own examples built the same way are run by tests/test_reference_examples.py)."""
    eigen = os.path.join(os.environ.get("TMPDIR", "/tmp"), "ungar_amd_reference_eigen", "eigen-3.4.0")
    exe = os.path.join(repo_root, "build", "optimization_test_eigen")
    if not os.path.exists(exe):
        pytest.skip("the real-Eigen build needs the reference's bundled Eigen at build time")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "PASSED" in r.stdout, r.stdout[-2000:]
    if os.path.isdir(os.path.join(eigen, "Eigen")):
        _build_and_run(repo_root, "helpers_test", extra=("-DUNGAR_AMD_USE_SYSTEM_EIGEN", "-I", eigen))
        os.remove(os.path.join(repo_root, "build", "helpers_test"))  # the next run rebuilds the default (built-in algebra) variant


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["1", "2"])  # accumulation mode of the Jacobian: forward, reverse
def test_helpers_on_the_device(repo_root, tmp_path, mode):
    """SURVEY.md section 8(a) row A10 on the DEVICE: Sign, Min, Abs, SmoothAbs, SmoothMin, Pow(x, y), the LOG and POLY relaxed
    barriers, quaternion inverse / normalized / slerp go through Autodiff::MakeFunction -> HIP emission -> hipcc -> gfx950 and their
    value / Jacobian / Hessian kernels are checked against closed forms on either side of every switching point
    (tests/cpp/helpers_device_test.cpp; reference utils.hpp:969-1021, autodiff/support/quaternion.hpp:34-192,
    soft_inequality_constraint.hpp:77-205)."""
    import subprocess
    exe = os.path.join(repo_root, "build", "helpers_device_test")
    assert os.path.exists(exe), "build/helpers_device_test missing: run __graft_entry__.build()"
    r = subprocess.run([exe, str(tmp_path / "codegen")], capture_output=True, text=True, timeout=1200, env={**os.environ, "UNGAR_AMD_JACOBIAN_MODE": mode})
    print(r.stdout[-4000:], r.stderr[-2000:])
    assert r.returncode == 0 and "helpers_device_test OK" in r.stdout
