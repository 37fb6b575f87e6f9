import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))  # helper modules next to the tests (c_checker.py)
if HERE not in sys.path:
    sys.path.insert(0, HERE)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def repo_root():
    return ROOT


@pytest.fixture(scope="session")
def shared_codegen(tmp_path_factory):
    """One model-cache folder per name for the whole session: C++ test programs that JIT the SAME functions (the three batched-quadruped tests, the two algebra
    variants of an example) compile them once -- the cache is keyed by the tape, the derivative orders, the architecture, the ROCm version and the flags
    (DESIGN section 8 N3), so a program whose tape differs simply compiles its own.  The cold path is still exercised by the first user of every folder, and the
    cache life cycle has its own tests (tests/test_cpp_facade.py)."""
    root = tmp_path_factory.mktemp("model_cache")

    def folder(name):
        path = root / name
        path.mkdir(exist_ok=True)
        return path

    folder.root = root  # (bench.py takes it as UNGAR_BENCH_CODEGEN: its SQP legs run the same programs as tests/test_batched_sqp.py, in <root>/batched_<problem>)
    return folder


@pytest.fixture
def measurement_library():
    """The measurement build of the library for the duration of one test (csrc/runtime/measurement.hpp): the agreement tests between two kernel routes select
    the routes through environment variables that only this build reads.  Yields its path (C++ programs: put its folder first in LD_LIBRARY_PATH)."""
    import ungar_amd
    path = ungar_amd.measurement_library_path()
    assert os.path.exists(path), f"{path} missing: run __graft_entry__.build()"
    ungar_amd.use_library(path)
    try:
        assert ungar_amd.load_library().ungar_measurement_build() == 1
        yield path
    finally:
        ungar_amd.use_library(None)


def measurement_env(extra=None):
    """Environment of a C++ test program that must resolve libungar_amd.so to the measurement build (DT_RUNPATH of the programs comes after LD_LIBRARY_PATH)."""
    import ungar_amd
    folder = os.path.dirname(ungar_amd.measurement_library_path())
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = folder + (os.pathsep + env["LD_LIBRARY_PATH"] if env.get("LD_LIBRARY_PATH") else "")
    env.update(extra or {})
    return env
