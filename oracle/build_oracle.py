"""TEST INFRASTRUCTURE.  Builds oracle/_gen/libnodes_cg.so -- the plain-C checker / CPU baseline.

The C sources are emitted by the product's code generator in its C dialect
(ungar_amd/csrc/tape/emit.hpp) and stand in for the C that CppADCodeGen generates for the reference
(include/ungar/autodiff/function.hpp:468-503): one instance per call, straight-line code.  They are
pinned against the independent torch oracle's golden vectors in tests/test_codegen_c.py.  The
reference JIT-compiles with `-O3 -g -march=native -mtune=native -ffast-math` (function.hpp:610-611);
this prebuilt library uses `-march=x86-64-v3` instead of native because it is built in one
container and executed on another host (bench.py rebuilds with -march=native on the box it times).
"""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
GEN = os.path.join(HERE, "_gen")
MODELS = ("quadrotor", "rc_car", "srbd", "anymal", "anymal_ad")
REFERENCE_FLAGS = ["-O3", "-g", "-march=native", "-mtune=native", "-ffast-math"]  # function.hpp:610-611
PORTABLE_FLAGS = ["-O3", "-march=x86-64-v3", "-ffast-math"]


def lib_path(tag: str = "portable") -> str:
    return os.path.join(GEN, f"libnodes_cg_{tag}.so")


def build(tag: str = "portable", flags=None, models=MODELS, force: bool = False) -> str:
    flags = list(flags if flags is not None else (REFERENCE_FLAGS if tag == "native" else PORTABLE_FLAGS))
    srcs = [os.path.join(GEN, f"{m}_cg.c") for m in models]
    out = lib_path(tag)
    if not force and os.path.exists(out) and all(os.path.getmtime(s) <= os.path.getmtime(out) for s in srcs):
        return out
    objs = []
    procs = []
    for s in srcs:
        o = s[:-2] + f".{tag}.o"
        objs.append(o)
        procs.append(subprocess.Popen(["gcc", *flags, "-fPIC", "-c", s, "-o", o]))
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError("gcc failed building the oracle C checker")
    subprocess.run(["gcc", "-shared", "-o", out, *objs, "-lm"], check=True)
    return out


if __name__ == "__main__":
    print(build())
