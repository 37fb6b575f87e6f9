"""Build driver for the ungar_amd native pieces (used by __graft_entry__.build()).

Steps (each skipped when its outputs are newer than its inputs):
  1. g++   csrc/codegen/codegen_main.cpp                 -> build/ungar_codegen
  2. run   ungar_codegen                                 -> csrc/gen/<model>_gen.hpp (+ oracle/_gen/<model>_cg.c)
  3. hipcc csrc/kernels/*.hip, csrc/runtime/c_api.cpp    -> ungar_amd/lib/libungar_amd.so   (gfx950)
The oracle's C checker / CPU baseline is built by oracle/build_oracle.py, not here.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ungar_amd", "csrc")
GEN = os.path.join(CSRC, "gen")
BUILD = os.path.join(ROOT, "build")
LIBDIR = os.path.join(ROOT, "ungar_amd", "lib")
LIB = os.path.join(LIBDIR, "libungar_amd.so")
MEASUREMENT_LIB = os.path.join(LIBDIR, "measurement", "libungar_amd.so")
ORACLE_GEN = os.path.join(ROOT, "oracle", "_gen")
RBD_MODELS = ("anymal_rnea", "anymal_crba", "anymal_minv", "anymal_feet", "anymal_centroidal")
MODELS = ("quadrotor", "rc_car", "srbd", "srbd_ineq", "quadrotor_ineq", "rc_car_ineq", "srbd_feet", "anymal", "anymal_ad", "anymal_reg") + RBD_MODELS
C_MODELS = ("quadrotor", "rc_car", "srbd", "anymal", "anymal_ad") + RBD_MODELS
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-Wno-unused-result"]


def _run(cmd, **kw):
    print("[ungar_amd build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, **kw)


def _newer(outputs, inputs) -> bool:
    """True when every output exists and is newer than every input."""
    try:
        out_t = min(os.path.getmtime(o) for o in outputs)
    except OSError:
        return False
    return all(os.path.getmtime(i) <= out_t for i in inputs)


def _tree(*dirs):
    files = []
    for d in dirs:
        for base, _, names in os.walk(d):
            if os.path.basename(base) == "gen":
                continue
            files += [os.path.join(base, n) for n in names if n.endswith((".hpp", ".cpp", ".hip", ".h"))]
    return files


def build_codegen() -> str:
    exe = os.path.join(BUILD, "ungar_codegen")
    srcs = _tree(os.path.join(CSRC, "tape"), os.path.join(CSRC, "models"), os.path.join(CSRC, "rbd"), os.path.join(CSRC, "codegen"))
    if not _newer([exe], srcs):
        os.makedirs(BUILD, exist_ok=True)
        _run(["g++", "-std=c++20", "-O2", "-DUNGAR_AMD_MEASUREMENT", "-o", exe, os.path.join(CSRC, "codegen", "codegen_main.cpp")])  # (a build tool: its emitter diagnostics stay selectable)
    return exe


def generate(exe: str):
    """Runs the code generator into a scratch directory and replaces only the files whose CONTENT changed, so that an
    edit to the generator recompiles just the kernels it affects (a from-scratch library build takes ~6 minutes)."""
    import filecmp
    import shutil
    outs = [os.path.join(GEN, f"{m}_gen.hpp") for m in MODELS + ("quadrotor_cost", "srbd_cost", "rc_car_cost", "anymal_cost", "anymal_quad", "anymal_tiles", "anymal_split", "anymal_rnea_quad", "anymal_crba_quad", "anymal_centroidal_quad")] + [os.path.join(ORACLE_GEN, f"{m}_cg.c") for m in C_MODELS]
    robot = os.path.join(ROOT, "ungar_amd", "data", "anymal_b.robot")
    stamp = os.path.join(BUILD, "codegen.stamp")
    if _newer(outs + [stamp], [exe, robot]):
        return
    tmp_gen, tmp_c = os.path.join(BUILD, "gen_tmp"), os.path.join(BUILD, "gen_tmp_c")
    for d in (tmp_gen, tmp_c):
        shutil.rmtree(d, ignore_errors=True)
        os.makedirs(d)
    os.makedirs(GEN, exist_ok=True)
    os.makedirs(ORACLE_GEN, exist_ok=True)
    # (--quad-pair-stores: the sinks of the lane-per-leg ANYmal program carry two entries of a column -- one 16-byte store in the kernel with paired stores, quad_kernel.hpp: PAIR)
    _run([exe, "--out", tmp_gen, "--c-oracle", tmp_c, "--anymal-robot", robot, "--lds-slots", "320", "--quad-pair-stores", "1"])
    for src_dir, dst_dir in ((tmp_gen, GEN), (tmp_c, ORACLE_GEN)):
        for name in sorted(os.listdir(src_dir)):
            src, dst = os.path.join(src_dir, name), os.path.join(dst_dir, name)
            if not os.path.exists(dst) or not filecmp.cmp(src, dst, shallow=False):
                print("[ungar_amd build] generated file changed:", os.path.relpath(dst, ROOT), flush=True)
                shutil.move(src, dst)
    with open(stamp, "w") as fh:
        fh.write("generated files are up to date with the code generator\n")


def build_library(jobs: int | None = None):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(BUILD, exist_ok=True)
    kernel_hdr = os.path.join(CSRC, "kernels", "node_kernel.hpp")
    abi_hdr = os.path.join(ROOT, "include", "ungar_amd.h")
    quad_deps = [os.path.join(CSRC, "kernels", "quad_kernel.hpp"), os.path.join(GEN, "anymal_quad_gen.hpp")]
    units = []  # (source, object, dependencies)
    for m in MODELS:
        src = os.path.join(CSRC, "kernels", f"model_{m}.hip")
        deps = [src, kernel_hdr, os.path.join(GEN, f"{m}_gen.hpp")]
        if m == "anymal":
            deps += quad_deps
        if m in ("anymal_rnea", "anymal_crba", "anymal_centroidal"):  # lane-per-leg programs of the joint torques / the inertia matrix / the centroidal momentum
            skeleton = "quad_crba_kernel.hpp" if m == "anymal_crba" else "quad_rnea_kernel.hpp"
            deps += [os.path.join(CSRC, "kernels", "quad_kernel.hpp"), os.path.join(CSRC, "kernels", skeleton), os.path.join(GEN, f"{m}_quad_gen.hpp")]
        units.append((src, os.path.join(BUILD, f"model_{m}.o"), deps))
    for name in sorted(os.listdir(os.path.join(CSRC, "kernels"))):
        if name.endswith(".hip") and not name.startswith("model_"):
            src = os.path.join(CSRC, "kernels", name)
            sqp_hdrs = [os.path.join(CSRC, "kernels", h) for h in ("ocp_sqp.hpp", "ocp_riccati.hpp", "ocp_barrier.hpp", "ocp_shooting.hpp")]
            extra = (sqp_hdrs + [os.path.join(CSRC, "kernels", "ocp_riccati_wave_kernel.hpp"), os.path.join(CSRC, "runtime", "kernel_jit.hpp")] if name.startswith(("ocp_riccati", "ocp_shooting")) else
                     [os.path.join(CSRC, "kernels", "ocp_assembly.hpp")] if name.startswith("ocp_assembly") else
                     quad_deps + [os.path.join(CSRC, "kernels", "quad_tile_kernel.hpp"), os.path.join(GEN, "anymal_tiles_gen.hpp")] if name == "quad_anymal_tiles.hip" else
                     quad_deps if name.startswith("quad_") else
                     [os.path.join(GEN, f"{name[5:-4]}_cost_gen.hpp"), os.path.join(CSRC, "kernels", "cost_kernel.hpp")] if name.startswith("cost_") else [])
            units.append((src, os.path.join(BUILD, name[:-4] + ".o"), [src, kernel_hdr] + extra))
    src = os.path.join(CSRC, "runtime", "c_api.cpp")
    units.append((src, os.path.join(BUILD, "c_api.o"), [src, kernel_hdr, abi_hdr, os.path.join(CSRC, "kernels", "ocp_assembly.hpp")]))
    src = os.path.join(CSRC, "runtime", "c_api_sqp.cpp")
    units.append((src, os.path.join(BUILD, "c_api_sqp.o"), [src, abi_hdr] + [os.path.join(CSRC, "kernels", h) for h in ("ocp_sqp.hpp", "ocp_riccati.hpp", "ocp_shooting.hpp")]))
    src = os.path.join(CSRC, "runtime", "function.cpp")
    units.append((src, os.path.join(BUILD, "function.o"), [src, abi_hdr, os.path.join(CSRC, "runtime", "jit_common.hpp")] + _tree(os.path.join(CSRC, "tape"))))
    src = os.path.join(CSRC, "runtime", "kernel_jit.cpp")
    units.append((src, os.path.join(BUILD, "kernel_jit.o"), [src, abi_hdr, os.path.join(CSRC, "runtime", "jit_common.hpp"), os.path.join(CSRC, "runtime", "kernel_jit.hpp")]))

    measurement_hdr = os.path.join(CSRC, "runtime", "measurement.hpp")
    units = [(s_, o_, d_ + [measurement_hdr]) for s_, o_, d_ in units]
    # the two comparison kernels (taped ABA / structured, lane per node: 40-60 k statements in one basic block)
    # spend > 90 % of their compile time in the machine schedulers; without them the from-scratch build drops
    # from 12.7 to ~6 minutes.  The product kernels keep the full pipeline.
    fast = ("model_anymal_ad.hip", "model_anymal_reg.hip")
    no_sched = ["-mllvm", "-enable-misched=false", "-mllvm", "-enable-post-misched=false"]

    # identity of the tape engine (recorder, derivative transforms, emitters, run-time factory): part of the key of
    # every run-time cache entry, so that editing any of these sources invalidates the entries they produced
    import hashlib
    h = hashlib.sha1()
    for f in sorted(_tree(os.path.join(CSRC, "tape")) + [os.path.join(CSRC, "runtime", "function.cpp")]):
        h.update(open(f, "rb").read())
    emitter_id = h.hexdigest()[:16]

    def compile_unit(u):
        src, obj, deps = u
        if not _newer([obj], deps):
            extra = [f'-DUNGAR_AMD_EMITTER_ID="{emitter_id}"'] if os.path.basename(src) == "function.cpp" else []
            _run(["hipcc", *HIPCC_FLAGS, *extra, *(no_sched if os.path.basename(src) in fast else []), "-c", src, "-o", obj])
        return obj

    # Measurement build (csrc/runtime/measurement.hpp): the translation units that read a measurement switch are compiled a second time with
    # -DUNGAR_AMD_MEASUREMENT; everything else is shared between the two libraries.  ungar_amd/lib/measurement/libungar_amd.so has the same
    # ABI and is what tools/ and the agreement tests between two kernel routes load; the shipped library contains none of the switches.
    measurement_dir = os.path.join(BUILD, "measurement")
    os.makedirs(measurement_dir, exist_ok=True)
    os.makedirs(os.path.dirname(MEASUREMENT_LIB), exist_ok=True)

    def reads_switches(src):
        if os.path.basename(src) in ("function.cpp", "c_api.cpp"):  # (function.cpp through tape/emit.hpp; c_api.cpp reports the build flavour)
            return True
        text = open(src).read()
        return "UNGAR_MEASUREMENT_SWITCH" in text or "UNGAR_AMD_MEASUREMENT_BUILD" in text

    measurement_units = [(src, os.path.join(measurement_dir, os.path.basename(obj)), deps) for src, obj, deps in units if reads_switches(src)]

    def compile_measurement_unit(u):
        src, obj, deps = u
        if not _newer([obj], deps):
            extra = [f'-DUNGAR_AMD_EMITTER_ID="{emitter_id}"'] if os.path.basename(src) == "function.cpp" else []
            _run(["hipcc", *HIPCC_FLAGS, "-DUNGAR_AMD_MEASUREMENT", *extra, "-c", src, "-o", obj])
        return obj

    with ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as pool:
        futures = [pool.submit(compile_unit, u) for u in units] + [pool.submit(compile_measurement_unit, u) for u in measurement_units]
        results = [f.result() for f in futures]
    objs, measurement_objs = results[:len(units)], results[len(units):]
    if not _newer([LIB], objs):
        _run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    replaced = {os.path.basename(o) for o in measurement_objs}
    all_measurement = [o for o in objs if os.path.basename(o) not in replaced] + measurement_objs
    if not _newer([MEASUREMENT_LIB], all_measurement):
        _run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", MEASUREMENT_LIB, *all_measurement])
    return LIB


def build_cpp_tests():
    """Host-side C++ programs exercising the facade headers (ungar_amd/include/ungar): run by
    tests/test_layout.py (CPU) and tests/test_cpp_facade.py (GPU box, prebuilt binaries travel)."""
    inc = os.path.join(ROOT, "ungar_amd", "include")
    hdrs = _tree(inc, os.path.join(CSRC, "tape"), os.path.join(CSRC, "rbd")) + [os.path.join(ROOT, "include", "ungar_amd.h")]
    jobs = []
    for name, link in (("layout_dump", False), ("function_test", True), ("quadrotor_ocp_test", True), ("rbd_test", True), ("optimization_test", False),
                       ("batched_quadrotor_test", True), ("batched_quadruped_test", True), ("batched_rc_car_test", True), ("helpers_device_test", True)):
        src = os.path.join(ROOT, "tests", "cpp", f"{name}.cpp")
        exe = os.path.join(BUILD, name)
        if _newer([exe], [src, *hdrs] + ([LIB] if link else [])):
            continue
        cmd = ["g++", "-std=c++20", "-O1", "-I", inc, "-o", exe, src]
        if link:
            cmd += ["-L", LIBDIR, "-lungar_amd", "-Wl,-rpath,$ORIGIN/../ungar_amd/lib", "-Wl,-rpath,/opt/rocm/lib"]
        jobs.append(cmd)
    # a user's OCP of stage sizes the library holds no prebuilt solver kernels for (tests/test_batched_sqp.py): two instantiations of one source
    user_src = os.path.join(ROOT, "tests", "cpp", "batched_user_ocp_test.cpp")
    for nx, nu, ne, parameters in ((10, 3, 0, 1), (20, 9, 4, 1), (10, 3, 2, 0)):  # (the last one: constant problem data -- a ShootingProblem without knot / instance parameters)
        exe = os.path.join(BUILD, f"batched_user_ocp_test_{nx}_{nu}_{ne}" + ("" if parameters else "_const"))
        if not _newer([exe], [user_src, *hdrs, LIB]):
            jobs.append(["g++", "-std=c++20", "-O1", f"-DUSER_NX={nx}", f"-DUSER_NU={nu}", f"-DUSER_NE={ne}", f"-DUSER_PARAMETERS={parameters}", "-I", inc, "-o", exe, user_src, "-L", LIBDIR,
                         "-lungar_amd", "-Wl,-rpath,$ORIGIN/../ungar_amd/lib", "-Wl,-rpath,/opt/rocm/lib"])
    # reference-side adapter of INTEGRATION.md section 2 (include/ungar_amd_model.hpp): plain C++17, no facade headers
    amd_src, amd_exe = os.path.join(ROOT, "tests", "cpp", "amd_model_test.cpp"), os.path.join(BUILD, "amd_model_test")
    if not _newer([amd_exe], [amd_src, os.path.join(ROOT, "include", "ungar_amd_model.hpp"), os.path.join(ROOT, "include", "ungar_amd.h"), LIB]):
        jobs.append(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-o", amd_exe, amd_src, "-L", LIBDIR, "-lungar_amd", "-Wl,-rpath,$ORIGIN/../ungar_amd/lib",
                     "-Wl,-rpath,/opt/rocm/lib"])
    # the same facade tests on the REAL Eigen 3.4 bundled with the reference (UNGAR_AMD_USE_SYSTEM_EIGEN), where it is present
    ref = os.environ.get("UNGAR_REFERENCE", "/root/reference")
    eigen_zip = os.path.join(ref, "external", "config", "eigen", "eigen-3.4.0.zip")
    if os.path.exists(eigen_zip):
        scratch = os.path.join(os.environ.get("TMPDIR", "/tmp"), "ungar_amd_reference_eigen")
        if not os.path.isdir(os.path.join(scratch, "eigen-3.4.0", "Eigen")):
            os.makedirs(scratch, exist_ok=True)
            _run(["unzip", "-q", "-o", eigen_zip, "-d", scratch])
        for name, link in (("function_test", True), ("quadrotor_ocp_test", True), ("optimization_test", False)):
            src = os.path.join(ROOT, "tests", "cpp", f"{name}.cpp")
            exe = os.path.join(BUILD, name + "_eigen")
            if _newer([exe], [src, *hdrs] + ([LIB] if link else [])):
                continue
            cmd = ["g++", "-std=c++20", "-O1", "-DUNGAR_AMD_USE_SYSTEM_EIGEN", "-I", os.path.join(scratch, "eigen-3.4.0"), "-I", inc, "-o", exe, src]
            if link:
                cmd += ["-L", LIBDIR, "-lungar_amd", "-Wl,-rpath,$ORIGIN/../ungar_amd/lib", "-Wl,-rpath,/opt/rocm/lib"]
            jobs.append(cmd)
    # 4-lane CPU simulator of the lane-per-leg program (tests/test_quad_program.py)
    sim_src, sim_lib = os.path.join(ROOT, "tests", "cpp", "quad_sim.cpp"), os.path.join(BUILD, "libquad_sim.so")
    quad_gen = os.path.join(GEN, "anymal_quad_gen.hpp")
    if os.path.exists(quad_gen) and not _newer([sim_lib], [sim_src, quad_gen]):
        jobs.append(["g++", "-std=c++20", "-O0", "-shared", "-fPIC", "-I", GEN, "-o", sim_lib, sim_src])
    with ThreadPoolExecutor(max_workers=4) as pool:
        list(pool.map(_run, jobs))
    build_reference_examples()
    build_reference_tests()


def build_reference_examples(names=("quadrotor", "rc_car", "quadruped", "function", "variable_map", "variable", "robot", "quantity")):
    """The reference's own example/mpc programs against ungar_amd's headers (oracle/ref_examples):
    only where the reference is present; the binaries land in oracle/_ref and travel to the GPU box."""
    script = os.path.join(ROOT, "oracle", "ref_examples", "build_examples.sh")
    ref = os.environ.get("UNGAR_REFERENCE", "/root/reference")
    todo = []
    for n in names:
        cands = [os.path.join(ref, "example", sub, f"{n}.example.cpp") for sub in ("mpc", "autodiff", "rbd", "")]
        src = next((c for c in cands if os.path.exists(c)), cands[0])
        exes = [os.path.join(ROOT, "oracle", "_ref", f"{n}_example")]
        if os.path.exists(os.path.join(ref, "external", "config", "eigen", "eigen-3.4.0.zip")):
            exes.append(exes[0] + "_eigen")  # second build on the real Eigen the reference bundles
            if os.sep + "rbd" + os.sep in src:
                exes = exes[1:]  # the rbd examples use Eigen expression forms the built-in algebra does not have: real Eigen only
        elif os.sep + "rbd" + os.sep in src:
            continue
        if os.path.exists(src) and not _newer(exes, [src, LIB, script] + _tree(os.path.join(ROOT, "ungar_amd", "include"))):
            todo.append(n)
    if todo:
        _run(["bash", script, *todo])


def build_reference_tests():
    """The reference's own unit tests (variable / function / soft_sqp) compiled unchanged against the facade
    (oracle/ref_tests): only where the reference is present; binaries land in oracle/_ref and travel to the GPU box."""
    script = os.path.join(ROOT, "oracle", "ref_tests", "build_ref_tests.sh")
    ref = os.environ.get("UNGAR_REFERENCE", "/root/reference")
    srcs = [os.path.join(ref, "test", *rel) for rel in (("variable.test.cpp",), ("autodiff", "function.test.cpp"), ("optimization", "soft_sqp.test.cpp"), ("rbd", "robot.test.cpp"))]
    if not all(os.path.exists(f) for f in srcs):
        return
    exes = [os.path.join(ROOT, "oracle", "_ref", "ref_variable_test")]
    if os.path.exists(os.path.join(ref, "external", "config", "eigen", "eigen-3.4.0.zip")):
        exes += [os.path.join(ROOT, "oracle", "_ref", f"ref_{n}_test_eigen") for n in ("variable", "function", "soft_sqp", "robot")]
        if os.path.exists(os.path.join(ref, "external", "config", "hana", "hana-boost-1.84.0.zip")) and os.path.exists(os.path.join(ref, "test", "utils", "utils.test.cpp")):
            exes.append(os.path.join(ROOT, "oracle", "_ref", "ref_utils_test_eigen"))
            srcs.append(os.path.join(ref, "test", "utils", "utils.test.cpp"))
    shim = os.path.join(ROOT, "tests", "gtest_shim", "gtest", "gtest.h")
    if not _newer(exes, srcs + [LIB, script, shim, os.path.join(ROOT, "ungar_amd", "data", "anymal_b.robot"), os.path.join(ROOT, "tools", "robot_to_urdf.py")] + _tree(os.path.join(ROOT, "ungar_amd", "include"))):
        _run(["bash", script])


def build_tools():
    """Standalone measurement harnesses of tools/ (not product code): FP64 issue-rate microbenchmarks and the ablation harness of
    the Gauss-Newton tiles kernel.  Built into tools/_bin (git-ignored; travels to the GPU box with the snapshot)."""
    out = os.path.join(ROOT, "tools", "_bin")
    os.makedirs(out, exist_ok=True)
    kernel = os.path.join(CSRC, "kernels", "gn_hessian_tiles.hip")
    for name, deps in (("gn_tiles_bench", [kernel]), ("valu_f64_peak", []), ("mfma_f64_peak", []), ("store_ceiling_tiles", []), ("store_issue_cost", []), ("resident_pingpong", [])):
        src = os.path.join(ROOT, "tools", f"{name}.hip")
        exe = os.path.join(out, name)
        if not _newer([exe], [src, *deps]):
            _run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-result", "-Wno-unused-value", "-o", exe, src])
    # the headline kernel's producer / consumer program at two wavefronts per SIMD next to the fused kernel: every entry of a launch
    # compared on the device (tests/test_quad_program.py), timings and counters (tools/gpu_split_occupancy.sh)
    split_src, split_exe = os.path.join(ROOT, "tools", "quad_split_bench.hip"), os.path.join(BUILD, "variants", "quad_split_bench")
    split_deps = [split_src, os.path.join(BUILD, "ungar_codegen"), os.path.join(ROOT, "tools", "make_split_bench.sh")] + _tree(os.path.join(CSRC, "kernels"), os.path.join(CSRC, "codegen"), os.path.join(CSRC, "tape"))
    if not _newer([split_exe], split_deps):
        _run(["bash", os.path.join(ROOT, "tools", "make_split_bench.sh")])


def build_all():
    generate(build_codegen())
    lib = build_library()
    build_cpp_tests()
    build_tools()
    return lib


if __name__ == "__main__":
    build_all()
    print(LIB)
    sys.exit(0)
