"""Builds the native libraries of diffcloth_amd in-tree with hipcc for gfx950 (cross-compiles without a GPU).

  libdiffcloth_hip.so   C-ABI engine (include/diffcloth_hip.h): host system builder + HIP kernels
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "diffcloth_amd", "csrc")
LIBDIR = os.path.join(ROOT, "diffcloth_amd", "lib")
LIB = os.path.join(LIBDIR, "libdiffcloth_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

ENGINE_SOURCES = ["dc_forward.hip", "dc_forward_res.hip", "dc_forward_pk.hip", "dc_forward_pk_defl.hip", "dc_forward_cl.hip", "dc_forward_cl_defl.hip", "dc_adjoint.hip", "dc_adjoint_cl.hip", "dc_selfcontact.hip", "dc_convert.hip", "dc_engine.hip", "dc_system.cpp", "dc_windows.cpp", "dc_packets.cpp", "dc_dense.cpp", "dc_deflate.cpp", "dc_spheremesh.cpp"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_engine(force=False, verbose=False):
    """Compiles every source to its own object (in parallel, only the stale ones) and links the shared library."""
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")] + [os.path.join(ROOT, "include", "diffcloth_hip.h")]
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wall", "-Wno-unused-function",
             "-I", os.path.join(ROOT, "include")] + os.environ.get("DC_CXXFLAGS", "").split()   # e.g. -DDC_PROFILE_PHASES
    stamp = os.path.join(objdir, "flags.txt")
    if not os.path.exists(stamp) or open(stamp).read() != " ".join(flags):
        force = True
    jobs, objs = [], []
    for name in ENGINE_SOURCES:
        src = os.path.join(CSRC, name)
        obj = os.path.join(objdir, name + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            jobs.append([HIPCC] + flags + ["-c"] + (["-x", "hip"] if name.endswith(".cpp") else []) + [src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as pool:
            list(pool.map(run, jobs))
    if jobs or not os.path.exists(LIB):
        run([HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs)
        with open(stamp, "w") as f:
            f.write(" ".join(flags))
    return LIB


def build_pymodule(force=False, verbose=False):
    """diffcloth_py: C++ host `Simulation` (csrc/host) + pybind11 surface of the reference, linked against the engine."""
    import sysconfig
    import pybind11
    host = os.path.join(CSRC, "host")
    srcs = [os.path.join(host, s) for s in ("simulation.cpp", "scene_tables.cpp", "optimize.cpp", "export.cpp", "pymodule.cpp")]
    out = os.path.join(LIBDIR, "diffcloth_py" + sysconfig.get_config_var("EXT_SUFFIX"))
    deps = srcs + [os.path.join(host, "simulation.h"), os.path.join(host, "optimize.h"), os.path.join(ROOT, "include", "diffcloth_hip.h"), LIB]
    if not force and not _stale(out, deps):
        return out
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall",
           "-I", pybind11.get_include(), "-I", sysconfig.get_paths()["include"], "-o", out] + srcs + \
          ["-L", LIBDIR, "-ldiffcloth_hip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build_engine(force="--force" in sys.argv, verbose=True))
    print(build_pymodule(force="--force" in sys.argv, verbose=True))
