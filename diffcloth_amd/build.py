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

ENGINE_SOURCES = ["dc_forward.hip", "dc_forward_res.hip", "dc_forward_pk.hip", "dc_adjoint.hip", "dc_selfcontact.hip", "dc_convert.hip", "dc_engine.hip", "dc_system.cpp"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_engine(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in ENGINE_SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in ("dc_device.h", "dc_devlib.h", "dc_system.h")] + [os.path.join(ROOT, "include", "diffcloth_hip.h")]
    if not force and not _stale(LIB, deps):
        return LIB
    cmd = [HIPCC, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=fast",
           "-Wall", "-Wno-unused-function", "-I", os.path.join(ROOT, "include"), "-o", LIB]
    cmd += os.environ.get("DC_CXXFLAGS", "").split()     # e.g. -DDC_PROFILE_PHASES for in-kernel phase timing
    for s in srcs:
        if s.endswith(".cpp"):
            cmd += ["-x", "hip", s]
        else:
            cmd += [s]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


def build_pymodule(force=False, verbose=False):
    """diffcloth_py: C++ host `Simulation` (csrc/host) + pybind11 surface of the reference, linked against the engine."""
    import sysconfig
    import pybind11
    host = os.path.join(CSRC, "host")
    srcs = [os.path.join(host, s) for s in ("simulation.cpp", "scene_tables.cpp", "pymodule.cpp")]
    out = os.path.join(LIBDIR, "diffcloth_py" + sysconfig.get_config_var("EXT_SUFFIX"))
    deps = srcs + [os.path.join(host, "simulation.h"), os.path.join(ROOT, "include", "diffcloth_hip.h"), LIB]
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
