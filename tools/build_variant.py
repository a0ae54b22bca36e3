"""Developer tool: an A/B build of the engine.  python tools/build_variant.py NAME "-DFLAG ..." file.hip [file.hip ...]
Compiles the listed sources with the extra flags into diffcloth_amd/lib/obj_NAME/ and links them with the standard objects of every
other source into diffcloth_amd/lib/libdiffcloth_hip_NAME.so (select it with DC_LIB=<path>, diffcloth_amd/capi.py)."""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffcloth_amd import build as B

name, extra, files = sys.argv[1], sys.argv[2].split(), sys.argv[3:]
if not os.environ.get('DC_SKIP_BASE'):      # DC_SKIP_BASE=1: link against the objects as they are (only the listed sources are recompiled)
    B.build_engine()
objdir = os.path.join(B.LIBDIR, "obj_" + name)
os.makedirs(objdir, exist_ok=True)
flags = [f"--offload-arch={B.ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wno-unused-function", "-I", os.path.join(ROOT, "include")] + extra
jobs, objs = [], []
for s in B.ENGINE_SOURCES:
    if s in files:
        o = os.path.join(objdir, s + ".o")
        jobs.append([B.HIPCC] + flags + ["-c", os.path.join(B.CSRC, s), "-o", o])
    else:
        o = os.path.join(B.LIBDIR, "obj", s + ".o")
    objs.append(o)
with ThreadPoolExecutor(max_workers=len(jobs)) as pool:
    list(pool.map(subprocess.check_call, jobs))
out = os.path.join(B.LIBDIR, f"libdiffcloth_hip_{name}.so")
subprocess.check_call([B.HIPCC, f"--offload-arch={B.ARCH}", "-shared", "-fPIC", "-o", out] + objs)
print(out)
