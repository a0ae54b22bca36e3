#!/usr/bin/env python3
"""Register / scratch budget of every gfx950 kernel instance the engine ships, read from the CODE OBJECTS (not from belief).

hipcc -c leaves the device code of a translation unit as a clang offload bundle in the `.hip_fatbin` section of the host object; this
unpacks the gfx950 ELF of every object under diffcloth_amd/lib/obj/ and reads the AMDGPU metadata note (`llvm-readelf --notes`):
vgpr_count, agpr_count, vgpr_spill_count, sgpr_count, sgpr_spill_count, private_segment_fixed_size (scratch bytes per lane),
group_segment_fixed_size (static LDS), max_flat_workgroup_size.

  python tools/kernel_resources.py                    table of all kernels  -> stdout
  python tools/kernel_resources.py --out profiles/r05_kernel_resources.txt
  python tools/kernel_resources.py --hot              only the instances bench.py / the BASELINE configurations launch

tests/test_kernel_resources.py asserts the budget of the launched instances."""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDIR = os.path.join(ROOT, "diffcloth_amd", "lib", "obj")
LLVM = os.environ.get("DC_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"

FIELDS = ("vgpr_count", "agpr_count", "vgpr_spill_count", "sgpr_count", "sgpr_spill_count", "private_segment_fixed_size",
          "group_segment_fixed_size", "max_flat_workgroup_size")

# The instances the headline (bench.py: 256 x 1 and 32 x 8) and the BASELINE configurations launch, by demangled-name prefix.
HOT = {
    "bench forward 256x1": "dc::k_pd_step_pk<512, 20, 12, true, false, true, false>",
    "bench adjoint 256x1": "dc::k_adjoint_step<1024, true, false, false, false>",
    "bench forward 32x8": "dc::k_pd_step_cl<512, 3, true, true, false>",      # (PIPE = true: the single-exchange CG, round 6)
    "bench adjoint 32x8": "dc::k_adjoint_step_cl<512, false, false>",
    "garments adjoint (block preconditioner)": "dc::k_adjoint_step<1024, true, false, true, false>",
    "garments adjoint 32x8 (block preconditioner)": "dc::k_adjoint_step_cl<512, true, false>",
}


def device_elf(obj, workdir):
    """gfx950 code object of a host object (None when the object has no device code)."""
    fat = os.path.join(workdir, os.path.basename(obj) + ".fatbin")
    r = subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj, fat],
                       capture_output=True, text=True)
    if r.returncode or not os.path.exists(fat) or os.path.getsize(fat) == 0:
        return None
    elf = fat + ".elf"
    r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}", f"--targets={TARGET}",
                        f"--output={elf}"], capture_output=True, text=True)
    if r.returncode or not os.path.exists(elf) or os.path.getsize(elf) == 0:
        return None
    return elf


def demangle(names):
    import shutil
    tool = shutil.which("c++filt") or shutil.which("llvm-cxxfilt") or os.path.join(LLVM, "llvm-cxxfilt")
    try:
        r = subprocess.run([tool] + names, capture_output=True, text=True)
    except OSError:
        return names
    if r.returncode:
        return names
    out = r.stdout.strip().split("\n")
    return [re.sub(r"^void ", "", re.sub(r"\(.*$", "", o)) for o in out]      # template instance without return type / argument list


def kernels_of(elf):
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", elf], capture_output=True, text=True, check=True).stdout
    out, cur = [], None
    for line in notes.split("\n"):
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "agpr_count":                # first key of a kernel entry (keys are sorted)
            cur = {"agpr_count": int(v)}
            out.append(cur)
        elif cur is not None and k in FIELDS:
            cur[k] = int(v)
        elif cur is not None and k == "symbol" and v.endswith(".kd"):      # (argument entries have .name keys of their own)
            cur["name"] = v[:-3]
    out = [k for k in out if "name" in k]
    for k, d in zip(out, demangle([k["name"] for k in out])):
        k["demangled"] = d
    return out


def collect(objdir=OBJDIR):
    """{object file: [kernel dict]} for every object with device code."""
    res = {}
    with tempfile.TemporaryDirectory() as wd:
        for f in sorted(os.listdir(objdir)):
            if not f.endswith(".o"):
                continue
            elf = device_elf(os.path.join(objdir, f), wd)
            if elf:
                ks = kernels_of(elf)
                if ks:
                    res[f] = ks
    return res


def find(res, prefix):
    for f, ks in res.items():
        for k in ks:
            if k["demangled"].startswith(prefix):
                return f, k
    return None, None


def table(res, hot_only=False):
    lines = ["# gfx950 kernel resources from the code objects in diffcloth_amd/lib/obj (tools/kernel_resources.py)",
             "# vgpr agpr vgpr_spill sgpr sgpr_spill scratch_B/lane static_LDS_B max_threads  kernel"]
    if hot_only:
        for tag, prefix in HOT.items():
            f, k = find(res, prefix)
            if k is None:
                lines.append(f"# {tag}: {prefix} NOT FOUND")
                continue
            lines.append(f"{k['vgpr_count']:4d} {k['agpr_count']:4d} {k['vgpr_spill_count']:5d} {k['sgpr_count']:4d} {k['sgpr_spill_count']:5d} "
                         f"{k['private_segment_fixed_size']:6d} {k['group_segment_fixed_size']:7d} {k['max_flat_workgroup_size']:5d}  {k['demangled']}   [{tag}; {f}]")
        return "\n".join(lines)
    for f, ks in res.items():
        lines.append(f"## {f}")
        for k in sorted(ks, key=lambda k: k["demangled"]):
            lines.append(f"{k['vgpr_count']:4d} {k['agpr_count']:4d} {k['vgpr_spill_count']:5d} {k['sgpr_count']:4d} {k['sgpr_spill_count']:5d} "
                         f"{k['private_segment_fixed_size']:6d} {k['group_segment_fixed_size']:7d} {k['max_flat_workgroup_size']:5d}  {k['demangled']}")
    return "\n".join(lines)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out")
    ap.add_argument("--hot", action="store_true")
    ap.add_argument("--objdir", default=OBJDIR)
    a = ap.parse_args()
    res = collect(a.objdir)
    txt = table(res, hot_only=True) + "\n\n" + ("" if a.hot else table(res) + "\n")
    if a.out:
        with open(a.out, "w") as f:
            f.write(txt)
    sys.stdout.write(txt)


if __name__ == "__main__":
    main()
