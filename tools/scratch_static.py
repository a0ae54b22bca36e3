"""Developer tool: where the scratch (register-spill) instructions of the hot kernel instances sit — inside or outside loops.

Reads the gfx950 code objects of diffcloth_amd/lib/obj (like tools/kernel_resources.py), disassembles each hot instance with llvm-objdump and
classifies every scratch_store / scratch_load — and every v_writelane / v_readlane, the instructions SGPR spills compile to — by its loop nesting depth (a loop = a backward branch and its target). A spill STORE inside a loop
is executed every iteration and shows up as HBM write traffic; one outside all loops is executed once per launch.
Usage: python tools/scratch_static.py [--out file]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_resources as kr

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def analyse(elf, mangled_substr):
    out = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", elf], capture_output=True, text=True).stdout.split("\n")
    res = []
    starts = [i for i, l in enumerate(out) if re.match(r"^[0-9a-f]+ <", l)]
    for si, s in enumerate(starts):
        name = out[s]
        if mangled_substr not in name:
            continue
        e = starts[si + 1] if si + 1 < len(starts) else len(out)
        base = int(name.split()[0], 16)
        loops, st, ld, rl, wl, n_ins = [], [], [], [], [], []
        for l in out[s + 1:e]:
            m = re.search(r"//\s*([0-9A-Fa-f]+):", l)
            if not m:
                continue
            a = int(m.group(1), 16)
            if "s_cbranch" in l or "s_branch" in l:
                t = re.search(r"\+0x([0-9a-fA-F]+)>", l)
                if t and base + int(t.group(1), 16) <= a:
                    loops.append((base + int(t.group(1), 16), a))
            n_ins.append(a)
            if "scratch_store" in l:
                st.append(a)
            if "scratch_load" in l:
                ld.append(a)
            if "v_readlane_b32" in l:
                rl.append(a)
            if "v_writelane_b32" in l:
                wl.append(a)
        depth = lambda a: sum(1 for t, b in loops if t <= a <= b)      # noqa: E731
        res.append((name.split("<", 1)[1].rstrip(">:"), e - s, len(loops), collections.Counter(min(depth(a), 3) for a in st),
                    collections.Counter(min(depth(a), 3) for a in ld), collections.Counter(min(depth(a), 3) for a in rl),
                    collections.Counter(min(depth(a), 3) for a in wl), collections.Counter(min(depth(a), 3) for a in n_ins)))
    return res


def main():
    lines = ["# scratch instructions of the hot gfx950 kernel instances by loop nesting depth (tools/scratch_static.py; depth 3 = 3 or deeper)",
             "# instance | instructions | loops | scratch_store at depth 0 / 1 / 2 / 3+ | scratch_load at depth 0 / 1 / 2 / 3+ | v_writelane (SGPR spill stores) 0 / 1 / 2 / 3+ | "
             "v_readlane (SGPR spill reloads + the product's deliberate row-table reads) 0 / 1 / 2 / 3+ | all instructions 0 / 1 / 2 / 3+"]
    with tempfile.TemporaryDirectory() as wd:
        res = kr.collect()
        for tag, prefix in kr.HOT.items():
            f, k = kr.find(res, prefix)
            if k is None:
                lines.append(f"{tag}: {prefix} NOT FOUND")
                continue
            elf = kr.device_elf(os.path.join(kr.OBJDIR, f), wd)
            for name, n, nl, st, ld, rl, wl, al in analyse(elf, k["name"].replace(".kd", "")):
                lines.append(f"{tag}: {prefix} | {n} | {nl} | {st[0]} / {st[1]} / {st[2]} / {st[3]} | {ld[0]} / {ld[1]} / {ld[2]} / {ld[3]} | "
                             f"{wl[0]} / {wl[1]} / {wl[2]} / {wl[3]} | {rl[0]} / {rl[1]} / {rl[2]} / {rl[3]} | {al[0]} / {al[1]} / {al[2]} / {al[3]}")
    text = "\n".join(lines)
    if "--out" in sys.argv:
        open(sys.argv[sys.argv.index("--out") + 1], "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
