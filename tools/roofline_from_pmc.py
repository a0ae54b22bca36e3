"""Developer script (runs on the GPU box at the end of tools/profile_round.sh): turns the rocprofv3 passes of one bench.py
configuration into profiles/<tag>_roofline.json, the file bench.py's `roofline` block quotes.

Per step kernel (the launches of the timed sweep = the last `launches` dispatches of that kernel in every pass):
  hbm_bytes   calibrated FETCH_SIZE + WRITE_SIZE (separate --pmc passes; the calibration kernels of tools/pmc_calib.hip give
              reported / true bytes for 4-byte-per-lane reads and writes: FETCH_SIZE reads 0.5 on gfx950, WRITE_SIZE 1.0)
  lds_frac    SQ_LDS_IDX_ACTIVE / (CUs_of_one_XCD * GRBM_GUI_ACTIVE): LDS-array cycles per CU-cycle (SQ counters are those of one
              XCD = 32 CUs; bank-conflict cycles included, `lds_conflict_share` says how many of them)
  valu_frac   4 * SQ_ACTIVE_INST_VALU / (128 SIMDs * GRBM_GUI_ACTIVE)   (SQ_ACTIVE_INST_* count quad-cycles summed over waves)
  wait_frac   SQ_WAIT_ANY / SQ_WAVE_CYCLES: share of its life a wave is parked at s_waitcnt / a barrier
and, from the bench line of the un-profiled run with the same arguments: algorithmic bytes, launch time, config_key."""
import collections
import csv
import glob
import json
import os
import sys


def counters(folder):
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dc::", "")
            rows[k][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    return rows


def last(rows, kernel_prefix, counter, n):
    tot, name = 0.0, None
    for k, d in rows.items():
        if k.split("<")[0] == kernel_prefix and counter in d:
            vals = [v for _, v in sorted(d[counter])]
            tot += sum(vals[-n:]); name = k
    return tot, name


def main(out, tag):
    line = json.load(open(os.path.join(out, "bench_line.json")))
    GiB = float(1 << 30)
    calib = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        rows = counters(os.path.join(out, "calib_" + c))
        for k, d in rows.items():
            if c in d:
                calib[(c, k)] = sorted(d[c])[-1][1] * 1024.0 / GiB     # counter unit KiB; every calibration kernel moves 1 GiB
    f4 = calib.get(("FETCH_SIZE", "k_read4"), 0.5) or 0.5
    w4 = calib.get(("WRITE_SIZE", "k_write4"), 1.0) or 1.0
    fetch = counters(os.path.join(out, "pmc_FETCH_SIZE")); write = counters(os.path.join(out, "pmc_WRITE_SIZE"))
    sqb = counters(os.path.join(out, "pmc_sq_b")); sqc = counters(os.path.join(out, "pmc_sq_c"))
    res = {"tag": tag, "config_key": line["config"]["config_key"], "command": line.get("_command", ""),
           "calibration": {f"{c}:{k}": v for (c, k), v in calib.items()}, "bench_value": line["value"], "kernels": {}}
    for ent in line["roofline"]["kernels"]:
        name, n = ent["kernel"], max(int(ent.get("launches", 1)), 1)
        fr, full = last(fetch, name, "FETCH_SIZE", n)
        wr, _ = last(write, name, "WRITE_SIZE", n)
        hbm = fr * 1024.0 / f4 + wr * 1024.0 / w4
        g1, _ = last(sqc, name, "GRBM_GUI_ACTIVE", n)
        lds, _ = last(sqc, name, "SQ_LDS_IDX_ACTIVE", n)
        conf, _ = last(sqc, name, "SQ_LDS_BANK_CONFLICT", n)
        wany, _ = last(sqc, name, "SQ_WAIT_ANY", n)
        wcyc_c, _ = last(sqc, name, "SQ_WAVE_CYCLES", n)
        g2, _ = last(sqb, name, "GRBM_GUI_ACTIVE", n)
        valu, _ = last(sqb, name, "SQ_ACTIVE_INST_VALU", n)
        k = {"kernel": full or name, "launches": n, "hbm_bytes": hbm, "fetch_raw_bytes": fr * 1024.0, "write_raw_bytes": wr * 1024.0,
             "algorithmic_bytes": ent["algorithmic_bytes"], "launch_ms_total": ent["avg_launch_ms"] * n,
             "hbm_frac": hbm / (ent["avg_launch_ms"] * n * 1e-3) / 8e12 if hbm else None,
             "lds_frac": lds / (32.0 * g1) if g1 else None, "lds_conflict_share": conf / lds if lds else None,
             "valu_frac": 4.0 * valu / (128.0 * g2) if g2 else None, "wait_frac": wany / wcyc_c if wcyc_c else None,
             # what the kernel writes beyond the stores its algorithm needs (bench.py: compulsory_write_bytes): register spills to scratch
             "scratch_write_bytes": max(wr * 1024.0 / w4 - ent["compulsory_write_bytes"], 0.0) if "compulsory_write_bytes" in ent else None}
        res["kernels"][name] = k
        print(name, json.dumps(k))
    path = os.path.join(out, f"{tag}_roofline.json")
    json.dump(res, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
