"""Developer script: HBM-side traffic of the step kernels from the rocprofv3 --pmc passes of tools/profile_round.sh,
corrected with the calibration passes (tools/pmc_calib.hip: known byte counts at 4 B and 16 B per lane), as
MI355X_MICROARCH.md (HBM section) prescribes. Writes <dir>/traffic.json: bytes per launch of the timed sweep."""
import csv, glob, json, os, sys, collections

def counters(folder, name):
    rows = collections.defaultdict(list)
    for f in glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                rows[r["Kernel_Name"].split("(")[0]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    return {k: [v for _, v in sorted(vs)] for k, vs in rows.items()}

def main(out):
    GiB = float(1 << 30)
    calib = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for k, v in counters(os.path.join(out, "calib_" + c), c).items():
            calib[(c, k)] = v[-1] * 1024.0 / GiB          # counter unit KiB; every calibration kernel moves 1 GiB
    print("calibration (reported / true bytes):", {f"{c}:{k}": round(v, 3) for (c, k), v in calib.items()})
    f4 = calib.get(("FETCH_SIZE", "k_read4"), 1.0); w4 = calib.get(("WRITE_SIZE", "k_write4"), 1.0)
    res = {"calibration": {f"{c}:{k}": v for (c, k), v in calib.items()}, "kernels": {}}
    fetch = counters(os.path.join(out, "pmc_FETCH_SIZE"), "FETCH_SIZE")
    write = counters(os.path.join(out, "pmc_WRITE_SIZE"), "WRITE_SIZE")
    for k in fetch:
        if "pd_step" not in k and "adjoint" not in k: continue
        fr = fetch[k][-1] * 1024.0; wr = write.get(k, [0.0])[-1] * 1024.0      # the last launch is the timed sweep
        res["kernels"][k.split("<")[0].replace("void ", "").replace("dc::", "")] = {"kernel": k, "fetch_raw_bytes": fr, "write_raw_bytes": wr,
                                                              "hbm_bytes_per_launch": fr / f4 + wr / w4}
        print(f"{k}: FETCH_SIZE {fr / 1e9:.2f} GB raw -> {fr / f4 / 1e9:.2f} GB, WRITE_SIZE {wr / 1e9:.2f} GB raw -> {wr / w4 / 1e9:.2f} GB, "
              f"traffic {(fr / f4 + wr / w4) / 1e9:.2f} GB per launch (timed sweep)")
    json.dump(res, open(os.path.join(out, "traffic.json"), "w"), indent=1)

if __name__ == "__main__":
    main(sys.argv[1])
