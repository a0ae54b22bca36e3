# developer script: L2 hit / miss counters of the step kernels (one rocprofv3 --pmc pass)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $R/gpurun_out/pmc_l2 -o x --output-format csv -- python $R/bench.py --steps 3 --warmup 2 --cpu-steps 0 --tshirt 0 --secondary none > $R/gpurun_out/pmc_l2.log 2>&1
python - <<PY
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$R/gpurun_out/pmc_l2/x_counter_collection.csv")):
    k = r["Kernel_Name"].split("(")[0]
    if "pd_step" in k or "adjoint" in k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    h = sum(d["TCC_HIT_sum"]); m = sum(d["TCC_MISS_sum"])
    print(k, {c: f"{sum(v):.3g}" for c, v in d.items()}, "hit rate %.3f" % (h / max(h + m, 1)))
PY
