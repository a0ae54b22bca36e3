# developer script: SQ counter passes (instruction mix, wait / busy cycles, LDS bank conflicts) over a short bench run;
# prints one table (means over the launches of each step kernel, whole GPU)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { rocprofv3 --kernel-trace --pmc $2 -d $R/gpurun_out/pmc_$1 -o x --output-format csv -- python $R/bench.py --steps 4 --warmup 3 --cpu-steps 0 > $R/gpurun_out/pmc_$1.log 2>&1; }
run a "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
run b "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
run c "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_LDS"
run d "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_INSTS_FLAT GRBM_GUI_ACTIVE"
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/pmc_[abcd]/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void dc::", "")
        if "pd_step" in k or "adjoint" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("| kernel | counter | value of the timed sweep (4 steps x 256 rollouts) |\n|---|---|---|")
for k, d in sorted(agg.items()):
    for c, v in sorted(d.items()):
        print(f"| {k} | {c} | {max(v):.4g} |")
PY
