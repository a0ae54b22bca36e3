set -x
OUT=gpurun_out/r05m; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 20 --warmup 5 --cpu-steps 0 --tshirt 0 --total-batch 32"
L=$PWD/diffcloth_amd/lib
DC_LIB=$L/libdiffcloth_hip_r04.so timeout 300 $B > $OUT/bench_b32_r04.log 2>&1
timeout 300 $B > $OUT/bench_b32.log 2>&1
for v in aclA aclB aclC; do DC_LIB=$L/libdiffcloth_hip_$v.so timeout 300 $B > $OUT/bench_b32_$v.log 2>&1; done
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-steps 0 --tshirt 0 > $OUT/bench_main.log 2>&1
python tools/bench_summary.py $OUT
python - <<'PY'
import json
for l in open('gpurun_out/r05m/bench_main.log'):
    if l.startswith('{"metric"'):
        print(json.loads(l)["config"]["slowest_rollout_over_mean"])
PY
