set -x
OUT=gpurun_out/r05q; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 20 --warmup 5 --cpu-steps 0 --tshirt 0"
L=$PWD/diffcloth_amd/lib
for v in noa prio eb3 eb5 noa prio; do DC_LIB=$L/libdiffcloth_hip_$v.so timeout 300 $B > $OUT/bench_${v}_$RANDOM.log 2>&1; done
python tools/bench_summary.py $OUT
