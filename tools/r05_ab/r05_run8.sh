set -x
OUT=gpurun_out/r05i; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 20 --warmup 5 --cpu-steps 0 --tshirt 0"
L=$PWD/diffcloth_amd/lib
DC_LIB=$L/libdiffcloth_hip_r04.so timeout 300 $B > $OUT/bench_r04.log 2>&1
timeout 300 $B > $OUT/bench_main.log 2>&1
DC_LIB=$L/libdiffcloth_hip_r04.so timeout 300 $B > $OUT/bench_r04_2.log 2>&1
timeout 300 $B > $OUT/bench_main2.log 2>&1
python tools/bench_summary.py $OUT
