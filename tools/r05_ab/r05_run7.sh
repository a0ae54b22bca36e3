set -x
OUT=gpurun_out/r05h; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 20 --warmup 5 --cpu-steps 0 --tshirt 0"
L=$PWD/diffcloth_amd/lib
timeout 300 $B > $OUT/bench_main.log 2>&1
DC_BWD_THREADS=512 timeout 300 $B > $OUT/bench_bwd512.log 2>&1
DC_LIB=$L/libdiffcloth_hip_phadj.so timeout 300 python bench.py --steps 3 --warmup 5 --cpu-steps 0 --tshirt 0 > $OUT/bench_phadj.log 2>&1; grep "phases adj" $OUT/bench_phadj.log | tail -2
python tools/bench_summary.py $OUT
