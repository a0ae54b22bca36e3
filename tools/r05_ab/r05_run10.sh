set -x
OUT=gpurun_out/r05k; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 20 --warmup 5 --cpu-steps 0 --tshirt 0"
L=$PWD/diffcloth_amd/lib
timeout 300 $B > $OUT/bench_main.log 2>&1
timeout 300 $B > $OUT/bench_main2.log 2>&1
timeout 300 $B --total-batch 32 > $OUT/bench_b32.log 2>&1
DC_LIB=$L/libdiffcloth_hip_ph.so timeout 300 python bench.py --steps 3 --warmup 5 --cpu-steps 0 --tshirt 0 > $OUT/bench_ph.log 2>&1; grep "phases pk" $OUT/bench_ph.log | tail -1
timeout 900 python -m pytest tests/test_gpu_bench_parity.py tests/test_gpu_configs.py -q -x -k "bench or forced_deflation or dress_7742_forward" > $OUT/parity.log 2>&1; tail -3 $OUT/parity.log
python tools/bench_summary.py $OUT
