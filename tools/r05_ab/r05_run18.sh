set -x
OUT=gpurun_out/r05t; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 20 --warmup 5 --cpu-steps 0 --tshirt 0"
L=$PWD/diffcloth_amd/lib
timeout 300 $B > $OUT/bench_main.log 2>&1
DC_LIB=$L/libdiffcloth_hip_xl.so timeout 300 $B > $OUT/bench_xl.log 2>&1
timeout 300 $B > $OUT/bench_main2.log 2>&1
DC_LIB=$L/libdiffcloth_hip_xl.so timeout 300 $B > $OUT/bench_xl2.log 2>&1
python tools/bench_summary.py $OUT
DC_LIB=$L/libdiffcloth_hip_xl.so timeout 600 python -m pytest tests/test_gpu_bench_parity.py -q -x -k "256-rollouts" > $OUT/parity.log 2>&1; tail -2 $OUT/parity.log
