set -x
OUT=gpurun_out/r05f; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 20 --warmup 5 --cpu-steps 0 --tshirt 0"
L=$PWD/diffcloth_amd/lib
timeout 300 $B > $OUT/bench_main.log 2>&1
timeout 300 $B --total-batch 32 > $OUT/bench_b32.log 2>&1
DC_LIB=$L/libdiffcloth_hip_r04.so timeout 300 $B --total-batch 32 > $OUT/bench_b32_r04.log 2>&1
python tools/bench_summary.py $OUT
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/gpu_tests.log 2>&1; tail -5 $OUT/gpu_tests.log
