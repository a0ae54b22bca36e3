OUT=gpurun_out/r03ab; mkdir -p $OUT
( BENCH_PARITY_N=40 timeout 1500 python -m pytest tests/test_gpu_bench_parity.py -q -s -k "256-rollouts" > $OUT/parity40.log 2>&1 ); grep -h "worst over\|passed\|failed\|assert" $OUT/parity40.log | cut -c1-250
