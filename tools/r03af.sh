OUT=gpurun_out/r03af; mkdir -p $OUT
( BENCH_PARITY_N=32 timeout 1500 python -m pytest tests/test_gpu_bench_parity.py -q -s -k "32-rollouts" > $OUT/parity32.log 2>&1 ); grep -h "worst over\|passed\|failed\|assert" $OUT/parity32.log | cut -c1-250
