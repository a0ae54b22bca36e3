set -x
OUT=gpurun_out/r05n; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 20 --warmup 5 --cpu-steps 0 --tshirt 0"
L=$PWD/diffcloth_amd/lib
timeout 300 $B > $OUT/bench_main.log 2>&1
timeout 300 $B --total-batch 32 > $OUT/bench_b32.log 2>&1
DC_LIB=$L/libdiffcloth_hip_r04.so timeout 300 $B --total-batch 32 > $OUT/bench_b32_r04.log 2>&1
python tools/bench_summary.py $OUT
timeout 1500 python -m pytest tests/test_gpu_garments10k.py -q -s > $OUT/garments.log 2>&1; grep -E "^\[|passed|failed|Error|assert" $OUT/garments.log | cut -c1-400 | tail -12
timeout 900 python tools/bench_configs.py > $OUT/other_configs.txt 2>&1; cat $OUT/other_configs.txt | cut -c1-330
timeout 900 python -m pytest tests/test_gpu_bench_parity.py -q -x -k bench_configuration > $OUT/parity.log 2>&1; tail -3 $OUT/parity.log
