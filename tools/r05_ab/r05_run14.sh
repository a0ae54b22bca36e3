set -x
OUT=gpurun_out/r05o; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 420 python -m pytest tests/test_gpu_garments10k.py -q -s -k dress > $OUT/garments.log 2>&1; grep -E "^\[|passed|failed|Error|assert" $OUT/garments.log | cut -c1-420 | tail -8
timeout 600 python -m pytest tests/test_gpu_bench_parity.py -q -x -k bench_configuration -s > $OUT/parity.log 2>&1; grep -E "near-cancelling|worst over|passed|failed" $OUT/parity.log | cut -c1-400
timeout 420 python -u tools/bench_configs.py > $OUT/other_configs.txt 2>&1; cat $OUT/other_configs.txt | cut -c1-330
