OUT=gpurun_out/r03w; mkdir -p $OUT
( timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_cluster.py -q -s > $OUT/t.log 2>&1 ); grep -h "^\[config\] B=\|garment\|passed\|failed\|FAILED" $OUT/t.log | cut -c1-400
