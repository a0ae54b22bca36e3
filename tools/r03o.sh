OUT=gpurun_out/r03o; mkdir -p $OUT
DC_DUMP_DIR=$OUT/dump timeout 300 python -m pytest tests/test_gpu_configs.py -q -s -k "hat and pressed" > $OUT/hat.log 2>&1
grep -h "^\[config\]\|^\[hat\]\|passed\|failed" $OUT/hat.log | cut -c1-300
