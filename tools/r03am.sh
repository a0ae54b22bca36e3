OUT=gpurun_out/r03am; mkdir -p $OUT
( timeout 1500 python -m pytest tests/test_gpu_configs.py -q -s > $OUT/t.log 2>&1 ); grep -h "^\[config\] B=\|PD iterations gpu .* against\|passed\|failed\|FAILED" $OUT/t.log | cut -c1-330
