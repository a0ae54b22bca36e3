OUT=gpurun_out/r03p; mkdir -p $OUT
run() { name=$1; shift; ( timeout ${TMO:-500} "$@" > $OUT/$name.log 2>&1; echo "rc=$?" >> $OUT/$name.log ); echo "=== $name"; grep -h "^\[config\] B=\|^\[hat\]\|oracle's own\|passed\|failed\|rc=\|Error\|assert" $OUT/$name.log | cut -c1-330 | tail -${TAILN:-12}; }
TMO=1800 TAILN=24 run rest python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_cluster.py tests/test_gpu_functional.py tests/test_gpu_reference_callers.py -q -s
