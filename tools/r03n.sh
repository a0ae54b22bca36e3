OUT=gpurun_out/r03n; mkdir -p $OUT
run() { name=$1; shift; ( timeout ${TMO:-500} "$@" > $OUT/$name.log 2>&1; echo "rc=$?" >> $OUT/$name.log ); echo "=== $name"; grep -h "^\[config\] rollout 255\|^\[config\] B=\|^\[hat\]\|passed\|failed\|rc=\|Error\|assert" $OUT/$name.log | cut -c1-330 | tail -${TAILN:-8}; }
run hat python -m pytest tests/test_gpu_configs.py -q -s -k "hat"
run dress python -m pytest tests/test_gpu_configs.py -q -s -k "dress and 256"
run novlo env DC_VLO=0 python -m pytest tests/test_gpu_configs.py -q -s -k "hat or (dress and 256)"
TMO=1500 TAILN=12 run rest python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_cluster.py tests/test_gpu_functional.py -q -x
run bench python bench.py --steps 20 --warmup 5 --tshirt 0 --cpu-steps 0
tail -1 $OUT/bench.log | cut -c1-600
