OUT=gpurun_out/r03m; mkdir -p $OUT
run() { name=$1; shift; ( timeout ${TMO:-400} "$@" > $OUT/$name.log 2>&1; echo "rc=$?" >> $OUT/$name.log ); echo "=== $name"; grep -h "^\[config\] rollout 255\|passed\|failed\|nprim\|rc=" $OUT/$name.log | cut -c1-330 | tail -${TAILN:-3}; }
run dflt python -m pytest tests/test_gpu_configs.py -q -s -k "dress and 256"
run verify env DC_ADJ_VERIFY=1 python -m pytest tests/test_gpu_configs.py -q -s -k "dress and 256"
run tol8 env DRESS_ADJ_TOL=1e-8 python -m pytest tests/test_gpu_configs.py -q -s -k "dress and 256"
run tol8v env DRESS_ADJ_TOL=1e-8 DC_ADJ_VERIFY=1 python -m pytest tests/test_gpu_configs.py -q -s -k "dress and 256"
run hat python -m pytest tests/test_gpu_configs.py -q -s -k "hat"
