OUT=gpurun_out/r03i; mkdir -p $OUT
L=$PWD/diffcloth_amd/lib
run() { name=$1; shift; ( timeout ${TMO:-400} "$@" > $OUT/$name.log 2>&1; echo "rc=$?" >> $OUT/$name.log ); echo "=== $name"; grep -h "phases\|^{\|passed\|failed\|rc=" $OUT/$name.log | cut -c1-260 | tail -${TAILN:-3}; }
TAILN=4 run ph env DC_LIB=$L/libdiffcloth_hip_ph.so python bench.py --steps 8 --warmup 5 --tshirt 0 --cpu-steps 0
run bench python bench.py --steps 20 --warmup 5 --tshirt 0 --cpu-steps 0
run bench32 python bench.py --steps 20 --warmup 5 --tshirt 0 --cpu-steps 0 --total-batch 32
TMO=900 run tests python -m pytest tests/test_gpu_parity.py tests/test_gpu_cluster.py tests/test_gpu_configs.py tests/test_gpu_bench_parity.py -q -x
