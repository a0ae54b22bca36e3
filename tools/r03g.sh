OUT=gpurun_out/r03g; mkdir -p $OUT
run() { name=$1; shift; ( timeout ${TMO:-400} "$@" > $OUT/$name.log 2>&1; echo "rc=$?" >> $OUT/$name.log ); echo "=== $name"; tail -${TAILN:-8} $OUT/$name.log | cut -c1-400; }
TMO=900 TAILN=25 run new python -m pytest tests/test_gpu_reference_callers.py tests/test_gpu_functional.py tests/test_gpu_cluster.py tests/test_gpu_fullsize.py tests/test_gpu_comm.py -q -s -x
TMO=1800 TAILN=12 run all python -m pytest tests -m gpu -q
