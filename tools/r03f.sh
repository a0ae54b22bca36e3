OUT=gpurun_out/r03f; mkdir -p $OUT
run() { name=$1; shift; ( timeout ${TMO:-400} "$@" > $OUT/$name.log 2>&1; echo "rc=$?" >> $OUT/$name.log ); echo "=== $name"; tail -${TAILN:-8} $OUT/$name.log | cut -c1-300; }
TAILN=2 run bench python bench.py --steps 20 --warmup 5 --tshirt 0 --cpu-steps 0
TAILN=2 run bench32 python bench.py --steps 20 --warmup 5 --tshirt 0 --cpu-steps 0 --total-batch 32
TAILN=4 run ab python tests/ab_adjoint.py c4,hat
TAILN=6 run par python -m pytest tests/test_gpu_parity.py tests/test_gpu_cluster.py -q -x
