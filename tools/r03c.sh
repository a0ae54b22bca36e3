OUT=gpurun_out/r03c; mkdir -p $OUT
run() { name=$1; shift; ( timeout ${TMO:-400} "$@" > $OUT/$name.log 2>&1; echo "rc=$?" >> $OUT/$name.log ); echo "=== $name"; tail -${TAILN:-8} $OUT/$name.log | cut -c1-330; }
TAILN=14 run ab_k1 python tests/ab_adjoint.py c4,hat --dump $OUT
TAILN=4 run ab_dress_k1 env DC_CLUSTER=1 python tests/ab_adjoint.py dress7k,dress
TAILN=4 run ab_dress_cl python tests/ab_adjoint.py dress7k,dress
TAILN=2 run bench python bench.py --steps 20 --warmup 5 --tshirt 0 --cpu-steps 0
TAILN=2 run bench_norec env DC_PRECISE_RECORD=0 python bench.py --steps 20 --warmup 5 --tshirt 0 --cpu-steps 0
TAILN=2 run bench32 python bench.py --steps 20 --warmup 5 --tshirt 0 --cpu-steps 0 --total-batch 32
TMO=1500 TAILN=40 run all python -m pytest tests -m gpu -q
