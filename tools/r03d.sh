OUT=gpurun_out/r03d; mkdir -p $OUT
run() { name=$1; shift; ( timeout ${TMO:-400} "$@" > $OUT/$name.log 2>&1; echo "rc=$?" >> $OUT/$name.log ); echo "=== $name"; tail -${TAILN:-8} $OUT/$name.log | cut -c1-330; }
TAILN=9 run ab_base python tests/ab_adjoint.py c4
TAILN=9 run ab_hybrid env DC_PRECISE_ALL=1 python tests/ab_adjoint.py c4,hat --dump $OUT
TAILN=9 run ab_full64 env DC_PRECISE_ALL=2 python tests/ab_adjoint.py c4
TAILN=2 run bench python bench.py --steps 20 --warmup 5 --tshirt 0 --cpu-steps 0
TAILN=2 run bench_hybrid env DC_PRECISE_ALL=1 python bench.py --steps 20 --warmup 5 --tshirt 0 --cpu-steps 0
TAILN=2 run bench_full64 env DC_PRECISE_ALL=2 python bench.py --steps 20 --warmup 5 --tshirt 0 --cpu-steps 0
TAILN=3 run ab_dress_cl python tests/ab_adjoint.py dress7k
TAILN=3 run ab_dress_k1 env DC_CLUSTER=1 python tests/ab_adjoint.py dress7k
