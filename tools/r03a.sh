OUT=gpurun_out/r03a; mkdir -p $OUT
INL=$PWD/diffcloth_amd/lib/libdiffcloth_hip_inline.so
run() { name=$1; shift; ( timeout 400 "$@" > $OUT/$name.log 2>&1; echo "rc=$?" >> $OUT/$name.log ); echo "=== $name"; tail -${TAILN:-8} $OUT/$name.log | cut -c1-330; }
TAILN=14 run ab_fn python tests/ab_adjoint.py c4,hat
TAILN=8 run ab_inline env DC_LIB=$INL python tests/ab_adjoint.py c4
TAILN=14 run ab_fp32 python tests/ab_adjoint.py c4,hat --fp32-only
TAILN=6 run ab_dress env DC_CLUSTER=1 python tests/ab_adjoint.py dress7k,dress
TAILN=2 run bench_fn python bench.py --steps 20 --warmup 5 --tshirt 0 --cpu-steps 0
TAILN=2 run bench_inline env DC_LIB=$INL python bench.py --steps 20 --warmup 5 --tshirt 0 --cpu-steps 0
TAILN=2 run bench_fp32 env DC_ADJ_FP32=1 python bench.py --steps 20 --warmup 5 --tshirt 0 --cpu-steps 0
TAILN=12 run parity python -m pytest tests/test_gpu_parity.py -x -q
