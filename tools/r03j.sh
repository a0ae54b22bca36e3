OUT=gpurun_out/r03j; mkdir -p $OUT
L=$PWD/diffcloth_amd/lib
run() { name=$1; shift; ( timeout ${TMO:-400} "$@" > $OUT/$name.log 2>&1; echo "rc=$?" >> $OUT/$name.log ); echo "=== $name"; grep -h "phases pk\|^{\|passed\|failed\|rc=\|worst" $OUT/$name.log | cut -c1-260 | tail -${TAILN:-3}; }
run base python bench.py --steps 20 --warmup 5 --tshirt 0 --cpu-steps 0
run tp env DC_LIB=$L/libdiffcloth_hip_tp.so python bench.py --steps 20 --warmup 5 --tshirt 0 --cpu-steps 0
TAILN=3 run tpph env DC_LIB=$L/libdiffcloth_hip_tpph.so python bench.py --steps 8 --warmup 5 --tshirt 0 --cpu-steps 0
TAILN=2 run tpab env DC_LIB=$L/libdiffcloth_hip_tp.so python tests/ab_adjoint.py c4
