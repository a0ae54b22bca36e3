OUT=gpurun_out/r03h; mkdir -p $OUT
L=$PWD/diffcloth_amd/lib
run() { name=$1; shift; ( timeout ${TMO:-400} "$@" > $OUT/$name.log 2>&1; echo "rc=$?" >> $OUT/$name.log ); echo "=== $name"; grep -h "phases\|^{" $OUT/$name.log | cut -c1-260 | tail -${TAILN:-3}; }
TAILN=6 run ph env DC_LIB=$L/libdiffcloth_hip_ph.so python bench.py --steps 8 --warmup 5 --tshirt 0 --cpu-steps 0
run eb4 python bench.py --steps 20 --warmup 5 --tshirt 0 --cpu-steps 0
run eb6 env DC_LIB=$L/libdiffcloth_hip_eb6.so python bench.py --steps 20 --warmup 5 --tshirt 0 --cpu-steps 0
run eb8 env DC_LIB=$L/libdiffcloth_hip_eb8.so python bench.py --steps 20 --warmup 5 --tshirt 0 --cpu-steps 0
