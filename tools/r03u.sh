OUT=gpurun_out/r03u; mkdir -p $OUT
L=$PWD/diffcloth_amd/lib
( DC_LIB=$L/libdiffcloth_hip_v9.so timeout 300 python bench.py --steps 4 --warmup 2 --tshirt 0 --cpu-steps 0 > $OUT/v9.log 2>&1 )
grep -h "phases adj" $OUT/v9.log | tail -3 | cut -c1-600
