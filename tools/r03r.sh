OUT=gpurun_out/r03r; mkdir -p $OUT
L=$PWD/diffcloth_amd/lib
( DC_LIB=$L/libdiffcloth_hip_ph.so timeout 300 python bench.py --steps 4 --warmup 2 --tshirt 0 --cpu-steps 0 > $OUT/ph.log 2>&1 )
grep -h "phases pk" $OUT/ph.log | tail -4 | cut -c1-400
