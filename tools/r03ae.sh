OUT=gpurun_out/r03ae; mkdir -p $OUT
L=$PWD/diffcloth_amd/lib
for v in nt base; do
if [ $v = base ]; then unset DC_LIB; else export DC_LIB=$L/libdiffcloth_hip_$v.so; fi
( timeout 300 python bench.py --steps 20 --warmup 5 --tshirt 0 --cpu-steps 0 > $OUT/bench_$v.log 2>&1 )
grep '"metric"' $OUT/bench_$v.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']; print('$v value',round(d['value'],1),'ms',round(d['ms_per_step'],2),[ (k['kernel'],round(k['ms_per_step'],2)) for k in d['roofline']['kernels']])"
done
