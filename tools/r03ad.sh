OUT=gpurun_out/r03ad; mkdir -p $OUT
for t in 512 1024; do
( DC_BWD_THREADS=$t timeout 300 python bench.py --steps 20 --warmup 5 --tshirt 0 --cpu-steps 0 > $OUT/bench_$t.log 2>&1 )
grep '"metric"' $OUT/bench_$t.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']; print('value',round(d['value'],1),'ms',round(d['ms_per_step'],2),[ (k['kernel'],round(k['ms_per_step'],2)) for k in d['roofline']['kernels']])"
done
