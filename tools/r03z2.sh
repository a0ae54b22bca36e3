mkdir -p gpurun_out/r03z
for i in 1 2; do
( timeout 300 python bench.py --steps 20 --warmup 5 --tshirt 0 --cpu-steps 0 > gpurun_out/r03z/bench6.log 2>&1 )
grep '"metric"' gpurun_out/r03z/bench6.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']; print('value',round(d['value'],1),'ms',round(d['ms_per_step'],2),[ (k['kernel'],round(k['ms_per_step'],2)) for k in d['roofline']['kernels']])"
done
