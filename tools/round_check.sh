OUT=gpurun_out/round_check; mkdir -p $OUT
for a in "" "--total-batch 32"; do
( timeout 300 python bench.py --steps 20 --warmup 5 --tshirt 0 --cpu-steps 0 $a > $OUT/bench.log 2>&1 )
grep '"metric"' $OUT/bench.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']; print('value',round(d['value'],1),'ms',round(d['ms_per_step'],2),'pd',round(c['mean_pd_iters_per_step'],2),'adj',c['mean_adjoint_iters_per_step'],[ (k['kernel'],round(k['ms_per_step'],2)) for k in d['roofline']['kernels']])"
done
( timeout 2400 python -m pytest tests -m gpu -q > $OUT/suite.log 2>&1 ); tail -8 $OUT/suite.log | cut -c1-300
