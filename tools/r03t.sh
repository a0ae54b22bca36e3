OUT=gpurun_out/r03t; mkdir -p $OUT
VARS="v8" bash tools/r03s.sh
( timeout 300 python bench.py --steps 20 --warmup 5 --tshirt 0 --cpu-steps 0 > $OUT/bench.log 2>&1 )
grep '"metric"' $OUT/bench.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']; print('value',round(d['value'],1),'ms',round(d['ms_per_step'],2),'pd',round(c['mean_pd_iters_per_step'],2),'dmu',c['dL_dmu_sum_over_job'],[ (k['kernel'],round(k['ms_per_step'],2)) for k in d['roofline']['kernels']])"
( timeout 300 python bench.py --steps 20 --warmup 5 --tshirt 0 --cpu-steps 0 --total-batch 32 > $OUT/bench32.log 2>&1 )
grep '"metric"' $OUT/bench32.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']; print('value',round(d['value'],1),'ms',round(d['ms_per_step'],2),[ (k['kernel'],round(k['ms_per_step'],2)) for k in d['roofline']['kernels']])"
( timeout 900 python -m pytest tests/test_gpu_bench_parity.py tests/test_gpu_parity.py -q -x > $OUT/parity.log 2>&1 ); tail -3 $OUT/parity.log
