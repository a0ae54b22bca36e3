OUT=gpurun_out/r03aa; mkdir -p $OUT
L=$PWD/diffcloth_amd/lib
( DC_LIB=$L/libdiffcloth_hip_v11.so timeout 300 python bench.py --steps 4 --warmup 2 --tshirt 0 --cpu-steps 0 > $OUT/ph.log 2>&1 )
grep -h "phases pk" $OUT/ph.log | tail -1 | cut -c1-600
( timeout 300 python bench.py --steps 20 --warmup 5 --tshirt 0 --cpu-steps 0 > $OUT/bench.log 2>&1 )
grep '"metric"' $OUT/bench.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']; print('value',round(d['value'],1),'ms',round(d['ms_per_step'],2),'pd',round(c['mean_pd_iters_per_step'],2),'cg',round(c['mean_cg_iters_per_pd_iter'],2),[ (k['kernel'],round(k['ms_per_step'],2)) for k in d['roofline']['kernels']])"
( timeout 900 python -m pytest tests/test_gpu_bench_parity.py -q -s -k "256-rollouts" > $OUT/parity.log 2>&1 ); grep -h "worst over\|passed\|failed" $OUT/parity.log | cut -c1-250
