OUT=gpurun_out/r03q; mkdir -p $OUT
for tol in 1e-3 3e-4; do
  ( timeout 300 python bench.py --steps 20 --warmup 5 --tshirt 0 --cpu-steps 0 --cg-tol $tol > $OUT/bench_$tol.log 2>&1 )
  grep '"metric"' $OUT/bench_$tol.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']
print('cg_tol',c['cg_rel_tol'],'value',round(d['value'],1),'ms',round(d['ms_per_step'],2),'pd',round(c['mean_pd_iters_per_step'],2),'cg',round(c['mean_cg_iters_per_pd_iter'],2),'adj',round(c['mean_adjoint_iters_per_step'],2),[ (k['kernel'],round(k['ms_per_step'],2)) for k in d['roofline']['kernels']])"
  ( BENCH_CG_TOL=$tol timeout 900 python -m pytest tests/test_gpu_bench_parity.py -q -s -k "256-rollouts" > $OUT/parity_$tol.log 2>&1 )
  grep -h "worst over\|passed\|failed\|assert" $OUT/parity_$tol.log | cut -c1-250 | tail -5
done
