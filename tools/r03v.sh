OUT=gpurun_out/r03v; mkdir -p $OUT
L=$PWD/diffcloth_amd/lib
( DC_LIB=$L/libdiffcloth_hip_v10.so timeout 300 python bench.py --steps 4 --warmup 2 --tshirt 0 --cpu-steps 0 > $OUT/ph.log 2>&1 )
grep -h "phases adj" $OUT/ph.log | tail -1 | cut -c1-600
grep -h "phases pk" $OUT/ph.log | tail -1 | cut -c1-600
( timeout 300 python bench.py --steps 20 --warmup 5 --tshirt 0 --cpu-steps 0 > $OUT/bench.log 2>&1 )
grep '"metric"' $OUT/bench.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']; print('value',round(d['value'],1),'ms',round(d['ms_per_step'],2),'pd',round(c['mean_pd_iters_per_step'],2),'adj',c['mean_adjoint_iters_per_step'],'dmu',c['dL_dmu_sum_over_job'],[ (k['kernel'],round(k['ms_per_step'],2)) for k in d['roofline']['kernels']])"
( timeout 1500 python -m pytest tests/test_gpu_bench_parity.py tests/test_gpu_parity.py -q -x > $OUT/parity.log 2>&1 ); tail -3 $OUT/parity.log | cut -c1-300
