#!/bin/bash
# round 6, call 16: straight-line boundary publishing in the split kernels (range-checked drop instead of exec-masked blocks), same-box A/B
OUT=gpurun_out/r06_16; mkdir -p $OUT
PREV=$PWD/diffcloth_amd/lib/libdiffcloth_hip_prev.so
bb() { tag=$1; tb=$2; shift; shift; ( env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --total-batch $tb --cpu-steps 0 --tshirt 0 --secondary none > $OUT/b_$tag.log 2>&1 ); python - "$OUT/b_$tag.log" "$tag" <<'P'
import sys,json
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); c=d['config']
    print(sys.argv[2],'value',round(d['value'],1),'ms',round(d['ms_per_step'],3),'pd',round(c['mean_pd_iters_per_step'],2),'cg/pd',round(c['mean_cg_iters_per_pd_iter'],3),'apps',round(c.get('mean_adjoint_operator_applications_per_step',0),2),[(k['kernel'],round(k['ms_per_step'],3)) for k in d['roofline']['kernels']], c['gradients_finite'])
except Exception as ex: print(sys.argv[2],'FAILED',ex); print(open(sys.argv[1]).read()[-1200:])
P
}
for i in 1 2 3; do
bb new_$i 32 X=1
bb prev_$i 32 DC_LIB=$PREV
done
( timeout 900 python -m pytest tests/test_gpu_cluster.py "tests/test_gpu_bench_parity.py::test_bench_configuration_matches_oracle[32-rollouts-split-over-8-workgroups]" "tests/test_gpu_fullsize.py" -q -x > $OUT/tests.log 2>&1 ); echo "tests rc=$?"; grep -E "passed|failed" $OUT/tests.log | tail -2
