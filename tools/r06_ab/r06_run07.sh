#!/bin/bash
# round 6, call 7: split kernels with the half-precision direction (forward) and CG-first (adjoint): A/B at 32 rollouts x 8, then the split tests
OUT=gpurun_out/r06_07; mkdir -p $OUT
bb() { tag=$1; tb=$2; shift; shift; ( env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --total-batch $tb --cpu-steps 0 --tshirt 0 --secondary "" > $OUT/b_$tag.log 2>&1 ); python - "$OUT/b_$tag.log" "$tag" <<'P'
import sys,json
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); c=d['config']
    print(sys.argv[2],'value',round(d['value'],1),'ms',round(d['ms_per_step'],3),'pd',round(c['mean_pd_iters_per_step'],2),'cg/pd',round(c['mean_cg_iters_per_pd_iter'],3),'adj',round(c['mean_adjoint_iters_per_step'],2),'adjcg',round(c.get('mean_adjoint_cg_iters_per_step',0),2),'apps',round(c.get('mean_adjoint_operator_applications_per_step',0),2),'cyc',round(c['mean_fp32_solves_per_adjoint'],2),[(k['kernel'],round(k['ms_per_step'],3)) for k in d['roofline']['kernels']])
except Exception as ex: print(sys.argv[2],'FAILED',ex); print(open(sys.argv[1]).read()[-1500:])
P
}
for i in 1 2; do
bb new_$i 32 DC_ADJ_CG=1
bb sx0_$i 32 DC_ADJ_CG=1 DC_SXCG=0
bb cg0_$i 32 DC_ADJ_CG=0
done
bb b64 64 DC_ADJ_CG=1
bb b128 128 DC_ADJ_CG=1
( timeout 900 python -m pytest tests/test_gpu_cluster.py "tests/test_gpu_bench_parity.py::test_bench_configuration_matches_oracle[32-rollouts-split-over-8-workgroups]" tests/test_gpu_fullsize.py -q -x -s > $OUT/tests.log 2>&1 ); echo "tests rc=$?"; grep -E "passed|failed" $OUT/tests.log | tail -2
grep -E "^\[bench parity\] worst|garment, seed" $OUT/tests.log | cut -c1-330
