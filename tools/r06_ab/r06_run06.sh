#!/bin/bash
# round 6, call 6: adjoint CG-first on a short leash (distribution + headline A/B); phase timers of the split kernels
OUT=gpurun_out/r06_06; mkdir -p $OUT
for cg in 1 0; do DC_ADJ_CG=$cg timeout 200 python tools/r06_ab/r06_cgdist.py > $OUT/cgdist_$cg.log 2>&1; echo "DC_ADJ_CG=$cg"; tail -5 $OUT/cgdist_$cg.log; done
bb() { tag=$1; tb=$2; shift; shift; ( env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --total-batch $tb --cpu-steps 0 --tshirt 0 --secondary "" > $OUT/b_$tag.log 2>&1 ); python - "$OUT/b_$tag.log" "$tag" <<'P'
import sys,json
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); c=d['config']
    print(sys.argv[2],'value',round(d['value'],1),'ms',round(d['ms_per_step'],3),'adj',round(c['mean_adjoint_iters_per_step'],2),'adjcg',round(c.get('mean_adjoint_cg_iters_per_step',0),2),'apps',round(c.get('mean_adjoint_operator_applications_per_step',0),2),'cyc',round(c['mean_fp32_solves_per_adjoint'],2),[(k['kernel'],round(k['ms_per_step'],3)) for k in d['roofline']['kernels']])
except Exception as ex: print(sys.argv[2],'FAILED',ex); print(open(sys.argv[1]).read()[-1500:])
P
}
for i in 1 2; do
bb cg1_$i 256 DC_ADJ_CG=1
bb cg0_$i 256 DC_ADJ_CG=0
done
PH=$PWD/diffcloth_amd/lib/libdiffcloth_hip_phcl.so
( DC_LIB=$PH timeout 200 python tools/debug_fold.py 32 100 1 5 1 6 > $OUT/phases_cl.log 2>&1 ); grep -E "phases" $OUT/phases_cl.log | cut -c1-500 | tail -12
( timeout 600 python -m pytest "tests/test_gpu_bench_parity.py::test_bench_configuration_matches_oracle[256-rollouts-one-workgroup-each]" tests/test_gpu_parity.py tests/test_gpu_random_scenes.py -q -x > $OUT/parity.log 2>&1 ); echo "parity rc=$?"; grep -E "passed|failed" $OUT/parity.log | tail -2
