#!/bin/bash
# round 6, call 4: adjoint CG-first (fixed) A/B on the headline; secondary configurations with the single-reduction CG built into the one-workgroup
# fp32-plane instances (variant library) against the default; bench parity at the shipped and at a looser inner tolerance
OUT=gpurun_out/r06_04; mkdir -p $OUT
SX=$PWD/diffcloth_amd/lib/libdiffcloth_hip_pksx.so
bb() { tag=$1; tb=$2; sec=$3; shift; shift; shift; ( env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --total-batch $tb --cpu-steps 0 --tshirt 0 --secondary "$sec" > $OUT/b_$tag.log 2>&1 ); python - "$OUT/b_$tag.log" "$tag" <<'P'
import sys,json
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); c=d['config']
    print(sys.argv[2],'value',round(d['value'],1),'ms',round(d['ms_per_step'],3),'pd',round(c['mean_pd_iters_per_step'],2),'cg/pd',round(c['mean_cg_iters_per_pd_iter'],3),'adj',round(c['mean_adjoint_iters_per_step'],2),'adjcg',round(c.get('mean_adjoint_cg_iters_per_step',0),2),'apps',round(c.get('mean_adjoint_operator_applications_per_step',0),2),'cyc',round(c['mean_fp32_solves_per_adjoint'],2),[(k['kernel'],round(k['ms_per_step'],3)) for k in d['roofline']['kernels']])
    for s in d.get('secondary_configs',[]): print('   ',s.get('workload','?')[:28],'r-steps/s',round(s.get('rollout_steps_per_s',0),1),'fwd',round(s.get('fwd_ms_per_step',0),2),'bwd',round(s.get('bwd_ms_per_step',0),2),'pd',round(s.get('mean_pd_iters_per_step',0),1),'cg/pd',round(s.get('mean_cg_iters_per_pd_iter',0),2),'adj',round(s.get('mean_adjoint_iters_per_step',0),1),'adjcg',round(s.get('mean_adjoint_cg_iters_per_step',0),1),'f64',round(s.get('fp64_fallback_iters_per_step',0),1),s.get('error',''))
except Exception as ex: print(sys.argv[2],'FAILED',ex); print(open(sys.argv[1]).read()[-1500:])
P
}
for i in 1 2; do
bb cg1_$i 256 "" DC_ADJ_CG=1
bb cg0_$i 256 "" DC_ADJ_CG=0
done
bb sec_default 256 "hat,sock,dress,perf_fabric" DC_ADJ_CG=1
bb sec_pksx 256 "hat,sock,dress,perf_fabric" DC_ADJ_CG=1 DC_LIB=$SX
bb sec_cg0 256 "hat,sock,dress,perf_fabric" DC_ADJ_CG=0
bb b32 32 "" DC_ADJ_CG=1
( timeout 600 python -m pytest "tests/test_gpu_bench_parity.py::test_bench_configuration_matches_oracle[256-rollouts-one-workgroup-each]" -q -x -s > $OUT/parity.log 2>&1 ); echo "parity rc=$?"; grep -E "passed|failed" $OUT/parity.log | tail -2
grep -E "^\[bench parity\] worst" $OUT/parity.log | cut -c1-400
( BENCH_CG_TOL=3e-4 timeout 600 python -m pytest "tests/test_gpu_bench_parity.py::test_bench_configuration_matches_oracle[256-rollouts-one-workgroup-each]" -q -x -s > $OUT/parity_3e4.log 2>&1 ); echo "parity(3e-4) rc=$?"; grep -E "passed|failed|^E  " $OUT/parity_3e4.log | tail -4
grep -E "^\[bench parity\] worst" $OUT/parity_3e4.log | cut -c1-400
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_cluster.py -q -x > $OUT/core.log 2>&1 ); echo "core rc=$?"; grep -E "passed|failed" $OUT/core.log | tail -2
