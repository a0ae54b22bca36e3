#!/bin/bash
# round 6, call 2: split kernels. A/B on one box: single-exchange CG on/off (DC_SXCG), the 512-thread split adjoint (variant library), then the
# tests that cover the split paths
OUT=gpurun_out/r06_02; mkdir -p $OUT
L512=$PWD/diffcloth_amd/lib/libdiffcloth_hip_adjcl512.so
b32() { tag=$1; shift; ( env "$@" timeout 200 python bench.py --steps 20 --warmup 5 --total-batch 32 --cpu-steps 0 --tshirt 0 --secondary "" > $OUT/b32_$tag.log 2>&1 ); python - "$OUT/b32_$tag.log" "$tag" <<'P'
import sys,json
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); c=d['config']
    print(sys.argv[2],'value',round(d['value'],1),'ms',round(d['ms_per_step'],3),'pd',round(c['mean_pd_iters_per_step'],2),'cg/pd',round(c['mean_cg_iters_per_pd_iter'],3),'adj',round(c['mean_adjoint_iters_per_step'],2),[(k['kernel'],round(k['ms_per_step'],3)) for k in d['roofline']['kernels']])
except Exception as ex: print(sys.argv[2],'FAILED',ex); print(open(sys.argv[1]).read()[-1500:])
P
}
for i in 1 2; do
b32 sx1_$i DC_SXCG=1
b32 sx0_$i DC_SXCG=0
b32 sx1_adj512_$i DC_SXCG=1 DC_LIB=$L512
done
( timeout 900 python -m pytest tests/test_gpu_cluster.py tests/test_gpu_bench_parity.py -q -x -s > $OUT/tests.log 2>&1 ); echo "tests rc=$?"; grep -E "passed|failed" $OUT/tests.log | tail -2
grep -E "^\[bench parity\] worst" $OUT/tests.log | cut -c1-400
( DC_LIB=$L512 timeout 600 python -m pytest tests/test_gpu_cluster.py "tests/test_gpu_bench_parity.py::test_bench_configuration_matches_oracle[32-rollouts-split-over-8-workgroups]" -q -x -s > $OUT/tests512.log 2>&1 ); echo "tests512 rc=$?"; grep -E "passed|failed" $OUT/tests512.log | tail -2
