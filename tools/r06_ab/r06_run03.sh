#!/bin/bash
# round 6, call 3: headline A/B on one box — round-5 library, new library with DC_ADJ_CG=0 / 1 — then bench parity, the split bench and the suites that
# cover the changed kernels (forward H16 single-reduction CG, adjoint CG-first, split single-exchange CG + skipped chunks)
OUT=gpurun_out/r06_03; mkdir -p $OUT
R05=$PWD/diffcloth_amd/lib/libdiffcloth_hip_r05.so
bb() { tag=$1; tb=$2; shift; shift; ( env "$@" timeout 200 python bench.py --steps 20 --warmup 5 --total-batch $tb --cpu-steps 0 --tshirt 0 --secondary "" > $OUT/b_$tag.log 2>&1 ); python - "$OUT/b_$tag.log" "$tag" <<'P'
import sys,json
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); c=d['config']
    print(sys.argv[2],'value',round(d['value'],1),'ms',round(d['ms_per_step'],3),'pd',round(c['mean_pd_iters_per_step'],2),'cg/pd',round(c['mean_cg_iters_per_pd_iter'],3),'adj',round(c['mean_adjoint_iters_per_step'],2),'adjcg',round(c.get('mean_adjoint_cg_iters_per_step',0),2),'cyc',round(c['mean_fp32_solves_per_adjoint'],2),[(k['kernel'],round(k['ms_per_step'],3)) for k in d['roofline']['kernels']])
except Exception as ex: print(sys.argv[2],'FAILED',ex); print(open(sys.argv[1]).read()[-1500:])
P
}
for i in 1 2; do
bb new_$i 256 DC_ADJ_CG=1
bb new_nocg_$i 256 DC_ADJ_CG=0
bb r05_$i 256 DC_LIB=$R05
done
bb b32 32 DC_ADJ_CG=1
bb b32_r05 32 DC_LIB=$R05
( timeout 900 python -m pytest tests/test_gpu_bench_parity.py tests/test_gpu_cluster.py -q -x -s > $OUT/parity.log 2>&1 ); echo "parity rc=$?"; grep -E "passed|failed" $OUT/parity.log | tail -2
grep -E "^\[bench parity\] worst" $OUT/parity.log | cut -c1-400
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_selfcontact.py tests/test_gpu_primitives.py tests/test_gpu_random_scenes.py tests/test_gpu_garments10k.py -q -x > $OUT/core.log 2>&1 ); echo "core rc=$?"; grep -E "passed|failed" $OUT/core.log | tail -2
