#!/bin/bash
# round 6, call 12: compiler scheduling strategies (-mllvm -amdgpu-sched-strategy=...) for the bench forward / adjoint instances, same-box A/B
OUT=gpurun_out/r06_12; mkdir -p $OUT
bb() { tag=$1; shift; ( env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --cpu-steps 0 --tshirt 0 --secondary none > $OUT/b_$tag.log 2>&1 ); python - "$OUT/b_$tag.log" "$tag" <<'P'
import sys,json
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); c=d['config']
    print(sys.argv[2],'value',round(d['value'],1),'ms',round(d['ms_per_step'],3),'pd',round(c['mean_pd_iters_per_step'],2),'apps',round(c.get('mean_adjoint_operator_applications_per_step',0),2),[(k['kernel'],round(k['ms_per_step'],3)) for k in d['roofline']['kernels']], 'finite', c['gradients_finite'])
except Exception as ex: print(sys.argv[2],'FAILED',ex); print(open(sys.argv[1]).read()[-800:])
P
}
bb default_1 X=1
for n in gcnmaxilp gcnmaxmemoryclause gcniterativeilp gcniterativeminreg; do bb $n DC_LIB=$PWD/diffcloth_amd/lib/libdiffcloth_hip_$n.so; done
bb default_2 X=1
