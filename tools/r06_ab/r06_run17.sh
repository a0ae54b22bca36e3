#!/bin/bash
# round 6, call 17: early hand-over of a stalled fp32 BiCGSTAB solve where the fp64 fall-back has the coarse level (hat, dress-7742): variants against the default
OUT=gpurun_out/r06_17; mkdir -p $OUT
bb() { tag=$1; shift; ( env "$@" timeout 300 python bench.py --steps 3 --warmup 2 --cpu-steps 0 --tshirt 0 --secondary hat > $OUT/b_$tag.log 2>&1 ); python - "$OUT/b_$tag.log" "$tag" <<'P'
import sys,json
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    for s in d.get('secondary_configs',[])[:1]: print(sys.argv[2], s.get('workload','?')[:28],'r-steps/s',round(s.get('rollout_steps_per_s',0),1),'fwd',round(s.get('fwd_ms_per_step',0),2),'bwd',round(s.get('bwd_ms_per_step',0),2),'adj',round(s.get('mean_adjoint_iters_per_step',0),1),'slowest',round(s.get('slowest_rollout_adjoint_iters_per_step',0),1),'f64',round(s.get('fp64_fallback_iters_per_step',0),1),'conv',s.get('adjoint_converged_fraction'),s.get('error',''))
except Exception as ex: print(sys.argv[2],'FAILED',ex); print(open(sys.argv[1]).read()[-1200:])
P
}
bb default X=1
for n in 30 80; do bb stall$n DC_LIB=$PWD/diffcloth_amd/lib/libdiffcloth_hip_stall$n.so; done
for n in 30 80; do
( DC_LIB=$PWD/diffcloth_amd/lib/libdiffcloth_hip_stall$n.so timeout 900 python -m pytest tests/test_gpu_configs.py -q -x -s -k "hat or 7742_vertices" > $OUT/tests_stall$n.log 2>&1 ); echo "stall$n tests rc=$?"; grep -E "passed|failed" $OUT/tests_stall$n.log | tail -1
grep -E "BiCGSTAB [0-9]+ in|adjoint: converged" $OUT/tests_stall$n.log | cut -c1-260 | head -12
done
( timeout 900 python -m pytest tests/test_gpu_configs.py -q -x -s -k "hat or 7742_vertices" > $OUT/tests_default.log 2>&1 ); echo "default tests rc=$?"; grep -E "BiCGSTAB [0-9]+ in|adjoint: converged" $OUT/tests_default.log | cut -c1-260 | head -12
