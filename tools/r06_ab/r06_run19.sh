#!/bin/bash
# round 6, call 19: fp32-plane instances of the one-workgroup forward kernel with the reductions' leading barriers removed (as the H16 instance has them):
# secondary configurations and the T-shirt evaluation, same-box A/B against the previous build; then the tests of those instances
OUT=gpurun_out/r06_19; mkdir -p $OUT
PREV=$PWD/diffcloth_amd/lib/libdiffcloth_hip_prev.so
bb() { tag=$1; shift; ( env "$@" timeout 300 python bench.py --steps 5 --warmup 3 --cpu-steps 0 --tshirt 1 --secondary sock,dress,perf_fabric > $OUT/b_$tag.log 2>&1 ); python - "$OUT/b_$tag.log" "$tag" <<'P'
import sys,json
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    print(sys.argv[2], 'tshirt eval0 fwd', round(d['secondary'].get('forward_s',0),3), 'fwd+bwd', round(d['secondary'].get('forward_plus_backward_s',0),3), 'loss', d['secondary'].get('loss'))
    for s in d.get('secondary_configs',[]): print('   ',s.get('workload','?')[:28],'r-steps/s',round(s.get('rollout_steps_per_s',0),1),'fwd',round(s.get('fwd_ms_per_step',0),2),'bwd',round(s.get('bwd_ms_per_step',0),2),'pd',round(s.get('mean_pd_iters_per_step',0),1),'cg/pd',round(s.get('mean_cg_iters_per_pd_iter',0),2),s.get('error',''))
except Exception as ex: print(sys.argv[2],'FAILED',ex); print(open(sys.argv[1]).read()[-1200:])
P
}
bb new_1 X=1
bb prev_1 DC_LIB=$PREV
bb new_2 X=1
bb prev_2 DC_LIB=$PREV
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fallbacks.py tests/test_gpu_random_scenes.py "tests/test_gpu_configs.py" -q -x > $OUT/tests.log 2>&1 ); echo "tests rc=$?"; grep -E "passed|failed" $OUT/tests.log | tail -1
