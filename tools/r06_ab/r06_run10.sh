#!/bin/bash
# round 6, call 10: the deferred x update of the H16 forward kernel, A/B against the previous build; bench parity
OUT=gpurun_out/r06_10; mkdir -p $OUT
PREV=$PWD/diffcloth_amd/lib/libdiffcloth_hip_prev.so
bb() { tag=$1; shift; ( env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --cpu-steps 0 --tshirt 0 --secondary none > $OUT/b_$tag.log 2>&1 ); python - "$OUT/b_$tag.log" "$tag" <<'P'
import sys,json
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); c=d['config']
    print(sys.argv[2],'value',round(d['value'],1),'ms',round(d['ms_per_step'],3),'pd',round(c['mean_pd_iters_per_step'],2),'cg/pd',round(c['mean_cg_iters_per_pd_iter'],3),[(k['kernel'],round(k['ms_per_step'],3)) for k in d['roofline']['kernels']])
except Exception as ex: print(sys.argv[2],'FAILED',ex); print(open(sys.argv[1]).read()[-1500:])
P
}
for i in 1 2 3; do
bb new_$i X=1
bb prev_$i DC_LIB=$PREV
done
( timeout 600 python -m pytest "tests/test_gpu_bench_parity.py::test_bench_configuration_matches_oracle[256-rollouts-one-workgroup-each]" -q -x -s > $OUT/parity.log 2>&1 ); echo "parity rc=$?"; grep -E "passed|failed" $OUT/parity.log | tail -2
grep -E "^\[bench parity\] worst" $OUT/parity.log | cut -c1-400
