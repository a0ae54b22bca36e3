#!/bin/bash
# round 6, call 13: split forward kernel — layered self friction evaluated redundantly by every part (DC_SELF_REDUNDANT=1, default) against part 0
# alone between two cross-part barriers (=0), same box; then the tests of the split paths
OUT=gpurun_out/r06_13; mkdir -p $OUT
bb() { tag=$1; tb=$2; shift; shift; ( env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --total-batch $tb --cpu-steps 0 --tshirt 0 --secondary none > $OUT/b_$tag.log 2>&1 ); python - "$OUT/b_$tag.log" "$tag" <<'P'
import sys,json
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); c=d['config']
    print(sys.argv[2],'value',round(d['value'],1),'ms',round(d['ms_per_step'],3),'pd',round(c['mean_pd_iters_per_step'],2),'cg/pd',round(c['mean_cg_iters_per_pd_iter'],3),'self',round(c['mean_self_contacts_per_step'],1),[(k['kernel'],round(k['ms_per_step'],3)) for k in d['roofline']['kernels']], c['gradients_finite'])
except Exception as ex: print(sys.argv[2],'FAILED',ex); print(open(sys.argv[1]).read()[-1200:])
P
}
for i in 1 2; do
bb red1_$i 32 DC_SELF_REDUNDANT=1
bb red0_$i 32 DC_SELF_REDUNDANT=0
done
bb b64 64 DC_SELF_REDUNDANT=1
bb b128 128 DC_SELF_REDUNDANT=1
( timeout 900 python -m pytest tests/test_gpu_cluster.py "tests/test_gpu_bench_parity.py::test_bench_configuration_matches_oracle[32-rollouts-split-over-8-workgroups]" tests/test_gpu_selfcontact.py "tests/test_gpu_configs.py::test_c4_dress_self_contact_batch" -q -x -s > $OUT/tests.log 2>&1 ); echo "tests rc=$?"; grep -E "passed|failed" $OUT/tests.log | tail -2
grep -E "^\[bench parity\] worst|^E  " $OUT/tests.log | cut -c1-330 | head
