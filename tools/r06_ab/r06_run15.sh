#!/bin/bash
# round 6, call 15: the fall-back path of the split forward's self friction (part 0 alone, what a rollout whose parts are not on one XCD gets) on the
# split tests, then the closing record on the final library
OUT=gpurun_out/r06_15; mkdir -p $OUT
( DC_SELF_REDUNDANT=0 timeout 600 python -m pytest tests/test_gpu_cluster.py "tests/test_gpu_bench_parity.py::test_bench_configuration_matches_oracle[32-rollouts-split-over-8-workgroups]" -q -x > $OUT/red0.log 2>&1 ); echo "DC_SELF_REDUNDANT=0 rc=$?"; grep -E "passed|failed" $OUT/red0.log | tail -1
bash tools/round_final.sh r06
