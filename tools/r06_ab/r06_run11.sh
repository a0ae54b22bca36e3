#!/bin/bash
# round 6, call 11: the round's closing record on the final library (whole GPU suite + ledger + default bench line), then a wider parity survey at the
# bench configuration (48 rollouts x 3 steps against the fp64 oracle) into its own ledger
bash tools/round_final.sh r06
export DC_LEDGER=1 DC_LEDGER_PATH=gpurun_out/r06_bench_parity_survey.json; rm -f $DC_LEDGER_PATH
( BENCH_PARITY_N=48 timeout 1500 python -m pytest "tests/test_gpu_bench_parity.py::test_bench_configuration_matches_oracle[256-rollouts-one-workgroup-each]" -q -x -s > gpurun_out/r06_survey.log 2>&1 ); echo "survey rc=$?"
grep -E "passed|failed|^\[bench parity\] worst" gpurun_out/r06_survey.log | cut -c1-400
