#!/bin/bash
# round 6, call 18: early hand-over adopted — hat / dress-7742 tests (one workgroup and split), bench hat; then the closing record
OUT=gpurun_out/r06_18; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_garments10k.py -q -x > $OUT/tests.log 2>&1 ); echo "tests rc=$?"; grep -E "passed|failed" $OUT/tests.log | tail -1
bash tools/round_final.sh r06
