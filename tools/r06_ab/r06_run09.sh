#!/bin/bash
# round 6, call 9: deflated global-memory forward kernel on the 17 562-vertex dress; fall-back kernel sets; new / tightened tests
OUT=gpurun_out/r06_09; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_garments10k.py tests/test_gpu_fallbacks.py "tests/test_gpu_fullsize.py" tests/test_gpu_pymodule.py -q -x -s > $OUT/tests.log 2>&1 ); echo "tests rc=$?"; grep -E "passed|failed" $OUT/tests.log | tail -2
grep -E "^\[dress 17 562\]|CG-first vs BiCGSTAB|^\[optimize helper\]" $OUT/tests.log | cut -c1-420
grep -E "^E  " $OUT/tests.log | head -10 | cut -c1-300
