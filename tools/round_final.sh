#!/bin/bash
# Closing GPU check of a round (run on the GPU box through gpurun):  tools/round_final.sh TAG
# The WHOLE -m gpu suite on the library as it is, with the parity ledger on, then the default bench line. Records the sha256 of the
# libraries the suite ran on in gpurun_out/TAG_gpu_suite.txt; `python tools/check_round_final.py profiles/TAG_gpu_suite.txt` (CPU, here)
# refuses — exit 1 — when the committed record is not of the library in the tree or the suite was not green (round 5's last
# kernel change was verified on 84 of 135 tests).
TAG=${1:-r06}
OUT=gpurun_out; mkdir -p $OUT
REC=$OUT/${TAG}_gpu_suite.txt
export DC_LEDGER=1 DC_LEDGER_PATH=$OUT/${TAG}_parity_ledger.json
rm -f $DC_LEDGER_PATH
{
  echo "# closing GPU suite of round $TAG: python -m pytest tests -m gpu -q   ($(date -u +%Y-%m-%dT%H:%M:%SZ))"
  echo "lib_sha256 $(sha256sum diffcloth_amd/lib/libdiffcloth_hip.so | cut -d' ' -f1) diffcloth_amd/lib/libdiffcloth_hip.so"
  echo "pymodule_sha256 $(sha256sum diffcloth_amd/lib/diffcloth_py*.so | cut -d' ' -f1) diffcloth_py"
  echo "oracle_sha256 $(sha256sum oracle/liboracle.so 2>/dev/null | cut -d' ' -f1) oracle/liboracle.so"
} > $REC
( timeout 2400 python -m pytest tests -m gpu -q -s > $OUT/${TAG}_suite.log 2>&1 ); rc=$?
echo "suite_rc $rc" >> $REC
grep -E "^[0-9]+ (passed|failed)|passed|failed" $OUT/${TAG}_suite.log | tail -1 | sed 's/^/summary /' >> $REC
grep -E "^\[ledger\]" $OUT/${TAG}_suite.log | cut -c1-2000 >> $REC
grep -E "^\[tshirt L-BFGS\]" $OUT/${TAG}_suite.log | cut -c1-300 >> $REC
grep -E "^FAILED|^ERROR" $OUT/${TAG}_suite.log | cut -c1-300 >> $REC
cat $REC
( timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench.err ); echo "bench rc=$?"
tail -c 600 $OUT/${TAG}_bench_line.json | head -c 300; echo
