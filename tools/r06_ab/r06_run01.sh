#!/bin/bash
# round 6, call 1: baseline of the round — whole GPU suite with the parity ledger, then the default bench line (with the new secondary configurations)
OUT=gpurun_out/r06_01; mkdir -p $OUT
export DC_LEDGER=1 DC_LEDGER_PATH=$OUT/parity_ledger.json
( timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/suite.log 2>&1 ); echo "suite rc=$?"
grep -E "passed|failed|error" $OUT/suite.log | tail -3
grep -E "^\[ledger\]|\[tshirt L-BFGS\]" $OUT/suite.log | cut -c1-600
( time timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err ); tail -3 $OUT/bench.err
python - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/r06_01/bench.json') if l.startswith('{')][-1])
print('value',round(d['value'],1),'ms',round(d['ms_per_step'],2),[(k['kernel'],round(k['ms_per_step'],2)) for k in d['roofline']['kernels']])
for s in d.get('secondary_configs',[]): print({k:(round(v,2) if isinstance(v,float) else v) for k,v in s.items()})
print(d.get('secondary'))
P
