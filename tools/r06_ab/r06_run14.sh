#!/bin/bash
# round 6, call 14: closing record on the final library (whole suite + ledger + default bench line), then the 32 x 8 profile again on that library
bash tools/round_final.sh r06
bash tools/profile_round.sh r06b --steps 20 --warmup 5 --total-batch 32 --secondary none 2>&1 | grep -E "^k_" | cut -c1-700
