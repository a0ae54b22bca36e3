#!/bin/bash
# round 6, call 8: round profiles (256 x 1 and 32 x 8) + L2 counters
bash tools/profile_round.sh r06a --steps 20 --warmup 5 --secondary none 2>&1 | tail -25
bash tools/profile_round.sh r06b --steps 20 --warmup 5 --total-batch 32 --secondary none 2>&1 | tail -25
bash tools/pmc_l2.sh 2>&1 | tail -6
