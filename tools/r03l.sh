OUT=gpurun_out/r03l; mkdir -p $OUT
run() { name=$1; shift; ( timeout ${TMO:-400} "$@" > $OUT/$name.log 2>&1; echo "rc=$?" >> $OUT/$name.log ); echo "=== $name"; grep -h "^\[config\] B=\|passed\|failed\|rc=" $OUT/$name.log | cut -c1-300 | tail -${TAILN:-3}; }
TMO=900 TAILN=12 run cfg python -m pytest tests/test_gpu_configs.py -q -s
timeout 1200 bash tools/profile_round.sh r03a --steps 20 --warmup 5 > $OUT/prof256.log 2>&1; tail -4 $OUT/prof256.log | cut -c1-300
timeout 1000 bash tools/profile_round.sh r03b --steps 20 --warmup 5 --total-batch 32 --tshirt 0 --cpu-steps 0 > $OUT/prof32.log 2>&1; tail -4 $OUT/prof32.log | cut -c1-300
