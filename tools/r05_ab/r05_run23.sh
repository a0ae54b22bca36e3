# final state of the round: full GPU suite, the other configurations, the bench line
OUT=gpurun_out/r05_run23; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 300 python -u tools/bench_configs.py > $OUT/configs.log 2>&1; grep -E "rollout-steps" $OUT/configs.log | cut -c1-200
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log | cut -c1-330
timeout 1200 python -m pytest tests -m gpu -q -x > $OUT/gpu_tests.log 2>&1; tail -6 $OUT/gpu_tests.log
