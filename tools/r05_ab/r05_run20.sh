# barrier-light reductions in all packet-forward instances + per-slot DPP reductions in the fp32 BiCGSTAB: other configurations, bench, full suite
OUT=gpurun_out/r05_run20; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 420 python -u tools/bench_configs.py > $OUT/configs.log 2>&1; grep -E "rollout-steps" $OUT/configs.log
timeout 300 python bench.py > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log | cut -c1-400
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/gpu_tests.log 2>&1; tail -8 $OUT/gpu_tests.log
