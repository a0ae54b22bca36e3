OUT=gpurun_out/${1:-r05t}; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1; tail -8 $OUT/gpu_tests.log
