set -x
OUT=gpurun_out/r05u; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_schedules.py tests/test_gpu_forces.py tests/test_gpu_pymodule.py -q -x -k "scheduled or forces or force or device_resident or falloff" > $OUT/forces.log 2>&1; tail -4 $OUT/forces.log
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-steps 0 --tshirt 0 > $OUT/bench_main.log 2>&1
python tools/bench_summary.py $OUT
