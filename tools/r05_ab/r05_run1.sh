set -x
OUT=gpurun_out/r05a; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 20 --warmup 5 --cpu-steps 0 --tshirt 0"
DC_LIB=$PWD/diffcloth_amd/lib/libdiffcloth_hip_r04.so timeout 300 $B > $OUT/bench_r04.log 2>&1; true
timeout 300 $B > $OUT/bench_new.log 2>&1; tail -c 600 $OUT/bench_new.log
DC_LIB=$PWD/diffcloth_amd/lib/libdiffcloth_hip_r04.so timeout 300 $B > $OUT/bench_r04_2.log 2>&1
timeout 300 $B > $OUT/bench_new_2.log 2>&1
DC_LIB=$PWD/diffcloth_amd/lib/libdiffcloth_hip_ph.so timeout 300 python bench.py --steps 3 --warmup 5 --cpu-steps 0 --tshirt 0 > $OUT/bench_ph.log 2>&1; grep "phases pk" $OUT/bench_ph.log | tail -3
timeout 900 python -m pytest tests/test_gpu_bench_parity.py -q -x > $OUT/parity.log 2>&1; tail -5 $OUT/parity.log
python tools/bench_summary.py $OUT
