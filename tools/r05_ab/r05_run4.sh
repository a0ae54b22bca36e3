set -x
OUT=gpurun_out/r05d; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 20 --warmup 5 --cpu-steps 0 --tshirt 0"
L=$PWD/diffcloth_amd/lib
DC_LIB=$L/libdiffcloth_hip_nopf.so timeout 300 $B > $OUT/bench_nopf.log 2>&1
DC_LIB=$L/libdiffcloth_hip_pf.so timeout 300 $B > $OUT/bench_pf.log 2>&1
DC_LIB=$L/libdiffcloth_hip_nopf.so timeout 300 $B > $OUT/bench_nopf2.log 2>&1
DC_LIB=$L/libdiffcloth_hip_pf.so timeout 300 $B > $OUT/bench_pf2.log 2>&1
DC_LIB=$L/libdiffcloth_hip_phpf.so timeout 300 python bench.py --steps 3 --warmup 5 --cpu-steps 0 --tshirt 0 > $OUT/bench_ph.log 2>&1; grep "phases pk" $OUT/bench_ph.log | tail -2
DC_LIB=$L/libdiffcloth_hip_pf.so timeout 900 python -m pytest tests/test_gpu_bench_parity.py -q -x > $OUT/parity.log 2>&1; tail -3 $OUT/parity.log
python tools/bench_summary.py $OUT
