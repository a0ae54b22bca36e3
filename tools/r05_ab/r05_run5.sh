set -x
OUT=gpurun_out/r05e; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 20 --warmup 5 --cpu-steps 0 --tshirt 0"
L=$PWD/diffcloth_amd/lib
$L/soffset_test > $OUT/soffset.log 2>&1; cat $OUT/soffset.log
timeout 300 $B > $OUT/bench_prev.log 2>&1
DC_LIB=$L/libdiffcloth_hip_nopf.so timeout 300 $B > $OUT/bench_nopf.log 2>&1
DC_LIB=$L/libdiffcloth_hip_pf.so timeout 300 $B > $OUT/bench_pf.log 2>&1
DC_LIB=$L/libdiffcloth_hip_nopf.so timeout 300 $B > $OUT/bench_nopf2.log 2>&1
DC_LIB=$L/libdiffcloth_hip_pf.so timeout 300 $B > $OUT/bench_pf2.log 2>&1
DC_LIB=$L/libdiffcloth_hip_phpf.so timeout 300 python bench.py --steps 3 --warmup 5 --cpu-steps 0 --tshirt 0 > $OUT/bench_phpf.log 2>&1; grep "phases pk" $OUT/bench_phpf.log | tail -1
DC_LIB=$L/libdiffcloth_hip_phnopf.so timeout 300 python bench.py --steps 3 --warmup 5 --cpu-steps 0 --tshirt 0 > $OUT/bench_phnopf.log 2>&1; grep "phases pk" $OUT/bench_phnopf.log | tail -1
DC_LIB=$L/libdiffcloth_hip_pf.so timeout 900 python -m pytest tests/test_gpu_bench_parity.py -q -x > $OUT/parity.log 2>&1; tail -3 $OUT/parity.log
python tools/bench_summary.py $OUT
