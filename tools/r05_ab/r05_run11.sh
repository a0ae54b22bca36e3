set -x
OUT=gpurun_out/r05l; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 20 --warmup 5 --cpu-steps 0 --tshirt 0 --total-batch 32"
L=$PWD/diffcloth_amd/lib
DC_LIB=$L/libdiffcloth_hip_r04.so timeout 300 $B > $OUT/bench_b32_r04.log 2>&1
timeout 300 $B > $OUT/bench_b32.log 2>&1
DC_LIB=$L/libdiffcloth_hip_r04.so timeout 300 $B > $OUT/bench_b32_r04_2.log 2>&1
timeout 300 $B > $OUT/bench_b32_2.log 2>&1
DC_LIB=$L/libdiffcloth_hip_phcl.so timeout 300 python bench.py --steps 2 --warmup 5 --cpu-steps 0 --tshirt 0 --total-batch 32 > $OUT/bench_phcl.log 2>&1; grep "phases cl part [07]" $OUT/bench_phcl.log | tail -4
DC_LIB=$L/libdiffcloth_hip_phacl.so timeout 300 python bench.py --steps 2 --warmup 5 --cpu-steps 0 --tshirt 0 --total-batch 32 > $OUT/bench_phacl.log 2>&1; grep "phases" $OUT/bench_phacl.log | tail -3
python tools/bench_summary.py $OUT
