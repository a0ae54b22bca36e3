# as r05_run25.sh with the list as the ONLY sparse form of the 1024-thread adjoint kernels (no fit -> y over all vertices), then the tests that
# cover the changed paths: all-in-contact sheet (no fit), bench parity, self contacts, primitives
OUT=gpurun_out/r05_run26; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
OLD=$PWD/diffcloth_amd/lib/libdiffcloth_hip_base.so
for i in 1 2; do
timeout 100 python bench.py --steps 20 --warmup 5 --cpu-steps 0 --tshirt 0 > $OUT/bench_new$i.log 2>&1; tail -1 $OUT/bench_new$i.log | cut -c1-250 | sed 's/^/new: /'
DC_LIB=$OLD timeout 100 python bench.py --steps 20 --warmup 5 --cpu-steps 0 --tshirt 0 > $OUT/bench_old$i.log 2>&1; tail -1 $OUT/bench_old$i.log | cut -c1-250 | sed 's/^/old: /'
done
timeout 100 python -u tools/bench_configs.py "perfFabric" "dress (3634" > $OUT/cfg_new.log 2>&1; grep -E "rollout-steps" $OUT/cfg_new.log | cut -c1-170 | sed 's/^/new: /'
DC_LIB=$OLD timeout 100 python -u tools/bench_configs.py "perfFabric" "dress (3634" > $OUT/cfg_old.log 2>&1; grep -E "rollout-steps" $OUT/cfg_old.log | cut -c1-170 | sed 's/^/old: /'
timeout 225 python -m pytest tests/test_gpu_garments10k.py::test_perf_fabric_96x96_sliding_on_the_slope_plane tests/test_gpu_selfcontact.py tests/test_gpu_bench_parity.py tests/test_gpu_primitives.py tests/test_gpu_parity.py -v -x 2>&1 | grep -E "PASSED|FAILED|ERROR|passed|failed" | cut -c1-150 > $OUT/tests.log; tail -40 $OUT/tests.log
