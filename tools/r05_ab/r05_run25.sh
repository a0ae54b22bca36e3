# adjoint: y of the contact vertices from an LDS list at the windows' staging (base) against loading the y plane for every span vertex (HEAD~: _base.so)
OUT=gpurun_out/r05_run25; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
OLD=$PWD/diffcloth_amd/lib/libdiffcloth_hip_base.so
for i in 1 2; do
timeout 200 python bench.py --steps 20 --warmup 5 --cpu-steps 0 --tshirt 0 > $OUT/bench_new$i.log 2>&1; tail -1 $OUT/bench_new$i.log | cut -c1-250 | sed 's/^/new: /'
DC_LIB=$OLD timeout 200 python bench.py --steps 20 --warmup 5 --cpu-steps 0 --tshirt 0 > $OUT/bench_old$i.log 2>&1; tail -1 $OUT/bench_old$i.log | cut -c1-250 | sed 's/^/old: /'
done
timeout 200 python -u tools/bench_configs.py "perfFabric" "dress (3634" > $OUT/cfg_new.log 2>&1; grep -E "rollout-steps" $OUT/cfg_new.log | cut -c1-170 | sed 's/^/new: /'
DC_LIB=$OLD timeout 200 python -u tools/bench_configs.py "perfFabric" "dress (3634" > $OUT/cfg_old.log 2>&1; grep -E "rollout-steps" $OUT/cfg_old.log | cut -c1-170 | sed 's/^/old: /'
timeout 200 python -m pytest tests/test_gpu_bench_parity.py -q -x > $OUT/parity.log 2>&1; tail -3 $OUT/parity.log
