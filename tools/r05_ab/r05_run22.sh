# as r05_run21.sh with the wave sums in fp64 through DPP (wave_sum_d)
OUT=gpurun_out/r05_run22; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
OLD=$PWD/diffcloth_amd/lib/libdiffcloth_hip_oldsum.so
for i in 1 2; do
timeout 200 python -u tools/bench_configs.py "tshirt x256" "C5 sock" "perfFabric" "C3 hat" "dress (3634" > $OUT/cfg_new$i.log 2>&1; grep -E "rollout-steps" $OUT/cfg_new$i.log | sed 's/^/new: /'
DC_LIB=$OLD timeout 200 python -u tools/bench_configs.py "tshirt x256" "C5 sock" "perfFabric" "C3 hat" "dress (3634" > $OUT/cfg_old$i.log 2>&1; grep -E "rollout-steps" $OUT/cfg_old$i.log | sed 's/^/old: /'
done
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_new.log 2>&1; tail -1 $OUT/bench_new.log | cut -c1-330
DC_LIB=$OLD timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_old.log 2>&1; tail -1 $OUT/bench_old.log | cut -c1-330
