# five rows in flight in the adjoint's three-vector passes (base) against four (-DDC_ADJ_VB3=4), same box
OUT=gpurun_out/r05_run24; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
OLD=$PWD/diffcloth_amd/lib/libdiffcloth_hip_vb4.so
for i in 1 2; do
timeout 200 python bench.py --steps 20 --warmup 5 --cpu-steps 0 --tshirt 0 > $OUT/bench_new$i.log 2>&1; tail -1 $OUT/bench_new$i.log | cut -c1-250 | sed 's/^/new: /'
DC_LIB=$OLD timeout 200 python bench.py --steps 20 --warmup 5 --cpu-steps 0 --tshirt 0 > $OUT/bench_old$i.log 2>&1; tail -1 $OUT/bench_old$i.log | cut -c1-250 | sed 's/^/old: /'
done
timeout 200 python -u tools/bench_configs.py "tshirt x256" "C5 sock" "perfFabric" "dress (3634" > $OUT/cfg_new.log 2>&1; grep -E "rollout-steps" $OUT/cfg_new.log | cut -c1-170 | sed 's/^/new: /'
DC_LIB=$OLD timeout 200 python -u tools/bench_configs.py "tshirt x256" "C5 sock" "perfFabric" "dress (3634" > $OUT/cfg_old.log 2>&1; grep -E "rollout-steps" $OUT/cfg_old.log | cut -c1-170 | sed 's/^/old: /'
