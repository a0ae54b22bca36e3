bash tools/profile_round.sh r03c --steps 20 --warmup 5 > gpurun_out/r03c_prof.log 2>&1
bash tools/profile_round.sh r03d --steps 20 --warmup 5 --total-batch 32 > gpurun_out/r03d_prof.log 2>&1
mkdir -p gpurun_out/r03z
( cd $GRAFT_REPO_ROOT && timeout 900 python tools/bench_configs.py > gpurun_out/r03z/configs.log 2>&1 ); tail -9 gpurun_out/r03z/configs.log | cut -c1-260
tail -5 gpurun_out/prof_r03c/roofline_summary.txt | cut -c1-400
