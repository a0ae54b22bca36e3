bash tools/round_check.sh 2>&1 | tail -6
bash tools/profile_round.sh r03e --steps 20 --warmup 5 > gpurun_out/r03e_prof.log 2>&1
bash tools/profile_round.sh r03f --steps 20 --warmup 5 --total-batch 32 > gpurun_out/r03f_prof.log 2>&1
tail -4 gpurun_out/prof_r03e/roofline_summary.txt | cut -c1-300
( timeout 600 python tools/bench_configs.py > gpurun_out/round_check/r03g_other_configs.txt 2>&1 ); grep "tshirt\|C5 sock\|^dress\|C3 hat" gpurun_out/round_check/r03g_other_configs.txt | cut -c1-200
