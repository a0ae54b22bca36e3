#!/bin/bash
# Round profile (run on the GPU box through gpurun):  tools/profile_round.sh TAG [bench.py arguments]
# rocprofv3 kernel statistics of the bench command, then counters in separate passes (MI355X_MICROARCH.md: one --pmc counter
# group per pass, never together with the sys/hip trace domains): FETCH_SIZE, WRITE_SIZE (+ the same passes over the calibration
# kernels of tools/pmc_calib.hip), two SQ groups. tools/roofline_from_pmc.py turns them into gpurun_out/prof_TAG/TAG_roofline.json
# (copy it to profiles/ to have bench.py quote it). Output: gpurun_out/prof_TAG/...
TAG=${1:-r02a}; shift
ARGS="$@"
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py $ARGS > $OUT/bench_plain.log 2>&1
grep '"metric"' $OUT/bench_plain.log | tail -1 > $OUT/bench_line.json
rocprofv3 --kernel-trace --stats -d $OUT/stats -o x --output-format csv -- python $R/bench.py $ARGS --cpu-steps 0 --tshirt 0 > $OUT/bench_under_rocprof.log 2>&1
grep '"metric"' $OUT/bench_under_rocprof.log | tail -1 > $OUT/bench_line_under_rocprof.json
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_$C -o x --output-format csv -- python $R/bench.py $ARGS --cpu-steps 0 --tshirt 0 > $OUT/pmc_$C.log 2>&1
  [ -x $R/tools/bin/pmc_calib ] && rocprofv3 --kernel-trace --pmc $C -d $OUT/calib_$C -o x --output-format csv -- $R/tools/bin/pmc_calib > $OUT/calib_$C.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $OUT/pmc_sq_b -o x --output-format csv -- python $R/bench.py $ARGS --cpu-steps 0 --tshirt 0 > $OUT/pmc_sq_b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $OUT/pmc_sq_c -o x --output-format csv -- python $R/bench.py $ARGS --cpu-steps 0 --tshirt 0 > $OUT/pmc_sq_c.log 2>&1
python $R/tools/roofline_from_pmc.py $OUT $TAG > $OUT/roofline_summary.txt 2>&1
cat $OUT/roofline_summary.txt
cp $OUT/stats/*/*kernel_stats.csv $OUT/${TAG}_kernel_stats.csv 2>/dev/null || find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_kernel_stats.csv \;
find $OUT -name "*_counter_collection.csv" -size +2M -delete      # keep the merge-back small
find $OUT -name "*kernel_trace.csv" -size +2M -delete
ls $OUT | head -40
