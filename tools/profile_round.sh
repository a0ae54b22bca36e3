#!/bin/bash
# Round profile (run on the GPU box through gpurun): rocprofv3 kernel statistics of the default bench command, then
# HBM-side byte counters in separate passes (MI355X_MICROARCH.md: one --pmc counter group per pass, never together
# with the sys/hip trace domains), the same passes over the calibration kernels (tools/pmc_calib.hip), and the summary
# tools/pmc_traffic.py makes of them. Output: gpurun_out/prof_$1/...
TAG=${1:-r01f}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o x --output-format csv -- python $R/bench.py > $OUT/bench_under_rocprof.log 2>&1
grep metric $OUT/bench_under_rocprof.log > $OUT/bench_line_under_rocprof.json
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_$C -o x --output-format csv -- python $R/bench.py --cpu-steps 0 > $OUT/pmc_$C.log 2>&1
  rocprofv3 --kernel-trace --pmc $C -d $OUT/calib_$C -o x --output-format csv -- $R/tools/bin/pmc_calib > $OUT/calib_$C.log 2>&1
done
python $R/bench.py > $OUT/bench_plain.log 2>&1
grep metric $OUT/bench_plain.log > $OUT/bench_line.json
python $R/tools/pmc_traffic.py $OUT > $OUT/traffic_summary.txt 2>&1
cat $OUT/traffic_summary.txt
find $OUT -name "*_counter_collection.csv" -size +2M -delete      # keep the merge-back small
ls -R $OUT | head -40
