#!/bin/bash
# Round profile (run on the GPU box through gpurun): rocprofv3 kernel statistics of the default bench command, then
# HBM-side byte counters in separate passes (MI355X_MICROARCH.md: one --pmc counter group per pass, never together
# with the sys/hip trace domains). Output: gpurun_out/prof_$1/...
TAG=${1:-r01c}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o x --output-format csv -- python $R/bench.py > $OUT/bench_under_rocprof.log 2>&1
grep metric $OUT/bench_under_rocprof.log > $OUT/bench_line_under_rocprof.json
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_$C -o x --output-format csv -- python $R/bench.py --steps 3 --warmup 5 --cpu-steps 0 > $OUT/pmc_$C.log 2>&1
done
python $R/bench.py > $OUT/bench_plain.log 2>&1
grep metric $OUT/bench_plain.log > $OUT/bench_line.json
ls -R $OUT | head -30
