OUT=gpurun_out/r06_05; mkdir -p $OUT
for cg in 1 0; do DC_ADJ_CG=$cg timeout 200 python tools/r06_ab/r06_cgdist.py > $OUT/cgdist_$cg.log 2>&1; echo "DC_ADJ_CG=$cg"; tail -5 $OUT/cgdist_$cg.log; done
