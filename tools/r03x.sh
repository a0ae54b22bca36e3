OUT=gpurun_out/r03x; mkdir -p $OUT
for e in "GARMENT_SEED=10" "GARMENT_SEED=11" "GARMENT_SEED=12"; do
( env $e timeout 600 python -m pytest tests/test_gpu_cluster.py -q -s -k "garment" > $OUT/t.log 2>&1 ); echo "== $e"; grep -h "garment, adjoint\|passed\|failed" $OUT/t.log | cut -c1-400
done
