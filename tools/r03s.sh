OUT=gpurun_out/r03s; mkdir -p $OUT
L=$PWD/diffcloth_amd/lib
for v in ${VARS:-v1 v2 v3 v4 v5 v6}; do
( DC_LIB=$L/libdiffcloth_hip_$v.so timeout 300 python bench.py --steps 4 --warmup 2 --tshirt 0 --cpu-steps 0 > $OUT/$v.log 2>&1 )
echo "== $v"; grep -h "phases pk" $OUT/$v.log | tail -1 | cut -c1-400
grep '"metric"' $OUT/$v.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('value',round(d['value'],1),[ (k['kernel'],round(k['ms_per_step'],2)) for k in d['roofline']['kernels']])"
done
