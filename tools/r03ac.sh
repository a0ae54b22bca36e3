OUT=gpurun_out/r03ac; mkdir -p $OUT
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.log 2> $OUT/bench_default.err ) 2> $OUT/time.txt
tail -1 $OUT/bench_default.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print({k:v for k,v in d.items() if k not in ('config','roofline','cpu_baseline','secondary')})
print('cpu_baseline',d.get('cpu_baseline'))
print('secondary',str(d.get('secondary'))[:300])
r=d['roofline']; print({k:r[k] for k in ('bound','kernel','achieved','peak','frac','traffic','traffic_source','lds_frac','valu_frac')})
print([ (k['kernel'],k['frac'],k.get('traffic_source')) for k in r['kernels']])"
cat $OUT/time.txt | tail -4
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
