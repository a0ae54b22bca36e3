OUT=gpurun_out/r03e; mkdir -p $OUT
run() { name=$1; shift; ( timeout ${TMO:-400} "$@" > $OUT/$name.log 2>&1; echo "rc=$?" >> $OUT/$name.log ); echo "=== $name"; tail -${TAILN:-8} $OUT/$name.log | cut -c1-400; }
TAILN=2 run bench python bench.py --steps 20 --warmup 5 --tshirt 0 --cpu-steps 0
TAILN=2 run bench32 python bench.py --steps 20 --warmup 5 --tshirt 0 --cpu-steps 0 --total-batch 32
TMO=1800 TAILN=60 run all python -m pytest tests -m gpu -q -s
grep -h "^\[config\]\|^\[bench parity\] worst\|^\[dress 7742\]\|^\[param grads\|^\[garment\|gradient rel err" $OUT/all.log | cut -c1-420 > $OUT/summary.txt
