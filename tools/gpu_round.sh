#!/bin/bash
# Developer script: a sequence of independent GPU checks, each under its own timeout, logs under gpurun_out/$1/
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
run() { name=$1; shift; echo "=== $name: $*"; ( timeout 420 "$@" > $OUT/$name.log 2>&1; echo "rc=$?" >> $OUT/$name.log ); tail -${TAILN:-6} $OUT/$name.log | cut -c1-400; }
for item in "$@"; do
  case $item in
    selfc) run selfc python -m pytest tests/test_gpu_selfcontact.py -x -q ;;
    debug)
      i=0
      while read -r cl cfg; do
        i=$((i+1)); export DC_CLUSTER=$cl; TAILN=7 run debug$i python tools/debug_fold.py $cfg; unset DC_CLUSTER
      done <<'CFG'
4 2 40 1 6 1
4 2 40 0 6 1
4 2 40 1 0 0
2 2 40 1 6 1
8 2 40 1 6 1
4 8 40 1 6 1
CFG
      ;;
    cluster) TAILN=30 run cluster python -m pytest tests/test_gpu_cluster.py -q -s ;;
    garment) TAILN=30 run garment python -m pytest tests/test_gpu_cluster.py -q -s -k garment ;;
    pym) TAILN=30 run pym python -m pytest tests/test_gpu_pymodule.py -q -x ;;
    pymk1) export DC_CLUSTER=1; TAILN=12 run pymk1 python -m pytest tests/test_gpu_pymodule.py -q -x -k "dress_twirl" -s; unset DC_CLUSTER ;;
    pymdress) TAILN=12 run pymdress python -m pytest tests/test_gpu_pymodule.py -q -x -k "dress_twirl" -s ;;
    big) TAILN=8 run big python -m pytest tests/test_gpu_fullsize.py -q -s -k beyond ;;
    bigk1) export DC_CLUSTER=1; TAILN=8 run bigk1 python -m pytest tests/test_gpu_fullsize.py -q -s -k beyond; unset DC_CLUSTER ;;
    bigk4) export DC_CLUSTER=4; TAILN=8 run bigk4 python -m pytest tests/test_gpu_fullsize.py -q -s -k beyond; unset DC_CLUSTER ;;
    refcall) TAILN=12 run refcall python -m pytest tests/test_gpu_reference_callers.py tests/test_gpu_rccl.py -q -s ;;
    hat) TAILN=14 run hat python -m pytest tests/test_gpu_configs.py -q -s -k hat ;;
    benchj) export DC_BLOCK_PRE=0; TAILN=3 run benchj python bench.py --steps 20 --warmup 5 --cpu-steps 0 --tshirt 0; unset DC_BLOCK_PRE ;;
    benchq) TAILN=3 run benchq python bench.py --steps 20 --warmup 5 --cpu-steps 0 --tshirt 0 ;;
    cfgs) TAILN=20 run cfgs python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_random_scenes.py -q ;;
    b32nopipe) export DC_PIPECG=0; TAILN=2 run b32nopipe python bench.py --steps 20 --warmup 5 --total-batch 32 --cpu-steps 0 --tshirt 0; unset DC_PIPECG ;;
    fallb) TAILN=30 run fallb python -m pytest tests/test_gpu_fallbacks.py -q ;;
    parity) TAILN=40 run parity python -m pytest tests/test_gpu_bench_parity.py -q -s ;;
    all) TAILN=15 run all python -m pytest tests -m gpu -q -x ;;
    final) bash tools/round_final.sh $TAG ;;      # the round's CLOSING record: whole suite + ledger + bench on the library as it is, its sha256 recorded (tools/check_round_final.py refuses a round whose record is not of the library in the tree)
    rest) TAILN=15 run rest python -m pytest tests -m gpu -q --deselect tests/test_gpu_bench_parity.py --deselect tests/test_gpu_cluster.py ;;
    bench) TAILN=3 run bench python bench.py --steps 20 --warmup 5 ;;
    bench32) TAILN=3 run bench32 python bench.py --steps 20 --warmup 5 --total-batch 32 --cpu-steps 0 --tshirt 0 ;;
    b32f0) TAILN=2 run b32f0 python bench.py --steps 20 --warmup 5 --total-batch 32 --cpu-steps 0 --fold-rows 0 ;;
    b32f0k1) TAILN=2 run b32f0k1 python bench.py --steps 20 --warmup 5 --total-batch 32 --cpu-steps 0 --fold-rows 0 --cluster 1 ;;
    b32k4) TAILN=2 run b32k4 python bench.py --steps 20 --warmup 5 --total-batch 32 --cpu-steps 0 --cluster 4 ;;
    b32k2) TAILN=2 run b32k2 python bench.py --steps 20 --warmup 5 --total-batch 32 --cpu-steps 0 --cluster 2 ;;
    b64) TAILN=2 run b64 python bench.py --steps 20 --warmup 5 --total-batch 64 --cpu-steps 0 --tshirt 0 ;;
    b128) TAILN=2 run b128 python bench.py --steps 20 --warmup 5 --total-batch 128 --cpu-steps 0 --tshirt 0 ;;
    g128) TAILN=2 run g128 python bench.py --steps 8 --warmup 4 --grid 128 --fold-rows 0 --cpu-steps 0 ;;
    g128k1) TAILN=2 run g128k1 python bench.py --steps 8 --warmup 4 --grid 128 --fold-rows 0 --cpu-steps 0 --cluster 1 ;;
    bench32k1) TAILN=3 run bench32k1 python bench.py --steps 20 --warmup 5 --total-batch 32 --cpu-steps 0 --cluster 1 ;;
    benchq0) export DC_CG_SEED=0; TAILN=3 run benchq0 python bench.py --steps 20 --warmup 5 --cpu-steps 0 --tshirt 0; unset DC_CG_SEED ;;
    bench32s0) export DC_CG_SEED=0; TAILN=3 run bench32s0 python bench.py --steps 20 --warmup 5 --total-batch 32 --cpu-steps 0 --tshirt 0; unset DC_CG_SEED ;;
    paritycore) TAILN=12 run paritycore python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py -q -x ;;
    *) echo "unknown item $item" ;;
  esac
done
