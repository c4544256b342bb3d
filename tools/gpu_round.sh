#!/bin/bash
# One GPU-box session: native kernel self-check, parity tests, smoke, bench, rocprof kernel trace.
# Usage (from the repo root on the GPU box):  bash tools/gpu_round.sh [tag]
set -o pipefail
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out
[ -z "$GRAFT_REPO_ROOT" ] && OUT=$(pwd)/gpurun_out
mkdir -p $OUT
cd $(dirname $0)/..
timeout -k 10 200 ./tests/native/kernel_check quick > $OUT/native_$TAG.log 2>&1 || { echo "native check failed"; tail -20 $OUT/native_$TAG.log; exit 1; }
echo "native ok: $(tail -1 $OUT/native_$TAG.log)"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_$TAG.log 2>&1; RC=$?
tail -25 $OUT/pytest_gpu_$TAG.log
[ $RC -ne 0 ] && exit $RC
timeout -k 10 300 python __graft_entry__.py smoke > $OUT/smoke_$TAG.log 2>&1 || { echo "smoke failed"; tail -20 $OUT/smoke_$TAG.log; exit 1; }
tail -4 $OUT/smoke_$TAG.log
timeout -k 10 600 python bench.py --steps 5 --warmup 2 > $OUT/bench_$TAG.log 2>&1 || { echo "bench failed"; tail -20 $OUT/bench_$TAG.log; exit 1; }
tail -2 $OUT/bench_$TAG.log
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --workload ref --cpu-clips 32 > $OUT/bench_ref_$TAG.log 2>&1 || { echo "bench ref failed"; tail -20 $OUT/bench_ref_$TAG.log; exit 1; }
tail -1 $OUT/bench_ref_$TAG.log
# rocprof kernel trace of the bench command (summary copied to profiles/ by hand)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o bench -- python3 /root/repo/bench.py --steps 5 --warmup 2 --cpu-clips 0 > $OUT/prof_$TAG.log 2>&1 || { echo "rocprof failed"; tail -20 $OUT/prof_$TAG.log; exit 1; }
tail -1 $OUT/prof_$TAG.log
find $OUT/prof_$TAG -name "*stats*" | head
