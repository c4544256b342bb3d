# Round-end measurements on the GPU box, in three calls (a gpurun call is limited to 20 minutes):
#   bash tools/run_round_profiles.sh r03z tests   -- pytest -m gpu, smoke, native kernel_check
#   bash tools/run_round_profiles.sh r03z lines   -- bench lines (headline, reference item shape, finetune, ViT)
#   bash tools/run_round_profiles.sh r03z prof    -- rocprofv3 kernel traces of the three workloads + the PMC passes of the folded block
set -o pipefail
O=gpurun_out/${1:-r03z}
PH=${2:-all}
mkdir -p $O
if [ "$PH" = tests ] || [ "$PH" = all ]; then
  python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; tail -3 $O/gputest.log; python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
  (cd tests/native && timeout -k 10 500 ./kernel_check quick > ../../$O/kernel_check.log 2>&1; tail -2 ../../$O/kernel_check.log)
fi
if [ "$PH" = lines ] || [ "$PH" = all ]; then
  python bench.py > $O/bench_line.json 2> $O/bench.err; tail -1 $O/bench.err
  python bench.py --workload ref --no-encode > $O/bench_ref_line.json 2>> $O/bench.err
  python tools/bench_finetune.py --steps 20 > $O/finetune_line.json 2>/dev/null
  python tools/bench_vit.py --frames 1024 --reps 2 > $O/vit_line.json 2>/dev/null
  python tools/bench_vit.py --frames 1024 --reps 2 --ln-fold 0 >> $O/vit_line.json 2>/dev/null
  python tools/bench_vit.py --frames 1024 --reps 2 --residual op >> $O/vit_line.json 2>/dev/null
  python tools/bench_vit.py --frames 1024 --reps 2 --backend torch >> $O/vit_line.json 2>/dev/null
  cat $O/vit_line.json $O/finetune_line.json | cut -c1-300
fi
if [ "$PH" = prof ] || [ "$PH" = all ]; then
  cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
  rocprofv3 --kernel-trace --stats -d $O/prof_bench -o bench --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-encode --cpu-clips 0 > $O/prof_bench.log 2>&1
  rocprofv3 --kernel-trace --stats -d $O/prof_ft -o ft --output-format csv -- python3 tools/bench_finetune.py --steps 5 > $O/prof_ft.log 2>&1
  rocprofv3 --kernel-trace --stats -d $O/prof_vit -o vit --output-format csv -- python3 tools/bench_vit.py --frames 256 --reps 2 > $O/prof_vit.log 2>&1
  bash tools/pmc_fold.sh ${1:-r03} > $O/pmc.log 2>&1; tail -1 $O/pmc.log | cut -c1-300
fi
ls $O
