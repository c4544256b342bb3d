set -e
mkdir -p gpurun_out/r03d
timeout -k 10 200 tests/native/kernel_check quick > gpurun_out/r03d/kc.log 2>&1; tail -1 gpurun_out/r03d/kc.log
for m in 0 1 3 7 0 1; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-encode --cpu-clips 0 --chain-ring $m > gpurun_out/r03d/bench_ring$m.json 2> gpurun_out/r03d/bench_ring$m.err
  python -c "import json;d=json.load(open('gpurun_out/r03d/bench_ring$m.json'));print('ring',$m,d['ms_per_step'],d['roofline']['avg_launch_ms'])"
done
for m in 0 1 7; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-encode --cpu-clips 0 --workload ref --chain-ring $m > gpurun_out/r03d/bench_ref_ring$m.json 2> gpurun_out/r03d/bench_ref_ring$m.err
  python -c "import json;d=json.load(open('gpurun_out/r03d/bench_ref_ring$m.json'));print('ref ring',$m,d['ms_per_step'])"
done
