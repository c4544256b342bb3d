# in-step A/B of the layer-chain options (bench.py --chain-ring MASK), one session
set -e
out=${1:-gpurun_out/ab_chain}
mkdir -p $out
for m in 7 15 7 15; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-encode --cpu-clips 0 --chain-ring $m > $out/bench_ring$m.json 2> $out/bench_ring$m.err
  python -c "import json;d=json.load(open('$out/bench_ring$m.json'));print('ring',$m,d['ms_per_step'],d['roofline']['avg_launch_ms'])"
done
for m in 7 15 7 15; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-encode --cpu-clips 0 --workload ref --chain-ring $m > $out/bench_ref_ring$m.json 2> $out/bench_ref_ring$m.err
  python -c "import json;d=json.load(open('$out/bench_ref_ring$m.json'));print('ref ring',$m,d['ms_per_step'])"
done
