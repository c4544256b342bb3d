"""Prints the step time, roofline fraction and the encode-stage entries of the last bench.py JSON line in a log file."""
import json
import sys

line = [x for x in open(sys.argv[1]) if x.startswith("{")][-1]
d = json.loads(line)
print("ms_per_step", d["ms_per_step"], "roofline.frac", d["roofline"]["frac"], "fold block ms", d["roofline"].get("avg_launch_ms"))
for k, v in d.get("encode_stage", {}).items():
    print(k, v.get("ms"), v.get("tflops"), v.get("spot_check_max_abs_diff_4_frames"), v.get("error"))
