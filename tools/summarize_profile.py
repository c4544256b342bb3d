"""Turns a `rocprofv3 --kernel-trace --stats --output-format csv` kernel_stats file into the markdown table
committed under profiles/.   python tools/summarize_profile.py <kernel_stats.csv> <out.md> "<title>" "<command>" ["note"]"""
import csv
import re
import sys


def short(name: str) -> str:
    name = name.replace("mra::(anonymous namespace)::", "").replace("_ZN3mra12_GLOBAL__N_1", "")
    name = re.sub(r"^void ", "", name)
    return name[:84]


def main():
    src, dst, title, cmd = sys.argv[1:5]
    note = sys.argv[5] if len(sys.argv) > 5 else ""
    rows = list(csv.DictReader(open(src)))
    total = sum(int(r["TotalDurationNs"]) for r in rows)
    with open(dst, "w") as f:
        f.write(f"# {title}\n\nCommand (GPU box, from /tmp): `{cmd}`\n\n{note}\n\n")
        f.write(f"Total kernel time {total / 1e6:.2f} ms over {sum(int(r['Calls']) for r in rows)} launches.\n\n")
        f.write("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
        trace = src.replace("kernel_stats", "kernel_trace")
        try:
            split = split_by_grid(trace, "gemm_ws_kernel")
        except OSError:
            split = {}
        for r in rows[:28]:
            f.write(f"| `{short(r['Name'])}` | {r['Calls']} | {int(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | "
                    f"{int(r['MinNs']) / 1e3:.1f} | {int(r['MaxNs']) / 1e3:.1f} | {100 * int(r['TotalDurationNs']) / total:.2f} |\n")
        if split:
            f.write("\n`gemm_ws_kernel` (the K/V projection) by launch shape, from the kernel trace of the same run:\n\n"
                    "| workgroups | launches | avg us | min us | max us |\n|---|---|---|---|---|\n")
            for g, (n, avg, lo, hi) in split.items():
                f.write(f"| {g} | {n} | {avg:.1f} | {lo:.1f} | {hi:.1f} |\n")


def split_by_grid(trace_csv: str, needle: str):
    """Per launch shape (grid size) statistics of one kernel from the kernel trace: the stats file averages the
    video and the audio launches of the same kernel together."""
    groups = {}
    for r in csv.DictReader(open(trace_csv)):
        if needle in r["Kernel_Name"]:
            groups.setdefault(int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return {g: (len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3) for g, v in sorted(groups.items(), reverse=True)}


if __name__ == "__main__":
    main()
