#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats (csv) output directory into a small markdown table
for profiles/: this repository's kernels only, pointwise launches split by GEMM shape (grid size).

    python tools/summarize_rocprof.py gpurun_out/prof_r01 profiles/r01_kernel_trace_summary.md "<command line>"
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    mm = re.search(r"\d+([a-z0-9_]+_kernel)(?:I((?:Li\d+E)+)E)?", name) if name.startswith("_ZN2bd") else None
    if mm:   # mangled (the demangler does not know _Float16): rebuild name<args>
        args = re.findall(r"Li(\d+)E", mm.group(2) or "")
        return mm.group(1) + ("<" + ", ".join(args) + ">" if args else "")
    m = re.search(r"(\w+_kernel)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def main():
    src, dst = sys.argv[1], sys.argv[2]
    cmd = sys.argv[3] if len(sys.argv) > 3 else ""
    trace = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
    if not trace:
        sys.exit("no kernel_trace.csv under " + src)
    rows = defaultdict(list)
    total = 0.0
    with open(trace[0]) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            total += dur
            if "bd::" not in name and "_ZN2bd" not in name:
                continue
            key = (short(name), int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r.get("Grid_Size_Y", 1)),
                   r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"])
            rows[key].append(dur)
    ours = sum(sum(v) for v in rows.values())
    with open(dst, "w") as out:
        out.write(f"# rocprofv3 --kernel-trace --stats summary\n\ncommand: `{cmd}`\n\n")
        out.write(f"source: `{os.path.relpath(trace[0])}` (GPU box scratch; this file is the committed summary)\n\n")
        out.write(f"all kernels: {total / 1e3:.3f} ms; this repo's kernels: {ours / 1e3:.3f} ms "
                  f"({100 * ours / total:.1f} %; the rest is torch generating the synthetic audio)\n\n")
        out.write("| kernel | workgroups (x, y) | VGPR | AGPR | LDS B | calls | avg us | min us | max us | total ms | % of ours |\n")
        out.write("|---|---|---|---|---|---|---|---|---|---|---|\n")
        for key, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
            out.write(f"| {key[0]} | {key[1]}, {key[2]} | {key[3]} | {key[4]} | {key[5]} | {len(v)} | "
                      f"{sum(v) / len(v):.1f} | {min(v):.1f} | {max(v):.1f} | {sum(v) / 1e3:.3f} | {100 * sum(v) / ours:.1f} |\n")
        by = defaultdict(list)
        for key, v in rows.items():
            by[key[0].split("<")[0]] += v
        out.write("\n| kernel family | calls | avg us | total ms |\n|---|---|---|---|\n")
        for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
            out.write(f"| {k} | {len(v)} | {sum(v) / len(v):.2f} | {sum(v) / 1e3:.3f} |\n")
    print(open(dst).read())


if __name__ == "__main__":
    main()
