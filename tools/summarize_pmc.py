#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counters (csv).  usage: summarize_pmc.py DIR [DIR ...]"""
import csv, glob, os, re, sys
from collections import defaultdict

def short(name):
    mm = re.search(r"\d+([a-z0-9_]+_kernel)(?:I((?:Li\d+E)+)E)?", name) if name.startswith("_ZN2bd") else None
    if mm:   # mangled (the demangler does not know _Float16): rebuild name<args>
        args = re.findall(r"Li(\d+)E", mm.group(2) or "")
        return mm.group(1) + ("<" + ", ".join(args) + ">" if args else "")
    m = re.search(r"(\w+_kernel)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:50]

acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "bd::" not in r["Kernel_Name"] and "_ZN2bd" not in r["Kernel_Name"]:
                continue
            key = (short(r["Kernel_Name"]), r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", ""))
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for v in acc.values() for c in v})
w = csv.writer(sys.stdout)
w.writerow(["kernel", "grid", "n"] + names)
for key, v in sorted(acc.items()):
    n = max(len(x) for x in v.values())
    w.writerow([key[0], key[1], n] + [f"{sum(v[c]) / len(v[c]):.4g}" if c in v else "" for c in names])
