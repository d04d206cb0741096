#!/usr/bin/env python3
"""HBM bytes per launch per kernel family from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

    python tools/pmc_traffic.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass> <out.json> "<command>"

bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB, and on gfx950 FETCH_SIZE counts a wide
coalesced read as half its size (MI355X_MICROARCH.md, HBM / rocprofv3 section).  Families are kernel names without
template arguments, as bench.py uses them; the per-launch figure is the average over the family's launches.
"""
import csv, glob, json, os, re, sys
from collections import defaultdict


def family(name):
    m = re.search(r"\d+([a-z0-9_]+_kernel)", name) if name.startswith("_ZN2bd") else re.search(r"(\w+_kernel)", name)
    if not m:
        return None
    fam = m.group(1)
    return fam


def collect(d, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            fam = family(r["Kernel_Name"])
            if fam and ("bd::" in r["Kernel_Name"] or "_ZN2bd" in r["Kernel_Name"]):
                tot[fam] += float(r["Counter_Value"])
                cnt[fam] += 1
    return tot, cnt


def main():
    fdir, wdir, out, cmd = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else ""
    ft, fc = collect(fdir, "FETCH_SIZE")
    wt, wc = collect(wdir, "WRITE_SIZE")
    fams = {}
    for fam in sorted(ft):
        n = fc[fam]
        fams[fam] = {"launches_counted": n,
                     "hbm_bytes_per_launch": int((2 * ft[fam] / n + wt.get(fam, 0.0) / max(wc.get(fam, 1), 1)) * 1024),
                     "fetch_kib_per_launch": round(ft[fam] / n, 1),
                     "write_kib_per_launch": round(wt.get(fam, 0.0) / max(wc.get(fam, 1), 1), 1)}
    json.dump({"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `{cmd}`; "
                         "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts half of wide coalesced "
                         "reads, MI355X_MICROARCH.md HBM section); per-launch average over the family's launches",
               "per_kernel_family": fams}, open(out, "w"), indent=1)
    print(json.dumps(fams, indent=1))


if __name__ == "__main__":
    main()
