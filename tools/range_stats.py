#!/usr/bin/env python3
"""Kernel time per roctx range: joins rocprofv3's marker trace with its kernel trace.

    rocprofv3 --marker-trace --kernel-trace --output-format csv -d OUT -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline
    python3 tools/range_stats.py OUT > profiles/r05_bench_with_extras_by_range.csv

Every C-ABI entry point of the library is a range named after itself (rc_capi.hip, guarded()); bench.py wraps its phases in ranges of its own
("headline:...", "extra:<name>", "workload:<label>:<rays>x<launches>").  A dispatch carries the correlation id of the innermost range that
was open on its thread when it was enqueued; it is reported under the OUTERMOST-BUT-ONE label that encloses that range in time -- the
workload when there is one, else the extra, else the entry point itself -- so that the small kernels in front of a launch (k_order_*) are
charged to the workload that launched them.  Durations are the dispatches' own start-to-end times on the device."""
import collections
import csv
import glob
import os
import sys


def main():
    root = sys.argv[1]
    markers, kernels = [], []
    for f in glob.glob(os.path.join(root, "**", "*_marker_api_trace.csv"), recursive=True):
        markers += list(csv.DictReader(open(f)))
    for f in glob.glob(os.path.join(root, "**", "*_kernel_trace.csv"), recursive=True):
        kernels += list(csv.DictReader(open(f)))
    rng = [(int(m["Start_Timestamp"]), int(m["End_Timestamp"]), m["Function"], int(m["Correlation_Id"]), m["Thread_Id"]) for m in markers]
    by_corr = {r[3]: r for r in rng}
    rng.sort()

    def label(r):
        """the most specific of workload: / extra: / headline: among the ranges that enclose r on its thread, else r's own name"""
        best, rank = r[2], 0
        for s, e, name, _, tid in rng:
            if s > r[0]:
                break
            if tid == r[4] and e >= r[1]:
                k = 3 if name.startswith("workload:") else (2 if name.startswith(("extra:", "headline:")) else 0)
                if k > rank:
                    best, rank = name, k
        return best
    cache = {}
    agg = collections.defaultdict(list)
    for k in kernels:
        r = by_corr.get(int(k["Correlation_Id"]))
        if r is None:
            lab = "(no range)"
        else:
            if r not in cache:
                cache[r] = label(r)
            lab = cache[r]
        name = k["Kernel_Name"]
        name = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
        agg[(lab, name)].append(int(k["End_Timestamp"]) - int(k["Start_Timestamp"]))
    w = csv.writer(sys.stdout)
    w.writerow(["range", "kernel", "dispatches", "total_us", "avg_us", "min_us", "max_us"])
    order = sorted(agg.items(), key=lambda kv: (kv[0][0], -sum(kv[1])))
    for (lab, name), d in order:
        w.writerow([lab, name[:110], len(d), round(sum(d) / 1e3, 1), round(sum(d) / len(d) / 1e3, 2), round(min(d) / 1e3, 2), round(max(d) / 1e3, 2)])


if __name__ == "__main__":
    main()
