#!/usr/bin/env python3
"""Per-kernel time inside the TIMED frames of a bench.py run, from a rocprofv3 --kernel-trace csv.

`rocprofv3 --stats` averages over the whole process: pre-roll, warm-up and timed frames together.  The pre-roll of a
pile scene is mostly free fall, so those averages understate what the timed frames cost.  This takes the last
`frames` frames of the trace (a frame starts at k_bounds in contacts mode) and prints, per kernel, launches per frame
and microseconds per substep, plus the idle time between kernels.

usage: timed_region_kernels.py <kernel_trace.csv> [frames=20] [substeps=20]
"""
import collections
import csv
import json
import sys


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").replace("xpbd::", "").split("(")[0]


def main():
    path = sys.argv[1]
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    substeps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [short(r["Kernel_Name"]) for r in rows]
    starts = [i for i, n in enumerate(names) if n == "k_bounds"]
    if len(starts) < frames:
        sys.exit("trace holds %d frames, %d asked" % (len(starts), frames))
    first = starts[-frames]
    last = max(i for i, n in enumerate(names) if n.startswith("k_pair_solve_derive"))
    rows, names = rows[first:last + 1], names[first:last + 1]
    per = collections.OrderedDict()
    for r, n in zip(rows, names):
        e = per.setdefault(n, [0, 0.0])
        e[0] += 1
        e[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
    busy = sum(v[1] for v in per.values())
    out = {"frames": frames, "substeps_per_frame": substeps, "span_us_per_substep": span / frames / substeps,
           "idle_us_per_substep": (span - busy) / frames / substeps, "kernels": {}}
    for n, (calls, us) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        out["kernels"][n] = {"launches_per_frame": calls / frames, "avg_us": us / calls, "us_per_substep": us / frames / substeps}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
