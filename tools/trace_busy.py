"""GPU busy fraction from a rocprofv3 ``*_kernel_trace.csv``: union of kernel intervals / wall span of the
steady-state part of the run (the last ``--tail-frac`` of the dispatches), plus the per-kernel share.

    python tools/trace_busy.py gpurun_out/prof/b_kernel_trace.csv --tail-frac 0.5 > gpurun_out/prof/busy.json
"""
import argparse
import csv
import json
import sys
from collections import defaultdict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--tail-frac", type=float, default=0.5)
    a = ap.parse_args()
    rows = []
    with open(a.trace, newline="") as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    rows = rows[int(len(rows) * (1.0 - a.tail_frac)):]
    t0, t1 = rows[0][0], max(e for _, e, _ in rows)
    busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
    gaps = []
    per = defaultdict(int)
    for s, e, name in rows:
        per[name.split("(")[0][:90]] += e - s
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    gaps.sort()
    tot = sum(per.values())
    out = {"dispatches": len(rows), "wall_ms": (t1 - t0) / 1e6, "busy_ms": busy / 1e6, "busy_frac": busy / (t1 - t0),
           "idle_gaps_over_100us": sum(g > 100_000 for g in gaps),
           "idle_ms_in_gaps_over_100us": sum(g for g in gaps if g > 100_000) / 1e6,
           "top_kernels_share_of_kernel_time": {k: round(v / tot, 4) for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:12]}}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
