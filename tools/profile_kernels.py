"""``--profile`` mode: rocprofv3 hardware counters per kernel, as one markdown table.

Runs the target command under rocprofv3 in separate counter passes (a pass holds at most 8 SQ, 4 TCC and
2 GRBM counters, and ``--pmc`` is never combined with a trace domain), then joins the per-dispatch CSVs by
kernel name.  The target must be a program rocprofv3 can start directly (``python3 script.py …``): no shell,
no ``env``, no re-exec.

    python tools/profile_kernels.py --out gpurun_out/pmc -- python3 benchmarks/bench_kernels.py --quick
    python tools/profile_kernels.py --out gpurun_out/pmc_bench -- python3 bench.py --steps 40 --warmup 5 --ramp-seconds 4

Output: ``<out>/pass{1,2,3}_counter_collection.csv`` (raw) and ``<out>/summary.md``.
"""
from __future__ import annotations

import argparse
import csv
import glob
import os
import subprocess
import sys
from collections import defaultdict

PASSES = [
    ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT",
     "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "TCC_HIT_sum", "TCC_MISS_sum", "GRBM_GUI_ACTIVE"],
    ["FETCH_SIZE", "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_ANY"],
    ["WRITE_SIZE"],
]


def run_pass(i: int, counters: list[str], out: str, cmd: list[str], timeout: int) -> str:
    d = os.path.join(out, f"pass{i}")
    os.makedirs(d, exist_ok=True)
    full = ["rocprofv3", "--pmc", *counters, "--output-format", "csv", "-d", d, "-o", "p", "--"] + cmd
    env = dict(os.environ, TMPDIR="/tmp")
    print("[profile]", " ".join(full), flush=True)
    subprocess.run(full, check=True, timeout=timeout, env=env, stdout=subprocess.DEVNULL)
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise RuntimeError(f"pass {i}: rocprofv3 wrote no counter_collection.csv under {d}")
    dst = os.path.join(out, f"pass{i}_counter_collection.csv")
    os.replace(files[0], dst)
    return dst


def load(path: str):
    """kernel → counter → mean value per dispatch, plus dispatch counts."""
    sums = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    with open(path, newline="") as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"]
            sums[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
    return {k: {c: v / max(len(disp[k]), 1) for c, v in cs.items()} for k, cs in sums.items()}, {k: len(v) for k, v in disp.items()}


def short(name: str) -> str:
    name = name.split("(")[0].replace("void ", "")
    return name if len(name) <= 70 else name[:67] + "…"


def summarize(paths: list[str], only: str) -> str:
    merged = defaultdict(dict)
    counts = {}
    for p in paths:
        vals, n = load(p)
        for k, cs in vals.items():
            merged[k].update(cs)
            counts.setdefault(k, n[k])
    rows = []
    for k, c in merged.items():
        if only and only not in k:
            continue
        cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0                       # summed over 8 XCDs
        wave_cyc = max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
        hit, miss = c.get("TCC_HIT_sum", 0.0), c.get("TCC_MISS_sum", 0.0)
        rows.append((cyc * counts[k], [
            f"`{short(k)}`", str(counts[k]), f"{c.get('SQ_WAVES', 0):.0f}", f"{cyc:.0f}",
            f"{c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(cyc * 1024, 1):.3f}",     # 1024 SIMDs
            f"{c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c.get('SQ_LDS_IDX_ACTIVE', 0), 1):.3f}" if "SQ_LDS_IDX_ACTIVE" in c else "–",
            f"{c.get('SQ_WAIT_INST_ANY', 0) / wave_cyc:.2f}", f"{c.get('SQ_ACTIVE_INST_ANY', 0) / wave_cyc:.2f}",
            f"{hit / max(hit + miss, 1):.2f}",
            f"{2 * c.get('FETCH_SIZE', 0) / 1e3:.1f}" if "FETCH_SIZE" in c else "–",   # KB; doubled: gfx950 tallies 128-B requests as 64 B
            f"{c.get('WRITE_SIZE', 0) / 1e3:.1f}" if "WRITE_SIZE" in c else "–",
        ]))
    rows.sort(key=lambda r: -r[0])
    head = ["kernel", "dispatches", "waves", "GPU cycles", "MFMA busy / (cycles·1024 SIMD)", "LDS conflict / LDS active",
            "wait frac", "active frac", "L2 hit", "HBM read MB", "HBM write MB"]
    lines = ["| " + " | ".join(head) + " |", "|" + "---|" * len(head)]
    lines += ["| " + " | ".join(r) + " |" for _, r in rows]
    return "\n".join(lines) + "\n"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/pmc")
    ap.add_argument("--only", default="dli::", help="substring filter on kernel names ('' = all kernels)")
    ap.add_argument("--timeout", type=int, default=300, help="seconds per pass")
    ap.add_argument("--summarize-only", action="store_true", help="rebuild summary.md from the CSVs already in --out")
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    cmd = a.cmd[1:] if a.cmd and a.cmd[0] == "--" else a.cmd
    if a.summarize_only:
        paths = sorted(glob.glob(os.path.join(a.out, "pass*_counter_collection.csv")))
        cmd = cmd or ["(see the raw CSVs)"]
    else:
        if not cmd:
            ap.error("give the program to profile after --")
        os.makedirs(a.out, exist_ok=True)
        paths = [run_pass(i + 1, counters, a.out, cmd, a.timeout) for i, counters in enumerate(PASSES)]
    md = ("# rocprofv3 counters per kernel (mean per dispatch)\n\ncommand: `" + " ".join(cmd) + "`\n\npasses: "
          + "; ".join(" ".join(p) for p in PASSES) + "\n\n" + summarize(paths, a.only))
    with open(os.path.join(a.out, "summary.md"), "w") as fh:
        fh.write(md)
    print(md)


if __name__ == "__main__":
    sys.exit(main())
