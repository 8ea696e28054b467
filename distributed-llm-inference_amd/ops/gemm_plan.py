"""Measured GEMM dispatch: which kernel runs a projection at which row count is decided by timing the
candidates ON THIS GPU at start-up, per (layer shape, row count), instead of by thresholds guessed once.

Why a table and not a rule: the hipGraph-timed sweep in ``profiles/midm_gemm_r3.md`` shows the vendor library 1.3–4.6 × off
the roofline ``max(weight bytes / 5.8 TB/s, FLOPs / 2.0 PFLOP/s)`` for 64 < M ≤ 1024 — the regime of decode-heavy TP = 8
steps and of lightly loaded replicas — with bad spots that move with M (Llama-3-8B qkv: 103 µs at M = 2304, 214 µs at
2816, 165 µs at 3328), while the hand-written 256×256×64 tile (``csrc/kernels/gemm256.hip``) with K cut into slices
(fp32 partial slabs + reduce pass, so that one or two row tiles still put a workgroup on every CU) wins some shapes by
1.2–2.2 × and loses others.  Row counts are few: every captured step is padded to a bucket (decode batch sizes, mixed
token counts in multiples of 256), so the table is keyed by the exact M the graphs will see; eager steps take the entry of
the next bucket up.

Candidates per (N, K, swiglu, M):  ``lib`` (hipBLASLt through torch, plus ``silu_and_mul`` for gate/up), ``tile:S`` —
the 256² tile with S K-slices (S = 1: no slabs; with the SwiGLU epilogue fused for gate/up) — and ``tile+tail``: the tile
with only the REMAINDER of its last, partly filled round of one workgroup per CU cut into K-slices (a 7 × 112-tile
gate/up GEMM is 3 rounds + 16 tiles: those 16 run as 128 slices next to each other instead of as a fourth round).  Weights rotate through all
layers of the model while timing, so no candidate finds its matrix in the 256 MB infinity cache.  Rank 0's table is
broadcast under TP: every rank of a group runs the same kernels.
"""
from __future__ import annotations

import os
import time
from typing import Optional

import torch

ENABLED = os.environ.get("DLI_GEMM_TUNE", "1") != "0"
CAPTURE_MODE = os.environ.get("DLI_CAPTURE_MODE", "thread_local")     # see ops.CAPTURE_MODE
MIN_ROWS = 65                 # at and below 64 rows the weight-streaming kernel (skinny_gemm.hip) has its own rules
TAIL = 100                    # table value: plain tiles for the whole rounds, K-split remainder (dense_gemm256 split_tail)
NUM_CU = 256


class GemmPlan:
    def __init__(self):
        self.table: dict = {}          # (N, K, swiglu) -> sorted [(M, splits)]  (splits 0 = library)
        self.report: dict = {}

    # ------------------------------------------------------------------ lookup (hot path: two dict probes)
    def splits_for(self, m: int, n: int, k: int, swiglu: bool) -> Optional[int]:
        """K-slices of the 256² tile for this call, 0 for the library path, None when the shape was never tuned."""
        rows = self.table.get((n, k, swiglu))
        if rows is None or m < MIN_ROWS:
            return None
        hit = rows.get(m)
        if hit is not None:
            return hit
        ms = rows["_sorted"]
        for mm in ms:                                       # the next bucket up (same tile count or one more)
            if mm >= m:
                return rows[mm]
        return rows[ms[-1]]

    # ------------------------------------------------------------------ tuning
    @staticmethod
    def tile_ok(n: int, k: int, swiglu: bool) -> bool:
        n_out = n // 2 if swiglu else n
        return n % 2 == 0 and k % 64 == 0 and n_out % (128 if swiglu else 256) == 0 and n * k * 2 < 2 ** 32

    @staticmethod
    def tail_slices(m: int, n: int, k: int, swiglu: bool) -> int:
        """K-slices the remainder tiles would be cut into (0: nothing to cut) — ``dense_gemm256_tail_plan`` in gemm256.hip."""
        n_out = n // 2 if swiglu else n
        tiles = (m + 255) // 256 * (n_out // (128 if swiglu else 256))
        r = tiles % NUM_CU
        if r == 0 or tiles < NUM_CU or r > NUM_CU // 2:
            return 0
        s = min(8, NUM_CU // r, k // 64 // 4)
        return s if s >= 2 else 0

    @staticmethod
    def candidates(m: int, n: int, k: int, swiglu: bool) -> list:
        n_out = n // 2 if swiglu else n
        tiles = (m + 255) // 256 * (n_out // (128 if swiglu else 256))
        out = [0, 1]
        s = 2
        while s <= 16 and s <= k // 256 and tiles * s <= 768:
            out.append(s)
            s *= 2
        if GemmPlan.tail_slices(m, n, k, swiglu):
            out.append(TAIL)
        return out

    @torch.inference_mode()
    def tune(self, groups: list, row_counts: list, device, budget_s: float = 60.0) -> dict:
        """``groups``: [(name, [weight tensors of identical shape, one per layer], swiglu)].  Fills ``self.table``."""
        from . import hip, silu_and_mul
        import torch.nn.functional as F
        lib = hip()
        t_begin = time.perf_counter()
        rows_all = sorted({int(m) for m in row_counts if m >= MIN_ROWS})
        for name, weights, swiglu in groups:
            w0 = weights[0]
            n, k = int(w0.shape[0]), int(w0.shape[1])
            if not (w0.is_cuda and w0.dtype == torch.bfloat16 and self.tile_ok(n, k, swiglu)) or (n, k, swiglu) in self.table:
                continue
            entry, rep = {}, {}
            for m in rows_all:
                if time.perf_counter() - t_begin > budget_s:
                    break
                if m * k * 2 >= 2 ** 32:
                    continue
                x = torch.randn(m, k, device=device, dtype=torch.float32).to(torch.bfloat16)
                times = {}
                for s in self.candidates(m, n, k, swiglu):
                    if s == 0:
                        fn = (lambda w: silu_and_mul(F.linear(x, w))) if swiglu else (lambda w: F.linear(x, w))
                    elif s == TAIL:
                        fn = lambda w: lib.dense_gemm256(x, w, swiglu, 1, None, True)
                    else:
                        fn = lambda w, s=s: lib.dense_gemm256(x, w, swiglu, s)
                    times[s] = self._time(fn, weights)
                best = min(times, key=times.get)
                if best != 0 and times[best] > 0.97 * times[0]:
                    best = 0                                 # the library keeps ties: fewer moving parts
                entry[m] = best
                rep[m] = {"lib_us": round(times[0], 1), "best_us": round(times[best], 1),
                          "best": "lib" if best == 0 else "tile+tail" if best == TAIL else f"tile:{best}"}
                if TAIL in times:
                    rep[m]["tail_us"], rep[m]["tile1_us"] = round(times[TAIL], 1), round(times[1], 1)
            if entry:
                entry["_sorted"] = sorted(entry)
                self.table[(n, k, swiglu)] = entry
                self.report[f"{name} N={n} K={k}" + (" swiglu" if swiglu else "")] = rep
        self.report["_seconds"] = round(time.perf_counter() - t_begin, 1)
        return self.report

    @staticmethod
    def _time(fn, weights, reps: int = 2) -> float:
        """µs per call inside a hipGraph that walks all the layers' weights ``reps`` times."""
        n = len(weights) * reps if len(weights) >= 8 else max(8, len(weights) * reps)
        fn(weights[0])
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.graph(g, stream=s, capture_error_mode=CAPTURE_MODE):
            for i in range(n):
                fn(weights[i % len(weights)])
        torch.cuda.synchronize()
        g.replay()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) * 1e3 / n

    # ------------------------------------------------------------------ (de)serialisation: TP ranks share rank 0's table
    def export(self) -> dict:
        return {f"{n},{k},{int(sw)}": {str(m): s for m, s in rows.items() if m != "_sorted"} for (n, k, sw), rows in self.table.items()}

    def load(self, blob: dict) -> None:
        self.table = {}
        for key, rows in blob.items():
            n, k, sw = key.split(",")
            entry = {int(m): int(s) for m, s in rows.items()}
            entry["_sorted"] = sorted(entry)
            self.table[(int(n), int(k), bool(int(sw)))] = entry

    def summary(self) -> dict:
        """How many (shape, M) entries left the library, and the largest measured gain."""
        moved = total = 0
        best = 1.0
        for name, rep in self.report.items():
            if name.startswith("_"):
                continue
            for m, r in rep.items():
                total += 1
                if r["best"] != "lib":
                    moved += 1
                    best = max(best, r["lib_us"] / max(r["best_us"], 1e-9))
        return {"entries": total, "own_tile_entries": moved, "largest_gain_vs_library": round(best, 2),
                "seconds": self.report.get("_seconds")}


PLAN = GemmPlan()
