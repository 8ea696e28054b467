"""A/B of work-list decode attention variants selected by environment switches read at library load
(`DLI_DECODE_NT`: non-temporal K/V loads): run once per setting, same box, compare TB/s.

    DLI_DECODE_NT=0 python benchmarks/ab_decode.py; DLI_DECODE_NT=1 python benchmarks/ab_decode.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_kernels as bk  # noqa: E402


def main():
    tag = {k: os.environ.get(k) for k in ("DLI_DECODE_NT",)}
    for rep in range(2):
        for B in (256, 1024, 1400):
            fn, nbytes, splits, mean = bk._decode_mixed_case(B, 770, order="longest_first", worklist={})
            t = bk.timeit(fn)
            print(json.dumps({"B": B, "mean_ctx": round(mean), "us": round(t * 1e6, 1), "TB/s": round(nbytes / t / 1e12, 3), **tag}), flush=True)


if __name__ == "__main__":
    main()
