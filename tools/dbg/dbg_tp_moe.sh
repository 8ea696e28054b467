#!/bin/bash
export DLI_TP_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
P=$(python -c "import numpy as np, json; r=np.random.default_rng(0); print(json.dumps([r.integers(3,1000,size=n).tolist() for n in (40,1200,7)]))")
for mode in 100000 100000 100000 100000 1024 1024 1024; do
DLI_OVERLAP_MIN_ROWS=$mode timeout -k 10 100 python tests/tp_gpu_script.py tiny-mixtral-tp 2 2 gloo "$P" > gpurun_out/dbg.log 2>&1; echo "overlap_min=$mode rc=$? $(grep RESULT gpurun_out/dbg.log | cut -c1-60)"
done
