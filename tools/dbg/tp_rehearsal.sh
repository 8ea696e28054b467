#!/bin/bash
# Real-shape TP/EP rehearsal on ONE GPU: all ranks share device 0 and talk over gloo; the custom all-reduce runs over
# HIP IPC between the processes.  Prints the greedy tokens of TP=1 and TP=2 for comparison.
#   bash tools/dbg/tp_rehearsal.sh "llama3-8b 1 1" "llama3-8b 2 1"
set -u
export DLI_TP_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
P='[[11,45,902,77,431,8,8,19,250,3,999,512,64,31,7],[5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,36,37,38,39,40]]'
( while sleep 45; do echo "[tick $(date +%T)]"; done ) &
TICK=$!
trap "kill $TICK" EXIT
for cfg in "$@"; do
  set -- $cfg
  out=gpurun_out/tp_rehearsal_$1_tp$2_ep$3.log
  export DLI_DUMP_LOGITS=gpurun_out/tp_rehearsal_$1_tp$2_ep$3.logits.pt
  timeout -k 10 ${TP_TIMEOUT:-300} python tests/tp_gpu_script.py $1 $2 $3 gloo "$P" > $out 2>&1
  rc=$?
  echo "$cfg rc=$rc $(grep '^RESULT' $out | cut -c1-400)"
  [ $rc -ne 0 ] && { tail -5 $out; exit 1; }
done
python - <<'PY'
import glob, re, torch
by = {}
for f in sorted(glob.glob("gpurun_out/tp_rehearsal_*.logits.pt")):
    m = re.match(r".*tp_rehearsal_(.+)_tp(\d+)_ep(\d+)\.logits\.pt", f)
    by.setdefault(m.group(1), {})[(int(m.group(2)), int(m.group(3)))] = torch.load(f, weights_only=True)
for model, d in by.items():
    if (1, 1) not in d:
        continue
    ref = d[(1, 1)]
    for k, v in d.items():
        if k != (1, 1):
            cos = torch.nn.functional.cosine_similarity(ref, v, dim=0).item()
            print(f"LOGITS {model} tp{k[0]} ep{k[1]} vs tp1: cosine {cos:.5f}  max|d| {(ref - v).abs().max():.4f}  std {ref.std():.4f}  argmax {int(ref.argmax())} / {int(v.argmax())}")
PY
exit 0
