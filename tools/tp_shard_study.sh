#!/bin/bash
# One TP = 8 rank's workload on the one GPU: bench.py --emulate-tp 8 with and without the mixed-step
# hipGraphs, then a kernel trace of the graph run → GPU-busy fraction and kernel mix (tools/trace_busy.py).
#   bash tools/tp_shard_study.sh llama3-70b r3e            (on the GPU box; results under gpurun_out/)
set -u
MODEL=${1:-llama3-70b}; TAG=${2:-r3}; EXTRA=${3:-}
OUT=gpurun_out; mkdir -p $OUT
COMMON="--emulate-tp 8 --model $MODEL --steps 200 --warmup 10 --ramp-seconds 20 --ramp-cap-seconds 40 --sustained-seconds 20 $EXTRA"
timeout -k 10 700 python bench.py $COMMON > $OUT/emu_${MODEL}_${TAG}_graphs.json 2> $OUT/emu_${MODEL}_${TAG}_graphs.err || { tail -5 $OUT/emu_${MODEL}_${TAG}_graphs.err; exit 1; }
QPS=$(python -c "import json,sys; print(json.load(open('$OUT/emu_${MODEL}_${TAG}_graphs.json'))['config']['offered_qps'])")
echo "offered load from the capacity probe: $QPS req/s"
timeout -k 10 700 python bench.py $COMMON --qps $QPS --no-mixed-graphs > $OUT/emu_${MODEL}_${TAG}_eager.json 2> $OUT/emu_${MODEL}_${TAG}_eager.err || { tail -5 $OUT/emu_${MODEL}_${TAG}_eager.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/emu_prof_${MODEL}_${TAG} -o t -- python3 $GRAFT_REPO_ROOT/bench.py --emulate-tp 8 --model $MODEL --steps 150 --warmup 10 --ramp-seconds 15 --ramp-cap-seconds 15 --sustained-seconds 5 --qps $QPS $EXTRA > $GRAFT_REPO_ROOT/$OUT/emu_${MODEL}_${TAG}_traced.json 2> $GRAFT_REPO_ROOT/$OUT/emu_${MODEL}_${TAG}_traced.err || { tail -5 $GRAFT_REPO_ROOT/$OUT/emu_${MODEL}_${TAG}_traced.err; exit 1; }
cd $GRAFT_REPO_ROOT
T=$(find $OUT/emu_prof_${MODEL}_${TAG} -name "*kernel_trace.csv" | head -1)
python tools/trace_busy.py $T --tail-frac 0.3 > $OUT/emu_${MODEL}_${TAG}_gpu_busy.json
S=$(find $OUT/emu_prof_${MODEL}_${TAG} -name "*kernel_stats.csv" | head -1)
head -25 $S > $OUT/emu_${MODEL}_${TAG}_kernel_stats_top.csv
rm -rf $OUT/emu_prof_${MODEL}_${TAG}
python - <<PY
import json
for k in ("graphs", "eager"):
    d = json.load(open("$OUT/emu_${MODEL}_${TAG}_%s.json" % k))
    r = d["rank0"]
    print(k, {x: d[x] for x in ("value", "ms_per_step", "host_ms_per_step", "total_tok_per_s", "steady", "sustained_tpot_p50_s")},
          {x: r[x] for x in ("graph_steps", "mixed_graph_steps", "eager_steps", "running", "step_tokens")})
b = json.load(open("$OUT/emu_${MODEL}_${TAG}_gpu_busy.json"))
print("busy", b["busy_frac"], b["wall_ms"], b["top_kernels_share_of_kernel_time"])
PY
