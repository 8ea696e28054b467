#!/bin/bash
# rocprofv3 kernel trace of a shortened headline bench → per-kernel stats + GPU-busy fraction (tools/trace_busy.py).
#   bash tools/profile_bench.sh r3u [extra bench args]      (on the GPU box; results under gpurun_out/)
set -u
TAG=${1:-r3}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 150 --warmup 10 --ramp-seconds 25 --ramp-cap-seconds 25 --sustained-seconds 10 "$@" > $OUT/bench_${TAG}_traced.json 2> $OUT/bench_${TAG}_traced.err || { tail -5 $OUT/bench_${TAG}_traced.err; exit 1; }
cd $GRAFT_REPO_ROOT
T=$(find $OUT/prof_$TAG -name "*kernel_trace.csv" | head -1)
python tools/trace_busy.py $T --tail-frac 0.3 > $OUT/bench_${TAG}_gpu_busy.json
S=$(find $OUT/prof_$TAG -name "*kernel_stats.csv" | head -1)
head -40 $S > $OUT/bench_${TAG}_kernel_stats.csv
rm -rf $OUT/prof_$TAG
cat $OUT/bench_${TAG}_gpu_busy.json
