#!/bin/bash
# Round profile: kernel-trace stats of the bench command + PMC passes of the same workload.
# usage (on the GPU box, via gpurun): tools/profile_round.sh r01
set -u
TAG=${1:-r01}
OUT=gpurun_out/profile_$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_trace" -- python3 bench.py --steps 10 --warmup 2 --streams 1 --no-cpu-baseline --no-extras > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench_under_rocprof.err"
cp $(find "$OUT/bench_trace" -name "*kernel_stats.csv" | head -1) "$OUT/${TAG}_bench_kernel_stats.csv"
# the closed-loop chain (rows f1-f3 + solve + store) under the same tracer
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/episode_trace" -- python3 tools/gpu_episode.py 8192 8 10 plan > "$OUT/episode_under_rocprof.log" 2>&1
grep -E "^\"Name\"|smpc" $(find "$OUT/episode_trace" -name "*kernel_stats.csv" | head -1) > "$OUT/${TAG}_episode_kernel_stats.csv"
tools/prof_pmc.sh "$OUT/pmc" > /dev/null 2>&1
cp "$OUT/pmc/pmc_summary.txt" "$OUT/${TAG}_pmc_summary.txt"
python3 bench.py > "$OUT/${TAG}_bench.json" 2> "$OUT/bench.err"
head -c 1500 "$OUT/${TAG}_bench_kernel_stats.csv"; echo; cat "$OUT/${TAG}_pmc_summary.txt" | grep -E "==|FETCH|WRITE|SQ_INSTS_VALU |SQ_INSTS_MFMA|SQ_WAVES|GRBM"; cat "$OUT/${TAG}_bench.json"
