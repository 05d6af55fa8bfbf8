#!/bin/bash
# Round profile: kernel trace + stats of the DEFAULT bench command (4 overlapped streams), of the single-stream bench,
# of the closed-loop chain, and PMC passes of the same workload.
# usage (on the GPU box, via gpurun): tools/profile_round.sh r02
set -u
TAG=${1:-r02}
OUT=gpurun_out/profile_$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
# (1) the default bench (what the driver runs, minus the extras that launch other kernels): per-dispatch begin / end
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_trace" -- python3 bench.py --no-cpu-baseline --no-extras > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench_under_rocprof.err"
cp $(find "$OUT/bench_trace" -name "*kernel_stats.csv" | head -1) "$OUT/${TAG}_bench_kernel_stats.csv"
python3 tools/trace_extract.py $(find "$OUT/bench_trace" -name "*kernel_trace.csv" | head -1) > "$OUT/${TAG}_bench_trace_overlap.txt"
# (2) the same bench on ONE stream (lone launches: what roofline.launch_ms_lone refers to)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench1_trace" -- python3 bench.py --steps 10 --warmup 2 --streams 1 --no-cpu-baseline --no-extras > "$OUT/bench1_under_rocprof.json" 2> "$OUT/bench1_under_rocprof.err"
cp $(find "$OUT/bench1_trace" -name "*kernel_stats.csv" | head -1) "$OUT/${TAG}_bench_streams1_kernel_stats.csv"
# (3) the closed-loop chain (rows f1-f3 + solve + store) under the same tracer
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/episode_trace" -- python3 tools/gpu_episode.py 8192 8 10 window 0 > "$OUT/episode_under_rocprof.log" 2>&1
grep -E "^\"Name\"|smpc" $(find "$OUT/episode_trace" -name "*kernel_stats.csv" | head -1) > "$OUT/${TAG}_episode_kernel_stats.csv"
# (3b) the same robots as three shards replayed from HIP graphs: begin / end of every kernel (overlap across streams)
rocprofv3 --kernel-trace --output-format csv -d "$OUT/episode3_trace" -- python3 tools/gpu_episode.py 8192 8 6 window -3 > "$OUT/episode3_under_rocprof.log" 2>&1
python3 tools/trace_extract.py $(find "$OUT/episode3_trace" -name "*kernel_trace.csv" | head -1) | tail -70 > "$OUT/${TAG}_episode_sharded_trace.txt"
# (4) PMC passes (separate runs, counters only)
tools/prof_pmc.sh "$OUT/pmc" > /dev/null 2>&1
cp "$OUT/pmc/pmc_summary.txt" "$OUT/${TAG}_pmc_summary.txt"
# (5) the plain default bench of the same box
python3 bench.py > "$OUT/${TAG}_bench.json" 2> "$OUT/bench.err"
head -c 1500 "$OUT/${TAG}_bench_kernel_stats.csv"; echo; head -30 "$OUT/${TAG}_bench_trace_overlap.txt"; cat "$OUT/${TAG}_pmc_summary.txt" | grep -E "==|FETCH|WRITE|SQ_INSTS_VALU |SQ_INSTS_MFMA|SQ_WAVES|GRBM"; cat "$OUT/${TAG}_bench.json"
