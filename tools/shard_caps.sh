#!/bin/bash
# Development probe: closed-loop tick time of sharded episodes (HIP graphs) under per-launch caps of the solve
# kernel's resident waves per CU. usage (GPU box): tools/shard_caps.sh > gpurun_out/shard_caps.log
for spec in "2 4" "2 6" "3 3" "3 4" "4 2" "4 3"; do
  set -- $spec
  echo "shards $1, cap $2 waves per CU per launch:"
  SMPC_MAX_WAVES_PER_CU=$2 timeout -k 10 120 python3 tools/gpu_episode.py 8192 8 10 plan -$1 2>&1 | grep shards
done
