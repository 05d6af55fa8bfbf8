#!/bin/bash
# Development probe: closed-loop tick time of sharded episodes (HIP graphs) under per-launch caps of the solve
# kernel's resident waves per CU. usage (GPU box): [SPECS="3 4;3 5"] tools/shard_caps.sh > gpurun_out/shard_caps.log
IFS=';' read -ra LIST <<< "${SPECS:-2 4;2 6;3 3;3 4;4 2;4 3}"
for spec in "${LIST[@]}"; do
  IFS=' ' read -r shards cap <<< "$spec"
  echo "shards $shards, cap $cap waves per CU per launch:"
  SMPC_SHARE=1 SMPC_MAX_WAVES_PER_CU=$cap timeout -k 10 120 python3 tools/gpu_episode.py 8192 8 10 plan -$shards 2>&1 | grep shards
done
