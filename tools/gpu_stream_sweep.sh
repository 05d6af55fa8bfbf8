#!/bin/bash
# Development probe: the bench's overlapped rate against the number of streams and the waves per CU a launch takes.
# usage (GPU box): tools/gpu_stream_sweep.sh
for w in 6 8 12; do for s in 3 4 5 6; do
  r=$(SMPC_LONE_WAVES_PER_CU=$w python3 bench.py --no-cpu-baseline --no-extras --streams $s --steps 60 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('%.3f ms/step  %.2f M solves/s' % (d['ms_per_step'], d['value']/1e6))")
  echo "waves/CU per launch $w, streams $s: $r"
done; done
