#!/bin/bash
# round 4 (last GPU minutes): streaming AR cross-attention with non-temporal memory loads (MSH_STREAM_XATTN_NT=1), config 5
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
{
for V in 0 1 0 1; do
  MSH_STREAM_XATTN_NT=$V timeout 60 python bench.py --workload streaming --steps 2 --warmup 1 --no-stream-profile 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stream xattn nt=$V', d['value'], d.get('ms_per_step'))"
done
} 2>&1 | tee gpurun_out/r5x_stream_nt.txt
