#!/bin/bash
# round 4: overlapped bench (4 / 5 lanes) against the workgroup shape of the absorbed cross-attention (how much of a CU it
# leaves to other lanes' kernels), the classic K/V form for reference, and what is resident when (kernel trace)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
TAG=${1:-r4t}
export TMPDIR=/tmp
FLAGS="--steps 16 --warmup 2 --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-c-api --no-fp8"
one() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], 'serial', d['config'].get('serial_steps_value'), 'xattn us', d['decode_step_us']['per_kernel_us'].get('dec_cross_attention'))"; }
{
for C in 84 81 42 43; do
  for F in 4 5; do
    MSH_XATTN_CFG=$C timeout 300 python bench.py --in-flight $F $FLAGS 2>/dev/null | one "cfg $C in-flight $F"
  done
done
MSH_XATTN_MIN_BATCH=100000 timeout 300 python bench.py --in-flight 4 $FLAGS 2>/dev/null | one "classic in-flight 4"
MSH_XATTN_MIN_BATCH=100000 timeout 300 python bench.py --in-flight 5 $FLAGS 2>/dev/null | one "classic in-flight 5"
} 2>&1 | tee gpurun_out/${TAG}_lanes_vs_shape.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_a -o t -- python $R/bench.py $FLAGS > /tmp/tr_a.log 2>&1)
f=$(find /tmp/tr_a -name "*kernel_trace.csv" | head -1)
{ tail -1 /tmp/tr_a.log | cut -c1-160; [ -n "$f" ] && python tools/trace_concurrency.py "$f"; } 2>&1 | tee gpurun_out/${TAG}_concurrency.txt
