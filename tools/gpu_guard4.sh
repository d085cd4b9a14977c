#!/bin/bash
# round 4: the kernels added this round (k_xattn.hip, k_crossq.hip), the decode GEMM tile order and the encoder attention
# under the guard allocator (MSH_GUARD_ALLOC=1: every buffer ends on an unmapped page) -- a slice of tools/gpu_guard.sh
set -u
TAG=${1:-guard4}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp MSH_GUARD_ALLOC=1
{
for f in tests/test_gpu_xattn.py tests/test_gpu_parity.py tests/test_gpu_guard.py tests/test_gpu_kv_fp8.py; do
  timeout 900 python -m pytest "$f" -m gpu -q -x -p no:cacheprovider > /tmp/guard_$(basename $f).log 2>&1
  echo "guard $(basename $f) rc=$?: $(tail -1 /tmp/guard_$(basename $f).log | cut -c1-160)"
  grep "FAILED\|Memory access fault" /tmp/guard_$(basename $f).log | cut -c1-250 | head -5
done
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('guard smoke ok')" 2>&1 | tail -1 | cut -c1-200
timeout 900 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --no-latency --no-streaming --no-c-api 2>/tmp/gb.err | cut -c1-220; tail -2 /tmp/gb.err | cut -c1-200
MSH_GUARD_ALIGN=16 MSH_GUARD_POISON=1 timeout 900 python -m pytest tests/test_gpu_xattn.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -1
} 2>&1 | tee gpurun_out/${TAG}_summary.txt
