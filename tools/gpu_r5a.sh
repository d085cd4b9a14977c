#!/bin/bash
# round 4: weight prefetch between decode GEMMs (MSH_DEC_PREFETCH): parity, the layer's kernels in sequence, bench
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
TAG=${1:-r5a}
{
timeout 900 python -m pytest tests/test_gpu_xattn.py tests/test_gpu_parity.py -q -x 2>&1 | tail -3
export MSH_CHAIN_MASKS=0x0c,0x60,0xc0,0xff
for P in 1 0 1 0; do echo "== prefetch=$P"; MSH_DEC_PREFETCH=$P timeout 300 python tools/chain_masks.py 2>&1 | grep -v amdgpu.ids | grep -v "round 1" | head -6; done
unset MSH_CHAIN_MASKS
FLAGS="--steps 16 --warmup 2 --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-c-api --no-fp8"
one() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], 'serial', d['config'].get('serial_steps_value'))"; }
for P in 1 0 1 0; do MSH_DEC_PREFETCH=$P timeout 300 python bench.py $FLAGS 2>/dev/null | one "prefetch=$P"; done
} 2>&1 | tee gpurun_out/${TAG}_prefetch.txt
