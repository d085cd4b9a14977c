#!/bin/bash
# is the slow state of the two-stage variant (seen on two of four boxes) tied to the kernel or to the allocation layout?
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
TAG=${1:-r4y}
export MSH_CHAIN_MASKS=0x3c,0xc3,0xff
{
for Q in 0 1 2 0; do echo "== MSH_XATTN_QT=$Q"; MSH_XATTN_QT=$Q timeout 300 python tools/chain_masks.py 2>&1 | grep -v amdgpu.ids | grep -v "round 1" | head -5; done
} | tee gpurun_out/${TAG}_chain_masks.txt
