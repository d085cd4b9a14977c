#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
TAG=${1:-r4w}
export MSH_CHAIN_MASKS=${MASKS:-0x3c,0x3d,0x3e,0x7c,0xbc,0x3f,0xfc,0xfe,0xfd,0x7f,0xbf,0xff,0xc3,0xcf,0xf3}
{
echo "== two-stage query kernel"; timeout 300 python tools/chain_masks.py 2>&1 | grep -v amdgpu.ids
echo "== merged-weight GEMM"; MSH_XATTN_QT=1 timeout 300 python tools/chain_masks.py 2>&1 | grep -v amdgpu.ids
} | tee gpurun_out/${TAG}_chain_masks.txt
