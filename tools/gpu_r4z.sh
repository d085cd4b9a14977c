#!/bin/bash
# slow processes of the decode sequence (seen on some boxes only): which of kernel / allocation layout triggers them?
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
TAG=${1:-r4z}
export MSH_CHAIN_MASKS=0xff
{
for i in 1 2 3; do
  for V in "1 0" "0 0" "0 1" "0 2" "1 1"; do
    set -- $V
    echo -n "arena=$1 qt=$2 run $i: "; MSH_WEIGHT_ARENA=$1 MSH_XATTN_QT=$2 timeout 300 python tools/chain_masks.py 2>&1 | grep "mask  0xff" | head -1 | sed 's/.*\]: //'
  done
done
} | tee gpurun_out/${TAG}_arena.txt
