#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/r5u_pytest.log 2>&1; tail -3 gpurun_out/r5u_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
MSH_CHAIN_MASKS=0xff timeout 120 python tools/chain_masks.py 2>&1 | grep "mask  0xff" | head -1 | cut -c1-80
