#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprofv3 kernel stats.  Everything lands in gpurun_out/.
# usage: tools/gpu_round.sh [tag] [bench args...]
set -u
TAG=${1:-r01}
shift || true
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx|Compute Unit" | head -8 > gpurun_out/${TAG}_rocminfo.txt
nproc >> gpurun_out/${TAG}_rocminfo.txt
echo "== pytest -m gpu" 
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -60 | tee gpurun_out/${TAG}_pytest.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/${TAG}_smoke.log
echo "== bench"
timeout 1200 python bench.py "$@" > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -c 3000 gpurun_out/${TAG}_bench.json; tail -5 gpurun_out/${TAG}_bench.err
echo "== rocprofv3"
rm -rf gpurun_out/${TAG}_prof
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG} -o prof -- python ${GRAFT_REPO_ROOT:-/root/repo}/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency > /tmp/prof_${TAG}.log 2>&1)
tail -3 /tmp/prof_${TAG}.log
mkdir -p gpurun_out/${TAG}_prof
find /tmp/prof_${TAG} -name "*stats*.csv" -exec cp {} gpurun_out/${TAG}_prof/ \; 2>/dev/null
find /tmp/prof_${TAG} | head -20
ls -la gpurun_out/${TAG}_prof | head
