#!/bin/bash
# streaming workload bench + kernel-level profile
tag=${1:-sb}
mkdir -p gpurun_out
timeout 900 python bench.py --workload streaming --steps 2 --warmup 1 > gpurun_out/${tag}_stream_bench.json 2> gpurun_out/${tag}_stream_bench.err
tail -3 gpurun_out/${tag}_stream_bench.err
cat gpurun_out/${tag}_stream_bench.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof -- python $GRAFT_REPO_ROOT/bench.py --workload streaming --steps 1 --warmup 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/${tag}_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f gpurun_out/${tag}_stream_kernel_stats.csv && head -30 $f | cut -c1-200
rm -rf gpurun_out/${tag}_prof
