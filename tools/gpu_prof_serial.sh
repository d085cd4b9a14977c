#!/bin/bash
# rocprofv3 kernel stats of the serial (one batch at a time) bench: per-kernel average durations
tag=${1:-prof}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof -o ${tag} -- python $GRAFT_REPO_ROOT/bench.py --cpu-clips 0 --steps 3 --warmup 1 --in-flight 1 --no-latency > $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench.err
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob
for f in glob.glob("gpurun_out/${tag}_prof/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:26]:
        print("%-105s n=%6s avg=%9.2f us  %5.1f%%" % (r["Name"][:105], r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
