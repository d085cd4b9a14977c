#!/bin/bash
# rocprofv3 kernel stats of the streaming workload (BASELINE config 5)
tag=${1:-sprof}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof -o ${tag} -- python $GRAFT_REPO_ROOT/bench.py --workload streaming --steps 1 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench.err
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, re
for f in glob.glob("gpurun_out/${tag}_prof/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:24]:
        n=re.sub(r"msh::\(anonymous namespace\)::","",r["Name"])
        print("%-100s n=%6s avg=%9.2f us  %5.1f%%" % (n[:100], r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
