#!/bin/bash
# quick check after a kernel change: offline parity tests, one bench line, per-kernel times from rocprofv3
tag=${1:-quick}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_capi.py -m gpu -x -q > gpurun_out/${tag}_pytest.log 2>&1
tail -3 gpurun_out/${tag}_pytest.log
timeout 300 python bench.py --cpu-clips 0 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof -o ${tag} -- python $GRAFT_REPO_ROOT/bench.py --cpu-clips 0 --steps 3 --warmup 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import json, csv, glob
j=json.loads(open("gpurun_out/${tag}_bench.json").read().strip().splitlines()[-1])
print("bench", j["value"], j["ms_per_step"], j.get("roofline"))
for f in glob.glob("gpurun_out/${tag}_prof/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:22]:
        print("%-110s n=%6s avg=%10.1f us  %5.1f%%" % (r["Name"][:110], r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
