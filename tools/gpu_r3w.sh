#!/bin/bash
# r3w: streaming tests, streaming bench (plain), then kernel averages under rocprofv3
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_streaming.py tests/test_gpu_capi_streaming.py -q -x 2>&1 | tail -4
timeout 600 python bench.py --workload streaming --steps 2 --warmup 1 > gpurun_out/r3w_stream.json 2> gpurun_out/r3w_stream.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3w_stream.json").read().strip().splitlines()[-1])
print("plain: value", d["value"], "ms_per_step", d["ms_per_step"], {k:v for k,v in d["streaming"].items() if k not in ("kernels","roofline")})
PY
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/w -o w -- python $R/bench.py --workload streaming --steps 1 --warmup 1 --no-stream-profile > $R/gpurun_out/r3w_traced.json 2> $R/gpurun_out/r3w_traced.err)
python - <<'PY'
import csv, glob, json, re
d=json.loads(open("gpurun_out/r3w_traced.json").read().strip().splitlines()[-1])
print("traced: value", d["value"], "ms_per_step", d["ms_per_step"])
for f in glob.glob("/tmp/w/**/*kernel_stats.csv", recursive=True):
    import shutil; shutil.copy(f, "gpurun_out/r3w_stream_kernel_stats.csv")
    for r in list(csv.DictReader(open(f)))[:16]:
        n=re.sub(r"msh::\(anonymous namespace\)::","",r["Name"]); n=re.sub(r"^void ","",n).split("(")[0]
        print("   %-78s n=%6s avg=%8.2f us %5.1f%%"%(n[:78], r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
