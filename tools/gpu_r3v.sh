#!/bin/bash
# r3v: streaming AR steps on FM operands: parity tests, then tile-shape sweep under rocprofv3 (kernel averages + bench value)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_streaming.py tests/test_gpu_capi_streaming.py -q -x 2>&1 | tail -4
for cfg in 0000 1222 4141 2151 off; do
  if [ "$cfg" = off ]; then export MSH_STREAM_FM=0; unset MSH_SFM_CFG; else unset MSH_STREAM_FM; export MSH_SFM_CFG=$cfg; fi
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/v_$cfg -o v -- python $R/bench.py --workload streaming --steps 1 --warmup 1 --no-stream-profile > $R/gpurun_out/r3v_$cfg.json 2> $R/gpurun_out/r3v_$cfg.err)
  python - "$cfg" <<'PY'
import csv, glob, json, re, sys
cfg=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/r3v_{cfg}.json").read().strip().splitlines()[-1])
    print(f"== cfg {cfg}: value {d['value']} ms_per_step {d['ms_per_step']} decode_ms {d['streaming']['decode_ms_per_step']}")
except Exception as e:
    print("== cfg", cfg, "bench failed", e)
for f in glob.glob(f"/tmp/v_{cfg}/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]:
        n=re.sub(r"msh::\(anonymous namespace\)::","",r["Name"]); n=re.sub(r"^void ","",n).split("(")[0]
        print("   %-78s n=%6s avg=%8.2f us %5.1f%%"%(n[:78], r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
done
