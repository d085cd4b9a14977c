#!/bin/bash
# Streaming engine: its GPU tests, then BASELINE config 5 once per given environment setting (same box).
set -u
export MSH_DEV_KNOBS=1
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
TAG=${1:-stream_ab}
timeout 900 python -m pytest tests/test_gpu_streaming.py tests/test_gpu_capi_streaming.py -m gpu -q > gpurun_out/${TAG}_pytest.log 2>&1
tail -4 gpurun_out/${TAG}_pytest.log
i=0
for E in "${@:2}"; do
  i=$((i+1))
  env $E timeout 600 python bench.py --workload streaming --steps 3 --warmup 1 > gpurun_out/${TAG}_${i}_bench_streaming.json 2> gpurun_out/${TAG}_${i}_bench_streaming.err
  echo "== $i: $E"
  python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_${i}_bench_streaming.json").read().strip().splitlines()[-1])
s = d["streaming"]
print(d["value"], d["ms_per_step"], {k: s[k] for k in ("frontend_ms_per_step", "encode_ms_per_step", "decode_ms_per_step", "draft_acceptance")})
PY
done
