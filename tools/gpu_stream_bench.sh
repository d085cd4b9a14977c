#!/bin/bash
# Streaming engine: its GPU tests, then BASELINE config 5 (64 streams x 10 s, 0.5 s updates, speculative) as its own bench line.
set -u
export MSH_DEV_KNOBS=1   # the library reads its developer switches only with this set
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
TAG=${1:-stream}
timeout 900 python -m pytest tests/test_gpu_streaming.py tests/test_gpu_capi_streaming.py -m gpu -q > gpurun_out/${TAG}_pytest.log 2>&1
tail -4 gpurun_out/${TAG}_pytest.log
timeout 600 python bench.py --workload streaming --steps 3 --warmup 1 > gpurun_out/${TAG}_bench_streaming.json 2> gpurun_out/${TAG}_bench_streaming.err
python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_bench_streaming.json").read().strip().splitlines()[-1])
s = d["streaming"]
print(d["value"], d["ms_per_step"], {k: s[k] for k in ("ms_per_update", "frontend_ms_per_step", "encode_ms_per_step", "decode_ms_per_step", "draft_acceptance")})
print(s["decoder_passes"])
PY
