#!/bin/bash
set -u
TAG=${1:-ab4}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
echo "== pytest -m gpu"
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v "^\[moonshine" | tail -30
for v in 64 0; do
  echo "== bench MSH_DEC64_M=$v"
  MSH_DEC64_M=$v timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_d${v}.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench_d${v}.json"))
print("value",d["value"],"ms_per_step",d["ms_per_step"],"latency",d["latency_ms"])
for r in d["kernels"]: print(f"  {r['kernel']:24s} {r['bound']:5s} {r['achieved']:9.1f} {r['unit']:8s} frac={r['frac']:.3f} ms/launch={r['ms_per_launch']:.4f} total={r['total_ms']:.2f}")
PY
done
