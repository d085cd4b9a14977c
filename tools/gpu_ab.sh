#!/bin/bash
# A/B session: parity tests on the current build, then short benches under different MSH_GEMM_MODE values.
set -u
TAG=${1:-ab}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -40 | tee gpurun_out/${TAG}_pytest.log
for mode in 1 2 0; do
  echo "== bench MSH_GEMM_MODE=$mode"
  MSH_GEMM_MODE=$mode timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_mode${mode}.json 2> gpurun_out/${TAG}_bench_mode${mode}.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/${TAG}_bench_mode${mode}.json"))
    print("value",d["value"],"ms_per_step",d["ms_per_step"],"latency",d.get("latency_ms"))
    for r in d["kernels"]: print(f"  {r['kernel']:24s} {r['bound']:5s} {r['achieved']:9.1f} {r['unit']:8s} frac={r['frac']:.3f} ms/launch={r['ms_per_launch']:.4f} total={r['total_ms']:.2f}")
except Exception as e:
    print("bench failed", e); print(open("gpurun_out/${TAG}_bench_mode${mode}.err").read()[-2000:])
PY
done
