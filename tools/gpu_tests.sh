#!/bin/bash
# parity + C-API tests and a default bench run on the GPU box
set -u
TAG=${1:-t}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
echo "== pytest -m gpu"
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v "^\[moonshine" | tail -40 | tee gpurun_out/${TAG}_pytest.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== bench"
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench.json"))
k=d.pop("kernels")
print(json.dumps(d))
for r in k: print(f"  {r['kernel']:24s} {r['bound']:5s} {r['achieved']:9.1f} {r['unit']:8s} frac={r['frac']:.3f} ms/launch={r['ms_per_launch']:.4f} total={r['total_ms']:.2f}")
PY
tail -2 gpurun_out/${TAG}_bench.err
