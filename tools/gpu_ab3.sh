#!/bin/bash
set -u
TAG=${1:-ab3}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print("value",d["value"],"ms_per_step",d["ms_per_step"],"latency",d.get("latency_ms"))
    for r in d["kernels"]: print(f"  {r['kernel']:24s} {r['bound']:5s} {r['achieved']:9.1f} {r['unit']:8s} frac={r['frac']:.3f} ms/launch={r['ms_per_launch']:.4f} total={r['total_ms']:.2f}")
except Exception as e:
    print("bench failed", e)
PY
}
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -30 | tee gpurun_out/${TAG}_pytest.log
echo "== bench b256"
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
show gpurun_out/${TAG}_bench.json; tail -2 gpurun_out/${TAG}_bench.err
echo "== rocprofv3 kernel stats b256"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG} -o prof -- python ${GRAFT_REPO_ROOT:-/root/repo}/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency > /tmp/prof_${TAG}.log 2>&1)
mkdir -p gpurun_out/${TAG}_prof
cp /tmp/prof_${TAG}/prof_kernel_stats.csv gpurun_out/${TAG}_prof/kernel_stats_b256.csv
echo "== rocprofv3 kernel stats b1"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}b1 -o prof -- python ${GRAFT_REPO_ROOT:-/root/repo}/bench.py --batch 1 --steps 5 --warmup 1 --no-cpu-baseline --no-latency > /tmp/prof_${TAG}b1.log 2>&1)
cp /tmp/prof_${TAG}b1/prof_kernel_stats.csv gpurun_out/${TAG}_prof/kernel_stats_b1.csv
python - <<PY
import csv
for f in ["gpurun_out/${TAG}_prof/kernel_stats_b256.csv","gpurun_out/${TAG}_prof/kernel_stats_b1.csv"]:
    rows=list(csv.DictReader(open(f)))
    print(f, len(rows))
    for r in rows[:24]:
        print("  %-100s calls=%s total_ms=%.3f avg_us=%.2f pct=%s" % (r["Name"].replace("msh::(anonymous namespace)::","")[:100], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, r["Percentage"]))
PY
