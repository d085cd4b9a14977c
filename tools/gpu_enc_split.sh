#!/bin/bash
# Encoder layer loop as two halves on two streams (MSH_ENC_SPLIT=1): bit check, then headline / serial A/B on one box.
set -u
export MSH_DEV_KNOBS=1
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
TAG=${1:-split}
timeout 600 python tools/scratch/enc_split_check.py 2>&1 | tail -4
for rep in 1 2; do
for N in 0 1 0q 1q; do
  Q=8; case $N in *q) Q=16;; esac; export GPU_MAX_HW_QUEUES=$Q
  MSH_ENC_SPLIT=${N%q} timeout 600 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-c-api --no-fp8 > gpurun_out/${TAG}_s${N}_${rep}.json 2> gpurun_out/${TAG}_s${N}_${rep}.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_s${N}_${rep}.json").read().strip().splitlines()[-1])
enc = sum(k["total_ms"] for k in d["kernels"] if k["kernel"].startswith(("enc_", "conv", "groupnorm", "pack")))
print("split $N rep $rep:", d["value"], d["ms_per_step"], "serial", d["serial_steps"]["value"], d["serial_steps"]["ms_per_step"], "ids", d["config"]["ids_match_serial_pass"])
PY
done
done
