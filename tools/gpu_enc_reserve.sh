#!/bin/bash
# Headline run (4 batches in flight) with the lanes' encoders on a CU-masked stream: MSH_ENC_CU_RESERVE = CUs of every XCD left
# to the other lanes' decode chains.  Same box, interleaved.
set -u
export MSH_DEV_KNOBS=1
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
TAG=${1:-reserve}
for rep in 1 2; do
for N in ${2:-0 2 4 8}; do
  MSH_ENC_CU_RESERVE=$N timeout 600 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-c-api --no-fp8 > gpurun_out/${TAG}_r${N}_${rep}.json 2> gpurun_out/${TAG}_r${N}_${rep}.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_r${N}_${rep}.json").read().strip().splitlines()[-1])
print("reserve $N rep $rep:", d["value"], d["ms_per_step"], "serial", d["serial_steps"]["value"], "ids", d["config"]["ids_match_serial_pass"])
PY
done
done
