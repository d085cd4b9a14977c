set -u
export MSH_DEV_KNOBS=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for rep in 1 2; do
for F in 3 4 5 6; do
  timeout 600 python bench.py --in-flight $F --steps 12 --warmup 2 --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-c-api --no-fp8 > gpurun_out/r6ab.json 2> gpurun_out/r6ab.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r6ab.json").read().strip().splitlines()[-1])
print("in-flight $F rep $rep:", d["value"], d["ms_per_step"], "serial", d["serial_steps"]["value"], "ids", d["config"]["ids_match_serial_pass"])
PY
done
done
