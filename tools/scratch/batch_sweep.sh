set -u
export MSH_DEV_KNOBS=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for cfg in "256 4" "512 4" "512 2" "1024 2" "1024 1" "2048 1"; do
  set -- $cfg
  timeout 600 python bench.py --batch $1 --in-flight $2 --steps 8 --warmup 1 --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-c-api --no-fp8 > gpurun_out/r6z_b$1_f$2.json 2> gpurun_out/r6z_b$1_f$2.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r6z_b$1_f$2.json").read().strip().splitlines()[-1])
    print("batch $1 in-flight $2:", d["value"], d["ms_per_step"], "serial", d["serial_steps"]["value"], d["serial_steps"]["ms_per_step"], "ids", d["config"]["ids_match_serial_pass"], "step_us", d.get("decode_step_us", {}).get("sum_of_chain_costs"))
except Exception as e:
    print("batch $1 in-flight $2 failed", e, open("gpurun_out/r6z_b$1_f$2.err").read()[-300:])
PY
done
