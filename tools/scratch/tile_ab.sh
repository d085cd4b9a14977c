set -u
export MSH_DEV_KNOBS=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for rep in 1 2; do
for E in "X=0" "MSH_LMHEAD_TALL=1"; do
  env $E timeout 600 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-c-api --no-fp8 > gpurun_out/r6ab.json 2> gpurun_out/r6ab.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r6ab.json").read().strip().splitlines()[-1])
print("$E rep $rep:", d["value"], d["ms_per_step"], "serial", d["serial_steps"]["value"], "ids", d["config"]["ids_match_serial_pass"], {k: round(v, 1) for k, v in d["decode_step_us"]["per_kernel_us"].items() if "gemm" in k})
PY
done
done
