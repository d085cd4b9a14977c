# generic in-flight A/B: bash tools/scratch/ab.sh "ENV1=a ENV2=b" "ENV1=c" ...   (each argument one environment setting; two repetitions)
set -u
export MSH_DEV_KNOBS=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for rep in 1 2; do
for E in "$@"; do
  env $E timeout 600 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-c-api --no-fp8 > gpurun_out/ab.json 2> gpurun_out/ab.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
    print("$E rep $rep:", d["value"], d["ms_per_step"], "serial", d["serial_steps"]["value"], "ids", d["config"]["ids_match_serial_pass"], "xattn", d["decode_step_us"]["per_kernel_us"]["dec_cross_attention"], "self", d["decode_step_us"]["per_kernel_us"]["dec_self_attention"])
except Exception as e:
    print("$E failed:", e, open("gpurun_out/ab.err").read()[-300:])
PY
done
done
