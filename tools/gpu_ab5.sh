#!/bin/bash
# A/B of the A-stationary kernel: microbench, then parity tests + bench with MSH_GEMM_MODE=4.
tag=${1:-ab5}
mkdir -p gpurun_out
timeout 300 python tools/gemm_microbench.py 2>&1 | grep -E "astat|cfg0" > gpurun_out/${tag}_micro.txt
MSH_GEMM_MODE=4 timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest_mode4.log 2>&1
tail -3 gpurun_out/${tag}_pytest_mode4.log
MSH_GEMM_MODE=4 timeout 300 python bench.py --cpu-clips 0 > gpurun_out/${tag}_bench_mode4.json 2> gpurun_out/${tag}_bench_mode4.err
timeout 300 python bench.py --cpu-clips 0 > gpurun_out/${tag}_bench_mode2.json 2> gpurun_out/${tag}_bench_mode2.err
cat gpurun_out/${tag}_micro.txt
python - <<PY
import json
for m in ("mode4","mode2"):
    try:
        j=json.loads(open("gpurun_out/${tag}_bench_%s.json"%m).read().strip().splitlines()[-1])
        print(m, j["value"], j["ms_per_step"])
        for k in j.get("kernels",[])[:14]: print("   ",k)
    except Exception as e: print(m,"ERR",e)
PY
