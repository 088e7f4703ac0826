#!/bin/bash
# same-box A/B of the two z-slide forms inside the sampling loop: one-wave-per-SIMD (default) vs matrix/helper-wave (CD_ZS_V1=1)
mkdir -p gpurun_out/ab1
for v in sw v1 sw v1; do
  if [ $v = v1 ]; then export CD_ZS_V1=1; else unset CD_ZS_V1; fi
  timeout -k 10 300 python bench.py --no-cpu --steps 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],2), round(d['config']['denoise_ms'],4), d['roofline']['avg_launch_us'], d['kernel_breakdown_ms_per_denoise'])" | tee -a gpurun_out/ab1/ab.log
done
