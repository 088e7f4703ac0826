#!/bin/bash
# A/B builds of the library in one GPU session: tools/ab_bench.sh <libA> <libB> [...]
for rep in 1 2; do
  for lib in "$@"; do
    CALODIFF_LIB=$lib timeout -k 10 300 python bench.py --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['value'],2), d['config']['denoise_ms'], d['kernel_breakdown_ms_per_denoise'])"
  done
done
