#!/bin/bash
# A/B one environment switch in one GPU session: tools/ab_env.sh "VAR=a" "VAR=b" [bench args]
A=$1; B=$2; shift 2
for rep in 1 2; do
  for e in "$A" "$B"; do
    env $e timeout -k 10 300 python bench.py --no-cpu "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$e', round(d['value'],2), d['config']['denoise_ms'], d['kernel_breakdown_ms_per_denoise'])"
  done
done
