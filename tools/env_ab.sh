#!/bin/bash
# same-box A/B of an environment switch of the product library, alternating runs:  bash tools/env_ab.sh CD_ATTN_COOP=4 [bench args...]
kv="$1"; shift
mkdir -p gpurun_out/envab
for rep in 1 2; do
  for v in default "$kv"; do
    if [ "$v" = default ]; then pre=""; else pre="$v"; fi
    env $pre timeout -k 10 300 python bench.py --no-extra --steps 2 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v $*', round(d['value'],2), round(d['config'].get('denoise_ms', d['ms_per_step']),4))" | tee -a gpurun_out/envab/ab.log
  done
done
