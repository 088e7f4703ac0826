#!/bin/bash
# same-box A/B of the default path against one environment switch:  bash tools/env_ab.sh CD_NO_BLOCK_SMALL=1 [bench args...]
sw="$1"; shift
mkdir -p gpurun_out/envab
for rep in 1 2; do
  for v in default "$sw"; do
    if [ "$v" = default ]; then e=""; else e="$v"; fi
    env $e timeout -k 10 300 python bench.py --no-extra --steps 2 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v $*', round(d['value'],2), round(d['config']['denoise_ms'],4))" | tee -a gpurun_out/envab/ab.log
  done
done
