#!/bin/bash
# same-box A/B of the product library against another build of it:  bash tools/lib_ab.sh <tag> [bench args...]
#   (CD_BUILD_TAG=<tag> CD_EXTRA_HIPCC_FLAGS=... python -m calodiffusion_amd.build makes calodiffusion_amd/lib/libcalodiff_hip_<tag>.so)
tag="$1"; shift
mkdir -p gpurun_out/libab
for rep in 1 2; do
  for v in default $tag; do
    if [ $v = default ]; then unset CALODIFF_LIB; else export CALODIFF_LIB=$PWD/calodiffusion_amd/lib/libcalodiff_hip_$tag.so; fi
    CD_SKIP_SRCHASH=1 timeout -k 10 300 python bench.py --no-cpu --steps 2 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v $*', round(d['value'],2), round(d['config'].get('denoise_ms', d['ms_per_step']),4), d.get('roofline',{}).get('avg_launch_us'))" | tee -a gpurun_out/libab/ab.log
  done
done
