#!/bin/bash
# rocprofv3 kernel stats of the default bench command, one-wave-per-SIMD form vs matrix/helper-wave form (CD_ZS_V1=1)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/${1:-zsrp}; mkdir -p $out
for v in sw v1; do
  if [ $v = v1 ]; then export CD_ZS_V1=1; else unset CD_ZS_V1; fi
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$v -o bench -- python3 bench.py --steps 1 --warmup 1 --no-extra > $out/bench_$v.json 2> $out/err_$v.txt
  echo "== $v rc=$?"; grep zslide $(find $out/$v -name "*kernel_stats.csv") | cut -d, -f1-8 | cut -c40-300
  python3 -c "import json;d=json.load(open('$out/bench_$v.json'));print(d['value'],d['config']['denoise_ms'])"
  cp $(find $out/$v -name "*kernel_stats.csv") $out/kernel_stats_$v.csv; rm -rf $out/$v
done
