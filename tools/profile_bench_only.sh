#!/bin/bash
# bash tools/profile_bench_only.sh <tag>: the bench lines and rocprofv3 kernel stats of profile_round.sh without its PMC / traffic / clock passes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/${1:-profb}; mkdir -p $out
stats() {
  name=$1; shift
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$name -o bench -- python3 bench.py --steps 1 --warmup 1 --no-extra --no-clocks "$@" > $out/${name}_bench_under_rocprof.json 2> $out/stats_$name.err; echo "stats $name rc=$?"
  cp $(find $out/stats_$name -name "*kernel_stats.csv" | head -1) $out/${name}_kernel_stats.csv; rm -rf $out/stats_$name
  timeout -k 10 400 python bench.py "$@" > $out/${name}_bench.json 2> $out/${name}_bench.err; echo "bench $name rc=$?"; cut -c1-160 $out/${name}_bench.json
}
stats dataset2
stats dataset3 --config dataset3
stats hgcal --config hgcal
