#!/bin/bash
# One GPU-box call that collects what profiles/ holds for a round:  bash tools/profile_round.sh <tag>
#   HBM traffic of the z-slide conv (PMC), its instruction / busy counters, rocprofv3 kernel stats of the default bench command,
#   and the bench lines of the other configurations (Dataset-3, HGCal, training).  Raw profiler output is deleted: gpurun only
#   merges back 64 MiB.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/${1:-prof}; mkdir -p $out
timeout -k 10 500 python3 tools/zs_traffic.py > $out/traffic.log 2>&1; echo "traffic rc=$?"; tail -3 $out/traffic.log
rm -rf gpurun_out/pmc_traffic
timeout -k 10 500 bash tools/zs_pmc.sh $(basename $out)_pmc > $out/pmc.log 2>&1; echo "pmc rc=$?"
cp gpurun_out/$(basename $out)_pmc/summary.txt $out/zslide_pmc_summary.txt; rm -rf gpurun_out/$(basename $out)_pmc; cat $out/zslide_pmc_summary.txt
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python3 bench.py --steps 1 --warmup 1 --no-cpu > $out/bench_under_rocprof.json 2> $out/stats.err; echo "stats rc=$?"
find $out/stats -name "*kernel_stats.csv" | head -3
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv; rm -rf $out/stats; tail -3 $out/stats.err
timeout -k 10 400 python bench.py --config dataset3 --batch 32 --no-cpu > $out/dataset3_bench.json 2> $out/d3.err; echo "d3 rc=$?"; tail -2 $out/d3.err
timeout -k 10 400 python bench.py --config hgcal --batch 16 --sample-steps 200 --no-cpu > $out/hgcal_bench.json 2> $out/hg.err; echo "hgcal rc=$?"; tail -2 $out/hg.err
timeout -k 10 400 python bench.py --mode train --no-cpu > $out/train_bench.json 2> $out/tr.err; echo "train rc=$?"; tail -2 $out/tr.err
du -sh gpurun_out; ls -la $out
