#!/bin/bash
# One GPU-box call that collects what profiles/ holds for a round:  bash tools/profile_round.sh <tag>
#   * HBM traffic (PMC) of the z-slide conv at the level-0 grid of every sampling configuration of BASELINE.json,
#   * its instruction / busy counters,
#   * HBM traffic (PMC) of the level-0 HBM passes (gn_apply, attention, shortcut conv, head, init conv) against their algorithmic bytes,
#   * rocprofv3 kernel stats of the headline bench command WITHOUT the side legs (--no-extra: the percentages are those of the
#     timed region), and of the Dataset-3, HGCal (the config's own 200-step DDPM) and training (batch 32) configurations,
#   * the bench lines of the same commands outside the profiler (with the clock / power sampled during the timed region).
# Raw profiler output is deleted: gpurun only merges back 64 MiB.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/${1:-prof}; mkdir -p $out
for cfg in dataset2 dataset3 hgcal; do
  timeout -k 10 400 python3 tools/zs_traffic.py --config $cfg > $out/traffic_$cfg.log 2>&1; echo "traffic $cfg rc=$?"; tail -1 $out/traffic_$cfg.log
  rm -rf gpurun_out/pmc_traffic
done
timeout -k 10 400 python3 tools/l0_traffic.py > $out/l0_traffic.log 2>&1; echo "l0 traffic rc=$?"; cp gpurun_out/l0_traffic_dataset2.json $out/ 2>/dev/null; rm -rf gpurun_out/pmc_l0
timeout -k 10 500 bash tools/zs_pmc.sh $(basename $out)_pmc > $out/pmc.log 2>&1; echo "pmc rc=$?"
cp gpurun_out/$(basename $out)_pmc/summary.txt $out/zslide_pmc_summary.txt; rm -rf gpurun_out/$(basename $out)_pmc; cat $out/zslide_pmc_summary.txt
stats() {  # stats <name> <bench args...>
  name=$1; shift
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$name -o bench -- python3 bench.py --steps 1 --warmup 1 --no-extra --no-clocks "$@" > $out/${name}_bench_under_rocprof.json 2> $out/stats_$name.err; echo "stats $name rc=$?"
  cp $(find $out/stats_$name -name "*kernel_stats.csv" | head -1) $out/${name}_kernel_stats.csv; rm -rf $out/stats_$name
  timeout -k 10 400 python bench.py "$@" > $out/${name}_bench.json 2> $out/${name}_bench.err; echo "bench $name rc=$?"; cut -c1-160 $out/${name}_bench.json
}
stats dataset2
stats dataset3 --config dataset3
stats hgcal --config hgcal
bash tools/train_prof.sh $(basename $out)_train > $out/train_prof.log 2>&1; echo "train prof rc=$?"  # (timed steps only: trace_stats.py)
for f in train_bench.json train_bench_under_rocprof.json train_kernel_stats.csv; do cp gpurun_out/$(basename $out)_train/$f $out/ 2>/dev/null; done; head -8 $out/train_prof.log | cut -c1-200
timeout -k 10 120 python tools/clock_trace.py --out $out/clock_trace.json > $out/clock_trace.log 2>&1; echo "clock trace rc=$?"
du -sh gpurun_out; ls -la $out
