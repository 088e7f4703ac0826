#!/bin/bash
# One GPU-box call: the GPU test suite, then (unless the tests were killed by their timeout) the default bench line.
#   gpurun --timeout 1200 -- 'bash tools/gpu_check.sh <tag> [pytest args...]'
tag=${1:-check}; shift
out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -q "$@" > $out/gpu_tests.log 2>&1; rc=$?
echo "pytest rc=$rc" | tee -a $out/gpu_tests.log
tail -4 $out/gpu_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests timed out: no further GPU step"; exit $rc; fi
timeout -k 10 600 python bench.py > $out/bench.json 2> $out/bench.err; brc=$?
echo "bench rc=$brc"; cat $out/bench.json
exit $rc
