#!/bin/bash
# bash tools/logrun.sh <log> <command...>: the command line first, then stdout AND stderr of the command, then its exit status, in
# one file -- so that a scratch log under gpurun_out/ always says what produced it (round 3's close.log did not).
log="$1"; shift
mkdir -p "$(dirname "$log")"
{ echo "+ $*"; "$@"; rc=$?; echo "[logrun] exit status $rc"; } > "$log" 2>&1
exit $rc
