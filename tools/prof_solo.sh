#!/bin/bash
# one-stream (bench.py --one-stream: engine stream mode 1) rocprofv3 kernel stats: every kernel alone on the chip, so the per-kernel
# durations are solo durations.  usage: tools/prof_solo.sh <tag> <bench args...>  -> gpurun_out/<tag>/kernel_stats.csv
TAG=$1; shift
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py "$@" --one-stream --no-cpu-baseline --no-profile > $OUT/bench.json 2> $OUT/bench.err
f=$(find $OUT/prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] || { echo "no kernel_stats.csv (see $OUT/bench.err)"; tail -5 $OUT/bench.err; exit 1; }
cp "$f" $OUT/kernel_stats.csv
rm -rf $OUT/prof
head -30 $OUT/kernel_stats.csv | cut -c1-180
tail -c 300 $OUT/bench.json
