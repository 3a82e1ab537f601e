#!/bin/bash
# rocprofv3 kernel-trace stats of a bench.py invocation in ONE-stream mode (per-kernel durations = a kernel alone on the chip;
# bench.py's roofline line of the same --one-stream command must agree with these averages) -> gpurun_out/<tag>/kernel_stats.csv
# usage: tools/prof_stats.sh <tag> <bench args...>
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py "$@" --one-stream --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
f=$(find $OUT/prof -name '*kernel_stats.csv' | head -1)
cp "$f" $OUT/kernel_stats.csv 2>/dev/null
rm -rf $OUT/prof
head -40 $OUT/kernel_stats.csv | cut -c1-200
tail -c 400 $OUT/bench.json
