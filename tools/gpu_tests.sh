#!/bin/bash
# GPU parity tests only (no -x: report every failure).  Log under gpurun_out/<tag>/pytest.log
TAG=${1:-tests}
shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
python -m pytest tests -m gpu -q "$@" 2>&1 | tail -150 > $OUT/pytest.log
tail -25 $OUT/pytest.log
