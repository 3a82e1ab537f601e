cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "split_products" -s 2>&1 | grep -E "relative L2|passed|failed|Error|assert" | head -20
for sp in 9 6; do
echo "== kernel tests FM_MFMA_SPLIT=$sp"
FM_MFMA_SPLIT=$sp timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|error" | tail -5
done
for sp in 0 9 6; do
  echo "== FM_MFMA_SPLIT=$sp"
  FM_MFMA_SPLIT=$sp python tools/probe_conv.py 256 0,1,6,11,16 2 2>&1 | grep conv
done
for sp in 0 9 6; do
FM_MFMA_SPLIT=$sp python bench.py --no-legs --no-cpu-baseline --sustain-s 0 --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r18 split $sp', d['ms_per_step'])"
done
