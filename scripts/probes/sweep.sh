timeout -k 10 300 python -u -m pytest tests/test_gpu_ops.py -x -q --timeout 120 --timeout-method thread -k "conv3x3" 2>&1 | tail -3 || exit 1
for b in 512 8192; do
  echo "== batch $b dconv"; timeout -k 10 120 python scripts/bench_conv.py --only fwd,dgrad --shapes L0b,L1b,L2b,L3b --batch $b 2>&1 | grep -v "^layer\|amdgpu.ids" || exit 1
done
