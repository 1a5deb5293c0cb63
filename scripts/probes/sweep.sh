timeout -k 10 300 python -u -m pytest tests/test_gpu_ops.py tests/test_zz_no_alloc.py -x -q --timeout 120 --timeout-method thread -k "conv3x3 or static" 2>&1 | tail -3 || exit 1
for b in 512; do
  echo "== batch $b"; timeout -k 10 120 python scripts/bench_conv.py --only wgrad --shapes L0b,L0c,L1b,L1c,L2b,L2c,L3b,BNb --batch $b 2>&1 | grep -v "^layer\|amdgpu.ids" || exit 1
  echo "== batch $b old"; UNET_TM_HWGRAD_OLD=1 timeout -k 10 120 python scripts/bench_conv.py --only wgrad --shapes L0b --batch $b 2>&1 | grep -v "^layer\|amdgpu.ids" || exit 1
done
