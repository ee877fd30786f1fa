# A/B of KFX_FUSE_XCD_SWIZZLE (fuse.hip, k_sdf_fuse_tiled): interleaved repetitions, boxes and runs differ by several per cent
for rep in 1 2 3; do for scene in room full; do for swz in 0 1; do
  KFX_FUSE_XCD_SWIZZLE=$swz python bench.py --scene $scene --math fast --steps 300 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rep$rep $scene swz=$swz', d['value'], d['kernels_ms']['sdf_fuse'], 'copy', d['roofline']['measured_copy_GBps'])"
done; done; done
