import sys, os, traceback
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from kangaroo_amd import roo
import test_gpu_fuzz as F
bad = []
first, count = int(sys.argv[1]), int(sys.argv[2])
for seed in range(first, first + count):
    try:
        F.test_gpu_fuzz_tiled_handover_any_camera(roo, seed)
    except Exception:
        bad.append(seed); print("FAIL", seed); traceback.print_exc(limit=3)
print("any-camera soak: %d seeds, %d failures %s" % (count, len(bad), bad[:20]))
sys.exit(1 if bad else 0)
