"""The multi-stream association-kernel measurement of bench.py on its own (for rocprofv3 PMC passes / kernel traces).
usage: python tools/batched_assoc.py [streams=32] [frame_batch=8] [identities=30] [W=1280] [H=720] [name=value ...]   (library options)
160 frames (galleries reach nn_budget = 100 rows at frame ~103), the last 32 timed."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from strongsort_yolo_amd.config import StrongSortConfig
a = [int(v) for v in sys.argv[1:] if "=" not in v]
opts = tuple(v for v in sys.argv[1:] if "=" in v)
n, fb, ids, W, H = (a + [32, 8, 30, 1280, 720][len(a):])[:5]
print(json.dumps(bench.batched_association(StrongSortConfig(), n_streams=n, n_ids=ids, W=W, H=H, frames=160, timed=32, frame_batch=fb, check=False, opts=opts)))
