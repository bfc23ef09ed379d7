"""The multi-stream association-kernel measurement of bench.py on its own (for rocprofv3 PMC passes).
usage: python tools/batched_assoc.py [streams=32] [frame_batch=8]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from strongsort_yolo_amd.config import StrongSortConfig
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
fb = int(sys.argv[2]) if len(sys.argv) > 2 else 8
print(json.dumps(bench.batched_association(StrongSortConfig(), n_streams=n, frames=128, timed=32, frame_batch=fb, check=False)))
