"""The 32-stream association-kernel measurement of bench.py on its own (for rocprofv3 PMC passes)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from strongsort_yolo_amd.config import StrongSortConfig
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
print(json.dumps(bench.batched_association(StrongSortConfig(), n_streams=n, frames=130, timed=20)))
