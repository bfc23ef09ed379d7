"""bench.tracker_only (rows a6-a10 alone, one stream, 32 frames per call) with library options.  usage: python tools/tracker_only.py [name=value ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from strongsort_yolo_amd.config import StrongSortConfig
for rep in range(2):
    r = bench.tracker_only(StrongSortConfig(), opts=tuple(sys.argv[1:]))
    print(json.dumps({"opts": sys.argv[1:], **{k: v for k, v in r.items() if k != "note"}}), flush=True)
