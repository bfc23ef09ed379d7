"""Same entry point as the reference script: `python yolo_multi_model.py --source ... --track --count`
(/root/reference/yolo_multi_model.py:341-354), running the MI355X hot path.  See strongsort_yolo_amd/cli.py."""
from strongsort_yolo_amd.cli import main

if __name__ == "__main__":
    main()
