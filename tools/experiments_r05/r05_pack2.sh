#!/bin/bash
cd $GRAFT_REPO_ROOT
for o in "assoc_pack=1" "assoc_pack=0" "assoc_pack=1 ids=100" "assoc_pack=0 ids=100" "assoc_pack=1 ids=8" "assoc_pack=0 ids=8" "assoc_pack=1 ids=24" "assoc_pack=0 ids=24"; do timeout 120 python tools/assoc_time.py $o 2>/dev/null; done
