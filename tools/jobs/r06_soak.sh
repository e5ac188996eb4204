#!/bin/bash
# round 6: soak of the seeded fuzz groups (the two new ones included: intra frames through the persistent kernel, one-launch partitions) over other seeds, x1 and x4 batches
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
timeout -k 10 500 python tools/soak.py 1 12 1 2>&1 | grep -v amdgpu.ids
timeout -k 10 500 python tools/soak.py 101 6 4 2>&1 | grep -v amdgpu.ids
