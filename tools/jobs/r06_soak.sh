#!/bin/bash
# round 6: soak of the seeded fuzz groups (incl. intra frames through the persistent kernel and the one-launch partitions: prediction, block costs) over other seeds, x1 / x4 / x16 batches
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
timeout -k 10 500 python tools/soak.py 1 40 1 2>&1 | grep -v amdgpu.ids
timeout -k 10 500 python tools/soak.py 101 20 4 2>&1 | grep -v amdgpu.ids
timeout -k 10 500 python tools/soak.py 201 6 16 2>&1 | grep -v amdgpu.ids
