#!/bin/bash
# round 6: a long soak of the fuzz groups on the final library: 300 seeds at x1, 60 at x8, 12 at x32
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
timeout -k 10 900 python tools/soak.py 1001 300 1 2>&1 | grep -v amdgpu.ids | tail -3
timeout -k 10 900 python tools/soak.py 2001 60 8 2>&1 | grep -v amdgpu.ids | tail -3
timeout -k 10 900 python tools/soak.py 3001 12 32 2>&1 | grep -v amdgpu.ids | tail -3
