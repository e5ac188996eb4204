#!/bin/bash
# the intra parity cases and timings on an A/B build of the library: tools/jobs/r06_intra_check_ab.sh <dir under hevcasm_amd/lib_ab>
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
export HEVCASM_HIP_LIB=$R/hevcasm_amd/lib_ab/$1/libhevcasm_hip.so
bash tools/jobs/r06_intra_check3.sh
