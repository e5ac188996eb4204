#!/bin/bash
# builds hevcasm_amd/lib_ab/if_stamps/libhevcasm_hip.so (intra_frame.hip with -DIF_STAMPS) -- run here; on the GPU box: tools/jobs/r06_intra_stamps.sh run
set -e
cd "$(dirname "$0")/../.."
if [ "$1" != "run" ]; then
  cd hevcasm_amd/csrc
  mkdir -p build_ab ../lib_ab/if_stamps
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -Wall -Wno-unused-result -Wno-unused-variable -Wno-unused-but-set-variable -I../../include -Ihost -DIF_STAMPS -c hip/intra_frame.hip -o build_ab/hip_intra_frame_stamps.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib_ab/if_stamps/libhevcasm_hip.so $(ls build/host_*.o build/hip_*.o | grep -v "hip_intra_frame.o\|pattern") build_ab/hip_intra_frame_stamps.o
  ls -la ../lib_ab/if_stamps/libhevcasm_hip.so
  exit 0
fi
HEVCASM_HIP_LIB=$PWD/hevcasm_amd/lib_ab/if_stamps/libhevcasm_hip.so timeout -k 10 200 python - <<'PY'
import ctypes, sys, numpy as np, torch
sys.path.insert(0, ".")
from hevcasm_amd import batch as B, workload as wl, capi
L = capi.lib()
fn = ctypes.CDLL(capi.LIB_PATH).hevcasm_hip_debug_intra_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
w = h = ctu = 64
F = 256
for name, split in (("all_4x4", (1.0, 1.0, 1.0, 1.0)), ("mixed", (1.0, 0.7, 0.5, 0.35))):
    maps = [wl.intra_tu_map(w, h, ctu, 100 + f, modes="mixed", split=split) for f in range(4)]
    job = B.IntraFrames(w, h, ctu, [maps[f % 4][0] for f in range(F)], [maps[f % 4][1] for f in range(F)], wl.intra_qp(27), c_idx=0, strong=1, stride=w)
    src = torch.cat([B.u8(wl.frame_2d(w, h, w, 7 + f)) for f in range(4)] * (F // 4)).contiguous()
    job.run(src); torch.cuda.synchronize()
    fn(None, 1)
    job.run(src); torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 8)()
    fn(out, 0)
    v = list(out)
    n4 = sum(int((maps[f % 4][0]["log2"] == 2).sum()) for f in range(F))
    print(name, "4x4 TUs", n4, "all TUs", v[5])
    print("  per 4x4 TU, s_memtime ticks (100 MHz? shader clock? -- ratios matter): fetch+ballot %.1f  predict %.1f  transform..rec %.1f  stores %.1f" % tuple(v[k] / max(n4, 1) for k in range(4)))
    print("  TU loop total per TU %.1f" % (v[4] / max(v[5], 1)))
PY
