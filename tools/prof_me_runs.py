import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from hevcasm_amd import batch as B, workload as wl
rng = int(sys.argv[1]) if len(sys.argv) > 1 else 16
cs = [wl.cost_frame_case(1920, 1080, 950 + k, border=80) for k in range(3)]
res = torch.zeros((max(c["n_pus"] for c in cs), 3), dtype=torch.int32, device="cuda")
for c in cs:
    B.me_full_search_runs(B.u8(c["src"]), c["stride_src"], B.u8(c["ref"]), c["stride_ref"], B.off32(c["src_off"]), B.off32(c["ref_off"]), c["runs"], rng, out=res[:c["n_pus"]])
torch.cuda.synchronize()
print("done", [c["n_pus"] for c in cs])
