"""Times the trailing-update kernel alone (one launch: panel 0 applied to every other panel) on random data.

    python tools/gemm_bench.py [n] [reps]        GPRC_LIB_SUFFIX selects an experimental build of the library

Used for kernel-structure experiments where a full factorisation would be meaningless (ablated variants)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gprc_amd as g
from gprc_amd import _native as nat

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
L = nat.lib()
n_pad = int(L.gprc_pad(n)); P = int(L.gprc_panel_count(n_pad))
packed = (torch.rand(int(L.gprc_packed_size(n_pad)), dtype=torch.float64, device="cuda") - 0.5) * 0.02
ctx = nat.Context(0, torch.cuda.current_stream().cuda_stream)
L.gprc_prof_enable(1)
for it in range(reps + 2):
    if it == 2:
        torch.cuda.synchronize(); L.gprc_prof_reset()
    nat.check(L.gprc_dev_update_trailing(ctx.handle, packed.data_ptr(), n_pad, 0, 1, P, 1))
torch.cuda.synchronize()
for name, r in nat.prof_summary().items():
    if r["count"]:
        print(f"{name}: {r['count']} launches, {r['ms'] / r['count']:.3f} ms each, {r['flops'] / r['ms'] / 1e9:.2f} TFLOP/s")
