"""Times the predict solve vt := vt L^-T alone (gprc_dev_solve_rows) on random data: the left-looking passes (solve_left_kernel,
the dominant kernel of the C4 step) at C4's row count with a smaller factor, so that one call takes a fraction of a second.

    python tools/solve_bench.py [n] [m] [reps]        GPRC_LIB_SUFFIX selects an experimental build of the library

Prints per-kind launch counts, milliseconds and algorithmic TFLOP/s from the library's own HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gprc_amd as g
from gprc_amd import _native as nat

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
m = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
L = nat.lib()
n_pad = int(L.gprc_pad(n)); P = int(L.gprc_panel_count(n_pad))
packed = (torch.rand(int(L.gprc_packed_size(n_pad)), dtype=torch.float64, device="cuda") - 0.5) * 0.02
winv = (torch.rand(int(L.gprc_winv_size(n_pad)), dtype=torch.float64, device="cuda") - 0.5) * 0.02
ld = m + 128
vt0 = (torch.rand(ld * n_pad, dtype=torch.float64, device="cuda") - 0.5)
vt = torch.empty_like(vt0)
st = torch.cuda.Stream()
ctx = nat.Context(0, st.cuda_stream)
with torch.cuda.stream(st):
    for it in range(reps + 1):
        vt.copy_(vt0)
        if it == 1:
            st.synchronize(); L.gprc_prof_reset(); L.gprc_prof_enable(1)
        nat.check(L.gprc_dev_solve_rows(ctx.handle, packed.data_ptr(), winv.data_ptr(), n_pad, vt.data_ptr(), ld, m))
    st.synchronize()
L.gprc_prof_enable(0)
tot = 0.0
for name, r in nat.prof_summary().items():
    if r["count"]:
        tot += r["ms"]
        print(f"{name}: {r['count']} launches, {r['ms'] / r['count']:.3f} ms each, {r['flops'] / r['ms'] / 1e9:.2f} TFLOP/s")
print(f"n={n} m={m}: {tot / reps:.2f} ms per solve, {float(m) * n_pad * n_pad / (tot / reps) * 1e-9:.2f} TFLOP/s overall", flush=True)
