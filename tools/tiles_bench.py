"""Time of one K = 512 trailing-update launch against its number of tiles (panel 0 applied to the panels [1, q_end)): separates the
fixed cost of a launch from the cost per generation of tiles.  python tools/tiles_bench.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gprc_amd
from gprc_amd import _native as nat
L = nat.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
n_pad = int(L.gprc_pad(n)); P = int(L.gprc_panel_count(n_pad))
st = torch.cuda.Stream()
ctx = nat.Context(0, st.cuda_stream)
with torch.cuda.stream(st):
    packed = (torch.rand(int(L.gprc_packed_size(n_pad)), dtype=torch.float64, device="cuda") - 0.5) * 0.02
    for q_end in [2, 3, 4, 5, 7, 9, 13, 17, 25, P]:
        if q_end > P: continue
        tiles = sum(16 * (P - q) - 6 for q in range(1, q_end))
        L.gprc_prof_reset(); L.gprc_prof_enable(1)
        for _ in range(6):
            nat.check(L.gprc_dev_update_trailing(ctx.handle, packed.data_ptr(), n_pad, 0, 1, q_end, 1))
        st.synchronize(); L.gprc_prof_enable(0)
        r = nat.prof_summary()["trailing_update"]
        ms = r["ms"] / r["count"]
        print(f"q_end={q_end:3d} tiles={tiles:6d} gens={tiles / 512:6.2f}  {ms * 1e3:8.1f} us  {r['flops'] / r['ms'] / 1e9:6.2f} TFLOP/s   us/gen={ms * 1e3 / (tiles / 512):6.1f}")
