"""Is the long-K GEMM rate a clock (power) question?  Runs (a) the vendor DGEMM (torch.mm fp64 = rocBLAS / hipBLASLt, 16384^3) and (b) the
library's predict solve (solve_left_kernel: gprc_dev_solve_rows, n = 16384 x m = 65536) back to back for several seconds each, reports the
achieved TFLOP/s per ~half-second window (a short burst and a sustained run are different things) and samples rocm-smi (power, sclk)
beside both.     python tools/sustained_clock.py [seconds]"""
import os, subprocess, sys, threading, time, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gprc_amd as g
from gprc_amd import _native as nat

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
samples, stop = [], False


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            pw = re.findall(r"Power \(W\): ([0-9.]+)", out)
            sc = re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
            samples.append((time.perf_counter(), float(pw[0]) if pw else float("nan"), int(sc[0]) if sc else -1))
        except Exception as e:   # noqa: BLE001
            samples.append((time.perf_counter(), float("nan"), -1))
        time.sleep(0.25)


def window_report(name, marks, flops_each):
    t0 = marks[0]
    rates = []
    i0 = 0
    for i in range(1, len(marks)):
        if marks[i] - marks[i0] >= 0.5 or i == len(marks) - 1:
            rates.append((marks[i0] - t0, (i - i0) * flops_each / (marks[i] - marks[i0]) / 1e12))
            i0 = i
    print(f"{name}: " + " ".join(f"[{a:.1f}s {r:.1f}]" for a, r in rates), flush=True)
    sm = [s for s in samples if marks[0] <= s[0] <= marks[-1]]
    if sm:
        print(f"   rocm-smi beside it: power {min(s[1] for s in sm):.0f}..{max(s[1] for s in sm):.0f} W, sclk {min(s[2] for s in sm)}..{max(s[2] for s in sm)} MHz "
              f"({len(sm)} samples)", flush=True)


th = threading.Thread(target=sampler, daemon=True); th.start()
# (a) vendor DGEMM
n = 16384
A = torch.rand(n, n, dtype=torch.float64, device="cuda"); B = torch.rand(n, n, dtype=torch.float64, device="cuda"); Cm = torch.empty_like(A)
torch.mm(A, B, out=Cm); torch.cuda.synchronize()
marks = [time.perf_counter()]
while marks[-1] - marks[0] < secs:
    torch.mm(A, B, out=Cm); torch.cuda.synchronize(); marks.append(time.perf_counter())
window_report("vendor DGEMM 16384^3 (torch.mm fp64)", marks, 2.0 * n ** 3)
del A, B, Cm
torch.cuda.empty_cache()
time.sleep(2.0)
# (b) the library's predict solve
L = nat.lib()
m = 65536
n_pad = int(L.gprc_pad(n))
packed = (torch.rand(int(L.gprc_packed_size(n_pad)), dtype=torch.float64, device="cuda") - 0.5) * 0.02
winv = (torch.rand(int(L.gprc_winv_size(n_pad)), dtype=torch.float64, device="cuda") - 0.5) * 0.02
ld = m + 128
vt = (torch.rand(ld * n_pad, dtype=torch.float64, device="cuda") - 0.5)
st = torch.cuda.Stream()
ctx = nat.Context(0, st.cuda_stream)
torch.cuda.synchronize()
with torch.cuda.stream(st):
    nat.check(L.gprc_dev_solve_rows(ctx.handle, packed.data_ptr(), winv.data_ptr(), n_pad, vt.data_ptr(), ld, m)); st.synchronize()
    marks = [time.perf_counter()]
    while marks[-1] - marks[0] < secs:
        nat.check(L.gprc_dev_solve_rows(ctx.handle, packed.data_ptr(), winv.data_ptr(), n_pad, vt.data_ptr(), ld, m)); st.synchronize()
        marks.append(time.perf_counter())
window_report("library predict solve n=16384 m=65536 (whole call: left-looking passes + in-panel solves)", marks, float(m) * n_pad * n_pad)
stop = True
