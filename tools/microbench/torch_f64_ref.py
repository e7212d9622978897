"""Known-good fp64 ceilings on this box via vendor libraries (rocBLAS / rocSOLVER through torch).
Test/bench infrastructure only -- never part of the product path."""
import time, torch
dev = "cuda:0"
print(torch.cuda.get_device_name(0))
def t(f, reps=3):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
for n in (4096, 8192, 16384):
    a = torch.randn(n, n, dtype=torch.float64, device=dev); b = torch.randn(n, n, dtype=torch.float64, device=dev)
    s = t(lambda: a @ b)
    print(f"torch fp64 matmul n={n}: {s*1e3:.2f} ms  {2*n**3/s*1e-12:.2f} TFLOP/s", flush=True)
    s = t(lambda: a @ b.T)
    print(f"torch fp64 matmul NT n={n}: {s*1e3:.2f} ms  {2*n**3/s*1e-12:.2f} TFLOP/s", flush=True)
n, k = 32768, 512
a = torch.randn(n, k, dtype=torch.float64, device=dev); c = torch.randn(n, n, dtype=torch.float64, device=dev)
s = t(lambda: torch.addmm(c, a, a.T, alpha=-1.0, out=c))
print(f"torch fp64 rank-512 update n={n}: {s*1e3:.2f} ms  {2*n*n*k/s*1e-12:.2f} TFLOP/s", flush=True)
del a, c
for n in (8192, 16384):
    x = torch.randn(n, n, dtype=torch.float64, device=dev); spd = x @ x.T + n * torch.eye(n, dtype=torch.float64, device=dev)
    s = t(lambda: torch.linalg.cholesky(spd), reps=2)
    print(f"torch fp64 cholesky n={n}: {s*1e3:.2f} ms  {n**3/3/s*1e-12:.2f} TFLOP/s", flush=True)
