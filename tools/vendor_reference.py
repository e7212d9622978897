"""An fp64 second opinion that shares nothing with libgprc_native: the GP predict step (R/GPRclass.R:127-165) built from
vendor LAPACK / BLAS on the GPU through torch.linalg (rocSOLVER potrf, rocBLAS trsm / gemm).  Used by
tests/test_gpu_fullsize.py (2048 rows of the timed configurations) and by bench.py (256 rows of the LAST TIMED step's
outputs: the variance leg that is independent of the library).  A floating-point kernel, so a torch fp64 reference is the
allowed checker; it is test / measurement infrastructure and never on the product path."""
import numpy as np
import torch


def lapack_reference_subset(kind, X, y, Xs_sub, noise, slab=4096, device=0):
    """Independent fp64 second opinion at sizes up to n = 65536 (K = 34 GB of the 288, factored in place): K is built in
    row slabs with direct (x - y)^2 sums, factored by the vendor Cholesky (torch.linalg -> rocSOLVER potrf, rocBLAS
    trsm / gemm), and only the SUBSET of test points is predicted.  Returns alpha, mean, var (numpy)."""
    dev = torch.device("cuda", int(device))
    Xt, yt, Xst = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (X.T, y, Xs_sub.T))
    n, d = Xt.shape

    def kern_into(out, A, B):
        out.zero_()
        for r in range(d):
            out.add_((A[:, r, None] - B[None, :, r]) ** 2)
        if kind == "sqrexp":
            out.mul_(-0.5).exp_()
        else:
            out.div_(2 * 1.5).add_(1.0).pow_(-1.5)
        return out

    K = torch.empty(n, n, dtype=torch.float64, device=dev)
    for c0 in range(0, n, slab):
        c1 = min(n, c0 + slab)
        kern_into(K[c0:c1, :], Xt[c0:c1], Xt)            # row slab of the symmetric K: contiguous in torch's row-major
    K.diagonal().add_(noise)
    # Vendor Cholesky, blocked by hand above 32768: torch.linalg.cholesky (hipSOLVER/rocSOLVER potrf) rejects n = 65536
    # with "invalid configuration argument" on this stack, so the factorisation runs as a right-looking sweep over
    # 16384-wide block columns built from the SAME vendor pieces -- potrf on the diagonal block, rocBLAS trsm and gemm
    # for the rest -- in place in K's lower triangle.  Nothing of this library is involved.
    nb = n if n <= 32768 else 16384
    for k0 in range(0, n, nb):
        k1 = min(n, k0 + nb)
        K[k0:k1, k0:k1] = torch.linalg.cholesky(K[k0:k1, k0:k1])
        if k1 < n:
            Lkk = K[k0:k1, k0:k1]
            K[k1:, k0:k1] = torch.linalg.solve_triangular(Lkk, K[k1:, k0:k1].T, upper=False).T     # L21 = K21 L11^-T
            for j0 in range(k1, n, nb):                                                               # trailing block columns
                j1 = min(n, j0 + nb)
                K[j0:, j0:j1] -= K[j0:, k0:k1] @ K[j0:j1, k0:k1].T
    L = K   # lower triangle = the factor; the strict upper part is never read below

    def forward(B):      # L^-1 B by block forward substitution (B: n x m), in place
        for k0 in range(0, n, nb):
            k1 = min(n, k0 + nb)
            if k0:
                B[k0:k1] -= L[k0:k1, :k0] @ B[:k0]
            B[k0:k1] = torch.linalg.solve_triangular(torch.tril(L[k0:k1, k0:k1]), B[k0:k1], upper=False)
        return B

    def backward(B):     # L^-T B
        for k1 in range(n, 0, -nb):
            k0 = max(0, k1 - nb)
            if k1 < n:
                B[k0:k1] -= L[k1:, k0:k1].T @ B[k1:]
            B[k0:k1] = torch.linalg.solve_triangular(torch.tril(L[k0:k1, k0:k1]).T, B[k0:k1], upper=True)
        return B

    alpha = backward(forward(yt[:, None].clone()))[:, 0]
    Ks = kern_into(torch.empty(n, Xst.shape[0], dtype=torch.float64, device=dev), Xt, Xst)
    mean = Ks.T @ alpha
    v = forward(Ks)
    var = 1.0 - (v * v).sum(0)
    out = alpha.cpu().numpy(), mean.cpu().numpy(), var.cpu().numpy()
    del K, L, Ks, v
    torch.cuda.empty_cache()
    return out


