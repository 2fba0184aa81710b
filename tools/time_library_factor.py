"""What the vendor libraries (through torch: hipSOLVER / rocBLAS / MAGMA) need for the factorisation steps of one
likelihood at N = 2048 / 8192 in float64 -- the yardstick for kernels_chol.hip."""
import time

import torch


def t(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    dev = torch.device("cuda", 0)
    for N in (2048, 8192):
        g = torch.Generator(device=dev).manual_seed(0)
        A = torch.randn(N, N, dtype=torch.float64, device=dev, generator=g)
        R = A @ A.T / N + torch.eye(N, dtype=torch.float64, device=dev)
        L = torch.linalg.cholesky(R)
        eye = torch.eye(N, dtype=torch.float64, device=dev)
        print("N=%d: cholesky %.2f ms, triangular inverse (solve_triangular vs I) %.2f ms, cholesky_inverse %.2f ms, syrk-like A@A.T %.2f ms"
              % (N, t(lambda: torch.linalg.cholesky(R)), t(lambda: torch.linalg.solve_triangular(L, eye, upper=False)),
                 t(lambda: torch.cholesky_inverse(L)), t(lambda: A @ A.T)))


if __name__ == "__main__":
    main()
