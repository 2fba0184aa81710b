"""What the library GEMM reaches on the same product k_contract performs (r chunk x L^-1): torch.mm in float64 goes to
rocBLAS / hipBLASLt, whose Tensile kernels use v_mfma_f64_16x16x4 (half the FP64 rate of v_mfma_f64_4x4x4 on gfx950,
profiles/r01_ubench_f64.txt) and cannot skip the upper triangle of L^-1."""
import time, torch
M, N = 65536, 2048
A = torch.rand(M, N, dtype=torch.float64, device="cuda")
V = torch.tril(torch.rand(N, N, dtype=torch.float64, device="cuda"))
for _ in range(3):
    C = A @ V.T
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    C = A @ V.T
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / 10
dense = 2.0 * M * N * N
print("torch.mm f64 (%d x %d) @ (%d x %d): %.2f ms = %.1f TF/s dense-equivalent, %.1f TF/s of the triangular (useful) flops"
      % (M, N, N, N, t * 1e3, dense / t / 1e12, dense / 2 / t / 1e12))
print("k_contract on the same chunk: 4.09 ms = 64.7 TF/s of the triangular flops (bench.py)")
