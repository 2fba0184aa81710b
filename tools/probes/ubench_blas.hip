// rocBLAS FP64 rates at the shapes the large-N fit path would hand to the library (MI355X): syrk with K = 256 / 512 on a
// growing trailing matrix, trmm against a triangular block, and plain gemm for comparison.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/ubench_blas.hip -lrocblas -o tools/probes/ubench_blas
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <cstdio>
#include <vector>
static double now_ms(hipEvent_t a, hipEvent_t b) { float ms = 0; hipEventElapsedTime(&ms, a, b); return ms; }
int main() {
  const int N = 8192;
  double *A, *B, *C;
  hipMalloc(&A, (size_t)N * N * 8); hipMalloc(&B, (size_t)N * N * 8); hipMalloc(&C, (size_t)N * N * 8);
  std::vector<double> h((size_t)N * N);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5;
  hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice); hipMemcpy(B, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  hipMemset(C, 0, (size_t)N * N * 8);
  rocblas_handle hd; rocblas_create_handle(&hd);
  rocblas_set_atomics_mode(hd, rocblas_atomics_not_allowed);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double one = 1.0, mone = -1.0, zero = 0.0;
  for (int k : {64, 256, 512}) {
    for (int n : {2048, 4096, 7936}) {
      rocblas_dsyrk(hd, rocblas_fill_lower, rocblas_operation_none, n, k, &mone, A, N, &one, C, N);
      hipEventRecord(e0); 
      for (int r = 0; r < 3; ++r) rocblas_dsyrk(hd, rocblas_fill_lower, rocblas_operation_none, n, k, &mone, A, N, &one, C, N);
      hipEventRecord(e1); hipEventSynchronize(e1);
      double ms = now_ms(e0, e1) / 3;
      printf("dsyrk  lower N   n=%5d k=%4d: %.3f ms  %.1f TF/s (n^2 k flops)\n", n, k, ms, (double)n * n * k / ms / 1e9);
      rocblas_dsyrk(hd, rocblas_fill_lower, rocblas_operation_transpose, n, k, &one, A, N, &one, C, N);
      hipEventRecord(e0);
      for (int r = 0; r < 3; ++r) rocblas_dsyrk(hd, rocblas_fill_lower, rocblas_operation_transpose, n, k, &one, A, N, &one, C, N);
      hipEventRecord(e1); hipEventSynchronize(e1);
      ms = now_ms(e0, e1) / 3;
      printf("dsyrk  lower T   n=%5d k=%4d: %.3f ms  %.1f TF/s\n", n, k, ms, (double)n * n * k / ms / 1e9);
      rocblas_dgemm(hd, rocblas_operation_none, rocblas_operation_transpose, n, n, k, &mone, A, N, A, N, &one, C, N);
      hipEventRecord(e0);
      for (int r = 0; r < 3; ++r) rocblas_dgemm(hd, rocblas_operation_none, rocblas_operation_transpose, n, n, k, &mone, A, N, A, N, &one, C, N);
      hipEventRecord(e1); hipEventSynchronize(e1);
      ms = now_ms(e0, e1) / 3;
      printf("dgemm  NT        n=%5d k=%4d: %.3f ms  %.1f TF/s (2 n^2 k flops)\n", n, k, ms, 2.0 * n * n * k / ms / 1e9);
    }
  }
  for (int b : {256, 512, 1024, 2048, 4096}) {
    rocblas_dtrmm(hd, rocblas_side_right, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit, b, b, &one, A, N, B, N, C, N);
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) rocblas_dtrmm(hd, rocblas_side_right, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit, b, b, &one, A, N, B, N, C, N);
    hipEventRecord(e1); hipEventSynchronize(e1);
    double ms = now_ms(e0, e1) / 3;
    printf("dtrmm  right low b=%5d: %.3f ms  %.1f TF/s (b^3 flops)\n", b, ms, (double)b * b * b / ms / 1e9);
    rocblas_dtrmm(hd, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit, b, b, &mone, A, N, B, N, C, N);
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) rocblas_dtrmm(hd, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit, b, b, &mone, A, N, B, N, C, N);
    hipEventRecord(e1); hipEventSynchronize(e1);
    ms = now_ms(e0, e1) / 3;
    printf("dtrmm  left  low b=%5d: %.3f ms  %.1f TF/s\n", b, ms, (double)b * b * b / ms / 1e9);
    rocblas_dgemm(hd, rocblas_operation_none, rocblas_operation_none, b, b, b, &one, A, N, B, N, &zero, C, N);
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) rocblas_dgemm(hd, rocblas_operation_none, rocblas_operation_none, b, b, b, &one, A, N, B, N, &zero, C, N);
    hipEventRecord(e1); hipEventSynchronize(e1);
    ms = now_ms(e0, e1) / 3;
    printf("dgemm  NN        b=%5d: %.3f ms  %.1f TF/s (2 b^3 flops)\n", b, ms, 2.0 * b * b * b / ms / 1e9);
  }
  return 0;
}
