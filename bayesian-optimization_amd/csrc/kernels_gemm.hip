// kernels_gemm.hip -- the small dense products around the factorisation that used to be library calls (rocBLAS dgemm /
// dgemv / dtrmm / dtrmv / ddot): polynomial trend bases (gpr.py:799-808 with a matrix F, trend.py:94-142), the REML trend
// terms (gpr.py:850-918), the small-batch posterior and the trend part of the one-point gradient.  libbogp links no BLAS.
//
//   k_gemm64     C (m x n) = alpha op(A) op(B) + beta C, column-major, any shape, any transposition: one 64 x 64 tile of C
//                per workgroup, both operands staged k-major into LDS with bounds-checked element loads (zero fill), 32-deep
//                k-blocks, v_mfma_f64_16x16x4_f64 from LDS (wave w: rows 16 w .., all 64 columns).  Deterministic: one
//                workgroup per tile walks k in order.  Built for correctness on odd shapes (p = 21 / 231 columns, m = 1
//                rows), not for the roofline -- the two trend products per candidate chunk that matter for throughput
//                run on k_mm128 (kernels_chol.hip, MM_GEN) whenever their shapes allow.
//   k_transpose_pad   out (ldo x cols_out, zero padded) = in^T for the column side of those products.
#include "bogp_device.h"
#include "bogp_internal.h"

namespace bogp {

namespace {
constexpr int GT = 64;        // tile edge
constexpr int GK = 32;        // k-block
constexpr int GP = 64 + 16;   // LDS pitch in doubles

typedef double d4g __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mfma16g(double a, double b, d4g& c) {
  asm("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

// tile[kk][x] = op(M)(x0 + x, k0 + kk), element (x, kk) of op(M) at M[x * sx + kk * sk]; zero outside (nx, nk).
// The thread -> element map follows the unit stride so that either orientation loads coalesced.
__device__ __forceinline__ void stage_generic(double* tile, const double* __restrict__ M, long sx, long sk, int x0, int k0, int nx,
                                              int nk, int tid) {
  if (sx == 1) {
#pragma unroll
    for (int p = 0; p < GT * GK / 256; ++p) {
      const int idx = tid + 256 * p;
      const int x = idx & (GT - 1), kk = idx >> 6;
      const bool in = x0 + x < nx && k0 + kk < nk;
      tile[kk * GP + x] = in ? M[(long)(x0 + x) + (long)(k0 + kk) * sk] : 0.0;
    }
  } else {
#pragma unroll
    for (int p = 0; p < GT * GK / 256; ++p) {
      const int idx = tid + 256 * p;
      const int kk = idx & (GK - 1), x = idx >> 5;
      const bool in = x0 + x < nx && k0 + kk < nk;
      tile[kk * GP + x] = in ? M[(long)(x0 + x) * sx + (long)(k0 + kk) * sk] : 0.0;
    }
  }
}
}  // namespace

__global__ __launch_bounds__(256) void k_gemm64(GemmArgs a) {
  __shared__ __attribute__((aligned(16))) double As[GK * GP];  // rows of C:    As[kk][i] = op(A)(i0 + i, k0 + kk)
  __shared__ __attribute__((aligned(16))) double Bs[GK * GP];  // columns of C: Bs[kk][j] = op(B)(k0 + kk, j0 + j)
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i0 = blockIdx.x * GT, j0 = blockIdx.y * GT;
  const int lk = lane >> 4, li = lane & 15;
  d4g c[4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) c[mi] = (d4g){0.0, 0.0, 0.0, 0.0};
  // the MFMAs are inline asm, invisible to the hazard recogniser: the moves that zero c must retire before the first one
  asm volatile("s_nop 7\n\ts_nop 7" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]));
  // op(A) lower triangular (tri = 1): columns k > row are zero -> stop at the tile's last row; upper (tri = 2): start at its first
  int kbeg = a.tri == 2 ? (i0 / GK) * GK : 0;
  int kend = a.tri == 1 ? min(a.k, i0 + GT) : a.k;
  const int nz = (int)gridDim.z, z = (int)blockIdx.z;
  if (nz > 1) {  // split K: slice z of the k-blocks of this tile
    const int nblk = (kend - kbeg + GK - 1) / GK, per = (nblk + nz - 1) / nz;
    const int b0 = kbeg + z * per * GK;
    kend = min(kend, b0 + per * GK);
    kbeg = b0;
  }
  for (int k0 = kbeg; k0 < kend; k0 += GK) {
    __syncthreads();  // the previous k-block has been consumed
    stage_generic(As, a.A, a.sai, a.sak, i0, k0, a.m, a.k, tid);
    stage_generic(Bs, a.B, a.sbj, a.sbk, j0, k0, a.n, a.k, tid);
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < GK / 4; ++ks) {
      const double rv = As[(4 * ks + lk) * GP + 16 * w + li];  // MFMA B operand: lane (k, row)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) mfma16g(Bs[(4 * ks + lk) * GP + 16 * mi + li], rv, c[mi]);  // A operand: lane (k, column)
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]));
  if (nz > 1) {
    // Split K (products with a small C and a long k, e.g. the p x p Gram matrices of the trend basis over k = N): every slice
    // parks its partial tile in scratch (write-through stores), the LAST slice to arrive at the tile's ticket adds all nz
    // partials in slice order -- the same sum whichever slice is last -- and applies alpha / beta.
    __shared__ bool s_last;
    const int tile = (int)(blockIdx.y * gridDim.x + blockIdx.x);
    double* mine = a.scratch + ((size_t)tile * nz + z) * (GT * GT);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int t = 0; t < 4; ++t) __hip_atomic_store(mine + (4 * mi + t) * 256 + tid, c[mi][t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (tid == 0) {
      const unsigned int ticket = __hip_atomic_fetch_add(a.tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = ticket == (unsigned int)nz - 1;
      if (s_last) __hip_atomic_store(a.tickets + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
    }
    __syncthreads();
    if (!s_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const double* all = a.scratch + (size_t)tile * nz * (GT * GT);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        // plain loads behind the acquire fence (they pipeline; a chain of atomic loads cost 130 us for 32 slices), added in slice order
        const double* pz = all + (4 * mi + t) * 256 + tid;
        double s = 0.0;
        int zz = 0;
        for (; zz + 8 <= nz; zz += 8) {
          double v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = pz[(size_t)(zz + u) * (GT * GT)];
#pragma unroll
          for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; zz < nz; ++zz) s += pz[(size_t)zz * (GT * GT)];
        c[mi][t] = s;
      }
  }
  // D[i][j] of tile mi: i = column 16 mi + 4 t + lk (component t), j = row 16 w + li
  const int row = i0 + 16 * w + li;
  if (row < a.m) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int col = j0 + 16 * mi + 4 * t + lk;
        if (col < a.n) {
          double* o = a.C + (long)row + (long)col * a.ldc;
          const double v = a.alpha * c[mi][t];
          *o = a.beta == 0.0 ? v : __builtin_fma(a.beta, *o, v);
        }
      }
  }
}

// C (m x n, ldc) = alpha op(A) (m x k) op(B) (k x n) + beta C; ta / tb != 0: the stored matrix is the transpose;
// tri = 1 / 2: op(A) is square lower / upper triangular with explicit zeros in the other triangle (only the k range shrinks)
hipError_t launch_gemm(int ta, int tb, int m, int n, int k, double alpha, const double* A, int lda, const double* B, int ldb,
                       double beta, double* C, int ldc, hipStream_t st, int tri, const GemmSplit* sp) {
  if (m <= 0 || n <= 0) return hipSuccess;
  GemmArgs a;
  a.A = A; a.B = B; a.C = C;
  a.m = m; a.n = n; a.k = k; a.ldc = ldc;
  a.sai = ta ? lda : 1; a.sak = ta ? 1 : lda;  // op(A)(i, kk)
  a.sbj = tb ? 1 : ldb; a.sbk = tb ? ldb : 1;  // op(B)(kk, j)
  a.alpha = alpha; a.beta = beta; a.tri = tri;
  a.scratch = nullptr; a.tickets = nullptr;
  const int gx = (m + GT - 1) / GT, gy = (n + GT - 1) / GT;
  // Split K when the tiles would leave most of the GPU idle over a long k: a handful of workgroups walking k = N alone is
  // ~2 us per k-block of pure latency (130 us at N = 2048 for ONE Gram matrix of the linear basis).
  int nz = 1;
  const int iters = (k + GK - 1) / GK;
  if (sp && sp->scratch && gx * gy <= sp->max_tiles && iters >= 8) {
    nz = min(32, iters / 2);
    while (nz > 1 && (size_t)gx * gy * nz * (GT * GT) > sp->cap) --nz;
    if (nz > 1) { a.scratch = sp->scratch; a.tickets = sp->tickets; }
  }
  hipLaunchKernelGGL(k_gemm64, dim3((unsigned)gx, (unsigned)gy, (unsigned)nz), 256, 0, st, a);
  return hipGetLastError();
}

// out[c + r * ldo] = in[r + c * ldi] for r < rows, c < cols; zero for c in [cols, ldo): the transposed, row-padded copy a
// column-side operand of k_mm128 needs (element (col, k) at out[col + k * ldo])
__global__ void k_transpose_pad(const double* __restrict__ in, int ldi, int rows, int cols, double* __restrict__ out, int ldo) {
  __shared__ double tl[32][33];
  const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
  for (int c = ty; c < 32; c += 8) {
    const int rr = r0 + tx, cc = c0 + c;
    tl[c][tx] = (rr < rows && cc < cols) ? in[(size_t)rr + (size_t)cc * ldi] : 0.0;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int rr = r0 + r, cc = c0 + tx;
    if (rr < rows && cc < ldo) out[(size_t)cc + (size_t)rr * ldo] = tl[tx][r];
  }
}
hipError_t launch_transpose_pad(const double* in, int ldi, int rows, int cols, double* out, int ldo, hipStream_t st) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_transpose_pad, dim3((unsigned)((rows + 31) / 32), (unsigned)((ldo + 31) / 32)), 256, 0, st, in, ldi, rows, cols, out, ldo);
  return hipGetLastError();
}

}  // namespace bogp
