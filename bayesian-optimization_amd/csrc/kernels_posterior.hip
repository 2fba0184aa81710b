// kernels_posterior.hip -- the candidates/sec hot path of libbogp on gfx950 (MI355X).
//
// Replaces GaussianProcess.predict (surrogate/gaussian_process/gpr.py:486-510):
//     dx = |X*_i - X_j|                  (M*N, d) temporary          -> never materialised
//     r  = corr(theta, dx)               (M, N)                      -> k_corr_chunk   (FP64 VALU)
//     mu = mean + r @ gamma                                          -> k_corr_chunk   (fused)
//     rt = solve_triangular(L, r.T)      N^2 flops per candidate     -> k_contract     (FP64 MFMA)
//     1 - sum(rt^2) [+ sum(u^2)]                                     -> k_contract epilogue (+ acquisition kernel)
//
// Design constants come from measurements on the target (tools/probes/ubench_f64*.hip, tools/probes/ubench_mfma16.hip; profiles/r01_ubench_f64.txt,
// r01_ubench_mfma16.txt, r01_ubench_mfma_stream.txt):
//   * v_mfma_f64_4x4x4_4b_f64 issues every 16 cycles (512 flop) = 32 flop/clk/SIMD = the 78.6 TF/s FP64 peak (kernel B below);
//     v_mfma_f64_16x16x4_f64 issues every 64 cycles (2048 flop) = the same rate -- but only with its accumulator in
//     architectural VGPRs (with AGPR accumulators, where the builtin puts them: ~130 cycles).  Kernel B' below, the DEFAULT,
//     is built on it: one A and one B register per 2048 flop instead of four rotated A registers.
//   * FP64 VALU work does not hide beside FP64 MFMA (same DP pipe: times add), so the kernel-matrix
//     producer must run ONCE per (candidate, training point) -- it is split into its own kernel and r is
//     staged through HBM/L2 in candidate chunks instead of being recomputed per column tile.
//   * lane layout of v_mfma_f64_4x4x4_4b_f64 (tools/probes/probe_mfma_layout.hip): A lane = 16k+4b+i,
//     B lane = 16k+4b+j, D lane = 16i+4b+j  (b = block 0..3).
//
// Triangular contraction.  With V = L^-1 (lower triangular, packed once per fit into B-fragment order),
// rt = V r and sum(rt^2) = sum_j (sum_{n<=j} V[j][n] r[n])^2.  A workgroup owns 64 candidates x 256 columns j
// (4 waves x 4 sixteen-wide column tiles, dealt so that every wave carries the same share of the diagonal) and walks n in
// blocks of 32; 16x16 blocks above the diagonal are skipped (executed flops ~= N^2 (1 + 16/N) per candidate).
// A 16x16 output tile is 4 instructions: instruction t pairs A row-block (b+t)%4 with B column-block b, so ONE
// B register and four rotated A reads (LDS) feed 2048 flop.
#include <algorithm>
#include <cstdlib>

#include "bogp_device.h"
#include "bogp_internal.h"

namespace bogp {

// ---------------------------------------------------------------------------------------------------
// Kernel A: correlation chunk  rT[n][m] = corr(theta, |x*_m - x_n|), + partial mean / trend dot products
//   grid (Mc/64, S): 64 candidates x one slice of the training set per workgroup; 256 threads.
//   thread (m = tid&63, g = tid>>6) produces r for 8 consecutive n per 32-block; the training rows are
//   wave-uniform => scalar loads from the theta-scaled transposed copy XthT[d][Np].
// ---------------------------------------------------------------------------------------------------
struct CorrDims {
  int64_t M, m0, Mc;
  int d, Np, nblk_per_split;
  int wld;  // PV > 0: row pitch of Wrow
};
// pointers are separate __restrict__ kernel arguments (not struct members) so that the wave-uniform reads of
// XthT / gamma / wvec are provably read-only and become scalar (SMEM) loads
// PV > 0 (polynomial trend with p <= PV columns under universal kriging, r04): the producer also forms the slice sums of
// T = W^T r, W = L^-T Ft (gpr.py:496-498: Ft^T L^-1 r).  Every 32-row block of r is parked in LDS on its way out and contracted there
// against the rows Wrow[n][0 .. PV) on the matrix cores -- wave g: candidates 16 g .. 16 g + 15 x PV / 16 column tiles,
// v_mfma_f64_16x16x4_f64 (the columns p .. PV of Wrow are zero padding) -- 16 MFMAs a block and wave beside ~450 FP64 VALU
// instructions.  The separate tile product it replaces re-read the whole 1-GiB correlation chunk for 21 live columns of a 128-column
// tile (+13 % per sweep for a linear trend at C3).  A first version with PV scalar-loaded FMAs a pair doubled the producer's time
// (SGPR-starved: 8 x 24 wave-uniform doubles a block): profiles/r04_trend_timing.txt.
template <int KERNEL, int PV>
__global__ __launch_bounds__(256) void k_corr_chunk(const double* __restrict__ Xs, const double* __restrict__ sqrt_theta,
                                                    const double* __restrict__ XthT, const double* __restrict__ gamma,
                                                    const double* __restrict__ wvec, double* __restrict__ rT,
                                                    double* __restrict__ mu_part, double* __restrict__ w_part,
                                                    const double* __restrict__ Wrow, double* __restrict__ t_part, CorrDims a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* xs = smem;  // [d][64] theta-scaled candidate tile, k-major
  const int tid = threadIdx.x;
  const int m = tid & 63;
  const int g = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t mc0 = (int64_t)blockIdx.x * 64;  // offset inside the chunk
  const int64_t mg0 = a.m0 + mc0;                // global candidate index of row 0
  const int d = a.d;

  for (int idx = tid; idx < 64 * d; idx += 256) {
    const int row = idx / d, k = idx - row * d;
    const int64_t gm = mg0 + row;
    const double v = gm < a.M ? Xs[gm * d + k] : 0.0;
    xs[k * 64 + row] = v * sqrt_theta[k];
  }
  __syncthreads();

  const double pexp = kernel_exponent<KERNEL>(sqrt_theta, d);
  const int nb0 = blockIdx.y * a.nblk_per_split * 32;
  const int nb1 = min(a.Np, nb0 + a.nblk_per_split * 32);
  double mu = 0.0, wd = 0.0;
  constexpr int NTT = PV > 0 ? PV / 16 : 1;
  typedef double d4t __attribute__((ext_vector_type(4)));
  d4t tacc[NTT];
#pragma unroll
  for (int c = 0; c < NTT; ++c) tacc[c] = (d4t){0.0, 0.0, 0.0, 0.0};
  double* rtile = smem + max(64 * d, 512);  // PV > 0: [32][64] the block of r being contracted (behind the candidate tile / the reduction arrays)
  const int lk = m >> 4, li = m & 15;
  for (int nb = nb0; nb < nb1; nb += 32) {
    const int n0 = nb + g * 8;
    double acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = dist_init<KERNEL>();
#pragma unroll 2
    for (int k = 0; k < d; ++k) {
      const double xk = xs[k * 64 + m];
      const double* __restrict__ xr = XthT + (size_t)k * a.Np + n0;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc[i] = dist_accumulate<KERNEL>(xk - xr[i], acc[i], pexp);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const double r = corr_profile<KERNEL>(acc[i], pexp);
      rT[(size_t)(n0 + i) * a.Mc + mc0 + m] = r;
      mu = __builtin_fma(r, gamma[n0 + i], mu);
      wd = __builtin_fma(r, wvec[n0 + i], wd);
      if (PV > 0) rtile[(g * 8 + i) * 64 + m] = r;
    }
    if (PV > 0) {
      __syncthreads();
      // A: lane (k, i) = r[candidate 16 g + i][n = nb + 4 ks + k];  B: lane (k, j) = W[n][16 t + j]
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const double av = rtile[(4 * ks + lk) * 64 + 16 * g + li];
        const double* __restrict__ wr = Wrow + (size_t)(nb + 4 * ks + lk) * a.wld + li;
#pragma unroll
        for (int t = 0; t < NTT; ++t) {
          const double bv = wr[16 * t];
          asm("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(tacc[t]) : "v"(av), "v"(bv));
        }
      }
      __syncthreads();  // the block is overwritten by the next trip
    }
  }
  // reduce the 4 n-groups (fixed order) -> partial sums of this training-set slice
  __syncthreads();
  double* red = smem;  // reuse: [2][4][64]
  red[g * 64 + m] = mu;
  red[256 + g * 64 + m] = wd;
  __syncthreads();
  if (tid < 64) {
    const double s0 = ((red[tid] + red[64 + tid]) + red[128 + tid]) + red[192 + tid];
    const double s1 = ((red[256 + tid] + red[320 + tid]) + red[384 + tid]) + red[448 + tid];
    mu_part[(size_t)blockIdx.y * a.Mc + mc0 + tid] = s0;
    w_part[(size_t)blockIdx.y * a.Mc + mc0 + tid] = s1;
  }
  if (PV > 0) {
    // the trend tiles: D[i][j] sits in lane 16 (i % 4) + j, component i / 4 (i = candidate in the fragment, j = column in the tile)
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int t = 0; t < NTT; ++t) asm volatile("" : "+v"(tacc[t]));
#pragma unroll
    for (int t = 0; t < NTT; ++t)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        t_part[((size_t)blockIdx.y * PV + 16 * t + li) * a.Mc + mc0 + 16 * g + 4 * c + lk] = tacc[t][c];
  }
}

// ---------------------------------------------------------------------------------------------------
// Kernel A' (r05, the DEFAULT for the squared-distance kernels without a fused polynomial trend): the same chunk with the
// weighted squared distance formed on the matrix cores.
//   s2(m, n) = |a_m|^2 + |b_n|^2 - 2 a_m . b_n          a = sqrt(theta) x*_m,  b = sqrt(theta) x_n
// The (sub, fma) pair per dimension of kernel A is 2 d FP64 VALU instructions per pair -- half of the producer at C3 -- and an FP64 MFMA
// moves exactly as many flops per clock as the FP64 VALU, so the cross term a . b as a 16 x 16 x 4 matrix product costs d
// FMA-equivalents per pair instead of 2 d, plus three for the assembly of s2.  The price is cancellation: the cross-term form carries an
// ABSOLUTE error of ~sqrt(d) eps (|a|^2 + |b|^2) / 2 where the difference form carries a RELATIVE error of ~d eps.  Hence the guard:
//   a pair with s2 * 64 < |a|^2 + |b|^2 (a candidate within an eighth of the typical distance of a training point; never, for
//   space-filling candidates in d >= 3; routinely, once a BO run concentrates around the incumbent) is recomputed in the difference form,
//   with kernel A's very operations in kernel A's order -- so a candidate ON a training point still gives s2 = 0, r = 1 and a clipped
//   MSE of exactly 0 (the guards of EI / MGFI, acquisition_fun.py:162-164, 274-275, see tests G7), and every pair has s2 to <= ~64 sqrt(d) eps
//   relative.  The branch is wave-uniform (ballot); a wave pays the slow path only for its flagged lanes' values.
// Layout: workgroup = 64 candidates x one slice of the training set (as kernel A); wave g owns training rows 16 g .. 16 g + 15 of every
// 64-row block and all 64 candidates: four 16 x 16 output tiles D[i = row][j = candidate] sharing one A fragment (training rows, a
// 128-byte run of XthT per dimension: L1 / L2 resident) against four B fragments (candidate tile, LDS, k-major: conflict-free).
// v_mfma_f64_16x16x4_f64: A lane 16 k + i, B lane 16 k + j, D[i][j] in lane 16 (i % 4) + j, component i / 4 (probe_mfma_layout).
// ---------------------------------------------------------------------------------------------------
template <int KERNEL>
// (four waves per SIMD: 128 VGPRs, a dozen spilled in the prologue -- 5.40 against 5.62 ms at three waves and 155 VGPRs, C3; profiles/r05_corr_mfma_ab.txt)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_corr_mfma(const double* __restrict__ Xs, const double* __restrict__ sqrt_theta,
                                                   const double* __restrict__ XthT, const double* __restrict__ xnorm,
                                                   const double* __restrict__ gamma, const double* __restrict__ wvec,
                                                   double* __restrict__ rT, double* __restrict__ mu_part, double* __restrict__ w_part,
                                                   CorrDims a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int d = a.d;
  const int KS = (d + 3) >> 2;       // k-steps of 4 dimensions; rows d .. 4 KS - 1 of the tile are zero
  double* xs = smem;                 // [4 KS][64] theta-scaled candidate tile, k-major
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int g = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t mc0 = (int64_t)blockIdx.x * 64;
  const int64_t mg0 = a.m0 + mc0;
  for (int idx = tid; idx < 64 * d; idx += 256) {
    const int row = idx / d, k = idx - row * d;
    const int64_t gm = mg0 + row;
    const double v = gm < a.M ? Xs[gm * d + k] : 0.0;
    xs[k * 64 + row] = v * sqrt_theta[k];
  }
  for (int idx = tid + 64 * d; idx < 64 * 4 * KS; idx += 256) xs[idx] = 0.0;
  __syncthreads();
  const double pexp = kernel_exponent<KERNEL>(sqrt_theta, d);
  const int li = lane & 15, lk = lane >> 4;
  double na[4];  // |a_m|^2 of this lane's four candidates m = 16 t + li, summed in dimension order (as k_scale_transpose sums |b_n|^2)
#pragma unroll
  for (int t = 0; t < 4; ++t) na[t] = 0.0;
  for (int k = 0; k < d; ++k) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const double v = xs[k * 64 + 16 * t + li];
      na[t] = __builtin_fma(v, v, na[t]);
    }
  }
  const int nb0 = blockIdx.y * a.nblk_per_split * 32;
  const int nb1 = min(a.Np, nb0 + a.nblk_per_split * 32);
  typedef double d4t __attribute__((ext_vector_type(4)));
  double mu[4] = {0.0, 0.0, 0.0, 0.0}, wd[4] = {0.0, 0.0, 0.0, 0.0};
  // (r06, from k_contract16d: every global address of the loop is a wave-uniform base + a lane-constant 32-bit offset, kept opaque inside the loop so that
  // hipcc selects the `v_offset, s[base]` form -- hoisted, each access cost a v_lshl_add_u64 on the FP64 pipe that the MFMAs and the profile's arithmetic share)
  unsigned offA = (unsigned)(((size_t)lk * a.Np + li) * sizeof(double));   // XthT: row lk of a k-step, training point li of the block
  unsigned offN = (unsigned)(lk * sizeof(double));                          // xnorm / gamma / w: training point 4 c + lk
  unsigned offR = (unsigned)(((size_t)lk * a.Mc + li) * sizeof(double));   // rT: row 4 c + lk of the block, candidate 16 t + li
  const char* const xthB = reinterpret_cast<const char*>(XthT);
  const char* const xnB = reinterpret_cast<const char*>(xnorm);
  const char* const gaB = reinterpret_cast<const char*>(gamma);
  const char* const wvB = reinterpret_cast<const char*>(wvec);
  char* const rtB = reinterpret_cast<char*>(rT + mc0);
  for (int n0 = nb0 + 16 * g; n0 < nb1; n0 += 64) {
#define BOGP_OPQ(o_) asm("" : "+v"(o_))  /* (beside EVERY access: the zero-extension must sit in the access's own basic block to be matched) */
    d4t acc[4];
    // a_m . b_n over the dimensions, four at a time
    {
      const char* __restrict__ apb = xthB + (size_t)n0 * sizeof(double);
      const double* bp = xs + lk * 64 + li;
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = (d4t){0.0, 0.0, 0.0, 0.0};
      constexpr int KB = 8;  // A fragments (128-byte runs of XthT: L2 round trips) requested together: 5.5 -> 5.1 ms at C3, 17.5 -> 14.3 ms at C5
      for (int ks0 = 0; ks0 < KS; ks0 += KB) {
        double av[KB];
#pragma unroll
        for (int u = 0; u < KB; ++u)
        {
          BOGP_OPQ(offA);
          av[u] = (4 * (ks0 + u) + lk) < d ? *reinterpret_cast<const double*>(apb + (size_t)4 * (ks0 + u) * a.Np * sizeof(double) + offA) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < KB; ++u) {
          if (ks0 + u < KS) {
            const int ks = ks0 + u;
            const double b0 = bp[ks * 256], b1 = bp[ks * 256 + 16], b2 = bp[ks * 256 + 32], b3 = bp[ks * 256 + 48];
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(av[u]), "v"(b0));
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[1]) : "v"(av[u]), "v"(b1));
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[2]) : "v"(av[u]), "v"(b2));
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[3]) : "v"(av[u]), "v"(b3));
          }
        }
      }
    }
    // this lane's four training rows n0 + 4 c + lk: norm, gamma, w (the loads fly while the matrix pipe drains)
    double nbv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      BOGP_OPQ(offN);
      nbv[c] = *reinterpret_cast<const double*>(xnB + (size_t)(n0 + 4 * c) * sizeof(double) + offN);
    }
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
#pragma unroll
    for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(acc[t]));
    double s2[4][4];
    bool near = false;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const double nab = na[t] + nbv[c];
        const double v = __builtin_fma(-2.0, acc[t][c], nab);
        s2[t][c] = v;
        near |= v * 64.0 < nab;
      }
    if (__builtin_expect(__ballot(near) != 0ull, 0)) {
      // difference form, kernel A's operations in kernel A's order, for all 16 values of the lane in ONE loop over the dimensions (compact code:
      // sixteen unrolled per-value loops cost 40 VGPRs and an occupancy step); only the FLAGGED values take it, so that r(x*_m, x_n) never
      // depends on which other pairs share the wave
      const double* __restrict__ xr = XthT + n0 + lk;
#pragma unroll
      for (int th = 0; th < 4; th += 2) {  // two candidate tiles at a time: eight accumulators
        double e[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int c = 0; c < 4; ++c) e[t][c] = 0.0;
        const double* xc = xs + 16 * th + li;
#pragma unroll 1
        for (int k = 0; k < d; ++k) {
          double xcv[2], xrv[4];
#pragma unroll
          for (int t = 0; t < 2; ++t) xcv[t] = xc[k * 64 + 16 * t];
#pragma unroll
          for (int c = 0; c < 4; ++c) xrv[c] = xr[(size_t)k * a.Np + 4 * c];
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const double df = xcv[t] - xrv[c];
              e[t][c] = __builtin_fma(df, df, e[t][c]);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (s2[th + t][c] * 64.0 < na[th + t] + nbv[c]) s2[th + t][c] = e[t][c];
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      char* __restrict__ rrow = rtB + (size_t)(n0 + 4 * c) * a.Mc * sizeof(double);
      BOGP_OPQ(offN);
      const double gv = *reinterpret_cast<const double*>(gaB + (size_t)(n0 + 4 * c) * sizeof(double) + offN);
      const double wv = *reinterpret_cast<const double*>(wvB + (size_t)(n0 + 4 * c) * sizeof(double) + offN);
      BOGP_OPQ(offR);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const double r = corr_profile<KERNEL>(s2[t][c], pexp);
        *reinterpret_cast<double*>(rrow + 16 * t * sizeof(double) + offR) = r;
        mu[t] = __builtin_fma(r, gv, mu[t]);
        wd[t] = __builtin_fma(r, wv, wd[t]);
      }
    }
  }
  // reduce over the 4 waves x 4 row groups of a lane (fixed order) -> partial sums of this slice
  __syncthreads();
  double* red = smem;  // [2][16][64]
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    red[(4 * g + lk) * 64 + 16 * t + li] = mu[t];
    red[1024 + (4 * g + lk) * 64 + 16 * t + li] = wd[t];
  }
  __syncthreads();
  if (tid < 64) {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      s0 += red[j * 64 + tid];
      s1 += red[1024 + j * 64 + tid];
    }
    mu_part[(size_t)blockIdx.y * a.Mc + mc0 + tid] = s0;
    w_part[(size_t)blockIdx.y * a.Mc + mc0 + tid] = s1;
  }
}

#undef BOGP_OPQ
// ---------------------------------------------------------------------------------------------------
// Kernel B: triangular contraction  ss_part[jg][m] = sum_{j in group jg} (sum_{n<=j} V[j][n] r[m][n])^2
// ---------------------------------------------------------------------------------------------------
constexpr int MR = 4;                // 16-row fragments per wave  (64 candidates)
constexpr int NWJ = 4;               // waves per workgroup (along j)
// NR = 16-column fragments per wave (template parameter): NR = 4 -> 256 columns per workgroup, 128 accumulator
// VGPRs, one workgroup per CU (512-register budget); NR = 2 -> 128 columns, two workgroups per CU.
constexpr int KB = 32;               // training points per staged block
constexpr int PITCH = 64 + 16;       // LDS row pitch in doubles: 640 B == 128 (mod 256) -> conflict-free A reads

// (r04: the first contraction kernel, k_contract on v_mfma_f64_4x4x4_4b_f64 with AGPR accumulators -- opt-in since r01 through
// BOGP_CONTRACT_MFMA=4x4, 20 % slower than k_contract16 -- was removed: tests/test_isa_lint.py found AGPR spills inside its main loop and an
// accumulator read ahead of its drain in the ROCm 7.2 build.  Its description stays in DESIGN.md section 5.2; the code is in git 26333e7.)

// ---------------------------------------------------------------------------------------------------
// Kernel B': the same contraction on v_mfma_f64_16x16x4_f64.
//
// The first probe of this instruction (builtin, accumulators where the compiler put them: AGPRs) measured 130-150
// cycles = half the FP64 rate, which is why kernel B was built on the 4x4x4 form.  With the accumulator tied in place in
// ARCHITECTURAL VGPRs it issues every 64.0 cycles = 77.5 TF/s, at 1-4 waves per SIMD (tools/probes/ubench_mfma16.hip,
// profiles/r01_ubench_mfma16.txt; AGPR accumulators: 130 cycles) -- and it needs ONE A register and ONE B register per
// 2048 flop where the 4x4x4 form needs four rotated A registers: a quarter of the LDS reads, no rotation, a quarter
// of the MFMA issue slots.  Lane layout (tools/probes/probe_mfma_layout.hip): A lane = 16 k + i, B lane = 16 k + j (the SAME
// order k_pack_V already produces), D[i][j] in lane 16 (i % 4) + j, register i / 4.
// Tiling, staging, guards and the epilogue's summation order are those of kernel B.
// ---------------------------------------------------------------------------------------------------
typedef double d4 __attribute__((ext_vector_type(4)));
#ifdef CONTRACT_TRACE
// (profiling builds only, `make EXTRA=-DCONTRACT_TRACE`: shader-clock stamps of every wave of every workgroup of k_contract16<4>, parked in LDS and
// written out at the end -- tools/contract_trace.py turns them into the MFMA-idle attribution of profiles/r06_contract_att.txt.  This image has
// no thread-trace decoder and the driver offers no PC sampling (same file), so the kernel keeps its own time.  The MFMAs are volatile here so
// that they stay on their side of a stamp.)
constexpr int TR_BLK = 64;                // blocks with stamps (N <= 2048)
constexpr int TR_NST = 8 + 8 * TR_BLK;    // words per wave: 8 header + 8 per block
__device__ __forceinline__ unsigned long long tr_now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t));
  return t;
}
#define BOGP_TR(p, i) ((p)[(i)] = tr_now())
#define BOGP_TR_PARAM , unsigned long long* trb
#define BOGP_TR_ARG(x) , (x)
__device__ __forceinline__ void mfma16_acc(double a, double b, d4& c) {
  asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
#else
#define BOGP_TR(p, i) ((void)0)
#define BOGP_TR_PARAM
#define BOGP_TR_ARG(x)
__device__ __forceinline__ void mfma16_acc(double a, double b, d4& c) {
  asm("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
#endif
// 16 passes: the result of the last MFMA must not be read by the VALU before it has left the pipe
#define BOGP_MFMA16_DRAIN() asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory")

template <int NR, bool GUARDED>
__device__ __forceinline__ void contract_block16(const double* __restrict__ tile, const double2* __restrict__ vp,
                                                 const size_t (&boff)[NR], const int (&jt)[NR], int aoff, int kb, int kp_last,
                                                 double2 (&bq)[4][NR], d4 (&acc)[MR][NR] BOGP_TR_PARAM) {
  // A fragments are read one k-step AHEAD of the MFMAs that use them (two register sets): with only four LDS reads per
  // sixteen MFMAs their latency would otherwise sit in front of every k-step
  double af[2][MR];
#pragma unroll
  for (int mi = 0; mi < MR; ++mi) af[0][mi] = tile[aoff + 16 * mi];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int kp = kb * 4 + s;
    const int kb16 = kp >> 1;
    if (s > 0) BOGP_TR(trb, s);
    {  // B fragments are requested TWO k-pairs ahead (four slots = the four k-pairs of a block; clamped at the end)
#ifdef CONTRACT_AB_FIXB  // every B fragment from the tile's first k-pair (L2 / L1-hot)
      const int kpn = min(0, kp_last);
#else
      const int kpn = min(kp + 2, kp_last);
#endif
#pragma unroll
      for (int ni = 0; ni < NR; ++ni) bq[(s + 2) & 3][ni] = vp[boff[ni] + (size_t)kpn * 64];
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int sub = 2 * s + h;
      if (sub < 7) {
        const double* trow = tile + (4 * (sub + 1)) * PITCH;
#pragma unroll
        for (int mi = 0; mi < MR; ++mi) af[(sub + 1) & 1][mi] = trow[aoff + 16 * mi];
      }
#pragma unroll
      for (int ni = 0; ni < NR; ++ni) {
        if (!GUARDED || kb16 <= jt[ni]) {
          const double bv = h == 0 ? bq[s][ni].x : bq[s][ni].y;
#pragma unroll
          for (int mi = 0; mi < MR; ++mi) mfma16_acc(af[sub & 1][mi], bv, acc[mi][ni]);
        }
      }
    }
  }
  BOGP_TR(trb, 4);
}

// NCP != 0 (16 / 32 / 64): the batched one-point path (kernels_point.hip, k_point_rhs_T).  The 64 "candidates" of a workgroup are
// then the right-hand-side columns [r | dr/dx_1 .. dr/dx_d | 0 ..] of 64 / NCP points, NCP columns each, and what a point needs
// from C = V rhs is not |C_c|^2 but the cross products  sum_j C_0[j] C_c[j]  with its FIRST column ((V^T V r) . dr/dx_k =
// (V r) . (V dr/dx_k): one pass over V serves the posterior AND its d input-derivatives, gpr.py:537-576).  Only the epilogue
// differs: the first row of every NCP-row group (fragment head, register 0, lane quarter 0) is broadcast to the other quarters
// and multiplies instead of the square; the sums land as the row-block records k_point_finish reads.
template <int NR, int NCP = 0>
#ifdef CONTRACT_TRACE
__global__ __launch_bounds__(256, NR == 2 ? 3 : 2) void k_contract16(ContractArgs a, unsigned long long* trace_out) {
#else
__global__ __launch_bounds__(256, NR == 2 ? 3 : 2) void k_contract16(ContractArgs a) {
#endif
  constexpr int JT16 = NWJ * NR;
  __shared__ __attribute__((aligned(16))) double lds[2 * KB * PITCH];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef CONTRACT_TRACE
  __shared__ unsigned long long trc_all[NWJ][TR_NST];
  unsigned long long* trc = trc_all[w];
  {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    trc[0] = (unsigned long long)hwid | ((unsigned long long)xcc << 32);
  }
  BOGP_TR(trc, 1);
#define BOGP_TRB(kb_) ((kb_) < TR_BLK ? trc + 8 + 8 * (kb_) : trc + 8)
#else
#define BOGP_TRB(kb_) 0
#endif
  const int nMt = a.nMt;
  // Column-group-major order, heaviest group first: concurrent workgroups then walk the SAME 256-column panel of V, which
  // stays hot in every XCD's L2 (the B fragments feed the MFMAs straight from global loads).  Measured alternative (r02,
  // tools/ab/ab_contract_order.sh): the nJ groups of one candidate tile back to back on one XCD -- r tiles shared in L2, but the
  // B reads then span all of V (16 MB against 4 MB of L2): 61.5 -> 85.4 ms per step.  r is re-read (nJ + 1) / 2 times instead.
  const int jg = a.nJ - 1 - (int)(blockIdx.x / nMt);
  const int mt = blockIdx.x % nMt;
  const int64_t mc0 = (int64_t)mt * 64;
  const int NJ16 = a.NJ16, NKP = a.NKP;
  const int kmax16 = min((jg + 1) * JT16, NJ16);
#ifdef CONTRACT_AB_NOZONE  // the 256-row diagonal zone is skipped (column group 0 then runs ONE block): what the zone costs = product - this
  const int nkb = max(jg * (JT16 / 2), 1);
#else
  const int nkb = kmax16 >> 1;
#endif
  const int nkb_full = jg * (JT16 / 2);
  const int kp_last = 2 * kmax16 - 1;

  int jt[NR];
  size_t boff[NR];
  bool valid[NR];
#pragma unroll
  for (int ni = 0; ni < NR; ++ni) {
    // serpentine tile assignment {w, 7-w, 8+w, 15-w}: in the diagonal zone tile t carries t+1 sixteen-row groups, so every
    // wave gets the same 34 of them (the plain interleave w + 4 ni gives wave 3 640 MFMAs against wave 0's 448)
    const int j = jg * JT16 + ((ni & 1) ? NWJ * (ni + 1) - 1 - w : NWJ * ni + w);
    valid[ni] = j < NJ16;
    jt[ni] = valid[ni] ? j : -1;
    boff[ni] = (size_t)min(j, NJ16 - 1) * NKP * 64;
  }

  d4 acc[MR][NR];
#pragma unroll
  for (int mi = 0; mi < MR; ++mi)
#pragma unroll
    for (int ni = 0; ni < NR; ++ni) acc[mi][ni] = (d4){0.0, 0.0, 0.0, 0.0};

  const int srow = tid >> 5;
  const int scol = (tid & 31) * 2;
  const double* __restrict__ rbase = a.rT + mc0 + scol;
  const size_t Mc = (size_t)a.Mc;
  // Two register sets for the staged r tiles: tile kb + 2 is requested while tile kb is contracted and tile kb + 1 (requested one
  // block earlier) is written to LDS at the END of the block -- every load has TWO blocks of MFMA work to arrive.  With one
  // set (one block of cover) the half-empty blocks of the diagonal zone no longer hid the ~2 us round trip to HBM.
  double2 sa0, sa1, sa2, sa3, sb0, sb1, sb2, sb3;
#define BOGP_STAGE_LOAD(S, kb_)                                                                      \
  do {                                                                                               \
    const double* p_ = rbase + (size_t)(BOGP_STAGE_ROW(kb_)*KB + srow) * Mc;                                       \
    S##0 = *reinterpret_cast<const double2*>(p_);                                                    \
    S##1 = *reinterpret_cast<const double2*>(p_ + 8 * Mc);                                           \
    S##2 = *reinterpret_cast<const double2*>(p_ + 16 * Mc);                                          \
    S##3 = *reinterpret_cast<const double2*>(p_ + 24 * Mc);                                          \
  } while (0)
#define BOGP_STAGE_STORE(S, buf_)                                                                    \
  do {                                                                                               \
    double* q_ = &lds[(buf_)*KB * PITCH + srow * PITCH + scol];                                      \
    *reinterpret_cast<double2*>(q_) = S##0;                                                          \
    *reinterpret_cast<double2*>(q_ + 8 * PITCH) = S##1;                                              \
    *reinterpret_cast<double2*>(q_ + 16 * PITCH) = S##2;                                             \
    *reinterpret_cast<double2*>(q_ + 24 * PITCH) = S##3;                                             \
  } while (0)

// (removal experiments of r06, profiles/r06_contract_att.txt: -DCONTRACT_AB_NOBAR drops the per-block barrier, -DCONTRACT_AB_NOSTS the stage
// stores -- both give WRONG sums and exist only to price the two phases; never defined in the product build)
#ifdef CONTRACT_AB_NOBAR
#define BOGP_LOOP_BARRIER() asm volatile("" ::: "memory")
#else
#define BOGP_LOOP_BARRIER() __syncthreads()
#endif
#ifdef CONTRACT_AB_NOSTS
#undef BOGP_STAGE_STORE
#define BOGP_STAGE_STORE(S, buf_) asm volatile("" :: "v"((S##0).x), "v"((S##1).x), "v"((S##2).x), "v"((S##3).x))
#endif
#ifdef CONTRACT_AB_NOSTL  // every stage load reads tile 0 (L2-hot): no HBM latency behind the stage stores
#define BOGP_STAGE_ROW(kb_) 0
#else
#define BOGP_STAGE_ROW(kb_) (kb_)
#endif
  const double2* __restrict__ vp = a.Vp + lane;
  double2 bq[4][NR];
#pragma unroll
  for (int ni = 0; ni < NR; ++ni) {
    bq[0][ni] = vp[boff[ni]];
    bq[1][ni] = vp[boff[ni] + (size_t)min(1, kp_last) * 64];
  }

  // A lane = 16 k + i reads row k of the k-step, candidate 16 mi + i: pitch 640 B puts the four rows of a read in four
  // different 128-byte bank groups
  const int aoff = (lane >> 4) * PITCH + (lane & 15);

  BOGP_STAGE_LOAD(sa, 0);
  BOGP_STAGE_STORE(sa, 0);
  BOGP_STAGE_LOAD(sa, min(1, nkb - 1));  // tile 1 -> set A (stored at the end of block 0)
  BOGP_TR(trc, 2);

  // Two loops, one body each (r06): the full blocks run WITHOUT the per-tile guards -- as one code path with a run-time guard the compiler put a
  // compare / select / branch chain in front of every four MFMAs of the full blocks too and moved three of the four tile groups out of line
  // (six taken branches a k-step); an if / else of the two bodies inside ONE loop made the register allocator shuffle the 128 accumulators
  // (734 spills).  nkb_full = 8 jg is even, so the zone starts on buffer 0.
#define BOGP_BLOCK_PAIR(G)                                                                                                    \
  do {                                                                                                                        \
    BOGP_LOOP_BARRIER(); /* tile kb is in lds[0]; every wave is done with lds[1] */                                           \
    BOGP_TR(BOGP_TRB(kb), 0);                                                                                                 \
    BOGP_STAGE_LOAD(sb, min(kb + 2, nkb - 1));                                                                                \
    contract_block16<NR, G>(&lds[0], vp, boff, jt, aoff, kb, kp_last, bq, acc BOGP_TR_ARG(BOGP_TRB(kb)));                     \
    BOGP_STAGE_STORE(sa, 1); /* tile kb + 1 */                                                                                \
    BOGP_TR(BOGP_TRB(kb), 5);                                                                                                 \
    if (kb + 1 >= nkb) break;                                                                                                 \
    BOGP_LOOP_BARRIER(); /* tile kb + 1 is in lds[1]; every wave is done with lds[0] */                                       \
    BOGP_TR(BOGP_TRB(kb + 1), 0);                                                                                             \
    BOGP_STAGE_LOAD(sa, min(kb + 3, nkb - 1));                                                                                \
    contract_block16<NR, G>(&lds[KB * PITCH], vp, boff, jt, aoff, kb + 1, kp_last, bq, acc BOGP_TR_ARG(BOGP_TRB(kb + 1)));    \
    BOGP_STAGE_STORE(sb, 0); /* tile kb + 2 */                                                                                \
    BOGP_TR(BOGP_TRB(kb + 1), 5);                                                                                             \
  } while (0)
  int kb = 0;
  for (; kb < nkb_full; kb += 2) BOGP_BLOCK_PAIR(false);
  for (; kb < nkb; kb += 2) BOGP_BLOCK_PAIR(true);
#undef BOGP_BLOCK_PAIR
  BOGP_TR(trc, 3);
#undef BOGP_STAGE_LOAD
#undef BOGP_STAGE_STORE
#undef BOGP_LOOP_BARRIER

  // ---- epilogue: D[i][j] sits in lane 16 (i % 4) + j, register i / 4 ---------------------------------
  // (BOGP_LINT_NO_DRAIN / BOGP_LINT_NO_FENCE: negative controls of tests/test_isa_lint.py -- builds WITHOUT the drain / the fences must
  // be flagged by the ISA lint; never defined in the product build)
#ifndef BOGP_LINT_NO_DRAIN
  BOGP_MFMA16_DRAIN();
#endif
  // (the drain has no register operands: an empty volatile asm per accumulator behind it keeps the epilogue's reads from being
  // scheduled above it -- the MFMAs are inline asm whose results look ready at once to the compiler; kernels_small.hip)
#ifndef BOGP_LINT_NO_FENCE
#pragma unroll
  for (int mi = 0; mi < MR; ++mi)
#pragma unroll
    for (int ni = 0; ni < NR; ++ni) asm volatile("" : "+v"(acc[mi][ni]));
#endif
  __syncthreads();
  BOGP_TR(trc, 4);
#ifdef CONTRACT_AB_NOEPI  // no reduction: one accumulator word per thread keeps the MFMAs alive
  if (acc[0][0][0] == 12345.678) a.ss_part[tid] = acc[1][1][1] + acc[2][2][2] + acc[3][3][3];
  return;
#endif
  // red[slot][wave][row] with a row pitch of 65 doubles: the writes (lanes = 4 rows x 16 slots) fall on (4 slot + row)
  // mod 32 = every bank pair twice, the reads (lanes = 64 consecutive rows) are conflict free.  (The first layout,
  // [wave][row][slot], made every read a 64-way bank conflict: 13.7k cycles of epilogue per workgroup.)
  constexpr int RP = 65;
  double* red = lds;                    // [16 slots][NWJ][RP]
  double* red2 = lds + 16 * NWJ * RP;   // [NWJ][64]: per-wave partial sums
  const int q = lane >> 4, jc = lane & 15;
  if constexpr (NCP == 0) {
#pragma unroll
    for (int mi = 0; mi < MR; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double s = 0.0;
#pragma unroll
        for (int ni = 0; ni < NR; ++ni)
          if (valid[ni]) s = __builtin_fma(acc[mi][ni][r], acc[mi][ni][r], s);
        red[(jc * NWJ + w) * RP + 16 * mi + 4 * r + q] = s;  // slot = column inside the 16-tile
      }
  } else {
    constexpr int FPG = NCP / 16;  // fragments per point
    double head[MR / FPG][NR];     // row 0 of each point, for this lane's column jc: D[16 mi0][jc] = lane jc, register 0
#pragma unroll
    for (int g = 0; g < MR / FPG; ++g)
#pragma unroll
      for (int ni = 0; ni < NR; ++ni) head[g][ni] = __shfl(acc[g * FPG][ni][0], jc);
#pragma unroll
    for (int mi = 0; mi < MR; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double s = 0.0;
#pragma unroll
        for (int ni = 0; ni < NR; ++ni)
          if (valid[ni]) s = __builtin_fma(acc[mi][ni][r], head[mi / FPG][ni], s);
        red[(jc * NWJ + w) * RP + 16 * mi + 4 * r + q] = s;
      }
  }
  __syncthreads();
  BOGP_TR(trc, 5);
  {  // every thread adds the 16 slots of one (wave, row) in a fixed order, then 64 threads add the four waves
    double s = 0.0;
#pragma unroll
    for (int sl = 0; sl < 16; ++sl) s += red[(sl * NWJ + w) * RP + lane];
    red2[w * 64 + lane] = s;
  }
  __syncthreads();
  if (tid < 64) {
    const double tot = ((red2[tid] + red2[64 + tid]) + red2[128 + tid]) + red2[192 + tid];
    if constexpr (NCP == 0) {
      a.ss_part[(size_t)jg * a.Mc + mc0 + tid] = tot;
    } else {  // record 1 + jg of point b: [0][c] = sum over this column group of C_0 C_c  (k_point_finish adds the groups)
      const int64_t m = mc0 + tid;
      const int64_t b = m / NCP;
      const int c = (int)(m - b * NCP);
      if (b < a.cross_B) a.ss_part[((size_t)b * (a.nJ + 1) + 1 + jg) * (2 * NCP) + c] = tot;
    }
  }
#ifdef CONTRACT_TRACE
  BOGP_TR(trc, 6);
  __syncthreads();
  if (trace_out) {
    unsigned long long* dst = trace_out + (size_t)blockIdx.x * (NWJ * TR_NST);
    const unsigned long long* src = &trc_all[0][0];
    for (int i = tid; i < NWJ * TR_NST; i += 256) dst[i] = src[i];
  }
#endif
}

// ---------------------------------------------------------------------------------------------------
// Kernel B'' (r06): the same contraction with NO LDS and NO barrier in the main loop.
//
// profiles/r06_contract_att.txt: in k_contract16 ONE wave does not keep its SIMD's MFMA pipe full -- the A reads (LDS) end up just in
// front of the MFMAs that use them, the B fragments (L2) retire behind the stage loads (HBM) on the in-order vmcnt, and whenever one of
// the two resident waves waits at the per-block barrier or stores its stage registers the pipe runs at what the other wave manages
// alone.  Here every wave fetches its own operands: the A fragment of a k-step is 4 rows x 16 candidates = four 128-byte runs of rT,
// read as ONE global_load_dwordx4 per PAIR of fragments (lane (k, i) takes candidates 2 i and 2 i + 1 of a 32-candidate half: fragment
// mi = 2 half + e holds candidates 32 half + 2 i + e -- any bijection of fragment rows onto candidates will do, the epilogue knows it),
// the B fragments as before.  A and B of k-pair t + DEPTH are requested before k-pair t is multiplied (DEPTH + 1 register slots of
// 32 VGPRs beside the 128 accumulators), the four waves of a workgroup run free of each other (the sibling waves' A requests hit
// L1 / L2 behind the first one) and meet only in the epilogue, which is k_contract16's and adds in the same order: the sums are
// bit-identical to k_contract16's.
// ---------------------------------------------------------------------------------------------------
#ifndef CONTRACT_D_DEPTH
#define CONTRACT_D_DEPTH 2
#endif
typedef double v2d_t __attribute__((ext_vector_type(2)));
// (A/B switches of profiles/r06_contract_direct_ab.txt: -DCONTRACT_D_NTA / -DCONTRACT_D_NTB = the A / B loads carry the non-temporal hint)
__device__ __forceinline__ double2 ld2_plain(const char* p) { return *reinterpret_cast<const double2*>(p); }
__device__ __forceinline__ double2 ld2_nt(const char* p) {
  const v2d_t v = __builtin_nontemporal_load(reinterpret_cast<const v2d_t*>(p));
  return make_double2(v.x, v.y);
}
#ifdef CONTRACT_D_NTA
#define BOGP_D_LDA ld2_nt
#else
#define BOGP_D_LDA ld2_plain
#endif
#ifdef CONTRACT_D_NTB
#define BOGP_D_LDB ld2_nt
#else
#define BOGP_D_LDB ld2_plain
#endif
constexpr int DD = CONTRACT_D_DEPTH;  // k-pairs in flight
constexpr int DR = DD + 1;            // register slots

#ifdef CONTRACT_D_TRACE
// (profiling builds only, -DCONTRACT_D_TRACE: shader-clock stamps of every wave -- entry, loop start, the start of EVERY k-pair, loop end, drain, reduction, exit --
// parked in LDS and written out at the end; tools/contract_d_trace.py -> profiles/r06_contract_d_trace.txt.  A k-pair's stamp is requested in front of its first load
// and waited for / stored behind its first four MFMAs, so that the scalar-memory round trip of s_memtime hides under matrix work.)
constexpr int DT_NST = 8 + 264;  // words per wave: 8 header + one per k-pair (N <= 2112)
#define BOGP_DT_NOW(idx_)                                                            \
  do {                                                                               \
    unsigned long long t_;                                                           \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_));                 \
    dtr[(idx_)] = t_;                                                                \
  } while (0)
#define BOGP_DT_ISSUE() asm volatile("s_memtime %0" : "=s"(dt_pending))
#define BOGP_DT_STORE(idx_)                                                          \
  do {                                                                               \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(dt_pending));                         \
    if ((idx_) < DT_NST) dtr[(idx_)] = dt_pending;                                   \
  } while (0)
__global__ __launch_bounds__(256, 2) void k_contract16d(ContractArgs a, unsigned long long* trace_out) {
#else
#define BOGP_DT_NOW(idx_) ((void)0)
#define BOGP_DT_ISSUE() ((void)0)
#define BOGP_DT_STORE(idx_) ((void)0)
__global__ __launch_bounds__(256, 2) void k_contract16d(ContractArgs a) {
#endif
  constexpr int NR = 4;
  constexpr int JT16 = NWJ * NR;
  constexpr int RP = 65;
  __shared__ __attribute__((aligned(16))) double lds[16 * NWJ * RP + NWJ * 64];  // the epilogue's reduction arrays only

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef CONTRACT_D_TRACE
  __shared__ unsigned long long dtr_all[NWJ][DT_NST];
  unsigned long long* dtr = dtr_all[w];
  unsigned long long dt_pending = 0;
  {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    dtr[0] = (unsigned long long)hwid | ((unsigned long long)xcc << 32);
  }
  BOGP_DT_NOW(1);
#endif
  const int nMt = a.nMt;
  const int jg = a.nJ - 1 - (int)(blockIdx.x / nMt);  // heaviest column group first (k_contract16)
  const int mt = blockIdx.x % nMt;
  const int64_t mc0 = (int64_t)mt * 64;
  const int NJ16 = a.NJ16, NKP = a.NKP;
  const int kmax16 = min((jg + 1) * JT16, NJ16);
  const int nkp_full = 2 * jg * JT16;    // k-pairs without a guard
#ifdef CONTRACT_AB_NOZONE  // (removal experiment, wrong sums: the diagonal zone is skipped -- what it costs = product - this)
  const int nkp = max(nkp_full, 2);
#else
  // k-pairs of this WAVE: its last tile (15 - w of the group, serpentine below) ends 2 w k-pairs before the group does -- the waves do not
  // wait for each other before the epilogue, so a wave that has nothing left to multiply stops requesting operands too
  const int nkp = min(2 * kmax16, max(2 * ((jg + 1) * JT16 - w), 2));
#endif
  const int kp_last = nkp - 1;

  int jt[NR];
  bool valid[NR];
  const char* vb[NR];
#pragma unroll
  for (int ni = 0; ni < NR; ++ni) {
    const int j = jg * JT16 + ((ni & 1) ? NWJ * (ni + 1) - 1 - w : NWJ * ni + w);  // serpentine (k_contract16)
    valid[ni] = j < NJ16;
    jt[ni] = valid[ni] ? j : -1;
    vb[ni] = reinterpret_cast<const char*>(a.Vp + (size_t)min(j, NJ16 - 1) * NKP * 64);
  }
  const size_t Mc = (size_t)a.Mc;
  const char* ab = reinterpret_cast<const char*>(a.rT + mc0);
  const size_t kp_stride = 8 * Mc * sizeof(double);                                                    // 8 rows a k-pair
  unsigned voffA0 = (unsigned)(((size_t)(lane >> 4) * Mc + 2 * (lane & 15)) * sizeof(double));   // k-step 0 of the pair
  unsigned voffA1 = voffA0 + (unsigned)(4 * Mc * sizeof(double));                                // k-step 1
  unsigned voffB = (unsigned)lane * 16u;

  d4 acc[MR][NR];
#pragma unroll
  for (int mi = 0; mi < MR; ++mi)
#pragma unroll
    for (int ni = 0; ni < NR; ++ni) acc[mi][ni] = (d4){0.0, 0.0, 0.0, 0.0};

  double2 av[DR][2][2], bv[DR][NR];
  // load g of a k-pair, in the order the groups below need them (a group waits for the loads up to its own, vmcnt retires in order):
  // 0, 1 = the A pairs of k-step 0 (candidate halves 0 / 1), 2 .. 5 = the B fragments of tiles 0 .. 3, 6, 7 = the A pairs of k-step 1
#define BOGP_D_LOAD1(slot, g, ap_, kpc_, vb_)                                                                         \
  do {                                                                                                                \
    if ((g) < 2) av[slot][0][(g) & 1] = BOGP_D_LDA((ap_) + voffA0 + (((g) & 1) ? 256 : 0));                           \
    else if ((g) >= 6) av[slot][1][(g) & 1] = BOGP_D_LDA((ap_) + voffA1 + (((g) & 1) ? 256 : 0));                     \
    else bv[slot][(g) - 2] = BOGP_D_LDB((vb_)[(g) - 2] + (size_t)BOGP_D_BROW(kpc_) * 1024 + voffB);                   \
  } while (0)
  // The k-pair's eight loads go out as ONE block in front of its 32 MFMAs (CONTRACT_D_LBLOCK = 8; 1 / 2 / 4 = one load in front of every group / every second /
  // every fourth: the first version of this kernel).  The phase trace (profiles/r06_contract_d_trace.txt) showed a FULL k-pair taking 4284 cycles where the two waves
  // of a SIMD need 4096: every hand-over of the issue port between them costs a few cycles, and a load between two groups of MFMAs is a hand-over.  58.4 -> 58.0 ms.
  // NO VALU IN THE LOOP: the address of every load is a scalar base (advanced by scalar arithmetic) + the lane's constant 32-bit offset, the `v_offset, s[base]`
  // form of global_load.  hipcc only selects that form when it sees the zero-extension of the offset next to the load; hoisted out of the loop (it is loop
  // invariant) the offset is a 64-bit register pair and every load gets a v_lshl_add_u64 -- a 64-bit VALU operation, i.e. one that runs on the SAME FP64 pipe as
  // the MFMAs: eight of them per k-pair and wave.  Passing the three offsets through an empty asm inside the loop keeps the zero-extension where the load is.
  // 58.0 -> 55.9 ms a step at C3 (0.917 -> 0.956 of peak), 454.9 -> 439.1 ms at C5.  (-DCONTRACT_D_NO_SADDR: the A/B switch of profiles/r06_contract_d_trace.txt.)
#ifndef CONTRACT_D_NO_SADDR
#define BOGP_D_OPAQUE_OFFSETS() asm volatile("" : "+v"(voffA0), "+v"(voffA1), "+v"(voffB))
#else
#define BOGP_D_OPAQUE_OFFSETS() ((void)0)
#endif
#ifndef CONTRACT_D_LBLOCK
#define CONTRACT_D_LBLOCK 8
#endif
#define BOGP_D_LOADS_AT(g, slot, ap_, kpc_, vb_)                                                                      \
  do {                                                                                                                \
    if ((g) % CONTRACT_D_LBLOCK == 0) {                                                                               \
      _Pragma("unroll") for (int l_ = 0; l_ < CONTRACT_D_LBLOCK; ++l_) BOGP_D_LOAD1(slot, (g) + l_, ap_, kpc_, vb_);  \
    }                                                                                                                 \
  } while (0)
  // one k-pair: 8 groups of 4 MFMAs (k-step h = g / 4, tile ni = g % 4); load g of k-pair kp_ + DD goes out in front of group g, so
  // that the requests ride in the shadow of the MFMAs instead of in a block between two k-pairs (the sched_barriers pin that order)
#define BOGP_D_KPAIR(G, u, kp_)                                                                                       \
  do {                                                                                                                \
    const int kpc_ = min((kp_) + DD, kp_last);                                                                        \
    const char* ap_ = ab + (size_t)BOGP_D_AROW(kpc_) * kp_stride;                                                     \
    const int k16_ = (kp_) >> 1;                                                                                      \
    BOGP_D_OPAQUE_OFFSETS();                                                                                          \
    /* diagonal zone: a tile that is past its diagonal at k-pair kpc_ needs no fragment -- its request goes to the address */ \
    /* of the wave's last tile instead (an L1 hit on a line that is on its way anyway, no second L2 request)            */ \
    const char* vbe_[NR];                                                                                             \
    _Pragma("unroll") for (int ni = 0; ni < NR; ++ni) vbe_[ni] = (!(G) || ((kpc_ >> 1) <= jt[ni])) ? vb[ni] : vb[NR - 1]; \
    BOGP_DT_ISSUE();                                                                                                  \
    _Pragma("unroll") for (int g = 0; g < 8; ++g) {                                                                   \
      if (g == 1) BOGP_DT_STORE(8 + (kp_));                                                                           \
      BOGP_D_LOADS_AT(g, ((u) + DD) % DR, ap_, kpc_, vbe_);                                                           \
      __builtin_amdgcn_sched_barrier(0);                                                                              \
      if (!(G) || k16_ <= jt[g & 3]) {                                                                                \
        const double b_ = (g >> 2) == 0 ? bv[u][g & 3].x : bv[u][g & 3].y;                                            \
        _Pragma("unroll") for (int mi = 0; mi < MR; ++mi)                                                             \
          mfma16_acc((mi & 1) ? av[u][g >> 2][mi >> 1].y : av[u][g >> 2][mi >> 1].x, b_, acc[mi][g & 3]);             \
      }                                                                                                               \
      __builtin_amdgcn_sched_barrier(0);                                                                              \
    }                                                                                                                 \
  } while (0)

#ifdef CONTRACT_AB_FIXB  // (removal experiment, wrong sums: every B fragment from the tile's first k-pair)
#define BOGP_D_BROW(k_) 0
#else
#define BOGP_D_BROW(k_) (k_)
#endif
#ifdef CONTRACT_AB_FIXA  // (removal experiment, wrong sums: every A pair from the tile's first k-pair -- L1 / L2 hits, no HBM latency)
#define BOGP_D_AROW(k_) 0
#else
#define BOGP_D_AROW(k_) (k_)
#endif
#pragma unroll
  for (int t = 0; t < DD; ++t) {
    const int kpc = min(t, kp_last);
    const char* ap = ab + (size_t)kpc * kp_stride;
#pragma unroll
    for (int g = 0; g < 8; ++g) BOGP_D_LOAD1(t, g, ap, kpc, vb);
  }

  BOGP_DT_NOW(2);
  int kp = 0;
  // full k-pairs, DR at a time (static register slots); what is left of them goes through the guarded loop (its guards hold there)
  for (; kp + DR <= nkp_full; kp += DR) {
#pragma unroll
    for (int u = 0; u < DR; ++u) BOGP_D_KPAIR(false, u, kp + u);
  }
  for (; kp < nkp; kp += DR) {
#pragma unroll
    for (int u = 0; u < DR; ++u)
      if (kp + u < nkp) BOGP_D_KPAIR(true, u, kp + u);
  }
#undef BOGP_D_KPAIR
#undef BOGP_D_LOAD1
  BOGP_DT_NOW(3);

  // ---- epilogue (k_contract16's): D[i][j] sits in lane 16 (i % 4) + j, register i / 4; row i of fragment mi = candidate
  // 32 (mi / 2) + 2 i + (mi % 2)
  // (BOGP_LINT_NO_DRAIN / BOGP_LINT_NO_FENCE: the negative controls of tests/test_isa_lint.py, as in k_contract16)
#ifndef BOGP_LINT_NO_DRAIN
  BOGP_MFMA16_DRAIN();
#endif
#ifndef BOGP_LINT_NO_FENCE
#pragma unroll
  for (int mi = 0; mi < MR; ++mi)
#pragma unroll
    for (int ni = 0; ni < NR; ++ni) asm volatile("" : "+v"(acc[mi][ni]));
#endif
#ifdef CONTRACT_AB_NOEPI  // (removal experiment: no reduction; one accumulator word per thread keeps the MFMAs alive)
  if (acc[0][0][0] == 12345.678) a.ss_part[tid] = acc[1][1][1] + acc[2][2][2] + acc[3][3][3];
  return;
#endif
  BOGP_DT_NOW(4);
  double* red = lds;                    // [16 slots][NWJ][RP]
  double* red2 = lds + 16 * NWJ * RP;   // [NWJ][64]: per-wave partial sums
  const int q = lane >> 4, jc = lane & 15;
#pragma unroll
  for (int mi = 0; mi < MR; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double s = 0.0;
#pragma unroll
      for (int ni = 0; ni < NR; ++ni)
        if (valid[ni]) s = __builtin_fma(acc[mi][ni][r], acc[mi][ni][r], s);
      red[(jc * NWJ + w) * RP + 32 * (mi >> 1) + 2 * (4 * r + q) + (mi & 1)] = s;  // slot = column inside the 16-tile
    }
  // the wave's own slots: no workgroup barrier needed between its writes and its reads
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  {
    double s = 0.0;
#pragma unroll
    for (int sl = 0; sl < 16; ++sl) s += red[(sl * NWJ + w) * RP + lane];
    red2[w * 64 + lane] = s;
  }
  BOGP_DT_NOW(5);
  __syncthreads();
  if (tid < 64) {
    const double tot = ((red2[tid] + red2[64 + tid]) + red2[128 + tid]) + red2[192 + tid];
    a.ss_part[(size_t)jg * a.Mc + mc0 + tid] = tot;
  }
#ifdef CONTRACT_D_TRACE
  BOGP_DT_NOW(6);
  dtr[7] = (unsigned long long)nkp | ((unsigned long long)nkp_full << 32);
  __syncthreads();
  if (trace_out) {
    unsigned long long* dst = trace_out + (size_t)blockIdx.x * (NWJ * DT_NST);
    const unsigned long long* src = &dtr_all[0][0];
    for (int i = tid; i < NWJ * DT_NST; i += 256) dst[i] = src[i];
  }
#endif
}
#undef BOGP_DT_NOW
#undef BOGP_DT_ISSUE
#undef BOGP_DT_STORE

// ---------------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------------
// (kernel A' for the squared-distance kernels without a fused trend; against kernel A everywhere: profiles/r05_corr_mfma_ab.txt)
static bool corr_mfma_enabled() { return true; }

hipError_t launch_corr_chunk(int kernel, const CorrArgs& a, int nMt, int S, hipStream_t st) {
  dim3 grid((unsigned)nMt, (unsigned)S);
  // (the general-nu Matern kernel is a squared-distance kernel too, but its profile -- K_nu: hundreds of operations a pair -- dwarfs the distance and
  // needs 270 VGPRs: it stays on kernel A, which holds it without spills)
  const bool sqdist = kernel == BOGP_KERNEL_SE || kernel == BOGP_KERNEL_MATERN12 || kernel == BOGP_KERNEL_MATERN32 || kernel == BOGP_KERNEL_MATERN52;
  if (sqdist && a.pv == 0 && a.xnorm && corr_mfma_enabled()) {
    const int KS = (a.d + 3) / 4;
    const size_t shm = (size_t)max(4 * KS * 64, 2048) * sizeof(double);  // <= 160 KB up to d = 320, like kernel A
    CorrDims dm{a.M, a.m0, a.Mc, a.d, a.Np, a.nblk_per_split, a.wld};
#define BOGP_LAUNCH_CORR_MFMA(K)                                                                                                      \
  do {                                                                                                                                \
    if (shm > 64 * 1024) {                                                                                                            \
      hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_corr_mfma<K>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                          (int)shm);                                                                                  \
      if (e_ != hipSuccess) return e_;                                                                                                \
    }                                                                                                                                 \
    hipLaunchKernelGGL((k_corr_mfma<K>), grid, 256, shm, st, a.Xs, a.sqrt_theta, a.XthT, a.xnorm, a.gamma, a.wvec, a.rT, a.mu_part,   \
                       a.w_part, dm);                                                                                                 \
  } while (0)
    switch (kernel) {
      case BOGP_KERNEL_SE: BOGP_LAUNCH_CORR_MFMA(BOGP_KERNEL_SE); break;
      case BOGP_KERNEL_MATERN12: BOGP_LAUNCH_CORR_MFMA(BOGP_KERNEL_MATERN12); break;
      case BOGP_KERNEL_MATERN32: BOGP_LAUNCH_CORR_MFMA(BOGP_KERNEL_MATERN32); break;
      default: BOGP_LAUNCH_CORR_MFMA(BOGP_KERNEL_MATERN52); break;
    }
#undef BOGP_LAUNCH_CORR_MFMA
    return hipGetLastError();
  }
  size_t shm = (size_t)(max(64 * a.d, 512) + (a.pv > 0 ? 32 * 64 : 0)) * sizeof(double);
  CorrDims dm{a.M, a.m0, a.Mc, a.d, a.Np, a.nblk_per_split, a.wld};
  // the candidate tile is 64 x d doubles of dynamic LDS: above the 64 KB default (d > 128) the kernel has to be allowed
  // more, up to the CU's 160 KB (d <= 320; occupancy then drops to one workgroup per CU, which only matters for the
  // ~10 % of the sweep this producer accounts for)
#define BOGP_LAUNCH_CORR_PV(K, PV)                                                                                       \
  do {                                                                                                                   \
    if (shm > 64 * 1024) {                                                                                               \
      hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_corr_chunk<K, PV>),                           \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);                         \
      if (e_ != hipSuccess) return e_;                                                                                   \
    }                                                                                                                    \
    hipLaunchKernelGGL((k_corr_chunk<K, PV>), grid, 256, shm, st, a.Xs, a.sqrt_theta, a.XthT, a.gamma, a.wvec, a.rT, a.mu_part, \
                       a.w_part, a.Wrow, a.t_part, dm);                                                                  \
  } while (0)
#define BOGP_LAUNCH_CORR(K)                                    \
  do {                                                         \
    if (a.pv == 0) BOGP_LAUNCH_CORR_PV(K, 0);                  \
    else if (a.pv == 16) BOGP_LAUNCH_CORR_PV(K, 16);           \
    else if (a.pv == 32) BOGP_LAUNCH_CORR_PV(K, 32);           \
    else return hipErrorInvalidValue;                          \
  } while (0)
  switch (kernel) {
    case BOGP_KERNEL_SE: BOGP_LAUNCH_CORR(BOGP_KERNEL_SE); break;
    case BOGP_KERNEL_MATERN12: BOGP_LAUNCH_CORR(BOGP_KERNEL_MATERN12); break;
    case BOGP_KERNEL_MATERN32: BOGP_LAUNCH_CORR(BOGP_KERNEL_MATERN32); break;
    case BOGP_KERNEL_ABSEXP: BOGP_LAUNCH_CORR(BOGP_KERNEL_ABSEXP); break;
    case BOGP_KERNEL_CUBIC: BOGP_LAUNCH_CORR(BOGP_KERNEL_CUBIC); break;
    case BOGP_KERNEL_GENEXP: BOGP_LAUNCH_CORR(BOGP_KERNEL_GENEXP); break;
    case BOGP_KERNEL_MATERN_NU: BOGP_LAUNCH_CORR(BOGP_KERNEL_MATERN_NU); break;
    default: BOGP_LAUNCH_CORR(BOGP_KERNEL_MATERN52); break;
  }
#undef BOGP_LAUNCH_CORR_PV
#undef BOGP_LAUNCH_CORR
  return hipGetLastError();
}

// (wider bases stay on the tile products: with 16 column tiles a wave -- PV = 256, quadratic trend at d = 20 -- the producer went from
// 6.8 to 88 ms per 1e6 candidates: 128 accumulator registers beside the distance loop and 16 un-staged B loads a k-step;
// profiles/r04_trend_timing.txt)
int corr_trend_columns(int p) { return p <= 1 ? 0 : (p <= 16 ? 16 : (p <= 32 ? 32 : 0)); }

// four 16-column fragments per wave = 256-column groups (the NR = 2 instantiation of k_contract16 -- 128-column groups, three workgroups
// per CU -- lost in r01 and was never the default; removed with its switch in r06)
static constexpr int contract_nr() { return 4; }

#ifdef CONTRACT_TRACE
static unsigned long long* g_trace = nullptr;
static size_t g_trace_words = 0, g_trace_used = 0;
static int g_trace_dims[4] = {0, 0, 0, 0};
hipError_t launch_contract(const ContractArgs& a, hipStream_t st) {
  const size_t need = (size_t)a.nMt * a.nJ * NWJ * TR_NST;
  if (need > g_trace_words) {
    if (g_trace) (void)hipFree(g_trace);
    hipError_t e = hipMalloc((void**)&g_trace, need * sizeof(unsigned long long));
    if (e != hipSuccess) return e;
    g_trace_words = need;
  }
  // the stamps of the LARGEST launch seen are kept (a sweep's last chunk is a short one); smaller launches run the same code without the dump
  const bool keep = need >= g_trace_used;
  if (keep) {
    g_trace_used = need;
    g_trace_dims[0] = a.nMt; g_trace_dims[1] = a.nJ; g_trace_dims[2] = a.NJ16; g_trace_dims[3] = TR_NST;
  }
  hipLaunchKernelGGL(k_contract16<4>, dim3((unsigned)(a.nMt * a.nJ)), 256, 0, st, a, keep ? g_trace : (unsigned long long*)nullptr);
  return hipGetLastError();
}
// the last launch's stamps: dims = {nMt, nJ, NJ16, words per wave}; out may be NULL (size query)
hipError_t debug_contract_trace(unsigned long long* out, size_t cap_words, size_t* used_words, int* dims) {
  if (used_words) *used_words = g_trace_used;
  if (dims) for (int i = 0; i < 4; ++i) dims[i] = g_trace_dims[i];
  if (!out) return hipSuccess;
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) return e;
  return hipMemcpy(out, g_trace, std::min(cap_words, g_trace_used) * sizeof(unsigned long long), hipMemcpyDeviceToHost);
}
#elif defined(CONTRACT_D_TRACE)
static unsigned long long* g_trace = nullptr;
static size_t g_trace_words = 0, g_trace_used = 0;
static int g_trace_dims[4] = {0, 0, 0, 0};
hipError_t launch_contract(const ContractArgs& a, hipStream_t st) {
  const size_t need = (size_t)a.nMt * a.nJ * NWJ * DT_NST;
  if (need > g_trace_words) {
    if (g_trace) (void)hipFree(g_trace);
    hipError_t e = hipMalloc((void**)&g_trace, need * sizeof(unsigned long long));
    if (e != hipSuccess) return e;
    g_trace_words = need;
  }
  const bool keep = need >= g_trace_used;  // the stamps of the LARGEST launch seen are kept
  if (keep) {
    g_trace_used = need;
    g_trace_dims[0] = a.nMt; g_trace_dims[1] = a.nJ; g_trace_dims[2] = a.NJ16; g_trace_dims[3] = DT_NST;
  }
  hipLaunchKernelGGL(k_contract16d, dim3((unsigned)(a.nMt * a.nJ)), 256, 0, st, a, keep ? g_trace : (unsigned long long*)nullptr);
  return hipGetLastError();
}
hipError_t debug_contract_trace(unsigned long long* out, size_t cap_words, size_t* used_words, int* dims) {
  if (used_words) *used_words = g_trace_used;
  if (dims) for (int i = 0; i < 4; ++i) dims[i] = g_trace_dims[i];
  if (!out) return hipSuccess;
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) return e;
  return hipMemcpy(out, g_trace, std::min(cap_words, g_trace_used) * sizeof(unsigned long long), hipMemcpyDeviceToHost);
}
#else
// BOGP_CONTRACT_DIRECT=0: the LDS-staged kernel (k_contract16<4>); the A/B switch of profiles/r06_contract_direct_ab.txt
static bool contract_direct() {
  static const bool on = [] {
    const char* e = getenv("BOGP_CONTRACT_DIRECT");
    return !(e && atoi(e) == 0);
  }();
  return on;
}

hipError_t launch_contract(const ContractArgs& a, hipStream_t st) {
  if (contract_direct()) hipLaunchKernelGGL(k_contract16d, dim3((unsigned)(a.nMt * a.nJ)), 256, 0, st, a);
  else hipLaunchKernelGGL(k_contract16<4>, dim3((unsigned)(a.nMt * a.nJ)), 256, 0, st, a);
  return hipGetLastError();
}
#endif

int contract_cols_per_group() { return NWJ * contract_nr() * 16; }

// the cross-product flavour for the batched one-point path: ncp = 16 / 32 / 64 right-hand-side columns per point; always the
// 256-column workgroups (a.nJ must be ceil(Np / 256)); a.ss_part = the points' record array, a.cross_B = points
hipError_t launch_contract_cross(const ContractArgs& a, int ncp, hipStream_t st) {
  const dim3 grid((unsigned)(a.nMt * a.nJ));
#ifdef CONTRACT_TRACE
  unsigned long long* none = nullptr;
  if (ncp == 16) hipLaunchKernelGGL((k_contract16<4, 16>), grid, 256, 0, st, a, none);
  else if (ncp == 32) hipLaunchKernelGGL((k_contract16<4, 32>), grid, 256, 0, st, a, none);
  else if (ncp == 64) hipLaunchKernelGGL((k_contract16<4, 64>), grid, 256, 0, st, a, none);
  else return hipErrorInvalidValue;
#else
  if (ncp == 16) hipLaunchKernelGGL((k_contract16<4, 16>), grid, 256, 0, st, a);
  else if (ncp == 32) hipLaunchKernelGGL((k_contract16<4, 32>), grid, 256, 0, st, a);
  else if (ncp == 64) hipLaunchKernelGGL((k_contract16<4, 64>), grid, 256, 0, st, a);
  else return hipErrorInvalidValue;
#endif
  return hipGetLastError();
}

}  // namespace bogp
