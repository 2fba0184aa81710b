// ubench_diag.hip -- where do the ~26 us of the 64 x 64 diagonal block (factor + inverse, one 256-thread workgroup) go?
//   V0  the r01 routine of kernels_chol.hip (16 steps of 4 columns, two workgroup barriers per step), with clock stamps around
//       its three phases (A: 4x4 potf2 + inverse by ONE thread; B: block row Y = M z by 16 threads; C: rank-4 update)
//   V1  the 16 threads of the block row all factor the diagonal tile (no hand-over of M): one barrier less per step
//   V2  V1 + third-order (Halley) rsqrt step instead of two Newton steps                      <- ADOPTED in r02
//   V3  steps of EIGHT columns (32 threads factor the 8 x 8 block; ONE rank-8 update per step; 16 barriers per block)
// Measured on MI355X (profiles/r02_ubench_diag.txt): V0 21.7-21.9 us per block, V1 20.9-21.2, V2 19.3-19.5, V3 24.1.
// Reading: a dependent FP64 operation costs ~26 cycles on the pivot chain (three fewer per pivot = -2 us per block); the
// rank-k update phase is issue bound (rank-8 costs exactly two rank-4), and the wider diagonal block puts ~2x the
// instructions into the ONE wave that carries the chain -- fewer barriers do not pay for that.
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/ubench_diag.hip -o tools/probes/ubench_diag
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include <cstdlib>

constexpr int CB = 64;
constexpr int DIAG_SB = 16 + 4 * CB + 2;

__device__ __forceinline__ double rsqrt_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);
  double e = __builtin_fma(-x * y, y, 1.0);
  y = __builtin_fma(y * 0.5, e, y);
  e = __builtin_fma(-x * y, y, 1.0);
  y = __builtin_fma(y * 0.5, e, y);
  return y;
}

template <bool STAMP>
__device__ __forceinline__ int diag_v0(const double* cs, double* sb, double (&lo)[4][4], double (&z)[4][4], int tid, long long* st) {
  const int tr = tid >> 4, tc = tid & 15;
  double* mini = sb;
  double* Y = sb + 16;
  double* flag = sb + 16 + 256;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      z[i][c] = cs[(4 * tr + i) * (CB + 1) + 4 * tc + c];
      lo[i][c] = 0.0;
    }
  if (tid == 0) flag[0] = 0.0;
  long long tA = 0, tB = 0, tC = 0, t0 = STAMP ? clock64() : 0;
  for (int jb = 0; jb < 16; ++jb) {
    if (tr == jb && tc == jb) {
      double l[4][4], iv[4];
      int bad = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        double piv = z[j][j];
        if (!(piv > 0.0)) {
          if (bad == 0) bad = 4 * jb + j + 1;
          piv = 1.0;
        }
        const double inv = rsqrt_nr(piv);
        double sq = piv * inv;
        sq = __builtin_fma(__builtin_fma(-sq, sq, piv), 0.5 * inv, sq);
        l[j][j] = sq;
        iv[j] = inv;
#pragma unroll
        for (int i = j + 1; i < 4; ++i) l[i][j] = z[i][j] * inv;
#pragma unroll
        for (int i = j + 1; i < 4; ++i)
#pragma unroll
          for (int c = j + 1; c <= i; ++c) z[i][c] = __builtin_fma(-l[i][j], l[c][j], z[i][c]);
      }
      double mm[4][4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (i < c) {
            mm[i][c] = 0.0;
          } else if (i == c) {
            mm[i][c] = iv[i];
          } else {
            double sacc = 0.0;
#pragma unroll
            for (int k = c; k < i; ++k) sacc = __builtin_fma(l[i][k], mm[k][c], sacc);
            mm[i][c] = -sacc * iv[i];
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          lo[i][c] = c <= i ? l[i][c] : 0.0;
          z[i][c] = mm[i][c];
          mini[4 * i + c] = mm[i][c];
          Y[i * CB + 4 * tc + c] = mm[i][c];
        }
      if (bad != 0 && flag[0] == 0.0) flag[0] = (double)bad;
    }
    __syncthreads();
    if (STAMP) { long long n = clock64(); tA += n - t0; t0 = n; }
    if (tr == jb && tc != jb) {
      double y[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          double sacc = 0.0;
#pragma unroll
          for (int k = 0; k <= i; ++k) sacc = __builtin_fma(mini[4 * i + k], z[k][c], sacc);
          y[i][c] = sacc;
        }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          z[i][c] = y[i][c];
          Y[i * CB + 4 * tc + c] = y[i][c];
        }
    }
    __syncthreads();
    if (STAMP) { long long n = clock64(); tB += n - t0; t0 = n; }
    if (tr > jb) {
      double lr[4][4], yc[4][4];
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          lr[i][k] = Y[k * CB + 4 * tr + i];
          yc[k][i] = Y[k * CB + 4 * tc + i];
        }
      if (tc == jb) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            lo[i][k] = lr[i][k];
            z[i][k] = 0.0;
          }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int k = 0; k < 4; ++k) z[i][c] = __builtin_fma(-lr[i][k], yc[k][c], z[i][c]);
    }
    if (STAMP) { long long n = clock64(); tC += n - t0; t0 = n; }
  }
  __syncthreads();
  if (STAMP && tid == 255) { st[0] = tA; st[1] = tB; st[2] = tC; }
  return (int)flag[0];
}


// 1/sqrt(x): hardware estimate + ONE third-order (Halley) step: e = 1 - x y^2, y' = y (1 + e/2 + 3 e^2/8).  Five dependent
// operations after v_rsq_f64 instead of the eight of two Newton steps; relative error ~ e0^3 (e0 ~ 2^-26) + rounding.
__device__ __forceinline__ double rsqrt_h(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  const double t = x * y;
  const double e = __builtin_fma(-t, y, 1.0);
  double p = __builtin_fma(0.375, e, 0.5);
  p = p * e;
  return __builtin_fma(y, p, y);
}

// V1: the 16 threads of block row jb ALL factor the 4x4 diagonal tile (published by its owner at the end of the previous
// step's update) and go straight on to their own piece of the block row: one barrier and one LDS round trip less per step.
template <bool STAMP, bool HALLEY>
__device__ __forceinline__ int diag_v1(const double* cs, double* sb, double (&lo)[4][4], double (&z)[4][4], int tid, long long* st) {
  const int tr = tid >> 4, tc = tid & 15;
  double* dtile = sb;           // [4][4] the current diagonal tile (lower part used)
  double* Y = sb + 16;          // [4][64]
  double* flag = sb + 16 + 256;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      z[i][c] = cs[(4 * tr + i) * (CB + 1) + 4 * tc + c];
      lo[i][c] = 0.0;
    }
  if (tid == 0) {
    flag[0] = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c) dtile[4 * i + c] = z[i][c];
  }
  __syncthreads();
  long long tA = 0, tB = 0, tC = 0, t0 = STAMP ? clock64() : 0;
  for (int jb = 0; jb < 16; ++jb) {
    if (tr == jb) {
      double a[4][4], l[4][4], iv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c <= i; ++c) a[i][c] = dtile[4 * i + c];
      int bad = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        double piv = a[j][j];
        const bool okp = piv > 0.0;
        bad = (!okp && bad == 0) ? 4 * jb + j + 1 : bad;
        piv = okp ? piv : 1.0;
        const double inv = HALLEY ? rsqrt_h(piv) : rsqrt_nr(piv);
        iv[j] = inv;
#pragma unroll
        for (int i = j + 1; i < 4; ++i) l[i][j] = a[i][j] * inv;
#pragma unroll
        for (int i = j + 1; i < 4; ++i)
#pragma unroll
          for (int c = j + 1; c <= i; ++c) a[i][c] = __builtin_fma(-l[i][j], l[c][j], a[i][c]);
        double sq = piv * inv;  // off the critical chain
        sq = __builtin_fma(__builtin_fma(-sq, sq, piv), 0.5 * inv, sq);
        l[j][j] = sq;
      }
      double mm[4][4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (i < c) {
            mm[i][c] = 0.0;
          } else if (i == c) {
            mm[i][c] = iv[i];
          } else {
            double sacc = 0.0;
#pragma unroll
            for (int k = c; k < i; ++k) sacc = __builtin_fma(l[i][k], mm[k][c], sacc);
            mm[i][c] = -sacc * iv[i];
          }
        }
      }
      if (tc == jb) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            lo[i][c] = c <= i ? l[i][c] : 0.0;
            z[i][c] = mm[i][c];
            Y[i * CB + 4 * tc + c] = mm[i][c];
          }
        if (bad != 0 && flag[0] == 0.0) flag[0] = (double)bad;
      } else {
        double y[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            double sacc = 0.0;
#pragma unroll
            for (int k = 0; k <= i; ++k) sacc = __builtin_fma(mm[i][k], z[k][c], sacc);
            y[i][c] = sacc;
          }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            z[i][c] = y[i][c];
            Y[i * CB + 4 * tc + c] = y[i][c];
          }
      }
    }
    __syncthreads();
    if (STAMP) { long long n = clock64(); tA += n - t0; t0 = n; }
    if (tr > jb) {
      double lr[4][4], yc[4][4];
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          lr[i][k] = Y[k * CB + 4 * tr + i];
          yc[k][i] = Y[k * CB + 4 * tc + i];
        }
      if (tc == jb) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            lo[i][k] = lr[i][k];
            z[i][k] = 0.0;
          }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int k = 0; k < 4; ++k) z[i][c] = __builtin_fma(-lr[i][k], yc[k][c], z[i][c]);
      if (tr == jb + 1 && tc == jb + 1) {  // the next diagonal tile is final: publish it for its block row
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int c = 0; c < 4; ++c) dtile[4 * i + c] = z[i][c];
      }
    }
    __syncthreads();
    if (STAMP) { long long n = clock64(); tC += n - t0; t0 = n; }
  }
  if (STAMP && tid == 255) { st[0] = tA; st[1] = tB; st[2] = tC; }
  return (int)flag[0];
}

// V3: steps of EIGHT columns.  The 32 threads of tile rows P = 2 p8 and P + 1 (one half wave) all factor the 8 x 8 diagonal
// block (published by its three owner threads at the end of the previous update); the top row forms Y_top = M11 z, the
// bottom row Y_bot = M22 (z - L21 Y_top) with Y_top handed over through LDS inside the wave (no workgroup barrier); the
// rest applies ONE rank-8 update.  16 workgroup barriers per block instead of 32, half the LDS round trips.
// L tiles are written into `Lout_lds` (the staging buffer, free after the load) as soon as they are final.
constexpr int DIAG_SB8 = 64 + 8 * CB + 2;
template <bool STAMP>
__device__ __forceinline__ int diag_v3(double* cs, double* sb, double (&z)[4][4], int tid, long long* st) {
  const int tr = tid >> 4, tc = tid & 15;
  double* dt = sb;             // [8][8] row-major, lower part
  double* Y = sb + 64;         // [8][64]
  double* flag = sb + 64 + 8 * CB;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) z[i][c] = cs[(4 * tr + i) * (CB + 1) + 4 * tc + c];
  __syncthreads();  // every tile is in registers: cs may now receive L
  if (tid == 0) flag[0] = 0.0;
  if (tr < 2 && tc <= tr) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c) dt[(4 * tr + i) * 8 + 4 * tc + c] = z[i][c];
  }
  __syncthreads();
  long long tA = 0, tC = 0, t0 = STAMP ? clock64() : 0;
  for (int p8 = 0; p8 < 8; ++p8) {
    const int P = 2 * p8;
    if (tr == P || tr == P + 1) {
      const bool bot = tr == P + 1;
      double a[8][8], iv[8], sq[8];
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int c = 0; c <= i; ++c) a[i][c] = dt[8 * i + c];
      int bad = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        double piv = a[j][j];
        const bool okp = piv > 0.0;
        bad = (!okp && bad == 0) ? 8 * p8 + j + 1 : bad;
        piv = okp ? piv : 1.0;
        const double inv = rsqrt_h(piv);
        iv[j] = inv;
#pragma unroll
        for (int i = j + 1; i < 8; ++i) a[i][j] = a[i][j] * inv;  // l[i][j]
#pragma unroll
        for (int i = j + 1; i < 8; ++i)
#pragma unroll
          for (int c = j + 1; c <= i; ++c) a[i][c] = __builtin_fma(-a[i][j], a[c][j], a[i][c]);
        double s_ = piv * inv;
        s_ = __builtin_fma(__builtin_fma(-s_, s_, piv), 0.5 * inv, s_);
        sq[j] = s_;
      }
      // inverse of a 4 x 4 lower factor (diagonal given by its reciprocals)
      double m11[4][4], m22[4][4];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            double v;
            if (i < c) {
              v = 0.0;
            } else if (i == c) {
              v = iv[4 * h + i];
            } else {
              double sacc = 0.0;
#pragma unroll
              for (int k = c; k < i; ++k) sacc = __builtin_fma(a[4 * h + i][4 * h + k], h ? m22[k][c] : m11[k][c], sacc);
              v = -sacc * iv[4 * h + i];
            }
            if (h) m22[i][c] = v; else m11[i][c] = v;
          }
      }
      if (!bot) {
        double y[4][4];
        if (tc == P) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              y[i][c] = m11[i][c];
              cs[(4 * tr + i) * (CB + 1) + 4 * tc + c] = c < i ? a[i][c] : (c == i ? sq[i] : 0.0);  // L11
            }
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              double sacc = 0.0;
#pragma unroll
              for (int k = 0; k <= i; ++k) sacc = __builtin_fma(m11[i][k], z[k][c], sacc);
              y[i][c] = sacc;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            z[i][c] = y[i][c];
            Y[i * CB + 4 * tc + c] = y[i][c];
          }
      }
      // hand-over inside the wave: the top row's Y is in LDS before the bottom row reads it
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      if (bot) {
        double y[4][4];
        if (tc == P + 1) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              y[i][c] = m22[i][c];
              cs[(4 * tr + i) * (CB + 1) + 4 * tc + c] = c < i ? a[4 + i][4 + c] : (c == i ? sq[4 + i] : 0.0);  // L22
            }
          if (bad != 0 && flag[0] == 0.0) flag[0] = (double)bad;
        } else {
          double t[4][4];  // z_bot - L21 Y_top  (z_bot = 0 for the tile left of the diagonal, whose L21 is final now)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              double sacc = tc == P ? 0.0 : z[i][c];
#pragma unroll
              for (int k = 0; k < 4; ++k) sacc = __builtin_fma(-a[4 + i][k], Y[k * CB + 4 * tc + c], sacc);
              t[i][c] = sacc;
            }
          if (tc == P) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int c = 0; c < 4; ++c) cs[(4 * tr + i) * (CB + 1) + 4 * tc + c] = a[4 + i][c];  // L21
          }
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              double sacc = 0.0;
#pragma unroll
              for (int k = 0; k <= i; ++k) sacc = __builtin_fma(m22[i][k], t[k][c], sacc);
              y[i][c] = sacc;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            z[i][c] = y[i][c];
            Y[(4 + i) * CB + 4 * tc + c] = y[i][c];
          }
      }
    }
    __syncthreads();
    if (STAMP) { long long n = clock64(); tA += n - t0; t0 = n; }
    if (tr > P + 1) {
      double lr[4][8], yc[8][4];
#pragma unroll
      for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          lr[i][k] = Y[k * CB + 4 * tr + i];
          yc[k][i] = Y[k * CB + 4 * tc + i];
        }
      if (tc == P || tc == P + 1) {  // this tile's piece of L is final; from here on the registers accumulate W
        const int o = tc == P ? 0 : 4;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            cs[(4 * tr + i) * (CB + 1) + 4 * tc + k] = tc == P ? lr[i][k] : lr[i][4 + k];
            z[i][k] = 0.0;
          }
        (void)o;
      }
      if (tc == P + 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int k = 4; k < 8; ++k) z[i][c] = __builtin_fma(-lr[i][k], yc[k][c], z[i][c]);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int k = 0; k < 8; ++k) z[i][c] = __builtin_fma(-lr[i][k], yc[k][c], z[i][c]);
      }
      if ((tr == P + 2 || tr == P + 3) && tc >= P + 2 && tc <= tr) {  // the next 8 x 8 diagonal block is final: publish it
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int c = 0; c < 4; ++c) dt[(4 * (tr - P - 2) + i) * 8 + 4 * (tc - P - 2) + c] = z[i][c];
      }
    }
    __syncthreads();
    if (STAMP) { long long n = clock64(); tC += n - t0; t0 = n; }
  }
  if (STAMP && tid == 255) { st[0] = tA; st[1] = 0; st[2] = tC; }
  return (int)flag[0];
}

template <int V, bool STAMP>
__global__ __launch_bounds__(256) void k_diag(const double* __restrict__ A, double* __restrict__ Lout, double* __restrict__ Wout, int reps,
                                              long long* st) {
  __shared__ __attribute__((aligned(16))) double cs[CB * (CB + 1)];
  __shared__ __attribute__((aligned(16))) double sb[DIAG_SB8];
  const int tid = threadIdx.x;
  double lo[4][4], z[4][4];
  long long t0 = clock64();
  for (int rep = 0; rep < reps; ++rep) {
    for (int e = tid; e < CB * CB; e += 256) {
      const int r = e & 63, c = e >> 6;
      cs[r * (CB + 1) + c] = A[(size_t)c * CB + r];
    }
    __syncthreads();
    if (V == 0) diag_v0<STAMP>(cs, sb, lo, z, tid, st);
    else if (V == 1) diag_v1<STAMP, false>(cs, sb, lo, z, tid, st);
    else if (V == 2) diag_v1<STAMP, true>(cs, sb, lo, z, tid, st);
    else diag_v3<STAMP>(cs, sb, z, tid, st);
    __syncthreads();
  }
  if (tid == 0) st[3] = clock64() - t0;
  const int tr = tid >> 4, tc = tid & 15;
  for (int c = 0; c < 4; ++c)
    for (int i = 0; i < 4; ++i) {
      const int r = 4 * tr + i, col = 4 * tc + c;
      Lout[col * CB + r] = r >= col ? (V == 3 ? cs[r * (CB + 1) + col] : lo[i][c]) : 0.0;
      Wout[col * CB + r] = tc <= tr ? z[i][c] : 0.0;
    }
}

int main() {
  std::vector<double> A(CB * CB), B(CB * CB);
  srand(1);
  for (auto& b : B) b = rand() / (double)RAND_MAX - 0.5;
  for (int i = 0; i < CB; ++i)
    for (int j = 0; j < CB; ++j) {
      double s = i == j ? 1.0 : 0.0;
      for (int k = 0; k < CB; ++k) s += 0.05 * B[i * CB + k] * B[j * CB + k];
      A[j * CB + i] = s;
    }
  std::vector<double> Lr(A);  // CPU Cholesky (column-major, lower)
  for (int j = 0; j < CB; ++j) {
    for (int k = 0; k < j; ++k)
      for (int i = j; i < CB; ++i) Lr[j * CB + i] -= Lr[k * CB + i] * Lr[k * CB + j];
    const double s = std::sqrt(Lr[j * CB + j]);
    for (int i = j; i < CB; ++i) Lr[j * CB + i] /= s;
  }
  double *dA, *dL, *dW;
  long long* dst;
  hipMalloc(&dA, sizeof(double) * CB * CB);
  hipMalloc(&dL, sizeof(double) * CB * CB);
  hipMalloc(&dW, sizeof(double) * CB * CB);
  hipMalloc(&dst, 8 * sizeof(long long));
  hipMemcpy(dA, A.data(), sizeof(double) * CB * CB, hipMemcpyHostToDevice);
  const int reps = 200;
  for (int pass = 0; pass < 8; ++pass) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    switch (pass) {
      case 0: hipLaunchKernelGGL((k_diag<0, false>), dim3(1), 256, 0, 0, dA, dL, dW, reps, dst); break;
      case 1: hipLaunchKernelGGL((k_diag<0, true>), dim3(1), 256, 0, 0, dA, dL, dW, reps, dst); break;
      case 2: hipLaunchKernelGGL((k_diag<1, false>), dim3(1), 256, 0, 0, dA, dL, dW, reps, dst); break;
      case 3: hipLaunchKernelGGL((k_diag<1, true>), dim3(1), 256, 0, 0, dA, dL, dW, reps, dst); break;
      case 4: hipLaunchKernelGGL((k_diag<2, false>), dim3(1), 256, 0, 0, dA, dL, dW, reps, dst); break;
      case 5: hipLaunchKernelGGL((k_diag<2, true>), dim3(1), 256, 0, 0, dA, dL, dW, reps, dst); break;
      case 6: hipLaunchKernelGGL((k_diag<3, false>), dim3(1), 256, 0, 0, dA, dL, dW, reps, dst); break;
      default: hipLaunchKernelGGL((k_diag<3, true>), dim3(1), 256, 0, 0, dA, dL, dW, reps, dst); break;
    }
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long st[4];
    hipMemcpy(st, dst, sizeof(st), hipMemcpyDeviceToHost);
    std::vector<double> L(CB * CB), W(CB * CB);
    hipMemcpy(L.data(), dL, sizeof(double) * CB * CB, hipMemcpyDeviceToHost);
    hipMemcpy(W.data(), dW, sizeof(double) * CB * CB, hipMemcpyDeviceToHost);
    double el = 0, ew = 0;
    for (int j = 0; j < CB; ++j)
      for (int i = j; i < CB; ++i) el = std::fmax(el, std::fabs(L[j * CB + i] - Lr[j * CB + i]));
    for (int i = 0; i < CB; ++i)  // W L = I
      for (int j = 0; j < CB; ++j) {
        double s = 0;
        for (int k = 0; k < CB; ++k) s += W[k * CB + i] * (k >= j ? Lr[j * CB + k] : 0.0);
        ew = std::fmax(ew, std::fabs(s - (i == j)));
      }
    printf("V%d %s: %.2f us per block (%d reps in one launch); total ticks/rep %lld; err L %.2e, |W L - I| %.2e\n", pass / 2, (pass & 1) ? "stamped" : "plain  ",
           ms * 1e3 / reps, reps, st[3] / reps, el, ew);
    if (pass & 1) printf("   last rep, thread 255: phase A (+barrier) %lld, B (+barrier) %lld, C %lld ticks per 16 steps\n", st[0], st[1], st[2]);
  }
  return 0;
}
