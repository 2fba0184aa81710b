// kernels_small.hip -- the whole sweep in ONE kernel for small training sets (Np <= 512, d <= 60, constant trend).
//
// Same mathematics as the chunked path of kernels_posterior.hip + kernels_acq.hip (GaussianProcess.predict, gpr.py:486-510,
// the criteria of acquisition_fun.py and np.argmax), different schedule.  At N <= 512 the chunked path is a poor fit
// (45 % of the FP64 peak for the whole step at C2): with one or two 256-column groups, two thirds of its
// 32-row blocks lie in the diagonal zone (per-block barrier + HBM round trip of the r tile over half-empty MFMA work), and
// the step is three launches + a final reduce.  Here a 512-thread workgroup owns 64 candidates end to end:
//
//   for each 256-row PANEL p of the training set (one or two):
//     produce  r[n][m] = corr(theta, |x*_m - x_n|) for the panel's rows straight into LDS (128 KB, XOR-swizzled so that
//              both the row-wise writes and the MFMA A-fragment reads are bank-conflict free) -- the FP64-VALU producer of
//              k_corr_chunk (scalar loads of the theta-scaled transposed training rows), + the partial sums r.gamma, r.w
//     contract every sixteen-wide column tile j of V = L^-1 against the panel with v_mfma_f64_16x16x4_f64 (accumulators in
//              architectural VGPRs, B fragments from the packed V of k_pack_V, two k-pairs ahead): wave w owns the tiles
//              {w, 15-w, 16+w, 31-w} -- the same triangle area for every wave in every panel -- and runs them with NO
//              barrier and NO global traffic for r; tiles whose diagonal has been passed drop out wave-uniformly
//   epilogue: sum of squares per candidate (LDS, fixed order), posterior (mu, MSE), the q criteria, optional full outputs,
//             block argmax; the LAST workgroup to finish (device-scope ticket) reduces the per-block winners -> one launch.
//
// Accumulators persist in registers across the panels (16 d4 per wave = 128 VGPRs), which is what bounds Np at 512.
// Algorithmic work per candidate: N^2 + N (3d + 5 + 2p) flop; compulsory HBM bytes: 8 d per candidate
// + V (2 MB at N = 512, L2-resident) -- the kernel is bound by the FP64 pipe (MFMA + the producer's VALU work, which
// share it and therefore add).
#include <algorithm>
#include <cstdlib>

#include "bogp_device.h"
#include "bogp_internal.h"

namespace bogp {

namespace {
constexpr int SM_MT = 64;        // candidates per workgroup
constexpr int SM_PANEL = 256;    // training rows resident in LDS at a time
constexpr int SM_WAVES = 8;      // 512 threads: two waves per SIMD; each owns 4 / 2 sixteen-column tiles (SM_NR: Np <= 512 / 256)

typedef double d4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mfma16s(double a, double b, d4s& c) {
  asm("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
#define BOGP_SM_DRAIN() asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory")

// One 32-row block (4 k-pairs x 2 k-steps) of the resident panel for the wave's SM_NR column tiles.  `tile` points at the
// block's first row in LDS (pitch 64 doubles, swizzled: element (row, c) sits at row * 64 + (c ^ 16 (row & 1))); aoffm[mi]
// is this lane's offset for fragment mi inside a 4-row k-step (the swizzle is folded in: the row parity of a lane is fixed).
// A tile is active while kb16 <= jt[ni] (wave-uniform); GUARDED (a run-time, wave-uniform flag: ONE inlined copy of this
// body keeps the kernel inside 256 VGPRs) = false when all four tiles are active for the whole block.
// SM_NR column tiles per wave and a RING of B-fragment slots (one k-pair each) per tile: fragments are requested RING - 2
// k-pairs ahead.  With four tiles per wave (Np > 256) RING = 4 as before; with two tiles per wave (Np <= 256) the registers the
// absent tiles would have taken carry an 8-slot ring instead: the wave has half the MFMAs per k-pair to cover each round trip
// to L2 (+7 % at N = 256; profiles/r03_sweep_scaling.txt).  PH = this block's position inside the RING / 4 blocks the caller
// unrolls, so that every slot index is a compile-time constant.
template <int SM_MR, int SM_NR, int RING, int PH, bool KPB>
__device__ __forceinline__ void small_block16(const bool GUARDED, const double* __restrict__ tile, const char* const (&vbase)[SM_NR],
                                              unsigned& voffB, const int (&jt)[SM_NR], const int (&aoffm)[SM_MR],
                                              int kb, int kp_clamp, double2 (&bq)[RING][SM_NR], d4s (&acc)[SM_MR][SM_NR]) {
  if constexpr (KPB) {
  // (r06, default: the k-pair's four B requests AND the eight A reads of its two k-steps in ONE block in front of its MFMAs -- one hand-over of the issue port a
  // k-pair instead of three; the LDS round trip is then exposed once a k-pair and left to the SIMD's other wave.  KPB = the eight-wave workgroups with four tiles a
  // wave (256 < Np <= 512): -1.7 % at N = 512, -3.5 % at 384; the schedules of Np <= 256 (two workgroups a CU, or two tiles a wave) lose 2-6 % with it and keep the
  // r02-r05 order below: A reads one k-step ahead of the MFMAs that use them)
  double af[2][SM_MR];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int kp = kb * 4 + s;
    int kb16 = kp >> 1;
    // (opaque per k-pair: shared between the two k-pairs of a sixteen-row group, hipcc parks each tile's test in a VGPR -- v_cndmask + v_cmp per tile and k-pair,
    // VALU instructions between the MFMA bursts -- instead of repeating a scalar compare)
    asm volatile("" : "+s"(kb16));
    constexpr int AHEAD = RING - 2;
    {
      const int kpn = min(kp + AHEAD, kp_clamp);
      asm volatile("" : "+v"(voffB));
#pragma unroll
      for (int ni = 0; ni < SM_NR; ++ni)
        bq[(4 * PH + s + AHEAD) & (RING - 1)][ni] = *reinterpret_cast<const double2*>(vbase[ni] + (size_t)kpn * 1024 + voffB);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const double* trow = tile + (4 * (2 * s + h)) * 64;
#pragma unroll
      for (int mi = 0; mi < SM_MR; ++mi) af[h][mi] = trow[aoffm[mi]];
    }
    __builtin_amdgcn_sched_barrier(0);
    // tile by tile, both k-steps of a tile behind ONE test: bursts of eight MFMAs, four tests a k-pair (every accumulator still sees k-step 0 before k-step 1)
#pragma unroll
    for (int ni = 0; ni < SM_NR; ++ni) {
      if (!GUARDED || kb16 <= jt[ni]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const double bv = h == 0 ? bq[(4 * PH + s) & (RING - 1)][ni].x : bq[(4 * PH + s) & (RING - 1)][ni].y;
#pragma unroll
          for (int mi = 0; mi < SM_MR; ++mi) mfma16s(af[h][mi], bv, acc[mi][ni]);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  } else {
  double af[2][SM_MR];
#pragma unroll
  for (int mi = 0; mi < SM_MR; ++mi) af[0][mi] = tile[aoffm[mi]];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int kp = kb * 4 + s;
    const int kb16 = kp >> 1;
    constexpr int AHEAD = RING - 2;
    {
      const int kpn = min(kp + AHEAD, kp_clamp);
      // (the schedules of Np <= 256 keep the 64-bit address adds: with the scalar-base form they measured 2-4 % SLOWER -- N = 128 0.990 -> 1.012 ms, N = 256
      // 2.193 -> 2.280 ms per 1e6 candidates; the offset is therefore not made opaque here and hipcc hoists its zero-extension as before)
#pragma unroll
      for (int ni = 0; ni < SM_NR; ++ni)
        bq[(4 * PH + s + AHEAD) & (RING - 1)][ni] = *reinterpret_cast<const double2*>(vbase[ni] + (size_t)kpn * 1024 + voffB);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int sub = 2 * s + h;
      if (sub < 7) {
        const double* trow = tile + (4 * (sub + 1)) * 64;
#pragma unroll
        for (int mi = 0; mi < SM_MR; ++mi) af[(sub + 1) & 1][mi] = trow[aoffm[mi]];
      }
#pragma unroll
      for (int ni = 0; ni < SM_NR; ++ni) {
        if (!GUARDED || kb16 <= jt[ni]) {
          const double bv = h == 0 ? bq[(4 * PH + s) & (RING - 1)][ni].x : bq[(4 * PH + s) & (RING - 1)][ni].y;
#pragma unroll
          for (int mi = 0; mi < SM_MR; ++mi) mfma16s(af[sub & 1][mi], bv, acc[mi][ni]);
        }
      }
    }
  }
  }
}
}  // namespace

// pointers read with wave-uniform addresses are separate __restrict__ arguments so that they become scalar loads
// SM_MR = sixteen-candidate A fragments per workgroup: 4 (64 candidates) for the bulk; 2 or 3 for the TAIL launch that
// spreads the last, incomplete round of 64-candidate workgroups over all CUs (the producer's lanes beyond 16 SM_MR idle,
// the contraction shrinks with SM_MR).
// NW = waves per workgroup.  8 (512 threads, 256-row panels, one workgroup per CU) up to Np = 512.  NW = 4 (r03; 256 threads,
// 128-row panels of 64 KB, four tiles {w, 7 - w, 8 + w, 15 - w} per wave) serves Np <= 256 with TWO workgroups per CU: below
// ~250 training points a workgroup that has the CU to itself is bound by its own latencies (load -> barrier -> produce ->
// barrier -> contract -> epilogue: 53 k cycles at N = 128 for < 10 k cycles of DP-pipe work), and a second, independent
// workgroup fills them.  A 4-wave workgroup carries each of its waves' sums as the two "virtual" waves of the 8-wave
// schedule they are made of (wave w = virtual waves w and 7 - w of the contraction, w and w + 4 of the producer), in the same
// order: the outputs are the SAME BITS as with NW = 8 (test_fused_small_sweep_tiles_per_wave_variants_are_bit_identical).
template <int KERNEL, int SM_MR, int SM_NR, int NW>
__global__ __launch_bounds__(64 * NW, 2) void k_sweep_small(const double* __restrict__ Xs, const double* __restrict__ sqrt_theta,
                                                        const double* __restrict__ XthT, const double* __restrict__ gamma,
                                                        const double* __restrict__ wvec, const double2* __restrict__ Vp,
                                                        SmallArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int NT = 64 * NW;      // threads
  constexpr int PANEL = 32 * NW;   // training rows resident in LDS at a time: every wave produces 32 of them
  constexpr int HW = 2 * NW;       // tiles per serpentine half
  static_assert(NW == 8 || (NW == 4 && SM_NR == 4), "4-wave workgroups carry four tiles per wave (Np <= 256)");
  double* rs = smem;                       // [PANEL][64] resident panel of r, swizzled
  double* xs = smem + PANEL * SM_MT;       // [d][64] theta-scaled candidate tile, k-major
  __shared__ int s_last;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int d = a.d, Np = a.Np, NJ16 = a.NJ16, NKP = a.NKP;
  constexpr int MT = 16 * SM_MR;  // candidates of this workgroup (lanes >= MT of the producer are idle)
  const int64_t mg0 = a.m_begin + (int64_t)blockIdx.x * MT;

  const double pexp = kernel_exponent<KERNEL>(sqrt_theta, d);
  const int d2 = (d + 1) & ~1;  // the producer walks the dimensions two at a time; xs is sized for d2 rows, row d (if any) is zero
  for (int idx = tid; idx < SM_MT * d; idx += NT) {
    const int row = idx / d, k = idx - row * d;
    const int64_t gm = mg0 + row;
    const double v = (row < MT && gm < a.M) ? Xs[gm * d + k] : 0.0;
    xs[k * SM_MT + row] = v * sqrt_theta[k];
  }
  for (int idx = tid; idx < SM_MT * (d2 - d); idx += NT) xs[d * SM_MT + idx] = 0.0;

  // column tiles of this wave: serpentine over the (up to) 32 tiles, so that in every panel every wave carries the same
  // number of sixteen-row groups: tiles {w, 15 - w} end inside panel 0, {16 + w, 31 - w} inside panel 1
  int jt[SM_NR];
  const char* vbase[SM_NR];  // packed V of tile ni (wave-uniform)
  bool valid[SM_NR];
#pragma unroll
  for (int ni = 0; ni < SM_NR; ++ni) {
    const int j = (ni >> 1) * HW + ((ni & 1) ? HW - 1 - w : w);
    valid[ni] = j < NJ16;
    jt[ni] = valid[ni] ? j : -1;
    vbase[ni] = reinterpret_cast<const char*>(Vp + (size_t)min(j, NJ16 - 1) * NKP * 64);
  }
  d4s acc[SM_MR][SM_NR];
#pragma unroll
  for (int mi = 0; mi < SM_MR; ++mi)
#pragma unroll
    for (int ni = 0; ni < SM_NR; ++ni) acc[mi][ni] = (d4s){0.0, 0.0, 0.0, 0.0};

  // A lane = 16 k + i reads row k of the k-step, candidate 16 mi + i; the row parity of a lane is (lane >> 4) & 1
  int aoffm[SM_MR];
  {
    const int lk = lane >> 4, li = lane & 15, par = lk & 1;
#pragma unroll
    for (int mi = 0; mi < SM_MR; ++mi) aoffm[mi] = lk * 64 + ((16 * mi + li) ^ (par << 4));
  }
  unsigned voffB = (unsigned)lane * 16u;
  const int kp_clamp = NKP - 1;

  // optional phase timing (BOGP_SMALL_STAMPS=1): wave-cycles spent producing / contracting / in the epilogue, summed
  // over all waves of the launch into a.stamps[0..3] (+ [4] = waves counted)
  long long t_mark = a.stamps ? clock64() : 0, t_prod = 0, t_mfma = 0;
#define BOGP_STAMP(acc_)                  \
  if (a.stamps) {                         \
    const long long now_ = clock64();     \
    acc_ += now_ - t_mark;                \
    t_mark = now_;                        \
  }
  long long t_pro = 0;
  double mu = 0.0, wd = 0.0;    // partial sums r . gamma, r . w over the rows this wave produces ...
  double mu1 = 0.0, wd1 = 0.0;  // ... NW = 4: those of the ODD panels (the rows of virtual wave w + 4), kept apart
  int jmax = -1;
#pragma unroll
  for (int ni = 0; ni < SM_NR; ++ni) jmax = max(jmax, jt[ni]);
  const int nkb_all = Np / 32;                                   // 32-row blocks of the training set
  const int nkb_w = a.need_var ? min(nkb_all, (jmax >> 1) + 1) : 0;  // ... that this wave's tiles need
  constexpr int RING = 16 / SM_NR;  // B-fragment slots per tile: 4 / 8 for 4 / 2 tiles per wave (64 VGPRs either way)
  constexpr int UNR = RING / 4;     // 32-row blocks per trip of the loop below (slot indices stay compile-time constants)
  double2 bq[RING][SM_NR];
#pragma unroll
  for (int kp0 = 0; kp0 < RING - 2; ++kp0)  // in flight while the first panel is produced
#pragma unroll
    for (int ni = 0; ni < SM_NR; ++ni) bq[kp0][ni] = *reinterpret_cast<const double2*>(vbase[ni] + (size_t)min(kp0, kp_clamp) * 1024 + voffB);
  // ONE loop over the 32-row blocks; at every panel boundary all waves meet, produce the next 256 rows of r into LDS and
  // meet again.  Between two boundaries a wave runs its blocks with no barrier; a wave whose tiles are finished idles at
  // the next boundary only (every wave carries the same MFMA count per panel, so they arrive together).
  for (int kb = 0; kb < nkb_all; kb += UNR) {
    if ((kb & (PANEL / 32 - 1)) == 0) {
      const int p = kb / (PANEL / 32);
      BOGP_STAMP(t_mfma);
      __syncthreads();  // every wave is done with the previous panel (and, for p = 0, the candidate tile is staged)
      BOGP_STAMP(t_pro);  // time spent waiting for the other waves
      const int nb = p * PANEL + w * 32;
      if (nb < Np) {
#pragma unroll 1
        for (int st = 0; st < 4; ++st) {
          const int n0 = nb + st * 8;
          double ac[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) ac[i] = dist_init<KERNEL>();
          // Software-pipelined, two dimensions per trip (r03): the 2 x 8 training values of trip t + 1 (two scalar loads) and the
          // lane's two candidate coordinates (LDS) are REQUESTED before trip t is computed and waited for at the top of
          // trip t + 1 -- one lgkmcnt(0) per trip, with 32 DP instructions of cover.  (r02 issued three loads, waited, computed:
          // with two waves per SIMD the SMEM round trip was exposed -- 430 instead of ~250 cycles per row of 64 pairs.)
          // Rows d .. d2-1 of xs / XthT are zero: they add (0 - 0)^2.  Same operations in the same order as before.
          const double* __restrict__ xp = XthT + n0;
          double c0[8], c1[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            c0[i] = xp[i];
            c1[i] = xp[(size_t)Np + i];
          }
          double xa0 = xs[lane], xa1 = xs[SM_MT + lane];
#pragma unroll 1
          for (int k = 0; k < d2; k += 2) {
            const int kn = k + 2 < d2 ? k + 2 : k;  // the last trip re-requests its own rows (never used)
            const double* __restrict__ y0 = xp + (size_t)kn * Np;
            const double* __restrict__ y1 = y0 + Np;
            double e0[8], e1[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              e0[i] = y0[i];
              e1[i] = y1[i];
            }
            const double xb0 = xs[kn * SM_MT + lane], xb1 = xs[(kn + 1) * SM_MT + lane];
#pragma unroll
            for (int i = 0; i < 8; ++i) ac[i] = dist_accumulate<KERNEL>(xa0 - c0[i], ac[i], pexp);
#pragma unroll
            for (int i = 0; i < 8; ++i) ac[i] = dist_accumulate<KERNEL>(xa1 - c1[i], ac[i], pexp);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              c0[i] = e0[i];
              c1[i] = e1[i];
            }
            xa0 = xb0;
            xa1 = xb1;
          }
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const double r = corr_profile<KERNEL>(ac[i]);
            const int nl = n0 + i - p * PANEL;
            rs[nl * 64 + (lane ^ ((nl & 1) << 4))] = r;
            if (NW == 8 || (p & 1) == 0) {
              mu = __builtin_fma(r, gamma[n0 + i], mu);
              wd = __builtin_fma(r, wvec[n0 + i], wd);
            } else {
              mu1 = __builtin_fma(r, gamma[n0 + i], mu1);
              wd1 = __builtin_fma(r, wvec[n0 + i], wd1);
            }
          }
        }
      }
      BOGP_STAMP(t_prod);
      __syncthreads();
      BOGP_STAMP(t_pro);
    }
#define BOGP_SMALL_BLOCK(PH_)                                                                                            \
    if (PH_ < UNR && kb + PH_ < nkb_w) {                                                                               \
      bool full = true;                                                                                                \
      _Pragma("unroll") for (int ni = 0; ni < SM_NR; ++ni) full = full && (jt[ni] >= 2 * (kb + PH_) + 1);               \
      const double* tile = rs + ((kb + PH_) & (PANEL / 32 - 1)) * 32 * 64;                                          \
      small_block16<SM_MR, SM_NR, RING, (PH_ < UNR ? PH_ : 0), (NW == 8 && SM_NR == 4)>(!full, tile, vbase, voffB, jt, aoffm, kb + PH_, kp_clamp, bq, acc); \
    }
    BOGP_SMALL_BLOCK(0)
    BOGP_SMALL_BLOCK(1)
    BOGP_SMALL_BLOCK(2)
    BOGP_SMALL_BLOCK(3)
#undef BOGP_SMALL_BLOCK
  }

  // ---- epilogue ---------------------------------------------------------------------------------------------------
  BOGP_SM_DRAIN();
  // The MFMAs are inline asm: to the compiler their results are ready at once, and the drain above (no register operands) does
  // not keep the epilogue's reads of the accumulators behind it.  With ONE tile per wave the scheduler did hoist the read of
  // the last-written component of the last accumulator above the drain (rows 60-63 of every workgroup wrong: found by
  // tests/test_gpu_driver.py at N = 100).  Naming every accumulator in an (empty) volatile asm behind the drain pins the order.
#pragma unroll
  for (int mi = 0; mi < SM_MR; ++mi)
#pragma unroll
    for (int ni = 0; ni < SM_NR; ++ni) asm volatile("" : "+v"(acc[mi][ni]));
  BOGP_STAMP(t_mfma);
  __syncthreads();
  BOGP_STAMP(t_pro);
  // D[i][j] of a 16 x 16 tile sits in lane 16 (i % 4) + j, register i / 4.  red[slot j][wave][row], row pitch 65: writes
  // and reads conflict free (as in k_contract16); then per (wave, row) the 16 slots in a fixed order, then the 8 waves
  constexpr int RP = 65;
  double* red = rs;                              // [16][8][RP]  (8 = the waves of the 8-wave schedule, real or virtual)
  double* red2 = rs + 16 * SM_WAVES * RP;        // [3][8][64]: ss, r.gamma, r.w per (virtual) wave
  {
    const int qd = lane >> 4, jc = lane & 15;
#pragma unroll
    for (int mi = 0; mi < SM_MR; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if constexpr (NW == 8) {
          double s = 0.0;
#pragma unroll
          for (int ni = 0; ni < SM_NR; ++ni)
            if (valid[ni]) s = __builtin_fma(acc[mi][ni][r], acc[mi][ni][r], s);
          red[(jc * SM_WAVES + w) * RP + 16 * mi + 4 * r + qd] = s;
        } else {  // tiles {w, 15 - w} = virtual wave w (ni 0, 3), {7 - w, 8 + w} = virtual wave 7 - w (ni 1, 2)
          double sa = 0.0, sb = 0.0;
          if (valid[0]) sa = __builtin_fma(acc[mi][0][r], acc[mi][0][r], sa);
          if (valid[3]) sa = __builtin_fma(acc[mi][3][r], acc[mi][3][r], sa);
          if (valid[1]) sb = __builtin_fma(acc[mi][1][r], acc[mi][1][r], sb);
          if (valid[2]) sb = __builtin_fma(acc[mi][2][r], acc[mi][2][r], sb);
          red[(jc * SM_WAVES + w) * RP + 16 * mi + 4 * r + qd] = sa;
          red[(jc * SM_WAVES + (7 - w)) * RP + 16 * mi + 4 * r + qd] = sb;
        }
      }
  }
  __syncthreads();
  if constexpr (NW == 8) {
    double s = 0.0;
#pragma unroll
    for (int sl = 0; sl < 16; ++sl) s += red[(sl * SM_WAVES + w) * RP + lane];
    red2[w * 64 + lane] = s;
    red2[(SM_WAVES + w) * 64 + lane] = mu;
    red2[(2 * SM_WAVES + w) * 64 + lane] = wd;
  } else {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int v = w + 4 * h;
      double s = 0.0;
#pragma unroll
      for (int sl = 0; sl < 16; ++sl) s += red[(sl * SM_WAVES + v) * RP + lane];
      red2[v * 64 + lane] = s;
      red2[(SM_WAVES + v) * 64 + lane] = h == 0 ? mu : mu1;
      red2[(2 * SM_WAVES + v) * 64 + lane] = h == 0 ? wd : wd1;
    }
  }
  __syncthreads();
  if (tid < 64) {
    double ss = 0.0, rg = 0.0, wr = 0.0;
#pragma unroll
    for (int ww = 0; ww < SM_WAVES; ++ww) {
      ss += red2[ww * 64 + tid];
      rg += red2[(SM_WAVES + ww) * 64 + tid];
      wr += red2[(2 * SM_WAVES + ww) * 64 + tid];
    }
    const int64_t gm = mg0 + tid;
    const bool ok = tid < MT && gm < a.M;
    double pm, pv;
    posterior_of_sums(rg, wr, a.need_var ? ss : 0.0, a.beta, a.G, a.estimate_trend, a.sigma2, pm, pv);
    if (ok && a.mu_out) a.mu_out[gm] = pm;
    if (ok && a.mse_out) a.mse_out[gm] = pv;
    const double y_hat = a.minimize ? pm : -1 * pm;
    const double sd = sqrt(pv);
    for (int c = 0; c < a.q; ++c) {
      double v = -INFINITY;
      int64_t idx = INT64_MAX;
      if (ok) {
        v = acq_value(a.acq_id[c], a.acq_par[c], y_hat, sd, a.plugin, a.sigma2);
        idx = gm;
        if (a.acq_out) a.acq_out[(size_t)c * a.M + gm] = v;
      }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        const double ov = shfl_xor_f64(v, off);
        const int64_t oi = shfl_xor_i64(idx, off);
        if (better(ov, oi, v, idx)) {
          v = ov;
          idx = oi;
        }
      }
      if (tid == 0) {
        a.blk_val[(size_t)c * a.nblk + a.blk_begin + blockIdx.x] = v;
        a.blk_idx[(size_t)c * a.nblk + a.blk_begin + blockIdx.x] = idx;
      }
    }
  }
  if (a.stamps && lane == 0) {
    const long long t_epi = clock64() - t_mark;
    atomicAdd((unsigned long long*)&a.stamps[0], (unsigned long long)t_prod);
    atomicAdd((unsigned long long*)&a.stamps[1], (unsigned long long)t_mfma);
    atomicAdd((unsigned long long*)&a.stamps[2], (unsigned long long)t_pro);
    atomicAdd((unsigned long long*)&a.stamps[3], (unsigned long long)t_epi);
    atomicAdd((unsigned long long*)&a.stamps[4], 1ull);
  }
#undef BOGP_STAMP
  if (a.q <= 0 || !a.final_launch) return;  // (a bulk launch followed by a tail launch leaves the reduce to the tail)
  // ---- the last workgroup to arrive reduces the per-block winners (deterministic: fixed scan order, index tie-break) ----
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned int ticket = atomicAdd(a.counter, 1u);
    s_last = ticket == gridDim.x - 1;
    if (s_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  if (!s_last) return;
  __shared__ double fv[NW];
  __shared__ int64_t fi[NW];
  const int64_t nslots = a.blk_begin + gridDim.x;  // slots written by the bulk launch (if any) + by this one
  for (int c = 0; c < a.q; ++c) {
    double v = -INFINITY;
    int64_t idx = INT64_MAX;
    for (int64_t k = tid; k < nslots; k += NT) {
      const double ov = __builtin_nontemporal_load(&a.blk_val[(size_t)c * a.nblk + k]);
      const int64_t oi = __builtin_nontemporal_load(&a.blk_idx[(size_t)c * a.nblk + k]);
      if (better(ov, oi, v, idx)) {
        v = ov;
        idx = oi;
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const double ov = shfl_xor_f64(v, off);
      const int64_t oi = shfl_xor_i64(idx, off);
      if (better(ov, oi, v, idx)) {
        v = ov;
        idx = oi;
      }
    }
    if (lane == 0) {
      fv[w] = v;
      fi[w] = idx;
    }
    __syncthreads();
    if (tid == 0) {
      for (int k = 1; k < NW; ++k)
        if (better(fv[k], fi[k], v, idx)) {
          v = fv[k];
          idx = fi[k];
        }
      a.best_val[c] = v;
      a.best_idx[c] = idx;
    }
    __syncthreads();
  }
  if (tid == 0) *a.counter = 0u;  // ready for the next launch on this stream
}

bool sweep_small_supported(int Np, int d, int kernel) {
  // generalized_exponential calls pow() per pair and dimension: inlined 16 times per producer trip it spills > 1000 VGPRs
  // next to the resident accumulators -- that kernel (values only, never fitted) keeps the chunked schedule
  if (kernel == BOGP_KERNEL_GENEXP || kernel == BOGP_KERNEL_MATERN_NU) return false;
  const char* e = getenv("BOGP_NO_FUSED_SMALL");  // read per call: the tests run both schedules in one process
  const bool off = e && atoi(e) != 0;
  return !off && Np <= 2 * SM_PANEL && d <= 60;  // LDS: 128 KB panel + 64 x roundup(d, 2) doubles <= 160 KB
}

// How a sweep over M candidates is cut into launches on a device with n_cu compute units: `bulk` 64-candidate workgroups
// (whole rounds of n_cu, or everything when the last round would be more than half full) and a tail of `tail_wg`
// workgroups of 16 * tail_mr candidates covering the rest in ONE short round.
void sweep_small_plan(int64_t M, int n_cu, int64_t* bulk, int* tail_mr, int64_t* tail_wg) {
  const int64_t g64 = (M + SM_MT - 1) / SM_MT;  // 64-candidate groups, the last one possibly ragged
  const int64_t rounds = g64 / n_cu;
  const int64_t rem_cand = M - rounds * n_cu * SM_MT;  // candidates beyond the complete rounds
  *bulk = g64;
  *tail_mr = 0;
  *tail_wg = 0;
  if (rounds == 0 || rem_cand <= 0) return;  // less than one round, or nothing left over
  // measured per round at N = 512, d = 10 (tools/small_mr_cost.py): 64 candidates per workgroup 107 us, 48: 94 us, 32: 79 us,
  // 16: 124 us (one accumulator per tile: every MFMA waits for its predecessor) -- so the tail uses 32 or 48, never 16
  const int64_t f16 = (rem_cand + 15) / 16;
  if ((f16 + 1) / 2 <= n_cu) {
    *bulk = rounds * n_cu; *tail_mr = 2; *tail_wg = (f16 + 1) / 2;
  } else if ((f16 + 2) / 3 <= n_cu) {
    *bulk = rounds * n_cu; *tail_mr = 3; *tail_wg = (f16 + 2) / 3;
  }
}

int64_t sweep_small_blocks(int64_t M, int n_cu) {
  int64_t bulk, tw;
  int tmr;
  sweep_small_plan(M, n_cu, &bulk, &tmr, &tw);
  return bulk + tw;
}

template <int MR, int NR, int NW>
static hipError_t launch_small_mr(int kernel, const SmallArgs& a, unsigned nwg, hipStream_t st) {
  // the panel + the candidate tile; the epilogue's reduction arrays ([16][8][65] + [3][8][64] doubles = 78 848 B) overlay them
  const size_t shm = std::max(((size_t)(32 * NW) * SM_MT + (size_t)SM_MT * ((a.d + 1) & ~1)) * sizeof(double),
                              (size_t)(16 * SM_WAVES * 65 + 3 * SM_WAVES * 64) * sizeof(double));
#define BOGP_LAUNCH_SMALL(K)                                                                                             \
  do {                                                                                                                   \
    hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sweep_small<K, MR, NR, NW>),                    \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);                           \
    if (e_ != hipSuccess) return e_;                                                                                     \
    hipLaunchKernelGGL((k_sweep_small<K, MR, NR, NW>), dim3(nwg), 64 * NW, shm, st, a.Xs, a.sqrt_theta, a.XthT, a.gamma, a.wvec, \
                       a.Vp, a);                                                                                         \
  } while (0)
  switch (kernel) {
    case BOGP_KERNEL_SE: BOGP_LAUNCH_SMALL(BOGP_KERNEL_SE); break;
    case BOGP_KERNEL_MATERN12: BOGP_LAUNCH_SMALL(BOGP_KERNEL_MATERN12); break;
    case BOGP_KERNEL_MATERN32: BOGP_LAUNCH_SMALL(BOGP_KERNEL_MATERN32); break;
    case BOGP_KERNEL_ABSEXP: BOGP_LAUNCH_SMALL(BOGP_KERNEL_ABSEXP); break;
    case BOGP_KERNEL_CUBIC: BOGP_LAUNCH_SMALL(BOGP_KERNEL_CUBIC); break;
    case BOGP_KERNEL_GENEXP: return hipErrorInvalidValue;  // sweep_small_supported() excludes it
    default: BOGP_LAUNCH_SMALL(BOGP_KERNEL_MATERN52); break;
  }
#undef BOGP_LAUNCH_SMALL
  return hipGetLastError();
}

// Four-wave workgroups, two per CU (k_sweep_small's NW = 4): Np <= 256, and a candidate tile of at most 16 KB (d <= 32) so that
// two 80-KB workgroups fit the CU's 160 KB of LDS.  BOGP_SMALL_NW=8 forces the one-workgroup-per-CU schedule for A/B runs.
static bool small_four_waves(int Np, int d) {
  const char* e = getenv("BOGP_SMALL_NW");
  return Np <= 256 && d <= 32 && !(e && atoi(e) == 8);
}

// Tiles per wave by training-set size: two for Np <= 256 (measured +7 % at N = 256, profiles/r03_sweep_scaling.txt; one tile per
// wave with a 16-slot ring was built too and measured the same as two at N = 128 -- below ~200 training points the kernel is
// bound by the latencies of a workgroup that has the CU to itself, not by the contraction -- so it is not instantiated).  The
// results are bit-identical whichever is taken: same tiles, same k order, same summation order in the epilogue
// (test_fused_small_sweep_tiles_per_wave_variants_are_bit_identical); BOGP_SMALL_NR=4 forces the r02 schedule for A/B runs.
static int small_nr(int Np) {
  const char* e = getenv("BOGP_SMALL_NR");
  return (Np <= 256 && !(e && atoi(e) == 4)) ? 2 : 4;
}
template <int MR>
static hipError_t launch_small_nr(int kernel, const SmallArgs& a, unsigned nwg, hipStream_t st) {
  if (small_nr(a.Np) == 2) return launch_small_mr<MR, 2, 8>(kernel, a, nwg, st);
  return launch_small_mr<MR, 4, 8>(kernel, a, nwg, st);
}

// `a` describes the whole sweep (M candidates, a.nblk = sweep_small_blocks(M, n_cu) partial-argmax slots)
hipError_t launch_sweep_small(int kernel, const SmallArgs& a0, int n_cu, hipStream_t st) {
  int64_t bulk, tail_wg;
  int tail_mr;
  sweep_small_plan(a0.M, n_cu, &bulk, &tail_mr, &tail_wg);
  SmallArgs a = a0;
  a.m_begin = 0;
  a.blk_begin = 0;
  // (with fewer workgroups than CUs nobody shares a CU and the eight-wave workgroup is the shorter one: 87 vs 100 us at N = 256, M = 1e4)
  const int64_t g64 = (a0.M + SM_MT - 1) / SM_MT;
  if (small_four_waves(a0.Np, a0.d) && g64 > n_cu) {  // one launch of 64-candidate workgroups (two per CU: no tail planning)
    if (g64 > a0.nblk) return hipErrorInvalidValue;
    a.final_launch = 1;
    return launch_small_mr<4, 4, 4>(kernel, a, (unsigned)g64, st);
  }
  a.final_launch = tail_wg == 0;
  hipError_t e = launch_small_nr<4>(kernel, a, (unsigned)bulk, st);
  if (e != hipSuccess || tail_wg == 0) return e;
  a.m_begin = bulk * SM_MT;
  a.blk_begin = bulk;
  a.final_launch = 1;
  return tail_mr == 3 ? launch_small_nr<3>(kernel, a, (unsigned)tail_wg, st) : launch_small_nr<2>(kernel, a, (unsigned)tail_wg, st);
}

}  // namespace bogp
