// Small dense GEMMs on one 64-lane wavefront with the f64 matrix cores (v_mfma_f64_16x16x4_f64), operands in LDS.
#pragma once
#include <type_traits>
#include "hb_math.hpp"

namespace hb {

// ---- small dense GEMMs on one wavefront ---------------------------------------------------------------------
// A block of MT x NT accumulator tiles (16 x 16 each) of v_mfma_f64_16x16x4_f64: A/B fragments are one f64 per lane
// (A[i = l&15][k = l>>4], B[k = l>>4][j = l&15]), accumulator rows (l>>4) + 4 r, column l&15
// (cdna_hip_programming.md §3, f64 layout).  All tiles advance together over K, so consecutive MFMAs are independent
// and each A/B fragment is read once per K-step.  The host build (tests/host_emu only) uses plain loops.
template <int MT, int NT>
struct WaveTile {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef double d4 __attribute__((ext_vector_type(4)));
  d4 acc[MT][NT];
#else
  double c[MT * 16][NT * 16];
#endif
};
template <int MT, int NT, class Ctx, class FC>
HB_HD void tile_init(const Ctx& cx, WaveTile<MT, NT>& t, int Mr, int Nr, FC c_init) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int li = cx.lane & 15, lk = cx.lane >> 4;
#pragma unroll
  for (int tm = 0; tm < MT; ++tm)
#pragma unroll
    for (int tn = 0; tn < NT; ++tn)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // c_init is evaluated for EVERY element, at indices clamped into the tile's live range, and the result is selected afterwards:
        // evaluated under the range test, the four elements of a lane were four branches with an LDS round trip each (c_init must
        // be free of side effects and must not branch around its own loads either — callers clamp and select the same way)
        const int row = 16 * tm + lk + 4 * r, col = 16 * tn + li;
        const double v = c_init(row < Mr ? row : Mr - 1, col < Nr ? col : Nr - 1);
        t.acc[tm][tn][r] = (row < Mr && col < Nr) ? v : 0.0;
      }
#else
  for (int i = 0; i < MT * 16; ++i)
    for (int j = 0; j < NT * 16; ++j) t.c[i][j] = (i < Mr && j < Nr) ? c_init(i, j) : 0.0;
#endif
}
// Accumulator start values from a row-major source, t(row, col) = src[row * LD + col] inside Mr x Nr and 0 outside.  Device: the four
// elements of a lane are requested unconditionally (one per-lane base + compile-time offsets, one wait) and selected afterwards — the
// lambda form above reads each element inside its own branch, four LDS round trips in a row.  The words of a whole 16-row tile behind
// `src` must be mapped (they are: every caller's source sits inside a larger LDS block).
template <int LD, int MT, int NT, class Ctx>
HB_HD void tile_init_rm(const Ctx& cx, WaveTile<MT, NT>& t, int Mr, int Nr, const double* src) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int li = cx.lane & 15, lk = cx.lane >> 4;
  const double* p0 = src + lk * LD + li;
#pragma unroll
  for (int tm = 0; tm < MT; ++tm)
#pragma unroll
    for (int tn = 0; tn < NT; ++tn) {
      double v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = p0[(16 * tm + 4 * r) * LD + 16 * tn];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * tm + lk + 4 * r, col = 16 * tn + li;
        t.acc[tm][tn][r] = (row < Mr && col < Nr) ? v[r] : 0.0;
      }
    }
#else
  (void)cx;
  for (int i = 0; i < MT * 16; ++i)
    for (int j = 0; j < NT * 16; ++j) t.c[i][j] = (i < Mr && j < Nr) ? src[i * LD + j] : 0.0;
#endif
}
// Accumulator start values that are zero except in ONE column: t(row, csel) = vec[row] for row < Mr.  (vec must be mapped up to
// element 16 MT - 1.)
template <int MT, int NT, class Ctx>
HB_HD void tile_init_col(const Ctx& cx, WaveTile<MT, NT>& t, int Mr, int csel, const double* vec) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int li = cx.lane & 15, lk = cx.lane >> 4;
#pragma unroll
  for (int tm = 0; tm < MT; ++tm) {
    double v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = vec[16 * tm + lk + 4 * r];
#pragma unroll
    for (int tn = 0; tn < NT; ++tn)
#pragma unroll
      for (int r = 0; r < 4; ++r) t.acc[tm][tn][r] = (16 * tm + lk + 4 * r < Mr && 16 * tn + li == csel) ? v[r] : 0.0;
  }
#else
  (void)cx;
  for (int i = 0; i < MT * 16; ++i)
    for (int j = 0; j < NT * 16; ++j) t.c[i][j] = (i < Mr && j == csel) ? vec[i] : 0.0;
#endif
}
// t += A B over K.  A(i,k) = TA ? A[k*LDA + i] : A[i*LDA + k],  B(k,j) = TB ? B[j*LDB + k] : B[k*LDB + j].
// K is padded to a multiple of 4; KR <= K is the real depth.  With KR == K there are no masks at all: the operands
// must then be zero-padded in k on both sides, and out-of-range rows / columns only feed discarded outputs.  With
// KR < K the fragments of the last step are selected to zero beyond KR (the memory behind them only has to be mapped).
struct NoScale { HB_HD double operator()(int) const { return 1.0; } };
struct AllSteps { HB_HD bool operator()(int) const { return true; } };
// `wk(k)` is an optional weight of the k-th term (diagonal scaling between A and B), e.g. a 0/1 row mask.  `step_live(j)` says
// whether K-step j (terms 4j .. 4j+3) contributes at all: it must be wave-uniform, and a step it rules out must have zero weights.
// PRE: request the operands of ALL K-steps before the first matrix instruction (K / 4 x (MT + NT) doubles in registers): the product is then
// one LDS latency + K / 4 dependent matrix instructions instead of K / 8 round trips — for the sweeps of small batches, whose stage is a
// latency chain (k_ric_bwd4: GEMM 1 of a stage 2 240 cycles before, cycle-counter trace of the ablation build).
template <int K, int LDA, bool TA, int LDB, bool TB = false, int KR = K, bool PRE = false, int MT, int NT, class Ctx, class FW = NoScale,
          class FL = AllSteps>
HB_HD void tile_mma(const Ctx& cx, WaveTile<MT, NT>& t, const double* A, const double* B, int Mr, int Nr, FW wk = FW(), FL step_live = FL()) {
  constexpr bool weighted = !std::is_same<FW, NoScale>::value;
#if defined(__HIP_DEVICE_COMPILE__)
  (void)Mr; (void)Nr;
  const int li = cx.lane & 15, lk = cx.lane >> 4;
  const double* ap = A + (TA ? lk * LDA + li : li * LDA + lk);
  const double* bp = B + (TB ? li * LDB + lk : lk * LDB + li);
  if constexpr (PRE) {
    static_assert(!weighted && std::is_same<FL, AllSteps>::value, "plain products only");
    constexpr int KS = (K + 3) / 4;
    double av[KS][MT], bv[KS][NT];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k0 = 4 * s;
      const bool live = (k0 + 4 <= KR) || (k0 + lk < KR);
#pragma unroll
      for (int tm = 0; tm < MT; ++tm) {
        av[s][tm] = ap[TA ? k0 * LDA + 16 * tm : 16 * tm * LDA + k0];
        if (k0 + 4 > KR) av[s][tm] = live ? av[s][tm] : 0.0;
      }
#pragma unroll
      for (int tn = 0; tn < NT; ++tn) {
        bv[s][tn] = bp[TB ? 16 * tn * LDB + k0 : k0 * LDB + 16 * tn];
        if (k0 + 4 > KR) bv[s][tn] = live ? bv[s][tn] : 0.0;
      }
    }
    asm volatile("" ::: "memory");   // (the reads stay in front of the matrix instructions)
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int tm = 0; tm < MT; ++tm)
#pragma unroll
        for (int tn = 0; tn < NT; ++tn) t.acc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s][tm], bv[s][tn], t.acc[tm][tn], 0, 0, 0);
    return;
  }
#pragma unroll
  for (int k0 = 0; k0 < K; k0 += 4) {
    if (!step_live(k0 / 4)) continue;
    double av[MT], bv[NT];
    const bool live = (k0 + 4 <= KR) || (k0 + lk < KR);
    double w = 1.0;
    if (weighted) w = wk(k0 + lk);
#pragma unroll
    for (int tm = 0; tm < MT; ++tm) {
      av[tm] = ap[TA ? k0 * LDA + 16 * tm : 16 * tm * LDA + k0];
      if (k0 + 4 > KR) av[tm] = live ? av[tm] : 0.0;
      if (weighted) av[tm] *= w;
    }
#pragma unroll
    for (int tn = 0; tn < NT; ++tn) {
      bv[tn] = bp[TB ? 16 * tn * LDB + k0 : k0 * LDB + 16 * tn];
      if (k0 + 4 > KR) bv[tn] = live ? bv[tn] : 0.0;
    }
#pragma unroll
    for (int tm = 0; tm < MT; ++tm)
#pragma unroll
      for (int tn = 0; tn < NT; ++tn) t.acc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[tm], bv[tn], t.acc[tm][tn], 0, 0, 0);
  }
#else
  (void)cx;
  for (int i = 0; i < Mr; ++i)
    for (int j = 0; j < Nr; ++j) {
      double acc = t.c[i][j];
      for (int k = 0; k < KR; ++k)
        if (step_live(k / 4)) acc += (weighted ? wk(k) : 1.0) * (TA ? A[k * LDA + i] : A[i * LDA + k]) * (TB ? B[j * LDB + k] : B[k * LDB + j]);
      t.c[i][j] = acc;
    }
#endif
}
// t += A Bt over K with the right operand taken from the ACCUMULATORS of an earlier product: B(k, j) = Bt(k, 16 tnb0 + j).  The accumulator
// layout (rows lk + 4 r of a 16-row tile, column li) IS the B-fragment layout of K-step 4 tm + r (k = 4 s + lk), so the fragments are the
// registers themselves: no LDS store of the earlier product, no operand loads, same terms in the same order as the product out of LDS
// (bit-identical).  A as in tile_mma; K a multiple of 4 with zero K-padding on the A side (rows of Bt beyond the real depth only have to be
// finite).  All A fragments are requested before the first matrix instruction.
template <int K, int LDA, bool TA, int MT, int NT, int MTB, int NTB, class Ctx>
HB_HD void tile_mma_bacc(const Ctx& cx, WaveTile<MT, NT>& t, const double* A, const WaveTile<MTB, NTB>& Bt, int tnb0, int Mr, int Nr) {
  static_assert(K % 4 == 0 && K <= 16 * MTB, "the contraction runs over the rows of Bt");
#if defined(__HIP_DEVICE_COMPILE__)
  (void)Mr; (void)Nr;
  const int li = cx.lane & 15, lk = cx.lane >> 4;
  const double* ap = A + (TA ? lk * LDA + li : li * LDA + lk);
  constexpr int KS = K / 4;
  double av[KS][MT];
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int tm = 0; tm < MT; ++tm) av[s][tm] = ap[TA ? 4 * s * LDA + 16 * tm : 16 * tm * LDA + 4 * s];
  asm volatile("" ::: "memory");   // (the reads stay in front of the matrix instructions)
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int tm = 0; tm < MT; ++tm)
#pragma unroll
      for (int tn = 0; tn < NT; ++tn)
        t.acc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s][tm], Bt.acc[s / 4][tnb0 + tn][s % 4], t.acc[tm][tn], 0, 0, 0);
#else
  (void)cx;
  for (int i = 0; i < Mr; ++i)
    for (int j = 0; j < Nr; ++j) {
      double acc = t.c[i][j];
      for (int k = 0; k < K; ++k) acc += (TA ? A[k * LDA + i] : A[i * LDA + k]) * Bt.c[k][16 * tnb0 + j];
      t.c[i][j] = acc;
    }
#endif
}
// t += o, element by element
template <int MT, int NT>
HB_HD void tile_add(WaveTile<MT, NT>& t, const WaveTile<MT, NT>& o) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  for (int tm = 0; tm < MT; ++tm)
#pragma unroll
    for (int tn = 0; tn < NT; ++tn) t.acc[tm][tn] += o.acc[tm][tn];
#else
  for (int i = 0; i < MT * 16; ++i)
    for (int j = 0; j < NT * 16; ++j) t.c[i][j] += o.c[i][j];
#endif
}
template <int MT, int NT, class Ctx, class FS>
HB_HD void tile_store(const Ctx& cx, const WaveTile<MT, NT>& t, int Mr, int Nr, FS store) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int li = cx.lane & 15, lk = cx.lane >> 4;
#pragma unroll
  for (int tm = 0; tm < MT; ++tm)
#pragma unroll
    for (int tn = 0; tn < NT; ++tn)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * tm + lk + 4 * r, col = 16 * tn + li;
        if (row < Mr && col < Nr) store(row, col, t.acc[tm][tn][r]);
      }
#else
  (void)cx;
  for (int i = 0; i < Mr; ++i)
    for (int j = 0; j < Nr; ++j) store(i, j, t.c[i][j]);
#endif
}

// Store with operands: `pre(row, col)` reads what the store of element (row, col) needs besides the accumulator — evaluated for every
// element at clamped indices, all loads of a lane in flight together — and `store(row, col, acc, pre)` runs inside the range test.
template <int MT, int NT, class Ctx, class FP, class FS>
HB_HD void tile_store_pre(const Ctx& cx, const WaveTile<MT, NT>& t, int Mr, int Nr, FP pre, FS store) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int li = cx.lane & 15, lk = cx.lane >> 4;
#pragma unroll
  for (int tm = 0; tm < MT; ++tm)
#pragma unroll
    for (int tn = 0; tn < NT; ++tn) {
      double p[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * tm + lk + 4 * r, col = 16 * tn + li;
        p[r] = pre(row < Mr ? row : Mr - 1, col < Nr ? col : Nr - 1);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * tm + lk + 4 * r, col = 16 * tn + li;
        if (row < Mr && col < Nr) store(row, col, t.acc[tm][tn][r], p[r]);
      }
    }
#else
  (void)cx;
  for (int i = 0; i < Mr; ++i)
    for (int j = 0; j < Nr; ++j) store(i, j, t.c[i][j], pre(i, j));
#endif
}

// Store into a plain row-major destination dst[row * LD + col] (times `scale`): the address of every element is one
// per-lane base plus a compile-time offset, so the store carries an immediate instead of rebuilding a 64-bit address
// per element as the generic lambda form does.
template <int LD, int MT, int NT, class Ctx>
HB_HD void tile_store_rm(const Ctx& cx, const WaveTile<MT, NT>& t, int Mr, int Nr, double* dst, double scale = 1.0) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int li = cx.lane & 15, lk = cx.lane >> 4;
  double* p0 = dst + lk * LD + li;
#pragma unroll
  for (int tm = 0; tm < MT; ++tm)
#pragma unroll
    for (int tn = 0; tn < NT; ++tn)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * tm + lk + 4 * r, col = 16 * tn + li;
        if (row < Mr && col < Nr) p0[(16 * tm + 4 * r) * LD + 16 * tn] = scale * t.acc[tm][tn][r];
      }
#else
  (void)cx;
  for (int i = 0; i < Mr; ++i)
    for (int j = 0; j < Nr; ++j) dst[i * LD + j] = scale * t.c[i][j];
#endif
}

// The same for a window of columns [c0, c1) of the tile (dst is the address of element (0, 0) of the tile in the destination, so a
// caller that wants the window moved sideways passes a shifted base).
template <int LD, int MT, int NT, class Ctx>
HB_HD void tile_store_rm_cols(const Ctx& cx, const WaveTile<MT, NT>& t, int Mr, int c0, int c1, double* dst, double scale = 1.0) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int li = cx.lane & 15, lk = cx.lane >> 4;
  double* p0 = dst + lk * LD + li;
#pragma unroll
  for (int tm = 0; tm < MT; ++tm)
#pragma unroll
    for (int tn = 0; tn < NT; ++tn)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * tm + lk + 4 * r, col = 16 * tn + li;
        if (row < Mr && col >= c0 && col < c1) p0[(16 * tm + 4 * r) * LD + 16 * tn] = scale * t.acc[tm][tn][r];
      }
#else
  (void)cx;
  for (int i = 0; i < Mr; ++i)
    for (int j = c0; j < c1; ++j) dst[i * LD + j] = scale * t.c[i][j];
#endif
}
// Overwrite column `col` of the tile: t(row, col) = f(row) for row < Mr.
template <int MT, int NT, class Ctx, class FV>
HB_HD void tile_set_col(const Ctx& cx, WaveTile<MT, NT>& t, int col, int Mr, FV f) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int li = cx.lane & 15, lk = cx.lane >> 4;
#pragma unroll
  for (int tm = 0; tm < MT; ++tm)
#pragma unroll
    for (int tn = 0; tn < NT; ++tn)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * tm + lk + 4 * r;
        const double v = f(row < Mr ? row : Mr - 1);
        if (16 * tn + li == col && row < Mr) t.acc[tm][tn][r] = v;
      }
#else
  (void)cx;
  for (int i = 0; i < Mr; ++i) t.c[i][col] = f(i);
#endif
}

}  // namespace hb
