// Backward / forward Riccati sweep over the projected stage records of one robot instance, one 64-lane
// workgroup per instance with the value function and the current stage resident in LDS.  This is the role HPIPM
// plays inside OCS2's SqpSolver for the reference (no inequality rows reach the QP in the shipped configuration,
// SURVEY.md B.5), followed by the node-level evaluation used by the filter line search.
#pragma once
#include "hb_lq.hpp"

namespace hb {

// LDS of the backward sweep: 2496 doubles = 19 968 B per instance -> 8 single-wave workgroups per CU (two per SIMD).
// Every matrix is kept in a padded layout chosen so that each MFMA operand of the three GEMM groups is "per-lane base +
// compile-time offset" (one address register per operand, offsets in the ds_read immediates) and so that the K-padding
// of the 16x16x4 tiles is zeros on both sides:
//   wide rows  (stride 36): [x block (22) | vector (1) | u block (12) | unused (1)]    [A~ b~ B~ .], M1, [P~ r~ R~ .], Hu
//   (the u block starts at column 23 so that x block + vector + up to 9 projected inputs fit two 16-column tiles)
//   narrow rows (stride 24): [x block (22) | vector (1) | zero (1)]                     S (cols 22,23 zero), T, [K~ k~ .]
// ABb and M1 carry two extra all-zero rows (K = 22 -> 24).  Buffers with disjoint lifetimes share storage (the
// workgroup is one wave: its LDS accesses complete in program order):
//   X   S | s (node start .. GEMM 1)  ->  M1 (GEMM 1 .. GEMM 3)  ->  T = new S | s (written by GEMM 3, symmetrised in place)
//   PH  [P~ r~ R~ .] (staged .. accumulator init of GEMM 2)  ->  Hu
// [Q~ q~] has no buffer of its own: it is dropped over the dead A~ block once GEMM 3 has read its operands.
struct RicLds {
  static constexpr int LDN = 24, LDW = 36, CV = 22, CU = 23;
  static constexpr int X = 0;                    // 24 x 36
  static constexpr int S = X;                    // 22 x 24
  static constexpr int s = S + 22 * LDN;         // 24
  static constexpr int ABb = X + 24 * LDW;       // 24 x 36
  static constexpr int PRr = ABb + 24 * LDW;     // 12 x 36
  static constexpr int Hu = PRr;                 // 12 x 36 : [Hux | hu | . | Huu]
  static constexpr int Kk = PRr + 12 * LDW;      // 12 x 24 : [K~ | k~ | .]
  static constexpr int Qs = ABb;                 // [Q~ upper triangle, packed | q~] 276, dropped over A~ between GEMM 3 and the store of T
  static constexpr int flag = Kk + 12 * LDN;     // 4
  static constexpr int total = flag + 4 + 44;    // slack: padded tile reads run up to 40 doubles past Kk
};
static_assert(RicLds::LDW == REC_LD && RicLds::CV == REC_CV && RicLds::CU == REC_CU, "the record is the LDS image");
static_assert(RicLds::PRr == RicLds::ABb + 24 * RicLds::LDW && REC_PR == REC_AB + 22 * REC_LD, "two straight copies");
static_assert(RicLds::total * 8 <= 20480, "k_ric_bwd: LDS per instance must allow 8 workgroups per CU");

// LDS of the FOUR-wavefront form of the backward sweep (k_ric_bwd4: small batches, one wavefront per SIMD of a CU working on one
// instance).  Same row formats as RicLds, but every buffer has its own storage: with four wavefronts a buffer that is reused inside
// a stage needs a workgroup barrier between its last reader and its next writer, and at the batch sizes this form is for (at most
// four instances per CU) LDS is not what limits residency.  30.8 KB per instance.
struct Ric4Lds {
  static constexpr int LDN = RicLds::LDN, LDW = RicLds::LDW, CV = RicLds::CV, CU = RicLds::CU;
  static constexpr int S = 0;                    // 22 x 24 (columns 22, 23 stay zero: K-padding)
  static constexpr int s = S + 22 * LDN;         // 24
  static constexpr int M1 = s + 24;              // 24 x 36 (rows 22, 23 stay zero)
  static constexpr int ABb = M1 + 24 * LDW;      // 24 x 36 (rows 22, 23 stay zero)
  static constexpr int PRr = ABb + 24 * LDW;     // 12 x 36, then Hu
  static constexpr int Hu = PRr;
  static constexpr int Kk = PRr + 12 * LDW;      // 12 x 24
  static constexpr int Qs = Kk + 12 * LDN;       // [Q~ upper triangle, packed | q~] 276 (the buffer keeps its 512)
  static constexpr int flag = Qs + 512;          // 4
  static constexpr int total = flag + 4 + 44;    // slack for the padded tile reads, as RicLds
};
static_assert(Ric4Lds::total * 8 <= 40960, "k_ric_bwd4: four instances per CU");

// One backward step on the staged record.  Updates S, s in place; writes the gains.
//   M1 = S [A~ b~ B~] (+ s),  Hu = B~' M1 + [P~ r~ R~],  K~ = -Huu^-1 [Hux hu],
//   S <- sym(Q~ + A~' M1_A + Hux' K~),  s <- q~ + A~' M1_b + Hux' k~          
// NTW = number of 16-column tiles of the wide operands that are formed: 3 in general, 2 when the stage has at most 9
// projected inputs (columns 0..31 = x block, vector, inputs 0..8): GEMM 1 24 instead of 36 MFMAs, GEMM 2 12 instead of 18.
// M1 STAYS IN THE ACCUMULATORS of GEMM 1 (round 6): the accumulator layout of a tile is the right-operand fragment layout of the products
// that contract over its rows (hb_tile.hpp tile_mma_bacc), so GEMM 2 (B~' M1) and the A~' M1 part of GEMM 3 take M1 straight from the
// registers — no store of M1 to LDS, no ordering point behind it, no operand loads of M1 (46 LDS instructions and two LDS round trips of a
// stage), same terms in the same order.  S | s stay untouched until the new S is stored.
template <int NTW, class Ctx>
HB_HD void ric_phase1(const Ctx& cx, double* lds, WaveTile<2, NTW>& t) {
  const double* sv = lds + RicLds::s;
  constexpr int NC = NTW * 16 < RicLds::LDW ? NTW * 16 : RicLds::LDW;
  // (start values and, in GEMM 2 / 3, all operands of a product requested together: since the build dropped machine LICM the sweep has
  // the registers for it; 1.70 -> 1.66 ms per 4096 x 100, 2.57 -> 2.50 standing; bit-identical)
  tile_init_col(cx, t, 22, RicLds::CV, sv);
  // S is EXACTLY symmetric (the previous stage stored the upper triangle of T mirrored; S = 0 at the end of the horizon), so the left operand
  // is read transposed, S(i, k) as S(k, i): 16 lanes then read 16 consecutive doubles instead of 16 doubles 24 apart (a four-way
  // bank conflict on every operand read).  The K-padding rows 22 / 23 of this view are s and a zero row; the zero rows 22 / 23
  // of [A~ b~ B~] cancel them.  Rows 22 / 23 of M1 itself come out exactly zero (columns 22 / 23 of S are the zero K-padding).
  tile_mma<24, RicLds::LDN, true, RicLds::LDW, false, 24, false>(cx, t, lds + RicLds::S, lds + RicLds::ABb, 22, NC);
}
// Factorisation of the leading NT x NT block of Huu and the 23 solves.  Every lane factors the (uniform) block redundantly in
// registers, lane c < 23 then solves its own right-hand side.  NT = 9 serves every stage with at most 9 projected inputs (single
// support: 6 contact-force + 3 kernel coordinates; flight: 6) — the padding rows of the record are R~ = I, B~ = 0, P~ = 0, r~ = 0,
// so their gains are exactly zero and the factor work drops by ~(9/12)^3.
//
// The factorisation is ROOT-FREE:  Huu = U D U',  U unit lower triangular, kept as U (strictly lower part) and r = 1 / d (diagonal).
// What a stage of the sweep costs on a small batch is the length of this dependency chain (one wavefront per SIMD: 512 instances
// spent 0.16 of k_ric_bwd4's 0.44 ms in the 9 x 9 LL' factorisation, 415 cycles per pivot through d > 0 ? -> select -> v_rsq_f64 ->
// refinement -> class select -> column scaling -> next diagonal).  Per pivot the chain is now
//     r_j = 1 / d_j (v_rcp_f64 + one cubic correction, 4 levels) -> select on d_j > 0 -> d_(j+1) -= t^2 r_j  (one FMA)
// — six levels instead of twelve — and the substitutions have no multiplication on their chains (unit diagonal):
//     z = U^-1 b,  w = r z,  y = U'^-1 w.
// A pivot that is not positive is replaced by 1 and reported (as before).
HB_HD double ric_rcp(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double r0 = __builtin_amdgcn_rcp(d);   // ~ 2^-23 relative
  const double e = fma(-d, r0, 1.0);
  return fma(r0, fma(e, e, e), r0);            // r0 (1 + e + e^2): error e^3
#else
  return 1.0 / d;
#endif
}
// U D U' of the leading NB x NB block of Huu, redundantly per lane in registers: on return L holds U below the diagonal and the
// reciprocals r on it.  Unless KEEP, lane 0 writes it back over the (dead) lower triangle of Huu.  Returns true when a pivot was
// not positive.
template <int NB, bool KEEP = false, class Ctx>
HB_HD bool ric_chol_block(const Ctx& cx, double* Hu, double (&L)[NB * (NB + 1) / 2]) {
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) L[i * (i + 1) / 2 + j] = Hu[i * RicLds::LDW + RicLds::CU + j];
  bool bad = false;
#pragma unroll
  for (int j = 0; j < NB; ++j) {   // right-looking: column j, then the rank-one update of what is left
    const double d = L[j * (j + 1) / 2 + j];
    const bool pos = d > 0.0;
    bad = bad || !pos;
    const double r = pos ? ric_rcp(d) : 1.0;
    L[j * (j + 1) / 2 + j] = r;
#pragma unroll
    for (int i = j + 1; i < NB; ++i) {   // row i: t = unnormalised entry (i, j); the rows above are already normalised (u)
      const double t = L[i * (i + 1) / 2 + j];
#pragma unroll
      for (int k = j + 1; k < i; ++k) L[i * (i + 1) / 2 + k] = fma(-t, L[k * (k + 1) / 2 + j], L[i * (i + 1) / 2 + k]);
      L[i * (i + 1) / 2 + i] = fma(-(t * t), r, L[i * (i + 1) / 2 + i]);   // the next pivots: one level behind r
      L[i * (i + 1) / 2 + j] = t * r;
    }
  }
  // The factor (lane-uniform) goes back to LDS over the lower triangle of Huu, which is dead from here on: the solves
  // then read it through broadcast loads and the registers are free again (explicit "spill" to LDS; a register
  // file that still held L here pushed the kernel's loop invariants into scratch memory).
  if (KEEP) return bad;  // the caller keeps the factor in registers (device, NB <= 9: no LDS round trip before the solves)
  if (cx.lane == 0) {
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) Hu[i * RicLds::LDW + RicLds::CU + j] = L[i * (i + 1) / 2 + j];
  }
  return bad;
}
// Rows 9..11 of the 12 x 12 factor against the leading block (U11, r1 in Lr): for each row the unnormalised entries t = U11^-1 a
// (unit lower: no division), the 3 x 3 Schur complement S = A22 - T21 diag(r1) T21' and its own U D U'.  Redundantly per lane;
// `rd(r, j)` reads entry j of row 9 + r of Huu.  Outputs: t21 (the rows BEFORE normalisation: u = t r1), S (U below, r on the diagonal).
template <class RD>
HB_HD bool ric_factor_tail(const double (&Lr)[45], RD rd, double (&t21)[3][9], double (&S)[6]) {
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      double sacc = rd(r, j);
#pragma unroll
      for (int k = 0; k < j; ++k) sacc = fma(-t21[r][k], Lr[j * (j + 1) / 2 + k], sacc);
      t21[r][j] = sacc;
    }
#pragma unroll
    for (int c = 0; c <= r; ++c) {
      double sacc = rd(r, 9 + c);
#pragma unroll
      for (int k = 0; k < 9; ++k) sacc = fma(-(t21[r][k] * t21[c][k]), Lr[k * (k + 1) / 2 + k], sacc);
      S[r * (r + 1) / 2 + c] = sacc;
    }
  }
  bool bad = false;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const double d = S[j * (j + 1) / 2 + j];
    const bool pos = d > 0.0;
    bad = bad || !pos;
    const double r = pos ? ric_rcp(d) : 1.0;
    S[j * (j + 1) / 2 + j] = r;
#pragma unroll
    for (int i = j + 1; i < 3; ++i) {
      const double t = S[i * (i + 1) / 2 + j];
#pragma unroll
      for (int k = j + 1; k < i; ++k) S[i * (i + 1) / 2 + k] = fma(-t, S[k * (k + 1) / 2 + j], S[i * (i + 1) / 2 + k]);
      S[i * (i + 1) / 2 + i] = fma(-(t * t), r, S[i * (i + 1) / 2 + i]);
      S[i * (i + 1) / 2 + j] = t * r;
    }
  }
  return bad;
}
// Hu: [Hux | hu | . | Huu] (12 rows of LDW), Kk: [K~ | k~ | .] (12 rows of LDN), flag: set to 1 when a pivot was not positive
template <int NT, class Ctx>
HB_HD void ric_factor_solve(const Ctx& cx, double* Hu, double* Kk, double* flag, double* gains) {
#if defined(__HIP_DEVICE_COMPILE__)
  // keep the loads of this width's triangle inside its branch: hoisted above the 9 / 12 dispatch they were spilled
  asm volatile("" ::: "memory");
#endif
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr bool reg_factor = true;   // (the factor never leaves the registers, either width)
#else
  constexpr bool reg_factor = false;
#endif
  constexpr int NB1 = NT <= 9 ? NT : 9;
  double Lr[NB1 * (NB1 + 1) / 2];
#if defined(__HIP_DEVICE_COMPILE__)
  double u21[3][9], S3[6];   // rows 9..11 of the 12 x 12 factor (NT == 12 only)
#endif
  if constexpr (NT <= 9) {
    const bool bad = ric_chol_block<NT, reg_factor>(cx, Hu, Lr);
    if (bad && cx.lane == 0) *flag = 1.0;
  } else {
    // 12 projected inputs (double support: 12 contact forces): blocked — the 9 x 9 leading block in registers as above, then
    // rows 9..11 against it and the 3 x 3 Schur complement (ric_factor_tail).
    static_assert(NT == 12, "blocked factorisation: 9 + 3");
#if defined(__HIP_DEVICE_COMPILE__)
    // Device: every lane continues redundantly in registers (the 9 x 9 block is still there).  (ric_chol_block<9, KEEP> leaves Huu
    // untouched: rows 9..11 are read from it.)
    bool bad = ric_chol_block<9, true>(cx, Hu, Lr);
    // rows 9..11 of the factor stay in registers as well (u = t r1 in place of t; 78 doubles with the leading block): the solves are
    // pure FMA chains for every row, no LDS round trip of the factor, no barrier.  (Rounds 3-5 parked these rows in LDS: with the record
    // prefetch held live across the stage by its `if (k > 0)` the register file had no room for them.)
    bad = ric_factor_tail(Lr, [Hu](int r, int j) { return Hu[(9 + r) * RicLds::LDW + RicLds::CU + j]; }, u21, S3) || bad;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int j = 0; j < 9; ++j) u21[r][j] *= Lr[j * (j + 1) / 2 + j];
#else
    // Host emulator (the lanes run one after the other): the same arithmetic from copies of the rows, then the whole factor to LDS
    bool bad = ric_chol_block<9, true>(cx, Hu, Lr);
    {
      double rows[3][12];
      for (int r = 0; r < 3; ++r)
        for (int j = 0; j < 12; ++j) rows[r][j] = Hu[(9 + r) * RicLds::LDW + RicLds::CU + j];
      double t21[3][9], S[6];
      bad = ric_factor_tail(Lr, [&rows](int r, int j) { return rows[r][j]; }, t21, S) || bad;
      cx.sync();
      if (cx.lane == 0) {
        for (int i = 0; i < 9; ++i)
          for (int j = 0; j <= i; ++j) Hu[i * RicLds::LDW + RicLds::CU + j] = Lr[i * (i + 1) / 2 + j];
        for (int r = 0; r < 3; ++r) {
          for (int j = 0; j < 9; ++j) Hu[(9 + r) * RicLds::LDW + RicLds::CU + j] = t21[r][j] * Lr[j * (j + 1) / 2 + j];
          for (int c = 0; c <= r; ++c) Hu[(9 + r) * RicLds::LDW + RicLds::CU + 9 + c] = S[r * (r + 1) / 2 + c];
        }
      }
    }
#endif
    if (bad && cx.lane == 0) *flag = 1.0;
  }
  if (!reg_factor) cx.sync();
  {
    const double* Lm = Hu + RicLds::CU;  // U(i, j) = Lm[i * LDW + j], j < i; the diagonal holds the reciprocals r
#if defined(__HIP_DEVICE_COMPILE__)
    // 12-wide (double support): the leading 9 x 9 block of the factor is still in this lane's registers (ric_chol_block left it
    // there), rows 9..11 in u21 / S3
    constexpr int NREG = NT <= 9 ? NT : 9;
#else
    constexpr int NREG = 0;
#endif
#if defined(__HIP_DEVICE_COMPILE__)
    auto Lf = [&Lr, Lm, &u21, &S3](int i, int j) {
      if (i < NREG) return Lr[i * (i + 1) / 2 + j];
      if constexpr (NT > 9) return j < 9 ? u21[i - 9][j] : S3[(i - 9) * (i - 8) / 2 + (j - 9)];
      return Lm[i * RicLds::LDW + j];
    };
#else
    auto Lf = [&Lr, Lm](int i, int j) {
      if (i < NREG) return Lr[i * (i + 1) / 2 + j];
      return Lm[i * RicLds::LDW + j];
    };
#endif
    for (int c = cx.lane; c < 23; c += cx.nlanes) {  // columns 0..21 = Hux, 22 = hu
      double y[NU_T];
#pragma unroll
      for (int a = 0; a < NT; ++a) {   // z = U^-1 (-h)
        double sacc = -Hu[a * RicLds::LDW + c];
#pragma unroll
        for (int k = 0; k < a; ++k) sacc = fma(-Lf(a, k), y[k], sacc);
        y[a] = sacc;
      }
#pragma unroll
      for (int a = NT - 1; a >= 0; --a) {   // y = U'^-1 (r z)
        double sacc = y[a] * Lf(a, a);
#pragma unroll
        for (int k = a + 1; k < NT; ++k) sacc = fma(-Lf(k, a), y[k], sacc);
        y[a] = sacc;
      }
#pragma unroll
      for (int a = NT; a < NU_T; ++a) y[a] = 0.0;
      double* gp = gains + (c < 22 ? c : 264);  // straight from the registers: lanes 0..21 write one row segment
      const int gs = c < 22 ? 22 : 1;
#pragma unroll
      for (int a = 0; a < NU_T; ++a) {
        Kk[a * RicLds::LDN + c] = y[a];
        gp[a * gs] = y[a];
      }
    }
  }
  cx.sync();
}
// `n_til` = number of projected inputs of the stage (contact-force + kernel coordinates, REC_META)
template <int NTW, class Ctx>
HB_HD void ric_phase2_gemm(const Ctx& cx, double* lds, const WaveTile<2, NTW>& m1) {
  const double* PRr = lds + RicLds::PRr;
  double* Hu = lds + RicLds::Hu;
  constexpr int NC = NTW * 16 < RicLds::LDW ? NTW * 16 : RicLds::LDW;
  WaveTile<1, NTW> t;
  tile_init_rm<RicLds::LDW>(cx, t, NU_T, NC, PRr);
  tile_mma_bacc<24, RicLds::LDW, true>(cx, t, lds + RicLds::ABb + RicLds::CU, m1, 0, NU_T, NC);
  tile_store(cx, t, NU_T, NC, [Hu](int a, int c, double v) { Hu[a * RicLds::LDW + c] = v; });  // over [P~ r~ R~]
  cx.sync();
}
// GEMM 1, GEMM 2 and the factorisation / solves of one stage
// `m1` receives the x block and the vector column of M1 (columns 0..31: what GEMM 3 contracts with)
template <class Ctx>
HB_HD void ric_phase12(const Ctx& cx, double* lds, double* gains, int n_til, WaveTile<2, 2>& m1, int dbg = 0) {
  static_assert(RicLds::CU + 9 <= 32, "x block, vector and 9 inputs must fit two tiles");
  if (n_til <= 9) {
    ric_phase1<2>(cx, lds, m1);
    HB_ABLATE_STOP(dbg == 21);
    ric_phase2_gemm<2>(cx, lds, m1);
    HB_ABLATE_STOP(dbg == 22);
    ric_factor_solve<9>(cx, lds + RicLds::Hu, lds + RicLds::Kk, lds + RicLds::flag, gains);
  } else {
    WaveTile<2, 3> m1w;
    ric_phase1<3>(cx, lds, m1w);
    HB_ABLATE_STOP(dbg == 21);
    ric_phase2_gemm<3>(cx, lds, m1w);
    HB_ABLATE_STOP(dbg == 22);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) m1.acc[tm][tn] = m1w.acc[tm][tn];
#else
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) m1.c[i][j] = m1w.c[i][j];
#endif
    ric_factor_solve<NU_T>(cx, lds + RicLds::Hu, lds + RicLds::Kk, lds + RicLds::flag, gains);
  }
}
// GEMM 3 in two halves: `ric_phase3_mma` accumulates T - [Q~ q~] = A~' [M1_A M1_b] + Hux' [K~ k~]; the caller then drops
// [Q~ (22 x 22 row-major) | q~ (22)] at RicLds::Qs — it has no buffer of its own and goes over the A~ block, dead once
// the operand reads are done — and `ric_phase3_finish` adds it and stores the new S | s.
// T is symmetric up to rounding, so only its upper block triangle is formed — tiles (0,0), (0,1) and (1,1): 27 MFMAs
// instead of 36 — and the new S is the upper triangle of T mirrored (exactly symmetric, no separate symmetrisation pass).
struct RicT3 {
  WaveTile<1, 2> t0;  // rows 0..15, columns 0..31
  WaveTile<1, 1> t1;  // rows 16..31, columns 16..31
};
template <class Ctx>
HB_HD void ric_phase3_mma(const Ctx& cx, double* lds, RicT3& t, const WaveTile<2, 2>& m1) {
  tile_init(cx, t.t0, 16, 23, [](int, int) { return 0.0; });
  tile_init(cx, t.t1, 6, 7, [](int, int) { return 0.0; });
  tile_mma_bacc<24, RicLds::LDW, true>(cx, t.t0, lds + RicLds::ABb, m1, 0, 16, 23);
  tile_mma_bacc<24, RicLds::LDW, true>(cx, t.t1, lds + RicLds::ABb + 16, m1, 1, 6, 7);
  tile_mma<NU_T, RicLds::LDW, true, RicLds::LDN, false, NU_T, true>(cx, t.t0, lds + RicLds::Hu, lds + RicLds::Kk, 16, 23);
  tile_mma<NU_T, RicLds::LDW, true, RicLds::LDN, false, NU_T, true>(cx, t.t1, lds + RicLds::Hu + 16, lds + RicLds::Kk + 16, 6, 7);
  cx.sync();
}
// Store of a block of T = new S | s (rows r0.., columns c0.. of the 22 x 23 result): element (i, c), i <= c < 22, goes to S(i, c) and S(c, i)
// with [Q~] added, column 22 is the vector s with q~ added.  Device: the (up to) four Q~ words of a lane are requested together, in front
// of the stores — element by element (each with its own branch, LDS read and wait) this store was 1 000 .. 1 650 cycles of a 8 700-cycle
// stage of k_ric_bwd4, more than the matrix instructions before it.
template <int LDN_, int MT, int NT, class Ctx>
HB_HD void ric_store_T(const Ctx& cx, const WaveTile<MT, NT>& t, int Mr, int Nr, int r0, int c0, double* S, double* sv, const double* Qs,
                       double* trash) {
#if defined(__HIP_DEVICE_COMPILE__)
  // Branch-free: a lane's column is the same for its four elements, the Q~ word of element r is (first word) + r x (4 or 88), and an element
  // that is not stored goes to `trash` (one dead LDS word) — no lane-mask arithmetic between the matrix instructions and the stores (written
  // with a branch per element, the mask juggling through scalar registers was ~ 600 of the stage's 8 700 cycles).
  const int li = cx.lane & 15, lk = cx.lane >> 4;
#pragma unroll
  for (int tm = 0; tm < MT; ++tm)
#pragma unroll
    for (int tn = 0; tn < NT; ++tn) {
      const int cl = 16 * tn + li, c = cl + c0, il0 = 16 * tm + lk, i0 = il0 + r0;
      const bool colv = c == RicLds::CV, colok = cl < Nr;
      double q[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {   // Qs = [Q~ upper triangle packed | q~] as in the record
        const int i = i0 + 4 * r;
        const int idx = colv ? REC_QT_PACKED + i : rec_Qidx(i, c);
        q[r] = Qs[(idx >= 0 && idx < REC_QT_PACKED + 22) ? idx : 0];
      }
      double* const pv = sv + i0;                  // + 4 r
      double* const pa = S + i0 * LDN_ + c;        // + 4 r LDN
      double* const pb = S + c * LDN_ + i0;        // + 4 r
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool in = colok && il0 + 4 * r < Mr;
        const bool mat = in && !colv && i0 + 4 * r <= c;
        const double w = t.acc[tm][tn][r] + q[r];
        double* a1 = mat ? pa + 4 * r * LDN_ : ((in && colv) ? pv + 4 * r : trash);
        double* a2 = mat ? pb + 4 * r : trash;
        *a1 = w;
        *a2 = w;
      }
    }
#else
  (void)trash;
  tile_store(cx, t, Mr, Nr, [S, sv, Qs, r0, c0](int il, int cl, double v) {
    const int i = il + r0, c = cl + c0;
    if (c == RicLds::CV) {
      sv[i] = v + Qs[REC_QT_PACKED + i];
    } else if (i <= c) {
      const double w = v + Qs[rec_Qidx(i, c)];
      S[i * LDN_ + c] = w;
      S[c * LDN_ + i] = w;
    }
  });
#endif
}
template <class Ctx>
HB_HD void ric_phase3_finish(const Ctx& cx, double* lds, const RicT3& t) {
  double* S = lds + RicLds::S;
  double* sv = lds + RicLds::s;
  const double* Qs = lds + RicLds::Qs;
  cx.sync();
  // The new S | s overwrite the old ones: every operand read of the GEMMs precedes these stores in the wave's program
  // order.  Element (i, c), i <= c < 22, goes to S(i, c) and S(c, i); column 22 is the vector s.
  ric_store_T<RicLds::LDN>(cx, t.t0, 16, 23, 0, 0, S, sv, Qs, lds + RicLds::flag + 2);
  ric_store_T<RicLds::LDN>(cx, t.t1, 6, 7, 16, 16, S, sv, Qs, lds + RicLds::flag + 2);
  // (columns 22, 23 of S — the zero K-padding — are never written: M1 no longer lives in LDS)
  cx.sync();
}
// Reference staging of one record (host emulation; the kernel batches its global loads instead).
template <class Ctx>
HB_HD void ric_stage(const Ctx& cx, double* lds, const double* rec) {
  for (int e = cx.lane; e < REC_QT; e += cx.nlanes) lds[e < REC_PR ? RicLds::ABb + e : RicLds::PRr + e - REC_PR] = rec[e];
  cx.sync();
}
template <class Ctx>
HB_HD void riccati_bwd_node(const Ctx& cx, double* lds, const double* rec, double* gains, int dbg = 0) {
  WaveTile<2, 2> m1;
  ric_phase12(cx, lds, gains, int(rec[REC_META]) + int(rec[REC_META + 1]), m1, dbg);
  HB_ABLATE_STOP(dbg == 21 || dbg == 22 || dbg == 23);  // profiling ablation markers (hb_config.reserved)
  RicT3 t;
  ric_phase3_mma(cx, lds, t, m1);
  double* Qs = lds + RicLds::Qs;
  for (int e = cx.lane; e < REC_QT_PACKED + 22; e += cx.nlanes) Qs[e] = rec[REC_QT + e];
  ric_phase3_finish(cx, lds, t);
}

struct FwdLds {
  static constexpr int dx = 0;        // 22 (+2: entries 22 / 23 stay 0.0 — the "zero slot" the padded terms of the wave form read)
  static constexpr int ut = 24;       // 12
  static constexpr int du = 36;       // 22 (+2)
  static constexpr int dxn = 60;      // 22 (+2)
  static constexpr int acc = 84;      // 4: armijo (filled by riccati_fwd_finish), merit, dyn, eq
  static constexpr int accp = 88;     // 22 (+2): per-entry partial sums of the Armijo directional derivative
  static constexpr int uz = 112;      // 6 (+2): u~ of the kernel directions, u~[n_f + b]
  static constexpr int pre = 120;     // 10 (+2): ke + Kx dx of the joint rows (formed next to u~ = k~ + K~ dx: both only need dx)
  static constexpr int small = 132;
  // staged copy of what one forward step reads (device kernel): [A~ b~ B~ .] rows | recovery data | gains
  static constexpr int AB = small;                 // rows 0..11 of [A~ b~ B~ .] (12 rows of 36)
  static constexpr int RX = AB + 12 * REC_LD;      // record elements [REC_KX, REC_RX_END)
  static constexpr int G = RX + (REC_RX_END - REC_KX);
  static constexpr int total = G + GAIN_SIZE;
};
static_assert((REC_RX_END - REC_KX) % 2 == 0 && (12 * REC_LD) % 2 == 0 && REC_KX % 2 == 0 && FwdLds::AB % 2 == 0, "16-byte staging");

// One forward step:   u~ = k~ + K~ dx,   dx+ = [A~ dx + B~ u~ + b~ ; joint rows q + dt qd],   du = recovered input step.
// Every dot product is split FOUR ways — partial q takes the entries q, q + 4, q + 8, ... and the partials are summed
// (p0 + p1) + (p2 + p3) — so that the wave form below can give each row to four lanes: with one wavefront per SIMD (512
// instances per GPU) the step is a chain of dependent LDS reads and FMAs, and its length is what the kernel costs.
//   phase A (needs dx):        u~ (12 rows) and pre = ke + Kx dx (10 rows): 22 rows x 22 terms
//   phase B (needs u~, pre):   dx+ rows 0..11: 34 terms over [dx ; u~];  joint rows: qd = pre + Z u~[n_f ..] (6 terms, Z zero-padded
//                              by k_lq), dx+ = dq + dx + dt qd
//   phase C:                   force rows of du (a copy of u~ or of the record), Armijo sums, outputs, dx <- dx+
// Host form (emulator): `ab` = rows 0..11 of [A~ b~ B~ .] of the stage record (stride REC_LD), `rx` = its recovery part (element
// REC_KX onwards), `gains` = [K~ | k~], all in ordinary memory.
HB_HD double fwd_sum4(const double* p) { return (p[0] + p[1]) + (p[2] + p[3]); }
template <class Ctx>
HB_HD void riccati_fwd_node(const Ctx& cx, double* lds, const double* ab, const double* rx, const double* gains, double* dx_out,
                            double* du_out) {
  double* dx = lds + FwdLds::dx;
  double* ut = lds + FwdLds::ut;
  double* du = lds + FwdLds::du;
  double* dxn = lds + FwdLds::dxn;
  double* acc = lds + FwdLds::acc;
  double* accp = lds + FwdLds::accp;
  double* uz = lds + FwdLds::uz;
  double* pre = lds + FwdLds::pre;
  const double* KX = rx;
  const double* KE = rx + (REC_KE - REC_KX);
  const double* Zk = rx + (REC_Z - REC_KX);
  const double* DF = rx + (REC_DF - REC_KX);
  const double* QF = rx + (REC_QF - REC_KX);
  const double* RF = rx + (REC_RF - REC_KX);
  const double* META = rx + (REC_META - REC_KX);
  const double* DQ = rx + (REC_DQ - REC_KX);
  const double dt = rx[REC_DT - REC_KX];
  const int n_f = int(META[0]), mode = int(META[2]);
  bool cf[HB_NC];
  mode_flags(mode, cf);
  for (int r = cx.lane; r < 22; r += cx.nlanes) {
    const double* row = r < 12 ? gains + r * 22 : KX + (r - 12) * 22;
    // the four partial sums side by side (partial q: entries q, q + 4, ...): four short chains instead of one of 22
    double p[4] = {r < 12 ? gains[264 + r] : KE[r - 12], 0.0, 0.0, 0.0};
#pragma unroll 1
    for (int c = 0; c < 20; c += 4) {
      p[0] = fma(row[c], dx[c], p[0]);
      p[1] = fma(row[c + 1], dx[c + 1], p[1]);
      p[2] = fma(row[c + 2], dx[c + 2], p[2]);
      p[3] = fma(row[c + 3], dx[c + 3], p[3]);
    }
    p[0] = fma(row[20], dx[20], p[0]);
    p[1] = fma(row[21], dx[21], p[1]);
    const double s = fwd_sum4(p);
    if (r < 12) {
      ut[r] = s;
      if (r >= n_f && r - n_f < 6) uz[r - n_f] = s;
    } else {
      pre[r - 12] = s;
    }
  }
  cx.sync();
  for (int i = cx.lane; i < 22; i += cx.nlanes) {
    if (i < 12) {
      const double* row = ab + i * REC_LD;   // [A~ (22) | b~ | B~ (12)]: entry j of [dx ; u~] multiplies row[j] (j < 22) or row[j + 1]
      double p[4] = {row[REC_CV], 0.0, 0.0, 0.0};
#pragma unroll 1
      for (int j = 0; j < 20; j += 4) {
        p[0] = fma(row[j], dx[j], p[0]);
        p[1] = fma(row[j + 1], dx[j + 1], p[1]);
        p[2] = fma(row[j + 2], dx[j + 2], p[2]);
        p[3] = fma(row[j + 3], dx[j + 3], p[3]);
      }
      p[0] = fma(row[20], dx[20], p[0]);
      p[1] = fma(row[21], dx[21], p[1]);
      p[2] = fma(row[23], ut[0], p[2]);
      p[3] = fma(row[24], ut[1], p[3]);
#pragma unroll 1
      for (int j = 24; j < 32; j += 4) {
        p[0] = fma(row[j + 1], ut[j - 22], p[0]);
        p[1] = fma(row[j + 2], ut[j - 21], p[1]);
        p[2] = fma(row[j + 3], ut[j - 20], p[2]);
        p[3] = fma(row[j + 4], ut[j - 19], p[3]);
      }
      p[0] = fma(row[33], ut[10], p[0]);
      p[1] = fma(row[34], ut[11], p[1]);
      dxn[i] = fwd_sum4(p);
    } else {
      const int k = i - 12;
      double s = pre[k];
      for (int b = 0; b < 6; ++b) s = fma(Zk[k * 6 + b], uz[b], s);   // columns >= nz of Z are 0.0 in the record
      du[i] = s;
      dxn[i] = fma(dt, s, DQ[k] + dx[i]);  // joint row of the step: (q + dt qd)+ - q_next
    }
  }
  cx.sync();
  for (int c = cx.lane; c < 22 + 3; c += cx.nlanes) {
    if (c < 22) {
      double duc;
      if (c < 12) {
        const int foot = c / 3;
        int col = 0;
        for (int f = 0; f < foot; ++f) col += cf[f] ? 3 : 0;
        duc = cf[foot] ? ut[col + c % 3] : DF[c];
      } else {
        duc = du[c];
      }
      accp[c] += fma(QF[c], dx[c], RF[c] * duc);   // Armijo directional derivative: one partial sum per entry, reduced at the end
      dx_out[c] = dx[c];
      du_out[c] = duc;
      dx[c] = dxn[c];
    } else {
      acc[c - 21] += META[c - 19];  // merit, dyn, eq  <-  cost*dt, dyn_sse*dt, eq_sse*dt
    }
  }
  cx.sync();
}

#if defined(__HIP_DEVICE_COMPILE__)
// Wave form of the same step, on the staged LDS copy of the record part and the gains (FwdLds::AB / RX / G).  What does not change from
// stage to stage — which LDS words a lane multiplies — is worked out once per kernel (FwdLane); a term a lane does not have reads the
// zero slot (0.0 x 0.0 added: the value is unchanged), so the 64 lanes run one instruction stream without a branch:
//   phase A: lane = 4 row + q for rows 0..15, then the same for rows 16..21 (two independent chains of 6 FMAs);
//   phase B: lanes 0..47 = 4 row + q of the twelve dx+ rows (9 FMAs), lanes 48..57 the joint rows (6 FMAs);
//   phase C: lane = entry.
struct FwdLane {
  int mA1, mA1e, cA1, wA1, mA2, mA2e, cA2, wA2;   // phase A: row bases, the base of term 5 (entries 20..23), constant, destination
  int mv[9], io, aDQ, aDX, wB;                    // phase B: LDS word of the matrix entry | of the vector entry << 16, per term
  bool q0, rowB, jointB, hasA2;
};
__device__ inline void fwd_lane_init(int l, FwdLane& L) {
  typedef FwdLds F;
  constexpr int Z0 = F::dx + 22;
  const int q = l & 3;
  L.q0 = q == 0;
  auto rowm = [](int r) { return r < 12 ? F::G + r * 22 : F::RX + (r - 12) * 22; };
  auto rowc = [](int r) { return r < 12 ? F::G + 264 + r : F::RX + (REC_KE - REC_KX) + r - 12; };
  auto roww = [](int r) { return r < 12 ? F::ut + r : F::pre + r - 12; };
  const int r1 = l >> 2, r2 = 16 + r1 < 22 ? 16 + r1 : 21;
  L.hasA2 = 16 + r1 < 22;
  L.mA1 = rowm(r1) + q;  L.mA1e = q < 2 ? L.mA1 + 20 : Z0;  L.cA1 = q == 0 ? rowc(r1) : Z0;  L.wA1 = roww(r1);
  L.mA2 = rowm(r2) + q;  L.mA2e = q < 2 ? L.mA2 + 20 : Z0;  L.cA2 = q == 0 ? rowc(r2) : Z0;  L.wA2 = roww(r2);
  L.rowB = l < 48;
  L.jointB = l >= 48 && l < 58;
  L.aDQ = Z0; L.aDX = Z0; L.io = Z0; L.wB = F::dxn + 22;   // (dxn + 22: padding)
#pragma unroll
  for (int t = 0; t < 9; ++t) L.mv[t] = Z0 | (Z0 << 16);
  if (L.rowB) {
    const int i = l >> 2, mB = F::AB + i * REC_LD;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int j = q + 4 * t;
      if (j < 34) L.mv[t] = (mB + (j < 22 ? j : j + 1)) | ((j < 22 ? F::dx + j : F::ut + j - 22) << 16);
    }
    if (q == 0) L.io = mB + REC_CV;
    L.wB = F::dxn + i;
  } else if (L.jointB) {
    const int k = l - 48;
#pragma unroll
    for (int t = 0; t < 6; ++t) L.mv[t] = (F::RX + (REC_Z - REC_KX) + k * 6 + t) | ((F::uz + t) << 16);
    L.io = F::pre + k;
    L.aDQ = F::RX + (REC_DQ - REC_KX) + k;
    L.aDX = F::dx + 12 + k;
    L.wB = F::dxn + 12 + k;
  }
}
// accp / accm: this lane's running Armijo partial (lane = entry < 22) and merit / dyn / eq sum (lanes 22..24), kept in registers
template <class Ctx>
__device__ inline void riccati_fwd_stage_wave(const Ctx& cx, double* lds, const FwdLane& L, double& accp, double& accm, double* dx_out,
                                              double* du_out) {
  typedef FwdLds F;
  const int l = cx.lane, q = l & 3;
  const double* META = lds + F::RX + (REC_META - REC_KX);
  const int n_f = __builtin_amdgcn_readfirstlane(int(META[0])), mode = __builtin_amdgcn_readfirstlane(int(META[2]));
  // ---- phase A
  {
    double m1[6], m2[6], v[6];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      m1[t] = lds[L.mA1 + 4 * t];
      m2[t] = lds[L.mA2 + 4 * t];
      v[t] = lds[F::dx + q + 4 * t];
    }
    m1[5] = lds[L.mA1e];
    m2[5] = lds[L.mA2e];
    v[5] = lds[F::dx + q + 20];
    double s1 = lds[L.cA1], s2 = lds[L.cA2];
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      s1 = fma(m1[t], v[t], s1);
      s2 = fma(m2[t], v[t], s2);
    }
    s1 = quad_sum_f64(s1);
    s2 = quad_sum_f64(s2);
    if (L.q0) {
      lds[L.wA1] = s1;
      const int b = (l >> 2) - n_f;
      if (l < 48 && b >= 0 && b < 6) lds[F::uz + b] = s1;
      if (L.hasA2) lds[L.wA2] = s2;
    }
  }
  cx.sync();
  // ---- phase B
  {
    double m[9], v[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      m[t] = lds[L.mv[t] & 0xffff];
      v[t] = lds[L.mv[t] >> 16];
    }
    double s = lds[L.io];
    const double dq = lds[L.aDQ], dxk = lds[L.aDX], dt = lds[F::RX + (REC_DT - REC_KX)];
#pragma unroll
    for (int t = 0; t < 9; ++t) s = fma(m[t], v[t], s);
    const double qs = quad_sum_f64(s);
    if (L.rowB) {
      if (L.q0) lds[L.wB] = qs;
    } else if (L.jointB) {
      lds[F::du + 12 + l - 48] = s;
      lds[L.wB] = fma(dt, s, dq + dxk);
    }
  }
  cx.sync();
  // ---- phase C
  if (l < 22) {
    int a_du = F::du + l;
    if (l < 12) {
      const int cfm = (mode == 2 || mode == 3 ? 5 : 0) | (mode == 1 || mode == 3 ? 10 : 0);   // mode_flags as a mask (feet 0 / 2 left, 1 / 3 right)
      const int foot = l / 3;
      const int col = 3 * __builtin_popcount(cfm & ((1 << foot) - 1));
      a_du = (cfm >> foot) & 1 ? F::ut + col + l - 3 * foot : F::RX + (REC_DF - REC_KX) + l;
    }
    const double dxc = lds[F::dx + l], duc = lds[a_du], qf = lds[F::RX + (REC_QF - REC_KX) + l], rf = lds[F::RX + (REC_RF - REC_KX) + l];
    const double dxn_c = lds[F::dxn + l];
    accp += fma(qf, dxc, rf * duc);
    dx_out[l] = dxc;
    du_out[l] = duc;
    lds[F::dx + l] = dxn_c;
  } else if (l < 25) {
    accm += META[l - 19];
  }
  cx.sync();
}
#endif
template <class Ctx>
HB_HD void riccati_fwd_finish(const Ctx& cx, double* lds) {
  if (cx.lane == 0) {
    double a = 0;
    for (int c = 0; c < 22; ++c) a += lds[FwdLds::accp + c];
    lds[FwdLds::acc] = a;
  }
  cx.sync();
}

// Value-only evaluation of one node for the line search (one thread per node):
// returns dt * (stage cost), dt * |defect|^2, dt * |equality constraints|^2  (OCS2 PerformanceIndex).
// `xnext(i)` delivers entry i of the next node's state (only the defect reads it, once per entry: the kernel forms the
// trial value from global memory on the fly instead of staging it next to x and u).
template <class XN>
HB_HD void node_value(const DevModel& M, const DevConfig& C, const double* x, const double* u, XN xnext,
                      const double* xref, const double* swing, double dt, int mode, double* out3) {
  bool cf[HB_NC];
  mode_flags(mode, cf);
  double f1[12], f2[12];
  Centroidal<double> c1;
  const double inv_m = rcp_t(M.total_mass);
  {
    centroidal_eval<double>(M, x + 9, x + 12, x, u + 12, c1);
    Vec3<double> fs, ms;
#pragma unroll
    for (int i = 0; i < HB_NC; ++i) {
      const Vec3<double> F(u[3 * i], u[3 * i + 1], u[3 * i + 2]);
      fs = fs + F;
      ms = ms + cross(c1.foot_rel[i] - c1.com_rel, F);
    }
    f1[0] = inv_m * fs.x; f1[1] = inv_m * fs.y; f1[2] = inv_m * fs.z - M.gravity;
    f1[3] = inv_m * ms.x; f1[4] = inv_m * ms.y; f1[5] = inv_m * ms.z;
    f1[6] = c1.v_lin.x; f1[7] = c1.v_lin.y; f1[8] = c1.v_lin.z;
    f1[9] = c1.euler_rate.x; f1[10] = c1.euler_rate.y; f1[11] = c1.euler_rate.z;
  }
  {
    double hn[6], zyx[3];
#pragma unroll
    for (int i = 0; i < 6; ++i) hn[i] = x[i] + dt * f1[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) zyx[i] = x[9 + i] + dt * f1[9 + i];
    Centroidal<double> c2;
    centroidal_eval_f<double>(M, zyx, [x, u, dt](int j) { return x[12 + j] + dt * u[12 + j]; }, hn,
                              [u](int j) { return u[12 + j]; }, c2);
    Vec3<double> fs, ms;
#pragma unroll
    for (int i = 0; i < HB_NC; ++i) {
      const Vec3<double> F(u[3 * i], u[3 * i + 1], u[3 * i + 2]);
      fs = fs + F;
      ms = ms + cross(c2.foot_rel[i] - c2.com_rel, F);
    }
    f2[0] = inv_m * fs.x; f2[1] = inv_m * fs.y; f2[2] = inv_m * fs.z - M.gravity;
    f2[3] = inv_m * ms.x; f2[4] = inv_m * ms.y; f2[5] = inv_m * ms.z;
    f2[6] = c2.v_lin.x; f2[7] = c2.v_lin.y; f2[8] = c2.v_lin.z;
    f2[9] = c2.euler_rate.x; f2[10] = c2.euler_rate.y; f2[11] = c2.euler_rate.z;
  }
  double dyn = 0;
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const double d = x[i] + 0.5 * dt * (f1[i] + f2[i]) - xnext(i);
    dyn += d * d;
  }
#pragma unroll
  for (int j = 0; j < HB_NJ; ++j) {
    const double d = x[12 + j] + dt * u[12 + j] - xnext(12 + j);
    dyn += d * d;
  }
  double cost = 0, eq = 0;
  int nc = 0;
  for (int i = 0; i < HB_NC; ++i) nc += cf[i];
  const double fz_nom = nc > 0 ? M.total_mass * M.gravity / nc : 0.0;
  for (int i = 0; i < HB_NX; ++i) {
    const double dx = x[i] - xref[i];
    cost += 0.5 * C.Q_diag[i] * dx * dx;
  }
  for (int i = 0; i < 12; ++i) {
    const double du = u[i] - ((i % 3 == 2 && cf[i / 3]) ? fz_nom : 0.0);
    cost += 0.5 * C.R_FF_diag[i] * du * du;
  }
  for (int k = 0; k < HB_NJ; ++k) {
    double s = 0;
    for (int l = 0; l < HB_NJ; ++l) s += C.R_jj[k * 10 + l] * u[12 + l];
    cost += 0.5 * u[12 + k] * s;
  }
  const RelaxedBarrierD fb{C.fb_mu, C.fb_delta};
  const RelaxedBarrierD bp{C.pos_b[0], C.pos_b[1]}, bv{C.vel_b[0], C.vel_b[1]}, bf{C.force_b[0], C.force_b[1]};
  for (int i = 0; i < HB_NC; ++i) {
    const double Fx = u[3 * i], Fy = u[3 * i + 1], Fz = u[3 * i + 2];
    const double pz = x[8] + c1.foot_rel[i].z;
    if (cf[i]) {
      const double h = C.friction_mu * (Fz + C.friction_gripper) - sqrt(Fx * Fx + Fy * Fy + C.friction_reg);
      cost += fb.value(h);
      const double r2 = c1.foot_vel[i].z + C.zv_gain * pz + C.zv_off;
      eq += c1.foot_vel[i].x * c1.foot_vel[i].x + c1.foot_vel[i].y * c1.foot_vel[i].y + r2 * r2;
    } else {
      const double* sw = swing + 6 * i;
      const double r0 = c1.foot_vel[i].z + C.kp_normal * pz - (sw[5] + C.kp_normal * sw[2]);
      const double r1 = C.xy_gain * (x[6] + c1.foot_rel[i].x) + c1.foot_vel[i].x - (sw[3] + C.xy_gain * sw[0]);
      const double r2 = C.xy_gain * (x[7] + c1.foot_rel[i].y) + c1.foot_vel[i].y - (sw[4] + C.xy_gain * sw[1]);
      eq += r0 * r0 + Fx * Fx + Fy * Fy + Fz * Fz;
      cost += 0.5 * C.soft_w * (r1 * r1 + r2 * r2);
    }
    cost += bf.value(Fz - C.force_lim[0]) + bf.value(C.force_lim[1] - Fz);
  }
#pragma unroll 1  // 40 barrier evaluations: rolled, or the inlined logarithms push the kernel beyond 512 registers
  for (int j = 0; j < HB_NJ; ++j) {
    cost += bp.value(x[12 + j] - M.q_lower[j]) + bp.value(M.q_upper[j] - x[12 + j]);
    cost += bv.value(u[12 + j] + M.qd_limit[j]) + bv.value(M.qd_limit[j] - u[12 + j]);
  }
  out3[0] = dt * cost;
  out3[1] = dt * dyn;
  out3[2] = dt * eq;
}

// Filter line-search acceptance (OCS2 FilterLinesearch::acceptStep).
HB_HD bool filter_accept(const DevConfig& C, double base_merit, double base_viol, double merit, double viol, double alpha,
                         double armijo) {
  if (viol > C.g_max) return viol < (1.0 - C.gamma_c) * base_viol;
  if (viol < C.g_min && base_viol < C.g_min && armijo < 0.0) return merit < base_merit + C.armijo * alpha * armijo;
  return (merit < base_merit - C.gamma_c * base_viol) || (viol < (1.0 - C.gamma_c) * base_viol);
}

}  // namespace hb
