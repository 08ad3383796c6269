// Small-state path (S <= 32: LG 20x20, SiteRM 20x20 / 21x21 / 4x4, toys).
//
// One workgroup per site l, NW wavefronts.  Wave 0 diagonalises the
// symmetrised rate matrix (jacobi_wave.hip.h); then the B buckets of the site
// are dealt round-robin to the waves.  A bucket never touches LDS for matrix
// data: U lives in registers in both MFMA operand forms and the three
// products of a bucket are chained through the accumulator registers
//
//   Pt  = I + t A + (U phi2(t lam)) U^T          (= expm(t A), symmetric)
//   Gt^T = -C^T / Pt / n                          (C^T streamed from HBM once)
//   T   = Gt U ;  W = U^T T ;  M += W o Phi(t)    (Daleckii-Krein)
//
// using  v_mfma_f64_16x16x4_f64 on a 32x32 (NT = 2) or 16x16 (NT = 1) padded
// frame.  The per-site epilogue forms dA = U M U^T and dQ = D^1/2 dA D^-1/2.
//
// HBM traffic: C^T once (B*S*S*8 bytes per site) + O(S^2) -- the algorithmic
// minimum of SURVEY.md 8(d).
#pragma once
#include "common.hip.h"
#include "jacobi_wave.hip.h"

#define CB_LS 33  // LDS row stride (doubles) of the 32x32 frames
#ifndef CB_SMALL_MIN_WGS
#define CB_SMALL_MIN_WGS 2
#endif


struct SmallArgs {
  int S, L, B;        // B = bucket stride of t / Ct / P
  const int *nlive;   // [L] buckets to visit per site (live ones are stored first), or null = B
  const double *t;    // [L,B]
  const double *Ct;   // [L,B,S,S]  (transposed counts)
  const double *Cq;   // S <= 20: the same counts in quad order [L][nq][TS*TS][64] (see quad_load_counts)
  int nq;             // quads per site in Cq
  const double *inv_n;  // [L]  1/n_l or 1
  const double *dirsum; // [L,S]  colsum_k - rowsum_k of sum_b C (direct pi term)
  const double *Q;    // [L,S,S]
  const double *pi;   // [L,S]
  double *loss;       // [L]
  double *dQ;         // [L,S,S] or null
  double *P;          // [L,B,S,S] (expm mode) or null
  double *lam_out;    // [L,S] (eigh mode) or null
  double *U_out;      // [L,S,S] (eigh mode) or null
  int *status;        // [L] sweeps used by the eigensolver
  int nchunk = 1;     // SMALL_EXPM only: workgroups per site -- each repeats the site's (cheap, deterministic) eigensolve and
                      // writes its share of the buckets.  A bank of a few sites with thousands of buckets (the 20 rate
                      // categories x 2047 nodes of a likelihood family, FastCherries' 129 x 20 log-bank as ONE site) otherwise
                      // runs on as many CUs as it has sites: 0.76 ms on 20 CUs, 0.9 ms on one.
};

// LDS carve-up (doubles)
template <int NW>
struct SmallLds {
  static constexpr int FRAME = 32 * CB_LS;
  static constexpr int A = 0;                 // A, later X
  static constexpr int G = A + FRAME;         // Jacobi G, later M
  static constexpr int V = G + FRAME;         // eigenvectors (column k at V + k*LS)
  static constexpr int LAM = V + FRAME;       // 32
  static constexpr int D = LAM + 32;          // sqrt(pi)
  static constexpr int TAB = D + 32;          // per wave: 4 x (F[32], E[32], H[32]) (one set per MFMA block)
  static constexpr int M4 = TAB + NW * 384;   // per wave: 16 TS^2 <= 576 doubles, the M accumulator of the 4x4-tile path
  static constexpr int RED = M4 + NW * 576;   // (NW/2) * 1024 reduction slots (min 1)
  static constexpr int LOSS = RED + ((NW / 2) > 0 ? (NW / 2) : 1) * 1024;
  static constexpr int LOSSTOT = LOSS + NW;
  static constexpr int TOTAL = LOSSTOT + 1;
};

enum { SMALL_LOSSGRAD = 0, SMALL_EXPM = 1, SMALL_EIGH = 2 };

// ---- the bank of one site, given A (LDS), V/lam (LDS), d (LDS) -------------
// Each wave returns its partial M (registers) and loss.
template <int NT, int KS>
struct SmallFrags {
  double UA[NT][KS];  // U[16 mt + (l&15)][4 s + (l>>4)]   (A form; also the B form of U^T)
  // The other operand form, UB[nt][s] = U[4 s + (l>>4)][16 nt + (l&15)], and the
  // eigenvalues are re-read from LDS where used (ub()): 30 ds_read_b64 per bucket
  // buy ~60 VGPRs, i.e. the second wave per SIMD.
};

// UB[x][s] = U[j = 4 s + hi][m = 16 x + lo] = V[m*LS + j]; LDS frames are zero padded to 32
__device__ __forceinline__ double ub(const double *sV, int x, int s, int lo, int hi) {
  return sV[(16 * x + lo) * CB_LS + 4 * s + hi];
}

template <int NT, int KS>
__device__ __forceinline__ void load_frags(SmallFrags<NT, KS> &f, const double *sV,
                                           const double * /*sLam*/, int S) {
  const int lane = threadIdx.x & 63;
  const int lo = lane & 15, hi = lane >> 4;
#pragma unroll
  for (int x = 0; x < NT; ++x) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int i = 16 * x + lo, k = 4 * s + hi;
      // U[i][k] = component i of eigenvector k = V[k*LS + i]
      f.UA[x][s] = (i < S && k < S) ? sV[k * CB_LS + i] : 0.0;
    }
  }
}

// Register slot (tile row-block x, register r) holds rows 16 x + (l>>4) + 4 r.
// S <= 4 KS, so slots with 16 x + 4 r >= 4 KS are dead for every lane: pruned at
// compile time (for S = 20: 10 of 16 slots survive).
#define SLOT_LIVE(x, r) (16 * (x) + 4 * (r) < 4 * KS)

template <int NT, int KS>
__device__ __forceinline__ void load_counts(double (&cval)[NT][NT][4], int S,
                                            const double *__restrict__ Ctb) {
  const int lane = threadIdx.x & 63;
  const int lo = lane & 15, hi = lane >> 4;
#pragma unroll
  for (int mt = 0; mt < NT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (SLOT_LIVE(mt, r)) {
          // clamped address + select: a guarded load costs a branch region each
          const int row = 16 * mt + hi + 4 * r, col = 16 * nt + lo;
          const double v = Ctb[min(row, S - 1) * S + min(col, S - 1)];
          cval[mt][nt][r] = (row < S && col < S) ? v : 0.0;
        }
}

template <int NT, int KS, int MODE>
__device__ __forceinline__ void small_bucket(const SmallFrags<NT, KS> &f, int S, double tb,
                                             double (&cval)[NT][NT][4],
                                             const double *__restrict__ Ct_next, double inv_n,
                                             const double *sA, const double *sD, const double *sV,
                                             double *tab /* wave-private F,E,H */,
                                             const double *sLam, double rho, d4 (&M)[NT][NT],
                                             double &lossacc, double *__restrict__ Pout) {
  const int lane = threadIdx.x & 63;
  const int lo = lane & 15, hi = lane >> 4;

  // Pt = U e^{t lam} U^T is evaluated as I + t A + U phi2(t lam) U^T while
  // t * rho <= 1 (rho >= spectral radius): entries that are O(t^2) keep full
  // RELATIVE accuracy.  Beyond that the split would cancel (I and t A against
  // U (-1 - x) U^T) and the plain form is used; no entry is tiny there.
  const bool split = tb * rho <= 1.0;
  // per-bucket spectral tables (lanes k < S), wave-private LDS
  if (lane < 32) {
    const double x = (lane < S) ? tb * sLam[lane] : 0.0;
    const double H = exp(0.5 * x);
    const double E = H * H;
    tab[lane] = split ? phi2(x) : (lane < S ? E : 0.0);
    tab[32 + lane] = E;
    tab[64 + lane] = H;
  }
  wave_lds_fence();

  // ---- Pt = I + t A + (U F) U^T ------------------------------------------
  double Fk[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) Fk[s] = tab[4 * s + hi];
  d4 g[NT][NT];
#pragma unroll
  for (int mt = 0; mt < NT; ++mt) {
    double uf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) uf[s] = f.UA[mt][s] * Fk[s];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < KS; ++s) acc = mfma_f64(uf[s], f.UA[nt][s], acc);
      g[mt][nt] = acc;
    }
  }
  // ---- epilogue: loss, Gt^T = -C^T / Pt / n  (in place) --------------------
  const double tsplit = split ? tb : 0.0, isplit = split ? 1.0 : 0.0;
#pragma unroll
  for (int mt = 0; mt < NT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (!SLOT_LIVE(mt, r)) {
          g[mt][nt][r] = 0.0;
          continue;
        }
        const int row = 16 * mt + hi + 4 * r, col = 16 * nt + lo;
        const bool valid = (row < S) && (col < S);
        double pt = g[mt][nt][r] + tsplit * sA[min(row, 31) * CB_LS + col] +
                    (row == col ? isplit : 0.0);
        pt = valid ? pt : 1.0;
        if (MODE == SMALL_EXPM) {
          // P[row][col] = Pt[row][col] d_col / d_row
          if (valid) Pout[row * S + col] = pt * sD[col] / sD[row];
        } else {
          const double c = cval[mt][nt][r];  // 0 on invalid slots
          const bool nz = c != 0.0;
          const double lg = fast_log(nz ? pt : 1.0);
          lossacc = fma(-c, lg, lossacc);
          g[mt][nt][r] = nz ? -c * inv_n * fast_rcp(pt) : 0.0;
        }
      }
  if (MODE == SMALL_EXPM) return;
  asm volatile("" : "+v"(lossacc));  // keep the logarithms here (see small_quad)
  // the counts of this bucket are consumed: start fetching the next bucket's
  if (Ct_next) load_counts<NT, KS>(cval, S, Ct_next);

  // ---- T = Gt U :  T[i][m] = sum_j Gt[i][j] U[j][m] -------------------------
  // A operand of k-step (jt, s) is register s of tile g[jt][it] (Gt^T[j][i]).
  d4 T[NT][NT];
#pragma unroll
  for (int it = 0; it < NT; ++it)
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
      d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int s = 0; s < 4; ++s)
          if (4 * jt + s < KS) acc = mfma_f64(g[jt][it][s], ub(sV, mt, 4 * jt + s, lo, hi), acc);
      T[it][mt] = acc;
    }
  // ---- W = U^T T, M += W o Phi ---------------------------------------------
  double EC[NT], HC[NT], LC[NT];
#pragma unroll
  for (int x = 0; x < NT; ++x) {
    EC[x] = tab[32 + ((16 * x + lo) & 31)];
    HC[x] = tab[64 + ((16 * x + lo) & 31)];
    LC[x] = sLam[(16 * x + lo) & 31];
  }
#pragma unroll
  for (int at = 0; at < NT; ++at)
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) {
      d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int it = 0; it < NT; ++it)
#pragma unroll
        for (int s = 0; s < 4; ++s)
          if (4 * it + s < KS) acc = mfma_f64(ub(sV, at, 4 * it + s, lo, hi), T[it][ct][s], acc);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (SLOT_LIVE(at, r)) {
          const int ra = (16 * at + hi + 4 * r) & 31;
          const double ER = tab[32 + ra], HR = tab[64 + ra];
          // divided difference, branch-free: Taylor form for |z| < 1/2, quotient otherwise
          const double dl = sLam[ra] - LC[ct];
          const double z = 0.5 * tb * dl;
          const bool near = fabs(z) < 0.5;
          const double taylor = tb * HR * HC[ct] * sinhc_small(near ? z : 0.0);
          const double quot = (ER - EC[ct]) * fast_rcp(near ? 1.0 : dl);
          M[at][ct][r] = fma(acc[r], near ? taylor : quot, M[at][ct][r]);
        }
    }
  wave_lds_fence();  // tab is rewritten by the next bucket
}

// ---- 4x4-tile path (S <= 24): FOUR buckets per wavefront pass, one per MFMA block ---------------
// With 16x16 tiles a 20-state matrix is padded to 32 x 32 (39 % useful work in the MFMAs and in the
// log / reciprocal / divided-difference epilogues).  v_mfma_f64_4x4x4f64 multiplies four
// independent 4x4x4 blocks per instruction: block b of every instruction belongs to bucket
// 4 quad + b, the TS x TS tiles of a matrix (TS = ceil(S / 4)) live in TS^2 registers (one double
// per lane), and nothing is padded for S = 20.  Same register chaining as the 16x16 path:
//   Pt tile (I,J)   = sum_K  A: (U F_b)(I,K)          B: UA[J][K] (= U^T(K,J) in B layout)
//   epilogue        -> G~^T tile (I,J) in place (counts are stored transposed)
//   T tile (It,Mt)  = sum_Jt A: g[Jt][It] (used as A = its transpose = G~(It,Jt))   B: U(Jt,Mt)
//                     stored over the dead g[Mt][It]
//   W tile (At,Ct)  = sum_It A: U^T(At,It) (= U(It,At) in B layout, same LDS words)  B: g[Ct][It] (= T(It,Ct))
//   M[At][Ct]      += W o Phi_b
// Lane = 16 q + 4 b + r: A layout (i = r, k = q), B layout (k = q, j = r), D layout (i = q, j = r).
// Counts of one quad (4 consecutive live buckets) in the order the lanes consume them:
// Cq[(I * TS + J) * 64 + lane] = Ct_bucket(4 quad + blk)[4 I + q][4 J + r], zero where the bucket
// or the row / column does not exist (written once by pack_counts_quad at cb_create): every load
// is 64 consecutive doubles.
template <int TS>
__device__ __forceinline__ void quad_load_counts(double (&cv)[TS][TS], const double *__restrict__ Cq) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int I = 0; I < TS; ++I)
#pragma unroll
    for (int J = 0; J < TS; ++J) cv[I][J] = Cq[(I * TS + J) * 64 + lane];
}

// LANEM: Mw is [TS*TS][64] (every lane owns its slots: no cross-lane traffic); else [TS*TS][16]
// (the four blocks are summed by shuffles first: a quarter of the LDS).
// SYM: every count matrix of the quad is symmetric (checked at cb_create; cherry counts and the SiteRM
// assembly with reverse transitions are, by construction).  Then Pt, G~ and W are symmetric: only the TS (TS + 1) / 2
// tiles on or above the diagonal get their MFMAs and their log / reciprocal / divided-difference epilogues (15 of
// 25 at 20 states: 275 instead of 375 MFMAs and ~28 % fewer vector instructions per quad), only their counts are
// loaded, the lower G~ tiles are the upper ones transposed (a 4 x 4 transposition inside the MFMA block = one lane
// permutation), and M is accumulated on its upper tiles only (the caller mirrors the sum).
// ULDS: the B-layout tiles of U are re-read from LDS where they are used (25 ds_read_b64 per tile row of the T and
// W phases) instead of living in 2 TS^2 registers across both phases -- the register peak drops by ~50, which is what
// lets three workgroups share a CU (sp_bank).
// LOGT: the logarithms of the loss by table (fast_log_table; `ltab` = 256 doubles of LDS filled by fast_log_table_fill)
// QLS / TABS: LDS row stride of the A / V frames and distance of the four blocks' spectral tables (doubles).  With the 33 / 96 of
// the 32 x 32 frames a `ds_read_b64` of a quad is a 2-way bank conflict in both tile layouts (rows q and q + 1, 33 doubles
// apart, overlap in 3 of their 4 words) and a 4-way one on the tables (96 doubles = 3 bank rows: the four blocks on the same
// banks): 2.06 conflict cycles per LDS instruction in sp_bank<5, true, true> (profiles/r03_sp_bank_sq_counters.json), in a
// kernel whose four SIMDs share one LDS.  36 / 100 (both = 4 mod 32) make every read of the quad conflict-free: the lanes of a
// 32-lane group address q * 36 + r, r * 36 + q or blk * 100 + {q, r} -- disjoint words.
// Round 5 (sp_bank is bound by vector-instruction ISSUE: 292 M VALU against 45 M MFMA instructions per launch,
// profiles/r04_sp_bank_sq_counters.json): the P epilogue sanitises its argument once (a zero count contributes 0 * log 1 and
// 0 * 1 without further selects), the table logarithm is guarded by ONE class compare and one select (NaN for anything but a
// positive finite argument) instead of three compares and three selects, and the divided difference takes its Taylor form only below |z| = 1/16 (five terms; above, the
// quotient loses at most three bits).
template <int TS, bool LANEM, bool SYM = false, bool ULDS = false, bool LOGT = false, int QLS = CB_LS, int TABS = 96>
__device__ __forceinline__ void small_quad(int S, double tb, const double *__restrict__ Cq, double inv_n,
                                           const double *sA, const double *sV, double *tabw,
                                           const double *sLam, double rho, double *Mw, double &lossacc,
                                           const double *ltab = nullptr) {
  const int lane = threadIdx.x & 63, q = lane >> 4, blk = (lane >> 2) & 3, r = lane & 3;
  // counts of this quad: issued first (coalesced, 64 consecutive doubles per load), consumed by the
  // epilogue after the table computation and the first MFMA row -- no registers held across quads
  double cv[TS][TS];
  if (SYM) {
#pragma unroll
    for (int I = 0; I < TS; ++I)
#pragma unroll
      for (int J = I; J < TS; ++J) cv[I][J] = Cq[(I * TS + J) * 64 + lane];
  } else {
    quad_load_counts<TS>(cv, Cq);
  }
  double *tab = tabw + blk * TABS;  // this block's F[32], E[32], H[32]
  const bool split = tb * rho <= 1.0;
  // spectral tables of the four buckets: 16 lanes per block, lanes (q, r) cover k = 4 q + r and + 16
  {
    const int k0 = 4 * q + r;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = k0 + 16 * h;
      const double x = (k < S) ? tb * sLam[k] : 0.0;
      const double H = exp(0.5 * x);
      const double E = H * H;
      tab[k] = split ? phi2(x) : (k < S ? E : 0.0);
      tab[32 + k] = E;
      tab[64 + k] = H;
    }
  }
  wave_lds_fence();
  // Register budget (2 waves per SIMD = 256 VGPRs): g 2 TS^2 + counts 2 TS^2 + ONE operand form of U
  // 2 TS^2 at a time.  The two forms are re-read from LDS at the start of their phase (the memory
  // clobbers keep the compiler from hoisting both out of the bucket loop) and the scheduler is
  // fenced between tile rows, otherwise it interleaves all TS^2 epilogues and spills ~300 registers.
  double g[TS][TS];
  {
    double Fk[TS];
#pragma unroll
    for (int K = 0; K < TS; ++K) Fk[K] = tab[4 * K + q];
    const double tsplit = split ? tb : 0.0, isplit = split ? 1.0 : 0.0;
    // ---- Pt row I, epilogue in place -> g[I][*] = G~^T tiles ---------------------------------
    // U tiles in A layout (U(I,K): U[4 I + r][4 K + q] = sV[(4 K + q) LS + 4 I + r]; the same words are
    // U^T(K,I) in B layout) are re-read from LDS for every row: 30 ds_read_b64 per row are cheap, the
    // 2 TS^2 registers they would occupy are not (the clobber keeps the compiler from caching them).
#pragma unroll
    for (int I = 0; I < TS; ++I) {
      asm volatile("" ::: "memory");
      double uf[TS];
#pragma unroll
      for (int K = 0; K < TS; ++K) uf[K] = sV[(4 * K + q) * QLS + 4 * I + r] * Fk[K];
#pragma unroll
      for (int K = 0; K < TS; ++K)       // K outer: TS independent accumulator chains in flight
#pragma unroll
        for (int J = SYM ? I : 0; J < TS; ++J)
          g[I][J] = mfma4_f64(uf[K], sV[(4 * K + q) * QLS + 4 * J + r], K == 0 ? 0.0 : g[I][J]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int J = SYM ? I : 0; J < TS; ++J) {
        const int row = 4 * I + q, col = 4 * J + r;
        double pt = g[I][J] + tsplit * sA[min(row, 31) * QLS + min(col, 31)] + (row == col ? isplit : 0.0);
        const double c = cv[I][J];   // (0 on padded slots: pack_counts_quad)
        // a zero count (a padded slot, an unobserved pair -- where rounding may leave Pt <= 0) contributes 0 * log 1 and 0 / 1
        const bool use = c != 0.0;
        pt = use ? pt : 1.0;
        // (SYM: an off-diagonal tile stands for its mirror image too)
        if (LOGT) {
          // (anything but a positive finite argument: NaN -- the loss must never look finite then)
          const double lg = is_pos_finite_nonzero(pt) ? fast_log_table_unchecked(pt, ltab) : NAN;
          lossacc = fma((SYM && J > I) ? -2.0 * c : -c, lg, lossacc);
        } else {
          lossacc = fma((SYM && J > I) ? -2.0 * c : -c, fast_log(pt), lossacc);
        }
        g[I][J] = -c * inv_n * fast_rcp(pt);
      }
      // pin the loss here: otherwise the compiler sinks all TS^2 logarithms (they feed nothing but
      // lossacc) to the end of the quad and keeps TS^2 Pt values alive across the T and W phases
      asm volatile("" : "+v"(lossacc));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (SYM) {
    // lower tiles of G~: tile (J, I) = tile (I, J)^T; lane (q, blk, r) takes the value of lane (r, blk, q)
    const int src = 16 * r + 4 * blk + q;
#pragma unroll
    for (int I = 0; I < TS; ++I)
#pragma unroll
      for (int J = I + 1; J < TS; ++J) g[J][I] = __shfl(g[I][J], src, 64);
  }
  asm volatile("" ::: "memory");
  double UB[ULDS ? 1 : TS][ULDS ? 1 : TS];  // B layout of U(I,K): U[4 I + q][4 K + r]  (= A layout of U^T(K,I))
  if (!ULDS) {
#pragma unroll
    for (int I = 0; I < TS; ++I)
#pragma unroll
      for (int K = 0; K < TS; ++K) UB[ULDS ? 0 : I][ULDS ? 0 : K] = sV[(4 * K + r) * QLS + 4 * I + q];
  }
  const double *ub0 = sV + r * QLS + q;   // UB(I, K) = ub0[4 K QLS + 4 I]
#define Q_UB(I, K) (ULDS ? ub0[4 * (K) * QLS + 4 * (I)] : UB[ULDS ? 0 : (I)][ULDS ? 0 : (K)])
  // ---- T(It,Mt) = sum_Jt G~(It,Jt) U(Jt,Mt), stored over g[Mt][It] ---------------------------
#pragma unroll
  for (int It = 0; It < TS; ++It) {
    if (ULDS) asm volatile("" ::: "memory");   // (keeps the compiler from hoisting the U reads out of the row loop)
    double acc[TS];
#pragma unroll
    for (int Jt = 0; Jt < TS; ++Jt)
#pragma unroll
      for (int Mt = 0; Mt < TS; ++Mt) acc[Mt] = mfma4_f64(g[Jt][It], Q_UB(Jt, Mt), Jt == 0 ? 0.0 : acc[Mt]);
#pragma unroll
    for (int Mt = 0; Mt < TS; ++Mt) g[Mt][It] = acc[Mt];
    __builtin_amdgcn_sched_barrier(0);
  }
  // ---- W(At,Ct) = sum_It U^T(At,It) T(It,Ct);  M += W o Phi -----------------------------------
#pragma unroll
  for (int Ct = 0; Ct < TS; ++Ct) {
    const int c = min(4 * Ct + r, 31);
    const double EC = tab[32 + c], HC = tab[64 + c], LC = sLam[c];
    if (ULDS) asm volatile("" ::: "memory");
    double acc[TS];
#pragma unroll
    for (int It = 0; It < TS; ++It)
#pragma unroll
      for (int At = 0; At < TS; ++At)
        if (!SYM || At <= Ct) acc[At] = mfma4_f64(Q_UB(It, At), g[Ct][It], It == 0 ? 0.0 : acc[At]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int At = 0; At < TS; ++At) {
      if (SYM && At > Ct) continue;   // W is symmetric: the caller mirrors the summed M
      const int ra = min(4 * At + q, 31);
      const double ER = tab[32 + ra], HR = tab[64 + ra];
      const double dl = sLam[ra] - LC;
      const double z = 0.5 * tb * dl;
      // Taylor form below |z| = 1/16 only: above, (e^a - e^b) / (a - b) loses log2(1 / (1 - e^-2|z|)) <= 3.1 bits
      const bool near = fabs(z) < 0.0625;
      const double taylor = tb * HR * HC * sinhc_tiny(near ? z : 0.0);
      const double quot = (ER - EC) * fast_rcp(near ? 1.0 : dl);
      double m = acc[At] * (near ? taylor : quot);
      if (LANEM) {
        // wave-private slot of this lane: one ds_add_f64, nothing to wait for
        __hip_atomic_fetch_add(&Mw[(At * TS + Ct) * 64 + lane], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      } else {
        // sum over the four buckets of this pass, then accumulate in the wave-private LDS array
        m += __shfl_xor(m, 4);
        m += __shfl_xor(m, 8);
        if (blk == 0) Mw[(At * TS + Ct) * 16 + 4 * q + r] += m;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  wave_lds_fence();  // the tables are rewritten by the next quad
#undef Q_UB
}

// ---- A = sym(D^1/2 Q D^-1/2) into LDS -----------------------------------------
__device__ __forceinline__ void small_build_A(int S, const double *__restrict__ Q,
                                              const double *__restrict__ pi, double *sA,
                                              double *sD) {
  for (int i = threadIdx.x; i < 32; i += blockDim.x) sD[i] = (i < S) ? sqrt(pi[i]) : 1.0;
  __syncthreads();
  for (int e = threadIdx.x; e < S * S; e += blockDim.x) {
    const int i = e / S, j = e - i * S;
    const double di = sD[i], dj = sD[j];
    sA[i * CB_LS + j] = 0.5 * (di * Q[i * S + j] / dj + dj * Q[j * S + i] / di);
  }
  __syncthreads();
}

// ---- dA = U M U^T in place: sG holds M on entry, dL/dA on exit ---------------------
// sX: >= 1024 doubles of scratch.  All threads call; ends with a barrier.
__device__ __forceinline__ void small_dA_from_M(int S, double *sG, const double *sV, double *sX) {
  // X = M U^T  (X[a][j] = sum_c M[a][c] U[j][c])
  for (int e = threadIdx.x; e < S * S; e += blockDim.x) {
    const int ra = e / S, j = e - ra * S;
    double acc = 0.0;
    for (int c = 0; c < S; ++c) acc = fma(sG[ra * CB_LS + c], sV[c * CB_LS + j], acc);
    sX[ra * 32 + j] = acc;
  }
  __syncthreads();
  // dA = U X
  for (int e = threadIdx.x; e < S * S; e += blockDim.x) {
    const int i = e / S, j = e - i * S;
    double acc = 0.0;
    for (int k = 0; k < S; ++k) acc = fma(sV[k * CB_LS + i], sX[k * 32 + j], acc);
    sG[i * CB_LS + j] = acc;
  }
  __syncthreads();
}

// ---- one site: A (LDS) -> loss, dL/dA (LDS) -----------------------------------
// Precondition : sA = A (symmetric, stride CB_LS), sD = sqrt(pi); all threads call.
// Postcondition: lds[LD::LOSS + NW - 1 ... ] unchanged; returns nothing, but
//   lds[LD::TOTAL - 1]  (slot "LOSSTOT") = loss of the site (thread 0 wrote it),
//   sG = dL/dA (free S x S matrix) when want_grad, sV / sLam = eigenvectors / values.
// Ends with a barrier.
template <int NT, int KS, int NW, int MODE>
__device__ __forceinline__ void small_site_eval(double *lds, int S, int B,
                                                const double *__restrict__ t_l,
                                                const double *__restrict__ Ct_l, double inv_n,
                                                const double *__restrict__ dirsum_l,
                                                double *__restrict__ P_l, bool want_grad,
                                                int *sweeps_out, bool warm = false,
                                                const double *__restrict__ Cq_l = nullptr, int b_lo = 0, int b_hi = -1) {
  using LD = SmallLds<NW>;
  if (b_hi < 0) b_hi = B;   // SMALL_EXPM: this workgroup's buckets [b_lo, b_hi)
  double *sA = lds + LD::A, *sG = lds + LD::G, *sV = lds + LD::V, *sLam = lds + LD::LAM,
         *sD = lds + LD::D;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int lo = lane & 15, hi = lane >> 4;
  if (wave == 0) {
    const int sweeps = wave_eigh_rate(S, sA, sG, sV, sLam, CB_LS, warm);
    if (lane == 0 && sweeps_out) *sweeps_out = sweeps;
  }
  __syncthreads();
  // zero padding of the 32 x 32 eigenvector frame and of lam (operands are read unguarded)
  for (int e = threadIdx.x; e < 32 * 32; e += blockDim.x) {
    const int k = e >> 5, i = e & 31;
    if (k >= S || i >= S) sV[k * CB_LS + i] = 0.0;
  }
  for (int k = S + threadIdx.x; k < 32; k += blockDim.x) sLam[k] = 0.0;
  __syncthreads();
  // Gershgorin: |lam| <= 2 max |A_ii| for a symmetrised rate matrix
  double rho = 0.0;
  for (int i = lane; i < S; i += 64) rho = fmax(rho, fabs(sA[i * CB_LS + i]));
  rho = 2.0 * wave_max(rho);
  double *tab = lds + LD::TAB + wave * 384;
  double *red = lds + LD::RED;
  if (MODE == SMALL_LOSSGRAD && KS <= 6) {
    // ---- 4x4 tiles, four buckets per pass (see small_quad) --------------------------------------
    constexpr int TS = KS <= 6 ? KS : 1;
    const int blk = (lane >> 2) & 3;
    double *Mw = lds + LD::M4 + wave * 576;
    for (int e = lane; e < 16 * TS * TS; e += 64) Mw[e] = 0.0;
    wave_lds_fence();
    double lossacc = 0.0;
    const int nquads = (B + 3) / 4;
    for (int qd = wave; qd < nquads; qd += NW) {
      const int bucket = 4 * qd + blk;
      const double tb = bucket < B ? t_l[bucket] : 0.0;   // a missing bucket: t = 0, no counts -> contributes nothing
      small_quad<TS, false>(S, tb, Cq_l + (size_t)qd * (TS * TS * 64), inv_n, sA, sV, tab, sLam, rho, Mw, lossacc);
    }
    lossacc = wave_sum(lossacc);
    if (lane == 0) lds[LD::LOSS + wave] = lossacc;
    if (want_grad) {
      // sum over the waves in a fixed order, into the frame sG[a][c]
      __syncthreads();
      for (int e = threadIdx.x; e < 16 * TS * TS; e += blockDim.x) {
        double tot = 0.0;
        for (int w = 0; w < NW; ++w) tot += lds[LD::M4 + w * 576 + e];
        const int tile = e >> 4, At = tile / TS, Ct = tile - At * TS;
        const int a = 4 * At + ((e >> 2) & 3), c = 4 * Ct + (e & 3);
        sG[a * CB_LS + c] = tot;
      }
    }
    __syncthreads();
  } else {
  SmallFrags<NT, KS> f;
  load_frags<NT, KS>(f, sV, sLam, S);
  d4 M[NT][NT];
#pragma unroll
  for (int x = 0; x < NT; ++x)
#pragma unroll
    for (int y = 0; y < NT; ++y) M[x][y] = d4{0.0, 0.0, 0.0, 0.0};
  double lossacc = 0.0;
  double cval[NT][NT][4];
  if (MODE == SMALL_LOSSGRAD && wave < B) load_counts<NT, KS>(cval, S, Ct_l + (size_t)wave * S * S);
  for (int b = (MODE == SMALL_EXPM ? b_lo : 0) + wave; b < (MODE == SMALL_EXPM ? b_hi : B); b += NW) {
    const double *next =
        (MODE == SMALL_LOSSGRAD && b + NW < B) ? Ct_l + (size_t)(b + NW) * S * S : nullptr;
    small_bucket<NT, KS, MODE>(f, S, t_l[b], cval, next, inv_n, sA, sD, sV, tab, sLam, rho, M,
                               lossacc, MODE == SMALL_EXPM ? P_l + (size_t)b * S * S : nullptr);
  }
  if (MODE == SMALL_EXPM) return;

  // ---- loss ------------------------------------------------------------------
  lossacc = wave_sum(lossacc);
  if (lane == 0) lds[LD::LOSS + wave] = lossacc;
  // ---- deterministic tree reduction of M over the waves -----------------------
  if (want_grad) {
    for (int stride = NW / 2; stride >= 1; stride >>= 1) {
      if (wave >= stride && wave < 2 * stride) {
        double *dst = red + (wave - stride) * 1024;
#pragma unroll
        for (int x = 0; x < NT; ++x)
#pragma unroll
          for (int y = 0; y < NT; ++y)
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[((x * NT + y) * 4 + r) * 64 + lane] = M[x][y][r];
      }
      __syncthreads();
      if (wave < stride) {
        const double *src = red + wave * 1024;
#pragma unroll
        for (int x = 0; x < NT; ++x)
#pragma unroll
          for (int y = 0; y < NT; ++y)
#pragma unroll
            for (int r = 0; r < 4; ++r) M[x][y][r] += src[((x * NT + y) * 4 + r) * 64 + lane];
      }
      __syncthreads();
    }
    // M (wave 0) -> LDS frame sG[a][c]
    if (wave == 0) {
#pragma unroll
      for (int x = 0; x < NT; ++x)
#pragma unroll
        for (int y = 0; y < NT; ++y)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int ra = 16 * x + hi + 4 * r, c = 16 * y + lo;
            if (ra < 32 && c < 32) sG[ra * CB_LS + c] = M[x][y][r];
          }
    }
  }
  __syncthreads();
  }
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int w = 0; w < NW; ++w) tot += lds[LD::LOSS + w];
    // direct pi term: -(1/n) sum_k log d_k (colsum_k - rowsum_k)
    double dir = 0.0;
    for (int k = 0; k < S; ++k) dir = fma(log(sD[k]), dirsum_l[k], dir);
    lds[LD::LOSSTOT] = (tot - dir) * inv_n;
  }
  if (!want_grad) {
    __syncthreads();
    return;
  }
  small_dA_from_M(S, sG, sV, lds + LD::RED);
}

// (21 .. 32 states: the tile sets of the two larger forms need more than 256 registers; their dispatch takes the
// four-wave workgroup -- one wave per SIMD, all 512 registers -- instead of spilling)
template <int NT, int KS, int NW, int MODE>
__global__ __launch_bounds__(NW * 64, (KS >= 6 && NW == 4) ? 1 : CB_SMALL_MIN_WGS) void small_bank_kernel(SmallArgs a) {
  extern __shared__ double lds[];
  using LD = SmallLds<NW>;
  double *sA = lds + LD::A, *sG = lds + LD::G, *sV = lds + LD::V, *sLam = lds + LD::LAM,
         *sD = lds + LD::D;
  const int nch = (MODE == SMALL_EXPM && a.nchunk > 1) ? a.nchunk : 1;
  const int l = blockIdx.x / nch, chunk = blockIdx.x - l * nch, S = a.S, B = a.B;

  if (MODE == SMALL_EIGH) {
    // a.Q holds the symmetric matrices themselves
    for (int e = threadIdx.x; e < S * S; e += blockDim.x)
      sA[(e / S) * CB_LS + (e % S)] = a.Q[(size_t)l * S * S + e];
    __syncthreads();
    if (threadIdx.x < 64) {
      const int sweeps = wave_eigh_rate(S, sA, sG, sV, sLam, CB_LS);
      if (threadIdx.x == 0 && a.status) a.status[l] = sweeps;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < S * S; e += blockDim.x) {
      const int i = e / S, k = e - i * S;
      a.U_out[(size_t)l * S * S + e] = sV[k * CB_LS + i];
    }
    for (int k = threadIdx.x; k < S; k += blockDim.x) a.lam_out[(size_t)l * S + k] = sLam[k];
    return;
  }
  small_build_A(S, a.Q + (size_t)l * S * S, a.pi + (size_t)l * S, sA, sD);
  const size_t lb = (size_t)l * B;
  small_site_eval<NT, KS, NW, MODE>(lds, S, a.nlive ? a.nlive[l] : B, a.t + lb, a.Ct + lb * S * S, a.inv_n[l],
                                    a.dirsum + (size_t)l * S,
                                    MODE == SMALL_EXPM ? a.P + lb * S * S : nullptr,
                                    a.dQ != nullptr, (a.status && chunk == 0) ? a.status + l : nullptr, false,
                                    a.Cq ? a.Cq + (size_t)l * a.nq * (KS * KS * 64) : nullptr,
                                    (int)((long long)B * chunk / nch), (int)((long long)B * (chunk + 1) / nch));
  if (MODE == SMALL_EXPM) return;
  if (threadIdx.x == 0) a.loss[l] = lds[LD::LOSSTOT];
  if (a.dQ == nullptr) return;
  // dQ = D^1/2 dA D^-1/2
  for (int e = threadIdx.x; e < S * S; e += blockDim.x) {
    const int i = e / S, j = e - i * S;
    a.dQ[(size_t)l * S * S + e] = sD[i] * sG[i * CB_LS + j] / sD[j];
  }
}
