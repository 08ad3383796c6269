// Warm solves of the large-path eigensolver WITHOUT the host in the loop (round 4).
//
// eigh_large_host.hip.h asks the host for a decision after every sweep (which rotation, how many squarings): a pinned-memory
// round trip per sweep, and a solve whose duration depends on how fast the host answers.  Here every decision is taken ON THE
// DEVICE and left in a control block (`ctl`); the host only enqueues a PLAN -- a fixed sequence of launches, predicted from the
// previous epoch's solve -- in which every kernel reads the control block and does what the state needs:
//
//   slot s  (a "rotation sweep"):  lge_gram   Gamma = G^T G straight from the column-major G (both operands k-contiguous, 32-byte
//                                             loads), X / Xf and the sweep statistics; the launch behind it takes the DECISION
//                                             (lge_decide_here): all pairs or far pairs only, polynomial order, scaling,
//                                             squarings, "this is the final sweep"
//                                  lge_gemm   the powers of X, the Paterson-Stockmeyer steps, the squarings and G <- G R, each
//                                             predicated on the decision (a launch that is not needed returns at once)
//                                  lge_p34    X^3, X^4 and the polynomial coefficients B0, B1, B2
//   band pass after slot s:        lgj_round  banded Jacobi rounds (jacobi_block.hip.h), run when the sweep was a masked one
//   lge_norms / lge_finish:        norms, sorting, U / lambda -- or, when the plan ran out before convergence, the STALL word:
//                                  the host then repeats this solve with the host-driven loop (the only host intervention, rare)
//
// A plan that under-provides is never wrong, only slower: a slot that cannot evaluate the polynomial order / squarings the state
// asks for applies exp(alpha X) with alpha < 1 (an exactly orthogonal partial rotation; "damped") and the next slot continues.
// Schedule: band pass, far-pair rotation, band pass, then plain first-order sweeps ("B F B L L L"): on the recorded Adam trajectory
// of the bench bank (profiles/tools/eigh_proto.py) the large-angle rotations of near-degenerate neighbours BEFORE the far-pair
// rotation leave the state at 2e-5 .. 2.5e-4 after one far rotation instead of 2e-3 (the exact rotations of near pairs otherwise
// mix their large cosines into the far pairs again), and the second masked sweep of the old schedule (9 products) disappears.
// exp(Y) is evaluated to 12th order on |Y| <= 0.5 (remainder 0.5^13 / 13! = 2e-14, typically 1e-16) -- one product more than the
// 8th-order form, two fewer than 8th order + Newton-Schulz polish.
#pragma once
#include "jacobi_block.hip.h"
#include "large_bank.hip.h"

// control block (unsigned long long words)
enum {
  EC_STALL = 0,    // sticky: the plan ended without convergence (or met a non-finite matrix); lge_finish leaves U alone
  EC_FINAL = 1,    // index of the final sweep slot (EC_NONE while unknown): slots above it return at once
  EC_NSWEEP = 2,   // sweeps run so far
  EC_MODE = 3,     // decision of the current sweep: order | masked << 8 | damped << 9 | second order << 10 | active << 16
  EC_SQ = 4,       // squarings of the current sweep
  EC_ERR = 5,      // 2: non-finite input
  EC_ARRIVE = 6,   // lge_gram's arrival counter (left at zero)
  EC_MASKED = 7,   // the last sweep rotated far pairs only (the band pass behind it runs)
  EC_SC = 8,       // bits of the scale s: Y = s X
  EC_BANDS = 9,    // band passes run so far
  EC_SIGMA = 10,   // (in the host's record only: lge_norms) bits of sigma = max |A_ii|: the host follows 2 sigma >= rho, the range of the bank's time basis
  EC_TBSTALE = 11, // lge_norms: 2 sigma left the range the time basis was built for (tbasis.hip.h); the bank returns at once
  EC_SKIP = 12,    // EC_STALL | EC_TBSTALE: the word the kernels of a time-basis bank look at
  EC_REC = 16,     // 4 words per sweep: cosine, row sum (all pairs), row sum (far pairs), EC_MODE | sq << 24
  EC_MAXREC = 12,
  EC_JSTATE = 64,  // two words of lgj_round's own (running maximum; "finished", which stays zero here)
  EC_T0 = 80,      // s_memrealtime (100 MHz) when lge_begin ran, EC_TSWEEP + k: when sweep k's decision was taken, EC_TEND: lge_norms
  EC_TSWEEP = 81,  //   -- where a solve's time goes WITHOUT a tracer attached (CB_DEBUG prints the differences)
  EC_TEND = 93,
  EC_NSTAMP = 95,  // -DCB_EIGH_STAMPS (diagnostic build): every kernel of the solve leaves (s_memrealtime << 8 | kernel id) at its entry
  EC_STAMPS = 96,  //   in EC_STAMPS + i, i counted in EC_NSTAMP: the solve's launch-by-launch timeline with no tracer attached
#ifdef CB_EIGH_STAMPS
  EC_WORDS = 96 + 96
#else
  EC_WORDS = 96
#endif
};
#ifdef CB_EIGH_STAMPS
__device__ __forceinline__ void lge_stamp(unsigned long long *ctl, int id) {
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    const unsigned long long i = atomicAdd(ctl + EC_NSTAMP, 1ull);
    if (i < 96ull) ctl[EC_STAMPS + i] = ((unsigned long long)__builtin_amdgcn_s_memrealtime() << 8) | (unsigned long long)id;
  }
}
#define LGE_STAMP(ctl, id) lge_stamp(const_cast<unsigned long long *>(ctl), id)
#else

#define LGE_STAMP(ctl, id) ((void)0)
#endif
#define EC_NONE 0xFFFFFFFFull
// Control words are read through the CONSTANT address space: all the words a kernel tests come in one batch of scalar loads.
// As plain loads the compiler follows the short-circuit order of the tests -- one dependent round trip per word, ~0.7 us each
// behind a kernel boundary (round 6: 28 us per solve).  A kernel that also writes the block does so after its own reads, in one
// thread; a workgroup that starts late and sees the new words reads the same decision or words it does not use.
#ifdef CB_PLAIN_WORDS
typedef const unsigned long long *lge_const_words;
#else
typedef const __attribute__((address_space(4))) unsigned long long *lge_const_words;
#endif

struct GramArgs {
  int LD, band, slot;
  int cap;       // highest polynomial order this slot's launches evaluate: 2, 4 or 12
  int nsq;       // squaring launches this slot has
  const double *G;   // [LD][LD] column-major: G[c][r]
  double *X, *Xf;
  double *Gm, *dg;   // slots with the second-order launch: Gamma itself and its diagonal (as X was built from it)
  int so;            // the slot has the second-order launch (lge_so)
  unsigned long long *acc;   // the sweep's statistics: LGE_ACC_WORDS words (see lge_fix)
  unsigned long long *ctl;
  double trigger;
};

// The statistics a sweep's decision needs -- max_i sum_j |x_ij| over all pairs and over far pairs (|X|_2 <= |X|_inf for an
// antisymmetric X), the largest squared cosine -- without a launch to reduce them (rounds 4-5: lge_decide, one workgroup
// reading 400 x 25 partial row sums, 6.2 us per sweep).  Each of lge_gram's workgroups adds the LARGEST row sum of its tile
// to the word of its block row: B_I = sum_J max_{i in I} sum_{j in J} |x_ij| >= every row sum of block row I, so max_I B_I
// bounds |X|_inf from above (1.0 - 1.3x on the recorded trajectory) and every rule below holds a fortiori.  (|X|_F was tried:
// cheaper still -- three words -- but for a dense generator the final-sweep rule c |X|^2 <= 2e-14 needs the row sums, which
// are up to sqrt(n) |X|_F: tests/test_gpu_long_horizon.py lost a digit at 64 states.)  The sums are in FIXED POINT (units of
// 2^-52; integer sums commute, so the result does not depend on the order the tiles arrive in and a solve stays reproducible
// bit for bit), the cosine as the bits of a non-negative double under an integer maximum, spread over four lines.  No arrival
// counter, no fence: the kernel boundary publishes them, and the FIRST launch behind lge_gram (lge_so, or lge_gemm<EG_P2> in
// a slot without it) takes the decision itself in every wave -- one 8-byte load per lane next to its control words, five
// shuffle steps, ~40 vector instructions (a vector instruction executed by all 24 waves of a CU costs a launch 10 ns: the same
// decision from 400 exact row sums, ~300 instructions and two barriers, cost 3 us per launch -- as much as the launch it
// replaced); its workgroup 0 also leaves the decision in the control block for the launches that follow.
#define LGE_ACC_NT 32                                 // block rows a set has room for (LD <= 512)
#define LGE_ACC_COS 64                                // [0, 32): B_I all pairs, [32, 64): far pairs, [64 + 16 k]: cosine bits, k < 4
#define LGE_ACC_WORDS 128
#define LGE_ACC_UNIT 4503599627370496.0               // 2^52: a row sum is below 2^9
__device__ __forceinline__ unsigned long long lge_fix(double x) {   // x >= 0 (a tile's share: at most 16 x pi / 4)
  return x > 0.0 ? (unsigned long long)(fmin(x, 16.0) * LGE_ACC_UNIT) + 1ull : 0ull;
}

struct DecArgs {
  int on = 0;        // this launch takes the sweep's decision
  int cap = 0, nsq = 0, so = 0;
  double trigger = 0.0;
  const unsigned long long *acc = nullptr;
};

__device__ __forceinline__ double lge_lim(int order) {
  return order == 2 ? 1e-5 : order == 4 ? 2e-3 : order == 8 ? 0.06 : 0.5;
}

// One 16 x 16 tile of Gamma per workgroup, K split over its 8 waves; operands straight from L2 as 32-byte pieces of the
// columns (chunk j of a column = its rows 16 j .. 16 j + 15; lane (lo, hi) holds rows 16 j + 4 hi .. + 3 of column lo, so
// one double4 feeds four MFMA k-steps and a wave instruction reads 16 runs of 128 bytes).  The diagonal entries the tile
// needs (Gamma_mm of its 16 rows, Gamma_nn of its 16 columns) are summed from the same registers in a fixed order, the same
// order in the tile (n, m) -- X stays exactly antisymmetric.
__global__ __launch_bounds__(512, 6) void lge_gram(GramArgs a) {   // 6 waves per SIMD = 3 workgroups per CU: 768 slots
  __shared__ double sRed[8][256];
  __shared__ double sDg[2][32][16];
  __shared__ double sD[2][16];
  __shared__ double sStat[3][4];
  const unsigned long long *ctl = a.ctl;
  LGE_STAMP(ctl, 4);
  {
    lge_const_words cw = (lge_const_words)ctl;
    const unsigned long long w_stall = cw[EC_STALL], w_final = cw[EC_FINAL];
    if (w_stall != 0ull || (unsigned long long)a.slot > w_final) return;
  }
  const int LD = a.LD, nt = LD / 16;
  const int tm = blockIdx.x / nt, tn = blockIdx.x - tm * nt;
  const int m0 = tm * 16, n0 = tn * 16;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lo = lane & 15, hi = lane >> 4;
  const double *rowA = a.G + (size_t)(m0 + lo) * LD + 4 * hi, *rowB = a.G + (size_t)(n0 + lo) * LD + 4 * hi;
  // (two accumulators, not four: 96 registers would leave two workgroups per CU = 512 slots for the 625 tiles, i.e. two rounds)
  // (requesting the first chunks BEFORE the control words are looked at, as lge_gemm does, was tried in round 6: the kernel sits at
  // its 80-register budget and spilled 26)
  d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
  double da = 0.0, db = 0.0;
  const int nchunks = LD / 16;
  for (int j0 = wave; j0 < nchunks; j0 += 32) {   // four chunks of this wave in flight
    d4 av[4], bv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = min(j0 + 8 * u, nchunks - 1);
      av[u] = *reinterpret_cast<const d4 *>(rowA + 16 * j);
      bv[u] = *reinterpret_cast<const d4 *>(rowB + 16 * j);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (j0 + 8 * u < nchunks) {
        acc0 = mfma_f64(av[u][0], bv[u][0], acc0);
        acc1 = mfma_f64(av[u][1], bv[u][1], acc1);
        acc0 = mfma_f64(av[u][2], bv[u][2], acc0);
        acc1 = mfma_f64(av[u][3], bv[u][3], acc1);
        da += (av[u][0] * av[u][0] + av[u][1] * av[u][1]) + (av[u][2] * av[u][2] + av[u][3] * av[u][3]);
        db += (bv[u][0] * bv[u][0] + bv[u][1] * bv[u][1]) + (bv[u][2] * bv[u][2] + bv[u][3] * bv[u][3]);
      }
    }
  }
  const d4 acc = acc0 + acc1;
#pragma unroll
  for (int r = 0; r < 4; ++r) sRed[wave][r * 64 + lane] = acc[r];
  sDg[0][wave * 4 + hi][lo] = da;
  sDg[1][wave * 4 + hi][lo] = db;
  __syncthreads();
  // (the 32 partials of a diagonal entry: four interleaved chains of eight and a fixed tree over them -- one chain of 32 dependent
  // LDS reads was 3 us in the middle of every launch; the order is the same in the tile (n, m): X stays exactly antisymmetric)
  if (threadIdx.x < 32) {
    const int w = threadIdx.x >> 4, c = threadIdx.x & 15;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
    for (int p = 0; p < 32; p += 4) {
      s0 += sDg[w][p][c];
      s1 += sDg[w][p + 1][c];
      s2 += sDg[w][p + 2][c];
      s3 += sDg[w][p + 3][c];
    }
    sD[w][c] = (s0 + s1) + (s2 + s3);
  }
  __syncthreads();
  double mc2 = 0.0;
  if (threadIdx.x < 256) {
    const int t = threadIdx.x, r = t >> 6, l = t & 63;
    const int ri = (l >> 4) + 4 * r, ci = l & 15;
    const int row = m0 + ri, col = n0 + ci;
    double g = 0.0;
#pragma unroll
    for (int w = 0; w < 8; ++w) g += sRed[w][t];
    const double gii = sD[0][ri], gjj = sD[1][ci];
    double x = 0.0;
    const double g2 = g * g, ab = gii * gjj;
    if (!(g2 == g2) || !(ab == ab)) mc2 = INFINITY;   // NaN in G: reported as an infinite cosine (fmax drops NaNs)
    if (row != col && g2 > ab * (CB_JAC_SKIP * CB_JAC_SKIP)) {
      mc2 = fmax(mc2, g2 * fast_rcp(ab));
      const double d = gjj - gii;
      // The pair's Jacobi angle theta = atan(2 g / d) / 2, of which g / d is the small-angle limit: identical for the far
      // pairs (|g / d| ~ 1e-3: they differ by (g / d)^2 / 3), but BOUNDED by pi / 4 for a near-degenerate pair -- an isolated
      // one (d -> 0 at a cosine of 1e-4: g / d = 2.8 on the recorded trajectory) then no longer pushes the row sum over the
      // all-pairs limit and into a masked sweep that has no band pass planned behind it (a stalled solve), and exp(X) rotates
      // it by exactly its Jacobi angle.  (atan and fast_rcp are odd: X stays exactly antisymmetric.)
      x = d != 0.0 ? 0.5 * atan(2.0 * g * fast_rcp(d)) : (g > 0.0 ? 0.78539816339744831 : -0.78539816339744831);
    }
    const int bd = row / JB_W - col / JB_W;
    const bool far = bd > a.band || -bd > a.band;
    a.X[(size_t)row * LD + col] = x;
    a.Xf[(size_t)row * LD + col] = far ? x : 0.0;
    if (a.so) {
      a.Gm[(size_t)row * LD + col] = g;
      if (row == col) a.dg[row] = gii;
    }
    double rs = fabs(x), rsf = far ? fabs(x) : 0.0;
    // row sums over this tile's 16 columns (lanes with equal l >> 4), then the largest over the four rows of the wave
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) {
      rs += __shfl_xor(rs, m);
      rsf += __shfl_xor(rsf, m);
      mc2 = fmax(mc2, __shfl_xor(mc2, m));
    }
#pragma unroll
    for (int m = 16; m < 64; m <<= 1) {
      rs = fmax(rs, __shfl_xor(rs, m));
      rsf = fmax(rsf, __shfl_xor(rsf, m));
      mc2 = fmax(mc2, __shfl_xor(mc2, m));
    }
    if (l == 0) {
      sStat[0][r] = rs;
      sStat[1][r] = rsf;
      sStat[2][r] = mc2;
    }
  }
  __syncthreads();
  if (threadIdx.x < 3) {   // (one statistic per thread)
    const int k = threadIdx.x;
    const double v = fmax(fmax(sStat[k][0], sStat[k][1]), fmax(sStat[k][2], sStat[k][3]));
    if (k < 2) atomicAdd(a.acc + k * LGE_ACC_NT + tm, lge_fix(v));
    else atomicMax(a.acc + LGE_ACC_COS + 16 * (blockIdx.x & 3), dbl_bits(v));
  }
}

// The DECISION of a sweep from lge_gram's statistics, taken by every wave of the first launch behind it (no LDS, no barrier:
// lane I holds B_I, lane 32 + I the far-pair B_I).  Workgroup 0 publishes it -- control words and the solve's record -- for the
// launches that follow.  The caller requests the statistics (lge_acc_request) together with its control words: one memory round
// trip for both.
struct LgeDecision {
  unsigned long long mode, sq, sc_bits;
};
struct LgeAcc {
  unsigned long long w = 0ull, c2 = 0ull;
};
__device__ __forceinline__ LgeAcc lge_acc_request(const DecArgs &d) {
  LgeAcc q;
  if (d.on) {
    q.w = d.acc[threadIdx.x & 63];
    lge_const_words ar = (lge_const_words)d.acc;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned long long c = ar[LGE_ACC_COS + 16 * i];
      q.c2 = c > q.c2 ? c : q.c2;
    }
  }
  return q;
}
__device__ __forceinline__ double lge_uniform(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ LgeDecision lge_decide_here(int slot, const DecArgs &d, const LgeAcc &q, unsigned long long *ctl,
                                                       unsigned long long nsweep) {
  double v = (double)q.w * (1.0 / LGE_ACC_UNIT);
#pragma unroll
  for (int m = 1; m < 32; m <<= 1) v = fmax(v, __shfl_xor(v, m));
  const double rowsum = lge_uniform(v, 0), rowsum_far = lge_uniform(v, 32);
  double c2;
  memcpy(&c2, &q.c2, 8);
  LgeDecision o{0ull, 0ull, 0x3FF0000000000000ull};
  const bool bad = q.c2 >= 0x7FF0000000000000ull;   // (lge_gram stores +inf for a NaN)
  bool masked = false, fin = false;
  if (!bad) {
    // all pairs at once only when the state is close enough for the small-angle limit to hold for the near-degenerate
    // neighbours too (the rule of eigh_large_host.hip.h)
    // (row sums of Jacobi ANGLES: one near-degenerate pair contributes at most pi / 4, so the all-pairs limit is 1 -- a row
    // with two large angles, i.e. a cluster, is still sent to the band passes)
    masked = c2 > d.trigger * d.trigger || rowsum > 1.0;
    const double rsu = masked ? rowsum_far : rowsum;
    // second-order generator (lge_so): the sweep then converges cubically -- it ends at ~ c |X|^2 instead of ~ c |X| --,
    // so it is the last one already when c |X|^2 <= 2e-14 (what a first-order sweep from 1e-8 leaves at worst)
    // (and only below |X| = 0.3: the correction is worth a launch when the sweep is about to converge, and its own size is
    // O(|X|^2) only there -- the order and the squarings below are chosen from X1's norm)
    const bool so = d.so != 0 && !masked && c2 > 1e-16 && rowsum <= 0.3;
    const double r2 = rowsum * rowsum;
    fin = !masked && (c2 <= 1e-16 || (so && c2 * r2 * r2 <= 4e-28));   // starts below 1e-8: ends at rounding level
    int order = rsu <= lge_lim(2) ? 2 : rsu <= lge_lim(4) ? 4 : rsu <= lge_lim(8) ? 8 : 12;
    if (order > d.cap) order = d.cap;
    const double lim = lge_lim(order);
    double sc = 1.0;
    while (rsu * sc > lim) {
      sc *= 0.5;
      ++o.sq;
    }
    unsigned long long damped = 0ull;
    if (o.sq > (unsigned long long)d.nsq) {   // the slot cannot square that often: a partial (still orthogonal) rotation
      o.sq = (unsigned long long)d.nsq;
      sc = lim / rsu;
      damped = 1ull;
    }
    o.sc_bits = dbl_bits(sc);
    o.mode = (unsigned long long)order | (masked ? 256ull : 0ull) | (damped << 9) | (so ? 1024ull : 0ull) | (1ull << 16);
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    const unsigned long long k = nsweep;
    ctl[EC_NSWEEP] = k + 1;
    if (bad) {
      ctl[EC_ERR] = 2ull;
      ctl[EC_STALL] = 1ull;
    }
    if (fin) ctl[EC_FINAL] = (unsigned long long)slot;
    ctl[EC_MASKED] = masked ? 1ull : 0ull;
    ctl[EC_MODE] = o.mode;
    ctl[EC_SQ] = o.sq;
    ctl[EC_SC] = o.sc_bits;
    if (k < EC_MAXREC) {
      ctl[EC_TSWEEP + k] = __builtin_amdgcn_s_memrealtime();
      ctl[EC_REC + 4 * k + 0] = dbl_bits(sqrt(c2));
      ctl[EC_REC + 4 * k + 1] = dbl_bits(rowsum);
      ctl[EC_REC + 4 * k + 2] = dbl_bits(rowsum_far);
      ctl[EC_REC + 4 * k + 3] = o.mode | (o.sq << 24) | ((unsigned long long)slot << 32);
    }
  }
  return o;
}

#ifdef CB_DECIDE_KERNEL   // (experiment: the decision as a launch of its own again -- one wave, the same function)
__global__ void lge_decide_k(int slot, DecArgs d, unsigned long long *ctl) {
  const LgeAcc q = lge_acc_request(d);
  lge_const_words cw = (lge_const_words)ctl;
  const unsigned long long w_stall = cw[EC_STALL], w_final = cw[EC_FINAL], w_nsw = cw[EC_NSWEEP];
  if (w_stall != 0ull || (unsigned long long)slot > w_final) return;
  (void)lge_decide_here(slot, d, q, ctl, w_nsw);
}
#endif

#ifdef CB_NANCHECK
// (debugging aid: max_k |A u_k - lambda_k u_k|_inf of the decomposition a solve left, as double bits under an integer maximum in word 92)
__global__ void lge_residual(int LD, const double *A, const double *Ut, const double *lam, unsigned long long *ctl) {
  const int k = blockIdx.x;
  double m = 0.0;
  for (int i = threadIdx.x; i < LD; i += blockDim.x) {
    double r = -lam[k] * Ut[(size_t)k * LD + i];
    for (int j = 0; j < LD; ++j) r = fma(A[(size_t)i * LD + j], Ut[(size_t)k * LD + j], r);
    m = fmax(m, fabs(r));
    if (!(r == r)) m = INFINITY;
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(ctl + 92, dbl_bits(m));
}
#endif
#ifdef CB_NANCHECK
// (debugging aid: |R^T R - I|_max of a sweep's rotation; the largest value of the solve and its slot in words 91 / 90)
__global__ void lge_orthcheck(int LD, const double *R, unsigned long long slot, unsigned long long *ctl) {
  lge_const_words cw = (lge_const_words)ctl;
  if (cw[EC_STALL] != 0ull || slot > cw[EC_FINAL] || !(cw[EC_MODE] >> 16)) return;
  const int k = blockIdx.x;
  double m = 0.0;
  for (int j = threadIdx.x; j < LD; j += blockDim.x) {
    double d = j == k ? -1.0 : 0.0;
    for (int i = 0; i < LD; ++i) d = fma(R[(size_t)i * LD + k], R[(size_t)i * LD + j], d);
    m = fmax(m, fabs(d));
    if (!(d == d)) m = INFINITY;
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) {
    const unsigned long long old = atomicMax(ctl + 91, dbl_bits(m));
    if (dbl_bits(m) > old) ctl[90] = slot;
  }
}
#endif
#ifdef CB_NANCHECK   // (debugging aid: the first buffer of a solve that holds a non-finite entry, as slot * 100 + tag in word 94)
__global__ void lge_nancheck(int LD, const double *buf, unsigned long long tag, unsigned long long *ctl) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)LD * LD) return;
  const double v = buf[i];
  if (!(v - v == 0.0)) atomicCAS(ctl + 94, 0ull, tag);
}
#endif

// what a launch of lge_gemm is (the control block decides whether it runs and on which operands)
enum { EG_P2 = 0, EG_P34, EG_T1, EG_RP, EG_SQ, EG_GR, EG_R4 };

struct EgArgs {
  int LD, slot, kind, q;             // q: index of an EG_SQ launch
  int cap;                           // the slot's highest polynomial order: 4 (EG_P2, EG_R4, EG_GR) or 12
  int early;                         // the plan expects this launch to RUN: operands that do not depend on the decision are requested
                                     // before the control words are looked at (a launch that then returns has paid for 26 loads:
                                     // 6 us instead of 4.4 -- so launches the plan expects to return at once do not do this)
  const unsigned long long *ctl;
  const double *X, *Xf;              // the generator (all pairs / far pairs or second order)
  double *P2, *P4;
  double *B0, *B1, *B2;              // lge_p34's outputs (the polynomial's coefficient matrices); cap 4: B0 = s X / 6 - s^2 P2 / 24
  double *T;                         // EG_T1's output (8th order: lge_p34 writes it)
  double *R[2], *Rt[2];              // R_q lives in R[q & 1], its transpose in Rt[q & 1]
  double *Rfin;                      // the rotation of the sweep (whichever launch completes it writes it HERE: EG_GR's operand is static)
  const double *Gin;                 // EG_GR
  double *Gout;
  DecArgs dec;                       // EG_P2 of a slot without lge_so: the first launch behind lge_gram takes the decision
  unsigned long long *zacc;          // EG_GR: the statistics lines of this slot, cleared for the slot after next
};

// The polynomial coefficients of exp(Y), Y = s X, from X, P2 = -X^2, P3 = X^3, P4 = X^4 (elementwise):
//   order 4:  R_0 = I + Y + Y^2/2 + Y^3/6 + Y^4/24
//   order 8:  B0 = I + Y + Y^2/2 + Y^3/6,  T = I/4! + Y/5! + Y^2/6! + Y^3/7! + Y^4/8!          (R_0 = B0 + Y^4 T)
//   order 12: B0, B1 = I/4! + .. + Y^3/7!, B2 = I/8! + Y/9! + Y^2/10! + Y^3/11! + Y^4/12!      (R_0 = B0 + Y^4 (B1 + Y^4 B2))
// (round 5 had a launch of its own for them, lge_poly -- 5 us per sweep that needs them; now lge_p34's epilogue, which holds
// X^3 and X^4 of its tile, forms them)

// element [4 s + hi][c0 + lo] of a k-major LD x LD operand as a buffer load: resource = the matrix, scalar offset = the k-step's
// rows (wave-uniform), vector offset = the lane's 32-bit byte offset -- thirteen loads in flight cost thirteen data registers
__device__ __forceinline__ double lge_ld(const double *base, int s, int LD, int c0, unsigned lane_bytes) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(base), 0, 0x7fffffff, 0x00027000);
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)lane_bytes, (int)((unsigned)(4 * s * LD + c0) * 8u), 0));
}

// K was split over the 8 waves of a tile: fold the upper four partial tiles into the lower four, then sum those (fixed order)
__device__ __forceinline__ double lge_fold8(double (*sRed)[256], d4 acc, int wave, int lane, int et) {
  if (wave >= 4) {
#pragma unroll
    for (int r = 0; r < 4; ++r) sRed[wave - 4][r * 64 + lane] = acc[r];
  }
  __syncthreads();
  if (wave < 4) {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] += sRed[wave][r * 64 + lane];
  }
  __syncthreads();
  if (wave < 4) {
#pragma unroll
    for (int r = 0; r < 4; ++r) sRed[wave][r * 64 + lane] = acc[r];
  }
  __syncthreads();
  return (sRed[0][et] + sRed[1][et]) + (sRed[2][et] + sRed[3][et]);
}

// out[m][n] = sum_k Aop[k][m] Bop[k][n]: one 16 x 16 tile per workgroup, K over 8 waves, with the operands, the epilogue and
// the decision to run at all taken from the control block.  One instantiation per kind.
// A launch is a chain of dependent memory latencies (~1 us each behind a kernel boundary, which leaves the L2 cold), not
// arithmetic; round 6 took three links out of it: ALL k-steps of a wave are in flight at once (13 at LD = 400; round 5: two
// batches of 7), operands that do not depend on the decision (P4, B2, T, R_q, G, what the epilogue adds) are requested BEFORE the
// control words are looked at, and the epilogue's terms in front of the K loop instead of behind the reduction.
#define LGE_UU 13
template <int KIND>
__global__ __launch_bounds__(512) void lge_gemm(EgArgs a) {
  __shared__ double sRed[4][256];
  const unsigned long long *ctl = a.ctl;
  LGE_STAMP(ctl, 10 + KIND);
  const int LD = a.LD, nt = LD / 16;
  const int tm = blockIdx.x / nt, tn = blockIdx.x - tm * nt;
  const int m0 = tm * 16, n0 = tn * 16;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, lo = lane & 15, hi = lane >> 4;
  const int nsteps = LD / 4;
  const unsigned loff = (unsigned)(hi * LD + lo) * 8u;
  const int et = threadIdx.x & 255, er = et >> 6, el = et & 63;
  const int row = m0 + (el >> 4) + 4 * er, col = n0 + (el & 15);
  const size_t idx = (size_t)row * LD + col;
  const bool ep = threadIdx.x < 256;
  // ---- operands known without the control block
  const double *Ap = nullptr, *Bp = nullptr, *Ep = nullptr;
  if (KIND == EG_T1) { Ap = a.P4; Bp = a.B2; Ep = a.B1; }
  if (KIND == EG_RP) { Ap = a.P4; Bp = a.T; Ep = a.B0; }
  if (KIND == EG_SQ) { Ap = a.Rt[a.q & 1]; Bp = a.R[a.q & 1]; }
  if (KIND == EG_GR) { Ap = a.Rfin; Bp = a.Gin; }
  if (KIND == EG_R4) { Ap = a.P2; Bp = a.B0; Ep = a.P2; }
  double av[LGE_UU], bv[LGE_UU], e1 = 0.0, ex = 0.0;
  const bool early = KIND != EG_P2 && a.early != 0;
  if (early) {
#pragma unroll
    for (int u = 0; u < LGE_UU; ++u) {
      const int sk = min(wave + 8 * u, nsteps - 1);
      av[u] = lge_ld(Ap, sk, LD, m0, loff);
      bv[u] = lge_ld(Bp, sk, LD, n0, loff);
    }
    if (Ep && ep) e1 = Ep[idx];
  }
  // ---- the decision: five words of one cache line, requested BEHIND the early operand loads (scalar loads return out of order:
  // a wait for the kernel's own arguments would be a wait for every scalar load issued before it)
  LgeAcc accq;
  if (KIND == EG_P2) accq = lge_acc_request(a.dec);
  // (through the constant address space: ONE batch of scalar loads.  The deciding launch also writes the control block -- its
  // workgroup 0, after this point; a workgroup that starts late and sees the new words reads the same decision or words it
  // does not use -- and as plain global loads the compiler makes three dependent round trips of the tests below: 1.9 us)
  lge_const_words cw = (lge_const_words)ctl;
  const unsigned long long w_stall = cw[EC_STALL], w_final = cw[EC_FINAL], w_nsw = cw[EC_NSWEEP];
  unsigned long long mode = cw[EC_MODE], w_sq = cw[EC_SQ], w_sc = cw[EC_SC];
  if (w_stall != 0ull || (unsigned long long)a.slot > w_final) return;
  if (KIND == EG_P2 && a.dec.on) {
    const LgeDecision dd = lge_decide_here(a.slot, a.dec, accq, const_cast<unsigned long long *>(ctl), w_nsw);
    mode = dd.mode;
    w_sq = dd.sq;
    w_sc = dd.sc_bits;
  }
  if (!(mode >> 16)) return;
  if (KIND == EG_GR && a.zacc && blockIdx.x == 0 && threadIdx.x < LGE_ACC_WORDS)   // (every reader of this slot's statistics has finished)
    a.zacc[threadIdx.x] = 0ull;
  const int order = (int)(mode & 255ull);
  const bool masked = (mode & (256ull | 1024ull)) != 0ull;   // far pairs only, OR the second-order generator: both live in Xf
  const int sq = (int)w_sq;
  double sc;
  memcpy(&sc, &w_sc, 8);
  const double *Xu = masked ? a.Xf : a.X;
  if (KIND == EG_R4 && order != 4) return;
  if (KIND == EG_T1 && order != 12) return;
  if (KIND == EG_RP && order < 8) return;      // (order 4 in a slot with the X^3 / X^4 launch: lge_p34 has written the rotation)
  if (KIND == EG_SQ && a.q >= sq) return;
  if (!early) {
    const double *pa = KIND == EG_P2 ? Xu : Ap, *pb = KIND == EG_P2 ? Xu : Bp;
#pragma unroll
    for (int u = 0; u < LGE_UU; ++u) {
      const int sk = min(wave + 8 * u, nsteps - 1);
      av[u] = lge_ld(pa, sk, LD, m0, loff);
      bv[u] = lge_ld(pb, sk, LD, n0, loff);
    }
    if (KIND != EG_P2 && Ep && ep) e1 = Ep[idx];
  }
  if ((KIND == EG_P2 || KIND == EG_R4) && ep) ex = Xu[idx];
  d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int u = 0; u < LGE_UU; ++u)
    if (wave + 8 * u < nsteps) acc = mfma_f64(av[u], bv[u], acc);
  for (int s0 = wave + 8 * LGE_UU; s0 < nsteps; s0 += 8 * LGE_UU) {   // (LD > 416)
    const double *pa = KIND == EG_P2 ? Xu : Ap, *pb = KIND == EG_P2 ? Xu : Bp;
#pragma unroll
    for (int u = 0; u < LGE_UU; ++u) {
      const int sk = min(s0 + 8 * u, nsteps - 1);
      av[u] = lge_ld(pa, sk, LD, m0, loff);
      bv[u] = lge_ld(pb, sk, LD, n0, loff);
    }
#pragma unroll
    for (int u = 0; u < LGE_UU; ++u)
      if (s0 + 8 * u < nsteps) acc = mfma_f64(av[u], bv[u], acc);
  }
  const double v = lge_fold8(sRed, acc, wave, lane, et);
  if (!ep) return;
  const double edl = row == col ? 1.0 : 0.0, s2 = sc * sc;
  if (KIND == EG_P2) {
    // X^T X = -X^2.  Second order: R = I + Y + Y^2 / 2 = I + s X - s^2 X^T X / 2 at once (s < 1: a damped sweep); a slot without
    // the X^3 / X^4 launch also leaves the second operand of its fourth-order product, s X / 6 - s^2 P2 / 24
    if (order == 2) {
      a.Rfin[idx] = fma(-0.5 * s2, v, fma(sc, ex, edl));
    } else {
      a.P2[idx] = v;
      if (a.cap == 4) a.B0[idx] = fma(-s2 * (1.0 / 24.0), v, (sc * (1.0 / 6.0)) * ex);
    }
  } else if (KIND == EG_R4) {
    // fourth order in one product:  R = I + Y + Y^2/2 + Y^2 (Y/6 + Y^2/24),  Y = s X,  Y^2 = -s^2 P2:  acc = P2 (s X / 6 - s^2 P2 / 24)
    a.Rfin[idx] = fma(-s2, v, fma(-0.5 * s2, e1, fma(sc, ex, edl)));
  } else if (KIND == EG_T1) {
    a.T[idx] = fma(s2 * s2, v, e1);                      // T = B1 + Y^4 B2   (Y^4 = s^4 P4, symmetric)
  } else if (KIND == EG_RP) {
    const double o = fma(s2 * s2, v, e1);                // R_0 = B0 + Y^4 T
    if (sq == 0) a.Rfin[idx] = o;
    else {
      a.R[0][idx] = o;
      a.Rt[0][(size_t)col * LD + row] = o;
    }
  } else if (KIND == EG_SQ) {                            // R_{q+1} = R_q R_q
    if (a.q == sq - 1) a.Rfin[idx] = v;
    else {
      a.R[(a.q + 1) & 1][idx] = v;
      a.Rt[(a.q + 1) & 1][(size_t)col * LD + row] = v;
    }
  } else {
    a.Gout[idx] = v;                                     // EG_GR: Gout[c'][r] = sum_c R[c][c'] Gin[c][r]
  }
}

// X^3 = X^T P2 and X^4 = P2^T P2 of one 16 x 16 tile in ONE workgroup (waves 0-3: X^3, waves 4-7: X^4, K over four waves each, two
// batches of 13 / 12 k-steps), then the polynomial's coefficient matrices from the four powers the workgroup now holds for its
// tile: B0, B1, B2 (order 12), B0, T (order 8) or the finished rotation (order 4).  The operands that are P2 are requested before
// the control words are looked at.
__global__ __launch_bounds__(512) void lge_p34(EgArgs a) {
  __shared__ double sRed[8][256];
  const unsigned long long *ctl = a.ctl;
  LGE_STAMP(ctl, 17);
  const int LD = a.LD, nt = LD / 16;
  const int tm = blockIdx.x / nt, tn = blockIdx.x - tm * nt;
  const int m0 = tm * 16, n0 = tn * 16;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, lo = lane & 15, hi = lane >> 4;
  const int w4 = wave & 3;
  const bool fourth = wave >= 4;   // (wave-uniform)
  const int nsteps = LD / 4;
  const unsigned loff = (unsigned)(hi * LD + lo) * 8u;
  const int et = threadIdx.x & 255, er = et >> 6, el = et & 63;
  const int row = m0 + (el >> 4) + 4 * er, col = n0 + (el & 15);
  const size_t idx = (size_t)row * LD + col;
  const bool ep = threadIdx.x < 256;
  double av[LGE_UU], bv[LGE_UU], e2 = 0.0;
  const bool early = a.early != 0;
  if (early) {
#pragma unroll
    for (int u = 0; u < LGE_UU; ++u) {
      const int sk = min(w4 + 4 * u, nsteps - 1);
      bv[u] = lge_ld(a.P2, sk, LD, n0, loff);
      if (fourth) av[u] = lge_ld(a.P2, sk, LD, m0, loff);
    }
    if (ep) e2 = a.P2[idx];
  }
  lge_const_words cw = (lge_const_words)ctl;
  const unsigned long long w_stall = cw[EC_STALL], w_final = cw[EC_FINAL], mode = cw[EC_MODE], w_sc = cw[EC_SC];
  if (w_stall != 0ull || (unsigned long long)a.slot > w_final) return;
  if (!(mode >> 16)) return;
  const int order = (int)(mode & 255ull);
  if (order < 4) return;
  const bool masked = (mode & (256ull | 1024ull)) != 0ull;
  double sc;
  memcpy(&sc, &w_sc, 8);
  const double *Xu = masked ? a.Xf : a.X;
  const double *Ap = fourth ? a.P2 : Xu;
  if (!early) {
#pragma unroll
    for (int u = 0; u < LGE_UU; ++u) {
      const int sk = min(w4 + 4 * u, nsteps - 1);
      bv[u] = lge_ld(a.P2, sk, LD, n0, loff);
      av[u] = lge_ld(Ap, sk, LD, m0, loff);
    }
    if (ep) e2 = a.P2[idx];
  } else if (!fourth) {
#pragma unroll
    for (int u = 0; u < LGE_UU; ++u) av[u] = lge_ld(Xu, min(w4 + 4 * u, nsteps - 1), LD, m0, loff);
  }
  double ex = 0.0;
  if (ep) ex = Xu[idx];
  d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int u = 0; u < LGE_UU; ++u)
    if (w4 + 4 * u < nsteps) acc = mfma_f64(av[u], bv[u], acc);
  for (int s0 = w4 + 4 * LGE_UU; s0 < nsteps; s0 += 4 * LGE_UU) {
#pragma unroll
    for (int u = 0; u < LGE_UU; ++u) {
      const int sk = min(s0 + 4 * u, nsteps - 1);
      av[u] = lge_ld(Ap, sk, LD, m0, loff);
      bv[u] = lge_ld(a.P2, sk, LD, n0, loff);
    }
#pragma unroll
    for (int u = 0; u < LGE_UU; ++u)
      if (s0 + 4 * u < nsteps) acc = mfma_f64(av[u], bv[u], acc);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) sRed[wave][r * 64 + lane] = acc[r];
  __syncthreads();
  if (!ep) return;
  const double p3 = (sRed[0][et] + sRed[1][et]) + (sRed[2][et] + sRed[3][et]);
  const double p4 = (sRed[4][et] + sRed[5][et]) + (sRed[6][et] + sRed[7][et]);
  const double dl = row == col ? 1.0 : 0.0;
  const double s2 = sc * sc;
  const double y1 = sc * ex, y2 = -s2 * e2, y3 = s2 * sc * p3, y4 = s2 * s2 * p4;
  const double b0 = dl + y1 + 0.5 * y2 + y3 * (1.0 / 6.0);
  if (order == 4) {
    a.Rfin[idx] = b0 + y4 * (1.0 / 24.0);
    return;
  }
  a.P4[idx] = p4;
  a.B0[idx] = b0;
  const double b1 = dl * (1.0 / 24.0) + y1 * (1.0 / 120.0) + y2 * (1.0 / 720.0) + y3 * (1.0 / 5040.0);
  if (order == 8) {
    a.T[idx] = b1 + y4 * (1.0 / 40320.0);
    return;
  }
  a.B1[idx] = b1;
  a.B2[idx] = dl * (1.0 / 40320.0) + y1 * (1.0 / 362880.0) + y2 * (1.0 / 3628800.0) + y3 * (1.0 / 39916800.0) + y4 * (1.0 / 479001600.0);
}

// Second-order generator of an all-pairs sweep.  With Gamma = D + E (E off-diagonal) the rotation exp(X) that diagonalises
// it satisfies, order by order (e^-X Gamma e^X = Gamma + [Gamma, X] + [[Gamma, X], X] / 2 + ..):
//     X1_mn = E_mn / (D_n - D_m)                              (lge_gram: the small-angle limit of every pair's rotation)
//     X2_mn = [E, X1]_mn / (2 (D_n - D_m)),   [E, X1] = E X1 - X1 E = (E X1) + (E X1)^T
// A first-order sweep leaves cosines ~ c |X|; with X = X1 + X2 it leaves ~ c |X|^2: on the recorded bench trajectory
// (profiles/tools/eigh_proto.py) 1.9 all-pairs sweeps per solve instead of 2.8, for ONE more launch per sweep: both products
// of the commutator on the same 16 x 16 tile (out[m][n] = sum_k E[k][m] X[k][n] + X[k][m] E[k][n], E symmetric, X
// antisymmetric), the division in the epilogue.  The corrected generator goes to Xf, which an all-pairs sweep does not
// use, and the products that follow take it from there (EC_MODE bit 10).
struct SoArgs {
  int LD, slot;
  const unsigned long long *ctl;
  const double *Gm, *dg, *X;
  double *Xs;   // = Xf
  int early;    // the plan expects the launch to run: its first operands are requested before the control block is looked at
  DecArgs dec;  // the first launch behind lge_gram: takes the sweep's decision
};

__global__ __launch_bounds__(512) void lge_so(SoArgs a) {
  __shared__ double sRed[4][256];
  const unsigned long long *ctl = a.ctl;
  LGE_STAMP(ctl, 6);
  const int LD = a.LD, nt = LD / 16;
  const int tm = blockIdx.x / nt, tn = blockIdx.x - tm * nt;
  const int m0 = tm * 16, n0 = tn * 16;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lo = lane & 15, hi = lane >> 4;
  d4 accA = {0.0, 0.0, 0.0, 0.0}, accB = accA;
  const int nsteps = LD / 4;
  const double *Em = a.Gm + m0 + lo, *En = a.Gm + n0 + lo, *Xm = a.X + m0 + lo, *Xn = a.X + n0 + lo;
  double e1[7], x1[7], x2[7], e2[7];
  auto issue = [&](int s0) {
#pragma unroll
    for (int u = 0; u < 7; ++u) {
      const int s = min(s0 + 8 * u, nsteps - 1);
      const int k = 4 * s + hi;
      const size_t krow = (size_t)k * LD;
      e1[u] = k == m0 + lo ? 0.0 : Em[krow];   // E = Gamma without its diagonal
      x1[u] = Xn[krow];
      x2[u] = Xm[krow];
      e2[u] = k == n0 + lo ? 0.0 : En[krow];
    }
  };
  if (a.early) issue(wave);
  // (the control words are requested BEHIND the early loads: see lge_gemm)
  const LgeAcc accq = lge_acc_request(a.dec);
  lge_const_words cw = (lge_const_words)ctl;   // (one batch of scalar loads: see lge_gemm)
  const unsigned long long w_stall = cw[EC_STALL], w_final = cw[EC_FINAL], w_nsw = cw[EC_NSWEEP];
  unsigned long long w_mode = cw[EC_MODE];
  if (w_stall != 0ull || (unsigned long long)a.slot > w_final) return;
  if (a.dec.on) w_mode = lge_decide_here(a.slot, a.dec, accq, const_cast<unsigned long long *>(ctl), w_nsw).mode;
  if (!(w_mode & 1024ull)) return;
  for (int s0 = wave; s0 < nsteps; s0 += 7 * 8) {   // seven k-steps of this wave in flight
    if (!(a.early && s0 == wave)) issue(s0);
#pragma unroll
    for (int u = 0; u < 7; ++u)
      if (s0 + 8 * u < nsteps) {
        accA = mfma_f64(e1[u], x1[u], accA);
        accB = mfma_f64(x2[u], e2[u], accB);
      }
  }
  d4 acc = accA + accB;   // (the tile (n, m) holds the same two sums with their roles exchanged: [E, X] stays exactly symmetric)
  if (wave >= 4) {
#pragma unroll
    for (int r = 0; r < 4; ++r) sRed[wave - 4][r * 64 + lane] = acc[r];
  }
  __syncthreads();
  if (wave < 4) {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] += sRed[wave][r * 64 + lane];
  }
  __syncthreads();
  if (wave < 4) {
#pragma unroll
    for (int r = 0; r < 4; ++r) sRed[wave][r * 64 + lane] = acc[r];
  }
  __syncthreads();
  if (threadIdx.x >= 256) return;
  const int t = threadIdx.x, r = t >> 6, l = t & 63;
  const int row = m0 + (l >> 4) + 4 * r, col = n0 + (l & 15);
  const double v = (sRed[0][t] + sRed[1][t]) + (sRed[2][t] + sRed[3][t]);
  const size_t idx = (size_t)row * LD + col;
  const double x1v = a.X[idx];
  const double d = a.dg[col] - a.dg[row];
  // (pairs lge_gram left alone -- orthogonal to rounding, or exactly degenerate -- stay as they are)
  // The pair's angle with the second-order coupling in it: x1 = atan(2 g / d) / 2 (lge_gram), so X1 + X2 in the small-angle limit
  // is (g + v / 2) / d -- taken through the SAME bounded form, atan(2 (g + v / 2) / d) / 2.  The plain sum x1 + v / (2 d) has no
  // bound: a near-degenerate pair (d -> 0) coupled through third columns got angles of several radians, four times the norm the
  // sweep's polynomial order was chosen for, and left a rotation 1e-6 from orthogonal (tests/test_gpu_s400_full.py, config 5 in
  // mixed precision, epoch 92 -- found in round 6; the debug build -DCB_NANCHECK prints |R^T R - I| of every sweep).
  // (g = Gamma_mn comes from the tile already in registers' reach: E = Gamma off the diagonal; atan and fast_rcp are odd and
  // v is symmetric: X stays exactly antisymmetric)
  double xs = x1v;
  if (row != col && x1v != 0.0 && d != 0.0) xs = 0.5 * atan(2.0 * (a.Gm[idx] + 0.5 * v) * fast_rcp(d));
  a.Xs[idx] = xs;
}

// A clean control block and clean statistics lines (blockDim.x == 256): the planned solve's prologue.  In the trainer it rides on
// lt_build (train_large.hip.h), which also leaves sigma = max |A_ii|; lge_begin is the launch of its own for every other caller.
__device__ void lge_reset_words(unsigned long long *ctl, unsigned long long *acc) {
  if (threadIdx.x < EC_WORDS)
    ctl[threadIdx.x] = threadIdx.x == EC_FINAL ? EC_NONE : threadIdx.x == EC_T0 ? (unsigned long long)__builtin_amdgcn_s_memrealtime() : 0ull;
  if (threadIdx.x < 2 * LGE_ACC_WORDS) acc[threadIdx.x] = 0ull;   // both sets of statistics lines
}
__global__ void lge_begin(int LD, const double *A, double *sigma, unsigned long long *ctl, unsigned long long *acc) {
  __shared__ double s[256];
  lge_reset_words(ctl, acc);
  double m = 0.0;
  for (int i = threadIdx.x; i < LD; i += 256) m = fmax(m, fabs(A[(size_t)i * LD + i]));
  s[threadIdx.x] = m;
  __syncthreads();
  for (int st = 128; st >= 1; st >>= 1) {
    if ((int)threadIdx.x < st) s[threadIdx.x] = fmax(s[threadIdx.x], s[threadIdx.x + st]);
    __syncthreads();
  }
  if (threadIdx.x == 0) *sigma = s[0] > 0.0 ? s[0] : 1.0;
}

// out[m][n] = sum_k Aop[k][m] Bop[k][n] (- (*sigma) Sub[m][n]): lge_gemm's tile (16 x 16, K over eight waves, all 13 k-steps of a wave
// in flight) for products whose operands are known at launch -- the warm start G = A' U_prev (Gc[k][r] = sum_j U_prev[j][k] A[j][r]
// - sigma Ut_prev[k][r]) and the two products that turn the bank's sum into dL/dA.  Nothing decides which operands: they are
// requested at once, and the word that says "the solve in front stalled: return" (skip) is looked at while they are in flight.
// (Rounds 2-5: the generic single-matrix product sg_gemm, two batches of seven k-steps behind its skip word: 8.5 us; this: 6.5.)
__global__ __launch_bounds__(512) void lge_plain(int LD, const double *Aop, const double *Bop, const double *Sub, const double *sigma,
                                                 double *out, const unsigned long long *skip) {
  __shared__ double sRed[4][256];
  const int nt = LD / 16;
  const int tm = blockIdx.x / nt, tn = blockIdx.x - tm * nt;
  const int m0 = tm * 16, n0 = tn * 16;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, lo = lane & 15, hi = lane >> 4;
  const int nsteps = LD / 4;
  const unsigned loff = (unsigned)(hi * LD + lo) * 8u;
  const int et = threadIdx.x & 255, er = et >> 6, el = et & 63;
  const int row = m0 + (el >> 4) + 4 * er, col = n0 + (el & 15);
  const size_t idx = (size_t)row * LD + col;
  double av[LGE_UU], bv[LGE_UU];
#pragma unroll
  for (int u = 0; u < LGE_UU; ++u) {
    const int sk = min(wave + 8 * u, nsteps - 1);
    av[u] = lge_ld(Aop, sk, LD, m0, loff);
    bv[u] = lge_ld(Bop, sk, LD, n0, loff);
  }
  const double e1 = (Sub && threadIdx.x < 256) ? Sub[idx] : 0.0;
  const double sg = sigma ? *sigma : 0.0;
  if (skip && *(lge_const_words)skip != 0ull) return;
  d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int u = 0; u < LGE_UU; ++u)
    if (wave + 8 * u < nsteps) acc = mfma_f64(av[u], bv[u], acc);
  for (int s0 = wave + 8 * LGE_UU; s0 < nsteps; s0 += 8 * LGE_UU) {   // (LD > 416)
#pragma unroll
    for (int u = 0; u < LGE_UU; ++u) {
      const int sk = min(s0 + 8 * u, nsteps - 1);
      av[u] = lge_ld(Aop, sk, LD, m0, loff);
      bv[u] = lge_ld(Bop, sk, LD, n0, loff);
    }
#pragma unroll
    for (int u = 0; u < LGE_UU; ++u)
      if (s0 + 8 * u < nsteps) acc = mfma_f64(av[u], bv[u], acc);
  }
  const double v = lge_fold8(sRed, acc, wave, lane, et);
  if (threadIdx.x < 256) out[idx] = fma(-sg, e1, v);
}

// A stalled solve continues (more slots on the same G): the stall word cleared, everything else kept.
__global__ void lge_resume(unsigned long long *ctl) {
  if (threadIdx.x == 0 && ctl[EC_ERR] == 0ull) ctl[EC_STALL] = 0ull;
}

// |g_k| per column of the buffer the final sweep wrote; workgroup 0 also settles the solve's outcome (STALL when the plan
// ended before a sweep started below 1e-8) and publishes the record of the solve to pinned host memory, where the host
// looks BEHIND the kernels it has already enqueued (one epoch of the bank is queued at that point).
// tb_rho_max > 0: the bank behind this solve runs in a time basis that serves spectra inside [-tb_rho_max, 0]; 2 sigma bounds
// the spectral radius (Gershgorin on the rate matrix), so 2 sigma > tb_rho_max makes the bank return at once (EC_SKIP) and
// the host repeat the evaluation with per-bucket products (train_host.hip.h).
__global__ void lge_norms(int LD, const double *G0, const double *G1, double *nrm, unsigned long long *ctl,
                          volatile unsigned long long *pin, unsigned long long seq, const double *sigma, double tb_rho_max) {
  LGE_STAMP(ctl, 20);
  lge_const_words cw = (lge_const_words)ctl;
  const unsigned long long fin = cw[EC_FINAL], w_stall = cw[EC_STALL];
  const bool stall = w_stall != 0ull || fin == EC_NONE;
  const bool stale = tb_rho_max > 0.0 && !(2.0 * (*sigma) <= tb_rho_max);
  // (the record for the host is written by an extra workgroup of lge_finish, the launch behind this one: its writes to pinned
  // host memory and their two system-scope fences kept this launch's workgroup 0 -- and with it the launch -- 2 us longer)
  (void)pin;
  (void)seq;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    ctl[EC_TBSTALE] = stale ? 1ull : 0ull;
    ctl[EC_SKIP] = (stall || stale) ? 1ull : 0ull;
    if (stall) ctl[EC_STALL] = 1ull;
  }
  if (stall) return;
  const double *Gc = ((fin + 1) & 1ull) ? G1 : G0;
  const int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (k >= LD) return;
  double nn = 0.0;
  for (int r = lane; r < LD; r += 64) {
    const double v = Gc[(size_t)k * LD + r];
    nn = fma(v, v, nn);
  }
  nn = wave_sum(nn);
  if (lane == 0) nrm[k] = sqrt(nn);
}

// The spectral tables of a time-basis bank (tbasis.hip.h, tb_tables) depend on lambda_k only in column k: the wave that places
// eigenvalue k writes them too, and the bank behind a planned solve needs no table launch of its own.
struct TbTableArgs {
  int ns = 0, nd = 0, ng = 0;            // all zero: no tables
  const double *tf = nullptr, *tg = nullptr;
  double *F = nullptr, *E = nullptr, *H = nullptr;
};
__device__ __forceinline__ void tb_table_column(const TbTableArgs &t, int LD, int k, double lk, int lane) {
  const int nf = t.ns + t.nd;
  for (int r = lane; r < nf + t.ng; r += 64) {
    if (r < nf) {
      const double tt = t.tf[r], x = tt * lk;
      t.F[(size_t)r * LD + k] = r < t.ns ? phi2(x) / (tt * tt) : exp(x);
    } else {
      const double x = t.tg[r - nf] * lk;
      t.E[(size_t)(r - nf) * LD + k] = exp(x);
      t.H[(size_t)(r - nf) * LD + k] = exp(0.5 * x);
    }
  }
}

// lgj_finish on the buffer of the final sweep; leaves U / lambda alone after a stall
__global__ void lge_finish(int LD, const double *G0, const double *G1, const double *nrm, const double *sigma, double *lam,
                           double *U, double *Ut, unsigned long long *ctl, TbTableArgs tb, volatile unsigned long long *pin,
                           unsigned long long seq) {
  lge_const_words cw = (lge_const_words)ctl;
  const unsigned long long w_stall = cw[EC_STALL], w_final = cw[EC_FINAL];
  if (blockIdx.x == gridDim.x - 1) {   // the extra workgroup: the solve's record for the host (blockDim.x == 256)
    if (pin && threadIdx.x < EC_WORDS) {
      const int i = threadIdx.x;
      pin[i] = i == EC_TEND ? (unsigned long long)__builtin_amdgcn_s_memrealtime() : i == EC_SIGMA ? dbl_bits(*sigma) : ctl[i];
      __threadfence_system();
#ifdef CB_NANCHECK
      if (i == 92) ctl[92] = 0ull;
#endif
    }
    __syncthreads();
    if (pin && threadIdx.x == 0) {
      pin[EC_WORDS] = seq;   // the word the host watches, written last
      __threadfence_system();
    }
    return;
  }
  if (w_stall != 0ull) return;
  const double *Gc = ((w_final + 1) & 1ull) ? G1 : G0;
  const int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (k >= LD) return;
  const double mine = nrm[k];
  // (the column is requested together with the norms -- where it goes is decided by them, what it holds is not: one round trip
  // instead of two in a row; columns longer than 512 take the rest afterwards)
  double gcol[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) gcol[u] = lane + 64 * u < LD ? Gc[(size_t)k * LD + lane + 64 * u] : 0.0;
  int before = 0;
  for (int j = lane; j < LD; j += 64) {
    const double o = nrm[j];
    before += (o > mine || (o == mine && j < k)) ? 1 : 0;
  }
  const int pos = (int)wave_sum((double)before);
  const double inv = -1.0 / mine;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int r = lane + 64 * u;
    if (r < LD) {
      const double v = gcol[u] * inv;
      Ut[(size_t)pos * LD + r] = v;
      U[(size_t)r * LD + pos] = v;
    }
  }
  for (int r = lane + 512; r < LD; r += 64) {
    const double v = Gc[(size_t)k * LD + r] * inv;
    Ut[(size_t)pos * LD + r] = v;
    U[(size_t)r * LD + pos] = v;
  }
  const double lk = *sigma - mine;
  if (lane == 0) lam[pos] = lk;
  if (tb.ns + tb.nd + tb.ng > 0) tb_table_column(tb, LD, pos, lk, lane);
}
