// Held-out log-likelihood by Felsenstein pruning in log space (reference:
// cherryml/evaluation/_likelihood.py:47-327, `dp_likelihood_computation`).
//
// The reference walks the tree in post-order, one python loop per (node, site), accumulating
//   dp[parent] += log(max(P_v (exp(dp[v] - m) * obs_v), 0)) + m,   m = max(dp[v]).
// Here the walk is LEVEL-SYNCHRONOUS: nodes of equal height are independent, so each height is one
// launch over (node of that height) x (block of units); a node's upward message
//   msg[v] = log(max(P_v w_v, 0)) + m_v,  w_v = exp(d_v - m_v) * obs_v,  d_v = sum_{children c} msg[c]
// is written once and summed by its parent in the reference's child order (bitwise the same dp sums).
// A "unit" is an independent site (S states, its own rate category) or a contacting pair of sites
// (S = S1 * S1 states, state = a * S1 + b).
//
//   tl_group_kernel  S <= 64: lane = (unit in wave, state row); w through LDS; P rows read directly
//   tl_mfma_kernel   S  > 64 (the 400-state pair model): 32 units per workgroup, arg = P_v W as
//                    v_mfma_f64_16x16x4 tiles (rows x units), W staged in LDS, P streamed from HBM
//   tl_leaf_kernel   S  > 64, leaves: W is a 0/1 vector, P_v W gathered from P_v's columns
#pragma once
#include "common.hip.h"
#include "common.hip.h"   // xcd_swizzle, mfma_f64

struct TlArgs {
  int S, S1;                   // states; S1 > 0: pair model over an S1-letter alphabet
  int n_nodes, n_units, NU;    // NU: units padded to the message layout's unit stride
  int root, n_level, n_blocks;  // this launch: nodes of one height x unit blocks (1-D grid)
  int RS;                      // tl_mfma_kernel: row splits per (node, unit block)
  const int *level_nodes;      // nodes of the height processed by this launch
  const int *child_ptr, *child_idx;   // CSR children, in the reference's child order
  const double *P;             // [cat][node][S][S] transition matrices of the edge above `node`
  const int *unit_cat;         // [n_units] rate category of each unit
  const signed char *code_a, *code_b;  // [node][unit] observed state (-1: unobserved); leaves only
  const double *pi_root;       // [S]
  double *msg;                 // upward messages (layout per kernel)
  double *ll;                  // [n_units]
  // S > 64 with a reversible model (round 6): the bank holds INTERNAL nodes only -- slot[v] (-1: none) -- and the leaves take their
  // messages from the model's eigendecomposition (tl_leaf_mfma_kernel)
  const int *slot;             // [n_nodes] or null (P indexed by node)
  const double *tnode;         // [n_nodes] rate x branch length above the node
  const double *U, *lam, *dsq, *sigma;   // U [LD][LD] row-major, lam [LD], dsq = sqrt(pi) [LD], sigma = max |A_ii|
  const double *TU, *TA;       // [nJ][LD]: sum_{j in J} U[j][k] d_j  and  sum_{j in J} A[i][j] d_j  (tl_tables_kernel)
  int LD;
};

__device__ __forceinline__ bool tl_observed(int S1, int k, int ca, int cb) {
  if (S1 > 0) return (ca < 0 || k / S1 == ca) && (cb < 0 || k % S1 == cb);
  return ca < 0 || ca == k;
}

// ------------------------------------------------------------------ S <= 64
// grid = nodes of the level x unit blocks, 64 threads.  msg layout [node][unit][S].
__global__ __launch_bounds__(64) void tl_group_kernel(TlArgs a) {
  __shared__ double sw[64];
  const int S = a.S, upw = 64 / S;
  const int g = threadIdx.x / S, r = threadIdx.x - g * S;
  const int node_i = blockIdx.x / a.n_blocks, blk = blockIdx.x - node_i * a.n_blocks;
  const int u = blk * upw + g;
  const bool act = g < upw && u < a.n_units;
  const int uu = act ? u : 0, gb = act ? g * S : 0;
  const int v = a.level_nodes[node_i];
  const int c0 = a.child_ptr[v], c1 = a.child_ptr[v + 1];
  double d = 0.0;
  for (int c = c0; c < c1; ++c) d += a.msg[((size_t)a.child_idx[c] * a.n_units + uu) * S + r];
  sw[threadIdx.x] = d;
  __syncthreads();
  double m = sw[gb];
  for (int k = 1; k < S; ++k) m = fmax(m, sw[gb + k]);
  __syncthreads();
  bool obs = true;
  if (c0 == c1) {  // leaf
    const size_t ci = (size_t)v * a.n_units + uu;
    obs = tl_observed(a.S1, r, a.code_a[ci], a.S1 > 0 ? a.code_b[ci] : -1);
  }
  sw[threadIdx.x] = obs ? exp(d - m) : 0.0;
  __syncthreads();
  double arg = 0.0;
  if (v == a.root) {
    for (int k = 0; k < S; ++k) arg = fma(a.pi_root[k], sw[gb + k], arg);
    if (act && r == 0) a.ll[u] = log(arg < 0.0 ? 0.0 : arg) + m;
  } else {
    const double *Pr = a.P + (((size_t)a.unit_cat[uu] * a.n_nodes + v) * S + r) * S;
    for (int k = 0; k < S; ++k) arg = fma(Pr[k], sw[gb + k], arg);
    if (act) a.msg[((size_t)v * a.n_units + uu) * S + r] = log(arg < 0.0 ? 0.0 : arg) + m;
  }
}

// ------------------------------------------------------------------ S > 64
// grid = nodes of the level x unit blocks of 16 NB x row splits (XCD-swizzled so that the workgroups of one
// node run on one XCD and share P_v in its L2), TL_NW waves.  msg layout [node][row][NU].
// Lane (lo, hi) owns unit lo of each of the NB unit blocks and, per 16-row tile, rows hi + 4 r -- the D layout
// of v_mfma_f64_16x16x4, so the lane that produced a message element is the lane that stores it.
//   prologue  every workgroup forms the WHOLE W (all S rows x its 16 NB units: d = sum of the children's
//             messages, m = max, W = exp(d - m) * obs) in LDS; wave w owns the row tiles w, w + TL_NW, ...
//   product   the workgroup of row split rs multiplies the row tiles t with t % RS == rs (wave w: tiles
//             (w + TL_NW j) RS + rs, j < TL_MAXT; 16 TL_NW TL_MAXT >= 512 rows) -- RS > 1 on the small levels
//             near the root, where one workgroup per (node, unit block) would leave most of the chip idle
//             and the level lasts as long as one workgroup streams its P_v.
// What bounds the product is the latency of streaming P_v (1.28 MB at 400 states, read once per workgroup,
// from HBM by the first block of a node), not the MFMA pipe: each lane keeps TL_DEPTH 16-column chunks of its
// rows in flight (32 bytes per row tile and chunk), the ring indexed by compile-time unrolling.
constexpr int TL_NW = 8, TL_MAXT = 4, TL_DEPTH = 3;
template <int NB>
__device__ __forceinline__ int tl_w_index(int k, int nb, int lo) { return (k * NB + (k >> 2) + nb) * 16 + lo; }  // hi groups 128 B apart mod 256

// The product of one wave: G row tiles (rows row0 + j row_step + 0..15) x NB unit blocks.  Within a 16-wide k
// chunk lane hi takes k = 16 kc + 4 hi + c in step c (A and B agree, the MFMA sums its 4 lanes-of-hi), so each
// lane reads 4 consecutive doubles of its P row per chunk.
template <int NB>
struct TlProd {
  const double *Pv, *sW;
  double *mv;                 // msg + v S NU
  int S, Sp, NU, row0, row_step, lo, hi;
  int u[NB];
  bool act[NB];
  double m[NB];
  int ldp, Kc;                // row stride of Pv and its column (contraction) extent: S, S for a transition matrix; LD, LD for U -- when
                              // LD > S the padding's eigenpairs (lam = 0, zero on the real rows) sit BETWEEN the real ones in the sorted order
  // LEAF (tl_leaf_mfma_kernel): acc = sum_k U[i][k] F_k TU[J_u][k]; arg = (split ? [i in J_u] d_i + t TA[J_u][i] : 0) + acc) / d_i
  const double *dsq, *ta[NB];
  int J[NB], S1;
  double tsplit;              // t when the split form I + tA + U phi2 U^T is used, else 0
  bool split;
};

// is state `row` in the observation set J?  J < S: that state; S + a: first letter a; S + S1 + b: second letter b; else all
__device__ __forceinline__ bool tl_in_set(int S, int S1, int row, int J) {
  if (J < S) return row == J;
  if (S1 > 0 && J < S + S1) return row / S1 == J - S;
  if (S1 > 0 && J < S + 2 * S1) return row % S1 == J - S - S1;
  return true;
}

template <int NB, int G, bool VEC, bool LEAF = false>
__device__ __forceinline__ void tl_product(const TlProd<NB> &p) {
  constexpr int DEPTH = NB == 1 ? TL_DEPTH - 1 : TL_DEPTH;   // (the 16-unit form is for tiny inputs; three stages spill there)
  const int S = p.S, lo = p.lo, hi = p.hi;
  const double *prow[G];
#pragma unroll
  for (int j = 0; j < G; ++j) {
    int row = p.row0 + j * p.row_step + lo;
    row = row < S ? row : S - 1;             // rows beyond S: results discarded
    prow[j] = p.Pv + (size_t)row * p.ldp;
  }
  // chunk kc of the lane's row of tile j; columns beyond S are clamped (W = 0 there)
  auto load_a = [&](int kc, int j, d4 &av) {
    const int k0 = 16 * kc + 4 * hi;
    if (VEC) {
      av = *reinterpret_cast<const d4 *>(prow[j] + (k0 + 3 < p.Kc ? k0 : p.Kc - 4));
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) av[c] = prow[j][k0 + c < p.Kc ? k0 + c : p.Kc - 1];
    }
  };
  auto step = [&](int kc, const d4 (&ac)[G], d4 (&acc)[G][NB]) {
    double bv[NB][4];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int c = 0; c < 4; ++c) bv[nb][c] = p.sW[tl_w_index<NB>(16 * kc + 4 * hi + c, nb, lo)];
#pragma unroll
    for (int j = 0; j < G; ++j)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[j][nb] = mfma_f64(ac[j][c], bv[nb][c], acc[j][nb]);
  };
  d4 acc[G][NB];
#pragma unroll
  for (int j = 0; j < G; ++j)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[j][nb] = d4{0.0, 0.0, 0.0, 0.0};
  const int nchunk = p.Sp / 16;
  d4 ring[DEPTH][G];
#pragma unroll
  for (int st = 0; st < DEPTH; ++st)
#pragma unroll
    for (int j = 0; j < G; ++j) load_a(st < nchunk ? st : nchunk - 1, j, ring[st][j]);
  int kc0 = 0;
  for (; kc0 + DEPTH <= nchunk; kc0 += DEPTH) {
#pragma unroll
    for (int st = 0; st < DEPTH; ++st) {
      const int kc = kc0 + st;
      d4 ac[G];
#pragma unroll
      for (int j = 0; j < G; ++j) ac[j] = ring[st][j];
      // refill this slot with the chunk DEPTH ahead (clamped: the tail re-reads the last chunk)
      const int kn = kc + DEPTH < nchunk ? kc + DEPTH : nchunk - 1;
#pragma unroll
      for (int j = 0; j < G; ++j) load_a(kn, j, ring[st][j]);
      step(kc, ac, acc);
    }
  }
#pragma unroll
  for (int st = 0; st < DEPTH - 1; ++st)
    if (kc0 + st < nchunk) step(kc0 + st, ring[st], acc);
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    if (!p.act[nb]) continue;
    double *mv = p.mv + p.u[nb];
#pragma unroll
    for (int j = 0; j < G; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = p.row0 + j * p.row_step + hi + 4 * r;
        double arg = acc[j][nb][r];
        if (LEAF && row < S) {
          const double di = p.dsq[row];
          if (p.split) arg += fma(p.tsplit, p.ta[nb][row], tl_in_set(S, p.S1, row, p.J[nb]) ? di : 0.0);
          arg /= di;
        }
        if (row < S) mv[(size_t)row * p.NU] = log(arg < 0.0 ? 0.0 : arg) + p.m[nb];
      }
  }
}

template <int NB>
__global__ __launch_bounds__(TL_NW * 64, 2) void tl_mfma_kernel(TlArgs a) {
  extern __shared__ double tl_lds[];
  double *sW = tl_lds;                     // [(Sp NB + Sp / 4)][16]
  const int S = a.S, nt = (S + 15) / 16, Sp = nt * 16, RS = a.RS;
  double *sR = sW + (size_t)(Sp * NB + Sp / 4 + NB) * 16;   // [TL_NW][NB][16] cross-wave reductions
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lo = lane & 15, hi = lane >> 4;
  const int vid = xcd_swizzle(blockIdx.x, gridDim.x);
  const int node_i = vid / (a.n_blocks * RS), rem = vid - node_i * (a.n_blocks * RS);
  const int blk = rem / RS, rs = rem - blk * RS;
  int u[NB];
  bool act[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    u[nb] = (blk * NB + nb) * 16 + lo;
    act[nb] = u[nb] < a.n_units;
  }
  const int v = a.level_nodes[node_i];
  const int c0 = a.child_ptr[v], c1 = a.child_ptr[v + 1];
  const int my_tiles = wave < nt ? (nt - wave + TL_NW - 1) / TL_NW : 0;

  double m[NB];
  {
    double d[TL_MAXT][NB][4];
#pragma unroll
    for (int j = 0; j < TL_MAXT; ++j)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) d[j][nb][r] = 0.0;
    // The children's messages: every load unconditional (rows clamped, unit columns exist up to NU, which the
    // host pads to the workgroup's 16 NB units), so that a child's 16 NB loads per lane are in flight at once --
    // with a branch per element they were issued one at a time, and that chain of latencies was a large part of
    // a workgroup's duration.  The sums keep the reference's child order.
    int roff[TL_MAXT][4];
#pragma unroll
    for (int j = 0; j < TL_MAXT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = (wave + TL_NW * j) * 16 + hi + 4 * r;
        roff[j][r] = (row < S ? row : S - 1) * a.NU;
      }
    for (int c = c0; c < c1; ++c) {
      const double *m0 = a.msg + (size_t)a.child_idx[c] * S * a.NU;
      double x0[TL_MAXT][NB][4];
#pragma unroll
      for (int j = 0; j < TL_MAXT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) x0[j][nb][r] = m0[roff[j][r] + u[nb]];
#pragma unroll
      for (int j = 0; j < TL_MAXT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) d[j][nb][r] += x0[j][nb][r];
    }
    // m = max over the unit's S rows: registers -> the 4 hi lanes -> the waves
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      double mm = -INFINITY;
#pragma unroll
      for (int j = 0; j < TL_MAXT; ++j)
        if (j < my_tiles)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if ((wave + TL_NW * j) * 16 + hi + 4 * r < S) mm = fmax(mm, d[j][nb][r]);
      mm = fmax(mm, __shfl_xor(mm, 16));
      mm = fmax(mm, __shfl_xor(mm, 32));
      if (hi == 0) sR[(wave * NB + nb) * 16 + lo] = mm;
    }
    __syncthreads();
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      double mm = sR[nb * 16 + lo];
#pragma unroll
      for (int w = 1; w < TL_NW; ++w) mm = fmax(mm, sR[(w * NB + nb) * 16 + lo]);
      m[nb] = mm;
    }
    double root_part[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      int ca = -1, cb = -1;
      if (c0 == c1 && act[nb]) {
        const size_t ci = (size_t)v * a.n_units + u[nb];
        ca = a.code_a[ci];
        cb = a.S1 > 0 ? a.code_b[ci] : -1;
      }
      root_part[nb] = 0.0;
#pragma unroll
      for (int j = 0; j < TL_MAXT; ++j)
        if (j < my_tiles)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = (wave + TL_NW * j) * 16 + hi + 4 * r;
            double w = 0.0;
            if (row < S && act[nb] && tl_observed(a.S1, row, ca, cb)) w = exp(d[j][nb][r] - m[nb]);
            sW[tl_w_index<NB>(row, nb, lo)] = w;
            if (v == a.root && row < S) root_part[nb] = fma(a.pi_root[row], w, root_part[nb]);
          }
    }
    __syncthreads();   // W complete; sR reads done
    if (v == a.root) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        double rp = root_part[nb];
        rp += __shfl_xor(rp, 16);
        rp += __shfl_xor(rp, 32);
        if (hi == 0) sR[(wave * NB + nb) * 16 + lo] = rp;
      }
      __syncthreads();
      if (wave == 0 && hi == 0 && rs == 0)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          if (act[nb]) {
            double arg = sR[nb * 16 + lo];
#pragma unroll
            for (int w = 1; w < TL_NW; ++w) arg += sR[(w * NB + nb) * 16 + lo];
            a.ll[u[nb]] = log(arg < 0.0 ? 0.0 : arg) + m[nb];
          }
      return;
    }
  }
  // arg[row][unit] = sum_k P_v[row][k] W[k][unit] (tl_product)
  // (slot map: the bank then holds the internal nodes only, one category)
  const double *Pv = a.slot ? a.P + (size_t)a.slot[v] * S * S : a.P + ((size_t)a.unit_cat[act[0] ? u[0] : 0] * a.n_nodes + v) * S * S;
  // all units of a block share the category (host contract: one category when S > 64)
  int g_tiles = 0;            // this wave's product tiles: (wave + TL_NW j) RS + rs, j < g_tiles
#pragma unroll
  for (int j = 0; j < TL_MAXT; ++j)
    if ((wave + TL_NW * j) * RS + rs < nt) g_tiles = j + 1;
  TlProd<NB> pr{Pv, sW, a.msg + (size_t)v * S * a.NU, S, Sp, a.NU, (wave * RS + rs) * 16, TL_NW * RS * 16, lo, hi, {}, {}, {}, S, S, nullptr,
                {}, {}, 0, 0.0, false};
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    pr.u[nb] = u[nb];
    pr.act[nb] = act[nb];
    pr.m[nb] = m[nb];
  }
  const bool vec = (S & 3) == 0;
  // (the tile count and the vector-load form are wave-uniform: one branch here, none inside the chunk loop, so
  // that the compiler counts the outstanding loads exactly instead of draining them at every join)
  switch (g_tiles * 2 + (vec ? 1 : 0)) {
    case 9: tl_product<NB, 4, true>(pr); break;
    case 8: tl_product<NB, 4, false>(pr); break;
    case 7: tl_product<NB, 3, true>(pr); break;
    case 6: tl_product<NB, 3, false>(pr); break;
    case 5: tl_product<NB, 2, true>(pr); break;
    case 4: tl_product<NB, 2, false>(pr); break;
    case 3: tl_product<NB, 1, true>(pr); break;
    case 2: tl_product<NB, 1, false>(pr); break;
    default: break;
  }
}

// LEAVES at S > 64 (half of a tree's nodes): a leaf has no children, so d = 0, m = 0 and W is the 0/1
// observation vector -- P_v W is a column of P_v (both sites observed: the product with the one-hot W is that
// entry EXACTLY, whatever the summation order), or a sum of S1 entries (one site a gap), or a row sum (both).
// Gathered instead of multiplied.  One workgroup per leaf: TL_LR rows of P_v at a time are staged in LDS by
// coalesced loads (P_v is streamed once), the row sums taken by the waves, then thread = (row, unit) gathers
// 1 .. S1 entries from LDS; units are the fastest index, so the message stores are contiguous.
constexpr int TL_LR = 16, TL_LT = 512;
__global__ __launch_bounds__(TL_LT) void tl_leaf_kernel(TlArgs a) {
  extern __shared__ double tl_lds[];
  const int S = a.S, S1 = a.S1, SR = S + 1;
  double *sP = tl_lds;              // [TL_LR][S + 1]
  double *sT = sP + TL_LR * SR;     // [TL_LR] row sums
  double *sM = sT + TL_LR;          // [TL_LR][2 S1] marginals: sum over the second site's state, over the first site's
  signed char *sCa = reinterpret_cast<signed char *>(sM + TL_LR * 2 * S1), *sCb = sCa + a.n_units;   // the leaf's state codes
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int v = a.level_nodes[blockIdx.x];
  const double *Pv = a.P + ((size_t)a.unit_cat[0] * a.n_nodes + v) * S * S;   // one category when S > 64
  const signed char *ca_v = a.code_a + (size_t)v * a.n_units;
  const signed char *cb_v = S1 > 0 ? a.code_b + (size_t)v * a.n_units : nullptr;
  double *mv = a.msg + (size_t)v * S * a.NU;
  for (int u = threadIdx.x; u < a.n_units; u += TL_LT) {
    sCa[u] = ca_v[u];
    sCb[u] = cb_v ? cb_v[u] : -1;
  }
  // a batch = TL_LR rows = nr S contiguous doubles; a thread's share is loaded into registers one batch AHEAD
  // (all loads independent and in flight while the previous batch is gathered), then written to LDS
  constexpr int PER = (TL_LR * 512 + TL_LT - 1) / TL_LT;   // S <= 512
  double x[PER];
  auto fetch = [&](int r0) {
    const double *src = Pv + (size_t)r0 * S;
    const int n = min(TL_LR, S - r0) * S;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int e = threadIdx.x + TL_LT * q;
      x[q] = src[e < n ? e : n - 1];
    }
  };
  // (row, column) of the thread's first element and the step of TL_LT elements, without a division per element
  const int i_first = threadIdx.x / S, k_first = threadIdx.x - i_first * S, i_step = TL_LT / S, k_step = TL_LT - i_step * S;
  const int iu_first = threadIdx.x / a.n_units, u_first = threadIdx.x - iu_first * a.n_units;
  const int iu_step = TL_LT / a.n_units, u_step = TL_LT - iu_step * a.n_units;
  fetch(0);
  for (int r0 = 0; r0 < S; r0 += TL_LR) {
    const int nr = min(TL_LR, S - r0);
    {
      int i = i_first, k = k_first;
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        if (i < nr) sP[i * SR + k] = x[q];
        i += i_step;
        k += k_step;
        if (k >= S) {
          k -= S;
          ++i;
        }
      }
    }
    __syncthreads();
    if (r0 + TL_LR < S) fetch(r0 + TL_LR);
    for (int i = wave; i < nr; i += TL_LT / 64) {
      double t = 0.0;
      for (int k = lane; k < S; k += 64) t += sP[i * SR + k];
      t = wave_sum(t);
      if (lane == 0) sT[i] = t;
    }
    // one site a gap: the sum over that site's S1 states, once per (row, known state) instead of once per unit;
    // four independent partial sums, so that the LDS reads are in flight together
    for (int t = threadIdx.x; t < nr * 2 * S1; t += TL_LT) {
      const int i = t / (2 * S1), w = t - i * 2 * S1;
      const double *p = sP + i * SR + (w < S1 ? w * S1 : w - S1);
      const int st = w < S1 ? 1 : S1;
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
      int b = 0;
      for (; b + 4 <= S1; b += 4) {
        s0 += p[b * st];
        s1 += p[(b + 1) * st];
        s2 += p[(b + 2) * st];
        s3 += p[(b + 3) * st];
      }
      for (; b < S1; ++b) s0 += p[b * st];
      sM[t] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    for (int i = iu_first, u = u_first; i < nr;) {
      const int ca = sCa[u], cb = sCb[u];
      // one LDS read whatever the observation: the entry, a marginal, or the row sum
      const double *src = S1 > 0 ? (ca >= 0 && cb >= 0 ? sP + i * SR + ca * S1 + cb
                                    : ca >= 0          ? sM + i * 2 * S1 + ca
                                    : cb >= 0          ? sM + i * 2 * S1 + S1 + cb
                                                       : sT + i)
                                 : (ca >= 0 ? sP + i * SR + ca : sT + i);
      const double arg = *src;
      mv[(size_t)(r0 + i) * a.NU + u] = log(arg < 0.0 ? 0.0 : arg);
      i += iu_step;
      u += u_step;
      if (u >= a.n_units) {
        u -= a.n_units;
        ++i;
      }
    }
    __syncthreads();
  }
}


// ---- leaves from the model's eigendecomposition (round 6) -------------------------------------------------------------------------
// A leaf's message is log of a column of P_v (both sites observed), of a sum of S1 columns (one site a gap) or of a row sum
// (both): sum_{j in J} P_v[i][j] for an observation set J that depends on the unit.  Round 5 formed ALL of P_v for every leaf --
// half of the bank's 262 GFLOP and 1.3 GB of its traffic at 1024 leaves -- to gather 128 of its 400 columns.  With
// P_v = D^-1/2 (I + t A + U phi2(t Lam) U^T) D^1/2 (t rho <= 1; else U e^{t Lam} U^T):
//   sum_{j in J} P_v[i][j] = ( [i in J] d_i + t TA[J][i] + sum_k U[i][k] F_v[k] TU[J][k] ) / d_i,
//   TU[J][k] = sum_{j in J} U[j][k] d_j,   TA[J][i] = sum_{j in J} A[i][j] d_j      (once per model: tl_tables_kernel)
// i.e. ONE [S x S] x [S x units] product per leaf with U as the streamed operand (1.28 MB shared by all leaves: L2) and the other
// formed in LDS from the unit's table row -- 41 instead of 77 MFLOP per leaf, no P_v.  The split keeps the relative accuracy of
// the O(t^2) entries exactly as the bank does (same rule, t 2 sigma <= 1).
__global__ __launch_bounds__(256) void tl_tables_kernel(int S, int S1, int LD, int nJ, const double *__restrict__ U,
                                                        const double *__restrict__ A, const double *__restrict__ dsq,
                                                        double *__restrict__ TU, double *__restrict__ TA) {
  const int J = blockIdx.x;
  for (int k = threadIdx.x; k < LD; k += 256) {
    // TU over all LD eigenpairs k (the padding's are zero on the real rows j), TA over the real states i = k < S
    double su = 0.0, sa = 0.0;
    if (J < S) {
      su = U[(size_t)J * LD + k] * dsq[J];
      sa = k < S ? A[(size_t)k * LD + J] * dsq[J] : 0.0;
    } else {
      // a marginal: the states of the set in ascending order (fixed summation order)
      for (int j = 0; j < S; ++j)
        if (tl_in_set(S, S1, j, J)) {
          su = fma(U[(size_t)j * LD + k], dsq[j], su);
          if (k < S) sa = fma(A[(size_t)k * LD + j], dsq[j], sa);
        }
    }
    TU[(size_t)J * LD + k] = su;
    TA[(size_t)J * LD + k] = sa;
  }
  (void)nJ;
}

// grid = leaves of the family x unit blocks of 16 NB, TL_NW waves; msg layout [node][row][NU] as tl_mfma_kernel
template <int NB>
__global__ __launch_bounds__(TL_NW * 64, 2) void tl_leaf_mfma_kernel(TlArgs a) {
  extern __shared__ double tl_lds[];
  double *sW = tl_lds;                     // [(Sp NB + Sp / 4)][16]
  const int S = a.S, nt = (S + 15) / 16, Sp = nt * 16, LD = a.LD;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lo = lane & 15, hi = lane >> 4;
  const int vid = xcd_swizzle(blockIdx.x, gridDim.x);
  const int node_i = vid / a.n_blocks, blk = vid - node_i * a.n_blocks;
  const int v = a.level_nodes[node_i];
  const double t = a.tnode[v];
  const bool split = t * 2.0 * (*a.sigma) <= 1.0;   // the bank's rule (k1_tile)
  int u[NB], J[NB];
  bool act[NB];
  const int nJ_all = S + (a.S1 > 0 ? 2 * a.S1 : 0);
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    u[nb] = (blk * NB + nb) * 16 + lo;
    act[nb] = u[nb] < a.n_units;
    int ca = -1, cb = -1;
    if (act[nb]) {
      const size_t ci = (size_t)v * a.n_units + u[nb];
      ca = a.code_a[ci];
      cb = a.S1 > 0 ? a.code_b[ci] : -1;
    }
    J[nb] = a.S1 > 0 ? (ca >= 0 && cb >= 0 ? ca * a.S1 + cb : ca >= 0 ? S + ca : cb >= 0 ? S + a.S1 + cb : nJ_all)
                     : (ca >= 0 ? ca : nJ_all);
  }
  // W[k][unit] = F_v[k] TU[J_unit][k]: wave w fills the k tiles w, w + TL_NW, ...
  for (int j = 0; j < TL_MAXT; ++j) {
    const int kt = wave + TL_NW * j;
    if (kt >= nt) break;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = kt * 16 + hi + 4 * r;
      // (k runs over all LD eigenpairs: Sp == LD)
      const double x = t * a.lam[k];
      const double f = split ? phi2(x) : exp(x);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) sW[tl_w_index<NB>(k, nb, lo)] = act[nb] ? f * a.TU[(size_t)J[nb] * LD + k] : 0.0;
    }
  }
  __syncthreads();
  int g_tiles = 0;
#pragma unroll
  for (int j = 0; j < TL_MAXT; ++j)
    if (wave + TL_NW * j < nt) g_tiles = j + 1;
  TlProd<NB> pr{a.U, sW, a.msg + (size_t)v * S * a.NU, S, Sp, a.NU, wave * 16, TL_NW * 16, lo, hi, {}, {}, {}, LD, LD, a.dsq, {}, {}, a.S1,
                split ? t : 0.0, split};
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    pr.u[nb] = u[nb];
    pr.act[nb] = act[nb];
    pr.m[nb] = 0.0;
    pr.J[nb] = J[nb];
    pr.ta[nb] = a.TA + (size_t)J[nb] * LD;
  }
  const bool vec = true;   // (U's rows are LD = 16 n doubles apart and LD long)
  switch (g_tiles * 2 + (vec ? 1 : 0)) {
    case 9: tl_product<NB, 4, true, true>(pr); break;
    case 8: tl_product<NB, 4, false, true>(pr); break;
    case 7: tl_product<NB, 3, true, true>(pr); break;
    case 6: tl_product<NB, 3, false, true>(pr); break;
    case 5: tl_product<NB, 2, true, true>(pr); break;
    case 4: tl_product<NB, 2, false, true>(pr); break;
    case 3: tl_product<NB, 1, true, true>(pr); break;
    case 2: tl_product<NB, 1, false, true>(pr); break;
    default: break;
  }
}
