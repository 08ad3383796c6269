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
//   tl_mfma_kernel   S  > 64 (the 400-state pair model): 16 units per workgroup, arg = P_v W as
//                    v_mfma_f64_16x16x4 tiles (rows x units), W staged in LDS, P streamed from HBM
#pragma once
#include "common.hip.h"
#include "common.hip.h"   // xcd_swizzle, mfma_f64

struct TlArgs {
  int S, S1;                   // states; S1 > 0: pair model over an S1-letter alphabet
  int n_nodes, n_units, NU;    // NU: units padded to the message layout's unit stride
  int root, n_level, n_blocks;  // this launch: nodes of one height x unit blocks (1-D grid)
  const int *level_nodes;      // nodes of the height processed by this launch
  const int *child_ptr, *child_idx;   // CSR children, in the reference's child order
  const double *P;             // [cat][node][S][S] transition matrices of the edge above `node`
  const int *unit_cat;         // [n_units] rate category of each unit
  const signed char *code_a, *code_b;  // [node][unit] observed state (-1: unobserved); leaves only
  const double *pi_root;       // [S]
  double *msg;                 // upward messages (layout per kernel)
  double *ll;                  // [n_units]
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
// grid = nodes of the level x unit blocks of 16 (XCD-swizzled so that the blocks of one node run
// on one XCD and share P_v in its L2), TL_NW waves; wave w owns the row tiles w, w + TL_NW, ...
// (16 rows each, TL_MAXT per wave at most => S <= 512).  msg layout [node][row][NU].
// Lane (lo, hi) owns unit lo and, per tile, rows hi + 4 r -- the D layout of v_mfma_f64_16x16x4,
// so the lane that produced a message element is the lane that stores it.
constexpr int TL_NW = 8, TL_MAXT = 4;
__device__ __forceinline__ int tl_w_index(int k, int lo) { return (k + (k >> 2)) * 16 + lo; }  // bank-spread rows

__global__ __launch_bounds__(TL_NW * 64, 4) void tl_mfma_kernel(TlArgs a) {
  extern __shared__ double tl_lds[];
  double *sW = tl_lds;                     // [(Sp + Sp / 4)][16]
  const int S = a.S, nt = (S + 15) / 16, Sp = nt * 16;
  double *sR = sW + (size_t)(Sp + Sp / 4) * 16;   // [TL_NW][16] cross-wave reductions
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lo = lane & 15, hi = lane >> 4;
  const int vid = xcd_swizzle(blockIdx.x, gridDim.x);
  const int node_i = vid / a.n_blocks, blk = vid - node_i * a.n_blocks;
  const int u = blk * 16 + lo;
  const bool act = u < a.n_units;
  const int v = a.level_nodes[node_i];
  const int c0 = a.child_ptr[v], c1 = a.child_ptr[v + 1];
  const int my_tiles = wave < nt ? (nt - wave + TL_NW - 1) / TL_NW : 0;

  double d[TL_MAXT][4];
#pragma unroll
  for (int j = 0; j < TL_MAXT; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) d[j][r] = 0.0;
  for (int c = c0; c < c1; ++c) {
    const double *mc = a.msg + (size_t)a.child_idx[c] * S * a.NU + u;
#pragma unroll
    for (int j = 0; j < TL_MAXT; ++j)
      if (j < my_tiles)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = (wave + TL_NW * j) * 16 + hi + 4 * r;
          if (row < S && act) d[j][r] += mc[(size_t)row * a.NU];
        }
  }
  // m = max over the unit's S rows: registers -> the 4 hi lanes -> the 4 waves
  double m = -INFINITY;
#pragma unroll
  for (int j = 0; j < TL_MAXT; ++j)
    if (j < my_tiles)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if ((wave + TL_NW * j) * 16 + hi + 4 * r < S) m = fmax(m, d[j][r]);
  m = fmax(m, __shfl_xor(m, 16));
  m = fmax(m, __shfl_xor(m, 32));
  if (hi == 0) sR[wave * 16 + lo] = m;
  __syncthreads();
  m = sR[lo];
#pragma unroll
  for (int w = 1; w < TL_NW; ++w) m = fmax(m, sR[w * 16 + lo]);
  int ca = -1, cb = -1;
  if (c0 == c1 && act) {
    const size_t ci = (size_t)v * a.n_units + u;
    ca = a.code_a[ci];
    cb = a.S1 > 0 ? a.code_b[ci] : -1;
  }
  double root_part = 0.0;
#pragma unroll
  for (int j = 0; j < TL_MAXT; ++j)
    if (j < my_tiles)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = (wave + TL_NW * j) * 16 + hi + 4 * r;
        double w = 0.0;
        if (row < S && act && tl_observed(a.S1, row, ca, cb)) w = exp(d[j][r] - m);
        sW[tl_w_index(row, lo)] = w;
        if (v == a.root && row < S) root_part = fma(a.pi_root[row], w, root_part);
      }
  __syncthreads();   // W complete; sR reads done
  if (v == a.root) {
    root_part += __shfl_xor(root_part, 16);
    root_part += __shfl_xor(root_part, 32);
    if (hi == 0) sR[wave * 16 + lo] = root_part;
    __syncthreads();
    if (wave == 0 && hi == 0 && act) {
      double arg = sR[lo];
#pragma unroll
      for (int w = 1; w < TL_NW; ++w) arg += sR[w * 16 + lo];
      a.ll[u] = log(arg < 0.0 ? 0.0 : arg) + m;
    }
    return;
  }
  // arg[row][unit] = sum_k P_v[row][k] W[k][unit].  Within a 16-wide k chunk lane hi takes
  // k = kk + 4 hi + c in step c (A and B agree, the MFMA sums its 4 lanes-of-hi), so each lane
  // reads 4 consecutive doubles of its P row per chunk.
  const double *Pv = a.P + ((size_t)a.unit_cat[act ? u : 0] * a.n_nodes + v) * S * S;
  // all 16 units of a block share the category (host contract: one category when S > 64)
  d4 acc[TL_MAXT];
#pragma unroll
  for (int j = 0; j < TL_MAXT; ++j) acc[j] = d4{0.0, 0.0, 0.0, 0.0};
  const bool vec = (S & 3) == 0;
  auto load_a = [&](int kk, int j, double (&av)[4]) {
    int row = (wave + TL_NW * j) * 16 + lo;
    row = row < S ? row : S - 1;             // rows beyond S: results discarded
    const double *p = Pv + (size_t)row * S;
    const int k0 = kk + 4 * hi;
    if (vec && k0 + 3 < S) {
      const d4 x = *reinterpret_cast<const d4 *>(p + k0);
      av[0] = x[0]; av[1] = x[1]; av[2] = x[2]; av[3] = x[3];
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) av[c] = p[k0 + c < S ? k0 + c : S - 1];   // W = 0 there
    }
  };
  double an[TL_MAXT][4];
#pragma unroll
  for (int j = 0; j < TL_MAXT; ++j)
    if (j < my_tiles) load_a(0, j, an[j]);
  for (int kk = 0; kk < Sp; kk += 16) {
    double ac[TL_MAXT][4], bv[4];
#pragma unroll
    for (int j = 0; j < TL_MAXT; ++j)
#pragma unroll
      for (int c = 0; c < 4; ++c) ac[j][c] = an[j][c];
    if (kk + 16 < Sp) {
#pragma unroll
      for (int j = 0; j < TL_MAXT; ++j)
        if (j < my_tiles) load_a(kk + 16, j, an[j]);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) bv[c] = sW[tl_w_index(kk + 4 * hi + c, lo)];
#pragma unroll
    for (int j = 0; j < TL_MAXT; ++j)
      if (j < my_tiles)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[j] = mfma_f64(ac[j][c], bv[c], acc[j]);
  }
  if (!act) return;
  double *mv = a.msg + (size_t)v * S * a.NU + u;
#pragma unroll
  for (int j = 0; j < TL_MAXT; ++j)
    if (j < my_tiles)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = (wave + TL_NW * j) * 16 + hi + 4 * r;
        const double arg = acc[j][r];
        if (row < S) mv[(size_t)row * a.NU] = log(arg < 0.0 ? 0.0 : arg) + m;
      }
}
