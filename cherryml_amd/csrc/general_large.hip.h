// General (non-reversible Q) path for S > 32 (one bank, L = 1): the algorithm of general_small.hip.h --
// scaling and squaring with a degree-18 Taylor polynomial (Horner) forward, its exact reverse-mode adjoint
// backward; the algorithm class of torch.matrix_exp, which is what the reference runs (trainer.py:170-172,
// 186) -- as BATCHED 80 x 80-tile GEMMs over the buckets on the float64 MFMA (large_bank.hip.h's tile).
//
//   X_b = a_b Q,  a_b = t_b / 2^{s_b},  s_b = max(0, ceil(log2 (t_b |Q|_1)))        (|X_b|_1 <= 1: 1/19! < 1e-17)
//   R_18 = I + X/18,  R_k = I + X R_{k+1} / k  (k = 17..1),  E_0 = R_1,  E_i = E_{i-1}^2 (i <= s_b),  P_b = E_{s_b}
//   loss -= <C_b, log P_b>;  Pbar = -C_b / P_b / n
//   Ebar_{i-1} = Ebar_i E_{i-1}^T + E_{i-1}^T Ebar_i
//   Hbar_1 = Ebar_0;  Xbar += Hbar_k R_{k+1}^T / k;  Hbar_{k+1} = X^T Hbar_k / k;  Xbar += Hbar_18 / 18
//   dL/dQ = sum_b a_b Xbar_b
//
// Every product has the form C[m][n] = sum_k Aop[k][m] Bop[k][n] (both operands k-major), so every matrix that
// is later contracted over its column index is stored twice, plain and transposed (the epilogue writes both):
// ~2 (17 + s_max + 3) matrices per bucket, 10 GB at 400 states and 129 buckets -- HBM is 288 GB.  About 50 + 3 s
// products per bucket against 3 for the spectral path: it exists for completeness (every 400-state mask the
// reference ships is symmetric) and parity, like its small-state twin.
#pragma once
#include "large_bank.hip.h"

#define GL_DEG 18

struct BgArgs {
  int LD, B;
  const double *A1, *B1;       // first operand pair, k-major [LD][LD] per bucket
  size_t sA1, sB1;             // bucket strides in doubles (0: one matrix shared by all buckets)
  const double *A2, *B2;       // optional second pair added to the same tile (null: none)
  size_t sA2, sB2;
  double *C, *CT;              // out [B][LD][LD] and (optional) its transpose, bucket stride sC
  size_t sC;
  const double *alpha;         // [B] per-bucket factor (null: 1)
  double scale;                // common factor
  double add_identity;         // + this on the diagonal
  int accumulate;              // C += result instead of C = result
  const int *nsq;              // [B] squarings of the bucket; the bucket takes part iff round <= nsq[b] (null: all do)
  int round;
};

__global__ __launch_bounds__(LG4_THREADS, 4) void bg_gemm(BgArgs a) {
  __shared__ double sAB[4 * LG_KT * LG_TM];
  double *sA = sAB, *sB = sAB + 2 * LG_KT * LG_TM;
  const int tilesN = (a.LD + LG_TN - 1) / LG_TN, tiles = tilesN * tilesN;
  const int vid = xcd_swizzle(blockIdx.x, gridDim.x);
  const int b = vid / tiles, tile = vid - b * tiles;
  if (a.nsq && a.round > a.nsq[b]) return;          // this bucket has no such squaring round (block-uniform)
  const int tm = tile / tilesN, tn = tile - tm * tilesN;
  const int m0 = tm * LG_TM, n0 = tn * LG_TN;
  d4 acc[5], ax0, ax1;
  GemmOperands<double> g1{a.A1 + (size_t)b * a.sA1, a.B1 + (size_t)b * a.sB1, a.LD, a.LD, a.LD, a.LD, a.LD, nullptr};
  lg4_gemm_tile<double, false, true>(g1, m0, n0, sA, sB, acc, ax0, ax1);
  if (a.A2) {
    GemmOperands<double> g2{a.A2 + (size_t)b * a.sA2, a.B2 + (size_t)b * a.sB2, a.LD, a.LD, a.LD, a.LD, a.LD, nullptr};
    lg4_gemm_tile<double, false, false>(g2, m0, n0, sA, sB, acc, ax0, ax1);
  }
  const double f = a.scale * (a.alpha ? a.alpha[b] : 1.0);
  double *__restrict__ C = a.C + (size_t)b * a.sC;
  double *__restrict__ CT = a.CT ? a.CT + (size_t)b * a.sC : nullptr;
  lg_for_each<double>(m0, n0, acc, ax0, ax1, [&](int row, int col, double v) {
    if (row < a.LD && col < a.LD) {
      double o = f * v + (row == col ? a.add_identity : 0.0);
      const size_t i = (size_t)row * a.LD + col;
      if (a.accumulate) o += C[i];
      C[i] = o;
      if (CT) CT[(size_t)col * a.LD + row] = o;
    }
  });
}

// QT, Qn = padded transposed / plain copies of Q [S][S]; colsum[j] = sum_i |Q_ij| (|Q|_1 = max over j)
__global__ void gl_prep(int S, int LD, const double *Q, double *Qn, double *QT, double *colsum) {
  const int j = blockIdx.x;     // column of Q
  __shared__ double s[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < LD; i += 256) {
    const double v = (i < S && j < S) ? Q[(size_t)i * S + j] : 0.0;
    Qn[(size_t)i * LD + j] = v;
    QT[(size_t)j * LD + i] = v;
    acc += fabs(v);
  }
  s[threadIdx.x] = acc;
  __syncthreads();
  for (int st = 128; st >= 1; st >>= 1) {
    if ((int)threadIdx.x < st) s[threadIdx.x] += s[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) colsum[j] = s[0];
}

// R_18 = I + a_b Q / 18  (plain and transposed), all buckets
__global__ void gl_first(int LD, int B, const double *Qn, const double *alpha, double *R, double *RT) {
  const size_t LL = (size_t)LD * LD, e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= LL * B) return;
  const int b = (int)(e / LL);
  const size_t i = e - (size_t)b * LL;
  const int row = (int)(i / LD), col = (int)(i - (size_t)row * LD);
  const double f = alpha[b] / GL_DEG;
  R[e] = f * Qn[i] + (row == col ? 1.0 : 0.0);
  RT[e] = f * Qn[(size_t)col * LD + row] + (row == col ? 1.0 : 0.0);
}

// From the final E = P_b (plain and transposed; slot nsq[b] of the squaring stack): expm output, or loss
// partial + Pbar (plain into G, transposed into GT; both in the ping-pong half selected by the bucket's parity).
// One 32 x 32 tile per workgroup; the transposed copy goes through LDS.
struct GlLoss {
  int S, LD, B;
  const double *E, *ET;    // squaring stacks [slot][B][LD][LD]
  const int *nsq;
  int slot_and;            // ~0: bucket b's result sits in slot nsq[b]; 1: two ping-pong slots (counts-free handles)
  const double *Ct;        // [B][LD][LD] transposed counts (padded)
  double inv_n;
  double *G, *GT;          // Pbar buffers (two halves each)
  size_t half;
  double *loss_part;       // [B * tiles32]
  double *P;               // expm mode: [B][S][S] out (then nothing else is written)
};
__global__ void gl_loss(GlLoss a) {
  __shared__ double tile[32][33];
  __shared__ double red[32];
  const int b = blockIdx.z, LD = a.LD;
  const size_t LL = (size_t)LD * LD, slot = ((size_t)(a.nsq[b] & a.slot_and) * a.B + b) * LL;
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  if (a.P) {
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
      const int i = i0 + r, j = j0 + threadIdx.x;
      if (i < a.S && j < a.S) a.P[((size_t)b * a.S + i) * a.S + j] = a.E[slot + (size_t)i * LD + j];
    }
    return;
  }
  // work on the TRANSPOSED copies (Ct is stored transposed): GT[j][i] = -Ct[j][i] / ET[j][i] / n.
  // Pbar starts in half (nsq[b] & 1) of the ping-pong buffers: backward squaring round i reads half i & 1 and
  // writes half (i - 1) & 1 for every bucket that takes part in it (i <= nsq[b]), so after round 1 ALL buckets --
  // those without squarings too -- hold Ebar_0 in half 0.
  const size_t par = (size_t)(a.nsq[b] & 1) * a.half;
  double acc = 0.0;
  for (int r = threadIdx.y; r < 32; r += blockDim.y) {
    const int j = j0 + r, i = i0 + threadIdx.x;       // element (j, i) of the transposed matrices
    double gbar = 0.0;
    if (j < LD && i < LD) {
      const double c = a.Ct[(size_t)b * LL + (size_t)j * LD + i];
      const double p = a.ET[slot + (size_t)j * LD + i];
      if (c != 0.0) {
        acc = fma(-c, log(p), acc);
        gbar = -c * a.inv_n / p;
      }
      a.GT[par + (size_t)b * LL + (size_t)j * LD + i] = gbar;
    }
    tile[r][threadIdx.x] = gbar;
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += blockDim.y) {
    const int i = i0 + r, j = j0 + threadIdx.x;       // element (i, j) of the plain matrix = tile[j - j0][i - i0]
    if (i < LD && j < LD) a.G[par + (size_t)b * LL + (size_t)i * LD + j] = tile[threadIdx.x][r];
  }
  // loss partial of this tile
  acc = wave_sum(acc);
  const int w = (threadIdx.y * blockDim.x + threadIdx.x) >> 6, nw = (blockDim.x * blockDim.y) >> 6;
  if (((threadIdx.y * blockDim.x + threadIdx.x) & 63) == 0) red[w] = acc;
  __syncthreads();
  if (threadIdx.x == 0 && threadIdx.y == 0) {
    double t = 0.0;
    for (int k = 0; k < nw; ++k) t += red[k];
    a.loss_part[((size_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = t;
  }
}

__global__ void gl_finish_loss(const double *part, int nparts, double inv_n, double *loss) {
  __shared__ double s[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) acc += part[i];
  s[threadIdx.x] = acc;
  __syncthreads();
  for (int st = 128; st >= 1; st >>= 1) {
    if ((int)threadIdx.x < st) s[threadIdx.x] += s[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) *loss = s[0] * inv_n;
}

// Xbar_b += a_b Hbar_18 / 18 (the last Horner term: R_19 = I)
__global__ void gl_last(int LD, int B, const double *H18, const double *alpha, double *Xbar) {
  const size_t LL = (size_t)LD * LD, e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= LL * B) return;
  const int b = (int)(e / LL);
  Xbar[e] += alpha[b] / GL_DEG * H18[e];
}

// dQ[S][S] = sum_b Xbar_b (padded) -- fixed order
__global__ void gl_reduce(int S, int LD, int B, const double *Xbar, double *dQ) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= S * S) return;
  const int i = e / S, j = e - i * S;
  const size_t LL = (size_t)LD * LD;
  double s0 = 0.0, s1 = 0.0;
  int b = 0;
  for (; b + 1 < B; b += 2) {
    s0 += Xbar[(size_t)b * LL + (size_t)i * LD + j];
    s1 += Xbar[(size_t)(b + 1) * LL + (size_t)i * LD + j];
  }
  if (b < B) s0 += Xbar[(size_t)b * LL + (size_t)i * LD + j];
  dQ[e] = s0 + s1;
}
