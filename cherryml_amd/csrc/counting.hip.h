// Counting kernels: HBM-bound byte/integer work -- one pass over
// the encoded sequences, integer atomics into the count tensor.  One wavefront per
// counted pair, lanes stride over its sites (or contact pairs); 4 pairs per workgroup.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cherrybank.h"

// cherryml/utils.py:35-56 in IEEE double arithmetic (same comparisons, same divisions)
__device__ __forceinline__ int cnt_quantize(double bl, const double *__restrict__ grid, int B) {
  if (bl < grid[0] || bl > grid[B - 1]) return -1;
  int lo = 0, hi = B;  // first index with grid[idx] >= bl  (np.searchsorted, side="left")
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (grid[mid] < bl) lo = mid + 1;
    else hi = mid;
  }
  if (lo == 0) return 0;
  const double left = grid[lo - 1], right = grid[lo];
  const double rel_left = bl / left - 1.0, rel_right = right / bl - 1.0;
  return (rel_left < rel_right) ? lo - 1 : lo;
}

// `counts` may be CNT_REPLICAS copies of the [B][S][S] tensor: workgroup g adds into copy
// g % replicas, so that the hot bins (identical residues on both leaves, i.e. the diagonal
// of a few buckets) are hit by 1/replicas of the atomics each -- same-address float/int
// atomics serialise at the memory side (MI355X_MICROARCH.md, Global float atomics:
// contention row).  count_reduce_replicas then sums the copies (integers: exact).
#define CNT_REPLICAS 32

__global__ void count_reduce_replicas(const unsigned long long *__restrict__ rep, int replicas,
                                      size_t nbins, unsigned long long *__restrict__ counts) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nbins) return;
  unsigned long long s = 0;
  for (int r = 0; r < replicas; ++r) s += rep[(size_t)r * nbins + i];
  counts[i] += s;
}

__global__ __launch_bounds__(256) void count_transitions_kernel(
    int S, int B, const double *__restrict__ grid, const int8_t *__restrict__ seqs,
    const double *__restrict__ rates, const cb_count_pair *__restrict__ pairs, long long n_pairs,
    int symmetric, unsigned long long *__restrict__ counts_base, int replicas) {
  const long long p = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= n_pairs) return;
  unsigned long long *counts = counts_base + (size_t)(blockIdx.x % replicas) * B * S * S;
  const cb_count_pair pr = pairs[p];
  const double total = pr.len_a + pr.len_b;
  const int8_t *sa = seqs + pr.seq_a, *sb = seqs + pr.seq_b;
  const double *rt = rates + pr.aux;
  for (int k = threadIdx.x & 63; k < pr.n; k += 64) {
    const int xa = sa[k], xb = sb[k];
    if (xa < 0 || xb < 0) continue;
    const int q = cnt_quantize(total * rt[k], grid, B);
    if (q < 0) continue;
    atomicAdd(&counts[((size_t)q * S + xa) * S + xb], 1ull);
    if (symmetric) atomicAdd(&counts[((size_t)q * S + xb) * S + xa], 1ull);
  }
}

// ---- LDS-privatised variant (used when the whole [B][S][S] histogram fits LDS as packed
// 16-bit bins: LG 20 states x 129 buckets = 51,600 bins = 103 KB).  Scattered 8-byte global
// atomics run in the slow "64 lanes -> 64 rows" regime (~0.08 TB/s); LDS atomics do not.
// Each workgroup owns `chunk` consecutive pairs, chosen by the host so that no 16-bit bin
// can overflow (2 * max_sites * chunk <= 65535), histograms them in LDS, and writes its
// histogram ONCE to its own slab; count_reduce_slabs sums the slabs (integers: exact).
#define CNT_LDS_THREADS 1024

// cnt_quantize on a grid held in LDS, branch-free lower bound.  The tie rule
// (bl/left - 1 < right/bl - 1) equals (bl^2 < left*right) in exact arithmetic; the rounded
// comparison can differ from it only when the two sides agree to ~8 ulp, so the divisions
// are only executed when |bl^2 - left*right| <= 1e-12 max(bl^2, left*right) (bit-exactness
// of the reference's rule is kept; the divergent branch is practically never taken).
// Four independent values per lane are searched together so that the dependent LDS reads of
// one search overlap the other three.
__device__ __forceinline__ void cnt_quantize_lds4(const double (&bl)[4], const double *g, int B, double g0,
                                                  double gl, int (&q)[4]) {
  int lo[4] = {0, 0, 0, 0};
  int len = B;
  while (len > 1) {
    const int half = len >> 1;
    double v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = g[lo[u] + half - 1];
#pragma unroll
    for (int u = 0; u < 4; ++u) lo[u] += (v[u] < bl[u]) ? half : 0;
    len -= half;
  }
  double left[4], right[4];
  int hi[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    // lo is the last index with g[idx] < bl, or 0: first index with g[idx] >= bl is lo or lo+1
    const double glo = g[lo[u]];
    const int first = lo[u] + ((glo < bl[u]) ? 1 : 0);
    hi[u] = first < B ? first : B - 1;
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    left[u] = g[hi[u] > 0 ? hi[u] - 1 : 0];
    right[u] = g[hi[u]];
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const bool outside = (bl[u] < g0) || (bl[u] > gl) || !(bl[u] == bl[u]);
    const double lr = left[u] * right[u], bb = bl[u] * bl[u];
    const double d = bb - lr;
    bool take_left = d < 0.0;
    if (fabs(d) <= 1e-12 * fmax(bb, lr)) take_left = (bl[u] / left[u] - 1.0) < (right[u] / bl[u] - 1.0);
    const int r = (hi[u] > 0 && take_left) ? hi[u] - 1 : hi[u];
    q[u] = outside ? -1 : r;
  }
}

__global__ __launch_bounds__(CNT_LDS_THREADS) void count_transitions_lds_kernel(
    int S, int B, const double *__restrict__ grid, const int8_t *__restrict__ seqs,
    const double *__restrict__ rates, const cb_count_pair *__restrict__ pairs, long long n_pairs,
    int symmetric, int chunk, unsigned *__restrict__ slabs, int words) {
  extern __shared__ unsigned hist[];
  double *g = reinterpret_cast<double *>(hist + ((words + 1) & ~1));
  for (int i = threadIdx.x; i < words; i += CNT_LDS_THREADS) hist[i] = 0u;
  for (int i = threadIdx.x; i < B; i += CNT_LDS_THREADS) g[i] = grid[i];
  __syncthreads();
  constexpr int NWV = CNT_LDS_THREADS / 64;
  const int lane = threadIdx.x & 63;
  const long long p0 = (long long)blockIdx.x * chunk;
  const long long p1 = p0 + chunk < n_pairs ? p0 + chunk : n_pairs;
  long long p = p0 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (p >= p1) goto flush;
  {
    // software pipeline: pair records are fetched two pairs ahead, the first 256 sites of the
    // next pair one pair ahead, so the dependent chain record -> sites -> search never stalls
    const double g0 = g[0], gl = g[B - 1];
    int xa[4], xb[4], nxa[4], nxb[4];
    double r[4], nr[4];
    auto load_sites = [&](const cb_count_pair &q, int k0, int *ya, int *yb, double *yr) {
      const int8_t *sa = seqs + q.seq_a, *sb = seqs + q.seq_b;
      const double *rt = rates + q.aux;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int kc = k0 + 64 * u + lane;
        kc = kc < q.n ? kc : q.n - 1;
        kc = kc > 0 ? kc : 0;
        ya[u] = sa[kc];
        yb[u] = sb[kc];
        yr[u] = rt[kc];
      }
    };
    auto count_sites = [&](const cb_count_pair &q, int k0, const int *ya, const int *yb, const double *yr) {
      const double total = q.len_a + q.len_b;
      const double bl[4] = {total * yr[0], total * yr[1], total * yr[2], total * yr[3]};
      int bqs[4];
      cnt_quantize_lds4(bl, g, B, g0, gl, bqs);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = k0 + 64 * u + lane;
        const int bq = bqs[u];
        if (k < q.n && ya[u] >= 0 && yb[u] >= 0 && bq >= 0) {
          const unsigned b1 = (unsigned)((bq * S + ya[u]) * S + yb[u]);
          atomicAdd(&hist[b1 >> 1], 1u << (16 * (b1 & 1)));
          if (symmetric) {
            const unsigned b2 = (unsigned)((bq * S + yb[u]) * S + ya[u]);
            atomicAdd(&hist[b2 >> 1], 1u << (16 * (b2 & 1)));
          }
        }
      }
    };
    cb_count_pair cur = pairs[p];
    cb_count_pair nxt = pairs[p + NWV < p1 ? p + NWV : p];
    load_sites(cur, 0, xa, xb, r);
    for (; p < p1; p += NWV) {
      const cb_count_pair nn = pairs[p + 2 * NWV < p1 ? p + 2 * NWV : p];
      const bool has_next = p + NWV < p1;
      if (has_next && nxt.n > 0) load_sites(nxt, 0, nxa, nxb, nr);
      if (cur.n > 0) count_sites(cur, 0, xa, xb, r);
      for (int k0 = 256; k0 < cur.n; k0 += 256) {  // long alignments: remaining chunks, unpipelined
        load_sites(cur, k0, xa, xb, r);
        count_sites(cur, k0, xa, xb, r);
      }
      cur = nxt;
      nxt = nn;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        xa[u] = nxa[u];
        xb[u] = nxb[u];
        r[u] = nr[u];
      }
    }
  }
flush:
  __syncthreads();
  unsigned *dst = slabs + (size_t)blockIdx.x * words;
  for (int i = threadIdx.x; i < words; i += CNT_LDS_THREADS) dst[i] = hist[i];
}

// 256 threads = 64 consecutive words x 4 slab lanes; every lane keeps 8 loads in flight.
__global__ __launch_bounds__(256) void count_reduce_slabs(const unsigned *__restrict__ slabs, int n_slabs,
                                                          int words, size_t nbins,
                                                          unsigned long long *__restrict__ counts) {
  __shared__ unsigned long long part[2][4][64];
  const int wl = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int w = blockIdx.x * 64 + wl;
  const int wc = w < words ? w : words - 1;
  unsigned long long lo = 0, hi = 0;
  int s = sl;
  for (; s + 28 < n_slabs; s += 32) {
    unsigned v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = slabs[(size_t)(s + 4 * u) * words + wc];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      lo += v[u] & 0xFFFFu;
      hi += v[u] >> 16;
    }
  }
  for (; s < n_slabs; s += 4) {
    const unsigned v = slabs[(size_t)s * words + wc];
    lo += v & 0xFFFFu;
    hi += v >> 16;
  }
  part[0][sl][wl] = lo;
  part[1][sl][wl] = hi;
  __syncthreads();
  if (threadIdx.x < 128) {
    const int half = threadIdx.x >> 6;
    const size_t bin = 2 * (size_t)w + half;
    if (w < words && bin < nbins)
      counts[bin] += part[half][0][wl] + part[half][1][wl] + part[half][2][wl] + part[half][3][wl];
  }
}

__global__ __launch_bounds__(256) void count_co_transitions_kernel(
    int S, int B, const double *__restrict__ grid, const int8_t *__restrict__ seqs,
    const int32_t *__restrict__ contacts, const cb_count_pair *__restrict__ pairs,
    long long n_pairs, int symmetric, unsigned long long *__restrict__ counts) {
  const long long p = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= n_pairs) return;
  const cb_count_pair pr = pairs[p];
  const int q = cnt_quantize(pr.len_a + pr.len_b, grid, B);
  if (q < 0) return;
  const int8_t *sa = seqs + pr.seq_a, *sb = seqs + pr.seq_b;
  const int32_t *ct = contacts + 2 * pr.aux;
  const size_t S2 = (size_t)S * S;
  unsigned long long *cq = counts + (size_t)q * S2 * S2;
  for (int c = threadIdx.x & 63; c < pr.n; c += 64) {
    const int i = ct[2 * c], j = ct[2 * c + 1];
    const int ai = sa[i], aj = sa[j], bi = sb[i], bj = sb[j];
    if (ai < 0 || aj < 0 || bi < 0 || bj < 0) continue;
    const size_t s1 = (size_t)ai * S + aj, s1r = (size_t)aj * S + ai;
    const size_t s2 = (size_t)bi * S + bj, s2r = (size_t)bj * S + bi;
    atomicAdd(&cq[s1 * S2 + s2], 1ull);
    atomicAdd(&cq[s1r * S2 + s2r], 1ull);
    if (symmetric) {
      atomicAdd(&cq[s2 * S2 + s1], 1ull);
      atomicAdd(&cq[s2r * S2 + s1r], 1ull);
    }
  }
}

// ---- SiteRM count / pseudocount assembly (reference _siterm/_site_specific_rate_matrix.py) ------
// raw[l][b][x][y] += 1 for every transition whose total length quantises to bucket b and whose
// two sequences carry states (x, y) at site l (:226-256).  One wavefront per transition, lanes
// over the sites; the increments are small integers, so double atomics are exact and order
// independent.  live[l * B + b] marks the (site, bucket) matrices that received anything.
// A transition spans pr.n sites and its family's first site is row pr.aux of the site axis (many
// families in one call, cb_siterm_assemble_batch; one family: aux = 0, n = n_sites).
__global__ __launch_bounds__(256) void siterm_raw_counts_kernel(
    int S, int B, const double *__restrict__ grid, const int8_t *__restrict__ seqs,
    const cb_count_pair *__restrict__ pairs, long long n_pairs, double *__restrict__ raw,
    int *__restrict__ live) {
  const long long p = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= n_pairs) return;
  const cb_count_pair pr = pairs[p];
  const int b = cnt_quantize(pr.len_a + pr.len_b, grid, B);
  if (b < 0) return;  // outside the grid: dropped (:240-247)
  const int8_t *sa = seqs + pr.seq_a, *sb = seqs + pr.seq_b;
  for (int k = threadIdx.x & 63; k < pr.n; k += 64) {
    const int x = sa[k], y = sb[k];
    if (x < 0 || y < 0) continue;
    const size_t l = (size_t)pr.aux + k;
    atomicAdd(&raw[((l * B + b) * S + x) * S + y], 1.0);
    live[l * B + b] = 1;
  }
}

// counts[l][b] = (1 - lambda) sym(raw[l][b]) + lambda * (l1 * prior[b_adj])     (:503-567), in place;
// sym = (R + R^T) / 2 when reverse transitions are included (:257-258);  l1 = sum(raw[l][b]);
// b_adj = quantization_idx(grid[b] * rate_l), clamped to the last / first bucket outside the grid.
// Products and sums are kept un-fused (__dmul_rn / __dadd_rn) so the result is the reference's
// float64 arithmetic bit for bit.  One workgroup of 64 threads per live (site, bucket).
__global__ __launch_bounds__(64) void siterm_mix_kernel(int S, int B, const double *__restrict__ grid,
                                                        const double *__restrict__ site_rates,
                                                        const double *__restrict__ prior, double lambda,
                                                        int include_reverse, const int *__restrict__ live,
                                                        double *__restrict__ counts) {
  const int l = blockIdx.x / B, b = blockIdx.x - l * B;
  if (!live[blockIdx.x]) return;
  double *M = counts + (size_t)blockIdx.x * S * S;
  const int lane = threadIdx.x;
  // all entries are multiples of 1/2 and small: the sum is exact in any order
  double part = 0.0;
  for (int e = lane; e < S * S; e += 64) part += M[e];
  const double l1 = wave_sum(part);
  if (!(l1 > 0.0)) return;
  const double tt = grid[b] * site_rates[l];
  int ba = cnt_quantize(tt, grid, B);
  if (ba < 0) ba = tt > grid[B - 1] ? B - 1 : 0;
  const double *P = prior + (size_t)ba * S * S;
  const double one_m = 1.0 - lambda;
  for (int e = lane; e < S * S; e += 64) {
    const int x = e / S, y = e - x * S;
    if (x > y) continue;  // the thread of (x, y), x <= y, also writes (y, x)
    double rxy = M[x * S + y], ryx = M[y * S + x];
    if (include_reverse) {
      const double s = __dmul_rn(__dadd_rn(rxy, ryx), 0.5);  // (R + R^T) / 2, symmetric sum is commutative
      rxy = s;
      ryx = s;
    }
    M[x * S + y] = __dadd_rn(__dmul_rn(rxy, one_m), __dmul_rn(__dmul_rn(l1, P[x * S + y]), lambda));
    if (x != y) M[y * S + x] = __dadd_rn(__dmul_rn(ryx, one_m), __dmul_rn(__dmul_rn(l1, P[y * S + x]), lambda));
  }
}
