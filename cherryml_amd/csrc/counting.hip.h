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

// ---- co-transitions, binned by bucket and privatised in LDS (reference _count_co_transitions.cpp:359-381) ------
// The histogram is [B][S^2][S^2] (165 MB of 8-byte bins at 20 letters, 129 buckets): far beyond LDS, and scattered
// 8-byte global atomics run in the guide's "64 lanes -> 64 rows" regime (0.08 TB/s).  The quantised length is a property
// of the PAIR, so the pass is organised by bucket:
//   co_bucket_kernel   q(pair) and the number of (pair, contact) events per bucket
//   co_plan_kernel     bucket offsets into ONE flat event array and the work list: (bucket, row block, event range)
//   co_expand_kernel   every event as 4 state codes in one 32-bit word (a_i, a_j, b_i, b_j; 0xFFFFFFFF = a gap), written
//                      to its bucket's range -- the only kernel that chases pair -> contacts -> sequence bytes
//   co_count_lds_kernel one workgroup per work item: R rows x S^2 columns of 32-bit bins in LDS (100 x 400 at 20 letters
//                      = 160,000 B), streams its event range (4 bytes per event, coalesced), adds the increments that
//                      fall into its rows, and adds its non-zero bins to the 8-byte global bins once (contiguous
//                      lanes -> contiguous addresses).  Integers throughout: bit-exact, order-independent.
#define CO_THREADS 1024
#define CO_LDS_WORDS 40000           // 160,000 B of the 163,840 a workgroup may declare
#define CO_EVENT_GAP 0xFFFFFFFFu

struct CoWork {
  int q, rb;               // bucket, row block (q < 0: an empty slot)
  int single, pad;         // the only chunk of its bucket: plain adds instead of atomics on the flush
  unsigned long long e0, e1;   // event range
};

__global__ __launch_bounds__(CO_THREADS) void co_bucket_kernel(int B, const double *__restrict__ grid,
                                                               const cb_count_pair *__restrict__ pairs, long long n_pairs,
                                                               int *__restrict__ qbuf,
                                                               unsigned long long *__restrict__ bucket_ev) {
  extern __shared__ unsigned long long co_ev[];   // [B]
  for (int b = threadIdx.x; b < B; b += CO_THREADS) co_ev[b] = 0ull;
  __syncthreads();
  const long long p = (long long)blockIdx.x * CO_THREADS + threadIdx.x;
  if (p < n_pairs) {
    const cb_count_pair pr = pairs[p];
    int q = cnt_quantize(pr.len_a + pr.len_b, grid, B);
    if (pr.n <= 0) q = -1;
    qbuf[p] = q;
    if (q >= 0) atomicAdd(&co_ev[q], (unsigned long long)pr.n);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < B; b += CO_THREADS)
    if (co_ev[b]) atomicAdd(&bucket_ev[b], co_ev[b]);
}

// one workgroup.  Chunks of a bucket's event range are sized so that the whole pass has about `target` work items;
// the row blocks of one chunk are `stride` slots apart inside a group of stride * nrb slots (stride = 8 = number of
// XCDs: consecutive workgroup ids go round the XCDs, so the nrb readers of one event range share an L2).
__global__ __launch_bounds__(256) void co_plan_kernel(int B, int nrb, int target, int max_work,
                                                      const unsigned long long *__restrict__ bucket_ev,
                                                      unsigned long long *__restrict__ bucket_off,
                                                      CoWork *__restrict__ work, int *__restrict__ n_work) {
  extern __shared__ unsigned long long co_plan[];   // [B] event offsets, then [B] first chunk id (as u64)
  unsigned long long *ev_off = co_plan, *first = co_plan + B;
  __shared__ unsigned long long chunk_s;
  __shared__ int total_chunks;
  // two exclusive prefix sums over the buckets (event offsets, first chunk ids) by a block-wide scan, 256 buckets a
  // pass: a single thread walking B LDS words costs a dependent LDS round trip per bucket (measured: 11 us at B = 129)
  __shared__ unsigned long long wsum[2][4];
  __shared__ unsigned long long carry[2];
  __shared__ unsigned long long total_ev;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // pass 0: the event total (for the chunk size)
  {
    unsigned long long v = 0;
    for (int b = threadIdx.x; b < B; b += 256) v += bucket_ev[b];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if (lane == 0) wsum[0][wv] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned long long off = wsum[0][0] + wsum[0][1] + wsum[0][2] + wsum[0][3];
      unsigned long long chunk = (off * (unsigned long long)nrb + target - 1) / (unsigned long long)target;
      if (chunk < 16384ull) chunk = 16384ull;
      chunk_s = chunk;
      total_ev = off;
      carry[0] = 0;
      carry[1] = 0;
    }
    __syncthreads();
  }
  {
    const unsigned long long chunk = chunk_s;
    for (int b0 = 0; b0 < B; b0 += 256) {
      const int b = b0 + threadIdx.x;
      const unsigned long long n = b < B ? bucket_ev[b] : 0ull;
      const unsigned long long k = (n + chunk - 1) / chunk;
      unsigned long long in0 = n, in1 = k;   // inclusive scans inside the wave
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long u0 = __shfl_up(in0, d, 64), u1 = __shfl_up(in1, d, 64);
        if (lane >= d) {
          in0 += u0;
          in1 += u1;
        }
      }
      if (lane == 63) {
        wsum[0][wv] = in0;
        wsum[1][wv] = in1;
      }
      __syncthreads();
      unsigned long long base0 = carry[0], base1 = carry[1];
      for (int w = 0; w < wv; ++w) {
        base0 += wsum[0][w];
        base1 += wsum[1][w];
      }
      if (b < B) {
        ev_off[b] = base0 + in0 - n;
        first[b] = base1 + in1 - k;
      }
      __syncthreads();
      if (threadIdx.x == 255) {
        carry[0] = base0 + in0;
        carry[1] = base1 + in1;
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      const int k = (int)carry[1];
      total_chunks = k;
      const int groups = (k + 7) / 8;
      n_work[0] = groups * 8 * nrb <= max_work ? groups * 8 * nrb : 0;   // (cannot exceed: max_work is the bound below)
      bucket_off[B] = total_ev;
    }
  }
  __syncthreads();
  for (int b = threadIdx.x; b < B; b += 256) bucket_off[b] = ev_off[b];
  __syncthreads();
  const unsigned long long chunk = chunk_s;
  const int slots = (total_chunks + 7) / 8 * 8 * nrb;
  if (slots > max_work) return;
  // one thread per SLOT (a hot bucket has dozens of chunks: a thread per bucket would write them one after another)
  const int nchunks = total_chunks;
  for (int i = threadIdx.x; i < slots; i += 256) {
    const int group = i / (8 * nrb), in = i - group * 8 * nrb;
    const int rb = in >> 3, cid = group * 8 + (in & 7);
    CoWork w;
    w.q = -1; w.rb = rb; w.single = 0; w.pad = 0; w.e0 = 0; w.e1 = 0;
    if (cid < nchunks) {
      int lo = 0, hi = B - 1;   // the last bucket whose first chunk id is <= cid (empty buckets share their successor's)
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((int)first[mid] <= cid) lo = mid;
        else hi = mid - 1;
      }
      const int b = lo, c = cid - (int)first[b];
      const unsigned long long n = bucket_ev[b];
      const int nch = (int)((n + chunk - 1) / chunk);
      const unsigned long long per = (n + nch - 1) / nch;
      w.q = b;
      w.single = nch == 1;
      w.e0 = ev_off[b] + (unsigned long long)c * per;
      w.e1 = w.e0 + per < ev_off[b] + n ? w.e0 + per : ev_off[b] + n;
    }
    work[i] = w;
  }
}

// 256 pairs per workgroup.  Phase A, thread = pair: the pair's offset inside its bucket's event range (two-level: LDS
// counters per bucket inside the workgroup, one returning global add per (workgroup, non-empty bucket)); its record
// goes to LDS.  Phase B, wave = 64 of those pairs with their events FLATTENED: lane = event, 64 consecutive events per
// step whatever the pairs' contact counts are (a pair with 65 contacts would otherwise cost a second, almost empty,
// pass), four steps in flight -- every load of a step is independent of the previous step.
#define CO_XP 256
#ifndef CO_CU
#define CO_CU 16  // events a thread of co_count_lds_kernel keeps in flight
#endif
#ifndef CO_XU
#define CO_XU 8   // steps of 64 events a wave keeps in flight
#endif
__global__ __launch_bounds__(CO_XP) void co_expand_kernel(
    int B, const int8_t *__restrict__ seqs, const int32_t *__restrict__ contacts,
    const cb_count_pair *__restrict__ pairs, long long n_pairs, const int *__restrict__ qbuf,
    const unsigned long long *__restrict__ bucket_off, unsigned long long *__restrict__ cursor,
    unsigned *__restrict__ events) {
  extern __shared__ unsigned long long co_x[];   // [B] counters / bases
  __shared__ unsigned long long p_base[CO_XP];   // first event of the pair in the global event array
  __shared__ long long p_sa[CO_XP], p_sb[CO_XP], p_aux[CO_XP];
  __shared__ int p_woff[CO_XP + 4];              // per wave: exclusive prefix of the pairs' event counts, [64] + total
  unsigned long long *cnt = co_x;
  for (int b = threadIdx.x; b < B; b += CO_XP) cnt[b] = 0ull;
  __syncthreads();
  const long long p = (long long)blockIdx.x * CO_XP + threadIdx.x;
  int q = -1, n = 0;
  unsigned long long local = 0;
  if (p < n_pairs) {
    q = qbuf[p];
    if (q >= 0) {
      const cb_count_pair pr = pairs[p];
      n = pr.n;
      p_sa[threadIdx.x] = pr.seq_a;
      p_sb[threadIdx.x] = pr.seq_b;
      p_aux[threadIdx.x] = pr.aux;
      local = atomicAdd(&cnt[q], (unsigned long long)n);
    }
  }
  __syncthreads();
  for (int b = threadIdx.x; b < B; b += CO_XP) {
    const unsigned long long c = cnt[b];
    if (c) cnt[b] = bucket_off[b] + atomicAdd(&cursor[b], c);
  }
  __syncthreads();
  p_base[threadIdx.x] = q >= 0 ? cnt[q] + local : 0ull;
  // wave-level exclusive scan of n (a pair outside the grid has none)
  const int lane = threadIdx.x & 63, w0 = threadIdx.x & ~63, wv = threadIdx.x >> 6;
  int incl = n;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int up = __shfl_up(incl, d, 64);
    if (lane >= d) incl += up;
  }
  int *woff = p_woff + wv * 65;   // 4 waves x 65 ints
  woff[lane] = incl - n;
  if (lane == 63) woff[64] = incl;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __syncthreads();
  const int total = woff[64];
  int pl = 0;   // this lane's pair inside the wave (monotone over the steps)
  for (int e0 = 0; e0 < total; e0 += CO_XU * 64) {
    int pi[CO_XU], ci[CO_XU];
    bool on[CO_XU];
#pragma unroll
    for (int u = 0; u < CO_XU; ++u) {
      const int e = e0 + 64 * u + lane;
      on[u] = e < total;
      const int ec = on[u] ? e : total - 1;
      while (pl < 63 && woff[pl + 1] <= ec) ++pl;
      pi[u] = w0 + pl;
      ci[u] = ec - woff[pl];
    }
    int2 ij[CO_XU];
#pragma unroll
    for (int u = 0; u < CO_XU; ++u) ij[u] = *reinterpret_cast<const int2 *>(contacts + 2 * ((size_t)p_aux[pi[u]] + ci[u]));
    int code[CO_XU][4];
#pragma unroll
    for (int u = 0; u < CO_XU; ++u) {
      const int8_t *sa = seqs + p_sa[pi[u]], *sb = seqs + p_sb[pi[u]];
      code[u][0] = sa[ij[u].x];
      code[u][1] = sa[ij[u].y];
      code[u][2] = sb[ij[u].x];
      code[u][3] = sb[ij[u].y];
    }
#pragma unroll
    for (int u = 0; u < CO_XU; ++u) {
      if (on[u]) {
        const bool gap = (code[u][0] | code[u][1] | code[u][2] | code[u][3]) < 0;
        events[p_base[pi[u]] + ci[u]] = gap ? CO_EVENT_GAP
                                            : ((unsigned)code[u][0] | (unsigned)code[u][1] << 8 |
                                               (unsigned)code[u][2] << 16 | (unsigned)code[u][3] << 24);
      }
    }
  }
}

template <bool SYM>
__global__ __launch_bounds__(CO_THREADS) void co_count_lds_kernel(int S, int R, const unsigned *__restrict__ events,
                                                                  const CoWork *__restrict__ work,
                                                                  const int *__restrict__ n_work,
                                                                  unsigned long long *__restrict__ counts) {
  extern __shared__ unsigned co_hist[];   // [rows][S^2]
  if ((int)blockIdx.x >= n_work[0]) return;
  const CoWork w = work[blockIdx.x];
  if (w.q < 0) return;
  const int S2 = S * S;
  const int r0 = w.rb * R;
  const int rows = R < S2 - r0 ? R : S2 - r0;
  const int nb = rows * S2;
  for (int i = threadIdx.x; i < nb; i += CO_THREADS) co_hist[i] = 0u;
  __syncthreads();
  const unsigned urows = (unsigned)rows;
  auto add = [&](int row, int col) {
    const unsigned rr = (unsigned)(row - r0);
    if (rr < urows) atomicAdd(&co_hist[rr * S2 + col], 1u);
  };
  for (unsigned long long e = w.e0 + threadIdx.x; e < w.e1; e += (unsigned long long)CO_CU * CO_THREADS) {
    unsigned ev[CO_CU];
#pragma unroll
    for (int u = 0; u < CO_CU; ++u) {
      const unsigned long long eu = e + (unsigned long long)u * CO_THREADS;
      ev[u] = eu < w.e1 ? events[eu] : CO_EVENT_GAP;
    }
#pragma unroll
    for (int u = 0; u < CO_CU; ++u) {
      if (ev[u] == CO_EVENT_GAP) continue;
      const int ai = ev[u] & 0xFF, aj = (ev[u] >> 8) & 0xFF, bi = (ev[u] >> 16) & 0xFF, bj = ev[u] >> 24;
      const int s1 = ai * S + aj, s1r = aj * S + ai, s2 = bi * S + bj, s2r = bj * S + bi;
      add(s1, s2);
      add(s1r, s2r);
      if (SYM) {
        add(s2, s1);
        add(s2r, s1r);
      }
    }
  }
  __syncthreads();
  unsigned long long *dst = counts + ((size_t)w.q * S2 + r0) * S2;
  if (w.single) {
    for (int i = threadIdx.x; i < nb; i += CO_THREADS) {
      const unsigned v = co_hist[i];
      if (v) dst[i] += v;
    }
  } else {
    for (int i = threadIdx.x; i < nb; i += CO_THREADS) {
      const unsigned v = co_hist[i];
      if (v) atomicAdd(&dst[i], (unsigned long long)v);
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

// ------------------------------------------------------------------------ JTT-IPW statistics
// The two S x S sums the closed-form initialiser is made of (reference: cherryml/estimation/_jtt_ipw.py:66-110):
//   F[i][j] = sum_b sym(C_b)[i][j],   R[i][j] = sum_b sym(C_b)[i][j] / t_b,   sym(C) = (C + C^T) / 2  (or C itself)
// from the resident count tensor (u64 integers in units of `unit`, as the counting kernels leave them, or doubles).
// One streaming pass: workgroup = (upper-triangular 16 x 16 tile pair, chunk of JT_BC buckets); thread (r, c) sums its
// entry of the tile AND of the mirror tile (both read along rows: 128-byte runs), the mirror sums are transposed once
// through LDS at the end (sums are linear), so every count is read exactly once.  Partials per chunk, summed in a fixed
// order by jtt_stats_reduce: the result does not depend on the launch geometry or on timing.
#define JT_BC 16
template <typename T>
__global__ __launch_bounds__(256) void jtt_stats_partial(int S, int B, const T *__restrict__ C, const double *__restrict__ grid,
                                                         double unit, int symmetrize, double *__restrict__ part) {
  __shared__ double sF[16][17], sR[16][17];
  const int nt = (S + 15) / 16;
  // tile pair index -> (ti <= tj)
  int ti = 0, rem = blockIdx.x;
  while (rem >= nt - ti) {
    rem -= nt - ti;
    ++ti;
  }
  const int tj = ti + rem;
  const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
  const int i = ti * 16 + r, j = tj * 16 + c;     // my entry of tile (ti, tj)
  const int i2 = tj * 16 + r, j2 = ti * 16 + c;   // my entry of the mirror tile (tj, ti)
  const bool ok = i < S && j < S, ok2 = i2 < S && j2 < S;
  const int b0 = blockIdx.y * JT_BC, b1 = min(B, b0 + JT_BC);
  double aF = 0.0, aR = 0.0, mF = 0.0, mR = 0.0;
  const size_t SS = (size_t)S * S;
  for (int b = b0; b < b1; ++b) {
    const double t = grid[b];
    const double v = ok ? (double)C[b * SS + (size_t)i * S + j] : 0.0;
    const double w = ok2 ? (double)C[b * SS + (size_t)i2 * S + j2] : 0.0;
    aF += v;
    aR += v / t;
    mF += w;
    mR += w / t;
  }
  double *pF = part + (size_t)blockIdx.y * 2 * SS, *pR = pF + SS;
  if (!symmetrize) {   // F = sum_b C_b: both tiles as they are
    if (ok) {
      pF[(size_t)i * S + j] = aF * unit;
      pR[(size_t)i * S + j] = aR * unit;
    }
    if (ok2 && ti != tj) {
      pF[(size_t)i2 * S + j2] = mF * unit;
      pR[(size_t)i2 * S + j2] = mR * unit;
    }
    return;
  }
  sF[r][c] = mF;
  sR[r][c] = mR;
  __syncthreads();
  const double symF = 0.5 * (aF + sF[c][r]) * unit, symR = 0.5 * (aR + sR[c][r]) * unit;
  __syncthreads();
  if (ok) {
    pF[(size_t)i * S + j] = symF;
    pR[(size_t)i * S + j] = symR;
  }
  if (ti != tj) {   // the mirror tile is the transpose: through LDS, so that the stores run along rows too
    sF[r][c] = symF;
    sR[r][c] = symR;
    __syncthreads();
    if (ok2) {
      pF[(size_t)i2 * S + j2] = sF[c][r];
      pR[(size_t)i2 * S + j2] = sR[c][r];
    }
  }
}

__global__ void jtt_stats_reduce(size_t n2, int nchunks, const double *__restrict__ part, double *__restrict__ F,
                                 double *__restrict__ R) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n2) return;
  const size_t SS = n2 / 2;
  double s = 0.0;
  for (int k = 0; k < nchunks; ++k) s += part[(size_t)k * n2 + e];
  if (e < SS) F[e] = s;
  else R[e - SS] = s;
}
