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
