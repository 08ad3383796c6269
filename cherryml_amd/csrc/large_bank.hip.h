// Large-state path (S > 32; co-evolution 400x400).  MFMA-bound.
//
// All internal matrices are LD x LD with LD = roundup(S, 16), zero padded, so
// every 16x16 MFMA tile is either fully inside or fully outside the matrix.
// Every product of the algorithm is brought to the single form
//
//     C[m][n] = sum_k Aop[k][m] * Bop[k][n]          ("TN", both operands k-major)
//
// by choosing which of U / U^T, G^T, T, M^T is stored, so one
// kernel template serves them all:
//   K1  Pt_b   = I + t_b A + (U^T diag(F_b))^T U^T      Aop = Ut (row-scaled), Bop = Ut
//       epilogue: loss partial, Gt_b^T = -C_b^T / Pt_b / n
//   K2  T_b    = Gt_b U                                  Aop = Gt_b^T, Bop = U
//   K3  Mt    += (T_b^T U) o Phi_b  over a chunk of b    Aop = T_b,   Bop = U
//   K4a X      = M U^T                                   Aop = Mt,    Bop = Ut
//   K4b dA     = U X  -> dQ = D^1/2 dA D^-1/2            Aop = Ut,    Bop = X
//
// Tile: 80 x 80 per workgroup of 4 wavefronts (400 = 5 * 80), one wave per SIMD; K is staged through
// LDS in steps of 16 with register prefetch of the next step.  K1..K3 are templates over the element
// type: float64, or float32 operands with float32 MFMA accumulation (cb_create(dtype = CB_F32): twice
// the MFMA rate, half the bytes; loss partials, the divided differences and the bucket sum stay float64).
#pragma once
#include "common.hip.h"
#include <type_traits>

#define LG_TM 80
#define LG_TN 80
#define LG_KT 16
#define LG4_THREADS 256

// Element type of the bank products: double (v_mfma_f64_16x16x4_f64, 78.6 TFLOP/s) or float
// (v_mfma_f32_16x16x4_f32, 157 TFLOP/s; cb_create(dtype = CB_F32)).  A / B operand lane maps are the
// same for both (lane l: A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15]); the C / D maps differ:
// register r of lane l is row (l >> 4) + 4 r in f64 and row 4 (l >> 4) + r in f32 (guide 3, "C/D layout").
typedef float f4 __attribute__((ext_vector_type(4)));
template <typename T> struct Mfma;
template <> struct Mfma<double> {
  typedef d4 acc_t;
  typedef double2 vec_t;                      // one 16-byte chunk of a panel row
  static constexpr int VEC = 2;
  static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) { return mfma_f64(a, b, c); }
  static __device__ __forceinline__ int row(int hi, int r) { return hi + 4 * r; }
  static constexpr int RSTEP = 4;             // row(hi, r) = row(hi, 0) + RSTEP * r
};
template <> struct Mfma<float> {
  typedef f4 acc_t;
  typedef float4 vec_t;
  static constexpr int VEC = 4;
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int hi, int r) { return 4 * hi + r; }
  static constexpr int RSTEP = 1;
};
__device__ __forceinline__ void vec_scale(double2 &v, double s) { v.x *= s; v.y *= s; }
__device__ __forceinline__ void vec_scale(float4 &v, float s) { v.x *= s; v.y *= s; v.z *= s; v.w *= s; }

template <typename T>
struct GemmOperands {
  const T *A;   // [K][lda] k-major
  const T *B;   // [K][ldb]
  int lda, ldb;
  int M, N, K;       // all multiples of 16
  const T *kscale;  // optional [K] multiplier applied to rows of Aop
};

// ---- the 80 x 80 tile with FOUR waves per workgroup, one per SIMD ----------------------------------
// (A 5-wave workgroup -- one wave per 16-row strip -- puts two of its waves on one SIMD, which one
// rotates from workgroup to workgroup, profiles/tools/simd_probe; with four workgroups on a CU the
// SIMDs then carry 4..7 waves and the busiest one sets the pace.)  The 25 MFMA tiles of the 80 x 80
// tile are dealt 7/6/6/6: wave w owns strip w (acc[0..4]) and tile (4, w) of the fifth strip (ax0);
// tile (4, 4) (ax1) is split over K among the four waves (6.25 tiles each), see the loop.
// Panel loads: thread -> (row tid / 16 of the 16-row K-step, 16-byte chunks tid % 16, + 16, + 32 of that
// row's 40 (f64) or 20 (f32)), so the loads of a thread share ONE row pointer and ONE row scale.  (A flat
// chunk = tid + 256 u mapping needed three pointers and three scales per operand: 18 loop-invariant
// registers, which the compiler spilled to scratch and re-loaded in every K-step: +170 MB of traffic
// in K1.)  SCALE is a template parameter and the scale a register: a nullable `const T *` made the
// compiler keep it in SCRATCH memory (a store and a load per K-step on the critical path).
template <typename T> struct Panel {
  static constexpr int VEC = Mfma<T>::VEC;
  static constexpr int CH = LG_TM / VEC;           // 16-byte chunks per panel row: 40 / 20
  static constexpr int NL = (CH + 15) / 16;        // loads per thread: 3 / 2
  static constexpr int LAST = CH - 16 * (NL - 1);  // threads (of 16 per row) that take part in the last one: 8 / 4
};
// Panel loads are BUFFER loads: address = resource (the operand's base, four SGPRs, made once per tile) + this thread's
// byte offsets (the same in every K-step) + the K-step's byte offset (a scalar): no vector address arithmetic in the K loop,
// and no bounds logic -- the MFMAs share the vector issue port, and the checks (six exec-mask branches, two dozen v_mov of
// zero fill per K-step) cost 22 us of the bank's 680.  K is a multiple of the K-step; a column past the matrix (a tile that
// hangs over its edge, the idle lanes of the last chunk) is CLAMPED to the row's last chunk instead of zero-filled: what it
// brings only reaches outputs past the edge, which every epilogue discards.
typedef unsigned int lg_u4 __attribute__((ext_vector_type(4)));
template <typename T> struct PanelSrc {
  __amdgpu_buffer_rsrc_t rsrc;
  int voff[Panel<T>::NL];   // bytes
};
template <typename T>
__device__ __forceinline__ PanelSrc<T> lg4_panel_src(const T *P, int ld, int cols_total, int c0, int tid) {
  constexpr int VEC = Panel<T>::VEC, NL = Panel<T>::NL;
  const int kr = tid >> 4, cq = tid & 15;
  PanelSrc<T> s;
  // (raw buffer, stride 0, range = 2 GB: the offsets below stay inside one LD x LD matrix)
  s.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(P), 0, 0x7fffffff, 0x00027000);
#pragma unroll
  for (int u = 0; u < NL; ++u) s.voff[u] = (kr * ld + min(c0 + VEC * (cq + 16 * u), cols_total - VEC)) * (int)sizeof(T);
  return s;
}
template <typename T, bool SCALE>
__device__ __forceinline__ void lg4_load_panel(const PanelSrc<T> &src, int ld, int k0, typename Mfma<T>::vec_t (&reg)[Panel<T>::NL],
                                               const T *__restrict__ kscale, T &sc, int tid) {
  typedef typename Mfma<T>::vec_t vec_t;
  constexpr int NL = Panel<T>::NL;
  const int soff = k0 * ld * (int)sizeof(T);
#pragma unroll
  for (int u = 0; u < NL; ++u)
    reg[u] = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(src.rsrc, src.voff[u], soff, 0));
  if (SCALE) sc = kscale[k0 + (tid >> 4)];
}
template <typename T, bool SCALE>
__device__ __forceinline__ void lg4_store_panel(T *s, const typename Mfma<T>::vec_t (&reg)[Panel<T>::NL], T sc, int tid) {
  typedef typename Mfma<T>::vec_t vec_t;
  constexpr int VEC = Panel<T>::VEC, NL = Panel<T>::NL, LAST = Panel<T>::LAST;
  const int kr = tid >> 4, cq = tid & 15;
  T *row = s + kr * LG_TM + VEC * cq;
#pragma unroll
  for (int u = 0; u < NL; ++u) {
    vec_t v = reg[u];
    if (SCALE) vec_scale(v, sc);
    if (u < NL - 1 || cq < LAST) *reinterpret_cast<vec_t *>(row + 16 * VEC * u) = v;
  }
}
// What the ticket loop of k123_bank has a tile do on the side, from inside its K loop where the round trips cost nothing:
//   deferred: the counter that announces the PREVIOUS tile of this workgroup, bumped behind the wait for this tile's first
//             panels (the wait for the previous tile's stores is then free);
//   draw:     the ticket counter the NEXT ticket is drawn from, at the third K-step; the answer is left in
//             *drawn_lds (a word of the panel buffer that no epilogue uses) once the loop is over.
struct BankHooks {
  unsigned int *deferred = nullptr;
  unsigned int *draw = nullptr;
  int *drawn_lds = nullptr;
};

// sA / sB hold TWO K-steps each (double buffer): one barrier per K-step; the global loads of step
// k+1 are in flight during the MFMAs of step k.  LDS row stride 80 elements is conflict-free for the
// operand reads in both widths (f64: 2 * 80 mod 64 = 32; f32: 80 mod 64 = 16, four k rows per read).
// ZERO = false: accumulate onto the tile the registers already hold (a second operand pair of the same
// output tile: general_large.hip.h); ax1 must then be complete in wave 0 and ZERO elsewhere on entry, which
// is how this routine leaves it.
//
// KG = 2 (round 5): the SAME tile on EIGHT waves -- two K-groups of four waves, group g takes the K-steps g, g + 2, ... with
// its own pair of panel buffers (sA / sB of group g = the caller's + g * 4 * LG_KT * LG_TM) and every wave the same seven
// accumulators as before.  Why: one 80 x 80 x 400 tile is 16.7 us of matrix pipe per SIMD, but a four-wave workgroup ALONE on
// its CU takes 30 us for it (one wave per SIMD: every K-step exposes its LDS round trip and barrier), and 65-70 us beside three
// others.  A launch is a chain of K1 -> K2 -> K3 tiles per bucket; with few buckets (the reference's real bank: 43; one rank's
// share of an 8-rank job: 17) the chip is never full and the chain of three 30-45 us tiles IS the launch, and the long bank
// ends in a drain of half-empty CUs at a third of the pipe.  Two waves per SIMD on ONE tile halve the tile's latency at the
// same occupancy (two workgroups of 80 KB per CU instead of four of 40 KB).  After the K loop the groups exchange partial
// accumulators through the (free) panel buffers and SHARE the epilogue: group 0 finishes the wave's tiles 0 .. 2 of its strip
// and tile (4, 4), group 1 tiles 3, 4 and (4, wave) -- see lg_owns_*.  Sums are commutative pairs: the bits do not depend on
// which group finishes a tile.  KG = 1 is the four-wave routine of rounds 2-4, bit for bit.
template <int KG> __device__ __forceinline__ bool lg_owns_acc(int kg, int j) { return KG == 1 || (kg == 0) == (j < 3); }
template <int KG> __device__ __forceinline__ bool lg_owns_ax0(int kg) { return KG == 1 || kg == 1; }
template <int KG> __device__ __forceinline__ bool lg_owns_ax1(int kg) { return KG == 1 || kg == 0; }   // (and wave 0 of the group)

template <typename T, bool SCALE, bool ZERO = true, int KG = 1>
__device__ __forceinline__ void lg4_gemm_tile(const GemmOperands<T> &g, int m0, int n0, T *sA, T *sB,
                                              typename Mfma<T>::acc_t (&acc)[5], typename Mfma<T>::acc_t &ax0,
                                              typename Mfma<T>::acc_t &ax1, int tid = threadIdx.x,
                                              const BankHooks &hooks = BankHooks{}) {
  // (tid: threadIdx.x -- or an opaque copy of it, k123_bank: what is derived from it then stays inside the tile)
  static_assert(KG == 1 || KG == 2, "one or two K-groups of four waves");
  static_assert(KG == 1 || ZERO, "the accumulate-onto form exists for four waves only");
  typedef typename Mfma<T>::acc_t acc_t;
  typedef typename Mfma<T>::vec_t vec_t;
  // (the wave index as a SCALAR: the tests on it in the K loop are then scalar branches, not v_cmp + exec masks)
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = KG == 1 ? 0 : wv >> 2, wave = wv & 3, t4 = tid & 255;   // K-group, wave of the group, thread of the group
  const int lane = tid & 63, lo = lane & 15, hi = lane >> 4;
  T *const sAll = sA;   // the whole panel buffer (KG groups x (A | B) x two K-steps): the exchange area behind the K loop
  if (KG > 1) {
    sA += kg * (4 * LG_KT * LG_TM);
    sB += kg * (4 * LG_KT * LG_TM);
  }
  if (ZERO) {
#pragma unroll
    for (int j = 0; j < 5; ++j) acc[j] = acc_t{};
    ax0 = acc_t{};
    ax1 = acc_t{};
  }
  vec_t ra[Panel<T>::NL], rb[Panel<T>::NL];
  T sc = T(1), one = T(1);
  const int nk = g.K / LG_KT, nit = (nk + KG - 1) / KG;   // K-steps; iterations of a group (group kg runs step it * KG + kg)
  const PanelSrc<T> srcA = lg4_panel_src<T>(g.A, g.lda, g.M, m0, t4), srcB = lg4_panel_src<T>(g.B, g.ldb, g.N, n0, t4);
  if (kg < nk) {
    lg4_load_panel<T, SCALE>(srcA, g.lda, kg * LG_KT, ra, g.kscale, sc, t4);
    lg4_load_panel<T, false>(srcB, g.ldb, kg * LG_KT, rb, nullptr, one, t4);
  }
  __syncthreads();  // the previous tile's readers of buffer 0 are done
  if (kg < nk) {
    lg4_store_panel<T, SCALE>(sA, ra, sc, t4);
    lg4_store_panel<T, false>(sB, rb, one, t4);
  }
  unsigned int *deferred = hooks.deferred;
  if (deferred) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's stores of the previous tile have been performed
  __syncthreads();
  if (deferred && tid == 0) __hip_atomic_fetch_add(deferred, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int drawn = -1;
  const int draw_it = nit > 2 ? 2 / KG : 0;   // (early: the answer has the rest of the tile; drawn in the middle of the loop or four
                                              // K-steps before its end, the epilogues waited 5-9 us for it -- a counter serves ~1.5
                                              // draws per us, the answers queue up)
  for (int it = 0; it < nit; ++it) {
    const int kt = it * KG + kg;
    const T *cA = sA + (it & 1) * (LG_KT * LG_TM), *cB = sB + (it & 1) * (LG_KT * LG_TN);
    if (hooks.draw && it == draw_it && tid == 0)
      drawn = (int)__hip_atomic_fetch_add(hooks.draw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool more = kt + KG < nk;   // (wave-uniform: kg is a scalar)
    if (more) {
      lg4_load_panel<T, SCALE>(srcA, g.lda, (kt + KG) * LG_KT, ra, g.kscale, sc, t4);
      lg4_load_panel<T, false>(srcB, g.ldb, (kt + KG) * LG_KT, rb, nullptr, one, t4);
    }
    if (KG == 1 || kt < nk) {
#pragma unroll
      for (int s = 0; s < LG_KT / 4; ++s) {
        const T av = cA[(4 * s + hi) * LG_TM + 16 * wave + lo];
        const T a4 = cA[(4 * s + hi) * LG_TM + 64 + lo];
        T bv[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) bv[j] = cB[(4 * s + hi) * LG_TN + 16 * j + lo];
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[j] = Mfma<T>::mma(av, bv[j], acc[j]);
        // tile (4, wave): column block `wave` (a wave-uniform choice among registers)
        // (read from LDS again rather than chosen among bv[0..3] with six v_cndmask per sub-step: the MFMAs share the vector
        // issue port, every VALU instruction of the K loop costs matrix time -- 0.707 -> 0.685 ms for the bank)
        const T bx = cB[(4 * s + hi) * LG_TN + 16 * wave + lo];
        ax0 = Mfma<T>::mma(a4, bx, ax0);
        // tile (4, 4): its K range is dealt round-robin to the four waves of the group (6.25 MFMA tiles each
        // instead of 7/6/6/6); the partial sums meet in wave 0 below
        if ((it & 3) == wave) ax1 = Mfma<T>::mma(a4, bv[4], ax1);
      }
    }
    if (more) {
      lg4_store_panel<T, SCALE>(sA + ((it + 1) & 1) * (LG_KT * LG_TM), ra, sc, t4);
      lg4_store_panel<T, false>(sB + ((it + 1) & 1) * (LG_KT * LG_TN), rb, one, t4);
    }
    __syncthreads();
  }
  if (KG == 1) {
    if (wave != 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) sA[(wave - 1) * 256 + r * 64 + lane] = ax1[r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) ax1[r] += (sA[r * 64 + lane] + sA[256 + r * 64 + lane]) + sA[512 + r * 64 + lane];
    } else {
      ax1 = acc_t{};     // (the partial sums now live in wave 0)
    }
  } else {
    // exchange area (the panel buffers are free: every wave is past the loop's last barrier): [7][256] the partial sums of
    // tile (4, 4) of every wave but (0, 0) | [4][3][256] group 0's partials of the tiles group 1 finishes (acc[3], acc[4],
    // ax0 of wave w) | [4][3][256] group 1's partials of acc[0 .. 2]: 7936 elements of the buffer's 10240
    T *px = sAll, *to1 = sAll + 7 * 256, *to0 = to1 + 12 * 256;
    if (wv != 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) px[(wv - 1) * 256 + r * 64 + lane] = ax1[r];
    }
    if (kg == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        to1[(wave * 3 + 0) * 256 + r * 64 + lane] = acc[3][r];
        to1[(wave * 3 + 1) * 256 + r * 64 + lane] = acc[4][r];
        to1[(wave * 3 + 2) * 256 + r * 64 + lane] = ax0[r];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        to0[(wave * 3 + 0) * 256 + r * 64 + lane] = acc[0][r];
        to0[(wave * 3 + 1) * 256 + r * 64 + lane] = acc[1][r];
        to0[(wave * 3 + 2) * 256 + r * 64 + lane] = acc[2][r];
      }
    }
    __syncthreads();
    if (kg == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[0][r] += to0[(wave * 3 + 0) * 256 + r * 64 + lane];
        acc[1][r] += to0[(wave * 3 + 1) * 256 + r * 64 + lane];
        acc[2][r] += to0[(wave * 3 + 2) * 256 + r * 64 + lane];
      }
      if (wave == 0) {   // fixed order: ((own + w1) + (w2 + w3)) + ((w4 + w5) + (w6 + w7))
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = r * 64 + lane;
          ax1[r] = ((ax1[r] + px[o]) + (px[256 + o] + px[512 + o])) + ((px[768 + o] + px[1024 + o]) + (px[1280 + o] + px[1536 + o]));
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[3][r] += to1[(wave * 3 + 0) * 256 + r * 64 + lane];
        acc[4][r] += to1[(wave * 3 + 1) * 256 + r * 64 + lane];
        ax0[r] += to1[(wave * 3 + 2) * 256 + r * 64 + lane];
      }
    }
  }
  __syncthreads();   // the panel buffer is free again (callers reuse it)
  if (hooks.draw && tid == 0) *hooks.drawn_lds = drawn;
}

// Diagnostic build only (-DCB_CLOCK_STAMP, profiles/tools/clock_probe.py): the clock the chip holds inside the K loops of
// K1..K3 = delta s_memtime / delta s_memrealtime * 100 MHz per workgroup (guide, 'DVFS give-back' item 6).  The stamps go to a
// buffer of their own that no kernel reads; the shipped library has none of this.
#ifdef CB_CLOCK_STAMP
// per tile (index = bucket * tiles of the stage + tile): [0] cycles of the K loop, [1] 100 MHz ticks of the K loop, [2] tick
// at the start of the tile, [3] tick at the end of the K loop, [4] tick at the end of the tile, [5] HW_ID | XCC_ID << 32
__device__ unsigned long long cb_clock_stamps[3][4096][6];
#define CB_STAMP_BEGIN(id)                                                                         \
  const int cs_id = (id);                                                                          \
  const unsigned long long cs_c0 = __builtin_amdgcn_s_memtime(), cs_r0 = __builtin_amdgcn_s_memrealtime()
#define CB_STAMP_END(kid)                                                                          \
  do {                                                                                             \
    const unsigned long long cs_c1 = __builtin_amdgcn_s_memtime(), cs_r1 = __builtin_amdgcn_s_memrealtime(); \
    if (threadIdx.x == 0 && cs_id < 4096) {                                                        \
      cb_clock_stamps[kid][cs_id][0] = cs_c1 - cs_c0;                                              \
      cb_clock_stamps[kid][cs_id][1] = cs_r1 - cs_r0;                                              \
      cb_clock_stamps[kid][cs_id][2] = cs_r0;                                                      \
      cb_clock_stamps[kid][cs_id][3] = cs_r1;                                                      \
      cb_clock_stamps[kid][cs_id][5] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |               \
                                       (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32;         \
    }                                                                                              \
  } while (0)
#define CB_STAMP_FINISH(kid)                                                                       \
  do {                                                                                             \
    if (threadIdx.x == 0 && cs_id < 4096) cb_clock_stamps[kid][cs_id][4] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define CB_STAMP_BEGIN(id) do { } while (0)
#define CB_STAMP_END(kid) do { } while (0)
#define CB_STAMP_FINISH(kid) do { } while (0)
#endif

// f(row, col, value) for every element of the tile this lane owns (KG = 2: of the MFMA tiles its K-group finishes)
template <typename T, int KG = 1, typename F>
__device__ __forceinline__ void lg_for_each(int m0, int n0, const typename Mfma<T>::acc_t (&acc)[5],
                                            const typename Mfma<T>::acc_t &ax0, const typename Mfma<T>::acc_t &ax1,
                                            F &&f, int tid = threadIdx.x) {
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), kg = KG == 1 ? 0 : wv >> 2, wave = wv & 3;
  const int lane = tid & 63, lo = lane & 15, hi = lane >> 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int rr = Mfma<T>::row(hi, r);
    const int row = m0 + 16 * wave + rr;
#pragma unroll
    for (int j = 0; j < 5; ++j)
      if (lg_owns_acc<KG>(kg, j)) f(row, n0 + 16 * j + lo, acc[j][r]);
    const int row4 = m0 + 64 + rr;
    if (lg_owns_ax0<KG>(kg)) f(row4, n0 + 16 * wave + lo, ax0[r]);
    if (wave == 0 && lg_owns_ax1<KG>(kg)) f(row4, n0 + 64 + lo, ax1[r]);
  }
}

__device__ __forceinline__ void lg_wave_lds_fence() {   // LDS visibility inside ONE wavefront (no workgroup barrier)
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// elementwise pieces of the K1 epilogue in the bank's element type (the loss itself is always
// accumulated in float64)
__device__ __forceinline__ double k1_log(double x) { return fast_log(x); }
__device__ __forceinline__ double k1_rcp(double x) { return fast_rcp(x); }
__device__ __forceinline__ float k1_log(float x) { return logf(x); }
__device__ __forceinline__ float k1_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// ------------------------------------------------------------------ K1
template <typename T, typename TG = T>   // TG: element type Gt^T is written in (CB_MIXED: T = double, TG = float)
struct K1Args {
  int S, LD, B;
  const T *Ut;           // [LD][LD]  Ut[k][i] = U[i][k]
  const T *A;            // [LD][LD]  symmetric
  const double *t;       // [B]
  const T *F;            // [B][LD]   phi2(t_b lam_k) (split) or exp(t_b lam_k)
  const double *sigma;   // max |A_ii|: bucket uses the split form iff 2 sigma t_b <= 1
  const T *Ct;           // [B][LD][LD] transposed counts (padded)
  TG *Gt;                // [B][LD][LD] out: Gt^T
  double *loss_part;     // [B * tiles] out
  double inv_n;
  const double *dsq;     // [LD] sqrt(pi) (expm mode)
  double *P;             // [B][S][S] (expm mode, T = double only) or null
  const unsigned long long *skip = nullptr;   // device word: non-zero => return at once (the planned eigensolve stalled; the
                                              // host repeats the evaluation, train_host.hip.h)
};

// Pt_b is symmetric: only the tilesN (tilesN + 1) / 2 tiles with tm <= tn run the main loop; an
// off-diagonal tile serves both (row, col) and (col, row) in its epilogue (same Pt value, its own
// count and its own Gt^T entry).  40 % fewer MFMAs than the full grid at LD = 400.
// WT (the fused bank kernel k123_bank below): the outputs another workgroup of the SAME launch reads are written through to
// memory (agent-scope stores), see there.
template <bool WT, typename T>
__device__ __forceinline__ void bank_store(T *p, T v) {
  if (WT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}

// one (bucket b, upper-triangular tile `tile`) of K1; sAB: 4 * LG_KT * LG_TM elements of LDS (A panels | B panels, two K-steps
// each; after the K loop: the transposition buffer); vid = b * tiles + tile indexes the loss partial
// RAW (the time-basis bank, tbasis.hip.h): the product U diag(F_b) U^T ITSELF -- no I + t A, no counts -- written to Gt[b] as a
// padded LD x LD matrix, its upper 80 x 80 tiles only: Psi_r of a skeleton bucket, or P_b of a long-branch bucket.
template <typename T, typename TG, bool EXPM, bool WT, int KG = 1, bool RAW = false>
__device__ __forceinline__ void k1_tile(const K1Args<T, TG> &a, int b, int tile, T *sAB, int tid,
                                        const BankHooks &hooks = BankHooks{}) {
  T *sA = sAB, *sB = sAB + 2 * LG_KT * LG_TM;
  typedef typename Mfma<T>::acc_t acc_t;
  const int tilesN = (a.LD + LG_TN - 1) / LG_TN, tiles = tilesN * (tilesN + 1) / 2;
  const int vid = b * tiles + tile;
  int tm = 0;
  while (tile >= tilesN - tm) {  // row tm of the upper triangle holds tilesN - tm tiles
    tile -= tilesN - tm;
    ++tm;
  }
  const int tn = tm + tile;
  const int m0 = tm * LG_TM, n0 = tn * LG_TN;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), kg = KG == 1 ? 0 : wv >> 2, wave = wv & 3, lane = tid & 63;
  const size_t boff = (size_t)b * a.LD * a.LD;
  GemmOperands<T> g{a.Ut, a.Ut, a.LD, a.LD, a.LD, a.LD, a.LD, a.F + (size_t)b * a.LD};
  acc_t acc[5], ax0, ax1;
  CB_STAMP_BEGIN(vid);
  lg4_gemm_tile<T, true, true, KG>(g, m0, n0, sA, sB, acc, ax0, ax1, tid, hooks);
  CB_STAMP_END(0);

  const double tb = a.t[b];
  const T tbT = (T)tb, inv_nT = (T)a.inv_n;
  const bool split = !RAW && tb * 2.0 * (*a.sigma) <= 1.0;  // see small_bank.hip.h
  const bool mirror = tm != tn;
  double lossacc = 0.0;
  const int lo = lane & 15, hi = lane >> 4;
  const T *__restrict__ Ct = a.Ct + boff;
  const T *__restrict__ Am = a.A;
  TG *__restrict__ Gt = a.Gt + boff;
  // One 16 x 16 MFMA tile at a time, in three phases: ALL its loads (counts, mirrored counts, I + tA), then
  // the arithmetic, then ALL its stores.  Written element by element (load, log, store, next element) the
  // epilogue compiled to 25 chains of global_load -> s_waitcnt vmcnt(0) -> ... -> store, i.e. 25 exposed HBM
  // latencies per wave (the counts are streamed, never cached): 40 % of K1's time.
  // The MIRRORED entries (col, row) of an off-diagonal tile -- same Pt value, their own count and Gt^T entry:
  // in the accumulator layout their addresses are 32-byte pieces of 64 different rows (K1 moved 1.8x its
  // algorithmic bytes).  So every wave transposes the tile's Pt values through a private 16 x 17 patch of
  // LDS (the panel buffers are free after the K loop; no workgroup barrier, the four waves stay independent)
  // and handles the mirrored entries with the lanes running along THEIR rows: 128-byte runs, like the
  // direct entries.  (Staging whole 80 x 48 parts through LDS with barriers in between gave the same
  // traffic, 400 MB, but serialised the waves: 0.245 -> 0.262 ms.  Issuing tile j + 1's count loads before tile j's
  // arithmetic -- a software pipeline over the 7 tiles of a wave -- costs 4 .. 56 spilled registers at the 128 the
  // four-workgroup occupancy allows and measured 0.240 against 0.236 ms without it: the other three workgroups of
  // the CU already cover those latencies.)
  T *sW = sAB + wv * (2 * 16 * 17);   // two patches per wave: log Pt, 1 / Pt
  auto tile_epilogue = [&](int rbase, int cbase, const acc_t &v) {   // wave-uniform tile origin
    if (rbase >= a.LD || cbase >= a.LD) return;
    const int col = cbase + lo;
    // register r of the tile is row rl0 + RSTEP r: one base offset + a constant stride per index array
    // (LD <= 1024: offsets inside one matrix fit an int)
    constexpr int RS = Mfma<T>::RSTEP;
    const int rl0 = Mfma<T>::row(hi, 0), step = RS * a.LD;
    const int idx0 = (rbase + rl0) * a.LD + col;
    const int idm0 = (cbase + rl0) * a.LD + rbase + lo;   // mirrored entry (cbase + rl, rbase + lo): lanes run along its row
    int rl[4], row[4], idx[4], idm[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      rl[r] = rl0 + RS * r;
      row[r] = rbase + rl[r];
      idx[r] = idx0 + r * step;
      idm[r] = idm0 + r * step;
    }
    if constexpr (RAW) {
      // (the upper 80 x 80 tiles only: tb_ew, the one reader, visits the upper block triangle -- a diagonal tile is computed whole)
#pragma unroll
      for (int r = 0; r < 4; ++r) Gt[idx[r]] = (TG)v[r];
      return;
    }
    T c1[4], c2[4], av[4], pt[4];
    if (!EXPM) {
#pragma unroll
      for (int r = 0; r < 4; ++r) c1[r] = Ct[idx[r]];
      if (mirror) {
#pragma unroll
        for (int r = 0; r < 4; ++r) c2[r] = Ct[idm[r]];
      }
    }
    if (split) {
#pragma unroll
      for (int r = 0; r < 4; ++r) av[r] = Am[idx[r]];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      pt[r] = v[r];
      if (split) pt[r] += tbT * av[r] + (row[r] == col ? T(1) : T(0));
      else if (row[r] >= a.S || col >= a.S) pt[r] = T(1);  // pad (never used: C = 0 there)
    }
    if (EXPM) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (row[r] < a.S && col < a.S) {
          a.P[(size_t)b * a.S * a.S + (size_t)row[r] * a.S + col] = (double)pt[r] * a.dsq[col] / a.dsq[row[r]];
          if (mirror) a.P[(size_t)b * a.S * a.S + (size_t)col * a.S + row[r]] = (double)pt[r] * a.dsq[row[r]] / a.dsq[col];
        }
      }
      return;
    }
    // log and reciprocal ONCE per Pt value: the mirrored entry (same value, other lane) takes them through the patches
    T lg[4], rc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      lg[r] = k1_log(pt[r]);
      rc[r] = k1_rcp(pt[r]);
    }
    TG g1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool nz = c1[r] != T(0);
      lossacc = fma(-(double)c1[r], (double)(nz ? lg[r] : T(0)), lossacc);   // (Pt <= 0 only where C = 0: rounding of a tiny entry)
      g1[r] = (TG)(nz ? -c1[r] * inv_nT * rc[r] : T(0));
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) bank_store<WT>(&Gt[idx[r]], g1[r]);
    if (mirror) {
      // value(row = rbase + rl[r], col = cbase + lo) -> patch[lo][rl[r]]; read back patch[rl[r]][lo] =
      // value(row = rbase + lo, col = cbase + rl[r]), that of the mirrored entry this lane now owns
      lg_wave_lds_fence();   // the previous tile's reads of the patches are done
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sW[lo * 17 + rl[r]] = lg[r];
        sW[16 * 17 + lo * 17 + rl[r]] = rc[r];
      }
      lg_wave_lds_fence();
      TG g2[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const T lm = sW[rl[r] * 17 + lo], rm = sW[16 * 17 + rl[r] * 17 + lo];
        const bool nz = c2[r] != T(0);
        lossacc = fma(-(double)c2[r], (double)(nz ? lm : T(0)), lossacc);
        g2[r] = (TG)(nz ? -c2[r] * inv_nT * rm : T(0));
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) bank_store<WT>(&Gt[idm[r]], g2[r]);
    }
  };
  // (KG = 2: the MFMA tiles this wave's K-group finishes -- wave-uniform scalar branches)
#pragma unroll
  for (int j = 0; j < 5; ++j)
    if (lg_owns_acc<KG>(kg, j)) tile_epilogue(m0 + 16 * wave, n0 + 16 * j, acc[j]);
  if (lg_owns_ax0<KG>(kg)) tile_epilogue(m0 + 64, n0 + 16 * wave, ax0);
  if (wave == 0 && lg_owns_ax1<KG>(kg)) tile_epilogue(m0 + 64, n0 + 64, ax1);
  if (EXPM || RAW) return;
  __syncthreads();   // the patches are read no more: the loss partials reuse the buffer
  lossacc = wave_sum(lossacc);
  // the LDS panels are free after the K loop (the tile routine ends with a barrier)
  double *sRed = reinterpret_cast<double *>(sAB);
  if (lane == 0) sRed[wv] = lossacc;
  __syncthreads();
  if (tid == 0) {
    double l = (sRed[0] + sRed[1]) + (sRed[2] + sRed[3]);
    if (KG == 2) l += (sRed[4] + sRed[5]) + (sRed[6] + sRed[7]);
    a.loss_part[vid] = l;
  }
  CB_STAMP_FINISH(0);
}

template <typename T, typename TG = T, bool EXPM = false, int KG = 1, bool RAW = false>   // EXPM: write P_b (cb_expm_bank) instead of loss / Gt
__global__ __launch_bounds__(LG4_THREADS * KG, 4) void k1_pt_loss_gt(K1Args<T, TG> a) {  // sixteen waves per CU
  if (a.skip && *a.skip != 0ull) return;
  __shared__ T sAB[KG * 4 * LG_KT * LG_TM];
  const int tilesN = (a.LD + LG_TN - 1) / LG_TN, tiles = tilesN * (tilesN + 1) / 2;
  const int vid = xcd_swizzle(blockIdx.x, gridDim.x);
  const int b = vid / tiles;
  k1_tile<T, TG, EXPM, false, KG, RAW>(a, b, vid - b * tiles, sAB, threadIdx.x);
}

// ------------------------------------------------------------------ K2
template <typename T>
struct K2Args {
  int LD;
  const T *Gt;  // [B][LD][LD]
  const T *U;   // [LD][LD]
  T *Tm;        // [B][LD][LD]
  const unsigned long long *skip = nullptr;   // as K1Args::skip
};

// one (bucket b, tile) of K2; sAB as in k1_tile
template <typename T, bool WT, int KG = 1>
__device__ __forceinline__ void k2_tile(const K2Args<T> &a, int b, int tile, T *sAB, int tid, const BankHooks &hooks = BankHooks{}) {
  T *sA = sAB, *sB = sAB + 2 * LG_KT * LG_TM;
  typedef typename Mfma<T>::acc_t acc_t;
  const int tilesN = (a.LD + LG_TN - 1) / LG_TN;
  const int tm = tile / tilesN, tn = tile - tm * tilesN;
  const int m0 = tm * LG_TM, n0 = tn * LG_TN;
  const size_t boff = (size_t)b * a.LD * a.LD;
  GemmOperands<T> g{a.Gt + boff, a.U, a.LD, a.LD, a.LD, a.LD, a.LD, nullptr};
  acc_t acc[5], ax0, ax1;
  CB_STAMP_BEGIN(b * tilesN * tilesN + tile);
  lg4_gemm_tile<T, false, true, KG>(g, m0, n0, sA, sB, acc, ax0, ax1, tid, hooks);
  CB_STAMP_END(1);
  T *__restrict__ Tm = a.Tm + boff;
  lg_for_each<T, KG>(m0, n0, acc, ax0, ax1, [&](int row, int col, T v) {
    if (row < a.LD && col < a.LD) bank_store<WT>(&Tm[(size_t)row * a.LD + col], v);
  }, tid);
  CB_STAMP_FINISH(1);
}

template <typename T, int KG = 1>
__global__ __launch_bounds__(LG4_THREADS * KG, 4) void k2_t_eq_g_u(K2Args<T> a) {
  if (a.skip && *a.skip != 0ull) return;
  __shared__ T sAB[KG * 4 * LG_KT * LG_TM];
  const int tilesN = (a.LD + LG_TN - 1) / LG_TN, tiles = tilesN * tilesN;
  const int vid = xcd_swizzle(blockIdx.x, gridDim.x);
  const int b = vid / tiles;
  k2_tile<T, false, KG>(a, b, vid - b * tiles, sAB, threadIdx.x);
}

// ------------------------------------------------------------------ K3
// Wt_b[c][a] = Phi_b[c][a] * sum_i T_b[i][c] U[i][a], one (bucket, tile) per workgroup,
// written over Gt_b (dead once K2 has produced T_b); k3_reduce then sums the buckets
// in a fixed order.  (A version that kept the running sum over a chunk of buckets in
// registers needed 2 accumulator sets: 256 VGPRs, one workgroup per CU.)
template <typename T>
struct K3Args {
  int LD, B;
  const T *Tm;           // [B][LD][LD]
  const T *U;            // [LD][LD]
  const double *t;       // [B]
  const double *lam;     // [LD]
  const double *E;       // [B][LD] exp(t lam)
  const double *H;       // [B][LD] exp(t lam / 2)
  T *W;                  // [B][LD][LD] out (aliases the Gt buffer)
  int sym;               // counts symmetric => Gt_b, hence W_b, symmetric: upper-triangular tiles only
  const unsigned long long *skip = nullptr;   // as K1Args::skip
};

// one (bucket b, tile) of K3; sAB as in k1_tile
template <typename T, int KG = 1>
__device__ __forceinline__ void k3_tile(const K3Args<T> &a, int b, int tile, T *sAB, int tid, const BankHooks &hooks = BankHooks{}) {
  T *sA = sAB, *sB = sAB + 2 * LG_KT * LG_TM;
  typedef typename Mfma<T>::acc_t acc_t;
  const int tilesN = (a.LD + LG_TN - 1) / LG_TN;
  int tm, tn;
  if (a.sym) {
    tm = 0;
    while (tile >= tilesN - tm) {
      tile -= tilesN - tm;
      ++tm;
    }
    tn = tm + tile;
  } else {
    tm = tile / tilesN;
    tn = tile - tm * tilesN;
  }
  const int m0 = tm * LG_TM, n0 = tn * LG_TN;
  const size_t boff = (size_t)b * a.LD * a.LD;
  GemmOperands<T> g{a.Tm + boff, a.U, a.LD, a.LD, a.LD, a.LD, a.LD, nullptr};
  acc_t acc[5], ax0, ax1;
  CB_STAMP_BEGIN(b * tilesN * tilesN + tm * tilesN + tn);
  lg4_gemm_tile<T, false, true, KG>(g, m0, n0, sA, sB, acc, ax0, ax1, tid, hooks);
  CB_STAMP_END(2);
  const double tb = a.t[b];
  const double *__restrict__ Eb = a.E + (size_t)b * a.LD, *__restrict__ Hb = a.H + (size_t)b * a.LD;
  const double *__restrict__ lam = a.lam;
  T *__restrict__ W = a.W + boff;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), kg = KG == 1 ? 0 : wv >> 2, wave = wv & 3;
  const int lane = tid & 63, lo = lane & 15, hi = lane >> 4;
  // per MFMA tile: the spectral tables of its rows and column first (all loads of the tile in flight
  // together; element by element they compiled to load -> wait -> store chains), then Phi, then the stores.
  // (the divided difference is evaluated in float64 in both widths: its cancellation-free form needs it)
  auto tile_epilogue = [&](int rbase, int cbase, const acc_t &v) {   // wave-uniform tile origin
    if (rbase >= a.LD || cbase >= a.LD) return;
    const int col = cbase + lo;
    const double lc = lam[col], ec = Eb[col], hc = Hb[col];
    double lr[4], er[4], hr[4];
    int row[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      row[r] = rbase + Mfma<T>::row(hi, r);
      lr[r] = lam[row[r]];
      er[r] = Eb[row[r]];
      hr[r] = Hb[row[r]];
    }
    T w[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) w[r] = (T)((double)v[r] * divdiff_fast(tb, lr[r], lc, er[r], ec, hr[r], hc));
#pragma unroll
    for (int r = 0; r < 4; ++r) W[(size_t)row[r] * a.LD + col] = w[r];   // (symmetric case: k3_reduce mirrors the sum, not every bucket)
  };
#pragma unroll
  for (int j = 0; j < 5; ++j)
    if (lg_owns_acc<KG>(kg, j)) tile_epilogue(m0 + 16 * wave, n0 + 16 * j, acc[j]);
  if (lg_owns_ax0<KG>(kg)) tile_epilogue(m0 + 64, n0 + 16 * wave, ax0);
  if (wave == 0 && lg_owns_ax1<KG>(kg)) tile_epilogue(m0 + 64, n0 + 64, ax1);
  CB_STAMP_FINISH(2);
}

template <typename T, int KG = 1>
__global__ __launch_bounds__(LG4_THREADS * KG, 4) void k3_w_phi(K3Args<T> a) {
  if (a.skip && *a.skip != 0ull) return;
  __shared__ T sAB[KG * 4 * LG_KT * LG_TM];
  const int tilesN = (a.LD + LG_TN - 1) / LG_TN;
  const int tiles = a.sym ? tilesN * (tilesN + 1) / 2 : tilesN * tilesN;
  const int vid = xcd_swizzle(blockIdx.x, gridDim.x);
  const int b = vid / tiles;
  k3_tile<T, KG>(a, b, vid - b * tiles, sAB, threadIdx.x);
}

// ------------------------------------------------------------------ K1 -> K2 -> K3 in ONE launch
// Measured with in-kernel stamps (-DCB_CLOCK_STAMP, profiles/tools/clock_probe.py + stamp_timeline.py,
// profiles/r04_bank_tile_timeline_*.txt): as three launches each of K1 / K2 / K3 spends its last ~90 us draining -- the four
// workgroups of a CU finish one after the other, and a workgroup that is alone on its CU keeps the f64 matrix pipe 25-35 %
// busy (its K loop is a latency chain: 70 us for 17 us of MFMAs) -- which is 30 % of K1; the chip's clock is NOT the limit
// (2.36-2.38 GHz inside the K loops).  Here the three products are ONE persistent launch of 4 workgroups per CU that draw
// (stage, bucket, tile) tickets, so that K2's tiles fill K1's drain and K3's fill K2's; only K3's own drain is left.
//  * Tickets: one queue per XCD (workgroups go round-robin over the XCDs: blockIdx.x % 8 is the home queue -- checked with
//    XCC_ID stamps: every queue was served by one XCD), holding the three stages of that XCD's share of the buckets in
//    order: all its K1 tiles, then its K2 tiles, then its K3 tiles, bucket by bucket, so that the tiles that share an operand
//    block run side by side on one L2 (what xcd_swizzle does for the separate kernels).  A workgroup whose queue is empty
//    draws from the next XCD's.
//  * Dependencies: K2 of bucket b reads ALL of Gt_b, K3 of bucket b all of T_b: one counter per (stage, bucket), bumped by
//    every finished tile; a K2 / K3 ticket waits until its bucket's counter is complete.  A queue hands out every K1 ticket
//    before its first K2 ticket, K1 tickets wait for nothing and a drawn ticket belongs to a RUNNING workgroup (the ticket
//    drawn ahead, below, is held by a workgroup whose current tile has all its inputs): no deadlock whatever the number of
//    resident workgroups.  In practice nobody waits: bucket b's K1 tiles are drawn ~one stage (150 us) before its K2 tiles.
//  * Visibility: the L2s of the XCDs are not coherent with each other inside a launch.  An agent-scope release (write back
//    the L2) per tile costs +55 us on K2, an acquire (invalidate it: U is gone) +48 us (measured with a fence in K2); so Gt
//    and T are written THROUGH to memory instead (agent-scope stores: +0 us), the producer waits for its stores (vmcnt)
//    before it bumps the counter, and the consumer needs no invalidate because no cache can hold an older copy of those
//    lines: inside this launch they are read only after they were written, and the launch boundary invalidated whatever
//    the previous epoch left.  (The counters are agent-scope atomics.)  W_b (K3's output, over Gt_b) is read by the NEXT
//    launch (k3_reduce): plain stores.  tests/test_gpu_s400_full.py compares this launch with the three separate ones bit
//    for bit (CB_BANK_UNFUSED=1).
struct BankQueueArgs {
  unsigned int *queue;   // [LG_NQ] tickets drawn per queue | [B] finished K1 tiles | [B] finished K2 tiles | [LG_NQ][claims] claim
                         // flags of the reserved first tickets; all set by lg_tables in front of the launch
  int B, tiles1, tiles2, tiles3;
  int claims;            // reserved tickets per queue (>= the workgroups of a launch that call one queue home)
  int test_no_claim;     // test hook (CB_BANK_TEST_NO_CLAIM): no workgroup takes its reserved ticket -- as if none of their owners
                         // were resident; they are then run by the workgroups that wait for them (the `help` path)
};

// Ticket `idx` of a queue that owns `nb` buckets -> (stage 0 / 1 / 2, local bucket, tile): the queue's K1 tiles bucket by
// bucket, then its K2 tiles, then its K3 tiles.  (Measured and dropped: a software pipeline over the buckets -- round r = the K1
// tiles of bucket r, the K2 tiles of bucket r - lag, the K3 tiles of bucket r - 2 lag, mixed in the proportion of their
// counts so that K1's long epilogues run beside K2 tiles -- 0.84 / 0.81 / 0.79 ms at lag 4 / 6 / 8 against 0.74 ms: a K2
// ticket ~250 tickets behind its bucket's K1 tickets still finds them in flight and waits.)
struct BankTicket { int stage, b, tile; };
__device__ __forceinline__ BankTicket bank_decode(const BankQueueArgs &a, int nb, int idx) {
  const int t1 = a.tiles1, t2 = a.tiles2, t3 = a.tiles3;
  BankTicket k;
  if (idx < nb * t1) k.stage = 0, k.b = idx / t1, k.tile = idx - k.b * t1;
  else if (idx < nb * (t1 + t2)) k.stage = 1, k.b = (idx - nb * t1) / t2, k.tile = idx - nb * t1 - k.b * t2;
  else k.stage = 2, k.b = (idx - nb * (t1 + t2)) / t3, k.tile = idx - nb * (t1 + t2) - k.b * t3;
  return k;
}
template <typename T1, typename TG>
struct K123Args {
  K1Args<T1, TG> k1;
  K2Args<TG> k2;
  K3Args<TG> k3;
  BankQueueArgs q;
};

#ifndef LG_NQ
#define LG_NQ 8   // ticket queues (a multiple of the 8 XCDs: queue q is served by the workgroups with blockIdx.x % LG_NQ == q, all on XCD q % 8)
#endif

// a wave-uniform copy (in SGPRs) of an argument block read from device memory
template <typename S>
__device__ __forceinline__ S uniform_copy(const S *p) {
  static_assert(sizeof(S) % 4 == 0, "dwords");
  S out;
  const unsigned int *src = reinterpret_cast<const unsigned int *>(p);
  unsigned int *dst = reinterpret_cast<unsigned int *>(&out);
#pragma unroll
  for (size_t i = 0; i < sizeof(S) / 4; ++i) dst[i] = (unsigned int)__builtin_amdgcn_readfirstlane((int)src[i]);
  return out;
}

// the same with the loads addressed to constant memory (scalar loads where the compiler sees a uniform address; the block
// is not written inside the launch that reads it)
template <typename S>
__device__ __forceinline__ S const_copy(const S *p) {
  static_assert(sizeof(S) % 4 == 0, "dwords");
  S out;
  const __attribute__((address_space(4))) unsigned int *src = (const __attribute__((address_space(4))) unsigned int *)p;
  unsigned int *dst = reinterpret_cast<unsigned int *>(&out);
#pragma unroll
  for (size_t i = 0; i < sizeof(S) / 4; ++i) dst[i] = (unsigned int)__builtin_amdgcn_readfirstlane((int)src[i]);
  return out;
}

// A pointer read from memory is a GENERIC pointer to the compiler (flat_load / flat_store: both wait counters, no
// global-only addressing modes): say that it points to global memory, as it knows of a pointer kernel argument.
template <typename P>
__device__ __forceinline__ P *as_global(P *p) {
  // (a plain cast to address_space(1) and back is folded away before the address-space inference runs; through an empty
  // asm the global pointer is opaque, and being an "s" operand it is in SGPRs)
  __attribute__((address_space(1))) P *g = (__attribute__((address_space(1))) P *)p;
  asm("" : "+s"(g));
  return (P *)g;
}
template <typename T, typename TG>
__device__ __forceinline__ void globalize(K1Args<T, TG> &a) {
  a.Ut = as_global(a.Ut); a.A = as_global(a.A); a.t = as_global(a.t); a.F = as_global(a.F); a.sigma = as_global(a.sigma);
  a.Ct = as_global(a.Ct); a.Gt = as_global(a.Gt); a.loss_part = as_global(a.loss_part); a.dsq = as_global(a.dsq);
  a.P = as_global(a.P); a.skip = as_global(a.skip);
}
template <typename T>
__device__ __forceinline__ void globalize(K2Args<T> &a) {
  a.Gt = as_global(a.Gt); a.U = as_global(a.U); a.Tm = as_global(a.Tm); a.skip = as_global(a.skip);
}
template <typename T>
__device__ __forceinline__ void globalize(K3Args<T> &a) {
  a.Tm = as_global(a.Tm); a.U = as_global(a.U); a.t = as_global(a.t); a.lam = as_global(a.lam); a.E = as_global(a.E);
  a.H = as_global(a.H); a.W = as_global(a.W); a.skip = as_global(a.skip);
}

// The argument block lives in DEVICE memory (lg_tables, the launch in front, copies it there from its own kernel arguments):
// as kernel arguments the ~40 pointers and sizes of the three stages stayed in SGPRs across the ticket loop (106 + 111
// spilled, and 57 spilled VGPRs in their wake); read where a stage needs them (uniform_copy: loads + v_readfirstlane, so that
// they are SGPRs again inside the stage) they cost a few loads per tile.
template <typename T1, typename TG, int KG = 1>
__global__ __launch_bounds__(LG4_THREADS * KG, 4) void k123_bank(const K123Args<T1, TG> *ap) {
  BankQueueArgs a = uniform_copy(&ap->q);
  a.queue = as_global(a.queue);
  {
    const unsigned long long *skip = ap->k1.skip;
    if (skip && *skip != 0ull) return;
  }
  constexpr size_t ELT = sizeof(T1) > sizeof(TG) ? sizeof(T1) : sizeof(TG);
  __shared__ __attribute__((aligned(16))) unsigned char smem[KG * 4 * LG_KT * LG_TM * ELT];   // 40 KB x KG in float64: sixteen waves per CU
  int *s_ticket = reinterpret_cast<int *>(smem);   // (queue, ticket) of this round, in the panel buffer: every thread has read
                                                   // them before the tile routine's first barrier, which precedes its first store
  unsigned int *tick = a.queue, *done1 = a.queue + LG_NQ, *done2 = done1 + a.B;
  const int per_bucket = a.tiles1 + a.tiles2 + a.tiles3;
  // Workgroups go round-robin over the XCDs (xcd_swizzle's assumption): blockIdx.x % 8 names the home queue.
  const int home = blockIdx.x & (LG_NQ - 1);
  auto queue_len = [&](int q) { return (a.B * (q + 1) / LG_NQ - a.B * q / LG_NQ) * per_bucket; };
  // Who may run a ticket.  A ticket DRAWN from a queue's counter belongs to a running workgroup, so whatever waits for it
  // will see it finished, with any number of resident workgroups.  Only the 128 workgroups of a queue all drawing their first
  // ticket in the first microsecond of the launch serialise on the counter (15-40 us before the last one starts).  So the
  // first n = min(home workgroups, K1 tickets of the queue) tickets -- K1 tiles, which wait for nothing -- are RESERVED:
  // workgroup blockIdx.x takes ticket blockIdx.x / 8 of its home queue by setting that ticket's claim flag (one uncontended
  // atomic), the counter starts at n.  A reserved ticket whose workgroup is not resident is not lost: a workgroup that waits
  // for a bucket's K1 tiles claims the bucket's unclaimed reserved tickets itself and runs them first (`help` below).
  // (Round 4 first handed out the first two tickets by index without claim flags: two such launches on one GPU -- two ranks
  // of the test hook sharing it -- then waited for tiles owned by workgroups that were not resident, each launch holding the
  // slots the other needed, until the scheduler's preemption timer let them through: 7.9 s per epoch.)
  // Between two tiles of a workgroup lay ~9 us of dependent round trips (the announcement of the finished tile behind a
  // wait for its stores, the draw, the dependency word, the argument block): 16 % of the launch.  So the NEXT ticket is drawn
  // from inside the current tile's K loop and the finished tile is announced from inside the next tile's (BankHooks), and the
  // argument blocks come through the scalar cache: 3-4 us are left.  (Drawing TWO tickets ahead, so that the dependency word
  // is in flight during a tile as well, was slower, 0.81 against 0.74 ms: a ticket then waits up to two tile times in its
  // holder's pipeline, and the K2 tickets of a queue's last buckets find their K1 tiles not even started.)
  unsigned int *claim = done2 + a.B;
  auto reserved = [&](int q) {   // reserved first tickets of queue q (lg_tables starts the counter there)
    const int homes = ((int)gridDim.x - q + LG_NQ - 1) / LG_NQ, k1 = (a.B * (q + 1) / LG_NQ - a.B * q / LG_NQ) * a.tiles1;
    return min(min(homes, k1), a.claims);
  };
  int *drawn_lds = reinterpret_cast<int *>(smem + sizeof(smem) - 16);   // the draw made inside the K loop (no epilogue uses
                                                                        // the end of the panel buffer)
  int nq = home, ni = -1;   // thread 0: the ticket drawn for this round (queue, raw index; -1: none yet)
  int own_stage = -1, own_b = 0, own_tile = 0, own_q = 0;   // a ticket put aside while this workgroup helps with its inputs
  unsigned int *pending = nullptr;                 // the finished tile that is still to be announced
  bool first = true;
  for (;;) {
    __syncthreads();   // the previous ticket's last readers of smem / s_ticket are done
    if (threadIdx.x == 0) {
      int stage = -1, b = 0, tile = 0, ready = 1, q = nq;
      if (own_stage >= 0) {   // back from helping: the ticket put aside
        stage = own_stage, b = own_b, tile = own_tile, q = own_q;
      } else {
        int idx = ni;
        if (first) {   // the reserved first ticket, if nobody has run it for us
          const int i0 = (int)(blockIdx.x / LG_NQ);
          if (i0 < reserved(home) && !a.test_no_claim &&
              __hip_atomic_exchange(claim + home * a.claims + i0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
            idx = i0;
        }
        if (idx < 0 && q == home) idx = (int)__hip_atomic_fetch_add(tick + q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (idx >= queue_len(q)) idx = -1;
        for (int tries = 0; idx < 0 && tries < LG_NQ; ++tries) {   // own queue empty: the next XCD's
          q = (q + 1) & (LG_NQ - 1);
          const int len = queue_len(q);
          if ((int)__hip_atomic_load(tick + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= len) continue;
          const int i = (int)__hip_atomic_fetch_add(tick + q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (i < len) idx = i;
        }
        if (idx >= 0) {
          const int b0 = a.B * q / LG_NQ, nb = a.B * (q + 1) / LG_NQ - b0;
          const BankTicket k = bank_decode(a, nb, idx);
          stage = k.stage, b = b0 + k.b, tile = k.tile;
        }
      }
      if (stage >= 0) {
        const unsigned int *dep = stage == 1 ? done1 + b : stage == 2 ? done2 + b : nullptr;
        const unsigned int need = (unsigned int)(stage == 1 ? a.tiles1 : a.tiles2);
        if (dep && __hip_atomic_load(dep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) ready = 0;
      }
      s_ticket[0] = stage;
      s_ticket[1] = b;
      s_ticket[2] = tile;
      s_ticket[3] = ready;
      s_ticket[4] = q;
    }
    first = false;
    __syncthreads();
    int stage = s_ticket[0], b = s_ticket[1], tile = s_ticket[2];
    const int q = s_ticket[4];
    const bool ready = s_ticket[3] != 0;
    own_stage = -1;
    if (stage < 0 || !ready) {
      // nothing left, or the ticket's inputs are not complete (rare: see above): announce the finished tile NOW -- the
      // inputs waited for may include it -- and wait; a K2 ticket that waits looks for unclaimed reserved K1 tickets of its
      // bucket meanwhile and, when it finds one, puts itself aside and runs that first
      if (pending) {
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's stores have been performed
        __syncthreads();                      // ... and every wave's
        if (threadIdx.x == 0) __hip_atomic_fetch_add(pending, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pending = nullptr;
      }
      if (stage < 0) return;
      __syncthreads();   // (s_ticket has been read by every thread)
      if (threadIdx.x == 0) {
        const unsigned int *dep = stage == 1 ? done1 + b : done2 + b;
        const unsigned int need = (unsigned int)(stage == 1 ? a.tiles1 : a.tiles2);
        const int bl = b - a.B * q / LG_NQ, nres = reserved(q);
        int help = -1;
        for (unsigned int spins = 0; __hip_atomic_load(dep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need; ++spins) {
          __builtin_amdgcn_s_sleep(4);
          if (stage != 1 || (spins & 63u) != 63u) continue;
          for (int i = bl * a.tiles1; i < (bl + 1) * a.tiles1 && i < nres && help < 0; ++i)
            if (__hip_atomic_load(claim + q * a.claims + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u &&
                __hip_atomic_exchange(claim + q * a.claims + i, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
              help = i - bl * a.tiles1;
          if (help >= 0) break;
        }
        s_ticket[5] = help;
      }
      __syncthreads();
      const int help = s_ticket[5];
      if (help >= 0) {   // this round: the reserved K1 tile nobody had started; next round: the ticket put aside
        own_stage = stage, own_b = b, own_tile = tile, own_q = q;
        stage = 0, tile = help;
      }
    }
    // an opaque copy of threadIdx.x per ticket: otherwise the lane offsets, panel addresses ... of ALL three stages are
    // computed once in front of the loop and stay live through every stage (50 spilled VGPRs in K1's epilogue)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const K123Args<T1, TG> *cap = ap;   // (the same for the argument block: its scalar loads would all be hoisted)
    asm volatile("" : "+s"(cap));
    BankHooks hooks;
    hooks.deferred = pending;
    if (own_stage < 0) {   // (a helper already holds its next ticket)
      hooks.draw = tick + q;
      hooks.drawn_lds = drawn_lds;
    }
    unsigned int *signal = nullptr;
    if (stage == 0) {
      K1Args<T1, TG> k1 = const_copy(&cap->k1);
      globalize(k1);
      k1_tile<T1, TG, false, true, KG>(k1, b, tile, reinterpret_cast<T1 *>(smem), tid, hooks);
      signal = done1 + b;
    } else if (stage == 1) {
      K2Args<TG> k2 = const_copy(&cap->k2);
      globalize(k2);
      k2_tile<TG, true, KG>(k2, b, tile, reinterpret_cast<TG *>(smem), tid, hooks);
      signal = done2 + b;
    } else {
      K3Args<TG> k3 = const_copy(&cap->k3);
      globalize(k3);
      k3_tile<TG, KG>(k3, b, tile, reinterpret_cast<TG *>(smem), tid, hooks);
    }
    pending = signal;
    if (own_stage < 0 && threadIdx.x == 0) {
      nq = q;
      ni = *drawn_lds;   // (thread 0 wrote it itself, behind the K loop)
    }
  }
}

// Mt = sum over chunks (fixed order => bitwise reproducible), always accumulated in float64.  sym
// (LD > 0): only the 80x80 tiles on or above the diagonal were written by k3_w_phi; sum those and
// mirror the SUM into the lower tiles (40 % less to read, and no mirrored stores per bucket).
template <typename T>
__device__ __forceinline__ void k3_reduce_body(const T *part, int nchunks, size_t n, double *out, int LD) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int row = 0, col = 0;
  if (LD > 0) {
    row = (int)(i / LD);
    col = (int)(i - (size_t)row * LD);
    if (row / LG_TM > col / LG_TN) return;
  }
  double s0 = 0.0, s1 = 0.0;   // two interleaved partial sums (even / odd chunks); eight loads in flight per thread
  int c = 0;
  for (; c + 7 < nchunks; c += 8) {
    T v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(c + u) * n + i];
#pragma unroll
    for (int u = 0; u < 8; u += 2) {
      s0 += (double)v[u];
      s1 += (double)v[u + 1];
    }
  }
  for (; c + 1 < nchunks; c += 2) {
    s0 += (double)part[(size_t)c * n + i];
    s1 += (double)part[(size_t)(c + 1) * n + i];
  }
  if (c < nchunks) s0 += (double)part[(size_t)c * n + i];
  const double s = s0 + s1;
  out[i] = s;
  if (LD > 0 && row / LG_TM < col / LG_TN) out[(size_t)col * LD + row] = s;
}

template <typename T>
__global__ void k3_reduce(const T *part, int nchunks, size_t n, double *out, int LD = 0) {
  k3_reduce_body<T>(part, nchunks, n, out, LD);
}

// loss = (sum of partials - direct term) * inv_n, one workgroup, fixed order
__device__ __forceinline__ void lg_finish_loss_body(const double *part, int nparts, int S, const double *dsq,
                                                    const double *dirsum, double inv_n, double *loss) {
  __shared__ double s[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) acc += part[i];
  double dir = 0.0;
  for (int k = threadIdx.x; k < S; k += 256) dir = fma(log(dsq[k]), dirsum[k], dir);
  s[threadIdx.x] = acc - dir;
  __syncthreads();
  for (int st = 128; st >= 1; st >>= 1) {
    if ((int)threadIdx.x < st) s[threadIdx.x] += s[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) *loss = s[0] * inv_n;
}

// the bucket sum and the loss in ONE launch (they do not depend on each other; behind the fused bank launch): the last
// workgroup takes the loss
struct LossArgs {
  const double *part;
  int nparts, S;
  const double *dsq, *dirsum;
  double inv_n;
  double *loss;
  const unsigned long long *skip = nullptr;   // as K1Args::skip: the bank launch in front returned at once (stalled solve),
                                              // h->loss / h->Mt keep the previous evaluation's values until the retry
};
template <typename T>
__global__ __launch_bounds__(256) void k3_reduce_loss(const T *part, int nchunks, size_t n, double *out, int LD, LossArgs l) {
  if (l.skip && *l.skip != 0ull) return;
  if (blockIdx.x + 1 == gridDim.x) lg_finish_loss_body(l.part, l.nparts, l.S, l.dsq, l.dirsum, l.inv_n, l.loss);
  else k3_reduce_body<T>(part, nchunks, n, out, LD);
}

// ------------------------------------------------------------------ the bucket sum WITHOUT the third product (round 5)
// M = sum_b W_b o Phi_b with W_b = U^T G~_b U needs one S^3 product per bucket (K3) only because Phi_b sits between the
// product and the sum.  But Phi_b,ij = (E_bi - E_bj) / (lam_i - lam_j), E_b = exp(t_b lam), is a difference of ONE-SIDED
// scalings, and for symmetric counts (G~_b = G~_b^T, so U^T G~_b = T_b^T with T_b = G~_b U, K2's output):
//     sum_b diag(E_b) W_b = [sum_b diag(E_b) T_b^T] U = Y^T U =: Le,      Y = sum_b T_b diag(E_b)      (no product per bucket)
//     M_ij = (Le_ij - Le_ji) / (lam_i - lam_j)
// i.e. the buckets are summed BEFORE the product: one streaming pass over T (ky_reduce_loss, the successor of k3_reduce_loss)
// and ONE S^3 product instead of B.  The difference Le - Le^T cancels for close eigenvalues (error ~ eps / (t |dlam|)), so
// pairs with |lam_i - lam_j| < delta take the series of the same function in (dlam)^2, whose terms are one-sided as well:
//     Phi_ij = t e^{t (lam_i + lam_j) / 2} sinhc(z) = t (E_i + E_j) / 2 * tanh(z) / z,   z = t dlam / 2
//     tanh(z) / z = sum_k a_k z^2k      =>      M_ij = sum_k a_k (dlam^2 / 4)^k (Lt_k,ij + Lt_k,ji) / 2,
//     Lt_k = sum_b diag(t_b^{2k+1} E_b) W_b = Y_k^T U,   Y_k = sum_b T_b diag(t_b^{2k+1} E_b)
// (the diagonal is the k = 0 term).  delta = 0.2 / max_b t_b keeps |z| <= 0.1 for every bucket: CB_PHI_TERMS = 6 terms leave
// 4e-15.  Measured against an 80-bit evaluation on the bench bank (profiles/tools/accumulated_phi_model.py): dL/dA within
// 6e-14 .. 3e-12 (the per-bucket form: 3e-16; the parity bar is 1e-10).  MFMA tiles per bucket: 40 (K1 15 + K2 25) instead
// of 55.  Symmetric counts only (cherry counts are, by construction); other banks keep K3.
#define CB_PHI_TERMS 6
struct YArgs {
  int LD, B;
  const double *t;     // [B]
  const double *E;     // [B][LD] exp(t lam)
  double *Y;           // [1 + CB_PHI_TERMS][LD][LD]: Y (scaling E_b), then Y_k (scaling t_b^{2k+1} E_b)
};
template <typename T>
__global__ __launch_bounds__(256) void ky_reduce_loss(const T *__restrict__ Tm, YArgs y, LossArgs l) {
  if (l.skip && *l.skip != 0ull) return;
  if (blockIdx.x + 1 == gridDim.x) {   // the last workgroup sums the loss partials (as k3_reduce_loss)
    lg_finish_loss_body(l.part, l.nparts, l.S, l.dsq, l.dirsum, l.inv_n, l.loss);
    return;
  }
  const size_t LL = (size_t)y.LD * y.LD, i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= LL) return;
  const int col = (int)(i % y.LD);
  double acc[1 + CB_PHI_TERMS];
#pragma unroll
  for (int k = 0; k <= CB_PHI_TERMS; ++k) acc[k] = 0.0;
  // buckets in their stored order, four in flight: the sums do not depend on the launch geometry
  int b = 0;
  for (; b + 3 < y.B; b += 4) {
    double v[4], e[4], tb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      v[u] = (double)Tm[(size_t)(b + u) * LL + i];
      e[u] = y.E[(size_t)(b + u) * y.LD + col];
      tb[u] = y.t[b + u];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const double ve = v[u] * e[u], t2 = tb[u] * tb[u];
      acc[0] += ve;
      double c = ve * tb[u];
#pragma unroll
      for (int k = 1; k <= CB_PHI_TERMS; ++k) {
        acc[k] += c;
        c *= t2;
      }
    }
  }
  for (; b < y.B; ++b) {
    const double ve = (double)Tm[(size_t)b * LL + i] * y.E[(size_t)b * y.LD + col], tb = y.t[b], t2 = tb * tb;
    acc[0] += ve;
    double c = ve * tb;
#pragma unroll
    for (int k = 1; k <= CB_PHI_TERMS; ++k) {
      acc[k] += c;
      c *= t2;
    }
  }
#pragma unroll
  for (int k = 0; k <= CB_PHI_TERMS; ++k) y.Y[(size_t)k * LL + i] = acc[k];
}

// M from Le = L[0] and Lt_k = L[1 + k] (each [LD][LD]); see above.  M is symmetric: written whole (K4a reads it as M^T).
template <int NT = CB_PHI_TERMS>   // (a template: the header is included by two translation units)
__global__ __launch_bounds__(256) void kphi_combine(int LD, const double *__restrict__ L, const double *__restrict__ lam,
                                                    double delta, double *__restrict__ Mt, const unsigned long long *skip) {
  if (skip && *skip != 0ull) return;
  const size_t LL = (size_t)LD * LD, idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= LL) return;
  const int i = (int)(idx / LD), j = (int)(idx - (size_t)i * LD);
  const size_t tr = (size_t)j * LD + i;
  const double dl = lam[i] - lam[j];
  double m;
  if (fabs(dl) < delta) {
    // tanh(z) / z = 1 - z^2/3 + 2 z^4/15 - 17 z^6/315 + 62 z^8/2835 - 1382 z^10/155925
    static_assert(NT == 6, "six series coefficients below");
    const double ak[NT] = {1.0, -1.0 / 3.0, 2.0 / 15.0, -17.0 / 315.0, 62.0 / 2835.0, -1382.0 / 155925.0};
    const double w = 0.25 * dl * dl;
    m = 0.0;
#pragma unroll
    for (int k = NT - 1; k >= 0; --k) m = fma(m, w, ak[k] * 0.5 * (L[(size_t)(1 + k) * LL + idx] + L[(size_t)(1 + k) * LL + tr]));
  } else {
    m = (L[idx] - L[tr]) / dl;
  }
  Mt[idx] = m;
}

// ------------------------------------------------------------------ K4 (plain / dQ epilogue)
struct K4Args {
  int S, LD;
  const double *Aop, *Bop;  // [LD][LD] each
  double *out;              // [LD][LD] or dQ [S][S]
  const double *dsq;        // non-null => out = dQ[i][j] = d_i * acc / d_j, unpadded S x S
  const double *sub;        // non-null => out = acc - (*sub_scale) * sub[row][col]
  const double *sub_scale;
  double *outT = nullptr;   // sg_gemm only: also write the transpose of the result
  double *diag = nullptr;   // sg_gemm only: also write the diagonal of the result
  // sg_gemm only: operands chosen ON THE DEVICE -- when *sel != 0 the non-null alternatives replace
  // Aop / Bop.  Lets the host enqueue the products of a first-order sweep before it knows which X
  // (all pairs or far pairs only) lgx_build decided on.
  const unsigned long long *sel = nullptr;
  const double *Aalt = nullptr, *Balt = nullptr;
  const unsigned long long *skip = nullptr;   // as K1Args::skip (K4 behind a bank launch that returned at once)
  size_t ystride = 0;       // sg_gemm only, != 0: gridDim.y independent products of the same kind from ONE argument block --
                            // product y reads Aop + y * ystride and writes out + y * ystride (the Lt_k = Y_k^T U of the bucket sum)
};

// Single-matrix products (K4a, K4b, the warm-start G0 = A' U_prev, the first-order eigen
// correction): one LD^3 GEMM is only (LD/80)^2 = 25 of the 80x80 tiles, i.e. 25 of 256 CUs and
// a 25-step serial K loop per tile (43 us at LD = 400).  Here a workgroup owns a 16 x 80 strip
// and its four waves split K (k-step s goes to wave s mod 4); operand fragments are read
// straight from L2 in MFMA layout (no LDS staging: the matrices are 1.3 MB), the four partial
// strips are summed through LDS in a fixed order.  (LD/16) x ceil(LD/80) = 125 workgroups.
// Same argument block and epilogues as k4_gemm, plus:
//   ns == 1 :  out = (row == col) + sub[row][col] - acc / 2      (R = I + X + X^2/2, X^2 = -X^T X)
//   ns == 2 :  out = alpha * acc + beta * sub[row][col]          (polynomial / Newton-Schulz steps of the
//                                                                 first-order sweeps, jacobi_block.hip.h)
// NJ = 16-column tiles per workgroup (strip 16 x 16 NJ): fewer -> more workgroups and shorter waves
// (these single 400^3 products are latency bound, not MFMA bound).
// gridDim.y == 2: two independent products in one launch (blockIdx.y == 1 takes `a2`): these launches
// are latency bound (2.6 us of launch floor + ~7 us of load -> MFMA -> reduce chain on half the CUs), so
// a pair costs what one does -- X^3 and X^4 of the first-order sweeps.
template <int NW, int UU, int NJ = 5>
__global__ __launch_bounds__(NW * 64) void sg_gemm(K4Args a1, K4Args a2, int ns, double alpha, double beta) {
  __shared__ double sRed[NW / 2 > 4 ? NW / 2 : 4][NJ][256];
  const K4Args &a = (blockIdx.y && a1.ystride == 0) ? a2 : a1;
  if (a.skip && *a.skip != 0ull) return;
  const int LD = a.LD, tilesN = (LD + 16 * NJ - 1) / (16 * NJ);
  const int tm = blockIdx.x / tilesN, tn = blockIdx.x - tm * tilesN;
  const int m0 = tm * 16, n0 = tn * 16 * NJ;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lo = lane & 15, hi = lane >> 4;
  d4 acc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) acc[j] = d4{0.0, 0.0, 0.0, 0.0};
  int ncol[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) ncol[j] = min(n0 + 16 * j + lo, LD - 1);  // clamped: tiles past LD are discarded
  const int nsteps = LD / 4;
  const bool alt = a.sel && *a.sel != 0ull;
  // (Gram-type products -- G^T G, X^T X, P2^T P2 -- are symmetric; computing only the tiles on or above the
  // diagonal and mirroring them was measured: no gain, eigh 0.300 -> 0.310 ms; the launches are latency chains.)
  const double *Ap = (alt && a.Aalt ? a.Aalt : a.Aop) + m0 + lo + (size_t)blockIdx.y * a.ystride, *Bp = alt && a.Balt ? a.Balt : a.Bop;
  for (int s0 = wave; s0 < nsteps; s0 += UU * NW) {   // UU k-steps of this wave in flight
    double av[UU], bv[UU][NJ];
#pragma unroll
    for (int u = 0; u < UU; ++u) {
      const int s = min(s0 + NW * u, nsteps - 1);
      const size_t krow = (size_t)(4 * s + hi) * LD;
      av[u] = Ap[krow];
#pragma unroll
      for (int j = 0; j < NJ; ++j) bv[u][j] = Bp[krow + ncol[j]];
    }
#pragma unroll
    for (int u = 0; u < UU; ++u) {
      if (s0 + NW * u < nsteps) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j] = mfma_f64(av[u], bv[u][j], acc[j]);
      }
    }
  }
  // K was split over the NW waves: fold the upper waves into the lower four, then sum those
  for (int half = NW / 2; half >= 4; half >>= 1) {
    if (wave >= half && wave < 2 * half) {
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) sRed[wave - half][j][r * 64 + lane] = acc[j][r];
    }
    __syncthreads();
    if (wave < half) {
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[j][r] += sRed[wave][j][r * 64 + lane];
    }
    __syncthreads();
  }
  if (wave < 4) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) sRed[wave][j][r * 64 + lane] = acc[j][r];
  }
  __syncthreads();
  if (threadIdx.x >= 256) return;
  const int t = threadIdx.x, r = t >> 6, l = t & 63;
  const int row = m0 + (l >> 4) + 4 * r;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int col = n0 + 16 * j + (l & 15);
    if (col >= LD) continue;
    const double v = (sRed[0][j][t] + sRed[1][j][t]) + (sRed[2][j][t] + sRed[3][j][t]);
    if (a.dsq) {
      if (row < a.S && col < a.S) a.out[(size_t)row * a.S + col] = a.dsq[row] * v / a.dsq[col];
    } else {
      const size_t idx = (size_t)row * LD + col + (size_t)blockIdx.y * a.ystride;
      double o;
      if (ns == 1) o = (row == col ? 1.0 : 0.0) + a.sub[idx] - 0.5 * v;
      else if (ns == 2) o = fma(alpha, v, beta * a.sub[idx]);
      else o = a.sub ? v - (*a.sub_scale) * a.sub[idx] : v;
      a.out[idx] = o;
      if (a.outT) a.outT[(size_t)col * LD + row] = o;       // transposed copy (the squarings need R^T)
      if (a.diag && row == col) a.diag[row] = o;
    }
  }
}

#ifndef CB_BANK_FUSED_TU   // (cb_bank_fused.hip compiles k123_bank only: the non-template kernels below have one home)
// ------------------------------------------------------------------ small helpers
// A = sym(D^1/2 Q D^-1/2) into padded LD x LD, dsq = sqrt(pi) (1 on the pad)
__global__ void lg_build_A(int S, int LD, const double *Q, const double *pi, double *A,
                           double *dsq) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < LD) dsq[idx] = idx < S ? sqrt(pi[idx]) : 1.0;
  if (idx >= LD * LD) return;
  const int i = idx / LD, j = idx - i * LD;
  double v = 0.0;
  if (i < S && j < S) {
    const double di = sqrt(pi[i]), dj = sqrt(pi[j]);
    v = 0.5 * (di * Q[(size_t)i * S + j] / dj + dj * Q[(size_t)j * S + i] / di);
  }
  A[idx] = v;
}

// spectral tables F = phi2(t lam), E = exp(t lam), H = exp(t lam / 2): [B][LD]
// (+ for the fused bank kernel that follows: its ticket queues and tile counters zeroed, its argument block copied to device
// memory)
struct NoBankArgs { BankQueueArgs q; };
template <typename ARGS = NoBankArgs>
__global__ void lg_tables(int LD, int B, const double *t, const double *lam,
                          const double *sigma, double *F, double *E, double *H, ARGS bank = ARGS{}, ARGS *bank_dst = nullptr,
                          int bank_grid = 0) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (bank_dst) {
    // (queue q's counter starts behind its reserved first tickets, k123_bank: min(home workgroups, K1 tickets, claims))
    // (a grid-stride loop: a small bank has fewer table entries than queue words)
    for (int i = idx; i < LG_NQ + 2 * bank.q.B + LG_NQ * bank.q.claims; i += (int)(gridDim.x * blockDim.x)) {
      unsigned int v = 0u;
      if (i < LG_NQ) {
        const int homes = (bank_grid - i + LG_NQ - 1) / LG_NQ, k1 = (bank.q.B * (i + 1) / LG_NQ - bank.q.B * i / LG_NQ) * bank.q.tiles1;
        v = (unsigned int)min(min(homes, k1), bank.q.claims);
      }
      bank.q.queue[i] = v;
    }
    if (idx == 0) *bank_dst = bank;
  }
  if (idx >= B * LD) return;
  const int b = idx / LD, k = idx - b * LD;
  const double x = t[b] * lam[k];
  const bool split = t[b] * 2.0 * (*sigma) <= 1.0;
  F[idx] = split ? phi2(x) : exp(x);
  E[idx] = exp(x);
  H[idx] = exp(0.5 * x);
}

__global__ void lg_finish_loss(const double *part, int nparts, int S, const double *dsq,
                               const double *dirsum, double inv_n, double *loss) {
  lg_finish_loss_body(part, nparts, S, dsq, dirsum, inv_n, loss);
}

// flag[0] |= 1 when some live bucket has C_b != C_b^T
template <typename T>
__global__ void lg_sym_check(int LD, const T *Ct, int *flag) {
  const size_t boff = (size_t)blockIdx.z * LD * LD;
  const int i = blockIdx.y * blockDim.y + threadIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < LD && j < i && Ct[boff + (size_t)i * LD + j] != Ct[boff + (size_t)j * LD + i]) atomicOr(flag, 1);
}

// pad + transpose counts at create time: Ct[b][j][i] = C[b][i][j]  (rounded to the bank's element type:
// counts are multiples of 1/4 and stay exact in float32 up to 2^22)
// destination bucket z holds source bucket src[z] (live buckets only, see cb_create)
template <typename T>
__global__ void lg_transpose_pad(int S, int LD, const double *C, T *Ct, const int *src) {
  __shared__ double tile[32][33];
  const int b = src[blockIdx.z];
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  for (int r = threadIdx.y; r < 32; r += blockDim.y) {
    const int i = i0 + r, j = j0 + threadIdx.x;
    tile[r][threadIdx.x] = (i < S && j < S) ? C[(size_t)b * S * S + (size_t)i * S + j] : 0.0;
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += blockDim.y) {
    const int j = j0 + r, i = i0 + threadIdx.x;
    if (j < LD && i < LD) Ct[(size_t)blockIdx.z * LD * LD + (size_t)j * LD + i] = (T)tile[threadIdx.x][r];
  }
}

// float32 copies of the per-epoch operands of the CB_F32 bank kernels: U, U^T, A [LD][LD] and F [B][LD]
__global__ void lg_cast_f32(size_t LL, size_t BL, const double *U, const double *Ut, const double *A, const double *F,
                            float *Uf, float *Utf, float *Af, float *Ff) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < LL) {
    Uf[i] = (float)U[i];
    Utf[i] = (float)Ut[i];
    Af[i] = (float)A[i];
  }
  if (i < BL) Ff[i] = (float)F[i];
}
#endif  // CB_BANK_FUSED_TU
