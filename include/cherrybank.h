/*
 * cherrybank.h -- C ABI of libcherrybank.so (HIP, gfx950 / MI355X).
 *
 * The drop-in boundary for CherryML's composite-likelihood hot path: the
 * quantised-branch-length matrix-exponential bank expm(t_b Q) and the
 * count-weighted log-likelihood / gradient reduction that the reference
 * evaluates once per optimiser epoch.  The reference has no FFI seam on this
 * path (it is Python calling torch); every entry point below names the
 * reference code it replaces (paths relative to the reference repository).
 *
 * Conventions
 *   - all matrices are row-major float64; S = number of states, L = number of
 *     independent problems ("sites"; L = 1 for the LG / co-evolution models),
 *     B = number of quantised branch lengths ("buckets").
 *   - every function returns 0 on success or a negative CB_E* code and leaves
 *     a message retrievable with cb_last_error() (thread-local).
 *   - the caller owns every buffer it passes; the library owns what is behind
 *     the opaque handle.  `flags & CB_PTR_DEVICE` says that the data pointers
 *     of that call are device pointers (e.g. torch `tensor.data_ptr()` of a
 *     ROCm tensor) and must then be valid on the handle's device; without it
 *     they are host pointers and the call copies synchronously.
 *   - one handle = one GPU = one HIP stream; a handle is not thread-safe, the
 *     library is re-entrant across handles and keeps no global state.
 */
#ifndef CHERRYBANK_H
#define CHERRYBANK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cb_bank *cb_handle;

enum {
  CB_OK = 0,
  CB_EINVAL = -1,   /* bad argument                                  */
  CB_EHIP = -2,     /* a HIP runtime call failed                     */
  CB_ENOMEM = -3,   /* device allocation failed                      */
  CB_ENUMERIC = -4, /* non-finite input / eigensolver did not converge */
  CB_EUNSUPPORTED = -5
};

enum { CB_F64 = 0, CB_F32 = 1, CB_MIXED = 2 }; /* cb_create: element type of the bank products */

enum {
  CB_PTR_DEVICE = 1, /* data pointers of this call are device pointers */
  CB_NORMALIZE = 2,  /* divide each site's loss (and gradient) by its total count
                        (trainer.py:176-177 `loss_normalization`)     */
  CB_NO_SYNC = 4,    /* with CB_PTR_DEVICE: enqueue only, do not wait  */
  CB_EXPM_ONLY = 8,  /* cb_create: no counts (C may be NULL); the handle serves cb_expm_bank / cb_eigh only */
  CB_TRAIN_RESUME = 16, /* cb_train_pande_reversible, S > 32: continue the optimisation the previous call on this
                         * handle ended (see there) */
  CB_PER_BUCKET_PRODUCTS = 32 /* cb_create, S > 32: always form W_b = U^T G_b U bucket by bucket.  By default a float64 bank
                        * with symmetric counts and >= 24 live buckets sums the buckets BEFORE the last product (one pass over
                        * T_b = G_b U and seven single products instead of a third S^3 product per bucket: 10 % off the
                        * 129-bucket epoch); its dL/dQ is then accurate to 1e-13 .. 1e-11 of its norm instead of 3e-16
                        * (differences of one-sided sums cancel for close eigenvalues), which Adam's per-parameter scaling
                        * turns into trajectories 1e-11 .. 1e-10 from the per-bucket ones after 50 epochs -- four orders inside the
                        * 1e-6 bar on the learned matrix.  csrc/large_bank.hip.h (ky_reduce_loss), DESIGN.md section 2.
                        * The flag also keeps the bank out of the TIME BASIS (below): every bucket then has its own three
                        * products, the reference point of the accuracy tests. */
};

/* ABI version of the loaded library (bumped on incompatible change). */
int cb_version(void);

/* Message of the last failing call on this thread ("" if none). */
const char *cb_last_error(void);

/* Number of visible HIP devices (0 when there is none; never fails). */
int cb_device_count(void);

/*
 * Upload a bank once: t[L*B] branch lengths and C[L*B*S*S] count matrices.
 * Replaces the per-epoch host->device copy of (t, C) in
 * cherryml/estimation/_ratelearn/trainer.py:164-167 and the tensorisation in
 * ratelearner.py:147-152 / _siterm/_cherryml_vectorized.py:297-299: the
 * counts stay resident in HBM (stored transposed per bucket, the layout the
 * kernels stream) for the life of the handle.
 * dtype = element type of the bank products P_b, G_b U, (T_b^T U) o Phi_b: CB_F64, or CB_F32
 * -- the reference's own arithmetic (ratelearner.py:98,107: float32 parameters, Q and matrix_exp; float64
 * counts and loss, :147-152) -- for S > 32: float32 operands and MFMA accumulation at twice the float64
 * matrix rate and half the bytes; the eigendecomposition, the loss accumulation, the divided differences,
 * the sum over buckets and every vector / matrix crossing this ABI stay float64.  CB_MIXED keeps P_b, the
 * loss and G_b = -C_b / P_b in float64 too (the O(t^2) entries of P_b, which divide counts, keep their
 * relative accuracy), rounds G_b to float32 once and runs the two contractions G_b U and (T_b^T U) o Phi_b
 * on the float32 MFMA.  S <= 32: a SINGLE bank (L == 1, S >= 4) takes CB_F32 / CB_MIXED through the same tile kernels
 * (padded to 32 columns; an arithmetic mode -- the reference's own LG arithmetic --, slower than the float64 small-state
 * kernels); batches of sites (L > 1) are CB_F64 only (CB_EUNSUPPORTED otherwise).
 */
int cb_create(int device, int S, int L, int B, int dtype, const double *t, const double *C,
              int flags, cb_handle *out);

void cb_destroy(cb_handle h);

/* own == 0: run all work of this handle on `hip_stream` (a hipStream_t, e.g.
 * torch's current stream; NULL is HIP's default stream, which is what torch
 * uses unless told otherwise).  own != 0: back to the handle's own stream. */
int cb_set_stream(cb_handle h, void *hip_stream, int own);

/* Total count n_l = sum(C[l]) per site, n[L] (host pointer). */
int cb_total_counts(cb_handle h, double *n);

/* Number of non-empty buckets (C[l,b] != 0) per site, nlive[L] (host pointer).  Empty
 * buckets add nothing to the loss or its gradient (the reference still exponentiates them,
 * trainer.py:170; only SiteRM compacts, _site_specific_rate_matrix.py:590-615): cb_create
 * stores the live buckets only and every loss / training entry point visits only those.
 * cb_expm_bank still returns all B buckets in the caller's order. */
int cb_live_buckets(cb_handle h, int *nlive);

/*
 * One evaluation of the epoch body for reversible Q:
 *   loss[l] = - sum_b <C[l,b], log expm(t[l,b] Q[l])>      (/ n_l with CB_NORMALIZE)
 *   dQ[l]   = d loss[l] / d Q[l]   as a free S x S matrix
 * Replaces trainer.py:170-177 (forward) and the `loss.backward()` of
 * trainer.py:186 down to the Q node, and _cherryml_vectorized.py:264-293,378.
 * pi[L*S] is the stationary distribution Q[l] is reversible with respect to
 * (softmax of the reference's `_pi` / `theta`, rate.py:184); it is used only
 * to symmetrise, A = D^1/2 Q D^-1/2, so that the bank is evaluated through one
 * symmetric eigendecomposition A = U diag(lam) U^T per matrix.
 * dQ may be NULL (loss only).
 * S > 32, >= 28 live buckets, symmetric counts (CB_F64 / CB_MIXED): the bank runs in its time basis (cb_time_basis_info), which
 * serves spectra up to 3 x 2 max|Q_ii| of the matrix it was built for.  A call outside cb_train_* therefore reads max|Q_ii| back
 * after its eigensolve -- one 8-byte copy and a wait on the handle's stream, also with CB_NO_SYNC | CB_PTR_DEVICE, whose
 * "enqueue only" contract does not hold for these banks -- and the FIRST call, or one whose max|Q_ii| has grown 2.4x or shrunk 64x
 * since the basis was built, spends ~30 ms of host time building a new one.  CB_PER_BUCKET_PRODUCTS at cb_create avoids both
 * (every bucket's own products, ~1.4x the time per evaluation at 129 buckets).
 */
int cb_loss_grad(cb_handle h, const double *Q, const double *pi, int flags,
                 double *loss, double *dQ);

/*
 * Same contract for a general (non-reversible) Q, e.g. the reference's
 * parameterisation under a non-symmetric mask (rate.py:182): scaling and
 * squaring with a Taylor polynomial forward, its exact adjoint backward
 * (the algorithm class of torch.matrix_exp, trainer.py:170-172,186).
 * S <= 32: one workgroup per site, everything in its LDS / L2 scratch; S > 32 (L == 1): the same algebra
 * as batched 80 x 80-tile float64 MFMA products over the buckets (about 50 + 3 s products per bucket with s
 * squarings -- ~17x the arithmetic of the spectral path; CB_F64 / CB_MIXED handles).
 */
int cb_loss_grad_general(cb_handle h, const double *Q, int flags, double *loss,
                         double *dQ);

/* Debug / oracle cross-check: P[L*B*S*S] = expm(t[l,b] Q[l]) by the same device
 * code path as cb_loss_grad (pi != NULL) or cb_loss_grad_general (pi == NULL). */
int cb_expm_bank(cb_handle h, const double *Q, const double *pi, int flags,
                 double *P);

/* Debug: symmetric eigendecomposition used by the bank, A[L*S*S] ->
 * lam[L*S], U[L*S*S] (columns are eigenvectors), by the device Jacobi solver. */
int cb_eigh(cb_handle h, const double *A, int flags, double *lam, double *U);

/*
 * Live per-phase timing with HIP events on the handle's stream (for bench.py's
 * roofline figure).  cb_profile(h, 1) makes every following cb_loss_grad record
 * events around its phases; cb_last_timings() waits for them and returns the
 * milliseconds of the last call: ms[CB_T_*], n = number of slots provided.
 * cb_profile(h, n) with n > 1: the C-driven large trainer (cb_train_pande_reversible, S > 32) records the events of every n-th
 * epoch only -- a completion event costs the epoch ~2.5 us, the six of an epoch 2 % of the 400-state headline; every other
 * profiled call records all of its own.  The averages of cb_timing_sums are over the recorded epochs.
 */
enum {
  CB_T_TOTAL = 0,   /* whole call                                   */
  CB_T_EIGH = 1,    /* symmetrise + eigendecomposition              */
  CB_T_K1 = 2,      /* large path: spectral tables (4 us) + the Pt / loss / Gt kernel; when the three bank products ran as
                       ONE launch (k123_bank, the default whenever the gradient is wanted): that whole launch, and
                       CB_T_K2 = CB_T_K3 = 0 */
  CB_T_K2 = 3,      /* large path: T = Gt U (one launch)            */
  CB_T_K3 = 4,      /* large path: Mt accumulation (one launch)     */
  CB_T_K4 = 5,      /* large path: reduce + back-rotation           */
  CB_T_SMALL = 6,   /* small path: the fused per-site kernel (one launch) */
  CB_T_ALLREDUCE = 7, /* sharded C-driven loop (cb_allreduce_setup): the two ncclAllReduce calls of an epoch */
  CB_T_COUNT = 8
};
int cb_profile(cb_handle h, int enable);
int cb_last_timings(cb_handle h, double *ms, int n);
/* Sums over all profiled cb_loss_grad calls since cb_profile(h, 1) was last
 * called: ms_sum[CB_T_*] and the number of calls (averages = sum / calls). */
int cb_timing_sums(cb_handle h, double *ms_sum, int n, int *calls);
/* sweeps used by the last large-path eigendecomposition: tournament (Jacobi) sweeps of the cold path
 * + hybrid / first-order sweeps of a warm-started solve */
int cb_last_sweeps(cb_handle h);
/* Warm solves of the large-path eigensolver run as device-controlled PLANS inside cb_train_pande_reversible (S > 32):
 * the sweep decisions are taken on the device and the host only looks at a solve's record while the epoch's bank kernels
 * are queued (csrc/eigh_planned.hip.h).  counts[0] = planned solves so far on this handle, counts[1] = how many of them
 * ran out of plan before converging and were continued with more slots ("stalls"; rare), counts[2] = sweeps of the last
 * planned solve, counts[3] = how often the host looked for a solve's record before the device had published it (each such
 * look is a spin on pinned memory with the epoch's bank kernels queued behind the solve).  The test hook CB_EIGH_HOST=1
 * (honoured only together with CB_TEST_HOOKS=1, like every switch that changes which kernels run) keeps the host-driven
 * solver of rounds 1-3 (counts stay 0). */
int cb_eigh_counters(cb_handle h, int *counts, int n);
/* Which kernels the last cb_train_* call on this handle launched (tests pin the form they compare): 1000 + 100 TS +
 * 10 sym + w3 = the site-parallel split sp_prepare / sp_bank<TS, sym, w3> / sp_finish (TS = ceil(S / 4) tiles, sym =
 * symmetric-count form, w3 = three workgroups per CU); 2000 = lg_prepare / lg_bank / lg_finish (one bank, 24 < S <= 32);
 * 3000 = the one-kernel trainer; 4000 = the C-driven S > 32 loop; 0 = none yet. */
int cb_last_kernel_form(cb_handle h);
/* How the last S > 32 evaluation on this handle ran its three bank products (chosen from the bank's shape, csrc/cherrybank.hip,
 * large_eval): bit 0 = K1 -> K2 (-> K3) as ONE persistent launch (k123_bank; else separate launches), bit 1 = eight-wave tiles
 * (two K-groups; else four waves per tile), bit 2 = the buckets were summed BEFORE the last product (symmetric counts: no
 * third product per bucket, csrc/large_bank.hip.h ky_reduce_loss / kphi_combine), bit 3 = the bank ran in the TIME BASIS
 * (csrc/tbasis.hip.h: the products on a few skeleton buckets, one elementwise kernel over all buckets in between; bits 0 and 2
 * are then clear).  0 before the first evaluation. */
int cb_last_bank_form(cb_handle h);
/* The time basis of the last evaluation that used one: n[0] = skeleton buckets of the short-branch forward family, n[1] =
 * long-branch buckets that keep their own product, n[2] = skeleton buckets of the gradient family, n[3] = how often the
 * basis has been built on this handle, n[4] = how many training epochs found their matrix outside the basis' range on the
 * device and were repeated with per-bucket products (n has room for 5 ints); rho_max = the spectral bound it serves.  All
 * zero when none has been built. */
int cb_time_basis_info(cb_handle h, int *n, double *rho_max);
/* HOST-ONLY (no GPU needed): the interpolative decomposition over the branch-length grid that the S > 32 bank uses when it has
 * >= 28 live buckets with symmetric counts (CB_F64, CB_MIXED).  Every per-bucket quantity of the bank is a smooth function of t_b on
 * the spectrum [-rho_max, 0]:  phi2(t_b lam) / t_b^2 = sum_r Ls[b][r] phi2(t_s(r) lam) / t_s(r)^2  (buckets with
 * t_b rho_max <= 8; kind[b] = -1), and  e^{t_b mu} = sum_r (Lg[b][r] t_g(r) / t_b) e^{t_g(r) mu}  (all buckets), to 1e-16.
 * n_out[0..2] = ns, nd, ng; kind[B] (-1, or the index k >= 0 of a long-branch bucket that keeps its own product);
 * skel_s[ns], skel_g[ng] = the skeleton buckets; Ls [B][24], Lg [B][40] zero padded; resid[2] = largest residuals on the
 * builder's sample grid.  Any output pointer but n_out may be NULL.  CB_EUNSUPPORTED when the grid needs more than 24 / 40
 * skeleton buckets.  Replaces nothing in the reference: there every bucket is an independent matrix exponential
 * (cherryml/estimation/_ratelearn/trainer.py:170-172). */
int cb_time_basis(int B, const double *t, double rho_max, int *n_out, int *kind, int *skel_s, int *skel_g, double *Ls,
                  double *Lg, double *resid);

/*
 * Fused optimiser for the reference's `pande_reversible` parameterisation
 * (rate.py:167-188) -- `num_epochs` iterations of
 *   Q = Q(upper_diag, log_pi, mask); loss, grad; best-iterate bookkeeping;
 *   Adam / SGD step
 * entirely on the device (trainer.py:156-218 + torch.optim.Adam as configured
 * in ratelearner.py:123-130: betas (0.9, 0.999), eps 1e-8, no weight decay).
 * S <= 32: one to three launches per epoch (LG-sized problems are launch-latency bound), any L;
 * S > 32: L == 1, the epoch driven from C over the whole chip (train_large.hip.h).
 * L > 1 = L independent problems, each with its own parameters (the reference's per-site SiteRM
 * loop, _siterm/_site_specific_rate_matrix.py:43-84, 659-684, as ONE batched launch sequence):
 *   upper_diag[L*S(S-1)/2], log_pi[L*S]  in: initial parameters, out: final ones
 *   mask[S*S]                        0/1 (symmetric), NULL = all ones; shared by the L problems
 *   loss_curve[num_epochs*L]         loss of every epoch (pre-step), [epoch][l]
 *   Q_best[L*S*S], Q_last[L*S*S]     as trainer.py:179-181,237-242
 *   Q_pow2[n_pow2*S*S]               Q at epochs 1,2,4,... (trainer.py:183-184); may be NULL; L == 1 only
 * All pointers are host pointers.
 * flags & CB_TRAIN_RESUME (S > 32): the call CONTINUES the optimisation the previous successful call on this
 * handle ended -- parameters, Adam moments and step count, best loss and Q_best are the handle's (the
 * upper_diag / log_pi inputs are ignored; the mask's contents, do_adam, lr and CB_NORMALIZE must be the same, else
 * CB_EINVAL), the eigensolver stays warm;
 * loss_curve holds this call's num_epochs losses, Q_pow2 must be NULL.  E epochs then E' resumed epochs equal
 * E + E' epochs of one call bit for bit.  (The reference has no such entry: its loop runs to the end in one
 * call; this is what lets a caller put a barrier or a checkpoint between epochs of the device-driven loop.)
 */
int cb_train_pande_reversible(cb_handle h, double *upper_diag, double *log_pi,
                              const double *mask, int num_epochs, double lr,
                              int do_adam, int flags, double *loss_curve,
                              double *Q_best, double *Q_last, double *Q_pow2,
                              int n_pow2);

/*
 * Fused optimiser for the SiteRM parameterisation (theta[L*N], Theta[L*N*N];
 * _cherryml_vectorized.py:173-262) with per-site best-iterate tracking
 * (:366-372) and Adam lr (:327).  Host pointers.
 *   res[L*N*N]                       best Q per site
 *   loss_per_epoch_per_site[E*L]     may be NULL
 */
int cb_train_siterm(cb_handle h, double *theta, double *Theta, int num_epochs,
                    double lr, int flags, double *res,
                    double *loss_per_epoch_per_site);

/*
 * The `time` column of the reference's df_res (trainer.py:207-217: seconds since the start of the loop, taken at the end
 * of every epoch) for the last cb_train_pande_reversible / cb_train_siterm call on this handle: seconds[e] = seconds from
 * the entry of that call to the end of epoch e's parameter step ON THE DEVICE (the loops never return to the host between
 * epochs: the device stamps its constant 100 MHz clock; site 0's epochs for L > 1).  n <= the epochs of that call.
 */
int cb_train_epoch_times(cb_handle h, double *seconds, int n);

/* ------------------------------------------------------------------------------
 * Counting stage: the producer of the bank.  Replaces the hot loops
 * of cherryml/counting/_count_transitions.py:98-123,143-197 (C++: _count_transitions.cpp:
 * 316-390) and cherryml/counting/_count_co_transitions.py:108-140,169-223 (C++:
 * _count_co_transitions.cpp:359-381).  The host parses trees / MSAs and decides which
 * sequence pairs are counted (edge, cherry, cherry++ pairing); the device does the
 * per-site work: quantise (len_a + len_b) * rate onto the grid exactly as
 * cherryml/utils.py:35-56 (nearest in relative error, ties right, outside -> dropped)
 * and histogram the states.  Counts are INTEGERS (u64 atomics): bit-exact and
 * order-independent; the caller multiplies by the unit (0.5 / 1 / 0.25).
 * Sequences are int8 codes: state index, or negative for any symbol outside the alphabet.
 * flags = 0: all pointers are host pointers, `counts` is overwritten; the call is stateless
 * and synchronous.  flags & CB_PTR_DEVICE: every pointer (grid, seqs, rates / contacts,
 * pairs, counts) is a device pointer on `device`, the kernel is enqueued on HIP's default
 * stream and ADDS into `counts` (the caller zeroes it, validates its offsets, and
 * synchronises): the form a resident pipeline and bench.py use.  In that form bits 8..23
 * of `flags` may carry the largest pair.n (sites per pair); when given and the [B][S][S]
 * histogram fits LDS, cb_count_transitions uses the LDS-privatised kernel.  cb_count_co_transitions bins its
 * (pair, contact) events by bucket and needs their total to size the event array: the host form sums pair.n, the resident
 * form reads the total back after its first two kernels (one 8-byte copy: this form synchronises the default stream
 * once per call); bits 8..23, when given, are only checked against it (CB_EINVAL when exceeded).  A resident contact list that
 * is not 8-byte aligned is counted by the slower scattered-atomic kernel (same counts).  The entry points share one scratch
 * buffer per device on HIP's default stream: do not run them concurrently on one device.
 */
typedef struct {
  int64_t seq_a, seq_b; /* byte offsets of the two encoded sequences in `seqs`        */
  int64_t aux;          /* transitions: offset (doubles) of the family's site rates;
                           co-transitions: offset (pairs) of its contact-pair list     */
  int32_t n;            /* transitions: number of sites; co: number of contact pairs  */
  int32_t reserved;
  double len_a, len_b;  /* branch length of the pair = len_a + len_b                  */
} cb_count_pair;

/* counts[B*S*S] += 1 at [q][a_k][b_k] (and at [q][b_k][a_k] when symmetric) for every
 * site k of every pair with q = quantization_idx((len_a + len_b) * rates[aux + k]). */
int cb_count_transitions(int device, int S, int B, const double *grid, const int8_t *seqs,
                         int64_t seqs_bytes, const double *rates, int64_t n_rates,
                         const cb_count_pair *pairs, int64_t n_pairs, int symmetric, int flags,
                         unsigned long long *counts);

/* counts[B*S^2*S^2]: for every pair with q = quantization_idx(len_a + len_b) and every
 * contact (i, j) of its list, states s = a_i*S + a_j, e = b_i*S + b_j and their site-swapped
 * versions s', e':  += 1 at (s,e), (s',e')  [and (e,s), (e',s') when symmetric]. */
int cb_count_co_transitions(int device, int S, int B, const double *grid, const int8_t *seqs,
                            int64_t seqs_bytes, const int32_t *contacts, int64_t n_contacts,
                            const cb_count_pair *pairs, int64_t n_pairs, int symmetric, int flags,
                            unsigned long long *counts);

/* JTT-IPW sufficient statistics (cherryml/estimation/_jtt_ipw.py:66-110 reduced to what is linear in the
 * counts): F[S][S] = sum_b sym(C_b), R[S][S] = sum_b sym(C_b) / grid[b], sym(C) = (C + C^T) / 2 when `symmetrize`, else C.
 * `counts` [B][S][S]: 8-byte unsigned integers in units of `unit` (what cb_count_transitions / cb_count_co_transitions
 * leave on the device; counts_f64 == 0) or doubles (counts_f64 != 0; multiplied by `unit` too).  One streaming pass over
 * the tensor, sums in a fixed order.  The closed form itself (pseudocounts, mask, rates) is O(S^2) host arithmetic:
 * cherryml_amd/estimation/_jtt_ipw.py.  flags = 0: host pointers, synchronous; CB_PTR_DEVICE: device pointers, enqueued on
 * HIP's default stream like the counting kernels (the caller synchronises). */
int cb_jtt_ipw_stats(int device, int S, int B, const void *counts, int counts_f64, const double *grid, double unit,
                     int symmetrize, int flags, double *F, double *R);

/* ---- multi-GPU inside the library (SURVEY 8b / 8e option 1) ------------------------------------------
 * After this call cb_loss_grad and cb_loss_grad_general return the sums over all ranks of the
 * communicator: every rank holds a shard of the buckets (its own cb_create), evaluates its partial
 * (loss, dL/dQ) normalised by the GLOBAL totals n_total[L] (CB_NORMALIZE) and the library enqueues
 * ncclAllReduce(sum) on both results on the handle's stream, in place -- S*S + 1 doubles per site per
 * call, no other traffic.  `rccl_comm` is an ncclComm_t; `nccl_allreduce_fn` is the address of
 * ncclAllReduce of the RCCL that created it (C/C++: (void *)&ncclAllReduce; Python:
 * ctypes.cast(librccl.ncclAllReduce, c_void_p)), so the library has no link-time dependency on RCCL.
 * rccl_comm == NULL switches the reduction off again.
 * cb_train_pande_reversible on such a handle (S > 32) runs the WHOLE sharded epoch loop from C: the
 * eigensolve is replicated, this rank's buckets give partial (loss, dL/dA), two ncclAllReduce calls per
 * epoch (1 and LD*LD doubles) sum them on the handle's stream, and every rank takes the identical Adam
 * step; the call itself is collective (it all-reduces the count margins of the direct log-pi term once)
 * and so is every later training call.  The fused small-state trainers (S <= 32) refuse such a handle
 * (CB_EUNSUPPORTED): a small bank does not shard, sites are independent.
 * Python: ShardedBank.enable_in_library_allreduce() (cherryml_amd/distributed.py) creates the
 * communicator through ctypes; torch.distributed's own collective is the fallback. */
int cb_allreduce_setup(cb_handle h, void *rccl_comm, void *nccl_allreduce_fn, const double *n_total);

/* ---- SiteRM count / pseudocount assembly of ONE family ----------------------------
 * Replaces the per-transition / per-site Python loops of
 * cherryml/_siterm/_site_specific_rate_matrix.py:189-261 (_get_raw_count_matrices) and :503-567
 * (pseudocounts + lambda mix inside _estimate_site_specific_rate_matrices_given_tree_and_site_rates):
 *   raw[l,b,x,y]  = number of transitions whose total length len_a + len_b quantises to bucket b
 *                   (utils.py:35-56; outside the grid: dropped) with states (x, y) at site l;
 *                   include_reverse != 0: raw <- (raw + raw^T) / 2
 *   counts[l,b]   = (1 - lambda) raw[l,b] + lambda * sum(raw[l,b]) * prior[b_adj(l,b)],
 *   b_adj(l,b)    = quantization_idx(grid[b] * site_rates[l]), clamped into the grid
 * seqs / pairs: as cb_count_transitions (int8 state codes, -1 = not a state; byte offsets; every
 * pair spans n_sites codes; pair.aux / pair.n are ignored).  prior [B][S][S] = diag(pi0) expm(t_b Q0).
 * All inputs are HOST pointers.  counts [n_sites][B][S][S] doubles: host pointer, or with
 * CB_PTR_DEVICE a device pointer (the tensor then never leaves the GPU and feeds cb_create).
 * The arithmetic is the reference's float64 arithmetic bit for bit. */
int cb_siterm_assemble(int device, int S, int B, int n_sites, const double *grid, const int8_t *seqs,
                       int64_t seqs_bytes, const cb_count_pair *pairs, int64_t n_pairs,
                       const double *site_rates, const double *prior, double lambda,
                       int include_reverse, int flags, double *counts, double *kernel_ms);
/* kernel_ms (may be NULL): GPU time of the clear + count + mix kernels, inputs resident, by HIP events. */
/* MANY families in one call (the reference runs SiteRM family by family over a process pool, utils.py:59-67):
 * the site axis of site_rates and counts is the families' sites concatenated (n_sites[f] each), pairs are
 * the families' transitions concatenated (n_pairs[f] each, byte offsets into the one seqs buffer; a pair of
 * family f spans n_sites[f] codes).  Sites are independent: the result equals cb_siterm_assemble's family
 * by family, one upload and one synchronisation for all of them, and the tensor feeds ONE cb_create /
 * cb_train_siterm over all families' sites (a single family rarely fills 256 compute units). */
int cb_siterm_assemble_batch(int device, int S, int B, int n_fam, const int *n_sites, const double *grid,
                             const int8_t *seqs, int64_t seqs_bytes, const cb_count_pair *pairs,
                             const int64_t *n_pairs, const double *site_rates, const double *prior,
                             double lambda, int include_reverse, int flags, double *counts,
                             double *kernel_ms);

/* ---- FastCherries branch lengths / site rates of ONE family -----------------------
 * Replaces cherryml/phylogeny_estimation/FastCherries/branch_length_estimation.cpp
 * (get_branch_lengths :60-103, get_site_rates :105-144, ble :146-241, initial bins :10-58) and the
 * bank of io_helpers.cpp:150-174.  All pointers are HOST pointers.
 *   logP  [T][R][S][S]  log expm(grid[t] * rates[r] * Q)
 *   cx,cy [n][L] int8   the two sequences of every cherry (state index, -1 = gap / unknown)
 * cb_ble_log_bank fills logP through the expm bank of this library (pi != NULL: Q is reversible
 * w.r.t. pi -> spectral kernels; NULL -> general scaling-and-squaring kernels).
 * cb_ble_branch_lengths / cb_ble_site_rates are the two bisection passes; cb_ble is the whole
 * coordinate ascent: initial site-rate bins from all_seqs [n_seqs][L] and the cumulative bin
 * weights[R], then site rates <-> branch lengths until the lengths stop changing or max_iters;
 * out: lengths_index[n] (into grid), rate_index[L] (into rates).
 * cb_site_rate_gather: cherryml/_siterm/fast_site_rates.pyx:8-47, tens [R][n][S][S],
 * best[L] = first r maximising log_prior[r] + sum_c tens[r][c][x][y] (states must be >= 0). */
int cb_ble_log_bank(int device, int S, int T, int R, const double *Q, const double *pi, const double *grid,
                    const double *rates, double *logP);
int cb_ble_branch_lengths(int device, int S, int T, int R, const double *logP, const int8_t *cx,
                          const int8_t *cy, int n, int L, const int *site_to_rate, int *lengths_index);
int cb_ble_site_rates(int device, int S, int T, int R, const double *logP, const int8_t *cx,
                      const int8_t *cy, int n, int L, const int *lengths_index, const double *priors,
                      int *rate_index);
int cb_ble(int device, int S, int T, int R, const double *logP, const int8_t *cx, const int8_t *cy, int n,
           int L, const int8_t *all_seqs, int n_seqs, const double *rates, const double *weights,
           int max_iters, int *lengths_index, int *rate_index, int *iterations, double *kernel_ms);
/* iterations (may be NULL): coordinate-ascent iterations run; kernel_ms (may be NULL): GPU time of
 * the ascent (first branch-length pass + all iterations), inputs already resident, by HIP events. */
/* cb_ble for MANY families in one call -- the reference maps families over a process pool
 * (cherryml/utils.py:59-67, phylogeny_estimation/_fast_cherries.py:185-281); here the bank is uploaded
 * once, the sequences in one transfer, and the ascents advance in lockstep with ONE read-back of all
 * convergence flags per round.  Family f has n[f] cherries and n_seqs[f] sequences of L[f] sites; cx, cy
 * ([n[f]][L[f]]) and all_seqs ([n_seqs[f]][L[f]]) are the families' arrays concatenated in order, and so
 * are the outputs lengths_index (sum n) and rate_index (sum L); iterations[n_fam] may be NULL.
 * Results equal cb_ble's family by family. */
int cb_ble_batch(int device, int S, int T, int R, const double *logP, int n_fam, const int *n,
                 const int *L, const int8_t *cx, const int8_t *cy, const int8_t *all_seqs,
                 const int *n_seqs, const double *rates, const double *weights, int max_iters,
                 int *lengths_index, int *rate_index, int *iterations, double *kernel_ms);
/* The DEVICE-RESIDENT form of cb_ble (round 6): the reference reads the rate matrix and computes its log-transition bank ONCE
 * (FastCherries main.cpp -> io_helpers.cpp:150-174) and then runs ble() family after family
 * (cherryml/phylogeny_estimation/_fast_cherries.py:191: one process per family, each with its own bank); cb_ble follows the
 * per-family signature and pays for 8 MB of bank upload, ten allocations and three host passes over every sequence byte on every
 * call (15 ms around 1 ms of kernels).  cb_ble_bank_create uploads logP [T][R][S][S] once and keeps it, with the Gamma(3, 3) rate
 * priors of :199-203, on `device`; cb_ble_bank_run is cb_ble on that bank -- the workspace is kept between calls, the range
 * check, both transposes and the per-site statistics of the initial bins (:10-34) run as kernels, the host sorts L integers.
 * Same outputs as cb_ble, bit for bit.  One call at a time per bank; cb_ble_bank_destroy frees it. */
typedef struct cb_ble_bank_s *cb_ble_bank;
int cb_ble_bank_create(int device, int S, int T, int R, const double *logP, const double *rates, cb_ble_bank *out);
int cb_ble_bank_run(cb_ble_bank bank, const int8_t *cx, const int8_t *cy, int n, int L, const int8_t *all_seqs, int n_seqs,
                    const double *weights, int max_iters, int *lengths_index, int *rate_index, int *iterations, double *kernel_ms);
int cb_ble_bank_destroy(cb_ble_bank bank);
int cb_site_rate_gather(int device, int S, int R, int n, int L, const double *tens, const int8_t *cx,
                        const int8_t *cy, const double *log_prior, int *best);

/* ---- held-out log-likelihood of ONE family under one model (tail of SURVEY 8f #4) ---------------
 * Replaces the per-(node, site) python loops of cherryml/evaluation/_likelihood.py:47-327
 * (`dp_likelihood_computation`: transition matrices :169-236, Felsenstein pruning in log space
 * :238-296) for one group of "units": independent sites (S1 = 0, S states, unit u evolves at
 * cat_rate[unit_cat[u]]) or contacting pairs of sites (S1 > 0, S = S1 * S1, pair state a * S1 + b).
 * All pointers are HOST pointers.
 *   Q [S][S]; pi_rev [S]: Q is reversible w.r.t. pi_rev -> spectral expm kernels (reference
 *   `reversible_*` = True); NULL -> general scaling-and-squaring kernels;
 *   pi_root [S]: the root distribution (reference pi_1 / pi_2)
 *   tree: n_nodes nodes; postorder[n_nodes] lists every node, children before parents (the last
 *   entry is the root); parent[v] (-1 for the root); length[v] = length of the edge above v.
 *   A node's children are summed in their postorder order (= the reference's Tree.children order)
 *   code_a, code_b [n_nodes][n_units] int8: observed state of the unit's (first, second) site at a
 *   LEAF, -1 = unobserved (gap / unknown symbol: all-ones observation, :105-126); rows of internal
 *   nodes are ignored (:152-167); code_b only when S1 > 0
 *   ll [n_units] out: log-likelihood of each unit (a pair's value is for both of its sites)
 *   kernel_ms (may be NULL) [2] out: GPU ms of {transition bank + pruning, pruning alone}, inputs
 *   already resident, by HIP events
 * S <= 512; S > 64 requires n_cats == 1 (the reference evaluates pairs at rate 1). */
int cb_tree_likelihood(int device, int S, int S1, const double *Q, const double *pi_rev,
                       const double *pi_root, int n_nodes, const int *postorder, const int *parent,
                       const double *length, int n_cats, const double *cat_rate, int n_units,
                       const int *unit_cat, const int8_t *code_a, const int8_t *code_b, double *ll,
                       double *kernel_ms);
/* MANY families under one model in one call (the reference maps families over a process pool,
 * _likelihood.py:474-600 with utils.py:59-67).  Family f has n_nodes[f] nodes, n_units[f] units and
 * n_cats[f] rate categories; postorder / parent / length (node indices local to the family), cat_rate,
 * unit_cat, code_a / code_b ([n_nodes[f]][n_units[f]]) and ll are the families' arrays concatenated in
 * order.  The batch shares the model: for S > 32 one counts-free bank handle serves every family and the
 * eigendecomposition of Q is computed ONCE (3 ms cold at 400 states, formerly per family); Q / pi uploads,
 * the message buffer and the stream are shared; with S > 32 and a reversible Q the host does not wait
 * between families (all branch lengths are uploaded once), otherwise once per family.
 * Results equal cb_tree_likelihood's family by family; kernel_ms as there, summed over the batch. */
int cb_tree_likelihood_batch(int device, int S, int S1, const double *Q, const double *pi_rev,
                             const double *pi_root, int n_fam, const int *n_nodes, const int *postorder,
                             const int *parent, const double *length, const int *n_cats,
                             const double *cat_rate, const int *n_units, const int *unit_cat,
                             const int8_t *code_a, const int8_t *code_b, double *ll, double *kernel_ms);
/* The model RESIDENT on the device (round 6): cb_tree_likelihood[_batch] takes the model with every call, like the reference's
 * per-family processes (cherryml/evaluation/_likelihood.py:474-600) -- Q uploaded, a counts-free expm handle made, the transition
 * bank ([cat][node][S][S]: 2.6 GB for a 1024-leaf family of the 400-state pair model) and the message buffer allocated and freed,
 * the model's eigendecomposition recomputed: 21 ms per such family around 10.7 ms of kernels.  cb_tl_model_create keeps all of that
 * on `device`; cb_tl_model_run is cb_tree_likelihood_batch's family arguments on that model (buffers grow to the largest family
 * met, the eigendecomposition is computed by the first run); same results.  One call at a time per model. */
typedef struct cb_tl_model_s *cb_tl_model;
int cb_tl_model_create(int device, int S, int S1, const double *Q, const double *pi_rev, const double *pi_root, cb_tl_model *out);
int cb_tl_model_run(cb_tl_model model, int n_fam, const int *n_nodes, const int *postorder, const int *parent, const double *length,
                    const int *n_cats, const double *cat_rate, const int *n_units, const int *unit_cat, const int8_t *code_a,
                    const int8_t *code_b, double *ll, double *kernel_ms);
int cb_tl_model_destroy(cb_tl_model model);

/* ---- count-matrix text format, host only ---------------------------------------
 * Replaces the tokenising loops of cherryml/io/_count_matrices.py:8-62 (read_count_matrices): `text` is the
 * file's body after its two header lines ("<B> matrices", "<S> states"); per bucket it holds q, a header
 * row of S state names and S rows "<state> v ... v" (any whitespace).  Writes q[B] and C[B,S,S] (values
 * bit-identical to Python's float()), and where the first bucket's S state names are in `text`
 * (label_off / label_len); every other header row / row label must equal them (CB_EINVAL otherwise, as
 * for a wrong token count or a non-number).  Buckets are parsed on n_threads host threads (<= 0: all).
 * No GPU is touched: 84 MB / 20.7 M tokens (S = 400, B = 129) in ~0.3 s instead of ~10 s. */
int cb_parse_count_matrices(const char *text, size_t len, int B, int S, double *q, double *C,
                            long long *label_off, int *label_len, int n_threads);

/* ---- FastCherries' cherry pairing, host only -------------------------------------------------------------
 * divide_and_pair of phylogeny_estimation/FastCherries/pairing_algorithms.cpp:77-175 on int8 sequences
 * [n][L] (state index, -1 unknown): the seeded divide-and-conquer over Hamming distances, with the
 * reference's decisions reproduced bit for bit (std::mt19937(seed); pivot index by libstdc++'s
 * uniform_int_distribution rule, scheme 0 = GCC >= 11 (Lemire), 1 = older scale-and-reject; tie rules).
 * pairs[2 * (n / 2)] receives the cherries as index pairs in the reference's order; returns their number. */
int cb_fc_divide_and_pair(const int8_t *seqs, int n, int L, unsigned seed, int scheme, int *pairs);

/* ---- writer side of the text formats, host only (io/_count_matrices.py:66-81, io/_rate_matrix.py) ----------
 * rows x cols values as "<label>\tv\t...\tv\n" lines, every value with the bytes of Python's repr()
 * (shortest round-trip digits, fixed notation while -4 < decimal point <= 16, ".0" on integers).
 * out needs rows * (longest label + 2 + 26 * cols) bytes. */
int cb_format_matrix_rows(const double *M, int rows, int cols, const char *labels, const long long *label_off,
                          const int *label_len, char *out, size_t cap, size_t *written);

#ifdef __cplusplus
}
#endif
#endif /* CHERRYBANK_H */
