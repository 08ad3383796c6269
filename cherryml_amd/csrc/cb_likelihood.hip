// libcherrybank: held-out log-likelihood by level-synchronous pruning.
#include "cb_internal.hip.h"
#include "common.hip.h"
#include "likelihood.hip.h"

// ---------------------------------------------------------------- held-out likelihood
extern "C" int cb_tree_likelihood(int device, int S, int S1, const double *Q, const double *pi_rev,
                                  const double *pi_root, int n_nodes, const int *postorder, const int *parent,
                                  const double *length, int n_cats, const double *cat_rate, int n_units,
                                  const int *unit_cat, const int8_t *code_a, const int8_t *code_b, double *ll,
                                  double *kernel_ms) {
  if (!Q || !pi_root || !postorder || !parent || !length || !cat_rate || !unit_cat || !code_a || !ll)
    return fail(CB_EINVAL, "cb_tree_likelihood: NULL argument");
  if (S < 2 || S > 16 * TL_NW * TL_MAXT || n_nodes < 1 || n_cats < 1 || n_units < 1)
    return fail(CB_EINVAL, "cb_tree_likelihood: bad sizes (S = %d, nodes = %d, categories = %d, units = %d)", S,
                n_nodes, n_cats, n_units);
  if (S1 < 0 || (S1 > 0 && (S1 * S1 != S || !code_b)))
    return fail(CB_EINVAL, "cb_tree_likelihood: pair model needs S = S1 * S1 and code_b");
  if (S > 64 && n_cats != 1)
    return fail(CB_EUNSUPPORTED, "cb_tree_likelihood: S > 64 takes one rate category (the reference evaluates pairs "
                "of sites at rate 1, _likelihood.py:214-230)");
  const int ndev = cb_device_count();
  if (ndev <= 0) return fail(CB_EHIP, "cb_tree_likelihood: no HIP device (this path has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(CB_EINVAL, "cb_tree_likelihood: device %d of %d", device, ndev);
  // ---- tree: heights, levels, children in post-order (= the reference's child order, _tree.py traversal)
  const int root = postorder[n_nodes - 1];
  std::vector<int> height(n_nodes, 0), nchild(n_nodes, 0), seen(n_nodes, 0);
  for (int i = 0; i < n_nodes; ++i) {
    const int v = postorder[i];
    if (v < 0 || v >= n_nodes || seen[v]) return fail(CB_EINVAL, "cb_tree_likelihood: postorder is not a permutation");
    seen[v] = 1;
    const int p = parent[v];
    if (i == n_nodes - 1) {
      if (p != -1) return fail(CB_EINVAL, "cb_tree_likelihood: the last node of postorder must be the root (parent -1)");
      break;
    }
    if (p < 0 || p >= n_nodes || seen[p]) return fail(CB_EINVAL, "cb_tree_likelihood: node %d precedes its child %d", p, v);
    if (!(length[v] >= 0.0) || !std::isfinite(length[v])) return fail(CB_EINVAL, "cb_tree_likelihood: length[%d] = %g", v, length[v]);
    height[p] = std::max(height[p], height[v] + 1);
    nchild[p]++;
  }
  std::vector<int> child_ptr(n_nodes + 1, 0), child_idx(std::max(n_nodes - 1, 1)), fill(n_nodes, 0);
  for (int v = 0; v < n_nodes; ++v) child_ptr[v + 1] = child_ptr[v] + nchild[v];
  for (int i = 0; i + 1 < n_nodes; ++i) {
    const int v = postorder[i], p = parent[v];
    child_idx[child_ptr[p] + fill[p]++] = v;
  }
  const int n_levels = height[root] + 1;
  std::vector<int> level_ptr(n_levels + 1, 0), level_nodes(n_nodes);
  for (int v = 0; v < n_nodes; ++v) level_ptr[height[v] + 1]++;
  for (int l = 0; l < n_levels; ++l) level_ptr[l + 1] += level_ptr[l];
  {
    std::vector<int> at(level_ptr.begin(), level_ptr.end() - 1);
    for (int i = 0; i < n_nodes; ++i) level_nodes[at[height[postorder[i]]]++] = postorder[i];
  }
  for (int u = 0; u < n_units; ++u)
    if (unit_cat[u] < 0 || unit_cat[u] >= n_cats) return fail(CB_EINVAL, "cb_tree_likelihood: unit_cat[%d] = %d", u, unit_cat[u]);
  for (int c = 0; c < n_cats; ++c)
    if (!(cat_rate[c] >= 0.0) || !std::isfinite(cat_rate[c])) return fail(CB_EINVAL, "cb_tree_likelihood: cat_rate[%d] = %g", c, cat_rate[c]);
  const int alpha = S1 > 0 ? S1 : S;
  for (int v = 0; v < n_nodes; ++v)
    if (!nchild[v])
      for (int u = 0; u < n_units; ++u) {
        const size_t i = (size_t)v * n_units + u;
        if (code_a[i] >= alpha || (S1 > 0 && code_b[i] >= alpha)) return fail(CB_EINVAL, "cb_tree_likelihood: state code out of range at node %d unit %d", v, u);
      }
  // ---- transition bank expm(rate_c * length_v * Q), [cat][node][S][S], by the bank's own expm kernels
  const bool large = S > 32;
  const int L = large ? 1 : n_cats, B = large ? n_cats * n_nodes : n_nodes;
  std::vector<double> t((size_t)n_cats * n_nodes);
  for (int c = 0; c < n_cats; ++c)
    for (int v = 0; v < n_nodes; ++v) t[(size_t)c * n_nodes + v] = v == root ? 0.0 : cat_rate[c] * length[v];
  cb_handle h = nullptr;
  int rc = cb_create(device, S, L, B, CB_F64, t.data(), nullptr, CB_EXPM_ONLY, &h);
  if (rc != CB_OK) return rc;
  struct Guard {
    cb_handle h;
    ~Guard() { cb_destroy(h); }
  } guard{h};
  if ((rc = cb_set_stream(h, nullptr, 0)) != CB_OK) return rc;
  const size_t SS = (size_t)S * S;
  std::vector<double> Qrep((size_t)L * SS), pirep;
  for (int l = 0; l < L; ++l) std::copy(Q, Q + SS, Qrep.begin() + (size_t)l * SS);
  if (pi_rev) {
    pirep.resize((size_t)L * S);
    for (int l = 0; l < L; ++l) std::copy(pi_rev, pi_rev + S, pirep.begin() + (size_t)l * S);
  }
  CbDevBufs d;
  const double *dQ = d.up(Qrep.data(), Qrep.size(), rc);
  const double *dpi = pi_rev ? d.up(pirep.data(), pirep.size(), rc) : nullptr;
  double *dP = d.up<double>(nullptr, (size_t)n_cats * n_nodes * SS, rc);
  const int NU = S > 64 ? (n_units + 15) / 16 * 16 : n_units;
  const size_t msg_count = (size_t)n_nodes * S * NU;
  double *dmsg = d.up<double>(nullptr, msg_count, rc);
  double *dll = d.up<double>(nullptr, n_units, rc);
  const double *dproot = d.up(pi_root, S, rc);
  const int *dlev = d.up(level_nodes.data(), n_nodes, rc), *dcp = d.up(child_ptr.data(), n_nodes + 1, rc);
  const int *dci = d.up(child_idx.data(), child_idx.size(), rc), *duc = d.up(unit_cat, n_units, rc);
  const int8_t *dca = d.up(code_a, (size_t)n_nodes * n_units, rc);
  const int8_t *dcb = S1 > 0 ? d.up(code_b, (size_t)n_nodes * n_units, rc) : nullptr;
  if (rc != CB_OK) return rc;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, evm = nullptr;
  if (kernel_ms) {
    HIP_TRY(hipEventCreate(&ev0));
    HIP_TRY(hipEventCreate(&ev1));
    HIP_TRY(hipEventCreate(&evm));
    HIP_TRY(hipStreamSynchronize(0));   // the timed region starts with resident inputs
    HIP_TRY(hipEventRecord(ev0, 0));
  }
  rc = cb_expm_bank(h, dQ, dpi, CB_PTR_DEVICE | CB_NO_SYNC, dP);
  if (kernel_ms) (void)hipEventRecord(evm, 0);
  if (rc == CB_OK) {
    TlArgs a{};
    a.S = S; a.S1 = S1; a.n_nodes = n_nodes; a.n_units = n_units; a.NU = NU; a.root = root;
    a.child_ptr = dcp; a.child_idx = dci; a.P = dP; a.unit_cat = duc;
    a.code_a = reinterpret_cast<const signed char *>(dca);
    a.code_b = reinterpret_cast<const signed char *>(dcb);
    a.pi_root = dproot; a.msg = dmsg; a.ll = dll;
    const int nt = (S + 15) / 16, Sp = nt * 16;
    const size_t lds = ((size_t)(Sp + Sp / 4) * 16 + TL_NW * 16) * sizeof(double);
    if (S > 64 && hipFuncSetAttribute(reinterpret_cast<const void *>(tl_mfma_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      rc = fail(CB_EHIP, "cb_tree_likelihood: cannot reserve %zu bytes of LDS", lds);
    for (int l = 0; l < n_levels && rc == CB_OK; ++l) {
      const int nl = level_ptr[l + 1] - level_ptr[l];
      a.n_blocks = S > 64 ? NU / 16 : (n_units + 64 / S - 1) / (64 / S);
      const int per_launch = std::max(1, (1 << 30) / a.n_blocks);   // keep the 1-D grid below 2^30 workgroups
      for (int y0 = 0; y0 < nl; y0 += per_launch) {
        TlArgs b = a;
        b.level_nodes = dlev + level_ptr[l] + y0;
        b.n_level = std::min(per_launch, nl - y0);
        const dim3 grid((unsigned)b.n_level * (unsigned)a.n_blocks);
        if (S > 64)
          hipLaunchKernelGGL(tl_mfma_kernel, grid, dim3(TL_NW * 64), lds, 0, b);
        else
          hipLaunchKernelGGL(tl_group_kernel, grid, dim3(64), 0, 0, b);
      }
    }
  }
  if (kernel_ms) {
    float ms = 0.f, ms_prune = 0.f;
    hipError_t e = hipEventRecord(ev1, 0);
    if (e == hipSuccess) e = hipEventSynchronize(ev1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, ev0, ev1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms_prune, evm, ev1);
    kernel_ms[0] = ms;
    kernel_ms[1] = ms_prune;
    (void)hipEventDestroy(ev0);
    (void)hipEventDestroy(ev1);
    (void)hipEventDestroy(evm);
    if (e != hipSuccess && rc == CB_OK) rc = fail(CB_EHIP, "cb_tree_likelihood: %s", hipGetErrorString(e));
  }
  if (rc != CB_OK) return rc;
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(ll, dll, n_units * sizeof(double), hipMemcpyDeviceToHost));
  return CB_OK;
}
