// Translation unit of the time-basis bank (tbasis.hip.h): the host-side interpolative decomposition over the branch-length
// grid, the spectral tables of the virtual buckets and the elementwise kernel between the products.
#include "tbasis.hip.h"

// Host-only diagnostic / test entry (include/cherrybank.h): the decomposition cb_train_* builds for a grid and a spectral bound.
extern "C" int cb_time_basis(int B, const double *t, double rho_max, int *n_out, int *kind, int *skel_s, int *skel_g, double *Ls,
                             double *Lg, double *resid) {
  if (!t || !n_out) return cb_fail(CB_EINVAL, "cb_time_basis: NULL argument");
  CbTimeBasisHost tb;
  if (!cb_tb_build(B, t, rho_max, tb))
    return cb_fail(CB_EUNSUPPORTED, "cb_time_basis: the grid needs more than %d / %d skeleton buckets (or an argument is not a "
                                    "positive finite number)", CB_TB_RS_MAX, CB_TB_RG_MAX);
  n_out[0] = tb.ns;
  n_out[1] = tb.nd;
  n_out[2] = tb.ng;
  if (kind) memcpy(kind, tb.kind.data(), (size_t)B * sizeof(int));
  if (skel_s) memcpy(skel_s, tb.skel_s.data(), tb.skel_s.size() * sizeof(int));
  if (skel_g) memcpy(skel_g, tb.skel_g.data(), tb.skel_g.size() * sizeof(int));
  if (Ls) memcpy(Ls, tb.Ls.data(), tb.Ls.size() * sizeof(double));
  if (Lg) memcpy(Lg, tb.Lg.data(), tb.Lg.size() * sizeof(double));
  if (resid) {
    resid[0] = tb.res_s;
    resid[1] = tb.res_g;
  }
  return CB_OK;
}
