// Test-infrastructure shim (own code): exposes the reference's vendored Pade matrix
// exponential r8mat_expm1 (cherryml/phylogeny_estimation/FastCherries/matrix_exponential/
// matrix_exponential.cpp:134-247, column-major n x n) through a C symbol so that the oracle
// can be cross-checked against the reference's own native expm.  Compiled together with the
// reference sources WHERE THEY LIE (see oracle/Makefile); nothing of the reference is copied.
#include <cstdlib>
#include <cstring>

double *r8mat_expm1(int n, double a[]);

extern "C" int ref_expm(int n, const double *a_colmajor, double *out_colmajor) {
  double *tmp = static_cast<double *>(std::malloc(sizeof(double) * n * n));
  if (!tmp) return -1;
  std::memcpy(tmp, a_colmajor, sizeof(double) * n * n);
  double *e = r8mat_expm1(n, tmp);
  std::free(tmp);
  if (!e) return -2;
  std::memcpy(out_colmajor, e, sizeof(double) * n * n);
  delete[] e;
  return 0;
}
