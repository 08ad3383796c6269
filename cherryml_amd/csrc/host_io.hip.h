// Host-side parser of the count-matrix text format (reference cherryml/io/_count_matrices.py:8-62:
// "<B> matrices\n<S> states\n" then per bucket: q, a header row of S state names, S rows of
// "<state> v ... v", any whitespace).  At S = 400 the file is 84 MB / 20.7 M tokens and the stage function
// spent ~10 s in Python tokenising it -- fifteen times the 500 optimiser epochs that follow on the GPU.
// Here: one sequential scan finds where every bucket starts, then the buckets are parsed on all host
// threads.  Numbers: Clinger's exact fast path (<= 15 significant digits, |decimal exponent| <= 22: one
// correctly rounded division / multiplication), strtod (correctly rounded) for everything else, so the
// values are bit-identical to Python's float().  No GPU involved.
#pragma once
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace cb_io {
inline bool is_ws(unsigned char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }

inline double parse_number(const char *p, size_t n, bool *ok) {
  static const double pow10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                   1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
  size_t i = 0;
  bool neg = false;
  if (i < n && (p[i] == '-' || p[i] == '+')) neg = p[i++] == '-';
  unsigned long long mant = 0;
  int digits = 0, frac = 0;
  bool any = false, simple = true;
  for (; i < n && p[i] >= '0' && p[i] <= '9'; ++i) {
    any = true;
    if (mant || p[i] != '0') ++digits;
    if (digits <= 15) mant = mant * 10 + (unsigned)(p[i] - '0');
    else simple = false;
  }
  if (i < n && p[i] == '.') {
    ++i;
    for (; i < n && p[i] >= '0' && p[i] <= '9'; ++i) {
      any = true;
      if (mant || p[i] != '0') ++digits;
      if (digits <= 15) {
        mant = mant * 10 + (unsigned)(p[i] - '0');
        ++frac;
      } else {
        simple = false;
      }
    }
  }
  int e10 = 0;
  if (simple && any && i < n && (p[i] == 'e' || p[i] == 'E')) {
    size_t j = i + 1;
    bool eneg = false;
    if (j < n && (p[j] == '-' || p[j] == '+')) eneg = p[j++] == '-';
    int ev = 0, nd = 0;
    for (; j < n && p[j] >= '0' && p[j] <= '9' && nd < 4; ++j, ++nd) ev = ev * 10 + (p[j] - '0');
    if (nd > 0 && j == n) {
      e10 = eneg ? -ev : ev;
      i = j;
    }
  }
  if (simple && any && i == n) {
    const int e = e10 - frac;
    if (e == 0) {
      *ok = true;
      return neg ? -(double)mant : (double)mant;
    }
    if (e < 0 && e >= -22) {
      *ok = true;
      const double v = (double)mant / pow10[-e];
      return neg ? -v : v;
    }
    if (e > 0 && e <= 22 && mant <= 9007199254740992ull / 10000000ull) {   // product still exact in a double
      const double v = (double)mant * pow10[e];
      if (v < 9007199254740992.0) {
        *ok = true;
        return neg ? -v : v;
      }
    }
  }
  char buf[128];
  if (n >= sizeof buf) {
    *ok = false;
    return 0.0;
  }
  memcpy(buf, p, n);
  buf[n] = 0;
  char *end = nullptr;
  const double v = strtod(buf, &end);
  *ok = end == buf + n && n > 0;
  return v;
}
}  // namespace cb_io

// text: the file's body after its two header lines.  q[B], C[B*S*S] out; label_off / label_len [S]: where
// the first bucket's header names are in `text`.  Every bucket's header row and row labels must equal
// them.  Returns CB_OK, or CB_EINVAL with a message (token count, bad number, label mismatch).
extern "C" int cb_parse_count_matrices(const char *text, size_t len, int B, int S, double *q, double *C,
                                       long long *label_off, int *label_len, int n_threads) {
  if (!text || !q || !C || !label_off || !label_len || B <= 0 || S <= 0)
    return fail(CB_EINVAL, "cb_parse_count_matrices: bad argument");
  const size_t per = 1 + (size_t)S + (size_t)S * (S + 1);
  std::vector<size_t> start(B + 1, len);
  size_t ntok = 0, i = 0;
  while (i < len) {   // one sequential scan: token count and where every bucket starts
    while (i < len && cb_io::is_ws((unsigned char)text[i])) ++i;
    if (i >= len) break;
    if (ntok % per == 0 && ntok / per < (size_t)B) start[ntok / per] = i;
    ++ntok;
    while (i < len && !cb_io::is_ws((unsigned char)text[i])) ++i;
  }
  if (ntok != (size_t)B * per)
    return fail(CB_EINVAL, "count matrices: expected %d blocks of %d states (%zu tokens), found %zu tokens", B, S,
                (size_t)B * per, ntok);
  std::vector<int> status(B, 0);
  auto next = [&](size_t &pos, size_t &tlen) {
    while (pos < len && cb_io::is_ws((unsigned char)text[pos])) ++pos;
    const size_t s = pos;
    while (pos < len && !cb_io::is_ws((unsigned char)text[pos])) ++pos;
    tlen = pos - s;
    return s;
  };
  {   // labels of the first bucket
    size_t pos = start[0], tl;
    next(pos, tl);
    for (int k = 0; k < S; ++k) {
      const size_t s = next(pos, tl);
      label_off[k] = (long long)s;
      label_len[k] = (int)tl;
    }
  }
  auto same_label = [&](size_t s, size_t tl, int k) {
    return (int)tl == label_len[k] && memcmp(text + s, text + label_off[k], tl) == 0;
  };
  auto work = [&](int b0, int b1) {
    for (int b = b0; b < b1; ++b) {
      size_t pos = start[b], tl;
      bool ok = true;
      size_t s = next(pos, tl);
      q[b] = cb_io::parse_number(text + s, tl, &ok);
      if (!ok) { status[b] = 1; continue; }
      for (int k = 0; k < S && !status[b]; ++k) {
        s = next(pos, tl);
        if (!same_label(s, tl, k)) status[b] = 2;
      }
      double *Cb = C + (size_t)b * S * S;
      for (int r = 0; r < S && !status[b]; ++r) {
        s = next(pos, tl);
        if (!same_label(s, tl, r)) { status[b] = 2; break; }
        for (int c = 0; c < S; ++c) {
          s = next(pos, tl);
          // the overwhelmingly common tokens of a count file
          if (tl == 3 && text[s] == '0' && text[s + 1] == '.' && text[s + 2] == '0') { Cb[(size_t)r * S + c] = 0.0; continue; }
          Cb[(size_t)r * S + c] = cb_io::parse_number(text + s, tl, &ok);
          if (!ok) { status[b] = 1; break; }
        }
      }
    }
  };
  int nt = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
  nt = std::max(1, std::min(nt, B));
  if (nt == 1) {
    work(0, B);
  } else {
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; ++t) pool.emplace_back(work, (int)((long long)B * t / nt), (int)((long long)B * (t + 1) / nt));
    for (auto &th : pool) th.join();
  }
  for (int b = 0; b < B; ++b) {
    if (status[b] == 1) return fail(CB_EINVAL, "count matrices: matrix %d holds a token that is not a number", b);
    if (status[b] == 2) return fail(CB_EINVAL, "count matrices: state labels of matrix %d differ from the first matrix", b);
  }
  return CB_OK;
}

// ---------------------------------------------------------------------------------------------------
// FastCherries' divide-and-conquer cherry pairing (reference
// phylogeny_estimation/FastCherries/pairing_algorithms.cpp:14-175), host only.  Same decisions as the
// reference bit for bit: std::mt19937(seed), the pivot index by libstdc++'s uniform_int_distribution rule
// (scheme 0: GCC >= 11, Lemire's multiply-shift with rejection; scheme 1: older scale-and-reject --
// written out here so that the result does not depend on the C++ library this file is built with),
// negated normalised Hamming distances as doubles, first-minimum / >= tie rules.  Sequences are int8
// state indices (-1 unknown) in one contiguous array, subsets are index vectors: no string maps, no
// per-distance hashing -- the reference's C++ spends its time there.
#include <random>
namespace cb_fc {
struct Pairer {
  const int8_t *seqs;
  int L;
  std::mt19937 rng;
  int scheme;
  std::vector<int> out;   // pairs, flattened

  size_t pick(size_t n) {
    if (scheme == 0) {
      unsigned long long product = (unsigned long long)rng() * n;
      unsigned low = (unsigned)product;
      if (low < n) {
        const unsigned threshold = (unsigned)(-(unsigned)n) % (unsigned)n;
        while (low < threshold) {
          product = (unsigned long long)rng() * n;
          low = (unsigned)product;
        }
      }
      return (size_t)(product >> 32);
    }
    const unsigned long long scaling = 0xFFFFFFFFull / n, past = n * scaling;
    for (;;) {
      const unsigned long long r = rng();
      if (r < past) return (size_t)(r / scaling);
    }
  }
  double dist(int a, int b) const {   // -(mismatches / compared sites), 0 when nothing to compare
    const int8_t *x = seqs + (size_t)a * L, *y = seqs + (size_t)b * L;
    int count = 0, d = 0;
    for (int i = 0; i < L; ++i) {
      const int ok = (x[i] != -1) & (y[i] != -1);
      count += ok;
      d += ok & (x[i] != y[i]);
    }
    if (count == 0) return 0.0;
    return d * -1.0 / count;
  }
  // farthest from x (first minimum of the negated distance); distances of all members to x in `ds`
  int farthest(const std::vector<int> &ids, int x, std::vector<double> &ds) const {
    double best = 1.7976931348623157e308;
    int y = -1;
    ds.resize(ids.size());
    for (size_t i = 0; i < ids.size(); ++i) {
      const double d = dist(ids[i], x);
      ds[i] = d;
      if (d < best) {
        best = d;
        y = (int)i;
      }
    }
    return y;   // position in ids
  }
  int divide(const std::vector<int> &ids) {   // returns the unpaired sequence or -1
    if (ids.size() == 2) {
      out.push_back(ids[0]);
      out.push_back(ids[1]);
      return -1;
    }
    if (ids.size() == 1) return ids[0];
    if (ids.empty()) return -1;
    std::vector<double> dx;
    int xp = (int)pick(ids.size());
    xp = farthest(ids, ids[xp], dx);
    const int yp = farthest(ids, ids[xp], dx);
    std::vector<int> cx, cy;
    for (size_t i = 0; i < ids.size(); ++i) {
      const bool closer_x = dx[i] >= dist(ids[i], ids[yp]);
      if (closer_x && (int)i != yp) cx.push_back(ids[i]);
      else cy.push_back(ids[i]);
    }
    dx.clear();
    dx.shrink_to_fit();
    const int ux = divide(cx);
    const int uy = divide(cy);
    if (ux >= 0 && uy >= 0) {
      out.push_back(ux);
      out.push_back(uy);
      return -1;
    }
    return ux >= 0 ? ux : uy;
  }
};
}  // namespace cb_fc

// pairs[2 * (n / 2)] out; returns the number of cherries (>= 0) or a negative error code.
// NB the order of the reference's result: cherries of the x side, then of the y side, then (x-unpaired,
// y-unpaired) -- which is the order the recursion above appends in.
extern "C" int cb_fc_divide_and_pair(const int8_t *seqs, int n, int L, unsigned seed, int scheme, int *pairs) {
  if (!seqs || !pairs || n < 0 || L < 0 || scheme < 0 || scheme > 1) return fail(CB_EINVAL, "cb_fc_divide_and_pair: bad argument");
  cb_fc::Pairer p{seqs, L, std::mt19937(seed), scheme, {}};
  p.out.reserve((size_t)n);
  std::vector<int> ids(n);
  for (int i = 0; i < n; ++i) ids[i] = i;
  p.divide(ids);
  for (size_t i = 0; i < p.out.size(); ++i) pairs[i] = p.out[i];
  return (int)(p.out.size() / 2);
}

// ---------------------------------------------------------------------------------------------------
// Writer side of the same format (cherryml/io/_count_matrices.py:66-81 writes every value with Python's
// repr): rows "<label>\tv\t...\tv\n" with the shortest digits that round-trip (std::to_chars) laid out
// by Python's rule -- fixed notation while -4 < decimal point position <= 16, else d.ddde+XX; integers
// get a ".0" -- so the bytes equal what "\t".join(map(repr, row)) produces.
#include <charconv>
namespace cb_io {
inline char *put_repr(char *p, double v) {
  if (v == 0.0) {
    if (std::signbit(v)) *p++ = '-';
    *p++ = '0'; *p++ = '.'; *p++ = '0';
    return p;
  }
  if (v != v) { *p++ = 'n'; *p++ = 'a'; *p++ = 'n'; return p; }
  if (v - v != 0.0) {   // +-inf
    if (v < 0) *p++ = '-';
    *p++ = 'i'; *p++ = 'n'; *p++ = 'f';
    return p;
  }
  char buf[40];
  auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::scientific);   // [-]d[.ddd]e[+-]XX, shortest
  const char *s = buf, *end = r.ptr;
  if (*s == '-') { *p++ = '-'; ++s; }
  char digits[24];
  int nd = 0;
  const char *e = s;
  while (e < end && *e != 'e') {
    if (*e != '.') digits[nd++] = *e;
    ++e;
  }
  int ex = 0;
  {
    const char *q = e + 1;
    bool neg = false;
    if (*q == '-' || *q == '+') neg = *q++ == '-';
    while (q < end) ex = ex * 10 + (*q++ - '0');
    if (neg) ex = -ex;
  }
  const int decpt = ex + 1;
  if (decpt > -4 && decpt <= 16) {
    if (decpt <= 0) {
      *p++ = '0'; *p++ = '.';
      for (int i = 0; i < -decpt; ++i) *p++ = '0';
      for (int i = 0; i < nd; ++i) *p++ = digits[i];
    } else if (decpt >= nd) {
      for (int i = 0; i < nd; ++i) *p++ = digits[i];
      for (int i = nd; i < decpt; ++i) *p++ = '0';
      *p++ = '.'; *p++ = '0';
    } else {
      for (int i = 0; i < decpt; ++i) *p++ = digits[i];
      *p++ = '.';
      for (int i = decpt; i < nd; ++i) *p++ = digits[i];
    }
    return p;
  }
  *p++ = digits[0];
  if (nd > 1) {
    *p++ = '.';
    for (int i = 1; i < nd; ++i) *p++ = digits[i];
  }
  *p++ = 'e';
  *p++ = ex < 0 ? '-' : '+';
  const int ax = ex < 0 ? -ex : ex;
  if (ax >= 100) *p++ = (char)('0' + ax / 100);
  *p++ = (char)('0' + (ax / 10) % 10);
  *p++ = (char)('0' + ax % 10);
  return p;
}
}  // namespace cb_io

// `rows` x `cols` values, row r prefixed by label r (bytes labels[label_off[r] .. + label_len[r])).
// out must hold rows * (max label + 1 + cols * 26) bytes; *written = bytes produced.
extern "C" int cb_format_matrix_rows(const double *M, int rows, int cols, const char *labels, const long long *label_off,
                                     const int *label_len, char *out, size_t cap, size_t *written) {
  if (!M || !labels || !label_off || !label_len || !out || !written || rows < 0 || cols < 0)
    return fail(CB_EINVAL, "cb_format_matrix_rows: bad argument");
  char *p = out;
  for (int r = 0; r < rows; ++r) {
    if ((size_t)(p - out) + (size_t)label_len[r] + 2 + (size_t)cols * 26 > cap)
      return fail(CB_EINVAL, "cb_format_matrix_rows: output buffer too small");
    memcpy(p, labels + label_off[r], (size_t)label_len[r]);
    p += label_len[r];
    const double *row = M + (size_t)r * cols;
    for (int c = 0; c < cols; ++c) {
      *p++ = '\t';
      p = cb_io::put_repr(p, row[c]);
    }
    *p++ = '\n';
  }
  *written = (size_t)(p - out);
  return CB_OK;
}
