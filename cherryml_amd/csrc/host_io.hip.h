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
