// Test-infrastructure shim (own code): C symbols over the reference's FastCherries PROGRAM
// (cherryml/phylogeny_estimation/FastCherries/fast_cherries.cpp: main :170-321,
// get_weights_for_initial_site_rates :147-167) and its pairing (pairing_algorithms.cpp: divide_and_pair
// :166-175).  Compiled together with the reference sources WHERE THEY LIE (oracle/Makefile; the
// program's `main` is renamed ref_fc_main on the compiler command line); nothing of the reference is copied.
#include <random>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "pairing_algorithms.h"

int ref_fc_main(int argc, char *argv[]);                                    // fast_cherries.cpp, renamed
std::vector<double> get_weights_for_initial_site_rates(const std::vector<double> &rate_categories);

// the whole program on files: argv as the reference's Python wrapper builds it
extern "C" int ref_fast_cherries_main(int argc, char **argv) { return ref_fc_main(argc, argv); }

extern "C" int ref_initial_weights(const double *rates, int R, double *out) {
  std::vector<double> w = get_weights_for_initial_site_rates(std::vector<double>(rates, rates + R));
  for (int i = 0; i < R; ++i) out[i] = w[i];
  return 0;
}

// sequences [n][L] (state index, -1 = unknown); sequence i is called "<i>"; out_pairs[2k], [2k+1] =
// the k-th cherry in the reference's order; returns the number of cherries
extern "C" int ref_divide_and_pair(const int *seqs, int n, int L, int seed, int *out_pairs) {
  std::vector<std::string> names(n);
  std::unordered_map<std::string, std::vector<int>> map;
  for (int i = 0; i < n; ++i) {
    names[i] = std::to_string(i);
    map[names[i]] = std::vector<int>(seqs + (size_t)i * L, seqs + (size_t)(i + 1) * L);
  }
  std::mt19937 rng(seed);
  std::vector<std::pair<std::string, std::string>> c = divide_and_pair(names, map, rng);
  for (size_t k = 0; k < c.size(); ++k) {
    out_pairs[2 * k] = std::stoi(c[k].first);
    out_pairs[2 * k + 1] = std::stoi(c[k].second);
  }
  return (int)c.size();
}
