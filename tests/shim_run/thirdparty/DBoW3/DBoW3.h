// FUNCTIONAL minimal stand-in (tests/shim_run/README.md): the two map types the shims fill; the vocabulary is an opaque key.
#pragma once
#include <cmath>
#include <map>
#include <string>
#include <vector>
#include <opencv2/core/core.hpp>
namespace DBoW3 {
enum LNorm { L1, L2 };
class BowVector : public std::map<unsigned, double> {
 public:
  void addWeight(unsigned id, double v) { (*this)[id] += v; }
  void normalize(LNorm) {
    double s = 0;
    for (auto &e : *this) s += std::fabs(e.second);
    if (s > 0) for (auto &e : *this) e.second /= s;
  }
};
class FeatureVector : public std::map<unsigned, std::vector<unsigned>> {
 public:
  void addFeature(unsigned id, unsigned i) { (*this)[id].push_back(i); }
};
class Vocabulary {};
}  // namespace DBoW3
