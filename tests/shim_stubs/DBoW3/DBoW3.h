// stub (see ../README.md)
#pragma once
#include <map>
#include <vector>
namespace DBoW3 {
enum LNorm { L1, L2 };
class BowVector : public std::map<unsigned, double> { public: void addWeight(unsigned, double); void normalize(LNorm); };
class FeatureVector : public std::map<unsigned, std::vector<unsigned>> { public: void addFeature(unsigned, unsigned); };
class Vocabulary;
}  // namespace DBoW3
