// stub (see ../../README.md)
#pragma once
#include <map>
#include <string>
#include <vector>
#include <opencv2/core/core.hpp>
namespace DBoW3 {
enum LNorm { L1, L2 };
typedef unsigned int WordId; typedef double WordValue; typedef unsigned int NodeId;
class BowVector : public std::map<unsigned, double> { public: void addWeight(unsigned, double); void normalize(LNorm); };
class FeatureVector : public std::map<unsigned, std::vector<unsigned>> { public: void addFeature(unsigned, unsigned); };
class Vocabulary {
 public:
  Vocabulary(); Vocabulary(const std::string &); Vocabulary(int k, int L); ~Vocabulary();
  void create(const std::vector<cv::Mat> &); void load(const std::string &); void save(const std::string &, bool = true) const;
  bool empty() const; unsigned size() const;
  void transform(const std::vector<cv::Mat> &, BowVector &) const; void transform(const cv::Mat &, BowVector &) const;
  void transform(const std::vector<cv::Mat> &, BowVector &, FeatureVector &, int levelsup) const;
  double score(const BowVector &, const BowVector &) const;
};
}  // namespace DBoW3
