// stub (see ../../README.md): the ceres types optimizer_ceres.h names in its functor declarations
#pragma once
#include <cstddef>
#include <vector>
namespace ceres {
class CostFunction {
 public:
  virtual ~CostFunction();
  virtual bool Evaluate(double const *const *parameters, double *residuals, double **jacobians) const = 0;
};
template <int kNumResiduals, int... Ns> class SizedCostFunction : public CostFunction { public: virtual ~SizedCostFunction(); };
template <class F, int kNumResiduals, int... Ns> class AutoDiffCostFunction : public SizedCostFunction<kNumResiduals, Ns...> {
 public:
  explicit AutoDiffCostFunction(F *);
  virtual bool Evaluate(double const *const *parameters, double *residuals, double **jacobians) const;
};
template <int kNumResiduals, int... Ns> class CostFunctionToFunctor {
 public:
  explicit CostFunctionToFunctor(CostFunction *);
  template <class... T> bool operator()(const T *... xs) const;
};
class LocalParameterization {
 public:
  virtual ~LocalParameterization();
  virtual bool Plus(const double *x, const double *delta, double *x_plus_delta) const = 0;
  virtual bool ComputeJacobian(const double *x, double *jacobian) const = 0;
  virtual int GlobalSize() const = 0; virtual int LocalSize() const = 0;
};
class LossFunction { public: virtual ~LossFunction(); };
class HuberLoss : public LossFunction { public: explicit HuberLoss(double); };
}  // namespace ceres
