/*
 * ba_oracle.c -- CPU restatement of myslam::Optimizer's pose-only and local bundle adjustment
 * (reference include/myslam/optimizer_ceres.h, src/optimizer_ceres.cpp) in plain C.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * PARITY UNPINNED: Ceres, Sophus and Eigen are absent from this image and the reference has no
 * tests.  The cost functions (B1-B3, B6, B7) are in-tree code and are restated 1:1.  What the
 * reference delegates to third parties is restated from the published algorithms:
 *   Sophus (old non-templated) SE3::exp / SE3::log / operator*       -> orc_se3_*
 *   ceres::AngleAxisToRotationMatrix                                 -> orc_angle_axis_to_R
 *   ceres::HuberLoss + Corrector (rho'' <= 0 => scale by sqrt(rho')) -> huber(), edge weights
 *   ceres::TrustRegionMinimizer + LevenbergMarquardtStrategy (1.13-2.1 control flow, defaults:
 *     initial radius 1e4, max 1e16, min_relative_decrease 1e-3, function_tolerance 1e-6,
 *     parameter_tolerance 1e-8, gradient_tolerance 1e-10, Jacobi scaling from iteration 0,
 *     LM diagonal clamp [1e-6, 1e32], monotonic steps)                -> lm_minimize()
 *   DENSE_SCHUR: eliminate point blocks, dense Cholesky of the reduced camera system,
 *     back-substitution                                              -> ba_solve()
 * DESIGN.md "LM contract" states the same rules for the HIP path.
 */
#include "oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define SMALL_EPS 1e-10 /* Sophus so3.h */

/* ----------------------------------------------------------------- SE3 ---- */
static void quat_normalize(double q[4]) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  q[0] /= n, q[1] /= n, q[2] /= n, q[3] /= n;
}
static void quat_mul(const double a[4], const double b[4], double o[4]) {
  double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  double y = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3];
  double z = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1];
  o[0] = w, o[1] = x, o[2] = y, o[3] = z;
}
static void quat_rotate(const double q[4], const double v[3], double o[3]) {
  /* Eigen QuaternionBase::_transformVector */
  double uv[3] = {q[2] * v[2] - q[3] * v[1], q[3] * v[0] - q[1] * v[2], q[1] * v[1] - q[2] * v[0]};
  uv[0] += uv[0], uv[1] += uv[1], uv[2] += uv[2];
  o[0] = v[0] + q[0] * uv[0] + (q[2] * uv[2] - q[3] * uv[1]);
  o[1] = v[1] + q[0] * uv[1] + (q[3] * uv[0] - q[1] * uv[2]);
  o[2] = v[2] + q[0] * uv[2] + (q[1] * uv[1] - q[2] * uv[0]);
}
static void quat_to_R(const double q[4], double R[9] /*row-major*/) {
  const double tx = 2 * q[1], ty = 2 * q[2], tz = 2 * q[3];
  const double twx = tx * q[0], twy = ty * q[0], twz = tz * q[0];
  const double txx = tx * q[1], txy = ty * q[1], txz = tz * q[1];
  const double tyy = ty * q[2], tyz = tz * q[2], tzz = tz * q[3];
  R[0] = 1 - (tyy + tzz), R[1] = txy - twz, R[2] = txz + twy;
  R[3] = txy + twz, R[4] = 1 - (txx + tzz), R[5] = tyz - twx;
  R[6] = txz - twy, R[7] = tyz + twx, R[8] = 1 - (txx + tyy);
}
static void so3_exp_theta(const double om[3], double q[4], double *theta) {
  *theta = sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
  double half = 0.5 * (*theta), imag, real = cos(half);
  if (*theta < SMALL_EPS) {
    double t2 = (*theta) * (*theta), t4 = t2 * t2;
    imag = 0.5 - 0.0208333 * t2 + 0.000260417 * t4;
  } else
    imag = sin(half) / (*theta);
  q[0] = real, q[1] = imag * om[0], q[2] = imag * om[1], q[3] = imag * om[2];
  quat_normalize(q);
}
static void so3_log_theta(const double q[4], double om[3], double *theta) {
  double n = sqrt(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  double w = q[0], f;
  if (n < SMALL_EPS) {
    f = 2. / w - 2. * (n * n) / (w * w * w);
  } else {
    /* the |w| < eps branch of Sophus is immediately overwritten by the next line there too */
    f = 2 * atan(n / w) / n;
  }
  *theta = f * n;
  om[0] = f * q[1], om[1] = f * q[2], om[2] = f * q[3];
}
static void mat3_mul(const double A[9], const double B[9], double C[9]) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}
static void hat(const double w[3], double O[9]) {
  O[0] = 0, O[1] = -w[2], O[2] = w[1];
  O[3] = w[2], O[4] = 0, O[5] = -w[0];
  O[6] = -w[1], O[7] = w[0], O[8] = 0;
}

/* Sophus::SE3::exp */
void orc_se3_exp(const double xi[6], double q[4], double t[3]) {
  const double *ups = xi, *om = xi + 3;
  double theta;
  so3_exp_theta(om, q, &theta);
  double Om[9], Om2[9], V[9];
  hat(om, Om);
  mat3_mul(Om, Om, Om2);
  if (theta < SMALL_EPS) {
    quat_to_R(q, V);
  } else {
    double t2 = theta * theta;
    double a = (1 - cos(theta)) / t2, b = (theta - sin(theta)) / (t2 * theta);
    for (int i = 0; i < 9; i++) V[i] = (i % 4 == 0 ? 1.0 : 0.0) + a * Om[i] + b * Om2[i];
  }
  for (int i = 0; i < 3; i++) t[i] = V[i * 3] * ups[0] + V[i * 3 + 1] * ups[1] + V[i * 3 + 2] * ups[2];
}

/* Sophus::SE3::log */
void orc_se3_log(const double q[4], const double t[3], double xi[6]) {
  double theta, om[3];
  so3_log_theta(q, om, &theta);
  double Om[9], Om2[9], Vinv[9];
  hat(om, Om);
  mat3_mul(Om, Om, Om2);
  double c = (theta < SMALL_EPS) ? (1. / 12.) : (1 - theta / (2 * tan(theta / 2))) / (theta * theta);
  for (int i = 0; i < 9; i++) Vinv[i] = (i % 4 == 0 ? 1.0 : 0.0) - 0.5 * Om[i] + c * Om2[i];
  for (int i = 0; i < 3; i++) xi[i] = Vinv[i * 3] * t[0] + Vinv[i * 3 + 1] * t[1] + Vinv[i * 3 + 2] * t[2];
  xi[3] = om[0], xi[4] = om[1], xi[5] = om[2];
}

void orc_se3_apply(const double q[4], const double t[3], const double p[3], double out[3]) {
  quat_rotate(q, p, out);
  out[0] += t[0], out[1] += t[1], out[2] += t[2];
}

/* PoseLocalParameterization::Plus, optimizer_ceres.cpp:44-53: log(exp(delta) * exp(x)) */
void orc_se3_plus(const double x[6], const double delta[6], double out[6]) {
  double qd[4], td[3], qx[4], tx[3], q[4], t[3], rt[3];
  orc_se3_exp(delta, qd, td);
  orc_se3_exp(x, qx, tx);
  quat_rotate(qd, tx, rt); /* SE3::operator*= : t += R * other.t ; q *= other.q ; normalize */
  t[0] = td[0] + rt[0], t[1] = td[1] + rt[1], t[2] = td[2] + rt[2];
  quat_mul(qd, qx, q);
  quat_normalize(q);
  orc_se3_log(q, t, out);
}

/* ceres::AngleAxisToRotationMatrix (column-major) */
void orc_angle_axis_to_R(const double aa[3], double R[9]) {
  const double theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (theta2 > DBL_EPSILON) {
    const double theta = sqrt(theta2);
    const double wx = aa[0] / theta, wy = aa[1] / theta, wz = aa[2] / theta;
    const double c = cos(theta), s = sin(theta);
    R[0] = c + wx * wx * (1.0 - c);
    R[1] = wz * s + wx * wy * (1.0 - c);
    R[2] = -wy * s + wx * wz * (1.0 - c);
    R[3] = wx * wy * (1.0 - c) - wz * s;
    R[4] = c + wy * wy * (1.0 - c);
    R[5] = wx * s + wy * wz * (1.0 - c);
    R[6] = wy * s + wx * wz * (1.0 - c);
    R[7] = -wx * s + wy * wz * (1.0 - c);
    R[8] = c + wz * wz * (1.0 - c);
  } else {
    R[0] = 1, R[1] = aa[2], R[2] = -aa[1];
    R[3] = -aa[2], R[4] = 1, R[5] = aa[0];
    R[6] = aa[1], R[7] = -aa[0], R[8] = 1;
  }
}

/* Optimizer::se3TransPoint<double>, optimizer_ceres.h:29-95 */
void orc_se3_trans_point(const double se3[6], const double pt[3], double result[3]) {
  const double upsilon[3] = {se3[0], se3[1], se3[2]};
  const double a0 = se3[3], a1 = se3[4], a2 = se3[5];
  const double theta2 = a0 * a0 + a1 * a1 + a2 * a2;
  if (theta2 > DBL_EPSILON) {
    const double theta = sqrt(theta2);
    const double costheta = cos(theta), sintheta = sin(theta);
    const double theta_inverse = 1.0 / theta;
    const double w[3] = {a0 * theta_inverse, a1 * theta_inverse, a2 * theta_inverse};
    const double w_cross_pt[3] = {w[1] * pt[2] - w[2] * pt[1], w[2] * pt[0] - w[0] * pt[2],
                                  w[0] * pt[1] - w[1] * pt[0]};
    const double tmp = (w[0] * pt[0] + w[1] * pt[1] + w[2] * pt[2]) * (1.0 - costheta);
    result[0] = pt[0] * costheta + w_cross_pt[0] * sintheta + w[0] * tmp;
    result[1] = pt[1] * costheta + w_cross_pt[1] * sintheta + w[1] * tmp;
    result[2] = pt[2] * costheta + w_cross_pt[2] * sintheta + w[2] * tmp;
    const double wxu[3] = {w[1] * upsilon[2] - w[2] * upsilon[1], w[2] * upsilon[0] - w[0] * upsilon[2],
                           w[0] * upsilon[1] - w[1] * upsilon[0]};
    const double wwxu[3] = {w[1] * wxu[2] - w[2] * wxu[1], w[2] * wxu[0] - w[0] * wxu[2],
                            w[0] * wxu[1] - w[1] * wxu[0]};
    result[0] += upsilon[0] + ((1.0 - costheta) / theta) * wxu[0] + ((theta - sintheta) / theta) * wwxu[0];
    result[1] += upsilon[1] + ((1.0 - costheta) / theta) * wxu[1] + ((theta - sintheta) / theta) * wwxu[1];
    result[2] += upsilon[2] + ((1.0 - costheta) / theta) * wxu[2] + ((theta - sintheta) / theta) * wwxu[2];
  } else {
    const double w_cross_pt[3] = {a1 * pt[2] - a2 * pt[1], a2 * pt[0] - a0 * pt[2], a0 * pt[1] - a1 * pt[0]};
    result[0] = pt[0] + w_cross_pt[0];
    result[1] = pt[1] + w_cross_pt[1];
    result[2] = pt[2] + w_cross_pt[2];
    const double wxu[3] = {a1 * upsilon[2] - a2 * upsilon[1], a2 * upsilon[0] - a0 * upsilon[2],
                           a0 * upsilon[1] - a1 * upsilon[0]};
    result[0] += upsilon[0] + wxu[0];
    result[1] += upsilon[1] + wxu[1];
    result[2] += upsilon[2] + wxu[2];
  }
}

/* PoseOnlySE3UV / PoseOnlyStereoSE3UVD / LocalBAProjectUV / LocalBAStereoProjectUVD ::Evaluate,
 * optimizer_ceres.cpp:67-102, 109-154, 320-372, 379-444.  Q-B1: residuals carry inv_sigma, the
 * Jacobians do not. */
int orc_edge_eval(const double pose[6], const double pt[3], const double obs[3], double inv_sigma,
                  const double cam[5], double r[3], double *Jp, double *Jl) {
  const double fx = cam[0], fy = cam[1], cx = cam[2], cy = cam[3], bf = cam[4];
  double pcam[3];
  orc_se3_trans_point(pose, pt, pcam);
  const double x = pcam[0], y = pcam[1], z = pcam[2];
  const double invz = 1.0 / z, invz_2 = invz * invz;
  const int stereo = !(obs[2] < 0); /* `if (ur < 0)` mono, :555 / `pix[2] < 0` :227 */
  const double uL_hat = fx * x * invz + cx;
  r[0] = (obs[0] - uL_hat) * inv_sigma;
  r[1] = (obs[1] - (fy * y * invz + cy)) * inv_sigma;
  if (stereo) r[2] = (obs[2] - (uL_hat - bf * invz)) * inv_sigma;
  if (Jp) {
    Jp[0] = -invz * fx, Jp[1] = 0, Jp[2] = x * invz_2 * fx;
    Jp[3] = x * y * invz_2 * fx, Jp[4] = -(1 + (x * x * invz_2)) * fx, Jp[5] = y * invz * fx;
    Jp[6] = 0, Jp[7] = -invz * fy, Jp[8] = y * invz_2 * fy;
    Jp[9] = (1 + y * y * invz_2) * fy, Jp[10] = -x * y * invz_2 * fy, Jp[11] = -x * invz * fy;
    if (stereo) {
      Jp[12] = Jp[0], Jp[13] = 0, Jp[14] = Jp[2] - bf * invz_2;
      Jp[15] = Jp[3] - bf * y * invz_2, Jp[16] = Jp[4] + bf * x * invz_2, Jp[17] = Jp[5];
    }
  }
  if (Jl) {
    double R[9];
    const double aa[3] = {pose[3], pose[4], pose[5]};
    orc_angle_axis_to_R(aa, R);
    Jl[0] = -fx * R[0] * invz + fx * x * R[2] * invz_2;
    Jl[1] = -fx * R[3] * invz + fx * x * R[5] * invz_2;
    Jl[2] = -fx * R[6] * invz + fx * x * R[8] * invz_2;
    Jl[3] = -fy * R[1] * invz + fy * y * R[2] * invz_2;
    Jl[4] = -fy * R[4] * invz + fy * y * R[5] * invz_2;
    Jl[5] = -fy * R[7] * invz + fy * y * R[8] * invz_2;
    if (stereo) {
      Jl[6] = Jl[0] - bf * R[2] * invz_2;
      Jl[7] = Jl[1] - bf * R[5] * invz_2;
      Jl[8] = Jl[2] - bf * R[8] * invz_2;
    }
  }
  return stereo ? 3 : 2;
}

/* ceres::HuberLoss::Evaluate */
static void huber(double a, double s, double rho[3]) {
  const double b = a * a;
  if (a > 0 && s > b) {
    const double r = sqrt(s);
    rho[0] = 2 * a * r - b;
    rho[1] = a / r > DBL_MIN ? a / r : DBL_MIN;
    rho[2] = -rho[1] / (2 * s);
  } else {
    rho[0] = s, rho[1] = 1, rho[2] = 0;
  }
}

/* ------------------------------------------------------- LM skeleton ---- */
typedef struct lm_problem {
  void *ctx;
  int nx;                                                     /* ambient size of x */
  /* evaluate at x: cost, (r', J'') with loss correction + Jacobi scaling; returns gradient max-norm.
   * first != 0 => compute the Jacobi scaling from this evaluation. */
  int (*linearize)(void *ctx, const double *x, int first, double *cost, double *gmax);
  /* solve with current linearisation; writes delta (ambient tangent, un-scaled) and model change */
  int (*step)(void *ctx, double radius, double *delta, double *model_cost_change);
  void (*plus)(void *ctx, const double *x, const double *delta, double *xn);
  int (*cost)(void *ctx, const double *x, double *cost);
  double (*norm)(void *ctx, const double *x, const double *y /*NULL => |x|, else |x-y|*/);
} lm_problem;

static void lm_minimize(lm_problem *P, double *x, int max_iterations, orc_lm_summary *sum) {
  const double min_relative_decrease = 1e-3, function_tolerance = 1e-6, parameter_tolerance = 1e-8,
               gradient_tolerance = 1e-10, max_radius = 1e16, min_radius = 1e-32;
  double radius = 1e4, decrease_factor = 2.0;
  double *delta = (double *)calloc(P->nx, sizeof(double));
  double *xc = (double *)malloc(sizeof(double) * P->nx);
  double x_cost = 0, gmax = 0;
  orc_lm_summary local;
  if (!sum) {
    memset(&local, 0, sizeof(local));
    sum = &local;
  }
  sum->max_iterations = max_iterations;
  sum->iterations = 0;
  sum->accepted = 0;
  sum->termination = 0;
  if (!P->linearize(P->ctx, x, 1, &x_cost, &gmax)) {
    sum->termination = 4;
    goto done;
  }
  sum->initial_cost = x_cost;
  double x_norm = P->norm(P->ctx, x, NULL);
  if (sum->trace_cost) sum->trace_cost[0] = x_cost;
  if (sum->trace_radius) sum->trace_radius[0] = radius;
  if (sum->trace_accepted) sum->trace_accepted[0] = 0;
  int invalid_streak = 0;
  int last_successful = 0;
  for (int it = 1;; it++) {
    /* FinalizeIterationAndCheckIfMinimizerCanContinue of the previous iteration */
    if (it - 1 >= max_iterations) {
      sum->termination = 0;
      break;
    }
    if (last_successful && gmax <= gradient_tolerance) {
      sum->termination = 3;
      break;
    }
    if (radius < min_radius) {
      sum->termination = 4;
      break;
    }
    sum->iterations = it;
    last_successful = 0;
    double model_change = 0;
    int ok = P->step(P->ctx, radius, delta, &model_change);
    if (!ok || !(model_change > 0.0)) { /* HandleInvalidStep */
      if (++invalid_streak >= 5) {
        sum->termination = 4;
        break;
      }
      radius = radius / decrease_factor; /* StepIsInvalid -> StepRejected(0) */
      decrease_factor *= 2.0;
      if (sum->trace_cost) sum->trace_cost[it] = x_cost;
      if (sum->trace_radius) sum->trace_radius[it] = radius;
      if (sum->trace_accepted) sum->trace_accepted[it] = 0;
      continue;
    }
    invalid_streak = 0;
    P->plus(P->ctx, x, delta, xc);
    double cand_cost;
    if (!P->cost(P->ctx, xc, &cand_cost)) cand_cost = DBL_MAX;
    /* ParameterToleranceReached */
    const double step_norm = P->norm(P->ctx, x, xc);
    if (step_norm <= parameter_tolerance * (x_norm + parameter_tolerance)) {
      sum->termination = 2;
      if (sum->trace_cost) sum->trace_cost[it] = x_cost;
      if (sum->trace_radius) sum->trace_radius[it] = radius;
      if (sum->trace_accepted) sum->trace_accepted[it] = 0;
      break;
    }
    /* FunctionToleranceReached */
    const double cost_change = x_cost - cand_cost;
    if (fabs(cost_change) <= function_tolerance * x_cost) {
      sum->termination = 1;
      if (sum->trace_cost) sum->trace_cost[it] = x_cost;
      if (sum->trace_radius) sum->trace_radius[it] = radius;
      if (sum->trace_accepted) sum->trace_accepted[it] = 0;
      break;
    }
    const double relative_decrease = cost_change / model_change;
    if (relative_decrease > min_relative_decrease) { /* HandleSuccessfulStep */
      memcpy(x, xc, sizeof(double) * P->nx);
      x_norm = P->norm(P->ctx, x, NULL);
      if (!P->linearize(P->ctx, x, 0, &x_cost, &gmax)) {
        sum->termination = 4;
        break;
      }
      const double t = 2.0 * relative_decrease - 1.0;
      double f = 1.0 - t * t * t;
      if (f < 1.0 / 3.0) f = 1.0 / 3.0;
      radius = radius / f;
      if (radius > max_radius) radius = max_radius;
      decrease_factor = 2.0;
      sum->accepted++;
      last_successful = 1;
      if (sum->trace_accepted) sum->trace_accepted[it] = 1;
    } else { /* HandleUnsuccessfulStep */
      radius = radius / decrease_factor;
      decrease_factor *= 2.0;
      if (sum->trace_accepted) sum->trace_accepted[it] = 0;
    }
    if (sum->trace_cost) sum->trace_cost[it] = x_cost;
    if (sum->trace_radius) sum->trace_radius[it] = radius;
  }
done:
  sum->final_cost = x_cost;
  sum->final_radius = radius;
  free(delta);
  free(xc);
}

/* dense symmetric positive definite solve (Cholesky, lower), in place on A (n x n row-major), b */
static int chol_solve(double *A, double *b, int n) {
  for (int j = 0; j < n; j++) {
    double d = A[j * n + j];
    for (int k = 0; k < j; k++) d -= A[j * n + k] * A[j * n + k];
    if (!(d > 0.0)) return 0;
    d = sqrt(d);
    A[j * n + j] = d;
    for (int i = j + 1; i < n; i++) {
      double s = A[i * n + j];
      for (int k = 0; k < j; k++) s -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = s / d;
    }
  }
  for (int i = 0; i < n; i++) {
    double s = b[i];
    for (int k = 0; k < i; k++) s -= A[i * n + k] * b[k];
    b[i] = s / A[i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = b[i];
    for (int k = i + 1; k < n; k++) s -= A[k * n + i] * b[k];
    b[i] = s / A[i * n + i];
  }
  return 1;
}

static int inv3_spd(const double H[9], double Hi[9]) {
  double A[9], e[3];
  for (int c = 0; c < 3; c++) {
    memcpy(A, H, sizeof(A));
    e[0] = e[1] = e[2] = 0;
    e[c] = 1;
    if (!chol_solve(A, e, 3)) return 0;
    Hi[0 * 3 + c] = e[0], Hi[1 * 3 + c] = e[1], Hi[2 * 3 + c] = e[2];
  }
  return 1;
}

/* ---------------------------------------------------------- pose only ---- */
typedef struct {
  int n;
  const double *pts, *obs, *inv_sigma, *cam;
  const uint8_t *active; /* per obs */
  double huber_mono, huber_stereo;
  double *J, *r; /* per obs 18 + 3, corrected + scaled */
  int *rows;
  double scale[6];
  double g[6];
} pose_ctx;

static int pose_linearize(void *vc, const double *x, int first, double *cost, double *gmax) {
  pose_ctx *c = (pose_ctx *)vc;
  double total = 0;
  double g[6] = {0, 0, 0, 0, 0, 0}, cn[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < c->n; i++) {
    c->rows[i] = 0;
    if (c->active && !c->active[i]) continue;
    double *J = c->J + (size_t)i * 18, *r = c->r + (size_t)i * 3;
    int m = orc_edge_eval(x, c->pts + 3 * i, c->obs + 3 * i, c->inv_sigma[i], c->cam, r, J, NULL);
    c->rows[i] = m;
    double s = 0;
    for (int k = 0; k < m; k++) s += r[k] * r[k];
    double rho[3];
    huber(m == 2 ? c->huber_mono : c->huber_stereo, s, rho);
    total += 0.5 * rho[0];
    const double w = sqrt(rho[1]);
    for (int k = 0; k < m; k++) r[k] *= w;
    for (int k = 0; k < m * 6; k++) J[k] *= w;
    for (int k = 0; k < m; k++)
      for (int a = 0; a < 6; a++) {
        g[a] += J[k * 6 + a] * r[k];
        cn[a] += J[k * 6 + a] * J[k * 6 + a];
      }
  }
  if (first)
    for (int a = 0; a < 6; a++) c->scale[a] = 1.0 / (1.0 + sqrt(cn[a]));
  for (int i = 0; i < c->n; i++)
    for (int k = 0; k < c->rows[i]; k++)
      for (int a = 0; a < 6; a++) c->J[(size_t)i * 18 + k * 6 + a] *= c->scale[a];
  double m = 0;
  for (int a = 0; a < 6; a++) {
    c->g[a] = g[a];
    if (fabs(g[a]) > m) m = fabs(g[a]);
  }
  *cost = total;
  *gmax = m;
  return isfinite(total);
}
static int pose_step(void *vc, double radius, double *delta, double *model_change) {
  pose_ctx *c = (pose_ctx *)vc;
  double H[36], b[6];
  memset(H, 0, sizeof(H));
  memset(b, 0, sizeof(b));
  for (int i = 0; i < c->n; i++) {
    const double *J = c->J + (size_t)i * 18, *r = c->r + (size_t)i * 3;
    for (int k = 0; k < c->rows[i]; k++)
      for (int a = 0; a < 6; a++) {
        b[a] += J[k * 6 + a] * r[k];
        for (int bb = 0; bb < 6; bb++) H[a * 6 + bb] += J[k * 6 + a] * J[k * 6 + bb];
      }
  }
  for (int a = 0; a < 6; a++) {
    double d = H[a * 6 + a];
    d = d < 1e-6 ? 1e-6 : (d > 1e32 ? 1e32 : d);
    H[a * 6 + a] += d / radius;
  }
  if (!chol_solve(H, b, 6)) return 0;
  double step[6];
  for (int a = 0; a < 6; a++) {
    if (!isfinite(b[a])) return 0;
    step[a] = -b[a];
  }
  double mc = 0;
  for (int i = 0; i < c->n; i++) {
    const double *J = c->J + (size_t)i * 18, *r = c->r + (size_t)i * 3;
    for (int k = 0; k < c->rows[i]; k++) {
      double mr = 0;
      for (int a = 0; a < 6; a++) mr += J[k * 6 + a] * step[a];
      mc += mr * (r[k] + mr / 2.0);
    }
  }
  *model_change = -mc;
  for (int a = 0; a < 6; a++) delta[a] = step[a] * c->scale[a];
  return 1;
}
static void pose_plus(void *vc, const double *x, const double *d, double *xn) {
  (void)vc;
  orc_se3_plus(x, d, xn);
}
static int pose_cost(void *vc, const double *x, double *cost) {
  pose_ctx *c = (pose_ctx *)vc;
  double total = 0;
  for (int i = 0; i < c->n; i++) {
    if (c->active && !c->active[i]) continue;
    double r[3];
    int m = orc_edge_eval(x, c->pts + 3 * i, c->obs + 3 * i, c->inv_sigma[i], c->cam, r, NULL, NULL);
    double s = 0;
    for (int k = 0; k < m; k++) s += r[k] * r[k];
    double rho[3];
    huber(m == 2 ? c->huber_mono : c->huber_stereo, s, rho);
    total += 0.5 * rho[0];
  }
  *cost = total;
  return isfinite(total);
}
static double pose_norm(void *vc, const double *x, const double *y) {
  (void)vc;
  double s = 0;
  for (int a = 0; a < 6; a++) {
    double d = y ? x[a] - y[a] : x[a];
    s += d * d;
  }
  return sqrt(s);
}

/* float chi2 classification shared by solvePoseOnlySE3 (:262-303) and local BA (:626-688,:718-754)
 * pc = camera-frame point (double).  Returns 1 if outlier.  Q-B2: float arithmetic. */
static int chi2_outlier_pose_only(const double pc[3], const double obs[3], float fx, float fy,
                                  float cx, float cy, float bf, double inv_sigma2_d) {
  const double x = pc[0], y = pc[1], z = pc[2];
  const float invz = (float)(1.0f / z);              /* `const float invz = 1.0f/z;` z double */
  const float u = (float)(fx * x * invz + cx);       /* float*double -> double -> float */
  const float v = (float)(fy * y * invz + cy);
  const float eu = (float)(u - obs[0]);
  const float ev = (float)(v - obs[1]);
  const float e2 = eu * eu + ev * ev;
  const float invSigma2 = (float)inv_sigma2_d;
  if (obs[2] < 0) return !(e2 * invSigma2 < 5.991f);
  const float ur = u - bf * invz;
  const float e_ur = (float)(ur - obs[2]);
  const float eu2 = e2 + e_ur * e_ur;
  return !(eu2 * invSigma2 < 7.815f);
}

/* Optimizer::solvePoseOnlySE3, optimizer_ceres.cpp:157-314 */
int orc_pose_only_solve(int n, const double *pts, const double *obs, const double *inv_sigma,
                        const double cam[5], double pose[6], uint8_t *outlier,
                        orc_lm_summary *sums) {
  if (n <= 0) return 0; /* :204-205 */
  double pose_backup[6];
  memcpy(pose_backup, pose, sizeof(pose_backup));
  for (int i = 0; i < n; i++) outlier[i] = 0; /* :199 */
  pose_ctx c;
  memset(&c, 0, sizeof(c));
  c.n = n, c.pts = pts, c.obs = obs, c.inv_sigma = inv_sigma, c.cam = cam;
  c.J = (double *)malloc(sizeof(double) * 18 * (size_t)n);
  c.r = (double *)malloc(sizeof(double) * 3 * (size_t)n);
  c.rows = (int *)malloc(sizeof(int) * (size_t)n);
  uint8_t *active = (uint8_t *)malloc(n);
  c.active = active;
  const float fx = (float)cam[0], fy = (float)cam[1], cx = (float)cam[2], cy = (float)cam[3],
              bf = (float)cam[4];
  int inlier_cnt = 0;
  double q[4], t[3];
  for (int iter = 0; iter < 2; iter++) {
    memcpy(pose, pose_backup, sizeof(pose_backup)); /* :215 */
    c.huber_mono = iter < 1 ? (double)sqrtf(5.991f) : 0.0;   /* :219 */
    c.huber_stereo = iter < 1 ? (double)sqrtf(7.815f) : 0.0; /* :220 */
    int nact = 0;
    for (int i = 0; i < n; i++) nact += (active[i] = !outlier[i]);
    lm_problem P = {&c, 6, pose_linearize, pose_step, pose_plus, pose_cost, pose_norm};
    if (nact > 0) lm_minimize(&P, pose, 10, sums ? &sums[iter] : NULL);
    orc_se3_exp(pose, q, t); /* :256-257 */
    inlier_cnt = 0;
    for (int i = 0; i < n; i++) {
      double pc[3];
      orc_se3_apply(q, t, pts + 3 * i, pc);
      const double is2 = inv_sigma[i] * inv_sigma[i]; /* invSigmas2 :196 */
      outlier[i] = (uint8_t)chi2_outlier_pose_only(pc, obs + 3 * i, fx, fy, cx, cy, bf, is2);
      inlier_cnt += !outlier[i];
    }
    if (inlier_cnt < 10) break; /* :306-307 */
  }
  /* frame->setPose(Tcw) with Tcw = exp(pose): the se3 vector is returned as is */
  free(c.J);
  free(c.r);
  free(c.rows);
  free(active);
  return inlier_cnt;
}

/* ----------------------------------------------------------- BA (Schur) -- */
typedef struct {
  int n_cams, n_pts, n_edges;
  const uint8_t *cam_fixed;
  const int32_t *e_cam, *e_pt;
  const double *e_obs, *e_inv_sigma, *cam;
  const uint8_t *edge_active;
  double huber_mono, huber_stereo;
  /* problem structure */
  int nf;          /* free cams present in the problem */
  int *cam_slot;   /* cam -> reduced index or -1 */
  uint8_t *pt_in;  /* point has >=1 active edge */
  uint8_t *cam_in; /* cam (free) has >=1 active edge */
  /* linearisation */
  double *Jp, *Jl, *r; /* per edge 18, 9, 3 */
  int *rows;
  double *scale_c, *scale_p; /* Jacobi scaling per cam (6) / point (3) */
  double gmax;
} ba_ctx;

#define XP(c) ((c) * 6)
#define XL(ctx, j) ((ctx)->n_cams * 6 + (j) * 3)

static void ba_structure(ba_ctx *c) {
  memset(c->pt_in, 0, c->n_pts);
  memset(c->cam_in, 0, c->n_cams);
  for (int e = 0; e < c->n_edges; e++) {
    if (c->edge_active && !c->edge_active[e]) continue;
    c->pt_in[c->e_pt[e]] = 1;
    if (!c->cam_fixed[c->e_cam[e]]) c->cam_in[c->e_cam[e]] = 1;
  }
  c->nf = 0;
  for (int k = 0; k < c->n_cams; k++) c->cam_slot[k] = c->cam_in[k] ? c->nf++ : -1;
}

static int ba_linearize(void *vc, const double *x, int first, double *cost, double *gmax) {
  ba_ctx *c = (ba_ctx *)vc;
  double total = 0;
  double *gc = (double *)calloc((size_t)c->n_cams * 6, sizeof(double));
  double *gl = (double *)calloc((size_t)c->n_pts * 3, sizeof(double));
  double *nc = (double *)calloc((size_t)c->n_cams * 6, sizeof(double));
  double *nl = (double *)calloc((size_t)c->n_pts * 3, sizeof(double));
  for (int e = 0; e < c->n_edges; e++) {
    c->rows[e] = 0;
    if (c->edge_active && !c->edge_active[e]) continue;
    const int ci = c->e_cam[e], pj = c->e_pt[e];
    double *Jp = c->Jp + (size_t)e * 18, *Jl = c->Jl + (size_t)e * 9, *r = c->r + (size_t)e * 3;
    int m = orc_edge_eval(x + XP(ci), x + XL(c, pj), c->e_obs + 3 * (size_t)e, c->e_inv_sigma[e],
                          c->cam, r, Jp, Jl);
    c->rows[e] = m;
    double s = 0;
    for (int k = 0; k < m; k++) s += r[k] * r[k];
    double rho[3];
    huber(m == 2 ? c->huber_mono : c->huber_stereo, s, rho);
    total += 0.5 * rho[0];
    const double w = sqrt(rho[1]);
    for (int k = 0; k < m; k++) r[k] *= w;
    for (int k = 0; k < m * 6; k++) Jp[k] *= w;
    for (int k = 0; k < m * 3; k++) Jl[k] *= w;
    for (int k = 0; k < m; k++) {
      if (!c->cam_fixed[ci])
        for (int a = 0; a < 6; a++) {
          gc[ci * 6 + a] += Jp[k * 6 + a] * r[k];
          nc[ci * 6 + a] += Jp[k * 6 + a] * Jp[k * 6 + a];
        }
      for (int a = 0; a < 3; a++) {
        gl[pj * 3 + a] += Jl[k * 3 + a] * r[k];
        nl[pj * 3 + a] += Jl[k * 3 + a] * Jl[k * 3 + a];
      }
    }
  }
  if (first) {
    for (int i = 0; i < c->n_cams * 6; i++) c->scale_c[i] = 1.0 / (1.0 + sqrt(nc[i]));
    for (int i = 0; i < c->n_pts * 3; i++) c->scale_p[i] = 1.0 / (1.0 + sqrt(nl[i]));
  }
  for (int e = 0; e < c->n_edges; e++) {
    const int ci = c->e_cam[e], pj = c->e_pt[e];
    for (int k = 0; k < c->rows[e]; k++) {
      for (int a = 0; a < 6; a++) c->Jp[(size_t)e * 18 + k * 6 + a] *= c->scale_c[ci * 6 + a];
      for (int a = 0; a < 3; a++) c->Jl[(size_t)e * 9 + k * 3 + a] *= c->scale_p[pj * 3 + a];
    }
  }
  double m = 0;
  for (int k = 0; k < c->n_cams; k++)
    if (c->cam_in[k])
      for (int a = 0; a < 6; a++)
        if (fabs(gc[k * 6 + a]) > m) m = fabs(gc[k * 6 + a]);
  for (int j = 0; j < c->n_pts; j++)
    if (c->pt_in[j])
      for (int a = 0; a < 3; a++)
        if (fabs(gl[j * 3 + a]) > m) m = fabs(gl[j * 3 + a]);
  free(gc), free(gl), free(nc), free(nl);
  *cost = total;
  *gmax = m;
  c->gmax = m;
  return isfinite(total);
}

/* Schur elimination on the stored (corrected, scaled) Jacobians.  lm_radius <= 0: no damping but
 * `point_damping` added to the point diagonals.  Outputs S (6nf)^2, rhs; keeps per-point inverse and
 * gl for back-substitution when Hinv/gl_out given. */
static int ba_schur_build(ba_ctx *c, double lm_radius, double point_damping, double *S, double *rhs,
                          double *Hinv /*n_pts*9*/, double *gl_out /*n_pts*3*/, double *Dc, double *Dl) {
  const int nf = c->nf, N = 6 * nf;
  memset(S, 0, sizeof(double) * (size_t)N * N);
  memset(rhs, 0, sizeof(double) * (size_t)N);
  /* edges grouped by point: build CSR on the fly */
  int *start = (int *)calloc((size_t)c->n_pts + 1, sizeof(int));
  for (int e = 0; e < c->n_edges; e++)
    if (c->rows[e]) start[c->e_pt[e] + 1]++;
  for (int j = 0; j < c->n_pts; j++) start[j + 1] += start[j];
  int *items = (int *)malloc(sizeof(int) * (size_t)(start[c->n_pts] > 0 ? start[c->n_pts] : 1));
  int *fill = (int *)calloc((size_t)c->n_pts, sizeof(int));
  for (int e = 0; e < c->n_edges; e++)
    if (c->rows[e]) items[start[c->e_pt[e]] + fill[c->e_pt[e]]++] = e;
  free(fill);
  /* camera blocks + their LM diagonal */
  double *dc = (double *)calloc((size_t)N, sizeof(double));
  for (int e = 0; e < c->n_edges; e++) {
    const int ci = c->e_cam[e];
    if (!c->rows[e] || c->cam_slot[ci] < 0) continue;
    const int o = 6 * c->cam_slot[ci];
    const double *Jp = c->Jp + (size_t)e * 18, *r = c->r + (size_t)e * 3;
    for (int k = 0; k < c->rows[e]; k++)
      for (int a = 0; a < 6; a++) {
        rhs[o + a] += Jp[k * 6 + a] * r[k];
        for (int b = 0; b < 6; b++) S[(size_t)(o + a) * N + o + b] += Jp[k * 6 + a] * Jp[k * 6 + b];
      }
  }
  for (int i = 0; i < N; i++) {
    double d = S[(size_t)i * N + i];
    d = d < 1e-6 ? 1e-6 : (d > 1e32 ? 1e32 : d);
    dc[i] = lm_radius > 0 ? d / lm_radius : 0.0;
    S[(size_t)i * N + i] += dc[i];
  }
  if (Dc) memcpy(Dc, dc, sizeof(double) * N);
  int ok = 1;
  for (int j = 0; j < c->n_pts && ok; j++) {
    if (start[j] == start[j + 1]) continue;
    double H[9] = {0}, g[3] = {0};
    for (int t = start[j]; t < start[j + 1]; t++) {
      const int e = items[t];
      const double *Jl = c->Jl + (size_t)e * 9, *r = c->r + (size_t)e * 3;
      for (int k = 0; k < c->rows[e]; k++)
        for (int a = 0; a < 3; a++) {
          g[a] += Jl[k * 3 + a] * r[k];
          for (int b = 0; b < 3; b++) H[a * 3 + b] += Jl[k * 3 + a] * Jl[k * 3 + b];
        }
    }
    for (int a = 0; a < 3; a++) {
      double d = H[a * 4];
      d = d < 1e-6 ? 1e-6 : (d > 1e32 ? 1e32 : d);
      double dl = lm_radius > 0 ? d / lm_radius : point_damping;
      if (Dl) Dl[j * 3 + a] = dl;
      H[a * 4] += dl;
    }
    double Hi[9];
    if (!inv3_spd(H, Hi)) {
      ok = 0;
      break;
    }
    if (Hinv) memcpy(Hinv + (size_t)j * 9, Hi, sizeof(Hi));
    if (gl_out) memcpy(gl_out + (size_t)j * 3, g, sizeof(g));
    double Hig[3];
    for (int a = 0; a < 3; a++) Hig[a] = Hi[a * 3] * g[0] + Hi[a * 3 + 1] * g[1] + Hi[a * 3 + 2] * g[2];
    /* W_e = Jp^T Jl (6x3), Y_e = W_e Hinv */
    for (int t1 = start[j]; t1 < start[j + 1]; t1++) {
      const int e1 = items[t1], c1 = c->cam_slot[c->e_cam[e1]];
      if (c1 < 0) continue;
      double W1[18] = {0}, Y1[18];
      for (int k = 0; k < c->rows[e1]; k++)
        for (int a = 0; a < 6; a++)
          for (int b = 0; b < 3; b++)
            W1[a * 3 + b] += c->Jp[(size_t)e1 * 18 + k * 6 + a] * c->Jl[(size_t)e1 * 9 + k * 3 + b];
      for (int a = 0; a < 6; a++)
        for (int b = 0; b < 3; b++)
          Y1[a * 3 + b] = W1[a * 3] * Hi[b] + W1[a * 3 + 1] * Hi[3 + b] + W1[a * 3 + 2] * Hi[6 + b];
      for (int a = 0; a < 6; a++)
        rhs[6 * c1 + a] -= W1[a * 3] * Hig[0] + W1[a * 3 + 1] * Hig[1] + W1[a * 3 + 2] * Hig[2];
      for (int t2 = start[j]; t2 < start[j + 1]; t2++) {
        const int e2 = items[t2], c2 = c->cam_slot[c->e_cam[e2]];
        if (c2 < 0) continue;
        double W2[18] = {0};
        for (int k = 0; k < c->rows[e2]; k++)
          for (int a = 0; a < 6; a++)
            for (int b = 0; b < 3; b++)
              W2[a * 3 + b] += c->Jp[(size_t)e2 * 18 + k * 6 + a] * c->Jl[(size_t)e2 * 9 + k * 3 + b];
        for (int a = 0; a < 6; a++)
          for (int b = 0; b < 6; b++)
            S[(size_t)(6 * c1 + a) * N + 6 * c2 + b] -=
                Y1[a * 3] * W2[b * 3] + Y1[a * 3 + 1] * W2[b * 3 + 1] + Y1[a * 3 + 2] * W2[b * 3 + 2];
      }
    }
  }
  free(dc);
  free(start);
  free(items);
  return ok;
}

static int ba_step(void *vc, double radius, double *delta, double *model_change) {
  ba_ctx *c = (ba_ctx *)vc;
  const int nf = c->nf, N = 6 * nf;
  double *S = (double *)malloc(sizeof(double) * (size_t)(N > 0 ? N : 1) * (N > 0 ? N : 1));
  double *y = (double *)malloc(sizeof(double) * (size_t)(N > 0 ? N : 1));
  double *Hinv = (double *)calloc((size_t)c->n_pts * 9, sizeof(double));
  double *gl = (double *)calloc((size_t)c->n_pts * 3, sizeof(double));
  int ok = ba_schur_build(c, radius, 0.0, S, y, Hinv, gl, NULL, NULL);
  if (ok && N > 0) ok = chol_solve(S, y, N);
  if (ok) {
    /* back-substitution: y_l = Hinv (gl - sum_e W_e^T y_c) */
    double *yl = gl; /* reuse: accumulate gl - W^T y */
    for (int e = 0; e < c->n_edges; e++) {
      const int cs = c->cam_slot[c->e_cam[e]];
      if (!c->rows[e] || cs < 0) continue;
      const int pj = c->e_pt[e];
      for (int k = 0; k < c->rows[e]; k++) {
        double jy = 0;
        for (int a = 0; a < 6; a++) jy += c->Jp[(size_t)e * 18 + k * 6 + a] * y[6 * cs + a];
        for (int b = 0; b < 3; b++) yl[pj * 3 + b] -= c->Jl[(size_t)e * 9 + k * 3 + b] * jy;
      }
    }
    memset(delta, 0, sizeof(double) * ((size_t)c->n_cams * 6 + (size_t)c->n_pts * 3));
    double *step_c = (double *)calloc((size_t)c->n_cams * 6, sizeof(double));
    double *step_l = (double *)calloc((size_t)c->n_pts * 3, sizeof(double));
    for (int k = 0; k < c->n_cams; k++)
      if (c->cam_slot[k] >= 0)
        for (int a = 0; a < 6; a++) step_c[k * 6 + a] = -y[6 * c->cam_slot[k] + a];
    for (int j = 0; j < c->n_pts; j++)
      if (c->pt_in[j])
        for (int a = 0; a < 3; a++)
          step_l[j * 3 + a] = -(Hinv[j * 9 + a * 3] * yl[j * 3] + Hinv[j * 9 + a * 3 + 1] * yl[j * 3 + 1] +
                                Hinv[j * 9 + a * 3 + 2] * yl[j * 3 + 2]);
    double mc = 0;
    for (int e = 0; e < c->n_edges; e++) {
      const int ci = c->e_cam[e], pj = c->e_pt[e];
      for (int k = 0; k < c->rows[e]; k++) {
        double mr = 0;
        for (int a = 0; a < 6; a++) mr += c->Jp[(size_t)e * 18 + k * 6 + a] * step_c[ci * 6 + a];
        for (int a = 0; a < 3; a++) mr += c->Jl[(size_t)e * 9 + k * 3 + a] * step_l[pj * 3 + a];
        mc += mr * (c->r[(size_t)e * 3 + k] + mr / 2.0);
      }
    }
    *model_change = -mc;
    for (int i = 0; i < c->n_cams * 6; i++) {
      delta[i] = step_c[i] * c->scale_c[i];
      if (!isfinite(delta[i])) ok = 0;
    }
    for (int i = 0; i < c->n_pts * 3; i++) {
      delta[c->n_cams * 6 + i] = step_l[i] * c->scale_p[i];
      if (!isfinite(delta[c->n_cams * 6 + i])) ok = 0;
    }
    free(step_c);
    free(step_l);
  }
  free(S), free(y), free(Hinv), free(gl);
  return ok;
}
static void ba_plus(void *vc, const double *x, const double *d, double *xn) {
  ba_ctx *c = (ba_ctx *)vc;
  memcpy(xn, x, sizeof(double) * ((size_t)c->n_cams * 6 + (size_t)c->n_pts * 3));
  for (int k = 0; k < c->n_cams; k++)
    if (c->cam_slot[k] >= 0) orc_se3_plus(x + XP(k), d + XP(k), xn + XP(k));
  for (int j = 0; j < c->n_pts; j++)
    if (c->pt_in[j])
      for (int a = 0; a < 3; a++) xn[XL(c, j) + a] = x[XL(c, j) + a] + d[XL(c, j) + a];
}
static int ba_cost(void *vc, const double *x, double *cost) {
  ba_ctx *c = (ba_ctx *)vc;
  double total = 0;
  for (int e = 0; e < c->n_edges; e++) {
    if (c->edge_active && !c->edge_active[e]) continue;
    double r[3];
    int m = orc_edge_eval(x + XP(c->e_cam[e]), x + XL(c, c->e_pt[e]), c->e_obs + 3 * (size_t)e,
                          c->e_inv_sigma[e], c->cam, r, NULL, NULL);
    double s = 0;
    for (int k = 0; k < m; k++) s += r[k] * r[k];
    double rho[3];
    huber(m == 2 ? c->huber_mono : c->huber_stereo, s, rho);
    total += 0.5 * rho[0];
  }
  *cost = total;
  return isfinite(total);
}
static double ba_norm(void *vc, const double *x, const double *y) {
  ba_ctx *c = (ba_ctx *)vc;
  double s = 0;
  for (int k = 0; k < c->n_cams; k++)
    if (c->cam_slot[k] >= 0)
      for (int a = 0; a < 6; a++) {
        double d = y ? x[XP(k) + a] - y[XP(k) + a] : x[XP(k) + a];
        s += d * d;
      }
  for (int j = 0; j < c->n_pts; j++)
    if (c->pt_in[j])
      for (int a = 0; a < 3; a++) {
        double d = y ? x[XL(c, j) + a] - y[XL(c, j) + a] : x[XL(c, j) + a];
        s += d * d;
      }
  return sqrt(s);
}

static void ba_ctx_init(ba_ctx *c, int n_cams, const uint8_t *cam_fixed, int n_pts, int n_edges,
                        const int32_t *e_cam, const int32_t *e_pt, const double *e_obs,
                        const double *e_inv_sigma, const uint8_t *edge_active, const double *cam,
                        double hm, double hs) {
  memset(c, 0, sizeof(*c));
  c->n_cams = n_cams, c->n_pts = n_pts, c->n_edges = n_edges;
  c->cam_fixed = cam_fixed, c->e_cam = e_cam, c->e_pt = e_pt, c->e_obs = e_obs;
  c->e_inv_sigma = e_inv_sigma, c->edge_active = edge_active, c->cam = cam;
  c->huber_mono = hm, c->huber_stereo = hs;
  c->cam_slot = (int *)malloc(sizeof(int) * (size_t)(n_cams > 0 ? n_cams : 1));
  c->pt_in = (uint8_t *)malloc(n_pts > 0 ? n_pts : 1);
  c->cam_in = (uint8_t *)malloc(n_cams > 0 ? n_cams : 1);
  c->Jp = (double *)malloc(sizeof(double) * 18 * (size_t)(n_edges > 0 ? n_edges : 1));
  c->Jl = (double *)malloc(sizeof(double) * 9 * (size_t)(n_edges > 0 ? n_edges : 1));
  c->r = (double *)malloc(sizeof(double) * 3 * (size_t)(n_edges > 0 ? n_edges : 1));
  c->rows = (int *)calloc((size_t)(n_edges > 0 ? n_edges : 1), sizeof(int));
  c->scale_c = (double *)malloc(sizeof(double) * 6 * (size_t)(n_cams > 0 ? n_cams : 1));
  c->scale_p = (double *)malloc(sizeof(double) * 3 * (size_t)(n_pts > 0 ? n_pts : 1));
  for (int i = 0; i < n_cams * 6; i++) c->scale_c[i] = 1.0;
  for (int i = 0; i < n_pts * 3; i++) c->scale_p[i] = 1.0;
  ba_structure(c);
}
static void ba_ctx_free(ba_ctx *c) {
  free(c->cam_slot), free(c->pt_in), free(c->cam_in), free(c->Jp), free(c->Jl), free(c->r);
  free(c->rows), free(c->scale_c), free(c->scale_p);
}

int orc_ba_lm(int n_cams, double *poses, const uint8_t *cam_fixed, int n_pts, double *points,
              int n_edges, const int32_t *e_cam, const int32_t *e_pt, const double *e_obs,
              const double *e_inv_sigma, const uint8_t *edge_active, const double cam[5],
              double huber_mono, double huber_stereo, int max_iterations, orc_lm_summary *sum) {
  ba_ctx c;
  ba_ctx_init(&c, n_cams, cam_fixed, n_pts, n_edges, e_cam, e_pt, e_obs, e_inv_sigma, edge_active,
              cam, huber_mono, huber_stereo);
  const int nx = n_cams * 6 + n_pts * 3;
  double *x = (double *)malloc(sizeof(double) * (size_t)nx);
  memcpy(x, poses, sizeof(double) * (size_t)n_cams * 6);
  memcpy(x + n_cams * 6, points, sizeof(double) * (size_t)n_pts * 3);
  lm_problem P = {&c, nx, ba_linearize, ba_step, ba_plus, ba_cost, ba_norm};
  int any = 0;
  for (int j = 0; j < n_pts; j++) any |= c.pt_in[j];
  if (any) lm_minimize(&P, x, max_iterations, sum);
  memcpy(poses, x, sizeof(double) * (size_t)n_cams * 6);
  memcpy(points, x + n_cams * 6, sizeof(double) * (size_t)n_pts * 3);
  free(x);
  ba_ctx_free(&c);
  return 0;
}

int orc_ba_schur(int n_cams, const double *poses, const uint8_t *cam_fixed, int n_pts,
                 const double *points, int n_edges, const int32_t *e_cam, const int32_t *e_pt,
                 const double *e_obs, const double *e_inv_sigma, const uint8_t *edge_active,
                 const double cam[5], double huber_mono, double huber_stereo, double point_damping,
                 double *S, double *b, double *cost) {
  ba_ctx c;
  ba_ctx_init(&c, n_cams, cam_fixed, n_pts, n_edges, e_cam, e_pt, e_obs, e_inv_sigma, edge_active,
              cam, huber_mono, huber_stereo);
  const int nx = n_cams * 6 + n_pts * 3;
  double *x = (double *)malloc(sizeof(double) * (size_t)nx);
  memcpy(x, poses, sizeof(double) * (size_t)n_cams * 6);
  memcpy(x + n_cams * 6, points, sizeof(double) * (size_t)n_pts * 3);
  double gmax;
  /* first=0 with unit scaling => unscaled Jacobians */
  ba_linearize(&c, x, 0, cost, &gmax);
  int ok = ba_schur_build(&c, -1.0, point_damping, S, b, NULL, NULL, NULL, NULL);
  free(x);
  int nf = c.nf;
  ba_ctx_free(&c);
  return ok ? nf : -1;
}

/* chi2 classification of a BA edge: optimizer_ceres.cpp:626-688 (and :703-755).
 * x,y,z are first narrowed to float (`const float x = pcam[0]`).  Returns 1 = outlier. */
static int chi2_outlier_lba(const double pose[6], const double pt[3], const double obs[3],
                            double inv_sigma, float fx, float fy, float cx, float cy, float bf,
                            int final_pass) {
  double pcam[3];
  orc_se3_trans_point(pose, pt, pcam);
  const float x = (float)pcam[0], y = (float)pcam[1], z = (float)pcam[2];
  if (z < 0.0f) return 1;
  const float invz = 1.0f / z;
  const float u = fx * x * invz + cx;
  const float v = fy * y * invz + cy;
  float eu, ev;
  if (!final_pass) { /* :634-635: static_cast<float>(pixel[k]) */
    eu = u - (float)obs[0];
    ev = v - (float)obs[1];
  } else { /* :726-727: float - double -> double -> float */
    eu = (float)(u - obs[0]);
    ev = (float)(v - obs[1]);
  }
  const float e2 = eu * eu + ev * ev;
  const float invSigma2 = (float)(inv_sigma * inv_sigma);
  const int mono = !final_pass ? ((float)obs[2] < 0) : (obs[2] < 0);
  if (mono) return e2 * invSigma2 > 5.991f;
  const float ur = u - bf * invz;
  const float e_ur = !final_pass ? ur - (float)obs[2] : (float)(ur - obs[2]);
  const float eu2 = e2 + e_ur * e_ur;
  return eu2 * invSigma2 > 7.815f;
}

/* Optimizer::solveLocalBAPoseAndPoint numerics, optimizer_ceres.cpp:530-755 */
int orc_local_ba(int n_cams, double *poses, const uint8_t *cam_fixed, int n_pts, double *points,
                 int n_edges, const int32_t *e_cam, const int32_t *e_pt, const double *e_obs,
                 const double *e_inv_sigma, const double cam[5], const volatile int *stop,
                 uint8_t *edge_erase, orc_lm_summary *sums) {
  const float fx = (float)cam[0], fy = (float)cam[1], cx = (float)cam[2], cy = (float)cam[3],
              bf = (float)cam[4];
  for (int e = 0; e < n_edges; e++) edge_erase[e] = 0;
  if (stop && *stop) return 1; /* :594-595 */
  orc_ba_lm(n_cams, poses, cam_fixed, n_pts, points, n_edges, e_cam, e_pt, e_obs, e_inv_sigma, NULL,
            cam, (double)sqrtf(5.991f), (double)sqrtf(7.815f), 5, sums ? &sums[0] : NULL);
  uint8_t *outl = (uint8_t *)calloc(n_edges > 0 ? n_edges : 1, 1);
  uint8_t *active = (uint8_t *)calloc(n_edges > 0 ? n_edges : 1, 1);
  if (!(stop && *stop)) { /* :612 */
    for (int e = 0; e < n_edges; e++) {
      outl[e] = (uint8_t)chi2_outlier_lba(poses + 6 * e_cam[e], points + 3 * e_pt[e],
                                          e_obs + 3 * (size_t)e, e_inv_sigma[e], fx, fy, cx, cy, bf, 0);
      active[e] = !outl[e];
    }
    orc_ba_lm(n_cams, poses, cam_fixed, n_pts, points, n_edges, e_cam, e_pt, e_obs, e_inv_sigma,
              active, cam, 0.0, 0.0, 10, sums ? &sums[1] : NULL);
  }
  for (int e = 0; e < n_edges; e++) { /* :703-755 */
    if (outl[e]) {
      edge_erase[e] = 1;
      continue;
    }
    edge_erase[e] = (uint8_t)chi2_outlier_lba(poses + 6 * e_cam[e], points + 3 * e_pt[e],
                                              e_obs + 3 * (size_t)e, e_inv_sigma[e], fx, fy, cx, cy, bf, 1);
  }
  free(outl);
  free(active);
  return 0;
}

/* ------------------------------------------------------------------ Sim3 (loop closure) ----
 * Optimizer::solveLoopSim3, optimizer_ceres.cpp:810-1030, with PoseOnlySim3 / PoseOnlyInverseSim3
 * (optimizer_ceres.h:211-267) over IntrinsicProjectionUV (optimizer_ceres.cpp:8-42).
 * Parameters: pose = [angle-axis(3); t(3)] and the scale s, both with the plain additive update
 * (no local parameterisation); points constant.  Per match two 2-row residual blocks, each with
 * its own Huber(sqrt(10)) loss:
 *   forward  r = (pix_curr  - K pi(s R P_match + t)) / sigma_curr
 *   inverse  r = (pix_match - K pi(R^T (P_curr - t) / s)) / sigma_match
 * Jacobians: the autodiff chain through CostFunctionToFunctor = (analytic d r/d p of
 * IntrinsicProjectionUV, which LACKS the 1/sigma factor -- same inconsistency as Q-B1) times the
 * exact derivative of the camera-frame point, here in closed form:
 *   d(R(w) p)/dw = -R [p]x Jr(w),   d(R(w)^T v)/dw = R^T [v]x Jl(w). */
static void aa_to_R_rowmajor(const double aa[3], double R[9]) {
  double C[9];
  orc_angle_axis_to_R(aa, C); /* column-major like ceres */
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) R[3 * i + j] = C[3 * j + i];
}
static void so3_jacobians(const double w[3], double Jr[9], double Jl[9]) {
  const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double W[9], W2[9];
  hat(w, W);
  mat3_mul(W, W, W2);
  double a, b;
  if (t2 > DBL_EPSILON) { /* same branch point as ceres::AngleAxisToRotationMatrix */
    const double th = sqrt(t2);
    a = (1.0 - cos(th)) / t2;
    b = (th - sin(th)) / (t2 * th);
  } else {
    a = 0.0, b = 0.0; /* R = I + [w]x there: the derivative of that first-order form */
  }
  for (int i = 0; i < 9; i++) {
    const double id = (i % 4 == 0) ? 1.0 : 0.0;
    Jr[i] = id - a * W[i] + b * W2[i];
    Jl[i] = id + a * W[i] + b * W2[i];
  }
}

/* IntrinsicProjectionUV::Evaluate */
static void intrinsic_uv(const double p[3], const double pix[2], const double cam[4], double isig,
                         double r[2], double J[6]) {
  const double x = p[0], y = p[1], z = p[2];
  const double invz = 1.0 / z, invz2 = invz * invz;
  r[0] = (pix[0] - (cam[0] * x * invz + cam[2])) * isig;
  r[1] = (pix[1] - (cam[1] * y * invz + cam[3])) * isig;
  if (J) {
    J[0] = -invz * cam[0], J[1] = 0, J[2] = x * invz2 * cam[0];
    J[3] = 0, J[4] = -invz * cam[1], J[5] = y * invz2 * cam[1];
  }
}

/* both residual blocks of one match at x = [aa, t, s]; J blocks are 2 x 7 row-major (column 6 = s) */
void orc_sim3_eval(const double x[7], const double cam_match[3], const double pix_curr[2], double isig_c,
                   const double cam_curr[3], const double pix_match[2], double isig_m,
                   const double cam[4], double r_fwd[2], double J_fwd[14], double r_inv[2], double J_inv[14]) {
  double R[9], Jr[9], Jl[9];
  aa_to_R_rowmajor(x, R);
  const double s = x[6];
  double Rp[3];
  for (int i = 0; i < 3; i++) Rp[i] = R[3 * i] * cam_match[0] + R[3 * i + 1] * cam_match[1] + R[3 * i + 2] * cam_match[2];
  const double p[3] = {s * Rp[0] + x[3], s * Rp[1] + x[4], s * Rp[2] + x[5]};
  double Juv[6];
  intrinsic_uv(p, pix_curr, cam, isig_c, r_fwd, J_fwd ? Juv : NULL);
  if (J_fwd || J_inv) so3_jacobians(x, Jr, Jl);
  if (J_fwd) {
    double P[9], RP[9], D[9]; /* dp/dw = -s R [Pm]x Jr */
    hat(cam_match, P);
    mat3_mul(R, P, RP);
    mat3_mul(RP, Jr, D);
    for (int k = 0; k < 2; k++) {
      for (int a = 0; a < 3; a++)
        J_fwd[7 * k + a] = -s * (Juv[3 * k] * D[a] + Juv[3 * k + 1] * D[3 + a] + Juv[3 * k + 2] * D[6 + a]);
      for (int a = 0; a < 3; a++) J_fwd[7 * k + 3 + a] = Juv[3 * k + a];
      J_fwd[7 * k + 6] = Juv[3 * k] * Rp[0] + Juv[3 * k + 1] * Rp[1] + Juv[3 * k + 2] * Rp[2];
    }
  }
  const double v[3] = {(cam_curr[0] - x[3]) / s, (cam_curr[1] - x[4]) / s, (cam_curr[2] - x[5]) / s};
  double q[3];
  for (int i = 0; i < 3; i++) q[i] = R[i] * v[0] + R[3 + i] * v[1] + R[6 + i] * v[2]; /* R^T v */
  intrinsic_uv(q, pix_match, cam, isig_m, r_inv, J_inv ? Juv : NULL);
  if (J_inv) {
    double V[9], VJ[9], D[9], Rt[9]; /* dq/dw = R^T [v]x Jl */
    hat(v, V);
    mat3_mul(V, Jl, VJ);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) Rt[3 * i + j] = R[3 * j + i];
    mat3_mul(Rt, VJ, D);
    for (int k = 0; k < 2; k++) {
      for (int a = 0; a < 3; a++)
        J_inv[7 * k + a] = Juv[3 * k] * D[a] + Juv[3 * k + 1] * D[3 + a] + Juv[3 * k + 2] * D[6 + a];
      for (int a = 0; a < 3; a++) /* dq/dt = -R^T / s */
        J_inv[7 * k + 3 + a] = -(Juv[3 * k] * Rt[a] + Juv[3 * k + 1] * Rt[3 + a] + Juv[3 * k + 2] * Rt[6 + a]) / s;
      J_inv[7 * k + 6] = -(Juv[3 * k] * q[0] + Juv[3 * k + 1] * q[1] + Juv[3 * k + 2] * q[2]) / s; /* dq/ds = -q/s */
    }
  }
}

typedef struct {
  int n, np; /* np = 6 (scale fixed) or 7 */
  const double *cam_match, *pix_curr, *isig_c, *cam_curr, *pix_match, *isig_m, *cam;
  const uint8_t *active;
  double huber;
  double *J, *r; /* per match: 2 blocks x 2 rows x 7, 4 residuals; corrected + scaled */
  double scale[7];
} sim3_ctx;

static int sim3_linearize(void *vc, const double *x, int first, double *cost, double *gmax) {
  sim3_ctx *c = (sim3_ctx *)vc;
  double total = 0, g[7] = {0}, cn[7] = {0};
  for (int i = 0; i < c->n; i++) {
    if (c->active && !c->active[i]) continue;
    double *J = c->J + (size_t)i * 28, *r = c->r + (size_t)i * 4;
    orc_sim3_eval(x, c->cam_match + 3 * i, c->pix_curr + 2 * i, c->isig_c[i], c->cam_curr + 3 * i,
                  c->pix_match + 2 * i, c->isig_m[i], c->cam, r, J, r + 2, J + 14);
    for (int blk = 0; blk < 2; blk++) {
      double *rb = r + 2 * blk, *Jb = J + 14 * blk, rho[3];
      huber(c->huber, rb[0] * rb[0] + rb[1] * rb[1], rho);
      total += 0.5 * rho[0];
      const double w = sqrt(rho[1]);
      rb[0] *= w, rb[1] *= w;
      for (int k = 0; k < 14; k++) Jb[k] *= w;
      for (int k = 0; k < 2; k++)
        for (int a = 0; a < c->np; a++) {
          g[a] += Jb[7 * k + a] * rb[k];
          cn[a] += Jb[7 * k + a] * Jb[7 * k + a];
        }
    }
  }
  if (first)
    for (int a = 0; a < c->np; a++) c->scale[a] = 1.0 / (1.0 + sqrt(cn[a]));
  for (int i = 0; i < c->n; i++) {
    if (c->active && !c->active[i]) continue;
    for (int k = 0; k < 4; k++)
      for (int a = 0; a < c->np; a++) c->J[(size_t)i * 28 + 7 * k + a] *= c->scale[a];
  }
  double m = 0;
  for (int a = 0; a < c->np; a++)
    if (fabs(g[a]) > m) m = fabs(g[a]);
  *cost = total;
  *gmax = m;
  return isfinite(total);
}
static int sim3_step(void *vc, double radius, double *delta, double *model_change) {
  sim3_ctx *c = (sim3_ctx *)vc;
  const int np = c->np;
  double H[49], b[7];
  memset(H, 0, sizeof(H));
  memset(b, 0, sizeof(b));
  for (int i = 0; i < c->n; i++) {
    if (c->active && !c->active[i]) continue;
    const double *J = c->J + (size_t)i * 28, *r = c->r + (size_t)i * 4;
    for (int k = 0; k < 4; k++)
      for (int a = 0; a < np; a++) {
        b[a] += J[7 * k + a] * r[k];
        for (int bb = 0; bb < np; bb++) H[a * np + bb] += J[7 * k + a] * J[7 * k + bb];
      }
  }
  for (int a = 0; a < np; a++) {
    double d = H[a * np + a];
    d = d < 1e-6 ? 1e-6 : (d > 1e32 ? 1e32 : d);
    H[a * np + a] += d / radius;
  }
  if (!chol_solve(H, b, np)) return 0;
  double step[7];
  for (int a = 0; a < np; a++) {
    if (!isfinite(b[a])) return 0;
    step[a] = -b[a];
  }
  double mc = 0;
  for (int i = 0; i < c->n; i++) {
    if (c->active && !c->active[i]) continue;
    const double *J = c->J + (size_t)i * 28, *r = c->r + (size_t)i * 4;
    for (int k = 0; k < 4; k++) {
      double mr = 0;
      for (int a = 0; a < np; a++) mr += J[7 * k + a] * step[a];
      mc += mr * (r[k] + mr / 2.0);
    }
  }
  *model_change = -mc;
  for (int a = 0; a < 7; a++) delta[a] = a < np ? step[a] * c->scale[a] : 0.0;
  return 1;
}
static void sim3_plus(void *vc, const double *x, const double *d, double *xn) {
  (void)vc;
  for (int a = 0; a < 7; a++) xn[a] = x[a] + d[a];
}
static int sim3_cost(void *vc, const double *x, double *cost) {
  sim3_ctx *c = (sim3_ctx *)vc;
  double total = 0;
  for (int i = 0; i < c->n; i++) {
    if (c->active && !c->active[i]) continue;
    double r[4], rho[3];
    orc_sim3_eval(x, c->cam_match + 3 * i, c->pix_curr + 2 * i, c->isig_c[i], c->cam_curr + 3 * i,
                  c->pix_match + 2 * i, c->isig_m[i], c->cam, r, NULL, r + 2, NULL);
    huber(c->huber, r[0] * r[0] + r[1] * r[1], rho);
    total += 0.5 * rho[0];
    huber(c->huber, r[2] * r[2] + r[3] * r[3], rho);
    total += 0.5 * rho[0];
  }
  *cost = total;
  return isfinite(total);
}
static double sim3_norm(void *vc, const double *x, const double *y) {
  sim3_ctx *c = (sim3_ctx *)vc;
  double s = 0;
  for (int a = 0; a < c->np; a++) { /* the free parameter blocks only */
    const double d = y ? x[a] - y[a] : x[a];
    s += d * d;
  }
  return sqrt(s);
}

/* chi2 > 10 tests of :916-948 / :996-1022 (double arithmetic; camera2pixel = fx x / z + cx) */
static int sim3_outlier(const double x[7], const double cam_match[3], const double pix_curr[2], double isig_c,
                        const double cam_curr[3], const double pix_match[2], double isig_m, const double cam[4]) {
  double R[9];
  aa_to_R_rowmajor(x, R);
  const double s = x[6];
  double p[3];
  for (int i = 0; i < 3; i++)
    p[i] = s * (R[3 * i] * cam_match[0] + R[3 * i + 1] * cam_match[1] + R[3 * i + 2] * cam_match[2]) + x[3 + i];
  const double eu = cam[0] * p[0] / p[2] + cam[2] - pix_curr[0], ev = cam[1] * p[1] / p[2] + cam[3] - pix_curr[1];
  if ((eu * eu + ev * ev) * isig_c * isig_c > 10.0) return 1;
  /* Smc = Scm^-1: rotation R^T, scale 1/s, translation -R^T t / s */
  double q[3];
  for (int i = 0; i < 3; i++) {
    const double Rtp = R[i] * cam_curr[0] + R[3 + i] * cam_curr[1] + R[6 + i] * cam_curr[2];
    const double Rtt = R[i] * x[3] + R[3 + i] * x[4] + R[6 + i] * x[5];
    q[i] = Rtp / s - Rtt / s;
  }
  const double fu = cam[0] * q[0] / q[2] + cam[2] - pix_match[0], fv = cam[1] * q[1] / q[2] + cam[3] - pix_match[1];
  return (fu * fu + fv * fv) * isig_m * isig_m > 10.0;
}

int orc_sim3_solve(int n, const double *cam_match, const double *pix_curr, const double *isig_curr,
                   const double *cam_curr, const double *pix_match, const double *isig_match,
                   const double cam[4], int fix_scale, double pose[6], double *scale, uint8_t *outlier,
                   orc_lm_summary *sums) {
  double x[7], x_in[7];
  memcpy(x, pose, 6 * sizeof(double));
  x[6] = *scale;
  memcpy(x_in, x, sizeof(x));
  for (int i = 0; i < n; i++) outlier[i] = 0;
  sim3_ctx c;
  memset(&c, 0, sizeof(c));
  c.n = n, c.np = fix_scale ? 6 : 7;
  c.cam_match = cam_match, c.pix_curr = pix_curr, c.isig_c = isig_curr, c.cam_curr = cam_curr;
  c.pix_match = pix_match, c.isig_m = isig_match, c.cam = cam;
  c.huber = (double)sqrtf(10.0f); /* sqrt(10.0f) is a float expression (:880) */
  c.J = (double *)malloc(sizeof(double) * 28 * (size_t)(n > 0 ? n : 1));
  c.r = (double *)malloc(sizeof(double) * 4 * (size_t)(n > 0 ? n : 1));
  uint8_t *active = (uint8_t *)malloc(n > 0 ? n : 1);
  c.active = active;
  for (int i = 0; i < n; i++) active[i] = 1;
  lm_problem P = {&c, 7, sim3_linearize, sim3_step, sim3_plus, sim3_cost, sim3_norm};
  if (sums) memset(sums, 0, 2 * sizeof(orc_lm_summary));
  if (n > 0) lm_minimize(&P, x, 10, sums ? &sums[0] : NULL);
  int outlier_cnt = 0;
  for (int i = 0; i < n; i++) {
    outlier[i] = (uint8_t)sim3_outlier(x, cam_match + 3 * i, pix_curr + 2 * i, isig_curr[i], cam_curr + 3 * i,
                                       pix_match + 2 * i, isig_match[i], cam);
    outlier_cnt += outlier[i];
  }
  int inliers = 0;
  if (n - outlier_cnt < 10) { /* :950-951: returns before Scm is written */
    memcpy(x, x_in, sizeof(x));
  } else {
    for (int i = 0; i < n; i++) active[i] = !outlier[i];
    lm_minimize(&P, x, outlier_cnt > 0 ? 10 : 5, sums ? &sums[1] : NULL);
    for (int i = 0; i < n; i++) {
      /* :996-1022 tests every match again, including those excluded from problem 2 */
      const int o = sim3_outlier(x, cam_match + 3 * i, pix_curr + 2 * i, isig_curr[i], cam_curr + 3 * i,
                                 pix_match + 2 * i, isig_match[i], cam);
      if (o) outlier[i] = 1;
      inliers += !o;
    }
  }
  memcpy(pose, x, 6 * sizeof(double));
  *scale = x[6];
  free(c.J);
  free(c.r);
  free(active);
  return inliers;
}

/* ------------------------------------------------------------- pose graph (loop closure) ----
 * Optimizer::solvePoseGraphLoop numerics, optimizer_ceres.cpp:1036-1305, cost functor PoseGraphLoop
 * (optimizer_ceres.h:269-325).  Node = Sim3 as unit quaternion (Eigen coefficient order x,y,z,w),
 * translation, scale; edge (1 -> 2) with measurement S21 = (q21, t21, s21):
 *   r[0:3] = 2 vec(q21 * q1 * q2^-1),  r[3:6] = s21 (q21 * t12) + t21,  r[6] = s21 s1 / s2   (Q-B4)
 *   with t12 = s1 (q1 * (-(q2^-1 * (t2 / s2)))) + t1.
 * Quaternions move by ceres::EigenQuaternionParameterization (x' = dq(delta) * x, dq = [sin|d| d/|d|,
 * cos|d|]), translations additively, scales are constant (fixScaleFlag is always true,
 * loopClosing.cpp:15); no loss; LM with an exact solve of the normal equations
 * (SPARSE_NORMAL_CHOLESKY), <= 20 iterations.  Jacobians (tangent, 7 x 6 per node = [rotation,
 * translation]) in closed form; they equal autodiff-ambient x ComputeJacobian because the
 * parameterisation's columns are tangent to the unit sphere. */
static void q_mul(const double a[4], const double b[4], double o[4]) { /* x,y,z,w */
  o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  o[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
  o[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
  o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}
static void q_conj(const double a[4], double o[4]) { o[0] = -a[0], o[1] = -a[1], o[2] = -a[2], o[3] = a[3]; }
static void q_rot(const double q[4], const double v[3], double o[3]) { /* Eigen _transformVector */
  const double uv[3] = {2 * (q[1] * v[2] - q[2] * v[1]), 2 * (q[2] * v[0] - q[0] * v[2]), 2 * (q[0] * v[1] - q[1] * v[0])};
  o[0] = v[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
  o[1] = v[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
  o[2] = v[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
}
static void q_to_R(const double q[4], double R[9]) { /* row-major, unit quaternion */
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  R[0] = 1 - 2 * (y * y + z * z), R[1] = 2 * (x * y - z * w), R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w), R[4] = 1 - 2 * (x * x + z * z), R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w), R[7] = 2 * (y * z + x * w), R[8] = 1 - 2 * (x * x + y * y);
}
/* ceres::EigenQuaternionParameterization::Plus */
void orc_quat_plus(const double q[4], const double d[3], double o[4]) {
  const double n = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  if (n > 0.0) {
    const double s = sin(n) / n;
    const double dq[4] = {s * d[0], s * d[1], s * d[2], cos(n)};
    q_mul(dq, q, o);
  } else {
    o[0] = q[0], o[1] = q[1], o[2] = q[2], o[3] = q[3];
  }
}

/* one edge: residual r[7]; J1, J2 (7 x 6 row-major: d/d[delta_rot, t]) or NULL */
void orc_pose_graph_edge(const double q1[4], const double t1[3], double s1, const double q2[4], const double t2[3],
                         double s2, const double qm[4], const double tm[3], double sm, double r[7], double *J1,
                         double *J2) {
  double q2i[4], q12[4], dq[4], t2s[3], w[3], t2inv[3], rq[3], t12[3], rt[3];
  q_conj(q2, q2i);
  q_mul(q1, q2i, q12);
  q_mul(qm, q12, dq);
  for (int k = 0; k < 3; k++) t2s[k] = (1.0 / s2) * t2[k];
  q_rot(q2i, t2s, w);
  for (int k = 0; k < 3; k++) t2inv[k] = -w[k];
  q_rot(q1, t2inv, rq);
  for (int k = 0; k < 3; k++) t12[k] = s1 * rq[k] + t1[k];
  q_rot(qm, t12, rt);
  for (int k = 0; k < 3; k++) r[k] = 2.0 * dq[k], r[3 + k] = sm * rt[k] + tm[k];
  r[6] = sm * s1 * (1.0 / s2);
  if (!J1 && !J2) return;
  double Rm[9], R1[9], R2[9];
  q_to_R(qm, Rm), q_to_R(q1, R1), q_to_R(q2, R2);
  if (J1) {
    memset(J1, 0, 42 * sizeof(double));
    /* d(2 vec(qm (1 + d^) q12))/dd: column a = 2 vec(qm * e_a * q12), e_a pure unit quaternion */
    for (int a = 0; a < 3; a++) {
      double e[4] = {0, 0, 0, 0}, tmp[4], col[4];
      e[a] = 1;
      q_mul(qm, e, tmp);
      q_mul(tmp, q12, col);
      for (int k = 0; k < 3; k++) J1[6 * k + a] = 2.0 * col[k];
    }
    /* d r_t / d delta1 = sm Rm s1 (-2 [R1 t2inv]x);  d r_t / d t1 = sm Rm */
    const double X[9] = {0, -rq[2], rq[1], rq[2], 0, -rq[0], -rq[1], rq[0], 0};
    double M[9];
    mat3_mul(Rm, X, M);
    for (int k = 0; k < 3; k++)
      for (int a = 0; a < 3; a++) {
        J1[6 * (3 + k) + a] = -2.0 * sm * s1 * M[3 * k + a];
        J1[6 * (3 + k) + 3 + a] = sm * Rm[3 * k + a];
      }
  }
  if (J2) {
    memset(J2, 0, 42 * sizeof(double));
    /* q2'^-1 = q2^-1 (1 - d^):  column a = -2 vec(dq0 * e_a), dq0 = qm q1 q2^-1 */
    for (int a = 0; a < 3; a++) {
      double e[4] = {0, 0, 0, 0}, col[4];
      e[a] = 1;
      q_mul(dq, e, col);
      for (int k = 0; k < 3; k++) J2[6 * k + a] = -2.0 * col[k];
    }
    /* t2inv' = t2inv - 2 R2^T [t2/s2]x d  =>  d r_t/d delta2 = sm Rm s1 R1 (-2 R2^T [t2/s2]x);
     * d r_t / d t2 = sm Rm s1 R1 (-R2^T / s2) */
    const double X[9] = {0, -t2s[2], t2s[1], t2s[2], 0, -t2s[0], -t2s[1], t2s[0], 0};
    double R2t[9], A[9], Bm[9], Cm[9];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) R2t[3 * i + j] = R2[3 * j + i];
    mat3_mul(Rm, R1, A);    /* Rm R1 */
    mat3_mul(A, R2t, Bm);   /* Rm R1 R2^T */
    mat3_mul(Bm, X, Cm);    /* Rm R1 R2^T [t2/s2]x */
    for (int k = 0; k < 3; k++)
      for (int a = 0; a < 3; a++) {
        J2[6 * (3 + k) + a] = -2.0 * sm * s1 * Cm[3 * k + a];
        J2[6 * (3 + k) + 3 + a] = -sm * s1 * (1.0 / s2) * Bm[3 * k + a];
      }
  }
}

typedef struct {
  int n_nodes, n_edges, fixed, nfree;
  const int32_t *e_i, *e_j;
  const double *qm, *tm, *sm; /* per edge 4, 3, 1 */
  const double *scales;       /* constant */
  int *slot;                  /* node -> free index or -1 */
  double *J, *r;              /* per edge 2 x 42, 7 */
  double *colscale;           /* 6 nfree */
  double *H, *g;              /* dense normal equations (scaled) */
} pg_ctx;

/* x = [q(4) t(3)] per node */
static int pg_linearize(void *vc, const double *x, int first, double *cost, double *gmax) {
  pg_ctx *c = (pg_ctx *)vc;
  const int n = 6 * c->nfree;
  double total = 0;
  double *cn = (double *)calloc(n, sizeof(double)), *g = (double *)calloc(n, sizeof(double));
  for (int e = 0; e < c->n_edges; e++) {
    const int a = c->e_i[e], b = c->e_j[e];
    double *J1 = c->J + (size_t)e * 84, *J2 = J1 + 42, *r = c->r + (size_t)e * 7;
    orc_pose_graph_edge(x + 7 * a, x + 7 * a + 4, c->scales[a], x + 7 * b, x + 7 * b + 4, c->scales[b], c->qm + 4 * e,
                        c->tm + 3 * e, c->sm[e], r, J1, J2);
    for (int k = 0; k < 7; k++) total += 0.5 * r[k] * r[k];
    const int sa = c->slot[a], sb = c->slot[b];
    for (int k = 0; k < 7; k++)
      for (int p = 0; p < 6; p++) {
        if (sa >= 0) g[6 * sa + p] += J1[6 * k + p] * r[k], cn[6 * sa + p] += J1[6 * k + p] * J1[6 * k + p];
        if (sb >= 0) g[6 * sb + p] += J2[6 * k + p] * r[k], cn[6 * sb + p] += J2[6 * k + p] * J2[6 * k + p];
      }
  }
  if (first)
    for (int i = 0; i < n; i++) c->colscale[i] = 1.0 / (1.0 + sqrt(cn[i]));
  double m = 0;
  for (int i = 0; i < n; i++)
    if (fabs(g[i]) > m) m = fabs(g[i]);
  /* scaled dense normal equations */
  memset(c->H, 0, sizeof(double) * (size_t)n * n);
  for (int i = 0; i < n; i++) c->g[i] = g[i] * c->colscale[i];
  for (int e = 0; e < c->n_edges; e++) {
    const int sl[2] = {c->slot[c->e_i[e]], c->slot[c->e_j[e]]};
    const double *Jb[2] = {c->J + (size_t)e * 84, c->J + (size_t)e * 84 + 42};
    for (int u = 0; u < 2; u++)
      for (int v = 0; v < 2; v++) {
        if (sl[u] < 0 || sl[v] < 0) continue;
        for (int p = 0; p < 6; p++)
          for (int q = 0; q < 6; q++) {
            double acc = 0;
            for (int k = 0; k < 7; k++) acc += Jb[u][6 * k + p] * Jb[v][6 * k + q];
            const int ip = 6 * sl[u] + p, iq = 6 * sl[v] + q;
            c->H[(size_t)ip * n + iq] += acc * c->colscale[ip] * c->colscale[iq];
          }
      }
  }
  free(cn);
  free(g);
  *cost = total;
  *gmax = m;
  return isfinite(total);
}
static int pg_step(void *vc, double radius, double *delta, double *model_change) {
  pg_ctx *c = (pg_ctx *)vc;
  const int n = 6 * c->nfree;
  double *A = (double *)malloc(sizeof(double) * (size_t)n * n), *b = (double *)malloc(sizeof(double) * n);
  memcpy(A, c->H, sizeof(double) * (size_t)n * n);
  memcpy(b, c->g, sizeof(double) * n);
  for (int i = 0; i < n; i++) {
    double d = c->H[(size_t)i * n + i];
    d = d < 1e-6 ? 1e-6 : (d > 1e32 ? 1e32 : d);
    A[(size_t)i * n + i] += d / radius;
  }
  int ok = chol_solve(A, b, n);
  if (ok)
    for (int i = 0; i < n; i++)
      if (!isfinite(b[i])) ok = 0;
  if (ok) {
    /* model cost change = -(g''.s + 0.5 s^T H'' s), s = -b */
    double gs = 0, sHs = 0;
    for (int i = 0; i < n; i++) {
      double row = 0;
      for (int j = 0; j < n; j++) row += c->H[(size_t)i * n + j] * b[j];
      sHs += b[i] * row;
      gs -= c->g[i] * b[i];
    }
    *model_change = -(gs + 0.5 * sHs);
    memset(delta, 0, sizeof(double) * 6 * c->n_nodes);
    for (int a = 0; a < c->n_nodes; a++)
      if (c->slot[a] >= 0)
        for (int p = 0; p < 6; p++) delta[6 * a + p] = -b[6 * c->slot[a] + p] * c->colscale[6 * c->slot[a] + p];
  }
  free(A);
  free(b);
  return ok;
}
static void pg_plus(void *vc, const double *x, const double *d, double *xn) {
  pg_ctx *c = (pg_ctx *)vc;
  for (int a = 0; a < c->n_nodes; a++) {
    orc_quat_plus(x + 7 * a, d + 6 * a, xn + 7 * a);
    for (int k = 0; k < 3; k++) xn[7 * a + 4 + k] = x[7 * a + 4 + k] + d[6 * a + 3 + k];
  }
}
static int pg_cost(void *vc, const double *x, double *cost) {
  pg_ctx *c = (pg_ctx *)vc;
  double total = 0;
  for (int e = 0; e < c->n_edges; e++) {
    const int a = c->e_i[e], b = c->e_j[e];
    double r[7];
    orc_pose_graph_edge(x + 7 * a, x + 7 * a + 4, c->scales[a], x + 7 * b, x + 7 * b + 4, c->scales[b], c->qm + 4 * e,
                        c->tm + 3 * e, c->sm[e], r, NULL, NULL);
    for (int k = 0; k < 7; k++) total += 0.5 * r[k] * r[k];
  }
  *cost = total;
  return isfinite(total);
}
static double pg_norm(void *vc, const double *x, const double *y) {
  pg_ctx *c = (pg_ctx *)vc;
  double s = 0;
  for (int a = 0; a < c->n_nodes; a++) {
    if (c->slot[a] < 0) continue; /* constant blocks are not part of the reduced program */
    for (int k = 0; k < 7; k++) {
      const double d = y ? x[7 * a + k] - y[7 * a + k] : x[7 * a + k];
      s += d * d;
    }
  }
  return sqrt(s);
}

/* nodes: quats (4n, x y z w), trans (3n) in/out; scales (n) constant.  Node `fixed` is constant. */
int orc_pose_graph_solve(int n_nodes, double *quats, double *trans, const double *scales, int fixed, int n_edges,
                         const int32_t *e_i, const int32_t *e_j, const double *q_meas, const double *t_meas,
                         const double *s_meas, int max_iterations, orc_lm_summary *sum) {
  pg_ctx c;
  memset(&c, 0, sizeof(c));
  c.n_nodes = n_nodes, c.n_edges = n_edges, c.fixed = fixed;
  c.e_i = e_i, c.e_j = e_j, c.qm = q_meas, c.tm = t_meas, c.sm = s_meas, c.scales = scales;
  c.slot = (int *)malloc(sizeof(int) * n_nodes);
  /* a node is in the problem when an edge touches it (parameter blocks are added by AddResidualBlock) */
  uint8_t *used = (uint8_t *)calloc(n_nodes, 1);
  for (int e = 0; e < n_edges; e++) used[e_i[e]] = used[e_j[e]] = 1;
  c.nfree = 0;
  for (int a = 0; a < n_nodes; a++) c.slot[a] = (used[a] && a != fixed) ? c.nfree++ : -1;
  const int n = 6 * c.nfree;
  c.J = (double *)malloc(sizeof(double) * 84 * (size_t)(n_edges > 0 ? n_edges : 1));
  c.r = (double *)malloc(sizeof(double) * 7 * (size_t)(n_edges > 0 ? n_edges : 1));
  c.colscale = (double *)malloc(sizeof(double) * (n > 0 ? n : 1));
  c.H = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1) * (n > 0 ? n : 1));
  c.g = (double *)malloc(sizeof(double) * (n > 0 ? n : 1));
  double *x = (double *)malloc(sizeof(double) * 7 * n_nodes);
  for (int a = 0; a < n_nodes; a++) {
    memcpy(x + 7 * a, quats + 4 * a, 4 * sizeof(double));
    memcpy(x + 7 * a + 4, trans + 3 * a, 3 * sizeof(double));
  }
  /* lm_minimize works on an ambient vector and a tangent delta of possibly different sizes: nx is
   * the larger (ambient) one */
  lm_problem P = {&c, 7 * n_nodes, pg_linearize, pg_step, pg_plus, pg_cost, pg_norm};
  if (n > 0 && n_edges > 0) lm_minimize(&P, x, max_iterations, sum);
  for (int a = 0; a < n_nodes; a++) {
    memcpy(quats + 4 * a, x + 7 * a, 4 * sizeof(double));
    memcpy(trans + 3 * a, x + 7 * a + 4, 3 * sizeof(double));
  }
  free(c.slot), free(used), free(c.J), free(c.r), free(c.colscale), free(c.H), free(c.g), free(x);
  return 0;
}
