// ba_math.h -- FP64 device/host math for the optimizer kernels: Sophus-style SE3 exp/log/compose,
// the reference's se3TransPoint and reprojection residual/Jacobians, Huber weights.
// Reference: include/myslam/optimizer_ceres.h:29-95, src/optimizer_ceres.cpp:44-154, 320-444.
#pragma once
#include <hip/hip_runtime.h>

namespace vo {
namespace ba {

#define VO_HD __host__ __device__ __forceinline__

// 1/x and 1/sqrt(x) for the per-observation arithmetic of the BA kernels: the hardware estimate refined by two Newton
// steps (error of a few 1e-16 relative, i.e. an ulp or two -- far inside every stated tolerance) in 5-7 instructions,
// against ~18 for an IEEE divide and ~25 + 18 for sqrt followed by a divide.  Host code keeps the plain operators.
VO_HD double inv_fast(double x) {
#ifdef __HIP_DEVICE_COMPILE__
  double r = __builtin_amdgcn_rcp(x);
  r = r * (2.0 - x * r);
  r = r * (2.0 - x * r);
  return r;
#else
  return 1.0 / x;
#endif
}
VO_HD double rsqrt_fast(double x) {
#ifdef __HIP_DEVICE_COMPILE__
  double y = __builtin_amdgcn_rsq(x);
  y = y * (1.5 - 0.5 * x * y * y);
  y = y * (1.5 - 0.5 * x * y * y);
  return y;
#else
  return 1.0 / sqrt(x);
#endif
}

constexpr double kSmallEps = 1e-10;          // Sophus SMALL_EPS
constexpr double kDblEps = 2.2204460492503131e-16;

struct Se3 {  // unit quaternion (w,x,y,z) + translation, like Sophus::SE3
  double q[4];
  double t[3];
};

VO_HD void quat_normalize(double q[4]) {
  const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  q[0] /= n, q[1] /= n, q[2] /= n, q[3] /= n;
}

VO_HD void quat_rotate(const double q[4], const double v[3], double o[3]) {
  double uv0 = q[2] * v[2] - q[3] * v[1], uv1 = q[3] * v[0] - q[1] * v[2], uv2 = q[1] * v[1] - q[2] * v[0];
  uv0 += uv0, uv1 += uv1, uv2 += uv2;
  o[0] = v[0] + q[0] * uv0 + (q[2] * uv2 - q[3] * uv1);
  o[1] = v[1] + q[0] * uv1 + (q[3] * uv0 - q[1] * uv2);
  o[2] = v[2] + q[0] * uv2 + (q[1] * uv1 - q[2] * uv0);
}

// sin and cos of a small argument without the range reduction of the library call: the fdlibm kernels (k_sin.c,
// k_cos.c: minimax polynomials on [-pi/4, pi/4], < 1 ulp), ~25 multiply-adds instead of ~250 instructions.  The
// arguments here are half rotation angles of LM steps and of camera poses; anything larger takes the library call.
VO_HD void sincos_small(double x, double *s, double *c) {
  if (fabs(x) > 0.78539816339744830962) {
    sincos(x, s, c);
    return;
  }
  const double z = x * x;
  const double rs = 8.33333333332248946124e-03 +
                    z * (-1.98412698298579493134e-04 +
                         z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
  *s = x + (z * x) * (-1.66666666666666324348e-01 + z * rs);
  const double rc = z * (4.16666666666666019037e-02 +
                         z * (-1.38888888888741095749e-03 +
                              z * (2.48015872894767294178e-05 +
                                   z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
  *c = 1.0 - (0.5 * z - z * rc);
}

// Sophus SE3::exp: tangent [upsilon; omega].  FAST (device, pose-only LM): polynomial sincos for small angles and
// Newton-refined reciprocals instead of IEEE divides -- an ulp or two from the plain form.
template <bool FAST = false>
VO_HD Se3 se3_exp(const double xi[6]) {
  Se3 T;
  const double wx = xi[3], wy = xi[4], wz = xi[5];
  const double th2 = wx * wx + wy * wy + wz * wz;
  const double rth = (FAST && th2 > 1e-280) ? rsqrt_fast(th2) : 0.0;  // FAST: 1 / theta, and theta = theta^2 / theta (below 1e-140: the zero rotation -- v_rsq_f64 of a denormal is not to be relied on)
  const double theta = FAST ? th2 * rth : sqrt(th2);
  const double half = 0.5 * theta;
  double sh, ch;
  if (FAST)
    sincos_small(half, &sh, &ch);
  else
    sincos(half, &sh, &ch);  // one range reduction for both; sin/cos(theta) follow from the half angle
  double imag;
  if (theta < kSmallEps) {
    const double t2 = theta * theta;
    imag = 0.5 - 0.0208333 * t2 + 0.000260417 * t2 * t2;
  } else {
    imag = FAST ? sh * rth : sh / theta;
  }
  T.q[0] = ch, T.q[1] = imag * wx, T.q[2] = imag * wy, T.q[3] = imag * wz;
  if (FAST) {
    const double rn = rsqrt_fast(T.q[0] * T.q[0] + T.q[1] * T.q[1] + T.q[2] * T.q[2] + T.q[3] * T.q[3]);
    T.q[0] *= rn, T.q[1] *= rn, T.q[2] *= rn, T.q[3] *= rn;
  } else {
    quat_normalize(T.q);
  }
  // V = I + a*Om + b*Om^2 ;  V*u = u + a (w x u) + b (w x (w x u))
  const double u[3] = {xi[0], xi[1], xi[2]};
  const double wxu[3] = {wy * u[2] - wz * u[1], wz * u[0] - wx * u[2], wx * u[1] - wy * u[0]};
  const double wwxu[3] = {wy * wxu[2] - wz * wxu[1], wz * wxu[0] - wx * wxu[2], wx * wxu[1] - wy * wxu[0]};
  if (theta < kSmallEps) {
    quat_rotate(T.q, u, T.t);  // V = R in the small-angle branch
  } else {
    const double t2 = theta * theta;
    double a, b;  // 1-cos = 2 sin^2(t/2)
    if (FAST) {
      const double it2 = rth * rth;
      a = (2.0 * sh * sh) * it2, b = (theta - 2.0 * sh * ch) * (it2 * rth);
    } else {
      a = (2.0 * sh * sh) / t2, b = (theta - 2.0 * sh * ch) / (t2 * theta);
    }
    T.t[0] = u[0] + a * wxu[0] + b * wwxu[0];
    T.t[1] = u[1] + a * wxu[1] + b * wwxu[1];
    T.t[2] = u[2] + a * wxu[2] + b * wwxu[2];
  }
  return T;
}

// Sophus SE3::log
VO_HD void se3_log(const Se3 &T, double xi[6]) {
  const double n = sqrt(T.q[1] * T.q[1] + T.q[2] * T.q[2] + T.q[3] * T.q[3]);
  const double w = T.q[0];
  double f;
  if (n < kSmallEps)
    f = 2. / w - 2. * (n * n) / (w * w * w);
  else
    f = 2 * atan(n / w) / n;
  const double theta = f * n;
  const double wx = f * T.q[1], wy = f * T.q[2], wz = f * T.q[3];
  // theta/2 = atan(n/w)  =>  tan(theta/2) = n/w: no second transcendental
  const double c = (theta < kSmallEps) ? (1. / 12.) : (1 - theta * w / (2 * n)) / (theta * theta);
  const double *t = T.t;
  const double wxt[3] = {wy * t[2] - wz * t[1], wz * t[0] - wx * t[2], wx * t[1] - wy * t[0]};
  const double wwxt[3] = {wy * wxt[2] - wz * wxt[1], wz * wxt[0] - wx * wxt[2], wx * wxt[1] - wy * wxt[0]};
  xi[0] = t[0] - 0.5 * wxt[0] + c * wwxt[0];
  xi[1] = t[1] - 0.5 * wxt[1] + c * wwxt[1];
  xi[2] = t[2] - 0.5 * wxt[2] + c * wwxt[2];
  xi[3] = wx, xi[4] = wy, xi[5] = wz;
}

// The same with the Newton-refined reciprocals of the FAST forms (pose-only LM: one log per iteration, in a dependent
// chain that nothing overlaps): 1 / n is the rsqrt that forms n, theta w / (2 n) = f w / 2.  The degenerate inputs
// (rotation below 1e-10, w = 0) take the plain function.
VO_HD void se3_log_fast(const Se3 &T, double xi[6]) {
#ifdef __HIP_DEVICE_COMPILE__
  const double n2 = T.q[1] * T.q[1] + T.q[2] * T.q[2] + T.q[3] * T.q[3];
  const double w = T.q[0];
  if (!(n2 >= 1e-18) || !(fabs(w) > 1e-8)) {
    se3_log(T, xi);
    return;
  }
  const double rn = rsqrt_fast(n2), n = n2 * rn;
  const double f = 2 * atan(n * inv_fast(w)) * rn;
  const double theta = f * n;
  const double wx = f * T.q[1], wy = f * T.q[2], wz = f * T.q[3];
  const double c = (theta < kSmallEps) ? (1. / 12.) : (1 - 0.5 * f * w) * inv_fast(theta * theta);
  const double *t = T.t;
  const double wxt[3] = {wy * t[2] - wz * t[1], wz * t[0] - wx * t[2], wx * t[1] - wy * t[0]};
  const double wwxt[3] = {wy * wxt[2] - wz * wxt[1], wz * wxt[0] - wx * wxt[2], wx * wxt[1] - wy * wxt[0]};
  xi[0] = t[0] - 0.5 * wxt[0] + c * wwxt[0];
  xi[1] = t[1] - 0.5 * wxt[1] + c * wwxt[1];
  xi[2] = t[2] - 0.5 * wxt[2] + c * wwxt[2];
  xi[3] = wx, xi[4] = wy, xi[5] = wz;
#else
  se3_log(T, xi);
#endif
}

// PoseLocalParameterization::Plus (optimizer_ceres.cpp:44-53): log(exp(delta) * exp(x)).
// The two-argument form takes exp(x) ready-made: the solve kernel forms it while the reduced
// system is still in flight, off the critical path behind the factorisation.
VO_HD void se3_plus_exp(const Se3 &B, const double d[6], double out[6]) {
  const Se3 A = se3_exp(d);
  Se3 Cc;
  double rt[3];
  quat_rotate(A.q, B.t, rt);
  Cc.t[0] = A.t[0] + rt[0], Cc.t[1] = A.t[1] + rt[1], Cc.t[2] = A.t[2] + rt[2];
  Cc.q[0] = A.q[0] * B.q[0] - A.q[1] * B.q[1] - A.q[2] * B.q[2] - A.q[3] * B.q[3];
  Cc.q[1] = A.q[0] * B.q[1] + A.q[1] * B.q[0] + A.q[2] * B.q[3] - A.q[3] * B.q[2];
  Cc.q[2] = A.q[0] * B.q[2] + A.q[2] * B.q[0] + A.q[3] * B.q[1] - A.q[1] * B.q[3];
  Cc.q[3] = A.q[0] * B.q[3] + A.q[3] * B.q[0] + A.q[1] * B.q[2] - A.q[2] * B.q[1];
  quat_normalize(Cc.q);
  se3_log(Cc, out);
}
VO_HD void se3_plus(const double x[6], const double d[6], double out[6]) { se3_plus_exp(se3_exp(x), d, out); }
// The same with exp(delta) * B handed back as well (C) and the FAST forms: the pose-only LM keeps exp(x) across its
// iterations -- the accepted candidate's C is the next B -- and evaluates residuals from C directly (pose_cache_se3).
VO_HD void se3_plus_keep(const Se3 &B, const double d[6], double out[6], Se3 &C) {
  const Se3 A = se3_exp<true>(d);
  double rt[3];
  quat_rotate(A.q, B.t, rt);
  C.t[0] = A.t[0] + rt[0], C.t[1] = A.t[1] + rt[1], C.t[2] = A.t[2] + rt[2];
  C.q[0] = A.q[0] * B.q[0] - A.q[1] * B.q[1] - A.q[2] * B.q[2] - A.q[3] * B.q[3];
  C.q[1] = A.q[0] * B.q[1] + A.q[1] * B.q[0] + A.q[2] * B.q[3] - A.q[3] * B.q[2];
  C.q[2] = A.q[0] * B.q[2] + A.q[2] * B.q[0] + A.q[3] * B.q[1] - A.q[1] * B.q[3];
  C.q[3] = A.q[0] * B.q[3] + A.q[3] * B.q[0] + A.q[1] * B.q[2] - A.q[2] * B.q[1];
  const double rn = rsqrt_fast(C.q[0] * C.q[0] + C.q[1] * C.q[1] + C.q[2] * C.q[2] + C.q[3] * C.q[3]);
  C.q[0] *= rn, C.q[1] *= rn, C.q[2] *= rn, C.q[3] *= rn;
  se3_log_fast(C, out);
}

// Optimizer::se3TransPoint<double> (optimizer_ceres.h:29-95) plus the rotation matrix of
// ceres::AngleAxisToRotationMatrix (same theta^2 > eps branch), R row-major here.
struct PoseCache {
  double R[9];  // row-major rotation
  double t[3];  // V * upsilon
};

VO_HD PoseCache pose_cache(const double se3[6]) {
  PoseCache P;
  const double a0 = se3[3], a1 = se3[4], a2 = se3[5];
  const double u0 = se3[0], u1 = se3[1], u2 = se3[2];
  const double theta2 = a0 * a0 + a1 * a1 + a2 * a2;
  if (theta2 > kDblEps) {
    const double theta = sqrt(theta2);
    double s, c;
    sincos(theta, &s, &c);
    const double wx = a0 / theta, wy = a1 / theta, wz = a2 / theta;
    P.R[0] = c + wx * wx * (1.0 - c);
    P.R[3] = wz * s + wx * wy * (1.0 - c);
    P.R[6] = -wy * s + wx * wz * (1.0 - c);
    P.R[1] = wx * wy * (1.0 - c) - wz * s;
    P.R[4] = c + wy * wy * (1.0 - c);
    P.R[7] = wx * s + wy * wz * (1.0 - c);
    P.R[2] = wy * s + wx * wz * (1.0 - c);
    P.R[5] = -wx * s + wy * wz * (1.0 - c);
    P.R[8] = c + wz * wz * (1.0 - c);
    const double wxu0 = wy * u2 - wz * u1, wxu1 = wz * u0 - wx * u2, wxu2 = wx * u1 - wy * u0;
    const double ww0 = wy * wxu2 - wz * wxu1, ww1 = wz * wxu0 - wx * wxu2, ww2 = wx * wxu1 - wy * wxu0;
    const double A = (1.0 - c) / theta, B = (theta - s) / theta;
    P.t[0] = u0 + A * wxu0 + B * ww0;
    P.t[1] = u1 + A * wxu1 + B * ww1;
    P.t[2] = u2 + A * wxu2 + B * ww2;
  } else {
    P.R[0] = 1, P.R[1] = -a2, P.R[2] = a1;
    P.R[3] = a2, P.R[4] = 1, P.R[5] = -a0;
    P.R[6] = -a1, P.R[7] = a0, P.R[8] = 1;
    P.t[0] = u0 + (a1 * u2 - a2 * u1);
    P.t[1] = u1 + (a2 * u0 - a0 * u2);
    P.t[2] = u2 + (a0 * u1 - a1 * u0);
  }
  return P;
}

// The same quantities from T = exp(se3) (unit quaternion, t = V * upsilon): no trigonometry.  Equal to
// pose_cache(log(T)) up to rounding (a few 1e-16).
VO_HD PoseCache pose_cache_se3(const Se3 &T) {
  PoseCache P;
  const double w = T.q[0], x = T.q[1], y = T.q[2], z = T.q[3];
  const double xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
  P.R[0] = 1.0 - 2.0 * (yy + zz), P.R[1] = 2.0 * (xy - wz), P.R[2] = 2.0 * (xz + wy);
  P.R[3] = 2.0 * (xy + wz), P.R[4] = 1.0 - 2.0 * (xx + zz), P.R[5] = 2.0 * (yz - wx);
  P.R[6] = 2.0 * (xz - wy), P.R[7] = 2.0 * (yz + wx), P.R[8] = 1.0 - 2.0 * (xx + yy);
  P.t[0] = T.t[0], P.t[1] = T.t[1], P.t[2] = T.t[2];
  return P;
}

VO_HD void trans_point(const PoseCache &P, const double p[3], double o[3]) {
  o[0] = P.R[0] * p[0] + P.R[1] * p[1] + P.R[2] * p[2] + P.t[0];
  o[1] = P.R[3] * p[0] + P.R[4] * p[1] + P.R[5] * p[2] + P.t[1];
  o[2] = P.R[6] * p[0] + P.R[7] * p[1] + P.R[8] * p[2] + P.t[2];
}

struct Cam {
  double fx, fy, cx, cy, bf;
};

// residual (with inv_sigma) and the reference's Jacobians (without it: Q-B1).
// rows: 2 mono (uR < 0) / 3 stereo.  Jp row-major rows x 6, Jl rows x 3.
template <bool WANT_JP, bool WANT_JL>
VO_HD int edge_eval(const PoseCache &P, const double pt[3], double ou, double ov, double our, double inv_sigma,
                    const Cam &K, double r[3], double *Jp, double *Jl) {
  double pc[3];
  trans_point(P, pt, pc);
  const double x = pc[0], y = pc[1], z = pc[2];
  const double invz = inv_fast(z), invz2 = invz * invz;
  const bool stereo = !(our < 0);
  const double uhat = K.fx * x * invz + K.cx;
  r[0] = (ou - uhat) * inv_sigma;
  r[1] = (ov - (K.fy * y * invz + K.cy)) * inv_sigma;
  r[2] = stereo ? (our - (uhat - K.bf * invz)) * inv_sigma : 0.0;
  if (WANT_JP) {
    Jp[0] = -invz * K.fx, Jp[1] = 0, Jp[2] = x * invz2 * K.fx;
    Jp[3] = x * y * invz2 * K.fx, Jp[4] = -(1 + (x * x * invz2)) * K.fx, Jp[5] = y * invz * K.fx;
    Jp[6] = 0, Jp[7] = -invz * K.fy, Jp[8] = y * invz2 * K.fy;
    Jp[9] = (1 + y * y * invz2) * K.fy, Jp[10] = -x * y * invz2 * K.fy, Jp[11] = -x * invz * K.fy;
    if (stereo) {
      Jp[12] = Jp[0], Jp[13] = 0, Jp[14] = Jp[2] - K.bf * invz2;
      Jp[15] = Jp[3] - K.bf * y * invz2, Jp[16] = Jp[4] + K.bf * x * invz2, Jp[17] = Jp[5];
    } else {
      Jp[12] = Jp[13] = Jp[14] = Jp[15] = Jp[16] = Jp[17] = 0;
    }
  }
  if (WANT_JL) {
    // column-major R of the reference indexed R[0],R[3],R[6] = first ROW of the rotation
    Jl[0] = -K.fx * P.R[0] * invz + K.fx * x * P.R[6] * invz2;
    Jl[1] = -K.fx * P.R[1] * invz + K.fx * x * P.R[7] * invz2;
    Jl[2] = -K.fx * P.R[2] * invz + K.fx * x * P.R[8] * invz2;
    Jl[3] = -K.fy * P.R[3] * invz + K.fy * y * P.R[6] * invz2;
    Jl[4] = -K.fy * P.R[4] * invz + K.fy * y * P.R[7] * invz2;
    Jl[5] = -K.fy * P.R[5] * invz + K.fy * y * P.R[8] * invz2;
    if (stereo) {
      Jl[6] = Jl[0] - K.bf * P.R[6] * invz2;
      Jl[7] = Jl[1] - K.bf * P.R[7] * invz2;
      Jl[8] = Jl[2] - K.bf * P.R[8] * invz2;
    } else {
      Jl[6] = Jl[7] = Jl[8] = 0;
    }
  }
  return stereo ? 3 : 2;
}

// ceres::HuberLoss: rho(s) and rho'(s); a <= 0 disables the loss.  Corrector: rho'' <= 0 for Huber,
// so residual and Jacobian are scaled by sqrt(rho') only.
VO_HD void huber(double a, double s, double &rho0, double &rho1) {
  const double b = a * a;
  if (a > 0 && s > b) {
    const double ri = rsqrt_fast(s), r = s * ri;
    rho0 = 2 * a * r - b;
    rho1 = a * ri;
    if (rho1 < 2.2250738585072014e-308) rho1 = 2.2250738585072014e-308;
  } else {
    rho0 = s;
    rho1 = 1;
  }
}

// 3x3 SPD inverse from the packed upper triangle h = {00,01,02,11,12,22}; returns packed inverse.
VO_HD bool inv3_sym(const double h[6], double o[6]) {
  // Cholesky h = L L^T
  const double l00 = sqrt(h[0]);
  if (!(h[0] > 0)) return false;
  const double l10 = h[1] / l00, l20 = h[2] / l00;
  const double d1 = h[3] - l10 * l10;
  if (!(d1 > 0)) return false;
  const double l11 = sqrt(d1);
  const double l21 = (h[4] - l20 * l10) / l11;
  const double d2 = h[5] - l20 * l20 - l21 * l21;
  if (!(d2 > 0)) return false;
  const double l22 = sqrt(d2);
  // inverse of L (lower)
  const double i00 = 1.0 / l00, i11 = 1.0 / l11, i22 = 1.0 / l22;
  const double i10 = -l10 * i00 * i11;
  const double i21 = -l21 * i11 * i22;
  const double i20 = -(l20 * i00 + l21 * i10) * i22;
  // H^-1 = L^-T L^-1
  o[0] = i00 * i00 + i10 * i10 + i20 * i20;
  o[1] = i10 * i11 + i20 * i21;
  o[2] = i20 * i22;
  o[3] = i11 * i11 + i21 * i21;
  o[4] = i21 * i22;
  o[5] = i22 * i22;
  return true;
}


// ---- Sim3 residual blocks of the loop-closure optimisation (reference optimizer_ceres.h:211-267 over
// IntrinsicProjectionUV, optimizer_ceres.cpp:8-42).  x = [angle-axis(3); t(3); s].
struct Sim3Frame {  // quantities of x shared by all matches
  double R[9];      // row-major rotation (ceres::AngleAxisToRotationMatrix, same theta^2 > eps branch)
  double Jr[9], Jl[9];  // right / left Jacobian of SO(3):  d(R p)/dw = -R [p]x Jr,  d(R^T v)/dw = R^T [v]x Jl
  double t[3], s;
};

VO_HD void mat3_mul_rm(const double A[9], const double B[9], double C[9]) {
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

VO_HD Sim3Frame sim3_frame(const double x[7], bool want_jac) {
  Sim3Frame F;
  const double se3[6] = {0, 0, 0, x[0], x[1], x[2]};
  const PoseCache P = pose_cache(se3);
#pragma unroll
  for (int i = 0; i < 9; i++) F.R[i] = P.R[i];
  F.t[0] = x[3], F.t[1] = x[4], F.t[2] = x[5], F.s = x[6];
  if (want_jac) {
    const double w0 = x[0], w1 = x[1], w2 = x[2];
    const double t2 = w0 * w0 + w1 * w1 + w2 * w2;
    const double W[9] = {0, -w2, w1, w2, 0, -w0, -w1, w0, 0};
    double W2[9];
    mat3_mul_rm(W, W, W2);
    double a = 0, b = 0;
    if (t2 > kDblEps) {
      const double th = sqrt(t2);
      a = (1.0 - cos(th)) / t2;
      b = (th - sin(th)) / (t2 * th);
    }
#pragma unroll
    for (int i = 0; i < 9; i++) {
      const double id = (i % 4 == 0) ? 1.0 : 0.0;
      F.Jr[i] = id - a * W[i] + b * W2[i];
      F.Jl[i] = id + a * W[i] + b * W2[i];
    }
  }
  return F;
}

// IntrinsicProjectionUV::Evaluate: residual with 1/sigma, Jacobian without it
template <bool WANT_J>
VO_HD void intrinsic_uv(const double p[3], double pu, double pv, const double cam[4], double isig, double r[2],
                        double J[6]) {
  const double x = p[0], y = p[1], z = p[2];
  const double invz = 1.0 / z, invz2 = invz * invz;
  r[0] = (pu - (cam[0] * x * invz + cam[2])) * isig;
  r[1] = (pv - (cam[1] * y * invz + cam[3])) * isig;
  if (WANT_J) {
    J[0] = -invz * cam[0], J[1] = 0, J[2] = x * invz2 * cam[0];
    J[3] = 0, J[4] = -invz * cam[1], J[5] = y * invz2 * cam[1];
  }
}

// r = [forward(2); inverse(2)], J = 4 x 7 row-major (column 6 = scale)
template <bool WANT_J>
VO_HD void sim3_eval(const Sim3Frame &F, const double Pm[3], double pcu, double pcv, double isc, const double Pc[3],
                     double pmu, double pmv, double ism, const double cam[4], double r[4], double *J) {
  const double *R = F.R;
  const double s = F.s;
  const double Rp[3] = {R[0] * Pm[0] + R[1] * Pm[1] + R[2] * Pm[2], R[3] * Pm[0] + R[4] * Pm[1] + R[5] * Pm[2],
                        R[6] * Pm[0] + R[7] * Pm[1] + R[8] * Pm[2]};
  const double p[3] = {s * Rp[0] + F.t[0], s * Rp[1] + F.t[1], s * Rp[2] + F.t[2]};
  double Juv[6];
  intrinsic_uv<WANT_J>(p, pcu, pcv, cam, isc, r, Juv);
  if (WANT_J) {
    const double Px[9] = {0, -Pm[2], Pm[1], Pm[2], 0, -Pm[0], -Pm[1], Pm[0], 0};
    double RP[9], D[9];
    mat3_mul_rm(R, Px, RP);
    mat3_mul_rm(RP, F.Jr, D);
#pragma unroll
    for (int k = 0; k < 2; k++) {
#pragma unroll
      for (int a = 0; a < 3; a++)
        J[7 * k + a] = -s * (Juv[3 * k] * D[a] + Juv[3 * k + 1] * D[3 + a] + Juv[3 * k + 2] * D[6 + a]);
#pragma unroll
      for (int a = 0; a < 3; a++) J[7 * k + 3 + a] = Juv[3 * k + a];
      J[7 * k + 6] = Juv[3 * k] * Rp[0] + Juv[3 * k + 1] * Rp[1] + Juv[3 * k + 2] * Rp[2];
    }
  }
  const double v[3] = {(Pc[0] - F.t[0]) / s, (Pc[1] - F.t[1]) / s, (Pc[2] - F.t[2]) / s};
  const double q[3] = {R[0] * v[0] + R[3] * v[1] + R[6] * v[2], R[1] * v[0] + R[4] * v[1] + R[7] * v[2],
                       R[2] * v[0] + R[5] * v[1] + R[8] * v[2]};  // R^T v
  intrinsic_uv<WANT_J>(q, pmu, pmv, cam, ism, r + 2, Juv);
  if (WANT_J) {
    const double V[9] = {0, -v[2], v[1], v[2], 0, -v[0], -v[1], v[0], 0};
    const double Rt[9] = {R[0], R[3], R[6], R[1], R[4], R[7], R[2], R[5], R[8]};
    double VJ[9], D[9];
    mat3_mul_rm(V, F.Jl, VJ);
    mat3_mul_rm(Rt, VJ, D);
#pragma unroll
    for (int k = 0; k < 2; k++) {
      double *Jk = J + 14 + 7 * k;
#pragma unroll
      for (int a = 0; a < 3; a++) Jk[a] = Juv[3 * k] * D[a] + Juv[3 * k + 1] * D[3 + a] + Juv[3 * k + 2] * D[6 + a];
#pragma unroll
      for (int a = 0; a < 3; a++)
        Jk[3 + a] = -(Juv[3 * k] * Rt[a] + Juv[3 * k + 1] * Rt[3 + a] + Juv[3 * k + 2] * Rt[6 + a]) / s;
      Jk[6] = -(Juv[3 * k] * q[0] + Juv[3 * k + 1] * q[1] + Juv[3 * k + 2] * q[2]) / s;
    }
  }
}

// chi2 > 10 tests of optimizer_ceres.cpp:916-948 / :996-1022 (double; camera2pixel = fx x / z + cx)
VO_HD bool sim3_outlier(const Sim3Frame &F, const double Pm[3], double pcu, double pcv, double isc, const double Pc[3],
                        double pmu, double pmv, double ism, const double cam[4]) {
  const double *R = F.R;
  const double s = F.s;
  double p[3];
#pragma unroll
  for (int i = 0; i < 3; i++) p[i] = s * (R[3 * i] * Pm[0] + R[3 * i + 1] * Pm[1] + R[3 * i + 2] * Pm[2]) + F.t[i];
  const double eu = cam[0] * p[0] / p[2] + cam[2] - pcu, ev = cam[1] * p[1] / p[2] + cam[3] - pcv;
  if ((eu * eu + ev * ev) * isc * isc > 10.0) return true;
  double q[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const double Rtp = R[i] * Pc[0] + R[3 + i] * Pc[1] + R[6 + i] * Pc[2];
    const double Rtt = R[i] * F.t[0] + R[3 + i] * F.t[1] + R[6 + i] * F.t[2];
    q[i] = Rtp / s - Rtt / s;
  }
  const double fu = cam[0] * q[0] / q[2] + cam[2] - pmu, fv = cam[1] * q[1] / q[2] + cam[3] - pmv;
  return (fu * fu + fv * fv) * ism * ism > 10.0;
}

}  // namespace ba
}  // namespace vo
