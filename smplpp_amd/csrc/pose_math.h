// The arithmetic of the pose step, written once with its operation order fixed (no floating-point contraction left to the
// compiler), so that pose_kernel (fk.hip) and the in-kernel pose of skin_kernel_h (skin_h.hip) produce the same bits from the
// same inputs.  Reference: Rodrigues /root/reference/src/BlendShape.cpp:803-844, joints src/JointRegression.cpp:583-598
// (through the folded regressor), chain and relative transforms src/WorldTransformation.cpp:508-677.
#pragma once

#include "common.h"

namespace smplpp_hip
{
// R = I + K sin(a) + K.K (1 - cos a), K = skew(theta / a), a = ||theta + 1e-8|| (eps per component, axis from the raw theta:
// src/BlendShape.cpp:813-815).  K.K is the reference's matmul with its structural zeros dropped (0.x terms are exact):
// (K.K)_rr = -(k_s^2 + k_t^2), (K.K)_rc = k_r k_c.
__device__ __forceinline__ void rodrigues9(float t0, float t1, float t2, float (&R)[9])
{
#pragma clang fp contract(off)
  const float eps = 1e-8f;
  const float a0 = t0 + eps, a1 = t1 + eps, a2 = t2 + eps;
  const float angle = sqrtf(__builtin_fmaf(a2, a2, __builtin_fmaf(a1, a1, a0 * a0)));
  const float k0 = t0 / angle, k1 = t1 / angle, k2 = t2 / angle;
  float s, co;
  sincosf(angle, &s, &co); // (one range reduction for both)
  const float c1 = 1.0f - co;
  const float q11 = k1 * k1, q22 = k2 * k2;
  const float d0 = -__builtin_fmaf(k1, k1, q22), d1 = -__builtin_fmaf(k0, k0, q22), d2 = -__builtin_fmaf(k0, k0, q11);
  const float p01 = k0 * k1, p02 = k0 * k2, p12 = k1 * k2;
  const float s0 = k0 * s, s1 = k1 * s, s2 = k2 * s;
  R[0] = __builtin_fmaf(d0, c1, 1.0f);
  R[1] = __builtin_fmaf(p01, c1, -s2);
  R[2] = __builtin_fmaf(p02, c1, s1);
  R[3] = __builtin_fmaf(p01, c1, s2);
  R[4] = __builtin_fmaf(d1, c1, 1.0f);
  R[5] = __builtin_fmaf(p12, c1, -s0);
  R[6] = __builtin_fmaf(p02, c1, -s1);
  R[7] = __builtin_fmaf(p12, c1, s0);
  R[8] = __builtin_fmaf(d2, c1, 1.0f);
}

// one joint coordinate through the folded regressor: J0 + sum_k JS[k] beta[k], ascending k
template<class JS, class BETA>
__device__ __forceinline__ float joint_coord(float j0, const JS & js, const BETA & beta)
{
  float s = j0;
#pragma unroll
  for(int k = 0; k < NB; k++) s = __builtin_fmaf(js[k], beta[k], s);
  return s;
}

// entry (r, c) of G_p . [R_i | t_i]: g = row r of G_p (g3 its translation), x = column c of [R_i | t_i]; c == 3 adds g3
__device__ __forceinline__ float chain_entry(float g0, float g1, float g2, float g3, float x0, float x1, float x2, bool is_t)
{
#pragma clang fp contract(off)
  float v = __builtin_fmaf(g2, x2, __builtin_fmaf(g1, x1, g0 * x0));
  if(is_t) v = v + g3;
  return v;
}

// translation of the relative transform: t - A . j (src/WorldTransformation.cpp:657-677)
__device__ __forceinline__ float relative_t(float t, float a0, float a1, float a2, float j0, float j1, float j2)
{
#pragma clang fp contract(off)
  return t - __builtin_fmaf(a2, j2, __builtin_fmaf(a1, j1, a0 * j0));
}
} // namespace smplpp_hip
