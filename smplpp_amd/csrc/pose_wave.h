// The pose step of ONE frame by ONE 64-lane wavefront (fp16x2 form): Rodrigues x24 (/root/reference/src/BlendShape.cpp:803-844),
// pose coefficients vec(R)[9:] - vec(I)[9:] (:865-895), joints = J0 + JS.beta (src/JointRegression.cpp:583-598, folded), FK
// chain over the kinematic tree and relative transforms (src/WorldTransformation.cpp:421-677); writes the fused kernel's
// operands A2h / G2h (common.h) and, when asked, G' / joints / rotations / 4x4 transforms in fp32.
//
// No workgroup barrier and a single global round trip for the inputs: everything a lane will need (its theta, its folded
// regressor row, its chain slots, its K-order chunk) is requested up front; the phases then hand over through a per-wavefront
// LDS scratch (LDS traffic of one wavefront executes in order: s_waitcnt lgkmcnt(0) is the only synchronisation) and the
// chain passes the parent's row from lane to lane (ds_bpermute).  Used by pose_kernel_w (fk.hip: four frames per workgroup)
// and by the cooperative prologue of skin_kernel_h (skin_h.hip).  Same arithmetic, same order as pose_kernel: pose_math.h.
#pragma once

#include "common.h"
#include "pose_math.h"

namespace smplpp_hip
{
constexpr int PW_SCR_FLOATS = NJ * 9 + NJ * 3 + NJ * 12 + 224 + 4; // sR | sJ | sG | sCoef | zero3: 3,616 bytes per wavefront
struct PoseWaveOut
{
  float * Gp;         // [n][24][12] or null
  float * joints;     // [n][24][3] or null
  float * rot;        // [n][24][9] or null
  float * xf44;       // [n][24][16] or null
  _Float16 * A2h;     // fragment-order operands of skin_kernel_h (whole 64-frame tiles), or null
  _Float16 * G2h;
};

__device__ __forceinline__ void pw_sync()
{
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// f: the frame (wave-uniform), lane: 0..63, scr: this wavefront's LDS scratch (PW_SCR_FLOATS floats, 16-byte aligned).
// ctab: the (level, slot) -> (joint, parent, parent's slot) table of a tree with at most CT_LEV levels of at most 5 joints
// (smplpp_model::chain_fast); kmap: K order / blend slot order of the fp16x2 form (smplpp_model::kmap).
__device__ __forceinline__ void pose_frame_wave(const int lane, const int64_t f, const float * __restrict__ beta,
                                                const float * __restrict__ theta, const float * __restrict__ J0,
                                                const float * __restrict__ JS, const int32_t * __restrict__ ctab, const int nlev,
                                                const int16_t * __restrict__ kmap, const float gscale, float * scr,
                                                const PoseWaveOut & o)
{
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  typedef short s8 __attribute__((ext_vector_type(8)));
  float(*sR)[9] = reinterpret_cast<float(*)[9]>(scr);
  float(*sJ)[3] = reinterpret_cast<float(*)[3]>(scr + NJ * 9);
  float(*sG)[12] = reinterpret_cast<float(*)[12]>(scr + NJ * 9 + NJ * 3);
  float * sCoef = scr + NJ * 9 + NJ * 3 + NJ * 12;
  float * sZero = sCoef + 224;

  // ---- everything from global memory, requested together
  float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;
  if(lane < NJ)
  {
    const float * th = theta + (f * (NJ + 1) + 1 + lane) * 3; // theta[:,1:,:] (src/SMPL.cpp:685-686)
    t0 = th[0];
    t1 = th[1];
    t2 = th[2];
  }
  float be[NB]; // (uniform address: scalar loads)
#pragma unroll
  for(int k = 0; k < NB; k++) be[k] = beta ? beta[f * NB + k] : 0.0f;
  float j0a = J0[lane], jsa[NB], j0b = 0.0f, jsb[NB]; // joint coordinates lane and 64 + lane (the latter: 8 lanes)
#pragma unroll
  for(int k = 0; k < NB; k++) jsa[k] = JS[lane * NB + k];
#pragma unroll
  for(int k = 0; k < NB; k++) jsb[k] = 0.0f;
  if(lane < NJ * 3 - 64)
  {
    j0b = J0[64 + lane];
#pragma unroll
    for(int k = 0; k < NB; k++) jsb[k] = JS[(64 + lane) * NB + k];
  }
  int cti[CT_LEV], ctp[CT_LEV], cts[CT_LEV];
#pragma unroll
  for(int L = 0; L < CT_LEV; L++)
  {
    cti[L] = ctp[L] = -1;
    cts[L] = 0;
  }
  if(lane < 60)
  {
    const int slot = lane / 12;
#pragma unroll
    for(int L = 0; L < CT_LEV; L++)
      if(L < nlev)
      {
        cti[L] = ctab[(L * 5 + slot) * 3 + 0];
        ctp[L] = ctab[(L * 5 + slot) * 3 + 1];
        cts[L] = ctab[(L * 5 + slot) * 3 + 2];
      }
  }
  s8 km = {0, 0, 0, 0, 0, 0, 0, 0}; // this lane's chunk of the K order (lanes 0..27), of the blend slot order (lanes 28..30)
  if(lane < 31) km = *reinterpret_cast<const s8 *>(kmap + 8 * lane);

  // ---- Rodrigues, coefficients (the reference's k order in the scratch), joints
  if(lane < 4) sZero[lane] = 0.0f;
  if(lane < NJ)
  {
    float R[9];
    rodrigues9(t0, t1, t2, R);
#pragma unroll
    for(int q = 0; q < 9; q++) sR[lane][q] = R[q];
    if(lane >= 1) // root joint has no pose corrective (src/BlendShape.cpp:884-887)
#pragma unroll
      for(int q = 0; q < 9; q++) sCoef[9 * (lane - 1) + q] = R[q] - ((q == 0 || q == 4 || q == 8) ? 1.0f : 0.0f);
    if(o.rot)
#pragma unroll
      for(int q = 0; q < 9; q++) o.rot[(f * NJ + lane) * 9 + q] = R[q];
  }
  else if(lane < NJ + 17) // [beta(10) | 1 | 0 x 6] behind the 207 coefficients
  {
    const int i = lane - NJ;
    float v = 0.0f;
#pragma unroll
    for(int k = 0; k < NB; k++) v = (i == k) ? be[k] : v;
    if(i == NB) v = 1.0f;
    sCoef[NP + i] = v;
  }
  {
    const float s = joint_coord(j0a, jsa, be);
    sJ[0][lane] = s;
    if(o.joints) o.joints[f * NJ * 3 + lane] = s;
    if(lane < NJ * 3 - 64)
    {
      const float s2 = joint_coord(j0b, jsb, be);
      sJ[0][64 + lane] = s2;
      if(o.joints) o.joints[f * NJ * 3 + 64 + lane] = s2;
    }
  }
  pw_sync();

  // ---- A operand chunk (lanes 0..27): 64 . [c | beta | 1 | 0] in the K order of the fp16x2 form, both pieces
  const int64_t ft = f >> 6;
  const int fh = (int)((f >> 5) & 1), r = (int)(f & 31);
  if(o.A2h && lane < 28)
  {
    const int c = lane, ks = c >> 1, h = c & 1;
    h8 hi, lo;
#pragma unroll
    for(int j = 0; j < 8; j++)
    {
      const int k = km[j]; // -1 = zero padding
      _Float16 a, b;
      split_f16x2((k >= 0 ? sCoef[k] : 0.0f) * HB_SA, a, b);
      hi[j] = a;
      lo[j] = b;
    }
    _Float16 * dst = o.A2h + ((((ft * HB_KS + ks) * 2 + fh) * 2) * 64 + (32 * h + r)) * 8;
    *reinterpret_cast<h8 *>(dst) = hi;
    *reinterpret_cast<h8 *>(dst + 64 * 8) = lo;
  }

  // ---- chain: G_0 = L_0, G_i = G_p(i) . L_i with L_i = [R_i | j_i - j_p(i)] (src/WorldTransformation.cpp:508-610), one tree
  // level per step; lane = (slot of the level, entry of the 3x4): the operand of every level (a column of R_i, or the
  // offset j_i - j_p) does not depend on the chain and is fetched up front, the parent's row comes out of the registers of
  // the lanes that computed it one level earlier
  if(lane < 60)
  {
    const int e = lane % 12, rr = e / 4, c = e % 4;
    float x0[CT_LEV], x1[CT_LEV], x2[CT_LEV];
#pragma unroll
    for(int L = 0; L < CT_LEV; L++)
    {
      const int i = cti[L] >= 0 ? cti[L] : 0, p = ctp[L];
      const float * a = (c < 3) ? &sR[i][c] : &sJ[i][0]; // column c of R_i (stride 3) or j_i (stride 1)
      const int st = (c < 3) ? 3 : 1;
      const float * b = (c == 3 && p >= 0) ? &sJ[p][0] : sZero;
      const float a0 = a[0], a1 = a[st], a2 = a[2 * st], b0 = b[0], b1 = b[1], b2 = b[2];
      x0[L] = a0 - b0;
      x1[L] = a1 - b1;
      x2[L] = a2 - b2;
      if(p < 0) x0[L] = (rr == 0) ? x0[L] : (rr == 1 ? x1[L] : x2[L]); // root: L_0 = [R_0 | j_0], entry (r, c) itself
    }
    float vprev = 0.0f;
#pragma unroll
    for(int L = 0; L < CT_LEV; L++)
    {
      if(L < nlev) // (wave-uniform)
      {
        const int i = cti[L], p = ctp[L], src = 12 * cts[L] + rr * 4;
        const float g0 = __shfl(vprev, src + 0, 64), g1 = __shfl(vprev, src + 1, 64), g2 = __shfl(vprev, src + 2, 64),
                    g3 = __shfl(vprev, src + 3, 64);
        float v = x0[L];
        if(p >= 0) v = chain_entry(g0, g1, g2, g3, x0[L], x1[L], x2[L], c == 3);
        if(i >= 0)
        {
          vprev = v;
          sG[i][e] = v;
        }
      }
    }
  }
  pw_sync();

  // ---- relative transforms: translation -= A_i . j_i (src/WorldTransformation.cpp:657-677); fp32 outputs when asked
  if(o.Gp || o.xf44)
  {
    for(int e = lane; e < NJ * 12; e += 64)
    {
      const int i = e / 12, q = e % 12, rr = q / 4, c = q % 4;
      float v = sG[i][q];
      if(c == 3) v = relative_t(v, sG[i][rr * 4 + 0], sG[i][rr * 4 + 1], sG[i][rr * 4 + 2], sJ[i][0], sJ[i][1], sJ[i][2]);
      if(o.Gp) o.Gp[(f * NJ + i) * 12 + q] = v;
      if(o.xf44) o.xf44[(f * NJ + i) * 16 + q] = v;
    }
    if(o.xf44)
      for(int e = lane; e < NJ * 4; e += 64) o.xf44[(f * NJ + e / 4) * 16 + 12 + e % 4] = (e % 4 == 3) ? 1.0f : 0.0f;
  }
  // the same transforms as the A operand of the blend MFMAs (rows = frames, k = blend slot): lane (entry e, chunk c) writes
  // both fp16x2 pieces of slots 8 c .. 8 c + 7 of entry e
  if(o.G2h && lane < 36)
  {
    const int e = lane / 3, c = lane % 3, r4 = e / 4, cc = e % 4;
    // slot -> joint of this chunk: lanes 28..30 hold kmap[224 + 8 c ..]
    h8 hi, lo;
#pragma unroll
    for(int j = 0; j < 8; j++)
    {
      const int i = __shfl((int)km[j], 28 + c, 64);
      float v = sG[i][e];
      if(cc == 3) v = relative_t(v, sG[i][r4 * 4 + 0], sG[i][r4 * 4 + 1], sG[i][r4 * 4 + 2], sJ[i][0], sJ[i][1], sJ[i][2]);
      _Float16 a, b;
      split_f16x2(v * gscale, a, b);
      hi[j] = a;
      lo[j] = b;
    }
    _Float16 * blk = o.G2h + (((ft * 2 + fh) * 12 + e) * 3072) / 2;
    if(c < 2)
    {
      *reinterpret_cast<h8 *>(blk + (32 * c + r) * 8) = hi;
      *reinterpret_cast<h8 *>(blk + 512 + (32 * c + r) * 8) = lo;
    }
    else
    {
      *reinterpret_cast<h8 *>(blk + 1024 + r * 8) = hi;
      *reinterpret_cast<h8 *>(blk + 1024 + 256 + r * 8) = lo;
    }
  }
}
} // namespace smplpp_hip
