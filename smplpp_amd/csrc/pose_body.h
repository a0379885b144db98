// The pose step of one frame (one 256-thread workgroup, four wavefronts) as a header: pose_kernel (fk.hip) is a wrapper around it,
// and any kernel that includes it executes the same instructions, so the same bits.  (Round 3 ran it as the tail of the IK solve
// kernel for the frame just updated: bit-identical, no gain — DESIGN.md §8 — and not kept.)  References: see fk.hip / pose_math.h.
#pragma once

#include "common.h"
#include "pose_math.h"

namespace smplpp_hip
{
typedef _Float16 pose_f16x8 __attribute__((ext_vector_type(8)));

struct PoseArgs
{
  const float * beta;       // [n][10] (nullable: zeros)
  const float * theta;      // [n][25][3]
  const float *J0, *JS, *JSp;
  const int32_t *parent, *lvl_off, *lvl_joint;
  int nlev;
  float * AT;               // K-major fp32 operand of the fp32 forms (nullable)
  int64_t ldA;
  float *Gp, *joints_out, *rot_out, *xf44_out;
  int64_t n;
  uint16_t * A3;            // bf16x3 operand (nullable)
  _Float16 *A2h, *G2h;      // fp16x2 operands (nullable)
  float gscale;
  const int32_t * ctab;     // chain table of the fast path (nullable)
  int * range_flag;
};

// LDS traffic of one wavefront is executed in order, so the phases of a single-wavefront section only need the compiler
// to keep that order (no s_barrier, and no vmcnt(0) drain of outstanding global stores as __syncthreads() would add).
__device__ __forceinline__ void wave_sync()
{
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// Workgroup barrier that orders LDS traffic only: __syncthreads() also waits for vmcnt(0), i.e. for every global store
// issued so far (A operand, rotations, joints) to reach L2 — microseconds per phase in a kernel that is pure latency.
__device__ __forceinline__ void block_sync_lds()
{
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// grid = n frames, block = 256 (four wavefronts per frame: the kernel is a chain of dependent latencies, so the work of a
// frame is spread over as many lanes as it has independent pieces).
// Trees with at most CT_LEV levels of at most 5 joints (SMPL: 9 levels; ctab != null) take the fast path:
//   phase 0  theta in, Rodrigues x24 (threads 0..23) BESIDE the 72 joint coordinates (threads 64..135: they need beta and the
//            folded regressor only); the chain wavefront fetches its table row
//   barrier 1
//   phase 1  220 pose/shape coefficients (threads 0..191) BESIDE the kinematic chain (wavefront 3): one tree LEVEL at a time,
//            lane = (joint of the level, entry of its 3x4 transform), operands from host-built LDS addresses in one batch,
//            the parent's row by ds_bpermute from the lanes that computed it
//   barrier 2
//   phase 2  the fragment chunks of the A operand (A2h / A3), relative transforms G', G2h fragments, 4x4 outputs
// Other trees: joints in phase 1, the chain in phase 2 with its look-ups in LDS, a third barrier, then the outputs.
// levels: [nlev + 1] offsets into lvl_joint, then the joints sorted by depth (built at model creation).
#ifndef PST
#define PST(i)
#define PSTC(i)
#endif
// theta_frame: the frame's [25][3] configuration (global memory in pose_kernel; the solve kernel hands over its LDS copy)
__device__ __forceinline__ void pose_body(const PoseArgs & pa, const int64_t f, const int tid, const float * __restrict__ theta_frame)
{
  // rotations [24][9] | joints [24][3] | zero[4] in ONE array: the chain's operand addresses are indices into it (CT_* below)
  __shared__ float sP[CT_P_SIZE];
  float(*sR)[9] = reinterpret_cast<float(*)[9]>(sP + CT_P_R);
  float(*sJ)[3] = reinterpret_cast<float(*)[3]>(sP + CT_P_J);
  float * const sZero = sP + CT_P_ZERO;
  __shared__ __attribute__((aligned(16))) float sG[NJ + 1][12]; // global transforms [A | g], 3x4 row-major (+ a spare row: dead chain lanes store there)
  __shared__ float sBeta[NB];
  __shared__ float sCoef[224]; // the A operand row of this frame: [c(207) | pa.beta(10) | 1 | 0...]
  __shared__ int sPar[NJ];
  __shared__ int sLvl[NJ + 1 + NJ];
  PST(0);
  // ---- phase 0 (the folded-regressor rows are fetched now, so their latency overlaps Rodrigues and the first barrier)
  // fast path (pa.ctab): the joints do not depend on the rotations — threads 64..135 compute them NOW, beside Rodrigues (pa.beta
  // straight from global memory: a uniform address), so that the chain wavefront can start at the first barrier
  const int jt = pa.ctab ? ((tid >= 64 && tid < 64 + NJ * 3) ? tid - 64 : -1) : (tid < NJ * 3 ? tid : -1);
  float j0v = 0.0f, jsv[NB];
#pragma unroll
  for(int k = 0; k < NB; k++) jsv[k] = 0.0f;
  if(jt >= 0)
  {
    if(pa.JSp) // [pa.JS row | pa.J0 | 0] in one 48-byte row: three loads instead of eleven
    {
      const float4 * row = reinterpret_cast<const float4 *>(pa.JSp + jt * 12);
      const float4 a = row[0], b = row[1], c = row[2];
      jsv[0] = a.x; jsv[1] = a.y; jsv[2] = a.z; jsv[3] = a.w;
      jsv[4] = b.x; jsv[5] = b.y; jsv[6] = b.z; jsv[7] = b.w;
      jsv[8] = c.x; jsv[9] = c.y;
      j0v = c.z;
    }
    else
    {
      j0v = pa.J0[jt];
#pragma unroll
      for(int k = 0; k < NB; k++) jsv[k] = pa.JS[jt * NB + k];
    }
  }
  // chain wavefront: this lane's row of the chain table (model.hip, CT_*): per level the joint of its slot, the pa.parent, the
  // pa.parent's slot, and WHERE its operand lies in sP — in registers
  int cti[CT_LEV], ctp[CT_LEV], cts[CT_LEV], cta[CT_LEV];
#pragma unroll
  for(int L = 0; L < CT_LEV; L++)
  {
    cti[L] = ctp[L] = -1;
    cts[L] = 0;
    cta[L] = CT_P_ZERO | (CT_P_ZERO << 10) | (1 << 20);
  }
  if(pa.ctab && tid >= 192 && tid < 192 + 60)
  {
    const int4 * row = reinterpret_cast<const int4 *>(pa.ctab + (tid - 192) * (2 * CT_LEV));
    int w[2 * CT_LEV];
#pragma unroll
    for(int q = 0; q < 2 * CT_LEV / 4; q++)
    {
      const int4 v = row[q];
      w[4 * q + 0] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
    }
#pragma unroll
    for(int L = 0; L < CT_LEV; L++)
    {
      const int i = w[2 * L] & 0xff, p = (w[2 * L] >> 8) & 0xff;
      cti[L] = i == 0xff ? -1 : i;
      ctp[L] = p == 0xff ? -1 : p;
      cts[L] = (w[2 * L] >> 16) & 0xff;
      cta[L] = w[2 * L + 1];
    }
  }
  if(tid >= 160 && tid < 160 + NB) sBeta[tid - 160] = pa.beta ? pa.beta[f * NB + (tid - 160)] : 0.0f;
  if(tid < 4) sZero[tid] = 0.0f;
  if(!pa.ctab) // (the tree tables in LDS serve the generic chain only)
  {
    if(tid >= 128 && tid < 128 + NJ) sPar[tid - 128] = pa.parent[tid - 128];
    if(tid >= 192 && tid < 192 + pa.nlev + 1) sLvl[tid - 192] = pa.lvl_off[tid - 192];
    if(tid >= 224 && tid < 224 + NJ) sLvl[NJ + 1 + tid - 224] = pa.lvl_joint[tid - 224];
  }
  if(tid < NJ)
  {
    float R[9];
    const float * th = theta_frame + (1 + tid) * 3; // theta[:,1:,:] (src/SMPL.cpp:685-686)
    rodrigues9(th[0], th[1], th[2], R);
#pragma unroll
    for(int q = 0; q < 9; q++) sR[tid][q] = R[q];
    if(pa.rot_out)
#pragma unroll
      for(int q = 0; q < 9; q++) pa.rot_out[(f * NJ + tid) * 9 + q] = R[q];
  }
  if(pa.ctab && jt >= 0) // joints (src/JointRegression.cpp:588-590 through the folded regressor)
  {
    float be[NB];
#pragma unroll
    for(int k = 0; k < NB; k++) be[k] = pa.beta ? pa.beta[f * NB + k] : 0.0f;
    const float s = joint_coord(j0v, jsv, be);
    sJ[jt / 3][jt % 3] = s;
    if(pa.joints_out) pa.joints_out[f * NJ * 3 + jt] = s;
  }
  PST(1);
  block_sync_lds();
  PST(2);
  // ---- phase 1: coefficient k = tid (root joint has no pose corrective: src/BlendShape.cpp:884-887) and joint coordinate tid
  // (fast path: the chain wavefront has its own work in this phase; threads 0..31 take its 32 coefficients too)
  for(int k = tid; k < 224 && (!pa.ctab || tid < 192); k += pa.ctab ? 192 : 256)
  {
    float a = 0.0f;
    if(k < NP)
    {
      const int q = k % 9;
      a = sR[1 + k / 9][q] - ((q == 0 || q == 4 || q == 8) ? 1.0f : 0.0f);
    }
    else if(k < NP + NB)
      a = sBeta[k - NP];
    else if(k == K_ONE)
      a = 1.0f;
    sCoef[k] = a;
    if(pa.AT && k < KP) pa.AT[(int64_t)k * pa.ldA + f] = a;
  }
  if(!pa.ctab && tid < NJ * 3) // joints (src/JointRegression.cpp:588-590 through the folded regressor)
  {
    const float s = joint_coord(j0v, jsv, sBeta);
    sJ[tid / 3][tid % 3] = s;
    if(pa.joints_out) pa.joints_out[f * NJ * 3 + tid] = s;
  }
  if(tid >= 192 && pa.ctab)
  {
    // chain: G_0 = L_0, G_i = G_p(i) . L_i with L_i = [R_i | j_i - j_p(i)] (src/WorldTransformation.cpp:508-610), level by
    // level; within a level the joints are independent (their parents are one level up).  Rotations and joints are both
    // complete at the first barrier, so the chain runs beside the coefficient phase.  The lane's operand of every level (a
    // column of R_i, or the offset j_i - j_p) does not depend on the chain: fetched up front, from addresses the host put
    // into the table (no per-level address arithmetic: that was half of this wavefront's time).
    PSTC(8);
    const int lane = tid - 192, e = lane % 12, r = e / 4, c = e % 4;
    float * const sGflat = &sG[0][0];
    float x0[CT_LEV], x1[CT_LEV], x2[CT_LEV];
#pragma unroll
    for(int L = 0; L < CT_LEV; L++)
    {
      // (every level of the table, live or not — dead ones point at the zero words: one batch of loads, one wait)
      const float * a = sP + (cta[L] & 0x3ff);
      const float * b = sP + ((cta[L] >> 10) & 0x3ff);
      const int st = cta[L] >> 20; // 3: a column of R_i, 1: j_i
      const float a0 = a[0], a1 = a[st], a2 = a[2 * st], b0 = b[0], b1 = b[1], b2 = b[2];
      x0[L] = a0 - b0;
      x1[L] = a1 - b1;
      x2[L] = a2 - b2;
      if(ctp[L] < 0) x0[L] = (r == 0) ? x0[L] : (r == 1 ? x1[L] : x2[L]); // root: L_0 = [R_0 | j_0], entry (r, c) itself
    }
    PSTC(9);
    // The pa.parent's row comes out of the REGISTERS of the lanes that computed it one level earlier (ds_bpermute through
    // __shfl: no LDS write -> wait -> read turn-around per level); the LDS copy is written on the side for phase 3.
    float vprev = 0.0f;
#pragma unroll
    for(int L = 0; L < CT_LEV; L++)
    {
      if(L < pa.nlev) // (wave-uniform)
      {
        const int i = cti[L], p = ctp[L], src = 12 * cts[L] + r * 4;
        const float g0 = __shfl(vprev, src + 0, 64), g1 = __shfl(vprev, src + 1, 64), g2 = __shfl(vprev, src + 2, 64),
                    g3 = __shfl(vprev, src + 3, 64);
        const float vc = chain_entry(g0, g1, g2, g3, x0[L], x1[L], x2[L], c == 3);
        const float v = p >= 0 ? vc : x0[L];
        // (selects instead of a divergent branch: lanes without a joint at this level keep their value and store to a spare word)
        const bool live = lane < 60 && i >= 0;
        vprev = live ? v : vprev;
        sGflat[live ? i * 12 + e : NJ * 12 + (lane & 3)] = v;
      }
    }
    PSTC(10);
  }
  block_sync_lds();
  PST(3);
  // ---- phase 2
  if(pa.A3 && tid < 84)
  {
    // bf16x3 pieces in MFMA fragment order (layout: common.h): chunk c = k / 8 is element block j of MFMA lane 32 h + r in
    // k-step ks = c / 2, h = c % 2; thread (c, s) writes the 16 bytes of piece s
    const int c = tid % 28, sp = tid / 28, ks = c >> 1, h = c & 1;
    const int64_t ftp = f >> 6;
    const int fh = (int)((f >> 5) & 1), r = (int)(f & 31);
    uint16_t pc[8];
#pragma unroll
    for(int j = 0; j < 8; j++)
    {
      uint16_t p0, p1, p2;
      split_bf16x3(sCoef[8 * c + j], p0, p1, p2);
      pc[j] = sp == 0 ? p0 : (sp == 1 ? p1 : p2);
    }
    uint16_t * dst = pa.A3 + ((((ftp * BB_KS + ks) * 2 + fh) * 3 + sp) * 64 + (32 * h + r)) * 8;
    uint4 w;
    w.x = pc[0] | ((uint32_t)pc[1] << 16);
    w.y = pc[2] | ((uint32_t)pc[3] << 16);
    w.z = pc[4] | ((uint32_t)pc[5] << 16);
    w.w = pc[6] | ((uint32_t)pc[7] << 16);
    *reinterpret_cast<uint4 *>(dst) = w;
  }
  if(pa.A2h && tid >= 96 && tid < 96 + 28)
  {
    // fp16x2 pieces in MFMA fragment order (layout: common.h): chunk c = k / 8 is element block j of MFMA lane 32 h + r in
    // k-step ks = c / 2, h = c % 2; both pieces of the chunk by one thread
    const int c = tid - 96, ks = c >> 1, h = c & 1;
    const int64_t ft = f >> 6;
    const int fh = (int)((f >> 5) & 1), r = (int)(f & 31);
    pose_f16x8 hi, lo;
#pragma unroll
    for(int j = 0; j < 8; j++)
    {
      _Float16 a, b;
      const float xs = sCoef[8 * c + j] * HB_SA;
      if(!(__builtin_fabsf(xs) <= 65504.0f)) atomicOr(pa.range_flag, 1); // outside fp16's range (|pa.beta| >= 1023) or not finite
      split_f16x2(xs, a, b);
      hi[j] = a;
      lo[j] = b;
    }
    _Float16 * dst = pa.A2h + ((((ft * HB_KS + ks) * 2 + fh) * 2) * 64 + (32 * h + r)) * 8;
    *reinterpret_cast<pose_f16x8 *>(dst) = hi;
    *reinterpret_cast<pose_f16x8 *>(dst + 64 * 8) = lo;
  }
  if(tid >= 192 && !pa.ctab)
  {
    // generic trees (deeper than CT_LEV levels or wider than 5 joints per level): the same chain with its look-ups in LDS
    const int lane = tid - 192, slot = lane / 12, e = lane % 12, r = e / 4, c = e % 4;
    for(int L = 0; L < pa.nlev; L++)
    {
      const int lo = sLvl[L], hi = sLvl[L + 1];
      for(int q0 = lo; q0 < hi; q0 += 5)
      {
        if(slot < 5 && q0 + slot < hi)
        {
          const int i = sLvl[NJ + 1 + q0 + slot], p = sPar[i];
          float v;
          if(p < 0)
            v = (c < 3) ? sR[i][r * 3 + c] : sJ[i][r];
          else if(c < 3)
            v = chain_entry(sG[p][r * 4 + 0], sG[p][r * 4 + 1], sG[p][r * 4 + 2], 0.0f, sR[i][0 * 3 + c], sR[i][1 * 3 + c], sR[i][2 * 3 + c], false);
          else
          {
            const float t0 = sJ[i][0] - sJ[p][0], t1 = sJ[i][1] - sJ[p][1], t2 = sJ[i][2] - sJ[p][2];
            v = chain_entry(sG[p][r * 4 + 0], sG[p][r * 4 + 1], sG[p][r * 4 + 2], sG[p][r * 4 + 3], t0, t1, t2, true);
          }
          sG[i][e] = v;
        }
      }
      wave_sync();
    }
  }
  PST(4);
  if(!pa.ctab) block_sync_lds(); // (fast path: the chain finished before the second barrier)
  PST(5);
  // ---- phase 3: relative transforms: translation -= A_i . j_i (src/WorldTransformation.cpp:657-677)
  for(int e = tid; e < NJ * 12; e += 256)
  {
    const int i = e / 12, q = e % 12, r = q / 4, c = q % 4;
    float v = sG[i][q];
    if(c == 3) v = relative_t(v, sG[i][r * 4 + 0], sG[i][r * 4 + 1], sG[i][r * 4 + 2], sJ[i][0], sJ[i][1], sJ[i][2]);
    if(pa.Gp) pa.Gp[(f * NJ + i) * 12 + q] = v;
    if(pa.xf44_out) pa.xf44_out[(f * NJ + i) * 16 + q] = v;
  }
  if(pa.xf44_out && tid < NJ * 4) pa.xf44_out[(f * NJ + tid / 4) * 16 + 12 + tid % 4] = (tid % 4 == 3) ? 1.0f : 0.0f;
  if(pa.G2h && tid >= 64 && tid < 64 + 36)
  {
    // the relative transforms once more as the A operand of the blend MFMAs of skin_h.hip (rows = frames, k = joint):
    // thread (entry e, chunk c) writes both fp16x2 pieces of joints 8 c .. 8 c + 7 of entry e (layout: common.h)
    const int e = (tid - 64) / 3, c = (tid - 64) % 3, r4 = e / 4, cc = e % 4;
    const int64_t ft = f >> 6;
    const int fh = (int)((f >> 5) & 1), r = (int)(f & 31);
    pose_f16x8 hi, lo;
#pragma unroll
    for(int j = 0; j < 8; j++)
    {
      const int i = 8 * c + j;
      float v = sG[i][e];
      if(cc == 3) v = relative_t(v, sG[i][r4 * 4 + 0], sG[i][r4 * 4 + 1], sG[i][r4 * 4 + 2], sJ[i][0], sJ[i][1], sJ[i][2]);
      _Float16 a, b;
      if(!(__builtin_fabsf(v * pa.gscale) <= 65504.0f)) atomicOr(pa.range_flag, 1); // a transform outside 16 x the template's extent
      split_f16x2(v * pa.gscale, a, b);
      hi[j] = a;
      lo[j] = b;
    }
    _Float16 * blk = pa.G2h + (((ft * 2 + fh) * 12 + e) * 3072) / 2;
    if(c < 2)
    {
      *reinterpret_cast<pose_f16x8 *>(blk + (32 * c + r) * 8) = hi;
      *reinterpret_cast<pose_f16x8 *>(blk + 512 + (32 * c + r) * 8) = lo;
    }
    else
    {
      *reinterpret_cast<pose_f16x8 *>(blk + 1024 + r * 8) = hi;
      *reinterpret_cast<pose_f16x8 *>(blk + 1024 + 256 + r * 8) = lo;
    }
  }
  PST(6);
}

} // namespace smplpp_hip
