// SMPL::launch (/root/reference/src/SMPL.cpp:671-737) as two gfx950 kernels.
//
//  pose_kernel   one workgroup (four wavefronts) per frame: Rodrigues x24 (src/BlendShape.cpp:803-844), pose coefficients
//                vec(R)[9:] - vec(I)[9:] (:865-895), joints = J0 + JS.beta (src/JointRegression.cpp:583-598, folded),
//                FK chain over the kinematic tree and relative transforms (src/WorldTransformation.cpp:421-677).
//                Writes the A operand of the fused kernel of the model's form (A2h + G2h fragments for skin_h.hip, A3 for
//                skin_b.hip, the K-major AT[220][ldA] for the fp32 forms) and G'[n][24][3x4] in fp32.
//  skin_kernel   (the FIRST form of the fused kernel, SMPLPP_SKIN=v; the default is skin_kernel_h, skin_h.hip) fused: rest = T + S.beta + P.c as ONE fp32 GEMM [frames x 220] x [220 x 3V] on
//                v_mfma_f32_32x32x2_f32 (exact fp32; bf16/fp16 operands would break the 1e-5 m bound), then in the
//                epilogue, on the accumulator registers, linear blend skinning with the frame tile's G' staged in LDS
//                (src/JointRegression.cpp:551-565, src/LinearBlendSkinning.cpp:445-553).  No [n,V,4,4] intermediate,
//                no rest-shape round trip through HBM.
//
// Tile: a 256-thread workgroup = 4 wavefronts = (32*FT frames) x (4 groups of 32 vertices); each wavefront owns
// FT x 3 accumulator tiles of 32x32 (frames x {x,y,z} of its 32 vertices), so a lane ends up holding rest_x/y/z of
// ONE vertex for 16*FT frames and skins them without any cross-lane traffic.
// XCD-aware mapping: blocks b and b+8 share an XCD (round-robin dispatch); all frame tiles of one vertex quad are
// given to one XCD, consecutively, so each 338 KB slice of Bm is pulled into that XCD's L2 once per launch.
#include "common.h"
#include "pose_math.h"
#include "trace.h"

namespace smplpp_hip
{
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// ---------------------------------------------------------------------------------------------- pose kernel
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
#ifdef POSE_STAMP
__device__ unsigned long long g_pose_stamps[16];
#define PST(i) if(blockIdx.x == 512 && threadIdx.x == 0) g_pose_stamps[i] = __builtin_amdgcn_s_memtime()
#define PSTC(i) if(blockIdx.x == 512 && threadIdx.x == 192) g_pose_stamps[i] = __builtin_amdgcn_s_memtime()
extern "C" int smplpp_debug_pose_stamps(unsigned long long * out)
{
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pose_stamps), sizeof(unsigned long long) * 16);
}
#else
#define PST(i)
#define PSTC(i)
#endif
__global__ __launch_bounds__(256) void pose_kernel(const float * __restrict__ beta, const float * __restrict__ theta,
                                                   const float * __restrict__ J0, const float * __restrict__ JS, const float * __restrict__ JSp,
                                                   const int32_t * __restrict__ parent, const int32_t * __restrict__ lvl_off,
                                                   const int32_t * __restrict__ lvl_joint, int nlev, float * __restrict__ AT, int64_t ldA,
                                                   float * __restrict__ Gp, float * __restrict__ joints_out,
                                                   float * __restrict__ rot_out, float * __restrict__ xf44_out, int64_t n,
                                                   uint16_t * __restrict__ A3, _Float16 * __restrict__ A2h,
                                                   _Float16 * __restrict__ G2h, float gscale, const int32_t * __restrict__ ctab,
                                                   int * __restrict__ range_flag)
{
  const int64_t f = blockIdx.x;
  const int tid = threadIdx.x;
  // rotations [24][9] | joints [24][3] | zero[4] in ONE array: the chain's operand addresses are indices into it (CT_* below)
  __shared__ float sP[CT_P_SIZE];
  float(*sR)[9] = reinterpret_cast<float(*)[9]>(sP + CT_P_R);
  float(*sJ)[3] = reinterpret_cast<float(*)[3]>(sP + CT_P_J);
  float * const sZero = sP + CT_P_ZERO;
  __shared__ __attribute__((aligned(16))) float sG[NJ + 1][12]; // global transforms [A | g], 3x4 row-major (+ a spare row: dead chain lanes store there)
  __shared__ float sBeta[NB];
  __shared__ float sCoef[224]; // the A operand row of this frame: [c(207) | beta(10) | 1 | 0...]
  __shared__ int sPar[NJ];
  __shared__ int sLvl[NJ + 1 + NJ];
  if(f >= n) return;
  PST(0);
  // ---- phase 0 (the folded-regressor rows are fetched now, so their latency overlaps Rodrigues and the first barrier)
  // fast path (ctab): the joints do not depend on the rotations — threads 64..135 compute them NOW, beside Rodrigues (beta
  // straight from global memory: a uniform address), so that the chain wavefront can start at the first barrier
  const int jt = ctab ? ((tid >= 64 && tid < 64 + NJ * 3) ? tid - 64 : -1) : (tid < NJ * 3 ? tid : -1);
  float j0v = 0.0f, jsv[NB];
#pragma unroll
  for(int k = 0; k < NB; k++) jsv[k] = 0.0f;
  if(jt >= 0)
  {
    if(JSp) // [JS row | J0 | 0] in one 48-byte row: three loads instead of eleven
    {
      const float4 * row = reinterpret_cast<const float4 *>(JSp + jt * 12);
      const float4 a = row[0], b = row[1], c = row[2];
      jsv[0] = a.x; jsv[1] = a.y; jsv[2] = a.z; jsv[3] = a.w;
      jsv[4] = b.x; jsv[5] = b.y; jsv[6] = b.z; jsv[7] = b.w;
      jsv[8] = c.x; jsv[9] = c.y;
      j0v = c.z;
    }
    else
    {
      j0v = J0[jt];
#pragma unroll
      for(int k = 0; k < NB; k++) jsv[k] = JS[jt * NB + k];
    }
  }
  // chain wavefront: this lane's row of the chain table (model.hip, CT_*): per level the joint of its slot, the parent, the
  // parent's slot, and WHERE its operand lies in sP — in registers
  int cti[CT_LEV], ctp[CT_LEV], cts[CT_LEV], cta[CT_LEV];
#pragma unroll
  for(int L = 0; L < CT_LEV; L++)
  {
    cti[L] = ctp[L] = -1;
    cts[L] = 0;
    cta[L] = CT_P_ZERO | (CT_P_ZERO << 10) | (1 << 20);
  }
  if(ctab && tid >= 192 && tid < 192 + 60)
  {
    const int4 * row = reinterpret_cast<const int4 *>(ctab + (tid - 192) * (2 * CT_LEV));
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
  if(tid >= 160 && tid < 160 + NB) sBeta[tid - 160] = beta ? beta[f * NB + (tid - 160)] : 0.0f;
  if(tid < 4) sZero[tid] = 0.0f;
  if(!ctab) // (the tree tables in LDS serve the generic chain only)
  {
    if(tid >= 128 && tid < 128 + NJ) sPar[tid - 128] = parent[tid - 128];
    if(tid >= 192 && tid < 192 + nlev + 1) sLvl[tid - 192] = lvl_off[tid - 192];
    if(tid >= 224 && tid < 224 + NJ) sLvl[NJ + 1 + tid - 224] = lvl_joint[tid - 224];
  }
  if(tid < NJ)
  {
    float R[9];
    const float * th = theta + (f * (NJ + 1) + 1 + tid) * 3; // theta[:,1:,:] (src/SMPL.cpp:685-686)
    rodrigues9(th[0], th[1], th[2], R);
#pragma unroll
    for(int q = 0; q < 9; q++) sR[tid][q] = R[q];
    if(rot_out)
#pragma unroll
      for(int q = 0; q < 9; q++) rot_out[(f * NJ + tid) * 9 + q] = R[q];
  }
  if(ctab && jt >= 0) // joints (src/JointRegression.cpp:588-590 through the folded regressor)
  {
    float be[NB];
#pragma unroll
    for(int k = 0; k < NB; k++) be[k] = beta ? beta[f * NB + k] : 0.0f;
    const float s = joint_coord(j0v, jsv, be);
    sJ[jt / 3][jt % 3] = s;
    if(joints_out) joints_out[f * NJ * 3 + jt] = s;
  }
  PST(1);
  block_sync_lds();
  PST(2);
  // ---- phase 1: coefficient k = tid (root joint has no pose corrective: src/BlendShape.cpp:884-887) and joint coordinate tid
  // (fast path: the chain wavefront has its own work in this phase; threads 0..31 take its 32 coefficients too)
  for(int k = tid; k < 224 && (!ctab || tid < 192); k += ctab ? 192 : 256)
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
    if(AT && k < KP) AT[(int64_t)k * ldA + f] = a;
  }
  if(!ctab && tid < NJ * 3) // joints (src/JointRegression.cpp:588-590 through the folded regressor)
  {
    const float s = joint_coord(j0v, jsv, sBeta);
    sJ[tid / 3][tid % 3] = s;
    if(joints_out) joints_out[f * NJ * 3 + tid] = s;
  }
  if(tid >= 192 && ctab)
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
    // The parent's row comes out of the REGISTERS of the lanes that computed it one level earlier (ds_bpermute through
    // __shfl: no LDS write -> wait -> read turn-around per level); the LDS copy is written on the side for phase 3.
    float vprev = 0.0f;
#pragma unroll
    for(int L = 0; L < CT_LEV; L++)
    {
      if(L < nlev) // (wave-uniform)
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
  if(A3 && tid < 84)
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
    uint16_t * dst = A3 + ((((ftp * BB_KS + ks) * 2 + fh) * 3 + sp) * 64 + (32 * h + r)) * 8;
    uint4 w;
    w.x = pc[0] | ((uint32_t)pc[1] << 16);
    w.y = pc[2] | ((uint32_t)pc[3] << 16);
    w.z = pc[4] | ((uint32_t)pc[5] << 16);
    w.w = pc[6] | ((uint32_t)pc[7] << 16);
    *reinterpret_cast<uint4 *>(dst) = w;
  }
  if(A2h && tid >= 96 && tid < 96 + 28)
  {
    // fp16x2 pieces in MFMA fragment order (layout: common.h): chunk c = k / 8 is element block j of MFMA lane 32 h + r in
    // k-step ks = c / 2, h = c % 2; both pieces of the chunk by one thread
    const int c = tid - 96, ks = c >> 1, h = c & 1;
    const int64_t ft = f >> 6;
    const int fh = (int)((f >> 5) & 1), r = (int)(f & 31);
    f16x8 hi, lo;
#pragma unroll
    for(int j = 0; j < 8; j++)
    {
      _Float16 a, b;
      const float xs = sCoef[8 * c + j] * HB_SA;
      if(!(__builtin_fabsf(xs) <= 65504.0f)) atomicOr(range_flag, 1); // outside fp16's range (|beta| >= 1023) or not finite
      split_f16x2(xs, a, b);
      hi[j] = a;
      lo[j] = b;
    }
    _Float16 * dst = A2h + ((((ft * HB_KS + ks) * 2 + fh) * 2) * 64 + (32 * h + r)) * 8;
    *reinterpret_cast<f16x8 *>(dst) = hi;
    *reinterpret_cast<f16x8 *>(dst + 64 * 8) = lo;
  }
  if(tid >= 192 && !ctab)
  {
    // generic trees (deeper than CT_LEV levels or wider than 5 joints per level): the same chain with its look-ups in LDS
    const int lane = tid - 192, slot = lane / 12, e = lane % 12, r = e / 4, c = e % 4;
    for(int L = 0; L < nlev; L++)
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
  if(!ctab) block_sync_lds(); // (fast path: the chain finished before the second barrier)
  PST(5);
  // ---- phase 3: relative transforms: translation -= A_i . j_i (src/WorldTransformation.cpp:657-677)
  for(int e = tid; e < NJ * 12; e += 256)
  {
    const int i = e / 12, q = e % 12, r = q / 4, c = q % 4;
    float v = sG[i][q];
    if(c == 3) v = relative_t(v, sG[i][r * 4 + 0], sG[i][r * 4 + 1], sG[i][r * 4 + 2], sJ[i][0], sJ[i][1], sJ[i][2]);
    if(Gp) Gp[(f * NJ + i) * 12 + q] = v;
    if(xf44_out) xf44_out[(f * NJ + i) * 16 + q] = v;
  }
  if(xf44_out && tid < NJ * 4) xf44_out[(f * NJ + tid / 4) * 16 + 12 + tid % 4] = (tid % 4 == 3) ? 1.0f : 0.0f;
  if(G2h && tid >= 64 && tid < 64 + 36)
  {
    // the relative transforms once more as the A operand of the blend MFMAs of skin_h.hip (rows = frames, k = joint):
    // thread (entry e, chunk c) writes both fp16x2 pieces of joints 8 c .. 8 c + 7 of entry e (layout: common.h)
    const int e = (tid - 64) / 3, c = (tid - 64) % 3, r4 = e / 4, cc = e % 4;
    const int64_t ft = f >> 6;
    const int fh = (int)((f >> 5) & 1), r = (int)(f & 31);
    f16x8 hi, lo;
#pragma unroll
    for(int j = 0; j < 8; j++)
    {
      const int i = 8 * c + j;
      float v = sG[i][e];
      if(cc == 3) v = relative_t(v, sG[i][r4 * 4 + 0], sG[i][r4 * 4 + 1], sG[i][r4 * 4 + 2], sJ[i][0], sJ[i][1], sJ[i][2]);
      _Float16 a, b;
      if(!(__builtin_fabsf(v * gscale) <= 65504.0f)) atomicOr(range_flag, 1); // a transform outside 16 x the template's extent
      split_f16x2(v * gscale, a, b);
      hi[j] = a;
      lo[j] = b;
    }
    _Float16 * blk = G2h + (((ft * 2 + fh) * 12 + e) * 3072) / 2;
    if(c < 2)
    {
      *reinterpret_cast<f16x8 *>(blk + (32 * c + r) * 8) = hi;
      *reinterpret_cast<f16x8 *>(blk + 512 + (32 * c + r) * 8) = lo;
    }
    else
    {
      *reinterpret_cast<f16x8 *>(blk + 1024 + r * 8) = hi;
      *reinterpret_cast<f16x8 *>(blk + 1024 + 256 + r * 8) = lo;
    }
  }
  PST(6);
}

// rows [n, ldA) of AT are padding for the last 32-frame tile: keep them zero (re-zeroed whenever n changes)
__global__ void zero_pad_kernel(float * __restrict__ AT, int64_t ldA, int64_t n)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t pad = ldA - n;
  if(i >= (int64_t)KP * pad) return;
  AT[(i / pad) * ldA + n + i % pad] = 0.0f;
}

// ---------------------------------------------------------------------------------------------- fused kernel
constexpr int KSTEPS = KP / 2; // 110 MFMA k-steps of 2
constexpr int UNR = 5;         // k-steps per software-pipeline chunk (110 = 22 * 5)

template<int FT, int MAXW>
__global__ __launch_bounds__(256, 2) void skin_kernel(const float * __restrict__ AT, int64_t ldA, const float * __restrict__ Bm,
                                                   int64_t ldB, const float * __restrict__ Gp, const float * __restrict__ theta,
                                                   const uint8_t * __restrict__ wIdx, const float * __restrict__ wVal,
                                                   const float * __restrict__ wSum, float * __restrict__ verts,
                                                   float * __restrict__ rest, int64_t n, int64_t V, int VGn, int nft)
{
  extern __shared__ __attribute__((aligned(16))) float lds[]; // [32*FT][24][12] G' + [32*FT][3] root translation
  constexpr int FRAMES = 32 * FT;
  float * sG = lds;
  float * sRoot = lds + FRAMES * NJ * 12;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int nq = (VGn + 3) / 4;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int q = (slot / nft) * 8 + xcd;
  const int ftile = slot % nft;
  if(q >= nq) return;
  const int64_t f0 = (int64_t)ftile * FRAMES;
  const int vg = q * 4 + wave;

  // stage G' and root translations of this frame tile (frames >= n read as zero)
  {
    const int64_t nvalid = (n - f0 < FRAMES ? n - f0 : FRAMES) * (NJ * 12);
    const float4 * src = reinterpret_cast<const float4 *>(Gp + f0 * NJ * 12);
    float4 * dst = reinterpret_cast<float4 *>(sG);
    for(int i = tid; i < FRAMES * NJ * 3; i += 256)
    {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if((int64_t)i * 4 < nvalid) v = src[i];
      dst[i] = v;
    }
    for(int i = tid; i < FRAMES * 3; i += 256)
    {
      int64_t f = f0 + i / 3;
      sRoot[i] = (f < n) ? theta[f * (NJ + 1) * 3 + i % 3] : 0.0f; // theta[:,0,:] (src/SMPL.cpp:726-727)
    }
  }

  f32x16 acc[FT][3];
#pragma unroll
  for(int t = 0; t < FT; t++)
#pragma unroll
    for(int x = 0; x < 3; x++)
#pragma unroll
      for(int r = 0; r < 16; r++) acc[t][x][r] = 0.0f;

  if(vg < VGn)
  {
    // operand pointers: lane l supplies A[row = l&31][k = l>>5] and B[k = l>>5][col = l&31]
    const float * Ap = AT + (int64_t)(lane >> 5) * ldA + f0 + (lane & 31);
    const float * Bp = Bm + (int64_t)(lane >> 5) * ldB + (int64_t)vg * (3 * VG) + (lane & 31);
    // Ping-pong operand buffers P/Q, each one chunk (UNR k-steps) deep, loop unrolled by two chunks so that no register
    // copies exist for the compiler to fold the two buffers back into one: the loads of a chunk are issued a full chunk
    // of MFMAs (UNR * FT * 3 * 64 cycles) before their first use.
    float aP[UNR][FT], bP[UNR][3], aQ[UNR][FT], bQ[UNR][3];
    auto load_chunk = [&](int c, float (&a)[UNR][FT], float (&b)[UNR][3]) {
      const float * An = Ap + (int64_t)(2 * UNR) * c * ldA;
      const float * Bn = Bp + (int64_t)(2 * UNR) * c * ldB;
#pragma unroll
      for(int u = 0; u < UNR; u++)
      {
#pragma unroll
        for(int t = 0; t < FT; t++) a[u][t] = An[(int64_t)(2 * u) * ldA + 32 * t];
#pragma unroll
        for(int x = 0; x < 3; x++) b[u][x] = Bn[(int64_t)(2 * u) * ldB + VG * x];
      }
    };
    auto mfma_chunk = [&](const float (&a)[UNR][FT], const float (&b)[UNR][3]) {
#pragma unroll
      for(int u = 0; u < UNR; u++)
#pragma unroll
        for(int t = 0; t < FT; t++)
#pragma unroll
          for(int x = 0; x < 3; x++) acc[t][x] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][t], b[u][x], acc[t][x], 0, 0, 0);
    };
    constexpr int NCH = KSTEPS / UNR; // 22 (even)
    load_chunk(0, aP, bP);
    for(int c = 0; c < NCH; c += 2)
    {
      load_chunk(c + 1, aQ, bQ);
      mfma_chunk(aP, bP);
      if(c + 2 < NCH) load_chunk(c + 2, aP, bP);
      mfma_chunk(aQ, bQ);
    }
  }
  __syncthreads(); // G' staged

  if(vg >= VGn) return;
  const int64_t v = (int64_t)vg * VG + (lane & 31);
  if(v >= V) return;
  // this lane's skinning weights
  int jidx[MAXW];
  float jw[MAXW];
#pragma unroll
  for(int i = 0; i < MAXW; i++)
  {
    jidx[i] = wIdx[v * MAXW + i];
    jw[i] = wVal[v * MAXW + i];
  }
  // cart = h[:3] / h[3] (src/LinearBlendSkinning.cpp:545-550) with h[3] = sum_j W[v,j], constant per vertex:
  // one reciprocal per lane instead of three IEEE divisions per (frame, vertex) — differs by <= 1 ulp (6e-8 m at 1 m)
  const float winv = 1.0f / wSum[v];
  float * vout = verts ? verts + v * 3 : nullptr;
  float * rout = rest ? rest + v * 3 : nullptr;
#pragma unroll
  for(int t = 0; t < FT; t++)
#pragma unroll
    for(int r = 0; r < 16; r++)
    {
      const int fl = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); // accumulator row -> frame in tile
      const int64_t f = f0 + fl;
      if(f >= n) continue;
      const float rx = acc[t][0][r], ry = acc[t][1][r], rz = acc[t][2][r];
      if(rout)
      {
        float * o = rout + f * V * 3;
        o[0] = rx;
        o[1] = ry;
        o[2] = rz;
      }
      if(!vout) continue;
      // M = sum_j W[v,j] G'_j (src/LinearBlendSkinning.cpp:463), rows of [A | b]
      float4 m0 = make_float4(0.f, 0.f, 0.f, 0.f), m1 = m0, m2 = m0;
      const float * g = sG + fl * (NJ * 12);
#pragma unroll
      for(int i = 0; i < MAXW; i++)
      {
        const float4 * gj = reinterpret_cast<const float4 *>(g + jidx[i] * 12);
        const float4 g0 = gj[0], g1 = gj[1], g2 = gj[2];
        const float w = jw[i];
        m0.x += w * g0.x; m0.y += w * g0.y; m0.z += w * g0.z; m0.w += w * g0.w;
        m1.x += w * g1.x; m1.y += w * g1.y; m1.z += w * g1.z; m1.w += w * g1.w;
        m2.x += w * g2.x; m2.y += w * g2.y; m2.z += w * g2.z; m2.w += w * g2.w;
      }
      // h = M [rest; 1] (:465-467); cart = h[:3] * (1 / h[3]); + root (:475)
      const float hx = ((m0.x * rx + m0.y * ry) + m0.z * rz) + m0.w;
      const float hy = ((m1.x * rx + m1.y * ry) + m1.z * rz) + m1.w;
      const float hz = ((m2.x * rx + m2.y * ry) + m2.z * rz) + m2.w;
      float * o = vout + f * V * 3;
      o[0] = hx * winv + sRoot[fl * 3 + 0];
      o[1] = hy * winv + sRoot[fl * 3 + 1];
      o[2] = hz * winv + sRoot[fl * 3 + 2];
    }
}

template<int FT, int MAXW>
static hipError_t launch_skin(const smplpp_model * m, int64_t n, const float * theta, float * verts, float * rest,
                              hipStream_t st)
{
  const int nft = (int)((n + 32 * FT - 1) / (32 * FT));
  const int nq = (int)((m->VGn + 3) / 4);
  const int grid = 8 * ((nq + 7) / 8) * nft;
  const size_t shmem = sizeof(float) * (size_t)(32 * FT) * (NJ * 12 + 3);
  static PerDeviceOnce once;
  {
    hipError_t e = lds_opt_in(once, m->device, reinterpret_cast<const void *>(&skin_kernel<FT, MAXW>), (int)shmem);
    if(e != hipSuccess) return e;
  }
  skin_kernel<FT, MAXW><<<dim3(grid), dim3(256), shmem, st>>>(m->ws.AT.as<float>(), m->ws.ldA, m->Bm, m->ldB,
                                                              m->ws.Gp.as<float>(), theta, m->wIdx, m->wVal, m->wSum, verts,
                                                              rest, n, m->V, (int)m->VGn, nft);
  return hipGetLastError();
}

template<int FT>
static hipError_t launch_skin_w(const smplpp_model * m, int64_t n, const float * theta, float * verts, float * rest,
                                hipStream_t st)
{
  switch(m->maxw)
  {
    case 4:
      return launch_skin<FT, 4>(m, n, theta, verts, rest, st);
    case 8:
      return launch_skin<FT, 8>(m, n, theta, verts, rest, st);
    default:
      return launch_skin<FT, NJ>(m, n, theta, verts, rest, st);
  }
}

hipError_t launch_skin_persistent(const smplpp_model * m, int64_t n, const float * theta, const float * Gp_padded, float * verts,
                                  float * rest, hipStream_t st); // skin_p.hip (fp32 MFMA, one wave/SIMD, epilogue in the MFMA shadow)
hipError_t launch_skin_bf16x3(const smplpp_model * m, int64_t n, const float * theta, float * verts, float * rest, hipStream_t st); // skin_b.hip
hipError_t launch_skin_f16x2(const smplpp_model * m, int64_t n, const float * theta, float * verts, float * rest, hipStream_t st);  // skin_h.hip

// Device-pointer FK (enqueue only).  Used by smplpp_fk and by the IK solver.
int fk_device(smplpp_model * m, int64_t n, const float * beta, const float * theta, float * verts, float * joints,
              float * xforms44, float * rest, float * poserot, hipStream_t st)
{
  Workspace & ws = m->ws;
  // Form of the fused kernel (m->form, from SMPLPP_SKIN at model creation): h (default,
  // skin_h.hip): fp16x2 operand pieces on the f16 matrix pipe, skinning on the matrix pipe too; b (skin_b.hip): bf16x3
  // pieces, VALU skinning in MFMA shadows; p (skin_p.hip): exact fp32 MFMA; v (skin_kernel above): the first form.
  // b needs <= 8 weights per vertex; p (32-bit output offsets) falls back to v for outputs of 2 GiB and more.
  char form = m->form; // (model creation already turned b / p into v for models with more than 8 weights per vertex)
  if(form == 'p' && n * m->V * 12 >= 0x7fffff00LL) form = 'v';
  const int64_t n64 = ((n + 63) / 64) * 64;
  HIP_TRY(ws.Gp.reserve(sizeof(float) * (size_t)n64 * NJ * 12)); // b / p stage whole frame tiles of G' (padding never stored)
  if(form == 'h')
  {
    HIP_TRY(ws.A2h.reserve((size_t)(n64 / 64) * HB_KS * HB_A_BYTES));
    HIP_TRY(ws.G2h.reserve((size_t)(n64 / 64) * HB_G_BYTES));
    pose_kernel<<<dim3((unsigned)n), dim3(256), 0, st>>>(beta, theta, m->J0, m->JS, m->JSp, m->parent, m->lvl, m->lvl + NJ + 1, m->nlev, nullptr, 0,
                                                         ws.Gp.as<float>(), joints, poserot, xforms44, n, nullptr,
                                                         (verts || rest) ? ws.A2h.as<_Float16>() : nullptr,
                                                         (verts || rest) ? ws.G2h.as<_Float16>() : nullptr, m->sG, m->chain_fast ? m->lvl + CT_OFF : nullptr, m->range_flag);
  }
  else if(form == 'b')
  {
    HIP_TRY(ws.A3.reserve((size_t)(n64 / 64) * BB_KS * BB_A_BYTES));
    pose_kernel<<<dim3((unsigned)n), dim3(256), 0, st>>>(beta, theta, m->J0, m->JS, m->JSp, m->parent, m->lvl, m->lvl + NJ + 1, m->nlev, nullptr, 0,
                                                         ws.Gp.as<float>(), joints, poserot, xforms44, n, ws.A3.as<uint16_t>(), nullptr, nullptr, 1.0f, m->chain_fast ? m->lvl + CT_OFF : nullptr, nullptr);
  }
  else
  {
    const int64_t ldA = n64;
    HIP_TRY(ws.AT.reserve(sizeof(float) * (size_t)KP * ldA));
    if(n64 > n) HIP_TRY(hipMemsetAsync(ws.Gp.as<float>() + n * NJ * 12, 0, sizeof(float) * (size_t)(n64 - n) * NJ * 12, st));
    ws.ldA = ldA;
    if(ldA > n)
    {
      int64_t cnt = (int64_t)KP * (ldA - n);
      zero_pad_kernel<<<dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st>>>(ws.AT.as<float>(), ldA, n);
    }
    pose_kernel<<<dim3((unsigned)n), dim3(256), 0, st>>>(beta, theta, m->J0, m->JS, m->JSp, m->parent, m->lvl, m->lvl + NJ + 1, m->nlev, ws.AT.as<float>(),
                                                         ldA, ws.Gp.as<float>(), joints, poserot, xforms44, n, nullptr, nullptr, nullptr, 1.0f, m->chain_fast ? m->lvl + CT_OFF : nullptr, nullptr);
  }
  HIP_TRY(hipGetLastError());
  if(verts || rest)
  {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if(m->profiling)
    {
      // (owned by the handle from the moment they exist: smplpp_profile_read / smplpp_model_destroy release them)
      // (owned by the handle as a PAIR: a failed second creation must not leave the begin/end list misaligned)
      HIP_TRY(hipEventCreate(&e0));
      if(hipError_t ee = hipEventCreate(&e1); ee != hipSuccess)
      {
        (void)hipEventDestroy(e0);
        return hip_fail(ee, "hipEventCreate", __FILE__, __LINE__);
      }
      m->prof_events.push_back(e0);
      m->prof_events.push_back(e1);
      HIP_TRY(hipEventRecord(e0, st));
    }
    if(form == 'h')
      HIP_TRY(launch_skin_f16x2(m, n, theta, verts, rest, st));
    else if(form == 'b')
      HIP_TRY(launch_skin_bf16x3(m, n, theta, verts, rest, st));
    else if(form == 'p' && ws.dummy.reserve(4096) == hipSuccess)
      HIP_TRY(launch_skin_persistent(m, n, theta, ws.Gp.as<float>(), verts, rest, st));
    else if(n <= 32)
      HIP_TRY(launch_skin_w<1>(m, n, theta, verts, rest, st));
    else
      HIP_TRY(launch_skin_w<2>(m, n, theta, verts, rest, st));
    if(m->profiling) HIP_TRY(hipEventRecord(e1, st));
  }
  return SMPLPP_OK;
}
} // namespace smplpp_hip

using namespace smplpp_hip;

extern "C" int smplpp_profile_enable(smplpp_model * m, int enable)
{
  if(!m) return fail(SMPLPP_ERR_INVALID, "smplpp_profile_enable: null model");
  m->profiling = enable != 0;
  return SMPLPP_OK;
}

extern "C" int smplpp_profile_read(smplpp_model * m, int64_t * launches, double * mean_ms)
{
  if(!m || !launches || !mean_ms) return fail(SMPLPP_ERR_INVALID, "smplpp_profile_read: null argument");
  HIP_TRY(hipSetDevice(m->device));
  double total = 0.0;
  const size_t pairs = m->prof_events.size() / 2;
  size_t good = 0;
  for(size_t i = 0; i < pairs; i++)
  {
    float ms = 0.0f;
    if(hipEventSynchronize(m->prof_events[2 * i + 1]) == hipSuccess &&
       hipEventElapsedTime(&ms, m->prof_events[2 * i], m->prof_events[2 * i + 1]) == hipSuccess)
    {
      total += ms;
      good++;
    }
  }
  for(hipEvent_t e : m->prof_events) (void)hipEventDestroy(e);
  m->prof_events.clear();
  (void)hipGetLastError();
  *launches = (int64_t)good;
  *mean_ms = good ? total / (double)good : 0.0;
  return SMPLPP_OK;
}

// reads and clears the range word of the fp16x2 form (the caller has synchronised the stream)
static int fk_range_status(smplpp_model * m, int * bits)
{
  *bits = 0;
  if(!m->range_flag || m->form != 'h') return SMPLPP_OK;
  HIP_TRY(hipMemcpy(bits, m->range_flag, sizeof(int), hipMemcpyDeviceToHost));
  if(*bits) HIP_TRY(hipMemset(m->range_flag, 0, sizeof(int)));
  return SMPLPP_OK;
}

extern "C" int smplpp_fk_status(smplpp_model * m, int * bits, void * stream)
{
  if(!m || !bits) return fail(SMPLPP_ERR_INVALID, "smplpp_fk_status: null argument");
  HIP_TRY(hipSetDevice(m->device));
  HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
  return fk_range_status(m, bits);
}

extern "C" int smplpp_fk(smplpp_model * m, int64_t n, const float * beta, const float * theta, float * verts, float * joints,
                         float * xforms, float * rest, int space, void * stream)
{
  if(!m) return fail(SMPLPP_ERR_INVALID, "Cannot launch a SMPL model!"); // src/SMPL.cpp:676
  if(n <= 0 || !beta || !theta) return fail(SMPLPP_ERR_INVALID, "Cannot launch a SMPL model!");
  if(space != SMPLPP_HOST && space != SMPLPP_DEVICE) return fail(SMPLPP_ERR_INVALID, "smplpp_fk: bad memory space");
  HIP_TRY(hipSetDevice(m->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  TraceRange tr_fwd("forward SMPL"); // the reference's span around SMPL::launch (node/node.cpp:752-781)
  if(space == SMPLPP_DEVICE) return fk_device(m, n, beta, theta, verts, joints, xforms, rest, nullptr, st);

  Workspace & ws = m->ws;
  const size_t nb = sizeof(float) * (size_t)n * NB, nt = sizeof(float) * (size_t)n * (NJ + 1) * 3;
  const size_t nv = sizeof(float) * (size_t)n * m->V * 3;
  HIP_TRY(ws.beta.reserve(nb));
  HIP_TRY(ws.theta.reserve(nt));
  if(verts) HIP_TRY(ws.verts.reserve(nv));
  if(rest) HIP_TRY(ws.rest.reserve(nv));
  if(joints) HIP_TRY(ws.joints.reserve(sizeof(float) * (size_t)n * NJ * 3));
  if(xforms) HIP_TRY(ws.xf44.reserve(sizeof(float) * (size_t)n * NJ * 16));
  HIP_TRY(hipMemcpyAsync(ws.beta.p, beta, nb, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(ws.theta.p, theta, nt, hipMemcpyHostToDevice, st));
  int rc = fk_device(m, n, ws.beta.as<float>(), ws.theta.as<float>(), verts ? ws.verts.as<float>() : nullptr,
                     joints ? ws.joints.as<float>() : nullptr, xforms ? ws.xf44.as<float>() : nullptr,
                     rest ? ws.rest.as<float>() : nullptr, nullptr, st);
  if(rc) return rc;
  if(verts) HIP_TRY(hipMemcpyAsync(verts, ws.verts.p, nv, hipMemcpyDeviceToHost, st));
  if(rest) HIP_TRY(hipMemcpyAsync(rest, ws.rest.p, nv, hipMemcpyDeviceToHost, st));
  if(joints) HIP_TRY(hipMemcpyAsync(joints, ws.joints.p, sizeof(float) * (size_t)n * NJ * 3, hipMemcpyDeviceToHost, st));
  if(xforms) HIP_TRY(hipMemcpyAsync(xforms, ws.xf44.p, sizeof(float) * (size_t)n * NJ * 16, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  int bits = 0;
  if(int rs = fk_range_status(m, &bits)) return rs;
  if(bits & 1)
    return fail(SMPLPP_ERR_NUMERIC, "smplpp_fk: an operand left the range of the fp16x2 form (|beta| < 1023, relative transforms within 16 x the "
                                    "template's extent): the vertices of such frames are not finite; create the model under SMPLPP_SKIN=b or p");
  return SMPLPP_OK;
}
