// ik_eval_kernel — node.cpp:798-877: residual and analytic forward-mode Jacobian of the tasks of one frame per workgroup (the header
// comment of ik.hip has the overview).  Included by ik.hip only.
#pragma once
#include "ik_types.h"

namespace smplpp_hip
{
// ------------------------------------------------------------------------------------------------ eval kernel
// derivative of Rodrigues (src/BlendShape.cpp:813-841) wrt theta_m, including the ||theta + eps|| angle
__device__ inline void rodrigues_grad_dev(const float * th, int m, float * dR)
{
  const float eps = 1e-8f;
  const float ae0 = th[0] + eps, ae1 = th[1] + eps, ae2 = th[2] + eps;
  const float a = sqrtf(ae0 * ae0 + ae1 * ae1 + ae2 * ae2);
  const float s = sinf(a), c = cosf(a);
  const float k[3] = {th[0] / a, th[1] / a, th[2] / a};
  const float K[9] = {0.f, -k[2], k[1], k[2], 0.f, -k[0], -k[1], k[0], 0.f};
  const float aem = (m == 0) ? ae0 : (m == 1 ? ae1 : ae2);
  const float da = aem / a;
  float dk[3];
  for(int x = 0; x < 3; x++) dk[x] = ((x == m) ? 1.0f : 0.0f) / a - th[x] * da / (a * a);
  const float dK[9] = {0.f, -dk[2], dk[1], dk[2], 0.f, -dk[0], -dk[1], dk[0], 0.f};
  for(int r = 0; r < 3; r++)
    for(int cc = 0; cc < 3; cc++)
    {
      float kk = 0.f, d1 = 0.f, d2 = 0.f;
      for(int q = 0; q < 3; q++)
      {
        kk += K[r * 3 + q] * K[q * 3 + cc];
        d1 += dK[r * 3 + q] * K[q * 3 + cc];
        d2 += K[r * 3 + q] * dK[q * 3 + cc];
      }
      dR[r * 3 + cc] = dK[r * 3 + cc] * s + K[r * 3 + cc] * c * da + (d1 + d2) * (1.0f - c) + kk * s * da;
    }
}

// d normalize(x) = (dx - n (n . dx)) / max(||x||, 1e-12)
__device__ inline void dnormalize_dev(const float * x, const float * dx, float * dn)
{
  float nrm = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  nrm = fmaxf(nrm, 1e-12f);
  const float n0 = x[0] / nrm, n1 = x[1] / nrm, n2 = x[2] / nrm;
  const float d = n0 * dx[0] + n1 * dx[1] + n2 * dx[2];
  dn[0] = (dx[0] - n0 * d) / nrm;
  dn[1] = (dx[1] - n1 * d) / nrm;
  dn[2] = (dx[2] - n2 * d) / nrm;
}

// the same derivative with ONE reciprocal (v_rcp_f32, 1 ulp) instead of six IEEE divisions (~10 instructions each): for
// Jacobian entries only — values that enter the residual keep the reference's x / norm
__device__ inline void dnormalize_jac(const float * x, const float * dx, float * dn)
{
  float nrm = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  nrm = fmaxf(nrm, 1e-12f);
  const float inv = __builtin_amdgcn_rcpf(nrm);
  const float n0 = x[0] * inv, n1 = x[1] * inv, n2 = x[2] * inv;
  const float d = n0 * dx[0] + n1 * dx[1] + n2 * dx[2];
  dn[0] = (dx[0] - n0 * d) * inv;
  dn[1] = (dx[1] - n1 * d) * inv;
  dn[2] = (dx[2] - n2 * d) * inv;
}

__device__ inline void actual_normal_dev(const ModelView & mv, const float * verts, int face, const float * w, float * nn)
{
  float acc[3] = {0.f, 0.f, 0.f};
  for(int i = 0; i < 3; i++) // src/IkTask.cpp:78-84
  {
    float vn[3];
    vertex_normal_dev(verts, mv.faces, mv.adjOff, mv.adjFace, mv.faces[face * 3 + i], vn);
    acc[0] += w[i] * vn[0];
    acc[1] += w[i] * vn[1];
    acc[2] += w[i] * vn[2];
  }
  normalize3(acc);
  nn[0] = acc[0];
  nn[1] = acc[1];
  nn[2] = acc[2];
}

// the same two evaluations from vertex normals computed once (the three normals do not depend on the weights; each costs
// ~30 dependent gathers)
__device__ inline void actual_normal_vn(const float * vn /*[3][3]*/, const float * w, float * nn)
{
  float acc[3] = {0.f, 0.f, 0.f};
  for(int i = 0; i < 3; i++)
  {
    acc[0] += w[i] * vn[i * 3];
    acc[1] += w[i] * vn[i * 3 + 1];
    acc[2] += w[i] * vn[i * 3 + 2];
  }
  normalize3(acc);
  nn[0] = acc[0];
  nn[1] = acc[1];
  nn[2] = acc[2];
}
__device__ inline void actual_pos_vn(const ModelView & mv, const float * verts, int face, const float * w, float off, const float * vn,
                                     float * p)
{
  for(int x = 0; x < 3; x++) // src/IkTask.cpp:64
  {
    float s = 0.f;
    for(int i = 0; i < 3; i++) s += verts[3 * mv.faces[face * 3 + i] + x] * w[i];
    p[x] = s;
  }
  if(off > 0.0f) // :66-69
  {
    float nn[3];
    actual_normal_vn(vn, w, nn);
    p[0] += off * nn[0];
    p[1] += off * nn[1];
    p[2] += off * nn[2];
  }
}

// ... and with the triangle's vertices already at hand (src/IkTask.cpp:64-69)
__device__ inline void actual_pos_tri(const float * tri /*[3][3]*/, const float * w, float off, const float * vn, float * p)
{
  for(int x = 0; x < 3; x++)
  {
    float s = 0.f;
    for(int i = 0; i < 3; i++) s += tri[i * 3 + x] * w[i];
    p[x] = s;
  }
  if(off > 0.0f)
  {
    float nn[3];
    actual_normal_vn(vn, w, nn);
    p[0] += off * nn[0];
    p[1] += off * nn[1];
    p[2] += off * nn[2];
  }
}

__device__ inline void actual_pos_dev(const ModelView & mv, const float * verts, int face, const float * w, float off, float * p)
{
  for(int x = 0; x < 3; x++) // src/IkTask.cpp:64
  {
    float s = 0.f;
    for(int i = 0; i < 3; i++) s += verts[3 * mv.faces[face * 3 + i] + x] * w[i];
    p[x] = s;
  }
  if(off > 0.0f) // :66-69
  {
    float nn[3];
    actual_normal_dev(mv, verts, face, w, nn);
    p[0] += off * nn[0];
    p[1] += off * nn[1];
    p[2] += off * nn[2];
  }
}

// LDS carve-up (floats) of ik_eval_kernel
constexpr int L_R = 0;                         // [24][9]
constexpr int L_J = L_R + NJ * 9;              // [24][3]
constexpr int L_G = L_J + NJ * 3;              // [24][12]  relative transforms [A | b]
constexpr int L_T = L_G + NJ * 12;             // [24][3]   local translations j_i - j_p(i)
constexpr int DRS = 28;                        // (27 + one pad word: seven 16-byte reads fetch a joint's three derivative matrices)
constexpr int L_DR = L_T + NJ * 3;             // [24][DRS]  d R_j / d theta_(j, axis a) at [j][9 a + e]
// d[A_i | b_i]/d theta_c is non-zero only when joint(c) is an ancestor of i (or i itself), and a joint has exactly one
// ancestor per depth: the table keeps, per joint, three columns per DEPTH (column slot 3 depth(joint(c)) + axis(c)) instead of
// all 72 — 31 KB (nine levels) instead of 83 KB of LDS, which is what lets six tasks with a normal term share the ring buffers below.
constexpr int DMAX = TREE_DMAX;                     // deepest kinematic tree served (SMPL: 9 levels); smplpp_ik_create checks
constexpr int L_DAB = L_DR + NJ * DRS;          // [24][CS][3][4]  per (joint, column slot): rows [dA_r | db_r] (one 16-byte LDS access per row)
constexpr int RVS = 28;                         // floats per ring vertex: rest 3 | Ablend 9 | wsum 1 | posed 3 | weights 4 | joints 4 | their ancestor masks 4
// The rest of the plan depends on three sizes the kernel is instantiated for (EvalPlan below):
//   DM   tree levels served: CS = 3 DM column slots per joint in the chain-derivative table
//   RC   ring vertices a task group can hold (one task: at most MAXRING = 40; six tasks on a valence-6 mesh: 72)
//   NG   tasks with a normal term / offset per group (their rings share the ring buffers)
// <9, 76, 6> (trees of SMPL's depth: a 6-target solve with normal terms is ONE group, 158 KB of the CU's 160) and
// <12, 64, 3> (deeper trees: the table takes 10 KB more).
template<int DM, int RC, int NG>
struct EvalPlan
{
  static constexpr int CS = 3 * DM;
  static constexpr int L_DBB = L_DAB + NJ * 12 * CS; // [24*3][10]
  static constexpr int L_RV = L_DBB + NJ * 3 * NB;   // [RC][RVS]
  static constexpr int L_DP = L_RV + RC * RVS;       // [RC][3][NQ]
  static constexpr int L_VN = L_DP + RC * 3 * NQ;    // per normal task of the group: [3][3] vertex normals + [3] their weighted sum
  static constexpr int L_END = L_VN + 12 * NG;
};
constexpr int L_ANC_BYTES = NJ * 4;            // int anc[24] (ancestor bit masks; depth = popcount - 1) after the float region

// Cross-stream hand-over through a device flag (the other stream waits with hipStreamWaitValue32): every workgroup of the
// producing kernel ends here; the last one to arrive publishes `tick`.  Measured on MI355X (tools/micro/waitvalue_cost.hip,
// join_cost.hip): the waiting stream's next kernel starts 1.4 us after the flag is written, against 11.6 us after an event
// recorded by the producer's stream fires (3.7 us when that event had fired more than 10 us before the waiter arrived).
// What the consumer kernels read from the producer is written with st_agent (write-through to the device's coherence point),
// so a workgroup only has to wait for its own stores: a device-scope release fence per workgroup would write back the whole
// L2 of its XCD 256 times per kernel — including the lines of the kernel running beside it (the fused FK kernel went from 17
// to 28 us that way).
template<int DMAX, int RCAP, int NGN, int MADJ>
__device__ __forceinline__ void ik_eval_body(const ModelView & mv, const TaskArrays & ta, const float * __restrict__ theta25,
                                             const float * __restrict__ verts_all, const float * __restrict__ rest_all,
                                             const float * __restrict__ Gp, const float * __restrict__ joints,
                                             const float * __restrict__ poserot, int K, int optimize_beta,
                                             int phi_live, int min_valid, float * __restrict__ pos804,
                                             double * __restrict__ e_out, double * __restrict__ J_out,
                                             int * __restrict__ skip, int dbg_stop, int tsplit, const int32_t * __restrict__ roles,
                                             const float * __restrict__ vjac, double * __restrict__ Jl_out)
{
  typedef EvalPlan<DMAX, RCAP, NGN> Plan;
  // MADJ: adjacent faces per vertex the normal Jacobian's tables hold (the model's: 12, or 16 for a topology with a vertex of
  // 13..16 faces — smplpp_model::madj; the per-face tables faceRing / faceMap are built with the same strides)
  constexpr int MRING = 3 * (MADJ + 1) + 1; // distinct vertices a task can touch
  constexpr int CS = Plan::CS, L_DBB = Plan::L_DBB, L_RV = Plan::L_RV, L_DP = Plan::L_DP, L_VN = Plan::L_VN, L_END = Plan::L_END;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ int s_tree[TREE_SIZE];
  __shared__ int s_par[NJ];
  __shared__ float s_gw[NJ * 3];          // world joint positions
  __shared__ uint8_t s_ancat[NJ * DMAX];  // ancestor of joint i at depth d
  static_assert(EVAL_NT >= 256 + NJ * DMAX, "one thread per (joint, depth) of the ancestor table");
  const int * sAnc = s_tree + TREE_ANC;
  const int nlev = mv.nlev;
  // grid = n * tsplit: when frames are fewer than CUs (mocap chains: 8 per GPU x 41 markers) a frame's tasks are split over
  // tsplit workgroups, each rebuilding the frame's derivative tables (4 us) for its contiguous share of the tasks
  const int64_t f = blockIdx.x / tsplit;
  const int part = (int)(blockIdx.x % tsplit);

  const int tid = threadIdx.x;
  if(dbg_stop == 19) return; // (timing experiments only: the launch itself — 5.7 us of the kernel's 24 at 8 chains, tools/eval_stops.sh)
  EVAL_STAMP(0);
  // this thread's entries of the chain-derivative table (dealt round-robin by smplpp_ik_create from the tree: the live ones fill the
  // first slots): joint | parent << 5 | column slot << 10 | row << 16 | (the column's joint is the joint itself) << 18; -1: none
  int role[DMAX];
#pragma unroll
  for(int L = 0; L < DMAX; L++) role[L] = roles[L * EVAL_NT + tid];
  const int nq = TD75 + (optimize_beta ? NB : 0);
  const int D = TD75 + 2 * K + (optimize_beta ? NB : 0);
  const float * verts = verts_all + f * mv.V * 3;
  const float * rest = rest_all + f * mv.V * 3;
  const int64_t tb = f * K; // task base
  const int per_part = (K + tsplit - 1) / tsplit;
  const int k_begin = part * per_part, k_end = (k_begin + per_part < K) ? k_begin + per_part : K;

  // ---- what phase A will read of the tasks is requested NOW: its addresses depend on nothing the kernel computes, and the
  // dependent pair face id -> ring list (two round trips) then runs beside the set-up and the chain-derivative steps instead
  // of in front of phase A.  (The re-projection that wrote faces, weights and targets has been waited for by the stream.)
  const int ntask = (k_end > k_begin) ? k_end - k_begin : 0; // (a part beyond the last task — more parts than tasks — has none)
  const bool a0_live = tid < ntask * (MRING + 1); // A0's first pass: one (task, ring-list word) per thread
  const int a0_t = a0_live ? tid / (MRING + 1) : 0, a0_q = a0_live ? tid % (MRING + 1) : 0;
  // (a workgroup whose share of the tasks is empty — more parts than tasks — or a thread without an item reads task 0 of its
  // frame: every address requested here lies inside the task arrays)
  const int64_t a0_k = a0_live ? tb + k_begin + a0_t : tb;
  const float a0_noff = ta.noff[a0_k], a0_nrmw = ta.nrmw[a0_k];
  const int a0_face = ta.face[a0_k];
  const bool a3_live = tid < ntask; // A3: one task per thread
  const int64_t a3_k = a3_live ? tb + k_begin + tid : tb;
  const float a3_off = ta.noff[a3_k], a3_wp = ta.posw[a3_k], a3_wn = ta.nrmw[a3_k];
  const float a3_w[3] = {ta.vw[a3_k * 3], ta.vw[a3_k * 3 + 1], ta.vw[a3_k * 3 + 2]};
  const float a3_tp[3] = {ta.tpos[a3_k * 3], ta.tpos[a3_k * 3 + 1], ta.tpos[a3_k * 3 + 2]};
  const float a3_tn[3] = {ta.tnrm[a3_k * 3], ta.tnrm[a3_k * 3 + 1], ta.tnrm[a3_k * 3 + 2]};
  const uint16_t a0_e = mv.faceRing[(int64_t)a0_face * (MRING + 1) + a0_q];

  // ---- set-up: every global load first (one round trip), then the frame constants into LDS
  static_assert(EVAL_NT >= NJ * 12 && IK_MAXK <= 64, "one element of each frame constant per thread; validity by one ballot");
  __shared__ int s_valid;
  {
    const float pw = (tid < K) ? ta.posw[tb + tid] : 0.0f;
    const float vR = (tid < NJ * 9) ? poserot[f * NJ * 9 + tid] : 0.0f;
    const float vJ = (tid < NJ * 3) ? joints[f * NJ * 3 + tid] : 0.0f;
    const float vG = (tid < NJ * 12) ? Gp[f * NJ * 12 + tid] : 0.0f;
    // tree tables of the model (common.h TREE_*): ancestor masks (with the joint itself; depth(i) = popcount - 1), joints by level
    const int vT = (tid < TREE_SIZE) ? mv.anc[tid] : 0;
    const int vP = (tid < NJ) ? mv.parent[tid] : 0;
    float th[3] = {0.f, 0.f, 0.f};
    if(tid < 72)
      for(int x = 0; x < 3; x++) th[x] = theta25[(f * (NJ + 1) + 1 + tid / 3) * 3 + x];
    // node.cpp:785 — a frame with too few valid markers skips the whole solve block (no task refresh either)
    if(tid < 64)
    {
      const unsigned long long m = __ballot(tid < K && pw > 0.0f);
      if(tid == 0) s_valid = __popcll(m);
    }
    if(tid < NJ * 9) lds[L_R + tid] = vR;
    if(tid < NJ * 3) lds[L_J + tid] = vJ;
    if(tid < NJ * 12) lds[L_G + tid] = vG;
    if(tid < TREE_SIZE) s_tree[tid] = vT;
    if(tid < NJ) s_par[tid] = vP;
    __syncthreads();
    const int sk = (s_valid < min_valid) ? 1 : 0;
    if(tid == 0) st_agent(&skip[f], sk); // (read by kernels of the other stream: see wg_signal)
    if(sk) return;
    if(tid < NJ * 3)
    {
      const int j = tid / 3, x = tid % 3, p = s_par[j];
      lds[L_T + tid] = (j == 0) ? lds[L_J + x] : lds[L_J + tid] - lds[L_J + p * 3 + x];
    }
    if(tid < 72)
    {
      float dR[9];
      rodrigues_grad_dev(th, tid % 3, dR);
      for(int q = 0; q < 9; q++) lds[L_DR + (tid / 3) * DRS + (tid % 3) * 9 + q] = dR[q];
    }
    else if(tid >= 128 && tid < 128 + NJ * 3) // world position of joint j: g_j = b_j + A_j . rest joint_j (the relative transform undone)
    {
      const int j = (tid - 128) / 3, x = (tid - 128) % 3;
      const float * G = lds + L_G + j * 12 + x * 4;
      s_gw[j * 3 + x] = G[3] + ((G[0] * lds[L_J + j * 3] + G[1] * lds[L_J + j * 3 + 1]) + G[2] * lds[L_J + j * 3 + 2]);
    }
    else if(tid >= 256 && tid < 256 + NJ * DMAX) // the ancestor of joint i at depth d (i itself at its own depth; none below it)
    {
      const int i = (tid - 256) / DMAX, d = (tid - 256) % DMAX;
      int j = i;
      const int dep = __popc(s_tree[TREE_ANC + i]) - 1;
      for(int q = dep; q > d; q--) j = s_par[j];
      s_ancat[i * DMAX + d] = (uint8_t)(d <= dep ? j : 0);
    }
    __syncthreads();
  }

  EVAL_STAMP(1);
  if(dbg_stop == 20) return; // (timing experiments only: SMPLPP_IK_DBG_STOP)
  // ---- chain derivatives.  Entry (joint i, ancestor depth da, axis ax, row r) = row r of
  // d[A_i | b_i]/d theta_c for c = (the ancestor a of i at depth da, axis ax).  CLOSED FORM (round 4; rounds 1-3 advanced one tree
  // level per barrier-separated step, each entry from the same entry of i's parent: nine steps of an LDS round trip): with A the
  // world rotations and g the world joint positions, A_i = A_a (R ... R_i) for every descendant i of a, hence
  //     dA_i = A_p(a) dR_a A_a' A_i = Omega A_i,   d g_i = Omega (g_i - g_a),   Omega = W A_a',  W = A_p(a) dR_(a, ax)
  // — one 3 x 3 per (ancestor, axis), then every entry on its own: two barriers instead of nine.  The joint's own column keeps
  // dA_a = W (Omega A_a up to rounding), the root's dR itself.  The beta columns (item 5: joints move, rotations do not; d g_i /
  // d beta_k = A_p . dt_i + d g_p is a sum along the path) keep their level-by-level steps, only where beta is optimised.
  {
    float * dgl = lds + L_DP;              // [24][3][NB] running d g of the beta columns (the dp buffer is unused until phase B)
    float * om = lds + L_DP + NJ * 3 * NB; // [72 (ancestor, axis)][3 rows][8]: W (3) | Omega (3) | Omega . g_a | -
    if(tid < NJ * 9)
    {
      const int a = tid / 9, ax = (tid / 3) % 3, r = tid % 3, p = s_par[a];
      const float * M = lds + L_DR + a * DRS + ax * 9;
      float W[3];
      if(a == 0) // (the reference's dA_0 = dR itself: no products with a unit row's zeros)
      {
        W[0] = M[r * 3];
        W[1] = M[r * 3 + 1];
        W[2] = M[r * 3 + 2];
      }
      else
      {
        const float * x = lds + L_G + p * 12 + r * 4;
        W[0] = x[0] * M[0] + x[1] * M[3] + x[2] * M[6];
        W[1] = x[0] * M[1] + x[1] * M[4] + x[2] * M[7];
        W[2] = x[0] * M[2] + x[1] * M[5] + x[2] * M[8];
      }
      const float * Aa = lds + L_G + a * 12;
      float O[3];
#pragma unroll
      for(int c = 0; c < 3; c++) O[c] = W[0] * Aa[c * 4] + W[1] * Aa[c * 4 + 1] + W[2] * Aa[c * 4 + 2];
      float * o = om + tid * 8;
      *reinterpret_cast<float4 *>(o) = make_float4(W[0], W[1], W[2], O[0]);
      *reinterpret_cast<float4 *>(o + 4) = make_float4(O[1], O[2], O[0] * s_gw[a * 3] + O[1] * s_gw[a * 3 + 1] + O[2] * s_gw[a * 3 + 2], 0.0f);
    }
    __syncthreads();
    // (role[u]: this thread's u-th entry, dealt round-robin by smplpp_ik_create: the live ones fill the first slots)
#pragma unroll
    for(int u = 0; u < DMAX; u++)
    {
      if(role[u] >= 0)
      {
        const int i = role[u] & 31, cs = (role[u] >> 10) & 63, r = (role[u] >> 16) & 3;
        const bool self = (role[u] >> 18) & 1;
        const int a = s_ancat[i * DMAX + cs / 3];
        const float * o = om + ((a * 3 + cs % 3) * 3 + r) * 8;
        const float4 o0 = *reinterpret_cast<const float4 *>(o), o1 = *reinterpret_cast<const float4 *>(o + 4);
        const float * Ai = lds + L_G + i * 12;
        float dA[3], dg;
        if(self)
        {
          dA[0] = o0.x;
          dA[1] = o0.y;
          dA[2] = o0.z;
          dg = 0.0f;
        }
        else
        {
#pragma unroll
          for(int c = 0; c < 3; c++) dA[c] = o0.w * Ai[c] + o1.x * Ai[4 + c] + o1.y * Ai[8 + c];
          dg = (o0.w * s_gw[i * 3] + o1.x * s_gw[i * 3 + 1] + o1.y * s_gw[i * 3 + 2]) - o1.z;
        }
        const float ji0 = lds[L_J + i * 3], ji1 = lds[L_J + i * 3 + 1], ji2 = lds[L_J + i * 3 + 2];
        *reinterpret_cast<float4 *>(lds + L_DAB + ((i * CS + cs) * 3 + r) * 4) =
            make_float4(dA[0], dA[1], dA[2], dg - (dA[0] * ji0 + dA[1] * ji1 + dA[2] * ji2));
      }
    }
    if(optimize_beta) // (uniform)
    {
      // (beside the chain entries' threads when the workgroup is large enough, else sharing threads with them)
      constexpr int BETA_T0 = (EVAL_NT >= 512 + NJ * NB) ? 512 : EVAL_NT - 256;
      static_assert(BETA_T0 >= 0 && BETA_T0 + NJ * NB <= EVAL_NT, "the beta columns take NJ * NB threads from BETA_T0");
      // thread (joint i, k) works at the joint's level; its regressor rows are loaded ahead of the steps
      const bool isb = tid >= BETA_T0 && tid < BETA_T0 + NJ * NB;
      const int bi = isb ? (tid - BETA_T0) / NB : 0, bk = isb ? (tid - BETA_T0) % NB : 0, bp = s_par[bi];
      const int blev = isb ? __popc(sAnc[bi]) - 1 : -1;
      float dj[3] = {0.f, 0.f, 0.f}, dt[3] = {0.f, 0.f, 0.f};
      if(isb)
        for(int x = 0; x < 3; x++)
        {
          dj[x] = mv.JS[(bi * 3 + x) * NB + bk];
          dt[x] = (bi == 0) ? dj[x] : dj[x] - mv.JS[(bp * 3 + x) * NB + bk];
        }
      for(int L = 0; L < nlev; L++)
      {
        if(blev == L)
        {
          float dgi[3];
          if(bi == 0)
            for(int x = 0; x < 3; x++) dgi[x] = dt[x];
          else
          {
            const float * Ap = lds + L_G + bp * 12;
            for(int r = 0; r < 3; r++) dgi[r] = (Ap[r * 4] * dt[0] + Ap[r * 4 + 1] * dt[1] + Ap[r * 4 + 2] * dt[2]) + dgl[(bp * 3 + r) * NB + bk];
          }
          const float * Ai = lds + L_G + bi * 12;
          for(int r = 0; r < 3; r++)
          {
            dgl[(bi * 3 + r) * NB + bk] = dgi[r];
            lds[L_DBB + (bi * 3 + r) * NB + bk] = dgi[r] - (Ai[r * 4] * dj[0] + Ai[r * 4 + 1] * dj[1] + Ai[r * 4 + 2] * dj[2]);
          }
        }
        __syncthreads();
      }
    }
    __syncthreads();
  }

  EVAL_STAMP(2);
  if(dbg_stop == 21) return;
  // ---- phase A, in four steps so that nothing walks dependent HBM gathers serially:
  //   A0  all threads: the ring lists from the per-face tables built with the model (topology only)
  //   A1  all threads: posed positions of the ring vertices -> LDS
  //   A2  one thread per (task, triangle vertex): vertex normal from those positions (tasks with a normal offset / term)
  //   A3  one thread per task: tangents, weight refresh, residual rows (node.cpp:803-820)
  __shared__ float s_vn[IK_MAXK][9];
  __shared__ uint16_t s_ringb[IK_MAXK][MRING + 1]; // (vertex ids fit 16 bits: smplpp_ik_create checks V)
  __shared__ uint8_t s_usen[IK_MAXK];               // the task differentiates a normal (normal term or normal offset)
  __shared__ int s_facel[IK_MAXK];                  // the task's face
  __shared__ uint8_t s_acnt[IK_MAXK][4];            // faces around each of its three vertices (<= 255: a larger count takes the general routine either way)
  // posed positions of the ring vertices of every task of this workgroup: [task][MRING][3], in the dp buffer of phase B
  // (free until then)
  static_assert(IK_MAXK * MRING * 3 <= RCAP * 3 * NQ, "s_rpos must fit the L_DP region");
  float(*s_rpos)[MRING][3] = reinterpret_cast<float(*)[MRING][3]>(lds + L_DP);
  // A0: ring lists from the per-face tables built with the model (topology only); the first pass from the words requested at
  // the kernel's start
  for(int item = tid; item < ntask * (MRING + 1); item += EVAL_NT)
  {
    const int t = item / (MRING + 1), q = item % (MRING + 1);
    const int k = k_begin + t;
    const bool first = item < EVAL_NT; // (item == tid)
    const bool use_normal = first ? ((a0_noff > 0.0f) || (a0_nrmw > 0.0f)) : ((ta.noff[tb + k] > 0.0f) || (ta.nrmw[tb + k] > 0.0f));
    const int face = first ? a0_face : ta.face[tb + k];
    const uint16_t e = first ? a0_e : mv.faceRing[(int64_t)face * (MRING + 1) + q];
    // slots 0..2 = the face's own vertices; with a normal term / offset also the distinct vertices of the faces around them
    s_ringb[t][q] = (q == 0 && !use_normal) ? (uint16_t)3 : e;
    if(q == 0)
    {
      s_usen[t] = use_normal ? 1 : 0;
      s_facel[t] = face; // (phase B's table loads start from LDS, not from another dependent HBM read)
    }
  }
  __syncthreads();
  __shared__ int s_rcum[IK_MAXK + 1]; // ring sizes of the workgroup's tasks, cumulated (offsets of the groups' ring buffers)
  __shared__ int s_gk[IK_MAXK + 1], s_ng; // phase B's task groups: first task of each (relative to k_begin), their number
  static_assert(IK_MAXK < 64, "one wavefront scans the ring sizes");
  if(tid < 64)
  {
    const int cnt_l = (tid < ntask) ? (int)s_ringb[tid][0] : 0, usen_l = (tid < ntask) ? (int)s_usen[tid] : 0;
    int incl = cnt_l;
    for(int o = 1; o < 64; o <<= 1)
    {
      const int up = __shfl_up(incl, o, 64);
      if(tid >= o) incl += up;
    }
    if(tid < ntask) s_rcum[tid + 1] = incl;
    if(tid == 0) s_rcum[0] = 0;
    // task groups of phase B, greedy by ring size: tasks with a normal term or offset (ring: the face's vertices and those of
    // their adjacent faces) go NGN to a group when their rings fit the LDS buffers together, position-only tasks (ring 3) as
    // many as fit; the two kinds are not mixed.  The wavefront walks the tasks in step, sizes out of registers (v_readlane).
    int ng = 0, k = 0;
    while(k < ntask)
    {
      k = __builtin_amdgcn_readfirstlane(k);
      if(tid == 0) s_gk[ng] = k;
      ng++;
      const int gn = __builtin_amdgcn_readlane(usen_l, k);
      int tot = 0, k2 = k;
      while(k2 < ntask)
      {
        k2 = __builtin_amdgcn_readfirstlane(k2);
        const int nrk = __builtin_amdgcn_readlane(cnt_l, k2), un2 = __builtin_amdgcn_readlane(usen_l, k2);
        if(k2 > k && (tot + nrk > RCAP || un2 != gn || (gn && k2 - k >= (dbg_stop == 40 ? 1 : NGN)))) break; // (40: dev switch, one per group)
        tot += nrk;
        k2++;
      }
      k = k2;
    }
    if(tid == 0)
    {
      s_gk[ng] = ntask;
      s_ng = ng;
    }
  }
  EVAL_STAMP(3);
  if(dbg_stop == 23) return;
  for(int item = tid; item < ntask * MRING; item += EVAL_NT) // A1
  {
    const int t = item / MRING, q = item % MRING;
    if(q < s_ringb[t][0])
    {
      const int v = s_ringb[t][1 + q];
      s_rpos[t][q][0] = verts[v * 3];
      s_rpos[t][q][1] = verts[v * 3 + 1];
      s_rpos[t][q][2] = verts[v * 3 + 2];
    }
  }
  __syncthreads();
  EVAL_STAMP(4);
  if(dbg_stop == 24) return;
  // A2: SMPL::calcVertexNormal (src/SMPL.cpp:527-535) with the adjacent faces' corners taken by ring slot — the unit normals
  // of the adjacent faces one thread per (task, triangle vertex, adjacent face), then the uniform sum per vertex in the
  // reference's order
  static_assert((IK_MAXK * MRING * 3 + IK_MAXK * 3 * MADJ * 3) <= RCAP * 3 * NQ, "s_rpos + s_fn must fit the L_DP region");
  float(*s_fn)[3 * MADJ][3] = reinterpret_cast<float(*)[3 * MADJ][3]>(lds + L_DP + IK_MAXK * MRING * 3);
  for(int item = tid; item < ntask * 3 * MADJ; item += EVAL_NT)
  {
    const int t = item / (3 * MADJ), ia = item % (3 * MADJ), i = ia / MADJ, a2 = ia % MADJ;
    if(s_usen[t])
    {
      const int u = s_ringb[t][1 + i];
      // (count and map entry in ONE round trip: the entry exists whether or not the vertex has that many faces)
      const uint8_t * mp = mv.faceMap + (int64_t)s_facel[t] * (3 * MADJ * 3) + ia * 3;
      const int m0 = mp[0], m1 = mp[1], m2 = mp[2];
      const int cnt = mv.adjOff[u + 1] - mv.adjOff[u];
      if(a2 == 0) s_acnt[t][i] = (uint8_t)(cnt < 255 ? cnt : 255); // (the sum below starts from LDS, not from a second round trip; smplpp_ik_create admits at most MADJ)
      if(a2 < cnt && cnt <= MADJ) face_normal_pts(s_rpos[t][m0], s_rpos[t][m1], s_rpos[t][m2], s_fn[t][ia]);
    }
  }
  __syncthreads();
  if(tid < 3 * ntask)
  {
    const int t = tid / 3, i = tid % 3;
    if(s_usen[t])
    {
      const int u = s_ringb[t][1 + i];
      const int cnt = s_acnt[t][i];
      float vn[3];
      if(cnt > MADJ) // more faces than the ring map covers: the general routine for the VALUE; the derivative tables of phase
      {                // B hold MADJ faces per vertex, so the frame is flagged and host-space callers get an error
        vertex_normal_dev(verts, mv.faces, mv.adjOff, mv.adjFace, u, vn);
        atomicOr(&ta.flags[f], 4);
      }
      else
      {
        float sum = 0.0f;
        for(int q = 0; q < cnt; q++) sum += 1.0f;
        const float w = 1.0f / sum;
        float acc[3] = {0.f, 0.f, 0.f};
        for(int a2 = 0; a2 < cnt; a2++)
        {
          const float * fn = s_fn[t][i * MADJ + a2];
          acc[0] += w * fn[0];
          acc[1] += w * fn[1];
          acc[2] += w * fn[2];
        }
        normalize3(acc);
        vn[0] = acc[0];
        vn[1] = acc[1];
        vn[2] = acc[2];
      }
      for(int x = 0; x < 3; x++) s_vn[t][i * 3 + x] = vn[x];
    }
  }
  __syncthreads();
  EVAL_STAMP(5);
  if(dbg_stop == 25) return;
  if(tid < ntask) // A3: every load first, every store last (a load behind a store waits for the store's round trip too)
  {
    const int k = k_begin + tid;
    const float * vnk = s_vn[tid];
    float tri[9];
#pragma unroll
    for(int i = 0; i < 3; i++)
#pragma unroll
      for(int x = 0; x < 3; x++) tri[i * 3 + x] = s_rpos[tid][i][x]; // ring slots 0..2 are the face's own vertices
    const float off = a3_off, wp = a3_wp, wn = a3_wn; // (requested at the kernel's start)
    float w[3] = {a3_w[0], a3_w[1], a3_w[2]};
    const float tp[3] = {a3_tp[0], a3_tp[1], a3_tp[2]};
    const float tn[3] = {a3_tn[0], a3_tn[1], a3_tn[2]};
    // calcTangents (src/IkTask.cpp:33-47)
    float t1[3] = {tri[3] - tri[0], tri[4] - tri[1], tri[5] - tri[2]};
    float t2[3];
    {
      float e2[3] = {tri[6] - tri[0], tri[7] - tri[1], tri[8] - tri[2]};
      float nn[3];
      cross3(t1, e2, nn);
      cross3(nn, t1, t2);
      normalize3(t1);
      normalize3(t2);
    }
    float pos[3];
    actual_pos_tri(tri, w, off, vnk, pos); // the point calcVertexWeights is differentiated at
    triangle_weights_dev(pos, tri, w);     // calcVertexWeights with phi_ == 0 (src/IkTask.cpp:49-57, node.cpp:804)
    float ap[3], an[3] = {0.f, 0.f, 0.f};
    actual_pos_tri(tri, w, off, vnk, ap);
    // the interpolated normal only when a term uses it, as node.cpp:811-819 does; smplpp_ik_get_tasks evaluates it on
    // demand for the others
    if(wn > 0.0f) actual_normal_vn(vnk, w, an);
    double e3 = 0.0; // :819
    if(wn > 0.0f)
    {
      const float dt = (an[0] * tn[0] + an[1] * tn[1]) + an[2] * tn[2];
      e3 = (double)(wn * (dt + 1.0f)); // :813-814
    }
#pragma unroll
    for(int x = 0; x < 3; x++)
    {
      ta.tang[(tb + k) * 6 + x * 2 + 0] = t1[x];
      ta.tang[(tb + k) * 6 + x * 2 + 1] = t2[x];
      pos804[(tb + k) * 3 + x] = pos[x];
      // (write-through like the re-projection's own store to the same word, proj_finish_kernel.  The side stream is forked by
      // the SOLVE kernel's start flag today, i.e. behind this kernel's end-of-kernel write-back, so a plain store would also be
      // ordered; when the fork was raised by this kernel's own flag it was not — a plain store could reach memory after the
      // re-projected weights and overwrite them — and one policy per word stays the rule: two kernels never write a word
      // with different policies)
      st_agent(&ta.vw[(tb + k) * 3 + x], w[x]);
      st_agent(&ta.apos[(tb + k) * 3 + x], ap[x]);
      e_out[(f * K + k) * 4 + x] = (double)(wp * (ap[x] - tp[x])); // node.cpp:807
    }
    e_out[(f * K + k) * 4 + 3] = e3;
    ta.roww[(tb + k) * 2] = wp;
    ta.roww[(tb + k) * 2 + 1] = wn;
    // the re-projection's cull radius when the query point is the actual position (no surface coordinate can move): the
    // exact distance to the task's own face, from the vertices already in registers (same evaluation as the scan's)
    st_agent(&ta.hint[tb + k], tri_sqdist_vals(tri[0], tri[1], tri[2], tri[3], tri[4], tri[5], tri[6], tri[7], tri[8], ap[0], ap[1], ap[2]).x);
  }
  EVAL_STAMP(6);
  if(dbg_stop == 28) return;
  lds_barrier(); // (global stores of this phase stay in flight: nothing reads them before the next full barrier)

  if(dbg_stop == 22) return;
  // ---- phase B: Jacobian rows (node.cpp:823-873).  Tasks are taken in GROUPS whose ring vertices fit the LDS buffers
  // together (a position-only task touches 3 vertices, so a 6-target solve is one group; a task with a normal term
  // touches up to MRING and forms a group of its own): each barrier-separated step then serves the whole group, and the
  // global-memory latencies of the tasks overlap instead of queueing.
  __shared__ int s_roff[IK_MAXK + 1]; // ring offset of task k inside its group's buffers
  __shared__ int s_rvert[RCAP];         // ring slot -> vertex
  __shared__ uint8_t s_map[NGN][3 * MADJ * 3]; // (vertex of the face, adjacent face, corner) -> slot in the task's ring, per normal task
  __shared__ int s_cnt[NGN][3];         // adjacent-face count of the face's three vertices
  __shared__ float s_dvn[NGN][NQ * 3 * 3]; // per (column, triangle vertex): derivative of the vertex normal (the normal itself, the same for every column: L_VN)
  __shared__ __attribute__((aligned(16))) float s_geo[NGN][3 * MADJ][12]; // per adjacent face of a triangle vertex: unit normal, |cross|, edges e1, e2
  constexpr int MAPN = 3 * MADJ * 3; // ring-slot map entries per normal task
  static_assert(RCAP + NGN * 3 <= 96 && 96 + NGN * MAPN <= EVAL_NT, "B1 hands the count / map loads to thread ranges beyond the ring threads");
  for(int g = 0; g < s_ng; g++)
  {
    // group [k_lo, k_hi) from the list thread 0 made behind the ring-size scan (with 12 wavefronts, bounds every thread
    // works out for itself cost the workgroup 12 issue slots per instruction)
    const int k_lo = k_begin + s_gk[g], k_hi = k_begin + s_gk[g + 1];
    const int total = s_rcum[k_hi - k_begin] - s_rcum[k_lo - k_begin];
    const bool grp_normal = s_usen[k_lo - k_begin] != 0;
    const int ngn = grp_normal ? k_hi - k_lo : 0; // normal tasks of this group (their index in the group: k - k_lo)
    {
      // ring tables of the group: offsets from the cumulated sizes, one thread per (task, ring slot)
      const int gbase = s_rcum[k_lo - k_begin];
      for(int item = tid; item < (k_hi - k_lo) * MRING; item += EVAL_NT)
      {
        const int kk = k_lo + item / MRING, i = item % MRING;
        const uint16_t * rg = s_ringb[kk - k_begin];
        const int off0 = s_rcum[kk - k_begin] - gbase;
        if(i == 0) s_roff[kk] = off0;
        if(i < rg[0]) s_rvert[off0 + i] = rg[1 + i];
      }
    }
    lds_barrier(); // (global stores of this phase stay in flight: nothing reads them before the next full barrier)
    if(k_lo == k_begin) EVAL_STAMP(8);
    // VPoser latent layout: this frame's d(vposer out)/dz [63][32] is requested HERE and dropped into LDS behind B2 — in front of
    // the first row stores of the group: loads and stores retire through one counter, and a load consumed behind B3's stores waited
    // for every one of them to be acknowledged (4 k cycles per group)
    constexpr int VJ_PER = (63 * 32 + EVAL_NT - 1) / EVAL_NT;
    // (the deep-tree plan's ring-vertex region is too small for it: there it goes into the vertex-normal derivatives' behind B3)
    constexpr bool SVJ_EARLY = RCAP * RVS >= 63 * 32;
    static_assert(SVJ_EARLY || NGN * NQ * 9 >= 63 * 32, "a place for the decoder Jacobian");
    float vjr[VJ_PER];
    if(Jl_out)
    {
#pragma unroll
      for(int u = 0; u < VJ_PER; u++) vjr[u] = (tid + u * EVAL_NT < 63 * 32) ? vjac[f * 63 * 32 + tid + u * EVAL_NT] : 0.0f;
    }

    if(tid < total) // B1: per ring vertex rest position, blended rotation, blended w
    {
      const int v = s_rvert[tid];
      float * rv = lds + L_RV + tid * RVS;
      rv[0] = rest[v * 3];
      rv[1] = rest[v * 3 + 1];
      rv[2] = rest[v * 3 + 2];
      float Ab[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      for(int m = 0; m < mv.maxw; m++)
      {
        const float wm = mv.wVal[(int64_t)v * mv.maxw + m];
        const float * A = lds + L_G + mv.wIdx[(int64_t)v * mv.maxw + m] * 12;
        for(int r = 0; r < 3; r++)
          for(int cc = 0; cc < 3; cc++) Ab[r * 3 + cc] += wm * A[r * 4 + cc];
      }
      for(int q = 0; q < 9; q++) rv[3 + q] = Ab[q];
      rv[12] = mv.wSum[v];
      for(int m = 0; m < 4; m++) // the (first four) skinning weights and joints, so that B2 does not re-read them per column
      {
        rv[16 + m] = (m < mv.maxw) ? mv.wVal[(int64_t)v * mv.maxw + m] : 0.0f;
        const int jm = (m < mv.maxw) ? (int)mv.wIdx[(int64_t)v * mv.maxw + m] : 0;
        rv[20 + m] = __int_as_float(jm);
        rv[24 + m] = __int_as_float(sAnc[jm]);
      }
      rv[13] = verts[v * 3]; // posed position: the normal chain of B3 reads its triangles from here, not from HBM
      rv[14] = verts[v * 3 + 1];
      rv[15] = verts[v * 3 + 2];
    }
    else if(tid >= 96 && tid < 96 + NGN * MAPN) // ring-slot maps of the group's normal tasks
    {
      const int gi = (tid - 96) / MAPN, j = (tid - 96) % MAPN;
      if(gi < ngn) s_map[gi][j] = mv.faceMap[(int64_t)s_facel[k_lo - k_begin + gi] * MAPN + j];
    }
    else if(tid >= RCAP && tid < RCAP + NGN * 3)
    {
      const int gi = (tid - RCAP) / 3, j = (tid - RCAP) % 3;
      if(gi < ngn)
      {
        const int u = s_ringb[k_lo - k_begin + gi][1 + j]; // ring slots 0..2: the face's own vertices, in its order
        s_cnt[gi][j] = mv.adjOff[u + 1] - mv.adjOff[u];
      }
    }
    __syncthreads();
    if(k_lo == k_begin) EVAL_STAMP(9);
    // B2: dp[rv][:, q]  (SURVEY.md §9 items 1-5).  One thread per (ring vertex, column group): the root translation triple,
    // one joint's three rotation columns (they share the vertex's weights, its rest position and the 27 pose-corrective
    // coefficients of that joint: loaded once instead of once per column), or one beta column.
    const int ngrp = 1 + NJ + (nq - TD75);
    for(int item = tid; item < total * ngrp; item += EVAL_NT)
    {
      const int r_ = item / ngrp, g = item - r_ * ngrp;
      const int v = s_rvert[r_];
      const float * rv = lds + L_RV + r_ * RVS;
      float * dpv = lds + L_DP + (r_ * 3) * NQ; // row r, column q: dpv[r * NQ + q]
      // the ring vertex's record in 16-byte words (the LDS pipe is what this phase is pressed against: instruction count matters)
      const float4 * rv4 = reinterpret_cast<const float4 *>(rv);
      const float4 q0 = rv4[0], q1 = rv4[1], q2 = rv4[2], qw = rv4[4], qj = rv4[5], qa = rv4[6];
      const float Ab[9] = {q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w}; // the blended rotation, row-major
      const float wsum = rv[12];
      const float iws = __builtin_amdgcn_rcpf(wsum); // (the constant homogeneous divide of SURVEY.md §9 item 4 as a reciprocal: Jacobian entries only)
      if(g == 0) // root translation: identity
      {
        for(int q = 0; q < 3; q++)
          for(int r = 0; r < 3; r++) dpv[r * NQ + q] = (r == q) ? 1.0f : 0.0f; // (wsum / wsum)
      }
      else if(g <= NJ)
      {
        const int jc = g - 1;
        const int cs0 = 3 * (__popc(sAnc[jc]) - 1), jbit = 1 << jc;
        const bool wlds = mv.maxw <= 4;
        // the 27 pose-corrective coefficients of (vertex, joint) are requested FIRST: their round trip (the item loop makes three
        // of them, one per pass) then runs beside the chain term below, which only reads LDS
        float Pc[3][9];
        if(jc >= 1)
        {
#pragma unroll
          for(int x = 0; x < 3; x++)
          {
            const float * Pv = mv.Pvm + ((int64_t)v * 3 + x) * NP + 9 * (jc - 1);
#pragma unroll
            for(int e = 0; e < 9; e++) Pc[x][e] = Pv[e];
          }
        }
        float acc[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}}; // [axis][row]
        // (two loops, not one with `wlds ? LDS : HBM` operands: a pointer that may be either compiles to flat loads, each
        // followed by a wait for EVERY outstanding load — the 27 requested above included)
        const float r0 = q0.x, r1 = q0.y, r2 = q0.z;
        auto chain_term = [&](float wm, int i) {
          const float4 * d = reinterpret_cast<const float4 *>(lds + L_DAB + (i * CS + cs0) * 12);
#pragma unroll
          for(int a = 0; a < 3; a++)
#pragma unroll
            for(int r = 0; r < 3; r++)
            {
              const float4 dr4 = d[a * 3 + r];
              acc[a][r] += wm * (((dr4.x * r0 + dr4.y * r1) + dr4.z * r2) + dr4.w);
            }
        };
        if(wlds)
        {
          const float wq[4] = {qw.x, qw.y, qw.z, qw.w}, jq[4] = {qj.x, qj.y, qj.z, qj.w}, aq[4] = {qa.x, qa.y, qa.z, qa.w};
#pragma unroll
          for(int m = 0; m < 4; m++)
          {
            // joint jc moves joint i only when it is its ancestor (or i itself): otherwise the term is exactly zero
            if(wq[m] != 0.0f && (__float_as_int(aq[m]) & jbit)) chain_term(wq[m], __float_as_int(jq[m]));
          }
        }
        else
          for(int m = 0; m < mv.maxw; m++)
          {
            const float wm = mv.wVal[(int64_t)v * mv.maxw + m];
            if(wm == 0.0f) continue;
            const int i = (int)mv.wIdx[(int64_t)v * mv.maxw + m];
            if(sAnc[i] & jbit) chain_term(wm, i);
          }
        if(jc >= 1) // pose correctives; the root joint has none (src/BlendShape.cpp:884-887)
        {
          float dr[3][3]; // [axis][coordinate x]
          float dRj[DRS];  // the joint's three derivative matrices, [9 a + e], in seven 16-byte reads
          {
            const float4 * d4 = reinterpret_cast<const float4 *>(lds + L_DR + jc * DRS);
#pragma unroll
            for(int u = 0; u < DRS / 4; u++)
            {
              const float4 w4 = d4[u];
              dRj[4 * u] = w4.x;
              dRj[4 * u + 1] = w4.y;
              dRj[4 * u + 2] = w4.z;
              dRj[4 * u + 3] = w4.w;
            }
          }
#pragma unroll
          for(int x = 0; x < 3; x++)
          {
#pragma unroll
            for(int a = 0; a < 3; a++)
            {
              float sacc = 0.f;
#pragma unroll
              for(int e = 0; e < 9; e++) sacc += Pc[x][e] * dRj[a * 9 + e];
              dr[a][x] = sacc;
            }
          }
#pragma unroll
          for(int a = 0; a < 3; a++)
#pragma unroll
            for(int r = 0; r < 3; r++) acc[a][r] += (Ab[r * 3] * dr[a][0] + Ab[r * 3 + 1] * dr[a][1]) + Ab[r * 3 + 2] * dr[a][2];
        }
#pragma unroll
        for(int a = 0; a < 3; a++)
#pragma unroll
          for(int r = 0; r < 3; r++) dpv[r * NQ + 3 + 3 * jc + a] = acc[a][r] * iws;
      }
      else
      {
        const int kb = g - 1 - NJ;
        float ds[3], acc[3];
        for(int x = 0; x < 3; x++) ds[x] = mv.Svm[((int64_t)v * 3 + x) * NB + kb];
        for(int r = 0; r < 3; r++) acc[r] = (Ab[r * 3] * ds[0] + Ab[r * 3 + 1] * ds[1]) + Ab[r * 3 + 2] * ds[2];
        for(int m = 0; m < mv.maxw; m++)
        {
          const float wm = mv.wVal[(int64_t)v * mv.maxw + m];
          if(wm == 0.0f) continue;
          const int i = mv.wIdx[(int64_t)v * mv.maxw + m];
          for(int r = 0; r < 3; r++) acc[r] += wm * lds[L_DBB + (i * 3 + r) * NB + kb];
        }
        for(int r = 0; r < 3; r++) dpv[r * NQ + TD75 + kb] = acc[r] * iws;
      }
    }
    // the column-independent half of B3n, once per adjacent face instead of once per (face, column): edges, unit normal and
    // |cross| of every face around the three vertices of each normal task (positions staged by B1)
    if((int)tid < ngn * 3 * MADJ)
    {
      const int gi = tid / (3 * MADJ), ia = tid % (3 * MADJ), i = ia / MADJ, a = ia % MADJ;
      int cnt = s_cnt[gi][i];
      if(cnt > MADJ) cnt = MADJ;
      if(a < cnt)
      {
        const float * rvb = lds + L_RV + s_roff[k_lo + gi] * RVS;
        // corners rotated (cyclically: same cross product) so that the first one is triangle vertex i itself — ring slot i —, which
        // every face around it contains: B3n then reads that vertex's derivative rows once per item, not once per face
        const uint8_t * mpb = s_map[gi] + ia * 3;
        int mp[3] = {mpb[0], mpb[1], mpb[2]};
        if(mp[1] == i)
        {
          mp[1] = mp[2];
          mp[2] = mp[0];
          mp[0] = i;
        }
        else if(mp[2] == i)
        {
          mp[2] = mp[1];
          mp[1] = mp[0];
          mp[0] = i;
        }
        const float * p0 = rvb + mp[0] * RVS + 13;
        const float * p1 = rvb + mp[1] * RVS + 13;
        const float * p2 = rvb + mp[2] * RVS + 13;
        const float e1[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
        const float e2[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
        float cr[3];
        cross3(e1, e2, cr);
        const float cn = fmaxf(sqrtf(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]), 1e-12f);
        float * ge = s_geo[gi][ia];
        ge[0] = cr[0] / cn;
        ge[1] = cr[1] / cn;
        ge[2] = cr[2] / cn;
        ge[3] = cn;
        ge[7] = 1.0f / cn;
        for(int x = 0; x < 3; x++)
        {
          ge[4 + x] = e1[x];
          ge[8 + x] = e2[x];
        }
        ge[11] = __int_as_float(mp[0] | (mp[1] << 8) | (mp[2] << 16)); // ring slots of the face's corners (B3n)
      }
    }
    __syncthreads();
    if(k_lo == k_begin) EVAL_STAMP(10);
    float * const svj = SVJ_EARLY ? lds + L_RV : &s_dvn[0][0]; // (nothing reads the ring-vertex records behind the barrier above; the next reader of svj is behind B3's)
    if(SVJ_EARLY && Jl_out)
    {
#pragma unroll
      for(int u = 0; u < VJ_PER; u++)
        if(tid + u * EVAL_NT < 63 * 32) svj[tid + u * EVAL_NT] = vjr[u];
    }
    // B3n (tasks with a normal term / offset, up to NGN to a group): the derivative of each of the three vertex normals,
    // one thread per (column, triangle vertex) — the chain n_f -> vn over ~6 adjacent faces is the long part of the
    // kernel for such tasks, and only nq of the 256 threads worked when a column's thread walked all three vertices
    if(ngn > 0) // d vertexNormal_i / dq  
    {
      for(int item = tid; item < ngn * nq * 3; item += EVAL_NT)
      {
        const int gi = (nq == TD75) ? item / (TD75 * 3) : item / (NQ * 3), qi = item - gi * (nq * 3);
        const int q = qi / 3, i = qi % 3;
        const int roff = s_roff[k_lo + gi];
        const float * dp = lds + L_DP + (roff * 3) * NQ; // this task's ring rows
        int cnt = s_cnt[gi][i];
        float sum = 0.f;
        for(int a = 0; a < cnt; a++) sum += 1.0f;
        const float aw = 1.0f / sum;
        if(cnt > MADJ) cnt = MADJ;
        float mu[3] = {0.f, 0.f, 0.f}, dmu[3] = {0.f, 0.f, 0.f};
        const float * dv = dp + (i * 3) * NQ + q; // triangle vertex i (ring slot i): the first corner of every face around it
        const float dv0 = dv[0], dv1 = dv[NQ], dv2 = dv[2 * NQ];
        for(int a = 0; a < cnt; a++)
        {
          // the adjacent face's geometry from s_geo (the same values every column used to recompute); its corners' ring slots
          // ride in the record's last word (three byte reads of the map per face and column otherwise)
          const float4 * ge = reinterpret_cast<const float4 *>(s_geo[gi][i * MADJ + a]);
          const float4 g0 = ge[0], g1 = ge[1], g2 = ge[2];
          const int mpw = __float_as_int(g2.w);
          const int mp[3] = {mpw & 255, (mpw >> 8) & 255, mpw >> 16};
          const float nh[3] = {g0.x, g0.y, g0.z}, icn = g1.w;
          const float e1[3] = {g1.x, g1.y, g1.z}, e2[3] = {g2.x, g2.y, g2.z};
          const float * d1 = dp + (mp[1] * 3) * NQ + q;
          const float * d2 = dp + (mp[2] * 3) * NQ + q;
          // (a face whose first corner is not slot i — a map that does not contain the vertex — cannot occur: the faces are
          // the ones adjacent to it)
          const float de1[3] = {d1[0] - dv0, d1[NQ] - dv1, d1[2 * NQ] - dv2};
          const float de2[3] = {d2[0] - dv0, d2[NQ] - dv1, d2[2 * NQ] - dv2};
          float t1[3], t2[3], dnf[3];
          cross3(de1, e2, t1);
          cross3(e1, de2, t2);
          const float dc[3] = {t1[0] + t2[0], t1[1] + t2[1], t1[2] + t2[2]};
          {
            // dnormalize_dev(cross, dc) with the unit normal and the norm taken from the table
            const float dd = nh[0] * dc[0] + nh[1] * dc[1] + nh[2] * dc[2];
            dnf[0] = (dc[0] - nh[0] * dd) * icn; // (a division per column and face before: one reciprocal per face now)
            dnf[1] = (dc[1] - nh[1] * dd) * icn;
            dnf[2] = (dc[2] - nh[2] * dd) * icn;
          }
          for(int x = 0; x < 3; x++)
          {
            mu[x] += aw * nh[x];
            dmu[x] += aw * dnf[x];
          }
        }
        float dvn[3];
        dnormalize_jac(mu, dmu, dvn);
        const float mn = fmaxf(sqrtf(mu[0] * mu[0] + mu[1] * mu[1] + mu[2] * mu[2]), 1e-12f);
        for(int x = 0; x < 3; x++)
        {
          const float vnx = mu[x] / mn;
          s_dvn[gi][(q * 3 + i) * 3 + x] = dvn[x];
          if(q == 0) lds[L_VN + gi * 12 + i * 3 + x] = vnx;
        }
      }
      __syncthreads();
    }
    if(k_lo == k_begin) EVAL_STAMP(11);
    for(int item = tid; item < (k_hi - k_lo) * nq; item += EVAL_NT) // B3: one (task, differentiation column) per thread
    {
      const int kq = (nq == TD75) ? item / TD75 : item / NQ;
      const int k = k_lo + kq, q = item - kq * nq;
      const float off = ta.noff[tb + k], wp = ta.posw[tb + k], wn = ta.nrmw[tb + k];
      const bool use_normal = (off > 0.0f) || (wn > 0.0f);
      const float w0 = ta.vw[(tb + k) * 3], w1 = ta.vw[(tb + k) * 3 + 1], w2 = ta.vw[(tb + k) * 3 + 2];
      double * Jk = J_out + ((f * K + k) * 4) * (int64_t)D;
      const float * dp = lds + L_DP + (s_roff[k] * 3) * NQ; // this task's ring rows
      float dn[3] = {0.f, 0.f, 0.f};
      if(use_normal) // d actualNormal / dq: the three vertex terms in order, as a single thread summed them
      {
        float msum[3] = {0.f, 0.f, 0.f}, dm[3] = {0.f, 0.f, 0.f};
        const float wv[3] = {w0, w1, w2};
        for(int i = 0; i < 3; i++)
          for(int x = 0; x < 3; x++)
          {
            msum[x] += wv[i] * lds[L_VN + (k - k_lo) * 12 + i * 3 + x];
            dm[x] += wv[i] * s_dvn[k - k_lo][(q * 3 + i) * 3 + x];
          }
        dnormalize_jac(msum, dm, dn);
        if(q == 0)
          for(int x = 0; x < 3; x++) lds[L_VN + (k - k_lo) * 12 + 9 + x] = msum[x];
      }
      const int jcol = (q < TD75) ? q : TD75 + 2 * K + (q - TD75);
      // VPoser latent layout: the columns that pass through ([pos 3 | root 3] <- 0..5, [aa22 | aa23] <- 69..74, beta) are written to
      // the latent rows as they are made (a copy pass behind the rows cost a second global round trip per 768 entries); the 63
      // body-joint columns are pulled back through the decoder's Jacobian behind the groups
      const int Dl = TD44 + 2 * K + (nq - TD75);
      const int lcol = !Jl_out ? -1 : (q < 6 ? q : (q < 69 ? -1 : (q < TD75 ? TD44 - 6 + (q - 69) : TD44 + 2 * K + (q - TD75))));
      double * Lk = Jl_out ? Jl_out + ((f * K + k) * 4) * (int64_t)Dl : nullptr;
      float nd = 0.f, rowv4[4];
      for(int x = 0; x < 3; x++)
      {
        float dpos = (w0 * dp[(0 * 3 + x) * NQ + q] + w1 * dp[(1 * 3 + x) * NQ + q]) + w2 * dp[(2 * 3 + x) * NQ + q];
        if(off > 0.0f) dpos += off * dn[x];
        rowv4[x] = wp * dpos;
        Jk[(int64_t)x * D + jcol] = (double)rowv4[x];
        if(lcol >= 0) Lk[(int64_t)x * Dl + lcol] = (double)rowv4[x];
        nd += dn[x] * ta.tnrm[(tb + k) * 3 + x];
      }
      rowv4[3] = (wn > 0.0f) ? wn * nd : 0.0f;
      Jk[(int64_t)3 * D + jcol] = (double)rowv4[3];
      if(lcol >= 0) Lk[(int64_t)3 * Dl + lcol] = (double)rowv4[3];
      // ... and the 63 body-joint columns stay in LDS for the pull-back: the four row entries (exact floats) IN PLACE of the first four
      // of the nine dp entries only this thread reads (column q of the task's own ring rows; every read of them is above)
      if(Jl_out && q >= 6 && q < 69)
      {
        float * stg = lds + L_DP + (s_roff[k] * 3) * NQ + q;
#pragma unroll
        for(int x = 0; x < 4; x++) stg[x * NQ] = rowv4[x];
      }
    }
    // phi columns of every task are zero except the task's own two (node.cpp:792, :834-839)
    for(int item = tid; item < (k_hi - k_lo) * 2 * K; item += EVAL_NT)
    {
      const int k = k_lo + item / (2 * K), c = item % (2 * K);
      const float plim = ta.philim[tb + k];
      double * Jk = J_out + ((f * K + k) * 4) * (int64_t)D;
      if(c / 2 != k || !(phi_live && plim > 0.0f))
      {
        for(int r = 0; r < 4; r++) Jk[(int64_t)r * D + TD75 + c] = 0.0;
        if(Jl_out)
        {
          const int Dl = TD44 + 2 * K + (nq - TD75);
          double * Lk = Jl_out + ((f * K + k) * 4) * (int64_t)Dl;
          for(int r = 0; r < 4; r++) Lk[(int64_t)r * Dl + TD44 + c] = 0.0;
        }
      }
    }
    lds_barrier(); // (global stores of this phase stay in flight: nothing reads them before the next full barrier)
    if(k_lo == k_begin) EVAL_STAMP(12);
    // ---- VPoser latent layout (node.cpp:761-772): the rows of this group over [pos 3 | root 3 | z 32 | aa22 3 | aa23 3 | phi | beta].
    // Columns 0..5 and 69..74 of J75 pass through (B3 / B4 write them beside the direct rows); columns 6..68 (joints 1..21) are
    // pulled back HERE through d(vposer out)/dz [63][32] of the frame, from the row entries B3 left in LDS: one (row, latent column)
    // per thread, the 63 terms in FOUR interleaved partial sums.  (Rounds 2-4: behind all groups, from the rows read back out of
    // global memory — a store -> load round trip through L2 and a second one for the decoder's Jacobian: 6.4 k cycles of a 50 k-cycle
    // evaluation at 8 chains; as a kernel of its own 19 us per iteration.)
    if(Jl_out)
    {
      if constexpr(!SVJ_EARLY)
      {
#pragma unroll
        for(int u = 0; u < VJ_PER; u++)
          if(tid + u * EVAL_NT < 63 * 32) svj[tid + u * EVAL_NT] = vjr[u];
        lds_barrier();
      }
      const int bdim = optimize_beta ? NB : 0, Dl = TD44 + 2 * K + bdim;
      // rows x latent columns in 16 x 16 tiles on the fp64 matrix pipe (v_mfma_f64_16x16x4_f64; lane l feeds A[l % 16][l / 16] and
      // B[l / 16][l % 16], receives D[4 r + l / 16][l % 16] in register r: tools/micro/mfma_f64_layout.hip), one tile per wavefront:
      // a lane converts 2 operands per 16 FMAs (one (row, column) per thread on the vector pipe converted 2 per FMA, and the
      // conversions, not the FMAs, were its 4 k cycles per group)
      typedef double d4 __attribute__((ext_vector_type(4)));
      const int nrw = 4 * (k_hi - k_lo), ntile = ((nrw + 15) >> 4) * 2;
      const int l = tid & 63, l16 = l & 15, lq = l >> 4;
      for(int t = tid >> 6; t < ntile; t += EVAL_NT / 64) // (wave-uniform)
      {
        const int rt = t >> 1, ct = t & 1, row = 16 * rt + l16;
        const bool rin = row < nrw;
        const int kr = k_lo + ((rin ? row : 0) >> 2);
        const float * jr = lds + L_DP + (s_roff[kr] * 3 + (row & 3)) * NQ + 6 + lq;
        const float * vj = svj + lq * 32 + 16 * ct + l16;
        float av[16], bv[16];
#pragma unroll
        for(int ks = 0; ks < 16; ks++)
        {
          const bool kin = 4 * ks + lq < 63;
          av[ks] = (rin && kin) ? jr[4 * ks] : 0.0f;
          bv[ks] = kin ? vj[4 * ks * 32] : 0.0f;
        }
        d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for(int ks = 0; ks < 16; ks++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)av[ks], (double)bv[ks], acc, 0, 0, 0);
#pragma unroll
        for(int r = 0; r < 4; r++)
        {
          const int orow = 16 * rt + 4 * r + lq;
          if(orow < nrw) Jl_out[((f * K + k_lo) * 4 + orow) * (int64_t)Dl + 6 + 16 * ct + l16] = acc[r];
        }
      }
    }
    if(tid < 2 * (k_hi - k_lo)) // B4: d/dphi through calcTriangleVertexWeights (vertices detached)
    {
      const int k = k_lo + tid / 2, c = tid % 2;
      const float plim = ta.philim[tb + k];
      if(phi_live && plim > 0.0f)
      {
        const int face = ta.face[tb + k];
        const float off = ta.noff[tb + k], wp = ta.posw[tb + k], wn = ta.nrmw[tb + k];
        const bool use_normal = (off > 0.0f) || (wn > 0.0f);
        double * Jk = J_out + ((f * K + k) * 4) * (int64_t)D;
        float tri[9];
        for(int i = 0; i < 3; i++)
          for(int x = 0; x < 3; x++) tri[i * 3 + x] = verts[3 * mv.faces[face * 3 + i] + x];
        // the point calcVertexWeights was evaluated at (node.cpp:804): pos + tangents . phi with phi == 0
        float pos[3] = {pos804[(tb + k) * 3], pos804[(tb + k) * 3 + 1], pos804[(tb + k) * 3 + 2]};
        float d[3][3], a[3], cr[3][3];
        for(int i = 0; i < 3; i++)
          for(int x = 0; x < 3; x++) d[i][x] = tri[i * 3 + x] - pos[x];
        for(int i = 0; i < 3; i++)
        {
          cross3(d[(i + 1) % 3], d[(i + 2) % 3], cr[i]);
          a[i] = sqrtf(cr[i][0] * cr[i][0] + cr[i][1] * cr[i][1] + cr[i][2] * cr[i][2]);
        }
        const float asum = (a[0] + a[1]) + a[2];
        const float nd[3] = {-ta.tang[(tb + k) * 6 + 0 * 2 + c], -ta.tang[(tb + k) * 6 + 1 * 2 + c], -ta.tang[(tb + k) * 6 + 2 * 2 + c]};
        float da[3], dasum = 0.f, dw[3];
        for(int i = 0; i < 3; i++)
        {
          float t1[3], t2[3];
          cross3(nd, d[(i + 2) % 3], t1);
          cross3(d[(i + 1) % 3], nd, t2);
          da[i] = (a[i] > 0.f) ? (cr[i][0] * (t1[0] + t2[0]) + cr[i][1] * (t1[1] + t2[1]) + cr[i][2] * (t1[2] + t2[2])) / a[i] : 0.f;
          dasum += da[i];
        }
        for(int i = 0; i < 3; i++) dw[i] = (da[i] - (a[i] / asum) * dasum) / asum;
        float dpos[3] = {0.f, 0.f, 0.f}, dnn[3] = {0.f, 0.f, 0.f};
        for(int i = 0; i < 3; i++)
          for(int x = 0; x < 3; x++) dpos[x] += dw[i] * tri[i * 3 + x];
        if(use_normal)
        {
          float dmm[3] = {0.f, 0.f, 0.f};
          for(int i = 0; i < 3; i++)
            for(int x = 0; x < 3; x++) dmm[x] += dw[i] * lds[L_VN + (k - k_lo) * 12 + i * 3 + x];
          dnormalize_jac(lds + L_VN + (k - k_lo) * 12 + 9, dmm, dnn);
        }
        float ndot = 0.f;
        const int Dl = TD44 + 2 * K + (nq - TD75);
        double * Lk = Jl_out ? Jl_out + ((f * K + k) * 4) * (int64_t)Dl : nullptr;
        for(int x = 0; x < 3; x++)
        {
          if(off > 0.0f) dpos[x] += off * dnn[x];
          Jk[(int64_t)x * D + TD75 + 2 * k + c] = (double)(wp * dpos[x]);
          if(Lk) Lk[(int64_t)x * Dl + TD44 + 2 * k + c] = (double)(wp * dpos[x]);
          ndot += dnn[x] * ta.tnrm[(tb + k) * 3 + x];
        }
        Jk[(int64_t)3 * D + TD75 + 2 * k + c] = (wn > 0.0f) ? (double)(wn * ndot) : 0.0;
        if(Lk) Lk[(int64_t)3 * Dl + TD44 + 2 * k + c] = (wn > 0.0f) ? (double)(wn * ndot) : 0.0;
      }
    }
    lds_barrier();
    if(k_lo == k_begin) EVAL_STAMP(13);
  }
  EVAL_STAMP(7);
}

template<int DMAX, int RCAP, int NGN, int MADJ = MAXADJ>
__global__ __launch_bounds__(EVAL_NT) void ik_eval_kernel(ModelView mv, TaskArrays ta, const float * __restrict__ theta25,
                                                      const float * __restrict__ verts_all, const float * __restrict__ rest_all,
                                                      const float * __restrict__ Gp, const float * __restrict__ joints,
                                                      const float * __restrict__ poserot, int K, int optimize_beta,
                                                      int phi_live, int min_valid, float * __restrict__ pos804,
                                                      double * __restrict__ e_out, double * __restrict__ J_out,
                                                      int * __restrict__ skip, int dbg_stop, int tsplit, const int32_t * __restrict__ roles,
                                                      const float * __restrict__ vjac, double * __restrict__ Jl_out)
{
  // (the side stream's fork is not raised here but by the solve kernel that follows, once its workgroups run: ik_solve_kernel)
  ik_eval_body<DMAX, RCAP, NGN, MADJ>(mv, ta, theta25, verts_all, rest_all, Gp, joints, poserot, K, optimize_beta, phi_live, min_valid, pos804, e_out,
                                J_out, skip, dbg_stop, tsplit, roles, vjac, Jl_out);
}

__global__ void ik_actual_normals_kernel(ModelView mv, TaskArrays ta, const float * __restrict__ verts_all, int K, int64_t nk)
{
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(t >= nk) return;
  const float * verts = verts_all + (t / K) * mv.V * 3;
  const float w[3] = {ta.vw[t * 3], ta.vw[t * 3 + 1], ta.vw[t * 3 + 2]};
  float an[3];
  actual_normal_dev(mv, verts, ta.face[t], w, an);
  for(int x = 0; x < 3; x++) ta.anrm[t * 3 + x] = an[x];
}

#ifdef SMPLPP_SOLVE_STAMPS
extern "C" int smplpp_debug_solve_stamps(unsigned long long * out)
{
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(smplpp_hip::g_solve_stamps), sizeof(unsigned long long) * 64 * 16);
}
#endif
#ifdef SMPLPP_EVAL_STAMPS
extern "C" int smplpp_debug_eval_stamps(unsigned long long * out)
{
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(smplpp_hip::g_eval_stamps), sizeof(unsigned long long) * 64 * 16);
}
#endif

// theta25 from the latent configuration (node.cpp:763-771)
__global__ void ik_splice_kernel(const float * __restrict__ g44, const float * __restrict__ vout /*[n,63]*/,
                                 float * __restrict__ theta25, int64_t n)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n * 75) return;
  const int64_t f = i / 75;
  const int c = (int)(i % 75);
  float v;
  if(c < 6)
    v = g44[f * TD44 + c];
  else if(c < 69)
  {
    if(!vout) return; // (pass-through entries only: the decoder writes its 63 angles into theta25 itself)
    v = vout[f * 63 + (c - 6)];
  }
  else
    v = g44[f * TD44 + 38 + (c - 69)];
  theta25[i] = v;
}
} // namespace smplpp_hip
