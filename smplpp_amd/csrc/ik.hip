// The IK loop of the reference (node/node.cpp:645-1002) for a batch of independent frames, one workgroup per frame.
//
//  ik_eval_kernel    node.cpp:798-877.  The reference gets each Jacobian row by a full reverse-mode autograd pass
//                    through the 6890-vertex FK graph (3-4 passes per task); here the same derivative is the analytic
//                    forward-mode Jacobian of only the vertices the tasks touch: per frame the chain
//                    derivatives dG'_i/dtheta_{j,k} of all 72 rotation columns are built once in LDS (41 KB: three columns per
//                    ancestor depth), one tree level per step, then every
//                    task reads them for its face vertices (and their 1-rings when a normal is involved: per-face ring
//                    tables built with the model).  In the VPoser layout it also writes the latent rows (node.cpp:761-772).
//  ik_solve_kernel   node.cpp:883-968: A = J^T J + damping in fp64 built straight from J staged through LDS, right-looking
//                    Cholesky of the packed lower-triangular augmented system in LDS (fp64) or the box QP by a primal
//                    active set around it, config update, query points for the re-projection.
//  proj_scan/finish  node.cpp:970-1001: exact closest point on the posed mesh (all 13776 faces, sphere-culled against the
//                    distance to each task's current face; each face gathered once per frame for all K queries), new
//                    face id and area-ratio weights.
// All fp32 where the reference is fp32 (FK, task geometry, autograd gradients), fp64 where it is fp64 (Eigen).
#include "mesh_device.h"
#include "staging.h"
#include "trace.h"
#include "signal.h"

#include <hip/hip_ext.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>

struct smplpp_vposer;

namespace smplpp_hip
{
int fk_device(smplpp_model * m, int64_t n, const float * beta, const float * theta, float * verts, float * joints,
              float * xforms44, float * rest, float * poserot, hipStream_t st, int range_slot, int * range_word = nullptr);
int vposer_forward_device(smplpp_vposer * v, int64_t n, const float * z, int64_t z_stride, float * out, int64_t out_stride,
                          float * jac, hipStream_t st, int64_t frame_base, bool value_like_jac = false, unsigned * sig_flag = nullptr,
                          unsigned * sig_counter = nullptr, unsigned sig_tick = 0u);

constexpr int TD75 = SMPLPP_THETA_DIM;        // 75
constexpr int TD44 = SMPLPP_LATENT_POSE_DIM;  // 44
constexpr int NQ = TD75 + NB;                 // differentiation columns handled per frame: theta(75) | beta(10)
constexpr int IK_MAXK = 48;                   // tasks per frame supported (the reference uses at most 41: MocapBody markers)
constexpr size_t SOLVE_LDS_MAX = 160 * 1024 - 1536; // dynamic LDS the solve kernels may ask for (160 KiB per CU, minus their static LDS: 1.2 KB)
constexpr int MAXD = TD75 + 2 * IK_MAXK + NB;  // 181: unknowns per frame supported by the in-LDS solver (every task count up to IK_MAXK, beta included)

struct TaskArrays
{
  int32_t * face;  // [n,K]
  float * vw;      // [n,K,3]
  float * tang;    // [n,K,3,2]
  float * tpos;    // [n,K,3]
  float * tnrm;    // [n,K,3]
  float * posw;    // [n,K]
  float * nrmw;    // [n,K]
  float * philim;  // [n,K]
  float * noff;    // [n,K]
  float * apos;    // [n,K,3]
  float * anrm;    // [n,K,3]
  float * hint;    // [n,K] squared distance of the actual position to the task's own face (cull radius of the re-projection)
  int * flags;     // [n] sticky per-frame status word (smplpp_ik_get_status): bit 0 a solve failed, bit 2 a task with a normal term
                   // touches a vertex with more than MAXADJ adjacent faces (its Jacobian rows are not supported; cleared by
                   // smplpp_ik_set_tasks, the solve skips the update of a frame that carries it)
  float * roww;    // [n,K,2] the (position, normal) task weights the LAST evaluation used: what decides which rows of J can be
                   // non-zero.  Written by ik_eval_kernel, read by ik_solve_kernel on the same stream — posw itself may already
                   // hold the NEXT frame's validity by then (the sequence driver's switch rides on the side stream's finish kernel)
};

struct ModelView
{
  const int32_t * faces;
  const int32_t * adjOff;
  const int32_t * adjFace;
  const int32_t * parent;
  const uint8_t * wIdx;
  const float * wVal;
  const float * wSum;
  const float * Pvm;
  const float * Svm;
  const float * JS;
  const uint16_t * faceRing; // [F][3 (madj + 1) + 2] per face: ring size, then the ring (common.h; madj = the model's table width: 12 or 16)
  const uint8_t * faceMap;   // [F][3 madj 3] (vertex of the face, adjacent face, corner) -> ring slot
  const int32_t * anc;       // [TREE_SIZE] tree tables (common.h): ancestor masks, joints by level
  int nlev;
  int64_t V;
  int maxw;
};

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

// workgroup barrier that orders LDS traffic only: global stores issued before it may still be in flight (__syncthreads
// waits for them too — one HBM round trip per barrier behind every phase that writes results out)
__device__ inline void lds_barrier()
{
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

#ifndef SMPLPP_EVAL_NT
#define SMPLPP_EVAL_NT 768
#endif
// threads per workgroup of ik_eval_kernel: one workgroup per frame owns a CU (158 KB of LDS), and its phases are bound by
// memory latency and per-item instruction count, so more wavefronts per SIMD both hide latency and shorten the item loops —
// but every instruction all threads execute alike (phase set-up, loop control) costs one issue slot per wavefront: 12
// wavefronts (170 registers each, nothing spilled) beat 16 by 5 % and 8 by 1 % on the 6-target solve
constexpr int EVAL_NT = SMPLPP_EVAL_NT;
#ifdef SMPLPP_EVAL_STAMPS
__device__ unsigned long long g_eval_stamps[64 * 16];
#define EVAL_STAMP(i) do { if(tid == 0 && blockIdx.x < 64) g_eval_stamps[blockIdx.x * 16 + (i)] = __builtin_readcyclecounter(); } while(0)
#else
#define EVAL_STAMP(i) do {} while(0)
#endif
#ifdef SMPLPP_SOLVE_STAMPS
__device__ unsigned long long g_solve_stamps[64 * 16];
#define SOLVE_STAMP(i) do { if(threadIdx.x == 0 && blockIdx.x < 64) g_solve_stamps[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while(0)
#else
#define SOLVE_STAMP(i) do {} while(0)
#endif
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

// ------------------------------------------------------------------------------------------------ solve kernel
__device__ inline int tri_idx(int i, int j)
{
  return i * (i + 1) / 2 + j; // i >= j
}
__device__ inline void tri_unpack(int item, int & i, int & j)
{
  i = (int)((sqrt(8.0 * (double)item + 1.0) - 1.0) * 0.5);
  while(tri_idx(i + 1, 0) <= item) i++;
  while(tri_idx(i, 0) > item) i--;
  j = item - tri_idx(i, 0);
}

// In-place right-looking Cholesky of the packed lower-triangular (nf+1)x(nf+1) augmented matrix [A b; b' *] held in
// LDS (fp64): the last row becomes y = L^-1 b, so forward substitution is free.  Every thread of the workgroup
// updates the trailing sub-matrix; two barriers per column.
__device__ inline void chol_aug(double * M, int nf, int * bad, double * dinv)
{
  const int tid = threadIdx.x, nt = blockDim.x;
  for(int j = 0; j < nf; j++)
  {
    double d = M[tri_idx(j, j)];
    if(!(d > 0.0))
    {
      if(tid == 0) *bad = 1;
      d = 1.0;
    }
    const double piv = sqrt(d);
    __syncthreads(); // everyone has read the pivot
    if(tid == 0) dinv[j] = 1.0 / piv;
    for(int i = j + tid; i <= nf; i += nt) M[tri_idx(i, j)] = (i == j) ? piv : M[tri_idx(i, j)] / piv;
    __syncthreads();
    // trailing update: rows i in (j, nf], columns k in (j, i]; the 256 threads tile the square as 16 x 16
    {
      const int ty = tid >> 4, tx = tid & 15;
      for(int i = j + 1 + ty; i <= nf; i += 16)
      {
        const double lij = M[tri_idx(i, j)];
        const int kend = (i < nf) ? i : nf - 1; // the (nf, nf) corner is never used
        for(int k = j + 1 + tx; k <= kend; k += 16) M[tri_idx(i, k)] -= lij * M[tri_idx(k, j)];
      }
    }
    __syncthreads();
  }
}

// back substitution L^T x = y (y = row nf of M), x returned in xs[0..nf); dinv[j] = 1 / L[j][j].
// Inside ONE wavefront: lane l keeps x[l], x[l + 64], x[l + 128] in registers, the pivot value travels by v_readlane and
// row j of L is a contiguous LDS read that does not depend on the recurrence — no workgroup barrier per column (the
// barrier-per-column form spent ~2 x nf barriers of four wavefronts on a strictly sequential chain).
__device__ inline double readlane_f64(double v, int lane)
{
  const uint64_t u = __builtin_bit_cast(uint64_t, v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, lane);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), lane);
  return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
__device__ inline void back_subst(const double * M, int nf, double * xs, const double * dinv)
{
  const int tid = threadIdx.x;
  if(nf > 192) // (not reached by any mode of the reference: D <= 75 + 2 * 41 + 10)
  {
    for(int i = tid; i < nf; i += blockDim.x) xs[i] = M[tri_idx(nf, i)];
    __syncthreads();
    for(int j = nf - 1; j >= 0; j--)
    {
      const double xj = xs[j] * dinv[j];
      __syncthreads();
      for(int k = tid; k < j; k += blockDim.x) xs[k] -= M[tri_idx(j, k)] * xj;
      if(tid == 0) xs[j] = xj;
      __syncthreads();
    }
    return;
  }
  __syncthreads(); // M and dinv are complete
  if(tid < 64)
  {
    // One wavefront, lane i keeps x_i (+64, +128).  The loop is a chain of nf steps whose cost is its instruction count (a
    // step used to be ~60 instructions, ~280 cycles): lane j's entry is never touched after step j (the row entries of lanes
    // >= j are read as zero), so nobody "owns" a finished entry inside the loop — entries stay unscaled and take their
    // 1/L_jj once, at the end; rows are read without exec masks (a lane beyond the row reads the zero word instead).
    __shared__ double s_zero;
    if(tid == 0) s_zero = 0.0;
    double x[3];
#pragma unroll
    for(int a = 0; a < 3; a++) x[a] = (tid + 64 * a < nf) ? M[tri_idx(nf, tid + 64 * a)] : 0.0;
    __builtin_amdgcn_wave_barrier();
    // three segments by the number of accumulators a row still reaches (rows 128.., 64..127, 0..63), each a loop without
    // branches whose next row and pivot are requested one step ahead (two steps: rows[2])
    auto segment = [&](auto na_tag, int jhi, int jlo) {
      constexpr int NA = decltype(na_tag)::value;
      if(jhi < jlo) return;
      auto fetch = [&](int j, double (&l)[NA], double & d) {
        const double * Lj = M + tri_idx(j, 0);
#pragma unroll
        for(int a = 0; a < NA - 1; a++) l[a] = Lj[tid + 64 * a];
        l[NA - 1] = *((tid + 64 * (NA - 1) < j) ? Lj + tid + 64 * (NA - 1) : &s_zero);
        d = dinv[j];
      };
      double l0[NA], l1[NA], d0, d1;
      fetch(jhi, l0, d0);
      fetch(jhi - 1 >= jlo ? jhi - 1 : jlo, l1, d1);
      for(int j = jhi; j >= jlo; j--)
      {
        double lc[NA];
#pragma unroll
        for(int a = 0; a < NA; a++) lc[a] = l0[a];
        const double dc = d0;
#pragma unroll
        for(int a = 0; a < NA; a++) l0[a] = l1[a];
        d0 = d1;
        fetch(j - 2 >= jlo ? j - 2 : jlo, l1, d1);
        const double xj = readlane_f64(x[NA - 1], j - 64 * (NA - 1)) * dc;
#pragma unroll
        for(int a = 0; a < NA; a++) x[a] = fma(-lc[a], xj, x[a]);
      }
    };
    segment(std::integral_constant<int, 3>{}, nf - 1, 128);
    segment(std::integral_constant<int, 2>{}, nf - 1 < 127 ? nf - 1 : 127, 64);
    segment(std::integral_constant<int, 1>{}, nf - 1 < 63 ? nf - 1 : 63, 0);
#pragma unroll
    for(int a = 0; a < 3; a++)
      if(tid + 64 * a < nf) xs[tid + 64 * a] = x[a] * dinv[tid + 64 * a];
  }
  __syncthreads();
}

// 1/sqrt(d) in fp64: hardware estimate (v_rsq_f64) + two Newton steps (relative error ~1e-16), an order of magnitude
// cheaper than sqrt() + a division on the pivot's critical path.
__device__ inline double fast_rsqrt(double d)
{
  double y = __builtin_amdgcn_rsq(d);
  y = y * (1.5 - 0.5 * d * y * y);
  y = y * (1.5 - 0.5 * d * y * y);
  return y;
}

// HBM -> LDS copy of cnt doubles by the 256 threads of the workgroup: eight loads in flight per thread (a one-load-per-
// iteration loop pays the full memory latency nine times for a 24 x 87 Jacobian)
__device__ inline void stage_rows(double * dst, const double * __restrict__ src, int cnt)
{
  const int tid = threadIdx.x;
  for(int q0 = 0; q0 < cnt; q0 += 256 * 8)
  {
    double t[8];
#pragma unroll
    for(int u = 0; u < 8; u++)
    {
      const int q = q0 + u * 256 + tid;
      t[u] = src[q < cnt ? q : cnt - 1];
    }
#pragma unroll
    for(int u = 0; u < 8; u++)
    {
      const int q = q0 + u * 256 + tid;
      if(q < cnt) dst[q] = t[u];
    }
  }
}

// the first W columns of cr rows (row stride D in HBM) packed at stride W in LDS, sixteen loads in flight per thread; rl
// (nullable): the rows to take, by index
__device__ inline void stage_rows_cols(double * dst, const double * __restrict__ src, int cr, int W, int D, const int * rl = nullptr)
{
  // (the copy is a chain of HBM round trips, ~1.5 us each with a single workgroup pulling: sixteen loads in flight per thread —
  // the 164 x 75 block of a motion solve in three round trips instead of six)
  const int tid = threadIdx.x, cnt = cr * W;
  constexpr int U = 16;
  for(int q0 = 0; q0 < cnt; q0 += 256 * U)
  {
    double t[U];
#pragma unroll
    for(int u = 0; u < U; u++)
    {
      const int q = q0 + u * 256 + tid, qq = q < cnt ? q : cnt - 1;
      const int rr = qq / W;
      t[u] = src[(int64_t)(rl ? rl[rr] : rr) * D + (qq - rr * W)];
    }
#pragma unroll
    for(int u = 0; u < U; u++)
    {
      const int q = q0 + u * 256 + tid;
      if(q < cnt) dst[q] = t[u];
    }
  }
}

// The same copy by LDS-DMA (buffer_load_dwordx4 ... lds: memory -> LDS without a register in between, 16 bytes per lane, 64
// consecutive 16-byte LDS slots per instruction from per-lane addresses), all of a wavefront's pieces in flight at once: the 123 live
// rows x 75 columns of a motion solve (74 KB) are 74 instructions for the whole workgroup and arrive in about one memory round trip,
// where stage_rows_cols took three (of sixteen 8-byte loads per thread each, ~1.5-2 us apiece with a single workgroup pulling).
// LDS rows have the EVEN stride Wp = W + (W & 1) doubles, so that every lane's 16 bytes lie inside one row; a row's last lane may
// carry one double of column W (or, on the last column, of the next row): it lands in the pad slot nobody reads.  The source rows are
// only 8-byte aligned (odd D): dword-aligned buffer loads.  rl (LDS): the rows to take.  dst must have room for the count rounded
// up to 64 slots (the caller checks).
__device__ inline void stage_rows_cols_dma(double * dst, const double * __restrict__ src, int cr, int W, int D, const int * rl, int rows_total)
{
  typedef __attribute__((address_space(3))) void * lds_ptr_t;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int SPR = (W + 1) >> 1, cnt = cr * SPR; // 16-byte slots per row, in all
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(src), 0, rows_total * D * 8, 0x00020000);
  // every row index is read from LDS BEFORE the first DMA is issued (the compiler cannot tell the DMA's LDS destination from the
  // other arrays of the dynamic LDS block: an LDS read behind a DMA waits for vmcnt(0))
  constexpr int U = 24; // 24 x 256 slots of 16 bytes = 96 KiB per round
  const unsigned magic = (unsigned)((0x100000000ull + (unsigned)SPR - 1) / (unsigned)SPR); // floor(d / SPR) = umulhi(d, magic) for d < 2^25 / SPR >= 2^18
  for(int base = wave * 64; base < cnt; base += 256 * U)
  {
    int voff[U];
#pragma unroll
    for(int u = 0; u < U; u++)
    {
      const int dd = base + lane + 256 * u;
      const int row = (int)__umulhi((unsigned)dd, magic), within = dd - row * SPR;
      // (lanes past the end ask beyond the descriptor's range: nothing is fetched, zeros land in the slack behind the block)
      voff[u] = (dd < cnt) ? (rl[row] * D + 2 * within) * 8 : 0x7ffffff0;
    }
    __builtin_amdgcn_sched_barrier(0);
    SOLVE_STAMP(12);
#pragma unroll
    for(int u = 0; u < U; u++)
      if(base + 256 * u < cnt) // (wave-uniform)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(reinterpret_cast<unsigned char *>(dst) + (size_t)(base + 256 * u) * 16), 16, voff[u], 0, 0, 0);
    SOLVE_STAMP(13);
  }
  // (Issuing these from the kernel's set-up, on the guess that theta alone is free, was tried: the compiler cannot tell the DMA's LDS
  // destination from the other arrays of the same dynamic LDS block and waits for vmcnt(0) in front of the NEXT LDS access, so
  // nothing overlapped — stop-timed, round 4.  Measured alone (tools/micro/stage_probe.hip): 1.9 us for the 74 KB block, ~16 B/clk,
  // the same cold or warm and for 8- or 16-byte-aligned rows; 4-byte DMA 6.8 us; sixteen 8-byte register loads per thread 6.8 us.)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// rows of the staged block a chunk may hold when it goes through LDS-DMA at (even) row stride Weven; < 4: no DMA
__device__ inline int dma_chunk_rows(int chunk_rows, int D, int Weven)
{
  return (int)(((int64_t)chunk_rows * D) / Weven) - (128 + Weven - 1) / Weven; // (1 KiB of slack: the last instruction's tail)
}

// Factorisation + both substitutions of the packed (r + 1) x (r + 1) augmented matrix [S v; v' *] by ONE wavefront, lane i
// owning row i. Register form (r <= RMAX <= 32): the row lives in registers, a column's entries reach the other lanes by
// v_readlane (an SGPR operand of the FMA), so a column costs its pivot's rsqrt plus (r - k) FMAs and no LDS round trip;
// the factor is written back packed and re-read by columns (independent loads, hoisted) for the back substitution, whose
// chain is then readlane + FMA only. w[0..r) = S^-1 v.
// EXACT: r == RMAX is known where the call is made (the 6-target solve: 24), so the column loop carries no `k < r` branch and the whole
// factorisation is ONE basic block: the scheduler then starts column k + 1's pivot chain (two v_readlane, rsqrt estimate, two Newton
// steps: ~100 cycles of dependent latency) as soon as row k + 1 has taken column k's update, beside the remaining updates of column k.
template<int RMAX, bool EXACT = false>
__device__ inline void chol_wave_reg(double * M, int r, double * w, int * bad)
{
  const int i = threadIdx.x; // < 64
  const bool act = i <= r;
  double row[RMAX];
#pragma unroll
  for(int j = 0; j < RMAX; j++) row[j] = (act && j <= i && j < r) ? M[tri_idx(i, j)] : 0.0;
  double myrinv = 0.0;
  bool badl = false;
#pragma unroll
  for(int k = 0; k < RMAX; k++)
  {
    if(EXACT || k < r) // uniform
    {
      double piv = readlane_f64(row[k], k);
      if(!(piv > 0.0))
      {
        badl = true;
        piv = 1.0;
      }
      const double ri = fast_rsqrt(piv);
      const double l = row[k] * ri; // lane k: sqrt(piv); lanes below the diagonal: L[i][k]; the rhs lane r: y[k]
      row[k] = l;
      if(i == k) myrinv = ri;
#pragma unroll
      for(int j = k + 1; j < RMAX; j++) row[j] = fma(-l, readlane_f64(l, j), row[j]);
    }
  }
#pragma unroll
  for(int j = 0; j < RMAX; j++)
    if(act && j <= i && j < r) M[tri_idx(i, j)] = row[j];
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  double col[RMAX + 1]; // col[k] = L[k][i] for k > i (k == r: y[i])
  col[0] = 0.0;
#pragma unroll
  for(int k = 1; k <= RMAX; k++) col[k] = (i < k && k <= r && i < r) ? M[tri_idx(k, i)] : 0.0;
  double acc = 0.0;
#pragma unroll
  for(int k = RMAX; k >= 1; k--)
    if(k == r) acc = col[k];
#pragma unroll
  for(int k = RMAX - 1; k >= 0; k--)
  {
    if(EXACT || k < r) // uniform
    {
      const double wk = readlane_f64(acc, k) * readlane_f64(myrinv, k);
      acc = (i == k) ? wk : fma(-col[k], wk, acc); // col[k] is 0 for lanes i >= k
    }
  }
  if(i < r) w[i] = acc;
  if(badl && i == 0) *bad = 1;
}

// LDS form for 32 < r <= 63 (left-looking on the packed matrix)
__device__ inline void chol_wave_lds(double * M, int r, double * w, int * bad)
{
  const int tid = threadIdx.x;
  const int i = tid;
  const bool act = i <= r;
  const double * Li = M + tri_idx(act ? i : 0, 0);
  double myrinv = 0.0;
  bool badl = false;
  for(int k0 = 0; k0 < r; k0++)
  {
    const int k = __builtin_amdgcn_readfirstlane(k0);
    const double * Lk = M + tri_idx(k, 0);
    double s = 0.0;
    if(act && i >= k)
    {
      double s1 = 0.0;
      s = Li[k];
      int m = 0;
      for(; m + 1 < k; m += 2)
      {
        s -= Li[m] * Lk[m];
        s1 -= Li[m + 1] * Lk[m + 1];
      }
      if(m < k) s -= Li[m] * Lk[m];
      s += s1;
    }
    double piv = readlane_f64(s, k);
    if(!(piv > 0.0))
    {
      badl = true;
      piv = 1.0;
    }
    const double ri = fast_rsqrt(piv);
    if(act && i >= k) M[tri_idx(i, k)] = (i == k) ? piv * ri : s * ri;
    if(i == k) myrinv = ri;
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
  }
  double yv = (i < r) ? M[tri_idx(r, i)] : 0.0; // y = L^-1 v (the augmented row)
  for(int k0 = r - 1; k0 >= 0; k0--)
  {
    const int k = __builtin_amdgcn_readfirstlane(k0);
    const double lk = (i < k) ? M[tri_idx(k, 0) + i] : 0.0;
    const double wk = readlane_f64(yv, k) * readlane_f64(myrinv, k);
    yv = (i == k) ? wk : yv - lk * wk;
  }
  if(i < r) w[i] = yv;
  if(badl && tid == 0) *bad = 1;
}

// Dual form of the damped free-set system for FEWER RESIDUAL ROWS THAN FREE UNKNOWNS (r = 4K < nf; the 6-target solve has
// r = 24 against 75): with G = the diagonal damping (> 0, node.cpp:887-904) and J_F the free columns,
//   (G + J_F' J_F)^-1 c = G^-1 c - G^-1 J_F' (I + J_F G^-1 J_F')^-1 J_F G^-1 c
// so the Cholesky factorisation is r x r instead of nf x nf — the same x = -LLT(A)^-1 b of node.cpp:933-938 to fp64
// round-off (S = I + Jf Jf' with Jf = J_F G^-1/2 is at least as well conditioned as A). Steps: gather the free columns
// into LDS (eight loads in flight), c / u = G^-1 c and the column scaling (one thread per column), S and v = J u (one
// element per thread), factorisation + both substitutions inside ONE wavefront (left-looking on the packed LDS matrix with
// v as the augmented last row; pivots travel by v_readlane, no workgroup barrier per column), x = u - G^-1/2 Jf' w.
// Returns A^-1 c in xs[0..nf) like back_subst(). Needs r <= 63, r * nf doubles in Jf, nf in us/ginv, r in w.
// pre (nullable): the first 2048 gathered entries, loaded by the caller at kernel start on the GUESS that the free set is
// columns 0 .. pre_nf - 1 (true whenever only theta is free); used when the guess holds.
// (PRE is a template parameter and `pre` a reference to the caller's registers: as a nullable pointer the eight doubles lived
// in scratch memory and came back through flat loads)
template<bool PRE>
__device__ __forceinline__ void solve_dual(double * M, const double * __restrict__ J, const double * rowv, double * Jf, const double * diag,
                                           const double * bpri, const int * idx, int nf, int D, int r, double * ginv, double * us, double * w,
                                           double * xs, int * bad, int dbg_stop, const double (&pre)[8], int pre_nf)
{
  const int tid = threadIdx.x;
  const int cnt = r * nf;
  const bool use_pre = PRE && pre_nf == nf && idx[nf - 1] == nf - 1; // (ascending, distinct: then idx is the identity; uniform)
  for(int q0 = 0; q0 < cnt; q0 += 256 * 8)
  {
    double t[8];
    if(use_pre && q0 == 0)
    {
#pragma unroll
      for(int u = 0; u < 8; u++) t[u] = pre[u];
    }
    else
#pragma unroll
    for(int u = 0; u < 8; u++)
    {
      int q = q0 + u * 256 + tid;
      q = q < cnt ? q : cnt - 1;
      const int i = q / nf, a = q - i * nf;
      t[u] = J[(int64_t)i * D + idx[a]];
    }
#pragma unroll
    for(int u = 0; u < 8; u++)
    {
      const int q = q0 + u * 256 + tid;
      if(q < cnt) Jf[q] = t[u];
    }
  }
  __syncthreads();
  SOLVE_STAMP(2);
  if(dbg_stop == 31) return; // (timing experiments only)
  if(tid < nf)
  {
    const int a = tid, q = idx[a];
    double c = bpri[q];
    for(int i = 0; i < r; i++) c += Jf[i * nf + a] * rowv[i];
    const double gi = 1.0 / diag[q];
    const double sg = sqrt(gi);
    ginv[a] = gi;
    xs[a] = c * gi;  // u
    us[a] = c * sg;  // u / sg: v = J u = Jf (u / sg)
    for(int i = 0; i < r; i++) Jf[i * nf + a] *= sg;
  }
  __syncthreads();
  SOLVE_STAMP(3);
  if(dbg_stop == 32) return;
  {
    // S = I + Jf Jf' (r x r) and the augmented row v' = us' Jf' on the fp64 matrix pipe (round 4): one 16 x 16 tile of the lower
    // triangle of rows 0..r per wavefront and turn, the nf free columns as the k dimension, four per v_mfma_f64_16x16x4_f64 (operand
    // and result layout: build_and_factor_reg).  The 325 dot products of 75 terms, one or two per thread with two LDS reads per term,
    // took 5.1 us of the 6-target solve's 25.
    typedef double d4 __attribute__((ext_vector_type(4)));
    const int wave = tid >> 6, l = tid & 63, l16 = l & 15, lq = l >> 4;
    const int ntr = (r + 16) >> 4; // tile rows covering rows 0..r
    const int ntile = ntr * (ntr + 1) / 2;
    for(int t = wave; t < ntile; t += 4) // (wave-uniform)
    {
      int ta = 0, tb = t;
      while(tb > ta)
      {
        tb -= ta + 1;
        ta++;
      }
      const int ia = 16 * ta + l16, ib = 16 * tb + l16;
      const double * pa = (ia < r) ? Jf + ia * nf : us; // (row r: the rhs; rows beyond: masked below)
      const double * pb = Jf + (ib < r ? ib : 0) * nf;
      const bool la = ia <= r, lb = ib < r;
      d4 acc = {0.0, 0.0, 0.0, 0.0};
      for(int k0 = 0; k0 < nf; k0 += 16)
      {
        double a[4], b[4];
#pragma unroll
        for(int u = 0; u < 4; u++)
        {
          const int k = k0 + 4 * u + lq, kk = k < nf ? k : 0;
          a[u] = pa[kk];
          b[u] = pb[kk];
          if(!(la && k < nf)) a[u] = 0.0;
          if(!(lb && k < nf)) b[u] = 0.0;
        }
#pragma unroll
        for(int u = 0; u < 4; u++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
      }
#pragma unroll
      for(int rr = 0; rr < 4; rr++)
      {
        const int i = 16 * ta + 4 * rr + lq, j = 16 * tb + l16;
        if(i >= j && i <= r && j < r) M[tri_idx(i, j)] = acc[rr] + (i == j ? 1.0 : 0.0);
      }
    }
  }
  __syncthreads();
  SOLVE_STAMP(4);
  if(dbg_stop == 33) return;
  if(tid < 64)
  {
    switch((r + 7) >> 3)
    {
      case 1: chol_wave_reg<8>(M, r, w, bad); break;
      case 2: chol_wave_reg<16>(M, r, w, bad); break;
      case 3:
        if(r == 24)
          chol_wave_reg<24, true>(M, r, w, bad);
        else
          chol_wave_reg<24>(M, r, w, bad);
        break;
      case 4: chol_wave_reg<32>(M, r, w, bad); break;
      default: chol_wave_lds(M, r, w, bad); break;
    }
  }
  __syncthreads();
  SOLVE_STAMP(5);
  if(dbg_stop == 34) return;
  if(tid < nf)
  {
    const int a = tid;
    double t = 0.0;
    for(int i = 0; i < r; i++) t += Jf[i * nf + a] * w[i];
    xs[a] = xs[a] - sqrt(ginv[a]) * t;
  }
  __syncthreads();
}

// Register-tiled build + factorisation of the augmented free-set system for nf + 1 <= 16 * NT: thread (ty, tx) of the
// 16 x 16 workgroup owns the elements (ty + 16a, tx + 16b), b <= a, in registers.  Per column ONE barrier: the column's
// holders publish its raw entries (and the pivot entry) to LDS, every thread then applies the rank-1 update to its own
// registers as acc -= raw_i * raw_k / d.  The scaled column is also written to the packed LDS matrix M for back_subst.
template<int NT>
__device__ inline void build_and_factor_reg(double * M, const double * __restrict__ J, const double * __restrict__ rowv, double * Jc,
                                            const double * diag, const double * bpri, const int * idx, int nf, int D, int rows,
                                            int chunk_rows, double * lraw /*[2][4][16*NT]*/, double * ldiag /*[4]*/, double * dinv /*[nf]*/, int * bad,
                                            int dbg_stop, const int * rlist /*[nlive] rows of J that are not identically zero*/, int nlive)
{
  // thread (ty, tx): tx in the HIGH bits, so the 16 holders of a column (one tx, all ty) sit in one wavefront and the other
  // three skip the publish path (extraction, rsqrt, LDS writes) instead of executing it for four lanes each
  const int tid = threadIdx.x, tx = tid >> 4, ty = tid & 15;
  double acc[NT][NT];
  // A_FF = J_F^T J_F and the rhs row J_F^T rowv (the Gram of the augmented operand [J_F | rowv]) on the fp64 matrix pipe:
  // v_mfma_f64_16x16x4_f64, one 16 x 16 tile of the lower triangle per accumulator, the wavefronts take tiles round-robin,
  // the rows of J (staged in LDS in chunks) are the k dimension, four per MFMA.  Lane l feeds A[i = l % 16][k = l / 16] and
  // B[k = l / 16][j = l % 16] and receives D[4 r + l / 16][l % 16] in register r (probed: tools/micro/mfma_f64_layout.hip).
  // The tiles go through the packed LDS matrix M into the register layout of the factorisation below.
  typedef double d4 __attribute__((ext_vector_type(4)));
  const int nitemM = (nf + 1) * (nf + 2) / 2;
  // Two forms of the Gram loop.  ROWS SPLIT OVER THE WAVEFRONTS (round 4; tile counts up to 6, partial sums in the row chunk's LDS
  // once the chunk is dead): wavefront w takes the row groups w, w + 4, ... and accumulates EVERY live tile from them — per group of
  // four rows NT operand reads feed NT (NT + 1) / 2 MFMAs (5 reads for 15), where the tile-per-wavefront form below pays two reads
  // per MFMA and is a chain of read -> wait -> 4 MFMAs per group: 31 groups x ~600 cycles for a capture solve against 8 x ~1100.
  // The four partial sums are added as (w0 + w2) + (w1 + w3) on the way into the factorisation's register layout.
  const bool ksplit = NT <= 6 && 2 * (NT * (NT + 1) / 2) * 256 + nitemM <= chunk_rows * D; // (uniform: two raw partials + one packed triangle fit the row chunk)
  if(ksplit)
  {
    constexpr int NTILE = NT <= 6 ? NT * (NT + 1) / 2 : 1, NTK = NT <= 6 ? NT : 1;
    const int wave = tid >> 6, l = tid & 63, l16 = l & 15, lq = l >> 4;
    d4 tacc[NTILE];
    int colT[NTK]; // column of J (>= 0), -1 the rhs entry, -2 nothing, of this lane's element of tile row / tile column t
#pragma unroll
    for(int t = 0; t < NTK; t++)
    {
      const int m = 16 * t + l16;
      colT[t] = (m < nf) ? idx[m] : (m == nf ? -1 : -2);
    }
#pragma unroll
    for(int u = 0; u < NTILE; u++) tacc[u] = d4{0.0, 0.0, 0.0, 0.0};
    const int W = nf > 0 ? idx[nf - 1] + 1 : 1;
    const bool whole = W == D && nlive == rows;
    const int Weven = W + (W & 1);
    const int crows_dma = dma_chunk_rows(chunk_rows, D, Weven);
    const bool dma = !whole && W < D && crows_dma >= 4 && (int64_t)rows * D * 8 < 0x7fffff00LL;
    const int Wp = dma ? Weven : W;
    const int crows = dma ? crows_dma : (int)(((int64_t)chunk_rows * D) / W);
    if(dbg_stop == 40) return; // (timing experiments only)
    SOLVE_STAMP(2);
    for(int c0 = 0; c0 < nlive; c0 += crows)
    {
      const int cr = (nlive - c0 < crows) ? nlive - c0 : crows;
      __syncthreads();
      SOLVE_STAMP(3);
      if(whole)
        stage_rows(Jc, J + (int64_t)c0 * D, cr * D);
      else if(dma)
        stage_rows_cols_dma(Jc, J, cr, W, D, rlist + c0, rows);
      else
        stage_rows_cols(Jc, J, cr, W, D, rlist + c0);
      SOLVE_STAMP(4);
      __syncthreads();
      SOLVE_STAMP(5);
      if(dbg_stop == 41) return; // (timing experiments only)
      for(int r0 = 4 * wave; r0 < cr; r0 += 16)
      {
        const int r = r0 + lq;
        const bool rin = r < cr;
        const double rv = rowv[rlist[c0 + (rin ? r : 0)]];
        const double * Jr = Jc + (rin ? r : 0) * Wp;
        double v[NTK], va[NTK], vb[NTK];
#pragma unroll
        for(int t = 0; t < NTK; t++) v[t] = Jr[colT[t] >= 0 ? colT[t] : 0];
#pragma unroll
        for(int t = 0; t < NTK; t++)
        {
          va[t] = !rin ? 0.0 : (colT[t] >= 0 ? v[t] : (colT[t] == -1 ? rv : 0.0));
          vb[t] = (rin && colT[t] >= 0) ? v[t] : 0.0;
        }
#pragma unroll
        for(int ta = 0; ta < NTK; ta++)
#pragma unroll
          for(int tb = 0; tb <= ta; tb++)
          {
            if(!(16 * ta <= nf && 16 * tb < nf)) continue; // (uniform)
            d4 & t = tacc[ta * (ta + 1) / 2 + tb];
            t = __builtin_amdgcn_mfma_f64_16x16x4f64(va[ta], vb[tb], t, 0, 0, 0);
          }
      }
    }
    __syncthreads(); // every wavefront is done with the row chunk: its LDS takes the partial sums of wavefronts 1..3
    SOLVE_STAMP(6);
    if(dbg_stop == 42) return;
    // the four partial sums meet in two stages: wavefronts 2 and 3 drop theirs as they lie (lane-linear, [tile][register][lane]: no index
    // arithmetic, no bank conflicts), wavefronts 0 and 1 add them in registers — (w0 + w2), (w1 + w3) — and write the packed
    // triangles the factorisation's layout is gathered from, two reads per element instead of four
    constexpr int PRAW = NTILE * 4 * 64; // doubles of one raw partial
    double * const praw = Jc + (size_t)((wave & 1) * PRAW);
    double * const ptri = Jc + 2 * PRAW; // wavefront 1's packed triangle (wavefront 0's: M)
    if(wave >= 2)
    {
#pragma unroll
      for(int u = 0; u < NTILE; u++)
#pragma unroll
        for(int rr = 0; rr < 4; rr++) praw[(u * 4 + rr) * 64 + l] = tacc[u][rr];
    }
    __syncthreads();
    if(wave < 2)
    {
#pragma unroll
      for(int u = 0; u < NTILE; u++)
#pragma unroll
        for(int rr = 0; rr < 4; rr++) tacc[u][rr] += praw[(u * 4 + rr) * 64 + l];
      double * P = wave == 0 ? M : ptri;
#pragma unroll
      for(int ta = 0; ta < NTK; ta++)
#pragma unroll
        for(int tb = 0; tb <= ta; tb++)
        {
          if(!(16 * ta <= nf && 16 * tb < nf)) continue;
#pragma unroll
          for(int rr = 0; rr < 4; rr++)
          {
            const int i = 16 * ta + 4 * rr + lq, k = 16 * tb + l16;
            if(i >= k && i <= nf && k < nf) P[tri_idx(i, k)] = tacc[ta * (ta + 1) / 2 + tb][rr];
          }
        }
    }
    __syncthreads();
#pragma unroll
    for(int a2 = 0; a2 < NT; a2++)
#pragma unroll
      for(int b2 = 0; b2 <= a2; b2++)
      {
        const int i = ty + 16 * a2, k = tx + 16 * b2;
        double sum = 0.0;
        if(i >= k && i <= nf && k < nf)
        {
          const int q = tri_idx(i, k);
          sum = M[q] + ptri[q];
        }
        acc[a2][b2] = sum;
      }
    __syncthreads(); // M is rewritten by the factorisation
  }
  else
  {
    constexpr int NTILE = NT * (NT + 1) / 2, TPW = (NTILE + 3) / 4;
    const int wave = tid >> 6, l = tid & 63, l16 = l & 15, lq = l >> 4;
    d4 tacc[TPW];
    int colA[TPW], colB[TPW]; // column of J (>= 0), -1 the rhs entry, -2 nothing, of this lane's A / B operand element
    bool live[TPW];
#pragma unroll
    for(int u = 0; u < TPW; u++)
    {
      const int t = wave + 4 * u;
      int ta = 0, tb = t; // tile t of the row-major lower triangle: (ta, tb), tb <= ta
      while(tb > ta)
      {
        tb -= ta + 1;
        ta++;
      }
      live[u] = t < NTILE && 16 * ta <= nf && 16 * tb < nf; // (wave-uniform)
      const int mi = 16 * ta + l16, mk = 16 * tb + l16;
      colA[u] = (mi < nf) ? idx[mi] : (mi == nf ? -1 : -2);
      colB[u] = (mk < nf) ? idx[mk] : -2;
      tacc[u] = d4{0.0, 0.0, 0.0, 0.0};
    }
    // only the columns up to the last free one are staged (the free set is ascending), rows packed at that width: a motion
    // solve with its surface coordinates pinned reads 75 of its 157 columns — half the traffic, and all 164 rows in ONE chunk
    const int W = nf > 0 ? idx[nf - 1] + 1 : 1;
    // ... and only the rows that can be non-zero (rlist): a task without a normal term has a zero fourth row, a missing marker
    // four zero rows — a quarter of the 164 rows of a capture solve.  Zero rows add exact zeros: the sums keep their bits.
    // LDS row stride: W, or the next even number when the block goes through LDS-DMA (16-byte slots: stage_rows_cols_dma; W < D,
    // so that a row's pad slot is filled from inside the same source row)
    const bool whole = W == D && nlive == rows;
    const int Weven = W + (W & 1);
    const int crows_dma = dma_chunk_rows(chunk_rows, D, Weven);
    const bool dma = !whole && W < D && crows_dma >= 4 && (int64_t)rows * D * 8 < 0x7fffff00LL;
    const int Wp = dma ? Weven : W;
    const int crows = dma ? crows_dma : (int)(((int64_t)chunk_rows * D) / W);
    if(dbg_stop == 40) return; // (timing experiments only)
    SOLVE_STAMP(2);
    for(int c0 = 0; c0 < nlive; c0 += crows)
    {
      const int cr = (nlive - c0 < crows) ? nlive - c0 : crows;
      __syncthreads();
      SOLVE_STAMP(3);
      if(whole)
        stage_rows(Jc, J + (int64_t)c0 * D, cr * D);
      else if(dma)
        stage_rows_cols_dma(Jc, J, cr, W, D, rlist + c0, rows);
      else
        stage_rows_cols(Jc, J, cr, W, D, rlist + c0);
      SOLVE_STAMP(4);
      __syncthreads();
      SOLVE_STAMP(5);
      if(dbg_stop == 41) return; // (timing experiments only)
      for(int r0 = 0; r0 < cr; r0 += 4)
      {
        const int r = r0 + lq;
        const bool rin = r < cr;
        const double rv = rowv[rlist[c0 + (rin ? r : 0)]];
        const double * Jr = Jc + (rin ? r : 0) * Wp;
        // (every tile's two operands are read first, then the MFMAs: a read -> wait -> MFMA pair per tile paid the LDS round
        // trip TPW times per four rows)
        double ja[TPW], jb[TPW];
#pragma unroll
        for(int u = 0; u < TPW; u++)
        {
          ja[u] = Jr[colA[u] >= 0 ? colA[u] : 0];
          jb[u] = Jr[colB[u] >= 0 ? colB[u] : 0];
        }
#pragma unroll
        for(int u = 0; u < TPW; u++)
        {
          if(!live[u]) continue; // (wave-uniform)
          const double va = !rin ? 0.0 : (colA[u] >= 0 ? ja[u] : (colA[u] == -1 ? rv : 0.0));
          const double vb = (rin && colB[u] >= 0) ? jb[u] : 0.0;
          tacc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(va, vb, tacc[u], 0, 0, 0);
        }
      }
    }
    __syncthreads();
    SOLVE_STAMP(6);
    if(dbg_stop == 42) return;
#pragma unroll
    for(int u = 0; u < TPW; u++)
    {
      if(!live[u]) continue;
      const int t = wave + 4 * u;
      int ta = 0, q = t;
      while(q > ta)
      {
        q -= ta + 1;
        ta++;
      }
      const int tb = q;
#pragma unroll
      for(int rr = 0; rr < 4; rr++)
      {
        const int i = 16 * ta + 4 * rr + lq, k = 16 * tb + l16;
        if(i >= k && i <= nf && k < nf) M[tri_idx(i, k)] = tacc[u][rr];
      }
    }
    __syncthreads();
#pragma unroll
    for(int a2 = 0; a2 < NT; a2++)
#pragma unroll
      for(int b2 = 0; b2 <= a2; b2++)
      {
        const int i = ty + 16 * a2, k = tx + 16 * b2;
        acc[a2][b2] = (i >= k && i <= nf && k < nf) ? M[tri_idx(i, k)] : 0.0;
      }
    __syncthreads(); // M is rewritten by the factorisation
  }
  // damping on the diagonal, the prior's rhs on the augmented row: branch-free with every index read first, then every value (one
  // conditional block per tile, each a pair of dependent LDS round trips, was 4 of the 5 us between the Gram and the factorisation)
  {
    const int ar = nf >> 4; // (uniform) the tile row that holds the rhs row i == nf
    int id_[NT], ib_[NT];
#pragma unroll
    for(int a = 0; a < NT; a++)
    {
      const int i = ty + 16 * a, k = tx + 16 * a;
      id_[a] = idx[(ty == tx && i < nf) ? i : 0];
      ib_[a] = idx[k < nf ? k : 0];
    }
    double dv[NT], bv[NT];
#pragma unroll
    for(int a = 0; a < NT; a++)
    {
      dv[a] = diag[id_[a]];
      bv[a] = bpri[ib_[a]];
    }
#pragma unroll
    for(int a = 0; a < NT; a++)
    {
      const int i = ty + 16 * a;
      acc[a][a] += (ty == tx && i < nf) ? dv[a] : 0.0;
#pragma unroll
      for(int b = 0; b <= a; b++)
        acc[a][b] += (a == ar && i == nf && tx + 16 * b < nf) ? bv[b] : 0.0;
    }
  }
  if(dbg_stop == 4) return;
  SOLVE_STAMP(7);
  // factorisation, FOUR columns per barrier (round 2: two; the loop is a chain of barrier -> pivot reciprocals -> update, and
  // its length, not its arithmetic, is what it costs: 38 steps of ~1.9 k cycles for the 76 columns of a motion solve).  The
  // holders of columns j_0 .. j_3 publish their raw entries R_q; one barrier later every thread forms, from those four published
  // columns alone, the corrected columns
  //   C_q = R_q - sum_{s<q} C_s L_qs,   L_qs = C_s[j_q] / d_s,   d_s = C_s[j_s]        (an LDL^T of the 4 x 4 pivot block)
  // for the rows and columns it owns and applies the rank-4 update  a_ik -= sum_s C_s[i] C_s[k] / d_s  to its registers (the
  // elements of column j_q itself take only the terms s < q and so become C_q).  Nothing but four reciprocals sits between the
  // barrier and the update; square roots are taken once, after the loop.
  // lraw: [2 (step parity)][4 (column of the step)][16 NT], zeroed: rows beyond nf are never published.
  constexpr int LS = 16 * NT;
  bool kcol[NT]; // tx + 16 a is a column of the system (not the rhs row, not padding)
#pragma unroll
  for(int a = 0; a < NT; a++) kcol[a] = tx + 16 * a < nf;
  for(int q = tid; q < 8 * LS; q += 256) lraw[q] = 0.0;
  __syncthreads();
  int par = 0;
#pragma unroll
  for(int bj = 0; bj < NT; bj++)
  {
    for(int jj = 0; jj < 16; jj += 4)
    {
      const int j0 = 16 * bj + jj;
      if(j0 >= nf) break; // uniform
      const int ncol = nf - j0 < 4 ? nf - j0 : 4; // uniform: live columns of this step
      double * lb = lraw + (par & 1) * 4 * LS;
      par++;
      if(tx >= jj && tx < jj + ncol) // the holders publish (all four columns live in tile column bj)
      {
        double * lp = lb + (tx - jj) * LS;
        const int jc = j0 + (tx - jj);
#pragma unroll
        for(int a = bj; a < NT; a++)
        {
          const int i = ty + 16 * a;
          if(i >= jc && i <= nf) lp[i] = acc[a][bj];
        }
      }
      __syncthreads();
      // LDL^T of the pivot block from the published entries R_s[j_q], s <= q (broadcast reads); dead columns: inv = 0, L = 0
      double inv[4], L[4][4];
      {
        double Cj[4][4]; // Cj[s][q] = C_s[j_q], q >= s
#pragma unroll
        for(int sidx = 0; sidx < 4; sidx++)
        {
#pragma unroll
          for(int q = sidx; q < 4; q++)
          {
            double v = (q < ncol) ? lb[sidx * LS + j0 + q] : 0.0;
#pragma unroll
            for(int t = 0; t < sidx; t++) v -= Cj[t][q] * L[sidx][t];
            Cj[sidx][q] = v;
          }
          double d = Cj[sidx][sidx];
          if(sidx < ncol && !(d > 0.0)) *bad = 1;
          if(!(sidx < ncol && d > 0.0)) d = 1.0;
          double r = __builtin_amdgcn_rcp(d);
          r = r * (2.0 - d * r);
          r = r * (2.0 - d * r);
          inv[sidx] = (sidx < ncol) ? r : 0.0;
#pragma unroll
          for(int q = sidx + 1; q < 4; q++) L[q][sidx] = (q < ncol) ? Cj[sidx][q] * inv[sidx] : 0.0;
        }
      }
      double ri[4][NT], sk[4][NT]; // per live column s: C_s at this thread's rows, C_s / d_s at its columns (zero where the update does not apply)
#pragma unroll
      for(int a = bj; a < NT; a++)
      {
        const int i = ty + 16 * a, k = tx + 16 * a;
        double ci[4], ck[4];
#pragma unroll
        for(int q = 0; q < 4; q++)
        {
          double vi = lb[q * LS + i], vk = lb[q * LS + k];
#pragma unroll
          for(int t = 0; t < q; t++)
          {
            vi -= ci[t] * L[q][t];
            vk -= ck[t] * L[q][t];
          }
          // (entries above a column's pivot are never published: whatever the slot holds there is masked, here and below)
          ci[q] = (a > bj || i > j0 + q) ? vi : 0.0;
          ck[q] = (a > bj || k > j0 + q) ? vk : 0.0;
          ri[q][a] = (q < ncol) ? ci[q] : 0.0;
          sk[q][a] = (q < ncol && kcol[a]) ? ck[q] * inv[q] : 0.0;
        }
      }
      // (the elements of column j_q itself get only the terms s < q — their sk[s >= q] is zero — and so become the corrected column)
#pragma unroll
      for(int a = bj; a < NT; a++)
#pragma unroll
        for(int b = bj; b <= a; b++)
          acc[a][b] -= (ri[0][a] * sk[0][b] + ri[1][a] * sk[1][b]) + (ri[2][a] * sk[2][b] + ri[3][a] * sk[3][b]);
    }
  }
  __syncthreads();
  SOLVE_STAMP(8);
  // reciprocal pivots 1/sqrt(d_k) from the final diagonal entries, once
#pragma unroll
  for(int a = 0; a < NT; a++)
  {
    const int i = ty + 16 * a;
    if(ty == tx && i < nf)
    {
      double d = acc[a][a];
      if(!(d > 0.0))
      {
        *bad = 1;
        d = 1.0;
      }
      dinv[i] = fast_rsqrt(d);
    }
  }
  __syncthreads();
  // the scaled factor for the back substitution, once: a column's raw entries are final once its pair has been processed
  // (later updates only touch columns to its right), and L_ik = raw_ik / piv_k, piv_k = d_k / sqrt(d_k) = raw_kk * dinv_k
#pragma unroll
  for(int a = 0; a < NT; a++)
#pragma unroll
    for(int b = 0; b <= a; b++)
    {
      const int i = ty + 16 * a, k = tx + 16 * b;
      if(k < nf && i >= k && i <= nf) M[tri_idx(i, k)] = acc[a][b] * dinv[k];
    }
  __syncthreads();
}

// One workgroup per frame.  Everything is built from J (staged through LDS in row chunks) — no D x D matrix in HBM.
// LDS (doubles): M packed (D+1)(D+2)/2 | Jc [chunk][D] | xs, xfull, diag, bpri, lo, hi [D each] | rowv [rows] | lraw [2][96],
// ldiag [2] ; ints idx, state [D].
// DUAL_ONLY: the instantiation for launches whose every pass is known on the host to take the dual form (4K < theta_dim:
// theta is always free, so the free set never shrinks below the residual rows). It does not carry the register-tiled
// primal factorisation, which is what sizes the general kernel's register file footprint (247 of the SIMD's 512
// registers per lane: the face scan that runs beside the solve then keeps one wavefront per SIMD instead of three).
// NTR: tiles of 16 the register-tiled primal factorisation covers (free unknowns + 1 <= 16 NTR): 6 for every mode but the
// 41-marker body solve (phi and beta live: 167 free unknowns), which gets its own instantiation with 11 — a second tile
// count inside one instantiation slowed the common path by 16 us (its register file footprint).
template<bool DUAL_ONLY, int NTR = 6>
__global__ __launch_bounds__(256) void ik_solve_kernel(TaskArrays ta, const double * __restrict__ e_all, const double * __restrict__ J_all,
                                                       float * __restrict__ theta, float * __restrict__ beta, float * __restrict__ pts,
                                                       int K, int theta_dim, int beta_dim, int phi_live, int enable_qp, int use_prior,
                                                       int chunk_rows, const int * __restrict__ skip, double * __restrict__ e2_out,
                                                       int * __restrict__ status, int * __restrict__ sticky, double * __restrict__ x_out, int dbg_stop, int m_dim,
                                                       float * __restrict__ theta25, float * __restrict__ theta_copy,
                                                       unsigned * __restrict__ go_flag, unsigned * __restrict__ go_counter, unsigned go_tick,
                                                       unsigned * __restrict__ done_flag, unsigned * __restrict__ done_counter, unsigned done_tick)
{
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int64_t f = blockIdx.x;
  const int tid = threadIdx.x;
  SOLVE_STAMP(0);
  // "Every workgroup of this kernel is on its CU": the re-projection on the side stream waits for THIS, not for the end of the
  // evaluation.  Both kernels become ready at the same instant, and when the face scan's 1536 workgroups were dispatched first
  // the solve's (one per frame, a whole SIMD's registers per wavefront, 150 KB of LDS) waited for them to drain: 77 us became
  // 105-118 us in most frames of a capture fit, on the critical path.  Nothing is published here (what the scan reads was
  // written by the kernel before this one), so no drain: a counter and, from the last workgroup to arrive, the flag.
  if(go_flag && tid == 0)
  {
    if(__hip_atomic_fetch_add(go_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1)
    {
      __hip_atomic_store(go_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(go_flag, go_tick, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  const int D = theta_dim + 2 * K + beta_dim, rows = 4 * K;
  const int64_t tb = f * K;
  double * M = sm;
  double * Jc = M + (m_dim + 1) * (m_dim + 2) / 2; // m_dim >= the number of free unknowns (host bound): phi pinned => D - 2K
  double * xs = Jc + chunk_rows * D;
  double * xfull = xs + D;
  double * diag = xfull + D;
  double * bpri = diag + D;
  double * lo = bpri + D;
  double * hi = lo + D;
  double * rowv = hi + D;
  double * lraw = rowv + rows; // [2][4][16 NTR] published columns of the register-tiled factorisation (four per step)
  double * ldiag = lraw + 128 * NTR; // [4] (spare)
  double * dinv = ldiag + 4;   // [D] reciprocal pivots for the back substitution
  int * idx = reinterpret_cast<int *>(dinv + D);
  int * state = idx + D; // 0 free, -1 at lo, +1 at hi, 2 pinned (empty box)
  double * ebuf = reinterpret_cast<double *>(state + D); // [rows] the residual, read from HBM once
  __shared__ int s_bad, s_nf, s_block, s_bside, s_done, s_anybound, s_wcnt[4], s_wany[4];
  __shared__ int s_rlist[DUAL_ONLY ? 1 : IK_MAXK * 4], s_nlive; // rows of J that are not identically zero (primal form: build_and_factor_reg)
  __shared__ double s_alpha, s_e2;
  // Everything the set-up reads from HBM is requested NOW, in one round trip: the skip flag, the residual, this thread's limit
  // and prior entry — and, in the dual-only instantiation, the Jacobian block the dual form will gather if only theta turns out
  // free (it does unless a QP pass pins something).  One after the other they were four dependent round trips of ~1.5 us each
  // in a kernel whose whole length is 28 us.
  const double * J = J_all + f * rows * (int64_t)D;
  const int skipf = skip[f];
  double e_pre[(IK_MAXK * 4 + 255) / 256];
#pragma unroll
  for(int u = 0; u < (IK_MAXK * 4 + 255) / 256; u++) e_pre[u] = (tid + 256 * u < rows) ? e_all[f * rows + tid + 256 * u] : 0.0;
  const int my_i = tid < D ? tid : 0; // (D <= 256 on this path: the per-variable set-up below takes one variable per thread then)
  const bool my_phi = my_i >= theta_dim && my_i < theta_dim + 2 * K;
  const float pl_pre = (D <= 256 && my_phi && phi_live) ? ta.philim[tb + (my_i - theta_dim) / 2] : 0.0f;
  const float th_pre = (D <= 256 && use_prior && my_i < theta_dim) ? theta[f * theta_dim + my_i] : 0.0f;
  // primal form: the weight that decides whether a row of J can be non-zero (rows 4k .. 4k+2: the task's position weight —
  // a missing marker has none —, row 4k+3: its normal weight), for the second wavefront's row list
  float rl_pre[(IK_MAXK * 4 + 63) / 64];
  if constexpr(!DUAL_ONLY)
  {
#pragma unroll
    for(int c = 0; c < (IK_MAXK * 4 + 63) / 64; c++)
    {
      const int r = 64 * c + (tid & 63), k = (r < rows ? r : 0) >> 2;
      rl_pre[c] = ta.roww[(tb + k) * 2 + (((r & 3) == 3) ? 1 : 0)]; // (the evaluation's copy: see TaskArrays::roww)
    }
  }
  double j_pre[8];
  if constexpr(DUAL_ONLY)
  {
    const int cnt = rows * theta_dim;
#pragma unroll
    for(int u = 0; u < 8; u++)
    {
      int q = u * 256 + tid;
      q = q < cnt ? q : cnt - 1;
      const int i = q / theta_dim, a = q - i * theta_dim;
      j_pre[u] = J[(int64_t)i * D + a];
    }
  }
  if(skipf)
  {
    if(tid == 0 && e2_out) e2_out[f] = 0.0;
    if(pts)
      for(int k = tid; k < K * 3; k += 256) pts[tb * 3 + k] = ta.apos[tb * 3 + k];
    if(theta_copy)
      for(int i = tid; i < theta_dim; i += 256) theta_copy[f * theta_dim + i] = theta[f * theta_dim + i];
    wg_signal(done_flag, done_counter, done_tick); // (every workgroup of the grid counts itself in)
    return;
  }
  __builtin_amdgcn_s_setprio(3); // a latency chain: its few wavefronts issue ahead of the face scan that shares the CU
#pragma unroll
  for(int u = 0; u < (IK_MAXK * 4 + 255) / 256; u++)
    if(tid + 256 * u < rows) ebuf[tid + 256 * u] = e_pre[u];
  __syncthreads();
  const double * e = ebuf;
  if(tid < 64)
  {
    // |e|^2 (node.cpp:893; Eigen's squaredNorm reduces in packets, so no summation order is "the reference's"): each lane squares
    // and adds its own (up to three) rows, then a fixed butterfly over the 64 lanes — ~400 cycles.  Round 3 walked the rows in
    // ascending order by v_readlane, a chain of `rows` dependent fp64 FMAs: 2 us of a 41-marker solve's set-up.
    double v[3];
#pragma unroll
    for(int a = 0; a < 3; a++) v[a] = (tid + 64 * a < rows) ? e[tid + 64 * a] : 0.0;
    double s = 0.0;
    if(rows <= 192)
    {
      s = v[0] * v[0];
      s = fma(v[1], v[1], s);
      s = fma(v[2], v[2], s);
#pragma unroll
      for(int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    }
    else
      for(int r = 0; r < rows; r++) s += e[r] * e[r];
    if(tid == 0)
    {
      s_e2 = s;
      // bit 2 of the frame's word (this stream's evaluation raised it: a normal term on a vertex beyond MAXADJ faces): its Jacobian
      // rows are truncated, so the update is skipped like one whose factorisation failed — an enqueue-only caller never moves on a
      // wrong Jacobian, and reads the reason in smplpp_ik_get_status
      s_bad = (sticky[f] & 4) ? 1 : 0;
      s_done = 0;
      if(e2_out) e2_out[f] = s;
    }
  }
  else if(!DUAL_ONLY && tid < 128) // beside the sum: the rows of J that can be non-zero, ascending
  {
    const int l = tid - 64;
    int base = 0;
#pragma unroll
    for(int c = 0; c < (IK_MAXK * 4 + 63) / 64; c++)
    {
      const int r = 64 * c + l;
      const bool lv = r < rows && (((r & 3) == 3) ? (rl_pre[c] > 0.0f) : (rl_pre[c] != 0.0f));
      const unsigned long long m = __ballot(lv);
      if(lv) s_rlist[base + __popcll(m & ((1ull << l) - 1ull))] = r;
      base += __popcll(m);
    }
    if(l == 0) s_nlive = base;
  }
  __syncthreads();
  for(int i = tid; i < D; i += 256)
  {
    const double reg = (i < theta_dim) ? 1e-3 : (i < theta_dim + 2 * K ? 1e-1 : 1e-3); // node.cpp:887-892
    double dg = reg + s_e2;                                                               // :893
    double bp = 0.0;
    if(use_prior && i < theta_dim) // :895-904 (VPoser latent layout)
    {
      const double w = (i < 6) ? 0.0 : (i >= theta_dim - 6 ? 1e3 : 1e-5);
      dg += w;
      bp = w * (double)(D <= 256 ? th_pre : theta[f * theta_dim + i]);
    }
    diag[i] = dg;
    bpri[i] = bp;
    // bounds (node.cpp:916-928); theta is free
    double l = -1e30, h = 1e30;
    int st = 0;
    if(i >= theta_dim && i < theta_dim + 2 * K)
    {
      const double pl = phi_live ? (double)(D <= 256 ? pl_pre : ta.philim[tb + (i - theta_dim) / 2]) : 0.0;
      if(enable_qp)
      {
        l = -pl;
        h = pl;
      }
      // with a zero limit the phi columns of J are zero: the QP pins x_phi = 0 and the LLT solution of the
      // block-diagonal system has x_phi = 0 as well, so the variable is removed from the system in both modes
      if(!(pl > 0.0))
      {
        l = 0.0;
        h = 0.0;
        st = 2;
      }
    }
    else if(i >= theta_dim + 2 * K && enable_qp)
    {
      l = -0.5; // :925
      h = 0.5;
    }
    lo[i] = l;
    hi[i] = h;
    state[i] = st;
    xfull[i] = 0.0;
  }
  __syncthreads();

  SOLVE_STAMP(1);
  if(dbg_stop == 1) return; // (timing experiments only: SMPLPP_IK_DBG_STOP)
  const int max_it = enable_qp ? 4 * D + 20 : 1;
  for(int it = 0; it < max_it; it++)
  {
    if(D <= 256)
    {
      // free-set index list in ascending order: one variable per thread, ballot + prefix over the four wavefronts
      // (a single thread walking `state` pays an LDS round trip per variable)
      int st = 2;
      if(tid < D) st = state[tid];
      const int is_free = (st == 0), is_b = ((st == -1 || st == 1) && xfull[tid < D ? tid : 0] != 0.0);
      const uint64_t m = __ballot(is_free);
      const int wave = tid >> 6, lane = tid & 63;
      // (a ballot per wavefront and one barrier: __syncthreads_or funnels every thread through an LDS atomic — 2.7 us here, stamped)
      const uint64_t mbnd = __ballot(is_b);
      if(lane == 0)
      {
        s_wcnt[wave] = __popcll(m);
        s_wany[wave] = mbnd != 0 ? 1 : 0;
      }
      __syncthreads();
      const int anyb = s_wany[0] | s_wany[1] | s_wany[2] | s_wany[3];
      int base = 0;
      for(int w = 0; w < wave; w++) base += s_wcnt[w];
      if(is_free) idx[base + __popcll(m & ((1ull << lane) - 1ull))] = tid;
      if(tid == 0)
      {
        s_nf = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
        s_anybound = anyb;
        s_alpha = 1.0;
        s_block = -1;
      }
    }
    else if(tid == 0)
    {
      int nf = 0, anyb = 0;
      for(int i = 0; i < D; i++)
      {
        if(state[i] == 0) idx[nf++] = i;
        if((state[i] == -1 || state[i] == 1) && xfull[i] != 0.0) anyb = 1;
      }
      s_nf = nf;
      s_anybound = anyb;
      s_alpha = 1.0;
      s_block = -1;
    }
    __syncthreads();
    SOLVE_STAMP(14);
    const int nf = s_nf;
    if(nf > m_dim) // (cannot happen: the host bound counts every variable that can be free)
    {
      if(tid == 0) s_bad = 1;
      __syncthreads();
      break;
    }
    const int nitem = (nf + 1) * (nf + 2) / 2;
    // rowv = e + J_B x_B  (b_F + A_FB x_B = J_F^T rowv); A = J^T J, b = J^T e (node.cpp:884-885), fp64
    for(int r = tid; r < rows; r += 256)
    {
      double s = e[r];
      if(s_anybound)
        for(int q = 0; q < D; q++)
          if(state[q] == -1 || state[q] == 1) s += J[(int64_t)r * D + q] * xfull[q];
      rowv[r] = s;
    }
    SOLVE_STAMP(15);
    const bool dual = rows < nf && rows <= 63 && chunk_rows >= rows && nf <= 192 && (DUAL_ONLY || dbg_stop != 9);
    if(DUAL_ONLY && !dual) // (cannot happen: the host selects this instantiation only when every pass qualifies)
    {
      if(tid == 0) s_bad = 1;
      __syncthreads();
      break;
    }
    if(dual)
    {
      __syncthreads();
      solve_dual<DUAL_ONLY>(M, J, rowv, Jc, diag, bpri, idx, nf, D, rows, dinv, lraw, lraw + 192, xs, &s_bad, dbg_stop, j_pre, theta_dim);
      if(dbg_stop >= 31 && dbg_stop <= 34) return;
    }
    else if constexpr(DUAL_ONLY)
    {
    }
    else if(nf + 1 <= 16 * NTR)
    {
      // registers, one barrier per column
      __syncthreads();
      build_and_factor_reg<NTR>(M, J, rowv, Jc, diag, bpri, idx, nf, D, rows, chunk_rows, lraw, ldiag, dinv, &s_bad, dbg_stop, s_rlist, s_nlive);
      if(dbg_stop == 4 || dbg_stop == 40 || dbg_stop == 41 || dbg_stop == 42) return;
    }
    else
    {
      for(int item = tid; item < nitem; item += 256) M[item] = 0.0;
      for(int c0 = 0; c0 < rows; c0 += chunk_rows)
      {
        const int cr = (rows - c0 < chunk_rows) ? rows - c0 : chunk_rows;
        __syncthreads();
        stage_rows(Jc, J + (int64_t)c0 * D, cr * D);
        __syncthreads();
        {
          const int ty = tid >> 4, tx = tid & 15; // 16 x 16 tiling of the lower triangle (+ the rhs row i == nf)
          for(int i = ty; i <= nf; i += 16)
          {
            const int ci = (i < nf) ? idx[i] : 0;
            const int jend = (i < nf) ? i : nf - 1;
            for(int j = tx; j <= jend; j += 16)
            {
              const int cj = idx[j];
              double s = 0.0;
              if(i < nf)
                for(int r = 0; r < cr; r++) s += Jc[r * D + ci] * Jc[r * D + cj];
              else
                for(int r = 0; r < cr; r++) s += Jc[r * D + cj] * rowv[c0 + r];
              M[tri_idx(i, j)] += s;
            }
          }
        }
      }
      __syncthreads();
      for(int a = tid; a < nf; a += 256)
      {
        M[tri_idx(a, a)] += diag[idx[a]];
        M[tri_idx(nf, a)] += bpri[idx[a]];
      }
      __syncthreads();
      chol_aug(M, nf, &s_bad, dinv);
    }
    if(dbg_stop == 2) return;
    SOLVE_STAMP(9);
    if constexpr(!DUAL_ONLY)
      if(!dual) back_subst(M, nf, xs, dinv);
    SOLVE_STAMP(10);
    if(dbg_stop == 3) return;
    if(!enable_qp)
    {
      for(int a = tid; a < nf; a += 256) xfull[idx[a]] = -xs[a]; // x = -LLT(A)^-1 b (node.cpp:938)
      __syncthreads();
      break;
    }
    // candidate x_F = -xs ; ratio test against the box: the first variable (ascending free-set order) with the smallest
    // step fraction below 1 blocks.  One variable per thread + a lexicographic (fraction, index) minimum — a single thread
    // walking the free set pays five LDS round trips per variable (10 us for the 75 unknowns of a motion solve)
    if(nf <= 256)
    {
      double al = 2.0;
      int who = 0x7fffffff, side = 0;
      if(tid < nf)
      {
        const int i = idx[tid];
        const double xn = -xs[tid], xo = xfull[i], dx = xn - xo;
        if(xn > hi[i] + 1e-14 && dx > 0)
        {
          al = (hi[i] - xo) / dx;
          side = 1;
        }
        else if(xn < lo[i] - 1e-14 && dx < 0)
        {
          al = (lo[i] - xo) / dx;
          side = -1;
        }
        if(side != 0 && al < 1.0)
          who = tid;
        else
          al = 2.0;
      }
      for(int o = 32; o > 0; o >>= 1)
      {
        const double oal = __shfl_xor(al, o, 64);
        const int owho = __shfl_xor(who, o, 64), oside = __shfl_xor(side, o, 64);
        if(oal < al || (oal == al && owho < who))
        {
          al = oal;
          who = owho;
          side = oside;
        }
      }
      __shared__ double s_ral[4];
      __shared__ int s_rwho[4], s_rside[4];
      if((tid & 63) == 0)
      {
        s_ral[tid >> 6] = al;
        s_rwho[tid >> 6] = who;
        s_rside[tid >> 6] = side;
      }
      __syncthreads();
      if(tid == 0)
      {
        double alpha = 1.0;
        int block = -1, bside = 0;
        for(int w = 0; w < 4; w++)
          if(s_rwho[w] != 0x7fffffff && s_ral[w] < alpha) // ascending wavefront order: ties keep the lower index
          {
            alpha = s_ral[w];
            block = s_rwho[w];
            bside = s_rside[w];
          }
        s_alpha = alpha;
        s_block = block;
        s_bside = bside;
      }
    }
    else if(tid == 0)
    {
      double alpha = 1.0;
      int block = -1, bside = 0;
      for(int a = 0; a < nf; a++)
      {
        const int i = idx[a];
        const double xn = -xs[a], dx = xn - xfull[i];
        if(xn > hi[i] + 1e-14 && dx > 0)
        {
          const double al = (hi[i] - xfull[i]) / dx;
          if(al < alpha) { alpha = al; block = a; bside = 1; }
        }
        else if(xn < lo[i] - 1e-14 && dx < 0)
        {
          const double al = (lo[i] - xfull[i]) / dx;
          if(al < alpha) { alpha = al; block = a; bside = -1; }
        }
      }
      s_alpha = alpha;
      s_block = block;
      s_bside = bside;
    }
    __syncthreads();
    for(int a = tid; a < nf; a += 256) xfull[idx[a]] += s_alpha * (-xs[a] - xfull[idx[a]]);
    __syncthreads();
    if(s_block >= 0)
    {
      if(tid == 0)
      {
        const int i = idx[s_block];
        state[i] = s_bside;
        xfull[i] = s_bside > 0 ? hi[i] : lo[i];
      }
      __syncthreads();
      continue;
    }
    // nothing sits on a bound (every motion-stage solve: phi pinned, beta fixed): the unconstrained step is the optimum
    {
      int atb = 0;
      for(int i = tid; i < D; i += 256) atb |= (state[i] == -1 || state[i] == 1);
      const uint64_t matb = __ballot(atb);
      __syncthreads(); // (s_wany's readers of the free-set step are long past)
      if((tid & 63) == 0) s_wany[tid >> 6] = matb != 0 ? 1 : 0;
      __syncthreads();
      if(!(s_wany[0] | s_wany[1] | s_wany[2] | s_wany[3]))
      {
        if(tid == 0) s_done = 1;
        __syncthreads();
        break;
      }
    }
    // multipliers of the bound variables: g = A x + b = J^T (e + J x) + diag x + bpri.  One wavefront per row, lanes
    // across the columns (a thread per row reads J with a stride of D doubles: 64 cache lines per load instruction)
    for(int r = tid >> 6; r < rows; r += 4)
    {
      double s = 0.0;
      for(int q = tid & 63; q < D; q += 64)
        if(state[q] != 2) s += J[(int64_t)r * D + q] * xfull[q];
      for(int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      if((tid & 63) == 0) rowv[r] = e[r] + s;
    }
    __syncthreads();
    for(int i = tid; i < D; i += 256)
    {
      double viol = 0.0;
      if(state[i] == -1 || state[i] == 1)
      {
        double g = diag[i] * xfull[i] + bpri[i];
        for(int r = 0; r < rows; r++) g += J[(int64_t)r * D + i] * rowv[r];
        viol = (state[i] < 0) ? -g : g; // at lo need g >= 0; at hi need g <= 0
      }
      xs[i] = viol; // xs is free between solves
    }
    __syncthreads();
    if(tid == 0)
    {
      double worst = 1e-12;
      int rel = -1;
      for(int i = 0; i < D; i++)
        if(xs[i] > worst) { worst = xs[i]; rel = i; }
      if(rel < 0)
        s_done = 1;
      else
        state[rel] = 0;
    }
    __syncthreads();
    if(s_done) break;
  }
  if(tid == 0)
  {
    status[f] = s_bad ? 1 : ((enable_qp && !s_done) ? 2 : 0);
    if(s_bad && !(sticky[f] & 4)) sticky[f] = sticky[f] | 1; // survives later solves (sequence driver); bit 2 is the evaluation's (TaskArrays::flags)
  }
  const bool ok = !s_bad;
  // config update (node.cpp:945-968), fp32
  for(int i = tid; i < theta_dim; i += 256)
  {
    float t = theta[f * theta_dim + i];
    if(ok)
    {
      t = t + (float)xfull[i];
      // (done_flag: the decoder's Jacobian kernel on the side stream reads the new latent behind that flag — write-through, signal.h)
      if(done_flag)
        st_agent(&theta[f * theta_dim + i], t);
      else
        theta[f * theta_dim + i] = t;
      // VPoser latent layout: the entries that pass through to theta25 (node.cpp:763-771) are kept current here
      if(theta25 && i < 6) theta25[f * TD75 + i] = t;
      if(theta25 && i >= 38) theta25[f * TD75 + 69 + (i - 38)] = t;
    }
    if(theta_copy) theta_copy[f * theta_dim + i] = t; // the sequence driver's record of this frame's result (last iteration of a frame)
  }
  for(int i = tid; i < beta_dim; i += 256)
    if(ok) beta[f * NB + i] += (float)xfull[theta_dim + 2 * K + i];
  for(int i = tid; pts && i < K * 3; i += 256) // p_k = actualPos_k + tangents_k . x_phi_k (:956-959); null: x_phi = 0 for all
  {
    const int k = i / 3, x = i % 3;
    const float p0 = ok ? (float)xfull[theta_dim + 2 * k] : 0.0f, p1 = ok ? (float)xfull[theta_dim + 2 * k + 1] : 0.0f;
    pts[tb * 3 + i] = ta.apos[tb * 3 + i] + (ta.tang[(tb + k) * 6 + x * 2] * p0 + ta.tang[(tb + k) * 6 + x * 2 + 1] * p1);
  }
  if(x_out)
    for(int i = tid; i < D; i += 256) x_out[f * D + i] = xfull[i];
  SOLVE_STAMP(11);
  // "this configuration is final": what the capture loops' side stream waits for before it makes the NEXT decoder Jacobian
  wg_signal(done_flag, done_counter, done_tick);
}

// node.cpp:970-1001 — re-projection of the K query points of every frame onto that frame's posed mesh.
//
// Uncoalesced 12-byte vertex gathers bound this step (every face needs three), so a face is gathered ONCE per frame and
// tested against all K queries: proj_scan_kernel (one workgroup per frame x face chunk) culls with the bounding-sphere
// test against each query's hint distance (exact distance to the task's current face), evaluates the exact distance of
// the few survivors and appends (distance, face) to a short per-(frame, task) list; proj_finish_kernel (one workgroup per
// frame) takes the minimum of each list, applies the tie rule (lowest face id within 1e-6 relative of the minimum — every
// face in that band passes the cull, whose slack is larger) and writes the new face id and area-ratio weights.  A list
// that overflows (a far-off hint, e.g. the very first iteration) falls back to the exhaustive block scan.
constexpr int PROJ_LIST = 512; // (generous since the lists only take faces at least as close as the task's own: see proj_scan_kernel)
constexpr int PROJ_MAXK = IK_MAXK;

typedef float f32x2 __attribute__((ext_vector_type(2)));

// KPR > 0: the (at most 2 * KPR) queries live in registers as KPR packed pairs and the cull runs on packed fp32
// (v_pk_add / v_pk_fma: two queries per instruction, no LDS read per (face, query)); KPR == 0: any K, queries from LDS.
// `hint` (nullable): squared distance of each query to its task's own face when the evaluation already has it.
// NBT: faces a thread takes per batch (all of a batch's loads are issued before any of its tests).  A chunk of at most 3 x 256 faces
// — the 64-chain capture fit: 13776 faces / 24 chunks = 574 — runs with 3: with 6, a thread's batch held 2.2 real faces and 3.8
// placeholders whose nine gathers each were issued all the same.
template<int KPR, int NBT = CP_BATCH>
__global__ __launch_bounds__(256) void proj_scan_kernel(ModelView mv, TaskArrays ta, const float * __restrict__ verts_all,
                                                         const float * __restrict__ pts, const float * __restrict__ hint, int64_t F,
                                                         int K, int chunks, const int * __restrict__ skip, int * __restrict__ list_cnt,
                                                         float * __restrict__ list_d, int * __restrict__ list_f, int dbg_stop)
{
  const int64_t f = blockIdx.x / chunks;
  const int chunk = blockIdx.x % chunks;
  if(skip[f]) return;
  const float * verts = verts_all + f * mv.V * 3;
  __shared__ float sp[PROJ_MAXK + 1][3];
  __shared__ float sreach[PROJ_MAXK + 1]; // sqrt of the hint distance: the cull radius of query k
  __shared__ float sbound[PROJ_MAXK + 1]; // the hint distance itself (squared), with slack: no candidate farther than that can win
  const int64_t tb = f * K;
  if((int)threadIdx.x < K)
  {
    const int k = threadIdx.x;
    const float * p = pts + (tb + k) * 3;
    sp[k][0] = p[0];
    sp[k][1] = p[1];
    sp[k][2] = p[2];
    float d;
    if(hint)
      d = hint[tb + k];
    else
    {
      float c[3];
      d = tri_sqdist_dev(verts, mv.faces, ta.face[tb + k], p, c);
    }
    sreach[k] = (d == d) ? sqrtf(d) : INFINITY;
    // The task's own face is a candidate, at exactly this distance (same evaluation): the minimum is <= it, and every face the
    // tie rule may prefer lies within 1e-6 relative of the minimum.  Survivors of the sphere cull beyond that bound are not
    // listed at all — the lists shrink from hundreds of entries (every face inside the cull sphere of a marker 15 mm off a
    // densely triangulated region: they overflowed in two of three frames of sample_walk.c3d and sent the finish kernel to
    // its exhaustive fallback) to the handful of faces at least as close as the current one.
    sbound[k] = (d == d) ? d * 1.00001f + 1e-30f : INFINITY;
  }
  else if((int)threadIdx.x == K) // the odd pair's second half: a query no face can reach
  {
    sp[K][0] = sp[K][1] = sp[K][2] = 1e18f;
    sreach[K] = 0.0f;
    sbound[K] = 0.0f;
  }
  __syncthreads();
  if(dbg_stop == 10) return; // (timing experiments only: SMPLPP_IK_DBG_STOP)
  f32x2 qx[KPR > 0 ? KPR : 1], qy[KPR > 0 ? KPR : 1], qz[KPR > 0 ? KPR : 1], qs[KPR > 0 ? KPR : 1];
  if(KPR > 0)
  {
#pragma unroll
    for(int q = 0; q < KPR; q++)
    {
      const int k0 = (2 * q < K) ? 2 * q : K, k1 = (2 * q + 1 < K) ? 2 * q + 1 : K;
      qx[q] = f32x2{sp[k0][0], sp[k1][0]};
      qy[q] = f32x2{sp[k0][1], sp[k1][1]};
      qz[q] = f32x2{sp[k0][2], sp[k1][2]};
      qs[q] = f32x2{sreach[k0], sreach[k1]};
    }
  }
  const int64_t per = (F + chunks - 1) / chunks;
  const int64_t f_lo = chunk * per, f_hi = (f_lo + per < F) ? f_lo + per : F;
  for(int64_t base = f_lo + threadIdx.x; base < f_hi; base += (int64_t)blockDim.x * NBT)
  {
    TriBatchT<NBT> t;
    load_tri_batch(verts, mv.faces, f_hi, base, blockDim.x, t);
    if(dbg_stop == 11) { if(t.v[0][0] == 12345.678f) list_cnt[0] = 1; continue; }
#pragma unroll
    for(int b = 0; b < NBT; b++)
    {
      if(!t.valid[b]) continue;
      const int64_t face = base + (int64_t)b * blockDim.x;
      const float * a = t.v[b];
      // bounding sphere about the centroid (tighter than the one about v0 used by the exhaustive scan)
      const float g[3] = {(a[0] + a[3] + a[6]) * (1.0f / 3.0f), (a[1] + a[4] + a[7]) * (1.0f / 3.0f), (a[2] + a[5] + a[8]) * (1.0f / 3.0f)};
      float r2 = 0.0f;
#pragma unroll
      for(int c = 0; c < 3; c++)
      {
        const float dx = a[c * 3] - g[0], dy = a[c * 3 + 1] - g[1], dz = a[c * 3 + 2] - g[2];
        r2 = fmaxf(r2, dx * dx + dy * dy + dz * dz);
      }
      const float r = __builtin_amdgcn_sqrtf(r2) * 1.000001f; // hardware sqrt (1 ulp) with its error folded into the slack
      // branch-free cull over the queries (one divergent branch per face, not per (face, query)), survivors afterwards
      uint64_t hit = 0;
      if(KPR > 0)
      {
#pragma unroll
        for(int q = 0; q < KPR; q++)
        {
          const f32x2 dx = qx[q] - g[0], dy = qy[q] - g[1], dz = qz[q] - g[2];
          const f32x2 d0 = dx * dx + dy * dy + dz * dz;
          const f32x2 reach = (qs[q] + r) * 1.00001f + 2e-6f;
          const f32x2 rr = reach * reach;
          hit |= (d0.x <= rr.x) ? (1ull << (2 * q)) : 0ull;
          hit |= (d0.y <= rr.y) ? (1ull << (2 * q + 1)) : 0ull;
        }
      }
      else
      {
        for(int k = 0; k < K; k++)
        {
          const float d0 = (sp[k][0] - g[0]) * (sp[k][0] - g[0]) + (sp[k][1] - g[1]) * (sp[k][1] - g[1]) + (sp[k][2] - g[2]) * (sp[k][2] - g[2]);
          const float reach = (sreach[k] + r) * 1.00001f + 2e-6f;
          hit |= (d0 <= reach * reach) ? (1ull << k) : 0ull;
        }
      }
      while(hit)
      {
        const int k = __builtin_ctzll(hit);
        hit &= hit - 1;
        // survivor: exact distance from the vertices already in registers (the shared, non-inlined evaluation)
        const float d = tri_sqdist_vals(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], sp[k][0], sp[k][1], sp[k][2]).x;
        if(!(d <= sbound[k])) continue; // farther than the task's own face: cannot be the closest (nor tie with it)
        const int slot = atomicAdd(&list_cnt[tb + k], 1);
        if(slot < PROJ_LIST)
        {
          list_d[(tb + k) * PROJ_LIST + slot] = d;
          list_f[(tb + k) * PROJ_LIST + slot] = (int)face;
        }
      }
    }
  }
}

__device__ __forceinline__ void proj_finish_body(const ModelView & mv, const TaskArrays & ta, const float * __restrict__ verts_all,
                                                 const float * __restrict__ pts, int64_t F, int K,
                                                 const int * __restrict__ skip, int * __restrict__ list_cnt,
                                                 const float * __restrict__ list_d, const int * __restrict__ list_f,
                                                 int * __restrict__ dbg, int tsplit)
{
  // grid = n * tsplit: with few frames per GPU a frame's tasks are shared out (see ik_eval_kernel); the exhaustive fallback
  // below is sequential over a workgroup's tasks
  const int64_t f = blockIdx.x / tsplit;
  const int part = (int)(blockIdx.x % tsplit), per_part = (K + tsplit - 1) / tsplit;
  const int k_begin = part * per_part, k_end = (k_begin + per_part < K) ? k_begin + per_part : K;
  const int64_t tb = f * K;
  if(skip[f]) return;
  const float * verts = verts_all + f * mv.V * 3;
  __shared__ int s_face[PROJ_MAXK];
  __shared__ int s_slow[PROJ_MAXK];
  // list minimum + tie rule: 32 lanes per task, eight tasks per pass (one thread per task walked its list with a dependent
  // global load per entry)
  for(int k0 = k_begin; k0 < k_end; k0 += 8)
  {
    const int k = k0 + (int)threadIdx.x / 32, l = (int)threadIdx.x % 32;
    const bool live = k < k_end;
    const int cnt = live ? list_cnt[tb + k] : 0;
    const bool usable = cnt >= 1 && cnt <= PROJ_LIST;
    const float * ld = list_d + (tb + (live ? k : 0)) * PROJ_LIST;
    const int * lf = list_f + (tb + (live ? k : 0)) * PROJ_LIST;
    float mn = INFINITY;
    if(usable)
      for(int q = l; q < cnt; q += 32) mn = fminf(mn, ld[q]);
    for(int o = 16; o > 0; o >>= 1) mn = fminf(mn, __shfl_xor(mn, o, 32));
    const float thr = mn * (1.0f + 1e-6f) + 1e-12f;
    int best = 0x7fffffff;
    if(usable)
      for(int q = l; q < cnt; q += 32)
        if(ld[q] <= thr && lf[q] < best) best = lf[q];
    for(int o = 16; o > 0; o >>= 1)
    {
      const int other = __shfl_xor(best, o, 32);
      best = other < best ? other : best;
    }
    if(live && l == 0)
    {
      const int face = (usable && best != 0x7fffffff) ? best : -1;
      list_cnt[tb + k] = 0; // ready for the next iteration
      s_face[k] = face;
      s_slow[k] = (face < 0) ? 1 : 0;
      if(dbg)
      {
        atomicAdd(&dbg[0], 1);
        if(cnt == 0) atomicAdd(&dbg[1], 1);
        if(cnt > PROJ_LIST) atomicAdd(&dbg[2], 1);
        if(face < 0 && usable) atomicAdd(&dbg[3], 1);
        atomicMax(&dbg[4], cnt);
      }
    }
  }
  __syncthreads();
  for(int k = k_begin; k < k_end; k++) // rare: exhaustive scan for the tasks whose list overflowed (or was empty / NaN)
  {
    if(!s_slow[k]) continue; // uniform across the workgroup
    __shared__ int64_t s_f64;
    closest_point_block(verts, mv.faces, F, pts + (tb + k) * 3, &s_f64, nullptr, nullptr, ta.face[tb + k]);
    if(threadIdx.x == 0) s_face[k] = (int)s_f64;
    __syncthreads();
  }
  if(k_begin + (int)threadIdx.x < k_end)
  {
    const int k = k_begin + threadIdx.x;
    const int face = s_face[k];
    float tri[9], w[3];
    const float * qp = pts + (tb + k) * 3;
    const float q0 = qp[0], q1 = qp[1], q2 = qp[2];
    for(int i = 0; i < 3; i++)
      for(int x = 0; x < 3; x++) tri[i * 3 + x] = verts[3 * mv.faces[face * 3 + i] + x];
    // the closest point from the triangle already in registers, through the one shared evaluation (tri_sqdist_dev would gather
    // the face's vertices a second time: two more dependent round trips in a kernel that is nothing but round trips)
    const float4 cp = tri_sqdist_vals(tri[0], tri[1], tri[2], tri[3], tri[4], tri[5], tri[6], tri[7], tri[8], q0, q1, q2);
    const float c[3] = {cp.y, cp.z, cp.w};
    triangle_weights_dev(c, tri, w); // calcVertexWeights(closest point), phi_ == 0 (:997-998)
    st_agent(&ta.face[tb + k], face); // (read by the evaluation on the other stream: see wg_signal)
    for(int i = 0; i < 3; i++) st_agent(&ta.vw[(tb + k) * 3 + i], w[i]);
  }
}

__global__ __launch_bounds__(256) void proj_finish_kernel(ModelView mv, TaskArrays ta, const float * __restrict__ verts_all,
                                                           const float * __restrict__ pts, int64_t F, int K,
                                                           const int * __restrict__ skip, int * __restrict__ list_cnt,
                                                           const float * __restrict__ list_d, const int * __restrict__ list_f,
                                                           int * __restrict__ dbg, int tsplit, unsigned * __restrict__ sig_flag,
                                                           unsigned * __restrict__ sig_counter, unsigned sig_tick,
                                                           const float * __restrict__ next_tpos, const uint8_t * __restrict__ next_valid,
                                                           int next_shared)
{
  if(next_tpos) // the sequence driver's frame switch (SeqHook): the evaluation that read the old targets is over, the next one
                // waits for this kernel; the solve running beside it takes its row list from ta.roww, not from posw
  {
    const int64_t f = blockIdx.x / tsplit;
    const int part = (int)(blockIdx.x % tsplit), per_part = (K + tsplit - 1) / tsplit;
    const int k = part * per_part + (int)threadIdx.x;
    if((int)threadIdx.x < per_part && k < K)
    {
      const int64_t i = f * K + k, j = next_shared ? (int64_t)k : i; // (shared: one capture for every chain, [K] per frame of the sequence)
      const bool v = next_valid[j] != 0;
      // write-through like everything else a kernel of the other stream reads behind the flag (wg_signal drains this
      // workgroup's stores to its XCD's L2, not to memory; the next evaluation's workgroups sit on other XCDs)
      st_agent(&ta.posw[i], v ? 1.0f : 0.0f);
      for(int x = 0; x < 3; x++) st_agent(&ta.tpos[i * 3 + x], v ? next_tpos[j * 3 + x] : 0.0f);
    }
  }
  proj_finish_body(mv, ta, verts_all, pts, F, K, skip, list_cnt, list_d, list_f, dbg, tsplit);
  wg_signal(sig_flag, sig_counter, sig_tick);
}

__global__ void clear_bits_kernel(int * p, int bits, int64_t n)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(i < n) p[i] &= ~bits;
}
__global__ void fill_f32_kernel(float * p, float v, int64_t n)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(i < n) p[i] = v;
}
__global__ void fill_nrm_kernel(float * p, int64_t n)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(i < n) p[i] = (i % 3 == 2) ? 1.0f : 0.0f;
}
__global__ void i64_to_i32_kernel(const int64_t * a, int32_t * b, int64_t n)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(i < n) b[i] = (int32_t)a[i];
}
__global__ void i32_to_i64_kernel(const int32_t * a, int64_t * b, int64_t n)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(i < n) b[i] = a[i];
}
__global__ void f64_to_f32_kernel(const double * a, float * b, int64_t n)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(i < n) b[i] = (float)a[i];
}
} // namespace smplpp_hip

using namespace smplpp_hip;

struct smplpp_ik
{
  smplpp_model * m = nullptr;
  smplpp_vposer * vp = nullptr;
  int64_t n = 0, K = 0;
  int64_t frame_base = 0; // global index of frame 0 when this solver holds a shard of a larger job (smplpp_ik_set_frame_base)
  int theta_dim = TD75;
  TaskArrays ta{};
  float *theta = nullptr, *beta = nullptr, *theta25 = nullptr, *vjac = nullptr;
  float *verts = nullptr, *rest = nullptr, *joints = nullptr, *poserot = nullptr, *pts = nullptr;
  double *e = nullptr, *J = nullptr, *Jl = nullptr, *e2 = nullptr, *xout = nullptr;
  int *skip = nullptr, *status = nullptr, *sticky = nullptr, *list_cnt = nullptr, *list_f = nullptr;
  int * range_word = nullptr; // this solver's own "an operand left the fp16x2 form's range" word (status bit 3): its loops' forward passes report here
  int32_t * roles = nullptr; // [DMAX][EVAL_NT] the chain-derivative entries of every thread of ik_eval_kernel (slot u: row u)
  float * list_d = nullptr;
  std::vector<void *> owned;
  bool have_eval = false;
  // re-projection beside the solve: when no task's surface coordinates can move (phiLimit_ <= 0 everywhere, or the
  // motion stage's forced zero, node.cpp:699) the query points are the actual positions the evaluation already wrote,
  // so the face scan does not depend on the solve and runs on a second stream while the solve is in flight
  // The posed mesh is double-buffered so that the side stream can still read iteration i's mesh (scan + finish) while the
  // main stream already writes iteration i+1's; the join sits in front of iteration i+1's evaluation, the first kernel that
  // reads what the finish kernel wrote (face, weights). Both events ride on a kernel's own completion signal
  // (hipExtLaunchKernelGGL): a separate hipEventRecord costs the recording stream ~7 us per iteration.
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool phi_locked = false;
  bool side_pending = false; // a finish kernel is in flight on the side stream; ev_join / the join flag marks its end
  // hand-over between the two streams through device flags (wg_signal + hipStreamWaitValue32) instead of events: [0] fork
  // flag, [16] its workgroup counter, [32] join flag, [48] its counter (one 64-byte line each)
  int * dbg_buf = nullptr;
  unsigned * sig = nullptr;
  unsigned tick_fork = 0, tick_join = 0, tick_done = 0; // ([64] / [80]: the solve's "configuration final" flag and its counter)
  bool use_flags = false;
  // Latent layout with few frames (a capture fit's chains: one decoder workgroup per frame, most of the chip idle).  The decoder's
  // VALUE is all the pose step, the fused kernel and the evaluation's direct rows need; its Jacobian (two thirds of the kernel's
  // 29 us) only the pull-back behind them.  So when another iteration follows, the side stream — idle once scan + finish are done,
  // well before the solve ends — waits for the solve's "configuration final" flag and makes the NEXT iteration's Jacobian there
  // (vposer_jac2_kernel<NF, false>, `out` null, raising the join flag at its end), while the main stream decodes the value with the
  // kernel's value-only instantiation (<NF, true>: the same bits, 18 us), poses, skins and joins: the Jacobian is there when the
  // evaluation (which pulls its rows back through it) starts.  Same kernels' arithmetic, another schedule: bit-identical
  // (tests/test_mocap_gpu.py).  SMPLPP_IK_LATENT_SPLIT=0/1 (read at creation) overrides the n <= 128 rule.
  double last_enqueue_us = 0.0; // host time of the last smplpp_ik_solve_sequence's / smplpp_ik_iterate's enqueue loop
  bool latent_split = false;
  bool jac_ahead = false; // the decoder Jacobian of the CURRENT configuration is (being) made on the side stream; the join flag follows it
  // development switches, read ONCE at creation (never in the per-call path): SMPLPP_DEBUG_SYNC, SMPLPP_IK_DBG_STOP,
  // SMPLPP_IK_OVERLAP=0 (re-projection behind the solve on one stream), SMPLPP_SCAN_BLOCKS
  bool dbg_sync = false, overlap_ok = true;
  int dbg_stop = 0;
  int64_t scan_blocks = 1536;
  int scan_form = -1; // development switch SMPLPP_SCAN_FORM (read at creation): 0 forces the K > 8 instantiations of the face scan
  float * vbuf[2] = {nullptr, nullptr};
  int vcur = 0;
};

template<class T>
static hipError_t dalloc(smplpp_ik * s, T ** p, size_t count)
{
  hipError_t e = hipMalloc((void **)p, sizeof(T) * std::max<size_t>(count, 1));
  if(e == hipSuccess) s->owned.push_back(*p);
  return e;
}

static ModelView view_of(const smplpp_model * m)
{
  ModelView mv;
  mv.faces = m->faces;
  mv.adjOff = m->adjOff;
  mv.adjFace = m->adjFace;
  mv.parent = m->parent;
  mv.wIdx = m->wIdx;
  mv.wVal = m->wVal;
  mv.wSum = m->wSum;
  mv.Pvm = m->Pvm;
  mv.Svm = m->Svm;
  mv.JS = m->JS;
  mv.faceRing = m->faceRing;
  mv.faceMap = m->faceMap;
  mv.anc = m->anc;
  mv.nlev = m->nlev;
  mv.V = m->V;
  mv.maxw = m->maxw;
  return mv;
}

extern "C" int smplpp_ik_destroy(smplpp_ik * s)
{
  if(!s) return SMPLPP_OK;
  (void)hipSetDevice(s->m->device);
  if(s->side) (void)hipStreamSynchronize(s->side);
  if(s->ev_fork) (void)hipEventDestroy(s->ev_fork);
  if(s->ev_join) (void)hipEventDestroy(s->ev_join);
  if(s->side) (void)hipStreamDestroy(s->side);
  for(void * p : s->owned) (void)hipFree(p);
  delete s;
  return SMPLPP_OK;
}

extern "C" int smplpp_ik_set_frame_base(smplpp_ik * s, int64_t frame_base)
{
  if(!s || frame_base < 0) return fail(SMPLPP_ERR_INVALID, "smplpp_ik_set_frame_base: bad argument");
  s->frame_base = frame_base;
  s->jac_ahead = false;
  return SMPLPP_OK;
}

extern "C" int smplpp_ik_create(smplpp_model * m, int64_t n, int64_t K, smplpp_vposer * vposer, smplpp_ik ** out)
{
  if(!m || !out || n <= 0 || K <= 0) return fail(SMPLPP_ERR_INVALID, "smplpp_ik_create: bad argument");
  *out = nullptr;
  if(m->F <= 0) return fail(SMPLPP_ERR_INVALID, "smplpp_ik_create: the model has no faces");
  if(m->nlev > DMAX) return fail(SMPLPP_ERR_INVALID, "smplpp_ik_create: kinematic trees deeper than 12 levels are not supported");
  if(m->V > 65535) return fail(SMPLPP_ERR_INVALID, "smplpp_ik_create: at most 65535 vertices are supported (ring tables hold 16-bit ids)");
  if(!m->faceRing || !m->faceMap || !m->anc) return fail(SMPLPP_ERR_INVALID, "smplpp_ik_create: the model carries no ring tables");
  // the evaluation advances the chain derivatives one tree level per step with one thread per (joint of the level, ancestor
  // depth, axis, row): the per-thread entries of every level (ik_eval_kernel: role), and every level must fit the workgroup
  // (SMPL: at most 5 x 9 x 9 = 405 of 1024)
  std::vector<int32_t> roles((size_t)DMAX * EVAL_NT, -1);
  {
    std::vector<int> depth(NJ, 0);
    std::vector<std::vector<int>> at(NJ + 1);
    for(int i = 0; i < NJ; i++)
    {
      depth[i] = i ? depth[m->h_parent[i]] + 1 : 0;
      at[depth[i]].push_back(i);
    }
    // every (joint i, ancestor depth da <= depth(i), axis, row) once, dealt to the threads ROUND-ROBIN: entry e goes to thread
    // e % EVAL_NT as its e / EVAL_NT-th (SMPL: 1.2 k entries, two per thread at most).  (Rounds 1-3 filled row L with the entries
    // of the joints at tree level L, the order their level-by-level recurrence needed; the closed form has no order, and with
    // that filling the first wavefronts held nine entries each while the last held none.)
    size_t e = 0;
    for(int L = 0; L < DMAX; L++)
    {
      const int per = 9 * (L + 1);
      for(int t = 0; t < (int)at[L].size() * per; t++, e++)
      {
        if(e >= roles.size()) return fail(SMPLPP_ERR_INVALID, "smplpp_ik_create: kinematic tree too wide for the evaluation kernel");
        const int ji = t / per, rem = t % per, da = rem / 9, a9 = rem % 9, i = at[L][ji];
        roles[e] = i | ((m->h_parent[i] & 31) << 5) | ((3 * da + a9 / 3) << 10) | ((a9 % 3) << 16) | ((da == L ? 1 : 0) << 18);
      }
    }
  }
  if(K > PROJ_MAXK) return fail(SMPLPP_ERR_INVALID, "smplpp_ik_create: at most 48 tasks per frame are supported");
  if(TD75 + 2 * K + NB > MAXD)
    return fail(SMPLPP_ERR_INVALID, "smplpp_ik_create: too many tasks for the in-LDS solver (75 + 2K + 10 must be <= 181)");
  // (a vertex with more than 16 adjacent faces — the widest table the evaluation is instantiated for — does not stop the solver from being created: position-only tasks anywhere and
  // normal-term tasks away from such a vertex are unaffected; a normal-term task that touches one is reported when it is evaluated)
  HIP_TRY(hipSetDevice(m->device));
  smplpp_ik * s = new smplpp_ik();
  s->m = m;
  s->vp = vposer;
  s->n = n;
  s->K = K;
  s->theta_dim = vposer ? TD44 : TD75;
  {
    const char * e;
    s->dbg_sync = getenv("SMPLPP_DEBUG_SYNC") != nullptr;
    if((e = getenv("SMPLPP_IK_DBG_STOP"))) s->dbg_stop = atoi(e);
    if((e = getenv("SMPLPP_IK_OVERLAP"))) s->overlap_ok = e[0] != '0';
    if(s->dbg_sync) s->overlap_ok = false;
    // workgroups of the face scan.  Few frames: 1536 in all (a capture fit's 64 chains: 24 chunks of 574 faces per frame, measured
    // against 9 / 18 / 36 chunks).  256 frames: TWO chunks per frame — the scan then runs beside kernels that fill the chip
    // themselves (solve, pose, FK: one workgroup per frame or per CU), and fewer, longer scan workgroups take less from them than
    // many short ones: configs[2] 89.2 -> 85.0 us per iteration in three alternating pairs on one box (6 chunks before); 512 frames
    // keep their three (2 and 3 measured level).  SMPLPP_SCAN_BLOCKS overrides.
    s->scan_blocks = (n >= 256 && n < 512) ? 2 * n : 1536;
    if(n >= 512 && K <= 8 && (m->F + 767) / 768 <= 32) s->scan_blocks = n * ((m->F + 767) / 768); // (chunks of at most 768 faces: the 80-register instantiation, below)
    if((e = getenv("SMPLPP_SCAN_BLOCKS"))) s->scan_blocks = atoll(e);
    if((e = getenv("SMPLPP_SCAN_FORM"))) s->scan_form = atoi(e);
    s->latent_split = vposer != nullptr && n <= 128 && s->dbg_stop == 0;
    // (a debug stop ends the solve kernel in front of its "configuration final" flag: never beside the schedule that waits for it)
    if((e = getenv("SMPLPP_IK_LATENT_SPLIT"))) s->latent_split = vposer != nullptr && e[0] != '0' && s->dbg_stop == 0;
  }
  const size_t nk = (size_t)n * K;
  const size_t Dmax = TD75 + 2 * K + NB;
#define A_(field, count)                                         \
  do                                                             \
  {                                                              \
    hipError_t _e = dalloc(s, &s->field, (count));               \
    if(_e != hipSuccess)                                         \
    {                                                            \
      int _rc = hip_fail(_e, #field, __FILE__, __LINE__);        \
      smplpp_ik_destroy(s);                                      \
      return _rc;                                                \
    }                                                            \
  } while(0)
  A_(ta.face, nk);
  A_(ta.vw, nk * 3);
  A_(ta.tang, nk * 6);
  A_(ta.tpos, nk * 3);
  A_(ta.tnrm, nk * 3);
  A_(ta.posw, nk);
  A_(ta.nrmw, nk);
  A_(ta.philim, nk);
  A_(ta.noff, nk);
  A_(ta.apos, nk * 3);
  A_(ta.anrm, nk * 3);
  A_(ta.hint, nk);
  A_(ta.roww, nk * 2);
  A_(theta, (size_t)n * s->theta_dim);
  A_(beta, (size_t)n * NB);
  A_(theta25, (size_t)n * 75);
  A_(vbuf[0], (size_t)n * m->V * 3);
  A_(vbuf[1], (size_t)n * m->V * 3);
  A_(rest, (size_t)n * m->V * 3);
  A_(joints, (size_t)n * NJ * 3);
  A_(poserot, (size_t)n * NJ * 9);
  A_(pts, nk * 3);
  A_(e, nk * 4);
  A_(J, nk * 4 * Dmax);
  A_(e2, (size_t)n);
  A_(xout, (size_t)n * Dmax);
  A_(roles, roles.size());
  A_(list_cnt, nk);
  A_(list_d, nk * PROJ_LIST);
  A_(list_f, nk * PROJ_LIST);
  A_(skip, (size_t)n);
  A_(status, (size_t)n);
  A_(sticky, (size_t)n);
  A_(range_word, 1);
  s->ta.flags = s->sticky;
  if(vposer)
  {
    A_(Jl, nk * 4 * Dmax);
    A_(vjac, (size_t)n * 63 * 32);
  }
#undef A_
  // (from here on a failure releases the solver with everything it already owns: arrays, side stream, events)
#define S_TRY(expr)                                            \
  do                                                           \
  {                                                            \
    hipError_t _e = (expr);                                    \
    if(_e != hipSuccess)                                       \
    {                                                          \
      int _rc = hip_fail(_e, #expr, __FILE__, __LINE__);       \
      smplpp_ik_destroy(s);                                    \
      return _rc;                                              \
    }                                                          \
  } while(0)
  // IkTask defaults (include/smplpp/IkTask.h:54-84)
  auto grid = [](size_t c) { return dim3((unsigned)((c + 255) / 256)); };
  S_TRY(hipMemset(s->ta.face, 0, sizeof(int32_t) * nk));
  fill_f32_kernel<<<grid(nk * 3), 256>>>(s->ta.vw, 1.0f / 3.0f, nk * 3);
  fill_f32_kernel<<<grid(nk * 6), 256>>>(s->ta.tang, 0.0f, nk * 6);
  fill_f32_kernel<<<grid(nk * 3), 256>>>(s->ta.tpos, 0.0f, nk * 3);
  fill_nrm_kernel<<<grid(nk * 3), 256>>>(s->ta.tnrm, nk * 3);
  fill_f32_kernel<<<grid(nk), 256>>>(s->ta.posw, 1.0f, nk);
  fill_f32_kernel<<<grid(nk), 256>>>(s->ta.nrmw, 1.0f, nk);
  fill_f32_kernel<<<grid(nk), 256>>>(s->ta.philim, 0.04f, nk);
  fill_f32_kernel<<<grid(nk), 256>>>(s->ta.noff, 0.0f, nk);
  S_TRY(hipMemset(s->theta, 0, sizeof(float) * n * s->theta_dim));
  S_TRY(hipMemset(s->theta25, 0, sizeof(float) * n * 75));
  S_TRY(hipMemset(s->beta, 0, sizeof(float) * n * NB));
  S_TRY(hipMemset(s->skip, 0, sizeof(int) * n));
  S_TRY(hipMemcpy(s->roles, roles.data(), sizeof(int32_t) * roles.size(), hipMemcpyHostToDevice));
  S_TRY(hipMemset(s->list_cnt, 0, sizeof(int) * nk));
  S_TRY(hipMemset(s->status, 0, sizeof(int) * n));
  S_TRY(hipMemset(s->sticky, 0, sizeof(int) * n));
  S_TRY(hipMemset(s->range_word, 0, sizeof(int)));
  s->verts = s->vbuf[0];
  // (default priority: a lowest-priority side stream — tried against the scan being dispatched ahead of the solve — halved the
  // latent-IK leg of bench.py, where several solvers' streams exist; what fixes that order is the solve kernel's own "all my
  // workgroups run" flag, see ik_solve_kernel)
  S_TRY(hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking));
  S_TRY(hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming));
  S_TRY(hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming));
  {
    S_TRY(dalloc(s, &s->sig, 128));
    S_TRY(hipMemset(s->sig, 0, sizeof(unsigned) * 128));
    // stream memory operations are optional in HIP: probe once (flag 0 >= 0 is satisfied at once); SMPLPP_IK_EVENTS=1 keeps events
    const char * e = getenv("SMPLPP_IK_EVENTS");
    if(!(e && e[0] != '0'))
    {
      s->use_flags = hipStreamWaitValue32(s->side, s->sig, 0u, hipStreamWaitValueGte, 0xffffffffu) == hipSuccess;
      (void)hipGetLastError();
    }
    if(!s->use_flags) s->latent_split = false; // (the hand-overs of that schedule are flags)
  }
  S_TRY(hipDeviceSynchronize());
#undef S_TRY
  *out = s;
  return SMPLPP_OK;
}

// copy caller array -> solver array with optional conversion
template<class Src, class Dst, class Conv>
static int set_array(const Src * src, Dst * dst, size_t count, int space, Conv conv)
{
  if(!src) return SMPLPP_OK;
  In<Src> in;
  HIP_TRY(in.init(src, count, space, nullptr));
  conv(in.d, dst, (int64_t)count);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  return SMPLPP_OK;
}

extern "C" int smplpp_ik_set_tasks(smplpp_ik * s, const int64_t * face_idx, const float * vertex_weights, const float * target_pos,
                                   const float * target_normal, const double * pos_task_weight, const double * normal_task_weight,
                                   const double * phi_limit, const double * normal_offset, int space)
{
  if(!s) return fail(SMPLPP_ERR_INVALID, "smplpp_ik_set_tasks: null solver");
  int rc = check_space(space, "smplpp_ik_set_tasks");
  if(rc) return rc;
  HIP_TRY(hipSetDevice(s->m->device));
  const size_t nk = (size_t)s->n * s->K;
  if(face_idx && space == SMPLPP_HOST)
    for(size_t i = 0; i < nk; i++)
      if(face_idx[i] < 0 || face_idx[i] >= s->m->F) return fail(SMPLPP_ERR_INVALID, "smplpp_ik_set_tasks: face index out of range");
  auto g = [](int64_t c) { return dim3((unsigned)((c + 255) / 256)); };
  auto cpf = [](const float * a, float * b, int64_t c) { (void)hipMemcpy(b, a, sizeof(float) * c, hipMemcpyDeviceToDevice); };
  auto cvd = [&](const double * a, float * b, int64_t c) { f64_to_f32_kernel<<<g(c), 256>>>(a, b, c); };
  auto cvi = [&](const int64_t * a, int32_t * b, int64_t c) { i64_to_i32_kernel<<<g(c), 256>>>(a, b, c); };
  // bit 2 of the status word (a normal term on a vertex beyond MAXADJ faces) belongs to the tasks the evaluation met: new faces,
  // weights or normal terms start clean, and the next evaluation raises it again where it still applies
  if(face_idx || normal_task_weight || normal_offset || vertex_weights)
  {
    clear_bits_kernel<<<g((int64_t)s->n), 256>>>(s->sticky, 4, (int64_t)s->n);
    HIP_TRY(hipGetLastError());
  }
  if((rc = set_array(face_idx, s->ta.face, nk, space, cvi))) return rc;
  if((rc = set_array(vertex_weights, s->ta.vw, nk * 3, space, cpf))) return rc;
  if((rc = set_array(target_pos, s->ta.tpos, nk * 3, space, cpf))) return rc;
  if((rc = set_array(target_normal, s->ta.tnrm, nk * 3, space, cpf))) return rc;
  if((rc = set_array(pos_task_weight, s->ta.posw, nk, space, cvd))) return rc;
  if((rc = set_array(normal_task_weight, s->ta.nrmw, nk, space, cvd))) return rc;
  if((rc = set_array(phi_limit, s->ta.philim, nk, space, cvd))) return rc;
  if((rc = set_array(normal_offset, s->ta.noff, nk, space, cvd))) return rc;
  if(phi_limit)
  {
    std::vector<float> h(nk);
    HIP_TRY(hipMemcpy(h.data(), s->ta.philim, sizeof(float) * nk, hipMemcpyDeviceToHost));
    bool locked = true;
    for(size_t i = 0; i < nk && locked; i++) locked = !(h[i] > 0.0f);
    s->phi_locked = locked;
  }
  return SMPLPP_OK;
}

extern "C" int smplpp_ik_set_config(smplpp_ik * s, const float * beta, const float * theta, int space)
{
  if(!s) return fail(SMPLPP_ERR_INVALID, "smplpp_ik_set_config: null solver");
  int rc = check_space(space, "smplpp_ik_set_config");
  if(rc) return rc;
  HIP_TRY(hipSetDevice(s->m->device));
  hipMemcpyKind kind = space == SMPLPP_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  if(beta) HIP_TRY(hipMemcpy(s->beta, beta, sizeof(float) * s->n * NB, kind));
  if(theta) HIP_TRY(hipMemcpy(s->theta, theta, sizeof(float) * s->n * s->theta_dim, kind));
  // status bit 1 (smplpp_ik_get_status) reports failures "since the configuration was set": a new configuration starts clean
  s->jac_ahead = false; // (a Jacobian made ahead belongs to the configuration it was made for)
  HIP_TRY(hipMemset(s->status, 0, sizeof(int) * s->n));
  HIP_TRY(hipMemset(s->sticky, 0, sizeof(int) * s->n));
  if(s->range_word) HIP_TRY(hipMemset(s->range_word, 0, sizeof(int))); // (status bit 3: same lifetime; this solver's own word)
  if(theta && s->vp) // latent layout: the entries that pass through to theta25 (the decoder fills the rest at every evaluation)
  {
    ik_splice_kernel<<<dim3((unsigned)((s->n * 75 + 255) / 256)), 256>>>(s->theta, nullptr, s->theta25, s->n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(nullptr));
  }
  return SMPLPP_OK;
}

extern "C" int smplpp_ik_get_config(smplpp_ik * s, float * beta, float * theta, int space)
{
  if(!s) return fail(SMPLPP_ERR_INVALID, "smplpp_ik_get_config: null solver");
  int rc = check_space(space, "smplpp_ik_get_config");
  if(rc) return rc;
  HIP_TRY(hipSetDevice(s->m->device));
  HIP_TRY(hipDeviceSynchronize());
  hipMemcpyKind kind = space == SMPLPP_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  if(beta) HIP_TRY(hipMemcpy(beta, s->beta, sizeof(float) * s->n * NB, kind));
  if(theta) HIP_TRY(hipMemcpy(theta, s->theta, sizeof(float) * s->n * s->theta_dim, kind));
  return SMPLPP_OK;
}

extern "C" int smplpp_ik_get_tasks(smplpp_ik * s, int64_t * face_idx, float * vertex_weights, float * tangents, float * actual_pos,
                                   float * actual_normal, int space)
{
  if(!s) return fail(SMPLPP_ERR_INVALID, "smplpp_ik_get_tasks: null solver");
  int rc = check_space(space, "smplpp_ik_get_tasks");
  if(rc) return rc;
  HIP_TRY(hipSetDevice(s->m->device));
  HIP_TRY(hipDeviceSynchronize());
  const size_t nk = (size_t)s->n * s->K;
  hipMemcpyKind kind = space == SMPLPP_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  if(face_idx)
  {
    Out<int64_t> o;
    HIP_TRY(o.init(face_idx, nk, space));
    i32_to_i64_kernel<<<dim3((unsigned)((nk + 255) / 256)), 256>>>(s->ta.face, o.d, (int64_t)nk);
    HIP_TRY(hipGetLastError());
    HIP_TRY(o.finish(nullptr));
    HIP_TRY(hipDeviceSynchronize());
  }
  if(vertex_weights) HIP_TRY(hipMemcpy(vertex_weights, s->ta.vw, sizeof(float) * nk * 3, kind));
  if(tangents) HIP_TRY(hipMemcpy(tangents, s->ta.tang, sizeof(float) * nk * 6, kind));
  if(actual_pos) HIP_TRY(hipMemcpy(actual_pos, s->ta.apos, sizeof(float) * nk * 3, kind));
  if(actual_normal)
  {
    // IkTask::calcActualNormal() evaluated on demand at the current task state (face, weights) and the last posed mesh
    ik_actual_normals_kernel<<<dim3((unsigned)((nk + 63) / 64)), 64>>>(view_of(s->m), s->ta, s->verts, (int)s->K, (int64_t)nk);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(actual_normal, s->ta.anrm, sizeof(float) * nk * 3, kind));
  }
  return SMPLPP_OK;
}

// forward + eval for all frames (enqueue only)
static int ik_forward_eval(smplpp_ik * s, int optimize_beta, int phi_live, int64_t min_valid, hipStream_t st, hipEvent_t eval_done = nullptr)
{
  smplpp_model * m = s->m;
  const int64_t n = s->n;
  const int K = (int)s->K;
  const float * th25 = s->theta;
  // latent_split: this configuration's decoder Jacobian is being made on the side stream (smplpp_ik::jac_ahead) — here only the value
  const bool jac_elsewhere = s->vp && s->jac_ahead;
  {
    TraceRange tr_fwd("forward SMPL"); // node.cpp:752-781 (the VPoser splice is inside that span there too)
    if(s->vp) // node.cpp:761-772
    {
      // the decoder writes its 63 angles straight into theta25[:, 6:69]; the pass-through entries (root translation / rotation,
      // joints 22-23) are kept current by whoever changes the configuration: smplpp_ik_set_config and the solve kernel's update
      int rc = jac_elsewhere ? vposer_forward_device(s->vp, n, s->theta + 6, TD44, s->theta25 + 6, 75, nullptr, st, s->frame_base, true)
                             : vposer_forward_device(s->vp, n, s->theta + 6, TD44, s->theta25 + 6, 75, s->vjac, st, s->frame_base);
      if(rc) return rc;
      th25 = s->theta25;
    }
    s->vcur ^= 1;
    s->verts = s->vbuf[s->vcur];
    int rc = fk_device(m, n, s->beta, th25, s->verts, s->joints, nullptr, s->rest, s->poserot, st, RANGE_INTERNAL, s->range_word); // node.cpp:777
    if(rc) return rc;
  }
  TraceRange tr_eval("calculate IK matrices"); // node.cpp:796-881
  if(s->side_pending) // the previous iteration's re-projection (side stream) wrote the faces / weights read from here on
  {
    if(s->use_flags)
      HIP_TRY(hipStreamWaitValue32(st, s->sig + 32, s->tick_join, hipStreamWaitValueGte, 0xffffffffu));
    else
      HIP_TRY(hipStreamWaitEvent(st, s->ev_join, 0));
    s->side_pending = false;
  }
  s->jac_ahead = false; // (consumed by the evaluation below: the join above covers the Jacobian kernel, which raised it)
  const bool deep = m->nlev > 9; // (ik_eval_kernel's instantiations: see EvalPlan)
  const bool wide = m->madj > MAXADJ; // a topology with a vertex of 13..16 faces: 16-face tables, fewer normal tasks per group
  const size_t shmem = sizeof(float) * (deep ? (wide ? EvalPlan<DMAX, 64, 3>::L_END : EvalPlan<DMAX, 64, 3>::L_END)
                                             : (wide ? EvalPlan<9, 76, 4>::L_END : EvalPlan<9, 76, 6>::L_END)) + L_ANC_BYTES;
  static PerDeviceOnce once_eval[4];
  const void * kfn = deep ? (wide ? reinterpret_cast<const void *>(&ik_eval_kernel<DMAX, 64, 3, MAXADJ_WIDE>) : reinterpret_cast<const void *>(&ik_eval_kernel<DMAX, 64, 3>))
                          : (wide ? reinterpret_cast<const void *>(&ik_eval_kernel<9, 76, 4, MAXADJ_WIDE>) : reinterpret_cast<const void *>(&ik_eval_kernel<9, 76, 6>));
  HIP_TRY(lds_opt_in(once_eval[(deep ? 1 : 0) + (wide ? 2 : 0)], m->device, kfn, (int)shmem));
  int tsplit = (n < 256) ? (int)(256 / n) : 1; // one round of workgroups (one per CU: its LDS is the evaluation's)
  if(tsplit > K) tsplit = K;
  if(tsplit < 1) tsplit = 1;
  if(s->use_flags) eval_done = nullptr; // (flags mode: the fork is the solve kernel's start flag; the evaluation's end is signalled in events mode only)
#define EVAL_(DM, RC, NG, MA)                                                                                                              \
  hipExtLaunchKernelGGL((ik_eval_kernel<DM, RC, NG, MA>), dim3((unsigned)(n * tsplit)), dim3(EVAL_NT), shmem, st, nullptr, eval_done, 0,     \
                        view_of(m), s->ta, th25, (const float *)s->verts, (const float *)s->rest, (const float *)m->ws.Gp.as<float>(),     \
                        (const float *)s->joints, (const float *)s->poserot, K, optimize_beta, phi_live, (int)min_valid, s->pts, s->e,     \
                        s->J, s->skip, s->dbg_stop, tsplit, s->roles, s->vp ? (const float *)s->vjac : (const float *)nullptr,             \
                        s->vp ? s->Jl : (double *)nullptr)
  if(deep && wide)
    EVAL_(DMAX, 64, 3, MAXADJ_WIDE);
  else if(deep)
    EVAL_(DMAX, 64, 3, MAXADJ);
  else if(wide)
    EVAL_(9, 76, 4, MAXADJ_WIDE);
  else
    EVAL_(9, 76, 6, MAXADJ);
#undef EVAL_
  HIP_TRY(hipGetLastError());
  s->have_eval = true;
  return SMPLPP_OK;
}

static int ik_check_valence(smplpp_ik * s);

extern "C" int smplpp_ik_eval(smplpp_ik * s, int optimize_beta, double * e, double * J, int space, void * stream)
{
  if(!s) return fail(SMPLPP_ERR_INVALID, "smplpp_ik_eval: null solver");
  int rc = check_space(space, "smplpp_ik_eval");
  if(rc) return rc;
  HIP_TRY(hipSetDevice(s->m->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  rc = ik_forward_eval(s, optimize_beta, 1, 0, st);
  if(rc) return rc;
  const int64_t D = s->theta_dim + 2 * s->K + (optimize_beta ? NB : 0);
  hipMemcpyKind kind = space == SMPLPP_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  if(e) HIP_TRY(hipMemcpyAsync(e, s->e, sizeof(double) * s->n * s->K * 4, kind, st));
  if(J) HIP_TRY(hipMemcpyAsync(J, s->vp ? s->Jl : s->J, sizeof(double) * s->n * s->K * 4 * D, kind, st));
  if(space == SMPLPP_HOST)
  {
    HIP_TRY(hipStreamSynchronize(st));
    if((rc = ik_check_valence(s))) return rc;
  }
  return SMPLPP_OK;
}

// `iters` iterations enqueued on st (+ the solver's side stream); leaves the last re-projection pending on the side
// stream (s->side_pending) — the caller joins (ik_join) before anything else may touch the task arrays or the mesh.
// What the sequence driver wants done around the LAST of the iterations: the configuration after it recorded (by the solve
// kernel itself) and the NEXT frame's targets put in place (by the re-projection's finish kernel, wherever it runs: the
// evaluation that read the old targets is over by then, nothing else reads them, and the next evaluation waits for it).
struct SeqHook
{
  float * theta_record = nullptr;        // [n][theta_dim]
  const float * next_tpos = nullptr;     // [n][K][3]
  const uint8_t * next_valid = nullptr;  // [n][K]
  int shared = 0;                        // next_tpos / next_valid are [K] / [K][3]: one capture for every chain
};

// more_follows: the caller enqueues another iteration right behind this call's last one (the sequence driver, frame after frame)
static int ik_iterate_enqueue(smplpp_ik * s, int iters, int enable_qp, int optimize_beta_from, int64_t min_valid, hipStream_t st,
                              const SeqHook * hook = nullptr, bool more_follows = false)
{
  int rc = SMPLPP_OK;
  smplpp_model * m = s->m;
  const int K = (int)s->K;
  static PerDeviceOnce once_solve[5];
  HIP_TRY(lds_opt_in(once_solve[0], m->device, reinterpret_cast<const void *>(&ik_solve_kernel<false>), (int)SOLVE_LDS_MAX));
  HIP_TRY(lds_opt_in(once_solve[1], m->device, reinterpret_cast<const void *>(&ik_solve_kernel<true>), (int)SOLVE_LDS_MAX));
  HIP_TRY(lds_opt_in(once_solve[2], m->device, reinterpret_cast<const void *>(&ik_solve_kernel<false, 11>), (int)SOLVE_LDS_MAX));
  HIP_TRY(lds_opt_in(once_solve[3], m->device, reinterpret_cast<const void *>(&ik_solve_kernel<false, 5>), (int)SOLVE_LDS_MAX));
  HIP_TRY(lds_opt_in(once_solve[4], m->device, reinterpret_cast<const void *>(&ik_solve_kernel<false, 3>), (int)SOLVE_LDS_MAX));
  if(s->use_flags && (s->tick_fork > 0x7fff0000u || s->tick_join > 0x7fff0000u || s->tick_done > 0x7fff0000u))
  {
    // the hand-over flags carry iteration numbers compared with >=: start over long before they could wrap
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipStreamSynchronize(s->side));
    HIP_TRY(hipMemset(s->sig, 0, sizeof(unsigned) * 128));
    s->tick_fork = s->tick_join = s->tick_done = 0;
  }
  const bool dbg = s->dbg_sync;
  const int dbg_stop = s->dbg_stop;
  const bool overlap_ok = s->overlap_ok;
  const int64_t scan_blocks = s->scan_blocks;
#define DBG_SYNC(tag)                                                            \
  if(dbg)                                                                        \
  {                                                                              \
    fprintf(stderr, "[smplpp dbg] it %d: %s ...\n", it, tag);                    \
    HIP_TRY(hipStreamSynchronize(st));                                           \
    fprintf(stderr, "[smplpp dbg] it %d: %s done\n", it, tag);                   \
  }
  for(int it = 0; it < iters; it++)
  {
    const int opt_beta = (optimize_beta_from >= 0 && it >= optimize_beta_from) ? 1 : 0; // node.cpp:655
    const int phi_live = (optimize_beta_from >= 0) ? (it >= optimize_beta_from ? 1 : 0) : 1; // :693-700
    // x_phi = 0 for every task (no task's surface coordinates can move): the query points are the actual positions the
    // evaluation wrote, so scan + finish run on the side stream beside the solve and the next iteration's pose / FK
    const bool beside = overlap_ok && (!phi_live || s->phi_locked);
    rc = ik_forward_eval(s, opt_beta, phi_live, min_valid, st, beside ? s->ev_fork : nullptr);
    if(rc) return rc;
    DBG_SYNC("forward+eval");
    const int beta_dim = opt_beta ? NB : 0;
    // LDS plan: packed system + vectors, the rest (up to a 150 KB total) for the J row chunk
    const int D = s->theta_dim + 2 * K + beta_dim, rows = 4 * K;
    // the packed system is sized for the unknowns that CAN be free: a pinned phi (zero limit, node.cpp:567,699) never is,
    // which leaves 75 of the 157 unknowns of a 41-marker motion solve and room for its 164 Jacobian rows in two chunks
    const int m_dim = D - ((!phi_live || s->phi_locked) ? 2 * K : 0);
    // the box of node.cpp:911-929 bounds phi and d beta only: with every phi pinned and beta fixed (each motion-stage solve) no
    // variable has a finite bound, the QP's optimum IS the LLT solution (x = 0 + 1.0 (x_llt - 0): the same bits), and the kernel
    // takes its LLT exit instead of a ratio test and a bound check that cannot find anything (six barriers)
    const int qp_k = (enable_qp && !((!phi_live || s->phi_locked) && beta_dim == 0)) ? 1 : 0;
    // tiles of 16 the register-tiled factorisation covers (176 < m_dim + 1: all-LDS path).  5 (round 4): the motion solve of a capture
    // fit has 75 unknowns that can be free (+ the rhs row = 76 <= 80): 15 register tiles per thread instead of 21 in every rank-4
    // update of its 19 column steps, its own instantiation like 11 (one tile count per instantiation: DESIGN.md §3.3)
    // (and the same fit in the 44-d latent layout has 44 + 1 <= 48: 6 register tiles per thread in its 11 steps)
    const int ntr_primal = (m_dim + 1 <= 48) ? 3 : (m_dim + 1 <= 80) ? 5 : ((m_dim + 1 <= 96 || m_dim + 1 > 176) ? 6 : 11);
    // theta is never bound, so the free set keeps at least theta_dim unknowns: with fewer residual rows than that every pass
    // (also every active-set pass of the QP) takes the dual form.  (Decided up here because the kernel's LDS plan depends on the
    // instantiation's tile count: ik_solve_kernel<true> carries the default, 6.)
    const bool dual_shape = rows < s->theta_dim && rows <= 63 && D <= 192 && dbg_stop != 9;
    const int ntr = dual_shape ? 6 : ntr_primal;
    const size_t fixed = sizeof(double) * ((size_t)(m_dim + 1) * (m_dim + 2) / 2 + 7 * (size_t)D + 2 * (size_t)rows + 128 * (size_t)ntr + 4) + sizeof(int) * 2 * (size_t)D;
    const size_t budget = SOLVE_LDS_MAX;
    int chunk_rows = (int)((budget - fixed) / (sizeof(double) * (size_t)D));
    if(chunk_rows > rows) chunk_rows = rows;
    if(chunk_rows < 4) return fail(SMPLPP_ERR_INVALID, "smplpp_ik_iterate: system too large for the in-LDS solver");
    const size_t solve_shmem = fixed + sizeof(double) * (size_t)chunk_rows * D;
    const bool dual_only = dual_shape && chunk_rows >= rows;
    if(dual_shape && !dual_only) return fail(SMPLPP_ERR_INVALID, "smplpp_ik_iterate: system too large for the in-LDS solver");
    const bool last = hook && it == iters - 1;
    float * theta_record = last ? hook->theta_record : nullptr;
    const bool go = beside && s->use_flags; // the side stream's fork: raised by the solve kernel once all its workgroups run
    if(go) s->tick_fork++;
    // latent_split: the NEXT iteration's decoder Jacobian on the side stream, behind this solve's "configuration final" flag
    const bool ahead = s->latent_split && go && !opt_beta && (it + 1 < iters || more_follows);
    if(ahead) s->tick_done++;
    unsigned * const done_flag = ahead ? s->sig + 64 : (unsigned *)nullptr;
    unsigned * const done_counter = ahead ? s->sig + 80 : (unsigned *)nullptr;
#define SOLVE_(DO) ik_solve_kernel<DO><<<dim3((unsigned)s->n), dim3(256), solve_shmem, st>>>(                                              \
    s->ta, s->e, s->vp ? s->Jl : s->J, s->theta, s->beta, beside ? nullptr : s->pts, K, s->theta_dim, beta_dim, phi_live, qp_k, \
    s->vp ? 1 : 0, chunk_rows, s->skip, s->e2, s->status, s->sticky, s->xout, dbg_stop, m_dim, s->vp ? s->theta25 : (float *)nullptr, theta_record, \
    go ? s->sig : (unsigned *)nullptr, go ? s->sig + 16 : (unsigned *)nullptr, s->tick_fork, done_flag, done_counter, s->tick_done)
#define SOLVE11_(NTR_) ik_solve_kernel<false, NTR_><<<dim3((unsigned)s->n), dim3(256), solve_shmem, st>>>(                                              \
    s->ta, s->e, s->vp ? s->Jl : s->J, s->theta, s->beta, beside ? nullptr : s->pts, K, s->theta_dim, beta_dim, phi_live, qp_k, \
    s->vp ? 1 : 0, chunk_rows, s->skip, s->e2, s->status, s->sticky, s->xout, dbg_stop, m_dim, s->vp ? s->theta25 : (float *)nullptr, theta_record, \
    go ? s->sig : (unsigned *)nullptr, go ? s->sig + 16 : (unsigned *)nullptr, s->tick_fork, done_flag, done_counter, s->tick_done)
    {
      TraceRange tr_solve("solve IK"); // node.cpp:907-943
      if(dual_only)
        SOLVE_(true);
      else if(ntr == 11)
        SOLVE11_(11);
      else if(ntr == 5)
        SOLVE11_(5);
      else if(ntr == 3)
        SOLVE11_(3);
      else
        SOLVE_(false);
    }
#undef SOLVE_
#undef SOLVE11_
    HIP_TRY(hipGetLastError());
    DBG_SYNC("solve");
    {
      TraceRange tr_proj("project point"); // node.cpp:974-988
      const float * qpts = beside ? s->ta.apos : s->pts;
      hipStream_t pst = st;
      if(beside)
      {
        if(s->use_flags)
          HIP_TRY(hipStreamWaitValue32(s->side, s->sig, s->tick_fork, hipStreamWaitValueGte, 0xffffffffu));
        else
          HIP_TRY(hipStreamWaitEvent(s->side, s->ev_fork, 0));
        pst = s->side;
      }
      int chunks = (int)(scan_blocks / s->n);
      chunks = chunks < 1 ? 1 : (chunks > 32 ? 32 : chunks);
      const float * hint = beside ? s->ta.hint : nullptr; // the evaluation's distance is to the ACTUAL position
      const dim3 sg((unsigned)(s->n * chunks));
#define SCAN_(KPR, NBT_) proj_scan_kernel<KPR, NBT_><<<sg, dim3(256), 0, pst>>>(view_of(m), s->ta, s->verts, qpts, hint, m->F, K, chunks, s->skip, \
                                                                   s->list_cnt, s->list_d, s->list_f, dbg_stop)
      const bool small_chunk = (m->F + chunks - 1) / chunks <= 3 * 256; // (a thread then meets at most three faces)
      // K <= 8 with 512 frames and more (configs[4]): the K > 8 instantiation on chunks of at most 768 faces — 80 registers, six
      // wavefronts per SIMD instead of three — is the faster one beside the decoder, whose workgroups wait for the scan's to drain
      // (44.8 against 51.6 us, the latent loop -4 %); at 256 frames the queries-in-registers form stays ahead (81.5 against 84.2 us)
      const bool many = s->scan_form < 0 ? (s->n >= 512 && small_chunk) : s->scan_form == 0;
      if(K <= 8 && many && small_chunk) SCAN_(0, 3);
      else if(K <= 8 && many) SCAN_(0, CP_BATCH);
      else if(K <= 4) SCAN_(2, CP_BATCH);
      else if(K <= 8) SCAN_(4, CP_BATCH);
      else if(small_chunk) SCAN_(0, 3);
      else SCAN_(0, CP_BATCH);
#undef SCAN_
      HIP_TRY(hipGetLastError());
      int *& dbg_buf = s->dbg_buf; // (SMPLPP_DEBUG_SYNC only; owned by the solver, on its device)
      if(dbg && !dbg_buf) HIP_TRY(dalloc(s, &dbg_buf, 8));
      if(dbg) HIP_TRY(hipMemsetAsync(dbg_buf, 0, sizeof(int) * 8, st));
      int fsplit = (s->n < 256) ? (int)(256 / s->n) : 1;
      if(fsplit > K) fsplit = K;
      if(fsplit < 1) fsplit = 1;
      const bool join_flag = beside && s->use_flags && !ahead; // (ahead: the Jacobian kernel behind the finish kernel raises the join)
      if(beside && s->use_flags) s->tick_join++;
      hipExtLaunchKernelGGL(proj_finish_kernel, dim3((unsigned)(s->n * fsplit)), dim3(256), 0, pst, nullptr,
                            (beside && !s->use_flags) ? s->ev_join : nullptr, 0,
                            view_of(m), s->ta, (const float *)s->verts, qpts, m->F, K, (const int *)s->skip, s->list_cnt, s->list_d,
                            s->list_f, dbg ? dbg_buf : (int *)nullptr, fsplit, join_flag ? s->sig + 32 : (unsigned *)nullptr,
                            join_flag ? s->sig + 48 : (unsigned *)nullptr, s->tick_join, last ? hook->next_tpos : (const float *)nullptr,
                            last ? hook->next_valid : (const uint8_t *)nullptr, last ? hook->shared : 0);
      HIP_TRY(hipGetLastError());
      if(ahead)
      {
        HIP_TRY(hipStreamWaitValue32(s->side, s->sig + 64, s->tick_done, hipStreamWaitValueGte, 0xffffffffu));
        rc = vposer_forward_device(s->vp, s->n, s->theta + 6, TD44, nullptr, 75, s->vjac, s->side, s->frame_base, false, s->sig + 32,
                                   s->sig + 48, s->tick_join);
        if(rc) return rc;
        s->jac_ahead = true;
      }
      if(beside) s->side_pending = true;
      if(dbg)
      {
        int h[8];
        HIP_TRY(hipMemcpy(h, dbg_buf, sizeof(h), hipMemcpyDeviceToHost));
        fprintf(stderr, "[smplpp dbg] it %d: project lists: tasks %d, empty %d, overflow %d, nan %d, max cnt %d\n", it, h[0], h[1], h[2], h[3], h[4]);
      }
    }
    DBG_SYNC("project");
  }
#undef DBG_SYNC
  return SMPLPP_OK;
}

static int ik_join(smplpp_ik * s, hipStream_t st)
{
  if(s->side_pending) // everything the caller does next on its stream is ordered behind the last re-projection
  {
    if(s->use_flags)
      HIP_TRY(hipStreamWaitValue32(st, s->sig + 32, s->tick_join, hipStreamWaitValueGte, 0xffffffffu));
    else
      HIP_TRY(hipStreamWaitEvent(st, s->ev_join, 0));
    s->side_pending = false;
  }
  return SMPLPP_OK;
}

static int ik_check_status(smplpp_ik * s, const int * flags)
{
  std::vector<int> h((size_t)s->n);
  HIP_TRY(hipMemcpy(h.data(), flags, sizeof(int) * s->n, hipMemcpyDeviceToHost));
  for(int64_t f = 0; f < s->n; f++)
    if(h[f] & 1) return fail(SMPLPP_ERR_NUMERIC, "LLT has numerical issue!"); // node.cpp:934-937
  return SMPLPP_OK;
}

// bit 2 of the sticky word: raised by the evaluation (see TaskArrays::flags)
static int ik_check_valence(smplpp_ik * s)
{
  std::vector<int> h((size_t)s->n);
  HIP_TRY(hipMemcpy(h.data(), s->sticky, sizeof(int) * s->n, hipMemcpyDeviceToHost));
  for(int64_t f = 0; f < s->n; f++)
    if(h[f] & 4)
      return fail(SMPLPP_ERR_INVALID, "a task with a normal term (normal weight or normal offset) touches a vertex with more than 12 adjacent "
                                      "faces: the Jacobian of such a term is not supported");
  return SMPLPP_OK;
}

extern "C" int smplpp_ik_iterate(smplpp_ik * s, int iters, int enable_qp, int optimize_beta_from, int64_t min_valid,
                                 double * e_sqnorm, int space, void * stream)
{
  if(!s || iters < 0) return fail(SMPLPP_ERR_INVALID, "smplpp_ik_iterate: bad argument");
  int rc = check_space(space, "smplpp_ik_iterate");
  if(rc) return rc;
  HIP_TRY(hipSetDevice(s->m->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const auto enq_t0 = std::chrono::steady_clock::now();
  rc = ik_iterate_enqueue(s, iters, enable_qp, optimize_beta_from, min_valid, st);
  s->last_enqueue_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - enq_t0).count();
  if(rc)
  {
    (void)ik_join(s, st);
    return rc;
  }
  if((rc = ik_join(s, st))) return rc;
  if(e_sqnorm)
  {
    hipMemcpyKind kind = space == SMPLPP_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    HIP_TRY(hipMemcpyAsync(e_sqnorm, s->e2, sizeof(double) * s->n, kind, st));
  }
  if(space == SMPLPP_HOST)
  {
    HIP_TRY(hipStreamSynchronize(st));
    if((rc = ik_check_valence(s))) return rc; // (first: such a frame's solve reports itself as skipped too)
    if((rc = ik_check_status(s, s->status))) return rc;
  }
  return SMPLPP_OK;
}

// node/node.cpp:1369-1407 with :681-700 — the frame loop of solveMocapMotion on the device: frame t's marker targets
// replace the task targets (a missing marker: target 0, weight 0), `warmup_iters` iterations on the first frame and
// `iters_per_frame` on every later one, warm-started; nothing returns to the host between frames.
__global__ void ik_seq_frame_kernel(const float * __restrict__ tpos_t, const uint8_t * __restrict__ valid_t, float * __restrict__ tpos,
                                    float * __restrict__ posw, int64_t nk, const float * __restrict__ theta, float * __restrict__ theta_prev_out,
                                    int64_t ntheta, int64_t shared_K)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(tpos_t && i < nk)
  {
    const int64_t j = shared_K > 0 ? i % shared_K : i; // (shared_K: the targets are one capture's [K] for every chain)
    const bool v = valid_t[j] != 0;
    posw[i] = v ? 1.0f : 0.0f;
    for(int x = 0; x < 3; x++) tpos[i * 3 + x] = v ? tpos_t[j * 3 + x] : 0.0f;
  }
  if(theta_prev_out && i < ntheta) theta_prev_out[i] = theta[i];
}

// Development hook (not part of include/smplpp_hip.h): host microseconds the last smplpp_ik_solve_sequence / smplpp_ik_iterate spent enqueueing.
extern "C" int smplpp_debug_ik_enqueue_us(smplpp_ik * s, double * out)
{
  if(!s || !out) return fail(SMPLPP_ERR_INVALID, "smplpp_debug_ik_enqueue_us: bad argument");
  *out = s->last_enqueue_us;
  return SMPLPP_OK;
}

static int ik_solve_sequence_impl(smplpp_ik * s, int64_t T, const float * target_pos, const uint8_t * valid, bool shared, int warmup_iters,
                                  int iters_per_frame, int enable_qp, int64_t min_valid, float * theta_out, int space, void * stream)
{
  if(!s || T <= 0 || !target_pos || !valid || !theta_out || warmup_iters < 0 || iters_per_frame < 0)
    return fail(SMPLPP_ERR_INVALID, "smplpp_ik_solve_sequence: bad argument");
  int rc = check_space(space, "smplpp_ik_solve_sequence");
  if(rc) return rc;
  HIP_TRY(hipSetDevice(s->m->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t nk = s->n * s->K, ntheta = s->n * s->theta_dim;
  const int64_t tk = shared ? s->K : nk; // targets per frame of the sequence as the caller holds them
  In<float> tp;
  In<uint8_t> vl;
  Out<float> th;
  HIP_TRY(tp.init(target_pos, (size_t)(T * tk * 3), space, st));
  HIP_TRY(vl.init(valid, (size_t)(T * tk), space, st));
  HIP_TRY(th.init(theta_out, (size_t)(T * ntheta), space));
  HIP_TRY(hipMemsetAsync(s->sticky, 0, sizeof(int) * s->n, st));
  const int64_t cnt = nk > ntheta ? nk : ntheta;
  const dim3 grid((unsigned)((cnt + 255) / 256));
  // frame 0's targets go in here; every later switch and every frame's record ride on the iterations themselves (SeqHook): no
  // kernel of its own between one frame's solve and the next frame's pose step
  ik_seq_frame_kernel<<<grid, 256, 0, st>>>(tp.d, vl.d, s->ta.tpos, s->ta.posw, nk, nullptr, nullptr, 0, shared ? s->K : 0);
  HIP_TRY(hipGetLastError());
  const auto enq_t0 = std::chrono::steady_clock::now();
  for(int64_t t = 0; t < T; t++)
  {
    const int iters = t == 0 ? warmup_iters : iters_per_frame;
    SeqHook hook;
    hook.theta_record = th.d + t * ntheta;
    if(t + 1 < T)
    {
      hook.next_tpos = tp.d + (t + 1) * tk * 3;
      hook.next_valid = vl.d + (t + 1) * tk;
      hook.shared = shared ? 1 : 0;
    }
    if(iters > 0)
      rc = ik_iterate_enqueue(s, iters, enable_qp, -1, min_valid, st, &hook, /*more_follows=*/t + 1 < T && iters_per_frame > 0);
    else // (no iteration to carry the hook)
    {
      ik_seq_frame_kernel<<<grid, 256, 0, st>>>(hook.next_tpos, hook.next_valid, s->ta.tpos, s->ta.posw, nk, s->theta, hook.theta_record, ntheta,
                                                shared ? s->K : 0);
      HIP_TRY(hipGetLastError());
    }
    if(rc) break;
  }
  const int jrc = ik_join(s, st);
  // (development figure, smplpp_debug_ik_enqueue_us: what the HOST spent handing the T frames' launches to the two streams — when
  // it approaches the frames' time on the GPU, the chains wait for the host)
  s->last_enqueue_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - enq_t0).count();
  if(rc) return rc;
  if(jrc) return jrc;
  HIP_TRY(th.finish(st));
  if(space == SMPLPP_HOST)
  {
    HIP_TRY(hipStreamSynchronize(st));
    if((rc = ik_check_valence(s))) return rc;
    if((rc = ik_check_status(s, s->sticky))) return rc;
  }
  return SMPLPP_OK;
}

extern "C" int smplpp_ik_solve_sequence(smplpp_ik * s, int64_t T, const float * target_pos, const uint8_t * valid, int warmup_iters,
                                        int iters_per_frame, int enable_qp, int64_t min_valid, float * theta_out, int space, void * stream)
{
  return ik_solve_sequence_impl(s, T, target_pos, valid, false, warmup_iters, iters_per_frame, enable_qp, min_valid, theta_out, space, stream);
}

// The same loop when every chain fits the SAME capture (the multi-restart fit: BASELINE configs[3], 64 restarts x one sequence):
// target_pos [T,K,3] and valid [T,K] once, handed to all n chains by the frame switch on the device — the caller neither builds nor
// uploads n copies (100 MB for 64 restarts of sample_walk.c3d; 83 ms of host work in front of 0.39 s of GPU work).
extern "C" int smplpp_ik_solve_sequence_shared(smplpp_ik * s, int64_t T, const float * target_pos, const uint8_t * valid, int warmup_iters,
                                               int iters_per_frame, int enable_qp, int64_t min_valid, float * theta_out, int space,
                                               void * stream)
{
  return ik_solve_sequence_impl(s, T, target_pos, valid, true, warmup_iters, iters_per_frame, enable_qp, min_valid, theta_out, space, stream);
}

// Per-frame outcome of the solves so far: flags[f] bit 0 = the last solve of frame f failed ("LLT has numerical issue!",
// node/node.cpp:934-937: the update of that frame was skipped), bit 1 = some solve since the last set_config /
// solve_sequence start failed, bit 2 = an evaluation since the tasks were last set (smplpp_ik_set_tasks clears it; so do set_config
// and solve_sequence) met a task with a normal term on a vertex of more than 12 adjacent faces: its Jacobian rows are not supported,
// and the solve skips that frame's update (bit 0 then reads 1 as for any skipped update).  SMPLPP_HOST calls of iterate / solve_sequence report the same condition as an error; a
// SMPLPP_DEVICE (enqueue-only) caller reads it here once its stream has reached the point of interest.
extern "C" int smplpp_ik_get_status(smplpp_ik * s, int32_t * flags, int space, void * stream)
{
  if(!s || !flags) return fail(SMPLPP_ERR_INVALID, "smplpp_ik_get_status: bad argument");
  int rc = check_space(space, "smplpp_ik_get_status");
  if(rc) return rc;
  HIP_TRY(hipSetDevice(s->m->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  std::vector<int> a((size_t)s->n), b((size_t)s->n);
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipMemcpy(a.data(), s->status, sizeof(int) * s->n, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(b.data(), s->sticky, sizeof(int) * s->n, hipMemcpyDeviceToHost));
  // bit 3: a forward pass INSIDE this solver's loops met an operand outside the fp16x2 form's range since the last set_config
  // (one word per solver — which frame is not recorded, so every frame of the batch carries it; such a frame's vertices are not
  // finite and its solve then fails on its own)
  int internal = 0;
  if(s->range_word && s->m->form_ik == 'h') HIP_TRY(hipMemcpy(&internal, s->range_word, sizeof(int), hipMemcpyDeviceToHost));
  std::vector<int32_t> h((size_t)s->n);
  for(int64_t f = 0; f < s->n; f++)
    h[(size_t)f] = (a[(size_t)f] == 1 ? 1 : 0) | ((b[(size_t)f] & 1) ? 2 : 0) | (b[(size_t)f] & 4) | ((internal & 1) ? 8 : 0);
  if(space == SMPLPP_HOST)
    memcpy(flags, h.data(), sizeof(int32_t) * (size_t)s->n);
  else
    HIP_TRY(hipMemcpy(flags, h.data(), sizeof(int32_t) * (size_t)s->n, hipMemcpyHostToDevice));
  return SMPLPP_OK;
}

extern "C" int smplpp_ik_get_vertices(smplpp_ik * s, float * verts, int space, void * stream)
{
  if(!s || !verts) return fail(SMPLPP_ERR_INVALID, "smplpp_ik_get_vertices: bad argument");
  if(!s->have_eval) return fail(SMPLPP_ERR_STATE, "Failed to get vertices of new pose!");
  int rc = check_space(space, "smplpp_ik_get_vertices");
  if(rc) return rc;
  HIP_TRY(hipSetDevice(s->m->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipMemcpyKind kind = space == SMPLPP_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  HIP_TRY(hipMemcpyAsync(verts, s->verts, sizeof(float) * s->n * s->m->V * 3, kind, st));
  if(space == SMPLPP_HOST) HIP_TRY(hipStreamSynchronize(st));
  return SMPLPP_OK;
}
