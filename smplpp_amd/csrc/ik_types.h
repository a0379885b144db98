// Shared by the IK kernels (ik_eval.h, ik_solve.h, ik_proj.h) and their host side (ik.hip): sizes, the task arrays and the model view
// the kernels take by value, the development stamps.
#pragma once
#include "mesh_device.h"
#include "staging.h"
#include "trace.h"
#include "signal.h"

#include <hip/hip_ext.h>

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
} // namespace smplpp_hip
