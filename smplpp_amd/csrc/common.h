// Internal definitions shared by the translation units of libsmplpp_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/smplpp_hip.h"

namespace smplpp_hip
{
constexpr int NJ = SMPLPP_JOINT_NUM;
constexpr int NB = SMPLPP_SHAPE_BASIS_DIM;
constexpr int NP = SMPLPP_POSE_BASIS_DIM;

// K dimension of the fused blend-shape GEMM: [pose coefficients 207 | beta 10 | 1 (template) | 0 0]
constexpr int KP = 220;
constexpr int K_BETA = NP;       // 207
constexpr int K_ONE = NP + NB;   // 217
// Column layout of the B operand: vertex group g = v / 32 owns columns [96 g, 96 g + 96): 32 x, then 32 y, 32 z.
constexpr int VG = 32;
__host__ __device__ inline int64_t bcol(int64_t v, int x)
{
  return (v / VG) * (3 * VG) + x * VG + (v % VG);
}

void set_error(const std::string & msg);
int fail(int code, const std::string & msg);
int hip_fail(hipError_t e, const char * what, const char * file, int line);

#define HIP_TRY(expr)                                                              \
  do                                                                               \
  {                                                                                \
    hipError_t _e = (expr);                                                        \
    if(_e != hipSuccess) return smplpp_hip::hip_fail(_e, #expr, __FILE__, __LINE__); \
  } while(0)

// Growable device buffer
struct DevBuf
{
  void * p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes)
  {
    if(bytes <= cap) return hipSuccess;
    if(p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 4;
    hipError_t e = hipMalloc(&p, want);
    if(e == hipSuccess) cap = want;
    return e;
  }
  void release()
  {
    if(p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template<class T>
  T * as() const
  {
    return static_cast<T *>(p);
  }
};

struct Workspace
{
  DevBuf AT;      // [KP][ldA] fp32, K-major A operand (pose coefficients | beta | 1)
  DevBuf Gp;      // [n][24][12] relative transforms, 3x4 row-major
  DevBuf joints;  // [n][24][3]
  DevBuf poserot; // [n][24][9]
  DevBuf beta, theta, verts, rest, xf44; // staging for host-pointer calls
  int64_t ldA = 0;
  // work queues of the persistent fused kernel (skin_q.hip)
  DevBuf q_ctr, q_desc, dummy; // dummy: write-only sink for masked-off lanes of branch-free epilogues
  int64_t q_n = -1;
  int q_grid = 0;
};
} // namespace smplpp_hip

struct smplpp_model
{
  int device = 0;
  int64_t V = 0, F = 0;
  int64_t VGn = 0;  // vertex groups of 32
  int64_t ldB = 0;  // VGn * 96
  int maxw = 0;     // skinning weights kept per vertex: 4, 8 or 24
  // device arrays
  float * Bm = nullptr;        // [KP][ldB]
  uint8_t * wIdx = nullptr;    // [VGn*32][maxw]
  float * wVal = nullptr;      // [VGn*32][maxw]
  float * wSum = nullptr;      // [VGn*32]  sum_j W[v,j] in ascending j (the blended homogeneous w)
  float * J0 = nullptr;        // [24][3]      Jreg . T
  float * JS = nullptr;        // [24][3][10]  Jreg . S
  int32_t * parent = nullptr;  // [24]
  int32_t * faces = nullptr;   // [F][3] 0-based
  int32_t * adjOff = nullptr;  // [V+1]
  int32_t * adjFace = nullptr; // [adjOff[V]] ascending face id per vertex
  float * Wdense = nullptr;    // [V][24] original weights (stage entry points / IK)
  float * Pvm = nullptr;       // [V][3][207] posedirs, vertex-major (IK Jacobian: pose-corrective term of a few vertices)
  float * Svm = nullptr;       // [V][3][10]  shapedirs, vertex-major (IK Jacobian: beta columns)
  // host mirrors
  std::vector<int32_t> h_parent, h_faces, h_adjOff, h_adjFace;
  // measurement hook (smplpp_profile_*)
  bool profiling = false;
  std::vector<hipEvent_t> prof_events; // begin/end pairs around the fused kernel
  smplpp_hip::Workspace ws;
};
