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
constexpr int CT_LEV = 12;       // tree levels the pose kernel's register-resident chain table covers
constexpr int CT_OFF = 52;       // smplpp_model::lvl: [25 level offsets | 24 joints by level | pad to 16 bytes | chain table 60 x CT_LEV x 2]
// pose_kernel's operand array sP: rotations [24][9] | joints [24][3] | zero [4]; the chain table holds, per chain lane (60 =
// 5 slots x 12 entries of a 3x4) and level, word 0 = joint | parent << 8 | parent's slot << 16 (0xff = none) and word 1 =
// index of the lane's operand in sP | index of what is subtracted from it << 10 | stride << 20
constexpr int CT_P_R = 0, CT_P_J = SMPLPP_JOINT_NUM * 9, CT_P_ZERO = CT_P_J + SMPLPP_JOINT_NUM * 3, CT_P_SIZE = CT_P_ZERO + 4;
// tree tables of a model for the IK evaluation (smplpp_model::anc, int32): ancestor bit masks [24] | level offsets
// [TREE_DMAX + 1] | joints sorted by level [24]
constexpr int TREE_DMAX = 12;
constexpr int TREE_ANC = 0, TREE_LVL = SMPLPP_JOINT_NUM, TREE_LVLJ = TREE_LVL + TREE_DMAX + 1, TREE_SIZE = TREE_LVLJ + SMPLPP_JOINT_NUM;
// slots of smplpp_model::range_flag: enqueue-only user launches (read by smplpp_fk_status), host-space user launches (each reads
// its own), launches from inside the IK / VPoser loops (intermediate iterates; the solve's own status reports what matters there)
constexpr int RANGE_DEVICE = 0, RANGE_HOST = 1, RANGE_INTERNAL = 2, RANGE_SLOTS = 4;
constexpr int MAXADJ = 12;                    // adjacent faces per vertex the IK normal Jacobian's tables hold by default (SMPL's mesh: at most 9)
constexpr int MAXADJ_WIDE = 16;               // ... for a topology with a vertex of 13..16 faces (smplpp_model::madj; its own instantiation of the evaluation)
constexpr int MAXRING = 3 * (MAXADJ + 1) + 1; // distinct vertices an IK task can touch: its face's and those of the faces around them
// Column layout of the B operand: vertex group g = v / 32 owns columns [96 g, 96 g + 96): 32 x, then 32 y, 32 z.
constexpr int VG = 32;
__host__ __device__ inline int64_t bcol(int64_t v, int x)
{
  return (v / VG) * (3 * VG) + x * VG + (v % VG);
}

// ---- bf16x3 form of the fused kernel (skin_b.hip): operands in MFMA fragment order, three bf16 pieces per fp32 value.
// A "piece" = the 1 KiB one wavefront feeds to one v_mfma_f32_32x32x16_bf16: lane l = 32 h + r holds k = 16 ks + 8 h + j,
// j = 0..7, of row (frame) / column (vertex coordinate) r.
//   A3 [ceil(n/64)][BB_KS][fh 2][piece s 3][64 lanes][8 bf16]           frame = 64 ftp + 32 fh + r
//   B3 [ceil(V/64)][BB_KS][vh 2][coordinate x 3][piece s 3][64][8]      vertex = 64 vgp + 32 vh + r
constexpr int BB_KS = 14;                 // k-steps of 16 (K = 220 padded to 224)
constexpr int BB_A_BYTES = 6 * 1024;      // A pieces of one (frame-tile pair, k-step)
constexpr int BB_B_BYTES = 18 * 1024;     // B pieces of one (vertex-group pair, k-step)
constexpr int BB_KSTEP_BYTES = BB_A_BYTES + BB_B_BYTES;
// x = p0 + p1 + p2 exactly (round-to-nearest-even pieces; finite inputs)
__host__ __device__ inline uint16_t bf16_rn_bits(float x)
{
  uint32_t u;
  memcpy(&u, &x, 4);
  return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__host__ __device__ inline float bf16_bits_to_float(uint16_t b)
{
  uint32_t u = (uint32_t)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
__host__ __device__ inline void split_bf16x3(float x, uint16_t & p0, uint16_t & p1, uint16_t & p2)
{
  p0 = bf16_rn_bits(x);
  const float r1 = x - bf16_bits_to_float(p0);
  p1 = bf16_rn_bits(r1);
  const float r2 = r1 - bf16_bits_to_float(p1);
  p2 = bf16_rn_bits(r2);
}

// ---- bf16x3 form, round 6 (skin_e.hip, "e" = exact: the form smplpp_fk runs by default).  Same pieces and fragment order as B3 /
// A3 above; what changes is who holds what: the A fragments of a frame tile live in registers, the relative transforms of the tile
// stay in LDS for a whole run of vertex groups, and only the basis streams — one 20 KiB image per k-step through a ring of four:
//   B3e [ceil(V/64)][EB_KS][20 KiB]: pieces 0..17 = [vh 2][coordinate x 3][piece s 3][64 lanes][8 bf16] (as a k-step of B3), then
//        2 KiB that make the image five 1 KiB DMA pieces per wavefront and carry the group's skinning tables:
//        k-step 0: jofs[64][4] int32 (byte offset of the joint's 3x4 inside a frame's G' record) | jw[64][4] fp32
//        k-step 1: winv[64] fp32 = 1 / sum_j W[v, j]            k-step 2 (models with 5..8 weights): jofs[64][4..7] | jw[64][4..7]
constexpr int EB_KS = 14;
constexpr int EB_IMG = 20 * 1024;
constexpr int EB_TAB_OFF = 18 * 1024;

// ---- fp16x2 form of the fused kernel (skin_h.hip; the IK loops' internal launches): every fp32 operand value x is carried as TWO fp16
// pieces of s.x (s a power of two chosen per operand so that the pieces stay in fp16's normal range):
// hi = fp16(s x), lo = fp16(s x - hi), |s x - hi - lo| <= 2^-22 |s x|; a product is the three MFMAs
// lo.hi + hi.lo + hi.hi (the dropped lo.lo term is < 2^-22 |a||b|).  All arrays are in MFMA fragment order for
// v_mfma_f32_32x32x16_f16: a "piece" is 1 KiB, lane l = 32 h + r holds k = 16 ks + 8 h + j (j = 0..7) of row/column r.
//   A2h [ceil(n/64)][HB_KS][fh 2][piece 2][64 lanes][8 fp16]                 frame = 64 ft + 32 fh + r; value sA.a
//   B2h [ceil(V/64)][HB_SLOTS][12 KiB]: slots 0..13 = k-steps: [vh 2][coordinate 3][piece 2][64][8]  (value sB.b)
//                                       slot 14 = skinning weights of the group: [ks 2][vh 2][piece 2][64][8] fp16 of
//                                       sW.W[v][joint k] (k >= 24: 0; k-step 1, eight live k: piece 1 = [hi | lo] over the
//                                       lane halves), then cw[64] fp32 = 1 / (sG sW sum_j W[v,j]), then padding
//   G2h [ceil(n/64)][fh 2][entry 12][3 KiB]: relative transforms as the A operand of the blend MFMAs (rows = frames,
//                                       k = joint): ks 0: [piece 2][64][8]; ks 1 (joints 16..23): [piece 2][32 lanes][8]
constexpr int HB_KS = 14;
constexpr int HB_SLOTS = 15;
constexpr int HB_A_BYTES = 4 * 1024;   // A pieces of one (frame tile, k-step)
constexpr int HB_IMG = 12 * 1024;      // one slot of B2h = one LDS ring image
constexpr int HB_G_BYTES = 72 * 1024;  // G2h of one frame tile
constexpr int HB_CW_OFF = 8 * 1024;    // cw[64] inside slot 14
// Round 5: a vertex group is 64 CONSECUTIVE vertices, classified by which of the skinning product's two k-steps its vertices'
// weights touch (joints 0..15 | joints 16..23: for SMPL the arms).  Model creation deals the groups over the eight XCD slices with the
// classes interleaved, and interleaves them again inside a slice; slot 14 carries, behind cw, the group's vertex ids (perm[64], -1 =
// no vertex: the outputs go to the ORIGINAL positions) and one word of flags — bit 0: some vertex of the group has a weight on joints
// 0..15, bit 1: on joints 16..23.  A group with one bit skips the other k-step's blend MFMAs and G' fragment reads (3 or 2 MFMAs per
// entry instead of 5); the products it skips are exact zeros, so no bit of the result moves.  (Sorting the VERTICES by class first
// was measured and rejected: 46 -> 62 us per step, the output rows of a group scattered over ~200 vertex positions.)
constexpr int HB_PERM_OFF = HB_CW_OFF + 256;   // perm[64] int32
constexpr int HB_FLAGS_OFF = HB_CW_OFF + 512;  // one int32
constexpr float HB_SA = 64.0f;         // scale of the A operand (|c| <= 2, |beta| < 1023)
constexpr float HB_SW = 16384.0f;      // scale of the skinning weights (|W| <= 1)
__host__ __device__ inline void split_f16x2(float xs, _Float16 & hi, _Float16 & lo) // xs: already scaled
{
  hi = (_Float16)xs;
  lo = (_Float16)(xs - (float)hi);
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

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is an opt-in per DEVICE: one flag per (call site, device)
struct PerDeviceOnce
{
  bool done[64] = {};
};
inline hipError_t lds_opt_in(PerDeviceOnce & o, int device, const void * fn, int bytes)
{
  if(o.done[device & 63]) return hipSuccess;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if(e == hipSuccess) o.done[device & 63] = true;
  else if(getenv("SMPLPP_DEBUG_LDS"))
  {
    hipFuncAttributes a;
    hipError_t e2 = hipFuncGetAttributes(&a, fn);
    fprintf(stderr, "[smplpp dbg] lds_opt_in(%d bytes) failed: %s; attributes (%s): static %zu, max dynamic %d, regs %d, max threads %d\n", bytes,
            hipGetErrorString(e), hipGetErrorString(e2), a.sharedSizeBytes, a.maxDynamicSharedSizeBytes, a.numRegs, a.maxThreadsPerBlock);
  }
  return e;
}
inline int device_cus(int device) // compute units of a device (cached per device)
{
  static int cus[64] = {};
  int & c = cus[device & 63];
  if(!c)
  {
    hipDeviceProp_t prop;
    c = (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  return c;
}

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
  DevBuf A3;      // the same coefficients as bf16x3 pieces in fragment order (skin_b.hip)
  DevBuf A2h;     // the same coefficients as fp16x2 pieces in fragment order (skin_h.hip)
  DevBuf G2h;     // relative transforms as fp16x2 pieces, the A operand of the blend MFMAs (skin_h.hip)
  DevBuf Gp;      // [n][24][12] relative transforms, 3x4 row-major
  DevBuf joints;  // [n][24][3]
  DevBuf poserot; // [n][24][9]
  DevBuf beta, theta, verts, rest, xf44; // staging for host-pointer calls
  int64_t ldA = 0;
  DevBuf dummy;   // write-only sink for masked-off lanes of branch-free epilogues (skin_p.hip)
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
  uint8_t * B3 = nullptr;      // Bm as bf16x3 pieces in MFMA fragment order (layout above)
  int64_t VGPn = 0;            // vertex-group pairs: ceil(V / 64)
  uint8_t * B2h = nullptr;     // bases + skinning weights as fp16x2 pieces in MFMA fragment order (layout above)
  float sB = 1.0f, sG = 1.0f;  // power-of-two scales of the basis operand and of the relative transforms (fp16 range)
  int * range_flag = nullptr;  // device words [RANGE_SLOTS]: bit 0 = a launch of the fp16x2 form met an operand outside fp16's range
  uint8_t * B3e = nullptr;     // bases + skinning tables of the exact form, one 20 KiB image per (vertex group, k-step) (layout above, EB_*)
  char form = 'e';             // fused-kernel form of smplpp_fk (SMPLPP_SKIN, read once at model creation): e | h | b | p | v
  char form_ik = 'h';          // ... of the IK / VPoser loops' internal launches (h unless SMPLPP_SKIN chose one form for everything)
  uint8_t * wIdx = nullptr;    // [VGn*32][maxw]
  float * wVal = nullptr;      // [VGn*32][maxw]
  float * wSum = nullptr;      // [VGn*32]  sum_j W[v,j] in ascending j (the blended homogeneous w)
  float * J0 = nullptr;        // [24][3]      Jreg . T
  float * JS = nullptr;        // [24][3][10]  Jreg . S
  float * JSp = nullptr;       // [72][12] the two once more, a 48-byte row per joint coordinate: [JS row (10) | J0 | 0] (pose_kernel: three 16-byte loads)
  int32_t * parent = nullptr;  // [24]
  int32_t * lvl = nullptr;     // [25 + 24] kinematic tree by depth: level offsets, then the joints sorted by level
  int nlev = 0;
  bool chain_fast = false;     // lvl also holds the pose kernel's chain table: per chain lane (60) and level (CT_LEV) joint | parent << 8 | parent's slot << 16 (0xff = none)
  int32_t * faces = nullptr;   // [F][3] 0-based
  int32_t * adjOff = nullptr;  // [V+1]
  int32_t * adjFace = nullptr; // [adjOff[V]] ascending face id per vertex
  int madj = smplpp_hip::MAXADJ;  // width of the IK ring tables below: MAXADJ, or MAXADJ_WIDE when some vertex has more than MAXADJ adjacent faces
  uint16_t * faceRing = nullptr; // [F][3 (madj + 1) + 2] IK ring of a task on face f: count, the face's three vertices, then the distinct
                                 // vertices of the faces adjacent to them in (vertex, adjacent face, corner) order (V <= 65535)
  uint8_t * faceMap = nullptr;   // [F][3 madj 3] (vertex of the face, adjacent face, corner) -> slot in that ring
  int32_t * anc = nullptr;       // [TREE_SIZE] tree tables of the IK evaluation (TREE_* above)
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
