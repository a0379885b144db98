// Model creation: replaces SMPL::init (/root/reference/src/SMPL.cpp:560-643) minus the JSON parse.
//
// HBM layout produced here (all fp32 unless noted, resident for the life of the handle; ~37 MB for SMPL):
//   Bm   [220][ldB]   B operand of the fused blend-shape GEMM, K-major: rows 0..206 posedirs, 207..216 shapedirs,
//                     217 template, 218..219 zero.  Columns are grouped per 32 vertices as [32 x | 32 y | 32 z] so
//                     that one 32x32 MFMA tile is one coordinate of 32 consecutive vertices (ldB = ceil(V/32)*96).
//   wIdx/wVal [Vpad][maxw]  skinning weights, the maxw (4, 8 or 24) non-zeros per vertex in ascending joint order.
//   wSum [Vpad]       sum_j W[v,j]: the blended homogeneous coordinate the reference divides by
//                     (src/LinearBlendSkinning.cpp:545-550).
//   J0 [24][3], JS [24][3][10]  joint regressor folded through template and shapedirs:
//                     joints = J0 + JS . beta  ==  Jreg . (T + S . beta)   (src/JointRegression.cpp:588-590)
//   faces, adjOff/adjFace      0-based faces and the per-vertex adjacent-face table (src/SMPL.cpp:620-640).
//   Pvm [V][3][207], Svm [V][3][10]  the bases once more in the file's vertex-major order, for the IK Jacobian of a
//                     handful of task vertices (a K-major gather would touch 207 cache lines per vertex).
#include "common.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace smplpp_hip
{
typedef _Float16 f16x8h __attribute__((ext_vector_type(8)));
static thread_local std::string g_last_error;

void set_error(const std::string & msg)
{
  g_last_error = msg;
}

int fail(int code, const std::string & msg)
{
  g_last_error = msg;
  return code;
}

int hip_fail(hipError_t e, const char * what, const char * file, int line)
{
  char buf[512];
  snprintf(buf, sizeof(buf), "HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
  g_last_error = buf;
  return SMPLPP_ERR_HIP;
}

// Bm[k][bcol(v,x)] <- P[v][x][k] (k < 207) | S[v][x][k-207] | T[v][x] | 0
__global__ void relayout_basis_kernel(const float * __restrict__ P, const float * __restrict__ S, const float * __restrict__ T,
                                      float * __restrict__ Bm, int64_t V, int64_t ldB)
{
  int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(col >= ldB) return;
  int64_t g = col / (3 * VG);
  int x = (int)((col % (3 * VG)) / VG);
  int64_t v = g * VG + col % VG;
  for(int k = 0; k < KP; k++)
  {
    float val = 0.0f;
    if(v < V)
    {
      if(k < NP)
        val = P[(v * 3 + x) * NP + k];
      else if(k < NP + NB)
        val = S[(v * 3 + x) * NB + (k - NP)];
      else if(k == K_ONE)
        val = T[v * 3 + x];
    }
    Bm[(int64_t)k * ldB + col] = val;
  }
}

// B3 <- Bm as bf16x3 pieces in MFMA fragment order (layout: common.h).  One thread per (vertex-group pair, k-step,
// piece, lane): 8 consecutive k of one column.
__global__ void relayout_basis_bf16x3_kernel(const float * __restrict__ Bm, int64_t ldB, int64_t V, int64_t nvgp,
                                             uint16_t * __restrict__ B3)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; // ((vgp * KS + ks) * 6 + vh * 3 + x) * 64 + lane
  if(i >= nvgp * BB_KS * 6 * 64) return;
  const int lane = (int)(i % 64), h = lane >> 5, r = lane & 31;
  const int vx = (int)((i / 64) % 6), vh = vx / 3, x = vx % 3;
  const int ks = (int)((i / (64 * 6)) % BB_KS);
  const int64_t vgp = i / (64 * 6 * BB_KS);
  const int64_t v = vgp * 64 + vh * 32 + r;
  uint16_t pc[3][8];
  for(int j = 0; j < 8; j++)
  {
    const int k = ks * 16 + 8 * h + j;
    const float val = (v < V && k < KP) ? Bm[(int64_t)k * ldB + bcol(v, x)] : 0.0f;
    split_bf16x3(val, pc[0][j], pc[1][j], pc[2][j]);
  }
  for(int s = 0; s < 3; s++)
  {
    uint16_t * dst = B3 + ((((vgp * BB_KS + ks) * 6 + vx) * 3 + s) * 64 + lane) * 8;
    for(int j = 0; j < 8; j++) dst[j] = pc[s][j];
  }
}

// B3e <- Bm as bf16x3 pieces in MFMA fragment order, one 20 KiB image per (vertex group, k-step) (layout: common.h, EB_*).  One
// thread per (vertex group, k-step, vertex half and coordinate, lane): 8 consecutive k of one column, its three pieces.
__global__ void relayout_basis_exact_kernel(const float * __restrict__ Bm, int64_t ldB, int64_t V, int64_t nvg, uint8_t * __restrict__ B3e)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; // ((vg * KS + ks) * 6 + vh * 3 + x) * 64 + lane
  if(i >= nvg * EB_KS * 6 * 64) return;
  const int lane = (int)(i % 64), h = lane >> 5, r = lane & 31;
  const int vx = (int)((i / 64) % 6), vh = vx / 3, x = vx % 3;
  const int ks = (int)((i / (64 * 6)) % EB_KS);
  const int64_t vg = i / (64 * 6 * EB_KS);
  const int64_t v = vg * 64 + vh * 32 + r;
  uint16_t pc[3][8];
  for(int j = 0; j < 8; j++)
  {
    const int k = ks * 16 + 8 * h + j;
    const float val = (v < V && k < KP) ? Bm[(int64_t)k * ldB + bcol(v, x)] : 0.0f;
    split_bf16x3(val, pc[0][j], pc[1][j], pc[2][j]);
  }
  for(int s = 0; s < 3; s++)
  {
    uint16_t * dst = reinterpret_cast<uint16_t *>(B3e + (vg * EB_KS + ks) * (int64_t)EB_IMG) + ((int64_t)(vx * 3 + s) * 64 + lane) * 8;
    for(int j = 0; j < 8; j++) dst[j] = pc[s][j];
  }
}
// the skinning tables of a vertex group, in the 2 KiB behind the fragments of its first k-steps (one thread per vertex slot)
__global__ void skin_tables_exact_kernel(const uint8_t * __restrict__ wIdx, const float * __restrict__ wVal, const float * __restrict__ wSum,
                                         int maxw, int64_t V, int64_t nvg, uint8_t * __restrict__ B3e)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= nvg * 64) return;
  const int64_t vg = i / 64, v = i;
  const int c = (int)(i % 64);
  uint8_t * g = B3e + vg * (int64_t)(EB_KS * EB_IMG);
  for(int q = 0; q < maxw && q < 8; q++)
  {
    uint8_t * tab = g + (q < 4 ? 0 : 2) * EB_IMG + EB_TAB_OFF;
    reinterpret_cast<int32_t *>(tab)[c * 4 + (q & 3)] = v < V ? (int32_t)wIdx[v * maxw + q] * 48 : 0;
    reinterpret_cast<float *>(tab + 1024)[c * 4 + (q & 3)] = v < V ? wVal[v * maxw + q] : 0.0f;
  }
  reinterpret_cast<float *>(g + EB_IMG + EB_TAB_OFF)[c] = v < V ? 1.0f / wSum[v] : 0.0f;
}

// B2h <- Bm and the skinning weights as fp16x2 pieces in MFMA fragment order (layout: common.h).  One thread per 16-byte
// chunk pair (hi, lo): slots 0..13: ((vg * 15 + ks) * 6 + vh * 3 + x) * 64 + lane; slot 14: weights, cw, padding.
// hperm [nvg * 64]: the vertex in each slot of each group (-1: none), gflags [nvg]: the group's k-step flags (common.h, HB_PERM_OFF).
__global__ void relayout_basis_f16x2_kernel(const float * __restrict__ Bm, int64_t ldB, const float * __restrict__ W,
                                            const float * __restrict__ wSum, int64_t V, int64_t nvg, float sB, float sG,
                                            uint8_t * __restrict__ B2h, const int32_t * __restrict__ hperm,
                                            const int32_t * __restrict__ gflags)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int per_vg = HB_KS * 6 * 64 + 4 * 64 + 64; // basis chunk pairs + weight chunk pairs + cw entries
  if(i >= nvg * per_vg) return;
  const int64_t vg = i / per_vg;
  const int q = (int)(i % per_vg);
  uint8_t * base = B2h + vg * (int64_t)(HB_SLOTS * HB_IMG);
  f16x8h hi, lo;
  if(q < HB_KS * 6 * 64)
  {
    const int lane = q % 64, h = lane >> 5, r = lane & 31;
    const int vx = (q / 64) % 6, vh = vx / 3, x = vx % 3, ks = q / (64 * 6);
    const int64_t v = hperm[vg * 64 + vh * 32 + r];
    for(int j = 0; j < 8; j++)
    {
      const int k = ks * 16 + 8 * h + j;
      const float val = (v >= 0 && k < KP) ? Bm[(int64_t)k * ldB + bcol(v, x)] : 0.0f;
      _Float16 a, b;
      split_f16x2(val * sB, a, b);
      hi[j] = a;
      lo[j] = b;
    }
    uint8_t * dst = base + ks * HB_IMG + ((vh * 3 + x) * 2) * 1024 + lane * 16;
    *reinterpret_cast<f16x8h *>(dst) = hi;
    *reinterpret_cast<f16x8h *>(dst + 1024) = lo;
  }
  else if(q < HB_KS * 6 * 64 + 4 * 64)
  {
    const int w = q - HB_KS * 6 * 64, lane = w % 64, h = lane >> 5, r = lane & 31, vh = (w / 64) % 2, ks = w / 128;
    const int64_t v = hperm[vg * 64 + vh * 32 + r];
    for(int j = 0; j < 8; j++)
    {
      const int k = ks * 16 + 8 * h + j; // joint
      const float val = (v >= 0 && k < NJ) ? W[v * NJ + k] : 0.0f;
      _Float16 a, b;
      split_f16x2(val * HB_SW, a, b);
      hi[j] = a;
      lo[j] = b;
    }
    uint8_t * dst = base + HB_KS * HB_IMG + ((ks * 2 + vh) * 2) * 1024 + lane * 16;
    *reinterpret_cast<f16x8h *>(dst) = hi;
    if(ks == 0)
      *reinterpret_cast<f16x8h *>(dst + 1024) = lo;
    else if(h == 0)
    {
      // k-step 1 has eight live k (joints 16..23): its second fragment is [hi | lo] over the two lane halves, so that one
      // MFMA against the G' hi piece (read by both halves) is Ghi.Whi + Ghi.Wlo (skin_h.hip)
      *reinterpret_cast<f16x8h *>(dst + 1024) = hi;
      *reinterpret_cast<f16x8h *>(dst + 1024 + 512) = lo;
    }
  }
  else
  {
    // cw = 1 / (sG sW h[3]) with h[3] = sum_j W[v,j], the homogeneous coordinate the reference divides by
    // (src/LinearBlendSkinning.cpp:545-550): one reciprocal per vertex (<= 1 ulp from the division)
    const int c = q - (HB_KS * 6 * 64 + 4 * 64);
    const int64_t v = hperm[vg * 64 + c];
    uint8_t * s14 = base + HB_KS * HB_IMG;
    float * cw = reinterpret_cast<float *>(s14 + HB_CW_OFF);
    cw[c] = v >= 0 ? (1.0f / wSum[v]) / (sG * HB_SW) : 0.0f;
    reinterpret_cast<int32_t *>(s14 + HB_PERM_OFF)[c] = (int32_t)v; // where this slot's vertex goes in the outputs
    // the flags word, then zeros up to the end of the slot (its DMA copies whole 12 KiB images)
    int32_t * tail = reinterpret_cast<int32_t *>(s14 + HB_FLAGS_OFF);
    for(int i = c; i < (HB_IMG - HB_FLAGS_OFF) / 4; i += 64) tail[i] = (i == 0) ? gflags[vg] : 0;
  }
}

// One block per (joint j, coordinate x, term t): t < 10 -> JS[j][x][t] = sum_v Jreg[j,v] S[v,x,t];
// t == 10 -> J0[j][x] = sum_v Jreg[j,v] T[v,x].  Wavefront (64-lane) shuffle reduction, then across the 4 waves.
__global__ __launch_bounds__(256) void fold_regressor_kernel(const float * __restrict__ Jreg, const float * __restrict__ S,
                                                              const float * __restrict__ T, float * __restrict__ J0,
                                                              float * __restrict__ JS, int64_t V)
{
  const int t = blockIdx.x % (NB + 1);
  const int x = (blockIdx.x / (NB + 1)) % 3;
  const int j = blockIdx.x / (3 * (NB + 1));
  double acc = 0.0; // fp64 partials: the reference's GEMM summation order over 6890 terms is BLAS-defined
  for(int64_t v = threadIdx.x; v < V; v += blockDim.x)
  {
    float w = Jreg[(int64_t)j * V + v];
    float b = (t < NB) ? S[(v * 3 + x) * NB + t] : T[v * 3 + x];
    acc += (double)w * (double)b;
  }
  for(int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  __shared__ double part[4];
  if((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if(threadIdx.x == 0)
  {
    double s = (part[0] + part[1]) + (part[2] + part[3]);
    if(t < NB)
      JS[(j * 3 + x) * NB + t] = (float)s;
    else
      J0[j * 3 + x] = (float)s;
  }
}

// [72][12]: row t = [JS[t][0..9] | J0[t] | 0]
__global__ void pack_regressor_rows_kernel(const float * __restrict__ J0, const float * __restrict__ JS, float * __restrict__ JSp)
{
  const int t = threadIdx.x;
  for(int k = 0; k < NB; k++) JSp[t * 12 + k] = JS[t * NB + k];
  JSp[t * 12 + 10] = J0[t];
  JSp[t * 12 + 11] = 0.0f;
}

template<class T>
static hipError_t upload(T ** dst, const T * src, size_t count)
{
  hipError_t e = hipMalloc((void **)dst, sizeof(T) * std::max<size_t>(count, 1));
  if(e != hipSuccess) return e;
  if(count) e = hipMemcpy(*dst, src, sizeof(T) * count, hipMemcpyHostToDevice);
  return e;
}
} // namespace smplpp_hip

using namespace smplpp_hip;

extern "C" const char * smplpp_last_error(void)
{
  return g_last_error.c_str();
}

extern "C" int smplpp_device_count(int * count)
{
  if(!count) return fail(SMPLPP_ERR_INVALID, "smplpp_device_count: null argument");
  *count = 0;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if(e != hipSuccess || n <= 0)
  {
    (void)hipGetLastError();
    return fail(SMPLPP_ERR_HIP, "no HIP device visible (libsmplpp_hip.so is MI355X-only and has no CPU path)");
  }
  *count = n;
  return SMPLPP_OK;
}

extern "C" int smplpp_model_destroy(smplpp_model * m)
{
  if(!m) return SMPLPP_OK;
  (void)hipSetDevice(m->device);
  for(hipEvent_t e : m->prof_events) (void)hipEventDestroy(e);
  m->prof_events.clear();
  void * ptrs[] = {m->Bm, m->B3, m->B3e, m->B2h, m->range_flag, m->lvl, m->wIdx, m->wVal, m->wSum, m->J0, m->JS, m->JSp, m->parent, m->faces, m->adjOff, m->adjFace, m->Wdense, m->Pvm, m->Svm, m->faceRing, m->faceMap, m->anc};
  for(void * p : ptrs)
    if(p) (void)hipFree(p);
  Workspace & w = m->ws;
  for(DevBuf * b : {&w.AT, &w.A3, &w.A2h, &w.G2h, &w.Gp, &w.joints, &w.poserot, &w.beta, &w.theta, &w.verts, &w.rest, &w.xf44, &w.dummy}) b->release();
  delete m;
  return SMPLPP_OK;
}

extern "C" int smplpp_model_create(int64_t V, int64_t F, const float * vt, const float * S, const float * P,
                                   const float * Jreg, const float * W, const int64_t * kintree, const int32_t * faces1,
                                   int device, smplpp_model ** out)
{
  if(!out) return fail(SMPLPP_ERR_INVALID, "smplpp_model_create: null output");
  *out = nullptr;
  if(V <= 0 || F < 0 || !vt || !S || !P || !Jreg || !W || !kintree || (F > 0 && !faces1))
    return fail(SMPLPP_ERR_INVALID, "Cannot initialize a SMPL model!"); // src/SMPL.cpp:616
  int ndev = 0;
  int rc = smplpp_device_count(&ndev);
  if(rc) return rc;
  if(device < 0 || device >= ndev) return fail(SMPLPP_ERR_INVALID, "Failed to fetch device index!"); // src/SMPL.cpp:295
  // kinematic tree: row 0 = parent (src/WorldTransformation.cpp:523); must be topologically ordered like SMPL's
  std::vector<int32_t> parent(NJ);
  parent[0] = -1;
  for(int i = 1; i < NJ; i++)
  {
    if(kintree[i] < 0 || kintree[i] >= i) return fail(SMPLPP_ERR_INVALID, "Cannot set kinematic tree: parent(i) must precede i");
    parent[i] = (int32_t)kintree[i];
  }
  for(int64_t i = 0; i < F * 3; i++)
    if(faces1[i] < 1 || faces1[i] > V) return fail(SMPLPP_ERR_INVALID, "face_indices must be 1-based vertex ids");

  HIP_TRY(hipSetDevice(device));
  smplpp_model * m = new smplpp_model();
  m->device = device;
  m->V = V;
  m->F = F;
  m->VGn = (V + VG - 1) / VG;
  m->ldB = m->VGn * 3 * VG;
  m->h_parent = parent;

#define TRY_OR_FREE(expr)                                      \
  do                                                           \
  {                                                            \
    hipError_t _e = (expr);                                    \
    if(_e != hipSuccess)                                       \
    {                                                          \
      int _rc = hip_fail(_e, #expr, __FILE__, __LINE__);       \
      smplpp_model_destroy(m);                                 \
      return _rc;                                              \
    }                                                          \
  } while(0)

  // --- blend bases -> Bm, regressor fold (device side; the raw arrays are only needed transiently) ---
  // Form of the fused kernel: read once, here.  Default: smplpp_fk runs e (skin_e.hip: fp32-exact operands, the reference's
  // arithmetic) and the IK / VPoser loops' internal launches h (skin_h.hip: fp16x2 operands, 3e-7 m); SMPLPP_SKIN = e | h | b | p | v
  // puts every launch on that form.  Only the operand layouts the chosen forms need stay resident.
  {
    const char * form_env = getenv("SMPLPP_SKIN");
    const char f = form_env ? form_env[0] : 0;
    if(f == 'e' || f == 'h' || f == 'b' || f == 'p' || f == 'v')
      m->form = m->form_ik = f;
    else
    {
      m->form = 'e';
      m->form_ik = 'h';
    }
  }
  auto uses = [&](char f) { return m->form == f || m->form_ik == f; };
  // (the vertex-major uploads are owned by the handle from the start, so a failure below frees them with it)
  TRY_OR_FREE(upload(&m->Pvm, P, (size_t)V * 3 * NP)); // kept: vertex-major copies serve the sparse IK Jacobian (contiguous 2.5 KB per vertex)
  TRY_OR_FREE(upload(&m->Svm, S, (size_t)V * 3 * NB));
  DevBuf dT, dJreg;
  auto free_tmp = [&]() {
    dT.release();
    dJreg.release();
  };
#define TRY_TMP(expr)                                          \
  do                                                           \
  {                                                            \
    hipError_t _e = (expr);                                    \
    if(_e != hipSuccess)                                       \
    {                                                          \
      int _rc = hip_fail(_e, #expr, __FILE__, __LINE__);       \
      free_tmp();                                              \
      smplpp_model_destroy(m);                                 \
      return _rc;                                              \
    }                                                          \
  } while(0)
  TRY_TMP(dT.reserve(sizeof(float) * (size_t)V * 3));
  TRY_TMP(dJreg.reserve(sizeof(float) * (size_t)NJ * V));
  TRY_TMP(hipMemcpy(dT.p, vt, sizeof(float) * (size_t)V * 3, hipMemcpyHostToDevice));
  TRY_TMP(hipMemcpy(dJreg.p, Jreg, sizeof(float) * (size_t)NJ * V, hipMemcpyHostToDevice));
  TRY_TMP(hipMalloc((void **)&m->Bm, sizeof(float) * (size_t)KP * m->ldB));
  TRY_TMP(hipMalloc((void **)&m->J0, sizeof(float) * NJ * 3));
  TRY_TMP(hipMalloc((void **)&m->JS, sizeof(float) * NJ * 3 * NB));
  relayout_basis_kernel<<<dim3((unsigned)((m->ldB + 255) / 256)), dim3(256)>>>(m->Pvm, m->Svm, dT.as<float>(), m->Bm, V, m->ldB);
  fold_regressor_kernel<<<dim3(NJ * 3 * (NB + 1)), dim3(256)>>>(dJreg.as<float>(), m->Svm, dT.as<float>(), m->J0, m->JS, V);
  TRY_TMP(hipMalloc((void **)&m->JSp, sizeof(float) * NJ * 3 * 12));
  pack_regressor_rows_kernel<<<dim3(1), dim3(NJ * 3)>>>(m->J0, m->JS, m->JSp);
  m->VGPn = (V + 63) / 64;
  // the split-operand kernels address their basis images with 32-bit buffer offsets: a mesh whose image would reach 2 GiB
  // (more than ~745k vertices for h, ~410k for b) takes the first form (64-bit addressing) from creation on
  for(char * f : {&m->form, &m->form_ik})
  {
    if(*f == 'h' && (int64_t)m->VGPn * HB_SLOTS * HB_IMG > 0x7fffff00LL) *f = 'v';
    if(*f == 'e' && (int64_t)m->VGPn * EB_KS * EB_IMG > 0x7fffff00LL) *f = 'v';
    if(*f == 'b' && (int64_t)m->VGPn * BB_KS * BB_B_BYTES > 0x7fffff00LL) *f = 'v';
  }
  if(uses('b') || uses('e')) // (e falls back to b for models with 5..8 weights per vertex: decided below, once they are counted)
  {
    TRY_TMP(hipMalloc((void **)&m->B3, (size_t)m->VGPn * BB_KS * BB_B_BYTES));
    const int64_t cnt = m->VGPn * BB_KS * 6 * 64;
    relayout_basis_bf16x3_kernel<<<dim3((unsigned)((cnt + 255) / 256)), dim3(256)>>>(m->Bm, m->ldB, V, m->VGPn,
                                                                                  reinterpret_cast<uint16_t *>(m->B3));
  }
  TRY_TMP(hipGetLastError());
  TRY_TMP(hipDeviceSynchronize());
  free_tmp();
#undef TRY_TMP

  // --- skinning weights: keep the non-zeros (real SMPL has <= 4 per vertex), dense fallback otherwise ---
  int maxnz = 0;
  for(int64_t v = 0; v < V; v++)
  {
    int nz = 0;
    for(int j = 0; j < NJ; j++) nz += (W[v * NJ + j] != 0.0f);
    maxnz = std::max(maxnz, nz);
  }
  m->maxw = maxnz <= 4 ? 4 : (maxnz <= 8 ? 8 : NJ);
  const int64_t Vpad = m->VGn * VG;
  std::vector<uint8_t> hIdx((size_t)Vpad * m->maxw, 0);
  std::vector<float> hVal((size_t)Vpad * m->maxw, 0.0f), hSum((size_t)Vpad, 1.0f);
  for(int64_t v = 0; v < V; v++)
  {
    int q = 0;
    float s = 0.0f;
    for(int j = 0; j < NJ; j++)
    {
      float w = W[v * NJ + j];
      s += w; // ascending j, fp32: h[3] = sum_j W[v,j] * 1 (src/LinearBlendSkinning.cpp:463-467)
      if(m->maxw == NJ)
      {
        hIdx[v * NJ + j] = (uint8_t)j;
        hVal[v * NJ + j] = w;
      }
      else if(w != 0.0f)
      {
        hIdx[v * m->maxw + q] = (uint8_t)j;
        hVal[v * m->maxw + q] = w;
        q++;
      }
    }
    hSum[v] = s;
  }
  TRY_OR_FREE(upload(&m->wIdx, hIdx.data(), hIdx.size()));
  TRY_OR_FREE(upload(&m->wVal, hVal.data(), hVal.size()));
  TRY_OR_FREE(upload(&m->wSum, hSum.data(), hSum.size()));
  TRY_OR_FREE(upload(&m->Wdense, W, (size_t)V * NJ));
  // e keeps at most 4 weights per vertex in registers, b and p at most 8: a model with more takes the next form from here on
  // (decided once, so that the layouts kept below are the ones the launches will read)
  for(char * f : {&m->form, &m->form_ik})
  {
    if(*f == 'e' && m->maxw > 4) *f = 'b';
    if(m->maxw > 8 && (*f == 'b' || *f == 'p')) *f = 'v';
  }
  if(uses('e'))
  {
    TRY_OR_FREE(hipMalloc((void **)&m->B3e, (size_t)m->VGPn * EB_KS * EB_IMG));
    TRY_OR_FREE(hipMemset(m->B3e, 0, (size_t)m->VGPn * EB_KS * EB_IMG));
    const int64_t cnt = m->VGPn * EB_KS * 6 * 64;
    relayout_basis_exact_kernel<<<dim3((unsigned)((cnt + 255) / 256)), dim3(256)>>>(m->Bm, m->ldB, V, m->VGPn, m->B3e);
    skin_tables_exact_kernel<<<dim3((unsigned)((m->VGPn * 64 + 255) / 256)), dim3(256)>>>(m->wIdx, m->wVal, m->wSum, m->maxw, V, m->VGPn, m->B3e);
    TRY_OR_FREE(hipGetLastError());
    TRY_OR_FREE(hipDeviceSynchronize());
  }
  if(uses('h'))
  {
    // fp16x2 operands: power-of-two scales that put the largest basis entry / a generous bound of the relative
    // translations (16 x the template's extent) just under fp16's range, so that both pieces of every value that matters
    // are normal fp16 numbers
    float bmax = 0.0f, tmax = 0.0f;
    for(int64_t i = 0; i < V * 3 * NP; i++) bmax = std::max(bmax, std::fabs(P[i]));
    for(int64_t i = 0; i < V * 3 * NB; i++) bmax = std::max(bmax, std::fabs(S[i]));
    for(int64_t i = 0; i < V * 3; i++) tmax = std::max(tmax, std::fabs(vt[i]));
    bmax = std::max(bmax, tmax);
    if(!(bmax > 0.0f) || !std::isfinite(bmax))
    {
      smplpp_model_destroy(m);
      return fail(SMPLPP_ERR_INVALID, "Cannot initialize a SMPL model!");
    }
    m->sB = std::exp2(std::floor(std::log2(32768.0f / bmax)));
    m->sG = std::exp2(std::floor(std::log2(32768.0f / (16.0f * tmax > 1.0f ? 16.0f * tmax : 1.0f))));
    TRY_OR_FREE(hipMalloc((void **)&m->B2h, (size_t)m->VGPn * HB_SLOTS * HB_IMG));
    // Vertex groups by skinning class (common.h, HB_PERM_OFF).  (1) A group is 64 CONSECUTIVE vertices and its class what their
    // weights touch — joints 0..15 only, both halves, joints 16..23 only.  (Sorting the VERTICES by class first, which makes 73 of
    // the synthetic model's 108 groups single-class instead of 11, was measured: the step went from 46 to 62 us — a group's 64
    // output rows of 12 bytes were then scattered over ~200 vertex positions, and the 85 MB of write-once output lost its
    // coalescing.  Models whose vertex order follows the body parts — SMPL's does — have their single-class groups as they are.)
    // (2) the groups dealt round-robin over the eight XCD slices of skin_kernel_h ([x nvg / 8, (x + 1) nvg / 8)), so that every XCD
    // gets the same mix; (3) inside a slice the classes interleaved by fractional rank, so that every workgroup's run of
    // consecutive groups gets it too (a slice of cheap groups beside a slice of full ones would finish with the full ones).
    const int64_t nvg = m->VGPn;
    std::vector<int32_t> hperm((size_t)nvg * 64, -1), gflags((size_t)nvg, 1);
    {
      std::vector<int> vcls((size_t)V);
      for(int64_t v = 0; v < V; v++)
      {
        bool lo = false, hi = false;
        for(int j = 0; j < NJ; j++)
          if(W[v * NJ + j] != 0.0f) (j < 16 ? lo : hi) = true;
        vcls[(size_t)v] = hi ? (lo ? 1 : 2) : 0;
      }
      std::vector<int> tflags((size_t)nvg, 0);
      for(int64_t t = 0; t < nvg; t++)
        for(int i = 0; i < 64 && t * 64 + i < V; i++)
        {
          const int c = vcls[(size_t)(t * 64 + i)];
          tflags[(size_t)t] |= (c == 0 ? 1 : (c == 1 ? 3 : 2));
        }
      // (2) + (3): per XCD slice the sorted groups it is dealt, then their order inside the slice
      std::vector<std::vector<int64_t>> bin(8);
      {
        int x = 0;
        for(int64_t t = 0; t < nvg; t++)
        {
          for(int tries = 0; tries < 8 && (int64_t)bin[x].size() >= (((x + 1) * nvg) >> 3) - ((x * nvg) >> 3); tries++) x = (x + 1) & 7;
          bin[x].push_back(t);
          x = (x + 1) & 7;
        }
      }
      int64_t g = 0;
      for(int x = 0; x < 8; x++)
      {
        int cnt[4] = {0, 0, 0, 0}, seen[4] = {0, 0, 0, 0};
        for(int64_t t : bin[x]) cnt[tflags[(size_t)t]]++;
        std::vector<std::pair<double, int64_t>> keyed;
        for(int64_t t : bin[x])
        {
          const int f = tflags[(size_t)t];
          keyed.push_back({(seen[f] + 0.5) / cnt[f], t});
          seen[f]++;
        }
        std::stable_sort(keyed.begin(), keyed.end(), [](const std::pair<double, int64_t> & a, const std::pair<double, int64_t> & b) { return a.first < b.first; });
        for(auto & kt : keyed)
        {
          const int64_t t = kt.second;
          for(int i = 0; i < 64 && t * 64 + i < V; i++) hperm[(size_t)(g * 64 + i)] = (int32_t)(t * 64 + i);
          gflags[(size_t)g] = tflags[(size_t)t] ? tflags[(size_t)t] : 1;
          g++;
        }
      }
    }
    DevBuf dPerm, dFlags;
    TRY_OR_FREE(dPerm.reserve(sizeof(int32_t) * hperm.size()));
    TRY_OR_FREE(dFlags.reserve(sizeof(int32_t) * gflags.size()));
    TRY_OR_FREE(hipMemcpy(dPerm.p, hperm.data(), sizeof(int32_t) * hperm.size(), hipMemcpyHostToDevice));
    TRY_OR_FREE(hipMemcpy(dFlags.p, gflags.data(), sizeof(int32_t) * gflags.size(), hipMemcpyHostToDevice));
    const int64_t cnt = m->VGPn * (HB_KS * 6 * 64 + 4 * 64 + 64);
    relayout_basis_f16x2_kernel<<<dim3((unsigned)((cnt + 255) / 256)), dim3(256)>>>(m->Bm, m->ldB, m->Wdense, m->wSum, V, m->VGPn,
                                                                                 m->sB, m->sG, m->B2h, dPerm.as<int32_t>(), dFlags.as<int32_t>());
    hipError_t le = hipGetLastError();
    if(le == hipSuccess) le = hipDeviceSynchronize();
    dPerm.release();
    dFlags.release();
    TRY_OR_FREE(le);
  }
  if(!uses('b') && m->B3)
  {
    (void)hipFree(m->B3);
    m->B3 = nullptr;
  }
  if(!uses('p') && !uses('v'))
  {
    (void)hipFree(m->Bm); // only the fp32-MFMA forms read the K-major fp32 basis
    m->Bm = nullptr;
  }
  TRY_OR_FREE(upload(&m->parent, parent.data(), parent.size()));
  {
    const int zero[RANGE_SLOTS] = {};
    TRY_OR_FREE(upload(&m->range_flag, zero, RANGE_SLOTS));
  }
  {
    // joints by depth: the FK chain advances one tree level per step (SMPL: 9 levels)
    std::vector<int32_t> depth(NJ, 0), lv(NJ + 1 + NJ, 0);
    int nlev = 1;
    for(int i = 1; i < NJ; i++)
    {
      depth[i] = depth[parent[i]] + 1;
      nlev = std::max(nlev, depth[i] + 1);
    }
    int pos = 0;
    for(int L = 0; L < nlev; L++)
    {
      lv[L] = pos;
      for(int i = 0; i < NJ; i++)
        if(depth[i] == L) lv[NJ + 1 + pos++] = i;
    }
    lv[nlev] = pos;
    m->nlev = nlev;
    // per (level, slot) the joint and its parent for the pose kernel's chain wavefront (5 joints of a level at a time, 12
    // lanes each): read once into registers instead of three dependent LDS look-ups per level.  chain_fast: the tree has at
    // most CT_LEV levels of at most 5 joints (SMPL: 9 levels, widest 5); other trees take the generic loop.
    m->chain_fast = nlev <= CT_LEV;
    lv.resize(CT_OFF + 60 * CT_LEV * 2, 0);
    for(int q = 0; q < 60 * CT_LEV; q++)
    {
      lv[CT_OFF + 2 * q] = 0x00ffff;
      lv[CT_OFF + 2 * q + 1] = CT_P_ZERO | (CT_P_ZERO << 10) | (1 << 20);
    }
    std::vector<int> slot_of(NJ, 0); // slot of a joint inside its level
    for(int L = 0; L < nlev && m->chain_fast; L++)
    {
      const int cnt = lv[L + 1] - lv[L];
      if(cnt > 5) m->chain_fast = false;
      for(int q = 0; q < cnt && q < 5; q++)
      {
        const int i = lv[NJ + 1 + lv[L] + q];
        slot_of[i] = q;
        const int p = parent[i];
        const int word = i | ((p >= 0 ? p : 0xff) << 8) | ((p >= 0 ? slot_of[p] : 0) << 16); // (the parent sits one level up: already placed)
        for(int e = 0; e < 12; e++)
        {
          const int c = e % 4;
          // the lane's operand: column c of R_i (stride 3), or j_i minus j_p (root: minus zero)
          const int aidx = c < 3 ? CT_P_R + i * 9 + c : CT_P_J + i * 3;
          const int bidx = (c == 3 && p >= 0) ? CT_P_J + p * 3 : CT_P_ZERO;
          lv[CT_OFF + ((q * 12 + e) * CT_LEV + L) * 2] = word;
          lv[CT_OFF + ((q * 12 + e) * CT_LEV + L) * 2 + 1] = aidx | (bidx << 10) | ((c < 3 ? 3 : 1) << 20);
        }
      }
    }
    TRY_OR_FREE(upload(&m->lvl, lv.data(), lv.size()));
  }

  // --- faces + adjacency (src/SMPL.cpp:620-640; emplace keeps one entry per (vertex, face)) ---
  m->h_faces.resize((size_t)F * 3);
  for(int64_t i = 0; i < F * 3; i++) m->h_faces[i] = faces1[i] - 1;
  std::vector<std::vector<int32_t>> adj((size_t)V);
  for(int64_t f = 0; f < F; f++)
    for(int i = 0; i < 3; i++)
    {
      auto & a = adj[m->h_faces[f * 3 + i]];
      if(a.empty() || a.back() != (int32_t)f) a.push_back((int32_t)f);
    }
  m->h_adjOff.assign((size_t)V + 1, 0);
  for(int64_t v = 0; v < V; v++) m->h_adjOff[v + 1] = m->h_adjOff[v] + (int32_t)adj[v].size();
  m->h_adjFace.reserve((size_t)m->h_adjOff[V]);
  for(int64_t v = 0; v < V; v++) m->h_adjFace.insert(m->h_adjFace.end(), adj[v].begin(), adj[v].end());
  TRY_OR_FREE(upload(&m->faces, m->h_faces.data(), m->h_faces.size()));
  TRY_OR_FREE(upload(&m->adjOff, m->h_adjOff.data(), m->h_adjOff.size()));
  TRY_OR_FREE(upload(&m->adjFace, m->h_adjFace.data(), m->h_adjFace.size()));
  {
    // tree tables of the IK evaluation (common.h TREE_*); trees deeper than TREE_DMAX keep the masks only (smplpp_ik_create
    // refuses them)
    std::vector<int32_t> tr(TREE_SIZE, -1), depth(NJ, 0);
    for(int i = 0; i < NJ; i++)
    {
      tr[TREE_ANC + i] = (1 << i) | (i ? tr[TREE_ANC + parent[i]] : 0);
      depth[i] = i ? depth[parent[i]] + 1 : 0;
    }
    int pos = 0;
    for(int L = 0; L <= TREE_DMAX; L++)
    {
      tr[TREE_LVL + L] = pos;
      for(int i = 0; i < NJ && L < TREE_DMAX; i++)
        if(depth[i] == L) tr[TREE_LVLJ + pos++] = i;
    }
    TRY_OR_FREE(upload(&m->anc, tr.data(), tr.size()));
  }
  if(V <= 65535 && F > 0)
  {
    // IK ring tables (topology only): what a task on face f touches when it differentiates a normal — the face's vertices
    // (slots 0..2), then the distinct vertices of the faces around them, first occurrence first; the map gives every
    // (vertex of the face, adjacent face, corner) its slot.  The tables hold `madj` faces per vertex: 12, or 16 when some vertex of
    // this topology has more (the evaluation then runs its 16-face instantiation; beyond 16 a task with a normal term on such a
    // vertex is reported, smplpp_ik_get_status bit 2) — and 3 (madj + 1) + 1 ring vertices.
    int maxval = 0;
    for(int64_t v = 0; v < V; v++) maxval = std::max<int>(maxval, m->h_adjOff[v + 1] - m->h_adjOff[v]);
    m->madj = maxval > MAXADJ ? MAXADJ_WIDE : MAXADJ;
    const int MADJ_ = m->madj, MRING_ = 3 * (MADJ_ + 1) + 1;
    std::vector<uint16_t> ring((size_t)F * (MRING_ + 1), 0);
    std::vector<uint8_t> map((size_t)F * 3 * MADJ_ * 3, 0);
    for(int64_t f = 0; f < F; f++)
    {
      uint16_t * rg = ring.data() + f * (MRING_ + 1);
      uint8_t * mp = map.data() + f * (3 * MADJ_ * 3);
      int nr = 0;
      for(int i = 0; i < 3; i++) rg[1 + nr++] = (uint16_t)m->h_faces[f * 3 + i];
      for(int i = 0; i < 3; i++)
      {
        const int32_t u = m->h_faces[f * 3 + i], b0 = m->h_adjOff[u];
        const int cnt = std::min<int>(m->h_adjOff[u + 1] - b0, MADJ_);
        for(int a = 0; a < cnt; a++)
          for(int cc = 0; cc < 3; cc++)
          {
            const int32_t v = m->h_faces[(int64_t)m->h_adjFace[b0 + a] * 3 + cc];
            int slot = -1;
            for(int q = 0; q < nr; q++)
              if(rg[1 + q] == (uint16_t)v) slot = q;
            if(slot < 0 && nr < MRING_)
            {
              slot = nr;
              rg[1 + nr++] = (uint16_t)v;
            }
            mp[(i * MADJ_ + a) * 3 + cc] = (uint8_t)(slot < 0 ? 0 : slot);
          }
      }
      rg[0] = (uint16_t)nr;
    }
    TRY_OR_FREE(upload(&m->faceRing, ring.data(), ring.size()));
    TRY_OR_FREE(upload(&m->faceMap, map.data(), map.size()));
  }
#undef TRY_OR_FREE
  *out = m;
  return SMPLPP_OK;
}

extern "C" int smplpp_model_info(const smplpp_model * m, int64_t * V, int64_t * F, int * wpv, int * device)
{
  if(!m) return fail(SMPLPP_ERR_INVALID, "smplpp_model_info: null model");
  if(V) *V = m->V;
  if(F) *F = m->F;
  if(wpv) *wpv = m->maxw;
  if(device) *device = m->device;
  return SMPLPP_OK;
}

extern "C" int smplpp_adjacent_faces(const smplpp_model * m, int64_t vertex, int64_t cap, int64_t * faces, float * weights,
                                     int64_t * count)
{
  if(!m || !count) return fail(SMPLPP_ERR_INVALID, "smplpp_adjacent_faces: null argument");
  if(vertex < 0 || vertex >= m->V) return fail(SMPLPP_ERR_INVALID, "smplpp_adjacent_faces: vertex out of range");
  int32_t b = m->h_adjOff[vertex], e = m->h_adjOff[vertex + 1];
  *count = e - b;
  float sum = 0.0f;
  for(int32_t q = b; q < e; q++) sum += 1.0f;
  for(int32_t q = b; q < e && q - b < cap; q++)
  {
    if(faces) faces[q - b] = m->h_adjFace[q];
    if(weights) weights[q - b] = 1.0f / sum; // uniform 1/deg (src/SMPL.cpp:630-639)
  }
  return SMPLPP_OK;
}
