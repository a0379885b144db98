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
#ifdef POSE_STAMP
namespace smplpp_hip
{
__device__ unsigned long long g_pose_stamps[16];
}
#define PST(i) if(blockIdx.x == 512 && threadIdx.x == 0) smplpp_hip::g_pose_stamps[i] = __builtin_amdgcn_s_memtime()
#define PSTC(i) if(blockIdx.x == 512 && threadIdx.x == 192) smplpp_hip::g_pose_stamps[i] = __builtin_amdgcn_s_memtime()
extern "C" int smplpp_debug_pose_stamps(unsigned long long * out)
{
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(smplpp_hip::g_pose_stamps), sizeof(unsigned long long) * 16);
}
#endif
#include "pose_body.h"
#include "trace.h"

namespace smplpp_hip
{
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// ---------------------------------------------------------------------------------------------- pose kernel
// (the body: pose_body.h)
__global__ __launch_bounds__(256) void pose_kernel(PoseArgs a)
{
  const int64_t f = blockIdx.x;
  if(f >= a.n) return;
  pose_body(a, f, (int)threadIdx.x, a.theta + f * ((NJ + 1) * 3));
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
hipError_t launch_skin_exact(const smplpp_model * m, int64_t n, const float * theta, float * verts, float * rest, hipStream_t st);  // skin_e.hip

// Device-pointer FK (enqueue only).  Used by smplpp_fk and by the IK solver.
// The pose step's arguments for the model's workspace (form h: with_ops adds the fused kernel's operand images; the other forms
// fill in their own operand).  The workspace buffers must have been reserved (fk_device does, before anything reads the result).
PoseArgs fk_pose_args(smplpp_model * m, int64_t n, const float * beta, const float * theta, float * joints, float * poserot, float * xforms44,
                      bool with_ops)
{
  Workspace & ws = m->ws;
  PoseArgs pa;
  pa.beta = beta;
  pa.theta = theta;
  pa.J0 = m->J0;
  pa.JS = m->JS;
  pa.JSp = m->JSp;
  pa.parent = m->parent;
  pa.lvl_off = m->lvl;
  pa.lvl_joint = m->lvl + NJ + 1;
  pa.nlev = m->nlev;
  pa.AT = nullptr;
  pa.ldA = 0;
  pa.Gp = ws.Gp.as<float>();
  pa.joints_out = joints;
  pa.rot_out = poserot;
  pa.xf44_out = xforms44;
  pa.n = n;
  pa.A3 = nullptr;
  pa.A2h = with_ops ? ws.A2h.as<_Float16>() : nullptr;
  pa.G2h = with_ops ? ws.G2h.as<_Float16>() : nullptr;
  pa.gscale = m->sG;
  pa.ctab = m->chain_fast ? m->lvl + CT_OFF : nullptr;
  pa.range_flag = m->range_flag;
  return pa;
}

// Form of the fused kernel a launch runs (decided once per launch, here, for both halves of the forward pass).  m->form (from
// SMPLPP_SKIN at model creation; default e) is what smplpp_fk runs: e (skin_e.hip) carries every fp32 operand exactly (bf16x3
// pieces, six MFMA products per fp32 product, fp32 VALU skinning) — the reference's arithmetic; h (skin_h.hip): fp16x2 pieces,
// 22-bit operands, skinning on the matrix pipe too; b (skin_b.hip): round 1's bf16x3 kernel; p (skin_p.hip): fp32 MFMA; v: the
// first form.  The IK / VPoser loops' internal launches (range_slot RANGE_INTERNAL: intermediate iterates whose mesh feeds the
// residual's few vertices and the re-projection's face scan) run m->form_ik: h unless SMPLPP_SKIN chose a form for everything.
// p (32-bit output offsets) falls back to v for outputs of 2 GiB and more.
static char launch_form(const smplpp_model * m, int64_t n, int range_slot)
{
  char form = range_slot == RANGE_INTERNAL ? m->form_ik : m->form;
  if(form == 'p' && n * m->V * 12 >= 0x7fffff00LL) form = 'v';
  return form;
}

// range_slot: which word of the model's range status a launch of the fp16x2 form reports to (common.h RANGE_*): enqueue-only user
// launches, host-space user launches and the IK / VPoser loops' internal launches each have their own, so that an intermediate IK
// iterate outside the range does not turn a later, in-range smplpp_fk into an error
// Pose step of a launch of form `form`: joints, relative transforms and the fused kernel's operand images (when `with_ops`) into
// the model's workspace.
static int fk_pose_device(smplpp_model * m, char form, int64_t n, const float * beta, const float * theta, float * joints, float * xforms44,
                          float * poserot, hipStream_t st, int * range_word, bool with_ops)
{
  Workspace & ws = m->ws;
  const int64_t n64 = ((n + 63) / 64) * 64;
  HIP_TRY(ws.Gp.reserve(sizeof(float) * (size_t)n64 * NJ * 12)); // e / b / p stage whole frame tiles of G' (padding never stored)
  if(form == 'h')
  {
    HIP_TRY(ws.A2h.reserve((size_t)(n64 / 64) * HB_KS * HB_A_BYTES));
    HIP_TRY(ws.G2h.reserve((size_t)(n64 / 64) * HB_G_BYTES));
    PoseArgs pa = fk_pose_args(m, n, beta, theta, joints, poserot, xforms44, with_ops);
    pa.range_flag = range_word;
    pose_kernel<<<dim3((unsigned)n), dim3(256), 0, st>>>(pa);
  }
  else if(form == 'e' || form == 'b')
  {
    HIP_TRY(ws.A3.reserve((size_t)(n64 / 64) * BB_KS * BB_A_BYTES));
    PoseArgs pa = fk_pose_args(m, n, beta, theta, joints, poserot, xforms44, false);
    pa.A3 = with_ops ? ws.A3.as<uint16_t>() : nullptr;
    pa.gscale = 1.0f;
    pa.range_flag = nullptr;
    pose_kernel<<<dim3((unsigned)n), dim3(256), 0, st>>>(pa);
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
    PoseArgs pa = fk_pose_args(m, n, beta, theta, joints, poserot, xforms44, false);
    pa.AT = ws.AT.as<float>();
    pa.ldA = ldA;
    pa.gscale = 1.0f;
    pa.range_flag = nullptr;
    pose_kernel<<<dim3((unsigned)n), dim3(256), 0, st>>>(pa);
  }
  HIP_TRY(hipGetLastError());
  return SMPLPP_OK;
}

// The fused kernel of form `form` from the workspace the pose step filled.  `theta` is read for the root translation only
// (theta[f, 0, :], stride 75).
static int fk_skin_device(smplpp_model * m, char form, int64_t n, const float * theta, float * verts, float * rest, hipStream_t st)
{
  Workspace & ws = m->ws;
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
    if(form == 'e')
      HIP_TRY(launch_skin_exact(m, n, theta, verts, rest, st));
    else if(form == 'h')
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

// range_word: where a launch of the fp16x2 form reports an operand outside its range — the model's word of `range_slot`, or the word
// of the IK solver whose loop the launch belongs to (each solver has its own: one solver's overflow is not another's status bit)
int fk_device(smplpp_model * m, int64_t n, const float * beta, const float * theta, float * verts, float * joints,
              float * xforms44, float * rest, float * poserot, hipStream_t st, int range_slot, int * range_word)
{
  const char form = launch_form(m, n, range_slot);
  int rc = fk_pose_device(m, form, n, beta, theta, joints, xforms44, poserot, st, range_word ? range_word : m->range_flag + range_slot, verts || rest);
  if(rc) return rc;
  return fk_skin_device(m, form, n, theta, verts, rest, st);
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

// reads and clears the enqueue-only launches' range word of the fp16x2 form (the caller has synchronised the stream)
static int fk_range_status(smplpp_model * m, int * bits)
{
  *bits = 0;
  if(!m->range_flag || m->form != 'h') return SMPLPP_OK;
  HIP_TRY(hipMemcpy(bits, m->range_flag + RANGE_DEVICE, sizeof(int), hipMemcpyDeviceToHost));
  if(*bits) HIP_TRY(hipMemset(m->range_flag + RANGE_DEVICE, 0, sizeof(int)));
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
  if(space == SMPLPP_DEVICE) return fk_device(m, n, beta, theta, verts, joints, xforms, rest, nullptr, st, RANGE_DEVICE, nullptr);

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
  // this call's own range word: cleared in front of the launch, read back on the launch stream beside the results
  const bool ranged = m->range_flag && m->form == 'h';
  if(ranged) HIP_TRY(hipMemsetAsync(m->range_flag + RANGE_HOST, 0, sizeof(int), st));
  int rc = fk_device(m, n, ws.beta.as<float>(), ws.theta.as<float>(), verts ? ws.verts.as<float>() : nullptr,
                     joints ? ws.joints.as<float>() : nullptr, xforms ? ws.xf44.as<float>() : nullptr,
                     rest ? ws.rest.as<float>() : nullptr, nullptr, st, RANGE_HOST, nullptr);
  if(rc) return rc;
  if(verts) HIP_TRY(hipMemcpyAsync(verts, ws.verts.p, nv, hipMemcpyDeviceToHost, st));
  if(rest) HIP_TRY(hipMemcpyAsync(rest, ws.rest.p, nv, hipMemcpyDeviceToHost, st));
  if(joints) HIP_TRY(hipMemcpyAsync(joints, ws.joints.p, sizeof(float) * (size_t)n * NJ * 3, hipMemcpyDeviceToHost, st));
  if(xforms) HIP_TRY(hipMemcpyAsync(xforms, ws.xf44.p, sizeof(float) * (size_t)n * NJ * 16, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  // (read AFTER the synchronisation, synchronously: an asynchronous copy into this frame's stack could still be pending when one of
  // the copies above fails and the function returns)
  int bits = 0;
  if(ranged) HIP_TRY(hipMemcpy(&bits, m->range_flag + RANGE_HOST, sizeof(int), hipMemcpyDeviceToHost));
  if(bits & 1)
    return fail(SMPLPP_ERR_NUMERIC, "smplpp_fk: an operand left the range of the fp16x2 form (|beta| < 1023, relative transforms within 16 x the "
                                    "template's extent): the vertices of such frames are not finite; create the model under SMPLPP_SKIN=b or p");
  return SMPLPP_OK;
}
