// skin_kernel_q — the fused blend-shape GEMM + linear blend skinning kernel as a persistent, work-queue kernel whose two
// co-resident workgroups per CU are deliberately OUT OF PHASE.
//
// Why: in the first form (skin_kernel, fk.hip) the two wavefronts that share a SIMD run the same program in lock step:
// both in the MFMA phase (halving each other's matrix-pipe rate), then both in the skinning epilogue (matrix pipe idle)
// — 37 % MFMA utilisation at batch 1024 (profiles/r01_pmc_skin_v1.txt).  MI355X_MICROARCH.md "Two waves that run the SAME
// program ...: try a stagger".  Here every workgroup loops over work items pulled from a queue; the second half of the
// grid starts on a HALF-size item (32 frames instead of 64), so from then on one wavefront of each SIMD is in its MFMA
// phase while its partner skins: the matrix pipe stays busy, VALU/LDS/store work rides in its shadow.
//
// Work units are 32-frame x 128-vertex tiles, ordered vertex-quad-major; an item is an aligned pair of units (64 frames,
// the accumulator budget of 2 waves/SIMD) or a single unit.  Per-XCD queues (label = blockIdx % 8: blocks b and b+8
// share an XCD) keep each 338 KB slice of Bm in ONE XCD's L2; a queue that runs dry steals from the others; the single
// units at the end of every queue are the stagger seeds and the fine-grained tail.  Counters are zeroed by a
// hipMemsetAsync node before each launch.  Placement assumptions affect speed only, never results.
#include "common.h"

namespace smplpp_hip
{
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int Q_KSTEPS = KP / 2; // 110
constexpr int Q_UNR = 5;
constexpr int Q_NQUEUE = 8;

struct QueueDesc // per XCD label
{
  int q_rows;      // vertex quads owned by this queue: q = label + 8 * i, i < q_rows
  int pair_units;  // units [0, pair_units) are handed out as aligned pairs (even)
  int total_units; // q_rows * nft1p
};

__device__ inline bool q_fetch(int * ctr, const QueueDesc * qd, int label, bool want_single_first, int & unit, int & cnt)
{
  // try the home queue first, then steal round-robin
  for(int s = 0; s < Q_NQUEUE; s++)
  {
    const int ql = (label + s) & (Q_NQUEUE - 1);
    const QueueDesc d = qd[ql];
    if(d.total_units == 0) continue;
    if(!want_single_first)
    {
      const int p = atomicAdd(&ctr[ql * 2], 1);
      if(2 * p + 1 < d.pair_units)
      {
        unit = (ql << 24) | (2 * p);
        cnt = 2;
        return true;
      }
    }
    const int sidx = atomicAdd(&ctr[ql * 2 + 1], 1);
    if(d.pair_units + sidx < d.total_units)
    {
      unit = (ql << 24) | (d.pair_units + sidx);
      cnt = 1;
      return true;
    }
    if(want_single_first) // no singles left here: fall back to a pair of the same queue
    {
      const int p = atomicAdd(&ctr[ql * 2], 1);
      if(2 * p + 1 < d.pair_units)
      {
        unit = (ql << 24) | (2 * p);
        cnt = 2;
        return true;
      }
    }
  }
  return false;
}

// One work item: FT tiles of 32 frames x this wavefront's 32 vertices.  Identical in structure to skin_kernel (fk.hip).
template<int FT, int MAXW>
__device__ __forceinline__ void q_item(const float * __restrict__ AT, int64_t ldA, const float * __restrict__ Bm, int64_t ldB,
                                       const float * __restrict__ Gp, const float * __restrict__ theta, const uint8_t * __restrict__ wIdx,
                                       const float * __restrict__ wVal, const float * __restrict__ wSum, float * __restrict__ verts,
                                       float * __restrict__ rest, int64_t n, int64_t V, int VGn, int64_t f0, int vg, float * sG, float * sRoot)
{
  constexpr int FRAMES = 32 * FT;
  const int tid = threadIdx.x, lane = tid & 63;
  // stage G' and root translations of this item (frames >= n read as zero)
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
    float a_cur[Q_UNR][FT], b_cur[Q_UNR][3], a_nxt[Q_UNR][FT], b_nxt[Q_UNR][3];
#pragma unroll
    for(int u = 0; u < Q_UNR; u++)
    {
#pragma unroll
      for(int t = 0; t < FT; t++) a_cur[u][t] = Ap[(int64_t)(2 * u) * ldA + 32 * t];
#pragma unroll
      for(int x = 0; x < 3; x++) b_cur[u][x] = Bp[(int64_t)(2 * u) * ldB + VG * x];
    }
    for(int c = 0; c < Q_KSTEPS / Q_UNR; c++)
    {
      if(c + 1 < Q_KSTEPS / Q_UNR)
      {
        const float * An = Ap + (int64_t)(2 * Q_UNR) * (c + 1) * ldA;
        const float * Bn = Bp + (int64_t)(2 * Q_UNR) * (c + 1) * ldB;
#pragma unroll
        for(int u = 0; u < Q_UNR; u++)
        {
#pragma unroll
          for(int t = 0; t < FT; t++) a_nxt[u][t] = An[(int64_t)(2 * u) * ldA + 32 * t];
#pragma unroll
          for(int x = 0; x < 3; x++) b_nxt[u][x] = Bn[(int64_t)(2 * u) * ldB + VG * x];
        }
      }
#pragma unroll
      for(int u = 0; u < Q_UNR; u++)
#pragma unroll
        for(int t = 0; t < FT; t++)
#pragma unroll
          for(int x = 0; x < 3; x++) acc[t][x] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[u][t], b_cur[u][x], acc[t][x], 0, 0, 0);
#pragma unroll
      for(int u = 0; u < Q_UNR; u++)
      {
#pragma unroll
        for(int t = 0; t < FT; t++) a_cur[u][t] = a_nxt[u][t];
#pragma unroll
        for(int x = 0; x < 3; x++) b_cur[u][x] = b_nxt[u][x];
      }
    }
  }
  __syncthreads(); // G' staged

  const int64_t v = (int64_t)vg * VG + (lane & 31);
  if(vg >= VGn || v >= V) return;
  int jidx[MAXW];
  float jw[MAXW];
#pragma unroll
  for(int i = 0; i < MAXW; i++)
  {
    jidx[i] = wIdx[v * MAXW + i];
    jw[i] = wVal[v * MAXW + i];
  }
  // cart = h[:3] / h[3] (src/LinearBlendSkinning.cpp:545-550) with h[3] = sum_j W[v,j], constant per vertex:
  // one reciprocal per lane instead of three IEEE divisions per (frame, vertex) — differs by <= 1 ulp
  const float winv = 1.0f / wSum[v];
  float * vout = verts ? verts + v * 3 : nullptr;
  float * rout = rest ? rest + v * 3 : nullptr;
#pragma unroll
  for(int t = 0; t < FT; t++)
#pragma unroll
    for(int r = 0; r < 16; r++)
    {
      const int fl = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); // accumulator row -> frame in item
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

template<int MAXW>
__global__ __launch_bounds__(256, 2) void skin_kernel_q(const float * __restrict__ AT, int64_t ldA, const float * __restrict__ Bm,
                                                        int64_t ldB, const float * __restrict__ Gp, const float * __restrict__ theta,
                                                        const uint8_t * __restrict__ wIdx, const float * __restrict__ wVal,
                                                        const float * __restrict__ wSum, float * __restrict__ verts,
                                                        float * __restrict__ rest, int64_t n, int64_t V, int VGn, int nft1p,
                                                        int * __restrict__ ctr, const QueueDesc * __restrict__ qdesc)
{
  extern __shared__ __attribute__((aligned(16))) float lds[]; // [64][24][12] G' + [64][3] root translation
  float * sG = lds;
  float * sRoot = lds + 64 * NJ * 12;
  __shared__ int s_unit, s_cnt;
  const int tid = threadIdx.x, wave = tid >> 6;
  const int label = blockIdx.x & (Q_NQUEUE - 1);
  bool first = true;
  const bool late_half = blockIdx.x >= (gridDim.x >> 1);

  for(;;)
  {
    __syncthreads(); // every wavefront is done with sG / s_unit of the previous item
    if(tid == 0)
    {
      int u = -1, c = 0;
      if(!q_fetch(ctr, qdesc, label, first && late_half, u, c)) u = -1;
      s_unit = u;
      s_cnt = c;
    }
    __syncthreads();
    const int unit = s_unit;
    if(unit < 0) break;
    first = false;
    const int ql = unit >> 24, uidx = unit & 0xffffff;
    const int q = ql + Q_NQUEUE * (uidx / nft1p);
    const int ft1 = uidx % nft1p;
    const int64_t f0 = (int64_t)ft1 * 32;
    if(f0 >= n) continue; // phantom tile of an odd frame-tile count
    const int vg = q * 4 + wave;
    if(s_cnt == 2 && f0 + 32 < n)
      q_item<2, MAXW>(AT, ldA, Bm, ldB, Gp, theta, wIdx, wVal, wSum, verts, rest, n, V, VGn, f0, vg, sG, sRoot);
    else
      q_item<1, MAXW>(AT, ldA, Bm, ldB, Gp, theta, wIdx, wVal, wSum, verts, rest, n, V, VGn, f0, vg, sG, sRoot);
  }
}

// host side ---------------------------------------------------------------------------------------------------------
template<int MAXW>
static hipError_t launch_q(smplpp_model * m, int64_t n, const float * theta, float * verts, float * rest, hipStream_t st)
{
  static int cus = 0;
  if(!cus)
  {
    hipDeviceProp_t prop;
    cus = (hipGetDeviceProperties(&prop, m->device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  Workspace & ws = m->ws;
  const int nft1 = (int)((n + 31) / 32);
  const int nft1p = (nft1 + 1) & ~1;
  const int nq = (int)((m->VGn + 3) / 4);
  const int total_units = nq * nft1p;
  int grid = 2 * cus;
  if(grid > total_units) grid = total_units < 8 ? 8 : (total_units + 7) & ~7;
  hipError_t e;
  if((e = ws.q_ctr.reserve(sizeof(int) * 2 * Q_NQUEUE)) != hipSuccess) return e;
  if((e = ws.q_desc.reserve(sizeof(QueueDesc) * Q_NQUEUE)) != hipSuccess) return e;
  if(ws.q_n != n || ws.q_grid != grid)
  {
    QueueDesc h[Q_NQUEUE];
    // singles per queue: one stagger seed for every late-half workgroup of this XCD + as many again for the tail
    const int singles_target = ((grid / Q_NQUEUE) + 1) & ~1;
    for(int x = 0; x < Q_NQUEUE; x++)
    {
      const int rows = (nq > x) ? (nq - x + Q_NQUEUE - 1) / Q_NQUEUE : 0;
      h[x].q_rows = rows;
      h[x].total_units = rows * nft1p;
      const int singles = singles_target < h[x].total_units ? singles_target : h[x].total_units;
      h[x].pair_units = (h[x].total_units - singles) & ~1;
    }
    // the descriptor table is tiny; a synchronous copy (only when the batch size changes) keeps lifetimes simple
    if((e = hipMemcpy(ws.q_desc.p, h, sizeof(h), hipMemcpyHostToDevice)) != hipSuccess) return e;
    ws.q_n = n;
    ws.q_grid = grid;
  }
  if((e = hipMemsetAsync(ws.q_ctr.p, 0, sizeof(int) * 2 * Q_NQUEUE, st)) != hipSuccess) return e;
  const size_t shmem = sizeof(float) * 64 * (NJ * 12 + 3);
  static bool attr_set = false;
  if(!attr_set)
  {
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&skin_kernel_q<MAXW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if(e != hipSuccess) return e;
    attr_set = true;
  }
  skin_kernel_q<MAXW><<<dim3(grid), dim3(256), shmem, st>>>(ws.AT.as<float>(), ws.ldA, m->Bm, m->ldB, ws.Gp.as<float>(), theta, m->wIdx, m->wVal,
                                                           m->wSum, verts, rest, n, m->V, (int)m->VGn, nft1p, ws.q_ctr.as<int>(),
                                                           ws.q_desc.as<QueueDesc>());
  return hipGetLastError();
}

hipError_t launch_skin_queue(smplpp_model * m, int64_t n, const float * theta, float * verts, float * rest, hipStream_t st)
{
  if(m->maxw == 4) return launch_q<4>(m, n, theta, verts, rest, st);
  if(m->maxw == 8) return launch_q<8>(m, n, theta, verts, rest, st);
  return hipErrorInvalidValue;
}
} // namespace smplpp_hip
