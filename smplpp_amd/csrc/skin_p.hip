// skin_kernel_p — persistent, software-pipelined form of the fused blend-shape GEMM + linear blend skinning kernel.
//
// Why: the first form (skin_kernel in fk.hip) runs two wavefronts per SIMD through the SAME program in lock step —
// both in the MFMA phase (sharing the matrix pipe), then both in the skinning epilogue (matrix pipe idle): 37 % MFMA
// utilisation at batch 1024.  Here ONE wavefront per SIMD owns the whole register file and walks a list of work items
// (32 frames x 32 vertices x {x,y,z} = 330 v_mfma_f32_32x32x2_f32); the skinning epilogue of item i is issued in the
// shadow of the MFMAs of item i+1 (an MFMA occupies the matrix pipe for 64 cycles; the ~70 VALU + 13 LDS reads of one
// epilogue row fit behind the 15 MFMAs of one k-chunk), so the matrix pipe never waits for the VALU work.
//
// Work decomposition: items are ordered vertex-group-major / frame-tile-minor and dealt to workgroups in contiguous
// runs, so a workgroup's four wavefronts stream the same <= 2 slices of Bm (84 KB each: L1/L2 hits) against different
// frame tiles.  No workgroup barrier anywhere: every wavefront is independent and owns a private LDS region holding
// the relative transforms G' (32 frames x 24 x 12 floats) and root translations of the item being skinned.
// Operands go L2 -> VGPR directly (one dword per lane per MFMA), double-buffered one k-chunk (960 MFMA-cycles) ahead.
#include "common.h"

#include <cstdlib>
#include <type_traits>
#include <utility>

namespace smplpp_hip
{
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v3f __attribute__((ext_vector_type(3), aligned(4)));
typedef unsigned v3u __attribute__((ext_vector_type(3)));
typedef float v4f __attribute__((ext_vector_type(4))); // first-class vector: an array of these is promoted to registers

constexpr int P_KSTEPS = KP / 2;          // 110
constexpr int P_UNR = 5;                  // k-steps per chunk
constexpr int P_CHUNKS = P_KSTEPS / P_UNR; // 22
constexpr int P_GWAVE = 32 * NJ * 12;     // floats of G' per item (9216)
constexpr int P_LDS_WAVE = P_GWAVE + 128; // + root translations (96) padded

template<class F, int... I>
__device__ __forceinline__ void static_for_impl(F && f, std::integer_sequence<int, I...>)
{
  (f(std::integral_constant<int, I>{}), ...);
}
template<int N, class F>
__device__ __forceinline__ void static_for(F && f)
{
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

struct ItemCtx // everything the epilogue of an item needs after its MFMAs are done
{
  int64_t f0;
  float * vout; // verts + v * 3 (nullptr: lane has no vertex)
  float * rout;
  float winv;
  int voff;     // byte offset of (frame f0 + 4 * half, vertex v) in an output array; out of range when the lane has no vertex
};

template<int MAXW, bool WANT_REST>
__global__ __launch_bounds__(256, 1) void skin_kernel_p(const float * __restrict__ AT, int64_t ldA, const float * __restrict__ Bm,
                                                        int64_t ldB, const float * __restrict__ Gp, const float * __restrict__ theta,
                                                        const uint8_t * __restrict__ wIdx, const float * __restrict__ wVal,
                                                        const float * __restrict__ wSum, float * __restrict__ verts,
                                                        float * __restrict__ rest, float * __restrict__ dummy, int64_t n, int64_t V, int VGn,
                                                        int nft, int items_per_block, int dbg_mode)
{
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
  float * sG = lds + wave * P_LDS_WAVE;
  float * sRoot = sG + P_GWAVE;
  const int total = VGn * nft;
  // XCD-aware run assignment: workgroups b and b + 8 share an XCD (round-robin dispatch), so give XCD x the x-th
  // CONTIGUOUS eighth of the item list: every 84 KB slice of Bm is then streamed by one XCD's L2 only.
  // (A wrong placement guess costs speed, never correctness: the runs tile the item list either way.)
  const int nb = gridDim.x, per_x = (nb + 7) >> 3;
  const int vb = (int)(blockIdx.x & 7) * per_x + (int)(blockIdx.x >> 3); // bijective when nb % 8 == 0
  const int t_begin = vb * items_per_block + wave;
  int t_end = (vb + 1) * items_per_block;
  if(t_end > total) t_end = total;
  if(t_begin >= t_end) return;

  f32x16 acc[3], accp[3];
  float abuf[2][P_UNR], bbuf[2][P_UNR][3];
  int jidx[MAXW], jidxp[MAXW];
  float jw[MAXW], jwp[MAXW];
  ItemCtx cur, prev;

  // Operand addressing: buffer loads (T8) — a 128-bit buffer descriptor per operand in SGPRs, a per-lane byte offset
  // that never changes (VGPR) and a wave-uniform byte offset per (item, k-row) in an SGPR: no per-load VALU address
  // arithmetic and no 64-bit address registers.
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(AT), 0, (int)(KP * ldA * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Bm), 0, (int)(KP * ldB * 4), 0x00020000);
  // outputs: same scheme; the range check of the descriptor (num_records = n * V * 12 bytes) drops the stores of frames
  // >= n and of lanes without a vertex (offset forced out of range), so the epilogue needs neither branches nor a dummy line
  // (the launcher only picks this form while n * V * 12 fits the 32-bit num_records)
  const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(verts, 0, (int)(verts ? n * V * 12 : 0), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(rest, 0, (int)(rest ? n * V * 12 : 0), 0x00020000);
  const int frameB = (int)(V * 12); // bytes per frame of output
  const int laneA = (int)(((int64_t)half * ldA + l31) * 4);
  const int laneB = (int)(((int64_t)half * ldB + l31) * 4);
  const int rowA = (int)(ldA * 4), rowB = (int)(ldB * 4); // bytes per k-row
  auto item_bases = [&](int t, int & Ab, int & Bb) {
    const int tu = __builtin_amdgcn_readfirstlane(t); // t is wave-uniform; say so
    const int vg = tu / nft, ft = tu % nft;
    Ab = ft * 32 * 4;
    Bb = vg * (3 * VG) * 4;
    if(dbg_mode == 1) // timing experiment: every item streams the same operand tile (pure L1/L2 hits)
    {
      Ab = 0;
      Bb = 0;
    }
  };
  auto ldA_ = [&](int Ab, int k2) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsA, laneA, Ab + k2 * rowA, 0)); };
  auto ldB_ = [&](int Bb, int k2, int x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsB, laneB + VG * 4 * x, Bb + k2 * rowB, 0));
  };
  auto load_chunk = [&](int Ab, int Bb, int c, float (&a)[P_UNR], float (&b)[P_UNR][3]) {
#pragma unroll
    for(int u = 0; u < P_UNR; u++)
    {
      const int k2 = 2 * (c * P_UNR + u);
      a[u] = ldA_(Ab, k2);
#pragma unroll
      for(int x = 0; x < 3; x++) b[u][x] = ldB_(Bb, k2, x);
    }
  };

  int Abase, Bbase;
  item_bases(t_begin, Abase, Bbase);
  load_chunk(Abase, Bbase, 0, abuf[0], bbuf[0]);

  // one work item; HP (compile time) = there is a previous item whose epilogue rides in this item's MFMA shadow.
  // Everything inside a chunk is straight-line code (stores of masked-off lanes go to a dummy line) so that the
  // scheduler sees MFMAs and epilogue in ONE region and can interleave them.
  auto do_item = [&](int t, auto hp_tag) {
    constexpr bool HP = decltype(hp_tag)::value;
    const int vg = t / nft, ft = t % nft;
    const int64_t v = (int64_t)vg * VG + l31;
    const bool has_v = v < V;
    cur.f0 = (int64_t)ft * 32;
    cur.vout = (has_v && verts) ? verts + v * 3 : nullptr;
    cur.rout = (has_v && WANT_REST && rest) ? rest + v * 3 : nullptr;
    cur.voff = has_v ? (int)(v * 12 + (int64_t)(4 * half) * frameB) : 0x7fffff00;
    {
      const int64_t vv = has_v ? v : 0;
#pragma unroll
      for(int i = 0; i < MAXW; i++)
      {
        jidx[i] = wIdx[vv * MAXW + i];
        jw[i] = wVal[vv * MAXW + i];
      }
      cur.winv = 1.0f / wSum[vv]; // one reciprocal per lane instead of IEEE divisions (<= 1 ulp: 6e-8 m at 1 m)
    }
    const int tn = (t + 4 < t_end) ? t + 4 : t; // next item (or this one again: harmless extra prefetch)
    int Abn, Bbn;
    item_bases(tn, Abn, Bbn);
#pragma unroll
    for(int x = 0; x < 3; x++)
#pragma unroll
      for(int r = 0; r < 16; r++) acc[x][r] = 0.0f;

    const v4f * gsrc = reinterpret_cast<const v4f *>(Gp + cur.f0 * (NJ * 12));
    static_for<P_CHUNKS>([&](auto cc) {
      constexpr int C = decltype(cc)::value;
      // Hand-placed instruction stream: one "slot" per MFMA.  After each MFMA is issued (it owns the matrix pipe for
      // 64 cycles) the wavefront issues that slot's share of everything else — operand loads for the next chunk, one
      // piece of the previous item's skinning row (LDS reads one slot ahead of the FMAs that consume them), G' staging
      // for the current item — so that all of it executes in the MFMA's shadow.  sched_barrier(0) pins the order.
      float rx = 0.f, ry = 0.f, rz = 0.f, rt0 = 0.f, rt1 = 0.f, rt2 = 0.f;
      float4 m0 = make_float4(0.f, 0.f, 0.f, 0.f), m1 = m0, m2 = m0, g0 = m0, g1 = m0, g2 = m0;
      v4f gstage[6];
      float rstage[2] = {0.f, 0.f};
      constexpr int R = C < 16 ? C : 0;
      const int fl = (R & 3) + 8 * (R >> 2) + 4 * half; // accumulator row -> frame in tile
      const float * gfr = sG + fl * (NJ * 12);
      static_for<P_UNR * 3>([&](auto ss) {
        constexpr int S = decltype(ss)::value;
        constexpr int U = S / 3, X = S % 3;
        acc[X] = __builtin_amdgcn_mfma_f32_32x32x2f32(abuf[C & 1][U], bbuf[C & 1][U][X], acc[X], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // ---- G' tile / root translation of the CURRENT item (chunks 16..21; the previous item's rows are done by then)
        if constexpr(C >= 16 && S < 6) gstage[S] = gsrc[((C - 16) * 6 + S) * 64 + lane];
        if constexpr(C == 16 && S == 6)
        {
          // root translation of frame fl, component x: theta[f, 0, x] (src/SMPL.cpp:726-727); 96 values per tile
          const int e0 = lane, e1 = 64 + lane;
          const int64_t fa = cur.f0 + e0 / 3, fb = cur.f0 + e1 / 3;
          rstage[0] = (fa < n) ? theta[fa * (NJ + 1) * 3 + e0 % 3] : 0.0f;
          rstage[1] = (lane < 32 && fb < n) ? theta[fb * (NJ + 1) * 3 + e1 % 3] : 0.0f;
        }
        if constexpr(C >= 16 && S >= 9) reinterpret_cast<v4f *>(sG)[((C - 16) * 6 + (S - 9)) * 64 + lane] = gstage[S - 9];
        if constexpr(C == 16 && S == 14)
        {
          sRoot[lane] = rstage[0];
          if(lane < 32) sRoot[64 + lane] = rstage[1];
        }
        // ---- operand prefetch: loads 3S .. 3S+2 of the 20 that make up the next chunk (chunk 0 of the next item last)
#pragma unroll
        for(int l = 3 * S; l < 3 * S + 3 && l < 4 * P_UNR; l++)
        {
          const int u = l / 4, w = l % 4; // per k-step: A, B.x, B.y, B.z
          const int cn = (C + 1 < P_CHUNKS) ? C + 1 : 0;
          const int Ab = (C + 1 < P_CHUNKS) ? Abase : Abn, Bb = (C + 1 < P_CHUNKS) ? Bbase : Bbn;
          const int k2 = 2 * (cn * P_UNR + u);
          if(w == 0)
            abuf[(C + 1) & 1][u] = ldA_(Ab, k2);
          else
            bbuf[(C + 1) & 1][u][w - 1] = ldB_(Bb, k2, w - 1);
        }
        // ---- one piece of row R of the PREVIOUS item (branch-free: dead lanes store to a dummy line)
        if constexpr(HP && C < 16)
        {
          if constexpr(S == 0)
          {
            rx = accp[0][R];
            ry = accp[1][R];
            rz = accp[2][R];
            if constexpr(WANT_REST)
            {
              v3f ov = {rx, ry, rz};
              __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(v3u, ov), rsR, prev.voff,
                                                    __builtin_amdgcn_readfirstlane((int)(prev.f0 + ((R & 3) + 8 * (R >> 2))) * frameB), 2);
            }
          }
          if constexpr(MAXW == 4)
          {
            // 4 joints x 3 matrix rows = 12 groups of 4 FMAs, one group per slot 1..12; joint j is read from LDS in slot 3j
            // (one MFMA ahead of its first use), the root translation in slot 12, the 9 + 3 closing FMAs in slots 13 / 14
            if constexpr(S >= 1 && S <= 12)
            {
              constexpr int J = (S - 1) / 3, ROW = (S - 1) % 3;
              const float w = jwp[J];
              if constexpr(ROW == 0) { m0.x += w * g0.x; m0.y += w * g0.y; m0.z += w * g0.z; m0.w += w * g0.w; }
              if constexpr(ROW == 1) { m1.x += w * g1.x; m1.y += w * g1.y; m1.z += w * g1.z; m1.w += w * g1.w; }
              if constexpr(ROW == 2) { m2.x += w * g2.x; m2.y += w * g2.y; m2.z += w * g2.z; m2.w += w * g2.w; }
            }
            if constexpr(S % 3 == 0 && S / 3 < 4)
            {
              const float4 * gj = reinterpret_cast<const float4 *>(gfr + jidxp[S / 3] * 12);
              g0 = gj[0];
              g1 = gj[1];
              g2 = gj[2];
            }
            if constexpr(S == 12)
            {
              rt0 = sRoot[fl * 3 + 0];
              rt1 = sRoot[fl * 3 + 1];
              rt2 = sRoot[fl * 3 + 2];
            }
            if constexpr(S == 13)
            {
              const float hx = ((m0.x * rx + m0.y * ry) + m0.z * rz) + m0.w;
              const float hy = ((m1.x * rx + m1.y * ry) + m1.z * rz) + m1.w;
              const float hz = ((m2.x * rx + m2.y * ry) + m2.z * rz) + m2.w;
              rx = hx; // carried to the store slot
              ry = hy;
              rz = hz;
            }
            if constexpr(S == 14)
            {
              // write-once output: non-temporal (aux = 2), so 85 MB of vertices do not evict the Bm slices from this XCD's L2
              v3f ov = {rx * prev.winv + rt0, ry * prev.winv + rt1, rz * prev.winv + rt2};
              __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(v3u, ov), rsV, prev.voff,
                                                    __builtin_amdgcn_readfirstlane((int)(prev.f0 + ((R & 3) + 8 * (R >> 2))) * frameB), 2);
            }
          }
          else
          {
            if constexpr(S >= 1 && S <= MAXW) // consume joint S-1 (read one slot earlier)
            {
              const float w = jwp[S - 1];
              m0.x += w * g0.x; m0.y += w * g0.y; m0.z += w * g0.z; m0.w += w * g0.w;
              m1.x += w * g1.x; m1.y += w * g1.y; m1.z += w * g1.z; m1.w += w * g1.w;
              m2.x += w * g2.x; m2.y += w * g2.y; m2.z += w * g2.z; m2.w += w * g2.w;
            }
            if constexpr(S < MAXW) // issue the LDS reads of joint S
            {
              const float4 * gj = reinterpret_cast<const float4 *>(gfr + jidxp[S] * 12);
              g0 = gj[0];
              g1 = gj[1];
              g2 = gj[2];
            }
            if constexpr(S == MAXW)
            {
              rt0 = sRoot[fl * 3 + 0];
              rt1 = sRoot[fl * 3 + 1];
              rt2 = sRoot[fl * 3 + 2];
            }
            if constexpr(S == MAXW + 1)
            {
              const float hx = ((m0.x * rx + m0.y * ry) + m0.z * rz) + m0.w;
              const float hy = ((m1.x * rx + m1.y * ry) + m1.z * rz) + m1.w;
              const float hz = ((m2.x * rx + m2.y * ry) + m2.z * rz) + m2.w;
              v3f ov = {hx * prev.winv + rt0, hy * prev.winv + rt1, hz * prev.winv + rt2};
              __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(v3u, ov), rsV, prev.voff,
                                                    __builtin_amdgcn_readfirstlane((int)(prev.f0 + ((R & 3) + 8 * (R >> 2))) * frameB), 2);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    });

    // the current item becomes the previous one
#pragma unroll
    for(int x = 0; x < 3; x++) accp[x] = acc[x];
#pragma unroll
    for(int i = 0; i < MAXW; i++)
    {
      jidxp[i] = jidx[i];
      jwp[i] = jw[i];
    }
    prev = cur;
    Abase = Abn;
    Bbase = Bbn;
  };

  do_item(t_begin, std::false_type{});
  for(int t = t_begin + 4; t < t_end; t += 4) do_item(t, std::true_type{});

  // tail: epilogue of the last item (its G' tile was written to LDS during its own chunks 17..21)
  static_for<16>([&](auto rr) {
    constexpr int R = decltype(rr)::value;
    const int fl = (R & 3) + 8 * (R >> 2) + 4 * half;
    const int64_t f = prev.f0 + fl;
    if(f < n)
    {
      const float rx = accp[0][R], ry = accp[1][R], rz = accp[2][R];
      if(WANT_REST && prev.rout)
      {
        float * o = prev.rout + f * V * 3;
        o[0] = rx;
        o[1] = ry;
        o[2] = rz;
      }
      if(prev.vout)
      {
        float4 m0 = make_float4(0.f, 0.f, 0.f, 0.f), m1 = m0, m2 = m0;
        const float * g = sG + fl * (NJ * 12);
#pragma unroll
        for(int i = 0; i < MAXW; i++)
        {
          const float4 * gj = reinterpret_cast<const float4 *>(g + jidxp[i] * 12);
          const float4 g0 = gj[0], g1 = gj[1], g2 = gj[2];
          const float w = jwp[i];
          m0.x += w * g0.x; m0.y += w * g0.y; m0.z += w * g0.z; m0.w += w * g0.w;
          m1.x += w * g1.x; m1.y += w * g1.y; m1.z += w * g1.z; m1.w += w * g1.w;
          m2.x += w * g2.x; m2.y += w * g2.y; m2.z += w * g2.z; m2.w += w * g2.w;
        }
        const float hx = ((m0.x * rx + m0.y * ry) + m0.z * rz) + m0.w;
        const float hy = ((m1.x * rx + m1.y * ry) + m1.z * rz) + m1.w;
        const float hz = ((m2.x * rx + m2.y * ry) + m2.z * rz) + m2.w;
        float * o = prev.vout + f * V * 3;
        o[0] = hx * prev.winv + sRoot[fl * 3 + 0];
        o[1] = hy * prev.winv + sRoot[fl * 3 + 1];
        o[2] = hz * prev.winv + sRoot[fl * 3 + 2];
      }
    }
  });
}

template<int MAXW, bool WANT_REST>
static hipError_t launch_p(const smplpp_model * m, int64_t n, const float * theta, const float * Gp_padded, float * verts, float * rest,
                           hipStream_t st)
{
  const int nft = (int)((n + 31) / 32);
  const int total = (int)m->VGn * nft;
  const int cus = device_cus(m->device);
  int blocks = (total + 3) / 4;
  if(blocks > cus) blocks = cus;
  blocks = (blocks + 7) & ~7; // the XCD-aware run assignment wants a multiple of 8 (idle runs exit at once)
  const int ipb = (total + blocks - 1) / blocks;
  const size_t shmem = sizeof(float) * 4 * P_LDS_WAVE;
  static PerDeviceOnce once;
  {
    hipError_t e = lds_opt_in(once, m->device, reinterpret_cast<const void *>(&skin_kernel_p<MAXW, WANT_REST>), (int)shmem);
    if(e != hipSuccess) return e;
  }
  skin_kernel_p<MAXW, WANT_REST><<<dim3(blocks), dim3(256), shmem, st>>>(m->ws.AT.as<float>(), m->ws.ldA, m->Bm, m->ldB, Gp_padded, theta,
                                                                        m->wIdx, m->wVal, m->wSum, verts, rest, m->ws.dummy.as<float>(), n, m->V,
                                                                        (int)m->VGn, nft, ipb, 0);
  return hipGetLastError();
}

hipError_t launch_skin_persistent(const smplpp_model * m, int64_t n, const float * theta, const float * Gp_padded, float * verts,
                                  float * rest, hipStream_t st)
{
  if(m->maxw == 4) return rest ? launch_p<4, true>(m, n, theta, Gp_padded, verts, rest, st) : launch_p<4, false>(m, n, theta, Gp_padded, verts, rest, st);
  if(m->maxw == 8) return rest ? launch_p<8, true>(m, n, theta, Gp_padded, verts, rest, st) : launch_p<8, false>(m, n, theta, Gp_padded, verts, rest, st);
  return hipErrorInvalidValue; // dense weights keep the first form
}
} // namespace smplpp_hip
