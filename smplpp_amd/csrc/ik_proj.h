// proj_scan_kernel / proj_finish_kernel — node.cpp:970-1001: re-projection of the task points onto the posed mesh; and the small
// conversion / fill kernels of the host API.  Included by ik.hip only.
#pragma once
#include "ik_types.h"

namespace smplpp_hip
{
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
