// Device-side geometry helpers shared by mesh.hip and ik.hip (fp32, the reference's operation order).
#pragma once
#include "common.h"

namespace smplpp_hip
{
__device__ inline void cross3(const float * a, const float * b, float * c)
{
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

// torch::nn::functional::normalize: x / max(||x||, 1e-12)
__device__ inline void normalize3(float * x)
{
  float nrm = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  nrm = fmaxf(nrm, 1e-12f);
  x[0] /= nrm;
  x[1] /= nrm;
  x[2] /= nrm;
}

// SMPL::calcNormal (src/SMPL.cpp:518-525)
__device__ inline void face_normal_pts(const float * v0, const float * v1, const float * v2, float * nn)
{
  float a[3] = {v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2]};
  float b[3] = {v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2]};
  cross3(a, b, nn);
  normalize3(nn);
}
__device__ inline void face_normal_dev(const float * verts, const int32_t * faces, int face, float * nn)
{
  face_normal_pts(verts + 3 * faces[face * 3 + 0], verts + 3 * faces[face * 3 + 1], verts + 3 * faces[face * 3 + 2], nn);
}

// SMPL::calcVertexNormal (src/SMPL.cpp:527-535), uniform weights 1/deg (:630-639), ascending face id
__device__ inline void vertex_normal_dev(const float * verts, const int32_t * faces, const int32_t * adjOff,
                                         const int32_t * adjFace, int vertex, float * nn)
{
  const int b = adjOff[vertex], e = adjOff[vertex + 1];
  float sum = 0.0f;
  for(int q = b; q < e; q++) sum += 1.0f;
  const float w = 1.0f / sum;
  float acc[3] = {0.f, 0.f, 0.f};
  for(int q = b; q < e; q++)
  {
    float fn[3];
    face_normal_dev(verts, faces, adjFace[q], fn);
    acc[0] += w * fn[0];
    acc[1] += w * fn[1];
    acc[2] += w * fn[2];
  }
  normalize3(acc);
  nn[0] = acc[0];
  nn[1] = acc[1];
  nn[2] = acc[2];
}

// calcTriangleVertexWeights (include/smplpp/toolbox/GeometryUtils.h:42-52)
__device__ inline void triangle_weights_dev(const float * pos, const float * tri /*[3][3]*/, float * w)
{
  float d[3][3], c[3];
  for(int i = 0; i < 3; i++)
    for(int x = 0; x < 3; x++) d[i][x] = tri[i * 3 + x] - pos[x];
  for(int i = 0; i < 3; i++)
  {
    cross3(d[(i + 1) % 3], d[(i + 2) % 3], c);
    w[i] = sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
  }
  float s = (w[0] + w[1]) + w[2];
  w[0] /= s;
  w[1] /= s;
  w[2] /= s;
}

// closest point on triangle abc to p (Ericson, Real-Time Collision Detection 5.1.5)
__device__ inline void closest_on_triangle_dev(const float * p, const float * a, const float * b, const float * c, float * out)
{
  float ab[3], ac[3], ap[3], bp[3], cp[3];
  for(int x = 0; x < 3; x++)
  {
    ab[x] = b[x] - a[x];
    ac[x] = c[x] - a[x];
    ap[x] = p[x] - a[x];
    bp[x] = p[x] - b[x];
    cp[x] = p[x] - c[x];
  }
#define SMPLPP_D3(u, v) (u[0] * v[0] + u[1] * v[1] + u[2] * v[2])
  const float d1 = SMPLPP_D3(ab, ap), d2 = SMPLPP_D3(ac, ap);
  const float d3 = SMPLPP_D3(ab, bp), d4 = SMPLPP_D3(ac, bp);
  const float d5 = SMPLPP_D3(ab, cp), d6 = SMPLPP_D3(ac, cp);
#undef SMPLPP_D3
  if(d1 <= 0.0f && d2 <= 0.0f)
  {
    out[0] = a[0]; out[1] = a[1]; out[2] = a[2];
    return;
  }
  if(d3 >= 0.0f && d4 <= d3)
  {
    out[0] = b[0]; out[1] = b[1]; out[2] = b[2];
    return;
  }
  const float vc = d1 * d4 - d3 * d2;
  if(vc <= 0.0f && d1 >= 0.0f && d3 <= 0.0f)
  {
    const float v = d1 / (d1 - d3);
    for(int x = 0; x < 3; x++) out[x] = a[x] + v * ab[x];
    return;
  }
  if(d6 >= 0.0f && d5 <= d6)
  {
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2];
    return;
  }
  const float vb = d5 * d2 - d1 * d6;
  if(vb <= 0.0f && d2 >= 0.0f && d6 <= 0.0f)
  {
    const float w = d2 / (d2 - d6);
    for(int x = 0; x < 3; x++) out[x] = a[x] + w * ac[x];
    return;
  }
  const float va = d3 * d6 - d5 * d4;
  if(va <= 0.0f && (d4 - d3) >= 0.0f && (d5 - d6) >= 0.0f)
  {
    const float w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
    for(int x = 0; x < 3; x++) out[x] = b[x] + w * (c[x] - b[x]);
    return;
  }
  const float denom = 1.0f / (va + vb + vc);
  const float v = vb * denom, w = vc * denom;
  for(int x = 0; x < 3; x++) out[x] = a[x] + ab[x] * v + ac[x] * w;
}

// Block-wide (256 threads) closest point of ONE query against all F faces of one frame's mesh.
// A point above a convex edge projects ONTO that edge, so exact ties between the two faces sharing it are common
// (every mocap task with a normal offset); the rule here (the CPU checker uses the same) is: among faces whose squared
// distance is within 1e-6 (relative) of the minimum, the LOWEST face id wins.  (libigl's own tie order depends on its
// AABB traversal and is not pinned by any reference test; the closest POINT is the same either way.)
// Results written by thread 0.
// noinline: both passes below must round identically (two inlined copies may contract FMAs differently, and for a
// query ON the surface the squared distance is pure rounding noise).
// The ONE copy of the exact point-triangle distance (values in, values out: callers that already hold the triangle in
// registers do not gather it again).  .x = squared distance, .yzw = closest point.
__device__ __attribute__((noinline)) inline float4 tri_sqdist_vals(float a0, float a1, float a2, float b0, float b1, float b2, float c0,
                                                                      float c1, float c2, float p0, float p1, float p2)
{
  const float a[3] = {a0, a1, a2}, b[3] = {b0, b1, b2}, cc[3] = {c0, c1, c2}, p[3] = {p0, p1, p2};
  float c[3];
  closest_on_triangle_dev(p, a, b, cc, c);
  const float dx = c[0] - p[0], dy = c[1] - p[1], dz = c[2] - p[2];
  return make_float4(dx * dx + dy * dy + dz * dz, c[0], c[1], c[2]);
}
__device__ inline float tri_sqdist_dev(const float * verts, const int32_t * faces, int64_t f, const float * p, float * c)
{
  const float * a = verts + 3 * faces[f * 3];
  const float * b = verts + 3 * faces[f * 3 + 1];
  const float * cc = verts + 3 * faces[f * 3 + 2];
  const float4 r = tri_sqdist_vals(a[0], a[1], a[2], b[0], b[1], b[2], cc[0], cc[1], cc[2], p[0], p[1], p[2]);
  c[0] = r.y;
  c[1] = r.z;
  c[2] = r.w;
  return r.x;
}

// Conservative cull: every point of a triangle is at least |p - v0| - max(|v1 - v0|, |v2 - v0|) from p, so the face
// cannot beat the bound `lim` (a squared distance) when |p - v0| > sqrt(lim) + r.  1e-5 relative slack keeps the test
// on the safe side of rounding; culled faces are never the minimum nor inside the tie band.
__device__ inline bool tri_culled_v(const float * a, const float * b, const float * c, const float * p, float sqrt_lim)
{
  const float e1 = (b[0] - a[0]) * (b[0] - a[0]) + (b[1] - a[1]) * (b[1] - a[1]) + (b[2] - a[2]) * (b[2] - a[2]);
  const float e2 = (c[0] - a[0]) * (c[0] - a[0]) + (c[1] - a[1]) * (c[1] - a[1]) + (c[2] - a[2]) * (c[2] - a[2]);
  const float r = sqrtf(fmaxf(e1, e2));
  const float d0 = (p[0] - a[0]) * (p[0] - a[0]) + (p[1] - a[1]) * (p[1] - a[1]) + (p[2] - a[2]) * (p[2] - a[2]);
  const float reach = (sqrt_lim + r) * 1.00001f + 1e-7f;
  return d0 > reach * reach;
}

// The scan is latency-bound (index load -> dependent vertex gathers), so faces are taken CP_BATCH at a time with all
// of a batch's loads issued before any of its tests.
constexpr int CP_BATCH = 6;
template<int NB>
struct TriBatchT
{
  float v[NB][9];
  bool valid[NB];
};
typedef TriBatchT<CP_BATCH> TriBatch;
template<int NB>
__device__ inline void load_tri_batch(const float * verts, const int32_t * faces, int64_t F, int64_t base, int stride, TriBatchT<NB> & t)
{
  int id[NB][3];
#pragma unroll
  for(int b = 0; b < NB; b++)
  {
    const int64_t f = base + (int64_t)b * stride;
    t.valid[b] = f < F;
    const int64_t ff = t.valid[b] ? f : 0;
    id[b][0] = faces[ff * 3];
    id[b][1] = faces[ff * 3 + 1];
    id[b][2] = faces[ff * 3 + 2];
  }
#pragma unroll
  for(int b = 0; b < NB; b++)
#pragma unroll
    for(int c = 0; c < 3; c++)
    {
      const float * src = verts + 3 * id[b][c];
      t.v[b][c * 3] = src[0];
      t.v[b][c * 3 + 1] = src[1];
      t.v[b][c * 3 + 2] = src[2];
    }
}

// `hint_face` (>= 0): a face known to be near the query (the task's current face); its exact distance seeds the bound
// so that almost every other face is culled by the sphere test.
__device__ inline void closest_point_block(const float * verts, const int32_t * faces, int64_t F, const float * point,
                                           int64_t * face_out, float * closest_out, float * sq_out, int hint_face = -1)
{
  const float p[3] = {point[0], point[1], point[2]};
  __shared__ float s_d[4];
  __shared__ int s_f[4];
  __shared__ float s_min;
  __shared__ int s_argmin;
  const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  // pass 1: minimum squared distance (and its face, the fallback if the tie band comes up empty, e.g. NaN input)
  float best = INFINITY;
  int bf = 0;
  if(hint_face >= 0 && (int64_t)hint_face < F)
  {
    float c[3];
    const float d = tri_sqdist_dev(verts, faces, hint_face, p, c);
    if(d == d) // not NaN
    {
      best = d;
      bf = hint_face;
    }
  }
  float sq = sqrtf(best);
  for(int64_t base = threadIdx.x; base < F; base += (int64_t)blockDim.x * CP_BATCH)
  {
    TriBatch t;
    load_tri_batch(verts, faces, F, base, blockDim.x, t);
#pragma unroll
    for(int b = 0; b < CP_BATCH; b++)
    {
      const int64_t f = base + (int64_t)b * blockDim.x;
      if(!t.valid[b]) continue;
      if(best < INFINITY && tri_culled_v(t.v[b], t.v[b] + 3, t.v[b] + 6, p, sq)) continue;
      float c[3];
      const float d = tri_sqdist_dev(verts, faces, f, p, c);
      if(d < best || (d == best && (int)f < bf))
      {
        best = d;
        bf = (int)f;
        sq = sqrtf(best);
      }
    }
  }
  for(int off = 32; off > 0; off >>= 1)
  {
    const float od = __shfl_down(best, off, 64);
    const int of = __shfl_down(bf, off, 64);
    if(od < best || (od == best && of < bf))
    {
      best = od;
      bf = of;
    }
  }
  if((threadIdx.x & 63) == 0)
  {
    s_d[wave] = best;
    s_f[wave] = bf;
  }
  __syncthreads();
  if(threadIdx.x == 0)
  {
    int w = 0;
    for(int i = 1; i < nw; i++)
      if(s_d[i] < s_d[w] || (s_d[i] == s_d[w] && s_f[i] < s_f[w])) w = i;
    s_min = s_d[w];
    s_argmin = s_f[w];
  }
  __syncthreads();
  // pass 2: lowest face id within the tie band
  const float thr = s_min * (1.0f + 1e-6f) + 1e-12f;
  const float sq_thr = sqrtf(thr);
  int cf = 0x7fffffff;
  for(int64_t base = threadIdx.x; base < F && (int)base < cf; base += (int64_t)blockDim.x * CP_BATCH)
  {
    TriBatch t;
    load_tri_batch(verts, faces, F, base, blockDim.x, t);
#pragma unroll
    for(int b = 0; b < CP_BATCH; b++)
    {
      const int64_t f = base + (int64_t)b * blockDim.x;
      if(!t.valid[b] || (int)f >= cf) continue; // ids ascend within a thread
      if(thr < INFINITY && tri_culled_v(t.v[b], t.v[b] + 3, t.v[b] + 6, p, sq_thr)) continue;
      float c[3];
      if(tri_sqdist_dev(verts, faces, f, p, c) <= thr) cf = (int)f;
    }
  }
  for(int off = 32; off > 0; off >>= 1)
  {
    const int of = __shfl_down(cf, off, 64);
    cf = of < cf ? of : cf;
  }
  __syncthreads(); // s_f is reused
  if((threadIdx.x & 63) == 0) s_f[wave] = cf;
  __syncthreads();
  if(threadIdx.x == 0)
  {
    int b = s_f[0];
    for(int i = 1; i < nw; i++) b = s_f[i] < b ? s_f[i] : b;
    if(b < 0 || (int64_t)b >= F) b = s_argmin; // empty band: never index out of range
    float c[3];
    const float d = tri_sqdist_dev(verts, faces, b, p, c);
    if(face_out) *face_out = b;
    if(closest_out)
    {
      closest_out[0] = c[0]; closest_out[1] = c[1]; closest_out[2] = c[2];
    }
    if(sq_out) *sq_out = d;
  }
  __syncthreads();
}
} // namespace smplpp_hip
