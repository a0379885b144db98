// Mesh-side queries on posed vertices: SMPL::calcNormal / calcVertexNormal (/root/reference/src/SMPL.cpp:518-535) and
// the closest-point projection the IK loop does through igl::point_mesh_squared_distance (node/node.cpp:970-989).
#include "mesh_device.h"
#include "staging.h"

#include <cmath>

namespace smplpp_hip
{
__global__ void face_normals_kernel(const float * __restrict__ verts, const int32_t * __restrict__ faces,
                                    const int64_t * __restrict__ ids, float * __restrict__ out, int64_t V, int64_t count,
                                    int64_t n)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n * count) return;
  const int64_t f = i / count;
  float nn[3];
  face_normal_dev(verts + f * V * 3, faces, (int)ids[i % count], nn);
  out[i * 3] = nn[0];
  out[i * 3 + 1] = nn[1];
  out[i * 3 + 2] = nn[2];
}

__global__ void vertex_normals_kernel(const float * __restrict__ verts, const int32_t * __restrict__ faces,
                                      const int32_t * __restrict__ adjOff, const int32_t * __restrict__ adjFace,
                                      const int64_t * __restrict__ ids, float * __restrict__ out, int64_t V, int64_t count,
                                      int64_t n)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n * count) return;
  const int64_t f = i / count;
  float nn[3];
  vertex_normal_dev(verts + f * V * 3, faces, adjOff, adjFace, (int)ids[i % count], nn);
  out[i * 3] = nn[0];
  out[i * 3 + 1] = nn[1];
  out[i * 3 + 2] = nn[2];
}

// one block per (frame, query point): every thread scans F/256 faces, then a (distance, face id) min-reduction with
// ties resolved to the lowest face id.
__global__ __launch_bounds__(256) void closest_points_kernel(const float * __restrict__ verts, const int32_t * __restrict__ faces,
                                                              const float * __restrict__ points, int64_t * __restrict__ face_out,
                                                              float * __restrict__ closest_out, float * __restrict__ sq_out,
                                                              int64_t V, int64_t F, int64_t K)
{
  const int64_t f = blockIdx.x / K;
  closest_point_block(verts + f * V * 3, faces, F, points + (int64_t)blockIdx.x * 3, face_out ? face_out + blockIdx.x : nullptr,
                      closest_out ? closest_out + (int64_t)blockIdx.x * 3 : nullptr, sq_out ? sq_out + blockIdx.x : nullptr);
}

// whole mesh: one thread per (frame, vertex)
__global__ void mesh_vertex_normals_kernel(const float * __restrict__ verts, const int32_t * __restrict__ faces,
                                           const int32_t * __restrict__ adjOff, const int32_t * __restrict__ adjFace,
                                           float * __restrict__ out, int64_t V, int64_t n)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n * V) return;
  const int64_t f = i / V;
  float nn[3];
  vertex_normal_dev(verts + f * V * 3, faces, adjOff, adjFace, (int)(i % V), nn);
  out[i * 3] = nn[0];
  out[i * 3 + 1] = nn[1];
  out[i * 3 + 2] = nn[2];
}

// ---- sweep grid (node/node.cpp:1023-1073, toolbox/GridUtils.hpp:26-61)
// bounding box of one frame's vertices: one block, out[0..2] = min, out[3..5] = max
__global__ __launch_bounds__(1024) void bounds_kernel(const float * __restrict__ verts, int64_t V, float * __restrict__ out)
{
  __shared__ float smin[3][16], smax[3][16];
  float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
  for(int64_t v = threadIdx.x; v < V; v += blockDim.x)
    for(int x = 0; x < 3; x++)
    {
      const float c = verts[v * 3 + x];
      // (fminf / fmaxf drop NaNs: a non-finite coordinate is turned into an infinite bound so that the host's check fires)
      const bool bad = !(__builtin_fabsf(c) <= 3.0e38f);
      mn[x] = bad ? -__builtin_inff() : fminf(mn[x], c);
      mx[x] = bad ? __builtin_inff() : fmaxf(mx[x], c);
    }
  for(int x = 0; x < 3; x++)
    for(int off = 32; off > 0; off >>= 1)
    {
      mn[x] = fminf(mn[x], __shfl_down(mn[x], off, 64));
      mx[x] = fmaxf(mx[x], __shfl_down(mx[x], off, 64));
    }
  if((threadIdx.x & 63) == 0)
    for(int x = 0; x < 3; x++)
    {
      smin[x][threadIdx.x >> 6] = mn[x];
      smax[x][threadIdx.x >> 6] = mx[x];
    }
  __syncthreads();
  if(threadIdx.x < 3)
  {
    float a = smin[threadIdx.x][0], b = smax[threadIdx.x][0];
    for(int w = 1; w < (int)(blockDim.x >> 6); w++)
    {
      a = fminf(a, smin[threadIdx.x][w]);
      b = fmaxf(b, smax[threadIdx.x][w]);
    }
    out[threadIdx.x] = a;
    out[3 + threadIdx.x] = b;
  }
}

// generalized winding number of the closed mesh at every grid point (igl::winding_number as called at node/node.cpp:1052):
// w(p) = sum_f Omega_f(p) / (4 pi), Omega_f = 2 atan2(a . (b x c), |a||b||c| + (a.b)|c| + (b.c)|a| + (c.a)|b|) with
// a, b, c = the face's vertices minus p.  One thread per grid point; the faces pass through LDS 256 at a time; fp32 terms
// like the reference's float matrices, fp64 sum.  Cell order: x outermost, z innermost (the reference's triple loop).
__global__ __launch_bounds__(256) void winding_kernel(const float * __restrict__ verts, const int32_t * __restrict__ faces, int64_t F,
                                                      int gx0, int gy0, int gz0, int ny, int nz, int64_t cells, float scale,
                                                      float * __restrict__ winding, uint8_t * __restrict__ inside)
{
  __shared__ float tri[256][9];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < cells;
  const int64_t ii = live ? i : 0;
  const float px = scale * (float)(gx0 + (int)(ii / ((int64_t)ny * nz)));
  const float py = scale * (float)(gy0 + (int)((ii / nz) % ny));
  const float pz = scale * (float)(gz0 + (int)(ii % nz));
  double acc = 0.0;
  for(int64_t f0 = 0; f0 < F; f0 += 256)
  {
    const int64_t f = f0 + threadIdx.x;
    if(f < F)
#pragma unroll
      for(int q = 0; q < 3; q++)
      {
        const float * v = verts + 3 * (int64_t)faces[f * 3 + q];
        tri[threadIdx.x][q * 3 + 0] = v[0];
        tri[threadIdx.x][q * 3 + 1] = v[1];
        tri[threadIdx.x][q * 3 + 2] = v[2];
      }
    __syncthreads();
    const int cnt = (int)(F - f0 < 256 ? F - f0 : 256);
    float part = 0.0f;
    for(int t = 0; t < cnt; t++)
    {
      const float ax = tri[t][0] - px, ay = tri[t][1] - py, az = tri[t][2] - pz;
      const float bx = tri[t][3] - px, by = tri[t][4] - py, bz = tri[t][5] - pz;
      const float cx = tri[t][6] - px, cy = tri[t][7] - py, cz = tri[t][8] - pz;
      const float la = sqrtf(ax * ax + ay * ay + az * az), lb = sqrtf(bx * bx + by * by + bz * bz), lc = sqrtf(cx * cx + cy * cy + cz * cz);
      const float det = ax * (by * cz - bz * cy) + ay * (bz * cx - bx * cz) + az * (bx * cy - by * cx);
      const float den = la * lb * lc + (ax * bx + ay * by + az * bz) * lc + (bx * cx + by * cy + bz * cz) * la + (cx * ax + cy * ay + cz * az) * lb;
      part += atan2f(det, den);
    }
    acc += (double)part;
    __syncthreads();
  }
  if(!live) return;
  const float w = (float)(acc / (2.0 * 3.14159265358979323846));
  if(winding) winding[i] = w;
  if(inside) inside[i] = w > 0.5f ? 1 : 0;
}

int closest_points_device(const smplpp_model * m, int64_t n, const float * verts, int64_t K, const float * points,
                          int64_t * face, float * closest, float * sqdist, hipStream_t st)
{
  closest_points_kernel<<<dim3((unsigned)(n * K)), dim3(256), 0, st>>>(verts, m->faces, points, face, closest, sqdist, m->V,
                                                                      m->F, K);
  HIP_TRY(hipGetLastError());
  return SMPLPP_OK;
}
} // namespace smplpp_hip

using namespace smplpp_hip;

static int normals_common(smplpp_model * m, int64_t n, const float * verts, int64_t count, const int64_t * ids,
                          float * normals, int space, void * stream, bool vertex)
{
  const char * fn = vertex ? "smplpp_vertex_normals" : "smplpp_face_normals";
  if(!m || n <= 0 || count <= 0 || !verts || !ids || !normals) return fail(SMPLPP_ERR_INVALID, std::string(fn) + ": bad argument");
  int rc = check_space(space, fn);
  if(rc) return rc;
  HIP_TRY(hipSetDevice(m->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  // ids are validated on the host when they are host memory
  if(space == SMPLPP_HOST)
    for(int64_t i = 0; i < count; i++)
      if(ids[i] < 0 || ids[i] >= (vertex ? m->V : m->F)) return fail(SMPLPP_ERR_INVALID, std::string(fn) + ": id out of range");
  In<float> v;
  In<int64_t> id;
  Out<float> o;
  HIP_TRY(v.init(verts, (size_t)n * m->V * 3, space, st));
  HIP_TRY(id.init(ids, (size_t)count, space, st));
  HIP_TRY(o.init(normals, (size_t)n * count * 3, space));
  unsigned grid = (unsigned)((n * count + 127) / 128);
  if(vertex)
    vertex_normals_kernel<<<dim3(grid), dim3(128), 0, st>>>(v.d, m->faces, m->adjOff, m->adjFace, id.d, o.d, m->V, count, n);
  else
    face_normals_kernel<<<dim3(grid), dim3(128), 0, st>>>(v.d, m->faces, id.d, o.d, m->V, count, n);
  hipError_t e = hipGetLastError();
  if(e == hipSuccess) e = o.finish(st);
  if(e == hipSuccess && space == SMPLPP_HOST) e = hipStreamSynchronize(st);
  HIP_TRY(e);
  return SMPLPP_OK;
}

extern "C" int smplpp_face_normals(smplpp_model * m, int64_t n, const float * verts, int64_t count, const int64_t * face_ids,
                                   float * normals, int space, void * stream)
{
  return normals_common(m, n, verts, count, face_ids, normals, space, stream, false);
}

extern "C" int smplpp_vertex_normals(smplpp_model * m, int64_t n, const float * verts, int64_t count,
                                     const int64_t * vertex_ids, float * normals, int space, void * stream)
{
  return normals_common(m, n, verts, count, vertex_ids, normals, space, stream, true);
}

extern "C" int smplpp_closest_points(smplpp_model * m, int64_t n, const float * verts, int64_t K, const float * points,
                                     int64_t * face, float * closest, float * sqdist, int space, void * stream)
{
  if(!m || n <= 0 || K <= 0 || !verts || !points) return fail(SMPLPP_ERR_INVALID, "smplpp_closest_points: bad argument");
  if(m->F <= 0) return fail(SMPLPP_ERR_INVALID, "smplpp_closest_points: model has no faces");
  int rc = check_space(space, "smplpp_closest_points");
  if(rc) return rc;
  HIP_TRY(hipSetDevice(m->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  In<float> v, p;
  Out<int64_t> fo;
  Out<float> co, so;
  HIP_TRY(v.init(verts, (size_t)n * m->V * 3, space, st));
  HIP_TRY(p.init(points, (size_t)n * K * 3, space, st));
  HIP_TRY(fo.init(face, (size_t)n * K, space));
  HIP_TRY(co.init(closest, (size_t)n * K * 3, space));
  HIP_TRY(so.init(sqdist, (size_t)n * K, space));
  rc = closest_points_device(m, n, v.d, K, p.d, fo.d, co.d, so.d, st);
  if(rc) return rc;
  hipError_t e = fo.finish(st);
  if(e == hipSuccess) e = co.finish(st);
  if(e == hipSuccess) e = so.finish(st);
  if(e == hipSuccess && space == SMPLPP_HOST) e = hipStreamSynchronize(st);
  HIP_TRY(e);
  return SMPLPP_OK;
}

// SMPL::calcVertexNormal (src/SMPL.cpp:527-535) for every vertex of every frame: normals [n,V,3]
extern "C" int smplpp_mesh_vertex_normals(smplpp_model * m, int64_t n, const float * verts, float * normals, int space, void * stream)
{
  if(!m || n <= 0 || !verts || !normals) return fail(SMPLPP_ERR_INVALID, "smplpp_mesh_vertex_normals: bad argument");
  if(m->F <= 0) return fail(SMPLPP_ERR_INVALID, "smplpp_mesh_vertex_normals: model has no faces");
  int rc = check_space(space, "smplpp_mesh_vertex_normals");
  if(rc) return rc;
  HIP_TRY(hipSetDevice(m->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  In<float> v;
  Out<float> o;
  HIP_TRY(v.init(verts, (size_t)n * m->V * 3, space, st));
  HIP_TRY(o.init(normals, (size_t)n * m->V * 3, space));
  mesh_vertex_normals_kernel<<<dim3((unsigned)((n * m->V + 255) / 256)), dim3(256), 0, st>>>(v.d, m->faces, m->adjOff, m->adjFace, o.d, m->V, n);
  hipError_t e = hipGetLastError();
  if(e == hipSuccess) e = o.finish(st);
  if(e == hipSuccess && space == SMPLPP_HOST) e = hipStreamSynchronize(st);
  HIP_TRY(e);
  return SMPLPP_OK;
}

// The sweep grid of node/node.cpp:1023-1073 for one frame of posed vertices: grid cells of GRID_SCALE = 2.5 cm
// (toolbox/GridUtils.hpp:28) from floor(min / scale) to ceil(max / scale) per axis, the generalized winding number of the
// mesh at every cell position, and the cells the reference enters into g_sweepGridList (winding number > 0.5).
extern "C" int smplpp_sweep_grid(smplpp_model * m, const float * verts, int32_t * grid_min, int32_t * grid_num, int64_t cap,
                                 float * winding, uint8_t * inside, int64_t * cells, int space, void * stream)
{
  if(!m || !verts || !grid_min || !grid_num || !cells || cap < 0) return fail(SMPLPP_ERR_INVALID, "smplpp_sweep_grid: bad argument");
  if(m->F <= 0) return fail(SMPLPP_ERR_INVALID, "smplpp_sweep_grid: model has no faces");
  int rc = check_space(space, "smplpp_sweep_grid");
  if(rc) return rc;
  HIP_TRY(hipSetDevice(m->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  In<float> v;
  HIP_TRY(v.init(verts, (size_t)m->V * 3, space, st));
  DevBuf bb;
  HIP_TRY(bb.reserve(sizeof(float) * 6));
  bounds_kernel<<<dim3(1), dim3(1024), 0, st>>>(v.d, m->V, bb.as<float>());
  float h[6];
  hipError_t e = hipGetLastError();
  if(e == hipSuccess) e = hipMemcpyAsync(h, bb.p, sizeof(h), hipMemcpyDeviceToHost, st);
  if(e == hipSuccess) e = hipStreamSynchronize(st);
  bb.release();
  HIP_TRY(e);
  const float scale = 0.025f; // GRID_SCALE
  int64_t total = 1;
  int g0[3], gn[3];
  for(int x = 0; x < 3; x++)
  {
    if(!std::isfinite(h[x]) || !std::isfinite(h[3 + x])) return fail(SMPLPP_ERR_NUMERIC, "smplpp_sweep_grid: non-finite vertices");
    // a grid of 2.5 cm cells around a body: anything beyond +-25 km is not a posed mesh (and would not fit an int / the product)
    if(std::fabs(h[x]) > 2.5e4f || std::fabs(h[3 + x]) > 2.5e4f) return fail(SMPLPP_ERR_NUMERIC, "smplpp_sweep_grid: vertices out of range");
    g0[x] = (int)std::floor(h[x] / scale);           // getGridIdxFloor (GridUtils.hpp:46-50)
    const int g1 = (int)std::ceil(h[3 + x] / scale); // getGridIdxCeil (:56-60)
    gn[x] = g1 - g0[x] + 1;
    grid_min[x] = g0[x];
    grid_num[x] = gn[x];
    total *= gn[x]; // (gn <= 2e6 + 2 each: the product of three fits int64)
  }
  *cells = total;
  const int64_t todo = total < cap ? total : cap;
  if(todo <= 0 || (!winding && !inside)) return SMPLPP_OK;
  Out<float> wo;
  Out<uint8_t> io;
  HIP_TRY(wo.init(winding, (size_t)todo, space));
  HIP_TRY(io.init(inside, (size_t)todo, space));
  winding_kernel<<<dim3((unsigned)((todo + 255) / 256)), dim3(256), 0, st>>>(v.d, m->faces, m->F, g0[0], g0[1], g0[2], gn[1], gn[2], todo, scale,
                                                                            wo.d, io.d);
  e = hipGetLastError();
  if(e == hipSuccess) e = wo.finish(st);
  if(e == hipSuccess) e = io.finish(st);
  if(e == hipSuccess && space == SMPLPP_HOST) e = hipStreamSynchronize(st);
  HIP_TRY(e);
  return SMPLPP_OK;
}
