// Mesh-side queries on posed vertices: SMPL::calcNormal / calcVertexNormal (/root/reference/src/SMPL.cpp:518-535) and
// the closest-point projection the IK loop does through igl::point_mesh_squared_distance (node/node.cpp:970-989).
#include "mesh_device.h"
#include "staging.h"

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
