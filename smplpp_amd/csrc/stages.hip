// Stage-level entry points: the four FK stage classes of the reference on ARBITRARY inputs, in their native
// layouts (the hot path fuses them; these exist so that a caller of the individual classes — and the reference's
// own stage KATs in src/toolbox/Tester.cpp — have a drop-in).  Simple one-thread-per-output kernels; not the hot path.
#include "staging.h"

namespace smplpp_hip
{
int fk_device(smplpp_model * m, int64_t n, const float * beta, const float * theta, float * verts, float * joints,
              float * xforms44, float * rest, float * poserot, hipStream_t st, int range_slot, int * range_word = nullptr);
__device__ void rodrigues_dev_stage(float t0, float t1, float t2, float * R);

__device__ void rodrigues_dev_stage(float t0, float t1, float t2, float * R)
{
  const float eps = 1e-8f;
  float a0 = t0 + eps, a1 = t1 + eps, a2 = t2 + eps; // src/BlendShape.cpp:813-814
  float angle = sqrtf(a0 * a0 + a1 * a1 + a2 * a2);
  float k0 = t0 / angle, k1 = t1 / angle, k2 = t2 / angle;
  float K[9] = {0.0f, -k2, k1, k2, 0.0f, -k0, -k1, k0, 0.0f};
  float s = sinf(angle), c1 = 1.0f - cosf(angle);
  for(int r = 0; r < 3; r++)
    for(int c = 0; c < 3; c++)
    {
      float kk = K[r * 3 + 0] * K[0 * 3 + c] + K[r * 3 + 1] * K[1 * 3 + c] + K[r * 3 + 2] * K[2 * 3 + c];
      R[r * 3 + c] = ((r == c) ? 1.0f : 0.0f) + K[r * 3 + c] * s + kk * c1;
    }
}

// BlendShape::rodrigues (src/BlendShape.cpp:803-844): one thread per (frame, joint)
__global__ void stage_rodrigues_kernel(const float * __restrict__ theta24, float * __restrict__ rot, int64_t count)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= count) return;
  float R[9];
  rodrigues_dev_stage(theta24[i * 3], theta24[i * 3 + 1], theta24[i * 3 + 2], R);
  for(int q = 0; q < 9; q++) rot[i * 9 + q] = R[q];
}

// BlendShape::shapeBlend / poseBlend (:670-683, :762-765): one wavefront per output element (frame, vertex, x);
// the 207-term dot product is a 64-lane shuffle reduction.
__global__ __launch_bounds__(256) void stage_blend_kernel(const float * __restrict__ beta, const float * __restrict__ rot,
                                                           const float * __restrict__ S, const float * __restrict__ P,
                                                           float * __restrict__ Bs, float * __restrict__ Bp, int64_t V,
                                                           int64_t n)
{
  const int lane = threadIdx.x & 63;
  const int64_t o = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); // output index in [0, n*V*3)
  if(o >= n * V * 3) return;
  const int64_t f = o / (V * 3), e = o % (V * 3);
  float sp = 0.0f, ss = 0.0f;
  for(int k = lane; k < NP; k += 64)
  {
    int q = (k + 9) % 9;
    float c = rot[f * NJ * 9 + 9 + k] - ((q == 0 || q == 4 || q == 8) ? 1.0f : 0.0f); // linRotMin :884-892
    sp += c * P[e * NP + k];
  }
  if(lane < NB) ss = beta[f * NB + lane] * S[e * NB + lane];
  for(int off = 32; off > 0; off >>= 1)
  {
    sp += __shfl_down(sp, off, 64);
    ss += __shfl_down(ss, off, 64);
  }
  if(lane == 0)
  {
    if(Bp) Bp[o] = sp;
    if(Bs) Bs[o] = ss;
  }
}

// JointRegression::linearCombine (:551-565)
__global__ void stage_combine_kernel(const float * __restrict__ T, const float * __restrict__ Bs, const float * __restrict__ Bp,
                                     float * __restrict__ rest, int64_t V3, int64_t n)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n * V3) return;
  rest[i] = (T[i % V3] + Bs[i]) + Bp[i];
}

// JointRegression::jointRegress (:583-598): one block per (frame, joint, x); wavefront reductions over V
__global__ __launch_bounds__(256) void stage_regress_kernel(const float * __restrict__ T, const float * __restrict__ Jreg,
                                                             const float * __restrict__ Bs, float * __restrict__ joints,
                                                             int64_t V)
{
  const int x = blockIdx.x % 3, j = (blockIdx.x / 3) % NJ;
  const int64_t f = blockIdx.x / (3 * NJ);
  double acc = 0.0;
  for(int64_t v = threadIdx.x; v < V; v += blockDim.x)
    acc += (double)Jreg[(int64_t)j * V + v] * (double)(T[v * 3 + x] + Bs[(f * V + v) * 3 + x]);
  for(int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  __shared__ double part[4];
  if((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if(threadIdx.x == 0) joints[(f * NJ + j) * 3 + x] = (float)((part[0] + part[1]) + (part[2] + part[3]));
}

// WorldTransformation::transform on arbitrary 3x3 "rotations" (src/WorldTransformation.cpp:421-677): one thread per frame
__global__ void stage_world_kernel(const int32_t * __restrict__ parent, const float * __restrict__ joints,
                                   const float * __restrict__ rot, float * __restrict__ out, int64_t n)
{
  int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(f >= n) return;
  float G[NJ][12];
  const float * J = joints + f * NJ * 3;
  const float * R = rot + f * NJ * 9;
  for(int i = 0; i < NJ; i++)
  {
    int p = parent[i];
    float t[3];
    for(int x = 0; x < 3; x++) t[x] = (i == 0) ? J[x] : J[i * 3 + x] - J[p * 3 + x];
    for(int r = 0; r < 3; r++)
    {
      if(i == 0)
      {
        for(int c = 0; c < 3; c++) G[0][r * 4 + c] = R[r * 3 + c];
        G[0][r * 4 + 3] = t[r];
      }
      else
      {
        for(int c = 0; c < 3; c++)
          G[i][r * 4 + c] = G[p][r * 4 + 0] * R[i * 9 + c] + G[p][r * 4 + 1] * R[i * 9 + 3 + c] + G[p][r * 4 + 2] * R[i * 9 + 6 + c];
        G[i][r * 4 + 3] = G[p][r * 4 + 0] * t[0] + G[p][r * 4 + 1] * t[1] + G[p][r * 4 + 2] * t[2] + G[p][r * 4 + 3];
      }
    }
  }
  for(int i = 0; i < NJ; i++)
  {
    float * o = out + (f * NJ + i) * 16;
    for(int r = 0; r < 3; r++)
    {
      for(int c = 0; c < 3; c++) o[r * 4 + c] = G[i][r * 4 + c];
      o[r * 4 + 3] = G[i][r * 4 + 3] - (G[i][r * 4 + 0] * J[i * 3] + G[i][r * 4 + 1] * J[i * 3 + 1] + G[i][r * 4 + 2] * J[i * 3 + 2]);
    }
    o[12] = o[13] = o[14] = 0.0f;
    o[15] = 1.0f;
  }
}

// LinearBlendSkinning::skinning with general 4x4 transforms (src/LinearBlendSkinning.cpp:445-553)
__global__ void stage_skin_kernel(const float * __restrict__ W, const float * __restrict__ rest, const float * __restrict__ G,
                                  const float * __restrict__ root, float * __restrict__ verts, int64_t V, int64_t n)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n * V) return;
  const int64_t f = i / V, v = i % V;
  float M[16];
  for(int q = 0; q < 16; q++) M[q] = 0.0f;
  for(int j = 0; j < NJ; j++)
  {
    float w = W[v * NJ + j];
    const float * g = G + (f * NJ + j) * 16;
    for(int q = 0; q < 16; q++) M[q] += w * g[q];
  }
  const float * r = rest + i * 3;
  float h[4];
  for(int a = 0; a < 4; a++) h[a] = ((M[a * 4] * r[0] + M[a * 4 + 1] * r[1]) + M[a * 4 + 2] * r[2]) + M[a * 4 + 3];
  for(int x = 0; x < 3; x++) verts[i * 3 + x] = h[x] / h[3] + (root ? root[f * 3 + x] : 0.0f);
}
} // namespace smplpp_hip

using namespace smplpp_hip;

static int enter(int device, int space, const char * fn)
{
  int rc = check_space(space, fn);
  if(rc) return rc;
  int ndev = 0;
  rc = smplpp_device_count(&ndev);
  if(rc) return rc;
  if(device < 0 || device >= ndev) return fail(SMPLPP_ERR_INVALID, "Failed to fetch device index!");
  HIP_TRY(hipSetDevice(device));
  return SMPLPP_OK;
}

extern "C" int smplpp_stage_blend_shape(int device, int64_t V, int64_t n, const float * beta, const float * theta24,
                                        const float * S, const float * P, float * shape_blend, float * pose_blend,
                                        float * pose_rot, int space, void * stream)
{
  if(V <= 0 || n <= 0 || !beta || !theta24 || !S || !P)
    return fail(SMPLPP_ERR_INVALID, "Cannot blend shape-dependented shape!"); // src/BlendShape.cpp:679
  int rc = enter(device, space, "smplpp_stage_blend_shape");
  if(rc) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  In<float> b, t, s, p;
  Out<float> bs, bp, rot;
  float * rot_tmp = nullptr;
  HIP_TRY(b.init(beta, (size_t)n * NB, space, st));
  HIP_TRY(t.init(theta24, (size_t)n * NJ * 3, space, st));
  HIP_TRY(s.init(S, (size_t)V * 3 * NB, space, st));
  HIP_TRY(p.init(P, (size_t)V * 3 * NP, space, st));
  HIP_TRY(bs.init(shape_blend, (size_t)n * V * 3, space));
  HIP_TRY(bp.init(pose_blend, (size_t)n * V * 3, space));
  HIP_TRY(rot.init(pose_rot, (size_t)n * NJ * 9, space));
  float * rd = rot.d;
  if(!rd)
  {
    HIP_TRY(hipMalloc((void **)&rot_tmp, sizeof(float) * (size_t)n * NJ * 9));
    rd = rot_tmp;
  }
  stage_rodrigues_kernel<<<dim3((unsigned)((n * NJ + 255) / 256)), dim3(256), 0, st>>>(t.d, rd, n * NJ);
  if(bs.d || bp.d)
    stage_blend_kernel<<<dim3((unsigned)((n * V * 3 + 3) / 4)), dim3(256), 0, st>>>(b.d, rd, s.d, p.d, bs.d, bp.d, V, n);
  hipError_t e = hipGetLastError();
  if(e == hipSuccess) e = bs.finish(st);
  if(e == hipSuccess) e = bp.finish(st);
  if(e == hipSuccess) e = rot.finish(st);
  if(e == hipSuccess && (space == SMPLPP_HOST || rot_tmp)) e = hipStreamSynchronize(st);
  if(rot_tmp) (void)hipFree(rot_tmp);
  HIP_TRY(e);
  return SMPLPP_OK;
}

extern "C" int smplpp_stage_joint_regression(int device, int64_t V, int64_t n, const float * T, const float * Jreg,
                                             const float * shape_blend, const float * pose_blend, float * rest_shape,
                                             float * joints, int space, void * stream)
{
  if(V <= 0 || n <= 0 || !T || !Jreg || !shape_blend || (rest_shape && !pose_blend))
    return fail(SMPLPP_ERR_INVALID, "Cannot linearly combine shapes!"); // src/JointRegression.cpp:561
  int rc = enter(device, space, "smplpp_stage_joint_regression");
  if(rc) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  In<float> t, jr, bs, bp;
  Out<float> rest, jo;
  HIP_TRY(t.init(T, (size_t)V * 3, space, st));
  HIP_TRY(jr.init(Jreg, (size_t)NJ * V, space, st));
  HIP_TRY(bs.init(shape_blend, (size_t)n * V * 3, space, st));
  HIP_TRY(bp.init(pose_blend, (size_t)n * V * 3, space, st));
  HIP_TRY(rest.init(rest_shape, (size_t)n * V * 3, space));
  HIP_TRY(jo.init(joints, (size_t)n * NJ * 3, space));
  if(rest.d)
    stage_combine_kernel<<<dim3((unsigned)((n * V * 3 + 255) / 256)), dim3(256), 0, st>>>(t.d, bs.d, bp.d, rest.d, V * 3, n);
  if(jo.d) stage_regress_kernel<<<dim3((unsigned)(n * NJ * 3)), dim3(256), 0, st>>>(t.d, jr.d, bs.d, jo.d, V);
  hipError_t e = hipGetLastError();
  if(e == hipSuccess) e = rest.finish(st);
  if(e == hipSuccess) e = jo.finish(st);
  if(e == hipSuccess && space == SMPLPP_HOST) e = hipStreamSynchronize(st);
  HIP_TRY(e);
  return SMPLPP_OK;
}

extern "C" int smplpp_stage_world_transformation(int device, int64_t n, const int64_t * kintree, const float * joints,
                                                 const float * pose_rot, float * xforms, int space, void * stream)
{
  if(n <= 0 || !kintree || !joints || !pose_rot || !xforms)
    return fail(SMPLPP_ERR_INVALID, "Cannot transform bones locally!"); // src/WorldTransformation.cpp:512
  int rc = enter(device, space, "smplpp_stage_world_transformation");
  if(rc) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // the kinematic tree is always a host array here (it is a model constant, src/SMPL.cpp:603-605)
  int32_t parent[NJ];
  parent[0] = -1;
  for(int i = 1; i < NJ; i++)
  {
    if(kintree[i] < 0 || kintree[i] >= i) return fail(SMPLPP_ERR_INVALID, "Cannot set kinematic tree: parent(i) must precede i");
    parent[i] = (int32_t)kintree[i];
  }
  In<int32_t> par;
  In<float> j, r;
  Out<float> o;
  HIP_TRY(par.init(parent, NJ, SMPLPP_HOST, st));
  HIP_TRY(hipStreamSynchronize(st)); // parent[] is a stack array
  HIP_TRY(j.init(joints, (size_t)n * NJ * 3, space, st));
  HIP_TRY(r.init(pose_rot, (size_t)n * NJ * 9, space, st));
  HIP_TRY(o.init(xforms, (size_t)n * NJ * 16, space));
  stage_world_kernel<<<dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st>>>(par.d, j.d, r.d, o.d, n);
  hipError_t e = hipGetLastError();
  if(e == hipSuccess) e = o.finish(st);
  if(e == hipSuccess) e = hipStreamSynchronize(st); // temporaries (parent) are freed on return
  HIP_TRY(e);
  return SMPLPP_OK;
}

extern "C" int smplpp_stage_skinning(int device, int64_t V, int64_t n, const float * weights, const float * rest_shape,
                                     const float * xforms, const float * root_pos, float * verts, int space, void * stream)
{
  if(V <= 0 || n <= 0 || !weights || !rest_shape || !xforms || !verts)
    return fail(SMPLPP_ERR_INVALID, "Cannot convert Cartesian coordinates to homogeneous one!"); // LinearBlendSkinning.cpp:509
  int rc = enter(device, space, "smplpp_stage_skinning");
  if(rc) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  In<float> w, r, g, rp;
  Out<float> o;
  HIP_TRY(w.init(weights, (size_t)V * NJ, space, st));
  HIP_TRY(r.init(rest_shape, (size_t)n * V * 3, space, st));
  HIP_TRY(g.init(xforms, (size_t)n * NJ * 16, space, st));
  HIP_TRY(rp.init(root_pos, (size_t)n * 3, space, st));
  HIP_TRY(o.init(verts, (size_t)n * V * 3, space));
  stage_skin_kernel<<<dim3((unsigned)((n * V + 255) / 256)), dim3(256), 0, st>>>(w.d, r.d, g.d, rp.d, o.d, V, n);
  hipError_t e = hipGetLastError();
  if(e == hipSuccess) e = o.finish(st);
  if(e == hipSuccess && space == SMPLPP_HOST) e = hipStreamSynchronize(st);
  HIP_TRY(e);
  return SMPLPP_OK;
}
