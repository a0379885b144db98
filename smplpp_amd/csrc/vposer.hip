// VPoser v2 decoder (/root/reference/src/VPoser.cpp) forward + Jacobian, one workgroup per frame.
//
//   z[32] -> Linear(32,512) -> LeakyReLU(0.01) -> Dropout(eval: identity) -> Linear(512,512) -> LeakyReLU
//         -> Linear(512,126) -> view[21,3,2] -> Gram-Schmidt (ContinousRotReprDecoderImpl::forward :129-141)
//         -> convertRotMatToAxisAngle (:25-120) -> [21,3]
// The reference differentiates this with autograd inside the IK loop (node/node.cpp:761-772); here d(out)/dz [63,32]
// is carried forward through the MLP next to the activations (32 tangent columns per row, held in LDS) and through
// the rotation tail with 6-wide dual numbers, so the same branches are taken for value and derivative.
// Weights are stored transposed ([in][out]) so that consecutive lanes read consecutive outputs.
#include "staging.h"

#include <cfloat>

struct smplpp_vposer
{
  int device = 0;
  float *w0t = nullptr, *b0 = nullptr, *w1t = nullptr, *b1 = nullptr, *w2t = nullptr, *b2 = nullptr;
};

namespace smplpp_hip
{
constexpr int LAT = SMPLPP_LATENT_DIM; // 32
constexpr int HID = 512;               // VPoser.h hiddenDim_
constexpr int OUT6 = 126;              // 6 * 21

struct D6
{
  float v;
  float d[6];
};
__device__ inline D6 mk(float v)
{
  D6 r;
  r.v = v;
  for(int i = 0; i < 6; i++) r.d[i] = 0.f;
  return r;
}
__device__ inline D6 operator+(const D6 & a, const D6 & b)
{
  D6 r;
  r.v = a.v + b.v;
  for(int i = 0; i < 6; i++) r.d[i] = a.d[i] + b.d[i];
  return r;
}
__device__ inline D6 operator-(const D6 & a, const D6 & b)
{
  D6 r;
  r.v = a.v - b.v;
  for(int i = 0; i < 6; i++) r.d[i] = a.d[i] - b.d[i];
  return r;
}
__device__ inline D6 operator*(const D6 & a, const D6 & b)
{
  D6 r;
  r.v = a.v * b.v;
  for(int i = 0; i < 6; i++) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
  return r;
}
__device__ inline D6 operator/(const D6 & a, const D6 & b)
{
  D6 r;
  r.v = a.v / b.v;
  for(int i = 0; i < 6; i++) r.d[i] = (a.d[i] - r.v * b.d[i]) / b.v;
  return r;
}
__device__ inline D6 operator*(float s, const D6 & a)
{
  D6 r;
  r.v = s * a.v;
  for(int i = 0; i < 6; i++) r.d[i] = s * a.d[i];
  return r;
}
__device__ inline D6 operator+(const D6 & a, float s)
{
  D6 r = a;
  r.v += s;
  return r;
}
__device__ inline D6 neg(const D6 & a)
{
  return -1.0f * a;
}
__device__ inline D6 dsqrt(const D6 & a)
{
  D6 r;
  r.v = sqrtf(a.v);
  for(int i = 0; i < 6; i++) r.d[i] = a.d[i] / (2.0f * r.v);
  return r;
}
__device__ inline D6 dacos(const D6 & a)
{
  D6 r;
  r.v = acosf(a.v);
  const float g = -1.0f / sqrtf(1.0f - a.v * a.v);
  for(int i = 0; i < 6; i++) r.d[i] = g * a.d[i];
  return r;
}
__device__ inline D6 dsin(const D6 & a)
{
  D6 r;
  r.v = sinf(a.v);
  const float c = cosf(a.v);
  for(int i = 0; i < 6; i++) r.d[i] = c * a.d[i];
  return r;
}
// torch::nn::functional::normalize of a 3-vector: x / max(||x||, 1e-12) (clamp_min passes no gradient when active)
__device__ inline void dnormalize3(const D6 * x, D6 * o)
{
  D6 n2 = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
  D6 n = dsqrt(n2);
  if(n.v < 1e-12f) n = mk(1e-12f);
  for(int i = 0; i < 3; i++) o[i] = x[i] / n;
}

// convertRotMatToAxisAngle (src/VPoser.cpp:25-120) on one matrix, value + derivative
__device__ inline void rotmat_to_aa(const D6 R[3][3], D6 aa[3])
{
  const float eps = FLT_EPSILON;
  const float epsSqrt = sqrtf(eps);
  const float epsSqrt2 = sqrtf(epsSqrt);
  const float kPi = 3.14159265358979323846f;
  D6 trace = R[0][0] + R[1][1] + R[2][2];
  D6 theta = dacos((float)((1.0 - (double)eps) * 0.5) * (trace + (-1.0f))); // :41
  D6 w[3] = {R[2][1] - R[1][2], R[0][2] - R[2][0], R[1][0] - R[0][1]};      // :43-49
  if(1.0f + trace.v < epsSqrt2) // near pi (:53-103)
  {
    D6 tn2[3];
    D6 one_m_tr = neg(trace) + 1.0f, three_m_tr = neg(trace) + 3.0f;
    for(int i = 0; i < 3; i++)
    {
      D6 s = (2.0f * R[i][i] + one_m_tr) / three_m_tr; // :54-56
      tn2[i] = dsqrt(s + eps) * theta;                   // :60
    }
    if(theta.v > kPi - 1e-4f) // :62-94
    {
      if(tn2[0].v > 0.0f)
      {
        if(R[0][1].v + R[1][0].v < 0.0f) tn2[1] = neg(tn2[1]);
        if(R[0][2].v + R[2][0].v < 0.0f) tn2[2] = neg(tn2[2]);
      }
      else if(tn2[1].v > 0.0f)
      {
        if(R[1][2].v + R[2][1].v < 0.0f) tn2[2] = neg(tn2[2]);
      }
    }
    else // :96-99
    {
      for(int i = 0; i < 3; i++)
        if(!(w[i].v >= 0.0f)) tn2[i] = neg(tn2[i]);
    }
    for(int i = 0; i < 3; i++) aa[i] = tn2[i];
  }
  else if(fabsf(3.0f - trace.v) < epsSqrt) // near zero: Taylor (:105-111)
  {
    D6 t2 = theta * theta;
    D6 f = (1.0f / 6.0f) * t2 + (7.0f / 360.0f) * (t2 * t2) + 1.0f;
    for(int i = 0; i < 3; i++) aa[i] = 0.5f * (w[i] * f);
  }
  else // :112-116
  {
    D6 f = theta / (2.0f * dsin(theta));
    for(int i = 0; i < 3; i++) aa[i] = w[i] * f;
  }
}

// ContinousRotReprDecoderImpl::forward (:129-141) on one joint's 6 numbers (view [3,2]) then -> axis-angle
__device__ inline void sixd_to_aa(const float * o6, float * aa_out, float * jac36 /*[3][6]*/)
{
  D6 c1[3], c2[3];
  for(int r = 0; r < 3; r++)
  {
    c1[r] = mk(o6[2 * r]);
    c1[r].d[2 * r] = 1.0f;
    c2[r] = mk(o6[2 * r + 1]);
    c2[r].d[2 * r + 1] = 1.0f;
  }
  D6 a1[3], a2[3], t[3];
  dnormalize3(c1, a1);
  D6 dot = a1[0] * c2[0] + a1[1] * c2[1] + a1[2] * c2[2];
  for(int r = 0; r < 3; r++) t[r] = c2[r] - dot * a1[r];
  dnormalize3(t, a2);
  D6 a3[3] = {a1[1] * a2[2] - a1[2] * a2[1], a1[2] * a2[0] - a1[0] * a2[2], a1[0] * a2[1] - a1[1] * a2[0]};
  D6 R[3][3];
  for(int r = 0; r < 3; r++)
  {
    R[r][0] = a1[r];
    R[r][1] = a2[r];
    R[r][2] = a3[r];
  }
  D6 aa[3];
  rotmat_to_aa(R, aa);
  for(int i = 0; i < 3; i++)
  {
    aa_out[i] = aa[i].v;
    if(jac36)
      for(int q = 0; q < 6; q++) jac36[i * 6 + q] = aa[i].d[q];
  }
}

// grid = n frames, block = 256.  LDS: a1/D1 then a2/D2, [512][VS] floats each (33 used: 32 tangent columns + the
// activation in column 32; the stride VS = 36 keeps every row 16-byte aligned so a row is read with nine ds_read_b128 —
// every lane reads the same row (broadcast), and 33 scalar LDS reads per k made the LDS issue slots the bottleneck);
// the layer-2 output [126][33] reuses the a1/D1 region.
constexpr int VS = 36;
__global__ __launch_bounds__(256) void vposer_kernel(const float * __restrict__ z, int64_t z_stride, const float * __restrict__ w0t,
                                                     const float * __restrict__ b0, const float * __restrict__ w1t,
                                                     const float * __restrict__ b1, const float * __restrict__ w2t,
                                                     const float * __restrict__ b2, float * __restrict__ out, int64_t out_stride,
                                                     float * __restrict__ jac, int want_jac)
{
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float * L1 = lds;                  // [512][VS]
  float * L2 = lds + HID * VS;       // [512][VS]
  float * sz = L2 + HID * VS;        // [32]
  float * so = L1;                   // [126][33]  (a1/D1 is dead once layer 1 is done)
  float * sSlope = sz + LAT;         // [512] LeakyReLU slopes of layer 1
  const int64_t f = blockIdx.x;
  const int tid = threadIdx.x;
  if(tid < LAT) sz[tid] = z[f * z_stride + tid];
  __syncthreads();
  // layer 0 (+ LeakyReLU 0.01): two rows per thread
  for(int row = tid; row < HID; row += 256)
  {
    float h = b0[row];
    for(int c = 0; c < LAT; c++) h += w0t[c * HID + row] * sz[c];
    const float slope = (h > 0.0f) ? 1.0f : 0.01f;
    L1[row * VS + 32] = h * slope;
    for(int c = 0; c < LAT; c++) L1[row * VS + c] = slope * w0t[c * HID + row];
  }
  __syncthreads();
  // layer 1 (+ LeakyReLU).  The activation column is a matrix-vector product on the VALU (two rows per thread); the 32
  // tangent columns are a [512 x 512] . [512 x 32] GEMM on v_mfma_f32_32x32x2_f32 (exact fp32): wavefront w owns the four
  // 32-row tiles 4w..4w+3, lane l feeds A[row = l % 32][k = l / 32] = W1[row][k] (K-major weights: coalesced) and
  // B[k = l / 32][col = l % 32] = D1[k][col] from LDS, weights prefetched eight k ahead.
  // With a Jacobian the activation column rides in the tangent loop below instead (same weights, VALU work in the MFMA
  // shadows); both forms sum the even and the odd k separately and add the halves (they agree to the last bits: 4e-8 rad).
  if(!want_jac)
  {
    const int r0 = tid, r1 = tid + 256;
    float h0 = 0.f, h1 = 0.f, h0o = 0.f, h1o = 0.f;
    constexpr int KU = 16;
    float wa[KU], wb[KU], wan[KU], wbn[KU];
#pragma unroll
    for(int u = 0; u < KU; u++)
    {
      wa[u] = w1t[u * HID + r0];
      wb[u] = w1t[u * HID + r1];
    }
    for(int k0 = 0; k0 < HID; k0 += KU)
    {
      const int kn = (k0 + KU < HID) ? k0 + KU : k0;
#pragma unroll
      for(int u = 0; u < KU; u++)
      {
        wan[u] = w1t[(kn + u) * HID + r0];
        wbn[u] = w1t[(kn + u) * HID + r1];
      }
#pragma unroll
      for(int u = 0; u < KU; u += 2)
      {
        const float dv = L1[(k0 + u) * VS + 32], dvo = L1[(k0 + u + 1) * VS + 32];
        h0 += wa[u] * dv;
        h1 += wb[u] * dv;
        h0o += wa[u + 1] * dvo;
        h1o += wb[u + 1] * dvo;
      }
#pragma unroll
      for(int u = 0; u < KU; u++)
      {
        wa[u] = wan[u];
        wb[u] = wbn[u];
      }
    }
    h0 = (h0 + h0o) + b1[r0];
    h1 = (h1 + h1o) + b1[r1];
    const float s0 = (h0 > 0.0f) ? 1.0f : 0.01f, s1 = (h1 > 0.0f) ? 1.0f : 0.01f;
    L2[r0 * VS + 32] = h0 * s0;
    L2[r1 * VS + 32] = h1 * s1;
    sSlope[r0] = s0;
    sSlope[r1] = s1;
  }
  if(want_jac)
  {
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    const int wave = tid >> 6, l = tid & 63, l31 = l & 31, lh = l >> 5;
    f32x16 acc[4];
    float hq[4] = {0.f, 0.f, 0.f, 0.f}; // activation of row 32 (4 wave + t) + l31 over this lane's k parity (lh)
#pragma unroll
    for(int t = 0; t < 4; t++)
#pragma unroll
      for(int r = 0; r < 16; r++) acc[t][r] = 0.0f;
    constexpr int KU = 8; // k-steps of 2 per batch
    float wq[KU][4], wn[KU][4];
#pragma unroll
    for(int u = 0; u < KU; u++)
#pragma unroll
      for(int t = 0; t < 4; t++) wq[u][t] = w1t[(2 * u + lh) * HID + 32 * (4 * wave + t) + l31];
    for(int k2 = 0; k2 < HID / 2; k2 += KU)
    {
      const int kn = (k2 + KU < HID / 2) ? k2 + KU : k2;
#pragma unroll
      for(int u = 0; u < KU; u++)
#pragma unroll
        for(int t = 0; t < 4; t++) wn[u][t] = w1t[(2 * (kn + u) + lh) * HID + 32 * (4 * wave + t) + l31];
#pragma unroll
      for(int u = 0; u < KU; u++)
      {
        const float bv = L1[(2 * (k2 + u) + lh) * VS + l31], av = L1[(2 * (k2 + u) + lh) * VS + 32];
#pragma unroll
        for(int t = 0; t < 4; t++)
        {
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[u][t], bv, acc[t], 0, 0, 0);
          hq[t] += wq[u][t] * av;
        }
      }
#pragma unroll
      for(int u = 0; u < KU; u++)
#pragma unroll
        for(int t = 0; t < 4; t++) wq[u][t] = wn[u][t];
    }
#pragma unroll
    for(int t = 0; t < 4; t++)
    {
      const int row = 32 * (4 * wave + t) + l31;
      const float ho = __shfl_xor(hq[t], 32, 64); // the other k parity of the same row
      const float h = ((lh ? ho + hq[t] : hq[t] + ho)) + b1[row]; // (even + odd) + bias, like the VALU form
      const float sl = (h > 0.0f) ? 1.0f : 0.01f;
      if(lh == 0)
      {
        L2[row * VS + 32] = h * sl;
        sSlope[row] = sl;
      }
    }
    __syncthreads(); // slopes are in LDS
#pragma unroll
    for(int t = 0; t < 4; t++)
#pragma unroll
      for(int r = 0; r < 16; r++)
      {
        const int row = 32 * (4 * wave + t) + (r & 3) + 8 * (r >> 2) + 4 * lh;
        L2[row * VS + l31] = sSlope[row] * acc[t][r];
      }
  }
  __syncthreads();
  // layer 2: 126 rows (activation column on the VALU, tangents on the matrix pipe: wavefront w owns rows 32w..32w+31)
  if(!want_jac && tid < OUT6)
  {
    float h = 0.f, ho = 0.f;
    constexpr int KU = 16;
    float wv[KU], wvn[KU];
#pragma unroll
    for(int u = 0; u < KU; u++) wv[u] = w2t[u * OUT6 + tid];
    for(int k0 = 0; k0 < HID; k0 += KU)
    {
      const int kn = (k0 + KU < HID) ? k0 + KU : k0;
#pragma unroll
      for(int u = 0; u < KU; u++) wvn[u] = w2t[(kn + u) * OUT6 + tid];
#pragma unroll
      for(int u = 0; u < KU; u += 2)
      {
        h += wv[u] * L2[(k0 + u) * VS + 32];
        ho += wv[u + 1] * L2[(k0 + u + 1) * VS + 32];
      }
#pragma unroll
      for(int u = 0; u < KU; u++) wv[u] = wvn[u];
    }
    so[tid * 33 + 32] = (h + ho) + b2[tid];
  }
  if(want_jac)
  {
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    const int wave = tid >> 6, l = tid & 63, l31 = l & 31, lh = l >> 5;
    const int arow = 32 * wave + l31;
    const bool alive = arow < OUT6;
    const int acol = alive ? arow : 0;
    f32x16 acc;
    float hq = 0.f; // activation of row arow over this lane's k parity
#pragma unroll
    for(int r = 0; r < 16; r++) acc[r] = 0.0f;
    constexpr int KU = 8;
    float wq[KU], wn[KU];
#pragma unroll
    for(int u = 0; u < KU; u++) wq[u] = alive ? w2t[(2 * u + lh) * OUT6 + acol] : 0.0f;
    for(int k2 = 0; k2 < HID / 2; k2 += KU)
    {
      const int kn = (k2 + KU < HID / 2) ? k2 + KU : k2;
#pragma unroll
      for(int u = 0; u < KU; u++)
      {
        const float wv = w2t[(2 * (kn + u) + lh) * OUT6 + acol];
        wn[u] = alive ? wv : 0.0f;
      }
#pragma unroll
      for(int u = 0; u < KU; u++)
      {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[u], L2[(2 * (k2 + u) + lh) * VS + l31], acc, 0, 0, 0);
        hq += wq[u] * L2[(2 * (k2 + u) + lh) * VS + 32];
      }
#pragma unroll
      for(int u = 0; u < KU; u++) wq[u] = wn[u];
    }
    {
      const float hx = __shfl_xor(hq, 32, 64);
      const float h = (lh ? hx + hq : hq + hx) + b2[acol];
      if(alive && lh == 0) so[arow * 33 + 32] = h;
    }
#pragma unroll
    for(int r = 0; r < 16; r++)
    {
      const int row = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if(row < OUT6) so[row * 33 + l31] = acc[r];
    }
  }
  __syncthreads();
  // rotation tail: 6D -> axis-angle and its 3 x 6 Jacobian, one thread per joint; the chain rule into the 32 latent columns
  // (63 x 32 entries, six terms each) by all threads, coalesced over the columns
  float * sj = L2; // [21][18]  (the a2/D2 region is dead once layer 2 is done)
  if(tid < 21)
  {
    float o6[6], aa[3], j36[18];
    for(int q = 0; q < 6; q++) o6[q] = so[(tid * 6 + q) * 33 + 32];
    sixd_to_aa(o6, aa, want_jac ? j36 : nullptr);
    for(int i = 0; i < 3; i++) out[f * out_stride + tid * 3 + i] = aa[i];
    if(want_jac)
      for(int q = 0; q < 18; q++) sj[tid * 18 + q] = j36[q];
  }
  if(want_jac)
  {
    __syncthreads();
    for(int item = tid; item < 63 * LAT; item += 256)
    {
      const int row = item / LAT, c = item % LAT, j = row / 3, i = row % 3;
      float s = 0.f;
      for(int q = 0; q < 6; q++) s += sj[j * 18 + i * 6 + q] * so[(j * 6 + q) * 33 + c];
      jac[(f * 63 + row) * LAT + c] = s;
    }
  }
}

__global__ void rotmat_to_aa_kernel(const float * __restrict__ rot, float * __restrict__ aa_out, int64_t n)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n) return;
  D6 R[3][3], aa[3];
  for(int r = 0; r < 3; r++)
    for(int c = 0; c < 3; c++) R[r][c] = mk(rot[i * 9 + r * 3 + c]);
  rotmat_to_aa(R, aa);
  for(int q = 0; q < 3; q++) aa_out[i * 3 + q] = aa[q].v;
}

int vposer_forward_device(smplpp_vposer * v, int64_t n, const float * z, int64_t z_stride, float * out, int64_t out_stride,
                          float * jac, hipStream_t st)
{
  const size_t shmem = sizeof(float) * (size_t)(2 * HID * VS + LAT + HID);
  static PerDeviceOnce once;
  HIP_TRY(lds_opt_in(once, v->device, reinterpret_cast<const void *>(&vposer_kernel), (int)shmem));
  vposer_kernel<<<dim3((unsigned)n), dim3(256), shmem, st>>>(z, z_stride, v->w0t, v->b0, v->w1t, v->b1, v->w2t, v->b2, out, out_stride,
                                                            jac, jac ? 1 : 0);
  HIP_TRY(hipGetLastError());
  return SMPLPP_OK;
}
} // namespace smplpp_hip

using namespace smplpp_hip;

extern "C" int smplpp_vposer_destroy(smplpp_vposer * v)
{
  if(!v) return SMPLPP_OK;
  (void)hipSetDevice(v->device);
  for(float * p : {v->w0t, v->b0, v->w1t, v->b1, v->w2t, v->b2})
    if(p) (void)hipFree(p);
  delete v;
  return SMPLPP_OK;
}

static hipError_t upload_t(float ** dst, const float * w, int out, int in)
{
  std::vector<float> t((size_t)out * in);
  for(int o = 0; o < out; o++)
    for(int i = 0; i < in; i++) t[(size_t)i * out + o] = w[(size_t)o * in + i]; // [out,in] -> [in][out]
  hipError_t e = hipMalloc((void **)dst, sizeof(float) * t.size());
  if(e != hipSuccess) return e;
  return hipMemcpy(*dst, t.data(), sizeof(float) * t.size(), hipMemcpyHostToDevice);
}

extern "C" int smplpp_vposer_create(int device, const float * w0, const float * b0, const float * w1, const float * b1,
                                    const float * w2, const float * b2, smplpp_vposer ** out)
{
  if(!out || !w0 || !b0 || !w1 || !b1 || !w2 || !b2) return fail(SMPLPP_ERR_INVALID, "smplpp_vposer_create: null argument");
  *out = nullptr;
  int ndev = 0;
  int rc = smplpp_device_count(&ndev);
  if(rc) return rc;
  if(device < 0 || device >= ndev) return fail(SMPLPP_ERR_INVALID, "Failed to fetch device index!");
  HIP_TRY(hipSetDevice(device));
  smplpp_vposer * v = new smplpp_vposer();
  v->device = device;
  hipError_t e = upload_t(&v->w0t, w0, HID, LAT);
  if(e == hipSuccess) e = upload_t(&v->w1t, w1, HID, HID);
  if(e == hipSuccess) e = upload_t(&v->w2t, w2, OUT6, HID);
  if(e == hipSuccess) e = upload_t(&v->b0, b0, 1, HID);
  if(e == hipSuccess) e = upload_t(&v->b1, b1, 1, HID);
  if(e == hipSuccess) e = upload_t(&v->b2, b2, 1, OUT6);
  if(e != hipSuccess)
  {
    int r = hip_fail(e, "vposer upload", __FILE__, __LINE__);
    smplpp_vposer_destroy(v);
    return r;
  }
  *out = v;
  return SMPLPP_OK;
}

extern "C" int smplpp_vposer_forward(smplpp_vposer * v, int64_t n, const float * z, float * out, float * jac, int space,
                                     void * stream)
{
  if(!v || n <= 0 || !z || !out) return fail(SMPLPP_ERR_INVALID, "smplpp_vposer_forward: bad argument");
  int rc = check_space(space, "smplpp_vposer_forward");
  if(rc) return rc;
  HIP_TRY(hipSetDevice(v->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  In<float> zi;
  Out<float> oo, jo;
  HIP_TRY(zi.init(z, (size_t)n * LAT, space, st));
  HIP_TRY(oo.init(out, (size_t)n * 63, space));
  HIP_TRY(jo.init(jac, (size_t)n * 63 * LAT, space));
  rc = vposer_forward_device(v, n, zi.d, LAT, oo.d, 63, jo.d, st);
  if(rc) return rc;
  hipError_t e = oo.finish(st);
  if(e == hipSuccess) e = jo.finish(st);
  if(e == hipSuccess && space == SMPLPP_HOST) e = hipStreamSynchronize(st);
  HIP_TRY(e);
  return SMPLPP_OK;
}

extern "C" int smplpp_rotmat_to_axis_angle(int device, int64_t n, const float * rot, float * aa, int space, void * stream)
{
  if(n <= 0 || !rot || !aa) return fail(SMPLPP_ERR_INVALID, "smplpp_rotmat_to_axis_angle: bad argument");
  int rc = check_space(space, "smplpp_rotmat_to_axis_angle");
  if(rc) return rc;
  int ndev = 0;
  rc = smplpp_device_count(&ndev);
  if(rc) return rc;
  if(device < 0 || device >= ndev) return fail(SMPLPP_ERR_INVALID, "Failed to fetch device index!");
  HIP_TRY(hipSetDevice(device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  In<float> ri;
  Out<float> ao;
  HIP_TRY(ri.init(rot, (size_t)n * 9, space, st));
  HIP_TRY(ao.init(aa, (size_t)n * 3, space));
  rotmat_to_aa_kernel<<<dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st>>>(ri.d, ao.d, n);
  hipError_t e = hipGetLastError();
  if(e == hipSuccess) e = ao.finish(st);
  if(e == hipSuccess && space == SMPLPP_HOST) e = hipStreamSynchronize(st);
  HIP_TRY(e);
  return SMPLPP_OK;
}
