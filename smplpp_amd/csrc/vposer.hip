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
#include "signal.h"

#include <cfloat>
#include <cmath>
#include <vector>

// Contraction decided by the SOURCE, not by the back end: a * b + c inside one expression is an fma, across statements it is not.
// (Under hipcc's default, fp-contract=fast, the instruction selector fuses wherever a product has a single use — and the one- and
// two-frame instantiations of vposer_jac2_kernel, whose products have different use counts, then rounded the SAME frame
// differently: a frame's bits must not depend on the batch it travels in.)
#pragma clang fp contract(on)

struct smplpp_vposer
{
  int device = 0;
  float *w0t = nullptr, *b0 = nullptr, *w1t = nullptr, *b1 = nullptr, *w2t = nullptr, *b2 = nullptr;
  // layers 1 and 2 once more as fp16x2 pieces in MFMA fragment order (the A operand of the tangent GEMMs, layout below)
  uint8_t *w1h = nullptr, *w2h = nullptr;
  float sW1 = 1.f, sW2 = 1.f, sD1 = 1.f, sD2 = 1.f; // power-of-two scales: weights of layers 1 / 2, tangent blocks of layers 0 / 1
  // vposer_jac2_kernel (several frames per workgroup): W0 once more as the B operand of layer 1's tangent GEMM (fragment order,
  // scale sD1, no slopes) and the constant product C10 = W1 . W0 [512][32] (fp32, from an fp64 sum on the host)
  uint8_t * w0h = nullptr;
  float * c10 = nullptr;
};

namespace smplpp_hip
{
constexpr int LAT = SMPLPP_LATENT_DIM; // 32
constexpr int HID = 512;               // VPoser.h hiddenDim_
constexpr int OUT6 = 126;              // 6 * 21

// dual number: a value and ND directional derivatives (ND = 6: all six inputs of a joint at once; ND = 1: one direction per
// thread — the components never mix, so both give the same bits)
template<int ND>
struct DN
{
  float v;
  float d[ND];
};
typedef DN<6> D6;
template<int ND>
__device__ inline DN<ND> mk(float v)
{
  DN<ND> r;
  r.v = v;
  for(int i = 0; i < ND; i++) r.d[i] = 0.f;
  return r;
}
template<int ND>
__device__ inline DN<ND> operator+(const DN<ND> & a, const DN<ND> & b)
{
  DN<ND> r;
  r.v = a.v + b.v;
  for(int i = 0; i < ND; i++) r.d[i] = a.d[i] + b.d[i];
  return r;
}
template<int ND>
__device__ inline DN<ND> operator-(const DN<ND> & a, const DN<ND> & b)
{
  DN<ND> r;
  r.v = a.v - b.v;
  for(int i = 0; i < ND; i++) r.d[i] = a.d[i] - b.d[i];
  return r;
}
template<int ND>
__device__ inline DN<ND> operator*(const DN<ND> & a, const DN<ND> & b)
{
  DN<ND> r;
  r.v = a.v * b.v;
  for(int i = 0; i < ND; i++) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
  return r;
}
template<int ND>
__device__ inline DN<ND> operator/(const DN<ND> & a, const DN<ND> & b)
{
  DN<ND> r;
  r.v = a.v / b.v;
  for(int i = 0; i < ND; i++) r.d[i] = (a.d[i] - r.v * b.d[i]) / b.v;
  return r;
}
template<int ND>
__device__ inline DN<ND> operator*(float s, const DN<ND> & a)
{
  DN<ND> r;
  r.v = s * a.v;
  for(int i = 0; i < ND; i++) r.d[i] = s * a.d[i];
  return r;
}
template<int ND>
__device__ inline DN<ND> operator+(const DN<ND> & a, float s)
{
  DN<ND> r = a;
  r.v += s;
  return r;
}
template<int ND>
__device__ inline DN<ND> neg(const DN<ND> & a)
{
  return -1.0f * a;
}
template<int ND>
__device__ inline DN<ND> dsqrt(const DN<ND> & a)
{
  DN<ND> r;
  r.v = sqrtf(a.v);
  for(int i = 0; i < ND; i++) r.d[i] = a.d[i] / (2.0f * r.v);
  return r;
}
template<int ND>
__device__ inline DN<ND> dacos(const DN<ND> & a)
{
  DN<ND> r;
  r.v = acosf(a.v);
  const float g = -1.0f / sqrtf(1.0f - a.v * a.v);
  for(int i = 0; i < ND; i++) r.d[i] = g * a.d[i];
  return r;
}
template<int ND>
__device__ inline DN<ND> dsin(const DN<ND> & a)
{
  DN<ND> r;
  r.v = sinf(a.v);
  const float c = cosf(a.v);
  for(int i = 0; i < ND; i++) r.d[i] = c * a.d[i];
  return r;
}
// torch::nn::functional::normalize of a 3-vector: x / max(||x||, 1e-12) (clamp_min passes no gradient when active)
template<int ND>
__device__ inline void dnormalize3(const DN<ND> * x, DN<ND> * o)
{
  DN<ND> n2 = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
  DN<ND> n = dsqrt(n2);
  if(n.v < 1e-12f) n = mk<ND>(1e-12f);
  for(int i = 0; i < 3; i++) o[i] = x[i] / n;
}

// convertRotMatToAxisAngle (src/VPoser.cpp:25-120) on one matrix, value + derivative
template<int ND>
__device__ inline void rotmat_to_aa(const DN<ND> R[3][3], DN<ND> aa[3])
{
  const float eps = FLT_EPSILON;
  const float epsSqrt = sqrtf(eps);
  const float epsSqrt2 = sqrtf(epsSqrt);
  const float kPi = 3.14159265358979323846f;
  DN<ND> trace = R[0][0] + R[1][1] + R[2][2];
  DN<ND> theta = dacos((float)((1.0 - (double)eps) * 0.5) * (trace + (-1.0f))); // :41
  DN<ND> w[3] = {R[2][1] - R[1][2], R[0][2] - R[2][0], R[1][0] - R[0][1]};      // :43-49
  if(1.0f + trace.v < epsSqrt2) // near pi (:53-103)
  {
    DN<ND> tn2[3];
    DN<ND> one_m_tr = neg(trace) + 1.0f, three_m_tr = neg(trace) + 3.0f;
    for(int i = 0; i < 3; i++)
    {
      DN<ND> s = (2.0f * R[i][i] + one_m_tr) / three_m_tr; // :54-56
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
    DN<ND> t2 = theta * theta;
    DN<ND> f = (1.0f / 6.0f) * t2 + (7.0f / 360.0f) * (t2 * t2) + 1.0f;
    for(int i = 0; i < 3; i++) aa[i] = 0.5f * (w[i] * f);
  }
  else // :112-116
  {
    DN<ND> f = theta / (2.0f * dsin(theta));
    for(int i = 0; i < 3; i++) aa[i] = w[i] * f;
  }
}

// ContinousRotReprDecoderImpl::forward (:129-141) on one joint's 6 numbers (view [3,2]) then -> axis-angle.  ND = 6: all six
// derivative directions (jac36 [3][6]); ND = 1: direction `dir` only (jac36 [3]: d aa / d o6[dir])
template<int ND>
__device__ inline void sixd_to_aa_dir(const float * o6, int dir, float * aa_out, float * jac)
{
  DN<ND> c1[3], c2[3];
  for(int r = 0; r < 3; r++)
  {
    c1[r] = mk<ND>(o6[2 * r]);
    c2[r] = mk<ND>(o6[2 * r + 1]);
    if(ND == 6)
    {
      c1[r].d[(2 * r) % ND] = 1.0f;
      c2[r].d[(2 * r + 1) % ND] = 1.0f;
    }
    else
    {
      c1[r].d[0] = (2 * r == dir) ? 1.0f : 0.0f;
      c2[r].d[0] = (2 * r + 1 == dir) ? 1.0f : 0.0f;
    }
  }
  DN<ND> a1[3], a2[3], t[3];
  dnormalize3(c1, a1);
  DN<ND> dot = a1[0] * c2[0] + a1[1] * c2[1] + a1[2] * c2[2];
  for(int r = 0; r < 3; r++) t[r] = c2[r] - dot * a1[r];
  dnormalize3(t, a2);
  DN<ND> a3[3] = {a1[1] * a2[2] - a1[2] * a2[1], a1[2] * a2[0] - a1[0] * a2[2], a1[0] * a2[1] - a1[1] * a2[0]};
  DN<ND> R[3][3];
  for(int r = 0; r < 3; r++)
  {
    R[r][0] = a1[r];
    R[r][1] = a2[r];
    R[r][2] = a3[r];
  }
  DN<ND> aa[3];
  rotmat_to_aa(R, aa);
  for(int i = 0; i < 3; i++)
  {
    aa_out[i] = aa[i].v;
    if(jac)
      for(int q = 0; q < ND; q++) jac[i * ND + q] = aa[i].d[q];
  }
}
__device__ inline void sixd_to_aa(const float * o6, float * aa_out, float * jac36 /*[3][6]*/)
{
  sixd_to_aa_dir<6>(o6, 0, aa_out, jac36);
}

// Value-only forward (no Jacobian: decoding a stored latent, node/node.cpp:1376-1391): grid = n frames, block = 256, exact
// fp32 on the VALU.  LDS: activations of layers 0 / 1 in column 32 of [512][VS] rows (the layout the first Jacobian form
// shared), the layer-2 output [126][33] reuses the first region.
constexpr int VS = 36;
__global__ __launch_bounds__(256) void vposer_kernel(const float * __restrict__ z, int64_t z_stride, const float * __restrict__ w0t,
                                                     const float * __restrict__ b0, const float * __restrict__ w1t,
                                                     const float * __restrict__ b1, const float * __restrict__ w2t,
                                                     const float * __restrict__ b2, float * __restrict__ out, int64_t out_stride,
                                                     float * __restrict__ /*unused*/, int /*unused*/)
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
  }
  __syncthreads();
  // layer 1 (+ LeakyReLU): a matrix-vector product on the VALU, two rows per thread, even and odd k summed separately
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
  __syncthreads();
  // layer 2: 126 rows
  if(tid < OUT6)
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
  __syncthreads();
  // rotation tail: 6D -> axis-angle and its 3 x 6 Jacobian, one thread per joint; the chain rule into the 32 latent columns
  // (63 x 32 entries, six terms each) by all threads, coalesced over the columns
  if(tid < 21)
  {
    float o6[6], aa[3];
    for(int q = 0; q < 6; q++) o6[q] = so[(tid * 6 + q) * 33 + 32];
    sixd_to_aa(o6, aa, nullptr);
    for(int i = 0; i < 3; i++) out[f * out_stride + tid * 3 + i] = aa[i];
  }
}

// ---- forward + Jacobian on the f16 matrix pipe.
// d(out)/dz is carried through the MLP as 32 tangent columns per row.  Layer 1: [512 x 512] . [512 x 32], layer 2:
// [126 x 512] . [512 x 32] per frame: v_mfma_f32_32x32x16_f16 with every fp32 operand as two fp16 pieces of a power-of-two
// multiple (common.h, "fp16x2": 22 significant bits, three products hi.hi + hi.lo + lo.hi) — 96 MFMAs of 32 cycles per
// 32-row tile where the exact-fp32 form (v_mfma_f32_32x32x2_f32, 1/16 of the rate) took 256 of 64.
//   A operand: the weights, split once at creation, in fragment order: w1h [16 row tiles][32 k-steps][piece 2][64 lanes][8 fp16]
//              (lane 32 h + r holds W[32 tile + r][16 ks + 8 h + j], j = 0..7), straight from L2 to registers;
//   B operand: the tangent block of the previous layer, written to LDS in fragment order by its producer:
//              Df [32 k-steps][piece 2][64 lanes (32 h + column)][8 fp16].
// The VALUE path (activations) rides on the same weight fragments, as one more B tile whose columns are the frames' activation
// vectors (fp16x2 pieces too, per-frame power-of-two scale) — no second (fp32) stream of the weights, which at 1 MB per layer and
// frame was what the first form of the kernel waited for.  Values carry 22-bit operands: 1e-6 relative to the fp32 VALU form of the
// value-only kernel.  LDS plan: struct VJ2 below.
typedef float f32x16v __attribute__((ext_vector_type(16)));
typedef float v4fv __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4v __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
constexpr int VJ_DF = 32 * 2 * 1024; // bytes of one tangent block in fragment order
constexpr int VJ_AF = 32 * 2 * 2 * 16; // bytes of one activation vector in fragment order

__device__ __forceinline__ f32x16v vmfma(const v4fv & a, const v4fv & b, const f32x16v & c)
{
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8v, a), __builtin_bit_cast(f16x8v, b), c, 0, 0, 0);
}
// per-frame power-of-two scale that puts max|v| of a 512-vector (two entries per thread) just under 2^14; red: [5] floats of LDS
__device__ __forceinline__ float vscale512(float v0, float v1, float * red)
{
  float m = fmaxf(fabsf(v0), fabsf(v1));
  for(int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  return exp2f(floorf(log2f(16384.0f / fmaxf(m, 1e-30f))));
}
// entry k of an activation vector into fragment order: k-step k / 16, half (k % 16) / 8, element k % 8
__device__ __forceinline__ void vput_act(unsigned char * Af, int k, float scaled)
{
  _Float16 hi, lo;
  split_f16x2(scaled, hi, lo);
  _Float16 * p = reinterpret_cast<_Float16 *>(Af + (k >> 4) * 64 + ((k >> 3) & 1) * 16) + (k & 7);
  p[0] = hi;
  p[16] = lo; // piece 1: + 32 bytes
}

#ifdef VPJ_STAMP
__device__ unsigned long long g_vpj_stamps[16];
#define VPJ_T(i) if(blockIdx.x == 0 && threadIdx.x == 0) { g_vpj_stamps[i] = __builtin_amdgcn_s_memtime(); g_vpj_stamps[8 + (i)] = __builtin_amdgcn_s_memrealtime(); }
extern "C" int smplpp_debug_vpj_stamps(unsigned long long * out)
{
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_vpj_stamps), sizeof(unsigned long long) * 16);
}
#else
#define VPJ_T(i)
#endif
// ---- forward + Jacobian, NF frames per workgroup (vposer_jac2_kernel).
// Round 2's kernel (one frame per workgroup with BOTH tangent blocks in LDS; removed in round 4, when the NF = 1 instantiation of
// this one proved at least as fast at every batch size and bit-identical per frame to NF = 2) was bound by its weight stream: per
// k-step a wavefront pulled 8 KB of W1 fragments for 384 cycles of MFMA (85 B/clk per CU against the 64 B/clk a CU's vector memory
// path delivers).  Here NF frames share every weight fragment (NF x the MFMAs per byte), made possible by NOT storing the layer-0
// tangent block:  T0 = diag(s) W0 with LeakyReLU slopes s in {1, 0.01}, so
//     W1 . T0 = 0.99 W1 . (m (.) W0) + 0.01 W1 . W0            (m = the rows with slope 1)
// — the B operand of layer 1 is ONE W0 fragment stream (w0h, shared by all frames and workgroups, from L2 like the weights),
// masked per frame in registers (a 1 KiB mask image per frame in LDS), and C10 = W1 . W0 is a constant of the model added in
// fp32.  LDS then only holds the layer-1 tangent blocks (64 KiB per frame): NF = 2 fits.
constexpr int VJ_GROUP = 2; // frames per k-loop rotation group (= the frames per workgroup of the many-frames instantiation)
template<int NF>
struct VJ2
{
  static constexpr int D2 = 0;                          // [NF][VJ_DF] layer-1 tangent blocks, fragment order
  static constexpr int A2 = NF * VJ_DF;                 // [NF][512] fp32 layer-1 activations
  static constexpr int SL = A2 + NF * HID * 4;          // [NF][512] fp32 layer-1 slopes
  static constexpr int AF = SL + NF * HID * 4;          // [NF][VJ_AF] activation fragments of the layer being consumed
  static constexpr int MK = AF + NF * VJ_AF;            // [NF][32 k-steps][2 halves][8 fp16] row masks of layer 0 (0xffff: slope 1)
  static constexpr int SZ = MK + NF * 1024;             // [NF][32] latents
  static constexpr int RED = SZ + NF * LAT * 4;         // [16] scratch of the scale reductions | 64 zero bytes (dead columns of the value tile)
  static constexpr int TOTAL = RED + 64 + 64;
};

// VO (round 5): the VALUE path alone — the same MFMAs on the same operands in the same order for the activation columns, so the
// decoded angles are the bits the full kernel decodes, in about half its time (no tangent MFMAs, no tangent blocks, no chain
// rule).  The capture loops use it to get theta25 early: the pose step, the fused kernel and the evaluation's direct rows need the
// value only, and the full kernel (the Jacobian; `out` null) then runs beside them on the side stream (ik_forward_eval).
// sig_*: the side-stream launch raises the join flag itself (signal.h); its Jacobian is then stored write-through.
template<int NF, bool VO>
__global__ __launch_bounds__(256) void vposer_jac2_kernel(const float * __restrict__ z, int64_t z_stride, const float * __restrict__ w0t,
                                                          const float * __restrict__ b0, const float * __restrict__ b1,
                                                          const float * __restrict__ b2, const uint8_t * __restrict__ w1h,
                                                          const uint8_t * __restrict__ w2h, const uint8_t * __restrict__ w0h,
                                                          const float * __restrict__ c10, float sD1, float sD2, float iW1, float iW2,
                                                          float * __restrict__ out, int64_t out_stride, float * __restrict__ jac, int64_t n,
                                                          int64_t frame_base, unsigned * __restrict__ sig_flag,
                                                          unsigned * __restrict__ sig_counter, unsigned sig_tick)
{
  typedef VJ2<NF> L;
  extern __shared__ __attribute__((aligned(16))) unsigned char vl[];
  float * red = reinterpret_cast<float *>(vl + L::RED);
  // A frame's bits must not depend on where its shard starts or how many frames run beside it (a job sharded over 1, 4 or 8 GPUs
  // decodes every latent to the same bits): workgroups are aligned to GROUPS of the GLOBAL frame index (frame_base = the global
  // index of this launch's frame 0), the k-loop rotation below is a function of the group, and a frame's arithmetic is the same in
  // every instantiation (its columns of the MFMA tiles never mix with its neighbour's).  f0 = this launch's index of the
  // workgroup's slot 0: negative in the first workgroup of a shard that starts inside a group.
  const int64_t group = frame_base / VJ_GROUP + (NF == VJ_GROUP ? (int64_t)blockIdx.x : 0);
  const int64_t f0 = (NF == VJ_GROUP) ? group * NF - frame_base : (int64_t)blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63, l31 = l & 31, lh = l >> 5;
  VPJ_T(0);
  // One frame per workgroup (few frames: every launch finds its XCD's L2 cold for the weights): what the later phases read from
  // global memory FIRST is requested in the kernel's first round trip.  With every CU streaming (two frames per workgroup, 512
  // frames) the same requests lengthened the kernel by 2 us of 43: that instantiation asks where it uses, as before.
  constexpr bool EARLY = NF == 1;
  // (the latents first: the wait for them must not stand behind everything requested below — loads retire in order)
  float zreg = 0.0f;
  if(EARLY && tid < NF * LAT)
  {
    int64_t f = f0 + tid / LAT; // (a workgroup's spare slot repeats a frame of the launch; never stored)
    f = f < 0 ? 0 : (f < n ? f : n - 1);
    zreg = z[f * z_stride + tid % LAT];
  }
  // ---- layer 0 (+ LeakyReLU 0.01), rows tid and tid + 256 of every frame: the 64 weights of the two rows are loaded once
  float w0a[LAT], w0b[LAT];
#pragma unroll
  for(int c = 0; c < LAT; c++)
  {
    w0a[c] = w0t[c * HID + tid];
    w0b[c] = w0t[c * HID + tid + 256];
  }
  // Everything the later phases read from global memory FIRST is requested here, in the kernel's first round trip: the biases of
  // layers 0 and 1 and the first three k-steps of layer 1's weight stream.  (Round 5, phase stamps of the one-frame value path: each
  // of these sat behind a barrier as a cold round trip of ~1.7 k cycles — the XCD's L2 does not keep the weights across launches —,
  // six of them in a 35 k-cycle kernel.)
  float bz0 = 0.0f, bz1 = 0.0f, bb0 = 0.0f, bb1 = 0.0f;
  if constexpr(EARLY)
  {
    bz0 = b0[tid];
    bz1 = b0[tid + 256];
    bb0 = b1[tid];
    bb1 = b1[tid + 256];
  }
  // (every workgroup starts its k loop elsewhere, so that the CUs do not ask the L2s for the same weight lines at the same
  // instant: -2 % per latent IK iteration; by the GLOBAL group, so that sharding moves no bit)
  const int rot = (int)(((unsigned long long)(NF == VJ_GROUP ? group : (frame_base + f0) / VJ_GROUP) * 5ull) & 31ull);
  const uint8_t * const ap1 = w1h + (size_t)(4 * wave) * (32 * 2048) + l * 16;
  const uint8_t * const bp1 = w0h + l * 16;
  v4fv st[4][4][2], sb[4][2];
  auto load_stage = [&](int sidx, int ks) {
#pragma unroll
    for(int t = 0; t < 4; t++)
#pragma unroll
      for(int p = 0; p < 2; p++) st[sidx][t][p] = *reinterpret_cast<const v4fv *>(ap1 + (size_t)t * (32 * 2048) + ks * 2048 + p * 1024);
    if constexpr(!VO)
    {
      sb[sidx][0] = *reinterpret_cast<const v4fv *>(bp1 + ks * 2048);
      sb[sidx][1] = *reinterpret_cast<const v4fv *>(bp1 + ks * 2048 + 1024);
    }
  };
  if constexpr(EARLY)
  {
    load_stage(0, rot);
    load_stage(1, (rot + 1) & 31);
    load_stage(2, (rot + 2) & 31);
  }
  if(tid >= 128 && tid < 144) reinterpret_cast<float *>(vl + L::RED + 64)[tid - 128] = 0.0f;
  if(tid < NF * LAT)
  {
    if constexpr(!EARLY)
    {
      int64_t f = f0 + tid / LAT;
      f = f < 0 ? 0 : (f < n ? f : n - 1);
      zreg = z[f * z_stride + tid % LAT];
    }
    reinterpret_cast<float *>(vl + L::SZ)[tid] = zreg;
  }
  __syncthreads();
  float sA1[NF];
#pragma unroll
  for(int q = 0; q < NF; q++)
  {
    const float * sz = reinterpret_cast<const float *>(vl + L::SZ) + q * LAT;
    float h0 = EARLY ? bz0 : b0[tid], h1 = EARLY ? bz1 : b0[tid + 256];
#pragma unroll
    for(int c = 0; c < LAT; c++)
    {
      h0 += w0a[c] * sz[c];
      h1 += w0b[c] * sz[c];
    }
    const bool p0 = h0 > 0.0f, p1 = h1 > 0.0f;
    h0 *= p0 ? 1.0f : 0.01f;
    h1 *= p1 ? 1.0f : 0.01f;
    // row masks in the B fragment's element order: row k is element k % 8 of lane half (k % 16) / 8 in k-step k / 16
    if constexpr(!VO)
    {
      unsigned short * mk = reinterpret_cast<unsigned short *>(vl + L::MK + q * 1024);
      mk[(tid >> 4) * 16 + ((tid >> 3) & 1) * 8 + (tid & 7)] = p0 ? 0xffffu : 0u;
      mk[((tid + 256) >> 4) * 16 + (((tid + 256) >> 3) & 1) * 8 + (tid & 7)] = p1 ? 0xffffu : 0u;
    }
    sA1[q] = vscale512(h0, h1, red);
    vput_act(vl + L::AF + q * VJ_AF, tid, h0 * sA1[q]);
    vput_act(vl + L::AF + q * VJ_AF, tid + 256, h1 * sA1[q]);
  }
  __syncthreads();
  VPJ_T(1);
  v4fv st2[8][2]; // layer 2's weight ring
  float bias2[16];
  // ---- layer 1: wavefront w owns the row tiles 4w .. 4w + 3 of EVERY frame; per k-step 8 W1 fragments and 2 W0 fragments from
  // L2 (prefetched three k-steps ahead), per frame a mask and 2 activation fragments from LDS, 12 MFMAs + 48 v_dot2
  {
    // The VALUE path (h1 = W1 a0 + b1) rides on the matrix pipe too: the activation vectors of the NF frames are columns 0 .. NF - 1
    // of ONE more B tile (the other columns zero), three MFMAs per row tile and k-step for all frames together.  (v_dot2c_f32_f16
    // on the same weight fragments, as round 2's kernel did it, issues once per 16 cycles: its 48 per frame and k-step cost twice
    // the 12 tangent MFMAs they were meant to hide behind.)
    f32x16v acc[NF][4], vacc[4];
#pragma unroll
    for(int t = 0; t < 4; t++)
    {
#pragma unroll
      for(int r = 0; r < 16; r++) vacc[t][r] = 0.0f;
#pragma unroll
      for(int q = 0; q < NF; q++)
#pragma unroll
        for(int r = 0; r < 16; r++) acc[q][t][r] = 0.0f;
    }
    // this lane's slice of the value tile: column l31 = frame (a zeroed slot of LDS for the columns beyond the frames)
    const unsigned char * const avp = (l31 < NF) ? vl + L::AF + l31 * VJ_AF + lh * 16 : vl + L::RED + 64;
    const int avs = (l31 < NF) ? 64 : 0;
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    // LDS operands of a k-step (value tile pieces, one row mask per frame), read one k-step ahead into a second register set
    v4fv vah[2], val[2];
    u4v mk[2][NF];
    auto load_lds = [&](int slot, int ks) {
      vah[slot] = *reinterpret_cast<const v4fv *>(avp + ks * avs);
      val[slot] = *reinterpret_cast<const v4fv *>(avp + ks * avs + (avs >> 1));
      if constexpr(!VO)
      {
#pragma unroll
        for(int q = 0; q < NF; q++) mk[slot][q] = *reinterpret_cast<const u4v *>(vl + L::MK + q * 1024 + ks * 32 + lh * 16);
      }
    };
    if constexpr(!EARLY)
    {
      load_stage(0, rot);
      load_stage(1, (rot + 1) & 31);
      load_stage(2, (rot + 2) & 31);
    }
    load_lds(0, rot); // (weight stages 0..2 of the one-frame instantiation: requested at the kernel's start)
    for(int k4 = 0; k4 < 32; k4 += 4)
    {
#pragma unroll
      for(int u = 0; u < 4; u++)
      {
        const int ks = (k4 + u + rot) & 31;
        load_stage((u + 3) & 3, (ks + 3) & 31);
        load_lds((u + 1) & 1, (ks + 1) & 31);
        // (pinned: with 420 registers live the scheduler otherwise sinks these loads to just in front of their use three k-steps
        // later — and the loop then waits out an L2 round trip per k-step)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for(int t = 0; t < 4; t++)
        {
          vacc[t] = vmfma(st[u][t][0], vah[u & 1], vacc[t]);
          vacc[t] = vmfma(st[u][t][0], val[u & 1], vacc[t]);
          vacc[t] = vmfma(st[u][t][1], vah[u & 1], vacc[t]);
        }
#pragma unroll
        for(int q = 0; q < (VO ? 0 : NF); q++)
        {
          const v4fv bh = __builtin_bit_cast(v4fv, __builtin_bit_cast(u4v, sb[u][0]) & mk[u & 1][q]);
          const v4fv bl = __builtin_bit_cast(v4fv, __builtin_bit_cast(u4v, sb[u][1]) & mk[u & 1][q]);
#pragma unroll
          for(int t = 0; t < 4; t++)
          {
            acc[q][t] = vmfma(st[u][t][0], bh, acc[q][t]);
            acc[q][t] = vmfma(st[u][t][0], bl, acc[q][t]);
            acc[q][t] = vmfma(st[u][t][1], bh, acc[q][t]);
          }
        }
      }
    }
    VPJ_T(2);
    // layer 2's first seven k-steps of weights and its bias are requested HERE, in front of layer 1's epilogue (two barriers and the
    // activation pass away from their first use)
    const uint8_t * const ap2 = w2h + (size_t)wave * (32 * 2048) + l * 16;
    auto load_ring2 = [&]() {
#pragma unroll
      for(int s3 = 0; s3 < 7; s3++)
      {
        st2[s3][0] = *reinterpret_cast<const v4fv *>(ap2 + s3 * 2048);
        st2[s3][1] = *reinterpret_cast<const v4fv *>(ap2 + s3 * 2048 + 1024);
      }
#pragma unroll
      for(int r = 0; r < 16; r++)
      {
        const int row = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * lh;
        bias2[r] = b2[row < OUT6 ? row : 0];
      }
    };
    if constexpr(EARLY) load_ring2();
    // the raw row sums of this wavefront's 128 rows sit in the lanes of column q < NF (frame q), rows (r & 3) + 8 (r >> 2) + 4 lh:
    // dropped into LDS as they are; bias, LeakyReLU and slope by all threads, two rows each, behind the barrier
    if(l31 < NF)
    {
      float * a2 = reinterpret_cast<float *>(vl + L::A2) + l31 * HID;
#pragma unroll
      for(int t = 0; t < 4; t++)
#pragma unroll
        for(int g = 0; g < 4; g++)
          *reinterpret_cast<v4fv *>(a2 + 32 * (4 * wave + t) + 8 * g + 4 * lh) = v4fv{vacc[t][4 * g], vacc[t][4 * g + 1], vacc[t][4 * g + 2], vacc[t][4 * g + 3]};
    }
    __syncthreads();
    {
      if constexpr(!EARLY)
      {
        bb0 = b1[tid];
        bb1 = b1[tid + 256];
      }
#pragma unroll
      for(int q = 0; q < NF; q++)
      {
        float * a2 = reinterpret_cast<float *>(vl + L::A2) + q * HID;
        float * sl1 = reinterpret_cast<float *>(vl + L::SL) + q * HID;
        const float iv = iW1 / sA1[q];
        const float h0 = a2[tid] * iv + bb0, h1 = a2[tid + 256] * iv + bb1;
        const float s0 = (h0 > 0.0f) ? 1.0f : 0.01f, s1 = (h1 > 0.0f) ? 1.0f : 0.01f;
        a2[tid] = h0 * s0;
        a2[tid + 256] = h1 * s1;
        sl1[tid] = s0;
        sl1[tid + 256] = s1;
      }
    }
    __syncthreads(); // slopes and activations of layer 1 are in LDS; every wavefront is done with the layer-0 fragments
    // T1 = slope1 (.) (0.01 C10 + 0.99 W1 (m (.) W0)) into the B fragments of layer 2: this lane holds column l31 and, per tile, the
    // rows (r & 3) + 8 (r >> 2) + 4 lh; the four rows of a register group g = r >> 2 are elements j = 4 lh .. 4 lh + 3 of lane
    // 32 (g & 1) + column in k-step 2 tile + (g >> 1): one 8-byte store per piece
    const float um = 0.99f * iW1 / sD1;
#pragma unroll
    for(int t = 0; t < (VO ? 0 : 4); t++)
#pragma unroll
      for(int g = 0; g < 4; g++)
      {
        const int tile = 4 * wave + t, row0 = 32 * tile + 8 * g + 4 * lh;
        // (C10 lies in this epilogue's order — [tile][g][lh][column][4 rows] — so that a lane's four rows are ONE 16-byte load and a
        // wavefront's request 1 KiB in a row: 16 loads per lane instead of 64 strided ones)
        const v4fv c4 = *reinterpret_cast<const v4fv *>(c10 + ((((size_t)tile * 4 + g) * 2 + lh) * 32 + l31) * 4);
        float cc[4];
#pragma unroll
        for(int i = 0; i < 4; i++) cc[i] = 0.01f * c4[i];
#pragma unroll
        for(int q = 0; q < NF; q++)
        {
          const float * sl1 = reinterpret_cast<const float *>(vl + L::SL) + q * HID;
          f16x4v hi, lo;
#pragma unroll
          for(int i = 0; i < 4; i++)
          {
            _Float16 a, b;
            split_f16x2(sl1[row0 + i] * (cc[i] + acc[q][t][4 * g + i] * um) * sD2, a, b);
            hi[i] = a;
            lo[i] = b;
          }
          unsigned char * dst = vl + L::D2 + q * VJ_DF + (2 * tile + (g >> 1)) * 2048 + (32 * (g & 1) + l31) * 16 + 8 * lh;
          *reinterpret_cast<f16x4v *>(dst) = hi;
          *reinterpret_cast<f16x4v *>(dst + 1024) = lo;
        }
      }
  }
  float sA2[NF];
#pragma unroll
  for(int q = 0; q < NF; q++)
  {
    const float * a2 = reinterpret_cast<const float *>(vl + L::A2) + q * HID;
    sA2[q] = vscale512(a2[tid], a2[tid + 256], red);
    vput_act(vl + L::AF + q * VJ_AF, tid, a2[tid] * sA2[q]);
    vput_act(vl + L::AF + q * VJ_AF, tid + 256, a2[tid + 256] * sA2[q]);
  }
  __syncthreads();
  VPJ_T(3);
  // ---- layer 2: wavefront w owns rows 32 w .. 32 w + 31 (126 live) of every frame
  {
    f32x16v acc[NF], vacc;
#pragma unroll
    for(int r = 0; r < 16; r++) vacc[r] = 0.0f;
#pragma unroll
    for(int q = 0; q < NF; q++)
#pragma unroll
      for(int r = 0; r < 16; r++) acc[q][r] = 0.0f;
    const unsigned char * const avp = (l31 < NF) ? vl + L::AF + l31 * VJ_AF + lh * 16 : vl + L::RED + 64;
    const int avs = (l31 < NF) ? 64 : 0;
    const uint8_t * ap = w2h + (size_t)wave * (32 * 2048) + l * 16;
    // (nine MFMAs per k-step: three k-steps of lead are 900 cycles, less than an L2 round trip under load — seven here; the first
    // seven were requested behind the layer-1 loop)
    auto & st = st2;
    if constexpr(!EARLY)
    {
#pragma unroll
      for(int s3 = 0; s3 < 7; s3++)
      {
        st2[s3][0] = *reinterpret_cast<const v4fv *>(ap + s3 * 2048);
        st2[s3][1] = *reinterpret_cast<const v4fv *>(ap + s3 * 2048 + 1024);
      }
#pragma unroll
      for(int r = 0; r < 16; r++)
      {
        const int row = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * lh;
        bias2[r] = b2[row < OUT6 ? row : 0];
      }
    }
    v4fv vah[2], val[2], dbh[2][NF], dbl[2][NF];
    auto load_lds = [&](int slot, int ks) {
      vah[slot] = *reinterpret_cast<const v4fv *>(avp + ks * avs);
      val[slot] = *reinterpret_cast<const v4fv *>(avp + ks * avs + (avs >> 1));
#pragma unroll
      for(int q = 0; q < (VO ? 0 : NF); q++)
      {
        dbh[slot][q] = *reinterpret_cast<const v4fv *>(vl + L::D2 + q * VJ_DF + ks * 2048 + l * 16);
        dbl[slot][q] = *reinterpret_cast<const v4fv *>(vl + L::D2 + q * VJ_DF + ks * 2048 + 1024 + l * 16);
      }
    };
    load_lds(0, 0);
    for(int k8 = 0; k8 < 32; k8 += 8)
    {
#pragma unroll
      for(int u = 0; u < 8; u++)
      {
        const int ks = k8 + u, kn = ks + 7 < 32 ? ks + 7 : 31;
        st[(u + 7) & 7][0] = *reinterpret_cast<const v4fv *>(ap + kn * 2048);
        st[(u + 7) & 7][1] = *reinterpret_cast<const v4fv *>(ap + kn * 2048 + 1024);
        load_lds((u + 1) & 1, ks + 1 < 32 ? ks + 1 : 31);
        __builtin_amdgcn_sched_barrier(0);
        vacc = vmfma(st[u][0], vah[u & 1], vacc);
        vacc = vmfma(st[u][0], val[u & 1], vacc);
        vacc = vmfma(st[u][1], vah[u & 1], vacc);
#pragma unroll
        for(int q = 0; q < (VO ? 0 : NF); q++)
        {
          acc[q] = vmfma(st[u][0], dbh[u & 1][q], acc[q]);
          acc[q] = vmfma(st[u][0], dbl[u & 1][q], acc[q]);
          acc[q] = vmfma(st[u][1], dbh[u & 1][q], acc[q]);
        }
      }
    }
    __syncthreads(); // every wavefront is done with the tangent blocks: the layer-2 output so [126][33] of frame q takes its place
    const float u2 = iW2 / sD2;
    if(l31 < NF) // the layer-2 values of frame l31: rows (r & 3) + 8 (r >> 2) + 4 lh of this wavefront's tile
    {
      float iv2 = iW2 / sA2[0];
#pragma unroll
      for(int q = 1; q < NF; q++) iv2 = (l31 == q) ? iW2 / sA2[q] : iv2;
      float * so = reinterpret_cast<float *>(vl + L::D2 + l31 * VJ_DF);
#pragma unroll
      for(int r = 0; r < 16; r++)
      {
        const int row = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if(row < OUT6) so[row * 33 + 32] = vacc[r] * iv2 + bias2[r];
      }
    }
#pragma unroll
    for(int q = 0; q < (VO ? 0 : NF); q++)
    {
      float * so = reinterpret_cast<float *>(vl + L::D2 + q * VJ_DF);
#pragma unroll
      for(int r = 0; r < 16; r++)
      {
        const int row = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if(row < OUT6) so[row * 33 + l31] = acc[q][r] * u2;
      }
    }
  }
  __syncthreads();
  VPJ_T(4);
  // ---- rotation tail: 6D -> axis-angle and its 3 x 6 Jacobian, one thread per (frame, joint); the chain rule into the 32 latent
  // columns by all threads
  static_assert(21 * 6 * NF <= 256, "one thread per (frame, joint, direction)");
  if(tid < 21 * 6 * NF) // one thread per (frame, joint, input direction): the value path six times over, one derivative column each
  {
    const int q = tid / (21 * 6), jd = tid % (21 * 6), j = jd / 6, dir = jd % 6;
    const float * so = reinterpret_cast<const float *>(vl + L::D2 + q * VJ_DF);
    float * sj = reinterpret_cast<float *>(vl + L::D2 + q * VJ_DF + 126 * 33 * 4); // [21][18] behind so
    float o6[6], aa[3], jc[3];
    for(int i = 0; i < 6; i++) o6[i] = so[(j * 6 + i) * 33 + 32];
    sixd_to_aa_dir<1>(o6, dir, aa, jc);
    if(out && dir == 0 && f0 + q >= 0 && f0 + q < n)
      for(int i = 0; i < 3; i++) out[(f0 + q) * out_stride + j * 3 + i] = aa[i];
    if constexpr(!VO)
      for(int i = 0; i < 3; i++) sj[j * 18 + i * 6 + dir] = jc[i];
  }
  if constexpr(VO)
  {
    VPJ_T(5);
    VPJ_T(6);
    return;
  }
  __syncthreads();
  VPJ_T(5);
  // one thread per (frame, joint, latent column): the six tangent entries of the column once for the joint's three output rows
  for(int item = tid; item < NF * 21 * LAT; item += 256)
  {
    const int q = item / (21 * LAT), it = item % (21 * LAT);
    if(f0 + q < 0 || f0 + q >= n) continue;
    const float * so = reinterpret_cast<const float *>(vl + L::D2 + q * VJ_DF);
    const float * sj = reinterpret_cast<const float *>(vl + L::D2 + q * VJ_DF + 126 * 33 * 4);
    const int j = it / LAT, c = it % LAT;
    float t6[6];
#pragma unroll
    for(int k = 0; k < 6; k++) t6[k] = so[(j * 6 + k) * 33 + c];
#pragma unroll
    for(int i = 0; i < 3; i++)
    {
      float s = 0.f;
#pragma unroll
      for(int k = 0; k < 6; k++) s += sj[j * 18 + i * 6 + k] * t6[k];
      if(sig_flag) // (read on another stream behind the flag: write-through, signal.h)
        st_agent(&jac[((f0 + q) * 63 + j * 3 + i) * LAT + c], s);
      else
        jac[((f0 + q) * 63 + j * 3 + i) * LAT + c] = s;
    }
  }
  VPJ_T(6);
  wg_signal(sig_flag, sig_counter, sig_tick);
}

// weights [out][in] -> fp16x2 pieces in MFMA fragment order: [ceil(out/32)][in/16][piece 2][64 lanes][8 fp16]
static hipError_t upload_frag(uint8_t ** dst, const float * w, int out, int in, float scale)
{
  const int tiles = (out + 31) / 32, ksn = in / 16;
  std::vector<_Float16> t((size_t)tiles * ksn * 2 * 64 * 8);
  for(int tile = 0; tile < tiles; tile++)
    for(int ks = 0; ks < ksn; ks++)
      for(int lane = 0; lane < 64; lane++)
        for(int j = 0; j < 8; j++)
        {
          const int row = 32 * tile + (lane & 31), k = 16 * ks + 8 * (lane >> 5) + j;
          const float v = row < out ? w[(size_t)row * in + k] * scale : 0.0f;
          _Float16 hi, lo;
          split_f16x2(v, hi, lo);
          const size_t o = ((((size_t)tile * ksn + ks) * 2) * 64 + lane) * 8 + j;
          t[o] = hi;
          t[o + 64 * 8] = lo;
        }
  hipError_t e = hipMalloc((void **)dst, sizeof(_Float16) * t.size());
  if(e != hipSuccess) return e;
  return hipMemcpy(*dst, t.data(), sizeof(_Float16) * t.size(), hipMemcpyHostToDevice);
}

__global__ void rotmat_to_aa_kernel(const float * __restrict__ rot, float * __restrict__ aa_out, int64_t n)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n) return;
  D6 R[3][3], aa[3];
  for(int r = 0; r < 3; r++)
    for(int c = 0; c < 3; c++) R[r][c] = mk<6>(rot[i * 9 + r * 3 + c]);
  rotmat_to_aa(R, aa);
  for(int q = 0; q < 3; q++) aa_out[i * 3 + q] = aa[q].v;
}

// value_like_jac (jac null): the decoded angles by the Jacobian kernel's own value path (vposer_jac2_kernel<NF, true>: bit-identical
// to what a call WITH jac writes to `out`), not by the exact-fp32 value kernel.  sig_flag / sig_counter / sig_tick (jac non-null): the
// launch raises that flag when its last workgroup is done and stores the Jacobian write-through (a consumer on another stream).
int vposer_forward_device(smplpp_vposer * v, int64_t n, const float * z, int64_t z_stride, float * out, int64_t out_stride,
                          float * jac, hipStream_t st, int64_t frame_base, bool value_like_jac, unsigned * sig_flag,
                          unsigned * sig_counter, unsigned sig_tick)
{
  if((jac || value_like_jac) && v->w0h && v->c10)
  {
    // One kernel, two instantiations with the SAME arithmetic per frame: more frames than CUs -> two frames per workgroup share
    // every weight fragment; fewer -> one frame per workgroup.  Either way a frame's bits are a function of the frame and of its
    // GLOBAL index's group (frame_base + local index) / 2 alone: not of the batch size, not of the shard it travels in.
    const bool vo = !jac;
    static PerDeviceOnce once[4];
#define VJ2_(NF_, VO_, GRID_)                                                                                                              \
  do                                                                                                                                       \
  {                                                                                                                                        \
    HIP_TRY(lds_opt_in(once[(NF_ - 1) * 2 + (VO_ ? 1 : 0)], v->device, reinterpret_cast<const void *>(&vposer_jac2_kernel<NF_, VO_>),    \
                       VJ2<NF_>::TOTAL));                                                                                                  \
    vposer_jac2_kernel<NF_, VO_><<<dim3((unsigned)(GRID_)), dim3(256), VJ2<NF_>::TOTAL, st>>>(                                             \
        z, z_stride, v->w0t, v->b0, v->b1, v->b2, v->w1h, v->w2h, v->w0h, v->c10, v->sD1, v->sD2, 1.0f / v->sW1, 1.0f / v->sW2, out,     \
        out_stride, jac, n, frame_base, sig_flag, sig_counter, sig_tick);                                                                  \
  } while(0)
    if(n > device_cus(v->device))
    {
      const int64_t groups = (frame_base % VJ_GROUP + n + VJ_GROUP - 1) / VJ_GROUP;
      if(vo)
        VJ2_(2, true, groups);
      else
        VJ2_(2, false, groups);
    }
    else
    {
      if(vo)
        VJ2_(1, true, n);
      else
        VJ2_(1, false, n);
    }
#undef VJ2_
    HIP_TRY(hipGetLastError());
    return SMPLPP_OK;
  }
  if(jac) return fail(SMPLPP_ERR_INVALID, "smplpp_vposer_forward: this decoder has no Jacobian operands");
  const size_t shmem = sizeof(float) * (size_t)(2 * HID * VS + LAT + HID);
  static PerDeviceOnce once;
  HIP_TRY(lds_opt_in(once, v->device, reinterpret_cast<const void *>(&vposer_kernel), (int)shmem));
  vposer_kernel<<<dim3((unsigned)n), dim3(256), shmem, st>>>(z, z_stride, v->w0t, v->b0, v->w1t, v->b1, v->w2t, v->b2, out, out_stride,
                                                            nullptr, 0);
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
  if(v->w1h) (void)hipFree(v->w1h);
  if(v->w2h) (void)hipFree(v->w2h);
  if(v->w0h) (void)hipFree(v->w0h);
  if(v->c10) (void)hipFree(v->c10);
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
  {
    // fp16x2 operands of the tangent GEMMs: power-of-two scales that keep every piece inside fp16's range.  Weights: the
    // largest entry just under 2^14.  Tangent blocks: |D1| <= max|W0|; |D2| <= 512 max|W1| max|W0| (every slope <= 1).
    auto amax = [](const float * w, size_t cnt) {
      float m = 0.0f;
      for(size_t i = 0; i < cnt; i++) m = std::fmax(m, std::fabs(w[i]));
      return m;
    };
    auto pow2_under = [](float bound, float m) { return std::exp2(std::floor(std::log2(bound / (m > 1e-30f ? m : 1e-30f)))); };
    const float m0 = amax(w0, (size_t)HID * LAT), m1 = amax(w1, (size_t)HID * HID), m2 = amax(w2, (size_t)OUT6 * HID);
    if(!std::isfinite(m0) || !std::isfinite(m1) || !std::isfinite(m2))
    {
      smplpp_vposer_destroy(v);
      return fail(SMPLPP_ERR_INVALID, "smplpp_vposer_create: non-finite weights");
    }
    v->sW1 = pow2_under(16384.0f, m1);
    v->sW2 = pow2_under(16384.0f, m2);
    v->sD1 = pow2_under(16384.0f, m0);
    v->sD2 = pow2_under(16384.0f, 512.0f * m1 * m0);
    if(e == hipSuccess) e = upload_frag(&v->w1h, w1, HID, HID, v->sW1);
    if(e == hipSuccess) e = upload_frag(&v->w2h, w2, OUT6, HID, v->sW2);
    if(e == hipSuccess)
    {
      // W0 [512][32] as the B operand of layer 1 (k = row of W0): [32 k-steps][piece 2][64 lanes (32 h + column)][8 fp16]
      std::vector<_Float16> t((size_t)32 * 2 * 64 * 8);
      for(int row = 0; row < HID; row++)
        for(int c = 0; c < LAT; c++)
        {
          _Float16 hi, lo;
          split_f16x2(w0[(size_t)row * LAT + c] * v->sD1, hi, lo);
          const size_t o = (((size_t)(row >> 4) * 2) * 64 + 32 * ((row >> 3) & 1) + c) * 8 + (row & 7);
          t[o] = hi;
          t[o + 64 * 8] = lo;
        }
      e = hipMalloc((void **)&v->w0h, sizeof(_Float16) * t.size());
      if(e == hipSuccess) e = hipMemcpy(v->w0h, t.data(), sizeof(_Float16) * t.size(), hipMemcpyHostToDevice);
    }
    if(e == hipSuccess)
    {
      std::vector<float> c((size_t)HID * LAT);
      for(int r = 0; r < HID; r++)
        for(int cc = 0; cc < LAT; cc++)
        {
          double sum = 0.0;
          for(int k = 0; k < HID; k++) sum += (double)w1[(size_t)r * HID + k] * (double)w0[(size_t)k * LAT + cc];
          // the layer-1 epilogue's order: row r = 32 tile + 8 g + 4 lh + i of column cc at [tile][g][lh][cc][i]
          c[((((size_t)(r >> 5) * 4 + ((r >> 3) & 3)) * 2 + ((r >> 2) & 1)) * 32 + cc) * 4 + (r & 3)] = (float)sum;
        }
      e = hipMalloc((void **)&v->c10, sizeof(float) * c.size());
      if(e == hipSuccess) e = hipMemcpy(v->c10, c.data(), sizeof(float) * c.size(), hipMemcpyHostToDevice);
    }
  }
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
  return smplpp_vposer_forward_at(v, n, 0, z, out, jac, space, stream);
}

extern "C" int smplpp_vposer_forward_at(smplpp_vposer * v, int64_t n, int64_t frame_base, const float * z, float * out, float * jac,
                                        int space, void * stream)
{
  if(!v || n <= 0 || frame_base < 0 || !z || !out) return fail(SMPLPP_ERR_INVALID, "smplpp_vposer_forward: bad argument");
  int rc = check_space(space, "smplpp_vposer_forward");
  if(rc) return rc;
  HIP_TRY(hipSetDevice(v->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  In<float> zi;
  Out<float> oo, jo;
  HIP_TRY(zi.init(z, (size_t)n * LAT, space, st));
  HIP_TRY(oo.init(out, (size_t)n * 63, space));
  HIP_TRY(jo.init(jac, (size_t)n * 63 * LAT, space));
  rc = vposer_forward_device(v, n, zi.d, LAT, oo.d, 63, jo.d, st, frame_base, false, nullptr, nullptr, 0u);
  if(rc) return rc;
  hipError_t e = oo.finish(st);
  if(e == hipSuccess) e = jo.finish(st);
  if(e == hipSuccess && space == SMPLPP_HOST) e = hipStreamSynchronize(st);
  HIP_TRY(e);
  return SMPLPP_OK;
}

// Development / test hook (not part of include/smplpp_hip.h): the decoded angles by the Jacobian kernel's VALUE-ONLY instantiation —
// what the capture loops use to have theta25 early.  Host pointers.  tests/test_vposer_gpu.py compares it bit for bit with the `out`
// of a call that also asks for the Jacobian.
extern "C" int smplpp_debug_vposer_value(smplpp_vposer * v, int64_t n, int64_t frame_base, const float * z, float * out)
{
  if(!v || n <= 0 || frame_base < 0 || !z || !out) return fail(SMPLPP_ERR_INVALID, "smplpp_debug_vposer_value: bad argument");
  HIP_TRY(hipSetDevice(v->device));
  In<float> zi;
  Out<float> oo;
  HIP_TRY(zi.init(z, (size_t)n * LAT, SMPLPP_HOST, nullptr));
  HIP_TRY(oo.init(out, (size_t)n * 63, SMPLPP_HOST));
  int rc = vposer_forward_device(v, n, zi.d, LAT, oo.d, 63, nullptr, nullptr, frame_base, true, nullptr, nullptr, 0u);
  if(rc) return rc;
  hipError_t e = oo.finish(nullptr);
  if(e == hipSuccess) e = hipStreamSynchronize(nullptr);
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
