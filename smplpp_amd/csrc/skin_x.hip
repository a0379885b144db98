// skin_kernel_x — the exact fused kernel (skin_e.hip's arithmetic, bit for bit) with its two jobs on two wavefronts per SIMD.
// Reference path: as skin_e.hip (src/BlendShape.cpp:670-683, 762-765; src/LinearBlendSkinning.cpp:445-553; src/SMPL.cpp:726-727).
//
// Why: skin_kernel_e's one wavefront per SIMD issues an item's 252 MFMAs AND the previous item's ~1100 vector instructions and
// ~200 LDS reads in order — 11.6-11.9 k cycles per item where the MFMAs alone take 8.5 k.  A second wavefront on the same SIMD issues
// beside the first (tools/micro/mfma_beside_valu.hip, profiles/r06_fk_second_wavefront_probe.txt): here wavefronts 0-3 ("matrix") only
// feed the matrix pipe and the LDS-DMA ring, wavefronts 4-7 ("skinning") only skin and store.
//
// What that needs, and how it fits:
//  * 512 threads = two wavefronts per SIMD = 256 registers each.  The matrix wavefront keeps pieces 1 and 2 of its A fragments
//    (112 registers), re-reads piece 3 (one 1 KiB load per k-step, two k-steps ahead, from L2) and holds ONE set of B fragments,
//    each coordinate's three reloaded as soon as its six MFMAs are issued.  The two roles are two separate loops, so their registers
//    overlap.
//  * The accumulators pass through LDS.  To keep that buffer at 24 KiB the two vertex halves of an item are HALF AN ITEM APART:
//    matrix wavefronts 0, 1 (vertices 0..31 of the group, "pair 0") run seven k-steps ahead of wavefronts 2, 3 (pair 1), each pair
//    streams its own 9 KiB sub-image of B3e's k-step images through its own ring of three, and every seven k-steps (a "phase") one
//    pair hands 64 frames x 32 vertices x 3 over; the four skinning wavefronts share it (16 frames each) and finish it within the phase.
//    LDS: 72 KiB transforms + 24 KiB hand-over + 2 x 3 x 10 KiB rings + 1 KiB root translations = 157 KiB.
//  * One s_barrier per k-step for all eight wavefronts.  A run of m items of one frame tile takes 2 m + 2 phases: pair 1 idles in the
//    first, pair 0 in the last but one (it only hands over), both in the last.
#include "common.h"

#include <cstdlib>
#include <type_traits>
#include <utility>

namespace smplpp_hip
{
namespace
{
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v3f __attribute__((ext_vector_type(3), aligned(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned v3u __attribute__((ext_vector_type(3)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int X_PH = 7;                                // k-steps per phase
constexpr int X_R = 3;                                 // ring images per pair
constexpr int X_IMG = 10 * 1024;                       // a pair's image of a k-step: nine fragment pieces + one spare piece
constexpr int X_SUB = 9 * 1024;                        // a pair's fragments inside a 20 KiB k-step image of B3e: at pair * X_SUB
constexpr int X_G_BYTES = 64 * NJ * 48;                // 73728: G' of 64 frames (a straight copy of Gp)
constexpr int X_LDS_X = X_G_BYTES;                     // hand-over buffer [3 coordinates][64 frames][32 vertices] fp32
constexpr int X_X_BYTES = 3 * 64 * 32 * 4;             // 24576
constexpr int X_LDS_RING = X_LDS_X + X_X_BYTES;        // [2 pairs][X_R][X_IMG]
constexpr int X_LDS_ROOT = X_LDS_RING + 2 * X_R * X_IMG;
constexpr int X_LDS_TOTAL = X_LDS_ROOT + 64 * 16;      // 160768 <= 163840
constexpr int X_SLOTS = 18;                            // MFMAs per k-step
constexpr int X_BAR = 6;                               // slot of a k-step that carries its barrier
constexpr int X_NDMA = 5;                              // ring DMAs per matrix wavefront and k-step (slots 7..11)
constexpr int X_A3_SLOT = 13;                          // slot of the piece-3 load of k-step KS + 2 (behind the k-step's last use of that register: slot 12)
constexpr int X_A_PRE = 4;                             // k-steps of A pieces 1, 2 loaded by the run's set-up; the first item loads the
constexpr int X_A_SLOT = 14;                           // rest, k-step KS + 4 in slot 14 of k-step KS
constexpr int X_GCH = X_G_BYTES / (256 * 16);          // 18 G' DMAs per skinning wavefront
// skinning role: a phase is 7 x 18 = 126 row slots; the lane's eight (frame, vertex) rows take 14 slots each from slot X_ROW0
#ifndef SKINX_RD
#define SKINX_RD 6
#endif
constexpr int X_RD = SKINX_RD;                         // slots between a matrix row's LDS read and the FMA group that uses it
constexpr int X_ROW0 = X_RD < 6 ? 6 : X_RD;
constexpr int X_PITCH = 14;
constexpr int X_ROOT_P = 2;
static_assert(X_LDS_TOTAL <= 160 * 1024, "LDS plan");
static_assert(X_ROW0 + 7 * X_PITCH + 13 < X_PH * X_SLOTS, "eight rows fit a phase");
static_assert(X_ROW0 + 1 - X_RD >= 1, "the first matrix-row read follows the hand-over reads");
#ifndef SKINX_ABL
#define SKINX_ABL 0 // timing ablations (development only; results are wrong when non-zero): 1 skinning wavefronts idle, 4 no MFMA, 8 no piece-3 loads, 16 no ring DMA, 32 no hand-over writes, 64 every ring DMA reads group 0's k-step 0 (L2-resident)
#endif

template<class F, int... I>
__device__ __forceinline__ void xstatic_for_impl(F && f, std::integer_sequence<int, I...>)
{
  (f(std::integral_constant<int, I>{}), ...);
}
template<int N, class F>
__device__ __forceinline__ void xstatic_for(F && f)
{
  xstatic_for_impl(f, std::make_integer_sequence<int, N>{});
}
// piece products in issue order (as skin_e.hip / skin_b.hip: the bits depend on it)
constexpr int X_PA[6] = {2, 0, 1, 1, 0, 0};
constexpr int X_PB[6] = {0, 2, 1, 0, 1, 0};

#define XSB() __builtin_amdgcn_sched_barrier(0)
// matrix wavefronts: behind the barrier of k-step KS the pair's image of k-step KS + 1 is complete — its DMAs were issued in k-step
// KS - 2 (slots 7..11); younger than the last of them are that k-step's piece-3 load (slot 13), the five DMAs and the load of k-step KS - 1:
// seven (a run's first item also loads A pieces there: more operations in flight than counted only waits for more)
__device__ __forceinline__ void x_barrier_matrix()
{
  asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ void x_barrier_plain()
{
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ void x_barrier_bare()
{
  asm volatile("s_barrier" ::: "memory");
}
__device__ __forceinline__ void x_full_barrier()
{
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
} // namespace

template<bool WANT_REST>
__global__ __launch_bounds__(512, 1) void skin_kernel_x(const uint8_t * __restrict__ A3, const uint8_t * __restrict__ B3e,
                                                        const float * __restrict__ Gp, const float * __restrict__ theta,
                                                        float * __restrict__ verts, float * __restrict__ rest, int64_t n, int64_t V,
                                                        int nvg, int nft)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  typedef __attribute__((address_space(3))) void * lds_ptr_t;
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // ---- work assignment: as skin_kernel_e (workgroup b on XCD b & 7, which owns an eighth of the vertex groups; contiguous runs)
  const int nbx = (int)(gridDim.x >> 3), xcd = (int)(blockIdx.x & 7), jb = (int)(blockIdx.x >> 3);
  const int vg0 = (xcd * nvg) >> 3, vg1 = ((xcd + 1) * nvg) >> 3, nvx = vg1 - vg0;
  const int cnt = nvx * nft;
  const int i0 = (int)(((unsigned)jb * (unsigned)cnt) / (unsigned)nbx), i1 = (int)(((unsigned)(jb + 1) * (unsigned)cnt) / (unsigned)nbx);
  if(i0 >= i1) return; // whole workgroup leaves
  const int frameB = (int)(V * 12);

  if(wave < 4)
  {
    // =============================================================== matrix wavefronts
    const int wf = wave & 1, pair = wave >> 1;
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(B3e), 0, (int)(nvg * EB_KS * EB_IMG), 0x00020000);
    // this wavefront's five pieces of its pair's sub-image: pieces wf * 5 + I (the tenth lands in the image's spare KiB)
    const int voffDma = pair * X_SUB + wf * (X_NDMA * 1024) + lane * 16;
    const int ringBase = X_LDS_RING + pair * (X_R * X_IMG);
    const unsigned char * imgV[X_R]; // this lane's fragments of the image of pair-local k-step (base + k): fragment (x, s) at + (3 x + s) * 1024
    int imgS[X_R];
#pragma unroll
    for(int k = 0; k < X_R; k++)
    {
      imgS[k] = ringBase + k * X_IMG;
      imgV[k] = lds + imgS[k] + lane * 16;
    }
    // hand-over: accumulator row r of coordinate c -> X[c][frame wf * 32 + rowc(r) + 4 half][vertex l31]
    unsigned char * const xLane = lds + X_LDS_X + ((wf * 32 + 4 * half) * 32 + l31) * 4;
    f32x16 acc[3];
    v4f a12[EB_KS][2]; // pieces 1, 2 of the A fragments of this wavefront's 32 frames, loaded once per run
    v4f a3[2];         // piece 3 of the A fragments by k-step parity (14 is even: the parity runs on across items); k-step KS + 2's
                       // is loaded behind k-step KS's last use of the register (the three a3.b1 products: slots 0, 6, 12)
    v4f bfr[3][3];     // B fragments [coordinate][piece] of the k-step in progress (a coordinate's three are reloaded behind its MFMAs)
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    acc[0] = acc[1] = acc[2] = zero16;

    auto dma = [&](auto itag, int vgBase, int ks, int dst) {
      constexpr int I = decltype(itag)::value;
      if constexpr(SKINX_ABL & 16) return;
      if constexpr(SKINX_ABL & 64)
      {
        vgBase = 0;
        ks = 0;
      }
      if constexpr(I < 4)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(lds + dst + wf * (X_NDMA * 1024)), 16, voffDma, vgBase + ks * EB_IMG, I * 1024, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(lds + dst + wf * (X_NDMA * 1024) + 4096), 16, voffDma, vgBase + ks * EB_IMG + 4096, 0, 0);
    };

    // one item of this pair: fourteen k-steps, one barrier each (the two pairs run the same code seven barriers apart).  FIRST: the run's
    // first item (nothing to hand over at its start; it loads the A pieces of k-steps 4..13).  Bcur / Bnext: byte bases of this item's and
    // the next item's vertex group in B3e.  (ONE body per kind of item: two half-item bodies joined by a branch made the register
    // allocator spill 249 registers at the join.)
    auto full_item = [&](auto ftag, const uint8_t * ap, int Bcur, int Bnext) {
      constexpr bool FIRST = decltype(ftag)::value;
      xstatic_for<EB_KS * X_SLOTS>([&](auto ss) {
        constexpr int S = decltype(ss)::value;
        constexpr int KS = S / X_SLOTS, KR = KS, M = S % X_SLOTS;
        constexpr int X = M / 6, Q = M % 6;
        constexpr int IMG = KR % X_R, IMGN = (KR + 1) % X_R;
        {
          const v4f av = X_PA[Q] == 2 ? a3[KS & 1] : a12[KS][X_PA[Q]];
          if constexpr(SKINX_ABL & 4)
          {
            if constexpr(KS == 0 && Q == 0) acc[X] = zero16;
            asm volatile("" ::"v"(av), "v"(bfr[X][X_PB[Q]]));
          }
          else if constexpr(KS == 0 && Q == 0)
            acc[X] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bfr[X][X_PB[Q]]), zero16, 0, 0, 0);
          else
            acc[X] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bfr[X][X_PB[Q]]), acc[X], 0, 0, 0);
        }
        XSB();
        if constexpr(M == X_BAR) x_barrier_matrix();
        // ---- fragments: coordinate 2 of THIS k-step in slots 0..2 (its MFMAs are slots 12..17), coordinates 0 / 1 of the NEXT k-step
        // in slots 6..8 / 12..14, each right behind the last MFMA that read the registers they replace
        if constexpr(M < 3) bfr[2][M] = *reinterpret_cast<const v4f *>(imgV[IMG] + (6 + M) * 1024);
        if constexpr(M >= 6 && M < 9) bfr[0][M - 6] = *reinterpret_cast<const v4f *>(imgV[IMGN] + (M - 6) * 1024);
        if constexpr(M >= 12 && M < 15) bfr[1][M - 12] = *reinterpret_cast<const v4f *>(imgV[IMGN] + (3 + M - 12) * 1024);
        XSB();
        // ---- ring DMA: k-step KS + 3 into the image this k-step has just finished with
        if constexpr(M > X_BAR && M <= X_BAR + X_NDMA)
        {
          constexpr int KN = KS + X_R;
          dma(std::integral_constant<int, M - X_BAR - 1>{}, KN < EB_KS ? Bcur : Bnext, KN < EB_KS ? KN : KN - EB_KS, imgS[IMG]);
        }
        // ---- piece 3 of the A fragments of k-step KS + 2 (the same frame tile whatever the item)
        if constexpr(M == X_A3_SLOT && !(SKINX_ABL & 8)) a3[KS & 1] = *reinterpret_cast<const v4f *>(ap + ((KS + 2) % EB_KS) * BB_A_BYTES + 2 * 1024);
        if constexpr(FIRST && M == X_A_SLOT && KS + X_A_PRE < EB_KS)
        {
          a12[KS + X_A_PRE][0] = *reinterpret_cast<const v4f *>(ap + (KS + X_A_PRE) * BB_A_BYTES);
          a12[KS + X_A_PRE][1] = *reinterpret_cast<const v4f *>(ap + (KS + X_A_PRE) * BB_A_BYTES + 1024);
        }
        // ---- hand-over: coordinates 0 and 1 leave in the last k-step (behind their last MFMAs), coordinate 2 of the PREVIOUS item in
        // the first six slots of the next one (its registers are first rewritten in slot 12)
        if constexpr(KS == EB_KS - 1 && M >= 6 && !(SKINX_ABL & 32))
        {
          constexpr int C = (M - 6) / 6, T = (M - 6) % 6, R0 = T < 4 ? 3 * T : 12 + 2 * (T - 4), NR = T < 4 ? 3 : 2;
#pragma unroll
          for(int r = R0; r < R0 + NR; r++)
            *reinterpret_cast<float *>(xLane + C * (64 * 32 * 4) + ((r & 3) + 8 * (r >> 2)) * 128) = acc[C][r];
        }
        if constexpr(!FIRST && KS == 0 && M < 6 && !(SKINX_ABL & 32))
        {
          constexpr int R0 = M < 4 ? 3 * M : 12 + 2 * (M - 4), NR = M < 4 ? 3 : 2;
#pragma unroll
          for(int r = R0; r < R0 + NR; r++)
            *reinterpret_cast<float *>(xLane + 2 * (64 * 32 * 4) + ((r & 3) + 8 * (r >> 2)) * 128) = acc[2][r];
        }
        XSB();
      });
      // the images rotate: fourteen k-steps mod three
      {
        const unsigned char * v0 = imgV[0], * v1 = imgV[1];
        const int s0 = imgS[0], s1 = imgS[1];
        imgV[0] = imgV[2]; imgS[0] = imgS[2];
        imgV[1] = v0; imgS[1] = s0;
        imgV[2] = v1; imgS[2] = s1;
      }
    };

    for(int i = i0; i < i1;)
    {
      const int ft = i / nvx;
      const int iend = (ft + 1) * nvx < i1 ? (ft + 1) * nvx : i1;
      const int m = iend - i, vgF = vg0 + (i - ft * nvx);
      x_full_barrier(); // every wavefront is done with the previous run's images, hand-over buffer and transforms
      const uint8_t * ap = A3 + ((int64_t)ft * EB_KS * 2 + wf) * 3072 + lane * 16;
      // ring prologue of this wavefront's pair: k-steps 0..2 of the run's first group; then the A pieces of the first k-steps
#pragma unroll
      for(int k = 0; k < X_R; k++)
      {
        imgS[k] = ringBase + k * X_IMG;
        imgV[k] = lds + imgS[k] + lane * 16;
      }
      xstatic_for<X_R>([&](auto dd) {
        constexpr int D = decltype(dd)::value;
        xstatic_for<X_NDMA>([&](auto ii) { dma(ii, vgF * (EB_KS * EB_IMG), D, imgS[D]); });
      });
#pragma unroll
      for(int ks = 0; ks < X_A_PRE; ks++)
      {
        a12[ks][0] = *reinterpret_cast<const v4f *>(ap + ks * BB_A_BYTES);
        a12[ks][1] = *reinterpret_cast<const v4f *>(ap + ks * BB_A_BYTES + 1024);
      }
      a3[0] = *reinterpret_cast<const v4f *>(ap + 2 * 1024);
      a3[1] = *reinterpret_cast<const v4f *>(ap + BB_A_BYTES + 2 * 1024);
      // k-step 0 has landed (behind it: two k-steps of DMAs and ten loads)
      asm volatile("s_waitcnt vmcnt(20)\n\ts_barrier" ::: "memory");
#pragma unroll
      for(int q = 0; q < 6; q++) bfr[q / 3][q % 3] = *reinterpret_cast<const v4f *>(imgV[0] + q * 1024);

      // pair 1 is one phase behind pair 0
      if(pair)
#pragma unroll
        for(int b = 0; b < X_PH; b++) x_barrier_plain();
      {
        const int Bf = vgF * (EB_KS * EB_IMG);
        full_item(std::true_type{}, ap, Bf, (m > 1 ? vgF + 1 : vgF) * (EB_KS * EB_IMG));
      }
      for(int j = 1; j < m; j++)
      {
        const int vgc = vgF + j;
        full_item(std::false_type{}, ap, vgc * (EB_KS * EB_IMG), (j + 1 < m ? vgc + 1 : vgc) * (EB_KS * EB_IMG));
      }
      // the last item's third coordinate (its hand-over barrier is the first of the seven that follow), then this pair only keeps
      // the barriers: two phases for pair 0, one for pair 1
#pragma unroll
      for(int r = 0; r < 16; r++) *reinterpret_cast<float *>(xLane + 2 * (64 * 32 * 4) + ((r & 3) + 8 * (r >> 2)) * 128) = acc[2][r];
      for(int b = 0; b < (pair ? 1 : 2) * X_PH; b++) x_barrier_plain();
      i = iend;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the last prefetches land before the wavefront ends
  }
  else
  {
    // =============================================================== skinning wavefronts
    const int e = wave - 4, etid = tid - 256;
    const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Gp), 0, (int)(nft * X_G_BYTES), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(verts, 0, (int)(verts ? n * V * 12 : 0), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(rest, 0, (int)(rest ? n * V * 12 : 0), 0x00020000);
    const int voffG = etid * 16;
    // this lane's rows: vertex l31 of the pair's 32, frames 16 e + 2 R + half (R = 0..7)
    const unsigned char * const xLane = lds + X_LDS_X + ((16 * e + half) * 32 + l31) * 4;      // + c * 8192 + R * 256
    const unsigned char * const gLane = lds + (16 * e + half) * (NJ * 48);                      // + R * 2304 + joint offset + matrix row * 16
    const v4f * const rootLane = reinterpret_cast<const v4f *>(lds + X_LDS_ROOT) + (16 * e + half); // + 2 R

    struct Tab
    {
      int jofs[4];
      float jw[4];
      float winv;
    };
    // skinning tables of vertex 32 pair + l31 of group vg, from B3e's image spares (common.h: k-step 0: jofs | jw, k-step 1: winv)
    auto load_tab = [&](int vg, int pr) {
      Tab t;
      const uint8_t * g = B3e + (int64_t)vg * (EB_KS * EB_IMG) + EB_TAB_OFF;
      const v4i jo = *reinterpret_cast<const v4i *>(g + (pr * 32 + l31) * 16);
      const v4f jv = *reinterpret_cast<const v4f *>(g + 1024 + (pr * 32 + l31) * 16);
      t.jofs[0] = jo.x; t.jofs[1] = jo.y; t.jofs[2] = jo.z; t.jofs[3] = jo.w;
      t.jw[0] = jv.x; t.jw[1] = jv.y; t.jw[2] = jv.z; t.jw[3] = jv.w;
      t.winv = *reinterpret_cast<const float *>(g + EB_IMG + (pr * 32 + l31) * 4);
      return t;
    };

    // one hand-over: 64 frames x 32 vertices of pair `pr`, group vg, frame tile ft.  Begins with the hand-over barrier, keeps the
    // other six of the phase.
    auto skin_phase = [&](int ft, int vg, int pr, const Tab & tb) {
      x_barrier_bare();
      float xr[8][3];
#pragma unroll
      for(int R = 0; R < 8; R++)
#pragma unroll
        for(int c = 0; c < 3; c++) xr[R][c] = *reinterpret_cast<const float *>(xLane + c * (64 * 32 * 4) + R * 256);
      const int v = vg * 64 + pr * 32 + l31;
      const int voff = v < (int)V ? v * 12 + half * frameB : 0x7fffff00;
      const int sb = __builtin_amdgcn_readfirstlane((ft * 64 + 16 * e) * frameB);
      const unsigned char * gj[4];
#pragma unroll
      for(int j = 0; j < 4; j++) gj[j] = gLane + tb.jofs[j];
      float rt0 = 0.f, rt1 = 0.f, rt2 = 0.f, tx = 0.f, ty = 0.f, ox = 0.f, oy = 0.f;
      v4f m0 = {0.f, 0.f, 0.f, 0.f}, m1 = m0, m2 = m0;
      constexpr int NSET = X_RD + 2;
      v4f gq[NSET];
      auto read_group = [&](auto r2tag, auto gtag) {
        constexpr int R2 = decltype(r2tag)::value, G = decltype(gtag)::value;
        gq[(R2 * 12 + G) % NSET] = *reinterpret_cast<const v4f *>(gj[G % 4] + R2 * (2 * NJ * 48) + (G / 4) * 16);
      };
      // slot P (0..13) of row R: the operations of skin_e.hip's row_piece, in its order (the bits depend on it)
      auto row_piece = [&](auto rtag, auto ptag) {
        constexpr int R = decltype(rtag)::value, P = decltype(ptag)::value;
        const float rx = xr[R][0], ry = xr[R][1], rz = xr[R][2];
        if constexpr(P == 0 && WANT_REST)
        {
          v3f ov = {rx, ry, rz};
          __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(v3u, ov), rsR, voff, sb + 2 * R * frameB, 0);
        }
        if constexpr(P >= 1 && P <= 12)
        {
          constexpr int G = P - 1, J = G % 4, MR = G / 4;
          const float w = tb.jw[J];
          const v4f gm = gq[(R * 12 + G) % NSET];
          v4f & mm = (MR == 0 ? m0 : (MR == 1 ? m1 : m2));
          if constexpr(J == 0)
          {
            mm.x = w * gm.x;
            mm.y = w * gm.y;
            mm.z = w * gm.z;
            mm.w = w * gm.w;
          }
          else
          {
            mm.x = __builtin_fmaf(w, gm.x, mm.x);
            mm.y = __builtin_fmaf(w, gm.y, mm.y);
            mm.z = __builtin_fmaf(w, gm.z, mm.z);
            mm.w = __builtin_fmaf(w, gm.w, mm.w);
          }
        }
        if constexpr(P == X_ROOT_P)
        {
          const v4f rt = rootLane[2 * R];
          rt0 = rt.x;
          rt1 = rt.y;
          rt2 = rt.z;
        }
        if constexpr(P == 5) tx = m0.y * ry;
        if constexpr(P == 6) tx = __builtin_fmaf(m0.x, rx, tx);
        if constexpr(P == 7) tx = __builtin_fmaf(m0.z, rz, tx);
        if constexpr(P == 8) tx = tx + m0.w;
        if constexpr(P == 9)
        {
          ox = __builtin_fmaf(tx, tb.winv, rt0);
          ty = m1.y * ry;
        }
        if constexpr(P == 10) ty = __builtin_fmaf(m1.x, rx, ty);
        if constexpr(P == 11) ty = __builtin_fmaf(m1.z, rz, ty);
        if constexpr(P == 12)
        {
          ty = ty + m1.w;
          oy = __builtin_fmaf(ty, tb.winv, rt1);
        }
        if constexpr(P == 13)
        {
          float tz = m2.y * ry;
          tz = __builtin_fmaf(m2.x, rx, tz);
          tz = __builtin_fmaf(m2.z, rz, tz);
          tz = tz + m2.w;
          v3f ov = {ox, oy, __builtin_fmaf(tz, tb.winv, rt2)};
          __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(v3u, ov), rsV, voff, sb + 2 * R * frameB, 0);
          // HAZARD (measured on gfx950, see skin_b.hip): one instruction between a 96-bit buffer store and the next VALU write to its data
          XSB();
          asm volatile("s_nop 1");
          XSB();
        }
      };
      xstatic_for<X_PH * X_SLOTS>([&](auto ss) {
        constexpr int S = decltype(ss)::value;
        // the hand-over buffer is rewritten by the OTHER pair from the last k-step of this phase on: its reads are over before the
        // phase's second barrier (they were issued first)
        if constexpr(S > 0 && S % X_SLOTS == 0)
        {
          if constexpr(S == X_SLOTS)
            x_barrier_plain();
          else
            x_barrier_bare();
        }
        constexpr int S2 = S + X_RD - X_ROW0 - 1;
        if constexpr(S2 >= 0 && S2 / X_PITCH < 8 && S2 % X_PITCH < 12)
          read_group(std::integral_constant<int, S2 / X_PITCH>{}, std::integral_constant<int, S2 % X_PITCH>{});
        constexpr int S1 = S - X_ROW0;
        if constexpr(S1 >= 0 && S1 / X_PITCH < 8)
          row_piece(std::integral_constant<int, S1 / X_PITCH>{}, std::integral_constant<int, S1 % X_PITCH>{});
        XSB();
      });
    };

    for(int i = i0; i < i1;)
    {
      const int ft = i / nvx;
      const int iend = (ft + 1) * nvx < i1 ? (ft + 1) * nvx : i1;
      const int m = iend - i, vgF = vg0 + (i - ft * nvx);
      x_full_barrier(); // (this wavefront's stores of the previous run need not have landed: vmcnt(0) here is only tidy)
      // run set-up: the tile's transforms HBM -> LDS by DMA (18 pieces per skinning wavefront), its root translations
      float tval = 0.0f;
      if(etid < 192)
      {
        const int64_t f = (int64_t)ft * 64 + etid / 3;
        if(f < n) tval = theta[f * ((NJ + 1) * 3) + etid % 3]; // theta[f, 0, :] (src/SMPL.cpp:726-727)
      }
      xstatic_for<X_GCH>([&](auto gg) {
        constexpr int GI = decltype(gg)::value;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsG, (lds_ptr_t)(lds + e * 1024 + GI * 4096), 16, voffG + GI * 4096, ft * X_G_BYTES, 0, 0);
      });
      if(etid < 192) *reinterpret_cast<float *>(lds + X_LDS_ROOT + (etid / 3) * 16 + (etid % 3) * 4) = tval;
      x_barrier_bare(); // (the matrix wavefronts' "k-step 0 has landed")
      Tab tb = load_tab(vgF, 0);
      for(int p = 0; p < 2 * m + 2; p++)
      {
        if(p < 2 || (SKINX_ABL & 1))
        {
          // nothing to skin yet; the transforms and root translations are published by the last barrier of phase 1
#pragma unroll
          for(int b = 0; b < X_PH; b++)
          {
            if(p == 1 && b == X_PH - 1)
              x_full_barrier();
            else
              x_barrier_bare();
          }
        }
        else
        {
          const int h = p - 2; // hand-over h: item h / 2, pair h % 2
          const Tab cur = tb;
          if(p + 1 < 2 * m + 2) tb = load_tab(vgF + ((h + 1) >> 1), (h + 1) & 1); // the next hand-over's tables (a phase ahead)
          skin_phase(ft, vgF + (h >> 1), h & 1, cur);
        }
      }
      i = iend;
    }
  }
}

namespace
{
template<bool WANT_REST>
hipError_t launch_x(const smplpp_model * m, int64_t n, const float * theta, float * verts, float * rest, hipStream_t st, int64_t f_off)
{
  const int nft = (int)((n + 63) / 64);
  const int nvg = (int)m->VGPn;
  const int per_xcd_items = ((nvg + 7) / 8) * nft;
  int nbx = device_cus(m->device) / 8;
  if(nbx > per_xcd_items) nbx = per_xcd_items;
  if(nbx < 1) nbx = 1;
  {
    const int rounds = (per_xcd_items + nbx - 1) / nbx;
    nbx = (per_xcd_items + rounds - 1) / rounds;
  }
  static PerDeviceOnce once;
  {
    hipError_t e = lds_opt_in(once, m->device, reinterpret_cast<const void *>(&skin_kernel_x<WANT_REST>), X_LDS_TOTAL);
    if(e != hipSuccess) return e;
  }
  skin_kernel_x<WANT_REST><<<dim3(nbx * 8), dim3(512), X_LDS_TOTAL, st>>>(
      m->ws.A3.as<uint8_t>() + (f_off / 64) * (int64_t)(BB_KS * BB_A_BYTES), m->B3e, m->ws.Gp.as<float>() + f_off * (NJ * 12),
      theta + f_off * ((NJ + 1) * 3), verts ? verts + f_off * m->V * 3 : nullptr, rest ? rest + f_off * m->V * 3 : nullptr, n, m->V, nvg, nft);
  return hipGetLastError();
}
} // namespace

// the exact form on two wavefronts per SIMD (models with at most four weights per vertex, as launch_skin_exact)
hipError_t launch_skin_split(const smplpp_model * m, int64_t n, const float * theta, float * verts, float * rest, hipStream_t st)
{
  int64_t per = (0x7fffff00LL / (m->V * 12)) & ~63LL;
  const int64_t per_g = (0x7fffff00LL / X_G_BYTES) * 64;
  if(per > per_g) per = per_g;
  if(per < 64) return hipErrorInvalidValue;
  for(int64_t off = 0; off < n; off += per)
  {
    const int64_t nn = (n - off < per) ? n - off : per;
    hipError_t e = rest ? launch_x<true>(m, nn, theta, verts, rest, st, off) : launch_x<false>(m, nn, theta, verts, rest, st, off);
    if(e != hipSuccess) return e;
  }
  return hipSuccess;
}
} // namespace smplpp_hip
