// skin_kernel_h — the fused blend-shape GEMM + linear blend skinning kernel on the f16 matrix pipe, fp32-level accuracy.
// Reference path: rest = T + S.beta + P.c (/root/reference/src/BlendShape.cpp:670-683, 762-765,
// src/JointRegression.cpp:551-565) and the skinning of src/LinearBlendSkinning.cpp:445-553 (+ src/SMPL.cpp:726-727).
//
// Operands: every fp32 value is carried as TWO fp16 pieces of a power-of-two multiple of it (common.h, "fp16x2"): 22
// significant bits, and a product costs THREE v_mfma_f32_32x32x16_f16 (lo.hi + hi.lo + hi.hi) instead of the six of the
// bf16x3 form (skin_b.hip).  Error of the representation alone, measured at real-SMPL magnitudes (|posedirs| <= 5e-2,
// |beta| <= 3, 1.5 rad rotations): 3e-7 m, below the accumulation error of a plain fp32 GEMM of the same data
// (tests/test_fp16x2_budget.py).
//
// Skinning on the matrix pipe.  M[f, v, e] = sum_j W[v, j] G'[f, j, e] (e = one of the 12 entries of the 3 x 4 transform)
// is a [frames x 24] . [24 x vertices] product per entry: the same fp16x2 scheme, 5 MFMAs per entry (K = 24: k-step 0
// takes three products, the half-live k-step 1 carries hi.hi + hi.lo in one MFMA and lo.hi in another), accumulator layout identical to the blend-shape GEMM's (row = frame, column = vertex), so
// out = M . [rest; 1] is 5 VALU instructions per (frame, vertex, coordinate) on registers.  This replaces the ~65 VALU
// instructions + 12 LDS gathers per (frame, vertex) that bound the bf16x3 form (its MFMA wavefronts issued 1079 VALU + 376
// LDS instructions per item), handles any number of weights per vertex, and needs no per-lane joint tables.
//
// Work item: 64 frames x 64 vertices, one 256-thread workgroup (one wavefront per SIMD, 2 x 2 wavefronts of 32 x 32).
// A workgroup owns a FRAME TILE for a run of consecutive vertex groups: its A operand (coefficients, 32 frames x 224 k x 2
// pieces = 112 VGPRs per lane) is loaded ONCE into registers, the G' operand of the frame tile (72 KiB) stays resident in
// LDS, and only the basis streams: one 12 KiB image per k-step through a ring of 7 LDS images filled by LDS-DMA
// (buffer_load_dwordx4 ... lds) seven slots ahead; slot 14 of an item carries the group's skinning weights.
// One raw s_barrier per slot publishes the next image (counted vmcnt: DMAs stay in flight across it).
// XCD x owns an eighth of the vertex groups (its slice of B2h, 2.4 MB, lives in that XCD's L2 and is read from HBM once).
// Round 5: a vertex group is 64 CONSECUTIVE vertices with a skinning class (common.h, HB_PERM_OFF: which k-steps of the skinning
// product its vertices' weights touch; the groups are dealt over and inside the XCD slices with the classes interleaved; slot 14
// carries the group's vertex ids and its flags).  A group whose weights live in one k-step runs the skinning phase in the instantiation that issues only that k-step's
// MFMAs and G' fragment reads — 3 (joints 0..15 only) or 2 (joints 16..23 only) MFMAs per entry instead of 5; exact zeros skipped.
#include "common.h"

#include <type_traits>
#include <utility>

namespace smplpp_hip
{
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v3f __attribute__((ext_vector_type(3), aligned(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v3u __attribute__((ext_vector_type(3)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int H_R = 7;                                  // ring images (15 slots per item: the ring rotates by one per item)
constexpr int H_LDS_RING = HB_G_BYTES;                  // [0, 72 KiB): G' operand of the frame tile
constexpr int H_LDS_TR = H_LDS_RING + H_R * HB_IMG;     // root translations of the 64 frames, (x, y, z, -) each
constexpr int H_LDS_TOTAL = H_LDS_TR + 64 * 16;         // 160768 <= 163840

template<class F, int... I>
__device__ __forceinline__ void hstatic_for_impl(F && f, std::integer_sequence<int, I...>)
{
  (f(std::integral_constant<int, I>{}), ...);
}
template<int N, class F>
__device__ __forceinline__ void hstatic_for(F && f)
{
  hstatic_for_impl(f, std::make_integer_sequence<int, N>{});
}

// ---- the tail of an item (its last 16 vertex stores) is spread over the 14 GEMM slots of the NEXT item: all workgroups
// run in step, so stores issued together arrive together — 16 stores per wavefront inside two slots were a 12.6 MB burst
// per item that the memory side absorbed in ~3000 cycles of store-issue stalls (measured: GEMM phase 38 -> 65 cycles per
// MFMA).  Position k of the tail (row k) sits in slot h_tail_slot(k) behind MFMA h_tail_m(k).
constexpr int wrap15(int u)
{
  return u < 0 ? u + HB_SLOTS : u;
}
constexpr int h_tail_slot(int k)
{
  return k <= 4 ? k : (k == 5 ? 4 : (k <= 10 ? k - 1 : (k == 11 ? 9 : k - 2)));
}
constexpr int h_tail_m(int k)
{
  return (k == 5 || k == 11) ? 8 : 5;
}
constexpr int h_tail_rows_in(int u) // tail rows issued in GEMM slot u
{
  int c = 0;
  for(int k = 0; k < 16; k++) c += h_tail_slot(k) == u;
  return c;
}
// ---- vector-memory bookkeeping for the counted waits.  Per slot u of an item the stream issues, in this order, behind the
// slot's barrier: 3 DMAs (the image of slot u + 7), then
//   u < 14 and the item carries the previous item's tail (ht): its vertex stores of that slot (1 or 2)
//   u < 9 and the item follows a frame tile set-up (!ht): 2 DMAs of the new G' image (18 in all; first reader: slot 13)
//   u == 14 and `rest` is wanted: 16 rest stores (blend phase)
constexpr int h_after_dma(int u, bool ht, bool rest)
{
  return u < 14 ? (ht ? h_tail_rows_in(u) : (u < 9 ? 2 : 0)) : ((u == 14 && rest) ? 16 : 0);
}
constexpr int h_slot_ops(int u, bool ht, bool rest)
{
  return 3 + h_after_dma(u, ht, rest);
}
// vmcnt of the barrier of slot s: the image of slot s + 1 was DMA'd behind the barrier of slot s - 6; everything issued
// after those DMAs may stay in flight.
// Slots of the previous item: it may or may not have carried a tail itself, so none of its tail stores are counted (the
// count must never exceed what was really issued behind the DMAs it protects; a smaller count only waits for older
// operations, all issued thousands of cycles earlier).  An item behind a frame tile set-up (!ht) has no previous item in
// flight at all: the set-up left the ring's DMAs only.
constexpr int h_barrier_vmcnt(int s, bool ht, bool rest)
{
  if(!ht && s == 13) return 12; // the barrier in front of the first G' reads: only the ring DMAs of slots 9..12 are younger than the image
  int c = s - 6 < 0 ? ((wrap15(s - 6) == 14 && rest && ht) ? 16 : 0) : h_after_dma(s - 6, ht, rest);
  for(int u = s - 5; u < s; u++) c += u < 0 ? 3 + ((wrap15(u) == 14 && rest && ht) ? 16 : 0) : h_slot_ops(u, ht, rest);
  return c < 63 ? c : 63;
}
static_assert(h_tail_rows_in(4) == 2 && h_tail_rows_in(9) == 2 && h_tail_rows_in(13) == 1 && h_tail_rows_in(14) == 0, "tail placement");
static_assert(h_barrier_vmcnt(0, false, true) == 15 && h_barrier_vmcnt(1, false, false) == 12 + 5 && h_barrier_vmcnt(0, true, true) == 31 &&
                  h_barrier_vmcnt(1, true, false) == 5 * 3 + 1 && h_barrier_vmcnt(8, true, false) == 15 + 7 && h_barrier_vmcnt(14, false, false) == 2 + 15,
              "window bookkeeping");

template<int VM>
__device__ __forceinline__ void h_barrier()
{
#if SKINH_ABL & 1
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(VM) : "memory");
#else
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(VM) : "memory");
#endif
}
__device__ __forceinline__ void h_full_barrier()
{
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
#define HSB() __builtin_amdgcn_sched_barrier(0)
#ifndef SKINH_STORE_AUX
#define SKINH_STORE_AUX 0 // cache policy of the output stores (bit 0 sc0, bit 1 nt, bit 4 sc1)
#endif
#ifndef SKINH_DMA_IMM
#define SKINH_DMA_IMM 1
#endif
#ifndef SKINH_ABL
#define SKINH_ABL 0 // timing ablations (development only; results are wrong when non-zero): 1 no s_barrier, 2 no DMA, 4 no GEMM MFMAs, 8 no blend phase, 16 no fragment reads, 32 no stores, 256 slot timestamps, 512 phase timestamps (cycle counter + 100 MHz real-time counter)
#endif
#if SKINH_ABL & (256 | 512)
__device__ unsigned long long g_hslot_times[8 * 256];
__device__ unsigned long long g_hwg_times[256 * 4];
#endif

template<bool WANT_REST>
__global__ __launch_bounds__(256, 1) void skin_kernel_h(const uint8_t * __restrict__ A2h, const uint8_t * __restrict__ B2h,
                                                        const uint8_t * __restrict__ G2h, const float * __restrict__ theta,
                                                        float * __restrict__ verts, float * __restrict__ rest, int64_t n,
                                                        int64_t V, int nvg, int nft, float cAB)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  typedef __attribute__((address_space(3))) void * lds_ptr_t;
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wf = wave & 1, wv = wave >> 1;
  // ---- work assignment.  Workgroup b runs on XCD b & 7 (round-robin dispatch; a wrong guess costs speed only).  XCD x owns
  // vertex groups [vg0, vg1); its items, frame tile major, are cut into contiguous runs, one per workgroup: a run stays
  // inside one frame tile as long as possible (the A registers and the G' image are reloaded when the frame tile changes).
  const int nbx = (int)(gridDim.x >> 3), xcd = (int)(blockIdx.x & 7), jb = (int)(blockIdx.x >> 3);
  const int vg0 = (xcd * nvg) >> 3, vg1 = ((xcd + 1) * nvg) >> 3, nvx = vg1 - vg0;
  const int cnt = nvx * nft;
  const int i0 = (int)(((unsigned)jb * (unsigned)cnt) / (unsigned)nbx), i1 = (int)(((unsigned)(jb + 1) * (unsigned)cnt) / (unsigned)nbx); // cnt < 2^26
  if(i0 >= i1) return; // whole workgroup leaves

#if SKINH_ABL & 512
  const unsigned long long t_start = __builtin_amdgcn_s_memtime(), r_start = __builtin_amdgcn_s_memrealtime();
#endif
  const int frameB = (int)(V * 12);
  const __amdgpu_buffer_rsrc_t rsB =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(B2h), 0, (int)(nvg * HB_SLOTS * HB_IMG), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsG =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(G2h), 0, (int)(nft * HB_G_BYTES), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(verts, 0, (int)(verts ? n * V * 12 : 0), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(rest, 0, (int)(rest ? n * V * 12 : 0), 0x00020000);
  const int voffDma = wave * 3072 + lane * 16; // this wavefront's three pieces of an image: + i * 1024

  // LDS addresses.  imgV[k]: this lane's B fragments in ring image k (fragment (x, p): + (2 x + p) * 1024); imgS[k]: the
  // image's byte offset (DMA destination of this wavefront: + wave * 3072 + i * 1024).  Rotated by one at every item.
  const unsigned char * imgV[H_R];
  int imgS[H_R];
#pragma unroll
  for(int k = 0; k < H_R; k++)
  {
    imgS[k] = H_LDS_RING + k * HB_IMG;
    imgV[k] = lds + imgS[k] + wv * 6144 + lane * 16;
  }
  const unsigned char * const gLane0 = lds + wf * (12 * 3072) + lane * 16;       // entry e, k-step 0, piece p: + e * 3072 + p * 1024
  const unsigned char * const gLane1 = lds + wf * (12 * 3072) + 2048 + l31 * 16; // entry e, k-step 1, piece p: + e * 3072 + p * 512
  const unsigned char * const trLane = lds + H_LDS_TR + (wf * 32 + 4 * half) * 16; // accumulator row R: + rowc(R) * 16

#if SKINH_ABL & (256 | 512)
  int dbg_item = 0;
#endif
  f32x16 acc[3], macc[3]; // (macc: entry E accumulates in macc[E % 3]; its rows are consumed until the first gap of entry E + 2)
  float tt[3][16]; // [coordinate][accumulator row]: sum_c M[x][c] rest[c] (+ the translation entry), unscaled; the tail applies cw
  v4f areg[HB_KS][2]; // [k-step][piece]: this wavefront's 32 frames, loaded once per frame tile
  v4f bfr[2][3][2];   // [k-step parity][coordinate][piece]: the six fragments of k-step S + 1 are read at MFMAs 2..4 of k-step S,
                      // so that they are seven MFMAs old at the next slot's barrier (its lgkmcnt(0) then never waits) and at their first use
  v4f gfr[2][4];      // [entry parity][2 ks + piece]: likewise, the fragments of entry E + 1 at MFMAs 0..3 of entry E
  v4f wfr[4];         // [2 ks + piece]
  v4f trb[2];         // root translations of the tail's rows in flight
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  struct Item
  {
    int voff; // byte offset of (frame 4 * half, vertex v) in an output array; out of range when the lane has no vertex
    int sb;   // byte offset of the wavefront's first frame (wave-uniform)
    float cw; // 1 / (sG sW sum_j W[v, j])
  } cur = {0x7fffff00, 0, 0.f}, prev = {0x7fffff00, 0, 0.f};

  auto mfma = [](const v4f & a, const v4f & b, const f32x16 & c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  };
  // piece i (0..2) of this wavefront's share of slot `slot` of vertex group `vg`, HBM/L2 -> ring image at byte offset dst
  auto dma = [&](auto itag, int vgBase, int slot, int dst) {
    constexpr int I = decltype(itag)::value;
    if constexpr(SKINH_ABL & 2) return;
    // the instruction's immediate offset moves the global address AND the LDS address: piece I of a wavefront's share lies
    // I KiB further in both, so the three DMAs of a slot share one M0 and one soffset (two scalar set-ups fewer per piece)
#if SKINH_DMA_IMM
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(lds + dst + wave * 3072), 16, voffDma, vgBase + slot * HB_IMG, I * 1024, 0);
#else
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(lds + dst + wave * 3072 + I * 1024), 16, voffDma,
                                             vgBase + slot * HB_IMG + I * 1024, 0, 0);
#endif
  };

  // row R of the tail of the previous item: verts = cw (M_t + cAB M_rot . acc) + root translation, in two halves that ride in
  // two consecutive MFMA gaps (the arithmetic, then the store): together they are ~10 vector instructions, and a gap hides six
  v3f ovh;
  auto tail_prep = [&](auto rtag) {
    constexpr int R = decltype(rtag)::value;
    const v4f tr = trb[R & 1];
    if constexpr(R + 1 < 16) trb[(R + 1) & 1] = *reinterpret_cast<const v4f *>(trLane + ((((R + 1) & 3) + 8 * ((R + 1) >> 2)) * 16));
    const float u = __builtin_fmaf(tt[2][R], cAB, macc[2][R]); // (entry 11: macc[11 % 3])
    ovh = v3f{__builtin_fmaf(tt[0][R], prev.cw, tr.x), __builtin_fmaf(tt[1][R], prev.cw, tr.y), __builtin_fmaf(u, prev.cw, tr.z)};
  };
  auto tail_store = [&](auto rtag) {
    constexpr int R = decltype(rtag)::value;
    constexpr int ROWC = (R & 3) + 8 * (R >> 2);
    // write-once output; the descriptor's range check drops frames >= n and vertex-less lanes
    if constexpr(SKINH_ABL & 32)
    {
      const float o0 = ovh.x, o1 = ovh.y, o2 = ovh.z; // (timing ablation: the row is computed, not stored)
      asm volatile("" ::"v"(o0), "v"(o1), "v"(o2));
    }
    else
      __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(v3u, ovh), rsV, prev.voff, prev.sb + ROWC * frameB, SKINH_STORE_AUX);
    // HAZARD (measured on gfx950, see skin_b.hip): keep one instruction between a 96-bit buffer store and the next VALU
    // write to its data registers
    asm volatile("s_nop 1");
  };
  auto tail_row = [&](auto rtag) {
    tail_prep(rtag);
    tail_store(rtag);
  };
  auto standalone_tail = [&]() {
    hstatic_for<16>([&](auto rr) {
      tail_row(rr);
      HSB();
    });
  };

  // ---- frame tile set-up: root translations into LDS, A fragments into registers (plain loads: the compiler waits for each
  // at its first use).  The G' image of the frame tile is DMA'd by the item that follows, two pieces per slot in its
  // slots 0..8 (its first reader is that item's slot 13): every CU starts at the same time, and what a workgroup pulls
  // before its first MFMA is served at ~20 B/clk per CU (measured: 68 loads issued back to back took 8000 cycles to ISSUE).
  // FIRST (the workgroup's first item): also the ring's prologue, slots 0..6 into images 0..6; only slot 0 is waited for.
  auto load_frame_tile = [&](int ft, auto first_tag) {
    constexpr bool FIRST = decltype(first_tag)::value;
    h_full_barrier();
#if SKINH_ABL & 512
    if(FIRST && blockIdx.x == 8 && tid == 0) g_hslot_times[70] = __builtin_amdgcn_s_memtime();
#endif
    float tval = 0.0f;
    if(tid < 192)
    {
      const int64_t f = (int64_t)ft * 64 + tid / 3;
      if(f < n) tval = theta[f * ((NJ + 1) * 3) + tid % 3]; // theta[f, 0, :] (src/SMPL.cpp:726-727)
    }
    const int vgF = vg0 + i0 % nvx;
    if constexpr(FIRST) // slot 0 ahead of everything else (completion is in order): the first MFMAs wait for it and for
    {                   // the first A fragments only
      dma(std::integral_constant<int, 0>{}, vgF * (HB_SLOTS * HB_IMG), 0, imgS[0]);
      dma(std::integral_constant<int, 1>{}, vgF * (HB_SLOTS * HB_IMG), 0, imgS[0]);
      dma(std::integral_constant<int, 2>{}, vgF * (HB_SLOTS * HB_IMG), 0, imgS[0]);
    }
    const uint8_t * ap = A2h + ((int64_t)ft * HB_KS * 2 + wf) * 2048 + lane * 16;
#pragma unroll
    for(int ks = 0; ks < HB_KS; ks++)
#pragma unroll
      for(int p = 0; p < 2; p++) areg[ks][p] = *reinterpret_cast<const v4f *>(ap + ks * HB_A_BYTES + p * 1024);
    if constexpr(FIRST)
      hstatic_for<H_R - 1>([&](auto dd) {
        constexpr int D = decltype(dd)::value + 1;
        dma(std::integral_constant<int, 0>{}, vgF * (HB_SLOTS * HB_IMG), D, imgS[D]);
        dma(std::integral_constant<int, 1>{}, vgF * (HB_SLOTS * HB_IMG), D, imgS[D]);
        dma(std::integral_constant<int, 2>{}, vgF * (HB_SLOTS * HB_IMG), D, imgS[D]);
      });
    if constexpr(FIRST)
    {
      h_barrier<28 + 18>(); // behind slot 0: the A loads and slots 1..6 may stay in flight
#pragma unroll
      for(int q = 0; q < 6; q++) bfr[0][q / 2][q % 2] = *reinterpret_cast<const v4f *>(imgV[0] + q * 1024);
    }
    if(tid < 192) *reinterpret_cast<float *>(lds + H_LDS_TR + (tid / 3) * 16 + (tid % 3) * 4) = tval;
#if SKINH_ABL & 512
    if(FIRST && blockIdx.x == 8 && tid == 0) g_hslot_times[72] = __builtin_amdgcn_s_memtime();
#endif
  };

  // ---- one work item.  HT (compile time): the tail of the previous item (its last 16 stores) rides in slots 0 and 1.
  // (ft, vg): the item's frame tile and vertex group; vgn: the vertex group of the item after it (the same group again behind the
  // workgroup's last item: its prefetches land in images nobody reads).  All three are carried by the caller from item to
  // item: the two divisions and the remainder by a run-time divisor that derived them here cost ~400 cycles per item
  auto do_item = [&](int ft_in, int vg_in, int vgn_in, auto ht_tag) {
    constexpr bool HT = decltype(ht_tag)::value;
    const int ft = __builtin_amdgcn_readfirstlane(ft_in), vg = __builtin_amdgcn_readfirstlane(vg_in), vgn = __builtin_amdgcn_readfirstlane(vgn_in);
    const int Bcur = vg * (HB_SLOTS * HB_IMG), Bnext = vgn * (HB_SLOTS * HB_IMG);
    // (cur.voff — where this lane's vertex goes in the outputs — comes from the group's perm[] in slot 14, at k-step 13: its first
    // user is the rest store of entry 2 of this item's skinning phase)
    cur.sb = __builtin_amdgcn_readfirstlane((ft * 64 + wf * 32) * frameB);
    int gcls = 3;

#if SKINH_ABL & 512
    if(blockIdx.x == 8 && tid == 0 && dbg_item < 8)
    {
      g_hslot_times[dbg_item * 8 + 0] = __builtin_amdgcn_s_memtime();
      g_hslot_times[dbg_item * 8 + 1] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    // ---- blend-shape GEMM: 14 k-steps of 9 MFMAs.  Per coordinate: hi.hi, hi.lo, lo.hi; the two B fragments of a coordinate
    // are re-read (for the next k-step) right behind their last MFMA, six MFMAs ahead of their next use.
    hstatic_for<HB_KS>([&](auto ss) {
      constexpr int S = decltype(ss)::value;
      hstatic_for<9>([&](auto mm) {
        constexpr int M = decltype(mm)::value, X = M / 3, Q = M % 3; // Q: 0 hi.hi, 1 Ahi.Blo, 2 Alo.Bhi
        const v4f & a = areg[S][Q == 2 ? 1 : 0];
        const v4f & b = bfr[S & 1][X][Q == 1 ? 1 : 0];
#if SKINH_ABL & 256
        if(blockIdx.x == 0 && tid == 0 && dbg_item < 8) g_hslot_times[dbg_item * 256 + S * 9 + M] = __builtin_readcyclecounter();
#endif
#if SKINH_ABL & 512
        if(M == 0 && S < 8 && blockIdx.x == 8 && tid == 0 && dbg_item == 0) g_hslot_times[80 + S] = __builtin_amdgcn_s_memtime();
#endif
        if constexpr(SKINH_ABL & 4)
        {
          if constexpr(S == 0 && Q == 0) acc[X] = zero16;
          asm volatile("" ::"v"(a), "v"(b));
        }
        else if constexpr(S == 0 && Q == 0)
          acc[X] = mfma(a, b, zero16);
        else
          acc[X] = mfma(a, b, acc[X]);
        HSB();
        // barrier of the slot: behind it image (S + 1) % 7 holds slot S + 1 (its DMAs have landed: vmcnt) and image S % 7 is
        // free for slot S + 7 (every wavefront's reads of it completed: lgkmcnt)
        if constexpr(M == 1) h_barrier<h_barrier_vmcnt(S, HT, WANT_REST)>();
        if constexpr(M >= 2 && M <= 4)
          dma(std::integral_constant<int, M - 2>{}, S + 7 < HB_SLOTS ? Bcur : Bnext, S + 7 < HB_SLOTS ? S + 7 : S + 7 - HB_SLOTS,
              imgS[S % H_R]);
        if constexpr(!HT && S < 9 && (M == 5 || M == 6)) // the G' image of a new frame tile: pieces 8 S + 4 (M - 5) + wave
        {
          constexpr int PC = (2 * S + (M - 5)) * 4;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsG, (lds_ptr_t)(lds + (PC + wave) * 1024), 16, lane * 16,
                                                   ft * HB_G_BYTES + (PC + wave) * 1024, 0, 0);
        }
        if constexpr(HT && (M == 5 || M == 7)) // a tail row placed at MFMA 5 / 8: arithmetic behind MFMA 5 / 7 ...
          hstatic_for<16>([&](auto kk) {
            constexpr int K = decltype(kk)::value;
            if constexpr(h_tail_slot(K) == S && h_tail_m(K) == (M == 5 ? 5 : 8)) tail_prep(kk);
          });
        if constexpr(HT && (M == 6 || M == 8)) // ... store behind MFMA 6 / 8
          hstatic_for<16>([&](auto kk) {
            constexpr int K = decltype(kk)::value;
            if constexpr(h_tail_slot(K) == S && h_tail_m(K) == (M == 6 ? 5 : 8)) tail_store(kk);
          });
        if constexpr(S < HB_KS - 1 && M >= 2 && M <= 4 && !(SKINH_ABL & 16)) // operand fragments of the next k-step, into the other register set
        {
          bfr[(S + 1) & 1][M - 2][0] = *reinterpret_cast<const v4f *>(imgV[(S + 1) % H_R] + (2 * (M - 2)) * 1024);
          bfr[(S + 1) & 1][M - 2][1] = *reinterpret_cast<const v4f *>(imgV[(S + 1) % H_R] + (2 * (M - 2) + 1) * 1024);
        }
        if constexpr(S == HB_KS - 1)
        {
          // slot 14's image: skinning weights of this vertex group (fragments) and the lane's cw; first G' fragments
          const unsigned char * wimg = imgV[HB_KS % H_R] - wv * 4096; // = image + wv * 2048 + lane * 16
          if constexpr(M >= 2 && M <= 5) wfr[M - 2] = *reinterpret_cast<const v4f *>(wimg + ((M - 2) / 2) * 4096 + ((M - 2) % 2) * 1024);
          if constexpr(M == 6)
          {
            cur.cw = *reinterpret_cast<const float *>(lds + imgS[HB_KS % H_R] + HB_CW_OFF + (wv * 32 + l31) * 4);
            const int pv = *reinterpret_cast<const int *>(lds + imgS[HB_KS % H_R] + HB_PERM_OFF + (wv * 32 + l31) * 4);
            cur.voff = pv >= 0 ? pv * 12 + (4 * half) * frameB : 0x7fffff00; // (V * 12 * 64 frames < 2^31: launch_skin_f16x2 cuts the batch)
            gcls = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int *>(lds + imgS[HB_KS % H_R] + HB_FLAGS_OFF));
          }
          if constexpr(M == 7)
          {
            gfr[0][0] = *reinterpret_cast<const v4f *>(gLane0);
            gfr[0][1] = *reinterpret_cast<const v4f *>(gLane0 + 1024);
          }
          if constexpr(M == 8)
          {
            gfr[0][2] = *reinterpret_cast<const v4f *>(gLane1);
            gfr[0][3] = *reinterpret_cast<const v4f *>(gLane1 + 512);
          }
        }
        HSB();
      });
    });

#if SKINH_ABL & 512
    if(blockIdx.x == 8 && tid == 0 && dbg_item < 8)
    {
      g_hslot_times[dbg_item * 8 + 2] = __builtin_amdgcn_s_memtime();
      g_hslot_times[dbg_item * 8 + 3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    // ---- skinning: 12 entries of 5 MFMAs; the VALU work of entry E - 1 rides behind the MFMAs of entry E.  K = 24 joints:
    // k-step 0 (joints 0..15) takes the three piece products; of k-step 1 only eight k are live (joints 16..23) and both lane
    // halves read the same G' piece, so Ghi1.Whi1 + Ghi1.Wlo1 is ONE MFMA against the weight fragment [Whi1 | Wlo1]
    // CLS = the group's flags: 1 joints 0..15 only (k-step 0: MFMAs 0, 1, 3), 2 joints 16..23 only (k-step 1: MFMAs 2, 4), 3 both
    auto skin_phase = [&](auto ctag) {
    constexpr int CLS = decltype(ctag)::value;
    constexpr bool K0 = (CLS & 1) != 0, K1 = (CLS & 2) != 0;
    hstatic_for<12>([&](auto ee) {
      constexpr int E = decltype(ee)::value, MPE = E & 1, ME = E % 3;
      hstatic_for<5>([&](auto bb) {
        constexpr int B = decltype(bb)::value;
        constexpr bool LIVE = (B == 0 || B == 1 || B == 3) ? K0 : K1; // this position's MFMA belongs to a k-step the group uses
        constexpr int FIRST = K0 ? 0 : 2;                             // the entry's first MFMA (starts from zero)
        // B: 0 Ghi0.Whi0, 1 Ghi0.Wlo0, 2 Ghi1.[Whi1 | Wlo1], 3 Glo0.Whi0, 4 Glo1.[Whi1 | 0]
        // (G' fragment index = 2 ks + piece; weight fragments: 0 Whi0, 1 Wlo0, 2 [Whi1 | 0], 3 [Whi1 | Wlo1])
        constexpr int GI = B < 2 ? 0 : (B == 2 ? 2 : (B == 3 ? 1 : 3));
        constexpr int WI = B == 0 ? 0 : (B == 1 ? 1 : (B == 2 ? 3 : (B == 3 ? 0 : 2)));
#if SKINH_ABL & 256
        if(blockIdx.x == 0 && tid == 0 && dbg_item < 8) g_hslot_times[dbg_item * 256 + 126 + E * 5 + B] = __builtin_readcyclecounter();
#endif
        if constexpr(SKINH_ABL & 8)
        {
          if constexpr(B == 0) macc[ME] = zero16;
        }
        else if constexpr(!LIVE)
          ; // (a k-step none of the group's vertices has a weight in: its products are exact zeros)
        else if constexpr(B == FIRST)
          macc[ME] = mfma(gfr[MPE][GI], wfr[WI], zero16);
        else
          macc[ME] = mfma(gfr[MPE][GI], wfr[WI], macc[ME]);
        HSB();
        if constexpr(E == 0 && B == 0) h_barrier<h_barrier_vmcnt(HB_KS, HT, WANT_REST)>(); // slot 14: publishes slot 0 of the next item
        if constexpr(E == 0 && B >= 1 && B <= 3) dma(std::integral_constant<int, B - 1>{}, Bnext, HB_KS + 7 - HB_SLOTS, imgS[HB_KS % H_R]);
        if constexpr(E == 1) // operand fragments of the next item's first k-step (image (14 + 1) % 7)
        {
          bfr[0][B / 2][B % 2] = *reinterpret_cast<const v4f *>(imgV[(HB_KS + 1) % H_R] + B * 1024);
          if constexpr(B == 4) bfr[0][2][1] = *reinterpret_cast<const v4f *>(imgV[(HB_KS + 1) % H_R] + 5 * 1024);
        }
        if constexpr(E < 11 && B <= 3 && !(SKINH_ABL & (8 | 16))) // G' fragments of the next entry, in the order of their first use
        {
          constexpr int I = B == 0 ? 0 : (B == 1 ? 2 : (B == 2 ? 1 : 3));
          if constexpr(I < 2 ? K0 : K1) // (only the k-steps the group uses)
            gfr[MPE ^ 1][I] = *reinterpret_cast<const v4f *>((I < 2 ? gLane0 + I * 1024 : gLane1 + (I - 2) * 512) + (E + 1) * 3072);
        }
        if constexpr(E == 11 && B == 4) // root translation of the tail's first row
          trb[0] = *reinterpret_cast<const v4f *>(trLane);
        // The VALU rows of entry F = (XF, CF) ride behind MFMAs 1..4 of entry F + 1 (three rows each: its last MFMA, issued one gap
        // before MFMA 1, has retired by then) and behind MFMA 0 of entry F + 2 (the last four): at most four vector instructions
        // and a fragment read per gap — six FMAs (6/5/5 rows over three gaps) ran ~8 cycles over each gap.  Three accumulators
        // in rotation, so that entry F + 2's own MFMAs do not touch the one still being read.
        auto blend_rows = [&](auto ftag, auto r0tag, auto r1tag) {
          constexpr int F = decltype(ftag)::value, XF = F / 4, CF = F % 4, MF = F % 3, R0 = decltype(r0tag)::value, R1 = decltype(r1tag)::value;
#pragma unroll
          for(int r = R0; r < R1; r++)
          {
            if constexpr(CF == 0) tt[XF][r] = macc[MF][r] * acc[0][r];
            if constexpr(CF == 1) tt[XF][r] = __builtin_fmaf(macc[MF][r], acc[1][r], tt[XF][r]);
            if constexpr(CF == 2) tt[XF][r] = __builtin_fmaf(macc[MF][r], acc[2][r], tt[XF][r]);
            if constexpr(CF == 3) tt[XF][r] = __builtin_fmaf(tt[XF][r], cAB, macc[MF][r]);
          }
        };
        if constexpr(!(SKINH_ABL & 8))
        {
          if constexpr(E >= 1 && B >= 1)
            blend_rows(std::integral_constant<int, E - 1>{}, std::integral_constant<int, 3 * (B - 1)>{}, std::integral_constant<int, 3 * B>{});
          if constexpr(E >= 2 && B == 0)
            blend_rows(std::integral_constant<int, E - 2>{}, std::integral_constant<int, 12>{}, std::integral_constant<int, 16>{});
          if constexpr(E == 11 && B == 4) // (entry 10 has no entry 12 behind it; entry 11's rows are the tail's)
            blend_rows(std::integral_constant<int, 10>{}, std::integral_constant<int, 12>{}, std::integral_constant<int, 16>{});
        }
        if constexpr(WANT_REST && E >= 2 && E <= 9 && B < 2)
        {
          constexpr int R = 2 * (E - 2) + B, ROWC = (R & 3) + 8 * (R >> 2);
          v3f ov = {acc[0][R] * cAB, acc[1][R] * cAB, acc[2][R] * cAB};
          __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(v3u, ov), rsR, cur.voff, cur.sb + ROWC * frameB, SKINH_STORE_AUX);
          asm volatile("s_nop 1");
        }
        HSB();
      });
    });
    };
    // (wave-uniform: the flags word of the group, read at k-step 13)
    if(gcls == 1)
      skin_phase(std::integral_constant<int, 1>{});
    else if(gcls == 2)
      skin_phase(std::integral_constant<int, 2>{});
    else
      skin_phase(std::integral_constant<int, 3>{});

    prev = cur;
#if SKINH_ABL & (256 | 512)
    dbg_item++;
#endif
    {
      const unsigned char * v0 = imgV[0];
      const int s0 = imgS[0];
#pragma unroll
      for(int k = 0; k < H_R - 1; k++)
      {
        imgV[k] = imgV[k + 1];
        imgS[k] = imgS[k + 1];
      }
      imgV[H_R - 1] = v0;
      imgS[H_R - 1] = s0;
    }
  };

  // runs of items inside one frame tile: the A registers are loop-invariant in the inner loop
  for(int i = i0; i < i1;)
  {
    const int ft = i / nvx;
    const int iend = (ft + 1) * nvx < i1 ? (ft + 1) * nvx : i1;
    if(i != i0)
    {
      standalone_tail();
      load_frame_tile(ft, std::false_type{});
    }
    else
      load_frame_tile(ft, std::true_type{});
    // (later items find the fragments of their first k-step read by the blend phase of the item before)
    // item k of the run: vertex group vgk; the item after it: the next group, the XCD's first one when the frame tile ends
    // there, the same one when the workgroup's items end there
    int vgk = vg0 + (i - ft * nvx);
    auto next_vg = [&](int k, int vgc) { return k + 1 < i1 ? (vgc + 1 < vg1 ? vgc + 1 : vg0) : vgc; };
    do_item(ft, vgk, next_vg(i, vgk), std::false_type{});
    for(int k = i + 1; k < iend; k++)
    {
      vgk++;
      do_item(ft, vgk, next_vg(k, vgk), std::true_type{});
    }
    i = iend;
  }
  standalone_tail();
#if SKINH_ABL & 512
  if(blockIdx.x == 8 && tid == 0)
  {
    g_hslot_times[64] = __builtin_amdgcn_s_memtime();
    g_hslot_times[65] = __builtin_amdgcn_s_memrealtime();
    g_hslot_times[66] = t_start;
    g_hslot_times[67] = r_start;
    g_hslot_times[68] = i1 - i0;
  }
  if(tid == 0 && blockIdx.x < 256)
  {
    g_hwg_times[blockIdx.x * 4 + 0] = r_start;
    g_hwg_times[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();
    g_hwg_times[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memtime() - t_start;
    g_hwg_times[blockIdx.x * 4 + 3] = (unsigned long long)(i1 - i0) | ((unsigned long long)__builtin_amdgcn_s_getreg((6 << 11) | 20) << 32);
  }
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the last prefetches land before the wavefront ends
}

static hipError_t launch_h(const smplpp_model * m, int64_t n, const float * theta, float * verts, float * rest, hipStream_t st,
                           int64_t f_off)
{
  const int nft = (int)((n + 63) / 64);
  const int nvg = (int)m->VGPn;
  // per XCD: ceil(nvg / 8) * nft items at most; no more workgroups per XCD than that, and no more than the CUs it has
  const int per_xcd_items = ((nvg + 7) / 8) * nft;
  int nbx = device_cus(m->device) / 8;
  if(nbx > per_xcd_items) nbx = per_xcd_items;
  if(nbx < 1) nbx = 1;
  // ... and no more than the longest workgroup's item count needs: 56 items per XCD (256 frames) are two rounds on 32 workgroups and
  // on 28 — the four CUs per XCD left alone are where the IK loops' side stream (face scan, finish kernel) runs beside this kernel,
  // whose workgroups share a CU with nothing (1024 frames: 224 items, seven rounds on 32: unchanged)
  {
    const int rounds = (per_xcd_items + nbx - 1) / nbx;
    nbx = (per_xcd_items + rounds - 1) / rounds;
  }
  const bool wr = rest != nullptr;
  static PerDeviceOnce once[2];
  {
    hipError_t e = lds_opt_in(once[wr], m->device, wr ? reinterpret_cast<const void *>(&skin_kernel_h<true>) : reinterpret_cast<const void *>(&skin_kernel_h<false>), H_LDS_TOTAL);
    if(e != hipSuccess) return e;
  }
  const float cAB = 1.0f / (HB_SA * m->sB);
  const uint8_t * A2 = m->ws.A2h.as<uint8_t>() + (f_off / 64) * (int64_t)(HB_KS * HB_A_BYTES);
  const uint8_t * G2 = m->ws.G2h.as<uint8_t>() + (f_off / 64) * (int64_t)HB_G_BYTES;
  const float * th = theta + f_off * ((NJ + 1) * 3);
  float * vo = verts ? verts + f_off * m->V * 3 : nullptr;
  float * ro = rest ? rest + f_off * m->V * 3 : nullptr;
  if(wr)
    skin_kernel_h<true><<<dim3(nbx * 8), dim3(256), H_LDS_TOTAL, st>>>(A2, m->B2h, G2, th, vo, ro, n, m->V, nvg, nft, cAB);
  else
    skin_kernel_h<false><<<dim3(nbx * 8), dim3(256), H_LDS_TOTAL, st>>>(A2, m->B2h, G2, th, vo, ro, n, m->V, nvg, nft, cAB);
  return hipGetLastError();
}

#if SKINH_ABL & (256 | 512)
extern "C" int smplpp_debug_hwg_times(unsigned long long * out)
{
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(smplpp_hip::g_hwg_times), sizeof(unsigned long long) * 256 * 4);
}
extern "C" int smplpp_debug_hslot_times(unsigned long long * out)
{
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(smplpp_hip::g_hslot_times), sizeof(unsigned long long) * 8 * 256);
}
#endif
// A2h / G2h must hold whole 64-frame tiles (padding content is irrelevant: the rows it feeds are never stored)
hipError_t launch_skin_f16x2(const smplpp_model * m, int64_t n, const float * theta, float * verts, float * rest, hipStream_t st)
{
  // the kernel addresses its outputs with 32-bit buffer offsets: longer batches go in launches of <= 2 GiB of vertices
  // ... and of few enough frames that the G2h offsets (nft * 72 KiB) stay below 2^31 too (small meshes)
  int64_t per = (0x7fffff00LL / (m->V * 12)) & ~63LL;
  const int64_t per_g = (0x7fffff00LL / HB_G_BYTES) * 64;
  if(per > per_g) per = per_g;
  if(per < 64) return hipErrorInvalidValue;
  for(int64_t off = 0; off < n; off += per)
  {
    const int64_t nn = (n - off < per) ? n - off : per;
    hipError_t e = launch_h(m, nn, theta, verts, rest, st, off);
    if(e != hipSuccess) return e;
  }
  return hipSuccess;
}
} // namespace smplpp_hip
