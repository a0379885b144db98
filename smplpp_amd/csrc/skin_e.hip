// skin_kernel_e — the fused blend-shape GEMM + linear blend skinning kernel in the reference's arithmetic: fp32-exact operands on
// the bf16 matrix pipe, fp32 skinning on the vector ALU.  The form smplpp_fk runs by default (round 6).
// Reference path: rest = T + S.beta + P.c (/root/reference/src/BlendShape.cpp:670-683, 762-765,
// src/JointRegression.cpp:551-565) and the skinning of src/LinearBlendSkinning.cpp:445-553 (+ src/SMPL.cpp:726-727).
//
// Arithmetic (as skin_b.hip, round 1): every fp32 operand of the GEMM is carried as THREE bf16 pieces, x = x1 + x2 + x3 exactly
// (8 + 8 + 8 significant bits = fp32's 24), and a product a.b is the six v_mfma_f32_32x32x16_bf16 a1b1 + a1b2 + a2b1 + a1b3 +
// a2b2 + a3b1 accumulated in fp32: the three dropped cross terms are below 2^-24 |a||b|, one fp32 rounding.  The skinning
// (M = sum_j W[v,j] G'_j, h = M [rest; 1], out = h / sum_j W[v,j] + root) is plain fp32 FMAs on the accumulators, per (frame,
// vertex), exactly the reference's operations; it rides in the MFMA shadows of the NEXT item.
//
// Skeleton (what skin_h.hip taught, applied to the exact form; skin_b.hip staged A, B and G' per item through three images):
//  * work item 64 frames x 64 vertices, one 256-thread workgroup, one wavefront per SIMD (2 x 2 wavefronts of 32 x 32);
//  * a workgroup owns a FRAME TILE for a run of consecutive vertex groups: its A fragments (32 frames x 224 k x 3 pieces = 168
//    registers per lane, AGPRs) are loaded once per run, the relative transforms of the tile (72 KiB, fp32) and its root
//    translations stay resident in LDS for the run, and only the basis streams: one 20 KiB image per k-step (18 KiB of
//    fragments + the group's skinning tables, five 1 KiB LDS-DMA pieces per wavefront) through a ring of FOUR images filled
//    three k-steps ahead (buffer_load_dwordx4 ... lds); per item a wavefront issues 70 DMAs (skin_b: 102);
//  * XCD x owns an eighth of the vertex groups: its 3.8 MB slice of B3e is read from HBM once and served from that XCD's L2;
//  * one raw s_barrier per k-step; every barrier that publishes DMA data carries a counted s_waitcnt vmcnt(N), N derived at
//    compile time from a table of what each slot issues (hipcc does not order LDS reads behind LDS-DMA writes).
#include "common.h"

#include <type_traits>
#include <utility>

namespace smplpp_hip
{
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v3f __attribute__((ext_vector_type(3), aligned(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned v3u __attribute__((ext_vector_type(3)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int E_R = 4;                               // ring images: the image of k-step d is image d mod 4 (rotated by 14 mod 4 per item)
constexpr int E_G_BYTES = 64 * NJ * 48;              // 73728: G' of 64 frames, [frame][joint][3 x 4] fp32 (a straight copy of Gp)
constexpr int E_LDS_RING = E_G_BYTES;
constexpr int E_LDS_ROOT = E_LDS_RING + E_R * EB_IMG; // root translations of the 64 frames, (x, y, z, -) each
constexpr int E_LDS_TOTAL = E_LDS_ROOT + 64 * 16;    // 156672 <= 163840
constexpr int E_SLOTS = 18;                          // MFMAs per k-step: 3 coordinates x 6 piece products
constexpr int E_NSLOT = EB_KS * E_SLOTS;             // 252 slots per item
constexpr int E_BAR = 6;                             // slot of a k-step that carries its barrier
constexpr int E_NDMA = 5;                            // ring DMAs per wavefront and k-step (slots E_BAR + 1 .. E_BAR + 5)
constexpr int E_PITCH = 13;                          // slots between epilogue rows (a row takes 14: its last overlaps the next's first)
constexpr int E_ROW0 = 12;                           // first epilogue slot
constexpr int E_ROW_END = E_ROW0 + 15 * E_PITCH + 13; // 220: last epilogue slot
#ifndef SKINE_RD_AHEAD
#define SKINE_RD_AHEAD 6
#endif
constexpr int E_RD_AHEAD = SKINE_RD_AHEAD;           // slots between a matrix row's LDS read and the FMA group that uses it
constexpr int E_ROOT_P = 2;                          // row slot that reads the root translation (first used in slot 9)
constexpr int E_GCHUNKS = E_G_BYTES / (256 * 16);    // 18 DMAs of 1 KiB per wavefront
constexpr int E_GDMA0 = 12 * E_SLOTS + E_BAR + 1;    // 223: first slot of the G' DMAs of a run's first item (one per slot)
#ifndef SKINE_A_PRE
#define SKINE_A_PRE 4
#endif
constexpr int E_A_PRE = SKINE_A_PRE;                 // k-steps of A fragments the frame tile set-up loads; a run's first item loads the
constexpr int E_A_SLOT = 13;                         // fragments of k-step KS + E_A_PRE in slot 13 of k-step KS (three plain loads)
#ifndef SKINE_ABL
#define SKINE_ABL 0 // timing ablations (development only; results are wrong when non-zero): 1 no epilogue, 2 no ring DMA, 4 no MFMA, 8 no barrier, 16 no fragment reads, 32 no stores, 512 per-workgroup timestamps
#endif
#ifndef SKINE_STORE_AUX
#define SKINE_STORE_AUX 0 // cache policy of the output stores (bit 0 sc0, bit 1 nt, bit 4 sc1)
#endif
#ifndef SKINE_DMA_AUX
#define SKINE_DMA_AUX 0 // cache policy of the ring DMAs
#endif
#ifndef SKINE_LGKM
#define SKINE_LGKM 1 // 1: the k-step barriers let the epilogue's youngest LDS reads stay in flight (counted lgkmcnt); 0: lgkmcnt(0)
#endif

template<class F, int... I>
__device__ __forceinline__ void estatic_for_impl(F && f, std::integer_sequence<int, I...>)
{
  (f(std::integral_constant<int, I>{}), ...);
}
template<int N, class F>
__device__ __forceinline__ void estatic_for(F && f)
{
  estatic_for_impl(f, std::make_integer_sequence<int, N>{});
}

// piece products in issue order (index into the A pieces, index into the B pieces): small terms first
constexpr int E_PA[6] = {2, 0, 1, 1, 0, 0};
constexpr int E_PB[6] = {0, 2, 1, 0, 1, 0};

// ---- compile-time bookkeeping of what each slot issues.  Order inside a slot: MFMA, [barrier], fragment / table reads, ring DMA,
// G' DMA, epilogue (its LDS reads, then its stores).
constexpr bool e_dma_slot(int S)
{
  return S % E_SLOTS > E_BAR && S % E_SLOTS <= E_BAR + E_NDMA;
}
// vector-memory instructions slot S issues BEHIND its ring DMA (hp: the item carries an epilogue; rest: it also stores `rest`)
constexpr int e_vmem_after_dma(int S, bool hp, bool rest)
{
  int c = 0;
  if(!hp && S >= E_GDMA0 && S < E_GDMA0 + E_GCHUNKS) c += 1; // G' DMA (a run's first item)
  if(!hp && S % E_SLOTS == E_A_SLOT && S / E_SLOTS + E_A_PRE < EB_KS) c += 3; // A fragments of k-step KS + 4 (a run's first item)
  if(hp && S >= E_ROW0)
    for(int rr = 0; rr < 16; rr++)
    {
      if(S - E_ROW0 - rr * E_PITCH == 13) c += 1;        // vertex store of row rr
      if(rest && S - E_ROW0 - rr * E_PITCH == 0) c += 1; // rest store of row rr
    }
  return c;
}
constexpr int e_vmem_ops(int S, bool hp, bool rest)
{
  return (e_dma_slot(S) ? 1 : 0) + e_vmem_after_dma(S, hp, rest);
}
// vmcnt of the barrier of k-step ks (slot 18 ks + 6, ahead of that slot's own DMA): the ring DMAs of k-step ks + 1 were issued in
// k-step ks - 3 (slots 7..11); everything issued behind the last of them may stay in flight.  A window that reaches into the
// previous item counts only what every kind of item issues there (its ring DMAs): a smaller count only waits for older
// operations.  k-step 0 also publishes the G' image a run's first item DMA'd in its slots 223..240: only the ring DMAs of its
// k-step 13 are younger.
constexpr int e_barrier_vmcnt(int ks, bool hp, bool rest)
{
  int c = 0;
  if(ks >= 3)
  {
    const int L = (ks - 3) * E_SLOTS + E_BAR + E_NDMA; // slot of the last DMA waited for
    c = e_vmem_after_dma(L, hp, rest);
    for(int S = L + 1; S < ks * E_SLOTS + E_BAR; S++) c += e_vmem_ops(S, hp, rest);
    return c < 63 ? c : 63;
  }
  const int first = ks == 0 ? E_GDMA0 + E_GCHUNKS : (ks + EB_KS - 3) * E_SLOTS + E_BAR + E_NDMA + 1;
  for(int S = first; S < E_NSLOT; S++) c += e_dma_slot(S) ? 1 : 0;
  for(int S = 0; S < ks * E_SLOTS + E_BAR; S++) c += e_vmem_ops(S, hp, rest);
  return c < 63 ? c : 63;
}
// LDS instructions the epilogue of the previous item issues in slot S (one matrix row of one joint, E_RD_AHEAD slots ahead of the
// group that multiplies it; the root translation of a row)
constexpr int e_epilogue_lds_ops(int S, int maxw)
{
  (void)maxw;
  int c = 0;
  for(int rr = 0; rr < 16; rr++)
  {
    const int d = S - (E_ROW0 + rr * E_PITCH); // row slot of row rr
    if(d + E_RD_AHEAD - 1 >= 0 && d + E_RD_AHEAD - 1 < 12) c += 1; // the read of group d + E_RD_AHEAD - 1
    if(d == E_ROOT_P) c += 1;                                      // root translation
  }
  return c;
}
// last slot of k-step ks - 1 that reads the image the barrier of k-step ks frees (the image of k-step ks): the nine fragment reads
// sit in slots 6..14; the image of k-step 1 also gives winv (slot 15 of k-step 0), the image of the next item's k-step 0 its
// skinning tables (slots 15, 16 of k-step 13)
constexpr int e_last_image_read(int ks)
{
  return ks == 1 ? 15 : (ks == 0 ? 16 : 14);
}
// lgkmcnt of the barrier of k-step ks: the epilogue reads issued behind the last read of the image being freed may stay in flight
// (LDS instructions of a wavefront complete in order; the kernel issues no scalar loads inside the loop)
constexpr int e_barrier_lgkm(int ks, bool hp, int maxw)
{
  if(!hp || !SKINE_LGKM) return 0;
  int c = 0;
  // from the slot of that last read on (inside a slot the epilogue's reads come behind it); slots before 0 are the previous item's
  // 18 * 13 + m: no epilogue reads there, its rows end at 220
  for(int q = (ks - 1) * E_SLOTS + e_last_image_read(ks); q < ks * E_SLOTS + E_BAR; q++)
    if(q >= 0) c += e_epilogue_lds_ops(q, maxw);
  return c < 15 ? c : 15;
}
static_assert(E_LDS_TOTAL <= 160 * 1024, "LDS plan");
static_assert(E_ROW_END < 12 * E_SLOTS + E_BAR, "the rows end before the slots a run's first item uses for its G' DMAs");
static_assert(E_ROW0 + 1 - E_RD_AHEAD > E_BAR, "the first G' read of an item follows the barrier that publishes the tile");
static_assert(E_GDMA0 + E_GCHUNKS <= E_NSLOT, "the G' DMAs fit the item");
static_assert(e_barrier_vmcnt(0, true, false) == 5 && (E_A_PRE != 4 || (e_barrier_vmcnt(1, false, false) == 10 + 3 && e_barrier_vmcnt(3, false, false) == 10 + 9 && e_barrier_vmcnt(5, false, false) == 10 + 9 && e_barrier_vmcnt(13, false, false) == 10 + 17)) &&
                  e_barrier_vmcnt(3, true, false) == 10 + 3 && e_barrier_vmcnt(5, true, false) == 10 + 4,
              "window bookkeeping");

template<int LGKM, int VM>
__device__ __forceinline__ void e_barrier()
{
#if SKINE_ABL & 8
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(%1)" ::"n"(VM), "n"(LGKM) : "memory");
#else
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(%1)\n\ts_barrier" ::"n"(VM), "n"(LGKM) : "memory");
#endif
}
__device__ __forceinline__ void e_full_barrier()
{
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
#define ESB() __builtin_amdgcn_sched_barrier(0)
#if SKINE_ABL & 512
__device__ unsigned long long g_ewg_times[256 * 8];
#endif
#if SKINE_ABL & 256
// slot timestamps (development only): wavefront 0 of workgroup 0 stamps the cycle counter at every slot of its first 8 items
__device__ unsigned long long g_eslot_times[8 * 256];
#endif

template<int MAXW, bool WANT_REST>
__global__ __launch_bounds__(256, 1) void skin_kernel_e(const uint8_t * __restrict__ A3, const uint8_t * __restrict__ B3e,
                                                        const float * __restrict__ Gp, const float * __restrict__ theta,
                                                        float * __restrict__ verts, float * __restrict__ rest, int64_t n, int64_t V,
                                                        int nvg, int nft)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  typedef __attribute__((address_space(3))) void * lds_ptr_t;
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wf = wave & 1, wv = wave >> 1;
  // ---- work assignment (as skin_kernel_h).  Workgroup b runs on XCD b & 7 (round-robin dispatch; a wrong guess costs speed
  // only).  XCD x owns vertex groups [vg0, vg1); its items, frame tile major, are cut into contiguous runs, one per workgroup.
  const int nbx = (int)(gridDim.x >> 3), xcd = (int)(blockIdx.x & 7), jb = (int)(blockIdx.x >> 3);
  const int vg0 = (xcd * nvg) >> 3, vg1 = ((xcd + 1) * nvg) >> 3, nvx = vg1 - vg0;
  const int cnt = nvx * nft;
  const int i0 = (int)(((unsigned)jb * (unsigned)cnt) / (unsigned)nbx), i1 = (int)(((unsigned)(jb + 1) * (unsigned)cnt) / (unsigned)nbx);
  if(i0 >= i1) return; // whole workgroup leaves
#if SKINE_ABL & 512
  const unsigned long long t_start = __builtin_amdgcn_s_memtime(), r_start = __builtin_amdgcn_s_memrealtime();
  unsigned long long t_first = 0, t_last = 0, t_drain0 = 0, t_run2 = 0;
#endif

  const int frameB = (int)(V * 12);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(B3e), 0, (int)(nvg * EB_KS * EB_IMG), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Gp), 0, (int)(nft * E_G_BYTES), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(verts, 0, (int)(verts ? n * V * 12 : 0), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(rest, 0, (int)(rest ? n * V * 12 : 0), 0x00020000);
  const int voffDma = wave * (E_NDMA * 1024) + lane * 16; // this wavefront's five pieces of an image: + i * 1024
  const int voffG = tid * 16;                             // chunk i of the G' tile: + i * 4096

  // LDS addresses.  imgS[k]: byte offset of ring image k; imgV[k]: this lane's B fragments in it (fragment (x, s): + (3 x + s) *
  // 1024).  Rotated by two at every item (14 k-steps mod 4 images).
  const unsigned char * imgV[E_R];
  int imgS[E_R];
#pragma unroll
  for(int k = 0; k < E_R; k++)
  {
    imgS[k] = E_LDS_RING + k * EB_IMG;
    imgV[k] = lds + imgS[k] + (wv * 9 * 64 + lane) * 16;
  }
  const int tabLane = EB_TAB_OFF + (wv * 32 + l31) * 16;                          // this lane's row of a 16-byte-per-vertex table in an image
  const unsigned char * const gLane = lds + (wf * 32 + 4 * half) * (NJ * 48);      // G' of accumulator row R: + rowc(R) * 1152
  const v4f * const rootLane = reinterpret_cast<const v4f *>(lds + E_LDS_ROOT) + (wf * 32 + 4 * half); // + rowc(R)

#if SKINE_ABL & 256
  int dbg_item = 0;
#endif
  f32x16 acc[3], accp[3];
  v4f areg[EB_KS][3]; // [k-step][piece]: this wavefront's 32 frames, loaded once per run
  v4f bfr[2][3][3];   // B fragments by k-step parity, [coordinate][piece]; a k-step's nine are read during the one before
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  struct Item
  {
    int voff; // byte offset of (frame 4 * half, vertex v) in an output array; out of range when the lane has no vertex
    int sb;   // byte offset of the wavefront's first frame (wave-uniform)
    float winv;
    int jofs[MAXW]; // byte offset of joint i's matrix inside a frame's G'
    float jw[MAXW];
  } cur, prev;
  int nx_jofs[MAXW]; // skinning tables of the NEXT item (read beside the fragments of its first k-step)
  float nx_jw[MAXW];
  cur.voff = prev.voff = 0x7fffff00;
  cur.sb = prev.sb = 0;
  cur.winv = prev.winv = 0.0f;
#pragma unroll
  for(int i = 0; i < MAXW; i++)
  {
    cur.jofs[i] = prev.jofs[i] = nx_jofs[i] = 0;
    cur.jw[i] = prev.jw[i] = nx_jw[i] = 0.0f;
  }

  // piece I (0..4) of this wavefront's share of k-step `ks` of the vertex group at byte base vgBase, HBM/L2 -> ring image at dst.
  // The instruction's immediate offset (12 bits) moves the global address AND the LDS address: pieces 0..3 share one M0.
  auto dma = [&](auto itag, int vgBase, int ks, int dst) {
    constexpr int I = decltype(itag)::value;
    if constexpr(SKINE_ABL & 2) return;
    if constexpr(I < 4)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(lds + dst + wave * (E_NDMA * 1024)), 16, voffDma, vgBase + ks * EB_IMG, I * 1024, SKINE_DMA_AUX);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(lds + dst + wave * (E_NDMA * 1024) + 4096), 16, voffDma, vgBase + ks * EB_IMG + 4096, 0, SKINE_DMA_AUX);
  };
  auto read_tables = [&](const unsigned char * img0) { // skinning tables of a group, from the image of its k-step 0
    const v4i jo = *reinterpret_cast<const v4i *>(img0 + tabLane);
    const v4f jv = *reinterpret_cast<const v4f *>(img0 + tabLane + 1024);
    nx_jofs[0] = jo.x; nx_jofs[1] = jo.y; nx_jofs[2] = jo.z; nx_jofs[3] = jo.w;
    nx_jw[0] = jv.x; nx_jw[1] = jv.y; nx_jw[2] = jv.z; nx_jw[3] = jv.w;
  };

  // ---- frame tile set-up: root translations into LDS, A fragments into registers (plain loads: the compiler waits for each at
  // its first use).  The G' image of the tile is DMA'd by the run's first item in its slots 223..240 (its first reader is the
  // epilogue that rides in the SECOND item, or the drain).  FIRST (the workgroup's first item): also the ring's prologue, k-steps
  // 0..3 into images 0..3; only k-step 0 is waited for.
  auto load_frame_tile = [&](int ft, auto first_tag) {
    constexpr bool FIRST = decltype(first_tag)::value;
    e_full_barrier(); // (later runs: every wavefront is done with the previous tile's G' image and root translations)
    float tval = 0.0f;
    if(tid < 192)
    {
      const int64_t f = (int64_t)ft * 64 + tid / 3;
      if(f < n) tval = theta[f * ((NJ + 1) * 3) + tid % 3]; // theta[f, 0, :] (src/SMPL.cpp:726-727)
    }
    const int vgF = (vg0 + i0 % nvx) * (EB_KS * EB_IMG);
    if constexpr(FIRST)
      estatic_for<E_NDMA>([&](auto ii) { dma(ii, vgF, 0, imgS[0]); });
    const uint8_t * ap = A3 + ((int64_t)ft * EB_KS * 2 + wf) * 3072 + lane * 16;
    // (the first four k-steps only: every CU starts at the same time and what a workgroup pulls before its first MFMA is served at
    // ~20 B/clk; the run's first item loads the rest, three fragments per k-step, four k-steps ahead of their first use)
#pragma unroll
    for(int ks = 0; ks < E_A_PRE; ks++)
#pragma unroll
      for(int s = 0; s < 3; s++) areg[ks][s] = *reinterpret_cast<const v4f *>(ap + ks * BB_A_BYTES + s * 1024);
    if constexpr(FIRST)
    {
      estatic_for<E_R - 1>([&](auto dd) {
        constexpr int D = decltype(dd)::value + 1;
        estatic_for<E_NDMA>([&](auto ii) { dma(ii, vgF, D, imgS[D]); });
      });
      e_barrier<0, E_A_PRE * 3 + (E_R - 1) * E_NDMA>(); // behind k-step 0: the A loads and k-steps 1..3 may stay in flight
#pragma unroll
      for(int q = 0; q < 9; q++) bfr[0][q / 3][q % 3] = *reinterpret_cast<const v4f *>(imgV[0] + q * 1024);
      read_tables(lds + imgS[0]);
    }
    if(tid < 192) *reinterpret_cast<float *>(lds + E_LDS_ROOT + (tid / 3) * 16 + (tid % 3) * 4) = tval;
  };

  // ---- one work item.  HP (compile time): the epilogue (skinning + stores) of the previous item rides in this item's MFMA
  // shadows; !HP: the first item of a run, which DMAs the run's G' image instead.
  auto do_item = [&](int ft_in, int vg_in, int vgn_in, auto hp_tag) {
    constexpr bool HP = decltype(hp_tag)::value;
    constexpr bool EPI = HP && !(SKINE_ABL & 1);
    const int ft = __builtin_amdgcn_readfirstlane(ft_in), vg = __builtin_amdgcn_readfirstlane(vg_in), vgn = __builtin_amdgcn_readfirstlane(vgn_in);
    const int Bcur = vg * (EB_KS * EB_IMG), Bnext = vgn * (EB_KS * EB_IMG), Gbase = ft * E_G_BYTES;
    {
      const int v = vg * 64 + wv * 32 + l31;
      cur.voff = v < (int)V ? v * 12 + (4 * half) * frameB : 0x7fffff00; // (V * 12 * 64 frames < 2^31: launch_skin_exact cuts the batch)
      cur.sb = __builtin_amdgcn_readfirstlane((ft * 64 + wf * 32) * frameB);
#pragma unroll
      for(int i = 0; i < MAXW; i++)
      {
        cur.jofs[i] = nx_jofs[i];
        cur.jw[i] = nx_jw[i];
      }
    }

    // epilogue state.  Row R of the previous item takes row slots P = 0..13 at a pitch of 13 (its last slot is the next row's
    // first).  P = 1..12: one FMA group each — group g = P - 1 is matrix row g / 4 of joint g % 4, four FMAs against the 16 bytes
    // one ds_read_b128 fetched E_RD_AHEAD slots earlier (ONE LDS read and at most six vector instructions per slot: a slot with
    // three reads or ten dependent instructions ran three MFMA times); h = M [rest; 1] follows each matrix row as it completes, in
    // the reference's order ((x + y) + z) + w, one instruction per slot; the store sits alone in slot 13.
    static_assert(MAXW == 4, "row schedule of four weights per vertex");
    float rt0 = 0.f, rt1 = 0.f, rt2 = 0.f, tx = 0.f, ty = 0.f, ox = 0.f, oy = 0.f;
    v4f m0 = {0.f, 0.f, 0.f, 0.f}, m1 = m0, m2 = m0;
    constexpr int NSET = E_RD_AHEAD + 2; // register sets of matrix rows in flight (E_RD_AHEAD + 1 are: read that many slots ahead of their use),
                                         // taken in the order of the reads: set (12 R + G) mod NSET
    v4f gq[NSET];

    // the LDS read of group G of row R2
    auto read_group = [&](auto r2tag, auto gtag) {
      constexpr int R2 = decltype(r2tag)::value, G = decltype(gtag)::value;
      constexpr int ROWC2 = (R2 & 3) + 8 * (R2 >> 2);
      gq[(R2 * 12 + G) % NSET] = *reinterpret_cast<const v4f *>(gLane + ROWC2 * (NJ * 48) + (G / 4) * 16 + prev.jofs[G % 4]);
    };
    // slot P (0..13) of row R of the previous item
    auto row_piece = [&](auto rtag, auto ptag) {
      constexpr int R = decltype(rtag)::value, P = decltype(ptag)::value;
      constexpr int ROWC = (R & 3) + 8 * (R >> 2); // + 4 * half: accumulator row -> frame in the wavefront's 32
      const float rx = accp[0][R], ry = accp[1][R], rz = accp[2][R];
      if constexpr(P == 0 && WANT_REST)
      {
        v3f ov = {rx, ry, rz};
        if constexpr(!(SKINE_ABL & 32)) __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(v3u, ov), rsR, prev.voff, prev.sb + ROWC * frameB, 0);
      }
      if constexpr(P >= 1 && P <= 12)
      {
        // scalar FMAs on purpose: packed f32 VALU beside MFMAs is an anti-lever (MI355X_MICROARCH.md, cycle constants)
        constexpr int G = P - 1, J = G % 4, MR = G / 4;
        const float w = prev.jw[J];
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
      if constexpr(P == E_ROOT_P)
      {
        const v4f rt = rootLane[ROWC];
        rt0 = rt.x;
        rt1 = rt.y;
        rt2 = rt.z;
      }
      // h = M [rest; 1], cart = h / sum_j W + root (src/LinearBlendSkinning.cpp:465-475, 545-550)
      // (x . M0 + y . M1 with the second product rounded and the first fused: what the compiler makes of skin_b.hip's expression —
      // tools/fk_e_check.py compares the two kernels bit for bit)
      if constexpr(P == 5) tx = m0.y * ry;
      if constexpr(P == 6) tx = __builtin_fmaf(m0.x, rx, tx);
      if constexpr(P == 7) tx = __builtin_fmaf(m0.z, rz, tx);
      if constexpr(P == 8) tx = tx + m0.w;
      if constexpr(P == 9)
      {
        ox = __builtin_fmaf(tx, prev.winv, rt0);
        ty = m1.y * ry;
      }
      if constexpr(P == 10) ty = __builtin_fmaf(m1.x, rx, ty);
      if constexpr(P == 11) ty = __builtin_fmaf(m1.z, rz, ty);
      if constexpr(P == 12)
      {
        ty = ty + m1.w;
        oy = __builtin_fmaf(ty, prev.winv, rt1);
      }
      if constexpr(P == 13)
      {
        float tz = m2.y * ry;
        tz = __builtin_fmaf(m2.x, rx, tz);
        tz = __builtin_fmaf(m2.z, rz, tz);
        tz = tz + m2.w;
        // write-once output; the descriptor's range check drops frames >= n and vertex-less lanes.
        // (An MFMA always follows before the next VALU write: see the store hazard note at the drain.)
        v3f ov = {ox, oy, __builtin_fmaf(tz, prev.winv, rt2)};
        if constexpr(SKINE_ABL & 32)
          asm volatile("" ::"v"(ov.x), "v"(ov.y), "v"(ov.z));
        else
          __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(v3u, ov), rsV, prev.voff, prev.sb + ROWC * frameB, SKINE_STORE_AUX);
      }
    };

    estatic_for<E_NSLOT>([&](auto ss) {
      constexpr int S = decltype(ss)::value;
      constexpr int KS = S / E_SLOTS, M = S % E_SLOTS;
      constexpr int X = M / 6, Q = M % 6;
      constexpr int AP = KS & 1, IMG = KS % E_R, IMGN = (KS + 1) % E_R;
#if SKINE_ABL & 256
      if(blockIdx.x == 0 && tid == 0 && dbg_item < 8) g_eslot_times[dbg_item * 256 + S] = __builtin_readcyclecounter();
#endif
      if constexpr(SKINE_ABL & 4)
      {
        if constexpr(KS == 0 && Q == 0) acc[X] = zero16;
        asm volatile("" ::"v"(areg[KS][E_PA[Q]]), "v"(bfr[AP][X][E_PB[Q]]));
      }
      else if constexpr(KS == 0 && Q == 0)
        acc[X] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, areg[KS][E_PA[Q]]), __builtin_bit_cast(bf16x8, bfr[AP][X][E_PB[Q]]), zero16, 0, 0, 0);
      else
        acc[X] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, areg[KS][E_PA[Q]]), __builtin_bit_cast(bf16x8, bfr[AP][X][E_PB[Q]]), acc[X], 0, 0, 0);
      ESB();

      // Barrier of the k-step.  Behind it image (KS + 1) % 4 holds k-step KS + 1 (its DMAs have landed: vmcnt) and image KS % 4 is
      // free for the DMAs of k-step KS + 4 (every wavefront's reads of it have completed: lgkmcnt).  k-step 0 also publishes the G'
      // image and the root translations of a new run.
      if constexpr(M == E_BAR) e_barrier<e_barrier_lgkm(KS, EPI, MAXW), e_barrier_vmcnt(KS, EPI, WANT_REST)>();
      // ---- operand fragments of the NEXT k-step, all nine behind this k-step's barrier (slots 6..14)
      if constexpr(M >= E_BAR && M < E_BAR + 9 && !(SKINE_ABL & 16))
      {
        constexpr int NP = (KS + 1) & 1, XX = (M - E_BAR) / 3, SP = (M - E_BAR) % 3;
        bfr[NP][XX][SP] = *reinterpret_cast<const v4f *>(imgV[IMGN] + (3 * XX + SP) * 1024);
      }
      // ---- the group's skinning tables: winv from the image of k-step 1; the NEXT item's joint offsets / weights from the image of
      // its k-step 0 (image 14 % 4, behind the barrier of k-step 13)
      if constexpr(KS == 0 && M == 15) cur.winv = *reinterpret_cast<const float *>(lds + imgS[1] + EB_TAB_OFF + (wv * 32 + l31) * 4);
      if constexpr(KS == EB_KS - 1 && M == 15)
      {
        const v4i jo = *reinterpret_cast<const v4i *>(lds + imgS[EB_KS % E_R] + tabLane);
        nx_jofs[0] = jo.x; nx_jofs[1] = jo.y; nx_jofs[2] = jo.z; nx_jofs[3] = jo.w;
      }
      if constexpr(KS == EB_KS - 1 && M == 16)
      {
        const v4f jv = *reinterpret_cast<const v4f *>(lds + imgS[EB_KS % E_R] + tabLane + 1024);
        nx_jw[0] = jv.x; nx_jw[1] = jv.y; nx_jw[2] = jv.z; nx_jw[3] = jv.w;
      }
      ESB(); // (the LDS instructions above are the ones the k-step barriers count from)

      // ---- ring DMA: k-step KS + 4 into the image this k-step has just finished with (slots 7..11, one piece each)
      if constexpr(M > E_BAR && M <= E_BAR + E_NDMA)
      {
        constexpr int KN = KS + E_R;
        dma(std::integral_constant<int, M - E_BAR - 1>{}, KN < EB_KS ? Bcur : Bnext, KN < EB_KS ? KN : KN - EB_KS, imgS[IMG]);
      }
      // ---- A fragments of k-step KS + 4 (a run's first item; the tile set-up loaded k-steps 0..3)
      if constexpr(!HP && M == E_A_SLOT && KS + E_A_PRE < EB_KS)
      {
        const uint8_t * ap = A3 + ((int64_t)ft * EB_KS * 2 + wf) * 3072 + lane * 16 + (KS + E_A_PRE) * BB_A_BYTES;
#pragma unroll
        for(int s = 0; s < 3; s++) areg[KS + E_A_PRE][s] = *reinterpret_cast<const v4f *>(ap + s * 1024);
      }
      // ---- G' tile of a new run: HBM -> LDS by DMA, slots 223..240; published by the first barrier of the next item (or the drain)
      if constexpr(!HP && S >= E_GDMA0 && S < E_GDMA0 + E_GCHUNKS)
      {
        constexpr int GI = S - E_GDMA0;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsG, (lds_ptr_t)(lds + wave * 1024 + GI * 4096), 16, voffG + GI * 4096, Gbase, 0, 0);
      }

      if constexpr(EPI)
      {
        // ---- the matrix row a group E_RD_AHEAD slots ahead will multiply (one LDS read per slot)
        constexpr int S2 = S + E_RD_AHEAD - E_ROW0 - 1;
        if constexpr(S2 >= 0 && S2 / E_PITCH < 16 && S2 % E_PITCH < 12)
          read_group(std::integral_constant<int, S2 / E_PITCH>{}, std::integral_constant<int, S2 % E_PITCH>{});
        // ---- this slot's piece(s) of the rows in progress (finishing slot of row R - 1 first, then the opening slot of row R)
        constexpr int S1 = S - E_ROW0;
        if constexpr(S1 >= 0 && S1 <= E_ROW_END - E_ROW0)
        {
          constexpr int R = S1 / E_PITCH < 16 ? S1 / E_PITCH : 15;
          if constexpr(R >= 1 && S1 - (R - 1) * E_PITCH == 13) row_piece(std::integral_constant<int, R - 1>{}, std::integral_constant<int, 13>{});
          if constexpr(S1 - R * E_PITCH <= 13) row_piece(std::integral_constant<int, R>{}, std::integral_constant<int, S1 - R * E_PITCH>{});
        }
      }
      ESB();
    });

    // the current item becomes the previous one; the images rotate (14 k-steps per item, 14 mod 4 = 2)
#if SKINE_ABL & 256
    if(blockIdx.x == 0 && tid == 0 && dbg_item < 8) g_eslot_times[dbg_item * 256 + 252] = __builtin_readcyclecounter();
    dbg_item++;
#endif
#pragma unroll
    for(int x = 0; x < 3; x++) accp[x] = acc[x];
    prev = cur;
    {
      const unsigned char * v0 = imgV[0], * v1 = imgV[1];
      const int s0 = imgS[0], s1 = imgS[1];
      imgV[0] = imgV[2]; imgS[0] = imgS[2];
      imgV[1] = imgV[3]; imgS[1] = imgS[3];
      imgV[2] = v0; imgS[2] = s0;
      imgV[3] = v1; imgS[3] = s1;
    }
  };

  // ---- the epilogue of a run's last item with nothing to hide behind (the G' tile may still be on its way: a run of one item)
  auto drain = [&]() {
    e_full_barrier();
    // the twelve matrix rows of accumulator row R + 1 are requested before row R is computed (two register sets)
    v4f gd[2][12], rtd[2];
    auto request = [&](auto rtag) {
      constexpr int R = decltype(rtag)::value;
      constexpr int ROWC = (R & 3) + 8 * (R >> 2);
      rtd[R & 1] = rootLane[ROWC];
#pragma unroll
      for(int g = 0; g < 12; g++) gd[R & 1][g] = *reinterpret_cast<const v4f *>(gLane + ROWC * (NJ * 48) + (g / 4) * 16 + prev.jofs[g % 4]);
    };
    request(std::integral_constant<int, 0>{});
    estatic_for<16>([&](auto rr) {
      constexpr int R = decltype(rr)::value;
      constexpr int ROWC = (R & 3) + 8 * (R >> 2);
      const v4f rt = rtd[R & 1];
      if constexpr(R + 1 < 16) request(std::integral_constant<int, R + 1>{});
      const float rx = accp[0][R], ry = accp[1][R], rz = accp[2][R];
      if constexpr(WANT_REST)
      {
        v3f ov = {rx, ry, rz};
        __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(v3u, ov), rsR, prev.voff, prev.sb + ROWC * frameB, 0);
        ESB();
        asm volatile("s_nop 1");
        ESB();
      }
      // (no MFMA is in flight here: packed fp32 math, two FMAs per instruction — the same products and sums as the slot stream's)
      typedef float v2f __attribute__((ext_vector_type(2)));
      v2f mlo[3], mhi[3];
#pragma unroll
      for(int g = 0; g < 12; g++)
      {
        const float w = prev.jw[g % 4];
        const v4f gm = gd[R & 1][g];
        const v2f w2 = {w, w}, glo = {gm.x, gm.y}, ghi = {gm.z, gm.w};
        if(g % 4 == 0)
        {
          mlo[g / 4] = w2 * glo;
          mhi[g / 4] = w2 * ghi;
        }
        else
        {
          mlo[g / 4] = __builtin_elementwise_fma(w2, glo, mlo[g / 4]);
          mhi[g / 4] = __builtin_elementwise_fma(w2, ghi, mhi[g / 4]);
        }
      }
      float m[3][4];
#pragma unroll
      for(int c = 0; c < 3; c++)
      {
        m[c][0] = mlo[c].x;
        m[c][1] = mlo[c].y;
        m[c][2] = mhi[c].x;
        m[c][3] = mhi[c].y;
      }
      // (the slot stream's operations in the slot stream's order: tools/fk_e_check.py compares the two paths bit for bit)
      float t[3];
#pragma unroll
      for(int c = 0; c < 3; c++)
      {
        t[c] = m[c][1] * ry;
        t[c] = __builtin_fmaf(m[c][0], rx, t[c]);
        t[c] = __builtin_fmaf(m[c][2], rz, t[c]);
        t[c] = t[c] + m[c][3];
      }
      v3f ov = {__builtin_fmaf(t[0], prev.winv, rt.x), __builtin_fmaf(t[1], prev.winv, rt.y), __builtin_fmaf(t[2], prev.winv, rt.z)};
      __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(v3u, ov), rsV, prev.voff, prev.sb + ROWC * frameB, SKINE_STORE_AUX);
      // HAZARD (measured on gfx950, see skin_b.hip): keep one instruction between a 96-bit buffer store and the next VALU write
      // to its data registers
      ESB();
      asm volatile("s_nop 1");
      ESB();
    });
  };

  // runs of items inside one frame tile: the A registers are loop-invariant in the inner loop
  for(int i = i0; i < i1;)
  {
    const int ft = i / nvx;
    const int iend = (ft + 1) * nvx < i1 ? (ft + 1) * nvx : i1;
    if(i != i0)
    {
#if SKINE_ABL & 512
      t_drain0 = __builtin_amdgcn_s_memtime();
#endif
      drain();
      load_frame_tile(ft, std::false_type{});
    }
    else
      load_frame_tile(ft, std::true_type{});
    // item k of the run: vertex group vgk; the item after it: the next group, the XCD's first one when the frame tile ends there,
    // the same one when the workgroup's items end there (its prefetches land in images nobody reads)
#if SKINE_ABL & 512
    if(i == i0) t_first = __builtin_amdgcn_s_memtime(); else t_run2 = __builtin_amdgcn_s_memtime();
#endif
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
#if SKINE_ABL & 512
  t_last = __builtin_amdgcn_s_memtime();
#endif
  drain();
#if SKINE_ABL & 512
  if(tid == 0 && blockIdx.x < 256)
  {
    g_ewg_times[blockIdx.x * 8 + 0] = r_start;
    g_ewg_times[blockIdx.x * 8 + 1] = __builtin_amdgcn_s_memrealtime();
    g_ewg_times[blockIdx.x * 8 + 2] = __builtin_amdgcn_s_memtime() - t_start;
    g_ewg_times[blockIdx.x * 8 + 3] = (unsigned long long)(i1 - i0);
    g_ewg_times[blockIdx.x * 8 + 4] = t_first - t_start;                          // prologue
    g_ewg_times[blockIdx.x * 8 + 5] = __builtin_amdgcn_s_memtime() - t_last;      // final drain
    g_ewg_times[blockIdx.x * 8 + 6] = t_run2 ? t_run2 - t_drain0 : 0;             // drain + set-up between two runs
    g_ewg_times[blockIdx.x * 8 + 7] = t_last - t_first;                           // first item .. last item (incl. a run change)
  }
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the last prefetches land before the wavefront ends
}

template<int MAXW, bool WANT_REST>
static hipError_t launch_e(const smplpp_model * m, int64_t n, const float * theta, float * verts, float * rest, hipStream_t st, int64_t f_off)
{
  const int nft = (int)((n + 63) / 64);
  const int nvg = (int)m->VGPn;
  // per XCD: ceil(nvg / 8) * nft items at most; no more workgroups per XCD than that, and no more than the CUs it has; and no
  // more than the longest workgroup's item count needs (as launch_h)
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
    hipError_t e = lds_opt_in(once, m->device, reinterpret_cast<const void *>(&skin_kernel_e<MAXW, WANT_REST>), E_LDS_TOTAL);
    if(e != hipSuccess) return e;
  }
  // f_off (a multiple of 64): first frame of this launch inside the workspace / caller arrays of a longer batch
  skin_kernel_e<MAXW, WANT_REST><<<dim3(nbx * 8), dim3(256), E_LDS_TOTAL, st>>>(
      m->ws.A3.as<uint8_t>() + (f_off / 64) * (int64_t)(BB_KS * BB_A_BYTES), m->B3e, m->ws.Gp.as<float>() + f_off * (NJ * 12),
      theta + f_off * ((NJ + 1) * 3), verts ? verts + f_off * m->V * 3 : nullptr, rest ? rest + f_off * m->V * 3 : nullptr, n, m->V, nvg, nft);
  return hipGetLastError();
}

#if SKINE_ABL & 256
extern "C" int smplpp_debug_eslot_times(unsigned long long * out)
{
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(smplpp_hip::g_eslot_times), sizeof(unsigned long long) * 8 * 256);
}
#endif
#if SKINE_ABL & 512
extern "C" int smplpp_debug_ewg_times(unsigned long long * out)
{
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(smplpp_hip::g_ewg_times), sizeof(unsigned long long) * 256 * 8);
}
#endif
// A3 / Gp must hold whole 64-frame tiles (padding content is irrelevant: the rows it feeds are never stored)
hipError_t launch_skin_exact(const smplpp_model * m, int64_t n, const float * theta, float * verts, float * rest, hipStream_t st)
{
  // the kernel addresses its outputs with 32-bit buffer offsets: longer batches go in launches of <= 2 GiB of vertices
  // ... and of few enough frames that the Gp offsets (nft * 72 KiB) stay below 2^31 too (small meshes)
  int64_t per = (0x7fffff00LL / (m->V * 12)) & ~63LL;
  const int64_t per_g = (0x7fffff00LL / E_G_BYTES) * 64;
  if(per > per_g) per = per_g;
  if(per < 64) return hipErrorInvalidValue;
  for(int64_t off = 0; off < n; off += per)
  {
    const int64_t nn = (n - off < per) ? n - off : per;
    hipError_t e = rest ? launch_e<4, true>(m, nn, theta, verts, rest, st, off) : launch_e<4, false>(m, nn, theta, verts, rest, st, off);
    if(e != hipSuccess) return e;
  }
  return hipSuccess;
}
} // namespace smplpp_hip
