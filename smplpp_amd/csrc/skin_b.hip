// skin_kernel_b — the fused blend-shape GEMM + linear blend skinning kernel on the bf16 matrix pipe, fp32-exact operands.
//
// Why: v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 MFMA rate, and the fp32 forms of this kernel (fk.hip, skin_p.hip) are
// bound by it.  Here every fp32 operand is carried as THREE bf16 pieces, x = x1 + x2 + x3 exactly (8 + 8 + 8 significant
// bits), and a product a.b is evaluated as the six bf16 MFMAs a1b1 + a1b2 + a2b1 + a1b3 + a2b2 + a3b1 accumulated in
// fp32 — the three dropped cross terms are below 2^-24 |a||b|, the size of one fp32 rounding.  6 MFMAs of 32 cycles do the
// work of 8 fp32 MFMAs of 64 cycles: 2.6x fewer matrix-pipe cycles at fp32-level accuracy (the parity tests compare it
// with the fp32-MFMA forms against an fp64-accumulating CPU restatement: tests/test_fk_gpu.py).
//
// Work item: 64 frames x 64 vertices (x 3 coordinates), one 256-thread workgroup (one wavefront per SIMD, 2 x 2
// wavefronts of 32 frames x 32 vertices each), 14 k-steps of 16.  Operands are stored in HBM in MFMA FRAGMENT ORDER — a
// "piece" is the 1 KiB a wavefront's 64 lanes feed to one MFMA (lane l = 32 h + r holds k = 8 h .. 8 h + 7 of row/column
// r) — so staging is a straight 24 KiB copy per k-step (6 pieces of A, 18 of B) into one of THREE LDS images, done by
// LDS-DMA (buffer_load_dwordx4 ... lds: 1 KiB per wavefront instruction, no VGPRs, and none of the 13-cycle VGPR->LDS
// transfer a ds_write_b128 costs — with register staging the LDS store path alone was ~40 % of the LDS's time); each
// wavefront reads its fragments back with lane-linear ds_read_b128 (conflict-free).  The four wavefronts share the
// staged bytes: HBM/L2 traffic per MFMA is half that of per-wavefront operand streams.  The G' tile of an item (72 KiB)
// arrives the same way.  hipcc does not order LDS reads behind LDS-DMA writes: every barrier that publishes DMA data
// carries an explicit counted s_waitcnt vmcnt(N), N = the vector-memory instructions issued after the last DMA it needs.
//
// Software pipeline (as skin_p.hip): the instruction stream of an item is 252 hand-placed "slots", one per MFMA; the
// skinning epilogue of the PREVIOUS item (16 accumulator rows at a pitch of 13 slots), the DMA of the operands three
// k-steps ahead, the fragment reads of the next coordinates and the G' tile of the current item all issue in the MFMA
// shadows.  One raw s_barrier per k-step (slot 6) orders the LDS images; DMAs stay in flight across it.
#include "common.h"

#include <cstdlib>
#include <type_traits>
#include <utility>

namespace smplpp_hip
{
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v3f __attribute__((ext_vector_type(3), aligned(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v3u __attribute__((ext_vector_type(3)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int B_NIMG = 3;                              // operand images in LDS (data of k-step d lives in image d mod 3)
constexpr int B_LDS_OP = B_NIMG * BB_KSTEP_BYTES;      // 73728
constexpr int B_LDS_G = 64 * NJ * 12 * 4;              // 73728: G' of 64 frames
constexpr int B_LDS_ROOT = 64 * 16;                    // root translation of 64 frames, one (x, y, z, -) per frame
constexpr int B_LDS_TOTAL = B_LDS_OP + B_LDS_G + B_LDS_ROOT;
constexpr int B_SLOTS = 18;                            // MFMAs per k-step: 3 coordinates x 6 piece products
constexpr int B_NSLOT = BB_KS * B_SLOTS;               // 252 slots per item
#ifndef SKINB_ABL
#define SKINB_ABL 0 // timing ablations (development only; results are wrong when non-zero): 1 no epilogue, 2 no staging, 4 no MFMA, 8 no barrier, 16 no fragment reads
#endif
constexpr int B_BAR = 6;                               // slot of a k-step that carries its barrier
constexpr int B_PITCH = 13;                            // slots between epilogue rows (a row takes 14: its last overlaps the next's first)
constexpr int B_ROW0 = 12;                             // first epilogue slot; the rows end at slot 220, before the barrier of
                                                       // k-step 12 (slot 222), after which the G' image is overwritten
constexpr int B_ROW_END = B_ROW0 + 15 * B_PITCH + 13;  // 220: last epilogue slot
constexpr int B_RD_AHEAD = 4;                          // extra slots between a joint's LDS reads and their use (<= 5: the first
                                                       // read of an item must stay behind the barrier of slot 6)
constexpr int B_ROOT_P = 9;                            // row slot that reads the root translation (used in slot 13)
constexpr int B_ROOT_KS = 11;                          // k-step whose slot 14 loads the root translations into a register
constexpr int B_GCHUNKS = B_LDS_G / (256 * 16);        // 18 DMAs of 1 KiB per wavefront
constexpr int B_GDMA0 = 12 * B_SLOTS + B_BAR + 1;      // 223: first slot of the G' DMAs (one per slot, 223..240)

template<class F, int... I>
__device__ __forceinline__ void bstatic_for_impl(F && f, std::integer_sequence<int, I...>)
{
  (f(std::integral_constant<int, I>{}), ...);
}
template<int N, class F>
__device__ __forceinline__ void bstatic_for(F && f)
{
  bstatic_for_impl(f, std::make_integer_sequence<int, N>{});
}

// piece products in issue order (index into the A pieces, index into the B pieces): small terms first
constexpr int B_PA[6] = {2, 0, 1, 1, 0, 0};
constexpr int B_PB[6] = {0, 2, 1, 0, 1, 0};

// ---- compile-time bookkeeping of what each slot issues (the counted waits of the barriers are derived from it)
// LDS instructions the epilogue of the previous item issues in slot S BEHIND the slot's sched_barrier line
// (joint matrices: 3 ds_read_b128 per joint; root translation: 1)
constexpr int epilogue_lds_ops(int S, int maxw)
{
  int c = 0;
  const int s2 = S + B_RD_AHEAD - B_ROW0;
  if(s2 >= 0 && s2 <= 15 * B_PITCH + 12)
  {
    const int r = s2 / B_PITCH < 16 ? s2 / B_PITCH : 15;
    for(int rr = r; rr >= 0 && rr >= r - 1; rr--) // a slot can belong to row rr (slots 0..12) and to row rr - 1 (slot 13)
    {
      const int p = s2 - rr * B_PITCH;
      if(p < 0 || p > 13) continue;
      for(int j = 0; j < maxw; j++)
        if(p == (3 * j) / (maxw / 4)) c += 3;
    }
  }
  const int s1 = S - B_ROW0;
  if(s1 >= 0)
    for(int rr = 0; rr < 16; rr++)
      if(s1 - rr * B_PITCH == B_ROOT_P) c += 1;
  return c;
}
// LDS instructions issued between the last fragment read of a k-step (slot 14, ahead of that slot's sched_barrier line) and
// the barrier in slot S = 18 ks + 6 of the next one: the epilogue reads of slots 14..17 and 0..5
constexpr int lds_ops_since_frag_reads(int S, int maxw)
{
  int c = 0;
  for(int q = S - 10; q < S; q++) c += epilogue_lds_ops(q, maxw);
  return c;
}
// vector-memory instructions slot S of an item issues (hp: the item carries an epilogue; rest: it also stores `rest`)
constexpr int vmem_ops(int S, bool hp, bool rest)
{
  const int m = S % B_SLOTS, ks = S / B_SLOTS;
  int c = 0;
  if(m > B_BAR && m <= B_BAR + 6) c += 1;                        // operand DMA
  if(S >= B_GDMA0 && S < B_GDMA0 + B_GCHUNKS) c += 1;            // G' DMA
  if(ks == B_ROOT_KS && m == 14) c += 1;                         // root translation load
  if(hp && S >= B_ROW0)
    for(int rr = 0; rr < 16; rr++)
    {
      if(S - B_ROW0 - rr * B_PITCH == 13) c += 1;                // vertex store of row rr
      if(rest && S - B_ROW0 - rr * B_PITCH == 0) c += 1;         // rest store of row rr
    }
  return c;
}
// vmcnt for the barrier of k-step ks (slot 6): the operand DMAs of k-step ks + 1 were issued in slots 7..12 of k-step
// ks - 2, the G' DMAs of the previous item in its slots 223..240; everything issued after the last of those may stay in
// flight.  Windows reaching into the previous item use the smaller (stricter) count of the two item kinds.
constexpr int barrier_vmcnt(int ks, bool hp, bool rest)
{
  int c = 0;
  if(ks >= 2)
  {
    for(int S = (ks - 2) * B_SLOTS + B_BAR + 7; S < ks * B_SLOTS + B_BAR; S++) c += vmem_ops(S, hp, rest);
    return c;
  }
  // k-steps 0 and 1: the window starts in the previous item (no epilogue stores counted there: stricter), and k-step 0
  // also needs the previous item's G' DMAs, the last of which was issued in slot B_GDMA0 + 17
  const int first = ks == 0 ? B_GDMA0 + B_GCHUNKS : (BB_KS - 1) * B_SLOTS + B_BAR + 7;
  for(int S = first; S < B_NSLOT; S++) c += vmem_ops(S, false, false);
  for(int S = 0; S < ks * B_SLOTS + B_BAR; S++) c += vmem_ops(S, false, false);
  return c;
}

// the tables above must describe exactly what the slot stream issues per item
constexpr int sum_vmem_ops(bool hp, bool rest)
{
  int c = 0;
  for(int S = 0; S < B_NSLOT; S++) c += vmem_ops(S, hp, rest);
  return c;
}
constexpr int sum_epilogue_lds_ops(int maxw)
{
  int c = 0;
  for(int S = 0; S < B_NSLOT; S++) c += epilogue_lds_ops(S, maxw);
  return c;
}
static_assert(sum_vmem_ops(false, false) == BB_KS * 6 + B_GCHUNKS + 1, "operand DMAs + G' DMAs + root load");
static_assert(sum_vmem_ops(true, false) == BB_KS * 6 + B_GCHUNKS + 1 + 16, "+ one vertex store per row");
static_assert(sum_vmem_ops(true, true) == BB_KS * 6 + B_GCHUNKS + 1 + 32, "+ one rest store per row");
static_assert(sum_epilogue_lds_ops(4) == 16 * (4 * 3 + 1) && sum_epilogue_lds_ops(8) == 16 * (8 * 3 + 1), "joint matrices + root per row");
static_assert(B_ROW_END < 12 * B_SLOTS + B_BAR, "the rows must end before the barrier after which the G' image is overwritten");
static_assert(B_ROW0 - B_RD_AHEAD > B_BAR, "the first G' read of an item must follow the barrier that publishes the tile");
static_assert(B_GDMA0 + B_GCHUNKS <= B_NSLOT, "the G' DMAs must fit the item");

// Barrier of a k-step: LGKM = LDS instructions of this wavefront that may stay in flight (epilogue reads issued behind
// slot 5's sched_barrier line), VM = vector-memory instructions that may stay in flight (see barrier_vmcnt).
template<int LGKM, int VM>
__device__ __forceinline__ void kstep_barrier()
{
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(%1)\n\ts_barrier" ::"n"(VM), "n"(LGKM) : "memory");
}

__device__ __forceinline__ void full_barrier()
{
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

#if SKINB_ABL & 256
// slot timestamps (development only): wavefront 0 of workgroup 0 stamps s_memtime at every slot of its first 8 items
__device__ unsigned long long g_slot_times[8 * 256];
#endif

template<int MAXW, bool WANT_REST>
__global__ __launch_bounds__(256, 1) void skin_kernel_b(const uint8_t * __restrict__ A3, const uint8_t * __restrict__ B3,
                                                        const float * __restrict__ Gp, const float * __restrict__ theta,
                                                        const uint8_t * __restrict__ wIdx, const float * __restrict__ wVal,
                                                        const float * __restrict__ wSum, float * __restrict__ verts,
                                                        float * __restrict__ rest, int64_t n, int64_t V, int nvgp, int nftp,
                                                        int items_per_block)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wf = wave & 1, wv = wave >> 1;
  // Work assignment, XCD- and L2-aware.  Workgroup b runs on XCD b & 7 (round-robin dispatch).  XCD x owns a contiguous
  // range of vertex-group pairs, i.e. a private 1/8 of B3 (each B3 byte is fetched from HBM by one XCD only), and its
  // workgroups take the range's items (vertex-group pair major, frame-tile pair minor) INTERLEAVED: workgroup j does
  // items j, j + nbx, j + 2 nbx, ...  So at any moment the ~32 workgroups of an XCD stream the same one or two 258 KB
  // slices of B3 (L2 hits for all but the first) against different frame tiles; a contiguous run per workgroup instead
  // puts 32 different slices (8 MB) through a 4 MB L2 and every staging load misses it (measured: 453 MB of L2 misses per
  // launch for 27 MB of B3).  A wrong placement guess costs speed, never correctness: the item lists tile the work either way.
  const int nbx = (int)(gridDim.x >> 3), xcd = (int)(blockIdx.x & 7), jb = (int)(blockIdx.x >> 3);
  const int vg0 = (xcd * nvgp) >> 3, vg1 = ((xcd + 1) * nvgp) >> 3;
  const int cnt = (vg1 - vg0) * nftp; // items of this XCD
  (void)items_per_block;
  if(jb >= cnt) return; // whole workgroup leaves: no barrier is ever skipped by a subset of its wavefronts

  // ---- descriptors (SGPR) and per-thread constant offsets (VGPR)
  const __amdgpu_buffer_rsrc_t rsA =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(A3), 0, (int)(nftp * BB_KS * BB_A_BYTES), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(B3), 0, (int)(nvgp * BB_KS * BB_B_BYTES), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsG =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Gp), 0, (int)(nftp * B_LDS_G), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(verts, 0, (int)(verts ? n * V * 12 : 0), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(rest, 0, (int)(rest ? n * V * 12 : 0), 0x00020000);
  // staging: chunk q = i * 256 + tid (16 B each) of the 24 KiB k-step image; chunks [0, 384) are A, the rest B.
  // i = 0: A for everyone; i = 1: A for wavefronts 0-1, B for 2-3 (wave-uniform choice); i >= 2: B.
  const bool mixA = wave < 2;
  const __amdgpu_buffer_rsrc_t rsM = mixA ? rsA : rsB;
  const int voff0 = tid * 16;
  const int voffM = mixA ? (256 + tid) * 16 : (tid - 128) * 16;
  const int voffB = (tid + 128) * 16; // i >= 2: + (i - 2) * 4096
  const int frameB = (int)(V * 12);
  const __amdgpu_buffer_rsrc_t rsT =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(theta), 0, (int)(n * (NJ + 1) * 12), 0x00020000);
  const int voffT = ((tid < 192 ? tid : 191) / 3) * ((NJ + 1) * 12) + ((tid < 192 ? tid : 191) % 3) * 4;

  typedef __attribute__((address_space(3))) void * lds_ptr_t;
  // LDS map: [3 operand images of 24 KiB][G' tile 72 KiB][root translations 1 KiB]
  const unsigned char * aImg[B_NIMG]; // this lane's A fragment of image k (piece s: + s * 1024); rotated at item boundaries
  const unsigned char * bImg[B_NIMG]; // this lane's B fragment of image k (piece (x, s): + (3 x + s) * 1024)
  int dmaImg[B_NIMG];                 // this wavefront's DMA destination in image k (chunk i: + i * 4096), LDS byte offset
#pragma unroll
  for(int k = 0; k < B_NIMG; k++)
  {
    aImg[k] = lds + k * BB_KSTEP_BYTES + (wf * 3 * 64 + lane) * 16;
    bImg[k] = lds + k * BB_KSTEP_BYTES + BB_A_BYTES + (wv * 9 * 64 + lane) * 16;
    dmaImg[k] = k * BB_KSTEP_BYTES + wave * 1024;
  }
  const int dmaG = B_LDS_OP + wave * 1024;                                                // G' tile chunk i: + i * 4096
  const unsigned char * const gLane = lds + B_LDS_OP + (wf * 32 + 4 * half) * (NJ * 48); // G' of frame row R: + rowc(R) * 1152
  float * const sRoot = reinterpret_cast<float *>(lds + B_LDS_OP + B_LDS_G);
  const v4f * const rootLane = reinterpret_cast<const v4f *>(sRoot) + (wf * 32 + 4 * half); // + rowc(R)
  float * const rootWr = sRoot + ((tid < 192 ? tid : 191) / 3) * 4 + (tid < 192 ? tid % 3 : 3); // threads >= 192 hit the pad word

  f32x16 acc[3], accp[3];
  v4f afr[2][3], bfr[2][3][3]; // operand fragments by k-step parity (B: [coordinate][piece]); a k-step's twelve are read during the one before
  float rstage = 0.0f;

  struct Item
  {
    int voff;    // byte offset of (frame 4 * half, vertex v) in an output array; out of range when the lane has no vertex
    float winv;
    int jofs[MAXW]; // byte offset of joint i's matrix inside a frame's G'
    float jw[MAXW];
  } cur, prev;

  auto item_bases = [&](int i, int & Ab, int & Bb, int & Gb) { // i: index into this XCD's item list
    const int iu = __builtin_amdgcn_readfirstlane(i);
    const int vgp = vg0 + iu / nftp, ftp = iu % nftp;
    Ab = ftp * (BB_KS * BB_A_BYTES);
    Bb = vgp * (BB_KS * BB_B_BYTES);
    Gb = ftp * B_LDS_G;
  };
  // chunk i (0..5) of one k-step image, HBM/L2 -> LDS by DMA (Ak / Bk: byte bases of that k-step in A3 / B3; dst: LDS byte
  // offset of this wavefront's share of the image)
  auto dma_chunk = [&](int i, int Ak, int Bk, int dst) {
    lds_ptr_t d = (lds_ptr_t)(lds + dst + i * 4096);
    if(i == 0)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, d, 16, voff0, Ak, 0, 0);
    else if(i == 1)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsM, d, 16, voffM, mixA ? Ak : Bk, 0, 0);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, d, 16, voffB + (i - 2) * 4096, Bk, 0, 0);
  };

  int f0_prev = 0;
#if SKINB_ABL & 256
  int dbg_item = 0;
#endif
  // ---- prologue: k-steps 0, 1, 2 of the first item into images 0, 1, 2
  {
    int Abase, Bbase, Gbase;
    item_bases(jb, Abase, Bbase, Gbase);
    (void)Gbase;
#pragma unroll
    for(int d = 0; d < 3; d++)
#pragma unroll
      for(int i = 0; i < 6; i++) dma_chunk(i, Abase + d * BB_A_BYTES, Bbase + d * BB_B_BYTES, dmaImg[d]);
    full_barrier();
#pragma unroll
    for(int sp = 0; sp < 3; sp++)
    {
      afr[0][sp] = *reinterpret_cast<const v4f *>(aImg[0] + sp * 1024);
#pragma unroll
      for(int x = 0; x < 3; x++) bfr[0][x][sp] = *reinterpret_cast<const v4f *>(bImg[0] + (3 * x + sp) * 1024);
    }
  }

  // one work item; HP (compile time) = there is a previous item whose epilogue rides in this item's MFMA shadows
  auto do_item = [&](int t, auto hp_tag) {
    constexpr bool HP = decltype(hp_tag)::value;
    constexpr bool EPI = HP && !(SKINB_ABL & 1);
    const int tu = __builtin_amdgcn_readfirstlane(t);
    const int vgp = vg0 + tu / nftp, ftp = tu % nftp;
    const int64_t v = (int64_t)vgp * 64 + wv * 32 + l31;
    const bool has_v = v < V;
    const int f0_cur = ftp * 64 + wf * 32; // first frame of this wavefront's 32 (wave-uniform: stays in an SGPR)
    const int sb_prev = __builtin_amdgcn_readfirstlane(f0_prev * frameB);
    int Abase, Bbase, Gbase; // byte bases of this item's operands (recomputed, not loop-carried: keeps them scalar)
    item_bases(t, Abase, Bbase, Gbase);
    cur.voff = has_v ? (int)(v * 12 + (int64_t)(4 * half) * frameB) : 0x7fffff00;
    {
      const int64_t vv = has_v ? v : 0;
#pragma unroll
      for(int i = 0; i < MAXW; i++)
      {
        cur.jofs[i] = (int)wIdx[vv * MAXW + i] * 48;
        cur.jw[i] = wVal[vv * MAXW + i];
      }
      cur.winv = 1.0f / wSum[vv];
    }
    const int tn = (t + nbx < cnt) ? t + nbx : t; // next item (or this one again: harmless extra prefetch)
    int Abn, Bbn, Gbn;
    item_bases(tn, Abn, Bbn, Gbn);
    (void)Gbn;
#pragma unroll
    for(int x = 0; x < 3; x++)
#pragma unroll
      for(int r = 0; r < 16; r++) acc[x][r] = 0.0f;

    // epilogue state (row R of the previous item lives in rxyz[R & 1]: the last slot of a row is the first of the next)
    float rxyz[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    float rt0 = 0.f, rt1 = 0.f, rt2 = 0.f, hx = 0.f, hy = 0.f;
    v4f m0 = {0.f, 0.f, 0.f, 0.f}, m1 = m0, m2 = m0;
    constexpr int NSET = MAXW; // one register set per joint of a row: a set is re-read 13 slots later, after its last use
    constexpr int GPS = MAXW / 4; // FMA groups (4 FMAs: one joint, one matrix row) per slot
    v4f gq[NSET][3];

    // LDS reads of the joints of row R2 that belong to row slot P2 (issued B_RD_AHEAD slots early)
    auto read_piece = [&](auto r2tag, auto p2tag) {
      constexpr int R2 = decltype(r2tag)::value, P2 = decltype(p2tag)::value;
      constexpr int ROWC2 = (R2 & 3) + 8 * (R2 >> 2);
#pragma unroll
      for(int j = 0; j < MAXW; j++)
        if(P2 == (3 * j) / GPS)
        {
          const unsigned char * gj = gLane + ROWC2 * (NJ * 48) + prev.jofs[j];
          gq[j % NSET][0] = *reinterpret_cast<const v4f *>(gj);
          gq[j % NSET][1] = *reinterpret_cast<const v4f *>(gj + 16);
          gq[j % NSET][2] = *reinterpret_cast<const v4f *>(gj + 32);
        }
    };
    // slot P (0..13) of row R of the previous item
    auto row_piece = [&](auto rtag, auto ptag) {
      constexpr int R = decltype(rtag)::value, P = decltype(ptag)::value;
      constexpr int ROWC = (R & 3) + 8 * (R >> 2); // + 4 * half: accumulator row -> frame in the wavefront's 32
      float & rx = rxyz[R & 1][0];
      float & ry = rxyz[R & 1][1];
      float & rz = rxyz[R & 1][2];
      if constexpr(P == 0)
      {
        rx = accp[0][R];
        ry = accp[1][R];
        rz = accp[2][R];
        if constexpr(WANT_REST)
        {
          v3f ov = {rx, ry, rz};
          __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(v3u, ov), rsR, prev.voff, sb_prev + ROWC * frameB, 2);
        }
      }
      if constexpr(P >= 1 && P <= 12)
      {
        // scalar FMAs on purpose: packed f32 VALU beside MFMAs is an anti-lever (MI355X_MICROARCH.md, cycle constants)
#pragma unroll
        for(int g = (P - 1) * GPS; g < P * GPS; g++)
        {
          const int j = g / 3, row = g % 3;
          const float w = prev.jw[j];
          const v4f gm = gq[j % NSET][row];
          v4f & mm = (row == 0 ? m0 : (row == 1 ? m1 : m2));
          if(j == 0)
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
      }
      if constexpr(P == B_ROOT_P)
      {
        const v4f rt = rootLane[ROWC];
        rt0 = rt.x;
        rt1 = rt.y;
        rt2 = rt.z;
      }
      if constexpr(P == 12 && MAXW == 4)
      {
        hx = ((m0.x * rx + m0.y * ry) + m0.z * rz) + m0.w;
        hy = ((m1.x * rx + m1.y * ry) + m1.z * rz) + m1.w;
      }
      if constexpr(P == 13)
      {
        if constexpr(MAXW != 4)
        {
          hx = ((m0.x * rx + m0.y * ry) + m0.z * rz) + m0.w;
          hy = ((m1.x * rx + m1.y * ry) + m1.z * rz) + m1.w;
        }
        const float hz = ((m2.x * rx + m2.y * ry) + m2.z * rz) + m2.w;
        // write-once output: non-temporal (aux = 2); the descriptor's range check drops frames >= n and vertex-less lanes.
        // (An MFMA always follows before the next VALU write: see the store hazard note at the drain.)
        v3f ov = {hx * prev.winv + rt0, hy * prev.winv + rt1, hz * prev.winv + rt2};
        __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(v3u, ov), rsV, prev.voff, sb_prev + ROWC * frameB, 2);
      }
    };

    bstatic_for<B_NSLOT>([&](auto ss) {
      constexpr int S = decltype(ss)::value;
      constexpr int KS = S / B_SLOTS, M = S % B_SLOTS;
      constexpr int X = M / 6, Q = M % 6;
      constexpr int AP = KS & 1, IMG = KS % B_NIMG, IMGN = (KS + 1) % B_NIMG;
#if SKINB_ABL & 256
      if(blockIdx.x == 0 && tid == 0 && dbg_item < 8) g_slot_times[dbg_item * 256 + S] = __builtin_readcyclecounter();
#endif
      if constexpr(!(SKINB_ABL & 4))
        acc[X] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, afr[AP][B_PA[Q]]),
                                                         __builtin_bit_cast(bf16x8, bfr[AP][X][B_PB[Q]]), acc[X], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);

      if constexpr(M == B_BAR && !(SKINB_ABL & 8))
      {
        // Barrier of the k-step.  After it: image (KS + 1) % 3 holds k-step KS + 1 (its DMAs have landed: vmcnt) and
        // image KS % 3 is free for the DMAs of k-step KS + 3 (every wavefront's reads of it have completed: lgkmcnt).
        // k-step 0 also publishes the root translations; k-step 12 retires every read of the G' tile before its DMAs overwrite it.
        // LDS instructions that may stay in flight: those issued after this wavefront's last read of image KS % 3 (slot 14 of
        // the previous k-step); for k-step 0 those after the root translation write of slot 5.
        constexpr int LG = !EPI || KS == 12 ? 0 : (KS == 0 ? epilogue_lds_ops(S - 1, MAXW) : lds_ops_since_frag_reads(S, MAXW));
        constexpr int VM = barrier_vmcnt(KS, EPI, WANT_REST);
        kstep_barrier<(LG < 15 ? LG : 15), (VM < 63 ? VM : 63)>();
      }
      // ---- operand fragments of the NEXT k-step, all twelve behind this k-step's barrier (slots 6..14): their image has
      // landed, they are >= 6 slots old at the next barrier (whose LDS wait is then free) and >= 6 slots ahead of their MFMAs
      if constexpr(M >= B_BAR && M < B_BAR + 9 && !(SKINB_ABL & 16))
      {
        constexpr int NP = (KS + 1) & 1, XX = (M - B_BAR) / 3, SP = (M - B_BAR) % 3;
        if constexpr(XX == 0) afr[NP][SP] = *reinterpret_cast<const v4f *>(aImg[IMGN] + SP * 1024);
        bfr[NP][XX][SP] = *reinterpret_cast<const v4f *>(bImg[IMGN] + (3 * XX + SP) * 1024);
      }
      // ---- root translations of the PREVIOUS item: register -> LDS, published by the barrier of k-step 0
      if constexpr(EPI && KS == 0 && M == B_BAR - 1) *rootWr = rstage;
      __builtin_amdgcn_sched_barrier(0); // (the LDS instructions above are the ones the k-step barrier has to wait for)

      // ---- operand DMA: k-step KS + 3 into the image this k-step has just finished with (slots 7..12, one chunk each)
      if constexpr(M > B_BAR && M <= B_BAR + 6 && !(SKINB_ABL & 2))
      {
        constexpr int KN = KS + 3;
        const int Ak = (KN < BB_KS ? Abase + KN * BB_A_BYTES : Abn + (KN - BB_KS) * BB_A_BYTES);
        const int Bk = (KN < BB_KS ? Bbase + KN * BB_B_BYTES : Bbn + (KN - BB_KS) * BB_B_BYTES);
        dma_chunk(M - B_BAR - 1, Ak, Bk, dmaImg[IMG]);
      }
      // ---- G' tile of the CURRENT item: HBM -> LDS by DMA, slots 223..240 (the rows of the previous item ended in slot 220
      // and the barrier of slot 222 said so for all four wavefronts); published by the first barrier of the next item
      if constexpr(S >= B_GDMA0 && S < B_GDMA0 + B_GCHUNKS && !(SKINB_ABL & 1))
      {
        constexpr int GI = S - B_GDMA0;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsG, (lds_ptr_t)(lds + dmaG + GI * 4096), 16, voff0 + GI * 4096, Gbase, 0, 0);
      }
      if constexpr(KS == B_ROOT_KS && M == 14)
      {
        // root translation theta[f, 0, :] (src/SMPL.cpp:726-727) of the block's 64 frames: 192 values, one per thread.
        // Buffer load: 32-bit offsets only (no 64-bit VALU address math in the MFMA stream); frames >= n read as 0.
        rstage = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsT, voffT, ftp * (64 * (NJ + 1) * 12), 0));
      }

      if constexpr(EPI)
      {
        // ---- joint matrices of the rows to come (LDS reads B_RD_AHEAD slots early)
        constexpr int S2 = S + B_RD_AHEAD - B_ROW0;
        if constexpr(S2 >= 0 && S2 <= 15 * B_PITCH + 12)
        {
          constexpr int R2 = S2 / B_PITCH < 16 ? S2 / B_PITCH : 15;
          if constexpr(S2 - R2 * B_PITCH <= 13) read_piece(std::integral_constant<int, R2>{}, std::integral_constant<int, S2 - R2 * B_PITCH>{});
          if constexpr(R2 >= 1 && S2 - (R2 - 1) * B_PITCH <= 13)
            read_piece(std::integral_constant<int, R2 - 1>{}, std::integral_constant<int, S2 - (R2 - 1) * B_PITCH>{});
        }
        // ---- this slot's piece(s) of the rows in progress (finishing slot of row R - 1 first, then the opening slot of row R)
        constexpr int S1 = S - B_ROW0;
        if constexpr(S1 >= 0 && S1 <= B_ROW_END - B_ROW0)
        {
          constexpr int R = S1 / B_PITCH < 16 ? S1 / B_PITCH : 15;
          if constexpr(R >= 1 && S1 - (R - 1) * B_PITCH == 13) row_piece(std::integral_constant<int, R - 1>{}, std::integral_constant<int, 13>{});
          if constexpr(S1 - R * B_PITCH <= 13) row_piece(std::integral_constant<int, R>{}, std::integral_constant<int, S1 - R * B_PITCH>{});
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });

    // the current item becomes the previous one; the images rotate (14 k-steps per item, 14 mod 3 = 2)
#pragma unroll
    for(int x = 0; x < 3; x++) accp[x] = acc[x];
    prev = cur;
    f0_prev = f0_cur;
    {
      const unsigned char * a0 = aImg[0], * b0 = bImg[0];
      const int d0 = dmaImg[0];
      aImg[0] = aImg[2]; bImg[0] = bImg[2]; dmaImg[0] = dmaImg[2];
      aImg[2] = aImg[1]; bImg[2] = bImg[1]; dmaImg[2] = dmaImg[1];
      aImg[1] = a0; bImg[1] = b0; dmaImg[1] = d0;
    }
  };

  do_item(jb, std::false_type{});
  for(int t = jb + nbx; t < cnt; t += nbx)
  {
#if SKINB_ABL & 256
    dbg_item++;
#endif
    do_item(t, std::true_type{});
  }

  // ---- drain: the epilogue of the last item with nothing to hide behind (its G' tile was DMA'd in its own slots 223..240)
  *rootWr = rstage;
  full_barrier();
  bstatic_for<16>([&](auto rr) {
    constexpr int R = decltype(rr)::value;
    constexpr int ROWC = (R & 3) + 8 * (R >> 2);
    const float rx = accp[0][R], ry = accp[1][R], rz = accp[2][R];
    const int soff = __builtin_amdgcn_readfirstlane((f0_prev + ROWC) * frameB);
    if constexpr(WANT_REST)
    {
      v3f ov = {rx, ry, rz};
      __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(v3u, ov), rsR, prev.voff, soff, 2);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_nop 1");
      __builtin_amdgcn_sched_barrier(0);
    }
    v4f m0 = {0.f, 0.f, 0.f, 0.f}, m1 = m0, m2 = m0;
#pragma unroll
    for(int i = 0; i < MAXW; i++)
    {
      const unsigned char * gj = gLane + ROWC * (NJ * 48) + prev.jofs[i];
      const v4f g0 = *reinterpret_cast<const v4f *>(gj), g1 = *reinterpret_cast<const v4f *>(gj + 16),
                g2 = *reinterpret_cast<const v4f *>(gj + 32);
      const float w = prev.jw[i];
      m0 = w * g0 + m0; // (drain only: packed math is fine here)
      m1 = w * g1 + m1;
      m2 = w * g2 + m2;
    }
    const float hx = ((m0.x * rx + m0.y * ry) + m0.z * rz) + m0.w;
    const float hy = ((m1.x * rx + m1.y * ry) + m1.z * rz) + m1.w;
    const float hz = ((m2.x * rx + m2.y * ry) + m2.z * rz) + m2.w;
    const v4f rt = rootLane[ROWC];
    v3f ov = {hx * prev.winv + rt.x, hy * prev.winv + rt.y, hz * prev.winv + rt.z};
    __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(v3u, ov), rsV, prev.voff, soff, 2);
    // HAZARD (measured on gfx950, not covered by hipcc's hazard recogniser when soffset is an SGPR): a VALU write to the
    // data registers of a 96-bit buffer store in the very next instruction lands before the store has read dword 1 of
    // lanes 12-15 of each 16.  In the slot stream above an MFMA always follows a store; here keep one instruction of distance.
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 1");
    __builtin_amdgcn_sched_barrier(0);
  });
}

template<int MAXW, bool WANT_REST>
static hipError_t launch_b(const smplpp_model * m, int64_t n, const float * theta, float * verts, float * rest, hipStream_t st, int64_t f_off)
{
  const int nftp = (int)((n + 63) / 64);
  const int nvgp = (int)m->VGPn;
  const int total = nvgp * nftp;
  const int cus = device_cus(m->device);
  // per XCD: ceil(nvgp / 8) * nftp items at most; no more workgroups per XCD than that, and no more than the CUs it has
  const int per_xcd_items = ((nvgp + 7) / 8) * nftp;
  int nbx = cus / 8;
  if(nbx > per_xcd_items) nbx = per_xcd_items;
  if(nbx < 1) nbx = 1;
  const int blocks = nbx * 8;
  const int ipb = 0;
  (void)total;
  static PerDeviceOnce once;
  {
    hipError_t e = lds_opt_in(once, m->device, reinterpret_cast<const void *>(&skin_kernel_b<MAXW, WANT_REST>), B_LDS_TOTAL);
    if(e != hipSuccess) return e;
  }
  // f_off (a multiple of 64): first frame of this launch inside the workspace / caller arrays of a longer batch
  skin_kernel_b<MAXW, WANT_REST><<<dim3(blocks), dim3(256), B_LDS_TOTAL, st>>>(
      m->ws.A3.as<uint8_t>() + (f_off / 64) * (int64_t)(BB_KS * BB_A_BYTES), m->B3, m->ws.Gp.as<float>() + f_off * (NJ * 12),
      theta + f_off * ((NJ + 1) * 3), m->wIdx, m->wVal, m->wSum, verts ? verts + f_off * m->V * 3 : nullptr,
      rest ? rest + f_off * m->V * 3 : nullptr, n, m->V, nvgp, nftp, ipb);
  return hipGetLastError();
}

#if SKINB_ABL & 256
extern "C" int smplpp_debug_slot_times(unsigned long long * out)
{
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(smplpp_hip::g_slot_times), sizeof(unsigned long long) * 8 * 256);
}
#endif
// Gp must hold whole 64-frame tiles (padding content is irrelevant: the rows it feeds are never stored)
hipError_t launch_skin_bf16x3(const smplpp_model * m, int64_t n, const float * theta, float * verts, float * rest, hipStream_t st)
{
  // the kernel addresses its outputs with 32-bit buffer offsets: longer batches go in launches of <= 2 GiB of vertices
  int64_t per = (0x7fffff00LL / (m->V * 12)) & ~63LL;
  if(per < 64) return hipErrorInvalidValue;
  for(int64_t off = 0; off < n; off += per)
  {
    const int64_t nn = (n - off < per) ? n - off : per;
    hipError_t e;
    if(m->maxw == 4)
      e = rest ? launch_b<4, true>(m, nn, theta, verts, rest, st, off) : launch_b<4, false>(m, nn, theta, verts, rest, st, off);
    else if(m->maxw == 8)
      e = rest ? launch_b<8, true>(m, nn, theta, verts, rest, st, off) : launch_b<8, false>(m, nn, theta, verts, rest, st, off);
    else
      return hipErrorInvalidValue; // dense weights keep the first form
    if(e != hipSuccess) return e;
  }
  return hipSuccess;
}
} // namespace smplpp_hip
