// ik_solve_kernel — node.cpp:883-968: damped normal equations in fp64 built straight from J, LLT / box QP, configuration update
// (the header comment of ik.hip has the overview).  Included by ik.hip only.
#pragma once
#include "ik_types.h"

namespace smplpp_hip
{
// ------------------------------------------------------------------------------------------------ solve kernel
__device__ inline int tri_idx(int i, int j)
{
  return i * (i + 1) / 2 + j; // i >= j
}
__device__ inline void tri_unpack(int item, int & i, int & j)
{
  i = (int)((sqrt(8.0 * (double)item + 1.0) - 1.0) * 0.5);
  while(tri_idx(i + 1, 0) <= item) i++;
  while(tri_idx(i, 0) > item) i--;
  j = item - tri_idx(i, 0);
}

// In-place right-looking Cholesky of the packed lower-triangular (nf+1)x(nf+1) augmented matrix [A b; b' *] held in
// LDS (fp64): the last row becomes y = L^-1 b, so forward substitution is free.  Every thread of the workgroup
// updates the trailing sub-matrix; two barriers per column.
__device__ inline void chol_aug(double * M, int nf, int * bad, double * dinv)
{
  const int tid = threadIdx.x, nt = blockDim.x;
  for(int j = 0; j < nf; j++)
  {
    double d = M[tri_idx(j, j)];
    if(!(d > 0.0))
    {
      if(tid == 0) *bad = 1;
      d = 1.0;
    }
    const double piv = sqrt(d);
    __syncthreads(); // everyone has read the pivot
    if(tid == 0) dinv[j] = 1.0 / piv;
    for(int i = j + tid; i <= nf; i += nt) M[tri_idx(i, j)] = (i == j) ? piv : M[tri_idx(i, j)] / piv;
    __syncthreads();
    // trailing update: rows i in (j, nf], columns k in (j, i]; the 256 threads tile the square as 16 x 16
    {
      const int ty = tid >> 4, tx = tid & 15;
      for(int i = j + 1 + ty; i <= nf; i += 16)
      {
        const double lij = M[tri_idx(i, j)];
        const int kend = (i < nf) ? i : nf - 1; // the (nf, nf) corner is never used
        for(int k = j + 1 + tx; k <= kend; k += 16) M[tri_idx(i, k)] -= lij * M[tri_idx(k, j)];
      }
    }
    __syncthreads();
  }
}

// back substitution L^T x = y (y = row nf of M), x returned in xs[0..nf); dinv[j] = 1 / L[j][j].
// Inside ONE wavefront: lane l keeps x[l], x[l + 64], x[l + 128] in registers, the pivot value travels by v_readlane and
// row j of L is a contiguous LDS read that does not depend on the recurrence — no workgroup barrier per column (the
// barrier-per-column form spent ~2 x nf barriers of four wavefronts on a strictly sequential chain).
__device__ inline double readlane_f64(double v, int lane)
{
  const uint64_t u = __builtin_bit_cast(uint64_t, v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, lane);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), lane);
  return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
__device__ inline void back_subst(const double * M, int nf, double * xs, const double * dinv)
{
  const int tid = threadIdx.x;
  if(nf > 192) // (not reached by any mode of the reference: D <= 75 + 2 * 41 + 10)
  {
    for(int i = tid; i < nf; i += blockDim.x) xs[i] = M[tri_idx(nf, i)];
    __syncthreads();
    for(int j = nf - 1; j >= 0; j--)
    {
      const double xj = xs[j] * dinv[j];
      __syncthreads();
      for(int k = tid; k < j; k += blockDim.x) xs[k] -= M[tri_idx(j, k)] * xj;
      if(tid == 0) xs[j] = xj;
      __syncthreads();
    }
    return;
  }
  __syncthreads(); // M and dinv are complete
  if(tid < 64)
  {
    // One wavefront, lane i keeps x_i (+64, +128).  The loop is a chain of nf steps whose cost is its instruction count (a
    // step used to be ~60 instructions, ~280 cycles): lane j's entry is never touched after step j (the row entries of lanes
    // >= j are read as zero), so nobody "owns" a finished entry inside the loop — entries stay unscaled and take their
    // 1/L_jj once, at the end; rows are read without exec masks (a lane beyond the row reads the zero word instead).
    __shared__ double s_zero;
    if(tid == 0) s_zero = 0.0;
    double x[3];
#pragma unroll
    for(int a = 0; a < 3; a++) x[a] = (tid + 64 * a < nf) ? M[tri_idx(nf, tid + 64 * a)] : 0.0;
    __builtin_amdgcn_wave_barrier();
    // three segments by the number of accumulators a row still reaches (rows 128.., 64..127, 0..63), each a loop without
    // branches whose next row and pivot are requested one step ahead (two steps: rows[2])
    auto segment = [&](auto na_tag, int jhi, int jlo) {
      constexpr int NA = decltype(na_tag)::value;
      if(jhi < jlo) return;
      auto fetch = [&](int j, double (&l)[NA], double & d) {
        const double * Lj = M + tri_idx(j, 0);
#pragma unroll
        for(int a = 0; a < NA - 1; a++) l[a] = Lj[tid + 64 * a];
        l[NA - 1] = *((tid + 64 * (NA - 1) < j) ? Lj + tid + 64 * (NA - 1) : &s_zero);
        d = dinv[j];
      };
      double l0[NA], l1[NA], d0, d1;
      fetch(jhi, l0, d0);
      fetch(jhi - 1 >= jlo ? jhi - 1 : jlo, l1, d1);
      for(int j = jhi; j >= jlo; j--)
      {
        double lc[NA];
#pragma unroll
        for(int a = 0; a < NA; a++) lc[a] = l0[a];
        const double dc = d0;
#pragma unroll
        for(int a = 0; a < NA; a++) l0[a] = l1[a];
        d0 = d1;
        fetch(j - 2 >= jlo ? j - 2 : jlo, l1, d1);
        const double xj = readlane_f64(x[NA - 1], j - 64 * (NA - 1)) * dc;
#pragma unroll
        for(int a = 0; a < NA; a++) x[a] = fma(-lc[a], xj, x[a]);
      }
    };
    segment(std::integral_constant<int, 3>{}, nf - 1, 128);
    segment(std::integral_constant<int, 2>{}, nf - 1 < 127 ? nf - 1 : 127, 64);
    segment(std::integral_constant<int, 1>{}, nf - 1 < 63 ? nf - 1 : 63, 0);
#pragma unroll
    for(int a = 0; a < 3; a++)
      if(tid + 64 * a < nf) xs[tid + 64 * a] = x[a] * dinv[tid + 64 * a];
  }
  __syncthreads();
}

// 1/sqrt(d) in fp64: hardware estimate (v_rsq_f64) + two Newton steps (relative error ~1e-16), an order of magnitude
// cheaper than sqrt() + a division on the pivot's critical path.
__device__ inline double fast_rsqrt(double d)
{
  double y = __builtin_amdgcn_rsq(d);
  y = y * (1.5 - 0.5 * d * y * y);
  y = y * (1.5 - 0.5 * d * y * y);
  return y;
}

// HBM -> LDS copy of cnt doubles by the 256 threads of the workgroup: eight loads in flight per thread (a one-load-per-
// iteration loop pays the full memory latency nine times for a 24 x 87 Jacobian)
__device__ inline void stage_rows(double * dst, const double * __restrict__ src, int cnt)
{
  const int tid = threadIdx.x;
  for(int q0 = 0; q0 < cnt; q0 += 256 * 8)
  {
    double t[8];
#pragma unroll
    for(int u = 0; u < 8; u++)
    {
      const int q = q0 + u * 256 + tid;
      t[u] = src[q < cnt ? q : cnt - 1];
    }
#pragma unroll
    for(int u = 0; u < 8; u++)
    {
      const int q = q0 + u * 256 + tid;
      if(q < cnt) dst[q] = t[u];
    }
  }
}

// the first W columns of cr rows (row stride D in HBM) packed at stride W in LDS, sixteen loads in flight per thread; rl
// (nullable): the rows to take, by index
__device__ inline void stage_rows_cols(double * dst, const double * __restrict__ src, int cr, int W, int D, const int * rl = nullptr)
{
  // (the copy is a chain of HBM round trips, ~1.5 us each with a single workgroup pulling: sixteen loads in flight per thread —
  // the 164 x 75 block of a motion solve in three round trips instead of six)
  const int tid = threadIdx.x, cnt = cr * W;
  constexpr int U = 16;
  for(int q0 = 0; q0 < cnt; q0 += 256 * U)
  {
    double t[U];
#pragma unroll
    for(int u = 0; u < U; u++)
    {
      const int q = q0 + u * 256 + tid, qq = q < cnt ? q : cnt - 1;
      const int rr = qq / W;
      t[u] = src[(int64_t)(rl ? rl[rr] : rr) * D + (qq - rr * W)];
    }
#pragma unroll
    for(int u = 0; u < U; u++)
    {
      const int q = q0 + u * 256 + tid;
      if(q < cnt) dst[q] = t[u];
    }
  }
}

// The same copy by LDS-DMA (buffer_load_dwordx4 ... lds: memory -> LDS without a register in between, 16 bytes per lane, 64
// consecutive 16-byte LDS slots per instruction from per-lane addresses), all of a wavefront's pieces in flight at once: the 123 live
// rows x 75 columns of a motion solve (74 KB) are 74 instructions for the whole workgroup and arrive in about one memory round trip,
// where stage_rows_cols took three (of sixteen 8-byte loads per thread each, ~1.5-2 us apiece with a single workgroup pulling).
// LDS rows have the EVEN stride Wp = W + (W & 1) doubles, so that every lane's 16 bytes lie inside one row; a row's last lane may
// carry one double of column W (or, on the last column, of the next row): it lands in the pad slot nobody reads.  The source rows are
// only 8-byte aligned (odd D): dword-aligned buffer loads.  rl (LDS): the rows to take.  dst must have room for the count rounded
// up to 64 slots (the caller checks).
__device__ inline void stage_rows_cols_dma(double * dst, const double * __restrict__ src, int cr, int W, int D, const int * rl, int rows_total)
{
  typedef __attribute__((address_space(3))) void * lds_ptr_t;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int SPR = (W + 1) >> 1, cnt = cr * SPR; // 16-byte slots per row, in all
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(src), 0, rows_total * D * 8, 0x00020000);
  // every row index is read from LDS BEFORE the first DMA is issued (the compiler cannot tell the DMA's LDS destination from the
  // other arrays of the dynamic LDS block: an LDS read behind a DMA waits for vmcnt(0))
  constexpr int U = 24; // 24 x 256 slots of 16 bytes = 96 KiB per round
  const unsigned magic = (unsigned)((0x100000000ull + (unsigned)SPR - 1) / (unsigned)SPR); // floor(d / SPR) = umulhi(d, magic) for d < 2^25 / SPR >= 2^18
  for(int base = wave * 64; base < cnt; base += 256 * U)
  {
    int voff[U];
#pragma unroll
    for(int u = 0; u < U; u++)
    {
      const int dd = base + lane + 256 * u;
      const int row = (int)__umulhi((unsigned)dd, magic), within = dd - row * SPR;
      // (lanes past the end ask beyond the descriptor's range: nothing is fetched, zeros land in the slack behind the block)
      voff[u] = (dd < cnt) ? (rl[row] * D + 2 * within) * 8 : 0x7ffffff0;
    }
    __builtin_amdgcn_sched_barrier(0);
    SOLVE_STAMP(12);
#pragma unroll
    for(int u = 0; u < U; u++)
      if(base + 256 * u < cnt) // (wave-uniform)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(reinterpret_cast<unsigned char *>(dst) + (size_t)(base + 256 * u) * 16), 16, voff[u], 0, 0, 0);
    SOLVE_STAMP(13);
  }
  // (Issuing these from the kernel's set-up, on the guess that theta alone is free, was tried: the compiler cannot tell the DMA's LDS
  // destination from the other arrays of the same dynamic LDS block and waits for vmcnt(0) in front of the NEXT LDS access, so
  // nothing overlapped — stop-timed, round 4.  Measured alone (tools/micro/stage_probe.hip): 1.9 us for the 74 KB block, ~16 B/clk,
  // the same cold or warm and for 8- or 16-byte-aligned rows; 4-byte DMA 6.8 us; sixteen 8-byte register loads per thread 6.8 us.)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// rows of the staged block a chunk may hold when it goes through LDS-DMA at (even) row stride Weven; < 4: no DMA
__device__ inline int dma_chunk_rows(int chunk_rows, int D, int Weven)
{
  return (int)(((int64_t)chunk_rows * D) / Weven) - (128 + Weven - 1) / Weven; // (1 KiB of slack: the last instruction's tail)
}

// Factorisation + both substitutions of the packed (r + 1) x (r + 1) augmented matrix [S v; v' *] by ONE wavefront, lane i
// owning row i. Register form (r <= RMAX <= 32): the row lives in registers, a column's entries reach the other lanes by
// v_readlane (an SGPR operand of the FMA), so a column costs its pivot's rsqrt plus (r - k) FMAs and no LDS round trip;
// the factor is written back packed and re-read by columns (independent loads, hoisted) for the back substitution, whose
// chain is then readlane + FMA only. w[0..r) = S^-1 v.
// EXACT: r == RMAX is known where the call is made (the 6-target solve: 24), so the column loop carries no `k < r` branch and the whole
// factorisation is ONE basic block: the scheduler then starts column k + 1's pivot chain (two v_readlane, rsqrt estimate, two Newton
// steps: ~100 cycles of dependent latency) as soon as row k + 1 has taken column k's update, beside the remaining updates of column k.
template<int RMAX, bool EXACT = false>
__device__ inline void chol_wave_reg(double * M, int r, double * w, int * bad)
{
  const int i = threadIdx.x; // < 64
  const bool act = i <= r;
  double row[RMAX];
#pragma unroll
  for(int j = 0; j < RMAX; j++) row[j] = (act && j <= i && j < r) ? M[tri_idx(i, j)] : 0.0;
  double myrinv = 0.0;
  bool badl = false;
#pragma unroll
  for(int k = 0; k < RMAX; k++)
  {
    if(EXACT || k < r) // uniform
    {
      double piv = readlane_f64(row[k], k);
      if(!(piv > 0.0))
      {
        badl = true;
        piv = 1.0;
      }
      const double ri = fast_rsqrt(piv);
      const double l = row[k] * ri; // lane k: sqrt(piv); lanes below the diagonal: L[i][k]; the rhs lane r: y[k]
      row[k] = l;
      if(i == k) myrinv = ri;
#pragma unroll
      for(int j = k + 1; j < RMAX; j++) row[j] = fma(-l, readlane_f64(l, j), row[j]);
    }
  }
#pragma unroll
  for(int j = 0; j < RMAX; j++)
    if(act && j <= i && j < r) M[tri_idx(i, j)] = row[j];
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  double col[RMAX + 1]; // col[k] = L[k][i] for k > i (k == r: y[i])
  col[0] = 0.0;
#pragma unroll
  for(int k = 1; k <= RMAX; k++) col[k] = (i < k && k <= r && i < r) ? M[tri_idx(k, i)] : 0.0;
  double acc = 0.0;
#pragma unroll
  for(int k = RMAX; k >= 1; k--)
    if(k == r) acc = col[k];
#pragma unroll
  for(int k = RMAX - 1; k >= 0; k--)
  {
    if(EXACT || k < r) // uniform
    {
      const double wk = readlane_f64(acc, k) * readlane_f64(myrinv, k);
      acc = (i == k) ? wk : fma(-col[k], wk, acc); // col[k] is 0 for lanes i >= k
    }
  }
  if(i < r) w[i] = acc;
  if(badl && i == 0) *bad = 1;
}

// LDS form for 32 < r <= 63 (left-looking on the packed matrix)
__device__ inline void chol_wave_lds(double * M, int r, double * w, int * bad)
{
  const int tid = threadIdx.x;
  const int i = tid;
  const bool act = i <= r;
  const double * Li = M + tri_idx(act ? i : 0, 0);
  double myrinv = 0.0;
  bool badl = false;
  for(int k0 = 0; k0 < r; k0++)
  {
    const int k = __builtin_amdgcn_readfirstlane(k0);
    const double * Lk = M + tri_idx(k, 0);
    double s = 0.0;
    if(act && i >= k)
    {
      double s1 = 0.0;
      s = Li[k];
      int m = 0;
      for(; m + 1 < k; m += 2)
      {
        s -= Li[m] * Lk[m];
        s1 -= Li[m + 1] * Lk[m + 1];
      }
      if(m < k) s -= Li[m] * Lk[m];
      s += s1;
    }
    double piv = readlane_f64(s, k);
    if(!(piv > 0.0))
    {
      badl = true;
      piv = 1.0;
    }
    const double ri = fast_rsqrt(piv);
    if(act && i >= k) M[tri_idx(i, k)] = (i == k) ? piv * ri : s * ri;
    if(i == k) myrinv = ri;
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
  }
  double yv = (i < r) ? M[tri_idx(r, i)] : 0.0; // y = L^-1 v (the augmented row)
  for(int k0 = r - 1; k0 >= 0; k0--)
  {
    const int k = __builtin_amdgcn_readfirstlane(k0);
    const double lk = (i < k) ? M[tri_idx(k, 0) + i] : 0.0;
    const double wk = readlane_f64(yv, k) * readlane_f64(myrinv, k);
    yv = (i == k) ? wk : yv - lk * wk;
  }
  if(i < r) w[i] = yv;
  if(badl && tid == 0) *bad = 1;
}

// Dual form of the damped free-set system for FEWER RESIDUAL ROWS THAN FREE UNKNOWNS (r = 4K < nf; the 6-target solve has
// r = 24 against 75): with G = the diagonal damping (> 0, node.cpp:887-904) and J_F the free columns,
//   (G + J_F' J_F)^-1 c = G^-1 c - G^-1 J_F' (I + J_F G^-1 J_F')^-1 J_F G^-1 c
// so the Cholesky factorisation is r x r instead of nf x nf — the same x = -LLT(A)^-1 b of node.cpp:933-938 to fp64
// round-off (S = I + Jf Jf' with Jf = J_F G^-1/2 is at least as well conditioned as A). Steps: gather the free columns
// into LDS (eight loads in flight), c / u = G^-1 c and the column scaling (one thread per column), S and v = J u (one
// element per thread), factorisation + both substitutions inside ONE wavefront (left-looking on the packed LDS matrix with
// v as the augmented last row; pivots travel by v_readlane, no workgroup barrier per column), x = u - G^-1/2 Jf' w.
// Returns A^-1 c in xs[0..nf) like back_subst(). Needs r <= 63, r * nf doubles in Jf, nf in us/ginv, r in w.
// pre (nullable): the first 2048 gathered entries, loaded by the caller at kernel start on the GUESS that the free set is
// columns 0 .. pre_nf - 1 (true whenever only theta is free); used when the guess holds.
// (PRE is a template parameter and `pre` a reference to the caller's registers: as a nullable pointer the eight doubles lived
// in scratch memory and came back through flat loads)
template<bool PRE>
__device__ __forceinline__ void solve_dual(double * M, const double * __restrict__ J, const double * rowv, double * Jf, const double * diag,
                                           const double * bpri, const int * idx, int nf, int D, int r, double * ginv, double * us, double * w,
                                           double * xs, int * bad, int dbg_stop, const double (&pre)[8], int pre_nf)
{
  const int tid = threadIdx.x;
  const int cnt = r * nf;
  const bool use_pre = PRE && pre_nf == nf && idx[nf - 1] == nf - 1; // (ascending, distinct: then idx is the identity; uniform)
  for(int q0 = 0; q0 < cnt; q0 += 256 * 8)
  {
    double t[8];
    if(use_pre && q0 == 0)
    {
#pragma unroll
      for(int u = 0; u < 8; u++) t[u] = pre[u];
    }
    else
#pragma unroll
    for(int u = 0; u < 8; u++)
    {
      int q = q0 + u * 256 + tid;
      q = q < cnt ? q : cnt - 1;
      const int i = q / nf, a = q - i * nf;
      t[u] = J[(int64_t)i * D + idx[a]];
    }
#pragma unroll
    for(int u = 0; u < 8; u++)
    {
      const int q = q0 + u * 256 + tid;
      if(q < cnt) Jf[q] = t[u];
    }
  }
  __syncthreads();
  SOLVE_STAMP(2);
  if(dbg_stop == 31) return; // (timing experiments only)
  if(tid < nf)
  {
    const int a = tid, q = idx[a];
    double c = bpri[q];
    for(int i = 0; i < r; i++) c += Jf[i * nf + a] * rowv[i];
    const double gi = 1.0 / diag[q];
    const double sg = sqrt(gi);
    ginv[a] = gi;
    xs[a] = c * gi;  // u
    us[a] = c * sg;  // u / sg: v = J u = Jf (u / sg)
    for(int i = 0; i < r; i++) Jf[i * nf + a] *= sg;
  }
  __syncthreads();
  SOLVE_STAMP(3);
  if(dbg_stop == 32) return;
  {
    // S = I + Jf Jf' (r x r) and the augmented row v' = us' Jf' on the fp64 matrix pipe (round 4): one 16 x 16 tile of the lower
    // triangle of rows 0..r per wavefront and turn, the nf free columns as the k dimension, four per v_mfma_f64_16x16x4_f64 (operand
    // and result layout: build_and_factor_reg).  The 325 dot products of 75 terms, one or two per thread with two LDS reads per term,
    // took 5.1 us of the 6-target solve's 25.
    typedef double d4 __attribute__((ext_vector_type(4)));
    const int wave = tid >> 6, l = tid & 63, l16 = l & 15, lq = l >> 4;
    const int ntr = (r + 16) >> 4; // tile rows covering rows 0..r
    const int ntile = ntr * (ntr + 1) / 2;
    for(int t = wave; t < ntile; t += 4) // (wave-uniform)
    {
      int ta = 0, tb = t;
      while(tb > ta)
      {
        tb -= ta + 1;
        ta++;
      }
      const int ia = 16 * ta + l16, ib = 16 * tb + l16;
      const double * pa = (ia < r) ? Jf + ia * nf : us; // (row r: the rhs; rows beyond: masked below)
      const double * pb = Jf + (ib < r ? ib : 0) * nf;
      const bool la = ia <= r, lb = ib < r;
      d4 acc = {0.0, 0.0, 0.0, 0.0};
      for(int k0 = 0; k0 < nf; k0 += 16)
      {
        double a[4], b[4];
#pragma unroll
        for(int u = 0; u < 4; u++)
        {
          const int k = k0 + 4 * u + lq, kk = k < nf ? k : 0;
          a[u] = pa[kk];
          b[u] = pb[kk];
          if(!(la && k < nf)) a[u] = 0.0;
          if(!(lb && k < nf)) b[u] = 0.0;
        }
#pragma unroll
        for(int u = 0; u < 4; u++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
      }
#pragma unroll
      for(int rr = 0; rr < 4; rr++)
      {
        const int i = 16 * ta + 4 * rr + lq, j = 16 * tb + l16;
        if(i >= j && i <= r && j < r) M[tri_idx(i, j)] = acc[rr] + (i == j ? 1.0 : 0.0);
      }
    }
  }
  __syncthreads();
  SOLVE_STAMP(4);
  if(dbg_stop == 33) return;
  if(tid < 64)
  {
    switch((r + 7) >> 3)
    {
      case 1: chol_wave_reg<8>(M, r, w, bad); break;
      case 2: chol_wave_reg<16>(M, r, w, bad); break;
      case 3:
        if(r == 24)
          chol_wave_reg<24, true>(M, r, w, bad);
        else
          chol_wave_reg<24>(M, r, w, bad);
        break;
      case 4: chol_wave_reg<32>(M, r, w, bad); break;
      default: chol_wave_lds(M, r, w, bad); break;
    }
  }
  __syncthreads();
  SOLVE_STAMP(5);
  if(dbg_stop == 34) return;
  if(tid < nf)
  {
    const int a = tid;
    double t = 0.0;
    for(int i = 0; i < r; i++) t += Jf[i * nf + a] * w[i];
    xs[a] = xs[a] - sqrt(ginv[a]) * t;
  }
  __syncthreads();
}

// Register-tiled build + factorisation of the augmented free-set system for nf + 1 <= 16 * NT: thread (ty, tx) of the
// 16 x 16 workgroup owns the elements (ty + 16a, tx + 16b), b <= a, in registers.  Per column ONE barrier: the column's
// holders publish its raw entries (and the pivot entry) to LDS, every thread then applies the rank-1 update to its own
// registers as acc -= raw_i * raw_k / d.  The scaled column is also written to the packed LDS matrix M for back_subst.
template<int NT>
__device__ inline void build_and_factor_reg(double * M, const double * __restrict__ J, const double * __restrict__ rowv, double * Jc,
                                            const double * diag, const double * bpri, const int * idx, int nf, int D, int rows,
                                            int chunk_rows, double * lraw /*[2][4][16*NT]*/, double * ldiag /*[4]*/, double * dinv /*[nf]*/, int * bad,
                                            int dbg_stop, const int * rlist /*[nlive] rows of J that are not identically zero*/, int nlive)
{
  // thread (ty, tx): tx in the HIGH bits, so the 16 holders of a column (one tx, all ty) sit in one wavefront and the other
  // three skip the publish path (extraction, rsqrt, LDS writes) instead of executing it for four lanes each
  const int tid = threadIdx.x, tx = tid >> 4, ty = tid & 15;
  double acc[NT][NT];
  // A_FF = J_F^T J_F and the rhs row J_F^T rowv (the Gram of the augmented operand [J_F | rowv]) on the fp64 matrix pipe:
  // v_mfma_f64_16x16x4_f64, one 16 x 16 tile of the lower triangle per accumulator, the wavefronts take tiles round-robin,
  // the rows of J (staged in LDS in chunks) are the k dimension, four per MFMA.  Lane l feeds A[i = l % 16][k = l / 16] and
  // B[k = l / 16][j = l % 16] and receives D[4 r + l / 16][l % 16] in register r (probed: tools/micro/mfma_f64_layout.hip).
  // The tiles go through the packed LDS matrix M into the register layout of the factorisation below.
  typedef double d4 __attribute__((ext_vector_type(4)));
  const int nitemM = (nf + 1) * (nf + 2) / 2;
  // Two forms of the Gram loop.  ROWS SPLIT OVER THE WAVEFRONTS (round 4; tile counts up to 6, partial sums in the row chunk's LDS
  // once the chunk is dead): wavefront w takes the row groups w, w + 4, ... and accumulates EVERY live tile from them — per group of
  // four rows NT operand reads feed NT (NT + 1) / 2 MFMAs (5 reads for 15), where the tile-per-wavefront form below pays two reads
  // per MFMA and is a chain of read -> wait -> 4 MFMAs per group: 31 groups x ~600 cycles for a capture solve against 8 x ~1100.
  // The four partial sums are added as (w0 + w2) + (w1 + w3) on the way into the factorisation's register layout.
  const bool ksplit = NT <= 6 && 2 * (NT * (NT + 1) / 2) * 256 + nitemM <= chunk_rows * D; // (uniform: two raw partials + one packed triangle fit the row chunk)
  if(ksplit)
  {
    constexpr int NTILE = NT <= 6 ? NT * (NT + 1) / 2 : 1, NTK = NT <= 6 ? NT : 1;
    const int wave = tid >> 6, l = tid & 63, l16 = l & 15, lq = l >> 4;
    d4 tacc[NTILE];
    int colT[NTK]; // column of J (>= 0), -1 the rhs entry, -2 nothing, of this lane's element of tile row / tile column t
#pragma unroll
    for(int t = 0; t < NTK; t++)
    {
      const int m = 16 * t + l16;
      colT[t] = (m < nf) ? idx[m] : (m == nf ? -1 : -2);
    }
#pragma unroll
    for(int u = 0; u < NTILE; u++) tacc[u] = d4{0.0, 0.0, 0.0, 0.0};
    const int W = nf > 0 ? idx[nf - 1] + 1 : 1;
    const bool whole = W == D && nlive == rows;
    const int Weven = W + (W & 1);
    const int crows_dma = dma_chunk_rows(chunk_rows, D, Weven);
    const bool dma = !whole && W < D && crows_dma >= 4 && (int64_t)rows * D * 8 < 0x7fffff00LL;
    const int Wp = dma ? Weven : W;
    const int crows = dma ? crows_dma : (int)(((int64_t)chunk_rows * D) / W);
    if(dbg_stop == 40) return; // (timing experiments only)
    SOLVE_STAMP(2);
    for(int c0 = 0; c0 < nlive; c0 += crows)
    {
      const int cr = (nlive - c0 < crows) ? nlive - c0 : crows;
      __syncthreads();
      SOLVE_STAMP(3);
      if(whole)
        stage_rows(Jc, J + (int64_t)c0 * D, cr * D);
      else if(dma)
        stage_rows_cols_dma(Jc, J, cr, W, D, rlist + c0, rows);
      else
        stage_rows_cols(Jc, J, cr, W, D, rlist + c0);
      SOLVE_STAMP(4);
      __syncthreads();
      SOLVE_STAMP(5);
      if(dbg_stop == 41) return; // (timing experiments only)
      for(int r0 = 4 * wave; r0 < cr; r0 += 16)
      {
        const int r = r0 + lq;
        const bool rin = r < cr;
        const double rv = rowv[rlist[c0 + (rin ? r : 0)]];
        const double * Jr = Jc + (rin ? r : 0) * Wp;
        double v[NTK], va[NTK], vb[NTK];
#pragma unroll
        for(int t = 0; t < NTK; t++) v[t] = Jr[colT[t] >= 0 ? colT[t] : 0];
#pragma unroll
        for(int t = 0; t < NTK; t++)
        {
          va[t] = !rin ? 0.0 : (colT[t] >= 0 ? v[t] : (colT[t] == -1 ? rv : 0.0));
          vb[t] = (rin && colT[t] >= 0) ? v[t] : 0.0;
        }
#pragma unroll
        for(int ta = 0; ta < NTK; ta++)
#pragma unroll
          for(int tb = 0; tb <= ta; tb++)
          {
            if(!(16 * ta <= nf && 16 * tb < nf)) continue; // (uniform)
            d4 & t = tacc[ta * (ta + 1) / 2 + tb];
            t = __builtin_amdgcn_mfma_f64_16x16x4f64(va[ta], vb[tb], t, 0, 0, 0);
          }
      }
    }
    __syncthreads(); // every wavefront is done with the row chunk: its LDS takes the partial sums of wavefronts 1..3
    SOLVE_STAMP(6);
    if(dbg_stop == 42) return;
    // the four partial sums meet in two stages: wavefronts 2 and 3 drop theirs as they lie (lane-linear, [tile][register][lane]: no index
    // arithmetic, no bank conflicts), wavefronts 0 and 1 add them in registers — (w0 + w2), (w1 + w3) — and write the packed
    // triangles the factorisation's layout is gathered from, two reads per element instead of four
    constexpr int PRAW = NTILE * 4 * 64; // doubles of one raw partial
    double * const praw = Jc + (size_t)((wave & 1) * PRAW);
    double * const ptri = Jc + 2 * PRAW; // wavefront 1's packed triangle (wavefront 0's: M)
    if(wave >= 2)
    {
#pragma unroll
      for(int u = 0; u < NTILE; u++)
#pragma unroll
        for(int rr = 0; rr < 4; rr++) praw[(u * 4 + rr) * 64 + l] = tacc[u][rr];
    }
    __syncthreads();
    if(wave < 2)
    {
#pragma unroll
      for(int u = 0; u < NTILE; u++)
#pragma unroll
        for(int rr = 0; rr < 4; rr++) tacc[u][rr] += praw[(u * 4 + rr) * 64 + l];
      double * P = wave == 0 ? M : ptri;
#pragma unroll
      for(int ta = 0; ta < NTK; ta++)
#pragma unroll
        for(int tb = 0; tb <= ta; tb++)
        {
          if(!(16 * ta <= nf && 16 * tb < nf)) continue;
#pragma unroll
          for(int rr = 0; rr < 4; rr++)
          {
            const int i = 16 * ta + 4 * rr + lq, k = 16 * tb + l16;
            if(i >= k && i <= nf && k < nf) P[tri_idx(i, k)] = tacc[ta * (ta + 1) / 2 + tb][rr];
          }
        }
    }
    __syncthreads();
#pragma unroll
    for(int a2 = 0; a2 < NT; a2++)
#pragma unroll
      for(int b2 = 0; b2 <= a2; b2++)
      {
        const int i = ty + 16 * a2, k = tx + 16 * b2;
        double sum = 0.0;
        if(i >= k && i <= nf && k < nf)
        {
          const int q = tri_idx(i, k);
          sum = M[q] + ptri[q];
        }
        acc[a2][b2] = sum;
      }
    __syncthreads(); // M is rewritten by the factorisation
  }
  else
  {
    constexpr int NTILE = NT * (NT + 1) / 2, TPW = (NTILE + 3) / 4;
    const int wave = tid >> 6, l = tid & 63, l16 = l & 15, lq = l >> 4;
    d4 tacc[TPW];
    int colA[TPW], colB[TPW]; // column of J (>= 0), -1 the rhs entry, -2 nothing, of this lane's A / B operand element
    bool live[TPW];
#pragma unroll
    for(int u = 0; u < TPW; u++)
    {
      const int t = wave + 4 * u;
      int ta = 0, tb = t; // tile t of the row-major lower triangle: (ta, tb), tb <= ta
      while(tb > ta)
      {
        tb -= ta + 1;
        ta++;
      }
      live[u] = t < NTILE && 16 * ta <= nf && 16 * tb < nf; // (wave-uniform)
      const int mi = 16 * ta + l16, mk = 16 * tb + l16;
      colA[u] = (mi < nf) ? idx[mi] : (mi == nf ? -1 : -2);
      colB[u] = (mk < nf) ? idx[mk] : -2;
      tacc[u] = d4{0.0, 0.0, 0.0, 0.0};
    }
    // only the columns up to the last free one are staged (the free set is ascending), rows packed at that width: a motion
    // solve with its surface coordinates pinned reads 75 of its 157 columns — half the traffic, and all 164 rows in ONE chunk
    const int W = nf > 0 ? idx[nf - 1] + 1 : 1;
    // ... and only the rows that can be non-zero (rlist): a task without a normal term has a zero fourth row, a missing marker
    // four zero rows — a quarter of the 164 rows of a capture solve.  Zero rows add exact zeros: the sums keep their bits.
    // LDS row stride: W, or the next even number when the block goes through LDS-DMA (16-byte slots: stage_rows_cols_dma; W < D,
    // so that a row's pad slot is filled from inside the same source row)
    const bool whole = W == D && nlive == rows;
    const int Weven = W + (W & 1);
    const int crows_dma = dma_chunk_rows(chunk_rows, D, Weven);
    const bool dma = !whole && W < D && crows_dma >= 4 && (int64_t)rows * D * 8 < 0x7fffff00LL;
    const int Wp = dma ? Weven : W;
    const int crows = dma ? crows_dma : (int)(((int64_t)chunk_rows * D) / W);
    if(dbg_stop == 40) return; // (timing experiments only)
    SOLVE_STAMP(2);
    for(int c0 = 0; c0 < nlive; c0 += crows)
    {
      const int cr = (nlive - c0 < crows) ? nlive - c0 : crows;
      __syncthreads();
      SOLVE_STAMP(3);
      if(whole)
        stage_rows(Jc, J + (int64_t)c0 * D, cr * D);
      else if(dma)
        stage_rows_cols_dma(Jc, J, cr, W, D, rlist + c0, rows);
      else
        stage_rows_cols(Jc, J, cr, W, D, rlist + c0);
      SOLVE_STAMP(4);
      __syncthreads();
      SOLVE_STAMP(5);
      if(dbg_stop == 41) return; // (timing experiments only)
      for(int r0 = 0; r0 < cr; r0 += 4)
      {
        const int r = r0 + lq;
        const bool rin = r < cr;
        const double rv = rowv[rlist[c0 + (rin ? r : 0)]];
        const double * Jr = Jc + (rin ? r : 0) * Wp;
        // (every tile's two operands are read first, then the MFMAs: a read -> wait -> MFMA pair per tile paid the LDS round
        // trip TPW times per four rows)
        double ja[TPW], jb[TPW];
#pragma unroll
        for(int u = 0; u < TPW; u++)
        {
          ja[u] = Jr[colA[u] >= 0 ? colA[u] : 0];
          jb[u] = Jr[colB[u] >= 0 ? colB[u] : 0];
        }
#pragma unroll
        for(int u = 0; u < TPW; u++)
        {
          if(!live[u]) continue; // (wave-uniform)
          const double va = !rin ? 0.0 : (colA[u] >= 0 ? ja[u] : (colA[u] == -1 ? rv : 0.0));
          const double vb = (rin && colB[u] >= 0) ? jb[u] : 0.0;
          tacc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(va, vb, tacc[u], 0, 0, 0);
        }
      }
    }
    __syncthreads();
    SOLVE_STAMP(6);
    if(dbg_stop == 42) return;
#pragma unroll
    for(int u = 0; u < TPW; u++)
    {
      if(!live[u]) continue;
      const int t = wave + 4 * u;
      int ta = 0, q = t;
      while(q > ta)
      {
        q -= ta + 1;
        ta++;
      }
      const int tb = q;
#pragma unroll
      for(int rr = 0; rr < 4; rr++)
      {
        const int i = 16 * ta + 4 * rr + lq, k = 16 * tb + l16;
        if(i >= k && i <= nf && k < nf) M[tri_idx(i, k)] = tacc[u][rr];
      }
    }
    __syncthreads();
#pragma unroll
    for(int a2 = 0; a2 < NT; a2++)
#pragma unroll
      for(int b2 = 0; b2 <= a2; b2++)
      {
        const int i = ty + 16 * a2, k = tx + 16 * b2;
        acc[a2][b2] = (i >= k && i <= nf && k < nf) ? M[tri_idx(i, k)] : 0.0;
      }
    __syncthreads(); // M is rewritten by the factorisation
  }
  // damping on the diagonal, the prior's rhs on the augmented row: branch-free with every index read first, then every value (one
  // conditional block per tile, each a pair of dependent LDS round trips, was 4 of the 5 us between the Gram and the factorisation)
  {
    const int ar = nf >> 4; // (uniform) the tile row that holds the rhs row i == nf
    int id_[NT], ib_[NT];
#pragma unroll
    for(int a = 0; a < NT; a++)
    {
      const int i = ty + 16 * a, k = tx + 16 * a;
      id_[a] = idx[(ty == tx && i < nf) ? i : 0];
      ib_[a] = idx[k < nf ? k : 0];
    }
    double dv[NT], bv[NT];
#pragma unroll
    for(int a = 0; a < NT; a++)
    {
      dv[a] = diag[id_[a]];
      bv[a] = bpri[ib_[a]];
    }
#pragma unroll
    for(int a = 0; a < NT; a++)
    {
      const int i = ty + 16 * a;
      acc[a][a] += (ty == tx && i < nf) ? dv[a] : 0.0;
#pragma unroll
      for(int b = 0; b <= a; b++)
        acc[a][b] += (a == ar && i == nf && tx + 16 * b < nf) ? bv[b] : 0.0;
    }
  }
  if(dbg_stop == 4) return;
  SOLVE_STAMP(7);
  // factorisation, FOUR columns per barrier (round 2: two; the loop is a chain of barrier -> pivot reciprocals -> update, and
  // its length, not its arithmetic, is what it costs: 38 steps of ~1.9 k cycles for the 76 columns of a motion solve).  The
  // holders of columns j_0 .. j_3 publish their raw entries R_q; one barrier later every thread forms, from those four published
  // columns alone, the corrected columns
  //   C_q = R_q - sum_{s<q} C_s L_qs,   L_qs = C_s[j_q] / d_s,   d_s = C_s[j_s]        (an LDL^T of the 4 x 4 pivot block)
  // for the rows and columns it owns and applies the rank-4 update  a_ik -= sum_s C_s[i] C_s[k] / d_s  to its registers (the
  // elements of column j_q itself take only the terms s < q and so become C_q).  Nothing but four reciprocals sits between the
  // barrier and the update; square roots are taken once, after the loop.
  // lraw: [2 (step parity)][4 (column of the step)][16 NT], zeroed: rows beyond nf are never published.
  constexpr int LS = 16 * NT;
  bool kcol[NT]; // tx + 16 a is a column of the system (not the rhs row, not padding)
#pragma unroll
  for(int a = 0; a < NT; a++) kcol[a] = tx + 16 * a < nf;
  for(int q = tid; q < 8 * LS; q += 256) lraw[q] = 0.0;
  __syncthreads();
  int par = 0;
#pragma unroll
  for(int bj = 0; bj < NT; bj++)
  {
    for(int jj = 0; jj < 16; jj += 4)
    {
      const int j0 = 16 * bj + jj;
      if(j0 >= nf) break; // uniform
      const int ncol = nf - j0 < 4 ? nf - j0 : 4; // uniform: live columns of this step
      double * lb = lraw + (par & 1) * 4 * LS;
      par++;
      if(tx >= jj && tx < jj + ncol) // the holders publish (all four columns live in tile column bj)
      {
        double * lp = lb + (tx - jj) * LS;
        const int jc = j0 + (tx - jj);
#pragma unroll
        for(int a = bj; a < NT; a++)
        {
          const int i = ty + 16 * a;
          if(i >= jc && i <= nf) lp[i] = acc[a][bj];
        }
      }
      __syncthreads();
      // LDL^T of the pivot block from the published entries R_s[j_q], s <= q (broadcast reads); dead columns: inv = 0, L = 0
      double inv[4], L[4][4];
      {
        double Cj[4][4]; // Cj[s][q] = C_s[j_q], q >= s
#pragma unroll
        for(int sidx = 0; sidx < 4; sidx++)
        {
#pragma unroll
          for(int q = sidx; q < 4; q++)
          {
            double v = (q < ncol) ? lb[sidx * LS + j0 + q] : 0.0;
#pragma unroll
            for(int t = 0; t < sidx; t++) v -= Cj[t][q] * L[sidx][t];
            Cj[sidx][q] = v;
          }
          double d = Cj[sidx][sidx];
          if(sidx < ncol && !(d > 0.0)) *bad = 1;
          if(!(sidx < ncol && d > 0.0)) d = 1.0;
          double r = __builtin_amdgcn_rcp(d);
          r = r * (2.0 - d * r);
          r = r * (2.0 - d * r);
          inv[sidx] = (sidx < ncol) ? r : 0.0;
#pragma unroll
          for(int q = sidx + 1; q < 4; q++) L[q][sidx] = (q < ncol) ? Cj[sidx][q] * inv[sidx] : 0.0;
        }
      }
      double ri[4][NT], sk[4][NT]; // per live column s: C_s at this thread's rows, C_s / d_s at its columns (zero where the update does not apply)
#pragma unroll
      for(int a = bj; a < NT; a++)
      {
        const int i = ty + 16 * a, k = tx + 16 * a;
        double ci[4], ck[4];
#pragma unroll
        for(int q = 0; q < 4; q++)
        {
          double vi = lb[q * LS + i], vk = lb[q * LS + k];
#pragma unroll
          for(int t = 0; t < q; t++)
          {
            vi -= ci[t] * L[q][t];
            vk -= ck[t] * L[q][t];
          }
          // (entries above a column's pivot are never published: whatever the slot holds there is masked, here and below)
          ci[q] = (a > bj || i > j0 + q) ? vi : 0.0;
          ck[q] = (a > bj || k > j0 + q) ? vk : 0.0;
          ri[q][a] = (q < ncol) ? ci[q] : 0.0;
          sk[q][a] = (q < ncol && kcol[a]) ? ck[q] * inv[q] : 0.0;
        }
      }
      // (the elements of column j_q itself get only the terms s < q — their sk[s >= q] is zero — and so become the corrected column)
#pragma unroll
      for(int a = bj; a < NT; a++)
#pragma unroll
        for(int b = bj; b <= a; b++)
          acc[a][b] -= (ri[0][a] * sk[0][b] + ri[1][a] * sk[1][b]) + (ri[2][a] * sk[2][b] + ri[3][a] * sk[3][b]);
    }
  }
  __syncthreads();
  SOLVE_STAMP(8);
  // reciprocal pivots 1/sqrt(d_k) from the final diagonal entries, once
#pragma unroll
  for(int a = 0; a < NT; a++)
  {
    const int i = ty + 16 * a;
    if(ty == tx && i < nf)
    {
      double d = acc[a][a];
      if(!(d > 0.0))
      {
        *bad = 1;
        d = 1.0;
      }
      dinv[i] = fast_rsqrt(d);
    }
  }
  __syncthreads();
  // the scaled factor for the back substitution, once: a column's raw entries are final once its pair has been processed
  // (later updates only touch columns to its right), and L_ik = raw_ik / piv_k, piv_k = d_k / sqrt(d_k) = raw_kk * dinv_k
#pragma unroll
  for(int a = 0; a < NT; a++)
#pragma unroll
    for(int b = 0; b <= a; b++)
    {
      const int i = ty + 16 * a, k = tx + 16 * b;
      if(k < nf && i >= k && i <= nf) M[tri_idx(i, k)] = acc[a][b] * dinv[k];
    }
  __syncthreads();
}

// One workgroup per frame.  Everything is built from J (staged through LDS in row chunks) — no D x D matrix in HBM.
// LDS (doubles): M packed (D+1)(D+2)/2 | Jc [chunk][D] | xs, xfull, diag, bpri, lo, hi [D each] | rowv [rows] | lraw [2][96],
// ldiag [2] ; ints idx, state [D].
// DUAL_ONLY: the instantiation for launches whose every pass is known on the host to take the dual form (4K < theta_dim:
// theta is always free, so the free set never shrinks below the residual rows). It does not carry the register-tiled
// primal factorisation, which is what sizes the general kernel's register file footprint (247 of the SIMD's 512
// registers per lane: the face scan that runs beside the solve then keeps one wavefront per SIMD instead of three).
// NTR: tiles of 16 the register-tiled primal factorisation covers (free unknowns + 1 <= 16 NTR): 6 for every mode but the
// 41-marker body solve (phi and beta live: 167 free unknowns), which gets its own instantiation with 11 — a second tile
// count inside one instantiation slowed the common path by 16 us (its register file footprint).
template<bool DUAL_ONLY, int NTR = 6>
__global__ __launch_bounds__(256) void ik_solve_kernel(TaskArrays ta, const double * __restrict__ e_all, const double * __restrict__ J_all,
                                                       float * __restrict__ theta, float * __restrict__ beta, float * __restrict__ pts,
                                                       int K, int theta_dim, int beta_dim, int phi_live, int enable_qp, int use_prior,
                                                       int chunk_rows, const int * __restrict__ skip, double * __restrict__ e2_out,
                                                       int * __restrict__ status, int * __restrict__ sticky, double * __restrict__ x_out, int dbg_stop, int m_dim,
                                                       float * __restrict__ theta25, float * __restrict__ theta_copy,
                                                       unsigned * __restrict__ go_flag, unsigned * __restrict__ go_counter, unsigned go_tick,
                                                       unsigned * __restrict__ done_flag, unsigned * __restrict__ done_counter, unsigned done_tick)
{
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int64_t f = blockIdx.x;
  const int tid = threadIdx.x;
  SOLVE_STAMP(0);
  // "Every workgroup of this kernel is on its CU": the re-projection on the side stream waits for THIS, not for the end of the
  // evaluation.  Both kernels become ready at the same instant, and when the face scan's 1536 workgroups were dispatched first
  // the solve's (one per frame, a whole SIMD's registers per wavefront, 150 KB of LDS) waited for them to drain: 77 us became
  // 105-118 us in most frames of a capture fit, on the critical path.  Nothing is published here (what the scan reads was
  // written by the kernel before this one), so no drain: a counter and, from the last workgroup to arrive, the flag.
  if(go_flag && tid == 0)
  {
    if(__hip_atomic_fetch_add(go_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1)
    {
      __hip_atomic_store(go_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(go_flag, go_tick, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  const int D = theta_dim + 2 * K + beta_dim, rows = 4 * K;
  const int64_t tb = f * K;
  double * M = sm;
  double * Jc = M + (m_dim + 1) * (m_dim + 2) / 2; // m_dim >= the number of free unknowns (host bound): phi pinned => D - 2K
  double * xs = Jc + chunk_rows * D;
  double * xfull = xs + D;
  double * diag = xfull + D;
  double * bpri = diag + D;
  double * lo = bpri + D;
  double * hi = lo + D;
  double * rowv = hi + D;
  double * lraw = rowv + rows; // [2][4][16 NTR] published columns of the register-tiled factorisation (four per step)
  double * ldiag = lraw + 128 * NTR; // [4] (spare)
  double * dinv = ldiag + 4;   // [D] reciprocal pivots for the back substitution
  int * idx = reinterpret_cast<int *>(dinv + D);
  int * state = idx + D; // 0 free, -1 at lo, +1 at hi, 2 pinned (empty box)
  double * ebuf = reinterpret_cast<double *>(state + D); // [rows] the residual, read from HBM once
  __shared__ int s_bad, s_nf, s_block, s_bside, s_done, s_anybound, s_wcnt[4], s_wany[4];
  __shared__ int s_rlist[DUAL_ONLY ? 1 : IK_MAXK * 4], s_nlive; // rows of J that are not identically zero (primal form: build_and_factor_reg)
  __shared__ double s_alpha, s_e2;
  // Everything the set-up reads from HBM is requested NOW, in one round trip: the skip flag, the residual, this thread's limit
  // and prior entry — and, in the dual-only instantiation, the Jacobian block the dual form will gather if only theta turns out
  // free (it does unless a QP pass pins something).  One after the other they were four dependent round trips of ~1.5 us each
  // in a kernel whose whole length is 28 us.
  const double * J = J_all + f * rows * (int64_t)D;
  const int skipf = skip[f];
  double e_pre[(IK_MAXK * 4 + 255) / 256];
#pragma unroll
  for(int u = 0; u < (IK_MAXK * 4 + 255) / 256; u++) e_pre[u] = (tid + 256 * u < rows) ? e_all[f * rows + tid + 256 * u] : 0.0;
  const int my_i = tid < D ? tid : 0; // (D <= 256 on this path: the per-variable set-up below takes one variable per thread then)
  const bool my_phi = my_i >= theta_dim && my_i < theta_dim + 2 * K;
  const float pl_pre = (D <= 256 && my_phi && phi_live) ? ta.philim[tb + (my_i - theta_dim) / 2] : 0.0f;
  const float th_pre = (D <= 256 && use_prior && my_i < theta_dim) ? theta[f * theta_dim + my_i] : 0.0f;
  // primal form: the weight that decides whether a row of J can be non-zero (rows 4k .. 4k+2: the task's position weight —
  // a missing marker has none —, row 4k+3: its normal weight), for the second wavefront's row list
  float rl_pre[(IK_MAXK * 4 + 63) / 64];
  if constexpr(!DUAL_ONLY)
  {
#pragma unroll
    for(int c = 0; c < (IK_MAXK * 4 + 63) / 64; c++)
    {
      const int r = 64 * c + (tid & 63), k = (r < rows ? r : 0) >> 2;
      rl_pre[c] = ta.roww[(tb + k) * 2 + (((r & 3) == 3) ? 1 : 0)]; // (the evaluation's copy: see TaskArrays::roww)
    }
  }
  double j_pre[8];
  if constexpr(DUAL_ONLY)
  {
    const int cnt = rows * theta_dim;
#pragma unroll
    for(int u = 0; u < 8; u++)
    {
      int q = u * 256 + tid;
      q = q < cnt ? q : cnt - 1;
      const int i = q / theta_dim, a = q - i * theta_dim;
      j_pre[u] = J[(int64_t)i * D + a];
    }
  }
  if(skipf)
  {
    if(tid == 0 && e2_out) e2_out[f] = 0.0;
    if(pts)
      for(int k = tid; k < K * 3; k += 256) pts[tb * 3 + k] = ta.apos[tb * 3 + k];
    if(theta_copy)
      for(int i = tid; i < theta_dim; i += 256) theta_copy[f * theta_dim + i] = theta[f * theta_dim + i];
    wg_signal(done_flag, done_counter, done_tick); // (every workgroup of the grid counts itself in)
    return;
  }
  __builtin_amdgcn_s_setprio(3); // a latency chain: its few wavefronts issue ahead of the face scan that shares the CU
#pragma unroll
  for(int u = 0; u < (IK_MAXK * 4 + 255) / 256; u++)
    if(tid + 256 * u < rows) ebuf[tid + 256 * u] = e_pre[u];
  __syncthreads();
  const double * e = ebuf;
  if(tid < 64)
  {
    // |e|^2 (node.cpp:893; Eigen's squaredNorm reduces in packets, so no summation order is "the reference's"): each lane squares
    // and adds its own (up to three) rows, then a fixed butterfly over the 64 lanes — ~400 cycles.  Round 3 walked the rows in
    // ascending order by v_readlane, a chain of `rows` dependent fp64 FMAs: 2 us of a 41-marker solve's set-up.
    double v[3];
#pragma unroll
    for(int a = 0; a < 3; a++) v[a] = (tid + 64 * a < rows) ? e[tid + 64 * a] : 0.0;
    double s = 0.0;
    if(rows <= 192)
    {
      s = v[0] * v[0];
      s = fma(v[1], v[1], s);
      s = fma(v[2], v[2], s);
#pragma unroll
      for(int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    }
    else
      for(int r = 0; r < rows; r++) s += e[r] * e[r];
    if(tid == 0)
    {
      s_e2 = s;
      // bit 2 of the frame's word (this stream's evaluation raised it: a normal term on a vertex beyond MAXADJ faces): its Jacobian
      // rows are truncated, so the update is skipped like one whose factorisation failed — an enqueue-only caller never moves on a
      // wrong Jacobian, and reads the reason in smplpp_ik_get_status
      s_bad = (sticky[f] & 4) ? 1 : 0;
      s_done = 0;
      if(e2_out) e2_out[f] = s;
    }
  }
  else if(!DUAL_ONLY && tid < 128) // beside the sum: the rows of J that can be non-zero, ascending
  {
    const int l = tid - 64;
    int base = 0;
#pragma unroll
    for(int c = 0; c < (IK_MAXK * 4 + 63) / 64; c++)
    {
      const int r = 64 * c + l;
      const bool lv = r < rows && (((r & 3) == 3) ? (rl_pre[c] > 0.0f) : (rl_pre[c] != 0.0f));
      const unsigned long long m = __ballot(lv);
      if(lv) s_rlist[base + __popcll(m & ((1ull << l) - 1ull))] = r;
      base += __popcll(m);
    }
    if(l == 0) s_nlive = base;
  }
  __syncthreads();
  for(int i = tid; i < D; i += 256)
  {
    const double reg = (i < theta_dim) ? 1e-3 : (i < theta_dim + 2 * K ? 1e-1 : 1e-3); // node.cpp:887-892
    double dg = reg + s_e2;                                                               // :893
    double bp = 0.0;
    if(use_prior && i < theta_dim) // :895-904 (VPoser latent layout)
    {
      const double w = (i < 6) ? 0.0 : (i >= theta_dim - 6 ? 1e3 : 1e-5);
      dg += w;
      bp = w * (double)(D <= 256 ? th_pre : theta[f * theta_dim + i]);
    }
    diag[i] = dg;
    bpri[i] = bp;
    // bounds (node.cpp:916-928); theta is free
    double l = -1e30, h = 1e30;
    int st = 0;
    if(i >= theta_dim && i < theta_dim + 2 * K)
    {
      const double pl = phi_live ? (double)(D <= 256 ? pl_pre : ta.philim[tb + (i - theta_dim) / 2]) : 0.0;
      if(enable_qp)
      {
        l = -pl;
        h = pl;
      }
      // with a zero limit the phi columns of J are zero: the QP pins x_phi = 0 and the LLT solution of the
      // block-diagonal system has x_phi = 0 as well, so the variable is removed from the system in both modes
      if(!(pl > 0.0))
      {
        l = 0.0;
        h = 0.0;
        st = 2;
      }
    }
    else if(i >= theta_dim + 2 * K && enable_qp)
    {
      l = -0.5; // :925
      h = 0.5;
    }
    lo[i] = l;
    hi[i] = h;
    state[i] = st;
    xfull[i] = 0.0;
  }
  __syncthreads();

  SOLVE_STAMP(1);
  if(dbg_stop == 1) return; // (timing experiments only: SMPLPP_IK_DBG_STOP)
  const int max_it = enable_qp ? 4 * D + 20 : 1;
  for(int it = 0; it < max_it; it++)
  {
    if(D <= 256)
    {
      // free-set index list in ascending order: one variable per thread, ballot + prefix over the four wavefronts
      // (a single thread walking `state` pays an LDS round trip per variable)
      int st = 2;
      if(tid < D) st = state[tid];
      const int is_free = (st == 0), is_b = ((st == -1 || st == 1) && xfull[tid < D ? tid : 0] != 0.0);
      const uint64_t m = __ballot(is_free);
      const int wave = tid >> 6, lane = tid & 63;
      // (a ballot per wavefront and one barrier: __syncthreads_or funnels every thread through an LDS atomic — 2.7 us here, stamped)
      const uint64_t mbnd = __ballot(is_b);
      if(lane == 0)
      {
        s_wcnt[wave] = __popcll(m);
        s_wany[wave] = mbnd != 0 ? 1 : 0;
      }
      __syncthreads();
      const int anyb = s_wany[0] | s_wany[1] | s_wany[2] | s_wany[3];
      int base = 0;
      for(int w = 0; w < wave; w++) base += s_wcnt[w];
      if(is_free) idx[base + __popcll(m & ((1ull << lane) - 1ull))] = tid;
      if(tid == 0)
      {
        s_nf = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
        s_anybound = anyb;
        s_alpha = 1.0;
        s_block = -1;
      }
    }
    else if(tid == 0)
    {
      int nf = 0, anyb = 0;
      for(int i = 0; i < D; i++)
      {
        if(state[i] == 0) idx[nf++] = i;
        if((state[i] == -1 || state[i] == 1) && xfull[i] != 0.0) anyb = 1;
      }
      s_nf = nf;
      s_anybound = anyb;
      s_alpha = 1.0;
      s_block = -1;
    }
    __syncthreads();
    SOLVE_STAMP(14);
    const int nf = s_nf;
    if(nf > m_dim) // (cannot happen: the host bound counts every variable that can be free)
    {
      if(tid == 0) s_bad = 1;
      __syncthreads();
      break;
    }
    const int nitem = (nf + 1) * (nf + 2) / 2;
    // rowv = e + J_B x_B  (b_F + A_FB x_B = J_F^T rowv); A = J^T J, b = J^T e (node.cpp:884-885), fp64
    for(int r = tid; r < rows; r += 256)
    {
      double s = e[r];
      if(s_anybound)
        for(int q = 0; q < D; q++)
          if(state[q] == -1 || state[q] == 1) s += J[(int64_t)r * D + q] * xfull[q];
      rowv[r] = s;
    }
    SOLVE_STAMP(15);
    const bool dual = rows < nf && rows <= 63 && chunk_rows >= rows && nf <= 192 && (DUAL_ONLY || dbg_stop != 9);
    if(DUAL_ONLY && !dual) // (cannot happen: the host selects this instantiation only when every pass qualifies)
    {
      if(tid == 0) s_bad = 1;
      __syncthreads();
      break;
    }
    if(dual)
    {
      __syncthreads();
      solve_dual<DUAL_ONLY>(M, J, rowv, Jc, diag, bpri, idx, nf, D, rows, dinv, lraw, lraw + 192, xs, &s_bad, dbg_stop, j_pre, theta_dim);
      if(dbg_stop >= 31 && dbg_stop <= 34) return;
    }
    else if constexpr(DUAL_ONLY)
    {
    }
    else if(nf + 1 <= 16 * NTR)
    {
      // registers, one barrier per column
      __syncthreads();
      build_and_factor_reg<NTR>(M, J, rowv, Jc, diag, bpri, idx, nf, D, rows, chunk_rows, lraw, ldiag, dinv, &s_bad, dbg_stop, s_rlist, s_nlive);
      if(dbg_stop == 4 || dbg_stop == 40 || dbg_stop == 41 || dbg_stop == 42) return;
    }
    else
    {
      for(int item = tid; item < nitem; item += 256) M[item] = 0.0;
      for(int c0 = 0; c0 < rows; c0 += chunk_rows)
      {
        const int cr = (rows - c0 < chunk_rows) ? rows - c0 : chunk_rows;
        __syncthreads();
        stage_rows(Jc, J + (int64_t)c0 * D, cr * D);
        __syncthreads();
        {
          const int ty = tid >> 4, tx = tid & 15; // 16 x 16 tiling of the lower triangle (+ the rhs row i == nf)
          for(int i = ty; i <= nf; i += 16)
          {
            const int ci = (i < nf) ? idx[i] : 0;
            const int jend = (i < nf) ? i : nf - 1;
            for(int j = tx; j <= jend; j += 16)
            {
              const int cj = idx[j];
              double s = 0.0;
              if(i < nf)
                for(int r = 0; r < cr; r++) s += Jc[r * D + ci] * Jc[r * D + cj];
              else
                for(int r = 0; r < cr; r++) s += Jc[r * D + cj] * rowv[c0 + r];
              M[tri_idx(i, j)] += s;
            }
          }
        }
      }
      __syncthreads();
      for(int a = tid; a < nf; a += 256)
      {
        M[tri_idx(a, a)] += diag[idx[a]];
        M[tri_idx(nf, a)] += bpri[idx[a]];
      }
      __syncthreads();
      chol_aug(M, nf, &s_bad, dinv);
    }
    if(dbg_stop == 2) return;
    SOLVE_STAMP(9);
    if constexpr(!DUAL_ONLY)
      if(!dual) back_subst(M, nf, xs, dinv);
    SOLVE_STAMP(10);
    if(dbg_stop == 3) return;
    if(!enable_qp)
    {
      for(int a = tid; a < nf; a += 256) xfull[idx[a]] = -xs[a]; // x = -LLT(A)^-1 b (node.cpp:938)
      __syncthreads();
      break;
    }
    // candidate x_F = -xs ; ratio test against the box: the first variable (ascending free-set order) with the smallest
    // step fraction below 1 blocks.  One variable per thread + a lexicographic (fraction, index) minimum — a single thread
    // walking the free set pays five LDS round trips per variable (10 us for the 75 unknowns of a motion solve)
    if(nf <= 256)
    {
      double al = 2.0;
      int who = 0x7fffffff, side = 0;
      if(tid < nf)
      {
        const int i = idx[tid];
        const double xn = -xs[tid], xo = xfull[i], dx = xn - xo;
        if(xn > hi[i] + 1e-14 && dx > 0)
        {
          al = (hi[i] - xo) / dx;
          side = 1;
        }
        else if(xn < lo[i] - 1e-14 && dx < 0)
        {
          al = (lo[i] - xo) / dx;
          side = -1;
        }
        if(side != 0 && al < 1.0)
          who = tid;
        else
          al = 2.0;
      }
      for(int o = 32; o > 0; o >>= 1)
      {
        const double oal = __shfl_xor(al, o, 64);
        const int owho = __shfl_xor(who, o, 64), oside = __shfl_xor(side, o, 64);
        if(oal < al || (oal == al && owho < who))
        {
          al = oal;
          who = owho;
          side = oside;
        }
      }
      __shared__ double s_ral[4];
      __shared__ int s_rwho[4], s_rside[4];
      if((tid & 63) == 0)
      {
        s_ral[tid >> 6] = al;
        s_rwho[tid >> 6] = who;
        s_rside[tid >> 6] = side;
      }
      __syncthreads();
      if(tid == 0)
      {
        double alpha = 1.0;
        int block = -1, bside = 0;
        for(int w = 0; w < 4; w++)
          if(s_rwho[w] != 0x7fffffff && s_ral[w] < alpha) // ascending wavefront order: ties keep the lower index
          {
            alpha = s_ral[w];
            block = s_rwho[w];
            bside = s_rside[w];
          }
        s_alpha = alpha;
        s_block = block;
        s_bside = bside;
      }
    }
    else if(tid == 0)
    {
      double alpha = 1.0;
      int block = -1, bside = 0;
      for(int a = 0; a < nf; a++)
      {
        const int i = idx[a];
        const double xn = -xs[a], dx = xn - xfull[i];
        if(xn > hi[i] + 1e-14 && dx > 0)
        {
          const double al = (hi[i] - xfull[i]) / dx;
          if(al < alpha) { alpha = al; block = a; bside = 1; }
        }
        else if(xn < lo[i] - 1e-14 && dx < 0)
        {
          const double al = (lo[i] - xfull[i]) / dx;
          if(al < alpha) { alpha = al; block = a; bside = -1; }
        }
      }
      s_alpha = alpha;
      s_block = block;
      s_bside = bside;
    }
    __syncthreads();
    for(int a = tid; a < nf; a += 256) xfull[idx[a]] += s_alpha * (-xs[a] - xfull[idx[a]]);
    __syncthreads();
    if(s_block >= 0)
    {
      if(tid == 0)
      {
        const int i = idx[s_block];
        state[i] = s_bside;
        xfull[i] = s_bside > 0 ? hi[i] : lo[i];
      }
      __syncthreads();
      continue;
    }
    // nothing sits on a bound (every motion-stage solve: phi pinned, beta fixed): the unconstrained step is the optimum
    {
      int atb = 0;
      for(int i = tid; i < D; i += 256) atb |= (state[i] == -1 || state[i] == 1);
      const uint64_t matb = __ballot(atb);
      __syncthreads(); // (s_wany's readers of the free-set step are long past)
      if((tid & 63) == 0) s_wany[tid >> 6] = matb != 0 ? 1 : 0;
      __syncthreads();
      if(!(s_wany[0] | s_wany[1] | s_wany[2] | s_wany[3]))
      {
        if(tid == 0) s_done = 1;
        __syncthreads();
        break;
      }
    }
    // multipliers of the bound variables: g = A x + b = J^T (e + J x) + diag x + bpri.  One wavefront per row, lanes
    // across the columns (a thread per row reads J with a stride of D doubles: 64 cache lines per load instruction)
    for(int r = tid >> 6; r < rows; r += 4)
    {
      double s = 0.0;
      for(int q = tid & 63; q < D; q += 64)
        if(state[q] != 2) s += J[(int64_t)r * D + q] * xfull[q];
      for(int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      if((tid & 63) == 0) rowv[r] = e[r] + s;
    }
    __syncthreads();
    for(int i = tid; i < D; i += 256)
    {
      double viol = 0.0;
      if(state[i] == -1 || state[i] == 1)
      {
        double g = diag[i] * xfull[i] + bpri[i];
        for(int r = 0; r < rows; r++) g += J[(int64_t)r * D + i] * rowv[r];
        viol = (state[i] < 0) ? -g : g; // at lo need g >= 0; at hi need g <= 0
      }
      xs[i] = viol; // xs is free between solves
    }
    __syncthreads();
    if(tid == 0)
    {
      double worst = 1e-12;
      int rel = -1;
      for(int i = 0; i < D; i++)
        if(xs[i] > worst) { worst = xs[i]; rel = i; }
      if(rel < 0)
        s_done = 1;
      else
        state[rel] = 0;
    }
    __syncthreads();
    if(s_done) break;
  }
  if(tid == 0)
  {
    status[f] = s_bad ? 1 : ((enable_qp && !s_done) ? 2 : 0);
    if(s_bad && !(sticky[f] & 4)) sticky[f] = sticky[f] | 1; // survives later solves (sequence driver); bit 2 is the evaluation's (TaskArrays::flags)
  }
  const bool ok = !s_bad;
  // config update (node.cpp:945-968), fp32
  for(int i = tid; i < theta_dim; i += 256)
  {
    float t = theta[f * theta_dim + i];
    if(ok)
    {
      t = t + (float)xfull[i];
      // (done_flag: the decoder's Jacobian kernel on the side stream reads the new latent behind that flag — write-through, signal.h)
      if(done_flag)
        st_agent(&theta[f * theta_dim + i], t);
      else
        theta[f * theta_dim + i] = t;
      // VPoser latent layout: the entries that pass through to theta25 (node.cpp:763-771) are kept current here
      if(theta25 && i < 6) theta25[f * TD75 + i] = t;
      if(theta25 && i >= 38) theta25[f * TD75 + 69 + (i - 38)] = t;
    }
    if(theta_copy) theta_copy[f * theta_dim + i] = t; // the sequence driver's record of this frame's result (last iteration of a frame)
  }
  for(int i = tid; i < beta_dim; i += 256)
    if(ok) beta[f * NB + i] += (float)xfull[theta_dim + 2 * K + i];
  for(int i = tid; pts && i < K * 3; i += 256) // p_k = actualPos_k + tangents_k . x_phi_k (:956-959); null: x_phi = 0 for all
  {
    const int k = i / 3, x = i % 3;
    const float p0 = ok ? (float)xfull[theta_dim + 2 * k] : 0.0f, p1 = ok ? (float)xfull[theta_dim + 2 * k + 1] : 0.0f;
    pts[tb * 3 + i] = ta.apos[tb * 3 + i] + (ta.tang[(tb + k) * 6 + x * 2] * p0 + ta.tang[(tb + k) * 6 + x * 2 + 1] * p1);
  }
  if(x_out)
    for(int i = tid; i < D; i += 256) x_out[f * D + i] = xfull[i];
  SOLVE_STAMP(11);
  // "this configuration is final": what the capture loops' side stream waits for before it makes the NEXT decoder Jacobian
  wg_signal(done_flag, done_counter, done_tick);
}
} // namespace smplpp_hip
