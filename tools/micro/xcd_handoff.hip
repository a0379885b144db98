// Micro-benchmark (dev probe): workgroup-to-workgroup hand-over INSIDE one kernel across XCDs, with the memory-model
// primitives the compiler provides (release add / acquire load at agent scope): is the data another XCD wrote always the
// new one, and what does a round cost?  256 workgroups (one per CU, all resident), per round: every workgroup writes 2 KiB,
// signals one of 16 counters (like the frame tiles of skin_kernel_h: 16 producers each), waits for the counter of the tile it
// consumes, reads the 16 x 2 KiB of that tile and checks every word.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/xcd_handoff.hip -o /tmp/xcd_handoff ; run: /tmp/xcd_handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

constexpr int NWG = 256, NT = 16, WORDS = 512; // per workgroup and round: 512 words = 2 KiB

__device__ __forceinline__ unsigned val(int round, int wg, int i)
{
  return (unsigned)round * 2654435761u + (unsigned)wg * 40503u + (unsigned)i;
}

template<int MODE> // 0: release add + acquire poll (agent scope); 1: relaxed add / relaxed poll + explicit fences
__global__ __launch_bounds__(256, 1) void handoff(unsigned * data, unsigned * counters, int rounds, unsigned * errors, unsigned long long * times)
{
  const int b = blockIdx.x, tid = threadIdx.x;
  const int my_tile = b & (NT - 1);            // producers of tile t: workgroups with b % 16 == t (spread over all XCDs)
  const int cons_tile = (b / NT) & (NT - 1);   // the tile this workgroup consumes
  unsigned bad = 0;
  unsigned long long t_wait = 0, t0 = 0;
  for(int r = 0; r < rounds; r++)
  {
    unsigned * mine = data + ((size_t)(r & 1) * NWG + b) * WORDS;
    for(int i = tid; i < WORDS; i += 256) mine[i] = val(r, b, i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if(tid == 0)
    {
      t0 = __builtin_amdgcn_s_memrealtime();
      if(MODE == 0)
        __hip_atomic_fetch_add(&counters[my_tile * 32], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      else
      {
        __atomic_thread_fence(__ATOMIC_RELEASE); // (system scope)
        __hip_atomic_fetch_add(&counters[my_tile * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      const unsigned want = (unsigned)(r + 1) * (NWG / NT);
      long long spins = 0;
      while(__hip_atomic_load(&counters[cons_tile * 32], MODE == 0 ? __ATOMIC_ACQUIRE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want)
      {
        __builtin_amdgcn_s_sleep(2);
        if(++spins > (1ll << 24)) break; // never hang the box
      }
      if(MODE != 0) __atomic_thread_fence(__ATOMIC_ACQUIRE);
      t_wait += __builtin_amdgcn_s_memrealtime() - t0;
    }
    __syncthreads();
    // read the consumed tile: 16 producers x 512 words
    for(int p = 0; p < NWG / NT; p++)
    {
      const int src = p * NT + cons_tile;
      const unsigned * theirs = data + ((size_t)(r & 1) * NWG + src) * WORDS;
      for(int i = tid; i < WORDS; i += 256) bad += (theirs[i] != val(r, src, i));
    }
    __syncthreads(); // (everyone has read before anyone's next-but-one round may overwrite: rounds alternate buffers)
  }
  if(bad) atomicAdd(errors, bad);
  if(tid == 0) times[b] = t_wait;
}

int main()
{
  unsigned *data, *counters, *errors;
  unsigned long long * times;
  hipMalloc(&data, (size_t)2 * NWG * WORDS * 4);
  hipMalloc(&counters, NT * 32 * 4);
  hipMalloc(&errors, 4);
  hipMalloc(&times, NWG * 8);
  for(int mode = 0; mode < 2; mode++)
    for(int rounds : {1, 200, 5000})
    {
      hipMemset(counters, 0, NT * 32 * 4);
      hipMemset(errors, 0, 4);
      hipMemset(data, 0xff, (size_t)2 * NWG * WORDS * 4);
      hipEvent_t a, b;
      hipEventCreate(&a);
      hipEventCreate(&b);
      hipEventRecord(a, 0);
      if(mode == 0)
        handoff<0><<<NWG, 256>>>(data, counters, rounds, errors, times);
      else
        handoff<1><<<NWG, 256>>>(data, counters, rounds, errors, times);
      hipEventRecord(b, 0);
      hipEventSynchronize(b);
      float ms = 0;
      hipEventElapsedTime(&ms, a, b);
      unsigned e = 0;
      std::vector<unsigned long long> t(NWG);
      hipMemcpy(&e, errors, 4, hipMemcpyDeviceToHost);
      hipMemcpy(t.data(), times, NWG * 8, hipMemcpyDeviceToHost);
      double mean = 0;
      for(auto v : t) mean += (double)v;
      mean /= NWG;
      printf("mode %d (%s), %5d rounds: %.2f us per round (kernel), signal+wait %.2f us per round (mean over workgroups), stale words %u\n", mode,
             mode == 0 ? "release add / acquire poll" : "fences + relaxed", rounds, ms * 1e3 / rounds, mean / 100.0 / rounds, e);
    }
  return 0;
}
