// smplpp_gather: the one collective of the path for a C++ multi-GPU host — result rows of every rank,
// in rank order, over RCCL.  RCCL is bound at run time: the symbols already loaded in the process (a host that created the
// communicator has them), else librccl.so.
#include "common.h"

#include <dlfcn.h>

#include <climits>

namespace
{
using nccl_fn_allgather = int (*)(const void *, void *, size_t, int, void *, hipStream_t);
using nccl_fn_bcast = int (*)(const void *, void *, size_t, int, int, void *, hipStream_t);
using nccl_fn_send = int (*)(const void *, size_t, int, int, void *, hipStream_t);
using nccl_fn_recv = int (*)(void *, size_t, int, int, void *, hipStream_t);
using nccl_fn_group = int (*)();
using nccl_fn_errstr = const char * (*)(int);
constexpr int NCCL_FLOAT32 = 7; // ncclFloat32 (rccl.h: ncclDataType_t)

struct Rccl
{
  nccl_fn_allgather allgather = nullptr;
  nccl_fn_bcast bcast = nullptr;
  nccl_fn_send send = nullptr;
  nccl_fn_recv recv = nullptr;
  nccl_fn_group gstart = nullptr, gend = nullptr;
  nccl_fn_errstr errstr = nullptr;
  bool tried = false;
};

Rccl & rccl()
{
  static Rccl r;
  if(r.tried) return r;
  r.tried = true;
  void * h = RTLD_DEFAULT;
  if(!dlsym(h, "ncclAllGather"))
  {
    h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if(!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if(!h) return r;
  }
  r.allgather = reinterpret_cast<nccl_fn_allgather>(dlsym(h, "ncclAllGather"));
  r.bcast = reinterpret_cast<nccl_fn_bcast>(dlsym(h, "ncclBroadcast"));
  r.send = reinterpret_cast<nccl_fn_send>(dlsym(h, "ncclSend"));
  r.recv = reinterpret_cast<nccl_fn_recv>(dlsym(h, "ncclRecv"));
  r.gstart = reinterpret_cast<nccl_fn_group>(dlsym(h, "ncclGroupStart"));
  r.gend = reinterpret_cast<nccl_fn_group>(dlsym(h, "ncclGroupEnd"));
  r.errstr = reinterpret_cast<nccl_fn_errstr>(dlsym(h, "ncclGetErrorString"));
  return r;
}
} // namespace

using namespace smplpp_hip;

// Slot of every rank's block in the gathered array (rank order, dist.shard_sizes): offsets in floats, [world + 1] entries, the
// last one the total.  Pure host arithmetic, shared by both collectives below (and testable without a GPU).
extern "C" int smplpp_gather_offsets(const int64_t * rows_per_rank, int world, int64_t row_floats, int64_t * offsets)
{
  if(!rows_per_rank || !offsets || world <= 0 || row_floats <= 0) return fail(SMPLPP_ERR_INVALID, "smplpp_gather_offsets: bad argument");
  int64_t off = 0;
  for(int i = 0; i < world; i++)
  {
    if(rows_per_rank[i] < 0) return fail(SMPLPP_ERR_INVALID, "smplpp_gather_offsets: negative row count");
    if(rows_per_rank[i] > 0 && row_floats > INT64_MAX / rows_per_rank[i]) return fail(SMPLPP_ERR_INVALID, "smplpp_gather_offsets: overflow");
    offsets[i] = off;
    if(off > INT64_MAX - rows_per_rank[i] * row_floats) return fail(SMPLPP_ERR_INVALID, "smplpp_gather_offsets: overflow");
    off += rows_per_rank[i] * row_floats;
  }
  offsets[world] = off;
  return SMPLPP_OK;
}

// Gather to ONE rank: every other rank sends its block once, on its own xGMI link, straight into its slot of root's array
// (grouped ncclSend / ncclRecv) — an eighth of the all-gather's traffic at eight ranks, and no rank but root holds the result.
extern "C" int smplpp_gather_to_root(void * comm, const float * send, float * recv, const int64_t * rows_per_rank, int world, int rank,
                                     int root, int64_t row_floats, void * stream)
{
  if(!comm || !rows_per_rank || world <= 0 || rank < 0 || rank >= world || root < 0 || root >= world || row_floats <= 0)
    return fail(SMPLPP_ERR_INVALID, "smplpp_gather_to_root: bad argument");
  if(rank == root && !recv) return fail(SMPLPP_ERR_INVALID, "smplpp_gather_to_root: root needs a receive array");
  std::vector<int64_t> off((size_t)world + 1);
  int rc = smplpp_gather_offsets(rows_per_rank, world, row_floats, off.data());
  if(rc) return rc;
  if(off[world] == 0) return SMPLPP_OK;
  if(rows_per_rank[rank] > 0 && !send) return fail(SMPLPP_ERR_INVALID, "smplpp_gather_to_root: null send block");
  Rccl & r = rccl();
  if(!r.send || !r.recv || !r.gstart || !r.gend)
    return fail(SMPLPP_ERR_HIP, "smplpp_gather_to_root: RCCL is not available (librccl.so could not be loaded)");
  hipStream_t st = static_cast<hipStream_t>(stream);
  auto check = [&](int e, const char * what) -> int {
    if(e == 0) return (int)SMPLPP_OK;
    return fail(SMPLPP_ERR_HIP, std::string("smplpp_gather_to_root: ") + what + " failed: " + (r.errstr ? r.errstr(e) : "RCCL error"));
  };
  if(rank != root)
  {
    if(rows_per_rank[rank] == 0) return SMPLPP_OK;
    return check(r.send(send, (size_t)(rows_per_rank[rank] * row_floats), NCCL_FLOAT32, root, comm, st), "ncclSend");
  }
  // root: its own block by a device copy (skipped when `send` already is its slot), the others by grouped receives
  if(rows_per_rank[root] > 0 && send != recv + off[root])
    HIP_TRY(hipMemcpyAsync(recv + off[root], send, sizeof(float) * (size_t)(rows_per_rank[root] * row_floats), hipMemcpyDeviceToDevice, st));
  if((rc = check(r.gstart(), "ncclGroupStart"))) return rc;
  for(int i = 0; i < world; i++)
  {
    if(i == root || rows_per_rank[i] == 0) continue;
    const int e = r.recv(recv + off[i], (size_t)(rows_per_rank[i] * row_floats), NCCL_FLOAT32, i, comm, st);
    if(e)
    {
      (void)r.gend();
      return check(e, "ncclRecv");
    }
  }
  return check(r.gend(), "ncclGroupEnd");
}

// Start-up check of the run-time binding for a host about to use the gather: one grouped ncclSend / ncclRecv pair of `rank` with
// ITSELF (RCCL completes a self-exchange inside the group as a device copy), through the very function pointers, datatype constant
// and group calls smplpp_gather_to_root uses.  A one-rank communicator never reaches those calls through the gather itself (it
// has no peer), so this is also how the send / receive leg is exercised on a one-GPU box (tests/test_gather_gpu.py).
extern "C" int smplpp_gather_selfcheck(void * comm, int rank, const float * send, float * recv, int64_t count_floats, void * stream)
{
  if(!comm || rank < 0 || !send || !recv || count_floats <= 0 || send == recv) return fail(SMPLPP_ERR_INVALID, "smplpp_gather_selfcheck: bad argument");
  Rccl & r = rccl();
  if(!r.send || !r.recv || !r.gstart || !r.gend)
    return fail(SMPLPP_ERR_HIP, "smplpp_gather_selfcheck: RCCL is not available (librccl.so could not be loaded)");
  hipStream_t st = static_cast<hipStream_t>(stream);
  auto check = [&](int e, const char * what) -> int {
    if(e == 0) return (int)SMPLPP_OK;
    return fail(SMPLPP_ERR_HIP, std::string("smplpp_gather_selfcheck: ") + what + " failed: " + (r.errstr ? r.errstr(e) : "RCCL error"));
  };
  int rc;
  if((rc = check(r.gstart(), "ncclGroupStart"))) return rc;
  const int es = r.send(send, (size_t)count_floats, NCCL_FLOAT32, rank, comm, st);
  const int er = es ? 0 : r.recv(recv, (size_t)count_floats, NCCL_FLOAT32, rank, comm, st);
  if(es || er)
  {
    (void)r.gend();
    return check(es ? es : er, es ? "ncclSend" : "ncclRecv");
  }
  return check(r.gend(), "ncclGroupEnd");
}

extern "C" int smplpp_gather(void * comm, const float * send, float * recv, const int64_t * rows_per_rank, int world, int rank,
                             int64_t row_floats, void * stream)
{
  if(!comm || !recv || !rows_per_rank || world <= 0 || rank < 0 || rank >= world || row_floats <= 0)
    return fail(SMPLPP_ERR_INVALID, "smplpp_gather: bad argument");
  Rccl & r = rccl();
  if(!r.allgather || !r.bcast || !r.gstart || !r.gend)
    return fail(SMPLPP_ERR_HIP, "smplpp_gather: RCCL is not available (librccl.so could not be loaded)");
  hipStream_t st = static_cast<hipStream_t>(stream);
  bool equal = true;
  int64_t total = 0;
  for(int i = 0; i < world; i++)
  {
    if(rows_per_rank[i] < 0) return fail(SMPLPP_ERR_INVALID, "smplpp_gather: negative row count");
    equal = equal && rows_per_rank[i] == rows_per_rank[0];
    total += rows_per_rank[i];
  }
  if(total == 0) return SMPLPP_OK;
  if(rows_per_rank[rank] > 0 && !send) return fail(SMPLPP_ERR_INVALID, "smplpp_gather: null send block");
  auto check = [&](int rc, const char * what) -> int {
    if(rc == 0) return (int)SMPLPP_OK;
    return fail(SMPLPP_ERR_HIP, std::string("smplpp_gather: ") + what + " failed: " + (r.errstr ? r.errstr(rc) : "RCCL error"));
  };
  if(equal) // (in place when `send` is the rank's own slot of `recv`, as RCCL defines it)
    return check(r.allgather(send, recv, (size_t)(rows_per_rank[0] * row_floats), NCCL_FLOAT32, comm, st), "ncclAllGather");
  // ragged shards (dist.shard_sizes: the first N % world ranks hold one row more): one broadcast per rank, fused in a group
  std::vector<int64_t> offs((size_t)world + 1);
  int rc = smplpp_gather_offsets(rows_per_rank, world, row_floats, offs.data());
  if(rc) return rc;
  if((rc = check(r.gstart(), "ncclGroupStart"))) return rc;
  for(int i = 0; i < world; i++)
  {
    float * slot = recv + offs[i];
    if(rows_per_rank[i] > 0)
    {
      const int e = r.bcast(i == rank ? (const void *)send : (const void *)slot, slot, (size_t)(rows_per_rank[i] * row_floats), NCCL_FLOAT32, i, comm, st);
      if(e)
      {
        (void)r.gend();
        return check(e, "ncclBroadcast");
      }
    }
  }
  return check(r.gend(), "ncclGroupEnd");
}
