// Tracing equivalent of the reference's timing log: the reference pushes std::chrono spans into a durationList and prints them
// every ten seconds (/root/reference/node/node.cpp:752-781 "forward SMPL", :796-881 "calculate IK matrices", :907-943
// "solve IK", :974-988 "project point").  Here the same names are roctx ranges around the corresponding ENQUEUES (the kernels
// run asynchronously; rocprofv3 correlates them), so `rocprofv3 --marker-trace --kernel-trace -- <program>` reads like that
// log.  roctx is resolved at run time, like RCCL: the symbols already in the process (a profiler that loaded them), else —
// only when SMPLPP_ROCTX is set — librocprofiler-sdk-roctx.so / libroctx64.so.  No hard dependency, no cost when absent (one
// predictable branch per range).
#pragma once

#include <dlfcn.h>

#include <cstdlib>

namespace smplpp_hip
{
struct RoctxApi
{
  int (*push)(const char *) = nullptr;
  int (*pop)() = nullptr;
};
inline const RoctxApi & roctx_api()
{
  static const RoctxApi api = [] {
    RoctxApi a;
    void * h = RTLD_DEFAULT;
    if(!dlsym(h, "roctxRangePushA"))
    {
      const char * e = getenv("SMPLPP_ROCTX");
      if(!e || e[0] == '0') return a;
      h = nullptr;
      for(const char * name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so",
                               "libroctx64.so.4", "libroctx64.so", "/opt/rocm/lib/libroctx64.so"})
        if((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
      if(!h) return a;
    }
    a.push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
    a.pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
    if(!a.push || !a.pop) a.push = nullptr, a.pop = nullptr;
    return a;
  }();
  return api;
}
struct TraceRange
{
  bool on;
  explicit TraceRange(const char * name) : on(roctx_api().push != nullptr)
  {
    if(on) roctx_api().push(name);
  }
  ~TraceRange()
  {
    if(on) roctx_api().pop();
  }
  TraceRange(const TraceRange &) = delete;
  TraceRange & operator=(const TraceRange &) = delete;
};
} // namespace smplpp_hip
