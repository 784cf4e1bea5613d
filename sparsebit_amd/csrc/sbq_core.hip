// sbq_core.hip -- status strings, launch checking and tuning knobs of libsbq.
#include <atomic>
#include <cstring>

#include "sbq_common.hpp"

namespace sbq {
namespace {
thread_local char g_last_hip_error[128] = "";
std::atomic<int> g_knobs[4] = {{-1}, {0}, {0}, {0}};
}  // namespace

int check_launch() {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) return SBQ_OK;
  std::strncpy(g_last_hip_error, hipGetErrorName(e), sizeof(g_last_hip_error) - 1);
  g_last_hip_error[sizeof(g_last_hip_error) - 1] = 0;
  return SBQ_ERR_LAUNCH;
}

int knob(int which) { return g_knobs[which & 3].load(std::memory_order_relaxed); }

// compute units of the current device (cached per device ordinal; 256 on an MI355X)
uint32_t cu_count() {
  constexpr int kMaxDev = 64;
  static std::atomic<uint32_t> cache[kMaxDev];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return 256;
  uint32_t n = cache[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    n = static_cast<uint32_t>(v);
    cache[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

}  // namespace sbq

extern "C" {

int sbq_version(void) { return SBQ_VERSION; }

const char* sbq_strerror(int status) {
  switch (status) {
    case SBQ_OK: return "ok";
    case SBQ_ERR_DTYPE: return "Kernel Failure, Invalid dtype of Input tensor";
    case SBQ_ERR_EMPTY: return "Kernel Failure, Tensor is empty";
    case SBQ_ERR_NULL: return "required pointer is NULL";
    case SBQ_ERR_ARG: return "inconsistent sizes or ranges";
    case SBQ_ERR_WORKSPACE: return "workspace too small or misaligned";
    case SBQ_ERR_LAUNCH: return "HIP kernel launch failed";
    case SBQ_ERR_ALIGN: return "pointer not aligned to its element size";
    default: return "unknown sbq status";
  }
}

const char* sbq_last_hip_error(void) { return sbq::g_last_hip_error; }

int sbq_set_tuning(int knob, int value) {
  if (knob < 0 || knob > 3) return SBQ_ERR_ARG;
  sbq::g_knobs[knob].store(value, std::memory_order_relaxed);
  return SBQ_OK;
}

}  // extern "C"
