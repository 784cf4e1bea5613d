// sbq_core.hip -- status strings, launch checking and tuning knobs of libsbq.
#include <atomic>
#include <climits>
#include <cstdint>
#include <cstring>

#include "sbq_common.hpp"

#include <mutex>
#include <vector>

namespace sbq {
namespace {
thread_local char g_last_hip_error[128] = "";
// Tuning knobs are per calling THREAD (benchmarks / A-B runs set them around their own calls): a host with several
// threads cannot have one thread's experiment change the kernels another thread's calls dispatch.
// A thread that never set a knob sees the PROCESS default (sbq_set_tuning(knob | SBQ_TUNING_PROCESS, v)): autograd runs
// backward kernels on its own threads, which a setting made on the Python main thread would otherwise never reach.
constexpr int kKnobUnset = INT32_MIN;
thread_local int g_knobs[4] = {kKnobUnset, kKnobUnset, kKnobUnset, kKnobUnset};
std::atomic<int> g_knob_defaults[4] = {{-1}, {0}, {0}, {0}};

// ---- zero-contract workspaces ---------------------------------------------------------------------------------
// The whole-tensor selection engine and the GPTQ mat-vec keep arrival counters / histogram copies in caller memory
// that must be zero when a call starts and that every call leaves zero.  Handed a dirty region, or one that another
// stream's call is still using, the kernels would return a wrong rank or a wrong sum -- silently.  So the library
// remembers which regions it has seen (address values only: nothing is owned, dereferenced or freed here):
//   * a region seen for the first time is zeroed by the library, on the call's stream, in front of the launch --
//     callers no longer have to hipMemset it;
//   * a region is bound to the stream of its last call; a call on ANOTHER stream is accepted only when that stream
//     is idle (hipStreamQuery), else SBQ_ERR_BUSY -- "not shared by calls that can run concurrently" is checked, not
//     assumed;
//   * sbq_workspace_release() forgets a region (before freeing it, or after writing to it).
struct WsEntry {
  int dev;
  char* begin;
  size_t bytes;  // zeroed so far
  hipStream_t stream;
};
std::mutex g_ws_mutex;
std::vector<WsEntry> g_ws;
}  // namespace

int workspace_guard(void* region, size_t bytes, hipStream_t st) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  char* begin = static_cast<char*>(region);
  std::lock_guard<std::mutex> lock(g_ws_mutex);
  // (the entry this thread hit last is tried first: a decode loop calls with the same region every time)
  thread_local size_t last_hit = 0;
  const size_t n_ws = g_ws.size();
  for (size_t probe = 0; probe < n_ws; ++probe) {
    const size_t idx = probe == 0 ? (last_hit < n_ws ? last_hit : 0) : (probe <= last_hit && last_hit < n_ws ? probe - 1 : probe);
    WsEntry& e = g_ws[idx];
    if (e.dev != dev || e.begin != begin) continue;
    last_hit = idx;
    if (e.stream != st) {
      if (hipStreamQuery(e.stream) == hipErrorNotReady) return SBQ_ERR_BUSY;
      (void)hipGetLastError();  // (a stream that no longer exists is idle too)
      e.stream = st;
    }
    if (bytes > e.bytes) {
      if (hipMemsetAsync(begin + e.bytes, 0, bytes - e.bytes, st) != hipSuccess) return check_launch();
      e.bytes = bytes;
    }
    return SBQ_OK;
  }
  // regions overlapping the new one are stale (the memory was freed and handed out again)
  for (size_t i = 0; i < g_ws.size();) {
    WsEntry& e = g_ws[i];
    if (e.dev == dev && e.begin < begin + bytes && begin < e.begin + e.bytes) {
      g_ws[i] = g_ws.back();
      g_ws.pop_back();
    } else {
      ++i;
    }
  }
  if (hipMemsetAsync(begin, 0, bytes, st) != hipSuccess) return check_launch();
  g_ws.push_back(WsEntry{dev, begin, bytes, st});
  return SBQ_OK;
}

int check_launch() {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) return SBQ_OK;
  std::strncpy(g_last_hip_error, hipGetErrorName(e), sizeof(g_last_hip_error) - 1);
  g_last_hip_error[sizeof(g_last_hip_error) - 1] = 0;
  return SBQ_ERR_LAUNCH;
}

int knob(int which) {
  const int v = g_knobs[which & 3];
  return v != kKnobUnset ? v : g_knob_defaults[which & 3].load(std::memory_order_relaxed);
}

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
    case SBQ_ERR_BUSY: return "workspace is still in use by a call on another stream";
    default: return "unknown sbq status";
  }
}

const char* sbq_last_hip_error(void) { return sbq::g_last_hip_error; }

int sbq_set_tuning(int knob, int value) {
  const bool process = (knob & SBQ_TUNING_PROCESS) != 0;
  knob &= ~SBQ_TUNING_PROCESS;
  if (knob < 0 || knob > 3) return SBQ_ERR_ARG;
  if (process) sbq::g_knob_defaults[knob].store(value, std::memory_order_relaxed);
  else sbq::g_knobs[knob] = value;
  return SBQ_OK;
}

int sbq_workspace_release(const void* workspace, size_t workspace_bytes) {
  if (!workspace) return SBQ_ERR_NULL;
  int dev = 0;
  (void)hipGetDevice(&dev);
  const char* lo = static_cast<const char*>(workspace);
  const char* hi = lo + workspace_bytes;
  std::lock_guard<std::mutex> lock(sbq::g_ws_mutex);
  for (size_t i = 0; i < sbq::g_ws.size();) {
    const sbq::WsEntry& e = sbq::g_ws[i];
    if (e.dev == dev && e.begin >= lo && e.begin < hi) {
      sbq::g_ws[i] = sbq::g_ws.back();
      sbq::g_ws.pop_back();
    } else {
      ++i;
    }
  }
  return SBQ_OK;
}

}  // extern "C"
