// C-ABI plumbing shared by every entry point: thread-local error text, version, device probe.
#include "common.h"

thread_local char g_sgl_mi355_err[512] = {0};

extern "C" const char* sgl_mi355_last_error(void) { return g_sgl_mi355_err; }

extern "C" int sgl_mi355_abi_version(void) { return 1; }

// Number of compute units of the current device (the reference's get_device_core_count,
// python/sglang/srt/utils.py, feeds the kv-split heuristic triton_backend.py:124-158).
extern "C" int sgl_mi355_device_cu_count(int device) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
    snprintf(g_sgl_mi355_err, sizeof(g_sgl_mi355_err), "hipGetDeviceProperties(%d) failed", device);
    return -1;
  }
  return prop.multiProcessorCount;
}
