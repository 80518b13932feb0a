// Error reporting and device queries of libpevae_hip.so.
#include "common.h"
#include "../../include/pevae_hip.h"
#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = "";

void pv_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" {

const char* pv_last_error(void) { return g_err; }

int pv_abi_version(void) { return PV_ABI_VERSION; }

int pv_device_info(int* cu_count, int* lds_bytes, char* arch, int arch_len) {
  int dev = 0;
  PV_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  PV_HIP(hipGetDeviceProperties(&prop, dev));
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (lds_bytes) *lds_bytes = (int)prop.maxSharedMemoryPerMultiProcessor;
  if (arch && arch_len > 0) {
    strncpy(arch, prop.gcnArchName, arch_len - 1);
    arch[arch_len - 1] = 0;
  }
  return PV_OK;
}

}  // extern "C"
