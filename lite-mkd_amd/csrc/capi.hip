// C-ABI plumbing shared by every entry point: last-error string, version, device probe.
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

extern "C" void lmkd_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* lmkd_last_error(void) { return g_err; }

extern "C" int lmkd_abi_version(void) { return 3; }      // 3 (round 3): ticket words in the BatchNorm / column-sum entry points, lmkd_dropout_mask_dev

// 0 when a gfx950 device is visible, negative otherwise (message in lmkd_last_error()).
extern "C" int lmkd_device_check(int device) {
  hipDeviceProp_t p;
  hipError_t e = hipGetDeviceProperties(&p, device);
  if (e != hipSuccess) {
    lmkd_set_error("lmkd_device_check: %s", hipGetErrorString(e));
    return LMKD_EHIP;
  }
  if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
    lmkd_set_error("lmkd_device_check: device %d is %s, this library is built for gfx950 only", device, p.gcnArchName);
    return LMKD_EINVAL;
  }
  return LMKD_OK;
}

// storage type of the trunk's activation / activation-gradient tensors in HBM: 0 = fp32 (default), 1 = bf16 (common.h)
int g_lmkd_act_bf16 = 0;
extern "C" int lmkd_set_activation_dtype(int mode) {
  LMKD_REQUIRE(mode == 0 || mode == 1, "lmkd_set_activation_dtype: 0 = fp32, 1 = bf16");
  g_lmkd_act_bf16 = mode;
  return LMKD_OK;
}
extern "C" int lmkd_get_activation_dtype(void) { return g_lmkd_act_bf16; }
