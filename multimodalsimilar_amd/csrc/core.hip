// Library-wide state of libmmsim_hip.so: last-error string, launch check, version / device queries.
#include "common.h"
#include <string.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

extern "C" void mmsim_set_error(const char* msg) {
  strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char* mmsim_last_error(void) { return g_err; }

int mmsim_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    char buf[384];
    snprintf(buf, sizeof(buf), "%s: launch failed: %s", what, hipGetErrorString(e));
    mmsim_set_error(buf);
    return MMSIM_ERR_LAUNCH;
  }
  return MMSIM_OK;
}

extern "C" int mmsim_version(void) { return 100; }

// Returns the number of visible HIP devices, or -1 with the error string set.
extern "C" int mmsim_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { mmsim_set_error(hipGetErrorString(e)); return -1; }
  return n;
}
