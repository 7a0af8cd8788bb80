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

// Device of the calling thread (0..63), for per-device one-time setup such as hipFuncSetAttribute opt-ins.
int mmsim_current_device(void) {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d > 63) d = 0;
  return d;
}

// Deterministic mode (verification): every cross-workgroup sum is formed in a fixed order -- reductions of partial slabs by ONE
// workgroup per output without atomics, no split-K (one adder per output element), one z-slice in the pooling kernels, a
// serial embedding scatter.  Results are then bit-identical from run to run (tests/test_gpu_determinism.py); the default mode
// keeps the faster forms whose fp32 atomic adds arrive in varying order (last-bit differences that train-mode BatchNorm at
// random init amplifies: see DESIGN.md section 5).
static int g_deterministic = 0;
extern "C" int mmsim_set_deterministic(int on) { g_deterministic = on ? 1 : 0; return MMSIM_OK; }
extern "C" int mmsim_get_deterministic(void) { return g_deterministic; }
int mmsim_deterministic(void) { return g_deterministic; }

// Step-seed word for hipGraph replay.  Dropout seeds are kernel ARGUMENTS, which a captured graph freezes; when this pointer is
// set, every dropout kernel adds *ptr (read on the device at run time) to its seed argument, so a replayed step draws new masks
// after the host has bumped the word (multimodalsimilar_amd.train.GraphedTrainStep).  NULL (default): seeds are the arguments.
static const unsigned long long* g_step_seed_ptr = nullptr;
extern "C" int mmsim_set_step_seed_ptr(const unsigned long long* dev_ptr) { g_step_seed_ptr = dev_ptr; return MMSIM_OK; }
const unsigned long long* mmsim_step_seed_ptr(void) { return g_step_seed_ptr; }

// 300: the image tower's forward tensors are fp16 (the *_bf16-named image entry points read fp16 where the header says so) and
// mmsim_embed_ln_bwd2 / mmsim_adamw_step2 exist; a binding generated from an older header must refuse a library >= 300.
extern "C" int mmsim_version(void) { return 300; }

// Returns the number of visible HIP devices, or -1 with the error string set.
extern "C" int mmsim_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { mmsim_set_error(hipGetErrorString(e)); return -1; }
  return n;
}
