// VALU issue-rate probe (gfx950): v_fma_f32, v_pk_fma_f32, v_dot2c_f32_bf16 -- wave-instructions per second per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned seed) {
  float a[16]; f2 p[8];
  const float x = (float)(threadIdx.x & 7) * 0.001f + 1.0f, y = 0.9999f;
  const unsigned ux = seed * (threadIdx.x + 1), uy = seed ^ 0x3f803f80u;
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = (float)i;
#pragma unroll
  for (int i = 0; i < 8; ++i) p[i] = (f2){(float)i, (float)i + 0.5f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
      } else if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"((f2){x, y}), "v"((f2){y, x}));
      } else if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a[i]) : "v"(ux), "v"(uy));
      } else if (MODE == 3) {        // f16 (high half of a packed word) x f32 + f32: no unpack instruction
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(ux), "v"(y));
      } else {                       // the unpack pair the bf16 tiles need today: v_lshlrev_b32 / v_and_b32
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(a[i]) : "v"(ux + i));
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += p[i][0] + p[i][1];
  if (s == 12345.678f) out[0] = s;
}
template <int MODE> static void run(const char* name, int per_iter) {
  float* o; hipMalloc(&o, 4);
  const int iters = 20000, blocks = 256 * 8;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, o, 100, 3u);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, o, iters, 3u);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double winstr = (double)blocks * 4 * iters * per_iter;
  printf("%-20s %.3f ms  %.1f G wave-instr/s chip  (%.2f per CU per ns; cycles/instr/SIMD at 2.4 GHz = %.2f)\n", name, ms, winstr / ms / 1e6,
         winstr / ms / 1e6 / 256, 2.4 * 4 * 256 / (winstr / ms / 1e6));
}
int main() { run<0>("v_fma_f32", 64); run<1>("v_pk_fma_f32", 32); run<2>("v_dot2c_f32_bf16", 64); run<3>("v_fma_mix_f32 (f16 hi)", 64); run<4>("v_and_b32", 64); return 0; }
