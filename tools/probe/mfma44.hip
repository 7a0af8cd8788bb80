// Issue-rate probe (gfx950) for the 16-block 4x4x4 MFMA -- the form a DEPTHWISE convolution maps to (block = channel):
//   v_mfma_f32_4x4x4_16b_f16: 16 independent 4x4x4 products per wave-instruction = 1 024 MACs.
// Modes: 0 = 8 independent accumulators, 1 = one dependent accumulator chain, 2 = 8 accumulators with one ds_read_b64 per MFMA (operand
// stream from LDS), 3 = v_mfma_f32_16x16x16_f16 (8 independent accumulators) for reference, 4 = v_fma_f32 for reference.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[16384];
  for (int i = threadIdx.x; i < 4096; i += 256) reinterpret_cast<float*>(lds)[i] = 0.001f * i;
  __syncthreads();
  f4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (f4){(float)i, 0.f, 1.f, 2.f};
  h4 a = {(_Float16)0.5f, (_Float16)0.25f, (_Float16)1.0f, (_Float16)(0.001f * threadIdx.x)}, b = {(_Float16)1.0f, (_Float16)0.5f, (_Float16)0.125f, (_Float16)2.0f};
  float x = 1.0001f, y = 0.9999f, s0 = 0.f;
  const unsigned laddr = (threadIdx.x & 63) * 8;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
      } else if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(a), "v"(b));
      } else if (MODE == 2) {
        h4 bb[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(bb[i]) : "v"(laddr), "i"(i * 512));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(bb[i]));
      } else if (MODE == 3) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s0) : "v"(x), "v"(y));
      }
    }
  }
  float s = s0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[0] = s;
}
template <int MODE> static void run(const char* name, double macs_per_instr) {
  float* o; hipMalloc(&o, 4);
  const int iters = 20000, blocks = 256 * 4;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, o, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, o, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double winstr = (double)blocks * 4 * iters * 32;
  printf("%-44s %.3f ms  %.1f G wave-instr/s chip, %.1f T MAC/s; cycles/instr/SIMD at 2.4 GHz = %.2f (waves per SIMD: %d)\n", name, ms, winstr / ms / 1e6,
         winstr * macs_per_instr / ms / 1e9, 2.4e9 * 4 * 256 / (winstr / ms * 1e3), blocks * 4 / 1024);
}
int main() {
  run<0>("mfma 4x4x4_16b f16, 8 accumulators", 1024); run<1>("mfma 4x4x4_16b f16, dependent chain", 1024);
  run<2>("mfma 4x4x4_16b f16 + ds_read_b64 each", 1024); run<3>("mfma 16x16x16 f16, 8 accumulators", 4096); run<4>("v_fma_f32 dependent", 64);
  return 0;
}
