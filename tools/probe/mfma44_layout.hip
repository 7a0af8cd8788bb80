// Operand / result lane maps of v_mfma_f32_4x4x4_16b_f16 (and _bf16) on gfx950, checked with exact small-integer data, plus: does an
// UNALIGNED ds_read_b64 (2-byte aligned address) return the right bytes on this system?
// Hypothesis: block = lane / 4;  A[i][k]: lane 4 blk + i, element k;  B[k][j]: lane 4 blk + j, element k;  D[i][j]: lane 4 blk + j, register i.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
__host__ __device__ inline float Aval(int blk, int i, int k) { return (float)((blk % 5) + 1) * (float)(1 + i) + 0.5f * (float)k; }
__host__ __device__ inline float Bval(int blk, int k, int j) { return (k == j ? 2.0f : 0.0f) + 0.25f * (float)(k * 4 + j) * ((blk & 1) ? 1.f : -1.f); }
__global__ void kern(float* out, float* out2, unsigned* out3) {
  const int lane = threadIdx.x, blk = lane >> 2, r = lane & 3;
  h4 a, b;
  s4 ab, bb;
  for (int e = 0; e < 4; ++e) {
    const float av = Aval(blk, r, e), bv = Bval(blk, e, r);
    a[e] = (_Float16)av; b[e] = (_Float16)bv;
    __bf16 t = (__bf16)av; ab[e] = __builtin_bit_cast(short, t); t = (__bf16)bv; bb[e] = __builtin_bit_cast(short, t);
  }
  f4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, acc, 0, 0, 0);
  acc2 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(ab, bb, acc2, 0, 0, 0);
  for (int i = 0; i < 4; ++i) { out[lane * 4 + i] = acc[i]; out2[lane * 4 + i] = acc2[i]; }
  __shared__ __attribute__((aligned(16))) unsigned short lds[512];
  for (int i = lane; i < 512; i += 64) lds[i] = (unsigned short)(i + 1000);
  __syncthreads();
  typedef unsigned short __attribute__((address_space(3))) * lptr;
  const unsigned addr = (unsigned)(uintptr_t)(lptr)(lds) + (unsigned)lane * 6u + 2u;       // halfs 3 lane + 1 .. + 4: only 2-byte aligned
  unsigned long long v;
  asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out3[lane * 2] = (unsigned)v; out3[lane * 2 + 1] = (unsigned)(v >> 32);
}
int main() {
  float *o, *o2; unsigned* o3;
  hipMalloc(&o, 256 * 4); hipMalloc(&o2, 256 * 4); hipMalloc(&o3, 128 * 4);
  hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, o, o2, o3);
  float h[256], h2[256]; unsigned h3[128];
  hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(h2, o2, sizeof(h2), hipMemcpyDeviceToHost); hipMemcpy(h3, o3, sizeof(h3), hipMemcpyDeviceToHost);
  int bad = 0, bad2 = 0;
  for (int lane = 0; lane < 64; ++lane)
    for (int i = 0; i < 4; ++i) {
      const int blk = lane >> 2, j = lane & 3;
      float ref = 0.f;
      for (int k = 0; k < 4; ++k) ref += Aval(blk, i, k) * Bval(blk, k, j);
      if (fabsf(h[lane * 4 + i] - ref) > 1e-3f * fabsf(ref) + 1e-3f) { if (bad < 6) printf("f16  lane %d reg %d: got %g want D[%d][%d] = %g\n", lane, i, h[lane * 4 + i], i, j, ref); ++bad; }
      if (fabsf(h2[lane * 4 + i] - ref) > 2e-2f * fabsf(ref) + 1e-2f) { if (bad2 < 6) printf("bf16 lane %d reg %d: got %g want %g\n", lane, i, h2[lane * 4 + i], ref); ++bad2; }
    }
  printf("4x4x4_16b f16 layout hypothesis: %s (%d mismatches); bf16: %s (%d)\n", bad ? "WRONG" : "ok", bad, bad2 ? "WRONG" : "ok", bad2);
  int badu = 0;
  for (int lane = 0; lane < 64; ++lane) {
    const unsigned base = 1000 + lane * 3 + 1;
    const unsigned want0 = base | ((base + 1) << 16), want1 = (base + 2) | ((base + 3) << 16);
    if (h3[lane * 2] != want0 || h3[lane * 2 + 1] != want1) { if (badu < 4) printf("unaligned ds_read_b64 lane %d: got %08x %08x want %08x %08x\n", lane, h3[lane * 2], h3[lane * 2 + 1], want0, want1); ++badu; }
  }
  printf("unaligned (2-byte) ds_read_b64: %s (%d mismatches)\n", badu ? "WRONG / unsupported" : "ok", badu);
  return 0;
}
