// EfficientNet (MBConv) kernels for gfx950, NHWC activations, fp32 statistics.
// Element types: every FORWARD tensor (conv outputs z, activations a, block outputs x) is fp16 (f16: 11-bit significand -- bf16
// storage alone cost 3.5-5 % on the image embedding, see common.h); every GRADIENT tensor (dy, dz, dx, resid) is bf16 (range).
// The image tower is HBM-bound: every kernel here maps a thread to one 8-channel octet
// (one 16-byte access) so a wave reads whole contiguous pixel rows, keeps its channel octet fixed so
// per-channel reductions (BatchNorm batch statistics, BN/SE backward sums, depthwise weight gradients)
// accumulate in registers, and leaves a block as ONE atomic per channel.
//   bn_stats / bn_finalize / bn_apply          train-mode BatchNorm2d (eps 1e-5, momentum 0.1) + SiLU
//   pool_bn_act                                mean_HW act(bn(z)) [* other]  (SE squeeze, global pool, dgate)
//   se_mlp_fwd / se_mlp_bwd / se_wgrad         squeeze-excite 1x1 convs (bias) + SiLU + sigmoid
//   bn_bwd_reduce / bn_bwd_apply               BN (+SiLU, +SE gate) backward
//   dwconv_fwd / dwconv_bwd_data / dwconv_bwd_weight   depthwise k3/k5, stride 1/2
//   stem_fwd / stem_wgrad                      3x3 s2 conv on the NCHW fp32 input image
//   bn1d_fwd / bn1d_bwd                        BatchNorm1d after the fc layer (cv_classifier.py:54)
// Pointwise (1x1) convs are GEMMs over [pixels, channels] (gemm.hip), optionally with the BN+SiLU+gate
// transform fused into the operand load.
#include "common.h"
#include <stdlib.h>

#ifndef DW_BD_PACKED
#define DW_BD_PACKED 0   // backward-data 3x3: 1 = all 3 x 6 chunks requested up front (packed), 0 = one kernel row at a time
#endif
// Exactly two waves per SIMD for the depthwise kernels: with a higher occupancy target hipcc's scheduler saves registers
// by re-using one destination for consecutive loads (load, wait, load, wait ...), which serialises a row's L2 round trips.
#define DW_OCC __attribute__((amdgpu_waves_per_eu(2, 2)))
struct CgMap { int G, nr, cg, rl; bool active; };
__device__ __forceinline__ CgMap cg_map(int C) {
  CgMap m;
  m.G = C >> 3;
  if (m.G >= 256) { m.nr = 1; m.rl = 0; m.cg = blockIdx.y * 256 + threadIdx.x; m.active = m.cg < m.G; }
  else { m.nr = 256 / m.G; m.cg = threadIdx.x % m.G; m.rl = threadIdx.x / m.G; m.active = m.rl < m.nr; }
  return m;
}
static inline int cg_grid_y(int C) { const int G = C >> 3; return G >= 256 ? (G + 255) / 256 : 1; }

// Map for the per-image pooling kernels: at most 32 octets per block, so wide layers get >= 8 row lanes per block (their
// planes are small: 14x14, 7x7) and the image's whole plane fits one block without a z-split and its atomics.
__device__ __forceinline__ CgMap pool_map(int C) {
  CgMap m;
  const int G = C >> 3;
  m.G = G >= 32 ? 32 : G;                       // octets per block (stride of the LDS reduction)
  m.nr = 256 / m.G;
  const int lc = threadIdx.x % m.G;
  m.rl = threadIdx.x / m.G;
  m.cg = blockIdx.y * m.G + lc;
  m.active = m.rl < m.nr && m.cg < G;
  return m;
}
static inline int pool_grid_y(int C) { const int G = C >> 3; return G >= 32 ? (G + 31) / 32 : 1; }
static inline int pool_nr(int C) { const int G = C >> 3; return 256 / (G >= 32 ? 32 : G); }

// Reduce acc[NV] over the row-lanes of a block (same channel octet) and STORE the block's partial result:
// value i goes to base[(i/8)*stat_stride + c0 + i%8], base = this block's slot of a [nparts][...] scratch.
// A second tiny kernel (reduce_partials) sums the slots: a thousand blocks atomically adding into the same few
// hundred addresses serialise at the memory side (~230 us per call measured) -- partial slabs do not.
template <int NV>
__device__ __forceinline__ void block_reduce_store(float (&acc)[NV], const CgMap& m, float* lds, float* base,
                                                   size_t stat_stride) {
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) lds[threadIdx.x * NV + i] = acc[i];
  __syncthreads();
  if (m.active && m.rl == 0) {
    for (int r = 1; r < m.nr; ++r)
#pragma unroll
      for (int i = 0; i < NV; ++i) acc[i] += lds[(m.cg + r * m.G) * NV + i];
#pragma unroll
    for (int i = 0; i < NV; ++i) base[(size_t)(i >> 3) * stat_stride + m.cg * 8 + (i & 7)] = acc[i];
  }
}

__device__ __forceinline__ void unpack8(const uint4& v, float (&f)[8]) {
  const bf8 b = __builtin_bit_cast(bf8, v);
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = bf2f(b[e]);
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  bf8 b;
#pragma unroll
  for (int e = 0; e < 8; ++e) b[e] = f2bf(f[e]);
  return __builtin_bit_cast(uint4, b);
}
// the same for fp16 chunks (forward tensors)
__device__ __forceinline__ void unpack8h(const uint4& v, float (&f)[8]) {
  const h8 b = __builtin_bit_cast(h8, v);
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = h2f(b[e]);
}
__device__ __forceinline__ uint4 pack8h(const float (&f)[8]) {
  h8 b;
#pragma unroll
  for (int e = 0; e < 8; ++e) b[e] = f2h(f[e]);
  return __builtin_bit_cast(uint4, b);
}
// Bounds-masked 16-byte load: the ADDRESS is always valid (callers clamp the coordinates) and the value is zeroed
// afterwards.  Writing `ok ? *p : 0` instead makes hipcc branch around every load and wait vmcnt(0) per element --
// the loads of a row then complete one L2 round trip after the other (cdna_hip_programming.md section 5, trap 4c).
__device__ __forceinline__ uint4 ld16_masked(const void* p, bool ok) {
  const uint4 v = *reinterpret_cast<const uint4*>(p);
  const unsigned int msk = ok ? 0xffffffffu : 0u;          // AND, not select: hipcc turns select-of-load back into a branch
  return make_uint4(v.x & msk, v.y & msk, v.z & msk, v.w & msk);
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }
// (u4v, PIPE_FIRST_USE, as_u4: common.h)
__device__ __forceinline__ void ld8f(const float* p, float (&f)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}

// out[i] += sum_p parts[p*n + i].  Block = 64 outputs x 4 waves; the part range is strided over (gridDim.y x 4) waves,
// reduced across the block's waves in LDS, and leaves as ONE atomic per output per block (gridDim.y <= 8 adders).
template <bool ATOMIC>
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ parts, int nparts, int n, float* out) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  float a0 = 0.f, a1 = 0.f;
  if (i < n) {
    const int step = gridDim.y * 4;
    int p = blockIdx.y * 4 + wv;
    for (; p + step < nparts; p += 2 * step) { a0 += parts[(size_t)p * n + i]; a1 += parts[(size_t)(p + step) * n + i]; }
    if (p < nparts) a0 += parts[(size_t)p * n + i];
  }
  red[wv][lane] = a0 + a1;
  __syncthreads();
  if (wv == 0 && i < n) {
    const float t = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    if (ATOMIC) atomicAdd(out + i, t);
    else out[i] += t;          // gridDim.y == 1: the only adder of this output, parts summed in a fixed order
  }
}
// Two slab reductions that share their row count in one launch (the depthwise backward's BatchNorm sums and tap-major weight
// gradient): blockIdx.x < gx_a works on job a, the rest on job b; otherwise reduce_partials_kernel.
template <bool ATOMIC>
__global__ __launch_bounds__(256) void reduce_partials2_kernel(const float* __restrict__ pa, int na, float* oa, const float* __restrict__ pb,
                                                               int nb, float* ob, int nparts, int gx_a) {
  __shared__ float red[4][64];
  const bool second = (int)blockIdx.x >= gx_a;
  const float* parts = second ? pb : pa;
  const int n = second ? nb : na;
  float* out = second ? ob : oa;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int i = ((int)blockIdx.x - (second ? gx_a : 0)) * 64 + lane;
  float a0 = 0.f, a1 = 0.f;
  if (i < n) {
    const int step = gridDim.y * 4;
    int p = blockIdx.y * 4 + wv;
    for (; p + step < nparts; p += 2 * step) { a0 += parts[(size_t)p * n + i]; a1 += parts[(size_t)(p + step) * n + i]; }
    if (p < nparts) a0 += parts[(size_t)p * n + i];
  }
  red[wv][lane] = a0 + a1;
  __syncthreads();
  if (wv == 0 && i < n) {
    const float t = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    if (ATOMIC) atomicAdd(out + i, t);
    else out[i] += t;
  }
}
void mmsim_launch_reduce2(const float* pa, int na, float* oa, const float* pb, int nb, float* ob, int nparts, hipStream_t s) {
  const int gx_a = (na + 63) / 64, gx_b = (nb + 63) / 64;
  int gy = nparts / 16; if (gy > 8) gy = 8; if (gy < 1) gy = 1;
  if (mmsim_deterministic())
    hipLaunchKernelGGL(reduce_partials2_kernel<false>, dim3(gx_a + gx_b, 1), dim3(256), 0, s, pa, na, oa, pb, nb, ob, nparts, gx_a);
  else
    hipLaunchKernelGGL(reduce_partials2_kernel<true>, dim3(gx_a + gx_b, gy), dim3(256), 0, s, pa, na, oa, pb, nb, ob, nparts, gx_a);
}

// ---- the reduction of a BatchNorm's statistics slab together with its finalisation, ONE launch, no atomics, no fences:
// a block owns 16 channels (their sum and sum-of-squares columns), its 16 row lanes walk the slab rows 8 loads at a time, an LDS tree
// finishes the sums in a fixed order, and the 16 channel threads write sums / mean / rstd / scale / shift / running statistics.
// (Letting the LAST workgroup of the atomic reduction finalise -- release fence + agent-scope ticket -- was built first and measured
// 1.3 ms/step SLOWER: each fence writes the XCD's dirty L2 lines back.  This form needs no cross-workgroup ordering at all.)
struct BnFinDev {
  const float* gamma; const float* beta; float* mean; float* rstd; float* scale; float* shift; float* run_mean; float* run_var;
  int C; float count, eps, momentum;
};
__global__ __launch_bounds__(256) void reduce_bn_finalize_kernel(const float* __restrict__ parts, int nparts, float* sums, BnFinDev f) {
  __shared__ float red[2][16][17];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  const bool ok = c < f.C;
  const size_t n = (size_t)2 * f.C;
  const float* p1 = parts + (ok ? c : 0);
  float a1 = 0.f, a2 = 0.f;
  for (int r0 = rl; r0 < nparts; r0 += 16 * 4) {        // four rows of each column pair in flight
    float v1[4], v2[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = min(r0 + 16 * q, nparts - 1);
      v1[q] = p1[(size_t)r * n]; v2[q] = p1[(size_t)r * n + f.C];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float m = (r0 + 16 * q < nparts) ? 1.f : 0.f;
      a1 += v1[q] * m; a2 += v2[q] * m;
    }
  }
  red[0][rl][cl] = a1; red[1][rl][cl] = a2;
  __syncthreads();
  if (rl == 0 && ok) {
    float s1 = sums[c], s2 = sums[f.C + c];            // accumulate contract of the slab reductions (pre-zeroed by the caller)
#pragma unroll
    for (int r = 0; r < 16; ++r) { s1 += red[0][r][cl]; s2 += red[1][r][cl]; }
    sums[c] = s1; sums[f.C + c] = s2;
    const float mu = s1 / f.count;
    const float var = fmaxf(s2 / f.count - mu * mu, 0.f);
    const float rs = rsqrtf(var + f.eps);
    f.mean[c] = mu; f.rstd[c] = rs;
    const float sc = f.gamma[c] * rs;
    f.scale[c] = sc; f.shift[c] = f.beta[c] - mu * sc;
    if (f.run_mean) {
      f.run_mean[c] = (1.f - f.momentum) * f.run_mean[c] + f.momentum * mu;
      f.run_var[c] = (1.f - f.momentum) * f.run_var[c] + f.momentum * var * (f.count / fmaxf(f.count - 1.f, 1.f));
    }
  }
}
// a BatchNorm finalisation waiting for the reduction of its statistics (per host thread; consumed by the next matching
// mmsim_launch_reduce, or launched on its own by mmsim_bn_finalize_flush)
struct BnFinArmed { bool armed = false; float* sums = nullptr; BnFinDev d; };
static thread_local BnFinArmed g_bnfin;
void mmsim_launch_reduce(const float* parts, int nparts, int n, float* out, int accumulate, hipStream_t s);
static void launch_reduce(const float* parts, int nparts, int n, float* out, int accumulate, hipStream_t s) {
  mmsim_launch_reduce(parts, nparts, n, out, accumulate, s);
}
void mmsim_launch_reduce(const float* parts, int nparts, int n, float* out, int accumulate, hipStream_t s) {
  if (!accumulate) (void)hipMemsetAsync(out, 0, (size_t)n * sizeof(float), s);
  if (g_bnfin.armed && g_bnfin.sums == out && n == 2 * g_bnfin.d.C && nparts <= 4096) {
    g_bnfin.armed = false;
    hipLaunchKernelGGL(reduce_bn_finalize_kernel, dim3((g_bnfin.d.C + 15) / 16), dim3(256), 0, s, parts, nparts, out, g_bnfin.d);
    return;
  }
  int gy = nparts / 16; if (gy > 8) gy = 8; if (gy < 1) gy = 1;
  if (mmsim_deterministic()) {
    hipLaunchKernelGGL(reduce_partials_kernel<false>, dim3((n + 63) / 64, 1), dim3(256), 0, s, parts, nparts, n, out);
    return;
  }
  hipLaunchKernelGGL(reduce_partials_kernel<true>, dim3((n + 63) / 64, gy), dim3(256), 0, s, parts, nparts, n, out);
}

// ------------------------------------------------------------------ BN statistics
__global__ __launch_bounds__(256) void bn_stats_kernel(const f16* __restrict__ z, float* parts, int P, int C, int rows_per_block) {
  __shared__ float lds[256 * 16];
  const CgMap m = cg_map(C);
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  if (m.active) {
    const int r0 = blockIdx.x * rows_per_block, r1 = min(P, r0 + rows_per_block);
#pragma unroll 4
    for (int r = r0 + m.rl; r < r1; r += m.nr) {
      float f[8];
      unpack8h(*reinterpret_cast<const uint4*>(z + (size_t)r * C + m.cg * 8), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) { acc[e] += f[e]; acc[8 + e] += f[e] * f[e]; }
    }
  }
  block_reduce_store<16>(acc, m, lds, parts + (size_t)blockIdx.x * 2 * C, (size_t)C);
}

// sums [2][C] -> mean, rstd, scale = gamma*rstd, shift = beta - mean*scale; running stats (momentum, unbiased var)
__global__ void bn_finalize_kernel(const float* sums, const float* gamma, const float* beta, float* mean, float* rstd,
                                   float* scale, float* shift, float* run_mean, float* run_var, int C, float count,
                                   float eps, float momentum) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float mu = sums[c] / count;
  const float var = fmaxf(sums[C + c] / count - mu * mu, 0.f);
  const float rs = rsqrtf(var + eps);
  mean[c] = mu; rstd[c] = rs;
  const float sc = gamma[c] * rs;
  scale[c] = sc; shift[c] = beta[c] - mu * sc;
  if (run_mean) {
    run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mu;
    run_var[c] = (1.f - momentum) * run_var[c] + momentum * var * (count / fmaxf(count - 1.f, 1.f));
  }
}

// out = act(scale*z + shift) (+ resid)
__global__ __launch_bounds__(256) void bn_apply_kernel(const f16* z, const float* scale, const float* shift,
                                                       const f16* resid, f16* out, size_t nchunks, int C, int act) {
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < nchunks; q += (size_t)gridDim.x * 256) {
    const int c0 = (int)((q * 8) % (size_t)C);
    float f[8], sc[8], sh[8];
    unpack8h(*reinterpret_cast<const uint4*>(z + q * 8), f);
    ld8f(scale + c0, sc); ld8f(shift + c0, sh);
#pragma unroll
    for (int e = 0; e < 8; ++e) { f[e] = f[e] * sc[e] + sh[e]; if (act) f[e] = silu_f(f[e]); }
    if (resid) {
      float r[8];
      unpack8h(*reinterpret_cast<const uint4*>(resid + q * 8), r);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] += r[e];
    }
    *reinterpret_cast<uint4*>(out + q * 8) = pack8h(f);
  }
}

// out[b,c] = mul * sum_hw act(scale*z+shift)[b,hw,c] * (other ? other[b,hw,c] : 1)      (fp32 [B,C])
template <bool OTHER>      // OTHER: `other` is given (compile time, so that the row loop's loads carry no branch)
__global__ __launch_bounds__(256) void pool_bn_act_kernel(const f16* __restrict__ z, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const bf16* __restrict__ other,
                                                          float* out, int HW, int C, int act, float mul, int rows_per_z,
                                                          f16* __restrict__ act_out) {
  __shared__ float lds[256 * 8];
  const CgMap m = pool_map(C);
  const int lc = threadIdx.x % m.G;
  const int b = blockIdx.x;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  if (m.active) {
    float sc[8], sh[8];
    ld8f(scale + m.cg * 8, sc); ld8f(shift + m.cg * 8, sh);
    const int rbeg = blockIdx.z * rows_per_z, rend = min(HW, rbeg + rows_per_z);
    // four rows per trip, all (bounds-masked) loads requested before the first use
    const f16* zb = z + (size_t)b * HW * C + m.cg * 8;
    const bf16* ob = OTHER ? other + (size_t)b * HW * C + m.cg * 8 : nullptr;      // a gradient tensor (bf16)
    // two row groups in flight, as in pool_bn_bwd_kernel (loads unmasked and unconditional)
    const int step = 4 * m.nr;
    auto load = [&](int r, u4v (&zr)[4], u4v (&orr)[4]) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const size_t off = (size_t)min(r + q * m.nr, rend - 1) * C;
        zr[q] = *reinterpret_cast<const u4v*>(zb + off);
        if (OTHER) orr[q] = *reinterpret_cast<const u4v*>(ob + off);
        else orr[q] = zr[q];
      }
    };
    auto comp = [&](int r, u4v (&zr)[4], u4v (&orr)[4]) __attribute__((always_inline)) {
      PIPE_FIRST_USE(zr, orr);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool ok = r + q * m.nr < rend;
        float f[8];
        unpack8h(as_u4(zr[q]), f);
        if (OTHER) {
          float o[8];
          unpack8(as_u4(orr[q]), o);
#pragma unroll
          for (int e = 0; e < 8; ++e) { float y = f[e] * sc[e] + sh[e]; if (act) y = silu_f(y); acc[e] += ok ? y * o[e] : 0.f; }
        } else {
          float y8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) { float y = f[e] * sc[e] + sh[e]; if (act) y = silu_f(y); y8[e] = y; acc[e] += ok ? y : 0.f; }
          // the activated tensor kept for the projection conv (its operand is then a2 * gate: one multiply per element in
          // the GEMM's staging instead of BN + SiLU + gate, which made those products VALU-bound)
          if (act_out && ok) *reinterpret_cast<uint4*>(act_out + ((size_t)b * HW + r + q * m.nr) * C + m.cg * 8) = pack8h(y8);
        }
      }
    };
    int r = rbeg + m.rl;
    if (r < rend) {
      u4v zA[4], oA[4], zB[4], oB[4];
      load(r, zA, oA);
      for (;;) {
        load(r + step, zB, oB);
        comp(r, zA, oA);
        r += step;
        if (r >= rend) break;
        load(r + step, zA, oA);
        comp(r, zB, oB);
        r += step;
        if (r >= rend) break;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 8; ++e) lds[threadIdx.x * 8 + e] = acc[e];
  __syncthreads();
  if (m.active && m.rl == 0) {
    for (int r = 1; r < m.nr; ++r)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += lds[(lc + r * m.G) * 8 + e];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (gridDim.z == 1) out[(size_t)b * C + m.cg * 8 + e] = acc[e] * mul;
      else atomicAdd(out + (size_t)b * C + m.cg * 8 + e, acc[e] * mul);
    }
  }
}

// ------------------------------------------------------------------ squeeze-excite MLP
// Both 1x1 convs of the SE branch are tiny (B x C x RD MACs) and latency-bound if one block walks an image serially.
// They are expressed with two generic, well-parallel kernels over weights stored [R][C] (C contiguous):
//   se_rowdot : out[b, r] = post( sum_c W[r][c] * pre(x[b, c]) + bias[r] )       (reduction over C: wave per (b, r))
//   se_colmix : out[b, c] = post( sum_r W[r][c] * pre(y[b, r]) + bias[c] )       (reduction over R: thread per (b, c))
// conv_reduce.weight is [RD][C] already; conv_expand.weight ([C][RD]) is transposed once per step into [RD][C].
enum { SE_PRE_NONE = 0, SE_PRE_SILU = 1, SE_PRE_DSIGMOID = 2 };     // DSIGMOID: x * g * (1 - g) with g = aux[b, c]
enum { SE_POST_NONE = 0, SE_POST_SIGMOID = 1, SE_POST_MUL_DSILU = 2 };   // MUL_DSILU: out * silu'(aux2[b, r])

__global__ __launch_bounds__(256) void se_rowdot_kernel(const float* __restrict__ W, const float* __restrict__ x,
                                                        const float* __restrict__ aux, const float* __restrict__ bias,
                                                        const float* __restrict__ aux2, float* out, float* out_silu, int B,
                                                        int C, int R, int pre, int post) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r = blockIdx.x;
  const float* w = W + (size_t)r * C;
#pragma unroll
  for (int k = 0; k < 2; ++k) {                       // 8 images per block, 2 per wave
    const int b = blockIdx.y * 8 + wv * 2 + k;
    if (b >= B) break;
    float a = 0.f;
    for (int c = lane * 4; c < C; c += 256) {          // C is a multiple of 8
      const float4 ww = *reinterpret_cast<const float4*>(w + c);
      float4 xx = *reinterpret_cast<const float4*>(x + (size_t)b * C + c);
      if (pre == SE_PRE_DSIGMOID) {
        const float4 g = *reinterpret_cast<const float4*>(aux + (size_t)b * C + c);
        xx.x *= g.x * (1.f - g.x); xx.y *= g.y * (1.f - g.y); xx.z *= g.z * (1.f - g.z); xx.w *= g.w * (1.f - g.w);
      }
      a += ww.x * xx.x + ww.y * xx.y + ww.z * xx.z + ww.w * xx.w;
    }
    a = wave_sum(a);
    if (lane == 0) {
      if (bias) a += bias[r];
      if (post == SE_POST_MUL_DSILU) a *= silu_grad_f(aux2[(size_t)b * R + r]);
      out[(size_t)b * R + r] = a;
      if (out_silu) out_silu[(size_t)b * R + r] = silu_f(a);
    }
  }
}

__global__ __launch_bounds__(256) void se_colmix_kernel(const float* __restrict__ W, const float* __restrict__ y,
                                                        const float* __restrict__ bias, float* out, int B, int C, int R,
                                                        int pre, int post) {
  __shared__ float ys[4][128];                        // R <= 128
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int b = blockIdx.y * 4 + wv;
  if (b < B)
    for (int r = lane; r < R; r += 64) {
      float v = y[(size_t)b * R + r];
      ys[wv][r] = pre == SE_PRE_SILU ? silu_f(v) : v;
    }
  __syncthreads();
  if (b >= B || c >= C) return;
  float a0 = bias ? bias[c] : 0.f, a1 = 0.f;
  int r = 0;
  for (; r + 1 < R; r += 2) { a0 += W[(size_t)r * C + c] * ys[wv][r]; a1 += W[(size_t)(r + 1) * C + c] * ys[wv][r + 1]; }
  if (r < R) a0 += W[(size_t)r * C + c] * ys[wv][r];
  float a = a0 + a1;
  if (post == SE_POST_SIGMOID) a = sigmoid_f(a);
  out[(size_t)b * C + c] = a;
}

// [C][R] <-> [R][C] (weights: overwrite; gradients: accumulate back)
__global__ void se_transpose_kernel(const float* in, float* out, int rows, int cols, int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  const int r = i / cols, c = i % cols;
  if (accumulate) out[(size_t)c * rows + r] += in[i];
  else out[(size_t)c * rows + r] = in[i];
}

// weight gradients (reductions over the batch).  One wave per (64 channels, 16 reduce-rows, quarter of the batch):
// the per-image vectors dpe[b,c], s[b,c] are loaded once per image and reused for 16 rows; hs / dr values are
// wave-uniform (scalar operands).  Partial sums leave as atomics (4 adders per address).
//   dWr[r][c] += sum_b dr[b,r] * s[b,c]      dWeT[r][c] += sum_b dpe[b,c] * hs[b,r]   (hs = silu(hr), dpe = dgate*g*(1-g))
//   dbr[r] += sum_b dr[b,r]                  dbe[c] += sum_b dpe[b,c]
__global__ __launch_bounds__(64) void se_wgrad_kernel(const float* __restrict__ dgate, const float* __restrict__ gate,
                                                      const float* __restrict__ dr, const float* __restrict__ hs,
                                                      const float* __restrict__ s, float* dWr, float* dbr, float* dWeT, float* dbe,
                                                      int B, int C, int R) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  const int r0 = blockIdx.y * 16;
  const int bq = (B + (int)gridDim.z - 1) / (int)gridDim.z, b0 = blockIdx.z * bq, b1 = min(B, b0 + bq);
  const bool cok = c < C;
  const int cc = min(c, C - 1);
  __shared__ float hsl[64][16], drl[64][16];      // this block's 16 hidden units of hs / dr for 64 batch rows (zero padded)
  float ae[16], ar[16], ab = 0.f, abr = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) { ae[k] = 0.f; ar[k] = 0.f; }
  for (int bc = b0; bc < b1; bc += 64) {
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 16; i += 64) {
      const int bb = i >> 4, k = i & 15;
      const bool ok = bc + bb < b1 && r0 + k < R;
      const size_t idx = (size_t)min(bc + bb, b1 - 1) * R + min(r0 + k, R - 1);
      const float m = ok ? 1.f : 0.f;
      hsl[bb][k] = hs[idx] * m;
      drl[bb][k] = dr[idx] * m;
    }
    __syncthreads();
    const int nb = min(64, b1 - bc);
    for (int bb = 0; bb < nb; bb += 4) {          // four batch rows per trip: 12 loads in flight
      float d[4], sv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const size_t off = (size_t)min(bc + bb + q, b1 - 1) * C + cc;
        const float m = (cok && bb + q < nb) ? 1.f : 0.f;
        const float g = gate[off];
        d[q] = dgate[off] * g * (1.f - g) * m;
        sv[q] = s[off] * m;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int bi = min(bb + q, 63);
        ab += d[q];
#pragma unroll
        for (int k = 0; k < 16; ++k) { ae[k] += d[q] * hsl[bi][k]; ar[k] += drl[bi][k] * sv[q]; }
      }
    }
    if (blockIdx.x == 0 && threadIdx.x < 16)
      for (int bb = 0; bb < nb; ++bb) abr += drl[bb][threadIdx.x];
  }
  if (cok) {
#pragma unroll
    for (int k = 0; k < 16; ++k)
      if (r0 + k < R) { atomicAdd(dWeT + (size_t)(r0 + k) * C + c, ae[k]); atomicAdd(dWr + (size_t)(r0 + k) * C + c, ar[k]); }
    if (blockIdx.y == 0) atomicAdd(dbe + c, ab);
  }
  if (blockIdx.x == 0 && threadIdx.x < 16 && r0 + threadIdx.x < R) atomicAdd(dbr + r0 + threadIdx.x, abr);
}

// ------------------------------------------------------------------ BN (+SiLU, +SE gate) backward
// Backward of the squeeze (SE) branch AND the batch sums of the depthwise BatchNorm backward in ONE pass over (z, dy):
// with u = scale z + shift, a = silu(u), a' = silu'(u), zh = (z - mean) rstd and dy the gradient w.r.t. the gated activation,
//   out5[0][b,c] = sum_hw a dy          (dgate, the SE backward input)
//   out5[1..4]   = sum_hw dy a',  sum_hw a',  sum_hw dy a' zh,  sum_hw a' zh
// The BN-backward sums of da = (dy gate + dsq / HW) a' are then  S1[c] = sum_b gate P1 + dsq/HW P2,  S2[c] = sum_b gate P3 +
// dsq/HW P4 (bn_bwd_sums_from_pool_kernel) -- dsq only exists after the SE backward, which needs dgate, so without this
// regrouping the reduction costs a second full pass over both tensors (bn_bwd_reduce_kernel).
__global__ __launch_bounds__(256) void pool_bn_bwd_kernel(const f16* __restrict__ z, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const bf16* __restrict__ dy,
                                                          float* out5, int B, int HW, int C, int rows_per_z) {
  __shared__ float lds[256 * 8];
  const CgMap m = pool_map(C);
  const int lc = threadIdx.x % m.G;
  const int b = blockIdx.x;
  float acc[5][8];
#pragma unroll
  for (int k = 0; k < 5; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[k][e] = 0.f;
  if (m.active) {
    float sc[8], sh[8], mu[8], rs[8];
    ld8f(scale + m.cg * 8, sc); ld8f(shift + m.cg * 8, sh); ld8f(mean + m.cg * 8, mu); ld8f(rstd + m.cg * 8, rs);
    const int rbeg = blockIdx.z * rows_per_z, rend = min(HW, rbeg + rows_per_z);
    const f16* zb = z + (size_t)b * HW * C + m.cg * 8;
    const bf16* db = dy + (size_t)b * HW * C + m.cg * 8;
    // Two row groups in flight (register sets A, B; the loop is unrolled by two): the next group's eight loads are requested before
    // the current group's arithmetic (~200 VALU instructions per row) instead of after it.  Loads are unmasked and unconditional --
    // rows past the end re-read the last row, their dy is zeroed where it is used -- so that every wait hipcc places is a counted one
    // in front of the arithmetic that needs it (an AND with the load, or a load behind `if`, puts a full wait next to the load).
    const int step = 4 * m.nr;
    auto load = [&](int r, u4v (&zr)[4], u4v (&dr)[4]) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const size_t off = (size_t)min(r + q * m.nr, rend - 1) * C;
        zr[q] = *reinterpret_cast<const u4v*>(zb + off);
        dr[q] = *reinterpret_cast<const u4v*>(db + off);
      }
    };
    auto comp = [&](int r, u4v (&zr)[4], u4v (&dr)[4]) __attribute__((always_inline)) {
      PIPE_FIRST_USE(zr, dr);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float ok = (r + q * m.nr < rend) ? 1.f : 0.f;     // a' of a row past the end is not zero
        float f[8], d[8];
        unpack8h(as_u4(zr[q]), f); unpack8(as_u4(dr[q]), d);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float u = f[e] * sc[e] + sh[e], sg = sigmoid_f(u);
          const float dd = d[e] * ok;
          const float a = u * sg, da = sg * (1.0f + u * (1.0f - sg)) * ok, zh = (f[e] - mu[e]) * rs[e];
          acc[0][e] += a * dd;
          acc[1][e] += dd * da; acc[2][e] += da;
          acc[3][e] += dd * da * zh; acc[4][e] += da * zh;
        }
      }
    };
    int r = rbeg + m.rl;
    if (r < rend) {
      u4v zA[4], dA[4], zB[4], dB[4];
      load(r, zA, dA);
      for (;;) {
        load(r + step, zB, dB);
        comp(r, zA, dA);
        r += step;
        if (r >= rend) break;
        load(r + step, zA, dA);
        comp(r, zB, dB);
        r += step;
        if (r >= rend) break;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) lds[threadIdx.x * 8 + e] = acc[k][e];
    __syncthreads();
    if (m.active && m.rl == 0) {
      float t[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] = acc[k][e];
      for (int r = 1; r < m.nr; ++r)
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] += lds[(lc + r * m.G) * 8 + e];
      float* o = out5 + ((size_t)k * B + b) * C + m.cg * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (gridDim.z == 1) o[e] = t[e];
        else atomicAdd(o + e, t[e]);
      }
    }
  }
}

// sums[0][c] = sum_b gate P1 + dsq/HW P2,  sums[1][c] = sum_b gate P3 + dsq/HW P4;  block = 64 channels x 4 batch lanes
__global__ __launch_bounds__(256) void bn_bwd_sums_from_pool_kernel(const float* __restrict__ out5, const float* __restrict__ gate,
                                                                    const float* __restrict__ dsq, float* sums, int B, int C, float inv_hw) {
  __shared__ float red[2][4][64];
  const int lane = threadIdx.x & 63, bl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float s1 = 0.f, s2 = 0.f;
  if (c < C) {
    const size_t BC = (size_t)B * C;
    const int bq = (B + gridDim.y - 1) / gridDim.y, b0 = blockIdx.y * bq, b1 = min(B, b0 + bq);
    for (int b = b0 + bl; b < b1; b += 4) {
      const size_t i = (size_t)b * C + c;
      const float g = gate[i], q = dsq[i] * inv_hw;
      s1 += g * out5[BC + i] + q * out5[2 * BC + i];
      s2 += g * out5[3 * BC + i] + q * out5[4 * BC + i];
    }
  }
  red[0][bl][lane] = s1; red[1][bl][lane] = s2;
  __syncthreads();
  if (bl == 0 && c < C) {           // sums are pre-zeroed by the caller (as for every BN-sum producer); gridDim.y adders per address
    atomicAdd(sums + c, (red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane]));
    atomicAdd(sums + C + c, (red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane]));
  }
}

struct BnBwd {
  const bf16* dy; const f16* z; const float* mean; const float* rstd; const float* scale; const float* shift;
  const float* gate; const float* dsq; int hw; int act; int P; int C; float inv_hw; FastDiv d_hw;
};
// da = (gate ? dy*g + dsq/HW : dy) * (act ? silu'(scale*z+shift) : 1)
// The per-channel vectors are loaded ONCE per thread (BnBwdCh) and the row loops below request FOUR rows' dy | z chunks before
// the first use: with one row per trip (load, wait, compute, store) a block's 30-40 trips ran one L2 round trip after the other
// and a 7-MB tensor took as long as a 70-MB one (37 us per launch whatever the size).
struct BnBwdCh { float mu[8], rs[8], sc[8], sh[8]; };
__device__ __forceinline__ void bn_bwd_ch(const BnBwd& p, int c0, BnBwdCh& ch) {
  ld8f(p.mean + c0, ch.mu); ld8f(p.rstd + c0, ch.rs);
  if (p.act) { ld8f(p.scale + c0, ch.sc); ld8f(p.shift + c0, ch.sh); }
}
__device__ __forceinline__ void bn_bwd_elem(const BnBwd& p, const BnBwdCh& ch, int r, int c0, const uint4& dv, const uint4& zv,
                                            float (&da)[8], float (&zh)[8]) {
  float d[8], z[8];
  unpack8(dv, d);
  unpack8h(zv, z);
  if (p.gate) {
    const int b = (int)fdiv((unsigned int)r, p.d_hw);
    float g[8], q[8];
    ld8f(p.gate + (size_t)b * p.C + c0, g); ld8f(p.dsq + (size_t)b * p.C + c0, q);
#pragma unroll
    for (int e = 0; e < 8; ++e) d[e] = d[e] * g[e] + q[e] * p.inv_hw;
  }
  if (p.act) {
#pragma unroll
    for (int e = 0; e < 8; ++e) d[e] *= silu_grad_f(z[e] * ch.sc[e] + ch.sh[e]);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { da[e] = d[e]; zh[e] = (z[e] - ch.mu[e]) * ch.rs[e]; }
}

__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(BnBwd p, float* parts, int rows_per_block) {
  __shared__ float lds[256 * 16];
  const CgMap m = cg_map(p.C);
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  if (m.active) {
    const int c0 = m.cg * 8;
    BnBwdCh ch;
    bn_bwd_ch(p, c0, ch);
    const int r0 = blockIdx.x * rows_per_block, r1 = min(p.P, r0 + rows_per_block);
    // two row groups in flight, as in pool_bn_bwd_kernel
    const int step = 4 * m.nr;
    auto load = [&](int r, u4v (&dv)[4], u4v (&zv)[4]) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const size_t off = (size_t)min(r + q * m.nr, r1 - 1) * p.C + c0;
        dv[q] = *reinterpret_cast<const u4v*>(p.dy + off);
        zv[q] = *reinterpret_cast<const u4v*>(p.z + off);
      }
    };
    auto comp = [&](int r, u4v (&dv)[4], u4v (&zv)[4]) __attribute__((always_inline)) {
      PIPE_FIRST_USE(dv, zv);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rq = r + q * m.nr;
        const float ok = rq < r1 ? 1.f : 0.f;
        float da[8], zh[8];
        bn_bwd_elem(p, ch, min(rq, r1 - 1), c0, as_u4(dv[q]), as_u4(zv[q]), da, zh);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float v = da[e] * ok; acc[e] += v; acc[8 + e] += v * zh[e]; }
      }
    };
    int r = r0 + m.rl;
    if (r < r1) {
      u4v dA[4], zA[4], dB[4], zB[4];
      load(r, dA, zA);
      for (;;) {
        load(r + step, dB, zB);
        comp(r, dA, zA);
        r += step;
        if (r >= r1) break;
        load(r + step, dA, zA);
        comp(r, dB, zB);
        r += step;
        if (r >= r1) break;
      }
    }
  }
  block_reduce_store<16>(acc, m, lds, parts + (size_t)blockIdx.x * 2 * p.C, (size_t)p.C);
}

// dz = scale * (da - S1/P - zh*S2/P); block 0 also does dgamma += S2, dbeta += S1
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(BnBwd p, const float* sums, bf16* dz, float* dgamma, float* dbeta,
                                                           int rows_per_block) {
  const CgMap m = cg_map(p.C);
  if (blockIdx.x == 0 && dgamma) {
    for (int c = blockIdx.y * 256 + threadIdx.x; c < p.C; c += 256 * gridDim.y) { dgamma[c] += sums[p.C + c]; dbeta[c] += sums[c]; }
  }
  if (!m.active) return;
  const int c0 = m.cg * 8;
  float s1[8], s2[8], sc[8];
  ld8f(sums + c0, s1); ld8f(sums + p.C + c0, s2); ld8f(p.scale + c0, sc);
  BnBwdCh ch;
  bn_bwd_ch(p, c0, ch);
  const float invP = 1.0f / (float)p.P;
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] *= invP; s2[e] *= invP; }
  const int r0 = blockIdx.x * rows_per_block, r1 = min(p.P, r0 + rows_per_block);
  // two row groups in flight, as in pool_bn_bwd_kernel
  const int step = 4 * m.nr;
  auto load = [&](int r, u4v (&dv)[4], u4v (&zv)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const size_t off = (size_t)min(r + q * m.nr, r1 - 1) * p.C + c0;
      dv[q] = *reinterpret_cast<const u4v*>(p.dy + off);
      zv[q] = *reinterpret_cast<const u4v*>(p.z + off);
    }
  };
  auto comp = [&](int r, u4v (&dv)[4], u4v (&zv)[4]) __attribute__((always_inline)) {
    PIPE_FIRST_USE(dv, zv);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int rq = r + q * m.nr;
      if (rq < r1) {
        float da[8], zh[8], o[8];
        bn_bwd_elem(p, ch, rq, c0, as_u4(dv[q]), as_u4(zv[q]), da, zh);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = sc[e] * (da[e] - s1[e] - zh[e] * s2[e]);
        *reinterpret_cast<uint4*>(dz + (size_t)rq * p.C + c0) = pack8(o);
      }
    }
  };
  int r = r0 + m.rl;
  if (r < r1) {
    u4v dA[4], zA[4], dB[4], zB[4];
    load(r, dA, zA);
    for (;;) {
      load(r + step, dB, zB);
      comp(r, dA, zA);
      r += step;
      if (r >= r1) break;
      load(r + step, dA, zA);
      comp(r, dB, zB);
      r += step;
      if (r >= r1) break;
    }
  }
}

// ------------------------------------------------------------------ depthwise conv
// weights are used in tap-major layout wT [K*K][C] (fp32) so a thread reads its 8 channels contiguously
__global__ void dw_weight_to_tap_major_kernel(const float* w, float* wT, int C, int KK) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < C * KK) { const int c = i / KK, t = i % KK; wT[(size_t)t * C + c] = w[i]; }
}
__global__ void dw_grad_from_tap_major_kernel(const float* gT, float* g, int C, int KK) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < C * KK) { const int c = i / KK, t = i % KK; g[i] += gT[(size_t)t * C + c]; }
}

// All depthwise layers of a network in ONE launch each way: segs[s] = {src offset, dst offset, C, K*K} (offsets in floats from the two
// base pointers; the flat parameter / gradient buffers and one contiguous tap-major buffer).  blockIdx.y = segment.
//   to_tap_major = 1:  dst[t*C + c]  = src[c*KK + t]      (weights [C][KK] -> wT [KK][C], once per step)
//   to_tap_major = 0:  dst[c*KK + t] += src[t*C + c]      (tap-major gradient -> the parameter's gradient, once per backward)
__global__ void dw_tap_major_batch_kernel(const long long* __restrict__ segs, const float* __restrict__ src_base, float* dst_base,
                                          int to_tap_major) {
  const long long so = segs[blockIdx.y * 4 + 0], dof = segs[blockIdx.y * 4 + 1];
  const int C = (int)segs[blockIdx.y * 4 + 2], KK = (int)segs[blockIdx.y * 4 + 3];
  const int n = C * KK;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int c = i / KK, t = i - c * KK;
    if (to_tap_major) dst_base[dof + (size_t)t * C + c] = src_base[so + i];
    else dst_base[dof + i] += src_base[so + (size_t)t * C + c];
  }
}

struct DwGeom { int B, Hi, Wi, Ho, Wo, C; FastDiv d_strip, d_rows; };   // d_strip / d_rows: strips per row, rows per image of the item space

// forward: a [B,Hi,Wi,C] -> z [B,Ho,Wo,C]; fused per-channel sum / sumsq of z (bf16-rounded) for the next BN.
// thread = (octet, strip of TW output pixels along W)
// V = 0: one kernel row at a time (few registers);  V = 1: all K x NIN input chunks of an item are requested up front and
// kept PACKED (bf16) until used, so a thread has K x NIN 16-byte loads in flight instead of NIN (latency-bound otherwise)
template <int K, int S, int TW, int V>
__global__ __launch_bounds__(256) DW_OCC void dwconv_fwd_kernel(const f16* __restrict__ a, const float* __restrict__ wT, f16* z, float* parts, DwGeom g,
                                                         int items_per_block) {
  constexpr int PAD = K / 2, NIN = (TW - 1) * S + K;
  __shared__ float lds[256 * 16];
  const CgMap m = cg_map(g.C);
  float st[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) st[i] = 0.f;
  const int nstrip = (g.Wo + TW - 1) / TW;
  const int nitems = g.B * g.Ho * nstrip;
  if (m.active) {
    const int i0 = blockIdx.x * items_per_block, i1 = min(nitems, i0 + items_per_block);
    for (int it = i0 + m.rl; it < i1; it += m.nr) {
      int strip, ho, b, rowi;
      fdivmod((unsigned int)it, g.d_strip, rowi, strip);
      fdivmod((unsigned int)rowi, g.d_rows, b, ho);
      const int wo0 = strip * TW;
      float acc[TW][8];
#pragma unroll
      for (int j = 0; j < TW; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[j][e] = 0.f;
      if constexpr (V == 1) {
        uint4 pk[K][NIN];
#pragma unroll
        for (int kh = 0; kh < K; ++kh) {
          const int hi = ho * S - PAD + kh;
#pragma unroll
          for (int x = 0; x < NIN; ++x) {
            const int wi = wo0 * S - PAD + x;
            pk[kh][x] = ld16_masked(a + (((size_t)b * g.Hi + clampi(hi, 0, g.Hi - 1)) * g.Wi + clampi(wi, 0, g.Wi - 1)) * g.C + m.cg * 8,
                                    hi >= 0 && hi < g.Hi && wi >= 0 && wi < g.Wi);
          }
        }
#pragma unroll
        for (int kh = 0; kh < K; ++kh) {
#pragma unroll
          for (int kw = 0; kw < K; ++kw) {
            float w[8];
            ld8f(wT + (size_t)(kh * K + kw) * g.C + m.cg * 8, w);
#pragma unroll
            for (int j = 0; j < TW; ++j) {
              float in[8];
              unpack8h(pk[kh][j * S + kw], in);
#pragma unroll
              for (int e = 0; e < 8; ++e) acc[j][e] += in[e] * w[e];
            }
          }
        }
      } else {
        // one kernel row per trip, the NEXT row's chunks requested before this row's FMAs (a trip used to be: request,
        // wait a full L2 round trip, compute -- K dependent round trips per item at two waves per SIMD)
        auto load_row = [&](int kh, uint4 (&dst)[NIN]) {
          const int hi = ho * S - PAD + kh;
          const bool hv = hi >= 0 && hi < g.Hi;
          const f16* arow = a + ((size_t)b * g.Hi + clampi(hi, 0, g.Hi - 1)) * g.Wi * g.C + m.cg * 8;
#pragma unroll
          for (int x = 0; x < NIN; ++x) {
            const int wi = wo0 * S - PAD + x;
            dst[x] = ld16_masked(arow + (size_t)clampi(wi, 0, g.Wi - 1) * g.C, hv && wi >= 0 && wi < g.Wi);
          }
        };
        uint4 raw[NIN], nxt[NIN];
        load_row(0, raw);
#pragma unroll 1
        for (int kh = 0; kh < K; ++kh) {
          if (kh + 1 < K) load_row(kh + 1, nxt);
          float in[NIN][8];
#pragma unroll
          for (int x = 0; x < NIN; ++x) unpack8h(raw[x], in[x]);
#pragma unroll
          for (int kw = 0; kw < K; ++kw) {
            float w[8];
            ld8f(wT + (size_t)(kh * K + kw) * g.C + m.cg * 8, w);
#pragma unroll
            for (int j = 0; j < TW; ++j)
#pragma unroll
              for (int e = 0; e < 8; ++e) acc[j][e] += in[j * S + kw][e] * w[e];
          }
#pragma unroll
          for (int x = 0; x < NIN; ++x) raw[x] = nxt[x];
        }
      }
#pragma unroll
      for (int j = 0; j < TW; ++j) {
        if (wo0 + j < g.Wo) {
          const uint4 o = pack8h(acc[j]);
          *reinterpret_cast<uint4*>(z + (((size_t)b * g.Ho + ho) * g.Wo + wo0 + j) * g.C + m.cg * 8) = o;
          float r[8];
          unpack8h(o, r);
#pragma unroll
          for (int e = 0; e < 8; ++e) { st[e] += r[e]; st[8 + e] += r[e] * r[e]; }
        }
      }
    }
  }
  block_reduce_store<16>(st, m, lds, parts + (size_t)blockIdx.x * 2 * g.C, (size_t)g.C);
}

// backward data: da[b,hi,wi,c] = sum_{kh,kw} dz[b,(hi+PAD-kh)/S,(wi+PAD-kw)/S,c] * w[kh,kw,c]   (divisible taps only)
// fused: dpre = da * silu'(scale*z1+shift) written to `out`, and BN-backward sums (sum dpre, sum dpre*zhat) of the
// producer's BatchNorm accumulated per channel.  thread = (octet, strip of TW input pixels along W)
template <int K, int S>
__global__ __launch_bounds__(256) DW_OCC void dwconv_bwd_data_kernel(const bf16* dz, const float* wT, const f16* z1, const float* mean,
                                                              const float* rstd, const float* scale, const float* shift,
                                                              const bf16* resid, bf16* out, float* parts, DwGeom g,
                                                              int items_per_block) {
  constexpr int TW = 4, PAD = K / 2;
  __shared__ float lds[256 * 16];
  const CgMap m = cg_map(g.C);
  float st[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) st[i] = 0.f;
  const int nstrip = (g.Wi + TW - 1) / TW;
  const int nitems = g.B * g.Hi * nstrip;
  if (m.active) {
    const int c0 = m.cg * 8;
    float mu[8], rs[8], sc[8], sh[8];
    if (z1) { ld8f(mean + c0, mu); ld8f(rstd + c0, rs); ld8f(scale + c0, sc); ld8f(shift + c0, sh); }
    const int i0 = blockIdx.x * items_per_block, i1 = min(nitems, i0 + items_per_block);
    for (int it = i0 + m.rl; it < i1; it += m.nr) {
      int strip, hi, b, rowi;
      fdivmod((unsigned int)it, g.d_strip, rowi, strip);
      fdivmod((unsigned int)rowi, g.d_rows, b, hi);
      const int wi0 = strip * TW;
      float acc[TW][8];
#pragma unroll
      for (int j = 0; j < TW; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[j][e] = 0.f;
      if constexpr (S == 1) {
        // stride 1: a correlation with the flipped kernel.  One dz row segment (TW + K - 1 chunks) is loaded per kernel
        // row and reused by all K taps of all TW outputs from registers.
        constexpr int NIN = TW + K - 1;
        if constexpr (K == 3 && DW_BD_PACKED) {
          // all K x NIN chunks requested up front and kept packed: K*NIN loads in flight per thread (see dwconv_fwd V = 1)
          uint4 pk[K][NIN];
#pragma unroll
          for (int kh = 0; kh < K; ++kh) {
            const int ho = hi + PAD - kh;
#pragma unroll
            for (int x = 0; x < NIN; ++x) {
              const int wo = wi0 - PAD + x;
              pk[kh][x] = ld16_masked(dz + (((size_t)b * g.Ho + clampi(ho, 0, g.Ho - 1)) * g.Wo + clampi(wo, 0, g.Wo - 1)) * g.C + c0,
                                      ho >= 0 && ho < g.Ho && wo >= 0 && wo < g.Wo);
            }
          }
#pragma unroll
          for (int kh = 0; kh < K; ++kh) {
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
              float w[8];
              ld8f(wT + (size_t)(kh * K + kw) * g.C + c0, w);
#pragma unroll
              for (int j = 0; j < TW; ++j) {
                float in[8];
                unpack8(pk[kh][j + K - 1 - kw], in);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[j][e] += in[e] * w[e];
              }
            }
          }
        } else {
        auto load_row = [&](int kh, uint4 (&dst)[NIN]) {      // next row requested ahead of this row's FMAs (see dwconv_fwd)
          const int ho = hi + PAD - kh;
          const bool hv = ho >= 0 && ho < g.Ho;
          const bf16* drow = dz + ((size_t)b * g.Ho + clampi(ho, 0, g.Ho - 1)) * g.Wo * g.C + c0;
#pragma unroll
          for (int x = 0; x < NIN; ++x) {
            const int wo = wi0 - PAD + x;
            dst[x] = ld16_masked(drow + (size_t)clampi(wo, 0, g.Wo - 1) * g.C, hv && wo >= 0 && wo < g.Wo);
          }
        };
        uint4 raw[NIN], nxt[NIN];
        load_row(0, raw);
#pragma unroll 1
        for (int kh = 0; kh < K; ++kh) {
          if (kh + 1 < K) load_row(kh + 1, nxt);
          float in[NIN][8];
#pragma unroll
          for (int x = 0; x < NIN; ++x) unpack8(raw[x], in[x]);
#pragma unroll
          for (int kw = 0; kw < K; ++kw) {
            float w[8];
            ld8f(wT + (size_t)(kh * K + kw) * g.C + c0, w);
#pragma unroll
            for (int j = 0; j < TW; ++j)
#pragma unroll
              for (int e = 0; e < 8; ++e) acc[j][e] += in[j + K - 1 - kw][e] * w[e];
          }
#pragma unroll
          for (int x = 0; x < NIN; ++x) raw[x] = nxt[x];
        }
        }
      } else {
      // stride 2: only kernel rows with hi + PAD - kh even reach a dz row, and within such a row the strip's four
      // pixels touch the 4 consecutive dz columns wi0/2 - 1 .. wi0/2 + 2 (wi0 is a multiple of 4): one masked 4-chunk
      // segment per row, tap parity resolved at compile time.
#pragma unroll 1
      for (int kh = (hi + PAD) & 1; kh < K; kh += 2) {
        const int ho = (hi + PAD - kh) >> 1;
        const bool hv = ho >= 0 && ho < g.Ho;
        const bf16* drow = dz + ((size_t)b * g.Ho + clampi(ho, 0, g.Ho - 1)) * g.Wo * g.C + c0;
        const int wb = (wi0 >> 1) - 1;
        float seg[4][8];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          unpack8(ld16_masked(drow + (size_t)clampi(wb + q, 0, g.Wo - 1) * g.C, hv && wb + q >= 0 && wb + q < g.Wo), seg[q]);
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
          float w[8];
          ld8f(wT + (size_t)(kh * K + kw) * g.C + c0, w);
#pragma unroll
          for (int j = 0; j < TW; ++j) {
            if (((j + PAD - kw) & 1) != 0) continue;         // compile-time after unrolling
            const int q = (j + PAD - kw + 4) / 2 - 1;         // = (j + PAD - kw) / 2 + 1 for the even values -2 .. 4
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[j][e] += seg[q][e] * w[e];
          }
        }
      }
      }
      // epilogue operands (z1 or the residual gradient) of all four pixels requested together, bounds-masked
      const char* eptr = z1 ? reinterpret_cast<const char*>(z1) : reinterpret_cast<const char*>(resid);      // 2-byte elements both: z1 fp16, resid bf16
      uint4 er[TW];
      const size_t off0 = (((size_t)b * g.Hi + hi) * g.Wi) * g.C + c0;
      if (eptr) {
#pragma unroll
        for (int j = 0; j < TW; ++j) er[j] = ld16_masked(eptr + 2 * (off0 + (size_t)min(wi0 + j, g.Wi - 1) * g.C), wi0 + j < g.Wi);
      }
#pragma unroll
      for (int j = 0; j < TW; ++j) {
        if (wi0 + j < g.Wi) {
          const size_t off = off0 + (size_t)(wi0 + j) * g.C;
          float o[8];
          if (z1) {
            float zz[8];
            unpack8h(er[j], zz);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = acc[j][e] * silu_grad_f(zz[e] * sc[e] + sh[e]);
            const uint4 pk = pack8(o);
            *reinterpret_cast<uint4*>(out + off) = pk;
            unpack8(pk, o);
#pragma unroll
            for (int e = 0; e < 8; ++e) { st[e] += o[e]; st[8 + e] += o[e] * (zz[e] - mu[e]) * rs[e]; }
          } else {      // plain transposed conv (+ residual gradient): the input was not a BN+SiLU output
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = acc[j][e];
            if (resid) {
              float r[8];
              unpack8(er[j], r);
#pragma unroll
              for (int e = 0; e < 8; ++e) o[e] += r[e];
            }
            *reinterpret_cast<uint4*>(out + off) = pack8(o);
          }
        }
      }
    }
  }
  if (z1) block_reduce_store<16>(st, m, lds, parts + (size_t)blockIdx.x * 2 * g.C, (size_t)g.C);
}

// backward weight (tap-major gT [K*K][C]): one kernel row kh per blockIdx.z; thread = (octet, output row lane)
template <int K, int S>
__global__ __launch_bounds__(256) void dwconv_bwd_weight_kernel(const bf16* __restrict__ dz, const f16* __restrict__ a, float* parts, DwGeom g,
                                                                int rows_per_block, const float* __restrict__ xf_scale,
                                                                const float* __restrict__ xf_shift) {
  constexpr int PAD = K / 2;
  __shared__ float lds[256 * 8 * K];
  const CgMap m = cg_map(g.C);
  const int kh = blockIdx.z;
  float acc[8 * K];
#pragma unroll
  for (int i = 0; i < 8 * K; ++i) acc[i] = 0.f;
  if (m.active) {
    const int c0 = m.cg * 8;
    const int nrows = g.B * g.Ho;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(nrows, r0 + rows_per_block);
    // xf_scale given: `a` is the pre-BatchNorm tensor and the operand silu(scale a + shift) is formed here (stride-2 blocks: the
    // activated tensor is then never stored)
    float sc[8], sh[8];
    if (xf_scale) { ld8f(xf_scale + c0, sc); ld8f(xf_shift + c0, sh); }
    for (int r = r0 + m.rl; r < r1; r += m.nr) {
      const int ho = r % g.Ho, b = r / g.Ho;
      const int hi = ho * S - PAD + kh;
      if (hi < 0 || hi >= g.Hi) continue;
      const f16* arow = a + (((size_t)b * g.Hi + hi) * g.Wi) * g.C + c0;
      const bf16* drow = dz + (((size_t)b * g.Ho + ho) * g.Wo) * g.C + c0;
      // strips of 4 output pixels: 4 dz chunks + (3 S + K) input chunks requested together (bounds-masked, no branches)
      constexpr int NIN = 3 * S + K;
      for (int w0 = 0; w0 < g.Wo; w0 += 4) {
        uint4 dr[4], ar[NIN];
#pragma unroll
        for (int j = 0; j < 4; ++j) dr[j] = ld16_masked(drow + (size_t)min(w0 + j, g.Wo - 1) * g.C, w0 + j < g.Wo);
#pragma unroll
        for (int x = 0; x < NIN; ++x) {
          const int wi = w0 * S - PAD + x;
          ar[x] = ld16_masked(arow + (size_t)clampi(wi, 0, g.Wi - 1) * g.C, wi >= 0 && wi < g.Wi);
        }
        float xin[NIN][8];
#pragma unroll
        for (int x = 0; x < NIN; ++x) unpack8h(ar[x], xin[x]);
        if (xf_scale) {
#pragma unroll
          for (int x = 0; x < NIN; ++x) {
            const int wi = w0 * S - PAD + x;
            const float ok = (wi >= 0 && wi < g.Wi) ? 1.f : 0.f;          // the zero padding is of the ACTIVATED tensor
#pragma unroll
            for (int e = 0; e < 8; ++e) xin[x][e] = h2f(f2h(silu_f(xin[x][e] * sc[e] + sh[e]))) * ok;      // rounded as the forward's staged a1 was
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float d[8];
          unpack8(dr[j], d);
#pragma unroll
          for (int kw = 0; kw < K; ++kw)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[kw * 8 + e] += d[e] * xin[j * S + kw][e];
        }
      }
    }
  }
  block_reduce_store<8 * K>(acc, m, lds, parts + ((size_t)blockIdx.x * K + kh) * K * g.C, (size_t)g.C);
}

// ------------------------------------------------------------------ stem conv 3x3 s2 p1 on the NCHW fp32 image
struct StemGeom { int B, Hi, Wi, Ho, Wo, Co; FastDiv d_wo, d_ho; };
// ---- the stem on the matrix cores.  K = 27 taps (+5 zero) is ONE 16x16x32 bf16 MFMA per 16 pixels x 16 channels, so the
// conv costs its image gathers and nothing else (the VALU version spent 216 FMAs + 54 LDS weight reads per pixel-octet and
// gathered every pixel's 27 taps once per channel octet).
// Forward: D^T[co][p] = W[co][tap] * xcol^T[tap][p].  Lane (p = lane & 15, kg = lane >> 4) gathers taps 8kg .. 8kg+7 of pixel p
// straight from the NCHW fp32 image (clamped address, AND-masked value), rounds them to bf16: that IS the second operand.
// First operand: W rows, register-resident.  Result: lane holds channels 16ct + 4kg .. +3 of pixel p -> one 8-byte store.
__device__ __forceinline__ void stem_tap(int tap, int& ci, int& kh, int& kw) { ci = tap / 9; kh = (tap - ci * 9) / 3; kw = tap - ci * 9 - kh * 3; }

template <int CT>
__global__ __launch_bounds__(256) void stem_fwd_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w, f16* z, float* parts,
                                                            StemGeom g, int pix_per_block) {
  __shared__ float red[4][128];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, pl = lane & 15, kg = lane >> 4;
  bf8 wf[CT], wl[CT];            // fp32 weights and pixels as bf16 hi + lo pairs (hi*hi + lo*hi + hi*lo): fp32-accurate products
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int tap = 8 * kg + j, co = 16 * ct + pl;
      const float v = (tap < 27 && co < g.Co) ? w[co * 27 + tap] : 0.f;
      wf[ct][j] = f2bf(v);
      wl[ct][j] = f2bf(v - bf2f(wf[ct][j]));
    }
  float st[CT][4], sq[CT][4];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int e = 0; e < 4; ++e) { st[ct][e] = 0.f; sq[ct][e] = 0.f; }
  const int npix = g.B * g.Ho * g.Wo;
  const int p0 = blockIdx.x * pix_per_block, p1 = min(npix, p0 + pix_per_block);
  const size_t plane = (size_t)g.Hi * g.Wi;
  for (int pb = p0 + wv * 16; pb < p1; pb += 64) {          // wave-uniform trip count: the MFMA needs every lane
    const bool pv = pb + pl < p1;
    const int p = min(pb + pl, p1 - 1);
    int wo, ho, b, rowi;
    fdivmod((unsigned int)p, g.d_wo, rowi, wo);
    fdivmod((unsigned int)rowi, g.d_ho, b, ho);
    const float* xb = x + (size_t)b * 3 * plane;
    bf8 xf, xl;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int ci, kh, kw;
      stem_tap(min(8 * kg + j, 26), ci, kh, kw);
      const int hi = ho * 2 - 1 + kh, wi = wo * 2 - 1 + kw;
      const unsigned int t = reinterpret_cast<const unsigned int*>(xb + ci * plane)[(size_t)clampi(hi, 0, g.Hi - 1) * g.Wi + clampi(wi, 0, g.Wi - 1)];
      const bool ok = 8 * kg + j < 27 && hi >= 0 && hi < g.Hi && wi >= 0 && wi < g.Wi;
      const float v = __uint_as_float(t & (ok ? 0xffffffffu : 0u));
      xf[j] = f2bf(v);
      xl[j] = f2bf(v - bf2f(xf[j]));
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      f4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[ct], xf, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ct], xl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ct], xf, acc, 0, 0, 0);
      const h4 o = {f2h(acc[0]), f2h(acc[1]), f2h(acc[2]), f2h(acc[3])};
      const int c0 = 16 * ct + 4 * kg;
      if (pv && c0 < g.Co) {
        *reinterpret_cast<h4*>(z + (size_t)p * g.Co + c0) = o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float r = h2f(o[e]); st[ct][e] += r; sq[ct][e] += r * r; }
      }
    }
  }
  // BatchNorm statistics of the ROUNDED outputs: over the 16 pixel lanes, then over the block's waves, one slab row per block
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a = st[ct][e], q = sq[ct][e];
#pragma unroll
      for (int o2 = 1; o2 < 16; o2 <<= 1) { a += __shfl_xor(a, o2, 64); q += __shfl_xor(q, o2, 64); }
      if (pl == 0) { red[wv][16 * ct + 4 * kg + e] = a; red[wv][64 + 16 * ct + 4 * kg + e] = q; }
    }
  __syncthreads();
  if (tid < 128) {
    const int co = tid & 63, which = tid >> 6;
    if (co < g.Co) parts[(size_t)blockIdx.x * 2 * g.Co + which * g.Co + co] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
  }
}

// Weight gradient: D[tap][co] = sum_p x^T[tap][p] dz[p][co], 32 pixels per MFMA.  Each wave parks a chunk's taps tap-major
// ([32 taps][32 pixels] bf16: row reads are the first operand) and its dz rows pixel-major ([32 pixels][Co]: the second operand
// through the transposing read), so dz is read from HBM once (the VALU version re-read every dz row once per tap).
template <int CT>
__global__ __launch_bounds__(256) void stem_wgrad_mfma_kernel(const bf16* __restrict__ dz, const float* __restrict__ x, float* parts, StemGeom g,
                                                              int pix_per_block) {
  constexpr int XP = 80, ZP = 144;             // row pitches in bytes: 32 pixels (64 B) + pad; <= 64 channels (128 B) + pad
  __shared__ __attribute__((aligned(16))) char xs_all[4][2 * 32 * XP];       // bf16 hi image, then the lo image (x = hi + lo: fp32-accurate)
  __shared__ __attribute__((aligned(16))) char zs_all[4][32 * ZP];
  __shared__ float red[32 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, pl = lane & 31, half = lane >> 5, kg = lane >> 4;
  char* xs = xs_all[wv];
  char* zs = zs_all[wv];
  for (int i = tid; i < 32 * 64; i += 256) red[i] = 0.f;
  for (int i = lane; i < 2 * 32 * XP / 4; i += 64) reinterpret_cast<unsigned int*>(xs)[i] = 0u;   // tap rows 27..31 stay zero
  for (int i = lane; i < 32 * ZP / 4; i += 64) reinterpret_cast<unsigned int*>(zs)[i] = 0u;       // channel columns >= Co stay zero
  f4 acc[2][CT];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[tt][ct] = f4{0.f, 0.f, 0.f, 0.f};
  const int npix = g.B * g.Ho * g.Wo, G = g.Co >> 3;
  const int p0 = blockIdx.x * pix_per_block, p1 = min(npix, p0 + pix_per_block);
  const size_t plane = (size_t)g.Hi * g.Wi;
  __syncthreads();
  for (int cb = p0; cb < p1; cb += 128) {            // block-uniform trip count (barriers inside); a wave past the end parks zeros
    const int pb = cb + wv * 32;
    {   // taps: lane = (pixel, half of the taps): 14 masked gathers
      const int p = min(pb + pl, p1 - 1);
      int wo, ho, b, rowi;
      fdivmod((unsigned int)p, g.d_wo, rowi, wo);
      fdivmod((unsigned int)rowi, g.d_ho, b, ho);
      const float* xb = x + (size_t)b * 3 * plane;
      float v[14];
#pragma unroll
      for (int j = 0; j < 14; ++j) {
        int ci, kh, kw;
        stem_tap(min(half * 14 + j, 26), ci, kh, kw);
        const int hi = ho * 2 - 1 + kh, wi = wo * 2 - 1 + kw;
        const unsigned int t = reinterpret_cast<const unsigned int*>(xb + ci * plane)[(size_t)clampi(hi, 0, g.Hi - 1) * g.Wi + clampi(wi, 0, g.Wi - 1)];
        const bool ok = half * 14 + j < 27 && hi >= 0 && hi < g.Hi && wi >= 0 && wi < g.Wi;
        v[j] = __uint_as_float(t & (ok ? 0xffffffffu : 0u));
      }
      // dz rows: 32 x G chunks of 16 bytes over the wave (a pixel past the end contributes zeros, whatever its taps were)
      uint4 dr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = min(lane + 64 * i, 32 * G - 1), row = c / G, ch = c - row * G;
        dr[i] = ld16_masked(dz + (size_t)min(pb + row, p1 - 1) * g.Co + ch * 8, pb + row < p1);
      }
#pragma unroll
      for (int j = 0; j < 14; ++j) {
        const bf16 hi16 = f2bf(v[j]);
        *reinterpret_cast<bf16*>(xs + (half * 14 + j) * XP + pl * 2) = hi16;
        *reinterpret_cast<bf16*>(xs + 32 * XP + (half * 14 + j) * XP + pl * 2) = f2bf(v[j] - bf2f(hi16));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i, row = c / G, ch = c - row * G;
        if (c < 32 * G) *reinterpret_cast<uint4*>(zs + row * ZP + ch * 16) = dr[i];
      }
    }
    __syncthreads();
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const bf8 zf = tr_frag16(zs, ZP, 0, 16 * ct, lane);
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const bf8 xf = *reinterpret_cast<const bf8*>(xs + (16 * tt + (lane & 15)) * XP + kg * 16);
        const bf8 xl = *reinterpret_cast<const bf8*>(xs + 32 * XP + (16 * tt + (lane & 15)) * XP + kg * 16);
        acc[tt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl, zf, acc[tt][ct], 0, 0, 0);
        acc[tt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, zf, acc[tt][ct], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // acc[tt][ct][e] = D[tap 16tt + 4kg + e][co 16ct + (lane & 15)]: the four waves' sums are added in wave order through LDS (no
  // atomics: their arrival order would change the last bit), then one partial-slab row per block [27 * Co]
  for (int wq = 0; wq < 4; ++wq) {
    if (wv == wq) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int e = 0; e < 4; ++e) red[(16 * tt + 4 * kg + e) * 64 + 16 * ct + (lane & 15)] += acc[tt][ct][e];
    }
    __syncthreads();
  }
  for (int i = tid; i < 27 * g.Co; i += 256) {
    const int co = i / 27, tap = i - co * 27;
    parts[(size_t)blockIdx.x * 27 * g.Co + i] = red[tap * 64 + co];
  }
}

// ------------------------------------------------------------------ BatchNorm1d on fp32 [B, C] (thread per channel)
__global__ void bn1d_fwd_kernel(const float* x, const float* gamma, const float* beta, float* y, float* mean_o, float* rstd_o,
                                float* run_mean, float* run_var, int B, int C, float eps, float momentum, int training) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float mu, var;
  if (training) {
    float s = 0.f, q = 0.f;
    for (int b = 0; b < B; ++b) { const float v = x[(size_t)b * C + c]; s += v; }
    mu = s / B;
    for (int b = 0; b < B; ++b) { const float d = x[(size_t)b * C + c] - mu; q += d * d; }
    var = q / B;
    run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mu;
    run_var[c] = (1.f - momentum) * run_var[c] + momentum * var * ((float)B / fmaxf((float)B - 1.f, 1.f));
  } else { mu = run_mean[c]; var = run_var[c]; }
  const float rs = rsqrtf(var + eps);
  if (mean_o) { mean_o[c] = mu; rstd_o[c] = rs; }
  for (int b = 0; b < B; ++b) y[(size_t)b * C + c] = (x[(size_t)b * C + c] - mu) * rs * gamma[c] + beta[c];
}
__global__ void bn1d_bwd_kernel(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                                float* dx, float* dgamma, float* dbeta, int B, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float mu = mean[c], rs = rstd[c];
  float s1 = 0.f, s2 = 0.f;
  for (int b = 0; b < B; ++b) { const float d = dy[(size_t)b * C + c]; s1 += d; s2 += d * (x[(size_t)b * C + c] - mu) * rs; }
  dgamma[c] += s2; dbeta[c] += s1;
  const float g = gamma[c] * rs;
  for (int b = 0; b < B; ++b) {
    const float zh = (x[(size_t)b * C + c] - mu) * rs;
    dx[(size_t)b * C + c] = g * (dy[(size_t)b * C + c] - s1 / B - zh * s2 / B);
  }
}

// elementwise helpers for the tower top
__global__ void dropout_cast_kernel(const float* x, f16* y, size_t n, unsigned long long seed, unsigned int stream, unsigned int thresh,
                                    float inv_keep, const unsigned long long* seed_dev) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = x[i];
  if (thresh) v = drop_keep(step_seed(seed, seed_dev), stream, i, thresh) ? v * inv_keep : 0.f;
  y[i] = f2h(v);
}
__global__ void dropout_bwd_kernel(const float* dy, float* dx, size_t n, unsigned long long seed, unsigned int stream, unsigned int thresh,
                                   float inv_keep, const unsigned long long* seed_dev) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = dy[i];
  if (thresh) v = drop_keep(step_seed(seed, seed_dev), stream, i, thresh) ? v * inv_keep : 0.f;
  dx[i] = v;
}
// global-average-pool backward fused with the head BN+SiLU backward input: dy[b,hw,c] = dpool[b,c] / HW   (bf16)
__global__ __launch_bounds__(256) void broadcast_pool_grad_kernel(const float* dpool, bf16* dy, int HW, int C, size_t nchunks) {
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < nchunks; q += (size_t)gridDim.x * 256) {
    const size_t el = q * 8;
    const int c0 = (int)(el % (size_t)C);
    const size_t b = el / ((size_t)HW * C);
    float f[8];
    ld8f(dpool + b * C + c0, f);
    const float inv = 1.0f / HW;
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] *= inv;
    *reinterpret_cast<uint4*>(dy + el) = pack8(f);
  }
}

// ================================================================= C-ABI
#define REQ_C8(C, name) MMSIM_REQUIRE((C) > 0 && ((C) % 8) == 0, name ": channel count must be a positive multiple of 8")

static int rows_per_block_for(int P, int nr) {
  // aim for ~1024 blocks, at least nr rows each
  int rpb = (P + 1023) / 1024;
  if (rpb < nr) rpb = nr;
  return rpb;
}
static int nr_of(int C) { const int G = C >> 3; return G >= 256 ? 1 : 256 / G; }

#define REQ_SCRATCH(need, name) MMSIM_REQUIRE(scratch && scratch_floats >= (unsigned long long)(need), name ": scratch too small")

extern "C" int mmsim_bn_stats(const void* z, float* sums, int P, int C, float* scratch, unsigned long long scratch_floats,
                              void* stream) {
  MMSIM_REQUIRE(z && sums && P > 0, "bn_stats: bad arguments"); REQ_C8(C, "bn_stats");
  const int rpb = rows_per_block_for(P, nr_of(C));
  const int nparts = (P + rpb - 1) / rpb;
  REQ_SCRATCH((size_t)nparts * 2 * C, "bn_stats");
  hipLaunchKernelGGL(bn_stats_kernel, dim3(nparts, cg_grid_y(C)), dim3(256), 0, (hipStream_t)stream, (const f16*)z, scratch, P, C, rpb);
  launch_reduce(scratch, nparts, 2 * C, sums, 1, (hipStream_t)stream);      /* sums are pre-zeroed by the caller (header contract) */
  return mmsim_check_launch("bn_stats");
}

// Attach the finalisation of a BatchNorm to the reduction of its statistics: the NEXT partial-slab reduction into `sums` issued on
// this host thread (inside mmsim_gemm_bf16_bnstats, mmsim_dwtile_fwd, mmsim_pw_*_fwd, mmsim_stem_fwd, ...) is done by
// reduce_bn_finalize_kernel, which also produces mean / rstd / scale / shift / running statistics.  mmsim_bn_finalize_flush launches the
// plain finalisation if nothing consumed the request, and is a no-op otherwise.
extern "C" int mmsim_bn_finalize_arm(float* sums, const float* gamma, const float* beta, float* mean, float* rstd, float* scale,
                                     float* shift, float* run_mean, float* run_var, int C, float count, float eps, float momentum) {
  MMSIM_REQUIRE(sums && gamma && beta && mean && rstd && scale && shift && C > 0 && count > 0, "bn_finalize_arm: bad arguments");
  g_bnfin.armed = true; g_bnfin.sums = sums;
  BnFinDev& d = g_bnfin.d;
  d.gamma = gamma; d.beta = beta; d.mean = mean; d.rstd = rstd; d.scale = scale; d.shift = shift; d.run_mean = run_mean; d.run_var = run_var;
  d.C = C; d.count = count; d.eps = eps; d.momentum = momentum;
  return 0;
}
extern "C" int mmsim_bn_finalize(const float* sums, const float* gamma, const float* beta, float* mean, float* rstd, float* scale,
                                 float* shift, float* run_mean, float* run_var, int C, float count, float eps, float momentum, void* stream);
extern "C" int mmsim_bn_finalize_flush(void* stream) {
  if (!g_bnfin.armed) return 0;
  g_bnfin.armed = false;
  const BnFinDev& d = g_bnfin.d;
  return mmsim_bn_finalize(g_bnfin.sums, d.gamma, d.beta, d.mean, d.rstd, d.scale, d.shift, d.run_mean, d.run_var, d.C, d.count, d.eps,
                           d.momentum, stream);
}

extern "C" int mmsim_bn_finalize(const float* sums, const float* gamma, const float* beta, float* mean, float* rstd, float* scale,
                                 float* shift, float* run_mean, float* run_var, int C, float count, float eps, float momentum,
                                 void* stream) {
  MMSIM_REQUIRE(sums && gamma && beta && mean && rstd && scale && shift && C > 0 && count > 0, "bn_finalize: bad arguments");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, gamma, beta, mean, rstd,
                     scale, shift, run_mean, run_var, C, count, eps, momentum);
  return mmsim_check_launch("bn_finalize");
}

extern "C" int mmsim_bn_apply(const void* z, const float* scale, const float* shift, const void* resid, void* out, int P, int C,
                              int act_silu, void* stream) {
  MMSIM_REQUIRE(z && scale && shift && out && P > 0, "bn_apply: bad arguments"); REQ_C8(C, "bn_apply");
  const size_t nch = (size_t)P * C / 8;
  size_t gsz = (nch + 255) / 256; if (gsz > 16384) gsz = 16384;
  hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)gsz), dim3(256), 0, (hipStream_t)stream, (const f16*)z, scale, shift,
                     (const f16*)resid, (f16*)out, nch, C, act_silu);
  return mmsim_check_launch("bn_apply");
}

static int pool_bn_act_impl(const void* z, const float* scale, const float* shift, const void* other, float* out, int B,
                            int HW, int C, int act_silu, float mul, void* act_out, void* stream) {
  MMSIM_REQUIRE(z && scale && shift && out && B > 0 && HW > 0, "pool_bn_act: bad arguments"); REQ_C8(C, "pool_bn_act");
  // large feature maps: split the HW range over blockIdx.z so that more than B blocks stream (few adders per output)
  int nz = 1;
  const int nr = pool_nr(C), gy = pool_grid_y(C);
  while (!mmsim_deterministic() && nz < 16 && HW / (nz * 2 * nr) >= 16 && B * gy * nz < 2048) nz *= 2;      // >= 16 rows per row lane and z-slice: a slice costs 8 C atomics
  const int rpz = (HW + nz - 1) / nz;
  if (nz > 1) (void)hipMemsetAsync(out, 0, (size_t)B * C * sizeof(float), (hipStream_t)stream);
  if (other) hipLaunchKernelGGL(pool_bn_act_kernel<true>, dim3(B, gy, nz), dim3(256), 0, (hipStream_t)stream, (const f16*)z, scale, shift,
                                (const bf16*)other, out, HW, C, act_silu, mul, rpz, (f16*)act_out);
  else hipLaunchKernelGGL(pool_bn_act_kernel<false>, dim3(B, gy, nz), dim3(256), 0, (hipStream_t)stream, (const f16*)z, scale, shift,
                          (const bf16*)nullptr, out, HW, C, act_silu, mul, rpz, (f16*)act_out);
  return mmsim_check_launch("pool_bn_act");
}

extern "C" int mmsim_pool_bn_act(const void* z, const float* scale, const float* shift, const void* other, float* out, int B,
                                 int HW, int C, int act_silu, float mul, void* stream) {
  return pool_bn_act_impl(z, scale, shift, other, out, B, HW, C, act_silu, mul, nullptr, stream);
}

extern "C" int mmsim_pool_bn_act_store(const void* z, const float* scale, const float* shift, void* act_out, float* out, int B,
                                       int HW, int C, float mul, void* stream) {
  MMSIM_REQUIRE(act_out, "pool_bn_act_store: act_out required");
  return pool_bn_act_impl(z, scale, shift, nullptr, out, B, HW, C, 1, mul, act_out, stream);
}

/* weT: scratch [RD][C] receiving conv_expand.weight transposed (reused by the backward of the same step) */
extern "C" int mmsim_pool_bn_bwd(const void* z, const float* scale, const float* shift, const float* mean, const float* rstd,
                                 const void* dy, float* out5, int B, int HW, int C, void* stream) {
  MMSIM_REQUIRE(z && scale && shift && mean && rstd && dy && out5 && B > 0 && HW > 0, "pool_bn_bwd: bad arguments"); REQ_C8(C, "pool_bn_bwd");
  const int nr = pool_nr(C), gy = pool_grid_y(C);
  int nz = 1;
  while (!mmsim_deterministic() && nz < 16 && HW / (nz * 2 * nr) >= 16 && B * gy * nz < 2048) nz *= 2;
  const int rpz = (HW + nz - 1) / nz;
  if (nz > 1) (void)hipMemsetAsync(out5, 0, (size_t)5 * B * C * sizeof(float), (hipStream_t)stream);
  hipLaunchKernelGGL(pool_bn_bwd_kernel, dim3(B, gy, nz), dim3(256), 0, (hipStream_t)stream, (const f16*)z, scale, shift,
                     mean, rstd, (const bf16*)dy, out5, B, HW, C, rpz);
  return mmsim_check_launch("pool_bn_bwd");
}

extern "C" int mmsim_bn_bwd_sums_from_pool(const float* out5, const float* gate, const float* dsq, float* sums, int B, int HW,
                                           int C, void* stream) {
  MMSIM_REQUIRE(out5 && gate && dsq && sums && B > 0 && HW > 0 && C > 0, "bn_bwd_sums_from_pool: bad arguments");
  hipLaunchKernelGGL(bn_bwd_sums_from_pool_kernel, dim3((C + 63) / 64, (B >= 64 && !mmsim_deterministic()) ? 8 : 1), dim3(256), 0, (hipStream_t)stream, out5, gate, dsq, sums, B, C,
                     1.0f / (float)HW);
  return mmsim_check_launch("bn_bwd_sums_from_pool");
}

extern "C" int mmsim_se_mlp_fwd(const float* s, const float* w_reduce, const float* b_reduce, const float* w_expand,
                                const float* b_expand, float* weT, float* hr, float* hs, float* gate, int B, int C, int RD,
                                void* stream) {
  MMSIM_REQUIRE(s && w_reduce && b_reduce && b_expand && weT && hr && hs && gate && B > 0 && C > 0 && RD > 0, "se_mlp_fwd: bad arguments");
  MMSIM_REQUIRE(C % 8 == 0 && RD <= 128, "se_mlp_fwd: C must be a multiple of 8 and RD <= 128");
  hipStream_t st = (hipStream_t)stream;
  if (w_expand)        // NULL: weT already holds conv_expand.weight transposed (made for all blocks at once, mmsim_dw_tap_major_batch)
    hipLaunchKernelGGL(se_transpose_kernel, dim3((C * RD + 255) / 256), dim3(256), 0, st, w_expand, weT, C, RD, 0);
  hipLaunchKernelGGL(se_rowdot_kernel, dim3(RD, (B + 7) / 8), dim3(256), 0, st, w_reduce, s, (const float*)nullptr, b_reduce,
                     (const float*)nullptr, hr, hs, B, C, RD, SE_PRE_NONE, SE_POST_NONE);
  hipLaunchKernelGGL(se_colmix_kernel, dim3((C + 63) / 64, (B + 3) / 4), dim3(256), 0, st, weT, hr, b_expand, gate, B, C, RD, SE_PRE_SILU,
                     SE_POST_SIGMOID);
  return mmsim_check_launch("se_mlp_fwd");
}

/* dr [B,RD], ds [B,C]: outputs; dweT: scratch [RD][C].  Weight / bias gradients are accumulated. */
extern "C" int mmsim_se_mlp_bwd(const float* dgate, const float* gate, const float* hr, const float* hs, const float* s,
                                const float* w_reduce, const float* weT, float* dr, float* ds, float* dweT, float* dw_reduce,
                                float* db_reduce, float* dw_expand, float* db_expand, int B, int C, int RD, void* stream) {
  MMSIM_REQUIRE(dgate && gate && hr && hs && s && w_reduce && weT && dr && ds && dweT && dw_reduce && db_reduce && db_expand,
                "se_mlp_bwd: null operand");
  MMSIM_REQUIRE(C % 8 == 0 && RD <= 128, "se_mlp_bwd: C must be a multiple of 8 and RD <= 128");
  hipStream_t st = (hipStream_t)stream;
  // dr[b,j] = (sum_c We[c,j] dpe[b,c]) * silu'(hr[b,j]),  dpe = dgate * g * (1 - g)
  hipLaunchKernelGGL(se_rowdot_kernel, dim3(RD, (B + 7) / 8), dim3(256), 0, st, weT, dgate, gate, (const float*)nullptr, hr, dr,
                     (float*)nullptr, B, C, RD, SE_PRE_DSIGMOID, SE_POST_MUL_DSILU);
  // ds[b,c] = sum_j Wr[j,c] dr[b,j]
  hipLaunchKernelGGL(se_colmix_kernel, dim3((C + 63) / 64, (B + 3) / 4), dim3(256), 0, st, w_reduce, dr, (const float*)nullptr, ds, B, C, RD,
                     SE_PRE_NONE, SE_POST_NONE);
  // dw_expand NULL: dweT is the caller's zeroed accumulation buffer (transposed into the gradient for all blocks at once later);
  // otherwise dweT is scratch: zeroed here and transposed-accumulated into dw_expand
  if (dw_expand) (void)hipMemsetAsync(dweT, 0, (size_t)C * RD * sizeof(float), st);
  hipLaunchKernelGGL(se_wgrad_kernel, dim3((C + 63) / 64, (RD + 15) / 16, mmsim_deterministic() ? 1 : 4), dim3(64), 0, st, dgate, gate, dr, hs, s, dw_reduce, db_reduce,
                     dweT, db_expand, B, C, RD);
  if (dw_expand) hipLaunchKernelGGL(se_transpose_kernel, dim3((C * RD + 255) / 256), dim3(256), 0, st, dweT, dw_expand, RD, C, 1);
  return mmsim_check_launch("se_mlp_bwd");
}

static BnBwd mk_bnbwd(const void* dy, const void* z, const float* mean, const float* rstd, const float* scale, const float* shift,
                      const float* gate, const float* dsq, int hw, int act, int P, int C) {
  BnBwd p;
  p.dy = (const bf16*)dy; p.z = (const f16*)z; p.mean = mean; p.rstd = rstd; p.scale = scale; p.shift = shift;
  p.gate = gate; p.dsq = dsq; p.hw = hw > 0 ? hw : 1; p.act = act; p.P = P; p.C = C; p.inv_hw = 1.0f / (float)(hw > 0 ? hw : 1); p.d_hw = make_fastdiv(hw > 0 ? hw : 1);
  return p;
}

extern "C" int mmsim_bn_bwd(const void* dy, const void* z, const float* mean, const float* rstd, const float* scale,
                            const float* shift, const float* gate, const float* dsq, int hw, int act_silu, float* sums,
                            int sums_ready, void* dz, float* dgamma, float* dbeta, int P, int C, float* scratch,
                            unsigned long long scratch_floats, void* stream) {
  MMSIM_REQUIRE(dy && z && mean && rstd && scale && shift && sums && dz && P > 0, "bn_bwd: bad arguments"); REQ_C8(C, "bn_bwd");
  MMSIM_REQUIRE((gate == nullptr) == (dsq == nullptr), "bn_bwd: gate and dsq come together");
  const BnBwd p = mk_bnbwd(dy, z, mean, rstd, scale, shift, gate, dsq, hw, act_silu, P, C);
  const int rpb = rows_per_block_for(P, nr_of(C));
  dim3 grid((P + rpb - 1) / rpb, cg_grid_y(C));
  if (!sums_ready) {
    REQ_SCRATCH((size_t)grid.x * 2 * C, "bn_bwd");
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, grid, dim3(256), 0, (hipStream_t)stream, p, scratch, rpb);
    launch_reduce(scratch, grid.x, 2 * C, sums, 1, (hipStream_t)stream);
  }
  hipLaunchKernelGGL(bn_bwd_apply_kernel, grid, dim3(256), 0, (hipStream_t)stream, p, sums, (bf16*)dz, dgamma, dbeta, rpb);
  return mmsim_check_launch("bn_bwd");
}

extern "C" int mmsim_dw_weight_to_tap_major(const float* w, float* wT, int C, int K, void* stream) {
  MMSIM_REQUIRE(w && wT && C > 0 && (K == 3 || K == 5), "dw_weight_to_tap_major: bad arguments");
  hipLaunchKernelGGL(dw_weight_to_tap_major_kernel, dim3((C * K * K + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, wT, C, K * K);
  return mmsim_check_launch("dw_weight_to_tap_major");
}
extern "C" int mmsim_dw_grad_from_tap_major(const float* gT, float* g, int C, int K, void* stream) {
  MMSIM_REQUIRE(gT && g && C > 0 && (K == 3 || K == 5), "dw_grad_from_tap_major: bad arguments");
  hipLaunchKernelGGL(dw_grad_from_tap_major_kernel, dim3((C * K * K + 255) / 256), dim3(256), 0, (hipStream_t)stream, gT, g, C, K * K);
  return mmsim_check_launch("dw_grad_from_tap_major");
}

static int dw_check(int B, int Hi, int Wi, int C, int K, int S, DwGeom* g) {
  MMSIM_REQUIRE(B > 0 && Hi > 0 && Wi > 0, "dwconv: bad geometry"); REQ_C8(C, "dwconv");
  MMSIM_REQUIRE((K == 3 || K == 5) && (S == 1 || S == 2), "dwconv: kernel 3/5 and stride 1/2 only");
  g->B = B; g->Hi = Hi; g->Wi = Wi; g->C = C;
  g->Ho = (Hi + 2 * (K / 2) - K) / S + 1; g->Wo = (Wi + 2 * (K / 2) - K) / S + 1;
  return MMSIM_OK;
}
#define DW_DISPATCH(KERNEL, ...)                                                                       \
  if (K == 3 && S == 1) hipLaunchKernelGGL((KERNEL<3, 1>), __VA_ARGS__);                               \
  else if (K == 3 && S == 2) hipLaunchKernelGGL((KERNEL<3, 2>), __VA_ARGS__);                          \
  else if (K == 5 && S == 1) hipLaunchKernelGGL((KERNEL<5, 1>), __VA_ARGS__);                          \
  else hipLaunchKernelGGL((KERNEL<5, 2>), __VA_ARGS__);

extern "C" int mmsim_dwconv_fwd(const void* a, const float* w_tap_major, void* z, float* sums, int B, int Hi, int Wi, int C, int K,
                                int S, float* scratch, unsigned long long scratch_floats, void* stream) {
  DwGeom g; int rc = dw_check(B, Hi, Wi, C, K, S, &g); if (rc) return rc;
  MMSIM_REQUIRE(a && w_tap_major && z && sums, "dwconv_fwd: null operand");
  // V (see the kernel): packed up-front loads pay for 3x3 (2x faster, 72-96 VGPRs of loads in flight); 5x5 would need 160
  // and is best served by the row-at-a-time form.  MMSIM_DW_VARIANT=0/1 forces one form (tools/bench_dw.py).
  static int variant = -2;
  if (variant == -2) { const char* e = getenv("MMSIM_DW_VARIANT"); variant = e ? atoi(e) % 10 : -1; }
  // K = 5 never takes the up-front form: its instantiations need 160 registers of loads in flight and compiled to 256 VGPRs + 1.5 KB of
  // scratch per lane -- they are no longer instantiated; MMSIM_DW_VARIANT only selects among the 3 x 3 forms.
  const int Vv = K == 5 ? 0 : (variant >= 0 ? variant : 1);
  g.d_strip = make_fastdiv((g.Wo + 3) / 4); g.d_rows = make_fastdiv(g.Ho);
  const int nitems = B * g.Ho * ((g.Wo + 3) / 4);
  const int ipb = rows_per_block_for(nitems, nr_of(C));
  dim3 grid((nitems + ipb - 1) / ipb, cg_grid_y(C));
  REQ_SCRATCH((size_t)grid.x * 2 * C, "dwconv_fwd");
#define DWF(KK, SS, TT, VV) hipLaunchKernelGGL((dwconv_fwd_kernel<KK, SS, TT, VV>), grid, dim3(256), 0, (hipStream_t)stream, (const f16*)a, w_tap_major, (f16*)z, scratch, g, ipb)
  if (K == 3 && S == 1) { if (Vv == 1) DWF(3, 1, 4, 1); else DWF(3, 1, 4, 0); }
  else if (K == 3 && S == 2) { if (Vv == 1) DWF(3, 2, 4, 1); else DWF(3, 2, 4, 0); }
  else if (K == 5 && S == 1) DWF(5, 1, 4, 0);
  else DWF(5, 2, 4, 0);
#undef DWF
  launch_reduce(scratch, grid.x, 2 * C, sums, 1, (hipStream_t)stream);
  return mmsim_check_launch("dwconv_fwd");
}

extern "C" int mmsim_dwconv_bwd_data(const void* dz, const float* w_tap_major, const void* z1, const float* mean, const float* rstd,
                                     const float* scale, const float* shift, const void* resid, void* dpre, float* sums, int B,
                                     int Hi, int Wi, int C, int K, int S, float* scratch, unsigned long long scratch_floats,
                                     void* stream) {
  DwGeom g; int rc = dw_check(B, Hi, Wi, C, K, S, &g); if (rc) return rc;
  MMSIM_REQUIRE(dz && w_tap_major && dpre, "dwconv_bwd_data: null operand");
  MMSIM_REQUIRE(!z1 || (mean && rstd && scale && shift && sums), "dwconv_bwd_data: fused BN+SiLU backward needs the BN state");
  MMSIM_REQUIRE(!(z1 && resid), "dwconv_bwd_data: resid only in the plain (z1 == NULL) form");
  g.d_strip = make_fastdiv((Wi + 3) / 4); g.d_rows = make_fastdiv(Hi);
  const int nitems = B * Hi * ((Wi + 3) / 4);
  const int ipb = rows_per_block_for(nitems, nr_of(C));
  dim3 grid((nitems + ipb - 1) / ipb, cg_grid_y(C));
  if (z1) REQ_SCRATCH((size_t)grid.x * 2 * C, "dwconv_bwd_data");
  DW_DISPATCH(dwconv_bwd_data_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)dz, w_tap_major, (const f16*)z1, mean,
              rstd, scale, shift, (const bf16*)resid, (bf16*)dpre, scratch, g, ipb)
  if (z1) launch_reduce(scratch, grid.x, 2 * C, sums, 1, (hipStream_t)stream);
  return mmsim_check_launch("dwconv_bwd_data");
}

static int dwconv_bwd_weight_impl(const void* dz, const void* a, const float* xf_scale, const float* xf_shift, float* g_tap_major, int B,
                                  int Hi, int Wi, int C, int K, int S, float* scratch, unsigned long long scratch_floats, void* stream);
extern "C" int mmsim_dwconv_bwd_weight(const void* dz, const void* a, float* g_tap_major, int B, int Hi, int Wi, int C, int K, int S,
                                       float* scratch, unsigned long long scratch_floats, void* stream) {
  return dwconv_bwd_weight_impl(dz, a, nullptr, nullptr, g_tap_major, B, Hi, Wi, C, K, S, scratch, scratch_floats, stream);
}
extern "C" int mmsim_dwconv_bwd_weight_xf(const void* dz, const void* z, const float* xf_scale, const float* xf_shift, float* g_tap_major,
                                          int B, int Hi, int Wi, int C, int K, int S, float* scratch, unsigned long long scratch_floats,
                                          void* stream) {
  MMSIM_REQUIRE(xf_scale && xf_shift, "dwconv_bwd_weight_xf: scale and shift required");
  return dwconv_bwd_weight_impl(dz, z, xf_scale, xf_shift, g_tap_major, B, Hi, Wi, C, K, S, scratch, scratch_floats, stream);
}
static int dwconv_bwd_weight_impl(const void* dz, const void* a, const float* xf_scale, const float* xf_shift, float* g_tap_major, int B,
                                  int Hi, int Wi, int C, int K, int S, float* scratch, unsigned long long scratch_floats, void* stream) {
  DwGeom g; int rc = dw_check(B, Hi, Wi, C, K, S, &g); if (rc) return rc;
  MMSIM_REQUIRE(dz && a && g_tap_major, "dwconv_bwd_weight: null operand");
  const int nrows = B * g.Ho;
  int rpb = (nrows + 127) / 128; const int nr = nr_of(C); if (rpb < nr) rpb = nr;
  dim3 grid((nrows + rpb - 1) / rpb, cg_grid_y(C), K);
  REQ_SCRATCH((size_t)grid.x * K * K * C, "dwconv_bwd_weight");
  DW_DISPATCH(dwconv_bwd_weight_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)dz, (const f16*)a, scratch, g, rpb, xf_scale, xf_shift)
  launch_reduce(scratch, grid.x, K * K * C, g_tap_major, 1, (hipStream_t)stream);      /* g_tap_major += */
  return mmsim_check_launch("dwconv_bwd_weight");
}

extern "C" int mmsim_dw_tap_major_batch(const void* segs_dev, int nseg, const float* src_base, float* dst_base, int to_tap_major,
                                        int max_elems, void* stream) {
  MMSIM_REQUIRE(segs_dev && src_base && dst_base && nseg > 0 && max_elems > 0, "dw_tap_major_batch: bad arguments");
  int gx = (max_elems + 255) / 256; if (gx > 64) gx = 64;
  hipLaunchKernelGGL(dw_tap_major_batch_kernel, dim3(gx, nseg), dim3(256), 0, (hipStream_t)stream, (const long long*)segs_dev, src_base,
                     dst_base, to_tap_major);
  return mmsim_check_launch("dw_tap_major_batch");
}

extern "C" int mmsim_stem_fwd(const float* x, const float* w, void* z, float* sums, int B, int Hi, int Wi, int Co, float* scratch,
                              unsigned long long scratch_floats, void* stream) {
  MMSIM_REQUIRE(x && w && z && sums && B > 0 && Hi > 1 && Wi > 1, "stem_fwd: bad arguments"); REQ_C8(Co, "stem_fwd");
  MMSIM_REQUIRE(Co <= 64, "stem_fwd: at most 64 output channels");
  StemGeom g; g.B = B; g.Hi = Hi; g.Wi = Wi; g.Ho = (Hi + 2 - 3) / 2 + 1; g.Wo = (Wi + 2 - 3) / 2 + 1; g.Co = Co; g.d_wo = make_fastdiv(g.Wo); g.d_ho = make_fastdiv(g.Ho);
  const int npix = B * g.Ho * g.Wo;
  const int ppb = ((npix + 4095) / 4096 + 63) / 64 * 64;          // <= 4096 blocks, whole 16-pixel x 4-wave trips
  const int nparts = (npix + ppb - 1) / ppb;
  REQ_SCRATCH((size_t)nparts * 2 * Co, "stem_fwd");
  const int CT = (Co + 15) / 16;
  if (CT == 1) hipLaunchKernelGGL((stem_fwd_mfma_kernel<1>), dim3(nparts), dim3(256), 0, (hipStream_t)stream, x, w, (f16*)z, scratch, g, ppb);
  else if (CT == 2) hipLaunchKernelGGL((stem_fwd_mfma_kernel<2>), dim3(nparts), dim3(256), 0, (hipStream_t)stream, x, w, (f16*)z, scratch, g, ppb);
  else if (CT == 3) hipLaunchKernelGGL((stem_fwd_mfma_kernel<3>), dim3(nparts), dim3(256), 0, (hipStream_t)stream, x, w, (f16*)z, scratch, g, ppb);
  else hipLaunchKernelGGL((stem_fwd_mfma_kernel<4>), dim3(nparts), dim3(256), 0, (hipStream_t)stream, x, w, (f16*)z, scratch, g, ppb);
  launch_reduce(scratch, nparts, 2 * Co, sums, 1, (hipStream_t)stream);
  return mmsim_check_launch("stem_fwd");
}

extern "C" int mmsim_stem_wgrad(const void* dz, const float* x, float* dw, int B, int Hi, int Wi, int Co, float* scratch,
                                unsigned long long scratch_floats, void* stream) {
  MMSIM_REQUIRE(dz && x && dw && B > 0, "stem_wgrad: bad arguments"); REQ_C8(Co, "stem_wgrad");
  MMSIM_REQUIRE(Co <= 64, "stem_wgrad: at most 64 output channels");
  StemGeom g; g.B = B; g.Hi = Hi; g.Wi = Wi; g.Ho = (Hi + 2 - 3) / 2 + 1; g.Wo = (Wi + 2 - 3) / 2 + 1; g.Co = Co; g.d_wo = make_fastdiv(g.Wo); g.d_ho = make_fastdiv(g.Ho);
  const int npix = B * g.Ho * g.Wo;
  const int ppb = ((npix + 511) / 512 + 127) / 128 * 128;         // <= 512 blocks (one slab row each), whole 4-wave x 32-pixel trips
  const dim3 grid((npix + ppb - 1) / ppb), block(256);
  REQ_SCRATCH((size_t)grid.x * 27 * Co, "stem_wgrad");
  const int CT = (Co + 15) / 16;
  if (CT == 1) hipLaunchKernelGGL((stem_wgrad_mfma_kernel<1>), grid, block, 0, (hipStream_t)stream, (const bf16*)dz, x, scratch, g, ppb);
  else if (CT == 2) hipLaunchKernelGGL((stem_wgrad_mfma_kernel<2>), grid, block, 0, (hipStream_t)stream, (const bf16*)dz, x, scratch, g, ppb);
  else if (CT == 3) hipLaunchKernelGGL((stem_wgrad_mfma_kernel<3>), grid, block, 0, (hipStream_t)stream, (const bf16*)dz, x, scratch, g, ppb);
  else hipLaunchKernelGGL((stem_wgrad_mfma_kernel<4>), grid, block, 0, (hipStream_t)stream, (const bf16*)dz, x, scratch, g, ppb);
  launch_reduce(scratch, grid.x, 27 * Co, dw, 1, (hipStream_t)stream);      /* dw += */
  return mmsim_check_launch("stem_wgrad");
}

extern "C" int mmsim_bn1d_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                              float* run_mean, float* run_var, int B, int C, float eps, float momentum, int training, void* stream) {
  MMSIM_REQUIRE(x && gamma && beta && y && run_mean && run_var && B > 0 && C > 0, "bn1d_fwd: bad arguments");
  hipLaunchKernelGGL(bn1d_fwd_kernel, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)stream, x, gamma, beta, y, mean, rstd, run_mean,
                     run_var, B, C, eps, momentum, training);
  return mmsim_check_launch("bn1d_fwd");
}
extern "C" int mmsim_bn1d_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, float* dx,
                              float* dgamma, float* dbeta, int B, int C, void* stream) {
  MMSIM_REQUIRE(dy && x && mean && rstd && gamma && dx && dgamma && dbeta && B > 0 && C > 0, "bn1d_bwd: bad arguments");
  hipLaunchKernelGGL(bn1d_bwd_kernel, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)stream, dy, x, mean, rstd, gamma, dx, dgamma,
                     dbeta, B, C);
  return mmsim_check_launch("bn1d_bwd");
}

extern "C" int mmsim_dropout_cast(const float* x, void* y_f16, unsigned long long n, float p, unsigned long long seed,
                                  unsigned int stream_id, void* stream) {
  MMSIM_REQUIRE(x && y_f16 && p >= 0.f && p < 1.f, "dropout_cast: bad arguments");
  if (n == 0) return MMSIM_OK;
  hipLaunchKernelGGL(dropout_cast_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (f16*)y_f16,
                     (size_t)n, seed, stream_id, p > 0.f ? (unsigned int)((double)p * 4294967296.0) : 0u, 1.0f / (1.0f - p), mmsim_step_seed_ptr());
  return mmsim_check_launch("dropout_cast");
}
extern "C" int mmsim_dropout_bwd(const float* dy, float* dx, unsigned long long n, float p, unsigned long long seed,
                                 unsigned int stream_id, void* stream) {
  MMSIM_REQUIRE(dy && dx && p >= 0.f && p < 1.f, "dropout_bwd: bad arguments");
  if (n == 0) return MMSIM_OK;
  hipLaunchKernelGGL(dropout_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, dx, (size_t)n, seed,
                     stream_id, p > 0.f ? (unsigned int)((double)p * 4294967296.0) : 0u, 1.0f / (1.0f - p), mmsim_step_seed_ptr());
  return mmsim_check_launch("dropout_bwd");
}
extern "C" int mmsim_broadcast_pool_grad(const float* dpool, void* dy, int B, int HW, int C, void* stream) {
  MMSIM_REQUIRE(dpool && dy && B > 0 && HW > 0, "broadcast_pool_grad: bad arguments"); REQ_C8(C, "broadcast_pool_grad");
  const size_t nch = (size_t)B * HW * C / 8;
  size_t gsz = (nch + 255) / 256; if (gsz > 16384) gsz = 16384;
  hipLaunchKernelGGL(broadcast_pool_grad_kernel, dim3((unsigned)gsz), dim3(256), 0, (hipStream_t)stream, dpool, (bf16*)dy, HW, C, nch);
  return mmsim_check_launch("broadcast_pool_grad");
}
