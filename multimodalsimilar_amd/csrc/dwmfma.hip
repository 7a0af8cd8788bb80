// 5 x 5 stride-1 depthwise convolution of the MBConv block on the MATRIX cores (gfx950), round 4.
// timm `conv_dw` (k = 5, s = 1: stages 2, 4, 5 of EfficientNet-B4 under cv_classifier.py:49) + the BatchNorm / SiLU / squeeze-excite
// arithmetic around it, forward and backward.  Element types as everywhere in the image tower: forward tensors fp16, gradients bf16.
//
// Why: the LDS-tiled VALU kernels (mbconv.hip, dwt_*<5>) run at 1.0-1.6 TB/s of their algorithmic bytes -- 4-5 x off the HBM roofline
// (r04 tools/bench_dwt.py: 28^2 x 336: 175 / 401 us forward / backward, 14^2 x 960: 121 / 335, 7^2 x 1632: 83 / 151): 25 (forward) or
// 50 (backward) fp32 MACs per element on the vector ALUs, their taps re-read from LDS per strip, and a 2.2 x halo of small tiles whose
// every staged element pays BatchNorm + SiLU.  The chip's vector FMA rate is 60 T MAC/s; the 16-block matrix instruction
//     v_mfma_f32_4x4x4_16b_{f16,bf16}:  16 INDEPENDENT 4 x 4 x 4 products per wave-instruction (block = 4 lanes), 254 T MAC/s measured
// (tools/probe/mfma44.hip) is what a depthwise convolution maps to: block = channel.  A row of a 5-tap correlation over 4 outputs,
//     out[x0 + i] = sum_kw w[kw] in[x0 + i + kw],   i = 0..3,
// is two 4 x 4 x 4 products with banded (Toeplitz) weight matrices:  D[i][j] += T0[i][k] in_j[x0 + k] + T1[i][k] in_j[x0 + 4 + k],
// T0[i][k] = w[k - i] (k >= i), T1[i][k] = w[4 + k - i] (k <= i): 20 of 32 MACs useful.  j = four image rows, so a wave-instruction
// produces 4 x 4 outputs of 16 channels for one kernel row; 5 kernel rows x 2 blocks = 10 instructions (x 2: the weights are split
// into a high and a low 16-bit part so that their rounding does not enter: fp32-accurate taps as in the VALU kernels).
//
// The matrix instruction wants, per lane, 4 consecutive PIXELS of one channel; the tensors are NHWC.  So tiles live in LDS PLANAR
// ([channel][row][x], 2-byte elements): the staging pass transposes on the way in (a thread loads 4 pixels x 8 channels, applies the
// producer's BatchNorm + SiLU, and writes eight 8-byte quads), the output pass transposes back (eight 8-byte reads, v_perm, four
// 16-byte stores).  A tile is a WHOLE PLANE (28^2, 14^2, 7^2: no halo is ever re-staged; the zero padding costs no arithmetic) of 16
// channels x NB images.  16 channels are 32 bytes of a pixel's row: the blocks that share a pixel's cache lines -- the C / 16 channel
// groups of one image tile -- are made consecutive IN ONE XCD (block id -> (xcd, slot): the per-XCD L2 serves the other three
// quarters of each line), so HBM sees every line once.
//
// LDS bank rule for the operand reads (ds_read_b64, lanes = 8 channels x 4 rows per half-wave): row pitch / 8 B odd and channel pitch
// / 8 B = 4 (mod 32) make the 32 eight-byte slots distinct.
#include "common.h"

struct Dw5Geom {
  int B, H, W, C;
  int NB;                  // images per tile
  int RG, XQ;              // 4-row groups per image, 4-pixel quads per row
  int IHt, IWp;            // input-tile rows per image (4 RG + 4) and row pitch in halfs (>= 4 XQ + 4, IWp / 4 odd)
  int CPI;                 // input-tile channel pitch, bytes ((CPI / 8) % 32 == 4)
  int OWp, CPO;            // output-tile row pitch in halfs (4 XQ) and channel pitch in bytes
  int ntiles;              // ceil(B / NB)
};

typedef unsigned long long u64;
__device__ __forceinline__ u64 lds_read_b64(const char* p) { return *reinterpret_cast<const u64*>(p); }
__device__ __forceinline__ void lds_write_b64(char* p, u64 v) { *reinterpret_cast<u64*>(p) = v; }
__device__ __forceinline__ unsigned int pack2h(float a, float b) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const h2 v = {f2h(a), f2h(b)};
  return __builtin_bit_cast(unsigned int, v);
}
__device__ __forceinline__ unsigned int pack2bf(float a, float b) {
  const bf2 v = {f2bf(a), f2bf(b)};
  return __builtin_bit_cast(unsigned int, v);
}
__device__ __forceinline__ f4 mfma44h(h4 a, h4 b, f4 c) { return __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f4 mfma44b(s4 a, s4 b, f4 c) { return __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, c, 0, 0, 0); }

// block id -> (image tile, channel group): ids round-robin over the 8 XCDs (MI355X_MICROARCH.md), so id % 8 picks the XCD and id / 8
// the slot in it; a slot sequence walks the channel groups of one tile before the next tile.  Speed only: any placement is correct.
__device__ __forceinline__ bool dw5_block(const Dw5Geom& g, int& tile, int& cg) {
  const int ncg = g.C >> 4;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tl = slot / ncg;
  cg = slot - tl * ncg;
  tile = tl * 8 + xcd;
  return tile < g.ntiles;
}

// Toeplitz operand pair of one kernel row for lane (channel, i): T0[i][k] = w[k - i] (k >= i), T1[i][k] = w[4 + k - i] (k <= i),
// as (hi, lo) 16-bit splits.  FLIP: the data-gradient convolution (taps reversed).
template <bool BF>
__device__ __forceinline__ void toeplitz_row(const float (&w)[5], int i, unsigned int (&t0h)[2], unsigned int (&t0l)[2], unsigned int (&t1h)[2],
                                             unsigned int (&t1l)[2]) {
  float a0[4], a1[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float v0 = 0.f, v1 = 0.f;
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      if (k - i == t) v0 = w[t];
      if (4 + k - i == t) v1 = w[t];
    }
    a0[k] = v0; a1[k] = v1;
  }
  auto split = [&](const float (&a)[4], unsigned int (&hi)[2], unsigned int (&lo)[2]) {
    float h[4], l[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (BF) { h[k] = bf2f(f2bf(a[k])); l[k] = a[k] - h[k]; }
      else { h[k] = h2f(f2h(a[k])); l[k] = a[k] - h[k]; }
    }
    if (BF) { hi[0] = pack2bf(h[0], h[1]); hi[1] = pack2bf(h[2], h[3]); lo[0] = pack2bf(l[0], l[1]); lo[1] = pack2bf(l[2], l[3]); }
    else { hi[0] = pack2h(h[0], h[1]); hi[1] = pack2h(h[2], h[3]); lo[0] = pack2h(l[0], l[1]); lo[1] = pack2h(l[2], l[3]); }
  };
  split(a0, t0h, t0l);
  split(a1, t1h, t1l);
}

// ------------------------------------------------------------------ forward
// in  z1 [B, H, W, C] fp16 (pre-BatchNorm expansion output; XF: a1 = silu(scale z1 + shift) is formed while staging, padding zero AFTER
//     the activation), wT [25][C] fp32 tap-major weights, out z2 [B, H, W, C] fp16, parts [ntiles][2 C]: per-tile sum / sum of squares
//     of the ROUNDED outputs (the next BatchNorm's statistics; summed by mmsim_launch_reduce).
template <bool XF>
__global__ __launch_bounds__(256) void dw5m_fwd_kernel(const f16* __restrict__ in, const float* __restrict__ scale, const float* __restrict__ shift,
                                                       const float* __restrict__ wT, f16* __restrict__ out, float* __restrict__ parts, Dw5Geom g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int tile, cg;
  if (!dw5_block(g, tile, cg)) return;
  char* tin = smem;
  char* tout = smem + 16 * g.CPI;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int img0 = tile * g.NB, nimg = min(g.NB, g.B - img0);
  const int cbase = cg * 16;
  // ---- matrix-domain lane roles and the Toeplitz weight operands of this lane's channel (5 kernel rows x 2 blocks x (hi, lo))
  const int mb = lane >> 2, mj = lane & 3;
  unsigned int th[5][2][2], tl[5][2][2];
  {
    const float* wc = wT + cbase + mb;
#pragma unroll
    for (int kh = 0; kh < 5; ++kh) {
      float w[5];
#pragma unroll
      for (int kw = 0; kw < 5; ++kw) w[kw] = wc[(size_t)(kh * 5 + kw) * g.C];
      toeplitz_row<false>(w, mj, th[kh][0], tl[kh][0], th[kh][1], tl[kh][1]);
    }
  }
  // ---- stage: NHWC global -> planar LDS (transposing), BatchNorm + SiLU on the way
  {
    const int o = tid & 1;                              // channel octet of the 16
    const int nq = g.IWp >> 2;                          // quads per tile row
    const int nitems = nimg * g.IHt * nq;
    float sc[8], sh[8];
    if (XF) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { sc[e] = scale[cbase + o * 8 + e]; sh[e] = shift[cbase + o * 8 + e]; }
    }
    for (int it = tid >> 1; it < nitems; it += 128) {
      const int n = it / (g.IHt * nq), r = it - n * (g.IHt * nq);
      const int ty = r / nq, tq = r - ty * nq;
      const int y = ty - 2;
      const bool rowok = y >= 0 && y < g.H;
      const f16* rowp = in + (((size_t)(img0 + n) * g.H + (rowok ? y : 0)) * g.W) * g.C + cbase + o * 8;
      uint4 v[4];
      bool ok[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int x = 4 * tq - 2 + p;
        ok[p] = rowok && x >= 0 && x < g.W;
        const uint4 t = *reinterpret_cast<const uint4*>(rowp + (size_t)min(max(x, 0), g.W - 1) * g.C);
        const unsigned int m = ok[p] ? 0xffffffffu : 0u;
        v[p] = make_uint4(t.x & m, t.y & m, t.z & m, t.w & m);
      }
      unsigned int q[8][2];                             // per channel: halfs (px0, px1), (px2, px3)
      if (XF) {
        float f[4][8];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const h8 hv = __builtin_bit_cast(h8, v[p]);
#pragma unroll
          for (int e = 0; e < 8; ++e) f[p][e] = ok[p] ? silu_f(h2f(hv[e]) * sc[e] + sh[e]) : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { q[e][0] = pack2h(f[0][e], f[1][e]); q[e][1] = pack2h(f[2][e], f[3][e]); }
      } else {
        const unsigned int d[4][4] = {{v[0].x, v[0].y, v[0].z, v[0].w}, {v[1].x, v[1].y, v[1].z, v[1].w}, {v[2].x, v[2].y, v[2].z, v[2].w},
                                      {v[3].x, v[3].y, v[3].z, v[3].w}};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const unsigned int sel = (e & 1) ? 0x07060302u : 0x05040100u;       // the high / low halves of two dwords
          q[e][0] = __builtin_amdgcn_perm(d[1][e >> 1], d[0][e >> 1], sel);
          q[e][1] = __builtin_amdgcn_perm(d[3][e >> 1], d[2][e >> 1], sel);
        }
      }
      char* dst = tin + (size_t)(o * 8) * g.CPI + ((size_t)(n * g.IHt + ty) * g.IWp + 4 * tq) * 2;
#pragma unroll
      for (int e = 0; e < 8; ++e) lds_write_b64(dst + (size_t)e * g.CPI, (u64)q[e][0] | ((u64)q[e][1] << 32));
    }
  }
  __syncthreads();
  // ---- matrix phase: item = (image, row group, x quad); a wave-instruction covers the 16 channels
  float s1 = 0.f, s2 = 0.f;
  {
    const int nitems = nimg * g.RG * g.XQ;
    const char* lbase = tin + (size_t)mb * g.CPI + (size_t)mj * g.IWp * 2;
    char* obase = tout + (size_t)mb * g.CPO + (size_t)mj * g.OWp * 2;
    for (int it = wave; it < nitems; it += 4) {
      const int n = it / (g.RG * g.XQ), r = it - n * (g.RG * g.XQ);
      const int rg = r / g.XQ, xq = r - rg * g.XQ;
      const char* src = lbase + ((size_t)(n * g.IHt + 4 * rg) * g.IWp + 4 * xq) * 2;
      u64 b0[5], b1[5];
#pragma unroll
      for (int kh = 0; kh < 5; ++kh) { b0[kh] = lds_read_b64(src + (size_t)kh * g.IWp * 2); b1[kh] = lds_read_b64(src + (size_t)kh * g.IWp * 2 + 8); }
      f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kh = 0; kh < 5; ++kh) {
        const h4 x0 = __builtin_bit_cast(h4, b0[kh]), x1 = __builtin_bit_cast(h4, b1[kh]);
        acc = mfma44h(__builtin_bit_cast(h4, (u64)th[kh][0][0] | ((u64)th[kh][0][1] << 32)), x0, acc);
        acc = mfma44h(__builtin_bit_cast(h4, (u64)tl[kh][0][0] | ((u64)tl[kh][0][1] << 32)), x0, acc);
        acc = mfma44h(__builtin_bit_cast(h4, (u64)th[kh][1][0] | ((u64)th[kh][1][1] << 32)), x1, acc);
        acc = mfma44h(__builtin_bit_cast(h4, (u64)tl[kh][1][0] | ((u64)tl[kh][1][1] << 32)), x1, acc);
      }
      // acc[i] = out[channel mb][row 4 rg + mj][x = 4 xq + i]
      const f16 r0 = f2h(acc[0]), r1 = f2h(acc[1]), r2 = f2h(acc[2]), r3 = f2h(acc[3]);
      typedef _Float16 h2 __attribute__((ext_vector_type(2)));
      const h2 p0 = {r0, r1}, p1 = {r2, r3};
      lds_write_b64(obase + ((size_t)(n * 4 * g.RG + 4 * rg) * g.OWp + 4 * xq) * 2,
                    (u64)__builtin_bit_cast(unsigned int, p0) | ((u64)__builtin_bit_cast(unsigned int, p1) << 32));
      const bool rowok = 4 * rg + mj < g.H;
      const float rr[4] = {h2f(r0), h2f(r1), h2f(r2), h2f(r3)};
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (rowok && 4 * xq + i < g.W) { s1 += rr[i]; s2 += rr[i] * rr[i]; }
    }
  }
  // statistics: sum over the 4 row lanes of a channel, then over the 4 waves
  s1 += dpp_mov_f<0xB1>(s1); s1 += dpp_mov_f<0x4E>(s1);
  s2 += dpp_mov_f<0xB1>(s2); s2 += dpp_mov_f<0x4E>(s2);
  __shared__ float red[4][32];
  if (mj == 0) { red[wave][mb] = s1; red[wave][16 + mb] = s2; }
  __syncthreads();
  if (tid < 32) {
    const float t = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    parts[(size_t)tile * 2 * g.C + (tid >> 4) * g.C + cbase + (tid & 15)] = t;
  }
  // ---- output pass: planar LDS -> NHWC global
  {
    const int o = tid & 1;
    const int nitems = nimg * g.H * g.XQ;
    for (int it = tid >> 1; it < nitems; it += 128) {
      const int n = it / (g.H * g.XQ), r = it - n * (g.H * g.XQ);
      const int y = r / g.XQ, xq = r - y * g.XQ;
      const char* src = tout + (size_t)(o * 8) * g.CPO + ((size_t)(n * 4 * g.RG + y) * g.OWp + 4 * xq) * 2;
      unsigned int d0[8], d1[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { const u64 t = lds_read_b64(src + (size_t)e * g.CPO); d0[e] = (unsigned int)t; d1[e] = (unsigned int)(t >> 32); }
      f16* dst = out + (((size_t)(img0 + n) * g.H + y) * g.W + 4 * xq) * g.C + cbase + o * 8;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        if (4 * xq + p < g.W) {
          const unsigned int sel = (p & 1) ? 0x07060302u : 0x05040100u;
          const unsigned int* d = (p & 2) ? d1 : d0;
          const uint4 ov = make_uint4(__builtin_amdgcn_perm(d[1], d[0], sel), __builtin_amdgcn_perm(d[3], d[2], sel),
                                      __builtin_amdgcn_perm(d[5], d[4], sel), __builtin_amdgcn_perm(d[7], d[6], sel));
          *reinterpret_cast<uint4*>(dst + (size_t)p * g.C) = ov;
        }
      }
    }
  }
}

// ================================================================= host side
void mmsim_launch_reduce(const float* parts, int nparts, int n, float* out, int accumulate, hipStream_t s);   // conv.hip

static int dw5_make_geom(Dw5Geom* g, int B, int H, int W, int C, int nb, size_t extra_planes_bytes_per_pixel, size_t* lds) {
  g->B = B; g->H = H; g->W = W; g->C = C;
  g->RG = (H + 3) / 4; g->XQ = (W + 3) / 4;
  g->IHt = 4 * g->RG + 4;
  g->IWp = 4 * g->XQ + 4;
  if (((g->IWp >> 2) & 1) == 0) g->IWp += 4;                        // row pitch / 8 B odd
  g->OWp = 4 * g->XQ;
  g->NB = nb;
  size_t cpi = (size_t)nb * g->IHt * g->IWp * 2;
  cpi = (cpi + 7) & ~(size_t)7;
  while (((cpi >> 3) & 31) != 4) cpi += 8;                          // channel pitch / 8 B = 4 (mod 32)
  g->CPI = (int)cpi;
  size_t cpo = (size_t)nb * 4 * g->RG * g->OWp * 2;
  cpo = (cpo + 7) & ~(size_t)7;
  while (((cpo >> 3) & 31) != 4) cpo += 8;
  g->CPO = (int)cpo;
  g->ntiles = (B + nb - 1) / nb;
  *lds = 16 * cpi + 16 * cpo + extra_planes_bytes_per_pixel * 16 * (size_t)nb * 4 * g->RG * g->OWp;
  return 0;
}

// images per tile: as many as keep the tile under `budget` bytes of LDS, at most 8, and a divisor-friendly count
static int dw5_pick_nb(int B, int H, int W, size_t budget, size_t extra) {
  int best = 1;
  for (int nb = 1; nb <= 8 && nb <= B; nb *= 2) {
    Dw5Geom g; size_t lds;
    dw5_make_geom(&g, B, H, W, 16, nb, extra, &lds);
    if (lds <= budget) best = nb;
  }
  return best;
}

extern "C" int mmsim_dw5m_eligible(int B, int H, int W, int C, int K, int S) {
  if (K != 5 || S != 1 || B <= 0 || C <= 0 || (C % 16) || H <= 0 || W <= 0 || H > 28 || W > 28) return 0;
  return 1;
}

static void dw5_optin() {
  static unsigned long long done = 0;
  const int dev = mmsim_current_device();
  if ((done >> dev) & 1) return;
  const int cap = 128 * 1024;
  (void)hipFuncSetAttribute((const void*)dw5m_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
  (void)hipFuncSetAttribute((const void*)dw5m_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
  done |= 1ull << dev;
}

// Same contract as mmsim_dwtile_fwd (mbconv.hip) for K = 5, S = 1; scratch >= ceil(B / NB) * 2 C floats (<= B * 2 C).
extern "C" int mmsim_dw5m_fwd(const void* in, const float* xf_scale, const float* xf_shift, const float* w_tap_major, void* z, float* sums,
                              int B, int H, int W, int C, float* scratch, unsigned long long scratch_floats, void* stream) {
  MMSIM_REQUIRE(in && w_tap_major && z && sums && scratch, "dw5m_fwd: null operand");
  MMSIM_REQUIRE((xf_scale == nullptr) == (xf_shift == nullptr), "dw5m_fwd: scale and shift come together");
  MMSIM_REQUIRE(mmsim_dw5m_eligible(B, H, W, C, 5, 1), "dw5m_fwd: shape not eligible (see mmsim_dw5m_eligible)");
  Dw5Geom g; size_t lds;
  const int nb = dw5_pick_nb(B, H, W, 64 * 1024, 0);
  dw5_make_geom(&g, B, H, W, C, nb, 0, &lds);
  MMSIM_REQUIRE(lds <= 128 * 1024, "dw5m_fwd: tile does not fit the LDS");
  MMSIM_REQUIRE(scratch_floats >= (unsigned long long)g.ntiles * 2 * C, "dw5m_fwd: scratch too small");
  dw5_optin();
  const int ncg = C / 16;
  const dim3 grid(8 * ((g.ntiles + 7) / 8) * ncg);
  hipStream_t s = (hipStream_t)stream;
  if (xf_scale) hipLaunchKernelGGL(dw5m_fwd_kernel<true>, grid, dim3(256), lds, s, (const f16*)in, xf_scale, xf_shift, w_tap_major, (f16*)z, scratch, g);
  else hipLaunchKernelGGL(dw5m_fwd_kernel<false>, grid, dim3(256), lds, s, (const f16*)in, xf_scale, xf_shift, w_tap_major, (f16*)z, scratch, g);
  mmsim_launch_reduce(scratch, g.ntiles, 2 * C, sums, 1, s);
  return mmsim_check_launch("dw5m_fwd");
}
