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
#include <stdlib.h>

// DWM_ABL (diagnostic variant builds only, tools/build_variant.sh; 0 in the product): 1 = no matrix instructions, 2 = no BatchNorm /
// SiLU while staging, 4 = no global stores, 8 = no global loads; backward: 16 = no weight-gradient phase, 32 = no data-gradient matrix
// instructions, 64 = no dz2 arithmetic, 128 = weight-gradient operands from ALIGNED addresses (wrong results: timing only)
#ifndef DWM_ABL
#define DWM_ABL 0
#endif

struct DwmGeom {
  int B, H, W, C;
  int NB;                  // images per tile (> 1 only when a band is the whole plane)
  int TRG, NBAND;          // 4-row groups per tile band, bands per image
  int XQ;                  // 4-pixel quads per row
  int IHt, IWp;            // input-tile rows per image (4 TRG + K - 1) and row pitch in halfs (>= 4 XQ + 4, IWp / 4 odd)
  int CPI;                 // input-tile channel pitch, bytes ((CPI / 8) % 32 == 4)
  int OWp, CPO;            // output-tile row pitch in halfs (4 XQ) and channel pitch in bytes
  int ntiles, TPB, ngroups;      // tiles = ceil(B / NB) * NBAND; consecutive tiles per block; blocks per channel group
  FastDiv d_nband, d_ncg, d_xq, d_mitems, d_nq, d_sitems, d_oitems;      // NBAND, C / 16, XQ, TRG * XQ, IWp / 4, IHt * IWp / 4, 4 TRG * XQ
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
__device__ __forceinline__ u64 mk64(unsigned int lo, unsigned int hi) { return (u64)lo | ((u64)hi << 32); }
__device__ __forceinline__ f4 mfma44h(u64 a, u64 b, f4 c) { return __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(h4, a), __builtin_bit_cast(h4, b), c, 0, 0, 0); }
__device__ __forceinline__ f4 mfma44b(u64 a, u64 b, f4 c) { return __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(s4, a), __builtin_bit_cast(s4, b), c, 0, 0, 0); }

// block id -> (tile group, channel group): ids round-robin over the 8 XCDs (MI355X_MICROARCH.md), so id % 8 picks the XCD and id / 8
// the slot in it; a slot sequence walks the channel groups of one tile group before the next group.  Speed only: any placement is correct.
__device__ __forceinline__ bool dwm_block(const DwmGeom& g, int& tgroup, int& cg) {
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  int tl;
  fdivmod((unsigned int)slot, g.d_ncg, tl, cg);
  tgroup = tl * 8 + xcd;
  return tgroup < g.ngroups;
}

// Toeplitz operand pair of one kernel row for lane (channel, i): T0[i][k] = w[k - i] (0 <= k - i < K), T1[i][k] = w[4 + k - i]
// (4 + k - i < K), each as a (hi, lo) pair of 16-bit values whose sum carries the fp32 tap.
template <int K, bool BF>
__device__ __forceinline__ void toeplitz_row(const float (&w)[K], int i, u64& t0h, u64& t0l, u64& t1h, u64& t1l) {
  float a0[4], a1[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float v0 = 0.f, v1 = 0.f;
#pragma unroll
    for (int t = 0; t < K; ++t) {
      if (k - i == t) v0 = w[t];
      if (4 + k - i == t) v1 = w[t];
    }
    a0[k] = v0; a1[k] = v1;
  }
  auto split = [&](const float (&a)[4], u64& hi, u64& lo) {
    float h[4], l[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      h[k] = BF ? bf2f(f2bf(a[k])) : h2f(f2h(a[k]));
      l[k] = a[k] - h[k];
    }
    if (BF) { hi = mk64(pack2bf(h[0], h[1]), pack2bf(h[2], h[3])); lo = mk64(pack2bf(l[0], l[1]), pack2bf(l[2], l[3])); }
    else { hi = mk64(pack2h(h[0], h[1]), pack2h(h[2], h[3])); lo = mk64(pack2h(l[0], l[1]), pack2h(l[2], l[3])); }
  };
  split(a0, t0h, t0l);
  split(a1, t1h, t1l);
}

// tile index -> first image, images in the tile, first output row of the band
__device__ __forceinline__ void dwm_tile(const DwmGeom& g, int t, int& img0, int& nimg, int& y0) {
  int it, band;
  fdivmod((unsigned int)t, g.d_nband, it, band);
  img0 = it * g.NB; nimg = min(g.NB, g.B - img0); y0 = band * 4 * g.TRG;
}

// ------------------------------------------------------------------ forward
// in  z1 [B, H, W, C] fp16 (pre-BatchNorm expansion output; XF: a1 = silu(scale z1 + shift) is formed while staging, padding zero AFTER
//     the activation), wT [K K][C] fp32 tap-major weights, out z2 [B, H, W, C] fp16, parts [ngroups][2 C]: per-block sum / sum of squares
//     of the ROUNDED outputs (the next BatchNorm's statistics; summed by mmsim_launch_reduce).
// Per tile: transform the prefetched rows into the planar input tile | barrier | request the NEXT tile's rows (they fly during the
// rest) | matrix phase -> planar output tile | barrier | output pass.  One staging item (4 pixels x 8 channels) per thread.
// HL: the taps as (high, low) 16-bit pairs (fp32-accurate weights, twice the matrix instructions and 20 more registers) instead of
// plain fp16 taps -- the forward 1x1 convolutions read fp16 weight shadows as well; measured both ways, r04.
template <int K, bool XF, bool HL>
__global__ __launch_bounds__(256, 4) void dwm_fwd_kernel(const f16* __restrict__ in, const float* __restrict__ scale, const float* __restrict__ shift,
                                                      const float* __restrict__ wT, f16* __restrict__ out, float* __restrict__ parts, DwmGeom g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float red[4][32];
  constexpr int PAD = K / 2;
  int tgroup, cg;
  if (!dwm_block(g, tgroup, cg)) return;
  char* tin = smem;
  char* tout = smem + 16 * g.CPI;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cbase = cg * 16;
  const int t0 = tgroup * g.TPB, t1 = min(g.ntiles, t0 + g.TPB);
  // ---- matrix-domain lane roles and the Toeplitz weight operands of this lane's channel (K kernel rows x 2 blocks x (hi, lo))
  const int mb = lane >> 2, mj = lane & 3;
  u64 th0[K], tl0[K], th1[K], tl1[K];
  {
    const float* wc = wT + cbase + mb;
#pragma unroll
    for (int kh = 0; kh < K; ++kh) {
      float w[K];
#pragma unroll
      for (int kw = 0; kw < K; ++kw) w[kw] = wc[(size_t)(kh * K + kw) * g.C];
      toeplitz_row<K, false>(w, mj, th0[kh], tl0[kh], th1[kh], tl1[kh]);
    }
  }
  // ---- staging role: (image n, tile row ty, tile quad tq) x channel octet o, the same for every tile
  const int so = tid & 1, sit = tid >> 1;
  const int nq = g.IWp >> 2;
  int sn, sr, sty, stq;
  fdivmod((unsigned int)sit, g.d_sitems, sn, sr);
  fdivmod((unsigned int)sr, g.d_nq, sty, stq);
  const bool s_on = sit < g.NB * g.IHt * nq;
  // the producer BatchNorm's scale | shift of the 16 channels: kept in LDS, read at the top of each transform (16 registers less
  // across the matrix phase, where the Toeplitz operands and the prefetched rows are live)
  __shared__ __attribute__((aligned(16))) float xfc[2][16];
  if (XF && tid < 32) xfc[tid >> 4][tid & 15] = (tid < 16 ? scale : shift)[cbase + (tid & 15)];
  uint4 v[4];
  unsigned int okm = 0;
  auto request = [&](int t) {
    int img0, nimg, y0;
    dwm_tile(g, t, img0, nimg, y0);
    const int y = y0 + sty - PAD;
    const bool rowok = s_on && sn < nimg && y >= 0 && y < g.H;
    const f16* rowp = in + (((size_t)(img0 + (sn < nimg ? sn : 0)) * g.H + (rowok ? y : 0)) * g.W) * g.C + cbase + so * 8;
    okm = 0;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int x = 4 * stq - PAD + p;
      const bool ok = rowok && x >= 0 && x < g.W;
      okm |= ok ? (1u << p) : 0u;
      if (DWM_ABL & 8) v[p] = make_uint4(0x3c003c00u + p, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
      else v[p] = *reinterpret_cast<const uint4*>(rowp + (size_t)min(max(x, 0), g.W - 1) * g.C);
    }
  };
  // ---- output role: (image n, row y of the band, quad xq) x octet
  const int orow = 4 * g.TRG;
  int on, orr, oy, oxq;
  fdivmod((unsigned int)sit, g.d_oitems, on, orr);
  fdivmod((unsigned int)orr, g.d_xq, oy, oxq);
  const bool o_on = sit < g.NB * orow * g.XQ;
  // ---- matrix role: this wave's items (wave, wave + 4, ...: at most four, the host checks NB TRG XQ <= 16), the same for every tile
  unsigned int mi_sd[4], mi_fl[4];        // packed: LDS offsets (source | destination << 16); image | row offset << 8 | x mask << 16
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int it = wave + 4 * m;
    int n, r, rg, xq;
    fdivmod((unsigned int)it, g.d_mitems, n, r);
    fdivmod((unsigned int)r, g.d_xq, rg, xq);
    const bool on_ = it < g.NB * g.TRG * g.XQ;
    const unsigned int src = mb * g.CPI + mj * g.IWp * 2 + ((n * g.IHt + 4 * rg) * g.IWp + 4 * xq) * 2;
    const unsigned int dst = mb * g.CPO + mj * g.OWp * 2 + ((n * orow + 4 * rg) * g.OWp + 4 * xq) * 2;
    mi_sd[m] = src | (dst << 16);                        // both tiles are < 64 KiB (host check)
    const int left = g.W - 4 * xq;
    mi_fl[m] = (on_ ? (unsigned int)n : 0xffu) | ((unsigned int)(4 * rg + mj) << 8) | ((left >= 4 ? 15u : ((1u << left) - 1u)) << 16);
  }
  float s1 = 0.f, s2 = 0.f;
  if (t0 < t1) request(t0);
  __syncthreads();                                       // xfc is visible
  for (int t = t0; t < t1; ++t) {
    int img0, nimg, y0;
    dwm_tile(g, t, img0, nimg, y0);
    // ---- transform + transpose the prefetched pixels into the planar tile (zero outside the image, AFTER the activation)
    if (s_on) {
      char* dst = tin + (so * 8) * g.CPI + ((sn * g.IHt + sty) * g.IWp + 4 * stq) * 2;
      if (XF && !(DWM_ABL & 2)) {
#pragma unroll
        for (int ce = 0; ce < 2; ++ce) {               // four channels at a time: a quarter of the working set live at once
          const float4 a = *reinterpret_cast<const float4*>(&xfc[0][so * 8 + ce * 4]), b = *reinterpret_cast<const float4*>(&xfc[1][so * 8 + ce * 4]);
          const float sc[4] = {a.x, a.y, a.z, a.w}, sh[4] = {b.x, b.y, b.z, b.w};
          unsigned int q[4][2];
#pragma unroll
          for (int hp = 0; hp < 2; ++hp) {
            float f[2][4];
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
              const int p = 2 * hp + pp;
              const unsigned int w0 = ce ? v[p].z : v[p].x, w1 = ce ? v[p].w : v[p].y;
              typedef _Float16 h2 __attribute__((ext_vector_type(2)));
              const h2 x0 = __builtin_bit_cast(h2, w0), x1 = __builtin_bit_cast(h2, w1);
              if ((okm >> p) & 1) {       // a real branch: padding pays no transcendentals
                f[pp][0] = silu_f((float)x0[0] * sc[0] + sh[0]); f[pp][1] = silu_f((float)x0[1] * sc[1] + sh[1]);
                f[pp][2] = silu_f((float)x1[0] * sc[2] + sh[2]); f[pp][3] = silu_f((float)x1[1] * sc[3] + sh[3]);
              } else {
                f[pp][0] = f[pp][1] = f[pp][2] = f[pp][3] = 0.f;
              }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) q[e][hp] = pack2h(f[0][e], f[1][e]);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) lds_write_b64(dst + (ce * 4 + e) * g.CPI, mk64(q[e][0], q[e][1]));
        }
      } else {
        unsigned int d[4][4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const unsigned int m = ((okm >> p) & 1) ? 0xffffffffu : 0u;
          d[p][0] = v[p].x & m; d[p][1] = v[p].y & m; d[p][2] = v[p].z & m; d[p][3] = v[p].w & m;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const unsigned int sel = (e & 1) ? 0x07060302u : 0x05040100u;       // the high / low halves of two dwords
          lds_write_b64(dst + e * g.CPI, mk64(__builtin_amdgcn_perm(d[1][e >> 1], d[0][e >> 1], sel), __builtin_amdgcn_perm(d[3][e >> 1], d[2][e >> 1], sel)));
        }
      }
    }
    __syncthreads();
    if (t + 1 < t1) request(t + 1);                     // in flight during the matrix phase and the output pass
    // ---- matrix phase: item = (image, row group, x quad), up to four per wave (decoded once, outside the tile loop); a
    // wave-instruction covers the 16 channels
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if ((int)(mi_fl[m] & 0xffu) < nimg) {              // wave-uniform (also false for the unused item slots)
        const char* src = tin + (mi_sd[m] & 0xffffu);
        u64 b0[K], b1[K];
#pragma unroll
        for (int kh = 0; kh < K; ++kh) { b0[kh] = lds_read_b64(src + kh * g.IWp * 2); b1[kh] = lds_read_b64(src + kh * g.IWp * 2 + 8); }
        f4 acc = {0.f, 0.f, 0.f, 0.f}, accl = {0.f, 0.f, 0.f, 0.f};      // two chains: the high and the low parts of the taps
        if (DWM_ABL & 1) {
          acc[0] = __builtin_bit_cast(float, (unsigned int)b0[0]); acc[1] = __builtin_bit_cast(float, (unsigned int)b1[K - 1]);
        } else {
#pragma unroll
          for (int kh = 0; kh < K; ++kh) {
            acc = mfma44h(th0[kh], b0[kh], acc);
            if (HL) accl = mfma44h(tl0[kh], b0[kh], accl);
            acc = mfma44h(th1[kh], b1[kh], acc);
            if (HL) accl = mfma44h(tl1[kh], b1[kh], accl);
          }
          if (HL) acc += accl;
        }
        // acc[i] = out[channel mb][row y0 + 4 rg + mj][x = 4 xq + i]
        const unsigned int p0 = pack2h(acc[0], acc[1]), p1 = pack2h(acc[2], acc[3]);
        lds_write_b64(tout + (mi_sd[m] >> 16), mk64(p0, p1));
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const h2 q0 = __builtin_bit_cast(h2, p0), q1 = __builtin_bit_cast(h2, p1);
        const float rr[4] = {(float)q0[0], (float)q0[1], (float)q1[0], (float)q1[1]};
        const unsigned int xm = mi_fl[m] >> 16;
        if (y0 + (int)((mi_fl[m] >> 8) & 0xffu) < g.H) {   // per lane: its image row
          if (xm == 15) {                                // wave-uniform: an interior quad
#pragma unroll
            for (int i = 0; i < 4; ++i) { s1 += rr[i]; s2 += rr[i] * rr[i]; }
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if ((xm >> i) & 1) { s1 += rr[i]; s2 += rr[i] * rr[i]; }
          }
        }
      }
    }
    __syncthreads();
    // ---- output pass: planar LDS -> NHWC global
    if (o_on && on < nimg && y0 + oy < g.H) {
      const char* src = tout + (so * 8) * g.CPO + ((on * orow + oy) * g.OWp + 4 * oxq) * 2;
      unsigned int d0[8], d1[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { const u64 tt = lds_read_b64(src + e * g.CPO); d0[e] = (unsigned int)tt; d1[e] = (unsigned int)(tt >> 32); }
      f16* dst = out + (((size_t)(img0 + on) * g.H + y0 + oy) * g.W + 4 * oxq) * g.C + cbase + so * 8;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        if (4 * oxq + p < g.W) {
          const unsigned int sel = (p & 1) ? 0x07060302u : 0x05040100u;
          const unsigned int* d = (p & 2) ? d1 : d0;
          const uint4 ov = make_uint4(__builtin_amdgcn_perm(d[1], d[0], sel), __builtin_amdgcn_perm(d[3], d[2], sel),
                                      __builtin_amdgcn_perm(d[5], d[4], sel), __builtin_amdgcn_perm(d[7], d[6], sel));
          if (!(DWM_ABL & 4) || ov.x == 0x12345678u) *reinterpret_cast<uint4*>(dst + (size_t)p * g.C) = ov;
        }
      }
    }
    // (the next tile's transform writes the input tile, last read before the barrier above; its matrix phase -- behind the next
    //  barrier -- is what overwrites the output tile this pass reads)
  }
  // statistics: sum over the 4 row lanes of a channel, then over the 4 waves
  s1 += dpp_mov_f<0xB1>(s1); s1 += dpp_mov_f<0x4E>(s1);
  s2 += dpp_mov_f<0xB1>(s2); s2 += dpp_mov_f<0x4E>(s2);
  if (mj == 0) { red[wave][mb] = s1; red[wave][16 + mb] = s2; }
  __syncthreads();
  if (tid < 32) {
    const float tsum = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    parts[(size_t)tgroup * 2 * g.C + (tid >> 4) * g.C + cbase + (tid & 15)] = tsum;
  }
}

// ------------------------------------------------------------------ fused backward (stride 1, IR block)
// The whole depthwise backward of one MBConv block, same contract as dwt_bwd_kernel<K, false> (mbconv.hip):
//   dz2  = depthwise-BatchNorm + SiLU + squeeze-excite-gate backward of (dy, z2), formed while the halo'd tile is staged (planar, bf16);
//   da1  = dz2 (*) flipped taps              (matrix cores, the forward's Toeplitz form with the taps reversed, bf16);
//   dpre = da1 silu'(bn1(z1)) -> HBM (bf16), with the expand BatchNorm's backward sums; a1 = silu(bn1(z1)) -> LDS (planar, bf16);
//   dW[kh][kw] = sum a1[y][x] dz2[y + P - kh][x + P - kw]    (matrix cores: for a dz2 row R and an x quad, A[i = kw][k] = the dz2 row
//              shifted by kw -- a 2-byte-aligned 8-byte LDS read per lane --, B[k][j = kh] = a1 rows R - 2 P + kh: D[kw][kh] += A B;
//              taps 0..3 and tap 4 are separate 4-blocks: 4 instructions per (row, quad), 25 of their 64 results used).
// Per tile: stage dz2 and z1 | barrier | request the next tile | data gradient + elementwise (matrix-domain lanes own a channel: their
// BatchNorm constants are 4 registers) | barrier | weight gradient + output pass | barrier.
struct DwmBwd {
  const bf16* dy; const f16* z2; const f16* z1;
  const float* sc2; const float* sh2; const float* mu2; const float* rs2; const float* sums2;     // depthwise BatchNorm
  const float* gate; const float* dsq;                                                              // [B, C] fp32
  const float* sc1; const float* sh1; const float* mu1; const float* rs1;                          // expand BatchNorm
  const float* wT; bf16* out; float* parts_bn; float* parts_w; float* dgamma2; float* dbeta2;
  float invP, inv_hw;
};

struct __attribute__((packed, aligned(2))) U64u { u64 v; };      // an 8-byte LDS read at a 2-byte-aligned address (gfx950: one ds_read_b64)

#ifndef DWM_BWD_WAVES
#define DWM_BWD_WAVES 3
#endif
template <int K>
__global__ __launch_bounds__(256, DWM_BWD_WAVES) void dwm_bwd_kernel(DwmBwd p, DwmGeom g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ __attribute__((aligned(16))) float cst[4][16];          // sc2 | sh2 | A | Bc of the 16 channels (dz2 = (dy gs + qs) silu'(sc2 z + sh2) - A - z Bc)
  constexpr int PAD = K / 2;
  int tgroup, cg;
  if (!dwm_block(g, tgroup, cg)) return;
  char* tdz = smem;                                   // dz2, halo'd tile, planar bf16 [16][NB][IHt][IWp]
  char* tz = smem + 16 * g.CPI;                       // z1 (fp16) -> a1 (bf16), centre tile [16][NB][4 TRG][OWp]
  char* tda = tz + 16 * g.CPO;                        // dpre (bf16), centre tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cbase = cg * 16;
  const int t0 = tgroup * g.TPB, t1 = min(g.ntiles, t0 + g.TPB);
  const int mb = lane >> 2, mj = lane & 3;
  const int orow = 4 * g.TRG;
  // ---- block constants
  if (tid < 16) {
    const int c = cbase + tid;
    const float sc = p.sc2[c], bc = sc * p.rs2[c] * p.sums2[g.C + c] * p.invP;
    cst[0][tid] = sc; cst[1][tid] = p.sh2[c]; cst[2][tid] = sc * p.sums2[c] * p.invP - p.mu2[c] * bc; cst[3][tid] = bc;
    if (tgroup == 0) { p.dgamma2[c] += p.sums2[g.C + c]; p.dbeta2[c] += p.sums2[c]; }      // the depthwise BatchNorm's own parameter gradients
  }
  // the flipped Toeplitz operands (data gradient), bf16
  u64 tw0[K], tw1[K];
  {
    const float* wc = p.wT + cbase + mb;
#pragma unroll
    for (int kh = 0; kh < K; ++kh) {
      float w[K];
#pragma unroll
      for (int kw = 0; kw < K; ++kw) w[kw] = wc[(size_t)((K - 1 - kh) * K + (K - 1 - kw)) * g.C];
      u64 l0, l1;
      toeplitz_row<K, true>(w, mj, tw0[kh], l0, tw1[kh], l1);
    }
  }
  const float c_sc1 = p.sc1[cbase + mb], c_sh1 = p.sh1[cbase + mb], c_mu1 = p.mu1[cbase + mb], c_rs1 = p.rs1[cbase + mb];
  // ---- staging roles (as the forward kernel): halo'd tile item and centre item of this thread pair
  const int so = tid & 1, sit = tid >> 1;
  const int nq = g.IWp >> 2;
  int sn, sr, sty, stq;
  fdivmod((unsigned int)sit, g.d_sitems, sn, sr);
  fdivmod((unsigned int)sr, g.d_nq, sty, stq);
  const bool s_on = sit < g.NB * g.IHt * nq;
  int on, orr, oy, oxq;
  fdivmod((unsigned int)sit, g.d_oitems, on, orr);
  fdivmod((unsigned int)orr, g.d_xq, oy, oxq);
  const bool o_on = sit < g.NB * orow * g.XQ;
  // ---- matrix role: the data-gradient items of this wave (as the forward kernel)
  unsigned int mi_sd[4], mi_fl[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int it = wave + 4 * m;
    int n, r, rg, xq;
    fdivmod((unsigned int)it, g.d_mitems, n, r);
    fdivmod((unsigned int)r, g.d_xq, rg, xq);
    const bool on_ = it < g.NB * g.TRG * g.XQ;
    const unsigned int src = mb * g.CPI + mj * g.IWp * 2 + ((n * g.IHt + 4 * rg) * g.IWp + 4 * xq) * 2;
    const unsigned int dst = mb * g.CPO + mj * g.OWp * 2 + ((n * orow + 4 * rg) * g.OWp + 4 * xq) * 2;
    mi_sd[m] = src | (dst << 16);
    const int left = g.W - 4 * xq;
    mi_fl[m] = (on_ ? (unsigned int)n : 0xffu) | ((unsigned int)(4 * rg + mj) << 8) | ((left >= 4 ? 15u : ((1u << left) - 1u)) << 16);
  }
  uint4 vd[4], vz[4], v1[4];
  unsigned int okm = 0, okc = 0;
  float4 vg[2], vq[2];                                  // gate / dsq of the staged image, this thread's channel octet
  auto request = [&](int t) {
    int img0, nimg, y0;
    dwm_tile(g, t, img0, nimg, y0);
    const int y = y0 + sty - PAD;
    const bool rowok = s_on && sn < nimg && y >= 0 && y < g.H;
    const size_t roff = (((size_t)(img0 + (sn < nimg ? sn : 0)) * g.H + (rowok ? y : 0)) * g.W) * g.C + cbase + so * 8;
    okm = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int x = 4 * stq - PAD + q;
      const bool ok = rowok && x >= 0 && x < g.W;
      okm |= ok ? (1u << q) : 0u;
      const size_t off = roff + (size_t)min(max(x, 0), g.W - 1) * g.C;
      vd[q] = *reinterpret_cast<const uint4*>(p.dy + off);
      vz[q] = *reinterpret_cast<const uint4*>(p.z2 + off);
    }
    {
      const size_t o = (size_t)(img0 + (sn < nimg ? sn : 0)) * g.C + cbase + so * 8;
      vg[0] = *reinterpret_cast<const float4*>(p.gate + o); vg[1] = *reinterpret_cast<const float4*>(p.gate + o + 4);
      vq[0] = *reinterpret_cast<const float4*>(p.dsq + o); vq[1] = *reinterpret_cast<const float4*>(p.dsq + o + 4);
    }
    {      // z1 rows of the centre tile
      const int rows_valid = min(orow, g.H - y0);
      const bool crow = o_on && on < nimg && oy < rows_valid;
      const f16* rowp = p.z1 + (((size_t)(img0 + (on < nimg ? on : 0)) * g.H + (crow ? y0 + oy : 0)) * g.W) * g.C + cbase + so * 8;
      okc = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int x = 4 * oxq + q;
        const bool ok = crow && x < g.W;
        okc |= ok ? (1u << q) : 0u;
        v1[q] = *reinterpret_cast<const uint4*>(rowp + (size_t)min(x, g.W - 1) * g.C);
      }
    }
  };
  f4 dw00 = {0.f, 0.f, 0.f, 0.f}, dw01 = dw00, dw10 = dw00, dw11 = dw00;
  float s1 = 0.f, s2 = 0.f;
  __syncthreads();                                      // cst
  for (int t = t0; t < t1; ++t) {
    int img0, nimg, y0;
    dwm_tile(g, t, img0, nimg, y0);
    const int rows_valid = min(orow, g.H - y0);
    // no register prefetch across tiles here (it cost 48 registers through the matrix phases and the forward kernel measured the
    // same with and without): the tile's rows are requested now, the other resident blocks cover the round trip
    request(t);
    // ---- z1 centre rows (prefetched with the tile) -> planar LDS first: their registers are free for the dz2 arithmetic
    if (o_on) {
      char* dst = tz + (so * 8) * g.CPO + ((on * orow + oy) * g.OWp + 4 * oxq) * 2;
      unsigned int d[4][4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const unsigned int m = ((okc >> q) & 1) ? 0xffffffffu : 0u;
        d[q][0] = v1[q].x & m; d[q][1] = v1[q].y & m; d[q][2] = v1[q].z & m; d[q][3] = v1[q].w & m;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const unsigned int sel = (e & 1) ? 0x07060302u : 0x05040100u;
        lds_write_b64(dst + e * g.CPO, mk64(__builtin_amdgcn_perm(d[1][e >> 1], d[0][e >> 1], sel), __builtin_amdgcn_perm(d[3][e >> 1], d[2][e >> 1], sel)));
      }
    }
    // ---- dz2 of the halo'd tile -> planar LDS, four channels at a time
    if (s_on) {
      char* dst = tdz + (so * 8) * g.CPI + ((sn * g.IHt + sty) * g.IWp + 4 * stq) * 2;
#pragma unroll
      for (int ce = 0; ce < 2; ++ce) {
        const int c4 = so * 8 + ce * 4;
        const float4 a0 = *reinterpret_cast<const float4*>(&cst[0][c4]), a1 = *reinterpret_cast<const float4*>(&cst[1][c4]);
        const float4 a2 = *reinterpret_cast<const float4*>(&cst[2][c4]), a3 = *reinterpret_cast<const float4*>(&cst[3][c4]);
        const float4 a4 = vg[ce], a5 = vq[ce];
        const float sc2[4] = {a0.x, a0.y, a0.z, a0.w}, sh2[4] = {a1.x, a1.y, a1.z, a1.w}, A[4] = {a2.x, a2.y, a2.z, a2.w};
        const float Bc[4] = {a3.x, a3.y, a3.z, a3.w};
        const float gs[4] = {a4.x * sc2[0], a4.y * sc2[1], a4.z * sc2[2], a4.w * sc2[3]};
        const float qs[4] = {a5.x * p.inv_hw * sc2[0], a5.y * p.inv_hw * sc2[1], a5.z * p.inv_hw * sc2[2], a5.w * p.inv_hw * sc2[3]};
        unsigned int q[4][2];
#pragma unroll
        for (int hp = 0; hp < 2; ++hp) {
          float f[2][4];
#pragma unroll
          for (int pp = 0; pp < 2; ++pp) {
            const int px = 2 * hp + pp;
            if (((okm >> px) & 1) && !(DWM_ABL & 64)) {       // a real branch: padding pays no transcendentals
              const unsigned int d0 = ce ? vd[px].z : vd[px].x, d1 = ce ? vd[px].w : vd[px].y;
              const unsigned int z0 = ce ? vz[px].z : vz[px].x, z1w = ce ? vz[px].w : vz[px].y;
              typedef _Float16 h2 __attribute__((ext_vector_type(2)));
              const h2 zz0 = __builtin_bit_cast(h2, z0), zz1 = __builtin_bit_cast(h2, z1w);
              const float d[4] = {__builtin_bit_cast(float, d0 << 16), __builtin_bit_cast(float, d0 & 0xffff0000u),
                                  __builtin_bit_cast(float, d1 << 16), __builtin_bit_cast(float, d1 & 0xffff0000u)};
              const float z[4] = {(float)zz0[0], (float)zz0[1], (float)zz1[0], (float)zz1[1]};
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float da = (d[e] * gs[e] + qs[e]) * silu_grad_f(z[e] * sc2[e] + sh2[e]);
                f[pp][e] = da - A[e] - z[e] * Bc[e];
              }
            } else {
              f[pp][0] = f[pp][1] = f[pp][2] = f[pp][3] = 0.f;
            }
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) q[e][hp] = pack2bf(f[0][e], f[1][e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) lds_write_b64(dst + (ce * 4 + e) * g.CPI, mk64(q[e][0], q[e][1]));
      }
    }
    __syncthreads();
    // ---- data gradient + elementwise, matrix domain: lane = (channel mb, row mj of the group), 4 pixels
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if ((int)(mi_fl[m] & 0xffu) < nimg) {
        const char* src = tdz + (mi_sd[m] & 0xffffu);
        u64 b0[K], b1[K];
#pragma unroll
        for (int kh = 0; kh < K; ++kh) { b0[kh] = lds_read_b64(src + kh * g.IWp * 2); b1[kh] = lds_read_b64(src + kh * g.IWp * 2 + 8); }
        const unsigned int doff = mi_sd[m] >> 16;
        const u64 zq = lds_read_b64(tz + doff);
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        if (DWM_ABL & 32) { acc[0] = __builtin_bit_cast(float, (unsigned int)b0[0]); acc[1] = __builtin_bit_cast(float, (unsigned int)b1[K - 1]); }
        else {
#pragma unroll
          for (int kh = 0; kh < K; ++kh) { acc = mfma44b(tw0[kh], b0[kh], acc); acc = mfma44b(tw1[kh], b1[kh], acc); }
        }
        const h4 zh = __builtin_bit_cast(h4, zq);
        const unsigned int xm = mi_fl[m] >> 16;
        const bool rowok = (int)((mi_fl[m] >> 8) & 0xffu) < rows_valid;
        float a1v[4], ov[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float z = (float)zh[i];
          const float uu = z * c_sc1 + c_sh1, sg = sigmoid_f(uu);
          const bool ok = rowok && ((xm >> i) & 1);
          a1v[i] = ok ? h2f(f2h(uu * sg)) : 0.f;            // the fp16 a1 the forward convolved
          ov[i] = acc[i] * sg * (1.0f + uu * (1.0f - sg));
          const float orr_ = bf2f(f2bf(ov[i]));
          if (ok) { s1 += orr_; s2 += orr_ * (z - c_mu1) * c_rs1; }
        }
        lds_write_b64(tz + doff, mk64(pack2bf(a1v[0], a1v[1]), pack2bf(a1v[2], a1v[3])));
        lds_write_b64(tda + doff, mk64(pack2bf(ov[0], ov[1]), pack2bf(ov[2], ov[3])));
      }
    }
    __syncthreads();
    // ---- weight gradient: wave w takes the dz2 rows w, w + 4, ... of the tile (all images), every x quad
    if (!(DWM_ABL & 16)) {
      const int nrows = nimg * g.IHt;
      int n = 0, rt = wave;                            // row -> (image, tile row), advanced incrementally
      while (rt >= g.IHt) { rt -= g.IHt; ++n; }
      for (int row = wave; row < nrows; row += 4) {
        const int yimg = y0 + rt - PAD;                // the dz2 row's image row: outside the image the tile row is zero
        if (yimg >= 0 && yimg < g.H) {
          const int yr0 = rt - (K - 1) + mj, yr1 = rt - (K - 1) + 4 + mj;      // a1 rows (band-relative) of kh = mj and kh = 4 + mj
          const bool v0 = yr0 >= 0 && yr0 < rows_valid, v1b = mj == 0 && yr1 >= 0 && yr1 < rows_valid;
          const u64 m0 = v0 ? ~0ull : 0ull, m1 = v1b ? ~0ull : 0ull;
          const char* arow0 = tz + mb * g.CPO + ((n * orow + min(max(yr0, 0), orow - 1)) * g.OWp) * 2;
          const char* arow1 = tz + mb * g.CPO + ((n * orow + min(max(yr1, 0), orow - 1)) * g.OWp) * 2;
          const char* drow = tdz + mb * g.CPI + ((n * g.IHt + rt) * g.IWp) * 2;
          for (int q0 = 0; q0 < g.XQ; q0 += 4) {        // four quads per trip: their sixteen LDS reads are in flight together
            u64 A0[4], A1[4], B0[4], B1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int q = min(q0 + u, g.XQ - 1);
              const u64 live = q0 + u < g.XQ ? ~0ull : 0ull;
              // the dz2 row's halfs [4 q, 4 q + 8) as two ALIGNED 8-byte reads; the kw-shifted window [4 q + 4 - mj, + 4) is cut out of
              // them in registers (a 2-byte-aligned ds_read_b64 works on gfx950 but measured 3 x the time of this whole phase)
              const u64 lo8 = lds_read_b64(drow + (4 * q) * 2), hi8 = lds_read_b64(drow + (4 * q + 4) * 2);
              if (DWM_ABL & 128) { A0[u] = hi8; }
              else {
                const unsigned int d0 = (unsigned int)lo8, d1 = (unsigned int)(lo8 >> 32), d2 = (unsigned int)hi8, d3 = (unsigned int)(hi8 >> 32);
                const unsigned int e0 = mj == 0 ? d2 : (mj == 3 ? d0 : d1), e1 = mj == 0 ? d3 : (mj == 3 ? d1 : d2), e2 = mj == 3 ? d2 : d3;
                const unsigned int sel = (mj & 1) ? 0x05040302u : 0x03020100u;      // odd shift: the upper half of one dword | the lower half of the next
                A0[u] = mk64(__builtin_amdgcn_perm(e1, e0, sel), __builtin_amdgcn_perm(e2, e1, sel));      // kw = mj
              }
              A1[u] = lo8;                                                           // kw = 4 + mj: only mj = 0 is a tap, its window is [4 q, + 4)
              B0[u] = lds_read_b64(arow0 + 8 * q) & m0 & live;
              B1[u] = lds_read_b64(arow1 + 8 * q) & m1 & live;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              dw00 = mfma44b(A0[u], B0[u], dw00);
              dw01 = mfma44b(A0[u], B1[u], dw01);
              dw10 = mfma44b(A1[u], B0[u], dw10);
              dw11 = mfma44b(A1[u], B1[u], dw11);
            }
          }
        }
        rt += 4;
        while (rt >= g.IHt) { rt -= g.IHt; ++n; }
      }
    }
    // ---- output pass: dpre planar -> NHWC global (bf16)
    if (o_on && on < nimg && oy < rows_valid) {
      const char* src = tda + (so * 8) * g.CPO + ((on * orow + oy) * g.OWp + 4 * oxq) * 2;
      unsigned int d0[8], d1[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { const u64 tt = lds_read_b64(src + e * g.CPO); d0[e] = (unsigned int)tt; d1[e] = (unsigned int)(tt >> 32); }
      bf16* dst = p.out + (((size_t)(img0 + on) * g.H + y0 + oy) * g.W + 4 * oxq) * g.C + cbase + so * 8;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (4 * oxq + q < g.W) {
          const unsigned int sel = (q & 1) ? 0x07060302u : 0x05040100u;
          const unsigned int* d = (q & 2) ? d1 : d0;
          *reinterpret_cast<uint4*>(dst + (size_t)q * g.C) = make_uint4(__builtin_amdgcn_perm(d[1], d[0], sel), __builtin_amdgcn_perm(d[3], d[2], sel),
                                                                        __builtin_amdgcn_perm(d[5], d[4], sel), __builtin_amdgcn_perm(d[7], d[6], sel));
        }
      }
    }
    __syncthreads();
  }
  // ---- per-block partial sums: the expand BatchNorm's backward sums and the K x K weight gradient of the 16 channels
  s1 += dpp_mov_f<0xB1>(s1); s1 += dpp_mov_f<0x4E>(s1);
  s2 += dpp_mov_f<0xB1>(s2); s2 += dpp_mov_f<0x4E>(s2);
  float* red = reinterpret_cast<float*>(smem);          // [wave][2 + K K][16]; the tiles are dead (barrier above)
  constexpr int NV = 2 + K * K;
  float* mine = red + (wave * NV) * 16 + mb;
  if (mj == 0) { mine[0] = s1; mine[16] = s2; }
  // dw00[i] at lane (mb, j): dW[kh = j][kw = i]; dw01[i] (j = 0): dW[4][i]; dw10[0] at lane j: dW[j][4]; dw11[0] (j = 0): dW[4][4]
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (mj < K && i < K) mine[(2 + mj * K + i) * 16] = dw00[i];
    if (K > 4 && mj == 0 && i < K) mine[(2 + 4 * K + i) * 16] = dw01[i];
  }
  if (K > 4 && mj < K) mine[(2 + mj * K + 4) * 16] = dw10[0];
  if (K > 4 && mj == 0) mine[(2 + 4 * K + 4) * 16] = dw11[0];
  __syncthreads();
  for (int i = tid; i < NV * 16; i += 256) {
    const float tsum = (red[i] + red[NV * 16 + i]) + (red[2 * NV * 16 + i] + red[3 * NV * 16 + i]);
    const int v = i >> 4, c = cbase + (i & 15);
    if (v < 2) p.parts_bn[(size_t)tgroup * 2 * g.C + (size_t)v * g.C + c] = tsum;
    else p.parts_w[(size_t)tgroup * K * K * g.C + (size_t)(v - 2) * g.C + c] = tsum;
  }
}

// ================================================================= host side
void mmsim_launch_reduce(const float* parts, int nparts, int n, float* out, int accumulate, hipStream_t s);   // conv.hip

static size_t dwm_pitch8(size_t bytes) {          // round a channel pitch up so that (pitch / 8) % 32 == 4
  bytes = (bytes + 7) & ~(size_t)7;
  while (((bytes >> 3) & 31) != 4) bytes += 8;
  return bytes;
}

// Geometry for kernel size K: band height (row groups) and images per tile chosen so that one staging pass of 128 thread pairs
// covers the input tile (<= 128 items of 4 pixels x 8 channels) and the output tile (<= 128 items); false if no choice fits.
static bool dwm_make_geom(DwmGeom* g, int B, int H, int W, int C, int K, size_t* lds) {
  g->B = B; g->H = H; g->W = W; g->C = C;
  const int RG = (H + 3) / 4;
  g->XQ = (W + 3) / 4;
  g->IWp = 4 * g->XQ + 4;
  if (((g->IWp >> 2) & 1) == 0) g->IWp += 4;                        // row pitch / 8 B odd
  g->OWp = 4 * g->XQ;
  const int nq = g->IWp / 4;
  int trg = 0;
  for (int t = RG; t >= 1; --t)                                     // the tallest band whose halo'd tile one pass covers
    if ((4 * t + K - 1) * nq <= 128 && 4 * t * g->XQ <= 128 && t * g->XQ <= 16) { trg = t; break; }
  if (!trg) return false;
  g->TRG = trg; g->NBAND = (RG + trg - 1) / trg;
  g->IHt = 4 * trg + K - 1;
  int nb = 1;
  if (g->NBAND == 1) while (nb < 8 && (nb + 1) * g->IHt * nq <= 128 && (nb + 1) * 4 * trg * g->XQ <= 128 && (nb + 1) * trg * g->XQ <= 16 && nb + 1 <= B) ++nb;
  g->NB = nb;
  g->CPI = (int)dwm_pitch8((size_t)nb * g->IHt * g->IWp * 2);
  g->CPO = (int)dwm_pitch8((size_t)nb * 4 * trg * g->OWp * 2);
  g->ntiles = ((B + nb - 1) / nb) * g->NBAND;
  // blocks per channel group: ~6 000 blocks per launch (256 CUs x 4-6 resident blocks x a few rounds), whole images per block
  const int ncg = C / 16;
  static int want_blocks = -1;      // MMSIM_DWM_BLOCKS: blocks per launch (tuning)
  if (want_blocks < 0) { const char* e = getenv("MMSIM_DWM_BLOCKS"); want_blocks = e ? atoi(e) : 4096; }
  int want = want_blocks / ncg; if (want < 8) want = 8;
  int tpb = (g->ntiles + want - 1) / want;
  tpb = ((tpb + g->NBAND - 1) / g->NBAND) * g->NBAND;              // a block walks all bands of its images: the halo rows stay in L2
  g->TPB = tpb; g->ngroups = (g->ntiles + tpb - 1) / tpb;
  g->d_nband = make_fastdiv(g->NBAND); g->d_ncg = make_fastdiv(ncg); g->d_xq = make_fastdiv(g->XQ); g->d_mitems = make_fastdiv(trg * g->XQ);
  g->d_nq = make_fastdiv(nq); g->d_sitems = make_fastdiv(g->IHt * nq); g->d_oitems = make_fastdiv(4 * trg * g->XQ);
  *lds = 16 * (size_t)g->CPI + 16 * (size_t)g->CPO;
  return true;
}

extern "C" int mmsim_dw5m_eligible(int B, int H, int W, int C, int K, int S) {
  if (K != 5 || S != 1 || B <= 0 || C <= 0 || (C % 16) || H <= 0 || W <= 0) return 0;
  DwmGeom g; size_t lds;
  return dwm_make_geom(&g, B, H, W, C, K, &lds) && lds <= 64 * 1024 ? 1 : 0;
}

// Same contract as mmsim_dwtile_fwd (mbconv.hip) for K = 5, S = 1; scratch >= B * 2 C floats.
extern "C" int mmsim_dw5m_fwd(const void* in, const float* xf_scale, const float* xf_shift, const float* w_tap_major, void* z, float* sums,
                              int B, int H, int W, int C, float* scratch, unsigned long long scratch_floats, void* stream) {
  MMSIM_REQUIRE(in && w_tap_major && z && sums && scratch, "dw5m_fwd: null operand");
  MMSIM_REQUIRE((xf_scale == nullptr) == (xf_shift == nullptr), "dw5m_fwd: scale and shift come together");
  MMSIM_REQUIRE(mmsim_dw5m_eligible(B, H, W, C, 5, 1), "dw5m_fwd: shape not eligible (see mmsim_dw5m_eligible)");
  DwmGeom g; size_t lds;
  dwm_make_geom(&g, B, H, W, C, 5, &lds);
  MMSIM_REQUIRE(scratch_floats >= (unsigned long long)g.ngroups * 2 * C, "dw5m_fwd: scratch too small");
  const int ncg = C / 16;
  const dim3 grid(8 * ((g.ngroups + 7) / 8) * ncg);
  hipStream_t s = (hipStream_t)stream;
  static int hl = -1;       // MMSIM_DWM_HL=1: (high, low) tap pairs
  if (hl < 0) { const char* e = getenv("MMSIM_DWM_HL"); hl = e ? atoi(e) : 0; }
#define DWM_F(XX, HH) hipLaunchKernelGGL((dwm_fwd_kernel<5, XX, HH>), grid, dim3(256), lds, s, (const f16*)in, xf_scale, xf_shift, w_tap_major, (f16*)z, scratch, g)
  if (xf_scale) { if (hl) DWM_F(true, true); else DWM_F(true, false); }
  else { if (hl) DWM_F(false, true); else DWM_F(false, false); }
#undef DWM_F
  mmsim_launch_reduce(scratch, g.ngroups, 2 * C, sums, 1, s);
  return mmsim_check_launch("dw5m_fwd");
}

void mmsim_launch_reduce2(const float* pa, int na, float* oa, const float* pb, int nb, float* ob, int nparts, hipStream_t s);      // conv.hip

// Same contract as mmsim_dwtile_bwd (mbconv.hip) for K = 5, S = 1, IR blocks (the expand BatchNorm state is required); scratch >= B * 27 C floats.
extern "C" int mmsim_dw5m_bwd(const void* dy, const void* z2, const float* scale2, const float* shift2, const float* mean2, const float* rstd2,
                              const float* sums2, const float* gate, const float* dsq, const void* z1, const float* scale1, const float* shift1,
                              const float* mean1, const float* rstd1, const float* w_tap_major, void* out, float* sums1, float* g_tap_major,
                              float* dgamma2, float* dbeta2, int B, int H, int W, int C, float* scratch, unsigned long long scratch_floats,
                              void* stream) {
  MMSIM_REQUIRE(dy && z2 && scale2 && shift2 && mean2 && rstd2 && sums2 && gate && dsq && z1 && scale1 && shift1 && mean1 && rstd1 && w_tap_major &&
                    out && sums1 && g_tap_major && dgamma2 && dbeta2 && scratch, "dw5m_bwd: null operand");
  MMSIM_REQUIRE(mmsim_dw5m_eligible(B, H, W, C, 5, 1), "dw5m_bwd: shape not eligible (see mmsim_dw5m_eligible)");
  DwmGeom g; size_t lds;
  dwm_make_geom(&g, B, H, W, C, 5, &lds);
  lds += 16 * (size_t)g.CPO;                           // the dpre tile
  const size_t red = 4 * 27 * 16 * sizeof(float);
  if (lds < red) lds = red;
  MMSIM_REQUIRE(lds <= 64 * 1024, "dw5m_bwd: tile does not fit");
  const size_t n_bn = (size_t)g.ngroups * 2 * C, n_w = (size_t)g.ngroups * 25 * C;
  MMSIM_REQUIRE(scratch_floats >= (unsigned long long)(n_bn + n_w), "dw5m_bwd: scratch too small");
  DwmBwd p;
  p.dy = (const bf16*)dy; p.z2 = (const f16*)z2; p.z1 = (const f16*)z1;
  p.sc2 = scale2; p.sh2 = shift2; p.mu2 = mean2; p.rs2 = rstd2; p.sums2 = sums2; p.gate = gate; p.dsq = dsq;
  p.sc1 = scale1; p.sh1 = shift1; p.mu1 = mean1; p.rs1 = rstd1; p.wT = w_tap_major; p.out = (bf16*)out;
  p.parts_bn = scratch; p.parts_w = scratch + n_bn; p.dgamma2 = dgamma2; p.dbeta2 = dbeta2;
  p.invP = 1.0f / (float)((size_t)B * H * W); p.inv_hw = 1.0f / (float)(H * W);
  const int ncg = C / 16;
  const dim3 grid(8 * ((g.ngroups + 7) / 8) * ncg);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL((dwm_bwd_kernel<5>), grid, dim3(256), lds, s, p, g);
  mmsim_launch_reduce2(p.parts_bn, 2 * C, sums1, p.parts_w, 25 * C, g_tap_major, g.ngroups, s);
  return mmsim_check_launch("dw5m_bwd");
}
