// LDS-tiled depthwise-convolution kernels of the MBConv block for gfx950 (NHWC activations, fp32 arithmetic).
// Element types as in conv.hip: FORWARD tensors (z1, z2, a2, z3, x and the 1x1-conv weight shadow the forward products read) are
// fp16, GRADIENT tensors (dy, dz, dx, resid) and the weight shadow the data-gradient products read are bf16.
//
// The round-1 depthwise kernels (conv.hip) read every input row straight from global memory once per kernel row: a 5x5
// output strip issues 40 16-byte loads for 4 outputs and the vector-memory path, not HBM, set their rate (1.6-2.8 TB/s).
// Here a workgroup stages an input tile WITH ITS HALO in LDS once (each global byte is requested ~1.3-1.6 times instead
// of K times) and every tap is an LDS read.  Staging through LDS also makes it free to TRANSFORM the operand on the way in,
// which removes whole passes over the widest tensors of the tower:
//   dwt_fwd   in = z1 (pre-BatchNorm expand output): a1 = silu(scale z1 + shift) is formed while the tile is staged, so the
//             activated tensor a1 is never written to / re-read from HBM (was: bn_apply pass + its output);
//             out = z2 (+ per-channel sum / sum of squares of the rounded outputs: the next BatchNorm's statistics).
//   dwt_bwd   (stride 1) ONE kernel for the whole depthwise backward: the tile staged in LDS is dz2, the gradient w.r.t. the
//             depthwise output, computed on the way in from (dy, z2) = the BatchNorm + SiLU + squeeze-excite-gate backward
//             (was a separate 3-tensor pass, bn_bwd_apply); from it both the data gradient (x silu'(bn(z1)) -> dpre1, with the
//             expand BatchNorm's backward sums) and the weight gradient (x a1, recomputed from z1) are formed in one sweep:
//             for centre pixel q and tap k, t = dz2[q + PAD - k]:  da1[q] += t w[k],  dW[k] += a1[q] t.
//             (was: bn_bwd_apply + dwconv_bwd_weight + dwconv_bwd_data = 8 passes over [pixels, mid] tensors; now 4.)
// Thread = (channel unit u, pixel lane pl): a unit is CPT = 8 (16-byte accesses) or 4 (8-byte) consecutive channels, a
// workgroup owns OG units x a spatial tile, consecutive lanes hold consecutive units of one pixel (contiguous bytes).
// Per-channel reductions (BN statistics, BN-backward sums, weight gradients) stay in registers over all the tiles a block
// walks and leave as ONE partial-slab row per block (summed by reduce_partials: no atomics, bitwise reproducible).
#include "common.h"

#ifndef DWT_ROW_UNPACK
#define DWT_ROW_UNPACK 1     // 1: a staged row is unpacked to fp32 ONCE and feeds all its taps (the late, small-plane 5x5 layers are
#endif                       //    VALU-bound: unpacking at every use doubled their instruction count); 0: unpack at every use

struct DwTile {
  int B, Hi, Wi, Ho, Wo, C;
  int TH, TW, IH, IW;                 // output tile; staged input tile (with halo)
  int total_tiles, tiles_per_block;   // B * tiles per image; consecutive tiles walked by one block
  int OG, NP, nstrips;                // channel units per block, pixel lanes (256 / OG), TW / 4
  int gx, gy;                         // logical grid: tile ranges x channel groups (see dwt_block_id)
  FastDiv d_tiles_img, d_tx, d_iw, d_strips;
};

// 1-D launch of gx x gy logical blocks (tile range bx, channel group by).  With a 2-D grid the channel groups of one tile range ran
// half a kernel apart, and when a group's bytes per pixel are not whole cache lines (C = 144: two groups of 144 B in 288-B pixels)
// every line was fetched from HBM twice (profiles/r04_pmc_image_tower.json: dwt_fwd<3,2> 1.79 x its algorithmic bytes).  Here the
// groups of a tile range are CONSECUTIVE IN ONE XCD (block id % 8 = XCD, id / 8 = its launch slot): the second reader hits that L2.
#ifndef DWT_GRID_2D
#define DWT_GRID_2D 0
#endif
__device__ __forceinline__ bool dwt_block_id(const DwTile& g, int& bx, int& by) {
#if DWT_GRID_2D          // A/B builds only (tools/build_variant.sh): the order of the former 2-D grid, x fastest
  by = blockIdx.x / g.gx; bx = blockIdx.x % g.gx;
  return by < g.gy;
#endif
  const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
  by = j % g.gy;
  bx = (j / g.gy) * 8 + xcd;
  return bx < g.gx;
}
static dim3 dwt_grid(DwTile* g, int gx, int gy) {
  g->gx = gx; g->gy = gy;
  return dim3(((gx + 7) / 8) * 8 * gy);
}

template <int CPT> struct UnitT;
template <> struct UnitT<8> { typedef uint4 T; };
template <> struct UnitT<4> { typedef uint2 T; };

template <int CPT> __device__ __forceinline__ void unpackN(const typename UnitT<CPT>::T& v, float (&f)[CPT]);
template <> __device__ __forceinline__ void unpackN<8>(const uint4& v, float (&f)[8]) {
  const bf8 b = __builtin_bit_cast(bf8, v);
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = bf2f(b[e]);
}
template <> __device__ __forceinline__ void unpackN<4>(const uint2& v, float (&f)[4]) {
  const bf4 b = __builtin_bit_cast(bf4, v);
#pragma unroll
  for (int e = 0; e < 4; ++e) f[e] = bf2f(b[e]);
}
template <int CPT> __device__ __forceinline__ typename UnitT<CPT>::T packN(const float (&f)[CPT]);
template <> __device__ __forceinline__ uint4 packN<8>(const float (&f)[8]) {
  bf8 b;
#pragma unroll
  for (int e = 0; e < 8; ++e) b[e] = f2bf(f[e]);
  return __builtin_bit_cast(uint4, b);
}
template <> __device__ __forceinline__ uint2 packN<4>(const float (&f)[4]) {
  bf4 b;
#pragma unroll
  for (int e = 0; e < 4; ++e) b[e] = f2bf(f[e]);
  return __builtin_bit_cast(uint2, b);
}
// fp16 chunks (forward tensors)
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
// bounds-masked load: the address is always valid (callers clamp), the value is AND-masked (a select of a load makes hipcc
// branch around it and wait vmcnt(0) per element, conv.hip)
template <int CPT> __device__ __forceinline__ typename UnitT<CPT>::T ldN_masked(const void* p, bool ok);
template <> __device__ __forceinline__ uint4 ldN_masked<8>(const void* p, bool ok) {
  const uint4 v = *reinterpret_cast<const uint4*>(p);
  const unsigned int m = ok ? 0xffffffffu : 0u;
  return make_uint4(v.x & m, v.y & m, v.z & m, v.w & m);
}
template <> __device__ __forceinline__ uint2 ldN_masked<4>(const void* p, bool ok) {
  const uint2 v = *reinterpret_cast<const uint2*>(p);
  const unsigned int m = ok ? 0xffffffffu : 0u;
  return make_uint2(v.x & m, v.y & m);
}
template <int CPT> __device__ __forceinline__ void ldNf(const float* p, float (&f)[CPT]) {
#pragma unroll
  for (int e = 0; e < CPT; e += 4) {
    const float4 a = *reinterpret_cast<const float4*>(p + e);
    f[e] = a.x; f[e + 1] = a.y; f[e + 2] = a.z; f[e + 3] = a.w;
  }
}
__device__ __forceinline__ int clampi2(int v, int lo, int hi) { return min(max(v, lo), hi); }

// Sum acc[NV] over the pixel lanes of a block (threads pl*OG + u, same u) through `red` (>= 256*CH floats), CH values per
// round; the pl == 0 thread of each unit ends up with the totals.
template <int NV, int CH>
__device__ __forceinline__ void lanes_reduce(float (&acc)[NV], float* red, int u, int pl, int OG, int NP) {
  static_assert(NV % CH == 0, "chunking");
#pragma unroll
  for (int c = 0; c < NV; c += CH) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < CH; ++i) red[threadIdx.x * CH + i] = acc[c + i];
    __syncthreads();
    if (pl == 0) {
      for (int r = 1; r < NP; ++r)
#pragma unroll
        for (int i = 0; i < CH; ++i) acc[c + i] += red[(r * OG + u) * CH + i];
    }
  }
}

// ------------------------------------------------------------------ forward
template <int K, int S, bool XF>
__global__ __launch_bounds__(256, 3) void dwt_fwd_kernel(const f16* in, const float* scale, const float* shift, const float* wT,
                                                         f16* out, float* parts, DwTile g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int PAD = K / 2, NIN = 3 * S + K;
  const int tid = threadIdx.x;
  int bx, by;
  if (!dwt_block_id(g, bx, by)) return;
  const int u = tid % g.OG, pl = tid / g.OG;
  const bool lane_ok = pl < g.NP;
  const int unit = by * g.OG + u;
  const bool cok = lane_ok && unit * 8 < g.C;
  const int c0 = cok ? unit * 8 : 0;
  const int npix = g.IH * g.IW;
  uint4* tile = reinterpret_cast<uint4*>(smem);
  float* wl = reinterpret_cast<float*>(smem + (size_t)npix * g.OG * 16);       // [tap][OG][8] weights of this channel group
  for (int i = tid; i < K * K * g.OG * 8; i += 256) {
    const int tap = i / (g.OG * 8), r = i - tap * (g.OG * 8);
    const int c = by * g.OG * 8 + r;
    wl[i] = c < g.C ? wT[(size_t)tap * g.C + c] : 0.f;
  }
  float st[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) st[i] = 0.f;
  const int t0 = bx * g.tiles_per_block, t1 = min(g.total_tiles, t0 + g.tiles_per_block);
  for (int t = t0; t < t1; ++t) {
    int b, ti, ty, tx;
    fdivmod((unsigned int)t, g.d_tiles_img, b, ti);
    fdivmod((unsigned int)ti, g.d_tx, ty, tx);
    const int oy0 = ty * g.TH, ox0 = tx * g.TW;
    const int iy0 = oy0 * S - PAD, ix0 = ox0 * S - PAD;
    __syncthreads();                      // the previous tile's taps have been read (first trip: the weights are staged)
    if (lane_ok) {
      float sc[8], sh[8];                 // per-phase loads: the channel vectors do not occupy registers across the tap loop
      if (XF) { ldNf<8>(scale + c0, sc); ldNf<8>(shift + c0, sh); }
      const f16* inb = in + (size_t)b * g.Hi * g.Wi * g.C + c0;
      for (int i0 = pl; i0 < npix; i0 += 4 * g.NP) {       // four pixels per trip: all loads requested before the first use
        uint4 v[4];
        bool ok[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = min(i0 + q * g.NP, npix - 1);
          int iy, ix;
          fdivmod((unsigned int)i, g.d_iw, iy, ix);
          const int hy = iy0 + iy, wx = ix0 + ix;
          ok[q] = cok && hy >= 0 && hy < g.Hi && wx >= 0 && wx < g.Wi;
          v[q] = ldN_masked<8>(inb + ((size_t)clampi2(hy, 0, g.Hi - 1) * g.Wi + clampi2(wx, 0, g.Wi - 1)) * g.C, ok[q]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = i0 + q * g.NP;
          if (i < npix) {
            if (XF) {       // a1 = silu(bn(z1)), staged as fp16; padding is zero AFTER the activation
              // a real branch, not a select: under a 5 x 5 window 40-60 % of a 14^2 / 7^2 tile's staged pixels are padding, and a
              // wave whose pixels are all outside the image skips the transcendentals altogether
              if (ok[q]) {
                float f[8];
                unpack8h(v[q], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = silu_f(f[e] * sc[e] + sh[e]);
                v[q] = pack8h(f);
              } else {
                v[q] = make_uint4(0, 0, 0, 0);
              }
            }
            tile[(size_t)i * g.OG + u] = v[q];
          }
        }
      }
    }
    __syncthreads();
    if (lane_ok) {
      const int nitems = g.TH * g.nstrips;
      for (int it = pl; it < nitems; it += g.NP) {
        int oy, sx;
        fdivmod((unsigned int)it, g.d_strips, oy, sx);
        float acc[4][8];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[j][e] = 0.f;
#pragma unroll 1
        for (int kh = 0; kh < K; ++kh) {
          const uint4* row = tile + ((size_t)(oy * S + kh) * g.IW + sx * 4 * S) * g.OG + u;
#if DWT_ROW_UNPACK
          float xin[NIN][8];
#pragma unroll
          for (int x = 0; x < NIN; ++x) unpack8h(row[(size_t)x * g.OG], xin[x]);
#pragma unroll
          for (int kw = 0; kw < K; ++kw) {
            float w[8];
            ldNf<8>(wl + ((kh * K + kw) * g.OG + u) * 8, w);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
              for (int e = 0; e < 8; ++e) acc[j][e] += xin[j * S + kw][e] * w[e];
          }
#else
          uint4 raw[NIN];
#pragma unroll
          for (int x = 0; x < NIN; ++x) raw[x] = row[(size_t)x * g.OG];
#pragma unroll
          for (int kw = 0; kw < K; ++kw) {
            float w[8];
            ldNf<8>(wl + ((kh * K + kw) * g.OG + u) * 8, w);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float xin[8];
              unpack8h(raw[j * S + kw], xin);
#pragma unroll
              for (int e = 0; e < 8; ++e) acc[j][e] += xin[e] * w[e];
            }
          }
#endif
        }
        const int gy = oy0 + oy;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int gx = ox0 + sx * 4 + j;
          if (cok && gy < g.Ho && gx < g.Wo) {
            const uint4 o = pack8h(acc[j]);
            *reinterpret_cast<uint4*>(out + (((size_t)b * g.Ho + gy) * g.Wo + gx) * g.C + c0) = o;
            float r[8];
            unpack8h(o, r);
#pragma unroll
            for (int e = 0; e < 8; ++e) { st[e] += r[e]; st[8 + e] += r[e] * r[e]; }
          }
        }
      }
    }
  }
  lanes_reduce<16, 16>(st, reinterpret_cast<float*>(smem), u, pl, g.OG, g.NP);
  if (cok && pl == 0) {
    float* slab = parts + (size_t)bx * 2 * g.C + c0;
#pragma unroll
    for (int e = 0; e < 8; ++e) { slab[e] = st[e]; slab[g.C + e] = st[8 + e]; }
  }
}

// ------------------------------------------------------------------ fused backward (stride 1)
struct DwBwd {
  const bf16* dy; const f16* z2; const f16* z1; const bf16* resid;      // z1: the depthwise conv's input (z1 of an IR block, x of a DS block)
  const float* sc2; const float* sh2; const float* mu2; const float* rs2; const float* sums2;     // depthwise BatchNorm
  const float* gate; const float* dsq;                                                              // [B, C] fp32
  const float* sc1; const float* sh1; const float* mu1; const float* rs1;                          // expand BatchNorm (IR blocks)
  const float* wT; bf16* out; float* parts_bn; float* parts_w; float* dgamma2; float* dbeta2;
  float invP, inv_hw;
  int RG;            // row groups of the weight-gradient phase: 256 / (OG * K)
};

// Round 3, measured and not kept (tools/bench_dwt.py, alternating builds on one box; SQ counters of this kernel, tools/pmc_dwt.sh:
// 31-41 % of a wave's cycles issue VALU instructions, 32-52 % are SQ_WAIT_ANY): the stage phase with two pixel groups in flight and raw
// (unmasked) loads, and the data phase's z1 rows requested before the tap loop with their first use tied behind it -- 6.72 against
// 6.70-6.79 ms over the stride-1 blocks' backward, 3 x 3 forms slightly slower, 5 x 5 slightly faster, the 5 x 5 form with the expand
// BatchNorm's backward spills (it sits at 256 registers).  The waits of this kernel are its three barriers per tile with unevenly
// loaded phases, not exposed load latency.
// PLAIN = the depthwise conv's input was the block input itself (DS block of stage 0): a1 = x read as is, no BN + SiLU
// backward on the data gradient, optional residual gradient added.
// Per tile three phases, each with its own thread mapping and only the registers it needs:
//   stage   thread (u, pixel lane): dz2 of the halo tile -> LDS (the depthwise BatchNorm + SiLU + gate backward on the way in)
//   data    thread (u, strip of 4 centre pixels): da1 = sum_k dz2[q + PAD - k] w[k]; a1 = silu(bn(z1)) -> LDS (centre tile);
//           dpre1 = da1 silu'(bn(z1)) -> HBM, with the expand BatchNorm's backward sums
//   weight  thread (u, kernel row kh, row group): dW[kh][kw] += a1[q] dz2[q + PAD - k] over its rows (K x 8 accumulators)
// S = 2 (round 4; the first block of stages 2, 3, 4, 6 -- until now bn_bwd_apply + dwconv_bwd_weight(_xf) + dwconv_bwd_data, the
// round-1 kernels: 8 passes over [pixels, mid] tensors, 2.0 ms per step): the same three phases with the CENTRE tile in INPUT space
// (TH x TW input pixels, TH even, TW a multiple of 4, so tile origins are even) and the dz2 halo tile in OUTPUT space
// ((TH/2 + 2) x (TW/2 + 2) output pixels from (oy0/2 - 1, ox0/2 - 1)).  Input pixel q meets output pixel o through tap k iff
// q + PAD - k = 2 o: per dimension one tap (q + PAD even... ) or two/three -- the row parity is a runtime test per item, the column
// parity is a compile-time one (strips start at multiples of 4), so the unrolled tap loops keep only the (K*K)/4 live taps per pixel.
template <int K, bool PLAIN, int S = 1>
__global__ __launch_bounds__(256, 2) void dwt_bwd_kernel(DwBwd p, DwTile g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int SW = 4, NQ = 4;                    // centre pixels per strip; staged pixels per trip
  constexpr int NIN = S == 1 ? SW + K - 1 : 4, PAD = K / 2;
  const int tid = threadIdx.x;
  int bx, by;
  if (!dwt_block_id(g, bx, by)) return;
  const int u = tid % g.OG, pl = tid / g.OG;
  const bool lane_ok = pl < g.NP;
  const int unit = by * g.OG + u;
  const bool cok = lane_ok && unit * 8 < g.C;
  const int c0 = cok ? unit * 8 : 0;
  const int npix = g.IH * g.IW, ncen = g.TH * g.TW;
  uint4* tile = reinterpret_cast<uint4*>(smem);                                            // dz2, halo tile [IH*IW][OG]
  uint4* cen = tile + (size_t)npix * g.OG;                                                  // a1, centre tile [TH*TW][OG]
  float* wl = reinterpret_cast<float*>(smem + (size_t)(npix + ncen) * g.OG * 16);          // [tap][OG][8]
  for (int i = tid; i < K * K * g.OG * 8; i += 256) {
    const int tap = i / (g.OG * 8), r = i - tap * (g.OG * 8);
    const int c = by * g.OG * 8 + r;
    wl[i] = c < g.C ? p.wT[(size_t)tap * g.C + c] : 0.f;
  }
  if (bx == 0 && cok && pl == 0) {            // dgamma += sum da zhat, dbeta += sum da (the BN-backward sums themselves)
#pragma unroll
    for (int e = 0; e < 8; ++e) { p.dgamma2[c0 + e] += p.sums2[g.C + c0 + e]; p.dbeta2[c0 + e] += p.sums2[c0 + e]; }
  }
  // weight-phase mapping: thread -> (unit wu, kernel row wkh, row group wrg)
  const int wu = tid % g.OG, wrest = tid / g.OG;
  const int wkh = wrest % K, wrg = wrest / K;
  const bool w_ok = wrg < p.RG && (by * g.OG + wu) * 8 < g.C;
  float st[16], dW[K * 8];
#pragma unroll
  for (int i = 0; i < 16; ++i) st[i] = 0.f;
#pragma unroll
  for (int i = 0; i < K * 8; ++i) dW[i] = 0.f;
  const int t0 = bx * g.tiles_per_block, t1 = min(g.total_tiles, t0 + g.tiles_per_block);
  for (int t = t0; t < t1; ++t) {
    int b, ti, ty, tx;
    fdivmod((unsigned int)t, g.d_tiles_img, b, ti);
    fdivmod((unsigned int)ti, g.d_tx, ty, tx);
    const int oy0 = ty * g.TH, ox0 = tx * g.TW;
    const int iy0 = S == 1 ? oy0 - PAD : (oy0 >> 1) - 1, ix0 = S == 1 ? ox0 - PAD : (ox0 >> 1) - 1;
    __syncthreads();
    // ---- stage: dz2 = scale (da - S1/P - zhat S2/P),  da = (dy gate + dsq/HW) silu'(scale z2 + shift)
    if (lane_ok) {
      // folded per-channel constants: dz2 = (dy gs + qs) silu'(sc2 z + sh2) - A - z Bc   with  gs = gate sc2, qs = dsq/HW sc2,
      // Bc = sc2 rstd S2/P, A = sc2 S1/P - mean Bc
      float sc2[8], sh2[8], gs[8], qs[8], A[8], Bc[8];
      {
        float mu2[8], rs2[8], m1[8], m2[8];
        ldNf<8>(p.sc2 + c0, sc2); ldNf<8>(p.sh2 + c0, sh2); ldNf<8>(p.mu2 + c0, mu2); ldNf<8>(p.rs2 + c0, rs2);
        ldNf<8>(p.sums2 + c0, m1); ldNf<8>(p.sums2 + g.C + c0, m2);
        ldNf<8>(p.gate + (size_t)b * g.C + c0, gs); ldNf<8>(p.dsq + (size_t)b * g.C + c0, qs);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          gs[e] *= sc2[e]; qs[e] *= p.inv_hw * sc2[e];
          Bc[e] = sc2[e] * rs2[e] * m2[e] * p.invP;
          A[e] = sc2[e] * m1[e] * p.invP - mu2[e] * Bc[e];
        }
      }
      const size_t img = (size_t)b * g.Ho * g.Wo * g.C + c0;
      auto request = [&](int i0, uint4 (&vd)[NQ], uint4 (&vz)[NQ], bool (&ok)[NQ]) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int i = min(i0 + q * g.NP, npix - 1);
          int iy, ix;
          fdivmod((unsigned int)i, g.d_iw, iy, ix);
          const int hy = iy0 + iy, wx = ix0 + ix;
          ok[q] = cok && hy >= 0 && hy < g.Ho && wx >= 0 && wx < g.Wo;
          const size_t off = img + ((size_t)clampi2(hy, 0, g.Ho - 1) * g.Wo + clampi2(wx, 0, g.Wo - 1)) * g.C;
          vd[q] = ldN_masked<8>(p.dy + off, ok[q]);
          vz[q] = ldN_masked<8>(p.z2 + off, ok[q]);
        }
      };
      uint4 vd[NQ], vz[NQ];
      bool ok[NQ];
      for (int i0 = pl; i0 < npix; i0 += NQ * g.NP) {
        request(i0, vd, vz, ok);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int i = i0 + q * g.NP;
          if (i < npix) {
            uint4 ov = make_uint4(0, 0, 0, 0);
            if (ok[q]) {      // a real branch: waves whose pixels are all padding skip the arithmetic (see dwt_fwd_kernel)
              float d[8], z[8], o[8];
              unpackN<8>(vd[q], d); unpack8h(vz[q], z);
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const float da = (d[e] * gs[e] + qs[e]) * silu_grad_f(z[e] * sc2[e] + sh2[e]);
                o[e] = da - A[e] - z[e] * Bc[e];
              }
              ov = packN<8>(o);
            }
            tile[(size_t)i * g.OG + u] = ov;
          }
        }
      }
    }
    __syncthreads();
    // ---- data gradient
    if (lane_ok) {
      float sc1[8], sh1[8], mu1[8], rs1[8];
      if (!PLAIN) { ldNf<8>(p.sc1 + c0, sc1); ldNf<8>(p.sh1 + c0, sh1); ldNf<8>(p.mu1 + c0, mu1); ldNf<8>(p.rs1 + c0, rs1); }
      const int nitems = g.TH * g.nstrips;
      for (int it = pl; it < nitems; it += g.NP) {
        int oy, sx;
        fdivmod((unsigned int)it, g.d_strips, oy, sx);
        const int gy = oy0 + oy, gx0 = ox0 + sx * SW;
        const bool rowok = cok && gy < g.Hi;
        const size_t off0 = (((size_t)b * g.Hi + min(gy, g.Hi - 1)) * g.Wi) * g.C + c0;
        uint4 zr[SW], rr[SW];
        bool pok[SW];
#pragma unroll
        for (int j = 0; j < SW; ++j) {
          pok[j] = rowok && gx0 + j < g.Wi;
          zr[j] = ldN_masked<8>(p.z1 + off0 + (size_t)min(gx0 + j, g.Wi - 1) * g.C, pok[j]);
          if (PLAIN && p.resid) rr[j] = ldN_masked<8>(p.resid + off0 + (size_t)min(gx0 + j, g.Wi - 1) * g.C, pok[j]);
        }
        float da[SW][8];
#pragma unroll
        for (int j = 0; j < SW; ++j)
#pragma unroll
          for (int e = 0; e < 8; ++e) da[j][e] = 0.f;
#pragma unroll 1
        for (int kh = 0; kh < K; ++kh) {
          if constexpr (S == 2) {
            const int par = oy + PAD - kh;                    // = 2 (output row) - 2 (tile's first output row + 1)
            if (par & 1) continue;
            const uint4* row = tile + ((size_t)((par >> 1) + 1) * g.IW + sx * (SW / 2)) * g.OG + u;
            uint4 raw[NIN];
#pragma unroll
            for (int x = 0; x < NIN; ++x) raw[x] = row[(size_t)x * g.OG];
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
              float w[8];
              ldNf<8>(wl + ((kh * K + kw) * g.OG + u) * 8, w);
#pragma unroll
              for (int j = 0; j < SW; ++j) {
                if ((j + PAD - kw) & 1) continue;             // compile time
                float tv[8];
                unpackN<8>(raw[((j + PAD - kw) >> 1) + 1], tv);
#pragma unroll
                for (int e = 0; e < 8; ++e) da[j][e] += tv[e] * w[e];
              }
            }
            continue;
          }
          const uint4* row = tile + ((size_t)(oy + K - 1 - kh) * g.IW + sx * SW) * g.OG + u;
#if DWT_ROW_UNPACK
          float tt[NIN][8];
#pragma unroll
          for (int x = 0; x < NIN; ++x) unpackN<8>(row[(size_t)x * g.OG], tt[x]);
#pragma unroll
          for (int kw = 0; kw < K; ++kw) {
            float w[8];
            ldNf<8>(wl + ((kh * K + kw) * g.OG + u) * 8, w);
#pragma unroll
            for (int j = 0; j < SW; ++j)
#pragma unroll
              for (int e = 0; e < 8; ++e) da[j][e] += tt[j + K - 1 - kw][e] * w[e];
          }
#else
          uint4 raw[NIN];
#pragma unroll
          for (int x = 0; x < NIN; ++x) raw[x] = row[(size_t)x * g.OG];
#pragma unroll
          for (int kw = 0; kw < K; ++kw) {
            float w[8];
            ldNf<8>(wl + ((kh * K + kw) * g.OG + u) * 8, w);
#pragma unroll
            for (int j = 0; j < SW; ++j) {
              float tv[8];
              unpackN<8>(raw[j + K - 1 - kw], tv);
#pragma unroll
              for (int e = 0; e < 8; ++e) da[j][e] += tv[e] * w[e];
            }
          }
#endif
        }
#pragma unroll
        for (int j = 0; j < SW; ++j) {
          float z[8], a1[8], o[8];
          unpack8h(zr[j], z);
          if (PLAIN) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { a1[e] = z[e]; o[e] = da[j][e]; }          // masked load: zero outside the image
            if (p.resid) {
              float r[8];
              unpackN<8>(rr[j], r);
#pragma unroll
              for (int e = 0; e < 8; ++e) o[e] += r[e];
            }
            if (pok[j]) *reinterpret_cast<uint4*>(p.out + off0 + (size_t)(gx0 + j) * g.C) = packN<8>(o);
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float uu = z[e] * sc1[e] + sh1[e], sg = sigmoid_f(uu);
              a1[e] = pok[j] ? uu * sg : 0.f;
              o[e] = da[j][e] * sg * (1.0f + uu * (1.0f - sg));
            }
            const uint4 pk = packN<8>(o);
            if (pok[j]) {
              *reinterpret_cast<uint4*>(p.out + off0 + (size_t)(gx0 + j) * g.C) = pk;
              unpackN<8>(pk, o);
#pragma unroll
              for (int e = 0; e < 8; ++e) { st[e] += o[e]; st[8 + e] += o[e] * (z[e] - mu1[e]) * rs1[e]; }
            }
          }
          cen[(size_t)(oy * g.TW + sx * SW + j) * g.OG + u] = pack8h(a1);       // the rounded (fp16) a1 the forward convolved
        }
      }
    }
    __syncthreads();
    // ---- weight gradient
    if (S == 2 && w_ok) {      // centre rows of this thread's kernel row's parity only
      for (int oy = ((PAD + wkh) & 1) + 2 * wrg; oy < g.TH; oy += 2 * p.RG) {
        const uint4* arow = cen + (size_t)(oy * g.TW) * g.OG + wu;
        const uint4* drow = tile + (size_t)((((oy + PAD - wkh) >> 1) + 1) * g.IW) * g.OG + wu;
        for (int sx = 0; sx < g.nstrips; ++sx) {
          uint4 ra[SW], rd[NIN];
#pragma unroll
          for (int j = 0; j < SW; ++j) ra[j] = arow[(size_t)(sx * SW + j) * g.OG];
#pragma unroll
          for (int x = 0; x < NIN; ++x) rd[x] = drow[(size_t)(sx * (SW / 2) + x) * g.OG];
#pragma unroll
          for (int j = 0; j < SW; ++j) {
            float a[8];
            unpack8h(ra[j], a);
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
              if ((j + PAD - kw) & 1) continue;               // compile time
              float tv[8];
              unpackN<8>(rd[((j + PAD - kw) >> 1) + 1], tv);
#pragma unroll
              for (int e = 0; e < 8; ++e) dW[kw * 8 + e] += a[e] * tv[e];
            }
          }
        }
      }
    }
    if (S == 1 && w_ok) {
      for (int oy = wrg; oy < g.TH; oy += p.RG) {
        const uint4* arow = cen + (size_t)(oy * g.TW) * g.OG + wu;
        const uint4* drow = tile + (size_t)((oy + K - 1 - wkh) * g.IW) * g.OG + wu;
        for (int sx = 0; sx < g.nstrips; ++sx) {
          uint4 ra[SW], rd[NIN];
#pragma unroll
          for (int j = 0; j < SW; ++j) ra[j] = arow[(size_t)(sx * SW + j) * g.OG];
#pragma unroll
          for (int x = 0; x < NIN; ++x) rd[x] = drow[(size_t)(sx * SW + x) * g.OG];
#if DWT_ROW_UNPACK
          float tv[NIN][8];
#pragma unroll
          for (int x = 0; x < NIN; ++x) unpackN<8>(rd[x], tv[x]);
#pragma unroll
          for (int j = 0; j < SW; ++j) {
            float a[8];
            unpack8h(ra[j], a);
#pragma unroll
            for (int kw = 0; kw < K; ++kw)
#pragma unroll
              for (int e = 0; e < 8; ++e) dW[kw * 8 + e] += a[e] * tv[j + K - 1 - kw][e];
          }
#else
#pragma unroll
          for (int j = 0; j < SW; ++j) {
            float a[8];
            unpack8h(ra[j], a);
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
              float tv[8];
              unpackN<8>(rd[j + K - 1 - kw], tv);
#pragma unroll
              for (int e = 0; e < 8; ++e) dW[kw * 8 + e] += a[e] * tv[e];
            }
          }
#endif
        }
      }
    }
  }
  float* red = reinterpret_cast<float*>(smem);
  if (!PLAIN) {
    lanes_reduce<16, 16>(st, red, u, pl, g.OG, g.NP);
    if (cok && pl == 0) {
      float* slab = p.parts_bn + (size_t)bx * 2 * g.C + c0;
#pragma unroll
      for (int e = 0; e < 8; ++e) { slab[e] = st[e]; slab[g.C + e] = st[8 + e]; }
    }
  }
  // weight-gradient partials: sum over the row groups (threads (wrg * K + wkh) * OG + wu), one slab row per block
#pragma unroll          // unrolled: dW must stay in registers (a runtime-indexed array goes to scratch)
  for (int kw = 0; kw < K; ++kw) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) red[tid * 8 + e] = w_ok ? dW[kw * 8 + e] : 0.f;
    __syncthreads();
    if (w_ok && wrg == 0) {
      float a[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] = dW[kw * 8 + e];
      for (int r = 1; r < p.RG; ++r)
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += red[(((r * K + wkh) * g.OG) + wu) * 8 + e];
      float* slab = p.parts_w + (size_t)bx * K * K * g.C + (size_t)(wkh * K + kw) * g.C + (by * g.OG + wu) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) slab[e] = a[e];
    }
  }
}

// ================================================================= host side
void mmsim_launch_reduce(const float* parts, int nparts, int n, float* out, int accumulate, hipStream_t s);   // conv.hip
void mmsim_launch_reduce2(const float* pa, int na, float* oa, const float* pb, int nb, float* ob, int nparts, hipStream_t s);


// Channel units per block: a divisor of the unit count near `target` octets (8 = 128 B per pixel) when there is one, so that
// no block carries idle lanes.  Planes so small that a tile is the whole plane (7x7) have fewer strips than a block of
// 8-octet groups has pixel lanes (14 strips for 42 lanes: two thirds of the block idles through the tap loop): they take
// wider groups so that strips ~ pixel lanes.
static int pick_og(int units, int target) {
  if (units <= target + target / 2) return units;           // the whole channel dimension in one group
  int best = 0;
  for (int d = target / 2; d <= 2 * target; ++d)
    if (units % d == 0 && (best == 0 || abs(d - target) < abs(best - target))) best = d;
  return best ? best : target;
}

// Output tile: TW a multiple of 4 (strips of four outputs per thread), chosen to minimise the staged-to-produced pixel
// ratio (halo + clipped overhang) under the LDS budget.
static void pick_tile(int Ho, int Wo, int K, int S, int og, bool bwd, int* th, int* tw) {
  double best = 1e30;
  *th = 4; *tw = 4;
  const long budget = bwd ? 72 * 1024 : 40 * 1024;      // backward: two workgroups per CU (registers), forward three to four
  for (int TW = 4; TW <= ((Wo + 3) & ~3) && TW <= 32; TW += 4) {
    for (int TH = 2; TH <= Ho && TH <= 32; ++TH) {
      const int IH = (TH - 1) * S + K, IW = (TW - 1) * S + K;
      const long lds = ((long)IH * IW + (bwd ? (long)TH * TW : 0)) * og * 16 + (long)K * K * og * 32;
      if (lds > budget) break;
      const int tx = (Wo + TW - 1) / TW, ty = (Ho + TH - 1) / TH;
      const double cost = ((double)tx * ty * IH * IW / (S * S) + 0.5 * tx * ty * TH * TW) / ((double)Ho * Wo);
      if (cost < best - 1e-9) { best = cost; *th = TH; *tw = TW; }
    }
  }
}

static int make_geom(DwTile* g, int B, int Hi, int Wi, int C, int K, int S, bool bwd, bool xf, size_t* lds, dim3* grid) {
  const int cpt = 8;
  MMSIM_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && C > 0 && (C % 8) == 0, "dwtile: bad geometry (C must be a multiple of 8)");
  MMSIM_REQUIRE((K == 3 || K == 5) && (S == 1 || S == 2), "dwtile: kernel 3/5 and stride 1/2 only");
  g->B = B; g->Hi = Hi; g->Wi = Wi; g->C = C;
  g->Ho = (Hi + 2 * (K / 2) - K) / S + 1; g->Wo = (Wi + 2 * (K / 2) - K) / S + 1;
  const int units = C / cpt;
  const int strips_plane = g->Ho * ((g->Wo + 3) / 4);
  g->OG = pick_og(units, strips_plane <= 16 ? 16 : 8);
  while (g->OG * K > 256) g->OG /= 2;
  g->NP = 256 / g->OG;
  pick_tile(g->Ho, g->Wo, K, S, g->OG, bwd, &g->TH, &g->TW);
  g->IH = (g->TH - 1) * S + K; g->IW = (g->TW - 1) * S + K;
  g->nstrips = g->TW / 4;
  const int tx = (g->Wo + g->TW - 1) / g->TW, ty = (g->Ho + g->TH - 1) / g->TH;
  g->total_tiles = B * tx * ty;
  const int gy = (units + g->OG - 1) / g->OG;
  int want = 2048 / gy; if (want < 1) want = 1;
  g->tiles_per_block = (g->total_tiles + want - 1) / want;
  g->d_tiles_img = make_fastdiv(tx * ty); g->d_tx = make_fastdiv(tx); g->d_iw = make_fastdiv(g->IW); g->d_strips = make_fastdiv(g->nstrips);
  size_t need = ((size_t)g->IH * g->IW + (bwd ? (size_t)g->TH * g->TW : 0)) * g->OG * 16 + (size_t)K * K * g->OG * 32;
  const size_t red = 256 * 16 * sizeof(float);            // the end-of-block reductions reuse the tile region
  *lds = need > red ? need : red;
  *grid = dwt_grid(g, (g->total_tiles + g->tiles_per_block - 1) / g->tiles_per_block, gy);
  return MMSIM_OK;
}

// Stride-2 backward: the centre tile lives in INPUT space (TH even, TW a multiple of 4), the staged dz2 tile in output space
// ((TH/2 + 2) x (TW/2 + 2)); DwTile's Ho / Wo stay the output plane, IH / IW the staged tile, tiles walk the input plane.
static int make_geom_bwd2(DwTile* g, int B, int Hi, int Wi, int C, int K, size_t* lds, dim3* grid) {
  MMSIM_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && C > 0 && (C % 8) == 0, "dwtile_bwd: bad geometry (C must be a multiple of 8)");
  MMSIM_REQUIRE(K == 3 || K == 5, "dwtile_bwd: kernel 3 / 5 only");
  g->B = B; g->Hi = Hi; g->Wi = Wi; g->C = C;
  g->Ho = (Hi + 2 * (K / 2) - K) / 2 + 1; g->Wo = (Wi + 2 * (K / 2) - K) / 2 + 1;
  const int units = C / 8;
  g->OG = pick_og(units, 8);
  while (g->OG * K > 256) g->OG /= 2;
  g->NP = 256 / g->OG;
  // tile: least (staged dy + z2 pixels, two tensors) + (z1 read + output written) per input pixel, under the LDS budget
  double best = 1e30;
  g->TH = 2; g->TW = 4;
  for (int TW = 4; TW <= ((Wi + 3) & ~3) && TW <= 32; TW += 4)
    for (int TH = 2; TH <= ((Hi + 1) & ~1) && TH <= 32; TH += 2) {
      const int IH = TH / 2 + 2, IW = TW / 2 + 2;
      const long need = ((long)IH * IW + (long)TH * TW) * g->OG * 16 + (long)K * K * g->OG * 32;
      if (need > 72 * 1024) break;
      const int tx = (Wi + TW - 1) / TW, ty = (Hi + TH - 1) / TH;
      const double cost = ((double)tx * ty * (2.0 * IH * IW + 2.0 * TH * TW)) / ((double)Hi * Wi);
      if (cost < best - 1e-9) { best = cost; g->TH = TH; g->TW = TW; }
    }
  g->IH = g->TH / 2 + 2; g->IW = g->TW / 2 + 2;
  g->nstrips = g->TW / 4;
  const int tx = (Wi + g->TW - 1) / g->TW, ty = (Hi + g->TH - 1) / g->TH;
  g->total_tiles = B * tx * ty;
  const int gy = (units + g->OG - 1) / g->OG;
  int want = 2048 / gy; if (want < 1) want = 1;
  g->tiles_per_block = (g->total_tiles + want - 1) / want;
  g->d_tiles_img = make_fastdiv(tx * ty); g->d_tx = make_fastdiv(tx); g->d_iw = make_fastdiv(g->IW); g->d_strips = make_fastdiv(g->nstrips);
  const size_t need = ((size_t)g->IH * g->IW + (size_t)g->TH * g->TW) * g->OG * 16 + (size_t)K * K * g->OG * 32;
  const size_t red = 256 * 16 * sizeof(float);
  *lds = need > red ? need : red;
  *grid = dwt_grid(g, (g->total_tiles + g->tiles_per_block - 1) / g->tiles_per_block, gy);
  return MMSIM_OK;
}

// more than 64 KiB of dynamic LDS needs the per-(function, device) opt-in
static void optin_bwd_lds() {
  static unsigned long long done = 0;
  const int dev = mmsim_current_device();
  if ((done >> dev) & 1) return;
  const int cap = 96 * 1024;
  (void)hipFuncSetAttribute((const void*)dwt_bwd_kernel<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
  (void)hipFuncSetAttribute((const void*)dwt_bwd_kernel<3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
  (void)hipFuncSetAttribute((const void*)dwt_bwd_kernel<5, true>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
  (void)hipFuncSetAttribute((const void*)dwt_bwd_kernel<5, false>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
  (void)hipFuncSetAttribute((const void*)dwt_bwd_kernel<3, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
  (void)hipFuncSetAttribute((const void*)dwt_bwd_kernel<5, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
  done |= 1ull << dev;
}

extern "C" int mmsim_dwtile_fwd(const void* in, const float* xf_scale, const float* xf_shift, const float* w_tap_major, void* z,
                                float* sums, int B, int Hi, int Wi, int C, int K, int S, float* scratch,
                                unsigned long long scratch_floats, void* stream) {
  MMSIM_REQUIRE(in && w_tap_major && z && sums && scratch, "dwtile_fwd: null operand");
  MMSIM_REQUIRE((xf_scale == nullptr) == (xf_shift == nullptr), "dwtile_fwd: scale and shift come together");
  DwTile g; size_t lds; dim3 grid;
  int rc = make_geom(&g, B, Hi, Wi, C, K, S, false, xf_scale != nullptr, &lds, &grid); if (rc) return rc;
  MMSIM_REQUIRE(scratch_floats >= (unsigned long long)g.gx * 2 * C, "dwtile_fwd: scratch too small");
  hipStream_t s = (hipStream_t)stream;
#define DWT_F(KK, SS, XX) hipLaunchKernelGGL((dwt_fwd_kernel<KK, SS, XX>), grid, dim3(256), lds, s, (const f16*)in, xf_scale, xf_shift, w_tap_major, (f16*)z, scratch, g)
#define DWT_FX(XX)                                                                        \
  if (K == 3 && S == 1) DWT_F(3, 1, XX); else if (K == 3 && S == 2) DWT_F(3, 2, XX);     \
  else if (K == 5 && S == 1) DWT_F(5, 1, XX); else DWT_F(5, 2, XX);
  if (xf_scale) { DWT_FX(true) } else { DWT_FX(false) }
#undef DWT_FX
#undef DWT_F
  mmsim_launch_reduce(scratch, g.gx, 2 * C, sums, 1, s);      /* sums += (pre-zeroed by the caller) */
  return mmsim_check_launch("dwtile_fwd");
}

static int dwtile_bwd_impl(const void* dy, const void* z2, const float* scale2, const float* shift2, const float* mean2,
                           const float* rstd2, const float* sums2, const float* gate, const float* dsq, const void* z1,
                           const float* scale1, const float* shift1, const float* mean1, const float* rstd1,
                           const void* resid, const float* w_tap_major, void* out, float* sums1, float* g_tap_major,
                           float* dgamma2, float* dbeta2, int B, int H, int W, int C, int K, int S, float* scratch,
                           unsigned long long scratch_floats, void* stream) {
  MMSIM_REQUIRE(dy && z2 && scale2 && shift2 && mean2 && rstd2 && sums2 && gate && dsq && z1 && w_tap_major && out && g_tap_major &&
                    dgamma2 && dbeta2 && scratch, "dwtile_bwd: null operand");
  const bool plain = scale1 == nullptr;
  MMSIM_REQUIRE(plain ? (!shift1 && !mean1 && !rstd1) : (shift1 && mean1 && rstd1 && sums1 && !resid),
                "dwtile_bwd: the expand BatchNorm state comes as a whole (IR block, no resid) or not at all (DS block)");
  MMSIM_REQUIRE(S == 1 || (S == 2 && !plain), "dwtile_bwd: stride 1, or stride 2 for an IR block (expand BatchNorm state given)");
  DwTile g; size_t lds; dim3 grid;
  int rc = S == 1 ? make_geom(&g, B, H, W, C, K, 1, true, true, &lds, &grid) : make_geom_bwd2(&g, B, H, W, C, K, &lds, &grid);
  if (rc) return rc;
  const size_t n_bn = plain ? 0 : (size_t)g.gx * 2 * C, n_w = (size_t)g.gx * K * K * C;
  MMSIM_REQUIRE(scratch_floats >= (unsigned long long)(n_bn + n_w), "dwtile_bwd: scratch too small");
  DwBwd p;
  p.dy = (const bf16*)dy; p.z2 = (const f16*)z2; p.z1 = (const f16*)z1; p.resid = (const bf16*)resid;
  p.sc2 = scale2; p.sh2 = shift2; p.mu2 = mean2; p.rs2 = rstd2; p.sums2 = sums2; p.gate = gate; p.dsq = dsq;
  p.sc1 = scale1; p.sh1 = shift1; p.mu1 = mean1; p.rs1 = rstd1; p.wT = w_tap_major; p.out = (bf16*)out;
  p.parts_bn = scratch; p.parts_w = scratch + n_bn; p.dgamma2 = dgamma2; p.dbeta2 = dbeta2;
  p.invP = 1.0f / (float)((size_t)B * g.Ho * g.Wo); p.inv_hw = 1.0f / (float)(g.Ho * g.Wo);      // the depthwise BatchNorm's / the SE pool's plane
  p.RG = 256 / (g.OG * K); if (p.RG < 1) p.RG = 1;
  MMSIM_REQUIRE(g.OG * K <= 256, "dwtile_bwd: channel group too wide for the weight-gradient mapping");
  hipStream_t s = (hipStream_t)stream;
  optin_bwd_lds();
#define DWT_B(KK, PP) hipLaunchKernelGGL((dwt_bwd_kernel<KK, PP>), grid, dim3(256), lds, s, p, g)
  if (S == 2) {
    if (K == 3) hipLaunchKernelGGL((dwt_bwd_kernel<3, false, 2>), grid, dim3(256), lds, s, p, g);
    else hipLaunchKernelGGL((dwt_bwd_kernel<5, false, 2>), grid, dim3(256), lds, s, p, g);
  }
  else if (K == 3) { if (plain) DWT_B(3, true); else DWT_B(3, false); }
  else { if (plain) DWT_B(5, true); else DWT_B(5, false); }
#undef DWT_B
  if (!plain) mmsim_launch_reduce2(p.parts_bn, 2 * C, sums1, p.parts_w, K * K * C, g_tap_major, g.gx, s);      // both slabs, one launch
  else mmsim_launch_reduce(p.parts_w, g.gx, K * K * C, g_tap_major, 1, s);
  return mmsim_check_launch("dwtile_bwd");
}

extern "C" int mmsim_dwtile_bwd(const void* dy, const void* z2, const float* scale2, const float* shift2, const float* mean2,
                                const float* rstd2, const float* sums2, const float* gate, const float* dsq, const void* z1,
                                const float* scale1, const float* shift1, const float* mean1, const float* rstd1,
                                const void* resid, const float* w_tap_major, void* out, float* sums1, float* g_tap_major,
                                float* dgamma2, float* dbeta2, int B, int H, int W, int C, int K, float* scratch,
                                unsigned long long scratch_floats, void* stream) {
  return dwtile_bwd_impl(dy, z2, scale2, shift2, mean2, rstd2, sums2, gate, dsq, z1, scale1, shift1, mean1, rstd1, resid, w_tap_major, out,
                         sums1, g_tap_major, dgamma2, dbeta2, B, H, W, C, K, 1, scratch, scratch_floats, stream);
}
// The same fused backward for a STRIDE-2 depthwise conv of an IR block: H x W is the conv's INPUT plane (z1, out), dy / z2 / gate / dsq
// live on the output plane ((H + 2 (K/2) - K) / 2 + 1 squared: timm's symmetric padding, cv_classifier.py:49).
extern "C" int mmsim_dwtile_bwd_s2(const void* dy, const void* z2, const float* scale2, const float* shift2, const float* mean2,
                                   const float* rstd2, const float* sums2, const float* gate, const float* dsq, const void* z1,
                                   const float* scale1, const float* shift1, const float* mean1, const float* rstd1,
                                   const float* w_tap_major, void* out, float* sums1, float* g_tap_major,
                                   float* dgamma2, float* dbeta2, int B, int H, int W, int C, int K, float* scratch,
                                   unsigned long long scratch_floats, void* stream) {
  return dwtile_bwd_impl(dy, z2, scale2, shift2, mean2, rstd2, sums2, gate, dsq, z1, scale1, shift1, mean1, rstd1, nullptr, w_tap_major, out,
                         sums1, g_tap_major, dgamma2, dbeta2, B, H, W, C, K, 2, scratch, scratch_floats, stream);
}

// ================================================================= fused backward of the expand stage (early MBConv blocks)
//   x [P, cin] -> conv_pw (W1 [mid, cin]) -> z1 [P, mid] -> bn1 (train) -> silu -> ...
// Given dpre = dLoss/d(bn1 output) (what mmsim_dwtile_bwd leaves) and bn1's backward sums, ONE streaming pass forms
//   dz1 = scale (dpre - S1/P - zhat S2/P)        (in registers, on the way into LDS; never written to HBM)
//   dx  = dz1 W1 (+ resid)                       (MFMA, A = the staged strip)
//   dW1 += dz1^T x                               (MFMA, both operands through transposing LDS reads; accumulators live in
//                                                 registers across all the strips a block walks, one partial slab per block)
// replacing bn_bwd_apply (3 passes over [P, mid]) and the two products that each re-read dz1 (2 passes): 2 passes instead of 5
// over the widest tensors of the tower.  Only where W1 and dW1 are small enough to live in LDS / registers -- mid <= 352,
// cin <= 64: stages 1-2, which is where those tensors exceed the 256 MB Infinity Cache and every pass is an HBM pass.
struct PwBwd {
  const bf16* dpre; const f16* z1; const f16* x; const bf16* resid; const bf16* w1;      // w1: the bf16 weight shadow
  const float* sc1; const float* mu1; const float* rs1; const float* sums1;
  bf16* dx; float* parts; float* dgamma; float* dbeta;
  int P, mid, cin, nstrips, KS;      // KS = ceil(mid / 32)
  float invP;
};

__device__ __forceinline__ f4 mfma16(bf8 first, bf8 second, f4 acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(first, second, acc, 0, 0, 0);      // acc[e] = C[second row lane&15][first row 4(lane>>4)+e]
}

__device__ __forceinline__ f4 mfma16h(h8 first, h8 second, f4 acc) {      // the same tile shape on fp16 operands (forward products)
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(first, second, acc, 0, 0, 0);
}

template <int BM, int CIN_T, int MAXW, int NTHR = 256, int NCH = 6>
__global__ __launch_bounds__(NTHR) void pw_expand_bwd_kernel(PwBwd p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NW = NTHR / 64;                           // waves per block; NCH: 16-byte chunks per thread and tensor of one strip (host checks)
  constexpr int XP = CIN_T * 32 + 16;                     // X / W1 image pitch (bytes)
  constexpr int MTD = BM / 16;                            // pixel tiles of the dgrad product
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int G = p.mid >> 3;                               // channel octets
  const int ZP = p.KS * 64 + 16;                          // Z image pitch (bytes): KS*32 channels + pad
  char* zimg = smem;                                      // [BM][KS*32] bf16: dz1 of the strip, k-major
  char* ximg = zimg + BM * ZP;                            // [BM][CIN_T*16] bf16: x of the strip
  char* wimg = ximg + BM * XP;                            // [KS*32][CIN_T*16] bf16: W1 (rows = mid)
  float* cst = reinterpret_cast<float*>(wimg + p.KS * 32 * XP);      // [3][mid]: scale, A, Bc
  // ---- one-time staging: zero the images (pads must be zero), W1, the folded BatchNorm-backward constants
  for (int i = tid; i < (BM * ZP + BM * XP + p.KS * 32 * XP) / 16; i += NTHR) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  for (int i = tid; i < p.mid * (p.cin >> 3); i += NTHR) {
    const int r = i / (p.cin >> 3), c = i - r * (p.cin >> 3);
    *reinterpret_cast<uint4*>(wimg + r * XP + c * 16) = *reinterpret_cast<const uint4*>(p.w1 + (size_t)r * p.cin + c * 8);
  }
  for (int c = tid; c < p.mid; c += NTHR) {
    const float sc = p.sc1[c], bc = sc * p.rs1[c] * p.sums1[p.mid + c] * p.invP;
    cst[c] = sc; cst[p.mid + c] = sc * p.sums1[c] * p.invP - p.mu1[c] * bc; cst[2 * p.mid + c] = bc;
    if (blockIdx.x == 0) { p.dgamma[c] += p.sums1[p.mid + c]; p.dbeta[c] += p.sums1[c]; }
  }
  const int nwt = ((p.mid + 15) >> 4) * CIN_T;            // weight-gradient tiles (16 x 16), dealt round-robin to the waves
  f4 dW[MAXW];
#pragma unroll
  for (int i = 0; i < MAXW; ++i) dW[i] = f4{0.f, 0.f, 0.f, 0.f};
  const int nchunks = BM * G;
  constexpr int NX = (BM * CIN_T * 2 + NTHR - 1) / NTHR;   // 16-byte chunks of the strip's x rows per thread (cin <= 16 CIN_T)
  constexpr int TPD = (MTD * CIN_T + NW - 1) / NW;         // dx tiles per wave
  const int xg = p.cin >> 3, nxch = BM * xg;
  u4v vd[NCH], vz[NCH], vx[NX];
  // the next strip's dpre | z1 | x rows: requested as a whole, consumed at the top of the next trip.  (Round 3: a second register set
  // with the loads issued TWO trips ahead measured 2-4 % slower than this form, 226-229 vs 220-223 us at 56^2 x 192 -- three of these
  // workgroups share a CU, the round trip is already covered; tools/bench_pwexpbwd.py.)
  auto request = [&](int strip) {
    const size_t base = (size_t)strip * BM * p.mid;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int q = min(tid + NTHR * i, nchunks - 1);
      vd[i] = *reinterpret_cast<const u4v*>(p.dpre + base + (size_t)q * 8);
      vz[i] = *reinterpret_cast<const u4v*>(p.z1 + base + (size_t)q * 8);
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int q = min(tid + NTHR * i, nxch - 1);          // rows of a strip are contiguous in x: chunk q of the strip
      vx[i] = *reinterpret_cast<const u4v*>(p.x + (size_t)strip * BM * p.cin + (size_t)q * 8);
    }
  };
  int strip = blockIdx.x;
  if (strip < p.nstrips) request(strip);
  for (; strip < p.nstrips; strip += gridDim.x) {
    __syncthreads();                     // the previous strip's MFMAs have read the images (first trip: the staging above)
    // every prefetched chunk is USED here, whether or not this thread stages it: a chunk skipped by `q < nchunks` below is a load
    // hipcc never saw waited for, and before its registers are written again (the next request) it then waits vmcnt(0) -- which
    // sat between the residual loads and the request and exposed both
#pragma unroll
    for (int i = 0; i < NCH; ++i) asm volatile("" : "+v"(vd[i]), "+v"(vz[i]));
#pragma unroll
    for (int i = 0; i < NX; ++i) asm volatile("" : "+v"(vx[i]));
    // ---- dz1 -> Z image, x -> X image
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int q = tid + NTHR * i;
      if (q < nchunks) {
        const int pix = q / G, un = q - pix * G;
        const bf8 d = __builtin_bit_cast(bf8, vd[i]);
        const h8 z = __builtin_bit_cast(h8, vz[i]);
        const float* cs = cst + un * 8;
        bf8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(cs[e] * bf2f(d[e]) - cs[p.mid + e] - h2f(z[e]) * cs[2 * p.mid + e]);
        *reinterpret_cast<bf8*>(zimg + pix * ZP + un * 16) = o;
      }
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int q = tid + NTHR * i;
      if (q < nxch) {
        const int pix = q / xg, un = q - pix * xg;
        // x is fp16 in HBM; the weight-gradient MFMA pairs it with the bf16 dz1 image, so it is staged as bf16
        const h8 xv = __builtin_bit_cast(h8, vx[i]);
        bf8 xo;
#pragma unroll
        for (int e = 0; e < 8; ++e) xo[e] = f2bf(h2f(xv[e]));
        *reinterpret_cast<bf8*>(ximg + pix * XP + un * 16) = xo;
      }
    }
    __syncthreads();
    // The residual rows of this wave's dx tiles are requested BEFORE the next strip's loads: the memory counter is in order, so a
    // residual load issued after them (as it was up to r03, inside the tile loop) made its wait a wait for the whole prefetch -- in
    // front of the MFMAs it was meant to fly under.
    bf4 rres[TPD];
#pragma unroll
    for (int i = 0; i < TPD; ++i) {
      const int t = wave + NW * i, mt = t / CIN_T, nt = t - mt * CIN_T;
      const int pix = mt * 16 + (lane & 15), ci = min(nt * 16 + (lane >> 4) * 4, p.cin - 4);
      rres[i] = bf4{(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
      if (p.resid) rres[i] = *reinterpret_cast<const bf4*>(p.resid + ((size_t)strip * BM + min(pix, BM - 1)) * p.cin + ci);   // launch-uniform branch
    }
    // the next strip's loads fly under the MFMAs.  UNCONDITIONAL (the last trip re-requests its own strip and drops it): behind
    // `if (next < nstrips)` the number of loads younger than the residual rows differs between the two paths, hipcc's counted wait
    // for those rows falls back to vmcnt(0), and the prefetch is waited for after the first tile's MFMAs.
    request(min(strip + (int)gridDim.x, p.nstrips - 1));
    // ---- dx = dz1 W1: (BM/16) x CIN_T tiles over the waves
#pragma unroll
    for (int i = 0; i < TPD; ++i) {
      const int t = wave + NW * i;
      if (t < MTD * CIN_T) {             // wave-uniform
        const int mt = t / CIN_T, nt = t - mt * CIN_T;
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < p.KS; ++ks) {
          const bf8 zf = *reinterpret_cast<const bf8*>(zimg + (mt * 16 + (lane & 15)) * ZP + (ks * 4 + (lane >> 4)) * 16);
          const bf8 wf = tr_frag16(wimg, XP, ks * 32, nt * 16, lane);
          acc = mfma16(wf, zf, acc);
        }
        const int pix = mt * 16 + (lane & 15), ci = nt * 16 + (lane >> 4) * 4;
        if (ci < p.cin) {                // cin is a multiple of 8: a 4-channel group is valid as a whole
          const size_t off = ((size_t)strip * BM + pix) * p.cin + ci;
          asm volatile("" : "+v"(rres[i]));      // first use HERE: without it the scheduler converts the rows to fp32 ahead of the
                                                 // prefetch above and the wait for them moves in front of it
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] += bf2f(rres[i][e]);
          const bf4 o = {f2bf(acc[0]), f2bf(acc[1]), f2bf(acc[2]), f2bf(acc[3])};
          *reinterpret_cast<bf4*>(p.dx + off) = o;
        }
      }
    }
    // ---- dW1 += dz1^T x: tiles (mid/16) x CIN_T, reduction over the strip's pixels
#pragma unroll
    for (int i = 0; i < MAXW; ++i) {
      const int t = wave + NW * i;
      if (t < nwt) {                     // wave-uniform: EXEC stays all ones for the transposing reads
        const int mt = t / CIN_T, nt = t - mt * CIN_T;
#pragma unroll
        for (int ks = 0; ks < BM / 32; ++ks) {
          const bf8 zt = tr_frag16(zimg, ZP, ks * 32, mt * 16, lane);
          const bf8 xt = tr_frag16(ximg, XP, ks * 32, nt * 16, lane);
          dW[i] = mfma16(xt, zt, dW[i]);
        }
      }
    }
  }
  // ---- this block's partial weight gradient: slab row [mid][cin] fp32
  float* slab = p.parts + (size_t)blockIdx.x * p.mid * p.cin;
#pragma unroll
  for (int i = 0; i < MAXW; ++i) {
    const int t = wave + NW * i;
    if (t < nwt) {
      const int mt = t / CIN_T, nt = t - mt * CIN_T;
      const int cm = mt * 16 + (lane & 15), ci = nt * 16 + (lane >> 4) * 4;
      if (cm < p.mid && ci < p.cin) *reinterpret_cast<f4*>(slab + (size_t)cm * p.cin + ci) = dW[i];
    }
  }
}

template <int BM, int CIN_T, int MAXW, int NTHR = 256, int NCH = 6>
static int launch_pw_expand_bwd(PwBwd p, float* dw1, hipStream_t s, float* scratch, unsigned long long scratch_floats) {
  p.nstrips = p.P / BM;
  p.KS = (p.mid + 31) / 32;
  const int ZP = p.KS * 64 + 16, XP = CIN_T * 32 + 16;
  const size_t lds = (size_t)BM * ZP + (size_t)BM * XP + (size_t)p.KS * 32 * XP + (size_t)3 * p.mid * 4;
  int grid = p.nstrips < 768 ? p.nstrips : 768;
  const size_t slab = (size_t)p.mid * p.cin;
  while (grid > 64 && (size_t)grid * slab > scratch_floats) grid /= 2;
  MMSIM_REQUIRE((size_t)grid * slab <= scratch_floats, "pw_expand_bwd: scratch too small");
  MMSIM_REQUIRE(BM * (p.mid >> 3) <= NTHR * NCH, "pw_expand_bwd: strip too wide for the staging registers");
  MMSIM_REQUIRE(((p.mid + 15) / 16) * CIN_T <= (NTHR / 64) * MAXW, "pw_expand_bwd: weight gradient does not fit the accumulators");
  static unsigned long long done = 0;
  const int dev = mmsim_current_device();
  if (!((done >> dev) & 1)) {
    (void)hipFuncSetAttribute((const void*)pw_expand_bwd_kernel<BM, CIN_T, MAXW, NTHR, NCH>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    done |= 1ull << dev;
  }
  MMSIM_REQUIRE(lds <= 120 * 1024, "pw_expand_bwd: LDS images too large");
  p.parts = scratch;
  hipLaunchKernelGGL((pw_expand_bwd_kernel<BM, CIN_T, MAXW, NTHR, NCH>), dim3(grid), dim3(NTHR), lds, s, p);
  mmsim_launch_reduce(scratch, grid, (int)slab, dw1, 1, s);
  return mmsim_check_launch("pw_expand_bwd");
}

// ---------------------------------------------------------------------------------------------------------
// Projection 1x1 conv of the early MBConv stages as ONE streaming pass (forward):
//   z3[p][co] = sum_c (a2[p][c] * gate[p / HW][c]) * W3[co][c]       + per-channel sum / sum of squares of the bf16-rounded z3
// At 112^2 / 56^2 / 28^2 the product is a pure HBM stream of a2 (mid <= 336 channels in, cout <= 64 out: 2 x cout FLOP per byte
// read), and the 128 x 128-tile GEMM spends its time on tile bookkeeping (K fits one or two 64-deep steps, 50-80 % of every B tile
// is padding): 1.5-2.6 TB/s.  Here a block walks a CONTIGUOUS range of BM-pixel strips: W3 stays in LDS for the block's lifetime,
// the strip's a2 chunks are requested one strip ahead, multiplied by the SE gate (the image's gate row is cached in LDS and only
// reloaded when the range enters the next image) and written k-major into LDS; the 16 x 16 x 32 MFMAs read both operands with
// plain 16-byte fragment reads.  Statistics stay in registers until the block ends (slab + mmsim_launch_reduce).
struct PwPrj {
  const f16* a2; const float* gate; const f16* w3; f16* z3; float* parts;      // w3: the fp16 weight shadow
  const float* xsc; const float* xsh;      // non-NULL: `a2` is the pre-BatchNorm tensor z2 and the operand is silu(xsc z2 + xsh) * gate
  int P, HW, B, mid, cout, nstrips, per_block, KS;
  FastDiv dhw;
};

template <int BM, int COUT_T, int NCH>
__global__ __launch_bounds__(256) void pw_project_fwd_kernel(PwPrj p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int MTD = BM / 16;                            // pixel tiles per strip
  constexpr int WM = MTD >= 4 ? 4 : MTD;                  // waves along the pixel tiles
  constexpr int WN = 4 / WM;                              // waves along the output-channel tiles
  constexpr int MT_W = MTD / WM, NT_W = COUT_T / WN;
  static_assert(MT_W >= 1 && NT_W >= 1 && MT_W * WM == MTD && NT_W * WN == COUT_T, "tile split");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int G = p.mid >> 3;
  const int ZP = p.KS * 64 + 16;                          // image pitch (bytes): KS*32 channels + pad
  char* zimg = smem;                                      // [BM][KS*32] bf16: gated a2 of the strip, k-major
  char* wimg = zimg + BM * ZP;                            // [COUT_T*16][KS*32] bf16: W3 (rows = cout, zero rows past cout)
  float* grow = reinterpret_cast<float*>(wimg + COUT_T * 16 * ZP);      // [2][mid]: gate rows of image cur_b, cur_b + 1
  float* xs = grow + 2 * p.mid;                           // [2][mid]: BatchNorm scale | shift of the operand transform (if any)
  float* red = xs + 2 * p.mid;                            // [4 waves][2][COUT_T*16]
  for (int i = tid; i < (BM * ZP + COUT_T * 16 * ZP) / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);
  for (int i = tid; i < 4 * 2 * COUT_T * 16; i += 256) red[i] = 0.f;
  if (p.xsc)
    for (int i = tid; i < p.mid; i += 256) { xs[i] = p.xsc[i]; xs[p.mid + i] = p.xsh[i]; }
  __syncthreads();
  for (int i = tid; i < p.cout * G; i += 256) {
    const int r = i / G, c = i - r * G;
    *reinterpret_cast<uint4*>(wimg + r * ZP + c * 16) = *reinterpret_cast<const uint4*>(p.w3 + (size_t)r * p.mid + c * 8);
  }
  const int mt0 = (wave % WM) * MT_W, nt0 = (wave / WM) * NT_W;
  float cs[NT_W][4], cq[NT_W][4];
#pragma unroll
  for (int n = 0; n < NT_W; ++n)
#pragma unroll
    for (int e = 0; e < 4; ++e) cs[n][e] = cq[n][e] = 0.f;
  const int nchunks = BM * G;
  uint4 va[NCH];
  auto request = [&](int strip) {
    const size_t base = (size_t)strip * BM * p.mid;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int q = min(tid + 256 * i, nchunks - 1);
      va[i] = *reinterpret_cast<const uint4*>(p.a2 + base + (size_t)q * 8);
    }
  };
  const int sbeg = blockIdx.x * p.per_block, send = min(p.nstrips, sbeg + p.per_block);
  int cur_b = -1;
  if (sbeg < send) request(sbeg);
  for (int strip = sbeg; strip < send; ++strip) {
    const int s0 = strip * BM;
    const int b0 = (int)fdiv((unsigned)s0, p.dhw);
    if (b0 != cur_b) {                   // block-uniform; every thread is past the previous strip's staging (its second barrier)
      for (int i = tid; i < 2 * p.mid; i += 256) {
        const int j = i >= p.mid, c = i - j * p.mid;
        grow[i] = p.gate[(size_t)min(b0 + j, p.B - 1) * p.mid + c];
      }
      cur_b = b0;
    }
    __syncthreads();                     // the previous strip's MFMAs have read the Z image; the gate rows are in place
    const int split = (b0 + 1) * p.HW - s0;              // pixels of this strip before the next image starts (HW >= BM: at most two images)
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int q = tid + 256 * i;
      if (q < nchunks) {
        const int pix = q / G, un = q - pix * G;
        const float* gr = grow + (pix >= split ? p.mid : 0) + un * 8;
        const h8 a = __builtin_bit_cast(h8, va[i]);
        h8 o;
        if (p.xsc) {      // launch-uniform: the activation is formed here, a2 is never stored
          const float* cs = xs + un * 8;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = f2h(silu_f(h2f(a[e]) * cs[e] + cs[p.mid + e]) * gr[e]);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = f2h(h2f(a[e]) * gr[e]);
        }
        *reinterpret_cast<h8*>(zimg + pix * ZP + un * 16) = o;
      }
    }
    __syncthreads();
    if (strip + 1 < send) request(strip + 1);            // the next strip's loads fly under the MFMAs
#pragma unroll
    for (int m = 0; m < MT_W; ++m) {
      const int mt = mt0 + m;
      f4 acc[NT_W];
#pragma unroll
      for (int n = 0; n < NT_W; ++n) acc[n] = f4{0.f, 0.f, 0.f, 0.f};
      for (int ks = 0; ks < p.KS; ++ks) {
        const h8 zf = *reinterpret_cast<const h8*>(zimg + (mt * 16 + (lane & 15)) * ZP + (ks * 4 + (lane >> 4)) * 16);
#pragma unroll
        for (int n = 0; n < NT_W; ++n) {
          const h8 wf = *reinterpret_cast<const h8*>(wimg + ((nt0 + n) * 16 + (lane & 15)) * ZP + (ks * 4 + (lane >> 4)) * 16);
          acc[n] = mfma16h(wf, zf, acc[n]);                 // acc[e] = z3[pixel lane&15][channel 4(lane>>4)+e] of tile (mt, nt0+n)
        }
      }
      const int pix = mt * 16 + (lane & 15);
#pragma unroll
      for (int n = 0; n < NT_W; ++n) {
        const int co = (nt0 + n) * 16 + (lane >> 4) * 4;
        const h4 o = {f2h(acc[n][0]), f2h(acc[n][1]), f2h(acc[n][2]), f2h(acc[n][3])};
        if (co < p.cout) *reinterpret_cast<h4*>(p.z3 + ((size_t)s0 + pix) * p.cout + co) = o;      // cout % 8 == 0: the group is valid as a whole
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float v = h2f(o[e]); cs[n][e] += v; cq[n][e] += v * v; }   // zero weight rows past cout add zeros
      }
    }
  }
  // ---- statistics of this block: lanes of one channel group (same lane >> 4) hold 16 different pixels
#pragma unroll
  for (int n = 0; n < NT_W; ++n)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int sh = 1; sh < 16; sh <<= 1) { cs[n][e] += __shfl_xor(cs[n][e], sh, 64); cq[n][e] += __shfl_xor(cq[n][e], sh, 64); }
    }
  if ((lane & 15) == 0) {
#pragma unroll
    for (int n = 0; n < NT_W; ++n)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int co = (nt0 + n) * 16 + (lane >> 4) * 4 + e;
        red[(wave * 2 + 0) * COUT_T * 16 + co] = cs[n][e];
        red[(wave * 2 + 1) * COUT_T * 16 + co] = cq[n][e];
      }
  }
  __syncthreads();
  if (tid < 2 * p.cout) {
    const int which = tid >= p.cout, co = tid - which * p.cout;
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) t += red[(w * 2 + which) * COUT_T * 16 + co];
    p.parts[(size_t)blockIdx.x * 2 * p.cout + which * p.cout + co] = t;
  }
}

template <int BM, int COUT_T, int NCH>
static int launch_pw_project_fwd(PwPrj p, float* sums, hipStream_t s, float* scratch, unsigned long long scratch_floats) {
  p.nstrips = p.P / BM;
  p.KS = (p.mid + 31) / 32;
  const int ZP = p.KS * 64 + 16;
  const size_t lds = (size_t)BM * ZP + (size_t)COUT_T * 16 * ZP + (size_t)4 * p.mid * 4 + (size_t)4 * 2 * COUT_T * 16 * 4;
  MMSIM_REQUIRE(lds <= 100 * 1024, "pw_project_fwd: LDS images too large");
  MMSIM_REQUIRE(BM * (p.mid >> 3) <= 256 * NCH, "pw_project_fwd: strip too wide for the staging registers");
  int grid = p.nstrips < 1024 ? p.nstrips : 1024;
  while (grid > 64 && (size_t)grid * 2 * p.cout > scratch_floats) grid /= 2;
  MMSIM_REQUIRE((size_t)grid * 2 * p.cout <= scratch_floats, "pw_project_fwd: scratch too small");
  p.per_block = (p.nstrips + grid - 1) / grid;
  grid = (p.nstrips + p.per_block - 1) / p.per_block;      // no empty blocks: every slab row is written
  static unsigned long long done = 0;
  const int dev = mmsim_current_device();
  if (!((done >> dev) & 1)) {
    (void)hipFuncSetAttribute((const void*)pw_project_fwd_kernel<BM, COUT_T, NCH>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    done |= 1ull << dev;
  }
  p.parts = scratch;
  hipLaunchKernelGGL((pw_project_fwd_kernel<BM, COUT_T, NCH>), dim3(grid), dim3(256), lds, s, p);
  mmsim_launch_reduce(scratch, grid, 2 * p.cout, sums, 1, s);      /* sums += (pre-zeroed by the caller) */
  return mmsim_check_launch("pw_project_fwd");
}

// variant by shape: 0 = not eligible
static int pw_project_variant(int P, int HW, int mid, int cout) {
  if (P <= 0 || HW <= 0 || (P % HW) || (mid % 8) || (cout % 8)) return 0;
  if (mid <= 48 && cout <= 32 && HW >= 256 && (P % 256) == 0) return 1;        // depthwise-separable blocks at 112^2
  if (mid <= 192 && cout <= 32 && HW >= 64 && (P % 64) == 0) return 2;         // 56^2 stage
  if (mid <= 192 && cout <= 64 && HW >= 32 && (P % 32) == 0) return 3;         // first 28^2 block (wider ones: the GEMM measures the same)
  return 0;
}
extern "C" int mmsim_pw_project_fwd_eligible(int P, int HW, int mid, int cout) { return pw_project_variant(P, HW, mid, cout) != 0; }

static int pw_project_fwd_impl(const void* a2, const float* xf_scale, const float* xf_shift, const float* gate, const void* w3_f16,
                               void* z3, float* sums, int P, int HW, int mid, int cout, float* scratch,
                               unsigned long long scratch_floats, void* stream);
extern "C" int mmsim_pw_project_fwd(const void* a2, const float* gate, const void* w3_f16, void* z3, float* sums, int P, int HW,
                                    int mid, int cout, float* scratch, unsigned long long scratch_floats, void* stream) {
  return pw_project_fwd_impl(a2, nullptr, nullptr, gate, w3_f16, z3, sums, P, HW, mid, cout, scratch, scratch_floats, stream);
}
extern "C" int mmsim_pw_project_fwd_xf(const void* z2, const float* xf_scale, const float* xf_shift, const float* gate, const void* w3_f16,
                                       void* z3, float* sums, int P, int HW, int mid, int cout, float* scratch,
                                       unsigned long long scratch_floats, void* stream) {
  MMSIM_REQUIRE(xf_scale && xf_shift, "pw_project_fwd_xf: scale and shift required");
  return pw_project_fwd_impl(z2, xf_scale, xf_shift, gate, w3_f16, z3, sums, P, HW, mid, cout, scratch, scratch_floats, stream);
}
static int pw_project_fwd_impl(const void* a2, const float* xf_scale, const float* xf_shift, const float* gate, const void* w3_f16,
                               void* z3, float* sums, int P, int HW, int mid, int cout, float* scratch,
                               unsigned long long scratch_floats, void* stream) {
  MMSIM_REQUIRE(a2 && gate && w3_f16 && z3 && sums && scratch, "pw_project_fwd: null operand");
  const int v = pw_project_variant(P, HW, mid, cout);
  MMSIM_REQUIRE(v != 0, "pw_project_fwd: shape not eligible (see mmsim_pw_project_fwd_eligible)");
  PwPrj p;
  p.xsc = xf_scale; p.xsh = xf_shift;
  p.a2 = (const f16*)a2; p.gate = gate; p.w3 = (const f16*)w3_f16; p.z3 = (f16*)z3;
  p.P = P; p.HW = HW; p.B = P / HW; p.mid = mid; p.cout = cout; p.dhw = make_fastdiv((unsigned)HW);
  if (v == 1) return launch_pw_project_fwd<256, 2, 6>(p, sums, (hipStream_t)stream, scratch, scratch_floats);
  if (v == 2) return launch_pw_project_fwd<64, 2, 6>(p, sums, (hipStream_t)stream, scratch, scratch_floats);
  return launch_pw_project_fwd<32, 4, 6>(p, sums, (hipStream_t)stream, scratch, scratch_floats);
}

// Expansion 1x1 conv of the 56^2 stage as one streaming pass (forward): z1[p][m] = sum_c x[p][c] W1[m][c] + the per-channel
// sum / sum of squares of the bf16-rounded z1.  K = cin <= 32 is ONE MFMA step: the product is a pure output stream (mid / cin = 6
// bytes written per byte read).  W1 stays in LDS; the strip's output tile is assembled in an LDS image and leaves as whole 16-byte
// row chunks by threads that each own one channel octet for the block's lifetime -- the same threads keep that octet's statistics
// in registers, so no second pass and no atomics (slab + mmsim_launch_reduce at the end).
struct PwExp {
  const f16* x; const f16* w1; f16* z1; float* parts;      // w1: the fp16 weight shadow
  int P, mid, cin, nstrips, per_block;
};

template <int BM, int CIN_T>
__global__ __launch_bounds__(256) void pw_expand_fwd_kernel(PwExp p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int XP = CIN_T * 32 + 16;                     // x / W1 image pitch (bytes)
  constexpr int KS = CIN_T / 2;                           // 32-deep reduction steps
  constexpr int MT_W = BM / 64;                           // pixel tiles per wave
  static_assert(MT_W >= 1 && (CIN_T % 2) == 0, "tile split");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int G = p.mid >> 3, XG = p.cin >> 3, NT = (p.mid + 15) >> 4;
  const int OP = NT * 32 + 16;                            // output image pitch (bytes)
  char* ximg = smem;                                      // [BM][CIN_T*16]
  char* wimg = ximg + BM * XP;                            // [NT*16][CIN_T*16]  W1 (rows = mid, zero rows past mid)
  char* oimg = wimg + NT * 16 * XP;                       // [BM][NT*16]
  for (int i = tid; i < (BM * XP + NT * 16 * XP) / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  for (int i = tid; i < p.mid * XG; i += 256) {
    const int r = i / XG, c = i - r * XG;
    *reinterpret_cast<uint4*>(wimg + r * XP + c * 16) = *reinterpret_cast<const uint4*>(p.w1 + (size_t)r * p.cin + c * 8);
  }
  // copy-out ownership: thread -> (channel octet un, pixel lane pl); NR pixel rows per pass
  const int NR = 256 / G;
  const int un = tid % G, pl = tid / G;
  const bool owner = pl < NR;
  float st[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) st[e] = 0.f;
  const int nxch = BM * XG;
  uint4 vx;
  auto request = [&](int strip) {
    vx = *reinterpret_cast<const uint4*>(p.x + (size_t)strip * BM * p.cin + (size_t)min(tid, nxch - 1) * 8);
  };
  static_assert(BM * 4 <= 256, "one x chunk per thread (cin <= 32)");
  const int sbeg = blockIdx.x * p.per_block, send = min(p.nstrips, sbeg + p.per_block);
  if (sbeg < send) request(sbeg);
  for (int strip = sbeg; strip < send; ++strip) {
    __syncthreads();                     // A: the previous strip's output image has been copied out
    if (tid < nxch) {
      const int pix = tid / XG, u = tid - pix * XG;
      *reinterpret_cast<uint4*>(ximg + pix * XP + u * 16) = vx;
    }
    __syncthreads();                     // B
    if (strip + 1 < send) request(strip + 1);
#pragma unroll
    for (int m = 0; m < MT_W; ++m) {
      const int mt = wave * MT_W + m;
      h8 xf[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        xf[ks] = *reinterpret_cast<const h8*>(ximg + (mt * 16 + (lane & 15)) * XP + (ks * 4 + (lane >> 4)) * 16);
      for (int nt = 0; nt < NT; ++nt) {
        f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const h8 wf = *reinterpret_cast<const h8*>(wimg + (nt * 16 + (lane & 15)) * XP + (ks * 4 + (lane >> 4)) * 16);
          acc = mfma16h(wf, xf[ks], acc);                  // acc[e] = z1[pixel lane&15][channel nt*16 + 4(lane>>4) + e]
        }
        const h4 o = {f2h(acc[0]), f2h(acc[1]), f2h(acc[2]), f2h(acc[3])};
        *reinterpret_cast<h4*>(oimg + (mt * 16 + (lane & 15)) * OP + (nt * 16 + (lane >> 4) * 4) * 2) = o;
      }
    }
    __syncthreads();                     // C: the output image is complete
    if (owner) {
      f16* dst = p.z1 + (size_t)strip * BM * p.mid + un * 8;
      for (int pix = pl; pix < BM; pix += NR) {
        const uint4 v = *reinterpret_cast<const uint4*>(oimg + pix * OP + un * 16);
        *reinterpret_cast<uint4*>(dst + (size_t)pix * p.mid) = v;
        float f[8];
        unpack8h(v, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) { st[e] += f[e]; st[8 + e] += f[e] * f[e]; }
      }
    }
  }
  // ---- statistics: sum the pixel lanes through LDS (the images are free), one slab row [2][mid] per block
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);            // [NR][16][G]
  if (owner) {
#pragma unroll
    for (int e = 0; e < 16; ++e) red[(pl * 16 + e) * G + un] = st[e];
  }
  __syncthreads();
  for (int i = tid; i < 2 * p.mid; i += 256) {
    const int which = i >= p.mid, c = i - which * p.mid;
    const int e = which * 8 + (c & 7), u = c >> 3;
    float t = 0.f;
    for (int r = 0; r < NR; ++r) t += red[(r * 16 + e) * G + u];
    p.parts[(size_t)blockIdx.x * 2 * p.mid + i] = t;
  }
}

extern "C" int mmsim_pw_expand_fwd_eligible(int P, int mid, int cin) {
  return P > 0 && (P % 64) == 0 && (mid % 8) == 0 && (cin % 8) == 0 && cin <= 32 && mid <= 192 && mid >= 64;
}

extern "C" int mmsim_pw_expand_fwd(const void* x, const void* w1_f16, void* z1, float* sums, int P, int mid, int cin, float* scratch,
                                   unsigned long long scratch_floats, void* stream) {
  MMSIM_REQUIRE(x && w1_f16 && z1 && sums && scratch, "pw_expand_fwd: null operand");
  MMSIM_REQUIRE(mmsim_pw_expand_fwd_eligible(P, mid, cin), "pw_expand_fwd: shape not eligible (see mmsim_pw_expand_fwd_eligible)");
  constexpr int BM = 64, CIN_T = 2;
  PwExp p;
  p.x = (const f16*)x; p.w1 = (const f16*)w1_f16; p.z1 = (f16*)z1; p.P = P; p.mid = mid; p.cin = cin;
  p.nstrips = P / BM;
  const int NT = (mid + 15) / 16, XP = CIN_T * 32 + 16, OP = NT * 32 + 16, G = mid / 8, NR = 256 / G;
  size_t lds = (size_t)BM * XP + (size_t)NT * 16 * XP + (size_t)BM * OP;
  const size_t red = (size_t)NR * 16 * G * 4;
  if (red > lds) lds = red;
  MMSIM_REQUIRE(lds <= 64 * 1024, "pw_expand_fwd: LDS images too large");
  int grid = p.nstrips < 1024 ? p.nstrips : 1024;
  while (grid > 64 && (size_t)grid * 2 * mid > scratch_floats) grid /= 2;
  MMSIM_REQUIRE((size_t)grid * 2 * mid <= scratch_floats, "pw_expand_fwd: scratch too small");
  p.per_block = (p.nstrips + grid - 1) / grid;
  grid = (p.nstrips + p.per_block - 1) / p.per_block;
  p.parts = scratch;
  hipLaunchKernelGGL((pw_expand_fwd_kernel<BM, CIN_T>), dim3(grid), dim3(256), lds, (hipStream_t)stream, p);
  mmsim_launch_reduce(scratch, grid, 2 * mid, sums, 1, (hipStream_t)stream);      /* sums += (pre-zeroed by the caller) */
  return mmsim_check_launch("pw_expand_fwd");
}

// Backward of the same conv as one streaming pass over a2:
//   d(a2 * gate)[p][c] = sum_co dz3[p][co] W3[co][c]        (bf16 out, consumed by mmsim_pool_bn_bwd / mmsim_dwtile_bwd)
//   dW3[co][c]        += sum_p  dz3[p][co] (a2[p][c] * gate[p / HW][c])
// Per strip: gated a2 and dz3 into LDS images (rows = pixels); the data gradient (K = cout: one or two MFMAs per 16 x 16 tile) goes
// through an LDS output image so that it leaves as whole 16-byte row chunks; the weight gradient reads both operands with
// transposing LDS reads and stays in registers for the block's lifetime (slab + mmsim_launch_reduce).
struct PwPrjBwd {
  const bf16* dz3; const f16* a2; const float* gate; const bf16* w3; bf16* da; float* parts;      // w3: the bf16 weight shadow
  const float* xsc; const float* xsh;      // as in PwPrj
  int P, HW, B, mid, cout, nstrips, per_block, KS;
  FastDiv dhw;
};

template <int BM, int COUT_T, int NCH, int NCX, int MAXW>
__global__ __launch_bounds__(256) void pw_project_bwd_kernel(PwPrjBwd p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int MTD = BM / 16, MT_W = MTD / 4;            // pixel tiles per strip / per wave
  constexpr int XP = COUT_T * 32 + 16;                    // dz3 image pitch (bytes)
  constexpr int KO = COUT_T / 2;                          // 32-deep steps of the data gradient's reduction over cout
  static_assert(MT_W >= 1 && (COUT_T % 2) == 0, "tile split");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int G = p.mid >> 3, XG = p.cout >> 3, NT = (p.mid + 15) >> 4;
  const int ZP = p.KS * 64 + 16;
  char* zimg = smem;                                      // [BM][KS*32]  gated a2 (rows = pixels)
  char* oimg = zimg + BM * ZP;                            // [BM][KS*32]  d(a2*gate) of the strip
  char* ximg = oimg + BM * ZP;                            // [BM][COUT_T*16]  dz3 (rows = pixels)
  char* wimg = ximg + BM * XP;                            // [COUT_T*16][KS*32]  W3 (rows = cout)
  float* grow = reinterpret_cast<float*>(wimg + COUT_T * 16 * ZP);      // [2][mid]
  float* xs = grow + 2 * p.mid;                           // [2][mid]: scale | shift of the operand transform (if any)
  for (int i = tid; i < (2 * BM * ZP + BM * XP + COUT_T * 16 * ZP) / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);
  if (p.xsc)
    for (int i = tid; i < p.mid; i += 256) { xs[i] = p.xsc[i]; xs[p.mid + i] = p.xsh[i]; }
  __syncthreads();
  for (int i = tid; i < p.cout * G; i += 256) {
    const int r = i / G, c = i - r * G;
    *reinterpret_cast<uint4*>(wimg + r * ZP + c * 16) = *reinterpret_cast<const uint4*>(p.w3 + (size_t)r * p.mid + c * 8);
  }
  const int nwt = COUT_T * NT;                            // weight-gradient tiles (16 x 16), dealt round-robin to the waves
  f4 dW[MAXW];
#pragma unroll
  for (int i = 0; i < MAXW; ++i) dW[i] = f4{0.f, 0.f, 0.f, 0.f};
  const int nchunks = BM * G, nxch = BM * XG;
  uint4 va[NCH], vx[NCX];
  auto request = [&](int strip) {
    const size_t base = (size_t)strip * BM * p.mid, xbase = (size_t)strip * BM * p.cout;
#pragma unroll
    for (int i = 0; i < NCH; ++i) va[i] = *reinterpret_cast<const uint4*>(p.a2 + base + (size_t)min(tid + 256 * i, nchunks - 1) * 8);
#pragma unroll
    for (int i = 0; i < NCX; ++i) vx[i] = *reinterpret_cast<const uint4*>(p.dz3 + xbase + (size_t)min(tid + 256 * i, nxch - 1) * 8);
  };
  const int sbeg = blockIdx.x * p.per_block, send = min(p.nstrips, sbeg + p.per_block);
  int cur_b = -1;
  if (sbeg < send) request(sbeg);
  for (int strip = sbeg; strip < send; ++strip) {
    const int s0 = strip * BM;
    const int b0 = (int)fdiv((unsigned)s0, p.dhw);
    if (b0 != cur_b) {
      for (int i = tid; i < 2 * p.mid; i += 256) {
        const int j = i >= p.mid, c = i - j * p.mid;
        grow[i] = p.gate[(size_t)min(b0 + j, p.B - 1) * p.mid + c];
      }
      cur_b = b0;
    }
    __syncthreads();                     // A: the previous strip's output image has been copied out, its MFMAs are done
    const int split = (b0 + 1) * p.HW - s0;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int q = tid + 256 * i;
      if (q < nchunks) {
        const int pix = q / G, un = q - pix * G;
        const float* gr = grow + (pix >= split ? p.mid : 0) + un * 8;
        const h8 a = __builtin_bit_cast(h8, va[i]);      // fp16 in HBM; staged as bf16: the weight-gradient MFMA pairs it with dz3 (bf16)
        bf8 o;
        if (p.xsc) {
          const float* cs = xs + un * 8;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = f2bf(silu_f(h2f(a[e]) * cs[e] + cs[p.mid + e]) * gr[e]);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = f2bf(h2f(a[e]) * gr[e]);
        }
        *reinterpret_cast<bf8*>(zimg + pix * ZP + un * 16) = o;
      }
    }
#pragma unroll
    for (int i = 0; i < NCX; ++i) {
      const int q = tid + 256 * i;
      if (q < nxch) {
        const int pix = q / XG, un = q - pix * XG;
        *reinterpret_cast<uint4*>(ximg + pix * XP + un * 16) = vx[i];
      }
    }
    __syncthreads();                     // B
    if (strip + 1 < send) request(strip + 1);
    // ---- d(a2*gate) = dz3 W3 into the output image: wave w owns pixel tiles w*MT_W ..., all channel tiles
#pragma unroll
    for (int m = 0; m < MT_W; ++m) {
      const int mt = wave * MT_W + m;
      bf8 xf[KO];
#pragma unroll
      for (int ks = 0; ks < KO; ++ks)
        xf[ks] = *reinterpret_cast<const bf8*>(ximg + (mt * 16 + (lane & 15)) * XP + (ks * 4 + (lane >> 4)) * 16);
      for (int nt = 0; nt < NT; ++nt) {
        f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KO; ++ks) acc = mfma16(tr_frag16(wimg, ZP, ks * 32, nt * 16, lane), xf[ks], acc);
        const bf4 o = {f2bf(acc[0]), f2bf(acc[1]), f2bf(acc[2]), f2bf(acc[3])};      // [pixel lane&15][channel nt*16 + 4(lane>>4) + e]
        *reinterpret_cast<bf4*>(oimg + (mt * 16 + (lane & 15)) * ZP + (nt * 16 + (lane >> 4) * 4) * 2) = o;
      }
    }
    // ---- dW3 += dz3^T (a2*gate): tiles COUT_T x NT, reduction over the strip's pixels
#pragma unroll
    for (int i = 0; i < MAXW; ++i) {
      const int t = wave + 4 * i;
      if (t < nwt) {                     // wave-uniform: EXEC stays all ones for the transposing reads
        const int mt = t / NT, nt = t - mt * NT;
#pragma unroll
        for (int ks = 0; ks < BM / 32; ++ks)
          dW[i] = mfma16(tr_frag16(zimg, ZP, ks * 32, nt * 16, lane), tr_frag16(ximg, XP, ks * 32, mt * 16, lane), dW[i]);
      }
    }
    __syncthreads();                     // C: the output image is complete
    {
      bf16* dst = p.da + (size_t)s0 * p.mid;
      for (int q = tid; q < nchunks; q += 256) {
        const int pix = q / G, un = q - pix * G;
        *reinterpret_cast<uint4*>(dst + (size_t)q * 8) = *reinterpret_cast<const uint4*>(oimg + pix * ZP + un * 16);
      }
    }
  }
  float* slab = p.parts + (size_t)blockIdx.x * p.cout * p.mid;
#pragma unroll
  for (int i = 0; i < MAXW; ++i) {
    const int t = wave + 4 * i;
    if (t < nwt) {
      const int mt = t / NT, nt = t - mt * NT;
      const int co = mt * 16 + (lane & 15), c = nt * 16 + (lane >> 4) * 4;
      if (co < p.cout && c < p.mid) *reinterpret_cast<f4*>(slab + (size_t)co * p.mid + c) = dW[i];      // mid % 8 == 0: whole group valid
    }
  }
}

template <int BM, int COUT_T, int NCH, int NCX, int MAXW>
static int launch_pw_project_bwd(PwPrjBwd p, float* dw3, hipStream_t s, float* scratch, unsigned long long scratch_floats) {
  p.nstrips = p.P / BM;
  p.KS = (p.mid + 31) / 32;
  const int ZP = p.KS * 64 + 16, XP = COUT_T * 32 + 16;
  const size_t lds = (size_t)2 * BM * ZP + (size_t)BM * XP + (size_t)COUT_T * 16 * ZP + (size_t)4 * p.mid * 4;
  MMSIM_REQUIRE(lds <= 100 * 1024, "pw_project_bwd: LDS images too large");
  MMSIM_REQUIRE(BM * (p.mid >> 3) <= 256 * NCH && BM * (p.cout >> 3) <= 256 * NCX, "pw_project_bwd: strip too wide for the staging registers");
  MMSIM_REQUIRE(COUT_T * ((p.mid + 15) / 16) <= 4 * MAXW, "pw_project_bwd: weight gradient does not fit the accumulators");
  const size_t slab = (size_t)p.cout * p.mid;
  int grid = p.nstrips < 1024 ? p.nstrips : 1024;
  while (grid > 64 && (size_t)grid * slab > scratch_floats) grid /= 2;
  MMSIM_REQUIRE((size_t)grid * slab <= scratch_floats, "pw_project_bwd: scratch too small");
  p.per_block = (p.nstrips + grid - 1) / grid;
  grid = (p.nstrips + p.per_block - 1) / p.per_block;
  static unsigned long long done = 0;
  const int dev = mmsim_current_device();
  if (!((done >> dev) & 1)) {
    (void)hipFuncSetAttribute((const void*)pw_project_bwd_kernel<BM, COUT_T, NCH, NCX, MAXW>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    done |= 1ull << dev;
  }
  p.parts = scratch;
  hipLaunchKernelGGL((pw_project_bwd_kernel<BM, COUT_T, NCH, NCX, MAXW>), dim3(grid), dim3(256), lds, s, p);
  mmsim_launch_reduce(scratch, grid, (int)slab, dw3, 1, s);      /* dW3 += */
  return mmsim_check_launch("pw_project_bwd");
}

static int pw_project_bwd_variant(int P, int HW, int mid, int cout) {
  if (P <= 0 || HW <= 0 || (P % HW) || (mid % 8) || (cout % 8)) return 0;
  if (mid <= 48 && cout <= 32 && HW >= 128 && (P % 128) == 0) return 1;        // depthwise-separable blocks at 112^2
  if (mid <= 192 && cout <= 32 && HW >= 64 && (P % 64) == 0) return 2;         // 56^2 stage
  return 0;
}
extern "C" int mmsim_pw_project_bwd_eligible(int P, int HW, int mid, int cout) { return pw_project_bwd_variant(P, HW, mid, cout) != 0; }

static int pw_project_bwd_impl(const void* dz3, const void* a2, const float* xf_scale, const float* xf_shift, const float* gate,
                               const void* w3_bf16, void* da, float* dw3, int P, int HW, int mid, int cout, float* scratch,
                               unsigned long long scratch_floats, void* stream);
extern "C" int mmsim_pw_project_bwd(const void* dz3, const void* a2, const float* gate, const void* w3_bf16, void* da, float* dw3,
                                    int P, int HW, int mid, int cout, float* scratch, unsigned long long scratch_floats, void* stream) {
  return pw_project_bwd_impl(dz3, a2, nullptr, nullptr, gate, w3_bf16, da, dw3, P, HW, mid, cout, scratch, scratch_floats, stream);
}
extern "C" int mmsim_pw_project_bwd_xf(const void* dz3, const void* z2, const float* xf_scale, const float* xf_shift, const float* gate,
                                       const void* w3_bf16, void* da, float* dw3, int P, int HW, int mid, int cout, float* scratch,
                                       unsigned long long scratch_floats, void* stream) {
  MMSIM_REQUIRE(xf_scale && xf_shift, "pw_project_bwd_xf: scale and shift required");
  return pw_project_bwd_impl(dz3, z2, xf_scale, xf_shift, gate, w3_bf16, da, dw3, P, HW, mid, cout, scratch, scratch_floats, stream);
}
static int pw_project_bwd_impl(const void* dz3, const void* a2, const float* xf_scale, const float* xf_shift, const float* gate,
                               const void* w3_bf16, void* da, float* dw3, int P, int HW, int mid, int cout, float* scratch,
                               unsigned long long scratch_floats, void* stream) {
  MMSIM_REQUIRE(dz3 && a2 && gate && w3_bf16 && da && dw3 && scratch, "pw_project_bwd: null operand");
  const int v = pw_project_bwd_variant(P, HW, mid, cout);
  MMSIM_REQUIRE(v != 0, "pw_project_bwd: shape not eligible (see mmsim_pw_project_bwd_eligible)");
  PwPrjBwd p;
  p.xsc = xf_scale; p.xsh = xf_shift;
  p.dz3 = (const bf16*)dz3; p.a2 = (const f16*)a2; p.gate = gate; p.w3 = (const bf16*)w3_bf16; p.da = (bf16*)da;
  p.P = P; p.HW = HW; p.B = P / HW; p.mid = mid; p.cout = cout; p.dhw = make_fastdiv((unsigned)HW);
  if (v == 1) return launch_pw_project_bwd<128, 2, 3, 2, 2>(p, dw3, (hipStream_t)stream, scratch, scratch_floats);
  return launch_pw_project_bwd<64, 2, 6, 1, 6>(p, dw3, (hipStream_t)stream, scratch, scratch_floats);
}

extern "C" int mmsim_pw_expand_bwd_eligible(int P, int mid, int cin) {
  if (P <= 0 || (P % 64) || (mid % 8) || (cin % 8)) return 0;
  if (cin <= 32 && mid <= 192) return 1;
  if (cin <= 64 && mid <= 336 && (P % 32) == 0) return 1;
  return 0;
}

extern "C" int mmsim_pw_expand_bwd(const void* dpre, const void* z1, const void* x, const void* resid, const void* w1_bf16,
                                   const float* scale1, const float* mean1, const float* rstd1, const float* sums1, void* dx,
                                   float* dw1, float* dgamma1, float* dbeta1, int P, int mid, int cin, float* scratch,
                                   unsigned long long scratch_floats, void* stream) {
  MMSIM_REQUIRE(dpre && z1 && x && w1_bf16 && scale1 && mean1 && rstd1 && sums1 && dx && dw1 && dgamma1 && dbeta1 && scratch,
                "pw_expand_bwd: null operand");
  MMSIM_REQUIRE(mmsim_pw_expand_bwd_eligible(P, mid, cin), "pw_expand_bwd: shape not eligible (see mmsim_pw_expand_bwd_eligible)");
  PwBwd p;
  p.dpre = (const bf16*)dpre; p.z1 = (const f16*)z1; p.x = (const f16*)x; p.resid = (const bf16*)resid; p.w1 = (const bf16*)w1_bf16;
  p.sc1 = scale1; p.mu1 = mean1; p.rs1 = rstd1; p.sums1 = sums1; p.dx = (bf16*)dx; p.dgamma = dgamma1; p.dbeta = dbeta1;
  p.P = P; p.mid = mid; p.cin = cin; p.invP = 1.0f / (float)P;
  if (cin <= 32 && mid <= 192) return launch_pw_expand_bwd<64, 2, 6>(p, dw1, (hipStream_t)stream, scratch, scratch_floats);
  // the 28^2 stage (mid = 336): one 8-wave block per CU over 64-pixel strips (110 KiB of images) -- the 4-wave / 32-pixel form ran at 1.4 TB/s
  return launch_pw_expand_bwd<64, 4, 11, 512, 6>(p, dw1, (hipStream_t)stream, scratch, scratch_floats);
}
