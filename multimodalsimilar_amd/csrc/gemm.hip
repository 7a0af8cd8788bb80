// bf16 MFMA GEMM for gfx950:  C[M,N] = op(A)[M,K] * op(B)[K,N]  (+ fused epilogues), fp32 accumulate.
//
// One kernel template serves the three products of a Linear / 1x1-conv layer without any transposed
// copies in HBM:
//   forward   Y  = X  W^T      A k-major [M][K],            B k-major  [N][K]   (TA=0, TB=1)
//   dgrad     dX = dY W        A k-major [M][N_out as K],   B row-major [K][N]  (TA=0, TB=0)
//   wgrad     dW = dY^T X      A stored  [K][M],            B stored   [K][N]   (TA=1, TB=0)
// k-major operands are staged [128 rows][64 k] (160-B pitch, conflict-free ds_read_b128); operands whose
// reduction index is the slow (row) index are staged as they lie in memory, [64 k][128 cols] with a 32-B
// column-block XOR swizzle, and fed to the MFMA through the hardware transposing read ds_read_b64_tr_b16.
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 v_mfma_f32_16x16x32_bf16 tiles.
// The MFMA is issued with the operands swapped (D^T = B^T A^T) so every lane owns 4 CONSECUTIVE output
// columns of one row: 8-byte bf16 / 16-byte f32 stores and float4 bias loads.
// Global->LDS staging is register-staged and split (issue loads for tile t+1, compute tile t, then write):
// one barrier per K-step on double-buffered LDS.  Blocks are remapped so that the tiles an XCD works on
// are neighbours (shared A panel in that XCD's L2).
#include "gemm_common.h"
#include <stdlib.h>
#include <string.h>

#define BM 128
#define BN 128
#define BK 64
#define KM_PITCH 160                       // bytes per k-major row (128 B data + 32 B pad)
#define OP_STAGE_BYTES (BM * KM_PITCH)     // 20480 >= 64*256 (transposed-staged operand)
#define STAGE_BYTES (2 * OP_STAGE_BYTES)

__device__ __forceinline__ int tr_key(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

// Ragged tiles are handled by clamping the ADDRESS into the operand and AND-masking the VALUE: `if (in range) v = load`
// makes hipcc branch around each of the 8 loads of a K-step and wait vmcnt(0) after every one (the loads then complete
// one L2 round trip after the other).
// The mask is a SEPARATE step (mask_tile), applied where the registers are parked in LDS: applied with the load -- as it was up to
// r03 -- the AND is a use of the loaded value inside the prefetch's own basic block, hipcc put its vmcnt waits there, IN FRONT of
// the K-step's MFMAs, and the "prefetch" was waited for at once: every K-step of every image-tower 1x1 convolution paid a full
// memory round trip (found in the ISA after the same pattern showed up in the attention backward's stamps).
template <bool TRANS>
__device__ __forceinline__ void load_tile(const bf16* __restrict__ X, int ld, int r0, int R, int k0, int kend,
                                          int tid, uint4 (&reg)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i;
    const bf16* src;
    if (!TRANS) {
      const int row = c >> 3, kc = c & 7;
      const int gr = r0 + row, gk = k0 + kc * 8;
      src = X + (size_t)min(gr, R - 1) * ld + (gk < kend ? gk : k0);
    } else {
      const int krow = c >> 4, rc = c & 15;
      const int gk = k0 + krow, gr = r0 + rc * 8;
      src = X + (size_t)min(gk, kend - 1) * ld + (gr < R ? gr : r0);
    }
    reg[i] = *reinterpret_cast<const uint4*>(src);
  }
}
template <bool TRANS>
__device__ __forceinline__ void mask_tile(int r0, int R, int k0, int kend, int tid, uint4 (&reg)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i;
    bool ok;
    if (!TRANS) ok = r0 + (c >> 3) < R && k0 + (c & 7) * 8 < kend;
    else ok = k0 + (c >> 4) < kend && r0 + (c & 15) * 8 < R;
    const unsigned int msk = ok ? 0xffffffffu : 0u;
    reg[i] = make_uint4(reg[i].x & msk, reg[i].y & msk, reg[i].z & msk, reg[i].w & msk);
  }
}

template <bool TRANS>
__device__ __forceinline__ void store_tile(char* lds, int tid, const uint4 (&reg)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i;
    if (!TRANS) {
      const int row = c >> 3, kc = c & 7;
      *reinterpret_cast<uint4*>(lds + row * KM_PITCH + kc * 16) = reg[i];
    } else {
      const int krow = c >> 4, rc = c & 15;
      const int pos = (((rc >> 1) ^ tr_key(krow)) << 1) | (rc & 1);
      *reinterpret_cast<uint4*>(lds + krow * 256 + pos * 16) = reg[i];
    }
  }
}

// BN-affine + SiLU (+ SE gate) applied to a staged tile in registers.  The "pixel" index is the row of a
// k-major operand / the k index of a transposed one; the channel index is the other one.
// The per-(image, channel) gate values are REQUESTED with the tile's own loads (xform_request, ahead of the current step's
// MFMAs) and applied after them (xform_tile): requested at the point of use they cost every K-step an exposed L2 round trip.
struct XfGate { float4 g0[4], g1[4]; };
template <bool TRANS>
__device__ __forceinline__ void xform_request(const GemmParams& p, int r0, int R, int k0, int kend, int tid, XfGate& xg) {
  if (!p.xf_gate) return;                // launch-uniform
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i;
    int pix, ch;
    if (!TRANS) { pix = r0 + (c >> 3); ch = k0 + (c & 7) * 8; if (pix >= R) pix = R - 1; if (ch >= kend) ch = k0; }
    else { pix = k0 + (c >> 4); ch = r0 + (c & 15) * 8; if (pix >= kend) pix = kend - 1; if (ch >= R) ch = r0; }
    const float* gp = p.xf_gate + (size_t)fdiv((unsigned int)pix, p.xf_dhw) * p.xf_C + ch;
    xg.g0[i] = *reinterpret_cast<const float4*>(gp);
    xg.g1[i] = *reinterpret_cast<const float4*>(gp + 4);
  }
}

// FMT (GemmParams::fmt): 0 bf16 in / bf16 out, 1 fp16 in / fp16 out, 2 fp16 in / bf16 out (the weight gradient's activation operand)
template <int FMT> __device__ __forceinline__ float xf_in(const uint4& v, int e) {
  if (FMT == 0) return bf2f(__builtin_bit_cast(bf8, v)[e]);
  return h2f(__builtin_bit_cast(h8, v)[e]);
}
template <int FMT> __device__ __forceinline__ uint4 xf_out(const float (&f)[8]) {
  if (FMT == 1) { h8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2h(f[e]);
    return __builtin_bit_cast(uint4, o); }
  bf8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = f2bf(f[e]);
  return __builtin_bit_cast(uint4, o);
}
// fmt 2 without an operand transform: the fp16 activation tile becomes bf16 in registers on its way to LDS (zeros stay zeros)
__device__ __forceinline__ void convert_tile_f16_bf16(uint4 (&reg)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = xf_in<2>(reg[i], e);
    reg[i] = xf_out<2>(f);
  }
}

template <bool TRANS, int FMT>
__device__ __forceinline__ void xform_tile(const GemmParams& p, int r0, int R, int k0, int kend, int tid, uint4 (&reg)[4], const XfGate& xg) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i;
    int pix, ch;
    bool ok;
    if (!TRANS) { pix = r0 + (c >> 3); ch = k0 + (c & 7) * 8; ok = pix < R && ch < kend; if (pix >= R) pix = R - 1; if (ch >= kend) ch = k0; }
    else { pix = k0 + (c >> 4); ch = r0 + (c & 15) * 8; ok = pix < kend && ch < R; if (pix >= kend) pix = kend - 1; if (ch >= R) ch = r0; }
    const uint4 v = reg[i];
    const float g[8] = {xg.g0[i].x, xg.g0[i].y, xg.g0[i].z, xg.g0[i].w, xg.g1[i].x, xg.g1[i].y, xg.g1[i].z, xg.g1[i].w};
    if (!p.xf_scale) {                   // launch-uniform: the operand is already activated, only the SE gate remains
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = xf_in<FMT>(v, e) * g[e];      // masked (zero) elements stay zero
      reg[i] = xf_out<FMT>(o);
      continue;
    }
    const float4 s0 = *reinterpret_cast<const float4*>(p.xf_scale + ch), s1 = *reinterpret_cast<const float4*>(p.xf_scale + ch + 4);
    const float4 h0 = *reinterpret_cast<const float4*>(p.xf_shift + ch), h1 = *reinterpret_cast<const float4*>(p.xf_shift + ch + 4);
    const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    const float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = silu_f(xf_in<FMT>(v, e) * sc[e] + sh[e]) * (p.xf_gate ? g[e] : 1.f);
    const uint4 ov = xf_out<FMT>(o);
    const unsigned int msk = ok ? 0xffffffffu : 0u;      // out-of-range elements must stay zero (silu(shift) != 0)
    reg[i] = make_uint4(ov.x & msk, ov.y & msk, ov.z & msk, ov.w & msk);
  }
}

// fragment of 16 "rows" (row index = lane&15) x 32 k for k-step ks, rows starting at rbase (multiple of 16)
template <bool TRANS>
__device__ __forceinline__ bf8 read_frag(const char* lds, int rbase, int ks, int lane) {
  if (!TRANS) {
    const int row = rbase + (lane & 15), kc = ks * 4 + (lane >> 4);
    return *reinterpret_cast<const bf8*>(lds + row * KM_PITCH + kc * 16);
  } else {
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int k = ks * 32 + 8 * g + q;
    const int cb = rbase >> 4;
    const int off = k * 256 + ((cb ^ tr_key(k)) << 5) + p * 8;
    typedef s4 __attribute__((address_space(3))) * lds_s4_ptr;
    s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(lds + off));
    s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(lds + off + 4 * 256));
    s8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return __builtin_bit_cast(bf8, r);
  }
}

// The epilogue of the generic kernels (gemm_body, gemm_ring_body): a wave's 64 x 64 sub-tile leaves through its LDS staging region
// (whole cache-line row segments); `smem` is free when this is called (the caller's last barrier is behind every operand read).
template <int FMT>
__device__ __forceinline__ void gemm_tail(const GemmParams& p, f4 (&acc)[4][4], const bool fs, const int tm, const int m0, const int n0,
                                          const int wm, const int wn, const int wave, const int lane, char* smem) {
  const int row0 = m0 + wm * 64, col0 = n0 + wn * 64;
  float* stg = reinterpret_cast<float*>(smem) + wave * (64 * EP_PITCH);
  if (p.stats) {                     // launch-uniform: bf16 output + BatchNorm column statistics of this M-tile
    float cs[4], cq[4];
    stats_epilogue<FMT == 1>(p, acc, row0, col0, lane, stg, cs, cq);
    __syncthreads();                 // all staging reads done: the region is reused for the cross-wave sum
    float* red = reinterpret_cast<float*>(smem);          // [wave][2][64]
    if (lane < 16) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { red[(wave * 2 + 0) * 64 + lane * 4 + e] = cs[e]; red[(wave * 2 + 1) * 64 + lane * 4 + e] = cq[e]; }
    }
    __syncthreads();
    if (wm == 0) {                   // waves 0, 1 (wn = 0, 1) add their partner wave (wm = 1) and write 64 columns each
      const int n = col0 + lane;
      if (n < p.N) {
        float* slab = p.stats + (size_t)tm * 2 * p.N;
        slab[n] = red[(wave * 2 + 0) * 64 + lane] + red[((wave + 2) * 2 + 0) * 64 + lane];
        slab[p.N + n] = red[(wave * 2 + 1) * 64 + lane] + red[((wave + 2) * 2 + 1) * 64 + lane];
      }
    }
    return;
  }
  if (FMT == 1 && !p.c_f32) { f16_epilogue(p, acc, row0, col0, lane, stg); return; }
  if (!p.c_f32) fast_epilogue_epi<0, true>(p, acc, row0, col0, lane, fs, stg);
  else if (p.atomic) fast_epilogue<EPI_NONE, 3, true>(p, acc, row0, col0, lane, fs, stg);
  else if (p.accum && p.epi == EPI_ROWFIX) fast_epilogue<EPI_ROWFIX, 2, true>(p, acc, row0, col0, lane, fs, stg);
  else if (p.accum) fast_epilogue<EPI_NONE, 2, true>(p, acc, row0, col0, lane, fs, stg);
  else if (p.epi == EPI_TANH) fast_epilogue<EPI_TANH, 1, true>(p, acc, row0, col0, lane, fs, stg);
  else if (p.epi == EPI_NONE) fast_epilogue<EPI_NONE, 1, true>(p, acc, row0, col0, lane, fs, stg);
  else if (p.epi == EPI_ROWFIX) fast_epilogue<EPI_ROWFIX, 1, true>(p, acc, row0, col0, lane, fs, stg);
  else gemm_epilogue(p, acc, row0, col0, lane, fs);          // f32 output with a fused epilogue: direct form
}

template <bool TA, bool TB_KMAJOR, int XF, int FMT = 0>   // XF: 0 none, 1 transform A, 2 transform B; FMT: GemmParams::fmt
__device__ __forceinline__ void gemm_body(const GemmParams& p, const int bid, char* smem) {
  static_assert(FMT != 2 || (XF != 1), "fmt 2 converts the B operand");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware bijective remap of the 1-D grid onto tiles (bid & 7 == blockIdx.x & 7 also in the paired launch: the second
  // product's blocks start at a multiple of 8)
  const int ntiles = p.tiles_m * p.tiles_n;
  const int nwg = ntiles * p.splits;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const int split = wg / ntiles, tile = wg - split * ntiles;
  const int tm = tile / p.tiles_n, tn = tile % p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = split * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int nk = (kend - kbeg + BK - 1) / BK;

  f4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};

  // One K-step of register-staged prefetch.  Two steps ahead (a second register set, loop unrolled by two) was built and measured
  // on the image tower's late-stage shapes (tools/bench_imgemm2.py, operands from HBM): 6.25 vs 5.75 ms over the 20 blocks -- SLOWER
  // (254-256 VGPRs, hipcc waits for the younger set's loads before parking the older one); removed.
  uint4 ra[4], rb[4];
  XfGate xg;
  load_tile<TA>(p.A, p.lda, m0, p.M, kbeg, kend, tid, ra);
  load_tile<!TB_KMAJOR>(p.B, p.ldb, n0, p.N, kbeg, kend, tid, rb);
  mask_tile<TA>(m0, p.M, kbeg, kend, tid, ra);
  mask_tile<!TB_KMAJOR>(n0, p.N, kbeg, kend, tid, rb);
  if (XF == 1) { xform_request<TA>(p, m0, p.M, kbeg, kend, tid, xg); xform_tile<TA, FMT>(p, m0, p.M, kbeg, kend, tid, ra, xg); }
  if (XF == 2) { xform_request<!TB_KMAJOR>(p, n0, p.N, kbeg, kend, tid, xg); xform_tile<!TB_KMAJOR, FMT>(p, n0, p.N, kbeg, kend, tid, rb, xg); }
  if (FMT == 2 && XF == 0) convert_tile_f16_bf16(rb);
  store_tile<TA>(smem, tid, ra);
  store_tile<!TB_KMAJOR>(smem + OP_STAGE_BYTES, tid, rb);
  __syncthreads();

  for (int t = 0; t < nk; ++t) {
    const int cur = t & 1;
    const char* la = smem + cur * STAGE_BYTES;
    const char* lb = la + OP_STAGE_BYTES;
    if (t + 1 < nk) {
      load_tile<TA>(p.A, p.lda, m0, p.M, kbeg + (t + 1) * BK, kend, tid, ra);
      load_tile<!TB_KMAJOR>(p.B, p.ldb, n0, p.N, kbeg + (t + 1) * BK, kend, tid, rb);
      if (XF == 1) xform_request<TA>(p, m0, p.M, kbeg + (t + 1) * BK, kend, tid, xg);
      if (XF == 2) xform_request<!TB_KMAJOR>(p, n0, p.N, kbeg + (t + 1) * BK, kend, tid, xg);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = read_frag<TA>(la, wm * 64 + i * 16, ks, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = read_frag<!TB_KMAJOR>(lb, wn * 64 + j * 16, ks, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = FMT == 1 ? __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, bfr[j]), __builtin_bit_cast(h8, af[i]), acc[i][j], 0, 0, 0)
                               : __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
    if (t + 1 < nk) {
      mask_tile<TA>(m0, p.M, kbeg + (t + 1) * BK, kend, tid, ra);
      mask_tile<!TB_KMAJOR>(n0, p.N, kbeg + (t + 1) * BK, kend, tid, rb);
      if (XF == 1) xform_tile<TA, FMT>(p, m0, p.M, kbeg + (t + 1) * BK, kend, tid, ra, xg);
      if (XF == 2) xform_tile<!TB_KMAJOR, FMT>(p, n0, p.N, kbeg + (t + 1) * BK, kend, tid, rb, xg);
      if (FMT == 2 && XF == 0) convert_tile_f16_bf16(rb);
      char* na = smem + (cur ^ 1) * STAGE_BYTES;
      store_tile<TA>(na, tid, ra);
      store_tile<!TB_KMAJOR>(na + OP_STAGE_BYTES, tid, rb);
    }
    __syncthreads();
  }

  // staged epilogue (whole cache-line row segments; the operand stages are free after the last barrier)
  gemm_tail<FMT>(p, acc, split == 0, tm, m0, n0, wm, wn, wave, lane, smem);
}

// =========================================================================================================================
// RING form of the 128 x 128 generic product (round 4): the same tile, wave layout and epilogues as gemm_body, but the
// operands arrive by LDS-DMA through a ring of 32-deep stages (16 KiB: A 128 rows x 64 B | B 128 x 64 B or 32 k-rows x 256 B) with
// TWO OR THREE STAGES IN FLIGHT per workgroup.  gemm_body keeps ONE register-staged K-step in flight: at the image tower's late-stage
// shapes (14 x 14 / 7 x 7 planes: M = 50 176 / 12 544 pixels, 112-448 / 672-2 688 channels) a K-step costs a memory round trip
// (~2 us against ~0.3 us of MFMAs) and the products ran at 0.8-2.9 TB/s of operand + result bytes.  Two workgroups per CU (<= 80 KiB).
//   iteration t:  vmcnt -> my pieces of stage t landed | barrier (everybody's landed; everybody is done reading stage t - 1)
//                 | issue stage t + NS - 1 into the slot of stage t - 1 | fragments of stage t | 16 MFMAs per wave
// Ragged edges: rows past M / N re-read the last row (their outputs are predicated off in the epilogues), 16-byte chunks past the K
// range and columns past N of a transposed operand read a zero chunk instead (g_rg_zero) -- the DMA cannot mask.
// GATE (forward projection conv): A = a2 * gate[pixel / HW][channel]; the gate rows of the <= 4 images a tile touches are staged once
// per tile as fp16 [4][Kp] and multiplied into the A fragments (v_pk_mul_f16) -- the product of two fp16 values rounded to fp16, where
// gemm_body's xform_tile multiplies by the fp32 gate: one more rounding of the gate (2^-11 relative).
typedef void __attribute__((address_space(3))) * rg_lds_ptr;
typedef const void __attribute__((address_space(1))) * rg_glb_ptr;
#define RG_STG 16384
__device__ __attribute__((aligned(64))) const unsigned int g_rg_zero[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

__device__ __forceinline__ bf8 rg_read_b128(uint32_t a, int imm_unused = 0) {
  bf8 r;
  asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(a));
  return r;
}
template <int OFF>
__device__ __forceinline__ bf8 rg_read_b128_off(uint32_t a) {
  bf8 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(a), "n"(OFF));
  return r;
}
template <int OFF>
__device__ __forceinline__ bf8 rg_read_tr_off(uint32_t a) {     // k rows 8g + q (lo) and + 4 (hi) of a 256-byte-row image
  s4 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(a), "n"(OFF));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a), "n"(OFF + 4 * 256));
  return __builtin_bit_cast(bf8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// per-lane source state of one operand: two 1-KiB pieces per wave and stage
struct RgSrc { const char* base[2]; int kofs; int kstep; bool colok[2]; };

template <bool TRANS>      // TRANS = false: X[row][k] (k contiguous); true: X[k][col]
__device__ __forceinline__ void rg_src_init(RgSrc& o, const bf16* X, int ld, int r0, int R, int kbeg, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int blk = wave * 2 + i;
    if (!TRANS) {
      const int row = blk * 16 + (lane >> 2), c = (lane & 3) ^ ((lane >> 4) & 3);
      o.base[i] = reinterpret_cast<const char*>(X + (size_t)min(r0 + row, R - 1) * ld + kbeg + c * 8);
      o.kofs = c * 8; o.colok[i] = true;
    } else {
      const int k = blk * 4 + (lane >> 4), pc = lane & 15;
      const int rc = (((pc >> 1) ^ tr_key(k)) << 1) | (pc & 1);
      const int col = r0 + rc * 8;
      o.colok[i] = col < R;
      o.base[i] = reinterpret_cast<const char*>(X + (size_t)(kbeg + k) * ld + (col < R ? col : r0));
      o.kofs = k;          // differs per piece: recomputed in rg_issue (blk * 4 + (lane >> 4))
    }
  }
  o.kstep = TRANS ? ld * 64 : 64;      // bytes per 32-deep stage
}
// stage t of the operand into `dst` (the operand's 8 KiB of a ring slot); kleft = kend - (kbeg + 32 t) > 0
template <bool TRANS>
__device__ __forceinline__ void rg_issue(const RgSrc& o, int t, int kleft, char* dst, int wave, int lane) {
  const char* zero = reinterpret_cast<const char*>(g_rg_zero) + (lane & 3) * 16;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int blk = wave * 2 + i;
    const char* src = o.base[i] + (size_t)t * o.kstep;
    bool ok;
    if (!TRANS) ok = o.kofs < kleft;
    else ok = (blk * 4 + (lane >> 4)) < kleft && o.colok[i];
    if (!ok) src = zero;
    __builtin_amdgcn_global_load_lds((rg_glb_ptr)src, (rg_lds_ptr)(dst + blk * 1024), 16, 0, 0);
  }
}

template <bool TB_KMAJOR, int FMT, bool GATE>
__device__ __forceinline__ void gemm_ring_body(const GemmParams& p, const int bid, char* smem, const int ns) {
  static_assert(!GATE || (FMT == 1 && TB_KMAJOR), "the fragment-side gate is the fp16 forward layout's");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int ntiles = p.tiles_m * p.tiles_n;
  const int nwg = ntiles * p.splits;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const int split = wg / ntiles, tile = wg - split * ntiles;
  const int tm = tile / p.tiles_n, tn = tile % p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = split * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int nk = (kend - kbeg + 31) >> 5;

  f4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};

  // the gate rows of this tile's images, fp16 [4][Kp] behind the ring (ordinary loads: requested first, so the counted waits below
  // only ever see the LDS-DMA pieces as the youngest operations)
  const int Kp = (p.K + 31) & ~31;
  uint32_t gaddr[4] = {0, 0, 0, 0};
  if (GATE) {
    char* gl = smem + ns * RG_STG;
    const int i0 = (int)fdiv((unsigned int)m0, p.xf_dhw);
    const int nimg = (int)fdiv((unsigned int)min(m0 + BM - 1, p.M - 1), p.xf_dhw) - i0 + 1;
    for (int e = tid * 8; e < 4 * Kp; e += 256 * 8) {
      const int img = e / Kp, k = e - img * Kp;
      h8 o = {0, 0, 0, 0, 0, 0, 0, 0};
      if (img < nimg && k < p.K) {
        const float* gp = p.xf_gate + (size_t)(i0 + img) * p.xf_C + k;
        const float4 g0 = *reinterpret_cast<const float4*>(gp), g1 = *reinterpret_cast<const float4*>(gp + 4);
        o = (h8){f2h(g0.x), f2h(g0.y), f2h(g0.z), f2h(g0.w), f2h(g1.x), f2h(g1.y), f2h(g1.z), f2h(g1.w)};
      }
      *reinterpret_cast<h8*>(gl + (size_t)e * 2) = o;
    }
    __syncthreads();         // the loop's bare s_barrier carries no wait for these stores
    const uint32_t gl0 = (uint32_t)(uintptr_t)(lds_s4_ptr)gl;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = min(m0 + wm * 64 + i * 16 + (lane & 15), p.M - 1);
      const int li = (int)fdiv((unsigned int)r, p.xf_dhw) - i0;
      gaddr[i] = gl0 + (uint32_t)((li * Kp + (lane >> 4) * 8) * 2);
    }
  }

  RgSrc sa, sb;
  rg_src_init<false>(sa, p.A, p.lda, m0, p.M, kbeg, wave, lane);
  rg_src_init<!TB_KMAJOR>(sb, p.B, p.ldb, n0, p.N, kbeg, wave, lane);
  const int klen = kend - kbeg;
  for (int t = 0; t < ns; ++t) {
    if (t < nk) {
      rg_issue<false>(sa, t, klen - 32 * t, smem + t * RG_STG, wave, lane);
      rg_issue<!TB_KMAJOR>(sb, t, klen - 32 * t, smem + t * RG_STG + 8192, wave, lane);
    }
  }
  // fragment read addresses: A rows (lane & 15) of the wave's 64, 16-byte chunk (lane >> 4) ^ ((row >> 2) & 3); B alike, or transposed
  // (read_frag<true>: k = 8 g + q, 16-column block cb = wn * 4 + j at ((cb ^ tr_key(k)) << 5): j enters through the XOR)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_s4_ptr)smem;
  const uint32_t fo = (uint32_t)((lane & 15) * 64 + (((lane >> 4) ^ ((lane >> 2) & 3)) << 4));
  const uint32_t a_rd = lds0 + wm * 4096 + fo;
  const int tk = 8 * (lane >> 4) + ((lane >> 2) & 3);             // tr_key(tk) == tr_key(tk + 4): bit 2 is not part of the key
  const uint32_t b_rd = TB_KMAJOR ? lds0 + 8192 + wn * 4096 + fo : lds0 + 8192 + (uint32_t)(tk * 256 + (lane & 3) * 8);
  uint32_t b_tr[4] = {0, 0, 0, 0};
  if (!TB_KMAJOR) {
#pragma unroll
    for (int j = 0; j < 4; ++j) b_tr[j] = (uint32_t)(((wn * 4 + j) ^ tr_key(tk)) << 5);
  }
  bf8 af[2][4], bfr[2][4];
  h8 gv[2][4];
  // fragments (and gate chunks) of stage T from ring slot SLOT into register set SET
#define RG_READ(SET, T, SLOT)                                                                                          \
  {                                                                                                                    \
    const uint32_t so = (uint32_t)(SLOT) * RG_STG;                                                                     \
    af[SET][0] = rg_read_b128_off<0>(a_rd + so); af[SET][1] = rg_read_b128_off<1024>(a_rd + so);                       \
    af[SET][2] = rg_read_b128_off<2048>(a_rd + so); af[SET][3] = rg_read_b128_off<3072>(a_rd + so);                    \
    if (TB_KMAJOR) {                                                                                                   \
      bfr[SET][0] = rg_read_b128_off<0>(b_rd + so); bfr[SET][1] = rg_read_b128_off<1024>(b_rd + so);                   \
      bfr[SET][2] = rg_read_b128_off<2048>(b_rd + so); bfr[SET][3] = rg_read_b128_off<3072>(b_rd + so);                \
    } else {                                                                                                           \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) bfr[SET][j] = rg_read_tr_off<0>(b_rd + so + b_tr[j]);              \
    }                                                                                                                  \
    if (GATE) {                                                                                                        \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) gv[SET][i] = __builtin_bit_cast(h8, rg_read_b128(gaddr[i] + (uint32_t)(T) * 64)); \
    }                                                                                                                  \
  }
#define RG_WAIT(YOUNGER)                                                        \
  {                                                                             \
    const int y_ = (YOUNGER);                                                   \
    if (y_ >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");              \
    else if (y_ == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          \
    else if (y_ == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");          \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       \
  }
  // One step: the MFMAs of stage t run on register set CUR while the fragments of stage t + 1 are read into set NXT behind them and
  // stage t + NS is requested into the slot stage t has just left (its fragments are in registers, everybody's: the barrier).
#define RG_STEP(CUR, NXT)                                                                                              \
  {                                                                                                                    \
    if (t + 1 < nk) RG_WAIT(min(ns - 2, nk - 2 - t))                  /* stage t + 1 landed (my pieces) */              \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                /* set CUR complete */                            \
    __builtin_amdgcn_s_barrier();                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
    int nslot = slot + 1; if (nslot == ns) nslot = 0;                                                                  \
    if (t + 1 < nk) RG_READ(NXT, t + 1, nslot)                                                                         \
    if (t + ns < nk) {                                                                                                 \
      rg_issue<false>(sa, t + ns, klen - 32 * (t + ns), smem + slot * RG_STG, wave, lane);                             \
      rg_issue<!TB_KMAJOR>(sb, t + ns, klen - 32 * (t + ns), smem + slot * RG_STG + 8192, wave, lane);                 \
    }                                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
    if (GATE) {                                                                                                        \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) af[CUR][i] = __builtin_bit_cast(bf8, __builtin_bit_cast(h8, af[CUR][i]) * gv[CUR][i]); \
    }                                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                      \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                    \
        acc[i][j] = FMT == 1 ? __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, bfr[CUR][j]), __builtin_bit_cast(h8, af[CUR][i]), acc[i][j], 0, 0, 0) \
                             : __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[CUR][j], af[CUR][i], acc[i][j], 0, 0, 0);   \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
    slot = nslot;                                                                                                      \
  }
  RG_WAIT(min(ns - 1, nk - 1))                 // stage 0 landed
  __builtin_amdgcn_s_barrier();
  RG_READ(0, 0, 0)
  int slot = 0;
  for (int t = 0; t < nk; t += 2) {
    RG_STEP(0, 1)
    if (t + 1 < nk) { ++t; RG_STEP(1, 0) --t; }
  }
#undef RG_STEP
#undef RG_WAIT
#undef RG_READ
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();           // every wave is past its last fragment read (and no DMA is in flight: the last wait was vmcnt(0))
  gemm_tail<FMT>(p, acc, split == 0, tm, m0, n0, wm, wn, wave, lane, smem);
}

template <bool TB_KMAJOR, int FMT, bool GATE>
__global__ __launch_bounds__(256, 2) void gemm_ring_kernel(GemmParams p, int ns) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  gemm_ring_body<TB_KMAJOR, FMT, GATE>(p, blockIdx.x, smem, ns);
}

template <bool TA, bool TB_KMAJOR, int XF = 0, int FMT = 0>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  gemm_body<TA, TB_KMAJOR, XF, FMT>(p, blockIdx.x, smem);
}

// Two products of one layer's backward in ONE launch: the weight gradient (A^T B, split-K; XF = 2 applies the operand transform
// to B) in blocks [0, nwg1) and the data gradient (A B, bf16 output with its epilogue) in blocks [n1r, n1r + nwg2), n1r = nwg1
// rounded up to 8.  At 14 x 14 / 7 x 7 each of them alone runs 1.3-2.5 rounds of tiles on the chip; together the data
// gradient's short blocks fill the tail of the weight gradient's long ones, and a layer issues one launch instead of two.
template <int XF, int FMT1 = 0>      // FMT1: format of the weight-gradient product (2: its activation operand is fp16); the data gradient is bf16
__global__ __launch_bounds__(256, 2) void gemm_bwd_pair_kernel(GemmParams p1, GemmParams p2, int nwg1, int n1r, int ns2) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int bid = blockIdx.x;
  if (bid < n1r) {
    if (bid < nwg1) gemm_body<true, false, XF, FMT1>(p1, bid, smem);
  } else if (ns2) {                  // launch-uniform: the data gradient in the ring form (ring_stages)
    gemm_ring_body<false, 0, false>(p2, bid - n1r, smem, ns2);
  } else {
    gemm_body<false, false, 0>(p2, bid - n1r, smem);
  }
}

// C-ABI -- see include/mmsim_hip.h for the contract.
bool gemm_fast_eligible(const GemmParams& p, int splits);
bool gemm_fast_rowfix(const GemmParams& p, int splits, int trans_a, int b_kmajor);
void gemm_fast_launch(GemmParams p, int trans_a, int b_kmajor, int splits, hipStream_t s);

static bool force_generic() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("MMSIM_GEMM_GENERIC"); v = (e && e[0] == '1') ? 1 : 0; }
  return v == 1;
}

// ---- paired launches (mmsim_gemm_group_begin / _end): see gemm_bwd_pair_kernel
struct GroupItem { GemmParams p; int trans_a, b_kmajor, xf, nwg; hipStream_t s; };
struct GroupState { bool active = false; int n = 0; GroupItem item[2]; };
static thread_local GroupState g_group;
static int launch_generic(const GemmParams& p, int trans_a, int b_kmajor, int xf_operand, int nwg, hipStream_t s);
static void generic_attr_optin();
static int ring_stages(const GemmParams& p, int trans_a, int b_kmajor, int xf_operand, size_t* lds_bytes);

static int group_flush() {
  GroupState& g = g_group;
  const int n = g.n;
  g.n = 0;
  if (n == 2) {
    const GroupItem &a = g.item[0], &b = g.item[1];
    const bool wgrad_first = a.trans_a && !a.b_kmajor && (a.xf == 0 || a.xf == 2);
    const bool dgrad_second = !b.trans_a && !b.b_kmajor && b.xf == 0 && b.p.fmt == 0;
    if (wgrad_first && dgrad_second && a.s == b.s && (a.p.fmt == 0 || a.p.fmt == 2)) {
      hipStream_t s = a.s;
      generic_attr_optin();
      const int n1r = (a.nwg + 7) & ~7;
      dim3 grid(n1r + b.nwg), block(256);
      const size_t lds = 2 * STAGE_BYTES;
      size_t rl = 0;
      int ns2 = ring_stages(b.p, 0, 0, 0, &rl);
      if (rl > lds) ns2 = 0;
      if (a.xf == 2 && a.p.fmt == 2) hipLaunchKernelGGL((gemm_bwd_pair_kernel<2, 2>), grid, block, lds, s, a.p, b.p, a.nwg, n1r, ns2);
      else if (a.xf == 2) hipLaunchKernelGGL((gemm_bwd_pair_kernel<2, 0>), grid, block, lds, s, a.p, b.p, a.nwg, n1r, ns2);
      else if (a.p.fmt == 2) hipLaunchKernelGGL((gemm_bwd_pair_kernel<0, 2>), grid, block, lds, s, a.p, b.p, a.nwg, n1r, ns2);
      else hipLaunchKernelGGL((gemm_bwd_pair_kernel<0, 0>), grid, block, lds, s, a.p, b.p, a.nwg, n1r, ns2);
      return mmsim_check_launch("gemm_bwd_pair");
    }
  }
  for (int i = 0; i < n; ++i) {
    const GroupItem& it = g.item[i];
    const int rc = launch_generic(it.p, it.trans_a, it.b_kmajor, it.xf, it.nwg, it.s);
    if (rc) return rc;
  }
  return 0;
}

static int gemm_impl(int trans_a, int b_kmajor, int M, int N, int K, const void* A, int lda, const void* B,
                     int ldb, void* C, int ldc, int c_is_f32, const float* bias, int epilogue,
                     const void* aux_in, void* aux_out, int ld_aux, float alpha, int split_k,
                     int accumulate, int xf_operand, const float* xf_scale, const float* xf_shift, const float* xf_gate,
                     int xf_hw, void* stream, float* stats = nullptr, float* colsum = nullptr, int fmt = 0,
                     const long long* arc_label = nullptr, float* arc_part = nullptr, int arc_C = 0, const Margin* arc_m = nullptr) {
  MMSIM_REQUIRE(M > 0 && N > 0 && K > 0, "gemm: M, N, K must be positive");
  MMSIM_REQUIRE(fmt >= 0 && fmt <= 2, "gemm: fmt must be 0 (bf16), 1 (fp16 operands) or 2 (fp16 B converted to bf16)");
  MMSIM_REQUIRE(fmt != 1 || (!trans_a && b_kmajor && xf_operand != 2 && split_k == 1 && (c_is_f32 || (epilogue == EPI_NONE && !bias))),
                "gemm: fmt 1 is the forward layout only (A, B k-major, no split-K); an fp16 output takes no bias / epilogue");
  MMSIM_REQUIRE(fmt != 2 || (trans_a && !b_kmajor && xf_operand != 1 && c_is_f32), "gemm: fmt 2 is the weight-gradient layout only (A^T B, f32 output)");
  MMSIM_REQUIRE(A && B && C, "gemm: null operand");
  MMSIM_REQUIRE((lda % 8) == 0 && (ldb % 8) == 0, "gemm: lda/ldb must be multiples of 8 elements (16-byte rows)");
  MMSIM_REQUIRE((ldc % 4) == 0, "gemm: ldc must be a multiple of 4");
  MMSIM_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0 && ((uintptr_t)C % 16) == 0,
                "gemm: operands must be 16-byte aligned");
  MMSIM_REQUIRE((epilogue >= 0 && epilogue <= 7) || (epilogue == EPI_ARCSTATS && arc_label && arc_part && arc_m), "gemm: unknown epilogue");
  MMSIM_REQUIRE(epilogue != EPI_ROWFIX || (c_is_f32 && split_k == 1 && bias && aux_in),
                "gemm: the row-fix epilogue needs an f32 output, no split-K, the [2][M] row vectors in bias and aux_in");
  MMSIM_REQUIRE(!(epilogue == EPI_GELU || epilogue == EPI_GELU_DGELU) || aux_out, "gemm: GELU epilogues need aux_out (pre-activation / gelu')");
  MMSIM_REQUIRE(!(epilogue == EPI_MUL_GELU_GRAD || epilogue == EPI_ADD || epilogue == EPI_MUL) || aux_in, "gemm: epilogue needs aux_in");
  MMSIM_REQUIRE(epilogue == EPI_NONE || (ld_aux % 4) == 0 || epilogue == EPI_TANH, "gemm: ld_aux must be a multiple of 4");
  MMSIM_REQUIRE(split_k >= 1, "gemm: split_k >= 1");
  if (mmsim_deterministic()) split_k = 1;          // one adder per output element: the "atomic" epilogue is then order-free
  MMSIM_REQUIRE(split_k == 1 || (c_is_f32 && epilogue == EPI_NONE), "gemm: split-K needs f32 output and no epilogue");
  MMSIM_REQUIRE(!accumulate || c_is_f32, "gemm: accumulate needs f32 output");
  MMSIM_REQUIRE(!accumulate || epilogue == EPI_NONE || epilogue == EPI_ROWFIX, "gemm: accumulate with this epilogue is not implemented");
  // leading dimensions must cover the extents rounded up to the 8-element load granule
  const int a_cols = trans_a ? M : K, b_cols = b_kmajor ? K : N;
  MMSIM_REQUIRE(lda >= ((a_cols + 7) & ~7) && ldb >= ((b_cols + 7) & ~7), "gemm: leading dimension too small");
  GemmParams p;
  p.A = (const bf16*)A; p.B = (const bf16*)B; p.C = C; p.bias = bias;
  p.aux_in = (const bf16*)aux_in; p.aux_out = (bf16*)aux_out;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ld_aux = ld_aux;
  p.c_f32 = c_is_f32; p.epi = epilogue; p.atomic = split_k > 1; p.accum = accumulate; p.alpha = alpha;
  p.xf_scale = xf_scale; p.xf_shift = xf_shift; p.xf_gate = xf_gate; p.xf_hw = xf_hw > 0 ? xf_hw : 1; p.xf_dhw = make_fastdiv(p.xf_hw);
  p.xf_C = (xf_operand == 1) ? K : N;
  p.stats = stats; p.band = 1; p.colsum = colsum; p.fmt = fmt;
  p.arc_label = arc_label; p.arc_part = arc_part; p.arc_C = arc_C;
  if (arc_m) p.arc_m = *arc_m; else memset(&p.arc_m, 0, sizeof(p.arc_m));
#ifdef MMSIM_ABLATE     // ablation object only (tools/bench_gemm_abl.py); the product library never reads this variable
  { static int dbg = -1; if (dbg < 0) { const char* e = getenv("MMSIM_GEMM_DBG"); dbg = e ? atoi(e) : 0; } p.dbg = dbg; }
#else
  p.dbg = 0;
#endif
  p.tiles_m = (M + BM - 1) / BM; p.tiles_n = (N + BN - 1) / BN;
  int kps = (K + split_k - 1) / split_k;
  kps = ((kps + BK - 1) / BK) * BK;
  p.k_per_split = kps;
  const int splits = (K + kps - 1) / kps;
  p.splits = splits;
  dim3 grid(p.tiles_m * p.tiles_n * splits), block(256);
  const size_t lds = 2 * STAGE_BYTES;
  hipStream_t s = (hipStream_t)stream;
  MMSIM_REQUIRE(epilogue != EPI_ARCSTATS || (xf_operand == 0 && !stats && fmt == 0 && !force_generic() && gemm_fast_eligible(p, splits)),
                "gemm: the softmax-statistics epilogue needs the pipelined 256 x 256 forward kernel");
  MMSIM_REQUIRE(!colsum || (trans_a && !b_kmajor && xf_operand == 0 && !stats && fmt == 0 && !force_generic() && gemm_fast_eligible(p, splits)),
                "gemm: the fused column sum needs the pipelined weight-gradient kernel");
  if (xf_operand == 0 && !stats && fmt == 0 && !force_generic() && (gemm_fast_eligible(p, splits) || gemm_fast_rowfix(p, splits, trans_a, b_kmajor))) {
    if (g_group.active) { const int rc = group_flush(); if (rc) return rc; }
    gemm_fast_launch(p, trans_a, b_kmajor, splits, s);
    return mmsim_check_launch("gemm_bf16_fast");
  }
  if (g_group.active) {                 // inside mmsim_gemm_group_begin / _end: generic-path products are parked, not launched
    if (g_group.n < 2 && !stats) {
      GroupItem& it = g_group.item[g_group.n++];
      it.p = p; it.trans_a = trans_a; it.b_kmajor = b_kmajor; it.xf = xf_operand; it.nwg = (int)grid.x; it.s = s;
      return 0;
    }
    const int rc = group_flush();      // anything else keeps program order: what is parked goes first
    if (rc) return rc;
  }
  return launch_generic(p, trans_a, b_kmajor, xf_operand, (int)grid.x, s);
}

static void generic_attr_optin() {
  const size_t lds = 2 * STAGE_BYTES;
  static unsigned long long attr_done = 0;          // per device
  const int dev = mmsim_current_device();
  if (!((attr_done >> dev) & 1)) {   // 80 KiB of dynamic LDS per block needs the opt-in on every instantiation
    (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel<false, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel<true, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel<false, true, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel<false, true, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel<true, false, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel<true, false, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_bwd_pair_kernel<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_bwd_pair_kernel<2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_bwd_pair_kernel<0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_bwd_pair_kernel<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done |= 1ull << dev;
  }
}

// Which products take the ring form, and with how many stages (0 = gemm_body): the forward layout (fp16 with an optional gate-only
// operand transform, or bf16) and the data-gradient layout (bf16, B stored [K][N]), K a multiple of 8 and at least three stages deep.
// MMSIM_GEMM_RING=0 sends everything to gemm_body (A/B runs).
static int ring_stages(const GemmParams& p, int trans_a, int b_kmajor, int xf_operand, size_t* lds_bytes) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("MMSIM_GEMM_RING"); on = e ? atoi(e) : 1; }
  if (!on || trans_a || (p.K % 8) || p.k_per_split < 96 || p.fmt == 2) return 0;
  if (!b_kmajor && (p.fmt != 0 || xf_operand != 0)) return 0;
  size_t gate = 0;
  if (xf_operand == 1) {
    if (p.xf_scale || !p.xf_gate || p.fmt != 1 || p.xf_hw < 43) return 0;       // <= 4 images per 128-row tile
    gate = (size_t)4 * ((p.K + 31) & ~31) * 2;
  } else if (xf_operand != 0) return 0;
  const size_t epi = (size_t)4 * 64 * EP_PITCH * sizeof(float);
  int ns = 4;
  if (ns * RG_STG + gate > 80 * 1024) ns = 3;
  if (ns * RG_STG + gate > 80 * 1024) return 0;
  size_t b = ns * RG_STG + gate;
  if (b < epi) b = epi;
  *lds_bytes = b;
  return ns;
}
static void ring_attr_optin() {
  static unsigned long long done = 0;
  const int dev = mmsim_current_device();
  if ((done >> dev) & 1) return;
  const int lds = 80 * 1024;
  (void)hipFuncSetAttribute((const void*)gemm_ring_kernel<true, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  (void)hipFuncSetAttribute((const void*)gemm_ring_kernel<true, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  (void)hipFuncSetAttribute((const void*)gemm_ring_kernel<true, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  (void)hipFuncSetAttribute((const void*)gemm_ring_kernel<false, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  done |= 1ull << dev;
}

static int launch_generic(const GemmParams& p, int trans_a, int b_kmajor, int xf_operand, int nwg, hipStream_t s) {
  {
    size_t rl = 0;
    const int ns = ring_stages(p, trans_a, b_kmajor, xf_operand, &rl);
    if (ns) {
      ring_attr_optin();
      dim3 grid(nwg), block(256);
      if (!b_kmajor) hipLaunchKernelGGL((gemm_ring_kernel<false, 0, false>), grid, block, rl, s, p, ns);
      else if (p.fmt == 1 && xf_operand == 1) hipLaunchKernelGGL((gemm_ring_kernel<true, 1, true>), grid, block, rl, s, p, ns);
      else if (p.fmt == 1) hipLaunchKernelGGL((gemm_ring_kernel<true, 1, false>), grid, block, rl, s, p, ns);
      else hipLaunchKernelGGL((gemm_ring_kernel<true, 0, false>), grid, block, rl, s, p, ns);
      return mmsim_check_launch("gemm_ring");
    }
  }
  generic_attr_optin();
  const size_t lds = 2 * STAGE_BYTES;
  dim3 grid(nwg), block(256);
  if (p.fmt == 1) {                // fp16 forward products (gemm_impl checked the layout)
    if (xf_operand == 1) hipLaunchKernelGGL((gemm_bf16_kernel<false, true, 1, 1>), grid, block, lds, s, p);
    else hipLaunchKernelGGL((gemm_bf16_kernel<false, true, 0, 1>), grid, block, lds, s, p);
  } else if (p.fmt == 2) {         // weight gradients with an fp16 activation operand
    if (xf_operand == 2) hipLaunchKernelGGL((gemm_bf16_kernel<true, false, 2, 2>), grid, block, lds, s, p);
    else hipLaunchKernelGGL((gemm_bf16_kernel<true, false, 0, 2>), grid, block, lds, s, p);
  } else if (xf_operand == 1) {
    hipLaunchKernelGGL((gemm_bf16_kernel<false, true, 1>), grid, block, lds, s, p);
  } else if (xf_operand == 2) {
    hipLaunchKernelGGL((gemm_bf16_kernel<true, false, 2>), grid, block, lds, s, p);
  } else if (!trans_a && b_kmajor) {
    hipLaunchKernelGGL((gemm_bf16_kernel<false, true>), grid, block, lds, s, p);
  } else if (!trans_a && !b_kmajor) {
    hipLaunchKernelGGL((gemm_bf16_kernel<false, false>), grid, block, lds, s, p);
  } else if (trans_a && !b_kmajor) {
    hipLaunchKernelGGL((gemm_bf16_kernel<true, false>), grid, block, lds, s, p);
  } else {
    hipLaunchKernelGGL((gemm_bf16_kernel<true, true>), grid, block, lds, s, p);
  }
  return mmsim_check_launch("gemm_bf16");
}

// Products issued between _begin and _end on the calling thread that take the generic kernel are parked (at most two) and
// launched by _end: a (weight gradient, data gradient) pair as ONE launch of gemm_bwd_pair_kernel, anything else one by one in
// program order.  Any other product (fast path, BatchNorm-statistics epilogue) first flushes what is parked.
extern "C" int mmsim_gemm_group_begin(void) {
  MMSIM_REQUIRE(!g_group.active, "gemm_group_begin: a group is already open on this thread");
  g_group.active = true; g_group.n = 0;
  return 0;
}
extern "C" int mmsim_gemm_group_end(void) {
  MMSIM_REQUIRE(g_group.active, "gemm_group_end: no open group");
  const int rc = group_flush();
  g_group.active = false;
  return rc;
}

// Weight gradient of a dense layer AND its bias gradient from one pass over dY:  C[M,N] (fp32) += A^T B  with A = dY stored [K][M],
// B = X stored [K][N] (split-K atomics), and colsum[M] += sum_k A[k][m].  Only for products the pipelined 256 x 256 kernel takes
// (M, N multiples of 256, K of 64 per split, >= 128 tiles x splits); the caller falls back to mmsim_gemm_bf16 + mmsim_colsum_bf16 when
// mmsim_gemm_bf16_wgrad_colsum_eligible says no.
extern "C" int mmsim_gemm_bf16_wgrad_colsum_eligible(int M, int N, int K, int split_k) {
  if (M <= 0 || N <= 0 || K <= 0 || split_k < 1 || (M % 256) || (N % 256)) return 0;
  if (mmsim_deterministic()) split_k = 1;
  int kps = (K + split_k - 1) / split_k;
  kps = ((kps + BK - 1) / BK) * BK;
  const int splits = (K + kps - 1) / kps;
  if ((K % 64) || (kps % 64)) return 0;
  const int t256 = (M / 256) * (N / 256);
  return (t256 >= 32 && t256 * splits >= 160) ? 1 : 0;      // the shapes gemm_fast_launch sends to gemm_pp64_kernel<true,false,256>
}
extern "C" int mmsim_gemm_bf16_wgrad_colsum(int M, int N, int K, const void* A, int lda, const void* B, int ldb, float* C, int ldc,
                                            float* colsum, int split_k, void* stream) {
  MMSIM_REQUIRE(colsum, "gemm_wgrad_colsum: colsum required");
  MMSIM_REQUIRE(mmsim_gemm_bf16_wgrad_colsum_eligible(M, N, K, split_k), "gemm_wgrad_colsum: shape not eligible (see _eligible)");
  MMSIM_REQUIRE(!force_generic(), "gemm_wgrad_colsum: MMSIM_GEMM_GENERIC=1 disables the pipelined kernel this entry point needs");
  return gemm_impl(1, 0, M, N, K, A, lda, B, ldb, C, ldc, 1, nullptr, 0, nullptr, nullptr, 0, 1.0f, split_k, 1, 0, nullptr, nullptr,
                   nullptr, 1, stream, nullptr, colsum);
}

extern "C" int mmsim_gemm_bf16(int trans_a, int b_kmajor, int M, int N, int K, const void* A, int lda, const void* B,
                               int ldb, void* C, int ldc, int c_is_f32, const float* bias, int epilogue,
                               const void* aux_in, void* aux_out, int ld_aux, float alpha, int split_k,
                               int accumulate, void* stream) {
  return gemm_impl(trans_a, b_kmajor, M, N, K, A, lda, B, ldb, C, ldc, c_is_f32, bias, epilogue, aux_in, aux_out, ld_aux,
                   alpha, split_k, accumulate, 0, nullptr, nullptr, nullptr, 1, stream);
}

// The fused ArcFace forward (head.py forward_loss; arcface.py:47-61 + nn.CrossEntropyLoss, multimodal_classifier_train.py:188):
//   cos[B, ld] = x_hat w_hat^T  (bf16 MFMA, f32 out; w_hat has ld rows, those >= C zero),
//   per row: lse, loss, argmax, the target's logit and margin slope -- from per-segment online-softmax statistics that the cosine
//   product's own epilogue leaves (pipelined 256 x 256 kernel: B % 256 == 0, ld % 256 == 0, D % 64 == 0, >= 128 tiles) or, for
//   other shapes, one pass of arcface_stats_kernel over the cosines; then the mean loss.
// part: scratch of >= B * ceil(ld / 64) * 4 floats.  rowst [B][4] = {lse, target logit, margin slope, -} is what
// mmsim_arcface_dcos_rowfix reads in the backward.  No logits / softmax / one-hot tensor exists; the cosines are written once here
// and read once in the backward.
bool gemm_fast_arcstats(const GemmParams& p);
extern "C" int mmsim_arcface_fwd_fused(const void* x_hat, const void* w_hat, float* cosm, int ld, const long long* label, float* part,
                                       unsigned long long part_floats, float* rowst, float* loss_b, long long* argmax, float* loss_mean,
                                       int B, int C, int D, float s, float m, int easy_margin, int* err_flag, void* stream) {
  MMSIM_REQUIRE(x_hat && w_hat && cosm && label && part && rowst && loss_b && err_flag, "arcface_fwd_fused: null operand");
  MMSIM_REQUIRE(B > 0 && C > 0 && D > 0 && ld >= C && (ld % 8) == 0 && (D % 8) == 0, "arcface_fwd_fused: ld >= C, ld and D multiples of 8");
  const Margin mg = mk_margin(s, m, easy_margin);
  GemmParams probe;
  memset(&probe, 0, sizeof(probe));
  probe.M = B; probe.N = ld; probe.K = D; probe.ldc = ld; probe.c_f32 = 1; probe.epi = EPI_ARCSTATS; probe.k_per_split = ((D + BK - 1) / BK) * BK;
  const bool fused = !force_generic() && !g_group.active && gemm_fast_arcstats(probe);
  const int nseg = fused ? ld / 64 : (C + 1023) / 1024;
  MMSIM_REQUIRE(part_floats >= (unsigned long long)B * nseg * 4, "arcface_fwd_fused: statistics scratch too small (B * ceil(ld / 64) * 4 floats)");
  int rc = gemm_impl(0, 1, B, ld, D, x_hat, D, w_hat, D, cosm, ld, 1, nullptr, fused ? EPI_ARCSTATS : EPI_NONE, nullptr, nullptr, 0, 1.0f, 1, 0, 0,
                     nullptr, nullptr, nullptr, 1, stream, nullptr, nullptr, 0, fused ? label : nullptr, fused ? part : nullptr, C, fused ? &mg : nullptr);
  if (rc) return rc;
  if (!fused) { rc = arcface_stats_launch(cosm, ld, label, part, B, C, nseg, mg, (hipStream_t)stream); if (rc) return rc; }
  return arcface_combine_launch(part, nseg, cosm, ld, label, rowst, loss_b, argmax, loss_mean, B, C, mg, err_flag, (hipStream_t)stream);
}

// The same product with an explicit element format (GemmParams::fmt): 1 = fp16 A, B (and C unless f32), forward layout only;
// 2 = weight gradient whose B operand (the activation) is fp16 and is converted to bf16 while staged; 0 = mmsim_gemm_bf16.
extern "C" int mmsim_gemm_fmt(int fmt, int trans_a, int b_kmajor, int M, int N, int K, const void* A, int lda, const void* B,
                              int ldb, void* C, int ldc, int c_is_f32, const float* bias, int epilogue,
                              const void* aux_in, void* aux_out, int ld_aux, float alpha, int split_k,
                              int accumulate, void* stream) {
  return gemm_impl(trans_a, b_kmajor, M, N, K, A, lda, B, ldb, C, ldc, c_is_f32, bias, epilogue, aux_in, aux_out, ld_aux,
                   alpha, split_k, accumulate, 0, nullptr, nullptr, nullptr, 1, stream, nullptr, nullptr, fmt);
}

// 1x1 conv whose input is silu(scale*z + shift) * gate[pixel / hw, channel] applied on the fly while staging:
//   xf_operand = 1: forward   C[P, Cout] = xf(A)[P, Cin] * B[Cout, Cin]^T        (A k-major, B k-major)
//   xf_operand = 2: wgrad     C[Cout, Cin] (+)= A[P, Cout]^T * xf(B)[P, Cin]      (both stored [P][.])
extern "C" int mmsim_gemm_bf16_xf(int xf_operand, int M, int N, int K, const void* A, int lda, const void* B, int ldb,
                                  void* C, int ldc, int c_is_f32, const float* xf_scale, const float* xf_shift,
                                  const float* xf_gate, int xf_hw, int split_k, int accumulate, void* stream) {
  MMSIM_REQUIRE(xf_operand == 1 || xf_operand == 2, "gemm_xf: xf_operand must be 1 (A, forward) or 2 (B, wgrad)");
  MMSIM_REQUIRE((xf_scale && xf_shift) || (!xf_scale && !xf_shift && xf_gate), "gemm_xf: scale and shift, or neither with a gate (gate-only operand)");
  MMSIM_REQUIRE(((xf_operand == 1 ? K : N) % 8) == 0, "gemm_xf: transformed channel count must be a multiple of 8");
  return gemm_impl(xf_operand == 2, xf_operand == 1, M, N, K, A, lda, B, ldb, C, ldc, c_is_f32, nullptr, 0, nullptr, nullptr, 0,
                   1.0f, split_k, accumulate, xf_operand, xf_scale, xf_shift, xf_gate, xf_hw, stream, nullptr, nullptr, xf_operand);
}

void mmsim_launch_reduce(const float* parts, int nparts, int n, float* out, int accumulate, hipStream_t s);   // conv.hip

// 1x1 conv (optionally with the BN + SiLU (+ gate) operand transform, xf_operand = 1) whose epilogue also accumulates the
// per-channel sum / sum of squares of the bf16-rounded output: the train-mode BatchNorm statistics of the conv output.
extern "C" int mmsim_gemm_bf16_bnstats(int xf_operand, int M, int N, int K, const void* A, int lda, const void* B, int ldb,
                                       void* C, int ldc, const float* xf_scale, const float* xf_shift, const float* xf_gate,
                                       int xf_hw, float* sums, float* scratch, unsigned long long scratch_floats, void* stream) {
  MMSIM_REQUIRE(xf_operand == 0 || xf_operand == 1, "gemm_bnstats: xf_operand must be 0 (plain) or 1 (transform A)");
  MMSIM_REQUIRE(xf_operand == 0 || (xf_scale && xf_shift) || (!xf_scale && !xf_shift && xf_gate),
                "gemm_bnstats: xf_operand 1 takes scale and shift, or neither with a gate (gate-only operand)");
  MMSIM_REQUIRE(sums && (N % 8) == 0, "gemm_bnstats: sums required, N must be a multiple of 8");
  const int tiles_m = (M + BM - 1) / BM;
  MMSIM_REQUIRE(scratch && (unsigned long long)tiles_m * 2 * N <= scratch_floats, "gemm_bnstats: scratch too small (need ceil(M/128)*2*N floats)");
  const int rc = gemm_impl(0, 1, M, N, K, A, lda, B, ldb, C, ldc, 0, nullptr, 0, nullptr, nullptr, 0, 1.0f, 1, 0, xf_operand,
                           xf_scale, xf_shift, xf_gate, xf_hw, stream, scratch, nullptr, 1);
  if (rc) return rc;
  mmsim_launch_reduce(scratch, tiles_m, 2 * N, sums, 1, (hipStream_t)stream);
  return mmsim_check_launch("gemm_bnstats");
}
