// Shared between the generic (register-staged, predicated) and the fast (LDS-DMA, aligned) GEMM kernels.
#pragma once
#include "common.h"

struct GemmParams {
  const bf16* A; const bf16* B; void* C; const float* bias; const bf16* aux_in; bf16* aux_out;
  int M, N, K, lda, ldb, ldc, ld_aux;
  int c_f32, epi, atomic, accum, k_per_split, tiles_m, tiles_n, splits;
  float alpha;
  // optional operand transform (1x1 conv after BN + SiLU + squeeze-excite): x -> silu(scale[c] x + shift[c]) * gate[b, c]
  const float* xf_scale; const float* xf_shift; const float* xf_gate; int xf_hw, xf_C;
  int dbg;   // ablation switches for tools/bench_gemm.py (MMSIM_GEMM_DBG): 1 no DMA, 2 no LDS reads, 4 no MFMA; 0 in production
};

enum { EPI_NONE = 0, EPI_GELU = 1, EPI_MUL_GELU_GRAD = 2, EPI_ADD = 3, EPI_TANH = 4 };

// Epilogue for a wave that owns a 64x64 output sub-tile as acc[4][4] (16x16 MFMA tiles, operands swapped so that
// a lane holds row m = (lane&15), columns n = (lane>>4)*4 + {0..3} of each tile).  row0/col0: the wave's origin.
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f4 (&acc)[4][4], int row0, int col0, int lane, bool first_split) {
  const int m0 = row0, n0 = col0, wm = 0, wn = 0;
  const bool add_bias = (p.bias != nullptr) && first_split;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + (lane & 15);
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
      if (n >= p.N) continue;
      float v[4] = {acc[i][j][0] * p.alpha, acc[i][j][1] * p.alpha, acc[i][j][2] * p.alpha, acc[i][j][3] * p.alpha};
      const bool full = (n + 3 < p.N);
      if (add_bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < p.N) v[e] += p.bias[n + e];
      }
      if (p.epi == EPI_GELU) {
        bf16* ao = p.aux_out + (size_t)m * p.ld_aux + n;
        if (full) {
          bf4 pre = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
          *reinterpret_cast<bf4*>(ao) = pre;
        } else {
          for (int e = 0; e < 4; ++e)
            if (n + e < p.N) ao[e] = f2bf(v[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_f(bf2f(f2bf(v[e])));   // gelu of the stored (rounded) pre-activation
      } else if (p.epi == EPI_MUL_GELU_GRAD) {
        const bf16* ai = p.aux_in + (size_t)m * p.ld_aux + n;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < p.N) v[e] *= gelu_grad_f(bf2f(ai[e]));
      } else if (p.epi == EPI_ADD) {
        const bf16* ai = p.aux_in + (size_t)m * p.ld_aux + n;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < p.N) v[e] += bf2f(ai[e]);
      } else if (p.epi == EPI_TANH) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = tanhf(v[e]);
      }
      if (p.c_f32) {
        float* c = reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n;
        if (p.atomic) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < p.N) atomicAdd(c + e, v[e]);
        } else if (full) {
          float4 o = make_float4(v[0], v[1], v[2], v[3]);
          if (p.accum) {
            const float4 old = *reinterpret_cast<const float4*>(c);
            o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
          }
          *reinterpret_cast<float4*>(c) = o;
        } else {
          for (int e = 0; e < 4; ++e)
            if (n + e < p.N) c[e] = p.accum ? c[e] + v[e] : v[e];
        }
      } else {
        bf16* c = reinterpret_cast<bf16*>(p.C) + (size_t)m * p.ldc + n;
        if (full) {
          bf4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
          *reinterpret_cast<bf4*>(c) = o;
        } else {
          for (int e = 0; e < 4; ++e)
            if (n + e < p.N) c[e] = f2bf(v[e]);
        }
      }
    }
  }
}
