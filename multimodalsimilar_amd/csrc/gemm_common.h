// Shared between the generic (register-staged, predicated) and the fast (LDS-DMA, aligned) GEMM kernels.
#pragma once
#include "common.h"

struct GemmParams {
  const bf16* A; const bf16* B; void* C; const float* bias; const bf16* aux_in; bf16* aux_out;
  int M, N, K, lda, ldb, ldc, ld_aux;
  int c_f32, epi, atomic, accum, k_per_split, tiles_m, tiles_n, splits;
  float alpha;
  // optional operand transform (1x1 conv after BN + SiLU + squeeze-excite): x -> silu(scale[c] x + shift[c]) * gate[b, c]
  const float* xf_scale; const float* xf_shift; const float* xf_gate; int xf_hw, xf_C; FastDiv xf_dhw;
  int band;       // gemm_pp64_kernel: tile-rows per band of the tile walk
  float* stats;   // gemm_bf16_kernel, bf16 output, no split: per-M-tile column sum / sumsq slab [tiles_m][2][N] (NULL = off)
  // gemm_pp64_kernel<.., GRP = true>: a SECOND product of the same N, K, layouts and split count in the same launch -- tile-rows
  // >= tiles_m1 belong to it (A2 / B2 / C2).  Used to run the attention-output and the q|k|v weight gradients of a layer as
  // one 256-block launch (16 + 48 output tiles x 4 splits) instead of two launches that each under-fill the chip.
  const bf16* A2; const bf16* B2; void* C2; int lda2, ldb2, ldc2, tiles_m1;
  // gemm_pp64_kernel<true, false, 256>: column sums of the transposed operand A ([K][M]: sum over k of A[k][m]) accumulated
  // atomically into colsum[M] by the tile column 0 blocks -- the bias gradient of a dense layer out of its weight-gradient
  // product (A = dY), instead of a separate pass over dY (NULL = off)
  float* colsum;
  // element format of the generic kernel's operands (gemm_bf16_kernel; the pipelined kernels are bf16 only):
  //   0  bf16 A, B, bf16 / f32 C
  //   1  fp16 A, B, fp16 / f32 C, v_mfma_f32_16x16x32_f16 (image-tower forward: activations and the weight shadow are fp16)
  //   2  bf16 A, fp16 B converted to bf16 while it is staged, f32 C (image-tower weight gradients: A = the bf16 gradient, B = the
  //      fp16 activation)
  int fmt;
  // EPI_ARCSTATS (forward layout, f32 C = the cosines [B, ldc]; gemm_pp64_kernel<false, true, 256> only): besides storing the cosines the
  // epilogue leaves, per output row and per 64-column segment, the online-softmax statistics of the scaled margin logits
  // (arcface.py:49-61 + the cross-entropy's log-sum-exp): arc_part[(m * N / 64 + col / 64)] = {max, sum exp(z - max), argmax (int bits), 0}.
  // Columns >= arc_C (the zero pad rows of w_hat) are left out; the margin is applied at column arc_label[m].
  const long long* arc_label; float* arc_part; int arc_C; Margin arc_m;
  int dbg;   // ablation object only (-DMMSIM_ABLATE): 1 no DMA, 4 no MFMA, 8 no epilogue; the product build ignores it
};

// EPI_ROWFIX: C[m][n] (+)= r[m] * (acc - aux_in[m][n] * r[M + m]) with the two fp32 row vectors r passed in `bias` ([2][M]):
// the backward of a row L2-normalisation folded into the product that feeds it (ArcFace weight gradient, head.py).
// EPI_GELU_DGELU / EPI_MUL: the GELU pair the text tower uses.  Forward: out = gelu(pre) and aux_out = gelu'(pre) (the erf is
// shared, one more exp) -- so the dgrad epilogue is ONE multiply per element by a prefetched operand instead of erf + exp on
// the GEMM's critical path (EPI_MUL_GELU_GRAD recomputes gelu' from the stored pre-activation: ~7 us of VALU per 256x256 tile).
enum { EPI_NONE = 0, EPI_GELU = 1, EPI_MUL_GELU_GRAD = 2, EPI_ADD = 3, EPI_TANH = 4, EPI_ROWFIX = 5, EPI_GELU_DGELU = 6, EPI_MUL = 7, EPI_ARCSTATS = 8 };

// gelu(x) = x Phi(x) and gelu'(x) = Phi(x) + x phi(x) from ONE exponential: Phi through the rational-times-Gaussian form of erfc
// (Abramowitz & Stegun 7.1.26: |error| <= 1.5e-7 on erf, i.e. fp32-level -- libm's erff is itself ~1e-7), whose Gaussian factor
// exp(-x^2 / 2) is exactly the one phi needs.  1 v_exp + 1 v_rcp + ~12 FMA per element instead of erff (~30) + exp: the erf was
// ~8 us of VALU per 256 x 256 tile on the FFN forward product's critical path.
__device__ __forceinline__ void gelu_both_f(float x, float& g, float& dg) {
  const float ax = fabsf(x);
  const float E = __expf(-0.5f * x * x);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * 0.70710678118654752f * ax);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float q = 0.5f * poly * E;                  // 1 - Phi(|x|)
  const float cdf = x >= 0.f ? 1.0f - q : q;
  g = x * cdf;
  dg = cdf + x * (0.39894228040143268f * E);
}

// Two elements at a time on the packed fp32 pipes (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: one issue slot for two lanes' worth of
// arithmetic): the polynomial, the products and the final combinations are 2-wide, only |x|, the two transcendentals and the sign select
// stay per element.  The GELU-pair epilogue is VALU work with the matrix pipes idle (~25 instructions per element on 134 M elements per
// launch), so issue slots are what it costs.
#ifndef GELU_PK
#define GELU_PK 1
#endif
typedef float __attribute__((ext_vector_type(2))) f2v;
__device__ __forceinline__ void gelu_both_pk(f2v x, f2v& g, f2v& dg) {
  const f2v ax = {fabsf(x[0]), fabsf(x[1])};
  const f2v xx = x * x * (-0.5f * 1.4426950408889634f);                  // exp(-x^2 / 2) = exp2(-x^2 / 2 * log2 e): no separate scaling multiply
  const f2v E = {__builtin_amdgcn_exp2f(xx[0]), __builtin_amdgcn_exp2f(xx[1])};
  const f2v den = ax * (0.3275911f * 0.70710678118654752f) + 1.0f;
  const f2v t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
  f2v poly = t * 1.061405429f + -1.453152027f;
  poly = poly * t + 1.421413741f;
  poly = poly * t + -0.284496736f;
  poly = poly * t + 0.254829592f;
  const f2v q = poly * t * E * 0.5f;                                      // 1 - Phi(|x|)
  const f2v cdf = {x[0] >= 0.f ? 1.0f - q[0] : q[0], x[1] >= 0.f ? 1.0f - q[1] : q[1]};
  g = x * cdf;
  dg = x * (E * 0.39894228040143268f) + cdf;
}

// Epilogue for full tiles, staged through LDS so that every global access is row-contiguous: the wave's 64x64
// fp32 tile is written to its private LDS region (the operand ring is free by then) and read back one row segment
// per 16 lanes: 128-B bf16 / 256-B fp32 stores and aux loads (whole cache lines), and for split-K one 256-B
// contiguous atomic wave-instruction per row (the shape float atomics run at full rate in, MI355X_MICROARCH.md).
// CMODE: 0 bf16 store, 1 f32 store, 2 f32 +=, 3 f32 atomic add.
#define EP_PITCH 68     // floats per staged row (64 + 4 pad: 16-byte aligned rows)
template <int EPI, int CMODE, bool CHECK = false>     // CHECK: predicate rows / columns against M, N (ragged tiles)
__device__ __forceinline__ void fast_epilogue(const GemmParams& p, f4 (&acc)[4][4], int row0, int col0, int lane, bool first_split,
                                              float* tile) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<f4*>(tile + (i * 16 + (lane & 15)) * EP_PITCH + j * 16 + (lane >> 4) * 4) = acc[i][j];
  if (CMODE == 3) {
#pragma unroll 8
    for (int r = 0; r < 64; ++r) {
      if (CHECK && (row0 + r >= p.M || col0 + lane >= p.N)) continue;
      atomicAdd(reinterpret_cast<float*>(p.C) + (size_t)(row0 + r) * p.ldc + col0 + lane, tile[r * EP_PITCH + lane] * p.alpha);
    }
    return;
  }
  const int c4 = (lane & 15) * 4, n = col0 + c4;
  // Epilogues that READ a [M, N] operand (GELU' / residual / multiplier): on full tiles all 16 row segments of the lane are
  // requested up front -- ahead of the staging round trip through LDS -- instead of four at a time inside the row loop (each
  // group of four paid a full memory round trip with the matrix pipe idle: ~30 us per 256x256 tile on the FFN dgrad product).
  constexpr bool PRE = !CHECK && CMODE != 3 && (EPI == EPI_MUL_GELU_GRAD || EPI == EPI_ADD || EPI == EPI_MUL);
  bf4 auxv[16];
  if (PRE) {
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const size_t m = (size_t)(row0 + it * 4 + (lane >> 4));
      if (EPI == EPI_ADD) auxv[it] = *reinterpret_cast<const bf4*>(p.aux_in + m * p.ld_aux + n);
      else auxv[it] = __builtin_nontemporal_load(reinterpret_cast<const bf4*>(p.aux_in + m * p.ld_aux + n));      // read once
    }
  }
  const bool add_bias = (p.bias != nullptr) && first_split && EPI != EPI_ROWFIX;
  float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
  if (add_bias) {
    if (!CHECK || n + 3 < p.N) bias = *reinterpret_cast<const float4*>(p.bias + n);
    else { if (n < p.N) bias.x = p.bias[n]; if (n + 1 < p.N) bias.y = p.bias[n + 1]; if (n + 2 < p.N) bias.z = p.bias[n + 2]; }
  }
  if (CHECK && n >= p.N) return;
  if (CHECK && n + 3 >= p.N) {            // ragged column tail (N % 4 != 0): element-wise, rare
    for (int r = lane >> 4; r < 64; r += 4) {
      if (row0 + r >= p.M) continue;
      const size_t m = (size_t)(row0 + r);
      const float bb[4] = {bias.x, bias.y, bias.z, bias.w};
      for (int e = 0; e < 4 && n + e < p.N; ++e) {
        float v = tile[r * EP_PITCH + c4 + e] * p.alpha + bb[e];
        if (EPI == EPI_GELU) { const bf16 pre = f2bf(v); p.aux_out[m * p.ld_aux + n + e] = pre; v = gelu_f(bf2f(pre)); }
        else if (EPI == EPI_GELU_DGELU) { float g, dg; gelu_both_f(bf2f(f2bf(v)), g, dg); p.aux_out[m * p.ld_aux + n + e] = f2bf(dg); v = g; }
        else if (EPI == EPI_MUL_GELU_GRAD) v *= gelu_grad_f(bf2f(p.aux_in[m * p.ld_aux + n + e]));
        else if (EPI == EPI_MUL) v *= bf2f(p.aux_in[m * p.ld_aux + n + e]);
        else if (EPI == EPI_ADD) v += bf2f(p.aux_in[m * p.ld_aux + n + e]);
        else if (EPI == EPI_ROWFIX) v = p.bias[m] * (v - bf2f(p.aux_in[m * p.ld_aux + n + e]) * p.bias[(size_t)p.M + m]);
        else if (EPI == EPI_TANH) v = tanhf(v);
        if (CMODE == 0) reinterpret_cast<bf16*>(p.C)[m * p.ldc + n + e] = f2bf(v);
        else if (CMODE == 1) reinterpret_cast<float*>(p.C)[m * p.ldc + n + e] = v;
        else reinterpret_cast<float*>(p.C)[m * p.ldc + n + e] += v;
      }
    }
    return;
  }
  auto row_body = [&](int it) __attribute__((always_inline)) {
    const int r = it * 4 + (lane >> 4);
    if (CHECK && row0 + r >= p.M) return;
    const size_t m = (size_t)(row0 + r);
    const f4 a = *reinterpret_cast<const f4*>(tile + r * EP_PITCH + c4);
    float v[4] = {a[0] * p.alpha + bias.x, a[1] * p.alpha + bias.y, a[2] * p.alpha + bias.z, a[3] * p.alpha + bias.w};
    if (EPI == EPI_GELU_DGELU) {
      bf4 dg;
#pragma unroll
      for (int e = 0; e < 4; ++e) { float g, d; gelu_both_f(bf2f(f2bf(v[e])), g, d); v[e] = g; dg[e] = f2bf(d); }      // of the rounded pre-activation
      __builtin_nontemporal_store(dg, reinterpret_cast<bf4*>(p.aux_out + m * p.ld_aux + n));      // only read again in backward
    } else if (EPI == EPI_MUL) {
      const bf4 x = PRE ? auxv[it] : __builtin_nontemporal_load(reinterpret_cast<const bf4*>(p.aux_in + m * p.ld_aux + n));
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= bf2f(x[e]);
    } else if (EPI == EPI_GELU) {
      bf4 pre = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
      __builtin_nontemporal_store(pre, reinterpret_cast<bf4*>(p.aux_out + m * p.ld_aux + n));      // only read again in backward
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = gelu_f(bf2f(pre[e]));
    } else if (EPI == EPI_MUL_GELU_GRAD) {
      const bf4 x = PRE ? auxv[it] : __builtin_nontemporal_load(reinterpret_cast<const bf4*>(p.aux_in + m * p.ld_aux + n));      // read once
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= gelu_grad_f(bf2f(x[e]));
    } else if (EPI == EPI_ADD) {
      const bf4 x = PRE ? auxv[it] : *reinterpret_cast<const bf4*>(p.aux_in + m * p.ld_aux + n);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += bf2f(x[e]);
    } else if (EPI == EPI_TANH) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = tanhf(v[e]);
    } else if (EPI == EPI_ROWFIX) {
      const bf4 x = *reinterpret_cast<const bf4*>(p.aux_in + m * p.ld_aux + n);
      const float rs = p.bias[m], rr = p.bias[(size_t)p.M + m];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = rs * (v[e] - bf2f(x[e]) * rr);
    } else if (EPI == EPI_ARCSTATS) {
      // v = four cosines of row m; the row's 64 columns of this wave sit in the 16 lanes of one DPP row (full tiles: EXEC is full)
      const long long y = p.arc_label[m];
      float z[4], mx = -3.0e38f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int col = n + e;
        z[e] = col < p.arc_C ? ((long long)col == y ? margin_fwd(v[e], p.arc_m, nullptr) : v[e]) * p.arc_m.s : -3.0e38f;
        mx = fmaxf(mx, z[e]);
      }
      mx = max16_dpp(mx);
      float sm = 0.f;
      int am = 0x7fffffff;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sm += z[e] > -1.0e38f ? __expf(z[e] - mx) : 0.f;
        if (z[e] == mx && am == 0x7fffffff) am = n + e;            // lowest index among equal maxima
      }
      sm = sum16_dpp(sm);
      am = min16_dpp_i(am);
      if ((lane & 15) == 0) {
        const f4 o = {mx, sm, __int_as_float(am), 0.f};
        *reinterpret_cast<f4*>(p.arc_part + ((size_t)m * (size_t)(p.N >> 6) + (size_t)(col0 >> 6)) * 4) = o;
      }
    }
    if (CMODE == 0) {
      bf4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
      __builtin_nontemporal_store(o, reinterpret_cast<bf4*>(reinterpret_cast<bf16*>(p.C) + m * p.ldc + n));
    } else {
      float* c = reinterpret_cast<float*>(p.C) + m * p.ldc + n;
      if (CMODE == 1) {
        *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        const float4 old = *reinterpret_cast<const float4*>(c);
        *reinterpret_cast<float4*>(c) = make_float4(v[0] + old.x, v[1] + old.y, v[2] + old.z, v[3] + old.w);
      }
    }
  };
  if (PRE) {          // the prefetched operands are register-array elements: their index must be a compile-time constant
#pragma unroll
    for (int it = 0; it < 16; ++it) row_body(it);
  } else {
#pragma unroll 4
    for (int it = 0; it < 16; ++it) row_body(it);
  }
}


// bf16 store of a 64x64 sub-tile (no bias / activation: the 1x1 convs) that also returns the column sums of the ROUNDED
// outputs -- the train-mode BatchNorm statistics of the conv output, so no separate pass re-reads it.
// cs / cq: lanes 0..15 end up with sum / sum of squares of columns col0 + 4*(lane&15) + {0..3} over this wave's valid rows.
template <bool F16 = false>
__device__ __forceinline__ void stats_epilogue(const GemmParams& p, f4 (&acc)[4][4], int row0, int col0, int lane, float* tile,
                                               float (&cs)[4], float (&cq)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<f4*>(tile + (i * 16 + (lane & 15)) * EP_PITCH + j * 16 + (lane >> 4) * 4) = acc[i][j];
  const int c4 = (lane & 15) * 4, n = col0 + c4;
#pragma unroll
  for (int e = 0; e < 4; ++e) cs[e] = cq[e] = 0.f;
  if (n < p.N) {                       // N % 8 == 0 (checked on the host): a 4-column group is valid as a whole
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
      const int r = it * 4 + (lane >> 4);
      if (row0 + r >= p.M) continue;
      const f4 a = *reinterpret_cast<const f4*>(tile + r * EP_PITCH + c4);
      if (F16) {
        h4 o = {f2h(a[0] * p.alpha), f2h(a[1] * p.alpha), f2h(a[2] * p.alpha), f2h(a[3] * p.alpha)};
        *reinterpret_cast<h4*>(reinterpret_cast<f16*>(p.C) + (size_t)(row0 + r) * p.ldc + n) = o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float v = h2f(o[e]); cs[e] += v; cq[e] += v * v; }
        continue;
      }
      bf4 o = {f2bf(a[0] * p.alpha), f2bf(a[1] * p.alpha), f2bf(a[2] * p.alpha), f2bf(a[3] * p.alpha)};
      *reinterpret_cast<bf4*>(reinterpret_cast<bf16*>(p.C) + (size_t)(row0 + r) * p.ldc + n) = o;
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float v = bf2f(o[e]); cs[e] += v; cq[e] += v * v; }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    cs[e] += __shfl_xor(cs[e], 16, 64); cs[e] += __shfl_xor(cs[e], 32, 64);
    cq[e] += __shfl_xor(cq[e], 16, 64); cq[e] += __shfl_xor(cq[e], 32, 64);
  }
}

// fp16 store of a 64x64 sub-tile, ragged tiles predicated, no bias / activation (fmt 1 without the statistics: the 1x1 convs of
// the image tower in tests and eval-side callers); N % 4 == 0 is the host's contract (ldc % 4), columns are checked per group
__device__ __forceinline__ void f16_epilogue(const GemmParams& p, f4 (&acc)[4][4], int row0, int col0, int lane, float* tile) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<f4*>(tile + (i * 16 + (lane & 15)) * EP_PITCH + j * 16 + (lane >> 4) * 4) = acc[i][j];
  const int c4 = (lane & 15) * 4, n = col0 + c4;
  if (n >= p.N) return;
#pragma unroll 4
  for (int it = 0; it < 16; ++it) {
    const int r = it * 4 + (lane >> 4);
    if (row0 + r >= p.M) continue;
    const f4 a = *reinterpret_cast<const f4*>(tile + r * EP_PITCH + c4);
    f16* c = reinterpret_cast<f16*>(p.C) + (size_t)(row0 + r) * p.ldc + n;
    if (n + 3 < p.N) {
      h4 o = {f2h(a[0] * p.alpha), f2h(a[1] * p.alpha), f2h(a[2] * p.alpha), f2h(a[3] * p.alpha)};
      *reinterpret_cast<h4*>(c) = o;
    } else {
      for (int e = 0; e < 4 && n + e < p.N; ++e) c[e] = f2h(a[e] * p.alpha);
    }
  }
}

// bf16-output epilogue for FULL tiles with 16-byte accesses (cdna_hip_programming.md T21): a lane owns 8 consecutive columns, so
// a 128-byte row segment of the wave's 64 x 64 sub-tile is 8 lanes x dwordx4 instead of 16 lanes x dwordx2 -- half the store
// (and aux load) instructions.  The epilogue of the 256 x 256 tiles is store-ISSUE-bound (~7 B/cycle/CU with every wave storing
// dwordx2: ~15 us per tile for the two-output GELU epilogue against ~23 us of main loop at K = 1024).
template <int EPI>
__device__ __forceinline__ void fast_epilogue_bf16_wide(const GemmParams& p, f4 (&acc)[4][4], int row0, int col0, int lane, bool first_split,
                                                        float* tile) {
  const int c8 = (lane & 7) * 8, n = col0 + c8, rsub = lane >> 3;
  constexpr bool RD = (EPI == EPI_MUL_GELU_GRAD || EPI == EPI_ADD || EPI == EPI_MUL);
  bf8 auxv[8];
  if (RD) {          // all eight row segments of the lane requested ahead of the staging round trip
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const size_t m = (size_t)(row0 + it * 8 + rsub);
      if (EPI == EPI_ADD) auxv[it] = *reinterpret_cast<const bf8*>(p.aux_in + m * p.ld_aux + n);
      else auxv[it] = __builtin_nontemporal_load(reinterpret_cast<const bf8*>(p.aux_in + m * p.ld_aux + n));
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<f4*>(tile + (i * 16 + (lane & 15)) * EP_PITCH + j * 16 + (lane >> 4) * 4) = acc[i][j];
  float bias[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (p.bias != nullptr && first_split) {
    const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n), b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
    bias[0] = b0.x; bias[1] = b0.y; bias[2] = b0.z; bias[3] = b0.w; bias[4] = b1.x; bias[5] = b1.y; bias[6] = b1.z; bias[7] = b1.w;
  }
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int r = it * 8 + rsub;
    const size_t m = (size_t)(row0 + r);
    const f4 a0 = *reinterpret_cast<const f4*>(tile + r * EP_PITCH + c8), a1 = *reinterpret_cast<const f4*>(tile + r * EP_PITCH + c8 + 4);
    float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = v[e] * p.alpha + bias[e];
    if (EPI == EPI_GELU_DGELU) {
      bf8 dg;
#if !GELU_PK      // A/B builds only (tools/build_variant.sh -DGELU_PK=0): the element-at-a-time form
#pragma unroll
      for (int e = 0; e < 8; ++e) { float g, d; gelu_both_f(bf2f(f2bf(v[e])), g, d); v[e] = g; dg[e] = f2bf(d); }
#else
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        f2v g, d;
        gelu_both_pk((f2v){v[e], v[e + 1]}, g, d);      // of the fp32 pre-activation (gelu and gelu' from the same value; no bf16 round trip)
        v[e] = g[0]; v[e + 1] = g[1]; dg[e] = f2bf(d[0]); dg[e + 1] = f2bf(d[1]);
      }
#endif
      __builtin_nontemporal_store(dg, reinterpret_cast<bf8*>(p.aux_out + m * p.ld_aux + n));
    } else if (EPI == EPI_GELU) {
      bf8 pre;
#pragma unroll
      for (int e = 0; e < 8; ++e) { pre[e] = f2bf(v[e]); v[e] = gelu_f(bf2f(pre[e])); }
      __builtin_nontemporal_store(pre, reinterpret_cast<bf8*>(p.aux_out + m * p.ld_aux + n));
    } else if (EPI == EPI_MUL) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= bf2f(auxv[it][e]);
    } else if (EPI == EPI_MUL_GELU_GRAD) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= gelu_grad_f(bf2f(auxv[it][e]));
    } else if (EPI == EPI_ADD) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += bf2f(auxv[it][e]);
    } else if (EPI == EPI_TANH) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = tanhf(v[e]);
    }
    bf8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(v[e]);
    __builtin_nontemporal_store(o, reinterpret_cast<bf8*>(reinterpret_cast<bf16*>(p.C) + m * p.ldc + n));
  }
}

template <int CMODE, bool CHECK = false>
__device__ __forceinline__ void fast_epilogue_epi(const GemmParams& p, f4 (&acc)[4][4], int row0, int col0, int lane, bool first_split,
                                                  float* tile) {
  if (CMODE == 0 && !CHECK && (p.ldc % 8) == 0 && (p.ld_aux % 8) == 0) {      // launch-uniform: bf16 output, full tiles, 16-byte rows
    switch (p.epi) {
      case EPI_GELU: fast_epilogue_bf16_wide<EPI_GELU>(p, acc, row0, col0, lane, first_split, tile); break;
      case EPI_MUL_GELU_GRAD: fast_epilogue_bf16_wide<EPI_MUL_GELU_GRAD>(p, acc, row0, col0, lane, first_split, tile); break;
      case EPI_ADD: fast_epilogue_bf16_wide<EPI_ADD>(p, acc, row0, col0, lane, first_split, tile); break;
      case EPI_TANH: fast_epilogue_bf16_wide<EPI_TANH>(p, acc, row0, col0, lane, first_split, tile); break;
      case EPI_GELU_DGELU: fast_epilogue_bf16_wide<EPI_GELU_DGELU>(p, acc, row0, col0, lane, first_split, tile); break;
      case EPI_MUL: fast_epilogue_bf16_wide<EPI_MUL>(p, acc, row0, col0, lane, first_split, tile); break;
      default: fast_epilogue_bf16_wide<EPI_NONE>(p, acc, row0, col0, lane, first_split, tile); break;
    }
    return;
  }
  switch (p.epi) {
    case EPI_GELU: fast_epilogue<EPI_GELU, CMODE, CHECK>(p, acc, row0, col0, lane, first_split, tile); break;
    case EPI_MUL_GELU_GRAD: fast_epilogue<EPI_MUL_GELU_GRAD, CMODE, CHECK>(p, acc, row0, col0, lane, first_split, tile); break;
    case EPI_ADD: fast_epilogue<EPI_ADD, CMODE, CHECK>(p, acc, row0, col0, lane, first_split, tile); break;
    case EPI_TANH: fast_epilogue<EPI_TANH, CMODE, CHECK>(p, acc, row0, col0, lane, first_split, tile); break;
    case EPI_GELU_DGELU: fast_epilogue<EPI_GELU_DGELU, CMODE, CHECK>(p, acc, row0, col0, lane, first_split, tile); break;
    case EPI_MUL: fast_epilogue<EPI_MUL, CMODE, CHECK>(p, acc, row0, col0, lane, first_split, tile); break;
    default: fast_epilogue<EPI_NONE, CMODE, CHECK>(p, acc, row0, col0, lane, first_split, tile); break;
  }
}


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
      } else if (p.epi == EPI_GELU_DGELU) {
        bf16* ao = p.aux_out + (size_t)m * p.ld_aux + n;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < p.N) { float g, dg; gelu_both_f(bf2f(f2bf(v[e])), g, dg); ao[e] = f2bf(dg); v[e] = g; }
      } else if (p.epi == EPI_MUL) {
        const bf16* ai = p.aux_in + (size_t)m * p.ld_aux + n;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < p.N) v[e] *= bf2f(ai[e]);
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
