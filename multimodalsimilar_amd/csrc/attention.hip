// Fused BERT self-attention for gfx950, forward and backward, for head_dim 64 and S in {32, 64, 128}.
//   forward : ctx = dropout(softmax(q k^T / sqrt(d) + mask)) v          (modeling_bert.py:111-136)
//   backward: recomputes P from q, k and the saved log-sum-exp; never stores the SxS matrix.
// One workgroup per (batch, head); S/32 waves.  All products run on v_mfma_f32_32x32x16_bf16 (16x16x32 for
// dQ).  Layout tricks (MI355X guide, "accumulator tile as the next MFMA's operand"):
//   * forward computes S^T = K Q^T so that a query's whole key row lives in one lane pair: softmax is
//     register-local plus one cross-half shuffle, and P^T is already the B operand of O^T = V^T P^T;
//   * backward keeps the key on the lane: S and dP accumulators are directly the B operands of
//     dV^T += dO^T P and dK^T += Q^T dS; only dS crosses LDS once (as dS^T) for dQ = dS K;
//   * operands whose reduction index is the LDS row index are read with ds_read_b64_tr_b16.
// q, k, v are column slices of the fused QKV activation [B*S, 3H]: q at h*64, k at H + h*64, v at 2H + h*64.
#include "common.h"

#define ROW_PITCH 144   // bytes, row reads (ds_read_b128) conflict-free
#define TRV_PITCH 192   // bytes, forward V image: transposed reads conflict-free
#define DST_PITCH 80    // bytes, dS^T image rows of 32 bf16

struct AttnParams {
  const bf16* qkv; const int64_t* mask; bf16* ctx; const bf16* dctx; float* lse; bf16* dqkv;
  float* dbias_parts;    // backward: per-batch-element column sums of dqkv, slab [B][3H] (NULL = off)
  int ld_qkv, ld_ctx, heads, H;
  float scale;
  unsigned long long seed; unsigned int stream, thresh; float inv_keep;
};

typedef s4 __attribute__((address_space(3))) * lds_s4_ptr;

__device__ __forceinline__ bf8 tr_pair(const char* base, int off_lo, int row_step_bytes8) {
  s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(base + off_lo));
  s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(base + off_lo + row_step_bytes8));
  s8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return __builtin_bit_cast(bf8, r);
}

// A-operand fragment of X^T for a 32x32x16 MFMA whose B operand is an accumulator tile:
// image rows = reduction index (row0 + 16*s2 + ...), image cols = output rows (col0 + lane&31).
__device__ __forceinline__ bf8 tr_frag32(const char* img, int pitch, int row0, int s2, int col0, int lane) {
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3, hh = g >> 1;
  const int row = row0 + 16 * s2 + 4 * hh + q;
  const int col = col0 + 16 * (g & 1) + 4 * p;
  return tr_pair(img, row * pitch + col * 2, 8 * pitch);
}
// operand fragment for a 16x16x32 MFMA: image rows = reduction index (row0 + 8*(lane>>4) + ...), cols col0 + lane&15
__device__ __forceinline__ bf8 tr_frag16(const char* img, int pitch, int row0, int col0, int lane) {
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int row = row0 + 8 * g + q;
  const int col = col0 + 4 * p;
  return tr_pair(img, row * pitch + col * 2, 4 * pitch);
}

__device__ __forceinline__ bf8 cvt8(const f16v& a, int s2, float mul) {
  bf8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = f2bf(a[8 * s2 + j] * mul);
  return r;
}

template <int NT>
__global__ __launch_bounds__(64 * NT) void attn_fwd_kernel(AttnParams p) {
  constexpr int S = 32 * NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;                       // [S][ROW_PITCH]
  char* Vs = smem + S * ROW_PITCH;       // [S][TRV_PITCH]
  float* mb = reinterpret_cast<float*>(smem + S * ROW_PITCH + S * TRV_PITCH);   // [S]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, hh = lane >> 5;
  const int bh = blockIdx.x, b = bh / p.heads, h = bh % p.heads;
  const bf16* base = p.qkv + (size_t)b * S * p.ld_qkv + h * 64;

  // stage K and V rows (8 x 16-byte chunks per row)
  {   // S*8 chunks over 64*NT threads = 4 trips: all 8 loads requested before the first LDS store
    uint4 kv[4], vv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + i * 64 * NT, row = c >> 3, ch = c & 7;
      kv[i] = *reinterpret_cast<const uint4*>(base + (size_t)row * p.ld_qkv + p.H + ch * 8);
      vv[i] = *reinterpret_cast<const uint4*>(base + (size_t)row * p.ld_qkv + 2 * p.H + ch * 8);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + i * 64 * NT, row = c >> 3, ch = c & 7;
      *reinterpret_cast<uint4*>(Ks + row * ROW_PITCH + ch * 16) = kv[i];
      *reinterpret_cast<uint4*>(Vs + row * TRV_PITCH + ch * 16) = vv[i];
    }
  }
  for (int i = tid; i < S; i += 64 * NT) mb[i] = (p.mask && p.mask[(size_t)b * S + i] == 0) ? -1e30f : 0.f;

  // Q^T fragments (B operand): lane = query column
  const int qrow = 32 * w + (lane & 31);
  bf8 qf[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk)
    qf[kk] = *reinterpret_cast<const bf8*>(base + (size_t)qrow * p.ld_qkv + 16 * kk + 8 * hh);
  __syncthreads();

  f16v sacc[NT];
#pragma unroll
  for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
    for (int i = 0; i < 16; ++i) sacc[kt][i] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const bf8 kf = *reinterpret_cast<const bf8*>(Ks + (32 * kt + (lane & 31)) * ROW_PITCH + (16 * kk + 8 * hh) * 2);
      sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], sacc[kt], 0, 0, 0);
    }
  }
  // softmax over keys: this lane holds keys {32kt + (r&3) + 8(r>>2) + 4hh}, partner lane^32 the rest
  float mx = -3.0e38f;
#pragma unroll
  for (int kt = 0; kt < NT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * hh;
      const float s = sacc[kt][r] * p.scale + mb[key];
      sacc[kt][r] = s;
      mx = fmaxf(mx, s);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < NT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = __expf(sacc[kt][r] - mx);
      sacc[kt][r] = e;
      sum += e;
    }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
  if (hh == 0 && p.lse) p.lse[(size_t)bh * S + qrow] = mx + __logf(sum);
  if (p.thresh) {                       // accumulator registers r, r+1 (r even) are keys k, k+1: one hash per pair
    const uint32_t dkey = drop_key(p.seed, p.stream);
    const unsigned long long rowbase = ((unsigned long long)bh * S + qrow) * S;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const int key = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * hh;
        const uint32_t bits = drop_bits(dkey, (rowbase + key) >> 1);
        sacc[kt][r] = drop_keep16(bits, 0, p.thresh) ? sacc[kt][r] * p.inv_keep : 0.f;
        sacc[kt][r + 1] = drop_keep16(bits, 1, p.thresh) ? sacc[kt][r + 1] * p.inv_keep : 0.f;
      }
  }
  // O^T[d][q] = sum_key V^T[d][key] P^T[key][q]
  f16v oacc[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
    for (int i = 0; i < 16; ++i) oacc[dt][i] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf8 pb = cvt8(sacc[kt], s2, inv);
        const bf8 vf = tr_frag32(Vs, TRV_PITCH, 32 * kt, s2, 32 * dt, lane);
        oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb, oacc[dt], 0, 0, 0);
      }
  }
  bf16* out = p.ctx + ((size_t)b * S + qrow) * p.ld_ctx + h * 64;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      bf4 o = {f2bf(oacc[dt][4 * g4]), f2bf(oacc[dt][4 * g4 + 1]), f2bf(oacc[dt][4 * g4 + 2]), f2bf(oacc[dt][4 * g4 + 3])};
      *reinterpret_cast<bf4*>(out + 32 * dt + 8 * g4 + 4 * hh) = o;
    }
}

template <int NT>
__global__ __launch_bounds__(64 * NT) void attn_bwd_kernel(AttnParams p) {
  constexpr int S = 32 * NT;
  constexpr int TPW = 8 / NT;   // dQ 16x16 tiles per wave per query tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Qs = smem;                          // [S][ROW_PITCH]
  char* Os = smem + S * ROW_PITCH;          // dO
  char* Ks = smem + 2 * S * ROW_PITCH;      // K
  char* Ds = smem + 3 * S * ROW_PITCH;      // dS^T [S keys][DST_PITCH] for the current query tile
  float* fl = reinterpret_cast<float*>(smem + 3 * S * ROW_PITCH + S * DST_PITCH);
  float* lse = fl; float* dl = fl + S; float* mb = fl + 2 * S;
  float* cb = fl + 3 * S;                   // [192] column sums of this (b, h)'s dq | dk | dv (bias gradient of the QKV projection)
  float cq[TPW][4];
#pragma unroll
  for (int t = 0; t < TPW; ++t) cq[t][0] = cq[t][1] = cq[t][2] = cq[t][3] = 0.f;
  if (p.dbias_parts)
    for (int i = threadIdx.x; i < 192; i += 64 * NT) cb[i] = 0.f;       // ordered before the adds by the barriers of the loop
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, hh = lane >> 5;
  const int bh = blockIdx.x, b = bh / p.heads, h = bh % p.heads;
  const bf16* base = p.qkv + (size_t)b * S * p.ld_qkv + h * 64;
  const bf16* dob = p.dctx + (size_t)b * S * p.ld_ctx + h * 64;
  const bf16* ob = p.ctx + (size_t)b * S * p.ld_ctx + h * 64;

  {   // 4 trips x 3 operands: all 12 loads requested before the first LDS store
    uint4 qv[4], kv[4], ov[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + i * 64 * NT, row = c >> 3, ch = c & 7;
      qv[i] = *reinterpret_cast<const uint4*>(base + (size_t)row * p.ld_qkv + ch * 8);
      kv[i] = *reinterpret_cast<const uint4*>(base + (size_t)row * p.ld_qkv + p.H + ch * 8);
      ov[i] = *reinterpret_cast<const uint4*>(dob + (size_t)row * p.ld_ctx + ch * 8);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + i * 64 * NT, row = c >> 3, ch = c & 7;
      *reinterpret_cast<uint4*>(Qs + row * ROW_PITCH + ch * 16) = qv[i];
      *reinterpret_cast<uint4*>(Ks + row * ROW_PITCH + ch * 16) = kv[i];
      *reinterpret_cast<uint4*>(Os + row * ROW_PITCH + ch * 16) = ov[i];
    }
  }
  for (int i = tid; i < S; i += 64 * NT) {
    mb[i] = (p.mask && p.mask[(size_t)b * S + i] == 0) ? -1e30f : 0.f;
    lse[i] = p.lse[(size_t)bh * S + i];
  }
  {  // delta[q] = sum_d dO[q][d] * O[q][d]; lane&31 = row of this wave's 32 rows, hh = half of d
    const int row = 32 * w + (lane & 31);
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const bf8 a = *reinterpret_cast<const bf8*>(dob + (size_t)row * p.ld_ctx + 32 * hh + 8 * c);
      const bf8 o = *reinterpret_cast<const bf8*>(ob + (size_t)row * p.ld_ctx + 32 * hh + 8 * c);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += bf2f(a[j]) * bf2f(o[j]);
    }
    acc += __shfl_xor(acc, 32, 64);
    if (hh == 0) dl[row] = acc;
  }
  // loop-invariant B operands: K^T and V^T columns for this wave's 32 keys (lane = key)
  const int key = 32 * w + (lane & 31);
  bf8 kreg[4], vreg[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    kreg[kk] = *reinterpret_cast<const bf8*>(base + (size_t)key * p.ld_qkv + p.H + 16 * kk + 8 * hh);
    vreg[kk] = *reinterpret_cast<const bf8*>(base + (size_t)key * p.ld_qkv + 2 * p.H + 16 * kk + 8 * hh);
  }
  __syncthreads();
  const float mbk = mb[key];
  const uint32_t dkey = drop_key(p.seed, p.stream);

  f16v dV[2], dK[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) { dV[dt][i] = 0.f; dK[dt][i] = 0.f; }

  for (int qt = 0; qt < NT; ++qt) {
    f16v X, dP;
#pragma unroll
    for (int i = 0; i < 16; ++i) { X[i] = 0.f; dP[i] = 0.f; }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int off = (32 * qt + (lane & 31)) * ROW_PITCH + (16 * kk + 8 * hh) * 2;
      const bf8 qa = *reinterpret_cast<const bf8*>(Qs + off);
      const bf8 oa = *reinterpret_cast<const bf8*>(Os + off);
      X = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kreg[kk], X, 0, 0, 0);
      dP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(oa, vreg[kk], dP, 0, 0, 0);
    }
    // X[r] -> P (dropped, for dV) ; dP[r] -> dS
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int q = 32 * qt + (r & 3) + 8 * (r >> 2) + 4 * hh;
      const float pr = __expf(X[r] * p.scale + mbk - lse[q]);
      float ks = 1.0f;
      if (p.thresh) {
        const unsigned long long idx = ((unsigned long long)bh * S + q) * S + key;
        ks = drop_keep16(drop_bits(dkey, idx >> 1), key & 1, p.thresh) ? p.inv_keep : 0.f;
      }
      X[r] = pr * ks;
      dP[r] = pr * (dP[r] * ks - dl[q]);
    }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf8 pb = cvt8(X, s2, 1.0f);
        const bf8 sb = cvt8(dP, s2, 1.0f);
        const bf8 of = tr_frag32(Os, ROW_PITCH, 32 * qt, s2, 32 * dt, lane);
        const bf8 qf = tr_frag32(Qs, ROW_PITCH, 32 * qt, s2, 32 * dt, lane);
        dV[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(of, pb, dV[dt], 0, 0, 0);
        dK[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf, sb, dK[dt], 0, 0, 0);
      }
    // dS^T[key][q_local] (bf16) -> LDS
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      bf4 o = {f2bf(dP[4 * g4]), f2bf(dP[4 * g4 + 1]), f2bf(dP[4 * g4 + 2]), f2bf(dP[4 * g4 + 3])};
      *reinterpret_cast<bf4*>(Ds + key * DST_PITCH + (8 * g4 + 4 * hh) * 2) = o;
    }
    __syncthreads();
    // dQ[q][d] = scale * sum_key dS[q][key] K[key][d] : 8 tiles of 16x16 per query tile, TPW per wave
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int tile = w * TPW + t, qs = tile >> 2, d4 = tile & 3;
      f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NT; ++ks) {
        const bf8 af = tr_frag16(Ds, DST_PITCH, 32 * ks, 16 * qs, lane);   // dS[q][key]: "rows" q
        const bf8 bf = tr_frag16(Ks, ROW_PITCH, 32 * ks, 16 * d4, lane);   // K[key][d]: "cols" d
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf, af, acc, 0, 0, 0);
      }
      const int q = 32 * qt + 16 * qs + (lane & 15);
      const int d = 16 * d4 + (lane >> 4) * 4;
      bf4 o = {f2bf(acc[0] * p.scale), f2bf(acc[1] * p.scale), f2bf(acc[2] * p.scale), f2bf(acc[3] * p.scale)};
      *reinterpret_cast<bf4*>(p.dqkv + ((size_t)b * S + q) * p.ld_qkv + h * 64 + d) = o;
#pragma unroll
      for (int e = 0; e < 4; ++e) cq[t][e] += bf2f(o[e]);
    }
    __syncthreads();
  }
  bf16* dk = p.dqkv + ((size_t)b * S + key) * p.ld_qkv + p.H + h * 64;
  bf16* dv = dk + p.H;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int d = 32 * dt + 8 * g4 + 4 * hh;
      bf4 ok = {f2bf(dK[dt][4 * g4] * p.scale), f2bf(dK[dt][4 * g4 + 1] * p.scale), f2bf(dK[dt][4 * g4 + 2] * p.scale),
                f2bf(dK[dt][4 * g4 + 3] * p.scale)};
      bf4 ov = {f2bf(dV[dt][4 * g4]), f2bf(dV[dt][4 * g4 + 1]), f2bf(dV[dt][4 * g4 + 2]), f2bf(dV[dt][4 * g4 + 3])};
      *reinterpret_cast<bf4*>(dk + d) = ok;
      *reinterpret_cast<bf4*>(dv + d) = ov;
      if (p.dbias_parts) {          // sums over this wave's 32 keys (the lanes of a half-wave), of the ROUNDED values
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float sk = bf2f(ok[e]), sv = bf2f(ov[e]);
#pragma unroll
          for (int o2 = 1; o2 < 32; o2 <<= 1) { sk += __shfl_xor(sk, o2, 64); sv += __shfl_xor(sv, o2, 64); }
          if ((lane & 31) == 0) { atomicAdd(cb + 64 + d + e, sk); atomicAdd(cb + 128 + d + e, sv); }
        }
      }
    }
  if (p.dbias_parts) {
#pragma unroll
    for (int t = 0; t < TPW; ++t) {       // dq: rows of a 16x16 tile sit on lanes 0..15 of each 16-lane group
      const int d4 = (w * TPW + t) & 3;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float sq = cq[t][e];
#pragma unroll
        for (int o2 = 1; o2 < 16; o2 <<= 1) sq += __shfl_xor(sq, o2, 64);
        if ((lane & 15) == 0) atomicAdd(cb + 16 * d4 + (lane >> 4) * 4 + e, sq);
      }
    }
    __syncthreads();
    for (int i = tid; i < 192; i += 64 * NT)
      p.dbias_parts[(size_t)b * 3 * p.H + (i >> 6) * p.H + h * 64 + (i & 63)] = cb[i];
  }
}

static int attn_check(int B, int S, int heads, int H, int ld_qkv, int ld_ctx, const void* qkv) {
  MMSIM_REQUIRE(B > 0 && heads > 0, "attention: B, heads must be positive");
  MMSIM_REQUIRE(S == 32 || S == 64 || S == 128, "attention: sequence length must be 32, 64 or 128");
  MMSIM_REQUIRE(H == heads * 64, "attention: head_dim must be 64 (H == heads*64)");
  MMSIM_REQUIRE(ld_qkv >= 3 * H && (ld_qkv % 8) == 0 && ld_ctx >= H && (ld_ctx % 8) == 0, "attention: bad leading dims");
  MMSIM_REQUIRE(((uintptr_t)qkv % 16) == 0, "attention: qkv must be 16-byte aligned");
  return MMSIM_OK;
}

extern "C" int mmsim_attn_fwd(const void* qkv, int ld_qkv, const long long* mask, void* ctx, int ld_ctx, float* lse,
                              int B, int S, int heads, int H, float dropout_p, unsigned long long seed,
                              unsigned int stream_id, void* stream) {
  int rc = attn_check(B, S, heads, H, ld_qkv, ld_ctx, qkv);
  if (rc) return rc;
  MMSIM_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "attention: dropout_p in [0,1)");
  AttnParams p;
  p.qkv = (const bf16*)qkv; p.mask = (const int64_t*)mask; p.ctx = (bf16*)ctx; p.dctx = nullptr; p.lse = lse; p.dqkv = nullptr;
  p.dbias_parts = nullptr;
  p.ld_qkv = ld_qkv; p.ld_ctx = ld_ctx; p.heads = heads; p.H = H; p.scale = 0.125f;
  p.seed = seed; p.stream = stream_id;
  p.thresh = dropout_p > 0.f ? (unsigned int)((double)dropout_p * 4294967296.0) : 0u;
  p.inv_keep = 1.0f / (1.0f - dropout_p);
  const int NT = S / 32;
  const size_t lds = (size_t)S * (ROW_PITCH + TRV_PITCH) + S * 4;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(B * heads), block(64 * NT);
  if (NT == 1) hipLaunchKernelGGL((attn_fwd_kernel<1>), grid, block, lds, s, p);
  else if (NT == 2) hipLaunchKernelGGL((attn_fwd_kernel<2>), grid, block, lds, s, p);
  else hipLaunchKernelGGL((attn_fwd_kernel<4>), grid, block, lds, s, p);
  return mmsim_check_launch("attn_fwd");
}

static int attn_bwd_impl(const void* qkv, int ld_qkv, const long long* mask, const void* ctx, const void* dctx, int ld_ctx,
                         const float* lse, void* dqkv, int B, int S, int heads, int H, float dropout_p, unsigned long long seed,
                         unsigned int stream_id, float* dbias_parts, void* stream) {
  int rc = attn_check(B, S, heads, H, ld_qkv, ld_ctx, qkv);
  if (rc) return rc;
  MMSIM_REQUIRE(ctx && dctx && lse && dqkv, "attention bwd: null operand");
  AttnParams p;
  p.qkv = (const bf16*)qkv; p.mask = (const int64_t*)mask; p.ctx = (bf16*)ctx; p.dctx = (const bf16*)dctx;
  p.lse = (float*)lse; p.dqkv = (bf16*)dqkv; p.dbias_parts = dbias_parts;
  p.ld_qkv = ld_qkv; p.ld_ctx = ld_ctx; p.heads = heads; p.H = H; p.scale = 0.125f;
  p.seed = seed; p.stream = stream_id;
  p.thresh = dropout_p > 0.f ? (unsigned int)((double)dropout_p * 4294967296.0) : 0u;
  p.inv_keep = 1.0f / (1.0f - dropout_p);
  const int NT = S / 32;
  const size_t lds = (size_t)S * (3 * ROW_PITCH + DST_PITCH) + 3 * S * 4 + 192 * 4;
  hipStream_t s = (hipStream_t)stream;
  static bool attr_done = false;
  if (!attr_done) {   // S = 128 needs 66 KiB of dynamic LDS
    (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * (3 * ROW_PITCH + DST_PITCH) + 3 * 128 * 4 + 192 * 4);
    attr_done = true;
  }
  dim3 grid(B * heads), block(64 * NT);
  if (NT == 1) hipLaunchKernelGGL((attn_bwd_kernel<1>), grid, block, lds, s, p);
  else if (NT == 2) hipLaunchKernelGGL((attn_bwd_kernel<2>), grid, block, lds, s, p);
  else hipLaunchKernelGGL((attn_bwd_kernel<4>), grid, block, lds, s, p);
  return mmsim_check_launch("attn_bwd");
}

extern "C" int mmsim_attn_bwd(const void* qkv, int ld_qkv, const long long* mask, const void* ctx, const void* dctx,
                              int ld_ctx, const float* lse, void* dqkv, int B, int S, int heads, int H,
                              float dropout_p, unsigned long long seed, unsigned int stream_id, void* stream) {
  return attn_bwd_impl(qkv, ld_qkv, mask, ctx, dctx, ld_ctx, lse, dqkv, B, S, heads, H, dropout_p, seed, stream_id, nullptr, stream);
}

void mmsim_launch_reduce(const float* parts, int nparts, int n, float* out, int accumulate, hipStream_t s);   // conv.hip

// attn_bwd that also accumulates dbias [3H] += column sums of dqkv (the q | k | v bias gradients) out of the kernel:
// every (batch, head) workgroup leaves the sums of its 3 x 64 columns in a [B][3H] slab (scratch), reduced by one small launch.
extern "C" int mmsim_attn_bwd_dbias(const void* qkv, int ld_qkv, const long long* mask, const void* ctx, const void* dctx,
                                    int ld_ctx, const float* lse, void* dqkv, float* dbias, int B, int S, int heads, int H,
                                    float dropout_p, unsigned long long seed, unsigned int stream_id, float* scratch,
                                    unsigned long long scratch_floats, void* stream) {
  MMSIM_REQUIRE(dbias && scratch && scratch_floats >= (unsigned long long)B * 3 * H, "attn_bwd_dbias: dbias and a scratch of B*3H floats required");
  const int rc = attn_bwd_impl(qkv, ld_qkv, mask, ctx, dctx, ld_ctx, lse, dqkv, B, S, heads, H, dropout_p, seed, stream_id, scratch, stream);
  if (rc) return rc;
  mmsim_launch_reduce(scratch, B, 3 * H, dbias, 1, (hipStream_t)stream);
  return mmsim_check_launch("attn_bwd_dbias");
}
