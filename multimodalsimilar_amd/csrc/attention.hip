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
#ifndef ATTN_FWD_WAVES
#define ATTN_FWD_WAVES 3      // 155 VGPRs, no spills: three 43.5 KiB workgroups per CU instead of two
#endif
#ifndef ATTN_BWD_WAVES
#define ATTN_BWD_WAVES 2
#endif
#define DST_PITCH 80    // bytes, dS^T image rows of 32 bf16
#ifndef ATTN_BWD_DS2
#define ATTN_BWD_DS2 1        // two dS^T images (alternating per query tile): the barrier after the dQ products is not needed
#endif
#ifndef ATTN_BWD_KLDS
#define ATTN_BWD_KLDS 1       // K^T fragments of the S / dS products read from the LDS image instead of held in 16 registers
#endif                        //   (measured at B = 256, S = 128, 16 heads: 158 -> 148 us; two dS^T images alone: no change)
#ifndef ATTN_BWD_MFMA_COLSUM
#define ATTN_BWD_MFMA_COLSUM 1   // v-bias column sums as ones x dV on the matrix cores (transposing reads of the dV image); 0: 2-byte LDS reads
#endif

struct AttnParams {
  const bf16* qkv; const int64_t* mask; bf16* ctx; const bf16* dctx; float* lse; bf16* dqkv;
  float* dbias_parts;    // backward: per-batch-element column sums of dqkv, slab [B][3H] (NULL = off)
  int ld_qkv, ld_ctx, heads, H;
  float scale;
  unsigned long long seed; const unsigned long long* seed_dev; unsigned int stream, thresh; float inv_keep;
};

__device__ __forceinline__ bf8 cvt8(const f16v& a, int s2, float mul) {
  bf8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = f2bf(a[8 * s2 + j] * mul);
  return r;
}

template <int NT>
__global__ __launch_bounds__(64 * NT) __attribute__((amdgpu_waves_per_eu(ATTN_FWD_WAVES))) void attn_fwd_kernel(AttnParams p) {
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
    const uint32_t dkey = drop_key(step_seed(p.seed, p.seed_dev), p.stream);
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

// Backward.  LDS holds the K image, ONE 32-row tile of Q and dO at a time (double-buffered: the next tile's q | dO | O rows
// are requested from HBM into registers before the current tile's products and parked in the other buffer after them, so
// the loads ride under the MFMA / exp work instead of in front of it), and the dS^T image: 48 KiB at S = 128.
// delta[q] = sum_d dO[q][d] O[q][d] comes out of the staging step (the thread that parks a 16-byte dO chunk also holds the
// matching O chunk: an 8-lane shuffle sum per row), so dO and O are read from HBM exactly once.
template <int NT>
__global__ __launch_bounds__(64 * NT) __attribute__((amdgpu_waves_per_eu(ATTN_BWD_WAVES))) void attn_bwd_kernel(AttnParams p) {
  constexpr int S = 32 * NT;
  constexpr int TPW = 8 / NT;   // dQ 16x16 tiles per wave per query tile
  constexpr int NTHR = 64 * NT;
  constexpr int CPT = 256 / NTHR;           // 16-byte chunks of a 32-row tile per thread and operand
  constexpr int TILE = 32 * ROW_PITCH;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Qt = smem;                          // [2][32][ROW_PITCH]
  char* Ot = smem + 2 * TILE;               // dO, [2][32][ROW_PITCH]
  char* Ks = smem + 4 * TILE;               // K [S][ROW_PITCH]
  char* Ds0 = Ks + S * ROW_PITCH;           // dS^T [S keys][DST_PITCH] for the current query tile (x2 with ATTN_BWD_DS2)
  float* fl = reinterpret_cast<float*>(Ds0 + (ATTN_BWD_DS2 ? 2 : 1) * S * DST_PITCH);
  float* lse = fl; float* dlt = fl + S; float* mb = fl + S + 64;     // dlt: [2][32] delta of the staged tiles
  float* cb = fl + 2 * S + 64;              // [NT][192] column sums of this (b, h)'s dq | dk | dv (bias gradient of the QKV projection),
                                            // one slot per wave: each wave adds into its own in program order, the slots are summed in wave order
                                            // (LDS float atomics from several waves arrive in varying order: last-bit run-to-run differences)
  float cq[TPW][4];
#pragma unroll
  for (int t = 0; t < TPW; ++t) cq[t][0] = cq[t][1] = cq[t][2] = cq[t][3] = 0.f;
  if (p.dbias_parts)
    for (int i = threadIdx.x; i < 192 * NT; i += NTHR) cb[i] = 0.f;       // ordered before the adds by the barriers of the loop
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, hh = lane >> 5;
  const int bh = blockIdx.x, b = bh / p.heads, h = bh % p.heads;
  const bf16* base = p.qkv + (size_t)b * S * p.ld_qkv + h * 64;
  const bf16* dob = p.dctx + (size_t)b * S * p.ld_ctx + h * 64;
  const bf16* ob = p.ctx + (size_t)b * S * p.ld_ctx + h * 64;

  uint4 tq[CPT], to[CPT], tc[CPT];          // the tile in flight: q, dO, O chunks
  auto issue = [&](int qt) {
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int c = tid + i * NTHR, row = 32 * qt + (c >> 3), ch = c & 7;
      tq[i] = *reinterpret_cast<const uint4*>(base + (size_t)row * p.ld_qkv + ch * 8);
      to[i] = *reinterpret_cast<const uint4*>(dob + (size_t)row * p.ld_ctx + ch * 8);
      tc[i] = *reinterpret_cast<const uint4*>(ob + (size_t)row * p.ld_ctx + ch * 8);
    }
  };
  auto park = [&](int bf) {
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int c = tid + i * NTHR, row = c >> 3, ch = c & 7;
      *reinterpret_cast<uint4*>(Qt + bf * TILE + row * ROW_PITCH + ch * 16) = tq[i];
      *reinterpret_cast<uint4*>(Ot + bf * TILE + row * ROW_PITCH + ch * 16) = to[i];
      const bf8 a = __builtin_bit_cast(bf8, to[i]), o = __builtin_bit_cast(bf8, tc[i]);
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += bf2f(a[j]) * bf2f(o[j]);
      acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64); acc += __shfl_xor(acc, 4, 64);   // the 8 chunks of a row
      if (ch == 0) dlt[bf * 32 + row] = acc;
    }
  };

  {   // K image (4 chunks per thread) + tile 0: every load requested before the first LDS store
    uint4 kv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + i * NTHR, row = c >> 3, ch = c & 7;
      kv[i] = *reinterpret_cast<const uint4*>(base + (size_t)row * p.ld_qkv + p.H + ch * 8);
    }
    issue(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + i * NTHR, row = c >> 3, ch = c & 7;
      *reinterpret_cast<uint4*>(Ks + row * ROW_PITCH + ch * 16) = kv[i];
    }
  }
  for (int i = tid; i < S; i += NTHR) {
    mb[i] = (p.mask && p.mask[(size_t)b * S + i] == 0) ? -1e30f : 0.f;
    lse[i] = p.lse[(size_t)bh * S + i];
  }
  // loop-invariant B operands: K^T and V^T columns for this wave's 32 keys (lane = key)
  const int key = 32 * w + (lane & 31);
  bf8 kreg[4], vreg[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    if (!ATTN_BWD_KLDS) kreg[kk] = *reinterpret_cast<const bf8*>(base + (size_t)key * p.ld_qkv + p.H + 16 * kk + 8 * hh);
    vreg[kk] = *reinterpret_cast<const bf8*>(base + (size_t)key * p.ld_qkv + 2 * p.H + 16 * kk + 8 * hh);
  }
  park(0);
  __syncthreads();
  const float mbk = mb[key];
  const uint32_t dkey = drop_key(step_seed(p.seed, p.seed_dev), p.stream);

  f16v dV[2], dK[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) { dV[dt][i] = 0.f; dK[dt][i] = 0.f; }

#pragma unroll 1
  for (int qt = 0; qt < NT; ++qt) {
    const int cur = qt & 1;
    const char* Qs = Qt + cur * TILE;       // rows of this query tile
    const char* Os = Ot + cur * TILE;
    const float* dl = dlt + cur * 32;
    char* Ds = Ds0 + (ATTN_BWD_DS2 ? cur * S * DST_PITCH : 0);
    if (qt + 1 < NT) issue(qt + 1);
    f16v X, dP;
#pragma unroll
    for (int i = 0; i < 16; ++i) { X[i] = 0.f; dP[i] = 0.f; }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int off = (lane & 31) * ROW_PITCH + (16 * kk + 8 * hh) * 2;
      const bf8 qa = *reinterpret_cast<const bf8*>(Qs + off);
      const bf8 oa = *reinterpret_cast<const bf8*>(Os + off);
      if (ATTN_BWD_KLDS) kreg[kk] = *reinterpret_cast<const bf8*>(Ks + key * ROW_PITCH + (16 * kk + 8 * hh) * 2);
      X = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kreg[kk], X, 0, 0, 0);
      dP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(oa, vreg[kk], dP, 0, 0, 0);
    }
    // X[r] -> P (dropped, for dV) ; dP[r] -> dS
    // Dropout mask: one hash serves the elements (q, key) and (q, key ^ 1), which sit on NEIGHBOURING LANES here (lane = key).
    // For the register pair (r, r + 1) = queries (q, q + 1) the even lane hashes q, the odd lane q + 1, and a quad-perm DPP
    // move swaps the results: 8 hashes + 8 moves per 16 elements instead of 16 hashes.
    const int par = key & 1;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      uint32_t bits0 = 0u, bits1 = 0u;
      if (p.thresh) {                       // kernel argument: wave-uniform, EXEC stays full for the DPP move
        const int qm = 32 * qt + (r & 3) + 8 * (r >> 2) + 4 * hh + par;
        const unsigned long long idx = ((unsigned long long)bh * S + qm) * S + key;
        const uint32_t mine = drop_bits(dkey, idx >> 1);
        const uint32_t other = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine, 0xB1, 0xF, 0xF, true);      // quad_perm [1,0,3,2]
        bits0 = par ? other : mine;
        bits1 = par ? mine : other;
      }
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        const int rr = r + k2;
        const int ql = (rr & 3) + 8 * (rr >> 2) + 4 * hh, q = 32 * qt + ql;
        const float pr = __expf(X[rr] * p.scale + mbk - lse[q]);
        float ks = 1.0f;
        if (p.thresh) ks = drop_keep16(k2 ? bits1 : bits0, par, p.thresh) ? p.inv_keep : 0.f;
        X[rr] = pr * ks;
        dP[rr] = pr * (dP[rr] * ks - dl[ql]);
      }
    }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf8 pb = cvt8(X, s2, 1.0f);
        const bf8 sb = cvt8(dP, s2, 1.0f);
        const bf8 of = tr_frag32(Os, ROW_PITCH, 0, s2, 32 * dt, lane);
        const bf8 qf = tr_frag32(Qs, ROW_PITCH, 0, s2, 32 * dt, lane);
        dV[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(of, pb, dV[dt], 0, 0, 0);
        dK[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf, sb, dK[dt], 0, 0, 0);
      }
    // dS^T[key][q_local] (bf16) -> LDS
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      bf4 o = {f2bf(dP[4 * g4]), f2bf(dP[4 * g4 + 1]), f2bf(dP[4 * g4 + 2]), f2bf(dP[4 * g4 + 3])};
      *reinterpret_cast<bf4*>(Ds + key * DST_PITCH + (8 * g4 + 4 * hh) * 2) = o;
    }
    if (qt + 1 < NT) park(cur ^ 1);       // the other buffer was last read before the previous iteration's first barrier
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
    // one dS^T image: the next tile's writes must wait for these reads.  Two images: the image written next was last read before
    // the barrier above of the PREVIOUS iteration, the Q / dO buffer parked next was last read before this iteration's barrier
    if (!ATTN_BWD_DS2) __syncthreads();
  }
  if (ATTN_BWD_DS2) __syncthreads();         // the K image and tile buffers are reused below
  // dK, dV leave through LDS (the K image and the Q / dO tile buffers are free after the loop's last barrier): the
  // accumulators hold 4 channels of one key per register quad -- written straight out that is 8-byte pieces of 32 different
  // cache lines per store; from the [key][64] images every row leaves as one 128-byte line, and the column sums for the k | v
  // bias gradients are 32 two-byte LDS reads per thread instead of ten 5-step shuffle trees per lane.
  char* dKs = Ks;                 // [S][ROW_PITCH]
  char* dVs = smem;               // [S][ROW_PITCH] over the four 32-row tile buffers (S <= 128)
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int d = 32 * dt + 8 * g4 + 4 * hh;
      const bf4 ok = {f2bf(dK[dt][4 * g4] * p.scale), f2bf(dK[dt][4 * g4 + 1] * p.scale), f2bf(dK[dt][4 * g4 + 2] * p.scale),
                      f2bf(dK[dt][4 * g4 + 3] * p.scale)};
      const bf4 ov = {f2bf(dV[dt][4 * g4]), f2bf(dV[dt][4 * g4 + 1]), f2bf(dV[dt][4 * g4 + 2]), f2bf(dV[dt][4 * g4 + 3])};
      *reinterpret_cast<bf4*>(dKs + key * ROW_PITCH + d * 2) = ok;
      *reinterpret_cast<bf4*>(dVs + key * ROW_PITCH + d * 2) = ov;
    }
  __syncthreads();
  {
    bf16* dkg = p.dqkv + (size_t)b * S * p.ld_qkv + p.H + h * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) {       // S*8 chunks of 16 bytes per tensor over 64*NT threads
      const int c = tid + i * NTHR, row = c >> 3, ch = c & 7;
      const uint4 kq = *reinterpret_cast<const uint4*>(dKs + row * ROW_PITCH + ch * 16);
      const uint4 vq = *reinterpret_cast<const uint4*>(dVs + row * ROW_PITCH + ch * 16);
      *reinterpret_cast<uint4*>(dkg + (size_t)row * p.ld_qkv + ch * 8) = kq;
      *reinterpret_cast<uint4*>(dkg + (size_t)row * p.ld_qkv + p.H + ch * 8) = vq;
    }
  }
  if (p.dbias_parts && ATTN_BWD_MFMA_COLSUM) {
    // key / value bias gradients: column sums of the ROUNDED dK / dV images as (all-ones [16 x 32 keys]) x image[32 keys x 16 d] on
    // the matrix cores -- this wave's 32 keys, four 16-column blocks per image: 8 MFMAs + 16 transposing reads per wave instead of
    // 64 two-byte LDS reads per lane.  Every row of the product is the same sum: row 0 is kept.  (The key-bias sum is analytically
    // zero -- softmax is invariant to a per-query constant -- and what is summed here is the rounding noise of the stored dK, as the
    // reference's autograd sums its own; kept so that dbias == column sums of the tensor that was written, exactly.)
    bf8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;
#pragma unroll
    for (int d4 = 0; d4 < 4; ++d4) {
      f4 ak = {0.f, 0.f, 0.f, 0.f}, av = {0.f, 0.f, 0.f, 0.f};
      ak = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag16(dKs, ROW_PITCH, 32 * w, 16 * d4, lane), ones, ak, 0, 0, 0);
      av = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag16(dVs, ROW_PITCH, 32 * w, 16 * d4, lane), ones, av, 0, 0, 0);
      // operands swapped as everywhere here: acc[e] = C[m = lane & 15][n = 16 d4 + 4 (lane >> 4) + e], all rows m equal
      if ((lane & 15) == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          cb[192 * w + 64 + 16 * d4 + 4 * (lane >> 4) + e] += ak[e];
          cb[192 * w + 128 + 16 * d4 + 4 * (lane >> 4) + e] += av[e];
        }
      }
    }
  } else if (p.dbias_parts) {          // sums of the ROUNDED values: thread = (column, wave's 32 keys)
    const int d = lane, k0 = 32 * w;
    float sk = 0.f, sv = 0.f;
#pragma unroll 8
    for (int r = 0; r < 32; ++r) {
      sk += bf2f(*reinterpret_cast<const bf16*>(dKs + (k0 + r) * ROW_PITCH + d * 2));
      sv += bf2f(*reinterpret_cast<const bf16*>(dVs + (k0 + r) * ROW_PITCH + d * 2));
    }
    cb[192 * w + 64 + d] += sk;            // this wave's slot: lane d is the only writer of column d
    cb[192 * w + 128 + d] += sv;
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
        if ((lane & 15) == 0) cb[192 * w + 16 * d4 + (lane >> 4) * 4 + e] += sq;      // same lane for a repeated d4: program order
      }
    }
    __syncthreads();
    for (int i = tid; i < 192; i += 64 * NT) {
      float t = cb[i];
#pragma unroll
      for (int ww = 1; ww < NT; ++ww) t += cb[192 * ww + i];
      p.dbias_parts[(size_t)b * 3 * p.H + (i >> 6) * p.H + h * 64 + (i & 63)] = t;
    }
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
  p.seed = seed; p.seed_dev = mmsim_step_seed_ptr(); p.stream = stream_id;
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

static size_t attn_bwd_lds(int S) {      // Q / dO tile pairs + K image + dS^T image + lse[S], delta[2][32], mask[S], bias sums[192]
  return (size_t)4 * 32 * ROW_PITCH + (size_t)S * (ROW_PITCH + (ATTN_BWD_DS2 ? 2 : 1) * DST_PITCH) + (size_t)(2 * S + 64 + 192 * (S / 32)) * 4;      // bias sums: one [192] slot per wave
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
  p.seed = seed; p.seed_dev = mmsim_step_seed_ptr(); p.stream = stream_id;
  p.thresh = dropout_p > 0.f ? (unsigned int)((double)dropout_p * 4294967296.0) : 0u;
  p.inv_keep = 1.0f / (1.0f - dropout_p);
  const int NT = S / 32;
  const size_t lds = attn_bwd_lds(S);
  hipStream_t s = (hipStream_t)stream;
  static unsigned long long attr_done = 0;          // per device
  const int dev = mmsim_current_device();
  if (!((attr_done >> dev) & 1)) {   // S = 128 needs 48 KiB of dynamic LDS
    (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)attn_bwd_lds(128));
    attr_done |= 1ull << dev;
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
