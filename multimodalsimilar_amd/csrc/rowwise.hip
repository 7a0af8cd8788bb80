// HBM-bound row kernels of the text tower and the head for gfx950: one wave (64 lanes) per row, 4-element
// vectors per lane, fp32 statistics, wave shuffles for the row reductions, per-wave register partials +
// one atomic per column per block for the column reductions (dgamma / dbeta / dbias).
//   embed_ln_fwd / embed_ln_bwd        BERT embeddings  (modeling_bert.py:68-108)
//   add_ln_fwd / ln_bwd                LayerNorm(dropout(t) + residual)  (modeling_bert.py:289-293, 347-351)
//   colsum                             bias gradients
//   l2norm_fwd / l2norm_bwd            F.normalize rows (arcface.py:47, multimodal_classifier.py:54-55)
#include "common.h"

#define MAXC 8   // max 4-element chunks per lane: H <= 2048

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const bf16* p) {
  const bf4 v = *reinterpret_cast<const bf4*>(p);
  return make_float4(bf2f(v[0]), bf2f(v[1]), bf2f(v[2]), bf2f(v[3]));
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(bf16* p, float4 v) {
  bf4 o = {f2bf(v.x), f2bf(v.y), f2bf(v.z), f2bf(v.w)};
  *reinterpret_cast<bf4*>(p) = o;
}

struct Drop { unsigned long long seed; const unsigned long long* seed_dev; unsigned int stream, thresh; float inv_keep; };
static Drop make_drop(float p, unsigned long long seed, unsigned int stream) {
  Drop d; d.seed = seed; d.seed_dev = mmsim_step_seed_ptr(); d.stream = stream;
  d.thresh = p > 0.f ? (unsigned int)((double)p * 4294967296.0) : 0u;
  d.inv_keep = 1.0f / (1.0f - p);
  return d;
}
__device__ __forceinline__ float4 drop4(float4 v, const Drop& d, unsigned long long idx) {     // idx: a multiple of 4
  if (!d.thresh) return v;
  const uint32_t key = drop_key(step_seed(d.seed, d.seed_dev), d.stream);
  const uint32_t h0 = drop_bits(key, idx >> 1), h1 = drop_bits(key, (idx >> 1) + 1);
  v.x = drop_keep16(h0, 0, d.thresh) ? v.x * d.inv_keep : 0.f;
  v.y = drop_keep16(h0, 1, d.thresh) ? v.y * d.inv_keep : 0.f;
  v.z = drop_keep16(h1, 0, d.thresh) ? v.z * d.inv_keep : 0.f;
  v.w = drop_keep16(h1, 1, d.thresh) ? v.w * d.inv_keep : 0.f;
  return v;
}

// normalise the row held in v[] (nc chunks per lane); returns mean / rstd
__device__ __forceinline__ void row_stats(const float4 (&v)[MAXC], int nc, int H, float eps, float& mean, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
    if (c < nc) s += v[c].x + v[c].y + v[c].z + v[c].w;
  mean = wave_sum(s) / H;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
    if (c < nc) {
      const float a = v[c].x - mean, b = v[c].y - mean, cc = v[c].z - mean, d = v[c].w - mean;
      q += a * a + b * b + cc * cc + d * d;
    }
  rstd = rsqrtf(wave_sum(q) / H + eps);
}

// ---------------------------------------------------------------- embeddings + LayerNorm
// Token / token-type / position indices index tables: an index outside its table raises the device error flag (read by the
// host where it reads the head's label flag) and is clamped, so that neither the gather nor the backward's scatter-add
// leaves its table (HF: nn.Embedding raises an IndexError).
struct EmbedIdx { int64_t id, tt, s; };
__device__ __forceinline__ EmbedIdx embed_idx(const int64_t* ids, const int64_t* tts, const int64_t* pids, int row, int S, int vocab,
                                              int tvocab, int max_pos, int* err) {
  EmbedIdx e;
  e.id = ids[row]; e.tt = tts ? tts[row] : 0; e.s = pids ? pids[row] : (int64_t)(row % S);
  const bool bad = e.id < 0 || e.id >= vocab || e.tt < 0 || e.tt >= tvocab || e.s < 0 || e.s >= max_pos;
  if (bad) {
    if ((threadIdx.x & 63) == 0) atomicOr(err, 1);
    e.id = e.id < 0 ? 0 : (e.id >= vocab ? vocab - 1 : e.id);
    e.tt = e.tt < 0 ? 0 : (e.tt >= tvocab ? tvocab - 1 : e.tt);
    e.s = e.s < 0 ? 0 : (e.s >= max_pos ? max_pos - 1 : e.s);
  }
  return e;
}

__global__ __launch_bounds__(256) void embed_ln_fwd_kernel(const int64_t* ids, const int64_t* tts, const int64_t* pids, const float* word,
                                                           const float* pos, const float* type, const float* gamma,
                                                           const float* beta, bf16* out, int M, int S, int H, float eps,
                                                           Drop dr, int vocab, int tvocab, int max_pos, int* err) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wv;
  if (row >= M) return;
  const EmbedIdx ei = embed_idx(ids, tts, pids, row, S, vocab, tvocab, max_pos, err);
  const int64_t id = ei.id, tt = ei.tt, s = ei.s;
  float4 v[MAXC];
  int nc = 0;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int col = lane * 4 + c * 256;
    if (col < H) {
      const float4 a = ld4(word + (size_t)id * H + col), b = ld4(pos + (size_t)s * H + col), t = ld4(type + (size_t)tt * H + col);
      v[c] = make_float4(a.x + b.x + t.x, a.y + b.y + t.y, a.z + b.z + t.z, a.w + b.w + t.w);
      nc = c + 1;
    }
  }
  float mean, rstd;
  row_stats(v, nc, H, eps, mean, rstd);
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int col = lane * 4 + c * 256;
    if (col < H) {
      const float4 g = ld4(gamma + col), bb = ld4(beta + col);
      float4 o = make_float4((v[c].x - mean) * rstd * g.x + bb.x, (v[c].y - mean) * rstd * g.y + bb.y,
                             (v[c].z - mean) * rstd * g.z + bb.z, (v[c].w - mean) * rstd * g.w + bb.w);
      o = drop4(o, dr, (unsigned long long)row * H + col);
      st4(out + (size_t)row * H + col, o);
    }
  }
}

// backward of the embedding block.  Each wave owns one position s and a chunk of the batch, so the
// position/type/gamma/beta gradients accumulate in registers and leave as one atomic row per wave.
__global__ __launch_bounds__(256) void embed_ln_bwd_kernel(const bf16* dout, const int64_t* ids, const int64_t* tts, const int64_t* pids,
                                                           const float* word, const float* pos, const float* type,
                                                           const float* gamma, float* dword, float* dpos, float* dtype,
                                                           float* dgamma, float* dbeta, int B, int S, int H, float eps,
                                                           int bchunk, Drop dr, int vocab, int tvocab, int max_pos, int* err, int serial,
                                                           float* parts) {
  __shared__ __attribute__((aligned(16))) float rowbuf[4][MAXC * 256];       // one d(embedding) row per wave
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int nchunks = (B + bchunk - 1) / bchunk;
  // serial (deterministic mode): ONE wave walks every (position, batch chunk) in order, so every atomic below has a single
  // adder issuing in program order; otherwise a wave per (position, batch chunk)
  const int gw_first = serial ? 0 : blockIdx.x * 4 + wv, gw_last = serial ? S * nchunks : gw_first + 1;
  const bool idle = serial ? (blockIdx.x != 0 || wv != 0) : (gw_first >= S * nchunks);
  if (idle && !parts) return;
  // Column sums that EVERY wave contributes to (dgamma, dbeta, the two token-type rows): with `parts` (never in the serial mode; one trip
  // of the loop below per wave) they leave through the workgroup's LDS into one slab row per workgroup, summed by
  // row_reduce_partials_kernel -- an idle wave of the last workgroup walks an empty row range and adds zeros.  As atomics they were
  // 2 048 adds per address at B = 256, S = 128, ~150 ns each, one after the other: 310 of the kernel's 460 us (r03, tools/bench_embed.py).
#pragma unroll 1
  for (int gw = gw_first; gw < gw_last; ++gw) {
  const int sw = gw % S, b0 = (gw / S) * bchunk, b1 = idle ? b0 : min(B, b0 + bchunk);
  float4 apos[MAXC], at0[MAXC], at1[MAXC], ag[MAXC], ab[MAXC];
#pragma unroll
  for (int c = 0; c < MAXC; ++c) apos[c] = at0[c] = at1[c] = ag[c] = ab[c] = make_float4(0, 0, 0, 0);
  for (int b = b0; b < b1; ++b) {
    const int row = b * S + sw;
    const EmbedIdx ei = embed_idx(ids, tts, pids, row, S, vocab, tvocab, max_pos, err);
    const int64_t id = ei.id, tt = ei.tt, s = ei.s;      // s == sw unless explicit position ids were given
    float4 v[MAXC], g[MAXC];
    int nc = 0;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int col = lane * 4 + c * 256;
      if (col < H) {
        const float4 a = ld4(word + (size_t)id * H + col), p2 = ld4(pos + (size_t)s * H + col), t = ld4(type + (size_t)tt * H + col);
        v[c] = make_float4(a.x + p2.x + t.x, a.y + p2.y + t.y, a.z + p2.z + t.z, a.w + p2.w + t.w);
        g[c] = drop4(ld4(dout + (size_t)row * H + col), dr, (unsigned long long)row * H + col);
        nc = c + 1;
      }
    }
    float mean, rstd;
    row_stats(v, nc, H, eps, mean, rstd);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int col = lane * 4 + c * 256;
      if (col < H) {
        const float4 gm = ld4(gamma + col);
        float4 xh = make_float4((v[c].x - mean) * rstd, (v[c].y - mean) * rstd, (v[c].z - mean) * rstd, (v[c].w - mean) * rstd);
        ag[c].x += g[c].x * xh.x; ag[c].y += g[c].y * xh.y; ag[c].z += g[c].z * xh.z; ag[c].w += g[c].w * xh.w;
        ab[c].x += g[c].x; ab[c].y += g[c].y; ab[c].z += g[c].z; ab[c].w += g[c].w;
        g[c] = make_float4(g[c].x * gm.x, g[c].y * gm.y, g[c].z * gm.z, g[c].w * gm.w);   // dxhat
        s1 += g[c].x + g[c].y + g[c].z + g[c].w;
        s2 += g[c].x * xh.x + g[c].y * xh.y + g[c].z * xh.z + g[c].w * xh.w;
        v[c] = xh;
      }
    }
    s1 = wave_sum(s1) / H; s2 = wave_sum(s2) / H;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int col = lane * 4 + c * 256;
      if (col < H) {
        const float4 de = make_float4(rstd * (g[c].x - s1 - v[c].x * s2), rstd * (g[c].y - s1 - v[c].y * s2),
                                      rstd * (g[c].z - s1 - v[c].z * s2), rstd * (g[c].w - s1 - v[c].w * s2));
        *reinterpret_cast<float4*>(rowbuf[wv] + col) = de;
        apos[c].x += de.x; apos[c].y += de.y; apos[c].z += de.z; apos[c].w += de.w;
        if (tt == 0) { at0[c].x += de.x; at0[c].y += de.y; at0[c].z += de.z; at0[c].w += de.w; }
        else { at1[c].x += de.x; at1[c].y += de.y; at1[c].z += de.z; at1[c].w += de.w; }
      }
    }
    // scatter-add of the row into the word-embedding gradient: through the wave's LDS row so that ONE atomic instruction
    // covers 256 contiguous bytes (lane = column); straight from the float4 registers every instruction touched 4 bytes
    // out of every 16 across eight cache lines, four times over
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float* dw = dword + (size_t)id * H;
#pragma unroll
    for (int j = 0; j < MAXC * 4; ++j) {
      const int col = j * 64 + lane;
      if (col < H) atomicAdd(dw + col, rowbuf[wv][col]);
    }
    if (pids) {      // explicit position ids: the position row differs from row to row, scatter-add it like the word row
      float* dp = dpos + (size_t)s * H;
#pragma unroll
      for (int j = 0; j < MAXC * 4; ++j) {
        const int col = j * 64 + lane;
        if (col < H) atomicAdd(dp + col, rowbuf[wv][col]);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int col = lane * 4 + c * 256;
    if (col < H) {
      float* a = dpos + (size_t)sw * H + col;
      if (!pids && !idle) { atomicAdd(a, apos[c].x); atomicAdd(a + 1, apos[c].y); atomicAdd(a + 2, apos[c].z); atomicAdd(a + 3, apos[c].w); }
      if (parts) continue;
      a = dtype + col;
      atomicAdd(a, at0[c].x); atomicAdd(a + 1, at0[c].y); atomicAdd(a + 2, at0[c].z); atomicAdd(a + 3, at0[c].w);
      a = dtype + H + col;
      atomicAdd(a, at1[c].x); atomicAdd(a + 1, at1[c].y); atomicAdd(a + 2, at1[c].z); atomicAdd(a + 3, at1[c].w);
      a = dgamma + col;
      atomicAdd(a, ag[c].x); atomicAdd(a + 1, ag[c].y); atomicAdd(a + 2, ag[c].z); atomicAdd(a + 3, ag[c].w);
      a = dbeta + col;
      atomicAdd(a, ab[c].x); atomicAdd(a + 1, ab[c].y); atomicAdd(a + 2, ab[c].z); atomicAdd(a + 3, ab[c].w);
    }
  }
  if (parts) {      // workgroup-uniform: slab row [4][H] = dgamma | dbeta | dtype[0] | dtype[1] of this workgroup
    float* slab = parts + (size_t)blockIdx.x * 4 * H;
#pragma unroll
    for (int vec = 0; vec < 4; ++vec) {
      __syncthreads();
#pragma unroll
      for (int c = 0; c < MAXC; ++c) {
        const int col = lane * 4 + c * 256;
        if (col < H) *reinterpret_cast<float4*>(rowbuf[wv] + col) = vec == 0 ? ag[c] : vec == 1 ? ab[c] : vec == 2 ? at0[c] : at1[c];
      }
      __syncthreads();
      for (int col = threadIdx.x * 4; col < H; col += 1024) {
        const float4 a = *reinterpret_cast<const float4*>(rowbuf[0] + col), b = *reinterpret_cast<const float4*>(rowbuf[1] + col);
        const float4 c4 = *reinterpret_cast<const float4*>(rowbuf[2] + col), d = *reinterpret_cast<const float4*>(rowbuf[3] + col);
        *reinterpret_cast<float4*>(slab + (size_t)vec * H + col) = make_float4((a.x + b.x) + (c4.x + d.x), (a.y + b.y) + (c4.y + d.y),
                                                                               (a.z + b.z) + (c4.z + d.z), (a.w + b.w) + (c4.w + d.w));
      }
    }
  }
  }
}

// ---------------------------------------------------------------- y = dropout(t) + resid ; h = LN(y)
__global__ __launch_bounds__(256) void add_ln_fwd_kernel(const bf16* t, const bf16* resid, const float* gamma,
                                                         const float* beta, bf16* y, bf16* hout, float* mean_o,
                                                         float* rstd_o, int M, int H, float eps, Drop dr) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wv;
  if (row >= M) return;
  float4 v[MAXC];
  int nc = 0;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int col = lane * 4 + c * 256;
    if (col < H) {
      float4 a = drop4(ld4(t + (size_t)row * H + col), dr, (unsigned long long)row * H + col);
      const float4 r = ld4(resid + (size_t)row * H + col);
      a = make_float4(a.x + r.x, a.y + r.y, a.z + r.z, a.w + r.w);
      // statistics are taken on the bf16-rounded sum that backward will read back
      bf4 rb = {f2bf(a.x), f2bf(a.y), f2bf(a.z), f2bf(a.w)};
      __builtin_nontemporal_store(rb, reinterpret_cast<bf4*>(y + (size_t)row * H + col));      // read again only in backward
      v[c] = make_float4(bf2f(rb[0]), bf2f(rb[1]), bf2f(rb[2]), bf2f(rb[3]));
      nc = c + 1;
    }
  }
  float mean, rstd;
  row_stats(v, nc, H, eps, mean, rstd);
  if (lane == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int col = lane * 4 + c * 256;
    if (col < H) {
      const float4 g = ld4(gamma + col), bb = ld4(beta + col);
      st4(hout + (size_t)row * H + col,
          make_float4((v[c].x - mean) * rstd * g.x + bb.x, (v[c].y - mean) * rstd * g.y + bb.y,
                      (v[c].z - mean) * rstd * g.z + bb.z, (v[c].w - mean) * rstd * g.w + bb.w));
    }
  }
}

// The same with 8 columns per lane per 512-column chunk (16-byte accesses) and TWO rows per wave in flight (H % 8 == 0, H <= 1024: the
// text tower): the one-row form above keeps 8 bytes per lane per stream in flight and ran at 4.7 TB/s on [32768, 1024].
#ifndef ADD_LN_FWD8
#define ADD_LN_FWD8 1
#endif
template <int NC>      // NC = ceil(H / 512)
__global__ __launch_bounds__(256) void add_ln_fwd8_kernel(const bf16* __restrict__ t, const bf16* __restrict__ resid, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, bf16* __restrict__ y, bf16* __restrict__ hout,
                                                          float* mean_o, float* rstd_o, int M, int H, float eps, Drop dr) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int row0 = (blockIdx.x * 4 + wv) * 2;
  if (row0 >= M) return;
  const bool two = row0 + 1 < M;
  bf8 tv[2][NC], rv[2][NC];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const size_t base = (size_t)(row0 + (q && two ? 1 : 0)) * H;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int col = lane * 8 + c * 512;
      if (col < H) {      // (the branch-free form that pays in ln_bwd8_kernel's row LOOP measured 1-2 % slower in this loop-free kernel)
        tv[q][c] = __builtin_nontemporal_load(reinterpret_cast<const bf8*>(t + base + col));      // the dense output: read once
        rv[q][c] = *reinterpret_cast<const bf8*>(resid + base + col);
      }
    }
  }
  float v[2][NC][8], s[2] = {0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const size_t base = (size_t)(row0 + (q && two ? 1 : 0)) * H;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int col = lane * 8 + c * 512;
      if (col < H) {
        float4 a0 = make_float4(bf2f(tv[q][c][0]), bf2f(tv[q][c][1]), bf2f(tv[q][c][2]), bf2f(tv[q][c][3]));
        float4 a1 = make_float4(bf2f(tv[q][c][4]), bf2f(tv[q][c][5]), bf2f(tv[q][c][6]), bf2f(tv[q][c][7]));
        a0 = drop4(a0, dr, (unsigned long long)base + col);
        a1 = drop4(a1, dr, (unsigned long long)base + col + 4);
        const float a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        bf8 rb;      // statistics are taken on the bf16-rounded sum that backward will read back
#pragma unroll
        for (int e = 0; e < 8; ++e) { rb[e] = f2bf(a[e] + bf2f(rv[q][c][e])); v[q][c][e] = bf2f(rb[e]); s[q] += v[q][c][e]; }
        if (q == 0 || two) __builtin_nontemporal_store(rb, reinterpret_cast<bf8*>(y + base + col));      // read again only in backward
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[q][c][e] = 0.f;
      }
    }
  }
  wave_sum2(s[0], s[1]);
  const float mean[2] = {s[0] / H, s[1] / H};
  float qq[2] = {0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int col = lane * 8 + c * 512;
      if (col < H) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[q][c][e] - mean[q]; qq[q] += d * d; }
      }
    }
  wave_sum2(qq[0], qq[1]);
  const float rstd[2] = {rsqrtf(qq[0] / H + eps), rsqrtf(qq[1] / H + eps)};
  if (lane == 0) {
    mean_o[row0] = mean[0]; rstd_o[row0] = rstd[0];
    if (two) { mean_o[row0 + 1] = mean[1]; rstd_o[row0 + 1] = rstd[1]; }
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int col = lane * 8 + c * 512;
    if (col < H) {
      const float4 g0 = ld4(gamma + col), g1 = ld4(gamma + col + 4), b0 = ld4(beta + col), b1 = ld4(beta + col + 4);
      const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if (q == 1 && !two) break;
        bf8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf((v[q][c][e] - mean[q]) * rstd[q] * g[e] + bb[e]);
        *reinterpret_cast<bf8*>(hout + (size_t)(row0 + q) * H + col) = o;      // the next GEMM's A operand: cached store
      }
    }
  }
}

// LayerNorm backward.  dh = dh_a (+ dh_b).  Outputs dy (gradient of the pre-LN sum: goes to the residual
// branch) and dt = dropout-mask(dy) (gradient of the dense output; same buffer as dy when p == 0), plus
// column sums: dgamma, dbeta, dbias (= colsum dt).
// out[i] += sum_p parts[p*part_stride + i]  (64 outputs x 4 waves per block; gridDim.y part-chunks, one atomic each)
// The slab row holds up to three vectors of H columns back to back (dgamma | dbeta | dbias): ONE launch reduces all of them, the
// output pointer is chosen per column (a NULL output is skipped).
struct RowOuts { float* o[4]; int H; };
template <bool ATOMIC>
__global__ __launch_bounds__(256) void row_reduce_partials_kernel(const float* __restrict__ parts, int nparts, size_t part_stride,
                                                                  int n, RowOuts outs) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  const int kk = i < n ? i / outs.H : 0;
  float* out = (kk == 0 ? outs.o[0] : kk == 1 ? outs.o[1] : kk == 2 ? outs.o[2] : outs.o[3]);
  if (out) out -= (size_t)kk * outs.H;          // so that out[i] addresses column i - kk*H of vector kk
  float a0 = 0.f, a1 = 0.f;
  if (i < n) {
    const int step = gridDim.y * 4;
    int p = blockIdx.y * 4 + wv;
    for (; p + step < nparts; p += 2 * step) { a0 += parts[(size_t)p * part_stride + i]; a1 += parts[(size_t)(p + step) * part_stride + i]; }
    if (p < nparts) a0 += parts[(size_t)p * part_stride + i];
  }
  red[wv][lane] = a0 + a1;
  __syncthreads();
  if (wv == 0 && i < n && out) {
    const float t = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    if (ATOMIC) atomicAdd(out + i, t);
    else out[i] += t;          // deterministic mode: one workgroup per output, fixed order
  }
}

template <int NC>     // NC = ceil(H / 256) chunks per lane: keeps the per-lane column accumulators at 3*4*NC registers
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16* dh_a, const bf16* dh_b, const bf16* y, const float* mean_i,
                                                     const float* rstd_i, const float* gamma, bf16* dy, bf16* dt,
                                                     float* parts, int M, int H, int rows_per_wave, Drop dr) {
  __shared__ float red[3][4][64 * 4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r0 = (blockIdx.x * 4 + wv) * rows_per_wave, r1 = min(M, r0 + rows_per_wave);
  float4 ag[NC], ab[NC], abias[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) ag[c] = ab[c] = abias[c] = make_float4(0, 0, 0, 0);
  for (int row = r0; row < r1; ++row) {
    const float mean = mean_i[row], rstd = rstd_i[row];
    float4 g[NC], xh[NC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int col = lane * 4 + c * 256;
      if (col < H) {
        float4 d = ld4(dh_a + (size_t)row * H + col);
        if (dh_b) {
          const float4 e = ld4(dh_b + (size_t)row * H + col);
          d = make_float4(d.x + e.x, d.y + e.y, d.z + e.z, d.w + e.w);
        }
        const float4 yy = ld4(y + (size_t)row * H + col), gm = ld4(gamma + col);
        xh[c] = make_float4((yy.x - mean) * rstd, (yy.y - mean) * rstd, (yy.z - mean) * rstd, (yy.w - mean) * rstd);
        ag[c].x += d.x * xh[c].x; ag[c].y += d.y * xh[c].y; ag[c].z += d.z * xh[c].z; ag[c].w += d.w * xh[c].w;
        ab[c].x += d.x; ab[c].y += d.y; ab[c].z += d.z; ab[c].w += d.w;
        g[c] = make_float4(d.x * gm.x, d.y * gm.y, d.z * gm.z, d.w * gm.w);
        s1 += g[c].x + g[c].y + g[c].z + g[c].w;
        s2 += g[c].x * xh[c].x + g[c].y * xh[c].y + g[c].z * xh[c].z + g[c].w * xh[c].w;
      }
    }
    s1 = wave_sum(s1) / H; s2 = wave_sum(s2) / H;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int col = lane * 4 + c * 256;
      if (col < H) {
        const float4 o = make_float4(rstd * (g[c].x - s1 - xh[c].x * s2), rstd * (g[c].y - s1 - xh[c].y * s2),
                                     rstd * (g[c].z - s1 - xh[c].z * s2), rstd * (g[c].w - s1 - xh[c].w * s2));
        st4(dy + (size_t)row * H + col, o);
        float4 od = o;
        if (dr.thresh) {
          od = drop4(o, dr, (unsigned long long)row * H + col);
          st4(dt + (size_t)row * H + col, od);
        }
        // the dense-output gradient the GEMMs consume is the bf16-rounded value
        abias[c].x += bf2f(f2bf(od.x)); abias[c].y += bf2f(f2bf(od.y)); abias[c].z += bf2f(f2bf(od.z)); abias[c].w += bf2f(f2bf(od.w));
      }
    }
  }
  // cross-wave reduction through LDS, then the block's partial column sums go to its slot parts[blockIdx.x][3][H]
  // (a second launch sums the slots: hundreds of blocks adding atomically into the same H addresses serialise)
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int col = lane * 4 + c * 256;
    if (c * 256 >= H) break;   // uniform
    __syncthreads();
    float4* r0p = reinterpret_cast<float4*>(&red[0][wv][lane * 4]);
    float4* r1p = reinterpret_cast<float4*>(&red[1][wv][lane * 4]);
    float4* r2p = reinterpret_cast<float4*>(&red[2][wv][lane * 4]);
    *r0p = ag[c]; *r1p = ab[c]; *r2p = abias[c];
    __syncthreads();
    if (wv < 3 && col < H) {
      float4 a = make_float4(0, 0, 0, 0);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float4 b = *reinterpret_cast<const float4*>(&red[wv][k][lane * 4]);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
      }
      *reinterpret_cast<float4*>(parts + ((size_t)blockIdx.x * 3 + wv) * H + col) = a;
    }
  }
}

// 8 columns per lane per 512-column chunk (16-byte loads / stores) and TWO rows per wave in flight: the one-row form
// above is latency-bound (each row is load -> two wave reductions -> store, 8 bytes per lane per stream in flight).
// (Round 3: a one-row, next-row-prefetched form at three waves per SIMD -- 162 VGPRs instead of 236 -- measured the same 72-74 us on
// [32768, 1024] with dropout, tools/bench_ln.py; not kept.)
// Loads and arithmetic of the row loop carry NO per-lane branch (r03): lanes past H read column 0 and have their gradient zeroed, the
// second gradient input is a template argument.  Behind `if (col < H)` / `if (dh_b)` hipcc could not pair a load with its use across
// the two branches, assumed loads still pending at the loop's back edge and opened every trip with s_waitcnt vmcnt(0) -- a wait for
// the previous trip's STORES to be acknowledged.
template <int NC, bool HASB>      // NC = ceil(H / 512); HASB: dh_b is given
__global__ __launch_bounds__(256) void ln_bwd8_kernel(const bf16* __restrict__ dh_a, const bf16* __restrict__ dh_b, const bf16* __restrict__ y,
                                                      const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                      const float* __restrict__ gamma, bf16* __restrict__ dy, bf16* __restrict__ dt,
                                                      float* parts, int M, int H, int rows_per_wave, Drop dr) {
  __shared__ float red[3][4][64 * 8];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r0 = (blockIdx.x * 4 + wv) * rows_per_wave, r1 = min(M, r0 + rows_per_wave);
  float ag[NC][8], ab[NC][8], abias[NC][8], gm[NC][8];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int col = lane * 8 + c * 512;
#pragma unroll
    for (int e = 0; e < 8; ++e) { ag[c][e] = ab[c][e] = abias[c][e] = 0.f; gm[c][e] = col < H ? gamma[col + e] : 0.f; }
  }
  for (int row = r0; row < r1; row += 2) {
    const bool two = row + 1 < r1;
    float g[2][NC][8], xh[2][NC][8], s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f}, rs[2];
    bf8 dv[2][NC], ev[2][NC], yv[2][NC];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const size_t base = (size_t)(row + (q && two ? 1 : 0)) * H;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int col = lane * 8 + c * 512, colc = col < H ? col : 0;
        dv[q][c] = *reinterpret_cast<const bf8*>(dh_a + base + colc);
        if (HASB) ev[q][c] = *reinterpret_cast<const bf8*>(dh_b + base + colc);
        yv[q][c] = *reinterpret_cast<const bf8*>(y + base + colc);
      }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int rr = row + (q && two ? 1 : 0);
      const float mean = mean_i[rr];
      rs[q] = rstd_i[rr];
      const bool live = q == 0 || two;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int col = lane * 8 + c * 512;
        const bool use = live && col < H;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float d = bf2f(dv[q][c][e]);
          if (HASB) d += bf2f(ev[q][c][e]);
          d = use ? d : 0.f;
          const float x = (bf2f(yv[q][c][e]) - mean) * rs[q];
          xh[q][c][e] = x;
          ag[c][e] += d * x; ab[c][e] += d;
          const float gg = d * gm[c][e];
          g[q][c][e] = gg;
          s1[q] += gg; s2[q] += gg * x;
        }
      }
    }
    wave_sum4(s1[0], s2[0], s1[1], s2[1]);       // the four row sums reduce together: one shuffle latency chain, not four
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (q == 1 && !two) break;
      const size_t base = (size_t)(row + q) * H;
      const float m1 = s1[q] / H, m2 = s2[q] / H;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int col = lane * 8 + c * 512;
        if (col < H) {
          float o[8];
          bf8 ob;
#pragma unroll
          for (int e = 0; e < 8; ++e) { o[e] = rs[q] * (g[q][c][e] - m1 - xh[q][c][e] * m2); ob[e] = f2bf(o[e]); }
          *reinterpret_cast<bf8*>(dy + base + col) = ob;
          if (dr.thresh) {
            const float4 a = drop4(make_float4(o[0], o[1], o[2], o[3]), dr, (unsigned long long)base + col);
            const float4 b = drop4(make_float4(o[4], o[5], o[6], o[7]), dr, (unsigned long long)base + col + 4);
            ob[0] = f2bf(a.x); ob[1] = f2bf(a.y); ob[2] = f2bf(a.z); ob[3] = f2bf(a.w);
            ob[4] = f2bf(b.x); ob[5] = f2bf(b.y); ob[6] = f2bf(b.z); ob[7] = f2bf(b.w);
            *reinterpret_cast<bf8*>(dt + base + col) = ob;
          }
          // the dense-output gradient the GEMMs consume is the bf16-rounded value
#pragma unroll
          for (int e = 0; e < 8; ++e) abias[c][e] += bf2f(ob[e]);
        }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int col = lane * 8 + c * 512;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[0][wv][lane * 8 + e] = ag[c][e]; red[1][wv][lane * 8 + e] = ab[c][e]; red[2][wv][lane * 8 + e] = abias[c][e]; }
    __syncthreads();
    if (wv < 3 && col < H) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int i = lane * 8 + e;
        parts[((size_t)blockIdx.x * 3 + wv) * H + col + e] = (red[wv][0][i] + red[wv][1][i]) + (red[wv][2][i] + red[wv][3][i]);
      }
    }
  }
}

// ---------------------------------------------------------------- column sums of a bf16 matrix
__global__ __launch_bounds__(256) void colsum_kernel(const bf16* x, int ld, float* out, int M, int N, int rows_per_block) {
  __shared__ float red[4][64 * 8];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int col = blockIdx.x * 512 + lane * 8;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (col < N) {
    // 8 rows per trip: eight independent 16-byte loads in flight per lane (one load per trip left the pass latency-bound);
    // rows past the end re-read the last row (valid address) and are AND-masked to zero
    for (int row = r0 + wv; row < r1; row += 32) {
      uint4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int rr = row + 4 * u;
        const uint4 t = *reinterpret_cast<const uint4*>(x + (size_t)min(rr, r1 - 1) * ld + col);
        const unsigned int msk = rr < r1 ? 0xffffffffu : 0u;
        v[u] = make_uint4(t.x & msk, t.y & msk, t.z & msk, t.w & msk);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bf8 b = __builtin_bit_cast(bf8, v[u]);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += bf2f(b[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[wv][lane * 8 + j] = acc[j];
  __syncthreads();
  if (wv == 0 && col < N) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float s = red[0][lane * 8 + j] + red[1][lane * 8 + j] + red[2][lane * 8 + j] + red[3][lane * 8 + j];
      if (col + j < N) atomicAdd(out + col + j, s);
    }
  }
}

// ---------------------------------------------------------------- row L2 normalisation
// out[r, col_off + c] = x[r,c] / max(||x_r||, eps); writes f32 and/or bf16 copies; inv_norm[r] saved.
template <typename TIN>
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const TIN* x, int ldx, float* out_f, bf16* out_b, int ldo, int col_off,
                                                         float* inv_norm, int R, int D, float eps, float post_scale) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wv;
  if (row >= R) return;
  float ss = 0.f;
  for (int col = lane * 4; col < D; col += 256) {
    const float4 v = ld4(x + (size_t)row * ldx + col);
    ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  ss = wave_sum(ss);
  const float inv = 1.0f / fmaxf(sqrtf(ss), eps);
  if (lane == 0 && inv_norm) inv_norm[row] = inv;
  const float sc = inv * post_scale;
  for (int col = lane * 4; col < D; col += 256) {
    const float4 v = ld4(x + (size_t)row * ldx + col);
    const float4 o = make_float4(v.x * sc, v.y * sc, v.z * sc, v.w * sc);
    if (out_f) st4(out_f + (size_t)row * ldo + col_off + col, o);
    if (out_b) st4(out_b + (size_t)row * ldo + col_off + col, o);
  }
}

// dx = (dxh - xh * (xh . dxh)) * inv,  xh = x * inv ;  dxh read from dxh[r, col_off + c] * pre_scale
template <typename TX>
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const TX* x, int ldx, const float* inv_norm, const float* dxh, int ldd,
                                                         int col_off, float* dx, int lddx, int R, int D, float pre_scale, int accumulate) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wv;
  if (row >= R) return;
  const float inv = inv_norm[row];
  float dot = 0.f;
  for (int col = lane * 4; col < D; col += 256) {
    const float4 v = ld4(x + (size_t)row * ldx + col);
    const float4 g = ld4(dxh + (size_t)row * ldd + col_off + col);
    dot += (v.x * g.x + v.y * g.y + v.z * g.z + v.w * g.w);
  }
  dot = wave_sum(dot) * inv * pre_scale;
  for (int col = lane * 4; col < D; col += 256) {
    const float4 v = ld4(x + (size_t)row * ldx + col);
    const float4 g = ld4(dxh + (size_t)row * ldd + col_off + col);
    float4 o = make_float4((g.x * pre_scale - v.x * inv * dot) * inv, (g.y * pre_scale - v.y * inv * dot) * inv,
                           (g.z * pre_scale - v.z * inv * dot) * inv, (g.w * pre_scale - v.w * inv * dot) * inv);
    if (accumulate) {
      const float4 old = ld4(dx + (size_t)row * lddx + col);
      o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
    }
    st4(dx + (size_t)row * lddx + col, o);
  }
}

// ================================================================= C-ABI
#define ROWS_GRID(M) dim3(((M) + 3) / 4)

extern "C" int mmsim_embed_ln_fwd(const long long* ids, const long long* token_types, const long long* position_ids,
                                  const float* word, const float* pos, const float* type, const float* gamma, const float* beta,
                                  void* out, int B, int S, int H, int vocab_size, int type_vocab_size, int max_positions,
                                  int* err_flag, float eps, float dropout_p, unsigned long long seed, unsigned int stream_id,
                                  void* stream) {
  MMSIM_REQUIRE(ids && word && pos && type && gamma && beta && out && err_flag, "embed_ln_fwd: null operand");
  MMSIM_REQUIRE(H % 4 == 0 && H <= 256 * MAXC, "embed_ln_fwd: H must be a multiple of 4 and <= 2048");
  MMSIM_REQUIRE(vocab_size > 0 && type_vocab_size > 0 && max_positions > 0, "embed_ln_fwd: table sizes must be positive");
  MMSIM_REQUIRE(position_ids || S <= max_positions, "embed_ln_fwd: S exceeds the position table");
  const int M = B * S;
  hipLaunchKernelGGL(embed_ln_fwd_kernel, ROWS_GRID(M), dim3(256), 0, (hipStream_t)stream, (const int64_t*)ids,
                     (const int64_t*)token_types, (const int64_t*)position_ids, word, pos, type, gamma, beta, (bf16*)out, M, S, H, eps,
                     make_drop(dropout_p, seed, stream_id), vocab_size, type_vocab_size, max_positions, err_flag);
  return mmsim_check_launch("embed_ln_fwd");
}

// the batch split of embed_ln_bwd over waves: ONE definition, used by the launch and by the scratch-size query
static int embed_bwd_bchunk(int B) { return B >= 64 ? 16 : (B >= 8 ? 4 : 1); }
static int embed_bwd_blocks(int B, int S) { const int c = embed_bwd_bchunk(B); return (S * ((B + c - 1) / c) + 3) / 4; }
extern "C" int mmsim_embed_ln_bwd_scratch_floats(int B, int S, int H) {
  if (B <= 0 || S <= 0 || H <= 0) return 0;
  const unsigned long long n = (unsigned long long)embed_bwd_blocks(B, S) * 4 * H;
  return n > 0x7fffffffull ? 0x7fffffff : (int)n;
}

extern "C" int mmsim_embed_ln_bwd2(const void* dout, const long long* ids, const long long* token_types, const long long* position_ids,
                                   const float* word, const float* pos, const float* type, const float* gamma, float* dword,
                                   float* dpos, float* dtype, float* dgamma, float* dbeta, int B, int S, int H, int vocab_size,
                                   int type_vocab_size, int max_positions, int* err_flag, float eps, float dropout_p,
                                   unsigned long long seed, unsigned int stream_id, float* scratch, unsigned long long scratch_floats,
                                   void* stream);
// the pre-scratch signature, kept for bindings generated from the version-200 header: atomics-only path
extern "C" int mmsim_embed_ln_bwd(const void* dout, const long long* ids, const long long* token_types, const long long* position_ids,
                                  const float* word, const float* pos, const float* type, const float* gamma, float* dword,
                                  float* dpos, float* dtype, float* dgamma, float* dbeta, int B, int S, int H, int vocab_size,
                                  int type_vocab_size, int max_positions, int* err_flag, float eps, float dropout_p,
                                  unsigned long long seed, unsigned int stream_id, void* stream) {
  return mmsim_embed_ln_bwd2(dout, ids, token_types, position_ids, word, pos, type, gamma, dword, dpos, dtype, dgamma, dbeta, B, S, H,
                             vocab_size, type_vocab_size, max_positions, err_flag, eps, dropout_p, seed, stream_id, nullptr, 0, stream);
}
extern "C" int mmsim_embed_ln_bwd2(const void* dout, const long long* ids, const long long* token_types, const long long* position_ids,
                                   const float* word, const float* pos, const float* type, const float* gamma, float* dword,
                                   float* dpos, float* dtype, float* dgamma, float* dbeta, int B, int S, int H, int vocab_size,
                                   int type_vocab_size, int max_positions, int* err_flag, float eps, float dropout_p,
                                   unsigned long long seed, unsigned int stream_id, float* scratch, unsigned long long scratch_floats,
                                   void* stream) {
  MMSIM_REQUIRE(dout && ids && word && dword && dpos && dtype && dgamma && dbeta && err_flag, "embed_ln_bwd: null operand");
  MMSIM_REQUIRE(H % 4 == 0 && H <= 256 * MAXC, "embed_ln_bwd: H must be a multiple of 4 and <= 2048");
  MMSIM_REQUIRE(vocab_size > 0 && type_vocab_size > 0 && type_vocab_size <= 2 && max_positions > 0,
                "embed_ln_bwd: table sizes must be positive (at most two token types)");
  const int serial = mmsim_deterministic();
  const int bchunk = embed_bwd_bchunk(B);
  const int nblk = serial ? 1 : embed_bwd_blocks(B, S);
  // the sums every wave adds to (dgamma, dbeta, the token-type rows) go through per-workgroup slab rows when the caller lends the
  // scratch for them (nblk x 4H floats); without it -- and in the deterministic single-wave mode -- they are atomics as before
  float* parts = (!serial && scratch && scratch_floats >= (unsigned long long)nblk * 4 * H) ? scratch : nullptr;
  hipLaunchKernelGGL(embed_ln_bwd_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const bf16*)dout,
                     (const int64_t*)ids, (const int64_t*)token_types, (const int64_t*)position_ids, word, pos, type, gamma, dword, dpos,
                     dtype, dgamma, dbeta, B, S, H, eps, bchunk, make_drop(dropout_p, seed, stream_id), vocab_size, type_vocab_size,
                     max_positions, err_flag, serial, parts);
  if (parts) {
    RowOuts outs;
    outs.o[0] = dgamma; outs.o[1] = dbeta; outs.o[2] = dtype; outs.o[3] = type_vocab_size > 1 ? dtype + H : nullptr; outs.H = H;
    int gy = nblk / 16; if (gy > 8) gy = 8; if (gy < 1) gy = 1;
    hipLaunchKernelGGL(row_reduce_partials_kernel<true>, dim3((4 * H + 63) / 64, gy), dim3(256), 0, (hipStream_t)stream,
                       parts, nblk, (size_t)4 * H, 4 * H, outs);
  }
  return mmsim_check_launch("embed_ln_bwd");
}

extern "C" int mmsim_add_ln_fwd(const void* t, const void* resid, const float* gamma, const float* beta, void* y, void* h,
                                float* mean, float* rstd, int M, int H, float eps, float dropout_p,
                                unsigned long long seed, unsigned int stream_id, void* stream) {
  MMSIM_REQUIRE(t && resid && gamma && beta && y && h && mean && rstd, "add_ln_fwd: null operand");
  MMSIM_REQUIRE(H % 4 == 0 && H <= 256 * MAXC, "add_ln_fwd: H must be a multiple of 4 and <= 2048");
  if (ADD_LN_FWD8 && (H % 8) == 0 && H <= 1024) {
    const dim3 grid((M + 7) / 8);
    if (H <= 512) hipLaunchKernelGGL((add_ln_fwd8_kernel<1>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)t, (const bf16*)resid,
                                     gamma, beta, (bf16*)y, (bf16*)h, mean, rstd, M, H, eps, make_drop(dropout_p, seed, stream_id));
    else hipLaunchKernelGGL((add_ln_fwd8_kernel<2>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)t, (const bf16*)resid,
                            gamma, beta, (bf16*)y, (bf16*)h, mean, rstd, M, H, eps, make_drop(dropout_p, seed, stream_id));
    return mmsim_check_launch("add_ln_fwd");
  }
  hipLaunchKernelGGL(add_ln_fwd_kernel, ROWS_GRID(M), dim3(256), 0, (hipStream_t)stream, (const bf16*)t, (const bf16*)resid,
                     gamma, beta, (bf16*)y, (bf16*)h, mean, rstd, M, H, eps, make_drop(dropout_p, seed, stream_id));
  return mmsim_check_launch("add_ln_fwd");
}

extern "C" int mmsim_ln_bwd(const void* dh_a, const void* dh_b, const void* y, const float* mean, const float* rstd,
                            const float* gamma, void* dy, void* dt, float* dgamma, float* dbeta, float* dbias, int M, int H,
                            float dropout_p, unsigned long long seed, unsigned int stream_id, float* scratch,
                            unsigned long long scratch_floats, void* stream) {
  MMSIM_REQUIRE(dh_a && y && mean && rstd && gamma && dy && dgamma && dbeta, "ln_bwd: null operand");
  MMSIM_REQUIRE(H % 4 == 0 && H <= 256 * MAXC, "ln_bwd: H must be a multiple of 4 and <= 2048");
  MMSIM_REQUIRE(dropout_p == 0.f || dt, "ln_bwd: dropout needs a separate dt buffer");
  int rpw = (M + 2047) / 2048;   // ~512 blocks of 4 waves
  if (rpw < 1) rpw = 1;
  const int nblk = (M + 4 * rpw - 1) / (4 * rpw);
  MMSIM_REQUIRE(scratch && scratch_floats >= (unsigned long long)nblk * 3 * H, "ln_bwd: scratch too small (need blocks*3*H floats)");
#define LN_BWD_LAUNCH(NCV)                                                                                           \
  hipLaunchKernelGGL((ln_bwd_kernel<NCV>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const bf16*)dh_a, (const bf16*)dh_b, \
                     (const bf16*)y, mean, rstd, gamma, (bf16*)dy, (bf16*)dt, scratch, M, H, rpw,                  \
                     make_drop(dropout_p, seed, stream_id))
  if (H % 8 == 0 && H <= 1024) {
#define LN_BWD8_LAUNCH(NCV, HB)                                                                                      \
  hipLaunchKernelGGL((ln_bwd8_kernel<NCV, HB>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const bf16*)dh_a, (const bf16*)dh_b, \
                     (const bf16*)y, mean, rstd, gamma, (bf16*)dy, (bf16*)dt, scratch, M, H, rpw,                  \
                     make_drop(dropout_p, seed, stream_id))
    if (H <= 512) { if (dh_b) LN_BWD8_LAUNCH(1, true); else LN_BWD8_LAUNCH(1, false); }
    else { if (dh_b) LN_BWD8_LAUNCH(2, true); else LN_BWD8_LAUNCH(2, false); }
#undef LN_BWD8_LAUNCH
  } else {
  const int nc = (H + 255) / 256;
  if (nc <= 1) LN_BWD_LAUNCH(1); else if (nc == 2) LN_BWD_LAUNCH(2); else if (nc == 3) LN_BWD_LAUNCH(3);
  else if (nc == 4) LN_BWD_LAUNCH(4); else LN_BWD_LAUNCH(8);
  }
#undef LN_BWD_LAUNCH
  RowOuts outs;
  outs.o[0] = dgamma; outs.o[1] = dbeta; outs.o[2] = dbias; outs.o[3] = nullptr; outs.H = H;
  int gy = nblk / 16; if (gy > 8) gy = 8; if (gy < 1) gy = 1;
  if (mmsim_deterministic())
    hipLaunchKernelGGL(row_reduce_partials_kernel<false>, dim3((3 * H + 63) / 64, 1), dim3(256), 0, (hipStream_t)stream,
                       scratch, nblk, (size_t)3 * H, 3 * H, outs);
  else
    hipLaunchKernelGGL(row_reduce_partials_kernel<true>, dim3((3 * H + 63) / 64, gy), dim3(256), 0, (hipStream_t)stream,
                       scratch, nblk, (size_t)3 * H, 3 * H, outs);
  return mmsim_check_launch("ln_bwd");
}

extern "C" int mmsim_colsum_bf16(const void* x, int ld, float* out, int M, int N, void* stream) {
  MMSIM_REQUIRE(x && out && M > 0 && N > 0, "colsum: bad arguments");
  MMSIM_REQUIRE(ld % 8 == 0 && ld >= ((N + 7) & ~7), "colsum: ld must be a multiple of 8 covering N rounded up to 8");
  // fewer, longer blocks: every block ends in one atomic per column.  Deterministic mode: ONE row chunk (one adder per column)
  const int rpb = mmsim_deterministic() ? M : (M >= 8192 ? 512 : 256);
  hipLaunchKernelGGL(colsum_kernel, dim3((N + 511) / 512, (M + rpb - 1) / rpb), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)x, ld, out, M, N, rpb);
  return mmsim_check_launch("colsum");
}

extern "C" int mmsim_l2norm_fwd(const void* x, int x_is_bf16, int ldx, float* out_f32, void* out_bf16, int ldo, int col_off,
                                float* inv_norm, int R, int D, float eps, float post_scale, void* stream) {
  MMSIM_REQUIRE(x && (out_f32 || out_bf16) && R > 0 && D > 0, "l2norm_fwd: bad arguments");
  MMSIM_REQUIRE(D % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && col_off % 4 == 0, "l2norm_fwd: D, ld, col_off must be multiples of 4");
  if (x_is_bf16)
    hipLaunchKernelGGL((l2norm_fwd_kernel<bf16>), ROWS_GRID(R), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, ldx, out_f32,
                       (bf16*)out_bf16, ldo, col_off, inv_norm, R, D, eps, post_scale);
  else
    hipLaunchKernelGGL((l2norm_fwd_kernel<float>), ROWS_GRID(R), dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx, out_f32,
                       (bf16*)out_bf16, ldo, col_off, inv_norm, R, D, eps, post_scale);
  return mmsim_check_launch("l2norm_fwd");
}

extern "C" int mmsim_l2norm_bwd(const void* x, int x_is_bf16, int ldx, const float* inv_norm, const float* dxh, int ldd,
                                int col_off, float* dx, int lddx, int R, int D, float pre_scale, int accumulate,
                                void* stream) {
  MMSIM_REQUIRE(x && inv_norm && dxh && dx && R > 0 && D > 0, "l2norm_bwd: bad arguments");
  MMSIM_REQUIRE(D % 4 == 0 && ldx % 4 == 0 && ldd % 4 == 0 && lddx % 4 == 0 && col_off % 4 == 0, "l2norm_bwd: multiples of 4 required");
  if (x_is_bf16)
    hipLaunchKernelGGL((l2norm_bwd_kernel<bf16>), ROWS_GRID(R), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, ldx, inv_norm,
                       dxh, ldd, col_off, dx, lddx, R, D, pre_scale, accumulate);
  else
    hipLaunchKernelGGL((l2norm_bwd_kernel<float>), ROWS_GRID(R), dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx,
                       inv_norm, dxh, ldd, col_off, dx, lddx, R, D, pre_scale, accumulate);
  return mmsim_check_launch("l2norm_bwd");
}
