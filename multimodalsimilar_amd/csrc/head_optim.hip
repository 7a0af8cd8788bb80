// ArcFace head epilogues, fused softmax-cross-entropy, small glue kernels and the fused AdamW step (gfx950).
//   arcface_margin      logits = s * (target ? margin(cos) : cos)                     arcface.py:49-61
//   arcface_ce          loss_b, argmax_b and dcos (bf16) from cos without materialising logits / softmax
//   dlogits_to_dcos     backward of arcface_margin for externally supplied dlogits
//   adamw               torch.optim.AdamW semantics over a flat fp32 buffer, also refreshes the bf16 shadow
#include "common.h"

__global__ __launch_bounds__(256) void arcface_margin_kernel(float* z, int ld, const int64_t* label, int B, int C, Margin m,
                                                             int* err) {
  const int b = blockIdx.y;
  const int64_t y = label[b];
  if (blockIdx.x == 0 && threadIdx.x == 0 && (y < 0 || y >= C)) atomicExch(err, 1);
  for (int c = (blockIdx.x * 256 + threadIdx.x) * 4; c < C; c += gridDim.x * 1024) {
    float* p = z + (size_t)b * ld + c;
    float4 v = *reinterpret_cast<float4*>(p);
    float a[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (c + e == y) a[e] = margin_fwd(a[e], m, nullptr);
      a[e] *= m.s;
    }
    if (c + 3 < C) *reinterpret_cast<float4*>(p) = make_float4(a[0], a[1], a[2], a[3]);
    else
      for (int e = 0; e < 4 && c + e < C; ++e) p[e] = a[e];
  }
}

__device__ __forceinline__ void block_reduce_ms(float& mx, float& sm, int& am, float* sh_f, int* sh_i) {
  // combine (max, sumexp, argmax) over the block
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(mx, o, 64), os = __shfl_xor(sm, o, 64);
    const int oa = __shfl_xor(am, o, 64);
    const float nm = fmaxf(mx, om);
    sm = sm * __expf(mx - nm) + os * __expf(om - nm);
    if (om > mx || (om == mx && oa < am)) am = oa;
    mx = nm;
  }
  if (lane == 0) { sh_f[wv] = mx; sh_f[4 + wv] = sm; sh_i[wv] = am; }
  __syncthreads();
  float M = sh_f[0], S = sh_f[4];
  int A = sh_i[0];
  for (int k = 1; k < 4; ++k) {
    const float om = sh_f[k], os = sh_f[4 + k];
    const int oa = sh_i[k];
    const float nm = fmaxf(M, om);
    S = S * __expf(M - nm) + os * __expf(om - nm);
    if (om > M || (om == M && oa < A)) A = oa;
    M = nm;
  }
  mx = M; sm = S; am = A;
}

// one block per row.  cos: f32 [B, ld].  dcos: bf16 [B, ld] (pad columns C..ld-1 zeroed).
__global__ __launch_bounds__(256) void arcface_ce_kernel(const float* cosm, int ld, const int64_t* label, float* loss,
                                                         long long* argmax, bf16* dcos, int C, Margin m, float gscale,
                                                         int* err) {
  __shared__ float sh_f[8];
  __shared__ int sh_i[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int64_t y = label[b];
  const bool bad = (y < 0 || y >= C);
  if (tid == 0 && bad) atomicExch(err, 1);
  const float* row = cosm + (size_t)b * ld;
  float zt = 0.f, slope = 1.f;
  if (!bad) zt = margin_fwd(row[y], m, &slope) * m.s;
  float mx = -3.0e38f, sm = 0.f;
  int am = 0x7fffffff;
  for (int c = tid * 4; c < C; c += 1024) {
    const float4 v = *reinterpret_cast<const float4*>(row + c);
    const float a[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (c + e < C) {
        const float z = (c + e == y) ? zt : a[e] * m.s;
        if (z > mx) { sm = sm * __expf(mx - z) + 1.0f; mx = z; am = c + e; }
        else sm += __expf(z - mx);
      }
    }
  }
  block_reduce_ms(mx, sm, am, sh_f, sh_i);
  const float lse = mx + __logf(sm);
  if (tid == 0) {
    loss[b] = bad ? 0.f : (lse - zt);
    if (argmax) argmax[b] = am;
  }
  if (!dcos) return;
  bf16* drow = dcos + (size_t)b * ld;
  for (int c = tid * 4; c < ld; c += 1024) {
    const float4 v = *reinterpret_cast<const float4*>(row + c);
    const float a[4] = {v.x, v.y, v.z, v.w};
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (c + e < C) {
        const bool tgt = (c + e == y);
        const float z = tgt ? zt : a[e] * m.s;
        float g = __expf(z - lse) - (tgt ? 1.0f : 0.0f);
        g *= gscale * m.s * (tgt ? slope : 1.0f);
        o[e] = g;
      } else o[e] = 0.f;
    }
    bf4 ob = {f2bf(o[0]), f2bf(o[1]), f2bf(o[2]), f2bf(o[3])};
    *reinterpret_cast<bf4*>(drow + c) = ob;
  }
}

// ---- class-sharded head (data parallelism with the [C, D] weight split over the ranks by class range).
// A rank holds the columns [c0, c0 + C) of the cosine matrix for ALL rows of the global batch.  Phase 1 leaves each row's
// partial softmax statistics over the local columns; the ranks exchange them (a few floats per row), and phase 2 forms dcos of
// the local columns from the row's GLOBAL log-sum-exp.  Labels are global class indices.
//   stats[b] = { local max logit, sum exp(logit - local max), target logit (0 if the target is another rank's), 1 if target here }
//   arg[b]   = global index of the local argmax (ties: lowest index)
__global__ __launch_bounds__(256) void arcface_ce_partial_kernel(const float* cosm, int ld, const int64_t* label, float* stats,
                                                                 long long* arg, int C, long long c0, long long c_total, Margin m,
                                                                 int* err) {
  __shared__ float sh_f[8];
  __shared__ int sh_i[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int64_t yg = label[b];
  if (tid == 0 && (yg < 0 || yg >= c_total)) atomicExch(err, 1);
  const int64_t y = yg - c0;                                   // local column of the target, if it is one of ours
  const bool here = y >= 0 && y < C;
  const float* row = cosm + (size_t)b * ld;
  const float zt = here ? margin_fwd(row[y], m, nullptr) * m.s : 0.f;
  float mx = -3.0e38f, sm = 0.f;
  int am = 0x7fffffff;
  for (int c = tid * 4; c < C; c += 1024) {
    const float4 v = *reinterpret_cast<const float4*>(row + c);
    const float a[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (c + e < C) {
        const float z = (here && c + e == y) ? zt : a[e] * m.s;
        if (z > mx) { sm = sm * __expf(mx - z) + 1.0f; mx = z; am = c + e; }
        else sm += __expf(z - mx);
      }
    }
  }
  block_reduce_ms(mx, sm, am, sh_f, sh_i);
  if (tid == 0) {
    float* o = stats + (size_t)b * 4;
    o[0] = mx; o[1] = sm; o[2] = zt; o[3] = here ? 1.f : 0.f;
    arg[b] = c0 + am;
  }
}

// dcos[b, c] = gscale[b] * s * (target ? slope : 1) * (exp(z - lse[b]) - [target])  for the local columns; bf16 [B, ld], pad zeroed
__global__ __launch_bounds__(256) void arcface_dcos_lse_kernel(const float* cosm, int ld, const int64_t* label, const float* lse,
                                                               const float* gscale, bf16* dcos, int C, long long c0, Margin m) {
  const int b = blockIdx.y;
  const int64_t y = label[b] - c0;
  const bool here = y >= 0 && y < C;
  const float* row = cosm + (size_t)b * ld;
  float zt = 0.f, slope = 1.f;
  if (here) zt = margin_fwd(row[y], m, &slope) * m.s;
  const float L = lse[b], gs = gscale[b] * m.s;
  bf16* drow = dcos + (size_t)b * ld;
  for (int c = (blockIdx.x * 256 + threadIdx.x) * 4; c < ld; c += gridDim.x * 1024) {
    const float4 v = *reinterpret_cast<const float4*>(row + c);
    const float a[4] = {v.x, v.y, v.z, v.w};
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (c + e < C) {
        const bool tgt = here && (c + e == y);
        const float z = tgt ? zt : a[e] * m.s;
        o[e] = (__expf(z - L) - (tgt ? 1.0f : 0.0f)) * gs * (tgt ? slope : 1.0f);
      } else o[e] = 0.f;
    }
    bf4 ob = {f2bf(o[0]), f2bf(o[1]), f2bf(o[2]), f2bf(o[3])};
    *reinterpret_cast<bf4*>(drow + c) = ob;
  }
}

// dcos = dlogits * s * (target ? slope(cos) : 1)   -> bf16 [B, ld] with zeroed pad
__global__ __launch_bounds__(256) void dlogits_to_dcos_kernel(const float* dlogits, int ld_dl, const float* cosm, int ld,
                                                              const int64_t* label, bf16* dcos, int C, Margin m) {
  const int b = blockIdx.y;
  const int64_t y = label ? label[b] : -1;
  for (int c = (blockIdx.x * 256 + threadIdx.x) * 4; c < ld; c += gridDim.x * 1024) {
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (c + e < C) {
        float g = dlogits[(size_t)b * ld_dl + c + e] * m.s;
        if (c + e == y) { float sl; margin_fwd(cosm[(size_t)b * ld + c + e], m, &sl); g *= sl; }
        o[e] = g;
      } else o[e] = 0.f;
    }
    bf4 ob = {f2bf(o[0]), f2bf(o[1]), f2bf(o[2]), f2bf(o[3])};
    *reinterpret_cast<bf4*>(dcos + (size_t)b * ld + c) = ob;
  }
}

// ---------------------------------------------------------------- glue
__global__ void cast_f32_bf16_kernel(const float* x, bf16* y, size_t n) {
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
  for (; i + 3 < n; i += stride) {
    const float4 v = *reinterpret_cast<const float4*>(x + i);
    bf4 o = {f2bf(v.x), f2bf(v.y), f2bf(v.z), f2bf(v.w)};
    *reinterpret_cast<bf4*>(y + i) = o;
  }
  if (i < n) for (size_t k = i; k < n && k < i + 4; ++k) y[k] = f2bf(x[k]);
}
__global__ void cast_f32_f16_kernel(const float* x, f16* y, size_t n) {
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
  for (; i + 3 < n; i += stride) {
    const float4 v = *reinterpret_cast<const float4*>(x + i);
    h4 o = {f2h(v.x), f2h(v.y), f2h(v.z), f2h(v.w)};
    *reinterpret_cast<h4*>(y + i) = o;
  }
  if (i < n) for (size_t k = i; k < n && k < i + 4; ++k) y[k] = f2h(x[k]);
}
__global__ void cast_bf16_f32_kernel(const bf16* x, float* y, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) y[i] = bf2f(x[i]);
}
// rows b*S (the [CLS] token) of h [B*S, H] -> out [B, H]
__global__ void gather_cls_kernel(const bf16* h, bf16* out, int S, int H) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < H; c += blockDim.x) out[(size_t)b * H + c] = h[(size_t)b * S * H + c];
}
// dh [B*S, H] = 0 except row b*S = src[b]
__global__ void scatter_cls_kernel(const bf16* src, bf16* dh, int S, int H) {
  const int row = blockIdx.x, b = row / S;
  const bool cls = (row % S) == 0;
  for (int c = threadIdx.x; c < H; c += blockDim.x) dh[(size_t)row * H + c] = cls ? src[(size_t)b * H + c] : f2bf(0.f);
}
// dpre = dpooled * (1 - pooled^2)
__global__ void tanh_bwd_kernel(const float* dpooled, const float* pooled, bf16* dpre, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const float p = pooled[i]; dpre[i] = f2bf(dpooled[i] * (1.0f - p * p)); }
}

// ---------------------------------------------------------------- fused AdamW over a flat buffer
// p <- p(1 - lr wd); m <- b1 m + (1-b1) g; v <- b2 v + (1-b2) g^2; p <- p - (lr/bc1) m / (sqrt(v)/sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void adamw_kernel(float* p, const float* g, float* m, float* v, bf16* shadow, f16* shadow16, size_t n,
                                                    float lr, float b1, float b2, float eps, float wd, float bc1,
                                                    float rsqrt_bc2, float gscale, const float* hyper) {
  if (hyper) { lr = hyper[0]; bc1 = hyper[1]; rsqrt_bc2 = hyper[2]; }      // step-dependent scalars from device memory (graph replay)
  size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  const size_t stride = (size_t)gridDim.x * 1024;
  for (; i < n; i += stride) {   // n is padded to a multiple of 4 by the host
    const float4 pg = *reinterpret_cast<const float4*>(g + i);
    float4 pp = *reinterpret_cast<const float4*>(p + i), pm = *reinterpret_cast<const float4*>(m + i),
           pv = *reinterpret_cast<const float4*>(v + i);
    float P[4] = {pp.x, pp.y, pp.z, pp.w}, G[4] = {pg.x, pg.y, pg.z, pg.w}, Mm[4] = {pm.x, pm.y, pm.z, pm.w},
          V[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gg = G[e] * gscale;
      P[e] *= (1.0f - lr * wd);
      Mm[e] = b1 * Mm[e] + (1.0f - b1) * gg;
      V[e] = b2 * V[e] + (1.0f - b2) * gg * gg;
      const float denom = sqrtf(V[e]) * rsqrt_bc2 + eps;
      P[e] -= (lr / bc1) * (Mm[e] / denom);
    }
    // every stream here is touched once per step: non-temporal stores keep 19 GB of optimiser state out of the L2 / MALL
    typedef float __attribute__((ext_vector_type(4))) fv4;
    __builtin_nontemporal_store((fv4){P[0], P[1], P[2], P[3]}, reinterpret_cast<fv4*>(p + i));
    __builtin_nontemporal_store((fv4){Mm[0], Mm[1], Mm[2], Mm[3]}, reinterpret_cast<fv4*>(m + i));
    __builtin_nontemporal_store((fv4){V[0], V[1], V[2], V[3]}, reinterpret_cast<fv4*>(v + i));
    if (shadow) {
      bf4 o = {f2bf(P[0]), f2bf(P[1]), f2bf(P[2]), f2bf(P[3])};
      __builtin_nontemporal_store(o, reinterpret_cast<bf4*>(shadow + i));
    }
    if (shadow16) {      // the image tower's forward products read their weights as fp16 (common.h)
      h4 o = {f2h(P[0]), f2h(P[1]), f2h(P[2]), f2h(P[3])};
      __builtin_nontemporal_store(o, reinterpret_cast<h4*>(shadow16 + i));
    }
  }
}

// AdamW for a row-normalised weight matrix (the ArcFace head): one wave per row does the same update as adamw_kernel and,
// with the updated row still in registers, leaves w_hat = w / max(||w||, eps) in bf16 and 1 / max(||w||, eps) for the NEXT
// forward -- F.normalize(self.weight) (arcface.py:47) costs no pass of its own (it was a 1.1 GB read + 0.56 GB write per step at
// 100 000 x 2816), and the bf16 shadow copy the plain kernel writes (unused by the head) is not written.
template <int MAXC>
__global__ __launch_bounds__(256) void adamw_rows_l2norm_kernel(float* p, const float* g, float* m, float* v, bf16* wh, float* inv_norm,
                                                                int R, int D, float lr, float b1, float b2, float eps, float wd, float bc1,
                                                                float rsqrt_bc2, float gscale, float l2eps, const float* hyper) {
  if (hyper) { lr = hyper[0]; bc1 = hyper[1]; rsqrt_bc2 = hyper[2]; }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wv;
  if (row >= R) return;
  typedef float __attribute__((ext_vector_type(4))) fv4;
  float P[MAXC][4];
  float ss = 0.f;
  const size_t base = (size_t)row * D;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int col = lane * 4 + c * 256;
    if (col < D) {
      const size_t i = base + col;
      const float4 pg = *reinterpret_cast<const float4*>(g + i);
      const float4 pp = *reinterpret_cast<const float4*>(p + i), pm = *reinterpret_cast<const float4*>(m + i),
                   pv = *reinterpret_cast<const float4*>(v + i);
      float G[4] = {pg.x, pg.y, pg.z, pg.w}, Mm[4] = {pm.x, pm.y, pm.z, pm.w}, V[4] = {pv.x, pv.y, pv.z, pv.w};
      P[c][0] = pp.x; P[c][1] = pp.y; P[c][2] = pp.z; P[c][3] = pp.w;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float gg = G[e] * gscale;
        P[c][e] *= (1.0f - lr * wd);
        Mm[e] = b1 * Mm[e] + (1.0f - b1) * gg;
        V[e] = b2 * V[e] + (1.0f - b2) * gg * gg;
        const float denom = sqrtf(V[e]) * rsqrt_bc2 + eps;
        P[c][e] -= (lr / bc1) * (Mm[e] / denom);
        ss += P[c][e] * P[c][e];
      }
      __builtin_nontemporal_store((fv4){P[c][0], P[c][1], P[c][2], P[c][3]}, reinterpret_cast<fv4*>(p + i));
      __builtin_nontemporal_store((fv4){Mm[0], Mm[1], Mm[2], Mm[3]}, reinterpret_cast<fv4*>(m + i));
      __builtin_nontemporal_store((fv4){V[0], V[1], V[2], V[3]}, reinterpret_cast<fv4*>(v + i));
    }
  }
  ss = wave_sum(ss);
  const float inv = 1.0f / fmaxf(sqrtf(ss), l2eps);
  if (lane == 0) inv_norm[row] = inv;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int col = lane * 4 + c * 256;
    if (col < D) {
      const bf4 o = {f2bf(P[c][0] * inv), f2bf(P[c][1] * inv), f2bf(P[c][2] * inv), f2bf(P[c][3] * inv)};
      *reinterpret_cast<bf4*>(wh + base + col) = o;          // read by the next forward's cosine GEMM: cached store
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// The fused head (forward_loss): the cosine product's epilogue (EPI_ARCSTATS, gemm_common.h) or -- for shapes the pipelined GEMM
// does not take -- arcface_stats_kernel leaves per (row, column segment) {max, sum exp(z - max), argmax, -} of the scaled margin
// logits z = s (c == y ? margin(cos) : cos)   (arcface.py:49-61).  arcface_combine_kernel folds a row's segments into its
// log-sum-exp, loss, argmax and the target's logit / margin slope; arcface_mean_kernel is CrossEntropyLoss's mean
// (multimodal_classifier_train.py:188).  Backward: arcface_dcos_rowfix_kernel.
__global__ __launch_bounds__(256) void arcface_stats_kernel(const float* __restrict__ cosm, int ld, const int64_t* __restrict__ label,
                                                            float* __restrict__ part, int C, int nseg, Margin m) {
  __shared__ float sh_f[8];
  __shared__ int sh_i[4];
  const int b = blockIdx.y, seg = blockIdx.x, tid = threadIdx.x;
  const int64_t y = label[b];
  const float* row = cosm + (size_t)b * ld;
  const int c = seg * 1024 + tid * 4;
  float mx = -3.0e38f, sm = 0.f;
  int am = 0x7fffffff;
  if (c < C) {
    const float4 v = *reinterpret_cast<const float4*>(row + c);
    const float a[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (c + e < C) {
        const float z = ((int64_t)(c + e) == y ? margin_fwd(a[e], m, nullptr) : a[e]) * m.s;
        if (z > mx) { sm = sm * __expf(mx - z) + 1.0f; mx = z; am = c + e; }
        else sm += __expf(z - mx);
      }
    }
  }
  block_reduce_ms(mx, sm, am, sh_f, sh_i);
  if (tid == 0) {
    float* o = part + ((size_t)b * nseg + seg) * 4;
    o[0] = mx; o[1] = sm; o[2] = __int_as_float(am); o[3] = 0.f;
  }
}

// one block per row: part [B][nseg][4] -> lse, loss, argmax, target logit zt and margin slope (kept for the backward)
__global__ __launch_bounds__(256) void arcface_combine_kernel(const float* __restrict__ part, int nseg, const float* __restrict__ cosm, int ld,
                                                              const int64_t* __restrict__ label, float* __restrict__ rowst, float* __restrict__ loss,
                                                              long long* __restrict__ argmax, int C, Margin m, int* err) {
  __shared__ float sh_f[8];
  __shared__ int sh_i[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int64_t y = label[b];
  const bool bad = (y < 0 || y >= C);
  if (tid == 0 && bad) atomicExch(err, 1);
  float mx = -3.0e38f, sm = 0.f;
  int am = 0x7fffffff;
  const float4* pr = reinterpret_cast<const float4*>(part) + (size_t)b * nseg;
  for (int i = tid; i < nseg; i += 256) {
    const float4 q = pr[i];
    const int qa = __float_as_int(q.z);
    if (q.y > 0.f) {                      // a segment with at least one valid column
      const float nm = fmaxf(mx, q.x);
      sm = sm * __expf(mx - nm) + q.y * __expf(q.x - nm);
      if (q.x > mx || (q.x == mx && qa < am)) am = qa;
      mx = nm;
    }
  }
  block_reduce_ms(mx, sm, am, sh_f, sh_i);
  if (tid == 0) {
    const float lse = mx + __logf(sm);
    float zt = 0.f, slope = 1.f;
    if (!bad) zt = margin_fwd(cosm[(size_t)b * ld + y], m, &slope) * m.s;
    float* o = rowst + (size_t)b * 4;
    o[0] = lse; o[1] = zt; o[2] = slope; o[3] = 0.f;
    loss[b] = bad ? 0.f : (lse - zt);
    if (argmax) argmax[b] = am;
  }
}

__global__ __launch_bounds__(256) void arcface_mean_kernel(const float* __restrict__ loss_b, int B, float* __restrict__ out) {
  __shared__ float red[4];
  float a = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) a += loss_b[i];
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = ((red[0] + red[1]) + (red[2] + red[3])) / (float)B;
}

// Backward of the fused head in ONE pass over the cosines: dcos[b][c] = (dloss / B) s (target ? slope : 1) (exp(z - lse[b]) - [target])
// (bf16 [B, ld], pad columns zero) AND the two row vectors of the weight gradient's row-fix epilogue, rowvec[c] = 1 / ||w_c|| and
// rowvec[C + c] = sum_b dcos[b][c] cos[b][c] (= w_hat_c . dW_hat_c), which used to be a second pass over both [B, C] matrices
// (arcface_rowfix_kernel).  A block owns 256 columns; wave w walks rows w, w + 4, ... with four rows of loads in flight; the
// column sums use the ROUNDED dcos (what the dW product reads) and are added over the four waves in a fixed order.
// dloss_dev: the upstream gradient of the mean loss as a DEVICE scalar (autograd hands it over as a tensor; no host sync).
__global__ __launch_bounds__(256) void arcface_dcos_rowfix_kernel(const float* __restrict__ cosm, int ld, const int64_t* __restrict__ label,
                                                                  const float* __restrict__ rowst, const float* __restrict__ dloss_dev,
                                                                  float gscale, const float* __restrict__ inv_w, bf16* __restrict__ dcos,
                                                                  float* __restrict__ rowvec, int B, int C, Margin m) {
  __shared__ float red[4][256];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = blockIdx.x * 256 + lane * 4;
  const float gs = (dloss_dev ? dloss_dev[0] : 1.0f) * gscale * m.s;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  auto one = [&](int b, const float4& v) __attribute__((always_inline)) {
    const int64_t y = label[b];
    const float4 st = *reinterpret_cast<const float4*>(rowst + (size_t)b * 4);      // lse, zt, slope
    const float a[4] = {v.x, v.y, v.z, v.w};
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool tgt = (int64_t)(c + e) == y;
      const float z = tgt ? st.y : a[e] * m.s;
      const float g = (__expf(z - st.x) - (tgt ? 1.0f : 0.0f)) * gs * (tgt ? st.z : 1.0f);
      o[e] = (c + e < C) ? g : 0.f;
    }
    const bf4 ob = {f2bf(o[0]), f2bf(o[1]), f2bf(o[2]), f2bf(o[3])};
    *reinterpret_cast<bf4*>(dcos + (size_t)b * ld + c) = ob;
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] += bf2f(ob[e]) * a[e];
  };
  int b = wv;
  for (; b + 12 < B; b += 16) {
    const float4 v0 = *reinterpret_cast<const float4*>(cosm + (size_t)b * ld + c);
    const float4 v1 = *reinterpret_cast<const float4*>(cosm + (size_t)(b + 4) * ld + c);
    const float4 v2 = *reinterpret_cast<const float4*>(cosm + (size_t)(b + 8) * ld + c);
    const float4 v3 = *reinterpret_cast<const float4*>(cosm + (size_t)(b + 12) * ld + c);
    one(b, v0); one(b + 4, v1); one(b + 8, v2); one(b + 12, v3);
  }
  for (; b < B; b += 4) one(b, *reinterpret_cast<const float4*>(cosm + (size_t)b * ld + c));
#pragma unroll
  for (int e = 0; e < 4; ++e) red[wv][lane * 4 + e] = acc[e];
  __syncthreads();
  const int cc = blockIdx.x * 256 + threadIdx.x;
  if (cc < C) {
    rowvec[cc] = inv_w[cc];
    rowvec[(size_t)C + cc] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
  }
}

// ================================================================= C-ABI
Margin mk_margin(float s, float m, int easy) {
  Margin r;
  r.s = s; r.cos_m = cosf(m); r.sin_m = sinf(m);
  r.th = cosf(3.14159265358979323846f - m); r.mm = sinf(3.14159265358979323846f - m) * m; r.easy = easy;
  // use double for the constants the reference computes with python floats
  r.cos_m = (float)cos((double)m); r.sin_m = (float)sin((double)m);
  r.th = (float)cos(3.14159265358979323846 - (double)m); r.mm = (float)(sin(3.14159265358979323846 - (double)m) * (double)m);
  return r;
}

extern "C" int mmsim_arcface_margin(float* logits, int ld, const long long* label, int B, int C, float s, float m,
                                    int easy_margin, int* err_flag, void* stream) {
  MMSIM_REQUIRE(logits && label && err_flag && B > 0 && C > 0, "arcface_margin: bad arguments");
  MMSIM_REQUIRE(ld % 4 == 0 && ld >= C, "arcface_margin: ld must be a multiple of 4 and >= C");
  int gx = (C + 1023) / 1024; if (gx > 64) gx = 64;
  hipLaunchKernelGGL(arcface_margin_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, logits, ld,
                     (const int64_t*)label, B, C, mk_margin(s, m, easy_margin), err_flag);
  return mmsim_check_launch("arcface_margin");
}

// launchers used by mmsim_arcface_fwd_fused (gemm.hip: it owns the cosine product)
int arcface_stats_launch(const float* cosm, int ld, const long long* label, float* part, int B, int C, int nseg, Margin m, hipStream_t s) {
  hipLaunchKernelGGL(arcface_stats_kernel, dim3(nseg, B), dim3(256), 0, s, cosm, ld, (const int64_t*)label, part, C, nseg, m);
  return mmsim_check_launch("arcface_stats");
}
int arcface_combine_launch(const float* part, int nseg, const float* cosm, int ld, const long long* label, float* rowst, float* loss_b,
                           long long* argmax, float* loss_mean, int B, int C, Margin m, int* err, hipStream_t s) {
  hipLaunchKernelGGL(arcface_combine_kernel, dim3(B), dim3(256), 0, s, part, nseg, cosm, ld, (const int64_t*)label, rowst, loss_b, argmax, C, m, err);
  if (loss_mean) hipLaunchKernelGGL(arcface_mean_kernel, dim3(1), dim3(256), 0, s, loss_b, B, loss_mean);
  return mmsim_check_launch("arcface_combine");
}

extern "C" int mmsim_arcface_dcos_rowfix(const float* cosm, int ld, const long long* label, const float* rowst, const float* dloss_dev,
                                         float grad_scale, const float* inv_w, void* dcos, float* rowvec, int B, int C, float s, float m,
                                         int easy_margin, void* stream) {
  MMSIM_REQUIRE(cosm && label && rowst && inv_w && dcos && rowvec && B > 0 && C > 0, "arcface_dcos_rowfix: bad arguments");
  MMSIM_REQUIRE(ld % 256 == 0 && ld >= C, "arcface_dcos_rowfix: ld must be a multiple of 256 and >= C (pad the class dimension)");
  hipLaunchKernelGGL(arcface_dcos_rowfix_kernel, dim3(ld / 256), dim3(256), 0, (hipStream_t)stream, cosm, ld, (const int64_t*)label, rowst,
                     dloss_dev, grad_scale, inv_w, (bf16*)dcos, rowvec, B, C, mk_margin(s, m, easy_margin));
  return mmsim_check_launch("arcface_dcos_rowfix");
}

extern "C" int mmsim_arcface_ce(const float* cosm, int ld, const long long* label, float* loss, long long* argmax,
                                void* dcos, int B, int C, float s, float m, int easy_margin, float grad_scale,
                                int* err_flag, void* stream) {
  MMSIM_REQUIRE(cosm && label && loss && err_flag && B > 0 && C > 0, "arcface_ce: bad arguments");
  MMSIM_REQUIRE(ld % 8 == 0 && ld >= C, "arcface_ce: ld must be a multiple of 8 and >= C");
  hipLaunchKernelGGL(arcface_ce_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, cosm, ld, (const int64_t*)label, loss,
                     argmax, (bf16*)dcos, C, mk_margin(s, m, easy_margin), grad_scale, err_flag);
  return mmsim_check_launch("arcface_ce");
}

extern "C" int mmsim_arcface_dlogits_to_dcos(const float* dlogits, int ld_dl, const float* cosm, int ld,
                                             const long long* label, void* dcos, int B, int C, float s, float m,
                                             int easy_margin, void* stream) {
  MMSIM_REQUIRE(dlogits && cosm && dcos && B > 0 && C > 0, "dlogits_to_dcos: bad arguments");
  MMSIM_REQUIRE(ld % 8 == 0 && ld >= C && ld_dl >= C, "dlogits_to_dcos: bad leading dimensions");
  int gx = (ld + 1023) / 1024; if (gx > 64) gx = 64;
  hipLaunchKernelGGL(dlogits_to_dcos_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, dlogits, ld_dl, cosm, ld,
                     (const int64_t*)label, (bf16*)dcos, C, mk_margin(s, m, easy_margin));
  return mmsim_check_launch("dlogits_to_dcos");
}

extern "C" int mmsim_arcface_ce_partial(const float* cosm, int ld, const long long* label, float* stats, long long* arg, int B,
                                        int C_local, long long class_offset, long long C_total, float s, float m, int easy_margin,
                                        int* err_flag, void* stream) {
  MMSIM_REQUIRE(cosm && label && stats && arg && err_flag && B > 0 && C_local > 0, "arcface_ce_partial: bad arguments");
  MMSIM_REQUIRE(ld % 8 == 0 && ld >= C_local, "arcface_ce_partial: ld must be a multiple of 8 and >= C_local");
  MMSIM_REQUIRE(class_offset >= 0 && class_offset + C_local <= C_total, "arcface_ce_partial: shard outside [0, C_total)");
  hipLaunchKernelGGL(arcface_ce_partial_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, cosm, ld, (const int64_t*)label, stats,
                     arg, C_local, class_offset, C_total, mk_margin(s, m, easy_margin), err_flag);
  return mmsim_check_launch("arcface_ce_partial");
}

extern "C" int mmsim_arcface_dcos_from_lse(const float* cosm, int ld, const long long* label, const float* lse,
                                           const float* row_scale, void* dcos, int B, int C_local, long long class_offset, float s,
                                           float m, int easy_margin, void* stream) {
  MMSIM_REQUIRE(cosm && label && lse && row_scale && dcos && B > 0 && C_local > 0, "arcface_dcos_from_lse: bad arguments");
  MMSIM_REQUIRE(ld % 8 == 0 && ld >= C_local, "arcface_dcos_from_lse: ld must be a multiple of 8 and >= C_local");
  int gx = (ld + 1023) / 1024; if (gx > 64) gx = 64;
  hipLaunchKernelGGL(arcface_dcos_lse_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, cosm, ld, (const int64_t*)label, lse,
                     row_scale, (bf16*)dcos, C_local, class_offset, mk_margin(s, m, easy_margin));
  return mmsim_check_launch("arcface_dcos_from_lse");
}

// rowvec[c] = inv_w[c], rowvec[C + c] = sum_b dcos[b][c] * cos[b][c]  ( = w_hat_c . dW_hat_c, because cos = x_hat W_hat^T and
// dW_hat = dcos^T x_hat): the two row vectors of the GEMM's row-fix epilogue, from one pass over the [B, C] matrices.
__global__ __launch_bounds__(256) void arcface_rowfix_kernel(const bf16* __restrict__ dcos, const float* __restrict__ cosm, int ld,
                                                             const float* __restrict__ inv_w, float* __restrict__ rowvec, int B, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int b = 0;
  for (; b + 4 <= B; b += 4) {            // four independent row loads in flight per lane
    const size_t o = (size_t)b * ld + c;
    a0 += bf2f(dcos[o]) * cosm[o];
    a1 += bf2f(dcos[o + ld]) * cosm[o + ld];
    a2 += bf2f(dcos[o + 2 * (size_t)ld]) * cosm[o + 2 * (size_t)ld];
    a3 += bf2f(dcos[o + 3 * (size_t)ld]) * cosm[o + 3 * (size_t)ld];
  }
  for (; b < B; ++b) a0 += bf2f(dcos[(size_t)b * ld + c]) * cosm[(size_t)b * ld + c];
  rowvec[c] = inv_w[c];
  rowvec[(size_t)C + c] = (a0 + a1) + (a2 + a3);
}

extern "C" int mmsim_arcface_rowfix(const void* dcos, const float* cosm, int ld, const float* inv_w, float* rowvec, int B, int C,
                                    void* stream) {
  MMSIM_REQUIRE(dcos && cosm && inv_w && rowvec && B > 0 && C > 0 && ld >= C, "arcface_rowfix: bad arguments");
  hipLaunchKernelGGL(arcface_rowfix_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const bf16*)dcos, cosm, ld,
                     inv_w, rowvec, B, C);
  return mmsim_check_launch("arcface_rowfix");
}

extern "C" int mmsim_adamw_rows_l2norm(float* p, const float* g, float* m, float* v, void* w_hat, float* inv_norm, int R, int D,
                                       float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                                       float l2_eps, const float* dev_hyper, void* stream) {
  MMSIM_REQUIRE(p && g && m && v && w_hat && inv_norm && R > 0 && D > 0, "adamw_rows_l2norm: null operand");
  MMSIM_REQUIRE(D % 4 == 0 && D <= 4096, "adamw_rows_l2norm: D must be a multiple of 4 and <= 4096");
  MMSIM_REQUIRE(step >= 1, "adamw: step is 1-based");
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  const dim3 grid((R + 3) / 4), block(256);
  hipStream_t s = (hipStream_t)stream;
#define ARL(MC) hipLaunchKernelGGL((adamw_rows_l2norm_kernel<MC>), grid, block, 0, s, p, g, m, v, (bf16*)w_hat, inv_norm, R, D, lr, beta1, \
                                   beta2, eps, weight_decay, (float)bc1, (float)(1.0 / sqrt(bc2)), grad_scale, l2_eps, dev_hyper)
  if (D <= 1024) ARL(4); else if (D <= 2048) ARL(8); else if (D <= 3072) ARL(12); else ARL(16);
#undef ARL
  return mmsim_check_launch("adamw_rows_l2norm");
}

static int grid_for(size_t n, int per_block) {
  size_t g = (n + per_block - 1) / per_block;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" int mmsim_cast_f32_to_bf16(const float* x, void* y, unsigned long long n, void* stream) {
  MMSIM_REQUIRE(x && y, "cast: null");
  if (n == 0) return MMSIM_OK;
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, (hipStream_t)stream, x, (bf16*)y, (size_t)n);
  return mmsim_check_launch("cast_f32_to_bf16");
}
extern "C" int mmsim_cast_f32_to_f16(const float* x, void* y, unsigned long long n, void* stream) {
  MMSIM_REQUIRE(x && y, "cast: null");
  if (n == 0) return MMSIM_OK;
  hipLaunchKernelGGL(cast_f32_f16_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, (hipStream_t)stream, x, (f16*)y, (size_t)n);
  return mmsim_check_launch("cast_f32_to_f16");
}
extern "C" int mmsim_cast_bf16_to_f32(const void* x, float* y, unsigned long long n, void* stream) {
  MMSIM_REQUIRE(x && y, "cast: null");
  if (n == 0) return MMSIM_OK;
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, y, (size_t)n);
  return mmsim_check_launch("cast_bf16_to_f32");
}
extern "C" int mmsim_gather_cls(const void* h, void* out, int B, int S, int H, void* stream) {
  MMSIM_REQUIRE(h && out && B > 0, "gather_cls: bad arguments");
  hipLaunchKernelGGL(gather_cls_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, (const bf16*)h, (bf16*)out, S, H);
  return mmsim_check_launch("gather_cls");
}
extern "C" int mmsim_scatter_cls(const void* src, void* dh, int B, int S, int H, void* stream) {
  MMSIM_REQUIRE(src && dh && B > 0, "scatter_cls: bad arguments");
  hipLaunchKernelGGL(scatter_cls_kernel, dim3(B * S), dim3(256), 0, (hipStream_t)stream, (const bf16*)src, (bf16*)dh, S, H);
  return mmsim_check_launch("scatter_cls");
}
extern "C" int mmsim_tanh_bwd(const float* dpooled, const float* pooled, void* dpre, unsigned long long n, void* stream) {
  MMSIM_REQUIRE(dpooled && pooled && dpre, "tanh_bwd: null");
  hipLaunchKernelGGL(tanh_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dpooled, pooled,
                     (bf16*)dpre, (size_t)n);
  return mmsim_check_launch("tanh_bwd");
}

extern "C" int mmsim_adamw_step2(float* p, const float* g, float* m, float* v, void* bf16_shadow, void* f16_shadow, unsigned long long n,
                                 float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                                 float grad_scale, const float* dev_hyper, void* stream);
extern "C" int mmsim_adamw_step(float* p, const float* g, float* m, float* v, void* bf16_shadow, unsigned long long n,
                                float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                                float grad_scale, const float* dev_hyper, void* stream) {
  return mmsim_adamw_step2(p, g, m, v, bf16_shadow, nullptr, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, dev_hyper, stream);
}
extern "C" int mmsim_adamw_step2(float* p, const float* g, float* m, float* v, void* bf16_shadow, void* f16_shadow, unsigned long long n,
                                 float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                                 float grad_scale, const float* dev_hyper, void* stream) {
  MMSIM_REQUIRE(p && g && m && v, "adamw: null operand");
  MMSIM_REQUIRE(n % 4 == 0, "adamw: flat buffer length must be a multiple of 4");
  MMSIM_REQUIRE(step >= 1, "adamw: step is 1-based");
  if (n == 0) return MMSIM_OK;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16*)bf16_shadow, (f16*)f16_shadow,
                     (size_t)n, lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)(1.0 / sqrt(bc2)), grad_scale, dev_hyper);
  return mmsim_check_launch("adamw");
}
