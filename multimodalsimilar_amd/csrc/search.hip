// Exhaustive inner-product top-k search on the MI355X: the step after the model in the reference's inference jobs
// (nlp_infer.py:139-152, daodian_infer.py:225-230 / 295-302: faiss.normalize_L2 + IndexFlat(METRIC_INNER_PRODUCT).search).
//   split_bf16_cat   fp32 rows -> bf16 [hi | lo | hi] (queries) or [hi | hi | lo] (database): ONE bf16 MFMA GEMM over the
//                    tripled K then yields hi.hi + lo.hi + hi.lo = the fp32 inner product to ~2^-16 relative (the
//                    lo.lo term is dropped), so rankings match an fp32 scan except for genuine near-ties;
//   topk_merge       folds a [nq, n] chunk of scores into the running per-row top-k lists (descending score, equal
//                    scores by ascending index); one wave per row, k selection rounds of a wave-wide arg-max.
#include "common.h"

__global__ __launch_bounds__(256) void split_bf16_cat_kernel(const float* __restrict__ x, int ldx, bf16* __restrict__ out, int ldo,
                                                             int R, int D, int db_side, float post_scale_rows) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)R * D) return;
  const int r = (int)(i / D), c = (int)(i - (size_t)r * D);
  const float v = x[(size_t)r * ldx + c] * post_scale_rows;
  const bf16 hi = f2bf(v);
  const bf16 lo = f2bf(v - bf2f(hi));
  bf16* o = out + (size_t)r * ldo;
  o[c] = hi;
  o[D + c] = db_side ? hi : lo;
  o[2 * D + c] = db_side ? lo : hi;
}

// candidate order: higher score first, equal scores by lower index
__device__ __forceinline__ bool better(float v, long long i, float bv, long long bi) { return v > bv || (v == bv && i < bi); }

template <int KMAX>
__global__ __launch_bounds__(256) void topk_merge_kernel(const float* __restrict__ scores, int ld, int nq, int n, long long col_offset,
                                                         int k, float* best_val, long long* best_idx, int first) {
  __shared__ float nv[4][KMAX];
  __shared__ long long ni[4][KMAX];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + w;
  if (row >= nq) return;                       // whole wave leaves together
  const float* s = scores + (size_t)row * ld;
  float* bv = best_val + (size_t)row * k;
  long long* bi = best_idx + (size_t)row * k;
  // the old list entry this lane contributes (lanes >= k: none)
  float ov = -INFINITY; long long oi = 0x7fffffffffffffffll;
  if (!first && lane < k) { ov = bv[lane]; oi = bi[lane]; }
  float pv = INFINITY; long long pi = -1;      // previously selected (value, index): later picks must be strictly worse
  for (int r = 0; r < k; ++r) {
    float cv = -INFINITY; long long ci = 0x7fffffffffffffffll;
    for (int j = lane; j < n; j += 64) {
      const float v = s[j];
      const long long gi = col_offset + j;
      if (better(pv, pi, v, gi) && better(v, gi, cv, ci)) { cv = v; ci = gi; }
    }
    if (oi != 0x7fffffffffffffffll && better(pv, pi, ov, oi) && better(ov, oi, cv, ci)) { cv = ov; ci = oi; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float v2 = __shfl_xor(cv, o, 64);
      const long long i2 = __shfl_xor(ci, o, 64);
      if (better(v2, i2, cv, ci)) { cv = v2; ci = i2; }
    }
    if (lane == 0) { nv[w][r] = cv; ni[w][r] = ci; }
    pv = cv; pi = ci;
  }
  __builtin_amdgcn_s_waitcnt(0);               // lane 0's LDS writes are visible to the wave (same wave: program order) 
  if (lane < k) { bv[lane] = nv[w][lane]; bi[lane] = (ni[w][lane] == 0x7fffffffffffffffll) ? -1 : ni[w][lane]; }
}

extern "C" int mmsim_split_bf16_cat(const float* x, int ldx, void* out, int ldo, int R, int D, int db_side, float scale, void* stream) {
  MMSIM_REQUIRE(x && out && R > 0 && D > 0 && ldx >= D && ldo >= 3 * D, "split_bf16_cat: bad arguments");
  const size_t n = (size_t)R * D;
  hipLaunchKernelGGL(split_bf16_cat_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, (bf16*)out, ldo, R,
                     D, db_side, scale);
  return mmsim_check_launch("split_bf16_cat");
}

extern "C" int mmsim_topk_merge(const float* scores, int ld, int nq, int n, long long col_offset, int k, float* best_val,
                                long long* best_idx, int first, void* stream) {
  MMSIM_REQUIRE(scores && best_val && best_idx && nq > 0 && n > 0 && ld >= n, "topk_merge: bad arguments");
  MMSIM_REQUIRE(k >= 1 && k <= 64, "topk_merge: 1 <= k <= 64 (one list entry per lane)");
  hipLaunchKernelGGL((topk_merge_kernel<64>), dim3((nq + 3) / 4), dim3(256), 0, (hipStream_t)stream, scores, ld, nq, n, col_offset, k,
                     best_val, best_idx, first);
  return mmsim_check_launch("topk_merge");
}
