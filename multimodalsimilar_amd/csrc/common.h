// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the two-tower + ArcFace step.
// Wave = 64 lanes everywhere; no other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(16))) float f16v;
// fp16: the storage type of the IMAGE tower's forward tensors (conv outputs, activations, its 1x1-conv weight shadow).  Same
// bytes and MFMA rate as bf16 with an 11-bit significand instead of 8: bf16 storage alone moved the image embedding by 3.5-5 %
// (north_star: 1e-2), fp16 by 0.6 % (oracle/effnet_ref.py emulate="fp16").  Gradients stay bf16 (range: per-element
// gradients of a 112x112 map sit below fp16's normal range).
typedef _Float16 f16;
// 16-byte chunk as a REGISTER vector: what prefetching row / strip loops hold their in-flight data in (a uint4 is a struct and cannot be
// an inline-asm register operand).  PIPE_FIRST_USE ties the first use of a prefetched group of N chunk pairs to the top of the code
// that consumes it: without it the scheduler lifts the group's conversions above the previous group's arithmetic and the vmcnt
// waits for its loads go with them ("Where hipcc puts s_waitcnt vmcnt").
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
#define PIPE_FIRST_USE_N(a, b, N) _Pragma("unroll") for (int q_ = 0; q_ < (N); ++q_) asm volatile("" : "+v"((a)[q_]), "+v"((b)[q_]))
#define PIPE_FIRST_USE(a, b) PIPE_FIRST_USE_N(a, b, 4)
#define PIPE_FIRST_USE1_N(a, N) _Pragma("unroll") for (int q_ = 0; q_ < (N); ++q_) asm volatile("" : "+v"((a)[q_]))
__device__ __forceinline__ uint4 as_u4(u4v v) { return __builtin_bit_cast(uint4, v); }
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(4))) _Float16 h4;
typedef __attribute__((ext_vector_type(2))) _Float16 h2;
typedef __attribute__((ext_vector_type(4))) short s4;
typedef __attribute__((ext_vector_type(8))) short s8;

#define MMSIM_OK 0
#define MMSIM_ERR_ARG 1
#define MMSIM_ERR_LAUNCH 2

extern "C" void mmsim_set_error(const char* msg);
int mmsim_check_launch(const char* what);
int mmsim_current_device(void);
const unsigned long long* mmsim_step_seed_ptr(void);      // core.hip: device word added to every dropout seed (graph replay), or NULL
int mmsim_deterministic(void);      // core.hip: fixed-order reductions everywhere (verification mode)

#define MMSIM_REQUIRE(cond, msg)                 \
  do {                                           \
    if (!(cond)) {                               \
      mmsim_set_error(msg);                      \
      return MMSIM_ERR_ARG;                      \
    }                                            \
  } while (0)

// Division of an index (< 2^31) by a launch constant: one v_mul_hi + shift instead of the ~35-instruction integer
// division sequence (the per-item  it % nstrip, it / nstrip % H, ...  decompositions cost as much as the arithmetic).
struct FastDiv { unsigned int mul, shr, d; };
static inline FastDiv make_fastdiv(unsigned int d) {
  FastDiv f; f.d = d; f.mul = 0; f.shr = 0;
  if (d > 1) {
    unsigned int l = 0;
    while ((1ull << l) < d) ++l;
    const unsigned int pw = 31 + l;
    f.mul = (unsigned int)(((1ull << pw) + d - 1) / d);
    f.shr = pw - 32;
  }
  return f;
}
__device__ __forceinline__ unsigned int fdiv(unsigned int n, const FastDiv& f) { return f.d == 1 ? n : (__umulhi(n, f.mul) >> f.shr); }
__device__ __forceinline__ void fdivmod(unsigned int n, const FastDiv& f, int& q, int& r) { q = (int)fdiv(n, f); r = (int)(n - (unsigned int)q * f.d); }

__device__ __forceinline__ float bf2f(bf16 x) { return (float)x; }
__device__ __forceinline__ bf16 f2bf(float x) { return (bf16)x; }
__device__ __forceinline__ float h2f(f16 x) { return (float)x; }
// round-to-nearest-even, saturating at +-65504 (a conv output beyond fp16's range must not become inf -> NaN downstream)
__device__ __forceinline__ f16 f2h(float x) { return (f16)__builtin_amdgcn_fmed3f(x, -65504.0f, 65504.0f); }

// Lane exchanges inside a 16-lane row as DPP moves (VALU rate) instead of ds_bpermute round trips through the LDS crossbar (~120 cycles
// each, and dependent in a reduction tree): quad_perm [1,0,3,2] and [2,3,0,1] for the partners 1 and 2 apart; after those every lane
// of a quad holds the quad's total, so the MIRRORED lane of the 8-lane half-row (row_half_mirror) / of the 16-lane row (row_mirror)
// supplies the other quad's / the other half's total.  EXEC must be full.  Partners 16 and 32 apart still go through __shfl_xor.
template <int CTRL> __device__ __forceinline__ float dpp_mov_f(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float sum8_dpp(float v) {       // every lane of an aligned 8-lane group ends with the group's sum
  v += dpp_mov_f<0xB1>(v); v += dpp_mov_f<0x4E>(v); v += dpp_mov_f<0x141>(v);
  return v;
}
__device__ __forceinline__ float sum16_dpp(float v) {      // the same for an aligned 16-lane row
  v = sum8_dpp(v); v += dpp_mov_f<0x140>(v);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  v = sum16_dpp(v);
  v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
  return v;
}
// two / four independent sums: the two remaining LDS-crossbar steps of each ride together (one latency chain, not two / four)
__device__ __forceinline__ void wave_sum2(float& a, float& b) {
  a = sum16_dpp(a); b = sum16_dpp(b);
  a += __shfl_xor(a, 16, 64); b += __shfl_xor(b, 16, 64);
  a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64);
}
__device__ __forceinline__ void wave_sum4(float& a, float& b, float& c, float& d) {
  a = sum16_dpp(a); b = sum16_dpp(b); c = sum16_dpp(c); d = sum16_dpp(d);
  a += __shfl_xor(a, 16, 64); b += __shfl_xor(b, 16, 64); c += __shfl_xor(c, 16, 64); d += __shfl_xor(d, 16, 64);
  a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64); c += __shfl_xor(c, 32, 64); d += __shfl_xor(d, 32, 64);
}
// The maximum must not see the zero a bound_ctrl DPP move supplies for a lane that EXEC disables (all-negative rows: masked softmax
// scores): update_dpp with old = v and bound_ctrl off makes a disabled / invalid source return the lane's OWN value, the identity
// of max.  (The sums keep bound_ctrl = zero, their identity.)
template <int CTRL> __device__ __forceinline__ float dpp_self_f(float x) {
  const int xi = __builtin_bit_cast(int, x);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(xi, xi, CTRL, 0xF, 0xF, false));
}
// maximum / integer minimum over an aligned 16-lane row: every lane ends with the row's value (EXEC full)
__device__ __forceinline__ float max16_dpp(float v) {
  v = fmaxf(v, dpp_self_f<0xB1>(v)); v = fmaxf(v, dpp_self_f<0x4E>(v)); v = fmaxf(v, dpp_self_f<0x141>(v)); v = fmaxf(v, dpp_self_f<0x140>(v));
  return v;
}
template <int CTRL> __device__ __forceinline__ int dpp_self_i(int x) { return __builtin_amdgcn_update_dpp(x, x, CTRL, 0xF, 0xF, false); }
__device__ __forceinline__ int min16_dpp_i(int v) {
  v = min(v, dpp_self_i<0xB1>(v)); v = min(v, dpp_self_i<0x4E>(v)); v = min(v, dpp_self_i<0x141>(v)); v = min(v, dpp_self_i<0x140>(v));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_self_f<0xB1>(v)); v = fmaxf(v, dpp_self_f<0x4E>(v)); v = fmaxf(v, dpp_self_f<0x141>(v)); v = fmaxf(v, dpp_self_f<0x140>(v));
  v = fmaxf(v, __shfl_xor(v, 16, 64)); v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}

// ArcFace additive angular margin (arcface.py:49-61), shared by the head kernels (head_optim.hip) and the cosine product's
// statistics epilogue (gemm_common.h, EPI_ARCSTATS).
struct Margin { float s, cos_m, sin_m, th, mm; int easy; };
// Deliberate, documented deviation (DESIGN.md "Numerics"): the reference computes sqrt(1 - cos^2) without a guard
// (arcface.py:49), which is NaN when |cos| > 1 and has an infinite gradient at |cos| = 1.  With bf16 unit vectors the
// cosine of a sample aligned with its class row can round to 1 + 2^-8, where the fp32 reference is finite: the radicand
// is floored at 1e-12 here, which changes nothing wherever the reference itself is finite to fp32 precision.
__device__ __forceinline__ float margin_fwd(float c, const Margin& m, float* slope) {
  const float sine = sqrtf(fmaxf(1.0f - c * c, 1e-12f));
  const float phi = c * m.cos_m - sine * m.sin_m;
  const bool take = m.easy ? (c > 0.f) : ((c - m.th) > 0.f);
  if (slope) *slope = take ? (m.cos_m + m.sin_m * c / sine) : 1.0f;
  return take ? phi : (m.easy ? c : c - m.mm);
}

Margin mk_margin(float s, float m, int easy);      // head_optim.hip: the constants the reference computes with python floats
int arcface_stats_launch(const float* cosm, int ld, const long long* label, float* part, int B, int C, int nseg, Margin m, hipStream_t s);
int arcface_combine_launch(const float* part, int nseg, const float* cosm, int ld, const long long* label, float* rowst, float* loss_b,
                           long long* argmax, float* loss_mean, int B, int C, Margin m, int* err, hipStream_t s);

// exact-erf GELU and its derivative (HF "gelu", modeling_bert.py:334-337)
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
  const float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
// v_rcp_f32 (1 ulp) instead of an IEEE division: `1.0f / y` expands to the ~10-instruction div_scale / fma / div_fixup
// sequence, which made every SiLU-carrying kernel (and the GEMM operand transform) VALU-bound
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_f(x); }
__device__ __forceinline__ float silu_grad_f(float x) {
  const float s = sigmoid_f(x);
  return s * (1.0f + x * (1.0f - s));
}

// Counter-based dropout RNG, deterministic and recomputed in backward.  One 32-bit hash serves TWO consecutive
// elements (16 bits each): keep element idx iff bits16(hash(idx >> 1), idx & 1) >= p * 2^16.  The (seed, stream) part
// of the key is loop-invariant, so an element costs about four integer instructions (it used to be two full hashes per
// element, which made the attention kernels VALU-bound on the mask alone).
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint64_t step_seed(uint64_t seed, const unsigned long long* dev) { return dev ? seed + *dev : seed; }
__device__ __forceinline__ uint32_t drop_key(uint64_t seed, uint32_t stream) {
  return hash32((uint32_t)seed ^ hash32(stream * 0x85EBCA6BU + (uint32_t)(seed >> 32) * 0x9E3779B9U + 0x632BE5ABU));
}
__device__ __forceinline__ uint32_t drop_bits(uint32_t key, uint64_t pair_idx) {     // 2 x 16 random bits for elements 2*pair_idx, +1
  return hash32((uint32_t)pair_idx ^ key ^ ((uint32_t)(pair_idx >> 32) * 0x9E3779B9U));
}
__device__ __forceinline__ bool drop_keep16(uint32_t bits, int odd, uint32_t thresh) {
  return ((odd ? bits >> 16 : bits & 0xffffU) >= (thresh >> 16));
}
__device__ __forceinline__ bool drop_keep(uint64_t seed, uint32_t stream, uint64_t idx, uint32_t thresh) {
  return drop_keep16(drop_bits(drop_key(seed, stream), idx >> 1), (int)(idx & 1), thresh);
}

// ---- transposing LDS reads (ds_read_b64_tr_b16, cdna_hip_programming.md T10): MFMA operand fragments whose reduction
// index is the LDS image's ROW index.  EXEC must be all ones at every call.
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

