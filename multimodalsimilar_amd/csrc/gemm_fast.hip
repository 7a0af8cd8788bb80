// Fast path of the bf16 MFMA GEMM for tile-aligned shapes (the text tower): 256x128x64 tiles, 512 threads = 8 waves
// (4 along M x 2 along N, 64x64 each), operands brought HBM -> LDS by LDS-DMA (global_load_lds, 16 B per lane, no
// VGPR staging), a 3-slot LDS ring with TWO K-tiles in flight behind a counted s_waitcnt vmcnt and ONE raw s_barrier
// per K-step (cdna_hip_programming.md section 5, "Pipelining across barriers" / "3-buf span").
//
// LDS-DMA writes lane-linearly (wave-uniform base + lane*16), so the bank-conflict swizzle is applied to the SOURCE
// address and undone on the read (guide rule 21):
//   k-major operand   [rows][64 k]  (128-B rows): 16-B chunk c of row r is stored at chunk c ^ ((r >> 1) & 7)
//                                                  -> ds_read_b128 fragments are conflict-free;
//   transposed operand [64 k][cols] (256/512-B rows): 32-B column block cb of k-row k at block cb ^ key(k)
//                                                  -> ds_read_b64_tr_b16 fragments are conflict-free.
// Per step t:  wait(tile t landed: vmcnt(6) leaves tile t+1 in flight) -> s_barrier -> issue tile t+2 into the
// slot read in step t-1 (every wave has passed the barrier, so nobody still reads it) -> 32 MFMAs per wave on tile t.
#include "gemm_common.h"
#include <stdlib.h>

#define FBM 256
#define FBN 128
#define FBK 64
#define FA_STAGE (FBM * 128)                 // 32 KiB either layout
#define FB_STAGE (FBN * 128)                 // 16 KiB either layout
#define F_STAGE (FA_STAGE + FB_STAGE)        // 48 KiB; x3 slots = 144 KiB of the CU's 160 KiB

typedef s4 __attribute__((address_space(3))) * lds_s4_ptr;
typedef void __attribute__((address_space(3))) * lds_void_ptr;
typedef const void __attribute__((address_space(1))) * glb_void_ptr;

__device__ __forceinline__ int ftr_key(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

// Transposing LDS read of one 16x32 MFMA operand fragment (two ds_read_b64_tr_b16, rows +0 and +4) as INLINE ASM.
// With the __builtin form hipcc (ROCm 7.2) cannot prove the read does not alias an in-flight LDS-DMA and inserts
// s_waitcnt vmcnt(0) in front of every transposed read, which serialises the whole operand pipeline (measured: DMA,
// LDS reads and MFMAs became purely additive in the NN / TN kernels).  The asm form is invisible to that pass; its
// completion is covered by the explicit s_waitcnt lgkmcnt(0) each K-step executes before the data is consumed.
// Plain 16-byte fragment read, also as inline asm: in the software-pipelined kernel the fragments read in step t are
// consumed in step t+1; a compiler-tracked load would make hipcc wait lgkmcnt(0) at the first MFMA of the next
// iteration -- AFTER the following slice's reads were issued -- and expose the LDS latency again.
__device__ __forceinline__ bf8 ds_read_b128_asm(const char* lds_ptr) {
  const uint32_t a = (uint32_t)(uintptr_t)(lds_s4_ptr)(lds_ptr);
  bf8 r;
  asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(a));
  return r;
}

template <int OFF_HI>
__device__ __forceinline__ bf8 tr_read_pair_asm(const char* lds_ptr) {
  const uint32_t a = (uint32_t)(uintptr_t)(lds_s4_ptr)(lds_ptr);
  s4 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a), "i"(OFF_HI));
  return __builtin_bit_cast(bf8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// Issue the LDS-DMA of one operand tile.  NI = wave-instructions per wave (ROWS*128 bytes / 8 waves / 1 KiB).
// TRANS = false: tile [ROWS][64 k] from X[(r0+row)*ld + k0 + ...];  TRANS = true: tile [64 k][ROWS] from X[(k0+k)*ld + r0 + ...]
template <bool TRANS, int ROWS>
__device__ __forceinline__ void dma_tile(const bf16* __restrict__ X, int ld, int r0, int k0, char* lds, int wave, int lane) {
  constexpr int NI = ROWS * 128 / (8 * 1024);
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int blk = wave * NI + i;                       // 1-KiB block of the tile image
    const bf16* src;
    if (!TRANS) {
      const int row = blk * 8 + (lane >> 3), pos = lane & 7;
      const int c = pos ^ ((row >> 1) & 7);
      src = X + (size_t)(r0 + row) * ld + k0 + c * 8;
    } else {
      constexpr int CPR = ROWS / 8;                      // 16-B chunks per k-row (16 or 32)
      constexpr int RPB = 64 / CPR;                      // k-rows per 1-KiB block (4 or 2)
      const int k = blk * RPB + lane / CPR, pc = lane % CPR;
      const int c = (((pc >> 1) ^ ftr_key(k)) << 1) | (pc & 1);
      src = X + (size_t)(k0 + k) * ld + r0 + c * 8;
    }
    __builtin_amdgcn_global_load_lds((glb_void_ptr)src, (lds_void_ptr)(lds + blk * 1024), 16, 0, 0);
  }
}

template <bool TRANS, int ROWS>
__device__ __forceinline__ bf8 ffrag(const char* lds, int rbase, int ks, int lane) {
  if (!TRANS) {
    const int row = rbase + (lane & 15), kc = ks * 4 + (lane >> 4);
    return *reinterpret_cast<const bf8*>(lds + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4));
  } else {
    constexpr int ROWB = ROWS * 2;
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int k = ks * 32 + 8 * g + q;
    const int cb = rbase >> 4;
    const int off = k * ROWB + ((cb ^ ftr_key(k)) << 5) + p * 8;
    return tr_read_pair_asm<4 * ROWB>(lds + off);
  }
}

template <bool TA, bool TB_KMAJOR>
__global__ __launch_bounds__(512, 2) void gemm_fast_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // 1-D grid over (split, tile); blocks that share operand panels (same split, neighbouring tiles) are made
  // consecutive and each XCD gets one contiguous chunk of them, so its private L2 serves the re-reads
  const int ntiles = p.tiles_m * p.tiles_n;
  const int nwg = ntiles * p.splits;
  const int bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const int split = wg / ntiles, tile = wg - split * ntiles;
  const int tm = tile / p.tiles_n, tn = tile % p.tiles_n;
  const int m0 = tm * FBM, n0 = tn * FBN;
  const int kbeg = split * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int nk = (kend - kbeg) / FBK;

  f4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};

  // prologue: tiles 0 and 1 in flight
  dma_tile<TA, FBM>(p.A, p.lda, m0, kbeg, smem, wave, lane);
  dma_tile<!TB_KMAJOR, FBN>(p.B, p.ldb, n0, kbeg, smem + FA_STAGE, wave, lane);
  if (nk > 1) {
    dma_tile<TA, FBM>(p.A, p.lda, m0, kbeg + FBK, smem + F_STAGE, wave, lane);
    dma_tile<!TB_KMAJOR, FBN>(p.B, p.ldb, n0, kbeg + FBK, smem + F_STAGE + FA_STAGE, wave, lane);
  }
  int slot = 0;
  for (int t = 0; t < nk; ++t) {
    // each wave issues 6 LDS-DMA instructions per tile: vmcnt(6) = "tile t landed, tile t+1 may still fly"
    if (t + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (t + 2 < nk) {
      int ns = slot + 2; if (ns >= 3) ns -= 3;
      dma_tile<TA, FBM>(p.A, p.lda, m0, kbeg + (t + 2) * FBK, smem + ns * F_STAGE, wave, lane);
      dma_tile<!TB_KMAJOR, FBN>(p.B, p.ldb, n0, kbeg + (t + 2) * FBK, smem + ns * F_STAGE + FA_STAGE, wave, lane);
    }
    const char* la = smem + slot * F_STAGE;
    const char* lb = la + FA_STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = ffrag<TA, FBM>(la, wm * 64 + i * 16, ks, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = ffrag<!TB_KMAJOR, FBN>(lb, wn * 64 + j * 16, ks, lane);
      if (TA || !TB_KMAJOR) {          // transposed fragments come from inline asm: wait for them explicitly (rule 18)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
    slot = slot + 1; if (slot >= 3) slot = 0;
  }
  const int row0 = m0 + wm * 64, col0 = n0 + wn * 64;
  const bool fs = split == 0;
  __syncthreads();                                  // every wave is done reading the operand ring: reuse it for staging
  float* stg = reinterpret_cast<float*>(smem) + wave * (64 * EP_PITCH);
  if (!p.c_f32) fast_epilogue_epi<0>(p, acc, row0, col0, lane, fs, stg);      // bf16 outputs carry the fused epilogues
  else if (p.atomic) fast_epilogue<EPI_NONE, 3>(p, acc, row0, col0, lane, fs, stg);
  else if (p.accum) fast_epilogue<EPI_NONE, 2>(p, acc, row0, col0, lane, fs, stg);
  else if (p.epi == EPI_TANH) fast_epilogue<EPI_TANH, 1>(p, acc, row0, col0, lane, fs, stg);
  else fast_epilogue<EPI_NONE, 1>(p, acc, row0, col0, lane, fs, stg);
}

// ---------------------------------------------------------------------------------------------------------
// 256x256x32 variant: twice the arithmetic intensity of the 256x128 tile (128 flop per staged byte, the L2->LDS
// stream is what bounds the smaller tile), 8 waves as 2 (M) x 4 (N) with 128x64 per wave (acc[8][4]), a 4-slot ring
// of 32-KiB K-slices with THREE slices in flight, one raw s_barrier per K-step of 32 MFMAs per wave.
// k-major rows are 64 B here: 16-B chunk c of row r is stored at c ^ G(r), G(r) = (-(r >> 2)) & 3 (conflict-free
// for the ds_read_b128 lane groups); transposed operands use the same 32-B column-block swizzle as above.
#define GBM 256
#define GBN 256
#define GBK 32
#define G_SLOTS 4

template <bool TRANS, int ROWS>      // ROWS = 256 or 128 rows (k-major) / columns (transposed) of the operand tile
__device__ __forceinline__ void dma_tile32(const bf16* __restrict__ X, int ld, int r0, int k0, char* lds, int wave, int lane) {
  constexpr int NI = ROWS / 128;        // 1-KiB wave-instructions per wave: the tile is ROWS*64 bytes
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int blk = wave * NI + i;
    const bf16* src;
    if (!TRANS) {
      const int row = blk * 16 + (lane >> 2), pos = lane & 3;
      const int c = pos ^ ((-(row >> 2)) & 3);
      src = X + (size_t)(r0 + row) * ld + k0 + c * 8;
    } else {
      constexpr int CPR = ROWS / 8;      // 16-B chunks per k-row (32 or 16); a 1-KiB block holds 64 / CPR k-rows
      const int k = blk * (64 / CPR) + lane / CPR, pc = lane % CPR;
      const int c = (((pc >> 1) ^ ftr_key(k)) << 1) | (pc & 1);
      src = X + (size_t)(k0 + k) * ld + r0 + c * 8;
    }
    __builtin_amdgcn_global_load_lds((glb_void_ptr)src, (lds_void_ptr)(lds + blk * 1024), 16, 0, 0);
  }
}

template <bool TRANS, int ROWS>
__device__ __forceinline__ bf8 gfrag(const char* lds, int rbase, int lane) {
  if (!TRANS) {
    const int row = rbase + (lane & 15), kc = lane >> 4;
    return ds_read_b128_asm(lds + row * 64 + ((kc ^ ((-(row >> 2)) & 3)) << 4));
  } else {
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int k = 8 * g + q;
    const int off = k * (ROWS * 2) + (((rbase >> 4) ^ ftr_key(k)) << 5) + p * 8;
    return tr_read_pair_asm<4 * ROWS * 2>(lds + off);
  }
}

template <bool TA, bool TB_KMAJOR, int BN>     // BN = 256: 2 x 4 waves of 128 x 64;  BN = 128: 4 x 2 waves of 64 x 64
__global__ __launch_bounds__(512, 2) void gemm_fast256_kernel(GemmParams p) {
  constexpr int MT = BN == 256 ? 8 : 4;                  // 16-row MFMA tiles per wave along M
  constexpr int WROWS = MT * 16;
  constexpr int A_BYTES = GBM * 64, B_BYTES = BN * 64, STAGE = A_BYTES + B_BYTES;
  constexpr int LPS = 2 + BN / 128;                      // LDS-DMA instructions per wave per K-slice
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = BN == 256 ? (wave >> 2) : (wave >> 1), wn = BN == 256 ? (wave & 3) : (wave & 1);

  const int ntiles = p.tiles_m * p.tiles_n;
  const int nwg = ntiles * p.splits;
  const int bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const int split = wg / ntiles, tile = wg - split * ntiles;
  const int tm = tile / p.tiles_n, tn = tile % p.tiles_n;
  const int m0 = tm * GBM, n0 = tn * BN;
  const int kbeg = split * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int nk = (kend - kbeg) / GBK;

  f4 acc[MT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};

  // Software pipeline (per K-slice t):  slice t lives in REGISTERS (fragment set A), slices t+1..t+3 are in the LDS
  // ring or in flight, and the step reads slice t+1's fragments into set B while the MFMAs of slice t run, so
  // neither the LDS-read latency nor the DMA latency sits on the MFMA critical path.  Because a slice's slot is
  // free as soon as its fragments are in registers, the 4-slot ring keeps FOUR slices ahead of the MFMAs.
#pragma unroll
  for (int t = 0; t < 4; ++t)
    if (t < nk && !(p.dbg & 1)) {
      dma_tile32<TA, GBM>(p.A, p.lda, m0, kbeg + t * GBK, smem + t * STAGE, wave, lane);
      dma_tile32<!TB_KMAJOR, BN>(p.B, p.ldb, n0, kbeg + t * GBK, smem + t * STAGE + A_BYTES, wave, lane);
    }
  bf8 afA[MT], bfA[4], afB[MT], bfB[4];
  {
    const int ahead = min(nk - 1, 3);                    // slices issued after slice 0
    if (ahead >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LPS) : "memory");
    else if (ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS) : "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < 4; ++j) bfA[j] = gfrag<!TB_KMAJOR, BN>(smem + A_BYTES, wn * 64 + j * 16, lane);
#pragma unroll
    for (int i = 0; i < MT; ++i) afA[i] = gfrag<TA, GBM>(smem, wm * WROWS + i * 16, lane);
  }
  // one pipeline step: consume (af0, bf0) = slice t, fill (af1, bf1) with slice t+1, refill the ring with slice t+4
#define GEMM256_STEP(af0, bf0, af1, bf1, T)                                                                         \
  {                                                                                                                 \
    const int t_ = (T);                                                                                             \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   /* slice t's fragments (asm + compiler LDS reads) landed */ \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
    if (t_ + 1 < nk) {                                                                                              \
      const int ahead = min(nk - 2 - t_, 2);           /* slices issued after slice t+1 */                        \
      if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS) : "memory");                                \
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");                               \
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                         \
      __builtin_amdgcn_s_barrier();                                                                                 \
      __builtin_amdgcn_sched_barrier(0);                                                                            \
      if (t_ + 4 < nk && !(p.dbg & 1)) {                                                                            \
        char* ns = smem + (t_ & 3) * STAGE;            /* slot of slice t: its fragments are in registers */        \
        dma_tile32<TA, GBM>(p.A, p.lda, m0, kbeg + (t_ + 4) * GBK, ns, wave, lane);                                 \
        dma_tile32<!TB_KMAJOR, BN>(p.B, p.ldb, n0, kbeg + (t_ + 4) * GBK, ns + A_BYTES, wave, lane);                \
      }                                                                                                             \
      const char* la = smem + ((t_ + 1) & 3) * STAGE;                                                               \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) bf1[j] = gfrag<!TB_KMAJOR, BN>(la + A_BYTES, wn * 64 + j * 16, lane); \
      _Pragma("unroll") for (int i = 0; i < MT; ++i) af1[i] = gfrag<TA, GBM>(la, wm * WROWS + i * 16, lane);        \
    }                                                                                                               \
    if (!(p.dbg & 4)) {                                                                                             \
      _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                                \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                               \
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf0[j], af0[i], acc[i][j], 0, 0, 0);                  \
    }                                                                                                               \
  }
  for (int t = 0; t < nk; t += 2) {
    GEMM256_STEP(afA, bfA, afB, bfB, t)
    if (t + 1 < nk) GEMM256_STEP(afB, bfB, afA, bfA, t + 1)
  }
#undef GEMM256_STEP
  const bool fs = split == 0;
  float* stg = reinterpret_cast<float*>(smem) + wave * (64 * EP_PITCH);
#pragma unroll
  for (int half = 0; half < MT / 4; ++half) {
    __syncthreads();          // ring (first pass) / previous staging pass fully consumed
    f4 (&a4)[4][4] = *reinterpret_cast<f4 (*)[4][4]>(&acc[half * 4][0]);
    const int row0 = m0 + wm * WROWS + half * 64, col0 = n0 + wn * 64;
    if (!p.c_f32) fast_epilogue_epi<0>(p, a4, row0, col0, lane, fs, stg);
    else if (p.atomic) fast_epilogue<EPI_NONE, 3>(p, a4, row0, col0, lane, fs, stg);
    else if (p.accum) fast_epilogue<EPI_NONE, 2>(p, a4, row0, col0, lane, fs, stg);
    else if (p.epi == EPI_TANH) fast_epilogue<EPI_TANH, 1>(p, a4, row0, col0, lane, fs, stg);
    else fast_epilogue<EPI_NONE, 1>(p, a4, row0, col0, lane, fs, stg);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Persistent form of the pipelined kernel: one workgroup per CU walks its share of the (split, tile) list, and the
// LDS-DMA prologue of tile k+1 (four K-slices) is issued BEFORE the epilogue of tile k, so the epilogue's stores and
// the next tile's first loads overlap instead of each tile paying prologue latency + epilogue tail back to back
// (at K = 1024 those were ~40 % of a tile's time).  The epilogue is staged through a small per-wave LDS region that
// sits BESIDE the operand ring (16 rows x 64 fp32 per wave, XOR-swizzled 16-B slots: conflict-free both ways), one
// 16-row MFMA tile row at a time, and leaves as whole 128-B / 256-B row segments.
template <int EPI, int CMODE>
__device__ __forceinline__ void chunk_epilogue(const GemmParams& p, const f4 (&a)[4], int row0, int col0, int lane,
                                               bool first_split, float* stg) {
  const int r = lane & 15, g = lane >> 4;
#pragma unroll
  for (int j = 0; j < 4; ++j) *reinterpret_cast<f4*>(stg + r * 64 + (((j * 4 + g) ^ r) << 2)) = a[j];
  if (CMODE == 3) {
#pragma unroll 4
    for (int rr = 0; rr < 16; ++rr)
      atomicAdd(reinterpret_cast<float*>(p.C) + (size_t)(row0 + rr) * p.ldc + col0 + lane,
                stg[rr * 64 + ((((lane >> 2) ^ rr) << 2) | (lane & 3))] * p.alpha);
    return;
  }
  const int c4 = lane & 15, n = col0 + c4 * 4;
  const bool add_bias = (p.bias != nullptr) && first_split;
  const float4 bias = add_bias ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int rr = it * 4 + g;
    const size_t m = (size_t)(row0 + rr);
    const f4 x = *reinterpret_cast<const f4*>(stg + rr * 64 + ((c4 ^ rr) << 2));
    float v[4] = {x[0] * p.alpha + bias.x, x[1] * p.alpha + bias.y, x[2] * p.alpha + bias.z, x[3] * p.alpha + bias.w};
    if (EPI == EPI_GELU) {
      bf4 pre = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
      *reinterpret_cast<bf4*>(p.aux_out + m * p.ld_aux + n) = pre;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = gelu_f(bf2f(pre[e]));
    } else if (EPI == EPI_MUL_GELU_GRAD) {
      const bf4 y = *reinterpret_cast<const bf4*>(p.aux_in + m * p.ld_aux + n);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= gelu_grad_f(bf2f(y[e]));
    } else if (EPI == EPI_ADD) {
      const bf4 y = *reinterpret_cast<const bf4*>(p.aux_in + m * p.ld_aux + n);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += bf2f(y[e]);
    } else if (EPI == EPI_TANH) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = tanhf(v[e]);
    }
    if (CMODE == 0) {
      bf4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
      *reinterpret_cast<bf4*>(reinterpret_cast<bf16*>(p.C) + m * p.ldc + n) = o;
    } else {
      float* c = reinterpret_cast<float*>(p.C) + m * p.ldc + n;
      if (CMODE == 1) {
        *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        const float4 old = *reinterpret_cast<const float4*>(c);
        *reinterpret_cast<float4*>(c) = make_float4(v[0] + old.x, v[1] + old.y, v[2] + old.z, v[3] + old.w);
      }
    }
  }
}

template <int MT, int EPI, int CMODE>
__device__ __forceinline__ void tile_epilogue(const GemmParams& p, f4 (&acc)[MT][4], int row0, int col0, int lane, bool fs,
                                              float* stg) {
#pragma unroll
  for (int i = 0; i < MT; ++i) chunk_epilogue<EPI, CMODE>(p, acc[i], row0 + i * 16, col0, lane, fs, stg);
}

template <bool TA, bool TB_KMAJOR, int BN>
__global__ __launch_bounds__(512, 2) void gemm_pers_kernel(GemmParams p) {
  constexpr int MT = BN == 256 ? 8 : 4;
  constexpr int WROWS = MT * 16;
  constexpr int A_BYTES = GBM * 64, B_BYTES = BN * 64, STAGE = A_BYTES + B_BYTES;
  constexpr int LPS = 2 + BN / 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = BN == 256 ? (wave >> 2) : (wave >> 1), wn = BN == 256 ? (wave & 3) : (wave & 1);
  float* stg = reinterpret_cast<float*>(smem + G_SLOTS * STAGE) + wave * (16 * 64);

  // this block's share of the (split, tile) list: XCD x owns one contiguous chunk, its blocks interleave inside it
  const int ntiles = p.tiles_m * p.tiles_n;
  const int nwg = ntiles * p.splits;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, bpx = gridDim.x >> 3;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int cstart = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int cend = cstart + q8 + (xcd < r8 ? 1 : 0);
  int w = cstart + idx;
  if (w >= cend) return;

  int split, m0, n0, kbeg, nk;
  auto decode = [&](int wg, int& sp, int& mm, int& nn, int& kb, int& nks) {
    sp = wg / ntiles;
    const int tile = wg - sp * ntiles;
    mm = (tile / p.tiles_n) * GBM; nn = (tile % p.tiles_n) * BN;
    kb = sp * p.k_per_split;
    nks = (min(p.K, kb + p.k_per_split) - kb) / GBK;
  };
  auto prologue = [&](int mm, int nn, int kb, int nks) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (t < nks) {
        dma_tile32<TA, GBM>(p.A, p.lda, mm, kb + t * GBK, smem + t * STAGE, wave, lane);
        dma_tile32<!TB_KMAJOR, BN>(p.B, p.ldb, nn, kb + t * GBK, smem + t * STAGE + A_BYTES, wave, lane);
      }
  };
  decode(w, split, m0, n0, kbeg, nk);
  prologue(m0, n0, kbeg, nk);

  while (true) {
    f4 acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
    bf8 afA[MT], bfA[4], afB[MT], bfB[4];
    {
      const int ahead = min(nk - 1, 3);
      if (ahead >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LPS) : "memory");
      else if (ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS) : "memory");
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfA[j] = gfrag<!TB_KMAJOR, BN>(smem + A_BYTES, wn * 64 + j * 16, lane);
#pragma unroll
      for (int i = 0; i < MT; ++i) afA[i] = gfrag<TA, GBM>(smem, wm * WROWS + i * 16, lane);
    }
#define GEMMP_STEP(af0, bf0, af1, bf1, T)                                                                           \
  {                                                                                                                 \
    const int t_ = (T);                                                                                             \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
    if (t_ + 1 < nk) {                                                                                              \
      const int ahead = min(nk - 2 - t_, 2);                                                                        \
      if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS) : "memory");                                \
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");                               \
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                         \
      __builtin_amdgcn_s_barrier();                                                                                 \
      __builtin_amdgcn_sched_barrier(0);                                                                            \
      if (t_ + 4 < nk) {                                                                                            \
        char* ns = smem + (t_ & 3) * STAGE;                                                                         \
        dma_tile32<TA, GBM>(p.A, p.lda, m0, kbeg + (t_ + 4) * GBK, ns, wave, lane);                                 \
        dma_tile32<!TB_KMAJOR, BN>(p.B, p.ldb, n0, kbeg + (t_ + 4) * GBK, ns + A_BYTES, wave, lane);                \
      }                                                                                                             \
      const char* la = smem + ((t_ + 1) & 3) * STAGE;                                                               \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) bf1[j] = gfrag<!TB_KMAJOR, BN>(la + A_BYTES, wn * 64 + j * 16, lane); \
      _Pragma("unroll") for (int i = 0; i < MT; ++i) af1[i] = gfrag<TA, GBM>(la, wm * WROWS + i * 16, lane);        \
    }                                                                                                               \
    _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                                  \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                 \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf0[j], af0[i], acc[i][j], 0, 0, 0);                    \
  }
    for (int t = 0; t < nk; t += 2) {
      GEMMP_STEP(afA, bfA, afB, bfB, t)
      if (t + 1 < nk) GEMMP_STEP(afB, bfB, afA, bfA, t + 1)
    }
#undef GEMMP_STEP
    // every wave has read its last fragments (lgkmcnt(0) above); after this barrier the ring may be refilled
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const int wn_ = w + bpx;
    const bool has_next = wn_ < cend;
    int split2 = 0, m2 = 0, n2 = 0, kb2 = 0, nk2 = 0;
    if (has_next) {
      decode(wn_, split2, m2, n2, kb2, nk2);
      prologue(m2, n2, kb2, nk2);                 // in flight while the epilogue below runs
    }
    const bool fs = split == 0;
    const int row0 = m0 + wm * WROWS, col0 = n0 + wn * 64;
    if (!p.c_f32) {
      switch (p.epi) {
        case EPI_GELU: tile_epilogue<MT, EPI_GELU, 0>(p, acc, row0, col0, lane, fs, stg); break;
        case EPI_MUL_GELU_GRAD: tile_epilogue<MT, EPI_MUL_GELU_GRAD, 0>(p, acc, row0, col0, lane, fs, stg); break;
        case EPI_ADD: tile_epilogue<MT, EPI_ADD, 0>(p, acc, row0, col0, lane, fs, stg); break;
        case EPI_TANH: tile_epilogue<MT, EPI_TANH, 0>(p, acc, row0, col0, lane, fs, stg); break;
        default: tile_epilogue<MT, EPI_NONE, 0>(p, acc, row0, col0, lane, fs, stg); break;
      }
    } else if (p.atomic) tile_epilogue<MT, EPI_NONE, 3>(p, acc, row0, col0, lane, fs, stg);
    else if (p.accum) tile_epilogue<MT, EPI_NONE, 2>(p, acc, row0, col0, lane, fs, stg);
    else if (p.epi == EPI_TANH) tile_epilogue<MT, EPI_TANH, 1>(p, acc, row0, col0, lane, fs, stg);
    else tile_epilogue<MT, EPI_NONE, 1>(p, acc, row0, col0, lane, fs, stg);
    if (!has_next) break;
    w = wn_; split = split2; m0 = m2; n0 = n2; kbeg = kb2; nk = nk2;
  }
}

static bool pipe_eligible(const GemmParams& p, int splits, int bn) {
  if (p.M % GBM || p.N % bn || p.K % GBK || p.k_per_split % GBK) return false;
  return (p.M / GBM) * (p.N / bn) * splits >= 128;        // enough blocks to fill the chip
}
static bool fast128_eligible(const GemmParams& p, int splits) {
  if (p.M % FBM || p.N % FBN) return false;
  if (p.K % FBK || p.k_per_split % FBK) return false;
  return true;
}

bool gemm_fast_eligible(const GemmParams& p, int splits) {
  if (p.c_f32 && !(p.epi == EPI_NONE || p.epi == EPI_TANH)) return false;   // f32 outputs: plain / tanh only
  if ((p.ldc % 4) || (p.ld_aux % 4)) return false;
  if ((long long)p.k_per_split * splits < p.K) return false;
  return pipe_eligible(p, splits, 256) || pipe_eligible(p, splits, 128) || fast128_eligible(p, splits);
}

static int tile_pref() {      // MMSIM_GEMM_TILE: 0 auto (default), 1 = old 256x128x64 kernel only, 2 = pipelined BN=128 only
  static int v = -1;
  if (v < 0) { const char* e = getenv("MMSIM_GEMM_TILE"); v = e ? atoi(e) : 0; }
  return v;
}

// MMSIM_GEMM_PERSIST=1 selects the persistent form.  Measured (round 1): no gain on the forward layout (the epilogue
// is executed by the same waves, so only the ~2 us prologue latency per tile is hidden) and a loss on the layouts with
// transposed operands, whose instantiations spill inside the K loop at 256 VGPRs; it stays opt-in until the address
// arithmetic is moved to immediates / SGPRs (DESIGN.md "open items").
static int persist_pref() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("MMSIM_GEMM_PERSIST"); v = e ? atoi(e) : 0; }
  return v;
}
static int num_cus() {
  static int n = 0;
  if (!n) { hipDeviceProp_t pr; int dev = 0; (void)hipGetDevice(&dev); n = (hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 256; }
  return n;
}

template <int BN>
static void launch_pers(GemmParams p, int trans_a, int b_kmajor, int splits, hipStream_t s) {
  p.tiles_m = p.M / GBM; p.tiles_n = p.N / BN; p.splits = splits;
  const int nwg = p.tiles_m * p.tiles_n * splits;
  int nblk = (num_cus() / 8) * 8;                      // one workgroup per CU, a multiple of the 8 XCDs
  if (nblk > ((nwg + 7) / 8) * 8) nblk = ((nwg + 7) / 8) * 8;
  dim3 grid(nblk), block(512);
  const size_t lds = G_SLOTS * (GBM * 64 + BN * 64) + 8 * 16 * 64 * 4;      // operand ring + per-wave epilogue staging
  static bool done = false;
  if (!done) {
    (void)hipFuncSetAttribute((const void*)gemm_pers_kernel<false, true, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_pers_kernel<false, false, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_pers_kernel<true, false, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_pers_kernel<true, true, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    done = true;
  }
  if (!trans_a && b_kmajor) hipLaunchKernelGGL((gemm_pers_kernel<false, true, BN>), grid, block, lds, s, p);
  else if (!trans_a && !b_kmajor) hipLaunchKernelGGL((gemm_pers_kernel<false, false, BN>), grid, block, lds, s, p);
  else if (trans_a && !b_kmajor) hipLaunchKernelGGL((gemm_pers_kernel<true, false, BN>), grid, block, lds, s, p);
  else hipLaunchKernelGGL((gemm_pers_kernel<true, true, BN>), grid, block, lds, s, p);
}

template <int BN>
static void launch_pipe(GemmParams p, int trans_a, int b_kmajor, int splits, hipStream_t s) {
  if (persist_pref()) { launch_pers<BN>(p, trans_a, b_kmajor, splits, s); return; }
  p.tiles_m = p.M / GBM; p.tiles_n = p.N / BN; p.splits = splits;
  dim3 grid(p.tiles_m * p.tiles_n * splits), block(512);
  const size_t lds = 8 * 64 * EP_PITCH * 4;      // 136 KiB: epilogue staging (8 waves x 64 x 68 floats) >= the operand ring
  static bool done = false;
  if (!done) {
    (void)hipFuncSetAttribute((const void*)gemm_fast256_kernel<false, true, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_fast256_kernel<false, false, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_fast256_kernel<true, false, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_fast256_kernel<true, true, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    done = true;
  }
  if (!trans_a && b_kmajor) hipLaunchKernelGGL((gemm_fast256_kernel<false, true, BN>), grid, block, lds, s, p);
  else if (!trans_a && !b_kmajor) hipLaunchKernelGGL((gemm_fast256_kernel<false, false, BN>), grid, block, lds, s, p);
  else if (trans_a && !b_kmajor) hipLaunchKernelGGL((gemm_fast256_kernel<true, false, BN>), grid, block, lds, s, p);
  else hipLaunchKernelGGL((gemm_fast256_kernel<true, true, BN>), grid, block, lds, s, p);
}

void gemm_fast_launch(GemmParams p, int trans_a, int b_kmajor, int splits, hipStream_t s) {
  const int pref = tile_pref();
  // wgrad-shaped products (A transposed: few output tiles, long reduction, split-K) take the narrower tile: more tiles
  // per split and half the atomic traffic per block; everything else prefers 256x256 (twice the flop per staged byte)
  if (pref != 1) {
    const bool e256 = pipe_eligible(p, splits, 256), e128 = pipe_eligible(p, splits, 128);
    if (!trans_a) {
      if (e256 && pref != 2) { launch_pipe<256>(p, trans_a, b_kmajor, splits, s); return; }
      if (e128) { launch_pipe<128>(p, trans_a, b_kmajor, splits, s); return; }
    } else {
      // wgrad-shaped products (A transposed, long reduction, split-K with atomics): measured on the text shapes, the
      // 256x256 pipelined kernel wins when the output has >= 32 such tiles (FFN, QKV: 960-1000 / 770 TFLOP/s), the
      // 256x128x64 kernel below wins on small outputs (attention-output projection: 720 vs 590)
      const int t256 = (p.M / GBM) * (p.N / 256);
      if ((e256 && t256 >= 32 && t256 * splits >= 160) || pref == 3) { launch_pipe<256>(p, trans_a, b_kmajor, splits, s); return; }
      if (!fast128_eligible(p, splits)) {
        if (e128) { launch_pipe<128>(p, trans_a, b_kmajor, splits, s); return; }
        if (e256) { launch_pipe<256>(p, trans_a, b_kmajor, splits, s); return; }
      }
    }
  }
  p.tiles_m = p.M / FBM; p.tiles_n = p.N / FBN;
  p.splits = splits;
  dim3 grid(p.tiles_m * p.tiles_n * splits), block(512);
  const size_t lds = 3 * F_STAGE;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)gemm_fast_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_fast_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_fast_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_fast_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  if (!trans_a && b_kmajor) hipLaunchKernelGGL((gemm_fast_kernel<false, true>), grid, block, lds, s, p);
  else if (!trans_a && !b_kmajor) hipLaunchKernelGGL((gemm_fast_kernel<false, false>), grid, block, lds, s, p);
  else if (trans_a && !b_kmajor) hipLaunchKernelGGL((gemm_fast_kernel<true, false>), grid, block, lds, s, p);
  else hipLaunchKernelGGL((gemm_fast_kernel<true, true>), grid, block, lds, s, p);
}
