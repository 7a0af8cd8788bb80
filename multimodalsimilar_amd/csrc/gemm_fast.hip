// Fast paths of the bf16 MFMA GEMM for tile-aligned shapes (the text tower and the ArcFace head), operands brought
// HBM -> LDS by LDS-DMA (global_load_lds, 16 B per lane, no VGPR staging):
//   gemm_pp64_kernel  256 x 256 (or 256 x 128) tiles, 64-deep slices, two wave groups in anti-phase -- the production
//                     kernel (described in front of it);
//   gemm_fast_kernel  256 x 128 x 64 tiles, 3-slot ring, ONE raw s_barrier per K-step, two K-tiles in flight behind a counted
//                     s_waitcnt vmcnt (cdna_hip_programming.md section 5, "3-buf span"): kept for the wgrad-shaped products
//                     with few output tiles (attention-output projection), where it measures faster.
//
// LDS-DMA writes lane-linearly (wave-uniform base + lane*16), so the bank-conflict swizzle is applied to the SOURCE
// address and undone on the read (guide rule 21):
//   k-major operand   [rows][64 k]  (128-B rows): 16-B chunk c of row r is stored at chunk c ^ ((r >> 1) & 7)
//                                                  -> ds_read_b128 fragments are conflict-free;
//   transposed operand [64 k][cols] (256/512-B rows): 32-B column block cb of k-row k at block cb ^ key(k)
//                                                  -> ds_read_b64_tr_b16 fragments are conflict-free.
// gemm_fast_kernel, per step t:  wait(tile t landed: vmcnt(6) leaves tile t+1 in flight) -> s_barrier -> issue tile t+2
// into the slot read in step t-1 (every wave has passed the barrier) -> 32 MFMAs per wave on tile t.
#include "gemm_common.h"
#include <string.h>
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

// one 1-KiB piece (piece i of the wave's NI) of the same tile: lets the caller place the pieces between other instructions
template <bool TRANS, int ROWS>
__device__ __forceinline__ void dma_piece(const bf16* __restrict__ X, int ld, int r0, int k0, char* lds, int wave, int lane, int i) {
  constexpr int NI = ROWS * 128 / (8 * 1024);
  const int blk = wave * NI + i;
  const bf16* src;
  if (!TRANS) {
    const int row = blk * 8 + (lane >> 3), pos = lane & 7;
    const int c = pos ^ ((row >> 1) & 7);
    src = X + (size_t)(r0 + row) * ld + k0 + c * 8;
  } else {
    constexpr int CPR = ROWS / 8;
    constexpr int RPB = 64 / CPR;
    const int k = blk * RPB + lane / CPR, pc = lane % CPR;
    const int c = (((pc >> 1) ^ ftr_key(k)) << 1) | (pc & 1);
    src = X + (size_t)(k0 + k) * ld + r0 + c * 8;
  }
  __builtin_amdgcn_global_load_lds((glb_void_ptr)src, (lds_void_ptr)(lds + blk * 1024), 16, 0, 0);
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

#define GBM 256
#ifndef PP64_A_IN_MFMA
#define PP64_A_IN_MFMA 1          // gemm_pp64_kernel: A_{u+2}'s LDS-DMA pieces issued between the odd step's MFMAs (see the kernel)
#endif
#define PP64_B_IN_MFMA 0          // B_{u+1} cannot move the same way: it would be issued and awaited within one step
// PP64_B_SPLIT: the second half of B_{u+1}'s LDS-DMA pieces is issued right behind the even step's first MFMAs (after MFMA rows 0
// and 1) instead of in its load phase, which then carries two pieces instead of four; they still have the rest of that MFMA phase and
// the odd step's load phase (~650 cycles) before the wait for slice u + 1.  Measured r03 (tools/bench_gemm.py, alternating builds on
// one box): weight-gradient layout +2-3 % (qkv 902-923 -> 937-943, ffn1 1 070 -> 1 090-1 110, o 695-706 -> 704-720 TFLOP/s), forward
// layout -2-3 %, data-gradient layout +-2 %: on for the weight-gradient layout (A transposed, B k-rows) only.
#ifndef PP64_B_SPLIT
#define PP64_B_SPLIT 1
#endif
// PP64_BAL (round 4): the BALANCED schedule.  The B fragments of a whole 64-deep slice are read in the even step (8 instead of 4
// fragment registers sets), which frees B's ring slot half a slice earlier, so that B_{u+2} -- not B_{u+1} -- is what a slice
// issues and EVERY LDS-DMA piece has more than a slice of time to land.  The eight pieces of a slice are then spread evenly over
// the four phases: two per load phase (where a piece costs ~130 issue cycles but runs under the partner wave's MFMA phase) and
// two per MFMA phase (~40 cycles each, exposed), instead of 4 + 0 + 0 + 4: with the measured costs the four intervals of a slice
// were 770 | 770 | 512 | 670 cycles (max of the two groups' phases) and become ~590 each.
//   even step 2u:   B1 | read B (both k-halves) and A k-half 0 of slice u; issue A_{u+2} pieces 0, 1;                 lgkmcnt(0) | B2 | MFMAs + A_{u+2} pieces 2, 3
//   odd  step 2u+1: B1 | read A k-half 1 of slice u; issue B_{u+2} first half; vmcnt(|A| + |B| / 2);                  lgkmcnt(0) | B2 | MFMAs + B_{u+2} second half
// WAR: A_{u+2} -> slot of A_{u-1}, last read in the lagging group's odd load phase of slice u - 1, which ends at the barrier that
// opens the leading group's even load phase of slice u; B_{u+2} -> slot of B_u, last read in the lagging group's even load phase
// of slice u, which ends at the barrier that opens the leading group's odd load phase.  RAW: slice u + 1 (A_{u+1} from the even
// phases, B_{u+1} from the odd phases of slice u - 1; A_1 | B_1 from the prologue) is waited for at the end of the odd LOAD phase of
// slice u, where exactly A_{u+2} and the first half of B_{u+2} are younger.
// Measured (r04, tools/bench_gemm.py, alternating builds on one box, M = 32 768): data-gradient layout +3.6 / +1.5 / +6 / -0.5 % on the
// qkv / o / ffn1 / ffn2 shapes (1 074 vs 1 041, 1 102 vs 1 086, 1 251-1 259 vs 1 173-1 197, 1 024-1 034 vs 1 025-1 058 TFLOP/s); forward layout
// -2 ... 0 % (qkv 1 055 vs 1 064-1 091); weight-gradient layout (which already splits its B pieces, PP64_B_SPLIT) +-3 %, 0 on average.  So the
// issue-cycle model above explains only part of a slice: the load path (64 KiB of LDS-DMA writes + 192 KiB of fragment reads per slice
// through one LDS) is within 20 % of the matrix path however its instructions are placed.  On for the data-gradient layout only
// (PP64_BAL_LAYOUT); 2 = every layout, 0 = off.
#ifndef PP64_BAL
#define PP64_BAL 1
#endif
#define PP64_BAL_LAYOUT(TA, TBK) (PP64_BAL == 2 || (PP64_BAL == 1 && !(TA) && !(TBK)))
#ifndef PP64_B_KEEP
#define PP64_B_KEEP 2             // pieces (of four at BN = 256) that stay in the load phase; the others follow MFMA rows 0, 1, ...
                                  // (1 measured the same as 2, 0 about 1 % below them on the weight-gradient products)
#endif

// Main-loop ablation switches (no DMA / no MFMA / no epilogue) exist ONLY in the separately compiled ablation object
// (-DMMSIM_ABLATE, tools/bench_gemm_abl.py builds it next to the product library): the product binary has no such code path.
#ifdef MMSIM_ABLATE
#define PP64_DBG(p) ((p).dbg)
#else
#define PP64_DBG(p) 0
#endif

// ---------------------------------------------------------------------------------------------------------
// Ping-pong kernel with 64-deep K-slices (the production path for tile-aligned products).
// 256 x BN output tiles, 8 waves (BN = 256: 2 x 4 waves of 128 x 64; BN = 128: 4 x 2 waves of 64 x 64), operands by
// LDS-DMA.  The eight waves are two groups of four (waves w and w+4 share a SIMD) running in ANTI-PHASE: group 1 executes
// one extra barrier before its loop, so while one wave of a SIMD issues its 32 MFMAs back to back its
// partner issues LDS reads and DMA descriptors.  With a single barrier per step both waves of a SIMD reach their ~500-cycle
// load section together and the matrix pipe idles through it (measured with the 32-deep predecessors of this kernel:
// single barrier 930, anti-phase 978, anti-phase + 64-deep slices 1056 TFLOP/s on the text shapes).
// A 64-deep slice makes every LDS-DMA wave-instruction move 8 rows x 128 B (whole cache lines) instead of 16 rows x 64 B:
// the texture-address path prices an instruction by the lines it touches, and with 64-B pieces the DMA stream alone (no
// MFMA) ran at 11.5 TB/s chip-wide, slower than the MFMAs it has to feed.
// LDS (160 KiB, all of it): A ring 3 x 32 KiB (the streamed operand: its next slice is issued three steps before it is
// read, enough for an HBM miss) + B ring 2 x (BN x 128 B).  A 64-deep slice is consumed in two
// 32-deep steps (one fragment set, 32 MFMAs per wave each):
//   even step 2u:   B1 | read k-half 0 of slice u; issue B_{u+1};                      lgkmcnt(0) | B2 | MFMAs
//   odd  step 2u+1: B1 | read k-half 1 of slice u; issue A_{u+2}; vmcnt(|A_{u+2}|);    lgkmcnt(0) | B2 | MFMAs
// Per-wave issue order is A0 B0 A1 | B1 A2 | B2 A3 | ..., so "all but the youngest A unit" at the end of an odd step
// means slice u+1 (A_{u+1}, B_{u+1}) has landed.  RAW / WAR: a unit is waited for at the
// end of the load phase BEFORE the one that first reads it, and a slot is refilled in the load phase AFTER the one
// that last read it (B_{u+1} -> slot of B_{u-1}, last read in step 2u-1; A_{u+2} -> slot of A_{u-1}, same).
template <bool TRANS, int ROWS>
__device__ __forceinline__ bf8 hfrag(const char* lds, int rbase, int ks, int lane) {
  if (!TRANS) {
    const int row = rbase + (lane & 15), kc = ks * 4 + (lane >> 4);
    return ds_read_b128_asm(lds + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4));
  } else {
    return ffrag<true, ROWS>(lds, rbase, ks, lane);
  }
}

template <bool TA, bool TB_KMAJOR, int BN, bool GRP = false>
__global__ __launch_bounds__(512, 2) void gemm_pp64_kernel(GemmParams p) {
  constexpr int MT = BN == 256 ? 8 : 4;
  constexpr int WROWS = MT * 16;
  constexpr int A_UNIT = GBM * 128, B_UNIT = BN * 128;     // bytes per 64-deep slice of each operand
  constexpr int A_LPU = 4;                                 // LDS-DMA instructions per wave per A unit
  constexpr int B_OFF = 3 * A_UNIT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = BN == 256 ? (wave >> 2) : (wave >> 1), wn = BN == 256 ? (wave & 3) : (wave & 1);
  const int grp = wave >> 2;

  const int ntiles = p.tiles_m * p.tiles_n;
  const int nwg = ntiles * p.splits;
  const int bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const int split = wg / ntiles, tile = wg - split * ntiles;
  // Tiles are walked in bands of p.band tile-rows, column by column inside a band: the ~32 tiles an XCD runs at once then
  // share band A panels (which stay in its 4-MiB L2 for the whole band) and 32/band B panels, instead of one tile-row
  // streaming every B panel through L2 (row-major order: PMC FETCH_SIZE was 5x the operand bytes on the QKV product).
  int tm, tn;
  if (p.band > 0) {
    const int band = tile / (p.band * p.tiles_n), within = tile - band * (p.band * p.tiles_n);
    const int rows = min(p.band, p.tiles_m - band * p.band);
    tn = within / rows; tm = band * p.band + (within - tn * rows);
  } else {
    // column groups of -p.band tile-columns, row by row inside a group: an XCD keeps its group's B panels (the weights: -band x
    // 512 KiB at K = 1024) in L2 for its whole chunk and streams each A panel once per group
    const int cg = -p.band;
    const int grp = tile / (cg * p.tiles_m), within = tile - grp * (cg * p.tiles_m);
    const int cols = min(cg, p.tiles_n - grp * cg);
    tm = within / cols; tn = grp * cg + (within - tm * cols);
  }
  // grouped launch: the tile-rows past tiles_m1 are the second product's (workgroup-uniform switch of the operand pointers)
  const bool second = GRP && tm >= p.tiles_m1;
  const bf16* gA = second ? p.A2 : p.A;
  const bf16* gB = second ? p.B2 : p.B;
  const int glda = second ? p.lda2 : p.lda, gldb = second ? p.ldb2 : p.ldb;
  const int m0 = (second ? tm - p.tiles_m1 : tm) * GBM, n0 = tn * BN;
  const int kbeg = split * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int ns = (kend - kbeg) / 64;

  f4 acc[MT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};

  // bias gradient riding on the weight-gradient product (GemmParams::colsum): in the tile-column-0 blocks wave (wm, wn) also sums
  // the A rows of its sub-tiles 2 wn, 2 wn + 1 over k with two extra MFMAs per 32-deep step against an all-ones operand
  constexpr bool COLSUM_OK = TA && !TB_KMAJOR && BN == 256 && !GRP;
  const bool do_colsum = COLSUM_OK && p.colsum != nullptr && tn == 0;
  f4 bacc[2] = {(f4){0.f, 0.f, 0.f, 0.f}, (f4){0.f, 0.f, 0.f, 0.f}};
  bf8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;
  const bool dma_on = !(PP64_DBG(p) & 1);
  constexpr int B_LPU = BN * 128 / (8 * 1024);             // LDS-DMA pieces per wave per B unit (4 or 2)
  constexpr bool BAL = PP64_BAL_LAYOUT(TA, TB_KMAJOR);
  if (dma_on) {
    dma_tile<TA, GBM>(gA, glda, m0, kbeg, smem, wave, lane);
    dma_tile<!TB_KMAJOR, BN>(gB, gldb, n0, kbeg, smem + B_OFF, wave, lane);
    if (ns > 1) dma_tile<TA, GBM>(gA, glda, m0, kbeg + 64, smem + A_UNIT, wave, lane);
    if (BAL && ns > 1) dma_tile<!TB_KMAJOR, BN>(gB, gldb, n0, kbeg + 64, smem + B_OFF + B_UNIT, wave, lane);
  }
  if (ns > 1) {
    if constexpr (BAL) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_LPU + B_LPU) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_LPU) : "memory");
  }
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (grp == 1) __builtin_amdgcn_s_barrier();            // the anti-phase offset
  bf8 af[MT], bfr[8];                                    // the balanced schedule holds the B fragments of both k-halves
  int sa = 0;                                            // A slot of slice u (u mod 3)
#define PP64_READ(KS)                                                                                    \
  {                                                                                                      \
    const char* la = smem + sa * A_UNIT;                                                                 \
    const char* lb = smem + B_OFF + (u & 1) * B_UNIT;                                                    \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) bfr[j] = hfrag<!TB_KMAJOR, BN>(lb, wn * 64 + j * 16, KS, lane); \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) af[i] = hfrag<TA, GBM>(la, wm * WROWS + i * 16, KS, lane);     \
  }
#define PP64_MFMA()                                                                                      \
  __builtin_amdgcn_s_barrier();                                                                          \
  __builtin_amdgcn_sched_barrier(0);                                                                     \
  if (!(PP64_DBG(p) & 4)) {                                                                                   \
    /* no s_setprio around the MFMAs: measured 1 % slower with it on this schedule (the partner wave is in its load phase) */ \
    _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                       \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                      \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);          \
    if constexpr (COLSUM_OK) {                                                                           \
      if (do_colsum) {            /* wave-uniform; af[] indexed with compile-time constants only */        \
        if (wn == 0) { bacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[0], bacc[0], 0, 0, 0); bacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[1], bacc[1], 0, 0, 0); } \
        else if (wn == 1) { bacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[2], bacc[0], 0, 0, 0); bacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[3], bacc[1], 0, 0, 0); } \
        else if (wn == 2) { bacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[MT > 4 ? 4 : 0], bacc[0], 0, 0, 0); bacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[MT > 4 ? 5 : 1], bacc[1], 0, 0, 0); } \
        else { bacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[MT > 4 ? 6 : 2], bacc[0], 0, 0, 0); bacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[MT > 4 ? 7 : 3], bacc[1], 0, 0, 0); } \
      }                                                                                                  \
    }                                                                                                    \
  }                                                                                                      \
  __builtin_amdgcn_sched_barrier(0);
  // source addresses of this wave's four A pieces, kept in registers and advanced by one slice per issue (recomputing them among
  // the MFMAs, with all fragments live, spilled): piece i of slice 2 first
  const bf16* asrc[4];
  {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int blk = wave * 4 + i;
      if (!TA) {
        const int row = blk * 8 + (lane >> 3), pos = lane & 7;
        asrc[i] = gA + (size_t)(m0 + row) * glda + kbeg + 2 * 64 + (pos ^ ((row >> 1) & 7)) * 8;
      } else {
        const int k = blk * 2 + lane / 32, pc = lane % 32;      // GBM = 256: 32 chunks per k-row, 2 k-rows per 1-KiB block
        asrc[i] = gA + (size_t)(kbeg + 2 * 64 + k) * glda + m0 + ((((pc >> 1) ^ ftr_key(k)) << 1) | (pc & 1)) * 8;
      }
    }
  }
  const size_t astep = TA ? (size_t)64 * glda : (size_t)64;
  const bf16* bsrc[B_LPU];                                 // the same for B: piece i of slice 1 (balanced schedule: 2) first
  constexpr int BS0 = BAL ? 128 : 64;
  {
#pragma unroll
    for (int i = 0; i < B_LPU; ++i) {
      const int blk = wave * B_LPU + i;
      if (TB_KMAJOR) {
        const int row = blk * 8 + (lane >> 3), pos = lane & 7;
        bsrc[i] = gB + (size_t)(n0 + row) * gldb + kbeg + BS0 + (pos ^ ((row >> 1) & 7)) * 8;
      } else {
        constexpr int CPR = BN / 8, RPB = 64 / CPR;
        const int k = blk * RPB + lane / CPR, pc = lane % CPR;
        bsrc[i] = gB + (size_t)(kbeg + BS0 + k) * gldb + n0 + ((((pc >> 1) ^ ftr_key(k)) << 1) | (pc & 1)) * 8;
      }
    }
  }
  const size_t bstep = TB_KMAJOR ? (size_t)64 : (size_t)64 * gldb;
  // PP64_A_IN_MFMA: the four LDS-DMA pieces of A_{u+2} are issued BETWEEN the MFMAs of the odd step (one piece after every quarter
  // of the wave's MFMAs) instead of in its load phase.  An LDS-DMA instruction costs 100-185 issue cycles inside a phase that
  // already carries ds_read_b128s and other pieces, 25-60 in the gaps of an MFMA stream (MI355X_MICROARCH.md, "LDS-DMA piece issue
  // cost"), and the load phases are what the partner wave's MFMA phase has to cover.  Order per wave unchanged (B_{u+1} in the even
  // step, then A_{u+2}); the wait for slice u + 1 stays at the end of the odd step's LOAD phase (see there).
#define PP64_MFMA_A() PP64_MFMA_X(doA, 4, asrc, astep, smem + sn * A_UNIT)
#define PP64_MFMA_B() PP64_MFMA_X(doB, B_LPU, bsrc, bstep, smem + B_OFF + ((u + 1) & 1) * B_UNIT)
#define PP64_MFMA_X(DOX, NPC, SRC, STEP, DST)                                                            \
  __builtin_amdgcn_s_barrier();                                                                          \
  __builtin_amdgcn_sched_barrier(0);                                                                     \
  if (!(PP64_DBG(p) & 4)) {                                                                              \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                                     \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                      \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);          \
      if (((i + 1) % (MT / NPC)) == 0 && DOX) {          /* DOX: wave-uniform */                         \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        __builtin_amdgcn_global_load_lds((glb_void_ptr)SRC[i / (MT / NPC)], (lds_void_ptr)(DST + (wave * NPC + i / (MT / NPC)) * 1024), 16, 0, 0); \
        SRC[i / (MT / NPC)] += STEP;                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                               \
      }                                                                                                  \
    }                                                                                                    \
    if constexpr (COLSUM_OK) {                                                                           \
      if (do_colsum) {                                                                                   \
        if (wn == 0) { bacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[0], bacc[0], 0, 0, 0); bacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[1], bacc[1], 0, 0, 0); } \
        else if (wn == 1) { bacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[2], bacc[0], 0, 0, 0); bacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[3], bacc[1], 0, 0, 0); } \
        else if (wn == 2) { bacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[MT > 4 ? 4 : 0], bacc[0], 0, 0, 0); bacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[MT > 4 ? 5 : 1], bacc[1], 0, 0, 0); } \
        else { bacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[MT > 4 ? 6 : 2], bacc[0], 0, 0, 0); bacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[MT > 4 ? 7 : 3], bacc[1], 0, 0, 0); } \
      }                                                                                                  \
    }                                                                                                    \
  }                                                                                                      \
  __builtin_amdgcn_sched_barrier(0);
#define PP64_COLSUM()                                                                                    \
    if constexpr (COLSUM_OK) {                                                                           \
      if (do_colsum) {                                                                                   \
        if (wn == 0) { bacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[0], bacc[0], 0, 0, 0); bacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[1], bacc[1], 0, 0, 0); } \
        else if (wn == 1) { bacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[2], bacc[0], 0, 0, 0); bacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[3], bacc[1], 0, 0, 0); } \
        else if (wn == 2) { bacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[MT > 4 ? 4 : 0], bacc[0], 0, 0, 0); bacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[MT > 4 ? 5 : 1], bacc[1], 0, 0, 0); } \
        else { bacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[MT > 4 ? 6 : 2], bacc[0], 0, 0, 0); bacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[MT > 4 ? 7 : 3], bacc[1], 0, 0, 0); } \
      }                                                                                                  \
    }
  // MFMA phase of one 32-deep step with pieces [P0, P1) of one operand's next unit issued behind MFMA rows 1, MT/2 + 1, ... (early in
  // the phase: a piece issued behind the last MFMAs would put its issue latency in front of the barrier)
#define PP64_MFMA_BAL(JB, DOX, P0, P1, SRC, STEP, DST, NPW)                                              \
  __builtin_amdgcn_s_barrier();                                                                          \
  __builtin_amdgcn_sched_barrier(0);                                                                     \
  if (!(PP64_DBG(p) & 4)) {                                                                              \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                                     \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                      \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[JB + j], af[i], acc[i][j], 0, 0, 0);     \
      constexpr int NPC = (P1) - (P0);                                                                   \
      constexpr int GAP = NPC > 0 ? MT / NPC : MT;                                                       \
      if (NPC > 0 && (i % GAP) == (GAP > 1 ? 1 : 0) && i / GAP < NPC && DOX) {                           \
        const int pc = (P0) + i / GAP;                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        __builtin_amdgcn_global_load_lds((glb_void_ptr)SRC[pc], (lds_void_ptr)(DST + (wave * NPW + pc) * 1024), 16, 0, 0); \
        SRC[pc] += STEP;                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                               \
      }                                                                                                  \
    }                                                                                                    \
    PP64_COLSUM()                                                                                        \
  }                                                                                                      \
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (BAL) {
  for (int u = 0; u < ns; ++u) {
    const bool donext = u + 2 < ns && dma_on;              // slice u + 2 exists: this slice issues A_{u+2} and B_{u+2}
    int sn = sa + 2; if (sn >= 3) sn -= 3;
    char* adst = smem + sn * A_UNIT;
    char* bdst = smem + B_OFF + (u & 1) * B_UNIT;          // B_{u+2} takes B_u's slot
    // ---- even step: B fragments of the whole slice, A fragments of k-half 0
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    {
      const char* la = smem + sa * A_UNIT;
      const char* lb = smem + B_OFF + (u & 1) * B_UNIT;
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = hfrag<!TB_KMAJOR, BN>(lb, wn * 64 + j * 16, 0, lane);
#pragma unroll
      for (int i = 0; i < MT; ++i) af[i] = hfrag<TA, GBM>(la, wm * WROWS + i * 16, 0, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[4 + j] = hfrag<!TB_KMAJOR, BN>(lb, wn * 64 + j * 16, 1, lane);
    }
    if (donext) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        __builtin_amdgcn_global_load_lds((glb_void_ptr)asrc[i], (lds_void_ptr)(adst + (wave * 4 + i) * 1024), 16, 0, 0);
        asrc[i] += astep;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PP64_MFMA_BAL(0, donext, 2, 4, asrc, astep, adst, 4)
    // ---- odd step: A fragments of k-half 1; B_u's slot is free (every wave has passed the barrier behind its even load phase)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    {
      const char* la = smem + sa * A_UNIT;
#pragma unroll
      for (int i = 0; i < MT; ++i) af[i] = hfrag<TA, GBM>(la, wm * WROWS + i * 16, 1, lane);
    }
    if (donext) {
#pragma unroll
      for (int i = 0; i < B_LPU / 2; ++i) {
        __builtin_amdgcn_global_load_lds((glb_void_ptr)bsrc[i], (lds_void_ptr)(bdst + (wave * B_LPU + i) * 1024), 16, 0, 0);
        bsrc[i] += bstep;
      }
      // slice u + 1 landed: all but A_{u+2} and the first half of B_{u+2}.  BEFORE the barrier below -- the other wave group, one
      // barrier ahead, reads slice u + 1 right behind it.
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_LPU + B_LPU / 2) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PP64_MFMA_BAL(4, donext, B_LPU / 2, B_LPU, bsrc, bstep, bdst, B_LPU)
    sa = sa + 1; if (sa >= 3) sa = 0;
  }
  } else {
  for (int u = 0; u < ns; ++u) {
    // ---- even step: k-half 0 of slice u
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    PP64_READ(0)
#if PP64_A_IN_MFMA && PP64_B_IN_MFMA
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    {
      const bool doB = u + 1 < ns && dma_on;
      PP64_MFMA_B()
    }
#else
    if constexpr (PP64_B_SPLIT && TA && !TB_KMAJOR) {
      const bool doB = u + 1 < ns && dma_on;
      char* bdst = smem + B_OFF + ((u + 1) & 1) * B_UNIT;
      constexpr int BKEEP = B_LPU == 4 ? PP64_B_KEEP : B_LPU / 2;
      if (doB) {
#pragma unroll
        for (int i = 0; i < BKEEP; ++i) {
          __builtin_amdgcn_global_load_lds((glb_void_ptr)bsrc[i], (lds_void_ptr)(bdst + (wave * B_LPU + i) * 1024), 16, 0, 0);
          bsrc[i] += bstep;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        if (i < B_LPU - BKEEP && doB) {
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_global_load_lds((glb_void_ptr)bsrc[BKEEP + i], (lds_void_ptr)(bdst + (wave * B_LPU + BKEEP + i) * 1024), 16, 0, 0);
          bsrc[BKEEP + i] += bstep;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if constexpr (COLSUM_OK) {
        if (do_colsum) {
          if (wn == 0) { bacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[0], bacc[0], 0, 0, 0); bacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[1], bacc[1], 0, 0, 0); }
          else if (wn == 1) { bacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[2], bacc[0], 0, 0, 0); bacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[3], bacc[1], 0, 0, 0); }
          else if (wn == 2) { bacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[MT > 4 ? 4 : 0], bacc[0], 0, 0, 0); bacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[MT > 4 ? 5 : 1], bacc[1], 0, 0, 0); }
          else { bacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[MT > 4 ? 6 : 2], bacc[0], 0, 0, 0); bacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[MT > 4 ? 7 : 3], bacc[1], 0, 0, 0); }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    } else {
      if (u + 1 < ns && dma_on)
        dma_tile<!TB_KMAJOR, BN>(gB, gldb, n0, kbeg + (u + 1) * 64, smem + B_OFF + ((u + 1) & 1) * B_UNIT, wave, lane);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      PP64_MFMA()
    }
#endif
    // ---- odd step: k-half 1 of slice u
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    PP64_READ(1)
#if PP64_A_IN_MFMA
    // Slice u + 1 (A_{u+1}: issued during the previous odd step's MFMAs, B_{u+1}: in this slice's even step) must have landed
    // BEFORE the barrier below: the other wave group, one barrier ahead, starts reading it right after that barrier.  (Waiting after
    // the MFMAs -- "all but the four pieces just issued" -- is one barrier too late for the lagging group's pieces: measured as wrong
    // results on the forward layout.)  Nothing younger is in flight at this point, so the wait is a full drain and nearly free.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    {
      const bool doA = u + 2 < ns && dma_on;
      int sn = sa + 2; if (sn >= 3) sn -= 3;
      PP64_MFMA_A()
    }
#else
    if (u + 2 < ns) {
      int sn = sa + 2; if (sn >= 3) sn -= 3;
      if (dma_on) dma_tile<TA, GBM>(gA, glda, m0, kbeg + (u + 2) * 64, smem + sn * A_UNIT, wave, lane);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_LPU) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PP64_MFMA()
#endif
    sa = sa + 1; if (sa >= 3) sa = 0;
  }
  }
#undef PP64_MFMA_BAL
#undef PP64_COLSUM
#undef PP64_MFMA_A
#undef PP64_MFMA_B
#undef PP64_MFMA_X
#undef PP64_READ
#undef PP64_MFMA
  if (grp == 0) __builtin_amdgcn_s_barrier();            // both groups have now executed 4 ns + 1 barriers
  if constexpr (COLSUM_OK) {
    // bacc[s][e] = sum_k A[k][m] for m = row (lane & 15) of sub-tile 2 wn + s, the same value in every column e / lane group
    if (do_colsum && (lane >> 4) == 0) {
      float* dst = p.colsum + m0 + wm * WROWS + (2 * wn) * 16 + (lane & 15);
      atomicAdd(dst, bacc[0][0]);
      atomicAdd(dst + 16, bacc[1][0]);
    }
  }
  // Round 3, measured and not kept: a PERSISTENT form (one workgroup per CU walks its CU's tiles; the next tile's first slices are
  // requested by LDS-DMA right after the main loop and land under the epilogue, which then has to work through small staging tiles in
  // the ring slots the prefetch leaves alone).  (a) A0 | B0 | A1 ahead, fp32 16-row slices through A slot 2 / B slot 1: forward
  // layout +3 / +2 / +4 % at K = 1024 (qkv 1 075-1 093 -> 1 108-1 124, o 1 103 -> 1 127, ffn1 1 180 -> 1 226 TFLOP/s), -3 % at K = 4096,
  // data-gradient layout +-2 %, the layer's eight products together 2.264 against 2.258-2.264 ms: the next tile's B1 is issued after
  // this tile's output stores, gfx9 has one in-order memory counter, so the first step's wait for B1 is a wait for the stores'
  // acknowledgements.  (b) B1 ahead as well (128 KiB; one slot left, so the epilogue does bias / GELU in the accumulator layout and
  // transposes bf16 slices): 256 registers, spills in the data-gradient form, -3 ... -16 %.  What persistence can hide is the
  // prologue's latency, not its sixteen 100-cycle DMA issues per wave; tools/bench_gemm.py, MMSIM_GEMM_PERSIST builds of this round.
  // Round 3, measured and not kept: de-phasing the CUs (the odd half of the first round of workgroups waits 25 / 50 / 100 % of a tile's
  // main-loop time before its prologue, so that one half's epilogue write burst meets the other half's main loop): 2.35-2.36 against
  // 2.32-2.33 ms per layer of GEMMs (tools/bench_gemm.py) -- slightly slower; the lockstep of the rounds is not what the epilogue costs.
  // Round 3, measured and not kept: pulling the NEXT workgroup's first operand slices (the workgroup 32 further in this XCD's
  // chunk: A slices 0 and 1, B slice 0, 96 KiB) into this XCD's L2 with register-destination loads issued between the two
  // epilogue halves, so that the chip-wide prologue burst (MI355X_MICROARCH.md: ~9k cycles per 96 KiB with every CU in its
  // prologue) finds its lines on-die: 1661 / 1681 us against 1675 / 1675 us over the eight forward / data-gradient products of a
  // layer (tools/bench_gemm_epi.py, alternating builds on one box) -- no gain; the burst is not an L2-miss problem.
  // THE ABORT OF THAT EXPERIMENT (gpurun_out/ab_pf2.log of round 3, HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION "beyond the largest legal
  // address", both A/B rounds of the first build): the successor was computed as wg + 32 with NO range check.  For the
  // last 32 workgroups of the launch (the tail of the last XCD's chunk) wg + 32 >= tiles x splits, so the decoded (split, tile) was past
  // the end -- tile row >= tiles_m (rows >= M of A) and, with split-K, split >= splits (k >= K) -- and the loads were REAL
  // register-destination loads, not hints: base + (row past M) * lda lies outside the operand and, for the tensor that sits last in
  // the allocator's range, outside the process's aperture.  Every other workgroup prefetched a valid successor, which is why small
  // test shapes (whose last workgroups' successors still landed inside a neighbouring allocation) passed.  The second build
  // (ab_pf1.log, eight minutes later) predicated the prefetch on wg + 32 < nwg and ran clean.  RULE for any next-tile prefetch brought
  // back here: (a) wave-uniform predicate `successor < tiles x splits`, (b) clamp the row / k indices to the operand anyway (clamped
  // address, discarded value: the idiom of the generic kernels), (c) run tools/bench_gemm.py once at M = 32 768 -- the fault needs the
  // tail workgroups of a LARGE launch.
  // Round 4, measured and not kept: the STAGGERED PERSISTENT form without idle time -- one workgroup per CU walking the virtual block ids
  // b, b + G, ..., the workgroups dealt into 2 or 4 phase groups, group g > 0 splitting ITS OWN first tile in K (upper K range first, the
  // fp32 accumulators parked in a private 256-KiB slab as a raw register dump, the lower K range at the very end on top of the parked sums:
  // no cross-workgroup communication), so that tile boundaries -- and the epilogue write bursts, which are what the K = 1 024 products
  // lose 30-47 % of their time to (qkv 207 us against 147 us of main loop, the GELU-pair product 370 against 196; 256 CUs store in
  // lockstep and each gets 1 / 256 of the HBM write bandwidth) -- stay g / groups of a tile apart for the whole launch.  Results correct;
  // First build: wrapping the kernel body in the tile loop made hipcc hoist every per-lane invariant (fragment read offsets, epilogue
  // addresses) out of the loop and across the main loop: 256 registers + 224-336 B of scratch per lane, 2 115-2 180 us over the eight
  // forward / data-gradient products of a layer (tools/bench_gemm_epi.py) against 1 710-1 733.  Second build: the thread index
  // "laundered" once per tile (`asm volatile("" : "+v"(tid))`, everything per-lane derives from it) -- 220-222 registers, no scratch.
  // Measured, alternating with the plain launch on one box, two rounds: 2 groups 1 718 / 1 731 us, 4 groups 1 730 / 1 739, 8 groups
  // 1 791 / 1 780 against 1 641 / 1 647 (and 1 640 / 1 630 for the kernel before the change): 5-9 % SLOWER, every shape, more groups
  // worse.  So the result of round 3's sleep stagger stands and is now free of its idle time: the lockstep of the rounds is NOT what
  // the epilogues cost.  What they cost is their own work: the GELU pair is ~25 vector instructions on each of 134 M elements (~56 us
  // of the chip's whole VALU rate per launch, with the matrix pipes idle) plus 2 x 268 MB of stores; de-phasing adds two pipeline
  // ramps and a 256-KiB park / restore per staggered workgroup and takes the L2 panel sharing of lockstep neighbours away.
  const bool fs = split == 0;
  float* stg = reinterpret_cast<float*>(smem) + wave * (64 * EP_PITCH);
  if (PP64_DBG(p) & 8) {
    float sacc = 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) sacc += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (sacc == 12345.678f) reinterpret_cast<float*>(p.C)[0] = sacc;
    return;
  }
  // the two 64-row halves of a wave's sub-tile, written out explicitly: the accumulator array must only ever be indexed
  // with compile-time constants (an un-unrolled loop over `half` sends all 128 accumulators to scratch)
  auto epi_half = [&](f4 (&a4)[4][4], int half) __attribute__((always_inline)) {
    __syncthreads();
    const int row0 = m0 + wm * WROWS + half * 64, col0 = n0 + wn * 64;
    if (!p.c_f32) fast_epilogue_epi<0>(p, a4, row0, col0, lane, fs, stg);
    else if (GRP && second) {              // the second product's output (grouped launches are split-K atomics only)
      GemmParams q = p;
      q.C = p.C2; q.ldc = p.ldc2;
      fast_epilogue<EPI_NONE, 3>(q, a4, row0, col0, lane, fs, stg);
    }
    else if (p.atomic) fast_epilogue<EPI_NONE, 3>(p, a4, row0, col0, lane, fs, stg);
    else if (TA && !TB_KMAJOR && BN == 256 && !GRP && p.epi == EPI_ROWFIX) {
      // the ArcFace weight gradient (head.py): rows past M (the class count is not a multiple of 256) are predicated
      if (p.accum) fast_epilogue<EPI_ROWFIX, 2, true>(p, a4, row0, col0, lane, fs, stg);
      else fast_epilogue<EPI_ROWFIX, 1, true>(p, a4, row0, col0, lane, fs, stg);
    }
    else if (p.accum) fast_epilogue<EPI_NONE, 2>(p, a4, row0, col0, lane, fs, stg);
    else if (p.epi == EPI_TANH) fast_epilogue<EPI_TANH, 1>(p, a4, row0, col0, lane, fs, stg);
    else if (!TA && TB_KMAJOR && BN == 256 && !GRP && p.epi == EPI_ARCSTATS) fast_epilogue<EPI_ARCSTATS, 1>(p, a4, row0, col0, lane, fs, stg);
    else fast_epilogue<EPI_NONE, 1>(p, a4, row0, col0, lane, fs, stg);
  };
  epi_half(*reinterpret_cast<f4 (*)[4][4]>(&acc[0][0]), 0);
  if constexpr (MT == 8) epi_half(*reinterpret_cast<f4 (*)[4][4]>(&acc[4][0]), 1);
}

// ---------------------------------------------------------------------------------------------------------
// Round 4, measured and not kept: ONE WAVE PER SIMD, 128 x 128 PER WAVE (gemm_w4_kernel; VERDICT r3 item 2 "measure it instead of costing
// it").  256 x 256 tile, four waves as 2 x 2, 8 x 8 MFMA tiles = 256 accumulators in the AGPR half of the unified file, a double-buffered
// fragment set (2 x 16 x 4 registers; 16 ds_read_b128 per 64 MFMAs instead of pp64's 12 per 32), the rings of gemm_pp64_kernel, ONE barrier
// per 64-deep slice (between its two steps), the next step's fragments read behind MFMA rows 0..3 and the LDS-DMA pieces of B_{u+2} then
// A_{u+3} behind the odd step's rows (B before A: vmcnt counts in issue order -- interleaving them let vmcnt(8) pass with half of B in
// flight, a cold-cache-only wrong result that the after-load re-check of tools/bench_gemm.py caught).  What it took to get a clean loop
// out of hipcc (148-208 VGPRs + 256 AGPRs, no scratch, the steady-state slice = 128 MFMAs + 32 reads + 16 DMAs + ~40 scalar):
//   (a) every fragment read as per-lane base + IMMEDIATE offset (`ds_read_b128 %0, %1 offset:%2`), every DMA piece as wave-uniform base +
//       one of two per-lane offsets; (b) the epilogue's four 64 x 64 quadrants written out (a runtime quadrant index sends all 256
//       accumulators through scratch); (c) the tail peeled so the steady loop is ONE basic block; (d) the MFMAs as in-place inline asm
//       ("+a" tied accumulator): with the builtin and no free AGPR the allocator rotates ~180 accumulator registers per slice at the loop
//       header (v_accvgpr_mov chains, ~1/3 of the loop's issue slots).
// Forward layout (NT, bias epilogue), M = 32 768, tools/bench_gemm.py, alternating processes on one box, TFLOP/s, w4 | pp64 | hipBLASLt:
//   qkv (N 3072, K 1024) 1 027 | 1 058 | 1 075      o (N 1024, K 1024) 1 030 | 1 082 | 1 144
//   ffn1 (N 4096, K 1024) 1 080 | 1 145 | 1 250     ffn2 (N 1024, K 4096) 1 339 | 1 315 | 1 434
// Solving tile time = (K / 64) s + o from the K = 1 024 and K = 4 096 rows: s = 1.44 us per slice and o = 10.3 us per tile for w4,
// s = 1.37 / o = 8.0 for hipBLASLt (pp64 sits between): a slice is 2 048 MFMA cycles, i.e. all three run the matrix pipes at an
// EFFECTIVE ~1.45 GHz in the main loop and differ by <= 5 % there; the fixed ~8-10 us per tile (prologue ramp + the 128-KiB epilogue
// with the matrix pipes idle) is 25-30 % of a K = 1 024 tile and is where the forms differ.  w4 halves the fragment bytes per MFMA
// and gains 2 % at K = 4 096, but its epilogue runs on four waves instead of eight and loses 3-6 % at K = 1 024.  Not kept; the
// >= 10 % main-loop gain that would have justified the NN / TN forms is not there to be had -- the main loop is not LDS-bound.
// The row-fix weight gradient (EPI_ROWFIX: C[M,N] f32 = r[m] (A^T B - aux r'[m]), the ArcFace head's dW with the backward of
// F.normalize folded in) on the pipelined 256 x 256 kernel although M (the class count) is not a multiple of 256: A is stored
// [K][lda] with lda covering M rounded up to 256 (head.py pads the class dimension of dcos with zero columns), so the operand
// tiles are read whole and only the epilogue predicates rows >= M.  It is a pure output stream (1.1 GB of fp32 at 100 000 x 2816,
// K = 256): the generic 128 x 128 kernel ran it at 1.5 TB/s.
static bool rowfix_eligible(const GemmParams& p, int splits) {
  return p.epi == EPI_ROWFIX && p.c_f32 && splits == 1 && !p.atomic && (p.N % 256) == 0 && (p.K % 64) == 0 && p.K >= 128 &&
         p.lda >= ((p.M + GBM - 1) / GBM) * GBM && (p.ld_aux % 4) == 0 && ((p.M + GBM - 1) / GBM) * (p.N / 256) >= 128;
}
static bool pipe_eligible(const GemmParams& p, int splits, int bn) {
  if (p.M % GBM || p.N % bn || p.K % 64 || p.k_per_split % 64) return false;
  return (p.M / GBM) * (p.N / bn) * splits >= 128;        // enough blocks to fill the chip
}
static bool fast128_eligible(const GemmParams& p, int splits) {
  if (p.M % FBM || p.N % FBN) return false;
  if (p.K % FBK || p.k_per_split % FBK) return false;
  return true;
}

bool gemm_fast_rowfix(const GemmParams& p, int splits, int trans_a, int b_kmajor) { return trans_a && !b_kmajor && rowfix_eligible(p, splits); }
bool gemm_fast_arcstats(const GemmParams& p) {        // the cosine product with the softmax-statistics epilogue: forward layout, f32 output, 256 x 256 tiles
  return p.epi == EPI_ARCSTATS && p.c_f32 && !p.atomic && !p.accum && (p.ldc % 4) == 0 && pipe_eligible(p, 1, 256);
}
bool gemm_fast_eligible(const GemmParams& p, int splits) {
  if (p.epi == EPI_ARCSTATS) return splits == 1 && gemm_fast_arcstats(p);
  if (p.c_f32 && !(p.epi == EPI_NONE || p.epi == EPI_TANH)) return false;   // f32 outputs: plain / tanh only (row-fix: gemm_fast_rowfix)
  if ((p.ldc % 4) || (p.ld_aux % 4)) return false;
  if ((long long)p.k_per_split * splits < p.K) return false;
  return pipe_eligible(p, splits, 256) || pipe_eligible(p, splits, 128) || fast128_eligible(p, splits);
}

static int tile_pref() {      // MMSIM_GEMM_TILE: 0 auto (default), 1 = old 256x128x64 kernel only, 2 = pipelined BN=128 only
  static int v = -1;
  if (v < 0) { const char* e = getenv("MMSIM_GEMM_TILE"); v = e ? atoi(e) : 0; }
  return v;
}

template <int BN>
static void launch_pipe(GemmParams p, int trans_a, int b_kmajor, int splits, hipStream_t s) {
  p.tiles_m = (p.M + GBM - 1) / GBM; p.tiles_n = p.N / BN; p.splits = splits;      // M % 256 != 0 only for the row-fix product (rowfix_eligible)
  dim3 grid(p.tiles_m * p.tiles_n * splits), block(512);
  // 160 KiB: A ring 3 x 32 KiB + B ring 2 x (BN x 128 B); never less than the epilogue staging (8 waves x 64 x 68 floats)
  const size_t lds_stage = 8 * 64 * EP_PITCH * 4, lds_ring = 3 * GBM * 128 + 2 * BN * 128;
  const size_t lds = lds_ring > lds_stage ? lds_ring : lds_stage;
  static unsigned long long done = 0;          // per device: the opt-in is a property of (function, device)
  const int dev = mmsim_current_device();
  if (!((done >> dev) & 1)) {
    (void)hipFuncSetAttribute((const void*)gemm_pp64_kernel<false, true, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_pp64_kernel<false, false, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_pp64_kernel<true, false, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_pp64_kernel<true, true, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    done |= 1ull << dev;
  }
  {
    static int band = -1000;         // MMSIM_GEMM_BAND: tile-rows per band of the tile walk (1 = row-major); negative: column groups
    if (band == -1000) { const char* e = getenv("MMSIM_GEMM_BAND"); band = e ? atoi(e) : 8; if (band == 0) band = 1; }
    p.band = (band < 0 && trans_a) ? 8 : band;       // column groups (negative) only for the activation-times-weight products
  }
  if (!trans_a && b_kmajor) hipLaunchKernelGGL((gemm_pp64_kernel<false, true, BN>), grid, block, lds, s, p);
  else if (!trans_a && !b_kmajor) hipLaunchKernelGGL((gemm_pp64_kernel<false, false, BN>), grid, block, lds, s, p);
  else if (trans_a && !b_kmajor) hipLaunchKernelGGL((gemm_pp64_kernel<true, false, BN>), grid, block, lds, s, p);
  else hipLaunchKernelGGL((gemm_pp64_kernel<true, true, BN>), grid, block, lds, s, p);
}

void gemm_fast_launch(GemmParams p, int trans_a, int b_kmajor, int splits, hipStream_t s) {
  if (p.colsum || (p.epi == EPI_ROWFIX && p.c_f32) || p.epi == EPI_ARCSTATS) { launch_pipe<256>(p, trans_a, b_kmajor, splits, s); return; }      // only this kernel sums the columns (host checked the shape)
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
  static unsigned long long attr_done = 0;
  const int dev = mmsim_current_device();
  if (!((attr_done >> dev) & 1)) {
    (void)hipFuncSetAttribute((const void*)gemm_fast_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_fast_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_fast_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_fast_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done |= 1ull << dev;
  }
  if (!trans_a && b_kmajor) hipLaunchKernelGGL((gemm_fast_kernel<false, true>), grid, block, lds, s, p);
  else if (!trans_a && !b_kmajor) hipLaunchKernelGGL((gemm_fast_kernel<false, false>), grid, block, lds, s, p);
  else if (trans_a && !b_kmajor) hipLaunchKernelGGL((gemm_fast_kernel<true, false>), grid, block, lds, s, p);
  else hipLaunchKernelGGL((gemm_fast_kernel<true, true>), grid, block, lds, s, p);
}

// Two weight-gradient products of one layer as ONE launch of the pipelined kernel: C1[M1,N] += A1^T B1 and C2[M2,N] += A2^T B2,
// all operands stored [K][.] (dY and X as they lie in memory), f32 outputs through split-K atomics.  See GemmParams::A2.
extern "C" int mmsim_gemm_bf16_wgrad_pair(int M1, int M2, int N, int K, const void* A1, int lda1, const void* B1, int ldb1, float* C1,
                                          int ldc1, const void* A2, int lda2, const void* B2, int ldb2, float* C2, int ldc2, int split_k,
                                          void* stream) {
  MMSIM_REQUIRE(A1 && B1 && C1 && A2 && B2 && C2, "gemm_wgrad_pair: null operand");
  MMSIM_REQUIRE(M1 > 0 && M2 > 0 && M1 % GBM == 0 && M2 % GBM == 0 && N > 0 && N % 256 == 0 && K > 0 && K % 64 == 0,
                "gemm_wgrad_pair: M1, M2, N must be multiples of 256 and K of 64");
  MMSIM_REQUIRE(split_k >= 1, "gemm_wgrad_pair: split_k >= 1");
  if (mmsim_deterministic()) split_k = 1;          // one adder per output element
  MMSIM_REQUIRE(lda1 >= M1 && lda2 >= M2 && ldb1 >= N && ldb2 >= N && ldc1 >= N && ldc2 >= N, "gemm_wgrad_pair: leading dimension too small");
  MMSIM_REQUIRE((lda1 % 8) == 0 && (lda2 % 8) == 0 && (ldb1 % 8) == 0 && (ldb2 % 8) == 0 && (ldc1 % 4) == 0 && (ldc2 % 4) == 0,
                "gemm_wgrad_pair: lda / ldb must be multiples of 8, ldc of 4");
  MMSIM_REQUIRE(((uintptr_t)A1 % 16) == 0 && ((uintptr_t)B1 % 16) == 0 && ((uintptr_t)C1 % 16) == 0 && ((uintptr_t)A2 % 16) == 0 &&
                    ((uintptr_t)B2 % 16) == 0 && ((uintptr_t)C2 % 16) == 0, "gemm_wgrad_pair: operands must be 16-byte aligned");
  int kps = (K + split_k - 1) / split_k;
  kps = ((kps + 63) / 64) * 64;
  const int splits = (K + kps - 1) / kps;
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = (const bf16*)A1; p.B = (const bf16*)B1; p.C = C1; p.lda = lda1; p.ldb = ldb1; p.ldc = ldc1;
  p.A2 = (const bf16*)A2; p.B2 = (const bf16*)B2; p.C2 = C2; p.lda2 = lda2; p.ldb2 = ldb2; p.ldc2 = ldc2;
  p.M = M1 + M2; p.N = N; p.K = K; p.c_f32 = 1; p.epi = EPI_NONE; p.atomic = 1; p.alpha = 1.0f; p.k_per_split = kps;
  p.tiles_m1 = M1 / GBM; p.tiles_m = (M1 + M2) / GBM; p.tiles_n = N / 256; p.splits = splits; p.xf_hw = 1; p.xf_dhw = make_fastdiv(1);
  { const char* e = getenv("MMSIM_GEMM_BAND"); p.band = e ? atoi(e) : 8; if (p.band < 1) p.band = 1; }
  const size_t lds_stage = 8 * 64 * EP_PITCH * 4, lds_ring = 3 * GBM * 128 + 2 * 256 * 128;
  const size_t lds = lds_ring > lds_stage ? lds_ring : lds_stage;
  static unsigned long long done = 0;
  const int dev = mmsim_current_device();
  if (!((done >> dev) & 1)) {
    (void)hipFuncSetAttribute((const void*)gemm_pp64_kernel<true, false, 256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    done |= 1ull << dev;
  }
  hipLaunchKernelGGL((gemm_pp64_kernel<true, false, 256, true>), dim3(p.tiles_m * p.tiles_n * splits), dim3(512), lds, (hipStream_t)stream, p);
  return mmsim_check_launch("gemm_wgrad_pair");
}
