// op16 MFMA GEMM with fused epilogue for gfx950:   C[M,N] = epi( A[M,K] * W[N,K]^T )
//
// Serves every nn.Linear / 1x1 conv / im2col'ed conv on the hot path (reference call sites: hieradet.py:61,79,
// sam2_utils.py:127-131, transformer.py:241-243,261, memory_attention.py:96, image_encoder.py:112, mask_decoder.py:
// 240-256, memory_encoder.py:103-105,171-175).  Both operands are K-contiguous (activation rows and nn.Linear weight
// rows), so the 32x32x16 op16 MFMA fragments (8 consecutive k per lane) are plain 16-byte LDS reads.
//
// Tile: BM x BN x 32, WM x WN waves, double-buffered LDS with 80-byte rows (conflict-free ds_read_b128), register
// staging issued before the MFMA phase and written after it (one barrier per k-step).
// Epilogue (all optional, fp32): + bias[n] -> activation -> * colscale[n] -> + residual[m % res_mod][n] -> op16|f32.
#include "common.h"
#include <stdlib.h>

struct GemmParams {
  const op16* A;
  const op16* W;
  const float* bias;
  const float* colscale;
  const void* res;
  void* C;
  int64_t lda, ldw, ldr, ldc;
  int64_t res_mod;
  int M, N, K;
  int act;          // 0 none, 1 gelu(erf), 2 relu, 3 sigmoid
  int res_is_16bit;  // residual dtype
  int out_is_16bit;  // output dtype
  // optional fused axial RoPE on the 16-bit output (RoPEAttention q/k projections, transformer.py:299-315): columns
  // n < rope_cols are rotated as adjacent pairs (2i, 2i+1) of a head of rope_D channels; row m belongs to position
  // l = m % rope_period of its batch, rotated when l < rope_n with table row l % rope_npos (rope_k_repeat tiling)
  const float* rope_cos;
  const float* rope_sin;
  int rope_cols, rope_D, rope_period, rope_n, rope_npos;
  // optional fused 2x2 max-pool of the OUTPUT over the token image (Hiera's pooled shortcut, hieradet.py:141-145): logical row
  // m = 4 * pooled_pixel + (dy * 2 + dx) reads A row ((b*pool_H + 2*y2 + dy) * pool_W + 2*x2 + dx); the four rows of a pooled
  // pixel are the four registers (e & 3) of one lane in the MFMA accumulator layout, so the pool is a register max and only the
  // pooled rows [M/4, N] are ever written.  pool_W == 0: off.
  int pool_H, pool_W;
  // q-pool form (16-bit out): columns n < poolq_cols (the q third of a fused qkv) are max-pooled into Q2 [M/4, poolq_cols];
  // the other columns (k, v) are stored to their source pixel's row of C (image order); the q columns of C are not written
  void* Q2;
  int64_t ldq;
  int poolq_cols;
  // token form (skinny kernel only): A is fp32 [M, K] (row stride lda in floats) and columns n < add_cols use A + A2 instead
  // (decoder tokens: queries + query_pe for the q | k thirds of a fused projection, queries alone for v)
  const float* A32;
  const float* A2;
  int add_cols;
  // 16-bit outputs larger than the 256 MB Infinity Cache are stored non-temporal: measured 125 -> 87 us on the 302 MB qkv output of
  // Hiera stage 2's first block, while outputs that fit are 15-20 % SLOWER with nt (they are absorbed by the cache and read back
  // from it by the next kernel)
  int store_nt;
  // split-K (register-staged kernel only): blockIdx.z owns the k-tiles [z * ksplit_tiles, (z+1) * ksplit_tiles) and adds its partial
  // product into the zeroed fp32 output with atomics (bias / residual contributed by z == 0).  For the weight-gradient GEMMs of the
  // backward pass: a [256 x 256] output reduced over 16k..64k tokens is 4 tiles, i.e. 4 workgroups walking K serially otherwise.
  int ksplit_tiles;
};

// A row (in elements of lda) that logical GEMM row m reads
__device__ __forceinline__ int64_t gemm_a_row(const GemmParams& p, int64_t m) {
  if (p.pool_W == 0) return m;
  // 32-bit arithmetic (M < 2^31 is checked on the host): the 64-bit form cost four ~100-instruction divisions per thread in the prologue
  // of every tile of the pooled GEMMs, whose main loop is 3-12 k-steps (round 4)
  const unsigned mu = (unsigned)m;
  const unsigned s = mu & 3u, pix = mu >> 2;
  const unsigned w2 = (unsigned)p.pool_W >> 1, h2 = (unsigned)p.pool_H >> 1;
  const unsigned t = pix / w2, x2 = pix - t * w2;
  const unsigned b = t / h2, y2 = t - b * h2;
  return (int64_t)((b * (unsigned)p.pool_H + 2 * y2 + (s >> 1)) * (unsigned)p.pool_W + 2 * x2 + (s & 1));
}

#ifdef MSAM2_GSTAMP
// diagnostic build only (tools/gemm_probe.hip): per-workgroup time stamps, never compiled into the product library
__device__ unsigned long long g_gstamp[8 * 8192];
__device__ int g_epi_mode;   // experiment: 1 = skip global stores, 2 = skip the LDS dump, 3 = both
#define GSTAMP(slot)                                                                                   \
  do {                                                                                                 \
    unsigned long long t_;                                                                             \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
    if (threadIdx.x == 0 && blockIdx.x < 8192) g_gstamp[blockIdx.x * 8 + (slot)] = t_;                \
  } while (0)
#define GSTAMP_NW(slot)                                                                                \
  do {                                                                                                 \
    unsigned long long t_;                                                                             \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                        \
    if (threadIdx.x == 0 && blockIdx.x < 8192) g_gstamp[blockIdx.x * 8 + (slot)] = t_;                \
  } while (0)
#else
#define GSTAMP(slot)
#define GSTAMP_NW(slot)
#endif

constexpr int GEMM_BK = 32;
constexpr int GEMM_LDS_STRIDE = 40;  // op16 elements per LDS row (32 data + 8 pad = 80 B)

// ---- shared epilogue.  Each wave re-lays its accumulators through a private fp32 LDS scratch ([32][TN+4], rows of the C tile
// contiguous) so that bias / activation / layer-scale / residual and the global stores work on 16-byte row segments
// (full 128..256-byte lines per 16 lanes) instead of one 2- or 4-byte element per lane.
// Generic version: any alignment, any N, every mode at run time.  Deliberately a ROLLED element loop over the slab in scratch
// (one element per lane per trip): it is only reached by odd shapes (N not a multiple of 4, unaligned views, 16-bit residuals,
// sigmoid), and keeping it tiny keeps its registers and code out of the way of the specialised paths below.
template <int FM, int FN>
__device__ __forceinline__ void gemm_epilogue_generic(const GemmParams& p, f32x16 (&acc)[FM][FN], float* scr, int64_t row0, int64_t col0,
                                                      int lane) {
  constexpr int TN = FN * 32;
  constexpr int SLD = TN + 4;
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int i = 0; i < FM; ++i) {
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) scr[((e & 3) + 8 * (e >> 2) + 4 * h) * SLD + j * 32 + r] = acc[i][j][e];
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's scratch writes have landed (scratch is wave-private)
#pragma unroll 1
    for (int t = 0; t < (32 * TN) / 64; ++t) {
      const int idx = t * 64 + lane;
      const int row = idx / TN, col = idx % TN;
      const int64_t m = row0 + i * 32 + row, n = col0 + col;
      if (m < p.M && n < p.N) {
        float x = scr[row * SLD + col] + (p.bias ? p.bias[n] : 0.f);
        if (p.act == 1) x = gelu_erf(x);
        else if (p.act == 2) x = fmaxf(x, 0.f);
        else if (p.act == 3) x = 1.f / (1.f + __expf(-x));
        if (p.colscale) x *= p.colscale[n];
        if (p.res) {
          const int64_t rr = p.res_mod > 0 ? (int64_t)((unsigned)m % (unsigned)p.res_mod) : m;
          x += p.res_is_16bit ? op2f(reinterpret_cast<const op16*>(p.res)[rr * p.ldr + n])
                              : reinterpret_cast<const float*>(p.res)[rr * p.ldr + n];
        }
        if (p.out_is_16bit) reinterpret_cast<op16*>(p.C)[m * p.ldc + n] = f2op(x);
        else reinterpret_cast<float*>(p.C)[m * p.ldc + n] = x;
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // the slab is re-used by the next i
  }
}

// Specialised epilogue for the combinations the hot path uses (C / residual / bias 16-byte aligned, N % 4 == 0, so every lane
// owns whole 4-column chunks): activation, residual kind and output type are template parameters, and each 32-row slab is done
// in three batched phases -- all scratch reads, all residual loads, then math + stores -- so LDS and memory latencies overlap
// instead of being paid per row group.  (The generic version above tests the modes per element: ~12 scalar branches per element,
// measured at 15k cycles per 128x128 tile, more than a 12-step K = 384 main loop; and on gfx9 vmcnt also counts stores, so a
// per-row "wait for the residual" there waits for the previous row's store round trip as well.)
template <int FM, int FN, int ACT, int RES /* 0 none, 1 f32 */, bool OUT16>
__device__ __forceinline__ void gemm_epilogue_spec(const GemmParams& p, f32x16 (&acc)[FM][FN], float* scr, int64_t row0, int64_t col0,
                                                   int lane) {
  constexpr int TN = FN * 32;
  constexpr int SLD = TN + 4;
  constexpr int CPR_ = TN / 4;
  constexpr int CH_ITERS = (32 * CPR_) / 64;
  constexpr int ROWS_PER_IT = 64 / CPR_;
  constexpr int BATCH = CH_ITERS < 4 ? CH_ITERS : 4;
  const int r = lane & 31, h = lane >> 5;
  const int col = (lane % CPR_) * 4;
  const int64_t n = col0 + col;
  const bool n_in = n < p.N;
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f}, cs4 = {1.f, 1.f, 1.f, 1.f};
  if (p.bias && n_in) bias4 = *reinterpret_cast<const f32x4*>(p.bias + n);
  if (p.colscale && n_in) cs4 = *reinterpret_cast<const f32x4*>(p.colscale + n);
  const unsigned res_mod = (unsigned)p.res_mod;
  const int row_in_it = lane / CPR_;
  const float* scr_rd = scr + row_in_it * SLD + col;
  const int64_t m_lane = row0 + row_in_it;
  const float* resp = reinterpret_cast<const float*>(p.res) + n;
  char* cp = reinterpret_cast<char*>(p.C) + n * (OUT16 ? 2 : 4);
  const int64_t ldc_b = p.ldc * (OUT16 ? 2 : 4);
#pragma unroll
  for (int i = 0; i < FM; ++i) {
#ifdef MSAM2_GSTAMP
    if (!(g_epi_mode & 2))
#endif
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) scr[((e & 3) + 8 * (e >> 2) + 4 * h) * SLD + j * 32 + r] = acc[i][j][e];
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's scratch writes have landed (scratch is wave-private)
#pragma unroll
    for (int b0 = 0; b0 < CH_ITERS; b0 += BATCH) {
      f32x4 v[BATCH], rv[BATCH];
      bool live[BATCH];
#pragma unroll
      for (int u = 0; u < BATCH; ++u) v[u] = *reinterpret_cast<const f32x4*>(scr_rd + (b0 + u) * ROWS_PER_IT * SLD);
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int64_t m = m_lane + i * 32 + (b0 + u) * ROWS_PER_IT;
        live[u] = n_in && m < p.M;
        rv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (RES == 1) {
          if (live[u]) {
            const int64_t rr = res_mod > 0 ? (int64_t)((unsigned)m % res_mod) : m;
            rv[u] = *reinterpret_cast<const f32x4*>(resp + rr * p.ldr);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int64_t m = m_lane + i * 32 + (b0 + u) * ROWS_PER_IT;
        f32x4 y;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float x = v[u][q] + bias4[q];
          if constexpr (ACT == 1) x = gelu_erf(x);
          else if constexpr (ACT == 2) x = fmaxf(x, 0.f);
          else if constexpr (ACT == 3) x = __builtin_amdgcn_rcpf(1.f + __expf(-x));
          y[q] = __builtin_fmaf(x, cs4[q], rv[u][q]);
        }
#ifdef MSAM2_GSTAMP
        if (g_epi_mode & 1) live[u] = live[u] && y[0] == 123.456f;
#endif
        if (live[u]) {
          if constexpr (OUT16) {
            op16x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = f2op(y[q]);
            *reinterpret_cast<op16x4*>(cp + m * ldc_b) = o;
          } else {
            *reinterpret_cast<f32x4*>(cp + m * ldc_b) = y;
          }
        }
      }
    }
  }
}

// Direct epilogue: stores straight from the MFMA accumulator layout with buffer_store_dword -- no LDS round trip at all.
//   fp32 out: one accumulator register is two 128-byte row segments per wave instruction (full-rate store shape).
//   16-bit out: adjacent lanes hold adjacent columns, so each even/odd lane pair swaps one value through DPP (quad_perm
//   [1,0,3,2]) and packs a dword: even lanes store (col r, r+1) of row(e), odd lanes (col r-1, r) of row(e+1).
// Rows past M fall outside the buffer descriptor's num_records and are dropped by the hardware range check (the row term is in
// the per-lane voffset because soffset is not range-checked on gfx9); columns past N are masked per lane.
template <int FM, int FN, int ACT, int RES /* 0 none, 1 f32 */, bool OUT16, bool ROPE = false, int TBMAX = 2>
__device__ __forceinline__ void gemm_epilogue_direct(const GemmParams& p, f32x16 (&acc)[FM][FN], int64_t row0, int64_t col0, int lane) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int ES = OUT16 ? 2 : 4;
  const int r = lane & 31, h = lane >> 5, odd = lane & 1;
  const auto c_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)((int64_t)p.M * p.ldc * ES), 0x00020000);
  const auto r_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res), 0, (int)((int64_t)p.M * p.ldr * 4), 0x00020000);
  float bias_j[FN], cs_j[FN];
  bool colok[FN];
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int64_t n = col0 + j * 32 + r;
    colok[j] = n < p.N;
    bias_j[j] = (p.bias && colok[j]) ? p.bias[n] : 0.f;
    cs_j[j] = (p.colscale && colok[j]) ? p.colscale[n] : 1.f;
  }
  auto fn = [&](float x, int j) {
    x += bias_j[j];
    if constexpr (ACT == 1) x = gelu_erf(x);
    else if constexpr (ACT == 2) x = fmaxf(x, 0.f);
    else if constexpr (ACT == 3) x = __builtin_amdgcn_rcpf(1.f + __expf(-x));
    return x * cs_j[j];
  };
  const int ldc_b = (int)p.ldc * ES, ldr_b = (int)p.ldr * 4;
  if constexpr (!OUT16) {
    const int vbase = (int)((row0 + 4 * h) * p.ldc + col0 + r) * 4;
    const int rbase = (int)((row0 + 4 * h) * p.ldr + col0 + r) * 4;
    // Residual loads and output stores share ONE in-order counter (vmcnt): a load issued behind a store waits for that store's round trip.
    // Tile by tile (16 loads, 16 stores, 16 loads, ...) the 2 x 2 tiles of a wave serialised three store round trips inside every
    // epilogue (round 4, found on the patch-embedding kernel: conv.hip); the tiles are now handled TB at a time -- all their residual
    // loads, then all their stores -- TB = 2 (32 extra registers) in the four-workgroups-per-CU kernel, all four tiles (TBMAX = 4) where the
    // register budget allows.
    constexpr int TB = (RES == 1 && FM * FN >= 2 && (FM * FN) % 2 == 0) ? ((TBMAX >= 4 && (FM * FN) % 4 == 0) ? 4 : 2) : 1;
#pragma unroll
    for (int t0 = 0; t0 < FM * FN; t0 += TB) {
      float rv[TB][16];
      if constexpr (RES == 1) {
#pragma unroll
        for (int u = 0; u < TB; ++u) {
          const int i = (t0 + u) / FN, j = (t0 + u) % FN;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int ro = i * 32 + (e & 3) + 8 * (e >> 2);
            rv[u][e] = colok[j] ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rsrc, rbase + ro * ldr_b + j * 128, 0, 0)) : 0.f;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < TB; ++u) {
        const int i = (t0 + u) / FN, j = (t0 + u) % FN;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int ro = i * 32 + (e & 3) + 8 * (e >> 2);
          float y = fn(acc[i][j][e], j);
          if constexpr (RES == 1) y += rv[u][e];
          if (colok[j]) {
            if (p.store_nt) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y), c_rsrc, vbase + ro * ldc_b + j * 128, 0, 2);
            else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y), c_rsrc, vbase + ro * ldc_b + j * 128, 0, 0);
          }
        }
      }
    }
  } else {
    static_assert(!OUT16 || RES == 0, "16-bit outputs carry no residual on the direct path");
    const int vbase = (int)((row0 + 4 * h + odd) * p.ldc + col0 + (r & ~1)) * 2;
    // RoPE bookkeeping: this lane's first row (the one of i = 0, k = 0) as (position in its batch, table row); later rows
    // add a compile-time offset < 128 and wrap
    int l_base = 0, pos_base = 0, hp = 0;
    if constexpr (ROPE) {
      l_base = (int)((unsigned)(row0 + 4 * h + odd) % (unsigned)p.rope_period);
      pos_base = (int)((unsigned)l_base % (unsigned)p.rope_npos);
      hp = p.rope_D >> 1;
    }
    // column side of the rotation, once per lane: channel pair inside its head (one integer division instead of one per output pair)
    int pr_j[ROPE ? FN : 1];
    bool colrot[ROPE ? FN : 1];
    if constexpr (ROPE) {
      int m = (int)((unsigned)((int)col0 + (r & ~1)) % (unsigned)p.rope_D);
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        pr_j[j] = m >> 1;
        colrot[j] = (int)col0 + j * 32 + (r & ~1) < p.rope_cols && colok[j];
        m += 32;
        while (m >= p.rope_D) m -= p.rope_D;
      }
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      // RoPE factors of this 32-row block: ALL of them loaded (unconditionally, index clamped) before its first store.  Loaded where they
      // were used -- one `if (rotates) { c = cos[..]; s = sin[..]; }` per pair of outputs, each followed by a store -- every one of the
      // FN * 8 loads waited for the previous store's round trip (vmcnt is one in-order counter; round 4): the RoPE projections ran at a
      // third of the plain ones' rate.
      float rc[ROPE ? FN : 1][ROPE ? 8 : 1], rs[ROPE ? FN : 1][ROPE ? 8 : 1];
      bool rot[ROPE ? FN : 1][ROPE ? 8 : 1];
      if constexpr (ROPE) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int e0 = 2 * k, ro = i * 32 + (e0 & 3) + 8 * (e0 >> 2);
          int l = l_base + ro, pos = pos_base + ro;
          // (ro < 128 <= period / npos is checked on the host: one wrap suffices)
          if (l >= p.rope_period) { l -= p.rope_period; pos = (int)((unsigned)l % (unsigned)p.rope_npos); }
          else if (pos >= p.rope_npos) pos -= p.rope_npos;
          const bool rowrot = l < p.rope_n;
          const int prow = pos * hp;
#pragma unroll
          for (int j = 0; j < FN; ++j) {
            rot[j][k] = rowrot && colrot[j];
            const int at = rot[j][k] ? prow + pr_j[j] : 0;
            rc[j][k] = p.rope_cos[at];
            rs[j][k] = p.rope_sin[at];
          }
        }
      }
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int e0 = 2 * k, ro = i * 32 + (e0 & 3) + 8 * (e0 >> 2);
          float x0, x1;
          if constexpr (ACT == 1) {                       // GELU on the pair: packed fp32 arithmetic (common.h)
            const f32x2 gp = gelu_erf2(f32x2{acc[i][j][e0] + bias_j[j], acc[i][j][e0 + 1] + bias_j[j]});
            x0 = gp[0] * cs_j[j];
            x1 = gp[1] * cs_j[j];
          } else {
            x0 = fn(acc[i][j][e0], j);
            x1 = fn(acc[i][j][e0 + 1], j);
          }
          if constexpr (!ROPE) {
            // convert first, then trade 16-bit halves: P = (x0 | x1 << 16), N = the neighbour lane's P;
            // even lanes keep (P.lo, N.lo) = row(e0) cols (r, r+1), odd lanes (N.hi, P.hi) = row(e0+1) cols (r-1, r)
            op16x2 own;
            own[0] = f2op(x0);
            own[1] = f2op(x1);
            const unsigned P = __builtin_bit_cast(unsigned, own);
            const unsigned N = (unsigned)__builtin_amdgcn_update_dpp(0, (int)P, 0xB1, 0xf, 0xf, false);
            const unsigned outw = __builtin_amdgcn_perm(N, P, odd ? 0x03020706u : 0x05040100u);
            if (colok[j]) {
              if (p.store_nt) __builtin_amdgcn_raw_buffer_store_b32(outw, c_rsrc, vbase + ro * ldc_b + j * 64, 0, 2);   // nt: stream past the caches
              else __builtin_amdgcn_raw_buffer_store_b32(outw, c_rsrc, vbase + ro * ldc_b + j * 64, 0, 0);
            }
            continue;
          }
          const float send = odd ? x0 : x1;
          const float recv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0xB1, 0xf, 0xf, false));
          float lo = odd ? recv : x0, hi = odd ? x1 : recv;
          if constexpr (ROPE) {
            if (rot[j][k]) {
              const float c = rc[j][k], sn = rs[j][k];
              const float re = lo, im = hi;
              lo = re * c - im * sn;
              hi = re * sn + im * c;
            }
          }
          op16x2 pk;
          pk[0] = f2op(lo);
          pk[1] = f2op(hi);
          if (colok[j]) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pk), c_rsrc, vbase + ro * ldc_b + j * 64, 0, 0);
        }
    }
  }
#endif
}

// >= 0: the epilogue of this problem runs on the direct (no LDS) path in that mode; -1: it needs the LDS scratch
__device__ __forceinline__ int gemm_epilogue_direct_mode(const GemmParams& p) {
  const bool aligned = ((p.N & 3) == 0) && ((p.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0) &&
                       (!p.res || (((p.ldr & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.res) & 15) == 0) && !p.res_is_16bit)) &&
                       (!p.bias || ((reinterpret_cast<uintptr_t>(p.bias) & 15) == 0)) &&
                       (!p.colscale || ((reinterpret_cast<uintptr_t>(p.colscale) & 15) == 0));
  const bool direct = aligned && p.res_mod == 0 && (int64_t)p.M * p.ldc * 4 < (1ll << 31) && (!p.res || (int64_t)p.M * p.ldr * 4 < (1ll << 31));
  const int mode = p.act * 4 + (p.res ? 2 : 0) + (p.out_is_16bit ? 1 : 0);
  return (direct && (mode == 1 || mode == 0 || mode == 2 || mode == 5 || mode == 9)) ? mode : -1;
}

// 2x2 max-pool epilogue (see GemmParams::pool_W): fp32 out [M/4, N] = max over the 4 sub-pixel rows + bias; one dword per lane,
// two 128-byte row segments per store instruction.
template <int FM, int FN>
__device__ __forceinline__ void gemm_epilogue_pool(const GemmParams& p, f32x16 (&acc)[FM][FN], int64_t row0, int64_t col0, int lane) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int r = lane & 31, h = lane >> 5;
  const auto c_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)(((int64_t)p.M >> 2) * p.ldc * 4), 0x00020000);
  const int vbase = (int)(((row0 >> 2) + h) * p.ldc + col0 + r) * 4;
  const int ldc_b = (int)p.ldc * 4;
  float bias_j[FN];                                          // every bias value before the first store (one in-order counter for loads and stores)
#pragma unroll
  for (int j = 0; j < FN; ++j) bias_j[j] = p.bias ? p.bias[min(col0 + j * 32 + r, (int64_t)p.N - 1)] : 0.f;
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int64_t n = col0 + j * 32 + r;
    const bool ok = n < p.N;
    const float b = bias_j[j];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float v = fmaxf(fmaxf(acc[i][j][4 * g], acc[i][j][4 * g + 1]), fmaxf(acc[i][j][4 * g + 2], acc[i][j][4 * g + 3])) + b;
        // rows 8g + 4h .. +3 of the slab -> pooled row (row0 + 32 i)/4 + 2g + h
        if (ok) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), c_rsrc, vbase + (i * 8 + 2 * g) * ldc_b + j * 128, 0, 0);
      }
  }
#endif
}

// q-pool epilogue (see GemmParams::poolq_cols): 16-bit outputs, adjacent lanes pack column pairs through DPP like the direct
// epilogue.  q fragments: max over the 4 sub-pixel registers, then even lanes store pooled row 2g'+h, odd lanes pooled row 2g'+1+h...
// k/v fragments: rows go back to image order, whose address is not affine in the accumulator index (a pooled pixel run may
// wrap to the next image row), so each (i, g) pair of rows resolves its pixel row through gemm_a_row.
template <int FM, int FN>
__device__ __forceinline__ void gemm_epilogue_qpool(const GemmParams& p, f32x16 (&acc)[FM][FN], int64_t row0, int64_t col0, int lane) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int r = lane & 31, h = lane >> 5, odd = lane & 1;
  const auto c_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)((int64_t)p.M * p.ldc * 2), 0x00020000);
  const auto q_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.Q2, 0, (int)(((int64_t)p.M >> 2) * p.ldq * 2), 0x00020000);
  auto swap = [&](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false)); };
  float bias_j[FN];                                          // every bias value before the first store (one in-order counter for loads and stores)
#pragma unroll
  for (int j = 0; j < FN; ++j) bias_j[j] = p.bias ? p.bias[min(col0 + j * 32 + r, (int64_t)p.N - 1)] : 0.f;
  // image-order pixel row of sub-pixel (0, 0) of every pooled pixel this lane holds (pooled pixels row0/4 + h + 8 i + 2 g): ONE pair of
  // 32-bit divisions for the first, the others by stepping through the pooled image -- was gemm_a_row per (j, i, g): FN * 8 * two 64-bit
  // divisions per lane, the larger part of this epilogue's vector work (round 4)
  int top_ig[FM][4];
  {
    const int w2 = p.pool_W >> 1, h2 = p.pool_H >> 1;
    const unsigned pix0 = (unsigned)(row0 >> 2) + (unsigned)h;
    const int x0 = (int)(pix0 % (unsigned)w2);
    const unsigned t0 = pix0 / (unsigned)w2;
    const int y0 = (int)(t0 % (unsigned)h2), b0 = (int)(t0 / (unsigned)h2);
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        int x = x0 + 8 * i + 2 * g, y = y0, bb = b0;
        while (x >= w2) {
          x -= w2;
          if (++y >= h2) { y = 0; ++bb; }
        }
        top_ig[i][g] = (bb * p.pool_H + 2 * y) * p.pool_W + 2 * x;
      }
  }
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int64_t ncol = col0 + j * 32;                      // first column of this fragment (wave-uniform)
    const int64_t n = ncol + r;
    const bool ok = n < p.N;
    const float b = bias_j[j];
    if (ncol < p.poolq_cols) {
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {                     // pooled rows (2 gp, 2 gp + 1) of this half-wave: g = 2 gp, 2 gp + 1
          float x0 = fmaxf(fmaxf(acc[i][j][8 * gp], acc[i][j][8 * gp + 1]), fmaxf(acc[i][j][8 * gp + 2], acc[i][j][8 * gp + 3])) + b;
          float x1 = fmaxf(fmaxf(acc[i][j][8 * gp + 4], acc[i][j][8 * gp + 5]), fmaxf(acc[i][j][8 * gp + 6], acc[i][j][8 * gp + 7])) + b;
          const float recv = swap(odd ? x0 : x1);
          const float lo = odd ? recv : x0, hi = odd ? x1 : recv;
          op16x2 pk;
          pk[0] = f2op(lo);
          pk[1] = f2op(hi);
          // even lanes: g = 2 gp -> pooled row row0/4 + 8 i + 4 gp + h; odd lanes: g = 2 gp + 1 -> + 2
          const int64_t prow = (row0 >> 2) + i * 8 + 4 * gp + 2 * odd + h;
          if (ok) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pk), q_rsrc, (int)(prow * p.ldq + ncol + (r & ~1)) * 2, 0, 0);
        }
    } else {
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          // logical rows row0 + 32 i + 8 g + 4 h + {0,1,2,3} = sub-pixels (dy, dx) of ONE pooled pixel
          const int64_t m = row0 + i * 32 + 8 * g + 4 * h;
          const bool live = m < p.M;
          const int64_t top = top_ig[i][g];                   // pixel row of (dy, dx) = (0, 0); (0,1) = +1, (1,0) = +W, (1,1) = +W+1
#pragma unroll
          for (int dy = 0; dy < 2; ++dy) {
            const float x0 = acc[i][j][4 * g + 2 * dy] + b, x1 = acc[i][j][4 * g + 2 * dy + 1] + b;
            const float recv = swap(odd ? x0 : x1);
            const float lo = odd ? recv : x0, hi = odd ? x1 : recv;
            op16x2 pk;
            pk[0] = f2op(lo);
            pk[1] = f2op(hi);
            const int64_t prow = top + dy * p.pool_W + odd;  // even lanes store sub-pixel dx = 0, odd lanes dx = 1
            if (ok && live) {
              if (p.store_nt) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pk), c_rsrc, (int)(prow * p.ldc + ncol + (r & ~1)) * 2, 0, 2);
              else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pk), c_rsrc, (int)(prow * p.ldc + ncol + (r & ~1)) * 2, 0, 0);
            }
          }
        }
    }
  }
#endif
}

template <int FM, int FN, int TBMAX = 2>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x16 (&acc)[FM][FN], float* scr, int64_t row0, int64_t col0,
                                              int lane) {
  if (p.pool_W) {
    if (p.Q2) gemm_epilogue_qpool<FM, FN>(p, acc, row0, col0, lane);
    else gemm_epilogue_pool<FM, FN>(p, acc, row0, col0, lane);
    return;
  }
  const bool aligned = ((p.N & 3) == 0) && ((p.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0) &&
                       (!p.res || (((p.ldr & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.res) & 15) == 0) && !p.res_is_16bit)) &&
                       (!p.bias || ((reinterpret_cast<uintptr_t>(p.bias) & 15) == 0)) &&
                       (!p.colscale || ((reinterpret_cast<uintptr_t>(p.colscale) & 15) == 0));
  // the direct path addresses C (and the residual) with 32-bit buffer offsets
  const bool direct = aligned && p.res_mod == 0 && (int64_t)p.M * p.ldc * 4 < (1ll << 31) && (!p.res || (int64_t)p.M * p.ldr * 4 < (1ll << 31));
  const int mode = !aligned ? -1 : (p.act * 4 + (p.res ? 2 : 0) + (p.out_is_16bit ? 1 : 0));
  if (direct) {
    switch (mode) {
      case 0 * 4 + 0 + 1:                                                                                  // linear -> 16-bit
        if (p.rope_cos) gemm_epilogue_direct<FM, FN, 0, 0, true, true>(p, acc, row0, col0, lane);
        else gemm_epilogue_direct<FM, FN, 0, 0, true>(p, acc, row0, col0, lane);
        return;
      case 0 * 4 + 0 + 0: gemm_epilogue_direct<FM, FN, 0, 0, false>(p, acc, row0, col0, lane); return;  // linear -> fp32
      case 0 * 4 + 2 + 0: gemm_epilogue_direct<FM, FN, 0, 1, false, false, TBMAX>(p, acc, row0, col0, lane); return;  // + fp32 residual -> fp32
      case 1 * 4 + 0 + 1: gemm_epilogue_direct<FM, FN, 1, 0, true>(p, acc, row0, col0, lane); return;   // GELU -> 16-bit
      case 2 * 4 + 0 + 1: gemm_epilogue_direct<FM, FN, 2, 0, true>(p, acc, row0, col0, lane); return;   // ReLU -> 16-bit
      default: break;
    }
  }
  if (mode == 0 * 4 + 2 + 0) gemm_epilogue_spec<FM, FN, 0, 1, false>(p, acc, scr, row0, col0, lane);   // patch embed: + pos[m % res_mod]
  else if (mode == 0 * 4 + 2 + 1) gemm_epilogue_spec<FM, FN, 0, 1, true>(p, acc, scr, row0, col0, lane);  // decoder image-side k|v|q: + (pe W^T)[m % L] -> 16-bit
  else gemm_epilogue_generic<FM, FN>(p, acc, scr, row0, col0, lane);
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void gemm_kernel(GemmParams p) {
  constexpr int NT = WM * WN * 64;
  constexpr int TM = BM / WM, TN = BN / WN;  // per-wave tile
  constexpr int FM = TM / 32, FN = TN / 32;  // 32x32 MFMA tiles per wave
  constexpr int A_CHUNKS = BM * 4, W_CHUNKS = BN * 4;
  constexpr int A_PER = (A_CHUNKS + NT - 1) / NT, W_PER = (W_CHUNKS + NT - 1) / NT;
  __shared__ __attribute__((aligned(16))) op16 lds[2 * (BM + BN) * GEMM_LDS_STRIDE];
  op16* As = lds;
  op16* Ws = lds + 2 * BM * GEMM_LDS_STRIDE;

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;
  // XCD-aware mapping is not needed for correctness; consecutive blockIdx.x walk N first so that A panels are reused in L2
  const int64_t m0 = (int64_t)blockIdx.y * BM;
  const int64_t n0 = (int64_t)blockIdx.x * BN;

  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  uint4 ra[A_PER], rw[W_PER];
  const int nk_all = (p.K + GEMM_BK - 1) / GEMM_BK;
  const int kt0 = p.ksplit_tiles > 0 ? (int)blockIdx.z * p.ksplit_tiles : 0;
  const int nk = p.ksplit_tiles > 0 ? min(nk_all, kt0 + p.ksplit_tiles) : nk_all;
  if (kt0 >= nk) return;

  // Row bases once (gemm_a_row is two 64-bit divisions for pooled row orders), loads unconditional with clamped addresses and the
  // out-of-range chunks zeroed afterwards: `if (in range) v = load` gave every chunk a basic block and a full vmcnt wait of its own --
  // three exposed latencies per k-tile of a kernel whose reductions are 3-5 tiles long (round 4).
  const op16* a_base[A_PER];
  const op16* w_base[W_PER];
  bool a_in[A_PER], w_in[W_PER];
#pragma unroll
  for (int i = 0; i < A_PER; ++i) {
    const int c = min(tid + i * NT, A_CHUNKS - 1);
    const int row = c >> 2;
    a_in[i] = tid + i * NT < A_CHUNKS && m0 + row < p.M;
    a_base[i] = p.A + gemm_a_row(p, min(m0 + row, (int64_t)p.M - 1)) * p.lda + (c & 3) * 8;
  }
#pragma unroll
  for (int i = 0; i < W_PER; ++i) {
    const int c = min(tid + i * NT, W_CHUNKS - 1);
    const int row = c >> 2;
    w_in[i] = tid + i * NT < W_CHUNKS && n0 + row < p.N;
    w_base[i] = p.W + min(n0 + row, (int64_t)p.N - 1) * p.ldw + (c & 3) * 8;
  }
  auto gload = [&](int kt) {
    const int k0 = kt * GEMM_BK;
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int kc = (min(tid + i * NT, A_CHUNKS - 1) & 3) * 8;
      const bool kin = k0 + kc < p.K;                       // (K % 8 == 0: a chunk is inside or outside as a whole)
      const uint4 v = *reinterpret_cast<const uint4*>(a_base[i] + (kin ? k0 : 0));
      ra[i] = (a_in[i] && kin) ? v : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < W_PER; ++i) {
      const int kc = (min(tid + i * NT, W_CHUNKS - 1) & 3) * 8;
      const bool kin = k0 + kc < p.K;
      const uint4 v = *reinterpret_cast<const uint4*>(w_base[i] + (kin ? k0 : 0));
      rw[i] = (w_in[i] && kin) ? v : make_uint4(0, 0, 0, 0);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int c = tid + i * NT;
      if (c < A_CHUNKS)
        *reinterpret_cast<uint4*>(As + (buf * BM + (c >> 2)) * GEMM_LDS_STRIDE + (c & 3) * 8) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < W_PER; ++i) {
      const int c = tid + i * NT;
      if (c < W_CHUNKS)
        *reinterpret_cast<uint4*>(Ws + (buf * BN + (c >> 2)) * GEMM_LDS_STRIDE + (c & 3) * 8) = rw[i];
    }
  };

  gload(kt0);
  lstore(0);
  __syncthreads();
  int cur = 0;
  for (int kt = kt0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      op16x8 af[FM], bfr[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i)
        af[i] = *reinterpret_cast<const op16x8*>(As + (cur * BM + wm * TM + i * 32 + r) * GEMM_LDS_STRIDE + ks * 16 + h * 8);
#pragma unroll
      for (int j = 0; j < FN; ++j)
        bfr[j] = *reinterpret_cast<const op16x8*>(Ws + (cur * BN + wn * TN + j * 32 + r) * GEMM_LDS_STRIDE + ks * 16 + h * 8);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = MSAM2_MFMA_32x32x16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) lstore(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue through wave-private LDS scratch (the operand LDS is free after the barrier)
  __syncthreads();
  static_assert(WM * WN * 32 * (TN + 4) * 4 <= 2 * (BM + BN) * GEMM_LDS_STRIDE * 2, "epilogue scratch must fit the operand LDS");
  if (p.ksplit_tiles > 0) {
    // split-K: fp32 atomics straight from the accumulator layout (lane = column, register = row)
    float* C = reinterpret_cast<float*>(p.C);
    const bool first = blockIdx.z == 0;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int64_t n = n0 + wn * TN + j * 32 + r;
      if (n >= p.N) continue;
      const float bv = (first && p.bias) ? p.bias[n] : 0.f;
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int64_t m = m0 + wm * TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (m < p.M) {
            float val = acc[i][j][e] + bv;
            if (first && p.res) val += reinterpret_cast<const float*>(p.res)[m * p.ldr + n];
            atomicAdd(C + m * p.ldc + n, val);
          }
        }
    }
    return;
  }
  gemm_epilogue<FM, FN>(p, acc, reinterpret_cast<float*>(lds) + wave * 32 * (TN + 4), m0 + wm * TM, n0 + wn * TN, lane);
}

// ------------------------------------------------------------------------------------------------------------------
// Large-shape variant: 128x128x64 tiles, operands streamed global -> LDS by LDS-DMA (global_load_lds_dwordx4, no staging
// registers), 2-stage ring per workgroup and 2 workgroups per CU so that a tile is always in flight behind the MFMA phase.
// LDS image per operand: [128 rows][64 k] op16 with 128-byte rows; the 16-byte chunk c of row r sits in slot
// c ^ ((r >> 1) & 7), which makes every ds_read_b128 fragment read (16-lane groups of the 32x32x16 operand map) conflict
// free.  The DMA writes LDS linearly (wave base + lane*16), so the swizzle is applied to the per-lane SOURCE address.
// Requires K % 64 == 0; rows beyond M / N are clamped on load and dropped in the epilogue.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void gemm_glds_kernel(GemmParams p) {
  constexpr int BM = 128, BN = 128, BK = 64, STAGE_BYTES = (BM + BN) * BK * 2;  // 32 KiB
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE_BYTES];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  // XCD-aware tile order: workgroups that share an XCD (ids congruent mod 8) walk the N tiles of the same M panel
  const int n_tiles_n = (p.N + BN - 1) / BN, n_tiles_m = (p.M + BM - 1) / BM;
  const int nwg = n_tiles_n * n_tiles_m;
  int bid = blockIdx.x;
  {
    const int q = nwg / 8, rem = nwg % 8, xcd = bid % 8;
    bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + bid / 8;
  }
  const int64_t m0 = (int64_t)(bid / n_tiles_n) * BM;
  const int64_t n0 = (int64_t)(bid % n_tiles_n) * BN;

  // this wave's 4 A pieces and 4 W pieces (a piece = 8 rows x 128 B = one DMA instruction): buffer descriptors in SGPRs,
  // loop-invariant 32-bit per-lane byte offsets, scalar k offset -> no 64-bit per-lane pointer arithmetic in the loop
  const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7fffffff, 0x00020000);
  const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0x7fffffff, 0x00020000);
  unsigned offsA[4], offsW[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    const int64_t ga = gemm_a_row(p, min(m0 + row, (int64_t)p.M - 1)), gw = min(n0 + row, (int64_t)p.N - 1);
    offsA[i] = (unsigned)(ga * p.lda * 2 + chunk * 16);
    offsW[i] = (unsigned)(gw * p.ldw * 2 + chunk * 16);
  }
  auto issue = [&](int kt, int stage) {
    unsigned char* base = lds + stage * STAGE_BYTES + wave * 4096;
    const unsigned so = (unsigned)kt * BK * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (__attribute__((address_space(3))) void*)(base + i * 1024), 16, offsA[i], so, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void*)(base + BM * BK * 2 + i * 1024), 16, offsW[i], so, 0, 0);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fragment read offsets (bytes) inside a stage, without the k-substep term
  int offA[2], offB[2], swz[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ra = wm * 64 + i * 32 + r, rb = wn * 64 + i * 32 + r;
    offA[i] = ra * 128;
    offB[i] = BM * BK * 2 + rb * 128;
    swz[0][i] = (ra >> 1) & 7;
    swz[1][i] = (rb >> 1) & 7;
  }

  const int nk = p.K / BK;
  // one DMA piece of tile kt (i < 4: A piece i, else W piece i-4)
  auto issue_piece = [&](int kt, int stage, int i) {
    unsigned char* base = lds + stage * STAGE_BYTES + wave * 4096;
    const unsigned so = (unsigned)kt * BK * 2;
    if (i < 4)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (__attribute__((address_space(3))) void*)(base + i * 1024), 16, offsA[i], so, 0, 0);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void*)(base + BM * BK * 2 + (i - 4) * 1024), 16,
                                               offsW[i - 4], so, 0, 0);
  };
  issue(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int st = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's DMA pieces of tile kt have landed
    __builtin_amdgcn_s_barrier();                        // ... and so have every other wave's; stage st^1 is free again
    const unsigned char* sb = lds + st * STAGE_BYTES;
    // all 16 operand fragments of the tile in one burst (64 VGPRs), then 16 MFMAs with the next tile's DMA pieces issued in
    // their shadow (2 per k-substep): LDS latency is paid once per tile and the DMA issue cost hides under the matrix pipe
    op16x8 af[4][2], bfr[4][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int c = 2 * ks + h;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[ks][i] = *reinterpret_cast<const op16x8*>(sb + offA[i] + ((c ^ swz[0][i]) << 4));
        bfr[ks][i] = *reinterpret_cast<const op16x8*>(sb + offB[i] + ((c ^ swz[1][i]) << 4));
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = MSAM2_MFMA_32x32x16(af[ks][i], bfr[ks][j], acc[i][j], 0, 0, 0);
      if (kt + 1 < nk) {
        issue_piece(kt + 1, st ^ 1, 2 * ks);
        issue_piece(kt + 1, st ^ 1, 2 * ks + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __builtin_amdgcn_s_barrier();
  gemm_epilogue<2, 2, 4>(p, acc, reinterpret_cast<float*>(lds) + wave * 32 * 68, m0 + wm * 64, n0 + wn * 64, lane);
}

// ------------------------------------------------------------------------------------------------------------------
// 4-stage variant: 128x128x32 tiles, LDS-DMA ring of four 16 KiB stages per workgroup (prefetch distance 3), 2 workgroups
// per CU, ONE barrier per k-step.  The deeper ring is what hides the L2 -> LDS latency (~1.5-2k cycles under load) that the
// 2-stage kernel above exposes every step; K only needs to be a multiple of 32, which also brings the K = 96 / 160 layers
// of Hiera stage 1 onto the DMA path.
// LDS image per operand: [128 rows][32 k] op16 = 64-byte rows; chunk c (0..3) of row r sits in slot c ^ ((r >> 2) & 3):
// every 16-lane ds_read_b128 group of the 32x32x16 operand map then covers 16 distinct (row mod 4, slot) positions of the
// 256-byte bank rows it touches (conflict free).
// ------------------------------------------------------------------------------------------------------------------
template <int NST, int OCC>
__global__ __launch_bounds__(256, OCC) void gemm_glds32_kernel(GemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 128, BN = 128, BK = 32, STAGE_BYTES = (BM + BN) * BK * 2;  // 16 KiB
  __shared__ __attribute__((aligned(1024))) unsigned char lds[(NST * STAGE_BYTES > 34816) ? NST * STAGE_BYTES : 34816];
  GSTAMP(0);
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  const int n_tiles_n = (p.N + BN - 1) / BN, n_tiles_m = (p.M + BM - 1) / BM;
  const int nwg = n_tiles_n * n_tiles_m;
  int bid = blockIdx.x;
  {
    const int q = nwg / 8, rem = nwg % 8, xcd = bid % 8;
    bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + bid / 8;
  }
  const int64_t m0 = (int64_t)(bid / n_tiles_n) * BM;
  const int64_t n0 = (int64_t)(bid % n_tiles_n) * BN;

  // a DMA piece = 16 rows x 64 B; the A tile has 8 pieces, so has the W tile: 2 + 2 per wave
  const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7fffffff, 0x00020000);
  const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0x7fffffff, 0x00020000);
  unsigned offsA[2], offsW[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wave * 2 + i) * 16 + (lane >> 2);
    const int chunk = (lane & 3) ^ ((row >> 2) & 3);
    const int64_t ga = gemm_a_row(p, min(m0 + row, (int64_t)p.M - 1)), gw = min(n0 + row, (int64_t)p.N - 1);
    offsA[i] = (unsigned)(ga * p.lda * 2 + chunk * 16);
    offsW[i] = (unsigned)(gw * p.ldw * 2 + chunk * 16);
  }
  auto issue = [&](int kt) {
    unsigned char* base = lds + (kt % NST) * STAGE_BYTES + wave * 2048;
    const unsigned so = (unsigned)kt * BK * 2;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (__attribute__((address_space(3))) void*)(base + i * 1024), 16, offsA[i], so, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void*)(base + BM * BK * 2 + i * 1024), 16, offsW[i], so, 0, 0);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  int offA[2], offB[2], swz[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ra = wm * 64 + i * 32 + r, rb = wn * 64 + i * 32 + r;
    offA[i] = ra * 64;
    offB[i] = BM * BK * 2 + rb * 64;
    swz[0][i] = (ra >> 2) & 3;
    swz[1][i] = (rb >> 2) & 3;
  }

  const int nk = p.K / BK;
  issue(0);
#pragma unroll
  for (int i = 1; i < NST - 1; ++i)
    if (nk > i) issue(i);
  for (int kt = 0; kt < nk; ++kt) {
#ifdef MSAM2_GSTAMP
    if (kt == 1) GSTAMP(1);
#endif
    const int ahead = min(nk - 1 - kt, NST - 2);         // DMA groups younger than tile kt still allowed in flight
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                        // tile kt visible to all waves; the stage of tile kt-1 is no longer read
    if (kt + NST - 1 < nk) issue(kt + NST - 1);
    const unsigned char* sb = lds + (kt % NST) * STAGE_BYTES;
    op16x8 af[2][2], bfr[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = 2 * ks + h;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[ks][i] = *reinterpret_cast<const op16x8*>(sb + offA[i] + ((c ^ swz[0][i]) << 4));
        bfr[ks][i] = *reinterpret_cast<const op16x8*>(sb + offB[i] + ((c ^ swz[1][i]) << 4));
      }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = MSAM2_MFMA_32x32x16(af[ks][i], bfr[ks][j], acc[i][j], 0, 0, 0);
  }
  GSTAMP(2);
  __builtin_amdgcn_s_barrier();
  GSTAMP_NW(6);
  gemm_epilogue<2, 2>(p, acc, reinterpret_cast<float*>(lds) + wave * 32 * 68, m0 + wm * 64, n0 + wn * 64, lane);
  GSTAMP_NW(7);
  GSTAMP(3);
#ifdef MSAM2_GSTAMP
  if (threadIdx.x == 0 && blockIdx.x < 8192) { unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); g_gstamp[blockIdx.x * 8 + 4] = hw; unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); g_gstamp[blockIdx.x * 8 + 5] = xcc; }
#endif
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// Wide-tile DMA variant: BM x BN x 32 tiles (256x128 / 256x256 ...), 4 waves in a 2x2 grid, each owning (BM/2) x (BN/2)
// (FM x FN 32x32 accumulators), NST-stage LDS-DMA ring with prefetch distance NST-1 and one barrier per k-step.
// Why: what bounds the 128x128 kernels is bytes in flight -- at 64 flop per DMA byte a CU needs ~64 B/clk from L2, i.e. more
// than 100 KiB in flight to cover the ~2k-cycle L2->LDS latency.  A 256x128 tile needs 1.33x fewer bytes per flop, twice the
// MFMA work per barrier, and with 3 stages keeps two tiles in flight per workgroup (2 workgroups per CU: 144 KiB of LDS).
// Same 64-byte-row image and chunk swizzle (c ^ ((r >> 2) & 3)) as gemm_glds32_kernel.
// ------------------------------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int BM, int BN, int NST, int OCC>
__global__ __launch_bounds__(256, OCC) void gemm_wide_kernel(GemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BK = 32, A_BYTES = BM * BK * 2, STAGE_BYTES = (BM + BN) * BK * 2;
  constexpr int FM = BM / 64, FN = BN / 64;         // 32x32 accumulators per wave
  constexpr int PA = BM / 64, PWW = BN / 64;        // DMA pieces (16 rows x 64 B) per wave per tile
  constexpr int G = PA + PWW;                       // DMA instructions per wave per k-step
  constexpr int SCR = 4 * 32 * (FN * 32 + 4) * 4;   // epilogue scratch
  static_assert(G * (NST - 1) <= 63, "vmcnt is a 6-bit counter");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[(NST * STAGE_BYTES > SCR) ? NST * STAGE_BYTES : SCR];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  const int n_tiles_n = (p.N + BN - 1) / BN, n_tiles_m = (p.M + BM - 1) / BM;
  const int nwg = n_tiles_n * n_tiles_m;
  int bid = blockIdx.x;
  {
    const int q = nwg / 8, rem = nwg % 8, xcd = bid % 8;
    bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + bid / 8;
  }
  const int64_t m0 = (int64_t)(bid / n_tiles_n) * BM;
  const int64_t n0 = (int64_t)(bid % n_tiles_n) * BN;

  const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7fffffff, 0x00020000);
  const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0x7fffffff, 0x00020000);
  unsigned offsA[PA], offsW[PWW];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int row = (wave * PA + i) * 16 + (lane >> 2);
    const int chunk = (lane & 3) ^ ((row >> 2) & 3);
    offsA[i] = (unsigned)(gemm_a_row(p, min(m0 + row, (int64_t)p.M - 1)) * p.lda * 2 + chunk * 16);
  }
#pragma unroll
  for (int i = 0; i < PWW; ++i) {
    const int row = (wave * PWW + i) * 16 + (lane >> 2);
    const int chunk = (lane & 3) ^ ((row >> 2) & 3);
    offsW[i] = (unsigned)(min(n0 + row, (int64_t)p.N - 1) * p.ldw * 2 + chunk * 16);
  }
  auto issue = [&](int kt) {
    unsigned char* base = lds + (kt % NST) * STAGE_BYTES;
    const unsigned so = (unsigned)kt * BK * 2;
#pragma unroll
    for (int i = 0; i < PA; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (__attribute__((address_space(3))) void*)(base + (wave * PA + i) * 1024), 16,
                                               offsA[i], so, 0, 0);
#pragma unroll
    for (int i = 0; i < PWW; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void*)(base + A_BYTES + (wave * PWW + i) * 1024),
                                               16, offsW[i], so, 0, 0);
  };

  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fragment byte offsets inside a stage for k-substep 0 / 1 (chunk 2*ks + h, swizzled by the row)
  int offA[FM][2], offB[FN][2];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int ra = wm * (BM / 2) + i * 32 + r;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) offA[i][ks] = ra * 64 + (((2 * ks + h) ^ ((ra >> 2) & 3)) << 4);
  }
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int rb = wn * (BN / 2) + j * 32 + r;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) offB[j][ks] = A_BYTES + rb * 64 + (((2 * ks + h) ^ ((rb >> 2) & 3)) << 4);
  }

  const int nk = p.K / BK;
  auto wait_groups = [&](int younger) {                   // wait until at most `younger` DMA groups of this wave are in flight
    if (NST >= 4 && younger >= 3) wait_vmcnt<(NST >= 4 ? 3 : 0) * G>();
    else if (NST >= 3 && younger == 2) wait_vmcnt<(NST >= 3 ? 2 : 0) * G>();
    else if (NST >= 2 && younger == 1) wait_vmcnt<G>();
    else wait_vmcnt<0>();
  };
  auto load_frags = [&](const unsigned char* sb, int ks, op16x8 (&af)[FM], op16x8 (&bfr)[FN]) {
#pragma unroll
    for (int i = 0; i < FM; ++i) af[i] = *reinterpret_cast<const op16x8*>(sb + offA[i][ks]);
#pragma unroll
    for (int j = 0; j < FN; ++j) bfr[j] = *reinterpret_cast<const op16x8*>(sb + offB[j][ks]);
  };
  auto mma = [&](const op16x8 (&af)[FM], const op16x8 (&bfr)[FN]) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = MSAM2_MFMA_32x32x16(af[i], bfr[j], acc[i][j], 0, 0, 0);
  };
  // Software pipeline across the barrier: the fragments of (tile kt+1, k-substep 0) are fetched right after the barrier that
  // publishes tile kt+1, in the shadow of tile kt's second-substep MFMAs -- the matrix pipe never waits for an LDS round trip.
  // The same barrier says every wave has taken its last fragments of tile kt, so tile kt+NST is streamed into kt's stage.
#pragma unroll
  for (int i = 0; i < NST; ++i)
    if (nk > i) issue(i);
  wait_groups(min(nk, NST) - 1);
  __builtin_amdgcn_s_barrier();
  op16x8 a0[FM], b0[FN], a1[FM], b1[FN];
  load_frags(lds, 0, a0, b0);
  for (int kt = 0; kt < nk; ++kt) {
    const unsigned char* sb = lds + (kt % NST) * STAGE_BYTES;
    load_frags(sb, 1, a1, b1);
    mma(a0, b0);
    if (kt + 1 < nk) {
      wait_groups(min(nk - 2 - kt, NST - 2));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (kt + NST < nk) issue(kt + NST);
      load_frags(lds + ((kt + 1) % NST) * STAGE_BYTES, 0, a0, b0);
    }
    mma(a1, b1);
  }
  __builtin_amdgcn_s_barrier();
  gemm_epilogue<FM, FN>(p, acc, reinterpret_cast<float*>(lds) + wave * 32 * (FN * 32 + 4), m0 + wm * (BM / 2), n0 + wn * (BN / 2), lane);
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// W-STATIONARY persistent variant for short reductions (K = 256 / 384: the qkv / fc1 projections of Hiera stage 3 and the
// memory attention's linear1 -- 16384 token rows against a weight of a few hundred KB; round 3).
//
// Why.  With K = 384 a 128x128 output tile is only 6 k-steps of 64: the tiled kernels above spend their life in the prologue (first
// DMA), the epilogue and at 2-stage-ring barriers (PMC of gemm_glds_kernel at 16384 x 1536 x 384: waves wait 50 % of their cycles,
// MFMA-busy 19 %; profiles/r03_gemm_*), and every tile re-streams its 96 KB W panel from L2.  Here a workgroup KEEPS its 128-row W
// panel (all of K: NK x 16 KB) in LDS for its whole life and walks a strided list of M tiles, streaming only A through a 4-stage
// LDS-DMA ring that runs on ACROSS tile boundaries (prefetch distance 3 chunks = ~1.5k MFMA cycles), so there is one prologue per
// workgroup instead of one per tile and half the L2 -> LDS traffic (A: N/128 re-reads, W: once per workgroup).
// The epilogue of tile t (bias, activation, 16-bit conversion, 32 dword stores per wave straight from the accumulator layout) is
// DEFERRED: the finished accumulators move to a second register set and one 32x32 block of them is retired inside each of the next
// tile's first four k-chunks, between that chunk's MFMAs -- with one wave per SIMD nothing else could hide the GELU's VALU work.
// vmcnt counts loads, LDS-DMA and stores together in issue order (MI355X_MICROARCH.md), so every wait is COUNTED: the ops younger than
// chunk g's four DMA pieces are the pieces of chunks g+1, g+2 and the stores of the epilogue slices issued since.
// One workgroup (4 waves, one per SIMD) per CU: 160 KB of LDS at K = 384.  Requires M % 128 == 0, N % 128 == 0, 16-bit output,
// no residual / column scale / RoPE / pooling; grid = groups x panels <= 256.
// ------------------------------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void wstat_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wstat_wait_dyn(int n) {      // n: multiple of 4, 0..40 (wave-uniform)
  switch (n >> 2) {
    case 0: wstat_wait<0>(); break;
    case 1: wstat_wait<4>(); break;
    case 2: wstat_wait<8>(); break;
    case 3: wstat_wait<12>(); break;
    case 4: wstat_wait<16>(); break;
    case 5: wstat_wait<20>(); break;
    case 6: wstat_wait<24>(); break;
    case 7: wstat_wait<28>(); break;
    case 8: wstat_wait<32>(); break;
    case 9: wstat_wait<36>(); break;
    default: wstat_wait<40>(); break;
  }
}

template <int NK, int ACT, bool NT, bool APF>
__global__ __launch_bounds__(512, 1) void gemm_wstat_kernel(GemmParams p, int n_panels, int groups) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 128, BN = 128, BK = 64, NST = 4;
  constexpr int KT_BYTES = BN * BK * 2;             // one k-tile of the W image / one A stage: 16 KiB
  constexpr int W_BYTES = NK * KT_BYTES;
  constexpr int LPC = 2;                            // DMA pieces per wave and chunk (16 pieces of 1 KiB over 8 waves)
  constexpr int SPR = 8;                            // stores per retired 32x32 block
  static_assert(NK >= 4, "the ring's prefetch distance (3 chunks) must stay inside one tile for the wait arithmetic below");
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];   // [W image: NK k-tiles | A ring: NST stages]
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;          // 8 waves (two per SIMD): 4 row slabs of 32 x 2 column halves of 64
  const int r = lane & 31, h = lane >> 5, odd = lane & 1;
  // XCD-aware placement (grid = 256 workgroups, dealt round-robin over the 8 XCDs: id % 8 share an XCD and its private L2).  The
  // n_panels workgroups of a GROUP stream the same A tiles at the same time, so a group sits on ONE XCD wherever it fits: each XCD's 32
  // slots hold F = 32 / n_panels whole groups; the R left-over slots per XCD are pooled (XCD-major) into further groups that span two
  // XCDs.  With id-order placement every A tile was fetched by all 8 L2s: ~100 MB through the fabric for a 12.6 MB operand.
  int panel, grp;
  {
    const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
    const int F = 32 / n_panels, R = 32 - F * n_panels;
    if (l < F * n_panels) {
      grp = xcd * F + l / n_panels;
      panel = l % n_panels;
    } else {
      const int q = (l - F * n_panels) + R * xcd;
      grp = 8 * F + q / n_panels;
      panel = q % n_panels;
    }
  }
  const int n_tiles_m = p.M / BM;
  const int my_tiles = grp < min(groups, n_tiles_m) ? (n_tiles_m - grp + groups - 1) / groups : 0;     // tiles grp, grp + groups, ...
  if (my_tiles == 0) return;
  const int64_t n0 = (int64_t)panel * BN;
  const int total = my_tiles * NK;

  const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7fffffff, 0x00020000);
  const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0x7fffffff, 0x00020000);
  // a piece = 8 rows x 128 B = one DMA instruction; 128-byte-row image, chunk c of row r in slot c ^ ((r >> 1) & 7) (as gemm_glds_kernel)
  unsigned offsA[LPC], offsW[LPC];
#pragma unroll
  for (int i = 0; i < LPC; ++i) {
    const int row = (wave * LPC + i) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    offsA[i] = (unsigned)((int64_t)row * p.lda * 2 + chunk * 16);
    offsW[i] = (unsigned)((n0 + row) * p.ldw * 2 + chunk * 16);
  }
  unsigned char* const ring = lds + W_BYTES;
  auto issue_a = [&](int g) {
    const int ti = g / NK, kt = g - ti * NK;
    const unsigned so = (unsigned)(((int64_t)(grp + ti * groups) * BM * p.lda + (int64_t)kt * BK) * 2);
    unsigned char* base = ring + (g & (NST - 1)) * KT_BYTES + wave * (LPC * 1024);
#pragma unroll
    for (int i = 0; i < LPC; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (__attribute__((address_space(3))) void*)(base + i * 1024), 16, offsA[i], so, 0, 0);
  };
  // this workgroup's columns never change: bias once, BEFORE any DMA is in flight (the compiler waits for these loads with vmcnt(0))
  float bias_j[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) bias_j[j] = p.bias ? p.bias[n0 + wn * 64 + j * 32 + r] : 0.f;
  // the values pass through an empty asm: the compiler waits for the two loads HERE (nothing else is in flight yet) and afterwards
  // treats them as plain registers -- otherwise it puts its own conservative s_waitcnt vmcnt(N) in front of their first use, in the
  // middle of the first retiring chunk, where it drains DMA pieces that are meant to stay in flight
  asm volatile("" : "+v"(bias_j[0]), "+v"(bias_j[1])::"memory");
  // ---- prologue: the whole W panel, then the first three A chunks (W is older than chunk 0: waiting for chunk 0 covers it)
#pragma unroll
  for (int kt = 0; kt < NK; ++kt)
#pragma unroll
    for (int i = 0; i < LPC; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void*)(lds + kt * KT_BYTES + (wave * LPC + i) * 1024), 16,
                                               offsW[i], (unsigned)kt * BK * 2, 0, 0);
#pragma unroll
  for (int g = 0; g < (APF ? NST : NST - 1); ++g)
    if (g < total) issue_a(g);

  // fragment read offsets inside a 16 KiB k-tile image, without the k-substep term
  const int ra = wm * 32 + r;
  const int offA = ra * 128, swzA = (ra >> 1) & 7;
  int offB[2], swzB[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int rb = wn * 64 + j * 32 + r;
    offB[j] = rb * 128;
    swzB[j] = (rb >> 1) & 7;
  }
  const auto c_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)((int64_t)p.M * p.ldc * 2), 0x00020000);
  const int ldc_b = (int)p.ldc * 2;
  const int vcol = (int)(n0 + wn * 64 + (r & ~1)) * 2;

  // one 32x32 block of a finished tile: bias -> activation -> 16-bit pairs through DPP -> 8 dword stores (gemm_epilogue_direct's
  // 16-bit form; rows past M cannot occur: M % 128 == 0)
  auto retire = [&](const f32x16& blk, int j, int row0) {
    const int vbase = (row0 + wm * 32 + 4 * h + odd) * ldc_b + vcol;
#pragma unroll
    for (int k = 0; k < SPR; ++k) {
      const int e0 = 2 * k, ro = (e0 & 3) + 8 * (e0 >> 2);
      float x0 = blk[e0] + bias_j[j], x1 = blk[e0 + 1] + bias_j[j];
      if constexpr (ACT == 1) {
        const f32x2 gp = gelu_erf2(f32x2{x0, x1});
        x0 = gp[0];
        x1 = gp[1];
      } else if constexpr (ACT == 2) { x0 = fmaxf(x0, 0.f); x1 = fmaxf(x1, 0.f); }
      op16x2 own;
      own[0] = f2op(x0);
      own[1] = f2op(x1);
      const unsigned P = __builtin_bit_cast(unsigned, own);
      const unsigned Nb = (unsigned)__builtin_amdgcn_update_dpp(0, (int)P, 0xB1, 0xf, 0xf, false);
      const unsigned outw = __builtin_amdgcn_perm(Nb, P, odd ? 0x03020706u : 0x05040100u);
      __builtin_amdgcn_raw_buffer_store_b32(outw, c_rsrc, vbase + ro * ldc_b + j * 64, 0, NT ? 2 : 0);
    }
  };

  // One tile: NK chunks, fully unrolled (kt is a compile-time constant).  `cur` accumulates; `prev` holds the PREVIOUS tile's finished
  // accumulators, whose two 32x32 blocks are retired in chunks 0 and 1 -- the two accumulator sets swap roles from tile to tile, so
  // nothing is ever copied.  Counted wait before chunk g: the ops younger than chunk g's LPC pieces are the pieces of chunks g+1, g+2
  // and the SPR stores of every retire issued in iterations g-3 .. g-1 (kt 0 and 1 of a tile that has a predecessor).
  // W fragments of k-tile kt (static image): 8 reads, issued a whole chunk ahead of the MFMAs that use them
  auto load_b = [&](int kt, op16x8 (&bf)[4][2]) __attribute__((always_inline)) {
    const unsigned char* sw = lds + kt * KT_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[ks][j] = *reinterpret_cast<const op16x8*>(sw + offB[j] + (((2 * ks + h) ^ swzB[j]) << 4));
  };
  auto load_a = [&](int stage, op16x8 (&af)[4]) __attribute__((always_inline)) {
    const unsigned char* sa = ring + stage * KT_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) af[ks] = *reinterpret_cast<const op16x8*>(sa + offA + (((2 * ks + h) ^ swzA) << 4));
  };
  op16x8 afr[2][4];                                   // (APF) A fragments, double buffered like the W fragments
  op16x8 bfr[2][4][2];                                // double buffer: chunk kt uses [kt & 1] (NK is even), loads [(kt + 1) & 1]
  static_assert(NK % 2 == 0, "the W-fragment double buffer alternates with the chunk index");
  auto tile = [&](f32x16 (&cur)[2], f32x16 (&prev)[2], int ti, bool has_prev, int prev_row0) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) cur[j][e] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NK; ++kt) {
      const int g = ti * NK + kt;
      // retire stores among the three previous iterations (chunks 0 / 1 of a tile that has a predecessor)
      int st = 0;
#pragma unroll
      for (int d = 1; d <= 3; ++d) {
        const int k2 = kt - d;                       // >= 0: this tile (retires iff has_prev); < 0: previous tile's chunk NK + k2
        if (k2 >= 0) st += (k2 < 2 && has_prev) ? SPR : 0;
        else st += ((NK + k2) < 2 && ti >= 2) ? SPR : 0;
      }
      if constexpr (!APF) {
        const int rem = total - 1 - g;
        if (rem >= 2) {
          if (st == 0) wstat_wait<2 * LPC>();
          else if (st == SPR) wstat_wait<2 * LPC + SPR>();
          else wstat_wait<2 * LPC + 2 * SPR>();
        } else if (rem == 1) {
          if (st == 0) wstat_wait<LPC>();
          else if (st == SPR) wstat_wait<LPC + SPR>();
          else wstat_wait<LPC + 2 * SPR>();
        } else {
          if (st == 0) wstat_wait<0>();
          else if (st == SPR) wstat_wait<SPR>();
          else wstat_wait<2 * SPR>();
        }
        __builtin_amdgcn_s_barrier();                   // chunk g is complete for every wave; stage (g-1) % NST is free again
        if (g + NST - 1 < total) issue_a(g + NST - 1);
        if (g == 0) load_b(0, bfr[0]);                  // the W image landed with chunk 0 (it is older): first fragments
        const unsigned char* sa = ring + (g & (NST - 1)) * KT_BYTES;
        op16x8 af[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) af[ks] = *reinterpret_cast<const op16x8*>(sa + offA + (((2 * ks + h) ^ swzA) << 4));
        // next chunk's W fragments: in flight under this chunk's MFMAs (the phases of the eight waves are locked by the barrier, so
        // without this every wave reads LDS, then every wave issues MFMAs)
        load_b((kt + 1) % NK, bfr[(kt + 1) & 1]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int j = 0; j < 2; ++j) cur[j] = MSAM2_MFMA_32x32x16(af[ks], bfr[kt & 1][ks][j], cur[j], 0, 0, 0);
      } else {
        // A fragments one chunk ahead as well: this iteration waits for chunk g+1, reads its fragments under chunk g's MFMAs and streams
        // chunk g+4 into the stage chunk g occupied (its fragments went to registers one iteration ago)
        if (g + 1 < total) {
          const int rem = total - 2 - g;                // chunks issued beyond g+1 (at most g+2, g+3)
          if (rem >= 2) {
            if (st == 0) wstat_wait<2 * LPC>();
            else if (st == SPR) wstat_wait<2 * LPC + SPR>();
            else wstat_wait<2 * LPC + 2 * SPR>();
          } else if (rem == 1) {
            if (st == 0) wstat_wait<LPC>();
            else if (st == SPR) wstat_wait<LPC + SPR>();
            else wstat_wait<LPC + 2 * SPR>();
          } else {
            if (st == 0) wstat_wait<0>();
            else if (st == SPR) wstat_wait<SPR>();
            else wstat_wait<2 * SPR>();
          }
        }
        __builtin_amdgcn_s_barrier();
        if (g + NST < total) issue_a(g + NST);
        if (g + 1 < total) load_a((g + 1) & (NST - 1), afr[(kt + 1) & 1]);
        load_b((kt + 1) % NK, bfr[(kt + 1) & 1]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int j = 0; j < 2; ++j) cur[j] = MSAM2_MFMA_32x32x16(afr[kt & 1][ks], bfr[kt & 1][ks][j], cur[j], 0, 0, 0);
      }
      if (has_prev && kt < 2) retire(prev[kt], kt, prev_row0);
    }
  };

  if constexpr (APF) {
    wstat_wait_dyn(LPC * min(total - 1, NST - 1));    // chunk 0 (and the W panel before it) has landed; chunks 1..3 may still fly
    __builtin_amdgcn_s_barrier();
    load_a(0, afr[0]);
    load_b(0, bfr[0]);
  }
  f32x16 accA[2], accB[2];
  int row_prev = 0;
  for (int ti = 0; ti < my_tiles; ti += 2) {
    tile(accA, accB, ti, ti > 0, row_prev);
    row_prev = (grp + ti * groups) * BM;
    if (ti + 1 < my_tiles) {
      tile(accB, accA, ti + 1, true, row_prev);
      row_prev = (grp + (ti + 1) * groups) * BM;
    }
  }
  // the last tile's epilogue
  if (my_tiles & 1) {
    retire(accA[0], 0, row_prev);
    retire(accA[1], 1, row_prev);
  } else {
    retire(accB[0], 0, row_prev);
    retire(accB[1], 1, row_prev);
  }
#endif
}

template <int NK, int ACT, bool NT, bool APF>
static void launch_wstat(const GemmParams& p, int n_panels, int groups, hipStream_t s) {
  constexpr int LDS = (NK + 4) * 128 * 64 * 2;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm_wstat_kernel<NK, ACT, NT, APF>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_wstat_kernel<NK, ACT, NT, APF>), dim3(256), dim3(512), LDS, s, p, n_panels, groups);
}

// the W-stationary kernel serves this problem: returns true after launching it
static bool gemm_try_wstat(const GemmParams& p, hipStream_t s, bool apf) {
  if (!p.out_is_16bit || p.res || p.colscale || p.rope_cos || p.pool_W || p.Q2 || p.res_mod) return false;
  if (!(p.K == 256 || p.K == 384) || p.N % 128 != 0 || p.M % 128 != 0 || p.M < 8192 || !(p.act == 0 || p.act == 1 || p.act == 2)) return false;
  if ((p.ldc & 1) || ((uintptr_t)p.C & 3) || (int64_t)p.M * p.ldc * 2 >= (1ll << 31) || (int64_t)p.M * p.lda * 2 >= (1ll << 31) ||
      (int64_t)p.N * p.ldw * 2 >= (1ll << 31)) return false;
  const int n_panels = p.N / 128;
  if (n_panels > 32) return false;                    // a group has to fit the 32 slots of an XCD
  const int groups = min(256 / n_panels, p.M / 128);
#define WSTAT2(NKV, NTV) \
  do { \
    if (apf) { \
      if (p.act == 0) launch_wstat<NKV, 0, NTV, true>(p, n_panels, groups, s); \
      else if (p.act == 1) launch_wstat<NKV, 1, NTV, true>(p, n_panels, groups, s); \
      else launch_wstat<NKV, 2, NTV, true>(p, n_panels, groups, s); \
    } else if (p.act == 0) launch_wstat<NKV, 0, NTV, false>(p, n_panels, groups, s); \
    else if (p.act == 1) launch_wstat<NKV, 1, NTV, false>(p, n_panels, groups, s); \
    else launch_wstat<NKV, 2, NTV, false>(p, n_panels, groups, s); \
  } while (0)
#define WSTAT(NKV) \
  do { \
    if (p.store_nt) WSTAT2(NKV, true); \
    else WSTAT2(NKV, false); \
  } while (0)
  if (p.K == 384) WSTAT(6);
  else WSTAT(4);
#undef WSTAT
#undef WSTAT2
  return true;
}

// ------------------------------------------------------------------------------------------------------------------
// W-STATIONARY, 256 ROWS PER STEP (round 4).  What the counters said about gemm_wstat_kernel (profiles/r03_gemm_qkv*):
// its L2 -> LDS traffic is right (W once per workgroup, A streamed) but a step is only 8 MFMAs per wave between two workgroup barriers
// -- 256 matrix cycles against a ~2 000-cycle step -- and every MFMA needs 1.5 fragment reads (32 x 64 wave tiles).  Here a workgroup
// works on TWO of its 128-row units at a time: 8 waves as 4 x 2, each a 64 x 64 block (16 MFMAs per wave and 64-deep k-step, one
// fragment read per MFMA as in the tiled kernels: half the barriers and 2/3 of the LDS reads per flop), W panel resident (NK x 16 KiB),
// A through an LDS-DMA ring of 32 KiB stages (256 rows x 64 k) that runs on across tile boundaries -- two stages at K = 384 (the
// workgroup then owns all 160 KiB), three at K = 256.  The two units of a pair are whatever the workgroup's strided list holds next
// (rows grp + u * groups), so the work split over the groups is as fine as the 128-row kernel's; an odd unit left over is paired with
// itself and its twin's results are dropped.  Deferred epilogue as there: the finished 64 x 64 accumulators stay in a second register
// set and one 32 x 32 block is retired in each of the next tile's first four k-steps.  Counted vmcnt waits (loads, LDS-DMA and stores
// share one in-order counter): the ops younger than step s's four pieces are the pieces of step s+1 (three-stage ring only) and the
// stores retired in the one or two iterations since.  Same k order per accumulator as every other GEMM kernel: bit-identical results.
// ------------------------------------------------------------------------------------------------------------------
template <int NK, int ACT, bool NT>
__global__ __launch_bounds__(512, 1) void gemm_wstat256_kernel(GemmParams p, int n_panels, int groups) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BN = 128, BK = 64;
  constexpr int KT_BYTES = BN * BK * 2;             // one k-tile of the W image: 16 KiB
  constexpr int W_BYTES = NK * KT_BYTES;
  constexpr int A_STAGE = 2 * KT_BYTES;             // 256 rows x 64 k: 32 KiB
  constexpr int NST = (160 * 1024 - W_BYTES) / A_STAGE >= 3 ? 3 : 2;
  constexpr int LPA = 4;                            // A pieces (1 KiB each) per wave and step: 32 pieces over 8 waves
  constexpr int SPR = 8;                            // stores per retired 32x32 block
  static_assert(NK >= 4, "one 32x32 block is retired in each of a tile's first four k-steps");
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];   // [W image: NK k-tiles | A ring: NST stages]
  const int tid = threadIdx.x;
  // diagnostic ladder (MSAM2_WS256_PROBE, host side; 0 in production): 1 no A DMA inside the loop, 2 + no retire, 3 + fragments read once,
  // 4 + no barrier -- what each piece of the step costs
  const int probe = p.add_cols;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;          // 4 row slabs of 64 (slabs 0, 1: unit a; 2, 3: unit b) x 2 column halves of 64
  const int half = wave >> 2;                       // which unit of the pair this wave loads (DMA) -- waves 0-3: a, 4-7: b
  const int r = lane & 31, h = lane >> 5, odd = lane & 1;
  int panel, grp;                                   // XCD-aware placement: see gemm_wstat_kernel
  {
    const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
    const int F = 32 / n_panels, R = 32 - F * n_panels;
    if (l < F * n_panels) {
      grp = xcd * F + l / n_panels;
      panel = l % n_panels;
    } else {
      const int q = (l - F * n_panels) + R * xcd;
      grp = 8 * F + q / n_panels;
      panel = q % n_panels;
    }
  }
  const int n_units = p.M / 128;
  const int my_units = grp < min(groups, n_units) ? (n_units - grp + groups - 1) / groups : 0;     // units grp, grp + groups, ...
  if (my_units == 0) return;
  const int n_pairs = (my_units + 1) >> 1;
  const int total = n_pairs * NK;
  const int64_t n0 = (int64_t)panel * BN;
  auto unit_row = [&](int pair, int hf) -> int {    // first row of unit `hf` of pair `pair` (an odd unit left over is its own twin)
    int u = 2 * pair + hf;
    if (u >= my_units) u = 2 * pair;
    return (grp + u * groups) * 128;
  };

  const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7fffffff, 0x00020000);
  const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0x7fffffff, 0x00020000);
  // a piece = 8 rows x 128 B = one DMA instruction; 128-byte-row images, chunk c of row r in slot c ^ ((r >> 1) & 7) (as gemm_glds_kernel)
  unsigned offsA[LPA], offsW[2];
#pragma unroll
  for (int i = 0; i < LPA; ++i) {
    const int row = ((wave & 3) * LPA + i) * 8 + (lane >> 3);            // row inside this wave's unit (0..127)
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    offsA[i] = (unsigned)((int64_t)row * p.lda * 2 + chunk * 16);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wave * 2 + i) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    offsW[i] = (unsigned)((n0 + row) * p.ldw * 2 + chunk * 16);
  }
  unsigned char* const ring = lds + W_BYTES;
  auto issue_a = [&](int s) {
    const int pair = s / NK, kt = s - pair * NK;
    const unsigned so = (unsigned)(((int64_t)unit_row(pair, half) * p.lda + (int64_t)kt * BK) * 2);
    unsigned char* base = ring + (s % NST) * A_STAGE + wave * (LPA * 1024);   // stage rows wave * 32 ..: unit a = rows 0..127, b = 128..255
#pragma unroll
    for (int i = 0; i < LPA; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (__attribute__((address_space(3))) void*)(base + i * 1024), 16, offsA[i], so, 0, 0);
  };
  float bias_j[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) bias_j[j] = p.bias ? p.bias[n0 + wn * 64 + j * 32 + r] : 0.f;
  asm volatile("" : "+v"(bias_j[0]), "+v"(bias_j[1])::"memory");       // waited for HERE, before any DMA is in flight (see gemm_wstat_kernel)
  // ---- prologue: the whole W panel, then the first NST - 1 steps of A (W is older than step 0: waiting for step 0 covers it)
#pragma unroll
  for (int kt = 0; kt < NK; ++kt)
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void*)(lds + kt * KT_BYTES + (wave * 2 + i) * 1024), 16,
                                               offsW[i], (unsigned)kt * BK * 2, 0, 0);
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < total) issue_a(s);

  int offA[2], swzA[2], offB[2], swzB[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ra = wm * 64 + i * 32 + r, rb = wn * 64 + i * 32 + r;
    offA[i] = ra * 128;
    swzA[i] = (ra >> 1) & 7;
    offB[i] = rb * 128;
    swzB[i] = (rb >> 1) & 7;
  }
  const auto c_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)((int64_t)p.M * p.ldc * 2), 0x00020000);
  const int ldc_b = (int)p.ldc * 2;
  const int vcol = (int)(n0 + wn * 64 + (r & ~1)) * 2;
  unsigned sink = 0;
  // one 32x32 block (i, j) of a finished tile; row0 = first row of this wave's 64-row slab
  auto retire = [&](const f32x16& blk, int i, int j, int row0) {
    const int vbase = (row0 + i * 32 + 4 * h + odd) * ldc_b + vcol + j * 64;
#pragma unroll
    for (int k = 0; k < SPR; ++k) {
      const int e0 = 2 * k, ro = (e0 & 3) + 8 * (e0 >> 2);
      float x0 = blk[e0] + bias_j[j], x1 = blk[e0 + 1] + bias_j[j];
      if constexpr (ACT == 1) {
        const f32x2 gp = gelu_erf2(f32x2{x0, x1});
        x0 = gp[0];
        x1 = gp[1];
      } else if constexpr (ACT == 2) { x0 = fmaxf(x0, 0.f); x1 = fmaxf(x1, 0.f); }
      op16x2 own;
      own[0] = f2op(x0);
      own[1] = f2op(x1);
      const unsigned P = __builtin_bit_cast(unsigned, own);
      const unsigned Nb = (unsigned)__builtin_amdgcn_update_dpp(0, (int)P, 0xB1, 0xf, 0xf, false);
      const unsigned outw = __builtin_amdgcn_perm(Nb, P, odd ? 0x03020706u : 0x05040100u);
      if (probe == 5) sink ^= outw;                     // diagnostic: the epilogue's vector work without its stores
      else __builtin_amdgcn_raw_buffer_store_b32(outw, c_rsrc, vbase + ro * ldc_b, 0, NT ? 2 : 0);
    }
  };
  // stores this wave issued in iteration s (a block of the previous tile is retired in the first four k-steps of every tile but the first)
  auto stores_in = [&](int s) -> int { return (s >= NK && (s % NK) < 4) ? SPR : 0; };

  op16x8 keep_a[2], keep_b[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) keep_a[i][e] = keep_b[i][e] = (op16)0.f;
  auto tile = [&](f32x16 (&cur)[2][2], f32x16 (&prev)[2][2], int pair, int prev_row0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) cur[i][j][e] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NK; ++kt) {
      const int s = pair * NK + kt;
      int n = (s >= 1 && probe < 2) ? stores_in(s - 1) : 0;
      if (probe == 5) n = 0;
      if constexpr (NST == 3) n += ((s + 1 < total && (probe < 1 || probe == 5 || s + 1 < NST - 1)) ? LPA : 0) + ((s >= 2 && probe < 2) ? stores_in(s - 2) : 0);
      wstat_wait_dyn(n);
      if (probe < 4 || probe == 5) __builtin_amdgcn_s_barrier();      // step s is complete for every wave; the stage of step s - 1 is free again
      if (s + NST - 1 < total && (probe < 1 || probe == 5)) issue_a(s + NST - 1);
      const unsigned char* sa = ring + (s % NST) * A_STAGE;
      const unsigned char* sw = lds + kt * KT_BYTES;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        op16x8 af[2], bf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if (probe < 3 || probe == 5 || (s == 0 && ks == 0)) {
            af[i] = *reinterpret_cast<const op16x8*>(sa + offA[i] + (((2 * ks + h) ^ swzA[i]) << 4));
            bf[i] = *reinterpret_cast<const op16x8*>(sw + offB[i] + (((2 * ks + h) ^ swzB[i]) << 4));
            keep_a[i] = af[i];
            keep_b[i] = bf[i];
          } else {
            af[i] = keep_a[i];
            bf[i] = keep_b[i];
          }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) cur[i][j] = MSAM2_MFMA_32x32x16(af[i], bf[j], cur[i][j], 0, 0, 0);
      }
      if (pair > 0 && kt < 4 && (probe < 2 || probe == 5)) retire(prev[kt >> 1][kt & 1], kt >> 1, kt & 1, prev_row0);
    }
  };

  f32x16 accA[2][2], accB[2][2];
  int row_prev = 0;
  for (int pair = 0; pair < n_pairs; pair += 2) {
    tile(accA, accB, pair, row_prev);
    row_prev = unit_row(pair, wm >> 1) + (wm & 1) * 64;
    if (pair + 1 < n_pairs) {
      tile(accB, accA, pair + 1, row_prev);
      row_prev = unit_row(pair + 1, wm >> 1) + (wm & 1) * 64;
    }
  }
  // the last tile's epilogue (the twin of an odd unit left over is dropped)
  if (2 * (n_pairs - 1) + (wm >> 1) < my_units) {
    if (n_pairs & 1) {
#pragma unroll
      for (int b = 0; b < 4; ++b) retire(accA[b >> 1][b & 1], b >> 1, b & 1, row_prev);
    } else {
#pragma unroll
      for (int b = 0; b < 4; ++b) retire(accB[b >> 1][b & 1], b >> 1, b & 1, row_prev);
    }
  }
  if (probe == 5 && sink == 0x12345u) ((unsigned*)p.C)[0] = sink;   // keeps the diagnostic's vector work alive
#endif
}

template <int NK, int ACT, bool NT>
static void launch_wstat256(const GemmParams& p, int n_panels, int groups, hipStream_t s) {
  constexpr int W_BYTES = NK * 128 * 64 * 2;
  constexpr int NST = (160 * 1024 - W_BYTES) / (32 * 1024) >= 3 ? 3 : 2;
  constexpr int LDS = W_BYTES + NST * 32 * 1024;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm_wstat256_kernel<NK, ACT, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_wstat256_kernel<NK, ACT, NT>), dim3(256), dim3(512), LDS, s, p, n_panels, groups);
}

// same problem class as gemm_try_wstat.  OPT-IN (MSAM2_GEMM_WSTAT256=1): measured round 4 (tools/wstat_ab.py, three boxes) the 256-row
// kernel is as fast as the 128-row one at K = 256 and 5-15 % slower at K = 384, where its A ring has only two stages -- the hypothesis
// it was built on (barriers per MFMA) was wrong.  What it is kept for is its diagnostic ladder (MSAM2_WS256_PROBE, tools/ws256_probe.sh),
// which says where a W-stationary launch spends its time: DESIGN.md section 3.2.
static bool gemm_try_wstat256(const GemmParams& p, hipStream_t s) {
  static const bool on = getenv("MSAM2_GEMM_WSTAT256") && getenv("MSAM2_GEMM_WSTAT256")[0] == '1';
  if (!on) return false;
  if (!p.out_is_16bit || p.res || p.colscale || p.rope_cos || p.pool_W || p.Q2 || p.res_mod) return false;
  if (!(p.K == 256 || p.K == 384) || p.N % 128 != 0 || p.M % 128 != 0 || p.M < 8192 || !(p.act == 0 || p.act == 1 || p.act == 2)) return false;
  if ((p.ldc & 1) || ((uintptr_t)p.C & 3) || (int64_t)p.M * p.ldc * 2 >= (1ll << 31) || (int64_t)p.M * p.lda * 2 >= (1ll << 31) ||
      (int64_t)p.N * p.ldw * 2 >= (1ll << 31)) return false;
  const int n_panels = p.N / 128;
  if (n_panels > 32) return false;
  const int groups = min(256 / n_panels, p.M / 128);
  GemmParams pp = p;
  static const int probe = getenv("MSAM2_WS256_PROBE") ? atoi(getenv("MSAM2_WS256_PROBE")) : 0;
  pp.add_cols = probe;
#define WS256(NKV, ACTV) \
  do { \
    if (p.store_nt) launch_wstat256<NKV, ACTV, true>(pp, n_panels, groups, s); \
    else launch_wstat256<NKV, ACTV, false>(pp, n_panels, groups, s); \
  } while (0)
  if (p.K == 384) {
    if (p.act == 0) WS256(6, 0);
    else if (p.act == 1) WS256(6, 1);
    else WS256(6, 2);
  } else {
    if (p.act == 0) WS256(4, 0);
    else if (p.act == 1) WS256(4, 1);
    else WS256(4, 2);
  }
#undef WS256
  return true;
}

// ------------------------------------------------------------------------------------------------------------------
// Skinny GEMM for M <= 32 (decoder tokens, hyper-network / IoU / object-pointer MLPs: 4..32 rows): one workgroup per 32
// output columns, its 4 waves split K, operand fragments come straight from global memory (row r, 8 consecutive k = one
// 16-byte load per lane per operand per MFMA, all independent => the whole reduction is in flight at once), partial 32x32
// tiles are summed through LDS.  The tiled kernel walks K serially with one global->LDS->MFMA round trip per 32 k: at
// K = 2048 that is 64 dependent trips (~40 us) for 0.03 GFLOP.
// ------------------------------------------------------------------------------------------------------------------
template <bool F32A>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmParams p) {
  __shared__ float part[4][32][33];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, h = lane >> 5;
  const int64_t n0 = (int64_t)blockIdx.x * 32;
  const int ksteps = p.K / 16;                      // K % 16 == 0 (host check)
  const int per = (ksteps + 3) / 4;
  const int s0 = wave * per, s1 = min(ksteps, s0 + per);
  const int64_t arow = (int64_t)min(r, p.M - 1) * p.lda + h * 8;
  const op16* ap = p.A + arow;
  const float* ap32 = p.A32 + arow;
  const bool add2 = F32A && p.A2 && n0 < p.add_cols;          // workgroup-uniform: add_cols is a multiple of 32
  const float* ap2 = add2 ? p.A2 + arow : ap32;               // always a valid address: the loads below are unconditional
  const op16* wp = p.W + min(n0 + r, (int64_t)p.N - 1) * p.ldw + h * 8;
  // (round 4: no conditional loads anywhere in this kernel.  Each `if (x) v = load` had become a basic block of its own with a full
  //  vmcnt wait behind it -- one exposed L2 latency per k-step of the tail loop and per bias / residual element of the store: most of
  //  the ~6 us this launch-bound kernel took)
  auto load_a = [&](int step) -> op16x8 {
    if constexpr (!F32A) {
      return *reinterpret_cast<const op16x8*>(ap + step * 16);
    } else {
      f32x4 lo = *reinterpret_cast<const f32x4*>(ap32 + step * 16), hi = *reinterpret_cast<const f32x4*>(ap32 + step * 16 + 4);
      const f32x4 lo2 = *reinterpret_cast<const f32x4*>(ap2 + step * 16), hi2 = *reinterpret_cast<const f32x4*>(ap2 + step * 16 + 4);
      op16x8 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = f2op(add2 ? lo[e] + lo2[e] : lo[e]);        // (without an addend the second operand is the first one re-read: an L1 hit)
        o[4 + e] = f2op(add2 ? hi[e] + hi2[e] : hi[e]);
      }
      return o;
    }
  };
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  // U k-steps of loads in flight; steps past s1 are clamped (re-read, not multiplied).  U = 4 when a wave owns at most 4 steps (K <= 256:
  // the token-side projections), so that the clamping does not double their loads
  auto run = [&](auto u_tag) __attribute__((always_inline)) {
    constexpr int U = decltype(u_tag)::value;
    for (int st = s0; st < s1; st += U) {
      op16x8 a[U], w[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int su = min(st + u, s1 - 1);
        a[u] = load_a(su);
        w[u] = *reinterpret_cast<const op16x8*>(wp + su * 16);
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (st + u < s1) acc = MSAM2_MFMA_32x32x16(a[u], w[u], acc, 0, 0, 0);
    }
  };
  if (per <= 4) run(std::integral_constant<int, 4>{});
  else run(std::integral_constant<int, 8>{});
#pragma unroll
  for (int e = 0; e < 16; ++e) part[wave][(e & 3) + 8 * (e >> 2) + 4 * h][r] = acc[e];
  // bias / column scale / residual of this thread's four outputs: loaded (clamped, unconditional per tensor) before the barrier
  int rowv[4], colv[4];
  float bv[4], cv[4], rv[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int idx = t * 256 + tid;
    rowv[t] = idx >> 5;
    colv[t] = idx & 31;
    bv[t] = 0.f; cv[t] = 1.f; rv[t] = 0.f;
  }
  if (p.bias) {
#pragma unroll
    for (int t = 0; t < 4; ++t) bv[t] = p.bias[min(n0 + colv[t], (int64_t)p.N - 1)];
  }
  if (p.colscale) {
#pragma unroll
    for (int t = 0; t < 4; ++t) cv[t] = p.colscale[min(n0 + colv[t], (int64_t)p.N - 1)];
  }
  if (p.res) {
    int64_t at[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int rc = min(rowv[t], p.M - 1);
      const int64_t rr = p.res_mod > 0 ? (int64_t)((unsigned)rc % (unsigned)p.res_mod) : rc;
      at[t] = rr * p.ldr + min(n0 + colv[t], (int64_t)p.N - 1);
    }
    if (p.res_is_16bit) {
#pragma unroll
      for (int t = 0; t < 4; ++t) rv[t] = op2f(reinterpret_cast<const op16*>(p.res)[at[t]]);
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) rv[t] = reinterpret_cast<const float*>(p.res)[at[t]];
    }
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int row = rowv[t], col = colv[t];
    const int64_t n = n0 + col;
    if (row < p.M && n < p.N) {
      float x = part[0][row][col] + part[1][row][col] + part[2][row][col] + part[3][row][col] + bv[t];
      if (p.act == 1) x = gelu_erf(x);
      else if (p.act == 2) x = fmaxf(x, 0.f);
      else if (p.act == 3) x = 1.f / (1.f + __expf(-x));
      x = x * cv[t] + rv[t];
      if (p.out_is_16bit) reinterpret_cast<op16*>(p.C)[(int64_t)row * p.ldc + n] = f2op(x);
      else reinterpret_cast<float*>(p.C)[(int64_t)row * p.ldc + n] = x;
    }
  }
}

template <int BM, int BN, int WM, int WN>
static void launch_gemm(const GemmParams& p, hipStream_t s) {
  dim3 grid(cdiv(p.N, BN), cdiv(p.M, BM));
  hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN>), grid, dim3(WM * WN * 64), 0, s, p);
}

// zero fill of a strided fp32 [M, N] view (split-K accumulates into it).  A kernel rather than hipMemset2DAsync: as a captured
// graph node the memset did not order reliably against the neighbouring kernel nodes (observed: replays of a captured training step
// accumulated into stale data).
__global__ __launch_bounds__(256) void gemm_zero_kernel(float* __restrict__ C, int64_t ldc, int M, int N) {
  const int64_t total = (int64_t)M * N;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) C[(i / N) * ldc + i % N] = 0.f;
}

static int gemm_launch(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const float* colscale,
                       const void* residual, int64_t ldr, int res_is_16bit, int64_t res_mod, void* C, int64_t ldc, int out_is_16bit,
                       int64_t M, int64_t N, int64_t K, int act, void* stream, const float* rope_cos, const float* rope_sin,
                       int rope_cols, int rope_D, int rope_period, int rope_n, int rope_npos, int pool_H = 0, int pool_W = 0,
                       void* Q2 = nullptr, int64_t ldq = 0, int poolq_cols = 0) {
  MSAM2_REQUIRE(A && W && C, "gemm: null operand");
  MSAM2_REQUIRE(M > 0 && N > 0 && K > 0, "gemm: empty problem M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
  MSAM2_REQUIRE(K % 8 == 0 && lda % 8 == 0 && ldw % 8 == 0, "gemm: K, lda, ldw must be multiples of 8 (16-byte rows)");
  MSAM2_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0, "gemm: A and W must be 16-byte aligned");
  MSAM2_REQUIRE(M < (1ll << 31) && N < (1ll << 31), "gemm: M/N too large");
  MSAM2_REQUIRE(act >= 0 && act <= 3, "gemm: bad activation %d", act);
  GemmParams p;
  p.A = (const op16*)A; p.W = (const op16*)W; p.bias = bias; p.colscale = colscale; p.res = residual; p.C = C;
  p.lda = lda; p.ldw = ldw; p.ldr = ldr; p.ldc = ldc; p.res_mod = res_mod;
  p.M = (int)M; p.N = (int)N; p.K = (int)K; p.act = act; p.res_is_16bit = res_is_16bit; p.out_is_16bit = out_is_16bit;
  p.rope_cos = rope_cos; p.rope_sin = rope_sin; p.rope_cols = rope_cols; p.rope_D = rope_D; p.rope_period = rope_period;
  p.rope_n = rope_n; p.rope_npos = rope_npos;
  p.pool_H = pool_H; p.pool_W = pool_W;
  p.Q2 = Q2; p.ldq = ldq; p.poolq_cols = poolq_cols;
  p.A32 = nullptr; p.A2 = nullptr; p.add_cols = 0;
  p.ksplit_tiles = 0;
  {
    const char* e = getenv("MSAM2_NT_BYTES");
    const long long thr = e ? atoll(e) : (30ll << 20);
    const char* e32 = getenv("MSAM2_NT_BYTES_F32");
    const long long thr32 = e32 ? atoll(e32) : (1ll << 60);
    p.store_nt = out_is_16bit ? ((long long)M * N * 2 > thr) : ((long long)M * N * 4 > thr32);
  }
  hipStream_t s = (hipStream_t)stream;
  const char* force = getenv("MSAM2_GEMM_V1");
  const char* var = getenv("MSAM2_GEMM_VARIANT");
  // N = 64 / 192 waste half a 128-wide tile: with a short reduction (K < 192) the register-staged 128x64 kernel is faster
  const bool dma_ok = !(force && force[0] == '1') && M >= 256 && N >= 96 && (N % 128 == 0 || N >= 256 || N % 64 != 0 || K >= 192) &&
                      M * lda * 2 < (1ll << 31) && N * ldw * 2 < (1ll << 31);
  const int tiles = cdiv(p.N, 128) * cdiv(p.M, 128);
  // Variant choice (measured on the benchmark step's shapes, tools/gemm_variants.py):
  //   few tiles (<= one per CU) and a long reduction  -> 3-stage pipelined 128x128x32 (one workgroup per CU has to hide its own latency)
  //   K % 64 == 0 and K >= 384                        -> 128x128x64 2-stage, 2 workgroups per CU (more MFMA work per barrier)
  //   otherwise                                       -> 128x128x32 2-stage, 4 workgroups per CU (short reductions: prologue/epilogue
  //                                                      of one workgroup overlap the main loops of the other three)
  // MSAM2_GEMM_VARIANT = 2 | 5 | 6 | 8 | 9 | 10 | 11 forces one (experiments).
  // split-K for small outputs with a very long reduction (the weight-gradient GEMMs of the backward pass, K = tokens): fp32 output,
  // no activation / column scale / residual.  No GEMM of the forward pass matches (their K <= 3072), so the forward stays
  // bit-reproducible; the atomics make the gradient sums order-dependent at fp32 round-off.
  if (!out_is_16bit && act == 0 && !colscale && !residual && !rope_cos && pool_W == 0 &&
      (int64_t)cdiv(p.M, 128) * cdiv(p.N, 128) <= 32 && K >= 4096 && !getenv("MSAM2_NO_SPLITK")) {
    const int nk_all = cdiv(p.K, GEMM_BK);
    int splits = (int)min((int64_t)64, (int64_t)(512 / (cdiv(p.M, 128) * cdiv(p.N, 128))));
    splits = max(1, min(splits, nk_all / 8));
    if (splits > 1) {
      p.ksplit_tiles = cdiv(nk_all, splits);
      hipLaunchKernelGGL(gemm_zero_kernel, dim3((unsigned)min((int64_t)1024, cdiv((int64_t)M * N, (int64_t)256))), dim3(256), 0, s, (float*)C, ldc,
                         (int)M, (int)N);
      dim3 grid(cdiv(p.N, 128), cdiv(p.M, 128), cdiv(nk_all, p.ksplit_tiles));
      hipLaunchKernelGGL((gemm_kernel<128, 128, 2, 2>), grid, dim3(256), 0, s, p);
      return msam2_check_launch("gemm(split-K)");
    }
  }
  {
    // W-stationary persistent kernel for the short-reduction 16-bit projections (measured, tools/wstat_sweep.py: qkv 16384 x 1152 x 384
    // 28.6 -> 21-25 us, memory-attention linear1 16384 x 2048 x 256 34 -> 25-30 us, 16384 x 768 x 256 16.6 -> 13.6 us).  GELU epilogues
    // too since the activation became a polynomial (common.h): with the erf form the tiled kernel hid the epilogue's VALU work better
    // behind its second workgroup (fc1 16384 x 1536 x 384: 42 vs 45 us); now 38.2 vs 37.8, and 16384 x 1024 x 256 21.1 vs 17.6.
    // MSAM2_GEMM_WSTAT = 0 never, 1 every shape it supports, 2 with A-fragment prefetch, 3 act 0 / 2 only (the round-3 default before).
    const char* ew = getenv("MSAM2_GEMM_WSTAT");
    const int wstat_mode = ew ? atoi(ew) : -1;
    const bool want = wstat_mode == 1 || wstat_mode == 2 || wstat_mode < 0 || (wstat_mode == 3 && (act == 0 || act == 2));
    if (want && !var && dma_ok && wstat_mode < 0 && gemm_try_wstat256(p, s)) return msam2_check_launch("gemm(w-stationary, 256 rows)");
    if (want && !var && dma_ok && gemm_try_wstat(p, s, wstat_mode == 2)) return msam2_check_launch("gemm(w-stationary)");
  }
  const int vv = var ? atoi(var) : (tiles <= 256 && K >= 1024 ? 8 : (K % 64 == 0 && K >= 384 ? 2 : 5));
  if (dma_ok && K % 32 == 0 && vv == 6) {
    hipLaunchKernelGGL((gemm_wide_kernel<256, 128, 3, 2>), dim3(cdiv(p.N, 128) * cdiv(p.M, 256)), dim3(256), 0, s, p);
  } else if (dma_ok && K % 32 == 0 && vv == 8) {
    hipLaunchKernelGGL((gemm_wide_kernel<128, 128, 3, 3>), dim3(tiles), dim3(256), 0, s, p);
  } else if (dma_ok && K % 32 == 0 && vv == 9) {
    hipLaunchKernelGGL((gemm_wide_kernel<256, 128, 2, 2>), dim3(cdiv(p.N, 128) * cdiv(p.M, 256)), dim3(256), 0, s, p);
  } else if (dma_ok && K % 32 == 0 && vv == 10) {
    hipLaunchKernelGGL((gemm_wide_kernel<128, 128, 2, 4>), dim3(tiles), dim3(256), 0, s, p);
  } else if (dma_ok && K % 32 == 0 && vv == 11) {
    hipLaunchKernelGGL((gemm_wide_kernel<128, 128, 4, 2>), dim3(tiles), dim3(256), 0, s, p);
  } else if (dma_ok && K % 64 == 0 && vv == 2) {
    hipLaunchKernelGGL(gemm_glds_kernel, dim3(tiles), dim3(256), 0, s, p);
  } else if (dma_ok && K % 32 == 0) {
    hipLaunchKernelGGL((gemm_glds32_kernel<2, 4>), dim3(tiles), dim3(256), 0, s, p);
  } else if (M <= 32 && K % 16 == 0) {
    hipLaunchKernelGGL((gemm_skinny_kernel<false>), dim3(cdiv(p.N, 32)), dim3(256), 0, s, p);
  } else if (M <= 32) launch_gemm<32, 128, 1, 4>(p, s);
  else if (N <= 32) launch_gemm<128, 32, 4, 1>(p, s);
  else if (N <= 64 || (N % 128 != 0 && N % 64 == 0 && N < 512)) launch_gemm<128, 64, 2, 2>(p, s);
  else launch_gemm<128, 128, 2, 2>(p, s);
  return msam2_check_launch("gemm");
}

// ------------------------------------------------------------------------------------------------------------------
// C[M, N] = A[M, K] B[K, N] with B stored k-MAJOR (row k holds the N outputs): the input-gradient GEMM dX = dY W of a linear layer takes
// the layer's weight W [out, in] exactly as the forward stores it -- no transposed weight copy (161 of them per training iteration at
// ~8.7 us each, latency not bytes: profiles/r03_train_iteration_kernel_stats.csv).  gemm_glds_kernel's structure with the second operand
// handled like gemm_tt_dma_kernel's slabs: a 64-row x 256-byte slab per k-tile by DMA (chunk c of row k at c ^ ((k & 3) << 2), applied on
// the source address), fragments by ds_read_b64_tr_b16 -- here with rows 8 h + q (+ 4) so that the transposed fragment carries the same
// k order as the ds_read_b128 fragment of A (k = 8 h + e).  K % 64 == 0 (slab rows past K cannot be zero-filled by a DMA).
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmParams p) {
  constexpr int BM = 128, BN = 128, BK = 64, A_BYTES = BM * BK * 2, STAGE_BYTES = A_BYTES + BK * BN * 2;  // 16 + 16 KiB
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE_BYTES];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5, li = lane & 15;
  const int n_tiles_n = (p.N + BN - 1) / BN, n_tiles_m = (p.M + BM - 1) / BM;
  const int nwg = n_tiles_n * n_tiles_m;
  int bid = blockIdx.x;
  {
    const int q = nwg / 8, rem = nwg % 8, xcd = bid % 8;
    bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + bid / 8;
  }
  const int64_t m0 = (int64_t)(bid / n_tiles_n) * BM;
  const int n0 = (bid % n_tiles_n) * BN;

  const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7fffffff, 0x00020000);
  const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0x7fffffff, 0x00020000);
  // A: 4 pieces per wave of 8 rows x 128 B (as gemm_glds_kernel); B: 4 pieces per wave of 4 k-rows x 256 B (as gemm_tt_dma_kernel)
  unsigned offsA[4], offsW[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    offsA[i] = (unsigned)(min(m0 + row, (int64_t)p.M - 1) * p.lda * 2 + chunk * 16);
    const int krow = 4 * (4 * wave + i) + (lane >> 4);
    const int col = ((lane & 15) ^ ((krow & 3) << 2)) * 8;
    offsW[i] = (unsigned)(krow * p.ldw * 2 + (n0 + (n0 + col < p.N ? col : 0)) * 2);
  }
  auto issue_piece = [&](int kt, int stage, int i) {
    unsigned char* base = lds + stage * STAGE_BYTES + wave * 4096;
    if (i < 4)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (__attribute__((address_space(3))) void*)(base + i * 1024), 16, offsA[i],
                                               (unsigned)kt * BK * 2, 0, 0);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void*)(base + A_BYTES + (i - 4) * 1024), 16, offsW[i - 4],
                                               (unsigned)kt * BK * (unsigned)p.ldw * 2u, 0, 0);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  int offA[2], swzA[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ra = wm * 64 + i * 32 + r;
    offA[i] = ra * 128;
    swzA[i] = (ra >> 1) & 7;
  }
  // transposed fragment of column block cb (32 columns) x k-step ks (16 rows) of the slab: lane 4 q + p of a 16-lane group addresses row
  // q of the block, columns 4 p .. 4 p + 3; the group of lanes (16 cgrp .. +15) of half h takes rows 8 h + q (elements 0..3) and
  // 8 h + 4 + q (elements 4..7) of columns 16 cgrp .. +15  ==  B[k = 8 h + e][n = lane & 31]
  const int vq = li >> 2, vp = li & 3, cgrp = (lane >> 4) & 1;
  const int t_row = (8 * h + vq) * 256 + ((vp & 1) << 3);
  const int t_sw = vq << 2, t_c0 = 2 * cgrp + (vp >> 1);
  typedef __attribute__((ext_vector_type(8))) short short8_t;
  auto bfrag = [&](const unsigned char* slab, int ks, int cb) -> op16x8 {
    const unsigned char* a0 = slab + t_row + (16 * ks) * 256 + (((cb * 4 + t_c0) ^ t_sw) << 4);
    const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(a0));
    const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(a0 + 4 * 256));
    short8_t t8;
    t8[0] = lo[0]; t8[1] = lo[1]; t8[2] = lo[2]; t8[3] = lo[3];
    t8[4] = hi[0]; t8[5] = hi[1]; t8[6] = hi[2]; t8[7] = hi[3];
    return __builtin_bit_cast(op16x8, t8);
  };

  const int nk = p.K / BK;
#pragma unroll
  for (int i = 0; i < 8; ++i) issue_piece(0, 0, i);
  for (int kt = 0; kt < nk; ++kt) {
    const int st = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's DMA pieces of tile kt have landed
    __builtin_amdgcn_s_barrier();                        // ... and so have every other wave's; stage st^1 is free again
    const unsigned char* sa = lds + st * STAGE_BYTES;
    const unsigned char* sb = sa + A_BYTES;
    op16x8 af[4][2], bfr[4][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int c = 2 * ks + h;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[ks][i] = *reinterpret_cast<const op16x8*>(sa + offA[i] + ((c ^ swzA[i]) << 4));
        bfr[ks][i] = bfrag(sb, ks, wn * 2 + i);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = MSAM2_MFMA_32x32x16(af[ks][i], bfr[ks][j], acc[i][j], 0, 0, 0);
      if (kt + 1 < nk) {
        issue_piece(kt + 1, st ^ 1, 2 * ks);
        issue_piece(kt + 1, st ^ 1, 2 * ks + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __builtin_amdgcn_s_barrier();
  gemm_epilogue<2, 2>(p, acc, reinterpret_cast<float*>(lds) + wave * 32 * 68, m0 + wm * 64, n0 + wn * 64, lane);
}

// out[M, N] = residual + a[M, K] b[K, N] (+ bias), b k-major (row pitch ldb); K % 64 == 0, N % 8 == 0; out fp32 or 16-bit.
extern "C" int msam2_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, const float* bias, const void* residual, int64_t ldr,
                             void* C, int64_t ldc, int out_is_16bit, int64_t M, int64_t N, int64_t K, void* stream) {
  MSAM2_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0, "gemm_nt: bad arguments");
  MSAM2_REQUIRE(K % 64 == 0 && N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0,
                "gemm_nt: K %% 64 == 0, N / lda / ldb multiples of 8 and 16-byte aligned operands");
  MSAM2_REQUIRE(M * lda * 2 < (1ll << 31) && K * ldb * 2 < (1ll << 31) && lda >= K && ldb >= N && ldc >= N, "gemm_nt: bad sizes");
  GemmParams p = {};
  p.A = (const op16*)A; p.W = (const op16*)B; p.bias = bias; p.res = residual; p.C = C;
  p.lda = lda; p.ldw = ldb; p.ldr = ldr; p.ldc = ldc;
  p.M = (int)M; p.N = (int)N; p.K = (int)K; p.out_is_16bit = out_is_16bit;
  p.rope_period = 1; p.rope_npos = 1;
  hipLaunchKernelGGL(gemm_nt_kernel, dim3((unsigned)(cdiv(N, 128) * cdiv(M, 128))), dim3(256), 0, (hipStream_t)stream, p);
  return msam2_check_launch("gemm_nt");
}

extern "C" int msam2_gemm(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const float* colscale,
                          const void* residual, int64_t ldr, int res_is_16bit, int64_t res_mod, void* C, int64_t ldc, int out_is_16bit,
                          int64_t M, int64_t N, int64_t K, int act, void* stream) {
  return gemm_launch(A, lda, W, ldw, bias, colscale, residual, ldr, res_is_16bit, res_mod, C, ldc, out_is_16bit, M, N, K, act, stream,
                     nullptr, nullptr, 0, 0, 1, 0, 1);
}

// Token-side linear layer of the two-way decoder (transformer.py:165-196, 239-263): M <= 32 rows (batch x tokens), A in fp32 straight
// from the residual stream; columns n < add_cols are computed from A + A2 (e.g. queries + query_pe for the q | k thirds of a fused
// projection, queries alone for the v third), the sum and the 16-bit conversion happen in the operand load.  One launch replaces
// the add, the cast and up to three projections.
extern "C" int msam2_gemm_tokens(const float* A, int64_t lda, const float* A2, int64_t add_cols, const void* W, int64_t ldw,
                                 const float* bias, const float* residual, int64_t ldr, void* C, int64_t ldc, int out_is_16bit,
                                 int64_t M, int64_t N, int64_t K, int act, void* stream) {
  MSAM2_REQUIRE(A && W && C, "gemm_tokens: null operand");
  MSAM2_REQUIRE(M > 0 && M <= 32 && N > 0 && K > 0 && K % 16 == 0, "gemm_tokens: needs 1 <= M <= 32 rows and K %% 16 == 0");
  MSAM2_REQUIRE(lda % 4 == 0 && ((uintptr_t)A & 15) == 0 && (!A2 || ((uintptr_t)A2 & 15) == 0) && ldw % 8 == 0 && ((uintptr_t)W & 15) == 0,
                "gemm_tokens: 16-byte aligned rows");
  MSAM2_REQUIRE(add_cols >= 0 && (add_cols % 32 == 0 || add_cols >= N) && act >= 0 && act <= 3,
                "gemm_tokens: add_cols must be a multiple of 32 (or >= N: every column)");
  GemmParams p = {};
  p.A32 = A; p.A2 = A2; p.add_cols = (int)(add_cols > N ? N + 31 : add_cols); p.lda = lda;
  p.W = (const op16*)W; p.ldw = ldw; p.bias = bias; p.res = residual; p.ldr = ldr; p.C = C; p.ldc = ldc;
  p.M = (int)M; p.N = (int)N; p.K = (int)K; p.act = act; p.out_is_16bit = out_is_16bit;
  hipLaunchKernelGGL((gemm_skinny_kernel<true>), dim3(cdiv(p.N, 32)), dim3(256), 0, (hipStream_t)stream, p);
  return msam2_check_launch("gemm_tokens");
}

// Linear layer followed by a 2x2/stride-2 max-pool over the token image (Hiera's pooled shortcut: `do_pool(self.proj(x_norm), self.pool)`,
// hieradet.py:141-145, 23-34): A is the [B*H*W, K] token image, C (fp32) = maxpool2x2(A W^T + bias) as [B*(H/2)*(W/2), N].  The
// un-pooled [B*H*W, N] map (201 MB at stage 1 -> 2) is never written: rows are gathered so that a pooled pixel's four sources are
// the four accumulator registers of one lane.
extern "C" int msam2_gemm_pool2x2(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, float* C, int64_t ldc,
                                  int64_t B, int64_t H, int64_t Wimg, int64_t N, int64_t K, void* stream) {
  MSAM2_REQUIRE(B > 0 && H > 0 && Wimg > 0 && H % 2 == 0 && Wimg % 2 == 0, "gemm_pool2x2: H and W must be even");
  const int64_t M = B * H * Wimg;
  MSAM2_REQUIRE(M >= 256 && (M / 4) * ldc * 4 < (1ll << 31) && M * lda * 2 < (1ll << 31), "gemm_pool2x2: problem out of the 32-bit offset range");
  return gemm_launch(A, lda, W, ldw, bias, nullptr, nullptr, 0, 0, 0, C, ldc, 0, M, N, K, 0, stream, nullptr, nullptr, 0, 0, 1, 0, 1,
                     (int)H, (int)Wimg);
}

// Fused qkv projection of a q-pooling Hiera block (hieradet.py:61-70): QKV (16-bit, image order) gets the k and v columns
// (n >= q_cols) of A W^T + bias; Q2 (16-bit) [B*(H/2)*(W/2), q_cols] = maxpool2x2 of the q columns.  The q columns of QKV are
// NOT written (the pooled q is the only q the block uses), and no separate pooling pass reads them back.
extern "C" int msam2_gemm_qkv_pool2x2(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* QKV, int64_t ldc,
                                      void* Q2, int64_t ldq, int64_t B, int64_t H, int64_t Wimg, int64_t N, int64_t K, int64_t q_cols,
                                      void* stream) {
  MSAM2_REQUIRE(QKV && Q2, "gemm_qkv_pool2x2: null output");
  MSAM2_REQUIRE(B > 0 && H > 0 && Wimg > 0 && H % 2 == 0 && Wimg % 2 == 0, "gemm_qkv_pool2x2: H and W must be even");
  MSAM2_REQUIRE(q_cols > 0 && q_cols % 32 == 0 && q_cols <= N && N % 2 == 0 && ldc % 2 == 0 && ldq % 2 == 0 &&
                    ((uintptr_t)QKV & 3) == 0 && ((uintptr_t)Q2 & 3) == 0,
                "gemm_qkv_pool2x2: q_cols must be a multiple of 32; outputs 4-byte aligned with even strides");
  const int64_t M = B * H * Wimg;
  MSAM2_REQUIRE(M >= 256 && M * ldc * 2 < (1ll << 31) && M * lda * 2 < (1ll << 31), "gemm_qkv_pool2x2: problem out of the 32-bit offset range");
  return gemm_launch(A, lda, W, ldw, bias, nullptr, nullptr, 0, 0, 0, QKV, ldc, 1, M, N, K, 0, stream, nullptr, nullptr, 0, 0, 1, 0, 1,
                     (int)H, (int)Wimg, Q2, ldq, (int)q_cols);
}

// Linear projection with the axial RoPE of RoPEAttention (transformer.py:299-315) fused into the store: C (16-bit) =
// rope(A W^T + bias) on columns n < rope_cols (heads of head_dim channels, adjacent pairs), for rows whose position in their batch
// l = m % rows_per_batch is < n_rope, using table row l % n_pos of cos/sin [n_pos, head_dim/2] (fp32).  Rotation happens on the
// fp32 accumulator, i.e. one rounding instead of the two of a projection followed by an in-place rotation.
extern "C" int msam2_gemm_rope(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* C, int64_t ldc,
                               int64_t M, int64_t N, int64_t K, const float* rope_cos, const float* rope_sin, int64_t rope_cols,
                               int64_t head_dim, int64_t rows_per_batch, int64_t n_rope, int64_t n_pos, void* stream) {
  MSAM2_REQUIRE(rope_cos && rope_sin, "gemm_rope: null table");
  MSAM2_REQUIRE(head_dim > 0 && head_dim % 4 == 0 && rope_cols >= 0 && rope_cols <= N && rope_cols % head_dim == 0,
                "gemm_rope: rope_cols must be whole heads of head_dim channels");
  MSAM2_REQUIRE(rows_per_batch >= 128 && n_pos >= 128 && n_rope >= 0 && n_rope <= rows_per_batch && M % rows_per_batch == 0,
                "gemm_rope: bad row geometry (rows_per_batch, n_pos >= 128; M a multiple of rows_per_batch)");
  MSAM2_REQUIRE(M >= 256 && N % 4 == 0 && ldc % 4 == 0 && ((uintptr_t)C & 15) == 0 && M * ldc * 4 < (1ll << 31) &&
                    (!bias || ((uintptr_t)bias & 15) == 0),
                "gemm_rope: needs the direct-store epilogue (M >= 256, 16-byte aligned C, N %% 4 == 0, C < 2 GiB)");
  return gemm_launch(A, lda, W, ldw, bias, nullptr, nullptr, 0, 0, 0, C, ldc, 1, M, N, K, 0, stream, rope_cos, rope_sin, (int)rope_cols,
                     (int)head_dim, (int)rows_per_batch, (int)n_rope, (int)n_pos);
}
