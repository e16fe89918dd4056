// Fused (flash-style) non-causal attention forward for gfx950, op16 in / fp32 softmax + accumulate / op16 out.
//
// Replaces the reference's three F.scaled_dot_product_attention call sites:
//   hieradet.py:72-76      Hiera windowed / global MHA        (D=96; windows gathered in-kernel, pad keys = qkv bias)
//   transformer.py:318     RoPEAttention in memory attention  (D=256, 1 head, N_k up to ~1e6 -> split-KV + merge)
//   transformer.py:258     two-way decoder Attention          (D=16/32 -> attn_small kernel below, not MFMA-shaped)
//
// MFMA kernel structure (per wave: 32 queries; per workgroup NW waves share K/V tiles of 32 keys in LDS):
//   S^T = K * Q^T   via mfma_f32_32x32x16_bf16(A = K rows from LDS (ds_read_b128), B = Q rows held in registers)
//                   -> lane owns one query column: 16 keys in registers, softmax statistics are lane-local
//   O^T += V^T * P^T via mfma(A = V^T fragments read with ds_read_b64_tr_b16 from the row-major V tile,
//                   B = the S^T accumulator itself converted to op16: its k-order is what the tr-read reproduces)
//   so the query stays on the lane for both products and the O rescale needs no cross-lane traffic.
#include "common.h"
#ifdef MSAM2_STAMP
// diagnostic build only (tools/attn_probe.hip): per-section cycle sums of one wave, never compiled into the product library
__device__ unsigned long long g_stamp[16];
__device__ unsigned long long g_wgtime[4096][4];   // per workgroup: realtime at entry, loop start, loop end, exit (100 MHz ticks)
#define STAMP(var)                                                                     \
  do {                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");         \
    __builtin_amdgcn_sched_barrier(0);                                                 \
  } while (0)
#define STAMP_DECL unsigned long long t0_ = 0, t1_ = 0, t2_ = 0, t3_ = 0, t4_ = 0, t5_ = 0, acc_[5] = {0, 0, 0, 0, 0}
#else
#define STAMP(var)
#define STAMP_DECL
#endif
#include <stdlib.h>
#include <type_traits>

// Online-softmax rescaling is lazy: the running reference m_run of a row moves only when a tile's maximum exceeds it by more than this
// many powers of two (a wave-wide vote), so P = exp2(s - m_run) <= 2^SLACK instead of <= 1 -- exact arithmetic is unchanged (the
// reference cancels in O / l), the 16-bit P keeps its relative precision (256 is far inside the fp16 range), and the rescale of the
// D/32 output accumulators (as many VALU slots as the tile's exponentials when D = 96) runs on the first tiles of a row only instead
// of on ~60 % of them.  0 reproduces the eager rescale.
#ifndef MSAM2_RESCALE_SLACK
#define MSAM2_RESCALE_SLACK 8.f
#endif

struct AttnParams {
  const op16 *q, *k, *v;
  op16* o;
  int64_t q_bs, q_hs, q_ts, k_bs, k_hs, k_ts, v_bs, v_hs, v_ts, o_bs, o_hs, o_ts;  // element strides
  int B, H, Lq, Lk;
  float scale_log2;
  // window gather (win=1): batch index z = b * (nwy*nwx) + window; tokens are ws x ws patches of an h x w token image
  int win, ws_q, ws_k, hq, wq, hk, wk, nwy, nwx;
  const float *kpad, *vpad;  // [H*D] fp32 rows standing in for zero-padded tokens (the qkv bias slices)
  // split-KV
  int splits;
  int defer_merge;  // split-KV partials stay in the workspace; msam2_attention_merge finishes (benchmark / overlap use)
  int split_begin, split_cnt;  // attn_kv64_kernel only: this launch computes splits [split_begin, split_begin + split_cnt) of `splits`
  // attn_kv64_kernel only: when non-null the key count is read from the device (1 <= *lk_dev <= Lk; Lk = the capacity the buffers and
  // the split count were sized for).  A hipGraph captured once for a padded memory bank then serves every fill level of the bucket.
  const int* lk_dev;
  op16* o_part;    // [splits][Bz][H][Lq][D] 16-bit, each split's own softmax-normalised output
  float* ml_part;  // [splits][Bz][H][Lq][2]  (running max in log2 domain, partial sum)
  // log-sum-exp rows [Bz][H][Lq] (log2 domain) for the backward; written by the merge kernel (split path) or by the
  // register-staged kernel (single pass); null = not wanted
  float* lse;
  // attention-probability dropout (train mode; attn_glds_kernel only): thr == 0 = off
  AttnDropout drop;
};

template <int D>
struct AttnCfg {
  static constexpr int BK = 32;
  static constexpr int KS = D * 2 + 16;                                  // K tile row stride (bytes)
  static constexpr int VS = D * 2 + (((D * 2) % 128 == 64) ? 0 : 64);    // V tile row stride: VS % 128 == 64
  static constexpr int STAGE = BK * (KS + VS);
  static constexpr int LDS_BYTES = 2 * STAGE;
};

__device__ __forceinline__ int64_t win_token_offset(int t, int ws, int wy, int wx, int himg, int wimg, bool& valid) {
  const int ty = t / ws, tx = t - ty * ws;
  const int y = wy * ws + ty, x = wx * ws + tx;
  valid = (y < himg) && (x < wimg);
  return (int64_t)y * wimg + x;
}

template <int D, int NW, bool WIN>
__global__ __launch_bounds__(NW * 64, (WIN && NW == 4) ? 3 : 1) void attn_fwd_kernel(AttnParams p) {
  using C = AttnCfg<D>;
  constexpr int NT = NW * 64;
  constexpr int DSTEPS = D / 16;   // k-steps of the QK^T product
  constexpr int DBLK = D / 32;     // 32-row blocks of O^T
  constexpr int CPR = D / 8;       // 16-byte chunks per K/V row
  constexpr int CHUNKS = C::BK * CPR;
  constexpr int PER = (CHUNKS + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, h = lane >> 5;
  const int head = blockIdx.y;
  const int zz = blockIdx.z;
  const int split = zz % p.splits;
  const int z = zz / p.splits;  // batch (x window) index
  int b = z, wy = 0, wx = 0;
  if (WIN) {
    const int nw = p.nwy * p.nwx;
    b = z / nw;
    const int w = z - b * nw;
    wy = w / p.nwx;
    wx = w - wy * p.nwx;
  }
  const op16* qb = p.q + (int64_t)b * p.q_bs + (int64_t)head * p.q_hs;
  const op16* kb = p.k + (int64_t)b * p.k_bs + (int64_t)head * p.k_hs;
  const op16* vb = p.v + (int64_t)b * p.v_bs + (int64_t)head * p.v_hs;

  // ---- this lane's query row -> B-operand fragments kept in registers
  const int qi = blockIdx.x * (NW * 32) + wave * 32 + r;
  bool qvalid = qi < p.Lq;
  int64_t qtok = qi;
  if (WIN && qvalid) qtok = win_token_offset(qi, p.ws_q, wy, wx, p.hq, p.wq, qvalid);
  op16x8 qf[DSTEPS];
#pragma unroll
  for (int s = 0; s < DSTEPS; ++s) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (qvalid) v = *reinterpret_cast<const uint4*>(qb + qtok * p.q_ts + s * 16 + h * 8);
    qf[s] = __builtin_bit_cast(op16x8, v);
  }

  // ---- key range of this split (multiple of BK per split except the tail)
  const int tiles_total = (p.Lk + C::BK - 1) / C::BK;
  const int tiles_per = (tiles_total + p.splits - 1) / p.splits;
  const int t_begin = split * tiles_per;
  const int t_end = min(tiles_total, t_begin + tiles_per);

  uint4 rk[PER], rv[PER];
  auto gload = [&](int tile) {
    const int key0 = tile * C::BK;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = tid + i * NT;
      uint4 kk = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
      if (c < CHUNKS) {
        const int key = key0 + c / CPR, dc = (c % CPR) * 8;
        if (key < p.Lk) {
          bool valid = true;
          int64_t tok = key;
          if (WIN) tok = win_token_offset(key, p.ws_k, wy, wx, p.hk, p.wk, valid);
          if (valid) {
            kk = *reinterpret_cast<const uint4*>(kb + tok * p.k_ts + dc);
            vv = *reinterpret_cast<const uint4*>(vb + tok * p.v_ts + dc);
          } else {
            op16x8 a, bb;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              a[e] = f2op(p.kpad[head * D + dc + e]);
              bb[e] = f2op(p.vpad[head * D + dc + e]);
            }
            kk = __builtin_bit_cast(uint4, a);
            vv = __builtin_bit_cast(uint4, bb);
          }
        }
      }
      rk[i] = kk;
      rv[i] = vv;
    }
  };
  auto lstore = [&](int buf) {
    unsigned char* base = smem + buf * C::STAGE;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = tid + i * NT;
      if (c < CHUNKS) {
        const int row = c / CPR, dc = (c % CPR) * 16;
        *reinterpret_cast<uint4*>(base + row * C::KS + dc) = rk[i];
        *reinterpret_cast<uint4*>(base + C::BK * C::KS + row * C::VS + dc) = rv[i];
      }
    }
  };

  f32x16 o[DBLK];
#pragma unroll
  for (int d = 0; d < DBLK; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[d][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  // per-lane LDS byte offsets
  const int k_off = r * C::KS + h * 16;                                  // + s*32 per k-step
  const int li = lane & 15;
  const int v_off = (4 * h + (li >> 2)) * C::VS + (16 * ((lane >> 4) & 1) + 4 * (li & 3)) * 2;  // + (16 s + 8 u) rows + dblk*64 B

  if (t_begin < t_end) {
    gload(t_begin);
    lstore(0);
  }
  __syncthreads();
  int cur = 0;
  for (int tile = t_begin; tile < t_end; ++tile) {
    if (tile + 1 < t_end) gload(tile + 1);
    const unsigned char* kbase = smem + cur * C::STAGE;
    const unsigned char* vbase = kbase + C::BK * C::KS;
    // S^T[key][query]
    f32x16 s;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
    for (int st = 0; st < DSTEPS; ++st) {
      const op16x8 kf = *reinterpret_cast<const op16x8*>(kbase + k_off + st * 32);
      s = MSAM2_MFMA_32x32x16(kf, qf[st], s, 0, 0, 0);
    }
    // online softmax (log2 domain); keys of register e: (e&3) + 8*(e>>2) + 4*h
    const int key0 = tile * C::BK;
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = key0 + (e & 3) + 8 * (e >> 2) + 4 * h;
      s[e] = (key < p.Lk) ? s[e] * p.scale_log2 : -INFINITY;
      mx = fmaxf(mx, s[e]);
    }
    mx = half_max(mx);
    const float m_new = fmaxf(m_run, mx);
    if (__any(m_new > m_run + MSAM2_RESCALE_SLACK)) {
      const float alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < DBLK; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[d][e] *= alpha;
      m_run = m_new;
    }
    float psum = 0.f;
    op16x8 pf[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float pe = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(s[e] - m_run);
      psum += pe;
      pf[e >> 3][e & 7] = f2op_fast(pe);
    }
    l_run += psum;
    // O^T[d][query] += V^T[d][key] P^T[key][query]
#pragma unroll
    for (int d = 0; d < DBLK; ++d) {
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const unsigned char* a0 = vbase + v_off + (16 * st) * C::VS + d * 64;
        const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (short4_t __attribute__((address_space(3)))*)(a0));
        const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (short4_t __attribute__((address_space(3)))*)(a0 + 8 * C::VS));
        typedef __attribute__((ext_vector_type(8))) short short8_t;
        short8_t vv8;
        vv8[0] = lo[0]; vv8[1] = lo[1]; vv8[2] = lo[2]; vv8[3] = lo[3];
        vv8[4] = hi[0]; vv8[5] = hi[1]; vv8[6] = hi[2]; vv8[7] = hi[3];
        o[d] = MSAM2_MFMA_32x32x16(__builtin_bit_cast(op16x8, vv8), pf[st], o[d], 0, 0, 0);
      }
    }
    if (tile + 1 < t_end) lstore(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue
  const float l_tot = half_sum(l_run);
  if (!qvalid) return;
  if (p.splits == 1) {
    const float inv = 1.f / l_tot;
    if constexpr (!WIN) {
      if (p.lse && h == 0) p.lse[((int64_t)z * p.H + head) * p.Lq + qi] = m_run + __log2f(l_tot);
    }
    op16* ob = p.o + (int64_t)b * p.o_bs + (int64_t)head * p.o_hs + qtok * p.o_ts;
#pragma unroll
    for (int d = 0; d < DBLK; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        op16x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = f2op(o[d][4 * g + e] * inv);
        *reinterpret_cast<op16x4*>(ob + d * 32 + 8 * g + 4 * h) = w;
      }
  } else {
    const int64_t Bz = gridDim.z / p.splits;
    const int64_t row = (((int64_t)split * Bz + z) * p.H + head) * p.Lq + qi;
    op16* op = p.o_part + row * D;
    const float inv = 1.f / l_tot;                            // > 0: every split owns at least one valid key
#pragma unroll
    for (int d = 0; d < DBLK; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        op16x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = f2op(o[d][4 * g + e] * inv);
        *reinterpret_cast<op16x4*>(op + d * 32 + 8 * g + 4 * h) = w;
      }
    if (h == 0) {
      p.ml_part[row * 2 + 0] = m_run;
      p.ml_part[row * 2 + 1] = l_tot;
    }
  }
}

// merge split-KV partials: one wave per (z, head, query), lane = 4 consecutive channels.  Partials are each split's normalised
// output in 16 bits (half the HBM round trip of fp32 sums) with its (max, sum) pair: out = sum_s w_s O_s / sum_s w_s,
// w_s = l_s 2^(m_s - M).
template <int D>
__global__ void attn_merge_kernel(AttnParams p, int Bz) {
  const int64_t gw = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const int64_t rows = (int64_t)Bz * p.H * p.Lq;
  if (gw >= rows) return;
  const int qi = gw % p.Lq;
  const int head = (gw / p.Lq) % p.H;
  const int z = gw / ((int64_t)p.Lq * p.H);
  float M = -INFINITY;
  const int d0 = lane * 4;
  float L = 0.f;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  constexpr int G = 8;
  if (p.splits <= G) {
    // every (max, sum) pair and every partial row of this query loaded up front (split index clamped: unconditional loads).  The serial
    // form below is a chain of 3 * splits dependent loads -- 12 memory latencies for 4 splits: the whole duration of a launch whose 40 MB
    // the chip moves in 6 us (round 4).  Same arithmetic in the same order: bit-identical.
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    f32x2_ ml[G];
    op16x4 t[G];
    const int d0c = d0 < D ? d0 : 0;
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const int64_t at = (int64_t)min(u, p.splits - 1) * rows + gw;
      ml[u] = *reinterpret_cast<const f32x2_*>(p.ml_part + at * 2);
      t[u] = *reinterpret_cast<const op16x4*>(p.o_part + at * D + d0c);
    }
#pragma unroll
    for (int u = 0; u < G; ++u)
      if (u < p.splits) M = fmaxf(M, ml[u][0]);
#pragma unroll
    for (int u = 0; u < G; ++u) {
      if (u < p.splits && ml[u][0] != -INFINITY) {
        const float w = ml[u][1] * __builtin_amdgcn_exp2f(ml[u][0] - M);
        L += w;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += w * op2f(t[u][e]);
      }
    }
  } else {
    for (int s = 0; s < p.splits; ++s) M = fmaxf(M, p.ml_part[((int64_t)s * rows + gw) * 2]);
    for (int s = 0; s < p.splits; ++s) {
      const float m = p.ml_part[((int64_t)s * rows + gw) * 2], l = p.ml_part[((int64_t)s * rows + gw) * 2 + 1];
      if (m == -INFINITY) continue;
      const float w = l * __builtin_amdgcn_exp2f(m - M);
      L += w;
      if (d0 < D) {
        const op16x4 t = *reinterpret_cast<const op16x4*>(p.o_part + ((int64_t)s * rows + gw) * D + d0);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += w * op2f(t[e]);
      }
    }
  }
  if (p.lse && lane == 0) p.lse[gw] = M + __log2f(L);
  if (d0 < D) {
    const float inv = 1.f / L;
    op16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = f2op(acc[e] * inv);
    *reinterpret_cast<op16x4*>(p.o + (int64_t)z * p.o_bs + (int64_t)head * p.o_hs + (int64_t)qi * p.o_ts + d0) = o;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// LDS-DMA variant for D = 128 / 256 (memory attention): K/V tiles go global -> LDS with global_load_lds_dwordx4 (no staging
// registers, no ds_write), which brings the D=256 kernel under 256 registers => 2 workgroups (8 waves) per CU, so one
// wave's softmax VALU and waits hide behind the other's MFMAs.  The DMA writes LDS linearly, so the conflict-avoiding
// layouts are XOR swizzles applied to the per-lane SOURCE address and to the fragment reads:
//   K image [32 keys][2D B]: 16-byte chunk c of key r sits at (c & ~15) | ((c & 15) ^ (r & 15))   (ds_read_b128 rows)
//   V image [32 keys][2D B]: chunk c of key r sits at c ^ ((r & 3) << 2)                          (ds_read_b64_tr_b16)
// ------------------------------------------------------------------------------------------------------------------
// DP = LDS row pitch in elements (power of two >= D).  D = 96 (Hiera global blocks) runs with DP = 128: each 192-byte K/V row is
// DMA'd into a 256-byte LDS row slot (12 of every 16 lanes carry data, the other 4 re-read the row start), so the same
// power-of-two swizzles apply.
template <int D, int DP, int NW, int OCC, bool DROP = false>
__global__ __launch_bounds__(NW * 64, OCC) void attn_glds_kernel(AttnParams p) {
#if defined(__HIP_DEVICE_COMPILE__)  // the buffer-descriptor builtins exist only in the device pass; the host pass just needs the launch stub
  constexpr int BK = 32, RB = DP * 2, CPR = DP / 8, GCPR = D / 8;  // LDS row bytes, chunks per LDS row, valid chunks per global row
  constexpr int TILE = BK * RB, STAGE = 2 * TILE;
  constexpr int KI = TILE / 1024;                            // DMA instructions per operand tile
  constexpr int PW = KI / NW;                                // per wave
  constexpr int DSTEPS = D / 16, DBLK = D / 32;
  static_assert(KI % NW == 0, "tile must split evenly over the waves");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE];

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int r = lane & 31, h = lane >> 5;
  // XCD-aware work mapping.  Workgroups are dealt round-robin over the 8 XCDs (id % 8 share an XCD), each with a private L2;
  // all query tiles of one (batch, head, split) stream the SAME K/V range, so they are placed on one XCD (a contiguous slice
  // of the remapped id space per XCD): K/V then comes from that XCD's L2 instead of being re-fetched from the Infinity Cache
  // by every XCD.  Pure speed: any placement is correct.
  const int gx = gridDim.x, gy = gridDim.y;
  const int nwg = gx * gy * gridDim.z;
  int lid = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
  {
    const int q8 = nwg / 8, rem = nwg % 8, xcd = lid % 8;
    lid = (xcd < rem ? xcd * (q8 + 1) : rem * (q8 + 1) + (xcd - rem) * q8) + lid / 8;
  }
  const int qtile = lid % gx;
  const int head = (lid / gx) % gy;
  const int zz = lid / (gx * gy);
  const int split = zz % p.splits, z = zz / p.splits;
  const op16* qb = p.q + (int64_t)z * p.q_bs + (int64_t)head * p.q_hs;
  const op16* kb = p.k + (int64_t)z * p.k_bs + (int64_t)head * p.k_hs;
  const op16* vb = p.v + (int64_t)z * p.v_bs + (int64_t)head * p.v_hs;

  const int qi = qtile * (NW * 32) + wave * 32 + r;
  const bool qvalid = qi < p.Lq;
  op16x8 qf[DSTEPS];
#pragma unroll
  for (int s = 0; s < DSTEPS; ++s) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (qvalid) v = *reinterpret_cast<const uint4*>(qb + (int64_t)qi * p.q_ts + s * 16 + h * 8);
    qf[s] = __builtin_bit_cast(op16x8, v);
  }

  // dropout stream position of this lane's query row (element index of key 0) and the effective seed
  // (DROP instances only: the inference instances carry none of this -- the D = 256 one sits exactly at its 256-register budget)
  uint64_t drop_base = 0, drop_seed = 0;
  if constexpr (DROP) {
    drop_base = p.drop.offset + (((uint64_t)z * p.H + head) * p.Lq + (qvalid ? qi : 0)) * (uint64_t)p.Lk;
    drop_seed = p.drop.seed + (p.drop.seed_dev ? *p.drop.seed_dev : 0ull);
  }

  const int tiles_total = (p.Lk + BK - 1) / BK;
  const int tiles_per = (tiles_total + p.splits - 1) / p.splits;
  const int t_begin = split * tiles_per;
  const int t_end = min(tiles_total, t_begin + tiles_per);
  // full tiles run in the pipelined loop; a partial last tile (only the split that owns it) is handled after the loop so
  // that the loop body carries no tail logic (and none of its registers)
  const int t_full_end = min(t_end, p.Lk / BK);

  // DMA sources: buffer descriptors (SGPRs) + loop-invariant 32-bit per-lane byte offsets + a scalar tile offset
  // (buffer_load_dwordx4 ... offen lds); piece j of this wave covers flat chunks (wave*PW + j)*64 + lane of the [32][CPR] image
  const auto k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)kb, 0, 0x7fffffff, 0x00020000);
  const auto v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)vb, 0, 0x7fffffff, 0x00020000);
  const int k_rb = (int)p.k_ts * 2, v_rb = (int)p.v_ts * 2;   // global row pitch in bytes
  unsigned koff[PW], voff[PW];
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    const int f = (wave * PW + j) * 64 + lane;
    const int row = f / CPR, c = f % CPR;
    const int kc = (c & ~15) | ((c & 15) ^ (row & 15)), vc = c ^ ((row & 3) << 2);   // global chunk held by LDS slot c of this row
    koff[j] = (unsigned)(row * k_rb + ((kc < GCPR ? kc : 0) << 4));
    voff[j] = (unsigned)(row * v_rb + ((vc < GCPR ? vc : 0) << 4));
  }
  auto issue = [&](int tile, int stage) {
    unsigned char* base = smem + stage * STAGE + wave * PW * 1024;
    const unsigned ks = (unsigned)(tile * BK) * (unsigned)k_rb, vs = (unsigned)(tile * BK) * (unsigned)v_rb;
#pragma unroll
    for (int j = 0; j < PW; ++j)
      glds16(k_rsrc, base + j * 1024, koff[j], ks);
#pragma unroll
    for (int j = 0; j < PW; ++j)
      glds16(v_rsrc, base + TILE + j * 1024, voff[j], vs);
  };

  f32x16 o[DBLK];
#pragma unroll
  for (int d = 0; d < DBLK; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[d][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  STAMP_DECL;

  // fragment read offsets.  K: key row r, chunk 2*st + h.  V (transposed read): lane supplies key 4h + q (+16 st, +8 u),
  // elements d0 + 16*cgrp + 4p .. +3 with q = li>>2, p = li&3.
  const int k_row = r * RB, k_x = r & 15;
  const int li = lane & 15, vq = li >> 2, vp = li & 3, cgrp = (lane >> 4) & 1;
  const int v_row = (4 * h + vq) * RB + ((vp & 1) << 3);
  const int v_sw = vq << 2, v_c0 = 2 * cgrp + (vp >> 1);

  auto compute = [&](int stage, int key0, const bool masked) __attribute__((always_inline)) {
    const unsigned char* kbase = smem + stage * STAGE;
    const unsigned char* vbase = kbase + TILE;
    f32x16 s;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = 0.f;
    if constexpr (D <= 128) {
      // registers to spare: software-pipelined K fragments (the pair for step g+2 is in flight while step g's MFMAs issue)
      auto kread = [&](int st) {
        const int c = 2 * st + h;
        return *reinterpret_cast<const op16x8*>(kbase + k_row + (((c & ~15) | ((c & 15) ^ k_x)) << 4));
      };
      op16x8 kf[2][2];
      kf[0][0] = kread(0);
      kf[0][1] = kread(1);
#pragma unroll
      for (int g2 = 0; g2 < DSTEPS; g2 += 2) {
        const int cur = (g2 >> 1) & 1;
        if (g2 + 2 < DSTEPS) {
          kf[cur ^ 1][0] = kread(g2 + 2);
          kf[cur ^ 1][1] = kread(g2 + 3);
        }
        s = MSAM2_MFMA_32x32x16(kf[cur][0], qf[g2], s, 0, 0, 0);
        s = MSAM2_MFMA_32x32x16(kf[cur][1], qf[g2 + 1], s, 0, 0, 0);
      }
    } else {
    // K fragments two at a time: the scheduling fence keeps the compiler from hoisting all D/16 reads at once, which would
      // push the kernel over its 256-register budget
  #pragma unroll
      for (int g2 = 0; g2 < DSTEPS; g2 += 2) {
        op16x8 kf[2];
  #pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int c = 2 * (g2 + u) + h;
          kf[u] = *reinterpret_cast<const op16x8*>(kbase + k_row + (((c & ~15) | ((c & 15) ^ k_x)) << 4));
        }
  #pragma unroll
        for (int u = 0; u < 2; ++u) s = MSAM2_MFMA_32x32x16(kf[u], qf[g2 + u], s, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    STAMP(t2_);
    // scores stay unscaled in s[]; the softmax scale (log2 domain) is folded into one fma per element: exp2(s*c - m)
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      if (masked) {
        const int key = key0 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (key >= p.Lk) s[e] = -INFINITY;
      }
      mx = fmaxf(mx, s[e]);
    }
    mx = half_max(mx) * p.scale_log2;   // scale > 0: max commutes with it
    const float m_new = fmaxf(m_run, mx);
    if (__any(m_new > m_run + MSAM2_RESCALE_SLACK)) {
      const float alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < DBLK; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[d][e] *= alpha;
      m_run = m_new;
    }
    op16x8 pf[2];
    // two elements per VALU slot where the ISA has packed fp32 ops (v_pk_fma_f32 for the exponent argument, v_pk_add_f32 for the
    // running sum): the softmax is issue-bound, and for D = 96 it is longer than the tile's MFMAs
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t sc2 = {p.scale_log2, p.scale_log2}, nm2 = {-m_run, -m_run};   // -m_run finite: every tile holds >= 1 valid key
    f32x2_t psum2 = {0.f, 0.f};
    if constexpr (DROP) {
      // train mode: the row sum (the softmax denominator) takes every probability, the P V product only the kept ones / (1 - p)
      const uint64_t ebase = drop_base + (uint64_t)(key0 + 4 * h);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float pe = __builtin_amdgcn_exp2f(__builtin_fmaf(s[e], p.scale_log2, -m_run));
        psum2[e & 1] += pe;
        const bool keep = dropout_keep(drop_seed, ebase + (uint64_t)((e & 3) + 8 * (e >> 2)), p.drop.thr);
        pf[e >> 3][e & 7] = f2op_fast(keep ? pe * p.drop.inv_keep : 0.f);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        const f32x2_t sv = {s[e], s[e + 1]};
        const f32x2_t t = __builtin_elementwise_fma(sv, sc2, nm2);
        const f32x2_t pe = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
        psum2 += pe;
        pf[e >> 3][e & 7] = f2op_fast(pe[0]);
        pf[e >> 3][(e & 7) + 1] = f2op_fast(pe[1]);
      }
    }
    l_run += psum2[0] + psum2[1];
    STAMP(t3_);
    if constexpr (D <= 128) {
      typedef __attribute__((ext_vector_type(8))) short short8_t;
      auto vread = [&](int d, int st) {
        const int cch = (d * 4 + v_c0) ^ v_sw;   // swizzled 16-byte chunk; (key & 3) == q for every read below
        const unsigned char* a0 = vbase + v_row + (16 * st) * RB + (cch << 4);
        const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(a0));
        const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(a0 + 8 * RB));
        short8_t vv8;
        vv8[0] = lo[0]; vv8[1] = lo[1]; vv8[2] = lo[2]; vv8[3] = lo[3];
        vv8[4] = hi[0]; vv8[5] = hi[1]; vv8[6] = hi[2]; vv8[7] = hi[3];
        return __builtin_bit_cast(op16x8, vv8);
      };
      op16x8 vf[2][2];
      vf[0][0] = vread(0, 0);
      vf[0][1] = vread(0, 1);
#pragma unroll
      for (int d = 0; d < DBLK; ++d) {
        const int cur = d & 1;
        if (d + 1 < DBLK) {
          vf[cur ^ 1][0] = vread(d + 1, 0);
          vf[cur ^ 1][1] = vread(d + 1, 1);
        }
        o[d] = MSAM2_MFMA_32x32x16(vf[cur][0], pf[0], o[d], 0, 0, 0);
        o[d] = MSAM2_MFMA_32x32x16(vf[cur][1], pf[1], o[d], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int d = 0; d < DBLK; ++d) {
  #pragma unroll
        for (int st = 0; st < 2; ++st) {
          typedef __attribute__((ext_vector_type(8))) short short8_t;
          const int cch = (d * 4 + v_c0) ^ v_sw;   // swizzled 16-byte chunk; (key & 3) == q for every read below
          const unsigned char* a0 = vbase + v_row + (16 * st) * RB + (cch << 4);
          const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(a0));
          const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(a0 + 8 * RB));
          short8_t vv8;
          vv8[0] = lo[0]; vv8[1] = lo[1]; vv8[2] = lo[2]; vv8[3] = lo[3];
          vv8[4] = hi[0]; vv8[5] = hi[1]; vv8[6] = hi[2]; vv8[7] = hi[3];
          o[d] = MSAM2_MFMA_32x32x16(__builtin_bit_cast(op16x8, vv8), pf[st], o[d], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // ONE barrier per tile: tile t's DMA pieces were issued a whole compute phase earlier (so the vmcnt wait is normally free);
  // the barrier then says both "every wave's pieces of tile t have landed" and "every wave has finished reading tile t-1",
  // which is what allows tile t+1 to be streamed into t-1's stage right after it, in the shadow of tile t's MFMAs.
  if (t_begin < t_full_end) issue(t_begin, 0);
  for (int tile = t_begin; tile < t_full_end; ++tile) {
    const int st_i = (tile - t_begin) & 1;
    STAMP(t0_);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (tile + 1 < t_full_end) issue(tile + 1, st_i ^ 1);
    STAMP(t1_);
    compute(st_i, tile * BK, false);
    STAMP(t4_);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    STAMP(t5_);
#ifdef MSAM2_STAMP
    acc_[0] += t1_ - t0_; acc_[1] += t2_ - t1_; acc_[2] += t3_ - t2_; acc_[3] += t4_ - t3_; acc_[4] += t5_ - t4_;
#endif
  }
#ifdef MSAM2_STAMP
  if (blockIdx.x == 3 && blockIdx.y == 0 && blockIdx.z == 1 && threadIdx.x == 64) {
    for (int i = 0; i < 5; ++i) g_stamp[i] = acc_[i];
    g_stamp[5] = (unsigned long long)(t_full_end - t_begin);
  }
#endif
  if (t_full_end < t_end) {
    // partial last tile: rows past Lk re-read the last valid key (their scores are masked to -inf)
    const int key0 = t_full_end * BK, last = p.Lk - 1 - key0;
    unsigned char* base = smem + wave * PW * 1024;
    __builtin_amdgcn_s_barrier();                             // every wave is done with the last full tile's stage
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      const int f = (wave * PW + j) * 64 + lane;
      const int row = f / CPR, c = f % CPR, rr = min(row, last);
      const int kc = (c & ~15) | ((c & 15) ^ (row & 15)), vc = c ^ ((row & 3) << 2);
      const unsigned ko = (unsigned)((key0 + rr) * k_rb + ((kc < GCPR ? kc : 0) << 4));
      const unsigned vo = (unsigned)((key0 + rr) * v_rb + ((vc < GCPR ? vc : 0) << 4));
      glds16(k_rsrc, base + j * 1024, ko, 0);
      glds16(v_rsrc, base + TILE + j * 1024, vo, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    compute(0, key0, true);
  }

  const float l_tot = half_sum(l_run);
  if (!qvalid) return;
  if (p.splits == 1) {
    const float inv = 1.f / l_tot;
    op16* ob = p.o + (int64_t)z * p.o_bs + (int64_t)head * p.o_hs + (int64_t)qi * p.o_ts;
#pragma unroll
    for (int d = 0; d < DBLK; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        op16x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = f2op(o[d][4 * g + e] * inv);
        *reinterpret_cast<op16x4*>(ob + d * 32 + 8 * g + 4 * h) = w;
      }
  } else {
    const int64_t Bz = gridDim.z / p.splits;
    const int64_t row = (((int64_t)split * Bz + z) * p.H + head) * p.Lq + qi;
    op16* op = p.o_part + row * D;
    const float inv = 1.f / l_tot;                            // > 0: every split owns at least one valid key
#pragma unroll
    for (int d = 0; d < DBLK; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        op16x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = f2op(o[d][4 * g + e] * inv);
        *reinterpret_cast<op16x4*>(op + d * 32 + 8 * g + 4 * h) = w;
      }
    if (h == 0) {
      p.ml_part[row * 2 + 0] = m_run;
      p.ml_part[row * 2 + 1] = l_tot;
    }
  }
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// Memory cross-attention with the value product contracted in the 64-channel memory space (attn_kv64_kernel).
// RoPEAttention (transformer.py:288-331) with kv_in_dim = 64 computes softmax(q k^T) (M W_v^T + b_v): the values carry no rotary
// encoding and every softmax row sums to one, so  P (M W_v^T + b_v) = (P M) W_v^T + b_v  exactly.  The kernel therefore streams the
// 64-channel memory rows M as "values" (O' = P M, [Lq, 64]) and the caller folds W_v into the out-projection: 5/8 of the MFMA work of
// the 256-wide formulation, a 4 KB instead of a 16 KB value tile per 32 keys, and a 32-register instead of a 128-register O
// accumulator -- which is what lets three workgroups (12 waves) share a CU and leaves room to keep K fragments in flight.
//   K image [32 keys][512 B]: 16-byte chunk c of key r at (c & ~15) | ((c & 15) ^ (r & 15))   (ds_read_b128 rows, as above)
//   V image [32 keys][128 B]: chunk c of key r at c ^ (((r >> 1) & 1) << 2): the four rows a transposed read touches (r, r+1, r+2,
//   r+3 of an aligned group) land on the four 64-byte quarters of the 256-byte bank row.
// ------------------------------------------------------------------------------------------------------------------
template <int NW, int OCC>
__global__ __launch_bounds__(NW * 64, OCC) void attn_kv64_kernel(AttnParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int D = 256, DV = 64, BK = 32;
  constexpr int RB = D * 2, CPR = D / 8;        // K image: row bytes, 16-byte chunks per row
  constexpr int RBV = DV * 2, CPRV = DV / 8;    // V image
  constexpr int TILE_K = BK * RB, TILE_V = BK * RBV, STAGE = TILE_K + TILE_V;
  constexpr int PWK = TILE_K / 1024 / NW, PWV = TILE_V / 1024 / NW;   // DMA pieces (1 KiB per wave-instruction) per wave
  constexpr int DSTEPS = D / 16, DBLK = DV / 32;
  static_assert(TILE_K % (1024 * NW) == 0 && TILE_V % (1024 * NW) == 0, "tiles must split evenly over the waves");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE];

#ifdef MSAM2_STAMP
  const unsigned long long wg_t0_ = __builtin_amdgcn_s_memrealtime();
#endif
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int r = lane & 31, h = lane >> 5;
  // XCD-aware work mapping (see attn_glds_kernel): the query tiles of one (batch, head, split) share an XCD's L2
  const int gx = gridDim.x, gy = gridDim.y;
  const int nwg = gx * gy * gridDim.z;
  int lid = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
  {
    const int q8 = nwg / 8, rem = nwg % 8, xcd = lid % 8;
    lid = (xcd < rem ? xcd * (q8 + 1) : rem * (q8 + 1) + (xcd - rem) * q8) + lid / 8;
  }
  const int qtile = lid % gx;
  const int head = (lid / gx) % gy;
  const int zz = lid / (gx * gy);
  const int split = p.split_begin + zz % p.split_cnt, z = zz / p.split_cnt;
  const op16* qb = p.q + (int64_t)z * p.q_bs + (int64_t)head * p.q_hs;
  const op16* kb = p.k + (int64_t)z * p.k_bs + (int64_t)head * p.k_hs;
  const op16* vb = p.v + (int64_t)z * p.v_bs + (int64_t)head * p.v_hs;

  const int qi = qtile * (NW * 32) + wave * 32 + r;
  const bool qvalid = qi < p.Lq;
  op16x8 qf[DSTEPS];
#pragma unroll
  for (int s = 0; s < DSTEPS; ++s) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (qvalid) v = *reinterpret_cast<const uint4*>(qb + (int64_t)qi * p.q_ts + s * 16 + h * 8);
    qf[s] = __builtin_bit_cast(op16x8, v);
  }

  const int Lk = p.lk_dev ? min(*p.lk_dev, p.Lk) : p.Lk;   // wave-uniform scalar load
  const int tiles_total = (Lk + BK - 1) / BK;
  const int tiles_per = (tiles_total + p.splits - 1) / p.splits;
  const int t_begin = split * tiles_per;
  const int t_end = min(tiles_total, t_begin + tiles_per);
  const int t_full_end = min(t_end, Lk / BK);

  const auto k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)kb, 0, 0x7fffffff, 0x00020000);
  const auto v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)vb, 0, 0x7fffffff, 0x00020000);
  const int k_rb = (int)p.k_ts * 2, v_rb = (int)p.v_ts * 2;   // global row pitch in bytes
  unsigned koff[PWK], voff[PWV];
#pragma unroll
  for (int j = 0; j < PWK; ++j) {
    const int f = (wave * PWK + j) * 64 + lane;
    const int row = f / CPR, c = f % CPR;
    koff[j] = (unsigned)(row * k_rb + (((c & ~15) | ((c & 15) ^ (row & 15))) << 4));
  }
#pragma unroll
  for (int j = 0; j < PWV; ++j) {
    const int f = (wave * PWV + j) * 64 + lane;
    const int row = f / CPRV, c = f % CPRV;
    voff[j] = (unsigned)(row * v_rb + ((c ^ (((row >> 1) & 1) << 2)) << 4));
  }
  auto issue = [&](int tile, int stage) {
    unsigned char* kdst = smem + stage * STAGE + wave * PWK * 1024;
    unsigned char* vdst = smem + stage * STAGE + TILE_K + wave * PWV * 1024;
    const unsigned ks = (unsigned)(tile * BK) * (unsigned)k_rb, vs = (unsigned)(tile * BK) * (unsigned)v_rb;
#pragma unroll
    for (int j = 0; j < PWK; ++j)
      glds16(k_rsrc, kdst + j * 1024, koff[j], ks);
#pragma unroll
    for (int j = 0; j < PWV; ++j)
      glds16(v_rsrc, vdst + j * 1024, voff[j], vs);
  };

  f32x16 o[DBLK];
#pragma unroll
  for (int d = 0; d < DBLK; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[d][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  STAMP_DECL;

  // fragment read offsets.  K: key row r, chunk 2*st + h.  V (transposed read): lane supplies key 4h + q (+16 st, +8 u),
  // elements d0 + 16*cgrp + 4p .. +3 with q = li>>2, p = li&3.
  const int k_row = r * RB, k_x = r & 15;
  const int li = lane & 15, vq = li >> 2, vp = li & 3, cgrp = (lane >> 4) & 1;
  const int v_row = (4 * h + vq) * RBV + ((vp & 1) << 3);
  const int v_sw = ((vq >> 1) & 1) << 2, v_c0 = 2 * cgrp + (vp >> 1);

  auto compute = [&](int stage, int key0, const bool masked) __attribute__((always_inline)) {
    const unsigned char* kbase = smem + stage * STAGE;
    const unsigned char* vbase = kbase + TILE_K;
    typedef __attribute__((ext_vector_type(8))) short short8_t;
    auto kread = [&](int st) {
      const int c = 2 * st + h;
      return *reinterpret_cast<const op16x8*>(kbase + k_row + (((c & ~15) | ((c & 15) ^ k_x)) << 4));
    };
    auto vread = [&](int d, int st) {
      const int cch = (d * 4 + v_c0) ^ v_sw;
      const unsigned char* a0 = vbase + v_row + (16 * st) * RBV + (cch << 4);
      const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(a0));
      const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(a0 + 8 * RBV));
      short8_t vv8;
      vv8[0] = lo[0]; vv8[1] = lo[1]; vv8[2] = lo[2]; vv8[3] = lo[3];
      vv8[4] = hi[0]; vv8[5] = hi[1]; vv8[6] = hi[2]; vv8[7] = hi[3];
      return __builtin_bit_cast(op16x8, vv8);
    };
    // S^T = K Q^T: a rolling window of KPF K fragments stays in flight ahead of the MFMA that consumes the oldest one
#ifndef MSAM2_KV64_KPF
#define MSAM2_KV64_KPF 4
#endif
    constexpr int KPF = MSAM2_KV64_KPF;
    f32x16 s;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = 0.f;
    op16x8 kf[KPF];
#pragma unroll
    for (int g = 0; g < KPF; ++g) kf[g] = kread(g);
#pragma unroll
    for (int g = 0; g < DSTEPS; ++g) {
      s = MSAM2_MFMA_32x32x16(kf[g % KPF], qf[g], s, 0, 0, 0);
      if (g + KPF < DSTEPS) kf[g % KPF] = kread(g + KPF);
    }
    STAMP(t2_);
    // the four V^T fragments of the tile (16 registers) are fetched under the softmax
    op16x8 vf[DBLK][2];
#pragma unroll
    for (int d = 0; d < DBLK; ++d)
#pragma unroll
      for (int st = 0; st < 2; ++st) vf[d][st] = vread(d, st);
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      if (masked) {
        const int key = key0 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (key >= Lk) s[e] = -INFINITY;
      }
      mx = fmaxf(mx, s[e]);
    }
    mx = half_max(mx) * p.scale_log2;   // scale > 0: max commutes with it
    const float m_new = fmaxf(m_run, mx);
    if (__any(m_new > m_run + MSAM2_RESCALE_SLACK)) {
      const float alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < DBLK; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[d][e] *= alpha;
      m_run = m_new;
    }
    op16x8 pf[2];
    float psum = 0.f;
    const float nm = -m_run;   // finite: every tile holds >= 1 valid key
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float pe = __builtin_amdgcn_exp2f(__builtin_fmaf(s[e], p.scale_log2, nm));
      psum += pe;
      pf[e >> 3][e & 7] = f2op_fast(pe);
    }
    l_run += psum;
    STAMP(t3_);
#pragma unroll
    for (int d = 0; d < DBLK; ++d)
#pragma unroll
      for (int st = 0; st < 2; ++st) o[d] = MSAM2_MFMA_32x32x16(vf[d][st], pf[st], o[d], 0, 0, 0);
  };

  // ONE barrier per tile (see attn_glds_kernel): tile t's pieces were issued a whole compute phase earlier
#ifdef MSAM2_STAMP
  const unsigned long long rt0_ = __builtin_amdgcn_s_memrealtime(), ct0_ = __builtin_amdgcn_s_memtime();
#endif
  if (t_begin < t_full_end) issue(t_begin, 0);
  for (int tile = t_begin; tile < t_full_end; ++tile) {
    const int st_i = (tile - t_begin) & 1;
    STAMP(t0_);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (tile + 1 < t_full_end) issue(tile + 1, st_i ^ 1);
    STAMP(t1_);
    compute(st_i, tile * BK, false);
    STAMP(t4_);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    STAMP(t5_);
#ifdef MSAM2_STAMP
    acc_[0] += t1_ - t0_; acc_[1] += t2_ - t1_; acc_[2] += t3_ - t2_; acc_[3] += t4_ - t3_; acc_[4] += t5_ - t4_;
#endif
  }
#ifdef MSAM2_STAMP
  if (blockIdx.x == 3 && blockIdx.y == 0 && blockIdx.z == 1 && threadIdx.x == 64) {
    for (int i = 0; i < 5; ++i) g_stamp[8 + i] = acc_[i];
    g_stamp[13] = (unsigned long long)(t_full_end - t_begin);
    g_stamp[14] = __builtin_amdgcn_s_memrealtime() - rt0_;      // 100 MHz ticks over the loop
    g_stamp[15] = __builtin_amdgcn_s_memtime() - ct0_;          // shader cycles over the loop
  }
#endif
  if (t_full_end < t_end) {
    // partial last tile: rows past Lk re-read the last valid key (their scores are masked to -inf)
    const int key0 = t_full_end * BK, last = Lk - 1 - key0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                             // every wave is done with the last full tile's stage
#pragma unroll
    for (int j = 0; j < PWK; ++j) {
      const int f = (wave * PWK + j) * 64 + lane;
      const int row = f / CPR, c = f % CPR, rr = min(row, last);
      const unsigned ko = (unsigned)((key0 + rr) * k_rb + (((c & ~15) | ((c & 15) ^ (row & 15))) << 4));
      glds16(k_rsrc, smem + (wave * PWK + j) * 1024, ko, 0);
    }
#pragma unroll
    for (int j = 0; j < PWV; ++j) {
      const int f = (wave * PWV + j) * 64 + lane;
      const int row = f / CPRV, c = f % CPRV, rr = min(row, last);
      const unsigned vo = (unsigned)((key0 + rr) * v_rb + ((c ^ (((row >> 1) & 1) << 2)) << 4));
      glds16(v_rsrc, smem + TILE_K + (wave * PWV + j) * 1024, vo, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    compute(0, key0, true);
  }

#ifdef MSAM2_STAMP
  const unsigned long long wg_t2_ = __builtin_amdgcn_s_memrealtime();
#endif
  const float l_tot = half_sum(l_run);
  if (!qvalid) return;
  // l_tot > 0 whenever the split owns a key (always with a host-side key count); a device-side count below the capacity can leave a
  // trailing split empty: it reports (max = -inf, sum = 0, O' = 0) and the merge gives it weight 0
  const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
  op16* dst;
  if (p.splits == 1) {
    dst = p.o + (int64_t)z * p.o_bs + (int64_t)head * p.o_hs + (int64_t)qi * p.o_ts;
  } else {
    const int64_t Bz = gridDim.z / p.split_cnt;
    const int64_t row = (((int64_t)split * Bz + z) * p.H + head) * p.Lq + qi;
    dst = p.o_part + row * DV;
    if (h == 0) {
      p.ml_part[row * 2 + 0] = m_run;
      p.ml_part[row * 2 + 1] = l_tot;
    }
  }
#pragma unroll
  for (int d = 0; d < DBLK; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      op16x4 w;
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = f2op(o[d][4 * g + e] * inv);
      *reinterpret_cast<op16x4*>(dst + d * 32 + 8 * g + 4 * h) = w;
    }
#ifdef MSAM2_STAMP
  if (threadIdx.x == 0) {
    const int wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (wg < 4096) {
      g_wgtime[wg][0] = wg_t0_; g_wgtime[wg][1] = rt0_; g_wgtime[wg][2] = wg_t2_;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      g_wgtime[wg][3] = __builtin_amdgcn_s_memrealtime();
    }
  }
#endif
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// attn_kv64x2_kernel: the same contraction with 64 queries per wave (two 32-query MFMA column blocks).
//
// Why.  In attn_kv64_kernel every MFMA consumes one fragment read from LDS (1 KiB per wave: a K fragment for S^T = K Q^T, a V^T
// fragment for O'^T = V^T P^T) while the Q / P operand sits in registers: four SIMDs x 1 KiB per 32-cycle MFMA = 128 B/clk, half of
// what the LDS array delivers (256 B/clk, MI355X_MICROARCH.md) before DMA writes and bank-group effects.  Here each fragment read
// feeds TWO MFMAs (query blocks a and b), which halves the LDS bytes (and their energy) per MFMA.
//
// How.  256 queries per workgroup (4 waves x 64), one workgroup per CU, one wave per SIMD with the whole 512-register file: the
// Q fragments of both blocks (128 registers) stay resident.  With a single wave per SIMD nothing hides the softmax behind another
// wave's MFMAs, so the loop is software-pipelined inside the wave: the 32 MFMAs of S(t+1) are independent of the softmax of S(t)
// and sit in the same basic block (the MFMA pipe runs them while the VALU does the exponentials), then the 8 MFMAs of P(t) V(t).
// That needs K(t+1) and V(t) at the same time while tile t+2 streams in: a three-stage LDS ring (60 KiB), still ONE barrier per
// tile -- at iteration t the barrier says "tile t+1 has landed for every wave" and "every wave is done with iteration t-1", which
// frees stage (t+2) % 3 == (t-1) % 3 for the DMA of tile t+2.
// Layouts, swizzles, split-KV outputs, lazy rescale and the device-side key count are those of attn_kv64_kernel.
// Measured (B=4, Lq=4096, Lk=16384, 4 splits, same box, interleaved): 171-181 us against 179-189 us for the 32-query kernel.  The
// gain is small because neither kernel is bound by what it looks bound by: with the softmax removed altogether this loop still takes
// 165-176 us, a deeper K prefetch window (8, 16) or a hand-interleaved MFMA / exp2 stream (slower: 192 us) change nothing -- the
// chip holds ~1.6 GHz under this MFMA density (MI355X_MICROARCH.md, DVFS give-back) and every variant lands on ~1.0 PFLOP/s of
// executed MFMA work.  What this kernel buys is the halved LDS read traffic (energy per MFMA) at equal cycles.
// ------------------------------------------------------------------------------------------------------------------
#ifndef MSAM2_KV64X2_KPF
#define MSAM2_KV64X2_KPF 4
#endif
template <int NW>
__global__ __launch_bounds__(NW * 64, 1) void attn_kv64x2_kernel(AttnParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int D = 256, DV = 64, BK = 32, QB = 2;
  constexpr int RB = D * 2, CPR = D / 8;
  constexpr int RBV = DV * 2, CPRV = DV / 8;
  constexpr int TILE_K = BK * RB, TILE_V = BK * RBV, STAGE = TILE_K + TILE_V;
  constexpr int PWK = TILE_K / 1024 / NW, PWV = TILE_V / 1024 / NW;
  constexpr int DSTEPS = D / 16, DBLK = DV / 32;
  static_assert(TILE_K % (1024 * NW) == 0 && TILE_V % (1024 * NW) == 0, "tiles must split evenly over the waves");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * STAGE];

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int r = lane & 31, h = lane >> 5;
  const int gx = gridDim.x, gy = gridDim.y;
  const int nwg = gx * gy * gridDim.z;
  int lid = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
  {
    const int q8 = nwg / 8, rem = nwg % 8, xcd = lid % 8;
    lid = (xcd < rem ? xcd * (q8 + 1) : rem * (q8 + 1) + (xcd - rem) * q8) + lid / 8;
  }
  const int qtile = lid % gx;
  const int head = (lid / gx) % gy;
  const int zz = lid / (gx * gy);
  const int split = p.split_begin + zz % p.split_cnt, z = zz / p.split_cnt;
  const op16* qb = p.q + (int64_t)z * p.q_bs + (int64_t)head * p.q_hs;
  const op16* kb = p.k + (int64_t)z * p.k_bs + (int64_t)head * p.k_hs;
  const op16* vb = p.v + (int64_t)z * p.v_bs + (int64_t)head * p.v_hs;

  int qi[QB];
  bool qvalid[QB];
  op16x8 qf[QB][DSTEPS];
#pragma unroll
  for (int b = 0; b < QB; ++b) {
    qi[b] = qtile * (NW * 64) + wave * 64 + b * 32 + r;
    qvalid[b] = qi[b] < p.Lq;
#pragma unroll
    for (int s = 0; s < DSTEPS; ++s) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (qvalid[b]) v = *reinterpret_cast<const uint4*>(qb + (int64_t)qi[b] * p.q_ts + s * 16 + h * 8);
      qf[b][s] = __builtin_bit_cast(op16x8, v);
    }
  }

  const int Lk = p.lk_dev ? min(*p.lk_dev, p.Lk) : p.Lk;   // wave-uniform scalar load
  const int tiles_total = (Lk + BK - 1) / BK;
  const int tiles_per = (tiles_total + p.splits - 1) / p.splits;
  const int t_begin = split * tiles_per;
  const int t_end = min(tiles_total, t_begin + tiles_per);
  const int t_full_end = min(t_end, Lk / BK);
  const int nt = t_end - t_begin;

  const auto k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)kb, 0, 0x7fffffff, 0x00020000);
  const auto v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)vb, 0, 0x7fffffff, 0x00020000);
  const int k_rb = (int)p.k_ts * 2, v_rb = (int)p.v_ts * 2;
  // DMA piece j of this wave: LDS row / swizzled chunk it fills (fixed), global offset of that row in a full tile
  int krow[PWK], kchunk[PWK], vrow[PWV], vchunk[PWV];
  unsigned koff[PWK], voff[PWV];
#pragma unroll
  for (int j = 0; j < PWK; ++j) {
    const int f = (wave * PWK + j) * 64 + lane;
    const int row = f / CPR, c = f % CPR;
    krow[j] = row;
    kchunk[j] = ((c & ~15) | ((c & 15) ^ (row & 15))) << 4;
    koff[j] = (unsigned)(row * k_rb + kchunk[j]);
  }
#pragma unroll
  for (int j = 0; j < PWV; ++j) {
    const int f = (wave * PWV + j) * 64 + lane;
    const int row = f / CPRV, c = f % CPRV;
    vrow[j] = row;
    vchunk[j] = (c ^ (((row >> 1) & 1) << 2)) << 4;
    voff[j] = (unsigned)(row * v_rb + vchunk[j]);
  }
  // tile -> stage: full tiles by scalar offset; the partial last tile re-reads its last valid key for the rows past Lk
  auto issue = [&](int tile, int stage) {
    unsigned char* kdst = smem + stage * STAGE + wave * PWK * 1024;
    unsigned char* vdst = smem + stage * STAGE + TILE_K + wave * PWV * 1024;
    if (tile < t_full_end) {
      const unsigned ks = (unsigned)(tile * BK) * (unsigned)k_rb, vs = (unsigned)(tile * BK) * (unsigned)v_rb;
#pragma unroll
      for (int j = 0; j < PWK; ++j)
        glds16(k_rsrc, kdst + j * 1024, koff[j], ks);
#pragma unroll
      for (int j = 0; j < PWV; ++j)
        glds16(v_rsrc, vdst + j * 1024, voff[j], vs);
    } else {
      const int key0 = tile * BK, last = Lk - 1 - key0;
#pragma unroll
      for (int j = 0; j < PWK; ++j) {
        const unsigned ko = (unsigned)((key0 + min(krow[j], last)) * k_rb + kchunk[j]);
        glds16(k_rsrc, kdst + j * 1024, ko, 0);
      }
#pragma unroll
      for (int j = 0; j < PWV; ++j) {
        const unsigned vo = (unsigned)((key0 + min(vrow[j], last)) * v_rb + vchunk[j]);
        glds16(v_rsrc, vdst + j * 1024, vo, 0);
      }
    }
  };

  f32x16 o[QB][DBLK];
  float m_run[QB], l_run[QB];
#pragma unroll
  for (int b = 0; b < QB; ++b) {
    m_run[b] = -INFINITY;
    l_run[b] = 0.f;
#pragma unroll
    for (int d = 0; d < DBLK; ++d)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[b][d][e] = 0.f;
  }

  const int k_row = r * RB, k_x = r & 15;
  const int li = lane & 15, vq = li >> 2, vp = li & 3, cgrp = (lane >> 4) & 1;
  const int v_row = (4 * h + vq) * RBV + ((vp & 1) << 3);
  const int v_sw = ((vq >> 1) & 1) << 2, v_c0 = 2 * cgrp + (vp >> 1);
  typedef __attribute__((ext_vector_type(8))) short short8_t;

  // S^T(a), S^T(b) = K Q_a^T, K Q_b^T: every K fragment read feeds two MFMAs; a rolling window of 4 fragments stays in flight
  auto s_phase = [&](int stage, f32x16 (&s)[QB]) __attribute__((always_inline)) {
    const unsigned char* kbase = smem + stage * STAGE;
    auto kread = [&](int st) {
      const int c = 2 * st + h;
      return *reinterpret_cast<const op16x8*>(kbase + k_row + (((c & ~15) | ((c & 15) ^ k_x)) << 4));
    };
#pragma unroll
    for (int b = 0; b < QB; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) s[b][e] = 0.f;
    constexpr int KPF = MSAM2_KV64X2_KPF;
    op16x8 kf[KPF];
#pragma unroll
    for (int g = 0; g < KPF; ++g) kf[g] = kread(g);
#pragma unroll
    for (int g = 0; g < DSTEPS; ++g) {
#pragma unroll
      for (int b = 0; b < QB; ++b) s[b] = MSAM2_MFMA_32x32x16(kf[g % KPF], qf[b][g], s[b], 0, 0, 0);
      if (g + KPF < DSTEPS) kf[g % KPF] = kread(g + KPF);
    }
  };
  // The online softmax in three pieces so that the long VALU part shares a basic block with independent MFMAs (one wave per SIMD:
  // whatever overlap there is has to be in program order): row maxima (cheap, run under the P V MFMAs), the rare reference move
  // (the only branch, between the blocks), exponentials + sums + 16-bit conversion (run under the next tile's S MFMAs).
  // Row maxima need no masking: rows of the partial last tile past Lk hold copies of the last valid key.
  auto row_max = [&](const f32x16 (&s)[QB], float (&mx)[QB]) __attribute__((always_inline)) {
#pragma unroll
    for (int b = 0; b < QB; ++b) {
      float m = -INFINITY;
#pragma unroll
      for (int e = 0; e < 16; ++e) m = fmaxf(m, s[b][e]);
      mx[b] = half_max(m) * p.scale_log2;   // scale > 0: max commutes with it
    }
  };
  auto move_reference = [&](const float (&mx)[QB]) __attribute__((always_inline)) {
#pragma unroll
    for (int b = 0; b < QB; ++b) {
      const float m_new = fmaxf(m_run[b], mx[b]);
      if (__any(m_new > m_run[b] + MSAM2_RESCALE_SLACK)) {
        const float alpha = (m_run[b] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run[b] - m_new);
        l_run[b] *= alpha;
        // The O' accumulators live in AGPRs (the Q fragments fill the VGPR half of the register file).  Written as plain C++ the
        // compiler hoists the 64 AGPR -> VGPR reads this path needs above the branch, i.e. into EVERY tile; the explicit, volatile
        // read / write keeps them here.  The compiler does not track hazards into inline asm: the s_nops cover the
        // "MFMA writes AGPR -> v_accvgpr_read" wait states of the P V MFMAs issued just before (rare path, 2 x 16 cycles).
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
        for (int d = 0; d < DBLK; ++d)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            float t;
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(o[b][d][e]));
            t *= alpha;
            asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(o[b][d][e]) : "v"(t));
          }
        m_run[b] = m_new;
      }
    }
  };
  auto exponentials = [&](const f32x16 (&s)[QB], op16x8 (&pf)[QB][2], int key0, const bool masked) __attribute__((always_inline)) {
#pragma unroll
    for (int b = 0; b < QB; ++b) {
      float psum = 0.f;
      const float nm = -m_run[b];   // finite: every tile holds >= 1 valid key
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float pe = __builtin_amdgcn_exp2f(__builtin_fmaf(s[b][e], p.scale_log2, nm));
        if (masked) {
          const int key = key0 + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (key >= Lk) pe = 0.f;
        }
        psum += pe;
        pf[b][e >> 3][e & 7] = f2op_fast(pe);
      }
      l_run[b] += psum;
    }
  };
  // the four V^T fragments of a tile (16 registers): fetched at the top of block A, a whole S phase before P V consumes them
  auto v_fetch = [&](int stage, op16x8 (&vf)[DBLK][2]) __attribute__((always_inline)) {
    const unsigned char* vbase = smem + stage * STAGE + TILE_K;
#pragma unroll
    for (int d = 0; d < DBLK; ++d)
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const int cch = (d * 4 + v_c0) ^ v_sw;
        const unsigned char* a0 = vbase + v_row + (16 * st) * RBV + (cch << 4);
        const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(a0));
        const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(a0 + 8 * RBV));
        short8_t vv8;
        vv8[0] = lo[0]; vv8[1] = lo[1]; vv8[2] = lo[2]; vv8[3] = lo[3];
        vv8[4] = hi[0]; vv8[5] = hi[1]; vv8[6] = hi[2]; vv8[7] = hi[3];
        vf[d][st] = __builtin_bit_cast(op16x8, vv8);
      }
  };
  auto pv_phase = [&](const op16x8 (&vf)[DBLK][2], const op16x8 (&pf)[QB][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int d = 0; d < DBLK; ++d)
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int b = 0; b < QB; ++b) o[b][d] = MSAM2_MFMA_32x32x16(vf[d][st], pf[b][st], o[b][d], 0, 0, 0);
  };

  if (nt > 0) {
    issue(t_begin, 0);
    if (nt > 1) {
      issue(t_begin + 1, 1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PWK + PWV) : "memory");   // tile 0 landed, tile 1 may still be in flight
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    // two score buffers used alternately (the loop is unrolled by two): no copy between iterations, the AGPR -> VGPR reads of the
    // scores happen where the exponentials consume them, inside block A
    f32x16 s0[QB], s1[QB];
    op16x8 pf[QB][2], vf[DBLK][2];
    float mx[QB];
    int st_cur = 0, st_nxt = 1, st_ld = 2;
    auto iteration = [&](int i, f32x16 (&s_cur)[QB], f32x16 (&s_nxt)[QB]) __attribute__((always_inline)) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // tile i+1 has landed for every wave; every wave is done with iteration i-1 => stage st_ld is free
      if (i + 2 < nt) issue(t_begin + i + 2, st_ld);
      // block A: V fragments of tile i, 32 MFMAs of S(i+1), the exponentials of tile i (independent of the MFMAs)
      v_fetch(st_cur, vf);
      s_phase(st_nxt, s_nxt);
      exponentials(s_cur, pf, 0, false);            // tile i is full: only the last tile can be partial
      // block B: 8 MFMAs of P(i) V(i) + the row maxima of tile i+1
      pv_phase(vf, pf);
      row_max(s_nxt, mx);
      move_reference(mx);                           // rare branch: rescales o (tile i included) and l
      const int t = st_cur;
      st_cur = st_nxt; st_nxt = st_ld; st_ld = t;
    };
    auto last = [&](f32x16 (&s_cur)[QB]) __attribute__((always_inline)) {
      v_fetch(st_cur, vf);
      if (t_full_end < t_end) exponentials(s_cur, pf, (t_end - 1) * BK, true);
      else exponentials(s_cur, pf, 0, false);
      pv_phase(vf, pf);
    };
    s_phase(0, s0);
    row_max(s0, mx);
    move_reference(mx);
    int i = 0;
    for (; i + 2 < nt; i += 2) {
      iteration(i, s0, s1);
      iteration(i + 1, s1, s0);
    }
    if (i + 1 < nt) {
      iteration(i, s0, s1);
      last(s1);
    } else {
      last(s0);
    }
  }

#pragma unroll
  for (int b = 0; b < QB; ++b) {
    const float l_tot = half_sum(l_run[b]);
    if (!qvalid[b]) continue;
    const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;   // an empty trailing split (device-side key count) reports (-inf, 0, 0)
    op16* dst;
    if (p.splits == 1) {
      dst = p.o + (int64_t)z * p.o_bs + (int64_t)head * p.o_hs + (int64_t)qi[b] * p.o_ts;
    } else {
      const int64_t Bz = gridDim.z / p.split_cnt;
      const int64_t row = (((int64_t)split * Bz + z) * p.H + head) * p.Lq + qi[b];
      dst = p.o_part + row * DV;
      if (h == 0) {
        p.ml_part[row * 2 + 0] = m_run[b];
        p.ml_part[row * 2 + 1] = l_tot;
      }
    }
#pragma unroll
    for (int d = 0; d < DBLK; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        op16x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = f2op(o[b][d][4 * g + e] * inv);
        *reinterpret_cast<op16x4*>(dst + d * 32 + 8 * g + 4 * h) = w;
      }
  }
#endif
}

#ifndef MSAM2_KV64_OCC
#define MSAM2_KV64_OCC 3
#endif
// 64 queries per wave (attn_kv64x2_kernel) whenever a workgroup's 256 queries exist; MSAM2_KV64_V1=1 keeps the 32-query kernel
static bool kv64_use_x2(const AttnParams& p) {
  static const bool v1 = getenv("MSAM2_KV64_V1") != nullptr;
  return !v1 && p.Lq >= 256;
}
static int launch_attn_kv64(const AttnParams& p, int Bz, hipStream_t s) {
  if (kv64_use_x2(p)) {
    dim3 grid(cdiv(p.Lq, 256), p.H, Bz * p.split_cnt);
    hipLaunchKernelGGL((attn_kv64x2_kernel<4>), grid, dim3(256), 0, s, p);
  } else {
    dim3 grid(cdiv(p.Lq, 128), p.H, Bz * p.split_cnt);
    hipLaunchKernelGGL((attn_kv64_kernel<4, MSAM2_KV64_OCC>), grid, dim3(256), 0, s, p);
  }
  if (p.splits > 1 && !p.defer_merge) {
    const int64_t rows = (int64_t)Bz * p.H * p.Lq;
    hipLaunchKernelGGL((attn_merge_kernel<64>), dim3(cdiv(rows * 64, 256)), dim3(256), 0, s, p, Bz);
  }
  return msam2_check_launch("attention_kv64_fwd");
}

// ------------------------------------------------------------------------------------------------------------------
// attn_g96x2_kernel: Hiera's global attention blocks (hieradet.py:58-83 with window_size == 0; D = 96, Lq = Lk = 4096 at 1024^2) in
// the 64-queries-per-wave structure of attn_kv64x2_kernel (round 3).
//
// What changes against attn_glds_kernel<96, 128, 4, 3> (32 queries per wave, three waves per SIMD, a barrier per 32 keys, 256-byte LDS
// rows for 192-byte K / V rows):
//   * 64 queries per wave: every K / V^T fragment read from LDS feeds two MFMAs; one 4-wave workgroup (256 queries) per CU with the
//     whole register file, the softmax of sub-tile i shares a basic block with the 12 independent MFMAs of S(i+1) and the 12 of P V (i);
//   * 64-key stages in a three-slot LDS ring, ONE barrier per 64 keys (two 32-key softmax sub-tiles per stage): at the barrier that
//     opens stage T "stage T+1 has landed for every wave" (its DMA was issued a whole stage earlier) and "every wave is done with
//     stage T-1", so stage T+2 streams into T-1's slot in the shadow of 48 MFMAs per wave;
//   * un-padded images: K [64 keys][192 B] with 16-byte chunk c of key r at (c & ~3) | ((c & 3) ^ ((r >> 2) & 3)) (rows 48 banks
//     apart repeat every 4: the XOR separates the four rows of one residue within a ds_read_b128 lane group), V [64][192 B] plain
//     (the four rows a transposed read touches sit in four different 64-byte bank groups) -- the layouts of attn_win_kernel; the DMA
//     applies the swizzle on its per-lane SOURCE address.  12 KB per operand and stage = 12 one-KiB DMA pieces, three per wave.
// Split-KV in units of 64-key stages (every split owns at least one: checked by the launcher), partials / merge / log-sum-exp rows as
// attn_glds_kernel.  Rows of the last stage past Lk re-read the last valid key; their probabilities are zeroed.
// ------------------------------------------------------------------------------------------------------------------
#ifndef MSAM2_G96_PROBE
#define MSAM2_G96_PROBE 0
#endif
// MREF (round 4): the softmax reference rides in the MFMA.  Q is pre-multiplied by scale * log2(e) when its fragments are loaded, and every
// score tile starts with one extra MFMA whose A operand is the constant "1 in k-slot 0" and whose B operand holds -m~ of the lane's
// query (m~: the running reference, kept exactly representable in the 16-bit operand type), so the accumulator IS the exponent:
// p = exp2(acc) -- one VALU op per score instead of fma + exp, no multiply in the row maximum (16 of ~80 vector instructions per 32 x 32
// tile and wave gone for one MFMA more: the kernel is vector-issue-bound).  When the reference moves (rare: lazy rescale)
// the pending score tile is shifted by m~_old - m~_new in the same branch that rescales O.  Any reference works for the softmax as long
// as the SAME one enters the probabilities and the row sum, so the rounding of m~ costs nothing; the merge of split partials reads m~.
#ifndef MSAM2_G96_MREF
#define MSAM2_G96_MREF 1
#endif
template <int QB, int NW>
__global__ __launch_bounds__(NW * 64, 1) void attn_g96x2_kernel(AttnParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int D = 96, BK = 32, SK = 64;
  constexpr bool MREF = MSAM2_G96_MREF != 0 && MSAM2_G96_PROBE == 0;
  constexpr int PROBE = MSAM2_G96_PROBE;   // diagnostic builds only (tools/g96_probe.sh): 1 no exp2, 2 no softmax, 3 + fragments read once, 4 + no DMA / barrier
  static_assert(QB * NW == 8, "256 queries per workgroup");
  constexpr int RB = D * 2, CPR = D / 8;                   // row bytes, 16-byte chunks per row
  constexpr int TILE = SK * RB, HALF = BK * RB, STAGE = 2 * TILE;
  constexpr int PIECES = TILE / 1024, PW = 2 * PIECES / NW;   // 1-KiB DMA pieces per operand and stage; pieces per wave (3 or 6)
  constexpr int DSTEPS = D / 16, DBLK = D / 32;
  static_assert((2 * PIECES) % NW == 0, "stage must split evenly over the waves");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];   // 3 * STAGE = 72 KB

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int r = lane & 31, h = lane >> 5;
  const int gx = gridDim.x, gy = gridDim.y;
  const int nwg = gx * gy * gridDim.z;
  int lid = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
  {
    const int q8 = nwg / 8, rem = nwg % 8, xcd = lid % 8;   // query tiles of one (batch, head, split) share an XCD's L2 (see attn_glds_kernel)
    lid = (xcd < rem ? xcd * (q8 + 1) : rem * (q8 + 1) + (xcd - rem) * q8) + lid / 8;
  }
  const int qtile = lid % gx;
  const int head = (lid / gx) % gy;
  const int zz = lid / (gx * gy);
  const int split = zz % p.splits, z = zz / p.splits;
  const op16* qb = p.q + (int64_t)z * p.q_bs + (int64_t)head * p.q_hs;
  const op16* kb = p.k + (int64_t)z * p.k_bs + (int64_t)head * p.k_hs;
  const op16* vb = p.v + (int64_t)z * p.v_bs + (int64_t)head * p.v_hs;

  int qi[QB];
  bool qvalid[QB];
  op16x8 qf[QB][DSTEPS];
#pragma unroll
  for (int b = 0; b < QB; ++b) {
    qi[b] = qtile * 256 + wave * (QB * 32) + b * 32 + r;
    qvalid[b] = qi[b] < p.Lq;
#pragma unroll
    for (int s = 0; s < DSTEPS; ++s) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (qvalid[b]) v = *reinterpret_cast<const uint4*>(qb + (int64_t)qi[b] * p.q_ts + s * 16 + h * 8);
      qf[b][s] = __builtin_bit_cast(op16x8, v);
      if constexpr (MREF) {
#pragma unroll
        for (int i = 0; i < 8; ++i) qf[b][s][i] = f2op((float)qf[b][s][i] * p.scale_log2);
      }
    }
  }
  // the reference MFMA's operands: A = 1 in k-slot 0 (lanes of the first k half), B = -m~ of this lane's query in k-slot 0
  op16x8 ones_a, mref[QB];
#pragma unroll
  for (int i = 0; i < 8; ++i) ones_a[i] = (op16)0.f;
  if (h == 0) ones_a[0] = (op16)1.f;
#pragma unroll
  for (int b = 0; b < QB; ++b)
#pragma unroll
    for (int i = 0; i < 8; ++i) mref[b][i] = (op16)0.f;

  const int Lk = p.Lk;
  const int stages_total = (Lk + SK - 1) / SK;
  const int stages_per = (stages_total + p.splits - 1) / p.splits;
  const int s_begin = split * stages_per;
  const int s_end = min(stages_total, s_begin + stages_per);
  const int ns = s_end - s_begin;                                        // >= 1 (launcher)
  const int key_end = min(Lk, s_end * SK);
  const int nt = (key_end - s_begin * SK + BK - 1) / BK;                 // 32-key sub-tiles of this split
  const bool partial = (key_end % BK) != 0;                              // the last sub-tile is short

  // DMA: the 2 * PIECES one-KiB pieces of a stage (K image, then V image, contiguous in LDS) are dealt to the waves in order, so the
  // first half of the waves streams K and the second half V -- one descriptor and one row pitch per wave, chosen once.
  // Piece j of this wave fills LDS chunks ((wave * PW + j) % PIECES) * 64 + lane of its operand's [64 keys][12 chunks] image.
  static_assert(PIECES % PW == 0, "a wave's pieces must not straddle the K / V boundary");
  const bool isv = wave * PW >= PIECES;                               // wave-uniform
  const int d_rb = (int)(isv ? p.v_ts : p.k_ts) * 2;                  // global row pitch in bytes
  const auto d_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(isv ? vb : kb), 0, 0x7fffffff, 0x00020000);
  int drow[PW];
  unsigned dchunk[PW], doff[PW];
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    const int f = ((wave * PW + j) % PIECES) * 64 + lane;
    const int row = f / CPR, c = f % CPR;
    drow[j] = row;
    // K: the global chunk LDS slot c of this row holds (the XOR is an involution); V: plain
    dchunk[j] = (unsigned)((isv ? c : ((c & ~3) | ((c & 3) ^ ((row >> 2) & 3)))) << 4);
    doff[j] = (unsigned)(row * d_rb) + dchunk[j];
  }
  auto issue = [&](int stg, int slot) {
    const unsigned char* dst = smem + slot * STAGE + wave * PW * 1024;
    const int key0 = stg * SK;
    unsigned vo[PW], so = 0;
    if (key0 + SK <= Lk) {
      so = (unsigned)key0 * (unsigned)d_rb;
#pragma unroll
      for (int j = 0; j < PW; ++j) vo[j] = doff[j];
    } else {
      const int last = Lk - 1 - key0;                                 // rows past Lk re-read the last key
#pragma unroll
      for (int j = 0; j < PW; ++j) vo[j] = (unsigned)((key0 + min(drow[j], last)) * d_rb) + dchunk[j];
    }
#pragma unroll
    for (int j = 0; j < PW; j += 3) glds16x3_asm(d_rsrc, dst + j * 1024, vo[j], vo[j + 1], vo[j + 2], so);
  };

  f32x16 o[QB][DBLK];
  float m_run[QB], l_run[QB];
#pragma unroll
  for (int b = 0; b < QB; ++b) {
    m_run[b] = -INFINITY;
    l_run[b] = 0.f;
#pragma unroll
    for (int d = 0; d < DBLK; ++d)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[b][d][e] = 0.f;
  }

  const int k_row = r * RB, k_x = (r >> 2) & 3;
  const int li = lane & 15;
  const int v_off = (4 * h + (li >> 2)) * RB + (16 * ((lane >> 4) & 1) + 4 * (li & 3)) * 2;
  typedef __attribute__((ext_vector_type(8))) short short8_t;

  // S^T(a), S^T(b) = K Q_a^T, K Q_b^T of one 32-key sub-tile: every K fragment read feeds QB MFMAs.  The six fragments of sub-tile
  // i+2 are read from LDS a whole iteration before their MFMAs (k_fetch at the top of iteration i; two register sets used
  // alternately): with one or two waves per SIMD nothing else hides the LDS latency in front of the first MFMA of a block.
  auto k_fetch = [&](int slot, int half, op16x8 (&kf)[DSTEPS]) __attribute__((always_inline)) {
    const unsigned char* kbase = smem + slot * STAGE + half * HALF + k_row;
#pragma unroll
    for (int g = 0; g < DSTEPS; ++g) {
      const int c = 2 * g + h;
      if constexpr (PROBE >= 3) kf[g] = qf[0][g];
      else kf[g] = *reinterpret_cast<const op16x8*>(kbase + (((c & ~3) | ((c & 3) ^ k_x)) << 4));
    }
  };
  auto s_mma = [&](const op16x8 (&kf)[DSTEPS], f32x16 (&s)[QB]) __attribute__((always_inline)) {
#pragma unroll
    for (int b = 0; b < QB; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) s[b][e] = 0.f;
    if constexpr (MREF) {
#pragma unroll
      for (int b = 0; b < QB; ++b) s[b] = MSAM2_MFMA_32x32x16(ones_a, mref[b], s[b], 0, 0, 0);     // every score starts at -m~(query)
    }
#pragma unroll
    for (int g = 0; g < DSTEPS; ++g)
#pragma unroll
      for (int b = 0; b < QB; ++b) s[b] = MSAM2_MFMA_32x32x16(kf[g], qf[b][g], s[b], 0, 0, 0);
  };
  // the online softmax in the three pieces of attn_kv64x2_kernel (row maxima need no masking: rows past Lk hold copies of the last key)
  auto row_max = [&](const f32x16 (&s)[QB], float (&mx)[QB]) __attribute__((always_inline)) {
#pragma unroll
    for (int b = 0; b < QB; ++b) {
      float m = -INFINITY;
#pragma unroll
      for (int e = 0; e < 16; ++e) m = fmaxf(m, s[b][e]);
      mx[b] = MREF ? half_max(m) : half_max(m) * p.scale_log2;     // MREF: already in log2 units, RELATIVE to the current reference
    }
  };
  auto move_reference = [&](const float (&mx)[QB], f32x16 (&s_pend)[QB]) __attribute__((always_inline)) {
#pragma unroll
    for (int b = 0; b < QB; ++b) {
      float m_new, alpha;
      bool moved;
      if constexpr (MREF) {
        // mx is relative to m~ (0 before the first tile: m_run = -inf marks "no reference yet")
        const bool first = m_run[b] == -INFINITY;
        moved = __any(first || mx[b] > MSAM2_RESCALE_SLACK);
        const float m_old = first ? 0.f : m_run[b];
        m_new = (float)f2op(m_old + fmaxf(mx[b], first ? -INFINITY : 0.f));      // exactly representable: it re-enters as an MFMA operand
        alpha = first ? 0.f : __builtin_amdgcn_exp2f(m_old - m_new);
        if (moved) {
          const float shift = m_old - m_new;
#pragma unroll
          for (int e = 0; e < 16; ++e) s_pend[b][e] += shift;                    // the pending tile was computed against the old reference
          mref[b][0] = h == 0 ? f2op(-m_new) : (op16)0.f;
        }
      } else {
        m_new = fmaxf(m_run[b], mx[b]);
        moved = __any(m_new > m_run[b] + MSAM2_RESCALE_SLACK);
        alpha = (m_run[b] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run[b] - m_new);
      }
      if (moved) {
        l_run[b] *= alpha;
        if constexpr (QB == 1) {
          // 256-register budget (two waves per SIMD), no inline asm naming AGPRs: hipcc then selects the VGPR form of every MFMA (O and S
          // live in VGPRs) and the softmax reads S without a v_accvgpr_read per element
#pragma unroll
          for (int d = 0; d < DBLK; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[b][d][e] *= alpha;
        } else {
          // O lives in AGPRs: explicit reads / writes keep the 96 AGPR <-> VGPR moves of this rare path out of every tile (see
          // attn_kv64x2_kernel); the s_nops cover the MFMA -> v_accvgpr_read wait states the compiler does not see into asm
          asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
          for (int d = 0; d < DBLK; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              float t;
              asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(o[b][d][e]));
              t *= alpha;
              asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(o[b][d][e]) : "v"(t));
            }
        }
        m_run[b] = m_new;
      }
    }
  };
  auto exponentials = [&](const f32x16 (&s)[QB], op16x8 (&pf)[QB][2], int key0, const bool masked) __attribute__((always_inline)) {
#pragma unroll
    for (int b = 0; b < QB; ++b) {
      float psum = 0.f;
      const float nm = -m_run[b];   // finite: every sub-tile holds >= 1 valid key
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float pe;
        if constexpr (PROBE >= 1) pe = s[b][e] * p.scale_log2 + nm;
        else if constexpr (MREF) pe = __builtin_amdgcn_exp2f(s[b][e]);          // the accumulator is the exponent
        else pe = __builtin_amdgcn_exp2f(__builtin_fmaf(s[b][e], p.scale_log2, nm));
        if (masked) {
          const int key = key0 + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (key >= Lk) pe = 0.f;
        }
        psum += pe;
        pf[b][e >> 3][e & 7] = f2op_fast(pe);
      }
      l_run[b] += psum;
    }
  };
  auto v_fetch = [&](int slot, int half, op16x8 (&vf)[DBLK][2]) __attribute__((always_inline)) {
    const unsigned char* vbase = smem + slot * STAGE + TILE + half * HALF + v_off;
#pragma unroll
    for (int d = 0; d < DBLK; ++d)
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const unsigned char* a0 = vbase + (16 * st) * RB + d * 64;
        const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(a0));
        const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(a0 + 8 * RB));
        short8_t vv8;
        vv8[0] = lo[0]; vv8[1] = lo[1]; vv8[2] = lo[2]; vv8[3] = lo[3];
        vv8[4] = hi[0]; vv8[5] = hi[1]; vv8[6] = hi[2]; vv8[7] = hi[3];
        vf[d][st] = __builtin_bit_cast(op16x8, vv8);
      }
  };
  auto pv_phase = [&](const op16x8 (&vf)[DBLK][2], const op16x8 (&pf)[QB][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int b = 0; b < QB; ++b)
#pragma unroll
      for (int d = 0; d < DBLK; ++d)
#pragma unroll
        for (int st = 0; st < 2; ++st) o[b][d] = MSAM2_MFMA_32x32x16(vf[d][st], pf[b][st], o[b][d], 0, 0, 0);
  };

  {
    issue(s_begin, 0);
    if (ns > 1) {
      issue(s_begin + 1, 1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");   // stage 0 landed, stage 1 may still be in flight
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    f32x16 s0[QB], s1[QB];
    op16x8 pf[QB][2], vf[DBLK][2], kfa[DSTEPS], kfb[DSTEPS];
    float mx[QB];
    if constexpr (PROBE >= 2) {
#pragma unroll
      for (int b = 0; b < QB; ++b) pf[b][0] = pf[b][1] = qf[b][0];
    }
    if constexpr (PROBE >= 3) v_fetch(0, 0, vf);
    int sl0 = 0, sl1 = 1, sl2 = 2, T = 0;
    // sub-tile 2T of stage T (slot sl0): opens the stage -- the only barrier of the stage
    auto even = [&](f32x16 (&s_cur)[QB], f32x16 (&s_nxt)[QB]) __attribute__((always_inline)) {
      if constexpr (PROBE < 4) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // stage T+1 has landed for every wave; every wave is done with stage T-1 => slot sl2 is free
        if (T + 2 < ns) issue(s_begin + T + 2, sl2);
      }
      if constexpr (PROBE < 3) v_fetch(sl0, 0, vf);
      k_fetch(sl1, 0, kfa);            // K(2T+2), used by the next (odd) iteration: stage T+1 has landed (the barrier above)
      s_mma(kfb, s_nxt);               // S(2T+1)
      if constexpr (PROBE < 2) exponentials(s_cur, pf, 0, false);
      else pf[0][0][0] = (op16)s_cur[0][0];
      pv_phase(vf, pf);
      if constexpr (PROBE < 2) {
        row_max(s_nxt, mx);
        move_reference(mx, s_nxt);
      }
    };
    // sub-tile 2T+1: its successor is the first half of stage T+1
    auto odd = [&](f32x16 (&s_cur)[QB], f32x16 (&s_nxt)[QB]) __attribute__((always_inline)) {
      if constexpr (PROBE < 3) v_fetch(sl0, 1, vf);
      k_fetch(sl1, 1, kfb);            // K(2T+3)
      s_mma(kfa, s_nxt);               // S(2T+2)
      if constexpr (PROBE < 2) exponentials(s_cur, pf, 0, false);
      else pf[0][0][0] = (op16)s_cur[0][0];
      pv_phase(vf, pf);
      if constexpr (PROBE < 2) {
        row_max(s_nxt, mx);
        move_reference(mx, s_nxt);
      }
      const int t = sl0;
      sl0 = sl1; sl1 = sl2; sl2 = t;
      ++T;
    };
    auto last = [&](int half, f32x16 (&s_cur)[QB]) __attribute__((always_inline)) {
      v_fetch(sl0, half, vf);
      if (partial) exponentials(s_cur, pf, s_begin * SK + (nt - 1) * BK, true);
      else exponentials(s_cur, pf, 0, false);
      pv_phase(vf, pf);
    };
    k_fetch(0, 0, kfa);
    s_mma(kfa, s0);
    k_fetch(0, 1, kfb);                // K(1): same stage
    row_max(s0, mx);
    move_reference(mx, s0);
    int i = 0;
    for (; i + 2 < nt; i += 2) {
      even(s0, s1);
      odd(s1, s0);
    }
    if (i + 1 < nt) {
      even(s0, s1);
      last(1, s1);
    } else {
      last(0, s0);
    }
  }

#pragma unroll
  for (int b = 0; b < QB; ++b) {
    const float l_tot = half_sum(l_run[b]);
    if (!qvalid[b]) continue;
    const float inv = 1.f / l_tot;                            // > 0: every split owns at least one valid key
    op16* dst;
    if (p.splits == 1) {
      dst = p.o + (int64_t)z * p.o_bs + (int64_t)head * p.o_hs + (int64_t)qi[b] * p.o_ts;
    } else {
      const int64_t Bz = gridDim.z / p.splits;
      const int64_t row = (((int64_t)split * Bz + z) * p.H + head) * p.Lq + qi[b];
      dst = p.o_part + row * D;
      if (h == 0) {
        p.ml_part[row * 2 + 0] = m_run[b];
        p.ml_part[row * 2 + 1] = l_tot;
      }
    }
#pragma unroll
    for (int d = 0; d < DBLK; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        op16x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = f2op(o[b][d][4 * g + e] * inv);
        *reinterpret_cast<op16x4*>(dst + d * 32 + 8 * g + 4 * h) = w;
      }
  }
#endif
}

// the 64-queries-per-wave kernel for D = 96: inference (no dropout), at least one 256-query workgroup, every split owning a 64-key stage
static bool g96x2_applies(const AttnParams& p) {
  static const bool off = getenv("MSAM2_G96_V1") != nullptr;
  if (off || p.drop.thr || p.Lq < 256) return false;
  const int stages = (p.Lk + 63) / 64, per = (stages + p.splits - 1) / p.splits;
  return (int64_t)(p.splits - 1) * per < stages && (int64_t)p.Lk * p.k_ts * 2 < (1ll << 31) && (int64_t)p.Lk * p.v_ts * 2 < (1ll << 31);
}

template <int D>
static int launch_attn_glds(const AttnParams& p, int Bz, hipStream_t s) {
  dim3 grid(cdiv(p.Lq, 128), p.H, Bz * p.splits);
  if (p.drop.thr) {
    // train-mode instances (mask generator in the softmax): one workgroup per CU, registers to spare
    if constexpr (D == 96) hipLaunchKernelGGL((attn_glds_kernel<96, 128, 4, 1, true>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((attn_glds_kernel<D, D, 4, 1, true>), grid, dim3(256), 0, s, p);
  } else if constexpr (D == 96) {
    if (g96x2_applies(p)) {
      constexpr int LDS = 3 * 2 * 64 * 192;
      static bool attr_set = false;
      static const bool x2 = getenv("MSAM2_G96_X2") != nullptr;   // 4 waves x 64 queries instead of 8 waves x 32
      if (!attr_set) {
        hipFuncSetAttribute((const void*)attn_g96x2_kernel<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        hipFuncSetAttribute((const void*)attn_g96x2_kernel<1, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
      }
      const dim3 g2(cdiv(p.Lq, 256), p.H, Bz * p.splits);
      if (x2) hipLaunchKernelGGL((attn_g96x2_kernel<2, 4>), g2, dim3(256), LDS, s, p);
      else hipLaunchKernelGGL((attn_g96x2_kernel<1, 8>), g2, dim3(512), LDS, s, p);
    } else {
      hipLaunchKernelGGL((attn_glds_kernel<96, 128, 4, 3>), grid, dim3(256), 0, s, p);
    }
  } else hipLaunchKernelGGL((attn_glds_kernel<D, D, 4, 2>), grid, dim3(256), 0, s, p);
  if (p.splits > 1 && !p.defer_merge) {
    const int64_t rows = (int64_t)Bz * p.H * p.Lq;
    hipLaunchKernelGGL((attn_merge_kernel<D>), dim3(cdiv(rows * 64, 256)), dim3(256), 0, s, p, Bz);
  }
  return msam2_check_launch("attention_fwd(glds)");
}

template <int D, int NW, bool WIN>
static int launch_attn(const AttnParams& p, int Bz, hipStream_t s) {
  using C = AttnCfg<D>;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)attn_fwd_kernel<D, NW, WIN>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    attr_set = true;
  }
  dim3 grid(cdiv(p.Lq, NW * 32), p.H, Bz * p.splits);
  // a single key tile per workgroup (the 4 x 4 windows of Hiera blocks 2 / 3: 16 keys) never touches the second stage: half the LDS
  // doubles the workgroups a CU holds, and these launches are bound by how many loads are in flight
  const int lds = ((p.Lk + C::BK - 1) / C::BK <= p.splits) ? C::STAGE : C::LDS_BYTES;
  hipLaunchKernelGGL((attn_fwd_kernel<D, NW, WIN>), grid, dim3(NW * 64), lds, s, p);
  if (p.splits > 1 && !p.defer_merge) {
    const int64_t rows = (int64_t)Bz * p.H * p.Lq;
    hipLaunchKernelGGL((attn_merge_kernel<D>), dim3(cdiv(rows * 64, 256)), dim3(256), 0, s, p, Bz);
  }
  return msam2_check_launch("attention_fwd");
}

template <int D, bool WIN>
static int dispatch_nw(const AttnParams& p, int Bz, hipStream_t s) {
  if (p.Lq <= 32) return launch_attn<D, 1, WIN>(p, Bz, s);
  if (p.Lq <= 64) return launch_attn<D, 2, WIN>(p, Bz, s);
  return launch_attn<D, 4, WIN>(p, Bz, s);
}

extern "C" size_t msam2_attention_workspace_bytes(int64_t Bz, int64_t H, int64_t Lq, int64_t D, int splits) {
  if (splits <= 1) return 0;
  return (size_t)splits * Bz * H * Lq * (D * sizeof(op16) + 2 * sizeof(float));
}

// every split must own at least one 32-key tile
static int attn_effective_splits(int64_t Lk, int splits) {
  const int tiles = (int)((Lk + 31) / 32);
  if (splits > tiles) splits = tiles;
  while (splits > 1 && (int64_t)(splits - 1) * ((tiles + splits - 1) / splits) >= tiles) --splits;
  return splits;
}

// Second half of a split-KV call made with a negative split count: combines the partial (max, sum, O) triples of the
// workspace into o.  Same B, H, Lq, Lk, D, |splits| and workspace as the msam2_attention_fwd call it completes.
extern "C" int msam2_attention_merge(void* o, const int64_t* o_strides, int64_t B, int64_t H, int64_t Lq, int64_t Lk, int64_t D,
                                     int splits, void* workspace, size_t workspace_bytes, void* stream) {
  MSAM2_REQUIRE(o && workspace && B > 0 && H > 0 && Lq > 0 && Lk > 0, "attention_merge: bad arguments");
  MSAM2_REQUIRE(D == 96 || D == 256 || D == 64 || D == 128, "attention_merge: head dim %lld not built", (long long)D);
  splits = attn_effective_splits(Lk, splits < 0 ? -splits : splits);
  MSAM2_REQUIRE(splits > 1, "attention_merge: nothing to merge");
  MSAM2_REQUIRE(workspace_bytes >= msam2_attention_workspace_bytes(B, H, Lq, D, splits), "attention_merge: workspace too small");
  AttnParams p = {};
  p.o = (op16*)o;
  p.o_bs = o_strides[0]; p.o_hs = o_strides[1]; p.o_ts = o_strides[2];
  p.B = (int)B; p.H = (int)H; p.Lq = (int)Lq; p.Lk = (int)Lk;
  p.splits = splits;
  p.o_part = (op16*)workspace;
  p.ml_part = reinterpret_cast<float*>(p.o_part + (size_t)splits * B * H * Lq * D);
  const int64_t rows = B * H * Lq;
  dim3 grid(cdiv(rows * 64, 256));
  hipStream_t s = (hipStream_t)stream;
  switch (D) {
    case 96: hipLaunchKernelGGL((attn_merge_kernel<96>), grid, dim3(256), 0, s, p, (int)B); break;
    case 256: hipLaunchKernelGGL((attn_merge_kernel<256>), grid, dim3(256), 0, s, p, (int)B); break;
    case 64: hipLaunchKernelGGL((attn_merge_kernel<64>), grid, dim3(256), 0, s, p, (int)B); break;
    default: hipLaunchKernelGGL((attn_merge_kernel<128>), grid, dim3(256), 0, s, p, (int)B); break;
  }
  return msam2_check_launch("attention_merge");
}

// q,k,v,o: op16 with element strides {batch, head, token}; the head dim D is contiguous.
// splits > 1: split-KV (flash-decoding) over `splits` key ranges + merge; splits < 0: the split pass only (see above).
static int attention_fwd_impl(const void* q, const int64_t* q_strides, const void* k, const int64_t* k_strides,
                              const void* v, const int64_t* v_strides, void* o, const int64_t* o_strides, int64_t B,
                              int64_t H, int64_t Lq, int64_t Lk, int64_t D, float scale, int splits, void* workspace,
                              size_t workspace_bytes, float* lse, void* stream, const AttnDropout* drop = nullptr) {
  MSAM2_REQUIRE(q && k && v && o, "attention: null tensor");
  MSAM2_REQUIRE(B > 0 && H > 0 && Lq > 0 && Lk > 0, "attention: empty problem");
  MSAM2_REQUIRE(D == 96 || D == 256 || D == 64 || D == 128, "attention: head dim %lld not built (96/256/64/128)", (long long)D);
  // splits < 0: run the split-KV pass only and leave the partials in the workspace for msam2_attention_merge
  const bool defer = splits < 0;
  if (defer) splits = -splits;
  MSAM2_REQUIRE(splits >= 1 && splits <= 64, "attention: bad split count %d", splits);
  // the log-sum-exp rows come from the merge kernel: at least two splits whenever the keys allow (the tuned single-pass kernels
  // stay untouched); a single tile of keys goes through the register-staged kernel, which writes them itself
  if (lse && splits < 2) splits = 2;
  MSAM2_REQUIRE(!(lse && defer), "attention: log-sum-exp output and a deferred merge exclude each other");
  for (int i = 0; i < 3; ++i)
    MSAM2_REQUIRE(q_strides[i] % 8 == 0 && k_strides[i] % 8 == 0 && v_strides[i] % 8 == 0 && o_strides[i] % 4 == 0,
                  "attention: strides must keep 16-byte row alignment");
  splits = attn_effective_splits(Lk, splits);
  MSAM2_REQUIRE(workspace_bytes >= msam2_attention_workspace_bytes(B, H, Lq, D, splits), "attention: workspace too small");
  MSAM2_REQUIRE(!defer || splits > 1, "attention: a deferred merge needs an effective split count > 1");
  AttnParams p = {};
  p.q = (const op16*)q; p.k = (const op16*)k; p.v = (const op16*)v; p.o = (op16*)o;
  p.q_bs = q_strides[0]; p.q_hs = q_strides[1]; p.q_ts = q_strides[2];
  p.k_bs = k_strides[0]; p.k_hs = k_strides[1]; p.k_ts = k_strides[2];
  p.v_bs = v_strides[0]; p.v_hs = v_strides[1]; p.v_ts = v_strides[2];
  p.o_bs = o_strides[0]; p.o_hs = o_strides[1]; p.o_ts = o_strides[2];
  p.B = (int)B; p.H = (int)H; p.Lq = (int)Lq; p.Lk = (int)Lk;
  p.scale_log2 = scale * 1.4426950408889634f;
  p.splits = splits;
  p.defer_merge = defer ? 1 : 0;
  p.o_part = (op16*)workspace;
  p.ml_part = workspace ? reinterpret_cast<float*>(p.o_part + (size_t)splits * B * H * Lq * D) : nullptr;
  p.lse = lse;
  MSAM2_REQUIRE(splits == 1 || workspace, "attention: split-KV needs a workspace");
  hipStream_t s = (hipStream_t)stream;
  const char* force = getenv("MSAM2_ATTN_V1");
  const bool v2 = !(force && force[0] == '1') && Lq > 64 && !(lse && splits == 1);
  if (drop && drop->thr) {
    p.drop = *drop;
    MSAM2_REQUIRE(v2 && (D == 96 || D == 128 || D == 256) && (D != 96 || k_strides[2] * 2 * 32 < (1ll << 31)),
                  "attention: dropout runs on the LDS-DMA kernel only (head dim 96 / 128 / 256, more than 64 queries)");
  }
  switch (D) {
    case 96: return (v2 && k_strides[2] * 2 * 32 < (1ll << 31)) ? launch_attn_glds<96>(p, (int)B, s) : dispatch_nw<96, false>(p, (int)B, s);
    case 256: return v2 ? launch_attn_glds<256>(p, (int)B, s) : dispatch_nw<256, false>(p, (int)B, s);
    case 64: return dispatch_nw<64, false>(p, (int)B, s);
    default: return v2 ? launch_attn_glds<128>(p, (int)B, s) : dispatch_nw<128, false>(p, (int)B, s);
  }
}

extern "C" int msam2_attention_fwd(const void* q, const int64_t* q_strides, const void* k, const int64_t* k_strides,
                                   const void* v, const int64_t* v_strides, void* o, const int64_t* o_strides, int64_t B,
                                   int64_t H, int64_t Lq, int64_t Lk, int64_t D, float scale, int splits, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  return attention_fwd_impl(q, q_strides, k, k_strides, v, v_strides, o, o_strides, B, H, Lq, Lk, D, scale, splits, workspace, workspace_bytes,
                            nullptr, stream);
}

// msam2_attention_fwd that also returns lse [B, H, Lq] fp32 = log2(sum_k 2^(scale * log2(e) * q.k)) per query row (what the
// backward needs to rebuild the probabilities).  Runs with max(splits, 2) splits: size the workspace for that.
extern "C" int msam2_attention_fwd_lse(const void* q, const int64_t* q_strides, const void* k, const int64_t* k_strides,
                                       const void* v, const int64_t* v_strides, void* o, const int64_t* o_strides, int64_t B,
                                       int64_t H, int64_t Lq, int64_t Lk, int64_t D, float scale, int splits, void* workspace,
                                       size_t workspace_bytes, float* lse, void* stream) {
  MSAM2_REQUIRE(lse, "attention_fwd_lse: null lse");
  return attention_fwd_impl(q, q_strides, k, k_strides, v, v_strides, o, o_strides, B, H, Lq, Lk, D, scale, splits, workspace, workspace_bytes,
                            lse, stream);
}

// msam2_attention_fwd_lse with dropout on the attention probabilities (F.scaled_dot_product_attention(dropout_p = p) as RoPEAttention
// calls it in train mode, transformer.py:317-318): probability (b, h, q, k) is kept iff element offset + ((b*H + h)*Lq + q)*Lk + k of the
// counter-based stream `seed` (+ *seed_dev, optional: msam2_counter_bump) says so, kept ones are scaled by 1 / (1 - p); the softmax
// denominator and the returned log-sum-exp rows are those of the UN-dropped probabilities.  msam2_attention_bwd_dropout with the same
// (p, seed, offset) re-creates the mask.  Head dim 96 / 128 / 256, Lq > 64.  No [Lq, Lk] tensor anywhere.
extern "C" int msam2_attention_fwd_lse_dropout(const void* q, const int64_t* q_strides, const void* k, const int64_t* k_strides,
                                               const void* v, const int64_t* v_strides, void* o, const int64_t* o_strides, int64_t B,
                                               int64_t H, int64_t Lq, int64_t Lk, int64_t D, float scale, int splits, void* workspace,
                                               size_t workspace_bytes, float* lse, float p, uint64_t seed, uint64_t offset,
                                               const void* seed_dev, void* stream) {
  MSAM2_REQUIRE(lse, "attention_fwd_lse_dropout: null lse");
  MSAM2_REQUIRE(p >= 0.f && p < 1.f, "attention_fwd_lse_dropout: p must be in [0, 1)");
  AttnDropout d = {(unsigned)fmin(4294967295.0, (double)p * 4294967296.0), 1.f / (1.f - p), seed, offset, (const uint64_t*)seed_dev};
  return attention_fwd_impl(q, q_strides, k, k_strides, v, v_strides, o, o_strides, B, H, Lq, Lk, D, scale, splits, workspace, workspace_bytes,
                            lse, stream, &d);
}

// ------------------------------------------------------------------------------------------------------------------
// Hiera windowed attention, whole window resident (attn_win_kernel): one workgroup per (window, head), one wave per 32 queries.
// The register-staged kernel above walks a window's keys in 32-key tiles with a global-load round trip and a barrier per tile; its
// MFMA work per tile (12 instructions at D = 96) is far shorter than that latency, so a 196-key window cost seven exposed memory
// latencies (37 us per stage-3 block: 17 % of the HBM roof).  Here every thread issues ALL of its K / V / Q gather loads up front
// (one latency), the window's K and V land in LDS once (196 x 192 B each: two workgroups per CU), one barrier, and every wave then
// runs its 32 queries against all keys out of LDS with no further synchronisation.
//   K image [Lk][192 B]: 16-byte chunk c of key r at (c & ~3) | ((c & 3) ^ ((r >> 2) & 3)): rows 48 banks apart repeat every 4,
//   the XOR separates the four rows of one residue in a ds_read_b128 lane group (conflict free).
//   V image [ceil8(Lk)][192 B], plain: the four rows of a transposed read sit in four different 64-byte bank groups.
// Zero-padded window tokens are unmasked keys whose K / V rows are kpad / vpad (the qkv bias), as in the reference.
// ------------------------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(512) void attn_win_kernel(AttnParams p) {
  static_assert(D == 96, "bank analysis of the 192-byte LDS rows is for D = 96");
  constexpr int RB = D * 2, CPR = D / 8, DSTEPS = D / 16, DBLK = D / 32, BK = 32;
  constexpr int BATCH = 6;                                   // gather chunks per operand and thread kept in flight at once
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef MSAM2_STAMP
  const unsigned long long ws0_ = __builtin_amdgcn_s_memrealtime();
  unsigned long long ws1_ = 0, ws2_ = 0, ws3_ = 0;
#endif
  const int tid = threadIdx.x, NT = blockDim.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, h = lane >> 5;
  const int head = blockIdx.x, z = blockIdx.y;
  const int nw = p.nwy * p.nwx;
  const int b = z / nw, w = z - b * nw;
  const int wy = w / p.nwx, wx = w - wy * p.nwx;
  const op16* qb = p.q + (int64_t)b * p.q_bs + (int64_t)head * p.q_hs;
  const op16* kb = p.k + (int64_t)b * p.k_bs + (int64_t)head * p.k_hs;
  const op16* vb = p.v + (int64_t)b * p.v_bs + (int64_t)head * p.v_hs;
  unsigned char* ksm = smem;
  unsigned char* vsm = smem + p.Lk * RB;
  const int lk8 = (p.Lk + 7) & ~7;

  // ---- this lane's query row (issued first: its latency overlaps the K / V gather)
  const int qi = wave * 32 + r;
  bool qvalid = qi < p.Lq;
  int64_t qtok = 0;
  if (qvalid) qtok = win_token_offset(qi, p.ws_q, wy, wx, p.hq, p.wq, qvalid);
  op16x8 qf[DSTEPS];
#pragma unroll
  for (int s = 0; s < DSTEPS; ++s) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (qvalid) v = *reinterpret_cast<const uint4*>(qb + qtok * p.q_ts + s * 16 + h * 8);
    qf[s] = __builtin_bit_cast(op16x8, v);
  }

  // ---- gather the window's K and V rows into LDS: BATCH chunks per operand per thread in flight, then the LDS writes
  const int chunks = p.Lk * CPR;
  for (int c0 = 0; c0 < chunks; c0 += BATCH * NT) {
    uint4 rk[BATCH], rv[BATCH];
#pragma unroll
    for (int i = 0; i < BATCH; ++i) {
      const int c = c0 + i * NT + tid;
      uint4 kk = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
      if (c < chunks) {
        const int key = c / CPR, dc = (c - key * CPR) * 8;
        bool valid;
        const int64_t tok = win_token_offset(key, p.ws_k, wy, wx, p.hk, p.wk, valid);
        if (valid) {
          kk = *reinterpret_cast<const uint4*>(kb + tok * p.k_ts + dc);
          vv = *reinterpret_cast<const uint4*>(vb + tok * p.v_ts + dc);
        } else {
          op16x8 a, bb;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            a[e] = f2op(p.kpad[head * D + dc + e]);
            bb[e] = f2op(p.vpad[head * D + dc + e]);
          }
          kk = __builtin_bit_cast(uint4, a);
          vv = __builtin_bit_cast(uint4, bb);
        }
      }
      rk[i] = kk;
      rv[i] = vv;
    }
#pragma unroll
    for (int i = 0; i < BATCH; ++i) {
      const int c = c0 + i * NT + tid;
      if (c < chunks) {
        const int key = c / CPR, cc = c - key * CPR;
        *reinterpret_cast<uint4*>(ksm + key * RB + (((cc & ~3) | ((cc & 3) ^ ((key >> 2) & 3))) << 4)) = rk[i];
        *reinterpret_cast<uint4*>(vsm + key * RB + (cc << 4)) = rv[i];
      }
    }
  }
  for (int c = chunks + tid; c < lk8 * CPR; c += NT) *reinterpret_cast<uint4*>(vsm + c * 16) = make_uint4(0, 0, 0, 0);   // V rows Lk .. ceil8(Lk): zero
  __syncthreads();
#ifdef MSAM2_STAMP
  ws1_ = __builtin_amdgcn_s_memrealtime();
#endif
  if (wave * 32 >= p.Lq) return;                             // (no barrier below)

  f32x16 o[DBLK];
#pragma unroll
  for (int d = 0; d < DBLK; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[d][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const int li = lane & 15;
  const int v_off = (4 * h + (li >> 2)) * RB + (16 * ((lane >> 4) & 1) + 4 * (li & 3)) * 2;
  const int ntiles = (p.Lk + BK - 1) / BK, nfull = p.Lk / BK;
  // one key tile; MASKED = the partial last tile (keys past Lk get -inf, V rows past ceil8(Lk) are not read)
  auto tile_step = [&](int tile, auto masked_tag) __attribute__((always_inline)) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    const int key0 = tile * BK;
    const int krow = key0 + r;                               // rows past Lk read the V region behind the K image: finite, masked below
    const unsigned char* kp = ksm + krow * RB;
    const int kx = (krow >> 2) & 3;
    f32x16 s;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
    for (int st = 0; st < DSTEPS; ++st) {
      const int c = 2 * st + h;
      const op16x8 kf = *reinterpret_cast<const op16x8*>(kp + (((c & ~3) | ((c & 3) ^ kx)) << 4));
      s = MSAM2_MFMA_32x32x16(kf, qf[st], s, 0, 0, 0);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      if constexpr (MASKED) {
        const int key = key0 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (key >= p.Lk) s[e] = -INFINITY;
      }
      mx = fmaxf(mx, s[e]);
    }
    mx = half_max(mx) * p.scale_log2;   // scale > 0: max commutes with it
    const float m_new = fmaxf(m_run, mx);
    if (__any(m_new > m_run + MSAM2_RESCALE_SLACK)) {
      const float alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < DBLK; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[d][e] *= alpha;
      m_run = m_new;
    }
    float psum = 0.f;
    op16x8 pf[2];
    const float nm = -m_run;                                 // finite: every tile holds >= 1 valid key
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float pe = __builtin_amdgcn_exp2f(__builtin_fmaf(s[e], p.scale_log2, nm));
      psum += pe;
      pf[e >> 3][e & 7] = f2op_fast(pe);
    }
    l_run += psum;
    const int nv = MASKED ? min(BK, lk8 - key0) : BK;        // V rows of this tile that exist in LDS (multiple of 8)
#pragma unroll
    for (int d = 0; d < DBLK; ++d) {
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        typedef __attribute__((ext_vector_type(8))) short short8_t;
        const unsigned char* a0 = vsm + (key0 + 16 * st) * RB + v_off + d * 64;
        short4_t lo = {0, 0, 0, 0}, hi = {0, 0, 0, 0};
        if (!MASKED || 16 * st < nv) lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(a0));
        if (!MASKED || 16 * st + 8 < nv) hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(a0 + 8 * RB));
        short8_t vv8;
        vv8[0] = lo[0]; vv8[1] = lo[1]; vv8[2] = lo[2]; vv8[3] = lo[3];
        vv8[4] = hi[0]; vv8[5] = hi[1]; vv8[6] = hi[2]; vv8[7] = hi[3];
        o[d] = MSAM2_MFMA_32x32x16(__builtin_bit_cast(op16x8, vv8), pf[st], o[d], 0, 0, 0);
      }
    }
  };
  for (int tile = 0; tile < nfull; ++tile) tile_step(tile, std::false_type{});
  if (nfull < ntiles) tile_step(nfull, std::true_type{});
  const float l_tot = half_sum(l_run);
#ifdef MSAM2_STAMP
  ws2_ = __builtin_amdgcn_s_memrealtime();
#endif
  const float inv = 1.f / l_tot;
  if (qvalid) {
  op16* ob = p.o + (int64_t)b * p.o_bs + (int64_t)head * p.o_hs + qtok * p.o_ts;
#pragma unroll
  for (int d = 0; d < DBLK; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      op16x4 wv;
#pragma unroll
      for (int e = 0; e < 4; ++e) wv[e] = f2op(o[d][4 * g + e] * inv);
      *reinterpret_cast<op16x4*>(ob + d * 32 + 8 * g + 4 * h) = wv;
    }
  }
#ifdef MSAM2_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ws3_ = __builtin_amdgcn_s_memrealtime();
  if (tid == 0) {
    const int wg = blockIdx.x + gridDim.x * blockIdx.y;
    if (wg < 4096) { g_wgtime[wg][0] = ws0_; g_wgtime[wg][1] = ws1_; g_wgtime[wg][2] = ws2_; g_wgtime[wg][3] = ws3_; }
  }
#endif
}

// (Round 4, not kept: a variant that pipelines the gather against the key tiles -- chunk c = round * threads + tid is key-major, a barrier per
//  round, the tiles that are whole run while later rounds are in flight -- was built three ways and measured with tools/win_probe.hip's stamps:
//  all rounds issued up front by LDS-DMA with per-round vmcnt waits: a CU's memory pipe serves its waves one after the other, not round by
//  round, so round 0 of the last wave lands at 6.5 us and the whole gather takes 13.5 us instead of 7.4 (24.5 us per launch against 22.6);
//  the same through staging registers, all rounds or a two-round ring: 143-165 registers live beside the accumulators, i.e. one workgroup
//  per CU (two rounds of workgroups) or spills, and a scratch reload waits on the same in-order counter as the gather.  DESIGN.md 3.3.)
// whole-window kernel: windows of 33 .. 256 queries whose K + V images (Lk + ceil8(Lk) rows of 192 B) let two workgroups share a CU
static bool attn_win_applies(const AttnParams& p) {
  const char* off = getenv("MSAM2_WIN_V1");
  if (off && off[0] == '1') return false;
  const int64_t lds = ((int64_t)p.Lk + ((p.Lk + 7) & ~7)) * 192;
  // (q-pooled windows -- a quarter of the queries against the full key set -- are gather-bound either way and measured on par with
  //  the tiled kernel: 26.7 vs 25.6 us at stage 3 -> 4; they stay there)
  return p.Lq > 32 && p.Lq <= 256 && p.Lk >= 32 && 2 * p.Lq >= p.Lk && lds <= 80 * 1024;
}

static int launch_attn_win(const AttnParams& p, int Bz, hipStream_t s) {
  const int lds = (p.Lk + ((p.Lk + 7) & ~7)) * 192;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)attn_win_kernel<96>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    attr_set = true;
  }
  // one wave per 32 queries; windows with few queries and many keys (q-pool blocks) get extra gather-only waves so that every thread
  // still moves its share of the K / V rows in one batch (they leave after the barrier)
  const int waves = min(8, max(cdiv(p.Lq, 32), cdiv((int64_t)p.Lk * 12, 6 * 64)));
  hipLaunchKernelGGL((attn_win_kernel<96>), dim3(p.H, Bz), dim3(waves * 64), lds, s, p);
  return msam2_check_launch("window_attention_fwd(win)");
}

// ------------------------------------------------------------------------------------------------------------------
// Hiera windows of 16 keys (4 x 4: the stage-2 blocks and the q-pooled stage 2 -> 3 block): one WAVE per unit of KW = 32 / Lk = 2 consecutive
// windows and one head (attn_tinywin_kernel, round 4; NT = 2 = one 64-key window per wave is built for experiments only).
// The tiled kernel gave each such window a workgroup of one wave with a two-stage LDS ring: five waves per CU, 9 KB in flight per wave,
// 39 / 67 us for 100 / 126 MB (stage-2 block, the q-pooled stage 2 -> 3 block; now 26 / 33 us).  Here a wave issues EVERY load of its unit
// first -- its 32 query rows straight into the MFMA operand layout (6 x 16 B per lane), the unit's 32 * NT key and value rows as 16-byte
// chunks (6 * NT per operand and lane) -- 18-30 KB in flight per wave and 12-16 waves per CU, then writes V into its PRIVATE LDS tile (no workgroup
// barrier anywhere; K never touches LDS: lane (r, h) loads key r's chunks 2 * st + h, the operand layout itself) and runs one or two 32-key steps.  With KW > 1 the 32 x 32 score tile is block diagonal: a query sees
// the keys of its own window, the other block is masked (the MFMAs are free here: the launch moves bytes).
// V tile [32 * NT][192 B], plain (transposed reads: the four rows of a read sit in four different 64-byte bank groups).  Windows must tile the image
// exactly (no padded tokens); anything else stays on the tiled kernel.
// ------------------------------------------------------------------------------------------------------------------
template <int D, int NT>
__global__ __launch_bounds__(256) void attn_tinywin_kernel(AttnParams p, int n_units, int KW) {
  static_assert(D == 96, "LDS bank analysis of the 192-byte rows is for D = 96");
  constexpr int RB = D * 2, CPR = D / 8, DSTEPS = D / 16, DBLK = D / 32, BK = 32;
  constexpr int ROWS = BK * NT, PER = ROWS * CPR / 64;         // K / V chunks per lane and operand (6 * NT)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = lane & 31, h = lane >> 5;
  const int z = blockIdx.x * 4 + wave;                         // (unit, head)
  if (z >= n_units * p.H) return;                              // (no workgroup barrier below)
  const int head = z % p.H, unit = z / p.H;
  const int nw = p.nwy * p.nwx;
  unsigned char* vsm = smem + wave * (ROWS * RB);

  // ---- this lane's query row: window unit * KW + r / Lq, token r % Lq of it
  const int nq = KW * p.Lq;
  const bool qvalid = r < nq;
  const int qwin = qvalid ? r / p.Lq : 0;
  int64_t qoff = 0;
  int qb_ = 0;
  {
    const int gw = unit * KW + qwin;
    qb_ = gw / nw;
    const int w = gw - qb_ * nw, wy = w / p.nwx, wx = w - wy * p.nwx;
    bool v_;
    qoff = win_token_offset(qvalid ? r - qwin * p.Lq : 0, p.ws_q, wy, wx, p.hq, p.wq, v_);
  }
  const op16* qrow = p.q + (int64_t)qb_ * p.q_bs + (int64_t)head * p.q_hs + qoff * p.q_ts;
  op16x8 qf[DSTEPS];
#pragma unroll
  for (int st = 0; st < DSTEPS; ++st) qf[st] = __builtin_bit_cast(op16x8, *reinterpret_cast<const uint4*>(qrow + st * 16 + h * 8));

  // ---- the unit's K rows straight into the MFMA operand layout (lane (r, h): key 32 * tile + r, chunks 2 * st + h), its V rows as
  //      16-byte chunks for the transposed reads: every load issued before the first LDS write
  auto key_row = [&](int row, int& b) -> int64_t {
    const int wi = row / p.Lk, t = row - wi * p.Lk;
    const int gw = unit * KW + wi;
    b = gw / nw;
    const int w = gw - b * nw, wy = w / p.nwx, wx = w - wy * p.nwx;
    bool v_;
    return win_token_offset(t, p.ws_k, wy, wx, p.hk, p.wk, v_);
  };
  op16x8 kf[NT][DSTEPS];
#pragma unroll
  for (int tile = 0; tile < NT; ++tile) {
    int b;
    const int64_t tok = key_row(tile * BK + r, b);
    const op16* krow = p.k + (int64_t)b * p.k_bs + (int64_t)head * p.k_hs + tok * p.k_ts;
#pragma unroll
    for (int st = 0; st < DSTEPS; ++st) kf[tile][st] = __builtin_bit_cast(op16x8, *reinterpret_cast<const uint4*>(krow + st * 16 + h * 8));
  }
  uint4 rv[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int c = lane + 64 * i;
    const int row = c / CPR, dc = (c - row * CPR) * 8;
    int b;
    const int64_t tok = key_row(row, b);
    rv[i] = *reinterpret_cast<const uint4*>(p.v + (int64_t)b * p.v_bs + (int64_t)head * p.v_hs + tok * p.v_ts + dc);
  }
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int c = lane + 64 * i;
    const int row = c / CPR, cc = c - row * CPR;
    *reinterpret_cast<uint4*>(vsm + row * RB + (cc << 4)) = rv[i];
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the wave reads only what it wrote itself

  f32x16 o[DBLK];
#pragma unroll
  for (int d = 0; d < DBLK; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[d][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const int li = lane & 15;
  const int v_off = (4 * h + (li >> 2)) * RB + (16 * ((lane >> 4) & 1) + 4 * (li & 3)) * 2;
#pragma unroll
  for (int tile = 0; tile < NT; ++tile) {
    const int key0 = tile * BK;
    f32x16 s;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
    for (int st = 0; st < DSTEPS; ++st) s = MSAM2_MFMA_32x32x16(kf[tile][st], qf[st], s, 0, 0, 0);
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = key0 + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (KW > 1 && key / p.Lk != qwin) s[e] = -INFINITY;      // the other windows of the unit
      mx = fmaxf(mx, s[e]);
    }
    mx = half_max(mx) * p.scale_log2;
    const float m_new = fmaxf(m_run, mx);                      // finite: every tile holds keys of the query's window
    if (NT > 1 && tile > 0) {
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < DBLK; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[d][e] *= alpha;
    }
    m_run = m_new;
    float psum = 0.f;
    op16x8 pf[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float pe = __builtin_amdgcn_exp2f(__builtin_fmaf(s[e], p.scale_log2, -m_run));
      psum += pe;
      pf[e >> 3][e & 7] = f2op_fast(pe);
    }
    l_run += psum;
#pragma unroll
    for (int d = 0; d < DBLK; ++d) {
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        typedef __attribute__((ext_vector_type(8))) short short8_t;
        const unsigned char* a0 = vsm + (key0 + 16 * st) * RB + v_off + d * 64;
        const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(a0));
        const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(a0 + 8 * RB));
        short8_t vv8;
        vv8[0] = lo[0]; vv8[1] = lo[1]; vv8[2] = lo[2]; vv8[3] = lo[3];
        vv8[4] = hi[0]; vv8[5] = hi[1]; vv8[6] = hi[2]; vv8[7] = hi[3];
        o[d] = MSAM2_MFMA_32x32x16(__builtin_bit_cast(op16x8, vv8), pf[st], o[d], 0, 0, 0);
      }
    }
  }
  const float inv = 1.f / half_sum(l_run);
  if (qvalid) {
    op16* ob = p.o + (int64_t)qb_ * p.o_bs + (int64_t)head * p.o_hs + qoff * p.o_ts;
#pragma unroll
    for (int d = 0; d < DBLK; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        op16x4 wv;
#pragma unroll
        for (int e = 0; e < 4; ++e) wv[e] = f2op(o[d][4 * g + e] * inv);
        *reinterpret_cast<op16x4*>(ob + d * 32 + 8 * g + 4 * h) = wv;
      }
  }
}

// windows of 16 keys (two per wave), tiling both token images exactly, D = 96.  (The 64-key windows of the q-pooled stage 1 -> 2 block run
// through the same code as NT = 2 -- `MSAM2_TINYWIN_64=1` -- but measured no better than the tiled kernel there: 72-77 us against 69 us,
// 251 MB; they stay on the tiled kernel.)
static bool attn_tinywin_applies(const AttnParams& p) {
  const char* off = getenv("MSAM2_NO_TINYWIN");
  if (off && off[0] == '1') return false;
  const bool exact = p.hk % p.ws_k == 0 && p.wk % p.ws_k == 0 && p.hq % p.ws_q == 0 && p.wq % p.ws_q == 0;
  if (!exact) return false;
  if (p.Lk == 64) {
    const char* on = getenv("MSAM2_TINYWIN_64");
    return on && on[0] == '1' && p.Lq <= 32;
  }
  return p.Lk == 16 && 2 * p.Lq <= 32 && (p.nwy * p.nwx) % 2 == 0;
}

static int launch_attn_tinywin(const AttnParams& p, int Bz, hipStream_t s) {
  const int KW = p.Lk == 16 ? 2 : 1, n_units = Bz / KW;
  const int64_t waves = (int64_t)n_units * p.H;
  const unsigned grid = (unsigned)((waves + 3) / 4);
  if (p.Lk == 16) hipLaunchKernelGGL((attn_tinywin_kernel<96, 1>), dim3(grid), dim3(256), 4 * 32 * 192, s, p, n_units, KW);
  else hipLaunchKernelGGL((attn_tinywin_kernel<96, 2>), dim3(grid), dim3(256), 4 * 64 * 192, s, p, n_units, KW);
  return msam2_check_launch("window_attention_fwd(tiny windows)");
}

// softmax(Q K^T * scale) V with 256-wide q / k rows and 64-wide value rows (attn_kv64_kernel): the memory cross-attention with the
// value projection folded out of the attention (O' = P M; the caller applies W_v and b_v behind it).  o: [.., 64] rows; workspace and
// merge as msam2_attention_fwd with D = 64 (msam2_attention_workspace_bytes(B, H, Lq, 64, splits), msam2_attention_merge(.., D = 64, ..)).
static int attention_kv64_impl(const void* q, const int64_t* q_strides, const void* k, const int64_t* k_strides, const void* v,
                               const int64_t* v_strides, void* o, const int64_t* o_strides, int64_t B, int64_t H, int64_t Lq,
                               int64_t Lk, float scale, int splits, int split_begin, int split_cnt, void* workspace,
                               size_t workspace_bytes, void* stream, const int* lk_dev = nullptr) {
  MSAM2_REQUIRE(q && k && v && (o || split_cnt >= 0), "attention_kv64: null tensor");
  MSAM2_REQUIRE(B > 0 && H > 0 && Lq > 0 && Lk > 0, "attention_kv64: empty problem");
  bool defer = splits < 0;
  if (defer) splits = -splits;
  MSAM2_REQUIRE(splits >= 1 && splits <= 64, "attention_kv64: bad split count %d", splits);
  for (int i = 0; i < 3; ++i)
    MSAM2_REQUIRE(q_strides[i] % 8 == 0 && k_strides[i] % 8 == 0 && v_strides[i] % 8 == 0 && (!o || o_strides[i] % 4 == 0),
                  "attention_kv64: strides must keep 16-byte row alignment");
  MSAM2_REQUIRE(Lk * k_strides[2] * 2 < (1ll << 31) && Lk * v_strides[2] * 2 < (1ll << 31), "attention_kv64: key range beyond the 2 GiB buffer window");
  const int eff = attn_effective_splits(Lk, splits);
  if (split_cnt >= 0) {   // partial launch: the caller passes the EFFECTIVE split count and a sub-range of it
    MSAM2_REQUIRE(eff == splits && eff > 1, "attention_kv64_partial: pass msam2_attention_effective_splits(Lk, splits) (> 1) as the split count");
    MSAM2_REQUIRE(split_begin >= 0 && split_cnt >= 0 && split_begin + split_cnt <= eff, "attention_kv64_partial: bad split range");
    defer = true;
    if (split_cnt == 0) return MSAM2_OK;
  } else {
    split_begin = 0;
    split_cnt = eff;
  }
  splits = eff;
  MSAM2_REQUIRE(workspace_bytes >= msam2_attention_workspace_bytes(B, H, Lq, 64, splits), "attention_kv64: workspace too small");
  MSAM2_REQUIRE(!defer || splits > 1, "attention_kv64: a deferred merge needs an effective split count > 1");
  MSAM2_REQUIRE(splits == 1 || workspace, "attention_kv64: split-KV needs a workspace");
  AttnParams p = {};
  p.q = (const op16*)q; p.k = (const op16*)k; p.v = (const op16*)v; p.o = (op16*)o;
  p.q_bs = q_strides[0]; p.q_hs = q_strides[1]; p.q_ts = q_strides[2];
  p.k_bs = k_strides[0]; p.k_hs = k_strides[1]; p.k_ts = k_strides[2];
  p.v_bs = v_strides[0]; p.v_hs = v_strides[1]; p.v_ts = v_strides[2];
  if (o) { p.o_bs = o_strides[0]; p.o_hs = o_strides[1]; p.o_ts = o_strides[2]; }
  p.B = (int)B; p.H = (int)H; p.Lq = (int)Lq; p.Lk = (int)Lk;
  p.scale_log2 = scale * 1.4426950408889634f;
  p.splits = splits;
  p.split_begin = split_begin;
  p.split_cnt = split_cnt;
  p.lk_dev = lk_dev;
  p.defer_merge = defer ? 1 : 0;
  p.o_part = (op16*)workspace;
  p.ml_part = workspace ? reinterpret_cast<float*>(p.o_part + (size_t)splits * B * H * Lq * 64) : nullptr;
  return launch_attn_kv64(p, (int)B, (hipStream_t)stream);
}

extern "C" int msam2_attention_kv64_fwd(const void* q, const int64_t* q_strides, const void* k, const int64_t* k_strides,
                                        const void* v, const int64_t* v_strides, void* o, const int64_t* o_strides, int64_t B,
                                        int64_t H, int64_t Lq, int64_t Lk, float scale, int splits, void* workspace,
                                        size_t workspace_bytes, void* stream) {
  MSAM2_REQUIRE(o, "attention_kv64: null output");
  return attention_kv64_impl(q, q_strides, k, k_strides, v, v_strides, o, o_strides, B, H, Lq, Lk, scale, splits, 0, -1, workspace, workspace_bytes, stream);
}

// The split count msam2_attention_fwd / _kv64_fwd actually run with for a requested one (every split owns >= one 32-key tile).
extern "C" int msam2_attention_effective_splits(int64_t Lk, int splits) { return attn_effective_splits(Lk, splits < 0 ? -splits : splits); }

// Splits [split_begin, split_begin + split_count) of a `splits`-way msam2_attention_kv64_fwd (splits = the EFFECTIVE count): their
// partial (max, sum, O') triples land in the same workspace slots as in the full call; nothing is merged.  The cross-GPU key split of
// the propagation chain: every rank computes its share, the slots are all-gathered, msam2_attention_merge
// (D = 64) finishes -- bit-identical to one rank computing all splits.
extern "C" int msam2_attention_kv64_partial(const void* q, const int64_t* q_strides, const void* k, const int64_t* k_strides,
                                            const void* v, const int64_t* v_strides, int64_t B, int64_t H, int64_t Lq, int64_t Lk,
                                            float scale, int splits, int split_begin, int split_count, void* workspace,
                                            size_t workspace_bytes, void* stream) {
  MSAM2_REQUIRE(split_count >= 0, "attention_kv64_partial: negative split count");
  return attention_kv64_impl(q, q_strides, k, k_strides, v, v_strides, nullptr, nullptr, B, H, Lq, Lk, scale, splits, split_begin, split_count,
                             workspace, workspace_bytes, stream);
}

// msam2_attention_kv64_fwd / _partial with the key count read ON THE DEVICE: Lk is the capacity (buffers, split count and workspace
// are sized for it, rows [*key_count_dev, Lk) of k / v must be readable but are never used), *key_count_dev in [1, Lk] the number
// of keys attended to.  The launch is the same for every fill level, so a hipGraph captured once per memory-bank bucket is replayed
// while the bank's object-pointer tail grows (volume.GraphedPropagation).  With *key_count_dev == Lk the result is bit-identical to
// the host-count entries; below it, the split boundaries follow the device count exactly as a host-count call with that Lk and the
// same effective split count would place them.
extern "C" int msam2_attention_kv64_dyn_fwd(const void* q, const int64_t* q_strides, const void* k, const int64_t* k_strides,
                                            const void* v, const int64_t* v_strides, void* o, const int64_t* o_strides, int64_t B,
                                            int64_t H, int64_t Lq, int64_t Lk, const void* key_count_dev, float scale, int splits,
                                            void* workspace, size_t workspace_bytes, void* stream) {
  MSAM2_REQUIRE(o && key_count_dev, "attention_kv64_dyn: null output / key count");
  return attention_kv64_impl(q, q_strides, k, k_strides, v, v_strides, o, o_strides, B, H, Lq, Lk, scale, splits, 0, -1, workspace, workspace_bytes,
                             stream, (const int*)key_count_dev);
}

extern "C" int msam2_attention_kv64_dyn_partial(const void* q, const int64_t* q_strides, const void* k, const int64_t* k_strides,
                                                const void* v, const int64_t* v_strides, int64_t B, int64_t H, int64_t Lq, int64_t Lk,
                                                const void* key_count_dev, float scale, int splits, int split_begin, int split_count,
                                                void* workspace, size_t workspace_bytes, void* stream) {
  MSAM2_REQUIRE(split_count >= 0 && key_count_dev, "attention_kv64_dyn_partial: negative split count / null key count");
  return attention_kv64_impl(q, q_strides, k, k_strides, v, v_strides, nullptr, nullptr, B, H, Lq, Lk, scale, splits, split_begin, split_count,
                             workspace, workspace_bytes, stream, (const int*)key_count_dev);
}

// Windowed attention straight from the un-partitioned token image (replaces window_partition + SDPA +
// window_unpartition, backbones/utils.py:16-62 + hieradet.py:72-76).  qkv tokens live at
// base + ((b*h + y)*w + x)*token_stride + head*head_stride; windows are ws x ws, zero-padded at the bottom/right, where a
// padded token's key/value rows are kpad/vpad (the qkv bias slices, because LayerNorm'ed input was zero-padded).
extern "C" int msam2_window_attention_fwd(const void* q, int64_t q_token_stride, int64_t q_head_stride, int64_t hq, int64_t wq,
                                          int64_t ws_q, const void* k, const void* v, int64_t kv_token_stride,
                                          int64_t kv_head_stride, int64_t hk, int64_t wk, int64_t ws_k, const float* kpad,
                                          const float* vpad, void* o, int64_t o_token_stride, int64_t o_head_stride, int64_t B,
                                          int64_t H, int64_t D, float scale, void* stream) {
  MSAM2_REQUIRE(q && k && v && o && kpad && vpad, "window_attention: null tensor");
  MSAM2_REQUIRE(D == 96 || D == 64 || D == 128, "window_attention: head dim %lld not built", (long long)D);
  MSAM2_REQUIRE(ws_q > 0 && ws_k > 0 && hq > 0 && wq > 0 && hk > 0 && wk > 0, "window_attention: bad geometry");
  const int nwy = (int)((hk + ws_k - 1) / ws_k), nwx = (int)((wk + ws_k - 1) / ws_k);
  MSAM2_REQUIRE((hq + ws_q - 1) / ws_q == nwy && (wq + ws_q - 1) / ws_q == nwx, "window_attention: q/kv window grids differ");
  MSAM2_REQUIRE(q_token_stride % 8 == 0 && kv_token_stride % 8 == 0 && q_head_stride % 8 == 0 && kv_head_stride % 8 == 0 &&
                    o_token_stride % 4 == 0 && o_head_stride % 4 == 0, "window_attention: strides must keep 16-byte alignment");
  AttnParams p = {};
  p.q = (const op16*)q; p.k = (const op16*)k; p.v = (const op16*)v; p.o = (op16*)o;
  p.q_bs = hq * wq * q_token_stride; p.q_hs = q_head_stride; p.q_ts = q_token_stride;
  p.k_bs = hk * wk * kv_token_stride; p.k_hs = kv_head_stride; p.k_ts = kv_token_stride;
  p.v_bs = p.k_bs; p.v_hs = kv_head_stride; p.v_ts = kv_token_stride;
  p.o_bs = hq * wq * o_token_stride; p.o_hs = o_head_stride; p.o_ts = o_token_stride;
  p.B = (int)B; p.H = (int)H; p.Lq = (int)(ws_q * ws_q); p.Lk = (int)(ws_k * ws_k);
  p.scale_log2 = scale * 1.4426950408889634f;
  p.win = 1; p.ws_q = (int)ws_q; p.ws_k = (int)ws_k; p.hq = (int)hq; p.wq = (int)wq; p.hk = (int)hk; p.wk = (int)wk;
  p.nwy = nwy; p.nwx = nwx; p.kpad = kpad; p.vpad = vpad; p.splits = 1;
  hipStream_t s = (hipStream_t)stream;
  const int Bz = (int)B * nwy * nwx;
  switch (D) {
    case 96:
      if (attn_win_applies(p)) return launch_attn_win(p, Bz, s);
      return attn_tinywin_applies(p) ? launch_attn_tinywin(p, Bz, s) : dispatch_nw<96, true>(p, Bz, s);
    case 64: return dispatch_nw<64, true>(p, Bz, s);
    default: return dispatch_nw<128, true>(p, Bz, s);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Small-head attention for the two-way decoder (transformer.py:239-263): D in {16, 32}, 8 heads, either very few
// queries (tokens -> image, Lq ~ 8, Lk = 4096) or very few keys (image -> tokens).  One wave per (batch, head, query);
// lanes stride over the keys with a private online softmax, then a butterfly merge.  fp32 math, op16 I/O.
// ------------------------------------------------------------------------------------------------------------------
template <int D>
__global__ void attn_small_kernel(const op16* q, const op16* k, const op16* v, op16* o, int64_t q_bs, int64_t q_ts,
                                  int64_t k_bs, int64_t k_ts, int64_t v_bs, int64_t v_ts, int64_t o_bs, int64_t o_ts, int B,
                                  int H, int Lq, int Lk, float scale_log2) {
  const int64_t gw = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (gw >= (int64_t)B * H * Lq) return;
  const int qi = gw % Lq;
  const int head = (gw / Lq) % H;
  const int b = gw / ((int64_t)Lq * H);
  float qv[D];
  const op16* qp = q + b * q_bs + (int64_t)qi * q_ts + head * D;
#pragma unroll
  for (int d = 0; d < D; d += 8) {
    const op16x8 t = *reinterpret_cast<const op16x8*>(qp + d);
#pragma unroll
    for (int e = 0; e < 8; ++e) qv[d + e] = op2f(t[e]) * scale_log2;
  }
  float m = -INFINITY, l = 0.f, acc[D];
#pragma unroll
  for (int d = 0; d < D; ++d) acc[d] = 0.f;
  for (int key = lane; key < Lk; key += 64) {
    const op16* kp = k + b * k_bs + (int64_t)key * k_ts + head * D;
    const op16* vp = v + b * v_bs + (int64_t)key * v_ts + head * D;
    float s = 0.f;
    float vv[D];
#pragma unroll
    for (int d = 0; d < D; d += 8) {
      const op16x8 t = *reinterpret_cast<const op16x8*>(kp + d);
      const op16x8 u = *reinterpret_cast<const op16x8*>(vp + d);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        s += qv[d + e] * op2f(t[e]);
        vv[d + e] = op2f(u[e]);
      }
    }
    const float mn = fmaxf(m, s);
    const float a = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m - mn);
    const float pe = __builtin_amdgcn_exp2f(s - mn);
    l = l * a + pe;
#pragma unroll
    for (int d = 0; d < D; ++d) acc[d] = acc[d] * a + pe * vv[d];
    m = mn;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float m2 = __shfl_xor(m, off, 64), l2 = __shfl_xor(l, off, 64);
    const float mn = fmaxf(m, m2);
    const float a1 = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m - mn);
    const float a2 = (m2 == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m2 - mn);
    l = l * a1 + l2 * a2;
#pragma unroll
    for (int d = 0; d < D; ++d) acc[d] = acc[d] * a1 + __shfl_xor(acc[d], off, 64) * a2;
    m = mn;
  }
  if (lane == 0) {
    op16* op = o + b * o_bs + (int64_t)qi * o_ts + head * D;
    const float inv = 1.f / l;
#pragma unroll
    for (int d = 0; d < D; ++d) op[d] = f2op(acc[d] * inv);
  }
}

// Few-queries form (tokens -> image, Lq ~ 8, Lk = 4096): one 1024-thread workgroup per (batch, head, query).  With one wave per
// query the 4096 keys were a 64-trip chain of dependent 32-byte loads (45 us of pure latency); here every lane owns Lk/1024
// keys whose loads are all in flight at once, followed by a wave butterfly and a 16-way merge through LDS.
template <int D>
__global__ __launch_bounds__(1024) void attn_fewq_kernel(const op16* q, const op16* k, const op16* v, op16* o, int64_t q_bs, int64_t q_ts,
                                                         int64_t k_bs, int64_t k_ts, int64_t v_bs, int64_t v_ts, int64_t o_bs, int64_t o_ts,
                                                         int B, int H, int Lq, int Lk, float scale_log2) {
  __shared__ float part[16][D + 2];
  const int gw = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qi = gw % Lq;
  const int head = (gw / Lq) % H;
  const int b = gw / (Lq * H);
  float qv[D];
  const op16* qp = q + b * q_bs + (int64_t)qi * q_ts + head * D;
#pragma unroll
  for (int d = 0; d < D; d += 8) {
    const op16x8 t = *reinterpret_cast<const op16x8*>(qp + d);
#pragma unroll
    for (int e = 0; e < 8; ++e) qv[d + e] = op2f(t[e]) * scale_log2;
  }
  float m = -INFINITY, l = 0.f, acc[D];
#pragma unroll
  for (int d = 0; d < D; ++d) acc[d] = 0.f;
  const op16* kb = k + b * k_bs + head * D;
  const op16* vb = v + b * v_bs + head * D;
#pragma unroll 4
  for (int key = threadIdx.x; key < Lk; key += 1024) {
    const op16* kp = kb + (int64_t)key * k_ts;
    const op16* vp = vb + (int64_t)key * v_ts;
    float s = 0.f;
    float vv[D];
#pragma unroll
    for (int d = 0; d < D; d += 8) {
      const op16x8 t = *reinterpret_cast<const op16x8*>(kp + d);
      const op16x8 u = *reinterpret_cast<const op16x8*>(vp + d);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        s += qv[d + e] * op2f(t[e]);
        vv[d + e] = op2f(u[e]);
      }
    }
    const float mn = fmaxf(m, s);
    const float a = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m - mn);
    const float pe = __builtin_amdgcn_exp2f(s - mn);
    l = l * a + pe;
#pragma unroll
    for (int d = 0; d < D; ++d) acc[d] = acc[d] * a + pe * vv[d];
    m = mn;
  }
  auto merge = [&](int off) {
    const float m2 = __shfl_xor(m, off, 64), l2 = __shfl_xor(l, off, 64);
    const float mn = fmaxf(m, m2);
    const float a1 = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m - mn);
    const float a2 = (m2 == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m2 - mn);
    l = l * a1 + l2 * a2;
#pragma unroll
    for (int d = 0; d < D; ++d) acc[d] = acc[d] * a1 + __shfl_xor(acc[d], off, 64) * a2;
    m = mn;
  };
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) merge(off);
  if (lane == 0) {
    part[wave][0] = m;
    part[wave][1] = l;
#pragma unroll
    for (int d = 0; d < D; ++d) part[wave][2 + d] = acc[d];
  }
  __syncthreads();
  if (wave == 0) {
    const int w = lane & 15;
    m = part[w][0];
    l = part[w][1];
#pragma unroll
    for (int d = 0; d < D; ++d) acc[d] = part[w][2 + d];
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) merge(off);
    if (lane == 0) {
      op16* op = o + b * o_bs + (int64_t)qi * o_ts + head * D;
      const float inv = 1.f / l;
#pragma unroll
      for (int d = 0; d < D; ++d) op[d] = f2op(acc[d] * inv);
    }
  }
}

// q/k/v/o: op16 [B, L, H*D] with element strides {batch, token}
// Few-keys form (image -> tokens, Lk <= 32): one THREAD per (batch, head, query); the handful of keys/values of a (batch, head)
// is read by every thread of it (L1 broadcast), scores and the softmax live in registers.
template <int D>
__global__ void attn_fewkeys_kernel(const op16* q, const op16* k, const op16* v, op16* o, int64_t q_bs, int64_t q_ts, int64_t k_bs,
                                    int64_t k_ts, int64_t v_bs, int64_t v_ts, int64_t o_bs, int64_t o_ts, int B, int H, int Lq, int Lk,
                                    float scale_log2) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)B * H * Lq) return;
  // consecutive threads walk the heads of one query first: their q/o accesses tile a contiguous H*D row
  const int head = gid % H;
  const int qi = (gid / H) % Lq;
  const int b = gid / ((int64_t)H * Lq);
  float qv[D];
  const op16* qp = q + b * q_bs + (int64_t)qi * q_ts + head * D;
#pragma unroll
  for (int d = 0; d < D; d += 8) {
    const op16x8 t = *reinterpret_cast<const op16x8*>(qp + d);
#pragma unroll
    for (int e = 0; e < 8; ++e) qv[d + e] = op2f(t[e]) * scale_log2;
  }
  float m = -INFINITY, l = 0.f, acc[D];
#pragma unroll
  for (int d = 0; d < D; ++d) acc[d] = 0.f;
  for (int key = 0; key < Lk; ++key) {
    const op16* kp = k + b * k_bs + (int64_t)key * k_ts + head * D;
    const op16* vp = v + b * v_bs + (int64_t)key * v_ts + head * D;
    float sc = 0.f;
#pragma unroll
    for (int d = 0; d < D; d += 8) {
      const op16x8 t = *reinterpret_cast<const op16x8*>(kp + d);
#pragma unroll
      for (int e = 0; e < 8; ++e) sc += qv[d + e] * op2f(t[e]);
    }
    const float mn = fmaxf(m, sc);
    const float al = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m - mn);
    const float pe = __builtin_amdgcn_exp2f(sc - mn);
    l = l * al + pe;
#pragma unroll
    for (int d = 0; d < D; d += 8) {
      const op16x8 u = *reinterpret_cast<const op16x8*>(vp + d);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[d + e] = acc[d + e] * al + pe * op2f(u[e]);
    }
    m = mn;
  }
  const float inv = 1.f / l;
  op16* op = o + b * o_bs + (int64_t)qi * o_ts + head * D;
#pragma unroll
  for (int d = 0; d < D; d += 8) {
    op16x8 w;
#pragma unroll
    for (int e = 0; e < 8; ++e) w[e] = f2op(acc[d + e] * inv);
    *reinterpret_cast<op16x8*>(op + d) = w;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// tokens -> image attention of the two-way decoder (a handful of queries against 4096 image keys, 8 heads of 16 channels) on the matrix
// pipe: ONE 1024-thread workgroup per (batch, head) serves ALL its queries (round 4).  attn_fewq_kernel gives every (batch, head, query) a
// workgroup of its own, so the head's K and V (262 KB) are read once per query -- 9 times -- by 288 workgroups in two rounds: 23 us.
// Here wave w owns keys [w * Lk / 16, ...): per 32-key step ONE MFMA gives the scores of all (<= 32) queries (S^T = K Q^T, the 16 channels
// are the whole reduction), the softmax runs per lane (lane = query), and O^T += V^T P^T is two more MFMAs with V^T read transposed from
// a private LDS tile ([32 keys][64 B]: 32 channel slots, the upper 16 zero, so that attn_win_kernel's transposed-read addressing applies
// unchanged).  Every K fragment and V chunk of the wave's key range is loaded up front (8 + 8 loads per lane in flight).  The 16 waves'
// (max, sum, O) are merged through LDS by the first 16 * Lq threads.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void attn_fewq16_kernel(const op16* __restrict__ q, const op16* __restrict__ k, const op16* __restrict__ v,
                                                           op16* __restrict__ o, int64_t q_bs, int64_t q_ts, int64_t k_bs, int64_t k_ts,
                                                           int64_t v_bs, int64_t v_ts, int64_t o_bs, int64_t o_ts, int H, int Lq, int Lk,
                                                           float scale_log2) {
  constexpr int D = 16, RB = 64, BK = 32, NW = 16, MAXS = 8;  // MAXS 32-key steps per wave: Lk <= 16 * 8 * 32 = 4096 (launcher)
  __shared__ __attribute__((aligned(16))) unsigned char vt[NW][BK * RB];        // per-wave V tile
  __shared__ float part[NW][32][18];                                          // per wave and query: m, l, O[0..15]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, h = lane >> 5;
  const int b = blockIdx.x / H, head = blockIdx.x - b * H;
  const op16* qb = q + (int64_t)b * q_bs + head * D;
  const op16* kb = k + (int64_t)b * k_bs + head * D;
  const op16* vb = v + (int64_t)b * v_bs + head * D;
  // upper 16 channel slots of every V row: zero, once
  {
    unsigned char* t = vt[wave];
    *reinterpret_cast<uint4*>(t + (lane >> 1) * RB + 32 + (lane & 1) * 16) = make_uint4(0, 0, 0, 0);
  }
  // query fragment (B operand): lane (r = query, h) holds channels 8 h .. 8 h + 7
  op16x8 qf;
  {
    uint4 t = make_uint4(0, 0, 0, 0);
    const uint4 ld = *reinterpret_cast<const uint4*>(qb + (int64_t)min(r, Lq - 1) * q_ts + h * 8);
    if (r < Lq) t = ld;
    qf = __builtin_bit_cast(op16x8, t);
  }
  // this wave's keys: steps of 32, every load issued before the first use (row indices clamped; masked below)
  const int per = (Lk + NW - 1) / NW;                         // keys per wave
  const int key_lo = wave * per, key_hi = min(Lk, key_lo + per);
  const int nsteps = max(0, (key_hi - key_lo + BK - 1) / BK);
  uint4 kfr[MAXS], vch[MAXS];
#pragma unroll
  for (int st = 0; st < MAXS; ++st) {
    const int kk = min(key_lo + st * BK + r, Lk - 1);                           // K fragment row (A operand): key r of the step
    kfr[st] = *reinterpret_cast<const uint4*>(kb + (int64_t)kk * k_ts + h * 8);
    const int kv = min(key_lo + st * BK + (lane >> 1), Lk - 1);                 // V chunk: key lane / 2, channel half lane & 1
    vch[st] = *reinterpret_cast<const uint4*>(vb + (int64_t)kv * v_ts + (lane & 1) * 8);
  }
  f32x16 oacc;
#pragma unroll
  for (int e = 0; e < 16; ++e) oacc[e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const int li = lane & 15;
  const int v_off = (4 * h + (li >> 2)) * RB + (16 * ((lane >> 4) & 1) + 4 * (li & 3)) * 2;
#pragma unroll
  for (int st = 0; st < MAXS; ++st) {
    if (st < nsteps) {                                                           // (wave-uniform)
      const int key0 = key_lo + st * BK;
      *reinterpret_cast<uint4*>(vt[wave] + (lane >> 1) * RB + (lane & 1) * 16) = vch[st];
      f32x16 sacc;
#pragma unroll
      for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
      sacc = MSAM2_MFMA_32x32x16(__builtin_bit_cast(op16x8, kfr[st]), qf, sacc, 0, 0, 0);
      float mx = -INFINITY;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = key0 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (key >= key_hi) sacc[e] = -INFINITY;
        mx = fmaxf(mx, sacc[e]);
      }
      mx = half_max(mx) * scale_log2;                                            // finite: every step holds >= 1 key of the range
      const float m_new = fmaxf(m_run, mx);
      const float alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int e = 0; e < 16; ++e) oacc[e] *= alpha;
      m_run = m_new;
      float psum = 0.f;
      op16x8 pf[2];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float pe = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[e], scale_log2, -m_run));
        psum += pe;
        pf[e >> 3][e & 7] = f2op_fast(pe);
      }
      l_run += psum;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                         // the wave's own V tile is in LDS
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        typedef __attribute__((ext_vector_type(8))) short short8_t;
        const unsigned char* a0 = vt[wave] + (16 * s2) * RB + v_off;
        const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(a0));
        const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(a0 + 8 * RB));
        short8_t vv8;
        vv8[0] = lo[0]; vv8[1] = lo[1]; vv8[2] = lo[2]; vv8[3] = lo[3];
        vv8[4] = hi[0]; vv8[5] = hi[1]; vv8[6] = hi[2]; vv8[7] = hi[3];
        oacc = MSAM2_MFMA_32x32x16(__builtin_bit_cast(op16x8, vv8), pf[s2], oacc, 0, 0, 0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                         // tile read before the next step overwrites it
    }
  }
  // ---- per wave: (m, l, O rows d = (e & 3) + 8 (e >> 2) + 4 h for e < 8) of query r; merged by thread (query, channel)
  const float l_tot = half_sum(l_run);
  {
    float* pp = part[wave][r];
    if (h == 0) { pp[0] = m_run; pp[1] = l_tot; }
#pragma unroll
    for (int e = 0; e < 8; ++e) pp[2 + (e & 3) + 8 * (e >> 2) + 4 * h] = oacc[e];
  }
  __syncthreads();
  if (tid < Lq * D) {
    const int qi = tid / D, d = tid - qi * D;
    float M = -INFINITY;
#pragma unroll
    for (int w = 0; w < NW; ++w) M = fmaxf(M, part[w][qi][0]);
    float L = 0.f, acc = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float mw = part[w][qi][0];
      const float wgt = (mw == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(mw - M);
      L += part[w][qi][1] * wgt;
      acc += part[w][qi][2 + d] * wgt;
    }
    o[(int64_t)b * o_bs + (int64_t)qi * o_ts + head * D + d] = f2op(acc / L);
  }
}


extern "C" int msam2_attention_small_fwd(const void* q, int64_t q_bs, int64_t q_ts, const void* k, int64_t k_bs, int64_t k_ts,
                                         const void* v, int64_t v_bs, int64_t v_ts, void* o, int64_t o_bs, int64_t o_ts,
                                         int64_t B, int64_t H, int64_t Lq, int64_t Lk, int64_t D, float scale, void* stream) {
  MSAM2_REQUIRE(q && k && v && o, "attention_small: null tensor");
  MSAM2_REQUIRE(D == 16 || D == 32, "attention_small: head dim %lld not built (16/32)", (long long)D);
  MSAM2_REQUIRE(B > 0 && H > 0 && Lq > 0 && Lk > 0, "attention_small: empty problem");
  MSAM2_REQUIRE(q_ts % 8 == 0 && k_ts % 8 == 0 && v_ts % 8 == 0 && q_bs % 8 == 0 && k_bs % 8 == 0 && v_bs % 8 == 0,
                "attention_small: strides must keep 16-byte alignment");
  const float sl = scale * 1.4426950408889634f;
  hipStream_t s = (hipStream_t)stream;
  if (Lk <= 32 && Lq >= 64 && o_ts % 8 == 0 && o_bs % 8 == 0) {
    dim3 g1(cdiv(B * H * Lq, 256));
    if (D == 16)
      hipLaunchKernelGGL((attn_fewkeys_kernel<16>), g1, dim3(256), 0, s, (const op16*)q, (const op16*)k, (const op16*)v, (op16*)o, q_bs,
                         q_ts, k_bs, k_ts, v_bs, v_ts, o_bs, o_ts, (int)B, (int)H, (int)Lq, (int)Lk, sl);
    else
      hipLaunchKernelGGL((attn_fewkeys_kernel<32>), g1, dim3(256), 0, s, (const op16*)q, (const op16*)k, (const op16*)v, (op16*)o, q_bs,
                         q_ts, k_bs, k_ts, v_bs, v_ts, o_bs, o_ts, (int)B, (int)H, (int)Lq, (int)Lk, sl);
    return msam2_check_launch("attention_small(fewkeys)");
  }
  {
    const char* off = getenv("MSAM2_NO_FEWQ16");
    if (D == 16 && Lq <= 32 && Lk >= 1024 && Lk <= 4096 && !(off && off[0] == '1') && q_ts % 8 == 0 && k_ts % 8 == 0 && v_ts % 8 == 0 &&
        q_bs % 8 == 0 && k_bs % 8 == 0 && v_bs % 8 == 0 && (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0) {
      hipLaunchKernelGGL(attn_fewq16_kernel, dim3((unsigned)(B * H)), dim3(1024), 0, s, (const op16*)q, (const op16*)k, (const op16*)v, (op16*)o,
                         q_bs, q_ts, k_bs, k_ts, v_bs, v_ts, o_bs, o_ts, (int)H, (int)Lq, (int)Lk, sl);
      return msam2_check_launch("attention_small(fewq, matrix pipe)");
    }
  }
  if (Lk >= 1024 && B * H * Lq <= 4096) {
    dim3 g2((unsigned)(B * H * Lq));
    if (D == 16)
      hipLaunchKernelGGL((attn_fewq_kernel<16>), g2, dim3(1024), 0, s, (const op16*)q, (const op16*)k, (const op16*)v, (op16*)o, q_bs, q_ts,
                         k_bs, k_ts, v_bs, v_ts, o_bs, o_ts, (int)B, (int)H, (int)Lq, (int)Lk, sl);
    else
      hipLaunchKernelGGL((attn_fewq_kernel<32>), g2, dim3(1024), 0, s, (const op16*)q, (const op16*)k, (const op16*)v, (op16*)o, q_bs, q_ts,
                         k_bs, k_ts, v_bs, v_ts, o_bs, o_ts, (int)B, (int)H, (int)Lq, (int)Lk, sl);
    return msam2_check_launch("attention_small(fewq)");
  }
  const int64_t waves = B * H * Lq;
  dim3 grid(cdiv(waves * 64, 256));
  if (D == 16)
    hipLaunchKernelGGL((attn_small_kernel<16>), grid, dim3(256), 0, s, (const op16*)q, (const op16*)k, (const op16*)v, (op16*)o,
                       q_bs, q_ts, k_bs, k_ts, v_bs, v_ts, o_bs, o_ts, (int)B, (int)H, (int)Lq, (int)Lk, sl);
  else
    hipLaunchKernelGGL((attn_small_kernel<32>), grid, dim3(256), 0, s, (const op16*)q, (const op16*)k, (const op16*)v, (op16*)o,
                       q_bs, q_ts, k_bs, k_ts, v_bs, v_ts, o_bs, o_ts, (int)B, (int)H, (int)Lq, (int)Lk, sl);
  return msam2_check_launch("attention_small");
}
