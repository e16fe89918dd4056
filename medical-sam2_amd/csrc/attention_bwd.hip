// Flash-style (O(L) memory) attention backward for gfx950: dQ, dK, dV of softmax(Q K^T * scale) V without materialising the
// [Lq, Lk] scores.  Replaces torch.autograd of F.scaled_dot_product_attention at transformer.py:318 (RoPEAttention of the memory
// attention, D = 256) and hieradet.py:72-76 (global Hiera attention, D = 64 / 96 / 128) in the training loops
// (func_3d/function.py:182-191, func_2d/function.py:246-259).
//
// One kernel structure, three roles.  As in the forward kernel (attention.hip) a lane OWNS one column of the score tile: its row
// of the owner tensor is held in registers as the MFMA B operand, the other tensor is streamed through LDS in 32-row tiles, and
// the accumulator of the first product is fed straight back as the B operand of the second one (A = transposed tile fragments
// via ds_read_b64_tr_b16), so no score ever leaves the registers:
//
//   role   owner (regs)   streamed (LDS)   first products                      second product             statistics
//   DQ     q, dO          k, v             S^T = K Q^T,  dP^T = V dO^T         dQ^T += K^T  dS^T          lse / delta of the lane's own row;
//                                                                                                         keys split over workgroups
//   DK     k, v           q, dO            S   = Q K^T,  dP   = dO V^T         dK^T += Q^T  dS            lse / delta per streamed row
//   DV     k              q, dO            S   = Q K^T                         dV^T += dO^T P             lse per streamed row
//
// delta = rowsum(dO o O) comes from a pre-pass that also converts dO to the 16-bit operand type; the log-sum-exp of every query
// row comes from the forward (msam2_attention_fwd_lse).  dK / dV rows have exactly one owner; the DQ pass has few owner workgroups
// when Lq is short against Lk (4096 queries x 16k keys), so its key range is split over workgroups whose partial dQ go to the
// workspace and are summed by a small reduction pass (no atomics anywhere: the gradients are run-to-run reproducible).  Cost: 8 score-tile products against the minimum of 5 -- the price of
// three simple passes (a first version of the DQ pass that found the statistics itself with an online softmax spilled 190 VGPRs
// and ran 3.5x slower than the DK pass on the same flops).
#include "common.h"
#include <stdlib.h>

namespace {

struct AttnBwdParams {
  const op16 *q, *k, *v, *do16;                       // do16: [B, H, Lq, D] contiguous 16-bit copy of dO (workspace)
  int64_t q_bs, q_hs, q_ts, k_bs, k_hs, k_ts, v_bs, v_hs, v_ts;
  const float* delta;                                 // [B, H, Lq]
  const float* lse;                                   // [B, H, Lq], log2 domain (from the forward)
  int ksplit;                                         // DQ role: workgroups per owner block over the streamed keys
  float* dq_part;                                     // DQ role, ksplit > 1: partial dQ [ksplit][B*H*Lq][D] (workspace)
  float *dq, *dk, *dv;
  int64_t dq_bs, dq_hs, dq_ts, dk_bs, dk_hs, dk_ts, dv_bs, dv_hs, dv_ts;
  int B, H, Lq, Lk;
  float scale_log2, scale;
  AttnDropout drop;                                   // attention-probability dropout of the forward (thr == 0: none)
};

constexpr int ROLE_DQ = 0, ROLE_DK = 1, ROLE_DV = 2;

template <int D>
struct BwdCfg {
  static constexpr int BK = 32;
  static constexpr int KS = D * 2 + 16;                                  // slot-1 tile row stride (bytes)
  static constexpr int VS = D * 2 + (((D * 2) % 128 == 64) ? 0 : 64);    // slot-2 tile row stride: VS % 128 == 64
  static constexpr int STAGE = BK * (KS + (VS > KS ? VS : KS));          // slot 2 is pitched KS or VS depending on the role
  static constexpr int STATS = 2 * 2 * BK * 4;                           // (lse, delta) of the streamed rows, double buffered
  static constexpr int LDS_BYTES = 2 * STAGE + STATS;
};

// delta[row] = sum_d dO[row][d] * O[row][d];  do16[row] = (op16) dO[row].  One wave per row.
template <int D>
__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(const float* __restrict__ d_o, int64_t do_bs, int64_t do_hs, int64_t do_ts,
                                                            const op16* __restrict__ o, int64_t o_bs, int64_t o_hs, int64_t o_ts,
                                                            op16* __restrict__ do16, float* __restrict__ delta, int H, int Lq, int64_t rows) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const int64_t t = row % Lq, bh = row / Lq;
  const int64_t b = bh / H, h = bh % H;
  float acc = 0.f;
  if (lane * 4 < D) {
    const f32x4 g = *reinterpret_cast<const f32x4*>(d_o + b * do_bs + h * do_hs + t * do_ts + lane * 4);
    const op16x4 ov = *reinterpret_cast<const op16x4*>(o + b * o_bs + h * o_hs + t * o_ts + lane * 4);
    op16x4 g16;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc += g[e] * op2f(ov[e]);
      g16[e] = f2op(g[e]);
    }
    *reinterpret_cast<op16x4*>(do16 + row * D + lane * 4) = g16;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (lane == 0) delta[row] = acc;
}

template <int D, int NW, int ROLE, bool DROP>
__global__ __launch_bounds__(NW * 64, 1) void attn_bwd_kernel(AttnBwdParams p) {
  using C = BwdCfg<D>;
  constexpr int NT = NW * 64;
  constexpr int DSTEPS = D / 16;   // k-steps of the first products
  constexpr int DBLK = D / 32;     // 32-row blocks of the transposed output
  constexpr int CPR = D / 8;       // 16-byte chunks per tile row
  constexpr int CHUNKS = C::BK * CPR;
  constexpr int PER = (CHUNKS + NT - 1) / NT;
  // slot-2 row pitch by role: the DV role reads it transposed (conflict-free at pitch = 64 mod 128 bytes), the other two read it row
  // by row with ds_read_b128 (conflict-free at the slot-1 pitch; 4-way conflicts at the other one)
  constexpr int VS2 = ROLE == ROLE_DV ? C::VS : C::KS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* stats = reinterpret_cast<float*>(smem + 2 * C::STAGE);          // [2 buffers][lse 32 | delta 32]

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, h = lane >> 5;
  // XCD-aware placement (as in the forward): workgroups are dealt round-robin over the 8 XCDs, each with a private L2; all owner
  // blocks of one (batch, head) stream the SAME rows, so they go to one XCD (a contiguous slice of the remapped id space).
  const int gx = gridDim.x, gy = gridDim.y;
  int lid = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
  {
    const int nwg = gx * gy * gridDim.z;
    const int q8 = nwg / 8, rem = nwg % 8, xcd = lid % 8;
    lid = (xcd < rem ? xcd * (q8 + 1) : rem * (q8 + 1) + (xcd - rem) * q8) + lid / 8;
  }
  const int bx = lid % gx, head = (lid / gx) % gy, b = lid / (gx * gy);
  const int64_t bh = (int64_t)b * p.H + head;

  // ---- roles: owner rows (registers) and streamed rows (LDS slots 1 / 2)
  const int n_own = ROLE == ROLE_DQ ? p.Lq : p.Lk;
  const int n_str = ROLE == ROLE_DQ ? p.Lk : p.Lq;
  const op16* qb = p.q + b * p.q_bs + head * p.q_hs;
  const op16* kb = p.k + b * p.k_bs + head * p.k_hs;
  const op16* vb = p.v + b * p.v_bs + head * p.v_hs;
  const op16* gb = p.do16 + bh * p.Lq * D;
  const op16* own1 = ROLE == ROLE_DQ ? qb : kb;
  const int64_t own1_ts = ROLE == ROLE_DQ ? p.q_ts : p.k_ts;
  const op16* own2 = ROLE == ROLE_DQ ? gb : vb;
  const int64_t own2_ts = ROLE == ROLE_DQ ? (int64_t)D : p.v_ts;
  const op16* str1 = ROLE == ROLE_DQ ? kb : qb;
  const int64_t str1_ts = ROLE == ROLE_DQ ? p.k_ts : p.q_ts;
  const op16* str2 = ROLE == ROLE_DQ ? vb : gb;
  const int64_t str2_ts = ROLE == ROLE_DQ ? p.v_ts : (int64_t)D;

  // ---- this lane's owner row -> B-operand fragments kept in registers
  const int nsplit = ROLE == ROLE_DQ ? p.ksplit : 1;
  const int nob = gx / nsplit;                         // owner blocks; neighbouring workgroups share a key range (L2 reuse), not an owner block
  const int split = bx / nob, oblk = bx - split * nob;
  const int oi = oblk * (NW * 32) + wave * 32 + r;
  const bool ovalid = oi < n_own;
  op16x8 f1[DSTEPS], f2[ROLE == ROLE_DV ? 1 : DSTEPS];
#pragma unroll
  for (int s = 0; s < DSTEPS; ++s) {
    uint4 a = make_uint4(0, 0, 0, 0), c = make_uint4(0, 0, 0, 0);
    if (ovalid) {
      a = *reinterpret_cast<const uint4*>(own1 + (int64_t)oi * own1_ts + s * 16 + h * 8);
      if constexpr (ROLE != ROLE_DV) c = *reinterpret_cast<const uint4*>(own2 + (int64_t)oi * own2_ts + s * 16 + h * 8);
    }
    f1[s] = __builtin_bit_cast(op16x8, a);
    if constexpr (ROLE != ROLE_DV) f2[s] = __builtin_bit_cast(op16x8, c);
  }
  const float delta_own = (ROLE == ROLE_DQ && ovalid) ? p.delta[bh * p.Lq + oi] : 0.f;
  const float lse_own = (ROLE == ROLE_DQ && ovalid) ? p.lse[bh * p.Lq + oi] : 0.f;
  // dropout stream: DQ owns a query row (element index of key 0 of that row), DK / DV own a key (element index of query 0, that key)
  // (DROP instances only: as a run-time branch the mask generator cost every instance 12-90 registers -- the D = 96 dK pass went from
  //  200 to 288 and lost its second wave per SIMD, 3.5 ms of the 2-D training iteration)
  uint64_t drop_seed = 0, drop_own = 0;
  if constexpr (DROP) {
    drop_seed = p.drop.seed + (p.drop.seed_dev ? *p.drop.seed_dev : 0ull);
    drop_own = p.drop.offset + (uint64_t)bh * p.Lq * (uint64_t)p.Lk +
               (ROLE == ROLE_DQ ? (uint64_t)(ovalid ? oi : 0) * (uint64_t)p.Lk : (uint64_t)(ovalid ? oi : 0));
  }

  const int tiles_all = (n_str + C::BK - 1) / C::BK;
  const int tiles_per = (tiles_all + nsplit - 1) / nsplit;
  const int t_begin = split * tiles_per, tiles = min(tiles_all, t_begin + tiles_per);
  if (t_begin >= tiles) return;
  uint4 r1[PER], r2[PER];
  float rs_lse = 0.f, rs_del = 0.f;
  auto gload = [&](int tile) {
    const int row0 = tile * C::BK;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = tid + i * NT;
      uint4 a = make_uint4(0, 0, 0, 0), g = make_uint4(0, 0, 0, 0);
      if (c < CHUNKS) {
        const int row = row0 + c / CPR, dc = (c % CPR) * 8;
        if (row < n_str) {
          a = *reinterpret_cast<const uint4*>(str1 + (int64_t)row * str1_ts + dc);
          g = *reinterpret_cast<const uint4*>(str2 + (int64_t)row * str2_ts + dc);
        }
      }
      r1[i] = a;
      r2[i] = g;
    }
    if (ROLE != ROLE_DQ && tid < C::BK) {
      const int row = row0 + tid;
      rs_lse = row < n_str ? p.lse[bh * p.Lq + row] : INFINITY;           // 2^(s - inf) = 0: rows past the end contribute nothing
      rs_del = row < n_str ? p.delta[bh * p.Lq + row] : 0.f;
    }
  };
  auto lstore = [&](int buf) {
    unsigned char* base = smem + buf * C::STAGE;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = tid + i * NT;
      if (c < CHUNKS) {
        const int row = c / CPR, dc = (c % CPR) * 16;
        *reinterpret_cast<uint4*>(base + row * C::KS + dc) = r1[i];
        *reinterpret_cast<uint4*>(base + C::BK * C::KS + row * VS2 + dc) = r2[i];
      }
    }
    if (ROLE != ROLE_DQ && tid < C::BK) {
      stats[buf * 2 * C::BK + tid] = rs_lse;
      stats[buf * 2 * C::BK + C::BK + tid] = rs_del;
    }
  };

  f32x16 acc[DBLK];
#pragma unroll
  for (int d = 0; d < DBLK; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[d][e] = 0.f;

  // per-lane LDS byte offsets: row reads (A operand of the first products) and transposed reads (A operand of the second)
  const int row1_off = r * C::KS + h * 16;                               // + st*32 per k-step
  const int row2_off = r * VS2 + h * 16;
  const int li = lane & 15;
  const int tr_row = 4 * h + (li >> 2), tr_col = (16 * ((lane >> 4) & 1) + 4 * (li & 3)) * 2;
  constexpr int TRS = ROLE == ROLE_DV ? VS2 : C::KS;                     // stride of the slot the second product transposes
  const int tr_off = tr_row * TRS + tr_col;                              // + (16 st) rows + dblk*64 B; second half + 8 rows

  gload(t_begin);
  lstore(0);
  __syncthreads();
  int cur = 0;
  for (int tile = t_begin; tile < tiles; ++tile) {
    if (tile + 1 < tiles) gload(tile + 1);
    const unsigned char* base1 = smem + cur * C::STAGE;
    const unsigned char* base2 = base1 + C::BK * C::KS;
    f32x16 s, dp;
#pragma unroll
    for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
    for (int st = 0; st < DSTEPS; ++st) {
      const op16x8 a = *reinterpret_cast<const op16x8*>(base1 + row1_off + st * 32);
      s = MSAM2_MFMA_32x32x16(a, f1[st], s, 0, 0, 0);
    }
    if constexpr (ROLE != ROLE_DV) {
#pragma unroll
      for (int st = 0; st < DSTEPS; ++st) {
        const op16x8 a = *reinterpret_cast<const op16x8*>(base2 + row2_off + st * 32);
        dp = MSAM2_MFMA_32x32x16(a, f2[st], dp, 0, 0, 0);
      }
    }
    // streamed row of register e: (e&3) + 8*(e>>2) + 4*h
    op16x8 wf[2];
    const int row0 = tile * C::BK;
    if constexpr (ROLE == ROLE_DQ) {
      // (rows past the end of the keys hold zeros in LDS: whatever weight they get multiplies a zero K^T column)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float pe = __builtin_amdgcn_exp2f(s[e] * p.scale_log2 - lse_own);
        float dpe = dp[e];
        if constexpr (DROP)                              // owner = query oi, streamed row = key: the forward's mask on dP
          dpe = dropout_keep(drop_seed, drop_own + (uint64_t)(row0 + (e & 3) + 8 * (e >> 2) + 4 * h), p.drop.thr) ? dpe * p.drop.inv_keep : 0.f;
        wf[e >> 3][e & 7] = f2op(pe * (dpe - delta_own));
      }
    } else {
      const float* st_lse = stats + cur * 2 * C::BK;
      const float* st_del = st_lse + C::BK;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
        const float pe = __builtin_amdgcn_exp2f(s[e] * p.scale_log2 - st_lse[row]);
        float m = 1.f;
        if constexpr (DROP)                              // owner = key oi, streamed row = query
          m = dropout_keep(drop_seed, drop_own + (uint64_t)(row0 + row) * (uint64_t)p.Lk, p.drop.thr) ? p.drop.inv_keep : 0.f;
        wf[e >> 3][e & 7] = f2op(ROLE == ROLE_DV ? pe * m : pe * (dp[e] * m - st_del[row]));
      }
    }
    // out^T[d][owner] += tile^T[d][row] w[row][owner]
    const unsigned char* trb = (ROLE == ROLE_DV ? base2 : base1) + tr_off;
#pragma unroll
    for (int d = 0; d < DBLK; ++d) {
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const unsigned char* a0 = trb + (16 * st) * TRS + d * 64;
        const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(a0));
        const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(a0 + 8 * TRS));
        typedef __attribute__((ext_vector_type(8))) short short8_t;
        short8_t t8;
        t8[0] = lo[0]; t8[1] = lo[1]; t8[2] = lo[2]; t8[3] = lo[3];
        t8[4] = hi[0]; t8[5] = hi[1]; t8[6] = hi[2]; t8[7] = hi[3];
        acc[d] = MSAM2_MFMA_32x32x16(__builtin_bit_cast(op16x8, t8), wf[st], acc[d], 0, 0, 0);
      }
    }
    if (tile + 1 < tiles) lstore(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue: fp32 rows of the owner (lane = owner row, registers = 4 consecutive channels per group)
  const float factor = ROLE == ROLE_DV ? 1.f : p.scale;
  if (!ovalid) return;
  float* out = ROLE == ROLE_DQ ? p.dq + b * p.dq_bs + head * p.dq_hs + (int64_t)oi * p.dq_ts
             : ROLE == ROLE_DK ? p.dk + b * p.dk_bs + head * p.dk_hs + (int64_t)oi * p.dk_ts
                               : p.dv + b * p.dv_bs + head * p.dv_hs + (int64_t)oi * p.dv_ts;
#pragma unroll
  for (int d = 0; d < DBLK; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 w;
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = acc[d][4 * g + e] * factor;
      if (ROLE == ROLE_DQ && nsplit > 1) {
        float* part = p.dq_part + (((int64_t)split * p.B * p.H + bh) * p.Lq + oi) * D;
        *reinterpret_cast<f32x4*>(part + d * 32 + 8 * g + 4 * h) = w;
      } else {
        *reinterpret_cast<f32x4*>(out + d * 32 + 8 * g + 4 * h) = w;
      }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// LDS-DMA variant (D = 128 / 256): the streamed tiles go global -> LDS with buffer_load ... lds (no staging registers, no LDS store
// instructions), in the forward kernel's swizzled images so that every fragment read is bank-conflict free:
//   row image  (ds_read_b128 rows):        16-byte chunk c of row r sits at (c & ~15) | ((c & 15) ^ (r & 15))
//   tr image   (ds_read_b64_tr_b16):       chunk c of row r sits at c ^ ((r & 3) << 2)
// Slot 1 is read both ways by the DQ / DK roles, so it is streamed twice (one image of each kind, 16 KB each at D = 256); slot 2 is
// a row image for DQ / DK and a tr image for DV.  One barrier per tile: tile t+1 is issued right after the barrier that says
// "tile t has landed and everybody is done with tile t-1".
// ------------------------------------------------------------------------------------------------------------------
template <int D, int NW, int ROLE, bool DROP>
__global__ __launch_bounds__(NW * 64, 1) void attn_bwd_dma_kernel(AttnBwdParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BK = 32, RB = D * 2, CPR = D / 8;
  constexpr int IMG = BK * RB;
  constexpr int NIMG = ROLE == ROLE_DV ? 2 : 3;          // [slot-1 row image][slot-1 tr image (DQ / DK)][slot-2 image]
  constexpr int STAGE = NIMG * IMG;
  constexpr int KI = IMG / 1024, PW = KI / NW;
  constexpr int DSTEPS = D / 16, DBLK = D / 32;
  static_assert(KI % NW == 0, "image must split evenly over the waves");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  float* stats = reinterpret_cast<float*>(smem + 2 * STAGE);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int gx = gridDim.x, gy = gridDim.y;
  int lid = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
  {
    const int nwg = gx * gy * gridDim.z;
    const int q8 = nwg / 8, rem = nwg % 8, xcd = lid % 8;
    lid = (xcd < rem ? xcd * (q8 + 1) : rem * (q8 + 1) + (xcd - rem) * q8) + lid / 8;
  }
  const int bx = lid % gx, head = (lid / gx) % gy, b = lid / (gx * gy);
  const int64_t bh = (int64_t)b * p.H + head;

  const int n_own = ROLE == ROLE_DQ ? p.Lq : p.Lk;
  const int n_str = ROLE == ROLE_DQ ? p.Lk : p.Lq;
  const op16* qb = p.q + b * p.q_bs + head * p.q_hs;
  const op16* kb = p.k + b * p.k_bs + head * p.k_hs;
  const op16* vb = p.v + b * p.v_bs + head * p.v_hs;
  const op16* gb = p.do16 + bh * p.Lq * D;
  const op16* own1 = ROLE == ROLE_DQ ? qb : kb;
  const int64_t own1_ts = ROLE == ROLE_DQ ? p.q_ts : p.k_ts;
  const op16* own2 = ROLE == ROLE_DQ ? gb : vb;
  const int64_t own2_ts = ROLE == ROLE_DQ ? (int64_t)D : p.v_ts;
  const op16* str1 = ROLE == ROLE_DQ ? kb : qb;
  const int str1_rb = (int)(ROLE == ROLE_DQ ? p.k_ts : p.q_ts) * 2;      // streamed row pitch in bytes
  const op16* str2 = ROLE == ROLE_DQ ? vb : gb;
  const int str2_rb = (int)(ROLE == ROLE_DQ ? p.v_ts : (int64_t)D) * 2;

  const int nsplit = ROLE == ROLE_DQ ? p.ksplit : 1;
  const int nob = gx / nsplit;
  const int split = bx / nob, oblk = bx - split * nob;
  const int oi = oblk * (NW * 32) + wave * 32 + r;
  const bool ovalid = oi < n_own;
  op16x8 f1[DSTEPS], f2[ROLE == ROLE_DV ? 1 : DSTEPS];
#pragma unroll
  for (int s = 0; s < DSTEPS; ++s) {
    uint4 a = make_uint4(0, 0, 0, 0), c = make_uint4(0, 0, 0, 0);
    if (ovalid) {
      a = *reinterpret_cast<const uint4*>(own1 + (int64_t)oi * own1_ts + s * 16 + h * 8);
      if constexpr (ROLE != ROLE_DV) c = *reinterpret_cast<const uint4*>(own2 + (int64_t)oi * own2_ts + s * 16 + h * 8);
    }
    f1[s] = __builtin_bit_cast(op16x8, a);
    if constexpr (ROLE != ROLE_DV) f2[s] = __builtin_bit_cast(op16x8, c);
  }
  const float delta_own = (ROLE == ROLE_DQ && ovalid) ? p.delta[bh * p.Lq + oi] : 0.f;
  const float lse_own = (ROLE == ROLE_DQ && ovalid) ? p.lse[bh * p.Lq + oi] : 0.f;
  // dropout stream: DQ owns a query row (element index of key 0 of that row), DK / DV own a key (element index of query 0, that key)
  // (DROP instances only: as a run-time branch the mask generator cost every instance 12-90 registers -- the D = 96 dK pass went from
  //  200 to 288 and lost its second wave per SIMD, 3.5 ms of the 2-D training iteration)
  uint64_t drop_seed = 0, drop_own = 0;
  if constexpr (DROP) {
    drop_seed = p.drop.seed + (p.drop.seed_dev ? *p.drop.seed_dev : 0ull);
    drop_own = p.drop.offset + (uint64_t)bh * p.Lq * (uint64_t)p.Lk +
               (ROLE == ROLE_DQ ? (uint64_t)(ovalid ? oi : 0) * (uint64_t)p.Lk : (uint64_t)(ovalid ? oi : 0));
  }

  const int tiles_all = (n_str + BK - 1) / BK;
  const int tiles_per = (tiles_all + nsplit - 1) / nsplit;
  const int t_begin = split * tiles_per, t_end = min(tiles_all, t_begin + tiles_per);
  if (t_begin >= t_end) return;
  const int t_full_end = min(t_end, n_str / BK);

  const auto rsrc1 = __builtin_amdgcn_make_buffer_rsrc((void*)str1, 0, 0x7fffffff, 0x00020000);
  const auto rsrc2 = __builtin_amdgcn_make_buffer_rsrc((void*)str2, 0, 0x7fffffff, 0x00020000);
  // per-lane source offsets: LDS slot f = (wave*PW + j)*64 + lane of an image holds global chunk swz(c) of row f / CPR
  auto src_off = [&](int j, int row_bytes, bool tr_image, int clamp_last) -> unsigned {
    const int f = (wave * PW + j) * 64 + lane;
    const int row = f / CPR, c = f % CPR;
    const int gc = tr_image ? (c ^ ((row & 3) << 2)) : ((c & ~15) | ((c & 15) ^ (row & 15)));
    return (unsigned)(min(row, clamp_last) * row_bytes + (gc << 4));
  };
  unsigned o1r[PW], o1t[ROLE == ROLE_DV ? 1 : PW], o2[PW];
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    o1r[j] = src_off(j, str1_rb, false, BK);
    if constexpr (ROLE != ROLE_DV) o1t[j] = src_off(j, str1_rb, true, BK);
    o2[j] = src_off(j, str2_rb, ROLE == ROLE_DV, BK);
  }
  auto issue = [&](int tile, int stage) {
    unsigned char* base = smem + stage * STAGE + wave * PW * 1024;
    const unsigned s1 = (unsigned)(tile * BK) * (unsigned)str1_rb, s2 = (unsigned)(tile * BK) * (unsigned)str2_rb;
#pragma unroll
    for (int j = 0; j < PW; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc1, (__attribute__((address_space(3))) void*)(base + j * 1024), 16, o1r[j], s1, 0, 0);
    if constexpr (ROLE != ROLE_DV) {
#pragma unroll
      for (int j = 0; j < PW; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc1, (__attribute__((address_space(3))) void*)(base + IMG + j * 1024), 16, o1t[j], s1, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < PW; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc2, (__attribute__((address_space(3))) void*)(base + (NIMG - 1) * IMG + j * 1024), 16, o2[j], s2, 0, 0);
  };
  // statistics of the streamed rows (DK / DV): loaded by 32 lanes, parked in LDS next to the stage they belong to
  float rs_lse = 0.f, rs_del = 0.f;
  auto stat_load = [&](int tile) {
    if (ROLE != ROLE_DQ && tid < BK) {
      const int row = tile * BK + tid;
      rs_lse = row < n_str ? p.lse[bh * p.Lq + row] : INFINITY;
      rs_del = row < n_str ? p.delta[bh * p.Lq + row] : 0.f;
    }
  };
  auto stat_store = [&](int buf) {
    if (ROLE != ROLE_DQ && tid < BK) {
      stats[buf * 2 * BK + tid] = rs_lse;
      stats[buf * 2 * BK + BK + tid] = rs_del;
    }
  };

  f32x16 acc[DBLK];
#pragma unroll
  for (int d = 0; d < DBLK; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[d][e] = 0.f;

  // fragment read offsets (see attention.hip: K image rows / V image transposed reads)
  const int k_row = r * RB, k_x = r & 15;
  const int li = lane & 15, vq = li >> 2, vp = li & 3, cgrp = (lane >> 4) & 1;
  const int v_row = (4 * h + vq) * RB + ((vp & 1) << 3);
  const int v_sw = vq << 2, v_c0 = 2 * cgrp + (vp >> 1);
  auto row_frag = [&](const unsigned char* img, int st) -> op16x8 {
    const int c = 2 * st + h;
    return *reinterpret_cast<const op16x8*>(img + k_row + (((c & ~15) | ((c & 15) ^ k_x)) << 4));
  };

  auto compute = [&](int stage, int row0, const bool masked) __attribute__((always_inline)) {
    const unsigned char* img1r = smem + stage * STAGE;
    const unsigned char* img2 = img1r + (NIMG - 1) * IMG;
    const unsigned char* imgt = ROLE == ROLE_DV ? img2 : img1r + IMG;
    f32x16 s, dp;
#pragma unroll
    for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
    for (int g2 = 0; g2 < DSTEPS; g2 += 2) {
      op16x8 a[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) a[u] = row_frag(img1r, g2 + u);
#pragma unroll
      for (int u = 0; u < 2; ++u) s = MSAM2_MFMA_32x32x16(a[u], f1[g2 + u], s, 0, 0, 0);
    }
    if constexpr (ROLE != ROLE_DV) {
#pragma unroll
      for (int g2 = 0; g2 < DSTEPS; g2 += 2) {
        op16x8 a[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) a[u] = row_frag(img2, g2 + u);
#pragma unroll
        for (int u = 0; u < 2; ++u) dp = MSAM2_MFMA_32x32x16(a[u], f2[g2 + u], dp, 0, 0, 0);
        }
    }
    op16x8 wf[2];
    if constexpr (ROLE == ROLE_DQ) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float pe = __builtin_amdgcn_exp2f(s[e] * p.scale_log2 - lse_own);
        if (masked && row0 + (e & 3) + 8 * (e >> 2) + 4 * h >= n_str) pe = 0.f;   // tail tile: clamped duplicates of the last key
        float dpe = dp[e];
        if constexpr (DROP)                              // owner = query oi, streamed row = key: the forward's mask on dP
          dpe = dropout_keep(drop_seed, drop_own + (uint64_t)(row0 + (e & 3) + 8 * (e >> 2) + 4 * h), p.drop.thr) ? dpe * p.drop.inv_keep : 0.f;
        wf[e >> 3][e & 7] = f2op(pe * (dpe - delta_own));
      }
    } else {
      const float* st_lse = stats + (stage & 1) * 2 * BK;
      const float* st_del = st_lse + BK;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
        const float pe = __builtin_amdgcn_exp2f(s[e] * p.scale_log2 - st_lse[row]);   // lse = +inf past the end: weight 0
        float m = 1.f;
        if constexpr (DROP)                              // owner = key oi, streamed row = query
          m = dropout_keep(drop_seed, drop_own + (uint64_t)(row0 + row) * (uint64_t)p.Lk, p.drop.thr) ? p.drop.inv_keep : 0.f;
        wf[e >> 3][e & 7] = f2op(ROLE == ROLE_DV ? pe * m : pe * (dp[e] * m - st_del[row]));
      }
    }
#pragma unroll
    for (int d = 0; d < DBLK; ++d) {
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        typedef __attribute__((ext_vector_type(8))) short short8_t;
        const int cch = (d * 4 + v_c0) ^ v_sw;
        const unsigned char* a0 = imgt + v_row + (16 * st) * RB + (cch << 4);
        const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(a0));
        const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(a0 + 8 * RB));
        short8_t t8;
        t8[0] = lo[0]; t8[1] = lo[1]; t8[2] = lo[2]; t8[3] = lo[3];
        t8[4] = hi[0]; t8[5] = hi[1]; t8[6] = hi[2]; t8[7] = hi[3];
        acc[d] = MSAM2_MFMA_32x32x16(__builtin_bit_cast(op16x8, t8), wf[st], acc[d], 0, 0, 0);
      }
    }
  };

  if (t_begin < t_full_end) {
    issue(t_begin, 0);
    stat_load(t_begin);
    stat_store(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the bare s_barrier below does not wait for this LDS write
  }
  for (int tile = t_begin; tile < t_full_end; ++tile) {
    const int st_i = (tile - t_begin) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (tile + 1 < t_full_end) {
      issue(tile + 1, st_i ^ 1);
      stat_load(tile + 1);
    }
    compute(st_i, tile * BK, false);
    if (tile + 1 < t_full_end) stat_store(st_i ^ 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (t_full_end < t_end) {
    // partial last tile: rows past the end re-read the last valid row (their weights are zero: masked / lse = +inf)
    const int row0 = t_full_end * BK, last = n_str - 1 - row0;
    __builtin_amdgcn_s_barrier();                             // every wave is done with the last full tile's stage
    unsigned char* base = smem + wave * PW * 1024;
    const unsigned s1 = (unsigned)row0 * (unsigned)str1_rb, s2 = (unsigned)row0 * (unsigned)str2_rb;
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc1, (__attribute__((address_space(3))) void*)(base + j * 1024), 16,
                                               src_off(j, str1_rb, false, last), s1, 0, 0);
      if constexpr (ROLE != ROLE_DV)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc1, (__attribute__((address_space(3))) void*)(base + IMG + j * 1024), 16,
                                                 src_off(j, str1_rb, true, last), s1, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc2, (__attribute__((address_space(3))) void*)(base + (NIMG - 1) * IMG + j * 1024), 16,
                                               src_off(j, str2_rb, ROLE == ROLE_DV, last), s2, 0, 0);
    }
    stat_load(t_full_end);
    stat_store(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    compute(0, row0, true);
  }

  const float factor = ROLE == ROLE_DV ? 1.f : p.scale;
  if (!ovalid) return;
  float* out = ROLE == ROLE_DQ ? p.dq + b * p.dq_bs + head * p.dq_hs + (int64_t)oi * p.dq_ts
             : ROLE == ROLE_DK ? p.dk + b * p.dk_bs + head * p.dk_hs + (int64_t)oi * p.dk_ts
                               : p.dv + b * p.dv_bs + head * p.dv_hs + (int64_t)oi * p.dv_ts;
#pragma unroll
  for (int d = 0; d < DBLK; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 w;
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = acc[d][4 * g + e] * factor;
      if (ROLE == ROLE_DQ && nsplit > 1) {
        float* part = p.dq_part + (((int64_t)split * p.B * p.H + bh) * p.Lq + oi) * D;
        *reinterpret_cast<f32x4*>(part + d * 32 + 8 * g + 4 * h) = w;
      } else {
        *reinterpret_cast<f32x4*>(out + d * 32 + 8 * g + 4 * h) = w;
      }
    }
#endif
}

// dq[row] = sum over the key splits of the DQ pass's partial rows (16M fp32 atomics cost ~0.5 ms at 4096 x 256 x 4 x 4 splits; this
// pass moves 80 MB instead)
__global__ __launch_bounds__(256) void attn_bwd_reduce_kernel(const float* __restrict__ part, float* __restrict__ x, int64_t bs, int64_t hs,
                                                              int64_t ts, int H, int L, int D4, int nsplit, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = i % D4;
    const int64_t row = i / D4;
    f32x4 acc = *reinterpret_cast<const f32x4*>(part + i * 4);
    for (int s = 1; s < nsplit; ++s) acc += *reinterpret_cast<const f32x4*>(part + ((int64_t)s * total + i) * 4);
    const int64_t t = row % L, bh = row / L;
    *reinterpret_cast<f32x4*>(x + (bh / H) * bs + (bh % H) * hs + t * ts + c * 4) = acc;
  }
}

template <int D, int NW, int ROLE, bool DROP>
void launch_role_t(const AttnBwdParams& p, hipStream_t s) {
  using C = BwdCfg<D>;
  const int n_own = ROLE == ROLE_DQ ? p.Lq : p.Lk;
  dim3 grid(cdiv(n_own, NW * 32) * (ROLE == ROLE_DQ ? p.ksplit : 1), p.H, p.B);
  if constexpr (D == 128 || D == 256) {
    // LDS-DMA variant: 32-bit source offsets (rows * pitch < 2^31 bytes), 16-byte aligned rows (checked by the entry point)
    static const bool no_dma = getenv("MSAM2_BWD_NO_DMA") != nullptr;
    const int64_t max_bytes = (int64_t)max(p.Lq, p.Lk) * max(max(p.q_ts, p.k_ts), max(p.v_ts, (int64_t)D)) * 2;
    if (!no_dma && max_bytes < (1ll << 31)) {
      constexpr int LDSB = 2 * (ROLE == ROLE_DV ? 2 : 3) * 32 * D * 2 + BwdCfg<D>::STATS;
      static bool attr_dma = false;
      if (!attr_dma) {
        hipFuncSetAttribute((const void*)attn_bwd_dma_kernel<D, NW, ROLE, DROP>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
        attr_dma = true;
      }
      hipLaunchKernelGGL((attn_bwd_dma_kernel<D, NW, ROLE, DROP>), grid, dim3(NW * 64), LDSB, s, p);
      return;
    }
  }
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)attn_bwd_kernel<D, NW, ROLE, DROP>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    attr_set = true;
  }
  hipLaunchKernelGGL((attn_bwd_kernel<D, NW, ROLE, DROP>), grid, dim3(NW * 64), C::LDS_BYTES, s, p);
}

template <int D, int NW, int ROLE>
void launch_role(const AttnBwdParams& p, hipStream_t s) {
  if (p.drop.thr) launch_role_t<D, NW, ROLE, true>(p, s);
  else launch_role_t<D, NW, ROLE, false>(p, s);
}

// DQ role: key splits per owner block -- until there are ~2 workgroups per CU, each split keeping at least 8 key tiles, at most
// DQ_MAX_SPLIT (the workspace holds that many partial dQ)
constexpr int DQ_MAX_SPLIT = 8;
int dq_key_splits(int64_t B, int64_t H, int64_t Lq, int64_t Lk) {
  const int64_t blocks4 = cdiv(Lq, (int64_t)128) * H * B;
  if (blocks4 >= 512) return 1;
  return (int)max((int64_t)1, min(min((int64_t)DQ_MAX_SPLIT, (int64_t)cdiv((int64_t)512, blocks4)), cdiv(Lk, (int64_t)32) / 8));
}

// workgroup size by fill: 4 waves per workgroup unless that leaves CUs idle
template <int D, int ROLE>
void launch_fill(AttnBwdParams p, hipStream_t s) {
  const int64_t n_own = ROLE == ROLE_DQ ? p.Lq : p.Lk;
  const int64_t blocks4 = cdiv(n_own, (int64_t)128) * p.H * p.B;
  p.ksplit = ROLE == ROLE_DQ ? dq_key_splits(p.B, p.H, p.Lq, p.Lk) : 1;
  if (blocks4 * p.ksplit >= 256 || n_own <= 32) launch_role<D, 4, ROLE>(p, s);
  else launch_role<D, 2, ROLE>(p, s);
  if (p.ksplit > 1) {
    const int64_t total = (int64_t)p.B * p.H * p.Lq * (D / 4);
    hipLaunchKernelGGL(attn_bwd_reduce_kernel, dim3((unsigned)min((int64_t)4096, cdiv(total, (int64_t)256))), dim3(256), 0, s, p.dq_part, p.dq, p.dq_bs,
                       p.dq_hs, p.dq_ts, p.H, p.Lq, D / 4, p.ksplit, total);
  }
}

template <int D>
int launch_all(const AttnBwdParams& p, const float* d_o, const int64_t* gs, const op16* o, const int64_t* os, op16* do16, float* delta,
               hipStream_t s) {
  const int64_t rows = (int64_t)p.B * p.H * p.Lq;
  hipLaunchKernelGGL((attn_bwd_prep_kernel<D>), dim3((unsigned)cdiv(rows, (int64_t)4)), dim3(256), 0, s, d_o, gs[0], gs[1], gs[2], o, os[0], os[1],
                     os[2], do16, delta, p.H, p.Lq, rows);
  launch_fill<D, ROLE_DQ>(p, s);
  launch_fill<D, ROLE_DK>(p, s);
  launch_fill<D, ROLE_DV>(p, s);
  return msam2_check_launch("attention_bwd");
}

}  // namespace

extern "C" size_t msam2_attention_bwd_workspace_bytes(int64_t B, int64_t H, int64_t Lq, int64_t D) {
  const size_t rows = (size_t)(B * H * Lq);
  const size_t base = rows * (size_t)(D * sizeof(op16) + sizeof(float));
  // partial dQ of the key-split DQ pass (only taken when the queries alone do not fill the chip)
  const size_t parts = cdiv(Lq, (int64_t)128) * H * B < 512 ? (size_t)DQ_MAX_SPLIT * rows * D * sizeof(float) : 0;
  return ((base + 255) & ~(size_t)255) + parts;
}

// q / k / v / o: 16-bit, element strides {batch, head, token}, channels contiguous; d_o fp32 with its own strides; dq / dk / dv fp32
// outputs with their own strides (token stride a multiple of 4 elements, 16-byte aligned rows).  o is the forward's output for the
// same q, k, v and lse [B, H, Lq] its log-sum-exp rows (msam2_attention_fwd_lse).  workspace: msam2_attention_bwd_workspace_bytes.
static int attention_bwd_impl(const void* q, const int64_t* q_strides, const void* k, const int64_t* k_strides, const void* v,
                              const int64_t* v_strides, const void* o, const int64_t* o_strides, const float* lse,
                              const float* d_o, const int64_t* do_strides, float* dq, const int64_t* dq_strides, float* dk, const int64_t* dk_strides,
                              float* dv, const int64_t* dv_strides, void* workspace, size_t workspace_bytes, int64_t B, int64_t H,
                              int64_t Lq, int64_t Lk, int64_t D, float scale, void* stream, const AttnDropout* drop) {
  MSAM2_REQUIRE(q && k && v && o && lse && d_o && dq && dk && dv && workspace, "attention_bwd: null pointer");
  MSAM2_REQUIRE(q_strides && k_strides && v_strides && o_strides && do_strides && dq_strides && dk_strides && dv_strides,
                "attention_bwd: null strides");
  MSAM2_REQUIRE(B > 0 && H > 0 && Lq > 0 && Lk > 0, "attention_bwd: empty problem");
  MSAM2_REQUIRE(B <= 65535 && H <= 65535 && Lq < (1ll << 31) && Lk < (1ll << 31), "attention_bwd: problem too large");
  MSAM2_REQUIRE(D == 64 || D == 96 || D == 128 || D == 256, "attention_bwd: head dim %lld not in {64, 96, 128, 256}", (long long)D);
  MSAM2_REQUIRE(workspace_bytes >= msam2_attention_bwd_workspace_bytes(B, H, Lq, D), "attention_bwd: workspace too small");
  const int64_t* in_strides[4] = {q_strides, k_strides, v_strides, o_strides};
  const void* in_ptrs[4] = {q, k, v, o};
  for (int i = 0; i < 4; ++i)
    MSAM2_REQUIRE(in_strides[i][2] % 8 == 0 && in_strides[i][1] % 8 == 0 && in_strides[i][0] % 8 == 0 && ((uintptr_t)in_ptrs[i] & 15) == 0,
                  "attention_bwd: 16-bit operand %d needs 16-byte aligned rows", i);
  const int64_t* f_strides[4] = {do_strides, dq_strides, dk_strides, dv_strides};
  const void* f_ptrs[4] = {d_o, dq, dk, dv};
  for (int i = 0; i < 4; ++i)
    MSAM2_REQUIRE(f_strides[i][2] % 4 == 0 && f_strides[i][1] % 4 == 0 && f_strides[i][0] % 4 == 0 && ((uintptr_t)f_ptrs[i] & 15) == 0,
                  "attention_bwd: fp32 tensor %d needs 16-byte aligned rows", i);
  AttnBwdParams p;
  p.q = (const op16*)q; p.k = (const op16*)k; p.v = (const op16*)v;
  p.q_bs = q_strides[0]; p.q_hs = q_strides[1]; p.q_ts = q_strides[2];
  p.k_bs = k_strides[0]; p.k_hs = k_strides[1]; p.k_ts = k_strides[2];
  p.v_bs = v_strides[0]; p.v_hs = v_strides[1]; p.v_ts = v_strides[2];
  op16* do16 = (op16*)workspace;
  float* delta = (float*)((char*)workspace + (size_t)(B * H * Lq) * D * sizeof(op16));
  p.do16 = do16; p.delta = delta; p.lse = lse; p.ksplit = 1;
  p.dq_part = reinterpret_cast<float*>((char*)workspace + (((size_t)(B * H * Lq) * (size_t)(D * sizeof(op16) + sizeof(float)) + 255) & ~(size_t)255));
  p.dq = dq; p.dk = dk; p.dv = dv;
  p.dq_bs = dq_strides[0]; p.dq_hs = dq_strides[1]; p.dq_ts = dq_strides[2];
  p.dk_bs = dk_strides[0]; p.dk_hs = dk_strides[1]; p.dk_ts = dk_strides[2];
  p.dv_bs = dv_strides[0]; p.dv_hs = dv_strides[1]; p.dv_ts = dv_strides[2];
  p.B = (int)B; p.H = (int)H; p.Lq = (int)Lq; p.Lk = (int)Lk;
  p.scale = scale; p.scale_log2 = scale * 1.4426950408889634f;
  p.drop = drop ? *drop : AttnDropout{0u, 1.f, 0ull, 0ull, nullptr};
  hipStream_t s = (hipStream_t)stream;
  switch (D) {
    case 64: return launch_all<64>(p, d_o, do_strides, (const op16*)o, o_strides, do16, delta, s);
    case 96: return launch_all<96>(p, d_o, do_strides, (const op16*)o, o_strides, do16, delta, s);
    case 128: return launch_all<128>(p, d_o, do_strides, (const op16*)o, o_strides, do16, delta, s);
    default: return launch_all<256>(p, d_o, do_strides, (const op16*)o, o_strides, do16, delta, s);
  }
}

extern "C" int msam2_attention_bwd(const void* q, const int64_t* q_strides, const void* k, const int64_t* k_strides, const void* v,
                                   const int64_t* v_strides, const void* o, const int64_t* o_strides, const float* lse,
                                   const float* d_o, const int64_t* do_strides, float* dq, const int64_t* dq_strides, float* dk, const int64_t* dk_strides,
                                   float* dv, const int64_t* dv_strides, void* workspace, size_t workspace_bytes, int64_t B, int64_t H,
                                   int64_t Lq, int64_t Lk, int64_t D, float scale, void* stream) {
  return attention_bwd_impl(q, q_strides, k, k_strides, v, v_strides, o, o_strides, lse, d_o, do_strides, dq, dq_strides, dk, dk_strides, dv,
                            dv_strides, workspace, workspace_bytes, B, H, Lq, Lk, D, scale, stream, nullptr);
}

// msam2_attention_bwd for a forward that ran with dropout on the attention probabilities (msam2_attention_fwd_lse_dropout with the same
// p / seed / offset / seed_dev): every pass re-creates the mask from the counter stream -- dV takes the kept probabilities / (1 - p),
// dQ / dK take dS = P o (mask o dP / (1 - p) - delta) with delta = rowsum(dO o O) of the dropped forward's output.  o and lse are the
// forward's outputs (lse: of the un-dropped probabilities).
extern "C" int msam2_attention_bwd_dropout(const void* q, const int64_t* q_strides, const void* k, const int64_t* k_strides, const void* v,
                                           const int64_t* v_strides, const void* o, const int64_t* o_strides, const float* lse,
                                           const float* d_o, const int64_t* do_strides, float* dq, const int64_t* dq_strides, float* dk,
                                           const int64_t* dk_strides, float* dv, const int64_t* dv_strides, void* workspace, size_t workspace_bytes,
                                           int64_t B, int64_t H, int64_t Lq, int64_t Lk, int64_t D, float scale, float p, uint64_t seed,
                                           uint64_t offset, const void* seed_dev, void* stream) {
  MSAM2_REQUIRE(p >= 0.f && p < 1.f, "attention_bwd_dropout: p must be in [0, 1)");
  AttnDropout d = {(unsigned)fmin(4294967295.0, (double)p * 4294967296.0), 1.f / (1.f - p), seed, offset, (const uint64_t*)seed_dev};
  return attention_bwd_impl(q, q_strides, k, k_strides, v, v_strides, o, o_strides, lse, d_o, do_strides, dq, dq_strides, dk, dk_strides, dv,
                            dv_strides, workspace, workspace_bytes, B, H, Lq, Lk, D, scale, stream, &d);
}
