// Fused pre-LN MLP block of the Hiera trunk for gfx950:   out = x + fc2( GELU( fc1( LayerNorm(x) ) ) )      (hieradet.py:166-167,
// sam2_utils.py:108-132 with nn.GELU, LayerNorm eps 1e-6)
//
// Why one kernel.  At the two high-resolution stages (dim 96 / 192, 262144 / 65536 tokens at 4 x 1024^2) the three launches of the block --
// LayerNorm, fc1 + GELU, fc2 + residual -- are bound by what they move, not by what they compute: the 4 x dim hidden activation
// (201 MB at stage 1) is written and read back once, the LayerNorm output once more, and K = 96 / 192 GEMMs spend their life in prologue
// and epilogue.  Here the hidden activation never leaves the registers:
//
//   * everything is computed TRANSPOSED so that a lane owns one token: H^T = W1 X^T (A = W1 rows from LDS, B = the token's normalised
//     row held in registers), then Y^T += W2 H^T where the fp32 accumulator of H^T, after bias + GELU + conversion, IS the B operand
//     (the accumulator of a 32x32x16 MFMA has the token on the lane and 8 consecutive k per register group: no LDS round trip, no
//     shuffle); the k order of that operand is permuted, so W2 is given with its hidden dimension pre-permuted the same way (host);
//   * a wave carries TB blocks of 32 tokens through the whole hidden dimension, so every weight fragment read from LDS feeds TB (fc1)
//     MFMAs: LDS traffic per MFMA is well under the 1 KB of the 128 x 128 GEMM tiles;
//   * the weights stream through LDS in hidden-dimension chunks of HC (both W1 rows and W2 columns of a chunk: 72 KB), double
//     buffered by LDS-DMA (buffer_load ... lds) one chunk ahead, ONE barrier per chunk; workgroups are persistent over token passes
//     and the chunk ring runs on across passes;
//   * LayerNorm runs on the lane's half row (the other half sits 32 lanes away: one cross-half add), the residual is added in the
//     store.
// LDS images: rows of 192 B use chunk' = (c & ~3) | ((c & 3) ^ ((r >> 2) & 3)), rows of 384 B chunk' = (c & ~7) | ((c & 7) ^ ((r >> 1) & 7))
// (both conflict free for the 32-row ds_read_b128 fragment reads); applied on the DMA source address and on the reads.
#include "common.h"

struct MlpFusedParams {
  const float* x;        // [T, DIM] fp32 residual stream
  float* out;            // [T, DIM]
  const float *ln_w, *ln_b, *b1, *b2;
  const op16* w1;        // [4*DIM, DIM]
  const op16* w2p;       // [DIM, 4*DIM], hidden index permuted inside every 32-block (see mlp_fused_permute)
  int64_t T;
  float eps;
  op16* out16;           // optional second output: the same rows in the 16-bit operand type (a stage's last block: the FPN's lateral GEMM operand)
};

template <int RB>
__device__ __forceinline__ int mlp_swz(int c, int r) {
  if constexpr (RB == 192) return (c & ~3) | ((c & 3) ^ ((r >> 2) & 3));
  else return (c & ~7) | ((c & 7) ^ ((r >> 1) & 7));
}

// NWV waves per workgroup (4 or 8).  With 8 (round 4) the workgroup still owns ONE weight ring in LDS, a wave carries TB = half as many
// token blocks, and every SIMD holds two waves: one wave's GELU / LayerNorm vector work and LDS reads run under the other's MFMAs, and a
// wave's registers (<= 256) no longer spill (the 4-wave form at dim 192 spilled 21-35 registers of a 512-register budget).
template <int DIM, int HC, int TB, int OCC = 1, int NWV = 4>
__global__ __launch_bounds__(NWV * 64, OCC) void mlp_fused_kernel(MlpFusedParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int HID = 4 * DIM, NCH = HID / HC, KS1 = DIM / 16, DB = DIM / 32, HB = HC / 32;
  constexpr int RB1 = DIM * 2, RB2 = HC * 2;                 // LDS row bytes of the W1 chunk [HC][DIM] and the W2 chunk [DIM][HC]
  constexpr int CPR1 = DIM / 8, CPR2 = HC / 8;               // 16-byte chunks per row
  constexpr int W1B = HC * RB1, W2B = DIM * RB2, BUF = W1B + W2B;
  constexpr int PIECES1 = W1B / 1024, PIECES2 = W2B / 1024, PW1 = (PIECES1 + NWV - 1) / NWV, PW2 = (PIECES2 + NWV - 1) / NWV;
  static_assert((RB1 == 192 || RB1 == 384) && (RB2 == 192 || RB2 == 384), "row swizzles are built for 192 / 384-byte rows");
  static_assert(2 * BUF + (3 * DIM + HID) * 4 <= 160 * 1024, "LDS budget");   // (pieces are dealt j * NWV + wave: a last, partial round is guarded)
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  float* prm = reinterpret_cast<float*>(smem + 2 * BUF);     // [gamma DIM | beta DIM | b2 DIM | b1 HID]

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int r = lane & 31, h = lane >> 5;
  for (int i = tid; i < DIM; i += NWV * 64) {
    prm[i] = p.ln_w[i];
    prm[DIM + i] = p.ln_b[i];
    prm[2 * DIM + i] = p.b2[i];
  }
  for (int i = tid; i < HID; i += NWV * 64) prm[3 * DIM + i] = p.b1[i];

  // ---- weight stream: per-lane source offsets of this wave's DMA pieces (chunk-independent part), scalar chunk offset added per issue
  const auto w1_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1, 0, HID * DIM * 2, 0x00020000);
  const auto w2_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2p, 0, HID * DIM * 2, 0x00020000);
  unsigned off1[PW1], off2[PW2];
#pragma unroll
  for (int j = 0; j < PW1; ++j) {
    const int f = min(j * NWV + wave, PIECES1 - 1) * 64 + lane, row = f / CPR1, c = f % CPR1;
    off1[j] = (unsigned)(row * RB1 + mlp_swz<RB1>(c, row) * 16);                       // W1 rows are DIM * 2 bytes in memory too
  }
#pragma unroll
  for (int j = 0; j < PW2; ++j) {
    const int f = min(j * NWV + wave, PIECES2 - 1) * 64 + lane, row = f / CPR2, c = f % CPR2;
    off2[j] = (unsigned)(row * HID * 2 + mlp_swz<RB2>(c, row) * 16);
  }
  auto issue = [&](int chunk, int buf) {
    unsigned char* d1 = smem + buf * BUF;
    unsigned char* d2 = smem + buf * BUF + W1B;
    const unsigned s1 = (unsigned)chunk * HC * RB1, s2 = (unsigned)chunk * HC * 2;
#pragma unroll
    for (int j = 0; j < PW1; ++j)
      if (j * NWV + wave < PIECES1)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w1_rsrc, (__attribute__((address_space(3))) void*)(d1 + (j * NWV + wave) * 1024), 16, off1[j], s1, 0, 0);
#pragma unroll
    for (int j = 0; j < PW2; ++j)
      if (j * NWV + wave < PIECES2)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w2_rsrc, (__attribute__((address_space(3))) void*)(d2 + (j * NWV + wave) * 1024), 16, off2[j], s2, 0, 0);
  };

  const int64_t pass_tokens = NWV * 32 * TB;
  const int64_t n_pass = (p.T + pass_tokens - 1) / pass_tokens;
  int64_t g_chunk = 0;                                       // running chunk count of this workgroup (buffer = parity)
  // RESIDENT (two chunks = the whole of W1 and W2 fit the two ring buffers: dim 96): both chunks are loaded ONCE and stay; the pass loop
  // then has no DMA, no wait and no barrier, so the 8 waves drift apart and one wave's LayerNorm prologue / store epilogue runs under the
  // others' MFMAs (round 4).  Otherwise the ring streams chunk c + 1 under chunk c, one barrier per chunk.
  constexpr bool RESIDENT = (NCH == 2) && NWV == 8;
  if ((int64_t)blockIdx.x < n_pass) {
    issue(0, 0);
    if constexpr (RESIDENT) issue(1, 1);
  }
  if constexpr (RESIDENT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                           // parameters visible (streaming form: the DMA is waited for inside the loop)

  for (int64_t pass = blockIdx.x; pass < n_pass; pass += gridDim.x) {
    // ---- this wave's TB token blocks: LayerNorm -> 16-bit B fragments
    op16x8 xf[TB][KS1];
    int64_t tok[TB];
#pragma unroll
    for (int tb = 0; tb < TB; ++tb) {
      const int64_t t = pass * pass_tokens + (int64_t)(wave * TB + tb) * 32 + r;
      tok[tb] = t;
      const float* xr = p.x + (t < p.T ? t : p.T - 1) * DIM;
      f32x4 v[KS1][2];
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < KS1; ++k)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          v[k][q] = *reinterpret_cast<const f32x4*>(xr + 16 * k + 8 * h + 4 * q);
          s += v[k][q][0] + v[k][q][1] + v[k][q][2] + v[k][q][3];
        }
      s += __shfl_xor(s, 32, 64);
      const float mean = s * (1.f / DIM);
      float q2 = 0.f;
#pragma unroll
      for (int k = 0; k < KS1; ++k)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float dlt = v[k][q][e] - mean;
            q2 += dlt * dlt;
          }
      q2 += __shfl_xor(q2, 32, 64);
      const float rstd = 1.0f / sqrtf(q2 * (1.f / DIM) + p.eps);
#pragma unroll
      for (int k = 0; k < KS1; ++k) {
        op16x8 f;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const f32x4 gm = *reinterpret_cast<const f32x4*>(prm + 16 * k + 8 * h + 4 * q);
          const f32x4 bt = *reinterpret_cast<const f32x4*>(prm + DIM + 16 * k + 8 * h + 4 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) f[4 * q + e] = f2op((v[k][q][e] - mean) * rstd * gm[e] + bt[e]);
        }
        xf[tb][k] = f;
      }
    }
    f32x16 accy[TB][DB];
#pragma unroll
    for (int tb = 0; tb < TB; ++tb)
#pragma unroll
      for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) accy[tb][d][e] = 0.f;

    for (int chunk = 0; chunk < NCH; ++chunk, ++g_chunk) {
      const int buf = RESIDENT ? chunk : (int)(g_chunk & 1);
      if constexpr (!RESIDENT) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of the chunk have landed
        __builtin_amdgcn_s_barrier();                        // ... every wave's; the other buffer is no longer read
        const bool more_here = chunk + 1 < NCH;
        const bool more = more_here || (pass + gridDim.x < n_pass);
        if (more) issue(more_here ? chunk + 1 : 0, buf ^ 1);
      }
      const unsigned char* w1s = smem + buf * BUF;
      const unsigned char* w2s = w1s + W1B;
#pragma unroll 1
      for (int hb = 0; hb < HB; ++hb) {
        // fc1 fragments of this hidden block: read once, used by all TB token blocks
        op16x8 a1[KS1];
        const int row1 = hb * 32 + r;
#pragma unroll
        for (int k = 0; k < KS1; ++k) a1[k] = *reinterpret_cast<const op16x8*>(w1s + row1 * RB1 + (mlp_swz<RB1>(2 * k + h, row1) << 4));
        // bias of the hidden rows this lane holds: (e & 3) + 8 (e >> 2) + 4 h  -> four float4 reads
        f32x4 bias1[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) bias1[g] = *reinterpret_cast<const f32x4*>(prm + 3 * DIM + chunk * HC + hb * 32 + 8 * g + 4 * h);
#pragma unroll
        for (int tb = 0; tb < TB; ++tb) {
          f32x16 acch;
#pragma unroll
          for (int e = 0; e < 16; ++e) acch[e] = bias1[e >> 2][e & 3];
#pragma unroll
          for (int k = 0; k < KS1; ++k) acch = MSAM2_MFMA_32x32x16(a1[k], xf[tb][k], acch, 0, 0, 0);
          op16x8 hf[2];
#pragma unroll
#ifdef MSAM2_MLP_NO_GELU   // timing experiment only: what the activation costs
          for (int e = 0; e < 16; ++e) hf[e >> 3][e & 7] = f2op(acch[e]);
#else
          for (int e = 0; e < 16; e += 2) {               // pairs: packed fp32 arithmetic (common.h)
            const f32x2 gp = gelu_erf2(f32x2{acch[e], acch[e + 1]});
            hf[e >> 3][e & 7] = f2op(gp[0]);
            hf[e >> 3][(e & 7) + 1] = f2op(gp[1]);
          }
#endif
          // fc2: Y^T[d][token] += W2[d][hidden] H^T[hidden][token]; operand k order = the accumulator's register order (permuted W2)
#pragma unroll
          for (int d = 0; d < DB; ++d) {
            const int row2 = d * 32 + r;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
              const op16x8 a2 = *reinterpret_cast<const op16x8*>(w2s + row2 * RB2 + (mlp_swz<RB2>(hb * 4 + 2 * s2 + h, row2) << 4));
              accy[tb][d] = MSAM2_MFMA_32x32x16(a2, hf[s2], accy[tb][d], 0, 0, 0);
            }
          }
        }
      }
    }
    // ---- epilogue: + b2 + residual, fp32 store (a lane owns a token; 4 consecutive channels per register group)
#pragma unroll
    for (int tb = 0; tb < TB; ++tb) {
      if (tok[tb] < p.T) {
        const float* xr = p.x + tok[tb] * DIM;
        float* yr = p.out + tok[tb] * DIM;
        // every residual value of the token block is loaded BEFORE its first store (round 4): loads and stores share one in-order
        // counter (vmcnt), so as load / add / store triples each residual load waited for the previous store's round trip -- DB * 4
        // of them per block, one after the other
        constexpr int DG = 1;                             // d-blocks per batch (4 float4 in flight; 3 = 12 float4 spills 19-35 registers and measures no gain: 120 -> 115 us, 82.5 -> 77.9 us with 1)
        static_assert(DB % DG == 0, "d-blocks in whole batches");
#pragma unroll
        for (int d0 = 0; d0 < DB; d0 += DG) {
          f32x4 xv[DG][4];
#pragma unroll
          for (int d = 0; d < DG; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) xv[d][g] = *reinterpret_cast<const f32x4*>(xr + (d0 + d) * 32 + 8 * g + 4 * h);
#pragma unroll
          for (int d = 0; d < DG; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int ch = (d0 + d) * 32 + 8 * g + 4 * h;
              const f32x4 bv = *reinterpret_cast<const f32x4*>(prm + 2 * DIM + ch);
              f32x4 o;
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = accy[tb][d0 + d][4 * g + e] + bv[e] + xv[d][g][e];
              *reinterpret_cast<f32x4*>(yr + ch) = o;
              if (p.out16) {
                op16x4 o2;
#pragma unroll
                for (int e = 0; e < 4; ++e) o2[e] = f2op(o[e]);
                *reinterpret_cast<op16x4*>(p.out16 + tok[tb] * DIM + ch) = o2;
              }
            }
        }
      }
    }
  }
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// dim 384 (Hiera stage 3: 11 blocks of hiera_s, 16384 tokens at 4 x 1024^2).  OPT-IN (MSAM2_MLP_384=1), NOT the default: correct
// (tests/test_kernels_gpu.py::test_ln_mlp_residual_fused, dim 384) and, as it stands, SLOWER than the three launches it would replace
// (LayerNorm 9 + fc1 33 + fc2 32 us in the step; 90 us as an isolated sequence; this kernel 110-118 us).  Why it was built: what bounds the
// two GEMMs is what a CU can take in (~33 GB/s per CU of activations out of the Infinity Cache, ~70 GB/s of weights out of its
// XCD's L2), and the 50 MB hidden map goes out and comes back through exactly that path; fused, a CU takes in 96 KB of activations and the
// 2.36 MB of W1 | W2 from L2 (~33 us at 70 GB/s) and the hidden map never exists.  Why it does not pay yet (removal ladder, round 4,
// 16384 tokens): row loads / residual / store in the token-per-lane layout 23 us, the 36 MFMAs per wave and chunk 41 us, LDS fragment
// reads 20 us, weight DMA issue + landing 18 us, GELU 14 us -- and the parts ADD UP to the kernel's time: with ONE wave per SIMD (the
// 144 KB ring + 96 + 96 row / accumulator registers leave room for no second one) nothing overlaps them; independent MFMA chains,
// read sub-bursts and sched_group_barrier-placed GELU instructions changed the listing and not the time.  What would: a second wave per
// SIMD (needs the fc1 duplication below gone: 8 waves of this split are MFMA-bound at 44 us) and row-major I / O staged through LDS.
//
// Same transposed arithmetic as mlp_fused_kernel (a lane owns a token; the GELU'd fc1 accumulator IS fc2's B operand), other work split:
//   * one workgroup = 64 tokens (256 workgroups at 16384 tokens: one per CU), 4 waves = 2 token blocks x 2 halves of the OUTPUT channels.
//     Both waves of a token block compute the block's fc1 (duplicated: 24 MFMAs) and each accumulates its 6 of the 12 fc2 output blocks
//     (12 MFMAs): 36 MFMAs per wave and 32-unit hidden chunk, 96 accumulator registers instead of 192 -- the whole row state of a wave
//     (96 operand + 96 accumulator registers) fits one SIMD's file;
//   * the weights stream in hidden chunks of 32 units (W1 rows [32][384] = 24 KB, W2 columns [384][32] = 24 KB) through a THREE-slot
//     LDS-DMA ring, two chunks (96 KB) in flight, one barrier per chunk;
//   * W2 is given chunk-major and already in its LDS image (msam2_mlp_fused_permute_w2 at dim 384): its DMA is a linear 24 KB copy of whole
//     128-byte lines (as [384][1536] rows a chunk would be 384 pieces of 64 bytes).
// LDS images: W1 rows of 768 B, 16-byte chunk c of row r at (c & ~15) | ((c & 15) ^ (r & 15)); W2 rows of 64 B, chunk c at c ^ ((r >> 2) & 3)
// (both conflict free for the 16-lane groups of a 32-row ds_read_b128 fragment read).
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int mlp_swz768(int c, int r) { return (c & ~15) | ((c & 15) ^ (r & 15)); }

__global__ __launch_bounds__(256, 1) void mlp_fused384_kernel(MlpFusedParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int DIM = 384, HID = 1536, HC = 32, NCH = HID / HC, KS1 = DIM / 16, DBW = 6, NST = 3;
  constexpr int W1B = HC * DIM * 2, W2B = DIM * HC * 2, BUF = W1B + W2B;   // 24 + 24 KB
  constexpr int PW = W1B / 1024 / 4;                                      // DMA pieces (1 KB) per wave and operand: 6
  static_assert(NST * BUF + (3 * DIM + HID) * 4 <= 160 * 1024, "LDS budget");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  float* prm = reinterpret_cast<float*>(smem + NST * BUF);   // [gamma DIM | beta DIM | b2 DIM | b1 HID]

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int r = lane & 31, h = lane >> 5;
  const int tb = wave >> 1, dh = wave & 1;
  for (int i = tid; i < DIM; i += 256) {
    prm[i] = p.ln_w[i];
    prm[DIM + i] = p.ln_b[i];
    prm[2 * DIM + i] = p.b2[i];
  }
  for (int i = tid; i < HID; i += 256) prm[3 * DIM + i] = p.b1[i];

  // ---- this wave's token block: row loads first (the weight DMAs queue behind them on the in-order counter, not in front)
  const int64_t t = (int64_t)blockIdx.x * 64 + tb * 32 + r;
  const float* xr = p.x + (t < p.T ? t : p.T - 1) * DIM;
  f32x4 v[KS1][2];
#pragma unroll
  for (int k = 0; k < KS1; ++k)
#pragma unroll
    for (int q = 0; q < 2; ++q) v[k][q] = *reinterpret_cast<const f32x4*>(xr + 16 * k + 8 * h + 4 * q);

  // ---- weight stream
  const auto w1_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1, 0, HID * DIM * 2, 0x00020000);
  const auto w2_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2p, 0, HID * DIM * 2, 0x00020000);
  unsigned off1[PW];
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    const int f = (j * 4 + wave) * 64 + lane, row = f / 48, c = f % 48;
    off1[j] = (unsigned)(row * 768 + mlp_swz768(c, row) * 16);
  }
  const unsigned off2 = (unsigned)(wave * 1024 + lane * 16);
  // ring step i (0 .. NCH) holds W1 chunk i (i < NCH) and W2 chunk i - 1 (i >= 1): fc1 runs ONE chunk ahead of fc2, so the GELU of chunk
  // i - 1 (vector pipe) sits beside the fc1 MFMAs of chunk i instead of between two dependent MFMA groups
  auto issue = [&](int i) {
    unsigned char* d1 = smem + (i % NST) * BUF;
    unsigned char* d2 = d1 + W1B;
    if (i < NCH) {
      const unsigned s = (unsigned)i * W1B;
#pragma unroll
      for (int j = 0; j < PW; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w1_rsrc, (__attribute__((address_space(3))) void*)(d1 + (j * 4 + wave) * 1024), 16, off1[j], s, 0, 0);
    }
    if (i >= 1) {
      const unsigned s = (unsigned)(i - 1) * W2B;
#pragma unroll
      for (int j = 0; j < PW; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w2_rsrc, (__attribute__((address_space(3))) void*)(d2 + (j * 4 + wave) * 1024), 16, off2, s + j * 4096, 0, 0);
    }
  };
  issue(0);
  issue(1);

  // ---- LayerNorm on the lane's half row -> 16-bit B fragments
  op16x8 xf[KS1];
  {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < KS1; ++k)
#pragma unroll
      for (int q = 0; q < 2; ++q) s += v[k][q][0] + v[k][q][1] + v[k][q][2] + v[k][q][3];
    s += __shfl_xor(s, 32, 64);
    const float mean = s * (1.f / DIM);
    float q2 = 0.f;
#pragma unroll
    for (int k = 0; k < KS1; ++k)
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dlt = v[k][q][e] - mean;
          q2 += dlt * dlt;
        }
    q2 += __shfl_xor(q2, 32, 64);
    const float rstd = 1.0f / sqrtf(q2 * (1.f / DIM) + p.eps);
    // parameters visible: the ds_writes have landed (lgkmcnt) + barrier; NOT __syncthreads(), whose fence would also wait for the 96 KB
    // of weight DMAs just issued
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int k = 0; k < KS1; ++k) {
      op16x8 f;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f32x4 gm = *reinterpret_cast<const f32x4*>(prm + 16 * k + 8 * h + 4 * q);
        const f32x4 bt = *reinterpret_cast<const f32x4*>(prm + DIM + 16 * k + 8 * h + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) f[4 * q + e] = f2op((v[k][q][e] - mean) * rstd * gm[e] + bt[e]);
      }
      xf[k] = f;
    }
  }
  f32x16 accy[DBW];
#pragma unroll
  for (int d = 0; d < DBW; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) accy[d][e] = 0.f;

  // One ring step: fc1 of hidden chunk `chunk` (W1 image at w1s) beside the GELU of the previous chunk's accumulator `aprev` -> hf.
  //   * W1 fragments in three sub-bursts of 8 (two register sets of 32): a burst is read while the previous one's 8 MFMAs run.  (Read ->
  //     wait -> MFMA one by one the chain ran at the LDS latency; all 24 at once cost 96 registers, which the compiler parked in AGPRs:
  //     184 v_accvgpr moves per step.)
  //   * two independent accumulation chains (even / odd k), summed at the end;
  //   * the GELU's vector instructions are placed BETWEEN the MFMAs (sched_group_barrier pattern: 1 MFMA, 6 VALU): one wave per SIMD, so
  //     nothing else fills the matrix pipe's 32-cycle slots.
  auto fc1_gelu = [&](const unsigned char* w1s, int chunk, f32x16& acch, const f32x16* aprev, op16x8 (&hf)[2]) {
    op16x8 fa[8], fb[8];
    auto rd = [&](op16x8 (&f)[8], int k0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) f[k] = *reinterpret_cast<const op16x8*>(w1s + r * 768 + (mlp_swz768(2 * (k0 + k) + h, r) << 4));
    };
    rd(fa, 0);
    rd(fb, 8);
    f32x4 bias1[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bias1[g] = *reinterpret_cast<const f32x4*>(prm + 3 * DIM + chunk * HC + 8 * g + 4 * h);
    f32x16 part[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      part[0][e] = bias1[e >> 2][e & 3];
      part[1][e] = 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 8; ++k) part[k & 1] = MSAM2_MFMA_32x32x16(fa[k], xf[k], part[k & 1], 0, 0, 0);
    if (aprev) {
#pragma unroll
      for (int e = 0; e < 6; e += 2) {
        const f32x2 gp = gelu_erf2(f32x2{(*aprev)[e], (*aprev)[e + 1]});
        hf[e >> 3][e & 7] = f2op(gp[0]);
        hf[e >> 3][(e & 7) + 1] = f2op(gp[1]);
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    rd(fa, 16);
#pragma unroll
    for (int k = 0; k < 8; ++k) part[k & 1] = MSAM2_MFMA_32x32x16(fb[k], xf[8 + k], part[k & 1], 0, 0, 0);
    if (aprev) {
#pragma unroll
      for (int e = 6; e < 12; e += 2) {
        const f32x2 gp = gelu_erf2(f32x2{(*aprev)[e], (*aprev)[e + 1]});
        hf[e >> 3][e & 7] = f2op(gp[0]);
        hf[e >> 3][(e & 7) + 1] = f2op(gp[1]);
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
      __builtin_amdgcn_sched_group_barrier(0x002, 6, 1);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 8; ++k) part[k & 1] = MSAM2_MFMA_32x32x16(fa[k], xf[16 + k], part[k & 1], 0, 0, 0);
    if (aprev) {
#pragma unroll
      for (int e = 12; e < 16; e += 2) {
        const f32x2 gp = gelu_erf2(f32x2{(*aprev)[e], (*aprev)[e + 1]});
        hf[e >> 3][e & 7] = f2op(gp[0]);
        hf[e >> 3][(e & 7) + 1] = f2op(gp[1]);
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 2);
      __builtin_amdgcn_sched_group_barrier(0x002, 5, 2);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 16; ++e) acch[e] = part[0][e] + part[1][e];
  };
  auto gelu16 = [&](const f32x16& acch, op16x8 (&hf)[2]) {
#pragma unroll
    for (int e = 0; e < 16; e += 2) {
      const f32x2 gp = gelu_erf2(f32x2{acch[e], acch[e + 1]});
      hf[e >> 3][e & 7] = f2op(gp[0]);
      hf[e >> 3][(e & 7) + 1] = f2op(gp[1]);
    }
  };
  auto fc2 = [&](const unsigned char* w2s, const op16x8 (&hf)[2]) {
    op16x8 a2[DBW][2];
#pragma unroll
    for (int d = 0; d < DBW; ++d) {
      const int row2 = (dh * DBW + d) * 32 + r;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) a2[d][s2] = *reinterpret_cast<const op16x8*>(w2s + row2 * 64 + (((2 * s2 + h) ^ ((row2 >> 2) & 3)) << 4));
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)                           // s2 outer: six independent accumulators between the two MFMAs of an output block
#pragma unroll
      for (int d = 0; d < DBW; ++d) accy[d] = MSAM2_MFMA_32x32x16(a2[d][s2], hf[s2], accy[d], 0, 0, 0);
  };

  f32x16 acur;
  // step 0: W1 chunk 0 only
  asm volatile("s_waitcnt vmcnt(12)" ::: "memory");          // step 1's 12 pieces may still be in flight
  __builtin_amdgcn_s_barrier();
  issue(2);
  {
    op16x8 hf0[2];
    fc1_gelu(smem, 0, acur, nullptr, hf0);
  }
#pragma unroll 1
  for (int i = 1; i < NCH; ++i) {
    // this wave's pieces of step i have landed (step i + 1's 12 -- 6 for the last step -- may still be in flight) ...
    if (i + 1 < NCH) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();                            // ... every wave's; the slot of step i - 1 is no longer read
    if (i + 2 <= NCH) issue(i + 2);
    const unsigned char* w1s = smem + (i % NST) * BUF;
    f32x16 anext;
    op16x8 hf[2];
    fc1_gelu(w1s, i, anext, &acur, hf);                      // chunk i on the matrix pipe, chunk i - 1's GELU on the vector pipe
    fc2(w1s + W1B, hf);                                      // chunk i - 1
    acur = anext;
  }
  {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    op16x8 hf[2];
    gelu16(acur, hf);
    fc2(smem + (NCH % NST) * BUF + W1B, hf);
  }

  // ---- epilogue: + b2 + residual; every residual value of three output blocks is loaded before their first store
  if (t < p.T) {
    const float* xres = p.x + t * DIM;
    float* yr = p.out + t * DIM;
    constexpr int DG = 3;
#pragma unroll
    for (int d0 = 0; d0 < DBW; d0 += DG) {
      f32x4 xv[DG][4];
#pragma unroll
      for (int d = 0; d < DG; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) xv[d][g] = *reinterpret_cast<const f32x4*>(xres + (dh * DBW + d0 + d) * 32 + 8 * g + 4 * h);
#pragma unroll
      for (int d = 0; d < DG; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int ch = (dh * DBW + d0 + d) * 32 + 8 * g + 4 * h;
          const f32x4 bv = *reinterpret_cast<const f32x4*>(prm + 2 * DIM + ch);
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = accy[d0 + d][4 * g + e] + bv[e] + xv[d][g][e];
          *reinterpret_cast<f32x4*>(yr + ch) = o;
          if (p.out16) {
            op16x4 o2;
#pragma unroll
            for (int e = 0; e < 4; ++e) o2[e] = f2op(o[e]);
            *reinterpret_cast<op16x4*>(p.out16 + t * DIM + ch) = o2;
          }
        }
    }
  }
#endif
}

// W2 [384, 1536] for mlp_fused384_kernel: chunk-major [48][384][32], each 64-byte row in its LDS image (16-byte chunk c at c ^ ((d >> 2) & 3)),
// hidden units of a chunk in the operand order of mlp_fused_permute_kernel
__global__ void mlp_fused384_pack_kernel(const op16* __restrict__ w2, op16* __restrict__ w2q, int dim, int hid) {
  const int64_t total = (int64_t)dim * hid;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int chunk = (int)(i / ((int64_t)dim * 32)), d = (int)((i / 32) % dim), q = (int)(i & 31);
    const int c = (q >> 3) ^ ((d >> 2) & 3), j = q & 7, pos = c * 8 + j;
    const int s = pos >> 4, hh = (pos >> 3) & 1;
    w2q[i] = w2[(int64_t)d * hid + chunk * 32 + 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3)];
  }
}

// hidden-dimension permutation of W2 [DIM, HID] that matches the k order in which an H^T accumulator is consumed as the B operand:
// position 16 s + 8 h + j of every 32-block holds hidden unit 16 s + 8 (j >> 2) + 4 h + (j & 3).
__global__ void mlp_fused_permute_kernel(const op16* __restrict__ w2, op16* __restrict__ w2p, int dim, int hid) {
  const int64_t total = (int64_t)dim * hid;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int d = (int)(i / hid), pos = (int)(i % hid);
    const int blk = pos >> 5, q = pos & 31, s = q >> 4, hh = (q >> 3) & 1, j = q & 7;
    w2p[i] = w2[(int64_t)d * hid + blk * 32 + 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3)];
  }
}

extern "C" int msam2_mlp_fused_permute_w2(const void* w2, void* w2p, int64_t dim, int64_t hidden, void* stream) {
  MSAM2_REQUIRE(w2 && w2p && dim > 0 && hidden % 32 == 0, "mlp_fused_permute_w2: bad arguments");
  if (dim == 384) {                                          // mlp_fused384_kernel's chunk-major LDS image
    MSAM2_REQUIRE(hidden == 1536, "mlp_fused_permute_w2: dim 384 takes hidden 1536");
    hipLaunchKernelGGL(mlp_fused384_pack_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, (const op16*)w2, (op16*)w2p, (int)dim, (int)hidden);
    return msam2_check_launch("mlp_fused_permute_w2(384)");
  }
  hipLaunchKernelGGL(mlp_fused_permute_kernel, dim3((unsigned)min((int64_t)1024, cdiv(dim * hidden, 256))), dim3(256), 0, (hipStream_t)stream,
                     (const op16*)w2, (op16*)w2p, (int)dim, (int)hidden);
  return msam2_check_launch("mlp_fused_permute_w2");
}

// 1 when the trunk should take msam2_ln_mlp_residual_fwd at this width (hidden = 4 * dim, GELU): dim 96 / 192 (Hiera stages 1 / 2; hiera_b+'s
// 112 / 224 are not built).  Dim 384 is built and callable but slower than its three launches (see mlp_fused384_kernel): MSAM2_MLP_384=1
// makes the trunk use it (A/B).
extern "C" int msam2_ln_mlp_residual_supported(int64_t dim) {
  static const bool use384 = [] { const char* e = getenv("MSAM2_MLP_384"); return e && e[0] == '1'; }();
  return dim == 96 || dim == 192 || (dim == 384 && use384);
}

static int launch_mlp_fused384(const MlpFusedParams& p, hipStream_t s) {
  constexpr int LDS = 3 * 49152 + (3 * 384 + 1536) * 4;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)mlp_fused384_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  hipLaunchKernelGGL(mlp_fused384_kernel, dim3((unsigned)((p.T + 63) / 64)), dim3(256), LDS, s, p);
  return msam2_check_launch("ln_mlp_residual_fwd(384)");
}

template <int DIM, int HC, int TB, int OCC = 1, int NWV = 4>
static int launch_mlp_fused(const MlpFusedParams& p, hipStream_t s) {
  constexpr int LDS = 2 * (2 * HC * DIM * 2) + (3 * DIM + 4 * DIM) * 4;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)mlp_fused_kernel<DIM, HC, TB, OCC, NWV>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  const int64_t n_pass = (p.T + NWV * 32 * TB - 1) / (NWV * 32 * TB);
  hipLaunchKernelGGL((mlp_fused_kernel<DIM, HC, TB, OCC, NWV>), dim3((unsigned)min((int64_t)256 * OCC, n_pass)), dim3(NWV * 64), LDS, s, p);
  return msam2_check_launch("ln_mlp_residual_fwd");
}

// out[T, dim] (fp32) = x + fc2(GELU(fc1(LayerNorm(x)))) with x fp32 [T, dim] contiguous, w1 16-bit [4 dim, dim], w2p 16-bit [dim, 4 dim]
// permuted by msam2_mlp_fused_permute_w2, biases / LayerNorm parameters fp32.  dim in {96, 192, 384}.
static int ln_mlp_residual_launch(const float* x, int64_t T, int64_t dim, const float* ln_w, const float* ln_b, float eps, const void* w1,
                                  const float* b1, const void* w2p, const float* b2, float* out, void* out16, void* stream) {
  MSAM2_REQUIRE(x && out && ln_w && ln_b && w1 && b1 && w2p && b2 && T > 0, "ln_mlp_residual: null tensor / empty problem");
  MSAM2_REQUIRE(dim == 96 || dim == 192 || dim == 384, "ln_mlp_residual: dim %lld not built (96 / 192 / 384)", (long long)dim);
  MSAM2_REQUIRE((((uintptr_t)x | (uintptr_t)out | (uintptr_t)w1 | (uintptr_t)w2p | (uintptr_t)ln_w | (uintptr_t)ln_b | (uintptr_t)b1 | (uintptr_t)b2) & 15) == 0 &&
                    ((uintptr_t)out16 & 7) == 0, "ln_mlp_residual: 16-byte aligned tensors");
  MSAM2_REQUIRE(x != out, "ln_mlp_residual: in-place not supported (a token's residual is re-read in the store)");
  MlpFusedParams p = {x, out, ln_w, ln_b, b1, b2, (const op16*)w1, (const op16*)w2p, T, eps, (op16*)out16};
  hipStream_t s = (hipStream_t)stream;
  if (dim == 384) {
    MSAM2_REQUIRE((T + 63) / 64 < (1ll << 31), "ln_mlp_residual: too many tokens");
    return launch_mlp_fused384(p, s);
  }
  static const bool four = [] { const char* e = getenv("MSAM2_MLP_WAVES4"); return e && e[0] == '1'; }();   // A/B: the 4-wave form
  if (four) {
    if (dim == 96) return launch_mlp_fused<96, 192, 4>(p, s);
    return launch_mlp_fused<192, 96, 2>(p, s);
  }
  if (dim == 96) return launch_mlp_fused<96, 192, 2, 1, 8>(p, s);
  return launch_mlp_fused<192, 96, 1, 1, 8>(p, s);
}

extern "C" int msam2_ln_mlp_residual_fwd(const float* x, int64_t T, int64_t dim, const float* ln_w, const float* ln_b, float eps, const void* w1,
                                         const float* b1, const void* w2p, const float* b2, float* out, void* stream) {
  return ln_mlp_residual_launch(x, T, dim, ln_w, ln_b, eps, w1, b1, w2p, b2, out, nullptr, stream);
}

// the same with the result also written in the 16-bit operand type (out16 [T, dim]): the last block of a Hiera stage, whose output is the
// operand of the FPN's lateral 1x1 convolution (image_encoder.py:95-110) -- no cast pass over the largest feature maps
extern "C" int msam2_ln_mlp_residual_fwd_dual(const float* x, int64_t T, int64_t dim, const float* ln_w, const float* ln_b, float eps,
                                              const void* w1, const float* b1, const void* w2p, const float* b2, float* out, void* out16,
                                              void* stream) {
  MSAM2_REQUIRE(out16, "ln_mlp_residual_dual: null 16-bit output");
  return ln_mlp_residual_launch(x, T, dim, ln_w, ln_b, eps, w1, b1, w2p, b2, out, out16, stream);
}
