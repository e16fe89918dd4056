// Backward building blocks for the fused LayerNorm / MLP pieces of the path (first slice): the
// gradient GEMMs reuse msam2_gemm (dX = dY W through a transposed weight copy, dW = dY^T X through transposed activations), so
// what is needed here is data movement (16-bit transpose), reductions (bias gradient) and the pointwise / row-wise derivative
// kernels.  Reference semantics: torch.autograd of nn.LayerNorm (hieradet.py:101-102, memory_attention.py:43-45), nn.GELU (exact
// erf) / nn.ReLU and nn.Linear (sam2_utils.py:108-132).
#include "common.h"
#include <stdlib.h>

// ------------------------------------------------------------------------------------------------------------------
// out[c][r] = in[r][c], 16-bit, 64x64 tiles through LDS (row pitch 66 halves: conflict-free both ways)
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose16_kernel(const op16* __restrict__ in, int64_t ldi, op16* __restrict__ out, int64_t ldo,
                                                          int R, int C) {
  __shared__ op16 tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < R && c < C) ? in[(int64_t)r * ldi + c] : f2op(0.f);
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < C && r < R) out[(int64_t)c * ldo + r] = tile[tx][i];
  }
}

extern "C" int msam2_transpose16(const void* in, int64_t ldi, void* out, int64_t ldo, int64_t rows, int64_t cols, void* stream) {
  MSAM2_REQUIRE(in && out && rows > 0 && cols > 0 && ldi >= cols && ldo >= rows, "transpose16: bad arguments");
  hipLaunchKernelGGL(transpose16_kernel, dim3(cdiv(cols, 64), cdiv(rows, 64)), dim3(256), 0, (hipStream_t)stream, (const op16*)in, ldi,
                     (op16*)out, ldo, (int)rows, (int)cols);
  return msam2_check_launch("transpose16");
}

// ------------------------------------------------------------------------------------------------------------------
// Weight-gradient GEMM without transposes: C[M, N] (fp32) = sum_k A[k][m] * B[k][n] with BOTH operands stored k-major
// (A = dY [tokens, out], B = X [tokens, in] exactly as the forward left them; C = dW [out, in]).  128x128x32 tiles, the 32 x 128
// operand slabs go row-major into LDS (row pitch 320 B = 64 mod 128) and both MFMA operands are read transposed with
// ds_read_b64_tr_b16 (the same k-order for both, so the dot products pair up).  The reduction (tokens: 16k..64k) is split over
// gridDim.z workgroups that add into the kernel-zeroed output with fp32 atomics.
// ------------------------------------------------------------------------------------------------------------------
struct GemmTTParams {
  const op16 *A, *B;
  float* C;
  float* a_colsum;   // optional [M]: sum_k A[k][m] (the bias gradient when A = dY), accumulated by the n-tile-0 workgroups into a zeroed vector
  int64_t lda, ldb, ldc;
  int M, N, K, ktiles_per_split;
  int accumulate;    // C and a_colsum already hold values to add to (msam2_gemm_tt_acc): no zeroing pass, every store is an atomic add
};

constexpr int TT_PITCH = 320;                 // bytes per LDS row (128 op16 + 64 B pad)
constexpr int TT_SLAB = 32 * TT_PITCH;        // one operand, one stage

__global__ __launch_bounds__(256) void gemm_tt_kernel(GemmTTParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[4 * TT_SLAB];   // [stage][A | B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5, li = lane & 15;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
  const int nk_all = (p.K + 31) / 32;
  const int kt0 = blockIdx.z * p.ktiles_per_split, kt1 = min(nk_all, kt0 + p.ktiles_per_split);
  if (kt0 >= kt1) return;
  uint4 ra[2], rb[2];
  const bool want_cs = p.a_colsum != nullptr && blockIdx.x == 0;
  float cs[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) cs[e] = 0.f;
  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + i * 256;
      const int row = c >> 4, col = (c & 15) * 8;
      const int64_t k = (int64_t)kt * 32 + row;
      uint4 a = make_uint4(0, 0, 0, 0), b = make_uint4(0, 0, 0, 0);
      if (k < p.K) {
        if (m0 + col < p.M) a = *reinterpret_cast<const uint4*>(p.A + k * p.lda + m0 + col);
        if (n0 + col < p.N) b = *reinterpret_cast<const uint4*>(p.B + k * p.ldb + n0 + col);
      }
      ra[i] = a;
      rb[i] = b;
      if (want_cs) {                                   // (workgroup-uniform branch)
        const op16x8 av = __builtin_bit_cast(op16x8, a);
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[e] += op2f(av[e]);
      }
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + i * 256;
      const int row = c >> 4, col = (c & 15) * 8;
      *reinterpret_cast<uint4*>(lds + (2 * buf) * TT_SLAB + row * TT_PITCH + col * 2) = ra[i];
      *reinterpret_cast<uint4*>(lds + (2 * buf + 1) * TT_SLAB + row * TT_PITCH + col * 2) = rb[i];
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // transposed-read lane offsets (same pattern as the attention kernels' V^T fragments): rows = k, columns = m / n
  const int tr_off = (4 * h + (li >> 2)) * TT_PITCH + (16 * ((lane >> 4) & 1) + 4 * (li & 3)) * 2;
  auto frag = [&](const unsigned char* slab, int st, int col0) -> op16x8 {
    const unsigned char* a0 = slab + tr_off + (16 * st) * TT_PITCH + col0 * 2;
    const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(a0));
    const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(a0 + 8 * TT_PITCH));
    typedef __attribute__((ext_vector_type(8))) short short8_t;
    short8_t t8;
    t8[0] = lo[0]; t8[1] = lo[1]; t8[2] = lo[2]; t8[3] = lo[3];
    t8[4] = hi[0]; t8[5] = hi[1]; t8[6] = hi[2]; t8[7] = hi[3];
    return __builtin_bit_cast(op16x8, t8);
  };
  gload(kt0);
  lstore(0);
  __syncthreads();
  int cur = 0;
  for (int kt = kt0; kt < kt1; ++kt) {
    if (kt + 1 < kt1) gload(kt + 1);
    const unsigned char* sa = lds + (2 * cur) * TT_SLAB;
    const unsigned char* sb = sa + TT_SLAB;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      op16x8 af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = frag(sa, st, wm * 64 + i * 32);
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[j] = frag(sb, st, wn * 64 + j * 32);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = MSAM2_MFMA_32x32x16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < kt1) lstore(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
  if (want_cs) {
    // thread (tid & 15) holds the sums of columns 8 (tid & 15) .. +8 over its rows: reduce the 16 row groups through LDS
    float* red = reinterpret_cast<float*>(lds);          // [16][128]; every wave is past its last fragment read (loop-end barrier)
#pragma unroll
    for (int e = 0; e < 8; ++e) red[(tid >> 4) * 128 + (tid & 15) * 8 + e] = cs[e];
    __syncthreads();
    if (tid < 128 && m0 + tid < p.M) {
      float t = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) t += red[g * 128 + tid];
      atomicAdd(p.a_colsum + m0 + tid, t);
    }
  }
  const bool atomic = gridDim.z > 1 || p.accumulate;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 64 + j * 32 + r;
    if (n >= p.N) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (m < p.M) {
          if (atomic) atomicAdd(p.C + (int64_t)m * p.ldc + n, acc[i][j][e]);
          else p.C[(int64_t)m * p.ldc + n] = acc[i][j][e];
        }
      }
  }
}

// LDS-DMA form of gemm_tt_kernel (round 3): the register-staged kernel above runs 8 MFMAs per wave and barrier behind a global-load
// round trip (150 TFLOP/s over the training iteration's weight gradients -- 13.6 % of the iteration, profiles/r03_train_*); this one is
// gemm_glds_kernel's structure on k-major operands: 128 x 128 x 64 tiles, both 64-row x 256-byte slabs streamed global -> LDS by DMA
// (one 1-KiB piece = 4 k-rows; 8 pieces per wave and tile), two stages, one barrier per 64 k; all 16 operand fragments of a tile are
// read in one burst (transposed reads, the V^T pattern of attn_glds_kernel<128>: 16-byte chunk c of row k sits at c ^ ((k & 3) << 2),
// applied on the DMA's source address) and the 16 MFMAs run with the next tile's DMA pieces issued in their shadow.
// K % 64 == 0 (rows past K cannot be zero-filled by a DMA); columns past M / N re-read chunk 0 of their row (results not stored).
__global__ __launch_bounds__(256, 2) void gemm_tt_dma_kernel(GemmTTParams p) {
  constexpr int BK = 64, RB = 256, SLAB = BK * RB, STAGE = 2 * SLAB;   // 32 KiB per stage
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int r = lane & 31, h = lane >> 5, li = lane & 15;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
  const int nk_all = p.K / BK;
  const int kt0 = blockIdx.z * p.ktiles_per_split, kt1 = min(nk_all, kt0 + p.ktiles_per_split);
  if (kt0 >= kt1) return;
  const bool want_cs = p.a_colsum != nullptr && blockIdx.x == 0;      // (workgroup-uniform)

  const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7fffffff, 0x00020000);
  const auto b_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, 0x7fffffff, 0x00020000);
  // piece i of this wave: k-rows 4 (4 wave + i) .. +3 of the slab; lane -> (row, LDS chunk slot), source chunk = slot ^ ((row & 3) << 2)
  unsigned offA[4], offB[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 4 * (4 * wave + i) + (lane >> 4);
    const int col = (((lane & 15) ^ ((row & 3) << 2))) * 8;
    offA[i] = (unsigned)(row * p.lda * 2 + (m0 + (m0 + col < p.M ? col : 0)) * 2);
    offB[i] = (unsigned)(row * p.ldb * 2 + (n0 + (n0 + col < p.N ? col : 0)) * 2);
  }
  auto issue_piece = [&](int kt, int stage, int i) {
    unsigned char* base = lds + stage * STAGE + (4 * wave) * 1024;
    if (i < 4)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (__attribute__((address_space(3))) void*)(base + i * 1024), 16, offA[i],
                                               (unsigned)kt * BK * (unsigned)p.lda * 2u, 0, 0);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, (__attribute__((address_space(3))) void*)(base + SLAB + (i - 4) * 1024), 16, offB[i - 4],
                                               (unsigned)kt * BK * (unsigned)p.ldb * 2u, 0, 0);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  float cs[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) cs[e] = 0.f;

  // transposed fragment of 32 columns (block cb of the slab) x 16 k-rows (step ks): rows 16 ks + 4 h + vq (+ 8), see attn_glds_kernel
  const int vq = li >> 2, vp = li & 3, cgrp = (lane >> 4) & 1;
  const int t_row = (4 * h + vq) * RB + ((vp & 1) << 3);
  const int t_sw = vq << 2, t_c0 = 2 * cgrp + (vp >> 1);
  typedef __attribute__((ext_vector_type(8))) short short8_t;
  auto frag = [&](const unsigned char* slab, int ks, int cb) -> op16x8 {
    const unsigned char* a0 = slab + t_row + (16 * ks) * RB + (((cb * 4 + t_c0) ^ t_sw) << 4);
    const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(a0));
    const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(a0 + 8 * RB));
    short8_t t8;
    t8[0] = lo[0]; t8[1] = lo[1]; t8[2] = lo[2]; t8[3] = lo[3];
    t8[4] = hi[0]; t8[5] = hi[1]; t8[6] = hi[2]; t8[7] = hi[3];
    return __builtin_bit_cast(op16x8, t8);
  };

#pragma unroll
  for (int i = 0; i < 8; ++i) issue_piece(kt0, 0, i);
  for (int kt = kt0; kt < kt1; ++kt) {
    const int st = (kt - kt0) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's DMA pieces of tile kt have landed
    __builtin_amdgcn_s_barrier();                        // ... and so have every other wave's; stage st^1 is free again
    const unsigned char* sa = lds + st * STAGE;
    const unsigned char* sb = sa + SLAB;
    op16x8 af[4][2], bfr[4][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[ks][i] = frag(sa, ks, wm * 2 + i);
        bfr[ks][i] = frag(sb, ks, wn * 2 + i);
      }
    if (want_cs) {
      // bias gradient: thread t adds rows (t >> 4) + 16 u of column chunk t & 15 of the A slab
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int row = (tid >> 4) + 16 * u;
        const op16x8 av = *reinterpret_cast<const op16x8*>(sa + row * RB + ((((tid & 15) ^ ((row & 3) << 2))) << 4));
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[e] += op2f(av[e]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = MSAM2_MFMA_32x32x16(af[ks][i], bfr[ks][j], acc[i][j], 0, 0, 0);
      if (kt + 1 < kt1) {
        issue_piece(kt + 1, st ^ 1, 2 * ks);
        issue_piece(kt + 1, st ^ 1, 2 * ks + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (want_cs) {
    __builtin_amdgcn_s_barrier();                        // every wave is past its last fragment read
    float* red = reinterpret_cast<float*>(lds);          // [16][128]
#pragma unroll
    for (int e = 0; e < 8; ++e) red[(tid >> 4) * 128 + (tid & 15) * 8 + e] = cs[e];
    __syncthreads();
    if (tid < 128 && m0 + tid < p.M) {
      float t = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) t += red[g * 128 + tid];
      atomicAdd(p.a_colsum + m0 + tid, t);
    }
  }
  const bool atomic = gridDim.z > 1 || p.accumulate;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 64 + j * 32 + r;
    if (n >= p.N) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (m < p.M) {
          if (atomic) atomicAdd(p.C + (int64_t)m * p.ldc + n, acc[i][j][e]);
          else p.C[(int64_t)m * p.ldc + n] = acc[i][j][e];
        }
      }
  }
}

__global__ __launch_bounds__(256) void gemm_tt_zero_kernel(float* __restrict__ C, int64_t ldc, int M, int N, float* __restrict__ colsum, int zero_c) {
  const int64_t total = zero_c ? (int64_t)M * N : 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) C[(i / N) * ldc + i % N] = 0.f;
  if (colsum)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < M; i += (int64_t)gridDim.x * 256) colsum[i] = 0.f;
}

static int gemm_tt_impl(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, float* a_colsum, int64_t M, int64_t N,
                        int64_t K, int accumulate, void* stream) {
  MSAM2_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0, "gemm_tt: bad arguments");
  MSAM2_REQUIRE(M % 8 == 0 && N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0,
                "gemm_tt: M, N, lda, ldb must be multiples of 8 and the operands 16-byte aligned");
  MSAM2_REQUIRE(M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31) && lda >= M && ldb >= N && ldc >= N, "gemm_tt: bad sizes");
  GemmTTParams p;
  p.A = (const op16*)A; p.B = (const op16*)B; p.C = C; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.a_colsum = a_colsum;
  p.M = (int)M; p.N = (int)N; p.K = (int)K;
  p.accumulate = accumulate;
  // LDS-DMA kernel: 64-row k-tiles, 32-bit DMA offsets (the whole operand within 2 GiB)
  static const bool no_dma = getenv("MSAM2_GEMM_TT_V1") != nullptr;
  const bool dma = !no_dma && K % 64 == 0 && K * lda * 2 < (1ll << 31) && K * ldb * 2 < (1ll << 31);
  const int64_t tiles = (int64_t)cdiv(M, 128) * cdiv(N, 128), nk = dma ? K / 64 : cdiv(K, 32);
  // splits: ~1 workgroup per CU (2 for the register-staged kernel), each split at least 8 (16) k-tiles long -- swept on the training
  // iteration's 20 shapes (tools/gemm_tt_bench.py, ms per iteration): register-staged 6.04; DMA kernel 4.85 / 4.07 / 3.92 / 3.96 / 4.39
  // / 5.38 at 128 / 192 / 256 / 320 / 512 / 1024 workgroups aimed at (more splits = more fp32 atomics per output element)
  static const int64_t env_mink = getenv("MSAM2_TT_MINK") ? atoll(getenv("MSAM2_TT_MINK")) : 0;   // sweeps (tools/gemm_tt_bench.py)
  static const int64_t env_wgs = getenv("MSAM2_TT_WGS") ? atoll(getenv("MSAM2_TT_WGS")) : 0;
  const int64_t mink = env_mink > 0 ? env_mink : (dma ? 8 : 16), wgs = env_wgs > 0 ? env_wgs : (dma ? 256 : 512);
  int64_t splits = max((int64_t)1, min(min((int64_t)128, cdiv(wgs, tiles)), nk / mink));
  p.ktiles_per_split = (int)cdiv(nk, splits);
  splits = cdiv(nk, p.ktiles_per_split);
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate && (splits > 1 || a_colsum))
    hipLaunchKernelGGL(gemm_tt_zero_kernel, dim3((unsigned)min((int64_t)1024, (M * N + 255) / 256)), dim3(256), 0, s, C, ldc, (int)M, (int)N, a_colsum,
                       splits > 1 ? 1 : 0);
  if (dma) hipLaunchKernelGGL(gemm_tt_dma_kernel, dim3(cdiv(N, 128), cdiv(M, 128), (unsigned)splits), dim3(256), 0, s, p);
  else hipLaunchKernelGGL(gemm_tt_kernel, dim3(cdiv(N, 128), cdiv(M, 128), (unsigned)splits), dim3(256), 0, s, p);
  return msam2_check_launch("gemm_tt");
}

extern "C" int msam2_gemm_tt(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, float* a_colsum, int64_t M, int64_t N,
                             int64_t K, void* stream) {
  return gemm_tt_impl(A, lda, B, ldb, C, ldc, a_colsum, M, N, K, 0, stream);
}

// msam2_gemm_tt that ADDS into C (and a_colsum): gradient accumulation, or outputs carved from a buffer that was zeroed once for a whole
// backward pass (the per-call zeroing launch of msam2_gemm_tt is 150 launches per training iteration).
extern "C" int msam2_gemm_tt_acc(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, float* a_colsum, int64_t M,
                                 int64_t N, int64_t K, void* stream) {
  return gemm_tt_impl(A, lda, B, ldb, C, ldc, a_colsum, M, N, K, 1, stream);
}

// ------------------------------------------------------------------------------------------------------------------
// Column sums (bias gradient): out[c] (+)= sum_r x[r][c].  grid (C/64, row slabs); fp32 atomics into a zeroed output.
// ------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, int64_t ldx, float* __restrict__ out, int64_t R, int C) {
  __shared__ float part[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), ty = threadIdx.x >> 6;
  const int64_t per = (R + gridDim.y - 1) / gridDim.y, r0 = blockIdx.y * per, r1 = min(R, r0 + per);
  float s = 0.f;
  if (c < C)
    for (int64_t r = r0 + ty; r < r1; r += 4) s += (float)x[r * ldx + c];
  part[ty][threadIdx.x & 63] = s;
  __syncthreads();
  if (ty == 0 && c < C) atomicAdd(out + c, part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}

extern "C" int msam2_colsum(const void* x, int x_is_16bit, int64_t ldx, float* out, int64_t rows, int64_t cols, void* stream) {
  MSAM2_REQUIRE(x && out && rows > 0 && cols > 0, "colsum: bad arguments");
  const unsigned slabs = (unsigned)min((int64_t)256, cdiv(rows, 256));
  dim3 grid(cdiv(cols, 64), slabs);
  if (x_is_16bit) hipLaunchKernelGGL((colsum_kernel<op16>), grid, dim3(256), 0, (hipStream_t)stream, (const op16*)x, ldx, out, rows, (int)cols);
  else hipLaunchKernelGGL((colsum_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx, out, rows, (int)cols);
  return msam2_check_launch("colsum");
}

// ------------------------------------------------------------------------------------------------------------------
// dpre = dy * act'(pre) as a 16-bit GEMM operand.  act: 1 = GELU (exact erf: Phi(x) + x phi(x)), 2 = ReLU.  pre 16-bit or fp32.
// ------------------------------------------------------------------------------------------------------------------
template <typename TP, typename TD>
__global__ void act_bwd_kernel(const TP* __restrict__ pre, const TD* __restrict__ dy, op16* __restrict__ out, int64_t n, int act) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float x = (float)pre[i], g = (float)dy[i];
    float d;
    if (act == 1) {
      const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
      d = cdf + x * 0.3989422804014327f * __expf(-0.5f * x * x);
    } else {
      d = x > 0.f ? 1.f : 0.f;
    }
    out[i] = f2op(g * d);
  }
}

// eight elements per thread and trip (16-byte accesses for the 16-bit operands), Phi(x) from the polynomial of common.h (the GELU epilogue's:
// |error| <= 1e-5 on Phi) and one v_exp for the density -- the scalar kernel above spends ~40 VALU operations per element in erff() and
// moves 2 bytes per lane and load (51 us for the 25 M elements of a stage-3 fc1 map: 2.9 TB/s).
template <typename TP, typename TD>
__global__ __launch_bounds__(256) void act_bwd_vec_kernel(const TP* __restrict__ pre, const TD* __restrict__ dy, op16* __restrict__ out, int64_t n8,
                                                          int act) {
  typedef __attribute__((ext_vector_type(8))) float f32x8_t;
  auto load8 = [](const auto* ptr) {
    f32x8_t v;
    if constexpr (sizeof(*ptr) == 4) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(ptr), b = *reinterpret_cast<const f32x4*>(ptr + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
    } else {
      const op16x8 a = *reinterpret_cast<const op16x8*>(ptr);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = op2f(a[e]);
    }
    return v;
  };
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const f32x8_t x = load8(pre + i * 8), g = load8(dy + i * 8);
    op16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float d;
      if (act == 1) {
#ifndef MSAM2_GELU_AS
        const float xc = __builtin_amdgcn_fmed3f(x[e], -MSAM2_GELU_X, MSAM2_GELU_X), u = xc * xc;
        float q = __builtin_fmaf(u, MSAM2_GELU_Q8, MSAM2_GELU_Q7);
        q = __builtin_fmaf(q, u, MSAM2_GELU_Q6);
        q = __builtin_fmaf(q, u, MSAM2_GELU_Q5);
        q = __builtin_fmaf(q, u, MSAM2_GELU_Q4);
        q = __builtin_fmaf(q, u, MSAM2_GELU_Q3);
        q = __builtin_fmaf(q, u, MSAM2_GELU_Q2);
        q = __builtin_fmaf(q, u, MSAM2_GELU_Q1);
        q = __builtin_fmaf(q, u, MSAM2_GELU_Q0);
        const float cdf = __builtin_fmaf(xc, q, 0.5f);
#else
        const float cdf = 0.5f * (1.f + fast_erf(x[e] * 0.70710678118654752f));
#endif
        d = cdf + x[e] * 0.3989422804014327f * __builtin_amdgcn_exp2f(x[e] * x[e] * -0.72134752044448170368f);   // + x phi(x)
      } else {
        d = x[e] > 0.f ? 1.f : 0.f;
      }
      o[e] = f2op(g[e] * d);
    }
    *reinterpret_cast<op16x8*>(out + i * 8) = o;
  }
}

extern "C" int msam2_act_bwd(const void* pre, int pre_is_16bit, const void* dy, int dy_is_16bit, void* out, int64_t n, int act, void* stream) {
  MSAM2_REQUIRE(pre && dy && out && n > 0 && (act == 1 || act == 2), "act_bwd: bad arguments");
  dim3 grid((unsigned)min((int64_t)16384, cdiv(n, 256))), block(256);
  hipStream_t s = (hipStream_t)stream;
  static const bool v1 = getenv("MSAM2_ACT_BWD_V1") != nullptr;
  if (!v1 && n % 8 == 0 && ((uintptr_t)pre & 15) == 0 && ((uintptr_t)dy & 15) == 0 && ((uintptr_t)out & 15) == 0) {
    const int64_t n8 = n / 8;
    dim3 g8((unsigned)min((int64_t)8192, cdiv(n8, 256)));
    if (pre_is_16bit && dy_is_16bit) hipLaunchKernelGGL((act_bwd_vec_kernel<op16, op16>), g8, block, 0, s, (const op16*)pre, (const op16*)dy, (op16*)out, n8, act);
    else if (pre_is_16bit) hipLaunchKernelGGL((act_bwd_vec_kernel<op16, float>), g8, block, 0, s, (const op16*)pre, (const float*)dy, (op16*)out, n8, act);
    else if (dy_is_16bit) hipLaunchKernelGGL((act_bwd_vec_kernel<float, op16>), g8, block, 0, s, (const float*)pre, (const op16*)dy, (op16*)out, n8, act);
    else hipLaunchKernelGGL((act_bwd_vec_kernel<float, float>), g8, block, 0, s, (const float*)pre, (const float*)dy, (op16*)out, n8, act);
    return msam2_check_launch("act_bwd");
  }
  if (pre_is_16bit && dy_is_16bit) hipLaunchKernelGGL((act_bwd_kernel<op16, op16>), grid, block, 0, s, (const op16*)pre, (const op16*)dy, (op16*)out, n, act);
  else if (pre_is_16bit) hipLaunchKernelGGL((act_bwd_kernel<op16, float>), grid, block, 0, s, (const op16*)pre, (const float*)dy, (op16*)out, n, act);
  else if (dy_is_16bit) hipLaunchKernelGGL((act_bwd_kernel<float, op16>), grid, block, 0, s, (const float*)pre, (const op16*)dy, (op16*)out, n, act);
  else hipLaunchKernelGGL((act_bwd_kernel<float, float>), grid, block, 0, s, (const float*)pre, (const float*)dy, (op16*)out, n, act);
  return msam2_check_launch("act_bwd");
}

// ------------------------------------------------------------------------------------------------------------------
// LayerNorm backward, one wave per row (C <= 1024):  xhat = (x - mean) rstd,  g = dy * gamma,
//   dx = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat)) [+ add: the gradient of the residual path around the norm],
//   dgamma += dy * xhat,   dbeta += dy   (fp32 atomics, zeroed by the caller)
// The statistics are recomputed from x (fp32 residual stream), so the forward saves nothing.
// ------------------------------------------------------------------------------------------------------------------
template <typename TD, int NI>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, int64_t ldx, const TD* __restrict__ dy, int64_t ldd,
                                                            const float* __restrict__ gamma, float* __restrict__ dx, int64_t ldo,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int64_t rows, int C,
                                                            float eps, const float* __restrict__ add, int64_t lda) {
  // a lane always handles the same NI columns (lane + 64 i), so dgamma / dbeta accumulate in registers over the workgroup's rows;
  // the four waves are combined through LDS once at the end (the first version did two LDS atomics per element: 50 us for 16k x 256)
  __shared__ float sg[4][NI * 64], sb[4][NI * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t rows_per_block = (rows + gridDim.x - 1) / gridDim.x;
  const int64_t r_begin = blockIdx.x * rows_per_block, r_end = min(rows, r_begin + rows_per_block);
  float gam[NI], ag[NI], ab[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c = lane + 64 * i;
    gam[i] = c < C ? gamma[c] : 0.f;
    ag[i] = ab[i] = 0.f;
  }
  const float inv_c = 1.f / (float)C;
  for (int64_t row = r_begin + wave; row < r_end; row += 4) {
    const float* xr = x + row * ldx;
    const TD* dr = dy + row * ldd;
    float xv[NI], gv[NI], s = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = lane + 64 * i;
      xv[i] = c < C ? xr[c] : 0.f;
      gv[i] = c < C ? (float)dr[c] : 0.f;
      s += xv[i];
    }
    const float mean = wave_sum(s) * inv_c;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const float d = (lane + 64 * i) < C ? xv[i] - mean : 0.f;
      q += d * d;
    }
    const float rstd = 1.f / sqrtf(wave_sum(q) * inv_c + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const float xh = (lane + 64 * i) < C ? (xv[i] - mean) * rstd : 0.f, g = gv[i] * gam[i];
      s1 += g;
      s2 += g * xh;
      ag[i] += gv[i] * xh;
      ab[i] += gv[i];
      xv[i] = xh;
      gv[i] = g;
    }
    const float m1 = wave_sum(s1) * inv_c, m2 = wave_sum(s2) * inv_c;
    float* o = dx + row * ldo;
    const float* ar = add ? add + row * lda : nullptr;     // gradient of the residual branch that by-passes the LayerNorm
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = lane + 64 * i;
      if (c < C) o[c] = rstd * (gv[i] - m1 - xv[i] * m2) + (ar ? ar[c] : 0.f);
    }
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    sg[wave][lane + 64 * i] = ag[i];
    sb[wave][lane + 64 * i] = ab[i];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    atomicAdd(dgamma + c, sg[0][c] + sg[1][c] + sg[2][c] + sg[3][c]);
    atomicAdd(dbeta + c, sb[0][c] + sb[1][c] + sb[2][c] + sb[3][c]);
  }
}

// Vector form (round 3): 16 lanes per row, four rows per wave in flight, 16-byte loads / stores (a 16-lane group covers 256 contiguous
// bytes per instruction), ALL of a row's loads (x, dy, the by-pass gradient) issued before the first reduction -- the kernel above walks
// one row per wave with three dependent 64-lane reductions per row and 4-byte accesses (48.8 us for 16384 x 384: 2 TB/s; it was
// 5.3 % of the training iteration).  dgamma / dbeta stay in registers per lane (its own 4-column chunks), are folded over the four
// row groups of a wave by two shuffles, over the four waves through LDS, and leave as one atomic per column and workgroup.
// C % 4 == 0, 16-byte aligned rows; CHUNKS = ceil(C / 64).
template <typename TD, int CHUNKS>
__global__ __launch_bounds__(256) void layernorm_bwd_vec_kernel(const float* __restrict__ x, int64_t ldx, const TD* __restrict__ dy, int64_t ldd,
                                                                const float* __restrict__ gamma, float* __restrict__ dx, int64_t ldo,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta, int64_t rows, int C,
                                                                float eps, const float* __restrict__ add, int64_t lda) {
  __shared__ float sg[4][CHUNKS * 64], sb[4][CHUNKS * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l16 = threadIdx.x & 15, grp = threadIdx.x >> 4;   // 16 groups per workgroup
  const int nch = C >> 2;
  const int64_t rows_per_block = (rows + gridDim.x - 1) / gridDim.x;
  const int64_t r_begin = blockIdx.x * rows_per_block, r_end = min(rows, r_begin + rows_per_block);
  f32x4 gam[CHUNKS], ag[CHUNKS], ab[CHUNKS];
#pragma unroll
  for (int j = 0; j < CHUNKS; ++j) {
    const int ch = l16 + 16 * j;
    gam[j] = ch < nch ? *reinterpret_cast<const f32x4*>(gamma + ch * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    ag[j] = ab[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float inv_c = 1.f / (float)C;
  auto sum16 = [](float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
  };
  for (int64_t row = r_begin + grp; row < r_end; row += 16) {
    const float* xr = x + row * ldx;
    const TD* dr = dy + row * ldd;
    const float* ar = add ? add + row * lda : nullptr;     // gradient of the residual branch that by-passes the LayerNorm
    f32x4 xv[CHUNKS], gv[CHUNKS], av[CHUNKS];
#pragma unroll
    for (int j = 0; j < CHUNKS; ++j) {
      const int ch = l16 + 16 * j;
      xv[j] = gv[j] = av[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (ch < nch) {
        xv[j] = *reinterpret_cast<const f32x4*>(xr + ch * 4);
        if constexpr (sizeof(TD) == 4) {
          gv[j] = *reinterpret_cast<const f32x4*>(dr + ch * 4);
        } else {
          const op16x4 t = *reinterpret_cast<const op16x4*>(dr + ch * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) gv[j][e] = op2f(t[e]);
        }
        if (ar) av[j] = *reinterpret_cast<const f32x4*>(ar + ch * 4);
      }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < CHUNKS; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) s += xv[j][e];
    const float mean = sum16(s) * inv_c;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < CHUNKS; ++j)
      if (l16 + 16 * j < nch) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = xv[j][e] - mean;
          q += d * d;
        }
      }
    const float rstd = 1.f / sqrtf(sum16(q) * inv_c + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < CHUNKS; ++j) {
      const bool in = l16 + 16 * j < nch;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh = in ? (xv[j][e] - mean) * rstd : 0.f, g = gv[j][e] * gam[j][e];
        s1 += g;
        s2 += g * xh;
        ag[j][e] += gv[j][e] * xh;
        ab[j][e] += gv[j][e];
        xv[j][e] = xh;
        gv[j][e] = g;
      }
    }
    const float m1 = sum16(s1) * inv_c, m2 = sum16(s2) * inv_c;
    float* o = dx + row * ldo;
#pragma unroll
    for (int j = 0; j < CHUNKS; ++j) {
      const int ch = l16 + 16 * j;
      if (ch < nch) {
        f32x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = rstd * (gv[j][e] - m1 - xv[j][e] * m2) + av[j][e];
        *reinterpret_cast<f32x4*>(o + ch * 4) = w;
      }
    }
  }
  // fold the four row groups of a wave (lanes l16, l16 + 16, + 32, + 48 hold the same columns), then the four waves through LDS
#pragma unroll
  for (int j = 0; j < CHUNKS; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a = ag[j][e], b = ab[j][e];
      a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
      b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
      if (lane < 16) {
        sg[wave][(l16 + 16 * j) * 4 + e] = a;
        sb[wave][(l16 + 16 * j) * 4 + e] = b;
      }
    }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    atomicAdd(dgamma + c, sg[0][c] + sg[1][c] + sg[2][c] + sg[3][c]);
    atomicAdd(dbeta + c, sb[0][c] + sb[1][c] + sb[2][c] + sb[3][c]);
  }
}

template <typename TD>
static void layernorm_bwd_launch(const float* x, int64_t ldx, const TD* dy, int64_t ldd, const float* gamma, float* dx, int64_t ldo, float* dgamma,
                                 float* dbeta, int64_t rows, int C, float eps, const float* add, int64_t lda, hipStream_t s) {
  static const int64_t env_grid = getenv("MSAM2_LNB_GRID") ? atoll(getenv("MSAM2_LNB_GRID")) : 0;
  // every workgroup ends with one atomic per column on the SAME 2 C addresses: at 1024 workgroups that serialisation is the kernel
  // (16384 x 384: 45 us at 1024, 33 at 512, 30 at 256, 37 at 128; 262144 x 96: 90 / 85 / 107 / 187) -> 256 for the short maps, 512 beyond
  const dim3 grid((unsigned)min(env_grid > 0 ? env_grid : (int64_t)(rows <= 32768 ? 256 : 512), cdiv(rows, 16))), block(256);
  {
    static const bool v1 = getenv("MSAM2_LN_BWD_V1") != nullptr;
    const bool vec = !v1 && C % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && ldd % 4 == 0 && (!add || lda % 4 == 0) && ((uintptr_t)x & 15) == 0 &&
                     ((uintptr_t)dx & 15) == 0 && ((uintptr_t)dy & (sizeof(TD) == 4 ? 15 : 7)) == 0 && (!add || ((uintptr_t)add & 15) == 0) &&
                     ((uintptr_t)gamma & 15) == 0 && C <= 384;   // <= 6 chunks per lane: beyond that the six per-lane arrays cost the second wave per SIMD
    if (vec) {
      const int chunks = cdiv(C / 4, 16);
#define LNV(CH) hipLaunchKernelGGL((layernorm_bwd_vec_kernel<TD, CH>), grid, block, 0, s, x, ldx, dy, ldd, gamma, dx, ldo, dgamma, dbeta, rows, C, eps, add, lda)
      if (chunks <= 1) LNV(1);
      else if (chunks <= 2) LNV(2);
      else if (chunks <= 3) LNV(3);
      else if (chunks <= 4) LNV(4);
      else LNV(6);
#undef LNV
      return;
    }
  }
#define LNB(NI) hipLaunchKernelGGL((layernorm_bwd_kernel<TD, NI>), grid, block, 0, s, x, ldx, dy, ldd, gamma, dx, ldo, dgamma, dbeta, rows, C, eps, add, lda)
  const int ni = cdiv(C, 64);
  if (ni <= 1) LNB(1);
  else if (ni <= 2) LNB(2);
  else if (ni <= 4) LNB(4);
  else if (ni <= 6) LNB(6);
  else if (ni <= 8) LNB(8);
  else if (ni <= 12) LNB(12);
  else LNB(16);
#undef LNB
}

extern "C" int msam2_layernorm_bwd(const float* x, int64_t ldx, const void* dy, int dy_is_16bit, int64_t ldd, const float* gamma, float* dx,
                                   int64_t ldo, float* dgamma, float* dbeta, int64_t rows, int64_t C, float eps, const float* add, int64_t ld_add,
                                   void* stream) {
  MSAM2_REQUIRE(x && dy && gamma && dx && dgamma && dbeta, "layernorm_bwd: null tensor");
  MSAM2_REQUIRE(rows > 0 && C > 0 && C <= 1024, "layernorm_bwd: C <= 1024");
  hipStream_t s = (hipStream_t)stream;
  if (dy_is_16bit) layernorm_bwd_launch<op16>(x, ldx, (const op16*)dy, ldd, gamma, dx, ldo, dgamma, dbeta, rows, (int)C, eps, add, ld_add, s);
  else layernorm_bwd_launch<float>(x, ldx, (const float*)dy, ldd, gamma, dx, ldo, dgamma, dbeta, rows, (int)C, eps, add, ld_add, s);
  return msam2_check_launch("layernorm_bwd");
}

// ------------------------------------------------------------------------------------------------------------------
// Row softmax pieces of the MATERIALISED attention backward (backward.attention_backward): with 288 GB of HBM the [Lq, Lk] score
// matrix of one (batch, head) fits (268 MB fp32 at 4096 x 16384), so the five gradient products run on the forward GEMM kernel and
// only these two row kernels are specific.  One wave per row, any Lk.
//   softmax_rows:      P (16-bit) = softmax(scale * S) row-wise, S fp32
//   softmax_bwd_rows:  dS (16-bit) = scale * P * (dP - sum_k P dP) row-wise, P 16-bit, dP fp32
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, int64_t lds_, op16* __restrict__ p, int64_t ldp,
                                                           int64_t rows, int64_t cols, float scale_log2) {
  const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* sr = s + row * lds_;
  float m = -INFINITY;
  for (int64_t c = lane; c < cols; c += 64) m = fmaxf(m, sr[c]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  m *= scale_log2;
  float l = 0.f;
  for (int64_t c = lane; c < cols; c += 64) l += __builtin_amdgcn_exp2f(sr[c] * scale_log2 - m);
  l = wave_sum(l);
  const float inv = 1.f / l;
  op16* pr = p + row * ldp;
  for (int64_t c = lane; c < cols; c += 64) pr[c] = f2op(__builtin_amdgcn_exp2f(sr[c] * scale_log2 - m) * inv);
}

extern "C" int msam2_softmax_rows(const float* s, int64_t ld_s, void* p, int64_t ld_p, int64_t rows, int64_t cols, float scale, void* stream) {
  MSAM2_REQUIRE(s && p && rows > 0 && cols > 0 && scale > 0.f, "softmax_rows: bad arguments");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(cdiv(rows * 64, 256)), dim3(256), 0, (hipStream_t)stream, s, ld_s, (op16*)p, ld_p, rows, cols,
                     scale * 1.4426950408889634f);
  return msam2_check_launch("softmax_rows");
}

__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const op16* __restrict__ p, int64_t ldp, const float* __restrict__ dp, int64_t ldd,
                                                               op16* __restrict__ ds, int64_t ldo, int64_t rows, int64_t cols, float scale) {
  const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const op16* pr = p + row * ldp;
  const float* dr = dp + row * ldd;
  float acc = 0.f;
  for (int64_t c = lane; c < cols; c += 64) acc += op2f(pr[c]) * dr[c];
  acc = wave_sum(acc);
  op16* o = ds + row * ldo;
  for (int64_t c = lane; c < cols; c += 64) o[c] = f2op(scale * op2f(pr[c]) * (dr[c] - acc));
}

extern "C" int msam2_softmax_bwd_rows(const void* p, int64_t ld_p, const float* dp, int64_t ld_dp, void* ds, int64_t ld_ds, int64_t rows,
                                      int64_t cols, float scale, void* stream) {
  MSAM2_REQUIRE(p && dp && ds && rows > 0 && cols > 0, "softmax_bwd_rows: bad arguments");
  hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3(cdiv(rows * 64, 256)), dim3(256), 0, (hipStream_t)stream, (const op16*)p, ld_p, dp, ld_dp,
                     (op16*)ds, ld_ds, rows, cols, scale);
  return msam2_check_launch("softmax_bwd_rows");
}

// ------------------------------------------------------------------------------------------------------------------
// ConvTranspose2d(k2, s2) tail without the fused norm / activation (training forward + its adjoint; the inference kernel is
// pixel_shuffle_kernel in conv.hip):  z[pix, c] = g[tok(pix), sub(pix) * C + c] + bias[c] + skip[pix, c]   (fp32 out), and
// dg[tok, sub * C + c] = dz[pix(tok, sub), c]   (16-bit GEMM operand).  pix = (b, Y, X) on the 2h x 2w grid, tok = (b, Y/2, X/2),
// sub = (Y & 1) * 2 + (X & 1).
// ------------------------------------------------------------------------------------------------------------------
__global__ void convt2x2_gather_kernel(const op16* __restrict__ g, const float* __restrict__ bias, const op16* __restrict__ skip,
                                       float* __restrict__ z, int B, int h, int w, int C) {
  const int H = 2 * h, W = 2 * w;
  const int64_t total = (int64_t)B * H * W * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = i % C;
    const int64_t pix = i / C;
    const int X = pix % W, Y = (pix / W) % H;
    const int64_t b = pix / ((int64_t)W * H);
    const int64_t tok = (b * h + Y / 2) * w + X / 2;
    const int sub = (Y & 1) * 2 + (X & 1);
    z[i] = op2f(g[tok * 4 * C + sub * C + c]) + bias[c] + (skip ? op2f(skip[i]) : 0.f);
  }
}

extern "C" int msam2_convt2x2_gather(const void* gemm_out, const float* bias, const void* skip, float* z, int64_t B, int64_t h, int64_t w,
                                     int64_t C, void* stream) {
  MSAM2_REQUIRE(gemm_out && bias && z && B > 0 && h > 0 && w > 0 && C > 0, "convt2x2_gather: bad arguments");
  const int64_t total = B * 4 * h * w * C;
  hipLaunchKernelGGL(convt2x2_gather_kernel, dim3((unsigned)min((int64_t)16384, cdiv(total, 256))), dim3(256), 0, (hipStream_t)stream,
                     (const op16*)gemm_out, bias, (const op16*)skip, z, (int)B, (int)h, (int)w, (int)C);
  return msam2_check_launch("convt2x2_gather");
}

template <typename T>
__global__ void convt2x2_scatter_grad_kernel(const T* __restrict__ dz, op16* __restrict__ dg, int B, int h, int w, int C) {
  const int H = 2 * h, W = 2 * w;
  const int64_t total = (int64_t)B * H * W * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = i % C;
    const int64_t pix = i / C;
    const int X = pix % W, Y = (pix / W) % H;
    const int64_t b = pix / ((int64_t)W * H);
    const int64_t tok = (b * h + Y / 2) * w + X / 2;
    const int sub = (Y & 1) * 2 + (X & 1);
    dg[tok * 4 * C + sub * C + c] = f2op((float)dz[i]);
  }
}

extern "C" int msam2_convt2x2_scatter_grad(const void* dz, int dz_is_16bit, void* dg, int64_t B, int64_t h, int64_t w, int64_t C, void* stream) {
  MSAM2_REQUIRE(dz && dg && B > 0 && h > 0 && w > 0 && C > 0, "convt2x2_scatter_grad: bad arguments");
  const int64_t total = B * 4 * h * w * C;
  dim3 grid((unsigned)min((int64_t)16384, cdiv(total, 256))), block(256);
  if (dz_is_16bit) hipLaunchKernelGGL((convt2x2_scatter_grad_kernel<op16>), grid, block, 0, (hipStream_t)stream, (const op16*)dz, (op16*)dg, (int)B, (int)h, (int)w, (int)C);
  else hipLaunchKernelGGL((convt2x2_scatter_grad_kernel<float>), grid, block, 0, (hipStream_t)stream, (const float*)dz, (op16*)dg, (int)B, (int)h, (int)w, (int)C);
  return msam2_check_launch("convt2x2_scatter_grad");
}

// ------------------------------------------------------------------------------------------------------------------
// Memory-encoder backward pieces (memory_encoder.py:17-58 mask down-sampler, 62-117 CXBlock under torch.autograd):
//   dwconv7x7:        plain depthwise 7x7 (pad 3) on fp32 NHWC tokens, taps [49, C]; flip = 1 correlates with the flipped kernel,
//                     i.e. the input gradient of the forward convolution
//   dwconv7x7_wgrad:  dW[tap][c] += sum_pixels dY[p][c] * X[p + offset(tap)][c]   (zeroed fp32 [49, C] output)
//   col2im3x3s2:      adjoint of im2col3x3s2 (k3 s2 p1): every input pixel gathers its <= 4 (ky, kx) contributions from the column
//                     gradient [B*(H/2)*(W/2), ld] with columns ordered (ky, kx, c)
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dwconv7x7_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                        float* __restrict__ y, int B, int H, int W, int C4, int flip) {
  const int64_t total = (int64_t)B * H * W * C4;
  const int C = C4 * 4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C4) * 4;
    const int64_t pix = i / C4;
    const int px = (int)(pix % W), py = (int)((pix / W) % H);
    const int64_t b = pix / ((int64_t)W * H);
    f32x4 acc = bias ? *reinterpret_cast<const f32x4*>(bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ky = 0; ky < 7; ++ky) {
      const int yy = py + ky - 3;
      if (yy < 0 || yy >= H) continue;
      for (int kx = 0; kx < 7; ++kx) {
        const int xx = px + kx - 3;
        if (xx < 0 || xx >= W) continue;
        const int tap = flip ? (6 - ky) * 7 + (6 - kx) : ky * 7 + kx;
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + ((b * H + yy) * W + xx) * C + c);
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w + tap * C + c);
        acc += xv * wv;
      }
    }
    *reinterpret_cast<f32x4*>(y + pix * C + c) = acc;
  }
}

extern "C" int msam2_dwconv7x7(const float* x, const float* w_tap_major, const float* bias, float* y, int64_t B, int64_t H, int64_t W, int64_t C,
                               int flip, void* stream) {
  MSAM2_REQUIRE(x && w_tap_major && y && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "dwconv7x7: bad arguments");
  const int64_t total = B * H * W * (C / 4);
  hipLaunchKernelGGL(dwconv7x7_kernel, dim3((unsigned)min((int64_t)16384, cdiv(total, (int64_t)256))), dim3(256), 0, (hipStream_t)stream, x, w_tap_major,
                     bias, y, (int)B, (int)H, (int)W, (int)(C / 4), flip);
  return msam2_check_launch("dwconv7x7");
}

// grid (C / 64, pixel slabs); thread = channel, 49 accumulators in registers
__global__ __launch_bounds__(64) void dwconv7x7_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw, int B,
                                                             int H, int W, int C) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  const int64_t npix = (int64_t)B * H * W;
  const int64_t per = (npix + gridDim.y - 1) / gridDim.y, p0 = blockIdx.y * per, p1 = min(npix, p0 + per);
  float acc[49];
#pragma unroll
  for (int t = 0; t < 49; ++t) acc[t] = 0.f;
  for (int64_t pix = p0; pix < p1; ++pix) {
    const int px = (int)(pix % W), py = (int)((pix / W) % H);
    const int64_t b = pix / ((int64_t)W * H);
    const float g = dy[pix * C + c];
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
      const int yy = py + ky - 3;
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) {
        const int xx = px + kx - 3;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) acc[ky * 7 + kx] += g * x[((b * H + yy) * W + xx) * C + c];
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 49; ++t) atomicAdd(dw + t * C + c, acc[t]);
}

extern "C" int msam2_dwconv7x7_wgrad(const float* x, const float* dy, float* dw_tap_major, int64_t B, int64_t H, int64_t W, int64_t C,
                                     void* stream) {
  MSAM2_REQUIRE(x && dy && dw_tap_major && B > 0 && H > 0 && W > 0 && C > 0, "dwconv7x7_wgrad: bad arguments");
  const int64_t npix = B * H * W;
  dim3 grid((unsigned)cdiv(C, (int64_t)64), (unsigned)min((int64_t)512, cdiv(npix, (int64_t)32)));
  hipLaunchKernelGGL(dwconv7x7_wgrad_kernel, grid, dim3(64), 0, (hipStream_t)stream, x, dy, dw_tap_major, (int)B, (int)H, (int)W, (int)C);
  return msam2_check_launch("dwconv7x7_wgrad");
}

__global__ __launch_bounds__(256) void col2im3x3s2_kernel(const float* __restrict__ dcols, int64_t ld, float* __restrict__ dx, int B, int H, int W,
                                                          int C) {
  const int64_t total = (int64_t)B * H * W * C;
  const int Ho = H / 2, Wo = W / 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int64_t pix = i / C;
    const int px = (int)(pix % W), py = (int)((pix / W) % H);
    const int64_t b = pix / ((int64_t)W * H);
    float acc = 0.f;
    // output (oy, ox) reads input (2 oy + ky - 1, 2 ox + kx - 1)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int t = py + 1 - ky;
      if (t < 0 || (t & 1) || (t >> 1) >= Ho) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int u = px + 1 - kx;
        if (u < 0 || (u & 1) || (u >> 1) >= Wo) continue;
        acc += dcols[((b * Ho + (t >> 1)) * Wo + (u >> 1)) * ld + (ky * 3 + kx) * C + c];
      }
    }
    dx[i] = acc;
  }
}

extern "C" int msam2_col2im3x3s2(const float* dcols, int64_t ld, float* dx, int64_t B, int64_t H, int64_t W, int64_t C, void* stream) {
  MSAM2_REQUIRE(dcols && dx && B > 0 && H > 0 && W > 0 && C > 0 && H % 2 == 0 && W % 2 == 0 && ld >= 9 * C, "col2im3x3s2: bad arguments");
  const int64_t total = B * H * W * C;
  hipLaunchKernelGGL(col2im3x3s2_kernel, dim3((unsigned)min((int64_t)16384, cdiv(total, (int64_t)256))), dim3(256), 0, (hipStream_t)stream, dcols, ld, dx,
                     (int)B, (int)H, (int)W, (int)C);
  return msam2_check_launch("col2im3x3s2");
}

// ------------------------------------------------------------------------------------------------------------------
// Adjoint of msam2_bilinear_upsample (align_corners = False): the reference's training loss is taken on the mask logits up-sampled to
// the video resolution (sam2_video_predictor.py:724-744 `_get_orig_video_res_output` -> func_3d/function.py:137-170), so the loss
// gradient has to come back down.  Gather form: every low-res pixel visits the high-res pixels whose two source taps can include
// it and re-derives their weights exactly as the forward does (no atomics, run-to-run reproducible).
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int P, int h, int w, int H, int W) {
  const float sy = (float)h / H, sx = (float)w / W;
  const float ry = (float)H / h, rx = (float)W / w;
  const int64_t total = (int64_t)P * h * w;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int x = (int)(i % w);
    const int y = (int)((i / w) % h);
    const int64_t pl = i / ((int64_t)w * h);
    // outputs whose source coordinate lies in (y - 1, y + 1): Y in ((y - 0.5) r - 0.5, (y + 1.5) r - 0.5); one extra on each side
    // for the clamping at the borders
    const int Y0 = max(0, (int)floorf((y - 0.5f) * ry - 1.5f)), Y1 = min(H - 1, (int)ceilf((y + 1.5f) * ry + 0.5f));
    const int X0 = max(0, (int)floorf((x - 0.5f) * rx - 1.5f)), X1 = min(W - 1, (int)ceilf((x + 1.5f) * rx + 0.5f));
    const float* g = dy + pl * (int64_t)H * W;
    float acc = 0.f;
    for (int Y = Y0; Y <= Y1; ++Y) {
      const float fy = fmaxf((Y + 0.5f) * sy - 0.5f, 0.f);
      const int y0 = (int)fy, y1 = min(y0 + 1, h - 1);
      const float ly = fy - y0;
      const float wy = (y0 == y ? 1.f - ly : 0.f) + (y1 == y ? ly : 0.f);
      if (wy == 0.f) continue;
      float row = 0.f;
      for (int X = X0; X <= X1; ++X) {
        const float fx = fmaxf((X + 0.5f) * sx - 0.5f, 0.f);
        const int x0 = (int)fx, x1 = min(x0 + 1, w - 1);
        const float lx = fx - x0;
        const float wx = (x0 == x ? 1.f - lx : 0.f) + (x1 == x ? lx : 0.f);
        row += wx * g[(int64_t)Y * W + X];
      }
      acc += wy * row;
    }
    dx[i] = acc;
  }
}

extern "C" int msam2_bilinear_upsample_bwd(const float* dy, float* dx, int64_t planes, int64_t h, int64_t w, int64_t H, int64_t W, void* stream) {
  MSAM2_REQUIRE(dy && dx && planes > 0 && h > 0 && w > 0 && H >= h && W >= w, "bilinear_upsample_bwd: bad arguments (up-sampling only)");
  const int64_t total = planes * h * w;
  hipLaunchKernelGGL(bilinear_bwd_kernel, dim3((unsigned)min((int64_t)16384, (total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, dx,
                     (int)planes, (int)h, (int)w, (int)H, (int)W);
  return msam2_check_launch("bilinear_upsample_bwd");
}

// ------------------------------------------------------------------------------------------------------------------
// Loss and optimiser pieces of a decoder fine-tuning step (func_3d/function.py:69 `criterion_G = BCEWithLogitsLoss(pos_weight)`,
// train_3d.py:50 `optim.Adam(sam_layers, lr=1e-4, betas=(0.9, 0.999), eps=1e-8)`):
//   bce_logits:  loss = mean( pos_weight * y * softplus(-x) + (1 - y) * softplus(x) ),  dx = (pos_weight * y * (s - 1) + (1 - y) * s) / n,
//                s = sigmoid(x); the loss is accumulated into a zeroed fp32 scalar.
//   adam_step:   m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  p -= lr * (m / (1 - b1^t)) / (sqrt(v / (1 - b2^t)) + eps)
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bce_logits_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ dx,
                                                         float* __restrict__ loss, int64_t n, float pos_weight) {
  float acc = 0.f;
  const float inv_n = 1.f / (float)n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = x[i], t = y[i];
    const float sp_pos = fmaxf(v, 0.f) + log1pf(__expf(-fabsf(v)));   // softplus(v)
    const float sp_neg = sp_pos - v;                                   // softplus(-v)
    acc += pos_weight * t * sp_neg + (1.f - t) * sp_pos;
    const float s = 1.f / (1.f + __expf(-v));
    dx[i] = (pos_weight * t * (s - 1.f) + (1.f - t) * s) * inv_n;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) atomicAdd(loss, acc * inv_n);
}

extern "C" int msam2_bce_logits(const float* logits, const float* target, float* dlogits, float* loss, int64_t n, float pos_weight,
                                void* stream) {
  MSAM2_REQUIRE(logits && target && dlogits && loss && n > 0, "bce_logits: bad arguments");
  hipLaunchKernelGGL(bce_logits_kernel, dim3((unsigned)min((int64_t)1024, cdiv(n, 256))), dim3(256), 0, (hipStream_t)stream, logits, target,
                     dlogits, loss, n, pos_weight);
  return msam2_check_launch("bce_logits");
}

__global__ void adam_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
                                 float lr, float b1, float b2, float eps, float bc1, float bc2) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
  }
}

extern "C" int msam2_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                               float eps, int64_t step, void* stream) {
  MSAM2_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, "adam_step: bad arguments");
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)min((int64_t)4096, cdiv(n, 256))), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                     exp_avg_sq, n, lr, beta1, beta2, eps, bc1, bc2);
  return msam2_check_launch("adam_step");
}

// Multi-tensor form (gradients are multiplied by grad_scale first: the inverse of the loss scale that keeps 16-bit backward operands in
// range): up to ADAM_CHUNK parameters per launch, their pointers passed by value in the kernel argument block (so the
// launch is capturable and needs no device-side table); blockIdx.y = parameter, blockIdx.x strides over its elements.
constexpr int ADAM_CHUNK = 24;
struct AdamTable {
  float* p[ADAM_CHUNK];
  const float* g[ADAM_CHUNK];
  float* m[ADAM_CHUNK];
  float* v[ADAM_CHUNK];
  int64_t n[ADAM_CHUNK];
};

// `step` (device, optional): the optimiser's step count t >= 1 kept in device memory so that a captured hipGraph advances it on every
// replay (adam_tick_kernel increments it at the head of each step) -- the bias corrections 1 - beta^t are then those of
// torch.optim.Adam on every replay, not the ones of the step the graph was captured at.  Non-finite gradient entries leave their
// parameter and moments untouched (an fp16 overflow upstream must not poison the optimiser state).
__global__ void adam_tick_kernel(int* step) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *step += 1;
}

__global__ __launch_bounds__(256) void adam_multi_kernel(AdamTable tb, float lr, float b1, float b2, float eps, float bc1, float bc2,
                                                         float gscale, float decay, const int* __restrict__ step,
                                                         int* __restrict__ skipped) {
  const int t = blockIdx.y;
  float* __restrict__ p = tb.p[t];
  const float* __restrict__ g = tb.g[t];
  float* __restrict__ m = tb.m[t];
  float* __restrict__ v = tb.v[t];
  const int64_t n = tb.n[t];
  if (step) {
    const float ts = (float)*step;
    bc1 = 1.f - __powf(b1, ts);
    bc2 = 1.f - __powf(b2, ts);
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gi = g[i] * gscale;
    if (!isfinite(gi)) {
      if (skipped) atomicAdd(skipped, 1);                 // rare by construction; visible to the host as DecoderAdam.skipped_elements
      continue;
    }
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = p[i] * decay - lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);          // decay = 1 - lr * weight_decay (AdamW, decoupled)
  }
}

// step_counter (int32 on the device, may be null): when given, the library increments it once (a kernel, so that it is part of a
// captured graph) and the update reads t from it; the host `step` is then ignored.  skipped_counter (int32 on the device, may be null):
// incremented once per gradient ELEMENT that was non-finite and therefore left its parameter untouched -- the overflow of a 16-bit
// backward operand (a stale loss scale) is silent otherwise.
extern "C" int msam2_adam_step_multi(void* const* params, const void* const* grads, void* const* exp_avg, void* const* exp_avg_sq,
                                     const int64_t* numel, int64_t count, float lr, float beta1, float beta2, float eps, int64_t step,
                                     float grad_scale, float weight_decay, void* step_counter, void* skipped_counter, void* stream) {
  MSAM2_REQUIRE(params && grads && exp_avg && exp_avg_sq && numel && count > 0 && (step >= 1 || step_counter), "adam_step_multi: bad arguments");
  const float bc1 = 1.f - powf(beta1, (float)(step >= 1 ? step : 1)), bc2 = 1.f - powf(beta2, (float)(step >= 1 ? step : 1));
  if (step_counter) hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (int*)step_counter);
  for (int64_t c0 = 0; c0 < count; c0 += ADAM_CHUNK) {
    const int nt = (int)min((int64_t)ADAM_CHUNK, count - c0);
    AdamTable tb;
    int64_t nmax = 0;
    for (int i = 0; i < ADAM_CHUNK; ++i) {
      const int64_t j = c0 + min(i, nt - 1);
      MSAM2_REQUIRE(params[j] && grads[j] && exp_avg[j] && exp_avg_sq[j] && numel[j] > 0, "adam_step_multi: null / empty entry %lld", (long long)j);
      tb.p[i] = (float*)params[j]; tb.g[i] = (const float*)grads[j]; tb.m[i] = (float*)exp_avg[j]; tb.v[i] = (float*)exp_avg_sq[j];
      tb.n[i] = numel[j];
      nmax = max(nmax, numel[j]);
    }
    dim3 grid((unsigned)min((int64_t)128, cdiv(nmax, 256)), (unsigned)nt);
    hipLaunchKernelGGL(adam_multi_kernel, grid, dim3(256), 0, (hipStream_t)stream, tb, lr, beta1, beta2, eps, bc1, bc2, grad_scale,
                       1.f - lr * weight_decay, (const int*)step_counter, (int*)skipped_counter);
  }
  return msam2_check_launch("adam_step_multi");
}

// ------------------------------------------------------------------------------------------------------------------
// Backward of the two-way decoder's small-head attention (transformer.py:239-263; 8 heads of 16 / 32 channels) where ONE side is a
// handful of tokens (T <= 32) and the other the 4096 image tokens.  gridDim.y workgroups per (batch, head); the long side is spread over
// their threads, the T x T-free quantities of the short side live in LDS, and the short side's gradients are wave-reduced sums
// over the long side.  q/k/v 16-bit token-major [B, L, H*D] (element strides given), dO / dq / dk / dv fp32 token-major contiguous.
//   SMALL_Q: few queries, many keys  (tokens -> image);   !SMALL_Q: many queries, few keys  (image -> tokens)
// ------------------------------------------------------------------------------------------------------------------
template <int D, bool SMALL_Q>
__global__ __launch_bounds__(256) void attn_small_bwd_kernel(const op16* __restrict__ q, int64_t q_bs, int64_t q_ts, const op16* __restrict__ k,
                                                             int64_t k_bs, int64_t k_ts, const op16* __restrict__ v, int64_t v_bs, int64_t v_ts,
                                                             const float* __restrict__ d_o, float* __restrict__ dq, float* __restrict__ dk,
                                                             float* __restrict__ dv, int H, int Lq, int Lk, float scale) {
  constexpr int TMAX = 32;
  __shared__ float s_a[TMAX][D + 1];     // short side operand 1: SMALL_Q ? q (pre-scaled) : k
  __shared__ float s_b[TMAX][D + 1];     // short side operand 2: SMALL_Q ? dO          : v
  __shared__ float s_g1[TMAX][D + 1];    // short side gradient 1: SMALL_Q ? dq : dk
  __shared__ float s_g2[TMAX][D + 1];    // short side gradient 2: SMALL_Q ? -  : dv
  __shared__ float s_m[4][TMAX], s_l[4][TMAX], s_d[4][TMAX];
  const int b = blockIdx.x / H, head = blockIdx.x % H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int C = H * D;
  const int T = SMALL_Q ? Lq : Lk, L = SMALL_Q ? Lk : Lq;
  const float sl2 = scale * 1.4426950408889634f;
  for (int i = tid; i < T * D; i += 256) {
    const int t = i / D, d = i % D;
    if (SMALL_Q) {
      s_a[t][d] = op2f(q[b * q_bs + (int64_t)t * q_ts + head * D + d]);
      s_b[t][d] = d_o[((int64_t)b * Lq + t) * C + head * D + d];
    } else {
      s_a[t][d] = op2f(k[b * k_bs + (int64_t)t * k_ts + head * D + d]);
      s_b[t][d] = op2f(v[b * v_bs + (int64_t)t * v_ts + head * D + d]);
    }
    s_g1[t][d] = 0.f;
    s_g2[t][d] = 0.f;
  }
  __syncthreads();
  if constexpr (SMALL_Q) {
    // ---- pass 1: per-query softmax statistics AND delta_t = sum_j p_tj (dO_t . v_j) over all keys, online per thread (running
    //      max m, sum l, weighted sum a), merged over the workgroup.  Every workgroup of a (batch, head) repeats it (k / v of one
    //      head are 256 KB: L2 traffic), the expensive pass 2 below is split over gridDim.y.
    float m[TMAX], l[TMAX], a[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) { m[t] = -INFINITY; l[t] = 0.f; a[t] = 0.f; }
    for (int j = tid; j < L; j += 256) {
      float kv[D], vv[D];
#pragma unroll
      for (int d = 0; d < D; ++d) {
        kv[d] = op2f(k[b * k_bs + (int64_t)j * k_ts + head * D + d]);
        vv[d] = op2f(v[b * v_bs + (int64_t)j * v_ts + head * D + d]);
      }
#pragma unroll
      for (int t = 0; t < TMAX; ++t) {
        if (t >= T) break;
        float sc = 0.f, dp = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) { sc += s_a[t][d] * kv[d]; dp += s_b[t][d] * vv[d]; }
        sc *= sl2;
        const float mn = fmaxf(m[t], sc);
        const float c0 = __builtin_amdgcn_exp2f(m[t] - mn), c1 = __builtin_amdgcn_exp2f(sc - mn);
        l[t] = l[t] * c0 + c1;
        a[t] = a[t] * c0 + c1 * dp;
        m[t] = mn;
      }
    }
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {                       // (unrolled with an early exit: m[] / l[] / a[] stay in registers)
      if (t >= T) break;
      float mm = m[t], ll = l[t], aa = a[t];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(mm, o, 64), l2 = __shfl_xor(ll, o, 64), a2 = __shfl_xor(aa, o, 64);
        const float mn = fmaxf(mm, m2);
        const float c0 = mm == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mm - mn), c1 = m2 == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m2 - mn);
        ll = ll * c0 + l2 * c1;
        aa = aa * c0 + a2 * c1;
        mm = mn;
      }
      if (lane == 0) { s_m[wave][t] = mm; s_l[wave][t] = ll; s_d[wave][t] = aa; }
    }
    __syncthreads();
    if (tid < T) {                                         // final (max, 1 / sum, delta) of query tid -> row 0 of the stat arrays
      float mm = -INFINITY;
      for (int w = 0; w < 4; ++w) mm = fmaxf(mm, s_m[w][tid]);
      float ll = 0.f, aa = 0.f;
      for (int w = 0; w < 4; ++w) {
        const float c = s_m[w][tid] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(s_m[w][tid] - mm);
        ll += s_l[w][tid] * c;
        aa += s_d[w][tid] * c;
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      s_m[0][tid] = mm;
      s_l[0][tid] = 1.f / ll;
      s_d[0][tid] = aa / ll;
    }
    __syncthreads();
    // ---- pass 2: dV_j, dK_j per key (thread-local), dQ_t reduced over this workgroup's keys and added to the zeroed output
    for (int j0 = blockIdx.y * 256; j0 < L; j0 += gridDim.y * 256) {
      const int j = j0 + tid;
      const bool live = j < L;
      float kv[D], vv[D], gk[D], gv[D];
#pragma unroll
      for (int d = 0; d < D; ++d) {
        kv[d] = live ? op2f(k[b * k_bs + (int64_t)j * k_ts + head * D + d]) : 0.f;
        vv[d] = live ? op2f(v[b * v_bs + (int64_t)j * v_ts + head * D + d]) : 0.f;
        gk[d] = gv[d] = 0.f;
      }
      for (int t = 0; t < T; ++t) {
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) { s += s_a[t][d] * kv[d]; dp += s_b[t][d] * vv[d]; }
        const float p = live ? __builtin_amdgcn_exp2f(s * sl2 - s_m[0][t]) * s_l[0][t] : 0.f;
        const float ds = scale * p * (dp - s_d[0][t]);
#pragma unroll
        for (int d = 0; d < D; ++d) {
          gv[d] += p * s_b[t][d];
          gk[d] += ds * s_a[t][d];
          const float c = wave_sum(ds * kv[d]);
          if (lane == 0) atomicAdd(&s_g1[t][d], c);
        }
      }
      if (live) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
          dk[((int64_t)b * Lk + j) * C + head * D + d] = gk[d];
          dv[((int64_t)b * Lk + j) * C + head * D + d] = gv[d];
        }
      }
    }
    __syncthreads();
    for (int i = tid; i < T * D; i += 256) atomicAdd(dq + ((int64_t)b * Lq + i / D) * C + head * D + i % D, s_g1[i / D][i % D]);
  } else {
    // few keys: everything about a query is thread-local; dK_t / dV_t are sums over the queries (workgroup partials added to
    // the zeroed outputs)
    for (int i0 = blockIdx.y * 256; i0 < L; i0 += gridDim.y * 256) {
      const int i = i0 + tid;
      const bool live = i < L;
      float qv[D], dov[D], gq[D];
#pragma unroll
      for (int d = 0; d < D; ++d) {
        qv[d] = live ? op2f(q[b * q_bs + (int64_t)i * q_ts + head * D + d]) : 0.f;
        dov[d] = live ? d_o[((int64_t)b * Lq + i) * C + head * D + d] : 0.f;
        gq[d] = 0.f;
      }
      float s[TMAX], dp[TMAX], mx = -INFINITY;
#pragma unroll
      for (int t = 0; t < TMAX; ++t) {
        if (t >= T) break;
        float a = 0.f, c = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) { a += qv[d] * s_a[t][d]; c += dov[d] * s_b[t][d]; }
        s[t] = a * sl2;
        dp[t] = c;
        mx = fmaxf(mx, s[t]);
      }
      float lsum = 0.f, delta = 0.f;
#pragma unroll
      for (int t = 0; t < TMAX; ++t) {
        if (t >= T) break;
        s[t] = __builtin_amdgcn_exp2f(s[t] - mx);
        lsum += s[t];
      }
      const float inv = 1.f / lsum;
#pragma unroll
      for (int t = 0; t < TMAX; ++t) {
        if (t >= T) break;
        s[t] *= inv;                       // p_it
        delta += s[t] * dp[t];
      }
#pragma unroll
      for (int t = 0; t < TMAX; ++t) {
        if (t >= T) break;
        const float p = live ? s[t] : 0.f;
        const float ds = scale * p * (dp[t] - delta);
#pragma unroll
        for (int d = 0; d < D; ++d) {
          gq[d] += ds * s_a[t][d];
          const float c1 = wave_sum(ds * qv[d]), c2 = wave_sum(p * dov[d]);
          if (lane == 0) { atomicAdd(&s_g1[t][d], c1); atomicAdd(&s_g2[t][d], c2); }
        }
      }
      if (live) {
#pragma unroll
        for (int d = 0; d < D; ++d) dq[((int64_t)b * Lq + i) * C + head * D + d] = gq[d];
      }
    }
    __syncthreads();
    for (int i = tid; i < T * D; i += 256) {
      atomicAdd(dk + ((int64_t)b * Lk + i / D) * C + head * D + i % D, s_g1[i / D][i % D]);
      atomicAdd(dv + ((int64_t)b * Lk + i / D) * C + head * D + i % D, s_g2[i / D][i % D]);
    }
  }
}

__global__ __launch_bounds__(256) void zero2_kernel(float* __restrict__ a, float* __restrict__ b, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    a[i] = 0.f;
    if (b) b[i] = 0.f;
  }
}

extern "C" int msam2_attention_small_bwd(const void* q, int64_t q_bs, int64_t q_ts, const void* k, int64_t k_bs, int64_t k_ts, const void* v,
                                         int64_t v_bs, int64_t v_ts, const float* d_o, float* dq, float* dk, float* dv, int64_t B, int64_t H,
                                         int64_t Lq, int64_t Lk, int64_t D, float scale, void* stream) {
  MSAM2_REQUIRE(q && k && v && d_o && dq && dk && dv && B > 0 && H > 0 && Lq > 0 && Lk > 0, "attention_small_bwd: bad arguments");
  MSAM2_REQUIRE(D == 16 || D == 32, "attention_small_bwd: head dim 16 / 32");
  MSAM2_REQUIRE(Lq <= 32 || Lk <= 32, "attention_small_bwd: one side must have at most 32 tokens");
  const bool small_q = Lq <= 32 && Lq <= Lk;
  const int64_t L = small_q ? Lk : Lq;
  // the long side is split over gridDim.y workgroups (~2 per CU over the launch); the short side's gradients are accumulated with
  // atomics into zeroed outputs
  const int64_t gy = max((int64_t)1, min(cdiv(L, 256), cdiv(512, B * H)));
  dim3 grid((unsigned)(B * H), (unsigned)gy), block(256);
  hipStream_t s = (hipStream_t)stream;
  // (zero fill by kernel, not hipMemsetAsync: see gemm_zero_kernel in gemm.hip)
  const int64_t short_n = B * (small_q ? Lq : Lk) * H * D;
  const dim3 zgrid((unsigned)min((int64_t)1024, cdiv(short_n, (int64_t)256)));
  if (small_q) hipLaunchKernelGGL(zero2_kernel, zgrid, dim3(256), 0, s, dq, (float*)nullptr, short_n);
  else hipLaunchKernelGGL(zero2_kernel, zgrid, dim3(256), 0, s, dk, dv, short_n);
#define ASB(DD, SQ)                                                                                                                      \
  hipLaunchKernelGGL((attn_small_bwd_kernel<DD, SQ>), grid, block, 0, s, (const op16*)q, q_bs, q_ts, (const op16*)k, k_bs, k_ts, (const op16*)v, \
                     v_bs, v_ts, d_o, dq, dk, dv, (int)H, (int)Lq, (int)Lk, scale)
  if (D == 16) { if (small_q) ASB(16, true); else ASB(16, false); }
  else { if (small_q) ASB(32, true); else ASB(32, false); }
#undef ASB
  return msam2_check_launch("attention_small_bwd");
}

// ------------------------------------------------------------------------------------------------------------------
// Adjoints of the Hiera trunk / FPN neck data-movement ops (the 2-D training loop differentiates the image encoder:
// func_2d/function.py:70-72 under grad, 246-259).
// ------------------------------------------------------------------------------------------------------------------
// MaxPool2d(2, 2) backward on token-major [B*H*W, C] maps (do_pool, hieradet.py:23-34): dy goes to the FIRST maximum of each 2x2
// window in scan order (what torch's max_pool2d keeps: a later element must be strictly greater), zeros elsewhere; every dx element
// is written.
template <typename TI>
__global__ void maxpool2x2_bwd_kernel(const TI* __restrict__ x, int64_t ldx, const float* __restrict__ dy, int64_t lddy, float* __restrict__ dx,
                                      int64_t lddx, int B, int H, int W, int C) {
  const int Ho = H / 2, Wo = W / 2;
  const int64_t total = (int64_t)B * Ho * Wo * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = i % C;
    int64_t t = i / C;
    const int xo = t % Wo;
    t /= Wo;
    const int yo = t % Ho;
    const int b = t / Ho;
    const int64_t base = ((int64_t)b * H + 2 * yo) * W + 2 * xo;
    const int64_t pos[4] = {base, base + 1, base + W, base + W + 1};
    int arg = 0;
    float best = (float)x[pos[0] * ldx + c];
#pragma unroll
    for (int j = 1; j < 4; ++j) {
      const float v = (float)x[pos[j] * ldx + c];
      if (v > best || (v != v && best == best)) { best = v; arg = j; }
    }
    const float g = dy[(((int64_t)b * Ho + yo) * Wo + xo) * lddy + c];
#pragma unroll
    for (int j = 0; j < 4; ++j) dx[pos[j] * lddx + c] = (j == arg) ? g : 0.f;
  }
}

extern "C" int msam2_maxpool2x2_bwd(const void* x, int x_is_16bit, int64_t ldx, const float* dy, int64_t lddy, float* dx, int64_t lddx,
                                    int64_t B, int64_t H, int64_t W, int64_t C, void* stream) {
  MSAM2_REQUIRE(x && dy && dx && B > 0 && C > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "maxpool2x2_bwd: bad arguments");
  const int64_t total = B * (H / 2) * (W / 2) * C;
  dim3 grid((unsigned)min((int64_t)8192, (total + 255) / 256)), block(256);
  if (x_is_16bit) hipLaunchKernelGGL((maxpool2x2_bwd_kernel<op16>), grid, block, 0, (hipStream_t)stream, (const op16*)x, ldx, dy, lddy, dx, lddx, (int)B, (int)H, (int)W, (int)C);
  else hipLaunchKernelGGL((maxpool2x2_bwd_kernel<float>), grid, block, 0, (hipStream_t)stream, (const float*)x, ldx, dy, lddy, dx, lddx, (int)B, (int)H, (int)W, (int)C);
  return msam2_check_launch("maxpool2x2_bwd");
}

// ------------------------------------------------------------------------------------------------------------------
// window_partition / window_unpartition (backbones/utils.py:16-62) as ONE data-movement kernel each way, in 16-byte chunks, for the
// attention BACKWARD of the Hiera trunk: its flash kernels take [batch, head, token, D] operands with one token stride, which a window
// of an un-partitioned token image does not have.  The training iteration used to build the seven window tensors of a block (q, k, v,
// dO in; dQ, dK, dV out) with torch's generic strided copies plus a torch.cat: 14 % of the whole iteration
// (profiles/r02_train_iteration_kernel_stats.csv).  (The forward never partitions: its window kernel gathers.)
//   partition:   img [B, H, W, heads * D] (row stride ld_img elements, any column offset folded into the pointer)
//                -> win [B * nWy * nWx, heads, ws * ws, D] contiguous; bottom / right padding rows take `fill` [heads * D] (same type) or 0
//   unpartition: the reverse, cropping the padding (every [B, H, W] row of img is written)
// es = element size in bytes (2 or 4); D * es must be a multiple of 16.
// ------------------------------------------------------------------------------------------------------------------
// IDX: the chunk index type -- unsigned (every volume of the training iteration: < 2^32 chunks) keeps the six divisions by run-time values
// 32-bit; with int64_t each costs ~5x as many instructions and the kernel was bound by them (12.8-16.8 us for 12.6 MB moved).
template <bool TO_WINDOWS, typename IDX>
__global__ void window_move_kernel(unsigned char* __restrict__ img, int64_t ld_img_b, unsigned char* __restrict__ win,
                                   const unsigned char* __restrict__ fill, int B, int H, int W, int heads, int cpd /* 16-byte chunks per D */,
                                   int ws, int nwy, int nwx) {
  const int L = ws * ws;
  const IDX total = TO_WINDOWS ? (IDX)B * nwy * nwx * heads * L * cpd : (IDX)B * H * W * heads * cpd;
  for (IDX i = (IDX)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (IDX)gridDim.x * blockDim.x) {
    int c, head, y, x, b;
    int64_t widx;
    if (TO_WINDOWS) {
      c = (int)(i % cpd);
      IDX t = i / cpd;
      const int tok = (int)(t % L);
      t /= L;
      head = (int)(t % heads);
      t /= heads;
      const int wx = (int)(t % nwx);
      t /= nwx;
      const int wy = (int)(t % nwy);
      b = (int)(t / nwy);
      y = wy * ws + tok / ws;
      x = wx * ws + tok % ws;
      widx = (int64_t)i;
    } else {
      c = (int)(i % cpd);
      IDX t = i / cpd;
      head = (int)(t % heads);
      t /= heads;
      x = (int)(t % W);
      t /= W;
      y = (int)(t % H);
      b = (int)(t / H);
      const int wy = y / ws, wx = x / ws, tok = (y % ws) * ws + (x % ws);
      widx = ((((int64_t)(b * nwy + wy) * nwx + wx) * heads + head) * L + tok) * cpd + c;
    }
    uint4* wp = reinterpret_cast<uint4*>(win) + widx;
    if (TO_WINDOWS) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (y < H && x < W) v = *reinterpret_cast<const uint4*>(img + (((int64_t)b * H + y) * W + x) * ld_img_b + ((int64_t)head * cpd + c) * 16);
      else if (fill) v = *reinterpret_cast<const uint4*>(fill + ((int64_t)head * cpd + c) * 16);
      *wp = v;
    } else {
      *reinterpret_cast<uint4*>(img + (((int64_t)b * H + y) * W + x) * ld_img_b + ((int64_t)head * cpd + c) * 16) = *wp;
    }
  }
}

extern "C" int msam2_window_move(void* img, int64_t ld_img, void* win, const void* fill, int64_t B, int64_t H, int64_t W, int64_t heads, int64_t D,
                                 int64_t ws, int elem_bytes, int to_windows, void* stream) {
  MSAM2_REQUIRE(img && win && B > 0 && H > 0 && W > 0 && heads > 0 && D > 0 && ws > 0, "window_move: bad arguments");
  MSAM2_REQUIRE((elem_bytes == 2 || elem_bytes == 4) && (D * elem_bytes) % 16 == 0 && (ld_img * elem_bytes) % 16 == 0 &&
                    (((uintptr_t)img | (uintptr_t)win | (uintptr_t)fill) & 15) == 0,
                "window_move: 16-byte chunks (D * element size, row stride and pointers must be multiples of 16 bytes)");
  const int nwy = (int)((H + ws - 1) / ws), nwx = (int)((W + ws - 1) / ws), cpd = (int)(D * elem_bytes / 16);
  const int64_t total = to_windows ? B * nwy * nwx * heads * ws * ws * cpd : B * H * W * heads * cpd;
  dim3 grid((unsigned)min((int64_t)16384, (total + 255) / 256)), block(256);
#define WMOVE(TW, IDX)                                                                                                                        \
  hipLaunchKernelGGL((window_move_kernel<TW, IDX>), grid, block, 0, (hipStream_t)stream, (unsigned char*)img, ld_img * elem_bytes, (unsigned char*)win, \
                     (const unsigned char*)fill, (int)B, (int)H, (int)W, (int)heads, cpd, (int)ws, nwy, nwx)
  const bool small = total + 16384ll * 256 < (1ll << 32);                 // the grid-stride increment must not wrap either
  if (to_windows) { if (small) WMOVE(true, unsigned); else WMOVE(true, int64_t); }
  else { if (small) WMOVE(false, unsigned); else WMOVE(false, int64_t); }
#undef WMOVE
  return msam2_check_launch("window_move");
}

// window_unpartition of fp32 windows into a 16-BIT token image (the gradients dq / dk / dv of the trunk's windowed attention leave the flash
// backward in fp32 window order and are consumed as the 16-bit operand of the fused-qkv weight / input gradient GEMMs): one pass instead
// of an fp32 un-partition (4 B read + 4 B written per element) followed by a cast (4 read + 2 written).  8 elements per thread.
__global__ void window_unpartition_cvt_kernel(op16* __restrict__ img, int64_t ld_img, const float* __restrict__ win, int B, int H, int W, int heads,
                                              int c8pd /* 8-element chunks per D */, int ws, int nwy, int nwx) {
  const int L = ws * ws;
  const unsigned total = (unsigned)B * H * W * heads * c8pd;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int c = (int)(i % c8pd);
    unsigned t = i / c8pd;
    const int head = (int)(t % heads);
    t /= heads;
    const int x = (int)(t % W);
    t /= W;
    const int y = (int)(t % H), b = (int)(t / H);
    const int wy = y / ws, wx = x / ws, tok = (y % ws) * ws + (x % ws);
    const float* src = win + (((((int64_t)(b * nwy + wy) * nwx + wx) * heads + head) * L + tok) * c8pd + c) * 8;
    const f32x4 a = *reinterpret_cast<const f32x4*>(src), bb = *reinterpret_cast<const f32x4*>(src + 4);
    op16x8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[e] = f2op(a[e]);
      o[4 + e] = f2op(bb[e]);
    }
    *reinterpret_cast<op16x8*>(img + (((int64_t)b * H + y) * W + x) * ld_img + ((int64_t)head * c8pd + c) * 8) = o;
  }
}

extern "C" int msam2_window_unpartition_cvt(void* img16, int64_t ld_img, const float* win, int64_t B, int64_t H, int64_t W, int64_t heads, int64_t D,
                                            int64_t ws, void* stream) {
  MSAM2_REQUIRE(img16 && win && B > 0 && H > 0 && W > 0 && heads > 0 && D > 0 && ws > 0, "window_unpartition_cvt: bad arguments");
  MSAM2_REQUIRE(D % 8 == 0 && ld_img % 8 == 0 && (((uintptr_t)img16 | (uintptr_t)win) & 15) == 0,
                "window_unpartition_cvt: D and the row stride must be multiples of 8 elements, pointers 16-byte aligned");
  const int64_t total = B * H * W * heads * (D / 8);
  MSAM2_REQUIRE(total + 16384ll * 256 < (1ll << 32), "window_unpartition_cvt: volume beyond 32-bit chunk indices");
  const int nwy = (int)((H + ws - 1) / ws), nwx = (int)((W + ws - 1) / ws);
  hipLaunchKernelGGL(window_unpartition_cvt_kernel, dim3((unsigned)min((int64_t)16384, (total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (op16*)img16, ld_img, win, (int)B, (int)H, (int)W, (int)heads, (int)(D / 8), (int)ws, nwy, nwx);
  return msam2_check_launch("window_unpartition_cvt");
}

// out[head * D + d] += sum of win[w][head][tok][d] over the window tokens that lie OUTSIDE the [H, W] image (the zero-padded tokens of
// window_partition): their dk / dv belong to the qkv bias (the LayerNorm'ed map is padded before the qkv Linear, hieradet.py:143-150).
// DETERMINISTIC (round 4; the first form added one fp32 atomic per window and column, so the qkv-bias gradient -- and through Adam the
// whole encoder -- changed in the last bits from run to run): a workgroup owns 16 columns, its 64 thread groups walk the (window, token
// row) pairs in a fixed round-robin, and the 64 partial sums are added in index order by one thread per column.  `out` is accumulated
// into (single writer per column), as before.
__global__ __launch_bounds__(1024) void window_pad_colsum_kernel(const float* __restrict__ win, float* __restrict__ out, int H, int W, int heads, int D,
                                                                 int ws, int nwy, int nwx, int n_win) {
  __shared__ float red[64][17];
  const int cl = threadIdx.x & 15, sub = threadIdx.x >> 4, width = heads * D;
  const int c = blockIdx.x * 16 + cl;
  float acc = 0.f;
  if (c < width) {
    const int head = c / D, d = c - head * D, L = ws * ws;
    for (int item = sub; item < n_win * ws; item += 64) {
      const int w = item / ws, ty = item - w * ws, wx = w % nwx, wy = (w / nwx) % nwy;
      const int y_in = min(ws, H - wy * ws), x_in = min(ws, W - wx * ws);     // tokens (ty < y_in && tx < x_in) are inside the image
      const int tx0 = ty < y_in ? x_in : 0;
      const float* base = win + (((int64_t)w * heads + head) * L + ty * ws) * D + d;
      for (int tx = tx0; tx < ws; ++tx) acc += base[(int64_t)tx * D];
    }
  }
  red[sub][cl] = acc;
  __syncthreads();
  if (sub == 0 && c < width) {
    float s = 0.f;
    for (int i = 0; i < 64; ++i) s += red[i][cl];
    out[c] += s;
  }
}

extern "C" int msam2_window_pad_colsum(const float* win, float* out, int64_t B, int64_t H, int64_t W, int64_t heads, int64_t D, int64_t ws,
                                       void* stream) {
  MSAM2_REQUIRE(win && out && B > 0 && H > 0 && W > 0 && heads > 0 && D > 0 && ws > 0, "window_pad_colsum: bad arguments");
  const int nwy = (int)((H + ws - 1) / ws), nwx = (int)((W + ws - 1) / ws);
  MSAM2_REQUIRE(B * nwy * nwx * ws < (1ll << 31), "window_pad_colsum: window count beyond 32-bit indices");
  if (H % ws == 0 && W % ws == 0) return 0;                                   // no padded token anywhere
  hipLaunchKernelGGL(window_pad_colsum_kernel, dim3((unsigned)((heads * D + 15) / 16)), dim3(1024), 0, (hipStream_t)stream, win, out, (int)H, (int)W,
                     (int)heads, (int)D, (int)ws, nwy, nwx, (int)(B * nwy * nwx));
  return msam2_check_launch("window_pad_colsum");
}

// adjoint of the FPN's nearest-2x top-down step (msam2_upsample2x_add, image_encoder.py:113-124): out[b,i,j,c] = sum of the 2x2 block
__global__ void sumpool2x2_kernel(const float* __restrict__ dy, float* __restrict__ out, int B, int H, int W, int C) {
  const int Ho = H / 2, Wo = W / 2;
  const int64_t total = (int64_t)B * Ho * Wo * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = i % C;
    int64_t t = i / C;
    const int xo = t % Wo;
    t /= Wo;
    const int yo = t % Ho;
    const int b = t / Ho;
    const int64_t base = (((int64_t)b * H + 2 * yo) * W + 2 * xo) * C + c;
    out[i] = dy[base] + dy[base + C] + dy[base + (int64_t)W * C] + dy[base + (int64_t)W * C + C];
  }
}

extern "C" int msam2_sumpool2x2(const float* dy, float* out, int64_t B, int64_t H, int64_t W, int64_t C, void* stream) {
  MSAM2_REQUIRE(dy && out && B > 0 && C > 0 && H % 2 == 0 && W % 2 == 0, "sumpool2x2: bad arguments");
  const int64_t total = B * (H / 2) * (W / 2) * C;
  hipLaunchKernelGGL(sumpool2x2_kernel, dim3((unsigned)min((int64_t)8192, (total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, out,
                     (int)B, (int)H, (int)W, (int)C);
  return msam2_check_launch("sumpool2x2");
}

// Adjoint of msam2_hiera_pos_embed (Hiera._get_pos_embed, hieradet.py:269-277): d_table fp32 [h*w, C] (the gradient of the position
// tokens, already summed over the batch) -> d pos_embed [C, bh, bw] through the transposed bicubic resize (same tap weights and border
// clamping as the forward: the resize is separable, so the weight of source row y for output row yy is a 1-D table) and
// d pos_embed_window [C, wsz, wsz] (sum over the tiling).
__device__ __forceinline__ float cubic1_b(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2_b(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

// Two passes, both gather form (no atomics: the gradients are run-to-run reproducible):
//   rows pass   one workgroup per output row yy, one thread per channel: the row's w x C gradients are read once (coalesced over the
//               channels) and reduced over xx into bw column sums weighted by the bicubic column weights of each source column plus
//               wsz plain sums per window column -> workspace [h][bw + wsz][C];
//   final pass  one workgroup per source pixel / window cell: the weighted sum over the 256 rows of the workspace.
// (The first version looped over the whole 256 x 256 table inside 113 workgroups of 96 threads: 7.2 ms per call, 11 % of a training
// iteration; this form reads the table once: ~25 MB.)
__device__ __forceinline__ float pos_bwd_weight(int o, int n_out, int n_src, int src) {
  // weight with which source index `src` enters output index `o` of the clamped 4-tap bicubic resize
  const float A = -0.75f;
  const float f = (o + 0.5f) * ((float)n_src / n_out) - 0.5f;
  const int i0 = (int)floorf(f);
  const float t = f - i0;
  const float tap[4] = {cubic2_b(t + 1.f, A), cubic1_b(t, A), cubic1_b(1.f - t, A), cubic2_b(2.f - t, A)};
  float acc = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a)
    if (min(max(i0 - 1 + a, 0), n_src - 1) == src) acc += tap[a];
  return acc;
}

template <int MAXB, int MAXW>
__global__ void hiera_pos_bwd_rows_kernel(const float* __restrict__ d_table, float* __restrict__ part, int C, int bw, int w, int wsz) {
  extern __shared__ float wx[];                              // [bw][w] column weights
  for (int i = threadIdx.x; i < bw * w; i += blockDim.x) wx[i] = pos_bwd_weight(i % w, w, bw, i / w);
  __syncthreads();
  const int yy = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float ab[MAXB], aw[MAXW];
#pragma unroll
    for (int j = 0; j < MAXB; ++j) ab[j] = 0.f;
#pragma unroll
    for (int j = 0; j < MAXW; ++j) aw[j] = 0.f;
    const float* row = d_table + (int64_t)yy * w * C + c;
    for (int x0 = 0; x0 < w; x0 += MAXW) {
#pragma unroll
      for (int j = 0; j < MAXW; ++j) {
        const int xx = x0 + j;                               // w % wsz == 0 and wsz <= MAXW: column xx belongs to window column xx % wsz
        if (xx < w) {
          const float v = row[(int64_t)xx * C];
          aw[xx % wsz] += v;
#pragma unroll
          for (int sx = 0; sx < MAXB; ++sx)
            if (sx < bw) ab[sx] += wx[sx * w + xx] * v;
        }
      }
    }
    float* o = part + (int64_t)yy * (bw + wsz) * C + c;
    for (int sx = 0; sx < bw; ++sx) o[(int64_t)sx * C] = ab[sx];
    for (int x = 0; x < wsz; ++x) o[(int64_t)(bw + x) * C] = aw[x];
  }
}

__global__ void hiera_pos_bwd_final_kernel(const float* __restrict__ part, float* __restrict__ d_bkg, float* __restrict__ d_win, int C, int bh,
                                           int bw, int h, int wsz) {
  extern __shared__ float wy[];                              // [h] row weights of this source row
  const int nb = bh * bw, slots = bw + wsz;
  if ((int)blockIdx.x < nb) {
    const int sy = blockIdx.x / bw, sx = blockIdx.x % bw;
    for (int i = threadIdx.x; i < h; i += blockDim.x) wy[i] = pos_bwd_weight(i, h, bh, sy);
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      float acc = 0.f;
      for (int yy = 0; yy < h; ++yy) acc += wy[yy] * part[((int64_t)yy * slots + sx) * C + c];
      d_bkg[((int64_t)c * bh + sy) * bw + sx] = acc;
    }
  } else {
    const int cell = blockIdx.x - nb, y = cell / wsz, x = cell % wsz;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      float acc = 0.f;
      for (int yy = y; yy < h; yy += wsz) acc += part[((int64_t)yy * slots + bw + x) * C + c];
      d_win[((int64_t)c * wsz + y) * wsz + x] = acc;
    }
  }
}

extern "C" size_t msam2_hiera_pos_embed_bwd_workspace_bytes(int64_t C, int64_t bw, int64_t h, int64_t window) {
  return (size_t)(h * (bw + window) * C) * sizeof(float);
}

extern "C" int msam2_hiera_pos_embed_bwd(const float* d_table, float* d_pos_embed, float* d_pos_embed_window, int64_t C, int64_t bh,
                                         int64_t bw, int64_t h, int64_t w, int64_t window, void* workspace, size_t workspace_bytes,
                                         void* stream) {
  MSAM2_REQUIRE(d_table && d_pos_embed && d_pos_embed_window && C > 0 && h % window == 0 && w % window == 0, "hiera_pos_embed_bwd: bad arguments");
  MSAM2_REQUIRE(bw <= 16 && window <= 8, "hiera_pos_embed_bwd: built for pos_embed up to 16 columns and windows up to 8 (7x7 / 8x8 in hiera_t / s, 14x14 / 8x8 in hiera_b+)");
  MSAM2_REQUIRE(workspace && workspace_bytes >= msam2_hiera_pos_embed_bwd_workspace_bytes(C, bw, h, window), "hiera_pos_embed_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  float* part = (float*)workspace;
  if (bw <= 8)
    hipLaunchKernelGGL((hiera_pos_bwd_rows_kernel<8, 8>), dim3((unsigned)h), dim3(128), (size_t)(bw * w) * sizeof(float), s, d_table, part, (int)C,
                       (int)bw, (int)w, (int)window);
  else
    hipLaunchKernelGGL((hiera_pos_bwd_rows_kernel<16, 8>), dim3((unsigned)h), dim3(128), (size_t)(bw * w) * sizeof(float), s, d_table, part, (int)C,
                       (int)bw, (int)w, (int)window);
  hipLaunchKernelGGL(hiera_pos_bwd_final_kernel, dim3((unsigned)(bh * bw + window * window)), dim3(128), (size_t)h * sizeof(float), s, part,
                     d_pos_embed, d_pos_embed_window, (int)C, (int)bh, (int)bw, (int)h, (int)window);
  return msam2_check_launch("hiera_pos_embed_bwd");
}

// ------------------------------------------------------------------------------------------------------------------
// Train-mode dropout (memory_attention.py:40-48,63,80,97-98: nn.Dropout(0.1) on the three residual branches and inside the FFN;
// transformer.py:317-318: dropout_p on the attention probabilities) with a COUNTER-BASED generator: element i of stream
// (seed, offset) is kept iff mix64(seed + (offset + i) * golden) >> 32 >= p * 2^32, so the backward re-creates the mask of the
// forward from the same (seed, offset) -- nothing is stored.  y = keep ? x / (1 - p) : 0  (+ residual).  The same call on a gradient
// is the backward.  (Not torch's Philox stream: masks match in distribution, not bit for bit -- parity is pinned with the oracle
// consuming the masks this kernel produces.)
// The stream id may come from the DEVICE (seed_dev, optional uint64: added to `seed` by every thread): a training step captured into a
// hipGraph bakes its by-value arguments in, so the per-forward sub-stream counter has to live in device memory and be advanced by a
// kernel of the step itself (msam2_counter_bump) for a replay to draw fresh masks.
// ------------------------------------------------------------------------------------------------------------------
template <typename TI, typename TO>
__global__ void dropout_kernel(const TI* __restrict__ x, int64_t ldx, const float* __restrict__ res, int64_t ldr, TO* __restrict__ y, int64_t ldy,
                               int64_t rows, int64_t cols, unsigned thr, float inv_keep, uint64_t seed, uint64_t offset,
                               const uint64_t* __restrict__ seed_dev) {
  if (seed_dev) seed += *seed_dev;
  const int64_t total = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols, c = i - r * cols;
    float v = dropout_keep(seed, offset + (uint64_t)i, thr) ? (float)x[r * ldx + c] * inv_keep : 0.f;
    if (res) v += res[r * ldr + c];
    y[r * ldy + c] = (TO)v;
  }
}

// x / y: [rows, cols] with row strides (elements); residual (optional) fp32.  The element index of the stream is r * cols + c.
extern "C" int msam2_dropout(const void* x, int x_is_16bit, int64_t ldx, const float* residual, int64_t ldr, void* y, int y_is_16bit, int64_t ldy,
                             int64_t rows, int64_t cols, float p, uint64_t seed, uint64_t offset, const void* seed_dev, void* stream) {
  MSAM2_REQUIRE(x && y && rows > 0 && cols > 0 && p >= 0.f && p < 1.f, "dropout: bad arguments");
  const unsigned thr = (unsigned)fmin(4294967295.0, (double)p * 4294967296.0);
  const float inv_keep = 1.f / (1.f - p);
  dim3 grid((unsigned)min((int64_t)8192, cdiv(rows * cols, 256))), block(256);
  hipStream_t s = (hipStream_t)stream;
#define DR(TI, TO) hipLaunchKernelGGL((dropout_kernel<TI, TO>), grid, block, 0, s, (const TI*)x, ldx, residual, ldr, (TO*)y, ldy, rows, cols, thr, inv_keep, seed, offset, (const uint64_t*)seed_dev)
  if (x_is_16bit && y_is_16bit) DR(op16, op16);
  else if (x_is_16bit) DR(op16, float);
  else if (y_is_16bit) DR(float, op16);
  else DR(float, float);
#undef DR
  return msam2_check_launch("dropout");
}

__global__ void counter_bump_kernel(uint64_t* ctr, uint64_t* snapshot) {
  const uint64_t v = *ctr + 1;
  *ctr = v;
  if (snapshot) *snapshot = v;
}

// *counter += 1 on the device, the new value also written to *snapshot (optional): the sub-stream counter of the train-mode dropout,
// advanced inside the step so that hipGraph replays move it (see above).
extern "C" int msam2_counter_bump(void* counter_u64, void* snapshot_u64, void* stream) {
  MSAM2_REQUIRE(counter_u64, "counter_bump: null counter");
  hipLaunchKernelGGL(counter_bump_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (uint64_t*)counter_u64, (uint64_t*)snapshot_u64);
  return msam2_check_launch("counter_bump");
}
