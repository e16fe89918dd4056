// bf16 MFMA GEMM with fused epilogue for gfx950:   C[M,N] = epi( A[M,K] * W[N,K]^T )
//
// Serves every nn.Linear / 1x1 conv / im2col'ed conv on the hot path (reference call sites: hieradet.py:61,79,
// sam2_utils.py:127-131, transformer.py:241-243,261, memory_attention.py:96, image_encoder.py:112, mask_decoder.py:
// 240-256, memory_encoder.py:103-105,171-175).  Both operands are K-contiguous (activation rows and nn.Linear weight
// rows), so the 32x32x16 bf16 MFMA fragments (8 consecutive k per lane) are plain 16-byte LDS reads.
//
// Tile: BM x BN x 32, WM x WN waves, double-buffered LDS with 80-byte rows (conflict-free ds_read_b128), register
// staging issued before the MFMA phase and written after it (one barrier per k-step).
// Epilogue (all optional, fp32): + bias[n] -> activation -> * colscale[n] -> + residual[m % res_mod][n] -> bf16|f32.
#include "common.h"

struct GemmParams {
  const bf16* A;
  const bf16* W;
  const float* bias;
  const float* colscale;
  const void* res;
  void* C;
  int64_t lda, ldw, ldr, ldc;
  int64_t res_mod;
  int M, N, K;
  int act;          // 0 none, 1 gelu(erf), 2 relu, 3 sigmoid
  int res_is_bf16;  // residual dtype
  int out_is_bf16;  // output dtype
};

constexpr int GEMM_BK = 32;
constexpr int GEMM_LDS_STRIDE = 40;  // bf16 elements per LDS row (32 data + 8 pad = 80 B)

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void gemm_bf16_kernel(GemmParams p) {
  constexpr int NT = WM * WN * 64;
  constexpr int TM = BM / WM, TN = BN / WN;  // per-wave tile
  constexpr int FM = TM / 32, FN = TN / 32;  // 32x32 MFMA tiles per wave
  constexpr int A_CHUNKS = BM * 4, W_CHUNKS = BN * 4;
  constexpr int A_PER = (A_CHUNKS + NT - 1) / NT, W_PER = (W_CHUNKS + NT - 1) / NT;
  __shared__ __attribute__((aligned(16))) bf16 lds[2 * (BM + BN) * GEMM_LDS_STRIDE];
  bf16* As = lds;
  bf16* Ws = lds + 2 * BM * GEMM_LDS_STRIDE;

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
  const int nk = (p.K + GEMM_BK - 1) / GEMM_BK;

  auto gload = [&](int kt) {
    const int k0 = kt * GEMM_BK;
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int c = tid + i * NT;
      const int row = c >> 2, kc = (c & 3) * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (c < A_CHUNKS && m0 + row < p.M && k0 + kc < p.K)
        v = *reinterpret_cast<const uint4*>(p.A + (m0 + row) * p.lda + k0 + kc);
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < W_PER; ++i) {
      const int c = tid + i * NT;
      const int row = c >> 2, kc = (c & 3) * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (c < W_CHUNKS && n0 + row < p.N && k0 + kc < p.K)
        v = *reinterpret_cast<const uint4*>(p.W + (n0 + row) * p.ldw + k0 + kc);
      rw[i] = v;
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

  gload(0);
  lstore(0);
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[FM], bfr[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i)
        af[i] = *reinterpret_cast<const bf16x8*>(As + (cur * BM + wm * TM + i * 32 + r) * GEMM_LDS_STRIDE + ks * 16 + h * 8);
#pragma unroll
      for (int j = 0; j < FN; ++j)
        bfr[j] = *reinterpret_cast<const bf16x8*>(Ws + (cur * BN + wn * TN + j * 32 + r) * GEMM_LDS_STRIDE + ks * 16 + h * 8);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) lstore(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // epilogue: lane owns column n = .. + r and rows (e&3) + 8*(e>>2) + 4*h of each 32x32 tile
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int64_t n = n0 + wn * TN + j * 32 + r;
    if (n >= p.N) continue;
    const float bias = p.bias ? p.bias[n] : 0.f;
    const float cs = p.colscale ? p.colscale[n] : 1.f;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t m = m0 + wm * TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (m >= p.M) continue;
        float v = acc[i][j][e] + bias;
        if (p.act == 1) v = gelu_erf(v);
        else if (p.act == 2) v = fmaxf(v, 0.f);
        else if (p.act == 3) v = 1.f / (1.f + __expf(-v));
        v *= cs;
        if (p.res) {
          const int64_t rr = p.res_mod > 0 ? (m % p.res_mod) : m;
          v += p.res_is_bf16 ? bf2f(reinterpret_cast<const bf16*>(p.res)[rr * p.ldr + n])
                             : reinterpret_cast<const float*>(p.res)[rr * p.ldr + n];
        }
        if (p.out_is_bf16) reinterpret_cast<bf16*>(p.C)[m * p.ldc + n] = f2bf(v);
        else reinterpret_cast<float*>(p.C)[m * p.ldc + n] = v;
      }
    }
  }
}

template <int BM, int BN, int WM, int WN>
static void launch_gemm(const GemmParams& p, hipStream_t s) {
  dim3 grid(cdiv(p.N, BN), cdiv(p.M, BM));
  hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WM, WN>), grid, dim3(WM * WN * 64), 0, s, p);
}

extern "C" int msam2_gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                               const float* colscale, const void* residual, int64_t ldr, int res_is_bf16, int64_t res_mod,
                               void* C, int64_t ldc, int out_is_bf16, int64_t M, int64_t N, int64_t K, int act,
                               void* stream) {
  MSAM2_REQUIRE(A && W && C, "gemm: null operand");
  MSAM2_REQUIRE(M > 0 && N > 0 && K > 0, "gemm: empty problem M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
  MSAM2_REQUIRE(K % 8 == 0 && lda % 8 == 0 && ldw % 8 == 0, "gemm: K, lda, ldw must be multiples of 8 (16-byte rows)");
  MSAM2_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0, "gemm: A and W must be 16-byte aligned");
  MSAM2_REQUIRE(M < (1ll << 31) && N < (1ll << 31), "gemm: M/N too large");
  MSAM2_REQUIRE(act >= 0 && act <= 3, "gemm: bad activation %d", act);
  GemmParams p;
  p.A = (const bf16*)A; p.W = (const bf16*)W; p.bias = bias; p.colscale = colscale; p.res = residual; p.C = C;
  p.lda = lda; p.ldw = ldw; p.ldr = ldr; p.ldc = ldc; p.res_mod = res_mod;
  p.M = (int)M; p.N = (int)N; p.K = (int)K; p.act = act; p.res_is_bf16 = res_is_bf16; p.out_is_bf16 = out_is_bf16;
  hipStream_t s = (hipStream_t)stream;
  if (M <= 32) launch_gemm<32, 128, 1, 4>(p, s);
  else if (N <= 32) launch_gemm<128, 32, 4, 1>(p, s);
  else if (N <= 64 || (N % 128 != 0 && N % 64 == 0 && N < 512)) launch_gemm<128, 64, 2, 2>(p, s);
  else launch_gemm<128, 128, 2, 2>(p, s);
  return msam2_check_launch("gemm_bf16");
}
