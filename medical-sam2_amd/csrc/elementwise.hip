// HBM-bound row / elementwise kernels of the per-slice forward (gfx950): LayerNorm, strided add+cast, 2x2 max-pool,
// nearest-2x add, axial RoPE, bilinear up-sampling, and the input-independent position tables.
#include "common.h"

// ------------------------------------------------------------------------------------------------------------------
// LayerNorm over the last dim (nn.LayerNorm / LayerNorm2d on NHWC tokens): hieradet.py:138,166; memory_attention.py:
// 60,73,94,162; transformer.py:173-194; sam2_utils.py:137-149.  One wave per row, fp32 statistics (two-pass on
// registers), optional exact-erf GELU on the way out.  C <= 1024.
// ------------------------------------------------------------------------------------------------------------------
template <typename T>
struct Vec4 {};
template <>
struct Vec4<float> {
  typedef f32x4 type;
};
template <>
struct Vec4<op16> {
  typedef op16x4 type;
};

// 16 lanes per row (4 rows per wave), each lane owns the 4-element chunks lane16, lane16+16, ... of its row: 16 lanes x 16 B
// (fp32) are one 256-byte line per load instruction; statistics are reduced over the 16-lane group with 4 shuffles.
// Round 4: every load of a lane -- its row chunks AND the weight / bias chunks -- is issued UNCONDITIONALLY up front (chunk indices
// past the row are clamped and their values zeroed afterwards).  The first form loaded each chunk inside `if (live && ch < nch)` and
// summed it at once: hipcc put every load in its own basic block behind an `s_waitcnt vmcnt(0)`, so a 384-wide row paid SIX memory
// latencies one after the other (and the weight / bias loads of chunk j + 1 sat behind the store of chunk j on the one in-order
// counter): 9.8 us for 16384 x 384, 3.8 TB/s on a copy-shaped kernel.
template <typename TI, typename TO, int CHUNKS>
__global__ void layernorm_kernel(const TI* __restrict__ x, int64_t ldx, const float* __restrict__ w, const float* __restrict__ b,
                                 TO* __restrict__ y, int64_t ldy, int64_t rows, int C, float eps, int act, op16* __restrict__ y2 = nullptr,
                                 int64_t ldy2 = 0) {
  typedef typename Vec4<TI>::type VI;
  typedef typename Vec4<TO>::type VO;
  const int64_t row = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4);
  const int l16 = threadIdx.x & 15;
  const bool live = row < rows;
  const int nch = C >> 2;
  const TI* xr = x + (live ? row : 0) * ldx;
  VI t[CHUNKS];
  constexpr bool WB_UP_FRONT = CHUNKS <= 6;                 // (wider rows: 96 more registers cost more than the serialised loads, measured at C = 768)
  f32x4 ww[WB_UP_FRONT ? CHUNKS : 1], bb[WB_UP_FRONT ? CHUNKS : 1];
#pragma unroll
  for (int j = 0; j < CHUNKS; ++j) {
    const int ch = min(l16 + 16 * j, nch - 1);             // clamped: always a valid address, one straight line of loads
    t[j] = *reinterpret_cast<const VI*>(xr + ch * 4);
  }
  if constexpr (WB_UP_FRONT) {
#pragma unroll
    for (int j = 0; j < CHUNKS; ++j) {
      const int ch = min(l16 + 16 * j, nch - 1);
      ww[j] = *reinterpret_cast<const f32x4*>(w + ch * 4);
      bb[j] = *reinterpret_cast<const f32x4*>(b + ch * 4);
    }
  }
  float v[CHUNKS][4];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < CHUNKS; ++j) {
    const bool in = l16 + 16 * j < nch;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[j][e] = in ? (float)t[j][e] : 0.f;
      s += v[j][e];
    }
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / C;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < CHUNKS; ++j) {
    const bool in = l16 + 16 * j < nch;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = in ? v[j][e] - mean : 0.f;
      q += d * d;
    }
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = 1.0f / sqrtf(q / C + eps);
  if (!live) return;
  TO* yr = y + row * ldy;
#pragma unroll
  for (int j = 0; j < CHUNKS; ++j) {
    const int ch = l16 + 16 * j;
    if (ch < nch) {
      VO o;
      f32x4 wj, bj;
      if constexpr (WB_UP_FRONT) {
        wj = ww[j];
        bj = bb[j];
      } else {
        wj = *reinterpret_cast<const f32x4*>(w + ch * 4);
        bj = *reinterpret_cast<const f32x4*>(b + ch * 4);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float u = (v[j][e] - mean) * rstd * wj[e] + bj[e];
        if (act == 1) u = gelu_erf_as(u);
        o[e] = f2out<TO>(u);
      }
      *reinterpret_cast<VO*>(yr + ch * 4) = o;
      if (y2) {                                              // second, 16-bit copy of the same row (msam2_layernorm_dual): the GEMM operand of the next layer
        op16x4 o2;
#pragma unroll
        for (int e = 0; e < 4; ++e) o2[e] = f2op((float)o[e]);
        *reinterpret_cast<op16x4*>(y2 + row * ldy2 + ch * 4) = o2;
      }
    }
  }
}

// scalar fallback for C % 4 != 0 or unaligned rows: one wave per row
template <typename TI, typename TO>
__global__ void layernorm_scalar_kernel(const TI* __restrict__ x, int64_t ldx, const float* __restrict__ w,
                                        const float* __restrict__ b, TO* __restrict__ y, int64_t ldy, int64_t rows, int C,
                                        float eps, int act) {
  const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const TI* xr = x + row * ldx;
  float v[16];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = lane + i * 64;
    v[i] = (c < C) ? (float)xr[c] : 0.f;
    s += v[i];
  }
  const float mean = wave_sum(s) / C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = lane + i * 64;
    const float d = (c < C) ? v[i] - mean : 0.f;
    q += d * d;
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / C + eps);
  TO* yr = y + row * ldy;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = lane + i * 64;
    if (c < C) {
      float o = (v[i] - mean) * rstd * w[c] + b[c];
      if (act == 1) o = gelu_erf_as(o);
      yr[c] = f2out<TO>(o);
    }
  }
}

extern "C" int msam2_layernorm(const void* x, int in_is_16bit, int64_t ldx, const float* weight, const float* bias, void* y,
                               int out_is_16bit, int64_t ldy, int64_t rows, int64_t C, float eps, int act, void* stream) {
  MSAM2_REQUIRE(x && y && weight && bias, "layernorm: null tensor");
  MSAM2_REQUIRE(rows > 0 && C > 0 && C <= 1024, "layernorm: rows=%lld C=%lld unsupported (C<=1024)", (long long)rows, (long long)C);
  hipStream_t s = (hipStream_t)stream;
  const bool vec = (C % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && (((uintptr_t)x & 15) == 0) && (((uintptr_t)y & 7) == 0) &&
                   (((uintptr_t)weight & 15) == 0) && (((uintptr_t)bias & 15) == 0);
  if (vec) {
    dim3 grid(cdiv(rows * 16, 256)), block(256);
    const int chunks = (int)((C / 4 + 15) / 16);
#define LN_V(TI, TO, CH) \
  hipLaunchKernelGGL((layernorm_kernel<TI, TO, CH>), grid, block, 0, s, (const TI*)x, ldx, weight, bias, (TO*)y, ldy, rows, (int)C, eps, act)
#define LN_VC(TI, TO)                                     \
  do {                                                    \
    if (chunks <= 2) LN_V(TI, TO, 2);                     \
    else if (chunks <= 4) LN_V(TI, TO, 4);                \
    else if (chunks <= 6) LN_V(TI, TO, 6);                \
    else if (chunks <= 12) LN_V(TI, TO, 12);              \
    else LN_V(TI, TO, 16);                                \
  } while (0)
    if (in_is_16bit && out_is_16bit) LN_VC(op16, op16);
    else if (in_is_16bit) LN_VC(op16, float);
    else if (out_is_16bit) LN_VC(float, op16);
    else LN_VC(float, float);
#undef LN_VC
#undef LN_V
  } else {
    dim3 grid(cdiv(rows * 64, 256)), block(256);
#define LN_LAUNCH(TI, TO) \
  hipLaunchKernelGGL((layernorm_scalar_kernel<TI, TO>), grid, block, 0, s, (const TI*)x, ldx, weight, bias, (TO*)y, ldy, rows, (int)C, eps, act)
    if (in_is_16bit && out_is_16bit) LN_LAUNCH(op16, op16);
    else if (in_is_16bit) LN_LAUNCH(op16, float);
    else if (out_is_16bit) LN_LAUNCH(float, op16);
    else LN_LAUNCH(float, float);
#undef LN_LAUNCH
  }
  return msam2_check_launch("layernorm");
}

// LayerNorm with TWO outputs: the fp32 rows (the residual stream the next block adds to) and their 16-bit copy (the operand of the next
// block's projections) from one pass -- replaces msam2_layernorm + msam2_add_cast where both are needed (two-way decoder, transformer.py:190-196).
extern "C" int msam2_layernorm_dual(const float* x, int64_t ldx, const float* weight, const float* bias, float* y, int64_t ldy, void* y16,
                                    int64_t ldy16, int64_t rows, int64_t C, float eps, void* stream) {
  MSAM2_REQUIRE(x && y && y16 && weight && bias, "layernorm_dual: null tensor");
  MSAM2_REQUIRE(rows > 0 && C > 0 && C <= 1024 && C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldy16 % 4 == 0,
                "layernorm_dual: rows=%lld C=%lld unsupported (C <= 1024, multiples of 4)", (long long)rows, (long long)C);
  MSAM2_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)weight | (uintptr_t)bias) & 15) == 0 && ((uintptr_t)y16 & 7) == 0, "layernorm_dual: alignment");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(cdiv(rows * 16, 256)), block(256);
  const int chunks = (int)((C / 4 + 15) / 16);
#define LN_D(CH) \
  hipLaunchKernelGGL((layernorm_kernel<float, float, CH>), grid, block, 0, s, x, ldx, weight, bias, y, ldy, rows, (int)C, eps, 0, (op16*)y16, ldy16)
  if (chunks <= 2) LN_D(2);
  else if (chunks <= 4) LN_D(4);
  else if (chunks <= 6) LN_D(6);
  else if (chunks <= 12) LN_D(12);
  else LN_D(16);
#undef LN_D
  return msam2_check_launch("layernorm_dual");
}

// ------------------------------------------------------------------------------------------------------------------
// out[i,j,c] = a[i,j,c] + alpha * b[i,j,c]   on a logical [D0, D1, C] volume; a and b carry arbitrary element strides
// for the two outer dims (0 stride = broadcast), the channel dim is contiguous; out is contiguous.  Covers every
// "x + pos" / cast / seq-first<->batch-first move of the path (memory_attention.py:139-147, 74-76; transformer.py:
// 175-190; mask_decoder.py:231; sam2_base.py:642).
// ------------------------------------------------------------------------------------------------------------------
template <typename TA, typename TB, typename TO>
__global__ void add_cast_kernel(const TA* __restrict__ a, int64_t a_s0, int64_t a_s1, const TB* __restrict__ b, int64_t b_s0,
                                int64_t b_s1, float alpha, TO* __restrict__ out, int64_t D0, int64_t D1, int C) {
  const int64_t total = D0 * D1 * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = i % C;
    const int64_t rj = i / C;
    const int64_t j = rj % D1, r = rj / D1;
    float v = (float)a[r * a_s0 + j * a_s1 + c];
    if (b) v += alpha * (float)b[r * b_s0 + j * b_s1 + c];
    out[i] = f2out<TO>(v);
  }
}

// vector form: C % 4 == 0 and 4-element groups aligned in a, b and out; one thread per group, 32-bit index math
template <typename TA, typename TB, typename TO>
__global__ void add_cast_vec_kernel(const TA* __restrict__ a, int64_t a_s0, int64_t a_s1, const TB* __restrict__ b, int64_t b_s0,
                                    int64_t b_s1, float alpha, TO* __restrict__ out, unsigned D0, unsigned D1, unsigned C4) {
  typedef typename Vec4<TA>::type VA;
  typedef typename Vec4<TB>::type VB;
  typedef typename Vec4<TO>::type VO;
  const unsigned rows = D0 * D1;
  const unsigned total = rows * C4;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const unsigned g = i % C4, rj = i / C4;
    const unsigned j = rj % D1, r = rj / D1;
    const VA va = *reinterpret_cast<const VA*>(a + r * a_s0 + j * a_s1 + g * 4);
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (float)va[e];
    if (b) {
      const VB vb = *reinterpret_cast<const VB*>(b + r * b_s0 + j * b_s1 + g * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += alpha * (float)vb[e];
    }
    VO o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = f2out<TO>(v[e]);
    *reinterpret_cast<VO*>(out + (int64_t)i * 4) = o;
  }
}

extern "C" int msam2_add_cast(const void* a, int a_is_16bit, int64_t a_s0, int64_t a_s1, const void* b, int b_is_16bit, int64_t b_s0,
                              int64_t b_s1, float alpha, void* out, int out_is_16bit, int64_t D0, int64_t D1, int64_t C,
                              void* stream) {
  MSAM2_REQUIRE(a && out, "add_cast: null tensor");
  MSAM2_REQUIRE(D0 > 0 && D1 > 0 && C > 0, "add_cast: empty volume");
  const int64_t total = D0 * D1 * C;
  hipStream_t s = (hipStream_t)stream;
  const int key = (a_is_16bit ? 4 : 0) | (b_is_16bit ? 2 : 0) | (out_is_16bit ? 1 : 0);
  const int asz = a_is_16bit ? 2 : 4, bsz = b_is_16bit ? 2 : 4, osz = out_is_16bit ? 2 : 4;
  const bool vec = (C % 4 == 0) && total / 4 < (1ll << 31) && (a_s0 % 4 == 0) && (a_s1 % 4 == 0) && (((uintptr_t)a % (4 * asz)) == 0) &&
                   (((uintptr_t)out % (4 * osz)) == 0) &&
                   (!b || ((b_s0 % 4 == 0) && (b_s1 % 4 == 0) && (((uintptr_t)b % (4 * bsz)) == 0)));
  if (vec) {
    dim3 grid((unsigned)min((int64_t)16384, (total / 4 + 255) / 256)), block(256);
#define ACV(TA, TB, TO)                                                                                                          \
  hipLaunchKernelGGL((add_cast_vec_kernel<TA, TB, TO>), grid, block, 0, s, (const TA*)a, a_s0, a_s1, (const TB*)b, b_s0, b_s1, alpha, \
                     (TO*)out, (unsigned)D0, (unsigned)D1, (unsigned)(C / 4))
    switch (key) {
      case 0: ACV(float, float, float); break;
      case 1: ACV(float, float, op16); break;
      case 2: ACV(float, op16, float); break;
      case 3: ACV(float, op16, op16); break;
      case 4: ACV(op16, float, float); break;
      case 5: ACV(op16, float, op16); break;
      case 6: ACV(op16, op16, float); break;
      default: ACV(op16, op16, op16); break;
    }
#undef ACV
    return msam2_check_launch("add_cast");
  }
  dim3 grid((unsigned)min((int64_t)8192, (total + 255) / 256)), block(256);
#define AC(TA, TB, TO)                                                                                                   \
  hipLaunchKernelGGL((add_cast_kernel<TA, TB, TO>), grid, block, 0, s, (const TA*)a, a_s0, a_s1, (const TB*)b, b_s0, b_s1, \
                     alpha, (TO*)out, D0, D1, (int)C)
  switch (key) {
    case 0: AC(float, float, float); break;
    case 1: AC(float, float, op16); break;
    case 2: AC(float, op16, float); break;
    case 3: AC(float, op16, op16); break;
    case 4: AC(op16, float, float); break;
    case 5: AC(op16, float, op16); break;
    case 6: AC(op16, op16, float); break;
    default: AC(op16, op16, op16); break;
  }
#undef AC
  return msam2_check_launch("add_cast");
}

// ------------------------------------------------------------------------------------------------------------------
// 2x2/stride-2 max pool on NHWC tokens (do_pool, hieradet.py:23-34): x [B,H,W,C] with token stride ldx -> [B,H/2,W/2,C]
// ------------------------------------------------------------------------------------------------------------------
template <typename TI, typename TO>
__global__ void maxpool2x2_kernel(const TI* __restrict__ x, int64_t ldx, TO* __restrict__ y, int64_t ldy, int B, int H, int W,
                                  int C) {
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
    const float v0 = (float)x[base * ldx + c], v1 = (float)x[(base + 1) * ldx + c];
    const float v2 = (float)x[(base + W) * ldx + c], v3 = (float)x[(base + W + 1) * ldx + c];
    y[(((int64_t)b * Ho + yo) * Wo + xo) * ldy + c] = (TO)fmaxf(fmaxf(v0, v1), fmaxf(v2, v3));
  }
}

extern "C" int msam2_maxpool2x2(const void* x, int in_is_16bit, int64_t ldx, void* y, int out_is_16bit, int64_t ldy, int64_t B,
                                int64_t H, int64_t W, int64_t C, void* stream) {
  MSAM2_REQUIRE(x && y, "maxpool2x2: null tensor");
  MSAM2_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "maxpool2x2: H, W must be even");
  const int64_t total = B * (H / 2) * (W / 2) * C;
  dim3 grid((unsigned)min((int64_t)8192, (total + 255) / 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
#define MP(TI, TO) \
  hipLaunchKernelGGL((maxpool2x2_kernel<TI, TO>), grid, block, 0, s, (const TI*)x, ldx, (TO*)y, ldy, (int)B, (int)H, (int)W, (int)C)
  if (in_is_16bit && out_is_16bit) MP(op16, op16);
  else if (in_is_16bit) MP(op16, float);
  else if (out_is_16bit) MP(float, op16);
  else MP(float, float);
#undef MP
  return msam2_check_launch("maxpool2x2");
}

// ------------------------------------------------------------------------------------------------------------------
// FPN top-down step (image_encoder.py:113-124, nearest, scale 2, sum fuse): y[b,i,j,c] += top[b,i/2,j/2,c], fp32 NHWC
// ------------------------------------------------------------------------------------------------------------------
__global__ void upsample2x_add_kernel(float* __restrict__ y, const float* __restrict__ top, int B, int H, int W, int C) {
  const int64_t total = (int64_t)B * H * W * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = i % C;
    int64_t t = i / C;
    const int xx = t % W;
    t /= W;
    const int yy = t % H;
    const int b = t / H;
    y[i] += top[(((int64_t)b * (H / 2) + yy / 2) * (W / 2) + xx / 2) * C + c];
  }
}

// four channels per thread, 32-bit index arithmetic (C % 4 == 0, fewer than 2^31 groups, 16-byte aligned tensors): the one-element form
// spends four 64-bit divisions on every 4 bytes it moves (14.9 us for the FPN's one top-down step; round 4)
__global__ void upsample2x_add4_kernel(float* __restrict__ y, const float* __restrict__ top, unsigned B, unsigned H, unsigned W, unsigned C4) {
  const unsigned total = B * H * W * C4;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const unsigned t0 = i / C4, c = i - t0 * C4;
    const unsigned t1 = t0 / W, xx = t0 - t1 * W;
    const unsigned b = t1 / H, yy = t1 - b * H;
    const f32x4 a = *reinterpret_cast<const f32x4*>(y + (int64_t)i * 4);
    const f32x4 tp = *reinterpret_cast<const f32x4*>(top + ((int64_t)(b * (H / 2) + yy / 2) * (W / 2) + xx / 2) * (C4 * 4) + c * 4);
    *reinterpret_cast<f32x4*>(y + (int64_t)i * 4) = a + tp;
  }
}

extern "C" int msam2_upsample2x_add(void* y, const void* top, int64_t B, int64_t H, int64_t W, int64_t C, void* stream) {
  MSAM2_REQUIRE(y && top, "upsample2x_add: null tensor");
  MSAM2_REQUIRE(H % 2 == 0 && W % 2 == 0 && B > 0 && C > 0, "upsample2x_add: bad shape");
  const int64_t total = B * H * W * C;
  if (C % 4 == 0 && total / 4 < (1ll << 31) && (((uintptr_t)y | (uintptr_t)top) & 15) == 0) {
    hipLaunchKernelGGL(upsample2x_add4_kernel, dim3((unsigned)min((int64_t)16384, (total / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (float*)y, (const float*)top, (unsigned)B, (unsigned)H, (unsigned)W, (unsigned)(C / 4));
    return msam2_check_launch("upsample2x_add");
  }
  hipLaunchKernelGGL(upsample2x_add_kernel, dim3((unsigned)min((int64_t)8192, (total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, (float*)y, (const float*)top, (int)B, (int)H, (int)W, (int)C);
  return msam2_check_launch("upsample2x_add");
}

// ------------------------------------------------------------------------------------------------------------------
// Axial RoPE (position_encoding.py:174-216; transformer.py:299-315).  Table: cos/sin [n_pos, D/2] fp32 where pair i
// < D/4 rotates with x = pos % side and the rest with y = pos / side.  In-place on op16 rows [B, L, ld]: rows
// l < n_rope of every batch are rotated with position l % n_pos (rope_k_repeat tiles the table over the keys).
// ------------------------------------------------------------------------------------------------------------------
__global__ void rope_table_kernel(float* __restrict__ cs, float* __restrict__ sn, int side, int D, float theta) {
  const int n_pairs = D / 2, nq = D / 4;
  const int64_t total = (int64_t)side * side * n_pairs;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int pr = i % n_pairs;
    const int pos = i / n_pairs;
    const int f = pr % nq;
    const float freq = 1.0f / powf(theta, (float)(4 * f) / (float)D);
    const float t = (pr < nq) ? (float)(pos % side) : (float)(pos / side);
    const float ang = t * freq;
    cs[i] = cosf(ang);
    sn[i] = sinf(ang);
  }
}

extern "C" int msam2_rope_table(float* cos_out, float* sin_out, int64_t side, int64_t D, float theta, void* stream) {
  MSAM2_REQUIRE(cos_out && sin_out && side > 0 && D % 4 == 0, "rope_table: bad arguments");
  hipLaunchKernelGGL(rope_table_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, cos_out, sin_out, (int)side, (int)D, theta);
  return msam2_check_launch("rope_table");
}

__global__ void rope_inplace_kernel(op16* __restrict__ x, int64_t bs, int64_t ld, int B, int L, int n_rope, int n_pos, int D,
                                    const float* __restrict__ cs, const float* __restrict__ sn) {
  const int hp = D / 2;
  const int64_t total = (int64_t)B * n_rope * hp;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int pr = i % hp;
    int64_t t = i / hp;
    const int l = t % n_rope;
    const int b = t / n_rope;
    op16x2* px = reinterpret_cast<op16x2*>(x + b * bs + (int64_t)l * ld) + pr;
    const op16x2 v = *px;
    const float re = (float)v[0], im = (float)v[1];
    const int pos = l % n_pos;
    const float c = cs[(int64_t)pos * hp + pr], s = sn[(int64_t)pos * hp + pr];
    op16x2 o;
    o[0] = (op16)(re * c - im * s);
    o[1] = (op16)(re * s + im * c);
    *px = o;
  }
}

extern "C" int msam2_rope_inplace(void* x, int64_t batch_stride, int64_t ld, int64_t B, int64_t L, int64_t n_rope, int64_t n_pos,
                                  int64_t D, const float* cos_t, const float* sin_t, void* stream) {
  MSAM2_REQUIRE(x && cos_t && sin_t, "rope: null tensor");
  MSAM2_REQUIRE(D % 4 == 0 && ld % 2 == 0 && batch_stride % 2 == 0 && n_rope >= 0 && n_rope <= L && n_pos > 0, "rope: bad shape");
  if (n_rope == 0) return MSAM2_OK;
  const int64_t total = B * n_rope * (D / 2);
  hipLaunchKernelGGL(rope_inplace_kernel, dim3((unsigned)min((int64_t)8192, (total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, (op16*)x, batch_stride, ld, (int)B, (int)L, (int)n_rope, (int)n_pos, (int)D, cos_t, sin_t);
  return msam2_check_launch("rope_inplace");
}

// ------------------------------------------------------------------------------------------------------------------
// Bilinear resize, align_corners=False, fp32 planes [P, h, w] -> [P, H, W] (F.interpolate at sam2_base.py:368-373)
// ------------------------------------------------------------------------------------------------------------------
__global__ void bilinear_kernel(const float* __restrict__ x, float* __restrict__ y, int P, int h, int w, int H, int W) {
  const float sy = (float)h / H, sx = (float)w / W;
  const int64_t total = (int64_t)P * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int X = i % W;
    int64_t t = i / W;
    const int Y = t % H;
    const int pl = t / H;
    float fy = fmaxf((Y + 0.5f) * sy - 0.5f, 0.f), fx = fmaxf((X + 0.5f) * sx - 0.5f, 0.f);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
    const float ly = fy - y0, lx = fx - x0;
    const float* p = x + (int64_t)pl * h * w;
    const float v = (1.f - ly) * ((1.f - lx) * p[y0 * w + x0] + lx * p[y0 * w + x1]) +
                    ly * ((1.f - lx) * p[y1 * w + x0] + lx * p[y1 * w + x1]);
    y[i] = v;
  }
}

// W % 4 == 0: four consecutive output pixels per thread, one 16-byte store (the scalar form above wrote 4 bytes per lane and did its
// index arithmetic in 64 bits: 21.7 us for the 16.8 MB of 4 x 1024^2 masks).  Same expression per pixel as the scalar form: bit-identical.
__global__ void bilinear4_kernel(const float* __restrict__ x, float* __restrict__ y, int P, int h, int w, int H, int W) {
  const float sy = (float)h / H, sx = (float)w / W;
  const int W4 = W >> 2;
  const unsigned total = (unsigned)P * (unsigned)H * (unsigned)W4;      // checked on the host: < 2^31
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const unsigned X4 = i % (unsigned)W4, t = i / (unsigned)W4;
    const unsigned Y = t % (unsigned)H, pl = t / (unsigned)H;
    const float fy = fmaxf((Y + 0.5f) * sy - 0.5f, 0.f);
    const int y0 = (int)fy, y1 = min(y0 + 1, h - 1);
    const float ly = fy - y0;
    const float* p0 = x + ((int64_t)pl * h + y0) * w;
    const float* p1 = x + ((int64_t)pl * h + y1) * w;
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int X = (int)X4 * 4 + e;
      const float fx = fmaxf((X + 0.5f) * sx - 0.5f, 0.f);
      const int x0 = (int)fx, x1 = min(x0 + 1, w - 1);
      const float lx = fx - x0;
      o[e] = (1.f - ly) * ((1.f - lx) * p0[x0] + lx * p0[x1]) + ly * ((1.f - lx) * p1[x0] + lx * p1[x1]);
    }
    *reinterpret_cast<f32x4*>(y + ((int64_t)pl * H + Y) * W + X4 * 4) = o;
  }
}

extern "C" int msam2_bilinear_upsample(const float* x, float* y, int64_t planes, int64_t h, int64_t w, int64_t H, int64_t W,
                                       void* stream) {
  MSAM2_REQUIRE(x && y && planes > 0 && h > 0 && w > 0 && H > 0 && W > 0, "bilinear: bad arguments");
  const int64_t total = planes * H * W;
  if (W % 4 == 0 && ((uintptr_t)y & 15) == 0 && total / 4 < (1ll << 31)) {
    hipLaunchKernelGGL(bilinear4_kernel, dim3((unsigned)min((int64_t)16384, (total / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       x, y, (int)planes, (int)h, (int)w, (int)H, (int)W);
    return msam2_check_launch("bilinear_upsample");
  }
  hipLaunchKernelGGL(bilinear_kernel, dim3((unsigned)min((int64_t)16384, (total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     x, y, (int)planes, (int)h, (int)w, (int)H, (int)W);
  return msam2_check_launch("bilinear_upsample");
}

// ------------------------------------------------------------------------------------------------------------------
// Input-independent tables, generated once per (model, size) on the device.
//  * sine_pos_2d: PositionEmbeddingSine.forward (position_encoding.py:78-112) -> token-major [h*w, C] fp32
//  * fourier_pe_grid: PromptEncoder.get_dense_pe (prompt_encoder.py:68-77, position_encoding.py:130-151) -> [h*w, C]
//  * hiera_pos_embed: Hiera._get_pos_embed (hieradet.py:269-277): bicubic (A=-0.75, align_corners=False, clamped
//    taps) resize of pos_embed [C, bh, bw] to h x w + tiled pos_embed_window [C, 8, 8] -> token-major [h*w, C] fp32
// ------------------------------------------------------------------------------------------------------------------
__global__ void sine_pos_kernel(float* __restrict__ out, int h, int w, int C, float temperature) {
  const int npf = C / 2;
  const float two_pi = 6.283185307179586f, eps = 1e-6f;
  const int64_t total = (int64_t)h * w * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = i % C;
    const int64_t t = i / C;
    const int xx = t % w, yy = t / w;
    const bool is_y = c < npf;
    const int k = is_y ? c : c - npf;
    const float e = is_y ? (yy + 1) / ((float)h + eps) * two_pi : (xx + 1) / ((float)w + eps) * two_pi;
    const float dim_t = powf(temperature, (float)(2 * (k / 2)) / (float)npf);
    const float a = e / dim_t;
    out[i] = (k & 1) ? cosf(a) : sinf(a);
  }
}

extern "C" int msam2_sine_pos_2d(float* out, int64_t h, int64_t w, int64_t C, float temperature, void* stream) {
  MSAM2_REQUIRE(out && h > 0 && w > 0 && C % 4 == 0, "sine_pos_2d: bad arguments");
  hipLaunchKernelGGL(sine_pos_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, out, (int)h, (int)w, (int)C, temperature);
  return msam2_check_launch("sine_pos_2d");
}

__global__ void fourier_grid_kernel(float* __restrict__ out, const float* __restrict__ G, int h, int w, int C) {
  const int nf = C / 2;
  const int64_t total = (int64_t)h * w * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = i % C;
    const int64_t t = i / C;
    const int xx = t % w, yy = t / w;
    const float cx = 2.f * ((xx + 0.5f) / w) - 1.f, cy = 2.f * ((yy + 0.5f) / h) - 1.f;
    const int f = c % nf;
    const float a = 6.283185307179586f * (cx * G[f] + cy * G[nf + f]);
    out[i] = (c < nf) ? sinf(a) : cosf(a);
  }
}

extern "C" int msam2_fourier_pe_grid(float* out, const float* gauss, int64_t h, int64_t w, int64_t C, void* stream) {
  MSAM2_REQUIRE(out && gauss && h > 0 && w > 0 && C % 2 == 0, "fourier_pe_grid: bad arguments");
  hipLaunchKernelGGL(fourier_grid_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, out, gauss, (int)h, (int)w, (int)C);
  return msam2_check_launch("fourier_pe_grid");
}

__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

__global__ void hiera_pos_kernel(float* __restrict__ out, const float* __restrict__ bkg, const float* __restrict__ win, int C,
                                 int bh, int bw, int h, int w, int wsz) {
  const float A = -0.75f;
  const float sy = (float)bh / h, sx = (float)bw / w;
  const int64_t total = (int64_t)h * w * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = i % C;
    const int64_t t = i / C;
    const int xx = t % w, yy = t / w;
    const float fy = (yy + 0.5f) * sy - 0.5f, fx = (xx + 0.5f) * sx - 0.5f;
    const int iy = (int)floorf(fy), ix = (int)floorf(fx);
    const float ty = fy - iy, tx = fx - ix;
    float wy[4] = {cubic2(ty + 1.f, A), cubic1(ty, A), cubic1(1.f - ty, A), cubic2(2.f - ty, A)};
    float wx[4] = {cubic2(tx + 1.f, A), cubic1(tx, A), cubic1(1.f - tx, A), cubic2(2.f - tx, A)};
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int y = min(max(iy - 1 + a, 0), bh - 1);
      float rowv = 0.f;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int x = min(max(ix - 1 + b, 0), bw - 1);
        rowv += wx[b] * bkg[((int64_t)c * bh + y) * bw + x];
      }
      acc += wy[a] * rowv;
    }
    out[i] = acc + win[((int64_t)c * wsz + (yy % wsz)) * wsz + (xx % wsz)];
  }
}

extern "C" int msam2_hiera_pos_embed(float* out, const float* pos_embed, const float* pos_embed_window, int64_t C, int64_t bh,
                                     int64_t bw, int64_t h, int64_t w, int64_t window, void* stream) {
  MSAM2_REQUIRE(out && pos_embed && pos_embed_window, "hiera_pos_embed: null tensor");
  MSAM2_REQUIRE(h % window == 0 && w % window == 0, "hiera_pos_embed: token grid must be a multiple of the window embedding");
  hipLaunchKernelGGL(hiera_pos_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, out, pos_embed, pos_embed_window, (int)C,
                     (int)bh, (int)bw, (int)h, (int)w, (int)window);
  return msam2_check_launch("hiera_pos_embed");
}

// ------------------------------------------------------------------------------------------------------------------
// Anti-aliased bilinear down-sampling by an integer factor (F.interpolate(..., mode="bilinear", antialias=True,
// align_corners=False) at sam2_base.py:321-327,421-427): separable triangle filter of half-width `f` source pixels,
// weights renormalised at the borders.  Optional affine on the input: v = x * in_scale + in_bias (mask -> +-10 logits).
// ------------------------------------------------------------------------------------------------------------------
__global__ void aa_downsample_kernel(const float* __restrict__ x, float* __restrict__ y, int P, int H, int W, int f, float in_scale,
                                     float in_bias) {
  const int Ho = H / f, Wo = W / f;
  const int64_t total = (int64_t)P * Ho * Wo;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int xo = i % Wo;
    int64_t t = i / Wo;
    const int yo = t % Ho;
    const int pl = t / Ho;
    const float cy = (yo + 0.5f) * f, cx = (xo + 0.5f) * f;
    const int y0 = max((int)(cy - f + 0.5f), 0), y1 = min((int)(cy + f + 0.5f), H);
    const int x0 = max((int)(cx - f + 0.5f), 0), x1 = min((int)(cx + f + 0.5f), W);
    float wys = 0.f, wxs = 0.f;
    for (int yy = y0; yy < y1; ++yy) wys += fmaxf(0.f, 1.f - fabsf((yy + 0.5f - cy) / f));
    for (int xx = x0; xx < x1; ++xx) wxs += fmaxf(0.f, 1.f - fabsf((xx + 0.5f - cx) / f));
    float acc = 0.f;
    const float* p = x + (int64_t)pl * H * W;
    for (int yy = y0; yy < y1; ++yy) {
      const float wy = fmaxf(0.f, 1.f - fabsf((yy + 0.5f - cy) / f)) / wys;
      float rowv = 0.f;
      for (int xx = x0; xx < x1; ++xx) rowv += (fmaxf(0.f, 1.f - fabsf((xx + 0.5f - cx) / f)) / wxs) * (p[yy * W + xx] * in_scale + in_bias);
      acc += wy * rowv;
    }
    y[i] = acc;
  }
}

extern "C" int msam2_aa_downsample(const float* x, float* y, int64_t planes, int64_t H, int64_t W, int64_t factor, float in_scale,
                                   float in_bias, void* stream) {
  MSAM2_REQUIRE(x && y && planes > 0 && factor >= 1 && H % factor == 0 && W % factor == 0, "aa_downsample: bad arguments");
  const int64_t total = planes * (H / factor) * (W / factor);
  hipLaunchKernelGGL(aa_downsample_kernel, dim3((unsigned)min((int64_t)8192, (total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                     y, (int)planes, (int)H, (int)W, (int)factor, in_scale, in_bias);
  return msam2_check_launch("aa_downsample");
}

// rows[b, :] = value where score[b] <= 0   (object-score gating of the mask logits, sam2_base.py:354-363)
__global__ void gate_rows_kernel(float* __restrict__ x, const float* __restrict__ score, float value, int64_t row_len) {
  const int b = blockIdx.y;
  if (score[b] > 0.f) return;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < row_len; i += (int64_t)gridDim.x * blockDim.x)
    x[(int64_t)b * row_len + i] = value;
}

extern "C" int msam2_gate_rows(float* x, const float* score, float value, int64_t B, int64_t row_len, void* stream) {
  MSAM2_REQUIRE(x && score && B > 0 && row_len > 0, "gate_rows: bad arguments");
  hipLaunchKernelGGL(gate_rows_kernel, dim3((unsigned)min((int64_t)1024, (row_len + 255) / 256), (unsigned)B), dim3(256), 0,
                     (hipStream_t)stream, x, score, value, row_len);
  return msam2_check_launch("gate_rows");
}

// out[b] = 1.0 if any x[b, :] > 0 else 0.0   (sam2_base.py:445-447)
__global__ void any_positive_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t row_len) {
  const int b = blockIdx.x;
  int found = 0;
  for (int64_t i = threadIdx.x; i < row_len; i += blockDim.x) found |= x[(int64_t)b * row_len + i] > 0.f;
  found = __syncthreads_or(found);
  if (threadIdx.x == 0) out[b] = found ? 1.f : 0.f;
}

extern "C" int msam2_any_positive(const float* x, float* out, int64_t B, int64_t row_len, void* stream) {
  MSAM2_REQUIRE(x && out && B > 0 && row_len > 0, "any_positive: bad arguments");
  hipLaunchKernelGGL(any_positive_kernel, dim3((unsigned)B), dim3(1024), 0, (hipStream_t)stream, x, out, row_len);
  return msam2_check_launch("any_positive");
}

// ------------------------------------------------------------------------------------------------------------------
// Image pre-processing of SAM2Transforms.__call__ (utils/transforms.py:22-37): uint8 HWC RGB -> /255 -> bilinear resize to
// S x S (align_corners=False, no antialias: identity when the image already has the model resolution) -> (x - mean) / std,
// written as fp32 [3, S, S].
// ------------------------------------------------------------------------------------------------------------------
__global__ void image_prep_kernel(const uint8_t* __restrict__ img, float* __restrict__ out, int H, int W, int S, float m0, float m1,
                                  float m2, float s0, float s1, float s2) {
  const float sy = (float)H / S, sx = (float)W / S;
  const int64_t total = (int64_t)3 * S * S;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int X = i % S;
    const int Y = (i / S) % S;
    const int c = i / ((int64_t)S * S);
    const float fy = fmaxf((Y + 0.5f) * sy - 0.5f, 0.f), fx = fmaxf((X + 0.5f) * sx - 0.5f, 0.f);
    const int y0 = min((int)fy, H - 1), x0 = min((int)fx, W - 1);
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float ly = fy - y0, lx = fx - x0;
    auto px = [&](int y, int x) { return (float)img[((int64_t)y * W + x) * 3 + c] * (1.0f / 255.0f); };
    const float v = (1.f - ly) * ((1.f - lx) * px(y0, x0) + lx * px(y0, x1)) + ly * ((1.f - lx) * px(y1, x0) + lx * px(y1, x1));
    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
    out[i] = (v - mean) / sd;
  }
}

extern "C" int msam2_image_prep(const uint8_t* img_hwc, float* out_chw, int64_t H, int64_t W, int64_t S, const float* mean3,
                                const float* std3, void* stream) {
  MSAM2_REQUIRE(img_hwc && out_chw && mean3 && std3 && H > 0 && W > 0 && S > 0, "image_prep: bad arguments (mean3/std3 are HOST pointers)");
  hipLaunchKernelGGL(image_prep_kernel, dim3((unsigned)min((int64_t)8192, (3 * S * S + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     img_hwc, out_chw, (int)H, (int)W, (int)S, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
  return msam2_check_launch("image_prep");
}

// ------------------------------------------------------------------------------------------------------------------
// SAM2Base._apply_non_overlapping_constraints (sam2_base.py:812-830): per pixel keep the highest-scoring object, clamp every
// other object's score to <= -10.  masks fp32 [n, P] (P = pixels), in -> out (may alias).
// ------------------------------------------------------------------------------------------------------------------
__global__ void non_overlap_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int64_t P) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
    int best = 0;
    float bv = x[i];
    for (int o = 1; o < n; ++o) {
      const float v = x[(int64_t)o * P + i];
      if (v > bv) { bv = v; best = o; }   // first maximum wins, like torch.argmax
    }
    for (int o = 0; o < n; ++o) {
      const float v = x[(int64_t)o * P + i];
      y[(int64_t)o * P + i] = (o == best) ? v : fminf(v, -10.0f);
    }
  }
}

extern "C" int msam2_non_overlap(const float* masks, float* out, int64_t n_obj, int64_t pixels, void* stream) {
  MSAM2_REQUIRE(masks && out && n_obj > 0 && pixels > 0, "non_overlap: bad arguments");
  hipLaunchKernelGGL(non_overlap_kernel, dim3((unsigned)min((int64_t)8192, (pixels + 255) / 256)), dim3(256), 0, (hipStream_t)stream, masks,
                     out, (int)n_obj, pixels);
  return msam2_check_launch("non_overlap");
}

// ------------------------------------------------------------------------------------------------------------------
// Segmentation-metric counts for eval_seg (func_3d/utils.py:139-214, func_2d/utils.py:505-580): for every threshold t, batch
// element b and class c, the integer counts  I = |pred > t  and  gt > t|,  P = |pred > t|,  G = |gt > t|  over the H*W pixels, in
// ONE pass over pred / gt (the reference thresholds, moves the maps to the CPU and reduces once per threshold).  IoU and Dice follow
// from the counts on the host (U = P + G - I).  counts: int32 [T, B*C, 3], zeroed by the caller.  T <= 8 thresholds per launch.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void seg_counts_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                         const float* __restrict__ thr, int T, int64_t P, int* __restrict__ counts,
                                                         int planes) {
  const int plane = blockIdx.y;
  float th[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) th[t] = t < T ? thr[t] : INFINITY;
  int ci[8], cp[8], cg[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) ci[t] = cp[t] = cg[t] = 0;
  const float* pp = pred + (int64_t)plane * P;
  const float* gp = gt + (int64_t)plane * P;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
    const float a = pp[i], g = gp[i];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int pa = a > th[t], pg = g > th[t];
      cp[t] += pa;
      cg[t] += pg;
      ci[t] += pa & pg;
    }
  }
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    if (t >= T) break;
    const int si = (int)wave_sum((float)ci[t]), sp = (int)wave_sum((float)cp[t]), sg = (int)wave_sum((float)cg[t]);  // < 2^24: exact
    if ((threadIdx.x & 63) == 0) {
      int* c = counts + ((int64_t)t * planes + plane) * 3;
      if (si) atomicAdd(c + 0, si);
      if (sp) atomicAdd(c + 1, sp);
      if (sg) atomicAdd(c + 2, sg);
    }
  }
}

extern "C" int msam2_seg_counts(const float* pred, const float* gt, const float* thresholds, int64_t n_thresholds, int64_t planes,
                                int64_t pixels, int* counts, void* stream) {
  MSAM2_REQUIRE(pred && gt && thresholds && counts, "seg_counts: null tensor");
  MSAM2_REQUIRE(n_thresholds >= 1 && n_thresholds <= 8 && planes > 0 && planes < 65536 && pixels > 0, "seg_counts: 1..8 thresholds per launch");
  const unsigned gx = (unsigned)min((int64_t)64, (pixels + 255) / 256);
  hipLaunchKernelGGL(seg_counts_kernel, dim3(gx, (unsigned)planes), dim3(256), 0, (hipStream_t)stream, pred, gt, thresholds,
                     (int)n_thresholds, pixels, counts, (int)planes);
  return msam2_check_launch("seg_counts");
}
