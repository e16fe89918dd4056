// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of libmsam2_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// 16-bit MFMA operand type of the whole library.  Default: IEEE fp16 (11-bit significand -> 8x finer operand rounding than
// bf16 at the same MFMA rate; every activation of this network sits well inside the fp16 range because residual streams,
// softmax statistics, LayerNorm and all logits stay fp32).  -DMSAM2_OPERAND_BF16 rebuilds the library on bf16 operands.
#if defined(MSAM2_OPERAND_BF16)
typedef __bf16 op16;
#define MSAM2_OPERAND_IS_FP16 0
#define MSAM2_MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#else
typedef _Float16 op16;
#define MSAM2_OPERAND_IS_FP16 1
#define MSAM2_MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#endif
typedef __attribute__((ext_vector_type(8))) op16 op16x8;
typedef __attribute__((ext_vector_type(4))) op16 op16x4;
typedef __attribute__((ext_vector_type(2))) op16 op16x2;
typedef __attribute__((ext_vector_type(4))) short short4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define MSAM2_OK 0
#define MSAM2_ERR_ARG -1
#define MSAM2_ERR_LAUNCH -2

// error string shared by every translation unit (api.hip owns it)
void msam2_set_error(const char* fmt, ...);
int msam2_check_launch(const char* what);

#define MSAM2_REQUIRE(cond, ...)          \
  do {                                    \
    if (!(cond)) {                        \
      msam2_set_error(__VA_ARGS__);       \
      return MSAM2_ERR_ARG;               \
    }                                     \
  } while (0)

__device__ __forceinline__ float op2f(op16 x) { return (float)x; }
// fp32 -> operand.  The fp16 build SATURATES at +-65504 (one v_med3_f32): a value beyond the fp16 range -- a real checkpoint's large
// MLP / qkv activation, an un-scaled gradient -- must not become inf and travel through softmax / LayerNorm / the optimiser state.
// Reductions across the two 32-lane halves of a wave (the row halves of a 32x32 MFMA accumulator): v_permlane32_swap_b32 (gfx950)
// exchanges the halves in the VALU -- __shfl_xor(v, 32) compiles to ds_bpermute_b32, an LDS round trip plus an lgkmcnt wait in the
// middle of every softmax tile.  Both results are identical in all 64 lanes and bit-equal to the shuffle forms (max, + commute).
__device__ __forceinline__ float half_max(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}
__device__ __forceinline__ float half_sum(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}

// NaN stays NaN.  f2op_fast is the plain conversion for values known to be bounded (softmax probabilities in [0, 1]).
__device__ __forceinline__ op16 f2op_fast(float x) { return (op16)x; }
#if MSAM2_OPERAND_IS_FP16 && !defined(MSAM2_NO_SATURATE)
__device__ __forceinline__ op16 f2op(float x) { return (op16)__builtin_amdgcn_fmed3f(x, -65504.f, 65504.f); }
#else
__device__ __forceinline__ op16 f2op(float x) { return (op16)x; }
#endif

// fp32 -> the output element type of a templated kernel (saturating for the fp16 operand type, identity for fp32)
template <typename T> __device__ __forceinline__ T f2out(float x);
template <> __device__ __forceinline__ float f2out<float>(float x) { return x; }
template <> __device__ __forceinline__ op16 f2out<op16>(float x) { return f2op(x); }

// exact-erf GELU (nn.GELU default).  erf by Abramowitz-Stegun 7.1.26 (|abs error| <= 1.5e-7, i.e. at fp32 round-off of the
// surrounding arithmetic and far below the op16 rounding of every consumer) -- ~4x fewer VALU operations than erff().
//   gelu(x) = 0.5 x (1 + erf(x / sqrt 2)),  erf(a) = sign(a) (1 - P(t) e^{-a^2}),  t = 1 / (1 + p |a|)
//           = max(x, 0) - 0.5 |x| P(t) e^{-a^2}        (both signs; no 1 - (1 - ..) cancellation on the negative side)
// Written on PAIRS: every multiply / fma of the polynomial is a packed fp32 instruction (v_pk_fma_f32 / v_pk_mul_f32: two elements per
// 4-cycle issue), only v_rcp_f32 and v_exp_f32 stay per element -- ~50 instead of ~72 issue cycles per element; the activation is
// 25-30 % of the fused MLP kernel and the whole epilogue of the fc1 GEMMs.  The scalar form is the same
// expression, so a value gets the same bits whichever form a kernel uses.
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
  const f32x2 ab = {fabsf(x[0]), fabsf(x[1])};
  const f32x2 ax = ab * 0.70710678118654752440f;
  const f32x2 d = 1.0f + 0.3275911f * ax;
  const f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  const f32x2 poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const f32x2 arg = ax * ax * (-1.44269504088896340736f);
  const f32x2 ex = {__builtin_amdgcn_exp2f(arg[0]), __builtin_amdgcn_exp2f(arg[1])};
  const f32x2 relu = {fmaxf(x[0], 0.f), fmaxf(x[1], 0.f)};
  return relu - (ab * 0.5f) * (poly * ex);
}
__device__ __forceinline__ float gelu_erf(float x) {
  const float ab = fabsf(x);
  const float ax = ab * 0.70710678118654752440f;
  const float d = 1.0f + 0.3275911f * ax;
  const float t = __builtin_amdgcn_rcpf(d);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float arg = ax * ax * (-1.44269504088896340736f);
  const float ex = __builtin_amdgcn_exp2f(arg);
  return fmaxf(x, 0.f) - (ab * 0.5f) * (poly * ex);
}
// erf itself (the GELU derivative of the backward pass)
__device__ __forceinline__ float fast_erf(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float e = 1.0f - poly * __expf(-ax * ax);
  return copysignf(e, x);
}

// Counter-based dropout mask shared by the element-wise dropout kernel (backward.hip) and the flash attention kernels (attention.hip,
// attention_bwd.hip): element idx of stream seed is KEPT iff the upper half of splitmix64(seed + idx * golden) is >= thr = p * 2^32.
__device__ __forceinline__ bool dropout_keep(uint64_t seed, uint64_t idx, unsigned thr) {
  uint64_t z = seed + idx * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (unsigned)(z >> 32) >= thr;
}
// attention-probability dropout of a flash kernel (F.scaled_dot_product_attention(dropout_p) in train mode, transformer.py:317-318):
// probability (b, h, q, k) uses element offset + ((b * H + h) * Lq + q) * Lk + k of the stream; thr == 0 switches it off.
struct AttnDropout {
  unsigned thr;
  float inv_keep;
  uint64_t seed, offset;
  const uint64_t* seed_dev;      // optional: added to seed on the device (msam2_counter_bump)
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
