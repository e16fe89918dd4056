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

// One 1-KiB LDS-DMA piece (buffer_load_dwordx4 ... offen lds: lane l's 16 bytes land at lds + 16 l), issued from INLINE ASM.
// Why not __builtin_amdgcn_raw_ptr_buffer_load_lds: hipcc tracks the builtin's LDS write on vmcnt and, before the next LDS read it cannot
// prove disjoint (every ds_read_b64_tr_b16 -- the intrinsic carries no alias information), inserts s_waitcnt vmcnt(0): the wave then
// sits out the full latency of the DMA it issued a few instructions earlier -- in every tile of a loop whose whole point is to stream
// tile t+2 under the MFMAs of tile t (found in the ISA of attn_kv64x2_kernel, round 3).  An asm statement is
// opaque to that bookkeeping; the kernels wait for their DMA themselves (s_waitcnt vmcnt(N) + s_barrier before the first read of a
// stage), which they did anyway.  M0 (the LDS base) is written in the statement that uses it and restored; s_nop 4 covers a
// just-written SGPR operand (descriptor / offsets) and the M0 write -> LDS-DMA hazard.  lds must be wave-uniform.
// -DMSAM2_DMA_BUILTIN keeps the builtin (A/B builds).
__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, const unsigned char* lds, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds), 16, voff, soff, 0, 0);
}
__device__ __forceinline__ void glds16_asm(__amdgpu_buffer_rsrc_t rsrc, const unsigned char* lds, unsigned voff, unsigned soff) {
#ifdef MSAM2_DMA_BUILTIN
  glds16(rsrc, lds, voff, soff);
#else
  const unsigned dst = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)(lds);
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 4\n\t"
      "buffer_load_dwordx4 %1, %2, %4 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(rsrc), "s"(dst), "s"(soff)
      : "memory");
#endif
}

// three consecutive pieces (LDS lds, lds + 1 KiB, lds + 2 KiB) in one statement: M0 saved / restored once
__device__ __forceinline__ void glds16x3_asm(__amdgpu_buffer_rsrc_t rsrc, const unsigned char* lds, unsigned v0, unsigned v1, unsigned v2,
                                             unsigned soff) {
#ifdef MSAM2_DMA_BUILTIN
  glds16(rsrc, lds, v0, soff);
  glds16(rsrc, lds + 1024, v1, soff);
  glds16(rsrc, lds + 2048, v2, soff);
#else
  const unsigned dst = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)(lds);
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %5\n\t"
      "s_nop 4\n\t"
      "buffer_load_dwordx4 %1, %4, %6 offen lds\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %2, %4, %6 offen lds\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %3, %4, %6 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(v0), "v"(v1), "v"(v2), "s"(rsrc), "s"(dst), "s"(soff)
      : "memory", "scc");
#endif
}

// exact-erf GELU (nn.GELU default): gelu(x) = x Phi(x), Phi(x) = 0.5 (1 + erf(x / sqrt 2)).
// Phi(x) - 0.5 is odd: x Q(x^2) with Q a degree-8 polynomial in x^2 (minimax fit of the GELU's own absolute error on |x| <= 4.5, end
// value pinned so that Phi(+-4.5) = 1 / 0; outside, x is clamped: Phi(-4.5) = 3.4e-6).  |gelu_poly - gelu| <= 5e-5 absolute over all x
// (tests/test_kernels_gpu.py::test_gelu_epilogue_accuracy; a tenth of the op16 rounding of the GEMM operands that produce x).
// No transcendental: one v_med3 per element, everything else packed fp32 (v_pk_mul_f32 / v_pk_fma_f32: two elements per issue) --
// ~26 issue cycles per element against ~48 for the Abramowitz-Stegun erf (v_rcp + v_exp, 8 cycles each, + 12 ops) that it replaces;
// the activation is the whole epilogue of the fc1 GEMMs (as long as their MFMAs at K = 384) and 25-30 % of the fused MLP kernel.
// The scalar form is the same fma chain, so a value gets the same bits whichever form a kernel uses.  -DMSAM2_GELU_AS keeps the old form.
typedef __attribute__((ext_vector_type(2))) float f32x2;
#ifndef MSAM2_GELU_AS
#define MSAM2_GELU_X 4.5f
#define MSAM2_GELU_Q0 3.987085521e-01f
#define MSAM2_GELU_Q1 -6.597723812e-02f
#define MSAM2_GELU_Q2 9.580635466e-03f
#define MSAM2_GELU_Q3 -1.036248752e-03f
#define MSAM2_GELU_Q4 8.120908024e-05f
#define MSAM2_GELU_Q5 -4.431632988e-06f
#define MSAM2_GELU_Q6 1.580232549e-07f
#define MSAM2_GELU_Q7 -3.283848526e-09f
#define MSAM2_GELU_Q8 2.999938145e-11f
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
  const f32x2 xc = {__builtin_amdgcn_fmed3f(x[0], -MSAM2_GELU_X, MSAM2_GELU_X), __builtin_amdgcn_fmed3f(x[1], -MSAM2_GELU_X, MSAM2_GELU_X)};
  const f32x2 u = xc * xc;
  auto bc = [](float c) { return f32x2{c, c}; };
  f32x2 q = __builtin_elementwise_fma(u, bc(MSAM2_GELU_Q8), bc(MSAM2_GELU_Q7));
  q = __builtin_elementwise_fma(q, u, bc(MSAM2_GELU_Q6));
  q = __builtin_elementwise_fma(q, u, bc(MSAM2_GELU_Q5));
  q = __builtin_elementwise_fma(q, u, bc(MSAM2_GELU_Q4));
  q = __builtin_elementwise_fma(q, u, bc(MSAM2_GELU_Q3));
  q = __builtin_elementwise_fma(q, u, bc(MSAM2_GELU_Q2));
  q = __builtin_elementwise_fma(q, u, bc(MSAM2_GELU_Q1));
  q = __builtin_elementwise_fma(q, u, bc(MSAM2_GELU_Q0));
  const f32x2 phi = __builtin_elementwise_fma(xc, q, bc(0.5f));
  return x * phi;
}
__device__ __forceinline__ float gelu_erf(float x) {
  const float xc = __builtin_amdgcn_fmed3f(x, -MSAM2_GELU_X, MSAM2_GELU_X);
  const float u = xc * xc;
  float q = __builtin_fmaf(u, MSAM2_GELU_Q8, MSAM2_GELU_Q7);
  q = __builtin_fmaf(q, u, MSAM2_GELU_Q6);
  q = __builtin_fmaf(q, u, MSAM2_GELU_Q5);
  q = __builtin_fmaf(q, u, MSAM2_GELU_Q4);
  q = __builtin_fmaf(q, u, MSAM2_GELU_Q3);
  q = __builtin_fmaf(q, u, MSAM2_GELU_Q2);
  q = __builtin_fmaf(q, u, MSAM2_GELU_Q1);
  q = __builtin_fmaf(q, u, MSAM2_GELU_Q0);
  return x * __builtin_fmaf(xc, q, 0.5f);
}
#endif
// The same function with erf by Abramowitz-Stegun 7.1.26 (|abs error| <= 1.5e-7: at fp32 round-off):  erf(a) = sign(a) (1 - P(t) e^{-a^2}),
// t = 1 / (1 + p |a|);  gelu(x) = max(x, 0) - 0.5 |x| P(t) e^{-a^2}   (both signs; no 1 - (1 - ..) cancellation on the negative side).
// For kernels with fp32 outputs that are nowhere near VALU-bound (LayerNorm + GELU of the mask decoder's up-scaling).
__device__ __forceinline__ float gelu_erf_as(float x) {
  const float ab = fabsf(x);
  const float ax = ab * 0.70710678118654752440f;
  const float d = 1.0f + 0.3275911f * ax;
  const float t = __builtin_amdgcn_rcpf(d);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float arg = ax * ax * (-1.44269504088896340736f);
  const float ex = __builtin_amdgcn_exp2f(arg);
  return fmaxf(x, 0.f) - (ab * 0.5f) * (poly * ex);
}
#ifdef MSAM2_GELU_AS
__device__ __forceinline__ float gelu_erf(float x) { return gelu_erf_as(x); }
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) { return f32x2{gelu_erf_as(x[0]), gelu_erf_as(x[1])}; }
#endif
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
