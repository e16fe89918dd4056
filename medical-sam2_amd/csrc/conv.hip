// Convolution-shaped pieces of the path (gfx950).  GEMM-shaped convs are lowered to im2col + msam2_gemm_bf16; the
// small-channel mask down-sampler and the depth-wise 7x7 are direct kernels with their LayerNorm2d(+GELU) fused.
#include "common.h"

// ------------------------------------------------------------------------------------------------------------------
// PatchEmbed (backbones/utils.py:84-95): Conv2d(3,E,k7,s4,p3) as im2col -> [B*(S/4)^2, 160] op16 (147 taps + zero pad),
// column order (c, ky, kx) to match weight.reshape(E, 147).
// ------------------------------------------------------------------------------------------------------------------
// one thread per 8 consecutive columns of a patch row (one 16-byte store; the 2-byte-per-thread version ran at 1.4 TB/s)
__global__ void im2col_patch_kernel(const float* __restrict__ img, op16* __restrict__ out, int B, int S) {
  const int So = S / 4;
  const int64_t total = (int64_t)B * So * So * 20;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int g = i % 20;
    int64_t t = i / 20;
    const int xo = t % So;
    t /= So;
    const int yo = t % So;
    const int b = t / So;
    op16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int col = g * 8 + e;
      float f = 0.f;
      if (col < 147) {
        const int c = col / 49, k = col % 49, ky = k / 7, kx = k % 7;
        const int y = yo * 4 - 3 + ky, x = xo * 4 - 3 + kx;
        if (y >= 0 && y < S && x >= 0 && x < S) f = img[(((int64_t)b * 3 + c) * S + y) * S + x];
      }
      v[e] = f2op(f);
    }
    *reinterpret_cast<op16x8*>(out + i * 8) = v;
  }
}

extern "C" int msam2_im2col_patch7x7s4(const float* img, void* out, int64_t B, int64_t S, void* stream) {
  MSAM2_REQUIRE(img && out && B > 0 && S > 0 && S % 4 == 0, "im2col_patch: bad arguments");
  MSAM2_REQUIRE(((uintptr_t)out & 15) == 0, "im2col_patch: output must be 16-byte aligned");
  const int64_t total = B * (S / 4) * (S / 4) * 20;
  hipLaunchKernelGGL(im2col_patch_kernel, dim3((unsigned)min((int64_t)16384, (total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, img, (op16*)out, (int)B, (int)S);
  return msam2_check_launch("im2col_patch7x7s4");
}

// ------------------------------------------------------------------------------------------------------------------
// PatchEmbed as ONE kernel (round 4): Conv2d(3, E, k7, s4, p3) + bias + position table, fp32 image in, fp32 tokens out, no im2col map.
// The two-launch form writes an 84 MB [tokens, 160] patch matrix (4 x 1024^2) with a gather kernel that runs at 1.8 TB/s and reads it
// back in the GEMM: 74 + 48 us in the step.  Here a workgroup stages the 3 x 7 image rows of a token-row segment (NTOK = 32 / 64 / 128
// tokens: 4 NTOK + 4 pixels per row) in LDS as 16-bit values and the MFMA A fragments are read straight from that image: the weight's
// reduction is RE-ORDERED to k' = (c * 7 + ky) * 8 + 1 + kx with a zero tap in FRONT of every run of seven (176 = 11 k-steps of 16;
// PatchEmbed._weight_perm), so the 8 consecutive k' of a fragment are 8 consecutive pixels of one (c, ky) row starting at the pixel
// before the token's window -- column 4 t of the staged row, whose column 0 is the 16-byte-aligned pixel 4 tx0 - 4: an aligned float4
// of the image becomes one 8-byte LDS write, a fragment two 8-byte reads (32 lanes x 8 B contiguous: conflict free).
// Wave w of the workgroup owns tokens 32 w .. 32 w + 31.  The permuted weight ([n][176], 34 KB at E = 96) is copied into LDS once per
// workgroup with a 368-byte row pitch (23 sixteen-byte chunks: odd, so the 16 rows of a ds_read_b128 lane group fall into 16 different
// bank groups) and its fragments are read per use -- in registers they cost 132 VGPRs and left one workgroup per CU; the workgroup walks
// segments with a grid stride.  What the first versions taught (tools/patch_embed_bench.py, 4 x 1024^2, E = 96; im2col + GEMM: 103 us):
//   * a rolled load -> convert -> ds_write staging loop pays the memory latency once per iteration: 105 us; all of a thread's image
//     loads issued at once: 93 us; the next segment's loads in flight under the current one's MFMAs and stores: no further change;
//   * 8-byte instead of 2-byte LDS writes: no change;
//   * the position table: loads and stores share ONE in-order counter (vmcnt), so a load issued behind a store waits for that store's
//     round trip -- as 48 interleaved load / add / store triples per lane the table cost 54 of 92 us; with every position value
//     loaded before the first store: **60 us** (E = 112: 72 against 112).  Loading them before the MFMAs as well costs 50 registers,
//     the second workgroup per CU, and gains nothing.
// Output straight from the accumulator layout: one dword per lane, 32 consecutive channels of a token = 128-byte row segments, bias
// and the position row (token index within its image: the batch-broadcast table of Hiera._get_pos_embed) added in the store.
// ------------------------------------------------------------------------------------------------------------------
template <int NT, int NTOK>
__global__ __launch_bounds__(256) void patch_embed_kernel(const float* __restrict__ img, const op16* __restrict__ wp, const float* __restrict__ bias,
                                                          const float* __restrict__ pos, float* __restrict__ out, int B, int S, int E) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int KS = 11, ROWS = 21;                       // k-steps of 16; (c, ky) rows of the staged image
  constexpr int nt = NTOK;                                // tokens per segment
  constexpr int P = 4 * nt + 8;                           // staged row pitch in pixels (16-bit each): 4 nt + 4 used, rows stay 16-byte aligned
  constexpr int WP = 368;                                 // LDS row pitch of the weight image in bytes
  constexpr int F4 = nt + 1;                              // aligned float4 loads per row: pixels 4 tx0 - 4 .. 4 tx0 + 4 nt - 1
  constexpr int TOTAL = ROWS * F4, ITERS = (TOTAL + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int r = lane & 31, h = lane >> 5;
  const int So = S / 4, segs_per_row = So / nt;
  unsigned char* swt = smem;                              // [NT * 32][WP]
  op16* simg = reinterpret_cast<op16*>(smem + NT * 32 * WP);
  for (int i = tid; i < NT * 32 * 22; i += 256) {         // 22 chunks of 16 bytes per 176-element row
    const int n = i / 22, c = i - n * 22;
    *reinterpret_cast<uint4*>(swt + n * WP + c * 16) = *reinterpret_cast<const uint4*>(wp + (int64_t)n * 176 + c * 8);
  }
  float bj[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) bj[j] = (j * 32 + r < E) ? bias[j * 32 + r] : 0.f;
  const int n_seg = B * So * segs_per_row;
  // the image loads of segment i + 1 are issued before segment i is computed and stored (ITERS float4 per thread)
  f32x4 v[ITERS];
  auto load_segment = [&](int seg) __attribute__((always_inline)) {
    const int sx = seg % segs_per_row, ty = (seg / segs_per_row) % So, b = seg / (segs_per_row * So);
    const int tx0 = sx * nt;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int i = tid + it * 256;
      const int row = i / F4, m = i - row * F4;
      const int c = row / 7, ky = row - c * 7;
      const int y = 4 * ty - 3 + ky, x = 4 * tx0 - 4 + 4 * m;
      v[it] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i < TOTAL && y >= 0 && y < S && x >= 0 && x < S) v[it] = *reinterpret_cast<const f32x4*>(img + (((int64_t)b * 3 + c) * S + y) * S + x);
    }
  };
  if ((int)blockIdx.x < n_seg) load_segment(blockIdx.x);
  for (int seg = blockIdx.x; seg < n_seg; seg += gridDim.x) {
    const int sx = seg % segs_per_row, ty = (seg / segs_per_row) % So, b = seg / (segs_per_row * So);
    const int tx0 = sx * nt;
    // ---- stage: LDS column j of row (c, ky) <-> pixel x = 4 tx0 - 4 + j, y = 4 ty - 3 + ky; outside the image: 0
    __syncthreads();                                      // every wave is done reading the previous segment's image (and the weights are in)
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int i = tid + it * 256;
      const int row = i / F4, m = i - row * F4;
      if (i < TOTAL) {
        op16x4 w4;
#pragma unroll
        for (int q = 0; q < 4; ++q) w4[q] = f2op(v[it][q]);
        *reinterpret_cast<op16x4*>(simg + row * P + 4 * m) = w4;
      }
    }
    __syncthreads();
    if (seg + (int)gridDim.x < n_seg) load_segment(seg + gridDim.x);      // in flight under this segment's MFMAs and stores
    if (wave * 32 < nt) {
      f32x16 acc[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
      const int t = wave * 32 + r;                        // this lane's token within the segment (A operand row)
      // all A fragments first (one LDS latency for the tile instead of one per k-step): the window of token t is columns 4 t .. 4 t + 7
      // = pixels 4 (tx0 + t) - 4 .. + 3, the first of which meets the zero tap
      op16x8 afr[KS];
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int R = 2 * s + h;                          // (c, ky) row of this half's 8 taps; row 21 does not exist: zero fragment
        if (R < ROWS) {                                   // (8-byte aligned only: two ds_read_b64)
          const op16* src = simg + R * P + 4 * t;
          const op16x4 lo = *reinterpret_cast<const op16x4*>(src), hi = *reinterpret_cast<const op16x4*>(src + 4);
          afr[s][0] = lo[0]; afr[s][1] = lo[1]; afr[s][2] = lo[2]; afr[s][3] = lo[3];
          afr[s][4] = hi[0]; afr[s][5] = hi[1]; afr[s][6] = hi[2]; afr[s][7] = hi[3];
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) afr[s][e] = (op16)0.f;
        }
      }
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const op16x8 af = afr[s];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const op16x8 wf = *reinterpret_cast<const op16x8*>(swt + (j * 32 + r) * WP + (s * 16 + h * 8) * 2);
          acc[j] = MSAM2_MFMA_32x32x16(af, wf, acc[j], 0, 0, 0);
        }
      }
      // ---- store: lane = channel, accumulator register e = token row (e & 3) + 8 (e >> 2) + 4 h
      const int64_t tok0 = ((int64_t)b * So + ty) * So + tx0 + wave * 32;      // first token of this wave, global
      const int64_t ptok0 = (int64_t)ty * So + tx0 + wave * 32;                // ... within its image (position table row)
      float pv[NT][16];                                   // every position value is loaded BEFORE the first store (header comment)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = j * 32 + r;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
          pv[j][e] = (pos && n < E) ? pos[(ptok0 + row) * E + n] : 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = j * 32 + r;
        if (n < E) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
            out[(tok0 + row) * E + n] = acc[j][e] + bj[j] + pv[j][e];
          }
        }
      }
    }
  }
#endif
}

// img fp32 [B, 3, S, S] -> out fp32 [B * (S/4)^2, E] = conv7x7s4p3(img) + bias (+ pos[token within image], pos fp32 [(S/4)^2, E] or null).
// w_perm: 16-bit [ceil(E / 32) * 32, 176], reduction order k' = (c * 7 + ky) * 8 + 1 + kx, zero at k' % 8 == 0 and in the padding rows / columns.
// Requires (S/4) % 32 == 0 and E <= 128.
extern "C" int msam2_patch_embed7x7s4(const float* img, const void* w_perm, const float* bias, const float* pos, float* out, int64_t B,
                                      int64_t S, int64_t E, void* stream) {
  MSAM2_REQUIRE(img && w_perm && bias && out && B > 0 && S > 0 && E > 0, "patch_embed: bad arguments");
  MSAM2_REQUIRE(S % 4 == 0 && (S / 4) % 32 == 0 && E <= 128, "patch_embed: needs (S / 4) %% 32 == 0 and E <= 128 (S=%lld E=%lld)", (long long)S, (long long)E);
  MSAM2_REQUIRE((((uintptr_t)img | (uintptr_t)w_perm) & 15) == 0, "patch_embed: image and weights must be 16-byte aligned");
  MSAM2_REQUIRE(B * (S / 4) * (S / 4) * E < (1ll << 40), "patch_embed: problem too large");
  const int So = (int)(S / 4);
  const int nt = So % 128 == 0 ? 128 : (So % 64 == 0 ? 64 : 32);     // tokens per segment: whole segments per token row
  const int n_seg = (int)B * So * (So / nt);
  const int NTn = (int)((E + 31) / 32);
  const int lds = NTn * 32 * 368 + 21 * (4 * nt + 8) * 2;
  const dim3 grid((unsigned)min(n_seg, 1024)), block(256);
#define PE_LAUNCH(NTV, TOKV) \
  do { \
    static bool attr_set = false; \
    if (!attr_set) {     /* (NT = 4, hiera_b+: 69 KB of dynamic LDS) */ \
      hipFuncSetAttribute((const void*)patch_embed_kernel<NTV, TOKV>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); \
      attr_set = true; \
    } \
    hipLaunchKernelGGL((patch_embed_kernel<NTV, TOKV>), grid, block, lds, (hipStream_t)stream, img, (const op16*)w_perm, bias, pos, out, (int)B, (int)S, (int)E); \
  } while (0)
#define PE_TOK(NTV) \
  do { \
    if (nt == 128) PE_LAUNCH(NTV, 128); \
    else if (nt == 64) PE_LAUNCH(NTV, 64); \
    else PE_LAUNCH(NTV, 32); \
  } while (0)
  switch (NTn) {
    case 1: PE_TOK(1); break;
    case 2: PE_TOK(2); break;
    case 3: PE_TOK(3); break;
    default: PE_TOK(4); break;
  }
#undef PE_TOK
#undef PE_LAUNCH
  return msam2_check_launch("patch_embed7x7s4");
}

// 3x3 / stride 2 / pad 1 im2col on NHWC op16: [B,H,W,C] -> [B*(H/2)*(W/2), ld], column order (ky, kx, c), zero fill
// up to ld (>= 9*C, multiple of 8).  One thread per 8-byte group of 4 channels (C % 4 == 0).
__global__ void im2col3x3s2_kernel(const op16* __restrict__ x, op16* __restrict__ out, int B, int H, int W, int C, int ld) {
  // 32-bit index arithmetic (the group count is checked on the host): as int64 the seven divisions / remainders per 8-byte group were the
  // kernel (round 4); the load is unconditional (clamped) and zeroed afterwards
  const unsigned Ho = H / 2, Wo = W / 2;
  const unsigned gpr = ld / 4;  // 4-element groups per output row
  const unsigned total = (unsigned)B * Ho * Wo * gpr;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const unsigned t0 = i / gpr, g = i - t0 * gpr;
    const unsigned t1 = t0 / Wo, xo = t0 - t1 * Wo;
    const unsigned b = t1 / Ho, yo = t1 - b * Ho;
    const unsigned col = g * 4;
    const unsigned k = min(col / (unsigned)C, 8u), c = col - (col / (unsigned)C) * C;
    const int y = 2 * (int)yo - 1 + (int)(k / 3), xx = 2 * (int)xo - 1 + (int)(k % 3);
    const bool in = col < 9u * C && y >= 0 && y < H && xx >= 0 && xx < W;
    const op16x4 ld4 = *reinterpret_cast<const op16x4*>(x + (((int64_t)b * H + min(max(y, 0), H - 1)) * W + min(max(xx, 0), W - 1)) * C + (col < 9u * C ? c : 0));
    const op16x4 zero = {f2op(0.f), f2op(0.f), f2op(0.f), f2op(0.f)};
    *reinterpret_cast<op16x4*>(out + (int64_t)i * 4) = in ? ld4 : zero;
  }
}

extern "C" int msam2_im2col3x3s2(const void* x, void* out, int64_t B, int64_t H, int64_t W, int64_t C, int64_t ld, void* stream) {
  MSAM2_REQUIRE(x && out && B > 0 && C > 0 && C % 4 == 0 && H % 2 == 0 && W % 2 == 0 && ld >= 9 * C && ld % 8 == 0,
                "im2col3x3s2: bad arguments");
  const int64_t total = B * (H / 2) * (W / 2) * (ld / 4);
  MSAM2_REQUIRE(total < (1ll << 31), "im2col3x3s2: more than 2^31 output groups");
  hipLaunchKernelGGL(im2col3x3s2_kernel, dim3((unsigned)min((int64_t)16384, (total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, (const op16*)x, (op16*)out, (int)B, (int)H, (int)W, (int)C, (int)ld);
  return msam2_check_launch("im2col3x3s2");
}

// ------------------------------------------------------------------------------------------------------------------
// MaskDownSampler stage (memory_encoder.py:37-54): Conv2d(CIN, COUT=4*CIN, k3, s2, p1) + LayerNorm2d(eps 1e-6) + GELU,
// one thread per output pixel holding all COUT channels.  When `mask_mode` != 0 the (single-channel fp32) input is the
// raw high-res mask logits and the scaled sigmoid / binarisation of sam2_base.py:686-696 is applied on the fly:
//   mode 1: sigmoid(x) * scale + bias      mode 2: (x > 0) * scale + bias
// weights: fp32 [COUT][CIN][3][3] (nn.Conv2d layout); output NHWC op16.
// ------------------------------------------------------------------------------------------------------------------
template <int CIN, int COUT, typename TI>
__global__ void conv3x3s2_ln_gelu_kernel(const TI* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                         const float* __restrict__ ln_w, const float* __restrict__ ln_b, op16* __restrict__ y,
                                         int B, int H, int W, int mask_mode, float mscale, float mbias) {
  __shared__ float ws[COUT * CIN * 9];
  for (int i = threadIdx.x; i < COUT * CIN * 9; i += blockDim.x) ws[i] = w[i];
  __syncthreads();
  const int Ho = H / 2, Wo = W / 2;
  const int64_t total = (int64_t)B * Ho * Wo;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int xo = i % Wo;
    int64_t t = i / Wo;
    const int yo = t % Ho;
    const int b = t / Ho;
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = bias[co];
    // (Round 4 tried the taps loaded unconditionally -- clamped, all nine in flight, predicated FMAs -- and one 8-byte store per pixel:
    //  19.3 -> 16.6 us at 4 x 1024^2.  The compiler orders that form's arithmetic differently: outputs moved by one 16-bit ulp in ~1e-4 of
    //  the rows, and the BPTT fixture's chained gradient -- whose decoder input gradient moves 8.5 % per 0.05 % of its input, DESIGN 7.2 --
    //  went from 0.8 % to 6 % off the reference's.  Both forms are within the kernel's own tolerance; the form the fixtures were validated
    //  with stays.)
    {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = 2 * yo - 1 + ky;
      if (yy < 0 || yy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int xx = 2 * xo - 1 + kx;
        if (xx < 0 || xx >= W) continue;
        const TI* px = x + (((int64_t)b * H + yy) * W + xx) * CIN;
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
          float v = (float)px[ci];
          if (mask_mode == 1) v = mscale / (1.f + __expf(-v)) + mbias;
          else if (mask_mode == 2) v = (v > 0.f ? mscale : 0.f) + mbias;
#pragma unroll
          for (int co = 0; co < COUT; ++co) acc[co] += v * ws[(co * CIN + ci) * 9 + ky * 3 + kx];
        }
      }
    }
    }
    float mean = 0.f;
#pragma unroll
    for (int co = 0; co < COUT; ++co) mean += acc[co];
    mean /= COUT;
    float var = 0.f;
#pragma unroll
    for (int co = 0; co < COUT; ++co) var += (acc[co] - mean) * (acc[co] - mean);
    const float rstd = 1.f / sqrtf(var / COUT + 1e-6f);
    op16* py = y + i * COUT;
#pragma unroll
    for (int co = 0; co < COUT; ++co) py[co] = f2op(gelu_erf((acc[co] - mean) * rstd * ln_w[co] + ln_b[co]));
  }
}

extern "C" int msam2_conv3x3s2_ln_gelu(const void* x, int in_is_16bit, const float* weight, const float* bias, const float* ln_w,
                                       const float* ln_b, void* y, int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout,
                                       int mask_mode, float mask_scale, float mask_bias, void* stream) {
  MSAM2_REQUIRE(x && weight && bias && ln_w && ln_b && y, "conv3x3s2_ln_gelu: null tensor");
  MSAM2_REQUIRE(H % 2 == 0 && W % 2 == 0 && B > 0, "conv3x3s2_ln_gelu: bad shape");
  MSAM2_REQUIRE(Cout == 4 * Cin && (Cin == 1 || Cin == 4 || Cin == 16), "conv3x3s2_ln_gelu: built for 1->4, 4->16, 16->64");
  MSAM2_REQUIRE(mask_mode == 0 || (Cin == 1 && !in_is_16bit), "conv3x3s2_ln_gelu: mask transform needs the fp32 1-channel input");
  MSAM2_REQUIRE(Cin == 1 ? !in_is_16bit : in_is_16bit, "conv3x3s2_ln_gelu: layer 1 takes fp32, later layers op16");
  const int64_t total = B * (H / 2) * (W / 2);
  dim3 grid((unsigned)min((int64_t)8192, (total + 127) / 128)), block(128);
  hipStream_t s = (hipStream_t)stream;
  if (Cin == 1)
    hipLaunchKernelGGL((conv3x3s2_ln_gelu_kernel<1, 4, float>), grid, block, 0, s, (const float*)x, weight, bias, ln_w, ln_b,
                       (op16*)y, (int)B, (int)H, (int)W, mask_mode, mask_scale, mask_bias);
  else if (Cin == 4)
    hipLaunchKernelGGL((conv3x3s2_ln_gelu_kernel<4, 16, op16>), grid, block, 0, s, (const op16*)x, weight, bias, ln_w, ln_b,
                       (op16*)y, (int)B, (int)H, (int)W, 0, 0.f, 0.f);
  else
    hipLaunchKernelGGL((conv3x3s2_ln_gelu_kernel<16, 64, op16>), grid, block, 0, s, (const op16*)x, weight, bias, ln_w, ln_b,
                       (op16*)y, (int)B, (int)H, (int)W, 0, 0.f, 0.f);
  return msam2_check_launch("conv3x3s2_ln_gelu");
}

// ------------------------------------------------------------------------------------------------------------------
// CXBlock head (memory_encoder.py:99-101): depth-wise Conv2d(C, C, k7, p3, groups=C) + LayerNorm2d(eps 1e-6) on NHWC
// fp32 -> normalised op16 (the A operand of pwconv1).  One wave per pixel, C = 256 -> 4 channels per lane.
// weights: fp32 [49][C] (tap-major, prepared from [C,1,7,7]).
// ------------------------------------------------------------------------------------------------------------------
// A wave owns DW_PIX horizontally adjacent pixels: per kernel row it loads the DW_PIX + 6 input columns and the 7 taps once and
// feeds all DW_PIX accumulators (4x fewer cache reads than one pixel per wave).
constexpr int DW_PIX = 4;
__global__ void dwconv7x7_ln_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                    const float* __restrict__ ln_w, const float* __restrict__ ln_b, op16* __restrict__ y, int B,
                                    int H, int W, int C) {
  const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const int gpr = (W + DW_PIX - 1) / DW_PIX;                   // pixel groups per image row
  if (grp >= (int64_t)B * H * gpr) return;
  const int x0 = (grp % gpr) * DW_PIX;
  const int yy = (grp / gpr) % H;
  const int b = grp / ((int64_t)gpr * H);
  const int c0 = lane * 4;
  const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + c0);
  f32x4 acc[DW_PIX];
#pragma unroll
  for (int p = 0; p < DW_PIX; ++p) acc[p] = bv;
  for (int ky = 0; ky < 7; ++ky) {
    const int y2 = yy - 3 + ky;
    if (y2 < 0 || y2 >= H) continue;
    f32x4 v[DW_PIX + 6], ww[7];
#pragma unroll
    for (int j = 0; j < DW_PIX + 6; ++j) {
      const int x2 = x0 - 3 + j;
      v[j] = (x2 >= 0 && x2 < W) ? *reinterpret_cast<const f32x4*>(x + (((int64_t)b * H + y2) * W + x2) * C + c0) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) ww[kx] = *reinterpret_cast<const f32x4*>(w + (ky * 7 + kx) * C + c0);
    // same tap order per pixel as the single-pixel form: ky outer, kx inner
#pragma unroll
    for (int kx = 0; kx < 7; ++kx)
#pragma unroll
      for (int p = 0; p < DW_PIX; ++p) acc[p] += v[p + kx] * ww[kx];
  }
#pragma unroll
  for (int p = 0; p < DW_PIX; ++p) {
    if (x0 + p >= W) break;
    const float mean = wave_sum(acc[p][0] + acc[p][1] + acc[p][2] + acc[p][3]) / C;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) q += (acc[p][e] - mean) * (acc[p][e] - mean);
    const float rstd = 1.f / sqrtf(wave_sum(q) / C + 1e-6f);
    op16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = f2op((acc[p][e] - mean) * rstd * ln_w[c0 + e] + ln_b[c0 + e]);
    const int64_t pix = ((int64_t)b * H + yy) * W + x0 + p;
    *reinterpret_cast<op16x4*>(y + pix * C + c0) = o;
  }
}

extern "C" int msam2_dwconv7x7_ln(const float* x, const float* weight_tap_major, const float* bias, const float* ln_w,
                                  const float* ln_b, void* y, int64_t B, int64_t H, int64_t W, int64_t C, void* stream) {
  MSAM2_REQUIRE(x && weight_tap_major && bias && ln_w && ln_b && y, "dwconv7x7_ln: null tensor");
  MSAM2_REQUIRE(C == 256, "dwconv7x7_ln: built for C=256 (one wave per pixel group, 4 channels per lane)");
  const int64_t groups = B * H * ((W + DW_PIX - 1) / DW_PIX);
  hipLaunchKernelGGL(dwconv7x7_ln_kernel, dim3(cdiv(groups * 64, 256)), dim3(256), 0, (hipStream_t)stream, x, weight_tap_major, bias,
                     ln_w, ln_b, (op16*)y, (int)B, (int)H, (int)W, (int)C);
  return msam2_check_launch("dwconv7x7_ln");
}

// ------------------------------------------------------------------------------------------------------------------
// ConvTranspose2d(k2, s2) tail (mask_decoder.py:244-247): the GEMM produced g[token(y,x)][(ky*2+kx)*C + co]; this kernel
// scatters it to pixel (2y+ky, 2x+kx), adds the conv bias and the high-res skip feature, then LayerNorm2d + GELU
// (first up-scaling) or GELU alone (second).  One wave per output pixel, lane = channel (C <= 64).  All NHWC.
// ------------------------------------------------------------------------------------------------------------------
__global__ void pixel_shuffle_kernel(const op16* __restrict__ g, const float* __restrict__ bias, const op16* __restrict__ skip,
                                     const float* __restrict__ ln_w, const float* __restrict__ ln_b, op16* __restrict__ y, int B,
                                     int h, int w, int C, int ppw) {
  // ppw output pixels per wave: 1 when LayerNorm needs the whole wave for one pixel's channels, 64 / C otherwise (C = 32 -> 2)
  const int64_t wv = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const int H = 2 * h, W = 2 * w;
  const int sub_px = ppw > 1 ? lane / C : 0;
  const int ch = ppw > 1 ? lane % C : lane;
  const int64_t pix = wv * ppw + sub_px;
  if (pix >= (int64_t)B * H * W) return;
  const int X = pix % W;
  const int Y = (pix / W) % H;
  const int b = pix / ((int64_t)W * H);
  const int64_t tok = ((int64_t)b * h + Y / 2) * w + X / 2;
  const int sub = (Y & 1) * 2 + (X & 1);
  float v = 0.f;
  if (ch < C) v = op2f(g[tok * 4 * C + sub * C + ch]) + bias[ch] + op2f(skip[pix * C + ch]);
  if (ln_w) {
    const float mean = wave_sum(ch < C ? v : 0.f) / C;
    const float d = ch < C ? v - mean : 0.f;
    const float rstd = 1.f / sqrtf(wave_sum(d * d) / C + 1e-6f);
    v = d * rstd * (ch < C ? ln_w[ch] : 0.f) + (ch < C ? ln_b[ch] : 0.f);
  }
  if (ch < C) y[pix * C + ch] = f2op(gelu_erf(v));
}

// Eight channels per lane (16-byte loads / stores), C / 8 lanes per output pixel, 512 / C pixels per wave: the one-channel-per-lane form
// above moves 128 bytes per wave instruction (32 us per call at the decoder's shapes); LayerNorm statistics are reduced over the
// pixel's C / 8 lanes with xor-shuffles.  C in {32, 64}.
template <int C, typename TS>
__global__ void pixel_shuffle8_kernel(const op16* __restrict__ g, const float* __restrict__ bias, const TS* __restrict__ skip,
                                      const float* __restrict__ ln_w, const float* __restrict__ ln_b, op16* __restrict__ y, int B, int h,
                                      int w) {
  constexpr int G = C / 8;                                   // lanes per pixel
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int H = 2 * h, W = 2 * w;
  const int64_t pix = gid / G;
  const int c0 = (int)(gid % G) * 8;
  const bool live = pix < (int64_t)B * H * W;
  const int64_t pp = live ? pix : 0;
  const int X = pp % W;
  const int Y = (pp / W) % H;
  const int b = pp / ((int64_t)W * H);
  const int64_t tok = ((int64_t)b * h + Y / 2) * w + X / 2;
  const int sub = (Y & 1) * 2 + (X & 1);
  const op16x8 gv = *reinterpret_cast<const op16x8*>(g + tok * 4 * C + sub * C + c0);
  float sk[8];
  if constexpr (sizeof(TS) == 4) {                            // fp32 skip (the FPN's own output type): no 16-bit copy pass in front of this kernel
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(skip + pp * C + c0), s1 = *reinterpret_cast<const f32x4*>(skip + pp * C + c0 + 4);
#pragma unroll
    for (int e = 0; e < 8; ++e) sk[e] = e < 4 ? s0[e] : s1[e - 4];
  } else {
    const op16x8 sv = *reinterpret_cast<const op16x8*>(skip + pp * C + c0);
#pragma unroll
    for (int e = 0; e < 8; ++e) sk[e] = op2f(sv[e]);
  }
  const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + c0), b1 = *reinterpret_cast<const f32x4*>(bias + c0 + 4);
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = op2f(gv[e]) + (e < 4 ? b0[e] : b1[e - 4]) + sk[e];
  if (ln_w) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s += v[e];
#pragma unroll
    for (int o = 1; o < G; o <<= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / C;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v[e] -= mean;
      q += v[e] * v[e];
    }
#pragma unroll
    for (int o = 1; o < G; o <<= 1) q += __shfl_xor(q, o, 64);
    const float rstd = 1.f / sqrtf(q / C + 1e-6f);
    const f32x4 w0 = *reinterpret_cast<const f32x4*>(ln_w + c0), w1 = *reinterpret_cast<const f32x4*>(ln_w + c0 + 4);
    const f32x4 l0 = *reinterpret_cast<const f32x4*>(ln_b + c0), l1 = *reinterpret_cast<const f32x4*>(ln_b + c0 + 4);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = v[e] * rstd * (e < 4 ? w0[e] : w1[e - 4]) + (e < 4 ? l0[e] : l1[e - 4]);
  }
  if (!live) return;
  op16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = f2op(gelu_erf(v[e]));
  *reinterpret_cast<op16x8*>(y + pix * C + c0) = o;
}

template <typename TS>
static int convt2x2_shuffle_launch(const void* gemm_out, const float* bias, const TS* skip, const float* ln_w, const float* ln_b, void* y,
                                   int64_t B, int64_t h, int64_t w, int64_t C, void* stream) {
  MSAM2_REQUIRE(gemm_out && bias && skip && y, "convt2x2_shuffle: null tensor");
  MSAM2_REQUIRE(C > 0 && C <= 64, "convt2x2_shuffle: C must be <= 64");
  const int64_t pix = B * 4 * h * w;
  const bool al = (((uintptr_t)gemm_out | (uintptr_t)skip | (uintptr_t)y | (uintptr_t)bias | (uintptr_t)ln_w | (uintptr_t)ln_b) & 15) == 0;
  if (al && (C == 64 || C == 32)) {
    const int64_t threads = pix * (C / 8);
    if (C == 64)
      hipLaunchKernelGGL((pixel_shuffle8_kernel<64, TS>), dim3(cdiv(threads, 256)), dim3(256), 0, (hipStream_t)stream, (const op16*)gemm_out, bias,
                         skip, ln_w, ln_b, (op16*)y, (int)B, (int)h, (int)w);
    else
      hipLaunchKernelGGL((pixel_shuffle8_kernel<32, TS>), dim3(cdiv(threads, 256)), dim3(256), 0, (hipStream_t)stream, (const op16*)gemm_out, bias,
                         skip, ln_w, ln_b, (op16*)y, (int)B, (int)h, (int)w);
    return msam2_check_launch("convt2x2_shuffle");
  }
  MSAM2_REQUIRE(sizeof(TS) == 2, "convt2x2_shuffle: the fp32-skip form needs C = 32 / 64 and 16-byte aligned tensors");
  const int ppw = (!ln_w && C <= 32 && 64 % C == 0) ? (int)(64 / C) : 1;
  hipLaunchKernelGGL(pixel_shuffle_kernel, dim3(cdiv(cdiv(pix, ppw) * 64, 256)), dim3(256), 0, (hipStream_t)stream, (const op16*)gemm_out,
                     bias, (const op16*)skip, ln_w, ln_b, (op16*)y, (int)B, (int)h, (int)w, (int)C, ppw);
  return msam2_check_launch("convt2x2_shuffle");
}

extern "C" int msam2_convt2x2_shuffle(const void* gemm_out, const float* bias, const void* skip, const float* ln_w,
                                      const float* ln_b, void* y, int64_t B, int64_t h, int64_t w, int64_t C, void* stream) {
  return convt2x2_shuffle_launch<op16>(gemm_out, bias, (const op16*)skip, ln_w, ln_b, y, B, h, w, C, stream);
}

// the same with the high-resolution skip features in fp32 (as the FPN returns them): C = 32 / 64
extern "C" int msam2_convt2x2_shuffle_f32skip(const void* gemm_out, const float* bias, const float* skip, const float* ln_w,
                                              const float* ln_b, void* y, int64_t B, int64_t h, int64_t w, int64_t C, void* stream) {
  return convt2x2_shuffle_launch<float>(gemm_out, bias, skip, ln_w, ln_b, y, B, h, w, C, stream);
}

// masks[n, k, p] = sum_c hyper[n, k, c] * up[n, p, c]   (mask_decoder.py:249-256), C = 32, K mask tokens; fp32 out
__global__ void hyper_masks_kernel(const float* __restrict__ hyper, const op16* __restrict__ up, float* __restrict__ masks, int n,
                                   int K, int P, int C) {
  __shared__ float hs[8 * 32];
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < K * C; i += blockDim.x) hs[i] = hyper[(int64_t)b * K * C + i];
  __syncthreads();
  for (int64_t pidx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pidx < P; pidx += (int64_t)gridDim.x * blockDim.x) {
    float u[32];
    const op16* pu = up + ((int64_t)b * P + pidx) * C;
#pragma unroll
    for (int c = 0; c < 32; c += 8) {
      const op16x8 t = *reinterpret_cast<const op16x8*>(pu + c);
#pragma unroll
      for (int e = 0; e < 8; ++e) u[c + e] = op2f(t[e]);
    }
    for (int k = 0; k < K; ++k) {
      float a = 0.f;
#pragma unroll
      for (int c = 0; c < 32; ++c) a += hs[k * 32 + c] * u[c];
      masks[((int64_t)b * K + k) * P + pidx] = a;
    }
  }
}

extern "C" int msam2_hyper_masks(const float* hyper, const void* upscaled, float* masks, int64_t n, int64_t K, int64_t P, int64_t C,
                                 void* stream) {
  MSAM2_REQUIRE(hyper && upscaled && masks, "hyper_masks: null tensor");
  MSAM2_REQUIRE(C == 32 && K > 0 && K <= 8 && n > 0 && P > 0, "hyper_masks: built for C=32, K<=8");
  hipLaunchKernelGGL(hyper_masks_kernel, dim3(cdiv(P, 256), (unsigned)n), dim3(256), 0, (hipStream_t)stream, hyper,
                     (const op16*)upscaled, masks, (int)n, (int)K, (int)P, (int)C);
  return msam2_check_launch("hyper_masks");
}

// ------------------------------------------------------------------------------------------------------------------
// Prompt embeddings (prompt_encoder.py:79-114; position_encoding.py:130-158): random-Fourier PE of (x+0.5, y+0.5)/S plus
// the per-label learned vector; label -1 -> not_a_point_embed only.  One block per point, thread = channel pair.
// tables: point_emb [4][C], not_a_point [C], gauss [2][C/2]; out fp32 [n_points_total, C]
// ------------------------------------------------------------------------------------------------------------------
__global__ void prompt_points_kernel(const float* __restrict__ xy, const int* __restrict__ labels, const float* __restrict__ gauss,
                                     const float* __restrict__ point_emb, const float* __restrict__ not_a_point,
                                     float* __restrict__ out, int n_out, int p_in, int p_out, int C, float inv_size) {
  const int pt = blockIdx.x;
  if (pt >= n_out) return;
  const int nf = C / 2;
  const int set = pt / p_out, j = pt - set * p_out;
  const bool pad = j >= p_in;                              // the padding point of prompt_encoder.py:87-91: (0, 0) with label -1
  const int src = set * p_in + j;
  const int lab = pad ? -1 : labels[src];
  const float px = pad ? 0.f : xy[2 * src], py = pad ? 0.f : xy[2 * src + 1];
  const float cx = 2.f * ((px + 0.5f) * inv_size) - 1.f, cy = 2.f * ((py + 0.5f) * inv_size) - 1.f;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const int f = c % nf;
    const float a = 6.283185307179586f * (cx * gauss[f] + cy * gauss[nf + f]);
    float v = (c < nf) ? sinf(a) : cosf(a);
    if (lab == -1) v = not_a_point[c];
    else if (lab >= 0 && lab < 4) v += point_emb[lab * C + c];
    out[(int64_t)pt * C + c] = v;
  }
}

extern "C" int msam2_prompt_points(const float* xy, const int* labels, const float* gauss, const float* point_emb,
                                   const float* not_a_point, float* out, int64_t n_points, int64_t C, float image_size,
                                   void* stream) {
  MSAM2_REQUIRE(xy && labels && gauss && point_emb && not_a_point && out, "prompt_points: null tensor");
  MSAM2_REQUIRE(n_points > 0 && C % 2 == 0, "prompt_points: bad shape");
  hipLaunchKernelGGL(prompt_points_kernel, dim3((unsigned)n_points), dim3(128), 0, (hipStream_t)stream, xy, labels, gauss,
                     point_emb, not_a_point, out, (int)n_points, (int)n_points, (int)n_points, (int)C, 1.0f / image_size);
  return msam2_check_launch("prompt_points");
}

// the same with `n_pad` padding points appended to each of the `n_sets` prompt sets inside the kernel (no zeros / full / cat launches):
// xy [n_sets, P, 2], labels [n_sets, P] -> out [n_sets, P + n_pad, C]
extern "C" int msam2_prompt_points_padded(const float* xy, const int* labels, const float* gauss, const float* point_emb,
                                          const float* not_a_point, float* out, int64_t n_sets, int64_t P, int64_t n_pad, int64_t C,
                                          float image_size, void* stream) {
  MSAM2_REQUIRE(xy && labels && gauss && point_emb && not_a_point && out, "prompt_points_padded: null tensor");
  MSAM2_REQUIRE(n_sets > 0 && P > 0 && n_pad >= 0 && C % 2 == 0 && n_sets * (P + n_pad) < (1ll << 31), "prompt_points_padded: bad shape");
  hipLaunchKernelGGL(prompt_points_kernel, dim3((unsigned)(n_sets * (P + n_pad))), dim3(128), 0, (hipStream_t)stream, xy, labels, gauss,
                     point_emb, not_a_point, out, (int)(n_sets * (P + n_pad)), (int)P, (int)(P + n_pad), (int)C, 1.0f / image_size);
  return msam2_check_launch("prompt_points_padded");
}

// ------------------------------------------------------------------------------------------------------------------
// Mask selection (sam2_base.py:354-385, mask_decoder.py:147-168,269-317) without a host round trip.
//   multimask=1: candidates are tokens 1..3, best = argmax IoU
//   multimask=0: token 0 unless its stability score < thresh, then best-IoU of tokens 1..3 (dynamic multimask, eval)
// gated by the object score (<= 0 -> NO_OBJ_SCORE).  Writes low_res [n,1,h,w], sel[n] (chosen token index 0..3),
// iou_sel[n].  masks: fp32 [n,4,h,w]; ious: [n,4]; obj: [n].
// ------------------------------------------------------------------------------------------------------------------
// grid (n, SELECT_SPLIT): every workgroup of an image evaluates the stability score itself (the token-0 mask is 256 KB and
// L2-resident; 16-byte loads, 8 in flight per lane) and then copies its 1/SELECT_SPLIT share of the chosen mask -- one launch,
// no workspace, and the latency of a single 256-thread sweep over the image no longer bounds the step.
constexpr int SELECT_SPLIT = 8;
__global__ __launch_bounds__(256) void select_mask_kernel(const float* __restrict__ masks, const float* __restrict__ ious,
                                                          const float* __restrict__ obj, float* __restrict__ low, int* __restrict__ sel,
                                                          float* __restrict__ iou_sel, int P, int multimask, int dynamic, float delta,
                                                          float thresh) {
  __shared__ float red_i[4], red_u[4];
  const int b = blockIdx.x;
  const float* m = masks + (int64_t)b * 4 * P;
  int best = 1;
  float bi = ious[b * 4 + 1];
  for (int k = 2; k < 4; ++k)
    if (ious[b * 4 + k] > bi) { bi = ious[b * 4 + k]; best = k; }
  int choice = best;
  if (!multimask) {
    choice = 0;
    if (dynamic) {
      float ai = 0.f, au = 0.f;
      const int P4 = P >> 2;   // P is a multiple of 4 on this path (checked on the host)
      const f32x4* m4 = reinterpret_cast<const f32x4*>(m);
      for (int i0 = threadIdx.x; i0 < P4; i0 += 8 * 256) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = i0 + u * 256;
          v[u] = i < P4 ? m4[i] : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            ai += v[u][q] > delta ? 1.f : 0.f;
            au += v[u][q] > -delta ? 1.f : 0.f;
          }
      }
      ai = wave_sum(ai);
      au = wave_sum(au);
      if ((threadIdx.x & 63) == 0) { red_i[threadIdx.x >> 6] = ai; red_u[threadIdx.x >> 6] = au; }
      __syncthreads();
      const float ti = red_i[0] + red_i[1] + red_i[2] + red_i[3], tu = red_u[0] + red_u[1] + red_u[2] + red_u[3];
      const float stab = tu > 0.f ? ti / tu : 1.f;
      if (!(stab >= thresh)) choice = best;
    }
  }
  if (threadIdx.x == 0 && blockIdx.y == 0) {
    sel[b] = choice;
    iou_sel[b] = ious[b * 4 + choice];
  }
  const bool appearing = obj[b] > 0.f;
  const float* src = m + (int64_t)choice * P;
  const int per = (P + SELECT_SPLIT - 1) / SELECT_SPLIT;
  const int lo = blockIdx.y * per, hi = min(P, lo + per);
  for (int i = lo + threadIdx.x; i < hi; i += 256) low[(int64_t)b * P + i] = appearing ? src[i] : -1024.0f;
}

extern "C" int msam2_select_mask(const float* masks, const float* ious, const float* obj_scores, float* low_res, int* sel,
                                 float* iou_sel, int64_t n, int64_t P, int multimask, int dynamic_stability, float delta,
                                 float thresh, void* stream) {
  MSAM2_REQUIRE(masks && ious && obj_scores && low_res && sel && iou_sel && n > 0 && P > 0, "select_mask: bad arguments");
  MSAM2_REQUIRE(P % 4 == 0 && ((uintptr_t)masks & 15) == 0, "select_mask: masks must be 16-byte aligned with H*W %% 4 == 0");
  hipLaunchKernelGGL(select_mask_kernel, dim3((unsigned)n, SELECT_SPLIT), dim3(256), 0, (hipStream_t)stream, masks, ious, obj_scores,
                     low_res, sel, iou_sel, (int)P, multimask, dynamic_stability, delta, thresh);
  return msam2_check_launch("select_mask");
}

// Gather row sel[b] (+offset) of a [n, T, C] tensor -> [n, C]  (sam2_base.py:375-383 token pick), fp32
__global__ void gather_rows_kernel(const float* __restrict__ x, const int* __restrict__ sel, float* __restrict__ y, int T, int C,
                                   int offset) {
  const int b = blockIdx.x;
  const int row = (sel ? sel[b] : 0) + offset;
  for (int c = threadIdx.x; c < C; c += blockDim.x) y[(int64_t)b * C + c] = x[((int64_t)b * T + row) * C + c];
}

extern "C" int msam2_gather_rows(const float* x, const int* sel, float* y, int64_t n, int64_t T, int64_t C, int64_t offset,
                                 void* stream) {
  MSAM2_REQUIRE(x && y && n > 0 && T > 0 && C > 0, "gather_rows: bad arguments");
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, x, sel, y, (int)T, (int)C, (int)offset);
  return msam2_check_launch("gather_rows");
}

// obj_ptr = lam * ptr + (1 - lam) * no_obj_ptr, lam = obj_score > 0   (sam2_base.py:389-400, fixed_no_obj_ptr)
__global__ void obj_ptr_mix_kernel(float* __restrict__ ptr, const float* __restrict__ obj, const float* __restrict__ no_obj_ptr,
                                   int C) {
  const int b = blockIdx.x;
  const bool app = obj[b] > 0.f;
  for (int c = threadIdx.x; c < C; c += blockDim.x)
    if (!app) ptr[(int64_t)b * C + c] = no_obj_ptr[c];
}

extern "C" int msam2_obj_ptr_mix(float* ptr, const float* obj_scores, const float* no_obj_ptr, int64_t n, int64_t C, void* stream) {
  MSAM2_REQUIRE(ptr && obj_scores && no_obj_ptr && n > 0, "obj_ptr_mix: bad arguments");
  hipLaunchKernelGGL(obj_ptr_mix_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, ptr, obj_scores, no_obj_ptr, (int)C);
  return msam2_check_launch("obj_ptr_mix");
}

// ------------------------------------------------------------------------------------------------------------------
// Non-overlapping k x k / stride k patches (PromptEncoder.mask_downscaling convs k2 s2, prompt_encoder.py:58-66; and
// SAM2Base.mask_downsample k4 s4, sam2_base.py:108): NHWC [B,H,W,C] -> op16 [B*(H/k)*(W/k), ld] with columns
// (ky, kx, c) and zero fill up to ld (>= k*k*C, multiple of 8 for the GEMM).
// ------------------------------------------------------------------------------------------------------------------
template <typename TI>
__global__ void space_to_depth_kernel(const TI* __restrict__ x, op16* __restrict__ out, int B, int H, int W, int C, int k, int ld) {
  const int Ho = H / k, Wo = W / k;
  const int64_t total = (int64_t)B * Ho * Wo * ld;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int col = i % ld;
    int64_t t = i / ld;
    const int xo = t % Wo;
    t /= Wo;
    const int yo = t % Ho;
    const int b = t / Ho;
    float v = 0.f;
    if (col < k * k * C) {
      const int c = col % C, kk = col / C, ky = kk / k, kx = kk % k;
      v = (float)x[(((int64_t)b * H + yo * k + ky) * W + xo * k + kx) * C + c];
    }
    out[i] = f2op(v);
  }
}

extern "C" int msam2_space_to_depth(const void* x, int in_is_16bit, void* out, int64_t B, int64_t H, int64_t W, int64_t C, int64_t k,
                                    int64_t ld, void* stream) {
  MSAM2_REQUIRE(x && out && B > 0 && C > 0 && k > 0 && H % k == 0 && W % k == 0 && ld >= k * k * C && ld % 8 == 0,
                "space_to_depth: bad arguments");
  const int64_t total = B * (H / k) * (W / k) * ld;
  dim3 grid((unsigned)min((int64_t)8192, (total + 255) / 256)), block(256);
  if (in_is_16bit)
    hipLaunchKernelGGL((space_to_depth_kernel<op16>), grid, block, 0, (hipStream_t)stream, (const op16*)x, (op16*)out, (int)B, (int)H,
                       (int)W, (int)C, (int)k, (int)ld);
  else
    hipLaunchKernelGGL((space_to_depth_kernel<float>), grid, block, 0, (hipStream_t)stream, (const float*)x, (op16*)out, (int)B, (int)H,
                       (int)W, (int)C, (int)k, (int)ld);
  return msam2_check_launch("space_to_depth");
}

// ------------------------------------------------------------------------------------------------------------------
// The mask decoder's token heads in ONE launch (mask_decoder.py:249-266): G independent 3-layer MLPs
// (sam2_utils.py:108-132: Linear-ReLU-Linear-ReLU-Linear, optional sigmoid) of width C = 256, each applied to one token of
// every batch element -- the 4 hyper-network MLPs (tokens 2..5 -> 32 channels), the IoU head (token 1 -> 4, sigmoid) and the
// object-score head (token 0 -> 1).  As separate GEMM calls these are ~40 launches of a few microseconds each.
// grid (G, B), 1024 threads: a wave owns 16 output rows of a layer (8 row loads in flight: the kernel is pure load latency); each row is one coalesced 512-byte read (lane = 4 k's)
// followed by a wave reduction; activations stay in LDS in fp32.
//   hs fp32 [B, T, C]; tok[g] = token index; w1/w2 16-bit [G, C, C]; w3 16-bit [G, C(out rows, zero padded), C]; b* fp32 [G, C];
//   out fp32 [B, G, C] (first out_dim[g] entries of each row valid).
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void token_mlp3_kernel(const float* __restrict__ hs, int64_t hs_bs, int64_t hs_ts,
                                                          const int* __restrict__ tok, const op16* __restrict__ w1,
                                                          const float* __restrict__ b1, const op16* __restrict__ w2,
                                                          const float* __restrict__ b2, const op16* __restrict__ w3,
                                                          const float* __restrict__ b3, const int* __restrict__ out_dim,
                                                          const int* __restrict__ sigmoid, float* __restrict__ out, int B,
                                                          const int* __restrict__ out_off, const int* __restrict__ out_ld) {
  constexpr int C = 256, RPW = 16;                            // 16 waves x 16 rows per layer, 8 rows (8 loads) in flight per wave
  __shared__ float act[2][C];
  const int g = blockIdx.x, b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x < C) act[0][threadIdx.x] = hs[(int64_t)b * hs_bs + (int64_t)tok[g] * hs_ts + threadIdx.x];
  __syncthreads();
  const op16* ws[3] = {w1 + (int64_t)g * C * C, w2 + (int64_t)g * C * C, w3 + (int64_t)g * C * C};
  const float* bs[3] = {b1 + g * C, b2 + g * C, b3 + g * C};
  const int n_out = out_dim[g];
  // packed output (out_off != nullptr): head g writes its n_out values of batch element b at out[out_off[g] + b * out_ld[g] + o], so that
  // heads of different widths land in contiguous tensors of their own (no slicing copies behind the kernel)
  float* orow = out_off ? out + out_off[g] + (int64_t)b * out_ld[g] : out + ((int64_t)b * gridDim.x + g) * C;
#pragma unroll
  for (int layer = 0; layer < 3; ++layer) {
    const float* xin = act[layer & 1];
    const f32x4 xv = *reinterpret_cast<const f32x4*>(xin + lane * 4);
    const int rows = layer == 2 ? n_out : C;
#pragma unroll
    for (int o0 = wave * RPW; o0 < wave * RPW + RPW; o0 += 8) {
      if (o0 >= rows) break;
      op16x4 wv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) wv[u] = *reinterpret_cast<const op16x4*>(ws[layer] + (int64_t)(o0 + u) * C + lane * 4);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        float s = op2f(wv[u][0]) * xv[0] + op2f(wv[u][1]) * xv[1] + op2f(wv[u][2]) * xv[2] + op2f(wv[u][3]) * xv[3];
        s = wave_sum(s) + bs[layer][o0 + u];
        if (lane == 0) {
          if (layer < 2) act[(layer + 1) & 1][o0 + u] = fmaxf(s, 0.f);
          else if (o0 + u < n_out) orow[o0 + u] = sigmoid[g] ? 1.f / (1.f + __expf(-s)) : s;
        }
      }
    }
    __syncthreads();
  }
}

static int token_mlp3_launch(const float* hs, int64_t hs_batch_stride, int64_t hs_token_stride, const int* token_index, const void* w1,
                             const float* b1, const void* w2, const float* b2, const void* w3, const float* b3, const int* out_dim,
                             const int* sigmoid_flag, float* out, const int* out_off, const int* out_ld, int64_t G, int64_t B, int64_t C,
                             void* stream) {
  MSAM2_REQUIRE(hs && token_index && w1 && b1 && w2 && b2 && w3 && b3 && out_dim && sigmoid_flag && out, "token_mlp3: null tensor");
  MSAM2_REQUIRE(C == 256 && G > 0 && B > 0 && G < 65536 && B < 65536, "token_mlp3: built for width 256");
  hipLaunchKernelGGL(token_mlp3_kernel, dim3((unsigned)G, (unsigned)B), dim3(1024), 0, (hipStream_t)stream, hs, hs_batch_stride,
                     hs_token_stride, token_index, (const op16*)w1, b1, (const op16*)w2, b2, (const op16*)w3, b3, out_dim, sigmoid_flag,
                     out, (int)B, out_off, out_ld);
  return msam2_check_launch("token_mlp3");
}

extern "C" int msam2_token_mlp3(const float* hs, int64_t hs_batch_stride, int64_t hs_token_stride, const int* token_index,
                                const void* w1, const float* b1, const void* w2, const float* b2, const void* w3, const float* b3,
                                const int* out_dim, const int* sigmoid_flag, float* out, int64_t G, int64_t B, int64_t C, void* stream) {
  return token_mlp3_launch(hs, hs_batch_stride, hs_token_stride, token_index, w1, b1, w2, b2, w3, b3, out_dim, sigmoid_flag, out, nullptr,
                           nullptr, G, B, C, stream);
}

// the same with a packed output: head g writes out[out_offset[g] + b * out_stride[g] + o], o < out_dim[g] (int32 device arrays of G entries)
extern "C" int msam2_token_mlp3_packed(const float* hs, int64_t hs_batch_stride, int64_t hs_token_stride, const int* token_index,
                                       const void* w1, const float* b1, const void* w2, const float* b2, const void* w3, const float* b3,
                                       const int* out_dim, const int* sigmoid_flag, float* out, const int* out_offset, const int* out_stride,
                                       int64_t G, int64_t B, int64_t C, void* stream) {
  MSAM2_REQUIRE(out_offset && out_stride, "token_mlp3_packed: null offset / stride table");
  return token_mlp3_launch(hs, hs_batch_stride, hs_token_stride, token_index, w1, b1, w2, b2, w3, b3, out_dim, sigmoid_flag, out, out_offset,
                           out_stride, G, B, C, stream);
}
