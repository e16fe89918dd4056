// Backward building blocks for the fused LayerNorm / MLP pieces of the path (first slice): the
// gradient GEMMs reuse msam2_gemm (dX = dY W through a transposed weight copy, dW = dY^T X through transposed activations), so
// what is needed here is data movement (16-bit transpose), reductions (bias gradient) and the pointwise / row-wise derivative
// kernels.  Reference semantics: torch.autograd of nn.LayerNorm (hieradet.py:101-102, memory_attention.py:43-45), nn.GELU (exact
// erf) / nn.ReLU and nn.Linear (sam2_utils.py:108-132).
#include "common.h"

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

extern "C" int msam2_act_bwd(const void* pre, int pre_is_16bit, const void* dy, int dy_is_16bit, void* out, int64_t n, int act, void* stream) {
  MSAM2_REQUIRE(pre && dy && out && n > 0 && (act == 1 || act == 2), "act_bwd: bad arguments");
  dim3 grid((unsigned)min((int64_t)16384, cdiv(n, 256))), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (pre_is_16bit && dy_is_16bit) hipLaunchKernelGGL((act_bwd_kernel<op16, op16>), grid, block, 0, s, (const op16*)pre, (const op16*)dy, (op16*)out, n, act);
  else if (pre_is_16bit) hipLaunchKernelGGL((act_bwd_kernel<op16, float>), grid, block, 0, s, (const op16*)pre, (const float*)dy, (op16*)out, n, act);
  else if (dy_is_16bit) hipLaunchKernelGGL((act_bwd_kernel<float, op16>), grid, block, 0, s, (const float*)pre, (const op16*)dy, (op16*)out, n, act);
  else hipLaunchKernelGGL((act_bwd_kernel<float, float>), grid, block, 0, s, (const float*)pre, (const float*)dy, (op16*)out, n, act);
  return msam2_check_launch("act_bwd");
}

// ------------------------------------------------------------------------------------------------------------------
// LayerNorm backward, one wave per row (C <= 1024):  xhat = (x - mean) rstd,  g = dy * gamma,
//   dx = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat)),   dgamma += dy * xhat,   dbeta += dy   (fp32 atomics, zeroed by the caller)
// The statistics are recomputed from x (fp32 residual stream), so the forward saves nothing.
// ------------------------------------------------------------------------------------------------------------------
template <typename TD>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, int64_t ldx, const TD* __restrict__ dy, int64_t ldd,
                                                            const float* __restrict__ gamma, float* __restrict__ dx, int64_t ldo,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int64_t rows, int C,
                                                            float eps) {
  __shared__ float sg[1024], sb[1024];
  for (int c = threadIdx.x; c < C; c += 256) sg[c] = sb[c] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t rows_per_block = (rows + gridDim.x - 1) / gridDim.x;
  const int64_t r_begin = blockIdx.x * rows_per_block, r_end = min(rows, r_begin + rows_per_block);
  for (int64_t row = r_begin + wave; row < r_end; row += 4) {
    const float* xr = x + row * ldx;
    const TD* dr = dy + row * ldd;
    float xv[16], gv[16], s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = lane + 64 * i;
      xv[i] = c < C ? xr[c] : 0.f;
      gv[i] = c < C ? (float)dr[c] : 0.f;
      s += xv[i];
    }
    const float mean = wave_sum(s) / C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = lane + 64 * i;
      const float d = c < C ? xv[i] - mean : 0.f;
      q += d * d;
    }
    const float rstd = 1.f / sqrtf(wave_sum(q) / C + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = lane + 64 * i;
      if (c < C) {
        const float xh = (xv[i] - mean) * rstd, g = gv[i] * gamma[c];
        s1 += g;
        s2 += g * xh;
        atomicAdd(&sg[c], gv[i] * xh);
        atomicAdd(&sb[c], gv[i]);
        xv[i] = xh;
        gv[i] = g;
      }
    }
    const float m1 = wave_sum(s1) / C, m2 = wave_sum(s2) / C;
    float* o = dx + row * ldo;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = lane + 64 * i;
      if (c < C) o[c] = rstd * (gv[i] - m1 - xv[i] * m2);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    atomicAdd(dgamma + c, sg[c]);
    atomicAdd(dbeta + c, sb[c]);
  }
}

extern "C" int msam2_layernorm_bwd(const float* x, int64_t ldx, const void* dy, int dy_is_16bit, int64_t ldd, const float* gamma, float* dx,
                                   int64_t ldo, float* dgamma, float* dbeta, int64_t rows, int64_t C, float eps, void* stream) {
  MSAM2_REQUIRE(x && dy && gamma && dx && dgamma && dbeta, "layernorm_bwd: null tensor");
  MSAM2_REQUIRE(rows > 0 && C > 0 && C <= 1024, "layernorm_bwd: C <= 1024");
  const unsigned blocks = (unsigned)min((int64_t)1024, cdiv(rows, 16));
  if (dy_is_16bit)
    hipLaunchKernelGGL((layernorm_bwd_kernel<op16>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, (const op16*)dy, ldd, gamma, dx, ldo,
                       dgamma, dbeta, rows, (int)C, eps);
  else
    hipLaunchKernelGGL((layernorm_bwd_kernel<float>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, (const float*)dy, ldd, gamma, dx,
                       ldo, dgamma, dbeta, rows, (int)C, eps);
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
