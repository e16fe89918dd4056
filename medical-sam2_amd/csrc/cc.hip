// 8-connected component labelling + per-pixel component area on uint8 masks, gfx950.
//
// Drop-in for the reference's only native op, sam2_train/csrc/connected_components.cu:213-282
// (`_C.get_connected_componnets`, called from utils/misc.py:47-63 for hole filling, misc.py:247-258).
// Same output contract (labels = 1 + index of the component's smallest 2x2-block corner, counts = component area), but
// laid out for MI355X: the whole batch runs in grid.z (5 launches in total instead of 6 per image), the union-find parent
// array lives in a caller-provided scratch buffer so the final labelling has no read/write race with path walking, and
// the area histogram uses wave-aggregated atomics (one atomic per wave when all 64 lanes sit in the same component,
// which is the common case for the large background region of `mask <= 0`).
#include "common.h"

namespace {

__device__ __forceinline__ int uf_find(const int* parent, int n) {
  int p = __hip_atomic_load(parent + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while (p != n) {
    n = p;
    p = __hip_atomic_load(parent + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return n;
}

__device__ __forceinline__ void uf_union(int* parent, int a, int b) {
  bool done;
  do {
    a = uf_find(parent, a);
    b = uf_find(parent, b);
    if (a < b) {
      const int old = atomicMin(parent + b, a);
      done = (old == b);
      b = old;
    } else if (b < a) {
      const int old = atomicMin(parent + a, b);
      done = (old == a);
      a = old;
    } else {
      done = true;
    }
  } while (!done);
}

__global__ void cc_init_kernel(int* __restrict__ parent, int W, int H) {
  const int col = (blockIdx.x * blockDim.x + threadIdx.x) * 2, row = (blockIdx.y * blockDim.y + threadIdx.y) * 2;
  if (row < H && col < W) {
    int* p = parent + (int64_t)blockIdx.z * W * H;
    p[row * W + col] = row * W + col;
  }
}

__global__ void cc_merge_kernel(const uint8_t* __restrict__ img_all, int* __restrict__ parent_all, int W, int H) {
  const int col = (blockIdx.x * blockDim.x + threadIdx.x) * 2, row = (blockIdx.y * blockDim.y + threadIdx.y) * 2;
  if (row >= H || col >= W) return;
  const uint8_t* img = img_all + (int64_t)blockIdx.z * W * H;
  int* parent = parent_all + (int64_t)blockIdx.z * W * H;
  const int idx = row * W + col;
  // 4x4 neighbourhood mask: bit (4*dy + dx) says "cell (row-1+dy, col-1+dx) touches a foreground pixel of this block"
  unsigned P = 0;
  if (img[idx]) P |= 0x777;
  if (row + 1 < H && img[idx + W]) P |= 0x777 << 4;
  if (col + 1 < W && img[idx + 1]) P |= 0x777 << 1;
  if (col == 0) P &= 0xEEEE;
  if (col + 1 >= W) P &= 0x3333;
  else if (col + 2 >= W) P &= 0x7777;
  if (row == 0) P &= 0xFFF0;
  if (row + 1 >= H) P &= 0xFF;
  if (!P) return;
  if ((P & 1u) && img[idx - W - 1]) uf_union(parent, idx, idx - 2 * W - 2);
  if (((P & 2u) && img[idx - W]) || ((P & 4u) && img[idx - W + 1])) uf_union(parent, idx, idx - 2 * W);
  if ((P & 8u) && img[idx + 2 - W]) uf_union(parent, idx, idx - 2 * W + 2);
  if (((P & 16u) && img[idx - 1]) || ((P & 256u) && img[idx + W - 1])) uf_union(parent, idx, idx - 2);
}

__global__ void cc_label_kernel(const uint8_t* __restrict__ img_all, const int* __restrict__ parent_all, int* __restrict__ labels_all,
                                int W, int H) {
  const int col = (blockIdx.x * blockDim.x + threadIdx.x) * 2, row = (blockIdx.y * blockDim.y + threadIdx.y) * 2;
  if (row >= H || col >= W) return;
  const int64_t off = (int64_t)blockIdx.z * W * H;
  const uint8_t* img = img_all + off;
  int* labels = labels_all + off;
  const int idx = row * W + col;
  const int y = uf_find(parent_all + off, idx) + 1;
  labels[idx] = img[idx] ? y : 0;
  if (col + 1 < W) labels[idx + 1] = img[idx + 1] ? y : 0;
  if (row + 1 < H) labels[idx + W] = img[idx + W] ? y : 0;
  if (col + 1 < W && row + 1 < H) labels[idx + W + 1] = img[idx + W + 1] ? y : 0;
}

__global__ void cc_count_kernel(const int* __restrict__ labels_all, int* __restrict__ hist_all, int n_pix) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t off = (int64_t)blockIdx.z * n_pix;
  const int y = (i < n_pix) ? labels_all[off + i] : 0;
  // wave-aggregated histogram update
  const int first = __builtin_amdgcn_readfirstlane(y);
  const unsigned long long same = __ballot(y == first && y > 0);
  if (y > 0) {
    if (y == first) {
      const int lane = threadIdx.x & 63;
      if (lane == __ffsll((long long)same) - 1) atomicAdd(hist_all + off + y - 1, (int)__popcll(same));
    } else {
      atomicAdd(hist_all + off + y - 1, 1);
    }
  }
}

__global__ void cc_area_kernel(const int* __restrict__ labels_all, const int* __restrict__ hist_all, int* __restrict__ counts_all,
                               int n_pix) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pix) return;
  const int64_t off = (int64_t)blockIdx.z * n_pix;
  const int y = labels_all[off + i];
  counts_all[off + i] = y > 0 ? hist_all[off + y - 1] : 0;
}

// hole filling helpers (utils/misc.py:247-258)
__global__ void threshold_kernel(const float* __restrict__ m, uint8_t* __restrict__ out, int64_t n, float thr, int above) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = above ? (m[i] > thr) : (m[i] <= thr);
}
__global__ void fill_kernel(float* __restrict__ m, const int* __restrict__ labels, const int* __restrict__ counts, int max_area,
                            int64_t n, float value) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    if (labels[i] > 0 && counts[i] <= max_area) m[i] = value;
}

// (a kernel, not hipMemsetAsync: memset nodes of a captured graph did not order reliably against neighbouring kernel nodes)
__global__ void cc_zero_kernel(int* __restrict__ x, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] = 0;
}

}  // namespace

extern "C" size_t msam2_cc_workspace_bytes(int64_t N, int64_t H, int64_t W) { return (size_t)(2 * N * H * W) * sizeof(int); }

extern "C" int msam2_cc_label(const uint8_t* img, int32_t* labels, int32_t* counts, int64_t N, int64_t H, int64_t W, void* workspace,
                              size_t workspace_bytes, void* stream) {
  MSAM2_REQUIRE(img && labels && counts && workspace, "cc_label: inputs must be device pointers");
  MSAM2_REQUIRE(N > 0 && H > 0 && W > 0, "cc_label: inputs must be [N, 1, H, W] shape");
  MSAM2_REQUIRE(H % 2 == 0, "cc_label: height must be a even number");
  MSAM2_REQUIRE(W % 2 == 0, "cc_label: width must be a even number");
  MSAM2_REQUIRE(H * W < (1ll << 31) && N <= 65535, "cc_label: image too large");
  MSAM2_REQUIRE(workspace_bytes >= msam2_cc_workspace_bytes(N, H, W), "cc_label: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  int* parent = (int*)workspace;
  int* hist = parent + N * H * W;
  hipLaunchKernelGGL(cc_zero_kernel, dim3((unsigned)min((int64_t)2048, cdiv(N * H * W, (int64_t)256))), dim3(256), 0, s, hist, N * H * W);
  dim3 blk(32, 8), grid(cdiv(W / 2, 32), cdiv(H / 2, 8), (unsigned)N);
  hipLaunchKernelGGL(cc_init_kernel, grid, blk, 0, s, parent, (int)W, (int)H);
  hipLaunchKernelGGL(cc_merge_kernel, grid, blk, 0, s, img, parent, (int)W, (int)H);
  hipLaunchKernelGGL(cc_label_kernel, grid, blk, 0, s, img, parent, labels, (int)W, (int)H);
  const int npix = (int)(H * W);
  dim3 g2(cdiv(npix, 256), 1, (unsigned)N);
  hipLaunchKernelGGL(cc_count_kernel, g2, dim3(256), 0, s, labels, hist, npix);
  hipLaunchKernelGGL(cc_area_kernel, g2, dim3(256), 0, s, labels, hist, counts, npix);
  return msam2_check_launch("cc_label");
}

extern "C" size_t msam2_fill_holes_workspace_bytes(int64_t N, int64_t H, int64_t W) {
  return (size_t)(N * H * W) * (1 + 4 * sizeof(int)) + 256;
}

// Small-component filling on fp32 mask scores [N,1,H,W], in place.  above = 0: components of (score <= threshold) ("holes");
// above = 1: components of (score > threshold) ("sprinkles"); components of area <= max_area get `fill_value`.
// Covers fill_holes_in_mask_scores (utils/misc.py:247-258: threshold 0, value 0.1) and SAM2Transforms.postprocess_masks
// (utils/transforms.py:74-98: holes -> threshold + 10, sprinkles -> threshold - 10).
extern "C" int msam2_fill_components(float* mask, int64_t N, int64_t H, int64_t W, int max_area, float threshold, int above,
                                     float fill_value, void* workspace, size_t workspace_bytes, void* stream) {
  MSAM2_REQUIRE(mask && workspace, "fill_components: null pointer");
  MSAM2_REQUIRE(max_area > 0, "fill_components: max_area must be positive");
  MSAM2_REQUIRE(workspace_bytes >= msam2_fill_holes_workspace_bytes(N, H, W), "fill_components: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int64_t n = N * H * W;
  int* labels = (int*)workspace;
  int* counts = labels + n;
  int* ccws = counts + n;
  uint8_t* bin = (uint8_t*)(ccws + 2 * n);
  hipLaunchKernelGGL(threshold_kernel, dim3((unsigned)min((int64_t)4096, (n + 255) / 256)), dim3(256), 0, s, mask, bin, n, threshold, above);
  const int rc = msam2_cc_label(bin, labels, counts, N, H, W, ccws, sizeof(int) * 2 * n, stream);
  if (rc != MSAM2_OK) return rc;
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)min((int64_t)4096, (n + 255) / 256)), dim3(256), 0, s, mask, labels, counts, max_area, n,
                     fill_value);
  return msam2_check_launch("fill_components");
}

// fill_holes_in_mask_scores (utils/misc.py:247-258): background components (score <= 0) of area <= max_area get score 0.1
extern "C" int msam2_fill_holes(float* mask, int64_t N, int64_t H, int64_t W, int max_area, void* workspace, size_t workspace_bytes,
                                void* stream) {
  return msam2_fill_components(mask, N, H, W, max_area, 0.0f, 0, 0.1f, workspace, workspace_bytes, stream);
}
