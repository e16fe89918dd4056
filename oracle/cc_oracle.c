/* CPU ORACLE (test infrastructure only; never linked into the product library).
 *
 * Plain-C restatement of the reference's only native op,
 *   sam2_train/csrc/connected_components.cu  (get_connected_componnets, lines 213-282):
 * 8-connected component labelling of uint8 masks [N,1,H,W] (H, W even) by a union-find over 2x2 pixel blocks,
 * followed by a per-pixel component-area map.  Output contract restated:
 *   labels[p] = 0 for background pixels; for foreground pixels 1 + (linear index of the top-left pixel of the
 *               component's root block), the root being the smallest block index in the component because every
 *               union hangs the larger root under the smaller one (union_, lines 42-60);
 *   counts[p] = number of foreground pixels in p's component (0 on background)   (lines 170-209).
 * The sequential order used here cannot change the result: the final root of a set is its minimum element.
 *
 * Build: see oracle/Makefile (gcc -O2 -shared -fPIC).  Pinned by tests/test_cc_oracle.py against hand-written
 * known-answer cases and scipy.ndimage.label(structure=ones((3,3))) (the reference's prebuilt _C.so is CUDA sm_89 /
 * CPython 3.12 and cannot be loaded in this image; its source needs nvcc + ATen CUDA headers => unbuildable here).
 */
#include <stdint.h>
#include <string.h>

static int32_t find_root(const int32_t *lab, int32_t n) {          /* find(), lines 27-31 */
    while (lab[n] != n) n = lab[n];
    return n;
}

static void unite(int32_t *lab, int32_t a, int32_t b) {            /* union_(), lines 42-60 (sequential form) */
    a = find_root(lab, a);
    b = find_root(lab, b);
    if (a < b) lab[b] = a;
    else if (b < a) lab[a] = b;
}

static void label_one(const uint8_t *img, int32_t *lab, int32_t *cnt, int32_t *scratch, int W, int H) {
    /* init_labeling, lines 62-70 */
    for (int r = 0; r < H; r += 2)
        for (int c = 0; c < W; c += 2) lab[r * W + c] = r * W + c;
    /* merge, lines 72-118: P collects which of the 16 neighbourhood cells around the block need testing */
    for (int r = 0; r < H; r += 2)
        for (int c = 0; c < W; c += 2) {
            const int idx = r * W + c;
            uint32_t P = 0;
            if (img[idx]) P |= 0x777;
            if (r + 1 < H && img[idx + W]) P |= 0x777 << 4;
            if (c + 1 < W && img[idx + 1]) P |= 0x777 << 1;
            if (c == 0) P &= 0xEEEE;
            if (c + 1 >= W) P &= 0x3333;
            else if (c + 2 >= W) P &= 0x7777;
            if (r == 0) P &= 0xFFF0;
            if (r + 1 >= H) P &= 0xFF;
            if (!P) continue;
            if (((P >> 0) & 1) && img[idx - W - 1]) unite(lab, idx, idx - 2 * W - 2);
            if ((((P >> 1) & 1) && img[idx - W]) || (((P >> 2) & 1) && img[idx - W + 1])) unite(lab, idx, idx - 2 * W);
            if (((P >> 3) & 1) && img[idx + 2 - W]) unite(lab, idx, idx - 2 * W + 2);
            if ((((P >> 4) & 1) && img[idx - 1]) || (((P >> 8) & 1) && img[idx + W - 1])) unite(lab, idx, idx - 2);
        }
    /* compression (lines 120-127) + final_labeling (129-168) */
    for (int r = 0; r < H; r += 2)
        for (int c = 0; c < W; c += 2) {
            const int idx = r * W + c;
            scratch[idx] = find_root(lab, idx) + 1;
        }
    for (int r = 0; r < H; r += 2)
        for (int c = 0; c < W; c += 2) {
            const int idx = r * W + c;
            const int32_t y = scratch[idx];
            lab[idx] = img[idx] ? y : 0;
            if (c + 1 < W) lab[idx + 1] = img[idx + 1] ? y : 0;
            if (r + 1 < H) lab[idx + W] = img[idx + W] ? y : 0;
            if (c + 1 < W && r + 1 < H) lab[idx + W + 1] = img[idx + W + 1] ? y : 0;
        }
    /* init_counting / final_counting, lines 170-209 */
    memset(scratch, 0, sizeof(int32_t) * (size_t)W * H);
    for (int p = 0; p < W * H; ++p)
        if (lab[p] > 0) scratch[lab[p] - 1] += 1;
    for (int p = 0; p < W * H; ++p) cnt[p] = lab[p] > 0 ? scratch[lab[p] - 1] : 0;
}

/* img: [N,H,W] uint8; labels, counts: [N,H,W] int32; scratch: [H*W] int32.  Returns 0, or -1 on odd H/W. */
int cc_oracle_label(const uint8_t *img, int32_t *labels, int32_t *counts, int32_t *scratch, int N, int H, int W) {
    if ((H & 1) || (W & 1)) return -1;
    for (int n = 0; n < N; ++n) {
        const size_t off = (size_t)n * H * W;
        memset(labels + off, 0, sizeof(int32_t) * (size_t)H * W);
        label_one(img + off, labels + off, counts + off, scratch, W, H);
    }
    return 0;
}
