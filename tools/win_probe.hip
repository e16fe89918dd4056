// Diagnostic only: per-workgroup timeline of attn_win_kernel at the stage-3 window shape (B=4, 64x64 tokens, 4 heads, ws 14), -DMSAM2_STAMP.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMSAM2_STAMP -w tools/win_probe.hip -o tools/win_probe.bin
#include "../medical-sam2_amd/csrc/api.hip"
#include "../medical-sam2_amd/csrc/attention.hip"
#include <vector>
#include <cstdio>
int main() {
  const int64_t B = 4, HW = 64, heads = 4, D = 96, ws = 14, dim = heads * D, T = B * HW * HW;
  std::vector<_Float16> h((size_t)T * 3 * dim);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (_Float16)(((int)(i * 2654435761u >> 20) % 200 - 100) * 0.005f);
  _Float16 *qkv, *o; float* bias;
  hipMalloc(&qkv, h.size() * 2); hipMalloc(&o, T * dim * 2); hipMalloc(&bias, 3 * dim * 4);
  hipMemcpy(qkv, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipMemset(bias, 0, 3 * dim * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 8; ++it) {
    if (it == 3) hipEventRecord(e0, nullptr);
    int rc = msam2_window_attention_fwd(qkv, 3 * dim, D, HW, HW, ws, qkv + dim, qkv + 2 * dim, 3 * dim, D, HW, HW, ws, bias + dim, bias + 2 * dim, o, dim, D,
                                        B, heads, D, 0.102f, nullptr);
    if (rc) { printf("error %s\n", msam2_last_error()); return 1; }
  }
  hipEventRecord(e1, nullptr);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  static unsigned long long wt[4096][4];
  hipMemcpyFromSymbol(wt, HIP_SYMBOL(g_wgtime), sizeof(wt));
  const int nwg = 400;
  unsigned long long t0 = ~0ull, t3 = 0;
  for (int i = 0; i < nwg; ++i) { if (wt[i][0] < t0) t0 = wt[i][0]; if (wt[i][3] > t3) t3 = wt[i][3]; }
  double a = 0, g = 0, c = 0, st = 0, mxin = 0;
  for (int i = 0; i < nwg; ++i) {
    a += (wt[i][0] - t0) * 0.01; g += (wt[i][1] - wt[i][0]) * 0.01; c += (wt[i][2] - wt[i][1]) * 0.01; st += (wt[i][3] - wt[i][2]) * 0.01;
    if ((wt[i][0] - t0) * 0.01 > mxin) mxin = (wt[i][0] - t0) * 0.01;
  }
  printf("%.1f us per launch; %d workgroups: first entry -> last exit %.1f us; entry mean %.1f max %.1f us; gather+barrier mean %.1f us; compute mean %.1f us; store mean %.1f us\n",
         ms * 1e3 / 5, nwg, (t3 - t0) * 0.01, a / nwg, mxin, g / nwg, c / nwg, st / nwg);
  return 0;
}
