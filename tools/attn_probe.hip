// Diagnostic only: per-section cycle shares of the D=256 attention loop (one wave), built with -DMSAM2_STAMP.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMSAM2_STAMP -I../medical-sam2_amd/csrc tools/attn_probe.hip -o /tmp/attn_probe
#include "../medical-sam2_amd/csrc/api.hip"
#include "../medical-sam2_amd/csrc/attention.hip"
#include <vector>
#include <cstdio>
int main() {
  const int64_t B = 4, H = 1, Lq = 4096, Lk = 16384, D = 256;
  const int splits = 4;
  std::vector<_Float16> h((size_t)B * Lk * D);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (_Float16)(((int)(i * 2654435761u >> 20) % 200 - 100) * 0.01f);
  _Float16 *q, *k, *v, *o;
  hipMalloc(&q, B * Lq * D * 2); hipMalloc(&k, B * Lk * D * 2); hipMalloc(&v, B * Lk * D * 2); hipMalloc(&o, B * Lq * D * 2);
  hipMemcpy(q, h.data(), B * Lq * D * 2, hipMemcpyHostToDevice);
  hipMemcpy(k, h.data(), B * Lk * D * 2, hipMemcpyHostToDevice);
  hipMemcpy(v, h.data(), B * Lk * D * 2, hipMemcpyHostToDevice);
  size_t wsb = msam2_attention_workspace_bytes(B, H, Lq, D, splits);
  void* ws; hipMalloc(&ws, wsb);
  int64_t qs[3] = {Lq * D, Lq * D, D}, ks[3] = {Lk * D, Lk * D, D};
  for (int it = 0; it < 5; ++it) {
    int rc = msam2_attention_fwd(q, qs, k, ks, v, ks, o, qs, B, H, Lq, Lk, D, 0.0625f, splits, ws, wsb, nullptr);
    if (rc) { printf("error %s\n", msam2_last_error()); return 1; }
  }
  hipDeviceSynchronize();
  unsigned long long st[16];
  hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamp), sizeof(st));
  const char* names[5] = {"dma-issue+vmcnt+barrierA", "QK (16 mfma + K reads)", "softmax", "PV (16 mfma + V reads)", "lgkm+barrierB"};
  double tot = 0; for (int i = 0; i < 5; ++i) tot += st[i];
  printf("tiles %llu, cycles/tile %.0f\n", st[5], tot / st[5]);
  for (int i = 0; i < 5; ++i) printf("  %-28s %8.0f cycles/tile  %5.1f %%\n", names[i], (double)st[i] / st[5], 100.0 * st[i] / tot);
  return 0;
}
