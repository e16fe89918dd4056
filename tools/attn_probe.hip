// Diagnostic only: per-section cycle shares of the D=256 attention loops (one wave), built with -DMSAM2_STAMP.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMSAM2_STAMP tools/attn_probe.hip -o tools/attn_probe.bin ; ./tools/attn_probe.bin [kv64 splits]
#include "../medical-sam2_amd/csrc/api.hip"
#include "../medical-sam2_amd/csrc/attention.hip"
#include <vector>
#include <cstdio>
#include <cstring>
int main(int argc, char** argv) {
  const bool d96 = argc > 1 && !strcmp(argv[1], "d96");   // Hiera global attention: B=4, 4 heads, 4096 x 4096 x 96, one pass
  const int64_t B = 4, H = d96 ? 4 : 1, Lq = 4096, Lk = d96 ? 4096 : 16384, D = d96 ? 96 : 256;
  const bool kv64 = argc > 1 && !strcmp(argv[1], "kv64");
  const int splits = argc > 2 ? atoi(argv[2]) : (d96 ? 1 : kv64 ? 6 : 4);
  std::vector<_Float16> h((size_t)B * H * (Lk > Lq ? Lk : Lq) * D);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (_Float16)(((int)(i * 2654435761u >> 20) % 200 - 100) * 0.01f);
  _Float16 *q, *k, *v, *o;
  hipMalloc(&q, B * H * Lq * D * 2); hipMalloc(&k, B * H * Lk * D * 2); hipMalloc(&v, B * H * Lk * D * 2); hipMalloc(&o, B * H * Lq * D * 2);
  hipMemcpy(q, h.data(), B * H * Lq * D * 2, hipMemcpyHostToDevice);
  hipMemcpy(k, h.data(), B * H * Lk * D * 2, hipMemcpyHostToDevice);
  hipMemcpy(v, h.data(), B * H * Lk * D * 2, hipMemcpyHostToDevice);
  size_t wsb = msam2_attention_workspace_bytes(B, H, Lq, D, splits);
  void* ws; hipMalloc(&ws, wsb);
  int64_t qs[3] = {H * Lq * D, Lq * D, D}, ks[3] = {H * Lk * D, Lk * D, D}, vs[3] = {Lk * 64, Lk * 64, 64}, os[3] = {Lq * 64, Lq * 64, 64};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 8; ++it) {
    if (it == 3) hipEventRecord(e0, nullptr);
    int rc = kv64 ? msam2_attention_kv64_fwd(q, qs, k, ks, v, vs, o, os, B, H, Lq, Lk, 0.0625f, -splits, ws, wsb, nullptr)
                  : msam2_attention_fwd(q, qs, k, ks, v, ks, o, qs, B, H, Lq, Lk, D, 0.0625f, d96 ? 1 : -splits, ws, wsb, nullptr);
    if (rc) { printf("error %s\n", msam2_last_error()); return 1; }
  }
  hipEventRecord(e1, nullptr);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long st[16];
  hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamp), sizeof(st));
  const int o0 = kv64 ? 8 : 0;
  const char* names[5] = {"dma-issue+vmcnt+barrier", "QK (16 mfma + K reads)", "softmax (+V reads)", "PV mfma", "lgkm wait"};
  double tot = 0; for (int i = 0; i < 5; ++i) tot += st[o0 + i];
  printf("%s splits %d: %.1f us per launch (stamped build), tiles %llu, cycles/tile %.0f\n", kv64 ? "kv64" : d96 ? "d96" : "d256", splits, ms * 1e3 / 5, st[o0 + 5], tot / st[o0 + 5]);
  if (kv64) {
    static unsigned long long wt[4096][4];
    hipMemcpyFromSymbol(wt, HIP_SYMBOL(g_wgtime), sizeof(wt));
    const int nwg = 32 * B * splits;
    unsigned long long t0 = ~0ull, t3 = 0;
    for (int i = 0; i < nwg; ++i) { if (wt[i][0] < t0) t0 = wt[i][0]; if (wt[i][3] > t3) t3 = wt[i][3]; }
    double s_in = 0, s_ls = 0, s_le = 0, s_ex = 0, mx_in = 0, mn_le = 1e9, mx_le = 0, mn_loop = 1e9, mx_loop = 0;
    for (int i = 0; i < nwg; ++i) {
      const double a = (wt[i][0] - t0) * 0.01, b = (wt[i][1] - t0) * 0.01, c = (wt[i][2] - t0) * 0.01, d = (wt[i][3] - t0) * 0.01;
      s_in += a; s_ls += b; s_le += c; s_ex += d;
      if (a > mx_in) mx_in = a; if (c < mn_le) mn_le = c; if (c > mx_le) mx_le = c;
      if (c - b < mn_loop) mn_loop = c - b; if (c - b > mx_loop) mx_loop = c - b;
    }
    printf("  last launch, %d workgroups: first entry -> last exit %.1f us; entry mean %.1f max %.1f; loop start mean %.1f; loop end mean %.1f min %.1f max %.1f; exit mean %.1f; loop length min %.1f max %.1f us\n",
           nwg, (t3 - t0) * 0.01, s_in / nwg, mx_in, s_ls / nwg, s_le / nwg, mn_le, mx_le, s_ex / nwg, mn_loop, mx_loop);
  }
  if (kv64) printf("  loop: %llu shader cycles in %.2f us => in-kernel clock %.2f GHz\n", st[15], st[14] * 0.01, st[15] / (st[14] * 10.0));
  for (int i = 0; i < 5; ++i) printf("  %-28s %8.0f cycles/tile  %5.1f %%\n", names[i], (double)st[o0 + i] / st[o0 + 5], 100.0 * st[o0 + i] / tot);
  return 0;
}
