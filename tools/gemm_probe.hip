// Diagnostic only: per-workgroup timeline of the 128x128x32 DMA GEMM (entry, first tile landed, loop end, epilogue end).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMSAM2_GSTAMP -Wno-unused-value -Wno-unused-result tools/gemm_probe.hip -o tools/gemm_probe.bin
#include "../medical-sam2_amd/csrc/api.hip"
#include "../medical-sam2_amd/csrc/gemm.hip"
#include <algorithm>
#include <cstdio>
#include <vector>
int main(int argc, char** argv) {
  const int64_t M = argc > 1 ? atoll(argv[1]) : 16384, N = argc > 2 ? atoll(argv[2]) : 1536, K = argc > 3 ? atoll(argv[3]) : 384;
  std::vector<_Float16> h((size_t)std::max(M, N) * K);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (_Float16)(((int)(i * 2654435761u >> 20) % 200 - 100) * 0.01f);
  _Float16 *a, *w, *c;
  float* bias;
  hipMalloc(&a, M * K * 2); hipMalloc(&w, N * K * 2); hipMalloc(&c, M * N * 2); hipMalloc(&bias, N * 4);
  hipMemcpy(a, h.data(), M * K * 2, hipMemcpyHostToDevice);
  hipMemcpy(w, h.data(), N * K * 2, hipMemcpyHostToDevice);
  hipMemset(bias, 0, N * 4);
  int mode = argc > 4 ? atoi(argv[4]) : 0;
  hipMemcpyToSymbol(HIP_SYMBOL(g_epi_mode), &mode, sizeof(int));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 6; ++it) {
    if (it == 5) hipEventRecord(e0, nullptr);
    int rc = msam2_gemm(a, K, w, K, bias, nullptr, nullptr, 0, 0, 0, c, N, 1, M, N, K, 0, nullptr);
    if (rc) { printf("error %s\n", msam2_last_error()); return 1; }
    if (it == 5) hipEventRecord(e1, nullptr);
  }
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const int tiles = (int)(((M + 127) / 128) * ((N + 127) / 128));
  const int nb = std::min(tiles, 8192);
  std::vector<unsigned long long> st(8 * 8192);
  hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_gstamp), st.size() * 8);
  unsigned long long t0 = ~0ull, t3 = 0;
  for (int b = 0; b < nb; ++b) { t0 = std::min(t0, st[b * 8]); t3 = std::max(t3, st[b * 8 + 3]); }
  double pro = 0, loop = 0, epi = 0, tot = 0;
  for (int b = 0; b < nb; ++b) {
    pro += st[b * 8 + 1] - st[b * 8]; loop += st[b * 8 + 2] - st[b * 8 + 1]; epi += st[b * 8 + 3] - st[b * 8 + 2]; tot += st[b * 8 + 3] - st[b * 8];
  }
  printf("M %lld N %lld K %lld: %d workgroups, event time %.1f us, span %llu ticks => %.1f ticks/us\n", (long long)M, (long long)N, (long long)K,
         tiles, ms * 1e3, t3 - t0, (t3 - t0) / (ms * 1e3));
  printf("mean per workgroup: entry->tile0+step0 %.0f, loop(steps 1..) %.0f, epilogue %.0f, total %.0f ticks\n", pro / nb, loop / nb, epi / nb, tot / nb);
  double e_bar = 0, e_issue = 0, e_drain = 0;
  for (int b = 0; b < nb; ++b) { e_bar += st[b * 8 + 6] - st[b * 8 + 2]; e_issue += st[b * 8 + 7] - st[b * 8 + 6]; e_drain += st[b * 8 + 3] - st[b * 8 + 7]; }
  printf("epilogue split: barrier %.0f, transpose+math+store issue %.0f, store drain (vmcnt 0) %.0f ticks\n", e_bar / nb, e_issue / nb, e_drain / nb);
  // start-time histogram (in 10%% buckets of the span) and per-XCC counts
  int hist[10] = {0}, xcc[16] = {0};
  for (int b = 0; b < nb; ++b) { hist[std::min(9, (int)((st[b * 8] - t0) * 10 / (t3 - t0 + 1)))]++; xcc[st[b * 8 + 5] & 15]++; }
  printf("start-time histogram:"); for (int i = 0; i < 10; ++i) printf(" %d", hist[i]); printf("\nper-XCC workgroups:");
  for (int i = 0; i < 8; ++i) printf(" %d", xcc[i]); printf("\n");
  {  // workgroups per CU (HW_ID: cu 11:8, sh 12, se 15:13) -- how a grid smaller than the chip's slots is spread
    std::vector<int> keys(nb);
    for (int b = 0; b < nb; ++b) keys[b] = (int)(((st[b * 8 + 5] & 15) << 16) | (st[b * 8 + 4] & 0xFF00));
    std::sort(keys.begin(), keys.end());
    int distinct = 0, mx = 0, run = 0;
    for (int b = 0; b < nb; ++b) { run = (b && keys[b] == keys[b - 1]) ? run + 1 : 1; if (run == 1) ++distinct; mx = std::max(mx, run); }
    printf("workgroups on %d distinct CUs, at most %d on one CU\n", distinct, mx);
  }
  // the first and last few workgroups in start order
  std::vector<int> ord(nb); for (int i = 0; i < nb; ++i) ord[i] = i;
  std::sort(ord.begin(), ord.end(), [&](int x, int y) { return st[x * 8] < st[y * 8]; });
  for (int i : {0, 1, nb / 4, nb / 2, 3 * nb / 4, nb - 2, nb - 1}) {
    const int b = ord[i];
    printf("  #%d wg %d: start %llu, pro %llu, loop %llu, epi %llu, hw 0x%llx xcc %llu\n", i, b, st[b * 8] - t0, st[b * 8 + 1] - st[b * 8],
           st[b * 8 + 2] - st[b * 8 + 1], st[b * 8 + 3] - st[b * 8 + 2], st[b * 8 + 4], st[b * 8 + 5]);
  }
  return 0;
}
