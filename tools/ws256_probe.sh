# diagnostic ladder of gemm_wstat256_kernel (MSAM2_WS256_PROBE: 1 no A DMA in the loop, 2 + no retire, 3 + fragments read once, 4 + no barrier,
# 5 = epilogue vector work without its stores) and the non-temporal store switch, on the step's three W-stationary shapes
for P in 0 1 2 3 4 5; do echo "== probe $P"; MSAM2_GEMM_WSTAT256=1 MSAM2_WS256_PROBE=$P python tools/wstat_ab.py child 2>&1 | grep -E "qkv  |fc1|linear1"; done
echo "== probe 0, plain (cached) stores"; MSAM2_GEMM_WSTAT256=1 MSAM2_NT_BYTES=100000000000 python tools/wstat_ab.py child 2>&1 | grep -E "qkv  |fc1|linear1"
echo "== 128-row kernel, nt"; python tools/wstat_ab.py child 2>&1 | grep -E "qkv  |fc1|linear1"
echo "== 128-row kernel, plain stores"; MSAM2_NT_BYTES=100000000000 python tools/wstat_ab.py child 2>&1 | grep -E "qkv  |fc1|linear1"
