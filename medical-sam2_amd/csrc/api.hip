// Library-wide pieces of the C-ABI: version, last-error string, launch checking, and hipGraph capture helpers used to
// replay a whole per-slice forward with a single launch.
#include "common.h"

#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void msam2_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int msam2_check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    msam2_set_error("%s: %s", what, hipGetErrorString(e));
    return MSAM2_ERR_LAUNCH;
  }
  return MSAM2_OK;
}

extern "C" const char* msam2_last_error(void) { return g_err; }
extern "C" int msam2_version(void) { return 100; }  // 0.1.0
// 16-bit operand type the library was built with: 1 = IEEE fp16 (default), 0 = bf16 (-DMSAM2_OPERAND_BF16)
extern "C" int msam2_operand_is_fp16(void) { return MSAM2_OPERAND_IS_FP16; }

// ---- hipGraph helpers: capture everything enqueued on `stream` between begin/end, replay with launch ----
extern "C" int msam2_graph_begin(void* stream) {
  const hipError_t e = hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) { msam2_set_error("graph_begin: %s", hipGetErrorString(e)); return MSAM2_ERR_LAUNCH; }
  return MSAM2_OK;
}

extern "C" int msam2_graph_end(void* stream, void** graph_exec_out) {
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture((hipStream_t)stream, &g);
  if (e != hipSuccess || !g) { msam2_set_error("graph_end: %s", hipGetErrorString(e)); return MSAM2_ERR_LAUNCH; }
  hipGraphExec_t ex = nullptr;
  e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  hipGraphDestroy(g);
  if (e != hipSuccess) { msam2_set_error("graph_instantiate: %s", hipGetErrorString(e)); return MSAM2_ERR_LAUNCH; }
  *graph_exec_out = (void*)ex;
  return MSAM2_OK;
}

extern "C" int msam2_graph_launch(void* graph_exec, void* stream) {
  const hipError_t e = hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream);
  if (e != hipSuccess) { msam2_set_error("graph_launch: %s", hipGetErrorString(e)); return MSAM2_ERR_LAUNCH; }
  return MSAM2_OK;
}

extern "C" int msam2_graph_destroy(void* graph_exec) {
  if (graph_exec) hipGraphExecDestroy((hipGraphExec_t)graph_exec);
  return MSAM2_OK;
}

// ---- HIP-event timing on an explicit stream (bench.py's roofline leg times kernels on the launch stream itself) ----
extern "C" int msam2_event_create(void** ev) {
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) { msam2_set_error("event_create failed"); return MSAM2_ERR_LAUNCH; }
  *ev = (void*)e;
  return MSAM2_OK;
}
extern "C" int msam2_event_record(void* ev, void* stream) {
  return hipEventRecord((hipEvent_t)ev, (hipStream_t)stream) == hipSuccess ? MSAM2_OK : MSAM2_ERR_LAUNCH;
}
extern "C" int msam2_event_elapsed_ms(void* start, void* stop, float* ms) {
  if (hipEventSynchronize((hipEvent_t)stop) != hipSuccess) return MSAM2_ERR_LAUNCH;
  return hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop) == hipSuccess ? MSAM2_OK : MSAM2_ERR_LAUNCH;
}
extern "C" int msam2_event_destroy(void* ev) {
  if (ev) hipEventDestroy((hipEvent_t)ev);
  return MSAM2_OK;
}
