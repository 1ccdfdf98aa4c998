// Runtime plumbing behind the C-ABI (include/winograd_mi355x.h): device selection,
// memory, streams, events.  Replaces the cuda* calls the reference's host drivers make
// (Test.c:15; Kernel128_winograd.cu:236-286) so that the C host needs no HIP headers.
#include "wino_common.h"

#include <atomic>
#include <cstring>
#include <mutex>
#include <vector>

namespace wino {

static thread_local char g_err[512] = "no error";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
  set_error("%s: %s (%s)", what, hipGetErrorName(e), hipGetErrorString(e));
  return WINO_E_HIP;
}

namespace {
struct SkScratch {
  int dev;
  hipStream_t stream;
  float* slabs;
  size_t slab_bytes;
  unsigned* tickets;
  size_t n_tickets;
};
std::mutex g_ws_mu;
std::vector<SkScratch> g_ws;
}  // namespace

int sk_scratch(int dev, hipStream_t s, size_t slab_bytes, size_t n_tickets, float** slabs, unsigned** tickets) {
  std::lock_guard<std::mutex> lock(g_ws_mu);
  SkScratch* ws = nullptr;
  for (auto& e : g_ws)
    if (e.dev == dev && e.stream == s) ws = &e;
  if (!ws) {
    g_ws.push_back(SkScratch{dev, s, nullptr, 0, nullptr, 0});
    ws = &g_ws.back();
  }
  if (ws->slab_bytes < slab_bytes) {
    if (ws->slabs) { WINO_HIP(hipDeviceSynchronize()); WINO_HIP(hipFree(ws->slabs)); ws->slabs = nullptr; ws->slab_bytes = 0; }
    size_t n = (size_t)32 << 20;   // 32 MiB covers every reference shape on 256 CUs
    while (n < slab_bytes) n *= 2;
    WINO_HIP(hipMalloc((void**)&ws->slabs, n));
    ws->slab_bytes = n;
  }
  if (ws->n_tickets < n_tickets) {
    if (ws->tickets) { WINO_HIP(hipDeviceSynchronize()); WINO_HIP(hipFree(ws->tickets)); ws->tickets = nullptr; ws->n_tickets = 0; }
    size_t n = 4096;
    while (n < n_tickets) n *= 2;
    WINO_HIP(hipMalloc((void**)&ws->tickets, n * sizeof(unsigned)));
    WINO_HIP(hipMemset(ws->tickets, 0, n * sizeof(unsigned)));
    ws->n_tickets = n;
  }
  *slabs = ws->slabs;
  *tickets = ws->tickets;
  return WINO_OK;
}

// Frees the scratch of `s` on every device (the stream is going away; its handle may be reused).
int sk_scratch_release(hipStream_t s) {
  std::lock_guard<std::mutex> lock(g_ws_mu);
  int cur = 0;
  WINO_HIP(hipGetDevice(&cur));
  for (size_t i = 0; i < g_ws.size();) {
    if (g_ws[i].stream != s) { i++; continue; }
    WINO_HIP(hipSetDevice(g_ws[i].dev));
    if (g_ws[i].slabs) WINO_HIP(hipFree(g_ws[i].slabs));
    if (g_ws[i].tickets) WINO_HIP(hipFree(g_ws[i].tickets));
    g_ws.erase(g_ws.begin() + (long)i);
  }
  WINO_HIP(hipSetDevice(cur));
  return WINO_OK;
}

int device_cus(int dev, int* cus) {
  static std::atomic<int> cache[64];
  int c = dev >= 0 && dev < 64 ? cache[dev].load() : 0;
  if (!c) {
    WINO_HIP(hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev));
    if (dev >= 0 && dev < 64) cache[dev].store(c);
  }
  *cus = c;
  return WINO_OK;
}

}  // namespace wino

using namespace wino;

extern "C" {

int wino_abi_version(void) { return WINO_ABI_VERSION; }

const char* wino_last_error_string(void) { return g_err; }

int wino_device_count(int* count) {
  if (!count) return WINO_E_ARG;
  *count = 0;
  WINO_HIP(hipGetDeviceCount(count));
  return WINO_OK;
}

int wino_set_device(int device) {
  WINO_HIP(hipSetDevice(device));
  return WINO_OK;
}

int wino_device_name(int device, char* buf, size_t buflen) {
  if (!buf || buflen == 0) return WINO_E_ARG;
  hipDeviceProp_t p;
  WINO_HIP(hipGetDeviceProperties(&p, device));
  snprintf(buf, buflen, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
  return WINO_OK;
}

int wino_malloc(void** dptr, size_t bytes) {
  if (!dptr) return WINO_E_ARG;
  WINO_HIP(hipMalloc(dptr, bytes ? bytes : 4));
  return WINO_OK;
}

int wino_free(void* dptr) {
  if (!dptr) return WINO_OK;
  WINO_HIP(hipFree(dptr));
  return WINO_OK;
}

int wino_memset(void* dptr, int value, size_t bytes) {
  WINO_HIP(hipMemset(dptr, value, bytes));
  return WINO_OK;
}

int wino_memcpy_h2d(void* dst, const void* src, size_t bytes) {
  WINO_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return WINO_OK;
}

int wino_memcpy_d2h(void* dst, const void* src, size_t bytes) {
  WINO_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return WINO_OK;
}

int wino_memcpy_d2d(void* dst, const void* src, size_t bytes) {
  WINO_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToDevice));
  return WINO_OK;
}

int wino_device_synchronize(void) {
  WINO_HIP(hipDeviceSynchronize());
  return WINO_OK;
}

int wino_stream_create(wino_stream_t* stream) {
  if (!stream) return WINO_E_ARG;
  hipStream_t s;
  WINO_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream = (wino_stream_t)s;
  return WINO_OK;
}

int wino_stream_destroy(wino_stream_t stream) {
  if (!stream) return WINO_OK;
  // the stream's work is done before its stream-K scratch goes away
  WINO_HIP(hipStreamSynchronize((hipStream_t)stream));
  if (int rc = sk_scratch_release((hipStream_t)stream)) return rc;
  WINO_HIP(hipStreamDestroy((hipStream_t)stream));
  return WINO_OK;
}

int wino_stream_synchronize(wino_stream_t stream) {
  WINO_HIP(hipStreamSynchronize((hipStream_t)stream));
  return WINO_OK;
}

int wino_event_create(void** event) {
  if (!event) return WINO_E_ARG;
  hipEvent_t e;
  WINO_HIP(hipEventCreate(&e));
  *event = (void*)e;
  return WINO_OK;
}

int wino_event_destroy(void* event) {
  if (!event) return WINO_OK;
  WINO_HIP(hipEventDestroy((hipEvent_t)event));
  return WINO_OK;
}

int wino_event_record(void* event, wino_stream_t stream) {
  WINO_HIP(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
  return WINO_OK;
}

int wino_event_elapsed_ms(void* start, void* stop, float* ms) {
  if (!ms) return WINO_E_ARG;
  WINO_HIP(hipEventSynchronize((hipEvent_t)stop));
  WINO_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return WINO_OK;
}

}  // extern "C"
