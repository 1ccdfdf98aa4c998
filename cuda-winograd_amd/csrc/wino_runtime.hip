// Runtime plumbing behind the C-ABI (include/winograd_mi355x.h): device selection,
// memory, streams, events.  Replaces the cuda* calls the reference's host drivers make
// (Test.c:15; Kernel128_winograd.cu:236-286) so that the C host needs no HIP headers.
#include "wino_common.h"

#include <atomic>
#include <cstring>
#include <mutex>
#include <utility>
#include <vector>

namespace wino {

static thread_local char g_err[512] = "no error";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static thread_local hipError_t g_last_status = hipSuccess;

int hip_fail(hipError_t e, const char* what) {
  g_last_status = e;
  set_error("%s: %s (%s)", what, hipGetErrorName(e), hipGetErrorString(e));
  return WINO_E_HIP;
}

namespace {
// One (device, stream)'s stream-K scratch.  Buffers are never freed or moved while the stream
// lives: a HIP graph captured on the stream has the slab and ticket pointers baked into its
// kernel arguments, and a launch may still be in flight when a larger shape arrives.  Growth
// allocates a new, larger generation and parks the old one in `retired` until
// wino_stream_destroy (sk_scratch_release); a graph captured against an old generation stays
// valid -- launches on one stream serialise, and every launch returns its counters to zero.
struct Retired {
  void* p;
  size_t n_tickets;   // 0: a slab buffer
};
struct SkScratch {
  int dev;
  hipStream_t stream;
  float* slabs;
  size_t slab_bytes;
  unsigned* tickets;
  size_t n_tickets;
  unsigned* err;      // pinned host word the kernels set on a dirty counter (never freed before the stream goes)
  bool failed;        // a launch that held this scratch failed on the host side
  std::vector<Retired> retired;
};
std::mutex g_ws_mu;
std::vector<SkScratch> g_ws;

SkScratch* find_ws(int dev, hipStream_t s) {
  for (auto& e : g_ws)
    if (e.dev == dev && e.stream == s) return &e;
  return nullptr;
}
}  // namespace

int sk_scratch(int dev, hipStream_t s, size_t slab_bytes, size_t n_tickets, SkBufs* out) {
  std::lock_guard<std::mutex> lock(g_ws_mu);
  SkScratch* ws = find_ws(dev, s);
  if (!ws) {
    unsigned* err = nullptr;
    WINO_HIP(hipHostMalloc((void**)&err, 64, hipHostMallocDefault));
    memset(err, 0, 64);
    g_ws.push_back(SkScratch{dev, s, nullptr, 0, nullptr, 0, err, false, {}});
    ws = &g_ws.back();
  }
  if (ws->failed || __atomic_load_n(ws->err, __ATOMIC_RELAXED) != 0) {
    set_error("the stream's ticket counters are in an unknown state (%s): call wino_stream_reset_scratch()",
              ws->failed ? "an earlier launch on it failed" : "a kernel found a counter that was not zero at launch");
    return WINO_E_STATE;
  }
  if (ws->slab_bytes < slab_bytes) {
    size_t n = (size_t)32 << 20;   // 32 MiB covers every reference shape on 256 CUs
    while (n < slab_bytes) n *= 2;
    float* fresh = nullptr;
    WINO_HIP(hipMalloc((void**)&fresh, n));
    if (ws->slabs) ws->retired.push_back(Retired{ws->slabs, 0});
    ws->slabs = fresh;
    ws->slab_bytes = n;
  }
  if (ws->n_tickets < n_tickets) {
    size_t n = 4096;
    while (n < n_tickets) n *= 2;
    unsigned* fresh = nullptr;
    WINO_HIP(hipMalloc((void**)&fresh, n * sizeof(unsigned)));
    // zeroed ON the launch stream: library streams are non-blocking, a null-stream memset would not be
    // ordered before the first launch that uses the counters (allocation never happens inside a capture)
    WINO_HIP(hipMemsetAsync(fresh, 0, n * sizeof(unsigned), s));
    if (ws->tickets) ws->retired.push_back(Retired{ws->tickets, ws->n_tickets});
    ws->tickets = fresh;
    ws->n_tickets = n;
  }
  out->slabs = ws->slabs;
  out->tickets = ws->tickets;
  out->err = ws->err;
  return WINO_OK;
}

void sk_mark_failed(int dev, hipStream_t s) {
  std::lock_guard<std::mutex> lock(g_ws_mu);
  if (SkScratch* ws = find_ws(dev, s)) ws->failed = true;
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
    if (g_ws[i].err) WINO_HIP(hipHostFree(g_ws[i].err));
    for (const Retired& r : g_ws[i].retired) WINO_HIP(hipFree(r.p));
    g_ws.erase(g_ws.begin() + (long)i);
  }
  WINO_HIP(hipSetDevice(cur));
  return WINO_OK;
}

// ---- developer knobs: the environment is read ONCE per process (first use), not per launch.
// Tests that sweep a knob change the environment and call wino_debug_reload_knobs().
namespace {
std::mutex g_knob_mu;
std::atomic<bool> g_knobs_ready{false};
Knobs g_knobs;
int env_num(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}
void read_knobs() {
  Knobs k;
  k.sk_grid = env_num("WINO_SK_GRID", 0);
  k.sk_min_iters = env_num("WINO_SK_MIN_ITERS", 0);
  const char* algo = getenv("WINO_3X3_ALGO");
  k.algo_3x3 = algo && !strcmp(algo, "big") ? 1 : algo && !strcmp(algo, "small") ? 2 : 0;
  k.sk_1x1 = env_num("WINO_1X1_SK", -1);
  k.sk_1x1_grid = env_num("WINO_1X1_SK_GRID", 0);
  k.sk_kp = env_num("WINO_SK_KP", 1);
  k.small_split = env_num("WINO_SMALL_SPLIT", 0);
  k.small3_ct = env_num("WINO_SMALL_CT", 0);
  const char* algo1 = getenv("WINO_1X1_ALGO");
  k.algo_1x1 = algo1 && !strcmp(algo1, "big") ? 1 : algo1 && !strcmp(algo1, "small") ? 2 : 0;
  k.small_ks = env_num("WINO_1X1_SMALL_KS", 0);
  k.small_rt = env_num("WINO_1X1_SMALL_RT", 0);
  k.small_ct = env_num("WINO_1X1_SMALL_CT", 0);
  g_knobs = k;
}
}  // namespace

Knobs knobs() {
  if (!g_knobs_ready.load(std::memory_order_acquire)) {
    std::lock_guard<std::mutex> lock(g_knob_mu);
    if (!g_knobs_ready.load()) {
      read_knobs();
      g_knobs_ready.store(true, std::memory_order_release);
    }
    return g_knobs;
  }
  std::lock_guard<std::mutex> lock(g_knob_mu);   // wino_debug_reload_knobs() writes g_knobs under the same mutex
  return g_knobs;
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

// hipGetErrorName of the status the calling thread's most recent memcpy / synchronise wrapper got
// from the runtime: the line the reference prints after each copy-back (Kernel128_winograd.cu:275,409).
const char* wino_last_status_name(void) { return hipGetErrorName(g_last_status); }

int wino_debug_reload_knobs(void) {
  std::lock_guard<std::mutex> lock(g_knob_mu);
  read_knobs();
  g_knobs_ready.store(true, std::memory_order_release);
  return WINO_OK;
}

// Invariant check for tests: every stream-K launch returns the ticket counters it used to zero (the
// last arriver of an item resets them), so between launches all counters of a stream's scratch -- the
// current generation and the retired ones a captured graph may still use -- read 0.  Synchronises the
// stream and counts the non-zero ones of every generation on the current device.
int wino_debug_tickets_in_use(wino_stream_t stream, long* nonzero) {
  if (!nonzero) return WINO_E_ARG;
  *nonzero = 0;
  int dev = 0;
  WINO_HIP(hipGetDevice(&dev));
  WINO_HIP(hipStreamSynchronize((hipStream_t)stream));
  std::vector<std::pair<unsigned*, size_t>> bufs;
  {
    std::lock_guard<std::mutex> lock(g_ws_mu);
    if (SkScratch* ws = find_ws(dev, (hipStream_t)stream)) {
      if (ws->tickets) bufs.push_back({ws->tickets, ws->n_tickets});
      for (const Retired& r : ws->retired)
        if (r.n_tickets) bufs.push_back({(unsigned*)r.p, r.n_tickets});
    }
  }
  for (auto& b : bufs) {
    std::vector<unsigned> host(b.second);
    WINO_HIP(hipMemcpy(host.data(), b.first, b.second * sizeof(unsigned), hipMemcpyDeviceToHost));
    for (unsigned v : host) *nonzero += v != 0;
  }
  return WINO_OK;
}

// Recovery after an aborted launch: waits for the stream, zeroes every ticket counter of its scratch on the
// current device (all generations; the slabs need no cleaning, they are written before they are read) and
// clears the error state, after which launches on the stream are accepted again.
int wino_stream_reset_scratch(wino_stream_t stream) {
  int dev = 0;
  WINO_HIP(hipGetDevice(&dev));
  WINO_HIP(hipStreamSynchronize((hipStream_t)stream));
  std::lock_guard<std::mutex> lock(g_ws_mu);
  SkScratch* ws = find_ws(dev, (hipStream_t)stream);
  if (!ws) return WINO_OK;
  if (ws->tickets) WINO_HIP(hipMemsetAsync(ws->tickets, 0, ws->n_tickets * sizeof(unsigned), (hipStream_t)stream));
  for (const Retired& r : ws->retired)
    if (r.n_tickets) WINO_HIP(hipMemsetAsync(r.p, 0, r.n_tickets * sizeof(unsigned), (hipStream_t)stream));
  WINO_HIP(hipStreamSynchronize((hipStream_t)stream));
  __atomic_store_n(ws->err, 0u, __ATOMIC_RELAXED);
  ws->failed = false;
  return WINO_OK;
}

// Waits for the stream and reports whether its scratch can be trusted: WINO_E_STATE when a launch on it
// failed, when a kernel drew a ticket on a counter that was not zero when its launch began, or when any
// counter is non-zero now (the authoritative test: every launch returns its counters to zero).  A stream
// found dirty stays refused until wino_stream_reset_scratch().
int wino_stream_check(wino_stream_t stream) {
  long nonzero = 0;
  if (int rc = wino_debug_tickets_in_use(stream, &nonzero)) return rc;   // synchronises the stream
  int dev = 0;
  WINO_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(g_ws_mu);
  SkScratch* ws = find_ws(dev, (hipStream_t)stream);
  if (!ws) return WINO_OK;
  if (nonzero) ws->failed = true;
  if (ws->failed || __atomic_load_n(ws->err, __ATOMIC_RELAXED) != 0) {
    set_error("the stream's ticket counters are in an unknown state (%ld non-zero): call wino_stream_reset_scratch()", nonzero);
    return WINO_E_STATE;
  }
  return WINO_OK;
}

// Test hook: overwrites one ticket counter of the stream's current scratch, as a launch that died mid-way
// would leave it.
int wino_debug_poison_ticket(wino_stream_t stream, long index, unsigned value) {
  int dev = 0;
  WINO_HIP(hipGetDevice(&dev));
  WINO_HIP(hipStreamSynchronize((hipStream_t)stream));
  unsigned* t = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_ws_mu);
    SkScratch* ws = find_ws(dev, (hipStream_t)stream);
    if (!ws || !ws->tickets || index < 0 || (size_t)index >= ws->n_tickets) {
      set_error("no ticket %ld in this stream's scratch", index);
      return WINO_E_ARG;
    }
    t = ws->tickets + index;
  }
  WINO_HIP(hipMemcpy(t, &value, sizeof(unsigned), hipMemcpyHostToDevice));
  return WINO_OK;
}

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
  g_last_status = hipSuccess;
  WINO_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return WINO_OK;
}

int wino_memcpy_d2h(void* dst, const void* src, size_t bytes) {
  g_last_status = hipSuccess;
  WINO_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return WINO_OK;
}

int wino_memcpy_d2d(void* dst, const void* src, size_t bytes) {
  g_last_status = hipSuccess;
  WINO_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToDevice));
  return WINO_OK;
}

int wino_device_synchronize(void) {
  g_last_status = hipSuccess;
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
