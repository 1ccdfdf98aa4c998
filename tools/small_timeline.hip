// Developer tool: where the 3x3 latency kernel's time goes at the reference's N = 1 (per-workgroup timeline).
// Launches the DIAG build of wino_f2_small_kernel<CT> (same source) back to back and prints, over the workgroups of
// the LAST launch, the median / 90th percentile of each phase, separately for the finishers (the workgroup that drew
// its block's last ticket) and the others, plus the spread of the workgroups' entry times.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Icuda-winograd_amd/csrc tools/small_timeline.hip -o tools/small_timeline
//   tools/small_timeline C N CT S
#include "wino_f2_small_kernel.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace wino::fused;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
namespace wino { void set_error(const char*, ...) {} int hip_fail(hipError_t, const char*) { return -1; } }

template <int CT>
static void run(int C, int N, int S) {
  const int K = C, nT16 = (N * 49 + 15) / 16, blocks = nT16 * (K / (16 * CT));
  float *in, *U, *b, *s, *out, *slabs; unsigned *tickets, *err; unsigned long long* dbg;
  CK(hipMalloc(&in, (size_t)N * 256 * C * 4)); CK(hipMalloc(&U, (size_t)16 * C * K * 4)); CK(hipMalloc(&b, K * 4)); CK(hipMalloc(&s, K * 4));
  CK(hipMalloc(&out, (size_t)N * 256 * K * 4)); CK(hipMalloc(&slabs, (size_t)blocks * 8 * 4096 * CT)); CK(hipMalloc(&tickets, blocks * 4));
  CK(hipMalloc(&err, 64)); CK(hipMalloc(&dbg, (size_t)blocks * S * 64));
  CK(hipMemset(tickets, 0, blocks * 4)); CK(hipMemset(err, 0, 64));
  std::vector<float> h((size_t)16 * C * K);
  for (auto& x : h) x = (float)rand() / (float)RAND_MAX - 0.5f;
  CK(hipMemcpy(U, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(in, h.data(), (size_t)N * 256 * C * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(b, h.data(), K * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(s, h.data() + K, K * 4, hipMemcpyHostToDevice));
  const SmallParams prm = {in, U, b, s, out, N, C, K, 1, slabs, tickets, err, dbg, Geo{}};
  for (int i = 0; i < 200; i++) hipLaunchKernelGGL((wino_f2_small_kernel<CT, false, true>), dim3(K / (16 * CT), nT16, S), dim3(256), 0, 0, prm);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> st((size_t)blocks * S * 8);
  CK(hipMemcpy(st.data(), dbg, st.size() * 8, hipMemcpyDeviceToHost));
  unsigned long long t0 = ~0ull, t1 = 0;
  for (size_t w = 0; w < (size_t)blocks * S; w++) { t0 = std::min(t0, st[w * 8]); t1 = std::max(t1, st[w * 8 + 7]); }
  const char* names[7] = {"entry -> first stage in LDS", "-> MFMAs + A^T m A done", "-> LDS level done", "-> slab drained",
                          "-> ticket drawn", "-> gather landed", "-> BN, stores drained"};
  printf("C = %d, N = %d, CT = %d, S = %d: %d workgroups; first entry -> last exit %.2f us\n", C, N, CT, S, blocks * S, (t1 - t0) / 100.0);
  for (int fin = 1; fin >= 0; fin--) {
    std::vector<double> ph[7], entry, exit_;
    for (size_t w = 0; w < (size_t)blocks * S; w++) {
      const unsigned long long* p = &st[w * 8];
      const bool finisher = S == 1 || p[6] > p[5];   // only a finisher stamps the gather
      if ((int)finisher != fin) continue;
      entry.push_back((p[0] - t0) / 100.0);
      const int last = finisher ? 7 : 5;
      exit_.push_back((p[last] - t0) / 100.0);
      if (S == 1) {   // no level 2: stamps 0..3, then 7
        for (int i = 0; i < 3; i++) ph[i].push_back((p[i + 1] - p[i]) / 100.0);
        ph[6].push_back((p[7] - p[3]) / 100.0);
      } else {
        for (int i = 0; i < last; i++) ph[i].push_back((p[i + 1] - p[i]) / 100.0);
      }
    }
    if (entry.empty()) continue;
    auto pct = [](std::vector<double> v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
    printf("  %s (%zu): entry at %.2f / %.2f us (median / p90), leaves at %.2f / %.2f\n", fin ? "finishers" : "others", entry.size(),
           pct(entry, 0.5), pct(entry, 0.9), pct(exit_, 0.5), pct(exit_, 0.9));
    for (int i = 0; i < 7; i++)
      if (!ph[i].empty()) printf("    %-32s %.2f / %.2f us\n", names[i], pct(ph[i], 0.5), pct(ph[i], 0.9));
  }
}

int main(int argc, char** argv) {
  const int C = argc > 1 ? atoi(argv[1]) : 256, N = argc > 2 ? atoi(argv[2]) : 1, CT = argc > 3 ? atoi(argv[3]) : 1, S = argc > 4 ? atoi(argv[4]) : 4;
  if (CT == 4) run<4>(C, N, S); else if (CT == 2) run<2>(C, N, S); else run<1>(C, N, S);
  return 0;
}
