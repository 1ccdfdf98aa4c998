// Developer tool: prices the parts of the fused Winograd kernel by timing ablated variants
// (see ABLATE in csrc/wino_f2_fused_kernel.h).  Not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -Iinclude -Icuda-winograd_amd/csrc tools/ablate_fused.hip -o tools/ablate_fused
#include "wino_f2_fused_kernel.h"

#include <cstdlib>
#include <algorithm>
#include <vector>

namespace wino { void set_error(const char*, ...) {} int hip_fail(hipError_t, const char*) { return -1; } }
using namespace wino::fused;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static float* g_slabs;
static unsigned* g_tickets;
static unsigned long long* g_dbg;   // stamps of the diagnostic builds
static unsigned* g_err;              // the library's dirty-counter word (device memory here)
static int g_grid = 256;   // logical workgroups (argv[2]); 0 = one whole item per workgroup

static int grid_for(int N, int K) {
  const int nTB = (N * 49 + TB - 1) / TB;
  if (g_grid) return g_grid;
  return nTB * (K / KB);   // one whole item per workgroup
}

template <int AB>
float run(const float* in, const float* U, const float* b, const float* s, float* out, int N, int C, int K, int reps) {
  CK(hipFuncSetAttribute((const void*)(wino_f2_fused_kernel<AB>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  const int nTB = (N * 49 + TB - 1) / TB;
  const int grid = grid_for(N, K);
  CK(hipMemset(g_tickets, 0, 65536 * 4));   // ablated variants may leave tickets behind
  const unsigned items = (unsigned)nTB * (K / KB), Tt = (items % grid) * (C / 8);
  // the library's tail placement (wino_f2_fused.hip: tail_groups, phase_order); WINO_TOOL_KP=0 gives round 2's scheme
  const char* kpenv = getenv("WINO_TOOL_KP");
  const int kblk = K / KB, kp = (kpenv && kpenv[0] == '0') || kblk <= 1 || grid % kblk ? 1 : kblk;
  const unsigned Tg = Tt / kp, Gp = grid / kp, q = Tg / Gp, rem = Tg % Gp;
  int P = 1, inv = 0, copies = (int)Gp;
  if (!(kpenv && kpenv[0] == '0') && rem == 0 && q > 0) {
    unsigned a = q % (C / 8), bb = C / 8;
    while (a) { const unsigned t = bb % a; bb = a; a = t; }
    const unsigned g = bb, PP = (C / 8) / g;
    if (PP > 1 && Gp % PP == 0) {
      const unsigned qq = (q / g) % PP;
      for (unsigned x = 1; x < PP; x++)
        if ((qq * x) % PP == 1) { P = (int)PP; inv = (int)x; copies = (int)(Gp / PP); break; }
    }
  }
  const FusedParams prm = {in, U, N, C, K, 1, nTB, (int)(items / grid), q, rem, kp, (int)Gp, P, inv, copies, wino::make_fastdiv((unsigned)kp), wino::make_fastdiv((unsigned)P), wino::make_fastdiv((unsigned)copies), Geo{}, b, s, out, g_slabs, g_tickets, g_err, g_dbg};
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; i++)
    hipLaunchKernelGGL((wino_f2_fused_kernel<AB>), dim3(grid), dim3(NTHREADS), LDS_BYTES, 0, prm);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; i++)
    hipLaunchKernelGGL((wino_f2_fused_kernel<AB>), dim3(grid), dim3(NTHREADS), LDS_BYTES, 0, prm);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / reps;
}

int main(int argc, char** argv) {
  const int C = argc > 1 ? atoi(argv[1]) : 256, K = C;
  if (argc > 2) g_grid = atoi(argv[2]);
  CK(hipMalloc(&g_slabs, (size_t)2 * 4096 * SLAB_BYTES));
  CK(hipMalloc(&g_tickets, 65536 * 4));
  CK(hipMalloc(&g_dbg, (size_t)4096 * 64 * 8));
  CK(hipMalloc(&g_err, 64));
  CK(hipMemset(g_err, 0, 64));
  std::vector<int> Ns = {1, 83, 128};
  const size_t maxN = 256;
  float *in, *U, *b, *s, *out;
  CK(hipMalloc(&in, maxN * 256 * C * 4)); CK(hipMalloc(&out, maxN * 256 * K * 4));
  CK(hipMalloc(&U, (size_t)16 * C * K * 4)); CK(hipMalloc(&b, K * 4)); CK(hipMalloc(&s, K * 4));
  std::vector<float> h(maxN * 256 * C);
  for (auto& x : h) x = (float)rand() / RAND_MAX - 0.5f;
  CK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(U, h.data(), (size_t)16 * C * K * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(b, h.data(), K * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(s, h.data() + K, K * 4, hipMemcpyHostToDevice));
  if (argc > 3 && argv[3][0] == 'w') {  // per-workgroup pass times of the stamped build, with each range's shape
    const int N = 128;
    run<0>(in, U, b, s, out, N, C, K, 3000);
    const int nTB = (N * 49 + TB - 1) / TB, wgs = grid_for(N, K), nch = C / 8;
    const unsigned items = (unsigned)nTB * (K / KB), ndp = items / wgs, Tt = (items % wgs) * nch, q = Tt / wgs, rem = Tt % wgs;
    run<16>(in, U, b, s, out, N, C, K, 20);
    std::vector<unsigned long long> st((size_t)wgs * 4);
    CK(hipMemcpy(st.data(), g_dbg, st.size() * 8, hipMemcpyDeviceToHost));
    printf("lg cycles tail_begin tail_len segments(tail) straddles\n");
    for (int l = 0; l < wgs; l++) {
      const unsigned t0 = sk_start(l, q, rem, wgs), t1 = sk_start(l + 1, q, rem, wgs);
      const int segs = t1 > t0 ? (int)((t1 - 1) / nch - t0 / nch + 1) : 0;
      printf("%d %llu %u %u %d %d ndp=%u\n", l, st[4 * l + 2] - st[4 * l], t0, t1 - t0, segs, (int)(t0 % nch != 0) + (int)(t1 % nch != 0), ndp);
    }
    return 0;
  }
  if (argc > 3 && argv[3][0] == 't') {   // timeline: where a launch's wall time goes (chip-wide 100 MHz stamps)
    const int N = argc > 4 ? atoi(argv[4]) : 128;
    run<0>(in, U, b, s, out, N, C, K, 3000);
    const float us_prod = run<0>(in, U, b, s, out, N, C, K, 200);
    const float us_tl = run<32768>(in, U, b, s, out, N, C, K, 200);   // the stamps of the LAST of 200 back-to-back launches stay
    const int wgs = grid_for(N, K);
    std::vector<unsigned long long> st((size_t)wgs * 8);
    CK(hipMemcpy(st.data(), g_dbg, st.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, tend = 0;
    std::vector<double> entry, first, lastep, exitt, pro, epi;
    for (int l = 0; l < wgs; l++) { t0 = std::min(t0, st[8 * l]); tend = std::max(tend, st[8 * l + 3]); }
    for (int l = 0; l < wgs; l++) {
      entry.push_back((st[8 * l] - t0) * 0.01); first.push_back((st[8 * l + 1] - t0) * 0.01);
      lastep.push_back((st[8 * l + 2] - t0) * 0.01); exitt.push_back((st[8 * l + 3] - t0) * 0.01);
      pro.push_back((st[8 * l + 1] - st[8 * l]) * 0.01); epi.push_back((st[8 * l + 3] - st[8 * l + 2]) * 0.01);
    }
    auto pr = [&](const char* name, std::vector<double> v) {
      std::sort(v.begin(), v.end());
      printf("  %-34s min %7.2f  p10 %7.2f  median %7.2f  p90 %7.2f  max %7.2f us\n", name, v[0], v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10], v.back());
    };
    printf("%s C=%d N=%d grid=%d: product %.2f us per launch, timeline build %.2f; first entry -> last exit %.2f us (10 ns ticks)\n", argv[0], C, N, wgs, us_prod, us_tl, (tend - t0) * 0.01);
    pr("entry (after the first entry)", entry); pr("first MFMA", first); pr("start of the last epilogue", lastep); pr("exit", exitt);
    pr("entry -> first MFMA", pro); pr("last epilogue -> exit", epi);
    {
      std::vector<double> req, land;
      for (int l = 0; l < wgs; l++) { req.push_back((st[8 * l + 6] - st[8 * l]) * 0.01); land.push_back((st[8 * l + 7] - st[8 * l]) * 0.01); }
      pr("entry -> first stage requested", req); pr("entry -> first stage landed", land);
    }
    if (argc > 5) {   // per workgroup, with the shape of its tail range
      const int nTB = (N * 49 + TB - 1) / TB, nch = C / 8;
      const unsigned items = (unsigned)nTB * (K / KB), Tt = (items % wgs) * nch, q = Tt / wgs, rem = Tt % wgs;
      printf("lg first_chunk len segs | first_mfma last_epi_start exit epi_len\n");
      for (int l = 0; l < wgs; l++) {
        const unsigned a = sk_start(l, q, rem, wgs), b2 = sk_start(l + 1, q, rem, wgs);
        const int segs = b2 > a ? (int)((b2 - 1) / nch - a / nch + 1) : 0;
        printf("%3d %2u %2u %d | %6.2f %7.2f %7.2f %5.2f\n", l, a % nch, b2 - a, segs, first[l], lastep[l], exitt[l], epi[l]);
      }
    }
    return 0;
  }
  if (argc > 3 && argv[3][0] == 'h') {   // timing only: what the parts of the A path cost
    run<0>(in, U, b, s, out, 128, C, K, 3000);
    for (int i = 0; i < 3; i++)
      printf("full %.1f   no v_point %.1f   no tmp_col + v_point (reads only) %.1f   no A path %.1f us\n", run<0>(in, U, b, s, out, 128, C, K, 200),
             run<4096>(in, U, b, s, out, 128, C, K, 200), run<8192>(in, U, b, s, out, 128, C, K, 200), run<32>(in, U, b, s, out, 128, C, K, 200));
    return 0;
  }
  if (argc > 3 && argv[3][0] == 'q') {  // quick mode: just the product kernel at N = 128, three trials of 50 launches
    run<0>(in, U, b, s, out, 128, C, K, 3000);   // clock ramp: ~0.4 s of the same kernel (see bench.py, preheat)
    float t[3];
    for (int i = 0; i < 3; i++) t[i] = run<0>(in, U, b, s, out, 128, C, K, 200);
    std::sort(t, t + 3);
    const int nTB = (128 * 49 + TB - 1) / TB, wgs = grid_for(128, K);
    const double iters = (double)nTB * (K / KB) * (C / 8) / wgs;
    run<16>(in, U, b, s, out, 128, C, K, 3);
    std::vector<unsigned long long> st((size_t)wgs * 4);
    CK(hipMemcpy(st.data(), g_dbg, st.size() * 8, hipMemcpyDeviceToHost));
    double cyc = 0, rt = 0, cmax = 0;
    for (int i = 0; i < wgs; i++) { const double c = (double)(st[4 * i + 2] - st[4 * i]); cyc += c; rt += (double)(st[4 * i + 3] - st[4 * i + 1]); cmax = std::max(cmax, c); }
    printf("%s C=%d grid=%d: %.1f / %.1f / %.1f us   loop+epilogues: %.1f cycles per MFMA per SIMD (slowest workgroup %.1f) at %.3f GHz\n",
           argv[0], C, wgs, t[0], t[1], t[2], cyc / wgs / iters / 128.0, cmax / iters / 128.0, cyc / rt * 0.1);
    return 0;
  }
  printf("C=K=%d   us per launch; items = K/64 * ceil(N*49/64); grid = %d logical workgroups (0: one item each)\n", C, g_grid);
  printf("%6s %6s %6s | %8s %8s %8s %8s %8s %8s %8s %8s %8s %8s\n", "N", "items", "grid", "full", "noRawDMA", "noUDMA", "noDMA", "noMFMA",
         "noBarr", "noStore", "noSlab", "noA", "noB");
  for (int N : Ns) {
    const int items = (K / 64) * ((N * 49 + 63) / 64);
    printf("%6d %6d %6d | %8.1f %8.1f %8.1f %8.1f %8.1f %8.1f %8.1f %8.1f %8.1f %8.1f\n", N, items, grid_for(N, K),
           run<0>(in, U, b, s, out, N, C, K, 20), run<1>(in, U, b, s, out, N, C, K, 20),
           run<2>(in, U, b, s, out, N, C, K, 20), run<3>(in, U, b, s, out, N, C, K, 20),
           run<4>(in, U, b, s, out, N, C, K, 20), run<8>(in, U, b, s, out, N, C, K, 20),
           run<512>(in, U, b, s, out, N, C, K, 20), run<1024>(in, U, b, s, out, N, C, K, 20),
           run<32>(in, U, b, s, out, N, C, K, 20), run<64>(in, U, b, s, out, N, C, K, 20));
  }
  {  // what the non-MFMA side costs on its own (N = 128): everything below skips the MFMAs
    const int N = 128;
    printf("noMFMA and ... : alone %.1f  noDMA %.1f  noBarr %.1f  noA %.1f  noB %.1f  noA+noB %.1f  noDMA+noA+noB %.1f  noDMA+noBarr+noA+noB %.1f  +noStore+noSlab %.1f\n",
           run<4>(in, U, b, s, out, N, C, K, 20), run<4 | 3>(in, U, b, s, out, N, C, K, 20), run<4 | 8>(in, U, b, s, out, N, C, K, 20),
           run<4 | 32>(in, U, b, s, out, N, C, K, 20), run<4 | 64>(in, U, b, s, out, N, C, K, 20), run<4 | 96>(in, U, b, s, out, N, C, K, 20),
           run<4 | 96 | 3>(in, U, b, s, out, N, C, K, 20), run<4 | 96 | 3 | 8>(in, U, b, s, out, N, C, K, 20),
           run<4 | 96 | 3 | 8 | 512 | 1024>(in, U, b, s, out, N, C, K, 20));
    printf("MFMA and ...   : noDMA+noBarr %.1f  noDMA+noBarr+noA+noB %.1f  +noStore+noSlab %.1f\n",
           run<3 | 8>(in, U, b, s, out, N, C, K, 20), run<3 | 8 | 96>(in, U, b, s, out, N, C, K, 20),
           run<3 | 8 | 96 | 512 | 1024>(in, U, b, s, out, N, C, K, 20));
  }
  {  // in-kernel clock of the main loop (diagnostic build, ABLATE bit 16)
    const int N = 128;
    const int nTB = (N * 49 + TB - 1) / TB, wgs = grid_for(N, K);
    const double iters = (double)nTB * (K / KB) * (C / 8) / wgs;   // chunk iterations per workgroup
    run<16>(in, U, b, s, out, N, C, K, 3);
    std::vector<unsigned long long> st((size_t)wgs * 4);
    CK(hipMemcpy(st.data(), g_dbg, st.size() * 8, hipMemcpyDeviceToHost));
    double cyc = 0, rt = 0, cmin = 1e30, cmax = 0;
    for (int i = 0; i < wgs; i++) { const double c = (double)(st[4 * i + 2] - st[4 * i]); cyc += c; rt += (double)(st[4 * i + 3] - st[4 * i + 1]); cmin = std::min(cmin, c); cmax = std::max(cmax, c); }
    printf("main loop, N=128: in-kernel clock %.3f GHz; cycles per WG pass mean %.0f min %.0f max %.0f (= %.1f cycles per MFMA per SIMD)\n",
           cyc / rt * 0.1, cyc / wgs, cmin, cmax, cyc / wgs / iters / 128.0);
  }
  {  // per-wave phase stamps (ABLATE bit 2048): barrier+DMA wait vs compute, in shader cycles
    const int N = 128;
    const int nTB = (N * 49 + TB - 1) / TB, wgs = grid_for(N, K);
    const double iters = (double)nTB * (K / KB) * (C / 8) / wgs;
    run<2048>(in, U, b, s, out, N, C, K, 2);
    std::vector<unsigned long long> st((size_t)wgs * 64);
    CK(hipMemcpy(st.data(), g_dbg, st.size() * 8, hipMemcpyDeviceToHost));
    double sum[8][7] = {{0}};
    for (int i = 0; i < wgs; i++) for (int w = 0; w < 8; w++) for (int k = 0; k < 7; k++) sum[w][k] += st[(i * 8 + w) * 8 + k];
    printf("per chunk iteration (%.2f per workgroup), mean over %d workgroups; stamps add overhead, read the SHARES:\n", iters, wgs);
    for (int w = 0; w < 8; w++)
      printf("  wave %d: wait(vmcnt+barrier) %6.0f  compute %6.0f cycles per iteration (wait share %4.1f%%) | epilogues per workgroup %7.0f cycles = barrier %6.0f + AtmA %6.0f + slab/ticket %6.0f + gather/finalize %6.0f\n", w,
             sum[w][0] / wgs / iters, sum[w][1] / wgs / iters, 100.0 * sum[w][0] / (sum[w][0] + sum[w][1]), sum[w][2] / wgs,
             sum[w][3] / wgs, sum[w][4] / wgs, sum[w][5] / wgs, sum[w][6] / wgs);
  }
  return 0;
}
