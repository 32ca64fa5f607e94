// Diagnostic build of the multi-tap weight-gradient kernel with in-kernel shader-clock stamps (tuning aid):
//   tools/build_wgrad3_stamp.sh && /tmp/wgrad3_stamp 64 64 64      (B H C)
#include "../lunaris_orion_amd/csrc/lo_internal.h"
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
extern unsigned long long* g_lo_wgrad3_stamps;
int main(int argc, char** argv) {
  int B = argc > 1 ? atoi(argv[1]) : 64, H = argc > 2 ? atoi(argv[2]) : 64, C = argc > 3 ? atoi(argv[3]) : 64;
  LoGeom g;
  if (lo_make_geom(&g, LO_CONV3_S1, B, H, H, C, C)) { printf("geom: %s\n", lo_get_error()); return 1; }
  size_t nx = (size_t)B * H * H * C;
  std::vector<_Float16> hx(nx), hd(nx);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto& v : hx) v = (_Float16)rnd();
  for (auto& v : hd) v = (_Float16)(0.1f * rnd());
  f16 *x, *dy; float *slab, *grad; unsigned long long* st;
  int ns = lo_wgrad3_nsplit(g);
  size_t welems = (size_t)C * 9 * C;
  hipMalloc(&x, nx * 2); hipMalloc(&dy, nx * 2); hipMalloc(&slab, (size_t)(ns + 1) * welems * 4); hipMalloc(&grad, welems * 4);
  hipMemcpy(x, hx.data(), nx * 2, hipMemcpyHostToDevice); hipMemcpy(dy, hd.data(), nx * 2, hipMemcpyHostToDevice);
  int wgs = (C / 64) * (C / 64) * ns;
  hipMalloc(&st, (size_t)wgs * 8 * 16 * 8); hipMemset(st, 0, (size_t)wgs * 8 * 16 * 8);
  g_lo_wgrad3_stamps = st;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int nsplit = 0;
  for (int i = 0; i < 3; ++i) lo_wgrad3_run(g, x, dy, slab, 0, &nsplit);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  const int it = 10;
  for (int i = 0; i < it; ++i) lo_wgrad3_run(g, x, dy, slab, 0, &nsplit);
  hipEventRecord(e1, 0); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= it;
  printf("B=%d H=%d C=%d: %.1f us (kernel only) %.1f TFLOP/s  workgroups=%d nsplit=%d\n", B, H, C, ms * 1e3, 2.0 * nx * 9 * C / ms / 1e9, wgs, nsplit);
  std::vector<unsigned long long> hs((size_t)wgs * 8 * 16);
  hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
  const char* nm[7] = {"prologue", "loop", "exchange", "store(g0)", "sum wait", "sum issue", "sum compute"};
  for (int grp = 0; grp < 2; ++grp) {
    std::vector<double> v[7];
    for (int t = 0; t < wgs; ++t)
      for (int wv = grp * 4; wv < grp * 4 + 4; ++wv) {
        unsigned long long* d = &hs[((size_t)t * 8 + wv) * 16];
        v[0].push_back((double)(d[1] - d[0])); v[1].push_back((double)(d[2] - d[1])); v[2].push_back((double)(d[3] - d[2]));
        v[3].push_back(grp == 0 ? (double)(d[7] - d[3]) : 0.0);
        for (int k = 0; k < 3; ++k) v[4 + k].push_back((double)d[4 + k]);
      }
    printf("group %d (median shader clocks):", grp);
    for (int k = 0; k < 7; ++k) { std::sort(v[k].begin(), v[k].end()); printf("  %s %.0f", nm[k], v[k][v[k].size() / 2]); }
    printf("\n");
  }
  return 0;
}
