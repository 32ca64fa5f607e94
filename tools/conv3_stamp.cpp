// Diagnostic build of the fused-tap 3x3 kernel with in-kernel shader-clock stamps (tuning aid, not part of the library):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DLO_STAMPS -x hip tools/conv3_stamp.cpp lunaris_orion_amd/csrc/lo_conv3.hip \
//         lunaris_orion_amd/csrc/lo_conv.hip -x c++ lunaris_orion_amd/csrc/lo_util.cpp -o /tmp/conv3_stamp && /tmp/conv3_stamp
#include "../lunaris_orion_amd/csrc/lo_internal.h"
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
extern unsigned long long* g_lo_conv3_stamps;
int main(int argc, char** argv) {
  int B = argc > 1 ? atoi(argv[1]) : 64, H = argc > 2 ? atoi(argv[2]) : 128, C = argc > 3 ? atoi(argv[3]) : 128;
  LoGeom g;
  if (lo_make_geom(&g, LO_CONV3_S1, B, H, H, C, C)) { printf("geom: %s\n", lo_get_error()); return 1; }
  size_t nx = (size_t)B * H * H * C, nw = (size_t)C * 9 * C;
  std::vector<_Float16> hx(nx), hw(nw);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto& v : hx) v = (_Float16)rnd();
  for (auto& v : hw) v = (_Float16)(0.1f * rnd());
  f16 *x, *w, *o; float* bias; unsigned long long* st;
  hipMalloc(&x, nx * 2); hipMalloc(&w, nw * 2); hipMalloc(&o, nx * 2); hipMalloc(&bias, C * 4);
  hipMemcpy(x, hx.data(), nx * 2, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), nw * 2, hipMemcpyHostToDevice);
  hipMemset(bias, 0, C * 4);
  const int bn = C % 128 == 0 ? 128 : 64;   // LO_HALO=3 picks 16x16 pixels x 64 channels for C = 64 (one patch buffer, two workgroups per CU)
  int tiles = B * (H / 16) * (H / 16) * (C / bn);
  hipMalloc(&st, (size_t)tiles * 8 * 16 * 8); hipMemset(st, 0, (size_t)tiles * 8 * 16 * 8);
  setenv("LO_HALO", "3", 1);
  g_lo_conv3_stamps = st;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) lo_conv_run(g, x, w, bias, nullptr, o, nullptr, nullptr, 1, 0);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  const int it = 10;
  for (int i = 0; i < it; ++i) if (lo_conv_run(g, x, w, bias, nullptr, o, nullptr, nullptr, 1, 0)) { printf("run: %s\n", lo_get_error()); return 1; }
  hipEventRecord(e1, 0); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= it;
  double fl = 2.0 * nx * 9 * C;
  printf("B=%d H=%d C=%d: %.1f us  %.1f TFLOP/s  tiles=%d\n", B, H, C, ms * 1e3, fl / ms / 1e9, tiles);
  std::vector<unsigned long long> hs((size_t)tiles * 8 * 16);
  hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
  const char* nm[7] = {"prologue", "loop", "epilogue", "sum R", "sum wait1", "sum M", "sum wait2"};
  for (int grp = 0; grp < 2; ++grp) {
    std::vector<double> v[7];
    for (int t = 0; t < tiles; ++t)
      for (int wv = grp * 4; wv < grp * 4 + 4; ++wv) {
        unsigned long long* d = &hs[((size_t)t * 8 + wv) * 16];
        v[0].push_back((double)(d[1] - d[0])); v[1].push_back((double)(d[2] - d[1])); v[2].push_back((double)(d[3] - d[2]));
        for (int k = 0; k < 4; ++k) v[3 + k].push_back((double)d[4 + k]);
      }
    printf("group %d (median shader clocks per workgroup):", grp);
    for (int k = 0; k < 7; ++k) { std::sort(v[k].begin(), v[k].end()); printf("  %s %.0f", nm[k], v[k][v[k].size() / 2]); }
    printf("\n");
  }
  // workgroup lifetime and the spread of start times
  std::vector<double> life;
  for (int t = 0; t < tiles; ++t) life.push_back((double)(hs[(size_t)t * 8 * 16 + 3] - hs[(size_t)t * 8 * 16 + 0]));
  std::sort(life.begin(), life.end());
  printf("workgroup lifetime median %.0f clocks; kernel %.1f us; %.1f workgroups per CU over the launch\n", life[life.size() / 2], ms * 1e3, tiles / 256.0);
  return 0;
}
