#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cmath>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// A: [16][128] bytes (e4m3), B: [16][128] bytes (row n, k) ; D[m][n] = sum_k A[m][k]*B[n][k]
__global__ void probe(const uint8_t* A, const uint8_t* B, float* D, int variant) {
  int lane = threadIdx.x;
  int r = lane & 15, q = lane >> 4;
  i32x8 a, b;
  const int* ap = reinterpret_cast<const int*>(A + r * 128 + q * 32);
  const int* bp = reinterpret_cast<const int*>(B + r * 128 + q * 32);
  for (int j = 0; j < 8; ++j) { a[j] = ap[j]; b[j] = bp[j]; }
  f32x4 c = {0, 0, 0, 0};
  // (a, b, c, cbsz(A fmt), blgp(B fmt), opsel_a, scale_a, opsel_b, scale_b)
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  for (int j = 0; j < 4; ++j) D[lane * 4 + j] = c[j];
}
static float e4m3(uint8_t v) {
  int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float x = e == 0 ? ldexpf(m / 8.f, -6) : ldexpf(1 + m / 8.f, e - 7);
  return s ? -x : x;
}
int main() {
  std::vector<uint8_t> A(16 * 128), B(16 * 128);
  uint32_t s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
  for (auto& v : A) { v = rnd() % 0x58; if (rnd() & 1) v |= 0x80; }   // finite moderate values
  for (auto& v : B) { v = rnd() % 0x58; if (rnd() & 1) v |= 0x80; }
  uint8_t *dA, *dB; float* dD;
  (void)hipMalloc(&dA, A.size()); (void)hipMalloc(&dB, B.size()); (void)hipMalloc(&dD, 256 * 4);
  (void)hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  probe<<<1, 64>>>(dA, dB, dD, 0);
  std::vector<float> D(256);
  (void)hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
  // expected: if operand map is row=lane&15, k=32*(lane>>4)+j  and C/D col=lane&15,row=4*(lane>>4)+reg:
  // with first operand "a" = rows of A indexed by... test both orientations
  double e1 = 0, e2 = 0;
  for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 4; ++j) {
    int col = lane & 15, row = 4 * (lane >> 4) + j;
    double r1 = 0, r2 = 0;
    for (int k = 0; k < 128; ++k) { r1 += (double)e4m3(A[row * 128 + k]) * e4m3(B[col * 128 + k]); r2 += (double)e4m3(A[col * 128 + k]) * e4m3(B[row * 128 + k]); }
    e1 = fmax(e1, fabs(r1 - D[lane * 4 + j])); e2 = fmax(e2, fabs(r2 - D[lane * 4 + j]));
  }
  printf("err if D[row][col]=A[row]·B[col]: %g ; if D[row][col]=A[col]·B[row]: %g ; sample D0=%g\n", e1, e2, D[0]);
  return 0;
}
