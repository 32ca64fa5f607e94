// SelfAttention2d (lunar_generate.py:56-78): out = gamma * (V softmax(Q^T K)^T) + x on a [B, C, H, W] map.
// The reference defines this module but never instantiates it, so it is built as a standalone op with its
// own parity test, not as part of the VAE graph.  fp32 throughout (module-level parity 1e-5); two kernels:
//   lo_attn_project : q, k (C/8 channels) and v (C channels) = 1x1 convs of x, channel-major [B][c][N]
//   lo_attn_fused   : per (64-query block, 64-channel group): pass A streams the keys for the row max / sum of
//                     exp, pass B recomputes the scores, normalises and accumulates P.V in registers; nothing of size
//                     N x N is ever written to memory.
#include "lo_internal.h"

__global__ void lo_attn_project_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                       float* __restrict__ y, int C, int Co, int N) {
  // y[bi][co][n] = b[co] + sum_ci w[co][ci] * x[bi][ci][n]
  const int n = blockIdx.x * blockDim.x + threadIdx.x, co = blockIdx.y, bi = blockIdx.z;
  if (n >= N) return;
  const float* xp = x + (size_t)bi * C * N + n;
  const float* wp = w + (size_t)co * C;
  float acc = b[co];
  for (int ci = 0; ci < C; ++ci) acc += wp[ci] * xp[(size_t)ci * N];
  y[((size_t)bi * Co + co) * N + n] = acc;
}

#define AT_Q 64   // queries per workgroup
#define AT_K 64   // keys per block
#define AT_C 64   // value channels per workgroup

__global__ __launch_bounds__(256) void lo_attn_fused_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ v, const float* __restrict__ x,
                                                            const float* __restrict__ gamma, float* __restrict__ out,
                                                            int C, int D, int N) {
  // q, k: [B][D][N]; v, x, out: [B][C][N].  thread = (query i = tid & 63, part = tid >> 6)
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* s_q = sm;                         // [D][64]
  float* s_k = s_q + D * AT_Q;             // [D][64]
  float* s_p = s_k + D * AT_K;             // [64 i][65]
  float* s_v = s_p + AT_Q * 65;            // [64 c][65]
  float* s_red = s_v + AT_C * 65;          // [4][64]
  const int tid = threadIdx.x, i = tid & 63, part = tid >> 6;
  const int q0 = blockIdx.x * AT_Q, c0 = blockIdx.y * AT_C, bi = blockIdx.z;
  const float* qb = q + (size_t)bi * D * N;
  const float* kb = k + (size_t)bi * D * N;
  const float* vb = v + (size_t)bi * C * N;
  for (int e = tid; e < D * AT_Q; e += 256) s_q[e] = qb[(size_t)(e / AT_Q) * N + q0 + (e % AT_Q)];
  // ---- pass A: running max and sum of exp over all keys (each part scans 16 keys of every block)
  float m = -INFINITY, l = 0.f;
  for (int j0 = 0; j0 < N; j0 += AT_K) {
    __syncthreads();
    for (int e = tid; e < D * AT_K; e += 256) s_k[e] = kb[(size_t)(e / AT_K) * N + j0 + (e % AT_K)];
    __syncthreads();
    for (int jj = part * 16; jj < part * 16 + 16; ++jj) {
      float s = 0.f;
      for (int c = 0; c < D; ++c) s += s_q[c * AT_Q + i] * s_k[c * AT_K + jj];
      float mn = fmaxf(m, s);
      l = l * __expf(m - mn) + __expf(s - mn);
      m = mn;
    }
  }
  // combine the 4 parts of a query
  s_red[part * 64 + i] = m;
  __syncthreads();
  float mt = fmaxf(fmaxf(s_red[i], s_red[64 + i]), fmaxf(s_red[128 + i], s_red[192 + i]));
  __syncthreads();
  s_red[part * 64 + i] = l * __expf(m - mt);
  __syncthreads();
  const float inv_l = 1.f / (s_red[i] + s_red[64 + i] + s_red[128 + i] + s_red[192 + i]);
  // ---- pass B: P = softmax rows, out[c][i] += sum_j v[c][j] P[i][j]; this thread owns 16 channels
  float acc[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) acc[c] = 0.f;
  for (int j0 = 0; j0 < N; j0 += AT_K) {
    __syncthreads();
    for (int e = tid; e < D * AT_K; e += 256) s_k[e] = kb[(size_t)(e / AT_K) * N + j0 + (e % AT_K)];
    for (int e = tid; e < AT_C * AT_K; e += 256) s_v[(e / AT_K) * 65 + (e % AT_K)] = vb[(size_t)(c0 + e / AT_K) * N + j0 + (e % AT_K)];
    __syncthreads();
    for (int jj = part * 16; jj < part * 16 + 16; ++jj) {
      float s = 0.f;
      for (int c = 0; c < D; ++c) s += s_q[c * AT_Q + i] * s_k[c * AT_K + jj];
      s_p[i * 65 + jj] = __expf(s - mt) * inv_l;
    }
    __syncthreads();
    for (int jj = 0; jj < AT_K; ++jj) {
      float p = s_p[i * 65 + jj];
#pragma unroll
      for (int c = 0; c < 16; ++c) acc[c] += s_v[(part * 16 + c) * 65 + jj] * p;
    }
  }
  const float gm = gamma[0];
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    size_t o = ((size_t)bi * C + c0 + part * 16 + c) * N + q0 + i;
    out[o] = gm * acc[c] + x[o];
  }
}

int lo_selfattn2d_fwd(const float* x, const float* wq, const float* bq, const float* wk, const float* bk, const float* wv,
                      const float* bv, const float* gamma, float* q, float* k, float* v, float* out, int B, int C, int N,
                      hipStream_t st) {
  LO_REQUIRE(C % 64 == 0 && N % 64 == 0, "lo_selfattn2d_fwd: C=%d and H*W=%d must be multiples of 64", C, N);
  const int D = C / 8;
  LO_REQUIRE(D <= 64, "lo_selfattn2d_fwd: C/8 = %d > 64 is not supported", D);
  dim3 blk(64), gq((N + 63) / 64, D, B), gv((N + 63) / 64, C, B);
  hipLaunchKernelGGL(lo_attn_project_kernel, gq, blk, 0, st, x, wq, bq, q, C, D, N);
  hipLaunchKernelGGL(lo_attn_project_kernel, gq, blk, 0, st, x, wk, bk, k, C, D, N);
  hipLaunchKernelGGL(lo_attn_project_kernel, gv, blk, 0, st, x, wv, bv, v, C, C, N);
  LO_LAUNCH_CHECK("attn_project");
  size_t lds = (size_t)(2 * D * 64 + 64 * 65 + 64 * 65 + 256) * sizeof(float);
  LoProfScope _p("lo_attn_fused", 2.0 * B * N * (double)N * (2.0 * D + C), 4.0 * B * N * (2.0 * D + 3.0 * C), st);
  hipLaunchKernelGGL(lo_attn_fused_kernel, dim3(N / 64, C / 64, B), dim3(256), lds, st, q, k, v, x, gamma, out, C, D, N);
  LO_LAUNCH_CHECK("attn_fused");
  return LO_OK;
}
