// The two convolutions of the VAE that are NOT dense contractions as direct, HBM-bound
// kernels:
//   first conv  Conv2d(3->64, k3, s2, p1) on fp32 NCHW input (lunar_generate.py:95)  K = 27
//   final conv  Conv2d(32->3, k3, s1, p1) + tanh (+ MSE partial sums)   (lunar_generate.py:192,227-228;
//               train_hybrid.py:859)                                                  N = 3
// plus their weight / data gradients.  fp32 VALU math, LDS-staged input tiles, coalesced NHWC fp16 stores.
#include "lo_common.h"

// =============================================================================================
// first conv forward: x fp32 [B,3,128,128] -> v fp16 NHWC [B,64,64,64] (+bias) + GN partials [B][64][8][2]
// one workgroup = one output row (64 pixels x 64 channels)
// =============================================================================================
#define FC_W 128
#define FC_OW 64
#define FC_CO 64

__device__ __forceinline__ void fc_stage_rows(const float* __restrict__ x, float (*xs)[3][FC_W + 2], int n, int oy, int tid) {
  // xs[ci][r][1 + ix] = x[n][ci][2*oy - 1 + r][ix], zero outside; column 0 is the left halo (ix = -1)
  for (int i = tid; i < 3 * 3 * (FC_W + 2); i += 256) {
    int col = i % (FC_W + 2);
    int r = (i / (FC_W + 2)) % 3;
    int ci = i / (3 * (FC_W + 2));
    int iy = 2 * oy - 1 + r, ix = col - 1;
    float v = 0.f;
    if ((unsigned)iy < 128u && (unsigned)ix < 128u) v = x[(((size_t)n * 3 + ci) * 128 + iy) * 128 + ix];
    xs[ci][r][col] = v;
  }
}

__global__ __launch_bounds__(256) void lo_first_conv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ bias, f16* __restrict__ v,
                                                                float* __restrict__ gn_partial) {
  __shared__ float xs[3][3][FC_W + 2];
  __shared__ float ws[27][FC_CO];
  __shared__ float red[256 * 4];
  const int tid = threadIdx.x, oy = blockIdx.x, n = blockIdx.y;
  fc_stage_rows(x, xs, n, oy, tid);
  for (int i = tid; i < 27 * FC_CO; i += 256) {
    int co = i % FC_CO, k = i / FC_CO;
    ws[k][co] = w[co * 27 + k];  // canonical [co][ci][r][s] -> k = ci*9 + r*3 + s
  }
  __syncthreads();
  const int px = tid >> 2, cg = tid & 3;
  float acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = bias[cg * 16 + j];
#pragma unroll
  for (int ci = 0; ci < 3; ++ci)
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        float xv = xs[ci][r][2 * px + s];
        const float* wr = &ws[ci * 9 + r * 3 + s][cg * 16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] += xv * wr[j];
      }
  f16x8 h0, h1;
  float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    h0[j] = (f16)acc[j];
    h1[j] = (f16)acc[8 + j];
    float a = (float)h0[j], b = (float)h1[j];
    s0 += a; q0 += a * a; s1 += b; q1 += b * b;
  }
  f16* dst = v + (((size_t)n * 64 + oy) * 64 + px) * FC_CO + cg * 16;
  *reinterpret_cast<f16x8*>(dst) = h0;
  *reinterpret_cast<f16x8*>(dst + 8) = h1;
  red[tid * 4 + 0] = s0; red[tid * 4 + 1] = q0; red[tid * 4 + 2] = s1; red[tid * 4 + 3] = q1;
  __syncthreads();
  if (tid < 16) {
    int grp = tid >> 1, which = tid & 1;       // group = 8 channels: cg = grp>>1, half = grp&1
    int cgx = grp >> 1, half = grp & 1;
    float tot = 0.f;
    for (int p = 0; p < 64; ++p) tot += red[(p * 4 + cgx) * 4 + half * 2 + which];
    gn_partial[(((size_t)n * 64 + oy) * 8 + grp) * 2 + which] = tot;
  }
}

// first conv weight gradient: partial[wg][co*27 + k] over 8 output rows per workgroup
__global__ __launch_bounds__(256) void lo_first_conv_wgrad_kernel(const float* __restrict__ x, const f16* __restrict__ dv,
                                                                  float* __restrict__ partial) {
  __shared__ float xs[3][3][FC_W + 2];
  __shared__ f16 ds[64][FC_CO];
  __shared__ float red[4][FC_CO][27];
  const int tid = threadIdx.x, n = blockIdx.y, rb = blockIdx.x;  // rb: block of 8 output rows
  const int co = tid & 63, part = tid >> 6;
  float acc[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) acc[k] = 0.f;
  for (int oy = rb * 8; oy < rb * 8 + 8; ++oy) {
    __syncthreads();
    fc_stage_rows(x, xs, n, oy, tid);
    const f16* src = dv + (((size_t)n * 64 + oy) * 64) * FC_CO;
    for (int i = tid; i < 64 * FC_CO / 8; i += 256)
      reinterpret_cast<f16x8*>(&ds[0][0])[i] = reinterpret_cast<const f16x8*>(src)[i];
    __syncthreads();
    for (int p = part * 16; p < part * 16 + 16; ++p) {
      float d = (float)ds[p][co];
#pragma unroll
      for (int ci = 0; ci < 3; ++ci)
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int s = 0; s < 3; ++s) acc[ci * 9 + r * 3 + s] += d * xs[ci][r][2 * p + s];
    }
  }
#pragma unroll
  for (int k = 0; k < 27; ++k) red[part][co][k] = acc[k];
  __syncthreads();
  float* out = partial + ((size_t)n * gridDim.x + rb) * (FC_CO * 27);
  for (int i = tid; i < FC_CO * 27; i += 256) {
    int c = i / 27, k = i % 27;
    out[i] = red[0][c][k] + red[1][c][k] + red[2][c][k] + red[3][c][k];
  }
}

// generic deterministic column sum:  out[j] = scale * sum_k partial[k*stride + j]
// block = 16 columns x 16 row-lanes; each lane sums rows lane, lane+16, ... then the 16 lane sums are added in order
__global__ __launch_bounds__(256) void lo_colsum_kernel(const float* __restrict__ partial, float* __restrict__ out, int nrow,
                                                        int ncol, int stride, float scale) {
  __shared__ float red[16][17];
  const int c = threadIdx.x & 15, r = threadIdx.x >> 4;
  const int j = blockIdx.x * 16 + c;
  float t = 0.f;
  if (j < ncol)
    for (int k = r; k < nrow; k += 16) t += partial[(size_t)k * stride + j];
  red[r][c] = t;
  __syncthreads();
  if (r == 0 && j < ncol) {
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) tot += red[q][c];
    out[j] = tot * scale;
  }
}

// =============================================================================================
// final conv: a4 fp16 NHWC [B,128,128,32] -> recon fp32 NCHW [B,3,128,128] = tanh(conv + bias)
// workgroup = 16x16 output pixels.  Optional MSE partial: mse_partial[wg] = sum (recon - target)^2
// =============================================================================================
#define LC_T 16
#define LC_TP (LC_T + 2)
#define LC_CI 32
#define LC_PITCH 80   // bytes per staged pixel (64 B of channels + 16 B pad)

__device__ __forceinline__ void lc_stage_a4(const f16* __restrict__ a4, unsigned char* tile, int n, int ty, int tx, int tid) {
  // tile[(y*18 + x)] = a4[n][ty*16 - 1 + y][tx*16 - 1 + x][0..31], zero outside
  for (int i = tid; i < LC_TP * LC_TP * 4; i += 256) {
    int ch = i & 3, p = i >> 2;
    int y = p / LC_TP, xx = p % LC_TP;
    int iy = ty * LC_T - 1 + y, ix = tx * LC_T - 1 + xx;
    u32x4 v = {0u, 0u, 0u, 0u};
    if ((unsigned)iy < 128u && (unsigned)ix < 128u)
      v = *reinterpret_cast<const u32x4*>(a4 + (((size_t)n * 128 + iy) * 128 + ix) * LC_CI + ch * 8);
    *reinterpret_cast<u32x4*>(tile + p * LC_PITCH + ch * 16) = v;
  }
}

__global__ __launch_bounds__(256) void lo_final_conv_fwd_kernel(const f16* __restrict__ a4, const float* __restrict__ w,
                                                                const float* __restrict__ bias, const float* __restrict__ target,
                                                                float* __restrict__ recon, float* __restrict__ mse_partial) {
  __shared__ __attribute__((aligned(16))) unsigned char tile[LC_TP * LC_TP * LC_PITCH];
  __shared__ __attribute__((aligned(16))) float wl[9][LC_CI][4];
  __shared__ float red[4];
  const int tid = threadIdx.x, tx = blockIdx.x, ty = blockIdx.y, n = blockIdx.z;
  lc_stage_a4(a4, tile, n, ty, tx, tid);
  for (int i = tid; i < 9 * LC_CI; i += 256) {
    int ci = i % LC_CI, t = i / LC_CI;
    wl[t][ci][0] = w[(0 * LC_CI + ci) * 9 + t];
    wl[t][ci][1] = w[(1 * LC_CI + ci) * 9 + t];
    wl[t][ci][2] = w[(2 * LC_CI + ci) * 9 + t];
    wl[t][ci][3] = 0.f;
  }
  __syncthreads();
  const int px = tid & 15, py = tid >> 4;
  float a0 = bias[0], a1 = bias[1], a2 = bias[2];
#pragma unroll 1
  for (int r = 0; r < 3; ++r)
#pragma unroll 1
    for (int s = 0; s < 3; ++s) {
      const unsigned char* p = tile + ((py + r) * LC_TP + px + s) * LC_PITCH;
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) {
        f16x8 h = *reinterpret_cast<const f16x8*>(p + ch * 16);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          f32x4 wv = *reinterpret_cast<const f32x4*>(&wl[r * 3 + s][ch * 8 + j][0]);
          float xv = (float)h[j];
          a0 += xv * wv[0]; a1 += xv * wv[1]; a2 += xv * wv[2];
        }
      }
    }
  const int oy = ty * LC_T + py, ox = tx * LC_T + px;
  float r0 = tanhf(a0), r1 = tanhf(a1), r2 = tanhf(a2);
  size_t o = ((size_t)n * 3 * 128 + oy) * 128 + ox;
  recon[o] = r0; recon[o + 128 * 128] = r1; recon[o + 2 * 128 * 128] = r2;
  if (mse_partial) {
    float d0 = r0 - target[o], d1 = r1 - target[o + 128 * 128], d2 = r2 - target[o + 2 * 128 * 128];
    float e = lo_wave_sum(d0 * d0 + d1 * d1 + d2 * d2);
    if ((tid & 63) == 0) red[tid >> 6] = e;
    __syncthreads();
    if (tid == 0) mse_partial[((size_t)n * gridDim.y + ty) * gridDim.x + tx] = red[0] + red[1] + red[2] + red[3];
  }
}

// ---------------------------------------------------------------------------------------------
// final conv backward (data + weight + bias gradients in one pass over the tiles)
//   dpre[co] = g[co] * (1 - recon[co]^2), where g = *coef * (recon - target)   (fused loss path, coef on device)
//                                         or    g = gscale * drecon            (explicit upstream gradient)
//   da4[pix][ci]   = sum_{r,s,co} dpre[pix + (1-r, 1-s)][co] * W[co][ci][r][s]          -> fp16 NHWC
//   dW[co][ci][rs] = sum_pix dpre[pix][co] * a4[pix + (r-1, s-1)][ci]  -> partial[wg][co*288 + ci*9 + rs], + bias [864..866]
// ---------------------------------------------------------------------------------------------
struct LcBwdArgs {
  const f16* a4;
  const float* w;
  const float* recon;
  const float* target;   // fused path
  const float* drecon;   // explicit path (or null)
  const float* coef;     // device scalar for the fused path
  float gscale;          // multiplier for the explicit path (loss scale)
  f16* da4;
  float* partial;        // [nwg][867]
};

__global__ __launch_bounds__(256) void lo_final_conv_bwd_kernel(LcBwdArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char tile[LC_TP * LC_TP * LC_PITCH];  // reused as reduction buffer
  __shared__ __attribute__((aligned(16))) float dp[LC_TP * LC_TP][4];
  __shared__ __attribute__((aligned(16))) float w2[9][3][LC_CI];
  __shared__ float bred[8][3];
  const int tid = threadIdx.x, ty = blockIdx.x, n = blockIdx.y;   // one workgroup = one row of 8 tiles
  const float cf = a.drecon ? a.gscale : *a.coef;
  for (int i = tid; i < 9 * 3 * LC_CI; i += 256) {
    int ci = i % LC_CI, co = (i / LC_CI) % 3, t = i / (3 * LC_CI);
    w2[t][co][ci] = a.w[(co * LC_CI + ci) * 9 + t];
  }
  const int ci = tid & 31, q = tid >> 5;
  float wacc[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) wacc[k] = 0.f;
  float bacc0 = 0.f, bacc1 = 0.f, bacc2 = 0.f;
  for (int tx = 0; tx < 8; ++tx) {
    __syncthreads();                     // previous tile fully consumed
    lc_stage_a4(a.a4, tile, n, ty, tx, tid);
    for (int p = tid; p < LC_TP * LC_TP; p += 256) {
      int y = p / LC_TP, xx = p % LC_TP;
      int iy = ty * LC_T - 1 + y, ix = tx * LC_T - 1 + xx;
      float d0 = 0.f, d1 = 0.f, d2 = 0.f;
      if ((unsigned)iy < 128u && (unsigned)ix < 128u) {
        size_t o = ((size_t)n * 3 * 128 + iy) * 128 + ix;
        float r0 = a.recon[o], r1 = a.recon[o + 16384], r2 = a.recon[o + 32768];
        float g0, g1, g2;
        if (a.drecon) { g0 = a.drecon[o]; g1 = a.drecon[o + 16384]; g2 = a.drecon[o + 32768]; }
        else { g0 = r0 - a.target[o]; g1 = r1 - a.target[o + 16384]; g2 = r2 - a.target[o + 32768]; }
        d0 = cf * g0 * (1.f - r0 * r0);
        d1 = cf * g1 * (1.f - r1 * r1);
        d2 = cf * g2 * (1.f - r2 * r2);
      }
      dp[p][0] = d0; dp[p][1] = d1; dp[p][2] = d2; dp[p][3] = 0.f;
    }
    __syncthreads();
    // ---- data gradient: thread = pixel, 32 input channels
    {
      const int px = tid & 15, py = tid >> 4;
      float acc[LC_CI];
#pragma unroll
      for (int j = 0; j < LC_CI; ++j) acc[j] = 0.f;
#pragma unroll 1
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          // output pixel (py,px) of da4 receives dpre at (py + 1 - r, px + 1 - s); tile coords are +1
          f32x4 d = *reinterpret_cast<const f32x4*>(&dp[(py + 2 - r) * LC_TP + (px + 2 - s)][0]);
#pragma unroll
          for (int co = 0; co < 3; ++co) {
            float dv = d[co];
#pragma unroll
            for (int j = 0; j < LC_CI; j += 4) {
              f32x4 wv = *reinterpret_cast<const f32x4*>(&w2[r * 3 + s][co][j]);
              acc[j] += dv * wv[0]; acc[j + 1] += dv * wv[1]; acc[j + 2] += dv * wv[2]; acc[j + 3] += dv * wv[3];
            }
          }
        }
      const int oy = ty * LC_T + py, ox = tx * LC_T + px;
      f16* dst = a.da4 + (((size_t)n * 128 + oy) * 128 + ox) * LC_CI;
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) {
        f16x8 h;
#pragma unroll
        for (int j = 0; j < 8; ++j) h[j] = (f16)acc[ch * 8 + j];
        *reinterpret_cast<f16x8*>(dst + ch * 8) = h;
      }
    }
    // ---- weight gradient: thread = (ci, pixel subset q of 32 pixels); 27 accumulators (tap, co) kept across tiles
    for (int pp = q * 32; pp < q * 32 + 32; ++pp) {
      int py = pp >> 4, px = pp & 15;
      f32x4 d = *reinterpret_cast<const f32x4*>(&dp[(py + 1) * LC_TP + px + 1][0]);
      if (ci == 0) { bacc0 += d[0]; bacc1 += d[1]; bacc2 += d[2]; }
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          float xv = (float)*reinterpret_cast<const f16*>(tile + ((py + r) * LC_TP + px + s) * LC_PITCH + ci * 2);
          wacc[(r * 3 + s) * 3 + 0] += d[0] * xv;
          wacc[(r * 3 + s) * 3 + 1] += d[1] * xv;
          wacc[(r * 3 + s) * 3 + 2] += d[2] * xv;
        }
    }
  }
  __syncthreads();                       // everyone is done reading the a4 tile: reuse it for the reduction
  float* red = reinterpret_cast<float*>(tile);   // [4 subsets][27][32] floats = 13824 B
  static_assert(LC_TP * LC_TP * LC_PITCH >= 4 * 27 * 32 * 4, "reduction buffer");
  float* outp = a.partial + ((size_t)n * gridDim.x + ty) * 867;
  if (ci == 0) { bred[q][0] = bacc0; bred[q][1] = bacc1; bred[q][2] = bacc2; }
  // two rounds of 4 pixel subsets each (fixed order)
  for (int round = 0; round < 2; ++round) {
    if ((q >> 2) == round) {
#pragma unroll
      for (int k = 0; k < 27; ++k) red[((q & 3) * 27 + k) * 32 + ci] = wacc[k];
    }
    __syncthreads();
    for (int i = tid; i < 27 * 32; i += 256) {
      float t = red[i] + red[27 * 32 + i] + red[2 * 27 * 32 + i] + red[3 * 27 * 32 + i];
      int k = i / 32, c = i % 32;       // k = (r*3+s)*3 + co
      int rs = k / 3, co = k % 3;
      size_t oidx = (size_t)co * 288 + c * 9 + rs;
      if (round == 0) outp[oidx] = t; else outp[oidx] += t;   // same thread owns oidx in both rounds
    }
    __syncthreads();
  }
  if (tid < 3) {
    float t = 0.f;
    for (int k = 0; k < 8; ++k) t += bred[k][tid];
    outp[864 + tid] = t;
  }
}

// =============================================================================================
// launchers
// =============================================================================================
int lo_first_conv_fwd(const float* x, const float* w, const float* bias, f16* v, float* gn_partial, int B, hipStream_t st) {
  LoProfScope _p("lo_first_conv_fwd", 2.0 * B * 4096 * 64 * 27, (double)B * (3 * 16384 * 4 + 4096 * 64 * 2), st);
  hipLaunchKernelGGL(lo_first_conv_fwd_kernel, dim3(64, B), dim3(256), 0, st, x, w, bias, v, gn_partial);
  LO_LAUNCH_CHECK("first_conv_fwd");
  return LO_OK;
}
// partial: B*8*1728 floats
int lo_first_conv_wgrad(const float* x, const f16* dv, float* partial, float* dw, int B, float scale, hipStream_t st) {
  LoProfScope _p("lo_first_conv_wgrad(+sum)", 2.0 * B * 4096 * 64 * 27, (double)B * (3 * 16384 * 4 + 4096 * 64 * 2), st);
  hipLaunchKernelGGL(lo_first_conv_wgrad_kernel, dim3(8, B), dim3(256), 0, st, x, dv, partial);
  LO_LAUNCH_CHECK("first_conv_wgrad");
  hipLaunchKernelGGL(lo_colsum_kernel, dim3((1728 + 15) / 16), dim3(256), 0, st, partial, dw, B * 8, 1728, 1728, scale);
  LO_LAUNCH_CHECK("first_conv_wgrad_sum");
  return LO_OK;
}
int lo_colsum(const float* partial, float* out, int nrow, int ncol, int stride, float scale, hipStream_t st) {
  hipLaunchKernelGGL(lo_colsum_kernel, dim3((ncol + 15) / 16), dim3(256), 0, st, partial, out, nrow, ncol, stride, scale);
  LO_LAUNCH_CHECK("colsum");
  return LO_OK;
}
int lo_final_conv_fwd(const f16* a4, const float* w, const float* bias, const float* target, float* recon,
                      float* mse_partial, int B, hipStream_t st) {
  LoProfScope _p("lo_final_conv_fwd", 2.0 * B * 16384 * 3 * 288, (double)B * 16384 * (32 * 2 + 3 * 4 * 2), st);
  hipLaunchKernelGGL(lo_final_conv_fwd_kernel, dim3(8, 8, B), dim3(256), 0, st, a4, w, bias, target, recon, mse_partial);
  LO_LAUNCH_CHECK("final_conv_fwd");
  return LO_OK;
}
// partial: B*64*867 floats ; dw [3][32][3][3], db [3]
int lo_final_conv_bwd(const f16* a4, const float* w, const float* recon, const float* target, const float* drecon,
                      const float* coef, float gscale, f16* da4, float* partial, float* dw, float* db, int B, float scale,
                      hipStream_t st) {
  LcBwdArgs a{a4, w, recon, target, drecon, coef, gscale, da4, partial};
  LoProfScope _p("lo_final_conv_bwd(+sums)", 4.0 * B * 16384 * 3 * 288, (double)B * 16384 * (32 * 2 * 2 + 3 * 4 * 2), st);
  hipLaunchKernelGGL(lo_final_conv_bwd_kernel, dim3(8, B), dim3(256), 0, st, a);
  LO_LAUNCH_CHECK("final_conv_bwd");
  hipLaunchKernelGGL(lo_colsum_kernel, dim3((864 + 15) / 16), dim3(256), 0, st, partial, dw, B * 8, 864, 867, scale);
  LO_LAUNCH_CHECK("final_conv_dw");
  // bias: columns 864..866 of the same partial matrix
  hipLaunchKernelGGL(lo_colsum_kernel, dim3(1), dim3(256), 0, st, partial + 864, db, B * 8, 3, 867, scale);
  LO_LAUNCH_CHECK("final_conv_db");
  return LO_OK;
}
