// The two convolutions of the VAE that are NOT dense contractions as direct, HBM-bound
// kernels:
//   first conv  Conv2d(3->64, k3, s2, p1) on fp32 NCHW input (lunar_generate.py:95)  K = 27
//   final conv  Conv2d(32->3, k3, s1, p1) + tanh (+ MSE partial sums)   (lunar_generate.py:192,227-228;
//               train_hybrid.py:859)                                                  N = 3
// plus their weight / data gradients.  fp32 VALU math, LDS-staged input tiles, coalesced NHWC fp16 stores.
#include "lo_common.h"
#include <stdlib.h>

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

// The same forward on MFMA, ROWS output rows per workgroup.  D[co][px] = sum_k W[co][k] X[k][px], k = (ci, r, s) < 27 padded to 32:
//   A (weights) : fp32 split into fp16 hi + lo halves, the four 16-channel fragments of a lane built once per workgroup
//   B (im2col)  : lane (px, k group g) gathers its eight k values from the staged fp32 input rows (eight ds_read_b32 at per-lane
//                 constant offsets + 2 * px), split into hi + lo as well; three MFMAs per 16 x 16 block (hi*hi + lo*hi + hi*lo)
//                 keep fp32-input accuracy (the direct kernel above spends 27 broadcast LDS reads x 16 FMAs per thread: 36 us)
//   wave w owns pixels 16 w .. 16 w + 15 of every row; a lane ends with 4 consecutive channels of one pixel per fragment; the row
//   is staged in LDS for 16-byte stores; GroupNorm sums (groups of 8 channels = the lane pairs g, g ^ 1) by shuffles, then waves
//   in fixed order.  The 2 ROWS + 1 input rows of the workgroup are staged once.
template <int ROWS>
__global__ __launch_bounds__(256) void lo_first_conv_fwd_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                     const float* __restrict__ bias, f16* __restrict__ v,
                                                                     float* __restrict__ gn_partial) {
  constexpr int NR = 2 * ROWS + 1, RP = FC_W + 2;
  __shared__ float xs[3][NR][RP];
  __shared__ float ws[FC_CO * 27];
  __shared__ __attribute__((aligned(16))) unsigned char so[FC_OW * 144];   // one output row: 64 px x (128 B + 16 B pad)
  __shared__ float red[4][8][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 15, g = lane >> 4;
  const int oy0 = blockIdx.x * ROWS, n = blockIdx.y;
  for (int i = tid; i < 3 * NR * RP; i += 256) {
    const int col = i % RP, r = (i / RP) % NR, ci = i / (RP * NR);
    const int iy = 2 * oy0 - 1 + r, ix = col - 1;
    float val = 0.f;
    if ((unsigned)iy < 128u && (unsigned)ix < 128u) val = x[(((size_t)n * 3 + ci) * 128 + iy) * 128 + ix];
    xs[ci][r][col] = val;
  }
  for (int i = tid; i < FC_CO * 27; i += 256) ws[i] = w[i];
  __syncthreads();
  // A fragments: lane (row co = 16 mf + m, k = 8 g .. 8 g + 7)
  f16x8 wh[4], wl[4];
#pragma unroll
  for (int mf = 0; mf < 4; ++mf)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 8 * g + j;
      const float wv = k < 27 ? ws[(16 * mf + m) * 27 + k] : 0.f;
      const f16 hi = (f16)wv;
      wh[mf][j] = hi;
      wl[mf][j] = (f16)(wv - (float)hi);
    }
  // B gather offsets of this lane's eight k (floats relative to xs[0][0][2 * px]); k >= 27 reads a valid address and is zeroed
  int koff[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = 8 * g + j, kk = k < 27 ? k : 0;
    const int ci = kk / 9, r = (kk % 9) / 3, sx = kk % 3;
    koff[j] = (ci * NR + r) * RP + sx;
  }
  f32x4 bv[4];
#pragma unroll
  for (int mf = 0; mf < 4; ++mf) bv[mf] = *reinterpret_cast<const f32x4*>(bias + 16 * mf + 4 * g);
  const float* xbase = &xs[0][0][0];
  const int px = 16 * wave + m;
#pragma unroll 1
  for (int rr = 0; rr < ROWS; ++rr) {
    f16x8 xh, xl;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float xv = xbase[koff[j] + 2 * rr * RP + 2 * px];
      if (8 * g + j >= 27) xv = 0.f;
      const f16 hi = (f16)xv;
      xh[j] = hi;
      xl[j] = (f16)(xv - (float)hi);
    }
    float gs[4], gq[4];
#pragma unroll
    for (int mf = 0; mf < 4; ++mf) {
      f32x4 acc = bv[mf];
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[mf], xh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[mf], xh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[mf], xl, acc, 0, 0, 0);
      // D[co = 16 mf + 4 g + j][px]
      const f16x4 h = {(f16)acc[0], (f16)acc[1], (f16)acc[2], (f16)acc[3]};
      *reinterpret_cast<f16x4*>(so + px * 144 + (16 * mf + 4 * g) * 2) = h;
      float s_ = 0.f, q_ = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float a = (float)h[j]; s_ += a; q_ += a * a; }
      gs[mf] = s_; gq[mf] = q_;
    }
    // GroupNorm sums of this row: over the 16 pixel lanes, then the lane pair (g, g ^ 1) = one group of 8 channels
#pragma unroll
    for (int mf = 0; mf < 4; ++mf) {
#pragma unroll
      for (int o = 1; o < 32; o <<= 1) { gs[mf] += __shfl_xor(gs[mf], o, 64); gq[mf] += __shfl_xor(gq[mf], o, 64); }
    }
    if (m == 0 && (g & 1) == 0) {
#pragma unroll
      for (int mf = 0; mf < 4; ++mf) { red[wave][2 * mf + (g >> 1)][0] = gs[mf]; red[wave][2 * mf + (g >> 1)][1] = gq[mf]; }
    }
    __syncthreads();
    const int oy = oy0 + rr;
    {   // 64 px x 8 chunks of 16 B = 512 chunks: two per thread, 128 contiguous bytes per pixel
      f16* dst = v + (((size_t)n * 64 + oy) * 64) * FC_CO;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c = tid + 256 * i, p = c >> 3, ch = c & 7;
        *reinterpret_cast<u32x4*>(dst + p * FC_CO + ch * 8) = *reinterpret_cast<const u32x4*>(so + p * 144 + ch * 16);
      }
    }
    if (tid < 16) {
      const int grp = tid >> 1, which = tid & 1;
      gn_partial[(((size_t)n * 64 + oy) * 8 + grp) * 2 + which] = red[0][grp][which] + red[1][grp][which] + red[2][grp][which] + red[3][grp][which];
    }
    __syncthreads();                       // `so` and `red` are rewritten by the next row
  }
}

// The same weight gradient on MFMA: D[co][k] = sum_p dv[p][co] X[p][k] with k = (ci, r, s) < 27 padded to 32 and the reduction
// over the 64 pixels of an output row in two 32-pixel MFMA steps.  The direct kernel above issues one broadcast LDS read per FMA
// (27 per pixel and thread) and is LDS-bound: 72 us for 0.9 GFLOP and 46 MB at batch 64; here a row costs 8 MFMAs per wave.
//   A[co][p]  : dv tile [64 p][64 co] fp16 in LDS, fragments by ds_read_b64_tr_b16 (lane group q of a wave takes pixels 4q..4q+3 and
//               16+4q..16+4q+3 of the step: the same k permutation on both operands leaves the sum unchanged)
//   B[p][k]   : im2col of the three staged input rows, built in LDS TRANSPOSED ([k][p], so a lane's four consecutive pixels are one
//               8-byte read), split into fp16 hi + lo halves (x = hi + lo to 2^-22): two accumulator sets, added at the end, keep
//               the fp32-input precision the gradient test asks for (1e-4 relative L2)
// Partial layout: partial[wg][co*27 + k] over ROWS output rows per workgroup (4: 1024 workgroups at batch 64, four per CU -- the
// kernel is a chain of load -> barrier -> build -> barrier -> MFMA per row, so concurrency across workgroups hides its latency).
template <int ROWS>
__global__ __launch_bounds__(256) void lo_first_conv_wgrad_mfma_kernel(const float* __restrict__ x, const f16* __restrict__ dv,
                                                                       float* __restrict__ partial) {
  __shared__ float xs[3][3][FC_W + 2];
  __shared__ __attribute__((aligned(16))) f16 ds[64][FC_CO];          // [pixel][co], 128-byte rows
  __shared__ __attribute__((aligned(16))) f16 xt[2][32][64 + 8];      // [hi / lo][k][pixel] (+8: rows 144 bytes apart)
  const int tid = threadIdx.x, n = blockIdx.y, rb = blockIdx.x;
  const int lane = tid & 63, wave = tid >> 6, i16 = lane & 15, q = lane >> 4;
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // the global loads of row oy + 1 (this thread's share: two 16-byte pieces of the dv row, up to five input elements) are issued
  // into registers before row oy is built and multiplied, and land in LDS at the top of the next iteration: the row loop was a
  // chain of load -> barrier -> build -> barrier -> MFMA with the load latency exposed four times per workgroup
  constexpr int XN = (3 * 3 * (FC_W + 2) + 255) / 256;
  f16x8 pd[2];
  float px[XN];
  auto prefetch = [&](int oy) __attribute__((always_inline)) {
    const f16* src = dv + (((size_t)n * 64 + oy) * 64) * FC_CO;
#pragma unroll
    for (int j = 0; j < 2; ++j) pd[j] = reinterpret_cast<const f16x8*>(src)[tid + 256 * j];
#pragma unroll
    for (int j = 0; j < XN; ++j) {
      const int i = tid + 256 * j;
      const int col = i % (FC_W + 2), r = (i / (FC_W + 2)) % 3, ci = i / (3 * (FC_W + 2));
      const int iy = 2 * oy - 1 + r, ix = col - 1;
      float v = 0.f;
      if (i < 3 * 3 * (FC_W + 2) && (unsigned)iy < 128u && (unsigned)ix < 128u) v = x[(((size_t)n * 3 + ci) * 128 + iy) * 128 + ix];
      px[j] = v;
    }
  };
  prefetch(rb * ROWS);
  for (int oy = rb * ROWS; oy < rb * ROWS + ROWS; ++oy) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) reinterpret_cast<f16x8*>(&ds[0][0])[tid + 256 * j] = pd[j];
#pragma unroll
    for (int j = 0; j < XN; ++j) {
      const int i = tid + 256 * j;
      if (i < 3 * 3 * (FC_W + 2)) (&xs[0][0][0])[i] = px[j];
    }
    if (oy + 1 < rb * ROWS + ROWS) prefetch(oy + 1);
    __syncthreads();
    for (int i = tid; i < 32 * 64; i += 256) {
      const int k = i >> 6, p = i & 63;
      float v = 0.f;
      if (k < 27) {
        const int ci = k / 9, r = (k % 9) / 3, sx = k % 3;
        v = xs[ci][r][2 * p + sx];
      }
      const f16 hi = (f16)v;
      xt[0][k][p] = hi;
      xt[1][k][p] = (f16)(v - (float)hi);
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      // A: transposed reads of the dv tile; this lane supplies row (pixel) 4q + (i16 >> 2), columns 16*wave + 4*(i16 & 3) ..
      const unsigned char* dbase = reinterpret_cast<const unsigned char*>(&ds[0][0]) + (size_t)(32 * ks + 4 * q + (i16 >> 2)) * 128 + wave * 32 + (i16 & 3) * 8;
      h16x4 a_lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((h16x4 __attribute__((address_space(3)))*)(dbase));
      h16x4 a_hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((h16x4 __attribute__((address_space(3)))*)(dbase + 16 * 128));
      const f16x8 af = {(f16)a_lo[0], (f16)a_lo[1], (f16)a_lo[2], (f16)a_lo[3], (f16)a_hi[0], (f16)a_hi[1], (f16)a_hi[2], (f16)a_hi[3]};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f16* xr = &xt[t >> 1][16 * (t & 1) + i16][32 * ks + 4 * q];
        const f16x4 b0 = *reinterpret_cast<const f16x4*>(xr), b1 = *reinterpret_cast<const f16x4*>(xr + 16);
        const f16x8 bf = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf, acc[t], 0, 0, 0);
      }
    }
  }
  // D[co][k]: lane column k = 16*(t & 1) + i16, rows co = 16*wave + 4q + r
  float* out = partial + ((size_t)n * gridDim.x + rb) * (FC_CO * 27);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int k = 16 * t + i16;
    if (k < 27) {
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(16 * wave + 4 * q + r) * 27 + k] = acc[t][r] + acc[2 + t][r];
    }
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
  if (j < ncol) {
    // eight loads in flight per lane, added in row order (the one-load-at-a-time loop chained nrow / 16 round trips: 13 us for 1.7 MB)
    int k = r;
    for (; k + 7 * 16 < nrow; k += 8 * 16) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(k + 16 * u) * stride + j];
#pragma unroll
      for (int u = 0; u < 8; ++u) t += v[u];
    }
    for (; k < nrow; k += 16) t += partial[(size_t)k * stride + j];
  }
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

__device__ __forceinline__ void lc_stage_a4_wide(const f16* __restrict__ a4, unsigned char* tile, int n, int ty, int tx, int tid) {
  // tile[(y*18 + x)] = a4[n][ty*16 - 1 + y][tx*16 - 1 + x][0..31], zero outside.  All six loads of a thread are issued before the
  // first LDS store (a rolled loop serialises six global latencies).  The backward kernel has no registers for this form (83 -> 131 us
  // when it was tried there): it stages its patches by LDS-DMA instead
  constexpr int NCH = LC_TP * LC_TP * 4, NIT = (NCH + 255) / 256;
  u32x4 v[NIT];
#pragma unroll
  for (int k = 0; k < NIT; ++k) {
    const int i = tid + k * 256;
    const int ch = i & 3, p = i >> 2;
    const int y = p / LC_TP, xx = p % LC_TP;
    const int iy = ty * LC_T - 1 + y, ix = tx * LC_T - 1 + xx;
    v[k] = (u32x4){0u, 0u, 0u, 0u};
    if (i < NCH && (unsigned)iy < 128u && (unsigned)ix < 128u)
      v[k] = *reinterpret_cast<const u32x4*>(a4 + (((size_t)n * 128 + iy) * 128 + ix) * LC_CI + ch * 8);
  }
#pragma unroll
  for (int k = 0; k < NIT; ++k) {
    const int i = tid + k * 256;
    if (i < NCH) *reinterpret_cast<u32x4*>(tile + (i >> 2) * LC_PITCH + (i & 3) * 16) = v[k];
  }
}

// The same forward on MFMA.  The direct kernel above issues 324 ds_read_b128 per output pixel (36 patch reads + 288 broadcast
// weight reads) and is LDS-bound: 63 us for 0.9 GFLOP and 92 MB at batch 64.  Here a tile row of 16 pixels is one 16x16 output
// block D[px][col] and a tap is one v_mfma_f32_16x16x32_f16 over the 32 input channels:
//   A[px][ci]  : 16 bytes of the staged patch at pixel (py + r, px + s), chunk g       (one ds_read_b128 per tap and tile row)
//   B[ci][col] : the tap's weights split w = hi + lo (fp16 halves, exact to 2^-22): columns 0..2 hold hi of co 0..2, columns
//                3..5 hold lo, so one MFMA per tap keeps fp32-weight precision (recon parity 2e-5; an fp16-weight form broke it
//                in round 1); column co + 3 is added to column co by a lane shuffle after the nine taps
// The 12 lanes with col < 3 hold four consecutive pixels each: 16-byte recon stores, 16-byte target loads.  tanh is
// 1 - 2 / (exp(2x) + 1) on the hardware exp / rcp (|err| < 3e-7): libm's tanhf on 12 of 64 lanes was half of the kernel's time.
__device__ __forceinline__ float lc_tanh(float x) { return 1.f - 2.f * __frcp_rn(__expf(2.f * x) + 1.f); }

__global__ __launch_bounds__(256) void lo_final_conv_fwd_mfma_kernel(const f16* __restrict__ a4, const float* __restrict__ w,
                                                                     const float* __restrict__ bias, const float* __restrict__ target,
                                                                     float* __restrict__ recon, float* __restrict__ mse_partial) {
  __shared__ __attribute__((aligned(16))) unsigned char tile[LC_TP * LC_TP * LC_PITCH];
  __shared__ __attribute__((aligned(16))) f16 wf[9 * 4 * 6 * 8];   // [tap][g][hi co0..2 | lo co0..2][8 channels]
  __shared__ float red[4];
  const int tid = threadIdx.x, tx = blockIdx.x, ty = blockIdx.y, n = blockIdx.z;
  const int lane = tid & 63, wave = tid >> 6, m = lane & 15, g = lane >> 4;
  lc_stage_a4_wide(a4, tile, n, ty, tx, tid);
  for (int i = tid; i < 3 * LC_CI * 9; i += 256) {    // coalesced read of the 864 weights, split, fragment order
    const int t = i % 9, ci = (i / 9) % LC_CI, co = i / (9 * LC_CI);
    const float wv = w[i];
    const f16 hi = (f16)wv;
    const int slot = ((t * 4 + (ci >> 3)) * 6 + co) * 8 + (ci & 7);
    wf[slot] = hi;
    wf[slot + 24] = (f16)(wv - (float)hi);
  }
  const float bv = m < 3 ? bias[m] : 0.f;
  __syncthreads();
  f16x8 wb[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    wb[t] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
    if (m < 6) wb[t] = *reinterpret_cast<const f16x8*>(&wf[((t * 4 + g) * 6 + m) * 8]);
  }
  float sq = 0.f;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int py = wave * 4 + rr;                      // tile row of this 16-pixel block
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int r = t / 3, sx = t % 3;
      const f16x8 af = *reinterpret_cast<const f16x8*>(tile + ((py + r) * LC_TP + m + sx) * LC_PITCH + g * 16);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, wb[t], acc, 0, 0, 0);
    }
    // D[px][col]: column = m, rows px = 4g + j; lo part sits three lanes up
    f32x4 lo;
#pragma unroll
    for (int j = 0; j < 4; ++j) lo[j] = __shfl_down(acc[j], 3, 64);
    if (m < 3) {
      const int oy = ty * LC_T + py, ox = tx * LC_T + 4 * g;
      const size_t o = (((size_t)n * 3 + m) * 128 + oy) * 128 + ox;
      f32x4 rv;
#pragma unroll
      for (int j = 0; j < 4; ++j) rv[j] = lc_tanh(acc[j] + lo[j] + bv);
      *reinterpret_cast<f32x4*>(recon + o) = rv;
      if (mse_partial) {
        const f32x4 tv = *reinterpret_cast<const f32x4*>(target + o);
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float d = rv[j] - tv[j]; sq += d * d; }
      }
    }
  }
  if (mse_partial) {
    const float e = lo_wave_sum(sq);
    if (lane == 0) red[wave] = e;
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

// MFMA form (the direct form was LDS-bound on broadcast weight reads: 216 ds_read_b128 per pixel):
//   dpre (fp16, [18x18][4]) and the staged a4 tile live in LDS;
//   data gradient   D[px][ci] = sum_k A[px][k] Wd[k][ci],  k = tap*3 + co (27 of 32): per tile row two
//                   v_mfma_f32_16x16x32_f16 (ci 0..15, 16..31); A gathered from dpre with per-lane byte offsets
//   weight gradient D[co][ci] += sum_px dpre[px][co] a4[px + tap][ci] per tap: K = 32 pixels (two tile rows), A = dpre^T
//                   (3 of 16 rows used), B by transposed LDS reads of the a4 tile; 18 accumulators kept across the 8 tiles
// ---------------------------------------------------------------------------------------------
// Global latencies are kept out of the per-tile chain (round 4).  The first form paid, per tile and one after the other, six
// global-load round trips of a rolled a4 staging loop, two more for recon / target, the MFMA phases and the store: ~8 us per tile,
// 63 us per launch (rocprofv3) for 160 MB of traffic; this form takes 45 us.  Here the NEXT tile's a4 patch is already on its way while this tile
// computes: LDS-DMA (global_load_lds_dwordx4, no registers) into the other of two patch buffers, and the next tile's recon / target
// (or upstream gradient) values into 12 registers.  The DMA writes pixels at 64-byte pitch (no pad possible: 64 lanes x 16 B land
// lane-linear).  Out-of-image pixels (and the tail of the last 1-KiB instruction) are fetched from a zero page.
// ---------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(256))) unsigned int lo_zero_page_e[64];
#define LC_PATCH_BYTES (21 * 1024)     // 324 pixels x 64 B = 20.25 KiB, in whole 1-KiB DMA instructions
__device__ __forceinline__ void lc_dma_a4(const f16* __restrict__ a4, unsigned int lds_base, int n, int ty, int tx, int wave, int lane) {
  const unsigned char* zpage = reinterpret_cast<const unsigned char*>(lo_zero_page_e);
#pragma unroll
  for (int i = 0; i < 6; ++i) {                             // 21 instructions per patch: 6 / 5 / 5 / 5 per wave
    const int k = wave + 4 * i;                             // wave-uniform (readfirstlane by the caller)
    if (k >= 21) break;
    const int q = k * 64 + lane, p = q >> 2, slot = q & 3;
    const int y = p / LC_TP, xx = p - y * LC_TP;
    const int iy = ty * LC_T - 1 + y, ix = tx * LC_T - 1 + xx;
    const void* src = zpage + lane * 4;
    if (p < LC_TP * LC_TP && (unsigned)iy < 128u && (unsigned)ix < 128u) src = a4 + (((size_t)n * 128 + iy) * 128 + ix) * LC_CI + slot * 8;
    lo_dma16(src, __builtin_amdgcn_readfirstlane(lds_base + k * 1024));
  }
}
__device__ __forceinline__ void lo_dma4(const void* gptr, unsigned int lds_off) {     // 64 lanes x 4 B -> 256 B at lds_off, lane-linear
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(gptr), "s"(lds_off) : "memory");
}
// recon and target (or the upstream gradient) of the 18 x 18 patch as fp32 [2 tensors][3 channels][384] in LDS.  Wave w fetches the
// pixel blocks 64 w .. 64 w + 63 and 256 + 64 w .. -- exactly the pixels its own threads convert (p = tid, tid + 256), so the
// readers need this wave's vmcnt(0) only, no barrier.
__device__ __forceinline__ void lc_dma_rt(const float* __restrict__ recon, const float* __restrict__ gsrc, unsigned int lds_base, int n,
                                          int ty, int tx, int wave, int lane) {
  const unsigned char* zpage = reinterpret_cast<const unsigned char*>(lo_zero_page_e);
#pragma unroll
  for (int jb = 0; jb < 2; ++jb) {
    const int j = wave + 4 * jb;
    if (j * 64 >= LC_TP * LC_TP) break;
    const int p = j * 64 + lane;
    const int y = p / LC_TP, xx = p - y * LC_TP;
    const int iy = ty * LC_T - 1 + y, ix = tx * LC_T - 1 + xx;
    const bool in = p < LC_TP * LC_TP && (unsigned)iy < 128u && (unsigned)ix < 128u;
    const size_t o = ((size_t)n * 3 * 128 + iy) * 128 + ix;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      lo_dma4(in ? (const void*)(recon + o + c * 16384) : (const void*)(zpage + lane * 4), __builtin_amdgcn_readfirstlane(lds_base + ((0 * 3 + c) * 384 + j * 64) * 4));
      lo_dma4(in ? (const void*)(gsrc + o + c * 16384) : (const void*)(zpage + lane * 4), __builtin_amdgcn_readfirstlane(lds_base + ((1 * 3 + c) * 384 + j * 64) * 4));
    }
  }
}
template <int TPW>
__global__ __launch_bounds__(256, 2) void lo_final_conv_bwd_kernel(LcBwdArgs a) {
  constexpr int SP = 80;                                          // staging pitch of the output tile (bytes per pixel)
  __shared__ __attribute__((aligned(1024))) unsigned char tile[2][LC_PATCH_BYTES];   // two a4 patches (64-byte pixels); [0] reused as reduction buffer
  __shared__ __attribute__((aligned(16))) f16 dp[LC_TP * LC_TP][8];   // dpre as fp16 hi [0..2] + lo [4..6] (hi + lo ~ 22 bits)
  __shared__ __attribute__((aligned(16))) unsigned char so[256 * SP];
  __shared__ __attribute__((aligned(256))) float rt[2][3][384];    // recon, target / upstream gradient of the patch (fp32, by DMA)
  __shared__ float bred[4][3];
  constexpr int WPR = 8 / TPW;
  const int tid = threadIdx.x, lane = tid & 63, ty = blockIdx.x / WPR, tx0 = (blockIdx.x % WPR) * TPW, n = blockIdx.y;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), m = lane & 15, g = lane >> 4;
  const float cf = a.drecon ? a.gscale : *a.coef;
  const unsigned int tile_lds = (unsigned int)(size_t)(&tile[0][0]);      // LDS byte address (the low 32 bits of a shared pointer)
  const unsigned int rt_lds = (unsigned int)(size_t)(&rt[0][0][0]);
  const float* gsrc = a.drecon ? a.drecon : a.target;
  int aoff[8];
  f16x8 wd[2];
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) {
    const int kidx = 8 * g + jj;
    const int tap = kidx / 3, co = kidx - tap * 3, r = tap / 3, sx = tap - r * 3;
    aoff[jj] = kidx < 27 ? (((2 - r) * LC_TP + (2 - sx)) * 8 + co) * 2 : -1;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      // unconditional load from a clamped index, then a select: a branch around each load made the 16 loads 16 dependent round trips
      const float wv = a.w[kidx < 27 ? (co * LC_CI + t * 16 + m) * 9 + tap : 0];
      wd[t][jj] = kidx < 27 ? (f16)wv : (f16)0.f;
    }
  }
  f32x4 wacc[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t) { wacc[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; wacc[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  float bacc0 = 0.f, bacc1 = 0.f, bacc2 = 0.f;
  const int tq = m >> 2, tsub = (m & 3) * 8;
  lc_dma_a4(a.a4, tile_lds, n, ty, tx0, wave, lane);
  lc_dma_rt(a.recon, gsrc, rt_lds, n, ty, tx0, wave, lane);
  for (int it = 0; it < TPW; ++it) {
    const int tx = tx0 + it, cur = it & 1;
    // ---- dpre of this tile from the values this wave's DMA has staged (zero outside the image: the zero page)
    // this wave's DMA of THIS tile (patch share + recon / target) has landed.  vmcnt counts loads, stores and LDS-DMA together in
    // issue order: the four da4 stores of the previous tile are younger than that DMA and may stay in flight
    if (it == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int p = tid + h * 256;
      if (p < LC_TP * LC_TP) {
        const int y = p / LC_TP, xx = p - y * LC_TP;
        float d[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float r0 = rt[0][c][p];
          const float g0 = a.drecon ? rt[1][c][p] : r0 - rt[1][c][p];
          d[c] = cf * g0 * (1.f - r0 * r0);
        }
        if (y >= 1 && y <= LC_T && xx >= 1 && xx <= LC_T) { bacc0 += d[0]; bacc1 += d[1]; bacc2 += d[2]; }
        const f16 h0 = (f16)d[0], h1 = (f16)d[1], h2 = (f16)d[2];
        dp[p][0] = h0; dp[p][1] = h1; dp[p][2] = h2; dp[p][3] = (f16)0.f;
        dp[p][4] = (f16)(d[0] - (float)h0); dp[p][5] = (f16)(d[1] - (float)h1); dp[p][6] = (f16)(d[2] - (float)h2); dp[p][7] = (f16)0.f;
      }
    }
    // raw barriers inside the loop: __syncthreads() would drain the LDS-DMA that is in flight across it (it is a pending LDS write
    // on the VM counter); the LDS traffic of the ordinary ds instructions is ordered by the explicit lgkmcnt(0)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                            // everybody's patch share has landed; dp is complete; the other patch buffer and rt are free
    if (it + 1 < TPW) {
      lc_dma_a4(a.a4, tile_lds + (cur ^ 1) * LC_PATCH_BYTES, n, ty, tx + 1, wave, lane);
      lc_dma_rt(a.recon, gsrc, rt_lds, n, ty, tx + 1, wave, lane);
    }
    const unsigned char* dpb = reinterpret_cast<const unsigned char*>(&dp[0][0]);
    const unsigned char* tb = &tile[cur][0];
    // ---- data gradient: wave w owns tile rows 4w .. 4w+3
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int py = wave * 4 + rr;
      const int base = (py * LC_TP + m) * 16;
      f16x8 af;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) af[jj] = aoff[jj] >= 0 ? *reinterpret_cast<const f16*>(dpb + base + aoff[jj]) : (f16)0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, wd[t], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) *reinterpret_cast<f16*>(so + (py * 16 + 4 * g + e) * SP + (t * 16 + m) * 2) = (f16)d[e];
      }
    }
    // ---- weight gradient: wave w owns the 32-pixel chunks 2w, 2w+1 (tile rows 2c, 2c+1)
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      const int c = wave * 2 + cc;
      f16x8 af, al;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int prow = 2 * c + (e >> 2) + 1, pcol = 4 * g + (e & 3) + 1;
        af[e] = m < 3 ? dp[prow * LC_TP + pcol][m] : (f16)0.f;
        al[e] = m < 3 ? dp[prow * LC_TP + pcol][4 + m] : (f16)0.f;
      }
#pragma unroll
      for (int t9 = 0; t9 < 9; ++t9) {
        const int r = t9 / 3, sx = t9 - r * 3;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          // (64-byte pixels: the two groups of four pixels a 32-lane half touches share banks, a 2-way conflict on a read path that
          // is 5x under the MFMA time; a swizzle would make every tap's address a register of its own -- 36 of them -- in a
          // kernel that has none to spare)
          const unsigned char* p0 = tb + ((2 * c + r) * LC_TP + 4 * g + tq + sx) * 64 + t * 32 + tsub;
          h16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((h16x4 __attribute__((address_space(3)))*)p0);
          h16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((h16x4 __attribute__((address_space(3)))*)(p0 + LC_TP * 64));
          f16x8 bf = (f16x8){(f16)lo[0], (f16)lo[1], (f16)lo[2], (f16)lo[3], (f16)hi[0], (f16)hi[1], (f16)hi[2], (f16)hi[3]};
          wacc[t9][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf, wacc[t9][t], 0, 0, 0);
          wacc[t9][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bf, wacc[t9][t], 0, 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // ---- staged data gradient -> global (16-byte chunks; a tile row is 1 KiB contiguous in NHWC): exactly four stores per thread
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = tid + k * 256;
      const int px = i >> 2, ch = i & 3;
      const int oy = ty * LC_T + (px >> 4), ox = tx * LC_T + (px & 15);
      *reinterpret_cast<u32x4*>(a.da4 + (((size_t)n * 128 + oy) * 128 + ox) * LC_CI + ch * 8) =
          *reinterpret_cast<const u32x4*>(so + px * SP + ch * 16);
    }
  }
  __syncthreads();                       // everyone is done with the patches: reuse the first for the reduction
  float* red = reinterpret_cast<float*>(&tile[0][0]);   // [4 waves][27][32] floats = 13824 B
  static_assert(LC_PATCH_BYTES >= 4 * 27 * 32 * 4, "reduction buffer");
  if (g == 0) {
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int co = 0; co < 3; ++co) red[(wave * 27 + t9 * 3 + co) * 32 + t * 16 + m] = wacc[t9][t][co];
  }
  bacc0 = lo_wave_sum(bacc0); bacc1 = lo_wave_sum(bacc1); bacc2 = lo_wave_sum(bacc2);
  if (lane == 0) { bred[wave][0] = bacc0; bred[wave][1] = bacc1; bred[wave][2] = bacc2; }
  __syncthreads();
  float* outp = a.partial + ((size_t)n * gridDim.x + blockIdx.x) * 867;
  for (int i = tid; i < 27 * 32; i += 256) {
    const float t = red[i] + red[27 * 32 + i] + red[2 * 27 * 32 + i] + red[3 * 27 * 32 + i];
    const int k = i / 32, ci = i % 32;       // k = tap*3 + co
    const int rs = k / 3, co = k % 3;
    outp[(size_t)co * 288 + ci * 9 + rs] = t;
  }
  if (tid < 3) outp[864 + tid] = bred[0][tid] + bred[1][tid] + bred[2][tid] + bred[3][tid];
}

// =============================================================================================
// launchers
// =============================================================================================
int lo_first_conv_fwd(const float* x, const float* w, const float* bias, f16* v, float* gn_partial, int B, hipStream_t st) {
  LoProfScope _p("lo_first_conv_fwd", 2.0 * B * 4096 * 64 * 27, (double)B * (3 * 16384 * 4 + 4096 * 64 * 2), st);
  hipLaunchKernelGGL((lo_first_conv_fwd_mfma_kernel<4>), dim3(16, B), dim3(256), 0, st, x, w, bias, v, gn_partial);
  LO_LAUNCH_CHECK("first_conv_fwd");
  return LO_OK;
}
// partial: B*16*1728 floats
int lo_first_conv_wgrad(const float* x, const f16* dv, float* partial, float* dw, int B, float scale, hipStream_t st) {
  LoProfScope _p("lo_first_conv_wgrad(+sum)", 2.0 * B * 4096 * 64 * 27, (double)B * (3 * 16384 * 4 + 4096 * 64 * 2), st);
  hipLaunchKernelGGL((lo_first_conv_wgrad_mfma_kernel<4>), dim3(16, B), dim3(256), 0, st, x, dv, partial);
  LO_LAUNCH_CHECK("first_conv_wgrad");
  hipLaunchKernelGGL(lo_colsum_kernel, dim3((1728 + 15) / 16), dim3(256), 0, st, partial, dw, B * 16, 1728, 1728, scale);
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
  hipLaunchKernelGGL(lo_final_conv_fwd_mfma_kernel, dim3(8, 8, B), dim3(256), 0, st, a4, w, bias, target, recon, mse_partial);
  LO_LAUNCH_CHECK("final_conv_fwd");
  return LO_OK;
}
// partial: B*64*867 floats ; dw [3][32][3][3], db [3]
int lo_final_conv_bwd(const f16* a4, const float* w, const float* recon, const float* target, const float* drecon,
                      const float* coef, float gscale, f16* da4, float* partial, float* dw, float* db, int B, float scale,
                      hipStream_t st, hipStream_t sum_st, hipEvent_t sum_after) {
  LcBwdArgs a{a4, w, recon, target, drecon, coef, gscale, da4, partial};
  LoProfScope _p("lo_final_conv_bwd(+sums)", 4.0 * B * 16384 * 3 * 288, (double)B * 16384 * (32 * 2 * 2 + 3 * 4 * 2), st);
  constexpr int rows = 8;    // 8 tiles (a whole row of tiles) per workgroup; 4 / 2 measured in round 2: 86 / 108 / 147 us
  // the two column sums only feed the gradient buffer: with sum_st they run beside the chain, behind the event the caller armed
  // for this launch (LO_LAUNCH_STOP; sum_after is that event)
  LO_LAUNCH_STOP(lo_final_conv_bwd_kernel<8>, dim3(rows, B), dim3(256), 0, st, a);
  LO_LAUNCH_CHECK("final_conv_bwd");
  if (sum_st && sum_after) { LO_HIP(hipStreamWaitEvent(sum_st, sum_after, 0)); st = sum_st; }
  hipLaunchKernelGGL(lo_colsum_kernel, dim3((864 + 15) / 16), dim3(256), 0, st, partial, dw, B * rows, 864, 867, scale);
  LO_LAUNCH_CHECK("final_conv_dw");
  // bias: columns 864..866 of the same partial matrix
  hipLaunchKernelGGL(lo_colsum_kernel, dim3(1), dim3(256), 0, st, partial + 864, db, B * rows, 3, 867, scale);
  LO_LAUNCH_CHECK("final_conv_db");
  return LO_OK;
}
