// Latent / loss / optimizer kernels of the VAE training step (all HBM-bound, fp32 math, deterministic reductions):
//   encoder-head split-K reduction fused with the reparameterisation and the KL partial sums
//        (lunar_generate.py:124-125,150-152,259-261; train_hybrid.py:862)
//   loss finalisation + gradient seeds (train_hybrid.py:859,862,886-889,895)
//   KL / reparameterisation backward
//   global gradient norm + clip coefficient (train_hybrid.py:913) and fused clip + AdamW (train_hybrid.py:504-509,921)
//   weight cast / transpose-cast helpers for the Linear layers
#include "lo_common.h"
#include <stdlib.h>

// ---------------------------------------------------------------------------------------------
// counter-based N(0,1) generator for throughput runs (parity runs pass eps explicitly)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t lo_mix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ __forceinline__ float lo_randn(uint64_t seed, uint64_t idx) {
  uint64_t h = lo_mix64(seed ^ lo_mix64(idx));
  float u1 = ((float)(uint32_t)(h >> 40) + 0.5f) * (1.0f / 16777216.0f);          // (0,1)
  float u2 = ((float)(uint32_t)((h >> 8) & 0xFFFFFF)) * (1.0f / 16777216.0f);    // [0,1)
  return sqrtf(-2.0f * __logf(u1)) * __cosf(6.28318530718f * u2);
}

// ---------------------------------------------------------------------------------------------
// head reduce: slab [nsplit][B][2L] -> mu, logvar (fp32 [B][L]), z = mu + eps*exp(0.5 logvar) (fp16 [B][L]),
// eps_out (fp32, the noise actually used), kl_partial[block] = sum (1 + lv - mu^2 - e^lv)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lo_head_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ bias,
                                                             const float* __restrict__ eps_in, uint64_t seed,
                                                             float* __restrict__ mu, float* __restrict__ logvar,
                                                             f16* __restrict__ z, float* __restrict__ eps_out,
                                                             float* __restrict__ kl_partial, int B, int L, int nsplit,
                                                             float* __restrict__ mu_user, float* __restrict__ logvar_user) {
  __shared__ float red[4];
  int i = blockIdx.x * 256 + threadIdx.x;
  float term = 0.f;
  if (i < B * L) {
    int b = i / L, l = i - b * L;
    float m = bias[l], lv = bias[L + l];
    int s = 0;
    for (; s + 8 <= nsplit; s += 8) {            // sixteen loads in flight, added in split order
      float pm[8], pl[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float* p = slab + ((size_t)(s + u) * B + b) * (2 * L);
        pm[u] = p[l];
        pl[u] = p[L + l];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { m += pm[u]; lv += pl[u]; }
    }
    for (; s < nsplit; ++s) {
      const float* p = slab + ((size_t)s * B + b) * (2 * L);
      m += p[l];
      lv += p[L + l];
    }
    float e = eps_in ? eps_in[i] : lo_randn(seed, (uint64_t)i);
    float sd = expf(0.5f * lv);
    mu[i] = m;
    logvar[i] = lv;
    if (mu_user) { mu_user[i] = m; logvar_user[i] = lv; }     // the caller's copies (no device-to-device copy afterwards)
    eps_out[i] = e;
    z[i] = (f16)(m + e * sd);
    term = 1.f + lv - m * m - expf(lv);
  }
  float w = lo_wave_sum(term);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) kl_partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// ---------------------------------------------------------------------------------------------
// loss finalize (one workgroup).  rw/kw: recon/kl weights, adv: mean advantage, ls: loss scale
// out: losses[0..3] = recon_loss, kl_loss, vae_loss, pg_loss ;  coefs[0] = d(scaled loss)/d recon multiplier on
// (recon - x) ; coefs[1] = kl gradient coefficient (loss_scale * kl_weight / accum / (B*L))
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lo_loss_finalize_kernel(const float* __restrict__ mse_partial, int n_mse,
                                                               const float* __restrict__ kl_partial, int n_kl,
                                                               float rw, float kw, float adv_host, float accum, float ls,
                                                               const float* __restrict__ adv_dev, float* __restrict__ losses,
                                                               float* __restrict__ coefs, float n_rec, float n_lat) {
  // (on the dependent chain of every step: the partial sums are fetched eight at a time and the 256 per-thread sums go through a
  // shuffle tree and four LDS words instead of one thread adding 512 LDS values -- 8.3 -> ~3 us; a fixed order either way)
  __shared__ double red[2][4];
  double a = 0.0, b = 0.0;
  for (int i0 = threadIdx.x; i0 < n_mse; i0 += 8 * 256) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = i0 + u * 256 < n_mse ? mse_partial[i0 + u * 256] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) a += (double)v[u];
  }
  for (int i = threadIdx.x; i < n_kl; i += 256) b += (double)kl_partial[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double sa = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]), sb = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    float recon_loss = (float)(sa / (double)n_rec);
    float kl_loss = (float)(-0.5 * sb / (double)n_lat);
    float adv = adv_dev ? adv_dev[0] : adv_host;
    float pg = -adv * recon_loss;
    losses[0] = recon_loss;
    losses[1] = kl_loss;
    losses[2] = (rw * recon_loss + kw * kl_loss + pg) / accum;
    losses[3] = pg;
    coefs[0] = ls * (rw - adv) / accum * 2.0f / n_rec;
    coefs[1] = ls * kw / accum / n_lat;
  }
}

// ---------------------------------------------------------------------------------------------
// latent backward.  dz (fp16, loss-scaled) arrives from decoder.fc's data gradient.
//   fused path   : dmu = ckl*mu + dz ;  dlv = ckl*0.5*(e^lv - 1) + dz*eps*0.5*e^{lv/2}
//   explicit path: additional upstream gmu / glv (fp32, unscaled) times gscale are added, ckl = 0
// output dml fp16 [B][2L] = [dmu | dlv]
// ---------------------------------------------------------------------------------------------
__global__ void lo_latent_bwd_kernel(const f16* __restrict__ dz, const float* __restrict__ mu, const float* __restrict__ logvar,
                                     const float* __restrict__ eps, const float* __restrict__ coefs,
                                     const float* __restrict__ gmu, const float* __restrict__ glv, float gscale,
                                     f16* __restrict__ dml, int B, int L) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * L) return;
  int b = i / L, l = i - b * L;
  float ckl = coefs ? coefs[1] : 0.f;
  float d = (float)dz[i];
  float lv = logvar[i];
  float dm = ckl * mu[i] + d;
  float dl = ckl * 0.5f * (expf(lv) - 1.f) + d * eps[i] * 0.5f * expf(0.5f * lv);
  if (gmu) dm += gscale * gmu[i];
  if (glv) dl += gscale * glv[i];
  dml[(size_t)b * 2 * L + l] = (f16)dm;
  dml[(size_t)b * 2 * L + L + l] = (f16)dl;
}

// column sums of an fp16 matrix [M][N] -> fp32 out[N]*scale  (Linear bias gradients; M = batch, tiny)
// out[j] = scale * sum_m x[m][j] (bias gradients of the Linear layers: M = batch rows).  256 threads = 32 column chunks of 8
// (16-byte loads) x 8 row groups; the row groups are added in a fixed order through LDS (bitwise reproducible).  The one-thread-
// per-column loop this replaces chained M dependent 2-byte loads: 18 us for a 4 MB matrix.
__global__ __launch_bounds__(256) void lo_colsum_f16_kernel(const f16* __restrict__ x, float* __restrict__ out, int M, int N, float scale) {
  __shared__ float red[8][32][8];
  const int cc = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int j0 = (blockIdx.x * 32 + cc) * 8;
  float t[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) t[q] = 0.f;
  if (j0 < N) {
    for (int m = rg; m < M; m += 32) {            // four rows in flight per thread
      f16x8 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (m + 8 * u < M) v[u] = *reinterpret_cast<const f16x8*>(x + (size_t)(m + 8 * u) * N + j0);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (m + 8 * u < M) {
#pragma unroll
          for (int q = 0; q < 8; ++q) t[q] += (float)v[u][q];
        }
    }
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) red[rg][cc][q] = t[q];
  __syncthreads();
  const int c = threadIdx.x;                        // one output column per thread
  const int j = blockIdx.x * 256 + c;
  if (j < N) {
    float tot = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) tot += red[r][c >> 3][c & 7];
    out[j] = tot * scale;
  }
}

// ---------------------------------------------------------------------------------------------
// sprite decode: uint8 HWC [B,128,128,3] (sprites_*.npy rows) -> float32 CHW, x/127.5 - 1  (train_hybrid.py:181-182)
// ---------------------------------------------------------------------------------------------
__global__ void lo_decode_sprites_kernel(const uint8_t* __restrict__ u8, float* __restrict__ out, int npix_total) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;   // one pixel (3 bytes) per thread
  if (i >= npix_total) return;
  int n = i >> 14, p = i & 16383;
  const uint8_t* s = u8 + (size_t)i * 3;
  float* d = out + (size_t)n * 3 * 16384 + p;
  d[0] = (float)s[0] / 127.5f - 1.0f;
  d[16384] = (float)s[1] / 127.5f - 1.0f;
  d[32768] = (float)s[2] / 127.5f - 1.0f;
}

// ---------------------------------------------------------------------------------------------
// casts
// ---------------------------------------------------------------------------------------------
__global__ void lo_cast_f32_f16_kernel(const float* __restrict__ src, f16* __restrict__ dst, size_t n4) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n4; i += stride) {
    f32x4 v = reinterpret_cast<const f32x4*>(src)[i];
    f16x4 h = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
    reinterpret_cast<f16x4*>(dst)[i] = h;
  }
}
// dst[c][r] = (f16) src[r][c]   (src fp32 [R][C]); 64x64 tiles through LDS
__global__ __launch_bounds__(256) void lo_transpose_cast_kernel(const float* __restrict__ src, f16* __restrict__ dst, int R, int C) {
  __shared__ float t[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    int r = r0 + i, c = c0 + tx;
    t[i][tx] = (r < R && c < C) ? src[(size_t)r * C + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    int c = c0 + i, r = r0 + tx;
    if (r < R && c < C) dst[(size_t)c * R + r] = (f16)t[tx][i];
  }
}

// ---------------------------------------------------------------------------------------------
// gradient norm + clip coefficient + AdamW over ONE flat fp32 buffer (all 72 tensors are views of it)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lo_sumsq_partial_kernel(const float* __restrict__ g, size_t n, float* __restrict__ partial) {
  __shared__ float red[4];
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  size_t stride = (size_t)gridDim.x * 256;
  float acc = 0.f;
  size_t n4 = n / 4;
  for (size_t k = i; k < n4; k += stride) {
    f32x4 v = reinterpret_cast<const f32x4*>(g)[k];
    acc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (size_t k = n4 * 4; k < n; ++k) acc += g[k] * g[k];
  float w = lo_wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// norm_out: [0] = total L2 norm, [1] = clip coefficient, [2] = 1 if finite else 0, [3] += 1 if not finite (sticky count of
// skipped updates: the host halves its loss scale when it sees the count move, like torch.cuda.amp.GradScaler)
// fail (may be null): the engine's "a fused-GroupNorm rendezvous ran out" word (lo_common.h, lo_arrive_and_wait).  Non-zero = some
// launch of this step produced garbage: the update is skipped exactly like one with a non-finite norm (the word is sticky: every
// later update is skipped too, until the host has seen it -- VAEStepper.metrics() / save_checkpoint raise).  It does not count as
// a loss-scale overflow (norm_out[3] is left alone).
__global__ __launch_bounds__(256) void lo_gradnorm_finalize_kernel(const float* __restrict__ partial, int n, float max_norm,
                                                                   float* __restrict__ norm_out, const unsigned int* __restrict__ fail) {
  __shared__ double red[256];
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) a += (double)partial[i];
  red[threadIdx.x] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < 256; ++i) s += red[i];
    float norm = (float)sqrt(s);
    float coef = max_norm / (norm + 1e-6f);
    if (coef > 1.0f) coef = 1.0f;
    bool finite = isfinite(norm);
    const bool lost = fail && *fail != 0u;
    norm_out[0] = norm;
    norm_out[1] = (finite && !lost) ? coef : 0.f;
    norm_out[2] = (finite && !lost) ? 1.f : 0.f;
    if (!finite) norm_out[3] += 1.f;
  }
}
// torch.optim.AdamW (decoupled weight decay, eps outside the bias-corrected sqrt) with the clip coefficient folded in.
// A non-finite gradient norm skips the update (GradScaler semantics).
// One element: every multiply-add is written out (no compiler contraction inside), so that each of the unrolled copies below and
// the scalar tail round identically -- the pipelined optimizer step updates the flat buffer in several launches over sub-ranges
// and must give the bits of the one-launch form (tests/test_vae_gpu.py::test_pipelined_optimizer_step_equals_the_plain_one).
__device__ __forceinline__ void lo_adamw_elem(float& p, float g, float& m, float& v, float coef, float decay, float omb1, float beta2,
                                              float omb2, float bc2_sqrt, float eps, float step_size) {
#pragma clang fp contract(off)
  const float gg = g * coef;
  const float pp = p * decay;                              // decay = 1 - lr * wd
  const float mm = __builtin_fmaf(gg - m, omb1, m);        // m + (g - m)(1 - beta1)
  const float v2 = __builtin_fmaf(omb2 * gg, gg, v * beta2);
  const float denom = sqrtf(v2) / bc2_sqrt + eps;
  p = __builtin_fmaf(-step_size, mm / denom, pp);
  m = mm;
  v = v2;
}
__global__ __launch_bounds__(256) void lo_adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, size_t n, const float* __restrict__ norm,
                                                       float lr, float beta1, float beta2, float eps, float wd, float bc1,
                                                       float bc2_sqrt, f16* __restrict__ cast) {
  // cast != null: the fp16 operand copy of the updated parameters (same layout) is written in the same pass -- the Linear layers'
  // forward operands; a skipped update leaves parameters and copy as they were
  const float coef = norm ? norm[1] : 1.0f;
  const bool ok = norm ? (norm[2] != 0.f) : true;
  if (!ok) return;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  size_t stride = (size_t)gridDim.x * 256;
  const float step_size = lr / bc1, decay = 1.0f - lr * wd, omb1 = 1.0f - beta1, omb2 = 1.0f - beta2;
  size_t n4 = n / 4;
  // U groups of four elements per thread and iteration, all 4 * U loads issued before the first use: the launch keeps HBM busy
  // from 256 workgroups (one per CU, 4 of its 32 wave slots): 0.273 ms for the 61 M parameters of the VAE (6.2 TB/s) against
  // 0.300 ms for the earlier shape, 2048 grid-stride workgroups, which also held EVERY wave slot of the chip for the kernel's
  // whole life (in the rocprofv3 timeline a main-stream kernel arriving beside a side-stream AdamW launch waits for it).  The
  // step itself does not change with the shape (LO_ADAMW_BLOCKS = 256 / 512 / 1024 / 2048: 21.1 - 21.5 k sprites/s, no order).
  constexpr int U = 4;
  for (size_t k0 = i; k0 < n4; k0 += U * stride) {
    f32x4 pv[U], gv[U], mv[U], vv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t k = k0 + u * stride;
      if (k < n4) {
        pv[u] = reinterpret_cast<f32x4*>(p)[k];
        gv[u] = reinterpret_cast<const f32x4*>(g)[k];
        mv[u] = reinterpret_cast<f32x4*>(m)[k];
        vv[u] = reinterpret_cast<f32x4*>(v)[k];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t k = k0 + u * stride;
      if (k < n4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float pe = pv[u][j], me = mv[u][j], ve = vv[u][j];
          lo_adamw_elem(pe, gv[u][j], me, ve, coef, decay, omb1, beta2, omb2, bc2_sqrt, eps, step_size);
          pv[u][j] = pe; mv[u][j] = me; vv[u][j] = ve;
        }
        reinterpret_cast<f32x4*>(p)[k] = pv[u];
        reinterpret_cast<f32x4*>(m)[k] = mv[u];
        reinterpret_cast<f32x4*>(v)[k] = vv[u];
        if (cast) reinterpret_cast<f16x4*>(cast)[k] = (f16x4){(f16)pv[u][0], (f16)pv[u][1], (f16)pv[u][2], (f16)pv[u][3]};
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (size_t k = n4 * 4; k < n; ++k) {
      float pe = p[k], me = m[k], ve = v[k];
      lo_adamw_elem(pe, g[k], me, ve, coef, decay, omb1, beta2, omb2, bc2_sqrt, eps, step_size);
      p[k] = pe; m[k] = me; v[k] = ve;
      if (cast) cast[k] = (f16)pe;
    }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
int lo_head_reduce(const float* slab, const float* bias, const float* eps_in, uint64_t seed, float* mu, float* logvar,
                   f16* z, float* eps_out, float* kl_partial, int B, int L, int nsplit, hipStream_t st, float* mu_user,
                   float* logvar_user) {
  int nb = (B * L + 255) / 256;
  LoProfScope _p("lo_head_reduce", 0, 4.0 * B * 2 * L * (nsplit + 2), st);
  hipLaunchKernelGGL(lo_head_reduce_kernel, dim3(nb), dim3(256), 0, st, slab, bias, eps_in, seed, mu, logvar, z, eps_out,
                     kl_partial, B, L, nsplit, mu_user, logvar_user);
  LO_LAUNCH_CHECK("head_reduce");
  return LO_OK;
}
int lo_loss_finalize(const float* mse_partial, int n_mse, const float* kl_partial, int n_kl, float rw, float kw, float adv,
                     const float* adv_dev, float accum, float ls, float* losses, float* coefs, float n_rec, float n_lat,
                     hipStream_t st) {
  LoProfScope _p("lo_loss_finalize", 0, 0, st);
  hipLaunchKernelGGL(lo_loss_finalize_kernel, dim3(1), dim3(256), 0, st, mse_partial, n_mse, kl_partial, n_kl, rw, kw, adv,
                     accum, ls, adv_dev, losses, coefs, n_rec, n_lat);
  LO_LAUNCH_CHECK("loss_finalize");
  return LO_OK;
}
int lo_latent_bwd(const f16* dz, const float* mu, const float* logvar, const float* eps, const float* coefs,
                  const float* gmu, const float* glv, float gscale, f16* dml, int B, int L, hipStream_t st) {
  int n = B * L;
  LoProfScope _p("lo_latent_bwd", 0, 0, st);
  LO_LAUNCH_STOP(lo_latent_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, st, dz, mu, logvar, eps, coefs, gmu, glv,
                     gscale, dml, B, L);
  LO_LAUNCH_CHECK("latent_bwd");
  return LO_OK;
}
int lo_colsum_f16(const f16* x, float* out, int M, int N, float scale, hipStream_t st) {
  LO_REQUIRE(N % 8 == 0, "lo_colsum_f16: N=%d must be a multiple of 8", N);
  LoProfScope _p("lo_colsum_f16", 0, 2.0 * M * N, st);
  hipLaunchKernelGGL(lo_colsum_f16_kernel, dim3((N + 255) / 256), dim3(256), 0, st, x, out, M, N, scale);
  LO_LAUNCH_CHECK("colsum_f16");
  return LO_OK;
}
// x[i] *= scale (a few thousand elements: the latent gradient leaving lo_vae_decoder_backward un-scaled)
__global__ void lo_scale_f32_kernel(float* __restrict__ x, size_t n, float scale) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] *= scale;
}
int lo_scale_f32(float* x, size_t n, float scale, hipStream_t st) {
  hipLaunchKernelGGL(lo_scale_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, n, scale);
  LO_LAUNCH_CHECK("scale_f32");
  return LO_OK;
}
int lo_cast_f32_f16(const float* src, f16* dst, size_t n, hipStream_t st) {
  LO_REQUIRE(n % 4 == 0, "lo_cast_f32_f16: n must be a multiple of 4");
  size_t n4 = n / 4;
  int nb = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  LoProfScope _p("lo_cast_f32_f16", 0, 6.0 * n, st);
  hipLaunchKernelGGL(lo_cast_f32_f16_kernel, dim3(nb), dim3(256), 0, st, src, dst, n4);
  LO_LAUNCH_CHECK("cast");
  return LO_OK;
}
int lo_transpose_cast(const float* src, f16* dst, int R, int C, hipStream_t st) {
  LoProfScope _p("lo_transpose_cast", 0, 6.0 * R * C, st);
  hipLaunchKernelGGL(lo_transpose_cast_kernel, dim3((C + 63) / 64, (R + 63) / 64), dim3(256), 0, st, src, dst, R, C);
  LO_LAUNCH_CHECK("transpose_cast");
  return LO_OK;
}
#define LO_NORM_BLOCKS 1024
// Split form for a caller that already holds the sum of squares of the tail [presummed_begin, n) in partial[HALF .. 2*HALF)
// (lo_sumsq_range, issued earlier beside other work): only the head [0, presummed_begin) is read here.
int lo_sumsq_range(const float* g, size_t begin, size_t end, float* partial, hipStream_t st) {
  LoProfScope _p("lo_gradnorm (early range)", 0, 4.0 * (end - begin), st);
  hipLaunchKernelGGL(lo_sumsq_partial_kernel, dim3(LO_NORM_BLOCKS / 2), dim3(256), 0, st, g + begin, end - begin, partial + LO_NORM_BLOCKS / 2);
  LO_LAUNCH_CHECK("sumsq_range");
  return LO_OK;
}
int lo_sumsq_blocks(const float* g, size_t n, float* partial, int nblocks, hipStream_t st) {
  LoProfScope _p("lo_gradnorm (early range)", 0, 4.0 * n, st);
  hipLaunchKernelGGL(lo_sumsq_partial_kernel, dim3(nblocks), dim3(256), 0, st, g, n, partial);
  LO_LAUNCH_CHECK("sumsq_blocks");
  return LO_OK;
}
int lo_gradnorm_split(const float* g, size_t presummed_begin, float max_norm, float* partial, float* norm_out, hipStream_t st,
                      const unsigned int* fail) {
  LoProfScope _p("lo_gradnorm", 0, 4.0 * presummed_begin, st);
  hipLaunchKernelGGL(lo_sumsq_partial_kernel, dim3(LO_NORM_BLOCKS / 2), dim3(256), 0, st, g, presummed_begin, partial);
  LO_LAUNCH_CHECK("sumsq_head");
  hipLaunchKernelGGL(lo_gradnorm_finalize_kernel, dim3(1), dim3(256), 0, st, partial, LO_NORM_BLOCKS, max_norm, norm_out, fail);
  LO_LAUNCH_CHECK("gradnorm_finalize");
  return LO_OK;
}
int lo_gradnorm(const float* g, size_t n, float max_norm, float* partial /*>=1024*/, float* norm_out, hipStream_t st,
                const unsigned int* fail) {
  LoProfScope _p("lo_gradnorm", 0, 4.0 * n, st);
  hipLaunchKernelGGL(lo_sumsq_partial_kernel, dim3(LO_NORM_BLOCKS), dim3(256), 0, st, g, n, partial);
  LO_LAUNCH_CHECK("sumsq");
  hipLaunchKernelGGL(lo_gradnorm_finalize_kernel, dim3(1), dim3(256), 0, st, partial, LO_NORM_BLOCKS, max_norm, norm_out, fail);
  LO_LAUNCH_CHECK("gradnorm_finalize");
  return LO_OK;
}
int lo_adamw(float* p, const float* g, float* m, float* v, size_t n, const float* norm, float lr, float beta1, float beta2,
             float eps, float wd, int step, hipStream_t st, f16* cast) {
  float bc1 = 1.0f - powf(beta1, (float)step);
  float bc2 = 1.0f - powf(beta2, (float)step);
  double bc1d = 1.0 - pow((double)beta1, (double)step), bc2d = 1.0 - pow((double)beta2, (double)step);
  (void)bc1; (void)bc2;
  LoProfScope _p(cast ? "lo_adamw(+fp16 copy)" : "lo_adamw", 0, (cast ? 30.0 : 28.0) * n, st);
  constexpr int max_blocks = 256;   // 256 / 512 / 1024 / 2048 measured in round 2: no difference on the step; one workgroup per CU
  const size_t want = (n / 4 + 255) / 256;
  const int nblk = want >= (size_t)max_blocks ? max_blocks : (want < 1 ? 1 : (int)want);
  // (non-temporal loads / stores for p, g, m, v were measured in round 4, here and in lo_adamw_lowrank: 22 420-22 579 sprites/s against
  // 22 475-22 843 with the default policy -- nothing, or slightly worse)
  LO_LAUNCH_STOP(lo_adamw_kernel, dim3(nblk), dim3(256), 0, st, p, g, m, v, n, norm, lr, beta1, beta2, eps, wd,
                     (float)bc1d, (float)sqrt(bc2d), cast);
  LO_LAUNCH_CHECK("adamw");
  return LO_OK;
}

int lo_decode_sprites(const uint8_t* u8, float* out, int B, hipStream_t st) {
  int n = B * 16384;
  LoProfScope _p("lo_decode_sprites", 0, 15.0 * n, st);
  hipLaunchKernelGGL(lo_decode_sprites_kernel, dim3((n + 255) / 256), dim3(256), 0, st, u8, out, n);
  LO_LAUNCH_CHECK("decode_sprites");
  return LO_OK;
}

// ---------------------------------------------------------------------------------------------
// Data-parallel gradient exchange helpers (lunaris_orion_amd/parallel.py, mode "direct": all-to-all reduce-scatter + all-gather
// over all xGMI links; no reference counterpart -- the reference has no distributed code).  One streaming pass each:
//   pack:    wire[i] = fp16(g[i] * scale)                       (fp16 wire format: halves the xGMI bytes)
//   shares:  share[i] = (sum_r recv[r * chunk + i]) * inv_world (this rank's share of the reduce-scatter; fp32 accumulation, ranks
//            added in rank order: the same bits on every run)
//   unpack:  g[i] = float(wire[i]) * inv_scale
// ---------------------------------------------------------------------------------------------
// saturating: a pre-clip gradient element above 65504 / scale must not become inf on the wire -- the averaged norm would be
// non-finite, the update skipped and the loss scale halved for an overflow the loss scale did not cause; NaN stays NaN
__device__ __forceinline__ float lo_sat16(float x) { return x != x ? x : fminf(fmaxf(x, -65504.0f), 65504.0f); }
__global__ __launch_bounds__(256) void lo_dp_pack_f16_kernel(const float* __restrict__ g, f16* __restrict__ wire, size_t n, float scale) {
  const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 4 <= n) {
      const float4 v = *reinterpret_cast<const float4*>(g + i);
      f16x4 o = {(f16)lo_sat16(v.x * scale), (f16)lo_sat16(v.y * scale), (f16)lo_sat16(v.z * scale), (f16)lo_sat16(v.w * scale)};
      *reinterpret_cast<f16x4*>(wire + i) = o;
    } else {
      for (size_t j = i; j < n; ++j) wire[j] = (f16)lo_sat16(g[j] * scale);
    }
  }
}
__global__ __launch_bounds__(256) void lo_dp_unpack_f16_kernel(const f16* __restrict__ wire, float* __restrict__ g, size_t n, float inv_scale) {
  const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 4 <= n) {
      const f16x4 v = *reinterpret_cast<const f16x4*>(wire + i);
      *reinterpret_cast<float4*>(g + i) = make_float4((float)v[0] * inv_scale, (float)v[1] * inv_scale, (float)v[2] * inv_scale, (float)v[3] * inv_scale);
    } else {
      for (size_t j = i; j < n; ++j) g[j] = (float)wire[j] * inv_scale;
    }
  }
}
// unpack + the partial sums of squares of the unpacked values (the early part of the gradient norm: partial[blockIdx.x], same
// layout as lo_sumsq_partial_kernel with gridDim.x blocks): one pass over the range instead of two
__global__ __launch_bounds__(256) void lo_dp_unpack_f16_sumsq_kernel(const f16* __restrict__ wire, float* __restrict__ g, size_t n, float inv_scale,
                                                                     float* __restrict__ partial) {
  __shared__ float red[4];
  const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
  float acc = 0.f;
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 4 <= n) {
      const f16x4 v = *reinterpret_cast<const f16x4*>(wire + i);
      const float a = (float)v[0] * inv_scale, b = (float)v[1] * inv_scale, c = (float)v[2] * inv_scale, d = (float)v[3] * inv_scale;
      *reinterpret_cast<float4*>(g + i) = make_float4(a, b, c, d);
      acc += a * a + b * b + c * c + d * d;
    } else {
      for (size_t j = i; j < n; ++j) { const float a = (float)wire[j] * inv_scale; g[j] = a; acc += a * a; }
    }
  }
  const float w = lo_wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
template <typename T>
__global__ __launch_bounds__(256) void lo_dp_sum_shares_kernel(const T* __restrict__ recv, T* __restrict__ share, int world, size_t chunk, float inv_world) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < chunk; i += stride) {
    float t = 0.f;
    for (int r = 0; r < world; ++r) t += (float)recv[(size_t)r * chunk + i];
    share[i] = (T)(t * inv_world);
  }
}
static inline int lo_dp_blocks(size_t work_items) {
  size_t b = (work_items + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}
int lo_dp_pack_f16_run(const float* g, f16* wire, size_t n, float scale, hipStream_t st) {
  LoProfScope _p("lo_dp_pack_f16", 0, 6.0 * n, st);
  hipLaunchKernelGGL(lo_dp_pack_f16_kernel, dim3(lo_dp_blocks((n + 3) / 4)), dim3(256), 0, st, g, wire, n, scale);
  LO_LAUNCH_CHECK("dp_pack_f16");
  return LO_OK;
}
int lo_dp_unpack_f16_run(const f16* wire, float* g, size_t n, float inv_scale, hipStream_t st) {
  LoProfScope _p("lo_dp_unpack_f16", 0, 6.0 * n, st);
  hipLaunchKernelGGL(lo_dp_unpack_f16_kernel, dim3(lo_dp_blocks((n + 3) / 4)), dim3(256), 0, st, wire, g, n, inv_scale);
  LO_LAUNCH_CHECK("dp_unpack_f16");
  return LO_OK;
}
int lo_dp_unpack_f16_sumsq_run(const f16* wire, float* g, size_t n, float inv_scale, float* scratch, hipStream_t st) {
  LoProfScope _p("lo_dp_unpack_f16(+sumsq)", 0, 6.0 * n, st);
  hipLaunchKernelGGL(lo_dp_unpack_f16_sumsq_kernel, dim3(LO_NORM_BLOCKS / 2), dim3(256), 0, st, wire, g, n, inv_scale, scratch + LO_NORM_BLOCKS / 2);
  LO_LAUNCH_CHECK("dp_unpack_f16_sumsq");
  return LO_OK;
}
int lo_dp_sum_shares_run(const void* recv, void* share, int world, size_t chunk, int is_f16, float inv_world, hipStream_t st) {
  LoProfScope _p("lo_dp_sum_shares", 0, (is_f16 ? 2.0 : 4.0) * (world + 1) * chunk, st);
  if (is_f16) hipLaunchKernelGGL((lo_dp_sum_shares_kernel<f16>), dim3(lo_dp_blocks(chunk)), dim3(256), 0, st, (const f16*)recv, (f16*)share, world, chunk, inv_world);
  else hipLaunchKernelGGL((lo_dp_sum_shares_kernel<float>), dim3(lo_dp_blocks(chunk)), dim3(256), 0, st, (const float*)recv, (float*)share, world, chunk, inv_world);
  LO_LAUNCH_CHECK("dp_sum_shares");
  return LO_OK;
}

// ---------------------------------------------------------------------------------------------
// nn.Module boundary under torch.amp.GradScaler (train_hybrid.py:246-247, 289-297, 899-923).  The autograd nodes receive upstream
// gradients that a foreign loop has already multiplied by ITS loss scale (GradScaler starts at 65 536); the backward carries fp16
// activation gradients, so it needs them in a fixed range whatever that scale is.  lo_grad_scale_pick takes max |g| over the
// upstream tensors and leaves r = 2^k with max|g| * r in [2, 4) (and 1/r) in device memory -- a power of two, so scaling by r
// before the backward and by 1/r after it is exact in fp32 and no host synchronisation is needed; all-zero or non-finite
// gradients give r = 1 (non-finite values then propagate into the parameter gradients, where GradScaler.unscale_ finds them).
// ---------------------------------------------------------------------------------------------
struct LoAmaxJobs { const float* p[5]; size_t n[5]; };
__global__ __launch_bounds__(256) void lo_amax_partial_kernel(LoAmaxJobs J, float* __restrict__ partial) {
  __shared__ float red[4];
  float a = 0.f;
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const float* g = J.p[j];
    if (!g) continue;
    const size_t n = J.n[j], n4 = n / 4;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n4; k += (size_t)gridDim.x * 256) {
      const f32x4 v = reinterpret_cast<const f32x4*>(g)[k];
      a = fmaxf(a, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
      for (size_t k = n4 * 4; k < n; ++k) a = fmaxf(a, fabsf(g[k]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a = fmaxf(a, __shfl_xor(a, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__global__ __launch_bounds__(256) void lo_amax_finalize_kernel(const float* __restrict__ partial, int n, float* __restrict__ out2) {
  __shared__ float red[4];
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) a = fmaxf(a, partial[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a = fmaxf(a, __shfl_xor(a, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    a = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float r = 1.0f;
    if (a > 0.f && isfinite(a)) {
      int e;
      (void)frexpf(a, &e);                 // a = m 2^e, m in [0.5, 1)
      e = e > 100 ? 100 : (e < -100 ? -100 : e);
      r = ldexpf(1.0f, 2 - e);             // a r in [2, 4)
    }
    out2[0] = r;
    out2[1] = 1.0f / r;
    out2[2] = a;
  }
}
__global__ __launch_bounds__(256) void lo_scale_copy_dev_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t n,
                                                                const float* __restrict__ scale) {
  const float s = scale[0];
  const size_t n4 = n / 4;
  for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n4; k += (size_t)gridDim.x * 256) {
    f32x4 v = reinterpret_cast<const f32x4*>(src)[k];
    v *= s;
    reinterpret_cast<f32x4*>(dst)[k] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (size_t k = n4 * 4; k < n; ++k) dst[k] = src[k] * s;
}
// x *= scale[0]; a set rendezvous-failure word poisons the result with NaN instead (the autograd path has no optimizer kernel of
// this library behind it that could skip the update: a foreign loop's GradScaler / clip_grad_norm_ sees the NaN)
__global__ __launch_bounds__(256) void lo_scale_dev_kernel(float* __restrict__ x, size_t n, const float* __restrict__ scale,
                                                           const unsigned int* __restrict__ fail) {
  const float s = (fail && *fail != 0u) ? __builtin_nanf("") : scale[0];
  const size_t n4 = n / 4;
  for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n4; k += (size_t)gridDim.x * 256) {
    f32x4 v = reinterpret_cast<f32x4*>(x)[k];
    v *= s;
    reinterpret_cast<f32x4*>(x)[k] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (size_t k = n4 * 4; k < n; ++k) x[k] *= s;
}
static inline int lo_stream_blocks(size_t n) {
  size_t b = (n / 4 + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}
int lo_grad_scale_pick_run(const float* const g[5], const size_t n[5], float* scratch /* >= 260 floats */, hipStream_t st) {
  LoAmaxJobs J;
  size_t total = 0;
  for (int j = 0; j < 5; ++j) { J.p[j] = g[j]; J.n[j] = g[j] ? n[j] : 0; total += J.n[j]; }
  size_t want = (total / 4 + 255) / 256;
  const int nb = (int)(want < 1 ? 1 : (want > 256 ? 256 : want));
  hipLaunchKernelGGL(lo_amax_partial_kernel, dim3(nb), dim3(256), 0, st, J, scratch + 4);
  LO_LAUNCH_CHECK("amax_partial");
  hipLaunchKernelGGL(lo_amax_finalize_kernel, dim3(1), dim3(256), 0, st, scratch + 4, nb, scratch);
  LO_LAUNCH_CHECK("amax_finalize");
  return LO_OK;
}
int lo_scale_copy_dev_run(const float* src, float* dst, size_t n, const float* scale_dev, hipStream_t st) {
  hipLaunchKernelGGL(lo_scale_copy_dev_kernel, dim3(lo_stream_blocks(n)), dim3(256), 0, st, src, dst, n, scale_dev);
  LO_LAUNCH_CHECK("scale_copy_dev");
  return LO_OK;
}
int lo_scale_dev_run(float* x, size_t n, const float* scale_dev, const unsigned int* fail, hipStream_t st) {
  hipLaunchKernelGGL(lo_scale_dev_kernel, dim3(lo_stream_blocks(n)), dim3(256), 0, st, x, n, scale_dev, fail);
  LO_LAUNCH_CHECK("scale_dev");
  return LO_OK;
}
