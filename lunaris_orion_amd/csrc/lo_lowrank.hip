// The three Linear layers of the VAE (fc_mu | fc_logvar as one [2L, 32768] matrix, decoder.fc [32768, L]; lunar_generate.py:124-125,
// 150-152, 165, 207-208) hold 82 % of its parameters, and at batch B their weight gradients dW = dY^T X are rank-B matrices (B = 64
// against 512 .. 32768 rows and columns).  The fused training step never writes them:
//
//   * clip_grad_norm_ (train_hybrid.py:913):  ||dW||_F^2 = sum_{b,b'} (dY dY^T)[b,b'] (X X^T)[b,b']  -- two B x B Gram matrices
//     (lo_gram_small for the factor with the short rows, lo_gram_dot for the one with the long rows; the second one multiplies its
//     partial Gram blocks with the first one's entries on the way and leaves one scalar per wave in the norm's partial-sum slots);
//   * AdamW (train_hybrid.py:921):  lo_adamw_lowrank forms each 16 x 16 gradient tile on the fly -- two v_mfma_f32_16x16x32_f16 per
//     tile at B = 64, operands the transposed factors X^T [K][B], dY^T [N][B] (8.6 MB in all: cache-resident) -- and runs the
//     element update of lo_adamw on the accumulators: 24 bytes of HBM traffic per parameter (+2 for the fp16 operand copy) instead
//     of 28 + 2, and neither the 4 bytes written by a weight-gradient GEMM nor the 4 read by the norm exist.
//
// The MFMA tile is laid out so that the accumulator rows are the MEMORY-CONTIGUOUS axis of the weight matrix: D[i][j] with i = k
// (input feature) and j = n (output feature), A = X^T rows, B = dY^T rows.  A lane then owns four consecutive k of one n
// (D row = 4 (lane >> 4) + r, column = lane & 15): one 16-byte load / store per tensor and tile, 64 contiguous bytes per matrix row
// and instruction, 256 per row over the four k tiles of a wave's 64 x 64 block.
// A materialising variant (same tiles, same arithmetic, gradient written to the flat buffer) exists for tests and for
// lo_vae_materialize_linear_grads.  No reference counterpart below aten::linear_backward / torch.optim.AdamW.
#include "lo_common.h"
#include "lo_internal.h"
#include <math.h>
#include <string.h>

// ---------------------------------------------------------------------------------------------
// dst[c][r] = src[r][c] (r < R), 0 for R <= r < Rp: a factor [B][C] -> its transposed, batch-padded form [C][Bp]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lo_transpose_pad_f16_kernel(const f16* __restrict__ src, f16* __restrict__ dst, int R, int C, int Rp) {
  __shared__ f16 t[64][72];
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
  const int tid = threadIdx.x;
  {
    const int r = tid >> 2, cc = (tid & 3) * 16;
    f16x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = a;
    if (r0 + r < R) {
      const f16* p = src + (size_t)(r0 + r) * C + c0 + cc;
      a = *reinterpret_cast<const f16x8*>(p);
      b = *reinterpret_cast<const f16x8*>(p + 8);
    }
    *reinterpret_cast<f16x8*>(&t[r][cc]) = a;
    *reinterpret_cast<f16x8*>(&t[r][cc + 8]) = b;
  }
  __syncthreads();
  {
    const int c = tid >> 2, rr = (tid & 3) * 16;
    if (r0 + rr < Rp) {
      f16x8 a, b;
#pragma unroll
      for (int q = 0; q < 8; ++q) { a[q] = t[rr + q][c]; b[q] = t[rr + 8 + q][c]; }
      f16* p = dst + (size_t)(c0 + c) * Rp + r0 + rr;
      *reinterpret_cast<f16x8*>(p) = a;
      *reinterpret_cast<f16x8*>(p + 8) = b;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Gram matrices.  src: a factor in its natural layout [B][n] (row b = sample b).  Fragment of rows 16 bi .. 16 bi + 15 for the
// 32 columns of step kc: lane l holds row 16 bi + (l & 15), columns 32 kc + 8 (l >> 4) .. + 7 (one 16-byte load; rows >= B read
// as zero).  The same register is the A operand (row index = output row) and the B operand (row index = output column) of
// v_mfma_f32_16x16x32_f16, so tile (bi, bj) of src src^T is mfma(frag(bi), frag(bj)).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ f16x8 lo_gram_frag(const f16* __restrict__ src, int B, int n, int bi, int kc, int lane) {
  const int r = bi * 16 + (lane & 15);
  f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  if (r >= B) return z;
  return *reinterpret_cast<const f16x8*>(src + (size_t)r * n + kc * 32 + (lane >> 4) * 8);
}
// the factor with the short rows (n = 2L or L): whole Gram matrix [Bp][Bp] (fp32) from one workgroup
__global__ __launch_bounds__(256) void lo_gram_small_kernel(const f16* __restrict__ src, int B, int n, int Bp, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int Bt = Bp / 16;
  for (int tile = wave; tile < Bt * Bt; tile += 4) {
    const int bi = tile / Bt, bj = tile - bi * Bt;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int kc = 0; kc < n / 32; ++kc)
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(lo_gram_frag(src, B, n, bi, kc, lane), lo_gram_frag(src, B, n, bj, kc, lane), acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(size_t)(bi * 16 + (lane >> 4) * 4 + r) * Bp + bj * 16 + (lane & 15)] = acc[r];
  }
}
// the factor with the long rows (n = 32768): one wave per (column chunk, 64 x 64 block of the Gram matrix); the block's partial
// sums are multiplied with the other factor's Gram entries and reduced to ONE scalar per wave:
//   partial[wave] = scale * sum_{b,b' in block} small[b][b'] * (sum_{c in chunk} src[b][c] src[b'][c])
// -- the chunk's share of ||dW||_F^2 (non-negative: it is the squared norm of the gradient's columns / rows of that chunk).
__global__ __launch_bounds__(64) void lo_gram_dot_kernel(const f16* __restrict__ src, int B, int n, int Bp, const float* __restrict__ small,
                                                         int steps_per_chunk, float scale, float* __restrict__ partial) {
  const int lane = threadIdx.x;
  const int Bt = Bp / 16;
  const int bi0 = blockIdx.y * 4, bj0 = blockIdx.z * 4;
  const int kc0 = blockIdx.x * steps_per_chunk;
  int kc1 = kc0 + steps_per_chunk;
  if (kc1 > n / 32) kc1 = n / 32;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool diag = bi0 == bj0;
  for (int kc = kc0; kc < kc1; ++kc) {
    f16x8 fa[4], fb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = lo_gram_frag(src, B, n, bi0 + i, kc, lane);
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[j] = diag ? fa[j] : lo_gram_frag(src, B, n, bj0 + j, kc, lane);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (bi0 + i < Bt && bj0 + j < Bt) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          s += acc[i][j][r] * small[(size_t)((bi0 + i) * 16 + (lane >> 4) * 4 + r) * Bp + (bj0 + j) * 16 + (lane & 15)];
      }
  s = lo_wave_sum(s);
  if (lane == 0) partial[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = s * scale;
}

// ---------------------------------------------------------------------------------------------
// AdamW on a weight matrix whose gradient is given by its factors
// ---------------------------------------------------------------------------------------------
struct LoLowrankArgs {
  float* p; float* m; float* v;   // [N][K] fp32 (parameter, Adam moments)
  f16* cast;                      // [N][K] fp16 operand copy of the updated parameter (may be null)
  float* gout;                    // materialising variant: the gradient [N][K]
  const f16* xt;                  // X^T  [K][Bp]
  const f16* yt;                  // dY^T [N][Bp]
  int N, K, Bp;
  float gscale;                   // 1 / loss scale
  const float* norm;              // [1] clip coefficient, [2] finite flag (lo_gradnorm)
  float lr, beta1, beta2, eps, wd, bc1, bc2_sqrt;
};

// the element update of lo_adamw (lo_train.hip): same operations in the same order, contraction off
__device__ __forceinline__ void lo_adamw_elem_lr(float& p, float g, float& m, float& v, float coef, float decay, float omb1, float beta2,
                                                 float omb2, float bc2_sqrt, float eps, float step_size) {
#pragma clang fp contract(off)
  const float gg = g * coef;
  const float pp = p * decay;
  const float mm = __builtin_fmaf(gg - m, omb1, m);
  const float v2 = __builtin_fmaf(omb2 * gg, gg, v * beta2);
  const float denom = sqrtf(v2) / bc2_sqrt + eps;
  p = __builtin_fmaf(-step_size, mm / denom, pp);
  m = mm;
  v = v2;
}

// One wave per 64 (n) x 64 (k) block, grid-stride over the blocks (k fastest: neighbouring waves walk along the rows of one
// 64-row band).  KB = Bp / 32 MFMA K steps.  Per block: the 4 x KB X^T fragments once; then four 16-row groups, each: KB dY^T
// fragments, 4 tiles x (p, m, v) 16-byte loads -- the NEXT group's twelve loads are issued before this group's MFMAs and update,
// so a lane keeps 12 - 24 loads in flight (48 - 96 KB per CU at one workgroup per CU: the streaming depth lo_adamw runs at).
template <int KB, bool MAT>
__global__ __launch_bounds__(256) void lo_adamw_lowrank_kernel(LoLowrankArgs a) {
  const float coef = a.norm ? a.norm[1] : 1.0f;
  if (!MAT && a.norm && a.norm[2] == 0.f) return;        // non-finite norm / lost launch: the update is skipped
  const int lane = threadIdx.x & 63, l15 = lane & 15, lq = lane >> 4;
  const int tiles_k = a.K / 64, total = (a.N / 64) * tiles_k;
  const float step_size = a.lr / a.bc1, decay = 1.0f - a.lr * a.wd, omb1 = 1.0f - a.beta1, omb2 = 1.0f - a.beta2;
  const int Bp = a.Bp;
  for (int t = blockIdx.x * 4 + (threadIdx.x >> 6); t < total; t += gridDim.x * 4) {
    const int tn = t / tiles_k, tk = t - tn * tiles_k;
    const int n0 = tn * 64, k0 = tk * 64;
    f16x8 xf[4][KB];
#pragma unroll
    for (int ik = 0; ik < 4; ++ik)
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
        xf[ik][kb] = *reinterpret_cast<const f16x8*>(a.xt + (size_t)(k0 + 16 * ik + l15) * Bp + 32 * kb + 8 * lq);
    f32x4 pv[2][4], mv[2][4], vv[2][4];
    f16x8 yf[2][KB];
    auto load_group = [&](int jn, int buf) {
      const size_t row = (size_t)(n0 + 16 * jn + l15);
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) yf[buf][kb] = *reinterpret_cast<const f16x8*>(a.yt + row * Bp + 32 * kb + 8 * lq);
      if (!MAT) {
        const size_t base = row * a.K + k0 + 4 * lq;
#pragma unroll
        for (int ik = 0; ik < 4; ++ik) {
          pv[buf][ik] = *reinterpret_cast<const f32x4*>(a.p + base + 16 * ik);
          mv[buf][ik] = *reinterpret_cast<const f32x4*>(a.m + base + 16 * ik);
          vv[buf][ik] = *reinterpret_cast<const f32x4*>(a.v + base + 16 * ik);
        }
      }
    };
    load_group(0, 0);
#pragma unroll
    for (int jn = 0; jn < 4; ++jn) {
      const int buf = jn & 1;
      if (jn + 1 < 4) load_group(jn + 1, buf ^ 1);
      const size_t base = (size_t)(n0 + 16 * jn + l15) * a.K + k0 + 4 * lq;
#pragma unroll
      for (int ik = 0; ik < 4; ++ik) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf[ik][kb], yf[buf][kb], acc, 0, 0, 0);
        if (MAT) {
          f32x4 g = {acc[0] * a.gscale, acc[1] * a.gscale, acc[2] * a.gscale, acc[3] * a.gscale};
          *reinterpret_cast<f32x4*>(a.gout + base + 16 * ik) = g;
        } else {
          f32x4 pe = pv[buf][ik], me = mv[buf][ik], ve = vv[buf][ik];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float pr = pe[r], mr = me[r], vr = ve[r];
            lo_adamw_elem_lr(pr, acc[r] * a.gscale, mr, vr, coef, decay, omb1, a.beta2, omb2, a.bc2_sqrt, a.eps, step_size);
            pe[r] = pr; me[r] = mr; ve[r] = vr;
          }
          *reinterpret_cast<f32x4*>(a.p + base + 16 * ik) = pe;
          *reinterpret_cast<f32x4*>(a.m + base + 16 * ik) = me;
          *reinterpret_cast<f32x4*>(a.v + base + 16 * ik) = ve;
          if (a.cast) *reinterpret_cast<f16x4*>(a.cast + base + 16 * ik) = (f16x4){(f16)pe[0], (f16)pe[1], (f16)pe[2], (f16)pe[3]};
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
int lo_lowrank_bp(int B) { return (B + 31) / 32 * 32; }
bool lo_lowrank_applies(int B, int N, int K) { return B >= 1 && B <= 128 && N % 64 == 0 && K % 64 == 0; }

int lo_transpose_pad_f16(const f16* src, f16* dst, int R, int C, int Rp, hipStream_t st) {
  LO_REQUIRE(C % 64 == 0 && Rp % 16 == 0 && Rp >= R, "lo_transpose_pad_f16: C=%d must be a multiple of 64, Rp=%d a multiple of 16 >= R=%d", C, Rp, R);
  LoProfScope _p("lo_factor_transpose", 0, 2.0 * ((double)R + Rp) * C, st);
  hipLaunchKernelGGL(lo_transpose_pad_f16_kernel, dim3(C / 64, (Rp + 63) / 64), dim3(256), 0, st, src, dst, R, C, Rp);
  LO_LAUNCH_CHECK("transpose_pad_f16");
  return LO_OK;
}

// ||dY^T X||_F^2 * scale^2 of one Linear layer into `partial[0 .. nslots)` (every slot is written).  fshort / flong: the two factors
// in their natural layouts [B][n_short], [B][n_long]; gram: [Bp][Bp] fp32 scratch.
int lo_lowrank_sumsq(const f16* fshort, int n_short, const f16* flong, int n_long, int B, float scale, float* gram, float* partial,
                     int nslots, hipStream_t st) {
  const int Bp = lo_lowrank_bp(B);
  LO_REQUIRE(n_short % 32 == 0 && n_long % 32 == 0 && Bp <= 128, "lo_lowrank_sumsq: bad shape");
  const int nblk = (Bp / 16 + 3) / 4;
  int chunks = nslots / (nblk * nblk);
  LO_REQUIRE(chunks >= 1, "lo_lowrank_sumsq: %d partial slots are too few", nslots);
  const int steps = n_long / 32;
  if (chunks > steps) chunks = steps;
  const int per = (steps + chunks - 1) / chunks;
  chunks = (steps + per - 1) / per;
  LoProfScope _p("lo_lowrank_sumsq", 0, 2.0 * B * ((double)n_short + n_long), st);
  hipLaunchKernelGGL(lo_gram_small_kernel, dim3(1), dim3(256), 0, st, fshort, B, n_short, Bp, gram);
  LO_LAUNCH_CHECK("gram_small");
  const int used = chunks * nblk * nblk;
  if (used < nslots) LO_HIP(hipMemsetAsync(partial + used, 0, (size_t)(nslots - used) * sizeof(float), st));
  hipLaunchKernelGGL(lo_gram_dot_kernel, dim3(chunks, nblk, nblk), dim3(64), 0, st, flong, B, n_long, Bp, gram, per, scale * scale, partial);
  LO_LAUNCH_CHECK("gram_dot");
  return LO_OK;
}

static int lowrank_launch(const LoLowrankArgs& a, bool mat, hipStream_t st) {
  const int KB = a.Bp / 32;
  const int total = (a.N / 64) * (a.K / 64);
  int nblk = (total + 3) / 4;
  if (nblk > 256) nblk = 256;                        // one workgroup per CU, like lo_adamw
#define LO_LR(KBV)                                                                                                     \
  do {                                                                                                                 \
    if (mat) hipLaunchKernelGGL((lo_adamw_lowrank_kernel<KBV, true>), dim3(nblk), dim3(256), 0, st, a);                \
    else LO_LAUNCH_STOP((lo_adamw_lowrank_kernel<KBV, false>), dim3(nblk), dim3(256), 0, st, a);                       \
  } while (0)
  if (KB == 1) LO_LR(1);
  else if (KB == 2) LO_LR(2);
  else if (KB == 3) LO_LR(3);
  else if (KB == 4) LO_LR(4);
  else { lo_set_error("lo_adamw_lowrank: batch padding %d not supported", a.Bp); return LO_ERR_ARG; }
#undef LO_LR
  LO_LAUNCH_CHECK("adamw_lowrank");
  return LO_OK;
}

int lo_adamw_lowrank(float* p, float* m, float* v, f16* cast, const f16* xt, const f16* yt, int N, int K, int B, float gscale,
                     const float* norm, float lr, float beta1, float beta2, float eps, float wd, int step, hipStream_t st) {
  LO_REQUIRE(lo_lowrank_applies(B, N, K), "lo_adamw_lowrank: shape B=%d N=%d K=%d not supported", B, N, K);
  LoLowrankArgs a;
  a.p = p; a.m = m; a.v = v; a.cast = cast; a.gout = nullptr; a.xt = xt; a.yt = yt; a.N = N; a.K = K; a.Bp = lo_lowrank_bp(B);
  a.gscale = gscale; a.norm = norm; a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = wd;
  a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  const double elems = (double)N * K;
  LoProfScope _p("lo_adamw_lowrank", 0, (cast ? 26.0 : 24.0) * elems, st);     // HBM-bound: rated against the HBM roofline (2 Bp FLOP per element ride along)
  return lowrank_launch(a, false, st);
}

int lo_lowrank_materialize(float* gout, const f16* xt, const f16* yt, int N, int K, int B, float gscale, hipStream_t st) {
  LO_REQUIRE(lo_lowrank_applies(B, N, K), "lo_lowrank_materialize: shape B=%d N=%d K=%d not supported", B, N, K);
  LoLowrankArgs a;
  memset(&a, 0, sizeof(a));
  a.gout = gout; a.xt = xt; a.yt = yt; a.N = N; a.K = K; a.Bp = lo_lowrank_bp(B); a.gscale = gscale;
  LoProfScope _p("lo_lowrank_materialize", 0, 4.0 * (double)N * K, st);
  return lowrank_launch(a, true, st);
}
