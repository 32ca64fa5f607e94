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
#include <stdlib.h>

// ---------------------------------------------------------------------------------------------
// dst[c][r] = src[r][c] (r < R), 0 for R <= r < Rp: a factor [B][C] -> its transposed, batch-padded form [C][Bp].  All factors of
// a step in ONE launch (job table by value): the four launches this replaces cost 7 us each for 8.6 MB in all.
// ---------------------------------------------------------------------------------------------
struct LoTransposeJob { const f16* src; f16* dst; int C, tile0; };
struct LoTransposeJobs { LoTransposeJob j[4]; int n, R, Rp; };
__global__ __launch_bounds__(256) void lo_transpose_pad_f16_kernel(LoTransposeJobs J) {
  __shared__ f16 t[64][72];
  int ji = 0;
#pragma unroll
  for (int q = 1; q < 4; ++q)
    if (q < J.n && (int)blockIdx.x >= J.j[q].tile0) ji = q;
  const LoTransposeJob job = J.j[ji];
  const int R = J.R, Rp = J.Rp, C = job.C;
  const int tile = blockIdx.x - job.tile0, ctiles = C / 64;
  const int c0 = (tile % ctiles) * 64, r0 = (tile / ctiles) * 64;
  const int tid = threadIdx.x;
  {
    const int r = tid >> 2, cc = (tid & 3) * 16;
    f16x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = a;
    if (r0 + r < R) {
      const f16* p = job.src + (size_t)(r0 + r) * C + c0 + cc;
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
      f16* p = job.dst + (size_t)(c0 + c) * Rp + r0 + rr;
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
// Both Linear layers in one launch each (blockIdx.y / blockIdx.z = layer): a first version with one launch per layer and kernel,
// one workgroup for the small matrix and dependent loads inside the step loops, took 36 us per layer; the loads are now issued in
// batches of 8 steps (small) / 4 steps (long) before the MFMAs that consume them.
// ---------------------------------------------------------------------------------------------
struct LoGramLayer { const f16* fshort; const f16* flong; float* gram; float* partial; int n_short, n_long, chunks, per, nslots; };
struct LoGramJobs { LoGramLayer l[2]; int B, Bp; float scale2; };
__device__ __forceinline__ f16x8 lo_gram_frag(const f16* __restrict__ src, int B, int n, int bi, int kc, int lane) {
  const int r = bi * 16 + (lane & 15);
  f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  if (r >= B) return z;
  return *reinterpret_cast<const f16x8*>(src + (size_t)r * n + kc * 32 + (lane >> 4) * 8);
}
// the factor with the short rows (n = 2L or L): its Gram matrix [Bp][Bp] (fp32).  One workgroup per 16 x 16 tile and layer; the
// four waves split the row length, each issues ALL its fragment loads before its first MFMA (up to 8 steps = 16 loads), and the
// four partial tiles are added in wave order through LDS.  (One 1024-thread workgroup per layer took 26 us: 256 KB of loads
// through one CU's vector memory path, four dependent batches.)
__global__ __launch_bounds__(256) void lo_gram_small_kernel(LoGramJobs J) {
  __shared__ f32x4 part[4][64];
  const LoGramLayer L = J.l[blockIdx.y];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int B = J.B, Bp = J.Bp, Bt = Bp / 16, n = L.n_short, steps = n / 32;
  const int tile = blockIdx.x;
  if (tile >= Bt * Bt) return;
  const int bi = tile / Bt, bj = tile - bi * Bt;
  const int per = (steps + 3) / 4, k0 = wave * per, k1 = k0 + per < steps ? k0 + per : steps;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int kc0 = k0; kc0 < k1; kc0 += 8) {
    f16x8 fa[8], fb[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int kc = kc0 + q < k1 ? kc0 + q : k1 - 1;
      fa[q] = lo_gram_frag(L.fshort, B, n, bi, kc, lane);
      fb[q] = bi == bj ? fa[q] : lo_gram_frag(L.fshort, B, n, bj, kc, lane);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (kc0 + q < k1) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[q], fb[q], acc, 0, 0, 0);
  }
  part[wave][lane] = acc;
  __syncthreads();
  if (wave == 0) {
    const f32x4 t = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
#pragma unroll
    for (int r = 0; r < 4; ++r) L.gram[(size_t)(bi * 16 + (lane >> 4) * 4 + r) * Bp + bj * 16 + (lane & 15)] = t[r];
  }
}
// the factor with the long rows (n = 32768): one wave per (column chunk, 64 x 64 block of the Gram matrix); the block's partial
// sums are multiplied with the other factor's Gram entries and reduced to ONE scalar per wave:
//   partial[slot] = scale^2 * sum_{b,b' in block} small[b][b'] * (sum_{c in chunk} src[b][c] src[b'][c])
// -- the chunk's share of ||dW||_F^2 (non-negative: it is the squared norm of the gradient's columns / rows of that chunk).
// Every one of the layer's `nslots` partial-sum slots is written (unused ones with zero).
__global__ __launch_bounds__(64) void lo_gram_dot_kernel(LoGramJobs J) {
  const LoGramLayer L = J.l[blockIdx.y];
  const int lane = threadIdx.x;
  const int B = J.B, Bp = J.Bp, Bt = Bp / 16, nblk = (Bt + 3) / 4, n = L.n_long;
  const int slot = blockIdx.x;
  const int blk = slot / L.chunks, chunk = slot - blk * L.chunks;
  if (blk >= nblk * nblk) {
    if (lane == 0 && slot < L.nslots) L.partial[slot] = 0.f;
    return;
  }
  const int bi0 = (blk / nblk) * 4, bj0 = (blk % nblk) * 4;
  const int kc0 = chunk * L.per;
  int kc1 = kc0 + L.per;
  if (kc1 > n / 32) kc1 = n / 32;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool diag = bi0 == bj0;
  for (int kb = kc0; kb < kc1; kb += 4) {
    f16x8 fa[4][4], fb[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int kc = kb + q < kc1 ? kb + q : kc1 - 1;
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[q][i] = lo_gram_frag(L.flong, B, n, bi0 + i, kc, lane);
      if (!diag) {
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[q][j] = lo_gram_frag(L.flong, B, n, bj0 + j, kc, lane);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (kb + q < kc1) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[q][i], diag ? fa[q][j] : fb[q][j], acc[i][j], 0, 0, 0);
      }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (bi0 + i < Bt && bj0 + j < Bt) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          s += acc[i][j][r] * L.gram[(size_t)((bi0 + i) * 16 + (lane >> 4) * 4 + r) * Bp + (bj0 + j) * 16 + (lane & 15)];
      }
  s = lo_wave_sum(s);
  if (lane == 0) L.partial[slot] = s * J.scale2;
}

// ---------------------------------------------------------------------------------------------
// AdamW on a weight matrix whose gradient is given by its factors
// ---------------------------------------------------------------------------------------------
struct LoLowrankArgs {
  float* p; float* m; float* v;   // [N][K] fp32 (parameter, Adam moments)
  f16* cast;                      // [N][K] fp16 operand copy of the updated parameter (may be null)
  f16* cast_t;                    // [K][N] the same, transposed (operand of the layer's data gradient; may be null)
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

// One wave owns a 64-column band of the matrix (k0 .. k0 + 63: its 4 x KB X^T fragments are loaded ONCE) and walks down the rows
// in 16-row groups: per group KB dY^T fragments and 4 tiles x (p, m, v) 16-byte loads.  The walk is software-pipelined over the
// groups, across 64-row tile boundaries: the loads of group g + 1 are issued before the MFMAs and the update of group g, so a lane
// always has 12 - 24 loads in flight; two workgroups per CU (8 waves, <= 256 registers each) keep ~100 KB per CU outstanding.
// Both matrices of the model share ONE launch (LoLowrankPair): one ramp and one tail instead of two.
// Neighbouring waves own neighbouring bands: the 256-byte row segments of the waves of a workgroup are contiguous in memory.
// (The first version -- one wave per 64 x 64 block, nothing in flight across blocks -- ran at 4.8 TB/s against lo_adamw's 6.2.)
struct LoLowrankPair { LoLowrankArgs l[2]; int nwg0; };      // workgroups [0, nwg0) update matrix 0, the rest matrix 1
// cast_t: the transposed fp16 copy leaves through LDS -- a wave's 64 x 64 block of updated parameters is collected [k][n] in the
// wave's own 9 KB of LDS (2-byte writes as the groups complete) and stored as 64 rows of 128 contiguous bytes once the block's four
// groups are done: 2 more bytes per parameter on this pass instead of a separate transpose pass over both fp16 copies (4 bytes per
// parameter, 45 us per step for the two matrices).
#define LR_TP 72     // LDS pitch of a transposed block in halves (64 + 8: rows 144 bytes apart)
template <int KB, bool MAT>
__global__ __launch_bounds__(256) void lo_adamw_lowrank_kernel(LoLowrankPair A) {
  __shared__ __attribute__((aligned(16))) f16 tbuf[MAT ? 1 : 4][MAT ? 8 : 64 * LR_TP];
  const bool second = (int)blockIdx.x >= A.nwg0;
  const LoLowrankArgs& a = second ? A.l[1] : A.l[0];
  const int bid = second ? (int)blockIdx.x - A.nwg0 : (int)blockIdx.x, nbl = second ? (int)gridDim.x - A.nwg0 : A.nwg0;
  const float coef = a.norm ? a.norm[1] : 1.0f;
  if (!MAT && a.norm && a.norm[2] == 0.f) return;        // non-finite norm / lost launch: the update is skipped
  const int lane = threadIdx.x & 63, l15 = lane & 15, lq = lane >> 4;
  const int bands = a.K / 64, tiles_n = a.N / 64;
  const int W = nbl * 4, gw = bid * 4 + (threadIdx.x >> 6);
  const int wpb = W >= bands ? W / bands : 1;            // waves per band
  const float step_size = a.lr / a.bc1, decay = 1.0f - a.lr * a.wd, omb1 = 1.0f - a.beta1, omb2 = 1.0f - a.beta2;
  const int Bp = a.Bp;
  for (int vw = gw; vw < bands * wpb; vw += W) {
    const int band = vw % bands, sub = vw / bands;
    if (sub >= tiles_n) continue;
    const int k0 = band * 64;
    const int ngroups = ((tiles_n - sub + wpb - 1) / wpb) * 4;
    f16x8 xf[4][KB];
#pragma unroll
    for (int ik = 0; ik < 4; ++ik)
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
        xf[ik][kb] = *reinterpret_cast<const f16x8*>(a.xt + (size_t)(k0 + 16 * ik + l15) * Bp + 32 * kb + 8 * lq);
    f32x4 pv[2][4], mv[2][4], vv[2][4];
    f16x8 yf[2][KB];
    auto row_of = [&](int g) { return (size_t)((sub + (g >> 2) * wpb) * 64 + 16 * (g & 3) + l15); };
    auto load_group = [&](int g, int buf) {
      const size_t row = row_of(g);
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
    auto do_group = [&](int g, int buf) {
      const size_t base = row_of(g) * a.K + k0 + 4 * lq;
#pragma unroll
      for (int ik = 0; ik < 4; ++ik) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf[ik][kb], yf[buf][kb], acc, 0, 0, 0);
        if (MAT) {
          f32x4 gr = {acc[0] * a.gscale, acc[1] * a.gscale, acc[2] * a.gscale, acc[3] * a.gscale};
          *reinterpret_cast<f32x4*>(a.gout + base + 16 * ik) = gr;
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
          if (a.cast_t) {
            f16* tb = &tbuf[threadIdx.x >> 6][0];
            const int nloc = 16 * (g & 3) + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) tb[(16 * ik + 4 * lq + r) * LR_TP + nloc] = (f16)pe[r];
          }
        }
      }
      if (!MAT && a.cast_t && (g & 3) == 3) {
        // the block's four groups are in: 64 rows (k) x 64 columns (n); instruction j stores rows 8 j .. 8 j + 7, 128 bytes each
        __builtin_amdgcn_wave_barrier();
        const f16* tb = &tbuf[threadIdx.x >> 6][0];
        const size_t n_first = (size_t)(sub + (g >> 2) * wpb) * 64;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int kk = 8 * j + (lane >> 3), ch = lane & 7;
          *reinterpret_cast<f16x8*>(a.cast_t + (size_t)(k0 + kk) * a.N + n_first + 8 * ch) = *reinterpret_cast<const f16x8*>(tb + kk * LR_TP + 8 * ch);
        }
        __builtin_amdgcn_wave_barrier();
      }
    };
    load_group(0, 0);
    for (int g = 0; g < ngroups; g += 2) {               // ngroups is a multiple of 4
      load_group(g + 1, 1);
      do_group(g, 0);
      if (g + 2 < ngroups) load_group(g + 2, 0);
      do_group(g + 1, 1);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Data parallel: the ranks all-gather their FACTORS (8.6 MB of fp16 per rank at batch 64 / latent 512, exact) instead of exchanging
// 201 MB of weight gradients, and every rank forms the averaged gradient itself:  dW = (1 / N) sum_r dY_r^T X_r  -- the same tiles
// with the contraction running over the N gathered blocks (K = N * Bp).  One wave per 64 x 64 block of the matrix, 16 accumulator
// tiles in registers; per (rank, 32-sample step) 4 + 4 fragment loads (L2-resident: 2 N KB per 4096 elements) and 16 MFMAs.
// xt / yt point at the factor inside block 0; rstride = elements from one rank's block to the next.
// ---------------------------------------------------------------------------------------------
struct LoGatherMatArgs { float* gout; const f16* xt; const f16* yt; size_t rstride; int N, K, Bp, world; float gscale; };
__global__ __launch_bounds__(256) void lo_lowrank_materialize_gathered_kernel(LoGatherMatArgs a) {
  const int lane = threadIdx.x & 63, l15 = lane & 15, lq = lane >> 4;
  const int tiles_k = a.K / 64, total = (a.N / 64) * tiles_k, KB = a.Bp / 32;
  for (int t = blockIdx.x * 4 + (threadIdx.x >> 6); t < total; t += gridDim.x * 4) {
    const int tn = t / tiles_k, tk = t - tn * tiles_k;
    const int n0 = tn * 64, k0 = tk * 64;
    f32x4 acc[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < a.world; ++r) {
      const f16* xr = a.xt + (size_t)r * a.rstride;
      const f16* yr = a.yt + (size_t)r * a.rstride;
      for (int kb = 0; kb < KB; ++kb) {
        f16x8 xf[4], yf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) xf[i] = *reinterpret_cast<const f16x8*>(xr + (size_t)(k0 + 16 * i + l15) * a.Bp + 32 * kb + 8 * lq);
#pragma unroll
        for (int j = 0; j < 4; ++j) yf[j] = *reinterpret_cast<const f16x8*>(yr + (size_t)(n0 + 16 * j + l15) * a.Bp + 32 * kb + 8 * lq);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf[i], yf[j], acc[j][i], 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4 g = {acc[j][i][0] * a.gscale, acc[j][i][1] * a.gscale, acc[j][i][2] * a.gscale, acc[j][i][3] * a.gscale};
        *reinterpret_cast<f32x4*>(a.gout + (size_t)(n0 + 16 * j + l15) * a.K + k0 + 16 * i + 4 * lq) = g;
      }
  }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
int lo_lowrank_bp(int B) { return (B + 31) / 32 * 32; }
bool lo_lowrank_applies(int B, int N, int K) { return B >= 1 && B <= 128 && N % 64 == 0 && K % 64 == 0; }

// up to four factors [B][C_i] -> [C_i][Bp] in one launch
int lo_transpose_pad_f16_multi(const f16* const* src, f16* const* dst, const int* C, int njobs, int R, int Rp, hipStream_t st) {
  LO_REQUIRE(njobs >= 1 && njobs <= 4 && Rp % 16 == 0 && Rp >= R, "lo_transpose_pad_f16: bad job list (n=%d R=%d Rp=%d)", njobs, R, Rp);
  LoTransposeJobs J;
  memset(&J, 0, sizeof(J));
  J.n = njobs; J.R = R; J.Rp = Rp;
  int tiles = 0;
  double bytes = 0;
  for (int i = 0; i < njobs; ++i) {
    LO_REQUIRE(C[i] % 64 == 0, "lo_transpose_pad_f16: C=%d must be a multiple of 64", C[i]);
    J.j[i].src = src[i]; J.j[i].dst = dst[i]; J.j[i].C = C[i]; J.j[i].tile0 = tiles;
    tiles += (C[i] / 64) * ((Rp + 63) / 64);
    bytes += 2.0 * ((double)R + Rp) * C[i];
  }
  LoProfScope _p("lo_factor_transpose", 0, bytes, st);
  hipLaunchKernelGGL(lo_transpose_pad_f16_kernel, dim3(tiles), dim3(256), 0, st, J);
  LO_LAUNCH_CHECK("transpose_pad_f16");
  return LO_OK;
}

// ||dY^T X||_F^2 * scale^2 of up to two Linear layers into their `nslots` partial-sum slots each (every slot is written).
// fshort / flong: the two factors of a layer in their natural layouts [B][n_short], [B][n_long]; gram: [Bp][Bp] fp32 scratch per layer.
int lo_lowrank_sumsq(const LoLowrankNorm* layers, int nlayers, int B, float scale, hipStream_t st) {
  const int Bp = lo_lowrank_bp(B);
  LO_REQUIRE(nlayers >= 1 && nlayers <= 2 && Bp <= 128, "lo_lowrank_sumsq: bad shape");
  const int nblk = (Bp / 16 + 3) / 4;
  LoGramJobs J;
  memset(&J, 0, sizeof(J));
  J.B = B; J.Bp = Bp; J.scale2 = scale * scale;
  int maxslots = 0;
  double bytes = 0;
  for (int i = 0; i < nlayers; ++i) {
    const LoLowrankNorm& in = layers[i];
    LO_REQUIRE(in.n_short % 32 == 0 && in.n_long % 32 == 0, "lo_lowrank_sumsq: row lengths must be multiples of 32");
    int chunks = in.nslots / (nblk * nblk);
    LO_REQUIRE(chunks >= 1, "lo_lowrank_sumsq: %d partial slots are too few", in.nslots);
    const int steps = in.n_long / 32;
    if (chunks > steps) chunks = steps;
    const int per = (steps + chunks - 1) / chunks;
    chunks = (steps + per - 1) / per;
    LoGramLayer& l = J.l[i];
    l.fshort = in.fshort; l.flong = in.flong; l.gram = in.gram; l.partial = in.partial;
    l.n_short = in.n_short; l.n_long = in.n_long; l.chunks = chunks; l.per = per; l.nslots = in.nslots;
    maxslots = in.nslots > maxslots ? in.nslots : maxslots;
    bytes += 2.0 * B * ((double)in.n_short + in.n_long);
  }
  LoProfScope _p("lo_lowrank_sumsq", 0, bytes, st);
  hipLaunchKernelGGL(lo_gram_small_kernel, dim3((Bp / 16) * (Bp / 16), nlayers), dim3(256), 0, st, J);
  LO_LAUNCH_CHECK("gram_small");
  hipLaunchKernelGGL(lo_gram_dot_kernel, dim3(maxslots, nlayers), dim3(64), 0, st, J);
  LO_LAUNCH_CHECK("gram_dot");
  return LO_OK;
}

static int lowrank_blocks(const LoLowrankArgs& a, int cap) {
  const int total = (a.N / 64) * (a.K / 64);
  int nblk = (total + 3) / 4;
  return nblk > cap ? cap : nblk;
}
static int lowrank_launch(const LoLowrankPair& A, int nblk_total, bool mat, hipStream_t st) {
  const int KB = A.l[0].Bp / 32;
#define LO_LR(KBV)                                                                                                                  \
  do {                                                                                                                              \
    if (mat) hipLaunchKernelGGL((lo_adamw_lowrank_kernel<KBV, true>), dim3(nblk_total), dim3(256), 0, st, A);                       \
    else LO_LAUNCH_STOP((lo_adamw_lowrank_kernel<KBV, false>), dim3(nblk_total), dim3(256), 0, st, A);                              \
  } while (0)
  if (KB == 1) LO_LR(1);
  else if (KB == 2) LO_LR(2);
  else if (KB == 3) LO_LR(3);
  else if (KB == 4) LO_LR(4);
  else { lo_set_error("lo_adamw_lowrank: batch padding %d not supported", A.l[0].Bp); return LO_ERR_ARG; }
#undef LO_LR
  LO_LAUNCH_CHECK("adamw_lowrank");
  return LO_OK;
}

// AdamW of up to two factored weight matrices in ONE launch (the workgroups are shared out in proportion to the element counts)
int lo_adamw_lowrank(const LoLowrankMat* mats, int nmat, int B, float gscale, const float* norm, float lr, float beta1, float beta2, float eps,
                     float wd, int step, hipStream_t st) {
  LO_REQUIRE(nmat >= 1 && nmat <= 2, "lo_adamw_lowrank: one or two matrices");
  LoLowrankPair A;
  memset(&A, 0, sizeof(A));
  double elems[2] = {0, 0}, bytes = 0;
  for (int i = 0; i < nmat; ++i) {
    const LoLowrankMat& mt = mats[i];
    LO_REQUIRE(lo_lowrank_applies(B, mt.N, mt.K), "lo_adamw_lowrank: shape B=%d N=%d K=%d not supported", B, mt.N, mt.K);
    LoLowrankArgs& a = A.l[i];
    a.p = mt.p; a.m = mt.m; a.v = mt.v; a.cast = mt.cast; a.cast_t = mt.cast_t; a.gout = nullptr; a.xt = mt.xt; a.yt = mt.yt; a.N = mt.N; a.K = mt.K; a.Bp = lo_lowrank_bp(B);
    a.gscale = gscale; a.norm = norm; a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = wd;
    a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    elems[i] = (double)mt.N * mt.K;
    bytes += (24.0 + (mt.cast ? 2.0 : 0.0) + (mt.cast_t ? 2.0 : 0.0)) * elems[i];
  }
  // workgroups of the larger matrix.  Two are resident per CU (208 registers); a longer grid measured better on the step, where the
  // kernel runs beside the next forward: 512 / 1024 / 2048 / 4096 / 8192 -> 23 310 / 23 590 / 23 790 / 23 230 / 23 190 sprites/s
  // (three interleaved rounds on one box)
  constexpr int cap = 2048;
  int nb0 = lowrank_blocks(A.l[0], cap), nb1 = 0;
  if (nmat == 2) {
    int c1 = (int)(cap * elems[1] / elems[0] + 0.5);
    nb1 = lowrank_blocks(A.l[1], c1 < 1 ? 1 : c1);
  }
  A.nwg0 = nb0;
  LoProfScope _p("lo_adamw_lowrank", 0, bytes, st);     // HBM-bound: rated against the HBM roofline (2 Bp FLOP per element ride along)
  return lowrank_launch(A, nb0 + nb1, false, st);
}

int lo_lowrank_materialize(float* gout, const f16* xt, const f16* yt, int N, int K, int B, float gscale, hipStream_t st) {
  LO_REQUIRE(lo_lowrank_applies(B, N, K), "lo_lowrank_materialize: shape B=%d N=%d K=%d not supported", B, N, K);
  LoLowrankPair A;
  memset(&A, 0, sizeof(A));
  LoLowrankArgs& a = A.l[0];
  a.gout = gout; a.xt = xt; a.yt = yt; a.N = N; a.K = K; a.Bp = lo_lowrank_bp(B); a.gscale = gscale;
  A.nwg0 = lowrank_blocks(a, 512);
  LoProfScope _p("lo_lowrank_materialize", 0, 4.0 * (double)N * K, st);
  return lowrank_launch(A, A.nwg0, true, st);
}

int lo_lowrank_materialize_gathered(float* gout, const f16* xt, const f16* yt, size_t rank_stride_elems, int world, int N, int K, int B,
                                    float gscale, hipStream_t st) {
  LO_REQUIRE(lo_lowrank_applies(B, N, K) && world >= 1, "lo_lowrank_materialize_gathered: shape B=%d N=%d K=%d world=%d not supported", B, N, K, world);
  LoGatherMatArgs a{gout, xt, yt, rank_stride_elems, N, K, lo_lowrank_bp(B), world, gscale};
  const int total = (N / 64) * (K / 64);
  int nblk = (total + 3) / 4;
  if (nblk > 1024) nblk = 1024;
  LoProfScope _p("lo_lowrank_materialize_gathered", 2.0 * a.Bp * world * (double)N * K, 4.0 * (double)N * K, st);
  hipLaunchKernelGGL(lo_lowrank_materialize_gathered_kernel, dim3(nblk), dim3(256), 0, st, a);
  LO_LAUNCH_CHECK("lowrank_materialize_gathered");
  return LO_OK;
}
