// SelfAttention2d (lunar_generate.py:56-78): out = gamma * (V softmax(Q^T K)^T) + x on a [B, C, H, W] map.
// The reference defines this module but never instantiates it, so it is built as a standalone op with its
// own parity test, not as part of the VAE graph.  Forward, default path (MFMA):
//   lo_attn_project_qkv  : q, k (C/8 channels) and v (C channels) = 1x1 convs of x, channel-major [B][c][N], ONE launch, on the exact
//                          fp32 matrix instruction (v_mfma_f32_16x16x4_f32: a k-ordered fmaf chain, same numerics as the VALU
//                          form; q / k / v are saved in fp32 for the backward)
//   lo_attn_mfma         : one wave per 16 queries, 128 value channels per workgroup in its accumulators; the four waves of a
//                          workgroup share every 64-key block of K and V through LDS (fp16, staged while the previous block is
//                          multiplied).  Per 32 keys: S^T = K^T Q and O += V P^T on v_mfma_f32_16x16x32_f16 (fp32 accumulate), the
//                          softmax online in ONE pass over the keys (running max / sum per query, wave-wide by two shuffles
//                          across the four 16-lane groups); the score accumulators ARE the P operand of the second product
//                          (its k slots are ordered the way the accumulator rows come out, and V is loaded in that order),
//                          so nothing of size N x N touches LDS or memory.
// LO_ATTN_FP32=1 selects the first-round fp32 VALU kernels below (two passes over the keys), kept as the strict-parity form.
#include "lo_internal.h"
#include <stdlib.h>

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

// ---- MFMA forward -------------------------------------------------------------------------------------------------
// Projections: y[bi][co][n] = b[co] + sum_ci w[co][ci] x[bi][ci][n].  Workgroup = 64 output channels x 64 positions, wave = 16 x 64.
// v_mfma_f32_16x16x4_f32: lane l holds A[row l&15][k = l>>4], B[k = l>>4][col l&15]; D: col = l&15, row = 4*(l>>4) + reg.
// (lo_attn_project_qkv_kernel below: the three projections of the module in one launch.)
// Fused attention.  q, k: [B][D][N] fp32, v / x / out: [B][C][N] fp32.  Workgroup = 64 queries (4 waves x 16) x 16*NT value
// channels; grid = (N/64, C/(16*NT), B).  Keys are walked in blocks of 64 that the four waves SHARE: the block's K rows and V rows
// are fetched with 16-byte loads (fp32, key-contiguous as they lie in memory), converted to fp16 and staged in LDS -- K
// transposed to [key][d] so that a lane's A fragment is one 16-byte read, V as [channel][key] -- while the previous block is
// being multiplied (registers hold the next block's loads across the compute phase; two LDS stages, one barrier per block).
// The round-2 form let every lane fetch its own operands from global memory with 4-byte loads inside the key loop (16 + 8 NT
// dependent loads per 32 keys, no sharing between the waves): 582 us at B = 8 / C = 64 / N = 4096.
// Per wave and 32-key step:
//   S^T tiles a (keys j0..j0+15) and b (j0+16..j0+31):  D[j][i] = sum_d K[d][j] Q[d][i]      (A = K^T rows, B = Q columns)
//     -> lane (i = lane&15, g = lane>>4) holds S[i][j0 + 4g + r] (a) and S[i][j0 + 16 + 4g + r] (b), r = 0..3
//   online softmax of query i over those 32 keys: in-lane over 8 values, then across the four groups g (shuffle xor 16, 32)
//   O^T += V P^T:  D[c][i] = sum_s V[c][key(s)] P[i][key(s)] with the k slot s = 8g + t  <->  key(s) = j0 + 4g + t (t < 4) or
//     j0 + 16 + 4g + (t - 4): exactly the eight probabilities the lane already holds; V is read in that order (two 8-byte LDS
//     reads per lane and channel tile).  The accumulators' column is the query, so the running rescale is lane-local.
#define AF_KP 72      // LDS pitch of a K row [key][d] and of a V row [channel][key], in halves (64 + 8)
template <int NT, int PF>      // PF = key blocks fetched ahead (register sets): 2 where the registers allow it (NT = 4)
__global__ __launch_bounds__(256) void lo_attn_mfma_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                           const float* __restrict__ v, const float* __restrict__ x,
                                                           const float* __restrict__ gamma, float* __restrict__ out, int C, int D, int N) {
  constexpr int CV = 16 * NT, VP = CV / 16;                   // value channels of the workgroup; V staging passes (16 rows each)
  __shared__ __attribute__((aligned(16))) f16 sK[2][64 * AF_KP];
  __shared__ __attribute__((aligned(16))) f16 sV[2][CV * AF_KP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, g = lane >> 4;
  const int q0 = blockIdx.x * 64 + wave * 16, c0 = blockIdx.y * CV, bi = blockIdx.z;
  const float* qb = q + (size_t)bi * D * N;
  const float* kb = k + (size_t)bi * D * N;
  const float* vb = v + ((size_t)bi * C + c0) * N;
  const int KD = (D + 31) >> 5;                              // 32-wide k steps of the score product (D = C/8 <= 64)
  const int DR = (D + 15) >> 4;                              // K staging passes (16 rows of d each)
  f16x8 qf[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int d = ks * 32 + 8 * g + t;
      // log2(e) rides on q: the scores come out of the MFMA in the base-2 domain and every exponential below is the hardware's
      // own v_exp_f32 (2^x) without the multiply that __expf puts in front of it (softmax is invariant: the same factor sits in the
      // running maximum)
      qf[ks][t] = (ks < KD && d < D) ? (f16)(qb[(size_t)d * N + q0 + fr] * 1.44269504088896341f) : (f16)0.f;
    }
  // zero the d-padding of both K stages once (rows are [key][d], d < 32 KD read; only d < D is ever written below)
  for (int e = tid; e < 2 * 64 * AF_KP; e += 256) (&sK[0][0])[e] = (f16)0.f;
  // staging roles: thread -> (row = tid >> 4, four consecutive keys 4 (tid & 15) .. + 3) of a 16-row pass
  const int srow = tid >> 4, skey = (tid & 15) * 4;
  f32x4 kreg[PF][4], vreg[PF][VP];
  auto fetch = [&](int set, int j0) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < 4; ++p)
      if (p < DR) {
        const int d = p * 16 + srow;
        kreg[set][p] = d < D ? *reinterpret_cast<const f32x4*>(kb + (size_t)d * N + j0 + skey) : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
    for (int p = 0; p < VP; ++p) vreg[set][p] = *reinterpret_cast<const f32x4*>(vb + (size_t)(p * 16 + srow) * N + j0 + skey);
  };
  auto stage = [&](int set, int st) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < 4; ++p)
      if (p < DR) {
        const int d = p * 16 + srow;
        if (d < D) {
#pragma unroll
          for (int e = 0; e < 4; ++e) sK[st][(skey + e) * AF_KP + d] = (f16)kreg[set][p][e];
        }
      }
#pragma unroll
    for (int p = 0; p < VP; ++p)
      *reinterpret_cast<f16x4*>(&sV[st][(p * 16 + srow) * AF_KP + skey]) =
          (f16x4){(f16)vreg[set][p][0], (f16)vreg[set][p][1], (f16)vreg[set][p][2], (f16)vreg[set][p][3]};
  };
  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  const int nblk = N / 64;
  fetch(0, 0);
  if (PF == 2 && nblk > 1) fetch(1, 64);
  __syncthreads();                                           // the zero fill is complete before the first K rows are written
  stage(0, 0);
  // block `blk` is multiplied from LDS stage blk & 1 while block blk + PF travels into register set blk % PF (free: its previous
  // content, block blk, was staged before this block's barrier) and block blk + 1 -- fetched PF - 1 blocks ago -- is staged at the
  // end.  With one block ahead (PF = 1) a block's two 32-key steps (~0.4 us) did not cover an L2 round trip: at B = 8 / C = 64 /
  // N = 4096 the kernel stalled on every one of its 64 blocks.
  auto body = [&](int blk, int par) __attribute__((always_inline)) {
    const int st = par;
    __syncthreads();                                         // stage `st` is complete; everybody has finished reading stage st ^ 1
    if (blk + PF < nblk) fetch(PF == 2 ? par : 0, (blk + PF) * 64);
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const int j0 = 32 * sub;
      f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (ks < KD) {
          const f16x8 ka = *reinterpret_cast<const f16x8*>(&sK[st][(j0 + fr) * AF_KP + ks * 32 + 8 * g]);
          const f16x8 kc = *reinterpret_cast<const f16x8*>(&sK[st][(j0 + 16 + fr) * AF_KP + ks * 32 + 8 * g]);
          sa = __builtin_amdgcn_mfma_f32_16x16x32_f16(ka, qf[ks], sa, 0, 0, 0);
          sb = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc, qf[ks], sb, 0, 0, 0);
        }
      }
      float mx = fmaxf(fmaxf(fmaxf(sa[0], sa[1]), fmaxf(sa[2], sa[3])), fmaxf(fmaxf(sb[0], sb[1]), fmaxf(sb[2], sb[3])));
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // 0 on the first step (m_run = -inf)
      float p[8], ls = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) { p[r] = __builtin_amdgcn_exp2f(sa[r] - m_new); p[4 + r] = __builtin_amdgcn_exp2f(sb[r] - m_new); }
#pragma unroll
      for (int r = 0; r < 8; ++r) ls += p[r];
      ls += __shfl_xor(ls, 16, 64);
      ls += __shfl_xor(ls, 32, 64);
      l_run = l_run * alpha + ls;
      m_run = m_new;
      f16x8 pf;
#pragma unroll
      for (int r = 0; r < 8; ++r) pf[r] = (f16)p[r];
      // the running maximum settles after the first blocks: the rescale of the NT accumulator tiles is skipped (wave-uniformly)
      // whenever no query of the wave has moved its maximum -- alpha is exactly 1 then
      if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0ull) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = acc[nt] * alpha;
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const f16* vp = &sV[st][(16 * nt + fr) * AF_KP + j0 + 4 * g];
        const f16x4 va = *reinterpret_cast<const f16x4*>(vp), vc = *reinterpret_cast<const f16x4*>(vp + 16);
        const f16x8 vf = {va[0], va[1], va[2], va[3], vc[0], vc[1], vc[2], vc[3]};
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, acc[nt], 0, 0, 0);
      }
    }
    if (blk + 1 < nblk) stage(PF == 2 ? par ^ 1 : 0, st ^ 1);   // stage st ^ 1 was last read in block blk - 1: free since this block's barrier
  };
  for (int blk = 0; blk < nblk; blk += 2) {
    body(blk, 0);
    if (blk + 1 < nblk) body(blk + 1, 1);
  }
  const float inv = 1.f / l_run, gm = gamma[0];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const size_t o = ((size_t)bi * C + c0 + 16 * nt + 4 * g + r) * N + q0 + fr;
      out[o] = gm * (acc[nt][r] * inv) + x[o];
    }
}

// q, k and v projections in ONE launch: blockIdx.y walks the 64-row tiles of [Wq; Wk; Wv].
// y[bi][co][n] = b[co] + sum_ci w[co][ci] x[bi][ci][n] on v_mfma_f32_16x16x32_f16 with BOTH operands split into fp16 halves,
// a = hi + lo (hi = fp16(a), lo = fp16(a - hi): 22 significant bits), three products per tile and k step -- hi hi, hi lo, lo hi; the
// dropped lo lo term is 2^-22 of the product -- accumulated in fp32: the accuracy of an fp32 product chain (q / k / v are saved in fp32
// for the backward) at 16/3 of the fp32 matrix instruction's rate.  The first form ran on v_mfma_f32_16x16x4_f32 with 4-byte operand
// loads in the k loop: 67 us of the module's 79 at B = 64 / C = 512 / N = 64 (rocprofv3).
// Workgroup = 64 output rows x 64 positions; per 32-channel k step the x tile [32][64] is fetched with 16-byte loads, split, and
// staged TRANSPOSED in LDS ([n][ci], hi and lo planes) so that a lane's B fragment -- eight consecutive channels of one position --
// is one 16-byte read shared by the four waves; the A fragments (eight consecutive channels of one weight row) come straight from
// global memory, two 16-byte loads.  Next step's loads are issued before this step's products; two LDS stages.
#define PJ_P 40       // LDS pitch of a staged x row [n][32 ci] in halves (32 + 8)
__device__ __forceinline__ void lo_split8(const f32x4 a, const f32x4 b, f16x8& hi, f16x8& lo) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    hi[e] = (f16)a[e]; lo[e] = (f16)(a[e] - (float)hi[e]);
    hi[4 + e] = (f16)b[e]; lo[4 + e] = (f16)(b[e] - (float)hi[4 + e]);
  }
}
__global__ __launch_bounds__(256) void lo_attn_project_qkv_kernel(const float* __restrict__ x, const float* __restrict__ wq,
                                                                  const float* __restrict__ bq, const float* __restrict__ wk,
                                                                  const float* __restrict__ bk, const float* __restrict__ wv,
                                                                  const float* __restrict__ bv, float* __restrict__ q, float* __restrict__ k,
                                                                  float* __restrict__ v, int C, int D, int N) {
  __shared__ __attribute__((aligned(16))) f16 sX[2][2][64 * PJ_P];      // [stage][hi / lo][n][ci]
  const int tq = (D + 63) / 64;                              // 64-row tiles of q (and of k)
  const int ty = blockIdx.y;
  const float* w = ty < tq ? wq : (ty < 2 * tq ? wk : wv);
  const float* b = ty < tq ? bq : (ty < 2 * tq ? bk : bv);
  float* y = ty < tq ? q : (ty < 2 * tq ? k : v);
  const int Co = ty < 2 * tq ? D : C;
  const int tile = ty < tq ? ty : (ty < 2 * tq ? ty - tq : ty - 2 * tq);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 64, co0 = tile * 64 + wave * 16, bi = blockIdx.z;
  const bool wave_on = co0 < Co;                             // wave-uniform; idle waves still help staging
  const int row = co0 + fr;
  const bool valid = wave_on && row < Co;
  const float* wr = w + (size_t)(valid ? row : 0) * C + 8 * g;
  // staging role: channel ci = tid & 31, positions 8 (tid >> 5) .. + 7.  Consecutive lanes take consecutive channels so that the
  // transposing 2-byte LDS writes of a wave land on 32 different banks (with the positions along the lanes the eight rows a wave
  // writes are 160 dwords apart: two banks for 64 lanes -- half of the first form's 36 us at B = 64 / C = 512 / N = 64)
  const int sci = tid & 31, sn = (tid >> 5) * 8;
  const float* xs = x + ((size_t)bi * C + sci) * N + n0 + sn;
  f32x4 xr[2], wa, wb;
  auto fetch = [&](int ci0) {
    xr[0] = *reinterpret_cast<const f32x4*>(xs + (size_t)ci0 * N);
    xr[1] = *reinterpret_cast<const f32x4*>(xs + (size_t)ci0 * N + 4);
    wa = valid ? *reinterpret_cast<const f32x4*>(wr + ci0) : (f32x4){0.f, 0.f, 0.f, 0.f};
    wb = valid ? *reinterpret_cast<const f32x4*>(wr + ci0 + 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  auto stage = [&](int st) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xv = e < 4 ? xr[0][e] : xr[1][e - 4];
      const f16 h = (f16)xv;
      sX[st][0][(sn + e) * PJ_P + sci] = h;
      sX[st][1][(sn + e) * PJ_P + sci] = (f16)(xv - (float)h);
    }
  };
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  fetch(0);
  f16x8 whi, wlo;
  lo_split8(wa, wb, whi, wlo);
  stage(0);
  const int nk = C / 32;
  for (int ks = 0; ks < nk; ++ks) {
    const int st = ks & 1;
    __syncthreads();                                         // stage st is complete; stage st ^ 1 has been read by everybody
    f16x8 ahi = whi, alo = wlo;
    if (ks + 1 < nk) fetch((ks + 1) * 32);
    if (wave_on) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f16x8 bh = *reinterpret_cast<const f16x8*>(&sX[st][0][(16 * t + fr) * PJ_P + 8 * g]);
        const f16x8 bl = *reinterpret_cast<const f16x8*>(&sX[st][1][(16 * t + fr) * PJ_P + 8 * g]);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, bh, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, bl, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, bh, acc[t], 0, 0, 0);
      }
    }
    if (ks + 1 < nk) {
      lo_split8(wa, wb, whi, wlo);
      stage(st ^ 1);
    }
  }
  if (!wave_on) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int co = co0 + 4 * g + r;
    if (co < Co) {
      const float bvv = b[co];
#pragma unroll
      for (int t = 0; t < 4; ++t) y[((size_t)bi * Co + co) * N + n0 + 16 * t + fr] = acc[t][r] + bvv;
    }
  }
}

int lo_selfattn2d_fwd(const float* x, const float* wq, const float* bq, const float* wk, const float* bk, const float* wv,
                      const float* bv, const float* gamma, float* q, float* k, float* v, float* out, int B, int C, int N,
                      hipStream_t st) {
  LO_REQUIRE(C % 64 == 0 && N % 64 == 0, "lo_selfattn2d_fwd: C=%d and H*W=%d must be multiples of 64", C, N);
  const int D = C / 8;
  LO_REQUIRE(D <= 64, "lo_selfattn2d_fwd: C/8 = %d > 64 is not supported", D);
  static const bool fp32_path = getenv("LO_ATTN_FP32") && atoi(getenv("LO_ATTN_FP32")) != 0;
  if (!fp32_path) {
    // C is a multiple of 64, so the k loop of the projection runs in whole 16-channel iterations
    hipLaunchKernelGGL(lo_attn_project_qkv_kernel, dim3(N / 64, 2 * ((D + 63) / 64) + C / 64, B), dim3(256), 0, st, x, wq, bq, wk, bk, wv, bv, q, k, v,
                       C, D, N);
    LO_LAUNCH_CHECK("attn_project_qkv");
    // value channels per workgroup: 128 (64 for a 64-channel map): the scores are recomputed per channel group (D = C/8 against 128
    // channels of P V: a third of the group's work) and in exchange the grid reaches every CU at the module's shapes -- B = 64 / C =
    // 512 / N = 64: 256 workgroups; B = 8 / C = 64 / N = 4096: 512
    const int CV = C % 128 == 0 ? 128 : 64;
    LoProfScope _p("lo_attn_mfma", 2.0 * B * N * (double)N * (D * (double)(C / CV) + C), 4.0 * B * N * (2.0 * D + 3.0 * C), st);
    const dim3 grid(N / 64, C / CV, B);
    if (CV == 128) hipLaunchKernelGGL((lo_attn_mfma_kernel<8, 1>), grid, dim3(256), 0, st, q, k, v, x, gamma, out, C, D, N);
    else hipLaunchKernelGGL((lo_attn_mfma_kernel<4, 2>), grid, dim3(256), 0, st, q, k, v, x, gamma, out, C, D, N);
    LO_LAUNCH_CHECK("attn_mfma");
    return LO_OK;
  }
  dim3 blk(64), gq((N + 63) / 64, D, B), gv((N + 63) / 64, C, B);
  hipLaunchKernelGGL(lo_attn_project_kernel, gq, blk, 0, st, x, wq, bq, q, C, D, N);
  hipLaunchKernelGGL(lo_attn_project_kernel, gq, blk, 0, st, x, wk, bk, k, C, D, N);
  hipLaunchKernelGGL(lo_attn_project_kernel, gv, blk, 0, st, x, wv, bv, v, C, C, N);
  LO_LAUNCH_CHECK("attn_project");
  size_t lds = (size_t)(2 * D * 64 + 64 * 65 + 64 * 65 + 256) * sizeof(float);
  static bool attr = false;
  if (!attr) { LO_HIP(hipFuncSetAttribute((const void*)lo_attn_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096)); attr = true; }
  LoProfScope _p("lo_attn_fused", 2.0 * B * N * (double)N * (2.0 * D + C), 4.0 * B * N * (2.0 * D + 3.0 * C), st);
  hipLaunchKernelGGL(lo_attn_fused_kernel, dim3(N / 64, C / 64, B), dim3(256), lds, st, q, k, v, x, gamma, out, C, D, N);
  LO_LAUNCH_CHECK("attn_fused");
  return LO_OK;
}

// =============================================================================================
// backward (fp32; off the VAE's hot path, built so that the module is trainable)
//   y = gamma * O + x,  O[c][i] = sum_j V[c][j] att[i][j],  att = softmax_j(S),  S[i][j] = sum_d Q[d][i] K[d][j]
//   dO = gamma dy;  t_i = sum_c dy[c][i] O[c][i]  (dgamma = sum t_i, delta_i = gamma t_i)
//   dP[i][j] = sum_c dO[c][i] V[c][j];  dS = att (dP - delta_i)
//   dQ[d][i] = sum_j dS[i][j] K[d][j];  dK[d][j] = sum_i dS[i][j] Q[d][i];  dV[c][j] = sum_i dO[c][i] att[i][j]
// dP is a sum over channels, and dS is linear in it: every 64-channel group contributes partial dQ / dK (group 0 also
// carries the -delta term), summed afterwards in a fixed order.  Nothing of size N x N is written to memory.
//   lo_attn_bwd_stats : row max / 1/sum of the softmax and t_part[b][cg][i]      grid (N/64, C/64, B)
//   lo_attn_bwd_dq    : dQ_part[b][cg][d][i]                                      grid (N/64 query blocks, C/64, B)
//   lo_attn_bwd_dkv   : dK_part[b][cg][d][j], dV[b][c][j]                         grid (N/64 key blocks,   C/64, B)
//   lo_attn_bwd_proj_* : 1x1-conv weight / bias gradients and dx
// =============================================================================================
__global__ __launch_bounds__(256) void lo_attn_bwd_stats_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                const float* __restrict__ v, const float* __restrict__ dy,
                                                                float* __restrict__ ml, float* __restrict__ t_part,
                                                                int C, int D, int N) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* s_q = sm;
  float* s_k = s_q + D * AT_Q;
  float* s_p = s_k + D * AT_K;
  float* s_v = s_p + AT_Q * 65;
  float* s_red = s_v + AT_C * 65;
  const int tid = threadIdx.x, i = tid & 63, part = tid >> 6;
  const int q0 = blockIdx.x * AT_Q, c0 = blockIdx.y * AT_C, bi = blockIdx.z;
  const float* qb = q + (size_t)bi * D * N;
  const float* kb = k + (size_t)bi * D * N;
  const float* vb = v + (size_t)bi * C * N;
  for (int e = tid; e < D * AT_Q; e += 256) s_q[e] = qb[(size_t)(e / AT_Q) * N + q0 + (e % AT_Q)];
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
  s_red[part * 64 + i] = m;
  __syncthreads();
  const float mt = fmaxf(fmaxf(s_red[i], s_red[64 + i]), fmaxf(s_red[128 + i], s_red[192 + i]));
  __syncthreads();
  s_red[part * 64 + i] = l * __expf(m - mt);
  __syncthreads();
  const float inv_l = 1.f / (s_red[i] + s_red[64 + i] + s_red[128 + i] + s_red[192 + i]);
  if (blockIdx.y == 0 && part == 0) {
    ml[((size_t)bi * N + q0 + i) * 2] = mt;
    ml[((size_t)bi * N + q0 + i) * 2 + 1] = inv_l;
  }
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
  float t = 0.f;
#pragma unroll
  for (int c = 0; c < 16; ++c) t += acc[c] * dy[((size_t)bi * C + c0 + part * 16 + c) * N + q0 + i];
  __syncthreads();
  s_red[part * 64 + i] = t;
  __syncthreads();
  if (part == 0) t_part[((size_t)bi * gridDim.y + blockIdx.y) * N + q0 + i] = s_red[i] + s_red[64 + i] + s_red[128 + i] + s_red[192 + i];
}

// MODE 0: workgroup = 64 queries (i), loops key blocks, writes dQ_part.  MODE 1: workgroup = 64 keys (j), loops query
// blocks, writes dK_part and dV.  thread = (row r = tid & 63 of the block the workgroup owns, part = tid >> 6)
template <int MODE>
__global__ __launch_bounds__(256) void lo_attn_bwd_main_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                               const float* __restrict__ v, const float* __restrict__ dy,
                                                               const float* __restrict__ gamma, const float* __restrict__ ml,
                                                               const float* __restrict__ t_part, float* __restrict__ dqk_part,
                                                               float* __restrict__ dv, int C, int D, int N) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* s_own = sm;                       // [D][64]  q (MODE 0) or k (MODE 1) of the owned block
  float* s_oth = s_own + D * 64;           // [D][64]  the other operand of the current pair
  float* s_ds = s_oth + D * 64;            // [64 own][65]  dS of the current pair
  float* s_att = s_ds + 64 * 65;           // [64 own][65]  att of the current pair (MODE 1: for dV)
  float* s_a = s_att + 64 * 65;            // [64 c][65]    dO chunk: columns = queries of the pair
  float* s_b = s_a + 64 * 65;              // [64 c][65]    V chunk: columns = keys of the pair
  float* s_st = s_b + 64 * 65;             // [3][64]       m, 1/l, delta of the queries of the pair
  const int tid = threadIdx.x, r = tid & 63, part = tid >> 6;
  const int o0 = blockIdx.x * 64, c0 = blockIdx.y * 64, bi = blockIdx.z, ncg = gridDim.y;
  const float gm = gamma[0];
  const float* qb = q + (size_t)bi * D * N;
  const float* kb = k + (size_t)bi * D * N;
  const float* vb = v + (size_t)bi * C * N;
  const float* dyb = dy + (size_t)bi * C * N;
  const float* own = MODE == 0 ? qb : kb;
  const float* oth = MODE == 0 ? kb : qb;
  for (int e = tid; e < D * 64; e += 256) s_own[e] = own[(size_t)(e / 64) * N + o0 + (e % 64)];
  float accq[16], accv[16];                // dQ / dK rows d = part*16 + dd (D <= 64); dV channels part*16 + c
#pragma unroll
  for (int c = 0; c < 16; ++c) { accq[c] = 0.f; accv[c] = 0.f; }
  for (int p0 = 0; p0 < N; p0 += 64) {
    const int i0 = MODE == 0 ? o0 : p0, j0 = MODE == 0 ? p0 : o0;   // queries i0.., keys j0.. of this pair
    __syncthreads();
    for (int e = tid; e < D * 64; e += 256) s_oth[e] = oth[(size_t)(e / 64) * N + p0 + (e % 64)];
    for (int e = tid; e < 64 * 64; e += 256) {
      const int c = e / 64, x = e % 64;
      s_a[c * 65 + x] = gm * dyb[(size_t)(c0 + c) * N + i0 + x];
      s_b[c * 65 + x] = vb[(size_t)(c0 + c) * N + j0 + x];
    }
    if (tid < 64) {
      s_st[tid] = ml[((size_t)bi * N + i0 + tid) * 2];
      s_st[64 + tid] = ml[((size_t)bi * N + i0 + tid) * 2 + 1];
      float t = 0.f;
      for (int g = 0; g < ncg; ++g) t += t_part[((size_t)bi * ncg + g) * N + i0 + tid];
      s_st[128 + tid] = gm * t;
    }
    __syncthreads();
    // att and dS for (own row r, 16 columns xx of the other block)
    const float* qv = MODE == 0 ? s_own : s_oth;
    const float* kv = MODE == 0 ? s_oth : s_own;
    for (int xx = part * 16; xx < part * 16 + 16; ++xx) {
      const int qi = MODE == 0 ? r : xx, kj = MODE == 0 ? xx : r;
      float s = 0.f;
      for (int d = 0; d < D; ++d) s += qv[d * 64 + qi] * kv[d * 64 + kj];
      const float att = __expf(s - s_st[qi]) * s_st[64 + qi];
      float dp = 0.f;
      for (int c = 0; c < 64; ++c) dp += s_a[c * 65 + qi] * s_b[c * 65 + kj];
      s_ds[r * 65 + xx] = att * (dp - (blockIdx.y == 0 ? s_st[128 + qi] : 0.f));
      if (MODE == 1) s_att[r * 65 + xx] = att;
    }
    __syncthreads();
    // dQ[d][i=r] += sum_x dS[r][x] K[d][x]  (MODE 0)     dK[d][j=r] += sum_x dS[r][x] Q[d][x]  (MODE 1)
    for (int xx = 0; xx < 64; ++xx) {
      const float ds = s_ds[r * 65 + xx];
#pragma unroll
      for (int dd = 0; dd < 16; ++dd) {
        const int d = part * 16 + dd;
        if (d < D) accq[dd] += ds * s_oth[d * 64 + xx];
      }
      if (MODE == 1) {
        // dV[c][j=r] += dO[c][i=xx] att[i=xx][j=r]
        const float at = s_att[r * 65 + xx];
#pragma unroll
        for (int c = 0; c < 16; ++c) accv[c] += s_a[(part * 16 + c) * 65 + xx] * at;
      }
    }
  }
#pragma unroll
  for (int dd = 0; dd < 16; ++dd) {
    const int d = part * 16 + dd;
    if (d < D) dqk_part[(((size_t)bi * ncg + blockIdx.y) * D + d) * N + o0 + r] = accq[dd];
  }
  if (MODE == 1) {
#pragma unroll
    for (int c = 0; c < 16; ++c) dv[((size_t)bi * C + c0 + part * 16 + c) * N + o0 + r] = accv[c];
  }
}

// g[b][d][n] = sum_cg part[b][cg][d][n]  (fixed order)
__global__ void lo_attn_sum_parts_kernel(const float* __restrict__ part, float* __restrict__ g, int ncg, int DN, int total) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int b = i / DN, r = i - b * DN;
  float t = 0.f;
  for (int c = 0; c < ncg; ++c) t += part[((size_t)b * ncg + c) * DN + r];
  g[i] = t;
}

// 1x1-conv parameter gradients: dw[co][ci] = sum_{b,n} g[b][co][n] x[b][ci][n], db[co] = sum_{b,n} g[b][co][n]
// block = 16 (co) x 16 (ci) threads, tiles of 64 positions through LDS
__global__ __launch_bounds__(256) void lo_attn_proj_wgrad_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                                 float* __restrict__ dw, float* __restrict__ db, int B, int Co,
                                                                 int C, int N) {
  __shared__ float s_g[16][65], s_x[16][65];
  const int tid = threadIdx.x, tci = tid & 15, tco = tid >> 4;
  const int co0 = blockIdx.x * 16, ci0 = blockIdx.y * 16;
  float acc = 0.f, accb = 0.f;
  for (int b = 0; b < B; ++b)
    for (int n0 = 0; n0 < N; n0 += 64) {
      __syncthreads();
      for (int e = tid; e < 16 * 64; e += 256) {
        const int rr = e / 64, nn = e % 64;
        s_g[rr][nn] = co0 + rr < Co ? g[((size_t)b * Co + co0 + rr) * N + n0 + nn] : 0.f;
        s_x[rr][nn] = x[((size_t)b * C + ci0 + rr) * N + n0 + nn];
      }
      __syncthreads();
      for (int nn = 0; nn < 64; ++nn) {
        acc += s_g[tco][nn] * s_x[tci][nn];
        if (tci == 0) accb += s_g[tco][nn];
      }
    }
  if (co0 + tco < Co) {
    dw[(size_t)(co0 + tco) * C + ci0 + tci] = acc;
    if (tci == 0 && blockIdx.y == 0) db[co0 + tco] = accb;
  }
}

// dx[b][c][n] = dy + sum_d wq[d][c] dq[b][d][n] + sum_d wk[d][c] dk[b][d][n] + sum_c' wv[c'][c] dv[b][c'][n]
__global__ void lo_attn_proj_dx_kernel(const float* __restrict__ dy, const float* __restrict__ wq, const float* __restrict__ wk,
                                       const float* __restrict__ wv, const float* __restrict__ dq, const float* __restrict__ dk,
                                       const float* __restrict__ dvv, float* __restrict__ dx, int C, int D, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  if (n >= N) return;
  float acc = dy[((size_t)b * C + c) * N + n];
  for (int d = 0; d < D; ++d) {
    acc += wq[(size_t)d * C + c] * dq[((size_t)b * D + d) * N + n];
    acc += wk[(size_t)d * C + c] * dk[((size_t)b * D + d) * N + n];
  }
  for (int cp = 0; cp < C; ++cp) acc += wv[(size_t)cp * C + c] * dvv[((size_t)b * C + cp) * N + n];
  dx[((size_t)b * C + c) * N + n] = acc;
}

// dgamma = sum of t_part (single workgroup, fixed order)
__global__ __launch_bounds__(256) void lo_attn_dgamma_kernel(const float* __restrict__ t_part, float* __restrict__ dgamma, int total) {
  __shared__ double red[256];
  double t = 0.0;
  for (int i = threadIdx.x; i < total; i += 256) t += (double)t_part[i];
  red[threadIdx.x] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < 256; ++i) s += red[i];
    dgamma[0] = (float)s;
  }
}

// scratch (floats): ml 2BN | t_part B*ncg*N | dq_part B*ncg*D*N | dk_part B*ncg*D*N | dq B*D*N | dk B*D*N | dv B*C*N
size_t lo_selfattn2d_bwd_scratch(int B, int C, int N) {
  const size_t ncg = C / 64, D = C / 8;
  return (size_t)2 * B * N + (size_t)B * ncg * N + 2 * (size_t)B * ncg * D * N + 2 * (size_t)B * D * N + (size_t)B * C * N;
}

int lo_selfattn2d_bwd(const float* x, const float* wq, const float* wk, const float* wv, const float* gamma, const float* q,
                      const float* k, const float* v, const float* dy, float* scratch, float* dx, float* dwq, float* dbq,
                      float* dwk, float* dbk, float* dwv, float* dbv, float* dgamma, int B, int C, int N, hipStream_t st) {
  LO_REQUIRE(C % 64 == 0 && N % 64 == 0, "lo_selfattn2d_bwd: C=%d and H*W=%d must be multiples of 64", C, N);
  const int D = C / 8, ncg = C / 64;
  LO_REQUIRE(D <= 64, "lo_selfattn2d_bwd: C/8 = %d > 64 is not supported", D);
  float* ml = scratch;
  float* t_part = ml + (size_t)2 * B * N;
  float* dq_part = t_part + (size_t)B * ncg * N;
  float* dk_part = dq_part + (size_t)B * ncg * D * N;
  float* dq = dk_part + (size_t)B * ncg * D * N;
  float* dk = dq + (size_t)B * D * N;
  float* dvv = dk + (size_t)B * D * N;
  const dim3 grid(N / 64, ncg, B);
  {
    size_t lds = (size_t)(2 * D * 64 + 64 * 65 + 64 * 65 + 256) * sizeof(float);
    static bool attr1 = false;
    if (!attr1) { LO_HIP(hipFuncSetAttribute((const void*)lo_attn_bwd_stats_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096)); attr1 = true; }
    hipLaunchKernelGGL(lo_attn_bwd_stats_kernel, grid, dim3(256), lds, st, q, k, v, dy, ml, t_part, C, D, N);
    LO_LAUNCH_CHECK("attn_bwd_stats");
  }
  {
    size_t lds = (size_t)(2 * D * 64 + 4 * 64 * 65 + 192) * sizeof(float);
    static bool attr2 = false;
    if (!attr2) {
      LO_HIP(hipFuncSetAttribute((const void*)lo_attn_bwd_main_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096));
      LO_HIP(hipFuncSetAttribute((const void*)lo_attn_bwd_main_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096));
      attr2 = true;
    }
    hipLaunchKernelGGL((lo_attn_bwd_main_kernel<0>), grid, dim3(256), lds, st, q, k, v, dy, gamma, ml, t_part, dq_part, nullptr, C, D, N);
    hipLaunchKernelGGL((lo_attn_bwd_main_kernel<1>), grid, dim3(256), lds, st, q, k, v, dy, gamma, ml, t_part, dk_part, dvv, C, D, N);
    LO_LAUNCH_CHECK("attn_bwd_main");
  }
  const int tot = B * D * N;
  hipLaunchKernelGGL(lo_attn_sum_parts_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, dq_part, dq, ncg, D * N, tot);
  hipLaunchKernelGGL(lo_attn_sum_parts_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, dk_part, dk, ncg, D * N, tot);
  hipLaunchKernelGGL(lo_attn_proj_wgrad_kernel, dim3((D + 15) / 16, C / 16), dim3(256), 0, st, dq, x, dwq, dbq, B, D, C, N);
  hipLaunchKernelGGL(lo_attn_proj_wgrad_kernel, dim3((D + 15) / 16, C / 16), dim3(256), 0, st, dk, x, dwk, dbk, B, D, C, N);
  hipLaunchKernelGGL(lo_attn_proj_wgrad_kernel, dim3(C / 16, C / 16), dim3(256), 0, st, dvv, x, dwv, dbv, B, C, C, N);
  hipLaunchKernelGGL(lo_attn_proj_dx_kernel, dim3(N / 64, C, B), dim3(64), 0, st, dy, wq, wk, wv, dq, dk, dvv, dx, C, D, N);
  hipLaunchKernelGGL(lo_attn_dgamma_kernel, dim3(1), dim3(256), 0, st, t_part, dgamma, B * ncg * N);
  LO_LAUNCH_CHECK("attn_bwd_proj");
  return LO_OK;
}
