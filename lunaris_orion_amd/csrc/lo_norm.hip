// GroupNorm(8) + Mish kernels (forward apply, backward reduce / apply, parameter-gradient finalize) on NHWC fp16
// activations with fp32 statistics.  HBM-bound streaming kernels: 16-byte vector loads/stores, one thread owns a
// fixed 8-channel chunk and strides over pixel rows; all cross-thread reductions are in a fixed order
// (bitwise run-to-run reproducible, no float atomics).
//
// Reference ops replaced: nn.GroupNorm(8, C) + nn.Mish (lunar_generate.py:37-38,42-43,96-97,103-104,110-111,
// 117-118,170-171,176-177,182-183,188-189), the ResBlock tail mish(out + identity) (:49-53) and the decoder
// skip additions (:212-222).
#include "lo_internal.h"
#include <stdlib.h>

#define GN_EPS 1e-5f

enum { GN_MODE_PLAIN = 0, GN_MODE_SKIP = 1, GN_MODE_RES = 2 };

int lo_gn_nchunk(int HW, int C) {
  constexpr long chunk_elems = 16384;     // smaller and larger chunks measured in round 1: the optimum except for the 128 x 128 layer
  // the 128x128-resolution layer (32 channels): twice the chunk halves the partial rows its backward apply pass re-reads
  // per workgroup (measured 50 -> 41 us; neutral for its other two passes, worse for every smaller layer)
  const long ce = (long)HW * C >= 524288 ? 2 * chunk_elems : chunk_elems;
  long e = (long)HW * C / ce;
  if (e < 1) e = 1;
  if (e > 256) e = 256;
  while (HW % e) --e;
  return (int)e;
}

struct GnFwdArgs {
  const f16* v;          // conv output (raw), [B][HW][C]
  const float* partial;  // [B][MT][8][2] (sum, sumsq) from the conv epilogue
  const float* gamma;
  const float* beta;
  const f16* other;      // skip (mode 1) or identity (mode 2)
  f16* y;
  float* stats;          // [B][8][2] mean, rstd (saved for backward)
  int HW, C, MT, nchunk, mode;
  uint8_t* y8;           // optional e4m3 copy of y * LO_F8_ACT_SCALE (operand of an fp8 conv), same layout
};

template <int MODE>   // the mode as a template parameter: the PLAIN launches do not carry the second operand's registers
__global__ __launch_bounds__(256) void lo_gn_fwd_kernel(GnFwdArgs a) {
  __shared__ float s_stat[16];
  const int tid = threadIdx.x, n = blockIdx.y, chunk = blockIdx.x;
  const int C = a.C, G = C >> 3, CC = C >> 3;   // CC = 16-byte chunks per pixel row (8 channels each)
  // issue the affine-parameter loads first: their latency overlaps the statistics reduction below
  const int cc = tid % CC, slot = tid / CC, nslot = 256 / CC;
  const int c0 = cc * 8;
  const f32x4 gm0 = *reinterpret_cast<const f32x4*>(a.gamma + c0), gm1 = *reinterpret_cast<const f32x4*>(a.gamma + c0 + 4);
  const f32x4 bt0 = *reinterpret_cast<const f32x4*>(a.beta + c0), bt1 = *reinterpret_cast<const f32x4*>(a.beta + c0 + 4);
  if (a.partial) {
    lo_gn_group_stats(a.partial, a.MT, n, 1.0f / ((float)a.HW * (float)G), s_stat, tid);
    __syncthreads();
    if (chunk == 0 && tid < 16 && a.stats) a.stats[n * 16 + tid] = s_stat[tid];
  } else {
    if (tid < 16) s_stat[tid] = a.stats[n * 16 + tid];
    __syncthreads();
  }
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int grp = (c0 + j) / G;
    float mean = s_stat[grp * 2], rstd = s_stat[grp * 2 + 1];
    float gm = j < 4 ? gm0[j & 3] : gm1[j & 3];
    float bt = j < 4 ? bt0[j & 3] : bt1[j & 3];
    lo_gn_scale_shift(gm, bt, mean, rstd, sc[j], sh[j]);
  }
  const int rows = a.HW / a.nchunk;
  const size_t base = ((size_t)n * a.HW + (size_t)chunk * rows) * C + c0;
  constexpr int U = 4;   // rows in flight per thread (memory-level parallelism)
  for (int r = slot; r < rows; r += U * nslot) {
    f16x8 h[U], o[MODE != GN_MODE_PLAIN ? U : 1];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int rr = r + u * nslot;
      if (rr < rows) {
        size_t off = base + (size_t)rr * C;
        h[u] = *reinterpret_cast<const f16x8*>(a.v + off);
        if constexpr (MODE != GN_MODE_PLAIN) o[u] = *reinterpret_cast<const f16x8*>(a.other + off);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int rr = r + u * nslot;
      if (rr < rows) {
        float yf[8];
        const f16x8 y = lo_gn_apply8(h[u], sc, sh, MODE, MODE != GN_MODE_PLAIN ? o[u] : h[u], a.y8 ? yf : nullptr);   // the arithmetic the fused conv epilogues share
        *reinterpret_cast<f16x8*>(a.y + base + (size_t)rr * C) = y;
        if (a.y8) {
          constexpr float S8 = LO_F8_ACT_SCALE;
          *reinterpret_cast<u32x2*>(a.y8 + base + (size_t)rr * C) = (u32x2){lo_pack4_fp8(yf[0] * S8, yf[1] * S8, yf[2] * S8, yf[3] * S8),
                                                                          lo_pack4_fp8(yf[4] * S8, yf[5] * S8, yf[6] * S8, yf[7] * S8)};
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// backward
//   y = f(u), u = xhat*gamma + beta, xhat = (v - mean)*rstd
//   mode PLAIN/SKIP : du = dy * mish'(u)            (the skip branch receives dy unchanged)
//   mode RES        : s = mish(u) + id ; ds = dy * mish'(s) (written out: gradient of the identity path);
//                     du = ds * mish'(u)
//   P1[n][chunk][c] = (sum du, sum du*xhat)
// ---------------------------------------------------------------------------------------------
struct GnBwdArgs {
  const f16* dy;
  const f16* v;
  const f16* other;     // identity (mode RES)
  const float* stats;   // [B][8][2]
  const float* gamma;
  const float* beta;
  f16* ds;              // mode RES: gradient wrt (mish(u) + id), fp16
  f16* dv;              // apply: gradient wrt conv output
  float* P1;            // [B][nchunk][C][2]
  float* P2;            // [B][nchunk][C]   (sum dv, for the conv bias gradient)
  int HW, C, nchunk, mode;
  int np1;              // rows of P1 per sample (nchunk, or the tile count of the fused data-gradient epilogue)
};

// two elements at a time: xhat = hv*rstd + nmr (nmr = -mean*rstd), u = hv*sc + sh.  PLAIN / SKIP: lo_gn_du2_plain (lo_common.h,
// shared with the data-gradient epilogues that fuse the apply pass); contraction off, fused multiply-adds written out
__device__ __forceinline__ void gn_du2(int mode, lo_f2 hv, lo_f2 dyv, lo_f2 ov, lo_f2 sc, lo_f2 sh, lo_f2 rstd, lo_f2 nmr,
                                       lo_f2& du, lo_f2& xhat, lo_f2& dsv) {
#pragma clang fp contract(off)
  if (mode == GN_MODE_RES) {
    xhat = lo_fma2(hv, rstd, nmr);
    const lo_f2 u = lo_fma2(hv, sc, sh);
    lo_f2 w, p, r, tau;
    lo_mish_parts2(u, w, p, r, tau);
    const lo_f2 gu = lo_fma2(u * 4.0f, (w * p) * (r * r), tau);
    dsv = dyv * lo_mish_grad2(lo_fma2(u, tau, ov));
    du = dsv * gu;
  } else {
    dsv = dyv;
    lo_gn_du2_plain(hv, dyv, sc, sh, rstd, nmr, du, xhat);
  }
}

// RES: the mode is a template parameter so that the PLAIN / SKIP launches (13 of 16 per step) do not carry the registers of
// the identity operand and of the second Mish (occupancy: these kernels are latency-bound streams of 64 elements per thread)
// G4: C = 32 (4 channels per group, two groups inside a thread's 8-channel chunk); otherwise the chunk lies in ONE group and the
// per-group factors (rstd, -mean*rstd, the two correction terms) are scalars instead of 8-element arrays
template <bool RES, bool G4>
__global__ __launch_bounds__(256) void lo_gn_bwd_reduce_kernel(GnBwdArgs a) {
  const int a_mode = RES ? GN_MODE_RES : GN_MODE_PLAIN;
  constexpr int NG = G4 ? 2 : 1;
  __shared__ float s_red[256 * 16];
  const int tid = threadIdx.x, n = blockIdx.y, chunk = blockIdx.x;
  const int C = a.C, G = C >> 3, CC = C >> 3;
  const int cc = tid % CC, slot = tid / CC, nslot = 256 / CC;
  const int c0 = cc * 8;
  float sc[8], sh[8], nmr[NG], rstd[NG];
#pragma unroll
  for (int q = 0; q < NG; ++q) {
    int grp = (c0 + 4 * q) / G;
    const float mean = a.stats[n * 16 + grp * 2];
    rstd[q] = a.stats[n * 16 + grp * 2 + 1];
    nmr[q] = -mean * rstd[q];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int q = G4 ? (j >> 2) : 0;
    sc[j] = a.gamma[c0 + j] * rstd[q];
    sh[j] = __builtin_fmaf(nmr[q], a.gamma[c0 + j], a.beta[c0 + j]);     // beta - mean * gamma * rstd
  }
  lo_f2 a1[4], a2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { a1[j] = (lo_f2){0.f, 0.f}; a2[j] = (lo_f2){0.f, 0.f}; }
  const int rows = a.HW / a.nchunk;
  const size_t base = ((size_t)n * a.HW + (size_t)chunk * rows) * C + c0;
  constexpr int U = 4;
  for (int r = slot; r < rows; r += U * nslot) {
    f16x8 h[U], d[U], o[RES ? U : 1];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int rr = r + u * nslot;
      if (rr < rows) {
        size_t off = base + (size_t)rr * C;
        h[u] = *reinterpret_cast<const f16x8*>(a.v + off);
        d[u] = *reinterpret_cast<const f16x8*>(a.dy + off);
        if constexpr (RES) o[u] = *reinterpret_cast<const f16x8*>(a.other + off);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int rr = r + u * nslot;
      if (rr < rows) {
        f16x8 dso;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          lo_f2 du, xh, dsv;
          lo_f2 ov = {0.f, 0.f};
          if constexpr (RES) ov = (lo_f2){(float)o[u][j], (float)o[u][j + 1]};
          gn_du2(a_mode, (lo_f2){(float)h[u][j], (float)h[u][j + 1]}, (lo_f2){(float)d[u][j], (float)d[u][j + 1]}, ov,
                 (lo_f2){sc[j], sc[j + 1]}, (lo_f2){sh[j], sh[j + 1]}, (lo_f2){rstd[G4 ? (j >> 2) : 0], rstd[G4 ? (j >> 2) : 0]},
                 (lo_f2){nmr[G4 ? (j >> 2) : 0], nmr[G4 ? (j >> 2) : 0]}, du, xh, dsv);
          a1[j >> 1] += du;
          a2[j >> 1] += du * xh;
          dso[j] = (f16)dsv[0];
          dso[j + 1] = (f16)dsv[1];
        }
        if (RES) *reinterpret_cast<f16x8*>(a.ds + base + (size_t)rr * C) = dso;
      }
    }
  }
  // reduce over the row slots in a fixed order
#pragma unroll
  for (int j = 0; j < 8; ++j) { s_red[tid * 16 + j * 2] = a1[j >> 1][j & 1]; s_red[tid * 16 + j * 2 + 1] = a2[j >> 1][j & 1]; }
  __syncthreads();
  // C*2 outputs; thread t handles output (c = t>>1 ... ) looping
  for (int o = tid; o < C * 2; o += 256) {
    int c = o >> 1, w = o & 1;
    int ccx = c >> 3, j = c & 7;
    float tot = 0.f;
    for (int s = 0; s < nslot; ++s) tot += s_red[(s * CC + ccx) * 16 + j * 2 + w];
    a.P1[(((size_t)n * a.nchunk + chunk) * C + c) * 2 + w] = tot;
  }
}

template <bool RES, bool G4>
__global__ __launch_bounds__(256) void lo_gn_bwd_apply_kernel(GnBwdArgs a) {
  const int a_mode = RES ? GN_MODE_RES : GN_MODE_PLAIN;
  constexpr int NG = G4 ? 2 : 1;
  __shared__ float s_red[256 * 8];
  __shared__ float s_c[16];       // per group: c1, c2
  __shared__ float s_g[512 * 2];  // per channel gamma-weighted sums (scratch)
  const int tid = threadIdx.x, n = blockIdx.y, chunk = blockIdx.x;
  const int C = a.C, G = C >> 3, CC = C >> 3;
  const int cc = tid % CC, slot = tid / CC, nslot = 256 / CC;
  const int c0 = cc * 8;
  // issue the per-thread parameter / statistics loads first: their latency overlaps the partial-sum reduction below
  const f32x4 gm0 = *reinterpret_cast<const f32x4*>(a.gamma + c0), gm1 = *reinterpret_cast<const f32x4*>(a.gamma + c0 + 4);
  const f32x4 bt0 = *reinterpret_cast<const f32x4*>(a.beta + c0), bt1 = *reinterpret_cast<const f32x4*>(a.beta + c0 + 4);
  const int grp_lo = c0 / G, grp_hi = (c0 + 7) / G;
  const f32x2 st_lo = *reinterpret_cast<const f32x2*>(a.stats + n * 16 + grp_lo * 2);
  const f32x2 st_hi = *reinterpret_cast<const f32x2*>(a.stats + n * 16 + grp_hi * 2);
  // per-channel totals over chunks, then gamma-weighted group sums
  for (int c = tid; c < C; c += 256) {
    float t1 = 0.f, t2 = 0.f;
    int k = 0;
    for (; k + 8 <= a.np1; k += 8) {        // eight rows' loads in flight, added in row order (the order the fused epilogues reproduce)
      f32x2 pv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) pv[u] = *reinterpret_cast<const f32x2*>(a.P1 + (((size_t)n * a.np1 + k + u) * C + c) * 2);
#pragma unroll
      for (int u = 0; u < 8; ++u) { t1 += pv[u][0]; t2 += pv[u][1]; }
    }
    for (; k < a.np1; ++k) {
      const float* p = a.P1 + (((size_t)n * a.np1 + k) * C + c) * 2;
      t1 += p[0];
      t2 += p[1];
    }
    float gm = a.gamma[c];
    s_g[c * 2] = gm * t1;
    s_g[c * 2 + 1] = gm * t2;
  }
  __syncthreads();
  if (tid < 16) {
    int grp = tid >> 1, w = tid & 1;
    float tot = 0.f;
    for (int c = grp * G; c < (grp + 1) * G; ++c) tot += s_g[c * 2 + w];
    s_c[tid] = tot / ((float)a.HW * (float)G);
  }
  __syncthreads();
  // dv = rstd (gamma du - k1 - xhat k2) = du*ka - kb - xhat*kc
  float sc[8], sh[8], nmr[NG], rstd[NG], kb[NG], kc[NG];     // ka[j] == sc[j] (gamma * rstd)
#pragma unroll
  for (int q = 0; q < NG; ++q) {
    const int grp = (c0 + 4 * q) / G;
    const bool hi = grp != grp_lo;
    const float mean = hi ? st_hi[0] : st_lo[0];
    rstd[q] = hi ? st_hi[1] : st_lo[1];
    nmr[q] = -mean * rstd[q];
    kb[q] = rstd[q] * s_c[grp * 2];
    kc[q] = rstd[q] * s_c[grp * 2 + 1];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int q = G4 ? (j >> 2) : 0;
    const float gm = j < 4 ? gm0[j & 3] : gm1[j & 3];
    sc[j] = gm * rstd[q];
    sh[j] = __builtin_fmaf(nmr[q], gm, j < 4 ? bt0[j & 3] : bt1[j & 3]);      // beta - mean * gamma * rstd
  }
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  const int rows = a.HW / a.nchunk;
  const size_t base = ((size_t)n * a.HW + (size_t)chunk * rows) * C + c0;
  constexpr int U = 4;
  for (int r = slot; r < rows; r += U * nslot) {
    f16x8 h[U], d[U], o[RES ? U : 1];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int rr = r + u * nslot;
      if (rr < rows) {
        size_t off = base + (size_t)rr * C;
        h[u] = *reinterpret_cast<const f16x8*>(a.v + off);
        d[u] = *reinterpret_cast<const f16x8*>(a.dy + off);
        if constexpr (RES) o[u] = *reinterpret_cast<const f16x8*>(a.other + off);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int rr = r + u * nslot;
      if (rr < rows) {
        f16x8 out;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          lo_f2 du, xh, dsv;
          lo_f2 ov = {0.f, 0.f};
          if constexpr (RES) ov = (lo_f2){(float)o[u][j], (float)o[u][j + 1]};
          gn_du2(a_mode, (lo_f2){(float)h[u][j], (float)h[u][j + 1]}, (lo_f2){(float)d[u][j], (float)d[u][j + 1]}, ov,
                 (lo_f2){sc[j], sc[j + 1]}, (lo_f2){sh[j], sh[j + 1]}, (lo_f2){rstd[G4 ? (j >> 2) : 0], rstd[G4 ? (j >> 2) : 0]},
                 (lo_f2){nmr[G4 ? (j >> 2) : 0], nmr[G4 ? (j >> 2) : 0]}, du, xh, dsv);
          const int q = G4 ? (j >> 2) : 0;
          const lo_f2 dv = lo_gn_dv2(du, xh, (lo_f2){sc[j], sc[j + 1]}, kb[q], kc[q]);
          const f16 d0 = (f16)dv[0], d1 = (f16)dv[1];
          out[j] = d0;
          out[j + 1] = d1;
          acc[j] += (float)d0;   // bias gradient = sum of the values the weight-gradient GEMM will also see
          acc[j + 1] += (float)d1;
        }
        *reinterpret_cast<f16x8*>(a.dv + base + (size_t)rr * C) = out;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) s_red[tid * 8 + j] = acc[j];
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    int ccx = c >> 3, j = c & 7;
    float tot = 0.f;
    for (int s = 0; s < nslot; ++s) tot += s_red[(s * CC + ccx) * 8 + j];
    a.P2[((size_t)n * a.nchunk + chunk) * C + c] = tot;
  }
}

// ---------------------------------------------------------------------------------------------
// (sample, group)-LOCAL backward: where one GroupNorm group of one sample fits a workgroup (HW * C/8 <= 8192 elements: the
// 16 x 16 x 256 and 8 x 8 x 512 layers), reduce and apply are ONE pass -- dy, v (and the identity of a ResBlock tail) are read once
// and kept in registers between the two phases, the group sums never leave the workgroup.  Replaces lo_gn_bwd_reduce +
// lo_gn_bwd_apply (two launches, two reads of every operand) for the layers whose reduction is not already fused into a
// data-gradient epilogue: the ResBlock tails of encoder stages 3 and 4.  One P1 / P2 row per sample.
//   grid (8 groups, B); thread = (8-channel chunk of the group, row slot); E = rows per thread <= 4.
// ---------------------------------------------------------------------------------------------
// SLAB: the activation gradient is not a tensor yet but the K-split partial sums of the data-gradient convolution that produces it
// (fp32 slabs [nsplit][B*HW][C]) plus an optional residual gradient: summed here, rounded to fp16 (what the tensor would have held)
// and, when somebody else reads it too (dy_out), stored.
struct GnSlabSrc { const float* slab; int nsplit; const f16* add_src; f16* dy_out; size_t split_stride; };
template <bool RES, int E, bool SLAB = false>
__global__ __launch_bounds__(256) void lo_gn_bwd_local_kernel(GnBwdArgs a, GnSlabSrc ss) {
  const int a_mode = RES ? GN_MODE_RES : GN_MODE_PLAIN;
  __shared__ float s_red[256 * 16];
  __shared__ float s_g[64 * 2];
  __shared__ float s_c[2];
  const int tid = threadIdx.x, grp = blockIdx.x, n = blockIdx.y;
  const int C = a.C, G = C >> 3, CG = G >> 3;             // CG = 16-byte chunks per pixel row inside the group
  const int cc = tid % CG, slot = tid / CG, nslot = 256 / CG;
  const int c0 = grp * G + cc * 8;
  const float mean = a.stats[n * 16 + grp * 2], rstd = a.stats[n * 16 + grp * 2 + 1];
  const float nmr = -mean * rstd;
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float gm = a.gamma[c0 + j];
    sc[j] = gm * rstd;
    sh[j] = __builtin_fmaf(nmr, gm, a.beta[c0 + j]);
  }
  const size_t base = (size_t)n * a.HW * C + c0;
  lo_f2 du[E][4], xh[E][4];
  f16x8 h[E], d[E], o[RES ? E : 1];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int r = slot + e * nslot;
    if (r < a.HW) {
      const size_t off = base + (size_t)r * C;
      h[e] = *reinterpret_cast<const f16x8*>(a.v + off);
      if constexpr (SLAB) {
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
        for (int sp = 0; sp < ss.nsplit; ++sp) {
          s0 += *reinterpret_cast<const f32x4*>(ss.slab + (size_t)sp * ss.split_stride + off);
          s1 += *reinterpret_cast<const f32x4*>(ss.slab + (size_t)sp * ss.split_stride + off + 4);
        }
        f16x8 dd = {(f16)s0[0], (f16)s0[1], (f16)s0[2], (f16)s0[3], (f16)s1[0], (f16)s1[1], (f16)s1[2], (f16)s1[3]};
        if (ss.add_src) {
          const f16x8 rr = *reinterpret_cast<const f16x8*>(ss.add_src + off);
#pragma unroll
          for (int j = 0; j < 8; ++j) dd[j] = (f16)((float)dd[j] + (float)rr[j]);
        }
        if (ss.dy_out) *reinterpret_cast<f16x8*>(ss.dy_out + off) = dd;
        d[e] = dd;
      } else {
        d[e] = *reinterpret_cast<const f16x8*>(a.dy + off);
      }
      if constexpr (RES) o[e] = *reinterpret_cast<const f16x8*>(a.other + off);
    }
  }
  lo_f2 a1[4], a2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { a1[j] = (lo_f2){0.f, 0.f}; a2[j] = (lo_f2){0.f, 0.f}; }
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int r = slot + e * nslot;
    if (r < a.HW) {
      f16x8 dso;
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        lo_f2 dsv;
        lo_f2 ov = {0.f, 0.f};
        if constexpr (RES) ov = (lo_f2){(float)o[e][j], (float)o[e][j + 1]};
        gn_du2(a_mode, (lo_f2){(float)h[e][j], (float)h[e][j + 1]}, (lo_f2){(float)d[e][j], (float)d[e][j + 1]}, ov, (lo_f2){sc[j], sc[j + 1]},
               (lo_f2){sh[j], sh[j + 1]}, (lo_f2){rstd, rstd}, (lo_f2){nmr, nmr}, du[e][j >> 1], xh[e][j >> 1], dsv);
        a1[j >> 1] += du[e][j >> 1];
        a2[j >> 1] += du[e][j >> 1] * xh[e][j >> 1];
        dso[j] = (f16)dsv[0];
        dso[j + 1] = (f16)dsv[1];
      }
      if (RES) *reinterpret_cast<f16x8*>(a.ds + base + (size_t)r * C) = dso;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) { du[e][j] = (lo_f2){0.f, 0.f}; xh[e][j] = (lo_f2){0.f, 0.f}; }
    }
  }
  // per-channel sums over the row slots (fixed order) -> P1 row of this sample; gamma-weighted group sums
#pragma unroll
  for (int j = 0; j < 8; ++j) { s_red[tid * 16 + j * 2] = a1[j >> 1][j & 1]; s_red[tid * 16 + j * 2 + 1] = a2[j >> 1][j & 1]; }
  __syncthreads();
  for (int q = tid; q < G * 2; q += 256) {
    const int c = q >> 1, w = q & 1, ccx = c >> 3, j = c & 7;
    float tot = 0.f;
    for (int sl = 0; sl < nslot; ++sl) tot += s_red[(sl * CG + ccx) * 16 + j * 2 + w];
    a.P1[((size_t)n * C + grp * G + c) * 2 + w] = tot;
    s_g[c * 2 + w] = a.gamma[grp * G + c] * tot;
  }
  __syncthreads();
  if (tid < 2) {
    float tot = 0.f;
    for (int c = 0; c < G; ++c) tot += s_g[c * 2 + tid];
    s_c[tid] = tot / ((float)a.HW * (float)G);
  }
  __syncthreads();
  const float kb = rstd * s_c[0], kc = rstd * s_c[1];
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int r = slot + e * nslot;
    if (r < a.HW) {
      f16x8 out;
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        const lo_f2 dv = lo_gn_dv2(du[e][j >> 1], xh[e][j >> 1], (lo_f2){sc[j], sc[j + 1]}, kb, kc);
        const f16 d0 = (f16)dv[0], d1 = (f16)dv[1];
        out[j] = d0; out[j + 1] = d1;
        acc[j] += (float)d0; acc[j + 1] += (float)d1;
      }
      *reinterpret_cast<f16x8*>(a.dv + base + (size_t)r * C) = out;
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 8; ++j) s_red[tid * 8 + j] = acc[j];
  __syncthreads();
  for (int c = tid; c < G; c += 256) {
    const int ccx = c >> 3, j = c & 7;
    float tot = 0.f;
    for (int sl = 0; sl < nslot; ++sl) tot += s_red[(sl * CG + ccx) * 8 + j];
    a.P2[(size_t)n * C + grp * G + c] = tot;
  }
}

// dgamma, dbeta, dbias from the partials (fixed summation order): block = 16 channels x 16 row-lanes
__global__ __launch_bounds__(256) void lo_gn_param_finalize_kernel(const float* __restrict__ P1, const float* __restrict__ P2,
                                                                   float* dgamma, float* dbeta, float* dbias, int nblk, int C,
                                                                   float scale) {
  __shared__ float red[3][16][17];
  const int cl = threadIdx.x & 15, r = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float g1 = 0.f, g2 = 0.f, b = 0.f;
  if (c < C)
    for (int k = r; k < nblk; k += 16) {
      f32x2 p = *reinterpret_cast<const f32x2*>(P1 + ((size_t)k * C + c) * 2);
      g1 += p[0];
      g2 += p[1];
      b += P2[(size_t)k * C + c];
    }
  red[0][r][cl] = g1; red[1][r][cl] = g2; red[2][r][cl] = b;
  __syncthreads();
  if (r < 3 && c < C) {
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) tot += red[r][q][cl];
    tot *= scale;
    if (r == 0) dbeta[c] = tot; else if (r == 1) dgamma[c] = tot; else dbias[c] = tot;
  }
}

// the same for every GroupNorm layer of the model in ONE launch (jobs in the kernel argument)
__global__ __launch_bounds__(256) void lo_gn_finalize_all_kernel(LoGnFinJobs jobs, float scale) {
  __shared__ float red[3][64][5];
  int jb = 0;
  while (jb + 1 < jobs.n && (int)blockIdx.x >= jobs.j[jb + 1].block0) ++jb;
  const LoGnFinJob& J = jobs.j[jb];
  const int C = J.C;
  const int cl = threadIdx.x & 3, r = threadIdx.x >> 2;       // 4 channels x 64 row lanes
  const int c = ((int)blockIdx.x - J.block0) * 4 + cl;
  float g1 = 0.f, g2 = 0.f, b = 0.f;
  if (c < C) {
    for (int k = r; k < J.nblk1; k += 64) {
      f32x2 p = *reinterpret_cast<const f32x2*>(J.P1 + ((size_t)k * C + c) * 2);
      g1 += p[0];
      g2 += p[1];
    }
    for (int k = r; k < J.nblk2; k += 64) b += J.P2[(size_t)k * C + c];
  }
  red[0][r][cl] = g1; red[1][r][cl] = g2; red[2][r][cl] = b;
  __syncthreads();
  if (threadIdx.x < 12) {
    const int which = threadIdx.x >> 2, c2 = threadIdx.x & 3;
    const int cg = ((int)blockIdx.x - J.block0) * 4 + c2;
    if (cg < C) {
      float tot = 0.f;
      for (int q = 0; q < 64; ++q) tot += red[which][q][c2];
      tot *= scale;
      if (which == 0) J.dbeta[cg] = tot; else if (which == 1) J.dgamma[cg] = tot; else J.dbias[cg] = tot;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// small layout helpers
// ---------------------------------------------------------------------------------------------
// [B][HW][C] (NHWC) <-> [B][C][HW] (the order nn.Flatten sees, lunar_generate.py:150 / .view at :208)
// Per-sample matrix transpose [R][Cc] -> [Cc][R] of fp16 through a 64 x 64 LDS tile: 128-byte coalesced rows on both sides
// (NHWC <-> NCHW flatten order of the 8 x 8 x 512 bottleneck around the Linear layers: R x Cc = 64 x 512 or 512 x 64).
// dst8: optional e4m3 copy of the output (fp8 operand mode).  R % 64 == 0, Cc % 64 == 0.
__global__ __launch_bounds__(256) void lo_transpose_tile_kernel(const f16* __restrict__ src, f16* __restrict__ dst, uint8_t* __restrict__ dst8,
                                                                int R, int Cc) {
  __shared__ f16 tile[64][66];
  const int n = blockIdx.z, r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const f16* s = src + (size_t)n * R * Cc;
  const int tc = (threadIdx.x & 7) * 8, tr = threadIdx.x >> 3;     // 8 chunks of 8 elements per row, 32 rows per pass
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int r = tr + 32 * p;
    const f16x8 v = *reinterpret_cast<const f16x8*>(s + (size_t)(r0 + r) * Cc + c0 + tc);
#pragma unroll
    for (int j = 0; j < 8; ++j) tile[r][tc + j] = v[j];
  }
  __syncthreads();
  f16* d = dst + (size_t)n * R * Cc;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int c = tr + 32 * p;                                     // output row = input column
    f16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = tile[tc + j][c];
    const size_t off = (size_t)(c0 + c) * R + r0 + tc;
    *reinterpret_cast<f16x8*>(d + off) = v;
    if (dst8) {
      constexpr float S8 = LO_F8_ACT_SCALE;
      *reinterpret_cast<u32x2*>(dst8 + (size_t)n * R * Cc + off) =
          (u32x2){lo_pack4_fp8((float)v[0] * S8, (float)v[1] * S8, (float)v[2] * S8, (float)v[3] * S8),
                  lo_pack4_fp8((float)v[4] * S8, (float)v[5] * S8, (float)v[6] * S8, (float)v[7] * S8)};
    }
  }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
// profiler name of a launch: the kernel name, or "<kernel> C<channels> HW<pixels> m<mode>" under LO_PROF_LAYERS
static const char* gn_layer_name(const char* base, int C, int HW, int mode) {
  if (!g_lo_prof_layers || !g_lo_prof_on) return base;
  struct Entry { const char* base; int C, HW, mode; char text[56]; };
  static Entry table[96];             // one stable string per distinct (kernel, shape, mode): the profiler keeps the pointer
  static int used = 0;
  for (int i = 0; i < used; ++i)
    if (table[i].base == base && table[i].C == C && table[i].HW == HW && table[i].mode == mode) return table[i].text;
  if (used == 96) return base;
  Entry& e = table[used++];
  e.base = base; e.C = C; e.HW = HW; e.mode = mode;
  snprintf(e.text, sizeof(e.text), "%s C%d HW%d m%d", base, C, HW, mode);
  return e.text;
}
int lo_gn_fwd(const f16* v, const float* partial, int MT, const float* gamma, const float* beta, const f16* other,
              f16* y, float* stats, int B, int HW, int C, int mode, hipStream_t st, uint8_t* y8) {
  LO_REQUIRE(C % 32 == 0 && C <= 512, "lo_gn_fwd: C=%d unsupported", C);
  GnFwdArgs a{v, partial, gamma, beta, other, y, stats, HW, C, MT, lo_gn_nchunk(HW, C), mode, y8};
  LoProfScope _p(gn_layer_name("lo_gn_fwd", C, HW, mode), 0, 2.0 * B * HW * C * (mode ? 3 : 2), st);
  if (mode == GN_MODE_PLAIN) hipLaunchKernelGGL(lo_gn_fwd_kernel<GN_MODE_PLAIN>, dim3(a.nchunk, B), dim3(256), 0, st, a);
  else if (mode == GN_MODE_SKIP) hipLaunchKernelGGL(lo_gn_fwd_kernel<GN_MODE_SKIP>, dim3(a.nchunk, B), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(lo_gn_fwd_kernel<GN_MODE_RES>, dim3(a.nchunk, B), dim3(256), 0, st, a);
  LO_LAUNCH_CHECK("gn_fwd");
  return LO_OK;
}

int lo_gn_bwd_nofinal(const f16* dy, const f16* v, const f16* other, const float* stats, const float* gamma, const float* beta,
                      f16* ds, f16* dv, float* P1, float* P2, int B, int HW, int C, int mode, hipStream_t st, int np1) {
  LO_REQUIRE(C % 32 == 0 && C <= 512, "lo_gn_bwd: C=%d unsupported", C);
  GnBwdArgs a{dy, v, other, stats, gamma, beta, ds, dv, P1, P2, HW, C, lo_gn_nchunk(HW, C), mode, 0};
  a.np1 = np1 > 0 ? np1 : a.nchunk;
  if (np1 == 0) {
    LoProfScope _p(gn_layer_name("lo_gn_bwd_reduce", C, HW, mode), 0, 2.0 * B * HW * C * (mode == 2 ? 4 : 2), st);
    const bool g4 = C == 32;
    if (mode == GN_MODE_RES) {
      if (g4) hipLaunchKernelGGL((lo_gn_bwd_reduce_kernel<true, true>), dim3(a.nchunk, B), dim3(256), 0, st, a);
      else hipLaunchKernelGGL((lo_gn_bwd_reduce_kernel<true, false>), dim3(a.nchunk, B), dim3(256), 0, st, a);
    } else {
      if (g4) hipLaunchKernelGGL((lo_gn_bwd_reduce_kernel<false, true>), dim3(a.nchunk, B), dim3(256), 0, st, a);
      else hipLaunchKernelGGL((lo_gn_bwd_reduce_kernel<false, false>), dim3(a.nchunk, B), dim3(256), 0, st, a);
    }
  }
  LO_LAUNCH_CHECK("gn_bwd_reduce");
  {
    LoProfScope _p(gn_layer_name("lo_gn_bwd_apply", C, HW, mode), 0, 2.0 * B * HW * C * (mode == 2 ? 4 : 3), st);
    const bool g4 = C == 32;
    if (mode == GN_MODE_RES) {
      if (g4) LO_LAUNCH_STOP((lo_gn_bwd_apply_kernel<true, true>), dim3(a.nchunk, B), dim3(256), 0, st, a);
      else LO_LAUNCH_STOP((lo_gn_bwd_apply_kernel<true, false>), dim3(a.nchunk, B), dim3(256), 0, st, a);
    } else {
      if (g4) LO_LAUNCH_STOP((lo_gn_bwd_apply_kernel<false, true>), dim3(a.nchunk, B), dim3(256), 0, st, a);
      else LO_LAUNCH_STOP((lo_gn_bwd_apply_kernel<false, false>), dim3(a.nchunk, B), dim3(256), 0, st, a);
    }
  }
  LO_LAUNCH_CHECK("gn_bwd_apply");
  return LO_OK;
}

// ---------------------------------------------------------------------------------------------
// Forward counterpart for split-K convolutions: one workgroup per (sample, group) sums the K-split slabs of its 8-channel chunks,
// adds the bias, rounds to fp16 (v, stored: the backward needs it), takes the group's mean / rstd over those fp16 values (fixed-order
// sums, double at the end like lo_gn_group_stats) and applies GroupNorm + Mish (lo_gn_apply8: the modes of lo_gn_fwd).
// ---------------------------------------------------------------------------------------------
struct SplitkGnFwdArgs {
  const float* slab; int nsplit; size_t split_stride;
  const float* bias; const float* gamma; const float* beta; const f16* other;
  f16* v; f16* y; float* stats;
  int HW, C, mode;
};
template <int E>
__global__ __launch_bounds__(256) void lo_splitk_gn_fwd_kernel(SplitkGnFwdArgs a) {
  __shared__ float s_red[256 * 2];
  __shared__ float s_stat[2];
  const int tid = threadIdx.x, grp = blockIdx.x, n = blockIdx.y;
  const int C = a.C, G = C >> 3, CG = G >> 3;
  const int cc = tid % CG, slot = tid / CG, nslot = 256 / CG;
  const int c0 = grp * G + cc * 8;
  const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.bias + c0), b1 = *reinterpret_cast<const f32x4*>(a.bias + c0 + 4);
  const size_t base = (size_t)n * a.HW * C + c0;
  f16x8 h[E];
  float s = 0.f, q = 0.f;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int r = slot + e * nslot;
    if (r < a.HW) {
      const size_t off = base + (size_t)r * C;
      f32x4 s0 = b0, s1 = b1;
      for (int sp = 0; sp < a.nsplit; ++sp) {
        s0 += *reinterpret_cast<const f32x4*>(a.slab + (size_t)sp * a.split_stride + off);
        s1 += *reinterpret_cast<const f32x4*>(a.slab + (size_t)sp * a.split_stride + off + 4);
      }
      h[e] = (f16x8){(f16)s0[0], (f16)s0[1], (f16)s0[2], (f16)s0[3], (f16)s1[0], (f16)s1[1], (f16)s1[2], (f16)s1[3]};
      *reinterpret_cast<f16x8*>(a.v + off) = h[e];
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float x = (float)h[e][j]; s += x; q += x * x; }
    }
  }
  s_red[tid * 2] = s; s_red[tid * 2 + 1] = q;
  __syncthreads();
  if (tid < 2) {
    double tot = 0.0;
    for (int t = 0; t < 256; ++t) tot += (double)s_red[t * 2 + tid];
    s_stat[tid] = (float)tot;
  }
  __syncthreads();
  if (tid == 0) {
    const double inv_m = 1.0 / ((double)a.HW * (double)G);
    const double mean = (double)s_stat[0] * inv_m;
    double var = (double)s_stat[1] * inv_m - mean * mean;
    if (var < 0.0) var = 0.0;
    const float mf = (float)mean, rf = (float)(1.0 / sqrt(var + (double)1e-5f));
    s_stat[0] = mf; s_stat[1] = rf;
    a.stats[n * 16 + grp * 2] = mf; a.stats[n * 16 + grp * 2 + 1] = rf;
  }
  __syncthreads();
  const float mean = s_stat[0], rstd = s_stat[1];
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) lo_gn_scale_shift(a.gamma[c0 + j], a.beta[c0 + j], mean, rstd, sc[j], sh[j]);
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int r = slot + e * nslot;
    if (r < a.HW) {
      const size_t off = base + (size_t)r * C;
      f16x8 o = h[e];
      if (a.mode != 0) o = *reinterpret_cast<const f16x8*>(a.other + off);
      *reinterpret_cast<f16x8*>(a.y + off) = lo_gn_apply8(h[e], sc, sh, a.mode, o);
    }
  }
}
int lo_splitk_gn_fwd(const float* slab, int nsplit, const float* bias, const float* gamma, const float* beta, const f16* other, f16* v,
                     f16* y, float* stats, int B, int HW, int C, int mode, hipStream_t st) {
  LO_REQUIRE(lo_gn_bwd_local_applies(HW, C) && slab && bias && v && y && stats && nsplit >= 1 && (mode == 0 || other),
             "lo_splitk_gn_fwd: bad arguments (HW=%d, C=%d)", HW, C);
  SplitkGnFwdArgs a{slab, nsplit, (size_t)B * HW * C, bias, gamma, beta, other, v, y, stats, HW, C, mode};
  const int E = HW / (256 / ((C >> 3) >> 3));
  LoProfScope _p(gn_layer_name("lo_splitk_gn_fwd", C, HW, mode), 0, (double)B * HW * C * (4.0 * nsplit + 4.0 + (mode ? 2.0 : 0.0)), st);
  const dim3 grid(8, B);
  if (E <= 1) hipLaunchKernelGGL((lo_splitk_gn_fwd_kernel<1>), grid, dim3(256), 0, st, a);
  else if (E == 2) hipLaunchKernelGGL((lo_splitk_gn_fwd_kernel<2>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((lo_splitk_gn_fwd_kernel<4>), grid, dim3(256), 0, st, a);
  LO_LAUNCH_CHECK("splitk_gn_fwd");
  return LO_OK;
}
int lo_splitk_gn_bwd(const float* slab, int nsplit, const f16* add_src, const f16* v, const float* stats, const float* gamma,
                     const float* beta, f16* dy_out, f16* dv, float* P1, float* P2, int B, int HW, int C, hipStream_t st) {
  LO_REQUIRE(lo_gn_bwd_local_applies(HW, C) && slab && v && stats && dv && P1 && P2 && nsplit >= 1, "lo_splitk_gn_bwd: bad arguments (HW=%d, C=%d)", HW, C);
  GnBwdArgs a{nullptr, v, nullptr, stats, gamma, beta, nullptr, dv, P1, P2, HW, C, 1, GN_MODE_PLAIN, 1};
  const GnSlabSrc ss{slab, nsplit, add_src, dy_out, (size_t)B * HW * C};
  const int E = HW / (256 / ((C >> 3) >> 3));
  LoProfScope _p(gn_layer_name("lo_splitk_gn_bwd", C, HW, 0), 0, (double)B * HW * C * (4.0 * nsplit + 4.0 + (add_src ? 2.0 : 0.0)), st);
  const dim3 grid(8, B);
  if (E <= 1) LO_LAUNCH_STOP((lo_gn_bwd_local_kernel<false, 1, true>), grid, dim3(256), 0, st, a, ss);
  else if (E == 2) LO_LAUNCH_STOP((lo_gn_bwd_local_kernel<false, 2, true>), grid, dim3(256), 0, st, a, ss);
  else LO_LAUNCH_STOP((lo_gn_bwd_local_kernel<false, 4, true>), grid, dim3(256), 0, st, a, ss);
  LO_LAUNCH_CHECK("splitk_gn_bwd");
  return LO_OK;
}

// the one-pass (sample, group)-local form: P1 / P2 get ONE row per sample
bool lo_gn_bwd_local_applies(int HW, int C) {
  const int G = C >> 3;
  return C % 64 == 0 && G <= 64 && (long)HW * G <= 8192 && HW % (256 / (G >> 3)) == 0 && HW / (256 / (G >> 3)) <= 4;
}
int lo_gn_bwd_local(const f16* dy, const f16* v, const f16* other, const float* stats, const float* gamma, const float* beta, f16* ds,
                    f16* dv, float* P1, float* P2, int B, int HW, int C, int mode, hipStream_t st) {
  LO_REQUIRE(lo_gn_bwd_local_applies(HW, C), "lo_gn_bwd_local: HW=%d, C=%d does not fit one workgroup per (sample, group)", HW, C);
  GnBwdArgs a{dy, v, other, stats, gamma, beta, ds, dv, P1, P2, HW, C, 1, mode, 1};
  const int E = HW / (256 / ((C >> 3) >> 3));
  LoProfScope _p(gn_layer_name("lo_gn_bwd_local", C, HW, mode), 0, 2.0 * B * HW * C * (mode == 2 ? 5 : 3), st);
  const dim3 grid(8, B);
  const GnSlabSrc ss{nullptr, 0, nullptr, nullptr, 0};
#define LO_GNL(RES_, E_) LO_LAUNCH_STOP((lo_gn_bwd_local_kernel<RES_, E_>), grid, dim3(256), 0, st, a, ss)
  if (mode == GN_MODE_RES) { if (E <= 1) LO_GNL(true, 1); else if (E == 2) LO_GNL(true, 2); else LO_GNL(true, 4); }
  else { if (E <= 1) LO_GNL(false, 1); else if (E == 2) LO_GNL(false, 2); else LO_GNL(false, 4); }
#undef LO_GNL
  LO_LAUNCH_CHECK("gn_bwd_local");
  return LO_OK;
}

int lo_gn_bwd(const f16* dy, const f16* v, const f16* other, const float* stats, const float* gamma, const float* beta,
              f16* ds, f16* dv, float* P1, float* P2, float* dgamma, float* dbeta, float* dbias, int B, int HW, int C,
              int mode, float scale, hipStream_t st) {
  int r = lo_gn_bwd_nofinal(dy, v, other, stats, gamma, beta, ds, dv, P1, P2, B, HW, C, mode, st, 0);
  if (r != LO_OK) return r;
  LoProfScope _p3("lo_gn_param_finalize", 0, 0, st);
  hipLaunchKernelGGL(lo_gn_param_finalize_kernel, dim3((C + 15) / 16), dim3(256), 0, st, P1, P2, dgamma, dbeta, dbias,
                     B * lo_gn_nchunk(HW, C), C, scale);
  LO_LAUNCH_CHECK("gn_param_finalize");
  return LO_OK;
}

int lo_gn_finalize_all(const LoGnFinJobs& jobs, float scale, hipStream_t st) {
  int nblocks = 0;
  for (int i = 0; i < jobs.n; ++i) nblocks += (jobs.j[i].C + 3) / 4;
  LoProfScope _p("lo_gn_finalize_all", 0, 0, st);
  hipLaunchKernelGGL(lo_gn_finalize_all_kernel, dim3(nblocks), dim3(256), 0, st, jobs, scale);
  LO_LAUNCH_CHECK("gn_finalize_all");
  return LO_OK;
}

int lo_nhwc_to_nchw_f16(const f16* src, f16* dst, int B, int HW, int C, hipStream_t st) {
  LO_REQUIRE(HW % 64 == 0 && C % 64 == 0, "lo_nhwc_to_nchw_f16: HW=%d, C=%d must be multiples of 64", HW, C);
  LoProfScope _p("lo_layout_transpose", 0, 4.0 * B * HW * C, st);
  LO_LAUNCH_STOP(lo_transpose_tile_kernel, dim3(C / 64, HW / 64, B), dim3(256), 0, st, src, dst, (uint8_t*)nullptr, HW, C);
  LO_LAUNCH_CHECK("nhwc_to_nchw");
  return LO_OK;
}
int lo_nchw_to_nhwc_f16(const f16* src, f16* dst, int B, int HW, int C, hipStream_t st, uint8_t* dst8) {
  LO_REQUIRE(HW % 64 == 0 && C % 64 == 0, "lo_nchw_to_nhwc_f16: HW=%d, C=%d must be multiples of 64", HW, C);
  LoProfScope _p("lo_layout_transpose", 0, 4.0 * B * HW * C, st);
  LO_LAUNCH_STOP(lo_transpose_tile_kernel, dim3(HW / 64, C / 64, B), dim3(256), 0, st, src, dst, dst8, C, HW);
  LO_LAUNCH_CHECK("nchw_to_nhwc");
  return LO_OK;
}

// Boundary forms of the same per-sample transpose [R][Cc] -> [Cc][R] with a type change and a scale (off the step's hot path):
// TI / TO = f16 or float.  Used where feature maps cross the nn.Module boundary as fp32 NCHW tensors (Encoder.forward's skip list,
// Decoder.forward's skips argument: lunar_generate.py:139-145, 211-222) while the library keeps fp16 NHWC.  R % 64 == 0, Cc % 64 == 0.
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void lo_transpose_convert_kernel(const TI* __restrict__ src, TO* __restrict__ dst, int R, int Cc, float scale) {
  __shared__ float tile[64][65];
  const int n = blockIdx.z, r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const TI* s = src + (size_t)n * R * Cc;
  const int tc = threadIdx.x & 63, tr = threadIdx.x >> 6;
#pragma unroll
  for (int p = 0; p < 16; ++p) tile[tr + 4 * p][tc] = (float)s[(size_t)(r0 + tr + 4 * p) * Cc + c0 + tc] * scale;
  __syncthreads();
  TO* d = dst + (size_t)n * R * Cc;
#pragma unroll
  for (int p = 0; p < 16; ++p) d[(size_t)(c0 + tr + 4 * p) * R + r0 + tc] = (TO)tile[tc][tr + 4 * p];
}
int lo_nhwc_f16_to_nchw_f32(const f16* src, float* dst, int B, int HW, int C, float scale, hipStream_t st) {
  LO_REQUIRE(HW % 64 == 0 && C % 64 == 0, "lo_nhwc_f16_to_nchw_f32: HW=%d, C=%d must be multiples of 64", HW, C);
  hipLaunchKernelGGL((lo_transpose_convert_kernel<f16, float>), dim3(C / 64, HW / 64, B), dim3(256), 0, st, src, dst, HW, C, scale);
  LO_LAUNCH_CHECK("nhwc_f16_to_nchw_f32");
  return LO_OK;
}
int lo_nchw_f32_to_nhwc_f16(const float* src, f16* dst, int B, int HW, int C, float scale, hipStream_t st) {
  LO_REQUIRE(HW % 64 == 0 && C % 64 == 0, "lo_nchw_f32_to_nhwc_f16: HW=%d, C=%d must be multiples of 64", HW, C);
  hipLaunchKernelGGL((lo_transpose_convert_kernel<float, f16>), dim3(HW / 64, C / 64, B), dim3(256), 0, st, src, dst, C, HW, scale);
  LO_LAUNCH_CHECK("nchw_f32_to_nhwc_f16");
  return LO_OK;
}
