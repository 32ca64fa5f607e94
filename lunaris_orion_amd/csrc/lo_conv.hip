// Implicit-GEMM convolution kernels for gfx950 (MI355X): fp16 NHWC activations, fp32 accumulation on
// v_mfma_f32_16x16x32_f16, LDS-staged operand tiles (register prefetch, two LDS stages, one barrier
// per K step), LDS-staged coalesced epilogue with fused bias / residual-add / GroupNorm partial sums.
//
// One kernel template covers every forward conv, transposed conv (as 4 sub-pixel phases) and every
// data-gradient of the VAE (LoGeom in lo_common.h), plus the Linear layers (1 tap, optional split-K).
// A second template (lo_wgrad_tn) computes weight gradients: both operands are reduced over the pixel
// index, which is the slow axis of NHWC, so their MFMA fragments are read with ds_read_b64_tr_b16.
//
// Reference ops replaced (PyTorch ATen, dispatched from /root/reference/lunar_generate.py):
//   conv2d            :36,41,95,102,109,116   conv_transpose2d :169,175,181,187   linear :124,125,165
#include "lo_common.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

// ---------------------------------------------------------------------------------------------
// geometry builder (host)
// ---------------------------------------------------------------------------------------------
int lo_make_geom(LoGeom* g, int kind, int B, int H, int W, int Cin, int Cout) {
  memset(g, 0, sizeof(*g));
  g->B = B; g->Hin = H; g->Win = W; g->Cin = Cin; g->Cout = Cout;
  auto tap = [&](int p, int dy, int dx, int r, int s, int S) {
    int t = g->T[p]++;
    g->dy[p][t] = (int8_t)dy; g->dx[p][t] = (int8_t)dx; g->rs[p][t] = (int8_t)(r * S + s);
  };
  switch (kind) {
    case LO_CONV3_S1:
    case LO_CONV3_S2: {
      int st = kind == LO_CONV3_S1 ? 1 : 2;
      g->Hout = H / st; g->Wout = W / st; g->GH = g->Hout; g->GW = g->Wout;
      g->in_stride = st; g->out_stride = 1; g->n_phase = 1;
      for (int r = 0; r < 3; ++r) for (int s = 0; s < 3; ++s) tap(0, r - 1, s - 1, r, s, 3);
      g->sn = Cin * 9; g->sc = 9;   // Conv2d weight [Cout][Cin][3][3]
      break;
    }
    case LO_CONV3_S1_DGRAD: {
      // dx[ih] = sum_r dy[ih + 1 - r] W[co][ci][r]; this op's output channel n = ci_fwd, reduced c = co_fwd
      g->Hout = H; g->Wout = W; g->GH = H; g->GW = W; g->in_stride = 1; g->out_stride = 1; g->n_phase = 1;
      for (int r = 0; r < 3; ++r) for (int s = 0; s < 3; ++s) tap(0, 1 - r, 1 - s, r, s, 3);
      g->sn = 9; g->sc = Cout * 9;  // W[co_fwd = c][ci_fwd = n][3][3], Cin_fwd = Cout of this op
      break;
    }
    case LO_CONV3_S2_DGRAD: {
      // forward: oh = (ih + 1 - r)/2.  Reads dy [B,H,W,Cin=Cout_fwd], writes dx [B,2H,2W,Cout=Cin_fwd].
      g->Hout = 2 * H; g->Wout = 2 * W; g->GH = H; g->GW = W; g->in_stride = 1; g->out_stride = 2; g->n_phase = 4;
      for (int ph = 0; ph < 2; ++ph) for (int pw = 0; pw < 2; ++pw) {
        int p = ph * 2 + pw; g->out_oy[p] = ph; g->out_ox[p] = pw;
        int nr = ph ? 2 : 1, ns = pw ? 2 : 1;
        int rr[2], dyy[2], ss[2], dxx[2];
        if (!ph) { rr[0] = 1; dyy[0] = 0; } else { rr[0] = 0; dyy[0] = 1; rr[1] = 2; dyy[1] = 0; }
        if (!pw) { ss[0] = 1; dxx[0] = 0; } else { ss[0] = 0; dxx[0] = 1; ss[1] = 2; dxx[1] = 0; }
        for (int a = 0; a < nr; ++a) for (int b = 0; b < ns; ++b) tap(p, dyy[a], dxx[b], rr[a], ss[b], 3);
      }
      g->sn = 9; g->sc = Cout * 9;
      break;
    }
    case LO_CONVT4_S2: {
      // oh = 2 ih - 1 + r.  ConvTranspose2d weight [Cin][Cout][4][4]
      g->Hout = 2 * H; g->Wout = 2 * W; g->GH = H; g->GW = W; g->in_stride = 1; g->out_stride = 2; g->n_phase = 4;
      for (int ph = 0; ph < 2; ++ph) for (int pw = 0; pw < 2; ++pw) {
        int p = ph * 2 + pw; g->out_oy[p] = ph; g->out_ox[p] = pw;
        int rr[2], dyy[2], ss[2], dxx[2];
        if (!ph) { rr[0] = 1; dyy[0] = 0; rr[1] = 3; dyy[1] = -1; } else { rr[0] = 0; dyy[0] = 1; rr[1] = 2; dyy[1] = 0; }
        if (!pw) { ss[0] = 1; dxx[0] = 0; ss[1] = 3; dxx[1] = -1; } else { ss[0] = 0; dxx[0] = 1; ss[1] = 2; dxx[1] = 0; }
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) tap(p, dyy[a], dxx[b], rr[a], ss[b], 4);
      }
      g->sn = 16; g->sc = Cout * 16;  // W[ci = c][co = n][4][4]
      break;
    }
    case LO_CONVT4_S2_DGRAD: {
      // din[ih] = sum_r dout[2 ih - 1 + r] W[ci][co][r]: a k4 s2 p1 convolution of dout.
      g->Hout = H / 2; g->Wout = W / 2; g->GH = g->Hout; g->GW = g->Wout; g->in_stride = 2; g->out_stride = 1; g->n_phase = 1;
      for (int r = 0; r < 4; ++r) for (int s = 0; s < 4; ++s) tap(0, r - 1, s - 1, r, s, 4);
      g->sn = Cin * 16; g->sc = 16;   // W[ci_fwd = n][co_fwd = c][4][4], Cout_fwd = Cin of this op
      break;
    }
    case LO_LINEAR: {
      g->Hout = H; g->Wout = W; g->GH = H; g->GW = W; g->in_stride = 1; g->out_stride = 1; g->n_phase = 1;
      tap(0, 0, 0, 0, 0, 1);
      g->sn = Cin; g->sc = 1;
      break;
    }
    default:
      lo_set_error("lo_make_geom: unknown kind %d", kind);
      return LO_ERR_ARG;
  }
  int off = 0;
  for (int p = 0; p < g->n_phase; ++p) { g->wofs[p] = off; off += Cout * g->T[p] * Cin; }
  return LO_OK;
}

static inline int geom_packed_elems(const LoGeom& g) {
  int off = 0;
  for (int p = 0; p < g.n_phase; ++p) off += g.Cout * g.T[p] * g.Cin;
  return off;
}

// ---------------------------------------------------------------------------------------------
// weight pack: canonical fp32 -> packed fp16  Wp[p][n][t*Cin + c] = W[n*sn + c*sc + rs[p][t]]
// (optionally with a row permutation for decoder.fc:  n -> row_perm(n))
// ---------------------------------------------------------------------------------------------
__global__ void lo_pack_weight_kernel(const float* __restrict__ w, f16* __restrict__ wp, LoGeom g, int total) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int p = 0;
#pragma unroll
  for (int q = 1; q < LO_MAX_PHASE; ++q) if (q < g.n_phase && i >= g.wofs[q]) p = q;
  int j = i - g.wofs[p];
  int K = g.T[p] * g.Cin;
  int n = j / K, k = j - n * K;
  int t = k / g.Cin, c = k - t * g.Cin;
  wp[i] = (f16)w[(size_t)n * g.sn + (size_t)c * g.sc + g.rs[p][t]];
}

// ---------------------------------------------------------------------------------------------
// NT implicit GEMM
// ---------------------------------------------------------------------------------------------
struct IgemmArgs {
  const f16* in;
  const f16* w;        // packed fp16 weights
  const float* bias;   // [Cout] or null
  const f16* add_src;  // same layout as out, or null
  f16* out;
  float* gn_partial;   // [B][MT][8][2] or null
  float* slab;         // split-K fp32 partials [nsplit][M][Cout] (SPLITK only)
  int M;               // rows per phase = B*GH*GW
  int nsplit;          // >= 1
  int ksteps_per_split;
  LoGeom g;
};

template <int BM, int BN, int BK, bool SPLITK>
__global__ __launch_bounds__(256) void lo_igemm_nt(IgemmArgs a) {
  constexpr int CPR = BK / 8;             // 16-byte chunks per tile row
  constexpr int RPP = 256 / CPR;          // rows loaded per pass
  constexpr int PA = BM / RPP, PB = BN / RPP;
  static_assert(PA >= 1 && PB >= 1, "tile too small for 256 threads");
  constexpr int PITCH = BK * 2 + 32;      // bytes; +32 makes the ds_read_b128 fragment reads conflict-free
  constexpr int A_BYTES = BM * PITCH, B_BYTES = BN * PITCH;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int WM = BM / 2, WN = BN / 2; // wave tile (2 x 2 waves)
  constexpr int MI = WM / 16, NI = WN / 16;
  constexpr int OPITCH = BN * 2 + 16;     // epilogue staging pitch (bytes)
  constexpr int LDS_BYTES = (2 * STAGE > BM * OPITCH + 4096) ? 2 * STAGE : BM * OPITCH + 4096;
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];

  const LoGeom& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int phase = SPLITK ? 0 : blockIdx.z;
  const int split = SPLITK ? blockIdx.z : 0;
  const int T = g.T[phase];
  const int KCB = g.Cin / BK;
  const int ksteps_total = T * KCB;
  int ks_begin = 0, ks_end = ksteps_total;
  if (SPLITK) {
    ks_begin = split * a.ksteps_per_split;
    ks_end = min(ksteps_total, ks_begin + a.ksteps_per_split);
  }
  const int Ktot = T * g.Cin;
  const f16* wbase = a.w + g.wofs[phase];

  // per-thread load coordinates
  const int lrow = tid / CPR, lchunk = tid % CPR;
  int a_pix[PA];       // (n*Hin) row base is folded below
  int a_iy0[PA], a_ix0[PA];
  bool a_ok[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    int m = m0 + lrow + i * RPP;
    a_ok[i] = m < a.M;
    int mm = a_ok[i] ? m : 0;
    int n_img = mm / (g.GH * g.GW);
    int rem = mm - n_img * (g.GH * g.GW);
    int gy = rem / g.GW, gx = rem - gy * g.GW;
    a_pix[i] = n_img * g.Hin;
    a_iy0[i] = gy * g.in_stride;
    a_ix0[i] = gx * g.in_stride;
  }
  const f16* b_ptr[PB];
#pragma unroll
  for (int i = 0; i < PB; ++i) b_ptr[i] = wbase + (size_t)(n0 + lrow + i * RPP) * Ktot + lchunk * 8;

  u32x4 ra[PA], rb[PB];
  auto gload = [&](int ks) __attribute__((always_inline)) {
    int t = ks / KCB;
    int c0 = (ks - t * KCB) * BK;
    int dy = g.dy[phase][t], dx = g.dx[phase][t];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      int iy = a_iy0[i] + dy, ix = a_ix0[i] + dx;
      bool ok = a_ok[i] && (unsigned)iy < (unsigned)g.Hin && (unsigned)ix < (unsigned)g.Win;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (ok) v = *reinterpret_cast<const u32x4*>(a.in + ((size_t)((a_pix[i] + iy) * g.Win + ix) * g.Cin + c0 + lchunk * 8));
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) rb[i] = *reinterpret_cast<const u32x4*>(b_ptr[i] + (size_t)ks * BK);
  };
  auto lstore = [&](int stage) __attribute__((always_inline)) {
    unsigned char* sa = smem + stage * STAGE;
    unsigned char* sb = sa + A_BYTES;
#pragma unroll
    for (int i = 0; i < PA; ++i) *reinterpret_cast<u32x4*>(sa + (lrow + i * RPP) * PITCH + lchunk * 16) = ra[i];
#pragma unroll
    for (int i = 0; i < PB; ++i) *reinterpret_cast<u32x4*>(sb + (lrow + i * RPP) * PITCH + lchunk * 16) = rb[i];
  };

  f32x4 acc[NI][MI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  if (ks_begin < ks_end) {
    gload(ks_begin);
    lstore(0);
    __syncthreads();
    int cur = 0;
    for (int ks = ks_begin; ks < ks_end; ++ks) {
      const bool more = ks + 1 < ks_end;
      if (more) gload(ks + 1);
      const unsigned char* sa = smem + cur * STAGE;
      const unsigned char* sb = sa + A_BYTES;
#pragma unroll
      for (int kk = 0; kk < BK / 32; ++kk) {
        f16x8 wf[NI], xf[MI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          wf[ni] = *reinterpret_cast<const f16x8*>(sb + (wn * WN + ni * 16 + fr) * PITCH + (kk * 4 + fq) * 16);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          xf[mi] = *reinterpret_cast<const f16x8*>(sa + (wm * WM + mi * 16 + fr) * PITCH + (kk * 4 + fq) * 16);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
      }
      if (more) lstore(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
  }

  // D^T block (ni, mi): lane holds pixel m = mi*16 + fr, channels n = ni*16 + fq*4 + {0..3}
  if (SPLITK) {
    float* slab = a.slab + (size_t)split * a.M * g.Cout;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      int m = m0 + wm * WM + mi * 16 + fr;
      if (m < a.M) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          int n = n0 + wn * WN + ni * 16 + fq * 4;
          *reinterpret_cast<f32x4*>(slab + (size_t)m * g.Cout + n) = acc[ni][mi];
        }
      }
    }
    return;
  }

  // ---- epilogue: bias, stage fp16 tile in LDS, then coalesced 16-byte stores (+ residual add, GN partials)
  unsigned char* so = smem;  // all waves are past the last barrier of the K loop
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    int nl = wn * WN + ni * 16 + fq * 4;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) bv = *reinterpret_cast<const f32x4*>(a.bias + n0 + nl);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      int ml = wm * WM + mi * 16 + fr;
      f32x4 v = acc[ni][mi] + bv;
      f16x4 h = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
      *reinterpret_cast<f16x4*>(so + ml * OPITCH + nl * 2) = h;
    }
  }
  __syncthreads();
  constexpr int OCPR = BN / 8, ORPP = 256 / OCPR, OP = BM / ORPP;
  const int orow = tid / OCPR, ochunk = tid % OCPR;
  const int G = g.Cout >> 3;          // channels per GroupNorm group (8 groups)
  float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
#pragma unroll
  for (int i = 0; i < OP; ++i) {
    int ml = orow + i * ORPP;
    int m = m0 + ml;
    if (m >= a.M) continue;
    f16x8 h = *reinterpret_cast<const f16x8*>(so + ml * OPITCH + ochunk * 16);
    int n_img = m / (g.GH * g.GW);
    int rem = m - n_img * (g.GH * g.GW);
    int gy = rem / g.GW, gx = rem - gy * g.GW;
    int oy = gy * g.out_stride + g.out_oy[phase], ox = gx * g.out_stride + g.out_ox[phase];
    size_t off = ((size_t)(n_img * g.Hout + oy) * g.Wout + ox) * g.Cout + n0 + ochunk * 8;
    if (a.add_src) {
      f16x8 r = *reinterpret_cast<const f16x8*>(a.add_src + off);
#pragma unroll
      for (int j = 0; j < 8; ++j) h[j] = (f16)((float)h[j] + (float)r[j]);
    }
    *reinterpret_cast<f16x8*>(a.out + off) = h;
    if (a.gn_partial) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { float x = (float)h[j]; s0 += x; q0 += x * x; }
#pragma unroll
      for (int j = 4; j < 8; ++j) { float x = (float)h[j]; s1 += x; q1 += x * x; }
    }
  }
  if (a.gn_partial) {
    // deterministic block reduction: per-thread (sum, sumsq) of the two 4-channel halves of its chunk
    float* red = reinterpret_cast<float*>(smem + BM * OPITCH);  // 256*4 floats
    red[tid * 4 + 0] = s0; red[tid * 4 + 1] = q0; red[tid * 4 + 2] = s1; red[tid * 4 + 3] = q1;
    __syncthreads();
    // groups covered by this tile: half-chunks of 4 channels -> group = (n0 + hc*4) / G
    const int ngroups = (BN >= G) ? BN / G : 1;
    if (tid < ngroups * 2) {
      int gl = tid >> 1, which = tid & 1;               // which: 0 = sum, 1 = sumsq
      int hc_begin = gl * G / 4, hc_end = (gl + 1) * G / 4;  // half-chunk range of this group inside the tile
      if (BN < G) { hc_begin = 0; hc_end = BN / 4; }
      float tot = 0.f;
      for (int r = 0; r < ORPP; ++r)
        for (int hc = hc_begin; hc < hc_end; ++hc) {
          int t = r * OCPR + (hc >> 1);
          tot += red[t * 4 + (hc & 1) * 2 + which];
        }
      // partial index: [n_img][mtile_in_sample (phase-major)][group][2]
      int per_sample = g.GH * g.GW;
      int n_img = m0 / per_sample;
      int mt = (m0 - n_img * per_sample) / BM;
      int MT = (per_sample / BM) * g.n_phase;
      int grp = (n0 / G) + gl;
      a.gn_partial[(((size_t)n_img * MT + phase * (per_sample / BM) + mt) * 8 + grp) * 2 + which] = tot;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// split-K slab reduction: out[m][n] = bias[n] + sum_s slab[s][m][n]   (fp32 and/or fp16 outputs)
// ---------------------------------------------------------------------------------------------
__global__ void lo_splitk_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ bias,
                                        float* __restrict__ out32, f16* __restrict__ out16, int M, int N, int nsplit) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * N) return;
  int n = i % N;
  float v = bias ? bias[n] : 0.f;
  for (int s = 0; s < nsplit; ++s) v += slab[(size_t)s * M * N + i];
  if (out32) out32[i] = v;
  if (out16) out16[i] = (f16)v;
}

// ---------------------------------------------------------------------------------------------
// TN weight-gradient GEMM:  dWp[p][n][t*Cin + c] = sum_m dy[m][n] * Xtap[m][c]
//   dy rows are the op's OUTPUT pixels (forward geometry), Xtap rows the input pixels a tap reads.
//   Tile: BMW output channels (n) x BNW input channels (c) for one (phase, tap); K loop over pixels in
//   steps of 32; grid.z splits the pixel range; fp32 partial slabs, reduced by lo_wgrad_reduce_kernel.
// ---------------------------------------------------------------------------------------------
struct WgradArgs {
  const f16* x;     // forward input  [B,Hin,Win,Cin]
  const f16* dy;    // grad of forward output [B,Hout,Wout,Cout]
  float* slab;      // [nsplit][packed elems]
  float* grad;      // canonical fp32 gradient, written directly when nsplit == 1 (direct mode)
  float scale;
  int direct;
  int M;            // pixels per phase = B*GH*GW
  int nsplit;
  int msteps_per_split;  // 32-pixel steps per split
  int packed_elems;
  LoGeom g;         // FORWARD geometry
};

template <int BMW, int BNW>
__global__ __launch_bounds__(256) void lo_wgrad_tn(WgradArgs a) {
  constexpr int BKP = 32;                        // pixels per K step
  constexpr int PITCH_A = BMW * 2 + 32, PITCH_B = BNW * 2 + 32;  // bytes; +32: conflict-free tr reads
  constexpr int A_BYTES = BKP * PITCH_A, B_BYTES = BKP * PITCH_B;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int CPA = BMW / 8, CPB = BNW / 8;    // 16-byte chunks per row
  constexpr int NA = (BKP * CPA + 255) / 256, NB = (BKP * CPB + 255) / 256;  // loads per thread
  constexpr int WM = BMW / 2, WN = BNW / 2, MI = WM / 16, NI = WN / 16;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

  const LoGeom& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int ntile_n = g.Cout / BMW;
  const int tn = blockIdx.x % ntile_n, tc = blockIdx.x / ntile_n;
  const int n0 = tn * BMW, c0 = tc * BNW;
  // blockIdx.y enumerates (phase, tap)
  int phase = 0, t = blockIdx.y;
  while (t >= g.T[phase]) { t -= g.T[phase]; ++phase; }
  const int dyo = g.dy[phase][t], dxo = g.dx[phase][t];
  const int split = blockIdx.z;
  const int ms_begin = split * a.msteps_per_split;
  const int ms_total = (a.M + BKP - 1) / BKP;
  const int ms_end = min(ms_total, ms_begin + a.msteps_per_split);

  u32x4 ra[NA], rb[NB];
  auto gload = [&](int ms) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int idx = tid + i * 256;
      int row = idx / CPA, ch = idx % CPA;
      int m = ms * BKP + row;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (m < a.M && idx < BKP * CPA) {
        int n_img = m / (g.GH * g.GW);
        int rem = m - n_img * (g.GH * g.GW);
        int gy = rem / g.GW, gx = rem - gy * g.GW;
        int oy = gy * g.out_stride + g.out_oy[phase], ox = gx * g.out_stride + g.out_ox[phase];
        v = *reinterpret_cast<const u32x4*>(a.dy + ((size_t)(n_img * g.Hout + oy) * g.Wout + ox) * g.Cout + n0 + ch * 8);
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      int idx = tid + i * 256;
      int row = idx / CPB, ch = idx % CPB;
      int m = ms * BKP + row;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (m < a.M && idx < BKP * CPB) {
        int n_img = m / (g.GH * g.GW);
        int rem = m - n_img * (g.GH * g.GW);
        int gy = rem / g.GW, gx = rem - gy * g.GW;
        int iy = gy * g.in_stride + dyo, ix = gx * g.in_stride + dxo;
        if ((unsigned)iy < (unsigned)g.Hin && (unsigned)ix < (unsigned)g.Win)
          v = *reinterpret_cast<const u32x4*>(a.x + ((size_t)(n_img * g.Hin + iy) * g.Win + ix) * g.Cin + c0 + ch * 8);
      }
      rb[i] = v;
    }
  };
  auto lstore = [&](int stage) __attribute__((always_inline)) {
    unsigned char* sa = smem + stage * STAGE;
    unsigned char* sb = sa + A_BYTES;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int idx = tid + i * 256;
      if (idx < BKP * CPA) *reinterpret_cast<u32x4*>(sa + (idx / CPA) * PITCH_A + (idx % CPA) * 16) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      int idx = tid + i * 256;
      if (idx < BKP * CPB) *reinterpret_cast<u32x4*>(sb + (idx / CPB) * PITCH_B + (idx % CPB) * 16) = rb[i];
    }
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // transposed fragment reads.  The 32 pixel rows of a K step are assigned to MFMA k positions by the
  // SAME permutation for both operands (lane group q reads rows 4q..4q+3 and 16+4q..16+4q+3), which keeps
  // every ds_read_b64_tr_b16 conflict-free at this pitch; any consistent k permutation leaves the sum unchanged.
  const int q16 = lane >> 4, i16 = lane & 15;
  const int trow = 4 * q16 + (i16 >> 2);   // row supplied by this lane (first read); +16 for the second
  const int tcol = (i16 & 3) * 4;          // first of the 4 columns this lane addresses
  if (ms_begin < ms_end) {
    gload(ms_begin);
    lstore(0);
    __syncthreads();
    int cur = 0;
    for (int ms = ms_begin; ms < ms_end; ++ms) {
      const bool more = ms + 1 < ms_end;
      if (more) gload(ms + 1);
      const unsigned char* sa = smem + cur * STAGE;
      const unsigned char* sb = sa + A_BYTES;
      f16x8 af[MI], bf[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        int col = wm * WM + mi * 16 + tcol;
        h16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
            (h16x4 __attribute__((address_space(3)))*)(sa + trow * PITCH_A + col * 2));
        h16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
            (h16x4 __attribute__((address_space(3)))*)(sa + (trow + 16) * PITCH_A + col * 2));
        af[mi] = (f16x8){(f16)lo[0], (f16)lo[1], (f16)lo[2], (f16)lo[3], (f16)hi[0], (f16)hi[1], (f16)hi[2], (f16)hi[3]};
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        int col = wn * WN + ni * 16 + tcol;
        h16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
            (h16x4 __attribute__((address_space(3)))*)(sb + trow * PITCH_B + col * 2));
        h16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
            (h16x4 __attribute__((address_space(3)))*)(sb + (trow + 16) * PITCH_B + col * 2));
        bf[ni] = (f16x8){(f16)lo[0], (f16)lo[1], (f16)lo[2], (f16)lo[3], (f16)hi[0], (f16)hi[1], (f16)hi[2], (f16)hi[3]};
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
      if (more) lstore(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
  }
  // D[n][c] block (mi, ni): lane holds column c = ni*16 + (lane&15), rows n = mi*16 + (lane>>4)*4 + j
  float* slab = a.slab + (size_t)split * a.packed_elems + g.wofs[phase];
  const int Ktot = g.T[phase] * g.Cin;
  const int rs = g.rs[phase][t];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      int c = c0 + wn * WN + ni * 16 + (lane & 15);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int n = n0 + wm * WM + mi * 16 + (lane >> 4) * 4 + j;
        if (a.direct) a.grad[(size_t)n * g.sn + (size_t)c * g.sc + rs] = acc[mi][ni][j] * a.scale;
        else slab[(size_t)n * Ktot + t * g.Cin + c] = acc[mi][ni][j];
      }
    }
}

// sum the split slabs and scatter to the canonical fp32 gradient:  grad[n*sn + c*sc + rs] = scale * sum_s slab
__global__ void lo_wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ grad, LoGeom g,
                                       int total, int nsplit, float scale) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int p = 0;
#pragma unroll
  for (int q = 1; q < LO_MAX_PHASE; ++q) if (q < g.n_phase && i >= g.wofs[q]) p = q;
  int j = i - g.wofs[p];
  int K = g.T[p] * g.Cin;
  int n = j / K, k = j - n * K;
  int t = k / g.Cin, c = k - t * g.Cin;
  float v = 0.f;
  for (int s = 0; s < nsplit; ++s) v += slab[(size_t)s * total + i];
  grad[(size_t)n * g.sn + (size_t)c * g.sc + g.rs[p][t]] = v * scale;
}

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
int lo_pack_weight(const float* w, f16* wp, const LoGeom& g, hipStream_t st) {
  int total = geom_packed_elems(g);
  hipLaunchKernelGGL(lo_pack_weight_kernel, dim3((total + 255) / 256), dim3(256), 0, st, w, wp, g, total);
  LO_LAUNCH_CHECK("pack_weight");
  return LO_OK;
}

template <int BM, int BN, int BK>
static int launch_igemm(const IgemmArgs& a, hipStream_t st) {
  const LoGeom& g = a.g;
  dim3 grid((a.M + BM - 1) / BM, g.Cout / BN, a.nsplit > 1 ? a.nsplit : g.n_phase);
  if (a.nsplit > 1)
    hipLaunchKernelGGL((lo_igemm_nt<BM, BN, BK, true>), grid, dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((lo_igemm_nt<BM, BN, BK, false>), grid, dim3(256), 0, st, a);
  LO_LAUNCH_CHECK("igemm");
  return LO_OK;
}

// Run one conv-like op.  `slab` + nsplit > 1 selects split-K (output = fp32 partials, caller reduces).
int lo_conv_run(const LoGeom& g, const f16* in, const f16* wp, const float* bias, const f16* add_src, f16* out,
                float* gn_partial, float* slab, int nsplit, hipStream_t st) {
  IgemmArgs a;
  a.in = in; a.w = wp; a.bias = bias; a.add_src = add_src; a.out = out; a.gn_partial = gn_partial; a.slab = slab;
  a.g = g;
  a.M = g.B * g.GH * g.GW;
  a.nsplit = nsplit < 1 ? 1 : nsplit;
  LO_REQUIRE(g.Cin % 32 == 0, "lo_conv_run: Cin=%d must be a multiple of 32", g.Cin);
  LO_REQUIRE(g.Cout % 32 == 0, "lo_conv_run: Cout=%d must be a multiple of 32", g.Cout);
  const int BK = (g.Cin % 64 == 0) ? 64 : 32;
  const int per_sample = g.GH * g.GW;
  int ksteps = 0;
  for (int p = 0; p < g.n_phase; ++p) ksteps = g.T[p] * (g.Cin / BK) > ksteps ? g.T[p] * (g.Cin / BK) : ksteps;
  a.ksteps_per_split = (ksteps + a.nsplit - 1) / a.nsplit;
  if (a.nsplit > 1) {
    LO_REQUIRE(g.n_phase == 1 && slab, "lo_conv_run: split-K needs a single phase and a slab");
    LO_REQUIRE(BK == 64 && g.Cout % 64 == 0, "lo_conv_run: split-K path needs Cin%%64==0 and Cout%%64==0");
    return launch_igemm<64, 64, 64>(a, st);
  }
  // tile choice: BM=128 when a sample's grid holds whole tiles and the launch still fills the chip
  int bm = (per_sample % 128 == 0 && (size_t)a.M * g.Cout * g.n_phase >= (size_t)128 * 128 * 384) ? 128 : 64;
  if (gn_partial) {
    LO_REQUIRE(per_sample % bm == 0, "lo_conv_run: GN partials need whole tiles per sample");
    LO_REQUIRE(a.M % bm == 0, "lo_conv_run: M %% BM");
  }
  if (BK == 32) {
    LO_REQUIRE(g.Cout % 64 == 0, "lo_conv_run: BK=32 path needs Cout%%64==0");
    if (bm == 128) return launch_igemm<128, 64, 32>(a, st);
    return launch_igemm<64, 64, 32>(a, st);
  }
  if (g.Cout % 128 == 0 && bm == 128) return launch_igemm<128, 128, 64>(a, st);
  if (g.Cout % 64 == 0) {
    // GN groups wider than the N tile are not supported by the fused partials
    if (gn_partial) LO_REQUIRE((g.Cout >> 3) <= 64, "lo_conv_run: group wider than tile");
    if (bm == 128) return launch_igemm<128, 64, 64>(a, st);
    return launch_igemm<64, 64, 64>(a, st);
  }
  if (bm == 128) return launch_igemm<128, 32, 64>(a, st);
  return launch_igemm<64, 32, 64>(a, st);
}

int lo_conv_tile_m(const LoGeom& g) {
  const int per_sample = g.GH * g.GW;
  size_t M = (size_t)g.B * per_sample;
  return (per_sample % 128 == 0 && M * g.Cout * g.n_phase >= (size_t)128 * 128 * 384) ? 128 : 64;
}

int lo_splitk_reduce(const float* slab, const float* bias, float* out32, f16* out16, int M, int N, int nsplit,
                     hipStream_t st) {
  int total = M * N;
  hipLaunchKernelGGL(lo_splitk_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, st, slab, bias, out32, out16, M, N, nsplit);
  LO_LAUNCH_CHECK("splitk_reduce");
  return LO_OK;
}

// number of pixel splits the wgrad launcher will use for this geometry (callers size the slab with it)
int lo_wgrad_nsplit(const LoGeom& g) {
  int bmw = g.Cout % 128 == 0 ? 128 : (g.Cout % 64 == 0 ? 64 : 32);
  int bnw = g.Cin % 128 == 0 ? 128 : (g.Cin % 64 == 0 ? 64 : 32);
  int taps = 0;
  for (int p = 0; p < g.n_phase; ++p) taps += g.T[p];
  long tiles = (long)(g.Cout / bmw) * (g.Cin / bnw) * taps;
  int M = g.B * g.GH * g.GW;
  int ms_total = (M + 31) / 32;
  long want = (1024 + tiles - 1) / tiles;  // aim at >= 1024 workgroups
  if (want < 1) want = 1;
  if (want > ms_total / 4) want = ms_total / 4 > 0 ? ms_total / 4 : 1;   // at least 4 K steps per split
  if (want > 256) want = 256;
  return (int)want;
}

int lo_wgrad_run(const LoGeom& g, const f16* x, const f16* dy, float* slab, float* grad, float scale, hipStream_t st) {
  WgradArgs a;
  a.x = x; a.dy = dy; a.slab = slab; a.g = g; a.grad = grad; a.scale = scale;
  a.M = g.B * g.GH * g.GW;
  a.packed_elems = geom_packed_elems(g);
  a.nsplit = lo_wgrad_nsplit(g);
  a.direct = (a.nsplit == 1 && g.sc == 1) ? 1 : 0;
  int ms_total = (a.M + 31) / 32;
  a.msteps_per_split = (ms_total + a.nsplit - 1) / a.nsplit;
  LO_REQUIRE(g.Cin % 32 == 0 && g.Cout % 32 == 0, "lo_wgrad_run: channels must be multiples of 32 (Cin=%d Cout=%d)", g.Cin, g.Cout);
  int taps = 0;
  for (int p = 0; p < g.n_phase; ++p) taps += g.T[p];
  int bmw = g.Cout % 128 == 0 ? 128 : (g.Cout % 64 == 0 ? 64 : 32);
  int bnw = g.Cin % 128 == 0 ? 128 : (g.Cin % 64 == 0 ? 64 : 32);
  dim3 grid((g.Cout / bmw) * (g.Cin / bnw), taps, a.nsplit);
#define LO_WG(BMW, BNW) hipLaunchKernelGGL((lo_wgrad_tn<BMW, BNW>), grid, dim3(256), 0, st, a)
  if (bmw == 128 && bnw == 128) LO_WG(128, 128);
  else if (bmw == 128 && bnw == 64) LO_WG(128, 64);
  else if (bmw == 64 && bnw == 128) LO_WG(64, 128);
  else if (bmw == 64 && bnw == 64) LO_WG(64, 64);
  else if (bmw == 32 && bnw == 64) LO_WG(32, 64);
  else if (bmw == 64 && bnw == 32) LO_WG(64, 32);
  else if (bmw == 128 && bnw == 32) LO_WG(128, 32);
  else if (bmw == 32 && bnw == 128) LO_WG(32, 128);
  else { lo_set_error("lo_wgrad_run: unsupported tile %dx%d", bmw, bnw); return LO_ERR_ARG; }
#undef LO_WG
  LO_LAUNCH_CHECK("wgrad_tn");
  if (a.direct) return LO_OK;
  int total = a.packed_elems;
  hipLaunchKernelGGL(lo_wgrad_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, st, slab, grad, g, total, a.nsplit, scale);
  LO_LAUNCH_CHECK("wgrad_reduce");
  return LO_OK;
}

size_t lo_wgrad_slab_bytes(const LoGeom& g) { return (size_t)lo_wgrad_nsplit(g) * geom_packed_elems(g) * sizeof(float); }
size_t lo_packed_weight_elems(const LoGeom& g) { return (size_t)geom_packed_elems(g); }
