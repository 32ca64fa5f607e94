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
#include "lo_internal.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

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
  if ((g->GH & (g->GH - 1)) || (g->GW & (g->GW - 1)) || g->GH < 1 || g->GW < 1) {
    lo_set_error("lo_make_geom: output grid %dx%d must be powers of two", g->GH, g->GW);
    return LO_ERR_ARG;
  }
  while ((1 << g->lgh) < g->GH) ++g->lgh;
  while ((1 << g->lgw) < g->GW) ++g->lgw;
  auto lg2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; };
  g->lg_hin = lg2(g->Hin); g->lg_win = lg2(g->Win); g->lg_cin = lg2(g->Cin);
  g->lg_hout = lg2(g->Hout); g->lg_wout = lg2(g->Wout); g->lg_cout = lg2(g->Cout);
  for (int p = 0; p < g->n_phase; ++p)
    for (int t = 0; t < g->T[p]; ++t) {
      g->dyc[p] |= (uint32_t)(g->dy[p][t] + 1) << (2 * t);
      g->dxc[p] |= (uint32_t)(g->dx[p][t] + 1) << (2 * t);
    }
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

// all packs of a model in ONE launch: jobs live in device memory (uploaded once per workspace by the executor).
// A block owns 16 output channels x 64 reduced channels x all taps: the canonical weight is read along its contiguous
// axis (the 9 or 16 taps of one (n, c) pair are adjacent floats; the pairs themselves are adjacent along c for forward
// convs / transposed-conv gradients and along n for the others), staged as fp16 in LDS, and written along c, the
// contiguous axis of the packed operand.  lo_pack_blocks() gives the block count of one job.
int lo_pack_blocks(const LoGeom& g) { return ((g.Cout + 15) / 16) * ((g.Cin + 63) / 64); }
__global__ __launch_bounds__(256) void lo_pack_all_kernel(const LoPackJob* __restrict__ jobs, int njobs, int block_base) {
  __shared__ __attribute__((aligned(16))) f16 tile[16][17][72];     // rows 144 bytes apart: 16-byte reads of eight consecutive c
  const int bid = (int)blockIdx.x + block_base;     // block_base: first block of a sub-range of the job table
  int j = 0;
  while (j + 1 < njobs && bid >= jobs[j + 1].block0) ++j;
  const LoPackJob& J = jobs[j];
  const LoGeom& g = J.g;
  const int tid = threadIdx.x;
  const int tiles_c = (g.Cin + 63) / 64;
  const int b = bid - J.block0;
  const int n0 = (b / tiles_c) * 16, c0 = (b % tiles_c) * 64;
  const bool n_fast = g.sn < g.sc;
  const int t_all = n_fast ? g.sn : g.sc;     // taps of the canonical weight (9, 16; 1 for a Linear)
  for (int q = tid; q < 1024; q += 256) {
    int nl, cl;
    if (n_fast) { nl = q & 15; cl = q >> 4; } else { cl = q & 63; nl = q >> 6; }
    const int n = n0 + nl, c = c0 + cl;
    if (n < g.Cout && c < g.Cin) {
      const float* src = J.src + (size_t)n * g.sn + (size_t)c * g.sc;
      for (int t = 0; t < t_all; ++t) tile[nl][t][cl] = (f16)src[t];
    }
  }
  __syncthreads();
  int sum_t = 0;
  for (int p = 0; p < g.n_phase; ++p) sum_t += g.T[p];
  // 16-byte stores of eight consecutive c (every Cin of the model is a multiple of 32): the 2-byte stores of the first form were
  // 65 us per step for the 15 M packed elements of the model (rocprofv3, round 4)
  const int c8 = (tid & 7) * 8, c = c0 + c8;
  for (int r = tid >> 3; r < 16 * sum_t; r += 32) {
    const int nl = r / sum_t;
    int t = r - nl * sum_t, p = 0;
    while (t >= g.T[p]) { t -= g.T[p]; ++p; }
    const int n = n0 + nl;
    if (n < g.Cout && c + 8 <= g.Cin)
      *reinterpret_cast<f16x8*>(J.dst + (size_t)g.wofs[p] + ((size_t)n * g.T[p] + t) * g.Cin + c) = *reinterpret_cast<const f16x8*>(&tile[nl][g.rs[p][t]][c8]);
    else if (n < g.Cout)
      for (int e = 0; e < 8 && c + e < g.Cin; ++e) J.dst[(size_t)g.wofs[p] + ((size_t)n * g.T[p] + t) * g.Cin + c + e] = tile[nl][g.rs[p][t]][c8 + e];
  }
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
  // fused GroupNorm-backward reduction (data-gradient ops): the output of this op is dL/da of a conv+GN+Mish layer
  // whose raw conv output is gb_v; the epilogue also emits P1[n][mtile][c] = (sum du, sum du*xhat), du = da*mish'(u)
  // teacher path: LeakyReLU(0.2) on (conv + bias) and per-channel BatchNorm partial sums of the stored values
  int act;                 // 0 none, 1 LeakyReLU(0.2)
  float* bn_partial;       // [M tiles][Cout][2] (sum, sumsq) or null
  const f16* gb_v;
  const float* gb_stats;   // [B][8][2] mean, rstd
  const float* gb_gamma;
  const float* gb_beta;
  float* gb_P1;
  f16* gb_dv;              // != null: the GroupNorm-backward apply runs here too (LoGnBwdFuse in lo_internal.h)
  float* gb_P2;
  unsigned int* gb_counter;
  unsigned int gb_target;
  unsigned int* gb_fail;
  int gb_keep_out;
  const float* f8_scale;   // fp8 operand path: [n_phase][Cout] dequantisation factor (weight row scale / activation scale)
  int out_pitch, out_choff;   // out_pitch > 0: `out` has out_pitch channels per pixel, this op's channels start at out_choff
  LoGnFuse gf;         // gf.y != null: GroupNorm + Mish of this output in the epilogue (sample rendezvous, lo_common.h)
  int M;               // rows per phase = B*GH*GW
  int nsplit;          // >= 1
  int ksteps_per_split;
  LoGeom g;
};

// 16 zero bytes x 16: source of every LDS-DMA lane whose row is padding / out of range
__device__ __attribute__((aligned(256))) unsigned int lo_zero_page[64];

#define LO_VMCNT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")

// Workgroups are dealt round-robin over the 8 XCDs (each with a private 4 MiB L2).  Map the linear block id so that
// every XCD works on ONE contiguous range of logical tile ids: tiles that re-read the same activations (the taps of a
// pixel tile, the N tiles of an M tile) then hit in that XCD's L2 instead of the Infinity Cache.  Bijective for any
// total (speed only, never correctness).
__device__ __forceinline__ int lo_xcd_remap(int bid, int total) {
  const int q = total >> 3, r = total & 7;
  const int xcd = bid & 7, loc = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

// Operand tiles go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4): no staging VGPRs and no ds_write pass (the
// ds_write_b128 path sustains only ~80 B/clk/CU, which made the register-staged version LDS-bound).  One
// wave-instruction fills 1 KiB of LDS linearly, so tiles are unpadded [rows][BK] and bank conflicts are removed by an
// XOR swizzle applied on the per-lane SOURCE chunk and again on the fragment read: chunk' = chunk ^ ((row >> 1) & (CPR-1)).
// NSTAGE LDS stages, NSTAGE-1 K steps in flight behind a counted s_waitcnt vmcnt + one raw s_barrier per K step.
// F8: both operands are OCP e4m3 bytes (in / w point at bytes, every element offset below is scaled by ES), BK = 128 elements
// so a tile row is the same 128 bytes as the fp16 BK = 64 row, and one K step is ONE v_mfma_scale_f32_16x16x128_f8f6f4 per
// 16x16 block (lane holds row lane&15 and 32 bytes of k: chunks fq and fq + 4 of the row; unit block scales) - twice the
// K per byte moved and per MFMA cycle.  The epilogue multiplies by f8_scale[phase][n] before the bias.
template <int BM, int BN, int BK, int NSTAGE, bool F8>
constexpr int igemm_lds_bytes() {   // the K-loop ring, or the epilogue's staging tile + reduction scratch, whichever is larger
  constexpr int ring = NSTAGE * (BM + BN) * BK * (F8 ? 1 : 2);
  constexpr int epi = BM * (BN * 2 + 16) + 16384 + 1024;
  return ring > epi ? ring : epi;
}

template <int BM, int BN, int BK, int NSTAGE, bool SPLITK, bool F8 = false>
__global__ __launch_bounds__(256) void lo_igemm_nt(IgemmArgs a) {
  constexpr int ES = F8 ? 1 : 2;          // bytes per operand element
  constexpr int CE = 16 / ES;             // elements per 16-byte chunk
  constexpr int CPR = BK / CE;            // 16-byte chunks per tile row
  constexpr int RPI = 64 / CPR;           // tile rows filled by one wave-instruction
  constexpr int ROWB = BK * ES;           // bytes per tile row
  static_assert(!(F8 && SPLITK) && (!F8 || BK == 128), "fp8 path: BK = 128, no split-K");
  constexpr int IA = BM / RPI / 4, IB = BN / RPI / 4;   // LDS-DMA instructions per wave and K step (A, B)
  static_assert(IA >= 1 && IB >= 1 && IA * RPI * 4 == BM && IB * RPI * 4 == BN, "tile / wave-instruction mismatch");
  constexpr int LPT = IA + IB;
  constexpr int D = NSTAGE - 1;           // K steps in flight
  static_assert(LPT * D <= 63, "vmcnt range");
  constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int WM = BM / 2, WN = BN / 2; // wave tile (2 x 2 waves)
  constexpr int MI = WM / 16, NI = WN / 16;
  constexpr int OPITCH = BN * 2 + 16;     // epilogue staging pitch (bytes)
  // dynamic LDS (igemm_lds_bytes<...>() bytes, passed by the launcher): the deep-pipeline instantiations exceed the 64 KB a static
  // array may have; it is the kernel's only LDS object, so it starts at offset 0 (1 KB-aligned DMA destinations)
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

  const LoGeom& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  // logical tile id -> (n tile fastest, then m tile, then phase / K split)
  const int NT = g.Cout / BN, MT = (a.M + BM - 1) / BM;
  const int tile_id = lo_xcd_remap(blockIdx.x, gridDim.x);
  int nt_i = tile_id % NT, mt_i = (tile_id / NT) % MT, z_i = tile_id / (NT * MT);
  if (!SPLITK && a.gf.y) {
    // fused GroupNorm: the tiles of one sample wait for each other, so they get adjacent ids (sample, phase, m tile, n tile) --
    // the default order puts the phase outermost, i.e. a sample's tiles a whole grid quarter apart
    const int rows_ps = (g.GH * g.GW) / BM, tiles_ps = rows_ps * g.n_phase * NT;
    const int n_s = tile_id / tiles_ps, r = tile_id - n_s * tiles_ps;
    const int row = r / NT;
    nt_i = r - row * NT;
    z_i = row / rows_ps;
    mt_i = n_s * rows_ps + (row - z_i * rows_ps);
  }
  const int m0 = mt_i * BM, n0 = nt_i * BN;
  const int phase = SPLITK ? 0 : z_i;
  const int split = SPLITK ? z_i : 0;
  const int T = g.T[phase];
  const int KCB = g.Cin / BK;
  const int ksteps_total = T * KCB;
  int ks_begin = 0, ks_end = ksteps_total;
  if (SPLITK) {
    ks_begin = split * a.ksteps_per_split;
    ks_end = min(ksteps_total, ks_begin + a.ksteps_per_split);
  }
  const int nk = ks_end - ks_begin;
  const int Ktot = T * g.Cin;
  const unsigned char* inb = reinterpret_cast<const unsigned char*>(a.in);
  const unsigned char* wbase = reinterpret_cast<const unsigned char*>(a.w) + (size_t)g.wofs[phase] * ES;
  const unsigned char* zpage = reinterpret_cast<const unsigned char*>(lo_zero_page);
  const uint32_t dyc = g.dyc[phase], dxc = g.dxc[phase];   // tap offsets in registers: no memory load inside the K loop

  // per-lane source coordinates of the LDS-DMA instructions this wave issues.  Everything that depends only on the
  // tap is recomputed when the tap changes; a K step then costs one add + one select per instruction.
  const int lrow = lane / CPR, lpos = lane % CPR;
  int a_base[IA], a_iy0[IA], a_ix0[IA];   // element offset of (n, iy0, ix0, chunk); -1 = row out of range
#pragma unroll
  for (int i = 0; i < IA; ++i) {
    int row = (wave * IA + i) * RPI + lrow;
    int m = m0 + row;
    bool ok = m < a.M;
    int mm = ok ? m : 0;
    int gx = mm & (g.GW - 1), gy = (mm >> g.lgw) & (g.GH - 1), n_img = mm >> (g.lgw + g.lgh);
    a_iy0[i] = ok ? gy * g.in_stride : -100000;       // makes every tap invalid for an out-of-range row
    a_ix0[i] = gx * g.in_stride;
    a_base[i] = ((n_img * g.Hin + gy * g.in_stride) * g.Win + gx * g.in_stride) * g.Cin +
                (lpos ^ ((row >> 1) & (CPR - 1))) * CE;  // logical chunk stored at this lane's LDS slot
  }
  int b_base[IB];
#pragma unroll
  for (int i = 0; i < IB; ++i) {
    int row = (wave * IB + i) * RPI + lrow;
    b_base[i] = (n0 + row) * Ktot + (lpos ^ ((row >> 1) & (CPR - 1))) * CE;
  }
  // issue-side K position (tap, channel block) and the per-tap source offsets
  int i_t = ks_begin / KCB, i_cb = ks_begin - i_t * KCB, i_ks = ks_begin;
  int a_tap[IA];
  auto set_tap = [&](int t) __attribute__((always_inline)) {
    int dy = (int)((dyc >> (2 * t)) & 3u) - 1, dx = (int)((dxc >> (2 * t)) & 3u) - 1;
    int delta = (dy * g.Win + dx) * g.Cin;
#pragma unroll
    for (int i = 0; i < IA; ++i) {
      bool ok = (unsigned)(a_iy0[i] + dy) < (unsigned)g.Hin && (unsigned)(a_ix0[i] + dx) < (unsigned)g.Win;
      a_tap[i] = ok ? a_base[i] + delta : -1;
    }
  };
  set_tap(i_t < T ? i_t : 0);

  auto issue = [&](int stage) __attribute__((always_inline)) {
    unsigned char* sa = smem + stage * STAGE;
    unsigned char* sb = sa + A_BYTES;
    const bool live = i_ks < ks_end;
    const int coff = i_cb * BK;
#pragma unroll
    for (int i = 0; i < IA; ++i) {
      const unsigned char* src = (live && a_tap[i] >= 0) ? inb + (size_t)(a_tap[i] + coff) * ES : zpage;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sa + (wave * IA + i) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      const unsigned char* src = live ? wbase + (size_t)(b_base[i] + i_ks * BK) * ES : zpage;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sb + (wave * IB + i) * 1024), 16, 0, 0);
    }
    ++i_ks;
    if (++i_cb == KCB) {
      i_cb = 0;
      ++i_t;
      if (i_t < T) set_tap(i_t);
    }
  };

  f32x4 acc[NI][MI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  // fragment read offsets (bytes inside a stage), swizzled like the DMA sources
  constexpr int NKK = F8 ? BK / 128 : BK / 32;   // MFMA k sub-steps per K step
  constexpr int CPF = F8 ? 2 : 1;                // 16-byte chunks per lane fragment
  int xoff[MI][NKK][CPF], woff[NI][NKK][CPF];
#pragma unroll
  for (int kk = 0; kk < NKK; ++kk)
#pragma unroll
    for (int cf = 0; cf < CPF; ++cf) {
      // F8: the lane's 32-byte fragment = chunks (fq, fq + 4) of the 128-byte row, the same two reads as the fp16 k sub-steps (the
      // adjacent pair (2 fq, 2 fq + 1) would touch only every other 16-byte slot per read: 2-way bank conflicts); both operands
      // go through this chunk -> k map, so the product is unchanged
      const int chunk = F8 ? kk * 8 + cf * 4 + fq : kk * 4 + fq;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        int R = wm * WM + mi * 16 + fr;
        xoff[mi][kk][cf] = R * ROWB + ((chunk ^ ((R >> 1) & (CPR - 1))) * 16);
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        int R = wn * WN + ni * 16 + fr;
        woff[ni][kk][cf] = A_BYTES + R * ROWB + ((chunk ^ ((R >> 1) & (CPR - 1))) * 16);
      }
    }

  if (nk > 0) {
#pragma unroll
    for (int s = 0; s < D; ++s) issue(s);
    int rs = 0;            // stage being read
    int ws = D % NSTAGE;   // stage being refilled
    for (int it = 0; it < nk; ++it) {
      LO_VMCNT(LPT * (D - 1));          // this wave's DMA for K step `it` has landed ...
      __builtin_amdgcn_s_barrier();     // ... and so has every other wave's; stage `ws` is no longer being read
      issue(ws);
      const unsigned char* sbase = smem + rs * STAGE;
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) {
        if constexpr (F8) {
          i32x8 wf[NI], xf[MI];
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            const i32x4 lo = *reinterpret_cast<const i32x4*>(sbase + woff[ni][kk][0]), hi = *reinterpret_cast<const i32x4*>(sbase + woff[ni][kk][1]);
            wf[ni] = (i32x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          }
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            const i32x4 lo = *reinterpret_cast<const i32x4*>(sbase + xoff[mi][kk][0]), hi = *reinterpret_cast<const i32x4*>(sbase + xoff[mi][kk][1]);
            xf[mi] = (i32x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          }
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)   // formats 0 / 0 = e4m3 x e4m3; block scales 0x7f = 2^0
              acc[ni][mi] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        } else {
          f16x8 wf[NI], xf[MI];
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) wf[ni] = *reinterpret_cast<const f16x8*>(sbase + woff[ni][kk][0]);
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) xf[mi] = *reinterpret_cast<const f16x8*>(sbase + xoff[mi][kk][0]);
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
              acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
        }
      }
      rs = (rs + 1 == NSTAGE) ? 0 : rs + 1;
      ws = (ws + 1 == NSTAGE) ? 0 : ws + 1;
    }
    LO_VMCNT(0);                        // drain the (dummy) tail DMAs before LDS is reused by the epilogue
  }
  __syncthreads();

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
    f32x4 sv = {1.f, 1.f, 1.f, 1.f};
    if constexpr (F8) sv = *reinterpret_cast<const f32x4*>(a.f8_scale + (size_t)phase * g.Cout + n0 + nl);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      int ml = wm * WM + mi * 16 + fr;
      f32x4 v = F8 ? acc[ni][mi] * sv + bv : acc[ni][mi] + bv;
      f16x4 h = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
      *reinterpret_cast<f16x4*>(so + ml * OPITCH + nl * 2) = h;
    }
  }
  __syncthreads();
  constexpr int OCPR = BN / 8, ORPP = 256 / OCPR, OP = BM / ORPP;
  const int orow = tid / OCPR, ochunk = tid % OCPR;
  const int G = g.Cout >> 3;          // channels per GroupNorm group (8 groups)
  float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
  // fused GN-backward reduction: per-thread constants of its 8 channels (the tile lies inside one sample)
  float gsc[8], gsh[8], ga1[8], ga2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { ga1[j] = 0.f; ga2[j] = 0.f; }
  if (a.gb_v) {
    const int n_img_t = m0 >> (g.lgw + g.lgh);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int c = n0 + ochunk * 8 + j;
      int grp = c / G;
      float mean = a.gb_stats[n_img_t * 16 + grp * 2], rstd = a.gb_stats[n_img_t * 16 + grp * 2 + 1];
      gsc[j] = a.gb_gamma[c] * rstd;
      gsh[j] = a.gb_beta[c] - mean * gsc[j];
    }
  }
#pragma unroll
  for (int i = 0; i < OP; ++i) {
    int ml = orow + i * ORPP;
    int m = m0 + ml;
    if (m >= a.M) continue;
    f16x8 h = *reinterpret_cast<const f16x8*>(so + ml * OPITCH + ochunk * 16);
    // the output grid is a power of two in both directions (lo_make_geom checks): shifts, not divisions
    int n_img = m >> (g.lgw + g.lgh);
    int gy = (m >> g.lgw) & (g.GH - 1), gx = m & (g.GW - 1);
    int oy = gy * g.out_stride + g.out_oy[phase], ox = gx * g.out_stride + g.out_ox[phase];
    size_t off = ((size_t)(n_img * g.Hout + oy) * g.Wout + ox) * g.Cout + n0 + ochunk * 8;
    if (a.add_src) {
      f16x8 r = *reinterpret_cast<const f16x8*>(a.add_src + off);
#pragma unroll
      for (int j = 0; j < 8; ++j) h[j] = (f16)((float)h[j] + (float)r[j]);
    }
    if (a.act == 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { float x = (float)h[j]; h[j] = (f16)(x > 0.f ? x : 0.2f * x); }
    }
    if (a.out_pitch > 0)   // concatenated output tensor (teacher feature extractor); add_src / gb_v are not used with it
      *reinterpret_cast<f16x8*>(a.out + ((size_t)(n_img * g.Hout + oy) * g.Wout + ox) * a.out_pitch + a.out_choff + n0 + ochunk * 8) = h;
    else if (a.gb_dv) {    // fused GroupNorm-backward apply: the activation gradient lives on in this tile (LDS); stored only if somebody else reads it
      *reinterpret_cast<f16x8*>(so + ml * OPITCH + ochunk * 16) = h;
      if (a.gb_keep_out) *reinterpret_cast<f16x8*>(a.out + off) = h;
    } else
      *reinterpret_cast<f16x8*>(a.out + off) = h;
    if (a.bn_partial) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { float x = (float)h[j]; ga1[j] += x; ga2[j] += x * x; }
    }
    if (a.gb_v) {
      f16x8 vv = *reinterpret_cast<const f16x8*>(a.gb_v + off);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float hv = (float)vv[j];
        float du = (float)h[j] * lo_mish_grad(hv * gsc[j] + gsh[j]);
        ga1[j] += du;
        ga2[j] += du * hv;          // sum du*xhat = rstd * (sum du*v - mean * sum du): finished after the loop
      }
    }
    if (a.gn_partial || a.gf.y) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { float x = (float)h[j]; s0 += x; q0 += x * x; }
#pragma unroll
      for (int j = 4; j < 8; ++j) { float x = (float)h[j]; s1 += x; q1 += x * x; }
    }
  }
  if (a.bn_partial) {
    // per-channel (sum, sumsq) of this tile, fixed summation order -> bn_partial[m tile][channel][2]
    float* red = reinterpret_cast<float*>(smem + BM * OPITCH);   // [256][16] floats
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[tid * 16 + j * 2] = ga1[j]; red[tid * 16 + j * 2 + 1] = ga2[j]; }
    __syncthreads();
    float* dst = a.bn_partial + ((size_t)(phase * MT + mt_i) * g.Cout + n0) * 2;
    for (int o = tid; o < BN * 2; o += 256) {
      int cl = o >> 1, w = o & 1;
      int ccx = cl >> 3, j = cl & 7;
      float tot = 0.f;
      for (int r = 0; r < ORPP; ++r) tot += red[(r * OCPR + ccx) * 16 + j * 2 + w];
      dst[o] = tot;
    }
    __syncthreads();
  }
  if (a.gb_v) {
    // the saved mean / rstd are read here, after the store loop, not held in registers across it
    const int n_img_t = m0 >> (g.lgw + g.lgh);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int grp = (n0 + ochunk * 8 + j) / G;
      float mean = a.gb_stats[n_img_t * 16 + grp * 2], rstd = a.gb_stats[n_img_t * 16 + grp * 2 + 1];
      ga2[j] = rstd * (ga2[j] - mean * ga1[j]);
    }
    // reduce (ga1, ga2) over the row slots in a fixed order: BN*2 outputs, one per thread (looped)
    float* red = reinterpret_cast<float*>(smem + BM * OPITCH);   // [256][16] floats
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[tid * 16 + j * 2] = ga1[j]; red[tid * 16 + j * 2 + 1] = ga2[j]; }
    __syncthreads();
    const int per_sample = g.GH * g.GW;
    const int n_img = m0 / per_sample;
    const int mt = (m0 - n_img * per_sample) / BM;
    const int MTs = (per_sample / BM) * g.n_phase;
    float* dst = a.gb_P1 + (((size_t)n_img * MTs + phase * (per_sample / BM) + mt) * g.Cout + n0) * 2;
    for (int o = tid; o < BN * 2; o += 256) {
      int cl = o >> 1, w = o & 1;
      int ccx = cl >> 3, j = cl & 7;
      float tot = 0.f;
      for (int r = 0; r < ORPP; ++r) tot += red[(r * OCPR + ccx) * 16 + j * 2 + w];
      if (a.gb_dv) __hip_atomic_store(reinterpret_cast<unsigned int*>(dst) + o, __float_as_uint(tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else dst[o] = tot;
    }
    if (a.gb_dv) {
      // ---- GroupNorm-backward APPLY of the producing layer, here: wait until every m tile of this sample (same n tile) has
      //      published its P1 row, form the gamma-weighted group sums in lo_gn_bwd_apply's order, turn the tile's activation
      //      gradient (still in LDS) into dv.  Single phase ops only (the launcher checks).
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      lo_arrive_and_wait(a.gb_counter + n_img * NT + nt_i, a.gb_target, a.gb_fail, tid);
      float* s_g = red;                   // [BN][2] gamma-weighted per-channel totals
      float* s_c = red + 2 * BN;          // [BN / G][2] group means of (gamma du, gamma du xhat)
      for (int c = tid; c < BN; c += 256) {
        float t1 = 0.f, t2 = 0.f;
        const unsigned int* p = reinterpret_cast<const unsigned int*>(a.gb_P1) + ((size_t)n_img * MTs * g.Cout + n0 + c) * 2;
        for (int k = 0; k < MTs; ++k) {
          t1 += __uint_as_float(__hip_atomic_load(p + (size_t)k * g.Cout * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
          t2 += __uint_as_float(__hip_atomic_load(p + (size_t)k * g.Cout * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
        const float gm = a.gb_gamma[n0 + c];
        s_g[c * 2] = gm * t1;
        s_g[c * 2 + 1] = gm * t2;
      }
      __syncthreads();
      if (tid < (BN / G) * 2) {
        const int gl = tid >> 1, w = tid & 1;
        float tot = 0.f;
        for (int c = gl * G; c < (gl + 1) * G; ++c) tot += s_g[c * 2 + w];
        s_c[tid] = tot / ((float)per_sample * (float)G);
      }
      __syncthreads();
      // this thread's 8 channels lie in ONE group (G >= 8: checked by the launcher)
      const int gl = (ochunk * 8) / G, grp = n0 / G + gl;
      const float mean = a.gb_stats[n_img * 16 + grp * 2], rstd = a.gb_stats[n_img * 16 + grp * 2 + 1];
      const float nmr = -mean * rstd, kb = rstd * s_c[gl * 2], kc = rstd * s_c[gl * 2 + 1];
      float sc[8], sh[8], acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float gm = a.gb_gamma[n0 + ochunk * 8 + j];
        sc[j] = gm * rstd;
        sh[j] = __builtin_fmaf(nmr, gm, a.gb_beta[n0 + ochunk * 8 + j]);
        acc[j] = 0.f;
      }
#pragma unroll
      for (int i = 0; i < OP; ++i) {
        const int ml = orow + i * ORPP;
        const int m = m0 + ml;
        if (m >= a.M) continue;
        const f16x8 d = *reinterpret_cast<const f16x8*>(so + ml * OPITCH + ochunk * 16);
        const int gy = (m >> g.lgw) & (g.GH - 1), gx = m & (g.GW - 1);
        const size_t off = ((size_t)(n_img * g.Hout + gy) * g.Wout + gx) * g.Cout + n0 + ochunk * 8;
        const f16x8 vv = *reinterpret_cast<const f16x8*>(a.gb_v + off);
        f16x8 outv;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          lo_f2 du, xh;
          lo_gn_du2_plain((lo_f2){(float)vv[j], (float)vv[j + 1]}, (lo_f2){(float)d[j], (float)d[j + 1]}, (lo_f2){sc[j], sc[j + 1]},
                          (lo_f2){sh[j], sh[j + 1]}, (lo_f2){rstd, rstd}, (lo_f2){nmr, nmr}, du, xh);
          const lo_f2 dv = lo_gn_dv2(du, xh, (lo_f2){sc[j], sc[j + 1]}, kb, kc);
          const f16 d0 = (f16)dv[0], d1 = (f16)dv[1];
          outv[j] = d0; outv[j + 1] = d1;
          acc[j] += (float)d0; acc[j + 1] += (float)d1;
        }
        *reinterpret_cast<f16x8*>(a.gb_dv + off) = outv;
      }
      // per-channel sums of dv of this tile (conv bias gradient partials), fixed order over the row slots
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 8; ++j) red[tid * 8 + j] = acc[j];
      __syncthreads();
      for (int c = tid; c < BN; c += 256) {
        const int ccx = c >> 3, j = c & 7;
        float tot = 0.f;
        for (int r = 0; r < ORPP; ++r) tot += red[(r * OCPR + ccx) * 8 + j];
        a.gb_P2[((size_t)n_img * MTs + mt) * g.Cout + n0 + c] = tot;
      }
    }
    if (a.gn_partial) __syncthreads();   // the section below reuses `red` (no caller sets both today; the barrier keeps that legal)
  }
  if (a.gn_partial || a.gf.y) {
    // deterministic block reduction of the per-thread (sum, sumsq) of the two 4-channel halves of each chunk:
    //   level 1: 256 threads, each adds ORPP/P row slots of one (chunk, value)   level 2: P partials -> chunk sums
    //   level 3: half-chunks of a group (<= 16) -> group sums
    float* red = reinterpret_cast<float*>(smem + BM * OPITCH);  // [256][4] floats, then [P][NV], then [NV]
    red[tid * 4 + 0] = s0; red[tid * 4 + 1] = q0; red[tid * 4 + 2] = s1; red[tid * 4 + 3] = q1;
    constexpr int NV = OCPR * 4;          // values per row slot
    constexpr int P = 256 / NV;           // parts
    constexpr int RPP2 = ORPP / P;        // row slots per part
    static_assert(P >= 1 && RPP2 * P == ORPP, "epilogue reduction shape");
    float* red2 = red + 1024;
    float* red3 = red2 + 256;
    float* s_x = red3 + 64;               // fused GroupNorm: this tile's exchange line (32 floats), then the sample's statistics (16)
    float* s_stat = s_x + 32;
    if (tid < 32) s_x[tid] = 0.f;
    __syncthreads();
    {
      const int o = tid % NV, part = tid / NV;
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < RPP2; ++r) t += red[((part * RPP2 + r) * OCPR) * 4 + o];
      red2[part * NV + o] = t;
    }
    __syncthreads();
    if (tid < NV) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < P; ++q) t += red2[q * NV + tid];
      red3[tid] = t;                      // index = chunk*4 + half*2 + which
    }
    __syncthreads();
    const int ngroups = BN / G;           // BN >= G is checked by the launcher
    const int per_sample = g.GH * g.GW;
    const int n_img = m0 / per_sample;
    const int mt = (m0 - n_img * per_sample) / BM;
    const int MTs = (per_sample / BM) * g.n_phase;
    const int row = phase * (per_sample / BM) + mt;
    if (tid < ngroups * 2) {
      int gl = tid >> 1, which = tid & 1;
      int hc_begin = gl * G / 4, hc_end = (gl + 1) * G / 4;
      float tot = 0.f;
      for (int hc = hc_begin; hc < hc_end; ++hc) tot += red3[(hc >> 1) * 4 + (hc & 1) * 2 + which];
      int grp = (n0 / G) + gl;
      if (a.gn_partial) a.gn_partial[(((size_t)n_img * MTs + row) * 8 + grp) * 2 + which] = tot;
      s_x[grp * 2 + which] = tot;
    }
    if (a.gf.y) {
      // ---- GroupNorm + Mish of this tile, once the whole sample's sums are known (lo_common.h: LoGnFuse)
      __syncthreads();
      lo_gn_rendezvous(a.gf, n_img, row, nt_i, G, BN, 1.0f / ((float)(per_sample * g.n_phase) * (float)G), s_x, s_stat, tid);
      if (row == 0 && nt_i == 0 && tid < 16 && a.gf.stats) a.gf.stats[n_img * 16 + tid] = s_stat[tid];
      float sc[8], sh[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = n0 + ochunk * 8 + j, grp = c / G;
        const float mean = s_stat[grp * 2], rstd = s_stat[grp * 2 + 1];
        lo_gn_scale_shift(a.gf.gamma[c], a.gf.beta[c], mean, rstd, sc[j], sh[j]);
      }
#pragma unroll
      for (int i = 0; i < OP; ++i) {
        const int ml = orow + i * ORPP;
        const int m = m0 + ml;
        if (m >= a.M) continue;
        const f16x8 h = *reinterpret_cast<const f16x8*>(so + ml * OPITCH + ochunk * 16);
        const int gy = (m >> g.lgw) & (g.GH - 1), gx = m & (g.GW - 1);
        const int oy = gy * g.out_stride + g.out_oy[phase], ox = gx * g.out_stride + g.out_ox[phase];
        const size_t off = ((size_t)(n_img * g.Hout + oy) * g.Wout + ox) * g.Cout + n0 + ochunk * 8;
        f16x8 o = h;
        if (a.gf.mode != 0) o = *reinterpret_cast<const f16x8*>(a.gf.other + off);
        *reinterpret_cast<f16x8*>(a.gf.y + off) = lo_gn_apply8(h, sc, sh, a.gf.mode, o);
      }
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
  int s = 0;
  for (; s + 8 <= nsplit; s += 8) {              // eight loads in flight, added in split order
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = slab[(size_t)(s + u) * M * N + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) v += t[u];
  }
  for (; s < nsplit; ++s) v += slab[(size_t)s * M * N + i];
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
  int taps;         // total taps over all phases
  int M;            // pixels per phase = B*GH*GW
  int nsplit;
  int msteps_per_split;  // 32-pixel steps per split
  int packed_elems;
  LoGeom g;         // FORWARD geometry
};

// LDS-DMA staged like lo_igemm_nt.  Tiles are [32 pixel rows][BMW or BNW channels] fp16, unpadded; the transposed
// fragment reads (ds_read_b64_tr_b16) are kept conflict-free by XOR-swizzling 32-byte blocks inside a row:
//   256-byte rows: block ^= row & 7        128-byte rows: block ^= (row >> 1) & 3
// Output channels n >= Cout (a 32-channel layer run with the 64-wide tile) read the zero page and are not stored.
template <int RB>
__device__ __forceinline__ int lo_tr_swz(int row) {
  return RB == 256 ? (row & 7) : ((row >> 1) & 3);
}

template <int BMW, int BNW, int NSTAGE, int BKP>
__global__ __launch_bounds__(256) void lo_wgrad_tn(WgradArgs a) {
  static_assert(BKP == 32 || BKP == 64, "pixels per K step");
  constexpr int RBA = BMW * 2, RBB = BNW * 2;    // row bytes
  static_assert((RBA == 128 || RBA == 256) && (RBB == 128 || RBB == 256), "tile rows must be 128 or 256 bytes");
  constexpr int A_BYTES = BKP * RBA, B_BYTES = BKP * RBB;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int IA = A_BYTES / 1024 / 4, IB = B_BYTES / 1024 / 4;   // LDS-DMA instructions per wave and K step
  constexpr int CPA = RBA / 16, CPB = RBB / 16;  // 16-byte chunks per row
  constexpr int RPA = 64 / CPA, RPB = 64 / CPB;  // rows per wave-instruction
  constexpr int LPT = IA + IB, D = NSTAGE - 1;
  constexpr int WM = BMW / 2, WN = BNW / 2, MI = WM / 16, NI = WN / 16;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NSTAGE * STAGE];

  const LoGeom& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int ntile_n = (g.Cout + BMW - 1) / BMW, ntile_c = g.Cin / BNW;
  // logical id -> (tap fastest, then channel tiles, then pixel split): workgroups that read the same pixels are adjacent
  const int wid = lo_xcd_remap(blockIdx.x, gridDim.x);
  int t = wid % a.taps;
  const int tile = (wid / a.taps) % (ntile_n * ntile_c);
  const int tn = tile % ntile_n, tc = tile / ntile_n;
  const int n0 = tn * BMW, c0 = tc * BNW;
  int phase = 0;                   // t enumerates (phase, tap)
  while (t >= g.T[phase]) { t -= g.T[phase]; ++phase; }
  const int dyo = g.dy[phase][t], dxo = g.dx[phase][t];
  const int ooy = g.out_oy[phase], oox = g.out_ox[phase];
  const int split = wid / (a.taps * ntile_n * ntile_c);
  const int ms_begin = split * a.msteps_per_split;
  const int ms_total = (a.M + BKP - 1) / BKP;
  const int ms_end = min(ms_total, ms_begin + a.msteps_per_split);
  const int nk = ms_end - ms_begin;
  const f16* zpage = reinterpret_cast<const f16*>(lo_zero_page);

  // per-lane constants of the DMA instructions
  int a_row[IA], a_col[IA], b_row[IB], b_col[IB];
#pragma unroll
  for (int i = 0; i < IA; ++i) {
    int row = (wave * IA + i) * RPA + lane / CPA, pos = lane % CPA;
    int chunk = (((pos >> 1) ^ lo_tr_swz<RBA>(row)) << 1) | (pos & 1);
    a_row[i] = row;
    a_col[i] = n0 + chunk * 8;
  }
#pragma unroll
  for (int i = 0; i < IB; ++i) {
    int row = (wave * IB + i) * RPB + lane / CPB, pos = lane % CPB;
    int chunk = (((pos >> 1) ^ lo_tr_swz<RBB>(row)) << 1) | (pos & 1);
    b_row[i] = row;
    b_col[i] = c0 + chunk * 8;
  }
  const int pmask_w = g.GW - 1, pmask_h = g.GH - 1;
  // all tensor dims on this path are powers of two (checked by the launcher): multiplies become shifts
  const int sh_hin = g.lg_hin, sh_win = g.lg_win, sh_cin = g.lg_cin, sh_hout = g.lg_hout, sh_wout = g.lg_wout;
  const bool cout_pow2 = g.lg_cout >= 0;

  auto issue = [&](int stage, int ms) __attribute__((always_inline)) {
    unsigned char* sa = smem + stage * STAGE;
    unsigned char* sb = sa + A_BYTES;
    const bool live = ms < ms_end;
#pragma unroll
    for (int i = 0; i < IA; ++i) {
      int m = ms * BKP + a_row[i];
      int gx = m & pmask_w, gy = (m >> g.lgw) & pmask_h, n_img = m >> (g.lgw + g.lgh);
      int oy = gy * g.out_stride + ooy, ox = gx * g.out_stride + oox;
      bool ok = live && m < a.M && a_col[i] < g.Cout;
      const int pix = (((n_img << sh_hout) + oy) << sh_wout) + ox;
      const f16* src = ok ? a.dy + (cout_pow2 ? ((size_t)pix << g.lg_cout) : (size_t)pix * g.Cout) + a_col[i] : zpage;
      lo_dma16(src, (unsigned int)(size_t)(sa + (wave * IA + i) * 1024));
    }
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      int m = ms * BKP + b_row[i];
      int gx = m & pmask_w, gy = (m >> g.lgw) & pmask_h, n_img = m >> (g.lgw + g.lgh);
      int iy = gy * g.in_stride + dyo, ix = gx * g.in_stride + dxo;
      bool ok = live && m < a.M && (unsigned)iy < (unsigned)g.Hin && (unsigned)ix < (unsigned)g.Win;
      const f16* src = ok ? a.x + ((size_t)((((n_img << sh_hin) + iy) << sh_win) + ix) << sh_cin) + b_col[i] : zpage;
      lo_dma16(src, (unsigned int)(size_t)(sb + (wave * IB + i) * 1024));
    }
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // transposed fragment reads.  Each group of 32 pixel rows is assigned to the MFMA k positions by the SAME
  // permutation for both operands (lane group q reads rows 4q..4q+3 and 16+4q..16+4q+3); any consistent k
  // permutation leaves the sum unchanged.
  const int q16 = lane >> 4, i16 = lane & 15;
  const int trow = 4 * q16 + (i16 >> 2);   // row supplied by this lane (first read); +16 for the second
  const int tsub = (i16 & 3) * 8;          // byte offset inside the 32-byte block
  constexpr int KS = BKP / 32;
  int aoff[MI][2 * KS], boff[NI][2 * KS];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int h = 0; h < 2 * KS; ++h) {
      int R = trow + 16 * h, blk = (wm * WM + mi * 16) / 16;
      aoff[mi][h] = R * RBA + ((blk ^ lo_tr_swz<RBA>(R)) * 32) + tsub;
    }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int h = 0; h < 2 * KS; ++h) {
      int R = trow + 16 * h, blk = (wn * WN + ni * 16) / 16;
      boff[ni][h] = A_BYTES + R * RBB + ((blk ^ lo_tr_swz<RBB>(R)) * 32) + tsub;
    }

  if (nk > 0) {
#pragma unroll
    for (int s = 0; s < D; ++s) issue(s, ms_begin + s);
    int rs = 0, ws = D % NSTAGE;
    for (int it = 0; it < nk; ++it) {
      LO_VMCNT(LPT * (D - 1));
      __builtin_amdgcn_s_barrier();
      issue(ws, ms_begin + it + D);
      const unsigned char* sbase = smem + rs * STAGE;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        f16x8 af[MI], bf[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          h16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((h16x4 __attribute__((address_space(3)))*)(sbase + aoff[mi][2 * ks]));
          h16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((h16x4 __attribute__((address_space(3)))*)(sbase + aoff[mi][2 * ks + 1]));
          af[mi] = (f16x8){(f16)lo[0], (f16)lo[1], (f16)lo[2], (f16)lo[3], (f16)hi[0], (f16)hi[1], (f16)hi[2], (f16)hi[3]};
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          h16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((h16x4 __attribute__((address_space(3)))*)(sbase + boff[ni][2 * ks]));
          h16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((h16x4 __attribute__((address_space(3)))*)(sbase + boff[ni][2 * ks + 1]));
          bf[ni] = (f16x8){(f16)lo[0], (f16)lo[1], (f16)lo[2], (f16)lo[3], (f16)hi[0], (f16)hi[1], (f16)hi[2], (f16)hi[3]};
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
      }
      rs = (rs + 1 == NSTAGE) ? 0 : rs + 1;
      ws = (ws + 1 == NSTAGE) ? 0 : ws + 1;
    }
    LO_VMCNT(0);
  }
  // D[n][c] block (mi, ni): lane holds column c = ni*16 + (lane&15), rows n = mi*16 + (lane>>4)*4 + j
  float* slab = a.slab + (size_t)split * a.packed_elems + g.wofs[phase];
  const int Ktot = g.T[phase] * g.Cin;
  const int rs_w = g.rs[phase][t];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      int c = c0 + wn * WN + ni * 16 + (lane & 15);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int n = n0 + wm * WM + mi * 16 + (lane >> 4) * 4 + j;
        if (n < g.Cout) {
          if (a.direct) a.grad[(size_t)n * g.sn + (size_t)c * g.sc + rs_w] = acc[mi][ni][j] * a.scale;
          else slab[(size_t)n * Ktot + t * g.Cin + c] = acc[mi][ni][j];
        }
      }
    }
}

// sum the split slabs and scatter to the canonical fp32 gradient:  grad[n*sn + c*sc + rs] = scale * sum_s slab
// block = 64 column threads x 4 split groups; a column = 4 consecutive packed elements (same n and tap, consecutive c:
// 16-byte slab loads); group sg sums splits sg, sg+4, ... (4 loads in flight), the four partial sums are added in a fixed
// order through LDS (bitwise reproducible).  Small weight tensors (36 k elements, up to 256 splits) get 4x the
// workgroups and 4x the loads in flight of a one-thread-per-column loop.
__device__ __forceinline__ void lo_wgrad_reduce_block(const float* __restrict__ slab, float* __restrict__ grad, const LoGeom& g,
                                                      int total, int nsplit, float scale, int bid) {
  __shared__ f32x4 part[4][64];
  const int col = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const int i = (bid * 64 + col) * 4;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (i < total) {
    const float* src = slab + i;
    int s = sg;
    for (; s + 12 < nsplit; s += 16) {
      f32x4 a0 = *reinterpret_cast<const f32x4*>(src + (size_t)s * total);
      f32x4 a1 = *reinterpret_cast<const f32x4*>(src + (size_t)(s + 4) * total);
      f32x4 a2 = *reinterpret_cast<const f32x4*>(src + (size_t)(s + 8) * total);
      f32x4 a3 = *reinterpret_cast<const f32x4*>(src + (size_t)(s + 12) * total);
      v += a0; v += a1; v += a2; v += a3;
    }
    for (; s < nsplit; s += 4) v += *reinterpret_cast<const f32x4*>(src + (size_t)s * total);
  }
  part[sg][col] = v;
  __syncthreads();
  if (sg != 0 || i >= total) return;
  v = part[0][col] + part[1][col] + part[2][col] + part[3][col];
  int p = 0;
#pragma unroll
  for (int q = 1; q < LO_MAX_PHASE; ++q) if (q < g.n_phase && i >= g.wofs[q]) p = q;
  int j = i - g.wofs[p];
  int K = g.T[p] * g.Cin;
  int n = j / K, k = j - n * K;
  int t = k / g.Cin, c = k - t * g.Cin;
  float* dst = grad + (size_t)n * g.sn + (size_t)c * g.sc + g.rs[p][t];
#pragma unroll
  for (int e = 0; e < 4; ++e) dst[(size_t)e * g.sc] = v[e] * scale;
}
__global__ __launch_bounds__(256) void lo_wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ grad, LoGeom g,
                                                              int total, int nsplit, float scale) {
  lo_wgrad_reduce_block(slab, grad, g, total, nsplit, scale, (int)blockIdx.x);
}
// The same reduction for the 3x3 convolutions (one phase, nine taps, canonical weight W[co][ci][3][3]: sc == 9) with COALESCED
// stores.  The kernel above walks the packed layout [n][tap][c] and scatters every value to n*sn + c*9 + rs: 4-byte stores 36 bytes
// apart, neighbouring lanes 144 bytes apart -- for the 512 x 512 layers (2.4 M elements from two slabs) that was 39 us per launch,
// against 6 us for the small layers whose time is the slab reads (rocprofv3, round 4: 176 us per step over the 15 launches).  Here a
// workgroup owns (output channel n, 64 input channels): its nine tap segments are summed over the splits (thread = (16-byte column,
// tap, split group)), staged in LDS, and leave as ONE contiguous run of 64 x 9 floats in 16-byte stores.  Same split-group order
// as above: the same bits.
__global__ __launch_bounds__(576) void lo_wgrad_reduce_rows_kernel(const float* __restrict__ slab, float* __restrict__ grad, LoGeom g,
                                                                   int total, int nsplit, float scale) {
  __shared__ float part[4][9][64];
  __shared__ int tap_of_rs[9];
  const int x = threadIdx.x, t = threadIdx.y, sg = threadIdx.z, G = blockDim.z;
  const int cblocks = g.Cin >> 6;
  const int n = blockIdx.x / cblocks, c0 = (blockIdx.x - n * cblocks) << 6;
  if (sg == 0 && x == 0) tap_of_rs[g.rs[0][t]] = t;
  const size_t i = (size_t)n * 9 * g.Cin + (size_t)t * g.Cin + c0 + 4 * x;
  const float* src = slab + i;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  int s = sg;
  for (; s + 3 * G < nsplit; s += 4 * G) {
    f32x4 a0 = *reinterpret_cast<const f32x4*>(src + (size_t)s * total);
    f32x4 a1 = *reinterpret_cast<const f32x4*>(src + (size_t)(s + G) * total);
    f32x4 a2 = *reinterpret_cast<const f32x4*>(src + (size_t)(s + 2 * G) * total);
    f32x4 a3 = *reinterpret_cast<const f32x4*>(src + (size_t)(s + 3 * G) * total);
    v += a0; v += a1; v += a2; v += a3;
  }
  for (; s < nsplit; s += G) v += *reinterpret_cast<const f32x4*>(src + (size_t)s * total);
  *reinterpret_cast<f32x4*>(&part[sg][t][4 * x]) = v;
  __syncthreads();
  const int w = x + 16 * (t + 9 * sg);
  if (w >= 144) return;
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int f = 4 * w + e, cc = f / 9, rs = f - cc * 9, tt = tap_of_rs[rs];
    float acc = part[0][tt][cc];
    for (int q = 1; q < G; ++q) acc += part[q][tt][cc];
    o[e] = acc * scale;
  }
  *reinterpret_cast<f32x4*>(grad + (size_t)n * g.sn + (size_t)c0 * 9 + 4 * w) = o;
}
// ... and for the 4x4 stride-2 transposed convolutions (four phases of four taps, canonical weight W[ci][co][4][4]: sn == 16,
// sc == Cout * 16, so the contiguous runs of the gradient are (co, rs) for a fixed ci).  A workgroup owns 16 reduced channels c
// (one 64-byte run of every packed segment) x 4 output channels n: thread = (16-byte column, n, phase, tap) sums its column over the
// splits in split order, the 16 x (4 x 16) block is staged in LDS and leaves as sixteen 256-byte runs in 16-byte stores (the
// scattering kernel needed 15 - 35 us for these four layers).
__global__ __launch_bounds__(256) void lo_wgrad_reduce_convt_kernel(const float* __restrict__ slab, float* __restrict__ grad, LoGeom g,
                                                                    int total, int nsplit, float scale) {
  __shared__ float outb[16][4][16];
  const int tid = threadIdx.x, x = tid & 3, seg = tid >> 2, nl = seg >> 4, p = (seg >> 2) & 3, t = seg & 3;
  const int cblocks = g.Cin >> 4;
  const int nb = blockIdx.x / cblocks, c0 = (blockIdx.x - nb * cblocks) << 4, n0 = nb << 2;
  const size_t i = (size_t)g.wofs[p] + (size_t)(n0 + nl) * 4 * g.Cin + (size_t)t * g.Cin + c0 + 4 * x;
  const float* src = slab + i;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  int s = 0;
  for (; s + 4 <= nsplit; s += 4) {
    f32x4 a0 = *reinterpret_cast<const f32x4*>(src + (size_t)s * total);
    f32x4 a1 = *reinterpret_cast<const f32x4*>(src + (size_t)(s + 1) * total);
    f32x4 a2 = *reinterpret_cast<const f32x4*>(src + (size_t)(s + 2) * total);
    f32x4 a3 = *reinterpret_cast<const f32x4*>(src + (size_t)(s + 3) * total);
    v += a0; v += a1; v += a2; v += a3;
  }
  for (; s < nsplit; ++s) v += *reinterpret_cast<const f32x4*>(src + (size_t)s * total);
  const int rs = g.rs[p][t];
#pragma unroll
  for (int e = 0; e < 4; ++e) outb[4 * x + e][nl][rs] = v[e] * scale;
  __syncthreads();
  const int cc = tid >> 4, q = tid & 15;
  *reinterpret_cast<f32x4*>(grad + (size_t)(c0 + cc) * g.sc + (size_t)n0 * 16 + 4 * q) =
      *reinterpret_cast<const f32x4*>(&outb[cc][q >> 2][(q & 3) * 4]);
}
static bool lo_wgrad_reduce_convt_applies(const LoGeom& g) {
  if (g.n_phase != 4 || g.sn != 16 || g.sc != g.Cout * 16 || g.Cin % 16 != 0 || g.Cout % 4 != 0) return false;
  for (int p = 0; p < 4; ++p)
    if (g.T[p] != 4 || g.wofs[p] != p * 4 * g.Cin * g.Cout) return false;
  return true;
}
static bool lo_wgrad_reduce_rows_applies(const LoGeom& g) {
  return g.n_phase == 1 && g.T[0] == 9 && g.sc == 9 && g.sn == 9 * g.Cin && g.wofs[0] == 0 && g.Cin % 64 == 0;
}
// slab [nsplit][packed] -> canonical fp32 gradient (scaled): the coalescing form where the geometry allows it
static int lo_wgrad_reduce_launch(const float* slab, float* grad, const LoGeom& g, int total, int nsplit, float scale, hipStream_t st) {
  if (lo_wgrad_reduce_rows_applies(g)) {
    const int G = nsplit >= 4 ? 4 : (nsplit < 1 ? 1 : nsplit);
    hipLaunchKernelGGL(lo_wgrad_reduce_rows_kernel, dim3(g.Cout * (g.Cin / 64)), dim3(16, 9, G), 0, st, slab, grad, g, total, nsplit, scale);
  } else if (lo_wgrad_reduce_convt_applies(g)) {
    hipLaunchKernelGGL(lo_wgrad_reduce_convt_kernel, dim3((g.Cout / 4) * (g.Cin / 16)), dim3(256), 0, st, slab, grad, g, total, nsplit, scale);
  } else {
    hipLaunchKernelGGL(lo_wgrad_reduce_kernel, dim3((total / 4 + 63) / 64), dim3(256), 0, st, slab, grad, g, total, nsplit, scale);
  }
  LO_LAUNCH_CHECK("wgrad_reduce");
  return LO_OK;
}
// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
int lo_pack_all(const LoPackJob* jobs_dev, int njobs, int nblocks, hipStream_t st, int block_base) {
  if (njobs <= 0 || nblocks <= 0) return LO_OK;
  LoProfScope _p("lo_pack_all", 0, 0, st);
  hipLaunchKernelGGL(lo_pack_all_kernel, dim3(nblocks), dim3(256), 0, st, jobs_dev, njobs, block_base);
  LO_LAUNCH_CHECK("pack_all");
  return LO_OK;
}

int lo_pack_weight(const float* w, f16* wp, const LoGeom& g, hipStream_t st) {
  int total = geom_packed_elems(g);
  LoProfScope _p("lo_pack_weight", 0, 6.0 * total, st);
  hipLaunchKernelGGL(lo_pack_weight_kernel, dim3((total + 255) / 256), dim3(256), 0, st, w, wp, g, total);
  LO_LAUNCH_CHECK("pack_weight");
  return LO_OK;
}

static inline double geom_flops(const LoGeom& g) {
  double f = 0;
  for (int p = 0; p < g.n_phase; ++p) f += 2.0 * g.B * g.GH * g.GW * (double)g.Cout * g.T[p] * g.Cin;
  return f;
}
// algorithmic bytes: input read once + output written once (fp16) + packed weights once
static inline double geom_bytes(const LoGeom& g) {
  return 2.0 * ((double)g.B * g.Hin * g.Win * g.Cin + (double)g.B * g.Hout * g.Wout * g.Cout + geom_packed_elems(g));
}

template <int BM, int BN, int BK, int NS, bool SK, bool F8 = false>
static int igemm_launch(dim3 grid, const IgemmArgs& a, hipStream_t st) {
  constexpr int lds = igemm_lds_bytes<BM, BN, BK, NS, F8>();
  static_assert(lds <= 160 * 1024, "LDS");
  if constexpr (lds > 65536) {
    static bool attr = false;
    if (!attr) {
      LO_HIP(hipFuncSetAttribute((const void*)lo_igemm_nt<BM, BN, BK, NS, SK, F8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      attr = true;
    }
  }
  LO_LAUNCH_STOP((lo_igemm_nt<BM, BN, BK, NS, SK, F8>), grid, dim3(256), lds, st, a);   // may carry the hand-over event of the dv it writes
  return LO_OK;
}

template <int BM, int BN, int BK>
static int launch_igemm(const IgemmArgs& a, hipStream_t st) {
  const LoGeom& g = a.g;
  static char name[64];
  snprintf(name, sizeof(name), "lo_igemm_nt<%d,%d,%d>%s", BM, BN, BK, a.nsplit > 1 ? "/splitK" : "");  // same text for every call of this instantiation
  // algorithmic bytes of the FUSED op: operands + output once, plus what its epilogue must read by definition: the residual
  // / skip gradient it adds (add_src) and the producing layer's raw conv output for the fused GroupNorm-backward reduction
  // (gb_v), each the size of the output
  const double out_bytes = 2.0 * (double)g.B * g.Hout * g.Wout * g.Cout;
  LoProfScope _p(lo_prof_geom_name(name, g), geom_flops(g), geom_bytes(g) + (a.add_src ? out_bytes : 0.0) + (a.gb_v ? out_bytes : 0.0), st);
  dim3 grid(((a.M + BM - 1) / BM) * (g.Cout / BN) * (a.nsplit > 1 ? a.nsplit : g.n_phase));
  constexpr int STAGE_BYTES = (BM + BN) * BK * 2;
  constexpr int NSTAGE = STAGE_BYTES >= 32768 ? 2 : (STAGE_BYTES >= 16384 ? 3 : 4);
  // short K loops (<= 9 steps: the 64-channel 3x3 layers) are prologue / epilogue bound: one LDS stage less puts a third
  // workgroup on the CU (measured 46.9 -> 39.8 us at 64 channels, 64x64; no gain on the longer loops; more stages where the grid
  // leaves LDS unused did not make any launch faster either: these launches sit at the L2 -> LDS rate)
  int ksteps_max = 0;
  for (int p = 0; p < g.n_phase; ++p) ksteps_max = g.T[p] * (g.Cin / BK) > ksteps_max ? g.T[p] * (g.Cin / BK) : ksteps_max;
  const bool shallow = ksteps_max <= 9;
  if (a.nsplit > 1)
    LO_CHECK(igemm_launch<BM, BN, BK, NSTAGE, true>(grid, a, st));
  else if (shallow && NSTAGE > 2)
    LO_CHECK(igemm_launch<BM, BN, BK, (NSTAGE > 2 ? NSTAGE - 1 : 2), false>(grid, a, st));
  else
    LO_CHECK(igemm_launch<BM, BN, BK, NSTAGE, false>(grid, a, st));
  LO_LAUNCH_CHECK("igemm");
  return LO_OK;
}

void lo_conv_pick_tile(const LoGeom& g, int* bm_out, int* bn_out);

// ---- fp8 operand path -------------------------------------------------------------------------------------------------
template <int BM, int BN>
static int launch_igemm_f8(const IgemmArgs& a, hipStream_t st) {
  const LoGeom& g = a.g;
  static char name[64];
  snprintf(name, sizeof(name), "lo_igemm_nt<%d,%d,128>/fp8", BM, BN);
  LoProfScope _p(lo_prof_geom_name(name, g), geom_flops(g), 0.5 * geom_bytes(g) + (double)g.B * g.Hout * g.Wout * g.Cout, st);
  dim3 grid(((a.M + BM - 1) / BM) * (g.Cout / BN) * g.n_phase);
  constexpr int STAGE_BYTES = (BM + BN) * 128;
  constexpr int NSTAGE = STAGE_BYTES >= 32768 ? 2 : (STAGE_BYTES >= 16384 ? 3 : 4);
  LO_CHECK((igemm_launch<BM, BN, 128, NSTAGE, false, true>(grid, a, st)));
  LO_LAUNCH_CHECK("igemm_f8");
  return LO_OK;
}
bool lo_conv_f8_applies(const LoGeom& g) {
  // every shape the e4m3 K step fits, also where a fused-tap fp16 kernel owns the fp16 form (conv by conv the e4m3 implicit GEMM is the
  // faster launch there too: profiles/r04_fp8_per_layer.txt; on the step +0.5 %).  LO_F8_FORCE=0: the round-2 selection (those shapes stay fp16)
  static const bool force = !(getenv("LO_F8_FORCE") && atoi(getenv("LO_F8_FORCE")) == 0);
  return g.Cin % 128 == 0 && g.Cout % 64 == 0 && (force || lo_conv3_tiles_per_image(g, false) == 0);
}
// Same op as lo_conv_run with both operands in e4m3: in8 = fp8(activation * LO_F8_ACT_SCALE) in the fp16 tensor's layout,
// w8 / wscale from lo_pack_f8_all.  Output, bias, residual add and GroupNorm partials as in the fp16 path (same tiles, so
// lo_conv_mt(g) rows of partials).
int lo_conv_run_f8(const LoGeom& g, const uint8_t* in8, const uint8_t* w8, const float* wscale, const float* bias, const f16* add_src,
                   f16* out, float* gn_partial, hipStream_t st) {
  LO_REQUIRE(lo_conv_f8_applies(g), "lo_conv_run_f8: geometry not supported (Cin %% 128, Cout %% 64)");
  IgemmArgs a;
  memset(&a, 0, sizeof(a));       // (a.gf.y = null: no fused GroupNorm on the fp8 path)
  a.in = reinterpret_cast<const f16*>(in8); a.w = reinterpret_cast<const f16*>(w8); a.f8_scale = wscale;
  a.bias = bias; a.add_src = add_src; a.out = out; a.gn_partial = gn_partial;
  a.g = g;
  a.M = g.B * g.GH * g.GW;
  a.nsplit = 1;
  int ksteps = 0;
  for (int p = 0; p < g.n_phase; ++p) ksteps = g.T[p] * (g.Cin / 128) > ksteps ? g.T[p] * (g.Cin / 128) : ksteps;
  a.ksteps_per_split = ksteps;
  int bm, bn;
  lo_conv_pick_tile(g, &bm, &bn);
  if (gn_partial) {
    LO_REQUIRE((g.GH * g.GW) % bm == 0 && a.M % bm == 0, "lo_conv_run_f8: GN partials need whole tiles per sample");
    LO_REQUIRE((g.Cout >> 3) <= bn, "lo_conv_run_f8: GroupNorm group wider than the N tile");
  }
  if (bm == 128 && bn == 128) return launch_igemm_f8<128, 128>(a, st);
  if (bm == 128 && bn == 64) return launch_igemm_f8<128, 64>(a, st);
  if (bm == 64 && bn == 128) return launch_igemm_f8<64, 128>(a, st);
  if (bm == 64 && bn == 64) return launch_igemm_f8<64, 64>(a, st);
  lo_set_error("lo_conv_run_f8: no kernel for tile %dx%d", bm, bn);
  return LO_ERR_ARG;
}

// fp16 packed weights -> e4m3 with one scale per (phase, output channel) row: scale = amax / 448 (1 for an all-zero row);
// wscale = scale / LO_F8_ACT_SCALE is what the conv epilogue multiplies by.  One workgroup per row; all layers of a model in
// one launch (job table in device memory, like lo_pack_all).
__device__ __forceinline__ void lo_pack_f8_row(const LoPackF8Job& J, int row) {   // row = phase * Cout + n
  __shared__ float s_red[4];
  const int p = row / J.Cout, n = row - p * J.Cout;
  const int K = J.K[p];
  const size_t o = (size_t)J.wofs[p] + (size_t)n * K;
  const f16* src = J.src + o;
  const int tid = threadIdx.x;
  float amax = 0.f;
  for (int k = tid * 8; k < K; k += 256 * 8) {
    f16x8 v = *reinterpret_cast<const f16x8*>(src + k);
#pragma unroll
    for (int q = 0; q < 8; ++q) amax = fmaxf(amax, fabsf((float)v[q]));
  }
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) amax = fmaxf(amax, __shfl_xor(amax, s, 64));
  if ((tid & 63) == 0) s_red[tid >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
  const float scale = amax > 0.f ? amax * (1.0f / LO_F8_MAX) : 1.0f;
  const float inv = 1.0f / scale;
  for (int k = tid * 8; k < K; k += 256 * 8) {
    f16x8 v = *reinterpret_cast<const f16x8*>(src + k);
    u32x2 q = {lo_pack4_fp8((float)v[0] * inv, (float)v[1] * inv, (float)v[2] * inv, (float)v[3] * inv),
               lo_pack4_fp8((float)v[4] * inv, (float)v[5] * inv, (float)v[6] * inv, (float)v[7] * inv)};
    *reinterpret_cast<u32x2*>(J.dst + o + k) = q;
  }
  if (tid == 0) J.scale[row] = scale * (1.0f / LO_F8_ACT_SCALE);
}
__global__ __launch_bounds__(256) void lo_pack_f8_kernel(const LoPackF8Job* __restrict__ jobs, int njobs, int block_base) {
  const int bid = (int)blockIdx.x + block_base;
  int j = 0;
  while (j + 1 < njobs && bid >= jobs[j + 1].block0) ++j;
  lo_pack_f8_row(jobs[j], bid - jobs[j].block0);
}
__global__ __launch_bounds__(256) void lo_pack_f8_one_kernel(LoPackF8Job J) { lo_pack_f8_row(J, (int)blockIdx.x); }
int lo_pack_f8_one(const LoGeom& g, const f16* wp, uint8_t* w8, float* wscale, hipStream_t st) {
  LoPackF8Job j;
  lo_pack_f8_job(&j, g, wp, w8, wscale, 0);
  hipLaunchKernelGGL(lo_pack_f8_one_kernel, dim3(g.n_phase * g.Cout), dim3(256), 0, st, j);
  LO_LAUNCH_CHECK("pack_f8_one");
  return LO_OK;
}
// x8 = e4m3(x * LO_F8_ACT_SCALE), saturating (stand-alone form of what the GroupNorm forward emits in fp8 mode)
__global__ __launch_bounds__(256) void lo_quantize_f8_kernel(const f16* __restrict__ x, uint8_t* __restrict__ x8, size_t n) {
  size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) {
    f16x4 v = *reinterpret_cast<const f16x4*>(x + i);
    *reinterpret_cast<uint32_t*>(x8 + i) = lo_pack4_fp8((float)v[0] * LO_F8_ACT_SCALE, (float)v[1] * LO_F8_ACT_SCALE,
                                                       (float)v[2] * LO_F8_ACT_SCALE, (float)v[3] * LO_F8_ACT_SCALE);
  }
}
int lo_quantize_f8(const f16* x, uint8_t* x8, size_t n, hipStream_t st) {
  LO_REQUIRE(n % 4 == 0, "lo_quantize_f8: element count must be a multiple of 4");
  hipLaunchKernelGGL(lo_quantize_f8_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, x, x8, n);
  LO_LAUNCH_CHECK("quantize_f8");
  return LO_OK;
}
void lo_pack_f8_job(LoPackF8Job* j, const LoGeom& g, const f16* src, uint8_t* dst, float* scale, int block0) {
  memset(j, 0, sizeof(*j));
  j->src = src; j->dst = dst; j->scale = scale; j->Cout = g.Cout; j->n_phase = g.n_phase; j->block0 = block0;
  for (int p = 0; p < g.n_phase; ++p) { j->K[p] = g.T[p] * g.Cin; j->wofs[p] = g.wofs[p]; }
}
int lo_pack_f8_all(const LoPackF8Job* jobs_dev, int njobs, int nblocks, hipStream_t st, int block_base) {
  if (njobs <= 0 || nblocks <= 0) return LO_OK;
  LoProfScope _p("lo_pack_f8", 0, 0, st);
  hipLaunchKernelGGL(lo_pack_f8_kernel, dim3(nblocks), dim3(256), 0, st, jobs_dev, njobs, block_base);
  LO_LAUNCH_CHECK("pack_f8");
  return LO_OK;
}

// Run one conv-like op.  `slab` + nsplit > 1 selects split-K (output = fp32 partials, caller reduces).
// Split-K plan for a single-phase convolution whose GroupNorm groups are (sample, group)-local (lo_gn_bwd_local_applies on ITS
// OUTPUT): 128 x 128 tiles, K split so that about two workgroups per CU run.  Returns the number of splits, or 0 when the op
// should stay on its one-launch kernel (long grids, fused-tap geometry that already fills the chip, too few K steps).
int lo_conv_splitk_plan(const LoGeom& g) {
  if (g.n_phase != 1 || g.T[0] < 9 || g.Cin % 64 != 0 || g.Cout % 128 != 0) return 0;
  const long M = (long)g.B * g.GH * g.GW;
  if (M % 128 != 0 || !lo_gn_bwd_local_applies(g.GH * g.GW, g.Cout)) return 0;
  const long tiles = (M / 128) * (g.Cout / 128);
  if (tiles >= 256) return 0;                         // the one-launch kernels already put a workgroup on every CU
  const int ksteps = g.T[0] * (g.Cin / 64);
  int ns = (int)((512 + tiles - 1) / tiles);
  while (ns > 1 && ksteps / ns < 8) --ns;             // at least 8 K steps per split
  return ns >= 2 ? ns : 0;
}

// Data-gradient op g with the fused GroupNorm-backward reduction: can its epilogue also run the APPLY pass (LoGnBwdFuse::dv)?
// Single-phase ops on lo_igemm_nt / lo_conv3x3_pp whose tiles lie inside one sample and whose 8-channel chunks lie inside one
// GroupNorm group.  mts / nt: P1 rows per sample and n tiles per row of the launch.
bool lo_conv_gnb_apply_tiles(const LoGeom& g, int* mts, int* nt) {
  if (g.n_phase != 1 || (g.Cout >> 3) < 8 || g.Cout % 64 != 0 || g.Cin % 64 != 0) return false;
  if (lo_conv3_tiles_per_image(g, false) > 0) return lo_conv3_gn_fuse_tiles(g, mts, nt);
  int bm, bn;
  lo_conv_pick_tile(g, &bm, &bn);
  const int per_sample = g.GH * g.GW;
  if (per_sample % bm != 0 || (g.Cout >> 3) > bn) return false;
  *mts = per_sample / bm;
  *nt = g.Cout / bn;
  return *nt <= 8 && *mts <= LO_GNF_MAX_TILES;
}

// Does the kernel lo_conv_run would launch for g (forward conv, no residual / gradient epilogue) take the fused GroupNorm epilogue,
// and with which tile grid per sample?  mts: partial rows per sample (m tiles x phases), nt: n tiles per row.
bool lo_conv_gn_fuse_tiles(const LoGeom& g, int* mts, int* nt) {
  if (lo_convt4_patch_applies(g)) return false;               // patch-resident transposed conv: own epilogue (not built)
  if (lo_conv3_tiles_per_image(g, false) > 0) return lo_conv3_gn_fuse_tiles(g, mts, nt);
  if (g.Cin % 64 != 0 || g.Cout % 64 != 0) return false;
  int bm, bn;
  lo_conv_pick_tile(g, &bm, &bn);
  const int per_sample = g.GH * g.GW;
  if (per_sample % bm != 0 || (g.Cout >> 3) > bn) return false;
  *mts = (per_sample / bm) * g.n_phase;
  *nt = g.Cout / bn;
  return *mts * *nt <= LO_GNF_MAX_TILES;
}

int lo_conv_run(const LoGeom& g, const f16* in, const f16* wp, const float* bias, const f16* add_src, f16* out,
                float* gn_partial, float* slab, int nsplit, hipStream_t st, const LoGnBwdFuse* gb, const LoConvExtra* ex,
                const LoGnFuse* gf) {
  IgemmArgs a;
  memset(&a.gf, 0, sizeof(a.gf));
  if (gf) {
    int mts = 0, nt = 0;
    LO_REQUIRE(nsplit <= 1 && !add_src && !gb && !ex && gf->y && gf->xbuf && gf->counter && gf->fail && gf->gamma && gf->beta &&
               (gf->mode == 0 || gf->other), "lo_conv_run: bad fused-GroupNorm arguments");
    LO_REQUIRE(lo_conv_gn_fuse_tiles(g, &mts, &nt) && mts == gf->MTs && nt == gf->NT,
               "lo_conv_run: fused GroupNorm asked for a geometry / tile grid the kernel does not have (check lo_conv_gn_fuse_tiles)");
    a.gf = *gf;
  }
  a.in = in; a.w = wp; a.bias = bias; a.add_src = add_src; a.out = out; a.gn_partial = gn_partial; a.slab = slab;
  a.gb_v = gb ? gb->v : nullptr; a.gb_stats = gb ? gb->stats : nullptr; a.gb_gamma = gb ? gb->gamma : nullptr;
  a.gb_beta = gb ? gb->beta : nullptr; a.gb_P1 = gb ? gb->P1 : nullptr;
  a.gb_dv = gb ? gb->dv : nullptr; a.gb_P2 = gb ? gb->P2 : nullptr; a.gb_counter = gb ? gb->counter : nullptr;
  a.gb_target = gb ? gb->target : 0u; a.gb_fail = gb ? gb->fail : nullptr; a.gb_keep_out = gb && gb->keep_out ? 1 : 0;
  if (gb && gb->dv) {
    int mts = 0, nt = 0;
    LO_REQUIRE(gb->P2 && gb->counter && gb->fail && lo_conv_gnb_apply_tiles(g, &mts, &nt),
               "lo_conv_run: fused GroupNorm-backward apply asked for a geometry the kernel does not support (check lo_conv_gnb_apply_tiles)");
  }
  a.act = ex ? ex->act : 0; a.bn_partial = ex ? ex->bn_partial : nullptr;
  a.out_pitch = ex ? ex->out_pitch : 0; a.out_choff = ex ? ex->out_choff : 0;
  LO_REQUIRE(a.out_pitch == 0 || (!add_src && !gb && a.out_pitch % 8 == 0 && a.out_choff % 8 == 0), "lo_conv_run: bad concatenated-output arguments");
  a.f8_scale = nullptr;
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
  if (a.nsplit == 1 && !add_src && !gb && !ex && !gf && lo_convt4_patch_applies(g))
    return lo_convt4_patch_run(g, in, wp, bias, out, gn_partial, st);   // last transposed conv of the decoder, patch-resident
  {
    // data gradient of the 64 -> 128 stride-2 conv, patch-resident, where the grid fills at least half the chip (batch >= 32):
    // +0.65 % on the step against the four-phase lo_igemm_nt launch (24 403-24 510 against 24 257-24 324 sprites/s, interleaved)
    if (a.nsplit == 1 && !bias && !gn_partial && !gb && !ex && !gf && lo_convs2d_patch_applies(g) && g.B * (g.Hin / 16) * (g.Win / 16) >= 128)
      return lo_convs2d_patch_run(g, in, wp, add_src, out, st);
  }
  if (a.nsplit == 1 && lo_conv3_tiles_per_image(g, ex != nullptr) > 0 && (!gb || lo_conv3_fuses_gnb(g)))
    return lo_conv3_run(g, in, wp, bias, add_src, out, gn_partial, st, ex, gb, gf);   // fused-tap kernel for 3x3 stride-1
  if (a.nsplit > 1) {
    LO_REQUIRE(g.n_phase == 1 && slab, "lo_conv_run: split-K needs a single phase and a slab");
    LO_REQUIRE(BK == 64 && g.Cout % 64 == 0, "lo_conv_run: split-K path needs Cin%%64==0 and Cout%%64==0");
    // convolutions with few output rows (the 8 x 8 stage: M = 4 096): 128 x 128 tiles halve the L2 -> LDS operand traffic of the
    // 64 x 64 tiles that the tile heuristic would need to fill the chip, and the K split fills it instead (lo_conv_splitk_plan)
    if (g.T[0] > 1 && a.M % 128 == 0 && g.Cout % 128 == 0) return launch_igemm<128, 128, 64>(a, st);
    return launch_igemm<64, 64, 64>(a, st);
  }
  // tile choice
  int bm, bn;
  lo_conv_pick_tile(g, &bm, &bn);
  if (gn_partial || gb || gf) {
    LO_REQUIRE(per_sample % bm == 0 && a.M % bm == 0, "lo_conv_run: GN partials need whole tiles per sample");
    LO_REQUIRE((g.Cout >> 3) <= bn, "lo_conv_run: GroupNorm group wider than the N tile");
  }
  if (BK == 32) {
    LO_REQUIRE(g.Cout % 64 == 0, "lo_conv_run: BK=32 path needs Cout%%64==0");
    if (bm == 128) return launch_igemm<128, 64, 32>(a, st);
    return launch_igemm<64, 64, 32>(a, st);
  }
  if (bm == 128 && bn == 128) return launch_igemm<128, 128, 64>(a, st);
  if (bm == 128 && bn == 64) return launch_igemm<128, 64, 64>(a, st);
  if (bm == 64 && bn == 128) return launch_igemm<64, 128, 64>(a, st);
  if (bm == 64 && bn == 64) return launch_igemm<64, 64, 64>(a, st);
  if (bm == 128 && bn == 32) return launch_igemm<128, 32, 64>(a, st);
  if (bm == 64 && bn == 32) return launch_igemm<64, 32, 64>(a, st);
  lo_set_error("lo_conv_run: no kernel for tile %dx%d", bm, bn);
  return LO_ERR_ARG;
}

// Largest tile (BM=128 preferred) that still gives >= 512 workgroups (two per CU); measured on MI355X at B=64:
// 128x128 wins at M=65536/N=128, 128x64 at M=16384/N=256, 64x64 at M=4096/N=512.
void lo_conv_pick_tile(const LoGeom& g, int* bm_out, int* bn_out) {
  const int per_sample = g.GH * g.GW;
  const size_t M = (size_t)g.B * per_sample;
  const int BK = (g.Cin % 64 == 0) ? 64 : 32;
  constexpr int min_wgs = 512;     // 256 / 384 / 512 / 768 / 1024 swept twice (rounds 1, 2): a plateau, 512 kept
  const int cand[4][2] = {{128, 128}, {128, 64}, {64, 128}, {64, 64}};
  int bm = 64, bn = (g.Cout % 64 == 0) ? 64 : 32;
  if (BK == 32) {
    bn = 64;
    bm = (per_sample % 128 == 0 && (M / 128) * (g.Cout / 64) * g.n_phase >= (size_t)min_wgs) ? 128 : 64;
  } else if (g.Cout % 64 != 0) {
    bm = (per_sample % 128 == 0 && (M / 128) * (g.Cout / 32) * g.n_phase >= (size_t)min_wgs) ? 128 : 64;
  } else {
    for (int i = 0; i < 4; ++i) {
      int cm = cand[i][0], cn = cand[i][1];
      if (per_sample % cm || g.Cout % cn || (g.Cout >> 3) > cn) continue;   // whole tiles per sample; GN group inside a tile
      bm = cm; bn = cn;
      if (((M + cm - 1) / cm) * (g.Cout / cn) * g.n_phase >= (size_t)min_wgs) break;
    }
  }
  *bm_out = bm; *bn_out = bn;
}
int lo_conv_tile_m(const LoGeom& g) {
  int bm, bn;
  lo_conv_pick_tile(g, &bm, &bn);
  return bm;
}
int lo_conv_bn_rows(const LoGeom& g) {
  int t = lo_conv3_tiles_per_image(g, true);
  if (t > 0) return g.B * t;
  return (int)(((size_t)g.B * g.GH * g.GW / lo_conv_tile_m(g)) * g.n_phase);
}
int lo_conv_gnb_rows(const LoGeom& g) {
  if (lo_conv3_fuses_gnb(g)) return lo_conv3_tiles_per_image(g, false);
  return (g.GH * g.GW / lo_conv_tile_m(g)) * g.n_phase;
}
// GroupNorm partial rows per sample written by lo_conv_run_f8 (always lo_igemm_nt's tiles, whatever fp16 kernel owns the geometry)
int lo_conv_mt_f8(const LoGeom& g) { return (g.GH * g.GW / lo_conv_tile_m(g)) * g.n_phase; }
int lo_conv_mt(const LoGeom& g) {
  int t = lo_conv3_tiles_per_image(g);
  if (t > 0) return t;
  t = lo_convt4_patch_tiles_per_image(g);
  if (t > 0) return t;
  return (g.GH * g.GW / lo_conv_tile_m(g)) * g.n_phase;
}

int lo_splitk_reduce(const float* slab, const float* bias, float* out32, f16* out16, int M, int N, int nsplit,
                     hipStream_t st) {
  int total = M * N;
  LoProfScope _p("lo_splitk_reduce", 0, 4.0 * total * (nsplit + 1), st);
  hipLaunchKernelGGL(lo_splitk_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, st, slab, bias, out32, out16, M, N, nsplit);
  LO_LAUNCH_CHECK("splitk_reduce");
  return LO_OK;
}

static inline int wgrad_bmw(const LoGeom& g) { return g.Cout % 128 == 0 ? 128 : 64; }
static inline int wgrad_bnw(const LoGeom& g) { return g.Cin % 128 == 0 ? 128 : 64; }

static inline int wgrad_bkp(const LoGeom& g) { return ((long)g.B * g.GH * g.GW) % 64 == 0 && (long)g.B * g.GH * g.GW >= 1024 ? 64 : 32; }

// LO_WGRAD_S2=0: the stride-2 layers' weight gradients through the per-tap kernel lo_wgrad_tn (the round-2 path; A/B)
static bool lo_wgrad_s2_enabled() {
  static const bool on = [] { const char* e = getenv("LO_WGRAD_S2"); return !(e && atoi(e) == 0); }();
  return on;
}

// number of pixel splits the wgrad launcher will use for this geometry (callers size the slab with it)
int lo_wgrad_nsplit(const LoGeom& g) {
  if (int n3 = lo_wgrad3_nsplit(g)) return n3;   // multi-tap kernel (3x3 stride 1)
  if (lo_wgrad_s2_enabled())
    if (int n2 = lo_wgrad2_nsplit(g)) return n2; // multi-tap kernel of the stride-2 layers (k3 s2, transposed k4 s2)
  int bmw = wgrad_bmw(g), bnw = wgrad_bnw(g), bkp = wgrad_bkp(g);
  int taps = 0;
  for (int p = 0; p < g.n_phase; ++p) taps += g.T[p];
  long tiles = (long)((g.Cout + bmw - 1) / bmw) * (g.Cin / bnw) * taps;
  int M = g.B * g.GH * g.GW;
  int ms_total = (M + bkp - 1) / bkp;
  constexpr int target = 768;      // 512 / 768 / 1024 swept in round 2: inside +-0.5 %
  long want = (target + tiles - 1) / tiles;
  // every split writes (and the reduce pass re-reads) one fp32 slab: keep the slab traffic under ~24 MB per launch,
  // but never go below one workgroup per CU
  const long slab_bytes = (long)geom_packed_elems(g) * 4;
  long cap = (24L << 20) / (slab_bytes > 0 ? slab_bytes : 1);
  long floor_wgs = (256 + tiles - 1) / tiles;
  if (cap < floor_wgs) cap = floor_wgs;
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  if (want > ms_total / 8) want = ms_total / 8 > 0 ? ms_total / 8 : 1;   // at least 8 K steps per split
  if (want > 256) want = 256;
  return (int)want;
}

// (one reduction launch for several layers -- a slab per layer, job table in the workspace -- was built in round 3 and measured 1 %
// SLOWER on the step in both forms tried, all layers at the end of the backward and one launch per stage: the per-layer launch
// right behind its GEMM finds the slab in the Infinity Cache, the merged one re-reads up to 0.2 GB from HBM in front of the join)
int lo_wgrad_run(const LoGeom& g, const f16* x, const f16* dy, float* slab, float* grad, float scale, hipStream_t st) {
  if (lo_wgrad3_nsplit(g) > 0) {
    int nsplit = 0;
    const int total = geom_packed_elems(g);
    {
      LoProfScope _p(lo_prof_geom_name("lo_wgrad3x3_mt", g), geom_flops(g), geom_bytes(g), st);
      int r = lo_wgrad3_run(g, x, dy, slab, st, &nsplit);
      if (r != LO_OK) return r;
    }
    LoProfScope _p2(lo_prof_geom_name("lo_wgrad_reduce", g), 0, 4.0 * total * (nsplit + 1), st);
    return lo_wgrad_reduce_launch(slab, grad, g, total, nsplit, scale, st);
  }
  if (lo_wgrad_s2_enabled() && lo_wgrad2_nsplit(g) > 0) {
    int nsplit = 0;
    const int total = geom_packed_elems(g);
    {
      LoProfScope _p(lo_prof_geom_name("lo_wgrad_s2_mt", g), geom_flops(g), geom_bytes(g), st);
      int r = lo_wgrad2_run(g, x, dy, slab, st, &nsplit);
      if (r != LO_OK) return r;
    }
    LoProfScope _p2(lo_prof_geom_name("lo_wgrad_reduce", g), 0, 4.0 * total * (nsplit + 1), st);
    return lo_wgrad_reduce_launch(slab, grad, g, total, nsplit, scale, st);
  }
  WgradArgs a;
  a.x = x; a.dy = dy; a.slab = slab; a.g = g; a.grad = grad; a.scale = scale;
  a.M = g.B * g.GH * g.GW;
  a.packed_elems = geom_packed_elems(g);
  a.nsplit = lo_wgrad_nsplit(g);
  a.direct = (a.nsplit == 1 && g.sc == 1) ? 1 : 0;
  const int bkp = wgrad_bkp(g);
  int ms_total = (a.M + bkp - 1) / bkp;
  a.msteps_per_split = (ms_total + a.nsplit - 1) / a.nsplit;
  LO_REQUIRE(g.Cin % 64 == 0 && g.Cout % 32 == 0, "lo_wgrad_run: need Cin %% 64 == 0 and Cout %% 32 == 0 (Cin=%d Cout=%d)", g.Cin, g.Cout);
  LO_REQUIRE(g.lg_hin >= 0 && g.lg_win >= 0 && g.lg_cin >= 0 && g.lg_hout >= 0 && g.lg_wout >= 0,
             "lo_wgrad_run: tensor dims must be powers of two");
  int taps = 0;
  for (int p = 0; p < g.n_phase; ++p) taps += g.T[p];
  int bmw = wgrad_bmw(g), bnw = wgrad_bnw(g);
  a.taps = taps;
  dim3 grid(((g.Cout + bmw - 1) / bmw) * (g.Cin / bnw) * taps * a.nsplit);
  constexpr int wg_stages = 3;     // 64-pixel steps: LDS stages (3 where the tile fits: +0.9 % on the step over 2)
  {
    LoProfScope _p(lo_prof_geom_name("lo_wgrad_tn", g), geom_flops(g), geom_bytes(g), st);
#define LO_WG(BMW, BNW)                                                                            \
  do {                                                                                             \
    if (bkp == 64 && wg_stages == 3 && (BMW + BNW) <= 192) hipLaunchKernelGGL((lo_wgrad_tn<BMW, BNW, 3, 64>), grid, dim3(256), 0, st, a);   \
    else if (bkp == 64) hipLaunchKernelGGL((lo_wgrad_tn<BMW, BNW, 2, 64>), grid, dim3(256), 0, st, a);   \
    else hipLaunchKernelGGL((lo_wgrad_tn<BMW, BNW, 3, 32>), grid, dim3(256), 0, st, a);             \
  } while (0)
    if (bmw == 128 && bnw == 128) LO_WG(128, 128);
    else if (bmw == 128 && bnw == 64) LO_WG(128, 64);
    else if (bmw == 64 && bnw == 128) LO_WG(64, 128);
    else LO_WG(64, 64);
#undef LO_WG
  }
  LO_LAUNCH_CHECK("wgrad_tn");
  if (a.direct) return LO_OK;
  int total = a.packed_elems;
  LoProfScope _p2(lo_prof_geom_name("lo_wgrad_reduce", g), 0, 4.0 * total * (a.nsplit + 1), st);
  return lo_wgrad_reduce_launch(slab, grad, g, total, a.nsplit, scale, st);
}

size_t lo_wgrad_slab_bytes(const LoGeom& g) { return (size_t)lo_wgrad_nsplit(g) * geom_packed_elems(g) * sizeof(float); }
size_t lo_packed_weight_elems(const LoGeom& g) { return (size_t)geom_packed_elems(g); }
