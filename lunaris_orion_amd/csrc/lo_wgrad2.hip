// Multi-tap weight gradient of the STRIDE-2 layers (the three k3 s2 convolutions of the encoder and the four k4 s2 transposed
// convolutions of the decoder: convolution_backward's weight output at lunar_generate.py:102,109,116 and :169,175,181,187):
//     Conv2d k3 s2 p1          dW[co][ci][r][s] = sum_{n,oy,ox} dy[n,oy,ox,co] * x [n, 2oy-1+r, 2ox-1+s, ci]      r, s in 0..2
//     ConvTranspose2d k4 s2 p1 dW[ci][co][r][s] = sum_{n,iy,ix} x [n,iy,ix,ci] * dy[n, 2iy-1+r, 2ix-1+s, co]      r, s in 0..3
// In both a COARSE tensor A (one row per position p = (n, y, x) of the coarse grid; dy resp. x) meets a FINE tensor F of twice the
// resolution at (2y - 1 + r, 2x - 1 + s).  lo_wgrad_tn gives every (phase, tap) its own workgroup: each A tile and each (shifted) F
// tile crosses the L2 -> LDS path 9 / 16 times (33 FLOP per LDS-DMA byte: 290 TFLOP/s in round 2).  Here -- the scheme of
// lo_wgrad3x3_mt carried over to stride 2 -- a workgroup owns 64 A-channels x 64 F-channels for ALL taps over a range of
// 32-position chunks (4 x 8 coarse positions): per chunk it stages the 32 A rows (4 KB) and the (8 + R - 1) x (16 + R - 1) fine
// patch ONCE (25 / 28 KB; 79 / 140 FLOP per LDS-DMA byte) and runs the 9 / 16 taps from transposed reads of the same patch.
//   patch image in LDS: fine pixel (py, px) at row (py * 2 + (px & 1)) * 10 + (px >> 1) -- columns DE-INTERLEAVED by parity, so
//   that the eight positions kx = 0..7 of a tap (fine column 2 kx + s: one parity) are eight CONSECUTIVE 128-byte rows: the
//   conflict-free pattern of lo_wgrad3x3_mt<4, 8> (32-byte blocks XOR-swizzled by (row >> 1) & 3) carries over unchanged
//   8 waves, all on the same chunk: wave = (A-channel half, F-channel half, tap half); 3 LDS stages, 2 chunks in flight behind
//   counted vmcnt (every wave issues exactly four 1-KB LDS-DMA instructions per chunk) + one barrier per chunk
//   output: fp32 slab [split][packed weight layout of the forward geometry], summed by lo_wgrad_reduce_kernel (fixed order)
#include "lo_internal.h"
#include <stdlib.h>

__device__ __attribute__((aligned(256))) unsigned int lo_zero_page_w2[64];
#define LO_VMCNT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")

struct Wgrad2Args {
  const f16* a;       // coarse tensor [B, Hc, Wc, Ca]
  const f16* f;       // fine tensor   [B, 2Hc, 2Wc, Cf]
  float* slab;        // [nsplit][packed elems]
  int B, Hc, Wc, Ca, Cf;
  int nsplit, chunks_per_split, nchunks;
  int packed_elems;
  int Cin;            // forward Cin (the packed layout's K = taps * Cin)
  int wofs[4];        // CONVT: element offset of the four phases inside the packed weight
};

__device__ __forceinline__ int lo_xcd_remap_w2(int bid, int total) {
  const int q = total >> 3, r = total & 7;
  const int xcd = bid & 7, loc = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}
__device__ __forceinline__ int lo_w2_swz(int row) { return (row >> 1) & 3; }

template <bool CONVT>
__global__ __launch_bounds__(512) void lo_wgrad_s2_mt(Wgrad2Args a) {
  constexpr int R = CONVT ? 4 : 3, NTAP = R * R;
  constexpr int TH = 4, TW = 8;                       // coarse positions per chunk (one 32-deep K step)
  constexpr int PH = 2 * TH + R - 1, PWH = 10;        // fine patch rows; half-columns per parity (2 TW + R - 1 <= 19 columns)
  constexpr int PROWS = PH * 2 * PWH;                 // LDS rows of the patch image
  constexpr int PQ = (PROWS + 7) / 8;                 // LDS-DMA instructions of the patch (8 rows each): 25 / 28
  static_assert(4 + PQ <= 32, "eight waves x four instructions per chunk");
  constexpr int A_BYTES = 32 * 128, P_BYTES = PQ * 1024;
  constexpr int STAGE = A_BYTES + P_BYTES;
  constexpr int NSTAGE = 3, D = 2;
  constexpr int NTW = (NTAP + 1) / 2;                 // taps per wave: 5 / 8
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NSTAGE * STAGE + 1024];   // + a dummy slot for the unused DMA slots

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wa = wave & 1, wf = (wave >> 1) & 1, wt = wave >> 2;
  const int tiles_f = (a.Cf + 63) >> 6, tiles = (a.Ca >> 6) * tiles_f;
  const int wid = lo_xcd_remap_w2(blockIdx.x, gridDim.x);
  const int tile = wid % tiles, split = wid / tiles;
  const int a0 = (tile / tiles_f) * 64, f0 = (tile % tiles_f) * 64;
  const int q_begin = split * a.chunks_per_split;
  const int q_end = min(a.nchunks, q_begin + a.chunks_per_split);
  const int nq = q_end - q_begin;
  const int chunks_x = a.Wc / TW, chunks_img = chunks_x * (a.Hc / TH);
  const int Hf = 2 * a.Hc, Wf = 2 * a.Wc;
  const f16* zpage = reinterpret_cast<const f16*>(lo_zero_page_w2);

  // ---- per-lane constants of this wave's four DMA slots q = wave * 4 + j: q < 4 -> A rows 8q .. 8q+7, else patch rows 8 (q - 4) ..
  int s_kind[4], s_off[4], s_py[4], s_px[4];            // kind 0 = A, 1 = patch, 2 = unused
  unsigned int s_lds[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int q = wave * 4 + j;
    const int row8 = lane >> 3, pos = lane & 7;
    if (q < 4) {
      const int arow = q * 8 + row8;                                             // coarse position of the chunk: (arow / TW, arow % TW)
      s_kind[j] = 0;
      s_py[j] = arow / TW; s_px[j] = arow % TW;
      s_off[j] = (s_py[j] * a.Wc + s_px[j]) * a.Ca + a0 + ((((pos >> 1) ^ lo_w2_swz(arow)) << 1) | (pos & 1)) * 8;
      s_lds[j] = (unsigned int)(q * 1024);
    } else if (q - 4 < PQ) {
      const int prow = (q - 4) * 8 + row8;
      const int py = prow / (2 * PWH), rem = prow - py * (2 * PWH), par = rem / PWH, hx = rem - par * PWH;
      const int px = 2 * hx + par;
      const int ch = f0 + ((((pos >> 1) ^ lo_w2_swz(prow)) << 1) | (pos & 1)) * 8;
      const bool ok = prow < PROWS && px < 2 * TW + R - 1 && ch < a.Cf;
      s_kind[j] = ok ? 1 : 2;
      s_py[j] = py - 1; s_px[j] = px - 1;                                          // relative to (2 y0, 2 x0)
      s_off[j] = ((py - 1) * Wf + (px - 1)) * a.Cf + ch;
      s_lds[j] = (unsigned int)(A_BYTES + (q - 4) * 1024);
    } else {
      s_kind[j] = 2; s_py[j] = s_px[j] = 0; s_off[j] = 0;
      s_lds[j] = (unsigned int)(NSTAGE * STAGE);                                   // the dummy slot (never read)
    }
  }
  int nx_q = q_begin;
  int nx_n = nx_q / chunks_img, nx_y, nx_x;
  {
    const int r = nx_q - nx_n * chunks_img;
    nx_y = (r / chunks_x) * TH;
    nx_x = (r % chunks_x) * TW;
  }
  const unsigned int smem_base = (unsigned int)(size_t)smem;
  auto issue = [&](int stage) __attribute__((always_inline)) {
    const bool live = nx_q < q_end;
    const int y0 = nx_y, x0 = nx_x;
    const size_t abase = ((size_t)(nx_n * a.Hc + y0) * a.Wc + x0) * a.Ca;
    const size_t fbase = ((size_t)(nx_n * Hf + 2 * y0) * Wf + 2 * x0) * a.Cf;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f16* src = zpage;
      if (live && s_kind[j] == 0) src = a.a + abase + s_off[j];
      if (live && s_kind[j] == 1) {
        const int fy = 2 * y0 + s_py[j], fx = 2 * x0 + s_px[j];
        if ((unsigned)fy < (unsigned)Hf && (unsigned)fx < (unsigned)Wf) src = a.f + (ptrdiff_t)fbase + s_off[j];
      }
      const unsigned int dst = s_lds[j] >= (unsigned int)(NSTAGE * STAGE) ? smem_base + s_lds[j] : smem_base + stage * STAGE + s_lds[j];
      lo_dma16(src, dst);
    }
    ++nx_q;
    nx_x += TW;
    if (nx_x >= a.Wc) { nx_x = 0; nx_y += TH; if (nx_y >= a.Hc) { nx_y = 0; ++nx_n; } }
  };

  // ---- transposed fragment reads: k index kk (0..31) = coarse position (kk / TW, kk % TW) of the chunk
  const int q16 = lane >> 4, i16 = lane & 15;
  const int trow = 4 * q16 + (i16 >> 2);
  const int tsub = (i16 & 3) * 8;
  int aoff[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int Rr = trow + 16 * h, blk = wa * 2 + mi;
      aoff[mi][h] = Rr * 128 + ((blk ^ lo_w2_swz(Rr)) * 32) + tsub;
    }
  // patch fragments of this wave's taps t = wt * NTW + tt: fine pixel of k index kk under tap (r, s) = (2 ky + r, 2 kx + s)
  int boff[NTW][2];
#pragma unroll
  for (int tt = 0; tt < NTW; ++tt)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int t = wt * NTW + tt;
      const int tq = t < NTAP ? t : 0;
      const int r = tq / R, s = tq % R;
      const int kk = trow + 16 * h, ky = kk / TW, kx = kk % TW;
      const int prow = ((2 * ky + r) * 2 + (s & 1)) * PWH + kx + (s >> 1);
      boff[tt][h] = A_BYTES + prow * 128 + (((wf * 2) ^ lo_w2_swz(prow)) * 32) + tsub;
    }

  f32x4 acc[NTW][2][2];
#pragma unroll
  for (int tt = 0; tt < NTW; ++tt)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) acc[tt][mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto tr = [&](const unsigned char* p) __attribute__((always_inline)) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4f16((h16x4 __attribute__((address_space(3)))*)p);
  };
  auto frag = [&](const unsigned char* lo_p, const unsigned char* hi_p) __attribute__((always_inline)) {
    h16x4 lo = tr(lo_p), hi = tr(hi_p);
    return (f16x8){(f16)lo[0], (f16)lo[1], (f16)lo[2], (f16)lo[3], (f16)hi[0], (f16)hi[1], (f16)hi[2], (f16)hi[3]};
  };
  const bool f_live = f0 + wf * 32 < a.Cf;           // a 32-channel fine tensor: the upper F half of the tile is padding

  if (nq > 0) {
#pragma unroll
    for (int s = 0; s < D; ++s) issue(s);
    int rs = 0, ws = D % NSTAGE;
    for (int it = 0; it < nq; ++it) {
      LO_VMCNT(4 * (D - 1));                 // this wave's four pieces of chunk `it` have landed; the next chunk may be in flight
      __builtin_amdgcn_s_barrier();
      issue(ws);
      if (f_live) {
        const unsigned char* sbase = smem + rs * STAGE;
        f16x8 af[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) af[mi] = frag(sbase + aoff[mi][0], sbase + aoff[mi][1]);
#pragma unroll
        for (int tt = 0; tt < NTW; ++tt) {
          if (NTAP % 2 == 0 || tt + 1 < NTW || wt == 0) {      // only the odd tap count's last tap is conditional
            f16x8 bf[2];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) bf[ni] = frag(sbase + (boff[tt][0] ^ (ni * 32)), sbase + (boff[tt][1] ^ (ni * 32)));
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
              for (int ni = 0; ni < 2; ++ni)
                acc[tt][mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mi], bf[ni], acc[tt][mi][ni], 0, 0, 0);
          }
        }
      }
      rs = (rs + 1 == NSTAGE) ? 0 : rs + 1;
      ws = (ws + 1 == NSTAGE) ? 0 : ws + 1;
    }
    LO_VMCNT(0);
  }
  if (!f_live) return;
  // D block (tap, mi, ni): lane holds F channel cf = ni*16 + (lane & 15), A channels ca = mi*16 + 4*(lane >> 4) + j
  float* slab = a.slab + (size_t)split * a.packed_elems;
#pragma unroll
  for (int tt = 0; tt < NTW; ++tt) {
    const int t = wt * NTW + tt;
    if (t >= NTAP) continue;
    const int r = t / R, s = t % R;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int cf = f0 + wf * 32 + ni * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int ca = a0 + wa * 32 + mi * 16 + (lane >> 4) * 4 + j;
          size_t idx;
          if (CONVT) {
            // A = x (input channel ci = ca), F = dy (output channel co = cf).  Forward geometry: tap (r, s) belongs to phase
            // p = 2 (r even) + (s even), index (r >> 1) * 2 + (s >> 1) inside it; packed [phase][co][t * Cin + ci]
            const int p = ((r & 1) ? 0 : 2) + ((s & 1) ? 0 : 1), tp = (r >> 1) * 2 + (s >> 1);
            idx = (size_t)a.wofs[p] + (size_t)cf * (4 * a.Cin) + tp * a.Cin + ca;
          } else {
            // A = dy (output channel co = ca), F = x (input channel ci = cf); packed [co][t * Cin + ci], t = r * 3 + s
            idx = (size_t)ca * (9 * a.Cin) + t * a.Cin + cf;
          }
          if (cf < a.Cf) slab[idx] = acc[tt][mi][ni][j];
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------
// launcher side
// ---------------------------------------------------------------------------------------------
static inline bool wgrad2_applies(const LoGeom& g) {
  const bool s2 = g.n_phase == 1 && g.T[0] == 9 && g.in_stride == 2 && g.out_stride == 1;                  // Conv2d k3 s2 p1 forward geometry
  const bool ct = g.n_phase == 4 && g.in_stride == 1 && g.out_stride == 2 && g.T[0] == 4 && g.T[3] == 4;   // ConvTranspose2d k4 s2 p1
  if (!s2 && !ct) return false;
  const int Hc = s2 ? g.Hout : g.Hin, Wc = s2 ? g.Wout : g.Win;
  const int Ca = s2 ? g.Cout : g.Cin, Cf = s2 ? g.Cin : g.Cout;
  return Hc % 4 == 0 && Wc % 8 == 0 && Ca % 64 == 0 && (Cf % 64 == 0 || Cf == 32);
}

int lo_wgrad2_nsplit(const LoGeom& g) {
  if (!wgrad2_applies(g)) return 0;
  const bool s2 = g.n_phase == 1;
  const int Hc = s2 ? g.Hout : g.Hin, Wc = s2 ? g.Wout : g.Win;
  const int Ca = s2 ? g.Cout : g.Cin, Cf = s2 ? g.Cin : g.Cout;
  const long tiles = (long)(Ca / 64) * ((Cf + 63) / 64);
  const long nchunks = (long)g.B * Hc * Wc / 32;
  // workgroups per launch: as in lo_wgrad3_nsplit (same table), few and long.  With the per-tap kernel's policy (>= 256 workgroups)
  // this kernel LOST 1 % on the step against the kernel it replaces (21 014-21 067 against 21 215-21 251) although alone it is
  // faster: a multi-tap tile has 9 / 16 times fewer tiles per layer, so filling 256 CUs took up to 32 position splits and 64 MB of
  // slab per layer.  With a floor of 128: 22 262-22 317 against 21 955-22 019 for the per-tap kernel on the same box (+1.3 %)
  constexpr int tgt = 256, flo = 128, capmb = 24;
  long want = (tgt + tiles - 1) / tiles;
  long packed = 0;
  for (int p = 0; p < g.n_phase; ++p) packed += (long)g.Cout * g.T[p] * g.Cin;
  const long slab_bytes = packed * 4;
  long cap = ((long)capmb << 20) / slab_bytes;                      // slab traffic (written here, re-read by the reduce pass)
  const long floor_wgs = (flo + tiles - 1) / tiles;         // but not below this many workgroups
  if (cap < floor_wgs) cap = floor_wgs;
  if (want > cap) want = cap;
  if (want > nchunks / 4) want = nchunks / 4 > 0 ? nchunks / 4 : 1;   // at least 4 chunks per split
  if (want < 1) want = 1;
  const long cps = (nchunks + want - 1) / want;
  return (int)((nchunks + cps - 1) / cps);                  // no empty splits: every slab is written
}

int lo_wgrad2_run(const LoGeom& g, const f16* x, const f16* dy, float* slab, hipStream_t st, int* nsplit_out) {
  LO_REQUIRE(wgrad2_applies(g), "lo_wgrad2_run: geometry not supported by the stride-2 multi-tap weight-gradient kernel");
  const bool s2 = g.n_phase == 1;
  Wgrad2Args a;
  a.a = s2 ? dy : x; a.f = s2 ? x : dy; a.slab = slab;
  a.B = g.B; a.Hc = s2 ? g.Hout : g.Hin; a.Wc = s2 ? g.Wout : g.Win;
  a.Ca = s2 ? g.Cout : g.Cin; a.Cf = s2 ? g.Cin : g.Cout;
  a.nchunks = g.B * a.Hc * a.Wc / 32;
  a.nsplit = lo_wgrad2_nsplit(g);
  a.chunks_per_split = (a.nchunks + a.nsplit - 1) / a.nsplit;
  a.packed_elems = 0;
  for (int p = 0; p < g.n_phase; ++p) a.packed_elems += g.Cout * g.T[p] * g.Cin;
  a.Cin = g.Cin;
  for (int p = 0; p < 4; ++p) a.wofs[p] = p < g.n_phase ? g.wofs[p] : 0;
  *nsplit_out = a.nsplit;
  const int tiles = (a.Ca / 64) * ((a.Cf + 63) / 64);
  dim3 grid(tiles * a.nsplit);
  if (s2) hipLaunchKernelGGL((lo_wgrad_s2_mt<false>), grid, dim3(512), 0, st, a);
  else hipLaunchKernelGGL((lo_wgrad_s2_mt<true>), grid, dim3(512), 0, st, a);
  LO_LAUNCH_CHECK("wgrad_s2_mt");
  return LO_OK;
}
