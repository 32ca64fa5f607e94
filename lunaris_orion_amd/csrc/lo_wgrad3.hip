// Multi-tap weight gradient of the 3x3 stride-1 convolutions (ResBlock convs; convolution_backward's
// weight output at lunar_generate.py:36,41):
//     dW[co][ci][r][s] = sum_{n,y,x} dy[n,y,x,co] * x[n, y+r-1, x+s-1, ci]
// lo_wgrad_tn gives every tap its own workgroup, so each dy tile and each (shifted) x tile crosses the L2 -> LDS path
// nine times and every fragment pair feeds ONE MFMA (2 LDS reads per MFMA: the kernel is LDS-read bound).  Here a
// workgroup owns 64 output x 64 input channels for ALL nine taps over a range of 32-pixel chunks: per chunk it stages the
// dy rows (32 x 64) and the x patch with its halo ((TH+2) x (TW+2) x 64) once, reads the two dy fragments once and runs
// the nine taps from shifted transposed reads of the same patch: 40 LDS reads per 36 MFMAs, 13 KB of LDS-DMA per 9.4 MFLOP.
//   8 waves = 2 groups x 4: group g takes the chunks q_begin + 2*it + g, each group with its own LDS stages and its own
//   accumulators (so every SIMD hosts two waves without doubling the slab traffic); group 0 issues its LDS-DMA before
//   its MFMAs, group 1 after them, so one wave of a SIMD issues DMA / addresses while the other one feeds the matrix pipe
//   wave w of a group: output-channel half (w & 1), input-channel half (w >> 1), 9 taps x 2 x 2 accumulator fragments
//   K loop: chunk = TH x TW pixels (2 x 16, or 4 x 8 for 8-pixel-wide maps), 4 LDS stages per group, 3 chunks in flight
//   end: group 1 hands its accumulators to group 0 through LDS (fixed order), group 0 writes the fp32 slab
//   output: fp32 slab [split][packed weight layout], summed in fixed order by lo_wgrad_reduce_kernel (reproducible)
#include "lo_internal.h"
#include <stdlib.h>

__device__ __attribute__((aligned(256))) unsigned int lo_zero_page_w3[64];
#define LO_VMCNT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")

struct Wgrad3Args {
  const f16* x;     // forward input  [B,H,W,Cin]
  const f16* dy;    // gradient of the forward output [B,H,W,Cout]
  float* slab;      // [nsplit][Cout*9*Cin]
  int B, H, W, Cin, Cout;
  int nsplit, chunks_per_split, nchunks;
  int packed_elems;
#ifdef LO_STAMPS
  unsigned long long* stamps;   // diagnostic build only (tools/wgrad3_stamp.cpp)
#endif
};
#ifdef LO_STAMPS
unsigned long long* g_lo_wgrad3_stamps = nullptr;
#define LO_T() __builtin_amdgcn_s_memtime()
#endif

__device__ __forceinline__ int lo_xcd_remap_w3(int bid, int total) {
  const int q = total >> 3, r = total & 7;
  const int xcd = bid & 7, loc = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

// 128-byte rows, 32-byte blocks XOR-swizzled by (row >> 1) & 3: the 8 consecutive rows a 32-lane half of a transposed
// read touches land on 8 distinct 32-byte slots of the 256-byte bank row whatever the first row is
__device__ __forceinline__ int lo_w3_swz(int row) { return (row >> 1) & 3; }

template <int TH, int TW>
__global__ __launch_bounds__(512) void lo_wgrad3x3_mt(Wgrad3Args a) {
  static_assert(TH * TW == 32 && TW % 8 == 0, "a chunk is one 32-pixel K step");
  constexpr int PWP = TW + 2, PROWS = (TH + 2) * PWP;
  constexpr int PQ = (PROWS + 7) / 8;            // LDS-DMA instructions of the patch (8 rows each)
  constexpr int PJ = (PQ + 3) / 4;               // ... per wave (wave w issues q = j*4 + w)
  constexpr int A_BYTES = 32 * 128;              // dy rows
  constexpr int P_BYTES = PQ * 1024;
  constexpr int STAGE = A_BYTES + P_BYTES;
  constexpr int NSTAGE = 4, D = 3;
  constexpr int XCHG_BYTES = 4 * 144 * 64 * 4;   // group 1 -> group 0 accumulator hand-over
  constexpr int LDS_BYTES = 2 * NSTAGE * STAGE > XCHG_BYTES ? 2 * NSTAGE * STAGE : XCHG_BYTES;
  __shared__ __attribute__((aligned(1024))) unsigned char smem_all[LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave8 >> 2, wave = wave8 & 3;
  unsigned char* const smem = smem_all + grp * (NSTAGE * STAGE);
  const int wco = wave & 1, wci = wave >> 1;
  const int tiles_ci = a.Cin >> 6, tiles = (a.Cout >> 6) * tiles_ci;
  const int wid = lo_xcd_remap_w3(blockIdx.x, gridDim.x);
  const int tile = wid % tiles, split = wid / tiles;       // tiles of one pixel range are adjacent: they share dy / x in L2
  const int co0 = (tile / tiles_ci) * 64, ci0 = (tile % tiles_ci) * 64;
  const int q_begin = split * a.chunks_per_split;
  const int q_end = min(a.nchunks, q_begin + a.chunks_per_split);
  const int nq = q_end - q_begin;
  const int chunks_x = a.W / TW, chunks_img = chunks_x * (a.H / TH);
  const f16* zpage = reinterpret_cast<const f16*>(lo_zero_page_w3);

  // ---- per-lane DMA constants.  dy: one instruction per wave (rows wave*8 .. +7 of the 32); patch: q = j*4 + wave
  const int drow = wave * 8 + (lane >> 3), dpos = lane & 7;
  const int d_ty = drow / TW, d_tx = drow % TW;
  const int d_lane = (d_ty * a.W + d_tx) * a.Cout + co0 + ((((dpos >> 1) ^ lo_w3_swz(drow)) << 1) | (dpos & 1)) * 8;
  int p_py[PJ], p_px[PJ], p_lane[PJ];
#pragma unroll
  for (int j = 0; j < PJ; ++j) {
    const int prow = (j * 4 + wave) * 8 + (lane >> 3), pos = lane & 7;
    const int py = prow / PWP, px = prow - py * PWP;
    p_py[j] = prow < PROWS ? py - 1 : -100000;             // rows past the patch read zeros
    p_px[j] = px - 1;
    p_lane[j] = ((py - 1) * a.W + (px - 1)) * a.Cin + ci0 + ((((pos >> 1) ^ lo_w3_swz(prow)) << 1) | (pos & 1)) * 8;
  }
  // next chunk this group stages: (image, y0, x0), advanced by two chunks per call without divisions
  int nx_q = q_begin + grp;
  int nx_n = nx_q / chunks_img, nx_y, nx_x;
  {
    const int r = nx_q - nx_n * chunks_img;
    nx_y = (r / chunks_x) * TH;
    nx_x = (r % chunks_x) * TW;
  }
  auto issue = [&](int stage) __attribute__((always_inline)) {
    unsigned char* sa = smem + stage * STAGE;
    unsigned char* sp = sa + A_BYTES;
    const bool live = nx_q < q_end;
    const int y0 = nx_y, x0 = nx_x;
    const int pix0 = (nx_n * a.H + y0) * a.W + x0;
    {
      const f16* src = live ? a.dy + ((size_t)pix0 * a.Cout + d_lane) : zpage;
      lo_dma16(src, (unsigned int)(size_t)(sa + wave * 1024));
    }
#pragma unroll
    for (int j = 0; j < PJ; ++j) {
      if (j * 4 + wave < PQ) {                              // wave-uniform
        const bool ok = live && (unsigned)(y0 + p_py[j]) < (unsigned)a.H && (unsigned)(x0 + p_px[j]) < (unsigned)a.W;
        const f16* src = ok ? a.x + ((size_t)pix0 * a.Cin + p_lane[j]) : zpage;
        lo_dma16(src, (unsigned int)(size_t)(sp + (j * 4 + wave) * 1024));
      }
    }
    nx_q += 2;
    nx_x += 2 * TW;
    if (nx_x >= a.W) {                                      // W is a multiple of TW; two chunks may wrap once (W == TW: twice)
      nx_x -= a.W; nx_y += TH;
      if (nx_x >= a.W) { nx_x -= a.W; nx_y += TH; }
      if (nx_y >= a.H) { nx_y -= a.H; ++nx_n; if (nx_y >= a.H) { nx_y -= a.H; ++nx_n; } }
    }
  };
  // LDS-DMA instructions this wave issues per chunk (wave-uniform, known at compile time per wave index)
  int lpt = 1;
#pragma unroll
  for (int j = 0; j < PJ; ++j) lpt += (j * 4 + wave < PQ) ? 1 : 0;

  // ---- transposed fragment reads.  k index kk (0..31) = chunk pixel (kk / TW, kk % TW); lane group g supplies rows
  // 4g .. 4g+3 (first read) and 16 + 4g .. (second read): the same k permutation for both operands.
  const int q16 = lane >> 4, i16 = lane & 15;
  const int trow = 4 * q16 + (i16 >> 2);
  const int tsub = (i16 & 3) * 8;
  int aoff[2][2];                       // dy fragments: [mi][first / second read]
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int R = trow + 16 * h, blk = wco * 2 + mi;
      aoff[mi][h] = R * 128 + ((blk ^ lo_w3_swz(R)) * 32) + tsub;
    }
  // patch fragments: row of k index kk under tap (r, s) = (kk / TW + r) * PWP + kk % TW + s; [tap][first / second read],
  // input-channel fragment ni = 0; ni = 1 is the same address ^ 32
  int boff[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int kk = trow + 16 * h;
      const int R = (kk / TW + t / 3) * PWP + kk % TW + t % 3;
      boff[t][h] = A_BYTES + R * 128 + (((wci * 2) ^ lo_w3_swz(R)) * 32) + tsub;
    }

  f32x4 acc[9][2][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) acc[t][mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto tr = [&](const unsigned char* p) __attribute__((always_inline)) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4f16((h16x4 __attribute__((address_space(3)))*)p);
  };
  auto frag = [&](const unsigned char* lo_p, const unsigned char* hi_p) __attribute__((always_inline)) {
    h16x4 lo = tr(lo_p), hi = tr(hi_p);
    return (f16x8){(f16)lo[0], (f16)lo[1], (f16)lo[2], (f16)lo[3], (f16)hi[0], (f16)hi[1], (f16)hi[2], (f16)hi[3]};
  };

  const int nit = (nq + 1) >> 1;
  auto compute = [&](int rs) __attribute__((always_inline)) {
    const unsigned char* sbase = smem + rs * STAGE;
    f16x8 af[2], bf[2][3][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) af[mi] = frag(sbase + aoff[mi][0], sbase + aoff[mi][1]);
#pragma unroll
    for (int tt = 0; tt < 3; ++tt)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) bf[0][tt][ni] = frag(sbase + (boff[tt][0] ^ (ni * 32)), sbase + (boff[tt][1] ^ (ni * 32)));
#pragma unroll
    for (int gq = 0; gq < 3; ++gq) {
      if (gq < 2) {
#pragma unroll
        for (int tt = 0; tt < 3; ++tt)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            bf[(gq + 1) & 1][tt][ni] = frag(sbase + (boff[3 * gq + 3 + tt][0] ^ (ni * 32)), sbase + (boff[3 * gq + 3 + tt][1] ^ (ni * 32)));
      }
#pragma unroll
      for (int tt = 0; tt < 3; ++tt)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            acc[3 * gq + tt][mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mi], bf[gq & 1][tt][ni], acc[3 * gq + tt][mi][ni], 0, 0, 0);
    }
  };
#ifdef LO_STAMPS
  const unsigned long long t_start = LO_T();
  unsigned long long t_loop = t_start, acc_w = 0, acc_i = 0, acc_c = 0;
#endif
  if (nit > 0) {
#pragma unroll
    for (int s = 0; s < D; ++s) issue(s);
#ifdef LO_STAMPS
    t_loop = LO_T();
#endif
    int rs = 0, ws = D % NSTAGE;
    for (int it = 0; it < nit; ++it) {
      // chunk `it` of this group has landed (this wave's share): the two newest chunks may still be in flight
#ifdef LO_STAMPS
      const unsigned long long ta = LO_T();
#endif
      if (lpt == 4) LO_VMCNT(8); else if (lpt == 3) LO_VMCNT(6); else LO_VMCNT(4);
      __builtin_amdgcn_s_barrier();
#ifdef LO_STAMPS
      const unsigned long long tb = LO_T();
#endif
      if (grp == 0) issue(ws);
#ifdef LO_STAMPS
      const unsigned long long tc = LO_T();
#endif
      compute(rs);
#ifdef LO_STAMPS
      asm volatile("s_nop 0" ::: "memory");
      const unsigned long long td = LO_T();
#endif
      if (grp == 1) issue(ws);
#ifdef LO_STAMPS
      const unsigned long long te = LO_T();
      acc_w += tb - ta; acc_i += (tc - tb) + (te - td); acc_c += td - tc;
#endif
      rs = (rs + 1 == NSTAGE) ? 0 : rs + 1;
      ws = (ws + 1 == NSTAGE) ? 0 : ws + 1;
    }
    LO_VMCNT(0);
  }
#ifdef LO_STAMPS
  const unsigned long long t_loop_end = LO_T();
#endif
  // ---- group 1 -> group 0: accumulators through LDS, [wave][register][lane] (conflict-free), added in a fixed order
  __syncthreads();
  float* xch = reinterpret_cast<float*>(smem_all) + (size_t)wave * 144 * 64 + lane;
  if (grp == 1) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int j = 0; j < 4; ++j) xch[(((t * 2 + mi) * 2 + ni) * 4 + j) * 64] = acc[t][mi][ni][j];
  }
  __syncthreads();
#ifdef LO_STAMPS
  if (a.stamps && lane == 0) {
    unsigned long long* d = a.stamps + ((size_t)blockIdx.x * 8 + wave8) * 16;
    d[0] = t_start; d[1] = t_loop; d[2] = t_loop_end; d[3] = LO_T(); d[4] = acc_w; d[5] = acc_i; d[6] = acc_c;
  }
#endif
  if (grp == 1) return;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][mi][ni][j] += xch[(((t * 2 + mi) * 2 + ni) * 4 + j) * 64];
  // D block (t, mi, ni): lane holds input channel c = ni*16 + (lane & 15), output channels n = mi*16 + 4*(lane >> 4) + j
  float* slab = a.slab + (size_t)split * a.packed_elems;
  const int Ktot = 9 * a.Cin;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int c = ci0 + wci * 32 + ni * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = co0 + wco * 32 + mi * 16 + (lane >> 4) * 4 + j;
          slab[(size_t)n * Ktot + t * a.Cin + c] = acc[t][mi][ni][j];
        }
      }
#ifdef LO_STAMPS
  if (a.stamps && lane == 0) a.stamps[((size_t)blockIdx.x * 8 + wave8) * 16 + 7] = LO_T();
#endif
}

// ---------------------------------------------------------------------------------------------
// launcher side
// ---------------------------------------------------------------------------------------------
static inline bool wgrad3_applies(const LoGeom& g) {
  if (g.n_phase != 1 || g.T[0] != 9 || g.in_stride != 1 || g.out_stride != 1) return false;
  if (g.Cin % 64 || g.Cout % 64 || g.Hin != g.Hout || g.Win != g.Wout) return false;
  if (g.Win % 16 == 0) return g.Hin % 2 == 0;
  return g.Win == 8 && g.Hin % 4 == 0;
}

// pixel splits of the multi-tap kernel for this geometry, 0 when it does not apply
int lo_wgrad3_nsplit(const LoGeom& g) {
  if (!wgrad3_applies(g)) return 0;
  const long tiles = (long)(g.Cout / 64) * (g.Cin / 64);
  const long nchunks = (long)g.B * g.Hin * g.Win / 32;
  // Workgroups per launch: these kernels run on the side stream BESIDE the dependent chain, and every CU one of their 512-thread
  // workgroups holds is lost to the chain.  Round 3, sprites/s on the step (interleaved on one box) | serial time of the 8 launches:
  //   target 256 / floor 256 (one workgroup per CU at least)   21 854-21 994 | 0.240 ms
  //   192 / 192                                                22 375-22 499 | 0.269 ms
  //   256 / 128                                                22 513-22 706 | 0.303 ms   <- shipped
  //   128 / 64                                                 22 544-22 747 | 0.340 ms
  // fewer, longer workgroups (and less slab traffic) win on the step although the launch alone takes longer
  constexpr int target = 256, flo = 128;
  long want = (target + tiles - 1) / tiles;
  const long slab_bytes = (long)g.Cout * 9 * g.Cin * 4;
  long cap = (24L << 20) / slab_bytes;                    // slab traffic (written here, re-read by the reduce pass)
  const long floor_wgs = (flo + tiles - 1) / tiles;        // but not below this many workgroups
  if (cap < floor_wgs) cap = floor_wgs;
  if (want > cap) want = cap;
  if (want > nchunks / 4) want = nchunks / 4 > 0 ? nchunks / 4 : 1;   // at least 4 chunks per split
  if (want < 1) want = 1;
  return (int)want;
}

int lo_wgrad3_run(const LoGeom& g, const f16* x, const f16* dy, float* slab, hipStream_t st, int* nsplit_out) {
  LO_REQUIRE(wgrad3_applies(g), "lo_wgrad3_run: geometry not supported by the multi-tap weight-gradient kernel");
  Wgrad3Args a;
  a.x = x; a.dy = dy; a.slab = slab;
  a.B = g.B; a.H = g.Hin; a.W = g.Win; a.Cin = g.Cin; a.Cout = g.Cout;
  a.nchunks = g.B * g.Hin * g.Win / 32;
  a.nsplit = lo_wgrad3_nsplit(g);
  a.chunks_per_split = (a.nchunks + a.nsplit - 1) / a.nsplit;
  a.nsplit = (a.nchunks + a.chunks_per_split - 1) / a.chunks_per_split;   // no empty splits: every slab is written
  a.packed_elems = g.Cout * 9 * g.Cin;
  *nsplit_out = a.nsplit;
#ifdef LO_STAMPS
  a.stamps = g_lo_wgrad3_stamps;
#endif
  const int tiles = (g.Cout / 64) * (g.Cin / 64);
  dim3 grid(tiles * a.nsplit);
  if (g.Win % 16 == 0) hipLaunchKernelGGL((lo_wgrad3x3_mt<2, 16>), grid, dim3(512), 0, st, a);
  else hipLaunchKernelGGL((lo_wgrad3x3_mt<4, 8>), grid, dim3(512), 0, st, a);
  LO_LAUNCH_CHECK("wgrad3x3_mt");
  return LO_OK;
}
