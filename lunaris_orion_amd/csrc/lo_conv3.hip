// Fused-tap implicit GEMM for 3x3 stride-1 convolutions (the eight ResBlock convs and their data gradients: 39 % of
// the step's MFMA work).
//
// lo_igemm_nt re-loads the activation tile once per tap: 32 KiB of LDS-DMA per 2.1 MFLOP (64 FLOP/B), which ties the
// kernel to the CU's 64 B/clk vector-memory path.  Here a workgroup owns a TH x TW patch of output pixels, loads the
// (TH+2) x (TW+2) input halo patch ONCE per 64-channel block and runs all nine taps from it; only the weight tile
// streams per tap.  Traffic per (tap, channel-block) step drops from (BM+BN)*128 B to BN*128 B + patch/9.
//
//   LDS:  2 patch buffers (double buffer over channel blocks)  +  NSB-deep ring of weight tiles
//   step s = (cb, tap):  wait(vmcnt) ; barrier ; issue weights(s+D) + one patch piece of cb+1 ; 9-tap MFMAs from LDS
// Same unpadded XOR-swizzled LDS image as lo_igemm_nt (chunk ^= (row >> 1) & 7), same LDS-staged epilogue (bias,
// residual add, GroupNorm partial sums).  Tap offsets come from LoGeom, so the data gradient (mirrored taps) is the
// same kernel.
#include "lo_internal.h"
#include <stdlib.h>

__device__ __attribute__((aligned(256))) unsigned int lo_zero_page3[64];
#define LO_VMCNT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")

__device__ __forceinline__ int lo_xcd_remap3(int bid, int total) {
  const int q = total >> 3, r = total & 7;
  const int xcd = bid & 7, loc = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

struct Conv3Args {
  const f16* in;
  const f16* w;        // packed [Cout][9*Cin]
  const float* bias;
  const f16* add_src;
  f16* out;
  float* gn_partial;   // [B][tiles per image][8][2] or null
  float* bn_partial;   // [B * tiles per image][Cout][2] or null (teacher epilogue)
  int act;             // 1: LeakyReLU(0.2) before the store (teacher epilogue)
  // transform-on-load (lo_conv3x3_pp<..., XF=true>; the teacher's fused block tail): the logical input is
  //   image rows 0..7 : xc [B][8][W][Cin] as stored        all other rows : T_nlev(...T_1(in)...),
  //   T_l(x)[c] = lrelu(x[c] + kx[l][class][c]),  class = 0 interior, 1 left column, 2 right column, 3 bottom row, 4 / 5 its corners
  const f16* xc;
  const f16* kx;       // [nlev][6][Cin] fp16
  int nlev;
  const float* f8_scale;   // lo_conv3x3_pp<..., F8>: [Cout] dequantisation factor (weight row scale / activation scale); in / w are e4m3 bytes
  LoGeom g;
#ifdef LO_STAMPS
  unsigned long long* stamps;   // diagnostic build only (tools/conv3_stamp.cpp): [workgroup][wave][16] shader-clock stamps
#endif
  // fused GroupNorm-backward reduction (lo_conv3x3_pp only; same contract as IgemmArgs::gb_* in lo_conv.hip): the output is the
  // gradient wrt the activation of a conv+GN+Mish layer whose raw conv output is gb_v; P1[n][tile][c] = (sum du, sum du*xhat)
  const f16* gb_v = nullptr;
  const float* gb_stats = nullptr;
  const float* gb_gamma = nullptr;
  const float* gb_beta = nullptr;
  float* gb_P1 = nullptr;
  f16* gb_dv = nullptr;    // != null: the GroupNorm-backward apply runs in this epilogue too (LoGnBwdFuse in lo_internal.h)
  float* gb_P2 = nullptr;
  unsigned int* gb_counter = nullptr;
  unsigned int gb_target = 0;
  unsigned int* gb_fail = nullptr;
  int gb_keep_out = 0;
  LoGnFuse gf = {};        // gf.y != null: GroupNorm + Mish of this output in the epilogue (lo_conv3x3_pp only; lo_common.h)
};
#ifdef LO_STAMPS
unsigned long long* g_lo_conv3_stamps = nullptr;
#define LO_T() __builtin_amdgcn_s_memtime()
#endif

// LDS image of the patch and of the weight tile: rows of 128 B (64 channels), 16-byte chunk c of row r stored at chunk
// c ^ (((r >> 1) & 3) << 1).  A ds_read_b128 is served in four groups of 16 lanes, {0-3, 12-15, 20-27}, {4-11, 16-19,
// 28-31} and the same +32 (MI355X_MICROARCH.md, LDS): a group reads 16 consecutive rows but mixes two k-chunks, c for
// rows 0-3 / 12-15 and c^1 for rows 4-11.  The 8 rows of one parity (one half of the 256-byte bank row) then carry chunk
// (((K0 + i) & 3) << 1) ^ [i in 2..5], i = 0..7: rows i and i+4 share the key and differ in the flip, for ANY first row
// K0 -- sixteen distinct 16-byte slots for every tap shift (the (row >> 1) & 7 key of lo_igemm_nt is conflict-free
// only for even K0, i.e. not for the dx = +-1 taps).
__device__ __forceinline__ int lo_pix16(int fr) { return fr; }
__device__ __forceinline__ int lo_swz3(int row) { return ((row >> 1) & 3) << 1; }

// ---------------------------------------------------------------------------------------------
// Ping-pong variant for long grids: 8 waves, 16x16 output pixels x BN channels per workgroup, one workgroup per CU.
// Waves 0-3 (group 0) and 4-7 (group 1) share the SIMDs pairwise and run the same program one barrier interval apart:
//     interval   2s      2s+1    2s+2
//     group 0    R(s)    M(s)    R(s+1)        R(s) = the 16 ds_read_b128 of step s = (channel block, tap)
//     group 1    M(s-1)  R(s)    M(s)          M(s) = DMA issue for step s+3, then the 32 MFMAs of step s
// so the MFMA pipe of every SIMD is fed by one wave while the other one reads LDS.  Weight ring: 4 stages, 3 steps in
// flight.  Ordering of the LDS-DMA data (counted vmcnt by the issuing wave, then a barrier the reader has passed):
//   group of step s' must be complete before barrier 2s'-1 (group 0 reads it in interval 2s', group 1 in 2s'+1):
//   group 0 waits at the end of M(s'-1) with steps s'+1, s'+2 still in flight, group 1 at the end of R(s'-1) with s'+1.
//   A ring stage is overwritten (issue in M(s), interval >= 2s+1) after barrier 2s, which group 1 reaches only after the
//   lgkmcnt(0) that retires its reads of step s-1 -- the last readers of that stage.
// ---------------------------------------------------------------------------------------------
// F8: both operands are OCP e4m3 bytes (a.in / a.w point at bytes): a patch / weight row is still 128 bytes = 128 channels, a
// step is (tap, 128-channel block) and costs ONE v_mfma_scale_f32_16x16x128_f8f6f4 per 16x16 block instead of two fp16 MFMAs
// (same LDS bytes per step, twice the K, half the MFMA cycles per FLOP).  A lane's 32-byte fragment is the pair of 16-byte
// chunks (fq, fq + 4) of its row -- the SAME two reads as the fp16 k sub-steps, so the conflict-free bank pattern carries over
// (the natural pair (2 fq, 2 fq + 1) touches only every other 16-byte slot per read: 2-way conflicts, measured 289 us per teacher
// conv against 236 with this form); which k a byte is multiplied as does not matter as long as both operands agree, and they do:
// both images are read through the same chunk -> k map.  The epilogue multiplies by f8_scale[n] before the bias.  (Teacher 3x3 convolutions of the dropout path in fp8 mode, BASELINE config 5.)
// ONECB: the launch has a single channel block (Cin == 64 for fp16): the second patch buffer is never filled, and without it two
// workgroups fit on a CU, so that one's prologue / epilogue overlaps the other's tap loop.
// XF: the teacher's folded block tail (levels of lrelu(x + k)) applied to the patch on load
enum { LO_PP_PLAIN = 0, LO_PP_XF = 1, LO_PP_F8 = 2 };     // operand mode: fp16, fp16 with the teacher's transform on load, e4m3
template <int BN, int TH, int MODE, bool ONECB = false>
__global__ __launch_bounds__(512) void lo_conv3x3_pp(Conv3Args a) {
  constexpr int TW = 16;                      // one 16-pixel fragment per tile row
  constexpr bool XF = MODE == LO_PP_XF, F8 = MODE == LO_PP_F8;
  constexpr int ES = F8 ? 1 : 2;              // bytes per operand element
  constexpr int CB = 128 / ES;                // channels per 128-byte row = channel block of one step
  constexpr int NW = 8, NTHR = 512, NSB = 4, D = 3;
  constexpr int BM = TH * TW;
  // patch rows are 20 pixels apart (18 used): the swizzle key of pixel pp + 20*mi is key(pp) ^ (mi & 1) * 4, so the eight
  // patch fragments of a step are two base registers (V, V ^ 64) plus immediates; same for the weight rows (16 apart)
  constexpr int PW = 20, PH = TH + 2, NPIX = PH * PW;
  constexpr int PQ = (NPIX + 7) / 8;          // LDS-DMA instructions per patch (8 rows of 128 B each)
  constexpr int PI = (PQ + NW - 1) / NW;      // ... per wave
  static_assert(PI <= 6, "patch pieces of the next channel block must land 3 steps before its first tap");
  constexpr int PATCH_BYTES = PQ * 1024;
  constexpr int IB = BN / 8 / NW;
  static_assert(IB >= 1, "weight tile smaller than one DMA instruction per wave");
  constexpr int B_BYTES = BN * 128;
  constexpr int WGM = 4, WM = BM / WGM, WN = BN / 2, MI = WM / 16, NI = WN / 16;
  constexpr int OPITCH = BN * 2 + 16;
  constexpr int XF_LEV = 2, XF_C = 128;       // transform constants staged in LDS: [2 levels][6 classes][128 channels] fp16
  constexpr int K_BYTES = XF ? XF_LEV * 6 * XF_C * 2 : 0;
  constexpr int NPB = ONECB ? 1 : 2;          // patch buffers
  constexpr int MAIN_BYTES = NPB * PATCH_BYTES + NSB * B_BYTES + K_BYTES;
  constexpr int EPI_BYTES = BM * OPITCH + NTHR * 64;
  constexpr int LDS_BYTES = MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES];
  unsigned char* const s_patch = smem;
  unsigned char* const s_b = smem + NPB * PATCH_BYTES;
  unsigned char* const s_k = s_b + NSB * B_BYTES;

  const LoGeom& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                  // waves w and w+4 share a SIMD
  const int wm = wave & 3, wn = wave >> 2;
  const int H = g.Hin, W = g.Win, Cin = g.Cin;
  const int tiles_x = W / TW, tiles_y = H / TH, tiles_img = tiles_x * tiles_y;
  const int NT = g.Cout / BN;
  const int tile_id = lo_xcd_remap3(blockIdx.x, gridDim.x);
  const int nt_i = tile_id % NT, pt_i = tile_id / NT;
  const int n_img = pt_i / tiles_img, t_img = pt_i - n_img * tiles_img;
  const int y0 = (t_img / tiles_x) * TH, x0 = (t_img % tiles_x) * TW;
  const int n0 = nt_i * BN;
  const int KCB = Cin / CB;
  const int nsteps = 9 * KCB;
  const int Ktot = 9 * Cin;
  const unsigned char* zpage = reinterpret_cast<const unsigned char*>(lo_zero_page3);
  const unsigned char* inb = reinterpret_cast<const unsigned char*>(a.in);
  const unsigned char* xcb = reinterpret_cast<const unsigned char*>(a.xc);
  const unsigned char* wb = reinterpret_cast<const unsigned char*>(a.w);
  const uint32_t dyc = g.dyc[0], dxc = g.dxc[0];

  int p_src[PI];       // BYTE offsets from here on
  unsigned p_xc = 0;   // XF: bit i = piece i of this lane comes from the compact rows buffer
#pragma unroll
  for (int i = 0; i < PI; ++i) {
    int pp = (wave * PI + i) * 8 + (lane >> 3), pos = lane & 7;
    int py = pp / PW, px = pp - py * PW;
    int iy = y0 - 1 + py, ix = x0 - 1 + px;
    bool ok = pp < NPIX && px < TW + 2 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    p_src[i] = ok ? (((n_img * H + iy) * W + ix) * Cin) * ES + ((pos ^ lo_swz3(pp)) * 16) : -1;
    if (XF && ok && iy < 8) {
      p_src[i] = (((n_img * 8 + iy) * W + ix) * Cin) * ES + ((pos ^ lo_swz3(pp)) * 16);
      p_xc |= 1u << i;
    }
  }
  int b_src[IB];
#pragma unroll
  for (int i = 0; i < IB; ++i) {
    int row = (wave * IB + i) * 8 + (lane >> 3), pos = lane & 7;
    b_src[i] = ((n0 + row) * Ktot) * ES + ((pos ^ lo_swz3(row)) * 16);
  }
  auto issue_patch_piece = [&](int buf, int cb, int piece) __attribute__((always_inline)) {
    // wave-uniform: nothing is issued for a piece past this wave's share or past the last channel block (the vmcnt waits
    // below count weight instructions only, which makes them conservative whenever patch pieces are in flight)
    const int q = wave * PI + piece;
    if (piece < PI && cb < KCB && q < PQ) {
      int off = -1;
#pragma unroll
      for (int i = 0; i < PI; ++i) if (i == piece) off = p_src[i];
      const unsigned char* base = (XF && ((p_xc >> piece) & 1u)) ? xcb : inb;
      const unsigned char* src = off >= 0 ? base + (off + cb * 128) : zpage;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(s_patch + buf * PATCH_BYTES + q * 1024), 16, 0, 0);
    }
  };
  auto issue_b = [&](int stage, int step) __attribute__((always_inline)) {
    const bool live = step < nsteps;          // past the end: same instruction count from the zero page (keeps vmcnt uniform)
    const int cb = step / 9, tap = step - cb * 9;
    const int koff = (tap * Cin) * ES + cb * 128;
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      const unsigned char* src = live ? wb + (b_src[i] + koff) : zpage;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(s_b + stage * B_BYTES + (wave * IB + i) * 1024), 16, 0, 0);
    }
  };

  const int fr = lane & 15, fq = lane >> 4;
  const int pp00 = (wm * MI + 1) * PW + fr + 1;   // patch pixel of fragment 0 at tap offset (0, 0)
  const int R0 = wn * WN + fr;
  const int fch = fq;                                      // 16-byte chunk of this lane's fragment at kk = 0 (kk = 1: fq + 4)
  const int w00 = R0 * 128 + ((fch ^ lo_swz3(R0)) * 16);   // weight fragment (ni = 0, kk = 0) inside a ring stage

  f32x4 acc[NI][MI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};

#ifdef LO_STAMPS
  const unsigned long long t_start = LO_T();
  unsigned long long acc_r = 0, acc_w1 = 0, acc_m = 0, acc_w2 = 0;
#endif
  // XF: in-place transform of a landed patch (all 8 waves, between two barriers): pixels of image rows >= 8 get
  // T_nlev(..T_1(x)); rows 0..7 came from xc already transformed, pixels outside the image stay zero (the padding)
  auto xform_patch = [&](int buf, int cb) __attribute__((always_inline)) {
    unsigned char* pb = s_patch + buf * PATCH_BYTES;
    for (int idx = tid; idx < NPIX * 8; idx += NTHR) {
      const int pp = idx >> 3, pos = idx & 7;
      const int py = pp / PW, px = pp - py * PW;
      const int iy = y0 - 1 + py, ix = x0 - 1 + px;
      if (px < TW + 2 && iy >= 8 && iy < H && (unsigned)ix < (unsigned)W) {
        const int cls = (iy == H - 1 ? 3 : 0) + (ix == 0 ? 1 : (ix == W - 1 ? 2 : 0));
        const int c = cb * 64 + ((pos ^ lo_swz3(pp)) * 8);
        f16x8 v = *reinterpret_cast<const f16x8*>(pb + pp * 128 + pos * 16);
        for (int lev = 0; lev < a.nlev; ++lev) {
          const f16x8 k = *reinterpret_cast<const f16x8*>(s_k + ((lev * 6 + cls) * XF_C + c) * 2);
          v = v + k;
          v = __builtin_elementwise_max(v, v * (f16)0.2f);
        }
        *reinterpret_cast<f16x8*>(pb + pp * 128 + pos * 16) = v;
      }
    }
  };
  if (XF) {
    for (int i = tid; i < a.nlev * 6 * XF_C / 8; i += NTHR)
      *reinterpret_cast<f16x8*>(s_k + i * 16) = *reinterpret_cast<const f16x8*>(a.kx + i * 8);
  }
  // ---- prologue: patch of channel block 0 and weight steps 0..2
#pragma unroll
  for (int i = 0; i < PI; ++i) issue_patch_piece(0, 0, i);
#pragma unroll
  for (int s = 0; s < D; ++s) issue_b(s, s);
  LO_VMCNT(IB * (D - 1));            // patch + step 0 landed (this wave's share)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // XF: the staged constants
  __builtin_amdgcn_s_barrier();
  if (XF) {
    xform_patch(0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  if (grp == 1) __builtin_amdgcn_s_barrier();     // group 1 runs one interval behind
#ifdef LO_STAMPS
  const unsigned long long t_loop = LO_T();
#endif

  int rs = 0, ws = D % NSB;
  int cb = 0, tap = 0;
  for (int step = 0; step < nsteps; ++step) {
#ifdef LO_STAMPS
    const unsigned long long ta = LO_T();
#endif
    // ---- R(step): the fragments of this step into registers, then (under their latency) the DMA for step + 3, whose
    //      ring stage was last read in R(step - 1) and retired before the barrier that ended it
    const int dy = (int)((dyc >> (2 * tap)) & 3u) - 1, dx = (int)((dxc >> (2 * tap)) & 3u) - 1;
    const int pp = pp00 + dy * PW + dx;
    const int v0 = (cb & 1) * PATCH_BYTES + pp * 128 + ((fch ^ lo_swz3(pp)) * 16);
    const int v1 = v0 ^ 64;
    const int u0 = rs * B_BYTES + w00;
    const int u1 = u0 ^ 64;
    // fp16: [kk] = the two 32-wide k sub-steps (chunks fq and fq + 4);  F8: [0] / [1] = the two halves of the one 32-byte fragment
    f16x8 wf[2][NI], xf[2][MI];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) wf[kk][ni] = *reinterpret_cast<const f16x8*>(s_b + (kk ? u1 : u0) + ni * 2048);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        xf[kk][mi] = *reinterpret_cast<const f16x8*>(s_patch + (((kk + mi) & 1) ? v1 : v0) + mi * (PW * 128));
    }
    issue_b(ws, step + D);
    issue_patch_piece((cb + 1) & 1, cb + 1, tap);
    if (grp == 1) LO_VMCNT(2 * IB);              // step+1 landed (this wave's share); steps +2, +3 may be in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef LO_STAMPS
    const unsigned long long tb = LO_T();
#endif
    __builtin_amdgcn_s_barrier();
#ifdef LO_STAMPS
    const unsigned long long tc = LO_T();
#endif
    // ---- M(step)
    __builtin_amdgcn_s_setprio(1);
    if constexpr (F8) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const i32x4 wl = __builtin_bit_cast(i32x4, wf[0][ni]), wh = __builtin_bit_cast(i32x4, wf[1][ni]);
        const i32x8 w8 = {wl[0], wl[1], wl[2], wl[3], wh[0], wh[1], wh[2], wh[3]};
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const i32x4 xl = __builtin_bit_cast(i32x4, xf[0][mi]), xh = __builtin_bit_cast(i32x4, xf[1][mi]);
          const i32x8 x8 = {xl[0], xl[1], xl[2], xl[3], xh[0], xh[1], xh[2], xh[3]};
          // formats 0 / 0 = e4m3 x e4m3; block scales 0x7f = 2^0
          acc[ni][mi] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w8, x8, acc[ni][mi], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[kk][ni], xf[kk][mi], acc[ni][mi], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    if (grp == 0) LO_VMCNT(2 * IB);              // step+1 landed; steps +2, +3 may be in flight
#ifdef LO_STAMPS
    const unsigned long long td = LO_T();
#endif
    __builtin_amdgcn_s_barrier();
#ifdef LO_STAMPS
    const unsigned long long te = LO_T();
    acc_r += tb - ta; acc_w1 += tc - tb; acc_m += td - tc; acc_w2 += te - td;
#endif
    rs = (rs + 1 == NSB) ? 0 : rs + 1;
    ws = (ws + 1 == NSB) ? 0 : ws + 1;
    if (++tap == 9) {
      tap = 0; ++cb;
      if (XF && cb < KCB) {
        // channel-block boundary: every wave waited for its own patch pieces (they are older than the two weight groups a
        // wave may have in flight); re-join the groups, transform the new patch, restart the stagger
        if (grp == 0) __builtin_amdgcn_s_barrier();
        xform_patch(cb & 1, cb);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (grp == 1) __builtin_amdgcn_s_barrier();
      }
    }
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();
#ifdef LO_STAMPS
  const unsigned long long t_loop_end = LO_T();
#endif
  LO_VMCNT(0);
  __syncthreads();

  // ---- epilogue: same LDS-staged form as lo_igemm_nt's
  unsigned char* so = smem;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    int nl = wn * WN + ni * 16 + fq * 4;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) bv = *reinterpret_cast<const f32x4*>(a.bias + n0 + nl);
    f32x4 sv = {1.f, 1.f, 1.f, 1.f};
    if constexpr (F8) sv = *reinterpret_cast<const f32x4*>(a.f8_scale + n0 + nl);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      int ml = wm * WM + mi * 16 + fr;
      f32x4 v = F8 ? acc[ni][mi] * sv + bv : acc[ni][mi] + bv;
      f16x4 h = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
      *reinterpret_cast<f16x4*>(so + ml * OPITCH + nl * 2) = h;
    }
  }
  __syncthreads();
  constexpr int OCPR = BN / 8, ORPP = NTHR / OCPR, OP = BM / ORPP;
  static_assert(OP >= 1 && BM % ORPP == 0, "epilogue row passes");
  const int orow = tid / OCPR, ochunk = tid % OCPR;
  const int G = g.Cout >> 3;
  float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
  float ga1[8], ga2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { ga1[j] = 0.f; ga2[j] = 0.f; }
  float gsc[8], gsh[8];   // fused GN-backward reduction: u = v * gsc + gsh for this thread's 8 channels (the tile lies inside one sample)
  const int my_img = n_img;
  if (a.gb_v) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = n0 + ochunk * 8 + j, gr = c / G;
      const float mean = a.gb_stats[my_img * 16 + gr * 2], rstd = a.gb_stats[my_img * 16 + gr * 2 + 1];
      gsc[j] = a.gb_gamma[c] * rstd;
      gsh[j] = a.gb_beta[c] - mean * gsc[j];
    }
  }
#pragma unroll
  for (int i = 0; i < OP; ++i) {
    int ml = orow + i * ORPP;
    int ty = ml / TW, tx = ml % TW;
    f16x8 h = *reinterpret_cast<const f16x8*>(so + ml * OPITCH + ochunk * 16);
    size_t off = ((size_t)(n_img * H + y0 + ty) * W + x0 + tx) * g.Cout + n0 + ochunk * 8;
    if (a.add_src) {
      f16x8 r = *reinterpret_cast<const f16x8*>(a.add_src + off);
#pragma unroll
      for (int j = 0; j < 8; ++j) h[j] = (f16)((float)h[j] + (float)r[j]);
    }
    if (a.act == 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { float x = (float)h[j]; h[j] = (f16)(x > 0.f ? x : 0.2f * x); }
    }
    if (a.gb_dv) {                                 // fused apply: the activation gradient stays in LDS; stored only if somebody else reads it
      *reinterpret_cast<f16x8*>(so + ml * OPITCH + ochunk * 16) = h;
      if (a.gb_keep_out) *reinterpret_cast<f16x8*>(a.out + off) = h;
    } else *reinterpret_cast<f16x8*>(a.out + off) = h;
    if (a.bn_partial) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { float x = (float)h[j]; ga1[j] += x; ga2[j] += x * x; }
    }
    if (a.gb_v) {
      const f16x8 vv = *reinterpret_cast<const f16x8*>(a.gb_v + off);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float hv = (float)vv[j];
        const float du = (float)h[j] * lo_mish_grad(hv * gsc[j] + gsh[j]);
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
    float* red = reinterpret_cast<float*>(smem + BM * OPITCH);   // [NTHR][16] floats
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[tid * 16 + j * 2] = ga1[j]; red[tid * 16 + j * 2 + 1] = ga2[j]; }
    __syncthreads();
    float* dst = a.bn_partial + ((size_t)pt_i * g.Cout + n0) * 2;
    for (int o = tid; o < BN * 2; o += NTHR) {
      int cl = o >> 1, w = o & 1;
      int ccx = cl >> 3, j = cl & 7;
      float tot = 0.f;
      for (int r = 0; r < ORPP; ++r) tot += red[(r * OCPR + ccx) * 16 + j * 2 + w];
      dst[o] = tot;
    }
  }
  if (a.gb_v) {
    if (a.bn_partial) __syncthreads();   // `red` is reused (no caller sets both; the barrier keeps that legal)
    // the saved mean / rstd are read here, after the store loop, not held in registers across it (same form as lo_igemm_nt)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int gr = (n0 + ochunk * 8 + j) / G;
      const float mean = a.gb_stats[my_img * 16 + gr * 2], rstd = a.gb_stats[my_img * 16 + gr * 2 + 1];
      ga2[j] = rstd * (ga2[j] - mean * ga1[j]);
    }
    float* red = reinterpret_cast<float*>(smem + BM * OPITCH);   // [NTHR][16] floats
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[tid * 16 + j * 2] = ga1[j]; red[tid * 16 + j * 2 + 1] = ga2[j]; }
    __syncthreads();
    float* dst = a.gb_P1 + (((size_t)n_img * tiles_img + t_img) * g.Cout + n0) * 2;
    for (int o = tid; o < BN * 2; o += NTHR) {
      int cl = o >> 1, w = o & 1;
      int ccx = cl >> 3, j = cl & 7;
      float tot = 0.f;
      for (int r = 0; r < ORPP; ++r) tot += red[(r * OCPR + ccx) * 16 + j * 2 + w];
      if (a.gb_dv) __hip_atomic_store(reinterpret_cast<unsigned int*>(dst) + o, __float_as_uint(tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else dst[o] = tot;
    }
    {
      if (a.gb_dv) {
        // ---- GroupNorm-backward APPLY of the producing layer (same sequence as lo_igemm_nt's epilogue; lo_internal.h LoGnBwdFuse)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        lo_arrive_and_wait(a.gb_counter + n_img * NT + nt_i, a.gb_target, a.gb_fail, tid);
        float* s_g = red;
        float* s_c = red + 2 * BN;
        for (int c = tid; c < BN; c += NTHR) {
          float t1 = 0.f, t2 = 0.f;
          const unsigned int* p = reinterpret_cast<const unsigned int*>(a.gb_P1) + ((size_t)n_img * tiles_img * g.Cout + n0 + c) * 2;
          for (int k = 0; k < tiles_img; ++k) {
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
          s_c[tid] = tot / ((float)(H * W) * (float)G);
        }
        __syncthreads();
        const int gl = (ochunk * 8) / G, gr = n0 / G + gl;
        const float mean = a.gb_stats[n_img * 16 + gr * 2], rstd = a.gb_stats[n_img * 16 + gr * 2 + 1];
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
          const int ty = ml / TW, tx = ml % TW;
          const f16x8 d = *reinterpret_cast<const f16x8*>(so + ml * OPITCH + ochunk * 16);
          const size_t off = ((size_t)(n_img * H + y0 + ty) * W + x0 + tx) * g.Cout + n0 + ochunk * 8;
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
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) red[tid * 8 + j] = acc[j];
        __syncthreads();
        for (int c = tid; c < BN; c += NTHR) {
          const int ccx = c >> 3, j = c & 7;
          float tot = 0.f;
          for (int r = 0; r < ORPP; ++r) tot += red[(r * OCPR + ccx) * 8 + j];
          a.gb_P2[((size_t)n_img * tiles_img + t_img) * g.Cout + n0 + c] = tot;
        }
      }
    }
    if (a.gn_partial) __syncthreads();
  }
  if (a.gn_partial || a.gf.y) {
    float* red = reinterpret_cast<float*>(smem + BM * OPITCH);
    red[tid * 4 + 0] = s0; red[tid * 4 + 1] = q0; red[tid * 4 + 2] = s1; red[tid * 4 + 3] = q1;
    constexpr int NV = OCPR * 4, P = NTHR / NV, RPP2 = ORPP / P;
    float* red2 = red + NTHR * 4;
    float* red3 = red2 + NTHR;
    float* s_x = red3 + 2 * NV;           // fused GroupNorm: this tile's exchange line (32 floats), then the sample's statistics (16)
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
      red3[tid] = t;
    }
    __syncthreads();
    const int ngroups = BN / G;
    if (tid < ngroups * 2) {
      int gl = tid >> 1, which = tid & 1;
      int hc_begin = gl * G / 4, hc_end = (gl + 1) * G / 4;
      float tot = 0.f;
      for (int hc = hc_begin; hc < hc_end; ++hc) tot += red3[(hc >> 1) * 4 + (hc & 1) * 2 + which];
      int grp2 = (n0 / G) + gl;
      if (a.gn_partial) a.gn_partial[(((size_t)n_img * tiles_img + t_img) * 8 + grp2) * 2 + which] = tot;
      s_x[grp2 * 2 + which] = tot;
    }
    if constexpr (!F8 && !XF) {
      if (a.gf.y) {
        // ---- GroupNorm + Mish of this tile once the whole sample's sums are known (lo_common.h: LoGnFuse); the tile is still in LDS
        __syncthreads();
        lo_gn_rendezvous(a.gf, n_img, t_img, nt_i, G, BN, 1.0f / ((float)(H * W) * (float)G), s_x, s_stat, tid);
        if (t_img == 0 && nt_i == 0 && tid < 16 && a.gf.stats) a.gf.stats[n_img * 16 + tid] = s_stat[tid];
        float sc[8], sh[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int c = n0 + ochunk * 8 + j, gr = c / G;
          const float mean = s_stat[gr * 2], rstd = s_stat[gr * 2 + 1];
          lo_gn_scale_shift(a.gf.gamma[c], a.gf.beta[c], mean, rstd, sc[j], sh[j]);
        }
#pragma unroll
        for (int i = 0; i < OP; ++i) {
          const int ml = orow + i * ORPP;
          const int ty = ml / TW, tx = ml % TW;
          const f16x8 h = *reinterpret_cast<const f16x8*>(so + ml * OPITCH + ochunk * 16);
          const size_t off = ((size_t)(n_img * H + y0 + ty) * W + x0 + tx) * g.Cout + n0 + ochunk * 8;
          f16x8 o = h;
          if (a.gf.mode != 0) o = *reinterpret_cast<const f16x8*>(a.gf.other + off);
          *reinterpret_cast<f16x8*>(a.gf.y + off) = lo_gn_apply8(h, sc, sh, a.gf.mode, o);
        }
      }
    }
  }
#ifdef LO_STAMPS
  if (a.stamps && lane == 0) {
    unsigned long long* d = a.stamps + ((size_t)blockIdx.x * 8 + wave) * 16;
    d[0] = t_start; d[1] = t_loop; d[2] = t_loop_end; d[3] = LO_T();
    d[4] = acc_r; d[5] = acc_w1; d[6] = acc_m; d[7] = acc_w2;
  }
#endif
}

// ---------------------------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------------------------
// LO_HALO: 0 = never, 2 = default (the shapes where it measured faster than lo_igemm_nt), 3 = every shape it can tile (what the
// forced parity test uses: small batches run on it too)
static inline int conv3_mode() {
  static const int m = getenv("LO_HALO") ? atoi(getenv("LO_HALO")) : 2;
  return m;
}
// tile of the 8-wave ping-pong kernel that gives >= 256 workgroups at batch 64: 16x16 pixels x 128 channels, 8x16 x 128 when that
// would leave CUs idle, 16x16 x 64 for 64 output channels
static inline bool conv3_tile(const LoGeom& g, int* th, int* tw, int* bn) {
  if (g.n_phase != 1 || g.T[0] != 9 || g.in_stride != 1 || g.out_stride != 1) return false;
  if (g.Cin % 64 || g.Cout % 64 || g.Win % 16 || g.Hin % 16) return false;
  const long t16 = (long)g.B * (g.Hin / 16) * (g.Win / 16);
  if (g.Cout % 128 == 0 && t16 * (g.Cout / 128) >= 256) { *th = 16; *tw = 16; *bn = 128; }
  else if (g.Cout % 128 == 0) { *th = 8; *tw = 16; *bn = 128; }
  else if (g.Cout == 64) { *th = 16; *tw = 16; *bn = 64; }
  else return false;
  return (g.Cout >> 3) <= *bn;   // a GroupNorm group must fit inside the N tile
}

// tiles per image of the fused-tap kernel for this geometry, or 0 when the kernel does not apply / is not selected
int lo_conv3_tiles_per_image(const LoGeom& g, bool need_bn) {
  int th, tw, bn;
  const int mode = conv3_mode();
  if (mode == 0 || !conv3_tile(g, &th, &tw, &bn)) return 0;
  if (mode == 2) {
    // default: where it measured faster than lo_igemm_nt -- the 16x16-pixel x 128-channel workgroup on
    // long grids (>= 4 tiles per CU: the teacher), and, for launches without the BatchNorm epilogue, any tile choice above that
    // puts a workgroup on every CU (the VAE's 64 / 128 / 256-channel ResBlock convolutions at batch 64: 44 -> 36, 34 -> 28,
    // 36 -> 33 us; +1.0 % on the step over three interleaved pairs)
    const long tiles = (long)g.B * (g.Hin / th) * (g.Win / tw) * (g.Cout / bn);
    const bool long_grid = th == 16 && bn == 128 && tiles >= 1024;
    if (!(long_grid || (!need_bn && tiles >= 256))) return 0;
  }
  return (g.Hin / th) * (g.Win / tw);
}

// does the kernel lo_conv3_run would launch for g carry the fused GroupNorm-backward epilogue?  (it always does)
bool lo_conv3_fuses_gnb(const LoGeom& g) { return lo_conv3_tiles_per_image(g, false) > 0; }

// fused GroupNorm epilogue (LoGnFuse / LoGnBwdFuse::dv): tile rows per sample and n tiles per row of the launch
bool lo_conv3_gn_fuse_tiles(const LoGeom& g, int* mts, int* nt) {
  int th, tw, bn;
  if (lo_conv3_tiles_per_image(g, false) <= 0 || !conv3_tile(g, &th, &tw, &bn)) return false;
  *mts = (g.Hin / th) * (g.Win / tw);
  *nt = g.Cout / bn;
  return *mts * *nt <= LO_GNF_MAX_TILES;
}

int lo_conv3_run(const LoGeom& g, const f16* in, const f16* wp, const float* bias, const f16* add_src, f16* out,
                 float* gn_partial, hipStream_t st, const LoConvExtra* ex, const LoGnBwdFuse* gb, const LoGnFuse* gf) {
  int th, tw, bn;
  LO_REQUIRE(conv3_tile(g, &th, &tw, &bn), "lo_conv3_run: geometry not supported by the fused-tap kernel");
  if (gf) {
    int mts = 0, nt = 0;
    LO_REQUIRE(!add_src && !gb && !ex && lo_conv3_gn_fuse_tiles(g, &mts, &nt) && mts == gf->MTs && nt == gf->NT,
               "lo_conv3_run: fused GroupNorm asked for a geometry / tile grid the kernel does not have");
  }
  Conv3Args a{in, wp, bias, add_src, out, gn_partial, ex ? ex->bn_partial : nullptr, ex ? ex->act : 0, nullptr, nullptr, 0, nullptr, g};
  if (gb) {
    a.gb_v = gb->v; a.gb_stats = gb->stats; a.gb_gamma = gb->gamma; a.gb_beta = gb->beta; a.gb_P1 = gb->P1;
    a.gb_dv = gb->dv; a.gb_P2 = gb->P2; a.gb_counter = gb->counter; a.gb_target = gb->target; a.gb_fail = gb->fail;
    a.gb_keep_out = gb->keep_out ? 1 : 0;
    LO_REQUIRE(!gb->dv || (g.Cout >> 3) >= 8, "lo_conv3_run: fused GroupNorm-backward apply needs groups of at least 8 channels");
  }
  if (gf) a.gf = *gf;
#ifdef LO_STAMPS
  a.stamps = g_lo_conv3_stamps;
#endif
  const int tiles = g.B * (g.Hin / th) * (g.Win / tw) * (g.Cout / bn);
  double flops = 2.0 * g.B * g.Hin * g.Win * (double)g.Cout * 9 * g.Cin;
  double bytes = 2.0 * ((double)g.B * g.Hin * g.Win * (g.Cin + g.Cout) + 9.0 * g.Cin * g.Cout);
  static char name[64];
  snprintf(name, sizeof(name), "lo_conv3x3_pp<%d,%dx%d>", bn, th, tw);
  // algorithmic bytes of the FUSED op, as in launch_igemm: the residual gradient it adds and the producing layer's raw conv output
  // that the fused GroupNorm-backward reduction reads, each the size of the output
  if (add_src) bytes += 2.0 * g.B * g.Hin * g.Win * g.Cout;
  if (gb) bytes += 2.0 * g.B * g.Hin * g.Win * g.Cout;
  LoProfScope _p(lo_prof_intern(name), flops, bytes, st);
  if (th == 8) LO_LAUNCH_STOP((lo_conv3x3_pp<128, 8, LO_PP_PLAIN>), dim3(tiles), dim3(512), 0, st, a);
  else if (bn == 64 && g.Cin == 64) LO_LAUNCH_STOP((lo_conv3x3_pp<64, 16, LO_PP_PLAIN, true>), dim3(tiles), dim3(512), 0, st, a);
  else if (bn == 64) LO_LAUNCH_STOP((lo_conv3x3_pp<64, 16, LO_PP_PLAIN>), dim3(tiles), dim3(512), 0, st, a);
  else LO_LAUNCH_STOP((lo_conv3x3_pp<128, 16, LO_PP_PLAIN>), dim3(tiles), dim3(512), 0, st, a);
  LO_LAUNCH_CHECK("conv3x3_pp");
  return LO_OK;
}

// The 16x16-pixel fused-tap kernel with the transform-on-load (see Conv3Args): used by the teacher to fold the ExpertBlock
// tail into the next block's conv1.  nlev = 0 is the plain kernel (any batch size: no grid-length heuristic here).
bool lo_conv3_pp_applies(const LoGeom& g) {
  return g.n_phase == 1 && g.T[0] == 9 && g.in_stride == 1 && g.out_stride == 1 && g.Cin % 64 == 0 && g.Cout % 128 == 0 &&
         g.Hin % 16 == 0 && g.Win % 16 == 0 && g.Hin >= 16 && g.Cin <= 128;
}
int lo_conv3_run_pp_xf(const LoGeom& g, const f16* in, const f16* xc, const f16* kx, int nlev, const f16* wp, const float* bias,
                       f16* out, hipStream_t st, const LoConvExtra* ex) {
  LO_REQUIRE(lo_conv3_pp_applies(g), "lo_conv3_run_pp_xf: geometry not supported");
  LO_REQUIRE(nlev >= 0 && nlev <= 2 && (nlev == 0 || (xc && kx)), "lo_conv3_run_pp_xf: bad transform arguments");
  Conv3Args a{in, wp, bias, nullptr, out, nullptr, ex ? ex->bn_partial : nullptr, ex ? ex->act : 0, xc, kx, nlev, nullptr, g};
#ifdef LO_STAMPS
  a.stamps = nullptr;
#endif
  const int tiles = g.B * (g.Hin / 16) * (g.Win / 16) * (g.Cout / 128);
  double flops = 2.0 * g.B * g.Hin * g.Win * (double)g.Cout * 9 * g.Cin;
  double bytes = 2.0 * ((double)g.B * g.Hin * g.Win * (g.Cin + g.Cout) + 9.0 * g.Cin * g.Cout);
  LoProfScope _p(nlev ? "lo_conv3x3_pp (transform on load)" : "lo_conv3x3_pp", flops, bytes, st);
  if (nlev) hipLaunchKernelGGL((lo_conv3x3_pp<128, 16, LO_PP_XF>), dim3(tiles), dim3(512), 0, st, a);
  else hipLaunchKernelGGL((lo_conv3x3_pp<128, 16, LO_PP_PLAIN>), dim3(tiles), dim3(512), 0, st, a);
  LO_LAUNCH_CHECK("conv3x3_pp_xf");
  return LO_OK;
}

// e4m3 operands (in8 = e4m3(x * LO_F8_ACT_SCALE) NHWC bytes, w8 = packed weights as e4m3 with one scale per output channel,
// wscale = that scale / LO_F8_ACT_SCALE: lo_pack_f8_one) on the 16x16-pixel fused-tap kernel; fp16 output, same epilogues.
bool lo_conv3_pp_f8_applies(const LoGeom& g) {
  return lo_conv3_pp_applies(g) && g.Cin % 128 == 0;
}
int lo_conv3_run_pp_f8(const LoGeom& g, const uint8_t* in8, const uint8_t* w8, const float* wscale, const float* bias, f16* out,
                       hipStream_t st, const LoConvExtra* ex) {
  LO_REQUIRE(lo_conv3_pp_f8_applies(g), "lo_conv3_run_pp_f8: geometry not supported");
  LO_REQUIRE(in8 && w8 && wscale && out, "lo_conv3_run_pp_f8: null argument");
  Conv3Args a{reinterpret_cast<const f16*>(in8), reinterpret_cast<const f16*>(w8), bias, nullptr, out, nullptr, ex ? ex->bn_partial : nullptr,
              ex ? ex->act : 0, nullptr, nullptr, 0, wscale, g};
#ifdef LO_STAMPS
  a.stamps = nullptr;
#endif
  const int tiles = g.B * (g.Hin / 16) * (g.Win / 16) * (g.Cout / 128);
  double flops = 2.0 * g.B * g.Hin * g.Win * (double)g.Cout * 9 * g.Cin;
  double bytes = (double)g.B * g.Hin * g.Win * (g.Cin + 2.0 * g.Cout) + 9.0 * g.Cin * g.Cout;
  LoProfScope _p("lo_conv3x3_pp<f8>", flops, bytes, st);
  hipLaunchKernelGGL((lo_conv3x3_pp<128, 16, LO_PP_F8>), dim3(tiles), dim3(512), 0, st, a);
  LO_LAUNCH_CHECK("conv3x3_pp_f8");
  return LO_OK;
}

// =============================================================================================
// Patch-resident forward of the last transposed convolution (ConvTranspose2d k4 s2 p1, 64 -> 32 channels, 64x64 -> 128x128;
// lunar_generate.py:187).  As four sub-pixel phases on lo_igemm_nt every input pixel crosses the L2 -> LDS path 16 times
// (4 phases x 4 taps, 537 MB per launch at batch 64) for 17 GFLOP: 66 us against a 12.6 us HBM floor.  Here a workgroup
// (8 waves) stages the 18 x 18 halo patch of a 16 x 16 input tile ONCE (41 KB) and runs all 16 (phase, tap) products from
// it, two passes (output row parity py = 0, 1) with that parity's two phases of packed weights in LDS (32 KB, lo_igemm_nt's
// own packed layout): 75 KB of LDS, two workgroups per CU.
//   product: D[co][px] = sum_k W[co][k] X[k][px]  (weights are the A operand), so a lane ends up with 4 consecutive output
//   channels of one output pixel: 8-byte stores, and -- GroupNorm(8, 32) has 4 channels per group -- exactly one group's
//   members, so the per-(sample, tile) GroupNorm sums come straight from the accumulators (fp16-rounded values, like the other
//   conv epilogues), reduced over the 16 pixel lanes by shuffles and over waves through LDS in a fixed order.
//   LDS: patch rows of 128 B (64 channels), 16-byte chunks XOR-swizzled by lo_swz3(pixel); weight rows of 512 B (4 taps x 64
//   channels), chunk c of row r stored at c ^ (r & 15): the 16 rows of an A fragment land on 16 distinct 16-byte slots.
// =============================================================================================
struct ConvT4PatchArgs {
  const f16* in;        // [B][H][W][64]
  const f16* w;         // packed by lo_pack_all for the LO_CONVT4_S2 geometry: [phase][32][4 * 64]
  const float* bias;    // [32] or null
  f16* out;             // [B][2H][2W][32]
  float* gn_partial;    // [B][tiles per image][8][2] or null
  LoGeom g;
  const f16* add_src;   // DG form: residual gradient added to the output (fp32, before the rounding), or null
};

// NCB = Cin / 64 channel blocks (1 or 2: all blocks of the patch stay resident), COUT = 32 or 64.  One weight buffer holds the two
// phases of a (parity, channel block): 2 * COUT rows of 512 B (4 taps x 64 channels); the accumulators run over the channel blocks.
// DG: the same kernel as the DATA GRADIENT of a stride-2 3x3 convolution (LO_CONV3_S2_DGRAD: four sub-pixel phases of 1 / 2 / 2 / 4
// taps reading the same 17 x 17 patch of dy, 9 tap products instead of 16): tap slots past T[phase] of the 512-byte weight rows are
// filled from the zero page and skipped in the K loop, the residual (skip) gradient is added in the coalesced store loop.  On lo_igemm_nt
// the four phases are separate tiles with K = 128 ... 512: prologue- and epilogue-bound (46 us for 9.7 GFLOP at batch 64).
template <int NCB, int COUT, bool DG = false>
__global__ __launch_bounds__(512) void lo_convt4_patch_fwd_kernel(ConvT4PatchArgs a) {
  constexpr int TH = 16, TW = 16, PW = TW + 2, NPIX = (TH + 2) * PW;      // 324 patch pixels
  constexpr int CIN = 64 * NCB, KT = 4 * CIN;                             // K per phase
  constexpr int NF = COUT / 16;                                           // 16-channel fragments of the output channels
  constexpr int PQ = (NPIX + 7) / 8;                                      // LDS-DMA instructions of one patch block (8 pixels each)
  constexpr int PATCH_BYTES = PQ * 1024;
  constexpr int W_BYTES = 2 * COUT * 512;                                 // two phases of one channel block
  constexpr int OCH = COUT / 8;                                           // 16-byte chunks per output pixel
  static_assert(W_BYTES == 16 * 32 * COUT * 2, "the output block of a parity is staged in the weight buffer");
  static_assert(NCB * PATCH_BYTES + W_BYTES + 512 <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NCB * PATCH_BYTES + W_BYTES];
  __shared__ float s_red[8][8][2];
  unsigned char* const s_patch = smem;
  unsigned char* const s_w = smem + NCB * PATCH_BYTES;
  const LoGeom& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = g.Hin, W = g.Win;
  const int tiles_x = W / TW, tiles_img = tiles_x * (H / TH);
  const int tile_id = lo_xcd_remap3(blockIdx.x, gridDim.x);
  const int n_img = tile_id / tiles_img, t_img = tile_id - n_img * tiles_img;
  const int y0 = (t_img / tiles_x) * TH, x0 = (t_img % tiles_x) * TW;
  const unsigned char* zpage = reinterpret_cast<const unsigned char*>(lo_zero_page3);
  const unsigned char* inb = reinterpret_cast<const unsigned char*>(a.in);
  const unsigned char* wb = reinterpret_cast<const unsigned char*>(a.w);

  // ---- patch: instruction q of block cb covers pixels 8q .. 8q+7 (lane = pixel * 8 + physical chunk)
  for (int q = wave; q < NCB * PQ; q += 8) {
    const int cb = q / PQ, ql = q - cb * PQ;
    const int pp = ql * 8 + (lane >> 3), pos = lane & 7;
    const int py = pp / PW, px = pp - py * PW;
    const int iy = y0 - 1 + py, ix = x0 - 1 + px;
    const bool ok = pp < NPIX && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    const unsigned char* src = ok ? inb + ((size_t)((n_img * H + iy) * W + ix) * CIN + cb * 64 + ((pos ^ lo_swz3(pp)) * 8)) * 2 : zpage;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(s_patch + q * 1024), 16, 0, 0);
  }
  // weights of output-row parity `par`, channel block cb: rows r = pw * COUT + co, 512 B each (tap t, 64 channels of the block:
  // 128-byte pieces of the packed row); instruction q covers rows 2q, 2q+1
  auto issue_w = [&](int par, int cb) __attribute__((always_inline)) {
    for (int q = wave; q < COUT; q += 8) {
      const int r = 2 * q + (lane >> 5), pos = lane & 31;
      const int pw = r / COUT, co = r - pw * COUT;
      const int p = par * 2 + pw;
      const int c = pos ^ (r & 15), t = c >> 3, c8 = c & 7;
      const int ktp = DG ? g.T[p] * CIN : KT;
      const unsigned char* src = (DG && t >= g.T[p]) ? zpage : wb + ((size_t)g.wofs[p] + (size_t)co * ktp + t * CIN + cb * 64 + c8 * 8) * 2;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(s_w + q * 1024), 16, 0, 0);
    }
  };
  issue_w(0, 0);

  const int fr = lane & 15, fq = lane >> 4;
  float gs[NF], gq[NF];                     // GroupNorm sums of this lane's 4 channels per fragment over its pixels
  f32x4 bv[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    gs[nf] = 0.f; gq[nf] = 0.f;
    bv[nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (a.bias) bv[nf] = *reinterpret_cast<const f32x4*>(a.bias + nf * 16 + 4 * fq);
  }

#pragma unroll 1
  for (int par = 0; par < 2; ++par) {
    f32x4 acc[2][2][NF];                     // [pw][row of the wave][nf]
#pragma unroll
    for (int pw = 0; pw < 2; ++pw)
#pragma unroll
      for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) acc[pw][rr][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int cb = 0; cb < NCB; ++cb) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                       // the patch (first pass) and this (parity, block)'s weights have landed for every wave
      const unsigned char* pb = s_patch + cb * PATCH_BYTES;
#pragma unroll
      for (int pw = 0; pw < 2; ++pw) {
        const int p = par * 2 + pw;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (DG && t >= g.T[p]) continue;
          const int dy = g.dy[p][t], dx = g.dx[p][t];
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            f16x8 wf[NF], xf[2];
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
              const int r = pw * COUT + nf * 16 + fr;
              const int chunk = t * 8 + kk * 4 + fq;
              wf[nf] = *reinterpret_cast<const f16x8*>(s_w + r * 512 + ((chunk ^ (r & 15)) * 16));
            }
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
              const int pp = (2 * wave + rr + 1 + dy) * PW + fr + 1 + dx;
              xf[rr] = *reinterpret_cast<const f16x8*>(pb + pp * 128 + (((kk * 4 + fq) ^ lo_swz3(pp)) * 16));
            }
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
              for (int nf = 0; nf < NF; ++nf)
                acc[pw][rr][nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nf], xf[rr], acc[pw][rr][nf], 0, 0, 0);
          }
        }
      }
      __syncthreads();                       // every wave is done with this block's weights
      if (cb + 1 < NCB) issue_w(par, cb + 1);
    }
    // ---- epilogue of this parity: D[co = nf*16 + 4 fq + j][px = fr].  The 16 x 32-pixel x COUT output block of the parity is
    //      staged in the weight buffer -- [row][ox][OCH chunks of 8 channels], chunk ^ ((ox >> 1) & (OCH - 1)) -- and leaves in
    //      16-byte stores along the contiguous output rows (direct 8-byte stores from the accumulators: 41 instead of 36 us)
#pragma unroll
    for (int pw = 0; pw < 2; ++pw)
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const int row = 2 * wave + rr, ox = 2 * fr + pw;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
          const f32x4 v = acc[pw][rr][nf] + bv[nf];
          const f16x4 h = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
          const int chunk = nf * 2 + (fq >> 1);
          *reinterpret_cast<f16x4*>(s_w + (row * 32 + ox) * (COUT * 2) + ((chunk ^ ((ox >> 1) & (OCH - 1))) * 16) + (fq & 1) * 8) = h;
#pragma unroll
          for (int j = 0; j < 4; ++j) { const float x = (float)h[j]; gs[nf] += x; gq[nf] += x * x; }
        }
      }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < OCH; ++i) {
      const int c = tid + 512 * i, chunk = c % OCH, k = c / OCH, ox = k & 31, row = k >> 5;
      u32x4 v = *reinterpret_cast<const u32x4*>(s_w + (row * 32 + ox) * (COUT * 2) + ((chunk ^ ((ox >> 1) & (OCH - 1))) * 16));
      const int oy = 2 * (y0 + row) + par;
      const size_t oo = ((size_t)(n_img * 2 * H + oy) * (2 * W) + 2 * x0 + ox) * COUT + chunk * 8;
      if (DG && a.add_src) {
        // the residual (skip) gradient joins here, on the coalesced side: fp16(fp16(acc) + r), the same two roundings as lo_igemm_nt's epilogue
        const f16x8 r = *reinterpret_cast<const f16x8*>(a.add_src + oo);
        f16x8 hv = *reinterpret_cast<f16x8*>(&v);
#pragma unroll
        for (int j = 0; j < 8; ++j) hv[j] = (f16)((float)hv[j] + (float)r[j]);
        v = *reinterpret_cast<u32x4*>(&hv);
      }
      *reinterpret_cast<u32x4*>(a.out + oo) = v;
    }
    if (par == 0) {
      __syncthreads();                       // the staged block has been read
      issue_w(1, 0);
    }
  }
  if (a.gn_partial) {
    // over the 16 pixel lanes (fixed shuffle tree; COUT = 64: a group is 8 channels = the lane pairs fq, fq ^ 1), then over the
    // 8 waves (fixed order).  Group of (nf, fq): COUT = 32 -> nf * 4 + fq, COUT = 64 -> nf * 2 + (fq >> 1)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { gs[nf] += __shfl_xor(gs[nf], o, 64); gq[nf] += __shfl_xor(gq[nf], o, 64); }
      if (COUT == 64) { gs[nf] += __shfl_xor(gs[nf], 16, 64); gq[nf] += __shfl_xor(gq[nf], 16, 64); }
    }
    if (fr == 0 && (COUT == 32 || (fq & 1) == 0)) {
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        const int grp = COUT == 32 ? nf * 4 + fq : nf * 2 + (fq >> 1);
        s_red[wave][grp][0] = gs[nf]; s_red[wave][grp][1] = gq[nf];
      }
    }
    __syncthreads();
    if (tid < 16) {
      const int grp = tid >> 1, which = tid & 1;
      float tot = 0.f;
      for (int w8 = 0; w8 < 8; ++w8) tot += s_red[w8][grp][which];
      a.gn_partial[(((size_t)n_img * tiles_img + t_img) * 8 + grp) * 2 + which] = tot;
    }
  }
}

// ConvTranspose2d k4 s2 p1 forward geometry with 64 -> 32 or 128 -> 64 channels on a map of whole 16 x 16 tiles
bool lo_convt4_patch_applies(const LoGeom& g) {
  if (g.n_phase != 4 || g.in_stride != 1 || g.out_stride != 2) return false;
  if (!((g.Cin == 64 && g.Cout == 32) || (g.Cin == 128 && g.Cout == 64))) return false;
  for (int p = 0; p < 4; ++p) if (g.T[p] != 4) return false;
  return g.Hin % 16 == 0 && g.Win % 16 == 0 && g.Hout == 2 * g.Hin && g.Wout == 2 * g.Win;
}
int lo_convt4_patch_tiles_per_image(const LoGeom& g) { return lo_convt4_patch_applies(g) ? (g.Hin / 16) * (g.Win / 16) : 0; }

// Data gradient of the stride-2 3x3 convolution 64 -> 128 (lunar_generate.py:102): dy [B][32][32][128] -> dx [B][64][64][64] (+ add_src)
bool lo_convs2d_patch_applies(const LoGeom& g) {
  if (g.n_phase != 4 || g.in_stride != 1 || g.out_stride != 2 || g.Cin != 128 || g.Cout != 64) return false;
  if (g.T[0] != 1 || g.T[1] != 2 || g.T[2] != 2 || g.T[3] != 4) return false;
  for (int p = 0; p < 4; ++p)
    for (int t = 0; t < g.T[p]; ++t) if (g.dy[p][t] < 0 || g.dy[p][t] > 1 || g.dx[p][t] < 0 || g.dx[p][t] > 1) return false;
  return g.Hin % 16 == 0 && g.Win % 16 == 0 && g.Hout == 2 * g.Hin && g.Wout == 2 * g.Win;
}
int lo_convs2d_patch_run(const LoGeom& g, const f16* dy, const f16* wp, const f16* add_src, f16* dx, hipStream_t st) {
  LO_REQUIRE(lo_convs2d_patch_applies(g), "lo_convs2d_patch_run: geometry not supported");
  ConvT4PatchArgs a{dy, wp, nullptr, dx, nullptr, g, add_src};
  const int tiles = g.B * (g.Hin / 16) * (g.Win / 16);
  double flops = 0;
  for (int p = 0; p < 4; ++p) flops += 2.0 * g.B * g.GH * g.GW * (double)g.Cout * g.T[p] * g.Cin;
  const double bytes = 2.0 * ((double)g.B * g.Hin * g.Win * g.Cin + (add_src ? 2.0 : 1.0) * (double)g.B * g.Hout * g.Wout * g.Cout + 9.0 * g.Cin * g.Cout);
  LoProfScope _p(lo_prof_geom_name("lo_convs2d_patch", g), flops, bytes, st);
  LO_LAUNCH_STOP((lo_convt4_patch_fwd_kernel<2, 64, true>), dim3(tiles), dim3(512), 0, st, a);
  LO_LAUNCH_CHECK("convs2d_patch");
  return LO_OK;
}

int lo_convt4_patch_run(const LoGeom& g, const f16* in, const f16* wp, const float* bias, f16* out, float* gn_partial, hipStream_t st) {
  LO_REQUIRE(lo_convt4_patch_applies(g), "lo_convt4_patch_run: geometry not supported");
  ConvT4PatchArgs a{in, wp, bias, out, gn_partial, g, nullptr};
  const int tiles = g.B * (g.Hin / 16) * (g.Win / 16);
  double flops = 0;
  for (int p = 0; p < 4; ++p) flops += 2.0 * g.B * g.GH * g.GW * (double)g.Cout * g.T[p] * g.Cin;
  const double bytes = 2.0 * ((double)g.B * g.Hin * g.Win * g.Cin + (double)g.B * g.Hout * g.Wout * g.Cout + 16.0 * g.Cin * g.Cout);
  LoProfScope _p("lo_convt4_patch_fwd", flops, bytes, st);
  if (g.Cin == 64) hipLaunchKernelGGL((lo_convt4_patch_fwd_kernel<1, 32>), dim3(tiles), dim3(512), 0, st, a);
  else hipLaunchKernelGGL((lo_convt4_patch_fwd_kernel<2, 64>), dim3(tiles), dim3(512), 0, st, a);
  LO_LAUNCH_CHECK("convt4_patch_fwd");
  return LO_OK;
}
