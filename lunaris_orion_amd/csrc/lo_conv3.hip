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
  LoGeom g;
};

template <int BN, int TH, int TW, int NSB>
__global__ __launch_bounds__(256) void lo_conv3x3_halo(Conv3Args a) {
  constexpr int BM = TH * TW;                 // output pixels per workgroup (128 or 64)
  constexpr int PW = TW + 2, PH = TH + 2, NPIX = PH * PW;
  constexpr int PI = (NPIX + 31) / 32;        // patch LDS-DMA instructions per wave (8 rows each, 4 waves)
  static_assert(PI <= 8, "patch pieces must fit in taps 0..7");
  constexpr int PATCH_BYTES = PI * 4 * 1024;
  constexpr int IB = BN / 8 / 4;              // weight-tile LDS-DMA instructions per wave and step
  constexpr int B_BYTES = BN * 128;
  constexpr int D = NSB - 1;                  // steps in flight
  constexpr int LPT = IB + 1;                 // DMA instructions per wave and step (weights + one patch piece / dummy)
  static_assert(LPT * D <= 63, "vmcnt range");
  constexpr int WM = BM / 2, WN = BN / 2, MI = WM / 16, NI = WN / 16;
  constexpr int OPITCH = BN * 2 + 16;
  constexpr int MAIN_BYTES = 2 * PATCH_BYTES + NSB * B_BYTES + 1024;   // + 1 KiB dummy slot
  constexpr int EPI_BYTES = BM * OPITCH + 6400;
  constexpr int LDS_BYTES = MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES];
  unsigned char* const s_patch = smem;
  unsigned char* const s_b = smem + 2 * PATCH_BYTES;
  unsigned char* const s_dummy = s_b + NSB * B_BYTES;

  const LoGeom& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int H = g.Hin, W = g.Win, Cin = g.Cin;
  const int tiles_x = W / TW, tiles_y = H / TH, tiles_img = tiles_x * tiles_y;
  const int NT = g.Cout / BN;
  const int tile_id = lo_xcd_remap3(blockIdx.x, gridDim.x);
  const int nt_i = tile_id % NT, pt_i = tile_id / NT;          // channel tile fastest: both n tiles share the patch in L2
  const int n_img = pt_i / tiles_img, t_img = pt_i - n_img * tiles_img;
  const int y0 = (t_img / tiles_x) * TH, x0 = (t_img % tiles_x) * TW;
  const int n0 = nt_i * BN;
  const int KCB = Cin / 64;
  const int nsteps = 9 * KCB;
  const int Ktot = 9 * Cin;
  const f16* zpage = reinterpret_cast<const f16*>(lo_zero_page3);
  const uint32_t dyc = g.dyc[0], dxc = g.dxc[0];

  // ---- per-lane DMA sources
  int p_src[PI];   // element offset of this lane's patch chunk at channel block 0, or -1
#pragma unroll
  for (int i = 0; i < PI; ++i) {
    int pp = (wave * PI + i) * 8 + (lane >> 3), pos = lane & 7;
    int py = pp / PW, px = pp - py * PW;
    int iy = y0 - 1 + py, ix = x0 - 1 + px;
    bool ok = pp < NPIX && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    p_src[i] = ok ? ((n_img * H + iy) * W + ix) * Cin + ((pos ^ ((pp >> 1) & 7)) * 8) : -1;
  }
  int b_src[IB];
#pragma unroll
  for (int i = 0; i < IB; ++i) {
    int row = (wave * IB + i) * 8 + (lane >> 3), pos = lane & 7;
    b_src[i] = (n0 + row) * Ktot + ((pos ^ ((row >> 1) & 7)) * 8);
  }
  auto issue_patch_piece = [&](int buf, int cb, int piece) __attribute__((always_inline)) {
    // piece is wave-uniform; pieces >= PI (and channel blocks past the end) go to the dummy slot
    int off = -1;
#pragma unroll
    for (int i = 0; i < PI; ++i) if (i == piece) off = p_src[i];
    const bool real = piece < PI && cb < KCB;
    const f16* src = (real && off >= 0) ? a.in + (off + cb * 64) : zpage;
    unsigned char* dst = real ? s_patch + buf * PATCH_BYTES + (wave * PI + piece) * 1024 : s_dummy;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };
  auto issue_b = [&](int stage, int step) __attribute__((always_inline)) {
    const bool live = step < nsteps;
    const int cb = step / 9, tap = step - cb * 9;
    const int koff = tap * Cin + cb * 64;
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      const f16* src = live ? a.w + (b_src[i] + koff) : zpage;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(s_b + stage * B_BYTES + (wave * IB + i) * 1024), 16, 0, 0);
    }
  };

  // ---- fragment read coordinates
  const int fr = lane & 15, fq = lane >> 4;
  int pp0[MI];     // patch pixel index of this lane's output pixel (tap offset 0) per fragment
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    int p = mi * 16 + fr;                       // pixel inside the wave's WM-pixel slab
    int ty = wm * (TH / 2) + p / TW, tx = p % TW;
    pp0[mi] = (ty + 1) * PW + tx + 1;
  }
  int woff[NI][2];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      int R = wn * WN + ni * 16 + fr;
      woff[ni][kk] = R * 128 + (((kk * 4 + fq) ^ ((R >> 1) & 7)) * 16);
    }

  f32x4 acc[NI][MI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- prologue: whole patch of channel block 0, then D step groups (weights + dummy)
#pragma unroll
  for (int i = 0; i < PI; ++i) issue_patch_piece(0, 0, i);
#pragma unroll
  for (int s = 0; s < D; ++s) {
    issue_b(s, s);
    issue_patch_piece(0, KCB, PI);   // dummy keeps the group size uniform
  }
  int rs = 0, ws = D % NSB;
  int cb = 0, tap = 0;
  for (int step = 0; step < nsteps; ++step) {
    LO_VMCNT(LPT * (D - 1));
    __builtin_amdgcn_s_barrier();
    issue_b(ws, step + D);
    issue_patch_piece((cb + 1) & 1, cb + 1, tap);   // pieces 0..PI-1 of the next block during taps 0..PI-1, dummies after
    const unsigned char* pbase = s_patch + (cb & 1) * PATCH_BYTES;
    const unsigned char* bbase = s_b + rs * B_BYTES;
    const int dy = (int)((dyc >> (2 * tap)) & 3u) - 1, dx = (int)((dxc >> (2 * tap)) & 3u) - 1;
    const int delta = dy * PW + dx;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      f16x8 wf[NI], xf[MI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) wf[ni] = *reinterpret_cast<const f16x8*>(bbase + woff[ni][kk]);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        int pp = pp0[mi] + delta;
        xf[mi] = *reinterpret_cast<const f16x8*>(pbase + pp * 128 + (((kk * 4 + fq) ^ ((pp >> 1) & 7)) * 16));
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
    }
    rs = (rs + 1 == NSB) ? 0 : rs + 1;
    ws = (ws + 1 == NSB) ? 0 : ws + 1;
    if (++tap == 9) { tap = 0; ++cb; }
  }
  LO_VMCNT(0);
  __syncthreads();

  // ---- epilogue (same structure as lo_igemm_nt): bias -> fp16 tile in LDS -> coalesced stores (+ add, GN partials)
  unsigned char* so = smem;
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
  const int G = g.Cout >> 3;
  float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
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
    *reinterpret_cast<f16x8*>(a.out + off) = h;
    if (a.gn_partial) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { float x = (float)h[j]; s0 += x; q0 += x * x; }
#pragma unroll
      for (int j = 4; j < 8; ++j) { float x = (float)h[j]; s1 += x; q1 += x * x; }
    }
  }
  if (a.gn_partial) {
    float* red = reinterpret_cast<float*>(smem + BM * OPITCH);
    red[tid * 4 + 0] = s0; red[tid * 4 + 1] = q0; red[tid * 4 + 2] = s1; red[tid * 4 + 3] = q1;
    constexpr int NV = OCPR * 4, P = 256 / NV, RPP2 = ORPP / P;
    float* red2 = red + 1024;
    float* red3 = red2 + 256;
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
      int grp = (n0 / G) + gl;
      a.gn_partial[(((size_t)n_img * tiles_img + t_img) * 8 + grp) * 2 + which] = tot;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------------------------
static inline bool conv3_tile(const LoGeom& g, int* th, int* tw, int* bn) {
  if (g.n_phase != 1 || g.T[0] != 9 || g.in_stride != 1 || g.out_stride != 1) return false;
  if (g.Cin % 64 || g.Cout % 64) return false;
  if (g.Hin % 8) return false;
  if (g.Win % 16 == 0) { *th = 8; *tw = 16; }
  else if (g.Win % 8 == 0) { *th = 8; *tw = 8; }
  else return false;
  static const int want_bn = getenv("LO_HALO_BN") ? atoi(getenv("LO_HALO_BN")) : 64;   // 64: two workgroups per CU
  *bn = (want_bn == 128 && g.Cout % 128 == 0) ? 128 : 64;
  if ((g.Cout >> 3) > *bn) return false;   // a GroupNorm group must fit inside the N tile
  return true;
}

// tiles per image of the halo kernel for this geometry, or 0 when the kernel does not apply / is disabled
int lo_conv3_tiles_per_image(const LoGeom& g) {
  // measured on MI355X (round 1): correct, but not faster than lo_igemm_nt yet (32 % LDS bank-conflict cycles on the
  // shifted fragment reads, more scalar work per step) -> opt-in until that is fixed
  static const bool off = getenv("LO_HALO") == nullptr;
  int th, tw, bn;
  if (off || !conv3_tile(g, &th, &tw, &bn)) return 0;
  return (g.Hin / th) * (g.Win / tw);
}

int lo_conv3_run(const LoGeom& g, const f16* in, const f16* wp, const float* bias, const f16* add_src, f16* out,
                 float* gn_partial, hipStream_t st) {
  int th, tw, bn;
  LO_REQUIRE(conv3_tile(g, &th, &tw, &bn), "lo_conv3_run: geometry not supported by the fused-tap kernel");
  Conv3Args a{in, wp, bias, add_src, out, gn_partial, g};
  const int tiles = g.B * (g.Hin / th) * (g.Win / tw) * (g.Cout / bn);
  double flops = 2.0 * g.B * g.Hin * g.Win * (double)g.Cout * 9 * g.Cin;
  double bytes = 2.0 * ((double)g.B * g.Hin * g.Win * (g.Cin + g.Cout) + 9.0 * g.Cin * g.Cout);
  static char name[64];
  snprintf(name, sizeof(name), "lo_conv3x3_halo<%d,%dx%d>", bn, th, tw);
  LoProfScope _p(name, flops, bytes, st);
  if (tw == 16 && bn == 64) hipLaunchKernelGGL((lo_conv3x3_halo<64, 8, 16, 3>), dim3(tiles), dim3(256), 0, st, a);
  else if (tw == 16 && bn == 128) hipLaunchKernelGGL((lo_conv3x3_halo<128, 8, 16, 3>), dim3(tiles), dim3(256), 0, st, a);
  else if (tw == 8 && bn == 64) hipLaunchKernelGGL((lo_conv3x3_halo<64, 8, 8, 3>), dim3(tiles), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((lo_conv3x3_halo<128, 8, 8, 3>), dim3(tiles), dim3(256), 0, st, a);
  LO_LAUNCH_CHECK("conv3x3_halo");
  return LO_OK;
}
