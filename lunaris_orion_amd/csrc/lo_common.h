// Common definitions for the lunaris-orion MI355X (gfx950 / CDNA4) kernels.
// Internal header: the public C ABI is include/lunaris_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 h16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));  // type of the tr16 builtin
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

#define LO_WAVE 64

// error classes returned across the C ABI
#define LO_OK 0
#define LO_ERR_ARG (-1)      // bad argument / unsupported shape
#define LO_ERR_HIP (-2)      // a HIP runtime call failed
#define LO_ERR_STATE (-3)    // object used in the wrong state

void lo_set_error(const char* fmt, ...);
int lo_check_hip(hipError_t e, const char* what);

#define LO_HIP(call)                                   \
  do {                                                 \
    int _e = lo_check_hip((call), #call);              \
    if (_e != LO_OK) return _e;                        \
  } while (0)
#define LO_LAUNCH_CHECK(name) LO_HIP(hipGetLastError())
#define LO_CHECK(...)                                  \
  do {                                                 \
    int _e = (__VA_ARGS__);                                   \
    if (_e != LO_OK) return _e;                        \
  } while (0)
#define LO_REQUIRE(cond, ...)                          \
  do {                                                 \
    if (!(cond)) {                                     \
      lo_set_error(__VA_ARGS__);                       \
      return LO_ERR_ARG;                               \
    }                                                  \
  } while (0)

// ---------------------------------------------------------------------------------------------
// Optional per-launch timing with HIP events ON THE LAUNCH STREAM (bench.py's roofline leg).  Off by default:
// one predictable branch per launcher.
// ---------------------------------------------------------------------------------------------
extern bool g_lo_prof_on;
extern bool g_lo_prof_layers;
extern const char* g_lo_prof_tag;   // when set, replaces the name of the records opened while it is set (per-call-site breakdowns)
void lo_prof_begin(const char* name, double flops, double bytes, hipStream_t st);
void lo_prof_end(hipStream_t st);
struct LoProfScope {
  hipStream_t st;
  bool on;
  LoProfScope(const char* name, double flops, double bytes, hipStream_t s) : st(s), on(g_lo_prof_on) {
    if (on) lo_prof_begin(g_lo_prof_tag ? g_lo_prof_tag : name, flops, bytes, s);
  }
  ~LoProfScope() {
    if (on) lo_prof_end(st);
  }
};

// Cross-stream hand-over without a marker packet.  hipEventRecord after a kernel puts a barrier packet with a signal behind it, and
// the NEXT kernel of that stream waits for the packet: measured (tools/probe/ev_probe.hip, MI355X) +3.5 .. 4.7 us on the producing
// stream per hand-over.  hipExtLaunchKernelGGL(..., stopEvent) binds the event to the kernel's own completion signal instead:
// +0.9 .. 1.3 us.  The host code that wants the launch which writes a buffer to carry an event sets g_lo_stop_event right before
// calling the launcher; the launch sites that can be that last launch use LO_LAUNCH_STOP, which consumes the variable (thread-local:
// engines may be driven from several host threads).  A caller checks that it was consumed and falls back to hipEventRecord if not.
extern thread_local hipEvent_t g_lo_stop_event;
#define LO_LAUNCH_STOP(kern, grid, block, lds, st, ...)                                                  \
  do {                                                                                                   \
    hipEvent_t se__ = g_lo_stop_event;                                                                   \
    g_lo_stop_event = nullptr;                                                                           \
    if (se__) hipExtLaunchKernelGGL(kern, grid, block, lds, st, nullptr, se__, 0, __VA_ARGS__);          \
    else hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__);                                    \
  } while (0)

// LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B -> 1 KiB at LDS offset lds_off, lane-linear) as inline asm.
// Kernels whose fragments come from ds_read_b64_tr_b16 use this form instead of __builtin_amdgcn_global_load_lds: after
// the builtin the compiler cannot tell the transposed reads from the DMA destinations and puts `s_waitcnt vmcnt(0)` in
// front of them, which serialises every K step on the loads just issued (measured: lo_wgrad_tn, round 1).  With the asm
// form the ordering is the kernel's own counted `s_waitcnt vmcnt(N)` + barrier, as written.  lds_off must be wave-uniform.
__device__ __forceinline__ void lo_dma16(const void* gptr, unsigned int lds_off) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(lds_off) : "memory");
}

// ---------------------------------------------------------------------------------------------
// Geometry of one implicit-GEMM convolution-like op (forward conv, transposed conv as sub-pixel
// phases, and every data-gradient of those).  All tensors are NHWC fp16.
//
//   for phase p, output-grid point (n, gy, gx), output channel co:
//     out[n, gy*out_stride + out_oy[p], gx*out_stride + out_ox[p], co] =
//        bias[co] + sum_{t < T[p]} sum_{ci} in[n, gy*in_stride + dy[p][t], gx*in_stride + dx[p][t], ci]
//                                         * Wp[p][co][t*Cin + ci]
//   (input positions outside [0,Hin)x[0,Win) read as zero).
// ---------------------------------------------------------------------------------------------
#define LO_MAX_TAPS 16
#define LO_MAX_PHASE 4
struct LoGeom {
  int B, Hin, Win, Cin;
  int Hout, Wout, Cout;
  int GH, GW;          // output grid per phase (powers of two)
  int lgh, lgw;        // log2(GH), log2(GW)
  int lg_hin, lg_win, lg_cin, lg_hout, lg_wout, lg_cout;   // log2 of the tensor dims, or -1 when not a power of two
  int in_stride, out_stride;
  int n_phase;
  int T[LO_MAX_PHASE];
  int out_oy[LO_MAX_PHASE], out_ox[LO_MAX_PHASE];
  int wofs[LO_MAX_PHASE];                 // element offset of phase p inside the packed weight
  int8_t dy[LO_MAX_PHASE][LO_MAX_TAPS];
  int8_t dx[LO_MAX_PHASE][LO_MAX_TAPS];
  int8_t rs[LO_MAX_PHASE][LO_MAX_TAPS];   // r*S+s of the canonical weight this tap reads
  uint32_t dyc[LO_MAX_PHASE], dxc[LO_MAX_PHASE];  // the same offsets, 2 bits per tap: (d + 1) << (2*t)
  // canonical weight addressing: W[n*sn + c*sc + rs]  (n = this op's output channel, c = reduced channel)
  int sn, sc;
};

struct LoPackJob {
  const float* src;   // canonical fp32 weight
  f16* dst;           // packed fp16 operand
  int total;          // packed elements
  int block0;         // first block of this job in the fused launch
  LoGeom g;
};

// GroupNorm(8) statistics of sample n from the conv epilogue's partial sums partial[n][MT][8][2]: s_stat[g*2] = mean, [g*2+1] = rstd
// (lo_gn_fwd, and the fused-tap kernel when it applies the normalisation on load)
__device__ __forceinline__ void lo_gn_group_stats(const float* partial, int MT, int n, float inv_m, float* s_stat, int tid) {
  // 64 threads: group = tid>>3, part = tid&7 ; double accumulation, fixed order
  if (tid < 64) {
    int grp = tid >> 3, part = tid & 7;
    double s = 0.0, q = 0.0;
    // rows part, part + 8, ...: eight rows' loads are issued together and added in row order (the rolled loop chained MT / 8
    // round trips in front of every GroupNorm pass of the forward: every workgroup of the pass waits for this prologue)
    for (int mt0 = part; mt0 < MT; mt0 += 64) {
      f32x2 pv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int mt = mt0 + 8 * i;
        pv[i] = mt < MT ? *reinterpret_cast<const f32x2*>(partial + (((size_t)n * MT + mt) * 8 + grp) * 2) : (f32x2){0.f, 0.f};
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (mt0 + 8 * i < MT) { s += (double)pv[i][0]; q += (double)pv[i][1]; }
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
      s += __shfl_xor(s, o, 64);
      q += __shfl_xor(q, o, 64);
    }
    if (part == 0) {
      double mean = s * inv_m;
      double var = q * inv_m - mean * mean;
      if (var < 0.0) var = 0.0;
      s_stat[grp * 2 + 0] = (float)mean;
      s_stat[grp * 2 + 1] = (float)(1.0 / sqrt(var + (double)1e-5f));
    }
  }
}


const char* lo_prof_geom_name(const char* base, const LoGeom& g);
const char* lo_prof_intern(const char* text);   // stable copy of a formatted name while the profiler is on   // per-layer profiler names under LO_PROF_LAYERS

enum LoConvKind {
  LO_CONV3_S1 = 0,        // Conv2d k3 s1 p1 forward
  LO_CONV3_S2 = 1,        // Conv2d k3 s2 p1 forward
  LO_CONVT4_S2 = 2,       // ConvTranspose2d k4 s2 p1 forward (4 sub-pixel phases)
  LO_CONV3_S1_DGRAD = 3,  // data gradient of k3 s1 p1
  LO_CONV3_S2_DGRAD = 4,  // data gradient of k3 s2 p1 (4 phases: 1,2,2,4 taps)
  LO_CONVT4_S2_DGRAD = 5, // data gradient of ConvTranspose k4 s2 p1 (= conv k4 s2 p1)
  LO_LINEAR = 6           // 1x1 "conv" on a [M,1,1,K] tensor: y = x W^T
};

// Build the geometry.  (H, W, Cin) describe the tensor this op READS, Cout the channels it WRITES.
int lo_make_geom(LoGeom* g, int kind, int B, int H, int W, int Cin, int Cout);

__device__ __forceinline__ float lo_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// mish(u) = u * tanh(softplus(u)) = u * t/(t+2), t = w(w+2), w = e^u   (lunar_generate.py:24-26; nn.Mish)
// Two-wide forms for the GroupNorm kernels (VALU-bound: v_pk_mul_f32 / v_pk_fma_f32 process two fp32 values per lane and
// instruction).  With p = e^u + 1:  t = p^2 - 1,  t + 2 = p^2 + 1,  tau = t r,  r = 1 / (p^2 + 1),
// mish = u tau,  mish' = tau + 4 u w p r^2.
typedef float lo_f2 __attribute__((ext_vector_type(2)));
// The two-wide helpers below are inlined into several kernels that must produce the SAME bits (the GroupNorm passes and the conv
// epilogues that fuse them; tests/test_gn_fuse_gpu.py): floating-point contraction is off inside them and every fused
// multiply-add is written out, so that no call site gets its own choice of which products to contract.
__device__ __forceinline__ lo_f2 lo_fma2(lo_f2 a, lo_f2 b, lo_f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ void lo_mish_parts2(lo_f2 u, lo_f2& w, lo_f2& p, lo_f2& r, lo_f2& tau) {
#pragma clang fp contract(off)
  const lo_f2 c = __builtin_elementwise_min(u, (lo_f2){20.0f, 20.0f});
  w = (lo_f2){__expf(c[0]), __expf(c[1])};
  p = w + 1.0f;
  const lo_f2 pp = p * p;
  const lo_f2 s = pp + 1.0f;
  r = (lo_f2){__builtin_amdgcn_rcpf(s[0]), __builtin_amdgcn_rcpf(s[1])};
  tau = (pp - 1.0f) * r;
}
__device__ __forceinline__ lo_f2 lo_mish2(lo_f2 u) {
#pragma clang fp contract(off)
  lo_f2 w, p, r, tau;
  lo_mish_parts2(u, w, p, r, tau);
  return u * tau;
}
__device__ __forceinline__ lo_f2 lo_mish_grad2(lo_f2 u) {
#pragma clang fp contract(off)
  lo_f2 w, p, r, tau;
  lo_mish_parts2(u, w, p, r, tau);
  return lo_fma2(u * 4.0f, (w * p) * (r * r), tau);
}
// mish(u) = u tanh(softplus(u)) = u t / (t + 2),  t = e^u (e^u + 2): one v_exp_f32 and one v_rcp_f32 (1 ulp; an IEEE
// division costs ~10 more VALU instructions and the GroupNorm kernels are VALU-bound on exactly this arithmetic)
__device__ __forceinline__ float lo_mish(float u) {
  float w = __expf(fminf(u, 20.0f));
  float t = w * (w + 2.0f);
  return u * t * __builtin_amdgcn_rcpf(t + 2.0f);
}
// d mish / du = tau + u (1 - tau^2) sigmoid(u),  tau = t / (t + 2);  (1 - tau^2) sigmoid(u) = 4 w (w + 1) / (t + 2)^2
__device__ __forceinline__ float lo_mish_grad(float u) {
  float w = __expf(fminf(u, 20.0f));
  float t = w * (w + 2.0f);
  float r = __builtin_amdgcn_rcpf(t + 2.0f);
  return t * r + u * (4.0f * w * (w + 1.0f)) * (r * r);
}

// ---------------------------------------------------------------------------------------------
// GroupNorm + Mish fused into the epilogue of the convolution that produces the tensor (forward).  GroupNorm(8, C) is per
// sample, so nothing here is a grid-wide dependency: the workgroups that hold the tiles of ONE sample exchange their per-group
// (sum, sum of squares) through memory and each normalises the tile it still has in LDS -- the separate lo_gn_fwd pass (one
// more read of the raw output, one launch, one kernel boundary per layer) disappears.
//
// Hand-off (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility", third row of the table of
// hand-offs measured with sc1 loads in place of an acquire -- no cache maintenance instruction anywhere):
//   producer: wave 0 writes its tile's exchange line, 32 floats = one whole 128-byte line, with ONE sc1 store instruction
//             (agent-scope relaxed atomic stores), waits for vmcnt(0), then lane 0 adds 1 to the sample's arrival counter
//             (agent-scope atomic);
//   consumer: lane 0 polls that counter with sc1 loads until every tile of the sample has arrived (counters are monotonic: the
//             host passes the value they reach at the end of THIS launch, so they are never reset), a workgroup barrier, then
//             64 threads read the sample's lines with sc1 loads and form mean / rstd exactly like lo_gn_group_stats (double
//             accumulation, same order: bitwise the statistics of the unfused path).
// Progress: a waiting workgroup holds its CU slot, so the tiles of a sample must be dispatched close together -- the launchers
// order tile ids sample-major and keep a sample's tiles on one XCD (lo_xcd_remap hands each XCD a contiguous range of tile
// ids; workgroups of one XCD are dispatched in order), which bounds the wait by the skew of a few neighbouring workgroups.  The
// poll is bounded anyway: after LO_GNF_SPIN_MAX polls the workgroup sets *fail and carries on (garbage out, no hang); the host
// reads the word with the step's losses.
// ---------------------------------------------------------------------------------------------
#define LO_GNF_SPIN_MAX (1 << 21)
#define LO_GNF_MAX_TILES 256        // exchange lines per sample and layer the workspace provides
struct LoGnFuse {
  float* xbuf;              // [B][MTs][NT][32] exchange lines
  unsigned int* counter;    // [B] arrival counters (monotonic)
  unsigned int target;      // counter value once every tile of this launch has arrived
  int MTs, NT;              // partial rows per sample (m tiles x phases) and n tiles per row of the launch
  const float* gamma;
  const float* beta;
  const f16* other;         // mode 1: skip tensor, mode 2: identity of the ResBlock; same layout as y
  f16* y;                   // activation out; nullptr = fusion off
  float* stats;             // [B][8][2] mean, rstd (saved for the backward)
  int mode;                 // 0 y = mish(u), 1 y = mish(u) + other, 2 y = mish(mish(u) + other)   (lo_gn_fwd's modes)
  unsigned int* fail;
};

// Every thread of the workgroup calls this (contains barriers).  s_x[32]: this tile's exchange line in LDS (slot grp*2 + which
// for the groups the tile covers, zero elsewhere); on return s_stat[grp*2] = mean, s_stat[grp*2+1] = rstd of sample n.
__device__ __forceinline__ void lo_gn_rendezvous(const LoGnFuse& f, int n, int row, int nt_i, int G, int BN, float inv_m,
                                                 const float* s_x, float* s_stat, int tid) {
  if (tid < 64) {
    if (tid < 32) {
      unsigned int* line = reinterpret_cast<unsigned int*>(f.xbuf) + (((size_t)n * f.MTs + row) * f.NT + nt_i) * 32;
      __hip_atomic_store(line + tid, __float_as_uint(s_x[tid]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the line (and this wave's earlier stores) have left before the count
    if (tid == 0) {
      __hip_atomic_fetch_add(f.counter + n, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0;
      while ((int)(__hip_atomic_load(f.counter + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - f.target) < 0) {
        __builtin_amdgcn_s_sleep(4);
        if (++spins > LO_GNF_SPIN_MAX) { __hip_atomic_store(f.fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      }
    }
  }
  __syncthreads();
  if (tid < 64) {     // same arithmetic and order as lo_gn_group_stats: group = tid >> 3, part = tid & 7
    const int grp = tid >> 3, part = tid & 7;
    const int nt_of = (grp * G) / BN;             // the n tile whose line carries this group
    double s = 0.0, q = 0.0;
    for (int mt = part; mt < f.MTs; mt += 8) {
      const unsigned int* line = reinterpret_cast<const unsigned int*>(f.xbuf) + (((size_t)n * f.MTs + mt) * f.NT + nt_of) * 32 + grp * 2;
      s += (double)__uint_as_float(__hip_atomic_load(line, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      q += (double)__uint_as_float(__hip_atomic_load(line + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
      s += __shfl_xor(s, o, 64);
      q += __shfl_xor(q, o, 64);
    }
    if (part == 0) {
      const double mean = s * inv_m;
      double var = q * inv_m - mean * mean;
      if (var < 0.0) var = 0.0;
      s_stat[grp * 2 + 0] = (float)mean;
      s_stat[grp * 2 + 1] = (float)(1.0 / sqrt(var + (double)1e-5f));
    }
  }
  __syncthreads();
}

// GroupNorm affine folded with the statistics: u = v * sc + sh,  sc = gamma * rstd,  sh = beta - mean * sc
__device__ __forceinline__ void lo_gn_scale_shift(float gamma, float beta, float mean, float rstd, float& sc, float& sh) {
#pragma clang fp contract(off)
  sc = gamma * rstd;
  sh = __builtin_fmaf(-mean, sc, beta);
}
// y = GroupNorm-affine + Mish of eight consecutive channels (one 16-byte chunk): lo_gn_fwd's arithmetic, shared with the conv
// epilogues that fuse it.  o: the second operand of modes 1 / 2.  yf (may be null): the eight results before the fp16 rounding.
__device__ __forceinline__ f16x8 lo_gn_apply8(f16x8 h, const float* sc, const float* sh, int mode, f16x8 o, float* yf = nullptr) {
#pragma clang fp contract(off)
  f16x8 y;
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    const lo_f2 hv = {(float)h[j], (float)h[j + 1]};
    lo_f2 m = lo_mish2(lo_fma2(hv, (lo_f2){sc[j], sc[j + 1]}, (lo_f2){sh[j], sh[j + 1]}));
    if (mode != 0) {
      const lo_f2 ov = {(float)o[j], (float)o[j + 1]};
      m = mode == 1 ? m + ov : lo_mish2(m + ov);
    }
    y[j] = (f16)m[0];
    y[j + 1] = (f16)m[1];
    if (yf) { yf[j] = m[0]; yf[j + 1] = m[1]; }
  }
  return y;
}

// ---- GroupNorm backward, element arithmetic shared by lo_gn_bwd_reduce / lo_gn_bwd_apply (lo_norm.hip) and the data-gradient
// epilogues that fuse the apply pass (same bits: contraction off, fused multiply-adds written out)
//   xhat = v rstd + nmr (nmr = -mean rstd),  u = v sc + sh,  du = dy mish'(u)   (modes PLAIN / SKIP)
__device__ __forceinline__ void lo_gn_du2_plain(lo_f2 hv, lo_f2 dyv, lo_f2 sc, lo_f2 sh, lo_f2 rstd, lo_f2 nmr, lo_f2& du, lo_f2& xhat) {
#pragma clang fp contract(off)
  xhat = lo_fma2(hv, rstd, nmr);
  du = dyv * lo_mish_grad2(lo_fma2(hv, sc, sh));
}
//   dv = rstd (gamma du - k1 - xhat k2) = du sc - kb - xhat kc   (sc = gamma rstd, kb = rstd k1, kc = rstd k2)
__device__ __forceinline__ lo_f2 lo_gn_dv2(lo_f2 du, lo_f2 xhat, lo_f2 sc, float kb, float kc) {
#pragma clang fp contract(off)
  return lo_fma2(-xhat, (lo_f2){kc, kc}, lo_fma2(du, sc, (lo_f2){-kb, -kb}));
}

// One lane announces this workgroup's arrival at a monotonic counter and waits (bounded) until it reaches `target`; every thread
// of the workgroup calls this.  The caller has already made its payload visible: every storing wave waited for vmcnt(0) and the
// workgroup passed a barrier (MI355X_MICROARCH.md, hand-off table: sc1 stores, one lane counts for the whole workgroup).
__device__ __forceinline__ void lo_arrive_and_wait(unsigned int* counter, unsigned int target, unsigned int* fail, int tid) {
  if (tid == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while ((int)(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
      __builtin_amdgcn_s_sleep(4);
      if (++spins > LO_GNF_SPIN_MAX) { __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
  }
  __syncthreads();
}

// ---- counter RNG of the teacher's dropout layers (nn.Dropout / nn.Dropout2d, lunar_evaluator.py:97-99,139-140,212,225,246,
// 253,353-397).  A dropout SITE of one forward call owns two 32-bit keys (lo_drop_site_keys: splitmix64 of the call's 64-bit
// seed and the site number); element `idx` of the site's tensor (this library's own index order, stated at each use) takes
// 16 bits of word(idx >> 1):  keep  <=>  bits >= thr,  thr = round(p * 65536)  (keep rate 1 - thr/65536; p = 0.1 -> 0.899994).
// Kept elements are scaled by 1/(1-p) like torch.  oracle/dropout_ref.py restates these three functions in numpy (bit-exact),
// so the CPU oracle and the reference (through forward hooks, oracle/make_golden.py) can run with the very same masks.
__host__ __device__ __forceinline__ uint64_t lo_splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
struct LoDropSite { uint32_t k0, k1; };
__host__ __device__ __forceinline__ LoDropSite lo_drop_site_keys(uint64_t call_seed, uint32_t site) {
  const uint64_t k = lo_splitmix64(call_seed ^ (0xD1B54A32D192ED03ull * (uint64_t)(site + 1u)));
  return LoDropSite{(uint32_t)k, (uint32_t)(k >> 32)};
}
__host__ __device__ __forceinline__ uint32_t lo_drop_word(LoDropSite s, uint32_t pair) {
  uint32_t x = pair ^ s.k0;
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  x += s.k1;
  x *= 0x9E3779B1U; x ^= x >> 15;
  return x;
}
// keep decision of element idx (one word per call: for runs of consecutive indices use lo_drop_word + the two halves)
__host__ __device__ __forceinline__ bool lo_drop_keep(LoDropSite s, uint32_t idx, uint32_t thr) {
  return ((lo_drop_word(s, idx >> 1) >> ((idx & 1u) * 16u)) & 0xFFFFu) >= thr;
}
// eight consecutive elements idx0 .. idx0+7 (idx0 % 8 == 0): bit j of the result = keep(idx0 + j)
__device__ __forceinline__ uint32_t lo_drop_keep8(LoDropSite s, uint32_t idx0, uint32_t thr) {
  uint32_t m = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const uint32_t x = lo_drop_word(s, (idx0 >> 1) + w);
    m |= ((x & 0xFFFFu) >= thr ? 1u : 0u) << (2 * w);
    m |= ((x >> 16) >= thr ? 1u : 0u) << (2 * w + 1);
  }
  return m;
}
// dropout sites of one teacher forward (E <= 8 experts, 3 blocks each)
#define LO_DS_FE 0u                                   // feature extractor, [B][HW][192] (NHWC index)
#define LO_DS_GATE 1u                                 // gate hidden layer, [B][256]
#define LO_DS_BLOCK(e, l, k) (2u + ((e) * 3u + (l)) * 4u + (k))   // k: 0 conv1 Dropout2d [B][C], 1 attn_drop [B][543][8][32],
                                                                  //    2 proj_drop [B][HW][C] (NHWC), 3 conv2 Dropout2d [B][C]
#define LO_DS_QUALITY(e) (100u + (e))                 // quality head hidden layer, [B][64]
#define LO_DS_SEM 110u                                // semantic head hidden layer [B][128]
#define LO_DS_STYLE 111u
#define LO_DS_PROMPT 112u

// ---- fp8 operand path (OCP e4m3, gfx950): activations are stored as fp8(value * LO_F8_ACT_SCALE) with saturation at the
// e4m3 maximum (448): 8 covers |value| <= 56 with 3 mantissa bits down to 2^-9 (subnormal step 2.4e-4); the GroupNorm +
// Mish outputs these buffers hold are O(1).  Weights carry one scale per output channel (lo_pack_f8_kernel).
#define LO_F8_ACT_SCALE 8.0f
#define LO_F8_MAX 448.0f
__device__ __forceinline__ uint32_t lo_pack4_fp8(float a, float b, float c, float d) {
  a = fminf(fmaxf(a, -LO_F8_MAX), LO_F8_MAX); b = fminf(fmaxf(b, -LO_F8_MAX), LO_F8_MAX);
  c = fminf(fmaxf(c, -LO_F8_MAX), LO_F8_MAX); d = fminf(fmaxf(d, -LO_F8_MAX), LO_F8_MAX);
  int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
  return (uint32_t)r;
}
