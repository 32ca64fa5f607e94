// Internal launcher prototypes shared by the kernel translation units and the C-ABI layer.
#pragma once
#include "lo_common.h"

const char* lo_get_error();

// lo_conv.hip
int lo_pack_weight(const float* w, f16* wp, const LoGeom& g, hipStream_t st);
int lo_pack_all(const LoPackJob* jobs_dev, int njobs, int nblocks, hipStream_t st, int block_base = 0);   // block_base: a sub-range of the table
int lo_pack_blocks(const LoGeom& g);   // blocks of one job in the fused pack launch
// Data-gradient epilogue that also runs the GroupNorm backward of the layer whose activation gradient it produces.  P1: the
// reduction (per-tile, per-channel sum du / sum du*xhat).  dv != null: the APPLY pass too -- the workgroups of a sample exchange
// their P1 rows (sc1 stores, arrival counter per (sample, n tile): lo_common.h lo_arrive_and_wait), form the group sums in the
// order lo_gn_bwd_apply uses and write dv (+ P2 = per-tile sums of dv for the conv bias gradient) instead of the activation
// gradient, which is then never stored.  lo_conv_gnb_apply_tiles: whether the kernel lo_conv_run picks supports it.
struct LoGnBwdFuse {
  const f16* v; const float* stats; const float* gamma; const float* beta; float* P1;
  f16* dv = nullptr; float* P2 = nullptr; unsigned int* counter = nullptr; unsigned int target = 0; unsigned int* fail = nullptr;
  bool keep_out = false;    // with dv: store the activation gradient too (somebody else reads it: the decoder's skip gradients)
};
bool lo_conv_gnb_apply_tiles(const LoGeom& gd, int* mts, int* nt);
int lo_conv_splitk_plan(const LoGeom& g);     // K splits for the few-rows convolutions (128 x 128 split-K tiles + a fused slab pass), 0 = no
// teacher epilogue: LeakyReLU(0.2), per-channel BN partial sums; out_pitch > 0: the output tensor has out_pitch channels per pixel
// and this op writes its Cout channels starting at channel out_choff (writing straight into a concatenated tensor)
struct LoConvExtra { int act; float* bn_partial; int out_pitch = 0; int out_choff = 0; };
int lo_conv_run(const LoGeom& g, const f16* in, const f16* wp, const float* bias, const f16* add_src, f16* out,
                float* gn_partial, float* slab, int nsplit, hipStream_t st, const LoGnBwdFuse* gb = nullptr,
                const LoConvExtra* ex = nullptr, const LoGnFuse* gf = nullptr);
// gf: GroupNorm + Mish of the output inside the epilogue (LoGnFuse, lo_common.h).  lo_conv_gn_fuse_tiles: whether the kernel
// lo_conv_run picks for g supports it, and the tile grid per sample the caller must put into gf->MTs / gf->NT
bool lo_conv_gn_fuse_tiles(const LoGeom& g, int* mts, int* nt);
bool lo_conv3_gn_fuse_tiles(const LoGeom& g, int* mts, int* nt);
// fp8 (e4m3) operand path of the forward convs
struct LoPackF8Job { const f16* src; uint8_t* dst; float* scale; int K[LO_MAX_PHASE]; int wofs[LO_MAX_PHASE]; int Cout, n_phase, block0; };
bool lo_conv_f8_applies(const LoGeom& g);
int lo_conv_run_f8(const LoGeom& g, const uint8_t* in8, const uint8_t* w8, const float* wscale, const float* bias, const f16* add_src,
                   f16* out, float* gn_partial, hipStream_t st);
void lo_pack_f8_job(LoPackF8Job* j, const LoGeom& g, const f16* src, uint8_t* dst, float* scale, int block0);   // blocks: n_phase * Cout
int lo_pack_f8_all(const LoPackF8Job* jobs_dev, int njobs, int nblocks, hipStream_t st, int block_base = 0);
int lo_pack_f8_one(const LoGeom& g, const f16* wp, uint8_t* w8, float* wscale, hipStream_t st);
int lo_quantize_f8(const f16* x, uint8_t* x8, size_t n, hipStream_t st);
int lo_conv_tile_m(const LoGeom& g);
int lo_conv_mt_f8(const LoGeom& g);   // the same for lo_conv_run_f8
int lo_conv_mt(const LoGeom& g);   // GroupNorm partial rows per sample the conv epilogue writes for this geometry
int lo_conv3_tiles_per_image(const LoGeom& g, bool need_bn = false);
int lo_conv_bn_rows(const LoGeom& g);
bool lo_conv3_pp_applies(const LoGeom& g);
int lo_conv3_run_pp_xf(const LoGeom& g, const f16* in, const f16* xc, const f16* kx, int nlev, const f16* wp, const float* bias,
                       f16* out, hipStream_t st, const LoConvExtra* ex);   // BatchNorm partial rows written by lo_conv_run(..., ex) for this geometry
bool lo_conv3_pp_f8_applies(const LoGeom& g);
int lo_conv3_run_pp_f8(const LoGeom& g, const uint8_t* in8, const uint8_t* w8, const float* wscale, const float* bias, f16* out,
                       hipStream_t st, const LoConvExtra* ex);
int lo_conv3_run(const LoGeom& g, const f16* in, const f16* wp, const float* bias, const f16* add_src, f16* out,
                 float* gn_partial, hipStream_t st, const LoConvExtra* ex = nullptr, const LoGnBwdFuse* gb = nullptr,
                 const LoGnFuse* gf = nullptr);
bool lo_conv3_fuses_gnb(const LoGeom& g);
bool lo_convt4_patch_applies(const LoGeom& g);   // patch-resident forward of the 64 -> 32 transposed convolution
int lo_convt4_patch_tiles_per_image(const LoGeom& g);
int lo_convt4_patch_run(const LoGeom& g, const f16* in, const f16* wp, const float* bias, f16* out, float* gn_partial, hipStream_t st);
bool lo_convs2d_patch_applies(const LoGeom& g);   // the same kernel as the data gradient of the stride-2 conv 64 -> 128 (all four phases from one dy patch)
int lo_convs2d_patch_run(const LoGeom& g, const f16* dy, const f16* wp, const f16* add_src, f16* dx, hipStream_t st);
int lo_conv_gnb_rows(const LoGeom& g);   // P1 rows per sample written by lo_conv_run(g, ..., gb) (fused GroupNorm-backward epilogue)
int lo_splitk_reduce(const float* slab, const float* bias, float* out32, f16* out16, int M, int N, int nsplit, hipStream_t st);
int lo_wgrad_nsplit(const LoGeom& g);
int lo_wgrad3_nsplit(const LoGeom& g);   // multi-tap 3x3 stride-1 weight-gradient kernel: pixel splits, 0 = does not apply
int lo_wgrad3_run(const LoGeom& g, const f16* x, const f16* dy, float* slab, hipStream_t st, int* nsplit_out);
int lo_wgrad2_nsplit(const LoGeom& g);   // multi-tap weight-gradient kernel of the stride-2 layers: position splits, 0 = does not apply
int lo_wgrad2_run(const LoGeom& g, const f16* x, const f16* dy, float* slab, hipStream_t st, int* nsplit_out);
int lo_wgrad_run(const LoGeom& g, const f16* x, const f16* dy, float* slab, float* grad, float scale, hipStream_t st);
size_t lo_wgrad_slab_bytes(const LoGeom& g);
size_t lo_packed_weight_elems(const LoGeom& g);

// lo_norm.hip
int lo_gn_nchunk(int HW, int C);
int lo_gn_fwd(const f16* v, const float* partial, int MT, const float* gamma, const float* beta, const f16* other, f16* y,
              float* stats, int B, int HW, int C, int mode, hipStream_t st, uint8_t* y8 = nullptr);   // y8: e4m3 copy of y * LO_F8_ACT_SCALE
int lo_gn_bwd(const f16* dy, const f16* v, const f16* other, const float* stats, const float* gamma, const float* beta,
              f16* ds, f16* dv, float* P1, float* P2, float* dgamma, float* dbeta, float* dbias, int B, int HW, int C,
              int mode, float scale, hipStream_t st);
struct LoGnFinJob { const float* P1; const float* P2; float* dgamma; float* dbeta; float* dbias; int nblk1, nblk2, C, block0; };
#define LO_GN_FIN_MAX 16
struct LoGnFinJobs { LoGnFinJob j[LO_GN_FIN_MAX]; int n; };
// np1 = 0: run the reduce pass here (P1 gets nchunk rows per sample); np1 > 0: P1 already holds np1 rows per sample
// (written by the fused data-gradient epilogue) and only the apply pass runs
int lo_gn_bwd_nofinal(const f16* dy, const f16* v, const f16* other, const float* stats, const float* gamma, const float* beta,
                      f16* ds, f16* dv, float* P1, float* P2, int B, int HW, int C, int mode, hipStream_t st, int np1 = 0);
int lo_gn_finalize_all(const LoGnFinJobs& jobs, float scale, hipStream_t st);
// Split-K convolution outputs (fp32 slabs [nsplit][B*HW][C]) consumed by ONE (sample, group)-local pass each:
//   forward:  v = fp16(bias + sum of the slabs), GroupNorm statistics of the group, y = GroupNorm + Mish (lo_gn_fwd's modes)
//   backward: dy = fp16(sum of the slabs + add_src), then the one-pass GroupNorm backward of lo_gn_bwd_local (plain mode)
int lo_splitk_gn_fwd(const float* slab, int nsplit, const float* bias, const float* gamma, const float* beta, const f16* other, f16* v,
                     f16* y, float* stats, int B, int HW, int C, int mode, hipStream_t st);
int lo_splitk_gn_bwd(const float* slab, int nsplit, const f16* add_src, const f16* v, const float* stats, const float* gamma,
                     const float* beta, f16* dy_out, f16* dv, float* P1, float* P2, int B, int HW, int C, hipStream_t st);
// reduce + apply in one pass where a (sample, group) fits a workgroup; ONE row of P1 / P2 per sample
bool lo_gn_bwd_local_applies(int HW, int C);
int lo_gn_bwd_local(const f16* dy, const f16* v, const f16* other, const float* stats, const float* gamma, const float* beta, f16* ds,
                    f16* dv, float* P1, float* P2, int B, int HW, int C, int mode, hipStream_t st);
int lo_nhwc_to_nchw_f16(const f16* src, f16* dst, int B, int HW, int C, hipStream_t st);
int lo_nchw_to_nhwc_f16(const f16* src, f16* dst, int B, int HW, int C, hipStream_t st, uint8_t* dst8 = nullptr);
int lo_nhwc_f16_to_nchw_f32(const f16* src, float* dst, int B, int HW, int C, float scale, hipStream_t st);   // module-boundary forms
int lo_nchw_f32_to_nhwc_f16(const float* src, f16* dst, int B, int HW, int C, float scale, hipStream_t st);

// lo_edge.hip
int lo_first_conv_fwd(const float* x, const float* w, const float* bias, f16* v, float* gn_partial, int B, hipStream_t st);
int lo_first_conv_wgrad(const float* x, const f16* dv, float* partial, float* dw, int B, float scale, hipStream_t st);
int lo_colsum(const float* partial, float* out, int nrow, int ncol, int stride, float scale, hipStream_t st);
int lo_final_conv_fwd(const f16* a4, const float* w, const float* bias, const float* target, float* recon,
                      float* mse_partial, int B, hipStream_t st);
int lo_final_conv_bwd(const f16* a4, const float* w, const float* recon, const float* target, const float* drecon,
                      const float* coef, float gscale, f16* da4, float* partial, float* dw, float* db, int B, float scale,
                      hipStream_t st, hipStream_t sum_st = nullptr, hipEvent_t sum_after = nullptr);   // sum_st: where the two column sums (dw, db) run

// lo_train.hip
int lo_head_reduce(const float* slab, const float* bias, const float* eps_in, uint64_t seed, float* mu, float* logvar,
                   f16* z, float* eps_out, float* kl_partial, int B, int L, int nsplit, hipStream_t st, float* mu_user = nullptr,
                   float* logvar_user = nullptr);
int lo_loss_finalize(const float* mse_partial, int n_mse, const float* kl_partial, int n_kl, float rw, float kw, float adv,
                     const float* adv_dev, float accum, float ls, float* losses, float* coefs, float n_rec, float n_lat,
                     hipStream_t st);
int lo_latent_bwd(const f16* dz, const float* mu, const float* logvar, const float* eps, const float* coefs,
                  const float* gmu, const float* glv, float gscale, f16* dml, int B, int L, hipStream_t st);
int lo_colsum_f16(const f16* x, float* out, int M, int N, float scale, hipStream_t st);
int lo_cast_f32_f16(const float* src, f16* dst, size_t n, hipStream_t st);
int lo_scale_f32(float* x, size_t n, float scale, hipStream_t st);
int lo_dp_pack_f16_run(const float* g, f16* wire, size_t n, float scale, hipStream_t st);     // data-parallel exchange helpers
int lo_dp_unpack_f16_run(const f16* wire, float* g, size_t n, float inv_scale, hipStream_t st);
int lo_dp_unpack_f16_sumsq_run(const f16* wire, float* g, size_t n, float inv_scale, float* scratch, hipStream_t st);
int lo_dp_sum_shares_run(const void* recv, void* share, int world, size_t chunk, int is_f16, float inv_world, hipStream_t st);
int lo_transpose_cast(const float* src, f16* dst, int R, int C, hipStream_t st);
int lo_gradnorm(const float* g, size_t n, float max_norm, float* partial, float* norm_out, hipStream_t st,
                const unsigned int* fail = nullptr);   // fail: the engine's rendezvous-failure word (non-zero skips the update)
int lo_sumsq_range(const float* g, size_t begin, size_t end, float* partial, hipStream_t st);   // -> partial[512 .. 1024)
int lo_gradnorm_split(const float* g, size_t presummed_begin, float max_norm, float* partial, float* norm_out, hipStream_t st,
                      const unsigned int* fail = nullptr);
// upstream-gradient normalisation of the nn.Module boundary (GradScaler-scaled gradients; see lo_train.hip)
int lo_grad_scale_pick_run(const float* const g[5], const size_t n[5], float* scratch, hipStream_t st);
int lo_scale_copy_dev_run(const float* src, float* dst, size_t n, const float* scale_dev, hipStream_t st);
int lo_scale_dev_run(float* x, size_t n, const float* scale_dev, const unsigned int* fail, hipStream_t st);
int lo_adamw(float* p, const float* g, float* m, float* v, size_t n, const float* norm, float lr, float beta1, float beta2,
             float eps, float wd, int step, hipStream_t st, f16* cast = nullptr);

// lo_attn.hip
int lo_selfattn2d_fwd(const float* x, const float* wq, const float* bq, const float* wk, const float* bk, const float* wv,
                      const float* bv, const float* gamma, float* q, float* k, float* v, float* out, int B, int C, int N,
                      hipStream_t st);
size_t lo_selfattn2d_bwd_scratch(int B, int C, int N);
int lo_selfattn2d_bwd(const float* x, const float* wq, const float* wk, const float* wv, const float* gamma, const float* q,
                      const float* k, const float* v, const float* dy, float* scratch, float* dx, float* dwq, float* dbq,
                      float* dwk, float* dbk, float* dwv, float* dbv, float* dgamma, int B, int C, int N, hipStream_t st);
int lo_decode_sprites(const uint8_t* u8, float* out, int B, hipStream_t st);

// lo_lowrank.hip: rank-B Linear-layer weight gradients as factors (Gram-matrix norm, AdamW that forms the gradient tiles itself)
int lo_lowrank_bp(int B);                          // batch rounded up to the MFMA K step (32)
bool lo_lowrank_applies(int B, int N, int K);
int lo_transpose_pad_f16_multi(const f16* const* src, f16* const* dst, const int* C, int njobs, int R, int Rp, hipStream_t st);
struct LoLowrankNorm { const f16* fshort; int n_short; const f16* flong; int n_long; float* gram; float* partial; int nslots; };
int lo_lowrank_sumsq(const LoLowrankNorm* layers, int nlayers, int B, float scale, hipStream_t st);
struct LoLowrankMat { float* p; float* m; float* v; f16* cast; f16* cast_t; const f16* xt; const f16* yt; int N, K; };   // W [N][K] (cast_t [K][N]); xt [K][Bp], yt [N][Bp]
int lo_adamw_lowrank(const LoLowrankMat* mats, int nmat, int B, float gscale, const float* norm, float lr, float beta1, float beta2,
                     float eps, float wd, int step, hipStream_t st);
int lo_lowrank_materialize(float* gout, const f16* xt, const f16* yt, int N, int K, int B, float gscale, hipStream_t st);
int lo_lowrank_materialize_gathered(float* gout, const f16* xt, const f16* yt, size_t rank_stride_elems, int world, int N, int K, int B,
                                    float gscale, hipStream_t st);
int lo_sumsq_blocks(const float* g, size_t n, float* partial, int nblocks, hipStream_t st);   // lo_train.hip: partial[0 .. nblocks)
