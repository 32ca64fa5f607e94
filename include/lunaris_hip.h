/* lunaris_hip.h — C ABI of liblunaris_hip.so: the MI355X (gfx950) kernels behind the Lunaris-Orion VAE training step.
 *
 * The reference (MeryylleA/Lunaris-Orion) has no FFI / operator-plugin interface: its hot path
 * (TrainingManager._process_batch, /root/reference/train_hybrid.py:838-954) calls PyTorch ATen ops directly.  This
 * header therefore DEFINES the boundary; every entry point cites the reference lines whose arithmetic it replaces.
 * INTEGRATION.md shows the ctypes binding the Python host uses (lunaris_orion_amd/_lib.py).
 *
 * Conventions
 *   - every pointer is a raw DEVICE pointer (tensor.data_ptr()); the caller owns all memory, including workspaces;
 *     the library never allocates or frees device memory and never synchronises the device;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = the default stream);
 *   - return value: 0 = ok, <0 = error class (-1 bad argument/unsupported shape, -2 HIP error, -3 bad state); the
 *     message is available from lo_last_error() (thread-local);
 *   - activations inside the library are fp16 NHWC; tensors crossing the boundary are fp32 in PyTorch's layouts
 *     (images / reconstructions NCHW, parameters in their nn.Module layouts);
 *   - gradients flowing between kernels are multiplied by `loss_scale` (fp16 range); every parameter gradient that
 *     leaves the library is un-scaled (true gradient, fp32).
 */
#ifndef LUNARIS_HIP_H
#define LUNARIS_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* lo_last_error(void);
int lo_version(void);

/* Per-launch timing with HIP events recorded on the launch stream (used by bench.py's roofline leg).  Enable, run,
 * synchronise the stream, then read record i: kernel name, elapsed ms, algorithmic FLOPs and bytes of that launch.
 * on = 1: records carry the kernel name; on = 2: conv / GroupNorm launches of the VAE step are named per layer
 * ("fwd kind<k> HxW Cin->Cout", "dgrad L<n> ...", "wgrad L<n> ..."), which is how bench.py sums the encoder conv stack. */
void lo_prof_enable(int on);
int lo_prof_count(void);
int lo_prof_get(int i, char* name, int name_cap, double* ms, double* flops, double* bytes);

/* ---- op kinds for the generic implicit-GEMM entry points ---------------------------------------------------- */
enum {
  LO_KIND_CONV3_S1 = 0,        /* nn.Conv2d(k3,s1,p1)           lunar_generate.py:36,41          */
  LO_KIND_CONV3_S2 = 1,        /* nn.Conv2d(k3,s2,p1)           lunar_generate.py:102,109,116    */
  LO_KIND_CONVT4_S2 = 2,       /* nn.ConvTranspose2d(k4,s2,p1)  lunar_generate.py:169,175,181,187 */
  LO_KIND_CONV3_S1_DGRAD = 3,  /* aten::convolution_backward (input grad) of the above           */
  LO_KIND_CONV3_S2_DGRAD = 4,
  LO_KIND_CONVT4_S2_DGRAD = 5,
  LO_KIND_LINEAR = 6           /* nn.Linear                      lunar_generate.py:124,125,165    */
};

/* ---- single-op entry points (used by the parity tests; the step executor below calls the same launchers) ----- */

/* Packed fp16 weight size (elements) for an op of `kind` reading [B,H,W,Cin] and writing Cout channels. */
size_t lo_packed_weight_elems_for(int kind, int B, int H, int W, int Cin, int Cout);
/* canonical fp32 weight (nn.Module layout) -> packed fp16 operand of the implicit GEMM */
int lo_pack_weight_for(int kind, int B, int H, int W, int Cin, int Cout, const float* w, void* wp, void* stream);
/* out = conv(in) + bias (+ add_src); optional GroupNorm partial sums [B][MT][8][2]; returns MT through *mt_out.
 * in/out/add_src: fp16 NHWC.  Replaces aten::convolution / aten::addmm. */
int lo_conv_forward(int kind, int B, int H, int W, int Cin, int Cout, const void* in, const void* wp, const float* bias,
                    const void* add_src, void* out, float* gn_partial, int* mt_out, void* stream);
/* e4m3 operand forms (see LO_VAE_FP8_FWD below; Cin % 128 == 0, Cout % 64 == 0): x8 = fp8(8 * x16) saturating at 448;
 * wp8 / wscale[n_phase][Cout] from the packed fp16 weight (one scale per output channel: amax / 448, folded with 1/8);
 * the conv itself, same outputs as lo_conv_forward. */
int lo_quantize_act_f8(const void* x16, void* x8, size_t n, void* stream);
int lo_pack_weight_f8_for(int kind, int B, int H, int W, int Cin, int Cout, const void* wp16, void* wp8, float* wscale, void* stream);
int lo_conv_forward_f8(int kind, int B, int H, int W, int Cin, int Cout, const void* in8, const void* wp8, const float* wscale,
                       const float* bias, const void* add_src, void* out, float* gn_partial, int* mt_out, void* stream);
/* The fused-tap 3x3 stride-1 convolution of the teacher (lunar_evaluator.py:242-243,249-250: Conv2d k3 p1 + LeakyReLU(0.2), and
 * the per-channel sums its BatchNorm needs) on fp16 or e4m3 operands; H, W % 16 == 0, Cin % 64 == 0 (fp8: % 128), Cout % 128 == 0,
 * Cin <= 128.  bn_partial (may be NULL): [B*(H/16)*(W/16)][Cout][2] = (sum, sum of squares) of the stored values per pixel tile. */
int lo_conv3x3_fused_tap_forward(int B, int H, int W, int Cin, int Cout, int fp8, const void* in, const void* wp, const float* wscale,
                                 const float* bias, int leaky_relu, void* out, float* bn_partial, void* stream);
/* Linear with split-K: y[M,N] (fp32 and/or fp16) = x[M,K] Wp[N,K]^T + bias.  slab: nsplit*M*N floats. */
int lo_linear_splitk(int M, int K, int N, const void* x, const void* wp, const float* bias, float* slab, int nsplit,
                     float* out32, void* out16, void* stream);
/* weight gradient of a FORWARD op `kind` (0,1,2,6): grad (canonical fp32 layout) = scale * sum_pixels dy (x) x. */
size_t lo_wgrad_slab_bytes_for(int kind, int B, int H, int W, int Cin, int Cout);
int lo_conv_wgrad(int kind, int B, int H, int W, int Cin, int Cout, const void* x, const void* dy, float* slab,
                  float* grad, float scale, void* stream);

/* GroupNorm(8,C)+Mish forward on the raw conv output v (fp16 NHWC): mode 0 y=mish(u); 1 y=mish(u)+other;
 * 2 y=mish(mish(u)+other) (ResBlock tail, lunar_generate.py:49-53).  stats[B][8][2] = (mean, rstd) is written. */
int lo_gn_mish_forward(const void* v, const float* gn_partial, int MT, const float* gamma, const float* beta,
                       const void* other, void* y, float* stats, int B, int HW, int C, int mode, void* stream);
/* backward of the above: dv (fp16), ds (mode 2: gradient of the identity branch), dgamma/dbeta/dbias (fp32 * scale).
 * P1: B*nchunk*C*2 floats, P2: B*nchunk*C floats of scratch (nchunk = lo_gn_nchunk_for(HW,C) <= 256). */
int lo_gn_nchunk_for(int HW, int C);
int lo_gn_mish_backward(const void* dy, const void* v, const void* other, const float* stats, const float* gamma,
                        const float* beta, void* ds, void* dv, float* P1, float* P2, float* dgamma, float* dbeta,
                        float* dbias, int B, int HW, int C, int mode, float scale, void* stream);

/* first conv Conv2d(3,64,k3,s2,p1) on fp32 NCHW images (lunar_generate.py:95) and its weight gradient */
int lo_first_conv_forward(const float* x, const float* w, const float* bias, void* v, float* gn_partial, int B, void* stream);
int lo_first_conv_wgrad_op(const float* x, const void* dv, float* partial /*B*16*1728*/, float* dw, int B, float scale, void* stream);
/* final conv Conv2d(32,3,k3,p1)+tanh (+MSE partial sums, B*64 floats) (lunar_generate.py:192,227-228; train_hybrid.py:859) */
int lo_final_conv_forward(const void* a4, const float* w, const float* bias, const float* target, float* recon,
                          float* mse_partial, int B, void* stream);
int lo_final_conv_backward(const void* a4, const float* w, const float* recon, const float* target, const float* drecon,
                           const float* coef_dev, float gscale, void* da4, float* partial /*B*64*867*/, float* dw,
                           float* db, int B, float scale, void* stream);

/* PixelArtDataset.__getitem__ arithmetic (train_hybrid.py:181-182) for a whole batch on the device:
 * uint8 HWC [B,128,128,3] -> float32 CHW [B,3,128,128] = x/127.5 - 1. */
int lo_decode_sprites_u8(const void* u8_hwc, float* out_chw, int B, void* stream);

/* SelfAttention2d.forward (lunar_generate.py:68-78; the module is defined but never instantiated by the reference):
 * out = gamma * (V softmax(Q^T K)^T) + x with Q,K = 1x1 convs to C/8 channels, V = 1x1 conv to C channels.  x/out fp32
 * [B,C,N=H*W]; q,k ([B,C/8,N]) and v ([B,C,N]) are caller-provided scratch/outputs.  C and N multiples of 64, C <= 512. */
int lo_selfattn2d_forward(const float* x, const float* wq, const float* bq, const float* wk, const float* bk, const float* wv,
                          const float* bv, const float* gamma, float* q, float* k, float* v, float* out, int B, int C,
                          int N, void* stream);
/* Backward of the same module (autograd of lunar_generate.py:68-78): q, k, v are the forward's outputs; scratch:
 * lo_selfattn2d_backward_scratch_elems(B, C, N) floats.  Writes dx [B,C,N] and the gradients of the three 1x1 convs and of
 * gamma (not accumulated: overwritten).  No N x N tensor is materialised; all sums are in a fixed order. */
size_t lo_selfattn2d_backward_scratch_elems(int B, int C, int N);
int lo_selfattn2d_backward(const float* x, const float* wq, const float* wk, const float* wv, const float* gamma,
                           const float* q, const float* k, const float* v, const float* dy, float* scratch, float* dx,
                           float* dwq, float* dbq, float* dwk, float* dbk, float* dwv, float* dbv, float* dgamma, int B,
                           int C, int N, void* stream);

/* clip_grad_norm_ + AdamW over one flat fp32 buffer (train_hybrid.py:913,921; :504-509).  scratch: 1024+4 floats;
 * scratch[1024..1026] = (grad norm, clip coef, finite flag) afterwards; scratch[1027] counts the calls whose gradient norm was
 * not finite (those updates are skipped, like a step under torch.cuda.amp.GradScaler, train_hybrid.py:917-923). */
int lo_clip_adamw_step(float* p, const float* g, float* m, float* v, size_t n, float max_norm, float lr, float beta1,
                       float beta2, float eps, float weight_decay, int step, float* scratch, void* stream);

/* lo_clip_adamw_step for a caller whose backward already summed the squares of [presummed_begin, n) into scratch[512..1024)
 * (lo_vae_set_gradnorm_scratch): only [0, presummed_begin) is read for the norm. */
int lo_clip_adamw_step_presummed(float* p, const float* g, float* m, float* v, size_t n, size_t presummed_begin, float max_norm,
                                 float lr, float beta1, float beta2, float eps, float weight_decay, int step, float* scratch,
                                 void* stream);

/* ---- the VAE step executor (LunarisCoreVAE.forward / backward, lunar_generate.py:263-276) -------------------- */
typedef struct LoVae LoVae;
int lo_vae_create(int batch, int latent_dim, LoVae** out);
/* The same with mode flags.  LO_VAE_FP8_FWD (BASELINE config 5): the forward convolutions whose input channel count is a
 * multiple of 128 (ResBlock / stride-2 convs at 128-512 channels, transposed convs at 512 / 256 / 128) run on OCP e4m3
 * operands (v_mfma_scale_f32_16x16x128_f8f6f4, fp32 accumulation): activations as fp8(8 * value) written next to the fp16
 * copy by the producing GroupNorm kernel, weights with one scale per output channel.  The backward is unchanged (fp16
 * operands).  No reference counterpart: the reference's low-precision mode is torch.cuda.amp fp16 autocast
 * (train_hybrid.py:850).  Unknown flag bits are an error. */
#define LO_VAE_FP8_FWD 1u
int lo_vae_create_ex(int batch, int latent_dim, unsigned flags, LoVae** out);
void lo_vae_destroy(LoVae* h);
/* Early gradient norm: with a scratch set (the 1028-float scratch of lo_clip_adamw_step), a single-call lo_vae_backward takes the
 * sum of squares of the range lo_vae_phase1_grad_range names as soon as it is final, beside the encoder backward, and
 * lo_clip_adamw_step_presummed(.., presummed_begin = begin of that range, ..) reads only the rest.  lo_vae_gradnorm_presummed
 * says whether the next backward will do so (it needs the library's side stream).  NULL scratch: off. */
int lo_vae_set_gradnorm_scratch(LoVae* h, float* scratch);
/* The same partial sum for a caller that exchanges gradients between processes (data parallel: the norm is that of the AVERAGED
 * gradients, so the library cannot take it inside the backward): sum of squares of flat_grads[begin, end) into scratch[512..1024),
 * to be enqueued behind the exchange of that range, on whatever stream ran it; lo_clip_adamw_step_presummed / the `presummed` form
 * of lo_vae_optimizer_step then read only [0, begin).  clip_grad_norm_'s norm (train_hybrid.py:913), split in two. */
int lo_gradnorm_early_range(const float* flat_grads, size_t begin, size_t end, float* scratch, void* stream);
/* Asynchronous hand-over of the phased backward's gradient ranges (data parallel).  By default lo_vae_backward_phase(1) and (3)
 * return with `stream` ordered behind everything that writes their range (the library's side stream is joined).  With on = 1 they
 * do not hold `stream` up: the range's completion is left as an event, and lo_vae_wait_handover(h, s) makes stream `s` -- the one
 * that runs the exchange -- wait for it.  Call it once per phase 1 / 3, before phase 3 / 4 is enqueued.  Phases 2 and 4 and the
 * single-call backward are unchanged.  No reference counterpart (DDP's bucket hooks are the nearest thing). */
int lo_vae_set_async_handover(LoVae* h, int on);
int lo_vae_wait_handover(LoVae* h, void* stream);
int lo_vae_gradnorm_presummed(const LoVae* h);
/* parameters live in ONE flat fp32 buffer; tensor i (state_dict order, 72 tensors) starts at this element offset */
int lo_vae_num_params(const LoVae* h);
size_t lo_vae_param_offset(const LoVae* h, int index);
size_t lo_vae_param_numel(const LoVae* h, int index);
size_t lo_vae_flat_elems(const LoVae* h);
size_t lo_vae_workspace_bytes(const LoVae* h);
/* refresh the packed fp16 operand copies from the fp32 master parameters (call after every parameter update) */
int lo_vae_pack(LoVae* h, const float* flat_params, void* ws, void* stream);
/* clip_grad_norm_ + AdamW (train_hybrid.py:913,921) + lo_vae_pack in ONE call, pipelined with the next forward: encoder stages
 * 1..3 (3 % of the parameters, what the next forward reads first) are updated on `stream`; the last encoder stage, the Linear
 * layers and the decoder (97 %), their fp16 casts and packs follow on the library's side stream beside the next encoder forward
 * (lo_vae_forward waits where it needs them).  p / g / m / v: the flat fp32 buffers of lo_vae_flat_elems() elements; scratch as for lo_clip_adamw_step; flags:
 * LO_OPT_PRESUMMED as for lo_clip_adamw_step_presummed, LO_OPT_SERIAL for the un-pipelined order.  The clip kernel also reads the
 * plan's rendezvous-failure word (lo_vae_sync_fail_word): when set, the update is skipped on the device like one with a non-finite
 * norm (GradScaler semantics, train_hybrid.py:917-923) -- and stays skipped until the host has looked.  Afterwards parameters from encoder.down4 on may still be in flight: call lo_vae_join before
 * reading them on any stream other than through this executor. */
#define LO_OPT_PRESUMMED 1   /* scratch[512..1024) already holds the sum of squares of the phase-1 range */
#define LO_OPT_SERIAL 2      /* no pipelining: norm, AdamW of everything and lo_vae_pack in order on `stream` */
int lo_vae_optimizer_step(LoVae* h, float* p, const float* g, float* m, float* v, void* ws, float max_norm, float lr, float beta1,
                          float beta2, float eps, float weight_decay, int step, float* scratch, int flags, void* stream);
int lo_vae_join(LoVae* h, void* stream);   /* `stream` waits for the side-stream work of lo_vae_pack / lo_vae_optimizer_step */
/* Factored Linear-layer gradients.  fc_mu | fc_logvar ([2L, 32768]) and decoder.fc ([32768, L]) hold 82 % of the parameters
 * (lunar_generate.py:124-125, 165) and at batch B their weight gradients dW = dY^T X have rank B.  With the mode on (batch <= 128),
 * a FUSED single-call lo_vae_backward does not write those two gradients into flat_grads: it keeps the factors (transposed,
 * batch-padded fp16 copies in the workspace), takes their share of clip_grad_norm_'s norm (train_hybrid.py:913) from two B x B Gram
 * matrices per layer -- ||dY^T X||_F^2 = sum_{b,b'} (dY dY^T)[b,b'] (X X^T)[b,b'] -- into the scratch set by
 * lo_vae_set_gradnorm_scratch (required), and lo_vae_optimizer_step (LO_OPT_PRESUMMED required) forms each gradient tile with MFMA
 * inside its AdamW pass (train_hybrid.py:921).  Same update to fp32 rounding; 12 of 38 bytes of HBM traffic per parameter less.
 * Explicit-gradient backwards (the autograd path) and lo_vae_backward_phase always write every gradient.
 * lo_vae_materialize_linear_grads: writes the two gradients of the last fused backward into flat_grads after all (tests, tools). */
int lo_vae_set_linear_factored(LoVae* h, int mode);   /* 0 off, 1 single process (above), 2 data parallel (below) */
/* Data parallel (mode 2; no reference counterpart: the reference has no distributed code).  lo_vae_backward_phase(1) then leaves the
 * two Linear layers' factors instead of their weight gradients: one contiguous block of the workspace (lo_vae_factor_block:
 * dml^T | xflat^T | Gfc^T | z^T, transposed, batch-padded fp16; 8.6 MB at batch 64 / latent 512).  The ranks all-gather the blocks
 * (exact, and 8.6 MB per rank on the wire instead of 201 MB of gradients) and each calls lo_vae_materialize_gathered_linear_grads
 * on the gathered buffer (block r at gathered + r * bytes): flat_grads of the two matrices = (1 / world) sum_r dY_r^T X_r / loss
 * scale -- the all-reduce(AVG) of the per-rank gradients to fp32 rounding.  Everything else of the range is exchanged as before. */
int lo_vae_factor_block(const LoVae* h, size_t* byte_offset, size_t* bytes);
int lo_vae_materialize_gathered_linear_grads(LoVae* h, const void* gathered, int world, float* flat_grads, void* stream);
int lo_vae_linear_factored(const LoVae* h);
int lo_vae_materialize_linear_grads(LoVae* h, void* ws, float* flat_grads, void* stream);
/* forward.  eps: explicit N(0,1) noise [B,L] or NULL (on-device counter RNG with `seed`).  target: images for the fused
 * MSE partial sums or NULL.  Outputs recon [B,3,128,128], mu, logvar [B,L] (fp32). */
int lo_vae_forward(LoVae* h, const float* x, const float* eps, uint64_t seed, const float* flat_params, void* ws,
                   float* recon, float* mu, float* logvar, const float* target, void* stream);
/* decoder only, without skip connections: LunarisCoreVAE.sample (lunar_generate.py:278-291).  z fp32 [B,L]. */
int lo_vae_decode(LoVae* h, const float* z, const float* flat_params, void* ws, float* recon, void* stream);
/* The two halves of that forward as the reference's sub-modules expose them (lunar_generate.py:273-275 calls them in turn;
 * callers may too).  Encoder.forward (:127-153): mu, logvar [B,L] and the skip list [B,64,64,64], [B,128,32,32], [B,256,16,16]
 * (fp32 NCHW; NULL = not wanted).  Decoder.forward(z, skips) (:194-229): n_skips = len(skips) in 0..3 with the reference's
 * guards (skips[2] is added after up1 when len >= 3, skips[1] after up2 when len >= 2, skips[0] after up3 when len >= 1). */
int lo_vae_encode(LoVae* h, const float* x, const float* flat_params, void* ws, float* mu, float* logvar, float* skip0, float* skip1,
                  float* skip2, void* stream);
int lo_vae_decode_skips(LoVae* h, const float* z, int n_skips, const float* skip0, const float* skip1, const float* skip2,
                        const float* flat_params, void* ws, float* recon, void* stream);
/* Their backward passes (autograd of the same lines).  lo_vae_decoder_backward follows lo_vae_decode_skips (or lo_vae_forward):
 * drecon [B,3,128,128] -> dz [B,L], the gradients of the skip maps (fp32 NCHW, NULL = not wanted) and the decoder's parameter
 * gradients (decoder.fc.weight to the end of flat_grads; nothing before it is written except the alignment gaps).
 * lo_vae_encoder_backward follows lo_vae_encode (or lo_vae_forward): upstream gradients of mu / logvar [B,L] and of the skip maps
 * (NULL = zero) -> the encoder's parameter gradients ([0, decoder.fc.weight) of flat_grads).  All gradients leave un-scaled. */
int lo_vae_decoder_backward(LoVae* h, const float* flat_params, void* ws, const float* recon, const float* drecon, float loss_scale,
                            float* dz, float* dskip0, float* dskip1, float* dskip2, float* flat_grads, void* stream);
int lo_vae_encoder_backward(LoVae* h, const float* x, const float* flat_params, void* ws, const float* gmu, const float* glv,
                            const float* gskip0, const float* gskip1, const float* gskip2, float loss_scale, float* flat_grads,
                            void* stream);
/* reduce the loss partial sums of the last forward; losses_dev[4] = recon_loss, kl_loss, vae_loss, pg_loss.
 * vae_loss = (recon_weight*recon + kl_weight*kl - mean_advantage*recon)/accum  (train_hybrid.py:886-889,895).
 * adv_dev (device scalar) overrides mean_advantage when not NULL.  Also prepares the gradient seeds for
 * lo_vae_backward(fused=1). */
int lo_vae_loss(LoVae* h, void* ws, float recon_weight, float kl_weight, float mean_advantage, const float* adv_dev,
                float accum, float loss_scale, float* losses_dev, void* stream);
/* backward into flat_grads (same layout as the parameters; every element is written).  fused=1: gradients of the
 * vae_loss prepared by lo_vae_loss (needs `target`).  fused=0: explicit upstream gradients drecon/gmu/glv (fp32, may
 * be NULL = zero). */
int lo_vae_backward(LoVae* h, const float* x, const float* flat_params, void* ws, const float* recon, const float* target,
                    int fused, const float* drecon, const float* gmu, const float* glv, float loss_scale,
                    float* flat_grads, void* stream);

/* GroupNorm + Mish run inside the producing convolution's epilogue wherever the kernel that owns the layer supports it (the
 * workgroups holding one sample's tiles exchange their partial sums through the workspace and wait for each other; csrc/lo_common.h,
 * LoGnFuse): fused_layers = how many of the 16 conv + GroupNorm layers of this plan do so.  The wait is bounded: byte_offset names a
 * 32-bit word of the workspace that a workgroup sets to 1 if its wait ran out (results of that step are then invalid); the host
 * reads it with the step's metrics.  No reference counterpart (aten::native_group_norm is a separate op there). */
int lo_vae_sync_fail_word(const LoVae* h, size_t* byte_offset, int* fused_layers);

/* The nn.Module boundary under torch.amp.GradScaler (train_hybrid.py:246-247, 289-297, 899-923: scaler.scale(loss).backward(),
 * unscale_, clip_grad_norm_, step).  The upstream gradients an autograd node receives are already multiplied by the caller's loss
 * scale; the backward carries fp16 activation gradients and needs them in a fixed range.  lo_grad_scale_pick: max |g| over up to five
 * fp32 tensors (NULL = absent) -> scratch[0] = r = 2^k with max|g| * r in [2, 4), scratch[1] = 1 / r, scratch[2] = max |g|
 * (scratch: >= 260 floats, device memory; no host synchronisation).  lo_scale_copy_dev: dst = src * scale_dev[0].
 * lo_grad_unscale_dev: x *= scale_dev[0] in place; fail_word (may be NULL): the plan's rendezvous-failure word -- when it is set
 * the result is NaN instead, which is how a lost launch reaches a foreign training loop (GradScaler.unscale_ /
 * clip_grad_norm_ see it; there is no optimizer kernel of this library behind the autograd path that could skip the update). */
int lo_grad_scale_pick(const float* g0, size_t n0, const float* g1, size_t n1, const float* g2, size_t n2, const float* g3, size_t n3,
                       const float* g4, size_t n4, float* scratch, void* stream);
int lo_scale_copy_dev(const float* src, float* dst, size_t n, const float* scale_dev, void* stream);
int lo_grad_unscale_dev(float* x, size_t n, const float* scale_dev, const void* fail_word, void* stream);
/* fp8 operand mode (LO_VAE_FP8_FWD, train_hybrid.py --mfma_precision fp8; BASELINE.json configs[4]): how many of the plan's 16
 * forward conv layers actually run on e4m3 operands: those with Cin % 128 == 0 and Cout % 64 == 0, except the 128 -> 64 transposed
 * conv, which stays on its patch-resident fp16 kernel (10 at batch 64; LO_F8_FORCE=0 also leaves the shapes of the fused-tap fp16
 * kernel in fp16: 6).  0 when the mode is off. */
int lo_vae_fp8_layers(const LoVae* h, int* layers);
/* where lo_vae_forward left an intermediate tensor inside the workspace (fp16 NHWC; dims4 = B, H, W, C), for parity tests
 * against the reference's hooked module outputs (lunar_generate.py:94-120, 168-190): which 0 = raw encoder conv output
 * (stage s, k = 0 strided conv, 1 / 2 ResBlock convs), 1 = raw decoder transposed-conv output, 2 = encoder stage (ResBlock)
 * output, 3 = decoder layer activation. */
int lo_vae_debug_tensor(const LoVae* h, int which, int s, int k, size_t* byte_offset, int* dims4);

/* ---- LunarMoETeacher.forward as executed (lunar_evaluator.py:408-462; feature_dim 128) ---------------------------- */
typedef struct LoTeacher LoTeacher;
int lo_teacher_create(int batch, int num_experts, int feature_dim, int embedding_dim, LoTeacher** out);
/* flags: LO_TEACHER_FP8_CONV = the 24 full-resolution 3x3 convolutions of the dropout path (train mode, dropout_p > 0) take OCP
 * e4m3 operands (v_mfma_scale_f32_16x16x128_f8f6f4; activations e4m3(8 x), one weight scale per output channel), fp16 outputs,
 * fp32 BatchNorm statistics; everything else unchanged (BASELINE config 5).  The default path and eval mode stay fp16. */
#define LO_TEACHER_FP8_CONV 1u
int lo_teacher_create_ex(int batch, int num_experts, int feature_dim, int embedding_dim, unsigned flags, LoTeacher** out);
void lo_teacher_destroy(LoTeacher* h);
/* state table in the reference's state_dict order (351 entries at the defaults); float tensors (parameters, BatchNorm
 * running statistics) live in ONE flat fp32 buffer at these element offsets; offset -1 = integer buffer kept by the host */
int lo_teacher_num_tensors(const LoTeacher* h);
const char* lo_teacher_tensor_name(const LoTeacher* h, int index);
size_t lo_teacher_tensor_numel(const LoTeacher* h, int index);
long long lo_teacher_tensor_offset(const LoTeacher* h, int index);
size_t lo_teacher_flat_elems(const LoTeacher* h);
size_t lo_teacher_workspace_bytes(const LoTeacher* h);
int lo_teacher_pack(LoTeacher* h, const float* flat_state, void* ws, void* stream);
/* training != 0: BatchNorm uses batch statistics and updates running_mean / running_var inside flat_state  Passing NULL for all five output
 * pointers makes it a statistics-only call (BatchNorm running statistics updated, pooling of the last block and heads
 * skipped): the first teacher call of _process_batch, train_hybrid.py:853-855, whose outputs are dead in the reference. */
/* dropout_p, drop_seed: the reference's six dropout sites (nn.Dropout after the branch concat lunar_evaluator.py:97-99,108;
 * attn_drop / proj_drop :139-140,212,225; nn.Dropout2d after both ExpertBlock convs :246,253; nn.Dropout in the gate and
 * every head :353-397) with torch semantics (elementwise / per (sample, channel), kept values scaled by 1/(1-p)), active
 * only when training != 0 and dropout_p > 0.  The masks come from a counter RNG keyed by (drop_seed, site, element): pass a
 * fresh drop_seed per call.  With dropout the constant-field shortcuts of the default path do not hold (proj_drop makes the
 * conv2 input a random field) and every 3x3 convolution runs in full; lo_teacher_last_path reports which path the last call
 * took: 0 sparse shortcuts, 1 dense (LO_T_DENSE=1), 2 dropout. */
int lo_teacher_forward(LoTeacher* h, const float* x, float* flat_state, void* ws, int training, float dropout_p,
                       uint64_t drop_seed, float* quality_scores, float* expert_weights, float* style_embedding,
                       float* prompt_embedding, float* semantic_score, void* stream);
int lo_teacher_last_path(const LoTeacher* h);
/* keep[i] (1 / 0) of element i < n of dropout site `site` for call seed drop_seed: the decisions lo_teacher_forward applies
 * (nn.Dropout semantics: lunar_evaluator.py:99,139-140; site numbering in csrc/lo_common.h, oracle/dropout_ref.py). */
int lo_dropout_mask(uint64_t drop_seed, int site, float dropout_p, size_t n, uint8_t* keep, void* stream);

/* gradients of teacher_loss = -(quality_weight/accum) * mean(quality_scores) for the parameters that receive gradients in
 * the reference step (gate.*, quality_heads.*; train_hybrid.py:891-904 with the reentrant-checkpoint quirk):
 * one contiguous range [begin,end) of the flat state layout.  rows: B*(end-begin) floats of scratch.  Replays the dropout
 * masks of the gate / quality-head hidden layers of the lo_teacher_forward call it follows. */
int lo_teacher_grad_range(const LoTeacher* h, size_t* begin_elem, size_t* end_elem);
int lo_teacher_heads_backward(LoTeacher* h, const float* flat_state, void* ws, const float* expert_weights, float coef,
                              float* rows, float* flat_grads, void* stream);
/* The same backward for arbitrary upstream gradients, i.e. what autograd needs to make quality_scores / expert_weights
 * differentiable outputs of LunarMoETeacher.forward (the reference calls teacher_loss.backward() on them,
 * train_hybrid.py:891-904; lunar_evaluator.py:417,431-432,455-462): d_quality [B][4] / d_weights [B][E] (either may be NULL =
 * zero).  The head inputs of the forward being differentiated are passed explicitly — lo_teacher_heads_saved says where
 * lo_teacher_forward left them inside the workspace (byte offsets and element counts of pooled extractor features [B][128],
 * pooled expert features [E][B][F], pre-weighting quality logits [B][E][4]), so a caller may copy them and run this backward
 * after later forward calls — together with that call's dropout_p (0 in eval mode) and drop_seed. */
int lo_teacher_heads_saved(const LoTeacher* h, size_t* byte_offsets3, size_t* elems3);
int lo_teacher_heads_backward_ex(LoTeacher* h, const float* flat_state, const float* pooled_f, const float* pooled_e,
                                 const float* raw_q, const float* expert_weights, const float* d_quality, const float* d_weights,
                                 float dropout_p, uint64_t drop_seed, float* rows, float* flat_grads, void* stream);
/* SURVEY §8 row F2, second half: the teacher trained "as documented" -- gradients for EVERY parameter on the path of the teacher
 * loss, experts and feature extractor included, i.e. what teacher_loss.backward() (train_hybrid.py:891-904) produces when the three
 * torch.utils.checkpoint.checkpoint calls of the model (lunar_evaluator.py:194-197, 266-275, 411-414) are non-reentrant.  Like the
 * reference under checkpointing, the trunk is recomputed block by block (BatchNorm with the batch statistics, running statistics
 * untouched) and then differentiated: 24 3x3 convs + BatchNorm(train) + Dropout2d, the chunk attention as executed (543 live rows,
 * attn_drop / proj_drop replayed from the call's seed), layer_scale, the shortcut conv of feature_dim != 128, the feature extractor's
 * depthwise / pointwise branches.  Must follow lo_teacher_forward(training = 1) on the same images (the heads' inputs and the
 * dropout stream of that call are used); coef = quality_weight / accum; gscale = power of two applied to the fp16 activation
 * gradients (parameter gradients come out unscaled; 64 * B * 16384 is a good value); bws = lo_teacher_full_backward_bytes(h) bytes
 * of scratch; rows as for lo_teacher_heads_backward; flat_grads = the whole state-table layout (lo_teacher_flat_elems floats):
 * every parameter on the loss path is written, everything else (the three heads the loss does not read, the softmax-invariant
 * relative-position tables, BatchNorm buffers) is zero.  After a plain lo_teacher_forward the trunk is recomputed first (one extra
 * forward: the reference's own cost under checkpointing); after lo_teacher_forward_keep on the same bws nothing is. */
size_t lo_teacher_full_backward_bytes(const LoTeacher* h);
/* LunarMoETeacher.forward in train mode (outputs and side effects of lo_teacher_forward(training = 1)) in plain form, with every tensor
 * the backward reads -- feature extractor and all 12 ExpertBlocks -- left inside bws: the forward to use in a step that ends in
 * lo_teacher_full_backward(..., the same bws), which then recomputes nothing (26 GB of kept tensors at batch 64 / feature_dim 128; above
 * 160 GB the plan keeps the block outputs only and the backward recomputes one block at a time). */
int lo_teacher_forward_keep(LoTeacher* h, const float* images_nchw, float* flat_state, void* ws, void* bws, float dropout_p,
                            uint64_t drop_seed, float* quality_scores, float* expert_weights, float* style_embedding,
                            float* prompt_embedding, float* semantic_score, void* stream);
int lo_teacher_full_backward(LoTeacher* h, const float* images_nchw, float* flat_state, void* ws, void* bws, const float* expert_weights,
                             float coef, float gscale, float* rows, float* flat_grads, void* stream);
/* The same backward for arbitrary upstream gradients d_quality [B][4] / d_weights [B][E] (either may be NULL) with the head inputs
 * (lo_teacher_heads_saved), dropout_p and call seed of the forward being differentiated passed explicitly: what the module's autograd
 * node calls (LunarMoETeacher(full_backward=True), so that teacher_loss.backward() of a foreign training loop fills every .grad).
 * Upstream gradients should be of order 1: a foreign loss scale is divided out first (lo_grad_scale_pick / lo_scale_copy_dev). */
int lo_teacher_full_backward_ex(LoTeacher* h, const float* images_nchw, float* flat_state, void* ws, void* bws, const float* pooled_f,
                                const float* pooled_e, const float* raw_q, const float* expert_weights, const float* d_quality,
                                const float* d_weights, float dropout_p, uint64_t drop_seed, float gscale, float* rows, float* flat_grads,
                                void* stream);
/* clip_grad_norm_(teacher.parameters()) + AdamW (train_hybrid.py:914, 922) with every teacher parameter live: the norm over the whole
 * flat gradient of lo_teacher_full_backward, the update over exactly the tensors that have a .grad in the reference (not the
 * BatchNorm buffers, not the style / prompt / semantic heads, whose .grad is None there).  m, v: lo_teacher_flat_elems floats each,
 * scratch: 1028 floats ([1024..1027] = norm, clip coefficient, finite flag, skipped count as for lo_clip_adamw_step).  The fp16 operand
 * copies are stale afterwards: lo_teacher_pack before the next forward. */
int lo_teacher_clip_adamw_full(LoTeacher* h, float* flat_state, const float* flat_grads, float* m, float* v, float max_norm, float lr,
                               float beta1, float beta2, float eps, float weight_decay, int step, float* scratch, void* stream);
int lo_teacher_full_param_count(const LoTeacher* h, size_t* tensors, size_t* elems);
/* reward / baseline / advantage bookkeeping of _process_batch (train_hybrid.py:870-892) on the device; state2 =
 * {baseline, initialised}; out7 = quality_loss, semantic_reward, quality_reward, baseline, advantage, teacher_loss,
 * mean(quality_scores); adv_dev = mean advantage (input of lo_vae_loss). */
int lo_hybrid_reward(const float* quality_scores, const float* semantic_score, int B, float semantic_weight,
                     float reward_scale, float momentum, float quality_weight, float accum, float* state2, float* out7,
                     float* adv_dev, void* stream);

/* The same backward in two calls, for data-parallel overlap: phase 1 = final conv, decoder, decoder.fc, latent, encoder
 * heads (afterwards the gradients of the three Linear layers — 82 % of the bytes, one contiguous range of the flat
 * buffer, see lo_vae_linear_grad_range — are final and can be all-reduced while phase 2 runs); phase 2 = encoder. */
int lo_vae_backward_phase(LoVae* h, int phase, const float* x, const float* flat_params, void* ws, const float* recon,
                          const float* target, int fused, const float* drecon, const float* gmu, const float* glv,
                          float loss_scale, float* flat_grads, void* stream);
int lo_vae_linear_grad_range(const LoVae* h, size_t* begin_elem, size_t* end_elem);
/* [begin, end) of the flat gradient buffer that is complete after lo_vae_backward_phase(.., phase = 1): fc_mu.weight up to
 * the end of the buffer (Linear layers, decoder convs, final conv).  The rest (encoder convs) is complete after phase 2. */
int lo_vae_phase1_grad_range(const LoVae* h, size_t* begin_elem, size_t* end_elem);
/* Three-call form for the data-parallel exchange: phase 1, then phase 3 (encoder stage 4 = down4, 94 % of the encoder's
 * gradient bytes; its range below is complete afterwards), then phase 4 (stages 3..1; [0, stage-4 begin) complete).  Phase 2
 * = 3 + 4 in one call. */
int lo_vae_stage4_grad_range(const LoVae* h, size_t* begin_elem, size_t* end_elem);

/* ---- data-parallel gradient exchange helpers (lunaris_orion_amd/parallel.py; the reference has no distributed code) ---------
 * One streaming pass each around the RCCL calls of the direct (all-to-all reduce-scatter + all-gather) exchange:
 * wire[i] = fp16(g[i] * scale);  share[i] = mean over the `world` received chunks recv[r * chunk + i] (fp32 accumulation, ranks in
 * order; fp16 or fp32 elements);  g[i] = float(wire[i]) * inv_scale. */
int lo_dp_pack_f16(const float* g, void* wire, size_t n, float scale, void* stream);
int lo_dp_unpack_f16(const void* wire, float* g, size_t n, float inv_scale, void* stream);
/* ... and, in the same pass, the sum of squares of the unpacked range into scratch[512..1024) (lo_gradnorm_early_range's result) */
int lo_dp_unpack_f16_sumsq(const void* wire, float* g, size_t n, float inv_scale, float* scratch, void* stream);
int lo_dp_sum_shares(const void* recv, void* share, int world, size_t chunk, int is_f16, void* stream);

#ifdef __cplusplus
}
#endif
#endif
