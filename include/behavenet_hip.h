/*
 * behavenet_hip.h -- C ABI of libbehavenet_hip.so: the MI355X (gfx950) kernels underneath the
 * BehaveNet conv-autoencoder hot path.
 *
 * The reference (themattinthehatt/behavenet) is pure Python and has no FFI of its own: its hot
 * path bottoms out in third-party ATen operators called from `torch.nn` modules.  Each entry
 * point below replaces one such operator call site (cited as reference file:line) with a
 * hand-written HIP kernel.  Everything above this boundary is Python glue.
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless stated otherwise;
 *   - tensors are fp32, contiguous, NCHW (images) or row-major (matrices); the fast convolution
 *     kernels need 16-byte aligned base pointers (any allocator gives that; a view at an odd
 *     offset is served by the shape-agnostic kernels instead);
 *   - the caller owns every buffer; the library never allocates, frees or synchronises;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the null stream);
 *   - return value: 0 on success, a positive hipError_t if a launch failed, a negative
 *     BN_E_* code for an argument/shape error.  No entry point throws or exits.
 *   - "accumulate != 0" means the gradient outputs are added to (+=) instead of overwritten,
 *     which is how the reference accumulates over its 200-frame chunks (aes.py:751-771).
 *
 * Deliberately NOT in this ABI: collectives.  SURVEY.md section 8(b) sketched
 * bn_comm_init / bn_allreduce_grads / bn_comm_destroy; they do not exist.  The gradient
 * exchange of the data-parallel path (one flat gradient arena, bucketed, overlapped with the
 * backward pass) is issued from Python through torch.distributed (backend "nccl" = RCCL over
 * xGMI; behavenet_amd/fitting/distributed.py), which already owns the communicator, its
 * streams and the rendezvous.  A second RCCL communicator behind this library would duplicate
 * that state for no kernel of this path: the library only ever sees device pointers, and the
 * all-reduce operates in place on the same arena the backward kernels accumulate into.
 */
#ifndef BEHAVENET_HIP_H
#define BEHAVENET_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BN_E_BADARG   (-1)  /* null pointer / non-positive size */
#define BN_E_SHAPE    (-2)  /* geometry not representable (e.g. kernel larger than supported) */
#define BN_E_WORKSPACE (-3) /* caller-provided workspace too small */

/* activation codes for fused epilogues */
#define BN_ACT_NONE    0
#define BN_ACT_LRELU   1    /* LeakyReLU(slope)  -- aes.py:114,341 */
#define BN_ACT_SIGMOID 2    /* Sigmoid           -- aes.py:330 */

typedef void* bn_stream_t;

/* library / build info */
int         bn_version(void);                 /* ABI version, currently 1 */
const char* bn_build_arch(void);              /* "gfx950" */
const char* bn_error_string(int code);        /* static string for a BN_E_* / hip code */

/* ------------------------------------------------------------------------------------------
 * Scratch memory.  Some kernels split their reduction dimension over workgroups and combine
 * the partial results in a second, fixed-order pass (deterministic; no atomics).  The caller
 * owns the scratch: ask how much a call needs (0 is common) and pass at least that much.
 * `op` is one of BN_OP_*; the twelve ints are the same geometry arguments, in the same order,
 * as the entry point's.
 * ------------------------------------------------------------------------------------------ */
#define BN_OP_CONV_FWD    1
#define BN_OP_CONV_BWD_D  2
#define BN_OP_CONV_BWD_W  3
#define BN_OP_CONVT_FWD   4
#define BN_OP_CONVT_BWD_D 5
#define BN_OP_CONVT_BWD_W 6
/* Test hook: on != 0 routes every convolution through the shape-agnostic kernels (so the
 * specialised ones can be cross-checked on the device).  Returns the previous setting.  The
 * environment variable BN_FORCE_GENERIC=1 sets the initial value. */
int bn_set_force_generic(int on);
/* Test hook: stride-1 layers with kernels larger than 5x5 run on four shifted copies of their big map, frames in
 * blocks whose copies stay below `bytes` (default and maximum 0x70000000: the kernels' 32-bit offsets); a small value
 * makes a few frames exercise the block loop.  0 restores the default.  Returns the previous setting. */
size_t bn_set_bigk1_block_bytes(size_t bytes);
size_t bn_conv_ws_bytes(int op, int N, int C, int H, int W, int K, int R, int S, int stride,
                        int off_t, int off_l, int P, int Q);
/* Kernels smaller than 5x5 (the reference's architecture search draws 3x3; configs/ae_jsons/ae_arch_2.json has 4x4:
 * /root/reference/behavenet/models/ae_model_architecture_generator.py:90-100) run on the 5x5 kernel families with
 * their taps embedded in 5x5 ones.  Every forward / data-gradient entry point makes that copy itself; a caller that
 * runs a whole stack of layers (and both roles of each) can make the copies of all layers in ONE launch instead:
 *   bn_conv_taps_bytes  bytes of the 5x5 copy if `op` (BN_OP_CONV_FWD / _BWD_D, BN_OP_CONVT_FWD / _BWD_D) pads this
 *                       geometry's taps, 0 if it does not (the copy is the same for the two ops of a layer)
 *   bn_conv_taps_pad    w5[j] <- 5x5 copy of w[j], j < n, one launch; geoms = n x 13 ints (op + the twelve geometry
 *                       arguments); BN_E_SHAPE if an op does not pad
 *   bn_conv_taps_hint   one-shot: the NEXT conv entry point called by this thread reads the 5x5 copy of `w` from `w5`
 *                       if it is called on weights `w` (and pads for itself otherwise); every conv entry point
 *                       clears the hint on return.  w5 = NULL clears it. */
size_t bn_conv_taps_bytes(int op, int N, int C, int H, int W, int K, int R, int S, int stride,
                          int off_t, int off_l, int P, int Q);
int bn_conv_taps_pad(int n, const float* const* w, float* const* w5, const int* geoms, bn_stream_t stream);
int bn_conv_taps_hint(const float* w, const float* w5);

/* ------------------------------------------------------------------------------------------
 * Convolution (replaces ZeroPad2d + nn.Conv2d + LeakyReLU, aes.py:81-86,113-114,145-155).
 *   y[n,k,p,q] = act( b[k] + sum_{c,r,s} x[n,c,p*stride+r-pad_t,q*stride+s-pad_l] * w[k,c,r,s] )
 * reads outside [0,H)x[0,W) are zero, so TF-"same" asymmetric padding needs no padded copy.
 * x:(N,C,H,W) w:(K,C,R,S) b:(K) or NULL  y:(N,K,P,Q)
 * ------------------------------------------------------------------------------------------ */
int bn_conv2d_fwd(const float* x, const float* w, const float* b, float* y,
                  int N, int C, int H, int W, int K, int R, int S, int stride,
                  int pad_t, int pad_l, int P, int Q,
                  int act, float slope, void* ws, size_t ws_bytes, bn_stream_t stream);

/* dx[n,c,h,w] = act'(dact_src[n,c,h,w]) * sum_{k,r,s} dy[n,k,p,q] * w[k,c,r,s],
 * p*stride+r-pad_t == h.  `dy` is the gradient w.r.t. the PRE-activation of this layer.
 * `dact_src` (nullable) is the saved post-activation input of this layer (= output of the layer
 * below); when given, the derivative of the lower layer's activation `dact` is applied in the
 * epilogue so dx is the lower layer's pre-activation gradient (autograd of aes.py:203-211). */
int bn_conv2d_bwd_data(const float* dy, const float* w, float* dx, const float* dact_src,
                       int N, int C, int H, int W, int K, int R, int S, int stride,
                       int pad_t, int pad_l, int P, int Q,
                       int dact, float slope, void* ws, size_t ws_bytes, bn_stream_t stream);

/* dw[k,c,r,s] (+)= sum_{n,p,q} dy[n,k,p,q] * x[n,c,p*stride+r-pad_t,q*stride+s-pad_l]
 * db[k]       (+)= sum_{n,p,q} dy[n,k,p,q]                      (db nullable) */
int bn_conv2d_bwd_weight(const float* x, const float* dy, float* dw, float* db,
                         int N, int C, int H, int W, int K, int R, int S, int stride,
                         int pad_t, int pad_l, int P, int Q,
                         int accumulate, void* ws, size_t ws_bytes, bn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Transposed convolution (replaces nn.ConvTranspose2d + crop F.pad(x,[-l,-r,-t,-b]) +
 * LeakyReLU/Sigmoid, aes.py:315-321,326-341,466-470).  The crop (or torch `padding`) is folded
 * into the output index range; the un-cropped tensor is never materialised:
 *   y[n,co,h,w] = act( b[co] + sum_{ci,r,s} x[n,ci,p,q] * w[ci,co,r,s] ),
 *   p*stride + r == h + crop_t,  q*stride + s == w + crop_l,  0<=h<Ho, 0<=w<Wo.
 * x:(N,Ci,Hi,Wi) w:(Ci,Co,R,S) b:(Co) or NULL  y:(N,Co,Ho,Wo)
 * ------------------------------------------------------------------------------------------ */
int bn_convT2d_fwd(const float* x, const float* w, const float* b, float* y,
                   int N, int Ci, int Hi, int Wi, int Co, int R, int S, int stride,
                   int crop_t, int crop_l, int Ho, int Wo,
                   int act, float slope, void* ws, size_t ws_bytes, bn_stream_t stream);

/* dx[n,ci,p,q] = act'(dact_src[n,ci,p,q]) * sum_{co,r,s} dy[n,co,p*stride+r-crop_t,...] * w[ci,co,r,s] */
int bn_convT2d_bwd_data(const float* dy, const float* w, float* dx, const float* dact_src,
                        int N, int Ci, int Hi, int Wi, int Co, int R, int S, int stride,
                        int crop_t, int crop_l, int Ho, int Wo,
                        int dact, float slope, void* ws, size_t ws_bytes, bn_stream_t stream);

/* dw[ci,co,r,s] (+)= sum_{n,p,q} x[n,ci,p,q] * dy[n,co,p*stride+r-crop_t,q*stride+s-crop_l]
 * db[co]        (+)= sum_{n,h,w} dy[n,co,h,w] */
int bn_convT2d_bwd_weight(const float* x, const float* dy, float* dw, float* db,
                          int N, int Ci, int Hi, int Wi, int Co, int R, int S, int stride,
                          int crop_t, int crop_l, int Ho, int Wo,
                          int accumulate, void* ws, size_t ws_bytes, bn_stream_t stream);

/* dpre[i] = dy[i] * act'(y[i]) where y is the saved POST-activation output
 * (autograd of LeakyReLU / Sigmoid at the top of a conv stack).  In-place (dpre == dy) allowed. */
/* y = act(x)  (the Sigmoid after the optional dense last decoder layer, aes.py:345-359) */
int bn_act_fwd(const float* x, float* y, size_t n, int act, float slope, bn_stream_t stream);
int bn_act_bwd(const float* dy, const float* y, float* dpre, size_t n,
               int act, float slope, bn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * nn.MaxPool2d(return_indices=True, ceil_mode) / nn.MaxUnpool2d of 'max_pooling' architectures
 * (aes.py:99-110,196-208,281-294,460-464).  planes = N*C; x: planes x (H,W); y: planes x (Ho,Wo);
 * idx: int32 position h*W+w of each maximum inside its (H,W) plane (torch's convention; ties go
 * to the first maximum in row-major window order).  The unpool pair works on flat planes:
 * in_plane = pooled H*W, out_plane = H*W of the tensor that was pooled.
 * ------------------------------------------------------------------------------------------ */
int bn_maxpool2d_fwd(const float* x, float* y, int* idx, int planes, int H, int W, int Ho, int Wo,
                     int k, int stride, int pad_t, int pad_l, bn_stream_t stream);
int bn_maxpool2d_bwd(const float* dy, const int* idx, float* dx, int planes, int H, int W,
                     int Ho, int Wo, int k, int stride, int pad_t, int pad_l, bn_stream_t stream);
/* Conv2d + the 2x2 / stride-2 max pooling + the activation behind it in ONE kernel (the reference's conv -> pool ->
 * LeakyReLU order, aes.py:200-211): y:(N,K,P/2,Q/2), idx:(N,K,P/2,Q/2) int32 = h * Q + w of each window's winner in
 * the (P, Q) plane of the convolution's output (torch's MaxPool2d(return_indices) convention, first maximum in row-major
 * window order) -- that output itself is never written.  Served for the stride-1 5x5 layers of a max-pooling architecture
 * on even maps: from 1 or 2 input channels onto a multiple of 16, and from a multiple of 4 onto a multiple of 32 where the
 * matrix-core kernel's tile is whole even row blocks (bn_conv2d_pool2_act_ok); BN_E_SHAPE otherwise
 * (the caller then runs bn_conv2d_fwd and bn_maxpool2d_act_fwd).  Backward: bn_maxpool2d_act_bwd on (y, idx), then the
 * layer's own weight gradient. */
int bn_conv2d_pool2_act_fwd(const float* x, const float* w, const float* b, float* y, int* idx,
                            int N, int C, int H, int W, int K, int R, int S, int stride,
                            int pad_t, int pad_l, int P, int Q, int act, float slope, bn_stream_t stream);
/* 1 if bn_conv2d_pool2_act_fwd serves this geometry (16-byte aligned operands assumed), else 0 */
int bn_conv2d_pool2_act_ok(int N, int C, int H, int W, int K, int R, int S, int stride,
                           int pad_t, int pad_l, int P, int Q);
/* Weight (+ bias, db nullable) gradient of such a layer straight from the POOLED side: dy, y (the saved output of
 * bn_conv2d_pool2_act_fwd) and idx are (N,K,P/2,Q/2); dw (+)= sum_n,windows dy act'(y) x[winner + tap], db (+)= sum dy act'(y).
 * The dense gradient of the convolution's output is never built.  Served for 1 or 2 input channels and 16 / 32 / 64 output
 * channels (the first layer); the scratch query returns 0 where it is not, the call BN_E_SHAPE. */
size_t bn_conv2d_pool2_bwd_weight_ws_bytes(int N, int C, int H, int W, int K, int R, int S, int stride,
                                           int pad_t, int pad_l, int P, int Q);
int bn_conv2d_pool2_bwd_weight(const float* x, const float* dy, const float* y, const int* idx, float* dw, float* db,
                               int N, int C, int H, int W, int K, int R, int S, int stride,
                               int pad_t, int pad_l, int P, int Q, int act, float slope, int accumulate,
                               void* ws, size_t ws_bytes, bn_stream_t stream);
/* 2x2 / stride-2 / unpadded max pooling of an even map WITH the activation that follows it (aes.py:204-211: conv ->
 * pool -> LeakyReLU), one pass each way: y = act(max), idx as bn_maxpool2d_fwd; dx = spread(dy * act'(y)).
 * BN_E_SHAPE if W / 2 is odd or a pointer is not 16-byte aligned (use bn_maxpool2d_fwd + the activation then). */
int bn_maxpool2d_act_fwd(const float* x, float* y, int* idx, int planes, int H, int W, int act, float slope,
                         bn_stream_t stream);
int bn_maxpool2d_act_bwd(const float* dy, const float* y, const int* idx, float* dx, int planes, int H, int W,
                         int act, float slope, bn_stream_t stream);
int bn_maxunpool2d_fwd(const float* x, const int* idx, float* y, int planes, int in_plane,
                       int out_plane, bn_stream_t stream);
/* The same for the indices of a 2x2 / stride-2 / unpadded pooling of a (2 Hi) x (2 Wi) map (every index lies inside its
 * own window -- the caller vouches for it; what nn.MaxPool2d(2, return_indices=True) hands to the decoder,
 * aes.py:204-207,460-464): one pass without the memset.  BN_E_SHAPE if Wi is odd or a pointer is not 16-byte aligned. */
int bn_maxunpool2d_fwd_k2(const float* x, const int* idx, float* y, int planes, int Hi, int Wi, bn_stream_t stream);
int bn_maxunpool2d_bwd(const float* dy, const int* idx, float* dx, int planes, int in_plane,
                       int out_plane, bn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * BatchNorm2d + activation for `ae_batch_norm = 1` architectures (replaces nn.BatchNorm2d +
 * LeakyReLU, aes.py:90-97,113-114,332-341).  x,y,dy,dx: (N,C,HW) fp32; per-channel vectors: (C).
 *   stats:    mean[c], var[c] (biased) over (N, HW) -- two-pass, deterministic
 *   finalize: invstd = 1/sqrt(var+eps); running = (1-momentum)*running + momentum*{mean,
 *             var*unbias}  (running_* nullable; pass momentum = 1/num_batches_tracked for
 *             nn.BatchNorm2d(momentum=None))
 *   act_fwd:  y = act((x-mean)*invstd*gamma + beta)      (eval mode: pass the running stats)
 *   act_bwd:  dy is the gradient w.r.t. y; dx w.r.t. x; dgamma/dbeta (+)= their sums;
 *             batch_stats=1 when mean/invstd are the batch's own (train mode), 0 when they are
 *             the running statistics (constants w.r.t. x)
 * ------------------------------------------------------------------------------------------ */
size_t bn_batchnorm_ws_bytes(int N, int C);
int bn_batchnorm_stats(const float* x, float* mean, float* var, int N, int C, int HW,
                       void* ws, size_t ws_bytes, bn_stream_t stream);
int bn_batchnorm_finalize(const float* mean, const float* var, float* invstd,
                          float* running_mean, float* running_var, int C, float eps,
                          float momentum, float unbias, bn_stream_t stream);
int bn_batchnorm_act_fwd(const float* x, const float* mean, const float* invstd,
                         const float* gamma, const float* beta, float* y, int N, int C, int HW,
                         int act, float slope, bn_stream_t stream);
int bn_batchnorm_act_bwd(const float* x, const float* y, const float* dy, const float* mean,
                         const float* invstd, const float* gamma, float* dx, float* dgamma,
                         float* dbeta, int accumulate, int batch_stats, int N, int C, int HW,
                         int act, float slope, void* ws, size_t ws_bytes, bn_stream_t stream);

/* The same two passes for a batch whose statistics are per CHUNK of frames (rows [bounds[2i],
 * bounds[2i+1]) of x, in order -- the reference runs its 200-frame chunks one after another through
 * nn.BatchNorm2d, aes.py:748-771 with :90-97): one call per layer.  mean / invstd: [n_chunks][C];
 * factors[i]: the running-estimate factor of chunk i's update (momentum, or 1 / num_batches_tracked);
 * bounds, factors: host arrays.  ws: bn_batchnorm_ws_bytes(largest chunk, C).
 * Round 4: both moments in ONE pass over x (shifted sums around the chunk's first value of the channel),
 * one small launch for mean / invstd / the running estimates / the device counter
 * num_batches_tracked (+= n_chunks; NULL: none, nn.BatchNorm2d's int64 buffer); the backward pass with
 * y == NULL (identity / LeakyReLU) rebuilds the sign of the activation's input from x through
 * (gamma, beta) with the forward pass's own fused multiply-adds instead of reading y back. */
int bn_batchnorm_train_fwd_chunks(const float* x, const float* gamma, const float* beta,
                                  float* running_mean, float* running_var,
                                  long long* num_batches_tracked, float* y, float* mean,
                                  float* invstd, const int* bounds, const float* factors,
                                  int n_chunks, int C, int HW, float eps, int act, float slope,
                                  void* ws, size_t ws_bytes, bn_stream_t stream);
int bn_batchnorm_act_bwd_chunks(const float* x, const float* y, const float* dy, const float* mean,
                                const float* invstd, const float* gamma, const float* beta,
                                float* dx, float* dgamma, float* dbeta, int accumulate,
                                const int* bounds, int n_chunks, int C, int HW, int act, float slope,
                                void* ws, size_t ws_bytes, bn_stream_t stream);

/* Split forms of the two reductions above for statistics taken over ALL ranks' frames (frame-sharded
 * data parallelism; the reference's single-device nn.BatchNorm2d sees the whole chunk, aes.py:90-97):
 * the caller all-reduces the per-channel SUMS between the passes.
 *   bn_batchnorm_moment:     sums[c] = sum x            (center == NULL)
 *                            sums[c] = sum (x-center[c])^2   -- NOT divided by the count
 *   bn_batchnorm_bwd_reduce: sum_dz[c] = sum dy act'(y),  sum_dzx[c] = sum dy act'(y) xhat
 *   bn_batchnorm_bwd_apply:  dx = gamma invstd (dz - sum_dz inv_count - xhat sum_dzx inv_count)
 * ws: bn_batchnorm_ws_bytes(N, C). */
int bn_batchnorm_moment(const float* x, const float* center, float* sums, int N, int C, int HW,
                        void* ws, size_t ws_bytes, bn_stream_t stream);
int bn_batchnorm_bwd_reduce(const float* x, const float* y, const float* dy, const float* mean,
                            const float* invstd, float* sum_dz, float* sum_dzx, int N, int C,
                            int HW, int act, float slope, void* ws, size_t ws_bytes,
                            bn_stream_t stream);
int bn_batchnorm_bwd_apply(const float* x, const float* y, const float* dy, const float* mean,
                           const float* invstd, const float* gamma, const float* sum_dz,
                           const float* sum_dzx, float* dx, int N, int C, int HW, float inv_count,
                           int act, float slope, bn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Dense latent projections (replaces nn.Linear, aes.py:121,125,266 and the PS-VAE heads
 * vaes.py:1288-1302).  MFMA (v_mfma_f32_32x32x2_f32: exact fp32).
 *   y[m,n] = b[n] + sum_k x[m,k] * w[n,k]        x:(M,K) w:(N,K) b:(N) or NULL  y:(M,N)
 *   ws: scratch of bn_linear_ws_bytes(M, K, N) bytes (may be NULL / 0: then a long reduction
 *       feeding few output tiles is not split over workgroups)
 * ------------------------------------------------------------------------------------------ */
size_t bn_linear_ws_bytes(int M, int K, int N);
int bn_linear_fwd(const float* x, const float* w, const float* b, float* y,
                  int M, int K, int N, void* ws, size_t ws_bytes, bn_stream_t stream);
/* dx[m,k] = act'(dact_src[m,k]) * sum_n dy[m,n] w[n,k]   (dx, dact_src nullable)
 * dw[n,k] (+)= sum_m dy[m,n] x[m,k]   db[n] (+)= sum_m dy[m,n]   (dw, db nullable) */
int bn_linear_bwd(const float* x, const float* w, const float* dy,
                  float* dx, const float* dact_src, int dact, float slope,
                  float* dw, float* db, int accumulate,
                  int M, int K, int N, void* ws, size_t ws_bytes, bn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Pixel losses (replaces losses.mse / losses.gaussian_ll, losses.py:56-59,84-96).
 * frame_sums[n] = sum_d (pred[n,d]-target[n,d])^2 * mask[n,d]   (mask nullable), d < D.
 * The caller turns frame sums into mse (sum/(N*D)) or gaussian ll; see fitting/losses.py.
 * ------------------------------------------------------------------------------------------ */
int bn_sqerr_frame_sums(const float* pred, const float* target, const float* mask,
                        float* frame_sums, int N, size_t D, bn_stream_t stream);
/* dpred[i] = (*gscale) * scale * 2 * (pred[i]-target[i]) * mask[i];  gscale: device scalar or NULL */
int bn_sqerr_bwd(const float* pred, const float* target, const float* mask, float* dpred,
                 size_t n, float scale, const float* gscale, bn_stream_t stream);
/* out[0] = sum_i in[i] * scale  (deterministic single-workgroup tree; n small) */
int bn_reduce_sum(const float* in, float* out, size_t n, float scale, bn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Variational tail (replaces vaes.reparameterize + losses.kl_div_to_std_normal,
 * vaes.py:33-35, losses.py:146-147).  NOTE std = exp(logvar), as in the reference.
 *   z = mu + eps*exp(logvar);  kl_rows[n] = 0.5*sum_d(exp(lv) - lv + mu^2 - 1)
 * ------------------------------------------------------------------------------------------ */
int bn_reparam_fwd(const float* mu, const float* logvar, const float* eps, float* z,
                   size_t n, bn_stream_t stream);
int bn_kl_rows(const float* mu, const float* logvar, float* kl_rows, int N, int D,
               bn_stream_t stream);
/* autograd of the two ops above:
 *   dlogvar[i] = dz[i] * (z[i] - mu[i])                       (dmu = dz needs no kernel)
 *   dmu[i] = s*mu[i], dlogvar[i] = s*0.5*(exp(logvar[i])-1),  s = scale * (*gscale) */
int bn_reparam_bwd(const float* dz, const float* z, const float* mu, float* dlogvar, size_t n,
                   bn_stream_t stream);
int bn_kl_bwd(const float* mu, const float* logvar, float* dmu, float* dlogvar, size_t n,
              float scale, const float* gscale, bn_stream_t stream);

/* Decomposed KL of the beta-TC-VAE / PS-VAE (replaces losses.decomposed_kl, losses.py:284-351,
 * and its autograd graph).  z, mu, logvar: (N, D) fp32, D <= 32.
 *   fwd: out3 = (index-code MI, total correlation, dimension-wise KL); log_qz (N) and lse (N, D)
 *        are saved for the backward pass; terms is 3N floats of scratch.
 *   bwd: g3 = the three upstream gradients (device); dz, dmu, dlogvar: (N, D). */
int bn_decomposed_kl_fwd(const float* z, const float* mu, const float* logvar, float* out3,
                         float* log_qz, float* lse, float* terms, int N, int D,
                         bn_stream_t stream);
int bn_decomposed_kl_bwd(const float* z, const float* mu, const float* logvar,
                         const float* log_qz, const float* lse, const float* g3, float* dz,
                         float* dmu, float* dlogvar, int N, int D, bn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Latent head of the PS-VAE (reference vaes.py:571-601 forward, :669-704 per-chunk loss terms):
 * everything between the encoder's two heads / the decomposed-KL kernels and the decoder, fused.
 *   y (N, L) supervised means, w (N, U) unsupervised means, logvar / eps / z (N, L + U), D = the
 *   diagonal label map (Dw, Db of length L; Db nullable), labels / lmask (N, L; lmask nullable).
 *   fwd:     z = [y | w] + eps exp(logvar); z_u, lv_u: the unsupervised columns of z / logvar as
 *            contiguous (N, U) tensors (inputs of bn_decomposed_kl_fwd); yhat = D y;
 *            row_sq[n] = sum_l (yhat - labels)^2 lmask; row_kl[n] = 0.5 sum_{l<L} (e^lv - lv + y^2 - 1)
 *   combine: per chunk c = rows [bounds[2c], bounds[2c+1]) (device ints), dkl3 = (n_chunks, 3) from
 *            bn_decomposed_kl_fwd: cols5[c] = (ll_labels, KL_s, MI, TC, DWKL),
 *            T[c] = -alpha ll_labels + KL_s + kl MI + beta TC + kl DWKL
 *   bwd:     gT[c] = dL/dT[c], dz = dL/dz from the decoder, gz_u / gmu_u / glv_u = the outputs of
 *            bn_decomposed_kl_bwd (called with g3 = gT[c] (kl, beta, kl)) -> dy, dw, dlogvar and the
 *            gradients of D (written, or added to when accumulate) */
int bn_psvae_head_fwd(const float* y, const float* w, const float* logvar, const float* eps,
                      const float* Dw, const float* Db, const float* labels, const float* lmask,
                      float* z, float* z_u, float* lv_u, float* yhat, float* row_sq,
                      float* row_kl, int N, int L, int U, bn_stream_t stream);
int bn_psvae_head_combine(const float* row_sq, const float* row_kl, const float* dkl3,
                          const int* bounds, int n_chunks, float alpha, float kl, float beta,
                          int L, float* T, float* cols5, bn_stream_t stream);
int bn_psvae_head_bwd(const float* dz, const float* gT, const int* bounds, int n_chunks,
                      const float* y, const float* logvar, const float* eps, const float* yhat,
                      const float* labels, const float* lmask, const float* Dw, const float* gz_u,
                      const float* gmu_u, const float* glv_u, float alpha, float* dy, float* dw,
                      float* dlogvar, float* dDw, float* dDb, int accumulate, int N, int L, int U,
                      bn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Last decoder layer fused with the pixel loss (replaces, in ONE pass, the ConvTranspose2d + crop
 * + Sigmoid of aes.py:315-330,466-470 and the squared error of losses.py:56-59 / 84-96 together
 * with their derivatives): for every frame n
 *     xhat   = act(convT(x, w, b))                     written only if xhat != NULL
 *     part[n][j], j < P:  sum_j part[n][j] = sum_{c,h,w} (xhat - target)^2 * mask
 *     dpre   = 2 (xhat - target) mask act'(xhat)       = d(frame sum)/d(pre-activation)
 * so that training never writes or re-reads xhat.  P = bn_convT2d_fwd_sqerr_parts(geometry)
 * (partial sums per frame, summed by the caller in index order: deterministic).  mask nullable.
 * `ws`: bn_convT2d_fwd_sqerr_ws_bytes(geometry, xhat != NULL) bytes (0 for the benchmark layer).
 * bn_scale_frames: t[n, :] *= frame_scale[n] * (group_scale ? group_scale[group_of_frame[n]] : 1)
 * -- the backward pass applies the per-chunk loss normalisation and the upstream gradient of a
 * frame's chunk to dpre with it (aes.py:751-771: one mean per 200-frame chunk).
 * ------------------------------------------------------------------------------------------ */
int bn_convT2d_fwd_sqerr_parts(int N, int Ci, int Hi, int Wi, int Co, int R, int S, int stride,
                               int crop_t, int crop_l, int Ho, int Wo);
size_t bn_convT2d_fwd_sqerr_ws_bytes(int N, int Ci, int Hi, int Wi, int Co, int R, int S,
                                     int stride, int crop_t, int crop_l, int Ho, int Wo,
                                     int with_xhat);
int bn_convT2d_fwd_sqerr(const float* x, const float* w, const float* b, const float* target,
                         const float* mask, float* xhat, float* dpre, float* part,
                         int N, int Ci, int Hi, int Wi, int Co, int R, int S, int stride,
                         int crop_t, int crop_l, int Ho, int Wo, int act, float slope,
                         void* ws, size_t ws_bytes, bn_stream_t stream);
int bn_scale_frames(float* t, const float* frame_scale, const float* group_scale,
                    const int* group_of_frame, int N, size_t D, bn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Optimiser (replaces torch.optim.Adam(amsgrad=True).step, training.py:284-286,352), over one
 * flat fp32 parameter arena.  `step` is 1-based.  weight_decay adds wd*p to the gradient.
 * ------------------------------------------------------------------------------------------ */
int bn_adam_amsgrad_step(float* p, const float* g, float* m, float* v, float* vmax, size_t n,
                         float lr, float beta1, float beta2, float eps, float weight_decay,
                         int step, bn_stream_t stream);

/* First encoder layer straight from the stored uint8 frames: y = act(conv(x_u8 / 255) + b).  The
 * reference converts every batch on the host (data_generator.py:251-263: astype(float32) / 255)
 * and feeds the float copy to nn.Conv2d (aes.py:81-86,153); here the conversion (an IEEE division,
 * bit-identical to numpy's) happens while the input patch is staged, so a frame is read as 1 byte
 * per pixel and no float copy of it is ever written.  Same geometry arguments as bn_conv2d_fwd;
 * `ws`: bn_conv2d_fwd_u8_ws_bytes(..., act) bytes for the SAME geometry and activation (0 for the
 * benchmark layer with BN_ACT_NONE / BN_ACT_LRELU; other geometries / activations convert into it
 * and run the float kernels; BN_E_WORKSPACE if it is missing or too small). */
size_t bn_conv2d_fwd_u8_ws_bytes(int N, int C, int H, int W, int K, int R, int S, int stride,
                                 int pad_t, int pad_l, int P, int Q, int act);
int bn_conv2d_fwd_u8(const unsigned char* x, const float* w, const float* b, float* y,
                     int N, int C, int H, int W, int K, int R, int S, int stride,
                     int pad_t, int pad_l, int P, int Q, int act, float slope,
                     void* ws, size_t ws_bytes, bn_stream_t stream);

/* uint8 frames -> float32/255 (replaces the host-side astype(float32)/255 of
 * data_generator.py:251-263 for device-resident uint8 trials) */
int bn_u8_to_unit_float(const unsigned char* in, float* out, size_t n, bn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * In-library kernel timing used by bench.py's roofline line: when enabled, every call of the
 * selected kernel family is bracketed by hipEvents on its own stream, and the dispatch of its
 * main kernel carries a second pair.
 * ------------------------------------------------------------------------------------------ */
#define BN_PROF_NONE        0
#define BN_PROF_CONV_FWD    1
#define BN_PROF_CONV_BWD_D  2
#define BN_PROF_CONV_BWD_W  3
#define BN_PROF_CONVT_FWD   4
#define BN_PROF_CONVT_BWD_D 5
#define BN_PROF_CONVT_BWD_W 6
#define BN_PROF_ADAM        7
#define BN_PROF_LINEAR_FWD  8   /* nn.Linear forward; C = in features, K = out features */
#define BN_PROF_LINEAR_BWD  9   /* nn.Linear backward: all launches of one bn_linear_bwd call */
/* select family + optional geometry filter (C<=0 / K<=0 = any).  Resets the accumulators. */
int bn_prof_select(int family, int C, int K);
/* the same, but only the nth (0-based) matching call after the selection is timed: tells layers
 * apart that share a family and a channel pair (ae_arch_2.json: four 64 -> 64 layers) */
int bn_prof_select_nth(int family, int C, int K, int nth);
/* host-synchronising: total milliseconds and call count since bn_prof_select, everything a
 * call launched (its combine / finish kernels, the two event records and the gaps included) */
int bn_prof_read(double* total_ms, long* launches);
/* the same calls, the MAIN kernel of each alone: interval of the events attached to its dispatch
 * (launches = 0 when the path that served the call does not attach them) */
int bn_prof_read_main(double* total_ms, long* launches);
/* on == 0: no bracketing event records (only the dispatch pair; nothing extra goes on the stream
 * inside a timed region).  Returns the previous setting; default on. */
int bn_prof_set_bracket(int on);
/* name of the device kernel that served the most recent launch of the selected family */
const char* bn_prof_kernel_name(void);
/* constant part of a dispatch-attached event interval (empty kernel, minimum over iters), us */
double bn_prof_dispatch_overhead_us(int iters, bn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* BEHAVENET_HIP_H */
