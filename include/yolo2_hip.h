/* yolo2_hip.h -- C ABI of the MI355X (gfx950) YOLOv2 hot path.
 *
 * The reference (ruiminshen/yolo-tf) has no FFI layer: its hot path is Python calling
 * TensorFlow-1.0 ops and NumPy.  This header is the boundary a maintainer would bind instead of
 * those ops (ctypes stub in INTEGRATION.md).  Each entry names the reference call site it
 * replaces (file:line into the reference tree).
 *
 * Conventions
 *   - plain pointers are DEVICE pointers unless marked host; the caller owns every buffer
 *     (outputs and workspaces included); the only allocation behind the ABI is the 32 KiB flag pool named below.
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default
 *     stream) and may be issued from several host threads at once.  Library-owned state, all of it: a thread-local
 *     error string; a thread-local record of the last conv / wgrad plan (diagnostics below); one 32 KiB per-device pool
 *     of stream-K hand-off flags, created on first use and handed out under a mutex (yolo2_conv2d_ws & co.; freed by
 *     yolo2_shutdown()); tuning knobs read once from the environment (YOLO2_* A/B switches); and the
 *     process-wide test switch yolo2_debug_set_wgrad_variant.
 *   - return value: 0 = OK, otherwise a YOLO2_E_* code; yolo2_last_error() describes it.
 *   - activations are NHWC with an explicit pixel stride `ld*` (elements between consecutive
 *     pixels, >= channel count, multiple of 8); lanes in [C, ld) are padding the kernels never
 *     write and always read as zero (the caller zero-fills buffers once at allocation).
 *   - dtype selects the storage/compute type of activations and prepared filters:
 *     YOLO2_F32 (parity mode, f32 MFMA, exact fmaf chains) or YOLO2_BF16 (bf16 MFMA, f32
 *     accumulate).  Parameters, statistics, gradients of parameters and optimizer state are
 *     always f32.
 */
#ifndef YOLO2_HIP_H
#define YOLO2_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { YOLO2_F32 = 0, YOLO2_BF16 = 1 };
enum {
    YOLO2_OK = 0,
    YOLO2_E_ARG = 1,      /* bad shape / alignment / enum */
    YOLO2_E_LAUNCH = 2,   /* HIP launch or runtime error */
    YOLO2_E_UNSUPPORTED = 3
};

int yolo2_abi_version(void);
const char *yolo2_last_error(void);
/* CRC32C (Castagnoli) of a HOST buffer, continuing from `crc` (0 to start): TFRecord / TensorBoard event / TF checkpoint files
 * (the reference reads and writes them through TensorFlow: utils/data/cache.py:95-100, train.py:141-145, detect.py:104-106) */
uint32_t yolo2_crc32c(const void *data, size_t n, uint32_t crc);
/* releases the library-owned stream-K flag pools (after synchronising their devices); any later call re-creates them */
int yolo2_shutdown(void);
/* Synchronises `stream` and reports errors only the device can detect: a stream-K convolution whose tile owner gave up waiting for a
 * partner workgroup's partial tile (the wait is bounded at 2 s; conv_shared.h y2_sk_wait_and_clear).  YOLO2_E_LAUNCH then means that
 * convolution outputs produced since the previous check are invalid; the flag pool is reset and the process may continue.  The counterpart
 * of the reference surfacing failures as exceptions instead of hanging (utils/postprocess.py:22-35 asserts, detect.py:70 check_numerics). */
int yolo2_check_async_errors(void *stream);
/* The asynchronous form: enqueues on `stream` a copy of the current device's YOLO2_ASYNC_ERROR_WORDS give-up counters into caller-owned PINNED
 * host memory and returns at once.  When the copy has completed (event / stream query) any non-zero word means yolo2_check_async_errors will
 * report YOLO2_E_LAUNCH: a host polls this every step (no synchronisation, one 32-byte copy) and only pays for the synchronising call when
 * something went wrong.  Same per-device semantics as yolo2_check_async_errors. */
#define YOLO2_ASYNC_ERROR_WORDS 8
int yolo2_async_error_snapshot(unsigned *host_words, void *stream);

/* ---- workspace sizes, in bytes, of every entry that takes caller-owned scratch (`ws`): pure host queries ----------------
 * yolo2_conv2d_workspace_bytes: the largest scratch any variant of yolo2_conv2d_ws / _bn / _bias_leaky can use for the
 * shape (stream-K: one f32 256x128 tile slot per CU = 33.5 MB on MI355X; K-sliced grids: the f32 [B*H*W][Nf] partial image);
 * a smaller buffer only disables variants.  The others are exact minimum sizes. */
size_t yolo2_conv2d_workspace_bytes(int B, int H, int W, int Cp, int Nf, int ksize, int dtype);
size_t yolo2_bn_workspace_bytes(int C);              /* yolo2_bn_stats*, yolo2_bn_leaky(_pool)_bwd_reduce */
size_t yolo2_bias_grad_workspace_bytes(int ld);
size_t yolo2_image_prep_workspace_bytes(int B);
size_t yolo2_loss_workspace_bytes(int B, int cells, int A);
size_t yolo2_nms_workspace_bytes(int B, int N, int C);
size_t yolo2_clip_workspace_bytes(int nseg);
size_t yolo2_augment_workspace_bytes(int B);

/* ---- convolution: slim.layers.conv2d, model/yolo2/inference.py:37-48,73-118 ------------------
 * Implicit-GEMM NHWC convolution, stride 1, SAME zero padding, ksize 1 or 3:
 *   O[b,h,w,n] = bias[n] + sum_{r,s,c<Cp} P[b,h+r-pad,w+s-pad,c] * F[n][(r*ksize+s)*Cp + c]
 * P: [B,H,W,ldp] (dtype), F: [Nf][ksize*ksize*Cp] (dtype, K-contiguous, from yolo2_filter_prep),
 * O: [B,H,W,ldo] (dtype), bias: f32[Nf] or NULL.  Forward uses the OHWI filter; the data
 * gradient is the same call with P = dY and the flipped/transposed filter. */
int yolo2_conv2d(const void *P, const void *F, const float *bias, void *O,
                 int B, int H, int W, int Cp, int ldp, int Nf, int ldo, int ksize,
                 int dtype, void *stream);

/* Same convolution with a caller-owned f32 workspace: when the M x N tile grid cannot fill the chip
 * (13x13 / 26x26 stages at small batch) the K loop is sliced across workgroups, partial tiles are
 * accumulated in `ws` (>= B*H*W*Nf floats for slicing to be considered; smaller = no slicing) and a
 * finishing kernel writes O.  Grids below one tile per CU with a long reduction instead run "stream-K": one
 * workgroup per CU, equal contiguous shares of the flat (tile, K step) space, partial tiles parked in `ws`
 * (needs >= CUs*256*128 floats = 33.6 MB on MI355X for the 256x128 tile, half for 128x128) and fixed up inside the same
 * launch through library-owned hand-off flags.
 * `ws` may hold anything on entry.  Results agree with yolo2_conv2d to f32 rounding of the partial sums. */
int yolo2_conv2d_ws(const void *P, const void *F, const float *bias, void *O, float *ws,
                    size_t ws_bytes, int B, int H, int W, int Cp, int ldp, int Nf, int ldo,
                    int ksize, int dtype, void *stream);

/* Filter gradient of the same convolution (tf.gradients of conv2d, train.py:127-129):
 *   dW[r,s,c,n] += sum_{b,h,w} X[b,h+r-pad,w+s-pad,c] * dY[b,h,w,n]       (HWIO, f32)
 * dW must be zeroed by the caller when yolo2_conv2d_wgrad_accumulates() says so: pixel-range splits accumulate with
 * f32 atomics. */
/* 1 if yolo2_conv2d_wgrad ADDS into dW for this shape (several pixel ranges per tile: the caller must zero dW first),
 * 0 if it overwrites every element (single range: dW may hold anything).  Pure host query, no launch. */
int yolo2_conv2d_wgrad_accumulates(int B, int H, int W, int Cin, int ldx, int Cout, int ldy, int ksize, int dtype);
int yolo2_conv2d_wgrad(const void *X, const void *dY, float *dW,
                       int B, int H, int W, int Cin, int ldx, int Cout, int ldy, int ksize,
                       int dtype, void *stream);

/* HWIO f32 master weights [k,k,Cin,Cout] -> the two K-contiguous operand layouts:
 *   Ffwd [Cout][k*k*ldcin]  : Ffwd[n][koff(r*k+s, c, ldcin)]  = W[r,s,c,n]
 *   Fdgr [Cin ][k*k*ldcout] : Fdgr[c][koff(r*k+s, n, ldcout)] = W[k-1-r,k-1-s,c,n]
 * (zero in the padding lanes).  Either output may be NULL.  K order inside a row, for ld channels and t = k*k taps:
 *   koff(tap, c, ld) = (c / kc) * t * kc + tap * kc + c % kc,   kc = 64 if (t > 1 and ld % 64 == 0) else ld
 * i.e. 64-channel chunk outermost, then tap, then channel: the 9 shifted reads of a pixel-tile x 64-channel slab are
 * consecutive K steps, so they hit in L2 (tap-major order measured 15x the algorithmic fabric traffic in the 13x13
 * stages).  The layout is an implementation detail shared by yolo2_filter_prep* and yolo2_conv2d*. */
int yolo2_filter_prep(const float *W, void *Ffwd, void *Fdgr, int ksize, int Cin, int ldcin,
                      int Cout, int ldcout, int dtype, void *stream);

/* The same for every layer of a network in one launch.  `descs_device` is a DEVICE array of n descriptors
 * sorted by first_block; layer i owns blocks [first_block_i, first_block_{i+1}) and needs
 * yolo2_filter_prep_blocks(ksize, ldcin, ldcout) = ksize^2 * ceil(ldcin / YOLO2_FILTER_PREP_TILE) * ceil(ldcout / YOLO2_FILTER_PREP_TILE_N) of them
 * (one block = one (c, n) tile of one tap); total_blocks = end of the last layer.  ldcin, ldcout must be multiples of 8 and Ffwd/Fdgr 16-byte aligned. */
#define YOLO2_FILTER_PREP_TILE 64       /* input channels per tile */
#define YOLO2_FILTER_PREP_TILE_N 128   /* filters per tile (round 6: 128 -- the master arrays then move in 512-byte runs) */
int yolo2_filter_prep_blocks(int ksize, int ldcin, int ldcout);
typedef struct yolo2_filter_desc {
    const float *W;      /* HWIO f32 master weights */
    void *Ffwd;          /* [Cout][k*k*ldcin]  or NULL */
    void *Fdgr;          /* [Cin ][k*k*ldcout] or NULL */
    int ksize, cin, ldcin, cout, ldcout, first_block;
} yolo2_filter_desc;
int yolo2_filter_prep_batch(const yolo2_filter_desc *descs_device, int n, int total_blocks, int dtype,
                            void *stream);

/* tf.train.AdamOptimizer's ApplyAdam (train.py:73-79,127) over every convolution filter of the descriptor table AND the operand layouts
 * of the updated filters, in one pass: equal, bit for bit, to yolo2_adam on the same elements followed by yolo2_filter_prep_batch.
 * descs[i].W must point into `params`; grads / m / v are arenas with the same element layout.  small_ranges_device: n_small pairs
 * (element offset, count) of the parameters that are not filters (gamma, beta, biases), updated by the same launch.  alpha is the
 * bias-corrected step size (lr * sqrt(1 - beta2^t) / (1 - beta1^t)), gscale multiplies the gradient (1 / world size).  A table may hold ONE
 * layer (first_block 0: a layer updated as soon as its gradient is final, while backward continues on another stream) or none (n = 0,
 * total_blocks = 0: only the small ranges). */
int yolo2_adam_filter_prep(const yolo2_filter_desc *descs_device, int n, int total_blocks, const long *small_ranges_device, int n_small,
                           float *params, const float *grads, float *m, float *v, float alpha, float beta1, float beta2, float eps,
                           float gscale, int dtype, void *stream);

/* Inference form of a batch-normalised layer with the moving statistics folded into the operands
 * (W' = W * gamma/sqrt(var+eps) per output channel, bias' = beta - mean*gamma/sqrt(var+eps)):
 *   O = leaky_relu(conv(P, F') + bias', alpha)      -- conv + batch_norm(is_training=False) + leaky_relu of
 * model/yolo2/inference.py:62-66,73-112 in one launch; as yolo2_conv2d_ws otherwise. */
int yolo2_conv2d_bias_leaky(const void *P, const void *F, const float *bias, void *O, float *ws, size_t ws_bytes, int B,
                            int H, int W, int Cp, int ldp, int Nf, int ldo, int ksize, float alpha, int dtype, void *stream);

/* Data-gradient convolution whose output dX is the gradient of a batch-normalised producer layer's activation
 * (a = leaky_relu(batch_norm(y)), model/yolo2/inference.py:62-66): as yolo2_conv2d_ws(dY, F, NULL, dX, ...) FOLLOWED BY
 * yolo2_bn_leaky_bwd_reduce(dX, ldo, Yprev, mean, var, gamma, beta, dgamma, dbeta, ...) -- the gradients of slim.batch_norm's
 * gamma/beta that tf.gradients derives (train.py:123-129) -- with the reduction done by the convolution's epilogue while the
 * tile is still on chip (partial rows in bn_part, f32 [2][YOLO2_BN_PART_ROWS][Nf], zero on entry and on return), so dX and
 * Yprev are not re-read by a separate pass.  Shapes the epilogue cannot take (Nf % (16/elem size) != 0, K-sliced grids) run
 * the two-step form internally with red_ws (>= yolo2_bn_workspace_bytes(Nf)): the results are the same up to summation order.
 * Yprev: [B*H*W][Nf] (pixel stride Nf).  pending: NULL -> dgamma/dbeta are final on return (stream order).  Non-NULL -> the
 * caller finishes: *pending = 1 means the sums are still in bn_part and yolo2_bn_part_to_grads must follow (a caller timing
 * the convolution launch alone brackets just this call); 0 means dgamma/dbeta were already written by the two-step form. */
int yolo2_conv2d_dgrad_bn(const void *dY, const void *F, void *dX, float *ws, size_t ws_bytes, int B, int H, int W, int Cp, int ldp,
                          int Nf, int ldo, int ksize, const void *Yprev, const float *mean, const float *var, const float *gamma,
                          const float *beta, float *dgamma, float *dbeta, float *bn_part, double *red_ws, float eps, float alpha,
                          int *pending, int dtype, void *stream);
/* 1 when yolo2_conv2d_dgrad_bn would take the sums from its epilogue for a data gradient with Nf output channels on B x H x W pixels, 0 when the launch
 * rule prefers a plain data gradient there (3x3, <= 64 channels, >= 100k pixels: the reduction then costs less as its own pass): a host that asks first
 * calls yolo2_conv2d_ws + yolo2_bn_leaky_bwd_reduce_part + yolo2_bn_leaky_bwd_apply_fin for such a layer (engine.Engine._bind does) */
int yolo2_conv2d_dgrad_bn_fuses(int B, int H, int W, int Nf, int ksize, int dtype);
/* column sums of the partial rows left by yolo2_conv2d_dgrad_bn: plane 0 -> dgamma, plane 1 -> dbeta; rows zeroed again */
int yolo2_bn_part_to_grads(float *bn_part, int C, float *dgamma, float *dbeta, void *stream);

/* Forward convolution of a batch-normalised layer (no bias): as yolo2_conv2d_ws, and the per-channel shifted sums
 *   sum_m (y[m,n] - shift[n]),  sum_m (y[m,n] - shift[n])^2        (y = the stored, rounded output)
 * are accumulated into bn_part, f32 [2][YOLO2_BN_PART_ROWS][Nf], by the convolution's own epilogue (f32 atomics on
 * row (tile % ROWS)); yolo2_bn_finalize turns them into the batch moments.  This replaces the separate statistics
 * pass over y of slim.batch_norm's tf.nn.moments (model/yolo2/inference.py:62-66).  bn_part must be all zero on entry
 * (yolo2_bn_finalize leaves it zero); shift is any per-channel estimate of the mean (the engine passes
 * moving_mean): it only conditions the single-pass variance.  Requires ldo == Nf. */
#define YOLO2_BN_PART_ROWS 256
int yolo2_conv2d_bn(const void *P, const void *F, void *O, float *ws, size_t ws_bytes, int B, int H, int W, int Cp, int ldp,
                    int Nf, int ldo, int ksize, const float *shift, float *bn_part, int dtype, void *stream);
/* mean[c] = shift[c] + S1/M, var[c] = S2/M - (S1/M)^2 (biased, f64), optional moving-average update as
 * yolo2_bn_stats_ema (moving_* may both be NULL); zeroes bn_part. */
int yolo2_bn_finalize(float *bn_part, const float *shift, long M, int C, float *mean, float *var, float *moving_mean,
                      float *moving_var, double decay, void *stream);

/* ---- Consumers that finalise the partial rows themselves (one launch instead of finalize + apply): slim.batch_norm's
 * tf.nn.moments + assign_moving_average + normalisation (model/yolo2/inference.py:62-66) and the dgamma / dbeta + dY of its gradient
 * (train.py:123-129), fed by the partial rows a producer left behind: yolo2_conv2d_bn / yolo2_conv2d_dgrad_bn (bn_part layout
 * [2][YOLO2_BN_PART_ROWS][C]) or a *_reduce_part call ([2][rows][C] in its ws).  `rows` = how many partial rows the producer used:
 * yolo2_last_bn_part_rows() right after the producing call on the same host thread (a producer wraps its tiles around few rows when
 * it has many, so that this prologue stays short), or *rows of the reduce_part call.  yolo2_bn_fin_supported(rows, C, dtype) says
 * whether the shape qualifies (C / (16 / elem size) a power of two; rows x channel-slice small enough that a per-workgroup
 * prologue beats a separate finalisation launch); otherwise use yolo2_bn_finalize / yolo2_bn_part_to_grads / the plain pairs.
 * The rows are only READ here.  zero / zero_floats (multiple of 4, 16-byte aligned; may be NULL / 0): a float range this launch
 * clears on the side -- the engine keeps two partial buffers and has each consumer clear the one its predecessor finished with, so
 * every producer finds zero rows without a memset launch.  Results equal the two-launch forms up to f64 summation order.
 * `shift` (the per-channel values the producer subtracted) is read by EVERY workgroup of the launch while the first workgroup of each
 * channel slice stores mean / var and updates moving_mean / moving_var in place: it must not alias any of them (YOLO2_E_ARG if it
 * does).  A caller that shifts by the moving mean hands in a copy taken before the step (engine.Engine.state_snap). */
/* yolo2_bn_leaky_pool_fin: ymax (optional, [B*(H/2)*(W/2)][C], dense) receives the raw convolution output AT the arg-max of each
 * window.  Only arg-max positions carry gradient, so the layer's backward reduction is then yolo2_bn_leaky_bwd_reduce_part over
 * (dP, ymax) at the POOLED resolution: a quarter of the rows, and neither Y nor idx are read.  A_full (optional, dense [B*H*W][C])
 * also receives the un-pooled activation, for a layer whose activation has a second reader (idx may then be NULL). */
int yolo2_last_bn_part_rows(void);
int yolo2_bn_fin_supported(int rows, int C, int dtype);
int yolo2_bn_leaky_fin(const void *Y, const float *bn_part, int rows, const float *shift, float *mean, float *var, float *moving_mean,
                       float *moving_var, double decay, const float *gamma, const float *beta, void *A, long M, int C, int lda,
                       float eps, float alpha, float *zero, long zero_floats, int dtype, void *stream);
int yolo2_bn_leaky_pool_fin(const void *Y, const float *bn_part, int rows, const float *shift, float *mean, float *var,
                            float *moving_mean, float *moving_var, double decay, const float *gamma, const float *beta, void *P,
                            unsigned char *idx, void *ymax, void *A_full, int B, int H, int W, int C, int ldp, float eps, float alpha,
                            float *zero, long zero_floats, int dtype, void *stream);
/* part: [2][rows][C] with plane_stride floats between the two planes (YOLO2_BN_PART_ROWS * C for bn_part, rows * C for a ws) */
int yolo2_bn_leaky_bwd_apply_fin(const void *dA, int ldda, const void *Y, const float *mean, const float *var, const float *gamma,
                                 const float *beta, const float *part, int rows, long plane_stride, float *dgamma, float *dbeta,
                                 void *dY, long M, int C, float eps, float alpha, float *zero, long zero_floats, int dtype, void *stream);
int yolo2_bn_leaky_pool_bwd_apply_fin(const void *dP, int lddp, const unsigned char *idx, const void *Y, const float *mean,
                                      const float *var, const float *gamma, const float *beta, const float *part, int rows,
                                      long plane_stride, float *dgamma, float *dbeta, void *dY, int B, int H, int W, int C, float eps,
                                      float alpha, float *zero, long zero_floats, int dtype, void *stream);
/* The image layer (3 channels in an 8-wide pixel, 32 filters, 3x3 + batch_norm + leaky_relu + 2x2 max_pool: reference model/yolo2/inference.py:62-66):
 * yolo2_bn_leaky_pool_bwd_apply_fin + yolo2_conv2d_wgrad in ONE launch.  The layer's output gradient (the largest tensor of the network) is formed
 * per 32-pixel row segment in LDS from the raw forward output Y [B,H,W,32], the pooled gradient dP [B,H/2,W/2,lddp] and the arg-max codes idx
 * [B,H/2,W/2,32] and consumed there by the filter-gradient MFMAs; dgamma / dbeta come from the partial rows like in the *_fin call.  dW [3][3][Cin][32]
 * f32 is ACCUMULATED into (zero it first); X [B,H,W,8]. */
int yolo2_first_layer_wgrad_bn(const void *X, const void *Y, const void *dP, int lddp, const unsigned char *idx, const float *mean, const float *var,
                               const float *gamma, const float *beta, const float *part, int rows, long plane_stride, float *dgamma, float *dbeta,
                               float *dW, int B, int H, int W, int Cin, float eps, float alpha, float *zero, long zero_floats, int dtype, void *stream);
/* pass 1 of the BN + leaky backward alone: partial rows [2][*rows][C] (f32) left in ws for a *_bwd_apply_fin call */
int yolo2_bn_leaky_bwd_reduce_part(const void *dA, int ldda, const void *Y, const float *mean, const float *var, const float *gamma,
                                   const float *beta, double *ws, int *rows, int rows_limit, long M, int C, float eps, float alpha,
                                   int dtype, void *stream);
int yolo2_bn_leaky_pool_bwd_reduce_part(const void *dP, int lddp, const unsigned char *idx, const void *Y, const float *mean,
                                        const float *var, const float *gamma, const float *beta, double *ws, int *rows,
                                        int rows_limit, int B, int H, int W, int C, float eps, float alpha, int dtype, void *stream);

/* ---- The image layer fused with its consumers: conv0 (3 -> 32 filters, 3x3; model/yolo2/inference.py:72-74 `net = conv2d(net, 32)` followed
 * by `max_pool2d`) is 0.9 % of the arithmetic but its raw output is the largest tensor of the network (5.5 M pixels x 32 channels per
 * image).  These entries never store it: each one re-computes the output patch it needs from the image (P: [B,H,W,8], 3 real channels;
 * F: the forward filter operand of yolo2_filter_prep, [32][72]) and applies its elementwise work to the accumulators.  Results equal
 *   yolo2_conv2d_bn + yolo2_bn_finalize                      (yolo2_first_layer_stats + yolo2_bn_finalize)
 *   yolo2_bn_leaky_pool on the stored output                 (yolo2_first_layer_bn_leaky_pool)
 *   yolo2_bn_leaky_pool_bwd_reduce / _bwd_apply               (yolo2_first_layer_pool_bwd_reduce + yolo2_bn_part_to_grads / ..._bwd_apply)
 * bit for bit up to f32 summation order: the recomputed output is rounded to dtype exactly where the unfused path stores it.
 * H and W even; bn_part: f32 [2][YOLO2_BN_PART_ROWS][32], zero on entry. */
int yolo2_first_layer_stats(const void *P, const void *F, int B, int H, int W, const float *shift, float *bn_part, int dtype, void *stream);
int yolo2_first_layer_bn_leaky_pool(const void *P, const void *F, const float *mean, const float *var, const float *gamma,
                                    const float *beta, void *Pout, unsigned char *idx, int B, int H, int W, int ldp, float eps,
                                    float alpha, int dtype, void *stream);
int yolo2_first_layer_pool_bwd_reduce(const void *P, const void *F, const void *dP, int lddp, const unsigned char *idx,
                                      const float *mean, const float *var, const float *gamma, const float *beta, float *bn_part,
                                      int B, int H, int W, float eps, float alpha, int dtype, void *stream);
int yolo2_first_layer_pool_bwd_apply(const void *P, const void *F, const void *dP, int lddp, const unsigned char *idx,
                                     const float *mean, const float *var, const float *gamma, const float *beta,
                                     const float *dgamma, const float *dbeta, void *dY, int B, int H, int W, float eps, float alpha,
                                     int dtype, void *stream);

/* ---- batch norm + leaky ReLU: closure model/yolo2/inference.py:62-66 + model/yolo/function.py:21-24
 * Y is the raw convolution output [M = B*H*W][C] (pixel stride C). */
/* batch mean and biased variance over M rows (tf.nn.moments); ws: >= 1025*C doubles of scratch
 * (per-block partial sums; any content) */
int yolo2_bn_stats(const void *Y, float *mean, float *var, double *ws, long M, int C,
                   int dtype, void *stream);
/* yolo2_bn_stats + yolo2_bn_ema in one pass (the moving-average update rides in the finalisation kernel) */
int yolo2_bn_stats_ema(const void *Y, float *mean, float *var, float *moving_mean, float *moving_var,
                       double decay, double *ws, long M, int C, int dtype, void *stream);
/* moving -= f32(1-decay)*(moving-batch)   (assign_moving_average, decay 0.999; 1-decay is formed in
 * double and then rounded, as TF's Python side does) */
int yolo2_bn_ema(float *moving_mean, float *moving_var, const float *mean, const float *var,
                 int C, double decay, void *stream);
/* A[m,c] = leaky(gamma*(Y-mean)/sqrt(var+eps)+beta); A has pixel stride lda (concat target) */
int yolo2_bn_leaky(const void *Y, const float *mean, const float *var, const float *gamma,
                   const float *beta, void *A, long M, int C, int lda, float eps, float alpha,
                   int dtype, void *stream);
/* backward, pass 1: dgamma[c] = sum g*xhat, dbeta[c] = sum g with g = dA * leaky'(z);
 * ws: >= 1024*C doubles of scratch */
int yolo2_bn_leaky_bwd_reduce(const void *dA, int ldda, const void *Y, const float *mean,
                              const float *var, const float *gamma, const float *beta,
                              float *dgamma, float *dbeta, double *ws, long M, int C,
                              float eps, float alpha, int dtype, void *stream);
/* backward, pass 2: dY = gamma*inv*(g - dbeta/M - xhat*dgamma/M) */
int yolo2_bn_leaky_bwd_apply(const void *dA, int ldda, const void *Y, const float *mean,
                             const float *var, const float *gamma, const float *beta,
                             const float *dgamma, const float *dbeta, void *dY, long M, int C,
                             float eps, float alpha, int dtype, void *stream);

/* ---- BN + leaky + 2x2/2 max pool fused (layers whose only consumer is the pool: model/yolo2/inference.py:70-95,
 * `net = slim.layers.max_pool2d(net)` right after a conv): the full-resolution activation and its gradient never reach
 * HBM.  P[b,oh,ow,c] = max over the window of leaky(bn(Y)) rounded to dtype; idx (may be NULL for inference) receives
 * one byte per pooled element: the window position 0..3 (row-major) of the FIRST maximum, which is where
 * tf.nn.max_pool's gradient goes.  The backward pair equals yolo2_maxpool_bwd followed by yolo2_bn_leaky_bwd_*. */
int yolo2_bn_leaky_pool(const void *Y, const float *mean, const float *var, const float *gamma, const float *beta, void *P,
                        unsigned char *idx, int B, int H, int W, int C, int ldp, float eps, float alpha, int dtype, void *stream);
int yolo2_bn_leaky_pool_bwd_reduce(const void *dP, int lddp, const unsigned char *idx, const void *Y, const float *mean,
                                   const float *var, const float *gamma, const float *beta, float *dgamma, float *dbeta,
                                   double *ws, int B, int H, int W, int C, float eps, float alpha, int dtype, void *stream);
int yolo2_bn_leaky_pool_bwd_apply(const void *dP, int lddp, const unsigned char *idx, const void *Y, const float *mean,
                                  const float *var, const float *gamma, const float *beta, const float *dgamma,
                                  const float *dbeta, void *dY, int B, int H, int W, int C, float eps, float alpha,
                                  int dtype, void *stream);

/* ---- max pool 2x2, SAME: slim.layers.max_pool2d, model/yolo2/inference.py:38,42,74,83,96 ------
 * stride 2 (H,W even) or stride 1 (pads bottom/right; tiny model).  Backward routes to the
 * first maximum in row-major window order. */
int yolo2_maxpool_fwd(const void *A, void *P, int B, int H, int W, int C, int stride,
                      int dtype, void *stream);
int yolo2_maxpool_bwd(const void *A, const void *dP, void *dA, int B, int H, int W, int C,
                      int stride, int dtype, void *stream);
/* dA += the routed gradient, rounded to the element type like yolo2_maxpool_bwd into a temporary + yolo2_add_inplace (stride 2 only):
 * the second writer of a tensor's gradient (Darknet-19's passthrough fan-out at the 26x26 stage) */
int yolo2_maxpool_bwd_acc(const void *A, const void *dP, void *dA, int B, int H, int W, int C, int dtype, void *stream);

/* ---- reorg / concat: model/yolo2/function.py:22-29, model/yolo2/inference.py:114-116 ----------
 * out[b,y,x,(sy*2+sx)*C+c] = in[b,2y+sy,2x+sx,c]; `out` has pixel stride ldo so it can be the
 * leading channels of the concat buffer.  yolo2_reorg_bwd is the inverse move. */
int yolo2_reorg(const void *in, void *out, int B, int H, int W, int C, int ldo, int dtype,
                void *stream);
int yolo2_reorg_bwd(const void *dout, int ldd, void *din, int B, int H, int W, int C, int dtype,
                    void *stream);
/* dst[m, 0:C] = src[m, 0:C] with independent pixel strides (tf.concat second operand / slice) */
int yolo2_copy_channels(const void *src, int lds, void *dst, int ldd, long M, int C, int dtype,
                        void *stream);
/* dst += src elementwise over n elements (passthrough gradient join) */
int yolo2_add_inplace(void *dst, const void *src, long n, int dtype, void *stream);
/* dbias[c] = sum_m dY[m,c]  (final conv biases, model/yolo2/inference.py:118); ws >= 512*ld doubles */
int yolo2_bias_grad(const void *dY, int ld, float *dbias, double *ws, long M, int C, int dtype,
                    void *stream);

/* ---- input: tf.image.per_image_standardization train.py:103 / utils/preprocess.py:23-25 --------
 * img: f32 [B,H,W,3] (0..255); out: dtype [B,H,W,8] (channels 3..7 zero).
 * mode 0: (x-mean)/max(std,1/sqrt(N)) per image; mode 1: x/255 (detect.py:37-38);
 * mode 2: plain cast.  ws: yolo2_image_prep_workspace_bytes(B) (mode 0: 64 partial (sum, sum of squares) pairs per image; any content). */
int yolo2_image_prep(const float *img, void *out, double *ws, int B, int HW, int mode, int dtype,
                     void *stream);

/* ---- head decode: Model.__init__, model/yolo2/__init__.py:28-59 (detection block :50-56) ------
 * logits [B,cells,ld] (channel order per anchor: iou,x,y,w,h,cls...), anchors f32[A][2] (w,h).
 * Outputs f32: conf [B,cells,A,C], xy_min/xy_max [B,cells,A,2]; nan_flag (int, device) is set
 * non-zero if any output is NaN/Inf (tf.check_numerics, detect.py:70). */
int yolo2_head_decode(const void *logits, int ld, const float *anchors, float *conf,
                      float *xy_min, float *xy_max, int *nan_flag, int B, int cell_h, int cell_w,
                      int A, int C, int dtype, void *stream);

/* The other Model attributes the reference's callers read (demo_detect.py:62: prob, iou, xy_min, wh; model/yolo2/__init__.py:36-56),
 * f32: iou [B,cells,A] = sigmoid(ch 0), prob [B,cells,A,C] = softmax(classes), xy [B,cells,A,2] = cell_xy + sigmoid(ch 1:3),
 * wh [B,cells,A,2] = exp(ch 3:5) * anchors.  Any output may be NULL (at least one must not be). */
int yolo2_head_decode_attrs(const void *logits, int ld, const float *anchors, float *iou, float *prob,
                            float *xy, float *wh, int B, int cell_h, int cell_w, int A, int C,
                            int dtype, void *stream);

/* ---- loss forward+backward: Objectives, model/yolo2/__init__.py:62-94 + Builder :114-119 -----
 * labels (f32): mask[B,cells], prob[B,cells,C], coords[B,cells,4], off_min/off_max[B,cells,2],
 * areas[B,cells].  hparam = HOST pointer to the 4 weights {iou_best, iou_normal, coords, prob}.
 * objectives (f32[4], same order) receives the UNWEIGHTED sums / cnt, cnt = B*cells*A.
 * dlogits (dtype, [B,cells,ld], may be NULL) receives d(total weighted loss)/d logits, padding
 * lanes zeroed.  ws: >= 4*ceil(B*cells*L/256) floats, L = smallest power of two >= A. */
int yolo2_loss(const void *logits, int ld, const float *anchors, const float *mask,
               const float *prob, const float *coords, const float *off_min, const float *off_max,
               const float *areas, const float *hparam, float *objectives, void *dlogits,
               float *ws, int B, int cell_h, int cell_w, int A, int C, int dtype, void *stream);
/* yolo2_loss in two halves: the kernel that writes dlogits and per-workgroup partial sums into ws (all a training step needs), and the
 * reduction of those partials into the four objective values, run only when they are read (summaries). */
int yolo2_loss_partials(const void *logits, int ld, const float *anchors, const float *mask, const float *prob,
                        const float *coords, const float *off_min, const float *off_max, const float *areas,
                        const float *hparam, void *dlogits, float *ws, int B, int cell_h, int cell_w, int A, int C, int dtype,
                        void *stream);
int yolo2_loss_objectives(const float *ws, float *objectives, int B, int cell_h, int cell_w, int A, void *stream);

/* ---- YOLO (v1) family, SURVEY 8f-4: model/yolo/__init__.py:37-100, model/yolo/inference.py:24-66 --------------------------
 * Network output row per image [cells*C class scores | cells*boxes*(iou, x, y, sqrt_w, sqrt_h)], all linear; `ld` = row stride.
 * yolo1_loss: Objectives + d(total weighted loss)/d(net) (label tensors and hparam order as yolo2_loss; prob is masked by the
 * cell mask, cnt = B*cells*boxes); ws as yolo2_loss with A = boxes_per_cell.  yolo1_head_decode: the detection block (conf = iou *
 * prob, corners in cell units). */
int yolo1_loss(const void *net, int ld, const float *mask, const float *prob, const float *coords, const float *off_min,
               const float *off_max, const float *areas, const float *hparam, float *objectives, void *dnet, float *ws,
               int B, int cell_h, int cell_w, int boxes_per_cell, int C, int dtype, void *stream);
int yolo1_head_decode(const void *net, int ld, float *conf, float *xy_min, float *xy_max, int *nan_flag, int B, int cell_h,
                      int cell_w, int boxes_per_cell, int C, int dtype, void *stream);
/* leaky_relu backward of an un-normalised layer, from its OUTPUT a: dZ = a >= 0 ? dA : alpha*dA (model/yolo/function.py:21-24;
 * n elements, a multiple of 8 (bf16) / 4 (f32)) */
int yolo2_leaky_bwd(const void *A, const void *dA, void *dZ, long n, float alpha, int dtype, void *stream);
/* slim.layers.dropout in training (model/yolo/inference.py:57,60): Y = X * keep / keep_prob, keep ~ Bernoulli(keep_prob) from a
 * counter-based hash of (seed, index); the byte mask is written for the backward pass.  seed == 0: `mask` is an input. */
int yolo2_dropout(const void *X, void *Y, unsigned char *mask, long n, float keep_prob, unsigned long long seed, int dtype,
                  void *stream);
int yolo2_dropout_bwd(const void *dY, const unsigned char *mask, void *dX, long n, float keep_prob, int dtype, void *stream);
/* slim.l2_regularizer(scale) on a weight tensor (model/yolo/inference.py:55): g += scale*w, *loss += scale*sum(w^2)/2 (device double) */
int yolo2_l2_regularizer(const float *w, float *g, long n, float scale, double *loss, void *stream);

/* ---- NMS: utils/postprocess.py:39-51 (iou :21-36), batched over images -----------------------
 * conf [B,N,C] f32 is updated in place exactly as the reference mutates it; order_out [B,N]
 * (int, may be NULL) receives per image the box indices in the order of the reference's
 * returned list (stable descending sort by the last class, ties carried from earlier classes).
 * ws: >= B*C*N ints. */
int yolo2_nms(float *conf, const float *xy_min, const float *xy_max, int *order_out, int *ws,
              int B, int N, int C, float threshold, float threshold_iou, void *stream);

/* ---- optimizers: train.py:70-80 (TF-1.0 Apply* kernels), flat f32 buffers of n elements ------ */
int yolo2_adam(float *w, const float *g, float *m, float *v, long n, float alpha, float beta1,
               float beta2, float eps, float gscale, void *stream);   /* alpha = lr*sqrt(1-b2^t)/(1-b1^t) */
int yolo2_momentum(float *w, const float *g, float *acc, long n, float lr, float momentum,
                   float gscale, void *stream);
int yolo2_sgd(float *w, const float *g, long n, float lr, float gscale, void *stream);
int yolo2_rmsprop(float *w, const float *g, float *ms, float *mom, long n, float lr, float decay,
                  float momentum, float eps, float gscale, void *stream);
int yolo2_adagrad(float *w, const float *g, float *acc, long n, float lr, float gscale,
                  void *stream);
int yolo2_adadelta(float *w, const float *g, float *acc, float *acc_update, long n, float lr,
                   float rho, float eps, float gscale, void *stream);
/* tf.train.FtrlOptimizer (train.py:78, config.ini [optimizer_ftrl]): TF-1.0 ApplyFtrl;
 * accum starts at initial_accumulator_value, linear at 0; lr_power <= 0 (-0.5 takes the sqrt form) */
int yolo2_ftrl(float *w, const float *g, float *accum, float *linear, long n, float lr, float lr_power,
               float l1, float l2, float gscale, void *stream);
/* x *= scale (1/world averaging of the all-reduced gradient ahead of clip_by_norm) */
int yolo2_scale(float *x, long n, float scale, void *stream);
/* x[a:b] = 0 for nranges half-open element ranges given as a HOST array {a0,b0,a1,b1,...}: the accumulating filter
 * gradients (yolo2_conv2d_wgrad_accumulates) are cleared per step, nothing else */
int yolo2_zero_ranges(float *x, const long *ranges_host, int nranges, void *stream);
/* Inference-time batch-norm folding for yolo2_conv2d_bias_leaky (slim.batch_norm is_training=False,
 * model/yolo2/inference.py:62-66): Wf[r,n] = W[r,n]*s[n], bias[n] = beta[n] - moving_mean[n]*s[n],
 * s = gamma/sqrt(moving_var+eps); W, Wf: HWIO f32 viewed as [rows = k*k*Cin][C]. */
int yolo2_bn_fold(const float *W, const float *gamma, const float *beta, const float *moving_mean,
                  const float *moving_var, float *Wf, float *bias, long rows, int C, float eps, void *stream);
/* per-tensor tf.clip_by_norm (slim create_train_op clip_gradient_norm, train.py:127-129):
 * seg_off int64[nseg+1] element offsets into g; ws >= nseg doubles */
int yolo2_clip_by_norm(float *g, const long *seg_off, int nseg, float clip, double *ws,
                       void *stream);

/* ---- diagnostics -----------------------------------------------------------------------------
 * fills out[64*4] with the raw result of ds_read_b64_tr_b16 over a 0..N ramp (layout self-test) */
int yolo2_selftest_tr16(short *out, void *stream);
/* Which kernel variant the calling thread's most recent yolo2_conv2d* / yolo2_conv2d_wgrad call launched (the choice is
 * shape-driven, so parity tests assert that the variant they mean to check is the one that ran):
 *   conv plan  out8 = {tile pixels BM, tile filters BN, waves, 16-byte chunks per K row, LDS stages,
 *                      split (0 none, 1 K-sliced + finishing kernel, 2 stream-K), grid x, grid y};  all -1 = first-layer kernel
 *   wgrad plan out8 = {tile channels BC, tile filters BN, waves, taps paired, pixel ranges, XCD-local placement,
 *                      workgroups, direct store (no atomics)};                                      all -1 = first-layer kernel */
int yolo2_debug_last_conv_plan(int *out8);
/* measurement hook: launches an empty kernel (bench.py calibrates the overhead of its HIP-event brackets with it) */
int yolo2_debug_noop(void *stream);
/* test hook (process-wide): 0 = per-tap 3x3 kernels only; 2 (and any other non-zero value) = the ping-pong tap-fused kernel (csrc/conv_pp.hip) wherever
 * its launch rule admits it; 3 = the loader / consumer member of the same family under the same rule (csrc/conv_s4.hip: four computing waves of
 * 128 x 64 + four loader waves; plan word `waves` = 4).  YOLO2_IGEMM_TAP sets the initial value. */
int yolo2_debug_set_igemm_tap(int mode);
/* experiments build of csrc/conv_s4.hip only (timing ablations, phase stamps); ignored by the product build */
int yolo2_debug_set_s4_abl(int abl);
/* test / A-B hook for the ping-pong kernel.  grid: 0 = by rule (default), 1 = stream-K over one workgroup per CU, 2 = one workgroup per
 * tile, >= 8 = that many workgroups, <= -2 = every tile cut into exactly -grid shares, -1 = keep; a stream-K grid never exceeds the
 * number of K steps.  sched: the kernel's LOAD-phase order (2 in the product build; others only in scripts/pp_experiments_build.sh's;
 * an order the library was not built with sends the layer to the per-tap kernels).  min_steps / min_share: the rule's gates (K steps
 * per tile, K steps per workgroup).  A negative argument keeps the current value. */
int yolo2_debug_set_pp(int grid, int sched, int min_steps, int min_share);
/* tests / A-B: cost units (in K steps) the stream-K partition of the ping-pong kernel charges to the workgroup that owns a tile -- it also waits for
 * and sums its partners' partial tiles and runs the epilogue, so it gets up to that many fewer K steps (csrc/conv_pp.hip "cost-balanced shares");
 * 0 = equal K-step shares.  The library clamps it below half a share.  YOLO2_PP_CV sets the initial value. */
int yolo2_debug_set_pp_cost(int cv);
/* tests: the stream-K owners' wait limit in microseconds (0 = the default, 2 s); unclamped != 0 lets a grid forced through yolo2_debug_set_pp
 * exceed the number of K steps -- a partition no owner can be served by, which must end in yolo2_check_async_errors() == YOLO2_E_LAUNCH */
int yolo2_debug_set_streamk_wait_us(int us, int unclamped);
int yolo2_debug_last_wgrad_plan(int *out8);
/* process-wide, tests / A-B only: 0 = product rule (bf16 3x3 layers: the row-of-taps kernel of csrc/conv_wgrad3.hip; everything else: the per-tap
 * transpose-read kernel of csrc/conv_wgrad.hip); 1 = scalar reference gather (layout-proof, slow); 2 = the per-tap kernel for every shape
 * (round 4's path); 10 + v = variant v (0..2) of the row-of-taps kernel for every shape it can take.  YOLO2_WGRAD_VARIANT sets the initial value. */
void yolo2_debug_set_wgrad_variant(int variant);
/* host-side plan of the row-of-taps filter-gradient kernel for a shape on a device with `cus` compute units (force_variant < 0: by rule):
 * out9 = {variant (-1: not taken), pixel ranges, padded pixels per range, workgroups, XCD mapping, plain stores, channel tile, filter tile, waves} */
int yolo2_debug_wgrad_row_plan(int B, int H, int W, int Cin, int Cout, int cus, int force_variant, int *out9);
/* the multiply-high division constants of that kernel's DMA address arithmetic: floor(q / d) == (uint64(q) * m >> 32) >> s for 0 <= q < 2^31 */
int yolo2_debug_magic_u32(unsigned d, unsigned *m, unsigned *s);

/* ---- on-device input pipeline: utils/data/__init__.py:50-109,162-175 + utils/preprocess.py:28-71 after JPEG
 * decode.  `src` holds the decoded uint8 RGB images back to back (any sizes); per image the caller supplies the
 * outcome of every random draw of the reference (tf.random_uniform / tf.cond):
 *   crop_*      tf.image.crop_to_bounding_box arguments (whole image when random_crop is not taken)
 *   flags       which branches were taken; brightness = delta added (tf.image.random_brightness, max_delta 63),
 *               saturation / contrast = factors in [0.5, 1.5], hue = delta in [-0.032, 0.032],
 *               noise_scale = U(5, 15) multiplying tf.truncated_normal (generated on the device from noise_seed)
 * out: f32 [B, H, W, 3] in 0..255 (tf.clip_by_value), ready for yolo2_image_prep.  ws: >= 3*B doubles when any image
 * takes the contrast branch (per-channel means of AdjustContrastv2). */
#define YOLO2_AUG_FLIP 1u
#define YOLO2_AUG_BRIGHTNESS 2u
#define YOLO2_AUG_SATURATION 4u
#define YOLO2_AUG_HUE 8u
#define YOLO2_AUG_CONTRAST 16u
#define YOLO2_AUG_NOISE 32u
#define YOLO2_AUG_GRAY 64u
typedef struct yolo2_augment_params {
    long long src_offset;            /* byte offset of the image's first pixel in src */
    int src_w, src_h;
    int crop_x, crop_y, crop_w, crop_h;
    unsigned flags;
    float brightness, saturation, hue, contrast, noise_scale;
    unsigned long long noise_seed;
} yolo2_augment_params;
int yolo2_augment_images(const unsigned char *src, const yolo2_augment_params *params_device, double *ws, float *out,
                         int B, int H, int W, int any_contrast, void *stream);

/* transform_labels (utils/data/__init__.py:112-145) for a batch: objects_coord [total,4] = (xmin,ymin,xmax,ymax) in
 * [0,1], objects_class [total], image b owns objects first_object[b] .. first_object[b+1]-1 (processed in order: the
 * last object of a cell wins, class bits accumulate).  Outputs are the loss's label tensors, each [B, cells, ...],
 * bit-exact with the reference.  error_flag (device int, caller zeroes it): bit 0 = object outside the grid or class
 * out of range (the reference raises IndexError), bit 1 = negative box extent (the reference asserts). */
int yolo2_transform_labels(const int *objects_class, const float *objects_coord, const int *first_object, float *mask,
                           float *prob, float *coords, float *offset_xy_min, float *offset_xy_max, float *areas, int B,
                           int classes, int cell_width, int cell_height, int *error_flag, void *stream);

/* ---- data-parallel support (new work: the reference has no multi-GPU path, README.md:99).  The collectives themselves run in
 * torch.distributed (RCCL); the library only provides what has to happen on the device around them. */
/* Stream-K convolution launches use n workgroups instead of one per CU (0 = default): a data-parallel process leaves the CUs an
 * RCCL ring occupies to it.  Process-wide; never affects results. */
int yolo2_set_stream_workgroups(int n);
int yolo2_get_stream_workgroups(void);
/* gradient wire format: round an f32 range to bf16 / widen it back (16-byte aligned pointers) */
int yolo2_cast_f32_bf16(const float *src, void *dst, long n, void *stream);
int yolo2_cast_bf16_f32(const void *src, float *dst, long n, void *stream);
/* test instrument: `workgroups` persistent workgroups, one whole CU each (160 KiB LDS), until *stop != 0 or max_us have passed;
 * *started counts the resident ones (what a collective's persistent kernels look like to the dispatcher) */
int yolo2_debug_occupy(int workgroups, int *stop, int *started, int max_us, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* YOLO2_HIP_H */
