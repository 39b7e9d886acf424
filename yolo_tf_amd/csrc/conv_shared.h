// Declarations shared by the implicit-GEMM translation units (conv_igemm.hip, conv_pp.hip).
#pragma once
#include "common.h"

#define Y2_OOB 0x80000000u   // any offset >= num_records makes the buffer DMA return zeros

// Stream-K hand-off flags (conv_igemm.hip owns the pool): one set = Y2_STREAM_FLAG_WORDS flag words -- one per workgroup -- followed by a
// status block: word [WORDS] counts waits that gave up, word [WORDS + 1] is the wait limit in wall-clock ticks (100 MHz; 0 = default).
#define Y2_STREAM_FLAG_WORDS 1024
#define Y2_STREAM_FLAG_STRIDE (Y2_STREAM_FLAG_WORDS + 64)
#define Y2_SK_DEFAULT_WAIT_TICKS 200000000u      // 2 s: five orders of magnitude above the longest legitimate wait (a partner's tail segment, < 100 us)
#if defined(__HIP_DEVICE_COMPILE__)
// The owner of a stream-K tile waits for the flag of partner workgroup p (called by ONE lane) and clears it.  The wait is bounded: a partition
// bug (a workgroup without work never raises its flag -- the hang behind launch_conv's grid clamp) or a lost partner must surface as an error,
// not as a hung device.  On a time-out the launch's results are wrong by construction (the owner sums an unfinished slot, and the late partner's
// flag may be consumed by a later launch of the same flag set): the status word makes yolo2_check_async_errors() return YOLO2_E_LAUNCH and re-zero
// the pool; hosts bound the exposure by polling it every step without a synchronisation (yolo2_async_error_snapshot: TrainSession.step,
// DetectSession.detect) and before anything is written to disk.  The fast path (flag already up) costs one load, as before.
__device__ __forceinline__ void y2_sk_wait_and_clear(unsigned *flags, int p) {
    if (__hip_atomic_load(flags + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
        const unsigned cfg = __hip_atomic_load(flags + Y2_STREAM_FLAG_WORDS + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long limit = cfg ? cfg : Y2_SK_DEFAULT_WAIT_TICKS, t0 = wall_clock64();
        while (__hip_atomic_load(flags + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > limit) {
                __hip_atomic_fetch_add(flags + Y2_STREAM_FLAG_WORDS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __hip_atomic_store(flags + p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // exactly one consumer per flag
}
#else
__device__ inline void y2_sk_wait_and_clear(unsigned *, int) {}      // (host pass of a __global__ body)
#endif

// Data-gradient launches whose output IS the gradient dA of a batch-normalised producer layer (a = leaky(bn(y))) can reduce that
// layer's BN + leaky backward sums in their epilogue: Y non-NULL selects it.  Per output element dz = dA * leaky'(z),
// xhat = (y - mean) * rstd; the tile adds its columns' sum(dz * xhat) [plane 0] and sum(dz) [plane 1] to the partial rows the
// forward statistics use.  Replaces one full read of dA and y (bn_bwd_reduce_kernel) per layer with a read of y alone, issued while
// the tile is still in LDS.
struct Y2BnBwd {
    const void *Y;      // pre-normalisation output of the producer layer, [M][Nf] (pixel stride = Nf)
    const float *mean, *var, *gamma, *beta;
    float eps, alpha;
    // bits cleared from the partial-row index mask (Y2_BN_PART_ROWS - 1): grids with more (pixel tile, wave row) pairs than rows wrap
    // around R = 256 >> popcount(stat_mask_inv) rows, chosen by the host so that the consumer that finalises the rows in its prologue
    // (yolo2_bn_leaky_fin & co.) reads few of them while same-address atomic adds stay rare (y2_stat_rows in conv_igemm.hip).  0 = all 256 rows.
    int stat_mask_inv;
};

// conv_pp.hip: ping-pong tap-fused 3x3 kernel (bf16, 256 x 128 tile); returns non-zero when the image is too wide for its halo buffers
int y2_conv3x3_pp_launch(const void *P, unsigned p_bytes, const void *F, unsigned f_bytes, const float *bias, void *O, float *ws, int H, int W, int Cp,
                         int ldp, int Nf, int ldo, int M, int NT, const float *bn_shift, float *bn_part, unsigned *sk_flags, float act_alpha,
                         const Y2BnBwd &bz, int k_rotate, int grid, int sched, int cv, hipStream_t st);

// conv_s4.hip: the loader / consumer member of the same family (four computing waves of 128 x 64, four loader waves); same contract
int y2_conv3x3_s4_launch(const void *P, unsigned p_bytes, const void *F, unsigned f_bytes, const float *bias, void *O, float *ws, int H, int W, int Cp,
                         int ldp, int Nf, int ldo, int M, int NT, const float *bn_shift, float *bn_part, unsigned *sk_flags, float act_alpha,
                         const Y2BnBwd &bz, int k_rotate, int grid, int abl, hipStream_t st);

// conv_wgrad3.hip: 3x3 filter gradient with one kernel row of taps per workgroup over a padded pixel index (bf16).  variant < 0: the shape is not
// taken (conv_wgrad.hip's per-tap kernel runs instead).
struct Y2W3Plan { int variant, ks, qchunk, blocks, remap, direct, BC, BN, waves; };
Y2W3Plan y2_wgrad3_plan(int B, int H, int W, int Cin, int Cout, int cus, int force_variant);
int y2_wgrad3_launch(const Y2W3Plan &p, const void *X, const void *dY, float *dW, int B, int H, int W, int Cin, int ldx, int Cout, int ldy, hipStream_t st);
void y2_magic_u32(unsigned d, unsigned *m, unsigned *s);

// conv_wgrad_c32.hip: 3x3 filter gradient for 32 input channels and 64 filters (Darknet-19 conv1), bf16: all nine taps in one workgroup over a run of image
// rows, every operand byte read once.  y2_w32_wgrad returns non-zero when the rows do not fit its LDS plan (the caller then takes the per-tap kernel).
bool y2_w32_shape(int Cin, int ldx, int Cout, int ldy, int ksize, int dtype);
int y2_w32_blocks(int B, int H, int W, int cus);      // workgroups of the launch; 0: not taken
int y2_w32_wgrad(const void *X, const void *dY, float *dW, int B, int H, int W, int cus, int *blocks, hipStream_t st);

// conv_c32.hip: persistent 3x3 forward for 32-channel inputs and 64 filters (Darknet-19 conv1), bf16.  y2_c32_fwd returns non-zero when the image is
// too wide for its LDS plan (the caller then takes the generic kernels).
bool y2_c32_shape(int Cp, int ldp, int Nf, int ldo, int ksize, int dtype);
int y2_c32_fwd(const void *P, const void *F, void *O, int B, int H, int W, const float *bias, float alpha, const float *bn_shift, float *bn_part,
               int cus, int *rows, hipStream_t st);

// conv_c64.hip: persistent 3x3 convolution for 64-channel inputs with 128 filters (Darknet-19 conv2 / conv4 forward) or 32 (conv1's data gradient), bf16,
// filters held in registers, one ring of pixel rows per workgroup.  y2_c64_fwd returns non-zero when the image is too wide for its LDS plan or the
// 32-filter form is asked for an epilogue it does not have (the caller then takes the generic kernels).
bool y2_c64_shape(int Cp, int ldp, int Nf, int ldo, int ksize, int dtype);
int y2_c64_fwd(const void *P, const void *F, void *O, int B, int H, int W, int Nf, const float *bias, float alpha, const float *bn_shift, float *bn_part,
               int cus, int *rows, hipStream_t st);

// conv_d1.hip: persistent 1x1 data gradient + the producer layer's BN / leaky backward sums for the wide early stages (conv3 / conv6), bf16
bool y2_d1_shape(int Cp, int ldp, int Nf, int ldo, int ksize, int dtype, long M);
int y2_d1_dgrad_bn(const void *P, const void *F, void *O, long M, int Cp, int Nf, float *bn_part, const Y2BnBwd &bz, int cus, int *rows, hipStream_t st);
