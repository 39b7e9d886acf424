// Declarations shared by the implicit-GEMM translation units (conv_igemm.hip, conv_pp.hip).
#pragma once
#include "common.h"

#define Y2_OOB 0x80000000u   // any offset >= num_records makes the buffer DMA return zeros

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
                         const Y2BnBwd &bz, int k_rotate, int grid, int sched, hipStream_t st);
