// 1x1 convolution DATA GRADIENT with the producer layer's batch-norm + leaky backward sums, for the wide early stages (bf16, gfx950).
//
// Replaces tf.gradients through slim.layers.conv2d + slim.batch_norm of reference model/yolo2/inference.py:79-84 (conv3 at 104 x 104, conv6 at 52 x 52:
// the 1x1 bottlenecks; train.py:127-129) where yolo2_conv2d_dgrad_bn's generic kernel is bound by LATENCY, not by what it moves: a 128 x 128 tile of a
// 64- or 128-channel reduction is one or two K steps of MFMA work, after which every tile pays, one after the other, the round trip of its y vectors,
// a cross-lane reduction of its sums and 128 atomic adds per wave -- 47.7 us for the 110 MB of conv3's gradient (20 us at HBM speed), 43.5 us for the
// 55 MB of conv6's (profiles/r06_bench_kernel_trace_single_stream.md).  Here
//   * workgroups are PERSISTENT over 128-pixel tiles; the filter (128 rows of this workgroup's column group) is staged into LDS once;
//   * the dY tile of tile t + 1 (LDS-DMA) and the producer's y vectors of tile t + 1 (registers) are requested before tile t's epilogue runs: no round
//     trip of a LOAD is waited for in the order it was issued (the stores' acknowledgement is: the co-resident workgroups cover it);
//   * a lane keeps the same eight channels in every store iteration of every tile: the BN-backward sums stay in sixteen registers for the whole launch
//     and meet across lanes ONCE, followed by one atomic add per value and wave (32 partial rows).
// Same arithmetic and rounding points as the generic epilogue (conv_igemm.hip): dX rounded to bf16, dz = dX_rounded * leaky'(z), xhat from the stored y.
#include "common.h"
#include "conv_shared.h"
#include <atomic>

#define D1_BM 128
#define D1_BN 128
#define D1_STAT_ROWS 32

// KCH = 16-channel groups of the reduction (4: 64 input channels of the data gradient, 8: 128).  ABL (experiments): +1 no output stores, +2 no y loads,
// +4 no sums arithmetic
template <int KCH, int ABL>
__global__ __launch_bounds__(512) void conv_d1_dgrad_bn_kernel(
    const bf16 *__restrict__ P, unsigned p_bytes, const bf16 *__restrict__ F, unsigned f_bytes, bf16 *__restrict__ O, int M, int Nf, int ntiles,
    float *__restrict__ bn_part, const Y2BnBwd bz) {
    typedef bf16 T;
    constexpr int ROWB = KCH * 32, CPR = ROWB / 16, RPP = 1024 / ROWB;          // bytes per operand row, 16-byte chunks per row, rows per 1 KiB DMA piece
    constexpr int FBYTES = D1_BN * ROWB, ABYTES = D1_BM * ROWB;
    constexpr int WROWS = 32, WROWB = 128, WSTRIDE = WROWB + 16, WCPR = 8, NIT = WROWS * WCPR / 64, VEC = 8;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[FBYTES + 2 * ABYTES + 8 * WROWS * WSTRIDE];
    unsigned char *const fl = smem, *const ab = smem + FBYTES, *const img = smem + FBYTES + 2 * ABYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;
    const int n0 = blockIdx.y * D1_BN;
    // source-side swizzle of a row's 16-byte chunks: 128-byte rows pair up in a 256-byte bank line ((row >> 1) & 7), 256-byte rows fill one (row & 15)
    auto swz = [](int row) { return CPR == 8 ? ((row >> 1) & 7) : (row & 15); };

    const __amdgpu_buffer_rsrc_t rsrcP = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(P), 0, p_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcF = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(F), 0, f_bytes, 0x00020000);
    typedef __attribute__((address_space(3))) void *lds_void_ptr;
    typedef const __attribute__((address_space(3))) bf16x8 *lds_frag_ptr;
    const int prow = lane / CPR, pchunk = lane % CPR;
    // ---- the filter rows n0 .. n0 + 127 (rows >= Nf lie beyond num_records: zeros), once
    for (int p = wave; p * 1024 < FBYTES; p += 8) {
        const int row = p * RPP + prow;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcF, (lds_void_ptr)(fl + p * 1024), 16, (unsigned)((n0 + row) * ROWB + ((pchunk ^ swz(row)) << 4)), 0, 0, 0);
    }
    auto stage = [&](int tile, int buf) {      // the dY rows of a tile (rows >= M lie beyond num_records: zeros)
        for (int p = wave; p * 1024 < ABYTES; p += 8) {
            const int row = p * RPP + prow;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcP, (lds_void_ptr)(ab + buf * ABYTES + p * 1024), 16,
                                                     (unsigned)((tile * D1_BM + row) * ROWB + ((pchunk ^ swz(row)) << 4)), 0, 0, 0);
        }
    };
    int tile = blockIdx.x;
    const int T_ = tile < ntiles ? (ntiles - tile + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    if (T_ > 0) stage(tile, 0);

    // ---- per-lane constants of the store loop: this lane's eight channels nb .. nb + 7 (the same in every iteration of every tile)
    const int nb = n0 + wn * 64 + (lane % WCPR) * VEC;
    float cmu[VEC], cinv[VEC], cga[VEC], cbt[VEC], ps[2][VEC];
#pragma unroll
    for (int k = 0; k < VEC; k += 4) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(bz.mean + nb + k), b = *reinterpret_cast<const f32x4 *>(bz.var + nb + k);
        const f32x4 c = *reinterpret_cast<const f32x4 *>(bz.gamma + nb + k), d = *reinterpret_cast<const f32x4 *>(bz.beta + nb + k);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            cmu[k + q] = a[q];
            cinv[k + q] = 1.0f / sqrtf(b[q] + bz.eps);
            cga[k + q] = c[q];
            cbt[k + q] = d[q];
            ps[0][k + q] = ps[1][k + q] = 0.f;
        }
    }
    // y vectors of the lane's NIT store positions of a tile: (row, chunk) = ((it * 64 + lane) / 8, lane % 8) of wave row wm
    Vec16<T> ycur[NIT], ynext[NIT];
    auto load_y = [&](int tile_) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int m = min(tile_ * D1_BM + wm * WROWS + (it * 64 + lane) / WCPR, M - 1);
            if (!(ABL & 2)) ynext[it] = ld16(reinterpret_cast<const T *>(bz.Y) + (long)m * Nf + nb);
            else ynext[it] = zero16<T>();
        }
    };
    if (T_ > 0) load_y(tile);

    // ---- fragment read addresses: dY rows 32 wm + l31 (buffer 0), filter rows 64 wn + 32 j + l31; the 16-k group kk is XORed in
    const unsigned lds0 = y2_lds_addr(smem);
    const int arow = wm * 32 + l31, brow = wn * 64 + l31;
    const unsigned a0 = lds0 + (unsigned)(FBYTES + arow * ROWB + ((half ^ swz(arow)) << 4));
    const unsigned b0 = lds0 + (unsigned)(brow * ROWB + ((half ^ swz(brow)) << 4));      // (swz(brow + 32) == swz(brow))
    unsigned char *const wreg = img + wave * (WROWS * WSTRIDE);

    for (int t = 0; t < T_; ++t) {
        tile = (int)blockIdx.x + t * (int)gridDim.x;
        const int buf = t & 1;
        // this wave's pieces of tile t (and the filter, t = 0), the y vectors of tile t, the stores of tile t - 1: everything older has completed
        // (loads and stores share the counter: a counted wait over a mix of the two is not safe)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < NIT; ++it) ycur[it] = ynext[it];
        __syncthreads();                                   // every wave's pieces have landed; every wave has left the other buffer
        if (t + 1 < T_) {
            stage(tile + (int)gridDim.x, buf ^ 1);
            load_y(tile + (int)gridDim.x);
        }
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        const unsigned boff = buf ? (unsigned)ABYTES : 0u;
#pragma unroll
        for (int kk = 0; kk < KCH; ++kk) {
            const bf16x8 fa = *(lds_frag_ptr)(uintptr_t)((a0 ^ (unsigned)(kk * 32)) + boff);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bf16x8 fb = *(lds_frag_ptr)(uintptr_t)((b0 ^ (unsigned)(kk * 32)) + (unsigned)(j * 32 * ROWB));
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[j], 0, 0, 0);
            }
        }
        // ---- round the 32 x 64 sub-tile into this wave's LDS image (LDS operations of one wave execute in order: its reads below see these writes)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 4 * half + (r & 3) + 8 * (r >> 2);
                *reinterpret_cast<T *>(wreg + row * WSTRIDE + (j * 32 + l31) * 2) = (T)acc[j][r];
            }
        // ---- 16-byte stores of the image rows + the BN / leaky backward sums of the rounded gradient (bn_bwd_reduce_kernel's arithmetic)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int id = it * 64 + lane;
            const int row = id / WCPR, ch = id % WCPR;
            const int m = tile * D1_BM + wm * WROWS + row;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(wreg + row * WSTRIDE + ch * 16);
            if (m < M) {
                if (!(ABL & 1)) *reinterpret_cast<f32x4 *>(O + (long)m * Nf + nb) = v;
                const Vec16<T> y = ycur[it];
                Vec16<T> d;
                d.v = __builtin_bit_cast(decltype(d.v), v);
                if (ABL & 4) { ps[0][it] += d.get(0) + y.get(1); continue; }
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    const float xh = (y.get(k) - cmu[k]) * cinv[k];
                    const float z = (y.get(k) - cmu[k]) * (cinv[k] * cga[k]) + cbt[k];
                    const float g = z >= 0.f ? d.get(k) : bz.alpha * d.get(k);
                    ps[0][k] += g * xh;
                    ps[1][k] += g;
                }
            }
        }
    }
    // ---- the lanes that share a channel chunk meet (lane % 8 fixed: offsets 8, 16, 32), then one atomic add per value
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        ps[0][k] = y2_lane_group_sum<WCPR>(ps[0][k]);
        ps[1][k] = y2_lane_group_sum<WCPR>(ps[1][k]);
    }
    // ... and the four wave rows of the workgroup (the same channels per column half) in LDS: a quarter of the atomic adds
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                       // every wave is done with the operand buffers
    float *const red = reinterpret_cast<float *>(ab);
    if (lane < WCPR) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) { red[(wave * WCPR + lane) * 2 * VEC + k] = ps[0][k]; red[(wave * WCPR + lane) * 2 * VEC + VEC + k] = ps[1][k]; }
    }
    __syncthreads();
    if (wm == 0 && lane < WCPR) {
#pragma unroll
        for (int r = 1; r < 4; ++r)
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                ps[0][k] += red[((2 * r + wn) * WCPR + lane) * 2 * VEC + k];
                ps[1][k] += red[((2 * r + wn) * WCPR + lane) * 2 * VEC + VEC + k];
            }
        const int slot = (int)(blockIdx.x & (D1_STAT_ROWS - 1));
        float *p1 = bn_part + (long)slot * Nf + nb, *p2 = bn_part + (long)(Y2_BN_PART_ROWS + slot) * Nf + nb;
#pragma unroll
        for (int k = 0; k < VEC; ++k) { unsafeAtomicAdd(p1 + k, ps[0][k]); unsafeAtomicAdd(p2 + k, ps[1][k]); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// the shapes this kernel takes: 1x1 data gradient, bf16, 64 or 128 input channels (of the gradient), a multiple of 128 output channels stored unpadded,
// enough pixels for the persistent form to matter
bool y2_d1_shape(int Cp, int ldp, int Nf, int ldo, int ksize, int dtype, long M) {
    return dtype == YOLO2_BF16 && ksize == 1 && (Cp == 64 || Cp == 128) && ldp == Cp && Nf % D1_BN == 0 && ldo == Nf && M >= 8192;
}
// -> 0 launched (*rows = partial rows touched), 1 not taken
int y2_d1_dgrad_bn(const void *P, const void *F, void *O, long M, int Cp, int Nf, float *bn_part, const Y2BnBwd &bz, int cus, int *rows, hipStream_t st) {
    if (M >= (1L << 23) || !bn_part || !bz.Y) return 1;
    const int ntiles = (int)((M + D1_BM - 1) / D1_BM), ngroups = Nf / D1_BN;
    int gx = cus / ngroups;
    if (gx < 1) gx = 1;
    if (gx > ntiles) gx = ntiles;
    const unsigned p_bytes = (unsigned)((size_t)M * Cp * 2), f_bytes = (unsigned)((size_t)Nf * Cp * 2);
    const dim3 grid(gx, ngroups);
#define D1_LAUNCH(KCHv, ABLv) conv_d1_dgrad_bn_kernel<KCHv, ABLv><<<grid, 512, 0, st>>>((const bf16 *)P, p_bytes, (const bf16 *)F, f_bytes, (bf16 *)O, (int)M, Nf, ntiles, bn_part, bz)
#ifdef Y2D1_EXPERIMENTS
    static const int abl = y2_env_int("YOLO2_D1_ABL", 0);      // timing ablations (wrong results): 1 no stores, 2 no y loads, 4 no sums arithmetic
    if (Cp == 64) {
        switch (abl) { case 1: D1_LAUNCH(4, 1); break; case 2: D1_LAUNCH(4, 2); break; case 4: D1_LAUNCH(4, 4); break; case 3: D1_LAUNCH(4, 3); break; case 7: D1_LAUNCH(4, 7); break; default: D1_LAUNCH(4, 0); }
    } else D1_LAUNCH(8, 0);
#else
    if (Cp == 64) D1_LAUNCH(4, 0); else D1_LAUNCH(8, 0);
#endif
#undef D1_LAUNCH
    if (rows) *rows = gx < D1_STAT_ROWS ? gx : D1_STAT_ROWS;
    return 0;
}
