// Direct 3x3 convolution for the FIRST layer (image input: 3 channels in an 8-wide pixel stride,
// 32 filters): forward and filter gradient, gfx950.
//
// conv0 is 0.9 % of the network's MACs but 5.5 M output pixels per image: it is HBM-bound (write 177 MB of
// outputs / read them back as dY at batch 16), and the generic implicit-GEMM kernels waste it -- the forward
// pads every tap's 8 channels to a 32-wide K step, the filter gradient re-reads dY once per tap (9x).
// Here one WAVE owns one segment of 32 consecutive output pixels of an image row:
//   * the 3 x 34-pixel input halo (16 B per pixel) and, for the gradient, the 32 x 32 dY block are DMA-ed
//     into a wave-private LDS slot (buffer_load ... lds; out-of-image rows / pixels and segment tails are
//     out-of-range offsets -> zeros), double buffered, no workgroup barrier in the loop;
//   * forward: all 9 taps x 8 channels form ONE reduction of length 72 (padded to 80 = 5 bf16 MFMA steps):
//     the A fragment of step s for pixel i is simply the 16 contiguous bytes of halo pixel (i + dw) in halo
//     row dh of tap 2s + (lane>>5); the filters (32 x 80) live in registers;
//   * filter gradient: rows = (tap, channel) = 72 (3 MFMA row tiles), cols = 32 filters, reduction = the
//     segment's 32 pixels; fragments by ds_read_b64_tr_b16 with per-lane tap-shifted addresses; each wave
//     keeps 3 accumulator tiles over all its segments, waves are combined through LDS and written with one
//     f32 atomic per output element per workgroup.
#include "common.h"
#include <type_traits>

#define Y2_OOB 0x80000000u

// ---------------------------------------------------------------------------------------------------
// shared staging: halo rows h-1, h, h+1, pixels w0-1 .. w0+32 of image b, 8 channels per pixel
// ---------------------------------------------------------------------------------------------------
template <int N> struct IdxPackFirst;                     // N arg-max code bytes as one integer
template <> struct IdxPackFirst<8> { typedef unsigned long long type; };
template <> struct IdxPackFirst<4> { typedef unsigned type; };
template <typename T> struct First {
    static constexpr int PXB = 8 * sizeof(T);              // bytes per halo pixel (16 bf16 / 32 f32)
    static constexpr int HPIECES = PXB / 16;               // DMA pieces per halo row (64 lanes x 16 B = 1 KiB)
    static constexpr int HROWB = HPIECES * 1024;           // LDS bytes reserved per halo row
    static constexpr int HALO = 3 * HROWB;
};

template <typename T>
__device__ __forceinline__ void stage_halo(const __amdgpu_buffer_rsrc_t &rsrcX, unsigned char *dst, int b, int h, int w0, int H, int W, int lane) {
    constexpr int PXB = First<T>::PXB, HP = First<T>::HPIECES;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int hh = h + r - 1;
#pragma unroll
        for (int p = 0; p < HP; ++p) {
            const int chunk = p * 64 + lane;                   // 16-byte chunk index inside the halo row
            const int px = chunk / HP;                         // halo pixel 0..33 (beyond: unused)
            const int ww = w0 - 1 + px;
            const bool ok = px < 34 && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W;
            const unsigned voff = ok ? (unsigned)((((long)b * H + hh) * W + ww) * PXB + (chunk % HP) * 16) : Y2_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (__attribute__((address_space(3))) void *)(dst + r * First<T>::HROWB + p * 1024), 16, voff, 0, 0, 0);
        }
    }
}

// this wave's first unit, its stride and the end of its XCD's band of units (grids that are a multiple of 8 blocks; else one band = all units)
__device__ __forceinline__ void y2_first_band(int units, int wave, int &stride, int &u, int &uend) {
    if ((gridDim.x & 7) == 0) {
        const int nbx = gridDim.x >> 3, xcd = blockIdx.x & 7, lb = blockIdx.x >> 3;
        const int band = (units + 7) >> 3;
        const int beg = xcd * band;
        uend = min(units, beg + band);
        u = beg + lb * 4 + wave;
        stride = nbx * 4;
    } else {
        uend = units;
        u = blockIdx.x * 4 + wave;
        stride = gridDim.x * 4;
    }
}

// ---------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void conv_first_fwd_kernel(const T *__restrict__ X, unsigned x_bytes, const T *__restrict__ F, T *__restrict__ Y,
                                                             int B, int H, int W, int units, const float *__restrict__ bn_shift,
                                                             float *__restrict__ bn_part) {
    constexpr int SLOT = First<T>::HALO, PXB = First<T>::PXB;
    constexpr int LOADS = 3 * First<T>::HPIECES;
    constexpr int KS = sizeof(T) == 2 ? 5 : 36;              // MFMA steps over the 72 (80) reduction values
    __shared__ __attribute__((aligned(1024))) unsigned char smem[4 * 2 * SLOT];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned char *my = smem + wave * 2 * SLOT;
    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(X), 0, x_bytes, 0x00020000);
    const int SW = (W + 31) / 32;

    // filters of column n = lane&31 in registers: F is [32][72] (K-contiguous), zero beyond 72
    typename std::conditional<sizeof(T) == 2, bf16x8, float>::type bf[KS];
    const int n = lane & 31, half = lane >> 5;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = s * 16 + half * 8 + e;
                bf[s][e] = k < 72 ? F[n * 72 + k] : (bf16)0.f;
            }
        } else {
            const int k = s * 2 + half;
            bf[s] = F[n * 72 + k];
        }
    }

    // optional batch-norm partial sums of the stored outputs (same contract as conv_igemm.hip's epilogue)
    const float sh = bn_part ? bn_shift[n] : 0.f;
    float s1 = 0.f, s2 = 0.f;

    // Units (32-pixel row segments) are dealt to the XCDs in contiguous BANDS (block b runs on XCD b % 8: observed, speed only): the three halo
    // rows of a segment are the rows of the segments 13 units before and after it, and with units strided over the whole grid those ran on
    // other XCDs at the same time -- every XCD's L2 fetched every image row about three times (148 MB for the 44 MB image,
    // profiles/r04_hbm_traffic_pmc.md).  Inside a band they are L2 hits.
    int stride, u, uend;
    y2_first_band(units, wave, stride, u, uend);
    if (u >= uend) return;
    auto decode = [&](int uu, int &b, int &h, int &w0) { w0 = (uu % SW) * 32; int t = uu / SW; h = t % H; b = t / H; };
    int b, h, w0;
    decode(u, b, h, w0);
    stage_halo<T>(rsrcX, my, b, h, w0, H, W, lane);
    int st = 0;
    for (; u < uend; u += stride) {
        const int un = u + stride;
        int nb = 0, nh = 0, nw0 = 0;
        if (un < uend) {
            decode(un, nb, nh, nw0);
            stage_halo<T>(rsrcX, my + (st ^ 1) * SLOT, nb, nh, nw0, H, W, lane);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");   // everything older than the just-issued DMA has landed
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned char *hs = my + st * SLOT;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const int i = lane & 31;                              // output pixel of this lane's A row
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if constexpr (sizeof(T) == 2) {
                int tap = 2 * s + half;
                if (tap > 8) tap = 8;                         // padding step: filters are zero there, data must merely be finite
                const bf16x8 a = *reinterpret_cast<const bf16x8 *>(hs + (tap / 3) * First<T>::HROWB + (i + tap % 3) * PXB);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bf[s], acc, 0, 0, 0);
            } else {
                const int k = 2 * s + half, tap = k >> 3, c = k & 7;
                const float a = *reinterpret_cast<const float *>(hs + (tap / 3) * First<T>::HROWB + (i + tap % 3) * PXB + c * 4);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bf[s], acc, 0, 0, 0);
            }
        }
        // D: col = lane&31 = filter, row = (r&3) + 8*(r>>2) + 4*(lane>>5) = pixel of the segment
        T *out = Y + (((long)b * H + h) * W + w0) * 32 + n;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int px = (r & 3) + 8 * (r >> 2) + 4 * half;
            if (w0 + px < W) {
                const T o = (T)acc[r];
                if (Y) out[(long)px * 32] = o;               // Y == NULL: statistics only (the output is recomputed by its consumers: fused kernels below)
                const float d = (float)o - sh;
                s1 += d;
                s2 += d * d;
            }
        }
        b = nb; h = nh; w0 = nw0;
        st ^= 1;
    }
    if (bn_part) {
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        if (lane < 32) {
            const int slot = (blockIdx.x * 4 + wave) & (Y2_BN_PART_ROWS - 1);
            unsafeAtomicAdd(bn_part + slot * 32 + n, s1);
            unsafeAtomicAdd(bn_part + (Y2_BN_PART_ROWS + slot) * 32 + n, s2);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Image layer fused with its consumers (round 3): the raw output of conv0 is 5.5 M pixels x 32 channels per image -- 177 MB at batch 16,
// 2.8 GB at batch 256 -- written once and read back three times (BN + leaky + pool forward, BN-backward reduce, BN-backward apply) while
// its input is 1/4 of that (16 B per pixel) and its arithmetic 0.9 % of the network.  These kernels never store it: every consumer
// RE-COMPUTES its 2 x 32-pixel patch from a 4 x 34-pixel input halo (ten MFMAs) and applies its own elementwise work to the accumulators:
//   MODE 0  forward:  y -> BN -> leaky -> 2x2 max pool -> P (+ first-max index), as yolo2_bn_leaky_pool on a stored y
//   MODE 1  backward: sum(g * xhat), sum(g) per channel into the partial rows (g = dP * leaky'(z) at the arg-max), as ..._pool_bwd_reduce
//   MODE 2  backward: dY (full resolution, for the filter gradient), as ..._pool_bwd_apply
// The recomputed y is rounded to T exactly where the unfused path stored it, so every result equals the unfused path's (same MFMA order,
// same rounding points); batch statistics come from a statistics-only pass of conv_first_fwd_kernel (Y = NULL).
// A wave owns one pooled row segment: output rows 2*h2, 2*h2+1, pixels w0 .. w0+31.  The 32x32 accumulator layout puts filter n in lane
// n (both halves) and pixel (r&3) + 8*(r>>2) + 4*half in register r: a pooling window is registers {r0, r0+1} of the two rows' tiles in
// ONE lane, so the pool, its arg-max and the routing of dP need no cross-lane traffic at all.
// ---------------------------------------------------------------------------------------------------
struct Y2FirstBn { const float *mean, *var, *gamma, *beta, *dgamma, *dbeta; float eps, alpha; };

template <typename T, int MODE>
__global__ __launch_bounds__(256) void conv_first_pool_kernel(const T *__restrict__ X, unsigned x_bytes, const T *__restrict__ F, T *__restrict__ Out,
                                                              unsigned char *__restrict__ idx, const T *__restrict__ dP, int lddp, float *__restrict__ part,
                                                              int B, int H, int W, int ldo, int units, const Y2FirstBn bn) {
    constexpr int PXB = First<T>::PXB, HROWB = First<T>::HROWB, SLOT = 4 * HROWB;
    constexpr int LOADS = 4 * First<T>::HPIECES;
    constexpr int KS = sizeof(T) == 2 ? 5 : 36;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[4 * 2 * SLOT];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned char *my = smem + wave * 2 * SLOT;
    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(X), 0, x_bytes, 0x00020000);
    const int SW = (W + 31) / 32, OH = H / 2, OW = W / 2;

    typename std::conditional<sizeof(T) == 2, bf16x8, float>::type bf[KS];
    const int n = lane & 31, half = lane >> 5;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = s * 16 + half * 8 + e;
                bf[s][e] = k < 72 ? F[n * 72 + k] : (bf16)0.f;
            }
        } else {
            bf[s] = F[n * 72 + s * 2 + half];
        }
    }
    // this lane's channel constants (one filter per lane)
    const float mu = bn.mean[n], inv = 1.0f / sqrtf(bn.var[n] + bn.eps), ga = bn.gamma[n], bt = bn.beta[n];
    const float sc = inv * ga;
    const float invM = 1.0f / (float)((long)B * H * W);
    const float dgm = MODE == 2 ? bn.dgamma[n] * invM : 0.f, dbm = MODE == 2 ? bn.dbeta[n] * invM : 0.f;
    float s0 = 0.f, s1 = 0.f;

    int stride, u, uend;
    y2_first_band(units, wave, stride, u, uend);      // contiguous unit bands per XCD (see conv_first_fwd_kernel)
    if (u >= uend) return;
    auto decode = [&](int uu, int &b, int &h2, int &w0) { w0 = (uu % SW) * 32; int t = uu / SW; h2 = t % OH; b = t / OH; };
    auto stage = [&](unsigned char *dst, int b, int h2, int w0) {      // halo rows 2*h2-1 .. 2*h2+2, pixels w0-1 .. w0+32
        constexpr int HP = First<T>::HPIECES;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int hh = 2 * h2 - 1 + r;
#pragma unroll
            for (int p = 0; p < HP; ++p) {
                const int chunk = p * 64 + lane;
                const int px = chunk / HP;
                const int ww = w0 - 1 + px;
                const bool ok = px < 34 && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W;
                const unsigned voff = ok ? (unsigned)((((long)b * H + hh) * W + ww) * PXB + (chunk % HP) * 16) : Y2_OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (__attribute__((address_space(3))) void *)(dst + r * HROWB + p * 1024), 16, voff, 0, 0, 0);
            }
        }
    };
    int b, h2, w0;
    decode(u, b, h2, w0);
    stage(my, b, h2, w0);
    int st = 0;
    for (; u < uend; u += stride) {
        const int un = u + stride;
        int nb = 0, nh2 = 0, nw0 = 0;
        if (un < uend) {
            decode(un, nb, nh2, nw0);
            stage(my + (st ^ 1) * SLOT, nb, nh2, nw0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned char *hs = my + st * SLOT;
        f32x16 acc[2];
        const int i = lane & 31;
#pragma unroll
        for (int rho = 0; rho < 2; ++rho) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rho][r] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if constexpr (sizeof(T) == 2) {
                    int tap = 2 * s + half;
                    if (tap > 8) tap = 8;
                    const bf16x8 a = *reinterpret_cast<const bf16x8 *>(hs + (rho + tap / 3) * HROWB + (i + tap % 3) * PXB);
                    acc[rho] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bf[s], acc[rho], 0, 0, 0);
                } else {
                    const int k = 2 * s + half, tap = k >> 3, c = k & 7;
                    const float a = *reinterpret_cast<const float *>(hs + (rho + tap / 3) * HROWB + (i + tap % 3) * PXB + c * 4);
                    acc[rho] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bf[s], acc[rho], 0, 0, 0);
                }
            }
        }
        // y as the unfused path stores it (rounded to T), window by window: registers r0, r0+1 of both rows = the 2x2 window of pooled
        // pixel p = 4q + 2*half + j of this segment
        const long prow = ((long)b * OH + h2) * OW + (w0 >> 1);       // first pooled pixel of the segment
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r0 = 4 * q + 2 * j, p = 4 * q + 2 * half + j;
                if (w0 + 2 * p >= W) continue;
                float y[4] = {(float)(T)acc[0][r0], (float)(T)acc[0][r0 + 1], (float)(T)acc[1][r0], (float)(T)acc[1][r0 + 1]};
                if (MODE == 0) {
                    float a[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float z = (y[k] - mu) * sc + bt;
                        a[k] = (float)(T)fmaxf(z, bn.alpha * z);
                    }
                    const float m = fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3]));
                    const int arg = a[0] == m ? 0 : a[1] == m ? 1 : a[2] == m ? 2 : 3;
                    Out[(prow + p) * ldo + n] = (T)m;
                    if (idx) idx[(prow + p) * 32 + n] = (unsigned char)arg;
                } else {
                    const float d = (float)dP[(prow + p) * lddp + n];
                    const int k = (int)idx[(prow + p) * 32 + n];
                    if (MODE == 1) {
                        const float yk = k == 0 ? y[0] : k == 1 ? y[1] : k == 2 ? y[2] : y[3];
                        const float xh = (yk - mu) * inv;
                        const float z = (yk - mu) * (inv * ga) + bt;
                        const float g = z >= 0.f ? d : bn.alpha * d;
                        s0 += g * xh;
                        s1 += g;
                    } else {
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) {
                            const float da = k == kk ? d : 0.f;
                            const float xh = (y[kk] - mu) * inv;
                            const float z = (y[kk] - mu) * (inv * ga) + bt;
                            const float g = z >= 0.f ? da : bn.alpha * da;
                            const long pix = ((long)b * H + 2 * h2 + (kk >> 1)) * W + w0 + 2 * p + (kk & 1);
                            Out[pix * 32 + n] = (T)((ga * inv) * (g - dbm - xh * dgm));
                        }
                    }
                }
            }
        b = nb; h2 = nh2; w0 = nw0;
        st ^= 1;
    }
    if (MODE == 1) {
        s0 += __shfl_xor(s0, 32, 64);
        s1 += __shfl_xor(s1, 32, 64);
        if (lane < 32) {
            const int slot = (blockIdx.x * 4 + wave) & (Y2_BN_PART_ROWS - 1);
            unsafeAtomicAdd(part + slot * 32 + n, s0);
            unsafeAtomicAdd(part + (Y2_BN_PART_ROWS + slot) * 32 + n, s1);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// filter gradient
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void conv_first_wgrad_kernel(const T *__restrict__ X, unsigned x_bytes, const T *__restrict__ dY, unsigned y_bytes,
                                                               float *__restrict__ dW, int B, int H, int W, int Cin, int units) {
    constexpr int PXB = First<T>::PXB, HALO = First<T>::HALO;
    constexpr int YROWB = 32 * sizeof(T);                    // dY bytes per pixel (64 / 128)
    constexpr int YPIECES = 32 * YROWB / 1024;               // 2 / 4
    constexpr int SLOT = HALO + YPIECES * 1024;
    constexpr int LOADS = 3 * First<T>::HPIECES + YPIECES;
    constexpr int RED = 4 * 3 * 16 * 64 * 4;                 // cross-wave reduction scratch (reuses the staging area)
    constexpr int SMEM = 4 * 2 * SLOT > RED ? 4 * 2 * SLOT : RED;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned char *my = smem + wave * 2 * SLOT;
    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(X), 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcY = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(dY), 0, y_bytes, 0x00020000);
    const int SW = (W + 31) / 32;

    auto stage = [&](unsigned char *dst, int b, int h, int w0) {
        stage_halo<T>(rsrcX, dst, b, h, w0, H, W, lane);
#pragma unroll
        for (int p = 0; p < YPIECES; ++p) {
            const int chunk = p * 64 + lane;
            const int px = chunk / (YROWB / 16);
            const bool ok = w0 + px < W;
            const unsigned voff = ok ? (unsigned)((((long)b * H + h) * W + w0 + px) * YROWB + (chunk % (YROWB / 16)) * 16) : Y2_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcY, (__attribute__((address_space(3))) void *)(dst + HALO + p * 1024), 16, voff, 0, 0, 0);
        }
    };

    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    int stride, u, uend;
    y2_first_band(units, wave, stride, u, uend);      // contiguous unit bands per XCD (see conv_first_fwd_kernel)
    auto decode = [&](int uu, int &b, int &h, int &w0) { w0 = (uu % SW) * 32; int t = uu / SW; h = t % H; b = t / H; };
    if (u < uend) {
        int b, h, w0;
        decode(u, b, h, w0);
        stage(my, b, h, w0);
    }
    const int g = lane >> 4, t16 = lane & 15;
    int st = 0;
    for (; u < uend; u += stride) {
        const int un = u + stride;
        if (un < uend) {
            int nb, nh, nw0;
            decode(un, nb, nh, nw0);
            stage(my + (st ^ 1) * SLOT, nb, nh, nw0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned char *hs = my + st * SLOT;
        const unsigned char *ys = hs + HALO;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 bfrag, afrag[3];
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int px = ks * 16 + 8 * (g >> 1) + 4 * r + (t16 >> 2);     // pixel row this lane addresses
                    {
                        const unsigned char *p = ys + px * YROWB + (16 * (g & 1) + 4 * (t16 & 3)) * 2;
                        s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p);
                        bf16x4 q = __builtin_bit_cast(bf16x4, v);
                        bfrag[4 * r] = q[0]; bfrag[4 * r + 1] = q[1]; bfrag[4 * r + 2] = q[2]; bfrag[4 * r + 3] = q[3];
                    }
#pragma unroll
                    for (int rt = 0; rt < 3; ++rt) {
                        const int row = 32 * rt + 16 * (g & 1) + 4 * (t16 & 3);       // first of the 4 (tap, channel) rows this lane addresses
                        int tap = row >> 3;
                        if (tap > 8) tap = 8;                                          // rows 72..95: discarded at the end
                        const unsigned char *p = hs + (tap / 3) * First<T>::HROWB + (px + tap % 3) * PXB + (row & 7) * 2;
                        s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p);
                        bf16x4 q = __builtin_bit_cast(bf16x4, v);
                        afrag[rt][4 * r] = q[0]; afrag[rt][4 * r + 1] = q[1]; afrag[rt][4 * r + 2] = q[2]; afrag[rt][4 * r + 3] = q[3];
                    }
                }
#pragma unroll
                for (int rt = 0; rt < 3; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[rt], bfrag, acc[rt], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const int px = ks * 2 + (lane >> 5);
                const float bv = *reinterpret_cast<const float *>(ys + px * YROWB + (lane & 31) * 4);
#pragma unroll
                for (int rt = 0; rt < 3; ++rt) {
                    const int row = 32 * rt + (lane & 31);
                    int tap = row >> 3;
                    if (tap > 8) tap = 8;
                    const float av = *reinterpret_cast<const float *>(hs + (tap / 3) * First<T>::HROWB + (px + tap % 3) * PXB + (row & 7) * 4);
                    acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[rt], 0, 0, 0);
                }
            }
        }
        st ^= 1;
    }

    // combine the 4 waves through LDS (the staging slots are dead now), then one atomic per element per workgroup
    __syncthreads();
    float *red = reinterpret_cast<float *>(smem);            // [4 waves][3 tiles][16 regs][64 lanes] = 48 KiB
#pragma unroll
    for (int rt = 0; rt < 3; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave * 3 + rt) * 16 + r) * 64 + lane] = acc[rt][r];
    __syncthreads();
    if (wave == 0) {
        const int n = lane & 31;
#pragma unroll
        for (int rt = 0; rt < 3; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * rt + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);    // (tap, channel)
                const int tap = row >> 3, c = row & 7;
                if (tap < 9 && c < Cin) {
                    float v = 0.f;
#pragma unroll
                    for (int w = 0; w < 4; ++w) v += red[((w * 3 + rt) * 16 + r) * 64 + lane];
#ifdef Y2FIRST_ABL_NOATOMIC
                    if (v == 123.456f) dW[0] = v;
#else
                    unsafeAtomicAdd(dW + ((long)tap * Cin + c) * 32 + n, v);
#endif
                }
            }
    }
}

#ifdef Y2FIRST_EXPERIMENTS
#define Y2F_ABL(bit) (abl & (bit))      // timing ablations (wrong results): 1 no BN arithmetic, 2 no transpose reads / MFMAs, 4 no y / dP / code loads, 8 no halo DMA
#else
#define Y2F_ABL(bit) 0
#endif
// ---------------------------------------------------------------------------------------------------
// filter gradient with the image layer's BN / leaky / pool backward applied on the way in (round 6)
// ---------------------------------------------------------------------------------------------------
// conv0's gradient tensor dY (416 x 416 x 32: 177 MB at batch 16) was written by bn_bwd_apply_fin_kernel<T, true> (72 us) for exactly one reader, the
// kernel above (56 us).  Here it never leaves the CU: a wave stages, next to its input halo, the RAW forward output y of its 32-pixel row segment and the
// 16 pooled gradients + arg-max bytes over it (LDS-DMA), turns y into dy IN PLACE in LDS -- the arithmetic of bn_bwd_apply_fin_kernel, element for
// element, rounded to T like the stored tensor was -- and runs the same transpose-read MFMA loop on it.  The workgroup's prologue folds the partial
// rows of the reduction pass into dgamma / dbeta like bn_bwd_apply_fin_kernel's (workgroup 0 writes them).  Replaces reference train.py:70-80's
// gradient of model/yolo2/inference.py:62-66 (conv0's slim.batch_norm + leaky_relu + max_pool2d) w.r.t. the filter: 405 + 221 MB of traffic -> 270 MB.
template <typename T>
__global__ __launch_bounds__(256, 3) void conv_first_wgrad_bn_kernel(const T *__restrict__ X, unsigned x_bytes, const T *__restrict__ Y, const T *__restrict__ dP, int lddp,
                                                                  const unsigned char *__restrict__ idx, const float *__restrict__ mean, const float *__restrict__ var,
                                                                  const float *__restrict__ gamma, const float *__restrict__ beta, const float *__restrict__ part, int rows,
                                                                  long plane, float *__restrict__ dgamma, float *__restrict__ dbeta, float *__restrict__ dW, int B, int H, int W,
                                                                  int Cin, int units, float eps, float alpha, float *__restrict__ zero, long zero_vec4, int abl) {
    constexpr int PXB = First<T>::PXB, HALO = First<T>::HALO;
    constexpr int N = 16 / (int)sizeof(T);                   // values per 16-byte chunk (8 / 4)
    constexpr int YROWB = 32 * sizeof(T);                    // y / dy bytes per pixel (64 / 128)
    constexpr int CH = YROWB / 16;                           // chunks per pixel (4 / 8)
    constexpr int NIT = 32 * CH / 64;                        // (pixel, chunk) pairs per lane and segment (2 / 4)
    constexpr int DYB = 32 * YROWB;                          // the segment's dy image (2 / 4 KB), one per wave: written and read by that wave only
    constexpr int SLOT = 2 * HALO + DYB;                     // two input halos (the next segment's streams in under this one's arithmetic) + dy
    constexpr int RED = 4 * 3 * 16 * 64 * 4;                 // cross-wave reduction scratch (reuses the staging area)
    constexpr int SMEM = 4 * SLOT > RED ? 4 * SLOT : RED;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM];
    __shared__ float cst[2][32];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned char *my = smem + wave * SLOT;
    unsigned char *const ys = my + 2 * HALO;
    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(X), 0, x_bytes, 0x00020000);
    const int SW = (W + 31) / 32, OH = H >> 1, OW = W >> 1;
    const int chunk = lane % CH;                             // 64 % CH == 0: a lane keeps its channel chunk in every iteration
    typedef typename IdxPackFirst<N>::type pack_t;

    // y, pooled gradient and arg-max codes of a segment go straight to registers, one segment ahead; only the halo and dy need LDS (8 KB per wave)
    struct Pre { Vec16<T> y[NIT], d[NIT]; pack_t p[NIT]; };
    auto fetch = [&](Pre &r, int b, int h, int w0) {
        if (Y2F_ABL(4)) return;
        const long row = ((long)b * H + h) * W + w0, prow = ((long)b * OH + (h >> 1)) * OW + (w0 >> 1);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int px = (it * 64 + lane) / CH;
            const bool ok = w0 + px < W;
            const long m = ok ? row + px : row, pm = ok ? prow + (px >> 1) : prow;        // (clamped: the values of a dead pixel are never used)
            r.y[it] = ld16(Y + m * 32 + chunk * N);
            r.d[it] = ld16(dP + pm * lddp + chunk * N);
            r.p[it] = *reinterpret_cast<const pack_t *>(idx + pm * 32 + chunk * N);
        }
    };

    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    int stride, u, uend;
    y2_first_band(units, wave, stride, u, uend);      // contiguous unit bands per XCD (see conv_first_fwd_kernel)
    auto decode = [&](int uu, int &b, int &h, int &w0) { w0 = (uu % SW) * 32; int t = uu / SW; h = t % H; b = t / H; };
    int cb = 0, ch = 0, cw0 = 0;
    Pre cur, nxt;
    if (u < uend) {
        decode(u, cb, ch, cw0);
        if (!Y2F_ABL(8)) stage_halo<T>(rsrcX, my, cb, ch, cw0, H, W, lane);       // in flight under the prologue
        fetch(cur, cb, ch, cw0);
    }

    // ---- dgamma / dbeta from the partial rows [2][rows][32] (f64 like slice_partial_sums): thread = (plane, 16 row groups, four channels): 16-byte
    // loads, rows / 16 independent ones per thread (one thread per (plane, channel, row quarter) walked 128 rows one 4-byte load at a time: 30 us of
    // prologue in front of every workgroup)
    {
        __shared__ double psum[2][16][32];
        const int pl = threadIdx.x >> 7, rg = (threadIdx.x >> 3) & 15, cq = threadIdx.x & 7;
        double a[4] = {0.0, 0.0, 0.0, 0.0};
        const float *pp = part + (long)pl * plane + cq * 4;
#pragma unroll 4
        for (int r = rg; r < rows; r += 16) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(pp + (long)r * 32);
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] += (double)v[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) psum[pl][rg][cq * 4 + j] = a[j];
        __syncthreads();
        if (threadIdx.x < 64) {
            const int pl2 = threadIdx.x >> 5, c2 = threadIdx.x & 31;
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < 16; ++k) t += psum[pl2][k][c2];
            const float v = (float)t;
            cst[pl2][c2] = v;
            if (blockIdx.x == 0) (pl2 ? dbeta : dgamma)[c2] = v;
        }
        __syncthreads();
    }
    if (zero) {
        const long nthreads = (long)gridDim.x * 256;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < zero_vec4; i += nthreads) reinterpret_cast<f32x4 *>(zero)[i] = z;
    }
    // per-channel constants of this lane's chunk.  bn_bwd_apply_fin_kernel computes dy = (ga inv) (g - dbeta / M - xh dgamma / M) with xh = (y - mu) inv,
    // z = (y - mu)(inv ga) + bt, g = z >= 0 ? da : alpha da; here the same value as two multiply-adds on four constants per channel (32 registers instead
    // of 48 -- the difference between two and three waves per SIMD):  z = y A + o,  dy = da (z >= 0 ? A : alpha A) + (y Bc + Cc)
    const float invM = 1.0f / (float)((long)B * H * W);
    float cA[N], cAa[N], cO[N], cB[N], cC[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const int c = chunk * N + j;
        const float mu = mean[c], inv = 1.0f / sqrtf(var[c] + eps);
        cA[j] = gamma[c] * inv;
        cAa[j] = alpha * cA[j];
        cO[j] = beta[c] - mu * cA[j];
        cB[j] = -cA[j] * inv * (cst[0][c] * invM);
        cC[j] = -cA[j] * (cst[1][c] * invM) - cB[j] * mu;
    }

    const int g = lane >> 4, t16 = lane & 15;
    int st = 0;
    for (; u < uend; u += stride) {
        // everything issued an iteration ago has landed: this segment's halo (LDS-DMA) and registers.  (One wait for all of it: a counted wait would
        // have to rely on the order hipcc leaves the register loads and the DMA in.)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int un = u + stride;
        int nb = 0, nh = 0, nw0 = 0;
        if (un < uend) {
            decode(un, nb, nh, nw0);
            if (!Y2F_ABL(8)) stage_halo<T>(rsrcX, my + (st ^ 1) * HALO, nb, nh, nw0, H, W, lane);
            fetch(nxt, nb, nh, nw0);
        }
        const unsigned char *hs = my + st * HALO;
        // ---- y -> dy (pixels beyond the row's end get zeros), rounded to T into this wave's dy image
        {
            const int rp2 = 2 * (ch & 1);
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int px = (it * 64 + lane) / CH;
                const int kk = rp2 + (px & 1);
                const bool live = cw0 + px < W;
                Vec16<T> o;
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const float yv = cur.y[it].get(j);
                    const bool sel = (int)((cur.p[it] >> (8 * j)) & 3) == kk;          // this pixel is its window's arg-max in channel j
                    const float z = yv * cA[j] + cO[j];
                    float coef = z >= 0.f ? cA[j] : cAa[j];
                    coef = sel ? coef : 0.f;
                    const float v = cur.d[it].get(j) * coef + (yv * cB[j] + cC[j]);
                    o.set(j, live ? (Y2F_ABL(1) ? yv : v) : 0.f);
                }
                *reinterpret_cast<Vec16<T> *>(ys + px * YROWB + chunk * 16) = o;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (Y2F_ABL(2)) {
        } else if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 bfrag, afrag[3];
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int px = ks * 16 + 8 * (g >> 1) + 4 * r + (t16 >> 2);     // pixel row this lane addresses
                    {
                        const unsigned char *p = ys + px * YROWB + (16 * (g & 1) + 4 * (t16 & 3)) * 2;
                        s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p);
                        bf16x4 q = __builtin_bit_cast(bf16x4, v);
                        bfrag[4 * r] = q[0]; bfrag[4 * r + 1] = q[1]; bfrag[4 * r + 2] = q[2]; bfrag[4 * r + 3] = q[3];
                    }
#pragma unroll
                    for (int rt = 0; rt < 3; ++rt) {
                        const int row = 32 * rt + 16 * (g & 1) + 4 * (t16 & 3);       // first of the 4 (tap, channel) rows this lane addresses
                        int tap = row >> 3;
                        if (tap > 8) tap = 8;                                          // rows 72..95: discarded at the end
                        const unsigned char *p = hs + (tap / 3) * First<T>::HROWB + (px + tap % 3) * PXB + (row & 7) * 2;
                        s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p);
                        bf16x4 q = __builtin_bit_cast(bf16x4, v);
                        afrag[rt][4 * r] = q[0]; afrag[rt][4 * r + 1] = q[1]; afrag[rt][4 * r + 2] = q[2]; afrag[rt][4 * r + 3] = q[3];
                    }
                }
#pragma unroll
                for (int rt = 0; rt < 3; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[rt], bfrag, acc[rt], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const int px = ks * 2 + (lane >> 5);
                const float bv = *reinterpret_cast<const float *>(ys + px * YROWB + (lane & 31) * 4);
#pragma unroll
                for (int rt = 0; rt < 3; ++rt) {
                    const int row = 32 * rt + (lane & 31);
                    int tap = row >> 3;
                    if (tap > 8) tap = 8;
                    const float av = *reinterpret_cast<const float *>(hs + (tap / 3) * First<T>::HROWB + (px + tap % 3) * PXB + (row & 7) * 4);
                    acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[rt], 0, 0, 0);
                }
            }
        }
        cb = nb; ch = nh; cw0 = nw0;
        cur = nxt;
        st ^= 1;
    }

    // combine the 4 waves through LDS (the staging slots are dead now), then one atomic per element per workgroup
    __syncthreads();
    float *red = reinterpret_cast<float *>(smem);            // [4 waves][3 tiles][16 regs][64 lanes] = 48 KiB
#pragma unroll
    for (int rt = 0; rt < 3; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave * 3 + rt) * 16 + r) * 64 + lane] = acc[rt][r];
    __syncthreads();
    if (wave == 0) {
        const int n = lane & 31;
#pragma unroll
        for (int rt = 0; rt < 3; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * rt + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);    // (tap, channel)
                const int tap = row >> 3, c = row & 7;
                if (tap < 9 && c < Cin) {
                    float v = 0.f;
#pragma unroll
                    for (int w = 0; w < 4; ++w) v += red[((w * 3 + rt) * 16 + r) * 64 + lane];
#ifdef Y2FIRST_ABL_NOATOMIC
                    if (v == 123.456f) dW[0] = v;
#else
                    unsafeAtomicAdd(dW + ((long)tap * Cin + c) * 32 + n, v);
#endif
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------
// host entry points used by yolo2_conv2d / yolo2_conv2d_wgrad when the shape matches
// ---------------------------------------------------------------------------------------------------
bool y2_first_layer_shape(int Cp, int ldp, int Nf, int ldo, int ksize) { return ksize == 3 && Cp == 8 && ldp == 8 && Nf == 32 && ldo == 32; }

int y2_first_layer_fwd(const void *P, const void *F, void *O, int B, int H, int W, int dtype, hipStream_t st, const float *bn_shift, float *bn_part) {
    const int units = B * H * ((W + 31) / 32);
    const int grid = units / 4 + 8 < 2048 ? (units / 4 + 8) / 8 * 8 : 2048;      // (a multiple of 8: one band of units per XCD)
    // the forward filter operand is [32][9*8]: the generic layout with ldcin = 8
    if (dtype == YOLO2_BF16)
        conv_first_fwd_kernel<bf16><<<grid, 256, 0, st>>>((const bf16 *)P, (unsigned)((size_t)B * H * W * 8 * 2), (const bf16 *)F, (bf16 *)O, B, H, W, units, bn_shift, bn_part);
    else
        conv_first_fwd_kernel<float><<<grid, 256, 0, st>>>((const float *)P, (unsigned)((size_t)B * H * W * 8 * 4), (const float *)F, (float *)O, B, H, W, units, bn_shift, bn_part);
    return 0;
}

int y2_first_layer_wgrad(const void *X, const void *dY, float *dW, int B, int H, int W, int Cin, int dtype, hipStream_t st) {
    const int units = B * H * ((W + 31) / 32);
    const int grid = units / 4 + 8 < 512 ? (units / 4 + 8) / 8 * 8 : 512;
    if (dtype == YOLO2_BF16)
        conv_first_wgrad_kernel<bf16><<<grid, 256, 0, st>>>((const bf16 *)X, (unsigned)((size_t)B * H * W * 8 * 2), (const bf16 *)dY, (unsigned)((size_t)B * H * W * 32 * 2), dW, B, H, W, Cin, units);
    else
        conv_first_wgrad_kernel<float><<<grid, 256, 0, st>>>((const float *)X, (unsigned)((size_t)B * H * W * 8 * 4), (const float *)dY, (unsigned)((size_t)B * H * W * 32 * 4), dW, B, H, W, Cin, units);
    return 0;
}

// ---- fused image layer (conv_first_pool_kernel): host side.  P: image [B,H,W,8] (3 real channels), F: forward filter operand [32][72]
static bool first_pool_ok(int B, int H, int W) { return B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0; }
// filter gradient of the image layer straight from its pooled output gradient (conv_first_wgrad_bn_kernel): include/yolo2_hip.h
extern "C" int yolo2_first_layer_wgrad_bn(const void *X, const void *Y, const void *dP, int lddp, const unsigned char *idx, const float *mean, const float *var,
                                          const float *gamma, const float *beta, const float *part, int rows, long plane, float *dgamma, float *dbeta, float *dW,
                                          int B, int H, int W, int Cin, float eps, float alpha, float *zero, long zero_floats, int dtype, void *stream) {
    Y2_CHECK_ARG(X && Y && dP && idx && mean && var && gamma && beta && part && dgamma && dbeta && dW && rows >= 1 && lddp >= 32 && Cin >= 1 && Cin <= 8 &&
                 first_pool_ok(B, H, W) && (dtype == YOLO2_F32 || dtype == YOLO2_BF16));
    Y2_CHECK_ARG(plane >= (long)rows * 32 && zero_floats >= 0 && zero_floats % 4 == 0 && (zero || zero_floats == 0) && ((uintptr_t)zero & 15) == 0);
    const long zero_vec4 = zero_floats / 4;
    if (zero_vec4 == 0) zero = nullptr;
    const size_t esz = dtype == YOLO2_BF16 ? 2 : 4;
    const size_t M = (size_t)B * H * W;
    Y2_CHECK_ARG(M * 32 * esz < (1ull << 32) && (M / 4) * (size_t)lddp * esz < (1ull << 32));
    Y2_CHECK_ARG(((uintptr_t)X & 15) == 0 && ((uintptr_t)Y & 15) == 0 && ((uintptr_t)dP & 15) == 0 && ((uintptr_t)idx & 15) == 0 && (lddp * esz) % 16 == 0 &&
                 ((uintptr_t)part & 15) == 0 && plane % 4 == 0);
    const int units = B * H * ((W + 31) / 32);
    // two workgroups per CU: measured 80 us against 91 with three (768 workgroups; every workgroup's prologue reads all partial rows) -- batch 16, 416 x 416
    static const int gmax = y2_env_int("YOLO2_FIRST_WGRAD_GRID", 512);
#ifdef Y2FIRST_EXPERIMENTS
    static const int abl = y2_env_int("YOLO2_FIRST_ABL", 0);
#else
    const int abl = 0;
#endif
    const int grid = units / 4 + 8 < gmax ? (units / 4 + 8) / 8 * 8 : gmax;      // (a multiple of 8: one band of units per XCD)
    hipStream_t st = (hipStream_t)stream;
    if (dtype == YOLO2_BF16)
        conv_first_wgrad_bn_kernel<bf16><<<grid, 256, 0, st>>>((const bf16 *)X, (unsigned)(M * 8 * 2), (const bf16 *)Y, (const bf16 *)dP, lddp, idx, mean, var, gamma, beta,
                                                               part, rows, plane, dgamma, dbeta, dW, B, H, W, Cin, units, eps, alpha, zero, zero_vec4, abl);
    else
        conv_first_wgrad_bn_kernel<float><<<grid, 256, 0, st>>>((const float *)X, (unsigned)(M * 8 * 4), (const float *)Y, (const float *)dP, lddp, idx, mean, var, gamma, beta,
                                                                part, rows, plane, dgamma, dbeta, dW, B, H, W, Cin, units, eps, alpha, zero, zero_vec4, abl);
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}

template <int MODE>
static int first_pool_launch(const void *P, const void *F, void *Out, unsigned char *idx, const void *dP, int lddp, float *part, int B, int H, int W, int ldo,
                             const Y2FirstBn &bn, int dtype, hipStream_t st) {
    const int units = B * (H / 2) * ((W + 31) / 32);
    const int grid = units / 4 + 8 < 2048 ? (units / 4 + 8) / 8 * 8 : 2048;      // (a multiple of 8: one band of units per XCD)
    if (dtype == YOLO2_BF16)
        conv_first_pool_kernel<bf16, MODE><<<grid, 256, 0, st>>>((const bf16 *)P, (unsigned)((size_t)B * H * W * 8 * 2), (const bf16 *)F, (bf16 *)Out, idx, (const bf16 *)dP, lddp,
                                                                 part, B, H, W, ldo, units, bn);
    else if (dtype == YOLO2_F32)
        conv_first_pool_kernel<float, MODE><<<grid, 256, 0, st>>>((const float *)P, (unsigned)((size_t)B * H * W * 8 * 4), (const float *)F, (float *)Out, idx, (const float *)dP, lddp,
                                                                  part, B, H, W, ldo, units, bn);
    else { yolo2_set_error("first layer: bad dtype %d", dtype); return YOLO2_E_ARG; }
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}
extern "C" int yolo2_first_layer_stats(const void *P, const void *F, int B, int H, int W, const float *shift, float *bn_part, int dtype, void *stream) {
    Y2_CHECK_ARG(P && F && shift && bn_part && B > 0 && H > 0 && W > 0 && (dtype == YOLO2_F32 || dtype == YOLO2_BF16));
    y2_first_layer_fwd(P, F, nullptr, B, H, W, dtype, (hipStream_t)stream, shift, bn_part);
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}
extern "C" int yolo2_first_layer_bn_leaky_pool(const void *P, const void *F, const float *mean, const float *var, const float *gamma, const float *beta, void *Pout,
                                               unsigned char *idx, int B, int H, int W, int ldp, float eps, float alpha, int dtype, void *stream) {
    Y2_CHECK_ARG(P && F && mean && var && gamma && beta && Pout && ldp >= 32 && first_pool_ok(B, H, W));
    const Y2FirstBn bn{mean, var, gamma, beta, nullptr, nullptr, eps, alpha};
    return first_pool_launch<0>(P, F, Pout, idx, nullptr, 0, nullptr, B, H, W, ldp, bn, dtype, (hipStream_t)stream);
}
extern "C" int yolo2_first_layer_pool_bwd_reduce(const void *P, const void *F, const void *dP, int lddp, const unsigned char *idx, const float *mean, const float *var,
                                                 const float *gamma, const float *beta, float *bn_part, int B, int H, int W, float eps, float alpha, int dtype,
                                                 void *stream) {
    Y2_CHECK_ARG(P && F && dP && idx && mean && var && gamma && beta && bn_part && lddp >= 32 && first_pool_ok(B, H, W));
    const Y2FirstBn bn{mean, var, gamma, beta, nullptr, nullptr, eps, alpha};
    return first_pool_launch<1>(P, F, nullptr, const_cast<unsigned char *>(idx), dP, lddp, bn_part, B, H, W, 0, bn, dtype, (hipStream_t)stream);
}
extern "C" int yolo2_first_layer_pool_bwd_apply(const void *P, const void *F, const void *dP, int lddp, const unsigned char *idx, const float *mean, const float *var,
                                                const float *gamma, const float *beta, const float *dgamma, const float *dbeta, void *dY, int B, int H, int W,
                                                float eps, float alpha, int dtype, void *stream) {
    Y2_CHECK_ARG(P && F && dP && idx && mean && var && gamma && beta && dgamma && dbeta && dY && lddp >= 32 && first_pool_ok(B, H, W));
    const Y2FirstBn bn{mean, var, gamma, beta, dgamma, dbeta, eps, alpha};
    return first_pool_launch<2>(P, F, dY, const_cast<unsigned char *>(idx), dP, lddp, nullptr, B, H, W, 32, bn, dtype, (hipStream_t)stream);
}
