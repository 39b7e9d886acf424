// Filter gradient of the 3x3 NHWC convolution with 32 input channels and 64 filters (Darknet-19 conv1, 208 x 208 pixels per image), bf16, gfx950.
//
// Replaces tf.gradients of slim.layers.conv2d w.r.t. its weights (reference train.py:127-129) for model/yolo2/inference.py:76 (conv1: 32 -> 64).
//
//   dW[dh][dw][c][n] = sum_m X[pix(m) + (dh, dw), c] * dY[m, n]            (zero where the shifted pixel leaves the image)
//
// The layer is 25.5 GFLOP over 133 MB (X 44 MB + dY 89 MB at batch 16): HBM-bound at ~25 us if every byte is read once.  The per-tap kernel
// (conv_wgrad.hip, two taps per 64-row tile) stages both operands five times: 52 us warm, 66 us inside a training step; the row-of-taps kernel
// (conv_wgrad3.hip) stages them three times, in three workgroups.  Here ONE workgroup owns all nine taps (288 x 64 outputs = 18 accumulator blocks)
// over a contiguous run of image rows and reads every byte once:
//   * LDS holds a ring of X image rows (one zero position in front, 64 bytes per position) and a ring of dY image rows (128 bytes per pixel), staged
//     by LDS-DMA exactly as they lie in HBM, D rows ahead.  Image row R needs X rows R - 1, R, R + 1: only ONE new X row and one dY row stream in per
//     step; the column shift dw is a row offset of the transpose read, the row shift dh a choice of ring slot -- no padded index, no mask, no division;
//   * twelve waves = 3 kernel rows x 4 pixel groups.  A wave holds the three taps of its kernel row (six 32 x 32 accumulators) and takes every fourth
//     16-pixel step of the row: 5 fragment reads (ds_read_b64_tr_b16 pairs) per 6 MFMAs; a wave whose kernel row leaves the image at this image row
//     sits the step out (wave-uniform);
//   * one barrier per image row; the four pixel groups are summed through LDS at the end and the workgroup adds its 18432 sums to dW with f32 atomics
//     (plain stores when the launch is one workgroup).
// Measured: profiles/r06_conv1_wgrad_c32.txt.
#include "common.h"
#include "conv_shared.h"
#include <atomic>
#include <type_traits>

struct W32Geo {
    int H, W, BH;          // image rows / columns, image rows of the batch
    int rpw;               // image rows per workgroup
    int KSN;               // 16-pixel steps per image row
    int NPX, NPY;          // 1 KiB DMA pieces per X slot (16 positions of 64 bytes) / per dY slot (8 pixels of 128 bytes)
    int NPW;               // DMA instructions per wave and stage (the same for every wave: the counted waits rely on it)
    int XSB, YSB;          // bytes per X / dY slot
    int y_off, dummy_off;  // LDS byte offsets of the dY slots and of the piece that absorbs the padding instructions
};

// D: image rows in flight ahead of the one being multiplied (2 where LDS allows: W <= 224).  ABL (timing ablations of -DY2W32_EXPERIMENTS builds, results
// wrong by design): 1 = no MFMA, 2 = no fragment reads, 4 = no DMA inside the loop, 32 = no output, 64 = no pixel-group reduction, 256 = the workgroup returns at once,
// 512 = wall-clock (100 MHz) stamps: every wave leaves {kernel, prologue, row loop, drain + first barrier, LDS reduction, atomics issued + acknowledged} cycles behind dW (the caller allocates 18432 + blocks x 12 x 16 floats)
template <int D, int ABL = 0>
__global__ __launch_bounds__(768) void conv_wgrad_c32_kernel(const bf16 *__restrict__ X, unsigned x_bytes, const bf16 *__restrict__ dY, unsigned y_bytes,
                                                             float *__restrict__ dW, W32Geo g) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    constexpr int NSX = D + 3, NSY = D + 1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pg = wave / 3, kr = wave - 3 * pg;            // pixel group; kernel row (dh = kr - 1)
    const int W = g.W, H = g.H;
    const int R0 = (int)blockIdx.x * g.rpw, R1 = min(R0 + g.rpw, g.BH);      // this workgroup's image rows (counted over the batch)
    if (R0 >= g.BH) return;
    const int nsteps = R1 - R0;
    unsigned long long tm0 = 0, tm1 = 0, tm2 = 0, tm3 = 0, tm4 = 0;
    if constexpr (ABL & 512) tm0 = wall_clock64();
    if constexpr (ABL & 256) { if (dW[0] == 123.456f) dW[1] = (float)nsteps; return; }

    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16 *>(X), 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcY = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16 *>(dY), 0, y_bytes, 0x00020000);
    typedef __attribute__((address_space(3))) void *lds_void_ptr;

    // ---- staging.  X slot: position p holds column p - 1 of the row (p = 0 and p > W: zeros); piece i = positions 16 i .. 16 i + 15, lane = (position
    // lane >> 2, 16-byte chunk lane & 3).  dY slot: piece j = pixels 8 j .. 8 j + 7, lane = (pixel lane >> 3, chunk lane & 7), source chunk swizzled with
    // bit 1 of the pixel (conv_wgrad.hip: the four pixel rows of a transpose read then land on four different bank quarters).
    const int xrow_l = lane >> 2, yrow_l = lane >> 3;
    const unsigned xchunk = (unsigned)((lane & 3) << 4);
    const unsigned ychunk = (unsigned)(((lane & 7) ^ (4 * ((yrow_l >> 1) & 1))) << 4);
    auto x_piece = [&](int i, int Rx, int slot) {
        const int c = 16 * i + xrow_l - 1;
        const bool ok = ((unsigned)c < (unsigned)W) & ((unsigned)Rx < (unsigned)g.BH) & (Rx <= R1);      // (rows behind R1 are never read)
        const unsigned voff = ok ? (unsigned)(Rx * W + c) * 64u + xchunk : Y2_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (lds_void_ptr)(smem + slot * g.XSB + i * 1024), 16, voff, 0, 0, 0);
    };
    auto y_piece = [&](int j, int Ry, int slot) {
        const unsigned voff = Ry < R1 ? (unsigned)(Ry * W + 8 * j + yrow_l) * 128u + ychunk : Y2_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcY, (lds_void_ptr)(smem + g.y_off + slot * g.YSB + j * 1024), 16, voff, 0, 0, 0);
    };
    // stage s = X row R0 + 1 + s (ring slot (s + 2) % NSX: row R0 - 1 + k lives in slot k % NSX) and dY row R0 + s (slot s % NSY); every wave issues
    // exactly NPW instructions (pieces beyond the stage's go to the dummy piece with an out-of-range source: they return zeros and cost no traffic)
    int i_sx = 2, i_sy = 0, i_s = 0;
    auto issue_stage = [&]() {
        if constexpr (!(ABL & 4)) {
            const int Rx = R0 + 1 + i_s, Ry = R0 + i_s;
            for (int k = 0; k < g.NPW; ++k) {
                const int i = wave + 12 * k;
                if (i < g.NPX) x_piece(i, Rx, i_sx);
                else if (i < g.NPX + g.NPY) y_piece(i - g.NPX, Ry, i_sy);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (lds_void_ptr)(smem + g.dummy_off), 16, Y2_OOB, 0, 0, 0);
            }
        }
        ++i_s;
        i_sx = i_sx + 1 == NSX ? 0 : i_sx + 1;
        i_sy = i_sy + 1 == NSY ? 0 : i_sy + 1;
    };
    // prologue: X rows R0 - 1 and R0 (slots 0 and 1), then stages 0 .. D - 1
    for (int i = wave; i < 2 * g.NPX; i += 12) {
        const int sl = i >= g.NPX ? 1 : 0;
        x_piece(i - sl * g.NPX, R0 - 1 + sl, sl);
    }
#pragma unroll
    for (int s = 0; s < D; ++s) issue_stage();

    f32x16 acc[3][2];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[d][j][r] = 0.f;

    // ---- per-lane read addresses: lane (gq, t) supplies pixel row 8 (gq >> 1) + (t >> 2), channel quad 16 (gq & 1) + 4 (t & 3) of a 16-pixel step and
    // receives 4 pixels of channel 16 (gq & 1) + t (conv_wgrad.hip; layout pinned by tests/test_kernels_gpu.py::test_tr16_layout).  Tap dw = d - 1 of output
    // column c reads X position c + d.  This wave's steps are pg, pg + 4, ...: 1 KiB of X and 2 KiB of dY per step.
    const int gq = lane >> 4, t = lane & 15;
    const int px = 8 * (gq >> 1) + (t >> 2), co = 16 * (gq & 1) + 4 * (t & 3);
    const unsigned lds0 = y2_lds_addr(smem);
    unsigned xl[3], yl[2];
#pragma unroll
    for (int d = 0; d < 3; ++d) xl[d] = (unsigned)(pg * 1024 + (px + d) * 64 + co * 2);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ch = j * 32 + co;
        yl[j] = (unsigned)(g.y_off + pg * 2048 + px * 128 + (((ch >> 3) ^ (4 * ((px >> 1) & 1))) << 4) + ((ch & 7) << 1));
    }
    const int kcount = (g.KSN - pg + 3) >> 2;              // 16-pixel steps of this wave per image row

    int r_img = R0 % H;                                     // row inside its image
    int c_sx = kr, c_sy = 0;                                // ring slots of X row R + dh and of dY row R
    u32x2 fa[3][2], fb[2][2];
    unsigned xs[3], ys[2];
#if defined(__HIP_DEVICE_COMPILE__)
    // one 16-pixel step: ten transpose reads (X tap 0, dY 0, dY 1, X tap 1, X tap 2), six MFMAs issued as their operands arrive (LDS returns in order:
    // "at most N outstanding" names the reads issued after the one needed).  The other two waves of the SIMD fill the matrix pipe meanwhile.
#define W32_MMA(d, j)                                                                                                              \
    do {                                                                                                                          \
        if constexpr (ABL & 1) acc[d][j][0] += __builtin_bit_cast(float, fa[d][0][0] ^ fa[d][1][1] ^ fb[j][0][0] ^ fb[j][1][1]);   \
        else acc[d][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y2_frag16(fa[d][0], fa[d][1]), y2_frag16(fb[j][0], fb[j][1]), acc[d][j], 0, 0, 0); \
    } while (0)
    auto kstep = [&]() {
        if constexpr (!(ABL & 2)) {
            fa[0][0] = y2_tr16_read_off<0>(xs[0]); fa[0][1] = y2_tr16_read_off<256>(xs[0]);
            fb[0][0] = y2_tr16_read_off<0>(ys[0]); fb[0][1] = y2_tr16_read_off<512>(ys[0]);
            fb[1][0] = y2_tr16_read_off<0>(ys[1]); fb[1][1] = y2_tr16_read_off<512>(ys[1]);
            fa[1][0] = y2_tr16_read_off<0>(xs[1]); fa[1][1] = y2_tr16_read_off<256>(xs[1]);
            fa[2][0] = y2_tr16_read_off<0>(xs[2]); fa[2][1] = y2_tr16_read_off<256>(xs[2]);
            y2_lgkm_wait4<6>(fa[0][0], fa[0][1], fb[0][0], fb[0][1]);
            W32_MMA(0, 0);
            y2_lgkm_wait2<4>(fb[1][0], fb[1][1]);
            W32_MMA(0, 1);
            y2_lgkm_wait2<2>(fa[1][0], fa[1][1]);
            W32_MMA(1, 0);
            W32_MMA(1, 1);
            y2_lgkm_wait2<0>(fa[2][0], fa[2][1]);
            W32_MMA(2, 0);
            W32_MMA(2, 1);
        } else {
#pragma unroll
            for (int d = 0; d < 3; ++d) { fa[d][0] = u32x2{xs[d], 0u}; fa[d][1] = u32x2{0u, xs[d]}; }
#pragma unroll
            for (int j = 0; j < 2; ++j) { fb[j][0] = u32x2{ys[j], 0u}; fb[j][1] = u32x2{0u, ys[j]}; }
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int j = 0; j < 2; ++j) W32_MMA(d, j);
        }
        xs[0] += 4096u; xs[1] += 4096u; xs[2] += 4096u; ys[0] += 8192u; ys[1] += 8192u;
    };
#undef W32_MMA
#else
    auto kstep = [&]() {};
#endif

    auto wait_stage = [&]() {                               // this wave's pieces of the oldest stage in flight have landed: (D - 1) NPW instructions may stay out
        const int n = (D - 1) * g.NPW;
        if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (n >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (n == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (n == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (n == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    if constexpr (ABL & 512) tm1 = wall_clock64();
    for (int s = 0; s < nsteps; ++s) {
        wait_stage();
        __builtin_amdgcn_s_barrier();                       // stage s is in LDS for every wave; every wave has left step s - 1 (its slots are refilled below)
        issue_stage();                                      // stage s + D
        if ((unsigned)(r_img + kr - 1) < (unsigned)H && kcount > 0) {
            const unsigned xb = lds0 + (unsigned)(c_sx * g.XSB), yb = lds0 + (unsigned)(c_sy * g.YSB);
#pragma unroll
            for (int d = 0; d < 3; ++d) xs[d] = xb + xl[d];
#pragma unroll
            for (int j = 0; j < 2; ++j) ys[j] = yb + yl[j];
            for (int i = 0; i < kcount; ++i) kstep();
        }
        r_img = r_img + 1 == H ? 0 : r_img + 1;
        c_sx = c_sx + 1 == NSX ? 0 : c_sx + 1;
        c_sy = c_sy + 1 == NSY ? 0 : c_sy + 1;
    }
    // ---- the four pixel groups hold partial sums over disjoint pixels: groups 2, 3 -> 0, 1, then 1 -> 0, through LDS (every DMA has landed -- the
    // padding instructions too, they write zeros -- and been read)
    if constexpr (ABL & 512) { __builtin_amdgcn_sched_barrier(0); tm2 = wall_clock64(); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if constexpr (ABL & 512) { __builtin_amdgcn_sched_barrier(0); tm3 = wall_clock64(); }
    f32x4 *img = reinterpret_cast<f32x4 *>(smem);
#pragma unroll
    for (int h = (ABL & 64) ? 0 : 2; h >= 1; h >>= 1) {
        if (pg >= h && pg < 2 * h) {
            f32x4 *dst = img + ((pg - h) * 3 + kr) * (6 * 4 * 64) + lane;
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const f32x4 v = {acc[d][j][4 * q4], acc[d][j][4 * q4 + 1], acc[d][j][4 * q4 + 2], acc[d][j][4 * q4 + 3]};
                        dst[((d * 2 + j) * 4 + q4) * 64] = v;
                    }
        }
        __syncthreads();
        if (pg < h) {
            const f32x4 *src = img + (pg * 3 + kr) * (6 * 4 * 64) + lane;
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const f32x4 v = src[((d * 2 + j) * 4 + q4) * 64];
                        acc[d][j][4 * q4] += v[0];
                        acc[d][j][4 * q4 + 1] += v[1];
                        acc[d][j][4 * q4 + 2] += v[2];
                        acc[d][j][4 * q4 + 3] += v[3];
                    }
        }
        if (h > 1) __syncthreads();
    }
    if constexpr (ABL & 512) {
        __builtin_amdgcn_sched_barrier(0);
        tm4 = wall_clock64();
        if (pg != 0) {
            if (lane == 0) {
                unsigned long long *dbg = reinterpret_cast<unsigned long long *>(dW + 18432) + ((long)blockIdx.x * 12 + wave) * 8;
                dbg[0] = tm4 - tm0; dbg[1] = tm1 - tm0; dbg[2] = tm2 - tm1; dbg[3] = tm3 - tm2; dbg[4] = tm4 - tm3; dbg[5] = 0; dbg[6] = (unsigned long long)nsteps; dbg[7] = 0;
            }
            return;
        }
    }
    if (pg != 0) return;
    if constexpr (ABL & 32) {
        if (acc[0][0][0] == 123.456f && acc[2][1][5] == 1.0f) dW[0] = acc[1][1][3];
        return;
    }
    // ---- output: dW is HWIO [tap][32][64]; accumulator register r of lane l is channel 4 (l >> 5) + (r & 3) + 8 (r >> 2), filter 32 j + (l & 31)
    auto write_out = [&](auto direct_) {
        constexpr bool DIRECT = decltype(direct_)::value;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            float *out = dW + (kr * 3 + d) * (32 * 64) + (4 * (lane >> 5)) * 64 + (lane & 31);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float *p = out + ((r & 3) + 8 * (r >> 2)) * 64 + 32 * j;
                    if (DIRECT) *p = acc[d][j][r];
                    else unsafeAtomicAdd(p, acc[d][j][r]);
                }
        }
    };
    if (gridDim.x == 1) write_out(std::true_type{});       // one workgroup: it owns dW (which may be dirty)
    else write_out(std::false_type{});
    if constexpr (ABL & 512) {
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long tm5 = wall_clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long tm6 = wall_clock64();
        if (lane == 0) {
            unsigned long long *dbg = reinterpret_cast<unsigned long long *>(dW + 18432) + ((long)blockIdx.x * 12 + wave) * 8;
            dbg[0] = tm6 - tm0; dbg[1] = tm1 - tm0; dbg[2] = tm2 - tm1; dbg[3] = tm3 - tm2; dbg[4] = tm4 - tm3; dbg[5] = tm5 - tm4; dbg[6] = (unsigned long long)nsteps; dbg[7] = tm6 - tm5;
        }
    }
}

static const bool g_w32_on = y2_env_int("YOLO2_WGRAD_C32", 1) != 0;

bool y2_w32_shape(int Cin, int ldx, int Cout, int ldy, int ksize, int dtype) {
    return g_w32_on && dtype == YOLO2_BF16 && ksize == 3 && Cin == 32 && ldx == 32 && Cout == 64 && ldy == 64;
}

// the launch plan: workgroups (0: W is not a multiple of 16 or the rows do not fit LDS -- the caller takes the per-tap kernel)
static int w32_plan(int B, int H, int W, int cus, W32Geo &g, int &D, size_t &lds) {
    if (W % 16 != 0 || W < 16 || B < 1 || H < 1) return 0;
    g.H = H; g.W = W; g.BH = B * H;
    g.KSN = W / 16;
    g.NPX = (W + 16) / 16; g.NPY = W / 8;
    g.NPW = (g.NPX + g.NPY + 11) / 12;
    g.XSB = (W + 16) * 64; g.YSB = W * 128;
    if (cus < 1) cus = 256;
    g.rpw = (g.BH + cus - 1) / cus;
    const size_t red = 6 * 6 * 4096;                                  // the accumulator images of two pixel groups
    D = 2;
    size_t need = (size_t)(D + 3) * g.XSB + (size_t)(D + 1) * g.YSB + 1024;
    if (need > 160 * 1024) { D = 1; need = (size_t)(D + 3) * g.XSB + (size_t)(D + 1) * g.YSB + 1024; }
    if (need > 160 * 1024) return 0;
    g.y_off = (D + 3) * g.XSB;
    g.dummy_off = g.y_off + (D + 1) * g.YSB;
    lds = need > red ? need : red;
    return (g.BH + g.rpw - 1) / g.rpw;
}

int y2_w32_blocks(int B, int H, int W, int cus) {
    W32Geo g;
    int D;
    size_t lds;
    return w32_plan(B, H, W, cus, g, D, lds);
}

// 0: launched (grid in *blocks).  Non-zero: not taken (see w32_plan)
int y2_w32_wgrad(const void *X, const void *dY, float *dW, int B, int H, int W, int cus, int *blocks, hipStream_t st) {
    W32Geo g;
    int D;
    size_t lds;
    const int grid = w32_plan(B, H, W, cus, g, D, lds);
    if (grid < 1) return 1;
    static std::atomic<size_t> lds_set[2][64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    const unsigned x_bytes = (unsigned)((size_t)g.BH * W * 64), y_bytes = (unsigned)((size_t)g.BH * W * 128);
#define W32_LAUNCH(Dv)                                                                                                                        \
    do {                                                                                                                                      \
        if (lds > lds_set[Dv - 1][dev].load(std::memory_order_relaxed)) {                                                                     \
            if (hipFuncSetAttribute((const void *)conv_wgrad_c32_kernel<Dv>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 1; \
            lds_set[Dv - 1][dev].store(lds, std::memory_order_relaxed);                                                                       \
        }                                                                                                                                     \
        conv_wgrad_c32_kernel<Dv><<<grid, 768, lds, st>>>((const bf16 *)X, x_bytes, (const bf16 *)dY, y_bytes, dW, g);                        \
    } while (0)
#if defined(Y2W32_EXPERIMENTS)
    {
        const char *e = getenv("YOLO2_W32_ABL");
        const int abl = e ? atoi(e) : 0;
#define W32_LAUNCH_ABL(A)                                                                                                                     \
    if (abl == A) {                                                                                                                           \
        if (hipFuncSetAttribute((const void *)conv_wgrad_c32_kernel<2, A>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 1; \
        conv_wgrad_c32_kernel<2, A><<<grid, 768, lds, st>>>((const bf16 *)X, x_bytes, (const bf16 *)dY, y_bytes, dW, g);                      \
        if (blocks) *blocks = grid;                                                                                                           \
        return 0;                                                                                                                             \
    }
        if (D == 2) {
            W32_LAUNCH_ABL(1) W32_LAUNCH_ABL(2) W32_LAUNCH_ABL(3) W32_LAUNCH_ABL(4) W32_LAUNCH_ABL(7) W32_LAUNCH_ABL(32) W32_LAUNCH_ABL(96) W32_LAUNCH_ABL(99) W32_LAUNCH_ABL(103) W32_LAUNCH_ABL(256) W32_LAUNCH_ABL(512)
        }
#undef W32_LAUNCH_ABL
    }
#endif
    if (D == 2) W32_LAUNCH(2);
    else W32_LAUNCH(1);
#undef W32_LAUNCH
    if (blocks) *blocks = grid;
    return 0;
}
