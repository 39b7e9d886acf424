// Filter gradient of the 3x3 NHWC convolution, one kernel ROW of taps per workgroup (gfx950, bf16).
//
// Replaces tf.gradients of slim.layers.conv2d w.r.t. its weights (reference train.py:127-129, call sites
// model/yolo2/inference.py:37-48,73-118) for the 3x3 layers; conv_wgrad.hip keeps the 1x1 layers, the f32 parity mode and the image layer.
//
//   dW[dh][dw][c][n] = sum_m X[pix(m) + (dh, dw), c] * dY[m, n]            (zero where the shifted pixel leaves the image)
//
// What the per-tap kernel of conv_wgrad.hip pays for, and what this one does instead:
//   * it stages X once PER TAP (nine times) with the SAME-padding test in every DMA piece.  Here the reduction runs over a PADDED pixel
//     index q: every image row is followed by one zero column (row pitch W + 1).  In that index the left / right neighbour of a pixel is
//     q -+ 1 for EVERY pixel -- the neighbour of a row's first pixel is the previous row's zero column -- so one staged X tile, read at
//     row offsets -1 / 0 / +1, serves the three taps of a kernel row with no mask anywhere in the loop, and the vertical shift dh is
//     folded into the DMA source address (rows that leave the image read as zeros through the buffer range check).  W / (W + 1) of the
//     MFMA work is real (93 % at 13x13, 96 % at 26x26).
//   * its waves hold 32 x 64 of ONE tap: 3 transpose reads per MFMA.  Here a wave holds 3 taps x 32 channels x 64 filters (six 32x32
//     accumulators): the two dY fragments of a 16-pixel step are shared by the three taps, 10 reads per 6 MFMAs.
//   * it pays one f32 atomic per output element and pixel range, ~5-7 M per launch on the 26x26 .. 104x104 layers (up to 35 us).  Here the
//     KW wave groups of a workgroup take alternate pixel tiles of the workgroup's range and are summed through LDS before anything leaves
//     the CU: one workgroup per CU, a quarter to an eighth of the atomics.
// Tiles are staged pixel-major by LDS-DMA exactly as they lie in HBM and gathered with ds_read_b64_tr_b16 (layout and swizzle of
// conv_wgrad.hip, pinned on hardware by tests/test_kernels_gpu.py::test_tr16_layout); the DMA source offsets are STATELESS: each piece
// derives (row, column) of its padded position with two multiply-high divisions by constants (the per-tap kernel carried (h, w) per
// piece slot through the loop).
#include "common.h"
#include "conv_shared.h"
#include <type_traits>

template <int I, int N, typename F> __device__ __forceinline__ void y2_static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        y2_static_for<I + 1, N>(f);
    }
}

#if defined(__HIP_DEVICE_COMPILE__)
// wait until at most N LDS operations are outstanding; the ten fragment registers of one 16-pixel step are tied to the wait
template <int N> __device__ __forceinline__ void y2_lgkm_wait10(u32x2 &a0, u32x2 &a1, u32x2 &a2, u32x2 &a3, u32x2 &a4, u32x2 &a5, u32x2 &b0, u32x2 &b1, u32x2 &b2, u32x2 &b3) {
    asm volatile("s_waitcnt lgkmcnt(%10)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "n"(N) : "memory");
}
#endif

// floor(q / d) == umulhi(q, m) >> s for 0 <= q < 2^31 (host: y2_magic_u32)
__device__ __forceinline__ unsigned y2_div_magic(unsigned q, unsigned m, unsigned s) { return __umulhi(q, m) >> s; }

// WC x WN waves tile the (channel, filter) extent of the workgroup: BC = 32 WC channels x BN = 64 WN filters x the three taps of kernel row dh;
// KW groups of those take alternate P-pixel tiles of the workgroup's padded pixel range.  NSTAGE ring slots of KW sub-tiles each.
template <int WC, int WN, int KW, int P, int NSTAGE>
__global__ __launch_bounds__(WC *WN *KW * 64) void conv_wgrad_row_kernel(
    const bf16 *__restrict__ X, unsigned x_bytes, const bf16 *__restrict__ dY, unsigned y_bytes, float *__restrict__ dW, int H, int W, int Cin, int ldx,
    int Cout, int ldy, int Mp, int CT, int NT, int qchunk, int KS, int remap, int direct, unsigned mW, unsigned sW, unsigned mH, unsigned sH) {
    constexpr int NWAVE = WC * WN * KW;
    constexpr int BC = 32 * WC, BN = 64 * WN;
    constexpr int XROWB = BC * 2, YROWB = BN * 2;                    // bytes per pixel row of a tile
    constexpr int XCH = XROWB / 16, YCH = YROWB / 16;                // 16-byte chunks per row
    constexpr int XRPI = 1024 / XROWB, YRPI = 1024 / YROWB;          // pixel rows per DMA instruction (1 KiB per wave-instruction)
    constexpr int XH = XRPI / 2;                                     // halo rows in front of the X tile (>= 1: the dw = -1 tap)
    constexpr int XR = P + 2 * XH;                                   // staged X rows: padded positions q0 - XH .. q0 + P + XH - 1
    constexpr int XP = XR / XRPI, YP = P / YRPI;                     // DMA pieces per sub-tile
    constexpr int XT = XR * XROWB, YT = P * YROWB, SUB = XT + YT, STAGE = KW * SUB;
    constexpr int NXP = KW * XP, NYP = KW * YP;                      // pieces per stage, dealt round-robin to the waves
    constexpr int XJ = (NXP + NWAVE - 1) / NWAVE, YJ = (NYP + NWAVE - 1) / NWAVE;
    constexpr int PIECES_MIN = NXP / NWAVE + NYP / NWAVE;            // what EVERY wave has issued per stage (some issue one more of each kind)
    constexpr int KSN = P / 16;                                      // 16-pixel MFMA steps per tile
    // swizzle (conv_wgrad.hip): chunk ^= 4 * ((row / rows_per_bank_line) % min(4, row_bytes / 64)) -- the 4 pixel rows of a transpose read land on 4
    // different bank quarters whatever row the read starts at (the tap offsets -1 / +1 shift the start row)
    constexpr int XRPL = XROWB >= 256 ? 1 : 256 / XROWB, XSWM = XROWB >= 256 ? 4 : (XROWB >= 64 ? XROWB / 64 : 1);
    constexpr int YRPL = YROWB >= 256 ? 1 : 256 / YROWB, YSWM = YROWB >= 256 ? 4 : YROWB / 64;
    static_assert(XH >= 1 && XR % XRPI == 0 && P % YRPI == 0 && P % 16 == 0, "tile geometry");
    static_assert(4 % (XRPL * XSWM) == 0 && 4 % (YRPL * YSWM) == 0, "the swizzle period (in pixel rows) divides the 4-row read step");
    static_assert(XRPI % 4 == 0 && YRPI % 4 == 0, "a lane's swizzle term is the same in every piece");
    static_assert(NSTAGE >= 2 && NSTAGE <= 4 && NSTAGE * STAGE <= 160 * 1024, "LDS ring");
    static_assert(KW == 1 || (KW / 2) * WC * WN * 6 * 4096 <= NSTAGE * STAGE, "the accumulator images of the wave-group reduction fit the ring");

    __shared__ __attribute__((aligned(1024))) unsigned char smem[NSTAGE * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kw = wave / (WC * WN), wcn = wave % (WC * WN), wc = wcn / WN, wn = wcn % WN;

    // ---- block -> (kernel row, channel tile, filter tile, pixel range).  Workgroups of one pixel range read the same X / dY rows: with >= 8
    // ranges they are placed on ONE XCD (block b runs on XCD b % 8: observed, speed only); a single range orders the tiles (channel tile,
    // filter tile, kernel row) with the row fastest, each XCD taking a contiguous run (conv_wgrad.hip remap 2).
    const int ncols = 3 * CT * NT;
    int col, by;
    if (remap == 2) {
        const int ntile = gridDim.x, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, qq = ntile >> 3, rr = ntile & 7;
        col = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;       // bijective for any grid size
        by = 0;
    } else if (remap == 1) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        by = (idx / ncols) * 8 + xcd;
        col = idx % ncols;
    } else {
        col = blockIdx.x % ncols;
        by = blockIdx.x / ncols;
    }
    if (by >= KS) return;
    const int dh = col % 3 - 1;
    const int nt = (col / 3) % NT, ct = col / (3 * NT);
    const int c0 = ct * BC, n0 = nt * BN;
    const int qb = by * qchunk;
    const int qe = min(Mp, qb + qchunk);
    if (qb >= qe) return;
    const int nsuper = (qe - qb + KW * P - 1) / (KW * P);

    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16 *>(X), 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcY = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16 *>(dY), 0, y_bytes, 0x00020000);

    // ---- per-lane DMA constants: row inside a piece, source channel chunk (swizzled), permanently-out-of-range flag folded into the offset
    const int xrow_l = lane / XCH, yrow_l = lane / YCH;
    unsigned x_cb, y_cb;
    {
        const int xc = c0 + (((lane % XCH) ^ (((xrow_l / XRPL) % XSWM) * 4)) * 8);
        x_cb = (xc < Cin && xc < ldx) ? (unsigned)xc * 2u : Y2_OOB;       // (offset + 2^31 stays beyond every operand: they are < 2^31 bytes)
        const int yc = n0 + (((lane % YCH) ^ (((yrow_l / YRPL) % YSWM) * 4)) * 8);
        y_cb = (yc < Cout && yc < ldy) ? (unsigned)yc * 2u : Y2_OOB;
    }
    const unsigned W1 = (unsigned)(W + 1), ldx2 = (unsigned)ldx * 2u, ldy2 = (unsigned)ldy * 2u;
    const int dhW = dh * W;

    int t_issue = 0, i_slot = 0;
    auto issue_stage = [&]() {
        unsigned char *base = smem + i_slot * STAGE;
        const int qs = qb + t_issue * (KW * P);
#pragma unroll
        for (int j = 0; j < XJ; ++j) {
            const int p = wave + j * NWAVE;                           // wave-uniform
            if (p < NXP) {
                const int sub = p / XP, pp = p - sub * XP;
                const unsigned q = (unsigned)(qs + sub * P - XH + pp * XRPI + xrow_l);      // padded position of this lane's row (wraps below 0: fails the range test)
                const unsigned R = y2_div_magic(q, mW, sW);           // image row counted over the whole batch
                const unsigned c = q - __umul24(R, W1);               // column, W = the zero column
                const unsigned r = R - __umul24(y2_div_magic(R, mH, sH), (unsigned)H);
                const bool ok = q < (unsigned)Mp && c < (unsigned)W && (unsigned)((int)r + dh) < (unsigned)H;
                const unsigned voff = ok ? __umul24((unsigned)((int)(q - R) + dhW), ldx2) + x_cb : Y2_OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (__attribute__((address_space(3))) void *)(base + sub * SUB + pp * 1024), 16, voff, 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < YJ; ++j) {
            const int p = wave + j * NWAVE;
            if (p < NYP) {
                const int sub = p / YP, pp = p - sub * YP;
                const unsigned q = (unsigned)(qs + sub * P + pp * YRPI + yrow_l);
                const unsigned R = y2_div_magic(q, mW, sW);
                const unsigned c = q - __umul24(R, W1);
                const bool ok = q < (unsigned)qe && c < (unsigned)W;                       // (rows beyond this block's range contribute nothing)
                const unsigned voff = ok ? __umul24(q - R, ldy2) + y_cb : Y2_OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcY, (__attribute__((address_space(3))) void *)(base + sub * SUB + XT + pp * 1024), 16, voff, 0, 0, 0);
            }
        }
        ++t_issue;
        i_slot = (i_slot + 1 == NSTAGE) ? 0 : i_slot + 1;
    };
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
        if (t_issue < nsuper) issue_stage();

    f32x16 acc[3][2];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[d][j][r] = 0.f;

    // ---- per-lane read addresses (relative to the stage): lane (g, t) supplies pixel row 8 (g >> 1) + (t >> 2), channel quad 16 (g & 1) + 4 (t & 3)
    // of its group's block and receives 4 pixels of channel 16 (g & 1) + t (conv_wgrad.hip).  Tap dw reads the X tile dw rows further.
    const int g = lane >> 4, t = lane & 15;
    const int px = 8 * (g >> 1) + (t >> 2), co = 16 * (g & 1) + 4 * (t & 3);
    unsigned xa[3], ya[2];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int i0 = XH + (d - 1) + px, ch = wc * 32 + co;
        xa[d] = (unsigned)(kw * SUB + i0 * XROWB + (((ch >> 3) ^ (((i0 / XRPL) % XSWM) * 4)) << 4) + ((ch & 7) << 1));
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ch = (wn * 2 + j) * 32 + co;
        ya[j] = (unsigned)(kw * SUB + XT + px * YROWB + (((ch >> 3) ^ (((px / YRPL) % YSWM) * 4)) << 4) + ((ch & 7) << 1));
    }
    const unsigned smem_base = y2_lds_addr(smem);

    int c_slot = 0;
    for (int s = 0; s < nsuper; ++s) {
        const int ahead = min(NSTAGE - 2, t_issue - 1 - s);           // stages issued behind the one consumed now
        if (NSTAGE >= 4 && ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES_MIN) : "memory");
        else if (NSTAGE >= 3 && ahead >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES_MIN) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t_issue < nsuper) issue_stage();
#if defined(__HIP_DEVICE_COMPILE__)
        const unsigned sb = smem_base + (unsigned)(c_slot * STAGE);
        c_slot = (c_slot + 1 == NSTAGE) ? 0 : c_slot + 1;
        unsigned xs[3], ys[2];
#pragma unroll
        for (int d = 0; d < 3; ++d) xs[d] = sb + xa[d];
#pragma unroll
        for (int j = 0; j < 2; ++j) ys[j] = sb + ya[j];
        u32x2 fa[2][3][2], fb[2][2][2];
        auto load = [&](auto ks_) {
            constexpr int ks = decltype(ks_)::value, bf = ks & 1;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                fa[bf][d][0] = y2_tr16_read_off<(16 * ks) * XROWB>(xs[d]);
                fa[bf][d][1] = y2_tr16_read_off<(16 * ks + 4) * XROWB>(xs[d]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                fb[bf][j][0] = y2_tr16_read_off<(16 * ks) * YROWB>(ys[j]);
                fb[bf][j][1] = y2_tr16_read_off<(16 * ks + 4) * YROWB>(ys[j]);
            }
        };
        load(std::integral_constant<int, 0>{});
        y2_static_for<0, KSN>([&](auto ks_) {
            constexpr int ks = decltype(ks_)::value, bf = ks & 1;
            if constexpr (ks + 1 < KSN) {
                load(std::integral_constant<int, ks + 1>{});
                y2_lgkm_wait10<10>(fa[bf][0][0], fa[bf][0][1], fa[bf][1][0], fa[bf][1][1], fa[bf][2][0], fa[bf][2][1], fb[bf][0][0], fb[bf][0][1], fb[bf][1][0], fb[bf][1][1]);
            } else {
                y2_lgkm_wait10<0>(fa[bf][0][0], fa[bf][0][1], fa[bf][1][0], fa[bf][1][1], fa[bf][2][0], fa[bf][2][1], fb[bf][0][0], fb[bf][0][1], fb[bf][1][0], fb[bf][1][1]);
            }
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[d][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y2_frag16(fa[bf][d][0], fa[bf][d][1]), y2_frag16(fb[bf][j][0], fb[bf][j][1]), acc[d][j], 0, 0, 0);
        });
#endif
    }

    // ---- the KW wave groups hold partial sums over disjoint pixel tiles: tree reduction through LDS (the ring is free: every DMA has landed
    // and been read), group 0 ends with the workgroup's sums
    if constexpr (KW > 1) {
        __syncthreads();
        f32x4 *img = reinterpret_cast<f32x4 *>(smem);
#pragma unroll
        for (int h = KW / 2; h >= 1; h >>= 1) {
            if (kw >= h && kw < 2 * h) {
                f32x4 *dst = img + ((kw - h) * (WC * WN) + wcn) * (6 * 4 * 64) + lane;
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
            if (kw < h) {
                const f32x4 *src = img + (kw * (WC * WN) + wcn) * (6 * 4 * 64) + lane;
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
        if (kw != 0) return;
    }

    // ---- output: rows = input channels, columns = filters (lanes along n: coalesced); dW is HWIO [tap][Cin][Cout]
    auto write_tile = [&](auto checked_tag) {
        constexpr bool CHECKED = decltype(checked_tag)::value;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            float *out = dW + (long)((dh + 1) * 3 + d) * Cin * Cout;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + (wn * 2 + j) * 32 + (lane & 31);
                if (CHECKED && n >= Cout) continue;
                const int cb = c0 + wc * 32 + 4 * (lane >> 5);
                float *colp = out + (long)cb * Cout + n;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dc = (r & 3) + 8 * (r >> 2);
                    if (!CHECKED || cb + dc < Cin) {
                        if (direct) colp[(long)dc * Cout] = acc[d][j][r];      // single pixel range: this workgroup owns the element
                        else unsafeAtomicAdd(colp + (long)dc * Cout, acc[d][j][r]);
                    }
                }
            }
        }
    };
    if (c0 + BC <= Cin && n0 + BN <= Cout) write_tile(std::false_type{});
    else write_tile(std::true_type{});
}

// floor(q / d) == umulhi(q, m) >> s for every 0 <= q < 2^31, d >= 2 (round-up magic number; powers of two take one bit less so that m fits 32 bits)
void y2_magic_u32(unsigned d, unsigned *m, unsigned *s) {
    int k = 0;
    while ((2u << k) <= d && k < 31) ++k;                    // k = floor(log2 d)
    const bool pow2 = (d & (d - 1)) == 0;
    const int L = pow2 ? 31 + k : 32 + k;
    *m = (unsigned)(((1ull << L) / d) + 1ull);
    *s = (unsigned)(L - 32);
}

// ---- plan: which instantiation, how many pixel ranges.  Variants: 0 = 32 x 64 tile, eight wave groups (<= 32 input channels); 1 = 64 x 64 tile,
// four wave groups (split reductions: the atomics per launch are workgroups x 12 K elements); 2 = 64 x 128 tile, two wave groups (single-range
// grids: 13x13 stages, plain stores).
struct Y2W3Geom { int BC, BN, KW, P, waves; };
static const Y2W3Geom g_w3_geom[3] = {{32, 64, 8, 32, 8}, {64, 64, 4, 32, 8}, {64, 128, 2, 64, 8}};

Y2W3Plan y2_wgrad3_plan(int B, int H, int W, int Cin, int Cout, int cus, int force_variant) {
    Y2W3Plan p = {};
    p.variant = -1;
    const long Mp = (long)B * H * (W + 1);
    if (H < 2 || W < 2 || Cin <= 8 || Mp + 8192 >= (1L << 24)) return p;       // (24-bit multiplies in the DMA address arithmetic; the image layer has its own kernel)
    if (cus <= 0) cus = 256;
    auto cols = [&](int v) { return 3L * cdiv(Cin, g_w3_geom[v].BC) * cdiv(Cout, g_w3_geom[v].BN); };
    int v;
    if (force_variant >= 0 && force_variant <= 2) v = force_variant;
    else if (Cin <= 32) v = 0;
    else if (cols(2) * 10 >= (long)cus * 6) v = 2;                             // the 64 x 128 tile grid alone gives >= 60 % of the CUs a workgroup
    else v = 1;
    const Y2W3Geom &gm = g_w3_geom[v];
    const long ncols = cols(v);
    const int step = gm.KW * gm.P;
    long ks = ncols >= cus ? 1 : cus / ncols;
    const long max_ks = Mp / (4L * step) > 1 ? Mp / (4L * step) : 1;          // >= four ring turns per workgroup
    if (ks > max_ks) ks = max_ks;
    if (v == 2 && force_variant < 0) ks = 1;
    p.qchunk = cdiv(cdiv(Mp, ks), step) * step;
    p.ks = cdiv(Mp, p.qchunk);
    p.variant = v;
    p.direct = p.ks == 1;
    p.remap = p.ks == 1 ? 2 : (p.ks >= 8 ? 1 : 0);
    p.blocks = (int)(p.remap == 1 ? ncols * (cdiv(p.ks, 8) * 8) : ncols * p.ks);
    p.BC = gm.BC; p.BN = gm.BN; p.waves = gm.waves;
    return p;
}

int y2_wgrad3_launch(const Y2W3Plan &p, const void *X, const void *dY, float *dW, int B, int H, int W, int Cin, int ldx, int Cout, int ldy, hipStream_t st) {
    const Y2W3Geom &gm = g_w3_geom[p.variant];
    const int Mp = B * H * (W + 1), CT = cdiv(Cin, gm.BC), NT = cdiv(Cout, gm.BN);
    const unsigned x_bytes = (unsigned)((size_t)B * H * W * ldx * 2), y_bytes = (unsigned)((size_t)B * H * W * ldy * 2);
    unsigned mW, sW, mH, sH;
    y2_magic_u32((unsigned)(W + 1), &mW, &sW);
    y2_magic_u32((unsigned)H, &mH, &sH);
#define Y2W3_LAUNCH(WCv, WNv, KWv, Pv, NSv)                                                                                                         \
    conv_wgrad_row_kernel<WCv, WNv, KWv, Pv, NSv><<<dim3(p.blocks), WCv * WNv * KWv * 64, 0, st>>>((const bf16 *)X, x_bytes, (const bf16 *)dY, y_bytes, dW, H, W, Cin, \
                                                                                                   ldx, Cout, ldy, Mp, CT, NT, p.qchunk, p.ks, p.remap, p.direct, mW, sW, mH, sH)
    switch (p.variant) {
        case 0: Y2W3_LAUNCH(1, 1, 8, 32, 2); break;
        case 1: Y2W3_LAUNCH(2, 1, 4, 32, 4); break;
        case 2: Y2W3_LAUNCH(2, 2, 2, 64, 3); break;
        default: return 1;
    }
#undef Y2W3_LAUNCH
    return 0;
}
