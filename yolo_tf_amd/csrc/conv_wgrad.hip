// Filter gradient of the NHWC convolution on MFMA (gfx950).
//
// Replaces tf.gradients of slim.layers.conv2d w.r.t. its weights (reference train.py:127-129,
// call sites model/yolo2/inference.py:37-48,73-118):
//   dW[tap][c][n] += sum_{m in pixel range} X[pix(m) + shift(tap), c] * dY[m, n]
//   GEMM view per tap: rows = Cin, cols = Cout, reduction = B*H*W pixels.
//
// The reduction index (pixels) is the strided one in NHWC for BOTH operands, so the MFMA fragments
// need a transpose.  Tiles are staged pixel-major [pixel][channel] exactly as they lie in HBM, by DMA
// (buffer_load_dwordx4 ... lds, 1 KiB = a few whole pixel rows per wave-instruction), and the bf16
// fragments are gathered with the gfx950 hardware transpose read ds_read_b64_tr_b16: each 16-lane
// group reads a 4-pixel x 16-channel block and every lane receives 4 pixels of its own channel; two
// reads give the 8 reduction slots of v_mfma_f32_32x32x16_bf16 (layout pinned on hardware by
// tests/test_kernels_gpu.py::test_tr16_layout).  DMA destinations are lane-linear, so instead of
// padding rows the 16-byte chunk index is XOR-ed with 4*(pixel_row_in_bank_line) on the source side and
// in the reads: the 4 pixel rows a transpose read touches land on 4 disjoint bank quarters
// (SQ_LDS_BANK_CONFLICT = 0).  The f32 parity path uses v_mfma_f32_32x32x2_f32 (one scalar per lane,
// plain ds_read_b32, lanes along channels).
// Image borders / SAME padding / tails: the X lane whose shifted pixel falls outside the image, and every
// lane beyond the pixel range or channel count, uses an out-of-range buffer offset -> the DMA writes 0.
// 3-stage LDS ring with counted vmcnt + raw s_barrier as in conv_igemm.hip (64-pixel reduction tiles on a 2-stage ring when a launch gives a
// CU about one workgroup).  The transpose reads are inline asm (the builtin drains the DMA ring: see common.h) with ONE address per
// fragment row and the row step in the instruction's offset field.
// (A 256 x 128 tile with 64-pixel reduction tiles -- 64 x 64 per wave, 144 KiB ring, one workgroup per CU -- was
// measured 1.5-1.9x slower on the 13x13 / 26x26 stages: the per-DMA-piece border bookkeeping and the single resident
// workgroup outweigh the halved LDS traffic; profiles/r01_wgrad_big_tile.txt.)
// Grid: x = (tap, c-tile, n-tile), y = pixel-range split; partial tiles are accumulated into the
// zero-initialised f32 dW with hardware f32 atomics (lanes run along n -> coalesced).  A tile grid that already covers the
// chip takes ONE pixel range and stores instead (yolo2_conv2d_wgrad_accumulates tells the caller which arena ranges still
// need zeroing).  Shapes: 128x128 tile with 8 waves for the wide layers, 64x64 with 4 waves for <= 64 channels / few-tile
// layers, two taps per 64-row tile for <= 32 input channels (PAIR).  The per-DMA-piece border bookkeeping ((h, w) of every
// staged pixel row, advanced branch-free by a constant pixel step) is the kernel's main non-MFMA cost.
#include "common.h"
#include "conv_shared.h"
#include <atomic>
#include <stdlib.h>
#include <type_traits>

// PAIR (BC = 64, Cin <= 32): the 64 tile rows hold TWO taps x 32 channels (rows 0-31: tap 2p, rows 32-63: tap 2p+1), so a
// 32-channel layer (conv1: 25 GFLOP) does not spend half of every MFMA on zero rows.
template <typename T, int BC, int BNN, int VARIANT, int NW = 4, int BKPv = 0, bool PAIR = false>
__global__ __launch_bounds__(NW * 64) void conv_wgrad_kernel(
    const T *__restrict__ X, unsigned x_bytes, const T *__restrict__ dY, unsigned y_bytes, float *__restrict__ dW, int H, int W,
    int Cin, int ldx, int Cout, int ldy, int ksize, int M, int CT, int NT, int mchunk, int remap, int direct) {
    constexpr int VEC = 16 / sizeof(T);
    constexpr int BKP = BKPv ? BKPv : (sizeof(T) == 2 ? 32 : 16);      // pixels per reduction tile
    constexpr int XROWB = BC * sizeof(T), YROWB = BNN * sizeof(T);   // bytes per pixel row of a tile
    constexpr int XCH = XROWB / 16, YCH = YROWB / 16;  // 16-byte chunks per row
    constexpr int XRPI = 1024 / XROWB, YRPI = 1024 / YROWB;           // pixel rows per DMA instruction
    constexpr int X_IT = BKP / XRPI / NW, Y_IT = BKP / YRPI / NW;     // DMA instructions per wave per tile
    constexpr int LOADS = X_IT + Y_IT;
    constexpr int NSTAGE = BKPv == 64 ? 2 : 3;            // (64-pixel reduction tiles: two stages keep the ring at the 32-pixel variant's LDS footprint)
    constexpr int XBYTES = BKP * XROWB, STAGE = XBYTES + BKP * YROWB;
    constexpr int WGM = NW / 2;                        // waves arranged WGM x 2 over the (c, n) tile
    constexpr int TM = BC / WGM / 32, TN = BNN / 64;
    constexpr int KSTEP = sizeof(T) == 2 ? 16 : 2;
    static_assert(X_IT >= 1 && Y_IT >= 1, "tile too small for one DMA piece per wave");
    // swizzle: chunk ^= 4 * ((row / rows_per_bank_line) % min(4, row_bytes / 64)): rows of >= 256 B all start on bank 0,
    // so the 4 pixel rows of a transpose read are moved to 4 different bank quarters
    constexpr int XRPL = XROWB >= 256 ? 1 : 256 / XROWB, XSWM = XROWB >= 256 ? 4 : XROWB / 64;
    constexpr int YRPL = YROWB >= 256 ? 1 : 256 / YROWB, YSWM = YROWB >= 256 ? 4 : YROWB / 64;

    __shared__ __attribute__((aligned(1024))) unsigned char smem[NSTAGE * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    static_assert(TM >= 1 && TN >= 1, "tile");
    // Block -> (tile, pixel range).  All tiles (tap, c-tile, n-tile) of one pixel range read the same X / dY
    // rows, so they are placed on ONE XCD (the dispatcher puts block b on XCD b % 8; observed, speed only):
    // XCD x walks the pixel ranges y = x, x+8, ... and for each runs all tiles back to back, which keeps a
    // range's rows in that XCD's private L2 instead of fetching them into up to 8 L2s.
    const int taps = ksize * ksize;
    const int tgroups = PAIR ? (taps + 1) / 2 : taps;         // tile rows along the tap axis
    const int ntiles = tgroups * CT * NT;
    int bx, by;
    int nt, ct, tgi;                                           // filter tile, channel tile, tap (group) index
    if (remap == 2) {
        // ONE pixel range (the tile grid alone covers the chip: 13x13 stages).  Every tile then streams the whole X column tile and dY
        // filter tile, so what the XCD's L2 can share is decided by the ORDER of the tiles: XCD x (blocks b % 8 == x) takes a contiguous
        // run of tiles in the order (channel tile, filter tile, tap) with the tap fastest -- the nine taps of a (ct, nt) pair run together
        // and read the same X and dY bytes at the same time, and an XCD works through few channel tiles (conv20: 3 of 24, 2 MB of X).
        // With the plain order (filter tile fastest, tap slowest) every XCD re-fetched all of X once per tap: 778 MB of fabric traffic
        // for 22 MB of operands on the 3072-channel layer (profiles/r02_hbm_traffic_pmc.md row 83).
        const int ntile = gridDim.x, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, q = ntile >> 3, r = ntile & 7;
        int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;       // bijective for any grid size
        tgi = t % tgroups; t /= tgroups;
        nt = t % NT;
        ct = t / NT;
        by = 0;
    } else {
        if (remap) {
            const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
            by = (idx / ntiles) * 8 + xcd;
            bx = idx % ntiles;
        } else {
            bx = blockIdx.x % ntiles;
            by = blockIdx.x / ntiles;
        }
        nt = bx % NT; bx /= NT;
        ct = bx % CT;
        tgi = bx / CT;
    }
    const int tap = PAIR ? 2 * tgi : tgi;                      // PAIR: first tap of the pair
    const int pad = ksize >> 1;
    const int dh = tap / ksize - pad, dw = tap % ksize - pad;
    const int tap2 = tap + 1;                                  // PAIR: second tap (may not exist: taps is odd)
    const int dh2 = tap2 / ksize - pad, dw2 = tap2 % ksize - pad;
    const int c0 = ct * BC, n0 = nt * BNN;
    const int mbeg = by * mchunk;
    const int mend = min(M, mbeg + mchunk);
    if (mbeg >= mend) return;

    // the dY descriptor ends at the last pixel of this block's range: rows beyond it read as zero
    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(X), 0, x_bytes, 0x00020000);
    const unsigned y_lim = min(y_bytes, (unsigned)mend * (unsigned)ldy * (unsigned)sizeof(T));
    const __amdgpu_buffer_rsrc_t rsrcY = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(dY), 0, y_lim, 0x00020000);

    // per-lane DMA descriptors
    const double rcp_hw = 1.0 / (double)(H * W);
    const float rcp_w = 1.0f / (float)W;
    unsigned x_voff[X_IT], y_voff[Y_IT];
    int xh[X_IT], xw[X_IT], ldh[X_IT], ldw[X_IT];
#pragma unroll
    for (int i = 0; i < X_IT; ++i) {
        const int r = (wave * X_IT + i) * XRPI + lane / XCH;                 // pixel row inside the tile
        const int chunk = (lane % XCH) ^ (((r / XRPL) % XSWM) * 4);          // source-side swizzle
        const bool second = PAIR && chunk * VEC >= 32;                       // this lane's chunk belongs to the pair's second tap
        const int c = PAIR ? (chunk * VEC) & 31 : c0 + chunk * VEC;
        ldh[i] = second ? dh2 : dh;
        ldw[i] = second ? dw2 : dw;
        const int m = mbeg + r;
        // (h, w) of pixel m without integer division (see conv_igemm.hip): reciprocal estimate + one correction step
        const int HW = H * W;
        int rem = m - (int)((double)m * rcp_hw) * HW;
        if (rem < 0) rem += HW; else if (rem >= HW) rem -= HW;
        int hh = (int)((float)rem * rcp_w);
        int ww = rem - hh * W;
        if (ww < 0) { ww += W; --hh; } else if (ww >= W) { ww -= W; ++hh; }
        xh[i] = hh + ldh[i];                                                 // shifted coordinates of this row's pixel
        xw[i] = ww + ldw[i];
        // channel chunk beyond Cin (or beyond the padded pixel stride, or a pair's missing second tap): permanently out of range
        x_voff[i] = (c < Cin && c < ldx && !(second && tap2 >= taps))
                        ? (unsigned)((long)(m + ldh[i] * W + ldw[i]) * ldx + c) * (unsigned)sizeof(T) : Y2_OOB;
    }
#pragma unroll
    for (int i = 0; i < Y_IT; ++i) {
        const int r = (wave * Y_IT + i) * YRPI + lane / YCH;
        const int chunk = (lane % YCH) ^ (((r / YRPL) % YSWM) * 4);
        const int n = n0 + chunk * VEC;
        y_voff[i] = (n < Cout && n < ldy) ? (unsigned)((long)(mbeg + r) * ldy + n) * (unsigned)sizeof(T) : Y2_OOB;
    }
    const int adv_h = (BKP / W) % H, adv_w = BKP % W;     // adv_h + 1 <= H is needed for the single row wrap below
    const unsigned x_step = (unsigned)BKP * (unsigned)ldx * (unsigned)sizeof(T);
    const unsigned y_step = (unsigned)BKP * (unsigned)ldy * (unsigned)sizeof(T);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (mend - mbeg + BKP - 1) / BKP;
    int kt_issue = 0, i_stage = 0;
    auto issue_next = [&]() {
        unsigned char *Xs = smem + i_stage * STAGE;
        unsigned char *Ys = Xs + XBYTES;
#pragma unroll
        for (int i = 0; i < X_IT; ++i) {
            // (rows past this block's pixel range need no test: their dY rows read as zero through rsrcY's limit)
            const bool ok = (unsigned)xh[i] < (unsigned)H && (unsigned)xw[i] < (unsigned)W;
            const unsigned voff = ok ? x_voff[i] : Y2_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (__attribute__((address_space(3))) void *)(Xs + (wave * X_IT + i) * 1024), 16, voff, 0, 0, 0);
            // advance this row slot by BKP pixels
            x_voff[i] += x_step;           // OOB + k*step stays >= 2^31 for every step taken (operands < 2^31 bytes)
            xw[i] += adv_w;                 // BKP pixels further = adv_h rows + adv_w columns (branch-free: this runs per DMA piece)
            xh[i] += adv_h;
            const bool wrap = xw[i] - ldw[i] >= W;
            xw[i] -= wrap ? W : 0;
            xh[i] += wrap ? 1 : 0;
            xh[i] -= (xh[i] - ldh[i] >= H) ? H : 0;
        }
#pragma unroll
        for (int i = 0; i < Y_IT; ++i) {
            const unsigned yv = y_voff[i];   // (a temporary: hipcc 7.2 drops the kernel's host stub when the captured array element is passed directly)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcY, (__attribute__((address_space(3))) void *)(Ys + (wave * Y_IT + i) * 1024), 16, yv, 0, 0, 0);
            y_voff[i] += y_step;
        }
        ++kt_issue;
        i_stage = (i_stage + 1 == NSTAGE) ? 0 : i_stage + 1;
    };
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
        if (kt_issue < nk) issue_next();

    const int g = lane >> 4, t = lane & 15;
    int c_stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const int ahead = min(NSTAGE - 2, kt_issue - 1 - kt);
        if (ahead >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt_issue < nk) issue_next();
        const unsigned char *Xs = smem + c_stage * STAGE;
        const unsigned char *Ys = Xs + XBYTES;
        c_stage = (c_stage + 1 == NSTAGE) ? 0 : c_stage + 1;
        if constexpr (sizeof(T) == 2 && VARIANT == 0) {
#if defined(__HIP_DEVICE_COMPILE__)
            // hardware transpose read: lane (g,t) supplies the address of pixel row (t>>2), channel quad (t&3) of its group's
            // 4x16 block and receives 4 pixels of channel t.  Issued through inline asm (common.h y2_tr16_read): the builtin form
            // made hipcc put `s_waitcnt vmcnt(0)` in front of the first read of every tile -- the DMA ring was drained each
            // iteration and the counted vmcnt above never had anything to count.  All reads of the tile are issued first, each K
            // step's MFMAs wait only for their own fragments.
            // Address of read (ks, r) = the lane's address of read (0, 0) + (16 ks + 4 r) pixel rows: the swizzle term depends on the
            // pixel row modulo RPL * SWM = 4 only, so it is the same for every (ks, r) and the row step goes into the instruction's
            // offset field -- one address add per fragment row and K step instead of one per read (12 of the ~85 instructions a
            // wave spends per four MFMAs; the kernels are instruction-issue bound, profiles/r02_igemm_tap.md section 3).
            constexpr int KSN = BKP / 16;
            static_assert(4 % (XRPL * XSWM) == 0 && 4 % (YRPL * YSWM) == 0, "the swizzle period (in pixel rows) divides the 4-row read step");
            u32x2 ra[KSN][TM][2], rb[KSN][TN][2];
            unsigned xa[TM], ya[TN];
            {
                const int px = 8 * (g >> 1) + (t >> 2);
                const int co = 16 * (g & 1) + 4 * (t & 3);                   // channel offset inside a 32-wide MFMA tile
                const int xsw = ((px / XRPL) % XSWM) * 4, ysw = ((px / YRPL) % YSWM) * 4;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int ch = (wm * TM + i) * 32 + co;
                    xa[i] = y2_lds_addr(Xs + px * XROWB + (((ch >> 3) ^ xsw) << 4) + ((ch & 7) << 1));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int ch = (wn * TN + j) * 32 + co;
                    ya[j] = y2_lds_addr(Ys + px * YROWB + (((ch >> 3) ^ ysw) << 4) + ((ch & 7) << 1));
                }
            }
#pragma unroll
            for (int ks = 0; ks < KSN; ++ks)
#pragma unroll
                for (int r = 0; r < 2; ++r) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        if (ks == 0 && r == 0) ra[ks][i][r] = y2_tr16_read_off<0>(xa[i]);
                        else if (ks == 0) ra[ks][i][r] = y2_tr16_read_off<4 * XROWB>(xa[i]);
                        else if (ks == 1 && r == 0) ra[ks][i][r] = y2_tr16_read_off<16 * XROWB>(xa[i]);
                        else if (ks == 1) ra[ks][i][r] = y2_tr16_read_off<20 * XROWB>(xa[i]);
                        else if (ks == 2 && r == 0) ra[ks][i][r] = y2_tr16_read_off<32 * XROWB>(xa[i]);
                        else if (ks == 2) ra[ks][i][r] = y2_tr16_read_off<36 * XROWB>(xa[i]);
                        else if (r == 0) ra[ks][i][r] = y2_tr16_read_off<48 * XROWB>(xa[i]);
                        else ra[ks][i][r] = y2_tr16_read_off<52 * XROWB>(xa[i]);
                    }
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if (ks == 0 && r == 0) rb[ks][j][r] = y2_tr16_read_off<0>(ya[j]);
                        else if (ks == 0) rb[ks][j][r] = y2_tr16_read_off<4 * YROWB>(ya[j]);
                        else if (ks == 1 && r == 0) rb[ks][j][r] = y2_tr16_read_off<16 * YROWB>(ya[j]);
                        else if (ks == 1) rb[ks][j][r] = y2_tr16_read_off<20 * YROWB>(ya[j]);
                        else if (ks == 2 && r == 0) rb[ks][j][r] = y2_tr16_read_off<32 * YROWB>(ya[j]);
                        else if (ks == 2) rb[ks][j][r] = y2_tr16_read_off<36 * YROWB>(ya[j]);
                        else if (r == 0) rb[ks][j][r] = y2_tr16_read_off<48 * YROWB>(ya[j]);
                        else rb[ks][j][r] = y2_tr16_read_off<52 * YROWB>(ya[j]);
                    }
                }
#pragma unroll
            for (int ks = 0; ks < KSN; ++ks) {
                constexpr int PER = 2 * (TM + TN);                         // reads per K step
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    if (ks == 0 && KSN == 2) y2_lgkm_wait2<PER>(ra[ks][i][0], ra[ks][i][1]);
                    else if (ks + 2 == KSN) y2_lgkm_wait2<PER>(ra[ks][i][0], ra[ks][i][1]);
                    else y2_lgkm_wait2<0>(ra[ks][i][0], ra[ks][i][1]);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (ks + 2 == KSN) y2_lgkm_wait2<PER>(rb[ks][j][0], rb[ks][j][1]);
                    else y2_lgkm_wait2<0>(rb[ks][j][0], rb[ks][j][1]);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y2_frag16(ra[ks][i][0], ra[ks][i][1]), y2_frag16(rb[ks][j][0], rb[ks][j][1]), acc[i][j], 0, 0, 0);
            }
#endif
        } else {
#pragma unroll
        for (int ks = 0; ks < BKP / KSTEP; ++ks) {
            if constexpr (sizeof(T) == 2) {
                bf16x8 af[TM], bf[TN];
                {
                    // reference gather (slow, layout-proof): 8 scalar LDS reads per fragment
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int px = ks * 16 + 8 * (lane >> 5) + e;
                        const int xsw = ((px / XRPL) % XSWM) * 4, ysw = ((px / YRPL) % YSWM) * 4;
#pragma unroll
                        for (int i = 0; i < TM; ++i) {
                            const int ch = (wm * TM + i) * 32 + (lane & 31);
                            af[i][e] = *reinterpret_cast<const bf16 *>(Xs + px * XROWB + (((ch >> 3) ^ xsw) << 4) + ((ch & 7) << 1));
                        }
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const int ch = (wn * TN + j) * 32 + (lane & 31);
                            bf[j][e] = *reinterpret_cast<const bf16 *>(Ys + px * YROWB + (((ch >> 3) ^ ysw) << 4) + ((ch & 7) << 1));
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
            } else {
                float af[TM], bf[TN];
                const int px = ks * 2 + (lane >> 5);
                const int xsw = ((px / XRPL) % XSWM) * 4, ysw = ((px / YRPL) % YSWM) * 4;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int ch = (wm * TM + i) * 32 + (lane & 31);
                    af[i] = *reinterpret_cast<const float *>(Xs + px * XROWB + (((ch >> 2) ^ xsw) << 4) + ((ch & 3) << 2));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int ch = (wn * TN + j) * 32 + (lane & 31);
                    bf[j] = *reinterpret_cast<const float *>(Ys + px * YROWB + (((ch >> 2) ^ ysw) << 4) + ((ch & 3) << 2));
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        }
    }

        }
    // epilogue: rows = input channels c, cols = filters n; dW is HWIO [tap][Cin][Cout]
    // PAIR: WGM = 2 and TM = 1, so wave row wm holds exactly one tap of the pair (rows 32*wm .. 32*wm+31 = its 32 channels)
    const int my_tap = PAIR ? tap + wm : tap;
    if (PAIR && my_tap >= taps) return;
    float *out = dW + (long)my_tap * Cin * Cout;
    auto write_tile = [&](auto checked_tag) {      // interior tiles skip the per-element channel tests (a branch each)
        constexpr bool CHECKED = decltype(checked_tag)::value;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 32 + (lane & 31);
            if (CHECKED && n >= Cout) continue;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int cb = (PAIR ? 0 : c0 + (wm * TM + i) * 32) + 4 * (lane >> 5);
                float *col = out + (long)cb * Cout + n;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dc = (r & 3) + 8 * (r >> 2);
                    if (!CHECKED || cb + dc < Cin) {
                        if (direct) col[(long)dc * Cout] = acc[i][j][r];      // single pixel range: this block owns the element
                        else unsafeAtomicAdd(col + (long)dc * Cout, acc[i][j][r]);
                    }
                }
            }
        }
    };
    if ((PAIR ? Cin == 32 : c0 + BC <= Cin) && n0 + BNN <= Cout) write_tile(std::false_type{});
    else write_tile(std::true_type{});
}

// tests / A-B only, process-wide by design: 0 = product rule; 1 = scalar reference gather (layout-proof, slow); 2 = the per-tap transpose-read
// kernel of this file for every shape (round 4's product path); 10 + v = conv_wgrad3.hip's variant v (0..2) for every 3x3 bf16 shape it can take
static std::atomic<int> g_wgrad_variant_raw{y2_env_int("YOLO2_WGRAD_VARIANT", 0)};
static std::atomic<int> g_wgrad_variant{0};          // what this file's kernels see: 1 = scalar gather, else 0
extern "C" void yolo2_debug_set_wgrad_variant(int v) {
    g_wgrad_variant_raw = v;
    g_wgrad_variant = (v == 1) ? 1 : 0;
}
static int wgrad_cus() {
    static int cached_cus = 0;
    if (!cached_cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        cached_cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return cached_cus;
}
// the row-of-taps kernel (conv_wgrad3.hip) takes the bf16 3x3 layers beyond the image layer
static Y2W3Plan wgrad_row_plan(int B, int H, int W, int Cin, int Cout, int ksize, int dtype) {
    Y2W3Plan none = {};
    none.variant = -1;
    const int raw = g_wgrad_variant_raw.load(std::memory_order_relaxed);
    if (dtype != YOLO2_BF16 || ksize != 3 || raw == 1 || raw == 2) return none;
    return y2_wgrad3_plan(B, H, W, Cin, Cout, wgrad_cus(), raw >= 10 ? raw - 10 : -1);
}
// plan of the calling thread's most recent launch: {BC, BNN, waves, PAIR, pixel ranges, XCD remap, blocks, direct store}
static thread_local int g_last_wgrad_plan[8] = {0, 0, 0, 0, 0, 0, 0, 0};
extern "C" int yolo2_debug_last_wgrad_plan(int *out8) {
    if (!out8) return YOLO2_E_ARG;
    for (int i = 0; i < 8; ++i) out8[i] = g_last_wgrad_plan[i];
    return YOLO2_OK;
}

// host-side planning of conv_wgrad3.hip, for tests without a GPU: out9 = {variant, ranges, padded pixels per range, blocks, remap, direct, BC, BN, waves}
extern "C" int yolo2_debug_wgrad_row_plan(int B, int H, int W, int Cin, int Cout, int cus, int force_variant, int *out9) {
    if (!out9) return YOLO2_E_ARG;
    const Y2W3Plan p = y2_wgrad3_plan(B, H, W, Cin, Cout, cus, force_variant);
    const int v[9] = {p.variant, p.ks, p.qchunk, p.blocks, p.remap, p.direct, p.BC, p.BN, p.waves};
    for (int i = 0; i < 9; ++i) out9[i] = v[i];
    return YOLO2_OK;
}
extern "C" int yolo2_debug_magic_u32(unsigned d, unsigned *m, unsigned *s) {
    if (d < 2 || !m || !s) return YOLO2_E_ARG;
    y2_magic_u32(d, m, s);
    return YOLO2_OK;
}

struct WgradPlan { int tiles, ks, remap, mchunk, blocks; };
// Pixel-range split (every block ends with one f32 atomic per output element, so fewer, longer blocks =
// less atomic traffic; more blocks = more latency hiding).  Measured on the Darknet-19 shapes
// (profiles/r01_wgrad_mapping_ab.txt): with >= 8 ranges the XCD-local placement wins (+35..45 %) at ~512
// blocks for the 128-wide tile / ~1024 for the 64-wide one; with fewer ranges (13x13 stages) it would
// leave XCDs idle, so those keep the plain mapping (~512 blocks of 8 waves for the 128-wide tile).
static bool wgrad_pairs_taps(int Cin, int ksize) { return ksize == 3 && Cin <= 32 && g_wgrad_variant == 0; }     // 64-row tile = two taps x 32 channels

static WgradPlan wgrad_plan(int M, int Cin, int Cout, int ksize, int BC, int BNN, int BKP) {
    WgradPlan p;
    const int CT = cdiv(Cin, BC), NT = cdiv(Cout, BNN);
    const int tgroups = (BC == 64 && wgrad_pairs_taps(Cin, ksize)) ? (ksize * ksize + 1) / 2 : ksize * ksize;
    p.tiles = tgroups * CT * NT;
    static const int env_target = y2_env_int("YOLO2_WGRAD_BLOCKS", 0);
    const int env_remap = -1;      // (XCD-local placement by rule below)
    // 128-wide tile (8 waves, two workgroups per CU): every block of the grid should be resident at once -- AT MOST 7/4 blocks per CU --
    // and keep >= 40 reduction tiles, else the 64 KB atomic epilogue and the ring prologue dominate.  26x26 256->512 (72 tiles),
    // profiles/r03_wgrad_split_sweep.txt: 8 ranges = 576 blocks 55.7 us, 6 = 432 blocks 49.6 us at batch 16; 42.2 -> 34.3 us at batch 8
    // with 4; 84.8 -> 79.6 us at batch 32 with 6.  64-wide tile: >= 8 reduction tiles per block.
    const int resident = 1;
    const bool res = BC >= 128 && resident && env_target <= 0;
    const int max_ks = res ? (M / (40 * BKP) > 1 ? M / (40 * BKP) : 1) : cdiv(M, 8 * BKP);
    // 64-wide tile: ~1024 blocks for the 3x3 layers, ~384 for the 1x1 layers (9x fewer tiles: more, shorter pixel ranges only add
    // atomic traffic; measured 19.6 -> 15.3 us on the 26x26 1x1 layers)
    const int def_target = BC >= 128 ? (res ? 448 : 512) : (ksize == 1 ? 384 : 1024);
    int target = env_target > 0 ? env_target : def_target;
    int ks = res ? (target / p.tiles > 1 ? target / p.tiles : 1) : cdiv(target, p.tiles);
    int remap = env_remap >= 0 ? env_remap : (ks >= 8 && max_ks >= 8);
    if (remap) {
        ks = res ? ks / 8 * 8 : cdiv(ks, 8) * 8;
        if (ks > max_ks) ks = max_ks / 8 * 8;
        if (ks < 8) remap = 0;
    }
    if (!remap) {
        if (env_target <= 0) target = def_target;
        ks = res ? (target / p.tiles > 1 ? target / p.tiles : 1) : cdiv(target, p.tiles);
        if (ks > max_ks) ks = max_ks;
    }
    // a tile grid that already covers the chip takes ONE pixel range: no atomics, plain stores (measured: the atomic epilogue
    // of a 2-way split costs more than the second workgroup per tile gains)
    if (p.tiles >= 256) { ks = 1; remap = 2; }      // (L2-stationary tile order: profiles/r03_l2_stationary_ab.md)
    if (ks < 1) ks = 1;
    p.mchunk = cdiv(cdiv(M, ks), BKP) * BKP;
    ks = cdiv(M, p.mchunk);
    if (remap == 1 && ks % 8 != 0) remap = (ks >= 8);         // rounding may have changed the count; ragged tail is fine
    p.ks = ks;
    p.remap = remap;
    p.blocks = remap == 1 ? p.tiles * (cdiv(ks, 8) * 8) : p.tiles * ks;
    return p;
}

template <typename T, int BC, int BNN, int BKPv = 0>
static void launch_wgrad(const void *X, const void *dY, float *dW, int B, int H, int W, int Cin, int ldx,
                         int Cout, int ldy, int ksize, hipStream_t st) {
    const int M = B * H * W;
    const int CT = cdiv(Cin, BC), NT = cdiv(Cout, BNN);
    constexpr int BKP = BKPv ? BKPv : (sizeof(T) == 2 ? 32 : 16);
    const WgradPlan pl = wgrad_plan(M, Cin, Cout, ksize, BC, BNN, BKP);
    const int mchunk = pl.mchunk, remap = pl.remap;
    const int direct = pl.ks == 1;                            // one pixel range: plain stores, dW need not be zeroed
    dim3 grid(pl.blocks);
    const unsigned x_bytes = (unsigned)((size_t)M * ldx * sizeof(T)), y_bytes = (unsigned)((size_t)M * ldy * sizeof(T));
    const int nw8 = 1;      // (8-wave workgroups for the 128-wide tile: profiles/r01_wgrad_variants.txt)
    {
        const bool w8 = BC >= 128 && g_wgrad_variant == 0 && nw8, pair = BC == 64 && g_wgrad_variant == 0 && wgrad_pairs_taps(Cin, ksize);
        const int plan[8] = {BC, BNN, w8 ? 8 : 4, pair ? 1 : 0, pl.ks, remap, pl.blocks, direct};
        for (int i = 0; i < 8; ++i) g_last_wgrad_plan[i] = plan[i];
    }
    if constexpr (BC >= 128) {
        if (g_wgrad_variant == 0 && nw8) {
            conv_wgrad_kernel<T, BC, BNN, 0, 8, BKPv><<<grid, 512, 0, st>>>((const T *)X, x_bytes, (const T *)dY, y_bytes, dW, H, W, Cin, ldx, Cout, ldy, ksize, M, CT, NT, mchunk, remap, direct);
            return;
        }
    }
    if constexpr (BC == 64) {
        if (g_wgrad_variant == 0 && wgrad_pairs_taps(Cin, ksize)) {
            conv_wgrad_kernel<T, BC, BNN, 0, 4, 0, true><<<grid, 256, 0, st>>>((const T *)X, x_bytes, (const T *)dY, y_bytes, dW, H, W, Cin, ldx, Cout, ldy, ksize, M, CT, NT, mchunk, remap, direct);
            return;
        }
    }
    if (g_wgrad_variant == 0)
        conv_wgrad_kernel<T, BC, BNN, 0, 4, BKPv><<<grid, 256, 0, st>>>((const T *)X, x_bytes, (const T *)dY, y_bytes, dW, H, W, Cin, ldx, Cout, ldy, ksize, M, CT, NT, mchunk, remap, direct);
    else
        conv_wgrad_kernel<T, BC, BNN, 1><<<grid, 256, 0, st>>>((const T *)X, x_bytes, (const T *)dY, y_bytes, dW, H, W, Cin, ldx, Cout, ldy, ksize, M, CT, NT, mchunk, remap, direct);
}

static bool wgrad_small_tile(int Cin, int Cout, int ksize) {
    // few 128x128 tiles (1x1 layers, 128->256 3x3): the 64x64 tile quarters the pixel-range split and its atomic traffic
    // (26 -> 20 us on the 1x1 layers; profiles/r01_wgrad_small_tiles.txt)
    const int small_below = 33;
    return Cin <= 64 || Cout <= 64 || ksize * ksize * cdiv(Cin, 128) * cdiv(Cout, 128) < small_below;
}

extern "C" int yolo2_conv2d_wgrad_accumulates(int B, int H, int W, int Cin, int ldx, int Cout, int ldy, int ksize, int dtype) {
    if (!(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0) || !(ksize == 1 || ksize == 3) || !(dtype == YOLO2_F32 || dtype == YOLO2_BF16)) return 1;
    static const bool first_direct = y2_env_int("YOLO2_FIRST_DIRECT", 1) != 0;
    if (first_direct && Cin <= 8 && y2_first_layer_shape(8, ldx, Cout, ldy, ksize)) return 1;      // cross-workgroup atomics
    if (g_wgrad_variant != 0) return 1;
    if (g_wgrad_variant_raw.load(std::memory_order_relaxed) == 0 && y2_w32_shape(Cin, ldx, Cout, ldy, ksize, dtype)) {
        const int blocks = y2_w32_blocks(B, H, W, wgrad_cus());
        if (blocks > 0) return blocks == 1 ? 0 : 1;                  // one workgroup stores, several add
    }
    {
        const Y2W3Plan rp = wgrad_row_plan(B, H, W, Cin, Cout, ksize, dtype);
        if (rp.variant >= 0) return rp.direct ? 0 : 1;
    }
    const int bkp = dtype == YOLO2_BF16 ? 32 : 16;
    const bool small = wgrad_small_tile(Cin, Cout, ksize);
    return wgrad_plan(B * H * W, Cin, Cout, ksize, small ? 64 : 128, small ? 64 : 128, bkp).ks == 1 ? 0 : 1;
}

extern "C" int yolo2_conv2d_wgrad(const void *X, const void *dY, float *dW, int B, int H, int W, int Cin,
                                  int ldx, int Cout, int ldy, int ksize, int dtype, void *stream) {
    Y2_CHECK_ARG(X && dY && dW);
    Y2_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0);
    Y2_CHECK_ARG(ksize == 1 || ksize == 3);
    Y2_CHECK_ARG(ldx >= Cin && ldy >= Cout);
    const int vec = dtype == YOLO2_BF16 ? 8 : 4;
    const size_t esz = dtype == YOLO2_BF16 ? 2 : 4;
    Y2_CHECK_ARG(ldx % vec == 0 && ldy % vec == 0);
    Y2_CHECK_ARG(((uintptr_t)X & 15) == 0 && ((uintptr_t)dY & 15) == 0);
    // 32-bit byte offsets below 2^31 on both operands (DMA descriptors)
    Y2_CHECK_ARG((size_t)B * H * W * (size_t)(ldx > ldy ? ldx : ldy) * esz < (1ull << 31));
    hipStream_t st = (hipStream_t)stream;
    static const bool first_direct = y2_env_int("YOLO2_FIRST_DIRECT", 1) != 0;
    if (first_direct && Cin <= 8 && y2_first_layer_shape(8, ldx, Cout, ldy, ksize) && (dtype == YOLO2_F32 || dtype == YOLO2_BF16)) {
        y2_first_layer_wgrad(X, dY, dW, B, H, W, Cin, dtype, st);                       // image layer: direct kernel (conv_first.hip)
        for (int i = 0; i < 8; ++i) g_last_wgrad_plan[i] = -1;
        Y2_CHECK_LAUNCH();
        return YOLO2_OK;
    }
    if (g_wgrad_variant_raw.load(std::memory_order_relaxed) == 0 && y2_w32_shape(Cin, ldx, Cout, ldy, ksize, dtype)) {
        int blocks = 0;                                                                 // conv1: all nine taps per workgroup (conv_wgrad_c32.hip)
        if (y2_w32_wgrad(X, dY, dW, B, H, W, wgrad_cus(), &blocks, st) == 0) {
            const int plan[8] = {32, 64, 12, 9, blocks, 0, blocks, blocks == 1};
            for (int i = 0; i < 8; ++i) g_last_wgrad_plan[i] = plan[i];
            Y2_CHECK_LAUNCH();
            return YOLO2_OK;
        }
    }
    {
        const Y2W3Plan rp = wgrad_row_plan(B, H, W, Cin, Cout, ksize, dtype);
        if (rp.variant >= 0) {
            const int plan[8] = {rp.BC, rp.BN, rp.waves, 3, rp.ks, rp.remap, rp.blocks, rp.direct};      // "pair" slot 3: three taps (a kernel row) per workgroup
            for (int i = 0; i < 8; ++i) g_last_wgrad_plan[i] = plan[i];
            if (y2_wgrad3_launch(rp, X, dY, dW, B, H, W, Cin, ldx, Cout, ldy, st) != 0) { yolo2_set_error("yolo2_conv2d_wgrad: no such row-kernel variant"); return YOLO2_E_ARG; }
            Y2_CHECK_LAUNCH();
            return YOLO2_OK;
        }
    }
    const bool small = wgrad_small_tile(Cin, Cout, ksize);
    if (small) {
        Y2_DISPATCH_DTYPE(dtype, launch_wgrad<T, 64, 64>(X, dY, dW, B, H, W, Cin, ldx, Cout, ldy, ksize, st));
    } else {
        // 64-pixel reduction tiles on a 2-stage ring (8 MFMAs per wave between barriers, same 64 KiB of LDS): pays when the launch gives
        // a CU about one workgroup -- single-range grids of up to 1.5 blocks per CU (the 512 -> 1024 13x13 layers: 41.9 -> 36.6 us);
        // with two or three workgroups per CU the 32-pixel / 3-stage form wins (conv18: 66.5 vs 73.1 us).
        bool use64 = false;
        if (dtype == YOLO2_BF16 && g_wgrad_variant == 0) {
            const int cached_cus = wgrad_cus();
            const WgradPlan pl = wgrad_plan(B * H * W, Cin, Cout, ksize, 128, 128, 32);
            use64 = pl.ks == 1 && 2 * pl.blocks <= 3 * cached_cus;
        }
        if (use64 && dtype == YOLO2_BF16) launch_wgrad<bf16, 128, 128, 64>(X, dY, dW, B, H, W, Cin, ldx, Cout, ldy, ksize, st);
        else Y2_DISPATCH_DTYPE(dtype, launch_wgrad<T, 128, 128>(X, dY, dW, B, H, W, Cin, ldx, Cout, ldy, ksize, st));
    }
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}
