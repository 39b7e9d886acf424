// Filter gradient of the 3x3 NHWC convolution, one kernel ROW of taps per workgroup (gfx950, bf16).
//
// Replaces tf.gradients of slim.layers.conv2d w.r.t. its weights (reference train.py:127-129, call sites
// model/yolo2/inference.py:37-48,73-118) for the 3x3 layers; conv_wgrad.hip keeps the 1x1 layers, the f32 parity mode and the image layer.
//
//   dW[dh][dw][c][n] = sum_m X[pix(m) + (dh, dw), c] * dY[m, n]            (zero where the shifted pixel leaves the image)
//
// What the per-tap kernel of conv_wgrad.hip pays for, and what this one does instead (DESIGN.md sections 3 and 5a):
//   * it stages X once PER TAP (nine times) with the SAME-padding test in every DMA piece.  Here the reduction runs over a PADDED pixel
//     index q: every image row is followed by one zero column (row pitch W + 1).  In that index the left / right neighbour of a pixel is
//     q -+ 1 for EVERY pixel -- the neighbour of a row's first pixel is the previous row's zero column -- so one staged X tile, read at
//     row offsets -1 / 0 / +1, serves the three taps of a kernel row with no mask anywhere in the loop, and the vertical shift dh is
//     folded into the DMA source address (rows that leave the image read as zeros through the buffer range check).  W / (W + 1) of the
//     MFMA work is real (93 % at 13x13, 96 % at 26x26).
//   * its waves hold 32 x 64 of ONE tap: 3 transpose reads per MFMA.  Here a wave holds 3 taps x 32 channels x 64 filters (six 32x32
//     accumulators): the two dY fragments of a 16-pixel step are shared by the three taps, 10 reads per 6 MFMAs.
//   * every barrier puts its eight waves into the same phase.  Here two wave groups run half a super-step apart: one in its MFMA phase (the
//     next step's fragment reads in the MFMA gaps, counted lgkmcnt), the other issuing DMA and writing the next offset table.
//   * it pays one f32 atomic per output element and pixel range, ~5-7 M per launch on the 26x26 .. 104x104 layers.  Here the KW wave groups
//     of a workgroup take alternate pixel tiles of the workgroup's range and are summed through LDS before anything leaves the CU: one
//     workgroup per CU, 2.4-3.1 M atomics per launch.
// Tiles are staged pixel-major by LDS-DMA exactly as they lie in HBM and gathered with ds_read_b64_tr_b16 (layout and swizzle of
// conv_wgrad.hip, pinned on hardware by tests/test_kernels_gpu.py::test_tr16_layout; the +-1-row shifted reads are conflict-free:
// SQ_LDS_BANK_CONFLICT = 0, profiles/r05_sq_counters.md).  DMA source offsets come from a per-stage table in LDS: one thread per staged ROW
// derives (image row, column) of its padded position with two multiply-high divisions by constants -- stateless, and eight times less
// arithmetic than every lane of every 1 KiB piece doing it for its own 16 bytes.
// Measured (profiles/r05_wgrad_rule_vs_pertap*.txt): 13x13 layers 37.7 / 67.4 / 167.8 -> 31.6 / 57.9 / 159 us, 26x26 .. 104x104 layers 46-49 -> 38-39 us
// at batch 16; what still bounds it is in DESIGN.md section 5a.
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
// ABL (timing ablations, -DY2W3_EXPERIMENTS builds only; results wrong by design): 1 = no MFMA, 2 = no fragment reads, 4 = no DMA inside the loop,
// 8 = DMA pieces issued with a constant source offset (no address arithmetic), 16 = no barrier, 32 = no output stores, 64 = no wave-group reduction,
// 128 = no prologue DMA either, 256 = the workgroup returns at once (launch cost of the geometry), 512 = s_memtime stamps at the phase boundaries of
// every super-step; each wave leaves {kernel, wait + barrier, DMA issue, reads + MFMA issue, table build, prologue, epilogue} cycle sums in dW (scripts/w3_phase_cycles.py)
template <int WC, int WN, int KW, int P, int NSTAGE, int ABL = 0>
__global__ __launch_bounds__(WC *WN *KW * 64) void conv_wgrad_row_kernel(
    const bf16 *__restrict__ X, unsigned x_bytes, const bf16 *__restrict__ dY, unsigned y_bytes, float *__restrict__ dW, int H, int W, int Cin, int ldx,
    int Cout, int ldy, int Mp, int CT, int NT, int qchunk, int KS, int remap, int direct, unsigned mW, unsigned sW, unsigned mH, unsigned sH) {
    constexpr int NWAVE = WC * WN * KW;
    // G wave groups run half a super-step apart (KW >= 2): while one group issues its MFMAs -- with the fragment reads in the gaps -- the other
    // issues the DMA pieces of a later stage and computes the next offset table, so the two waves a SIMD hosts (one of each group: a
    // workgroup's waves go to the SIMDs cyclically) are in complementary phases instead of contending for the matrix pipe and then both
    // leaving it idle (measured on the lockstep form: 1536 MFMA cycles in a 2609-cycle super-step, profiles/r05_w3_phase_cycles_lockstep.txt).
    // A group stages, and reads, only its own KW / G sub-tiles of a stage.
    constexpr int G = KW >= 2 ? 2 : 1;
    constexpr int GWAVE = NWAVE / G, KWG = KW / G, NTHRG = GWAVE * 64;
    constexpr int BC = 32 * WC, BN = 64 * WN;
    constexpr int XROWB = BC * 2, YROWB = BN * 2;                    // bytes per pixel row of a tile
    constexpr int XCH = XROWB / 16, YCH = YROWB / 16;                // 16-byte chunks per row
    constexpr int XRPI = 1024 / XROWB, YRPI = 1024 / YROWB;          // pixel rows per DMA instruction (1 KiB per wave-instruction)
    constexpr int XH = XRPI / 2;                                     // halo rows in front of the X tile (>= 1: the dw = -1 tap)
    constexpr int XR = P + 2 * XH;                                   // staged X rows: padded positions q0 - XH .. q0 + P + XH - 1
    constexpr int XP = XR / XRPI, YP = P / YRPI;                     // DMA pieces per sub-tile
    constexpr int XT = XR * XROWB, YT = P * YROWB, SUB = XT + YT, STAGE = KW * SUB;
    constexpr int NXP = KWG * XP, NYP = KWG * YP;                    // pieces per stage and group, dealt round-robin to the group's waves
    constexpr int XJ = (NXP + GWAVE - 1) / GWAVE, YJ = (NYP + GWAVE - 1) / GWAVE;
    constexpr int PIECES_MIN = NXP / GWAVE + NYP / GWAVE;            // what EVERY wave has issued per stage (some issue one more of each kind)
    constexpr int KSN = P / 16;                                      // 16-pixel MFMA steps per tile
    // swizzle (conv_wgrad.hip): chunk ^= 4 * ((row / rows_per_bank_line) % min(4, row_bytes / 64)) -- the 4 pixel rows of a transpose read land on 4
    // different bank quarters whatever row the read starts at (the tap offsets -1 / +1 shift the start row; measured: SQ_LDS_BANK_CONFLICT = 0)
    constexpr int XRPL = XROWB >= 256 ? 1 : 256 / XROWB, XSWM = XROWB >= 256 ? 4 : (XROWB >= 64 ? XROWB / 64 : 1);
    constexpr int YRPL = YROWB >= 256 ? 1 : 256 / YROWB, YSWM = YROWB >= 256 ? 4 : YROWB / 64;
    static_assert(KW == 1 || KW % 2 == 0, "two wave groups");
    static_assert(XH >= 1 && XR % XRPI == 0 && P % YRPI == 0 && P % 16 == 0, "tile geometry");
    static_assert(4 % (XRPL * XSWM) == 0 && 4 % (YRPL * YSWM) == 0, "the swizzle period (in pixel rows) divides the 4-row read step");
    static_assert(XRPI % 4 == 0 && YRPI % 4 == 0, "a lane's swizzle term is the same in every piece");
    static_assert(NSTAGE >= 2 && NSTAGE <= 4, "LDS ring");
    static_assert(KW == 1 || (KW / 2) * WC * WN * 6 * 4096 <= NSTAGE * STAGE, "the accumulator images of the wave-group reduction fit the ring");

    // Source-offset table (DMA address generation): one u32 per staged ROW of a group's stage -- KWG * XR X rows, then KWG * P dY rows, then one
    // permanently-out-of-range slot -- double-buffered by stage parity, one pair per group.  Each row's offset is computed ONCE (one thread
    // per row, a stage ahead of its use) instead of once per 16-byte chunk of the row by every lane of every DMA piece: the 64 lanes of a
    // piece cover only 4-16 rows, and that 8-fold redundant arithmetic was 5 VALU + 2.4 SALU per MFMA (profiles/r05_w3_sq_counters_v1.md).
    constexpr int NENT_X = KWG * XR, NENT = NENT_X + KWG * P;        // table entries per stage and group
    constexpr int NEPT = (NENT + 1 + NTHRG - 1) / NTHRG;             // entries a thread computes per stage (the out-of-range slot included)
    constexpr int TBLB = ((NENT + 1) * 4 + 15) / 16 * 16;            // bytes per group and parity
    static_assert(NSTAGE * STAGE + G * 2 * TBLB <= 160 * 1024, "LDS: ring + offset tables");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NSTAGE * STAGE + G * 2 * TBLB];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned long long tm_start = 0;
    if constexpr (ABL & 512) tm_start = __builtin_amdgcn_s_memtime();
    const int kw = wave / (WC * WN), wcn = wave % (WC * WN), wc = wcn / WN, wn = wcn % WN;
    const int grp = wave / GWAVE, gwv = wave % GWAVE, gtid = tid - grp * NTHRG;      // wave group, wave and thread inside it

    // ---- block -> (kernel row, channel tile, filter tile, pixel range).  Workgroups of one pixel range read the same X / dY rows: with >= 8
    // ranges they are placed on ONE XCD (block b runs on XCD b % 8: observed, speed only); a single range orders the tiles (channel tile,
    // filter tile, kernel row) with the row fastest, each XCD taking a contiguous run (conv_wgrad.hip remap 2).
    const int ncols = 3 * CT * NT;
    int col, by;
    if (remap == 2) {
        // contiguous runs of the (range-major) workgroup list per XCD, bijective for any grid size: one range (13x13 stages) = the L2-stationary tile
        // order; several ranges = every XCD works through one or two ranges' tiles back to back, whatever the range count (remap 1 needs a multiple
        // of 8 ranges to load the XCDs evenly)
        const int ntile = gridDim.x, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, qq = ntile >> 3, rr = ntile & 7;
        const int w = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
        by = w / ncols;
        col = w - by * ncols;
    } else if (remap == 1) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        by = (idx / ncols) * 8 + xcd;
        col = idx % ncols;
    } else {
        col = blockIdx.x % ncols;
        by = blockIdx.x / ncols;
    }
    if (by >= KS) return;
    if constexpr (ABL & 256) { if (dW[0] == 123.456f) dW[1] = (float)col; return; }
    const int dh = col % 3 - 1;
    const int nt = (col / 3) % NT, ct = col / (3 * NT);
    const int c0 = ct * BC, n0 = nt * BN;
    const int qb = by * qchunk;
    const int qe = min(Mp, qb + qchunk);
    if (qb >= qe) return;
    const int nsuper = (qe - qb + KW * P - 1) / (KW * P);

    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16 *>(X), 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcY = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16 *>(dY), 0, y_bytes, 0x00020000);

    // ---- per-lane DMA constants: row inside a piece, source channel chunk (swizzled)
    const int xrow_l = lane / XCH, yrow_l = lane / YCH;
    const int xc = c0 + (((lane % XCH) ^ (((xrow_l / XRPL) % XSWM) * 4)) * 8);
    const int yc = n0 + (((lane % YCH) ^ (((yrow_l / YRPL) % YSWM) * 4)) * 8);
    const bool xc_ok = xc < Cin && xc < ldx, yc_ok = yc < Cout && yc < ldy;
    const unsigned x_cb = xc_ok ? (unsigned)xc * 2u : 0u, y_cb = yc_ok ? (unsigned)yc * 2u : 0u;
    const unsigned W1 = (unsigned)(W + 1), ldx2 = (unsigned)ldx * 2u, ldy2 = (unsigned)ldy * 2u;
    const int dhW = dh * W;
    const unsigned smem_base = y2_lds_addr(smem);
    const unsigned tbl_base = smem_base + (unsigned)(NSTAGE * STAGE + grp * 2 * TBLB);       // this group's pair of tables
    // table read address of this lane per DMA piece (parity 0; parity 1 = + TBLB through the instruction's offset field): the row's entry, or
    // the out-of-range slot for lanes whose channel chunk lies beyond the tensor
    unsigned tx[XJ], ty[YJ];
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
        const int p = gwv + j * GWAVE, sub = p / XP, pp = p - sub * XP;
        tx[j] = tbl_base + 4u * (unsigned)((xc_ok && p < NXP) ? sub * XR + pp * XRPI + xrow_l : NENT);
    }
#pragma unroll
    for (int j = 0; j < YJ; ++j) {
        const int p = gwv + j * GWAVE, sub = p / YP, pp = p - sub * YP;
        ty[j] = tbl_base + 4u * (unsigned)((yc_ok && p < NYP) ? NENT_X + sub * P + pp * YRPI + yrow_l : NENT);
    }
    // the entries this thread computes per stage: position offset of the row relative to the stage's first padded position
    int ent_off[NEPT];
    unsigned ent_addr[NEPT];
#pragma unroll
    for (int k = 0; k < NEPT; ++k) {
        const int e = gtid + k * NTHRG;
        ent_off[k] = grp * (KWG * P) + (e < NENT_X ? (e / XR) * P - XH + e % XR : e - NENT_X);      // (dY rows: sub * P + i = e - NENT_X)
        ent_addr[k] = tbl_base + 4u * (unsigned)(e < NENT ? e : NENT);          // (threads beyond the table rewrite the out-of-range slot)
    }
    // writes this group's table of stage t (parity t & 1).  Stateless: (image row, column) of a padded position by two multiply-high divisions.
    auto build_table = [&](int t) {
#if defined(__HIP_DEVICE_COMPILE__)
        const int qs = qb + t * (KW * P);
        const unsigned par = (t & 1) ? (unsigned)TBLB : 0u;
#pragma unroll
        for (int k = 0; k < NEPT; ++k) {
            const int e = gtid + k * NTHRG;
            const unsigned q = (unsigned)(qs + ent_off[k]);           // wraps below 0: fails the range test
            const unsigned R = y2_div_magic(q, mW, sW);               // image row counted over the whole batch
            const unsigned c = q - __umul24(R, W1);                   // column; W = the zero column
            const unsigned r = R - __umul24(y2_div_magic(R, mH, sH), (unsigned)H);
            const bool isx = e < NENT_X;
            // X rows: the pixel dh image rows away (zero outside the image); dY rows: the pixel itself, zero beyond this block's range
            const bool ok = (c < (unsigned)W) & (isx ? (q < (unsigned)Mp) & ((unsigned)((int)r + dh) < (unsigned)H) : (q < (unsigned)qe)) & (e < NENT);
            const unsigned src = isx ? __umul24((unsigned)((int)(q - R) + dhW), ldx2) : __umul24(q - R, ldy2);
            const unsigned v = ok ? src : Y2_OOB;
            asm volatile("ds_write_b32 %0, %1" ::"v"(ent_addr[k] + par), "v"(v) : "memory");
        }
#endif
    };

    int t_issue = 0, i_slot = 0;
    // DMA pieces of this group's sub-tiles of stage t_issue: every lane reads its row's source offset from the table (inline asm: a compiler-visible
    // LDS read would be preceded by vmcnt(0) while DMA is in flight, common.h), adds its channel chunk and issues the 1 KiB piece
    auto issue_stage = [&](auto par_) {
#if defined(__HIP_DEVICE_COMPILE__)
        constexpr int PAR = decltype(par_)::value * TBLB;
        unsigned char *base = smem + i_slot * STAGE + grp * (KWG * SUB);
        unsigned ex[XJ], ey[YJ];
#pragma unroll
        for (int j = 0; j < XJ; ++j) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(ex[j]) : "v"(tx[j]), "n"(PAR) : "memory");
#pragma unroll
        for (int j = 0; j < YJ; ++j) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(ey[j]) : "v"(ty[j]), "n"(PAR) : "memory");
#pragma unroll
        for (int j = 0; j < XJ; ++j) {
            const int p = gwv + j * GWAVE;                            // wave-uniform
            if (p < NXP) {
                const int sub = p / XP, pp = p - sub * XP;
                asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(ex[j]) : "n"(XJ + YJ - 1 - j) : "memory");
                const unsigned voff = (ABL & 8) ? x_cb + (unsigned)lane * 16u : ex[j] + x_cb;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (__attribute__((address_space(3))) void *)(base + sub * SUB + pp * 1024), 16, voff, 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < YJ; ++j) {
            const int p = gwv + j * GWAVE;
            if (p < NYP) {
                const int sub = p / YP, pp = p - sub * YP;
                asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(ey[j]) : "n"(YJ - 1 - j) : "memory");
                const unsigned voff = (ABL & 8) ? y_cb + (unsigned)lane * 16u : ey[j] + y_cb;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcY, (__attribute__((address_space(3))) void *)(base + sub * SUB + XT + pp * 1024), 16, voff, 0, 0, 0);
            }
        }
#endif
        ++t_issue;
        i_slot = (i_slot + 1 == NSTAGE) ? 0 : i_slot + 1;
    };
    // prologue.  One group (lockstep): the first NSTAGE - 1 stages in flight, the table of the next one written.  Two groups: NSTAGE stages in
    // flight (a stage's slot is refilled right after its MFMA phase), the table of stage NSTAGE written.
    constexpr int PRO = G == 2 ? NSTAGE : NSTAGE - 1;
    if constexpr (!(ABL & 128)) {
        y2_static_for<0, PRO>([&](auto t_) {
            constexpr int t = decltype(t_)::value;
            if (t < nsuper) {
                build_table(t);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                issue_stage(std::integral_constant<int, t & 1>{});
            }
        });
        if (PRO < nsuper) build_table(PRO);
    }

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

    unsigned long long tm_k0 = tm_start, tm_pro = 0, tm_a = 0, tm_b = 0, tm_c = 0, tm_d = 0, tm_wait = 0, tm_issue = 0, tm_mfma = 0, tm_tbl = 0;
    if constexpr (ABL & 512) tm_pro = __builtin_amdgcn_s_memtime();
    int c_slot = 0;
    // the MFMA phase of one tile: 10 transpose reads + 6 MFMAs per 16-pixel step, the reads of step k + 1 issued ahead of the MFMAs of step k
    u32x2 fa[2][3][2], fb[2][2][2];        // two fragment sets: step k + 1 is read while step k multiplies
    unsigned xs[3], ys[2];                 // this wave's read addresses in the slot being consumed
    // first fragment set of the next tile, issued AHEAD of the barrier in front of its MFMA phase (two groups: from the LOAD phase, once the
    // mid-step barrier has certified the stage as landed) so that the phase does not open with an exposed LDS round trip
    auto begin_tile = [&]() {
#if defined(__HIP_DEVICE_COMPILE__)
        const unsigned sb = smem_base + (unsigned)(c_slot * STAGE);
        c_slot = (c_slot + 1 == NSTAGE) ? 0 : c_slot + 1;
#pragma unroll
        for (int d = 0; d < 3; ++d) xs[d] = sb + xa[d];
#pragma unroll
        for (int j = 0; j < 2; ++j) ys[j] = sb + ya[j];
        if constexpr (!(ABL & 2)) {      // (the order the counted waits of mfma_phase assume: X tap 0, dY 0, dY 1, X tap 1, X tap 2)
            fa[0][0][0] = y2_tr16_read_off<0>(xs[0]); fa[0][0][1] = y2_tr16_read_off<4 * XROWB>(xs[0]);
            fb[0][0][0] = y2_tr16_read_off<0>(ys[0]); fb[0][0][1] = y2_tr16_read_off<4 * YROWB>(ys[0]);
            fb[0][1][0] = y2_tr16_read_off<0>(ys[1]); fb[0][1][1] = y2_tr16_read_off<4 * YROWB>(ys[1]);
            fa[0][1][0] = y2_tr16_read_off<0>(xs[1]); fa[0][1][1] = y2_tr16_read_off<4 * XROWB>(xs[1]);
            fa[0][2][0] = y2_tr16_read_off<0>(xs[2]); fa[0][2][1] = y2_tr16_read_off<4 * XROWB>(xs[2]);
        } else {
#pragma unroll
            for (int d = 0; d < 3; ++d) { fa[0][d][0] = u32x2{xs[d], 0u}; fa[0][d][1] = u32x2{0u, xs[d]}; }
#pragma unroll
            for (int j = 0; j < 2; ++j) { fb[0][j][0] = u32x2{ys[j], 0u}; fb[0][j][1] = u32x2{0u, ys[j]}; }
        }
#endif
    };
    // The MFMA phase of one tile.  A wave in this phase is alone on its SIMD's matrix pipe (its partner is in the LOAD phase), and a wave issues in
    // order: ten reads issued in a block between two batches of MFMAs left the pipe idle for their ~70 cycles of issue, four times per tile
    // (measured: 1033 cycles per 24 MFMAs).  So the reads of step k + 1 go INTO the gaps of step k's MFMAs, two behind each of the first five
    // (<= 5 issue slots fit a 32-cycle MFMA), in the order the next step needs them (X tap 0, dY 0, dY 1, X tap 1, X tap 2), with counted waits:
    // LDS returns in order, so "at most N outstanding" names exactly the reads issued after the one needed.  sched_barrier pins the order.
    auto mfma_phase = [&]() {
#if defined(__HIP_DEVICE_COMPILE__)
#define Y2W3_MMA(bf, d, j) acc[d][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y2_frag16(fa[bf][d][0], fa[bf][d][1]), y2_frag16(fb[bf][j][0], fb[bf][j][1]), acc[d][j], 0, 0, 0)
        y2_static_for<0, KSN>([&](auto ks_) {
            constexpr int ks = decltype(ks_)::value, bf = ks & 1, nb = bf ^ 1;
            constexpr bool NEXT = ks + 1 < KSN;
            constexpr int XO = (16 * (ks + 1)) * XROWB, YO = (16 * (ks + 1)) * YROWB;
            if constexpr (ABL & 3) {       // (ablations: no reads and / or no MFMAs; the block form)
                if constexpr (NEXT && !(ABL & 2)) {
#pragma unroll
                    for (int d = 0; d < 3; ++d) { fa[nb][d][0] = y2_tr16_read_off<XO>(xs[d]); fa[nb][d][1] = y2_tr16_read_off<XO + 4 * XROWB>(xs[d]); }
#pragma unroll
                    for (int j = 0; j < 2; ++j) { fb[nb][j][0] = y2_tr16_read_off<YO>(ys[j]); fb[nb][j][1] = y2_tr16_read_off<YO + 4 * YROWB>(ys[j]); }
                } else if constexpr (NEXT) {
#pragma unroll
                    for (int d = 0; d < 3; ++d) { fa[nb][d][0] = u32x2{xs[d], 0u}; fa[nb][d][1] = u32x2{0u, xs[d]}; }
#pragma unroll
                    for (int j = 0; j < 2; ++j) { fb[nb][j][0] = u32x2{ys[j], 0u}; fb[nb][j][1] = u32x2{0u, ys[j]}; }
                }
                y2_lgkm_wait10<NEXT && !(ABL & 2) ? 10 : 0>(fa[bf][0][0], fa[bf][0][1], fa[bf][1][0], fa[bf][1][1], fa[bf][2][0], fa[bf][2][1], fb[bf][0][0], fb[bf][0][1], fb[bf][1][0], fb[bf][1][1]);
                if constexpr (ABL & 1) {
#pragma unroll
                    for (int d = 0; d < 3; ++d)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[d][j][0] += __builtin_bit_cast(float, fa[bf][d][0][0] ^ fa[bf][d][1][1] ^ fb[bf][j][0][0] ^ fb[bf][j][1][1]);
                } else {
#pragma unroll
                    for (int d = 0; d < 3; ++d)
#pragma unroll
                        for (int j = 0; j < 2; ++j) Y2W3_MMA(bf, d, j);
                }
                return;
            }
            // outstanding reads behind the one a wait names: steady state 6 / 6 / 6 / 8 (the next step's reads are already in the queue),
            // last step of the tile 6 / 4 / 2 / 0
            y2_lgkm_wait4<6>(fa[bf][0][0], fa[bf][0][1], fb[bf][0][0], fb[bf][0][1]);
            __builtin_amdgcn_sched_barrier(0);
            Y2W3_MMA(bf, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (NEXT) { fa[nb][0][0] = y2_tr16_read_off<XO>(xs[0]); fa[nb][0][1] = y2_tr16_read_off<XO + 4 * XROWB>(xs[0]); }
            y2_lgkm_wait2<NEXT ? 6 : 4>(fb[bf][1][0], fb[bf][1][1]);
            __builtin_amdgcn_sched_barrier(0);
            Y2W3_MMA(bf, 0, 1);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (NEXT) { fb[nb][0][0] = y2_tr16_read_off<YO>(ys[0]); fb[nb][0][1] = y2_tr16_read_off<YO + 4 * YROWB>(ys[0]); }
            y2_lgkm_wait2<NEXT ? 6 : 2>(fa[bf][1][0], fa[bf][1][1]);
            __builtin_amdgcn_sched_barrier(0);
            Y2W3_MMA(bf, 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (NEXT) { fb[nb][1][0] = y2_tr16_read_off<YO>(ys[1]); fb[nb][1][1] = y2_tr16_read_off<YO + 4 * YROWB>(ys[1]); }
            __builtin_amdgcn_sched_barrier(0);
            Y2W3_MMA(bf, 1, 1);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (NEXT) { fa[nb][1][0] = y2_tr16_read_off<XO>(xs[1]); fa[nb][1][1] = y2_tr16_read_off<XO + 4 * XROWB>(xs[1]); }
            y2_lgkm_wait2<NEXT ? 8 : 0>(fa[bf][2][0], fa[bf][2][1]);
            __builtin_amdgcn_sched_barrier(0);
            Y2W3_MMA(bf, 2, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (NEXT) { fa[nb][2][0] = y2_tr16_read_off<XO>(xs[2]); fa[nb][2][1] = y2_tr16_read_off<XO + 4 * XROWB>(xs[2]); }
            __builtin_amdgcn_sched_barrier(0);
            Y2W3_MMA(bf, 2, 1);
            __builtin_amdgcn_sched_barrier(0);
        });
#undef Y2W3_MMA
#endif
    };
    // waits until stage s has landed: the stages behind it may stay in flight
    auto wait_stage = [&](int ahead) {
        if (NSTAGE >= 4 && ahead >= 3) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(3 * PIECES_MIN) : "memory");
        else if (NSTAGE >= 3 && ahead >= 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * PIECES_MIN) : "memory");
        else if (ahead >= 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PIECES_MIN) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    };
    // One super-step.  SP = s & 1 (the table parities are compile-time offsets).
    //   one group:  stage s landed, table of stage s + NSTAGE - 1 written -> barrier -> that stage's DMA pieces -> MFMA phase -> next table.
    //   two groups: stage s landed -> barrier -> MFMA phase -> barrier (the group has left the slot) -> DMA pieces of stage s + NSTAGE into it ->
    //               table of the stage after.  Group 1 runs one barrier behind group 0: between two barriers one group is in its MFMA phase, the
    //               other in its LOAD phase.
    auto super_step = [&](int s, auto sp_) {
        constexpr int SP = decltype(sp_)::value;
        if constexpr (ABL & 512) { tm_a = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
        if constexpr (G == 1) {
            wait_stage(min(NSTAGE - 2, t_issue - 1 - s));
            if (!(ABL & 16)) __builtin_amdgcn_s_barrier();
            if constexpr (ABL & 512) { tm_b = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
            if (!(ABL & 4) && t_issue < nsuper) issue_stage(std::integral_constant<int, (SP + NSTAGE - 1) & 1>{});
            if constexpr (ABL & 512) { tm_c = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
            begin_tile();
            mfma_phase();
            if constexpr (ABL & 512) { __builtin_amdgcn_sched_barrier(0); tm_d = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
            if (!(ABL & 4) && s + NSTAGE < nsuper) build_table(s + NSTAGE);
        } else {
            if (!(ABL & 16)) __builtin_amdgcn_s_barrier();
            if constexpr (ABL & 512) { tm_b = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
            mfma_phase();                                                    // (its first fragment set is already in registers)
            if (s + 1 < nsuper) wait_stage(min(NSTAGE - 2, t_issue - 2 - s));      // this wave's pieces of stage s + 1 have landed ...
            if constexpr (ABL & 512) { __builtin_amdgcn_sched_barrier(0); tm_c = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
            if (!(ABL & 16)) __builtin_amdgcn_s_barrier();                   // ... and so have every wave's; the group has left slot s
            if (!(ABL & 4) && t_issue < nsuper) issue_stage(std::integral_constant<int, (SP + NSTAGE) & 1>{});
            if constexpr (ABL & 512) { __builtin_amdgcn_sched_barrier(0); tm_d = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
            if (!(ABL & 4) && s + NSTAGE + 1 < nsuper) build_table(s + NSTAGE + 1);
            if (s + 1 < nsuper) begin_tile();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // table written (read behind two barriers), fragments in registers
        }
        if constexpr (ABL & 512) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long tm_e = __builtin_amdgcn_s_memtime();
            // one group: wait, DMA issue, MFMA phase, table.  two groups: top barrier, MFMA phase (the "issue" slot), mid barrier + DMA issue (the "mfma" slot), table + fragment preload
            tm_wait += tm_b - tm_a; tm_issue += tm_c - tm_b; tm_mfma += tm_d - tm_c; tm_tbl += tm_e - tm_d;
        }
    };
    if constexpr (G == 2) {
        wait_stage(min(NSTAGE - 1, t_issue - 1));          // stage 0 has landed (every wave's pieces: barrier), first fragments
        if (!(ABL & 16)) __builtin_amdgcn_s_barrier();
        begin_tile();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (grp == 1 && !(ABL & 16)) __builtin_amdgcn_s_barrier();           // group 1 runs half a super-step behind
    }
    for (int s = 0; s < nsuper; s += 2) {
        super_step(s, std::integral_constant<int, 0>{});
        if (s + 1 < nsuper) super_step(s + 1, std::integral_constant<int, 1>{});
    }
    if (G == 2 && grp == 0 && !(ABL & 16)) __builtin_amdgcn_s_barrier();     // (every wave executes the same number of barriers)

    unsigned long long tm_epi = 0;
    if constexpr (ABL & 512) tm_epi = __builtin_amdgcn_s_memtime();
    // ---- the KW wave groups hold partial sums over disjoint pixel tiles: tree reduction through LDS (the ring is free: every DMA has landed
    // and been read), group 0 ends with the workgroup's sums
    if constexpr (KW > 1 && !(ABL & 64)) {
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
    if constexpr ((ABL & 512) != 0) {
        // (stamped build: the real stores, then the sums behind them -- the caller allocates blocks * waves * 8 extra floats pairs at the end of dW)
        if (c0 + BC <= Cin && n0 + BN <= Cout) write_tile(std::false_type{});
        else write_tile(std::true_type{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long tm_end = __builtin_amdgcn_s_memtime();
        if (lane == 0) {
            unsigned long long *dbg = reinterpret_cast<unsigned long long *>(dW + (long)9 * Cin * Cout) + ((long)blockIdx.x * NWAVE + wave) * 8;
            dbg[0] = tm_end - tm_k0; dbg[1] = tm_wait; dbg[2] = tm_issue; dbg[3] = tm_mfma; dbg[4] = tm_tbl; dbg[5] = tm_pro - tm_k0; dbg[6] = tm_end - tm_epi; dbg[7] = (unsigned long long)nsuper;
        }
        return;
    }
    if constexpr (ABL & 32) {
        if (acc[0][0][0] == 123.456f && acc[2][1][5] == 1.0f) dW[0] = acc[1][1][3];       // (keeps the accumulators alive, stores nothing)
        return;
    }
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

// ---- plan: which instantiation, how many pixel ranges.  Measured per layer against the per-tap kernel (profiles/r05_w3_ab_pingpong.txt):
//   variant 2: 64 x 128 tile, two wave groups of four waves, 64-pixel tiles, 3 stages  -- single-range grids (13x13 stages), plain stores
//   variant 4: 128 x 128 tile, eight waves in lockstep (one group)                      -- the same, when its grid quantises no worse (1024 -> 1024 at 13x13)
//   variant 5: 64 x 64 tile, four wave groups in two phases, 64-pixel tiles, 2 stages   -- split reductions (26x26 .. 104x104): workgroups x 12 K atomics
// (variants 0, 1, 3 -- 32 x 64 / 64 x 64 with 32-pixel tiles, four-wave workgroups -- lost everywhere and are only built with -DY2W3_EXPERIMENTS;
// layers with <= 32 input channels stay on the per-tap kernel's tap-pair form: 56-60 us against 110+.)
struct Y2W3Geom { int BC, BN, KW, P, waves; };
static const Y2W3Geom g_w3_geom[6] = {{32, 64, 8, 32, 8}, {64, 64, 4, 32, 8}, {64, 128, 2, 64, 8}, {64, 128, 1, 64, 4}, {128, 128, 1, 64, 8}, {64, 64, 4, 64, 8}};
static bool w3_built(int v) {
#ifdef Y2W3_EXPERIMENTS
    return v >= 0 && v <= 5;
#else
    return v == 2 || v == 4 || v == 5;
#endif
}

Y2W3Plan y2_wgrad3_plan(int B, int H, int W, int Cin, int Cout, int cus, int force_variant) {
    Y2W3Plan p = {};
    p.variant = -1;
    const long Mp = (long)B * H * (W + 1);
    if (H < 2 || W < 2 || Cin <= 8 || Mp + 8192 >= (1L << 24)) return p;       // (24-bit multiplies in the DMA address arithmetic; the image layer has its own kernel)
    if (cus <= 0) cus = 256;
    auto cols = [&](int v) { return 3L * cdiv(Cin, g_w3_geom[v].BC) * cdiv(Cout, g_w3_geom[v].BN); };
    auto fill = [&](int v) { const long c = cols(v); return (double)c / (double)(cdiv(c, cus) * (long)cus); };      // share of the CU slots the last round leaves busy
    int v;
    if (force_variant >= 0) {
        if (!w3_built(force_variant)) return p;
        v = force_variant;
    } else if (Cin <= 32) return p;
    else if (cols(2) * 10 >= (long)cus * 6) v = (cols(4) * 10 >= (long)cus * 6 && fill(4) >= fill(2)) ? 4 : 2;      // the tile grid alone gives >= 60 % of the CUs a workgroup
    else v = 5;
    const Y2W3Geom &gm = g_w3_geom[v];
    const long ncols = cols(v);
    const int step = gm.KW * gm.P;
    long ks = ncols >= cus ? 1 : cus / ncols;
    const long max_ks = Mp / (4L * step) > 1 ? Mp / (4L * step) : 1;          // >= four ring turns per workgroup
    if (ks > max_ks) ks = max_ks;
    static const int env_ks = y2_env_int("YOLO2_W3_KS", 0), env_remap = y2_env_int("YOLO2_W3_REMAP", -1);      // (sweeps)
    if (env_ks > 0) ks = env_ks;
    // >= 8 ranges placed range-per-XCD (remap 1) need a multiple of 8 to load the XCDs evenly (10 ranges gave two XCDs 48 workgroups and the others
    // 24: the 52x52 layers ran 57 us instead of 39); YOLO2_W3_REMAP=2 takes contiguous runs per XCD instead and any range count
    else if (ks >= 8 && env_remap != 2) ks = ks / 8 * 8;
    if ((v == 2 || v == 4) && force_variant < 0) ks = 1;
    p.qchunk = cdiv(cdiv(Mp, ks), step) * step;
    p.ks = cdiv(Mp, p.qchunk);
    p.variant = v;
    p.direct = p.ks == 1;
    p.remap = p.ks == 1 ? 2 : (p.ks >= 8 ? 1 : 0);
    if (env_remap >= 0 && p.ks > 1) p.remap = env_remap;
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
#ifdef Y2W3_EXPERIMENTS
    static const int abl = y2_env_int("YOLO2_W3_ABL", 0);
#define Y2W3_ABL(WCv, WNv, KWv, Pv, NSv, Av)                                                                                                        \
    conv_wgrad_row_kernel<WCv, WNv, KWv, Pv, NSv, Av><<<dim3(p.blocks), WCv * WNv * KWv * 64, 0, st>>>((const bf16 *)X, x_bytes, (const bf16 *)dY, y_bytes, dW, H, W, \
                                                                                                       Cin, ldx, Cout, ldy, Mp, CT, NT, p.qchunk, p.ks, p.remap, p.direct, mW, sW, mH, sH)
#define Y2W3_ABL_CASES(WCv, WNv, KWv, Pv, NSv)                                                                          \
    switch (abl) {                                                                                                      \
        case 1: Y2W3_ABL(WCv, WNv, KWv, Pv, NSv, 1); return 0;                                                          \
        case 2: Y2W3_ABL(WCv, WNv, KWv, Pv, NSv, 2); return 0;                                                          \
        case 3: Y2W3_ABL(WCv, WNv, KWv, Pv, NSv, 3); return 0;                                                          \
        case 4: Y2W3_ABL(WCv, WNv, KWv, Pv, NSv, 4); return 0;                                                          \
        case 7: Y2W3_ABL(WCv, WNv, KWv, Pv, NSv, 7); return 0;                                                          \
        case 8: Y2W3_ABL(WCv, WNv, KWv, Pv, NSv, 8); return 0;                                                          \
        case 16: Y2W3_ABL(WCv, WNv, KWv, Pv, NSv, 16); return 0;                                                        \
        case 6: Y2W3_ABL(WCv, WNv, KWv, Pv, NSv, 6); return 0;                                                          \
        case 5: Y2W3_ABL(WCv, WNv, KWv, Pv, NSv, 5); return 0;                                                          \
        case 39: Y2W3_ABL(WCv, WNv, KWv, Pv, NSv, 39); return 0;                                                        \
        case 231: Y2W3_ABL(WCv, WNv, KWv, Pv, NSv, 231); return 0;                                                      \
        case 256: Y2W3_ABL(WCv, WNv, KWv, Pv, NSv, 256); return 0;                                                      \
        case 512: Y2W3_ABL(WCv, WNv, KWv, Pv, NSv, 512); return 0;                                                      \
        case 103: Y2W3_ABL(WCv, WNv, KWv, Pv, NSv, 103); return 0;                                                      \
        case 32: Y2W3_ABL(WCv, WNv, KWv, Pv, NSv, 32); return 0;                                                        \
        default: break;                                                                                                 \
    }
    if (abl && p.variant == 2) { Y2W3_ABL_CASES(2, 2, 2, 64, 3) }
    if (abl && p.variant == 5) { Y2W3_ABL_CASES(2, 1, 4, 64, 2) }
    if (abl == 512 && p.variant == 4) { Y2W3_ABL(4, 2, 1, 64, 3, 512); return 0; }
#endif
    switch (p.variant) {
#ifdef Y2W3_EXPERIMENTS
        case 0: Y2W3_LAUNCH(1, 1, 8, 32, 2); break;
        case 1: Y2W3_LAUNCH(2, 1, 4, 32, 4); break;
        case 3: Y2W3_LAUNCH(2, 2, 1, 64, 3); break;
#endif
        case 2: Y2W3_LAUNCH(2, 2, 2, 64, 3); break;
        case 4: Y2W3_LAUNCH(4, 2, 1, 64, 3); break;
        case 5: Y2W3_LAUNCH(2, 1, 4, 64, 2); break;
        default: return 1;
    }
#undef Y2W3_LAUNCH
    return 0;
}
