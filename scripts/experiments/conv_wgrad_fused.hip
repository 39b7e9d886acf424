// EXPERIMENT RECORD -- not built, not part of libyolo2hip.so (removed from the product in round 2).
// Tap-fused 3x3 filter gradient: one workgroup owns a 64-channel x 64-filter tile for all nine taps; bands of whole image rows
// are staged once as zero-padded patches, every tap is a constant LDS offset.  It is CORRECT (it passed
// tests/test_kernels_gpu.py::test_conv_wgrad_tap_fused on 9 shapes incl. ragged bands and strided operands before removal)
// and it cuts the operand DMA 9x, but it is SLOWER than conv_wgrad.hip on every Darknet-19 layer
// (profiles/r02_wgrad_fused_and_bkp64.txt):
//   v1 single-buffered, 2 workgroups/CU:  conv20 170 us (old kernel 174), conv18 76 (69), conv13 79 (47), conv8 82 (56)
//   v2 double-buffered, hand-pipelined asm reads, 1 workgroup/CU (this file): conv20 244, conv18 86, conv13 68
// Why: its fragments come from ds_read_b64_tr_b16 (2.2 reads per MFMA at 9 x 16 accumulator registers per wave), and 8-byte
// LDS reads only reach their rate with ~4+ waves per SIMD (MI355X_MICROARCH.md, LDS): 144 accumulator registers cap the kernel
// at 1-2 waves per SIMD, where conv_wgrad.hip runs 6.  Band ranges below ~192 tiles also pay f32 atomics on 9x larger
// tiles.  What DID transfer to the product: issuing the transpose reads through inline asm (the builtin made hipcc drain
// the DMA ring with vmcnt(0) before every tile: conv_wgrad.hip gained 10-25 % from that alone).
// Tap-fused filter gradient of the 3x3 NHWC convolution on MFMA (gfx950, bf16).
//
// Replaces tf.gradients of slim.layers.conv2d w.r.t. its weights (reference train.py:127-129, call sites
// model/yolo2/inference.py:73-117) for the 3x3 layers with >= 64 input and output channels:
//   dW[tap][c][n] = sum_m X[pix(m) + shift(tap), c] * dY[m, n]
//
// conv_wgrad.hip gives every (tap, c-tile, n-tile) its own workgroup: each tap re-stages the same dY tile and a shifted
// copy of the same X rows (9x the operand DMA), with one barrier per 4 MFMAs.  Here ONE workgroup owns a
// (64-channel, 64-filter) tile for ALL NINE taps:
//   * the reduction runs over BANDS of whole image rows (R rows x W pixels <= 208 pixels = 13 MFMA K steps; a 13x13 image
//     is one band).  Per band the dY rows are staged once, and X once as a zero-padded (R+2) x (W+2) patch: the halo ring
//     is realised by the buffer descriptor's range check (out-of-image lanes use an out-of-range offset -> the DMA writes
//     zeros), so every tap is a constant LDS offset ((dh)*(W+2) + dw positions) from the same patch -- no masks, no per-tap
//     border bookkeeping, no re-staging;
//   * each wave holds a 32 x 32 accumulator tile per tap (9 x 16 = 144 registers); per K step it reads the dY fragment once
//     (2 transpose reads) and one X fragment per tap (2 transpose reads + 1 address add each) for 9 MFMAs, and a band is
//     99..117 MFMAs per wave between two barriers;
//   * both operands are reduction-strided in NHWC, so tiles are staged pixel-major and gathered with ds_read_b64_tr_b16 as
//     in conv_wgrad.hip.  LDS holds each operand as two 32-channel planes with 64-byte rows: the four pixel rows and two
//     channel halves a 32-lane transpose read touches then fall on 8 disjoint bank octets (row stride 16 banks), so no XOR
//     swizzle is needed and tap / K-step offsets stay additive;
//   * the pixel -> padded-position map p(q) = q + 2*floor(q/W) + W + 3 is tabulated once per lane (2 entries per K step);
//   * single-buffered: two workgroups per CU (<= 80 KiB each) overlap one's DMA with the other's MFMAs.
// Grid: x = (c-tile, n-tile, band range).  A tile grid that fills the chip takes one range and stores; otherwise ranges
// accumulate with f32 atomics into the zeroed dW (yolo2_conv2d_wgrad_accumulates reports which).
#include "common.h"
#include <atomic>
#include <mutex>
#include <stdlib.h>

#define Y2_OOB 0x80000000u
#define Y2_WF_NKS 13                      // K steps (16 pixels) per band: a band holds <= 208 pixels
#define Y2_WF_YPLANE (Y2_WF_NKS * 16 * 64)

__device__ __forceinline__ bf16x8 wf_frag(const unsigned char *p0, const unsigned char *p1) {
    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p0);
    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p1);
    const bf16x4 a = __builtin_bit_cast(bf16x4, v0), b = __builtin_bit_cast(bf16x4, v1);
    bf16x8 f;
    f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3];
    f[4] = b[0]; f[5] = b[1]; f[6] = b[2]; f[7] = b[3];
    return f;
}

#define Y2_WF_XSLOTS 14                   // X DMA instructions per wave and band (<= 54 per band over 4 waves)
#define Y2_WF_YSLOTS 8                    // dY DMA instructions per wave and band (26 per band over 4 waves, whole plane pairs)

__global__ __launch_bounds__(256, 1) void conv_wgrad_fused_kernel(
    const bf16 *__restrict__ X, unsigned x_bytes, const bf16 *__restrict__ dY, unsigned y_bytes, float *__restrict__ dW, int B, int H, int W,
    int Cin, int ldx, int Cout, int ldy, int R, int NT, int tiles, int bands_per_split, int direct) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;                   // waves 2 (channels) x 2 (filters), 32 x 32 each
    const int W2 = W + 2;
    const int xpos = (R + 2) * W2;                             // padded positions of a band
    const int xrows = (xpos + 15) & ~15;                       // whole DMA instructions: 16 rows of 64 B
    const int XPLANE = xrows * 64;
    const int STAGE = 2 * XPLANE + 2 * Y2_WF_YPLANE;           // [2 X planes][xrows][32 ch] + [2 dY planes][208 px][32 filters]; two stages

    const int tile = blockIdx.x % tiles, split = blockIdx.x / tiles;
    const int nt = tile % NT, ct = tile / NT;
    const int c0 = ct * 64, n0 = nt * 64;
    const int nb = (H + R - 1) / R;                            // bands per image
    const int bands = B * nb;
    const int band_beg = split * bands_per_split, band_end = min(bands, band_beg + bands_per_split);
    if (band_beg >= band_end) return;

    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16 *>(X), 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcY = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16 *>(dY), 0, y_bytes, 0x00020000);

    // transpose-read geometry (layout pinned by tests/test_kernels_gpu.py::test_tr16_layout): 16-lane group g, lane t of it supplies
    // the address of pixel row (t >> 2), channel quad (t & 3) and receives 4 pixels of channel 16*(g&1) + t
    const int g = lane >> 4, t = lane & 15;
    const int q0 = 8 * (g >> 1) + (t >> 2);                    // this lane's first pixel inside a K step (+ 4 for the second read)
    const int lane_col = 32 * (g & 1) + 8 * (t & 3);           // byte offset inside a 64-byte row
    const int band_px = R * W;
    const float rcp_w = 1.0f / (float)W, rcp_w2 = 1.0f / (float)W2;
    // padded position of band pixel q: (q / W + 1) * (W + 2) + q % W + 1, as an LDS byte offset of this lane's X plane
    unsigned xt[Y2_WF_NKS][2];
#pragma unroll
    for (int ks = 0; ks < Y2_WF_NKS; ++ks)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            int q = q0 + 16 * ks + 4 * r;
            q = q < band_px ? q : band_px - 1;                // rows past the band: their dY rows are zero, any valid position will do
            int h = (int)(((float)q + 0.5f) * rcp_w);
            int w = q - h * W;
            if (w < 0) { w += W; --h; } else if (w >= W) { w -= W; ++h; }
            xt[ks][r] = (unsigned)(wm * XPLANE + ((h + 1) * W2 + (w + 1)) * 64 + lane_col);
        }
    const unsigned ybase = (unsigned)(2 * XPLANE + wn * Y2_WF_YPLANE + q0 * 64 + lane_col);

    // DMA slot tables of this wave (the band geometry is the same for every band): instruction i of the band = plane i & 1, row
    // block i >> 1; wave w takes the PAIRS (2k, 2k+1) with k % 4 == w, so the two 64-byte halves of a pixel's 128 bytes are fetched
    // back to back by one wave.  x_rel: byte offset from the band's first pixel (row r0, column 0, channel c0), x_hh: padded row
    // (validity depends on the band's position in the image), OOB / -1 for positions outside the patch or in the zero columns.
    const int n_xi = 2 * (xrows >> 4);
    unsigned x_rel[Y2_WF_XSLOTS], x_lds[Y2_WF_XSLOTS];
    int x_hh[Y2_WF_XSLOTS];
#pragma unroll
    for (int sl = 0; sl < Y2_WF_XSLOTS; ++sl) {
        const int i = ((sl >> 1) * 4 + wave) * 2 + (sl & 1);
        const int plane = i & 1, blk = i >> 1;
        const int pos = blk * 16 + (lane >> 2);
        int hh = (int)(((float)pos + 0.5f) * rcp_w2);
        int ww = pos - hh * W2;
        if (ww < 0) { ww += W2; --hh; } else if (ww >= W2) { ww -= W2; ++hh; }
        const bool ok = i < n_xi && pos < xpos && ww >= 1 && ww <= W;
        x_rel[sl] = (unsigned)((((long)(hh - 1) * W + (ww - 1)) * ldx + plane * 32 + (lane & 3) * 8) * 2);      // may be "negative": added mod 2^32
        x_hh[sl] = ok ? hh : -1;
        x_lds[sl] = (unsigned)(plane * XPLANE + blk * 1024);
    }
    unsigned y_rel[Y2_WF_YSLOTS], y_lds[Y2_WF_YSLOTS];
    int y_q[Y2_WF_YSLOTS];
#pragma unroll
    for (int sl = 0; sl < Y2_WF_YSLOTS; ++sl) {
        const int i = ((sl >> 1) * 4 + wave) * 2 + (sl & 1);
        const int plane = i & 1, blk = i >> 1;
        y_q[sl] = blk * 16 + (lane >> 2);
        y_rel[sl] = (unsigned)(((long)y_q[sl] * ldy + plane * 32 + (lane & 3) * 8) * 2);
        y_lds[sl] = (unsigned)(2 * XPLANE + plane * Y2_WF_YPLANE + blk * 1024);
    }
    // slot sl of this wave exists iff its instruction index is below the band's count: wave-uniform
    auto x_slot_on = [&](int sl) { return (((sl >> 1) * 4 + wave) * 2 + (sl & 1)) < n_xi; };
    auto y_slot_on = [&](int sl, int nks) { return (((sl >> 1) * 4 + wave) * 2 + (sl & 1)) < 2 * nks; };

    struct Band { unsigned xbase, ybase; int lo, hi, px, nks; };
    auto band_of = [&](int band) {
        Band d;
        const int b = band / nb, r0 = (band - b * nb) * R;
        const int rows = min(R, H - r0);
        d.px = rows * W;
        d.nks = (d.px + 15) >> 4;
        d.lo = r0 > 0 ? 0 : 1;                                  // padded rows lo..hi lie inside the image
        d.hi = min(R + 1, H - r0);
        const long m0 = ((long)b * H + r0) * W;
        d.xbase = (unsigned)((m0 * ldx + c0) * 2);
        d.ybase = (unsigned)((m0 * ldy + n0) * 2);
        return d;
    };
    auto issue_x = [&](const Band &d, int sl, unsigned stage_off) {
        const bool ok = x_hh[sl] >= d.lo && x_hh[sl] <= d.hi;
        const unsigned voff = ok ? d.xbase + x_rel[sl] : Y2_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (__attribute__((address_space(3))) void *)(smem + stage_off + x_lds[sl]), 16, voff, 0, 0, 0);
    };
    auto issue_y = [&](const Band &d, int sl, unsigned stage_off) {
        const unsigned voff = y_q[sl] < d.px ? d.ybase + y_rel[sl] : Y2_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcY, (__attribute__((address_space(3))) void *)(smem + stage_off + y_lds[sl]), 16, voff, 0, 0, 0);
    };
    // slot order of a band's prefetch: the Y slots first (fewer, needed by every K step), then X
    auto issue_slot = [&](const Band &d, int s, unsigned stage_off) {      // s: 0 .. YSLOTS + XSLOTS - 1, compile-time at every call site
        if (s < Y2_WF_YSLOTS) { if (y_slot_on(s, d.nks)) issue_y(d, s, stage_off); }
        else if (x_slot_on(s - Y2_WF_YSLOTS)) issue_x(d, s - Y2_WF_YSLOTS, stage_off);
    };
    constexpr int NSLOT = Y2_WF_YSLOTS + Y2_WF_XSLOTS;

    f32x16 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

    // prologue: the first band into stage 0
    Band cur = band_of(band_beg);
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) issue_slot(cur, s, 0u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned stage = 0;
    for (int band = band_beg; band < band_end; ++band) {
        const bool more = band + 1 < band_end;
        const Band nxt = band_of(more ? band + 1 : band);
        const unsigned cs = stage * STAGE, ns = (stage ^ 1u) * STAGE;
        // K steps of this band.  The next band's DMA is issued two slots per K step so that its address arithmetic hides under the
        // MFMAs.  Fragment reads are software-pipelined by hand two taps ahead (and across K steps) with counted lgkmcnt waits:
        // stream order ... A(ks,tap) A(ks,tap+1) A(ks,tap+2) ..., with B(ks+1) slotted in before A(ks+1,0); consuming A(ks,tap)
        // may leave exactly the reads issued after it outstanding (LDS returns in order).
#if defined(__HIP_DEVICE_COMPILE__)
        const unsigned lds0 = y2_lds_addr(smem) + cs;
        constexpr int D = 4;                                  // prefetch distance in taps (2 reads each): <= 2*D + 2 = 10 LDS reads in flight
        u32x2 Af[D + 1][2], Bf[2][2];
        auto issue_a = [&](int ks, int tap, int buf) {
            const unsigned toff = (unsigned)(((tap / 3 - 1) * W2 + (tap % 3 - 1)) * 64);
            Af[buf][0] = y2_tr16_read(lds0 + xt[ks][0] + toff);
            Af[buf][1] = y2_tr16_read(lds0 + xt[ks][1] + toff);
        };
        auto issue_b = [&](int ks, int buf) {
            const unsigned yp = lds0 + ybase + (unsigned)ks * 1024u;
            Bf[buf][0] = y2_tr16_read(yp);
            Bf[buf][1] = y2_tr16_read_off<256>(yp);
        };
        auto wait_n = [&](int n, u32x2 &a, u32x2 &b) {        // n is a constant after unrolling: one branch survives
            switch (n) {
                case 0: y2_lgkm_wait2<0>(a, b); break;
                case 2: y2_lgkm_wait2<2>(a, b); break;
                case 4: y2_lgkm_wait2<4>(a, b); break;
                case 6: y2_lgkm_wait2<6>(a, b); break;
                case 8: y2_lgkm_wait2<8>(a, b); break;
                default: y2_lgkm_wait2<10>(a, b); break;
            }
        };
        issue_b(0, 0);
#pragma unroll
        for (int tp = 0; tp < D; ++tp) issue_a(0, tp, tp);
#pragma unroll
        for (int ks = 0; ks < Y2_WF_NKS; ++ks) {
            if (more) {
                if (2 * ks < NSLOT) issue_slot(nxt, 2 * ks, ns);
                if (2 * ks + 1 < NSLOT) issue_slot(nxt, 2 * ks + 1, ns);
            }
            if (ks < cur.nks) {
                const bool last = ks + 1 >= cur.nks;            // wave-uniform
                const int ks1 = ks + 1 < Y2_WF_NKS ? ks + 1 : ks;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int n = ks * 9 + tap;                   // item number in the stream: buffer n % (D + 1)
                    u32x2 &a0 = Af[n % (D + 1)][0], &a1 = Af[n % (D + 1)][1];
                    if (tap + D <= 8) {
                        issue_a(ks, tap + D, (n + D) % (D + 1));
                        wait_n(2 * D, a0, a1);
                    } else if (!last) {
                        if (tap + D == 9) issue_b(ks1, (ks + 1) & 1);
                        issue_a(ks1, tap + D - 9, (n + D) % (D + 1));
                        wait_n(2 * D + 2, a0, a1);              // the window holds B(ks+1) too
                    } else {
                        wait_n(2 * (8 - tap), a0, a1);
                    }
                    if (tap == 0) wait_n(2 * D + 2, Bf[ks & 1][0], Bf[ks & 1][1]);      // B(ks) precedes A(ks,0) in the stream: a register tie, no extra wait
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y2_frag16(a0, a1), y2_frag16(Bf[ks & 1][0], Bf[ks & 1][1]), acc[tap], 0, 0, 0);
                }
            }
        }
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the next band has landed ...
        __syncthreads();                                       // ... for every wave, and every wave is done reading this one
        cur = nxt;
        stage ^= 1u;
    }

    // epilogue: dW is HWIO [tap][Cin][Cout]; C/D layout of the 32x32 MFMA: col (filter) = lane & 31, row (channel) = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int n = n0 + wn * 32 + (lane & 31);
    const int cb = c0 + wm * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        float *col = dW + ((long)tap * Cin + cb) * Cout + n;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dc = (r & 3) + 8 * (r >> 2);
            if (direct) col[(long)dc * Cout] = acc[tap][r];
            else unsafeAtomicAdd(col + (long)dc * Cout, acc[tap][r]);
        }
    }
}

// ---- host side --------------------------------------------------------------------------------------------------------
// 0 = off, 1 = measured per-shape choice (default), 2 = every eligible shape (A/B runs, tests); process-wide
static std::atomic<int> g_wf_mode{-1};
static int wf_mode() {
    int m = g_wf_mode.load();
    if (m < 0) {
        m = getenv("YOLO2_WGRAD_FUSED") ? atoi(getenv("YOLO2_WGRAD_FUSED")) : 1;
        g_wf_mode.store(m);
    }
    return m;
}
extern "C" void yolo2_debug_set_wgrad_fused(int mode) { g_wf_mode.store(mode); }

// eligible shapes: bf16, 3x3, whole 64-channel / 64-filter tiles, image rows that fit a band
bool y2_wgrad_fused_plan(int B, int H, int W, int Cin, int ldx, int Cout, int ldy, int ksize, int dtype, Y2WgradFusedPlan *out) {
    if (wf_mode() == 0 || dtype != YOLO2_BF16 || ksize != 3 || Cin % 64 || Cout % 64 || W > 16 * Y2_WF_NKS || Cin > ldx || Cout > ldy) return false;
    Y2WgradFusedPlan p;
    p.R = 16 * Y2_WF_NKS / W;
    if (p.R > H) p.R = H;
    const int xrows = (((p.R + 2) * (W + 2)) + 15) & ~15;
    p.lds_bytes = 2 * (2 * xrows * 64 + 2 * Y2_WF_YPLANE);      // two stages
    if (p.lds_bytes > 160 * 1024 || 2 * (xrows / 16) > 4 * Y2_WF_XSLOTS) return false;
    p.tiles = (Cin / 64) * (Cout / 64);
    const int bands = B * cdiv(H, p.R);
    if (wf_mode() == 1) {
        // measured on the Darknet-19 shapes (profiles/r02_wgrad_fused.txt): pays where the tile grid is large enough that few
        // band ranges are needed; the early layers (1-8 tiles, hundreds of ranges) drown in atomics
        if (p.tiles < 16) return false;
    }
    // band ranges: ~256 workgroups (one per CU: the two LDS stages take most of it); a tile grid that already fills most of the chip
    // takes one range and stores
    int splits = p.tiles >= 192 ? 1 : cdiv(256, p.tiles);
    if (splits > bands) splits = bands;
    p.bands_per_split = cdiv(bands, splits);
    p.splits = cdiv(bands, p.bands_per_split);
    *out = p;
    return true;
}

int y2_wgrad_fused_launch(const void *X, const void *dY, float *dW, int B, int H, int W, int Cin, int ldx, int Cout, int ldy,
                          const Y2WgradFusedPlan &p, hipStream_t st, int *plan8) {
    static std::once_flag once;
    static hipError_t attr_rc = hipSuccess;
    std::call_once(once, [] { attr_rc = hipFuncSetAttribute((const void *)conv_wgrad_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    if (attr_rc != hipSuccess) return 1;
    const size_t M = (size_t)B * H * W;
    const unsigned x_bytes = (unsigned)(M * ldx * 2), y_bytes = (unsigned)(M * ldy * 2);
    const int NT = Cout / 64;
    const int direct = p.splits == 1;
    conv_wgrad_fused_kernel<<<p.tiles * p.splits, 256, p.lds_bytes, st>>>((const bf16 *)X, x_bytes, (const bf16 *)dY, y_bytes, dW, B, H, W, Cin, ldx, Cout,
                                                                            ldy, p.R, NT, p.tiles, p.bands_per_split, direct);
    if (plan8) {
        const int v[8] = {64, 64, 4, 9, p.splits, 0, p.tiles * p.splits, direct};      // "pair" slot = taps per workgroup
        for (int i = 0; i < 8; ++i) plan8[i] = v[i];
    }
    return 0;
}
