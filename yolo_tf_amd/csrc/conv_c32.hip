// 3x3 convolution FORWARD for 32-channel inputs and 64 filters (Darknet-19 conv1: 208 x 208 pixels per image), bf16, gfx950.
//
// Replaces slim.layers.conv2d of reference model/yolo2/inference.py:75-76 (conv1: 32 -> 64, 3x3, SAME, no bias under batch_norm) where the
// generic implicit-GEMM kernel is at its worst: the reduction is only 9 taps x 32 channels = 288 long, so a 128 x 64 tile of the per-tap
// kernel runs nine K steps of TWO MFMAs per wave between barriers and pays a prologue and an epilogue per 18 MFMAs -- 74 us for 25.5 GFLOP
// (344 TFLOP/s) although the layer moves only 133 MB.  What this kernel does instead:
//   * persistent workgroups (one per CU) walk 512-position tiles of ONE CONTIGUOUS RANGE of positions each; the whole filter (64 x 288 bf16, 37 KB) is
//     loaded into LDS ONCE per workgroup;
//   * the pixels run over a PADDED index q (row pitch W + 1, image pitch H + 1: one zero column per row, one zero row per image -- the
//     filter-gradient kernel's idea, conv_wgrad3.hip): all nine taps are constant row offsets dh (W + 1) + dw into the staged pixel rows with no
//     (pixel, tap) mask anywhere; the DMA source offset of a staged row decides whether it is a pixel or a zero;
//   * the staged rows form a RING over the workgroup's range (round 6, from conv_c64.hip): a tile's 2 (W + 2) halo rows are the previous tile's, only
//     its 512 new rows stream in (by LDS-DMA, under the MFMAs) -- 45 MB per launch instead of the 82 MB two whole halo images per tile re-staged;
//   * a wave computes 64 positions x 64 filters (72 MFMAs per tile on 72 + 72 ds_read_b128), one raw barrier per tile;
//   * the MFMA operand roles are swapped (D = F^T X^T: column = lane & 31 = POSITION, rows = filters) and the filter rows are permuted
//     (c32_slot_filter) so that a lane's 32 accumulator rows are 32 consecutive filters of its position; a 4 x 4 transpose over the lanes of a
//     quad (DPP) then makes every store instruction write whole 128-byte rows: straight from registers, no LDS staging of the output tile.
//     The batch-norm partial sums (same contract as conv_igemm.hip's epilogue) run per lane over its position column and meet once, after the
//     last tile: across lanes on the VALU (DPP), across the eight waves in LDS, 128 atomics per workgroup into 32 partial rows.
// W / (W + 1) x H / (H + 1) of the MFMA work is real (99 % at 208 x 208).
//
// Measured (batch 16, 208 x 208; microbenchmark, one call): round 5 (two halo images, ds_bpermute butterfly, 1024 atomics per workgroup into 228 rows)
// plain store 35.0 us, + statistics 55.2 us; round 6 31.6-32.7 / 39.6 us; the generic kernel 70-74 us.  s_memtime stamps (scripts/c32_phase_cycles.py,
// STATS=1) showed where the 20 us between "plain" and "+ statistics" were: 12 us in the final butterfly (64 values x 5 ds_bpermute per wave: all
// through the LDS pipe) and most of the rest in same-address atomics; the per-tile arithmetic of the sums costs ~0.9 us per tile.  The kernel is
// bound by what it moves between L2 and the CUs, not by its 72 MFMAs per wave and tile: 133 MB at the ~4.2 TB/s DMA issue and stores sustain together
// (a wave's DMA pieces take ~0.8k cycles to ISSUE per tile and its eight stores ~1k, against ~12 ticks per piece from cache:
// scripts/experiments/lds_dma_issue.hip).  Tried in round 5 and measured no better: two wave groups half a tile apart (MFMA loop of one over the
// epilogue of the other), DMA pieces interleaved into the MFMA steps, half the workgroups delayed by half a tile.  What did pay: whole-row stores (the
// quad transpose: store issue 1.8k -> 1.0k ticks per tile), packed-f32 branch-free statistics (epilogue 700 -> 380 instructions).
#include "common.h"
#include "conv_shared.h"
#include <atomic>

#define C32_TP 512                 // padded positions per tile (8 waves x 64)
#define C32_FROWB 592              // LDS bytes per filter row: 576 + 16 (row stride 148 banks: the 16 lanes of a read group hit 16 different bank quads)
#define C32_FBYTES (64 * C32_FROWB)
#define C32_STAT_ROWS 32           // partial rows the statistics are spread over (the consumer's prologue sums every row in use)

// MFMA output row rho of filter block j lands in register r = (rho & 3) + 4 (rho >> 3) of the lanes of half h = (rho >> 2) & 1.  Giving that row
// filter 32 h + 16 j + r makes the 32 values of a lane (2 blocks x 16 registers) 32 CONSECUTIVE filters: 64 contiguous bytes of its position's row.
__device__ __forceinline__ int c32_slot_filter(int n) {
    const int j = n >> 5, rho = n & 31;
    return 32 * ((rho >> 2) & 1) + 16 * j + (rho & 3) + 4 * (rho >> 3);
}

__device__ __forceinline__ unsigned c32_div(unsigned q, unsigned m, unsigned s) { return __umulhi(q, m) >> s; }

// MODE 0: plain store; 1: + batch-norm partial sums of the stored values (training forward); 2: + bias, leaky_relu (inference, folded BN)
#ifdef Y2C32_EXPERIMENTS
__device__ unsigned long long c32_stamps[256 * 8];
#define C32_STAMP(k) do { if (abl & 8) { const unsigned long long t_ = __builtin_readcyclecounter(); ph[k] += t_ - tl; tl = t_; } } while (0)
#define C32_ABL(bit) (abl & (bit))
#else
#define C32_STAMP(k) do { } while (0)
#define C32_ABL(bit) 0
#endif

struct C32Geo {
    int H, W, M, Mp;       // image rows / columns, pixels, padded positions of the batch
    int L;                 // padded positions per workgroup (a multiple of the 512-position tile)
    int RR;                // ring rows of 64 bytes (a multiple of 64)
    int HLa;               // ring row of the workgroup's first position: W + 2 rounded up to 16
    int NP0;               // 16-row pieces staged before the first tile
    unsigned mP, sP, mH, sH;
};

template <int MODE>
__global__ __launch_bounds__(512) void conv_c32_fwd_kernel(
    const bf16 *__restrict__ X, unsigned x_bytes, const bf16 *__restrict__ F, unsigned f_bytes, bf16 *__restrict__ O, C32Geo g,
    const float *__restrict__ vec, float *__restrict__ bn_part, float alpha, int abl) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char *const fl = smem;                        // filter image [64][C32_FROWB]
    unsigned char *const hb = smem + C32_FBYTES;           // the ring: RR rows x 64 bytes
    const int H = g.H, W = g.W, Mp = g.Mp;
    const unsigned mP = g.mP, sP = g.sP, mH = g.mH, sH = g.sH;
    const int RRB = g.RR * 64, NPR = g.RR >> 4;             // ring bytes, ring pieces
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const unsigned P1 = (unsigned)(W + 1), H1 = (unsigned)(H + 1);
    const int s0 = (int)blockIdx.x * g.L;                   // this workgroup's positions [s0, e0): ONE contiguous range, so that a tile's halo rows
    const int e0 = min(s0 + g.L, Mp);                       // are the previous tile's and only its 512 new rows are staged (conv_c64.hip's ring)
    if (s0 >= Mp) return;
    const int pbase = s0 - g.HLa;                          // padded position of ring row 0 (may be negative: those rows stage as zeros)

    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16 *>(X), 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcF = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16 *>(F), 0, f_bytes, 0x00020000);
    typedef __attribute__((address_space(3))) void *lds_void_ptr;

    // ---- the filter, once: 37 pieces of 1 KiB; LDS byte p of the image = filter p / 592, byte p % 592 of its row (the 16 padding bytes read as zeros)
    for (int p = wave; p * 1024 < C32_FBYTES; p += 8) {
        const int pos = p * 1024 + lane * 16;
        const int n = pos / C32_FROWB, off = pos - n * C32_FROWB;       // LDS slot n = MFMA row (j = n >> 5, rho = n & 31) holds filter c32_slot_filter(n)
        const unsigned voff = (off < 576 && n < 64) ? (unsigned)(c32_slot_filter(n) * 576 + off) : Y2_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcF, (lds_void_ptr)(fl + p * 1024), 16, voff, 0, 0, 0);
    }

    // ---- ring staging: piece j = ring rows 16 j .. 16 j + 15 (64 bytes each) = padded positions pbase + 16 j ..; lane = (row = lane >> 2, 16-byte chunk =
    // lane & 3), source chunk swizzled with (row >> 2) & 3 so that the 16 rows a ds_read_b128 lane group touches -- whatever row a tap offset starts
    // them at -- land on 16 different bank quads (RR is a multiple of 16: the term survives the wrap).  A staged row is a pixel, or zeros (zero column,
    // zero row, outside the batch).
    const int prow = lane >> 2;
    const unsigned pchunk = (unsigned)(((lane & 3) ^ ((lane >> 4) & 3)) * 16);
    // one piece: q -> (R = q / (W + 1), img = R / (H + 1)); the pixel index is m = q - R - img W (q = R (W + 1) + c, R = img (H + 1) + r, m = (img H + r) W + c)
    auto stage_piece = [&](int j, int slot) {
        const unsigned q = (unsigned)(pbase + j * 16 + prow);                  // wraps below 0: fails the range test
        const unsigned R = c32_div(q, mP, sP);                                 // padded image row over the whole batch
        const unsigned img = c32_div(R, mH, sH);
        const unsigned c = q - __umul24(R, P1), r = R - __umul24(img, H1);
        const bool ok = (q < (unsigned)Mp) & (c < (unsigned)W) & (r < (unsigned)H);
        const unsigned m = q - R - __umul24(img, (unsigned)W);
        const unsigned voff = ok ? m * 64u + pchunk : Y2_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (lds_void_ptr)(hb + slot * 1024), 16, voff, 0, 0, 0);
    };
    for (int j = wave; j < g.NP0; j += 8) stage_piece(j, j);                  // rows [0, HLa + TP + W + 2): the first tile and both its halos
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the filter and this wave's pieces of the first tile have landed

    // ---- read addresses (ring offset 0), relative to the ring.  A (pixels, the MFMA's B operand): one per tap for position block 0; a tile / block moves
    // it by a multiple of 32 rows (same swizzle: (row >> 2) & 3), wrapped at the ring's end; the two 16-channel groups of a tap differ by XOR 32 (bit 1
    // of the chunk index).  B (filters, the MFMA's A operand): row l31 (+ 32 j), 8 k-values of this lane's half.
    const unsigned lds0 = y2_lds_addr(smem);
    unsigned abase[9];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
        const int shift = (tp / 3 - 1) * (W + 1) + (tp % 3 - 1);
        const int row = wave * 64 + l31 + g.HLa + shift;
        abase[tp] = (unsigned)(row * 64 + (((half) ^ ((row >> 2) & 3)) << 4));
    }
    const unsigned ring0 = lds0 + (unsigned)C32_FBYTES;
    const unsigned baddr = lds0 + (unsigned)(l31 * C32_FROWB + half * 16);       // filter block 0; block 1 is 32 rows on

    // per-filter constants of this lane's 2 x 16 accumulator rows: filter f(j, r) = 32 half + 16 j + r (c32_slot_filter).  MODE 2 keeps the bias in
    // registers; MODE 1 keeps the statistics' shift (moving mean) in LDS behind the halo buffers and reads it per epilogue (its 64 sum registers
    // leave no room for 32 more)
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    float cst[2][16];
    f32x2 t1[2][8], t2[2][8];                               // packed pairs: the sums run on v_pk_add_f32 / v_pk_fma_f32
    float *const shl = reinterpret_cast<float *>(hb + RRB);
    if (MODE == 1 && tid < 64) shl[tid] = vec ? vec[tid] : 0.f;
    if (MODE == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the loop's barriers are raw: this write is on its way before the first one)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = 32 * half + 16 * j + r;
            cst[j][r] = (MODE == 2 && vec) ? vec[f] : 0.f;
            t1[j][r >> 1][r & 1] = 0.f;
            t2[j][r >> 1][r & 1] = 0.f;
        }

    typedef const __attribute__((address_space(3))) bf16x8 *lds_frag_ptr;
    const int T = (e0 - s0 + C32_TP - 1) / C32_TP;          // tiles of this workgroup
    int pslot = g.NP0;                                      // ring piece the next tile's first new piece goes to (NP0 + 32 t, wrapped)
    if (pslot >= NPR) pslot -= NPR;
    unsigned tro = 0;                                       // ring byte offset of the current tile (512 t rows, wrapped)
    int tb = s0;                                            // first position of the current tile
    f32x16 acc[2][2];                                                // [filter block j][position block i]
#ifdef Y2C32_EXPERIMENTS
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl = __builtin_readcyclecounter();
    const unsigned long long t00 = tl;
#endif
    auto mfma_phase = [&](int t) {
        if (t + 1 < T && !C32_ABL(4)) {                      // this wave's pieces of the next tile's 512 new rows, over rows every wave left before the barrier
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = wave + 8 * u;
                int sl = pslot + k;
                if (sl >= NPR) sl -= NPR;
                stage_piece(g.NP0 + t * 32 + k, sl);
            }
        }
        pslot += 32;
        if (pslot >= NPR) pslot -= NPR;
        C32_STAMP(1);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
        // 18 steps (tap, 16-channel group) of 4 fragment reads + 4 MFMAs; the reads of step s + 1 are issued ahead of the MFMAs of step s (two
        // fragment sets: left to itself hipcc keeps one and waits lgkmcnt(0) in front of nearly every MFMA)
        bf16x8 fa[2][2], fb[2][2];
        unsigned ad[2];
        auto load = [&](int st, int set) {
            const int tp = st >> 1, kk = st & 1;
            if (kk == 0) {                                  // a tap's addresses are formed when its first group is read
                unsigned x = abase[tp] + tro;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    x = min(x, x - (unsigned)RRB);          // wrap (x < ring: the difference wraps to a huge value)
                    ad[i] = x + ring0;
                    x += 2048u;
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[set][i] = *(lds_frag_ptr)(uintptr_t)(ad[i] ^ (unsigned)(kk * 32));
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[set][j] = *(lds_frag_ptr)(uintptr_t)(baddr + (unsigned)(j * 32 * C32_FROWB + tp * 64 + kk * 32));
        };
        if (!C32_ABL(1)) {
        load(0, 0);
#pragma unroll
        for (int st = 0; st < 18; ++st) {
            if (st + 1 < 18) load(st + 1, (st + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);       // (pins the order: the scheduler otherwise sinks every read to just above its MFMA)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[st & 1][j], fa[st & 1][i], acc[j][i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        C32_STAMP(2);
    };
    auto epilogue_phase = [&](int t) {
        // ---- epilogue: D[filter][position].  Round (MODE 2: bias + leaky first), statistics of the rounded values, pack pairs, swap halves:
        // a lane holds filters 32 half + [0, 32) of its position.
#if defined(__HIP_DEVICE_COMPILE__)
        // the next tile's pieces (issued in front of this tile's MFMAs) and the previous tile's stores, BEFORE this tile's stores are issued --
        // nothing waits for a store it has just issued; stores and DMA share the counter and a counted wait over a mix of the two is not safe
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        C32_STAMP(3);
        const bool odd1 = (lane & 1) != 0, odd2 = (lane & 2) != 0;
        const unsigned lo = (unsigned)(64 * half + 16 * (lane & 3));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned q = (unsigned)(tb + wave * 64 + i * 32 + l31);
            const unsigned R = c32_div(q, mP, sP);
            const unsigned c = q - __umul24(R, P1);
            const unsigned img = c32_div(R, mH, sH);
            const unsigned r_ = R - __umul24(img, H1);
            const bool live = (q < (unsigned)e0) & (c < (unsigned)W) & (r_ < (unsigned)H);
            const unsigned m = __umul24(__umul24(img, (unsigned)H) + r_, (unsigned)W) + c;
            const unsigned moff = live ? m * 128u : 0xffffffffu;           // byte offset of this lane's position row (Mp < 2^24)
            const float lf = live ? 1.0f : 0.0f;
            const f32x2 lf2 = {lf, lf};
            unsigned ch[4][4];                                               // [16-byte chunk k: filters 32 half + 8 k ..][dword]
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x2 sv[8];
                if (MODE == 1) {
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const float4 v4 = *reinterpret_cast<const float4 *>(shl + 32 * half + 16 * j + 4 * r4);
                        sv[2 * r4] = f32x2{v4.x, v4.y}; sv[2 * r4 + 1] = f32x2{v4.z, v4.w};
                    }
                }
#pragma unroll
                for (int qd = 0; qd < 8; ++qd) {
                    float v0 = acc[j][i][2 * qd], v1 = acc[j][i][2 * qd + 1];
                    if (MODE == 2) {
                        v0 += cst[j][2 * qd]; v1 += cst[j][2 * qd + 1];
                        v0 = fmaxf(v0, alpha * v0); v1 = fmaxf(v1, alpha * v1);
                    }
                    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
                    const bf16x2 o2 = {(bf16)v0, (bf16)v1};                   // one v_cvt_pk_bf16_f32: the stored pair
                    const unsigned pk = __builtin_bit_cast(unsigned, o2);
                    ch[2 * j + (qd >> 2)][qd & 3] = pk;
                    if (MODE == 1) {
                        // statistics of the ROUNDED values, branch-free and two filters per instruction (packed f32): a dead position (zero
                        // column / row, beyond the batch) contributes 0 through its lane's factor
                        const f32x2 y2 = {__builtin_bit_cast(float, pk << 16), __builtin_bit_cast(float, pk & 0xffff0000u)};
                        const f32x2 d2 = (y2 - sv[qd]) * lf2;
                        t1[j][qd] += d2;
                        t2[j][qd] = __builtin_elementwise_fma(d2, d2, t2[j][qd]);
                    }
                    if ((qd & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // (bounds the scheduler's appetite: unrolled flat it spilled 53 registers)
                }
            }
            // 4 x 4 transpose over the four lanes of a quad (four consecutive positions), two butterfly stages of DPP quad permutes: lane b ends
            // with chunk b of the quad's positions 0..3 in registers 0..3.  Store k then writes position k of every quad from 4 + 4 lanes (the two
            // halves): a whole 128-byte row per position.  (A lane storing its own 64 bytes as four 16-byte pieces touches 32 rows per
            // instruction, 32 bytes each.)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
#pragma unroll
                for (int k = 0; k < 4; k += 2) {
                    const unsigned x = ch[k][d], y = ch[k + 1][d];
                    const unsigned xs = (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true), ys = (unsigned)__builtin_amdgcn_mov_dpp((int)y, 0xB1, 0xF, 0xF, true);
                    ch[k][d] = odd1 ? ys : x;
                    ch[k + 1][d] = odd1 ? y : xs;
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const unsigned x = ch[k][d], y = ch[k + 2][d];
                    const unsigned xs = (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xF, 0xF, true), ys = (unsigned)__builtin_amdgcn_mov_dpp((int)y, 0x4E, 0xF, 0xF, true);
                    ch[k][d] = odd2 ? ys : x;
                    ch[k + 2][d] = odd2 ? y : xs;
                }
            }
            unsigned oo[4];
            oo[0] = (unsigned)__builtin_amdgcn_mov_dpp((int)moff, 0x00, 0xF, 0xF, true);
            oo[1] = (unsigned)__builtin_amdgcn_mov_dpp((int)moff, 0x55, 0xF, 0xF, true);
            oo[2] = (unsigned)__builtin_amdgcn_mov_dpp((int)moff, 0xAA, 0xF, 0xF, true);
            oo[3] = (unsigned)__builtin_amdgcn_mov_dpp((int)moff, 0xFF, 0xF, 0xF, true);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned off = oo[k] + lo;                            // position k of the quad, this lane's 16 bytes of its row
                if (oo[k] != 0xffffffffu && !C32_ABL(2))
                    *reinterpret_cast<u32x4 *>(reinterpret_cast<unsigned char *>(O) + off) = u32x4{ch[k][0], ch[k][1], ch[k][2], ch[k][3]};
            }
        }
        C32_STAMP(4);
        C32_STAMP(5);
#endif
#ifdef Y2C32_EXPERIMENTS
        ph[7] += 1;
#endif
    };
    // One RAW barrier per tile (__syncthreads() would also wait for the output stores in flight, vmcnt(0)): every wave's pieces of tile t have landed
    // (each waited for its own in its previous epilogue) and every wave has left the rows the new pieces overwrite.
    for (int t = 0; t < T; ++t) {
        __builtin_amdgcn_s_barrier();
        C32_STAMP(0);
        mfma_phase(t);
        epilogue_phase(t);
        tb += C32_TP;
        tro += (unsigned)(C32_TP * 64);
        if (tro >= (unsigned)RRB) tro -= (unsigned)RRB;
    }
    if (MODE == 1 && bn_part) {
        // a lane holds the sums of its 32 filters over its position column: they meet across the 32 lanes of each half on the VALU (DPP: quad, quad pair,
        // row, then lane 15 of rows 0 / 2 into rows 1 / 3 -- six instructions per value).  The ds_bpermute butterfly this replaces -- 64 values x 5 steps
        // per wave, all through the LDS pipe -- took 12 us per workgroup (s_memtime stamps), a quarter of the launch.
        float v1[32], v2[32];
        auto halfsum = [](float v) {
            v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));        // quad_perm [1,0,3,2]
            v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));        // quad_perm [2,3,0,1]
            v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));       // row_half_mirror: the other quad of 8
            v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));       // row_mirror: the other 8 of 16
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false)); // row_bcast:15 into rows 1 and 3
            return v;                                                                                                           // lanes 16-31 / 48-63: the half's total
        };
#pragma unroll
        for (int k = 0; k < 32; ++k) { v1[k] = halfsum(t1[k >> 4][(k & 15) >> 1][k & 1]); v2[k] = halfsum(t2[k >> 4][(k & 15) >> 1][k & 1]); }
        // ... and across the eight waves in LDS (the ring is idle now) before anything leaves the CU: 128 atomics per workgroup instead of 1024 onto the
        // same 128 addresses (same-address f32 atomics serialise at L2, ~0.1 us each and more under contention: measured 20 us of this launch)
        __builtin_amdgcn_s_barrier();                              // every wave is past its last MFMA loop (raw: __syncthreads() would also wait for the output stores in flight)
        float *const red = reinterpret_cast<float *>(hb);           // [wave][plane][filter]
        if (l31 == 16) {
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                red[(wave * 2 + 0) * 64 + 32 * half + k] = v1[k];   // filter 32 half + 16 j + r with k = 16 j + r
                red[(wave * 2 + 1) * 64 + 32 * half + k] = v2[k];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (tid < 128) {
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) a += red[w * 128 + tid];
            const int slot = (int)(blockIdx.x & (C32_STAT_ROWS - 1));
            unsafeAtomicAdd(bn_part + ((tid >> 6) * Y2_BN_PART_ROWS + slot) * 64 + (tid & 63), a);
        }
    }
    C32_STAMP(5);
#ifdef Y2C32_EXPERIMENTS
    if ((abl & 8) && tid == 0) {
        ph[6] = __builtin_readcyclecounter() - t00;
        for (int k = 0; k < 8; ++k) c32_stamps[blockIdx.x * 8 + k] = ph[k];
    }
#endif
}

#ifdef Y2C32_EXPERIMENTS
extern "C" int yolo2_debug_c32_stamps(void *host_out) { return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(c32_stamps), sizeof(unsigned long long) * 256 * 8); }
#endif

bool y2_c32_shape(int Cp, int ldp, int Nf, int ldo, int ksize, int dtype) { return dtype == YOLO2_BF16 && ksize == 3 && Cp == 32 && ldp == 32 && Nf == 64 && ldo == 64; }

// -> 0 launched (*rows = partial rows touched when bn_part), 1 the shape does not fit this kernel's LDS plan (caller takes the generic kernels)
int y2_c32_fwd(const void *P, const void *F, void *O, int B, int H, int W, const float *bias, float alpha, const float *bn_shift, float *bn_part,
               int cus, int *rows, hipStream_t st) {
    const long Mp = (long)B * (H + 1) * (W + 1);
    if (Mp + 2 * C32_TP >= (1L << 24) || H < 1 || W < 1 || cus < 1) return 1;
    C32Geo g;
    g.H = H; g.W = W; g.M = B * H * W; g.Mp = (int)Mp;
    g.L = (int)(((Mp + cus - 1) / cus + C32_TP - 1) / C32_TP) * C32_TP;
    const int grid = (int)((Mp + g.L - 1) / g.L);
    g.HLa = (W + 2 + 15) & ~15;
    g.NP0 = (g.HLa + C32_TP + W + 2 + 15) / 16;
    g.RR = (16 * g.NP0 + C32_TP + 63) & ~63;                 // the rows of tile t (from its front halo on) and the new rows of tile t + 1
    const size_t lds = (size_t)C32_FBYTES + (size_t)g.RR * 64 + 256;      // filter image, the ring, 64 shifts
    if (lds > 160 * 1024) return 1;
    const int M = g.M;
#ifdef Y2C32_EXPERIMENTS
    static const int abl = y2_env_int("YOLO2_C32_ABL", 0);      // timing ablations (wrong results): 1 no MFMA loop, 2 no stores, 4 no DMA after the first tile, 8 phase stamps
#else
    const int abl = 0;
#endif
    y2_magic_u32((unsigned)(W + 1), &g.mP, &g.sP);
    y2_magic_u32((unsigned)(H + 1), &g.mH, &g.sH);
    const unsigned x_bytes = (unsigned)((size_t)M * 32 * 2), f_bytes = 64u * 288u * 2u;
    static std::atomic<size_t> lds_set[3][64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
#define C32_LAUNCH(MODEv, vecp)                                                                                                              \
    do {                                                                                                                                     \
        if (lds > lds_set[MODEv][dev].load(std::memory_order_relaxed)) {                                                                     \
            if (hipFuncSetAttribute((const void *)conv_c32_fwd_kernel<MODEv>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 1; \
            lds_set[MODEv][dev].store(lds, std::memory_order_relaxed);                                                                       \
        }                                                                                                                                    \
        conv_c32_fwd_kernel<MODEv><<<grid, 512, lds, st>>>((const bf16 *)P, x_bytes, (const bf16 *)F, f_bytes, (bf16 *)O, g, vecp, bn_part, alpha, abl); \
    } while (0)
    if (bn_part) C32_LAUNCH(1, bn_shift);
    else if (bias || alpha != 1.0f) C32_LAUNCH(2, bias);      // (an activation without a bias: the constants read as zeros)
    else C32_LAUNCH(0, (const float *)nullptr);
#undef C32_LAUNCH
    if (rows) *rows = grid < C32_STAT_ROWS ? grid : C32_STAT_ROWS;
    return 0;
}
