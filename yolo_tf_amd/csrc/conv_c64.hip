// 3x3 convolution for 64-channel inputs with the FILTER IN REGISTERS, bf16, gfx950: 128 filters (Darknet-19 conv2 / conv4 forward, 104 x 104 pixels per
// image) and 32 filters (conv1's data gradient: 64 -> 32 channels at 208 x 208); at the end of the file the same recipe for 128-channel inputs and 64
// filters (conv2 / conv4's data gradients), the K extent split over wave pairs (conv_c64_wide_kernel).
//
// Replaces slim.layers.conv2d of reference model/yolo2/inference.py:78,82 (conv2 / conv4: 64 -> 128, 3x3, SAME, no bias under batch_norm) and the input
// gradient of :76 (conv1), where the generic per-tap implicit-GEMM kernel pays a prologue and an epilogue per nine K steps: 43 us for 25.5 GFLOP
// (590 TFLOP/s) on a layer that moves 66 MB, 58 us for conv1's gradient (133 MB).  The persistent padded-index recipe of conv_c32.hip (conv1), with the
// changes the wider layer forces: 128 filters x 576 k are 147 KB -- they do not fit LDS beside the halo -- so
//   * a wave holds 32 filters x 576 k as 36 MFMA A-operand fragments (144 VGPRs), loaded once per workgroup lifetime.  128 filters: waves 0-3 / 4-7 hold
//     the four filter groups and take alternate 64-position blocks of a 256-position tile, two blocks (two passes of two 32 x 32 accumulators) per tile;
//     32 filters: every wave holds the whole filter and takes 32 positions of the tile (two accumulators over alternate K steps);
//   * a workgroup walks ONE CONTIGUOUS RANGE of the padded index, and LDS holds one RING of pixel rows (128 bytes each, source-side XOR swizzle) over it:
//     a tile needs the 2 (W + 2) positions around its own, which the previous tile has already staged -- only the 256 NEW rows of the next tile stream in by
//     LDS-DMA while this tile's MFMAs run on one pixel fragment read each (a fragment is used once per wave: 1 KiB per MFMA, like conv_c32.hip).  Two
//     whole halo images per tile (the first version of this file, and conv_c32.hip) staged 1.8x the input at 104 x 104 and do not fit at 208 x 208;
//   * the pixels run over the PADDED index q (row pitch W + 1, image pitch H + 1): nine constant row offsets into the ring, no (pixel, tap) mask;
//   * operand roles swapped and filter rows permuted (c64_slot_filter) so that a lane's 16 accumulator rows are 16 consecutive filters of its position:
//     32 contiguous bytes per lane, stored straight from registers after a 2 x 2 exchange inside lane pairs (each store instruction then writes 32
//     contiguous bytes per lane pair, 64 with the other half-wave); batch-norm partial sums per lane over its position column, met across lanes once
//     after the last tile (same partial-row contract as conv_igemm.hip).
// W / (W + 1) x H / (H + 1) of the MFMA work is real (98 % at 104 x 104).
#include "common.h"
#include "conv_shared.h"
#include <atomic>

#define C64_TP 256                 // padded positions per tile
#define C64_KSTEPS 36              // 9 taps x 4 sixteen-channel groups
#define C64_STAT_ROWS 32           // partial rows the statistics are spread over (the consumer's prologue sums every row in use: few rows, 8 workgroups each)
#define C64_FPITCH 1168            // LDS row pitch of a staged filter row (292 dwords = 36 mod 64: the 16 rows of a ds_read_b128 lane group hit 16 different bank quads)

// MFMA output row rho lands in register r = (rho & 3) + 4 (rho >> 3) of the lanes of half h = (rho >> 2) & 1: giving that row filter 16 h + r makes
// the 16 values of a lane 16 CONSECUTIVE filters (32 contiguous bytes of its position's row)
__device__ __forceinline__ int c64_slot_filter(int rho) { return 16 * ((rho >> 2) & 1) + (rho & 3) + 4 * (rho >> 3); }
__device__ __forceinline__ unsigned c64_div(unsigned q, unsigned m, unsigned s) { return __umulhi(q, m) >> s; }

struct C64Geo {
    int H, W, M, Mp;       // image rows / columns, pixels, padded positions of the batch
    int L;                 // padded positions per workgroup (a multiple of 128 / 256 for 128 / 32 filters)
    int RR;                // ring rows (a multiple of 64)
    int HLa;               // ring row of the workgroup's first position: W + 2 rounded up to 8
    int NP0;               // 8-row pieces staged before the first tile
    int shl_off;           // byte offset of the 128 statistics shifts
    unsigned mP, sP, mH, sH;
};

// MODE 0: plain store; 1: + batch-norm partial sums of the stored values (training forward); 2: + bias, leaky_relu (inference, folded BN)
// NFG: filter groups of 32 (4: 128 filters; 1: 32 filters)
template <int MODE, int NFG>
__global__ __launch_bounds__(512) void conv_c64_fwd_kernel(
    const bf16 *__restrict__ X, unsigned x_bytes, const bf16 *__restrict__ F, bf16 *__restrict__ O, C64Geo g,
    const float *__restrict__ vec, float *__restrict__ bn_part, float alpha) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    constexpr int NB = NFG == 4 ? 2 : 1;                    // 32-position blocks of a wave per pass
    constexpr int NPASS = NFG == 4 ? 2 : 1;                 // passes per tile
    constexpr int ROWB = NFG * 64;                          // bytes per output pixel
    const int H = g.H, W = g.W, Mp = g.Mp;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fg = NFG == 4 ? (wave & 3) : 0;              // filter group (32 filters)
    const int pg = NFG == 4 ? (wave >> 2) : wave;          // position group: 64-position blocks alternate between the two groups / 32 positions of eight
    const int half = lane >> 5, l31 = lane & 31;
    const unsigned P1 = (unsigned)(W + 1), H1 = (unsigned)(H + 1);
    const int s0 = (int)blockIdx.x * g.L;                   // this workgroup's positions [s0, e0)
    const int e0 = min(s0 + g.L, Mp);
    if (s0 >= Mp) return;
    const int RRB = g.RR * 128, NPR = g.RR >> 3;             // ring bytes, ring pieces
    const int pbase = s0 - g.HLa;                          // padded position of ring row 0 (may be negative: those rows stage as zeros)

    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16 *>(X), 0, x_bytes, 0x00020000);
    typedef __attribute__((address_space(3))) void *lds_void_ptr;
    typedef const __attribute__((address_space(3))) bf16x8 *lds_frag_ptr;

    // ---- the filter, once, into registers: fragment st = (tap, 16-channel group kk) of MFMA row l31 = filter 32 fg + c64_slot_filter(l31), k = 8 half .. + 7
    // (F rows are [tap][channel] for a 64-channel layer: common.h y2_filter_koff; the data gradient's operand has the same form, taps flipped).  Read
    // straight from global memory a lane would touch 36 separate 16-byte pieces 1152 bytes apart from its neighbour's -- measured ~9 us per workgroup,
    // a third of the launch at batch 16 -- so the rows go through LDS: rounds of 64 filters (one of 32 for NFG = 1), each staged by LDS-DMA (coalesced
    // at the source) into the space behind the first tile's ring rows with a row pitch of 1168 bytes (the 16 padding bytes of a row read as zeros),
    // then picked up by the waves that hold those filter groups.  The first tile's rows stream into the ring meanwhile (issued before this block's waits).
    bf16x8 filt[C64_KSTEPS];
    auto load_filter = [&]() {
        constexpr int RPR = NFG == 4 ? 64 : 32, FROUND = RPR * C64_FPITCH;
        const __amdgpu_buffer_rsrc_t rsrcF = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16 *>(F), 0, (unsigned)(NFG * 32) * 1152u, 0x00020000);
        unsigned char *const fst = smem + g.NP0 * 1024;            // ring rows behind the first tile's are idle until tile 1 is staged
#pragma unroll 1
        for (int rd = 0; rd < (NFG == 4 ? 2 : 1); ++rd) {
            for (int p = wave; p * 1024 < FROUND; p += 8) {
                const int pos = p * 1024 + lane * 16;
                const int n = pos / C64_FPITCH, off = pos - n * C64_FPITCH;
                const unsigned voff = (off < 1152 && n < RPR) ? (unsigned)((RPR * rd + n) * 1152 + off) : Y2_OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcF, (lds_void_ptr)(fst + p * 1024), 16, voff, 0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                       // the round's rows are in LDS
            if (NFG == 1 || (fg >> 1) == rd) {
                const unsigned a0 = y2_lds_addr(fst) + (unsigned)((32 * (fg & 1) + c64_slot_filter(l31)) * C64_FPITCH + 16 * half);
#pragma unroll
                for (int st = 0; st < C64_KSTEPS; ++st) filt[st] = *(lds_frag_ptr)(uintptr_t)(a0 + (unsigned)(st * 32));
            }
            __syncthreads();                                       // ... and read: the next round (or tile 1's rows) may overwrite them
        }
    };

    // ---- ring staging: piece j = ring rows 8 j .. 8 j + 7 (128 bytes each) = padded positions pbase + 8 j ..; lane = (row = lane >> 3, 16-byte chunk =
    // lane & 7), source chunk swizzled with (ring row >> 1) & 7 so that the 16 rows a ds_read_b128 lane group touches land on 16 different bank quads
    // (RR is a multiple of 16: the term survives the wrap).  A staged row is a pixel, or zeros (zero column, zero row, outside the batch).
    const int prow = lane >> 3;
    auto stage_piece = [&](int j, int slot) {
        const int row = j * 8 + prow;
        const unsigned q = (unsigned)(pbase + row);                           // wraps below 0: fails the range test
        const unsigned R = c64_div(q, g.mP, g.sP);                            // padded image row over the whole batch
        const unsigned img = c64_div(R, g.mH, g.sH);
        const unsigned c = q - __umul24(R, P1), r = R - __umul24(img, H1);
        const bool ok = (q < (unsigned)Mp) & (c < (unsigned)W) & (r < (unsigned)H);
        const unsigned m = q - R - __umul24(img, (unsigned)W);                // pixel index: q = R (W + 1) + c, R = img (H + 1) + r, m = (img H + r) W + c
        const unsigned voff = ok ? m * 128u + (unsigned)((((lane & 7) ^ ((row >> 1) & 7))) << 4) : Y2_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (lds_void_ptr)(smem + slot * 1024), 16, voff, 0, 0, 0);
    };
    for (int j = wave; j < g.NP0; j += 8) stage_piece(j, j);               // rows [0, HLa + TP + W + 2): the first tile and both its halos (NP0 <= NPR)
    load_filter();                                                   // (its waits also cover this wave's pieces of the first tile)

    // ---- read addresses (ring offset 0, position block 0), relative to the ring: one per tap; the 16-channel group kk is XORed in (bits 5-6), the lane's
    // half in bit 4.  A tile / pass / block moves them by a multiple of 32 rows (the swizzle term is the same 32 rows on), wrapped at the ring's end.
    const unsigned lds0 = y2_lds_addr(smem);
    unsigned abase[9];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
        const int shift = (tp / 3 - 1) * (W + 1) + (tp % 3 - 1);
        const int row = pg * (NFG == 4 ? 64 : 32) + l31 + g.HLa + shift;
        abase[tp] = (unsigned)(row * 128 + ((half ^ ((row >> 1) & 7)) << 4));
    }

    // per-filter constants of this lane's 16 accumulator rows (filter 32 fg + 16 half + r): MODE 2 the bias, MODE 1 the statistics' shift (moving mean)
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    // (MODE 1 keeps the shift in LDS behind the ring and reads it per epilogue: its sum registers leave no room for 16 more beside the filter)
    f32x2 cst[8];                                            // packed pairs: the epilogue runs on v_pk_add_f32 / v_pk_fma_f32
    float *const shl = reinterpret_cast<float *>(smem + g.shl_off);
    float tq[8];                                             // MODE 1: running sum of (plane = lane bit 0, filter 2 qd + lane bit 1) over this lane's quad of positions
#pragma unroll
    for (int qd = 0; qd < 8; ++qd) tq[qd] = 0.f;
    if (MODE == 1) {
        if (tid < 128) shl[tid] = vec ? vec[tid] : 0.f;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the loop's barriers are raw: this write is on its way before the first one)
    }
#pragma unroll
    for (int r2 = 0; r2 < 8; ++r2) {
        const int f = 32 * fg + 16 * half + 2 * r2;
        cst[r2] = (MODE == 2 && vec) ? f32x2{vec[f], vec[f + 1]} : f32x2{0.f, 0.f};
    }

    const int T = (e0 - s0 + C64_TP - 1) / C64_TP;          // tiles of this workgroup (the last one may be partial: whole passes are skipped)
    int pslot = g.NP0;                                      // ring piece the next tile's first new piece goes to (NP0 + 32 t, wrapped)
    if (pslot >= NPR) pslot -= NPR;
    unsigned tro = 0;                                       // ring byte offset of the current tile (256 t rows, wrapped)
    // One RAW barrier per tile (__syncthreads() would also wait for the output stores in flight): every wave's pieces of tile t have landed (each waited
    // for its own at the end of its previous tile) and every wave has left the rows the new pieces overwrite.
    for (int t = 0; t < T; ++t) {
        __builtin_amdgcn_s_barrier();
        const int tb = s0 + t * C64_TP;
        if (t + 1 < T) {                                     // this wave's pieces of the next tile's 256 new rows
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
#pragma unroll 1
        for (int pa = 0; pa < NPASS; ++pa) {
            if (tb + pa * 128 >= e0) break;                 // (partial last tile)
            f32x16 acc[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            unsigned poff = tro + (unsigned)(pa * 16384);      // pass pa: blocks 128 positions on
            if (poff >= (unsigned)RRB) poff -= (unsigned)RRB;
            // 36 steps (tap, 16-channel group) of NB fragment reads + NB MFMAs (NFG = 1: one read, the accumulators alternate); the reads of step s + 1
            // are issued ahead of the MFMAs of step s (two fragment sets).  A tap's addresses are formed when its first group is read.
            bf16x8 fb[2][NB];
            unsigned ad[NB];
            auto load = [&](int st, int set) {
                const int tp = st >> 2, kk = st & 3;
                if (kk == 0) {
                    unsigned x = abase[tp] + poff;
#pragma unroll
                    for (int i = 0; i < NB; ++i) {
                        x = min(x, x - (unsigned)RRB);              // wrap (x < ring: the difference wraps to a huge value)
                        ad[i] = x + lds0;
                        x += 4096u;
                    }
                }
#pragma unroll
                for (int i = 0; i < NB; ++i) fb[set][i] = *(lds_frag_ptr)(uintptr_t)(ad[i] ^ (unsigned)(kk * 32));
            };
            load(0, 0);
#pragma unroll
            for (int st = 0; st < C64_KSTEPS; ++st) {
                if (st + 1 < C64_KSTEPS) load(st + 1, (st + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);       // (pins the order: the scheduler otherwise sinks every read to just above its MFMA)
                if (NFG == 4) {
#pragma unroll
                    for (int i = 0; i < NB; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(filt[st], fb[st & 1][i], acc[i], 0, 0, 0);
                } else {
                    acc[st & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(filt[st], fb[st & 1][0], acc[st & 1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (NFG == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][r] += acc[1][r];
            }
            // ---- epilogue of the pass: D[filter][position].  Round (MODE 2: bias + leaky first), statistics of the rounded values, pack pairs: a lane holds
            // filters 32 fg + 16 half + [0, 16) of its position = two 16-byte chunks.
#if defined(__HIP_DEVICE_COMPILE__)
            if (pa == NPASS - 1 || tb + (pa + 1) * 128 >= e0)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the next tile's pieces (issued in front of this tile's MFMAs) and the previous stores: nothing waits for a store it has just issued
            const bool odd = (lane & 1) != 0;
            // (1) round the position blocks to packed bf16 pairs (MODE 2: bias + leaky first): the accumulators are dead after this
            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
            unsigned pk[NB][8];                                                 // [position block][filter pair]: two 16-byte chunks per block
            float lfi[NB];
            unsigned moffi[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const unsigned q = (unsigned)(tb + (NFG == 4 ? pa * 128 + pg * 64 + i * 32 : pg * 32) + l31);
                const unsigned R = c64_div(q, g.mP, g.sP);
                const unsigned c = q - __umul24(R, P1);
                const unsigned img = c64_div(R, g.mH, g.sH);
                const unsigned r_ = R - __umul24(img, H1);
                const bool live = (q < (unsigned)e0) & (c < (unsigned)W) & (r_ < (unsigned)H);
                const unsigned m = __umul24(__umul24(img, (unsigned)H) + r_, (unsigned)W) + c;
                moffi[i] = live ? m * (unsigned)ROWB + (unsigned)(fg * 64 + half * 32) : 0xffffffffu;      // byte offset of this lane's 32 bytes (Mp < 2^23)
                lfi[i] = live ? 1.0f : 0.0f;
#pragma unroll
                for (int qd = 0; qd < 8; ++qd) {
                    float v0 = acc[i][2 * qd], v1 = acc[i][2 * qd + 1];
                    if (MODE == 2) {
                        v0 += cst[qd][0]; v1 += cst[qd][1];
                        v0 = fmaxf(v0, alpha * v0); v1 = fmaxf(v1, alpha * v1);
                    }
                    const bf16x2 o2 = {(bf16)v0, (bf16)v1};                   // one v_cvt_pk_bf16_f32: the stored pair
                    pk[i][qd] = __builtin_bit_cast(unsigned, o2);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // (2) stores.  2 x 2 exchange inside lane pairs (two consecutive positions): store 0 carries the EVEN lane's position -- the even lane its chunk 0,
            // the odd lane the even lane's chunk 1 -- store 1 the odd lane's position likewise: each store instruction writes 32 contiguous bytes per lane
            // pair (64 with the other half-wave) instead of 16 bytes per lane in 32 different rows
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                unsigned s0_[4], s1_[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const unsigned x = pk[i][d], y = pk[i][4 + d];
                    const unsigned xs = (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true), ys = (unsigned)__builtin_amdgcn_mov_dpp((int)y, 0xB1, 0xF, 0xF, true);
                    s0_[d] = odd ? ys : x;
                    s1_[d] = odd ? y : xs;
                }
                const unsigned mo_even = (unsigned)__builtin_amdgcn_mov_dpp((int)moffi[i], 0xA0, 0xF, 0xF, true);      // quad_perm [0,0,2,2]: the pair's even lane
                const unsigned mo_odd = (unsigned)__builtin_amdgcn_mov_dpp((int)moffi[i], 0xF5, 0xF, 0xF, true);       // quad_perm [1,1,3,3]: the pair's odd lane
                const unsigned lo16 = odd ? 16u : 0u;
                if (mo_even != 0xffffffffu)
                    *reinterpret_cast<u32x4 *>(reinterpret_cast<unsigned char *>(O) + mo_even + lo16) = u32x4{s0_[0], s0_[1], s0_[2], s0_[3]};
                if (mo_odd != 0xffffffffu)
                    *reinterpret_cast<u32x4 *>(reinterpret_cast<unsigned char *>(O) + mo_odd + lo16) = u32x4{s1_[0], s1_[1], s1_[2], s1_[3]};
            }
            __builtin_amdgcn_sched_barrier(0);
            // (3) statistics of the ROUNDED values, branch-free, two filters per instruction (a dead position contributes 0 through its lane's factor).  No
            // register file for 32 running sums per lane, no LDS atomics (measured: ~3 cycles per lane and add): the four lanes of a quad -- four
            // consecutive positions -- fold their four values per filter pair (sum / square x two filters) with two DPP exchanges, after which lane
            // (bit 1, bit 0) of the quad carries (filter parity, plane) of the pair: ONE running sum per filter pair and lane, eight registers
            if (MODE == 1 && bn_part) {
                const bool odd2 = (lane & 2) != 0;
#pragma unroll
                for (int qd = 0; qd < 8; ++qd) {
                    const f32x2 svq = *reinterpret_cast<const f32x2 *>(shl + 32 * fg + 16 * half + 2 * qd);
                    f32x2 a1 = {0.f, 0.f}, a2 = {0.f, 0.f};
#pragma unroll
                    for (int i = 0; i < NB; ++i) {
                        const f32x2 lf2 = {lfi[i], lfi[i]};
                        const f32x2 y2 = {__builtin_bit_cast(float, pk[i][qd] << 16), __builtin_bit_cast(float, pk[i][qd] & 0xffff0000u)};
                        const f32x2 d2 = (y2 - svq) * lf2;
                        a1 += d2;
                        a2 = __builtin_elementwise_fma(d2, d2, a2);
                    }
                    // lanes differing in bit 0: the even lane keeps the sums, the odd lane the squares
                    float k0, k1;
                    {
                        const float t0_ = odd ? a1[0] : a2[0], t1_ = odd ? a1[1] : a2[1];
                        const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, t0_), 0xB1, 0xF, 0xF, true));
                        const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, t1_), 0xB1, 0xF, 0xF, true));
                        k0 = (odd ? a2[0] : a1[0]) + r0;
                        k1 = (odd ? a2[1] : a1[1]) + r1;
                    }
                    // lanes differing in bit 1: bit 1 clear keeps filter 2 qd, set keeps filter 2 qd + 1
                    {
                        const float s_ = odd2 ? k0 : k1;
                        const float r_ = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s_), 0x4E, 0xF, 0xF, true));
                        tq[qd] += (odd2 ? k1 : k0) + r_;
                    }
                }
            }
#endif
        }
        tro += (unsigned)(C64_TP * 128);
        if (tro >= (unsigned)RRB) tro -= (unsigned)RRB;
    }
    if (MODE == 1 && bn_part) {
        // the quads of a half-wave meet (three exchange steps over the eight quads, all eight values in flight per step); the first quad of each half writes
#pragma unroll
        for (int o = 4; o <= 16; o <<= 1) {
            float x[8];
#pragma unroll
            for (int qd = 0; qd < 8; ++qd) x[qd] = __shfl_xor(tq[qd], o, 64);
#pragma unroll
            for (int qd = 0; qd < 8; ++qd) tq[qd] += x[qd];
        }
        if (l31 < 4) {
            const int slot = (int)(blockIdx.x & (C64_STAT_ROWS - 1));
            const int plane = lane & 1, par = (lane >> 1) & 1;
#pragma unroll
            for (int qd = 0; qd < 8; ++qd)
                unsafeAtomicAdd(bn_part + (plane * Y2_BN_PART_ROWS + slot) * 128 + 32 * fg + 16 * half + 2 * qd + par, tq[qd]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// 128-channel inputs, 64 filters: the data gradients of conv2 / conv4 (dY 128 channels at 104 x 104 -> dX 64 channels; reference
// model/yolo2/inference.py:78,82 under tf.gradients), as the forward convolution of dY with the flipped operand.
// ---------------------------------------------------------------------------------------------------
// The operand is 64 filters x 1152 k = 147 KB again, but now K is twice a wave's register budget: the eight waves are (filter group of 32) x (CHANNEL
// HALF) x (position group).  A wave holds 32 filters x the 576 k of its 64-channel half (K order of the operand: channel chunk outermost), reads only its
// half of every pixel row -- the ring is kept as two planes of 128-byte half rows, each in the format of the kernel above -- and ends a tile with a partial
// sum over half of K.  The two waves of a pair then swap one 32 x 32 block each through LDS (4 KB per wave) and finish one block each: the additions, the
// rounding and the stores are split between them as well.  128-position tiles (two planes of ring + the exchange area are exactly 160 KB at 104 x 104).
#define C64W_TP 128
__global__ __launch_bounds__(512) void conv_c64_wide_kernel(const bf16 *__restrict__ X, unsigned x_bytes, const bf16 *__restrict__ F, bf16 *__restrict__ O, C64Geo g) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    constexpr int TP = C64W_TP;
    const int H = g.H, W = g.W, Mp = g.Mp;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fg = wave & 1, kh = (wave >> 1) & 1, pg = wave >> 2;      // filter group (32 of 64), channel half, position group (64 of the tile's 128)
    const int half = lane >> 5, l31 = lane & 31;
    const unsigned P1 = (unsigned)(W + 1), H1 = (unsigned)(H + 1);
    const int s0 = (int)blockIdx.x * g.L;
    const int e0 = min(s0 + g.L, Mp);
    if (s0 >= Mp) return;
    const int RRB = g.RR * 128, NPR = g.RR >> 3;             // bytes / pieces of ONE plane
    const int pbase = s0 - g.HLa;
    unsigned char *const xchg = smem + 2 * RRB;             // 8 x 4 KB

    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16 *>(X), 0, x_bytes, 0x00020000);
    typedef __attribute__((address_space(3))) void *lds_void_ptr;
    typedef const __attribute__((address_space(3))) bf16x8 *lds_frag_ptr;

    // ---- ring staging: piece j = rows 8 j .. 8 j + 7 of BOTH planes (plane p = channels 64 p .. 64 p + 63 of the row: 128 bytes, swizzled like above)
    const int prow = lane >> 3;
    auto stage_piece = [&](int j, int slot) {
        const int row = j * 8 + prow;
        const unsigned q = (unsigned)(pbase + row);
        const unsigned R = c64_div(q, g.mP, g.sP);
        const unsigned img = c64_div(R, g.mH, g.sH);
        const unsigned c = q - __umul24(R, P1), r = R - __umul24(img, H1);
        const bool ok = (q < (unsigned)Mp) & (c < (unsigned)W) & (r < (unsigned)H);
        const unsigned m = q - R - __umul24(img, (unsigned)W);
        const unsigned voff = ok ? m * 256u + (unsigned)((((lane & 7) ^ ((row >> 1) & 7))) << 4) : Y2_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (lds_void_ptr)(smem + slot * 1024), 16, voff, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (lds_void_ptr)(smem + RRB + slot * 1024), 16, ok ? voff + 128u : Y2_OOB, 0, 0, 0);
    };
    for (int j = wave; j < g.NP0; j += 8) stage_piece(j, j);

    // ---- the filter into registers: four rounds (filter group, channel half) of 32 rows x 1152 bytes, staged at pitch 1168 behind the first tile's rows of
    // plane 1 (that plane's tail and the exchange area are idle until the first tile ends), picked up by the two waves that hold that (group, half)
    bf16x8 filt[C64_KSTEPS];
    {
        constexpr int FROUND = 32 * C64_FPITCH;
        const __amdgpu_buffer_rsrc_t rsrcF = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16 *>(F), 0, 64u * 2304u, 0x00020000);
        unsigned char *const fst = smem + RRB + g.NP0 * 1024;
#pragma unroll 1
        for (int rd = 0; rd < 4; ++rd) {
            const int fgr = rd & 1, khr = rd >> 1;
            for (int p = wave; p * 1024 < FROUND; p += 8) {
                const int pos = p * 1024 + lane * 16;
                const int n = pos / C64_FPITCH, off = pos - n * C64_FPITCH;
                const unsigned voff = (off < 1152 && n < 32) ? (unsigned)((32 * fgr + n) * 2304 + khr * 1152 + off) : Y2_OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcF, (lds_void_ptr)(fst + p * 1024), 16, voff, 0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (fg == fgr && kh == khr) {
                const unsigned a0 = y2_lds_addr(fst) + (unsigned)(c64_slot_filter(l31) * C64_FPITCH + 16 * half);
#pragma unroll
                for (int st = 0; st < C64_KSTEPS; ++st) filt[st] = *(lds_frag_ptr)(uintptr_t)(a0 + (unsigned)(st * 32));
            }
            __syncthreads();
        }
    }

    const unsigned lds0 = y2_lds_addr(smem) + (unsigned)(kh * RRB);      // this wave's plane
    unsigned abase[9];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
        const int shift = (tp / 3 - 1) * (W + 1) + (tp % 3 - 1);
        const int row = pg * 64 + l31 + g.HLa + shift;
        abase[tp] = (unsigned)(row * 128 + ((half ^ ((row >> 1) & 7)) << 4));
    }

    const int T = (e0 - s0 + TP - 1) / TP;
    int pslot = g.NP0;
    if (pslot >= NPR) pslot -= NPR;
    unsigned tro = 0;
    for (int t = 0; t < T; ++t) {
        __builtin_amdgcn_s_barrier();                       // (raw: every wave's pieces of this tile have landed -- each waited for its own before its last stores)
        const int tb = s0 + t * TP;
        if (t + 1 < T) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {                    // the next tile's 128 new rows: 16 pieces x two planes
                const int k = wave + 8 * u;
                int sl = pslot + k;
                if (sl >= NPR) sl -= NPR;
                stage_piece(g.NP0 + t * (TP / 8) + k, sl);
            }
        }
        pslot += TP / 8;
        if (pslot >= NPR) pslot -= NPR;
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        bf16x8 fb[2][2];
        unsigned ad[2];
        auto load = [&](int st, int set) {
            const int tp = st >> 2, kk = st & 3;
            if (kk == 0) {
                unsigned x = abase[tp] + tro;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    x = min(x, x - (unsigned)RRB);
                    ad[i] = x + lds0;
                    x += 4096u;
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) fb[set][i] = *(lds_frag_ptr)(uintptr_t)(ad[i] ^ (unsigned)(kk * 32));
        };
        load(0, 0);
#pragma unroll
        for (int st = 0; st < C64_KSTEPS; ++st) {
            if (st + 1 < C64_KSTEPS) load(st + 1, (st + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(filt[st], fb[st & 1][i], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#if defined(__HIP_DEVICE_COMPILE__)
        // ---- the pair (same filters, same positions, the other channel half) swaps one block each: this wave keeps block kh and gives block 1 - kh
        f32x16 keep, give;
#pragma unroll
        for (int r = 0; r < 16; ++r) { keep[r] = kh ? acc[1][r] : acc[0][r]; give[r] = kh ? acc[0][r] : acc[1][r]; }
        {
            f32x4 *const xw = reinterpret_cast<f32x4 *>(xchg + wave * 4096);
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) xw[q4 * 64 + lane] = f32x4{give[4 * q4], give[4 * q4 + 1], give[4 * q4 + 2], give[4 * q4 + 3]};
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const f32x4 *const xr = reinterpret_cast<const f32x4 *>(xchg + (wave ^ 2) * 4096);
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const f32x4 v = xr[q4 * 64 + lane];
                keep[4 * q4] += v[0]; keep[4 * q4 + 1] += v[1]; keep[4 * q4 + 2] += v[2]; keep[4 * q4 + 3] += v[3];
            }
        }
        // ---- epilogue of block kh: round, pack pairs, 2 x 2 exchange inside lane pairs, 16-byte stores (as in the kernel above)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the next tile's pieces and the previous stores
        {
            const bool odd = (lane & 1) != 0;
            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
            const unsigned q = (unsigned)(tb + pg * 64 + kh * 32 + l31);
            const unsigned R = c64_div(q, g.mP, g.sP);
            const unsigned c = q - __umul24(R, P1);
            const unsigned img = c64_div(R, g.mH, g.sH);
            const unsigned r_ = R - __umul24(img, H1);
            const bool live = (q < (unsigned)e0) & (c < (unsigned)W) & (r_ < (unsigned)H);
            const unsigned m = __umul24(__umul24(img, (unsigned)H) + r_, (unsigned)W) + c;
            const unsigned moff = live ? m * 128u + (unsigned)(fg * 64 + half * 32) : 0xffffffffu;
            unsigned pk[8];
#pragma unroll
            for (int qd = 0; qd < 8; ++qd) {
                const bf16x2 o2 = {(bf16)keep[2 * qd], (bf16)keep[2 * qd + 1]};
                pk[qd] = __builtin_bit_cast(unsigned, o2);
            }
            unsigned s0_[4], s1_[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const unsigned x = pk[d], y = pk[4 + d];
                const unsigned xs = (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true), ys = (unsigned)__builtin_amdgcn_mov_dpp((int)y, 0xB1, 0xF, 0xF, true);
                s0_[d] = odd ? ys : x;
                s1_[d] = odd ? y : xs;
            }
            const unsigned mo_even = (unsigned)__builtin_amdgcn_mov_dpp((int)moff, 0xA0, 0xF, 0xF, true);
            const unsigned mo_odd = (unsigned)__builtin_amdgcn_mov_dpp((int)moff, 0xF5, 0xF, 0xF, true);
            const unsigned lo16 = odd ? 16u : 0u;
            if (mo_even != 0xffffffffu)
                *reinterpret_cast<u32x4 *>(reinterpret_cast<unsigned char *>(O) + mo_even + lo16) = u32x4{s0_[0], s0_[1], s0_[2], s0_[3]};
            if (mo_odd != 0xffffffffu)
                *reinterpret_cast<u32x4 *>(reinterpret_cast<unsigned char *>(O) + mo_odd + lo16) = u32x4{s1_[0], s1_[1], s1_[2], s1_[3]};
        }
#endif
        tro += (unsigned)(TP * 128);
        if (tro >= (unsigned)RRB) tro -= (unsigned)RRB;
    }
}

// -> 0 launched, 1 the shape does not fit the LDS plan (caller takes the generic kernels)
static int y2_c64_wide(const void *P, const void *F, void *O, int B, int H, int W, int cus, hipStream_t st) {
    const long Mp = (long)B * (H + 1) * (W + 1);
    if (Mp + 2 * C64W_TP >= (1L << 23) || H < 1 || W < 1 || cus < 1) return 1;
    C64Geo g;
    g.H = H; g.W = W; g.M = B * H * W; g.Mp = (int)Mp;
    g.L = (int)(((Mp + cus - 1) / cus + C64W_TP - 1) / C64W_TP) * C64W_TP;
    const int grid = (int)((Mp + g.L - 1) / g.L);
    g.HLa = (W + 2 + 7) & ~7;
    g.NP0 = (g.HLa + C64W_TP + W + 2 + 7) / 8;
    g.RR = (8 * g.NP0 + C64W_TP + 63) & ~63;
    g.shl_off = 0;
    const size_t plane = (size_t)g.RR * 128, lds = 2 * plane + 8 * 4096;
    // room for a filter round (32 rows at pitch 1168 = 37 pieces) behind the first tile's rows of plane 1
    if (lds > 160 * 1024 || (plane - (size_t)g.NP0 * 1024) + 8 * 4096 < 37 * 1024) return 1;
    y2_magic_u32((unsigned)(W + 1), &g.mP, &g.sP);
    y2_magic_u32((unsigned)(H + 1), &g.mH, &g.sH);
    static std::atomic<size_t> lds_set[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (lds > lds_set[dev].load(std::memory_order_relaxed)) {
        if (hipFuncSetAttribute((const void *)conv_c64_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 1;
        lds_set[dev].store(lds, std::memory_order_relaxed);
    }
    conv_c64_wide_kernel<<<grid, 512, lds, st>>>((const bf16 *)P, (unsigned)((size_t)g.M * 128 * 2), (const bf16 *)F, (bf16 *)O, g);
    return 0;
}

// 64 -> 128 (the forward of conv2 / conv4) or 64 -> 32 (as the forward convolution the data gradient of conv1 is: dY in, flipped taps in the operand)
bool y2_c64_shape(int Cp, int ldp, int Nf, int ldo, int ksize, int dtype) {
    if (dtype != YOLO2_BF16 || ksize != 3) return false;
    if (Cp == 128 && ldp == 128) return Nf == 64 && ldo == 64;      // ... or 128 -> 64: the data gradients of conv2 / conv4 (conv_c64_wide_kernel)
    return Cp == 64 && ldp == 64 && ((Nf == 128 && ldo == 128) || (Nf == 32 && ldo == 32));
}

// -> 0 launched (*rows = partial rows touched when bn_part), 1 the shape does not fit this kernel's LDS plan or its 32-filter form has no such epilogue
// (caller takes the generic kernels)
int y2_c64_fwd(const void *P, const void *F, void *O, int B, int H, int W, int Nf, const float *bias, float alpha, const float *bn_shift, float *bn_part,
               int cus, int *rows, hipStream_t st) {
    if (Nf == 64) {      // 128-channel input (y2_c64_shape): plain stores only
        if (bn_part || bias || alpha != 1.0f) return 1;
        if (rows) *rows = 0;      // (no statistics)
        return y2_c64_wide(P, F, O, B, H, W, cus, st);
    }
    const long Mp = (long)B * (H + 1) * (W + 1);
    if (Mp + 2 * C64_TP >= (1L << 23) || H < 1 || W < 1 || cus < 1) return 1;
    const bool narrow = Nf == 32;
    if (narrow && (bn_part || bias || alpha != 1.0f)) return 1;
    C64Geo g;
    g.H = H; g.W = W; g.M = B * H * W; g.Mp = (int)Mp;
    const int gran = narrow ? 256 : 128;
    g.L = (int)(((Mp + cus - 1) / cus + gran - 1) / gran) * gran;
    const int grid = (int)((Mp + g.L - 1) / g.L);
    g.HLa = (W + 2 + 7) & ~7;
    g.NP0 = (g.HLa + C64_TP + W + 2 + 7) / 8;
    g.RR = (8 * g.NP0 + C64_TP + 63) & ~63;                  // the rows of tile t (from its front halo on) and the new rows of tile t + 1
    const size_t fround = (size_t)(narrow ? 32 : 64) * C64_FPITCH;
    size_t lds = (size_t)g.RR * 128;
    if (lds < (size_t)g.NP0 * 1024 + (fround + 1023) / 1024 * 1024) lds = (size_t)g.NP0 * 1024 + (fround + 1023) / 1024 * 1024;      // ... and room for a filter round behind the first tile's rows
    g.shl_off = (int)lds;
    lds += 512;                                              // 128 shifts
    if (lds > 160 * 1024) return 1;
    y2_magic_u32((unsigned)(W + 1), &g.mP, &g.sP);
    y2_magic_u32((unsigned)(H + 1), &g.mH, &g.sH);
    const unsigned x_bytes = (unsigned)((size_t)g.M * 64 * 2);
    static std::atomic<size_t> lds_set[4][64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
#define C64_LAUNCH(SLOT, MODEv, NFGv, vecp)                                                                                                  \
    do {                                                                                                                                     \
        if (lds > lds_set[SLOT][dev].load(std::memory_order_relaxed)) {                                                                      \
            if (hipFuncSetAttribute((const void *)conv_c64_fwd_kernel<MODEv, NFGv>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 1; \
            lds_set[SLOT][dev].store(lds, std::memory_order_relaxed);                                                                        \
        }                                                                                                                                    \
        conv_c64_fwd_kernel<MODEv, NFGv><<<grid, 512, lds, st>>>((const bf16 *)P, x_bytes, (const bf16 *)F, (bf16 *)O, g, vecp, bn_part, alpha); \
    } while (0)
    if (narrow) C64_LAUNCH(3, 0, 1, (const float *)nullptr);
    else if (bn_part) C64_LAUNCH(1, 1, 4, bn_shift);
    else if (bias || alpha != 1.0f) C64_LAUNCH(2, 2, 4, bias);      // (an activation without a bias: the constants read as zeros)
    else C64_LAUNCH(0, 0, 4, (const float *)nullptr);
#undef C64_LAUNCH
    if (rows) *rows = grid < C64_STAT_ROWS ? grid : C64_STAT_ROWS;
    return 0;
}
