// 3x3 convolution FORWARD for 64-channel inputs and 128 filters (Darknet-19 conv2 / conv4: 104 x 104 pixels per image), bf16, gfx950.
//
// Replaces slim.layers.conv2d of reference model/yolo2/inference.py:78,82 (conv2 / conv4: 64 -> 128, 3x3, SAME, no bias under batch_norm) where the
// generic per-tap implicit-GEMM kernel pays a prologue and an epilogue per nine K steps: 43 us for 25.5 GFLOP (590 TFLOP/s) on a layer that moves
// 66 MB.  The persistent padded-index recipe of conv_c32.hip (conv1), with the one change the wider layer forces: 128 filters x 576 k are 147 KB -- they
// do not fit LDS beside the halo buffers -- so THE FILTER LIVES IN REGISTERS:
//   * a wave holds 32 filters x 576 k as 36 MFMA A-operand fragments (144 VGPRs), loaded once per workgroup lifetime; waves 0-3 / 4-7 hold the four
//     filter groups and take the first / second 128 positions of a 256-position tile, 64 positions (two 32 x 32 accumulators) per pass;
//   * LDS holds nothing but two halo buffers (256 + 2 (W + 2) rows of 128 bytes, source-side XOR swizzle): the next tile's halo streams in by LDS-DMA
//     while this tile's 288 MFMAs per wave run on one pixel fragment read each (a fragment is used once per wave: 1 KiB per MFMA, like conv_c32.hip);
//   * the pixels run over the PADDED index q (row pitch W + 1, image pitch H + 1): nine constant row offsets into the staged halo, no (pixel, tap) mask;
//   * operand roles swapped and filter rows permuted (c64_slot_filter) so that a lane's 16 accumulator rows are 16 consecutive filters of its position:
//     32 contiguous bytes per lane, stored straight from registers after a 2 x 2 exchange inside lane pairs (each store instruction then writes 32
//     contiguous bytes per lane pair, 64 with the other half-wave); batch-norm partial sums per lane over its position column, met across lanes once
//     after the last tile (same partial-row contract as conv_igemm.hip).
// W / (W + 1) x H / (H + 1) of the MFMA work is real (98 % at 104 x 104).
#include "common.h"
#include "conv_shared.h"
#include <atomic>

#define C64_TP 256                 // padded positions per tile (2 halves x 2 passes x 64)
#define C64_KSTEPS 36              // 9 taps x 4 sixteen-channel groups
#define C64_STAT_ROWS 32           // partial rows the statistics are spread over (the consumer's prologue sums every row in use: few rows, 8 workgroups each)

// MFMA output row rho lands in register r = (rho & 3) + 4 (rho >> 3) of the lanes of half h = (rho >> 2) & 1: giving that row filter 16 h + r makes
// the 16 values of a lane 16 CONSECUTIVE filters (32 contiguous bytes of its position's row)
__device__ __forceinline__ int c64_slot_filter(int rho) { return 16 * ((rho >> 2) & 1) + (rho & 3) + 4 * (rho >> 3); }
__device__ __forceinline__ unsigned c64_div(unsigned q, unsigned m, unsigned s) { return __umulhi(q, m) >> s; }

// MODE 0: plain store; 1: + batch-norm partial sums of the stored values (training forward); 2: + bias, leaky_relu (inference, folded BN)
template <int MODE>
__global__ __launch_bounds__(512) void conv_c64_fwd_kernel(
    const bf16 *__restrict__ X, unsigned x_bytes, const bf16 *__restrict__ F, bf16 *__restrict__ O, int H, int W, int M, int Mp,
    int ntiles, int HR, const float *__restrict__ vec, float *__restrict__ bn_part, float alpha, unsigned mP, unsigned sP, unsigned mH, unsigned sH) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int HB = HR * 128;                               // bytes per halo buffer
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fg = wave & 3, ph = wave >> 2;               // filter group (32 filters), position half of the tile
    const int half = lane >> 5, l31 = lane & 31;
    const unsigned P1 = (unsigned)(W + 1), H1 = (unsigned)(H + 1);
    const int HL = W + 2;                                  // halo rows in front of a tile: the farthest tap is (W + 1) + 1 positions back

    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16 *>(X), 0, x_bytes, 0x00020000);
    typedef __attribute__((address_space(3))) void *lds_void_ptr;
    typedef const __attribute__((address_space(3))) bf16x8 *lds_frag_ptr;

    // ---- the filter, once, into registers: fragment st = (tap, 16-channel group kk) of MFMA row l31 = filter 32 fg + c64_slot_filter(l31), k = 8 half .. + 7
    // (Ffwd rows are [tap][channel] for a 64-channel layer: common.h y2_filter_koff).  Read straight from global memory a lane would touch 36 separate
    // 16-byte pieces 1152 bytes apart from its neighbour's -- measured ~9 us per workgroup, a third of the launch at batch 16 -- so the rows go through LDS:
    // two rounds of 64 filters, each staged by LDS-DMA (coalesced at the source) into the space behind halo buffer 0 with a row pitch of 1168 bytes
    // (292 dwords = 36 mod 64: the 16 rows of a ds_read_b128 lane group hit 16 different bank quads; the 16 padding bytes of a row read as zeros), then
    // picked up by the four waves that hold those filter groups.  Tile 0's halo streams into buffer 0 meanwhile (issued below, before this block's waits).
    bf16x8 filt[C64_KSTEPS];
    auto load_filter = [&]() {
        constexpr int FPITCH = 1168, FROUND = 64 * FPITCH;          // 74752 bytes per round (73 pieces of 1 KiB)
        const __amdgpu_buffer_rsrc_t rsrcF = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16 *>(F), 0, 128u * 1152u, 0x00020000);
        unsigned char *const fst = smem + HB;                      // behind halo buffer 0: buffer 1 and the tail of the allocation are idle until tile 1 is staged
#pragma unroll 1
        for (int rd = 0; rd < 2; ++rd) {
            for (int p = wave; p * 1024 < FROUND; p += 8) {
                const int pos = p * 1024 + lane * 16;
                const int n = pos / FPITCH, off = pos - n * FPITCH;
                const unsigned voff = (off < 1152 && n < 64) ? (unsigned)((64 * rd + n) * 1152 + off) : Y2_OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcF, (lds_void_ptr)(fst + p * 1024), 16, voff, 0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                       // the round's 64 rows are in LDS
            if ((fg >> 1) == rd) {
                const unsigned a0 = y2_lds_addr(fst) + (unsigned)((32 * (fg & 1) + c64_slot_filter(l31)) * FPITCH + 16 * half);
#pragma unroll
                for (int st = 0; st < C64_KSTEPS; ++st) filt[st] = *(lds_frag_ptr)(uintptr_t)(a0 + (unsigned)(st * 32));
            }
            __syncthreads();                                       // ... and read: the next round (or tile 1's halo) may overwrite them
        }
    };

    // ---- halo staging: piece p of a buffer = rows 8 p .. 8 p + 7 (128 bytes each); lane = (row = lane >> 3, 16-byte chunk = lane & 7), source chunk swizzled
    // with (row >> 1) & 7 so that the 16 rows a ds_read_b128 lane group touches land on 16 different bank quads.  A staged row is padded position
    // q0 - HL + row: a pixel, or zeros (zero column, zero row, outside the batch).
    const int npieces = HR >> 3;
    const int prow = lane >> 3;
    auto stage_piece = [&](int q0, int buf, int p) {
        const int row = p * 8 + prow;
        const unsigned q = (unsigned)(q0 + row);                              // wraps below 0: fails the range test
        const unsigned R = c64_div(q, mP, sP);                                // padded image row over the whole batch
        const unsigned img = c64_div(R, mH, sH);
        const unsigned c = q - __umul24(R, P1), r = R - __umul24(img, H1);
        const bool ok = (q < (unsigned)Mp) & (c < (unsigned)W) & (r < (unsigned)H);
        const unsigned m = q - R - __umul24(img, (unsigned)W);                // pixel index: q = R (W + 1) + c, R = img (H + 1) + r, m = (img H + r) W + c
        const unsigned voff = ok ? m * 128u + (unsigned)((((lane & 7) ^ ((row >> 1) & 7))) << 4) : Y2_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (lds_void_ptr)(smem + buf * HB + p * 1024), 16, voff, 0, 0, 0);
    };
    auto stage = [&](int tile, int buf) {
        const int q0 = tile * C64_TP - HL;
        for (int p = wave; p < npieces; p += 8) stage_piece(q0, buf, p);
    };
    int tile = blockIdx.x;
    if (tile < ntiles) stage(tile, 0);
    load_filter();                                                   // (its waits also cover this wave's pieces of the first tile)

    // ---- read addresses (buffer 0, pass 0, position block 0): one per tap; the 16-channel group kk is XORed in (bits 5-6), the lane's half in bit 4.
    // Position block i of pass pa is 32 (2 pa + i) rows = (2 pa + i) * 4096 bytes on: the swizzle term ((row >> 1) & 7) is the same 32 rows on.
    const unsigned lds0 = y2_lds_addr(smem);
    unsigned aaddr[9];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
        const int shift = (tp / 3 - 1) * (W + 1) + (tp % 3 - 1);
        const int row = ph * 128 + l31 + HL + shift;
        aaddr[tp] = lds0 + (unsigned)(row * 128 + ((half ^ ((row >> 1) & 7)) << 4));
    }

    // per-filter constants of this lane's 16 accumulator rows (filter 32 fg + 16 half + r): MODE 2 the bias, MODE 1 the statistics' shift (moving mean)
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    // (MODE 1 keeps the shift in LDS behind the halo buffers and reads it per epilogue: its 32 sum registers leave no room for 16 more beside the filter)
    // and the per-filter sums in LDS as well (stat[0][f] = sum d, stat[1][f] = sum d^2 of this workgroup): a pass reduces its lanes' 32 partial sums across
    // the 32 positions of a half-wave by a halving exchange -- after five steps lane j holds the total of value j -- and adds them with one LDS atomic per lane
    f32x2 cst[8];                                            // packed pairs: the epilogue runs on v_pk_add_f32 / v_pk_fma_f32
    float *const shl = reinterpret_cast<float *>(smem + 2 * HB);
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

    const int T = tile < ntiles ? (ntiles - tile + (int)gridDim.x - 1) / (int)gridDim.x : 0;      // tiles of this workgroup
    // One RAW barrier per tile (__syncthreads() would also wait for the output stores in flight): every wave's pieces of tile t have landed (each waited
    // for its own at the end of its previous tile) and every wave has left the other buffer.
    for (int t = 0; t < T; ++t) {
        __builtin_amdgcn_s_barrier();
        tile = (int)blockIdx.x + t * (int)gridDim.x;
        const int buf = t & 1;
        if (t + 1 < T) stage(tile + (int)gridDim.x, buf ^ 1);      // this wave's pieces of the next tile, into the buffer every wave left before the barrier
        const unsigned boff = buf ? (unsigned)HB : 0u;
#pragma unroll 1
        for (int pa = 0; pa < 2; ++pa) {
            f32x16 acc[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            const unsigned poff = boff + (unsigned)(pa * 8192);
            // 36 steps (tap, 16-channel group) of 2 fragment reads + 2 MFMAs; the reads of step s + 1 are issued ahead of the MFMAs of step s (two fragment sets)
            bf16x8 fb[2][2];
            auto load = [&](int st, int set) {
                const int tp = st >> 2, kk = st & 3;
                const unsigned pa_ = (aaddr[tp] ^ (unsigned)(kk * 32)) + poff;
#pragma unroll
                for (int i = 0; i < 2; ++i) fb[set][i] = *(lds_frag_ptr)(uintptr_t)(pa_ + (unsigned)(i * 4096));
            };
            load(0, 0);
#pragma unroll
            for (int st = 0; st < C64_KSTEPS; ++st) {
                if (st + 1 < C64_KSTEPS) load(st + 1, (st + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);       // (pins the order: the scheduler otherwise sinks every read to just above its MFMA)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(filt[st], fb[st & 1][i], acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- epilogue of the pass: D[filter][position].  Round (MODE 2: bias + leaky first), statistics of the rounded values, pack pairs: a lane holds
            // filters 32 fg + 16 half + [0, 16) of its position = two 16-byte chunks.
#if defined(__HIP_DEVICE_COMPILE__)
            if (pa == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the next tile's pieces (issued in front of this tile's MFMAs) and the previous stores: nothing waits for a store it has just issued
            const bool odd = (lane & 1) != 0;
            // (1) round both position blocks to packed bf16 pairs (MODE 2: bias + leaky first): the accumulators are dead after this
            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
            unsigned pk[2][8];                                                  // [position block][filter pair]: two 16-byte chunks per block
            float lfi[2];
            unsigned moffi[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const unsigned q = (unsigned)(tile * C64_TP + ph * 128 + pa * 64 + i * 32 + l31);
                const unsigned R = c64_div(q, mP, sP);
                const unsigned c = q - __umul24(R, P1);
                const unsigned img = c64_div(R, mH, sH);
                const unsigned r_ = R - __umul24(img, H1);
                const bool live = (q < (unsigned)Mp) & (c < (unsigned)W) & (r_ < (unsigned)H);
                const unsigned m = __umul24(__umul24(img, (unsigned)H) + r_, (unsigned)W) + c;
                moffi[i] = live ? m * 256u + (unsigned)(fg * 64 + half * 32) : 0xffffffffu;      // byte offset of this lane's 32 bytes (Mp < 2^23)
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
            for (int i = 0; i < 2; ++i) {
                unsigned s0[4], s1[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const unsigned x = pk[i][d], y = pk[i][4 + d];
                    const unsigned xs = (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true), ys = (unsigned)__builtin_amdgcn_mov_dpp((int)y, 0xB1, 0xF, 0xF, true);
                    s0[d] = odd ? ys : x;
                    s1[d] = odd ? y : xs;
                }
                const unsigned mo_even = (unsigned)__builtin_amdgcn_mov_dpp((int)moffi[i], 0xA0, 0xF, 0xF, true);      // quad_perm [0,0,2,2]: the pair's even lane
                const unsigned mo_odd = (unsigned)__builtin_amdgcn_mov_dpp((int)moffi[i], 0xF5, 0xF, 0xF, true);       // quad_perm [1,1,3,3]: the pair's odd lane
                const unsigned lo16 = odd ? 16u : 0u;
                if (mo_even != 0xffffffffu)
                    *reinterpret_cast<u32x4 *>(reinterpret_cast<unsigned char *>(O) + mo_even + lo16) = u32x4{s0[0], s0[1], s0[2], s0[3]};
                if (mo_odd != 0xffffffffu)
                    *reinterpret_cast<u32x4 *>(reinterpret_cast<unsigned char *>(O) + mo_odd + lo16) = u32x4{s1[0], s1[1], s1[2], s1[3]};
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
                    for (int i = 0; i < 2; ++i) {
                        const f32x2 lf2 = {lfi[i], lfi[i]};
                        const f32x2 y2 = {__builtin_bit_cast(float, pk[i][qd] << 16), __builtin_bit_cast(float, pk[i][qd] & 0xffff0000u)};
                        const f32x2 d2 = (y2 - svq) * lf2;
                        a1 += d2;
                        a2 = __builtin_elementwise_fma(d2, d2, a2);
                    }
                    // lanes differing in bit 0: the even lane keeps the sums, the odd lane the squares
                    float k0, k1;
                    {
                        const float s0_ = odd ? a1[0] : a2[0], s1_ = odd ? a1[1] : a2[1];
                        const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s0_), 0xB1, 0xF, 0xF, true));
                        const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s1_), 0xB1, 0xF, 0xF, true));
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

bool y2_c64_shape(int Cp, int ldp, int Nf, int ldo, int ksize, int dtype) { return dtype == YOLO2_BF16 && ksize == 3 && Cp == 64 && ldp == 64 && Nf == 128 && ldo == 128; }

// -> 0 launched (*rows = partial rows touched when bn_part), 1 the shape does not fit this kernel's LDS plan (caller takes the generic kernels)
int y2_c64_fwd(const void *P, const void *F, void *O, int B, int H, int W, const float *bias, float alpha, const float *bn_shift, float *bn_part,
               int cus, int *rows, hipStream_t st) {
    const long Mp = (long)B * (H + 1) * (W + 1);
    const int HR = (C64_TP + 2 * (W + 2) + 7) / 8 * 8;
    size_t lds = 2 * (size_t)HR * 128 + 512;      // two halo buffers, 128 shifts
    if (lds < (size_t)HR * 128 + 74752) lds = (size_t)HR * 128 + 74752;      // ... and room for a filter round behind halo buffer 0
    if (lds > 160 * 1024 || Mp + C64_TP >= (1L << 23) || H < 1 || W < 1) return 1;
    const int M = B * H * W, ntiles = (int)((Mp + C64_TP - 1) / C64_TP);
    const int grid = ntiles < cus ? ntiles : cus;
    unsigned mP, sP, mH, sH;
    y2_magic_u32((unsigned)(W + 1), &mP, &sP);
    y2_magic_u32((unsigned)(H + 1), &mH, &sH);
    const unsigned x_bytes = (unsigned)((size_t)M * 64 * 2);
    static std::atomic<size_t> lds_set[3][64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
#define C64_LAUNCH(MODEv, vecp)                                                                                                              \
    do {                                                                                                                                     \
        if (lds > lds_set[MODEv][dev].load(std::memory_order_relaxed)) {                                                                     \
            if (hipFuncSetAttribute((const void *)conv_c64_fwd_kernel<MODEv>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 1; \
            lds_set[MODEv][dev].store(lds, std::memory_order_relaxed);                                                                       \
        }                                                                                                                                    \
        conv_c64_fwd_kernel<MODEv><<<grid, 512, lds, st>>>((const bf16 *)P, x_bytes, (const bf16 *)F, (bf16 *)O, H, W, M, (int)Mp, ntiles, HR, \
                                                          vecp, bn_part, alpha, mP, sP, mH, sH);                                                  \
    } while (0)
    if (bn_part) C64_LAUNCH(1, bn_shift);
    else if (bias || alpha != 1.0f) C64_LAUNCH(2, bias);      // (an activation without a bias: the constants read as zeros)
    else C64_LAUNCH(0, (const float *)nullptr);
#undef C64_LAUNCH
    if (rows) *rows = grid < C64_STAT_ROWS ? grid : C64_STAT_ROWS;
    return 0;
}
