// 3x3 convolution FORWARD for 32-channel inputs and 64 filters (Darknet-19 conv1: 208 x 208 pixels per image), bf16, gfx950.
//
// Replaces slim.layers.conv2d of reference model/yolo2/inference.py:75-76 (conv1: 32 -> 64, 3x3, SAME, no bias under batch_norm) where the
// generic implicit-GEMM kernel is at its worst: the reduction is only 9 taps x 32 channels = 288 long, so a 128 x 64 tile of the per-tap
// kernel runs nine K steps of TWO MFMAs per wave between barriers and pays a prologue and an epilogue per 18 MFMAs -- 74 us for 25.5 GFLOP
// (344 TFLOP/s) although the layer moves only 133 MB.  What this kernel does instead:
//   * persistent workgroups (one per CU) walk 512-position tiles; the whole filter (64 x 288 bf16, 37 KB) is loaded into LDS ONCE per workgroup;
//   * the pixels run over a PADDED index q (row pitch W + 1, image pitch H + 1: one zero column per row, one zero row per image -- the
//     filter-gradient kernel's idea, conv_wgrad3.hip): all nine taps are constant row offsets dh (W + 1) + dw into ONE staged halo image with no
//     (pixel, tap) mask anywhere; the DMA source offset of a staged row decides whether it is a pixel or a zero;
//   * a wave computes 64 positions x 64 filters (72 MFMAs per tile on 72 + 72 ds_read_b128), no barrier inside a tile, the next tile's halo
//     streams into the other buffer meanwhile;
//   * the MFMA operand roles are swapped (D = F^T X^T: column = lane & 31 = POSITION, rows = filters) and the filter rows are permuted
//     (c32_slot_filter) so that a lane's 32 accumulator rows are 32 consecutive filters of its position; a 4 x 4 transpose over the lanes of a
//     quad (DPP) then makes every store instruction write whole 128-byte rows: straight from registers, no LDS staging of the output tile.
//     The batch-norm partial sums (same contract as conv_igemm.hip's epilogue) run per lane over its position column and meet across lanes
//     once, after the last tile.
// W / (W + 1) x H / (H + 1) of the MFMA work is real (99 % at 208 x 208).
//
// Measured (batch 16, 208 x 208, MODE 1): 55 us against 70-74 us of the generic kernel.  The kernel is bound by what it moves between L2 and the
// CUs, not by its 72 MFMAs per wave and tile: 1365 tiles x 60 KB of halo (the 420 halo rows of a 512-position tile re-stage 82 % of it) + 89 MB
// of output = 171 MB at the ~4.7 TB/s the DMA issue and the stores sustain together (s_memtime stamps per phase: a wave's eight DMA pieces
// take ~1.7k ticks to ISSUE and its eight stores ~1k, against ~12 ticks per piece from cache: scripts/experiments/lds_dma_issue.hip) -- 36 us if
// everything overlapped.  Tried on top and measured no better: two wave groups half a tile apart (MFMA loop of one over the epilogue of the
// other), DMA pieces interleaved into the MFMA steps, half the workgroups delayed by half a tile.  What did pay: whole-row stores (the quad
// transpose: store issue 1.8k -> 1.0k ticks per tile), packed-f32 branch-free statistics (epilogue 700 -> 380 instructions).
#include "common.h"
#include "conv_shared.h"
#include <atomic>

#define C32_TP 512                 // padded positions per tile (8 waves x 64)
#define C32_FROWB 592              // LDS bytes per filter row: 576 + 16 (row stride 148 banks: the 16 lanes of a read group hit 16 different bank quads)
#define C32_FBYTES (64 * C32_FROWB)

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

template <int MODE>
__global__ __launch_bounds__(512) void conv_c32_fwd_kernel(
    const bf16 *__restrict__ X, unsigned x_bytes, const bf16 *__restrict__ F, unsigned f_bytes, bf16 *__restrict__ O, int H, int W, int M, int Mp,
    int ntiles, int HR, const float *__restrict__ vec, float *__restrict__ bn_part, float alpha, unsigned mP, unsigned sP, unsigned mH, unsigned sH, int abl) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char *const fl = smem;                        // filter image [64][C32_FROWB]
    unsigned char *const hb = smem + C32_FBYTES;           // two halo buffers of HR rows x 64 bytes
    const int HB = HR * 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const unsigned P1 = (unsigned)(W + 1), H1 = (unsigned)(H + 1);
    const int HL = W + 2;                                  // halo rows in front of a tile: the farthest tap is (W + 1) + 1 positions back

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

    // ---- halo staging: piece p of a buffer = rows 16 p .. 16 p + 15 (64 bytes each); lane = (row = lane >> 2, 16-byte chunk = lane & 3), source chunk
    // swizzled with (row >> 2) & 3 so that the 16 rows a ds_read_b128 lane group touches -- whatever row a tap offset starts them at -- land on 16
    // different bank quads.  A staged row is padded position q0 - HL + row: a pixel, or zeros (zero column, zero row, outside the batch).
    const int npieces = HR >> 4;
    const int prow = lane >> 2;
    const unsigned pchunk = (unsigned)(((lane & 3) ^ ((lane >> 4) & 3)) * 16);
    // one piece: q -> (R = q / (W + 1), img = R / (H + 1)); the pixel index is m = q - R - img W (q = R (W + 1) + c, R = img (H + 1) + r, m = (img H + r) W + c)
    auto stage_piece = [&](int q0, int buf, int p) {
        const unsigned q = (unsigned)(q0 + p * 16 + prow);                     // wraps below 0: fails the range test
        const unsigned R = c32_div(q, mP, sP);                                 // padded image row over the whole batch
        const unsigned img = c32_div(R, mH, sH);
        const unsigned c = q - __umul24(R, P1), r = R - __umul24(img, H1);
        const bool ok = (q < (unsigned)Mp) & (c < (unsigned)W) & (r < (unsigned)H);
        const unsigned m = q - R - __umul24(img, (unsigned)W);
        const unsigned voff = ok ? m * 64u + pchunk : Y2_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (lds_void_ptr)(hb + buf * HB + p * 1024), 16, voff, 0, 0, 0);
    };
    auto stage = [&](int tile, int buf, int p0, int pstep) {
        const int q0 = tile * C32_TP - HL;
        for (int p = p0; p < npieces; p += pstep) stage_piece(q0, buf, p);
    };
    int tile = blockIdx.x;
    if (tile < ntiles) stage(tile, 0, wave, 8);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the filter and this wave's pieces of the first tile have landed

    // ---- read addresses (buffer 0; the other buffer is + HB).  A (pixels, the MFMA's B operand): one per (tap, position block); the two 16-channel
    // groups of a tap differ by XOR 32 (bit 1 of the chunk index).  B (filters, the MFMA's A operand): row l31 (+ 32 j), 8 k-values of this lane's half.
    const unsigned lds0 = y2_lds_addr(smem);
    unsigned aaddr[9];                                     // position block 0; block 1 is 32 rows = + 2048 bytes on (same swizzle: (row >> 2) & 3)
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
        const int shift = (tp / 3 - 1) * (W + 1) + (tp % 3 - 1);
        const int row = wave * 64 + l31 + HL + shift;
        aaddr[tp] = lds0 + (unsigned)(C32_FBYTES + row * 64 + (((half) ^ ((row >> 2) & 3)) << 4));
    }
    const unsigned baddr = lds0 + (unsigned)(l31 * C32_FROWB + half * 16);       // filter block 0; block 1 is 32 rows on

    // per-filter constants of this lane's 2 x 16 accumulator rows: filter f(j, r) = 32 half + 16 j + r (c32_slot_filter).  MODE 2 keeps the bias in
    // registers; MODE 1 keeps the statistics' shift (moving mean) in LDS behind the halo buffers and reads it per epilogue (its 64 sum registers
    // leave no room for 32 more)
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    float cst[2][16];
    f32x2 t1[2][8], t2[2][8];                               // packed pairs: the sums run on v_pk_add_f32 / v_pk_fma_f32
    float *const shl = reinterpret_cast<float *>(hb + 2 * HB);
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
    const int T = tile < ntiles ? (ntiles - tile + (int)gridDim.x - 1) / (int)gridDim.x : 0;      // tiles of this workgroup
    f32x16 acc[2][2];                                                // [filter block j][position block i]
#ifdef Y2C32_EXPERIMENTS
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl = __builtin_readcyclecounter();
    const unsigned long long t00 = tl;
#endif
    auto mfma_phase = [&](int t) {
        tile = (int)blockIdx.x + t * (int)gridDim.x;
        const int buf = t & 1;
        if (t + 1 < T && !C32_ABL(4)) stage(tile + (int)gridDim.x, buf ^ 1, wave, 8);       // this wave's pieces of the next tile, into the buffer every wave left before the barrier
        C32_STAMP(1);
        const unsigned boff = buf ? (unsigned)HB : 0u;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
        // 18 steps (tap, 16-channel group) of 4 fragment reads + 4 MFMAs; the reads of step s + 1 are issued ahead of the MFMAs of step s (two
        // fragment sets: left to itself hipcc keeps one and waits lgkmcnt(0) in front of nearly every MFMA)
        bf16x8 fa[2][2], fb[2][2];
        auto load = [&](int st, int set) {
            const int tp = st >> 1, kk = st & 1;
            const unsigned pa = (aaddr[tp] ^ (unsigned)(kk * 32)) + boff;
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[set][i] = *(lds_frag_ptr)(uintptr_t)(pa + (unsigned)(i * 2048));
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
        tile = (int)blockIdx.x + t * (int)gridDim.x;
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
            const unsigned q = (unsigned)(tile * C32_TP + wave * 64 + i * 32 + l31);
            const unsigned R = c32_div(q, mP, sP);
            const unsigned c = q - __umul24(R, P1);
            const unsigned img = c32_div(R, mH, sH);
            const unsigned r_ = R - __umul24(img, H1);
            const bool live = (q < (unsigned)Mp) & (c < (unsigned)W) & (r_ < (unsigned)H);
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
    // (each waited for its own in its previous epilogue) and every wave has left the other buffer.
    for (int t = 0; t < T; ++t) {
        __builtin_amdgcn_s_barrier();
        C32_STAMP(0);
        mfma_phase(t);
        epilogue_phase(t);
    }
    if (MODE == 1 && bn_part) {
        // a lane holds the sums of its 32 filters over its position column: they meet across the 32 lanes of each half.  Step-major -- all 64
        // exchanges of a step in flight together, then the adds: value-major (64 chains of 5 dependent ds_bpermute round trips, a wait behind each)
        // cost 21 us per workgroup, a third of the kernel
        float v1[32], v2[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) { v1[k] = t1[k >> 4][(k & 15) >> 1][k & 1]; v2[k] = t2[k >> 4][(k & 15) >> 1][k & 1]; }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
            float x1[32], x2[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) { x1[k] = __shfl_xor(v1[k], o, 64); x2[k] = __shfl_xor(v2[k], o, 64); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 32; ++k) { v1[k] += x1[k]; v2[k] += x2[k]; }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (l31 == 0) {
            const int slot = (int)(blockIdx.x & (Y2_BN_PART_ROWS - 1));
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const int j = k >> 4, r = k & 15;
                const int f = 32 * half + 16 * j + r;
                unsafeAtomicAdd(bn_part + slot * 64 + f, v1[k]);
                unsafeAtomicAdd(bn_part + (Y2_BN_PART_ROWS + slot) * 64 + f, v2[k]);
            }
        }
    }
}

#ifdef Y2C32_EXPERIMENTS
extern "C" int yolo2_debug_c32_stamps(void *host_out) { return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(c32_stamps), sizeof(unsigned long long) * 256 * 8); }
#endif

bool y2_c32_shape(int Cp, int ldp, int Nf, int ldo, int ksize, int dtype) { return dtype == YOLO2_BF16 && ksize == 3 && Cp == 32 && ldp == 32 && Nf == 64 && ldo == 64; }

// -> 0 launched (*rows = partial rows touched when bn_part), 1 the shape does not fit this kernel's LDS plan (caller takes the generic kernels)
int y2_c32_fwd(const void *P, const void *F, void *O, int B, int H, int W, const float *bias, float alpha, const float *bn_shift, float *bn_part,
               int cus, int *rows, hipStream_t st) {
    const long Mp = (long)B * (H + 1) * (W + 1);
    const int HR = (C32_TP + 2 * (W + 2) + 15) / 16 * 16;
    const size_t lds = (size_t)C32_FBYTES + 2 * (size_t)HR * 64 + 256;      // filter image, two halo buffers, 64 shifts
    if (lds > 160 * 1024 || Mp + C32_TP >= (1L << 24) || H < 1 || W < 1) return 1;
    const int M = B * H * W, ntiles = (int)((Mp + C32_TP - 1) / C32_TP);
    const int grid = ntiles < cus ? ntiles : cus;
#ifdef Y2C32_EXPERIMENTS
    static const int abl = y2_env_int("YOLO2_C32_ABL", 0);      // timing ablations (wrong results): 1 no MFMA loop, 2 no stores, 4 no DMA after the first tile, 8 phase stamps
#else
    const int abl = 0;
#endif
    unsigned mP, sP, mH, sH;
    y2_magic_u32((unsigned)(W + 1), &mP, &sP);
    y2_magic_u32((unsigned)(H + 1), &mH, &sH);
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
        conv_c32_fwd_kernel<MODEv><<<grid, 512, lds, st>>>((const bf16 *)P, x_bytes, (const bf16 *)F, f_bytes, (bf16 *)O, H, W, M, (int)Mp, ntiles, HR, \
                                                          vecp, bn_part, alpha, mP, sP, mH, sH, abl);                                             \
    } while (0)
    if (bn_part) C32_LAUNCH(1, bn_shift);
    else if (bias || alpha != 1.0f) C32_LAUNCH(2, bias);      // (an activation without a bias: the constants read as zeros)
    else C32_LAUNCH(0, (const float *)nullptr);
#undef C32_LAUNCH
    if (rows) *rows = grid < Y2_BN_PART_ROWS ? grid : Y2_BN_PART_ROWS;
    return 0;
}
