// NHWC implicit-GEMM convolution on MFMA (gfx950), forward and data-gradient.
//
// Replaces slim.layers.conv2d (reference model/yolo2/inference.py:37-48,73-118) and its
// tf.gradients input-gradient (train.py:127-129).
//
//   O[m, n] = bias[n] + sum_{tap, c} P[pix(m) + shift(tap), c] * F[n][tap*Cp + c]
//   GEMM view: M = B*H*W output pixels, N = Nf filters, K = ksize^2 * Cp.
//
//   K order of a filter row: 64-channel chunk, tap, channel (common.h y2_filter_koff): the 9 shifted re-reads of a
//   pixel-tile x 64-channel slab are consecutive K steps and hit in the XCD's L2.
//
// Structure (one workgroup = NW waves = a BM(M) x BN(N) output tile; a K step = CH 16-byte chunks of one tap's channels):
//   * Operands go HBM/L2 -> LDS by DMA: buffer_load_dwordx4 ... lds (1 KiB per wave-instruction, no
//     VGPR round trip, no ds_write pass).  The DMA writes LDS lane-linearly (wave-uniform base +
//     lane*16 B), so rows cannot be padded; the b128 fragment reads are kept bank-conflict-free by an
//     XOR swizzle applied on the SOURCE side: the lane that owns LDS position (row, slot) fetches global
//     chunk  slot ^ ((row / rows_per_256B) % CH); the fragment reads apply the same involution
//     (SQ_LDS_BANK_CONFLICT = 0 measured).
//   * SAME padding, image borders, the M tail, filter-row tail and channel tail are all realised by the
//     buffer descriptor's hardware range check: a lane that must contribute zeros uses an out-of-range
//     offset and the DMA writes 0 -- no branches, no pointer selects, 32-bit offsets only.  Which of the
//     9 taps are inside the image for a pixel row is a 9-bit mask (outer product of a row and a column 3-bit set)
//     computed once per lane per tile, with reciprocal-multiply pixel decoding (no hardware integer divider).
//     Instruction overhead per MFMA decided the short-reduction layers twice: the first version spent 11 VALU + 12 SALU
//     per MFMA on per-piece address arithmetic (17 % MFMA utilisation); later the tile prologue's divisions and the
//     per-element row test of the epilogue cost conv1..conv8 15-35 %.
//   * NSTAGE-deep LDS ring: the DMA of tile t+NSTAGE-1 is issued right after the barrier of tile t and
//     only tile t's own pieces are waited for (counted s_waitcnt vmcnt(N) + raw s_barrier; a
//     __syncthreads() would drain every DMA in flight).
//   * A = pixels (rows), B = filters (cols): the 32x32 accumulator layout puts 32 consecutive output
//     channels of one pixel in 32 consecutive lanes -> 64 B (bf16) / 128 B (f32) store segments.
//   * Shapes (chosen per layer from measurements, launch_conv below): 128x128 tile, 8 waves, 128-byte K rows on a 2-stage
//     ring (64 KiB: two workgroups per CU, 8 MFMAs per wave between barriers) for full grids; 64-byte rows / 3 stages
//     when the channel count is not a multiple of 64; 64- and 32-filter tiles for the narrow layers.
//   * Under-filled grids (13x13 / 26x26 stages at batch 16: 88-176 tiles for 256 CUs) run stream-K: one workgroup per CU,
//     equal contiguous shares of the flat (tile, K step) space, partial tiles parked in the workspace and fixed up by
//     the tile's owner through sc1 (agent-coherent, fence-free) accesses -- see the SPLITK == 2 epilogue; long
//     reductions take a 256x128 tile (8 waves of 64x64, 16 MFMAs per barrier).  Grids <= 128 tiles with a medium
//     reduction slice the K loop over gridDim.y with f32 atomics + a finishing kernel (SPLITK == 1).
//   * The epilogue can also emit the batch-norm partial sums of the stored outputs (yolo2_conv2d_bn), apply
//     bias + leaky ReLU (yolo2_conv2d_bias_leaky, BN-folded inference), and -- for a data gradient whose output is the gradient of a
//     batch-normalised producer layer -- reduce that layer's dgamma / dbeta sums from the tile image (yolo2_conv2d_dgrad_bn).
//   * 3x3 layers with >= 1024 input channels on images up to 55 wide take conv3x3_tap_kernel (further down): one halo image per
//     64-channel chunk serves all nine taps, software-pipelined K loop.  Its stream-K tiles walk their 64-channel chunks in a
//     ROTATED order so that the ~32 workgroups of an XCD read the same bytes of a filter slab (7 MB > the 4 MB L2) at the same
//     time: 3.6x less fabric traffic on the 3072-channel layer (profiles/r03_l2_stationary_ab.md; the time moved 1.8 %: issue-bound).
//   * Statistics rows: a producer leaves one partial row per (pixel tile, wave row) while that is no more than the consumer's
//     prologue reads (y2_stat_rows_limit: 128 rows for >= 128 bf16 channels), else it wraps around 16-128 rows with f32 atomics; the
//     kernel that consumes the moments finishes them (elementwise.hip *_fin kernels) -- no finalisation launch.
//   * blockIdx -> tile: filter tile fastest (the blocks of XCD b%8 keep one filter slab in their L2), or,
//     when the filter operand is small, one contiguous run of M tiles per XCD (halo rows shared in L2).
#include "common.h"
#include "conv_shared.h"
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include <type_traits>

template <typename T> struct Mma;
template <> struct Mma<bf16> {
    static constexpr int KSTEP = 16;
    typedef bf16x8 Frag;
    static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    static constexpr int KSTEP = 2;
    typedef float Frag;
    static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
};

// (Timing-ablation hooks -- no epilogue / no MFMA / no fragment reads / no DMA builds -- live in scripts/experiments/conv_igemm_ablation_hooks.patch,
// applied to a COPY of this file by scripts/abl_build.sh; the product source carries none.)
// (Y2_STREAM_FLAG_WORDS, conv_shared.h: one stream-K flag word per workgroup in a library-owned pool; the workspace holds one f32 tile slot each)
// SPLITK: 0 = one workgroup per output tile; 1 = K loop sliced over gridDim.y; 2 = stream-K: gridDim.x workgroups (one per
// CU) share the flat (tile, K step) space in equal contiguous ranges.  1 and 2 accumulate f32 partial tiles with atomics.
template <typename T, int BN, int WGN, int NSTAGE, int KS, int SPLITK, bool CTAIL, int CH = 4, int NW = 4, int BMv = 128, bool BNBWD = false>
__global__ __launch_bounds__(NW * 64) void conv_igemm_kernel(
    const T *__restrict__ P, unsigned p_bytes, const T *__restrict__ F, unsigned f_bytes, const float *__restrict__ bias,
    T *__restrict__ O, float *__restrict__ Oacc, int H, int W, int Cp, int ldp, int Nf, int ldo, int M, int NT, int remap,
    const float *__restrict__ bn_shift, float *__restrict__ bn_part, unsigned *__restrict__ sk_flags, float act_alpha, int wide_store,
    const Y2BnBwd bz) {
    constexpr int BM = BMv;                // pixels per tile: 128, or 256 (8 waves of 64 x 64)
    constexpr int TAPS = KS * KS;
    constexpr int VEC = 16 / sizeof(T);
    constexpr int BK = CH * VEC;           // CH = 16-byte chunks per tile row: 4 -> 32 bf16 / 16 f32 per K step
    constexpr int ROWB = CH * 16;          // bytes per tile row
    constexpr int RPL = 256 / ROWB;        // rows per 256-byte LDS bank line
    constexpr int RPI = 64 / CH;           // rows per DMA instruction (1 KiB)
    constexpr int WGM = NW / WGN;          // NW waves per workgroup, arranged WGM x WGN over the output tile
    constexpr int TM = BM / WGM / 32, TN = BN / WGN / 32;
    constexpr int A_IT = BM / RPI / NW;    // DMA instructions per wave per tile
    constexpr int B_PIECES = BN / RPI;
    constexpr int B_IT = (B_PIECES + NW - 1) / NW;
    constexpr int LOADS = A_IT + B_IT;   // counted on vmcnt; identical in every wave (idle B slots still issue, all-OOB)
    constexpr int PAD = KS / 2;
    constexpr int STAGE = (BM + B_IT * NW * RPI) * ROWB;
    static_assert(A_IT >= 1, "at least one A piece per wave");
    static_assert(NSTAGE >= 2 && NSTAGE <= 4 && TM >= 1 && TN >= 1, "tile");

    __shared__ __attribute__((aligned(1024))) unsigned char smem[NSTAGE * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int MT = (M + BM - 1) / BM;
    // K order (common.h y2_filter_koff): channel chunk, tap, BK-wide step inside the chunk
    const int KC = y2_kchunk(Cp, TAPS);
    const int kpc = (KC + BK - 1) / BK;   // K tiles per (chunk, tap)
    const int nk = (Cp / KC) * TAPS * kpc;
    // stream-K: this workgroup's range of the flat (tile, K step) space; workgroup b runs on XCD b % 8 and the
    // ranges of one XCD are contiguous, tiles ordered filter-tile-major (an XCD works on ~NT/8 filter slabs)
    long su = 0, su_end = 0, su_total = 0;
    int wx = 0;
    if (SPLITK == 2) {
        su_total = (long)MT * NT * nk;
        const int G = gridDim.x, b = blockIdx.x;
        wx = (G & 7) ? b : (b & 7) * (G >> 3) + (b >> 3);
        su = wx * su_total / G;
        su_end = (wx + 1) * su_total / G;
    }

    const __amdgpu_buffer_rsrc_t rsrcP = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(P), 0, p_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcF = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(F), 0, f_bytes, 0x00020000);
    const int lrow = lane / CH, lslot = lane % CH;
    const double rcp_hw = 1.0 / (double)(H * W);
    const float rcp_w = 1.0f / (float)W;

  for (bool first_seg = true;; first_seg = false) {
    int nt, mt, kt_beg = 0, kt_end = nk;
    if (SPLITK == 2) {
        if (su >= su_end) break;
        const int t = (int)(su / nk);
        kt_beg = (int)(su - (long)t * nk);
        kt_end = (int)min((long)nk, kt_beg + (su_end - su));
        su += kt_end - kt_beg;
        nt = t / MT;
        mt = t - nt * MT;
        if (!first_seg) __syncthreads();      // every wave is done reading the previous segment's LDS stages
    } else {
        int tile = blockIdx.x;
        if (remap) {   // one contiguous run of tiles per XCD (bijective for any grid size)
            const int ntile = gridDim.x, xcd = tile & 7, idx = tile >> 3, q = ntile >> 3, r = ntile & 7;
            tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        }
        nt = tile % NT;
        mt = tile / NT;
        if (SPLITK == 1) {
            const int per = (nk + gridDim.y - 1) / gridDim.y;
            kt_beg = blockIdx.y * per;
            kt_end = min(nk, kt_beg + per);
            if (kt_beg >= kt_end) return;
        }
    }
    const int m0 = mt * BM, n0 = nt * BN;

    // per-lane source descriptors: byte offset of (row, swizzled chunk) and the set of taps inside the image
    unsigned a_voff[A_IT], a_mask[A_IT], a_cb[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int r = (wave * A_IT + i) * RPI + lrow;
        a_cb[i] = (unsigned)((lslot ^ ((r / RPL) % CH)) * 16);   // byte offset of this lane's chunk inside the K step
        const int m = m0 + r;
        unsigned mask = 0;
        if (m < M) {
            // (b, h, w) of pixel m without integer division (no hardware divider: ~25 instructions each): reciprocal estimate
            // + one correction step.  m < 2^31 needs the f64 estimate (error << 1); rem < H*W < 2^24 is exact in f32.
            const int HW = H * W;
            int bq = (int)((double)m * rcp_hw);
            int rem = m - bq * HW;
            if (rem < 0) rem += HW; else if (rem >= HW) rem -= HW;
            int h = (int)((float)rem * rcp_w);
            int w = rem - h * W;
            if (w < 0) { w += W; --h; } else if (w >= W) { w -= W; ++h; }
            if (KS == 3) {      // tap (r, s) is inside the image iff row h+r-1 and column w+s-1 are: outer product of two 3-bit sets
                const unsigned cm = (w > 0 ? 1u : 0u) | 2u | (w < W - 1 ? 4u : 0u);
                mask = (h > 0 ? cm : 0u) | (cm << 3) | (h < H - 1 ? cm << 6 : 0u);
            } else {
                mask = 1u;
            }
        }
        a_mask[i] = mask;
        a_voff[i] = (unsigned)m * (unsigned)ldp * (unsigned)sizeof(T) + a_cb[i];
    }
    unsigned b_voff[B_IT], b_cb[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int piece = wave * B_IT + i;
        const int r = piece * RPI + lrow;
        b_cb[i] = (unsigned)((lslot ^ ((r / RPL) % CH)) * 16);
        const int n = n0 + r;
        const bool ok = piece < B_PIECES && n < Nf;
        b_voff[i] = ok ? (unsigned)n * (unsigned)(KS * KS * Cp) * (unsigned)sizeof(T) + b_cb[i] : Y2_OOB;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // issue cursor (wave-uniform scalars), NSTAGE-1 tiles ahead of the compute cursor
    int kt_issue = kt_beg;
    int i_chunk0 = (kt_beg / (TAPS * kpc)) * KC;            // first channel of the chunk
    int i_tap = (kt_beg / kpc) % TAPS;
    int i_c0 = i_chunk0 + (kt_beg % kpc) * BK;              // first channel of the step
    int i_dh = i_tap / KS - PAD, i_dw = i_tap % KS - PAD;
    int i_stage = 0;
    const unsigned cp_bytes = (unsigned)Cp * (unsigned)sizeof(T);
    auto issue_next = [&]() {
        const unsigned tapbit = 1u << i_tap;
        const unsigned offA = (unsigned)((i_dh * W + i_dw) * ldp + i_c0) * (unsigned)sizeof(T);   // may wrap: added mod 2^32
        const unsigned offB = (unsigned)(i_chunk0 * TAPS + i_tap * KC + (i_c0 - i_chunk0)) * (unsigned)sizeof(T);
        const unsigned c0b = (unsigned)i_c0 * (unsigned)sizeof(T);
        unsigned char *As = smem + i_stage * STAGE;
        unsigned char *Bs = As + BM * ROWB;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            bool ok = (a_mask[i] & tapbit) != 0;
            if (CTAIL) ok = ok && (c0b + a_cb[i] < cp_bytes);
            const unsigned voff = ok ? a_voff[i] + offA : Y2_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcP, (__attribute__((address_space(3))) void *)(As + (wave * A_IT + i) * 1024), 16, voff, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            unsigned voff = b_voff[i] + offB;           // an OOB row stays out of range: OOB + offB < 2^32 and >= 2^31
            if (CTAIL) voff = (c0b + b_cb[i] < cp_bytes) ? voff : Y2_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcF, (__attribute__((address_space(3))) void *)(Bs + (wave * B_IT + i) * 1024), 16, voff, 0, 0, 0);
        }
        ++kt_issue;
        i_stage = (i_stage + 1 == NSTAGE) ? 0 : i_stage + 1;
        i_c0 += BK;
        if (i_c0 >= i_chunk0 + KC) {
            if (++i_tap == TAPS) { i_tap = 0; i_chunk0 += KC; i_dh = -PAD; i_dw = -PAD; }
            else if (++i_dw > PAD) { i_dw = -PAD; ++i_dh; }
            i_c0 = i_chunk0;
        }
    };
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
        if (kt_issue < kt_end) issue_next();

    // fragment read addressing: row i = lane&31 (+32 per MFMA tile); 32 rows = a whole number of swizzle periods
    const int frow = lane & 31;
    const int fsw = (frow / RPL) % CH;
    int c_stage = 0;
    for (int kt = kt_beg; kt < kt_end; ++kt) {
        // wait for tile kt only: up to NSTAGE-2 younger tiles stay in flight across the barrier
        const int ahead = min(NSTAGE - 2, kt_issue - 1 - kt);
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LOADS) : "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();      // tile kt complete in LDS for every wave; tile kt-1's buffer is free
        if (kt_issue < kt_end) issue_next();
        const unsigned char *As = smem + c_stage * STAGE + (wm * TM * 32 + frow) * ROWB;
        const unsigned char *Bs = smem + c_stage * STAGE + (BM + wn * TN * 32 + frow) * ROWB;
        c_stage = (c_stage + 1 == NSTAGE) ? 0 : c_stage + 1;
        // 128-byte rows, 8 waves of 32 x 64 (bf16): all twelve fragment reads of the K step are issued before its first MFMA (48 VGPRs)
        // and the MFMAs wait with a falling lgkmcnt; left to itself hipcc issues read, wait, MFMA, read, wait, MFMA ... and every
        // MFMA pair pays a full LDS latency (the tap-fused kernel below gained 15 % from the same change).
        constexpr bool HOIST = sizeof(T) == 2 && CH == 8 && NW == 8 && BM == 128;
        if constexpr (HOIST) {
            constexpr int KK = BK / Mma<T>::KSTEP;
            typename Mma<T>::Frag af[KK][TM], bf[KK][TN];
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                const int boff = ((kk * 2 + (lane >> 5)) ^ fsw) * 16;
#pragma unroll
                for (int i = 0; i < TM; ++i) af[kk][i] = *reinterpret_cast<const typename Mma<T>::Frag *>(As + i * 32 * ROWB + boff);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[kk][j] = *reinterpret_cast<const typename Mma<T>::Frag *>(Bs + j * 32 * ROWB + boff);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = Mma<T>::mma(af[kk][i], bf[kk][j], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
        } else
#pragma unroll
        for (int kk = 0; kk < BK / Mma<T>::KSTEP; ++kk) {
            typename Mma<T>::Frag af[TM], bf[TN];
            int boff;
            if constexpr (sizeof(T) == 2) {
                const int q = kk * 2 + (lane >> 5);            // 16-byte chunk holding this lane's 8 k values
                boff = (q ^ fsw) * 16;
            } else {
                const int e = kk * 2 + (lane >> 5);            // float index inside the row
                boff = (((e >> 2) ^ fsw) * 16) + (e & 3) * 4;
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                af[i] = *reinterpret_cast<const typename Mma<T>::Frag *>(As + i * 32 * ROWB + boff);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bf[j] = *reinterpret_cast<const typename Mma<T>::Frag *>(Bs + j * 32 * ROWB + boff);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = Mma<T>::mma(af[i], bf[j], acc[i][j]);
                }
        }
    }

    if (SPLITK == 2) {
        // Stream-K fix-up without atomics.  A workgroup's range is [tail of a tile][whole tiles][head of a tile]; the
        // workgroup holding K step 0 of a tile owns it.  A tail (kt_beg > 0) is always the FIRST segment of its
        // workgroup: it is parked in that workgroup's slot of the workspace (accumulator-register layout, coalesced)
        // and published through a flag, before the workgroup can wait for anything -- so owners only ever wait for
        // work that is never itself blocked.  The owner adds the parked parts of the workgroups after it and writes
        // the finished tile (+ bias) directly.
        unsigned *flags = sk_flags;       // library-owned, all zero between launches (each flag is cleared by its consumer)
        float *slots = Oacc;
        constexpr int SLOT = BM * BN;
        // Slots are written and read 16 bytes per lane with sc1 buffer accesses: a dword sc1 store is one fabric write each
        // (~6x the time per byte of a dwordx4 one), and the 22..54 MB of parked partials per launch were a fifth of the
        // long-reduction launches.  Slot image = accumulator registers in groups of four: [group][lane][4].
        const __amdgpu_buffer_rsrc_t rsrcS = __builtin_amdgcn_make_buffer_rsrc(slots, 0, (unsigned)((size_t)gridDim.x * SLOT * sizeof(float)), 0x00020000);
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
        const unsigned slot_lane = (unsigned)(((size_t)wave * (TM * TN * 16 * 64) + (size_t)lane * 4) * sizeof(float));
        if (kt_beg > 0) {
            const unsigned mine = (unsigned)((size_t)wx * SLOT * sizeof(float)) + slot_lane;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const f32x4 v = {acc[i][j][4 * q4], acc[i][j][4 * q4 + 1], acc[i][j][4 * q4 + 2], acc[i][j][4 * q4 + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrcS, mine + ((i * TN + j) * 4 + q4) * 1024, 0, 16);
                    }
            // Agent-scope write-through accesses (sc1: written through / read past the non-coherent per-XCD L2) instead of
            // release/acquire fences: an agent-scope fence writes back and invalidates the whole XCD L2, which evicts the
            // filter slabs and input tiles every other workgroup of the XCD is streaming (measured: +100 us per launch).
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's slot stores have been acknowledged
            __syncthreads();
            if (tid == 0) __hip_atomic_store(flags + wx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
        if (kt_end < nk) {
            const long tile_end = su - kt_end + nk;       // su was already advanced past this segment
            const int G = gridDim.x;
            long covered = su;
            for (int p = wx + 1; covered < tile_end; ++p) {
                if (tid == 0) y2_sk_wait_and_clear(flags, p);      // (bounded: conv_shared.h)
                __syncthreads();
                const unsigned theirs = (unsigned)((size_t)p * SLOT * sizeof(float)) + slot_lane;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) {
                            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcS, theirs + ((i * TN + j) * 4 + q4) * 1024, 0, 16));
                            acc[i][j][4 * q4] += v[0];
                            acc[i][j][4 * q4 + 1] += v[1];
                            acc[i][j][4 * q4 + 2] += v[2];
                            acc[i][j][4 * q4 + 3] += v[3];
                        }
                covered = (long)(p + 1) * su_total / G;
            }
        }
    }
    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    // With bn_part != NULL (forward of a batch-normalised layer) the tile also contributes its columns' shifted sums
    // sum(y - shift), sum((y - shift)^2) of the STORED (rounded) outputs to one of Y2_BN_PART_ROWS partial rows: the
    // statistics pass over y (a full re-read of every activation, 21 launches per step) is gone.
    const bool stats = !BNBWD && SPLITK != 1 && bn_part != nullptr;
    const bool bstats = BNBWD && SPLITK != 1 && bn_part != nullptr;      // (host: only with the wide-store epilogue below)
    // The partial rows are indexed by (pixel tile, wave row).  When those fit the 256 rows every (row, filter) has exactly one writer:
    // plain stores, bitwise-reproducible statistics -- and the f32 atomics of the 13x13 stages (each a fabric round trip) were 10 us of
    // a 77 us launch (profiles/r02_igemm_ablation.txt).  Larger layers wrap around the rows and keep the atomic adds.
    const bool stats_unique = bz.stat_mask_inv == 0;       // the host found a row for every (pixel tile, wave row) pair
    auto write_tile = [&](auto checked_tag) {      // interior tiles skip the per-element row test (one VALU compare + branch each)
        constexpr bool CHECKED = decltype(checked_tag)::value;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 32 + (lane & 31);
            if (n >= Nf) continue;
            const float bv = (SPLITK != 1 && bias) ? bias[n] : 0.f;
            const float sh = stats ? bn_shift[n] : 0.f;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int mb = m0 + (wm * TM + i) * 32 + 4 * (lane >> 5);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mb + (r & 3) + 8 * (r >> 2);
                    if (!CHECKED || m < M) {
                        if (SPLITK == 1) unsafeAtomicAdd(Oacc + (long)m * Nf + n, acc[i][j][r]);
                        else {
                            float v = acc[i][j][r] + bv;
                            if (act_alpha != 1.0f) v = fmaxf(v, act_alpha * v);      // leaky ReLU of a BN-folded inference layer
                            const T o = (T)v;
                            O[(long)m * ldo + n] = o;
                            const float d = (float)o - sh;
                            s1 += d;
                            s2 += d * d;
                        }
                    }
                }
            }
            if (stats) {      // lanes l and l^32 hold the same column (both pass the n < Nf test together)
                s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 32, 64);
                if (lane < 32) {
                    const int slot = (mt * WGM + wm) & ((Y2_BN_PART_ROWS - 1) ^ bz.stat_mask_inv);
                    float *p1 = bn_part + (long)slot * Nf + n, *p2 = bn_part + (long)(Y2_BN_PART_ROWS + slot) * Nf + n;
                    if (stats_unique) { *p1 = s1; *p2 = s2; }
                    else { unsafeAtomicAdd(p1, s1); unsafeAtomicAdd(p2, s2); }
                }
            }
        }
    };
    // Wide-store epilogue.  The accumulator layout puts ONE output element per lane per register: storing straight from it is
    // 16*TM*TN two-byte stores per lane, and the store queue, not HBM, bounded the short-reduction layers (a 128x128x576 tile is
    // 4.6k MFMA cycles; its 32 narrow store instructions per wave took longer than that).  Instead each wave rounds its
    // sub-tile into a private, padded LDS image (the DMA ring is idle by now), computes the statistics from the rounded values
    // on the way, and reads it back row-major: 16 bytes (8 bf16 / 4 f32 consecutive filters of one pixel) per lane per store.
    {
    constexpr int WROWS = TM * 32, WROWB = TN * 32 * (int)sizeof(T), WSTRIDE = WROWB + 16, WCPR = WROWB / 16;
    constexpr bool WIDE_FITS = SPLITK != 1 && NW * WROWS * WSTRIDE <= NSTAGE * STAGE;
    if (WIDE_FITS && wide_store) {
        // producer-layer operands of the fused BN-backward sums: this lane's VEC per-channel constants and the y vectors of the NIT
        // (pixel, chunk) positions it stores below -- all issued back to back, consumed after the staging loop (one exposed latency per
        // tile instead of one per store).  Stream-K workgroups (256 VGPRs per wave) issue them ahead of the staging loop.
        constexpr int NIT = WROWS * WCPR / 64;
        constexpr int YG = NIT < 4 ? NIT : 4;       // y vectors in flight per lane (the 256-pixel tile spills with all 8)
        constexpr bool BZ_EARLY = SPLITK == 2 && BM == 128;
        float cmu[VEC], cinv[VEC], cga[VEC], cbt[VEC], ps[2][VEC];
        Vec16<T> yv[YG];
        const int bz_nb = min(n0 + wn * TN * 32 + (lane % WCPR) * VEC, Nf - VEC);      // (clamped: out-of-range chunks are never summed)
        auto bz_load_y = [&](int it0) {
#pragma unroll
            for (int u = 0; u < YG; ++u) {
                const int m = min(m0 + wm * WROWS + ((it0 + u) * 64 + lane) / WCPR, M - 1);
                yv[u] = ld16(reinterpret_cast<const T *>(bz.Y) + (long)m * Nf + bz_nb);
            }
        };
        auto bz_prefetch = [&]() {
            bz_load_y(0);
#pragma unroll
            for (int k = 0; k < VEC; k += 4) {
                const f32x4 a = *reinterpret_cast<const f32x4 *>(bz.mean + bz_nb + k), b = *reinterpret_cast<const f32x4 *>(bz.var + bz_nb + k);
                const f32x4 c = *reinterpret_cast<const f32x4 *>(bz.gamma + bz_nb + k), d = *reinterpret_cast<const f32x4 *>(bz.beta + bz_nb + k);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    cmu[k + q] = a[q];
                    cinv[k + q] = 1.0f / sqrtf(b[q] + bz.eps);
                    cga[k + q] = c[q];
                    cbt[k + q] = d[q];
                    ps[0][k + q] = ps[1][k + q] = 0.f;
                }
            }
        };
        if (bstats && BZ_EARLY) bz_prefetch();
        __syncthreads();                       // every wave has finished reading the last K step's stage
        unsigned char *wreg = smem + wave * (WROWS * WSTRIDE);
        const bool tail = m0 + BM > M;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 32 + (lane & 31);
            const bool n_ok = n < Nf;
            const float bv = (bias && n_ok) ? bias[n] : 0.f;
            const float sh = (stats && n_ok) ? bn_shift[n] : 0.f;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = i * 32 + 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2);
                    float v = acc[i][j][r] + bv;
                    if (act_alpha != 1.0f) v = fmaxf(v, act_alpha * v);
                    const T o = (T)v;
                    *reinterpret_cast<T *>(wreg + row * WSTRIDE + (j * 32 + (lane & 31)) * (int)sizeof(T)) = o;
                    if (stats && (!tail || m0 + wm * WROWS + row < M)) {
                        const float d = (float)o - sh;
                        s1 += d;
                        s2 += d * d;
                    }
                }
            }
            if (stats) {
                s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 32, 64);
                if (lane < 32 && n_ok) {
                    const int slot = (mt * WGM + wm) & ((Y2_BN_PART_ROWS - 1) ^ bz.stat_mask_inv);
                    float *p1 = bn_part + (long)slot * Nf + n, *p2 = bn_part + (long)(Y2_BN_PART_ROWS + slot) * Nf + n;
                    if (stats_unique) { *p1 = s1; *p2 = s2; }      // one writer per (row, filter): a store into the zeroed row
                    else { unsafeAtomicAdd(p1, s1); unsafeAtomicAdd(p2, s2); }
                }
            }
        }
        // (LDS operations of one wave execute in order: its own reads below see its own writes above)
        // BN + leaky backward sums of the producer layer (bz): a lane keeps the same VEC filters in every iteration (64 % WCPR == 0)
        static_assert(64 % WCPR == 0, "a lane stays on one column chunk");
        if (bstats && !BZ_EARLY) bz_prefetch();
#pragma unroll
        for (int it = 0; it < WROWS * WCPR / 64; ++it) {
            const int id = it * 64 + lane;
            const int row = id / WCPR, ch = id % WCPR;
            const int m = m0 + wm * WROWS + row;
            const int n = n0 + wn * TN * 32 + ch * VEC;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(wreg + row * WSTRIDE + ch * 16);
            if (bstats && it && it % YG == 0) bz_load_y(it);
            if (m < M && n < Nf) {
                *reinterpret_cast<f32x4 *>(O + (long)m * ldo + n) = v;
                if (bstats) {      // same arithmetic as bn_bwd_reduce_kernel (elementwise.hip), on the rounded gradient just stored
                    const Vec16<T> y = yv[it % YG];
                    Vec16<T> d;
                    d.v = __builtin_bit_cast(decltype(d.v), v);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        const float xh = (y.get(k) - cmu[k]) * cinv[k];
                        const float z = (y.get(k) - cmu[k]) * (cinv[k] * cga[k]) + cbt[k];
                        const float g = z >= 0.f ? d.get(k) : bz.alpha * d.get(k);
                        ps[0][k] += g * xh;
                        ps[1][k] += g;
                    }
                }
            }
        }
        if (bstats) {
            // (the lanes that share a channel chunk meet on the VALU: common.h y2_lane_group_sum)
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                ps[0][k] = y2_lane_group_sum<WCPR>(ps[0][k]);
                ps[1][k] = y2_lane_group_sum<WCPR>(ps[1][k]);
            }
            // The WGM wave rows of the tile hold sums of the SAME channels.  Where the LDS plan leaves room behind the tile image they meet there and the
            // tile leaves ONE partial row per filter (the host counts rows per pixel tile then: y2_bnbwd_rows_per_tile) -- a WGM-th of the adds and of
            // the (tile, wave row) pairs competing for the partial rows: same-address f32 atomics serialise at ~0.1 us each
            constexpr bool RED = BNBWD && WGM > 1 && NW * WROWS * WSTRIDE + NW * WCPR * 2 * VEC * 4 <= NSTAGE * STAGE;
            const int nb = n0 + wn * TN * 32 + lane * VEC;
            bool writer = lane < WCPR && nb < Nf;
            int row_id = mt * WGM + wm;
            if constexpr (RED) {
                float *const red = reinterpret_cast<float *>(smem + NW * WROWS * WSTRIDE);
                if (lane < WCPR) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) { red[(wave * WCPR + lane) * 2 * VEC + k] = ps[0][k]; red[(wave * WCPR + lane) * 2 * VEC + VEC + k] = ps[1][k]; }
                }
                __syncthreads();
                writer = writer && wm == 0;
                row_id = mt;
                if (writer) {
#pragma unroll
                    for (int r = 1; r < WGM; ++r)
#pragma unroll
                        for (int k = 0; k < VEC; ++k) {
                            ps[0][k] += red[((r * WGN + wn) * WCPR + lane) * 2 * VEC + k];
                            ps[1][k] += red[((r * WGN + wn) * WCPR + lane) * 2 * VEC + VEC + k];
                        }
                }
            }
            if (writer) {
                const int slot = row_id & ((Y2_BN_PART_ROWS - 1) ^ bz.stat_mask_inv);
                float *p1 = bn_part + (long)slot * Nf + nb, *p2 = bn_part + (long)(Y2_BN_PART_ROWS + slot) * Nf + nb;
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    if (stats_unique) { p1[k] = ps[0][k]; p2[k] = ps[1][k]; }
                    else { unsafeAtomicAdd(p1 + k, ps[0][k]); unsafeAtomicAdd(p2 + k, ps[1][k]); }
                }
            }
        }
    } else if (m0 + BM <= M) write_tile(std::false_type{});
    else write_tile(std::true_type{});
    }
    if (SPLITK != 2) break;
  }
}

// f32 partial sums [M][Nf] -> O (dtype, pixel stride ldo) + bias
template <typename T>
__global__ void splitk_finish_kernel(const float *__restrict__ acc, const float *__restrict__ bias, T *__restrict__ O, long M, int Nf, int ldo, float act_alpha) {
    const long total = M * Nf;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long m = i / Nf;
        const int n = (int)(i - m * Nf);
        float v = acc[i] + (bias ? bias[n] : 0.f);
        if (act_alpha != 1.0f) v = fmaxf(v, act_alpha * v);
        O[m * ldo + n] = (T)v;
    }
}

struct Tune { int target_blocks; int wide; int remap; int stream; int cus; int stream_max_tiles; int bm256; };
static const Tune &tune() {   // shape-selection constants (each the measured best: profiles/r01_igemm_*.txt, r03_igemm_variant_sweep.txt); two run-time switches remain
    static Tune t = [] {
        Tune v{352, 1, -1, 1, 256, 255, 1};   // K slicing only for grids below half the chip (see choose_ksplit)
        v.stream = y2_env_int("YOLO2_IGEMM_STREAM", v.stream);       // 0: no stream-K anywhere (A/B, debugging a hand-off)
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            v.cus = prop.multiProcessorCount;
        v.cus = y2_env_int("YOLO2_IGEMM_STREAM_WGS", v.cus);      // workgroups of a stream-K launch (default: one per CU; yolo2_set_stream_workgroups at run time)
        v.stream_max_tiles = 3 * v.cus;
        return v;
    }();
    return t;
}
// Stream-K hand-off flags: Y2_STREAM_FLAG_SETS sets of Y2_STREAM_FLAG_WORDS words per device, allocated and zeroed once; a
// launch takes the next set round-robin (launches on different streams may overlap) and leaves it zero (every flag is
// cleared by the one workgroup that waits for it), so no per-launch memset node is needed (16 of them cost 83 us a step).
#define Y2_STREAM_FLAG_SETS 8
// The ONE piece of library-owned device state (declared in include/yolo2_hip.h): 32 KiB per device, created on first use
// under a mutex, handed out round-robin under the same mutex (host threads may launch concurrently), freed by yolo2_shutdown().
static std::mutex g_flag_mutex;
static unsigned *g_flag_pool[64] = {nullptr};
static unsigned g_flag_counter[64] = {0};
static std::atomic<unsigned> g_sk_wait_ticks{0};       // 0 = Y2_SK_DEFAULT_WAIT_TICKS (tests shorten it: yolo2_debug_set_streamk_wait_us)
static std::atomic<int> g_sk_unclamped{0};             // tests only: lets a forced grid exceed the number of K steps (the partition bug the wait bound exists for)
__global__ void async_error_gather_kernel(const unsigned *__restrict__ pool, unsigned *__restrict__ host_words) {
    if (threadIdx.x < Y2_STREAM_FLAG_SETS)
        host_words[threadIdx.x] = __hip_atomic_load(pool + (size_t)threadIdx.x * Y2_STREAM_FLAG_STRIDE + Y2_STREAM_FLAG_WORDS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the current device's pool, created on first use (caller holds g_flag_mutex); does NOT advance the rotating set counter
static unsigned *ensure_pool_locked(int dev) {
    if (!g_flag_pool[dev]) {
        unsigned *p = nullptr;
        const size_t bytes = (size_t)Y2_STREAM_FLAG_SETS * Y2_STREAM_FLAG_STRIDE * sizeof(unsigned);
        if (hipMalloc((void **)&p, bytes) != hipSuccess) return nullptr;
        if (hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipFree(p); return nullptr; }
        g_flag_pool[dev] = p;
    }
    return g_flag_pool[dev];
}
static unsigned *stream_flags() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(g_flag_mutex);
    unsigned *pool = ensure_pool_locked(dev);
    if (!pool) return nullptr;
    return pool + (size_t)(g_flag_counter[dev]++ % Y2_STREAM_FLAG_SETS) * Y2_STREAM_FLAG_STRIDE;
}
// Enqueues, on `stream`, a copy of the CURRENT device's give-up counters (one word per flag set) into caller-owned pinned host memory: the
// asynchronous form of yolo2_check_async_errors.  A host that finds a non-zero word once the copy has completed (event / stream query) calls
// yolo2_check_async_errors, which reports and re-zeroes the pool.  Costs one 32-byte device-to-host copy; nothing is synchronised.
extern "C" int yolo2_async_error_snapshot(unsigned *host_words, void *stream) {
    if (!host_words) { yolo2_set_error("yolo2_async_error_snapshot: host_words is NULL"); return YOLO2_E_ARG; }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { yolo2_set_error("yolo2_async_error_snapshot: no current device"); return YOLO2_E_LAUNCH; }
    unsigned *pool;
    {
        std::lock_guard<std::mutex> lock(g_flag_mutex);
        pool = g_flag_pool[dev];
    }
    if (!pool) {      // no stream-K launch has run on this device yet: nothing can have given up
        for (int i = 0; i < YOLO2_ASYNC_ERROR_WORDS; ++i) host_words[i] = 0u;
        return YOLO2_OK;
    }
    static_assert(YOLO2_ASYNC_ERROR_WORDS == Y2_STREAM_FLAG_SETS, "one status word per flag set");
    // one eight-lane kernel writing straight into the pinned (device-mapped) host words: a strided 2-D copy of eight words goes through the runtime's
    // staging path and costs tens of microseconds of stream time (measured: +0.4 % on the training step when enqueued every eighth step)
    async_error_gather_kernel<<<1, 64, 0, (hipStream_t)stream>>>(pool, host_words);
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}
// Synchronises `stream` and reports what only the device can know: a stream-K owner that gave up waiting for a partner's partial tile
// (y2_sk_wait_and_clear).  The pool is re-zeroed so that the process can go on, but every result since the previous check is suspect.
// Per device: reads and resets the pool of the CURRENT device only (a process driving several devices checks each under hipSetDevice).
extern "C" int yolo2_check_async_errors(void *stream) {
    int dev = 0;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        yolo2_set_error("yolo2_check_async_errors: HIP error: %s", hipGetErrorString(hipGetLastError()));
        return YOLO2_E_LAUNCH;
    }
    std::lock_guard<std::mutex> lock(g_flag_mutex);
    if (!g_flag_pool[dev]) return YOLO2_OK;
    unsigned status[Y2_STREAM_FLAG_SETS] = {0};
    if (hipMemcpy2D(status, sizeof(unsigned), g_flag_pool[dev] + Y2_STREAM_FLAG_WORDS, Y2_STREAM_FLAG_STRIDE * sizeof(unsigned), sizeof(unsigned),
                    Y2_STREAM_FLAG_SETS, hipMemcpyDeviceToHost) != hipSuccess) {
        yolo2_set_error("yolo2_check_async_errors: reading the stream-K status words failed");
        return YOLO2_E_LAUNCH;
    }
    unsigned gave_up = 0;
    for (int i = 0; i < Y2_STREAM_FLAG_SETS; ++i) gave_up += status[i];
    if (!gave_up) return YOLO2_OK;
    (void)hipDeviceSynchronize();
    (void)hipMemset(g_flag_pool[dev], 0, (size_t)Y2_STREAM_FLAG_SETS * Y2_STREAM_FLAG_STRIDE * sizeof(unsigned));
    const unsigned ticks = g_sk_wait_ticks.load(std::memory_order_relaxed);
    for (int i = 0; ticks && i < Y2_STREAM_FLAG_SETS; ++i)
        (void)hipMemcpy(g_flag_pool[dev] + (size_t)i * Y2_STREAM_FLAG_STRIDE + Y2_STREAM_FLAG_WORDS + 1, &ticks, sizeof(ticks), hipMemcpyHostToDevice);
    yolo2_set_error("stream-K hand-off: %u wait(s) for a partner workgroup's partial tile gave up; convolution outputs since the last check are invalid", gave_up);
    return YOLO2_E_LAUNCH;
}
// tests: wait limit of the stream-K owners in microseconds (0 = default, 2 s) and the grid clamp (unclamped != 0 lets yolo2_debug_set_pp force
// more workgroups than K steps -- an unserviceable partition)
extern "C" int yolo2_debug_set_streamk_wait_us(int us, int unclamped) {
    if (us < 0 || (unsigned)us > 0xffffffffu / 100u) { yolo2_set_error("yolo2_debug_set_streamk_wait_us: us outside 0 .. %u (100 MHz ticks in 32 bits)", 0xffffffffu / 100u); return YOLO2_E_ARG; }
    const unsigned ticks = (unsigned)us * 100u;
    g_sk_wait_ticks.store(ticks, std::memory_order_relaxed);
    g_sk_unclamped.store(unclamped, std::memory_order_relaxed);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { yolo2_set_error("yolo2_debug_set_streamk_wait_us: no current device"); return YOLO2_E_LAUNCH; }
    std::lock_guard<std::mutex> lock(g_flag_mutex);
    if (!ensure_pool_locked(dev)) { yolo2_set_error("yolo2_debug_set_streamk_wait_us: no flag pool"); return YOLO2_E_LAUNCH; }
    (void)hipDeviceSynchronize();
    for (int i = 0; i < Y2_STREAM_FLAG_SETS; ++i)
        if (hipMemcpy(g_flag_pool[dev] + (size_t)i * Y2_STREAM_FLAG_STRIDE + Y2_STREAM_FLAG_WORDS + 1, &ticks, sizeof(ticks), hipMemcpyHostToDevice) != hipSuccess)
            return YOLO2_E_LAUNCH;
    return YOLO2_OK;
}
extern "C" int yolo2_shutdown(void) {
    std::lock_guard<std::mutex> lock(g_flag_mutex);
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (int d = 0; d < 64; ++d)
        if (g_flag_pool[d]) {
            if (hipSetDevice(d) == hipSuccess) { (void)hipDeviceSynchronize(); (void)hipFree(g_flag_pool[d]); }
            g_flag_pool[d] = nullptr;
            g_flag_counter[d] = 0;
        }
    (void)hipSetDevice(cur);
    return YOLO2_OK;
}

// number of K slices: when the M x N tile grid alone cannot fill 256 CUs with ~2-3 resident workgroups
// each, slice K so that it does, keeping >= 8 K tiles per slice
static int choose_ksplit(int tiles, int nk, int target) {
    // measured (profiles/): pays for the 88-tile data gradients; elsewhere atomics + the finishing pass cost more than
    // the 8-wave / wide-K variants gain
    if (target <= 0 || nk < 128 || tiles > 128) return 1;
    int ks = (target + tiles - 1) / tiles;
    int max_ks = nk / 8;
    if (ks > max_ks) ks = max_ks;
    return ks < 1 ? 1 : ks;
}

// the plan of the calling thread's most recent launch (yolo2_debug_last_conv_plan): tests assert that the variant they
// mean to check is the one that ran, since the choice is shape-driven
static thread_local int g_last_plan[8] = {0, 0, 0, 0, 0, 0, 0, 0};
// Partial rows a statistics-producing launch touches (yolo2_last_bn_part_rows): what the consumer's prologue has to sum.  Up to
// Y2_BN_PART_ROWS (pixel tile, wave row) pairs get a row each (plain stores, one writer per element); larger grids wrap around
// R rows with f32 atomic adds, R chosen for <= ~170 adds per address (the 104x104 layer; 20-90 elsewhere) and <= 128 rows to read.
static thread_local int g_last_stat_rows = Y2_BN_PART_ROWS;
// most rows the consumer's prologue accepts for Nf channels (elementwise.hip fin_shape_ok: rows x slice x 8 bytes <= 128 KB), <= 256
static int y2_stat_rows_limit(int Nf, int vec) {
    int lpr = Nf / vec;
    if (lpr > 16) lpr = 16;
    if (lpr < 1) lpr = 1;
    const long lim = (128L << 10) / ((long)lpr * vec * 8);
    return lim > Y2_BN_PART_ROWS ? Y2_BN_PART_ROWS : (int)lim;
}
static int y2_stat_rows(long wave_rows, int limit) {
    if (wave_rows <= limit) return (int)wave_rows;         // one row per (pixel tile, wave row): plain stores
    // (<= ~16 adds per address: same-address f32 atomics serialise at ~0.1 us each -- 42 per address cost the 52 x 52 BN-backward launch 12 us)
    int r = 16;
    while (r < 128 && (long)r * 16 < wave_rows) r <<= 1;
    while (r > limit) r >>= 1;
    return r;
}
// (batch 8, 26x26 stage: 172 unique rows x 512 channels were more than the consumer reads -- those five layers fell back to the
// separate finalisation launches; they now wrap around 16 rows like the larger grids)
static Y2BnBwd y2_with_stat_rows(Y2BnBwd bz, long wave_rows, int Nf, int vec) {
    const int limit = y2_stat_rows_limit(Nf, vec);
    const int r = y2_stat_rows(wave_rows, limit);
    g_last_stat_rows = r;
    bz.stat_mask_inv = wave_rows <= limit ? 0 : (Y2_BN_PART_ROWS - 1) ^ (r - 1);
    return bz;
}
extern "C" int yolo2_last_bn_part_rows(void) { return g_last_stat_rows; }
#define Y2_IGEMM_ARGS (const T *)P, p_bytes, (const T *)F, f_bytes, bias, (T *)O, ws, H, W, Cp, ldp, Nf, ldo, M, NT, remap, bn_shift
#define Y2_IGEMM_BM(BMv, BNv, WGNv, NSv, KSv, SPLITv, CTv, CHv, NWv, gridv)                                        \
    do {                                                                                                           \
        const dim3 g_ = (gridv);                                                                                   \
        const int plan_[8] = {BMv, BNv, NWv, CHv, NSv, SPLITv, (int)g_.x, (int)g_.y};                              \
        for (int i_ = 0; i_ < 8; ++i_) g_last_plan[i_] = plan_[i_];                                                \
        const bool fusedbw_ = bz.Y && SPLITv != 1 && !CTv && wide_store && igemm_wide_fits<T, BMv, BNv, WGNv, NSv, CHv, NWv>();                      \
        const Y2BnBwd bz_ = y2_with_stat_rows(bz, (long)cdiv(M, BMv) * (fusedbw_ ? y2_bnbwd_rows_per_tile<T, BMv, BNv, WGNv, NSv, CHv, NWv>() : (NWv / WGNv)), Nf, VEC); \
        if (!bz.Y)                                                                                                 \
            conv_igemm_kernel<T, BNv, WGNv, NSv, KSv, SPLITv, CTv, CHv, NWv, BMv, false><<<g_, NWv * 64, 0, st>>>(  \
                Y2_IGEMM_ARGS, bn_part, sk_flags, act_alpha, wide_store, bz_);                                     \
        else if (SPLITv != 1 && !CTv && wide_store && igemm_wide_fits<T, BMv, BNv, WGNv, NSv, CHv, NWv>())         \
            conv_igemm_kernel<T, BNv, WGNv, NSv, KSv, (SPLITv == 1 ? 0 : SPLITv), false, CHv, NWv, BMv, true><<<g_, NWv * 64, 0, st>>>( \
                Y2_IGEMM_ARGS, bn_part, sk_flags, act_alpha, wide_store, bz_);                                     \
        else {      /* this variant has no on-chip tile image to reduce from: the caller runs the two-step form */ \
            *stats_done = false;                                                                                   \
            conv_igemm_kernel<T, BNv, WGNv, NSv, KSv, SPLITv, CTv, CHv, NWv, BMv, false><<<g_, NWv * 64, 0, st>>>(  \
                Y2_IGEMM_ARGS, nullptr, sk_flags, act_alpha, wide_store, Y2BnBwd{nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, 0}); \
        }                                                                                                          \
    } while (0)
#define Y2_IGEMM(BNv, WGNv, NSv, KSv, SPLITv, CTv, CHv, NWv, gridv) Y2_IGEMM_BM(128, BNv, WGNv, NSv, KSv, SPLITv, CTv, CHv, NWv, gridv)
// kernel size x channel tail (4-chunk rows only; 8-chunk rows require Cp % (8*VEC) == 0)
#define Y2_IGEMM_KS_CT(BNv, WGNv, NSv, SPLITv, NWv, gridv)                              \
    do {                                                                               \
        if (ksize == 3) {                                                              \
            if (ctail) Y2_IGEMM(BNv, WGNv, NSv, 3, SPLITv, true, 4, NWv, gridv);       \
            else Y2_IGEMM(BNv, WGNv, NSv, 3, SPLITv, false, 4, NWv, gridv);            \
        } else {                                                                       \
            if (ctail) Y2_IGEMM(BNv, WGNv, NSv, 1, SPLITv, true, 4, NWv, gridv);       \
            else Y2_IGEMM(BNv, WGNv, NSv, 1, SPLITv, false, 4, NWv, gridv);            \
        }                                                                              \
    } while (0)
#define Y2_IGEMM_KS_WIDE(SPLITv, gridv)                                                 \
    do {                                                                               \
        if (ksize == 3) Y2_IGEMM(128, 2, 3, 3, SPLITv, false, 8, 8, gridv);            \
        else Y2_IGEMM(128, 2, 3, 1, SPLITv, false, 8, 8, gridv);                       \
    } while (0)

// partial rows one tile of a BN-backward instantiation writes: 1 when its wave rows meet in LDS (mirror of RED in the kernel), else one per wave row
template <typename T, int BMv, int BN, int WGN, int NSTAGE, int CH, int NW>
static constexpr int y2_bnbwd_rows_per_tile() {
    constexpr int RPI = 64 / CH, ROWB = CH * 16, WGM = NW / WGN, TM = BMv / WGM / 32, TN = BN / WGN / 32, VEC = 16 / (int)sizeof(T);
    constexpr int B_IT = (BN / RPI + NW - 1) / NW, STAGE = (BMv + B_IT * NW * RPI) * ROWB;
    constexpr int WROWS = TM * 32, WROWB = TN * 32 * (int)sizeof(T), WSTRIDE = WROWB + 16, WCPR = WROWB / 16;
    return (WGM > 1 && NW * WROWS * WSTRIDE + NW * WCPR * 2 * VEC * 4 <= NSTAGE * STAGE) ? 1 : WGM;
}
// does this instantiation have the LDS-transposed (wide-store) epilogue?  (mirror of WIDE_FITS in the kernel)
template <typename T, int BMv, int BN, int WGN, int NSTAGE, int CH, int NW>
static constexpr bool igemm_wide_fits() {
    constexpr int RPI = 64 / CH, ROWB = CH * 16, WGM = NW / WGN, TM = BMv / WGM / 32, TN = BN / WGN / 32;
    constexpr int B_IT = (BN / RPI + NW - 1) / NW, STAGE = (BMv + B_IT * NW * RPI) * ROWB;
    return NW * (TM * 32) * (TN * 32 * (int)sizeof(T) + 16) <= NSTAGE * STAGE;
}

// tap-fused 3x3 variant on/off (default: env YOLO2_IGEMM_TAP, else on); yolo2_debug_set_igemm_tap flips it at run time so that
// tests can compare both variants on the same inputs
// 0 = per-tap kernels only, non-zero = the ping-pong tap-fused kernel (conv_pp.hip) where its launch rule admits it (default).  (The round-2
// tap-fused kernel it replaced -- slower on every layer it took, profiles/r04_pp2_sched_b16.txt column tap-r2 -- is gone.)
// 3 = the loader / consumer member of the tap-fused family (conv_s4.hip: four computing waves of 128 x 64 + four loader waves), same launch rule
static std::atomic<int> g_igemm_tap{y2_env_int("YOLO2_IGEMM_TAP", 2)};
extern "C" int yolo2_debug_set_igemm_tap(int mode) {
    g_igemm_tap.store(mode == 3 ? 3 : (mode != 0 ? 2 : 0), std::memory_order_relaxed);
    return YOLO2_OK;
}
static std::atomic<int> g_s4_abl{0};      // experiments build of conv_s4.hip: timing ablations / phase stamps (yolo2_debug_set_s4_abl)
extern "C" int yolo2_debug_set_s4_abl(int abl) {
    g_s4_abl.store(abl, std::memory_order_relaxed);
    return YOLO2_OK;
}
// ping-pong kernel knobs (A/B runs and tests): grid 0 = by rule, 1 = stream-K (one workgroup per CU), 2 = one workgroup per tile;
// dmapos 0/1 = DMA pieces at the head of the LOAD phase / inside the MFMA phase; min_steps, min_share = the launch gates below (< 0: keep)
static std::atomic<int> g_pp_grid{0};
static std::atomic<int> g_pp_dmapos{y2_env_int("YOLO2_PP_SCHED", 2)};      // conv_pp.hip SCHED (2 = fragment reads, then the DMA pieces)
static std::atomic<long> g_pp_min_steps{18};
static std::atomic<long> g_pp_min_share{24};
static const int g_pp_long_share = y2_env_int("YOLO2_PP_LONG_SHARE", 26);      // (0 = round 4's rule, for the A/B)
// cost units charged to the owner of a stream-K tile, in K steps (conv_pp.hip "cost-balanced shares"); 0 = equal K-step shares (round 5).  6 = what
// a second prologue costs a two-segment workgroup: the dominant launches 54.46 -> 53.9 us, the step -12 us, batch 8 / multi-scale equal or better;
// 8 and more lose again -- the owner then waits for partners that take longer (profiles/r06_pp_cv_sweep.txt, its last block)
static std::atomic<int> g_pp_cv{y2_env_int("YOLO2_PP_CV", 6)};
extern "C" int yolo2_debug_set_pp_cost(int cv) {
    if (cv < 0 || cv > 4096) { yolo2_set_error("yolo2_debug_set_pp_cost: 0 .. 4096 K steps"); return YOLO2_E_ARG; }
    g_pp_cv.store(cv, std::memory_order_relaxed);
    return YOLO2_OK;
}
extern "C" int yolo2_debug_set_pp(int grid, int dmapos, int min_steps, int min_share) {
    if (grid != -1) g_pp_grid.store(grid, std::memory_order_relaxed);
    if (dmapos >= 0) g_pp_dmapos.store(dmapos, std::memory_order_relaxed);
    if (min_steps >= 0) g_pp_min_steps.store(min_steps, std::memory_order_relaxed);
    if (min_share >= 0) g_pp_min_share.store(min_share, std::memory_order_relaxed);
    return YOLO2_OK;
}

// Stream-K launches take exactly one workgroup per CU.  A process that shares the GPU with persistent kernels of another stream -- an
// RCCL ring holds one workgroup per channel for the whole collective -- would get a second, mostly idle wave of workgroups on the CUs
// that are left; yolo2_set_stream_workgroups(n) (data-parallel sessions: CUs - [mi355x] comm_cus) sizes the launches for the CUs that
// are free.  0 restores the default (all CUs, or YOLO2_IGEMM_STREAM_WGS).  Correctness never depends on the value: owners only wait
// for parked tails, and a tail is the first thing its workgroup does (deadlock-free under any residency; tests/test_streamk_occupied_gpu.py).
static std::atomic<int> g_stream_wgs{0};
extern "C" int yolo2_set_stream_workgroups(int n) {
    if (n < 0 || n > Y2_STREAM_FLAG_WORDS) { yolo2_set_error("yolo2_set_stream_workgroups: 0 (default) .. %d", Y2_STREAM_FLAG_WORDS); return YOLO2_E_ARG; }
    g_stream_wgs.store(n, std::memory_order_relaxed);
    return YOLO2_OK;
}
extern "C" int yolo2_get_stream_workgroups(void) {
    const int n = g_stream_wgs.load(std::memory_order_relaxed);
    return n > 0 ? n : tune().cus;
}
static Tune tune_now() {
    Tune t = tune();
    const int n = g_stream_wgs.load(std::memory_order_relaxed);
    if (n > 0) { t.cus = n; t.stream_max_tiles = 3 * n; }
    return t;
}

template <typename T>
static int launch_conv(const void *P, const void *F, const float *bias, void *O, float *ws, size_t ws_bytes, int B, int H, int W,
                       int Cp, int ldp, int Nf, int ldo, int ksize, hipStream_t st, const float *bn_shift, float *bn_part, bool *stats_done, float act_alpha,
                       const Y2BnBwd bz = Y2BnBwd{nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, 0}) {
    const int M = B * H * W;
    const int MT = cdiv(M, 128);
    constexpr int VEC = 16 / sizeof(T);
    constexpr int BK = 4 * VEC;
    const Tune tu = tune_now();
    const unsigned p_bytes = (unsigned)((size_t)M * ldp * sizeof(T));
    const unsigned f_bytes = (unsigned)((size_t)Nf * ksize * ksize * Cp * sizeof(T));
    const bool ctail = (Cp % BK) != 0;
    unsigned *sk_flags = nullptr;
    // 16-byte epilogue stores need 16-byte aligned pixel rows, and a partial last chunk may only spill into this tensor's own
    // padding lanes (written as zeros), never into a neighbouring concat slice
    const int wide_store = ((uintptr_t)O & 15) == 0 && ldo % VEC == 0 && (Nf % VEC == 0 || ldo < Nf + VEC);
    // XCD mapping: filter operand small -> contiguous M runs per XCD; else filter tiles pinned per XCD
    const int remap = tu.remap >= 0 ? tu.remap : (f_bytes <= (3u << 19) ? 1 : 0);
    const int MT2 = cdiv(M, 256), NT2 = cdiv(Nf, 128);
    const long nk_wide = (Cp % (8 * VEC) == 0) ? (long)ksize * ksize * (Cp / (8 * VEC)) : 0;
    // Long reductions on under-filled grids (13x13 stages: 1024+ input channels): 256 x 128 tile, 8 waves of 64 x 64
    // (half the LDS fragment traffic and 25 % less DMA per flop than 32 x 64 per wave), always stream-K.  Measured
    // (profiles/r01_igemm_bm256.txt): +23 % on the 3072-channel layer; shorter reductions, fuller grids and grids under
    // 64 tiles (each tile cut into > 4 parts: the owner's serial fix-up) lose.
    if constexpr (std::is_same<T, bf16>::value) {
        const int tap_mode = g_igemm_tap.load(std::memory_order_relaxed);
        // Ping-pong tap-fused kernel (conv_pp.hip): 3x3 layers on images up to 55 wide whose input is a multiple of 64 channels.  Its
        // fixed cost per launch (~24 us: prologue, stream-K hand-off, epilogue) against ~0.65 us per K step decides where it wins
        // (profiles/r04_pp2_*.txt, r04_pp3_*.txt, batch 16 and 8):
        //   * a tile grid that gives 60-100 % of the CUs one whole tile each: one workgroup per tile, no hand-off (26x26 256->512 forward
        //     30.8 vs 35.1 us; 52x52 data gradients 31 vs 36.5 us);
        //   * else stream-K over one workgroup per CU when a workgroup's share is >= 24 K steps and a tile is cut into at most four shares
        //     (the owner adds its partners' parked partials one after the other: the 13x13 512 -> 1024 forward, 88 tiles, 38.9 vs 41.0 us;
        //     its data gradient, 44 tiles of the same 24.75 steps per workgroup, 42.4 vs 41.1 us: per-tap; >= 1024-channel 13x13 layers
        //     55 vs 60-75 us, 126 vs 144-181 us);
        //   * a data gradient that also reduces the producer's BN-backward sums takes it from 10 steps per workgroup when it has >= 8192
        //     pixels: the per-tap kernels' form of that epilogue costs 12-30 us there, this one 4-7.
        if (tap_mode != 0 && ksize == 3 && Cp % 64 == 0 && W <= 55 && Nf > 64 && wide_store && ws && tu.stream &&
            (long)9 * (Cp / 64) >= g_pp_min_steps.load(std::memory_order_relaxed) && tu.cus <= Y2_STREAM_FLAG_WORDS) {
            const long tiles_t = (long)MT2 * NT2, units_p = tiles_t * 9 * (Cp / 64);
            const int gm = g_pp_grid.load(std::memory_order_relaxed);
            const bool can_stream = (size_t)tu.cus * 256 * 128 * sizeof(float) <= ws_bytes;
            int grid = 0;
            if (gm == 1) grid = can_stream ? tu.cus : 0;
            else if (gm == 2) grid = tiles_t <= Y2_STREAM_FLAG_WORDS ? (int)tiles_t : 0;
            else if (gm >= 8) grid = (can_stream && gm <= tu.cus) ? gm : 0;                  // explicit workgroup count (sweeps)
            else if (gm <= -2) grid = (can_stream && tiles_t * -gm <= tu.cus) ? (int)(tiles_t * -gm) : 0;      // -P: every tile cut into exactly P shares
            else if (tiles_t * 10 >= (long)tu.cus * 6 && tiles_t <= tu.cus) grid = (int)tiles_t;
            else if (can_stream && ((units_p >= g_pp_min_share.load(std::memory_order_relaxed) * tu.cus && tiles_t * 4 >= tu.cus) ||
                                    (bz.Y && M >= 8192 && units_p >= 10L * tu.cus) ||
                                    // long shares (>= 26 steps per workgroup) outweigh a tile cut into 5-8 of them -- batch 8, 48 tiles: conv20
                                    // forward 79 vs 119 us, conv18 41.2 vs 44.6 (profiles/r04_pp6_b8.txt); the COCO-80 batch-8 step 2.84 -> 2.82 ms
                                    // (profiles/r05_long_share_b8.txt)
                                    (g_pp_long_share > 0 && units_p >= (long)g_pp_long_share * tu.cus && tiles_t * 8 >= tu.cus))) grid = tu.cus;
            // every workgroup of a stream-K launch must hold at least one K step: an owner waits for the flag of EVERY workgroup whose range
            // lies inside its tile, and one without work never raises it (only a forced grid on a tiny problem gets here: the rule above
            // asks for >= 10 steps per workgroup)
            if (grid > units_p && !g_sk_unclamped.load(std::memory_order_relaxed)) grid = (int)units_p;
            if (grid > 0 && (sk_flags = stream_flags()) != nullptr) {
                const int plan_[8] = {256, 128, 8, 8, 18, 2, grid, 1};      // "stages" 18: nine taps per halo image, two phases per tap
                for (int i_ = 0; i_ < 8; ++i_) g_last_plan[i_] = plan_[i_];
                // partial rows: forward statistics one per (pixel tile, wave row); BN-backward sums of the ping-pong kernel one per pixel tile (its wave
                // rows meet in LDS); the loader / consumer kernel one per (pixel tile, 64-row group)
                const Y2BnBwd bz_ = y2_with_stat_rows(bz, (bz.Y && tap_mode != 3) ? (long)MT2 : (long)MT2 * 4, Nf, VEC);
                // owner cost of the cost-balanced partition: below half a workgroup's share, so that every workgroup keeps real K steps (an owner
                // waits for the flag of every workgroup inside its tile); whole-tile grids have no shares to balance
                int cv = grid == tiles_t ? 0 : g_pp_cv.load(std::memory_order_relaxed);
                if (cv > units_p / grid / 2) cv = (int)(units_p / grid / 2);
                g_last_plan[7] = 1 + (cv << 8);      // plan word grid_y: bits 8.. = the owner cost in use
                if (tap_mode == 3) {
                    g_last_plan[2] = 4;             // plan word waves: four COMPUTING waves (+ four loaders)
                    if (y2_conv3x3_s4_launch(P, p_bytes, F, f_bytes, bias, O, ws, H, W, Cp, ldp, Nf, ldo, M, NT2, bn_shift, bn_part, sk_flags, act_alpha, bz_, 1, grid,
                                             g_s4_abl.load(std::memory_order_relaxed), st) == 0)
                        return 0;
                    g_last_plan[2] = 8;
                }
                if (y2_conv3x3_pp_launch(P, p_bytes, F, f_bytes, bias, O, ws, H, W, Cp, ldp, Nf, ldo, M, NT2, bn_shift, bn_part, sk_flags, act_alpha, bz_,
                                         /* K rotation of the stream-K tiles (profiles/r03_l2_stationary_ab.md) */ 1, grid, g_pp_dmapos.load(std::memory_order_relaxed), cv, st) == 0)
                    return 0;
            }
        }
    }
    if (Nf > 64 && tu.bm256 && ws && tu.stream && nk_wide >= 128 && MT2 * NT2 >= 64 && MT2 * NT2 <= 3 * tu.cus && tu.cus <= Y2_STREAM_FLAG_WORDS &&
        (size_t)tu.cus * 256 * 128 * sizeof(float) <= ws_bytes && (sk_flags = stream_flags()) != nullptr) {
        const int NT = NT2;
        dim3 grid(tu.cus);
        if (ksize == 3) Y2_IGEMM_BM(256, 128, 2, 3, 3, 2, false, 8, 8, grid);
        else Y2_IGEMM_BM(256, 128, 2, 3, 1, 2, false, 8, 8, grid);
    } else if (Nf > 64) {
        // 128 x 128 tile, 8 waves (4 x 2, each 32 x 64): two waves per SIMD even when a CU holds a single workgroup.
        // Grids of <= 256 tiles (the 13x13 / 26x26 stages at batch 16) cannot fill the chip with workgroups, so they
        // take 128-byte K rows (16 MFMAs per wave between barriers, 96 KiB ring, one workgroup per CU); larger grids
        // keep 64-byte rows (48 KiB ring, 3 workgroups per CU).  Measured in profiles/r01_igemm_variants.txt.
        const int NT = cdiv(Nf, 128);
        // K slicing (grids <= 128 tiles with a long reduction) multiplies the workgroup count, so it pairs with the
        // narrow variant (3 workgroups per CU); unsliced small grids take the wide one.
        int ks = ws ? choose_ksplit(MT * NT, ksize * ksize * cdiv(Cp, 4 * VEC), tu.target_blocks) : 1;
        if (ks > 1 && (size_t)M * Nf * sizeof(float) > ws_bytes) ks = 1;
        const bool wide = tu.wide && ks == 1 && MT * NT <= 256 && Cp % (8 * VEC) == 0;
        // stream-K: a grid that cannot give every CU a tile (13x13 stages at batch 16: 176 tiles, 88 for the narrower
        // data gradients) is run by exactly one 8-wave wide-row workgroup per CU, each taking an equal contiguous share
        // of the flat (tile, K step) space (>= 24 K steps each, else the two partial-tile epilogues dominate).  Grids of
        // 1-3 tiles per CU gain only when the reduction is long (>= 96 K steps per tile; measured, profiles/)
        const long units = (long)MT * NT * ksize * ksize * (Cp / (8 * VEC));
        const bool stream = tu.stream && ws && tu.wide && Cp % (8 * VEC) == 0 && units >= 24L * tu.cus &&
                            (MT * NT < tu.cus || (MT * NT <= tu.stream_max_tiles && units / (MT * NT) >= 96)) &&
                            tu.cus <= Y2_STREAM_FLAG_WORDS && (size_t)tu.cus * 128 * 128 * sizeof(float) <= ws_bytes &&
                            (sk_flags = stream_flags()) != nullptr;
        if (stream) {
            dim3 grid(tu.cus);
            if (ksize == 3) Y2_IGEMM(128, 2, 3, 3, 2, false, 8, 8, grid);
            else Y2_IGEMM(128, 2, 3, 1, 2, false, 8, 8, grid);
        } else if (ks > 1) {
            *stats_done = false;        // partial tiles meet only in the finishing kernel
            if (hipMemsetAsync(ws, 0, (size_t)M * Nf * sizeof(float), st) != hipSuccess) return 1;
            dim3 grid(MT * NT, ks);
            Y2_IGEMM_KS_CT(128, 2, 3, 1, 4, grid);     // 4-wave workgroups: 3 per CU, measured best with slicing
            long total = (long)M * Nf;
            int g = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
            splitk_finish_kernel<T><<<g, 256, 0, st>>>(ws, bias, (T *)O, M, Nf, ldo, act_alpha);
        } else {
            ws = nullptr;
            dim3 grid(MT * NT);
            if (wide) Y2_IGEMM_KS_WIDE(0, grid);
            else if (Cp % (8 * VEC) == 0) {
                // full grids with >= 64 channels: 128-byte rows on a 2-stage ring (64 KiB: two workgroups per CU, 8 MFMAs per
                // wave between barriers instead of 4) -- +8..19 % on the 52x52 / 26x26 stages (profiles/r01_igemm_wide_ns2.txt)
                if (ksize == 3) Y2_IGEMM(128, 2, 2, 3, 0, false, 8, 8, grid);
                else Y2_IGEMM(128, 2, 2, 1, 0, false, 8, 8, grid);
            } else Y2_IGEMM_KS_CT(128, 2, 3, 0, 8, grid);
        }
    } else {
        const int NT = 1;
        ws = nullptr;
        dim3 grid(MT);
        if (ksize == 3 && Cp % (8 * VEC) == 0) {
            // 3x3 with <= 64 filters and >= 64 channels (the 208x208 / 104x104 data gradients): 128-byte rows, 2-stage ring;
            // +35 % / +20 % on those two launches; the 1x1 layers measured no gain (profiles/r01_igemm_wide_ns2.txt)
            if (Nf > 32) Y2_IGEMM(64, 1, 2, 3, 0, false, 8, 4, grid);
            else Y2_IGEMM(32, 1, 2, 3, 0, false, 8, 4, grid);
        } else if (Nf > 32) Y2_IGEMM_KS_CT(64, 1, 3, 0, 4, grid);
        else Y2_IGEMM_KS_CT(32, 1, 3, 0, 4, grid);
    }
    return 0;
}

// Does yolo2_conv2d_dgrad_bn take the producer layer's sums from its own epilogue for this shape?  (The answer the launch rule gives before the variant
// is known: K-sliced variants still fall back inside the call.)  A/B: YOLO2_FUSE_BN_BWD=0 never; YOLO2_FUSE_BN_BWD_NARROW=1 also the 3x3 data gradients
// with <= 64 filters on >= 100k pixels -- conv4's at batch 16, where the fused epilogue costs the per-tap kernel 32 us and the separate reduction 19:
// those run plain by default, and a host that asks first (yolo2_conv2d_dgrad_bn_fuses) schedules reduction + folded apply itself.
static bool y2_dgrad_bn_fuses(int B, int H, int W, int Nf, int ksize, int dtype) {
    static const bool fuse = y2_env_int("YOLO2_FUSE_BN_BWD", 1) != 0;
    static const bool fuse_narrow = y2_env_int("YOLO2_FUSE_BN_BWD_NARROW", 0) != 0;
    const int vec = dtype == YOLO2_BF16 ? 8 : 4;
    return fuse && Nf % vec == 0 && (fuse_narrow || !(ksize == 3 && Nf <= 64 && (long)B * H * W >= 100000));
}
extern "C" int yolo2_conv2d_dgrad_bn_fuses(int B, int H, int W, int Nf, int ksize, int dtype) {
    return (B > 0 && H > 0 && W > 0 && Nf > 0 && (dtype == YOLO2_F32 || dtype == YOLO2_BF16) && y2_dgrad_bn_fuses(B, H, W, Nf, ksize, dtype)) ? 1 : 0;
}

static int conv2d_impl(const void *P, const void *F, const float *bias, void *O, float *ws, size_t ws_bytes, int B, int H,
                       int W, int Cp, int ldp, int Nf, int ldo, int ksize, int dtype, void *stream, const char *fn,
                       const float *bn_shift = nullptr, float *bn_part = nullptr, float act_alpha = 1.0f, const Y2BnBwd *bwd = nullptr,
                       float *dgamma = nullptr, float *dbeta = nullptr, double *red_ws = nullptr, int *pending = nullptr) {
    if (!(P && F && O) || !(B > 0 && H > 0 && W > 0 && Cp > 0 && Nf > 0) || !(ksize == 1 || ksize == 3) || !(ldp >= Cp && ldo >= Nf)) {
        yolo2_set_error("%s: argument check failed: pointers / extents / ksize / strides", fn);
        return YOLO2_E_ARG;
    }
    const int vec = dtype == YOLO2_BF16 ? 8 : 4;
    const size_t esz = dtype == YOLO2_BF16 ? 2 : 4;
    // the DMA path addresses both operands with 32-bit byte offsets below 2^31
    // (only the two DMA-read operands; the output is addressed with 64-bit pointers)
    if ((size_t)B * H * W * (size_t)ldp * esz >= (1ull << 31) || (size_t)Nf * ksize * ksize * Cp * esz >= (1ull << 31) || (size_t)B * H * W >= (1ull << 31) ||
        Cp % vec || ldp % vec || ((uintptr_t)P & 15) || ((uintptr_t)F & 15)) {
        yolo2_set_error("%s: argument check failed: operand >= 2 GiB or misaligned (channels must be a multiple of %d)", fn, vec);
        return YOLO2_E_ARG;
    }
    int rc = 0;
    static const bool first_direct = y2_env_int("YOLO2_FIRST_DIRECT", 1) != 0;
    if (first_direct && !bias && !bwd && act_alpha == 1.0f && y2_first_layer_shape(Cp, ldp, Nf, ldo, ksize)) {      // image layer: direct kernel (conv_first.hip)
        if (dtype != YOLO2_F32 && dtype != YOLO2_BF16) { yolo2_set_error("%s: bad dtype %d", fn, dtype); return YOLO2_E_ARG; }
        y2_first_layer_fwd(P, F, O, B, H, W, dtype, (hipStream_t)stream, bn_shift, bn_part);
        for (int i = 0; i < 8; ++i) g_last_plan[i] = -1;      // direct first-layer kernel (conv_first.hip)
        g_last_stat_rows = Y2_BN_PART_ROWS;
        Y2_CHECK_LAUNCH();
        return YOLO2_OK;
    }
    static const bool c32_direct = y2_env_int("YOLO2_C32", 1) != 0;
    if (c32_direct && !bwd && y2_c32_shape(Cp, ldp, Nf, ldo, ksize, dtype) && ((uintptr_t)O & 15) == 0) {      // 32 -> 64 channels (conv1): persistent kernel (conv_c32.hip)
        const Tune tu = tune_now();
        int rows = 0;
        if (y2_c32_fwd(P, F, O, B, H, W, bias, act_alpha, bn_shift, bn_part, tu.cus, &rows, (hipStream_t)stream) == 0) {
            const int plan_[8] = {512, 64, 8, 8, 9, 0, rows, 1};
            for (int i = 0; i < 8; ++i) g_last_plan[i] = plan_[i];
            if (bn_part) g_last_stat_rows = rows;
            Y2_CHECK_LAUNCH();
            return YOLO2_OK;
        }
    }
    static const bool c64_direct = y2_env_int("YOLO2_C64", 1) != 0;
    if (c64_direct && !bwd && y2_c64_shape(Cp, ldp, Nf, ldo, ksize, dtype) && ((uintptr_t)O & 15) == 0 && ((uintptr_t)F & 15) == 0) {      // 64 -> 128 channels (conv2 / conv4), 64 -> 32 (conv1's data gradient), 128 -> 64 (conv2 / conv4's data gradients): filters in registers (conv_c64.hip)
        const Tune tu = tune_now();
        int rows = 0;
        if (y2_c64_fwd(P, F, O, B, H, W, Nf, bias, act_alpha, bn_shift, bn_part, tu.cus, &rows, (hipStream_t)stream) == 0) {
            const int plan_[8] = {Nf == 64 ? 128 : 256, Nf, 8, 8, 9, 0, rows, 1};      // (positions per tile, filters)
            for (int i = 0; i < 8; ++i) g_last_plan[i] = plan_[i];
            if (bn_part) g_last_stat_rows = rows;
            Y2_CHECK_LAUNCH();
            return YOLO2_OK;
        }
    }
    bool stats_done = true;
    if (bwd) {      // data gradient + the producer layer's BN/leaky backward sums (yolo2_conv2d_dgrad_bn)
        static const bool d1_direct = y2_env_int("YOLO2_D1", 1) != 0;
        const bool can = y2_dgrad_bn_fuses(B, H, W, Nf, ksize, dtype);
        if (can && d1_direct && bn_part && y2_d1_shape(Cp, ldp, Nf, ldo, ksize, dtype, (long)B * H * W) && ((uintptr_t)O & 15) == 0 && ((uintptr_t)bwd->Y & 15) == 0) {
            // 1x1 data gradient of the wide early stages (conv3 / conv6): persistent kernel with prefetched y vectors and launch-long sums (conv_d1.hip)
            const Tune tu = tune_now();
            int rows = 0;
            if (y2_d1_dgrad_bn(P, F, O, (long)B * H * W, Cp, Nf, bn_part, *bwd, tu.cus, &rows, (hipStream_t)stream) == 0) {
                const int plan_[8] = {128, 128, 8, Cp / 8, 1, 0x100, rows, Nf / 128};
                for (int i = 0; i < 8; ++i) g_last_plan[i] = plan_[i];
                g_last_stat_rows = rows;
                Y2_CHECK_LAUNCH();
                if (pending) *pending = 1;
                return pending ? YOLO2_OK : y2_bn_part_to_grads(bn_part, Nf, dgamma, dbeta, (hipStream_t)stream);
            }
        }
        if (can) {
            Y2_DISPATCH_DTYPE(dtype, rc = launch_conv<T>(P, F, bias, O, ws, ws_bytes, B, H, W, Cp, ldp, Nf, ldo, ksize, (hipStream_t)stream,
                                                         nullptr, bn_part, &stats_done, act_alpha, *bwd));
        } else {
            stats_done = false;
            Y2_DISPATCH_DTYPE(dtype, rc = launch_conv<T>(P, F, bias, O, ws, ws_bytes, B, H, W, Cp, ldp, Nf, ldo, ksize, (hipStream_t)stream,
                                                         nullptr, nullptr, &stats_done, act_alpha));
            stats_done = false;
        }
        if (rc) { yolo2_set_error("%s: workspace memset failed", fn); return YOLO2_E_LAUNCH; }
        Y2_CHECK_LAUNCH();
        g_last_plan[5] |= stats_done ? 0x100 : 0;      // plan word 5 (split mode): bit 8 = the sums came from the epilogue
        if (pending) *pending = stats_done ? 1 : 0;
        if (stats_done) return pending ? YOLO2_OK : y2_bn_part_to_grads(bn_part, Nf, dgamma, dbeta, (hipStream_t)stream);
        return yolo2_bn_leaky_bwd_reduce(O, ldo, bwd->Y, bwd->mean, bwd->var, bwd->gamma, bwd->beta, dgamma, dbeta, red_ws, (long)B * H * W, Nf,
                                         bwd->eps, bwd->alpha, dtype, stream);
    }
    Y2_DISPATCH_DTYPE(dtype, rc = launch_conv<T>(P, F, bias, O, ws, ws_bytes, B, H, W, Cp, ldp, Nf, ldo, ksize, (hipStream_t)stream,
                                                 bn_shift, bn_part, &stats_done, act_alpha));
    if (rc) { yolo2_set_error("%s: workspace memset failed", fn); return YOLO2_E_LAUNCH; }
    Y2_CHECK_LAUNCH();
    if (bn_part && !stats_done)       // K-sliced path: statistics from a pass over the finished output
        return y2_colsum_into(O, ldo, (long)B * H * W, Nf, bn_shift, bn_part, dtype, (hipStream_t)stream, &g_last_stat_rows);
    return YOLO2_OK;
}

extern "C" size_t yolo2_conv2d_workspace_bytes(int B, int H, int W, int Cp, int Nf, int ksize, int dtype) {
    // the largest scratch any variant of yolo2_conv2d_ws / _bn / _bias_leaky can use for this shape: one f32 256x128 tile slot
    // per CU (stream-K) or the f32 [M][Nf] partial-sum image (K-sliced grids); smaller buffers only disable variants
    if (!(B > 0 && H > 0 && W > 0 && Cp > 0 && Nf > 0) || !(dtype == YOLO2_F32 || dtype == YOLO2_BF16)) return 0;
    const Tune &tu = tune();
    const size_t M = (size_t)B * H * W;
    const int vec = dtype == YOLO2_BF16 ? 8 : 4;
    size_t need = (size_t)tu.cus * 256 * 128 * sizeof(float);
    const long tiles = (long)cdiv((long)M, 128) * cdiv(Nf, 128);
    if (choose_ksplit((int)(tiles > (1 << 30) ? (1 << 30) : tiles), ksize * ksize * cdiv(Cp, 4 * vec), tu.target_blocks) > 1 && M * Nf * sizeof(float) > need)
        need = M * Nf * sizeof(float);
    return need;
}

extern "C" int yolo2_debug_last_conv_plan(int *out8) {
    if (!out8) return YOLO2_E_ARG;
    for (int i = 0; i < 8; ++i) out8[i] = g_last_plan[i];
    return YOLO2_OK;
}

extern "C" int yolo2_conv2d(const void *P, const void *F, const float *bias, void *O, int B, int H,
                            int W, int Cp, int ldp, int Nf, int ldo, int ksize, int dtype, void *stream) {
    return conv2d_impl(P, F, bias, O, nullptr, 0, B, H, W, Cp, ldp, Nf, ldo, ksize, dtype, stream, "yolo2_conv2d");
}

extern "C" int yolo2_conv2d_bn(const void *P, const void *F, void *O, float *ws, size_t ws_bytes, int B, int H, int W, int Cp, int ldp,
                               int Nf, int ldo, int ksize, const float *shift, float *bn_part, int dtype, void *stream) {
    if (!shift || !bn_part || ldo != Nf) {
        yolo2_set_error("yolo2_conv2d_bn: argument check failed: shift / bn_part NULL or ldo != Nf");
        return YOLO2_E_ARG;
    }
    return conv2d_impl(P, F, nullptr, O, ws, ws_bytes, B, H, W, Cp, ldp, Nf, ldo, ksize, dtype, stream, "yolo2_conv2d_bn", shift, bn_part);
}

extern "C" int yolo2_conv2d_dgrad_bn(const void *dY, const void *F, void *dX, float *ws, size_t ws_bytes, int B, int H, int W, int Cp, int ldp,
                                     int Nf, int ldo, int ksize, const void *Yprev, const float *mean, const float *var, const float *gamma,
                                     const float *beta, float *dgamma, float *dbeta, float *bn_part, double *red_ws, float eps, float alpha,
                                     int *pending, int dtype, void *stream) {
    if (!Yprev || !mean || !var || !gamma || !beta || !dgamma || !dbeta || !bn_part || !red_ws) {
        yolo2_set_error("yolo2_conv2d_dgrad_bn: argument check failed: a producer-layer pointer is NULL");
        return YOLO2_E_ARG;
    }
    const Y2BnBwd bz{Yprev, mean, var, gamma, beta, eps, alpha, 0};
    return conv2d_impl(dY, F, nullptr, dX, ws, ws_bytes, B, H, W, Cp, ldp, Nf, ldo, ksize, dtype, stream, "yolo2_conv2d_dgrad_bn", nullptr, bn_part, 1.0f,
                       &bz, dgamma, dbeta, red_ws, pending);
}
extern "C" int yolo2_bn_part_to_grads(float *bn_part, int C, float *dgamma, float *dbeta, void *stream) {
    if (!bn_part || !dgamma || !dbeta || C <= 0) { yolo2_set_error("yolo2_bn_part_to_grads: argument check failed"); return YOLO2_E_ARG; }
    return y2_bn_part_to_grads(bn_part, C, dgamma, dbeta, (hipStream_t)stream);
}

extern "C" int yolo2_conv2d_bias_leaky(const void *P, const void *F, const float *bias, void *O, float *ws, size_t ws_bytes, int B, int H, int W,
                                       int Cp, int ldp, int Nf, int ldo, int ksize, float alpha, int dtype, void *stream) {
    if (!bias) { yolo2_set_error("yolo2_conv2d_bias_leaky: bias is NULL"); return YOLO2_E_ARG; }
    return conv2d_impl(P, F, bias, O, ws, ws_bytes, B, H, W, Cp, ldp, Nf, ldo, ksize, dtype, stream, "yolo2_conv2d_bias_leaky", nullptr, nullptr, alpha);
}

extern "C" int yolo2_conv2d_ws(const void *P, const void *F, const float *bias, void *O, float *ws, size_t ws_bytes, int B,
                               int H, int W, int Cp, int ldp, int Nf, int ldo, int ksize, int dtype, void *stream) {
    return conv2d_impl(P, F, bias, O, ws, ws_bytes, B, H, W, Cp, ldp, Nf, ldo, ksize, dtype, stream, "yolo2_conv2d_ws");
}
