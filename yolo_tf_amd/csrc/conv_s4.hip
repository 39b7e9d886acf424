// Loader / consumer tap-fused 3x3 implicit GEMM (bf16, 256 x 128 tile): four COMPUTE waves of 128 x 64 (one per SIMD) + four LOADER waves.  Round 6.
//
// Replaces slim.layers.conv2d for the 3x3 layers with >= 64-channel-multiple inputs on images up to 55 wide (reference
// model/yolo2/inference.py:81-117) and its tf.gradients input gradient (train.py:127-129): same operand layouts, halo image, filter ring, stream-K
// decomposition, hand-off and epilogue contracts as conv3x3_pp_kernel (conv_pp.hip).  What changes is the division of labour inside the workgroup.
//
// Why (profiles/r06_pp_phase_cycles_abl.txt, r06_mfma_lds_loop.txt): in the ping-pong kernel a wave's LOAD phase -- 16 ds_read_b128 + 3 LDS-DMA
// instructions -- takes 530-560 cycles against the 512 cycles of its partner's 16 MFMAs, so the matrix pipe waits at every barrier: 1340 cycles per K step
// for 1024 of MFMA work.  The stamps say where the LOAD phase goes: ~45 cycles of ISSUE per DMA instruction (the memory system pushing back: without the
// DMA the same loop runs 1116 cycles per step) and 128 KiB of fragment reads per step (1 KiB per MFMA for 64 x 64 per wave).  Here
//   * waves 0-3 (one per SIMD) COMPUTE 128 x 64 each: 6 fragment reads per 8 MFMAs (768 B per MFMA, 96 KiB per step), each read placed behind one MFMA of
//     the previous 16-k group (two fragment sets) -- measured 1037 cycles per step in isolation for this stream; they never issue a DMA instruction;
//   * waves 4-7 (one per SIMD) are LOADERS: per K step each issues 4 filter pieces of step s + D and 2 halo pieces of the next chunk, waits (counted vmcnt)
//     for its pieces of step s + 1 and meets the computing waves at the step's ONE barrier.  Memory push-back stalls a loader, not an MFMA stream.
//   Hazards (s = K step, stage = s mod NSB, D = NSB - 1):
//     RAW  the barrier B(s) sits, for a computing wave, between its 16-k groups 2 and 3 of step s; a loader arrives there only after its pieces of step
//          s + 1 have landed; the computing waves' first reads of step s + 1 (behind the MFMAs of group 3) come after B(s).
//     WAR  a computing wave retires ALL its reads of step s (lgkmcnt(0)) before B(s); the loader's pieces of step s + D + 1 -- the stage of step s -- are
//          issued after B(s).  The halo buffer of chunk c + 1 was last read in chunk c - 1 and is refilled from tap 0 of chunk c on.
//   * epilogue: the computing waves round their accumulators into the per-wave LDS image (+ forward statistics); ALL eight waves then store the image
//     rows (and reduce the BN-backward sums) -- the loaders take half of every image.
#include "common.h"
#include "conv_shared.h"
#include <type_traits>

#define Y2S_BM 256
#define Y2S_BN 128
#define Y2S_BBYTES (Y2S_BN * 128)
#define Y2S_FPL 4                          // filter pieces per loader and K step (16 KiB stage / 4 loaders)
#define Y2S_HPL 2                          // halo pieces per loader and K step
#define Y2S_LOADS (Y2S_FPL + Y2S_HPL)      // DMA instructions per loader and K step (idle ones write zeros into the sink KiB)

#ifdef Y2S_EXPERIMENTS
#define Y2S_PHASES 16
__device__ unsigned long long y2s_phase[1024 * 2 * Y2S_PHASES];
extern "C" int yolo2_debug_s4_phases(void *host_out) { return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(y2s_phase), sizeof(unsigned long long) * 1024 * 2 * Y2S_PHASES); }
#define Y2S_STAMP(k) do { if (A_PHASES) { const unsigned long long t_ = wall_clock64(); ph_[k] += t_ - ph_tl; ph_tl = t_; } } while (0)
#else
#define Y2S_STAMP(k) do { } while (0)
#endif

// HROWS = halo rows held (>= 256 + 2 W + 2, multiple of 8), NSB = filter ring depth (the loaders run NSB - 1 steps ahead of the reads).
// ABL (experiments build): +1 no MFMAs, +2 no fragment reads, +4 no DMA inside the K loop, +8 no epilogue, +16 no hand-off traffic, +32 wall-clock phase sums
template <bool BNBWD, int HROWS, int NSB, int ABL>
__global__ __launch_bounds__(512) void conv3x3_s4_kernel(
    const bf16 *__restrict__ P, unsigned p_bytes, const bf16 *__restrict__ F, unsigned f_bytes, const float *__restrict__ bias,
    bf16 *__restrict__ O, float *__restrict__ slots, int H, int W, int Cp, int ldp, int Nf, int ldo, int M, int NT,
    const float *__restrict__ bn_shift, float *__restrict__ bn_part, unsigned *__restrict__ flags, float act_alpha, const Y2BnBwd bz, int k_rotate) {
    typedef bf16 T;
    constexpr int BM = Y2S_BM, BN = Y2S_BN, NCW = 4, NLW = 4, WGN = 2, WGM = 2, TM = 4, TN = 2, VEC = 8, ROWB = 128, TAPS = 9;
    constexpr int HBYTES = HROWS * 128, HB = HBYTES + 1024, RING = 2 * HB, D = NSB - 1;
    constexpr int HPIECES = HROWS / 8, HSLOTS = (HPIECES + NLW * Y2S_HPL - 1) / (NLW * Y2S_HPL);
    constexpr bool A_NOMFMA = (ABL & 1) != 0, A_NOREAD = (ABL & 2) != 0, A_NODMA = (ABL & 4) != 0, A_NOEPI = (ABL & 8) != 0, A_NOHANDOFF = (ABL & 16) != 0;
    constexpr bool A_PHASES = (ABL & 32) != 0;
    constexpr bool A_NOLGKM = (ABL & 64) != 0;      // (timing only, unsafe: no lgkmcnt(0) in front of the step barrier)
    // halo pieces of chunk c+1 ride in the slots of taps 0 .. HSLOTS-1 of chunk c and must be covered by the loaders' wait at the end of tap 8
    static_assert(HROWS % 8 == 0 && HSLOTS <= TAPS + 1 - D && D >= 2 && RING + NSB * Y2S_BBYTES <= 160 * 1024, "LDS plan");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[RING + NSB * Y2S_BBYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= NCW;
    const int MT = (M + BM - 1) / BM;
    const int nk = (Cp / 64) * TAPS;                     // K steps per tile: (64-channel chunk, tap)
    const long su_total = (long)MT * NT * nk;
    const int G = gridDim.x;
    const int wx = (G & 7) ? (int)blockIdx.x : (int)((blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3));
    auto share_end = [&](int w) { return (long)(w + 1) * su_total / G; };
    long su = wx * su_total / G;
    const long su_end = share_end(wx);
    // K rotation of the stream-K tiles (conv3x3_pp_kernel; profiles/r03_l2_stationary_ab.md): chunk c of tile t lives at memory chunk (c + rot_t) mod nch
    const int nch = Cp / 64;
    const double share = (double)su_total / (double)G;
    const int shares_per_tile = (int)((double)nk / share + 0.5);
    const double drift_chunks = k_rotate && shares_per_tile >= 2 ? ((double)shares_per_tile * share - (double)nk) / (double)TAPS : 0.0;

    const __amdgpu_buffer_rsrc_t rsrcP = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(P), 0, p_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcF = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(F), 0, f_bytes, 0x00020000);
    const double rcp_hw = 1.0 / (double)(H * W);
    const float rcp_w = 1.0f / (float)W;
    unsigned char *const zero0 = smem + HBYTES;         // the zero KiB of halo buffer 0
    unsigned char *const sink = smem + HB + HBYTES;     // ... of halo buffer 1: also the sink of idle DMA slots (behind the tile image of the epilogue)
    typedef __attribute__((address_space(3))) void *lds_void_ptr;
    typedef const __attribute__((address_space(3))) bf16x8 *lds_frag_ptr;

  bool park_pending = false;
#ifdef Y2S_EXPERIMENTS
  unsigned long long ph_[Y2S_PHASES] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ph_tl = 0;
  if (A_PHASES) { ph_tl = wall_clock64(); ph_[10] = ph_tl; }
#endif
  for (bool first_seg = true;; first_seg = false) {
    if (su >= su_end) break;
    const int t = (int)(su / nk);
    const int kt_beg = (int)(su - (long)t * nk);
    const int kt_end = (int)min((long)nk, kt_beg + (su_end - su));
    su += kt_end - kt_beg;
    const int nt = t / MT, mt = t - nt * MT;
    if (!first_seg) __syncthreads();                     // every wave is done with the previous segment's LDS (tile image of its epilogue included) and its DMA / stores
    const int m0 = mt * BM, n0 = nt * BN;
    int rot;                                             // this tile's chunk rotation, 0 <= rot < nch
    {
        const long r = (long)__builtin_floor(-(double)t * drift_chunks + 0.5);
        rot = (int)(r % nch);
        if (rot < 0) rot += nch;
        rot = __builtin_amdgcn_readfirstlane(rot);
    }
    auto mem_chunk = [&](int c) { const int m = c + rot; return m >= nch ? m - nch : m; };      // logical chunk (0 <= c <= nch) -> chunk in memory
    const int c_first = kt_beg / TAPS;

    // per-segment copy of the lane id the compiler cannot see through (conv_pp.hip: nothing a segment precomputes per lane stays live across another's epilogue)
    int lane_s = lane;
    asm volatile("" : "+v"(lane_s));

    f32x16 acc[TM][TN];      // (zeroed for every wave: left undefined in the loaders it becomes a value carried around the segment loop, and hipcc spills it)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if (loader) {
        // =============================================== LOADER wave l: every DMA instruction of the segment ===============================================
        const int l = wave - NCW;
        // halo: piece p = rows 8 p .. 8 p + 7; slot j of this loader carries pieces (j * NLW + l) * HPL + {0, 1}.  Rows before pixel 0 wrap to offsets
        // >= 2^31 and rows past the last pixel lie beyond num_records: both read as zeros.  Source-side swizzle ((row >> 1) & 7).
        const int hrl = lane_s >> 3;                     // row inside a piece
        auto issue_halo = [&](int slot, int hc, int hc_mem, bool hreal) {
#pragma unroll
            for (int q = 0; q < Y2S_HPL; ++q) {
                const int piece = (slot * NLW + l) * Y2S_HPL + q;
                const bool real = hreal && piece < HPIECES;
                unsigned char *dst = real ? smem + (hc & 1) * HB + piece * 1024 : sink;
                const int hr = piece * 8 + hrl;
                const unsigned voff = (unsigned)(m0 - (W + 1) + hr) * (unsigned)ldp * 2u + (unsigned)(((lane_s & 7) ^ ((hr >> 1) & 7)) * 16) + (unsigned)hc_mem * 128u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcP, (lds_void_ptr)dst, 16, real ? voff : Y2_OOB, 0, 0, 0);
            }
        };
        // filter: stage = 128 rows x 128 B = 16 pieces; this loader's pieces are l * 4 .. l * 4 + 3 (rows 32 l .. 32 l + 31); filters >= Nf lie beyond num_records
        unsigned b_voff[Y2S_FPL];
#pragma unroll
        for (int i = 0; i < Y2S_FPL; ++i) {
            const int r = (l * Y2S_FPL + i) * 8 + hrl;
            b_voff[i] = (unsigned)(n0 + r) * (unsigned)(TAPS * Cp) * 2u + (unsigned)(((lane_s & 7) ^ ((r >> 1) & 7)) * 16);
        }
        auto issue_filter = [&](int kb, unsigned offB, int bstage) {
            const bool breal = kb < kt_end;
            unsigned char *Bs = smem + RING + bstage * Y2S_BBYTES + l * (Y2S_FPL * 1024);
#pragma unroll
            for (int i = 0; i < Y2S_FPL; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcF, (lds_void_ptr)(Bs + i * 1024), 16, breal ? b_voff[i] + offB : Y2_OOB, 0, 0, 0);
        };
        auto filt_off = [&](int kb) {                    // byte offset of K step kb inside a filter row (run-time form: prologue only)
            const int bc = kb / TAPS, bt = kb - bc * TAPS;
            return (unsigned)(mem_chunk(bc) * TAPS * 64 + bt * 64) * 2u;
        };
        {   // prologue: both zero KiB (the previous segment's epilogue staged its tile image over the first), the whole halo of the first chunk (+ the pieces of
            // the next chunk whose slots this segment starts behind), then the filter tiles of steps kt_beg .. kt_beg + D - 1 into ring stages 0 .. D-1
            if (l == 0) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcP, (lds_void_ptr)zero0, 16, Y2_OOB, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcP, (lds_void_ptr)sink, 16, Y2_OOB, 0, 0, 0);
            }
            const int c_tap = kt_beg - c_first * TAPS;
#pragma unroll
            for (int j = 0; j < HSLOTS; ++j) issue_halo(j, c_first, mem_chunk(c_first), true);
#pragma unroll
            for (int j = 0; j < HSLOTS; ++j) issue_halo(j, c_first + 1, mem_chunk(c_first + 1), j < c_tap && (c_first + 1) * TAPS < kt_end);
            // (prologue slots carry Y2S_LOADS instructions, filter pieces first, like the steady-state slots the counted waits of the first steps assume)
#pragma unroll
            for (int d = 0; d < D; ++d) {
                issue_filter(kt_beg + d, filt_off(kt_beg + d), d);
                issue_halo(0, 0, 0, false);
            }
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * Y2S_LOADS) : "memory");      // zero KiBs, halo and filter tile kt_beg have landed (this loader's; the newer slots stay in flight)
        __builtin_amdgcn_s_barrier();                    // P: the segment's first operands are in LDS (all loaders)
        Y2S_STAMP(0);
        int kt = kt_beg, stage_i = D;
        for (int c = c_first; c * TAPS < kt_end; ++c) {
            const int lo = max(0, kt_beg - c * TAPS), hi = min(TAPS, kt_end - c * TAPS);       // taps of this chunk inside the segment
            const int mc0 = mem_chunk(c), mc1 = mem_chunk(c + 1);
            const bool next_ok = (c + 1) * TAPS < kt_end;
            auto step = [&](auto tap_tag) {
                constexpr int tp = decltype(tap_tag)::value;
                if (tp >= lo && tp < hi) {
                    constexpr int bt = (tp + D) % TAPS;
                    const unsigned offB = (unsigned)(((tp + D >= TAPS) ? mc1 : mc0) * TAPS * 64 + bt * 64) * 2u;
                    if (!A_NODMA) {
                        issue_filter(kt + D, offB, stage_i);
                        issue_halo(tp, c + 1, mc1, tp < HSLOTS && next_ok);
                        // this loader's pieces of step kt + 1 have landed (the newest D - 1 slots stay in flight)
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * Y2S_LOADS) : "memory");
                    }
                    __builtin_amdgcn_s_barrier();        // B(kt)
                    ++kt;
                    stage_i = stage_i == NSB - 1 ? 0 : stage_i + 1;
                }
            };
            step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
            step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
            step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{}); step(std::integral_constant<int, 8>{});
        }
        Y2S_STAMP(1);
    } else {
        // =============================================== COMPUTE wave: 128 x 64 of the tile, registers and LDS reads only ===============================================
        const int wm = wave / WGN, wn = wave % WGN;
        const int frow_s = lane_s & 31;
        // ---- per-lane index arithmetic (f64 reciprocal pixel decode, 36 read addresses), under the latency of the loaders' prologue
        unsigned amask[TM];                              // 9 bits per fragment row: which of the nine taps of the row's pixel lie inside the image
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * (TM * 32) + i * 32 + frow_s;
            unsigned mask = 0;
            if (m < M) {
                const int HW = H * W;
                int bq = (int)((double)m * rcp_hw);
                int rem = m - bq * HW;
                if (rem < 0) rem += HW; else if (rem >= HW) rem -= HW;
                int h = (int)((float)rem * rcp_w);
                int w = rem - h * W;
                if (w < 0) { w += W; --h; } else if (w >= W) { w -= W; ++h; }
                const unsigned cm = (w > 0 ? 1u : 0u) | 2u | (w < W - 1 ? 4u : 0u);
                mask = (h > 0 ? cm : 0u) | (cm << 3) | (h < H - 1 ? cm << 6 : 0u);
            }
            amask[i] = mask;
        }
        // ---- LDS read addresses.  A: one per (tap, fragment row), valid for the halo buffer of the current chunk parity; the 16-k group kk is XORed in
        // (bits 5-6 come from the swizzle term alone: every other summand is a multiple of 128).  B: one per 16-k group.
        const unsigned lds0 = y2_lds_addr(smem);
        const unsigned hi16 = (unsigned)(lane_s >> 5) << 4;
        unsigned aaddr[TAPS][TM];
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp) {
            const int dh = tp / 3 - 1, dw = tp % 3 - 1;
            const unsigned toffb = (unsigned)(((W + 1) + dh * W + dw) * ROWB);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const unsigned hb = (unsigned)((wm * (TM * 32) + i * 32 + frow_s) * ROWB) + toffb;      // halo row * 128
                const bool ok = ((amask[i] >> tp) & 1u) != 0u;
                const unsigned sw = ((hb >> 4) & 0x70u) ^ hi16;           // ((row >> 1) & 7) << 4, folded with this lane's half of the k group
                // masked (pixel, tap): the buffer's zero KiB, at the same offset inside a 256-byte bank line as the real row
                aaddr[tp][i] = lds0 + (unsigned)((c_first & 1) * HB) + (ok ? hb : (unsigned)HBYTES + (hb & 0x80u)) + sw;
            }
        }
        unsigned baddr[4];
        {
            const unsigned brow = lds0 + (unsigned)(RING + (wn * TN * 32 + frow_s) * ROWB);
            const unsigned bx = hi16 ^ ((unsigned)((frow_s >> 1) & 7) << 4);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) baddr[kk] = brow + (bx ^ (unsigned)(kk * 32));
        }
        if (park_pending) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the parked partial tile of the previous (tail) segment: write-through stores acknowledged
        __builtin_amdgcn_s_barrier();                    // P
        if (park_pending) {
            if (tid == 0) __hip_atomic_store(flags + wx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            park_pending = false;
        }
        Y2S_STAMP(0);

        bf16x8 fa[2][TM], fb[2][TN];
        if (A_NOREAD) {
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int e = 0; e < 8; ++e) fa[s_][i][e] = (bf16)(float)(lane_s + e);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e) fb[s_][j][e] = (bf16)(float)(lane_s - e);
            }
        }
        int kt = kt_beg, stage_r = 0;
        // one fragment read: n-th of group (tap tp, 16-k group kk) of ring stage `so` into set `set` (n < TM: pixel row n; else filter row n - TM)
        auto frag_read = [&](int n, const unsigned (&arow)[TM], int kk, unsigned so, int set) {
            if (A_NOREAD) return;
            if (n < TM) fa[set][n] = *(lds_frag_ptr)(uintptr_t)(arow[n] ^ (unsigned)(kk * 32));
            else fb[set][n - TM] = *(lds_frag_ptr)(uintptr_t)(baddr[kk] + so + (unsigned)((n - TM) * 32 * ROWB));
        };
        // eight MFMAs of one 16-k group on fragment set `set`, one read of the NEXT group (arow_n / kk_n / so_n -> set ^ 1) behind each of the first six
        auto group = [&](int set, const unsigned (&arow_n)[TM], int kk_n, unsigned so_n, bool reads) {
            int n = 0;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (!A_NOMFMA) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[set][i], fb[set][j], acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (reads && n < TM + TN) frag_read(n, arow_n, kk_n, so_n, set ^ 1);
                    __builtin_amdgcn_sched_barrier(0);
                    ++n;
                }
        };
        // the segment's first fragments: group 0 of step kt_beg into set 0
        {
            const int tp0 = kt_beg - c_first * TAPS;
            // (run-time tap index: the nine candidates through a uniform switch -- executed once per segment)
            unsigned arow0[TM];
#pragma unroll
            for (int tp = 0; tp < TAPS; ++tp)
                if (tp == tp0) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) arow0[i] = aaddr[tp][i];
                }
#pragma unroll
            for (int n = 0; n < TM + TN; ++n) frag_read(n, arow0, 0, 0u, 0);
        }
        for (int c = c_first; c * TAPS < kt_end; ++c) {
            const int lo = max(0, kt_beg - c * TAPS), hi = min(TAPS, kt_end - c * TAPS);
            const unsigned delta = (c & 1) ? (unsigned)(-HB) : (unsigned)HB;                   // to the other halo buffer
            auto step = [&](auto tap_tag) {
                constexpr int tp = decltype(tap_tag)::value;
                constexpr int tpn = (tp + 1) % TAPS;
                if (tp >= lo && tp < hi) {
                    const unsigned so = (unsigned)(stage_r * Y2S_BBYTES);
                    const int stage_n = stage_r == NSB - 1 ? 0 : stage_r + 1;
                    const unsigned so_n = (unsigned)(stage_n * Y2S_BBYTES);
                    group(0, aaddr[tp], 1, so, true);
                    group(1, aaddr[tp], 2, so, true);
                    group(0, aaddr[tp], 3, so, true);
                    // every read of step kt has returned (group 3's fragments included): after B(kt) the loaders may overwrite its ring stage, and -- after
                    // the last tap of a chunk -- its halo buffer
                    if (!A_NOLGKM) __builtin_amdgcn_s_waitcnt(0xc07f);
                    __builtin_amdgcn_s_barrier();        // B(kt): step kt + 1 is in LDS
                    __builtin_amdgcn_sched_barrier(0);
                    // group 3, with the first fragments of step kt + 1 behind it: the next tap of this chunk (its addresses have not moved yet), or tap 0 of
                    // the next chunk (aaddr[0] was moved to the other halo buffer when this chunk's tap 0 ended)
                    group(1, aaddr[tpn], 0, so_n, kt + 1 < kt_end);
                    ++kt;
                    stage_r = stage_n;
                }
                // the next chunk reads the other halo buffer (every tap's addresses move, whether or not this segment ran the tap)
#pragma unroll
                for (int i = 0; i < TM; ++i) aaddr[tp][i] += delta;
            };
            step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
            step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
            step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{}); step(std::integral_constant<int, 8>{});
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        Y2S_STAMP(1);
#ifdef Y2S_EXPERIMENTS
        if (A_PHASES) ph_[9] += 1;
#endif
    }

    // Everything below indexes by (lane_e, wave_e): copies the compiler cannot see through, so that none of the hand-off / epilogue address arithmetic
    // is hoisted above the K loop (conv_pp.hip).
    int lane_e = lane, wave_e = wave;
    asm volatile("" : "+v"(lane_e), "+s"(wave_e));
    const bool comp_e = wave_e < NCW;
    const int cw_e = wave_e & (NCW - 1);                 // compute-wave index (a loader helps with the image of compute wave wave_e - 4)
    const int wm_e = cw_e / WGN, wn_e = cw_e % WGN;
    // ---- stream-K hand-off (as in conv3x3_pp_kernel: the workgroup holding K step 0 of a tile owns it; a tail segment is parked)
    {
        constexpr int SLOT = BM * BN;
        const __amdgpu_buffer_rsrc_t rsrcS = __builtin_amdgcn_make_buffer_rsrc(slots, 0, (unsigned)((size_t)gridDim.x * SLOT * sizeof(float)), 0x00020000);
        const unsigned slot_lane = (unsigned)(((size_t)cw_e * (TM * TN * 16 * 64) + (size_t)lane_e * 4) * sizeof(float));
        if (A_NOHANDOFF) { if (kt_beg > 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); continue; } }
        else if (kt_beg > 0) {
            if (comp_e) {
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
            }
            // published -- stores acknowledged, flag raised -- from inside the next segment's prologue, or after the loop when this was the workgroup's only segment
            park_pending = true;
            Y2S_STAMP(2);
            continue;
        }
        if (!A_NOHANDOFF && kt_end < nk) {
            const long tile_end = su - kt_end + nk;
            // every partner's flag at once: lane i of wave 0 waits for (and clears) the flag of partner wx + 1 + i
            int np_ = 0;
            for (long cov = su; cov < tile_end; ++np_) cov = share_end(wx + 1 + np_);
            if (wave_e == 0)
                for (int i = lane_e; i < np_; i += 64) y2_sk_wait_and_clear(flags, wx + 1 + i);      // (bounded: conv_shared.h)
            __syncthreads();
            Y2S_STAMP(3);
            if (comp_e) {
                for (int p = wx + 1; p <= wx + np_; ++p) {
                    const unsigned theirs = (unsigned)((size_t)p * SLOT * sizeof(float)) + slot_lane;
#pragma unroll
                    for (int ih = 0; ih < TM; ih += 2) {      // (two halves: 16 loads of 16 bytes in flight per lane)
#pragma unroll
                        for (int i = ih; i < ih + 2; ++i)
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
                    }
                }
            }
            Y2S_STAMP(4);
        }
    }

    if (A_NOEPI) {       // (ablation: keep the accumulators alive, store nothing)
        if (comp_e && acc[0][0][0] == 123.456f && acc[1][1][5] == 1.0f) O[0] = (bf16)acc[0][1][3];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        continue;
    }
    // ---- epilogue: the tile rounded into per-compute-wave LDS images (128 rows x 64 filters, + the forward statistics of the rounded values), then 16-byte
    // stores of the image rows by ALL eight waves (a loader takes the second half of the rows of compute wave wave - 4's image), with the producer layer's
    // BN-backward sums taken from the stored values
    {
        const bool stats = !BNBWD && bn_part != nullptr;
        const bool bstats = BNBWD && bn_part != nullptr;
        const bool stats_unique = bz.stat_mask_inv == 0;       // the host found a row for every (pixel tile, 64-row group) pair
        constexpr int WROWS = TM * 32, WROWB = TN * 32 * 2, WSTRIDE = WROWB + 16, WCPR = WROWB / 16, HROWS_W = WROWS / 2, NIT = HROWS_W * WCPR / 64, YG = 4;
        static_assert(NCW * WROWS * WSTRIDE <= RING && NCW * WROWS * WSTRIDE <= HB + HBYTES, "tile image fits the halo buffers, in front of the idle-DMA sink");
        __syncthreads();                                      // every compute wave has finished reading the last step's operands (idle DMA slots write behind the image)
        Y2S_STAMP(5);
        unsigned char *wreg = smem + cw_e * (WROWS * WSTRIDE);
        if (comp_e) {
            float bvj[TN], shj[TN];
            bool nokj[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + (wn_e * TN + j) * 32 + (lane_e & 31);
                nokj[j] = n < Nf;
                bvj[j] = (bias && nokj[j]) ? bias[n] : 0.f;
                shj[j] = (stats && nokj[j]) ? bn_shift[n] : 0.f;
            }
            const bool tail = m0 + BM > M;
            auto stage_tile = [&](auto act_tag, auto tail_tag) {
                constexpr bool ACT = decltype(act_tag)::value, TAIL = decltype(tail_tag)::value;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float bv = bvj[j], sh = shj[j];
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {          // the statistics rows are per 64 pixel rows: (pixel tile, wm, half) -> one partial row
                        float s1 = 0.f, s2 = 0.f;
#pragma unroll
                        for (int i = 2 * hf; i < 2 * hf + 2; ++i) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int row = i * 32 + 4 * (lane_e >> 5) + (r & 3) + 8 * (r >> 2);
                                float v = acc[i][j][r] + bv;
                                if (ACT) v = fmaxf(v, act_alpha * v);
                                const T o = (T)v;
                                *reinterpret_cast<T *>(wreg + row * WSTRIDE + (j * 32 + (lane_e & 31)) * 2) = o;
                                if (stats && (!TAIL || m0 + wm_e * WROWS + row < M)) {
                                    const float d = (float)o - sh;
                                    s1 += d;
                                    s2 += d * d;
                                }
                            }
                        }
                        if (stats) {
                            s1 += __shfl_xor(s1, 32, 64);
                            s2 += __shfl_xor(s2, 32, 64);
                            if (lane_e < 32 && nokj[j]) {
                                const int n = n0 + (wn_e * TN + j) * 32 + (lane_e & 31);
                                const int slot = (mt * 4 + wm_e * 2 + hf) & ((Y2_BN_PART_ROWS - 1) ^ bz.stat_mask_inv);
                                float *p1 = bn_part + (long)slot * Nf + n, *p2 = bn_part + (long)(Y2_BN_PART_ROWS + slot) * Nf + n;
                                if (stats_unique) { *p1 = s1; *p2 = s2; }
                                else { unsafeAtomicAdd(p1, s1); unsafeAtomicAdd(p2, s2); }
                            }
                        }
                    }
                }
            };
            if (act_alpha != 1.0f) { if (tail) stage_tile(std::true_type{}, std::true_type{}); else stage_tile(std::true_type{}, std::false_type{}); }
            else if (tail) stage_tile(std::false_type{}, std::true_type{});
            else stage_tile(std::false_type{}, std::false_type{});
        }
        __syncthreads();                                      // the four images are complete
        Y2S_STAMP(6);
        // ---- store loop: this wave takes rows [hrow0, hrow0 + 64) of image cw_e
        const int hrow0 = comp_e ? 0 : HROWS_W;
        float cmu[VEC], cinv[VEC], cga[VEC], cbt[VEC], ps[2][VEC];
        Vec16<T> yv[YG];
        const int bz_nb = min(n0 + wn_e * TN * 32 + (lane_e % WCPR) * VEC, Nf - VEC);
        auto bz_load_y = [&](int it0) {
#pragma unroll
            for (int u = 0; u < YG; ++u) {
                const int m = min(m0 + wm_e * WROWS + hrow0 + ((it0 + u) * 64 + lane_e) / WCPR, M - 1);
                yv[u] = ld16(reinterpret_cast<const T *>(bz.Y) + (long)m * Nf + bz_nb);
            }
        };
        if (bstats) {
            bz_load_y(0);
            int nb_c = bz_nb;
            asm volatile("" : "+v"(nb_c));
#pragma unroll
            for (int k = 0; k < VEC; k += 4) {
                const f32x4 a = *reinterpret_cast<const f32x4 *>(bz.mean + nb_c + k), b = *reinterpret_cast<const f32x4 *>(bz.var + nb_c + k);
                const f32x4 c = *reinterpret_cast<const f32x4 *>(bz.gamma + nb_c + k), d = *reinterpret_cast<const f32x4 *>(bz.beta + nb_c + k);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    cmu[k + q] = a[q];
                    cinv[k + q] = 1.0f / sqrtf(b[q] + bz.eps);
                    cga[k + q] = c[q];
                    cbt[k + q] = d[q];
                    ps[0][k + q] = ps[1][k + q] = 0.f;
                }
            }
        }
        static_assert(NIT % YG == 0, "whole groups of y vectors");
#pragma unroll 1
        for (int g0 = 0; g0 < NIT; g0 += YG) {
            if (bstats && g0) bz_load_y(g0);
#pragma unroll
            for (int u = 0; u < YG; ++u) {
                const int id = (g0 + u) * 64 + lane_e;
                const int row = hrow0 + id / WCPR, ch = id % WCPR;
                const int m = m0 + wm_e * WROWS + row;
                const int n = n0 + wn_e * TN * 32 + ch * VEC;
                const f32x4 v = *reinterpret_cast<const f32x4 *>(wreg + row * WSTRIDE + ch * 16);
                if (m < M && n < Nf) {
                    *reinterpret_cast<f32x4 *>(O + (long)m * ldo + n) = v;
                    if (bstats) {
                        const Vec16<T> y = yv[u];
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
        }
        if (bstats) {
#pragma unroll
            for (int off = WCPR; off < 64; off <<= 1)
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    ps[0][k] += __shfl_xor(ps[0][k], off, 64);
                    ps[1][k] += __shfl_xor(ps[1][k], off, 64);
                }
            const int nb = n0 + wn_e * TN * 32 + lane_e * VEC;
            if (lane_e < WCPR && nb < Nf) {
                const int slot = (mt * 4 + wm_e * 2 + (comp_e ? 0 : 1)) & ((Y2_BN_PART_ROWS - 1) ^ bz.stat_mask_inv);
                float *p1 = bn_part + (long)slot * Nf + nb, *p2 = bn_part + (long)(Y2_BN_PART_ROWS + slot) * Nf + nb;
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    if (stats_unique) { p1[k] = ps[0][k]; p2[k] = ps[1][k]; }
                    else { unsafeAtomicAdd(p1 + k, ps[0][k]); unsafeAtomicAdd(p2 + k, ps[1][k]); }
                }
            }
        }
        Y2S_STAMP(7);
        // the idle DMA slots have landed before this workgroup's LDS is reused (next segment) or released (kernel end); the output stores ride along
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        Y2S_STAMP(12);
    }
  }
  if (park_pending) {       // the parked tail was this workgroup's last segment
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(flags + wx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      Y2S_STAMP(8);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (no LDS-DMA may be in flight when the workgroup's LDS is released)
#ifdef Y2S_EXPERIMENTS
  if (A_PHASES && lane == 0 && (wave == 0 || wave == 4)) {
      ph_[11] = wall_clock64();
      unsigned long long *dst = y2s_phase + ((size_t)wx * 2 + (wave >> 2)) * Y2S_PHASES;
#pragma unroll
      for (int k = 0; k < Y2S_PHASES; ++k) dst[k] = ph_[k];
  }
#endif
}

// Launch (called by conv_igemm.hip launch_conv once it has decided that the shape takes the tap-fused family and the loader / consumer member of it):
// `grid` workgroups share the flat (tile, K step) space -- one per CU = stream-K; one per tile = whole tiles, no hand-off.  abl: experiments build only.
int y2_conv3x3_s4_launch(const void *P, unsigned p_bytes, const void *F, unsigned f_bytes, const float *bias, void *O, float *ws, int H, int W, int Cp,
                         int ldp, int Nf, int ldo, int M, int NT, const float *bn_shift, float *bn_part, unsigned *sk_flags, float act_alpha,
                         const Y2BnBwd &bz, int k_rotate, int grid, int abl, hipStream_t st) {
#define Y2S_LAUNCH(BWDv, HRv, NSBv, ABLv)                                                                                                   \
    conv3x3_s4_kernel<BWDv, HRv, NSBv, ABLv><<<dim3(grid), 512, 0, st>>>((const bf16 *)P, p_bytes, (const bf16 *)F, f_bytes, bias, (bf16 *)O, ws, \
                                                                         H, W, Cp, ldp, Nf, ldo, M, NT, bn_shift, bn_part, sk_flags, act_alpha, bz, k_rotate)
    if (W > 55) return 1;
    const bool bwd = bz.Y != nullptr;
#ifdef Y2S_EXPERIMENTS
#define Y2S_ABL_CASE(ABLv)                                                                                         \
    case ABLv:                                                                                                     \
        if (W <= 27) Y2S_LAUNCH(false, 312, 5, ABLv); else Y2S_LAUNCH(false, 368, 4, ABLv);                        \
        return 0;
    switch (abl) {
        Y2S_ABL_CASE(1) Y2S_ABL_CASE(2) Y2S_ABL_CASE(4) Y2S_ABL_CASE(8) Y2S_ABL_CASE(16) Y2S_ABL_CASE(32) Y2S_ABL_CASE(32 + 64) Y2S_ABL_CASE(32 + 4) Y2S_ABL_CASE(32 + 2) Y2S_ABL_CASE(32 + 1) Y2S_ABL_CASE(1 + 2 + 4) Y2S_ABL_CASE(8 + 16)
        default: break;
    }
#undef Y2S_ABL_CASE
#endif
    (void)abl;
    if (W <= 27) { if (bwd) Y2S_LAUNCH(true, 312, 5, 0); else Y2S_LAUNCH(false, 312, 5, 0); }
    else { if (bwd) Y2S_LAUNCH(true, 368, 4, 0); else Y2S_LAUNCH(false, 368, 4, 0); }
    return 0;
#undef Y2S_LAUNCH
}
