// Ping-pong tap-fused 3x3 implicit GEMM (bf16, 256 x 128 tile, 8 waves of 64 x 64), forward and data gradient.  Round 4.
//
// Replaces slim.layers.conv2d for the 3x3 layers with >= 64-channel-multiple inputs on images up to 55 wide (reference
// model/yolo2/inference.py:81-117) and its tf.gradients input gradient (train.py:127-129).  Same operand layouts, halo image, stream-K
// decomposition and epilogues as conv3x3_tap_kernel (conv_igemm.hip); what changes is WHO does WHAT WHEN inside a K step.
//
// What the counters said about the round-2 kernel (profiles/r02_igemm_tap.md, r02_final_sq_counters.md): a K step (one tap of one
// 64-channel chunk: 16 MFMAs, 16 ds_read_b128 and 3 DMA pieces per wave) took 0.95 us, of which the 32 MFMAs of a SIMD are 0.45 us
// and the instruction skeleton around them 0.52 us -- the two simply ADD UP, because every barrier puts all eight waves in the same
// phase: both waves of a SIMD compete for the matrix pipe together and then idle it together.  Here the two waves of a SIMD are kept in
// OPPOSITE phases by construction (the 8-wave attention schedule of MI355X_MICROARCH.md "Two waves per SIMD"):
//
//     waves 0-3 (one per SIMD)   LOAD(s) | MFMA(s) | LOAD(s+1) | MFMA(s+1) | ...
//     waves 4-7 (one per SIMD)           | LOAD(s) | MFMA(s)   | LOAD(s+1) | MFMA(s+1) ...        ("|" = s_barrier of the whole workgroup)
//
//   LOAD(s): the step's 3 DMA pieces (NSB-1 steps ahead), its 16 fragment reads into ONE register set, counted vmcnt, lgkmcnt(0).
//   MFMA(s): sixteen back-to-back MFMAs on registers only.  The matrix pipe of a SIMD always has exactly one owner; the partner's LDS
//   latency, DMA issue and address arithmetic run beside it.  The second group starts one barrier late and the first group pads one
//   barrier at the end of a segment, so both execute the same number of barriers.
//   Hazards (s = K step, slot = s mod NSB of the filter ring, D = NSB-1 = DMA distance):
//     RAW  a wave's pieces of step s+1 are waited for (counted vmcnt) at the end of its LOAD(s), i.e. at least one barrier before
//          the first group reads them in LOAD(s+1);
//     WAR  the DMA of step s+D lands in the slot of step s-1, whose last reader (second group, LOAD(s-1)) retired its reads with
//          lgkmcnt(0) one barrier before the first group issues that DMA in LOAD(s).
//   * The nine taps are unrolled: tap offsets, swizzle terms and image-border masks become 18 precomputed LDS addresses per lane
//     (one per tap and fragment row; masked (pixel, tap) pairs point at a zero KiB inside each halo buffer, so switching halo buffers is
//     one add per address) -- the ~46 VALU + ~60 SALU of per-step addressing of the round-2 kernel shrink to ~14 VALU.
//   * One fragment register set (64 VGPRs) instead of two: ~170 VGPRs.
#include "common.h"
#include "conv_shared.h"
#include <type_traits>

#define Y2P_BM 256
#define Y2P_BN 128
#define Y2P_BBYTES (Y2P_BN * 128)
#define Y2P_LOADS 3                        // DMA instructions per wave per K step: 1 halo slot + 2 filter pieces

// DMA instructions in the slot of tap t (any integer: taps wrap around the nine of a chunk): halo piece (or its idle stand-in) + two filter pieces
constexpr int y2p_slot_loads(int t, int hslots, bool skipidle) {
    const int t9 = ((t % 9) + 9) % 9;
    return (!skipidle || t9 < hslots) ? 3 : 2;
}
// How many of its newest DMA instructions a wave may leave in flight at the end of LOAD(tap tp) so that its pieces of the NEXT step have landed
constexpr int y2p_leave(int order, int D, int tp, int hslots, bool skipidle) {
    int n = 0;
    if (order == 1) { for (int k = 1; k <= D - 2; ++k) n += y2p_slot_loads(tp - k, hslots, skipidle); }       // slots issued in MFMA(kt-1) .. MFMA(kt-D+2)
    else if (order == 4) {      // filter pairs of LOAD(kt-D+2 .. kt) + the halo pieces of MFMA(kt-D+1 .. kt-1)
        n = 2 * (D - 1);
        for (int k = 1; k <= D - 1; ++k) n += y2p_slot_loads(tp - k, hslots, skipidle) - 2;
    } else { for (int k = 0; k <= D - 2; ++k) n += y2p_slot_loads(tp - k, hslots, skipidle); }                  // slots issued in LOAD(kt) .. LOAD(kt-D+2)
    return n;
}
// ORDER 7: what the SECOND group may leave in flight at the end of MFMA(tap tp) so that its pieces of step kt+2 have landed (the first group
// reads that step's filter fragments before the second group's next LOAD-phase wait): the slots of LOAD(kt) .. LOAD(kt-D+3)
constexpr int y2p_leave_m(int D, int tp, int hslots, bool skipidle) {
    int n = 0;
    for (int k = 0; k <= D - 3; ++k) n += y2p_slot_loads(tp - k, hslots, skipidle);
    return n;
}

// HROWS = halo rows held (>= 256 + 2 W + 2, multiple of 8), NSB = filter ring depth (DMA runs NSB-1 steps ahead of the reads).
// SCHED (A/B of where a step's three DMA pieces are issued; bits 0-2 = order):
//   0  DMA pieces, then the fragment reads                       1  all pieces behind the first four MFMAs of the MFMA phase
//   2  fragment reads, then the pieces (issued under the reads' latency)      3  reads, lgkmcnt(0), then the pieces (read-free gap)
//   4  reads + the two filter pieces in the LOAD phase, the halo piece behind the first four MFMAs
//   7  as 2, with the FILTER fragments of step s+1 read behind the last MFMA of step s (into the registers those MFMAs just released; the
//      second group then also waits, at the end of its MFMA phase, for its own DMA pieces of step s+2): 3-5 % slower than 2 on every 13x13 /
//      26x26 layer (conv20 forward 150 vs 143 us on one box, profiles/r04_pp7_b16.txt) -- experiments build only
//   (5 = as 2 with the pixel fragments of step s+1 read behind the MFMAs of step s, and 6 = no ping-pong at all, one barrier per step with
//    every fragment read trailing the MFMAs that release its registers, were built and measured: conv20 forward 137 and 147 us against 129 for
//    order 2 -- any read or DMA instruction inside a wave's MFMA stream delays its next MFMA -- profiles/r04_pp3_b16.txt, r04_pp5_b16.txt;
//    the code is in the history of this file)
//   +8  slots of taps >= HSLOTS carry no halo piece at all (2 instead of 3 instructions; the counted waits use the exact per-tap sums)
//   +16 s_setprio 1 for the MFMA phase
#ifdef Y2P_EXPERIMENTS
#define Y2P_PHASES 16
__device__ unsigned long long y2p_phase[1024 * 2 * Y2P_PHASES];
extern "C" int yolo2_debug_pp_phases(void *host_out) { return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(y2p_phase), sizeof(unsigned long long) * 1024 * 2 * Y2P_PHASES); }
#define Y2P_STAMP(k) do { if (A_PHASES) { const unsigned long long t_ = wall_clock64(); ph_[k] += t_ - ph_tl; ph_tl = t_; } } while (0)
#else
#define Y2P_STAMP(k) do { } while (0)
#endif
template <bool BNBWD, int HROWS, int NSB, int SCHED>
__global__ __launch_bounds__(512) void conv3x3_pp_kernel(
    const bf16 *__restrict__ P, unsigned p_bytes, const bf16 *__restrict__ F, unsigned f_bytes, const float *__restrict__ bias,
    bf16 *__restrict__ O, float *__restrict__ slots, int H, int W, int Cp, int ldp, int Nf, int ldo, int M, int NT,
    const float *__restrict__ bn_shift, float *__restrict__ bn_part, unsigned *__restrict__ flags, float act_alpha, const Y2BnBwd bz, int k_rotate, int cv) {
    typedef bf16 T;
    constexpr int BM = Y2P_BM, BN = Y2P_BN, NW = 8, WGN = 2, WGM = 4, TM = 2, TN = 2, VEC = 8, ROWB = 128, TAPS = 9;
    constexpr int HBYTES = HROWS * 128, HB = HBYTES + 1024, RING = 2 * HB, D = NSB - 1;
    constexpr int HPIECES = HROWS / 8, HSLOTS = (HPIECES + NW - 1) / NW;
    constexpr int ORDER = SCHED & 7;
    constexpr bool SKIPIDLE = (SCHED & 8) != 0, PRIO = (SCHED & 16) != 0;
    // timing ablations (wrong results by design; scripts/pp_sweep.py): +64 no MFMAs, +128 no fragment reads, +256 no DMA inside the K loop
    constexpr bool A_NOMFMA = (SCHED & 64) != 0, A_NOREAD = (SCHED & 128) != 0, A_NODMA = (SCHED & 256) != 0;
    constexpr bool A_NOEPI = (SCHED & 512) != 0, A_NOHANDOFF = (SCHED & 1024) != 0;
    // +4096 (with +512): s_memtime stamps at the phase boundaries of every K step; each wave leaves its sums in O as eight 64-bit words
    // {kernel cycles, LOAD issue, LOAD wait, barrier after LOAD, MFMA issue, barrier after MFMA, steps, workgroup * 8 + wave}
    // (scripts/pp_phase_cycles.py: cycles per phase, and the shader clock = kernel cycles / kernel duration)
    constexpr bool A_TIME = (SCHED & 4096) != 0;
    // +8192 (product configuration otherwise): wall-clock (100 MHz) sums per phase OUTSIDE the K loop -- prologue, loop, park, flag waits, partner reads,
    // drain, staging, stores, final publication -- of waves 0 and 4 of every workgroup, left in y2p_phase (scripts/pp_fixed_cost.py)
    constexpr bool A_PHASES = (SCHED & 8192) != 0;
    static_assert(!A_TIME || A_NOEPI, "the cycle stamps go where the output tile would");      // +512 no epilogue (stores, statistics), +1024 no stream-K hand-off traffic
    // halo pieces of chunk c+1 ride in the slots of taps 0 .. HSLOTS-1 of chunk c and must be covered by the wait at the end of LOAD(tap 8)
    // (ORDER 4 issues the halo piece half a step later: one slot less)
    static_assert(HROWS % 8 == 0 && HSLOTS <= TAPS + (ORDER == 4 ? 0 : 1) - D && D >= 2 && RING + NSB * Y2P_BBYTES <= 160 * 1024, "LDS plan");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[RING + NSB * Y2P_BBYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int MT = (M + BM - 1) / BM;
    const int nk = (Cp / 64) * TAPS;                     // K steps per tile: (64-channel chunk, tap)
    const long su_total = (long)MT * NT * nk;
    const int G = gridDim.x;
    const int wx = (G & 7) ? (int)blockIdx.x : (int)((blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3));
    // Cost-balanced shares (round 6).  The workgroup that holds K step 0 of a tile pays for more than K steps: it waits for and reads its partners'
    // parked partials and runs the tile's epilogue, and when its range also holds the tail of the previous tile it pays a second prologue --
    // measured 14 us between the first and the last workgroup to finish with equal K-step shares (profiles/r06_pp_fixed_cost.txt).  The flat space is
    // therefore cut in VIRTUAL units: every tile is nk K steps preceded by `cv` cost units that map to no K step; workgroup w takes the w-th equal
    // share of the virtual space, so the owner of a tile gets up to cv fewer real K steps.  cv = 0 is round 5's partition.  (launch_conv keeps
    // cv below half a share: every workgroup of a stream-K launch holds at least one K step.)
    const long vt = (long)nk + cv;
    const long v_total = (long)MT * NT * vt;
    auto v2r = [&](long v) { const long t_ = v / vt, o_ = v - t_ * vt; return t_ * nk + (o_ > cv ? o_ - cv : 0); };
    auto share_end = [&](int w) { return G == MT * NT ? (long)(w + 1) * nk : v2r((long)(w + 1) * v_total / G); };      // (one workgroup per tile: exactly)
    long su = wx == 0 ? 0 : share_end(wx - 1);
    const long su_end = share_end(wx);
    // K rotation of the stream-K tiles (conv3x3_tap_kernel; profiles/r03_l2_stationary_ab.md): chunk c of tile t lives at memory chunk (c + rot_t) mod nch
    const int nch = Cp / 64;
    const double share = (double)su_total / (double)G;
    const int shares_per_tile = (int)((double)nk / share + 0.5);
    const double drift_chunks = k_rotate && shares_per_tile >= 2 ? ((double)shares_per_tile * share - (double)nk) / (double)TAPS : 0.0;

    const __amdgpu_buffer_rsrc_t rsrcP = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(P), 0, p_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcF = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(F), 0, f_bytes, 0x00020000);
    const double rcp_hw = 1.0 / (double)(H * W);
    const float rcp_w = 1.0f / (float)W;
    unsigned char *const zero0 = smem + HBYTES;         // the zero KiB of halo buffer 0
    // the sink of idle DMA slots (out-of-range DMA writes zeros): the zero KiB of halo buffer 1, which lies BEHIND the tile image the epilogue stages
    // over the halo buffers -- idle slots still in flight when the K loop ends then need no drain before the staging starts (1 us per tile)
    unsigned char *const sink = smem + HB + HBYTES;
    typedef __attribute__((address_space(3))) void *lds_void_ptr;
    typedef const __attribute__((address_space(3))) bf16x8 *lds_frag_ptr;

  bool park_pending = false;
#ifdef Y2P_EXPERIMENTS
  unsigned long long tm_k0 = 0, tm_issue = 0, tm_wait = 0, tm_barl = 0, tm_mfma = 0, tm_barm = 0, tm_steps = 0;
  if (A_TIME) tm_k0 = __builtin_amdgcn_s_memtime();
  unsigned long long ph_[Y2P_PHASES] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ph_tl = 0;
  if (A_PHASES) { ph_tl = wall_clock64(); ph_[10] = ph_tl; }
#endif
  for (bool first_seg = true;; first_seg = false) {
    if (su >= su_end) break;
    const int t = (int)(su / nk);
    const int kt_beg = (int)(su - (long)t * nk);
    const int kt_end = (int)min((long)nk, kt_beg + (su_end - su));
    su += kt_end - kt_beg;
    const int nt = t / MT, mt = t - nt * MT;
    if (!first_seg) __syncthreads();                     // every wave is done with the previous segment's LDS (tile image of its epilogue included)
    const int m0 = mt * BM, n0 = nt * BN;
    int rot;                                             // this tile's chunk rotation, 0 <= rot < nch
    {
        const long r = (long)__builtin_floor(-(double)t * drift_chunks + 0.5);
        rot = (int)(r % nch);
        if (rot < 0) rot += nch;
        rot = __builtin_amdgcn_readfirstlane(rot);
    }
    auto mem_chunk = [&](int c) { const int m = c + rot; return m >= nch ? m - nch : m; };      // logical chunk (0 <= c <= nch) -> chunk in memory

    // per-segment copy of the lane id the compiler cannot see through: everything a segment precomputes per lane is recomputed from it,
    // so no per-lane value of one segment (or of the kernel prologue) stays live across the epilogue of another
    int lane_s = lane;
    asm volatile("" : "+v"(lane_s));
    const int frow_s = lane_s & 31;
    // both zero KiB (the previous segment's epilogue staged its tile image over them)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcP, (lds_void_ptr)zero0, 16, Y2_OOB, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcP, (lds_void_ptr)(zero0 + HB), 16, Y2_OOB, 0, 0, 0);

    // halo DMA descriptor: slot j of this wave is piece j * 8 + wave = halo rows 8 * piece .. + 7.  Rows before pixel 0 wrap to offsets
    // >= 2^31 and rows past the last pixel lie beyond num_records: both read as zeros.  Source-side swizzle ((row >> 1) & 7).
    const int hr0 = wave * 8 + (lane_s >> 3);
    const unsigned h_voff0 = (unsigned)(m0 - (W + 1) + hr0) * (unsigned)ldp * 2u + (unsigned)(((lane_s & 7) ^ ((hr0 >> 1) & 7)) * 16);
    const unsigned h_stride = 64u * (unsigned)ldp * 2u;
    // filter DMA descriptor (two pieces per wave: rows r and r + 8; filters >= Nf lie beyond num_records)
    const int br0 = wave * 16 + (lane_s >> 3);
    unsigned b_voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = br0 + 8 * i;
        b_voff[i] = (unsigned)(n0 + r) * (unsigned)(TAPS * Cp) * 2u + (unsigned)(((lane_s & 7) ^ ((r >> 1) & 7)) * 16);
    }
    const int c_first = kt_beg / TAPS;
    // one DMA slot: halo piece `hs` of chunk `hc` (or an idle write into the zero KiB) + the filter tile of K step `kb` into ring stage `bstage`
    auto issue_halo = [&](int hs, int hc, int hc_mem, bool hreal) {
        const bool real = hreal && hs * NW + wave < HPIECES;
        unsigned char *dst = real ? smem + (hc & 1) * HB + (hs * NW + wave) * 1024 : sink;
        const unsigned voff = real ? h_voff0 + (unsigned)hs * h_stride + (unsigned)hc_mem * 128u : Y2_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcP, (lds_void_ptr)dst, 16, voff, 0, 0, 0);
    };
    auto issue_filter = [&](int kb, unsigned offB, int bstage) {
        const bool breal = kb < kt_end;
        unsigned char *Bs = smem + RING + bstage * Y2P_BBYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            unsigned char *dst = Bs + (wave * 2 + i) * 1024;             // (an idle slot zero-fills a stage no valid step reads any more)
            const unsigned voff = breal ? b_voff[i] + offB : Y2_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcF, (lds_void_ptr)dst, 16, voff, 0, 0, 0);
        }
    };
    auto filt_off = [&](int kb) {                        // byte offset of K step kb inside a filter row (run-time form: prologue only)
        const int bc = kb / TAPS, bt = kb - bc * TAPS;
        return (unsigned)(mem_chunk(bc) * TAPS * 64 + bt * 64) * 2u;
    };

    {   // prologue: the whole halo of the first chunk (+ the pieces of the next chunk whose slots this segment starts behind),
        // then the filter tiles of steps kt_beg .. kt_beg + D - 1 into ring stages 0 .. D-1
        const int c_tap = kt_beg - c_first * TAPS;
#pragma unroll
        for (int j = 0; j < HSLOTS; ++j) {
            const bool real = j * NW + wave < HPIECES;
            unsigned char *dst = real ? smem + (c_first & 1) * HB + (j * NW + wave) * 1024 : sink;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcP, (lds_void_ptr)dst, 16, real ? h_voff0 + (unsigned)j * h_stride + (unsigned)mem_chunk(c_first) * 128u : Y2_OOB, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < HSLOTS; ++j) {           // slots 0 .. c_tap-1 of the NEXT chunk's halo would have been issued by now
            const bool real = j < c_tap && j * NW + wave < HPIECES && (c_first + 1) * TAPS < kt_end;
            unsigned char *dst = real ? smem + ((c_first + 1) & 1) * HB + (j * NW + wave) * 1024 : sink;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcP, (lds_void_ptr)dst, 16, real ? h_voff0 + (unsigned)j * h_stride + (unsigned)mem_chunk(c_first + 1) * 128u : Y2_OOB, 0, 0, 0);
        }
        // (prologue slots always carry three instructions, filter pieces first: never fewer, never in a later position, than the counted
        // waits of the first steps assume for the steady-state slots they stand in for)
#pragma unroll
        for (int d = 0; d < D; ++d) {
            issue_filter(kt_beg + d, filt_off(kt_beg + d), d);
            issue_halo(0, 0, 0, false);
        }
    }
    // ---- per-lane index arithmetic (f64 reciprocal pixel decode, 18 read addresses), under the latency of the prologue's DMA
    // which of the nine taps of this lane_s's two fragment rows lie inside the image
    unsigned amask = 0;                                  // 9 bits per fragment row, row i at bit 16 * i
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
        amask |= mask << (16 * i);
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- LDS read addresses.  A: one per (tap, fragment row), valid for the halo buffer of the current chunk parity; the 16-k group
    // kk is XORed in (bits 5-6 come from the swizzle term alone: every other summand is a multiple of 128).  B: one per 16-k group.
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
            const bool ok = ((amask >> (16 * i + tp)) & 1u) != 0u;
            const unsigned sw = ((hb >> 4) & 0x70u) ^ hi16;           // ((row >> 1) & 7) << 4, folded with this lane_s's half of the k group
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

    if (park_pending) {
        // The tail segment this workgroup ran before this one parked its partial tile with write-through stores and came straight here:
        // their acknowledgement latency ran under this prologue's address arithmetic and DMA issue.  vmcnt(0), not a counted wait: loads
        // and stores share the counter and only "everything older has completed" is certain for a mix of the two.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(flags + wx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        park_pending = false;
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ORDER == 7 ? D - 2 : D - 1) * Y2P_LOADS) : "memory");      // zero KiBs, halo and filter tile kt_beg have landed (the newer slots stay in flight)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (wave >= 4) __builtin_amdgcn_s_barrier();         // second group: one phase behind the first
    __builtin_amdgcn_sched_barrier(0);
    Y2P_STAMP(0);

    bf16x8 fa[4][TM], fb[4][TN];
    if (A_NOREAD) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) { fa[kk][i][e] = (bf16)(float)(lane_s + e); fb[kk][i][e] = (bf16)(float)(lane_s - e); }
    }
#ifdef Y2P_EXPERIMENTS
    unsigned long long tm_pa = 0, tm_p1 = 0, tm_p2 = 0, tm_p3 = 0;      // (A_TIME) stamps of the previous step, consumed behind the next lgkmcnt(0)
    bool tm_prev = false;
#endif
    int kt = kt_beg;                                     // the K step the next executed phase pair belongs to
    int stage_r = 0, stage_i = D;                        // ring stage read by step kt / filled by the DMA slot of step kt (= stage of step kt + D)

    for (int c = c_first; c * TAPS < kt_end; ++c) {
        const int lo = max(0, kt_beg - c * TAPS), hi = min(TAPS, kt_end - c * TAPS);       // taps of this chunk inside the segment
        const int mc0 = mem_chunk(c), mc1 = mem_chunk(c + 1);
        const bool next_ok = (c + 1) * TAPS < kt_end;
        const unsigned delta = (c & 1) ? (unsigned)(-HB) : (unsigned)HB;                   // to the other halo buffer
        auto step = [&](auto tap_tag) {
            constexpr int tp = decltype(tap_tag)::value;
            if (tp >= lo && tp < hi) {
                constexpr int bt = (tp + D) % TAPS;
                constexpr bool has_halo = !SKIPIDLE || tp < HSLOTS;       // does this tap's slot carry a halo instruction at all?
                const unsigned offB = (unsigned)(((tp + D >= TAPS) ? mc1 : mc0) * TAPS * 64 + bt * 64) * 2u;
                auto dma_filter = [&]() { if (!A_NODMA) issue_filter(kt + D, offB, stage_i); };
                auto dma_halo = [&]() { if (has_halo && !A_NODMA) issue_halo(tp, c + 1, mc1, tp < HSLOTS && next_ok); };
                // DMA instructions this wave may leave in flight at the end of LOAD(tp): everything issued after its pieces of step kt+1
                constexpr int LEAVE = y2p_leave(ORDER, D, tp, HSLOTS, SKIPIDLE);
                // ---- LOAD phase
#ifdef Y2P_EXPERIMENTS
                unsigned long long tm_0 = 0, tm_a = 0, tm_1 = 0, tm_2 = 0, tm_3 = 0;
                if (A_TIME) { tm_0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#endif
                if (ORDER == 0) { dma_halo(); dma_filter(); }
                const unsigned so = (unsigned)(stage_r * Y2P_BBYTES);
                if (!A_NOREAD) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                        for (int i = 0; i < TM; ++i) fa[kk][i] = *(lds_frag_ptr)(uintptr_t)(aaddr[tp][i] ^ (unsigned)(kk * 32));
                    if (ORDER != 7 || kt == kt_beg) {       // (order 7: only a segment's first step reads its filter fragments here)
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) {
                            const unsigned bb = baddr[kk] + so;
#pragma unroll
                            for (int j = 0; j < TN; ++j) fb[kk][j] = *(lds_frag_ptr)(uintptr_t)(bb + (unsigned)(j * 32 * ROWB));
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (ORDER == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (ORDER == 2 || ORDER == 3 || ORDER == 7) { dma_halo(); dma_filter(); }
                if (ORDER == 4) dma_filter();
                __builtin_amdgcn_sched_barrier(0);
#ifdef Y2P_EXPERIMENTS
                if (A_TIME) { tm_a = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#endif
                // this wave's pieces of step kt+1 have landed (the newest LEAVE instructions stay in flight)
                if (!A_NODMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LEAVE) : "memory");
                // lgkmcnt(0) through the builtin, not asm: hipcc's own wait insertion must KNOW that every fragment read has returned here, or
                // it counts the LOAD-phase reads as still outstanding and puts lgkmcnt(3) / lgkmcnt(1) in front of the later MFMA groups --
                // which then wait for the ORDER-5 prefetch reads issued a few instructions earlier (gfx9 encoding: vmcnt 63, expcnt 7, lgkmcnt 0)
                __builtin_amdgcn_s_waitcnt(0xc07f);
#ifdef Y2P_EXPERIMENTS
                if (A_TIME) {       // every stamp issued so far has returned (the wait above counts scalar memory too)
                    if (tm_prev) { tm_wait += tm_p1 - tm_pa; tm_barl += tm_p2 - tm_p1; tm_mfma += tm_p3 - tm_p2; tm_barm += tm_0 - tm_p3; }
                    tm_issue += tm_a - tm_0;
                    ++tm_steps;
                    __builtin_amdgcn_sched_barrier(0);
                    tm_1 = __builtin_amdgcn_s_memtime();
                    __builtin_amdgcn_sched_barrier(0);
                }
#endif
                if (A_NOMFMA && !A_NOREAD) {        // (ablation: the fragments count as used)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) asm volatile("" ::"v"(fa[kk][0]), "v"(fa[kk][1]), "v"(fb[kk][0]), "v"(fb[kk][1]));
                }
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // ---- MFMA phase: registers only
#ifdef Y2P_EXPERIMENTS
                if (A_TIME) { tm_2 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#endif
                if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            if (!A_NOMFMA) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk][i], fb[kk][j], acc[i][j], 0, 0, 0);
                    if ((ORDER == 1 || ORDER == 4) && kk == 0) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (ORDER == 1) { dma_halo(); dma_filter(); } else dma_halo();
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (PRIO) __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
#ifdef Y2P_EXPERIMENTS
                if (A_TIME) {
                    tm_3 = __builtin_amdgcn_s_memtime();
                    __builtin_amdgcn_sched_barrier(0);
                    tm_pa = tm_a; tm_p1 = tm_1; tm_p2 = tm_2; tm_p3 = tm_3; tm_prev = true;
                }
#endif
                if (ORDER == 7) {
                    // every MFMA of this step is issued: the filter fragments of step kt+1 go into the same registers now, their latency
                    // under the last MFMAs, the barrier and the next LOAD phase's pixel reads
                    if (!A_NOREAD && kt + 1 < kt_end) {
                        const unsigned so1 = (unsigned)((stage_r == NSB - 1 ? 0 : stage_r + 1) * Y2P_BBYTES);
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) {
                            const unsigned bb = baddr[kk] + so1;
#pragma unroll
                            for (int j = 0; j < TN; ++j) fb[kk][j] = *(lds_frag_ptr)(uintptr_t)(bb + (unsigned)(j * 32 * ROWB));
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (!A_NODMA && wave >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(y2p_leave_m(D, tp, HSLOTS, SKIPIDLE)) : "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                ++kt;
                stage_r = stage_r == NSB - 1 ? 0 : stage_r + 1;
                stage_i = stage_i == NSB - 1 ? 0 : stage_i + 1;
            }
            // the next chunk reads the other halo buffer (every tap's addresses move, whether or not this segment ran the tap)
#pragma unroll
            for (int i = 0; i < TM; ++i) aaddr[tp][i] += delta;
        };
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 3>{});
        step(std::integral_constant<int, 4>{});
        step(std::integral_constant<int, 5>{});
        step(std::integral_constant<int, 6>{});
        step(std::integral_constant<int, 7>{});
        step(std::integral_constant<int, 8>{});
    }
    __builtin_amdgcn_sched_barrier(0);
    if (wave < 4) __builtin_amdgcn_s_barrier();          // first group pads the barrier the second group took at the start
    __builtin_amdgcn_sched_barrier(0);
    Y2P_STAMP(1);
#ifdef Y2P_EXPERIMENTS
    if (A_PHASES) ph_[9] += 1;
#endif

    // Everything below indexes by (lane_e, wave_e): copies the compiler cannot see through, so that none of the hand-off / epilogue
    // address arithmetic is hoisted above the K loop (it was: 37 VGPRs of it spilled to scratch in the BN-backward variant, reloaded --
    // behind a vmcnt(0) that drains the DMA ring -- in every chunk of the loop).
    int lane_e = lane, wave_e = wave;
    asm volatile("" : "+v"(lane_e), "+s"(wave_e));
    const int wm_e = wave_e / WGN, wn_e = wave_e % WGN;
    // ---- stream-K hand-off (as in conv_igemm_kernel: the workgroup holding K step 0 of a tile owns it; a tail segment is parked)
    {
        constexpr int SLOT = BM * BN;
        const __amdgpu_buffer_rsrc_t rsrcS = __builtin_amdgcn_make_buffer_rsrc(slots, 0, (unsigned)((size_t)gridDim.x * SLOT * sizeof(float)), 0x00020000);
        const unsigned slot_lane = (unsigned)(((size_t)wave_e * (TM * TN * 16 * 64) + (size_t)lane_e * 4) * sizeof(float));
        if (A_NOHANDOFF) { if (kt_beg > 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); continue; } }
        else if (kt_beg > 0) {
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
            if constexpr (BNBWD) {
                // (the BN-backward instantiation publishes at once: with the deferred form below hipcc's allocation of its 245 registers spills)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) __hip_atomic_store(flags + wx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                // published -- stores acknowledged, flag raised -- from inside the next segment's prologue, or after the loop when this was the
                // workgroup's only segment
                park_pending = true;
            }
            Y2P_STAMP(2);
            continue;
        }
        if (!A_NOHANDOFF && kt_end < nk) {
            const long tile_end = su - kt_end + nk;
            long covered = su;
            // every partner's flag at once: lane i of wave 0 waits for (and clears) the flag of partner wx + 1 + i -- the flag loads are one
            // cross-die round trip each (~1.2 us), one after the other they cost an owner with two partners 2.6 us (profiles/r06_pp_fixed_cost.txt)
            {
                int np_ = 0;
                for (long cov = su; cov < tile_end; ++np_) cov = share_end(wx + 1 + np_);
                if (wave_e == 0)
                    for (int i = lane_e; i < np_; i += 64) y2_sk_wait_and_clear(flags, wx + 1 + i);      // (bounded: conv_shared.h)
                __syncthreads();
                Y2P_STAMP(3);
            }
            for (int p = wx + 1; covered < tile_end; ++p) {
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
                covered = share_end(p);
#ifdef Y2P_EXPERIMENTS
                if (A_PHASES) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); Y2P_STAMP(4); }
#endif
            }
        }
    }

    if (A_NOEPI) {       // (ablation: keep the accumulators alive, store nothing)
        if (acc[0][0][0] == 123.456f && acc[1][1][5] == 1.0f) O[0] = (bf16)acc[0][1][3];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        continue;
    }
    // ---- epilogue: the wide-store form of conv_igemm_kernel (tile rounded into a per-wave_e LDS image, 16-byte stores), with the
    // forward statistics or the producer layer's BN-backward sums taken from the rounded values
    {
        const bool stats = !BNBWD && bn_part != nullptr;
        const bool bstats = BNBWD && bn_part != nullptr;
        const bool stats_unique = bz.stat_mask_inv == 0;       // the host found a row for every (pixel tile, wave_e row) pair
        constexpr int WROWS = TM * 32, WROWB = TN * 32 * 2, WSTRIDE = WROWB + 16, WCPR = WROWB / 16, NIT = WROWS * WCPR / 64, YG = 4;
        static_assert(NW * WROWS * WSTRIDE <= RING, "tile image fits the halo buffers");
        static_assert(NW * WROWS * WSTRIDE <= HB + HBYTES, "the idle-DMA sink (zero KiB of halo buffer 1) lies behind the tile image");
        float cmu[VEC], cinv[VEC], cga[VEC], cbt[VEC], ps[2][VEC];
        Vec16<T> yv[YG];
        const int bz_nb = min(n0 + wn_e * TN * 32 + (lane_e % WCPR) * VEC, Nf - VEC);
        auto bz_load_y = [&](int it0) {
#pragma unroll
            for (int u = 0; u < YG; ++u) {
                const int m = min(m0 + wm_e * WROWS + ((it0 + u) * 64 + lane_e) / WCPR, M - 1);
                yv[u] = ld16(reinterpret_cast<const T *>(bz.Y) + (long)m * Nf + bz_nb);
            }
        };
        // per-column constants of both accumulator columns, requested before the drain below (their latency runs under it)
        float bvj[TN], shj[TN];
        bool nokj[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn_e * TN + j) * 32 + (lane_e & 31);
            nokj[j] = n < Nf;
            bvj[j] = (bias && nokj[j]) ? bias[n] : 0.f;
            shj[j] = (stats && nokj[j]) ? bn_shift[n] : 0.f;
        }
        // (idle DMA slots of the last steps may still be in flight: they write zeros into ring stages and the sink KiB, all behind the tile image)
        if (BNBWD && bstats) bz_load_y(0);                    // (in flight under the staging loop)
        __syncthreads();                                      // every wave has finished reading the last step's operands
        Y2P_STAMP(5);
        unsigned char *wreg = smem + wave_e * (WROWS * WSTRIDE);
        const bool tail = m0 + BM > M;
        // Staging: rounded tile into this wave's LDS image (+ the statistics of the rounded values).  Two copies of the 64-element loop,
        // chosen by ONE uniform branch: the leaky ReLU of a BN-folded inference layer and the row test of the last pixel tile each cost
        // two to three VALU per element when they are tested inside it (~10 % of the ~7 us a tile's epilogue takes).
        auto stage_tile = [&](auto act_tag, auto tail_tag) {
            constexpr bool ACT = decltype(act_tag)::value, TAIL = decltype(tail_tag)::value;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float bv = bvj[j], sh = shj[j];
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
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
                        const int slot = (mt * WGM + wm_e) & ((Y2_BN_PART_ROWS - 1) ^ bz.stat_mask_inv);
                        float *p1 = bn_part + (long)slot * Nf + n, *p2 = bn_part + (long)(Y2_BN_PART_ROWS + slot) * Nf + n;
                        if (stats_unique) { *p1 = s1; *p2 = s2; }
                        else { unsafeAtomicAdd(p1, s1); unsafeAtomicAdd(p2, s2); }
                    }
                }
            }
        };
        if (act_alpha != 1.0f) { if (tail) stage_tile(std::true_type{}, std::true_type{}); else stage_tile(std::true_type{}, std::false_type{}); }
        else if (tail) stage_tile(std::false_type{}, std::true_type{});
        else stage_tile(std::false_type{}, std::false_type{});
        Y2P_STAMP(6);
        if (bstats) {
            // (the per-channel constants through a second opaque copy of the column index: hipcc otherwise hoists their loads above the staging
            // loop, where the 64 accumulator registers are still live -- the BN-backward instantiations then need 245 .. 256+ registers)
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
        // (groups of YG rows, NOT unrolled across groups: fully unrolled, the eight iterations' loads were all hoisted to the top and the
        // BN-backward variant needed more than 256 registers)
#pragma unroll 1
        for (int g0 = 0; g0 < NIT; g0 += YG) {
            if (bstats && g0) bz_load_y(g0);
#pragma unroll
            for (int u = 0; u < YG; ++u) {
                const int id = (g0 + u) * 64 + lane_e;
                const int row = id / WCPR, ch = id % WCPR;
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
            // (the lanes that share a channel chunk meet on the VALU: common.h y2_lane_group_sum)
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                ps[0][k] = y2_lane_group_sum<WCPR>(ps[0][k]);
                ps[1][k] = y2_lane_group_sum<WCPR>(ps[1][k]);
            }
            // The four wave rows of the tile hold sums of the SAME channels: they meet in LDS (the 7 KiB between the tile image and the idle-DMA sink) and
            // the tile leaves ONE partial row per filter -- a quarter of the adds, and at most 169 pixel tiles instead of 676 (tile, wave row) pairs
            // competing for the partial rows: same-address f32 atomics were 12 us of the 52 x 52 launch's 52 (profiles/r06_pp_bn_epilogue.txt)
            static_assert(NW * WROWS * WSTRIDE + NW * WCPR * 2 * VEC * 4 <= HB + HBYTES, "reduction scratch fits between the tile image and the sink");
            float *const red = reinterpret_cast<float *>(smem + NW * WROWS * WSTRIDE);
            if (lane_e < WCPR) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) { red[(wave_e * WCPR + lane_e) * 2 * VEC + k] = ps[0][k]; red[(wave_e * WCPR + lane_e) * 2 * VEC + VEC + k] = ps[1][k]; }
            }
            __syncthreads();
            const int nb = n0 + wn_e * TN * 32 + lane_e * VEC;
            if (wm_e == 0 && lane_e < WCPR && nb < Nf) {
#pragma unroll
                for (int r = 1; r < WGM; ++r)
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        ps[0][k] += red[((r * WGN + wn_e) * WCPR + lane_e) * 2 * VEC + k];
                        ps[1][k] += red[((r * WGN + wn_e) * WCPR + lane_e) * 2 * VEC + VEC + k];
                    }
                const int slot = mt & ((Y2_BN_PART_ROWS - 1) ^ bz.stat_mask_inv);
                float *p1 = bn_part + (long)slot * Nf + nb, *p2 = bn_part + (long)(Y2_BN_PART_ROWS + slot) * Nf + nb;
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    if (stats_unique) { p1[k] = ps[0][k]; p2[k] = ps[1][k]; }
                    else { unsafeAtomicAdd(p1 + k, ps[0][k]); unsafeAtomicAdd(p2 + k, ps[1][k]); }
                }
            }
        }
        Y2P_STAMP(7);
        // the idle DMA slots have landed before this workgroup's LDS is reused (next segment) or released (kernel end); the output stores ride along
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        Y2P_STAMP(12);
    }
  }
  if (park_pending) {       // the parked tail was this workgroup's last segment
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(flags + wx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      Y2P_STAMP(8);
  }
#ifdef Y2P_EXPERIMENTS
  if (A_PHASES && lane == 0 && (wave == 0 || wave == 4)) {
      ph_[11] = wall_clock64();
      unsigned long long *dst = y2p_phase + ((size_t)wx * 2 + (wave >> 2)) * Y2P_PHASES;
#pragma unroll
      for (int k = 0; k < Y2P_PHASES; ++k) dst[k] = ph_[k];
  }
#endif
#ifdef Y2P_EXPERIMENTS
  if (A_TIME && lane == 0) {
      unsigned long long *dbg = reinterpret_cast<unsigned long long *>(O) + ((size_t)wx * NW + wave) * 8;
      dbg[0] = __builtin_amdgcn_s_memtime() - tm_k0; dbg[1] = tm_issue; dbg[2] = tm_wait; dbg[3] = tm_barl;
      dbg[4] = tm_mfma; dbg[5] = tm_barm; dbg[6] = tm_steps; dbg[7] = (unsigned long long)(wx * NW + wave);
  }
#endif
}

// Launch (called by conv_igemm.hip launch_conv once it has decided that the shape takes this kernel): `grid` workgroups share the
// flat (tile, K step) space -- one per CU = stream-K; one per tile = whole tiles, no hand-off.
int y2_conv3x3_pp_launch(const void *P, unsigned p_bytes, const void *F, unsigned f_bytes, const float *bias, void *O, float *ws, int H, int W, int Cp,
                         int ldp, int Nf, int ldo, int M, int NT, const float *bn_shift, float *bn_part, unsigned *sk_flags, float act_alpha,
                         const Y2BnBwd &bz, int k_rotate, int grid, int sched, int cv, hipStream_t st) {
#define Y2P_LAUNCH(BWDv, HRv, NSBv, SCv)                                                                                                    \
    conv3x3_pp_kernel<BWDv, HRv, NSBv, SCv><<<dim3(grid), 512, 0, st>>>((const bf16 *)P, p_bytes, (const bf16 *)F, f_bytes, bias, (bf16 *)O, ws, \
                                                                        H, W, Cp, ldp, Nf, ldo, M, NT, bn_shift, bn_part, sk_flags, act_alpha, bz, k_rotate, cv)
#define Y2P_CASE(SCv)                                                                                              \
    case SCv:                                                                                                      \
        if (W <= 27) { if (bwd) Y2P_LAUNCH(true, 312, 5, SCv); else Y2P_LAUNCH(false, 312, 5, SCv); }             \
        else { if (bwd) Y2P_LAUNCH(true, 368, 4, SCv); else Y2P_LAUNCH(false, 368, 4, SCv); }                     \
        return 0;
    if (W > 55) return 1;      // (the grid <= K steps clamp is launch_conv's: a workgroup without a K step never raises its flag, and its owner's bounded wait gives up)
    const bool bwd = bz.Y != nullptr || (sched & 32) != 0;      // (+32, measurements only: the BN-backward instantiation without its sums)
#ifdef Y2P_EXPERIMENTS      // (scripts/pp_experiments_build.sh: the SCHED variants and timing ablations behind profiles/r04_pp*.txt)
#define Y2P_ABL_CASE(SCv)                                                                                          \
    case SCv:                                                                                                      \
        if (SCv == 2 + 8192 && bwd) { if (W <= 27) Y2P_LAUNCH(true, 312, 5, SCv); else Y2P_LAUNCH(true, 368, 4, SCv); return 0; }      \
        if (W <= 27) Y2P_LAUNCH(false, 312, 5, SCv); else Y2P_LAUNCH(false, 368, 4, SCv);                          \
        return 0;
    if (sched >= 64) {
        switch (sched) {
            Y2P_ABL_CASE(2 + 64) Y2P_ABL_CASE(2 + 128) Y2P_ABL_CASE(2 + 256) Y2P_ABL_CASE(2 + 64 + 128) Y2P_ABL_CASE(2 + 128 + 256) Y2P_ABL_CASE(2 + 64 + 256)
            Y2P_ABL_CASE(2 + 64 + 128 + 256) Y2P_ABL_CASE(2 + 64 + 128 + 256 + 512) Y2P_ABL_CASE(2 + 64 + 128 + 256 + 1024) Y2P_ABL_CASE(2 + 64 + 128 + 256 + 512 + 1024)
            Y2P_ABL_CASE(2 + 512) Y2P_ABL_CASE(2 + 1024) Y2P_ABL_CASE(2 + 512 + 4096) Y2P_ABL_CASE(2 + 8192) Y2P_ABL_CASE(2 + 256 + 512 + 4096) Y2P_ABL_CASE(2 + 128 + 512 + 4096) Y2P_ABL_CASE(2 + 64 + 512 + 4096)
            default: return 1;
        }
    }
#undef Y2P_ABL_CASE
    switch (sched & 31) {
        Y2P_CASE(0) Y2P_CASE(1) Y2P_CASE(3) Y2P_CASE(4) Y2P_CASE(7) Y2P_CASE(10) Y2P_CASE(12) Y2P_CASE(18) Y2P_CASE(26)
        default: break;
    }
#endif
    switch (sched & 31) {
        Y2P_CASE(2)
        default: break;
    }
#undef Y2P_CASE
#undef Y2P_LAUNCH
    return 1;
}
