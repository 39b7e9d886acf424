// EXPERIMENT RECORD -- not built, not part of libyolo2hip.so (round 2).
// Persistent, cross-tile pipelined implicit-GEMM: gridDim.x resident workgroups walk contiguous ranges of the flat (tile, K step)
// space; the DMA issue cursor owns the source descriptors and keeps running across tile boundaries.  CORRECT (it passed the
// conv parity tests and tests/test_bench_shapes_gpu.py as the stream-K and full-grid variant) but SLOWER than the kernels
// in csrc/conv_igemm.hip on every layer (graph-timed, batch 16): conv2 46.5 -> 62.1 us, conv5 41.8 -> 56.3, conv13 44.5 -> 49.7,
// conv18 73.6 -> 80.2, conv20 186 -> 200.  Why: CDNA4 has ONE vector-memory counter for loads and stores.  A persistent
// workgroup's epilogue stores sit behind the already-issued DMAs of the next tile, so the next tile's first counted
// `s_waitcnt vmcnt(N)` also waits for the stores' acknowledgements (a non-persistent workgroup simply exits and lets them
// drain), and the per-tile descriptor arithmetic lands in the middle of the K loop.  A design that keeps global stores off
// the compute waves (a store wave fed through LDS) is the way to make persistence pay; see DESIGN.md section 5.
// It compiled inside csrc/conv_igemm.hip (it uses that file's Mma<T>, Y2_OOB and launch macros).
// ---------------------------------------------------------------------------------------------------------------------
// Persistent, cross-tile pipelined form of the kernel above.  gridDim.x workgroups stay resident and walk a contiguous range
// of the flat (tile, K step) space -- whole tiles (`whole_tiles`: full grids, no hand-off) or equal fractional shares
// (stream-K: the hand-off protocol of the kernel above).  The difference is the pipeline: the DMA issue cursor owns the
// per-lane source descriptors and simply keeps running across tile boundaries, so while a tile's epilogue (stores,
// statistics, parking, fix-up waits) runs, the first stages of the next tile are already in flight, and the descriptor
// arithmetic of the next tile (pixel decode, tap masks) is issued under the current tile's MFMAs instead of in front of an
// empty pipeline.  In the non-persistent form every tile paid DMA latency + prologue + epilogue back to back
// (profiles/r02_igemm_ablation.txt: the phases of a 128x128x576 tile add up almost serially).
// Requires Cp % (CH * VEC) == 0 (no channel tail).  The epilogue stores straight from the accumulators: the LDS ring is busy.
template <typename T, int BN, int WGN, int NSTAGE, int KS, int CH, int NW, int BMv>
__global__ __launch_bounds__(NW * 64) void conv_igemm_flat_kernel(
    const T *__restrict__ P, unsigned p_bytes, const T *__restrict__ F, unsigned f_bytes, const float *__restrict__ bias,
    T *__restrict__ O, float *__restrict__ Oacc, int H, int W, int Cp, int ldp, int Nf, int ldo, int M, int NT,
    const float *__restrict__ bn_shift, float *__restrict__ bn_part, unsigned *__restrict__ sk_flags, float act_alpha, int whole_tiles) {
    constexpr int BM = BMv;
    constexpr int TAPS = KS * KS;
    constexpr int VEC = 16 / sizeof(T);
    constexpr int BK = CH * VEC;
    constexpr int ROWB = CH * 16;
    constexpr int RPL = 256 / ROWB;
    constexpr int RPI = 64 / CH;
    constexpr int WGM = NW / WGN;
    constexpr int TM = BM / WGM / 32, TN = BN / WGN / 32;
    constexpr int A_IT = BM / RPI / NW;
    constexpr int B_PIECES = BN / RPI;
    constexpr int B_IT = (B_PIECES + NW - 1) / NW;
    constexpr int LOADS = A_IT + B_IT;
    constexpr int PAD = KS / 2;
    constexpr int STAGE = (BM + B_IT * NW * RPI) * ROWB;
    static_assert(A_IT >= 1 && NSTAGE >= 2 && NSTAGE <= 4 && TM >= 1 && TN >= 1, "tile");

    __shared__ __attribute__((aligned(1024))) unsigned char smem[NSTAGE * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int MT = (M + BM - 1) / BM;
    const int KC = y2_kchunk(Cp, TAPS);
    const int kpc = KC / BK;
    const int nk = (Cp / KC) * TAPS * kpc;
    // this workgroup's range: workgroup b runs on XCD b % 8; the ranges of one XCD are contiguous, tiles filter-tile-major
    const int G = gridDim.x, b = blockIdx.x;
    const int wx = (G & 7) ? b : (b & 7) * (G >> 3) + (b >> 3);
    const long ntiles = (long)MT * NT, su_total = ntiles * nk;
    const long su = whole_tiles ? (wx * ntiles / G) * nk : wx * su_total / G;
    const long su_end = whole_tiles ? ((wx + 1) * ntiles / G) * nk : (wx + 1) * su_total / G;
    if (su >= su_end) return;

    const __amdgpu_buffer_rsrc_t rsrcP = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(P), 0, p_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcF = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(F), 0, f_bytes, 0x00020000);
    const int lrow = lane / CH, lslot = lane % CH;
    const double rcp_hw = 1.0 / (double)(H * W);
    const float rcp_w = 1.0f / (float)W;

    // ---- issue side: source descriptors of the tile the cursor is in, and the cursor itself
    unsigned a_voff[A_IT], a_mask[A_IT], a_cb[A_IT], b_voff[B_IT], b_cb[B_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) a_cb[i] = (unsigned)((lslot ^ ((((wave * A_IT + i) * RPI + lrow) / RPL) % CH)) * 16);
#pragma unroll
    for (int i = 0; i < B_IT; ++i) b_cb[i] = (unsigned)((lslot ^ ((((wave * B_IT + i) * RPI + lrow) / RPL) % CH)) * 16);
    long iu = su;
    int i_kt = 0, i_chunk0 = 0, i_tap = 0, i_c0 = 0, i_dh = 0, i_dw = 0, i_stage = 0;
    auto setup_issue = [&](long u) {
        const int t = (int)(u / nk);
        const int kt = (int)(u - (long)t * nk);
        const int nt = t / MT, mt = t - nt * MT;
        const int m0 = mt * BM, n0 = nt * BN;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int m = m0 + (wave * A_IT + i) * RPI + lrow;
            unsigned mask = 0;
            if (m < M) {      // (b, h, w) of pixel m without the integer divider: see the kernel above
                const int HW = H * W;
                const int bq = (int)((double)m * rcp_hw);
                int rem = m - bq * HW;
                if (rem < 0) rem += HW; else if (rem >= HW) rem -= HW;
                int h = (int)((float)rem * rcp_w);
                int w = rem - h * W;
                if (w < 0) { w += W; --h; } else if (w >= W) { w -= W; ++h; }
                if (KS == 3) {
                    const unsigned cm = (w > 0 ? 1u : 0u) | 2u | (w < W - 1 ? 4u : 0u);
                    mask = (h > 0 ? cm : 0u) | (cm << 3) | (h < H - 1 ? cm << 6 : 0u);
                } else {
                    mask = 1u;
                }
            }
            a_mask[i] = mask;
            a_voff[i] = (unsigned)m * (unsigned)ldp * (unsigned)sizeof(T) + a_cb[i];
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int piece = wave * B_IT + i;
            const int n = n0 + piece * RPI + lrow;
            b_voff[i] = (piece < B_PIECES && n < Nf) ? (unsigned)n * (unsigned)(KS * KS * Cp) * (unsigned)sizeof(T) + b_cb[i] : Y2_OOB;
        }
        i_kt = kt;
        i_chunk0 = (kt / (TAPS * kpc)) * KC;
        i_tap = (kt / kpc) % TAPS;
        i_c0 = i_chunk0 + (kt % kpc) * BK;
        i_dh = i_tap / KS - PAD;
        i_dw = i_tap % KS - PAD;
    };
    auto issue_next = [&]() {
        const unsigned tapbit = 1u << i_tap;
        const unsigned offA = (unsigned)((i_dh * W + i_dw) * ldp + i_c0) * (unsigned)sizeof(T);   // may wrap: added mod 2^32
        const unsigned offB = (unsigned)(i_chunk0 * TAPS + i_tap * KC + (i_c0 - i_chunk0)) * (unsigned)sizeof(T);
        unsigned char *As = smem + i_stage * STAGE;
        unsigned char *Bs = As + BM * ROWB;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const unsigned voff = (a_mask[i] & tapbit) ? a_voff[i] + offA : Y2_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcP, (__attribute__((address_space(3))) void *)(As + (wave * A_IT + i) * 1024), 16, voff, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const unsigned voff = b_voff[i] + offB;            // an OOB row stays out of range
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcF, (__attribute__((address_space(3))) void *)(Bs + (wave * B_IT + i) * 1024), 16, voff, 0, 0, 0);
        }
        ++iu;
        i_stage = (i_stage + 1 == NSTAGE) ? 0 : i_stage + 1;
        if (++i_kt == nk) {                                    // the cursor leaves this tile: descriptors of the next one
            if (iu < su_end) setup_issue(iu);
            return;
        }
        i_c0 += BK;
        if (i_c0 >= i_chunk0 + KC) {
            if (++i_tap == TAPS) { i_tap = 0; i_chunk0 += KC; i_dh = -PAD; i_dw = -PAD; }
            else if (++i_dw > PAD) { i_dw = -PAD; ++i_dh; }
            i_c0 = i_chunk0;
        }
    };
    setup_issue(su);
#pragma unroll
    for (int st = 0; st < NSTAGE - 1; ++st)
        if (iu < su_end) issue_next();

    const int frow = lane & 31;
    const int fsw = (frow / RPL) % CH;
    int c_stage = 0;
    long cu = su;
    unsigned *flags = sk_flags;
    float *slots = Oacc;
    constexpr int SLOT = BM * BN;
    const bool stats = bn_part != nullptr;
    const bool stats_unique = MT * WGM <= Y2_BN_PART_ROWS;
    while (cu < su_end) {
        const int t = (int)(cu / nk);
        const int kt_beg = (int)(cu - (long)t * nk);
        const int kt_end = (int)min((long)nk, kt_beg + (su_end - cu));
        const int nt = t / MT, mt = t - nt * MT;
        const int m0 = mt * BM, n0 = nt * BN;
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int kt = kt_beg; kt < kt_end; ++kt, ++cu) {
            const int ahead = min(NSTAGE - 2, (int)(iu - 1 - cu));
            if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LOADS) : "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();      // unit cu complete in LDS for every wave; the previous unit's buffer is free
            if (iu < su_end) issue_next();
            const unsigned char *As = smem + c_stage * STAGE + (wm * TM * 32 + frow) * ROWB;
            const unsigned char *Bs = smem + c_stage * STAGE + (BM + wn * TN * 32 + frow) * ROWB;
            c_stage = (c_stage + 1 == NSTAGE) ? 0 : c_stage + 1;
#pragma unroll
            for (int kk = 0; kk < BK / Mma<T>::KSTEP; ++kk) {
                typename Mma<T>::Frag af[TM], bf[TN];
                int boff;
                if constexpr (sizeof(T) == 2) {
                    boff = ((kk * 2 + (lane >> 5)) ^ fsw) * 16;
                } else {
                    const int e = kk * 2 + (lane >> 5);
                    boff = (((e >> 2) ^ fsw) * 16) + (e & 3) * 4;
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const typename Mma<T>::Frag *>(As + i * 32 * ROWB + boff);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const typename Mma<T>::Frag *>(Bs + j * 32 * ROWB + boff);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = Mma<T>::mma(af[i], bf[j], acc[i][j]);
            }
        }
        if (!whole_tiles) {
            // stream-K hand-off (protocol and reasoning: SPLITK == 2 in the kernel above)
            const __amdgpu_buffer_rsrc_t rsrcS = __builtin_amdgcn_make_buffer_rsrc(slots, 0, (unsigned)((size_t)G * SLOT * sizeof(float)), 0x00020000);
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
                // every slot store acknowledged -- which also retires this wave's DMAs in flight (they are older or interleaved):
                // the counted waits of the next segment stay conservative, never short
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) __hip_atomic_store(flags + wx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                continue;
            }
            if (kt_end < nk) {
                const long tile_end = (long)(t + 1) * nk;
                long covered = cu;
                for (int p = wx + 1; covered < tile_end; ++p) {
                    if (tid == 0) {
                        while (__hip_atomic_load(flags + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(8);
                        __hip_atomic_store(flags + p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
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
        // epilogue (C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)) + batch-norm partial sums
        const bool tail = m0 + BM > M;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 32 + (lane & 31);
            if (n >= Nf) continue;
            const float bv = bias ? bias[n] : 0.f;
            const float sh = stats ? bn_shift[n] : 0.f;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int mb = m0 + (wm * TM + i) * 32 + 4 * (lane >> 5);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mb + (r & 3) + 8 * (r >> 2);
                    if (!tail || m < M) {
                        float v = acc[i][j][r] + bv;
                        if (act_alpha != 1.0f) v = fmaxf(v, act_alpha * v);
                        const T o = (T)v;
                        O[(long)m * ldo + n] = o;
                        const float d = (float)o - sh;
                        s1 += d;
                        s2 += d * d;
                    }
                }
            }
            if (stats) {
                s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 32, 64);
                if (lane < 32) {
                    const int slot = (mt * WGM + wm) & (Y2_BN_PART_ROWS - 1);
                    float *p1 = bn_part + (long)slot * Nf + n, *p2 = bn_part + (long)(Y2_BN_PART_ROWS + slot) * Nf + n;
                    if (stats_unique) { *p1 = s1; *p2 = s2; }
                    else { unsafeAtomicAdd(p1, s1); unsafeAtomicAdd(p2, s2); }
                }
            }
        }
    }
}

