// How many shader cycles does one K step (64 channels of one tap) of a 256 x 128 output tile cost, as a function of WHO reads WHAT from LDS WHEN?
// Isolates the inner loop of the 3x3 forward / data-gradient kernel (conv_pp.hip) from its DMA, prologue and epilogue: the operand tiles sit in LDS
// (pixels 256 rows x 128 B, filters 128 rows x 128 B, the kernel's XOR swizzle), every step re-reads them and issues the step's MFMAs.
//   P    8 waves (2 per SIMD) of 64 x 64, the ping-pong schedule of conv_pp.hip: LOAD (16 ds_read_b128) | barrier | MFMA (16) | barrier, groups half a step apart
//   S4   4 waves (1 per SIMD) of 128 x 64: per 16-k group 6 reads for the NEXT group issued ahead of the 8 MFMAs of the current one (two fragment sets)
//   S4i  as S4 with each read placed behind one MFMA of the current group (reads inside the MFMA stream)
//   T4   4 waves of 64 x 128 (2 pixel + 4 filter fragments per group)
//   S8   8 waves (2 per SIMD) of 128 x 64 on a 256 x 256 tile (filters 256 rows), no ping-pong: both waves of a SIMD run the S4 stream
//   L4   8 waves: waves 0-3 run the S4i stream (no DMA instruction of their own), waves 4-7 (one per SIMD) are LOADERS: 5 LDS-DMA pieces per step each, a counted
//        vmcnt wait, the step's barrier -- producer / consumer specialisation
//   each with BAR = 0 / 1: no barrier in the loop / one s_barrier per K step (what a DMA ring hand-over needs)
// Ideal: 32 MFMAs x 32 cycles = 1024 cycles per SIMD and step for the 256 x 128 tile (S8: 2048 for twice the work; reported per 256 x 128).
// build: hipcc --offload-arch=gfx950 -O3 scripts/experiments/mfma_lds_loop.hip -o mfma_lds_loop ; run on the GPU box: ./mfma_lds_loop
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef const __attribute__((address_space(3))) bf16x8 *lds_frag_ptr;

__device__ __forceinline__ unsigned lds_addr(const void *p) { return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void *)p; }
// row r of a tile of 128-byte rows, 16-k group kk, lane half h: byte offset with the kernel's swizzle ((r >> 1) & 7)
__device__ __forceinline__ unsigned frag_off(int r, int kk, int h) { return (unsigned)(r * 128 + ((((kk << 1) | h) ^ ((r >> 1) & 7)) << 4)); }

// MODE 0 = P, 1 = S4, 2 = S4i, 3 = T4, 4 = S8
// DMA = 1: every step also streams 20 KiB from global memory (L2-resident, 4 MiB window) into an LDS ring by LDS-DMA -- 16 KiB of filters + ~4 KiB of
// halo, what a K step of the real kernel moves -- issued by all waves (P: 3 per wave behind the reads; S4 / S4i / T4: 5 per wave inside the first two
// 16-k groups), with a counted vmcnt wait three steps behind
template <int MODE, int BAR, int DMA>
__global__ __launch_bounds__(MODE == 0 || MODE == 4 || MODE == 5 ? 512 : 256) void loop_kernel(unsigned long long *out, float *sink, int iters, const unsigned char *gsrc) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    constexpr int BROWS = MODE == 4 ? 256 : 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < (256 + BROWS) * 128 / 4; i += blockDim.x) reinterpret_cast<unsigned *>(smem)[i] = 0x3f803f80u + (unsigned)(i & 7);      // small bf16 values
    __syncthreads();
    const unsigned a0 = lds_addr(smem), b0 = a0 + 256 * 128;
    const int r31 = lane & 31, h = lane >> 5;
    constexpr int TM = MODE == 0 ? 2 : (MODE == 3 ? 2 : 4), TN = MODE == 0 ? 2 : (MODE == 3 ? 4 : 2);
    constexpr bool LOADERS = MODE == 5;
    // wave -> (wm, wn): P 4 x 2 of 64 x 64; S4 2 x 2 of 128 x 64; T4 4 x 1 of 64 x 128; S8 2 x 4 of 128 x 64
    const int wn_cnt = MODE == 0 ? 2 : (MODE == 3 ? 1 : (MODE == 4 ? 4 : 2));
    const int wm = (wave & (LOADERS ? 3 : 7)) / wn_cnt, wn = wave % wn_cnt;
    unsigned aaddr[4][TM], baddr[4][TN];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int i = 0; i < TM; ++i) aaddr[kk][i] = a0 + frag_off(wm * TM * 32 + i * 32 + r31, kk, h);
#pragma unroll
        for (int j = 0; j < TN; ++j) baddr[kk][j] = b0 + frag_off(wn * TN * 32 + j * 32 + r31, kk, h);
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    __syncthreads();
    // LDS-DMA ring behind the operand tiles: 4 stages of 20 KiB; piece p of a step = 1 KiB
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(gsrc), 0, 4u << 20, 0x00020000);
    typedef __attribute__((address_space(3))) void *lds_void_ptr;
    unsigned char *ring = smem + (256 + BROWS) * 128;
    constexpr int NWV = (MODE == 0 || MODE == 4 || MODE == 5) ? 8 : 4, PPW = MODE == 0 || MODE == 4 ? 3 : 5;      // pieces per wave and step (24 / 20 KiB per step)
    unsigned gvoff = (unsigned)((blockIdx.x * 37 + wave) * 1024 + lane * 16) & ((4u << 20) - 1);
    auto dma = [&](int it, int k) {
        if (DMA) {
            unsigned char *dst = ring + ((it & 3) * (NWV * PPW) + wave * PPW + k) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_ptr)dst, 16, gvoff, 0, 0, 0);
            gvoff = (gvoff + 8192u * 3u) & ((4u << 20) - 1);
        }
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (LOADERS && wave >= 4) {
        for (int it = 0; it < iters; ++it) {
            dma(it, 0); dma(it, 1); dma(it, 2); dma(it, 3); dma(it, 4);
            asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    } else
    if constexpr (MODE == 0) {
        bf16x8 fa[4][TM], fb[4][TN];
        if (wave >= 4) __builtin_amdgcn_s_barrier();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[kk][i] = *(lds_frag_ptr)(uintptr_t)aaddr[kk][i];
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[kk][j] = *(lds_frag_ptr)(uintptr_t)baddr[kk][j];
            }
            __builtin_amdgcn_sched_barrier(0);
            dma(it, 0); dma(it, 1); dma(it, 2);
            if (DMA) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk][i], fb[kk][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        if (wave < 4) __builtin_amdgcn_s_barrier();
    } else {
        bf16x8 fa[2][TM], fb[2][TN];
        auto load = [&](int kk, int set) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[set][i] = *(lds_frag_ptr)(uintptr_t)aaddr[kk][i];
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[set][j] = *(lds_frag_ptr)(uintptr_t)baddr[kk][j];
        };
        load(0, 0);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int set = kk & 1;
                if constexpr (MODE != 2 && MODE != 5) {
                    load((kk + 1) & 3, set ^ 1);
                    if (kk == 0) { dma(it, 0); dma(it, 1); dma(it, 2); }
                    if (kk == 1) { dma(it, 3); dma(it, 4); }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[set][i], fb[set][j], acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    // one read behind each of the first TM + TN MFMAs of the group
                    int n = 0;
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[set][i], fb[set][j], acc[i][j], 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);
                            if (n < TM) fa[set ^ 1][n] = *(lds_frag_ptr)(uintptr_t)aaddr[(kk + 1) & 3][n];
                            else if (n < TM + TN) fb[set ^ 1][n - TM] = *(lds_frag_ptr)(uintptr_t)baddr[(kk + 1) & 3][n - TM];
                            else if (!LOADERS && kk == 0 && n == TM + TN) { dma(it, 0); dma(it, 1); dma(it, 2); }
                            else if (!LOADERS && kk == 1 && n == TM + TN) { dma(it, 3); dma(it, 4); }
                            __builtin_amdgcn_sched_barrier(0);
                            ++n;
                        }
                }
            }
            if (DMA && !LOADERS) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            if (BAR) __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 123.456f) sink[0] = s;
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE, int BAR, int DMA>
static void run(const char *name, unsigned long long *out, float *sink, int cus, const unsigned char *gsrc) {
    const int iters = 4000, threads = (MODE == 0 || MODE == 4 || MODE == 5) ? 512 : 256, waves = MODE == 5 ? 4 : threads / 64;
    const size_t lds = (size_t)(256 + (MODE == 4 ? 256 : 128)) * 128 + 96 * 1024;      // operand tiles + the DMA ring (4 x 24 KiB); > 80 KB: one workgroup per CU
    hipFuncSetAttribute((const void *)loop_kernel<MODE, BAR, DMA>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        loop_kernel<MODE, BAR, DMA><<<cus, threads, lds>>>(out, sink, iters, gsrc);
        hipEventRecord(b);
        hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
    }
    unsigned long long h[256 * 8];
    hipMemcpy(h, out, sizeof(unsigned long long) * cus * 8, hipMemcpyDeviceToHost);
    double mean = 0, mx = 0;
    for (int i = 0; i < cus; ++i)
        for (int w = 0; w < waves; ++w) { const double c = (double)h[i * 8 + w] / iters; mean += c; if (c > mx) mx = c; }
    mean /= cus * waves;
    const double work = MODE == 4 ? 2.0 : 1.0;
    const double flop = (double)cus * iters * 256.0 * 128 * 64 * 2 * work;
    printf("%-4s barrier %d DMA %d: %7.1f cycles per step and 256x128 tile (max wave %7.1f)   %6.1f ns per step   %6.0f TFLOP/s   clock %.2f GHz\n", name, BAR, DMA, mean / work, mx / work,
           ms * 1e6 / iters / work, flop / ms * 1e-9, mean / (ms * 1e6 / iters));
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount > 256 ? 256 : p.multiProcessorCount;
    unsigned long long *out;
    float *sink;
    hipMalloc(&out, sizeof(unsigned long long) * 256 * 8);
    hipMalloc(&sink, 64);
    printf("%s, %d CUs; ideal 1024 cycles per step (32 MFMAs of 32 cycles per SIMD)\n", p.gcnArchName, cus);
    unsigned char *gsrc;
    hipMalloc(&gsrc, 4u << 20);
    hipMemset(gsrc, 0x3f, 4u << 20);
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 1, 0>("P", out, sink, cus, gsrc);
        run<0, 1, 1>("P", out, sink, cus, gsrc);
        run<1, 1, 0>("S4", out, sink, cus, gsrc);
        run<1, 1, 1>("S4", out, sink, cus, gsrc);
        run<2, 0, 0>("S4i", out, sink, cus, gsrc);
        run<2, 1, 0>("S4i", out, sink, cus, gsrc);
        run<2, 1, 1>("S4i", out, sink, cus, gsrc);
        run<3, 1, 0>("T4", out, sink, cus, gsrc);
        run<3, 1, 1>("T4", out, sink, cus, gsrc);
        run<4, 1, 0>("S8", out, sink, cus, gsrc);
        run<4, 1, 1>("S8", out, sink, cus, gsrc);
        run<5, 1, 1>("L4", out, sink, cus, gsrc);
    }
    return 0;
}
