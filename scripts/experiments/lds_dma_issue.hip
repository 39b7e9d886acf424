// How long does a wave take to ISSUE a run of 1 KiB LDS-DMA loads (buffer_load_dwordx4 ... lds)?  (a) a new M0 (LDS base) per instruction,
// (b) one M0 and the instruction's immediate offset (4 pieces per M0), (c) as (a) with the addresses computed by two magic divisions per piece.
// s_memtime ticks per piece, one workgroup per CU, WAVES waves issuing at once.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void *lds_void_ptr;
template <int VARIANT, int NP>
__global__ __launch_bounds__(512) void k(const unsigned short *X, unsigned bytes, unsigned long long *out, int waves, unsigned m, unsigned sft) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= waves) return;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(X), 0, bytes, 0x00020000);
    const unsigned base = (blockIdx.x * 8 + wave) * (NP * 1024u);
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (VARIANT == 0) {
#pragma unroll
        for (int p = 0; p < NP; ++p)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_ptr)(smem + (wave * NP + p) * 1024), 16, base + p * 1024 + lane * 16, 0, 0, 0);
    } else if (VARIANT == 1) {
#pragma unroll
        for (int p = 0; p < NP; p += 4) {
            lds_void_ptr l = (lds_void_ptr)(smem + (wave * NP + p) * 1024);
            const unsigned vo = base + p * 1024 + lane * 16;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, l, 16, vo, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, l, 16, vo, 0, 1024, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, l, 16, vo, 0, 2048, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, l, 16, vo, 0, 3072, 0);
        }
    } else {
        for (int p = 0; p < NP; ++p) {
            const unsigned q = base / 64 + p * 16 + (lane >> 2);
            const unsigned R = __umulhi(q, m) >> sft, img = __umulhi(R, m) >> sft;
            const unsigned mm = q - R - img * 208u + R + img * 208u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_ptr)(smem + (wave * NP + p) * 1024), 16, mm * 64 + (lane & 3) * 16, 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_readcyclecounter();
    if (lane == 0) { out[(blockIdx.x * 8 + wave) * 2] = t1 - t0; out[(blockIdx.x * 8 + wave) * 2 + 1] = t2 - t0; }
}
int main() {
    const int NP = 16, G = 256;
    unsigned bytes = G * 8 * NP * 1024;
    unsigned short *X; unsigned long long *out;
    hipMalloc(&X, bytes); hipMemset(X, 1, bytes); hipMalloc(&out, G * 8 * 2 * 8);
    std::vector<unsigned long long> h(G * 8 * 2);
    for (int waves : {1, 4, 8})
        for (int v = 0; v < 3; ++v) {
            for (int rep = 0; rep < 2; ++rep) {
                hipMemset(out, 0, G * 8 * 2 * 8);
                const size_t lds = 8 * NP * 1024;
                if (v == 0) { hipFuncSetAttribute((const void *)k<0, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); k<0, 16><<<G, 512, lds>>>(X, bytes, out, waves, 0x4e6b9d43u, 6); }
                if (v == 1) { hipFuncSetAttribute((const void *)k<1, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); k<1, 16><<<G, 512, lds>>>(X, bytes, out, waves, 0x4e6b9d43u, 6); }
                if (v == 2) { hipFuncSetAttribute((const void *)k<2, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); k<2, 16><<<G, 512, lds>>>(X, bytes, out, waves, 0x4e6b9d43u, 6); }
                hipDeviceSynchronize();
            }
            hipMemcpy(h.data(), out, G * 8 * 2 * 8, hipMemcpyDeviceToHost);
            double a = 0, b = 0; int n = 0;
            for (int i = 0; i < G * 8; ++i) if (h[2 * i]) { a += h[2 * i]; b += h[2 * i + 1]; ++n; }
            printf("waves %d variant %d: issue %.0f ticks per piece, issue+landed %.0f ticks total (%d pieces per wave)\n", waves, v, a / n / NP, b / n, NP);
        }
    return 0;
}
