// Shared device/host helpers for the gfx950 YOLOv2 kernels (wave64, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../include/yolo2_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

void yolo2_set_error(const char *fmt, ...);

#if defined(__HIP_DEVICE_COMPILE__)
// ds_read_b64_tr_b16 through inline asm.  The builtin form is a compiler-visible LDS read: while an LDS-DMA (buffer_load ... lds)
// is in flight hipcc 7.2 puts `s_waitcnt vmcnt(0)` in front of every such read (it cannot tell the DMA's destination from the
// read's address), which drains the whole DMA ring each K step -- measured in round 1's filter-gradient kernel, whose counted
// vmcnt waits never had anything left to count.  An asm read is invisible to that pass; its completion is waited for by hand
// with y2_lgkm_wait*, whose "+v" operands tie the wait to the destination registers (nothing that uses them can be scheduled
// above it; cdna_hip_programming.md section 5.7 form (ii)).  LDS reads of one wave return in order.
__device__ __forceinline__ u32x2 y2_tr16_read(unsigned lds_addr) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(lds_addr) : "memory");
    return v;
}
template <int OFF> __device__ __forceinline__ u32x2 y2_tr16_read_off(unsigned lds_addr) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=&v"(v) : "v"(lds_addr), "n"(OFF) : "memory");     // (&: the address register is re-used by the next read)
    return v;
}
// wait until at most N LDS operations of this wave are outstanding; a, b (and c, d): the registers whose data this wait covers
template <int N> __device__ __forceinline__ void y2_lgkm_wait2(u32x2 &a, u32x2 &b) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void y2_lgkm_wait4(u32x2 &a, u32x2 &b, u32x2 &c, u32x2 &d) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}
__device__ __forceinline__ bf16x8 y2_frag16(u32x2 lo, u32x2 hi) {
    const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ unsigned y2_lds_addr(const void *p) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}
// Sum of v over the lanes that share lane % G (G = 4 .. 64 lanes per group), left in EVERY lane -- on the VALU: row rotations (DPP) inside the
// 16-lane rows, v_permlane16_swap / v_permlane32_swap across them (with both operands the same value, "swap the odd rows of the first with the even rows
// of the second" leaves {row 0, row 0, row 2, row 2} and {row 1, row 1, row 3, row 3}: their sum is the pair's total in both rows; likewise the halves).
// The ds_bpermute butterfly (__shfl_xor) this replaces goes through the LDS pipe: ~7.5 cycles per exchange and CU when eight waves do it at once.
// (inline asm: with the builtin hipcc 7.2 adds the first result to itself; the s_nop cover the VALU -> permlane hazards the compiler cannot see)
__device__ __forceinline__ float y2_rows_pair_sum(float v) {
    unsigned a = __builtin_bit_cast(unsigned, v), b = a;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
__device__ __forceinline__ float y2_halves_sum(float v) {
    unsigned a = __builtin_bit_cast(unsigned, v), b = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
template <int G> __device__ __forceinline__ float y2_lane_group_sum(float v) {
    static_assert(G == 4 || G == 8 || G == 16 || G == 32 || G == 64, "lanes per group");
    if (G <= 4) v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));      // row_ror:4
    if (G <= 8) v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));      // row_ror:8
    if (G <= 16) v = y2_rows_pair_sum(v);
    if (G <= 32) v = y2_halves_sum(v);
    return v;
}
// ... the same for a double: every exchange moves the two 32-bit halves (the partner's VALUE is added, so both operands of the swaps are copies)
__device__ __forceinline__ double y2_f64_from(unsigned lo, unsigned hi) { return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo); }
template <int G> __device__ __forceinline__ double y2_lane_group_sum_f64(double v) {
    static_assert(G == 4 || G == 8 || G == 16 || G == 32 || G == 64, "lanes per group");
    auto halves = [](double x, unsigned &lo, unsigned &hi) { const unsigned long long b = __builtin_bit_cast(unsigned long long, x); lo = (unsigned)b; hi = (unsigned)(b >> 32); };
    unsigned lo, hi;
    if (G <= 4) {
        halves(v, lo, hi);
        v += y2_f64_from((unsigned)__builtin_amdgcn_mov_dpp((int)lo, 0x124, 0xF, 0xF, true), (unsigned)__builtin_amdgcn_mov_dpp((int)hi, 0x124, 0xF, 0xF, true));
    }
    if (G <= 8) {
        halves(v, lo, hi);
        v += y2_f64_from((unsigned)__builtin_amdgcn_mov_dpp((int)lo, 0x128, 0xF, 0xF, true), (unsigned)__builtin_amdgcn_mov_dpp((int)hi, 0x128, 0xF, 0xF, true));
    }
    if (G <= 16) {      // rows 0 <-> 1, 2 <-> 3: after the swap with a copy, (a, b) = (even row's value, odd row's value) in both rows of a pair
        halves(v, lo, hi);
        unsigned lo2 = lo, hi2 = hi;
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\ts_nop 1" : "+v"(lo), "+v"(lo2), "+v"(hi), "+v"(hi2));
        v = y2_f64_from(lo, hi) + y2_f64_from(lo2, hi2);
    }
    if (G <= 32) {
        halves(v, lo, hi);
        unsigned lo2 = lo, hi2 = hi;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\ts_nop 1" : "+v"(lo), "+v"(lo2), "+v"(hi), "+v"(hi2));
        v = y2_f64_from(lo, hi) + y2_f64_from(lo2, hi2);
    }
    return v;
}
#else
__device__ inline unsigned y2_lds_addr(const void *) { return 0u; }      // (host pass of a __global__ body)
template <int G> __device__ inline float y2_lane_group_sum(float v) { return v; }
template <int G> __device__ inline double y2_lane_group_sum_f64(double v) { return v; }
#endif

// first-layer direct convolution (conv_first.hip), used by yolo2_conv2d / yolo2_conv2d_wgrad when the shape matches
bool y2_first_layer_shape(int Cp, int ldp, int Nf, int ldo, int ksize);
int y2_first_layer_fwd(const void *P, const void *F, void *O, int B, int H, int W, int dtype, hipStream_t st,
                       const float *bn_shift = nullptr, float *bn_part = nullptr);
int y2_first_layer_wgrad(const void *X, const void *dY, float *dW, int B, int H, int W, int Cin, int dtype, hipStream_t st);

// batch-norm partial sums produced by the convolution epilogue: [2][Y2_BN_PART_ROWS][C] f32 (elementwise.hip finalises)
#define Y2_BN_PART_ROWS YOLO2_BN_PART_ROWS
// the 2 x Y2_BN_PART_ROWS partial rows of a fused BN-backward data gradient -> dgamma (plane 0), dbeta (plane 1); rows zeroed again
int y2_bn_part_to_grads(float *part, int C, float *dgamma, float *dbeta, hipStream_t st);
int y2_colsum_into(const void *Y, int ld, long M, int C, const float *shift, float *part, int dtype, hipStream_t st, int *rows_used = nullptr);

#define Y2_CHECK_ARG(cond)                                                          \
    do {                                                                            \
        if (!(cond)) {                                                              \
            yolo2_set_error("%s: argument check failed: %s", __func__, #cond);      \
            return YOLO2_E_ARG;                                                     \
        }                                                                           \
    } while (0)

#define Y2_CHECK_LAUNCH()                                                           \
    do {                                                                            \
        hipError_t e_ = hipGetLastError();                                          \
        if (e_ != hipSuccess) {                                                     \
            yolo2_set_error("%s: HIP error: %s", __func__, hipGetErrorString(e_));  \
            return YOLO2_E_LAUNCH;                                                  \
        }                                                                           \
    } while (0)

// K order of the MFMA filter operands (Ffwd / Fdgr rows): channel chunk outermost, tap, channel inside the chunk.
// With taps outermost a 3x3 layer re-reads its input tile 9 times with the whole channel extent in between, which
// overflows the 4 MiB XCD L2 in the 13x13 stages (measured 15x the algorithmic fabric traffic); with 64-channel
// chunks the 9 shifted re-reads of a 128-pixel x 64-channel slab are back to back.  Rows whose channel extent is
// not a multiple of 64 keep a single chunk (= tap-major).
#define Y2_KCHUNK 64
__host__ __device__ static inline int y2_kchunk(int ld, int taps) { return (taps > 1 && ld % Y2_KCHUNK == 0) ? Y2_KCHUNK : ld; }
__host__ __device__ static inline long y2_filter_koff(int tap, int c, int ld, int taps) {
    const int kc = y2_kchunk(ld, taps);
    return (long)(c / kc) * taps * kc + (long)tap * kc + c % kc;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
// The library's A/B switches are integers read once from the environment; unset = the measured default.
static inline int y2_env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

// 16-byte vector of T: 4 floats or 8 bf16
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    f32x4 v;
    __device__ __forceinline__ float get(int i) const { return v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = x; }
};
template <> struct Vec16<bf16> {
    static constexpr int N = 8;
    bf16x8 v;
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = (bf16)x; }
};

template <typename T> __device__ __forceinline__ Vec16<T> ld16(const T *p) {
    Vec16<T> r;
    r.v = *reinterpret_cast<const decltype(r.v) *>(p);
    return r;
}
template <typename T> __device__ __forceinline__ void st16(T *p, const Vec16<T> &x) {
    *reinterpret_cast<decltype(x.v) *>(p) = x.v;
}
template <typename T> __device__ __forceinline__ Vec16<T> zero16() {
    Vec16<T> r;
#pragma unroll
    for (int i = 0; i < Vec16<T>::N; ++i) r.set(i, 0.f);
    return r;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

#define Y2_DISPATCH_DTYPE(dtype, ...)                                   \
    do {                                                                \
        if ((dtype) == YOLO2_F32) { typedef float T; __VA_ARGS__; }     \
        else if ((dtype) == YOLO2_BF16) { typedef bf16 T; __VA_ARGS__; } \
        else { yolo2_set_error("%s: bad dtype %d", __func__, (int)(dtype)); return YOLO2_E_ARG; } \
    } while (0)
