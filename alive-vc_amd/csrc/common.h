// Shared helpers for the gfx950 kernels of libalive_vc.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>
#include <initializer_list>
#include "../../include/alive_vc.h"

void alive_set_error(const char* fmt, ...);

#define ALIVE_CHECK_ARG(cond, ...)                                   \
    do {                                                              \
        if (!(cond)) {                                                \
            alive_set_error(__VA_ARGS__);                             \
            return ALIVE_ERR_ARG;                                     \
        }                                                             \
    } while (0)

#define ALIVE_CHECK_LAUNCH(what)                                      \
    do {                                                              \
        hipError_t e_ = hipGetLastError();                            \
        if (e_ != hipSuccess) {                                       \
            alive_set_error("%s: %s", what, hipGetErrorString(e_));   \
            return ALIVE_ERR_LAUNCH;                                  \
        }                                                             \
    } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute: one opt-in flag per launch site and device
// (a process that moves to a second GPU must opt in there too), safe against concurrent host threads.
struct LdsOptIn {
    std::atomic<uint64_t> done{0};
    hipError_t ensure(std::initializer_list<const void*> fns, int bytes) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        const uint64_t bit = 1ull << (dev & 63);
        if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
        for (const void* f : fns) {
            e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e != hipSuccess) return e;
        }
        done.fetch_or(bit, std::memory_order_release);
        return hipSuccess;
    }
};

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Bump allocator over the caller-provided workspace (no hipMalloc in any entry point).
struct Arena {
    char* base;
    size_t off;
    explicit Arena(void* p) : base((char*)p), off(0) {}
    template <typename T>
    T* take(size_t n) {
        off = align_up(off, 256);
        T* p = (T*)(base + off);
        off += n * sizeof(T);
        return p;
    }
    size_t used() const { return align_up(off, 256); }
};

// The accumulation-chain gap (DESIGN.md 3.2b', tools/repro_filter_block64.sh): when the first MFMA of a NEW accumulation chain is
// issued after the last MFMA of a finished chain whose accumulator is read only LATER, beside the new chain's MFMAs, the finished
// chain's last-written register can come back wrong on gfx950 (hipcc 7.2 provides wait states for the first read only) unless the
// wave idles >= 8 wait states between the two chains.  The scheduling barriers pin the nop between the two chains: an asm
// statement alone is not ordered against MFMA builtins (hipcc hoisted the first MFMA of a step above it in filter_small.hip).
// tools/mfma_hazard_scan.py::chain_gap_scan checks every listing for the pattern (tests/test_host_logic.py).
#define ALIVE_CHAIN_GAP(NOP)                                \
    do {                                                    \
        __builtin_amdgcn_sched_barrier(0);                  \
        asm volatile("s_nop " #NOP ::: "memory");           \
        __builtin_amdgcn_sched_barrier(0);                  \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifdef __HIPCC__
// ---- plain fp16 operands (round 5: AliveConv.precision 3, AliveGemm.planes 1) ----
// One plane, one MFMA per product: fp16 (11 significand bits, 2^-12 per operand) instead of the bf16 planes' 8.  The conversion rounds to
// nearest even and SATURATES at +-65504 (a value beyond fp16's range must not become an infinity inside a GEMM); magnitudes below 6.1e-5
// are fp16 subnormals (absolute resolution 6e-8).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// Saturations are COUNTED (one counter per translation unit, summed by alive_f16_saturations): a checkpoint or an input whose activations
// leave fp16's range must not pass silently -- module/pipeline.py clears the counters when a conversion starts, reads them when it ends
// and repeats the batch on bf16 planes (fp32's range) when they are not zero.
// The getter SYNCHRONISES THE DEVICE before it reads (the callers' streams are non-blocking torch streams: a read on the null stream
// alone could overtake GEMMs still running), reads the CURRENT device's counter, and returns -1 when any runtime call fails; the clear
// form is an asynchronous 4-byte copy in the order of `stream` (graph-capturable, no synchronisation).
static __device__ unsigned alive_f16_sat_count;
#define ALIVE_F16_SAT_GETTER(NAME)                                                                                          \
    int NAME(int reset) {                                                                                                   \
        unsigned v = 0, z = 0;                                                                                              \
        if (hipDeviceSynchronize() != hipSuccess) return -1;                                                                \
        if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(alive_f16_sat_count), sizeof(v)) != hipSuccess) return -1;                   \
        if (reset && hipMemcpyToSymbol(HIP_SYMBOL(alive_f16_sat_count), &z, sizeof(z)) != hipSuccess) return -1;            \
        return (int)(v > 0x7fffffffu ? 0x7fffffffu : v);                                                                    \
    }                                                                                                                       \
    int NAME##_clear(void* stream) {                                                                                        \
        void* p = nullptr;                                                                                                  \
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(alive_f16_sat_count)) != hipSuccess) return -1;                              \
        return hipMemsetAsync(p, 0, sizeof(unsigned), (hipStream_t)stream) == hipSuccess ? 0 : -1;                          \
    }
// count = false: the values are known to be don't-cares (the halo columns of a fused tile hold whatever their dependency cone left)
__device__ __forceinline__ unsigned pack_f16x2(float a, float b, bool count = true) {
    typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
    if (__builtin_expect(count && (fabsf(a) > 65504.0f || fabsf(b) > 65504.0f), 0)) atomicAdd(&alive_f16_sat_count, 1u);
    a = __builtin_amdgcn_fmed3f(a, -65504.0f, 65504.0f);
    b = __builtin_amdgcn_fmed3f(b, -65504.0f, 65504.0f);
    const f16x2_t h = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ f32x16 mfma_f16(bf16x8 a, bf16x8 b, f32x16 c) {      // the operand registers hold fp16 bits
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// exact-erf GELU in the association order of ATen's CPU kernel: (0.5*x) * (1 + erf(x/sqrt2))
__device__ __forceinline__ float gelu_erf(float x) {
    return (0.5f * x) * (1.0f + erff(x * 0.70710678118654752440f));
}

// F.interpolate(mode='linear', align_corners=False) source coordinate:
//   src = fmaf(scale, i + 0.5, -0.5) clamped at 0 (single rounding), see DESIGN.md "interp"
struct Lerp {
    int i0, i1;
    float w0, w1;
};
__device__ __forceinline__ Lerp lerp_coord(int i, float ratio, int in_len) {
    float src = fmaf(ratio, (float)i + 0.5f, -0.5f);
    src = src < 0.0f ? 0.0f : src;
    int i0 = (int)src;
    if (i0 > in_len - 1) i0 = in_len - 1;
    Lerp r;
    r.i0 = i0;
    r.i1 = i0 + (i0 < in_len - 1 ? 1 : 0);
    r.w1 = src - (float)i0;
    r.w0 = 1.0f - r.w1;
    return r;
}
// value = fma(w0, a, round(w1*b))  -- bit-exact with ATen's CPU upsample_linear1d
__device__ __forceinline__ float lerp_apply(const Lerp& l, float a, float b) {
    return fmaf(l.w0, a, l.w1 * b);
}

__device__ __forceinline__ unsigned short f32_to_bf16_rn(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
#endif
