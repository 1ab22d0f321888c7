// Stand-alone probe for the accumulation-chain hazard of DESIGN.md 3.2b' (gfx950, hipcc 7.2):
//   chain A (NA dependent v_mfma_f32_32x32x16_bf16 into a[0:15]) -> [GAP wait states] -> chain B into a[16:31], with A's sixteen
//   accumulator registers read (v_accvgpr_read) only LATER, one group of four behind every fifth MFMA of chain B.
// Every lane compares the late reads with the same chain A computed and read at once (no chain B behind it).
// RESULT (round 5, one MI355X, 256 blocks x 500 repetitions): 0 mismatches at gap 0, 8 and 16 -- this minimal shape (compiler MFMA
// builtins, order pinned by scheduling barriers, an LDS write and a block barrier between the chains) does NOT reproduce the defect of the
// fused 64-channel FilterBlock; the reproducer of record stays the full kernel (tools/repro_filter_block64.sh, -DALIVE_FB64_NO_CHAIN_GAP).
// What the first version of this probe did show: with the MFMAs written as INLINE ASM hipcc knows nothing of their latency and copies the
// accumulator right behind the chain without wait states -- registers 11..15 came back stale in 70-100 % of the lanes at every gap.  That is
// a property of inline-asm MFMAs, not of the hardware, and the reason no kernel of this library issues its MFMAs through asm.
//   hipcc -O3 --offload-arch=gfx950 tools/repro/mfma_chain_hazard.hip -o /tmp/chain_hazard && /tmp/chain_hazard [gap]
// prints the number of (lane, register) mismatches per accumulator register over all blocks and repetitions.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int GAP>
__global__ __launch_bounds__(256, 1) void probe(const unsigned short* __restrict__ rnd, int reps, unsigned* __restrict__ bad) {
    const int tid = threadIdx.x;
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = *(const bf16x8*)(rnd + ((tid * 4 + i) * 8) % 32768);
        b[i] = *(const bf16x8*)(rnd + ((tid * 4 + i + 4096) * 8) % 32768);
    }
    __shared__ float lds[256 * 4];
    for (int rep = 0; rep < reps; ++rep) {
        // Everything through the compiler's own MFMA builtins (it then inserts the wait states it knows about); the ORDER is pinned by
        // scheduling barriers exactly as in filter_mid.hip: one region per k-step, the late reads of chain A spread over chain B's regions.
        f32x16 ref, accA, accB;
        for (int r = 0; r < 16; ++r) { ref[r] = (float)r; accA[r] = (float)r; accB[r] = 0.5f * r; }
#pragma unroll
        for (int s = 0; s < 20; ++s) ref = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s & 3], b[(s + rep) & 3], ref, 0, 0, 0);
        float want[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) want[r] = ref[r];
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(want[r]));          // read at once, in VGPRs
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 20; ++s) accA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s & 3], b[(s + rep) & 3], accA, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        lds[tid] = want[0];
        __syncthreads();
        if (GAP > 0) asm volatile("s_nop %0" ::"n"(GAP - 1));
        __builtin_amdgcn_sched_barrier(0);
        float got[16];
#pragma unroll
        for (int s = 0; s < 20; ++s) {
            accB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(s + 1) & 3], b[s & 3], accB, 0, 0, 0);
            accB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(s + 2) & 3], b[s & 3], accB, 0, 0, 0);
            accB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(s + 3) & 3], b[s & 3], accB, 0, 0, 0);
            if (s % 5 == 0) {
                const int g = s / 5;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    got[4 * g + e] = accA[4 * g + e] * 1.0f;
                    asm volatile("" : "+v"(got[4 * g + e]));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (__float_as_uint(got[r]) != __float_as_uint(want[r])) atomicAdd(&bad[r], 1u);
        float sink = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) sink += accB[r];
        if (sink == 12345.678f) bad[16] = 1;
    }
}

int main(int argc, char** argv) {
    const int gap = argc > 1 ? atoi(argv[1]) : 0;
    unsigned short* rnd; unsigned* bad;
    (void)hipMalloc(&rnd, 65536); (void)hipMalloc(&bad, 17 * 4); (void)hipMemset(bad, 0, 17 * 4);
    unsigned short h[32768];
    srand(1);
    for (int i = 0; i < 32768; ++i) h[i] = (unsigned short)(0x3c00 + (rand() & 0x3ff));      // bf16 values in [0.0078, 0.0156): sums stay finite
    (void)hipMemcpy(rnd, h, 65536, hipMemcpyHostToDevice);
    if (gap == 0) probe<0><<<256, 256>>>(rnd, 500, bad);
    else if (gap <= 8) probe<8><<<256, 256>>>(rnd, 500, bad);
    else probe<16><<<256, 256>>>(rnd, 500, bad);
    unsigned out[17];
    (void)hipMemcpy(out, bad, 17 * 4, hipMemcpyDeviceToHost);
    printf("gap %d wait states: mismatches per accumulator register:", gap == 0 ? 0 : (gap <= 8 ? 8 : 16));
    for (int r = 0; r < 16; ++r) printf(" %u", out[r]);
    printf("\n");
    return 0;
}
