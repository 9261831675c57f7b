// issue_rate.hip -- measured issue rates of the instruction classes k_best8 is made of (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 -o issue_rate tools/micro/issue_rate.hip ; run: ./issue_rate
// Each kernel runs `waves_per_simd` waves on every SIMD of every CU, each executing N independent
// instructions of one class in a loop; the rate is reported per SIMD (VALU) or per CU (SALU) per clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP 64
template <int KIND>
__global__ void __launch_bounds__(64) k(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    uint32_t s0 = seed, s1 = seed * 3, s2 = seed * 5, s3 = seed * 7;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < REP / 8; r++) {
            if (KIND == 0) {   // v_and_b32 / v_add_u32 mix, 8 independent chains
                asm volatile("v_add_u32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n"
                             "v_add_u32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_xor_b32 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(a0));
            } else if (KIND == 1) {   // v_pk_add_u16
                asm volatile("v_pk_add_u16 %0, %0, %8\n v_pk_add_u16 %1, %1, %8\n v_pk_add_u16 %2, %2, %8\n v_pk_add_u16 %3, %3, %8\n"
                             "v_pk_add_u16 %4, %4, %8\n v_pk_add_u16 %5, %5, %8\n v_pk_add_u16 %6, %6, %8\n v_pk_add_u16 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(a0));
            } else if (KIND == 2) {   // s_add_u32 / s_and_b32, 4 independent chains x2
                asm volatile("s_add_u32 %0, %0, %4\n s_and_b32 %1, %1, %4\n s_add_u32 %2, %2, %4\n s_xor_b32 %3, %3, %4\n"
                             "s_add_u32 %0, %0, %4\n s_and_b32 %1, %1, %4\n s_add_u32 %2, %2, %4\n s_xor_b32 %3, %3, %4\n"
                             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "s"(seed) : "scc");
            } else if (KIND == 3) {   // v_readlane_b32 (VALU -> SGPR)
                asm volatile("v_readlane_b32 %0, %4, 1\n v_readlane_b32 %1, %4, 2\n v_readlane_b32 %2, %4, 3\n v_readlane_b32 %3, %4, 4\n"
                             "v_readlane_b32 %0, %5, 1\n v_readlane_b32 %1, %5, 2\n v_readlane_b32 %2, %5, 3\n v_readlane_b32 %3, %5, 4\n"
                             : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(a0), "v"(a1));
            } else if (KIND == 4) {   // 1:1 interleave of VALU and SALU (do the two units run side by side for one wave?)
                asm volatile("v_add_u32 %0, %0, %8\n s_add_u32 %4, %4, %9\n v_and_b32 %1, %1, %8\n s_and_b32 %5, %5, %9\n"
                             "v_add_u32 %2, %2, %8\n s_add_u32 %6, %6, %9\n v_xor_b32 %3, %3, %8\n s_xor_b32 %7, %7, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(a4), "s"(seed) : "scc");
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ s0 ^ s1 ^ s2 ^ s3;
}

template <int KIND>
static void run(const char *name, int waves_per_simd, int n_cu, double clock_ghz) {
    const int blocks = n_cu * 4 * waves_per_simd, iters = 400;
    uint32_t *out;
    hipMalloc(&out, (size_t)blocks * 64 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<blocks, 64>>>(out, 100, 1);
    hipEventRecord(e0);
    k<KIND><<<blocks, 64>>>(out, iters, 1);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)blocks * iters * REP, cyc = ms * 1e-3 * clock_ghz * 1e9;
    fflush(stdout);
    printf("%-28s waves/SIMD %d: %.3f ms  %.3f wave-instr per clock per SIMD  (%.3f per CU)\n", name, waves_per_simd, ms, instr / cyc / (n_cu * 4), instr / cyc / n_cu);
    hipFree(out);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int n_cu = p.multiProcessorCount;
    const double ghz = p.clockRate * 1e-6;
    printf("%s: %d CUs, %.2f GHz nominal\n", p.name, n_cu, ghz);
    fflush(stdout);
    for (int w : {1, 2, 5, 8}) {
        run<0>("VALU int (and/add/xor)", w, n_cu, ghz);
        run<1>("VALU v_pk_add_u16", w, n_cu, ghz);
        run<2>("SALU (add/and/xor)", w, n_cu, ghz);
        run<3>("v_readlane_b32", w, n_cu, ghz);
        run<4>("VALU+SALU interleaved", w, n_cu, ghz);
    }
    return 0;
}
