// latency.hip -- what one dependent memory round trip costs a wave of k_best8 (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 -o latency tools/micro/latency.hip ; run: ./latency
// One-wave blocks, W per CU, each chasing a random cycle through a buffer of F bytes:
//   kind 0: one 4-byte load per lane of a 256-byte row (the table rows of the walk: the whole wave reads one line pair)
//   kind 1: lanes 0..7 read 8 consecutive dwords, replicated x8 (the stream words of the walk)
// and, per hop, `extra` independent row loads issued in front of the dependent one (the abandoned rows of a pipeline that is
// being restarted: data returns in order, so the dependent load waits for them).
// Reported: shader-clock cycles (s_memtime) and ns (s_memrealtime, 100 MHz) per hop, averaged over the waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>

template <int EXTRA>
__global__ void __launch_bounds__(64) chase(const uint32_t *__restrict__ buf, uint32_t n_rows, int hops, int kind, int nt, unsigned long long *out, uint32_t *sink) {
    const uint32_t lane = threadIdx.x;
    uint32_t row = (blockIdx.x * 2654435761u) & (n_rows - 1u);
    uint32_t acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int h = 0; h < hops; h++) {
        uint32_t e[EXTRA ? EXTRA : 1] = {0};
#pragma unroll
        for (int x = 0; x < EXTRA; x++) {   // independent loads in front (addresses from the previous row, not from this hop's data); all in flight together
            const uint32_t r2 = (row * 40503u + (uint32_t)x * 9176u + 77u) & (n_rows - 1u);   // (n_rows is a power of two)
            e[x] = nt ? __builtin_nontemporal_load(buf + (uint64_t)r2 * 64 + lane) : buf[(uint64_t)r2 * 64 + lane];
        }
        uint32_t v;
        if (kind == 0) v = buf[(uint64_t)row * 64 + lane];
        else v = buf[(uint64_t)row * 64 + (lane & 7u)];
#pragma unroll
        for (int x = 0; x < EXTRA; x++) acc += e[x];
        row = (uint32_t)__builtin_amdgcn_readfirstlane((int)v) & (n_rows - 1u);   // dword 0 of a row names the next row
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = r1 - r0; }
    sink[blockIdx.x * 64 + lane] = acc + row;
}

int main(int argc, char **argv) {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int n_cu = p.multiProcessorCount;
    const int hops = 2000;
    printf("device: %s, %d CUs\n", p.name, n_cu);
    const size_t foot[] = {1u << 20, 64u << 20, 256u << 20, 2048ull << 20};
    const int waves[] = {1, 16};
    for (size_t F : foot) {
        const uint32_t n_rows = (uint32_t)(F / 256);
        std::vector<uint32_t> perm(n_rows), host((size_t)n_rows * 64, 0);
        std::iota(perm.begin(), perm.end(), 0u);
        std::mt19937 rng(7);
        std::shuffle(perm.begin(), perm.end(), rng);
        for (uint32_t i = 0; i < n_rows; i++) host[(size_t)perm[i] * 64] = perm[(i + 1) % n_rows];   // one cycle through all rows
        for (uint32_t i = 0; i < n_rows; i++) for (int k = 1; k < 64; k++) host[(size_t)i * 64 + k] = host[(size_t)i * 64];
        uint32_t *buf;
        hipMalloc(&buf, (size_t)n_rows * 256);
        hipMemcpy(buf, host.data(), (size_t)n_rows * 256, hipMemcpyHostToDevice);
        for (int W : waves) {
            for (int kind = 0; kind < 2; kind++) {
                for (int extra : {0, 1, 2, 4, 8, 16, 108, 116}) {
                    if (kind == 1 && extra) continue;
                    const int nt = extra >= 100;
                    if (nt) extra -= 100;
                    const int blocks = n_cu * W;
                    unsigned long long *out; uint32_t *sink;
                    hipMalloc(&out, (size_t)blocks * 16); hipMalloc(&sink, (size_t)blocks * 256);
                    for (int h : {200, hops}) {
                        switch (extra) {
                            case 0: chase<0><<<blocks, 64>>>(buf, n_rows, h, kind, nt, out, sink); break;
                            case 1: chase<1><<<blocks, 64>>>(buf, n_rows, h, kind, nt, out, sink); break;
                            case 2: chase<2><<<blocks, 64>>>(buf, n_rows, h, kind, nt, out, sink); break;
                            case 4: chase<4><<<blocks, 64>>>(buf, n_rows, h, kind, nt, out, sink); break;
                            case 8: chase<8><<<blocks, 64>>>(buf, n_rows, h, kind, nt, out, sink); break;
                            default: chase<16><<<blocks, 64>>>(buf, n_rows, h, kind, nt, out, sink); break;
                        }
                    }
                    hipDeviceSynchronize();
                    std::vector<unsigned long long> o((size_t)blocks * 2);
                    hipMemcpy(o.data(), out, o.size() * 8, hipMemcpyDeviceToHost);
                    double c = 0, r = 0;
                    for (int b = 0; b < blocks; b++) { c += (double)o[2 * b]; r += (double)o[2 * b + 1]; }
                    printf("footprint %5zu MB  waves/CU %2d  %-6s extra %d : %7.0f cycles  %7.0f ns per hop\n", F >> 20, W, kind ? "words" : (nt ? "row,nt" : "row"), extra,
                           c / blocks / hops, r / blocks / hops * 10.0);
                    hipFree(out); hipFree(sink);
                }
            }
        }
        hipFree(buf);
        fflush(stdout);
    }
    return 0;
}
