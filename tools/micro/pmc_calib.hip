// pmc_calib.hip -- known byte counts in the access patterns of the placement kernels, to calibrate rocprofv3's
// FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md: only the 16 B/lane streaming read is calibrated there).
//   hipcc --offload-arch=gfx950 -O3 -o pmc_calib tools/micro/pmc_calib.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o p --output-format csv -- ./pmc_calib   (and WRITE_SIZE)
// Every kernel moves exactly BYTES bytes (1 GiB, larger than L2 + Infinity Cache) once.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
static const size_t BYTES = 1ull << 30;
__global__ void read_row4(const uint32_t *p, uint32_t *sink, size_t n_rows) {    // k_best8's row fetch: a wave reads one 256-B row, 4 B per lane
    const size_t wave = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64, lane = threadIdx.x & 63;
    uint32_t acc = 0;
    for (size_t r = wave; r < n_rows; r += (size_t)gridDim.x * (blockDim.x / 64)) acc ^= p[r * 64 + lane];
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void read_vec16(const uint4 *p, uint32_t *sink, size_t n) {           // 16 B per lane streaming read (the guide's calibrated case)
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void write_row4(uint32_t *p, size_t n) {                              // k_fill_table: 4 B per lane streaming store
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}
__global__ void write_vec16(uint4 *p, size_t n) {                                // chunk records of k_best8: 16 B per lane
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
__global__ void atomic_xor4(uint32_t *p, size_t n) {                             // k_scatter_entries: one 4-B atomic per thread, a different 256-B row each
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) atomicXor(&p[(i * 64) % (BYTES / 4)], 1u);
}
int main() {
    void *buf; uint32_t *sink;
    hipMalloc(&buf, BYTES); hipMalloc(&sink, 4);
    hipMemset(buf, 1, BYTES);
    hipDeviceSynchronize();
    read_row4<<<4096, 256>>>((const uint32_t *)buf, sink, BYTES / 256);
    read_vec16<<<4096, 256>>>((const uint4 *)buf, sink, BYTES / 16);
    write_row4<<<4096, 256>>>((uint32_t *)buf, BYTES / 4);
    write_vec16<<<4096, 256>>>((uint4 *)buf, BYTES / 16);
    atomic_xor4<<<4096, 256>>>((uint32_t *)buf, BYTES / 256);    // 4 Mi atomics, each on its own 256-B row
    hipDeviceSynchronize();
    printf("each kernel moved %zu bytes (atomic_xor4: %zu atomics of 4 B)\n", BYTES, BYTES / 256);
    return 0;
}
