// Calibration of the FETCH_SIZE / WRITE_SIZE counters for the access shapes the
// search kernels use: a known number of bytes is read with 4-byte-per-lane and
// with 16-byte-per-lane coalesced loads, and as a 4-byte gather at a 128-byte
// stride (one word per cache line, the hub-matrix shape).  Run under
// rocprofv3 --pmc FETCH_SIZE (and WRITE_SIZE) and compare with the bytes printed.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__global__ void read4(const uint32_t *p, size_t n, uint32_t *out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void read16(const uint4 *p, size_t n, uint32_t *out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = p[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void gather128(const uint32_t *p, size_t nlines, uint32_t *out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nlines; i += (size_t)gridDim.x * blockDim.x)
        acc += p[i * 32];
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void write4(uint32_t *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}

int main() {
    const size_t bytes = (size_t)2 << 30;  // 2 GiB, far beyond the 256 MiB last-level cache
    uint32_t *buf, *out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    read4<<<4096, 256>>>(buf, bytes / 4, out);
    read16<<<4096, 256>>>((const uint4 *)buf, bytes / 16, out);
    gather128<<<4096, 256>>>(buf, bytes / 128, out);
    write4<<<4096, 256>>>(buf, bytes / 4);
    hipDeviceSynchronize();
    printf("read4 bytes %zu\nread16 bytes %zu\ngather128 words %zu (x4 B useful, x128 B lines)\nwrite4 bytes %zu\n", bytes, bytes,
           bytes / 128, bytes);
    return 0;
}
