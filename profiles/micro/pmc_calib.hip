// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the ACCESS SHAPES this library's chain kernels use: every kernel
// below moves a KNOWN number of bytes over buffers far larger than the 256 MB Infinity Cache; run once under `rocprofv3 --pmc FETCH_SIZE` and once under
// `--pmc WRITE_SIZE` (profiles/scripts/r4_pmc_calib.sh) and divide. MI355X_MICROARCH.md gives one calibrated point (16 B/lane coalesced streams: the
// counter reports half) and calls every other width uncalibrated.
//   rd_wide16      16 B per lane, coalesced (the guide's calibrated shape)
//   rd_coal8 / 4   8 / 4 B per lane, coalesced (the merge join's key / record streams)
//   rd_lane_runs   every lane streams its OWN contiguous run of 16-byte records, four records (64 B) per step - chain_lane20's anchors
//   rd_gather_line one 16-byte load per lane from a random 64-byte line (the probe join's table lines; the index joins' bucket reads)
//   rd_gather4     one 4-byte load per lane at a random address
//   wr_wide16      16 B per lane coalesced stores        wr_coal8   8 B per lane coalesced
//   wr_scatter8    one 8-byte store per lane at a random address (the merge join's records by query position)
//   wr_scatter16   one 16-byte store per lane at a random 16-byte slot (anchors dealt to pairs)
// build: hipcc --offload-arch=gfx950 -O3 profiles/micro/pmc_calib.hip -o profiles/micro/pmc_calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

__global__ void rd_wide16(const uint4* __restrict__ a, size_t n16, uint32_t* sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = a[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void rd_coal8(const uint2* __restrict__ a, size_t n8, uint32_t* sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) { const uint2 v = a[i]; acc ^= v.x ^ v.y; }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void rd_coal4(const uint32_t* __restrict__ a, size_t n4, uint32_t* sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) acc ^= a[i];
    if (acc == 0x12345678u) *sink = acc;
}
// lane t owns records [t * run, (t + 1) * run): 4 records per step, like chain_lane_body's anchor stream
__global__ void rd_lane_runs(const uint4* __restrict__ a, size_t n_lanes, uint32_t run, uint32_t* sink) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_lanes) return;
    const uint4* p = a + t * run;
    uint32_t acc = 0;
    for (uint32_t i = 0; i + 4 <= run; i += 4) { const uint4 v0 = p[i], v1 = p[i + 1], v2 = p[i + 2], v3 = p[i + 3]; acc ^= v0.x ^ v1.y ^ v2.z ^ v3.w; }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void rd_gather_line(const uint4* __restrict__ a, size_t n_lines, size_t n_loads, uint32_t* sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_loads; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = a[(mix(i) % n_lines) * 4]; acc ^= v.x; }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void rd_gather4(const uint32_t* __restrict__ a, size_t n4, size_t n_loads, uint32_t* sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_loads; i += (size_t)gridDim.x * blockDim.x) acc ^= a[mix(i) % n4];
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void wr_wide16(uint4* __restrict__ a, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) a[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
__global__ void wr_coal8(uint2* __restrict__ a, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) a[i] = make_uint2((uint32_t)i, 1);
}
__global__ void wr_scatter8(uint2* __restrict__ a, size_t n8, size_t n_stores) {      // a permutation-like scatter: every slot written about once
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_stores; i += (size_t)gridDim.x * blockDim.x) a[mix(i) % n8] = make_uint2((uint32_t)i, 1);
}
__global__ void wr_scatter16(uint4* __restrict__ a, size_t n16, size_t n_stores) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_stores; i += (size_t)gridDim.x * blockDim.x) a[mix(i) % n16] = make_uint4((uint32_t)i, 1, 2, 3);
}

int main() {
    const size_t BYTES = 4ull << 30;      // 4 GiB: sixteen times the Infinity Cache
    void *buf; uint32_t* sink;
    CK(hipMalloc(&buf, BYTES)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 1, BYTES));
    const dim3 g(256 * 16), b(256);
    const size_t n16 = BYTES / 16, n8 = BYTES / 8, n4 = BYTES / 4, n_lines = BYTES / 64;
    const uint32_t run = 192;                                   // records per lane: a 20 kb chunk's anchors at c = 125
    const size_t n_lanes = n16 / run, n_gather = 1ull << 28;    // 268 M random loads / stores
    hipLaunchKernelGGL(rd_wide16, g, b, 0, 0, (const uint4*)buf, n16, sink);
    hipLaunchKernelGGL(rd_coal8, g, b, 0, 0, (const uint2*)buf, n8, sink);
    hipLaunchKernelGGL(rd_coal4, g, b, 0, 0, (const uint32_t*)buf, n4, sink);
    hipLaunchKernelGGL(rd_lane_runs, dim3((uint32_t)((n_lanes + 127) / 128)), dim3(128), 0, 0, (const uint4*)buf, n_lanes, run, sink);
    hipLaunchKernelGGL(rd_gather_line, g, b, 0, 0, (const uint4*)buf, n_lines, n_gather, sink);
    hipLaunchKernelGGL(rd_gather4, g, b, 0, 0, (const uint32_t*)buf, n4, n_gather, sink);
    hipLaunchKernelGGL(wr_wide16, g, b, 0, 0, (uint4*)buf, n16);
    hipLaunchKernelGGL(wr_coal8, g, b, 0, 0, (uint2*)buf, n8);
    hipLaunchKernelGGL(wr_scatter8, g, b, 0, 0, (uint2*)buf, n8, n_gather);
    hipLaunchKernelGGL(wr_scatter16, g, b, 0, 0, (uint4*)buf, n16, n_gather);
    CK(hipDeviceSynchronize());
    // the bytes each kernel asked for (what its counter is divided into)
    printf("rd_wide16 %zu\nrd_coal8 %zu\nrd_coal4 %zu\nrd_lane_runs %zu\nrd_gather_line %zu %zu\nrd_gather4 %zu %zu\nwr_wide16 %zu\nwr_coal8 %zu\nwr_scatter8 %zu %zu\nwr_scatter16 %zu %zu\n",
           BYTES, BYTES, BYTES, n_lanes * (size_t)(run / 4 * 4) * 16, n_gather * 16, n_gather * 64, n_gather * 4, n_gather * 64, BYTES, BYTES, n_gather * 8, n_gather * 64, n_gather * 16, n_gather * 64);
    return 0;
}
