// Micro-benchmark behind the binned join of Gb-scale pairs: what does a scatter of 8-byte records cost when a workgroup's targets are confined to a window of W records
// (the window's lines are completed in the XCD's L2 before they leave) against a scatter over the whole array (every store its own 32-byte write)?
// hipcc --offload-arch=gfx950 -O3 scatter_window.hip -o scatter_window && ./scatter_window
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// one thread per record, target anywhere: i -> (i * A) mod n (n a power of two, A odd: a permutation)
__global__ __launch_bounds__(256) void scatter_all(const uint2* __restrict__ in, uint2* __restrict__ out, uint32_t n, uint32_t A) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) out[(i * A) & (n - 1)] = in[i];
}
// a workgroup per window of W records (W a power of two): reads (low, record) pairs in sequence, writes inside its window; workgroups of window w run on XCD w % 8
__global__ __launch_bounds__(1024) void scatter_win(const uint32_t* __restrict__ low, const uint2* __restrict__ in, uint2* __restrict__ out, uint32_t W, uint32_t per_wg) {
    const uint32_t xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const uint32_t wgs_per_win = W / per_wg;                 // workgroups that share a window (all on one XCD)
    const uint32_t win = (slot / wgs_per_win) * 8 + xcd, part = slot % wgs_per_win;
    const size_t base = (size_t)win * W;
    for (uint32_t j = part * per_wg + threadIdx.x; j < (part + 1) * per_wg; j += 1024) out[base + low[base + j]] = in[base + j];
}
__global__ void fill_low(uint32_t* low, uint32_t n, uint32_t W, uint32_t A) { const uint32_t i = blockIdx.x * 256u + threadIdx.x; if (i < n) low[i] = ((i & (W - 1)) * A) & (W - 1); }

int main() {
    const uint32_t n = 1u << 27;      // 128 M records of 8 bytes = 1 GiB
    uint2 *in, *out; uint32_t* low;
    CK(hipMalloc(&in, (size_t)n * 8)); CK(hipMalloc(&out, (size_t)n * 8)); CK(hipMalloc(&low, (size_t)n * 4));
    CK(hipMemset(in, 1, (size_t)n * 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto time = [&](auto launch, const char* what) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(a); for (int r = 0; r < 5; r++) launch(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
        printf("%-34s %8.3f ms  %7.1f G records/s  (%5.2f TB/s of 16 B read+write per record)\n", what, ms, n / ms / 1e6, n * 16.0 / ms / 1e9);
    };
    time([&] { hipMemcpyAsync(out, in, (size_t)n * 8, hipMemcpyDeviceToDevice, 0); }, "copy");
    time([&] { hipLaunchKernelGGL(scatter_all, dim3(n / 256), dim3(256), 0, 0, in, out, n, 0x9E3779B1u); }, "scatter over the whole array");
    for (uint32_t lw = 13; lw <= 22; lw += 1) {
        const uint32_t W = 1u << lw;
        hipLaunchKernelGGL(fill_low, dim3(n / 256), dim3(256), 0, 0, low, n, W, 0x9E3779B1u);
        for (uint32_t per_wg : {W, W / 4 >= 8192 ? W / 4 : 0u, W / 16 >= 8192 ? W / 16 : 0u}) {
            if (!per_wg) continue;
            char what[96]; snprintf(what, sizeof what, "window 2^%u records, %u wg/window", lw, W / per_wg);
            time([&] { hipLaunchKernelGGL(scatter_win, dim3(n / per_wg), dim3(1024), 0, 0, low, in, out, W, per_wg); }, what);
        }
    }
    return 0;
}
