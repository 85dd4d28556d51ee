// Microbenchmark: HBM bandwidth an MI355X sustains for plain streaming kernels (read-only sum, copy), as the
// measured counterpart of the 8 TB/s spec figure used in bench.py's HBM roofline.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/hbm_copy.hip -o tools/micro/build/hbm_copy
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void k_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void k_read(const uint4* __restrict__ src, unsigned* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint4 v = src[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;            // keeps the loads alive, (almost) never true
}

template <class F>
static double time_ms(F launch, int reps) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main() {
    printf("kernel,bytes_per_buffer,grid,ms,GB_per_s\n");
    for (size_t mb : {512, 2048, 8192}) {             // well past the 256 MB Infinity Cache from the second size on
        const size_t bytes = mb << 20, n = bytes / sizeof(uint4);
        uint4 *src, *dst;
        unsigned* out;
        if (hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&dst, bytes) != hipSuccess) { printf("alloc failed at %zu MB\n", mb); break; }
        (void)hipMalloc(&out, 4);
        (void)hipMemset(src, 1, bytes); (void)hipMemset(dst, 0, bytes);
        for (int grid : {2048, 8192, 32768}) {
            double ms = time_ms([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, src, out, n); }, 5);
            printf("read,%zu,%d,%.3f,%.0f\n", bytes, grid, ms, bytes / ms / 1e6);
            ms = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, src, dst, n); }, 5);
            printf("copy,%zu,%d,%.3f,%.0f\n", bytes, grid, ms, 2.0 * bytes / ms / 1e6);
        }
        (void)hipFree(src); (void)hipFree(dst); (void)hipFree(out);
    }
    return 0;
}
