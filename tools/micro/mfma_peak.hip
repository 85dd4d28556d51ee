// Microbenchmark: what the matrix pipes of an MI355X sustain for the two MFMA shapes the engine uses,
// as a function of waves per SIMD and of how many independent accumulator chains a wave keeps in flight.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o tools/micro/build/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int CHAINS>
__global__ __launch_bounds__(64) void k_f32(float* out, int iters, float a0, float b0) {
    f32x4 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = {0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x, b = b0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

// CHAINS accumulators, each fed RUN dependent MFMAs back to back before the next accumulator's turn,
// distinct A/B registers per k-step (what a register-blocked GEMM inner loop looks like)
template <int CHAINS, int RUN>
__global__ __launch_bounds__(64) void k_f32_runs(float* out, int iters, float a0, float b0) {
    f32x4 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = {0.f, 0.f, 0.f, 0.f};
    float a[RUN], b[RUN];
    for (int k = 0; k < RUN; ++k) { a[k] = a0 + threadIdx.x + k; b[k] = b0 + k; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c)
#pragma unroll
            for (int k = 0; k < RUN; ++k) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k], b[k], acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int CHAINS>
__global__ __launch_bounds__(64) void k_bf16(float* out, int iters, float a0) {
    f32x4 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = {0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(a0 + i); b[i] = (__bf16)(a0 - i); }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <class F>
static double time_ms(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* out;
    hipMalloc(&out, 1024 * 8 * 64 * sizeof(float));
    const int iters = 20000;
    printf("shape,chains,waves_per_simd,ms,TFLOPs\n");
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 1024 * wps;
        {
            double ms = time_ms([&] { hipLaunchKernelGGL(k_f32<1>, dim3(blocks), dim3(64), 0, 0, out, iters, 1.f, 2.f); });
            printf("f32_16x16x4,1,%d,%.3f,%.1f\n", wps, ms, 2048.0 * iters * 1 * blocks / ms / 1e9);
            ms = time_ms([&] { hipLaunchKernelGGL(k_f32<4>, dim3(blocks), dim3(64), 0, 0, out, iters, 1.f, 2.f); });
            printf("f32_16x16x4,4,%d,%.3f,%.1f\n", wps, ms, 2048.0 * iters * 4 * blocks / ms / 1e9);
            ms = time_ms([&] { hipLaunchKernelGGL((k_f32_runs<4, 4>), dim3(blocks), dim3(64), 0, 0, out, iters / 4, 1.f, 2.f); });
            printf("f32_16x16x4 4acc x run4,4,%d,%.3f,%.1f\n", wps, ms, 2048.0 * (iters / 4) * 16 * blocks / ms / 1e9);
            ms = time_ms([&] { hipLaunchKernelGGL((k_f32_runs<8, 4>), dim3(blocks), dim3(64), 0, 0, out, iters / 4, 1.f, 2.f); });
            printf("f32_16x16x4 8acc x run4,8,%d,%.3f,%.1f\n", wps, ms, 2048.0 * (iters / 4) * 32 * blocks / ms / 1e9);
            ms = time_ms([&] { hipLaunchKernelGGL((k_f32_runs<8, 1>), dim3(blocks), dim3(64), 0, 0, out, iters, 1.f, 2.f); });
            printf("f32_16x16x4 8acc x run1,8,%d,%.3f,%.1f\n", wps, ms, 2048.0 * iters * 8 * blocks / ms / 1e9);
            ms = time_ms([&] { hipLaunchKernelGGL(k_bf16<1>, dim3(blocks), dim3(64), 0, 0, out, iters, 1.f); });
            printf("bf16_16x16x32,1,%d,%.3f,%.1f\n", wps, ms, 16384.0 * iters * 1 * blocks / ms / 1e9);
            ms = time_ms([&] { hipLaunchKernelGGL(k_bf16<4>, dim3(blocks), dim3(64), 0, 0, out, iters, 1.f); });
            printf("bf16_16x16x32,4,%d,%.3f,%.1f\n", wps, ms, 16384.0 * iters * 4 * blocks / ms / 1e9);
        }
    }
    return 0;
}
