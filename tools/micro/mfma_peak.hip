// Microbenchmark: what the matrix pipes of an MI355X sustain for the two MFMA shapes the engine uses, as a
// function of waves per SIMD and of the dependency pattern between consecutive MFMAs of one wave.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o tools/micro/build/mfma_peak
// Patterns (ACC accumulators, RUN consecutive MFMAs into the same accumulator before moving on; a
// sched_barrier after every MFMA keeps the emitted order equal to the source order -- left alone, the
// scheduler interleaves independent accumulators, so check the ISA before trusting a label):
//   chain        ACC=1          every MFMA waits for the previous one
//   interleaved  ACC=4, RUN=1   the dependent predecessor is 4 instructions back
//   runs         ACC=4, RUN=4   4 dependent MFMAs back to back, then the next accumulator
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int ACC, int RUN>
__global__ __launch_bounds__(64) void k_f32(float* out, int iters, float a0, float b0) {
    f32x4 acc[ACC];
    for (int c = 0; c < ACC; ++c) acc[c] = {0.f, 0.f, 0.f, 0.f};
    float a[4], b[4];
    for (int k = 0; k < 4; ++k) { a[k] = a0 + threadIdx.x + k; b[k] = b0 + k; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < ACC; ++c)
#pragma unroll
            for (int k = 0; k < RUN; ++k) {
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k & 3], b[k & 3], acc[c], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    float s = 0.f;
    for (int c = 0; c < ACC; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int ACC, int RUN>
__global__ __launch_bounds__(64) void k_bf16(float* out, int iters, float a0) {
    f32x4 acc[ACC];
    for (int c = 0; c < ACC; ++c) acc[c] = {0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(a0 + i); b[i] = (__bf16)(a0 - i); }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < ACC; ++c)
#pragma unroll
            for (int k = 0; k < RUN; ++k) {
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[c], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    float s = 0.f;
    for (int c = 0; c < ACC; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <class F>
static double time_ms(F launch) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

#define RUN_F32(name, A, R, n)                                                                                   \
    { double ms = time_ms([&] { hipLaunchKernelGGL((k_f32<A, R>), dim3(blocks), dim3(64), 0, 0, out, n, 1.f, 2.f); }); \
      printf("f32_16x16x4,%s,%d,%.3f,%.1f\n", name, wps, ms, 2048.0 * (n) * (A) * (R) * blocks / ms / 1e9); }
#define RUN_BF16(name, A, R, n)                                                                                  \
    { double ms = time_ms([&] { hipLaunchKernelGGL((k_bf16<A, R>), dim3(blocks), dim3(64), 0, 0, out, n, 1.f); });    \
      printf("bf16_16x16x32,%s,%d,%.3f,%.1f\n", name, wps, ms, 16384.0 * (n) * (A) * (R) * blocks / ms / 1e9); }

int main() {
    float* out;
    (void)hipMalloc(&out, 1024 * 8 * 64 * sizeof(float));
    printf("shape,pattern,waves_per_simd,ms,TFLOPs\n");
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 1024 * wps;
        RUN_F32("chain", 1, 1, 16000)
        RUN_F32("interleaved4", 4, 1, 4000)
        RUN_F32("runs4x4", 4, 4, 1000)
        RUN_F32("runs4x2", 4, 2, 2000)
        RUN_BF16("chain", 1, 1, 16000)
        RUN_BF16("interleaved4", 4, 1, 4000)
        RUN_BF16("runs4x4", 4, 4, 1000)
    }
    return 0;
}
