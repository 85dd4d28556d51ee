// Hardware / toolchain check: are packed-float32 vector instructions (v_pk_add_f32, v_pk_mul_f32) of one wave still correct
// while OTHER waves of the same SIMD issue bf16 MFMAs?
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/pk_xdl.hip -o tools/micro/build/pk_xdl
//
// Round 4: beside a bf16 network role in the five-values layout (tools/micro/gru_b20_device.h) ~0.7 % of the float32 MFCC frames
// of the fused launch came out slightly wrong; a capture build (tools/gpu_b20_capture.py) showed the inputs intact and the
// transform registers wrong from the FIRST radix-4 pass + twiddle multiplication on, in lanes 48..63 of the wave; the same
// library with -target-feature -packed-fp32-ops showed none.  This kernel isolates the pattern: role A runs butterflies and
// complex multiplications in packed form (no fused multiply-add: -ffp-contract=off) and checks them against the same
// operations in unpacked instructions; role B issues v_mfma_f32_16x16x32_bf16 in one of several forms.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float add_s(float a, float b) { float r; asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float sub_s(float a, float b) { float r; asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float mul_s(float a, float b) { float r; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

__device__ __forceinline__ void role_pk(const float* tw, int lane, int iters, unsigned seed, unsigned* bad_lanes, unsigned* bad_total) {
    unsigned bad = 0;
    f32x2 a = {1.0f + lane, 2.0f + lane};
    const f32x2 w = {tw[2 * lane], tw[2 * lane + 1]};                  // per-lane "twiddle" from LDS
    for (int i = 0; i < iters; ++i) {
        const float s = (float)((i * 2654435761u + seed) & 1023) * (1.0f / 64.0f);
        const f32x2 c = {s, s + 1.0f};
        // packed: butterfly, then two complex multiplications written as packed multiplies + packed add / subtract
        const f32x2 u = a + c, d = a - c;
        const f32x2 uw = u * w, us = f32x2{u[1], u[0]} * w;             // (u0 w0, u1 w1), (u1 w0, u0 w1)
        const f32x2 dw = d * c, ds = f32x2{d[1], d[0]} * c;
        const f32x2 x = {uw[0] - uw[1], us[0] + us[1]};
        const f32x2 y = {dw[0] - dw[1], ds[0] + ds[1]};
        const f32x2 z = x + y;
        // the same roundings in unpacked instructions
        const float u0 = add_s(a[0], c[0]), u1 = add_s(a[1], c[1]), d0 = sub_s(a[0], c[0]), d1 = sub_s(a[1], c[1]);
        const float x0 = sub_s(mul_s(u0, w[0]), mul_s(u1, w[1])), x1 = add_s(mul_s(u1, w[0]), mul_s(u0, w[1]));
        const float y0 = sub_s(mul_s(d0, c[0]), mul_s(d1, c[1])), y1 = add_s(mul_s(d1, c[0]), mul_s(d0, c[1]));
        const float z0 = add_s(x0, y0), z1 = add_s(x1, y1);
        bad += (__float_as_uint(z[0]) != __float_as_uint(z0)) + (__float_as_uint(z[1]) != __float_as_uint(z1));
        a = f32x2{z0 * 0.001f + 1.0f + lane, z1 * 0.001f + 2.0f + lane};
    }
    if (bad) { atomicAdd(bad_total, bad); atomicAdd(bad_lanes + lane, bad); }
}

// partner 1: D == C accumulate; 2: D != C (ping-pong between two register sets); 3: C = 0, then a chain on the result, operands
// changing (the shape of a projection + recurrence step)
__device__ __forceinline__ void role_mfma(float* out, int iters, int lane, int kind) {
    f32x4 acc[4], alt[4];
    for (int c = 0; c < 4; ++c) { acc[c] = {0.f, 0.f, 0.f, 0.f}; alt[c] = {0.f, 0.f, 0.f, 0.f}; }
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + lane + i); b[i] = (__bf16)0.5f; }
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < iters; ++i) {
        if (kind == 1) {
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[c], 0, 0, 0);
        } else if (kind == 2) {
#pragma unroll
            for (int c = 0; c < 4; ++c) alt[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[c], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, alt[c], 0, 0, 0);
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) alt[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, zero4, 0, 0, 0);
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, alt[c], 0, 0, 0);
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
            bf16x2 p; p[0] = (__bf16)acc[0][0]; p[1] = (__bf16)acc[1][1];
            a[i & 7] = p[0]; b[(i + 3) & 7] = p[1];
        }
    }
    float s = 0.f;
    for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3] + alt[c][0];
    if (s == 12345.678f) out[lane] = s;
}

__global__ __launch_bounds__(1024) void k_pk(unsigned* bad_lanes, unsigned* bad_total, float* out, int wa, int partner, int iters, int iters_b) {
    __shared__ float tw[128];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x < 128) tw[threadIdx.x] = 0.5f + 0.001f * threadIdx.x;
    __syncthreads();
    if (wave < wa) role_pk(tw, lane, iters, blockIdx.x * 977u + wave, bad_lanes, bad_total);
    else if (partner) role_mfma(out, iters_b, lane, partner);
}

int main() {
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    unsigned* bad; float* out;
    (void)hipMalloc(&bad, 65 * 4); (void)hipMalloc(&out, 4096);
    printf("pk_waves,mfma_kind,mfma_waves,iterations,values_checked,mismatches,mismatches_in_lanes_48_63\n");
    for (int partner = 0; partner < 4; ++partner)
        for (int wa : {4, 12}) {
            const int wb = partner ? 4 : 0, iters = 400000;
            (void)hipMemset(bad, 0, 65 * 4);
            const dim3 grid(cus * 2), block(64 * (wa + wb));
            hipLaunchKernelGGL(k_pk, grid, block, 0, 0, bad, bad + 64, out, wa, partner, iters, iters * 2);
            (void)hipDeviceSynchronize();
            unsigned h[65];
            (void)hipMemcpy(h, bad, sizeof h, hipMemcpyDeviceToHost);
            unsigned hi = 0;
            for (int l = 48; l < 64; ++l) hi += h[l];
            printf("%d,%d,%d,%d,%.3g,%u,%u\n", wa, partner, wb, iters, (double)grid.x * wa * 64 * 2.0 * iters, h[64], hi);
            fflush(stdout);
        }
    return 0;
}
