// Microbenchmark: which execution pipes of one gfx950 SIMD overlap when DIFFERENT waves use them.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/pipe_overlap.hip -o tools/micro/build/pipe_overlap
//
// The fused update launch puts an MFMA-bound role (GRU) and a VALU / LDS-bound role (MFCC frames, float64 or float32)
// on the same SIMDs.  Whether that can cost max(roles) or must cost sum(roles) is a property of the hardware:
// this kernel gives every SIMD of every compute unit WA waves of role A and WB waves of role B (a workgroup of
// 4 * (WA + WB) waves, wave w on SIMD w % 4: checked through HW_ID), each wave spinning on ONE kind of instruction
// with enough independent accumulators to be issue-bound, and times A alone, B alone and A + B together.
//   overlap = (t_A + t_B - t_AB) / min(t_A, t_B)      1.0: fully concurrent, 0.0: serialised
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum Role { kIdle = 0, kMfma = 1, kF64 = 2, kF32 = 3, kLds = 4, kMfma4 = 5, kPk32 = 6, kMfmaChain = 7, kMfmaChain2 = 8, kBf16 = 9, kBf16Chain = 10, kLdsOnly = 11, kCvt = 12, kBf16Ops = 13 };
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void role_mfma(float* out, int iters, int lane) {
    f32x4 acc[4];
    for (int c = 0; c < 4; ++c) acc[c] = {0.f, 0.f, 0.f, 0.f};
    const float a = 1.0f + lane, b = 0.5f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    if (s == 12345.678f) out[lane] = s;
}
// ONE dependent chain (what a GRU wave's timestep looks like): the wave is not ready between two MFMAs
template <int ACC>
__device__ __forceinline__ void role_mfma_chain(float* out, int iters, int lane) {
    f32x4 acc[ACC];
    for (int c = 0; c < ACC; ++c) acc[c] = {0.f, 0.f, 0.f, 0.f};
    const float a = 1.0f + lane, b = 0.5f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < ACC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < ACC; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    if (s == 12345.678f) out[lane] = s;
}
// (round 4, second session) the XDL matrix pipe: v_mfma_f32_16x16x32_bf16 (four passes).  The f32-input MFMAs above run at
// the f32 VECTOR rate (MI355X_MICROARCH.md); whether the bf16 ones share the VALU as well is the question
template <int ACC>
__device__ __forceinline__ void role_bf16(float* out, int iters, int lane) {
    f32x4 acc[ACC];
    for (int c = 0; c < ACC; ++c) acc[c] = {0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + lane + i); b[i] = (__bf16)0.5f; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < ACC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < ACC; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    if (s == 12345.678f) out[lane] = s;
}
// the same with EIGHT different A / B operand register sets (a real kernel's MFMAs read different registers each time:
// 12 VGPRs per 16 cycles -- does the operand traffic of the XDL pipe crowd out the vector pipe's?)
__device__ __forceinline__ void role_bf16_ops(float* out, int iters, int lane) {
    f32x4 acc[4];
    for (int c = 0; c < 4; ++c) acc[c] = {0.f, 0.f, 0.f, 0.f};
    bf16x8 a[8], b[8];
    for (int o = 0; o < 8; ++o)
        for (int i = 0; i < 8; ++i) { a[o][i] = (__bf16)(1.0f + lane + i + o); b[o][i] = (__bf16)(0.5f + o); }
    for (int o = 0; o < 8; ++o) asm volatile("" : "+v"(a[o]), "+v"(b[o]));
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(4 * k + c) & 7], b[(4 * k + c + 3) & 7], acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    if (s == 12345.678f) out[lane] = s;
}
__device__ __forceinline__ void role_mfma4(float* out, int iters, int lane) {     // two-pass 4x4x1
    f32x4 acc[4];
    for (int c = 0; c < 4; ++c) acc[c] = {0.f, 0.f, 0.f, 0.f};
    const float a = 1.0f + lane, b = 0.5f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    if (s == 12345.678f) out[lane] = s;
}
__device__ __forceinline__ void role_f64(float* out, int iters, int lane) {
    double acc[8];
    for (int c = 0; c < 8; ++c) acc[c] = 1.0 + c + lane;
    const double m = 1.0000001, d = 1e-9;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = __builtin_fma(acc[c], m, d);
    }
    double s = 0;
    for (int c = 0; c < 8; ++c) s += acc[c];
    if (s == 12345.678) out[lane] = (float)s;
}
__device__ __forceinline__ void role_f32(float* out, int iters, int lane) {
    float acc[8];
    for (int c = 0; c < 8; ++c) acc[c] = 1.0f + c + lane;
    const float m = 1.0000001f, d = 1e-9f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < 8; ++c) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[c]) : "v"(m), "v"(d));      // (left to the compiler the loop becomes v_pk_fma_f32)
    }
    float s = 0;
    for (int c = 0; c < 8; ++c) s += acc[c];
    if (s == 12345.678f) out[lane] = s;
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void role_pk32(float* out, int iters, int lane) {       // v_pk_fma_f32
    f32x2 acc[8];
    for (int c = 0; c < 8; ++c) acc[c] = {1.0f + c + lane, 2.0f + c};
    const f32x2 m = {1.0000001f, 0.9999999f}, d = {1e-9f, 2e-9f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = __builtin_elementwise_fma(acc[c], m, d);
    }
    float s = 0;
    for (int c = 0; c < 8; ++c) s += acc[c][0] + acc[c][1];
    if (s == 12345.678f) out[lane] = s;
}
__device__ __forceinline__ void role_lds(float* out, int iters, int lane, double* lds) {
    double s = 0;
    typedef const volatile __attribute__((address_space(3))) double* lds_vdouble;      // (a volatile GENERIC pointer becomes flat loads)
    const lds_vdouble p = (lds_vdouble)(lds + lane);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < 8; ++c) s += p[c * 64];
    }
    if (s == 12345.678) out[lane] = (float)s;
}

__device__ __forceinline__ void role_lds_only(float* out, int iters, int lane, double* lds) {     // LDS reads, ONE add per 8 reads
    typedef const volatile __attribute__((address_space(3))) double* lds_vdouble;
    const lds_vdouble p = (lds_vdouble)(lds + lane);
    double s = 0;
    for (int i = 0; i < iters; ++i) {
        double v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = p[c * 64];
        double u = v[0];
#pragma unroll
        for (int c = 1; c < 8; ++c) asm volatile("" : "+v"(u) : "v"(v[c]));      // the loaded values are "used" without a VALU instruction
        s += u;
    }
    if (s == 12345.678) out[lane] = (float)s;
}
__device__ __forceinline__ void role_cvt(float* out, int iters, int lane) {       // the split arithmetic: f32 -> bf16 pieces (v_cvt_pk_bf16_f32 + subtract)
    float acc[8];
    for (int c = 0; c < 8; ++c) acc[c] = 1.0f + c + lane;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            unsigned u;
            asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(u) : "v"(acc[c]));
            asm volatile("v_sub_f32 %0, %0, %1" : "+v"(acc[c]) : "v"(u));
        }
    }
    float s = 0;
    for (int c = 0; c < 8; ++c) s += acc[c];
    if (s == 12345.678f) out[lane] = s;
}
__device__ __forceinline__ void run_role(int role, float* out, int iters, int lane, double* lds) {
    switch (role) {
        case kMfma: role_mfma(out, iters, lane); break;
        case kMfma4: role_mfma4(out, iters, lane); break;
        case kF64: role_f64(out, iters, lane); break;
        case kF32: role_f32(out, iters, lane); break;
        case kPk32: role_pk32(out, iters, lane); break;
        case kLds: role_lds(out, iters, lane, lds); break;
        case kMfmaChain: role_mfma_chain<1>(out, iters, lane); break;
        case kMfmaChain2: role_mfma_chain<2>(out, iters, lane); break;
        case kBf16: role_bf16<4>(out, iters, lane); break;
        case kBf16Chain: role_bf16<1>(out, iters, lane); break;
        case kLdsOnly: role_lds_only(out, iters, lane, lds); break;
        case kCvt: role_cvt(out, iters, lane); break;
        case kBf16Ops: role_bf16_ops(out, iters, lane); break;
        default: break;
    }
}

// waves [0, 4 WA) run role A, waves [4 WA, 4 (WA + WB)) role B; wave w sits on SIMD w % 4 (recorded in simd_out)
__device__ __forceinline__ void set_prio(int p) {
    if (p == 1) __builtin_amdgcn_s_setprio(1);
    else if (p == 2) __builtin_amdgcn_s_setprio(2);
    else if (p == 3) __builtin_amdgcn_s_setprio(3);
}
__global__ __launch_bounds__(1024) void k_overlap(float* out, int* simd_out, int wa, int role_a, int iters_a, int role_b, int iters_b, int prio_a, int prio_b) {
    __shared__ double lds[8 * 64 + 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x < 8 * 64) lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (blockIdx.x == 0 && lane == 0) simd_out[wave] = (int)(__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4) & 3);
    if (wave < 4 * wa) { set_prio(prio_a); run_role(role_a, out, iters_a, lane, lds); }
    else { set_prio(prio_b); run_role(role_b, out, iters_b, lane, lds); }
}

static const char* kNames[] = {"idle", "mfma16x16x4", "fma_f64", "fma_f32", "lds_read_b64", "mfma4x4x1", "pk_fma_f32", "mfma_1chain", "mfma_2chains", "mfma16x16x32_bf16", "bf16_1chain", "lds_read_only", "and_sub_f32", "bf16_8_operand_sets"};

int main(int argc, char** argv) {
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    float* out; int* simd;
    (void)hipMalloc(&out, 4096); (void)hipMalloc(&simd, 64 * sizeof(int));
    auto time_ms = [&](int wa, int wb, int ra, int ia, int rb, int ib, int pa = 0, int pb = 0) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        const int threads = 64 * 4 * (wa + wb);
        hipLaunchKernelGGL(k_overlap, dim3(cus), dim3(threads), 0, 0, out, simd, wa, ra, ia, rb, ib, pa, pb);
        (void)hipDeviceSynchronize();
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k_overlap, dim3(cus), dim3(threads), 0, 0, out, simd, wa, ra, ia, rb, ib, pa, pb);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        return (double)best;
    };
    // iteration counts giving each role ~1 ms alone at one wave per SIMD
    const int it_mfma = 20000, it_f64 = 80000, it_f32 = 80000, it_lds = 40000, it_mfma4 = 80000, it_pk = 80000;
    int its[14] = {0, it_mfma, it_f64, it_f32, it_lds, it_mfma4, it_pk, it_mfma, it_mfma / 2, 10000, 40000, it_lds, 40000, 10000};
    printf("role_a,waves_a,prio_a,role_b,waves_b,prio_b,t_a_ms,t_b_ms,t_ab_ms,overlap\n");
    struct Case { int ra, wa, rb, wb, pa, pb; };
    const bool xdl_only = argc > 1 && argv[1][0] == 'x';
    const Case cases_xdl[] = {
        {kBf16, 1, kF64, 1}, {kBf16, 1, kF32, 1}, {kBf16, 1, kPk32, 1}, {kBf16, 1, kLds, 1}, {kBf16, 1, kLdsOnly, 1}, {kBf16, 1, kCvt, 1}, {kBf16, 1, kBf16, 1}, {kBf16, 1, kMfma, 1},
        {kBf16, 2, kF64, 2}, {kBf16, 2, kF32, 2}, {kBf16, 1, kF64, 3}, {kBf16, 1, kF32, 3}, {kBf16Chain, 1, kF64, 1}, {kBf16Chain, 1, kF32, 1}, {kBf16Chain, 2, kF64, 2},
        {kMfma, 1, kLdsOnly, 1}, {kMfma, 1, kF32, 1}, {kMfma, 1, kF64, 1},
        {kBf16Ops, 1, kF32, 1}, {kBf16Ops, 1, kF64, 1}, {kBf16Ops, 1, kCvt, 1}, {kBf16Ops, 2, kF32, 2}, {kBf16Ops, 1, kF32, 3}, {kBf16Ops, 1, kLdsOnly, 1},
    };
    const Case cases_all[] = {
        {kMfma, 1, kF64, 1}, {kMfma, 1, kF32, 1}, {kMfma, 1, kPk32, 1}, {kMfma, 1, kLds, 1}, {kMfma, 1, kMfma, 1}, {kF64, 1, kF64, 1}, {kF64, 1, kLds, 1},
        {kMfma, 2, kF64, 2}, {kMfma, 2, kF32, 2}, {kMfma, 1, kF64, 3}, {kMfma, 2, kLds, 2}, {kMfma4, 1, kF64, 1}, {kMfma4, 1, kMfma, 1},
        {kF32, 1, kF64, 1}, {kF32, 1, kLds, 1},
        // arbitration: who is preferred when both waves have an instruction ready (s_setprio)
        {kMfma, 1, kF64, 1, 0, 3}, {kMfma, 1, kF64, 1, 3, 0}, {kMfma, 1, kF32, 1, 0, 3}, {kMfma, 2, kF64, 2, 0, 3}, {kMfma, 1, kLds, 1, 0, 3},
        // an MFMA wave that is NOT always ready: one or two dependent chains (a GRU wave), equal and unequal priorities
        {kMfmaChain, 1, kF64, 1, 0, 0}, {kMfmaChain, 1, kF64, 1, 0, 3}, {kMfmaChain, 1, kF64, 1, 3, 0},
        {kMfmaChain2, 1, kF64, 1, 0, 0}, {kMfmaChain2, 1, kF64, 1, 0, 3}, {kMfmaChain2, 2, kF64, 2, 0, 3}, {kMfmaChain2, 2, kF64, 2, 0, 0},
        {kMfmaChain2, 2, kF32, 2, 0, 3}, {kMfmaChain2, 2, kLds, 2, 0, 3},
    };
    const Case* cases = xdl_only ? cases_xdl : cases_all;
    const int n_cases = xdl_only ? (int)(sizeof cases_xdl / sizeof(Case)) : (int)(sizeof cases_all / sizeof(Case));
    for (int ci = 0; ci < n_cases; ++ci) {
        const Case& c = cases[ci];
        const double ta = time_ms(c.wa, c.wb, c.ra, its[c.ra], kIdle, 0, c.pa, c.pb);
        const double tb = time_ms(c.wa, c.wb, kIdle, 0, c.rb, its[c.rb], c.pa, c.pb);
        const double tab = time_ms(c.wa, c.wb, c.ra, its[c.ra], c.rb, its[c.rb], c.pa, c.pb);
        const double mn = ta < tb ? ta : tb;
        printf("%s,%d,%d,%s,%d,%d,%.4f,%.4f,%.4f,%.3f\n", kNames[c.ra], c.wa, c.pa, kNames[c.rb], c.wb, c.pb, ta, tb, tab, (ta + tb - tab) / mn);
        fflush(stdout);
    }
    int h[64];
    (void)hipMemcpy(h, simd, sizeof h, hipMemcpyDeviceToHost);
    printf("simd of waves 0..15 of workgroup 0:");
    for (int i = 0; i < 16; ++i) printf(" %d", h[i]);
    printf("\n");
    return 0;
}
