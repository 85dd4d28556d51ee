// Stand-alone reproducer attempt for the round-4 "float32 frame beside an XDL role" miscompute
// (profiles/round4/r4v_b20_fused_corruption.log): the exact instruction sequence hipcc emits for the float32 frame role
// between the first and the second radix-4 pass of mfcc_wave_frame<float, ...> (ISA of fused_update_bf16_kernel<float, ...>,
// ROCm 7.2, gfx950) -- first-pass twiddle multiplications fed by ds_read_b64, the eight v_permlane32/16_swap of exchange_b
// with the next pass' twiddle reads in between, SIX v_pk_add_f32 straight behind the last swap, then plain v_add/v_sub_f32
// that OVERWRITE registers the packed adds have just read -- as ONE inline-asm block on fixed registers, against the same
// arithmetic with every instruction separated by s_nop 7 and the packed adds written as scalar adds.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/pk_swap_hazard.hip -o tools/micro/build/pk_swap_hazard
//
// Role A waves run the sequence (VARIANT picks which guard is inserted), role B waves issue dense v_mfma_f32_16x16x32_bf16
// chains (the five-values network issues 9 per timestep back to back) on the same SIMDs.
// Candidates this separates:
//   RAW  v_permlane16_swap writes v11 / v13  ->  v_pk_add_f32 reads v[10:11] in the very next issue slot;
//   WAR  v_pk_add_f32 reads v[22:23]         ->  v_add_f32 / v_sub_f32 write v22 / v23 in the next two slots
//        (packed float32 shares the XDL datapath, tools/micro/pipe_overlap.hip: if its operand fetch is delayed by another
//        wave's MFMA, a following plain VALU write could overtake it).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Io { float r[8]; };

// inputs: v4, v5 = first butterfly output (no twiddle); v20 v21 / v18 v19 / v22 v23 = the three outputs to be twiddled;
// v14 = LDS address of this lane's first-pass twiddles (3 x 8 bytes, 512 apart); v42 = address of the second-pass ones.
#define LOAD_INPUTS \
    "v_mov_b32 v4, %8\n v_mov_b32 v5, %9\n v_mov_b32 v20, %10\n v_mov_b32 v21, %11\n" \
    "v_mov_b32 v18, %12\n v_mov_b32 v19, %13\n v_mov_b32 v22, %14\n v_mov_b32 v23, %15\n" \
    "v_mov_b32 v14, %16\n v_mov_b32 v42, %17\n s_nop 7\n"
#define STORE_OUTPUTS \
    "s_nop 7\n v_mov_b32 %0, v16\n v_mov_b32 %1, v17\n v_mov_b32 %2, v4\n v_mov_b32 %3, v5\n" \
    "v_mov_b32 %4, v12\n v_mov_b32 %5, v13\n v_mov_b32 %6, v10\n v_mov_b32 %7, v11\n"
#define CLOBBERS "v4", "v5", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v42", "memory"
#define OUTS "=&v"(o.r[0]), "=&v"(o.r[1]), "=&v"(o.r[2]), "=&v"(o.r[3]), "=&v"(o.r[4]), "=&v"(o.r[5]), "=&v"(o.r[6]), "=&v"(o.r[7])
#define INS "v"(in.r[0]), "v"(in.r[1]), "v"(in.r[2]), "v"(in.r[3]), "v"(in.r[4]), "v"(in.r[5]), "v"(in.r[6]), "v"(in.r[7]), "v"(a1), "v"(a2)

// G0: guard between the last swap and the first packed add; G1: guard between the last packed add and the plain adds
#define FAST_SEQ(G0, G1) \
    "ds_read_b64 v[10:11], v14\n" \
    "ds_read_b64 v[12:13], v14 offset:512\n" \
    "ds_read_b64 v[14:15], v14 offset:1024\n" \
    "s_waitcnt lgkmcnt(2)\n" \
    "v_mul_f32_e32 v16, v21, v11\n v_mul_f32_e32 v17, v20, v11\n v_fma_f32 v16, v20, v10, -v16\n v_fmac_f32_e32 v17, v21, v10\n" \
    "s_waitcnt lgkmcnt(1)\n" \
    "v_mul_f32_e32 v10, v19, v13\n v_mul_f32_e32 v11, v18, v13\n v_fma_f32 v10, v18, v12, -v10\n v_fmac_f32_e32 v11, v19, v12\n" \
    "s_waitcnt lgkmcnt(0)\n" \
    "v_mul_f32_e32 v12, v23, v15\n v_mul_f32_e32 v13, v22, v15\n v_fma_f32 v12, v22, v14, -v12\n v_fmac_f32_e32 v13, v23, v14\n" \
    "v_permlane32_swap_b32_e32 v4, v10\n v_permlane32_swap_b32_e32 v5, v11\n v_permlane32_swap_b32_e32 v16, v12\n v_permlane32_swap_b32_e32 v17, v13\n" \
    "ds_read_b64 v[14:15], v42\n ds_read_b64 v[18:19], v42 offset:128\n ds_read_b64 v[20:21], v42 offset:256\n" \
    "v_permlane16_swap_b32_e32 v4, v16\n v_permlane16_swap_b32_e32 v5, v17\n v_permlane16_swap_b32_e32 v10, v12\n v_permlane16_swap_b32_e32 v11, v13\n" \
    G0 \
    "v_pk_add_f32 v[22:23], v[4:5], v[10:11]\n" \
    "v_pk_add_f32 v[4:5], v[4:5], v[10:11] neg_lo:[0,1] neg_hi:[0,1]\n" \
    "v_pk_add_f32 v[10:11], v[16:17], v[12:13]\n" \
    "v_pk_add_f32 v[12:13], v[16:17], v[12:13] neg_lo:[0,1] neg_hi:[0,1]\n" \
    "v_pk_add_f32 v[16:17], v[22:23], v[10:11]\n" \
    "v_pk_add_f32 v[10:11], v[22:23], v[10:11] neg_lo:[0,1] neg_hi:[0,1]\n" \
    G1 \
    "v_add_f32_e32 v22, v4, v13\n v_sub_f32_e32 v23, v5, v12\n v_sub_f32_e32 v24, v4, v13\n v_add_f32_e32 v25, v12, v5\n" \
    "s_waitcnt lgkmcnt(2)\n" \
    "v_mul_f32_e32 v4, v23, v15\n v_mul_f32_e32 v5, v22, v15\n" \
    "s_waitcnt lgkmcnt(1)\n" \
    "v_mul_f32_e32 v12, v11, v19\n v_mul_f32_e32 v13, v10, v19\n" \
    "v_fma_f32 v4, v22, v14, -v4\n v_fmac_f32_e32 v5, v23, v14\n v_fma_f32 v12, v10, v18, -v12\n v_fmac_f32_e32 v13, v11, v18\n" \
    "s_waitcnt lgkmcnt(0)\n" \
    "v_mul_f32_e32 v10, v25, v21\n v_mul_f32_e32 v11, v24, v21\n v_fma_f32 v10, v24, v20, -v10\n v_fmac_f32_e32 v11, v25, v20\n"

#define N7 "s_nop 7\n"
// the same arithmetic, nothing packed, idle slots everywhere
#define SLOW_SEQ \
    "ds_read_b64 v[10:11], v14\n ds_read_b64 v[12:13], v14 offset:512\n ds_read_b64 v[14:15], v14 offset:1024\n s_waitcnt lgkmcnt(0)\n" N7 \
    "v_mul_f32_e32 v16, v21, v11\n" N7 "v_mul_f32_e32 v17, v20, v11\n" N7 "v_fma_f32 v16, v20, v10, -v16\n" N7 "v_fmac_f32_e32 v17, v21, v10\n" N7 \
    "v_mul_f32_e32 v10, v19, v13\n" N7 "v_mul_f32_e32 v11, v18, v13\n" N7 "v_fma_f32 v10, v18, v12, -v10\n" N7 "v_fmac_f32_e32 v11, v19, v12\n" N7 \
    "v_mul_f32_e32 v12, v23, v15\n" N7 "v_mul_f32_e32 v13, v22, v15\n" N7 "v_fma_f32 v12, v22, v14, -v12\n" N7 "v_fmac_f32_e32 v13, v23, v14\n" N7 \
    "v_permlane32_swap_b32_e32 v4, v10\n" N7 "v_permlane32_swap_b32_e32 v5, v11\n" N7 "v_permlane32_swap_b32_e32 v16, v12\n" N7 "v_permlane32_swap_b32_e32 v17, v13\n" N7 \
    "ds_read_b64 v[14:15], v42\n ds_read_b64 v[18:19], v42 offset:128\n ds_read_b64 v[20:21], v42 offset:256\n s_waitcnt lgkmcnt(0)\n" N7 \
    "v_permlane16_swap_b32_e32 v4, v16\n" N7 "v_permlane16_swap_b32_e32 v5, v17\n" N7 "v_permlane16_swap_b32_e32 v10, v12\n" N7 "v_permlane16_swap_b32_e32 v11, v13\n" N7 \
    "v_add_f32_e32 v22, v4, v10\n" N7 "v_add_f32_e32 v23, v5, v11\n" N7 "v_sub_f32_e32 v4, v4, v10\n" N7 "v_sub_f32_e32 v5, v5, v11\n" N7 \
    "v_add_f32_e32 v10, v16, v12\n" N7 "v_add_f32_e32 v11, v17, v13\n" N7 "v_sub_f32_e32 v12, v16, v12\n" N7 "v_sub_f32_e32 v13, v17, v13\n" N7 \
    "v_add_f32_e32 v16, v22, v10\n" N7 "v_add_f32_e32 v17, v23, v11\n" N7 "v_sub_f32_e32 v10, v22, v10\n" N7 "v_sub_f32_e32 v11, v23, v11\n" N7 \
    "v_add_f32_e32 v22, v4, v13\n" N7 "v_sub_f32_e32 v23, v5, v12\n" N7 "v_sub_f32_e32 v24, v4, v13\n" N7 "v_add_f32_e32 v25, v12, v5\n" N7 \
    "v_mul_f32_e32 v4, v23, v15\n" N7 "v_mul_f32_e32 v5, v22, v15\n" N7 "v_mul_f32_e32 v12, v11, v19\n" N7 "v_mul_f32_e32 v13, v10, v19\n" N7 \
    "v_fma_f32 v4, v22, v14, -v4\n" N7 "v_fmac_f32_e32 v5, v23, v14\n" N7 "v_fma_f32 v12, v10, v18, -v12\n" N7 "v_fmac_f32_e32 v13, v11, v18\n" N7 \
    "v_mul_f32_e32 v10, v25, v21\n" N7 "v_mul_f32_e32 v11, v24, v21\n" N7 "v_fma_f32 v10, v24, v20, -v10\n" N7 "v_fmac_f32_e32 v11, v25, v20\n" N7

template <int VARIANT>
__device__ __forceinline__ Io run_fast(const Io& in, unsigned a1, unsigned a2) {
    Io o;
    if constexpr (VARIANT == 0) asm volatile(LOAD_INPUTS FAST_SEQ("", "") STORE_OUTPUTS : OUTS : INS : CLOBBERS);                    // as emitted
    else if constexpr (VARIANT == 1) asm volatile(LOAD_INPUTS FAST_SEQ("s_nop 1\n", "") STORE_OUTPUTS : OUTS : INS : CLOBBERS);     // swap -> packed guarded
    else if constexpr (VARIANT == 2) asm volatile(LOAD_INPUTS FAST_SEQ("", "s_nop 1\n") STORE_OUTPUTS : OUTS : INS : CLOBBERS);     // packed -> plain write guarded
    else asm volatile(LOAD_INPUTS FAST_SEQ("s_nop 7\n", "s_nop 7\n") STORE_OUTPUTS : OUTS : INS : CLOBBERS);                         // both, generously
    return o;
}
__device__ __forceinline__ Io run_slow(const Io& in, unsigned a1, unsigned a2) {
    Io o;
    asm volatile(LOAD_INPUTS SLOW_SEQ STORE_OUTPUTS : OUTS : INS : CLOBBERS);
    return o;
}

template <int VARIANT>
__device__ __forceinline__ void role_frame(const float* lds_tw, int lane, int iters, unsigned seed, unsigned* bad_lanes, unsigned* bad_total) {
    unsigned bad = 0;
    const unsigned a1 = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)(lds_tw + 2 * lane);              // 3 x 512 B apart
    const unsigned a2 = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)(lds_tw + 384 + 2 * (lane & 15));  // 3 x 128 B apart
    Io in;
    for (int k = 0; k < 8; ++k) in.r[k] = (float)((lane * 37 + k * 101 + (int)(seed & 255)) % 2001 - 1000);
    for (int i = 0; i < iters; ++i) {
        const Io f = run_fast<VARIANT>(in, a1, a2);
        const Io s = run_slow(in, a1, a2);
#pragma unroll
        for (int k = 0; k < 8; ++k) bad += __float_as_uint(f.r[k]) != __float_as_uint(s.r[k]);
#pragma unroll
        for (int k = 0; k < 8; ++k) in.r[k] = fmaf(s.r[(k + 3) & 7], 0.03125f, (float)((lane + k + i) & 63) - 31.5f);
    }
    if (bad) { atomicAdd(bad_total, bad); atomicAdd(bad_lanes + lane, bad); }
}

__device__ __forceinline__ void role_mfma(float* out, int iters, int lane) {
    f32x4 acc[9];
    for (int c = 0; c < 9; ++c) acc[c] = {0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (1 + lane + i)); b[i] = (__bf16)0.5f; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < 9; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[c], 0, 0, 0);
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        bf16x2 p; p[0] = (__bf16)(acc[0][0] * 1e-3f); p[1] = (__bf16)(acc[1][1] * 1e-3f);
        a[i & 7] = p[0]; b[(i + 3) & 7] = p[1];
    }
    float s = 0.f;
    for (int c = 0; c < 9; ++c) s += acc[c][0] + acc[c][3];
    if (s == 12345.678f) out[lane] = s;
}

template <int VARIANT>
__global__ __launch_bounds__(1024) void k_hazard(unsigned* bad_lanes, unsigned* bad_total, float* out, int wa, int wb, int iters, int iters_b) {
    __shared__ __attribute__((aligned(16))) float tw[384 + 96 + 32];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 384 + 96 + 32; i += blockDim.x) tw[i] = cosf(0.01f * i + 0.3f);
    __syncthreads();
    if (wave < wa) role_frame<VARIANT>(tw, lane, iters, blockIdx.x * 977u + wave, bad_lanes, bad_total);
    else if (wave < wa + wb) role_mfma(out, iters_b, lane);
}

template <int VARIANT>
static void run(int cus, unsigned* bad, float* out, int wa, int wb, int iters) {
    (void)hipMemset(bad, 0, 65 * 4);
    const dim3 grid(cus * 2), block(64 * (wa + wb));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_hazard<VARIANT>, grid, block, 0, 0, bad, bad + 64, out, wa, wb, iters, iters * 6);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned h[65];
    (void)hipMemcpy(h, bad, sizeof h, hipMemcpyDeviceToHost);
    unsigned hi = 0;
    for (int l = 48; l < 64; ++l) hi += h[l];
    printf("%d,%d,%d,%d,%.3g,%u,%u,%.1f\n", VARIANT, wa, wb, iters, (double)grid.x * wa * 64 * 8.0 * iters, h[64], hi, ms);
    fflush(stdout);
}

int main() {
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    unsigned* bad; float* out;
    (void)hipMalloc(&bad, 65 * 4); (void)hipMalloc(&out, 4096);
    printf("variant,frame_waves,mfma_waves,iterations,values_checked,mismatches,mismatches_in_lanes_48_63,ms\n");
    const int iters = 60000;
    for (int wb : {0, 4, 8})
        for (int wa : {4, 8}) {
            run<0>(cus, bad, out, wa, wb, iters);
            run<1>(cus, bad, out, wa, wb, iters);
            run<2>(cus, bad, out, wa, wb, iters);
            run<3>(cus, bad, out, wa, wb, iters);
        }
    return 0;
}
