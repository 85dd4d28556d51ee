// Microbenchmark / hardware check: does a wave's ds_write -> ds_read hand-off between its own lanes need a counter wait?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_order.hip -o tools/micro/build/lds_order
//
// The MFCC frame role hands values from lane to lane of ONE wave through wave-private LDS with no s_waitcnt in between
// ("LDS instructions of one wave execute in program order", mfcc_device.h: group_sync).  Round 4 saw float32 frames go
// slightly wrong -- stale values -- when bf16-MFMA-heavy waves shared their compute unit.  This kernel isolates the
// pattern: waves of role A repeat {4 x ds_write_b64 (stride-5 slots), compiler fence, 4 x ds_read_b64 from other lanes,
// check} with values that change every iteration; waves of role B spin on v_mfma_f32_16x16x32_bf16 (or idle).
// Any mismatch is a read that did not see the write issued before it (or saw a later one).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kStride = 5;          // complex slots per lane (the exchange padding of the frame kernel)

__device__ __forceinline__ void fence_only() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void fence_wait() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

template <bool WAIT, int BYTES>
__device__ __forceinline__ unsigned role_lds(unsigned long long* S, int lane, int iters) {
    typedef volatile __attribute__((address_space(3))) unsigned long long* lds_ptr;
    unsigned bad = 0;
    for (int i = 0; i < iters; ++i) {
        // values that identify (iteration, lane, register)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned long long v = ((unsigned long long)(unsigned)i << 32) | (unsigned)(lane * 4 + r);
            if (BYTES == 16) { S[2 * (lane * kStride + r)] = v; S[2 * (lane * kStride + r) + 1] = ~v; }
            else S[lane * kStride + r] = v;
        }
        if (WAIT) fence_wait(); else fence_only();
        const int src = (lane * 7 + i) & 63;                 // another lane, a different one every iteration
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = (r + i) & 3;
            const unsigned long long want = ((unsigned long long)(unsigned)i << 32) | (unsigned)(src * 4 + rr);
            unsigned long long got;
            if (BYTES == 16) got = *(lds_ptr)(S + 2 * (src * kStride + rr));
            else got = *(lds_ptr)(S + src * kStride + rr);
            bad += got != want;
        }
        if (WAIT) fence_wait(); else fence_only();
    }
    return bad;
}

__device__ __forceinline__ void role_mfma(float* out, int iters, int lane) {
    f32x4 acc[4];
    for (int c = 0; c < 4; ++c) acc[c] = {0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + lane + i); b[i] = (__bf16)0.5f; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    if (s == 12345.678f) out[lane] = s;
}
__device__ __forceinline__ void role_mfma_f32(float* out, int iters, int lane) {
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

// waves [0, wa): LDS role; waves [wa, wa + wb): partner role (0 none, 1 bf16 MFMA, 2 f32 MFMA)
template <bool WAIT, int BYTES>
__global__ __launch_bounds__(1024) void k_order(unsigned* bad_out, float* out, int wa, int partner, int iters, int iters_b) {
    extern __shared__ unsigned long long smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < wa) {
        const unsigned bad = role_lds<WAIT, BYTES>(smem + (size_t)wave * 64 * kStride * (BYTES / 8), lane, iters);
        if (bad) atomicAdd(bad_out, bad);
    } else if (partner == 1) role_mfma(out, iters_b, lane);
    else if (partner == 2) role_mfma_f32(out, iters_b, lane);
}

int main() {
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    unsigned* bad; float* out;
    (void)hipMalloc(&bad, 4); (void)hipMalloc(&out, 4096);
    printf("bytes,wait,lds_waves,partner,partner_waves,iterations,reads_checked,mismatches\n");
    const char* pn[] = {"none", "mfma_bf16_16x16x32", "mfma_f32_16x16x4"};
    for (int bytes : {8, 16})
        for (int wait = 0; wait < 2; ++wait)
            for (int partner = 0; partner < 3; ++partner)
                for (int wa : {4, 8}) {
                    const int wb = partner ? 4 : 0, iters = 200000;
                    (void)hipMemset(bad, 0, 4);
                    const size_t lds = (size_t)wa * 64 * kStride * bytes;
                    const dim3 grid(cus * 2), block(64 * (wa + wb));
                    if (bytes == 8 && !wait) hipLaunchKernelGGL((k_order<false, 8>), grid, block, lds, 0, bad, out, wa, partner, iters, iters * 3);
                    if (bytes == 8 && wait) hipLaunchKernelGGL((k_order<true, 8>), grid, block, lds, 0, bad, out, wa, partner, iters, iters * 3);
                    if (bytes == 16 && !wait) hipLaunchKernelGGL((k_order<false, 16>), grid, block, lds, 0, bad, out, wa, partner, iters, iters * 3);
                    if (bytes == 16 && wait) hipLaunchKernelGGL((k_order<true, 16>), grid, block, lds, 0, bad, out, wa, partner, iters, iters * 3);
                    (void)hipDeviceSynchronize();
                    unsigned h = 0;
                    (void)hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
                    printf("%d,%d,%d,%s,%d,%d,%.3g,%u\n", bytes, wait, wa, pn[partner], wb, iters, (double)grid.x * wa * 64 * 4.0 * iters, h);
                    fflush(stdout);
                }
    return 0;
}
