// Microbenchmark: what EVERY compute unit can pull per clock when all of them stream the SAME few megabytes again and again
// -- the access pattern of the streamed-weight wide GRU (gru_wide_device.h: one workgroup per compute unit walks the whole
// weight set once per timestep), as the measured counterpart of the 34.5 TB/s (~56 B/clk/CU) L2 figure of MI355X_MICROARCH.md.
// Decides what an XDL ("three bf16 pieces") form of that kernel can gain: 6 B per weight instead of 4 at 2.7 x the matrix rate.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/l2_stream.hip -o tools/micro/build/l2_stream
// Workgroups of 256 threads, one per compute unit (256) or two; each walks `bytes` with 16-byte loads, `ahead` wave-instructions
// (1 KB each) in flight per wave, `passes` times.  Reported: B/clk/CU at the 2.4 GHz nominal clock and aggregate TB/s.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int AHEAD>
__global__ __launch_bounds__(256) void k_stream(const uint4* __restrict__ src, unsigned* __restrict__ out, const int n16, const int passes, const int stagger) {
    // wave w of a workgroup walks quarter w of the buffer (as the four waves of the wide kernel walk their own weight streams)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int per_wave = n16 / 4;
    const uint4* base = src + (size_t)wave * per_wave + lane;
    const int steps = per_wave / 64;
    unsigned acc = 0;
    for (int i = (blockIdx.x >> 3) & 31; stagger && i > 0; --i) __builtin_amdgcn_s_sleep(64);
    for (int p = 0; p < passes; ++p) {
        uint4 v[AHEAD];
#pragma unroll
        for (int a = 0; a < AHEAD; ++a) v[a] = base[(size_t)(a < steps ? a : 0) * 64];
        for (int s = 0; s < steps; s += AHEAD) {
#pragma unroll
            for (int a = 0; a < AHEAD; ++a) {
                acc += (v[a].x ^ v[a].w) + (v[a].y ^ v[a].z);       // (every component: with two of four used, hipcc narrows the loads to dwords)
                const int nx = s + a + AHEAD;
                v[a] = base[(size_t)(nx < steps ? nx : 0) * 64];
            }
        }
#pragma unroll
        for (int a = 0; a < AHEAD; ++a) acc += v[a].y + v[a].z + v[a].x + v[a].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    unsigned* out;
    (void)hipMalloc(&out, 4);
    printf("kbytes,workgroups,ahead,stagger,passes,ms,B_per_clk_per_CU_at_2.4GHz,aggregate_TB_per_s\n");
    for (int kb : {1024, 2400, 3000, 3600, 4200, 4800, 7200}) {
        const size_t bytes = (size_t)kb << 10;
        const int n16 = (int)(bytes / 16) / 256 * 256;
        uint4* src;
        (void)hipMalloc(&src, bytes);
        (void)hipMemset(src, 1, bytes);
        for (int wgs : {cus, 2 * cus})
            for (int stagger : {0, 1})
                for (int ahead : {4, 8, 16}) {
                    const int passes = 58;
                    auto launch = [&] {
                        if (ahead == 4) hipLaunchKernelGGL(k_stream<4>, dim3(wgs), dim3(256), 0, 0, src, out, n16, passes, stagger);
                        else if (ahead == 8) hipLaunchKernelGGL(k_stream<8>, dim3(wgs), dim3(256), 0, 0, src, out, n16, passes, stagger);
                        else hipLaunchKernelGGL(k_stream<16>, dim3(wgs), dim3(256), 0, 0, src, out, n16, passes, stagger);
                    };
                    launch();
                    (void)hipDeviceSynchronize();
                    hipEvent_t e0, e1;
                    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
                    (void)hipEventRecord(e0);
                    launch();
                    (void)hipEventRecord(e1);
                    (void)hipEventSynchronize(e1);
                    float ms = 0;
                    (void)hipEventElapsedTime(&ms, e0, e1);
                    const double total = (double)n16 * 16 * passes * wgs;
                    printf("%d,%d,%d,%d,%d,%.3f,%.1f,%.2f\n", kb, wgs, ahead, stagger, passes, ms,
                           total / cus / (ms * 1e-3 * 2.4e9), total / (ms * 1e-3) / 1e12);
                    fflush(stdout);
                }
        (void)hipFree(src);
    }
    return 0;
}
