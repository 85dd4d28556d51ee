// Microbenchmark of the stock GRU window (H = 20, F = 13, T = 29) at one 16-stream tile per compute unit: the shipped
// four-wave shapes side by side, with shader-clock section timers of one wave per workgroup (-DPE_GRU_TIMERS).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -DPE_CW_BIG_BOX -DPE_GRU_TIMERS -I mycroft_precise_amd/csrc -I include \
//         tools/micro/gru_chain.hip -o tools/micro/build/gru_chain && tools/micro/build/gru_chain [streams]
// Prints, per shape: launch time (HIP events over back-to-back launches), the largest deviation from a plain float32
// CPU evaluation, whether the shapes agree bit for bit, and the mean section stamps over the workgroups.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#ifndef PE_CW_BIG_BOX
#define PE_CW_BIG_BOX 1       // every split variant of gru_tile_cw in one binary: the larger mailbox area
#endif
#include "gru_device.h"
#include "gru_cw_device.h"
#include "gru_cw_pack.h"

using namespace pe;
constexpr int H = 20, F = 13, T = 29, SLOTS = 32;

__global__ __launch_bounds__(256) void k_mw5(const GruArgs a) {
    __shared__ __attribute__((aligned(16))) float S[3 * 5 * 64 + 256];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    gru_tile_mw5<false>(a, blockIdx.x, wave, threadIdx.x & 63, S);
}
template <bool VF, int VAR = 0>
__global__ __launch_bounds__(256) void k_cw(const GruArgs a) {
    extern __shared__ __attribute__((aligned(16))) float Sd[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    gru_tile_cw<VF, VAR>(a, blockIdx.x, wave, threadIdx.x & 63, Sd);
}
__global__ __launch_bounds__(64) void k_one(const GruArgs a) { gru_tile<5, kRing, false>(a, blockIdx.x, threadIdx.x); }
__global__ __launch_bounds__(64) void k_v(const GruArgs a) { gru_tile_v<kRing, false>(a, blockIdx.x, threadIdx.x); }

template <class Tv> static Tv* upload(const std::vector<Tv>& v) {
    Tv* d; hipMalloc(&d, v.size() * sizeof(Tv)); hipMemcpy(d, v.data(), v.size() * sizeof(Tv), hipMemcpyHostToDevice); return d;
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IOLBF, 0);      // (a variant that hangs must not take the lines before it along)
    const int n_streams = argc > 1 ? atoi(argv[1]) : 4096, tiles = (n_streams + 15) / 16;
    std::vector<float> kernel((size_t)F * 3 * H), rec((size_t)H * 3 * H), bias(3 * H), wdv(H);
    srand(7);
    auto rnd = [](float s) { return s * ((float)rand() / RAND_MAX * 2.f - 1.f); };
    for (auto& v : kernel) v = rnd(0.5f);
    for (auto& v : rec) v = rnd(0.4f);
    for (auto& v : bias) v = rnd(0.3f);
    for (auto& v : wdv) v = rnd(0.8f);
    const float bd = 0.1f;
    std::vector<float> ring((size_t)tiles * SLOTS * 16 * 16, 0.f);
    std::vector<uint32_t> ke(tiles * 16);
    for (auto& k : ke) k = (uint32_t)(T + rand() % 1000);
    for (size_t i = 0; i < ring.size(); ++i) ring[i] = (i % 16) < (size_t)F ? rnd(2.0f) : 0.f;

    // old packing (engine.hip pack_gru_weights, R = 5, NT = 4)
    const int R = 5, NT = 4;
    std::vector<float> wx((size_t)NT * 4 * 64, 0.f), wr1((size_t)NT * R * 64, 0.f), wr2((size_t)NT * R * 64, 0.f), pb((size_t)NT * 4 * 64, 0.f), wd((size_t)R * 64, 0.f);
    for (int tile = 0; tile < NT; ++tile)
        for (int lane = 0; lane < 64; ++lane) {
            const int i = lane & 15, g = lane >> 4;
            const int reg = i & 3, gout = i >> 2, slot = 4 * tile + reg, gate = slot / R, rho = slot % R, u = 4 * rho + gout;
            if (slot < 3 * R && u < H) {
                const int col = gate * H + u;
                for (int kk = 0; kk < 4; ++kk) if (4 * g + kk < F) wx[((size_t)tile * 4 + kk) * 64 + lane] = kernel[(size_t)(4 * g + kk) * 3 * H + col];
                for (int rs = 0; rs < R; ++rs) if (4 * rs + g < H) (gate < 2 ? wr1 : wr2)[((size_t)tile * R + rs) * 64 + lane] = rec[(size_t)(4 * rs + g) * 3 * H + col];
            }
            for (int q = 0; q < 4; ++q) {
                const int s2 = 4 * tile + q, g2 = s2 / R, r2 = s2 % R, u2 = 4 * r2 + g;
                if (s2 < 3 * R && u2 < H) pb[((size_t)tile * 4 + q) * 64 + lane] = bias[g2 * H + u2];
            }
        }
    for (int rho = 0; rho < R; ++rho) for (int lane = 0; lane < 64; ++lane) if (4 * rho + (lane >> 4) < H) wd[(size_t)rho * 64 + lane] = wdv[4 * rho + (lane >> 4)];
    const std::vector<float> cw = pack_gru_cw(kernel.data(), rec.data(), bias.data(), F, H);

    GruArgs a{};
    a.n_streams = n_streams; a.n_features = T; a.n_in = F; a.units = H;
    a.wx = upload(wx); a.wxd = a.wx; a.wr1 = upload(wr1); a.wr2 = upload(wr2); a.bias = upload(pb); a.wd = upload(wd); a.dense_bias = bd;
    a.cw = upload(cw);
    a.ring = upload(ring); a.st_ke = upload(ke); a.ring_slots = SLOTS; a.predict_ke = 0;
    float* d_out; hipMalloc(&d_out, tiles * 16 * 4); a.out = d_out;
    a.waves_per_tile = 4;

    // CPU check (float32, plain order)
    std::vector<float> ref(n_streams);
    for (int s = 0; s < n_streams; ++s) {
        float h[H] = {0};
        const int tile = s / 16, j = s % 16;
        for (int t = 0; t < T; ++t) {
            const float* x = &ring[(((size_t)tile * SLOTS + ((ke[s] - T + t) & (SLOTS - 1))) * 16 + j) * 16];
            float z[H], r[H], hh[H];
            for (int u = 0; u < H; ++u) {
                float az = bias[u], ar = bias[H + u];
                for (int k = 0; k < F; ++k) { az += x[k] * kernel[(size_t)k * 3 * H + u]; ar += x[k] * kernel[(size_t)k * 3 * H + H + u]; }
                for (int k = 0; k < H; ++k) { az += h[k] * rec[(size_t)k * 3 * H + u]; ar += h[k] * rec[(size_t)k * 3 * H + H + u]; }
                z[u] = fminf(fmaxf(0.2f * az + 0.5f, 0.f), 1.f); r[u] = fminf(fmaxf(0.2f * ar + 0.5f, 0.f), 1.f);
            }
            for (int u = 0; u < H; ++u) {
                float ac = bias[2 * H + u];
                for (int k = 0; k < F; ++k) ac += x[k] * kernel[(size_t)k * 3 * H + 2 * H + u];
                for (int k = 0; k < H; ++k) ac += r[k] * h[k] * rec[(size_t)k * 3 * H + 2 * H + u];
                hh[u] = ac;
            }
            for (int u = 0; u < H; ++u) h[u] = z[u] * h[u] + (1.f - z[u]) * hh[u];
        }
        float p = bd;
        for (int u = 0; u < H; ++u) p += h[u] * wdv[u];
        ref[s] = 1.f / (1.f + expf(-p));
    }

    const size_t cw_lds = kCwLdsBytes;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<std::vector<float>> outs;
    auto run = [&](const char* name, int which) {
        auto launch = [&]() {
            if (which == 0) hipLaunchKernelGGL(k_mw5, dim3(tiles), dim3(256), 0, 0, a);
            else if (which == 1) hipLaunchKernelGGL(k_cw<true>, dim3(tiles), dim3(256), cw_lds, 0, a);
            else if (which == 4) hipLaunchKernelGGL(k_cw<false>, dim3(tiles), dim3(256), cw_lds, 0, a);
            else if (which == 5) hipLaunchKernelGGL((k_cw<false, 1>), dim3(tiles), dim3(256), cw_lds, 0, a);
            else if (which == 6) hipLaunchKernelGGL((k_cw<false, 2>), dim3(tiles), dim3(256), cw_lds, 0, a);
            else if (which == 7) hipLaunchKernelGGL((k_cw<false, 3>), dim3(tiles), dim3(256), cw_lds, 0, a);
            else if (which == 8) hipLaunchKernelGGL((k_cw<false, 5>), dim3(tiles), dim3(256), cw_lds, 0, a);
            else if (which == 9) hipLaunchKernelGGL((k_cw<false, 4>), dim3(tiles), dim3(256), cw_lds, 0, a);
            else if (which == 10) hipLaunchKernelGGL((k_cw<false, 8>), dim3(tiles), dim3(256), cw_lds, 0, a);
            else if (which == 11) hipLaunchKernelGGL((k_cw<false, 21>), dim3(tiles), dim3(256), cw_lds, 0, a);
            else if (which == 12) hipLaunchKernelGGL((k_cw<false, 20>), dim3(tiles), dim3(256), cw_lds, 0, a);
            else if (which == 13) hipLaunchKernelGGL((k_cw<false, 37>), dim3(tiles), dim3(256), cw_lds, 0, a);
            else if (which == 2) hipLaunchKernelGGL(k_one, dim3(tiles), dim3(64), 0, 0, a);
            else if (which == 3) hipLaunchKernelGGL(k_v, dim3(tiles), dim3(64), 0, 0, a);
        };
        hipMemset(d_out, 0, tiles * 16 * 4);
        launch();
        if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed: %s\n", name, hipGetErrorString(hipGetLastError())); return; }
        std::vector<float> out(n_streams);
        hipMemcpy(out.data(), d_out, n_streams * 4, hipMemcpyDeviceToHost);
        double worst = 0;
        for (int s = 0; s < n_streams; ++s) worst = fmax(worst, fabs((double)ref[s] - out[s]));
        for (int i = 0; i < 20; ++i) launch();
        hipDeviceSynchronize();
        const int reps = 300;
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("%-8s %d streams: %.2f us per launch (back to back), max |p - cpu| = %.3g, out[0] = %.7f\n", name, n_streams, ms / reps * 1e3, worst, out[0]);
#ifdef PE_GRU_TIMERS
        if (which <= 1 || which >= 4) {
            std::vector<unsigned long long> tm(256 * 32);
            hipMemcpyFromSymbol(tm.data(), HIP_SYMBOL(pe_gru_timers), tm.size() * 8);
            const int nb = tiles < 256 ? tiles : 256;
            const int idx[] = {1, 2, 3, 4, 5, 6, 7, 8, 9};
            printf("   stamps (mean over %d workgroups, cycles after the kernel top):", nb);
            for (int k : idx) {
                double sum = 0; int n = 0;
                for (int b = 0; b < nb; ++b) if (tm[b * 32 + k] > tm[b * 32]) { sum += (double)(tm[b * 32 + k] - tm[b * 32]); ++n; }
                if (n) printf("  [%d] %.0f", k, sum / n);
            }
            printf("\n");
            std::vector<unsigned long long> zero(256 * 32, 0);
            hipMemcpyToSymbol(HIP_SYMBOL(pe_gru_timers), zero.data(), zero.size() * 8);
        }
#endif
        outs.push_back(out);
    };
    run("mw5", 0);
    run("cw_valu", 1);
    run("cw_mfma4", 4);
    run("one", 2);
    run("v", 3);
    run("cw_var1", 5);       // r partial sums of units 16..19 issued with the X chain
    run("cw_var2", 6);       // candidate of units 16..19 on Z2, all of z on Z1
    run("cw_var3", 7);       // both
    run("cw_var5", 8);       // r partial sums with the X chain + candidate chain / partial sums pinned
    run("cw_var4", 9);       // candidate chain / partial sums pinned only
    run("cw_var8", 10);      // R's whole timestep hand-scheduled
    run("cw_var21", 11);     // var5 + R's mailbox reads before the barrier (tag-validated)
    run("cw_var20", 12);     // var4 + the same
    run("cw_var37", 13);     // var5 without the s_barrier of a timestep: every hand-off tag-polled
    for (size_t k = 1; k < outs.size(); ++k) {
        int diff = 0; double md = 0;
        const int base = k != 3 ? 1 : 0;      // the re-tiled shapes among themselves, the old ones among themselves
        for (int s = 0; s < n_streams; ++s) { if (memcmp(&outs[base][s], &outs[k][s], 4)) ++diff; md = fmax(md, fabs((double)outs[base][s] - outs[k][s])); }
        printf("shape %zu vs shape %d: %d of %d outputs differ in bits (max %.3g)\n", k, base, diff, n_streams, md);
    }
    return 0;
}
