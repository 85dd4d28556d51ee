// Prototype + microbenchmark: the stock GRU window (H = 20, F = 13, T = 29) with FOUR STREAMS PER WAVE on
// v_mfma_f32_4x4x1_16B_f32 -- the 16 blocks of the instruction are 16 groups of 4 gate rows, the 4 columns of a block are
// the wave's 4 streams, one MFMA per contraction index k.  The four waves of a 16-stream tile never talk to each other:
// the hand-off between the two phases of a timestep (candidate needs r * h) and between timesteps (every block needs
// h as its B operand) goes through 720 bytes of wave-private LDS, no barrier.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/gru_quad.hip -o tools/micro/build/gru_quad && tools/micro/build/gru_quad
// Prints the launch time at 4096 streams (one tile per compute unit, as in the engine) and the largest deviation from a
// plain float32 CPU evaluation of the same windows.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int H = 20, F = 13, T = 29, SLOTS = 32, ROW = 16;
#ifndef NACC
#define NACC 2
#endif

struct Args {
    const float* wq;      // [16 + 20][64]  A operands: feature k (0..15), then source unit k (0..19)
    const float* bias4;   // [4][64]        accumulator init of output register i
    const float* wd4;     // [4][64]        dense kernel of the unit a z lane owns in register i (0 elsewhere)
    const float* ring;    // [tiles][SLOTS][16 streams][ROW]
    const unsigned* first;// [streams] slot of the window's first row
    float* out;
    float bd;
    int n_streams;
};

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float hsig(float v) { return __builtin_amdgcn_fmed3f(__builtin_fmaf(0.2f, v, 0.5f), 0.0f, 1.0f); }
__device__ __forceinline__ void wave_sync() { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); }

// block roles: z 0..4, r 5,6,7,13,14, candidate 8..12 (32 lanes above z: one v_permlane32_swap brings c to z)
__host__ __device__ inline int role_of(int b) { return b < 5 ? 0 : (b < 8 || b == 13 || b == 14) ? 1 : (b < 13) ? 2 : 3; }
__host__ __device__ inline int group_of(int b) { return b < 5 ? b : b < 8 ? b - 5 : b < 13 ? b - 8 : b - 10; }

__global__ __launch_bounds__(256) void gru_quad_kernel(const Args a) {
    __shared__ __attribute__((aligned(16))) float lds[4][3 * 4 * H + 16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = lane >> 2, j = lane & 3;
    const int role = role_of(b), ug = group_of(b);
    const int tile = blockIdx.x, j16 = 4 * wave + j;
    const int stream = tile * 16 + j16;
    float* hbuf = lds[wave];                    // [4][H]
    float* rbuf = hbuf + 4 * H;                 // [4][H]
    float* zbuf = rbuf + 4 * H;                 // [H] zeros (+ pad)
    if (lane < H + 12) zbuf[lane] = 0.f;
    for (int i = lane; i < 8 * H; i += 64) hbuf[i] = 0.f;       // h0 = 0, rh irrelevant
    // resident A operands
    float wx[F], wr[H];
#pragma unroll
    for (int k = 0; k < F; ++k) wx[k] = a.wq[k * 64 + lane];
#pragma unroll
    for (int k = 0; k < H; ++k) wr[k] = a.wq[(16 + k) * 64 + lane];
    f32x4 bias;
#pragma unroll
    for (int i = 0; i < 4; ++i) bias[i] = a.bias4[i * 64 + lane];
    const unsigned first = a.first[stream < a.n_streams ? stream : 0];
    const float* xbase = a.ring + ((size_t)tile * SLOTS * 16 + j16) * ROW;
    auto load_x = [&](int t, f32x4 (&x)[4]) {
        const int tc = t < T ? t : T - 1;
        const float* p = xbase + (size_t)((first + (unsigned)tc) & (SLOTS - 1)) * 16 * ROW;
#pragma unroll
        for (int q = 0; q < 4; ++q) x[q] = *reinterpret_cast<const f32x4*>(p + 4 * q);
    };
    // where this lane reads its B operands from: h for z / r rows, r*h for candidate rows, zeros otherwise
    const float* b1 = (role <= 1) ? hbuf + j * H : zbuf;
    const float* b2 = (role == 2) ? rbuf + j * H : zbuf;
    const int own = j * H + 4 * ug;             // this lane's four units in hbuf / rbuf
    f32x4 x[4];
    load_x(0, x);
    wave_sync();
    f32x4 hb[5];
#pragma unroll
    for (int m = 0; m < 5; ++m) hb[m] = *reinterpret_cast<const f32x4*>(b1 + 4 * m);
    f32x4 hown = *reinterpret_cast<const f32x4*>(hbuf + own);
    for (int t = 0; t < T; ++t) {
        f32x4 xn[4];
        load_x(t + 1, xn);
        // input projection (could run a step ahead; the compiler is free to move it)
        f32x4 acc[NACC];
        acc[0] = bias;
#pragma unroll
        for (int n = 1; n < NACC; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < F; ++k) acc[k % NACC] = mfma4(wx[k], x[k >> 2][k & 3], acc[k % NACC]);
        // phase 1: z and r rows += U h
#pragma unroll
        for (int k = 0; k < H; ++k) acc[k % NACC] = mfma4(wr[k], hb[k >> 2][k & 3], acc[k % NACC]);
        f32x4 s1 = acc[0];
#pragma unroll
        for (int n = 1; n < NACC; ++n) s1 += acc[n];
        f32x4 g;
#pragma unroll
        for (int i = 0; i < 4; ++i) g[i] = hsig(s1[i]);
        if (role == 1) {
            f32x4 rh;
#pragma unroll
            for (int i = 0; i < 4; ++i) rh[i] = g[i] * hown[i];
            *reinterpret_cast<f32x4*>(rbuf + own) = rh;
        }
        wave_sync();
        f32x4 rb[5];
#pragma unroll
        for (int m = 0; m < 5; ++m) rb[m] = *reinterpret_cast<const f32x4*>(b2 + 4 * m);
        // phase 2: candidate rows += U (r * h)
        acc[0] = s1;
#pragma unroll
        for (int n = 1; n < NACC; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < H; ++k) acc[k % NACC] = mfma4(wr[k], rb[k >> 2][k & 3], acc[k % NACC]);
        f32x4 c = acc[0];
#pragma unroll
        for (int n = 1; n < NACC; ++n) c += acc[n];
        // candidate rows sit 32 lanes above the z rows of the same units
        f32x4 cz;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned hi = __float_as_uint(c[i]), lo = 0;
            const auto r = __builtin_amdgcn_permlane32_swap(hi, lo, false, false);     // hi[32..63] <-> lo[0..31]
            cz[i] = __uint_as_float(r[1]);
        }
        if (role == 0) {
            f32x4 hn;
#pragma unroll
            for (int i = 0; i < 4; ++i) hn[i] = __builtin_fmaf(g[i], hown[i], (1.0f - g[i]) * cz[i]);
            *reinterpret_cast<f32x4*>(hbuf + own) = hn;
        }
        wave_sync();
#pragma unroll
        for (int m = 0; m < 5; ++m) hb[m] = *reinterpret_cast<const f32x4*>(b1 + 4 * m);
        hown = *reinterpret_cast<const f32x4*>(hbuf + own);
#pragma unroll
        for (int q = 0; q < 4; ++q) x[q] = xn[q];
    }
    // dense + sigmoid: z lanes hold h of their units
    float part = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) part = __builtin_fmaf(hown[i], a.wd4[i * 64 + lane], part);
    float tot = 0.f;
#pragma unroll
    for (int bb = 0; bb < 5; ++bb) tot += __shfl(part, 4 * bb + j, 64);
    if (lane < 4 && stream < a.n_streams) a.out[stream] = 1.0f / (1.0f + expf(-(tot + a.bd)));
}

int main() {
    const int n_streams = 4096, tiles = n_streams / 16;
    std::vector<float> kernel((size_t)F * 3 * H), rec((size_t)H * 3 * H), bias(3 * H), wd(H);
    srand(7);
    auto rnd = [](float s) { return s * ((float)rand() / RAND_MAX * 2.f - 1.f); };
    for (auto& v : kernel) v = rnd(0.5f);
    for (auto& v : rec) v = rnd(0.4f);
    for (auto& v : bias) v = rnd(0.3f);
    for (auto& v : wd) v = rnd(0.8f);
    const float bd = 0.1f;
    std::vector<float> ring((size_t)tiles * SLOTS * 16 * ROW, 0.f);
    std::vector<unsigned> first(n_streams);
    for (int s = 0; s < n_streams; ++s) first[s] = (unsigned)(rand() % 1000);
    for (size_t i = 0; i < ring.size(); ++i) ring[i] = (i % ROW) < (size_t)F ? rnd(2.0f) : 0.f;
    // pack
    std::vector<float> wq((size_t)36 * 64, 0.f), bias4(4 * 64, 0.f), wd4(4 * 64, 0.f);
    for (int lane = 0; lane < 64; ++lane) {
        const int b = lane >> 2, i = lane & 3, role = role_of(b), u = 4 * group_of(b) + i;
        if (role == 3 || u >= H) continue;
        const int col = role == 0 ? u : role == 1 ? H + u : 2 * H + u;
        for (int k = 0; k < F; ++k) wq[(size_t)k * 64 + lane] = kernel[(size_t)k * 3 * H + col];
        for (int k = 0; k < H; ++k) wq[(size_t)(16 + k) * 64 + lane] = rec[(size_t)k * 3 * H + col];
    }
    for (int lane = 0; lane < 64; ++lane)
        for (int i = 0; i < 4; ++i) {
            const int b = lane >> 2, role = role_of(b), u = 4 * group_of(b) + i;
            if (role == 3 || u >= H) continue;
            bias4[i * 64 + lane] = bias[(role == 0 ? 0 : role == 1 ? H : 2 * H) + u];
            if (role == 0) wd4[i * 64 + lane] = wd[u];
        }
    Args a{};
    float *d_wq, *d_b4, *d_wd4, *d_ring, *d_out; unsigned* d_first;
    hipMalloc(&d_wq, wq.size() * 4); hipMalloc(&d_b4, bias4.size() * 4); hipMalloc(&d_wd4, wd4.size() * 4);
    hipMalloc(&d_ring, ring.size() * 4); hipMalloc(&d_out, n_streams * 4); hipMalloc(&d_first, n_streams * 4);
    hipMemcpy(d_wq, wq.data(), wq.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_b4, bias4.data(), bias4.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_wd4, wd4.data(), wd4.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_ring, ring.data(), ring.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_first, first.data(), n_streams * 4, hipMemcpyHostToDevice);
    a.wq = d_wq; a.bias4 = d_b4; a.wd4 = d_wd4; a.ring = d_ring; a.first = d_first; a.out = d_out; a.bd = bd; a.n_streams = n_streams;
    hipLaunchKernelGGL(gru_quad_kernel, dim3(tiles), dim3(256), 0, 0, a);
    std::vector<float> out(n_streams);
    hipMemcpy(out.data(), d_out, n_streams * 4, hipMemcpyDeviceToHost);
    // CPU check (float32)
    double worst = 0;
    for (int s = 0; s < n_streams; s += 7) {
        float h[H] = {0};
        const int tile = s / 16, j16 = s % 16;
        for (int t = 0; t < T; ++t) {
            const float* x = &ring[(((size_t)tile * SLOTS + ((first[s] + t) & (SLOTS - 1))) * 16 + j16) * ROW];
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
        for (int u = 0; u < H; ++u) p += h[u] * wd[u];
        p = 1.f / (1.f + expf(-p));
        worst = fmax(worst, fabs((double)p - out[s]));
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(gru_quad_kernel, dim3(tiles), dim3(256), 0, 0, a);
    hipDeviceSynchronize();
    const int reps = 500;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gru_quad_kernel, dim3(tiles), dim3(256), 0, 0, a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("gru_quad NACC=%d: %d streams, %.2f us per launch, max |p - cpu| = %.3g (out[0]=%.6f)\n", NACC, n_streams, ms / reps * 1e3, worst, out[0]);
    return worst < 1e-4 ? 0 : 1;
}
