// Stock-width GRU (17 <= H <= 20, classic tiling of gru_device.h), TWO 16-stream tiles per wave, for gfx950.
//
// Network: /root/reference/precise/model.py:76-82, executed by Runner.predict
// (/root/reference/precise/network_runner.py:69-74); equations and tiling as in gru_device.h (gru_tile<5>).
//
// Why.  In the throughput regime (tens of thousands of streams per GPU) gru_tile<5> needs all four wave slots of a
// SIMD -- one dependent chain per wave, four chains per SIMD -- to keep the matrix pipe issuing; the MFCC frame waves of
// the same update then find no slot, and the fused launch costs the SUM of its two roles although they use different
// pipes (MFMA vs VALU + LDS).  Here a wave carries the chains of two tiles: the weight registers (the MFMA A operands:
// 41 of them) are shared, only accumulators, state and the feature-row prefetch double.  Two such waves per SIMD hold
// the four chains that used to take four waves, and the other two wave slots hold frame waves of the SAME launch
// (fused_update_mix_kernel, kernels.hip).
//
// MEASURED AND REJECTED (round 4; profiles/round4/r4b_*, r4c_*): on gfx950 an MFMA keeps the VALU of its SIMD
// from issuing for its whole pass count -- tools/micro/pipe_overlap.hip: an MFMA wave and a v_fma_f64 / v_fma_f32 /
// v_pk_fma_f32 wave on one SIMD take t_A + t_B together, whatever their priorities and whether the MFMAs form one
// dependent chain or four independent ones -- so co-resident roles cannot cost less than the sum of their MFMA and VALU
// time.  The mixed launch took 157-172 us per update at 65536 streams against 160 us for the round-3 launch; the pair
// kernel alone 78.2 us against 77.2.  Kept for tuning builds (-DPE_TUNING: PE_PAIR=1), not part of the product library.
//
// Bit-identical to gru_tile<5>: per accumulator the MFMAs are issued in the same order (bias as the first C operand,
// the four input k-steps, the recurrent k-steps rho = 0..4), and the gate arithmetic is the same spelled-out sequence.
// The two tiles are independent instruction streams that the scheduler interleaves.
#pragma once
#include "../../mycroft_precise_amd/csrc/gru_device.h"

namespace pe {

// Biases of the four output tiles as accumulator inits, fetched from LDS at every timestep instead of living in 16
// registers: lane (g, j) needs bias[tile][q] of row 4 g + q, i.e. the float4 at (tile * 4 + g) -- 16 lanes read the same
// 16 bytes (a broadcast, conflict-free).  256 bytes per workgroup.
constexpr int kPairBiasFloats = 4 * 4 * 4;

__device__ __forceinline__ void pair_bias_to_lds(const GruArgs& a, float* B, const int lane) {
    // bias[(t * 4 + q) * 64 + lane] with lane = 16 g + j is the bias of row 4 g + q of tile t (any j): lanes j == 0 copy
    if ((lane & 15) == 0) {
        const int g = lane >> 4;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) B[(t * 4 + g) * 4 + q] = a.bias[(t * 4 + q) * 64 + lane];
    }
}

// The input kernel's A operands for LDS: WX[(tile * 64 + lane) * 4 + kk] -- one 16-byte read per lane, output tile and
// timestep (each lane its own 16 bytes: conflict-free) instead of 16 resident registers.  4 KB per workgroup.
constexpr int kPairWxFloats = 4 * 64 * 4;
__device__ __forceinline__ void pair_wx_to_lds(const GruArgs& a, float* WX, const int lane) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        f32x4 w;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) w[kk] = a.wx[(t * 4 + kk) * 64 + lane];
        *reinterpret_cast<f32x4*>(WX + (t * 64 + lane) * 4) = w;
    }
}
constexpr int kPairLdsFloats = kPairBiasFloats + kPairWxFloats;       // [bias | wx]

// tiles tile0 and tile1 (tile1 >= n_tiles: the wave computes tile0 twice and stores once).  BIAS_LDS: the accumulator
// inits come from `B` (pair_bias_to_lds, visible to this wave), else from 16 registers; WX_LDS: the input kernel's A
// operands come from B + kPairBiasFloats (pair_wx_to_lds), else from 16 registers -- with both in LDS the wave fits the
// 128-register budget it shares with the frame role in the fused launch.
template <bool BIAS_LDS, bool WX_LDS = false>
__device__ __forceinline__ void gru_tile_pair(const GruArgs& a, const int tile0, const int tile1_in, const int n_tiles, const int lane, const float* B) {
#pragma clang fp contract(off)      // every fusion in the gate arithmetic is spelled out: all kernel shapes round alike
    constexpr int R = 5;
    using G = GruShape<R>;
    static_assert(G::NT == 4 && G::P1_END == 3 && G::P2_BEGIN == 2, "stock width");
    const int g = lane >> 4, j = lane & 15;
    const bool two = tile1_in < n_tiles;
    const int tiles[2] = {tile0, two ? tile1_in : tile0};
    const int T = a.n_features;

    float wx[WX_LDS ? 1 : 4][4], wr1[3][R], wr2[2][R];
    f32x4 bias[BIAS_LDS ? 1 : 4];
    const float* const WXL = B + kPairBiasFloats + lane * 4;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (!WX_LDS)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) wx[WX_LDS ? 0 : t][kk] = a.wx[(t * 4 + kk) * 64 + lane];
        if (!BIAS_LDS)
#pragma unroll
            for (int q = 0; q < 4; ++q) bias[t][q] = a.bias[(t * 4 + q) * 64 + lane];
    }
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int rho = 0; rho < R; ++rho) wr1[t][rho] = a.wr1[(t * R + rho) * 64 + lane];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int rho = 0; rho < R; ++rho) wr2[t][rho] = a.wr2[((t + 2) * R + rho) * 64 + lane];

    const uint32_t mask = (uint32_t)(a.ring_slots - 1);
    const float* xbase[2];
    uint32_t first[2];
    long long stream[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        stream[p] = (long long)tiles[p] * kTileStreams + j;
        uint32_t ke = a.st_ke[stream[p]];                                    // (counters exist for padded streams too)
        if (a.predict_ke) {
            // running beside the MFCC role of the same update: the emitted-frame count this update will produce, from the
            // state before it (same arithmetic as gru_tile / mfcc_book_tile)
            const int q = a.st_q[stream[p]];
            const uint32_t kc = a.st_kc[stream[p]];
            const int avail = q + a.chunk;
            const int nnew = avail >= a.frame_len ? 1 + (avail - a.frame_len) / a.hop : 0;
            const int qn = avail - nnew * a.hop;
            const int m = qn + a.hop * (int)(kc + (uint32_t)nnew - ke);
            if (m >= a.window) ke += 1u + (uint32_t)((m - a.window) / a.hop);
        }
        if (stream[p] >= a.n_streams) ke = 0u;                               // as gru_tile: padded streams read the zeroed rows
        first[p] = ke - (uint32_t)T;
        xbase[p] = a.ring + ((size_t)tiles[p] * a.ring_slots * kTileStreams + j) * kRowFloats + 4 * g;
    }
    auto load_x = [&](const int p, const int t) -> f32x4 {
        const int tc = t < T ? t : T - 1;
        const uint32_t slot = (first[p] + (uint32_t)tc) & mask;
        return *reinterpret_cast<const f32x4*>(xbase[p] + (size_t)slot * kTileStreams * kRowFloats);
    };

    float h[2][R];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int rho = 0; rho < R; ++rho) h[p][rho] = 0.f;
    f32x4 x[2] = {load_x(0, 0), load_x(1, 0)};
    for (int t = 0; t < T; ++t) {
        f32x4 xn[2] = {load_x(0, t + 1), load_x(1, t + 1)};
        f32x4 acc[2][4];
        // input projection, bias as the initial accumulator
#pragma unroll
        for (int tl = 0; tl < 4; ++tl) {
            // (volatile, named-address-space reads: these operands are loop-invariant, and a plain read would be hoisted out
            //  of the time loop into the very registers the LDS copy is there to save)
            typedef const volatile __attribute__((address_space(3))) f32x4* lds_vf4;
            f32x4 b0;
            if (BIAS_LDS) b0 = *(lds_vf4)(B + (tl * 4 + g) * 4);
            else b0 = bias[BIAS_LDS ? 0 : tl];
            f32x4 w;
            if (WX_LDS) w = *(lds_vf4)(WXL + tl * 256);
            else w = f32x4{wx[WX_LDS ? 0 : tl][0], wx[WX_LDS ? 0 : tl][1], wx[WX_LDS ? 0 : tl][2], wx[WX_LDS ? 0 : tl][3]};
#pragma unroll
            for (int p = 0; p < 2; ++p) acc[p][tl] = mfma(w[0], x[p][0], b0);
#pragma unroll
            for (int kk = 1; kk < 4; ++kk)
#pragma unroll
                for (int p = 0; p < 2; ++p) acc[p][tl] = mfma(w[kk], x[p][kk], acc[p][tl]);
        }
        // phase 1: + h . U on the tiles that hold r rows (tiles 1, 2), k-outer as in gru_tile
#pragma unroll
        for (int rho = 0; rho < R; ++rho)
#pragma unroll
            for (int tl = 1; tl < 3; ++tl)
#pragma unroll
                for (int p = 0; p < 2; ++p) acc[p][tl] = mfma(wr1[tl][rho], h[p][rho], acc[p][tl]);
        float rh[2][R];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int rho = 0; rho < R; ++rho) {
                const int sr = R + rho;
                rh[p][rho] = hard_sigmoid(acc[p][sr >> 2][sr & 3]) * h[p][rho];
            }
        // phase 2: + (r*h) . U on the candidate tiles (2, 3), and the z-only tile's (0) share of phase 1
#pragma unroll
        for (int rho = 0; rho < R; ++rho) {
#pragma unroll
            for (int tl = 2; tl < 4; ++tl)
#pragma unroll
                for (int p = 0; p < 2; ++p) acc[p][tl] = mfma(wr2[tl - 2][rho], rh[p][rho], acc[p][tl]);
#pragma unroll
            for (int p = 0; p < 2; ++p) acc[p][0] = mfma(wr1[0][rho], h[p][rho], acc[p][0]);
        }
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int rho = 0; rho < R; ++rho) {
                const float z = hard_sigmoid(acc[p][rho >> 2][rho & 3]);
                const int sh = 2 * R + rho;
                h[p][rho] = gru_blend(z, h[p][rho], acc[p][sh >> 2][sh & 3]);
            }
        x[0] = xn[0]; x[1] = xn[1];
    }

    // Dense(1) + sigmoid: reduce over this lane's units, then over the four lane groups
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        float part = 0.f;
#pragma unroll
        for (int rho = 0; rho < R; ++rho) part = fmaf(h[p][rho], a.wd[rho * 64 + lane], part);
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        if (stream[p] < a.n_streams && g == 0 && (p == 0 || two)) {
            const float logit = part + a.dense_bias;
            a.out[stream[p]] = 1.0f / (1.0f + expf(-logit));
        }
    }
}

}  // namespace pe
