// Wide / stacked GRU + Dense forward (BASELINE.json configs[3]: "wide GRU, 256 hidden units, 2 stacked
// layers ... MFMA-bound regime"), float32 on v_mfma_f32_16x16x4_f32, gfx950.
//
// The reference builds exactly one small GRU (model.py:76-82); this is the same recurrence
// (gru_device.h header) for H = 64 * TPW <= 256 units and one or two layers
// (GRU(H, return_sequences) -> GRU(H) -> Dense(1, sigmoid)), where the weights (786 KB per H=256
// matrix pair) no longer fit registers.
//
// One workgroup of WAVES waves (4; 8 is a build switch that measured slower) owns one tile of 16 streams
// for the whole window:
//   * wave w owns hidden units [H/WAVES * w, H/WAVES * (w+1)) of every gate: TPW output tiles of z, of r
//     and of the candidate.  Row 4 g + reg of tile tau is unit 16 tau + 4 reg + g, so a lane's four D
//     registers of a tile are k-slot g of four consecutive k-steps (rho = 4 tau + reg) of the next
//     contraction: a tile's new state goes to LDS as ONE ds_write_b128 per lane and comes back as the
//     B operands of four MFMAs with ONE ds_read_b128;
//   * the full hidden state (and r*h) of each layer lives in LDS as [rho / 4][lane][4] floats
//     (16 KB per layer and kind at H = 256), written by the owning wave, read by all four;
//   * weights are STREAMED from L2 every timestep as MFMA A operands, packed per wave and phase as
//     [k-group of 4][tile][lane] float4, double-buffered one k-group (32 MFMAs = ~1000 cycles) ahead;
//     2.4 MB of weights per timestep and workgroup at H = 256 x 2 layers = 32 B/clk/CU, half of what a
//     CU can pull from its XCD's L2, where the 2.4 MB stay resident;
//   * per layer and timestep: phase 1 (z, r: x W + h U), barrier, phase 2 (candidate: x W + (r*h) U,
//     state update), barrier.  2352 MFMAs per wave and timestep at H = 256 x 2: the matrix cores, not
//     the hand-offs, set the time.
#pragma once
#include "gru_device.h"

namespace pe {

// acc[tl] += sum over k-groups of W . B, B operands (float4 = four consecutive k-steps) from LDS.
// Weights arrive through a register ring, PE_WIDE_PF k-groups (each NT x 4 MFMAs = NT x 128 cycles of
// matrix work) ahead of their use; the B quad one k-group ahead.  n4 is a multiple of 4.  Measured at
// 256 x 2 layers, 4096 streams: distance 1 1.254 ms per launch, distance 2 1.305, distance 3 spills.
#ifndef PE_WIDE_PF
#define PE_WIDE_PF 1
#endif
template <int NT>
__device__ __forceinline__ void wide_accumulate(f32x4 (&acc)[NT], const float4* __restrict__ w, const float* B, const int n4) {
    constexpr int D = PE_WIDE_PF;
    float4 st[4][NT];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const int rd = d < n4 ? d : n4 - 1;
#pragma unroll
        for (int tl = 0; tl < NT; ++tl) st[d][tl] = w[(rd * NT + tl) * 64];
    }
    float4 b = *reinterpret_cast<const float4*>(B);
#pragma unroll 1
    for (int r4 = 0; r4 < n4; r4 += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rn = r4 + u + D < n4 ? r4 + u + D : n4 - 1;            // clamped prefetch, no branch
#pragma unroll
            for (int tl = 0; tl < NT; ++tl) st[(u + D) & 3][tl] = w[(rn * NT + tl) * 64];
            const int rb = r4 + u + 1 < n4 ? r4 + u + 1 : n4 - 1;
            const float4 bn = *reinterpret_cast<const float4*>(B + rb * 256);
            // the loads above must be ISSUED here: left alone, the scheduler sinks them to the end of the MFMA
            // block below (to reuse registers) and the next k-group then waits out the whole L2 latency
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tl = 0; tl < NT; ++tl) {
                acc[tl] = mfma(st[u][tl].x, b.x, acc[tl]);
                acc[tl] = mfma(st[u][tl].y, b.y, acc[tl]);
                acc[tl] = mfma(st[u][tl].z, b.z, acc[tl]);
                acc[tl] = mfma(st[u][tl].w, b.w, acc[tl]);
            }
            __builtin_amdgcn_sched_barrier(0);
            b = bn;
        }
    }
}

// same with the B operands in registers (layer 0: the 16-float feature row, one k-group)
template <int NT>
__device__ __forceinline__ void wide_accumulate_x(f32x4 (&acc)[NT], const float4* __restrict__ w, const f32x4& x) {
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) {
        const float4 a = w[tl * 64];
        acc[tl] = mfma(a.x, x[0], acc[tl]);
        acc[tl] = mfma(a.y, x[1], acc[tl]);
        acc[tl] = mfma(a.z, x[2], acc[tl]);
        acc[tl] = mfma(a.w, x[3], acc[tl]);
    }
}

// MODE as in gru_device.h (kFeats / kRing / kRows).  LDS: [layer][HB | RH][H/16][64 lanes][4] floats.
template <int TPW, int MODE, int WAVES>
__device__ __forceinline__ void gru_wide_tile(const WideArgs& wa, const int tile, const int wave, const int lane, float* lds) {
#pragma clang fp contract(off)      // every fusion in the gate arithmetic is spelled out: all kernel shapes round alike
    const GruArgs& a = wa.base;
    constexpr int H = 16 * TPW * WAVES, H16 = H / 16;
    const int g = lane >> 4, j = lane & 15;
    const long long stream = (long long)tile * kTileStreams + j;
    const bool valid = stream < a.n_streams;
    const int T = a.n_features;
    const int L = wa.n_layers;

    // ---- input addressing (k-step kk of the layer-0 input projection <-> feature 4 g + kk) ------------
    const float* xbase = nullptr;
    uint32_t first = 0;
    const uint32_t mask = (uint32_t)(a.ring_slots - 1);
    if (MODE == kRing) {
        const long long sid = gru_stream_of(a, stream, valid);      // the stream whose record and ring rows this lane reads
        const uint32_t ke = gru_window_end(a, sid);
        first = ke - (uint32_t)T;
        xbase = a.ring + gru_ring_cell(a, sid) * kRowFloats + 4 * g;
    } else if (MODE == kRows) {
        xbase = a.feats + ((size_t)(valid ? stream : 0) * a.row_stride) * kRowFloats + 4 * g;
    } else {
        xbase = a.feats + (size_t)(valid ? stream : 0) * T * a.n_in;
    }
    auto load_x = [&](int t) -> f32x4 {
        const int tc = t < T ? t : T - 1;
        if (MODE == kRing)
            return *reinterpret_cast<const f32x4*>(xbase + (size_t)((first + (uint32_t)tc) & mask) * kTileStreams * kRowFloats);
        if (MODE == kRows) return *reinterpret_cast<const f32x4*>(xbase + (size_t)tc * kRowFloats);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const float* p = xbase + (size_t)tc * a.n_in + 4 * g;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) if (valid && 4 * g + kk < a.n_in) v[kk] = p[kk];
        return v;
    };

    // ---- LDS state: h0 = 0 ------------------------------------------------------------------------------
    float* HB[2] = {lds, lds + 2 * H16 * 256};
    float* RH[2] = {lds + H16 * 256, lds + 3 * H16 * 256};
    for (int i = threadIdx.x; i < 4 * H16 * 256; i += 64 * WAVES) lds[i] = 0.f;
    float hown[2][TPW][4];
#pragma unroll
    for (int l = 0; l < 2; ++l)
#pragma unroll
        for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
            for (int q = 0; q < 4; ++q) hown[l][tp][q] = 0.f;
    __syncthreads();

    f32x4 x = load_x(0);
    for (int t = 0; t < T; ++t) {
        const f32x4 xn = load_x(t + 1);
#pragma unroll
        for (int l = 0; l < 2; ++l) {
            if (l >= L) break;
            const WideLayerArgs& W = wa.layer[l];
            const float* Xin = l == 0 ? nullptr : HB[l - 1] + lane * 4;
            // ---- phase 1: z and r of this wave's units ----------------------------------------------------
            f32x4 acc[2 * TPW];
#pragma unroll
            for (int tl = 0; tl < 2 * TPW; ++tl)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[tl][q] = W.b1[((wave * 2 * TPW + tl) * 4 + q) * 64 + lane];
            const float4* wx1 = W.wx1 + (size_t)wave * W.kx4 * 2 * TPW * 64 + lane;
            if (l == 0) wide_accumulate_x<2 * TPW>(acc, wx1, x);
            else wide_accumulate<2 * TPW>(acc, wx1, Xin, W.kx4);
            wide_accumulate<2 * TPW>(acc, W.wr1 + (size_t)wave * H16 * 2 * TPW * 64 + lane, HB[l] + lane * 4, H16);
            float z[TPW][4];
#pragma unroll
            for (int tp = 0; tp < TPW; ++tp) {
                float4 rh;
                z[tp][0] = hard_sigmoid(acc[tp][0]); z[tp][1] = hard_sigmoid(acc[tp][1]);
                z[tp][2] = hard_sigmoid(acc[tp][2]); z[tp][3] = hard_sigmoid(acc[tp][3]);
                rh.x = hard_sigmoid(acc[TPW + tp][0]) * hown[l][tp][0];
                rh.y = hard_sigmoid(acc[TPW + tp][1]) * hown[l][tp][1];
                rh.z = hard_sigmoid(acc[TPW + tp][2]) * hown[l][tp][2];
                rh.w = hard_sigmoid(acc[TPW + tp][3]) * hown[l][tp][3];
                *reinterpret_cast<float4*>(RH[l] + ((wave * TPW + tp) * 64 + lane) * 4) = rh;
            }
            __syncthreads();
            // ---- phase 2: candidate and state update ---------------------------------------------------------
            f32x4 acc2[TPW];
#pragma unroll
            for (int tl = 0; tl < TPW; ++tl)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc2[tl][q] = W.b2[((wave * TPW + tl) * 4 + q) * 64 + lane];
            const float4* wx2 = W.wx2 + (size_t)wave * W.kx4 * TPW * 64 + lane;
            if (l == 0) wide_accumulate_x<TPW>(acc2, wx2, x);
            else wide_accumulate<TPW>(acc2, wx2, Xin, W.kx4);
            wide_accumulate<TPW>(acc2, W.wr2 + (size_t)wave * H16 * TPW * 64 + lane, RH[l] + lane * 4, H16);
#pragma unroll
            for (int tp = 0; tp < TPW; ++tp) {
                float4 hn;
#pragma unroll
                for (int q = 0; q < 4; ++q) hown[l][tp][q] = gru_blend(z[tp][q], hown[l][tp][q], acc2[tp][q]);
                hn.x = hown[l][tp][0]; hn.y = hown[l][tp][1]; hn.z = hown[l][tp][2]; hn.w = hown[l][tp][3];
                *reinterpret_cast<float4*>(HB[l] + ((wave * TPW + tp) * 64 + lane) * 4) = hn;
            }
            __syncthreads();
        }
        x = xn;
    }

    // ---- Dense(1) + sigmoid over the last layer's state: lane -> lane groups -> waves -------------------------
    float part = 0.f;
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float hv = (L == 2) ? hown[1][tp][q] : hown[0][tp][q];
            part = fmaf(hv, wa.wd[((wave * TPW + tp) * 4 + q) * 64 + lane], part);
        }
    part += __shfl_xor(part, 16);
    part += __shfl_xor(part, 32);
    if (g == 0) lds[wave * 16 + j] = part;
    __syncthreads();
    if (wave == 0 && g == 0 && valid) {
        float logit = lds[j];
#pragma unroll
        for (int wv = 1; wv < WAVES; ++wv) logit += lds[wv * 16 + j];
        a.out[stream] = 1.0f / (1.0f + expf(-(logit + a.dense_bias)));
    }
}

}  // namespace pe
