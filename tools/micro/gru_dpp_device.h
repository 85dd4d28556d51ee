// Latency-shaped GRU + Dense forward for FEW streams (<= one 16-stream tile per compute unit), gfx950.
//
// Same network and recurrence as gru_device.h (model.py:76-82; Keras GRUCell, reset_after = False, hard_sigmoid,
// linear candidate).  At BASELINE configs[1] (4096 streams = 256 tiles for 1024 SIMDs) the window is a chain of
// 29 timesteps x 2 dependent phases, and what bounds an update is the length of that chain on ONE SIMD, not the
// machine's matrix throughput: the 16x16x4 MFMA formulation pays, per phase, five dependent MFMAs (200 cycles)
// plus either a cross-wave LDS hand-off with a barrier (four waves per tile) or the issue time of every output
// tile on one matrix pipe (one wave per tile) -- 1300-1850 cycles per timestep measured.  Here a phase is a run
// of INDEPENDENT fused multiply-adds with no hand-off at all:
//   * a 16-lane row owns one stream, a wave four streams, a 256-thread workgroup one tile: 1024 waves at 4096 streams,
//     one per SIMD;
//   * lane i of a row owns hidden unit i (its z, r, candidate and h) and, replicated in every quad, unit 16 + (i & 3);
//     the H <= 20 values a gate needs from the other lanes arrive INSIDE the multiply-add: v_fmac_f32_dpp with
//     row_ror:n reads lane (i -+ n) of the row, quad_perm:[q,q,q,q] reads lane q of the quad -- sixteen rotations
//     of the primary units plus four broadcasts of the replicated ones = the whole recurrent matvec, no LDS, no
//     shuffle instruction, no wait;
//   * each lane keeps the 120 recurrent weights it needs for its two units in registers, fetched once per launch
//     in the order the rotations deliver the sources (the lane -> source map is MEASURED with the same DPP controls
//     at kernel start, so no assumption about the rotation's direction is baked in);
//   * the input projection x.W + b of every frame comes from the projection rows the MFCC stage stored (proj_ring):
//     six floats per lane and timestep, requested two timesteps ahead.
// Float32 fused multiply-adds in a fixed order: deterministic, identical for a stream wherever it sits.
// Requires 17 <= units <= 20 (R = 5 slot layout of the projection rows) and a.proj_ring.
#pragma once
#include "../../mycroft_precise_amd/csrc/gru_device.h"

namespace pe {

// v_mov_b32_dpp with a compile-time control; the compiler folds it into the consuming v_fmac_f32 (GCNDPPCombine)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }

constexpr int kDppRowRor0 = 0x120;                       // row_ror:n = 0x120 + n (n = 1..15)
constexpr int dpp_quad(int q) { return q | (q << 2) | (q << 4) | (q << 6); }     // quad_perm:[q,q,q,q]

// acc += dpp(src) * w as ONE instruction: the DPP control rides on the multiply-add's first operand, so the 20
// sources of a gate cost no instruction of their own.  (Written as inline assembly: with several multiply-adds
// sharing one rotated source the compiler keeps a separate v_mov_b32_dpp per rotation.)  A VALU write of `src`
// must be two wait states old before a DPP read of it: PE_DPP_SETTLE() at the top of every matvec.
#define PE_DPP_SETTLE() asm volatile("s_nop 1" ::: "memory")
#define PE_FMAC_ROR(ACC, SRC, W, N) asm("v_fmac_f32_dpp %0, %1, %2 row_ror:" #N " row_mask:0xf bank_mask:0xf" : "+v"(ACC) : "v"(SRC), "v"(W))
#define PE_FMAC_QUAD(ACC, SRC, W, Q) asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[" #Q "," #Q "," #Q "," #Q "] row_mask:0xf bank_mask:0xf" : "+v"(ACC) : "v"(SRC), "v"(W))

// c += sum over the 20 source units: the lane's own primary source, fifteen rotations of `a` (the other primary
// units of the row) and four quad broadcasts of `b` (units 16..19, replicated per quad); w[n] is the weight of the
// source the n-th step delivers to this lane
// two partial sums per output row (even / odd steps): eight independent accumulator chains in phase 1 and four in
// phase 2, so that a dependent multiply-add never has to wait for its predecessor's result
#define PE_DPP_ROW(E, O, A, B, W)                                                                                  \
    E = fmaf(A, W[0], E);           PE_FMAC_ROR(O, A, W[1], 1);                                                     \
    PE_FMAC_ROR(E, A, W[2], 2);     PE_FMAC_ROR(O, A, W[3], 3);     PE_FMAC_ROR(E, A, W[4], 4);   PE_FMAC_ROR(O, A, W[5], 5);   \
    PE_FMAC_ROR(E, A, W[6], 6);     PE_FMAC_ROR(O, A, W[7], 7);     PE_FMAC_ROR(E, A, W[8], 8);   PE_FMAC_ROR(O, A, W[9], 9);   \
    PE_FMAC_ROR(E, A, W[10], 10);   PE_FMAC_ROR(O, A, W[11], 11);   PE_FMAC_ROR(E, A, W[12], 12); PE_FMAC_ROR(O, A, W[13], 13); \
    PE_FMAC_ROR(E, A, W[14], 14);   PE_FMAC_ROR(O, A, W[15], 15);                                                   \
    PE_FMAC_QUAD(E, B, W[16], 0);   PE_FMAC_QUAD(O, B, W[17], 1);   PE_FMAC_QUAD(E, B, W[18], 2); PE_FMAC_QUAD(O, B, W[19], 3);

__device__ __forceinline__ void dpp_matvec4(const float a, const float b, const float (&w0)[20], const float (&w1)[20],
                                            const float (&w2)[20], const float (&w3)[20], float& c0, float& c1, float& c2, float& c3) {
    // four output rows at once (z and r of the lane's two units)
    float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
    PE_DPP_SETTLE();
    PE_DPP_ROW(c0, o0, a, b, w0)
    PE_DPP_ROW(c1, o1, a, b, w1)
    PE_DPP_ROW(c2, o2, a, b, w2)
    PE_DPP_ROW(c3, o3, a, b, w3)
    c0 += o0; c1 += o1; c2 += o2; c3 += o3;
}

__device__ __forceinline__ void dpp_matvec2(const float a, const float b, const float (&w0)[20], const float (&w1)[20], float& c0, float& c1) {
    float o0 = 0.f, o1 = 0.f;
    PE_DPP_SETTLE();
    PE_DPP_ROW(c0, o0, a, b, w0)
    PE_DPP_ROW(c1, o1, a, b, w1)
    c0 += o0; c1 += o1;
}

// which unit each of the 20 steps delivers to this lane: the same DPP controls applied to the lane's own unit ids
__device__ __forceinline__ void dpp_source_units(const int unit_a, const int unit_b, int (&src)[20]) {
    src[0] = unit_a;
#define PE_SRC(N) src[N] = dpp_i<kDppRowRor0 + N>(unit_a);
    PE_SRC(1) PE_SRC(2) PE_SRC(3) PE_SRC(4) PE_SRC(5) PE_SRC(6) PE_SRC(7) PE_SRC(8)
    PE_SRC(9) PE_SRC(10) PE_SRC(11) PE_SRC(12) PE_SRC(13) PE_SRC(14) PE_SRC(15)
#undef PE_SRC
    src[16] = dpp_i<dpp_quad(0)>(unit_b); src[17] = dpp_i<dpp_quad(1)>(unit_b);
    src[18] = dpp_i<dpp_quad(2)>(unit_b); src[19] = dpp_i<dpp_quad(3)>(unit_b);
}

// One workgroup (256 threads) = one tile of 16 streams; wave w serves streams 4w .. 4w+3 of the tile.
// a.rk: the Keras recurrent kernel [H][3H] as uploaded (gate order z | r | h); a.wd_plain: dense kernel [H].
__device__ __forceinline__ void gru_tile_dpp(const GruArgs& a, const int tile, const int wave, const int lane) {
#pragma clang fp contract(off)      // every fusion in the gate arithmetic is spelled out: all kernel shapes round alike
    const int H = a.units, T = a.n_features;
    const int row = lane >> 4, i = lane & 15;
    const int j = 4 * wave + row;                              // stream within the tile
    const long long stream = (long long)tile * kTileStreams + j;
    const bool valid = stream < a.n_streams;
    const int uA = i, uB = 16 + (i & 3);
    const bool hasB = uB < H;

    // ---- resident weights: w[gate][unit A | B][step] in the order the DPP steps deliver the sources -----------
    int src[20];
    dpp_source_units(uA, uB, src);
    float wzA[20], wrA[20], wcA[20], wzB[20], wrB[20], wcB[20];
    const int H3 = 3 * H;
    const int colB = hasB ? uB : 0;
    // (unconditional loads from clamped rows, zeroed afterwards: 120 independent requests, one wait)
#pragma unroll
    for (int n = 0; n < 20; ++n) {
        const int k = src[n] < H ? src[n] : 0;
        const float* rk = a.rk + (size_t)k * H3;
        wzA[n] = rk[uA]; wrA[n] = rk[H + uA]; wcA[n] = rk[2 * H + uA];
        wzB[n] = rk[colB]; wrB[n] = rk[H + colB]; wcB[n] = rk[2 * H + colB];
    }
#pragma unroll
    for (int n = 0; n < 20; ++n) {
        const bool ok = src[n] < H;
        if (!ok) { wzA[n] = 0.f; wrA[n] = 0.f; wcA[n] = 0.f; }
        if (!ok || !hasB) { wzB[n] = 0.f; wrB[n] = 0.f; wcB[n] = 0.f; }
    }
    const float wdA = a.wd_plain[uA], wdB = hasB && (i < 4) ? a.wd_plain[uB] : 0.f;     // units 16..19 counted once

    // ---- window position (as the other ring-fed kernels) -----------------------------------------------------
    const long long sc = valid ? stream : 0;
    uint32_t ke = a.st_ke[sc];
    if (a.predict_ke) {
        const int q = a.st_q[sc];
        const uint32_t kc = a.st_kc[sc];
        const int avail = q + a.chunk;
        const int nnew = avail >= a.frame_len ? 1 + (avail - a.frame_len) / a.hop : 0;
        const int qn = avail - nnew * a.hop;
        const int m = qn + a.hop * (int)(kc + (uint32_t)nnew - ke);
        if (m >= a.window) ke += 1u + (uint32_t)((m - a.window) / a.hop);
    }
    const uint32_t first = ke - (uint32_t)T;
    const uint32_t mask = (uint32_t)(a.ring_slots - 1);
    // projection of (gate, unit): slot = gate * 5 + rho, unit = 4 rho + g, output tile slot / 4, accumulator slot % 4;
    // the (tile, slot-of-the-ring) block is laid out [output tile][stream][g][q] (gru_device.h: proj_base)
    const int tile_c = valid ? tile : 0, j_c = valid ? j : 0;
    const float* prow = a.proj_ring + (size_t)tile_c * a.ring_slots * kTileStreams * kProjRow;
    auto elem = [&](int gate, int u) -> int {
        const int slot = gate * 5 + (u >> 2), g = u & 3;
        return (slot >> 2) * kProjTileStride + (j_c * 4 + g) * 4 + (slot & 3);
    };
    const int ezA = elem(0, uA), erA = elem(1, uA), ecA = elem(2, uA), ezB = elem(0, uB), erB = elem(1, uB), ecB = elem(2, uB);
    struct Proj { float zA, rA, cA, zB, rB, cB; };
    auto load_p = [&](int t) -> Proj {
        const int tc = t < T ? t : T - 1;
        const float* p = prow + (size_t)((first + (uint32_t)tc) & mask) * kTileStreams * kProjRow;
        Proj r;
        r.zA = p[ezA]; r.rA = p[erA]; r.cA = p[ecA];
        r.zB = p[ezB]; r.rB = p[erB]; r.cB = p[ecB];
        return r;
    };

    float hA = 0.f, hB = 0.f;
    Proj p0 = load_p(0), p1 = load_p(1), p2 = load_p(2), p3 = load_p(3);
    for (int t = 0; t < T; ++t) {
        const Proj p4 = load_p(t + 4);          // four timesteps ahead (the rows come from L2 / MALL, one wave per SIMD)
        float zA = p0.zA, rA = p0.rA, zB = p0.zB, rB = p0.rB;
        dpp_matvec4(hA, hB, wzA, wrA, wzB, wrB, zA, rA, zB, rB);
        zA = hard_sigmoid(zA); zB = hard_sigmoid(zB);
        const float rhA = hard_sigmoid(rA) * hA, rhB = hard_sigmoid(rB) * hB;
        float cA = p0.cA, cB = p0.cB;
        dpp_matvec2(rhA, rhB, wcA, wcB, cA, cB);
        hA = gru_blend(zA, hA, cA);
        hB = hasB ? gru_blend(zB, hB, cB) : 0.f;
        p0 = p1; p1 = p2; p2 = p3; p3 = p4;
    }

    // Dense(1) + sigmoid over the row
    float part = fmaf(hA, wdA, hB * wdB);
    part += __shfl_xor(part, 1, 16);
    part += __shfl_xor(part, 2, 16);
    part += __shfl_xor(part, 4, 16);
    part += __shfl_xor(part, 8, 16);
    if (valid && i == 0) a.out[stream] = 1.0f / (1.0f + expf(-(part + a.dense_bias)));
}

}  // namespace pe
