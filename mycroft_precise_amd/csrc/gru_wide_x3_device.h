// Wide / stacked GRU + Dense forward with the float32 products formed on the XDL (bf16) matrix pipe -- BASELINE.json
// configs[3] ("wide GRU, 256 hidden units, 2 stacked layers ... MFMA-bound regime"), gfx950.  pe_set_gru_tiling(e, 2).
//
// Same network, same recurrence and the same streamed-weight structure as gru_wide_device.h (model.py:76-82 by extension:
// one 256-thread workgroup owns a tile of 16 streams for the whole window, wave w owns the hidden units [H/4 w, H/4 (w+1)) of
// every gate, the state lives in LDS, the weights are streamed from L2 every timestep), and the arithmetic of gru_x3_device.h:
// every operand as three bf16 pieces that add up to the float32 value exactly, six piece products per multiplication,
// float32 accumulate / gates / state -- the float32 tolerance at the float32 kernels' distance to a float64 evaluation.
//
// What is different from the small x3 kernel, and why:
//   * THE WEIGHTS STAY FLOAT32 IN MEMORY AND ARE SPLIT IN REGISTERS, ON THE MATRIX PIPE, EVERY TIMESTEP.  A wide network is
//     bound by what a compute unit can pull from its XCD's L2 (measured, tools/micro/l2_stream.hip: 50 B/clk/CU while the
//     set stays below ~3.6 MB): pre-split weights are 6 bytes per weight (3.5 MB per timestep at 256 x 2: 29.5 us per timestep
//     against 11.8 us of matrix work), float32 weights 4 bytes (2.4 MB: 19.7 us).  A lane's float4 of weights is, element for
//     element, a tile in accumulator layout (lane (i, gk) holds W[16 kappa + 4 gk + e][row i], e = 0..3: column i, rows
//     4 gk + e), so x3_split_tile splits it with two MFMAs against -I and three packed conversions: 2 + 3 = 5 bf16 MFMAs of
//     4 passes (80 XDL cycles) per (output tile, 16 source units) against 4 f32-input MFMAs of 8 passes (128 cycles of the
//     whole SIMD) in gru_wide_device.h -- matrix time and L2 time per timestep then balance at ~20 us each.
//   * Row i of output tile tau is unit 16 tau + i, and lane group gk supplies the source units 16 kappa + 4 gk .. + 3 of
//     k-group kappa in k-slots 8 gk + e (first piece) and 8 gk + 4 + e (second piece): a lane's four outputs of tile tau ARE
//     its four B-operand values of k-group kappa = tau, so the owning wave splits them once ([hi | lo] 16 bytes + [mid] 8
//     bytes per lane and k-group, 24 KB per state vector at H = 256) and every wave reads them back with one ds_read_b128 +
//     one ds_read_b64 per k-group:
//         m0:  A = [W_hi | W_mid]   B = [h_hi  | h_hi ]
//         m1:  A = [W_hi | W_mid]   B = [h_mid | h_mid]
//         m2:  A = [W_lo | W_hi ]   B = [h_hi  | h_lo ]
//   * LDS: h of layer 0, h of layer 1 and ONE r.h buffer (r.h of layer l is dead when layer l + 1 writes its own): 72 KB.
#pragma once
#include "gru_wide_device.h"
#include "gru_x3_device.h"

namespace pe {

// B operands of one k-group as they sit in LDS
struct WideX3B { uint4 hl; uint2 md; };
__device__ __forceinline__ WideX3B wide_x3_read_b(const uint4* HL, const uint2* MD, const int kappa) {
    WideX3B b;
    b.hl = HL[kappa * 64];
    b.md = MD[kappa * 64];
    return b;
}

// acc[tl] += W(tile tl, k-group) . B for ONE k-group, nothing overlapped (the layer-0 input: B operands in registers)
template <int NT>
__device__ __forceinline__ void wide_x3_kgroup(f32x4 (&acc)[NT], const float4 (&w)[NT], const WideX3B& b, const X3Ident& id) {
    const bf16x8 b_hh = __builtin_bit_cast(bf16x8, uint4{b.hl.x, b.hl.y, b.hl.x, b.hl.y});
    const bf16x8 b_mm = __builtin_bit_cast(bf16x8, uint4{b.md.x, b.md.y, b.md.x, b.md.y});
    const bf16x8 b_hl = __builtin_bit_cast(bf16x8, b.hl);
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) {
        const f32x4 v = {w[tl].x, w[tl].y, w[tl].z, w[tl].w};
        uint4 keep = {0u, 0u, 0u, 0u};
        const X3Ops o = x3_split_tile(v, keep, id);                   // b0 = [W_hi | W_mid], b2 = [W_lo | W_hi]
        acc[tl] = mfma_bf16(__builtin_bit_cast(bf16x8, o.b0), b_hh, acc[tl]);
        acc[tl] = mfma_bf16(__builtin_bit_cast(bf16x8, o.b0), b_mm, acc[tl]);
        acc[tl] = mfma_bf16(__builtin_bit_cast(bf16x8, o.b2), b_hl, acc[tl]);
    }
}

// acc[tl] += sum over the k-groups of TWO segments (input part: nA k-groups of wA against the state vector HLa / MDa, then the
// recurrent part: nB k-groups of wB against HLb / MDb; nA may be 0) of W . B -- the hot loop of the kernel, written as a
// software pipeline with EVERY instruction pinned where it stands (one __builtin_amdgcn_sched_barrier per issue group):
// while the three product MFMAs of k-group k run for the NT tiles, the float32 weights of k-group k + 1 are split
// (conversions riding behind the MFMAs, the two remainder MFMAs of a tile NT issue slots apart from the conversions that feed
// and consume them), the weights of k-groups k + 2, k + 3 are in flight from L2 and the B operands of k + 1 from LDS.
// Left to the scheduler, the first version of this loop issued a tile's five MFMAs and six conversions as one dependent
// chain with s_nop padding: 188 cycles per (tile, k-group) against 80 of matrix work (1.34 ms per launch at 256 x 2).
//   iteration k:   G1  NT x { m0(k, t);                 hi(k+1, t) = cvt(raw(k+1, t)) }
//                  G2  NT x { r1(k+1, t) = raw - hi;    m1(k, t);   lo(k, t) = cvt(r2(k, t)), [lo | hi](k, t) }
//                  G3  NT x { m2(k, t);                 mid(k+1, t) = cvt(r1(k+1, t)) }
//                  G4  NT x { r2(k+1, t) = r1 - mid }
// 5 NT MFMAs and 8 NT vector instructions per iteration.  Registers: four raw sets (k + 1 being split, k + 2 and k + 3 landing,
// k's second remainder until G2), two piece sets (k in the products, k + 1 being made).
// (wA / wB are WAVE-UNIFORM pointers and the lane is added at the load: the loads then take a scalar base and one 32-bit lane
//  offset, and nothing about a segment's addresses lives in vector registers across the time loop)
template <int NT>
__device__ __forceinline__ void wide_x3_accumulate(f32x4 (&acc)[NT], const float4* __restrict__ wA, const int nA, const uint4* HLa, const uint2* MDa,
                                                   const float4* __restrict__ wB, const int nB, const uint4* HLb, const uint2* MDb, const X3Ident& id, const int lane) {
    const int n_k = nA + nB;                                   // a multiple of 4
    auto wptr = [&](int k) -> const float4* { k = k < n_k ? k : n_k - 1; return k < nA ? wA + (size_t)k * NT * 64 : wB + (size_t)(k - nA) * NT * 64; };
    auto bread = [&](int k) -> WideX3B { k = k < n_k ? k : n_k - 1; return k < nA ? wide_x3_read_b(HLa, MDa, k) : wide_x3_read_b(HLb, MDb, k - nA); };
    f32x4 raw[4][NT];               // set (k & 3): float32 weights of k-group k -> first remainder -> second remainder
    uint4 b0[2][NT], b2[2][NT];     // set (k & 1): [W_hi | W_mid], [W_lo | W_hi]
    auto request = [&](const int set, const int k) {
        const float4* p = wptr(k);
#pragma unroll
        for (int t = 0; t < NT; ++t) { const float4 v = p[t * 64 + lane]; raw[set][t] = f32x4{v.x, v.y, v.z, v.w}; }
    };
    // prologue: k-groups 0 .. 2 requested, k-group 0 split on the spot
    request(0, 0); request(1, 1); request(2, 2);
    WideX3B bc = bread(0);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        b0[0][t] = uint4{0u, 0u, 0u, 0u};
        b0[1][t] = uint4{0u, 0u, 0u, 0u};
        b0[0][t].x = x3_pk(raw[0][t][0], raw[0][t][1]); b0[0][t].y = x3_pk(raw[0][t][2], raw[0][t][3]);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) raw[0][t] = mfma_bf16(id.lo_half, __builtin_bit_cast(bf16x8, b0[0][t]), raw[0][t]);
#pragma unroll
    for (int t = 0; t < NT; ++t) { b0[0][t].z = x3_pk(raw[0][t][0], raw[0][t][1]); b0[0][t].w = x3_pk(raw[0][t][2], raw[0][t][3]); }
#pragma unroll
    for (int t = 0; t < NT; ++t) raw[0][t] = mfma_bf16(id.hi_half, __builtin_bit_cast(bf16x8, b0[0][t]), raw[0][t]);
#pragma unroll 1
    for (int k4 = 0; k4 < n_k; k4 += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k4 + u;
            const int c = u & 1, n = c ^ 1;                    // piece sets of k and k + 1
            const int rc = u, rn = (u + 1) & 3;                // raw sets of k and k + 1
            request((u + 3) & 3, k + 3);
            const WideX3B bnext = bread(k + 1);
            const bf16x8 b_hh = __builtin_bit_cast(bf16x8, uint4{bc.hl.x, bc.hl.y, bc.hl.x, bc.hl.y});
            const bf16x8 b_mm = __builtin_bit_cast(bf16x8, uint4{bc.md.x, bc.md.y, bc.md.x, bc.md.y});
            const bf16x8 b_hl = __builtin_bit_cast(bf16x8, bc.hl);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NT; ++t) {                     // G1
                acc[t] = mfma_bf16(__builtin_bit_cast(bf16x8, b0[c][t]), b_hh, acc[t]);
                __builtin_amdgcn_sched_barrier(0);
                b0[n][t].x = x3_pk(raw[rn][t][0], raw[rn][t][1]);
                b0[n][t].y = x3_pk(raw[rn][t][2], raw[rn][t][3]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {                     // G2
                raw[rn][t] = mfma_bf16(id.lo_half, __builtin_bit_cast(bf16x8, b0[n][t]), raw[rn][t]);
                __builtin_amdgcn_sched_barrier(0);
                b2[c][t].x = x3_pk(raw[rc][t][0], raw[rc][t][1]);
                b2[c][t].y = x3_pk(raw[rc][t][2], raw[rc][t][3]);
                __builtin_amdgcn_sched_barrier(0);
                acc[t] = mfma_bf16(__builtin_bit_cast(bf16x8, b0[c][t]), b_mm, acc[t]);
                __builtin_amdgcn_sched_barrier(0);
                b2[c][t].z = b0[c][t].x;
                b2[c][t].w = b0[c][t].y;
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {                     // G3
                acc[t] = mfma_bf16(__builtin_bit_cast(bf16x8, b2[c][t]), b_hl, acc[t]);
                __builtin_amdgcn_sched_barrier(0);
                b0[n][t].z = x3_pk(raw[rn][t][0], raw[rn][t][1]);
                b0[n][t].w = x3_pk(raw[rn][t][2], raw[rn][t][3]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {                     // G4
                raw[rn][t] = mfma_bf16(id.hi_half, __builtin_bit_cast(bf16x8, b0[n][t]), raw[rn][t]);
                __builtin_amdgcn_sched_barrier(0);
            }
            bc = bnext;
        }
    }
}

// split the four values a lane holds of one output tile and publish them as the B operands of k-group `kappa`
__device__ __forceinline__ void wide_x3_publish(const f32x4& v, uint4* HL, uint2* MD, const int kappa, uint4& keep, const X3Ident& id) {
    const X3Ops o = x3_split_tile(v, keep, id);                       // b0 = [hi | mid], b2 = [lo | hi]
    HL[kappa * 64] = uint4{o.b0.x, o.b0.y, o.b2.x, o.b2.y};           // [hi | lo]
    MD[kappa * 64] = uint2{o.b0.z, o.b0.w};
}

// KS = 2: EIGHT waves per workgroup, two per SIMD -- waves w and w + 4 own the same output tiles and each walks HALF of the
// k-groups of every contraction (partial sums meet in LDS): on gfx950 the vector instructions of a wave do not overlap its own
// MFMAs (measured here too: 7.2 k vector instructions x 4 cycles + 3.06 k MFMAs x 16 cycles = 78 k cycles per timestep predicted,
// 81 k measured with the weight loads ablated), but they do run under the MFMAs of ANOTHER wave of the SIMD.
// LDS: three state vectors (h layer 0, h layer 1, r.h), each [H/16 k-groups][64 lanes] uint4 followed by [H/16][64] uint2;
// then, [wave][tile][64 lanes] float4 each: PS (partial sums of the second half: z and r tiles), ZS (the lead wave's own z
// partial sum, then z itself until the blend), HO (the lead wave's float32 state of both layers).
template <int TPW> constexpr size_t wide_x3_lds_bytes() { return (size_t)3 * (4 * TPW) * 64 * 24 + (size_t)4 * (2 * TPW + TPW + 2 * TPW) * 64 * 16; }

template <int TPW, int MODE, int KS>
__device__ __forceinline__ void gru_wide_x3_tile(const WideArgs& wa, const int tile, const int wave_in_wg, const int lane, unsigned char* lds) {
#pragma clang fp contract(off)      // every fusion in the gate arithmetic is spelled out: all kernel shapes round alike
    const GruArgs& a = wa.base;
    constexpr int WAVES = 4, H = 16 * TPW * WAVES, H16 = H / 16;
    constexpr int kVecBytes = H16 * 64 * 24;
    static_assert(KS == 1 || (KS == 2 && H16 % 8 == 0), "the k-split halves must be multiples of four k-groups");
    const int wave = wave_in_wg & 3, half = KS == 2 ? wave_in_wg >> 2 : 0;
    const bool lead = half == 0;            // the half that owns biases, gates, state and output
    const int g = lane >> 4, j = lane & 15;
    const long long stream = (long long)tile * kTileStreams + j;
    const bool valid = stream < a.n_streams;
    const int T = a.n_features;
    const int L = wa.n_layers;
    const X3Ident ident = x3_identity(lane);

    // ---- input addressing (lane group g supplies features 4 g .. 4 g + 3 of the layer-0 input) -----------------------
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

    // ---- LDS ---------------------------------------------------------------------------------------------------------------
    uint4* const HL0 = reinterpret_cast<uint4*>(lds) + lane;                           // state vector v: HL0 + v * (kVecBytes / 16)
    uint2* const MD0 = reinterpret_cast<uint2*>(lds + H16 * 64 * 16) + lane;           //                 MD0 + v * (kVecBytes / 8)
    auto HLv = [&](int v) -> uint4* { return HL0 + (size_t)v * (kVecBytes / 16); };
    auto MDv = [&](int v) -> uint2* { return MD0 + (size_t)v * (kVecBytes / 8); };
    f32x4* const PS = reinterpret_cast<f32x4*>(lds + 3 * kVecBytes) + (size_t)wave * 2 * TPW * 64 + lane;
    f32x4* const ZS = reinterpret_cast<f32x4*>(lds + 3 * kVecBytes + (size_t)WAVES * 2 * TPW * 64 * 16) + (size_t)wave * TPW * 64 + lane;
    f32x4* const HO = reinterpret_cast<f32x4*>(lds + 3 * kVecBytes + (size_t)WAVES * 3 * TPW * 64 * 16) + (size_t)wave * 2 * TPW * 64 + lane;
    for (int i = threadIdx.x; i < 3 * kVecBytes / 4; i += 64 * WAVES * KS) reinterpret_cast<uint32_t*>(lds)[i] = 0u;      // h0 = 0: all pieces zero
    if (lead) {
#pragma unroll
        for (int i = 0; i < 2 * TPW; ++i) HO[i * 64] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    uint4 keep_s = {0u, 0u, 0u, 0u}, keep_x = {0u, 0u, 0u, 0u};
    __syncthreads();

    constexpr int HK = H16 / KS;
    const int ko = half * HK;
    // one contraction: nA k-groups of wA against state vector svA, then nB k-groups of wB (from its k-group kB on) against svB
    auto contract = [&](f32x4 (&acc)[TPW], const float4* wA, const int nA, const int svA, const float4* wB, const int nB, const int svB, const int kB) {
        wide_x3_accumulate<TPW>(acc, wA, nA, HLv(svA), MDv(svA), wB + (size_t)kB * TPW * 64, nB, HLv(svB) + kB * 64, MDv(svB) + kB * 64, ident, lane);
    };
    auto contract_x = [&](f32x4 (&acc)[TPW], const float4* wx, const WideX3B& xb) {          // the layer-0 input: one k-group, B operands in registers
        float4 w0[TPW];
#pragma unroll
        for (int tl = 0; tl < TPW; ++tl) w0[tl] = wx[tl * 64 + lane];
        wide_x3_kgroup<TPW>(acc, w0, xb, ident);
    };
    f32x4 x = load_x(0);
    for (int t = 0; t < T; ++t) {
        const f32x4 xn = load_x(t + 1);
        // the layer-0 input as B operands (the lead waves split the same four features of their lanes: 2 MFMAs)
        WideX3B xb;
        {
            const X3Ops o = x3_split_tile(x, keep_x, ident);
            xb.hl = uint4{o.b0.x, o.b0.y, o.b2.x, o.b2.y};
            xb.md = uint2{o.b0.z, o.b0.w};
        }
#pragma unroll 1
        for (int l = 0; l < L; ++l) {
            const WideLayerArgs& W = wa.layer[l];
            const int kin = W.kx4;                               // k-groups of the input contraction: 1 (layer 0: the feature row) or H16
            // weight streams, gate-major: [gate][wave][k-group][tile][lane] float4
            const size_t gx = (size_t)WAVES * kin * TPW * 64, gr = (size_t)WAVES * H16 * TPW * 64;
            f32x4 acc[TPW];
            // ---- phase 1: z, then r of this wave's units (TPW output tiles each) ------------------------------------------
#pragma unroll 1
            for (int gate = 0; gate < 2; ++gate) {
#pragma unroll
                for (int tl = 0; tl < TPW; ++tl)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[tl][q] = lead ? W.b1[((wave * 2 * TPW + gate * TPW + tl) * 4 + q) * 64 + lane] : 0.f;
                const size_t ox = gate * gx + (size_t)wave * kin * TPW * 64, orr = gate * gr + (size_t)wave * H16 * TPW * 64;
                const float4* const wx = W.wx1 + ox;
                const float4* const wr = W.wr1 + orr;
                if (l == 0) {
                    if (lead) contract_x(acc, wx, xb);
                    contract(acc, wr, 0, 0, wr, HK, 0, ko);
                } else if (KS == 1) {
                    contract(acc, wx, kin, 0, wr, H16, 1, 0);
                } else {            // the lead half walks the input part (state of layer 0), the other half the recurrent part
                    if (lead) contract(acc, wx, 0, 0, wx, H16, 0, 0);
                    else contract(acc, wr, 0, 1, wr, H16, 1, 0);
                }
                if (!lead) {
#pragma unroll
                    for (int tl = 0; tl < TPW; ++tl) PS[(gate * TPW + tl) * 64] = acc[tl];
                } else if (gate == 0) {
#pragma unroll
                    for (int tl = 0; tl < TPW; ++tl) ZS[tl * 64] = acc[tl];            // (the r tiles stay in registers)
                }
            }
            if (KS == 2) __syncthreads();
            if (lead) {
#pragma unroll
                for (int tp = 0; tp < TPW; ++tp) {
                    f32x4 zp = ZS[tp * 64], rp = acc[tp];
                    if (KS == 2) { zp += PS[tp * 64]; rp += PS[(TPW + tp) * 64]; }
                    const f32x4 ho = HO[(l * TPW + tp) * 64];
                    f32x4 z, rh;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        z[q] = hard_sigmoid(zp[q]);
                        rh[q] = hard_sigmoid(rp[q]) * ho[q];
                    }
                    ZS[tp * 64] = z;
                    wide_x3_publish(rh, HLv(2), MDv(2), wave * TPW + tp, keep_s, ident);
                }
            }
            __syncthreads();
            // ---- phase 2: candidate and state update ---------------------------------------------------------------------
#pragma unroll
            for (int tl = 0; tl < TPW; ++tl)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[tl][q] = lead ? W.b2[((wave * TPW + tl) * 4 + q) * 64 + lane] : 0.f;
            {
                const size_t ox = (size_t)wave * kin * TPW * 64, orr = (size_t)wave * H16 * TPW * 64;
                const float4* const wx = W.wx2 + ox;
                const float4* const wr = W.wr2 + orr;
                if (l == 0) {
                    if (lead) contract_x(acc, wx, xb);
                    contract(acc, wr, 0, 2, wr, HK, 2, ko);
                } else if (KS == 1) {
                    contract(acc, wx, kin, 0, wr, H16, 2, 0);
                } else {
                    if (lead) contract(acc, wx, 0, 0, wx, H16, 0, 0);
                    else contract(acc, wr, 0, 2, wr, H16, 2, 0);
                }
            }
            if (KS == 2) {
                if (!lead) {
#pragma unroll
                    for (int tl = 0; tl < TPW; ++tl) PS[tl * 64] = acc[tl];
                }
                __syncthreads();
            }
            if (lead) {
#pragma unroll
                for (int tp = 0; tp < TPW; ++tp) {
                    f32x4 cp = acc[tp];
                    if (KS == 2) cp += PS[tp * 64];
                    const f32x4 ho = HO[(l * TPW + tp) * 64], z = ZS[tp * 64];
                    f32x4 hn;
#pragma unroll
                    for (int q = 0; q < 4; ++q) hn[q] = gru_blend(z[q], ho[q], cp[q]);
                    HO[(l * TPW + tp) * 64] = hn;
                    wide_x3_publish(hn, HLv(l), MDv(l), wave * TPW + tp, keep_s, ident);
                }
            }
            __syncthreads();
        }
        x = xn;
    }

    // ---- Dense(1) + sigmoid over the last layer's state: lane -> lane groups -> waves -----------------------------------
    float part = 0.f;
    if (lead) {
#pragma unroll
        for (int tp = 0; tp < TPW; ++tp) {
            const f32x4 hv = HO[((L == 2 ? 1 : 0) * TPW + tp) * 64];
#pragma unroll
            for (int q = 0; q < 4; ++q) part = fmaf(hv[q], wa.wd[((wave * TPW + tp) * 4 + q) * 64 + lane], part);
        }
    }
    part += __shfl_xor(part, 16);
    part += __shfl_xor(part, 32);
    float* red = reinterpret_cast<float*>(lds);
    if (g == 0 && lead) red[wave * 16 + j] = part;
    __syncthreads();
    if (wave_in_wg == 0 && g == 0 && valid) {
        float logit = red[j];
#pragma unroll
        for (int wv = 1; wv < WAVES; ++wv) logit += red[wv * 16 + j];
        a.out[stream] = 1.0f / (1.0f + expf(-(logit + a.dense_bias)));
    }
}

}  // namespace pe
