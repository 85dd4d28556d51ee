// Stock-width GRU (17 <= H <= 20, R = 5) re-tiled so that H = 20 costs three MFMA tiles instead of four, for gfx950.
//
// Network: /root/reference/precise/model.py:76-82, executed by Runner.predict
// (/root/reference/precise/network_runner.py:69-74); equations as in gru_device.h.
//
// Tiling.  TZ = {z of units 0..15}, TX = {r of units 0..15}, TC = {candidate of units 0..15} are full 16-row tiles
// (unit 4 q + g <-> register q of lane group g, as everywhere else).  The three gates of the LAST FOUR units (16..19:
// the quarter tile that made H = 20 cost a fourth tile and five more eight-pass MFMAs per phase) are formed from
// partial sums: lane (g, stream) accumulates, over the five source units 4 rho + g its lane group holds, the products
// for each of the four target units 16 + a -- either as 5 v_mfma_f32_4x4x1_16B_f32 (two passes each; block = (lane
// group, stream quad), A = U[4 rho + g][16 + (lane & 3)], B = the lane's own h[rho]) or as 20 v_fma_f32 with the same
// operands in the same order (bit-identical: a K = 1 MFMA is one fma per output; NOT true of the K = 4 form: four chained
// 4x4x1 MFMAs and one 16x16x4 round differently, measured) -- and v_sum4 (two v_permlane swaps, three adds, fixed order)
// reduces over the four lane groups and delivers unit 16 + g to lane group g.
//
// Two shapes, bit-identical to each other:
//   * gru_tile_v   one wave per tile: 15 eight-pass + 15 two-pass recurrent MFMAs per timestep where gru_tile<5> issues
//                  25 eight-pass ones (what the other launches of a re-tiled engine use; in the throughput regime it is
//                  5 % slower than gru_tile<5>, so large engines keep the classic tiling);
//   * gru_tile_cw  few tiles (one or two per compute unit: up to 8192 streams): the window is a chain of 29 timesteps x 2
//                  dependent mat-vecs, and what counts is the length of ONE timestep on ONE wave.  gru_tile_mw5 splits
//                  the gate rows of a tile over four waves and pays two workgroup hand-offs (LDS write, s_barrier, LDS
//                  read: ~190 cycles each, measured with tools/micro/gru_chain.hip) ON that chain every timestep.
//                  Here the chain stays on one wave (R: r of all units, candidate, blend); z -- needed only by the
//                  blend, a whole timestep later -- and the input projections (16 of the 41 MFMAs of a timestep) are
//                  taken off it by three helper waves; one s_barrier per timestep, which R reaches last.  The whole
//                  feature ring of the tile (32 slots x 1 KB) is staged in LDS by all four waves in the SAME round trip
//                  as the stream counters (which slot is which timestep is decided afterwards, per lane), so the chain
//                  contains no global access.  The splits that were built and timed are listed at gru_tile_cw below
//                  and in DESIGN.md 4.2 (form 1) / profiles/DESIGN_notebook_r1-r5.md 4.2b.
#pragma once
#include "gru_device.h"
#include "gru_cw_pack.h"

namespace pe {


// LDS of one tile (floats): the mailboxes (CwBox), then the staged ring
struct CwLds {
    static constexpr int BOX = 0;                       // mailboxes (CwBox)
    static constexpr int XR = 2048;                     // [32 slots][64][4]
    static constexpr int FLOATS = XR + kCwSlots * 256;
};
constexpr size_t kCwLdsBytes = (size_t)CwLds::FLOATS * sizeof(float);

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}

// p[a] of lane group g' = partial sum of target a over the source units of g'.  Returns, in lane group g, the total of
// target g: (p_g' + p_g'+2 over the two halves) then (even row + odd row).  Fixed order of additions.
__device__ __forceinline__ float v_sum4(const f32x4& p) {
#pragma clang fp contract(off)
    const auto s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(p[0]), __float_as_uint(p[2]), false, false);
    const auto s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(p[1]), __float_as_uint(p[3]), false, false);
    const float q0 = __uint_as_float(s0[0]) + __uint_as_float(s0[1]);      // lower half: target 0, upper half: target 2
    const float q1 = __uint_as_float(s1[0]) + __uint_as_float(s1[1]);      // lower half: target 1, upper half: target 3
    const auto s2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(q0), __float_as_uint(q1), false, false);
    return __uint_as_float(s2[0]) + __uint_as_float(s2[1]);
}

// partial sums of the four targets as an fma chain over rho (what five 4x4x1 MFMAs starting from C = 0 compute)
__device__ __forceinline__ f32x4 v_partials(const float (&w)[4][5], const float (&v)[5]) {
    f32x4 p;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        float acc = __builtin_fmaf(w[a][0], v[0], 0.0f);
#pragma unroll
        for (int rho = 1; rho < 5; ++rho) acc = __builtin_fmaf(w[a][rho], v[rho], acc);
        p[a] = acc;
    }
    return p;
}

// LDS traffic of this wave drained, then the workgroup barrier.  The wait is the BUILTIN (s_waitcnt 0xc07f =
// lgkmcnt(0), other counters untouched): the compiler's own wait-count pass sees it and does not wait again for
// mailbox reads that are known to have landed.  WHERE the barrier sits in a wave's instruction stream is part of the
// design (R must reach it late, the helpers early), so nothing is scheduled across it.
__device__ __forceinline__ void cw_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
// a value loaded before a loop is waited for HERE, not at its first use inside the loop (where the s_waitcnt would be
// issued again every iteration)
__device__ __forceinline__ void cw_pin(float& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void cw_pin(f32x4& v) { asm volatile("" : "+v"(v)); }      // (also: computed HERE, not sunk below a later loop)

// post-update emitted-frame count of a stream (same arithmetic as gru_tile / mfcc_book_tile)
__device__ __forceinline__ uint32_t cw_window_end(const GruArgs& a, const long long sid) { return gru_window_end(a, sid); }

// x.W + b of one 16-row tile: bias as the initial accumulator, then the four k-steps in order
__device__ __forceinline__ f32x4 cw_xproj(const float (&w)[4], const f32x4& b, const f32x4& x) {
    f32x4 acc = mfma(w[0], x[0], b);
#pragma unroll
    for (int kk = 1; kk < 4; ++kk) acc = mfma(w[kk], x[kk], acc);
    return acc;
}

// use_delta (vectorization.py:53-59): + (x_t - x_(t-1)) . W[F .. 2F-1] on top of an input projection, k-steps in order
// (gru_tile does the same: bias, features, differences, then the recurrent chain)
__device__ __forceinline__ f32x4 cw_dproj(const float (&w)[4], f32x4 acc, const f32x4& d) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) acc = mfma(w[kk], d[kk], acc);
    return acc;
}

// One timestep of the recurrence on one wave, given the four accumulator inits (x.W + b of TZ, TX, TC and of the quarter
// tile TV = {z, r, candidate of units 16..19 in registers 0, 1, 2}).  VF: the partial sums as VALU fma chains (weights
// wf) instead of 4x4x1 MFMAs (weights wv).  h[rho] <-> unit 4 rho + g.
template <bool VF>
struct CwStep {
    float wrZ[5], wrX[5], wrC[5];
    float wv[3][5];             // 4x4x1 A operands           (!VF)
    float wf[3][4][5];          // per-target weights         (VF)
    __device__ __forceinline__ void load(const float* cw, const int lane) {
#pragma unroll
        for (int rho = 0; rho < 5; ++rho) {
            wrZ[rho] = cw[CwPack::WR + (0 * 5 + rho) * 64 + lane];
            wrX[rho] = cw[CwPack::WR + (1 * 5 + rho) * 64 + lane];
            wrC[rho] = cw[CwPack::WR + (2 * 5 + rho) * 64 + lane];
        }
#pragma unroll
        for (int gate = 0; gate < 3; ++gate)
#pragma unroll
            for (int rho = 0; rho < 5; ++rho) {
                if (!VF) wv[gate][rho] = cw[CwPack::WV + (gate * 5 + rho) * 64 + lane];
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    if (VF) wf[gate][a][rho] = cw[CwPack::WF + ((gate * 4 + a) * 5 + rho) * 64 + lane];
            }
    }
    __device__ __forceinline__ void pin() {
#pragma unroll
        for (int rho = 0; rho < 5; ++rho) { cw_pin(wrZ[rho]); cw_pin(wrX[rho]); cw_pin(wrC[rho]); }
#pragma unroll
        for (int gate = 0; gate < 3; ++gate)
#pragma unroll
            for (int rho = 0; rho < 5; ++rho) {
                if (!VF) cw_pin(wv[gate][rho]);
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    if (VF) cw_pin(wf[gate][a][rho]);
            }
    }
    __device__ __forceinline__ f32x4 partials(const int gate, const float (&v)[5]) const {
        if (VF) return v_partials(wf[gate], v);
        f32x4 p = mfma4(wv[gate][0], v[0], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
        for (int rho = 1; rho < 5; ++rho) p = mfma4(wv[gate][rho], v[rho], p);
        return p;
    }
    // h <- h(t+1).  accZ / accX / accC / accV: the inits of this timestep.
    __device__ __forceinline__ void step(float (&h)[5], f32x4 accZ, f32x4 accX, f32x4 accC, const f32x4& accV) const {
#pragma clang fp contract(off)      // every fusion in the gate arithmetic is spelled out: all kernel shapes round alike
        // phase 1: + h.U for the r rows (what the candidate waits for) and the z rows, interleaved: two independent
        // chains keep the matrix pipe issuing every 32 cycles where one dependent chain issues every 40
#pragma unroll
        for (int rho = 0; rho < 5; ++rho) {
            accX = mfma(wrX[rho], h[rho], accX);
            accZ = mfma(wrZ[rho], h[rho], accZ);
        }
        const f32x4 pr = partials(1, h);
        const f32x4 pz = partials(0, h);
        float rh[5];
        rh[4] = hard_sigmoid(accV[1] + v_sum4(pr)) * h[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) rh[q] = hard_sigmoid(accX[q]) * h[q];
        // phase 2: + (r*h).U for the candidate rows
#pragma unroll
        for (int rho = 0; rho < 5; ++rho) accC = mfma(wrC[rho], rh[rho], accC);
        const f32x4 pc = partials(2, rh);
        const float z4 = hard_sigmoid(accV[0] + v_sum4(pz));
        const float c4 = accV[2] + v_sum4(pc);
#pragma unroll
        for (int q = 0; q < 4; ++q) h[q] = gru_blend(hard_sigmoid(accZ[q]), h[q], accC[q]);
        h[4] = gru_blend(z4, h[4], c4);
    }
};

// Dense(1) + sigmoid from h (unit 4 rho + g in lane group g)
__device__ __forceinline__ void cw_head(const GruArgs& a, const float (&h)[5], const float (&wd)[5], const long long stream, const bool valid, const int g) {
    float part = 0.f;
#pragma unroll
    for (int rho = 0; rho < 5; ++rho) part = fmaf(h[rho], wd[rho], part);
    part += __shfl_xor(part, 16);
    part += __shfl_xor(part, 32);
    if (valid && g == 0) a.out[stream] = 1.0f / (1.0f + expf(-(part + a.dense_bias)));
}

// ---- one wave per tile ----------------------------------------------------------------------------------------
// DELTA: use_delta models (compile-time here: sixteen more resident weights)
template <int MODE, bool DELTA = false>
__device__ __forceinline__ void gru_tile_v(const GruArgs& a, const int tile, const int lane) {
    const int g = lane >> 4, j = lane & 15;
    const long long stream = (long long)tile * kTileStreams + j;
    const bool valid = stream < a.n_streams;
    const int T = a.n_features;
    const float* cw = a.cw;
    CwStep<false> S;
    S.load(cw, lane);
    float wx[4][4], wxd[DELTA ? 4 : 1][4], wd[5];
    f32x4 bias[4];
    constexpr bool delta = DELTA;
#pragma unroll
    for (int tl = 0; tl < 4; ++tl)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            wx[tl][kk] = cw[CwPack::WX + (tl * 4 + kk) * 64 + lane];
            if constexpr (DELTA) wxd[tl][kk] = cw[CwPack::WXD + (tl * 4 + kk) * 64 + lane];
            bias[tl][kk] = cw[CwPack::BIAS + (tl * 4 + kk) * 64 + lane];
        }
#pragma unroll
    for (int rho = 0; rho < 5; ++rho) wd[rho] = a.wd[rho * 64 + lane];

    const float* xbase;
    uint32_t first = 0;
    const uint32_t mask = (uint32_t)(a.ring_slots - 1);
    if (MODE == kRing) {
        const long long sid = gru_stream_of(a, stream, valid);      // the stream whose record and ring rows this lane reads
        first = cw_window_end(a, sid) - (uint32_t)T;
        xbase = a.ring + gru_ring_cell(a, sid) * kRowFloats + 4 * g;
    } else if (MODE == kRows) {
        const long long w = valid ? stream : 0;               // padded lanes shadow window 0
        xbase = a.feats + ((size_t)w * a.row_stride) * kRowFloats + 4 * g;
    } else {
        xbase = a.feats + (size_t)stream * T * (delta ? 2 * a.n_in : a.n_in);        // explicit [n][T][F] batch (Runner.predict)
    }
    const int frow = delta ? 2 * a.n_in : a.n_in;             // floats per timestep of an explicit batch (it carries its delta columns)
    auto load_d = [&](int t) -> f32x4 {                       // kFeats: the batch's own delta columns
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (!valid || t >= T) return v;
        const float* p = xbase + (size_t)t * frow + a.n_in + 4 * g;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) v[kk] = (4 * g + kk < a.n_in) ? p[kk] : 0.f;
        return v;
    };
    auto load_x = [&](int t) -> f32x4 {
        const int tc = t < T ? t : T - 1;
        if (MODE == kRing) {
            const uint32_t slot = (first + (uint32_t)tc) & mask;
            return *reinterpret_cast<const f32x4*>(xbase + (size_t)slot * kTileStreams * kRowFloats);
        }
        if (MODE == kRows) return *reinterpret_cast<const f32x4*>(xbase + (size_t)tc * kRowFloats);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (!valid || t >= T) return v;
        const float* p = xbase + (size_t)t * frow + 4 * g;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) v[kk] = (4 * g + kk < a.n_in) ? p[kk] : 0.f;
        return v;
    };
    float h[5];
#pragma unroll
    for (int rho = 0; rho < 5; ++rho) h[rho] = 0.f;
    f32x4 x = load_x(0), xprev = x;
    for (int t = 0; t < T; ++t) {
        const f32x4 xn = load_x(t + 1);
        f32x4 aZ = cw_xproj(wx[kTZ], bias[kTZ], x), aX = cw_xproj(wx[kTX], bias[kTX], x);
        f32x4 aC = cw_xproj(wx[kTC], bias[kTC], x), aV = cw_xproj(wx[kTV], bias[kTV], x);
        if constexpr (DELTA) {
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
            if (MODE == kFeats) d = load_d(t);
            else if (t > 0) d = x - xprev;
            aZ = cw_dproj(wxd[kTZ], aZ, d); aX = cw_dproj(wxd[kTX], aX, d);
            aC = cw_dproj(wxd[kTC], aC, d); aV = cw_dproj(wxd[kTV], aV, d);
            xprev = x;
        }
        S.step(h, aZ, aX, aC, aV);
        x = xn;
    }
    cw_head(a, h, wd, stream, valid, g);
}

// ---- four waves per tile: R (r, candidate, blend) | Z1 (z of units 0..15) | Z2 (z of units 16..19, TV inits) | P (TX, TC inits)
// What was measured on the way here (tools/micro/gru_chain.hip, 4096 streams = one tile per compute unit): a lone wave
// issues one eight-pass MFMA per ~40 cycles whatever the dependencies, two-pass MFMAs cost it ~20 cycles apiece and
// VALU fma chains for the partial sums MORE than that; a value handed from one wave to another through LDS takes
// ~100 cycles per hop (write -> visible -> read) plus the synchronisation, ~190 with an s_barrier, no less with
// tag-polled mailboxes.  Kernel times: gru_tile_mw5 16.3 us; the whole recurrence on one wave + projections on helpers
// 16.9; r of units 16..19 and all of z on helpers, tag-polled mailboxes and no barrier at all 15.2; THIS split 14.2:
//   * R keeps the two dependent chains of a timestep -- X = h.U on TX, then C = (r*h).U on TC, 5 MFMAs each -- and
//     the r / candidate partial sums of units 16..19 (needed within the timestep: a round trip to another wave is
//     longer than the 10 two-pass MFMAs);
//   * z is needed only by the blend at the very end: R publishes h(t) + a tag (fire and forget), Z1 / Z2 poll the tag,
//     run the z chain / partials and publish z(t) before barrier B(t) -- the ONE s_barrier of a timestep, which R
//     reaches last; behind it R reads z(t) and the accumulator inits of step t + 1 in one go;
//   * the input projections (x.W + b: 16 of the 41 MFMAs of a timestep in gru_tile) are computed one timestep ahead
//     by the helpers (Z1 its own, Z2 the quarter tile's, P those of TX and TC) and reach R as ready-made
//     accumulator inits through parity-buffered LDS mailboxes.
struct CwBox {                  // LDS mailboxes (floats), [64 lanes][4] each unless noted
    static constexpr int PX = 0;                // [2 parities] TX init
    static constexpr int PC = PX + 512;         // [2] TC init
    static constexpr int PV = PC + 512;         // [2][64][2] r, candidate init of units 16..19
    static constexpr int SH4 = PV + 256;        // h of units 4 q + g
    static constexpr int SH1 = SH4 + 256;       // [64] h of unit 16 + g
    static constexpr int SZ4 = SH1 + 64;        // z of units 4 q + g
    static constexpr int SZ1 = SZ4 + 256;       // [64] z of unit 16 + g
    static constexpr int TAG = SZ1 + 64;        // [1] timestep whose h is in SH4 / SH1
    static constexpr int END = TAG + 4;
    // (kernels.hip keeps four role-slot ints at 1984 .. 1987)
};
static_assert(CwBox::END <= 1984 && CwBox::END <= CwLds::XR, "mailboxes overlap the role slots / the staged ring");

// (the other splits and instruction orders of this timestep that were built and timed -- candidate of units 16..19 on Z2, R's
//  timestep hand-interleaved, mailbox reads before the barrier with tag validation, no barrier at all, XDL chains on R -- are in
//  the history of this file and in profiles/DESIGN_notebook_r1-r5.md 4.2b / 4.6; what is here is the one that won:
//  the r partial sums ride in the slack of the X chain, the candidate chain and its partial sums are ONE pinned sequence)
template <bool VF>
__device__ __forceinline__ void gru_tile_cw(const GruArgs& a, const int tile, const int wave, const int lane, float* S) {
#pragma clang fp contract(off)      // every fusion in the gate arithmetic is spelled out: all kernel shapes round alike
    const int g = lane >> 4, j = lane & 15;
    const long long stream = (long long)tile * kTileStreams + j;
    const bool valid = stream < a.n_streams;
    const int T = a.n_features;
    const float* cw = a.cw;

    // ---- one round trip: the counters, this wave's share of the tile's ring, the weights ------------------------
    // (a.ring_slots == kCwSlots here: cw_four_waves_ok)
    const long long sid = gru_stream_of(a, stream, valid);      // the stream whose record and ring rows this lane reads
    const float* xbase = a.ring + gru_ring_cell(a, sid) * kRowFloats + 4 * g;
    f32x4 stage[kCwSlots / 4];
#pragma unroll
    for (int i = 0; i < kCwSlots / 4; ++i) stage[i] = *reinterpret_cast<const f32x4*>(xbase + (size_t)(wave + 4 * i) * kTileStreams * kRowFloats);
    // (the counters are only REQUESTED here: the arithmetic on them -- first_slot() -- comes after a role has requested its
    //  weights.  Loads return in order: consumed up here, the counters made every role wait for its share of the ring before
    //  the weight loads even went out, a second round trip in the prologue: 4.1 k cycles instead of ~2.3 k)
    const KeRequest ke_req = gru_ke_request(a, sid);      // the stream's record, both sides (records exist for padded streams too)
    uint32_t first = 0;
    auto first_slot = [&]() {                               // cw_window_end(a, sid) - T, from the values requested above
        first = gru_ke_resolve(a, ke_req) - (uint32_t)T;
    };
    float* const XR = S + CwLds::XR;
    float* const L4 = S + lane * 4;
    float* const L2 = S + lane * 2;
    float* const L1 = S + lane;
    // (volatile accesses through a GENERIC pointer would become flat loads / stores: name the LDS address space)
    typedef volatile __attribute__((address_space(3))) int* lds_vint;
    typedef const volatile __attribute__((address_space(3))) float* lds_vfloat;
    typedef const volatile __attribute__((address_space(3))) f32x4* lds_vf4;
    const lds_vint TAG = (lds_vint)(S + CwBox::TAG);
    auto x_row = [&](int t) -> f32x4 {                         // the stream's timestep t from the staged ring
        const int tc = t < T ? t : T - 1;
        const uint32_t slot = (first + (uint32_t)tc) & (uint32_t)(kCwSlots - 1);
        return *reinterpret_cast<const f32x4*>(XR + (slot * 64 + lane) * 4);
    };
    auto stage_out = [&]() {
#pragma unroll
        for (int i = 0; i < kCwSlots / 4; ++i) *reinterpret_cast<f32x4*>(XR + ((wave + 4 * i) * 64 + lane) * 4) = stage[i];
    };
    // h(t) as R published it: spin on the tag, then the lane's own position (DS operations of a wave execute in order,
    // R stores h before the tag and this wave reads the tag first)
    auto wait_h = [&](int t, float (&h)[5]) {
        for (;;) {
            const int tag = *TAG;
            const f32x4 h4 = *(lds_vf4)(L4 + CwBox::SH4);
            h[4] = *(lds_vfloat)(L1 + CwBox::SH1);
            h[0] = h4[0]; h[1] = h4[1]; h[2] = h4[2]; h[3] = h4[3];
            if (__builtin_amdgcn_readfirstlane(tag) == t) break;
        }
    };
    auto partials = [&](const float (&wf)[4][5], const float (&wv)[5], const float (&v)[5]) -> f32x4 {
        if (VF) return v_partials(wf, v);
        f32x4 p = mfma4(wv[0], v[0], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
        for (int rho = 1; rho < 5; ++rho) p = mfma4(wv[rho], v[rho], p);
        return p;
    };
    auto load_v = [&](const int gate, float (&wf)[4][5], float (&wv)[5]) {
#pragma unroll
        for (int rho = 0; rho < 5; ++rho) {
            if (!VF) wv[rho] = cw[CwPack::WV + (gate * 5 + rho) * 64 + lane];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (VF) wf[q][rho] = cw[CwPack::WF + ((gate * 4 + q) * 5 + rho) * 64 + lane];
        }
    };
    auto pin_v = [&](float (&wf)[4][5], float (&wv)[5]) {
#pragma unroll
        for (int rho = 0; rho < 5; ++rho) {
            if (!VF) cw_pin(wv[rho]);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (VF) cw_pin(wf[q][rho]);
        }
    };

    // use_delta: the helpers add (x_t - x_(t-1)) . W[F .. 2F-1] to the inits they compute (nothing at t = 0); a wave-uniform
    // runtime flag -- the helpers have the registers and the issue slots (P goes from 8 to 16 eight-pass MFMAs per timestep)
    const bool delta = a.use_delta != 0;
    auto load_wxd = [&](const int tl, float (&w)[4]) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) w[kk] = delta ? cw[CwPack::WXD + (tl * 4 + kk) * 64 + lane] : 0.f;
    };

    if (wave == 0) {
        // ================= R ===============================================================================
        float wrX[5], wrC[5], wfr[4][5], wfc[4][5], wvr[5], wvc[5], wd[5];
#pragma unroll
        for (int rho = 0; rho < 5; ++rho) {
            wrX[rho] = cw[CwPack::WR + (1 * 5 + rho) * 64 + lane];
            wrC[rho] = cw[CwPack::WR + (2 * 5 + rho) * 64 + lane];
            wd[rho] = a.wd[rho * 64 + lane];
        }
        load_v(1, wfr, wvr);
        load_v(2, wfc, wvc);
        stage_out();
        float h[5];
#pragma unroll
        for (int rho = 0; rho < 5; ++rho) h[rho] = 0.f;
        *reinterpret_cast<f32x4*>(L4 + CwBox::SH4) = f32x4{0.f, 0.f, 0.f, 0.f};
        L1[CwBox::SH1] = 0.f;
        if (lane == 0) *TAG = 0;
#pragma unroll
        for (int rho = 0; rho < 5; ++rho) { cw_pin(wrX[rho]); cw_pin(wrC[rho]); cw_pin(wd[rho]); }
        pin_v(wfr, wvr);
        pin_v(wfc, wvc);
        cw_barrier();                       // ring staged, h(0) = 0 published
        cw_barrier();                       // inits of step 0 in the mailboxes
        f32x4 accX = *reinterpret_cast<const f32x4*>(L4 + CwBox::PX);
        f32x4 accC = *reinterpret_cast<const f32x4*>(L4 + CwBox::PC);
        float ri = L2[CwBox::PV], ci = L2[CwBox::PV + 1];
        __builtin_amdgcn_s_waitcnt(0xc07f);
        for (int t = 0; t < T; ++t) {
            // phase 1: r.  Units 0..15: + h.U on TX; units 16..19: partial sums, reduced across the lane groups
            f32x4 pr;
            if (!VF) {
                // the partial sums depend on h only: issued WITH the X chain (a two-pass MFMA in the slack behind every
                // eight-pass one), their reduction runs under the chain's tail and r of units 16..19 is ready with the rest
                // (pinned instruction by instruction: left alone the scheduler moves the whole partial-sum chain behind the X chain)
                accX = mfma(wrX[0], h[0], accX);
                __builtin_amdgcn_sched_barrier(0);
                pr = mfma4(wvr[0], h[0], f32x4{0.f, 0.f, 0.f, 0.f});
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rho = 1; rho < 5; ++rho) {
                    accX = mfma(wrX[rho], h[rho], accX);
                    __builtin_amdgcn_sched_barrier(0);
                    pr = mfma4(wvr[rho], h[rho], pr);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int rho = 0; rho < 5; ++rho) accX = mfma(wrX[rho], h[rho], accX);
                pr = partials(wfr, wvr, h);
            }
            float rh[5];
            rh[4] = hard_sigmoid(ri + v_sum4(pr)) * h[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) rh[q] = hard_sigmoid(accX[q]) * h[q];
            // phase 2: candidate
            float c4 = 0.f;
            if (!VF) {
                // the candidate chain and its partial sums, interleaved and pinned like phase 1
                f32x4 pc;
                accC = mfma(wrC[0], rh[0], accC);
                __builtin_amdgcn_sched_barrier(0);
                pc = mfma4(wvc[0], rh[0], f32x4{0.f, 0.f, 0.f, 0.f});
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rho = 1; rho < 5; ++rho) {
                    accC = mfma(wrC[rho], rh[rho], accC);
                    __builtin_amdgcn_sched_barrier(0);
                    pc = mfma4(wvc[rho], rh[rho], pc);
                    __builtin_amdgcn_sched_barrier(0);
                }
                c4 = ci + v_sum4(pc);
            } else {
#pragma unroll
                for (int rho = 0; rho < 5; ++rho) accC = mfma(wrC[rho], rh[rho], accC);
                const f32x4 pc = partials(wfc, wvc, rh);
                c4 = ci + v_sum4(pc);
            }
            const int nb = (t + 1) & 1;
            f32x4 z, nX, nC;
            float z4, nri, nci;
            auto read_boxes = [&]() {
                z = *(lds_vf4)(L4 + CwBox::SZ4);
                z4 = *(lds_vfloat)(L1 + CwBox::SZ1);
                nX = *(lds_vf4)(L4 + CwBox::PX + nb * 256);
                nC = *(lds_vf4)(L4 + CwBox::PC + nb * 256);
                nri = *(lds_vfloat)(L2 + CwBox::PV + nb * 128);
                nci = *(lds_vfloat)(L2 + CwBox::PV + nb * 128 + 1);
            };
            cw_barrier();                                                   // B(t): z(t) and the inits of step t + 1 are in LDS
            read_boxes();
            f32x4 hn;
#pragma unroll
            for (int q = 0; q < 4; ++q) hn[q] = h[q] = gru_blend(z[q], h[q], accC[q]);
            h[4] = gru_blend(z4, h[4], c4);
            *reinterpret_cast<f32x4*>(L4 + CwBox::SH4) = hn;
            L1[CwBox::SH1] = h[4];
            if (lane == 0) *TAG = t + 1;
            accX = nX; accC = nC; ri = nri; ci = nci;
        }
        cw_head(a, h, wd, stream, valid, g);
    } else if (wave == 1) {
        // ================= Z1: z of units 0..15 ==============================================================
        float wrZ[5], wx[4], bb[4];
#pragma unroll
        for (int rho = 0; rho < 5; ++rho) wrZ[rho] = cw[CwPack::WR + (0 * 5 + rho) * 64 + lane];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) { wx[kk] = cw[CwPack::WX + (kTZ * 4 + kk) * 64 + lane]; bb[kk] = cw[CwPack::BIAS + (kTZ * 4 + kk) * 64 + lane]; }
        float wxd[4];
        load_wxd(kTZ, wxd);
        first_slot();
        stage_out();
#pragma unroll
        for (int rho = 0; rho < 5; ++rho) cw_pin(wrZ[rho]);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) { cw_pin(wx[kk]); cw_pin(bb[kk]); cw_pin(wxd[kk]); }
        const f32x4 bias = {bb[0], bb[1], bb[2], bb[3]};
        cw_barrier();
        f32x4 xc = x_row(0);
        f32x4 accZ = cw_xproj(wx, bias, xc);
        f32x4 xn = x_row(1);
        cw_barrier();
        for (int t = 0; t < T; ++t) {
            f32x4 nZ = cw_xproj(wx, bias, xn);              // next step's init, while h(t) is on its way
            if (delta) { nZ = cw_dproj(wxd, nZ, xn - xc); xc = xn; }
            xn = x_row(t + 2);
            cw_pin(nZ);
            float h[5];
            wait_h(t, h);
#pragma unroll
            for (int rho = 0; rho < 5; ++rho) accZ = mfma(wrZ[rho], h[rho], accZ);
            f32x4 z;
#pragma unroll
            for (int q = 0; q < 4; ++q) z[q] = hard_sigmoid(accZ[q]);
            *reinterpret_cast<f32x4*>(L4 + CwBox::SZ4) = z;
            cw_barrier();                                                   // B(t)
            accZ = nZ;
        }
    } else if (wave == 2) {
        // ================= Z2: z of units 16..19; inits of the quarter tile =================================
        float wfz[4][5], wvz[5], wx[4], bb[4];
        load_v(0, wfz, wvz);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) { wx[kk] = cw[CwPack::WX + (kTV * 4 + kk) * 64 + lane]; bb[kk] = cw[CwPack::BIAS + (kTV * 4 + kk) * 64 + lane]; }
        float wxd[4];
        load_wxd(kTV, wxd);
        first_slot();
        stage_out();
        pin_v(wfz, wvz);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) { cw_pin(wx[kk]); cw_pin(bb[kk]); cw_pin(wxd[kk]); }
        const f32x4 bias = {bb[0], bb[1], bb[2], bb[3]};
        cw_barrier();
        f32x4 xc = x_row(0);
        f32x4 v = cw_xproj(wx, bias, xc);
        float zi = v[0];
        L2[CwBox::PV] = v[1];
        L2[CwBox::PV + 1] = v[2];
        f32x4 xn = x_row(1);
        cw_barrier();
        for (int t = 0; t < T; ++t) {
            const int nb = (t + 1) & 1;
            v = cw_xproj(wx, bias, xn);                      // inits of step t + 1, while h(t) is on its way
            if (delta) { v = cw_dproj(wxd, v, xn - xc); xc = xn; }
            xn = x_row(t + 2);
            cw_pin(v);
            L2[CwBox::PV + nb * 128] = v[1];
            L2[CwBox::PV + nb * 128 + 1] = v[2];
            float h[5];
            wait_h(t, h);
            L1[CwBox::SZ1] = hard_sigmoid(zi + v_sum4(partials(wfz, wvz, h)));
            cw_barrier();                                                   // B(t)
            zi = v[0];
        }
    } else {
        // ================= P: inits of TX and TC ===========================================================
        float w0[4], w1[4], c0[4], c1[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            w0[kk] = cw[CwPack::WX + (kTX * 4 + kk) * 64 + lane];
            w1[kk] = cw[CwPack::WX + (kTC * 4 + kk) * 64 + lane];
            c0[kk] = cw[CwPack::BIAS + (kTX * 4 + kk) * 64 + lane];
            c1[kk] = cw[CwPack::BIAS + (kTC * 4 + kk) * 64 + lane];
        }
        float d0[4], d1[4];
        load_wxd(kTX, d0);
        load_wxd(kTC, d1);
        first_slot();
        stage_out();
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) { cw_pin(w0[kk]); cw_pin(w1[kk]); cw_pin(c0[kk]); cw_pin(c1[kk]); cw_pin(d0[kk]); cw_pin(d1[kk]); }
        const f32x4 b0 = {c0[0], c0[1], c0[2], c0[3]}, b1 = {c1[0], c1[1], c1[2], c1[3]};
        cw_barrier();                       // ring staged
        f32x4 xc = x_row(0);
        *reinterpret_cast<f32x4*>(L4 + CwBox::PX) = cw_xproj(w0, b0, xc);
        *reinterpret_cast<f32x4*>(L4 + CwBox::PC) = cw_xproj(w1, b1, xc);
        f32x4 xn = x_row(1);
        cw_barrier();
        for (int t = 0; t < T; ++t) {
            const int nb = (t + 1) & 1;
            f32x4 p0 = cw_xproj(w0, b0, xn), p1 = cw_xproj(w1, b1, xn);     // inits of step t + 1
            if (delta) { const f32x4 d = xn - xc; p0 = cw_dproj(d0, p0, d); p1 = cw_dproj(d1, p1, d); xc = xn; }
            *reinterpret_cast<f32x4*>(L4 + CwBox::PX + nb * 256) = p0;
            *reinterpret_cast<f32x4*>(L4 + CwBox::PC + nb * 256) = p1;
            xn = x_row(t + 2);
            cw_barrier();                                                   // B(t)
        }
    }
}

}  // namespace pe
