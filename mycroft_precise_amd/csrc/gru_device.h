// GRU + Dense(1, sigmoid) forward over the sliding [T x F] feature window, for gfx950.
//
// Network: /root/reference/precise/model.py:76-82 (GRU(units, activation='linear') -> Dense(1,
// 'sigmoid')), executed by Runner.predict (/root/reference/precise/network_runner.py:69-74).
// Keras GRUCell (implementation 1, reset_after=False, hard_sigmoid), h0 = 0 for every window:
//     z = hs(x Wz + bz + h Uz)      r = hs(x Wr + br + h Ur)      hs(v) = clip(0.2 v + 0.5, 0, 1)
//     hh = x Wh + bh + (r*h) Uh     h' = z*h + (1-z)*hh           p = sigmoid(h_T . Wd + bd)
//
// Mapping to the machine: the three gate matmuls are the one dense contraction of the path, so
// they run on the f32 matrix cores as v_mfma_f32_16x16x4_f32 with the problem TRANSPOSED:
//     gates^T [gate rows x 16 streams] = W^T [gate rows x K] . x^T / h^T [K x 16 streams]
//   * N = 16 streams of one tile, one wave per tile; weights are the A operand and stay in
//     registers for the whole window (one VGPR per tile and k-step), biases are the C operand of
//     the first MFMA of each chain;
//   * hidden unit u lives in lane group g = u % 4 (lane >> 4), register rho = u / 4.  Gate values
//     are laid out in "slots": slot s = gate*R + rho occupies MFMA output register s % 4 of tile
//     s / 4, whose row 4g + reg is unit 4 rho + g.  So z, r, candidate and h of one unit sit in
//     the same lane, and the D registers of one step ARE the B operands (k-slot g) of the next
//     step's recurrent MFMAs -- no transpose, no LDS, no cross-lane traffic inside the recurrence;
//   * tiles holding z/r slots accumulate h.U in phase 1, tiles holding candidate slots accumulate
//     (r*h).U in phase 2; a tile holding both kinds gets both, with the other rows' weights zero;
//   * the feature ring is stored [tile][slot][stream][16 floats], so lane (g, j) fetches features
//     4g..4g+3 of stream j with one 16-byte load and the wave reads 1 KiB contiguous per timestep
//     (k-step kk of the input projection <-> feature 4g + kk).
#pragma once
#include "pe_common.h"

namespace pe {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float hard_sigmoid(float v) {
    // Keras/TF: clip(0.2 * x + 0.5, 0, 1), as ONE fused multiply-add with the clamp (one rounding where TF does two:
    // <= 1 ulp apart, inside the 1e-4 parity budget by three orders of magnitude).  The fusion is spelled out: left to
    // the compiler's contraction it came out fused in one kernel shape and as multiply + add in another, and the
    // shapes then differed in the last bit.
    return __builtin_amdgcn_fmed3f(__builtin_fmaf(0.2f, v, 0.5f), 0.0f, 1.0f);
}

// h' = z h + (1 - z) c with a fixed sequence of roundings (subtract, multiply, fused multiply-add), for the same reason
__device__ __forceinline__ float gru_blend(float z, float h, float c) {
#pragma clang fp contract(off)
    const float t = (1.0f - z) * c;
    return __builtin_fmaf(z, h, t);
}

__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

template <int R> struct GruShape {
    static constexpr int SLOTS = 3 * R;
    static constexpr int NT = (SLOTS + 3) / 4;       // output tiles
    static constexpr int P1_END = (2 * R + 3) / 4;   // tiles [0, P1_END) hold z/r slots
    static constexpr int P2_BEGIN = (2 * R) / 4;     // tiles [P2_BEGIN, NT) hold candidate slots
    static constexpr int NP2 = NT - P2_BEGIN;
};

// Where a wave's timestep inputs come from.
enum GruInput {
    kFeats = 0,     // explicit [n][T][F] float32 batch                       (Runner.predict)
    kRing = 1,      // the streaming feature ring + per-stream frame counters (Listener.update)
    kRows = 2       // one long [n_frames][16] float32 feature sequence, window w = rows
                    // [w*stride, w*stride + T)                               (simulate.py:92-104)
};

// Input projections of one (stream tile, ring slot): 4 KB laid out [output tile tl][stream j][lane group g][q], so
// that the float4 a lane (g, j) needs for output tile tl sits at ((tl * 16 + j) * 4 + g) * 4 floats and ONE wave-wide
// load of a tile's accumulators reads 1 KB contiguous (a row-per-stream layout made every lane touch its own 64-byte
// segment: 340 cycles of address processing per timestep in the one-wave kernel).  Slot `s` of the ring is
// s * kTileStreams * kProjRow floats further; output tile tl adds tl * kTileStreams * 16 floats.
__device__ __forceinline__ const float* proj_base(const GruArgs& a, const long long sid, int g) {
    return a.proj_ring + (size_t)(sid >> 4) * a.ring_slots * kTileStreams * kProjRow + (size_t)((int)(sid & 15) * 4 + g) * 4;
}
constexpr int kProjTileStride = kTileStreams * 16;       // floats between the accumulators of consecutive output tiles

// One wave = one tile of 16 streams, whole window, weights resident in registers.
// PROJ: every timestep starts from the input projection x.W + b that the MFCC stage stored beside the feature row
// (a.proj_ring), instead of recomputing it with 4 MFMAs per output tile.
// KX = 2: feature rows of 32 floats (17..32 coefficients per frame: general ListenerParams, params.py:28-118): the input
// projection runs over two 16-feature groups (8 k-steps per output tile instead of 4)
template <int R, int MODE, bool PROJ = false, int KX = 1>
__device__ __forceinline__ void gru_tile(const GruArgs& a, const int tile, const int lane) {
#pragma clang fp contract(off)      // every fusion in the gate arithmetic is spelled out: all kernel shapes round alike
    constexpr bool FROM_RING = MODE == kRing;
    constexpr int RF = kRowFloats * KX;           // floats per feature row
    static_assert(!PROJ || (MODE == kRing && GruShape<R>::NT <= 4), "projection rows hold 4 output tiles");
    static_assert(KX == 1 || !PROJ, "projection rows exist for 16-float feature rows");
    using G = GruShape<R>;
    const int g = lane >> 4, j = lane & 15;
    const long long stream = (long long)tile * kTileStreams + j;
    const bool valid = stream < a.n_streams;
    const int T = a.n_features;

    // ---- resident operands ------------------------------------------------------------
    float wx[G::NT][4], wx2[KX == 2 ? G::NT : 1][4], wxd[G::NT][4], wr1[G::P1_END][R], wr2[G::NP2][R], wd[R];
    f32x4 bias[G::NT];
    const bool delta = KX == 1 && a.use_delta != 0;      // add_deltas (vectorization.py:53-59): F more inputs = x_t - x_(t-1)
#pragma unroll
    for (int t = 0; t < G::NT; ++t) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            wx[t][kk] = a.wx[(t * 4 + kk) * 64 + lane];
            if (KX == 2) wx2[t][kk] = a.wx[((G::NT + t) * 4 + kk) * 64 + lane];      // features 16 + 4 g + kk
            wxd[t][kk] = delta ? a.wxd[(t * 4 + kk) * 64 + lane] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) bias[t][q] = a.bias[(t * 4 + q) * 64 + lane];
    }
#pragma unroll
    for (int t = 0; t < G::P1_END; ++t)
#pragma unroll
        for (int rho = 0; rho < R; ++rho) wr1[t][rho] = a.wr1[(t * R + rho) * 64 + lane];
#pragma unroll
    for (int t = 0; t < G::NP2; ++t)
#pragma unroll
        for (int rho = 0; rho < R; ++rho) wr2[t][rho] = a.wr2[((t + G::P2_BEGIN) * R + rho) * 64 + lane];
#pragma unroll
    for (int rho = 0; rho < R; ++rho) wd[rho] = a.wd[rho * 64 + lane];

    // ---- input addressing -------------------------------------------------------------
    const float* xbase = nullptr;
    uint32_t first = 0;           // ring: frame index of timestep 0
    const uint32_t mask = (uint32_t)(a.ring_slots - 1);
    const long long sid = FROM_RING ? gru_stream_of(a, stream, valid) : 0;      // the stream whose record and ring rows this lane reads
    if (FROM_RING) {
        const uint32_t ke = gru_window_end(a, sid);
        first = ke - (uint32_t)T;
        xbase = a.ring + gru_ring_cell(a, sid) * RF + 4 * g;
    } else if (MODE == kRows) {
        const long long w = valid ? stream : 0;               // padded lanes shadow window 0
        xbase = a.feats + ((size_t)w * a.row_stride) * RF + 4 * g;
    } else {
        xbase = a.feats + (size_t)stream * T * (delta ? 2 * a.n_in : a.n_in);
    }
    const int frow = delta ? 2 * a.n_in : a.n_in;         // floats per timestep of an explicit batch
    // hi = 1: the second 16-feature group of a 32-float row (KX = 2)
    auto load_xg = [&](int t, const int hi) -> f32x4 {
        if (FROM_RING) {
            // no branch: rows of padded streams exist (zeroed), t is clamped to the last row, so the
            // prefetch stays in flight across the timestep instead of being waited for at a join
            const int tc = t < T ? t : T - 1;
            const uint32_t slot = (first + (uint32_t)tc) & mask;
            return *reinterpret_cast<const f32x4*>(xbase + (size_t)slot * kTileStreams * RF + 16 * hi);
        }
        if (MODE == kRows) {
            const int tc = t < T ? t : T - 1;
            return *reinterpret_cast<const f32x4*>(xbase + (size_t)tc * RF + 16 * hi);
        }
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (!valid || t >= T) return v;
        const float* p = xbase + (size_t)t * frow + 16 * hi + 4 * g;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) v[kk] = (16 * hi + 4 * g + kk < a.n_in) ? p[kk] : 0.f;
        return v;
    };
    auto load_x = [&](int t) -> f32x4 { return load_xg(t, 0); };
    // explicit batches carry their delta columns (Runner.predict receives add_deltas output)
    auto load_d = [&](int t) -> f32x4 {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (!valid || t >= T) return v;
        const float* p = xbase + (size_t)t * frow + a.n_in + 4 * g;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) v[kk] = (4 * g + kk < a.n_in) ? p[kk] : 0.f;
        return v;
    };

    float h[R];
#pragma unroll
    for (int rho = 0; rho < R; ++rho) h[rho] = 0.f;

    const float* pbase = PROJ ? proj_base(a, sid, g) : nullptr;
    auto load_p = [&](int t, f32x4 (&p)[G::NT]) {
        const int tc = t < T ? t : T - 1;
        const uint32_t slot = (first + (uint32_t)tc) & mask;
        const f32x4* q = reinterpret_cast<const f32x4*>(pbase + (size_t)slot * kTileStreams * kProjRow);
#pragma unroll
        for (int tl = 0; tl < G::NT; ++tl) p[tl] = q[tl * (kProjTileStride / 4)];
    };
    f32x4 x = {0.f, 0.f, 0.f, 0.f};
    // projection rows are requested PD timesteps ahead: at a few thousand streams they come from the other XCDs' L2 /
    // the Infinity Cache (the rows of one engine exceed one XCD's 4 MB), a microsecond away, and a single wave per SIMD
    // has nothing else to hide that behind
    constexpr int PD = 3;
    f32x4 pq[PD][G::NT];
    if (PROJ) {
#pragma unroll
        for (int d = 0; d < PD; ++d) load_p(d, pq[d]);
    } else {
        x = load_x(0);
    }
    f32x4 xprev = {0.f, 0.f, 0.f, 0.f};
    f32x4 xh = {0.f, 0.f, 0.f, 0.f};
    if (KX == 2) xh = load_xg(0, 1);
    for (int t = 0; t < T; ++t) {
        f32x4 xn = {0.f, 0.f, 0.f, 0.f}, xhn = {0.f, 0.f, 0.f, 0.f};
        f32x4 acc[G::NT];
        if (PROJ) {
#pragma unroll
            for (int tl = 0; tl < G::NT; ++tl) acc[tl] = pq[0][tl];

#pragma unroll
            for (int d = 0; d + 1 < PD; ++d)
#pragma unroll
                for (int tl = 0; tl < G::NT; ++tl) pq[d][tl] = pq[d + 1][tl];
#ifndef PE_GRU_ABL_NOLOAD
            load_p(t + PD, pq[PD - 1]);          // prefetch
#endif
        } else {
            xn = load_x(t + 1);                  // prefetch next timestep's features
            if (KX == 2) xhn = load_xg(t + 1, 1);
            // input projection, bias as the initial accumulator
#pragma unroll
            for (int tl = 0; tl < G::NT; ++tl) {
                acc[tl] = mfma(wx[tl][0], x[0], bias[tl]);
#pragma unroll
                for (int kk = 1; kk < 4; ++kk) acc[tl] = mfma(wx[tl][kk], x[kk], acc[tl]);
                if (KX == 2)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) acc[tl] = mfma(wx2[tl][kk], xh[kk], acc[tl]);
            }
        }
        if (delta && !PROJ) {
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
            if (MODE == kFeats) d = load_d(t);
            else if (t > 0) d = x - xprev;
#pragma unroll
            for (int tl = 0; tl < G::NT; ++tl)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) acc[tl] = mfma(wxd[tl][kk], d[kk], acc[tl]);
            xprev = x;
        }
        // phase 1: + h . U for the z / r rows.  (Measured alternatives, MI355X, gru_many_kernel<5> over 2048
        // tiles: this k-outer order 65.4 us; tile-outer runs of R dependent MFMAs 66-67 us; each chain split
        // into two independent halves added at the end 69.3 us, and 16.9 instead of 15.6 us for the 4-wave
        // kernel -- the extra adds and hazards cost more than the shorter dependent chain saves.)
        // Issue order: the tiles that hold r rows first -- r is all the candidate phase waits for -- and the tiles that
        // hold ONLY z rows (z is not needed before the state update) between the candidate MFMAs, where their
        // five-deep dependent chain costs nothing and fills the matrix pipe while the VALU forms r * h.  Per accumulator
        // the order of the MFMAs is unchanged, so every kernel shape still produces the same bits.
        constexpr int ZT_END = R / 4;                  // tiles [0, ZT_END) hold z slots only
#pragma unroll
        for (int rho = 0; rho < R; ++rho)
#pragma unroll
            for (int tl = ZT_END; tl < G::P1_END; ++tl) acc[tl] = mfma(wr1[tl][rho], h[rho], acc[tl]);
        float z[R], rh[R];
#pragma unroll
        for (int rho = 0; rho < R; ++rho) {
            const int sr = R + rho;
            const float r = hard_sigmoid(acc[sr >> 2][sr & 3]);
            rh[rho] = r * h[rho];
        }
        // phase 2: + (r*h) . U for the candidate rows (+ the z-only tiles' share of phase 1)
#pragma unroll
        for (int rho = 0; rho < R; ++rho) {
#ifndef PE_GRU_ABL_NOPHASE2
#pragma unroll
            for (int tl = G::P2_BEGIN; tl < G::NT; ++tl) acc[tl] = mfma(wr2[tl - G::P2_BEGIN][rho], rh[rho], acc[tl]);
#endif
#pragma unroll
            for (int tl = 0; tl < ZT_END; ++tl) acc[tl] = mfma(wr1[tl][rho], h[rho], acc[tl]);
        }
#pragma unroll
        for (int rho = 0; rho < R; ++rho) z[rho] = hard_sigmoid(acc[rho >> 2][rho & 3]);
#pragma unroll
        for (int rho = 0; rho < R; ++rho) {
            const int sh = 2 * R + rho;
            const float hh = acc[sh >> 2][sh & 3];
            h[rho] = gru_blend(z[rho], h[rho], hh);
        }
        x = xn;
        xh = xhn;
    }

    // Dense(1) + sigmoid: reduce over this lane's units, then over the four lane groups
    float part = 0.f;
#pragma unroll
    for (int rho = 0; rho < R; ++rho) part = fmaf(h[rho], wd[rho], part);
    part += __shfl_xor(part, 16);
    part += __shfl_xor(part, 32);
    if (valid && g == 0) {
        const float logit = part + a.dense_bias;
        a.out[stream] = 1.0f / (1.0f + expf(-logit));
    }
}


// ---- four waves per tile ------------------------------------------------------------------------
// With few tiles (4096 streams = 256 tiles on 1024 SIMDs) one wave per tile leaves three quarters
// of the matrix cores idle and the window is a 29-step dependent chain of 41 MFMAs.  Here the four
// waves of a workgroup share one tile: wave w owns output tiles {w, w+4, ...}; every wave keeps
// the full hidden state in registers; per timestep the gate values cross waves through LDS twice
// (z/r after phase 1, candidates after phase 2), 64 floats per slot, and every lane reads back
// exactly the lane position it would have owned.  The next step's input projection is issued while
// the candidates travel from LDS.
// (A five-wave variant with one wave per (tile, phase) role measured the same 21 us stand-alone
// but needs 320-thread workgroups, which halves the residency of the fused launch: rejected.)
template <int R, bool PROJ = false>
__device__ __forceinline__ void gru_tile_mw(const GruArgs& a, const int tile, const int wave, const int lane,
                                            float* S /* [3R][64] floats of LDS */) {
#pragma clang fp contract(off)      // every fusion in the gate arithmetic is spelled out: all kernel shapes round alike
    using G = GruShape<R>;
    constexpr int MAXT = (G::NT + 3) / 4;
    static_assert(!PROJ || MAXT == 1, "projection rows hold 4 output tiles");
    const int g = lane >> 4, j = lane & 15;
    const long long stream = (long long)tile * kTileStreams + j;
    const bool valid = stream < a.n_streams;
    const int T = a.n_features;

    float wx[MAXT][4], wr1[MAXT][R], wr2[MAXT][R], wd[R];
    f32x4 bias[MAXT];
    bool own[MAXT], p1[MAXT], p2[MAXT];
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
        const int tl = wave + 4 * i;
        own[i] = tl < G::NT;
        p1[i] = own[i] && tl < G::P1_END;
        p2[i] = own[i] && tl >= G::P2_BEGIN;
        const int tc = own[i] ? tl : 0;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) wx[i][kk] = a.wx[(tc * 4 + kk) * 64 + lane];
#pragma unroll
        for (int q = 0; q < 4; ++q) bias[i][q] = a.bias[(tc * 4 + q) * 64 + lane];
#pragma unroll
        for (int rho = 0; rho < R; ++rho) {
            wr1[i][rho] = a.wr1[(tc * R + rho) * 64 + lane];
            wr2[i][rho] = a.wr2[(tc * R + rho) * 64 + lane];
        }
    }
#pragma unroll
    for (int rho = 0; rho < R; ++rho) wd[rho] = a.wd[rho * 64 + lane];

    const long long sid = gru_stream_of(a, stream, valid);      // the stream whose record and ring rows this lane reads
    const uint32_t ke = gru_window_end(a, sid);
    const uint32_t first = ke - (uint32_t)T;
    const uint32_t mask = (uint32_t)(a.ring_slots - 1);
    const float* xbase = a.ring + gru_ring_cell(a, sid) * kRowFloats + 4 * g;
    const float* pbase = PROJ ? proj_base(a, sid, g) + kProjTileStride * (wave < G::NT ? wave : 0) : nullptr;
    auto load_x = [&](int t) -> f32x4 {
        const int tc = t < T ? t : T - 1;
        const uint32_t slot = (first + (uint32_t)tc) & mask;
        if (PROJ) return *reinterpret_cast<const f32x4*>(pbase + (size_t)slot * kTileStreams * kProjRow);
        return *reinterpret_cast<const f32x4*>(xbase + (size_t)slot * kTileStreams * kRowFloats);
    };
    // the accumulators a timestep starts from: x.W + b, computed here or (PROJ) fetched as stored by the MFCC stage
    auto xproj = [&](const f32x4& x, f32x4 (&acc)[MAXT]) {
#pragma unroll
        for (int i = 0; i < MAXT; ++i) {
            if (!own[i]) continue;
            if (PROJ) { acc[i] = x; continue; }
            acc[i] = mfma(wx[i][0], x[0], bias[i]);
#pragma unroll
            for (int kk = 1; kk < 4; ++kk) acc[i] = mfma(wx[i][kk], x[kk], acc[i]);
        }
    };

    float h[R], z[R];
#pragma unroll
    for (int rho = 0; rho < R; ++rho) { h[rho] = 0.f; z[rho] = 0.f; }
    f32x4 acc[MAXT];
#pragma unroll
    for (int i = 0; i < MAXT; ++i) acc[i] = bias[i];
    f32x4 x = load_x(0);
    xproj(x, acc);
    x = load_x(1);
    for (int t = 0; t < T; ++t) {
        // phase 1: + h . U on the tiles holding z / r slots, publish hard-sigmoided gates
#pragma unroll
        for (int i = 0; i < MAXT; ++i) {
            if (!p1[i]) continue;
#pragma unroll
            for (int rho = 0; rho < R; ++rho) acc[i] = mfma(wr1[i][rho], h[rho], acc[i]);
            const int base = 4 * (wave + 4 * i);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (base + q < 2 * R) S[(base + q) * 64 + lane] = hard_sigmoid(acc[i][q]);
        }
        __syncthreads();
        float rh[R];
#pragma unroll
        for (int rho = 0; rho < R; ++rho) {
            z[rho] = S[rho * 64 + lane];
            rh[rho] = S[(R + rho) * 64 + lane] * h[rho];
        }
        // phase 2: + (r*h) . U on the tiles holding candidate slots, publish candidates
#pragma unroll
        for (int i = 0; i < MAXT; ++i) {
            if (!p2[i]) continue;
#pragma unroll
            for (int rho = 0; rho < R; ++rho) acc[i] = mfma(wr2[i][rho], rh[rho], acc[i]);
            const int base = 4 * (wave + 4 * i);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (base + q >= 2 * R && base + q < 3 * R) S[(base + q) * 64 + lane] = acc[i][q];
        }
        __syncthreads();
        float hh[R];
#pragma unroll
        for (int rho = 0; rho < R; ++rho) hh[rho] = S[(2 * R + rho) * 64 + lane];
        // next step's input projection while the candidates travel from LDS
        const f32x4 xn = load_x(t + 2);
        xproj(x, acc);
        x = xn;
#pragma unroll
        for (int rho = 0; rho < R; ++rho) h[rho] = gru_blend(z[rho], h[rho], hh[rho]);
    }

    if (wave == 0) {
        float part = 0.f;
#pragma unroll
        for (int rho = 0; rho < R; ++rho) part = fmaf(h[rho], wd[rho], part);
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        if (valid && g == 0) a.out[stream] = 1.0f / (1.0f + expf(-(part + a.dense_bias)));
    }
}

// ---- four waves per tile, stock width (17 <= H <= 20, R = 5) ----------------------------------------
// Same split as gru_tile_mw -- tile 0 = {z0..z3}, tile 1 = {z4, r0, r1, r2}, tile 2 = {r3, r4 | c0, c1},
// tile 3 = {c2, c3, c4} -- with the input projections moved off the critical path: waves 0/1 compute
// theirs while waves 2/3 run phase 2; wave 3, idle during phase 1, computes its own AND tile 2's and
// hands the latter to wave 2 through LDS.  What stays serial per timestep is two 5-MFMA chains and
// two LDS hand-offs.
// KX = 2: feature rows of 32 floats (17..32 coefficients per frame, general ListenerParams): the input projection runs over
// two 16-feature groups, in the order gru_tile<5, MODE, false, 2> issues them
template <bool PROJ, int KX = 1>
__device__ __forceinline__ void gru_tile_mw5(const GruArgs& a, const int tile, const int wave, const int lane,
                                             float* S /* [15][64] gate slots + [64][4] tile-2 projection */) {
#pragma clang fp contract(off)      // every fusion in the gate arithmetic is spelled out: all kernel shapes round alike
    constexpr int R = 5;
    constexpr int RF = kRowFloats * KX;
    static_assert(KX == 1 || !PROJ, "projection rows exist for 16-float feature rows");
    float* X2 = S + 3 * R * 64;
    const int g = lane >> 4, j = lane & 15;
    const long long stream = (long long)tile * kTileStreams + j;
    const bool valid = stream < a.n_streams;
    const int T = a.n_features;

    float wx[4], wx2[4], wxh[4], wx2h[4], wrA[R], wrB[R], wd[R];
    f32x4 bias, bias2;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        wx[kk] = a.wx[(wave * 4 + kk) * 64 + lane];
        wx2[kk] = a.wx[(2 * 4 + kk) * 64 + lane];
        wxh[kk] = KX == 2 ? a.wx[((4 + wave) * 4 + kk) * 64 + lane] : 0.f;      // features 16 + 4 g + kk
        wx2h[kk] = KX == 2 ? a.wx[((4 + 2) * 4 + kk) * 64 + lane] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        bias[q] = a.bias[(wave * 4 + q) * 64 + lane];
        bias2[q] = a.bias[(2 * 4 + q) * 64 + lane];
    }
#pragma unroll
    for (int rho = 0; rho < R; ++rho) {
        wrA[rho] = a.wr1[(wave * R + rho) * 64 + lane];      // phase-1 rows of this wave's tile (zero for tile 3)
        wrB[rho] = a.wr2[(wave * R + rho) * 64 + lane];      // phase-2 rows (zero for tiles 0, 1)
        wd[rho] = a.wd[rho * 64 + lane];
    }

    const long long sid = gru_stream_of(a, stream, valid);      // the stream whose record and ring rows this lane reads
    const uint32_t ke = gru_window_end(a, sid);
    const uint32_t first = ke - (uint32_t)T;
    const uint32_t mask = (uint32_t)(a.ring_slots - 1);
    // what a wave fetches per timestep: the feature row (4 features per lane) -- or, PROJ, the input projection of
    // ITS OWN output tile as the MFCC stage stored it (then no wave computes projections and nothing is handed over)
    const float* xbase = PROJ ? proj_base(a, sid, g) + kProjTileStride * wave
                              : a.ring + gru_ring_cell(a, sid) * RF + 4 * g;
    const size_t xstride = (size_t)kTileStreams * (PROJ ? kProjRow : RF);
    struct XRow { f32x4 lo, hi; };
    auto load_x = [&](int t) -> XRow {
        const int tc = t < T ? t : T - 1;
        const uint32_t slot = (first + (uint32_t)tc) & mask;
        XRow r;
        r.lo = *reinterpret_cast<const f32x4*>(xbase + (size_t)slot * xstride);
        r.hi = KX == 2 ? *reinterpret_cast<const f32x4*>(xbase + (size_t)slot * xstride + 16) : f32x4{0.f, 0.f, 0.f, 0.f};
        return r;
    };
    auto xproj = [&](const float (&w)[4], const float (&wh)[4], const f32x4& b, const XRow& x) -> f32x4 {
        if (PROJ) return x.lo;
        f32x4 acc = mfma(w[0], x.lo[0], b);
#pragma unroll
        for (int kk = 1; kk < 4; ++kk) acc = mfma(w[kk], x.lo[kk], acc);
        if (KX == 2)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc = mfma(wh[kk], x.hi[kk], acc);
        return acc;
    };
    auto lds_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    float* Sl = S + lane;

    float h[R], z[R];
#pragma unroll
    for (int rho = 0; rho < R; ++rho) { h[rho] = 0.f; z[rho] = 0.f; }
    f32x4 accx = xproj(wx, wxh, bias, load_x(0));       // this wave's tile, timestep 0
    XRow x1 = load_x(1);

    for (int t = 0; t < T; ++t) {
        if (wave == 3) {
            // phase 1 of the others: projections of timestep t+1 for tile 3 (own) and tile 2 (wave 2's)
            const f32x4 an = xproj(wx, wxh, bias, x1);
            if (!PROJ) {
                const f32x4 a2 = xproj(wx2, wx2h, bias2, x1);
                *reinterpret_cast<f32x4*>(X2 + lane * 4) = a2;
            }
            x1 = load_x(t + 2);
            lds_barrier();                                           // A
            float rr[R];
#pragma unroll
            for (int rho = 0; rho < R; ++rho) { z[rho] = Sl[rho * 64]; rr[rho] = Sl[(R + rho) * 64]; }
            f32x4 acc = accx;
#pragma unroll
            for (int rho = 0; rho < R; ++rho) acc = mfma(wrB[rho], rr[rho] * h[rho], acc);
            Sl[12 * 64] = acc[0]; Sl[13 * 64] = acc[1]; Sl[14 * 64] = acc[2];
            lds_barrier();                                           // B
            accx = an;
        } else if (wave == 2) {
            f32x4 acc = accx;
#pragma unroll
            for (int rho = 0; rho < R; ++rho) acc = mfma(wrA[rho], h[rho], acc);
            Sl[8 * 64] = hard_sigmoid(acc[0]); Sl[9 * 64] = hard_sigmoid(acc[1]);
            lds_barrier();                                           // A
            float rr[R];
#pragma unroll
            for (int rho = 0; rho < R; ++rho) { z[rho] = Sl[rho * 64]; rr[rho] = Sl[(R + rho) * 64]; }
            f32x4 an;
            if (PROJ) { an = x1.lo; x1 = load_x(t + 2); }
            else an = *reinterpret_cast<const f32x4*>(X2 + lane * 4);
#pragma unroll
            for (int rho = 0; rho < R; ++rho) acc = mfma(wrB[rho], rr[rho] * h[rho], acc);
            Sl[10 * 64] = acc[2]; Sl[11 * 64] = acc[3];
            lds_barrier();                                           // B
            accx = an;
        } else {
            f32x4 acc = accx;
#pragma unroll
            for (int rho = 0; rho < R; ++rho) acc = mfma(wrA[rho], h[rho], acc);
#pragma unroll
            for (int q = 0; q < 4; ++q) Sl[(4 * wave + q) * 64] = hard_sigmoid(acc[q]);
            lds_barrier();                                           // A
#pragma unroll
            for (int rho = 0; rho < R; ++rho) z[rho] = Sl[rho * 64];
            accx = xproj(wx, wxh, bias, x1);                         // phase 2 of the others
            x1 = load_x(t + 2);
            lds_barrier();                                           // B
        }
        float hh[R];
#pragma unroll
        for (int rho = 0; rho < R; ++rho) hh[rho] = Sl[(2 * R + rho) * 64];
#pragma unroll
        for (int rho = 0; rho < R; ++rho) h[rho] = gru_blend(z[rho], h[rho], hh[rho]);
    }

    if (wave == 0) {
        float part = 0.f;
#pragma unroll
        for (int rho = 0; rho < R; ++rho) part = fmaf(h[rho], wd[rho], part);
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        if (valid && g == 0) a.out[stream] = 1.0f / (1.0f + expf(-(part + a.dense_bias)));
    }
}

template <int R, bool PROJ = false, int KX = 1>
__device__ __forceinline__ void gru_tile_mw_any(const GruArgs& a, const int tile, const int wave, const int lane, float* S) {
    static_assert(KX == 1 || R == 5, "32-float feature rows on four waves: the stock width only");
    if constexpr (R == 5) gru_tile_mw5<PROJ, KX>(a, tile, wave, lane, S);
    else gru_tile_mw<R, PROJ>(a, tile, wave, lane, S);
}

}  // namespace pe
