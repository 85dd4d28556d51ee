// bf16-operand GRU + Dense forward (BASELINE.json configs[4]: "bf16 MFCC+GRU ... tol 1e-2"), gfx950.
//
// Same network and recurrence as gru_device.h (model.py:76-82), but the gate matmuls run on
// v_mfma_f32_16x16x32_bf16: weights, features and the hidden state are rounded to bf16 as MFMA
// operands, accumulation, gate non-linearities and the state update stay float32.  K = 32 swallows
// the whole contraction (13 features / <= 32 hidden units) in ONE MFMA per output tile, so a
// timestep is 6 (input) + 4 (z, r) + 2 (candidate) MFMAs of ~17 cycles instead of 41 of 32 -- one
// wave per 16-stream tile, no LDS, no cross-wave hand-off.
//
// Layout: hidden unit u = 8 g + i lives in lane group g = lane >> 4, register i (0..7).  Output tile
// (gate, t) holds rows 4 g + q  <->  unit 8 g + 4 t + q, so a lane's eight z / r / candidate values of
// the two tiles of a gate are exactly its eight units, and bf16(h) packed from those registers IS the
// B operand of the next step (B: lane (g, j) supplies k = 8 g .. 8 g + 7 for stream j).
#pragma once
#include "gru_device.h"

namespace pe {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x4 mfma_bf16(const bf16x8& a, const bf16x8& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ bf16x8 pack_bf16(const float (&v)[8]) {
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (__bf16)v[i];
    return o;
}

// MODE as in gru_device.h (kFeats / kRing / kRows).  Tiles: 0,1 = z, 2,3 = r, 4,5 = candidate.
// DELTA (use_delta) is a compile-time switch: the first differences consume a loaded row as soon as it is loaded, which
// would put a wait for the row prefetch in front of every timestep's MFMAs of the plain network as well.
// RB: the ring holds bf16 rows (ring_precision = 1; a compile-time switch so that the deep row prefetch of that format
// costs 4 registers per step of distance and the float32-row variant none)
template <int MODE, bool DELTA = false, bool RB = false>
__device__ __forceinline__ void gru_tile_bf16(const GruArgs& a, const int tile, const int lane) {
#pragma clang fp contract(off)      // every fusion in the gate arithmetic is spelled out: all kernel shapes round alike
    const int g = lane >> 4, j = lane & 15;
    const long long stream = (long long)tile * kTileStreams + j;
    const bool valid = stream < a.n_streams;
    const int T = a.n_features;

    // resident operands: 6 + 6 tiles x 4 VGPRs of bf16 weights.  The biases ride in the input contraction: k slots
    // 30 and 31 of the x operand hold 1.0 and the matching weight columns the bias split into two bf16 terms (hi + lo,
    // residual <= 2^-17 |b|) -- no 24 accumulator-init registers, which is what lets this tile share a 128-register
    // launch with the MFCC role without scratch traffic in its time loop
    bf16x8 wx[6], wr[6];
    const uint4* wxs = reinterpret_cast<const uint4*>(a.wx_bf16);
    const uint4* wrs = reinterpret_cast<const uint4*>(a.wr_bf16);
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const uint4 u = wxs[t * 64 + lane], v = wrs[t * 64 + lane];
        wx[t] = __builtin_bit_cast(bf16x8, u);
        wr[t] = __builtin_bit_cast(bf16x8, v);
    }
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    const float* xbase = nullptr;
    uint32_t first = 0;
    const uint32_t mask = (uint32_t)(a.ring_slots - 1);
    if (MODE == kRing) {
        const long long sid = gru_stream_of(a, stream, valid);      // the stream whose record and ring rows this lane reads
        const uint32_t ke = gru_window_end(a, sid);
        first = ke - (uint32_t)T;
        xbase = RB ? reinterpret_cast<const float*>(reinterpret_cast<const __bf16*>(a.ring) + gru_ring_cell(a, sid) * kRowFloats)
                   : a.ring + gru_ring_cell(a, sid) * kRowFloats;
    } else if (MODE == kRows) {
        const long long w = valid ? stream : 0;
        xbase = a.feats + ((size_t)w * a.row_stride) * kRowFloats;
    } else {
        xbase = a.feats + (size_t)(valid ? stream : 0) * T * (a.use_delta ? 2 * a.n_in : a.n_in);
    }
    // use_delta (vectorization.py:53-59): K = 32 holds the 13 features in k = 0..15 AND their first differences in
    // k = 16..31, so the same single MFMA per output tile covers the doubled input; lane groups 2, 3 form x_t - x_(t-1)
    // from the rows they fetch (zero at the first timestep); an explicit batch carries its delta columns
    constexpr bool delta = DELTA;
    const int frow = delta ? 2 * a.n_in : a.n_in;
    // lane group g supplies k = 8 g .. 8 g + 7: features 8 (g & 1) .. + 7 (groups 0, 1), zeros or deltas (groups 2, 3).
    // A row is REQUESTED one timestep ahead and only turned into an operand (converted, differenced) at the top of the
    // step that uses it: anything that touches a loaded value earlier puts the L2 round trip into every timestep.
    constexpr bool from_bf16 = MODE == kRing && RB;
    struct XRaw { float v[from_bf16 ? 1 : 8]; uint4 u; };
    const int fg = 8 * (g & 1);                                // first feature of this lane group's slice
    const bool wants = g < 2 || delta;
    auto request_x = [&](int t) -> XRaw {
        XRaw r;
        r.u = uint4{0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < (from_bf16 ? 1 : 8); ++i) r.v[i] = 0.f;
        const int tc = t < T ? t : T - 1;
        if constexpr (from_bf16) {
            // bf16 rows: the 16 bytes a lane group needs ARE its MFMA operand
            const __bf16* p = reinterpret_cast<const __bf16*>(xbase) + (size_t)((first + (uint32_t)tc) & mask) * kTileStreams * kRowFloats + fg;
            if (wants) r.u = *reinterpret_cast<const uint4*>(p);
        } else if constexpr (MODE == kFeats) {
            const float* p = xbase + (size_t)tc * frow + (g >= 2 ? a.n_in : 0);      // groups 2, 3: the batch's delta columns
#pragma unroll
            for (int i = 0; i < 8; ++i) if (valid && wants && fg + i < a.n_in) r.v[i] = p[fg + i];
        } else if (wants) {
            const float* p = (MODE == kRing)
                ? xbase + (size_t)((first + (uint32_t)tc) & mask) * kTileStreams * kRowFloats + fg
                : xbase + (size_t)tc * kRowFloats + fg;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(p), hi = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) { r.v[i] = lo[i]; r.v[4 + i] = hi[i]; }
        }
        return r;
    };
    float vprev[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) vprev[i] = 0.f;
    auto make_x = [&](const XRaw& r, int t) -> bf16x8 {
        if constexpr (from_bf16 && !delta) return __builtin_bit_cast(bf16x8, r.u);   // (zeros for groups 2, 3)
        float v[8];
        if constexpr (from_bf16) {
            const bf16x8 b = __builtin_bit_cast(bf16x8, r.u);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (float)b[i];
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = r.v[i];
            if (MODE == kFeats) return pack_bf16(v);                                  // (an explicit batch carries its delta columns)
        }
        if (delta && g >= 2) {
            float d[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { d[i] = t > 0 ? v[i] - vprev[i] : 0.f; vprev[i] = v[i]; }
            return pack_bf16(d);
        }
        return pack_bf16(v);
    };

    float h[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = 0.f;
    // Rows are requested PF timesteps ahead.  With one or two waves per SIMD (8192 streams per GPU: the shard of BASELINE
    // configs[4]) nothing else hides the L2 / Infinity-Cache round trip of a row: one step ahead, every timestep lasted
    // as long as that round trip (~980 cycles for 12 MFMAs of 16); bf16 rows cost 4 registers per step of distance.
    constexpr int PF = from_bf16 ? 3 : 1;
    XRaw raw[PF];
#pragma unroll
    for (int d = 0; d < PF; ++d) raw[d] = request_x(d);
    for (int t = 0; t < T; ++t) {
        uint4 xu = __builtin_bit_cast(uint4, make_x(raw[0], t));
        if (g == 3) xu.w = 0x3F803F80u;                          // k = 30, 31: bf16(1.0) against the bias columns
        const bf16x8 x = __builtin_bit_cast(bf16x8, xu);
#pragma unroll
        for (int d = 0; d + 1 < PF; ++d) raw[d] = raw[d + 1];
        raw[PF - 1] = request_x(t + PF);
        f32x4 acc[6];
#pragma unroll
        for (int tl = 0; tl < 6; ++tl) acc[tl] = mfma_bf16(wx[tl], x, zero4);
        const bf16x8 hb = pack_bf16(h);
#pragma unroll
        for (int tl = 0; tl < 4; ++tl) acc[tl] = mfma_bf16(wr[tl], hb, acc[tl]);
        float z[8], rh[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            z[i] = hard_sigmoid(acc[i >> 2][i & 3]);
            rh[i] = hard_sigmoid(acc[2 + (i >> 2)][i & 3]) * h[i];
        }
        const bf16x8 rhb = pack_bf16(rh);
#pragma unroll
        for (int tl = 4; tl < 6; ++tl) acc[tl] = mfma_bf16(wr[tl], rhb, acc[tl]);
#pragma unroll
        for (int i = 0; i < 8; ++i) h[i] = gru_blend(z[i], h[i], acc[4 + (i >> 2)][i & 3]);
    }

    float part = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) part = fmaf(h[i], a.wd_bf16[i * 64 + lane], part);
    part += __shfl_xor(part, 16);
    part += __shfl_xor(part, 32);
    if (valid && g == 0) a.out[stream] = 1.0f / (1.0f + expf(-(part + a.dense_bias)));
}

}  // namespace pe
