// bf16-operand GRU + Dense forward for networks of <= 20 units (BASELINE.json configs[4]; the stock network), gfx950.
//
// Same arithmetic contract as gru_bf16_device.h -- weights, features and the hidden state rounded to bf16 as operands of
// v_mfma_f32_16x16x32_bf16, float32 accumulate / gates / state, tolerance 1e-2 (model.py:76-82, network_runner.py:69-74)
// -- in the unit layout of gru_x3_device.h: lane group g = lane >> 4 owns units 4 g .. 4 g + 3 and unit 16 + g, output
// tiles TZ / TR / TC = z / r / candidate of units 0..15 (row 4 g + q <-> unit 4 g + q) and the quarter tile TQ (row
// 4 g + 0 / 1 / 2 = z / r / candidate of unit 16 + g), evaluated against h and against r.h.
// Why: gru_bf16_device.h keeps unit 8 g + i in register i of lane group g, i.e. EIGHT gate values per lane for twenty units
// (lane group 3 idle, group 2 half idle) and six output tiles per operand: 12 MFMAs and ~70 vector instructions per
// timestep.  On gfx950 the time of a SIMD is the SUM of its XDL time and its vector-issue time (tools/micro/
// pipe_overlap.hip), and at one wave per tile the vector instructions of a timestep are its dependent chain.  Here a lane
// holds FIVE values: 9 MFMAs (4 input, 3 + 2 recurrent) and ~45 vector instructions per timestep.
//   * recurrent operand: k-slot 8 g + e of lane group g <-> its own unit e (e = 0..4), slots 5..7 zero;
//   * input operand: k-slot 8 g + e <-> feature 4 g + e (e = 0..3: ONE 16-byte / 8-byte load of the row) and, with
//     use_delta (vectorization.py:53-59), its first difference in slot 8 g + 4 + e; the bias rides as the weight rows of
//     pseudo-features F and F + 1 against 1.0 (hi + lo: residual <= 2^-17 |b|), F <= 14.
//
// Shipped in round 5 as the bf16 network of every engine it fits (<= 20 units, <= 14 features): network launch 17.7 vs 26.9 us
// at 65 536 streams, 7.2 vs 9.5 us at 8192 (profiles/round5/r5a_time_packed_unpacked.log).  Round 4 built it and held it back:
// in the FUSED launch ~0.7 % of the float32 MFCC frames computed beside this role came out slightly wrong, timing-dependent
// (profiles/round4/r4v_b20_fused_corruption.log).  Every kernel that hosts the float32 frame role is now compiled WITHOUT
// packed float32 instructions (kernels.hip: PE_NO_PK_F32) -- 0 wrong streams in every run of rounds 4 and 5, 1e8-frame soaks
// at 8192 and 65 536 streams clean (tools/gpu_frame_soak.py), and the frame role is 2-3 % FASTER that way.  The instruction
// pair behind the round-4 effect was not isolated (tools/micro/pk_swap_hazard.hip replays the emitted sequence: clean).
#pragma once
#include "gru_bf16_device.h"

namespace pe {

__device__ __forceinline__ f32x4 b20_mfma(const bf16x8& a, const bf16x8& b, const f32x4& c) { return mfma_bf16(a, b, c); }

// two float32 -> one dword of two bf16 (round to nearest even), element 0 in the low half
__device__ __forceinline__ uint32_t b20_pk(float a, float b) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 p;
    p[0] = (__bf16)a;
    p[1] = (__bf16)b;
    return __builtin_bit_cast(uint32_t, p);
}

// MODE as in gru_device.h (kFeats / kRing / kRows); DELTA, RB as in gru_bf16_device.h (compile-time switches)
template <int MODE, bool DELTA = false, bool RB = false>
__device__ __forceinline__ void gru_tile_b20(const GruArgs& a, const int tile, const int lane) {
#pragma clang fp contract(off)      // every fusion in the gate arithmetic is spelled out: all kernel shapes round alike
    const int g = lane >> 4, j = lane & 15;
    const long long stream = (long long)tile * kTileStreams + j;
    const bool valid = stream < a.n_streams;
    const int T = a.n_features;
    constexpr bool from_bf16 = MODE == kRing && RB;

    // resident A operands: 4 + 4 tiles x 4 VGPRs
    const uint4* blob = reinterpret_cast<const uint4*>(a.b20);
    bf16x8 ar[kB20Tiles], ax[kB20Tiles];
#pragma unroll
    for (int t = 0; t < kB20Tiles; ++t) {
        ar[t] = __builtin_bit_cast(bf16x8, blob[kB20ArOff + t * 64 + lane]);
        ax[t] = __builtin_bit_cast(bf16x8, blob[kB20AxOff + t * 64 + lane]);
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
        xbase = a.feats + (size_t)(valid ? stream : 0) * T * (DELTA ? 2 * a.n_in : a.n_in);
    }
    const int f0 = 4 * g;                                      // first feature of this lane group
    // pseudo-features F and F + 1 (the bias, hi and lo) against bf16(1.0): which halves of this lane group's two feature dwords
    uint32_t one01 = 0u, one23 = 0u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bool is_one = f0 + i == a.n_in || f0 + i == a.n_in + 1;
        const uint32_t bits = is_one ? (i & 1 ? 0x3F800000u : 0x00003F80u) : 0u;
        if (i < 2) one01 |= bits; else one23 |= bits;
    }
    // a requested row: f32 rows as four floats (an explicit batch: + its four delta columns), bf16 rows as two dwords
    struct XRaw { f32x4 v, d; uint32_t u0, u1; };
    auto request_x = [&](int t) -> XRaw {
        XRaw r;
        r.v = zero4; r.d = zero4; r.u0 = 0u; r.u1 = 0u;
        const int tc = t < T ? t : T - 1;
        if constexpr (from_bf16) {
            const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const __bf16*>(xbase) + (size_t)((first + (uint32_t)tc) & mask) * kTileStreams * kRowFloats + f0);
            r.u0 = u.x; r.u1 = u.y;
        } else if constexpr (MODE == kFeats) {
            const float* p = xbase + (size_t)tc * (DELTA ? 2 * a.n_in : a.n_in);
#pragma unroll
            for (int i = 0; i < 4; ++i) if (valid && f0 + i < a.n_in) { r.v[i] = p[f0 + i]; if (DELTA) r.d[i] = p[a.n_in + f0 + i]; }
        } else {
            const float* p = (MODE == kRing)
                ? xbase + (size_t)((first + (uint32_t)tc) & mask) * kTileStreams * kRowFloats + f0
                : xbase + (size_t)tc * kRowFloats + f0;
            r.v = *reinterpret_cast<const f32x4*>(p);
        }
        return r;
    };
    f32x4 vprev = zero4;
    // the input operand of a timestep: [x (4 features) | x_t - x_(t-1) (use_delta) or zeros], the bias slots OR-ed in
    auto make_x = [&](const XRaw& r, const int t) -> bf16x8 {
        uint32_t x01, x23, d01 = 0u, d23 = 0u;
        if constexpr (from_bf16) {
            x01 = r.u0; x23 = r.u1;
            if constexpr (DELTA) {
                const f32x4 v = {__builtin_bit_cast(float, r.u0 << 16), __builtin_bit_cast(float, r.u0 & 0xffff0000u),
                                 __builtin_bit_cast(float, r.u1 << 16), __builtin_bit_cast(float, r.u1 & 0xffff0000u)};
                const f32x4 d = t > 0 ? v - vprev : zero4;
                vprev = v;
                d01 = b20_pk(d[0], d[1]); d23 = b20_pk(d[2], d[3]);
            }
        } else {
            x01 = b20_pk(r.v[0], r.v[1]); x23 = b20_pk(r.v[2], r.v[3]);
            if constexpr (DELTA) {
                f32x4 d = r.d;                                   // (an explicit batch carries its delta columns)
                if constexpr (MODE != kFeats) { d = t > 0 ? r.v - vprev : zero4; vprev = r.v; }
                d01 = b20_pk(d[0], d[1]); d23 = b20_pk(d[2], d[3]);
            }
        }
        return __builtin_bit_cast(bf16x8, uint4{x01 | one01, x23 | one23, d01, d23});
    };
    auto project = [&](const XRaw& r, const int t, f32x4 (&acc)[kB20Tiles]) {
        const bf16x8 x = make_x(r, t);
#pragma unroll
        for (int tl = 0; tl < kB20Tiles; ++tl) acc[tl] = b20_mfma(ax[tl], x, zero4);
    };
    auto operand = [&](const f32x4& v, const float v4) -> bf16x8 {
        return __builtin_bit_cast(bf16x8, uint4{b20_pk(v[0], v[1]), b20_pk(v[2], v[3]), b20_pk(v4, 0.f), 0u});
    };

    f32x4 h = zero4;            // own units 0..3
    float h4 = 0.f;             // own unit 4
    f32x4 accx[kB20Tiles];
    project(request_x(0), 0, accx);
    // two row buffers with fixed roles, each requested two timesteps before the projection that consumes it (gru_x3_device.h)
    XRaw row_a = request_x(1), row_b = request_x(2);
    auto step = [&](const int t, XRaw& row) {
        const bf16x8 hb = operand(h, h4);
        const f32x4 az = b20_mfma(ar[0], hb, accx[0]);
        const f32x4 arr = b20_mfma(ar[1], hb, accx[1]);
        const f32x4 aq = b20_mfma(ar[3], hb, accx[3]);
        const f32x4 cx = accx[2], qx = accx[3];
        __builtin_amdgcn_sched_barrier(0);
        project(row, t + 1, accx);
        row = request_x(t + 3);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 z, rh;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            z[i] = hard_sigmoid(az[i]);
            rh[i] = hard_sigmoid(arr[i]) * h[i];
        }
        const float z4 = hard_sigmoid(aq[0]);
        const float rh4 = hard_sigmoid(aq[1]) * h4;
        const bf16x8 rhb = operand(rh, rh4);
        const f32x4 ac = b20_mfma(ar[2], rhb, cx);
        const f32x4 aq2 = b20_mfma(ar[3], rhb, qx);
#pragma unroll
        for (int i = 0; i < 4; ++i) h[i] = gru_blend(z[i], h[i], ac[i]);
        h4 = gru_blend(z4, h4, aq2[2]);
    };
    int t = 0;
    for (; t + 1 < T; t += 2) {
        step(t, row_a);
        step(t + 1, row_b);
    }
    if (t < T) step(t, row_a);

    const float* wd = reinterpret_cast<const float*>(blob + kB20WdOff);
    float part = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) part = fmaf(h[i], wd[i * 64 + lane], part);
    part = fmaf(h4, wd[4 * 64 + lane], part);
    part += __shfl_xor(part, 16);
    part += __shfl_xor(part, 32);
    if (valid && g == 0) a.out[stream] = 1.0f / (1.0f + expf(-(part + a.dense_bias)));
}

}  // namespace pe
