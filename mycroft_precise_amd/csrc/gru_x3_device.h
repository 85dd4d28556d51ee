// Float32 GRU + Dense forward on the XDL matrix pipe: every float32 operand split into three bf16 pieces (gfx950).
//
// Same network and recurrence as gru_device.h (model.py:76-82, network_runner.py:69-74).  Why a third form exists:
// on gfx950 the f32-input MFMAs (v_mfma_f32_16x16x4_f32) run at the f32 VECTOR rate and keep every other instruction
// of their SIMD from issuing for their whole pass count, while the bf16 MFMAs (the XDL pipe, 16 x the rate) run beside
// VALU and LDS work of other waves (tools/micro/pipe_overlap.hip, profiles/round4/r4x_pipe_overlap_xdl.csv).  In the
// throughput regime -- more stream tiles than the machine has SIMDs -- the float32 network therefore costs the SUM of
// its matrix time and the MFCC role's vector time.  Here the gate matmuls are float32 PRODUCTS formed on the XDL pipe:
//     v = v_hi + v_mid + v_lo          three bf16 pieces, round-to-nearest-even of the running remainder: the pieces
//                                      carry >= 24 significant bits, so the sum is v exactly (bf16 has float32's exponent)
//     w . v ~= w_hi v_hi + w_hi v_mid + w_mid v_hi + w_mid v_mid + w_hi v_lo + w_lo v_hi
// six exact products (8 x 8 significant bits each), accumulated in float32 by the MFMA; the three dropped terms are
// below 2^-23 |w v|, the size of ONE float32 rounding of the product -- the result differs from a float32 fma chain by
// what two float32 summation orders differ by (tests: the float32 guard of tests/test_gpu_parity.py, and its distance
// to a float64 evaluation equals the float32 kernels').  State, gate arithmetic and the head stay float32.
//
// Layout (one wave = one tile of 16 streams, K = 32 per MFMA):
//   * lane group g = lane >> 4 owns units 4 g .. 4 g + 3 and unit 16 + g: "own unit" o = 0..4;
//   * output tiles: TZ, TR, TC = z / r / candidate of units 0..15 (row 4 g + q <-> unit 4 g + q, so a lane's four output
//     registers are its own units 0..3), TQ = the quarter tile: row 4 g + 0 / 1 / 2 = z / r / candidate of unit 16 + g;
//     TQ is evaluated twice per timestep (against h for its z and r rows, against r.h for its candidate row);
//   * recurrent contraction, 20 units x 6 terms = 120 k-slots = 4 MFMAs per output tile.  Lane group g supplies the k-slots
//     8 g .. 8 g + 7 of every MFMA from the units it owns -- the D registers of one step become the B operands of the
//     next without any cross-lane traffic:
//         m = 0:  A = [W_hi  (own 0..3) | W_hi  (own 0..3)]   B = [h_hi  (0..3) | h_mid (0..3)]
//         m = 1:  A = [W_mid (own 0..3) | W_mid (own 0..3)]   B = the same registers as m = 0
//         m = 2:  A = [W_hi  (own 0..3) | W_lo  (own 0..3)]   B = [h_lo  (0..3) | h_hi  (0..3)]
//         m = 3:  A = [W_hi, W_hi, W_mid, W_mid, W_hi, W_lo, 0, 0] of own unit 4,   B = [hi, mid, hi, mid, lo, hi, 0, 0]
//   * input contraction: lane group g supplies features 4 g .. 4 g + 3 (ONE 16-byte load of the ring row), 3 MFMAs per
//     output tile with the same piece pattern as m = 0..2.  The bias rides as the weight row of pseudo-feature F against
//     x = 1.0 (its three pieces add up to the float32 bias exactly): no accumulator-init registers.  F <= 15.
// 32 MFMAs of 4 passes per timestep (512 XDL cycles) against 41 of 8 passes (1312 cycles of the whole SIMD) in
// gru_device.h; ~115 vector instructions per timestep for the splits and the gates.
#pragma once
#include "gru_bf16_device.h"

namespace pe {

// two float32 -> one dword of two bf16 (round to nearest even), element 0 in the low half
__device__ __forceinline__ uint32_t x3_pk(float a, float b) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 p;
    p[0] = (__bf16)a;
    p[1] = (__bf16)b;
    return __builtin_bit_cast(uint32_t, p);
}
__device__ __forceinline__ float x3_lo_f(uint32_t p) { return __builtin_bit_cast(float, p << 16); }

// The split of the four values a lane holds in TILE layout (register q <-> row 4 g + q: gate outputs, the state, a
// feature row) runs on the matrix pipe too: with A = -I restricted to the k-slots that carry a piece,
//     D = C + A.B = v - piece        exactly (the remainder of a rounding is representable),
// so a level of the split is one packed conversion per pair and ONE MFMA instead of unpack / unpack / subtract per value.
// `keep` is the operand [hi (own 0..3) | mid (own 0..3)] itself: the first MFMA reads its hi half (the mid half still
// holds the previous split's pieces -- finite, against zeros of A), the second its mid half.
struct X3Ident { bf16x8 lo_half, hi_half; };       // -1 at k-slot e = (row & 3) (resp. 4 + (row & 3)) of k-group row >> 2
__device__ __forceinline__ X3Ident x3_identity(const int lane) {
    const int i = lane & 15, gk = lane >> 4;
    const uint32_t m1 = 0xBF80u;                    // bf16(-1.0)
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    if ((i >> 2) == gk) w[(i & 3) >> 1] = (i & 1) ? (m1 << 16) : m1;
    X3Ident r;
    r.lo_half = __builtin_bit_cast(bf16x8, uint4{w[0], w[1], 0u, 0u});
    r.hi_half = __builtin_bit_cast(bf16x8, uint4{0u, 0u, w[0], w[1]});
    return r;
}
struct X3Ops { uint4 b0, b2; };                     // [hi | mid], [lo | hi]
__device__ __forceinline__ X3Ops x3_split_tile(const f32x4 v, uint4& keep, const X3Ident& id) {
    keep.x = x3_pk(v[0], v[1]);
    keep.y = x3_pk(v[2], v[3]);
    const f32x4 r1 = mfma_bf16(id.lo_half, __builtin_bit_cast(bf16x8, keep), v);
    keep.z = x3_pk(r1[0], r1[1]);
    keep.w = x3_pk(r1[2], r1[3]);
    const f32x4 r2 = mfma_bf16(id.hi_half, __builtin_bit_cast(bf16x8, keep), r1);
    X3Ops o;
    o.b0 = keep;
    o.b2 = uint4{x3_pk(r2[0], r2[1]), x3_pk(r2[2], r2[3]), keep.x, keep.y};
    return o;
}
// own unit 4 (one value per lane) on the vector pipe: [hi, mid | hi, mid | lo, hi | 0, 0]
__device__ __forceinline__ uint4 x3_split_one(const float v) {
#pragma clang fp contract(off)
    const uint32_t hi = x3_pk(v, 0.f);
    const float r1 = v - x3_lo_f(hi);                                  // exact
    const uint32_t mid = x3_pk(r1, 0.f);
    const float r2 = r1 - x3_lo_f(mid);                                // exact
    const uint32_t lo = x3_pk(r2, 0.f);                                // exact (<= 8 significant bits left)
    const uint32_t hm = hi | (mid << 16);
    const uint32_t lh = lo | (hi << 16);
    return uint4{hm, hm, lh, 0u};
}

template <int MODE>
__device__ __forceinline__ void gru_tile_x3(const GruArgs& a, const int tile, const int lane) {
#pragma clang fp contract(off)      // every fusion in the gate arithmetic is spelled out (gru_device.h)
    const int g = lane >> 4, j = lane & 15;
    const long long stream = (long long)tile * kTileStreams + j;
    const bool valid = stream < a.n_streams;
    const int T = a.n_features;

    // resident A operands: (16 + 12) x 4 VGPRs, and the two halves of -I
    const uint4* blob = reinterpret_cast<const uint4*>(a.x3);
    bf16x8 ar[kX3Tiles][kX3RecOps], ax[kX3Tiles][kX3InOps];
#pragma unroll
    for (int t = 0; t < kX3Tiles; ++t) {
#pragma unroll
        for (int m = 0; m < kX3RecOps; ++m) ar[t][m] = __builtin_bit_cast(bf16x8, blob[kX3ArOff + (t * kX3RecOps + m) * 64 + lane]);
#pragma unroll
        for (int m = 0; m < kX3InOps; ++m) ax[t][m] = __builtin_bit_cast(bf16x8, blob[kX3AxOff + (t * kX3InOps + m) * 64 + lane]);
    }
    const X3Ident ident = x3_identity(lane);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    const float* xbase = nullptr;
    uint32_t first = 0;
    const uint32_t mask = (uint32_t)(a.ring_slots - 1);
    if (MODE == kRing) {
        const long long sid = gru_stream_of(a, stream, valid);      // the stream whose record and ring rows this lane reads
        const uint32_t ke = gru_window_end(a, sid);
        first = ke - (uint32_t)T;
        xbase = a.ring + gru_ring_cell(a, sid) * kRowFloats;
    } else if (MODE == kRows) {
        const long long w = valid ? stream : 0;
        xbase = a.feats + ((size_t)w * a.row_stride) * kRowFloats;
    } else {
        xbase = a.feats + (size_t)(valid ? stream : 0) * T * a.n_in;
    }
    const int f0 = 4 * g;                                      // first feature of this lane group
    const int fbias = a.n_in - f0;                             // which of the four is pseudo-feature F (x = 1.0), if any
    auto request_x = [&](int t) -> f32x4 {
        const int tc = t < T ? t : T - 1;
        if constexpr (MODE == kFeats) {
            f32x4 r = zero4;
            const float* p = xbase + (size_t)tc * a.n_in;
#pragma unroll
            for (int i = 0; i < 4; ++i) if (valid && f0 + i < a.n_in) r[i] = p[f0 + i];
            return r;
        } else {
            const float* p = (MODE == kRing)
                ? xbase + (size_t)((first + (uint32_t)tc) & mask) * kTileStreams * kRowFloats + f0
                : xbase + (size_t)tc * kRowFloats + f0;
            return *reinterpret_cast<const f32x4*>(p);
        }
    };
    uint4 keep_x = {0u, 0u, 0u, 0u}, keep_h = {0u, 0u, 0u, 0u};
    // x.W + b of one timestep for the four output tiles
    auto project = [&](f32x4 x, f32x4 (&acc)[kX3Tiles]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) if (i == fbias) x[i] = 1.0f;
        const X3Ops o = x3_split_tile(x, keep_x, ident);
        const bf16x8 b0 = __builtin_bit_cast(bf16x8, o.b0), b2 = __builtin_bit_cast(bf16x8, o.b2);
#pragma unroll
        for (int t = 0; t < kX3Tiles; ++t) {
            acc[t] = mfma_bf16(ax[t][0], b0, zero4);
            acc[t] = mfma_bf16(ax[t][1], b0, acc[t]);
            acc[t] = mfma_bf16(ax[t][2], b2, acc[t]);
        }
    };
    auto recur = [&](const bf16x8 (&w)[kX3RecOps], const X3Ops& o, const uint4& b3, f32x4 c) -> f32x4 {
        c = mfma_bf16(w[0], __builtin_bit_cast(bf16x8, o.b0), c);
        c = mfma_bf16(w[1], __builtin_bit_cast(bf16x8, o.b0), c);
        c = mfma_bf16(w[2], __builtin_bit_cast(bf16x8, o.b2), c);
        c = mfma_bf16(w[3], __builtin_bit_cast(bf16x8, b3), c);
        return c;
    };

    f32x4 h = zero4;            // own units 0..3
    float h4 = 0.f;             // own unit 4
    f32x4 accx[kX3Tiles];
    project(request_x(0), accx);
    // Two row buffers with fixed roles (no copies of data in flight): a row is requested two timesteps before the
    // projection that consumes it, and the request is pinned where it stands (left alone it sinks to its use)
    f32x4 row_a = request_x(1), row_b = request_x(2);
    auto step = [&](const int t, f32x4& row) {
        // phase 1: z, r (and the quarter tile's z / r rows) from h
        const X3Ops oh = x3_split_tile(h, keep_h, ident);
        const uint4 oh4 = x3_split_one(h4);
        const f32x4 az = recur(ar[0], oh, oh4, accx[0]);
        const f32x4 arr = recur(ar[1], oh, oh4, accx[1]);
        const f32x4 aq = recur(ar[3], oh, oh4, accx[3]);
        const f32x4 cx = accx[2], qx = accx[3];
        // the next timestep's input projection fills the matrix pipe while the gates are formed (and stays here: hoisted to
        // the top of the step, its row would be awaited right behind the request of the other buffer)
        __builtin_amdgcn_sched_barrier(0);
        project(row, accx);
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
        // phase 2: candidates from r.h
        const X3Ops orh = x3_split_tile(rh, keep_h, ident);
        const uint4 orh4 = x3_split_one(rh4);
        const f32x4 ac = recur(ar[2], orh, orh4, cx);
        const f32x4 aq2 = recur(ar[3], orh, orh4, qx);
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

    const float* wd = reinterpret_cast<const float*>(blob + kX3WdOff);
    float part = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) part = fmaf(h[i], wd[i * 64 + lane], part);
    part = fmaf(h4, wd[4 * 64 + lane], part);
    part += __shfl_xor(part, 16);
    part += __shfl_xor(part, 32);
    if (valid && g == 0) a.out[stream] = 1.0f / (1.0f + expf(-(part + a.dense_bias)));
}

}  // namespace pe
