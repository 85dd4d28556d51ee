// Streaming-state bookkeeping of the MFCC front end and the small helpers its kernels share.
//
// The frames themselves are computed by mfcc_wave_device.h (one frame per wave).  What lives here is the part of
// Listener.update_vectors (/root/reference/precise/network_runner.py:125-146) that is pure integer / byte work:
// which samples are left over after an update, how many frames have been computed and how many of them the
// reference would already have in its feature window.
//   * PCM is read straight from the caller's chunk; only the <= 511 samples of the one frame that straddles two
//     calls are carried in HBM ("carry", int16), and the 36 % of every window that numpy's rfft crop makes dead is
//     never read;
//   * per stream: q = samples held toward the next frame to compute (negative inside the dead zone between
//     windows), kc = frames computed, ke = frames emitted (visible to the network); counters wrap mod 2^32 --
//     one 16-byte record per stream and side (pe_common.h: StreamRec).
#pragma once
#include "pe_common.h"

namespace pe {

template <class R> struct RealK;
template <> struct RealK<double> {
    static constexpr double C1 = 0.92387953251128673848;   // cos(pi/8)
    static constexpr double S1 = 0.38268343236508978178;   // sin(pi/8)
    static constexpr double H = 0.70710678118654752440;    // sqrt(1/2)
    static constexpr double EPS = 2.220446049250313e-16;   // np.finfo(float).eps (sonopy safe_log)
    static constexpr double INV_FFT = 1.0 / 512.0;
    static constexpr double INV_I16 = 1.0 / 32768.0;       // util.py:37
    // int16 samples enter the FFT unscaled; the exact power-of-two factor 2^-30 = INV_I16^2 is folded into the
    // power scale instead (bit-identical: scaling by powers of two commutes with every rounding in between)
    static constexpr double PSCALE_I16 = (1.0 / 512.0) / 1073741824.0;
};
template <> struct RealK<float> {
    static constexpr float C1 = 0.92387953251128673848f;
    static constexpr float S1 = 0.38268343236508978178f;
    static constexpr float H = 0.70710678118654752440f;
    static constexpr float EPS = 2.220446049250313e-16f;
    static constexpr float INV_FFT = 1.0f / 512.0f;
    static constexpr float INV_I16 = 1.0f / 32768.0f;
    static constexpr float PSCALE_I16 = (1.0f / 512.0f) / 1073741824.0f;
};

__device__ __forceinline__ double real_log(double x) { return log(x); }
__device__ __forceinline__ float real_log(float x) { return logf(x); }
__device__ __forceinline__ double real_fma(double a, double b, double c) { return fma(a, b, c); }
__device__ __forceinline__ float real_fma(float a, float b, float c) { return fmaf(a, b, c); }

// LDS instructions of one wave execute in program order, so a hand-off between lanes of the same
// wave needs no counter wait -- only the compiler must not move DS accesses across it.
__device__ __forceinline__ void group_sync() {
#if defined(PE_GROUP_SYNC_WAIT) && PE_GROUP_SYNC_WAIT      // (bisecting aid of round 4: tools/micro/gru_b20_device.h)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
    asm volatile("" ::: "memory");
#endif
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// One 16-lane group per stream (one workgroup of 256 threads = one tile of 16 streams of this launch): after a call that
// appended n_updates chunks of C samples, move the leftover samples to the stream's OTHER side (the frame tasks of the
// same launch still read the current one), advance the counters update by update exactly as single updates would, publish
// them as the other side's record (stamped with this call's number: current from the next call on), and record the
// emitted-frame counter after EVERY update (ke_hist, when given) -- that is what tells a batched network launch which window
// each update saw.  Streams that are not in this launch (pe_update_subset) are not touched.
template <class R>
__device__ __forceinline__ void mfcc_book_tile(const MfccStreamArgs<R>& a, const int tile, const int wave_in_tile = -1) {
    const StreamGeom& geo = a.geo;
    // (wave_in_tile: workgroups of more than four waves keep the books of several tiles, four waves each)
    const int lane = threadIdx.x & 63, wave = wave_in_tile >= 0 ? wave_in_tile : (int)(threadIdx.x >> 6);
    const int grp = lane >> 4, r = lane & 15;
    const int j = wave * 4 + grp;
    const long long s = (long long)tile * kTileStreams + j;       // position in this launch: PCM row
    if (s >= geo.n_streams) return;
    const long long sid = a.ids ? (long long)a.ids[s] : s;        // the stream: state, leftover PCM
    const int C = a.chunk, hop = geo.hop, flen = geo.frame_len, U = a.n_updates;
    const size_t update_stride = (size_t)geo.n_streams * C;
    const RecPair both = rec_request(a.st.rec, a.st.n_padded, sid);
    const int side = rec_side(both, a.st.call);
    const StreamRec now = rec_pick(both, side);
    const int q = now.q;
    const uint32_t kc = now.kc;
    const int avail = q + U * C;
    // (divisions by the hop and the chunk length through the host-made reciprocals: a runtime integer division is ~40
    //  instructions, and this role had six of them per stream)
    auto by_hop = [&](int x) -> int { return (int)a.div_hop.div((uint32_t)x); };
    auto by_chunk = [&](int x) -> int { return (int)a.div_chunk.div((uint32_t)x); };
    const int nnew = avail >= flen ? 1 + by_hop(avail - flen) : 0;
    const size_t carry_side = (size_t)a.st.n_padded * kCarryCap;
    const int16_t* car = a.st.carry + (size_t)side * carry_side + (size_t)sid * kCarryCap;
    const int16_t* base = a.pcm + (size_t)s * C;
    // virtual sample v (0 <= v < avail): carry below q, chunk (v - q) / C above
    auto vsample = [&](int v) -> int {
        if (v < q) return (int)car[v];
        const int w = v - q, u = by_chunk(w);
        return (int)base[(size_t)u * update_stride + (w - u * C)];
    };
    // dword loads of (even, odd) pairs: every quantity that shifts a pair boundary must be even, and the span
    // may cross at most one chunk boundary
    const bool fast = a.pcm_pairs_ok && ((q | hop | C) & 1) == 0 && (C >= flen || U == 1);
    // dwords c = C0 .. C0 + 7 of the lane (sample pairs 32 c + 2 r): unconditional loads from clamped positions -- a load
    // inside an exec-masked block is waited for before the next such block starts
    auto fetch = [&](int vb, int limit, const int c0, int (&dst)[8]) {
        if (fast && ((vb | limit) & 1) == 0) {
            const int w0 = vb - q;                              // < 0: the span starts inside the carry
            int u0 = 0, off0 = w0;
            if (w0 >= 0) { u0 = by_chunk(w0); off0 = w0 - u0 * C; }
            const int16_t* rowu = base + (size_t)u0 * update_stride;
            const ptrdiff_t wrap = (ptrdiff_t)update_stride - C;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int n = 32 * (c0 + c) + 2 * r;
                const int nn = n < limit ? n : 0;
                const int v = vb + nn, off = off0 + nn;
                const int16_t* p = (v < q) ? (car + v) : (rowu + off + (off >= C ? wrap : 0));
                const int val = *reinterpret_cast<const int*>(p);
                dst[c] = n < limit ? val : 0;
            }
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int n = 32 * (c0 + c) + 2 * r;
                const int lo = n < limit ? (vsample(vb + n) & 0xffff) : 0;
                const int hi = n + 1 < limit ? vsample(vb + n + 1) : 0;
                dst[c] = lo | (hi << 16);
            }
        }
    };
    const int qn = avail - nnew * hop;
    int16_t* const carw = a.st.carry + (size_t)(side ^ 1) * carry_side + (size_t)sid * kCarryCap;
    auto put = [&](const int c0, const int (&left)[8]) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int n = 32 * (c0 + c) + 2 * r;
            if (n + 1 < qn) *reinterpret_cast<int*>(carw + n) = left[c];
            else if (n < qn) carw[n] = (int16_t)(left[c] & 0xffff);
        }
    };
    // Single updates whose leftovers lie inside the chunk (every update of 1024-sample chunks: a completed frame puts the next
    // one's start, >= hop samples on, past anything the carry held): eight samples per load.  Lane r of the group moves samples
    // 8 r + 128 c .. + 8 with one 16-byte load (source 4-byte aligned: q and hop are even) and one aligned 16-byte store, and
    // the group's lanes 0 .. 2 the up to three sample pairs behind the last full eight: 5 loads where the dword form below
    // issues 16, and a quarter of its address arithmetic (round 5: the role was 55 of ~500 vector instructions per stream and
    // update of the launch, profiles/round5/r5h_mfcc_quad.log).
    const int vb1 = nnew * hop;
    if (a.keep) {
        // the leftover stays in this call's chunk (the next call's `head`): nothing to move
    } else if (U == 1 && C >= 8 && __all(!fast ? 0 : (qn <= 0 || vb1 >= q))) {
        struct __attribute__((packed, aligned(4))) Pcm8 { int d[4]; };
        const int16_t* src = base + (vb1 - q);                   // sample 0 of the leftover
        const int full8 = qn > 0 ? (qn & ~7) : 0;                // samples covered by whole eights
        const int rest = qn > 0 ? ((qn & 7) >> 1) : 0;           // sample pairs behind them (0 .. 3)
        const int nhalf = __any(qn > 256) ? 2 : 1;
        for (int half = 0; half < nhalf; ++half) {
            Pcm8 v[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int n = 128 * (2 * half + c) + 8 * r;
                v[c] = *reinterpret_cast<const Pcm8*>(n < full8 ? src + n : base);      // (unconditional, from a clamped position)
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int n = 128 * (2 * half + c) + 8 * r;
                if (n < full8) *reinterpret_cast<int4*>(carw + n) = int4{v[c].d[0], v[c].d[1], v[c].d[2], v[c].d[3]};
            }
        }
        if (__any(rest > 0)) {
            const int val = *reinterpret_cast<const int*>(r < rest ? src + full8 + 2 * r : base);
            if (r < rest) *reinterpret_cast<int*>(carw + full8 + 2 * r) = val;
        }
    } else
    // the leftover of a stream is up to 511 samples = 16 dwords per lane of its group, but in most updates it is shorter than
    // 256 samples or empty (q < 0: the next frame starts inside the next chunk): the two halves are wave-uniform branches
    if (__any(qn > 0)) {
        int left[8];
        const int vb = qn > 0 ? nnew * hop : q;                 // (streams with nothing to move read their chunk's first samples and store nothing)
        fetch(vb, qn > 0 ? qn : 0, 0, left);
        put(0, left);
        if (__any(qn > 256)) {
            fetch(vb, qn > 0 ? qn : 0, 8, left);
            put(8, left);
        }
    }
    if (r == 0) {
        int qu = q;
        uint32_t kcu = kc, ke = now.ke;
        for (int u = 0; u < U; ++u) {                          // the counters update by update, as pe_update moves them
            const int av = qu + C;
            const int nn = av >= flen ? 1 + by_hop(av - flen) : 0;
            qu = av - nn * hop;
            kcu += (uint32_t)nn;
            const int m = qu + hop * (int)(kcu - ke);
            if (m >= geo.window) ke += 1u + (uint32_t)by_hop(m - geo.window);
            if (a.ke_hist) a.ke_hist[(size_t)u * a.n_padded + sid] = ke;
        }
        a.st.rec[rec_at(a.st.n_padded, sid, side ^ 1)] = StreamRec{qu, kcu, ke, a.st.call};      // one 16-byte store
    }
}


}  // namespace pe
