// General front end: the MFCC pipeline of /root/reference/precise/vectorization.py:36-39 (sonopy.mfcc_spec; legacy:
// speechpy) for ANY ListenerParams (/root/reference/precise/params.py:28-118) -- n_fft a power of two from 64 to
// 2048, up to 128 filters, up to 32 coefficients -- where the one-frame-per-wave kernel of mfcc_wave_device.h is
// built for the stock shape (512-point transform = 4 points per lane, <= 64 filters, <= 16 coefficients).
//
// Same contract, simpler machine mapping: one frame per wave, everything after the sample fetch in the wave's LDS.
//   * packed real FFT: z[n] = x[2n] + i x[2n+1] (M = n_fft / 2 points), radix-2 decimation in time with the
//     bit-reversal folded into the first store, log2 M stages of M / 2 butterflies spread over the 64 lanes, twiddles
//     W_M^k from a host-built table; then the real split X[k] = E[k] + W_N^k O[k] against Z[M - k];
//   * power = (re^2 + im^2) / n_fft; sparse mel filterbank as host-built CSR (lane f adds the non-zeros of filter f in
//     bin order); log with the vectorizer's zero handling; DCT-II (ortho) as a dense [n_mfcc][n_filt] table; coefficient
//     0 := log of the total power.
// HBM-bound by design like the stock kernel (2 bytes per sample in, n_mfcc floats per frame out); it is the coverage
// path, not the headline path: a wave walks the transform's dependent LDS round trips one frame at a time.
#pragma once
#include "pe_common.h"
#include "mfcc_device.h"
#include "mfcc_wave_device.h"      // wave_sum

namespace pe {

constexpr int kGeneralRun = 16;     // filterbank entries per lane run (mel tables)
constexpr int kGeneralDctCols = 32; // columns of the transposed DCT table (>= n_mfcc)

struct GeneralTables {          // device pointers, built by engine.hip (build_general_tables)
    const void* tw;             // [M / 2] complex R: W_M^k = exp(-2 pi i k / M)
    const void* wn;             // [M / 2 + 1] complex R: W_N^k = exp(-2 pi i k / N)
    // Filterbank as lane RUNS (the stock kernel's scheme, for any matrix): every filter's non-zeros, in bin order, are cut
    // into runs of <= kGeneralRun entries; run r sits in lane r % 64 of round r / 64 (a filter's runs are consecutive).
    // A lane multiplies its run (weights and bins padded with 0 / bin 0), stores ONE partial sum; lane f adds the
    // partial sums of filter f in run order.  All loads of a round are coalesced and in flight together.
    const void* run_w;          // [n_rounds][kGeneralRun][64] R
    const int* run_bin;         // [n_rounds][kGeneralRun][64]
    const int* run_ptr;         // [n_filt + 1] first run of every filter
    const void* dct_t;          // [n_filt][kGeneralDctCols] R: DCT-II (ortho) transposed, row f = the weights of filter f for every coefficient
    int n_fft, log2m, n_filt, n_mfcc, log_mode, n_rounds;
    // n_fft that is not a power of two (np.fft.rfft takes any length, vectorization.py:36-39): Bluestein's chirp-z form
    // of the DFT over a complex transform of length L = 2^log2m >= 2 n_fft - 1 (general_frame_blue); null otherwise
    const void* chirp;          // [n_fft] complex R: exp(-i pi n^2 / n_fft)
    const void* bhat;           // [L] complex R: FFT_L of the wrapped conjugate chirp, / L, in BIT-REVERSED order
};

#ifndef PE_GEN_ABL
#define PE_GEN_ABL 0        // tuning aid (wrong results): 1 no leftover copy, 2 samples = 0 (no PCM loads), 4 no FFT stages, 8 no mel / log / DCT
#endif
constexpr int kGeneralMaxFft = 2048;
constexpr int kGeneralMaxBlueFft = 1024;     // non-power-of-two n_fft: the Bluestein transform of 2048 complex points is what one wave's LDS holds in float64
constexpr int kGeneralMaxFilt = 128;
constexpr int kGeneralMaxMfcc = 32;

// LDS of one wave, in reals: Z[M] complex, P[M + 1], LM[n_filt + 1], PART[64 n_rounds].  Up to n_fft = 1024 the power
// spectrum OVERLAYS the transform (general_frame_t reads all of a lane's Z before anything writes P): 9.2 instead of 13.3 KB
// per wave in float64 at n_fft = 1024, i.e. sixteen instead of twelve waves per compute unit -- 4096 streams in ONE round of
// resident waves instead of two (measured: 47.7 -> us per update).
__host__ __device__ inline bool general_overlay(int n_fft) { return n_fft <= 1024; }
// lengths that take the packed real transform: powers of two from 64 on (a butterfly per lane); 16 and 32 go the Bluestein way
__host__ __device__ inline bool general_is_pow2(int n_fft) { return (n_fft & (n_fft - 1)) == 0 && n_fft >= 64; }
// Bluestein transform length for a non-power-of-two n_fft: the next power of two >= 2 n_fft - 1
__host__ __device__ inline int general_blue_log2(int n_fft) { int b = 7; while ((1 << b) < 2 * n_fft - 1) ++b; return b; }      // (>= 128 points: a butterfly per lane)
__host__ __device__ inline size_t general_lds_bytes(int real_size, int n_fft, int n_filt, int n_rounds) {
    if (!general_is_pow2(n_fft))        // Z[L] complex (the power spectrum overlays it), LM, PART
        return (size_t)real_size * (2 * (1 << general_blue_log2(n_fft)) + (n_filt + 1) + 3 + 64 * n_rounds);
    const int M = n_fft / 2;
    return (size_t)real_size * (2 * M + (general_overlay(n_fft) ? 0 : M + 1) + (n_filt + 1) + 3 + 64 * n_rounds);
}

__device__ __forceinline__ int bit_reverse(int v, int bits) { return (int)(__brev((unsigned)v) >> (32 - bits)); }

// The part of a frame behind the power spectrum: P[0 .. n_fft / 2] in LDS, psum = this lane's share of the total power.
template <class R>
__device__ __forceinline__ void general_frame_tail(const GeneralTables& t, const R* P, R* LM, R* PART, const int lane, R psum, R (&coeff)[1]) {
    using K = RealK<R>;
    group_sync();
    psum = wave_sum(psum);
    auto vlog = [&](R x) -> R {
        // sonopy clips at eps (safe_log); speechpy replaces exact zeros only (zero_handling): 0 < x < eps stays x
        return real_log(t.log_mode == 0 ? (x > K::EPS ? x : K::EPS) : (x == R(0) ? K::EPS : x));
    };
    if (!(PE_GEN_ABL & 8)) {
        // filterbank: every lane one run per round (all table loads of a round first), one partial sum each
        const R* rw = static_cast<const R*>(t.run_w);
        for (int r = 0; r < t.n_rounds; ++r) {
            R acc = R(0);
            constexpr int HB = kGeneralRun / 2;                 // two batches of reads: half the registers in flight
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                R wv[HB];
                int bv[HB];
#pragma unroll
                for (int i = 0; i < HB; ++i) {
                    wv[i] = rw[((size_t)r * kGeneralRun + h * HB + i) * 64 + lane];
                    bv[i] = t.run_bin[((size_t)r * kGeneralRun + h * HB + i) * 64 + lane];
                }
#pragma unroll
                for (int i = 0; i < HB; ++i) acc = real_fma(wv[i], P[bv[i]], acc);
            }
            PART[r * 64 + lane] = acc;
        }
        group_sync();
        for (int f = lane; f < t.n_filt; f += 64) {
            R acc = R(0);
            for (int i = t.run_ptr[f]; i < t.run_ptr[f + 1]; ++i) acc += PART[i];
            LM[f] = vlog(acc);
        }
    }
    if (lane == 0) LM[t.n_filt] = vlog(psum);
    group_sync();
    // DCT-II (ortho): lane c + 32 h adds the filters of half h for coefficient c (transposed table: a row per filter, the
    // 32 coefficients side by side), the halves are added across the wave; coefficient 0 := log of the total power
    R c = R(0);
    if (!(PE_GEN_ABL & 8)) {
        const R* dt = static_cast<const R*>(t.dct_t);
        const int col = lane & 31, hf = lane >> 5, nh = (t.n_filt + 1) >> 1;
        const int f0 = hf * nh, f1 = f0 + nh < t.n_filt ? f0 + nh : t.n_filt;
        int f = f0;
        for (; f + 4 <= f1; f += 4) {
            R dv[4], lv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { dv[i] = dt[(size_t)(f + i) * kGeneralDctCols + col]; lv[i] = LM[f + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i) c = real_fma(dv[i], lv[i], c);
        }
        for (; f < f1; ++f) c = real_fma(dt[(size_t)f * kGeneralDctCols + col], LM[f], c);
        c += __shfl_xor(c, 32);
    }
    if (lane == 0) c = LM[t.n_filt];
    coeff[0] = lane < t.n_mfcc ? c : R(0);
    group_sync();
}

// One frame of M = 2^BITS packed complex points.  point(n) returns point n (samples 2n, 2n + 1 as reals, already scaled,
// zero beyond the frame length) for 0 <= n < M; it is called for all of a lane's points BEFORE any of them is used, so a
// point() made of plain loads keeps them all in flight.
// After the call: coefficient c (c < n_mfcc) sits in lane c of `coeff`; LM[f] holds the log-mel energy of filter f.
template <class R, int BITS, class Point>
__device__ __forceinline__ void general_frame_t(const GeneralTables& t, R* S, const int lane, Point point, R (&coeff)[1]) {
    constexpr int M = 1 << BITS, NP = (M + 63) / 64, NB = (M / 2 + 63) / 64, NS = (M / 2 + 1 + 63) / 64;
    constexpr bool OVERLAY = BITS <= 9;              // (general_overlay: n_fft <= 1024)
    cplx<R>* Z = reinterpret_cast<cplx<R>*>(S);
    R* P = OVERLAY ? S : S + 2 * M;
    R* LM = S + 2 * M + (OVERLAY ? 0 : M + 1);
    R* PART = LM + (t.n_filt + 1) + 3;
    const cplx<R>* tw = static_cast<const cplx<R>*>(t.tw);
    const cplx<R>* wn = static_cast<const cplx<R>*>(t.wn);
    // packed points.  Lane l fetches points n = l + 64 k (coalesced), which in bit-reversed order are the NP CONSECUTIVE
    // elements NP bitrev6(l) + bitrev(k) of Z: the first LS = log2 NP stages of the decimation-in-time transform pair only
    // elements of that group, so they run on the lane's registers (twiddles W_2, W_4, ... : wave-uniform table reads) and
    // the group is stored once -- every stage through LDS moves the whole transform in and out of it, and with sixteen
    // waves per compute unit the LDS pipe, not latency, is what a stage costs.
#ifndef PE_GEN_LOCAL
#define PE_GEN_LOCAL 1
#endif
    constexpr int LS = (PE_GEN_LOCAL && BITS >= 6) ? (BITS - 6 < 3 ? BITS - 6 : 3) : 0;       // (<= 8 points in registers at a time)
    constexpr int GP = 1 << LS, NG = NP / GP, EB = BITS >= 6 ? BITS - 6 - LS : 0;            // points per group, groups per lane
    static_assert(LS == 0 || NG * GP == NP, "points per lane");
    if constexpr (LS == 0) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int n = lane + 64 * k;
            const cplx<R> pt = (PE_GEN_ABL & 2) ? cplx<R>{R(n), R(1)} : point(n < M ? n : M - 1);
            if (n < M) Z[bit_reverse(n, BITS)] = pt;
        }
    } else {
        // group g of the lane = its points k = kk NG + g (kk = 0 .. GP - 1), element bitrev(kk) of the GP consecutive
        // elements from NP bitrev6(l) + bitrev(g) GP on
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            cplx<R> pt[GP];
#pragma unroll
            for (int kk = 0; kk < GP; ++kk) {
                const int n = lane + 64 * (kk * NG + g);
                pt[kk] = (PE_GEN_ABL & 2) ? cplx<R>{R(n), R(1)} : point(n);
            }
            cplx<R> gq[GP];
#pragma unroll
            for (int r = 0; r < GP; ++r) gq[r] = pt[(int)(__builtin_bitreverse32((unsigned)r) >> (32 - LS))];
#pragma unroll
            for (int s = 0; s < ((PE_GEN_ABL & 4) ? 0 : LS); ++s) {
                const int half = 1 << s;
#pragma unroll
                for (int jb = 0; jb < GP / 2; ++jb) {
                    const int pos = jb & (half - 1), i0 = ((jb >> s) << (s + 1)) + pos, i1 = i0 + half;
                    const cplx<R> w = tw[pos * (M >> (s + 1))];
                    const R tr = w.x * gq[i1].x - w.y * gq[i1].y, ti = w.x * gq[i1].y + w.y * gq[i1].x;
                    const cplx<R> av = gq[i0];
                    gq[i0] = cplx<R>{av.x + tr, av.y + ti};
                    gq[i1] = cplx<R>{av.x - tr, av.y - ti};
                }
            }
            const int gr = EB > 0 ? (int)(__builtin_bitreverse32((unsigned)g) >> (32 - (EB > 0 ? EB : 1))) : 0;
            const int base = NP * bit_reverse(lane, 6) + gr * GP;
#pragma unroll
            for (int r = 0; r < GP; ++r) Z[base + r] = gq[r];
        }
    }
    group_sync();
    // radix-2 decimation in time, <= 4 butterflies of a lane at a time.  Up to 1024-point transforms (one batch per stage) the
    // twiddles of a stage are requested one stage ahead (their addresses depend on the lane and the stage only: with the
    // loads inside the stage every stage waited for a cache round trip per butterfly, in series)
    constexpr int BB0 = sizeof(R) == 8 ? 2 : 4, BB = NB < BB0 ? NB : BB0;
    constexpr bool AHEAD = NB <= 4;
    constexpr int NA = AHEAD ? NB : 1;
    auto tw_index = [&](const int s, const int k) -> int {
        const int jb = lane + 64 * k, half = 1 << s, tstep = M >> (s + 1);
        const int jc = jb < M / 2 ? jb : 0;
        return (jc & (half - 1)) * tstep;
    };
    cplx<R> wa[NA];
    if (AHEAD)
#pragma unroll
        for (int k = 0; k < NA; ++k) wa[k] = tw[tw_index(LS, k)];
#pragma unroll 1
    for (int s = LS; s < ((PE_GEN_ABL & 4) ? 0 : BITS); ++s) {      // (not unrolled: every stage's twiddles would be hoisted to the top)
        const int half = 1 << s;
        cplx<R> wnx[NA];
        if (AHEAD) {
            const int sn = s + 1 < BITS ? s + 1 : s;
#pragma unroll
            for (int k = 0; k < NA; ++k) wnx[k] = tw[tw_index(sn, k)];
        }
#pragma unroll
        for (int k0 = 0; k0 < NB; k0 += BB) {
            cplx<R> w[BB], av[BB], bv[BB];
            int i0[BB];
#pragma unroll
            for (int k = 0; k < BB; ++k) {
                const int jb = lane + 64 * (k0 + k), jc = jb < M / 2 ? jb : 0;
                i0[k] = ((jc >> s) << (s + 1)) + (jc & (half - 1));
                w[k] = AHEAD ? wa[AHEAD ? k0 + k : 0] : tw[tw_index(s, k0 + k)];
                av[k] = Z[i0[k]]; bv[k] = Z[i0[k] + half];
            }
#pragma unroll
            for (int k = 0; k < BB; ++k) {
                const R tr = w[k].x * bv[k].x - w[k].y * bv[k].y, ti = w[k].x * bv[k].y + w[k].y * bv[k].x;
                if (lane + 64 * (k0 + k) < M / 2) {
                    Z[i0[k]] = cplx<R>{av[k].x + tr, av[k].y + ti};
                    Z[i0[k] + half] = cplx<R>{av[k].x - tr, av[k].y - ti};
                }
            }
        }
        group_sync();
        if (AHEAD)
#pragma unroll
            for (int k = 0; k < NA; ++k) wa[k] = wnx[k];
    }
    // real split + power spectrum; the total power as per-lane partial sums in bin order, then one wave reduction
    const R inv_n = R(1) / R(t.n_fft);
    R psum = R(0);
    // (OVERLAY: P shares the LDS of Z -- every lane first reads ALL the transform values it needs, then the wave
    //  synchronises, then the powers are written; otherwise batches of 2 / 4 bins per lane)
    constexpr int SB0 = sizeof(R) == 8 ? 2 : 4, SB = OVERLAY ? NS : (NS < SB0 ? NS : SB0);
#pragma unroll
    for (int i0 = 0; i0 < NS; i0 += SB) {
        cplx<R> wk[SB], zkv[SB], zmv[SB];
#pragma unroll
        for (int i = 0; i < SB; ++i) {
            const int k = lane + 64 * (i0 + i), kc = k <= M / 2 ? k : 0;
            wk[i] = wn[kc];
            zkv[i] = Z[kc & (M - 1)];
            zmv[i] = Z[(M - kc) & (M - 1)];
        }
        if (OVERLAY) group_sync();
#pragma unroll
        for (int i = 0; i < SB; ++i) {
            const int k = lane + 64 * (i0 + i);
            if (k > M / 2) continue;
            const cplx<R> zk = zkv[i], zm = zmv[i];
            // X[k] = E + W_N^k O, X[M - k] = conj(E) - conj(W_N^k O) with E = (zk + conj zm) / 2, O = -i (zk - conj zm) / 2
            const R er = R(0.5) * (zk.x + zm.x), ei = R(0.5) * (zk.y - zm.y);
            const R orr = R(0.5) * (zk.y + zm.y), oi = R(-0.5) * (zk.x - zm.x);
            const R wr = wk[i].x * orr - wk[i].y * oi, wi = wk[i].x * oi + wk[i].y * orr;
            const R ar = er + wr, ai = ei + wi;                 // X[k]
            const R br = er - wr, bi = -ei + wi;                // X[M - k]
            const R pa = (ar * ar + ai * ai) * inv_n, pb = (br * br + bi * bi) * inv_n;
            P[k] = pa;
            psum += pa;
            if (k != M - k) { P[M - k] = pb; psum += pb; }
        }
    }
    general_frame_tail<R>(t, P, LM, PART, lane, psum, coeff);
}

// ---- n_fft that is not a power of two: Bluestein -------------------------------------------------------------------------
// X[k] = sum_n x[n] e^(-2 pi i n k / N) = w[k] sum_n (x[n] w[n]) conj(w[k - n]),  w[m] = e^(-i pi m^2 / N): a circular
// convolution of length L = 2^BITS >= 2 N - 1, done with two complex radix-2 transforms in the wave's LDS --
//   a[n] = x[n] w[n] (zero beyond N), natural order  ->  decimation in FREQUENCY (output bit-reversed)
//   ->  pointwise product with the host-made FFT_L(conj-chirp) / L, stored bit-reversed as well
//   ->  decimation in TIME with conjugate twiddles (input bit-reversed, output natural) = the inverse transform
// -- no permutation pass.  |w[k]| = 1, so the power spectrum is |c[k]|^2 / N directly.  sample(n): sample n of the frame as R
// (already scaled; zero beyond the frame length), 0 <= n < N.
template <class R, int BITS, class Sample>
__device__ __forceinline__ void general_frame_blue(const GeneralTables& t, R* S, const int lane, Sample sample, R (&coeff)[1]) {
    constexpr int L = 1 << BITS, NB = L / 2 / 64;
    const int N = t.n_fft, bins = N / 2 + 1;
    cplx<R>* Z = reinterpret_cast<cplx<R>*>(S);
    R* P = S;                                        // overlays Z (all of a lane's c[k] are read before any P is written)
    R* LM = S + 2 * L;
    R* PART = LM + (t.n_filt + 1) + 3;
    const cplx<R>* tw = static_cast<const cplx<R>*>(t.tw);           // W_L^k, k < L / 2
    const cplx<R>* chirp = static_cast<const cplx<R>*>(t.chirp);
    const cplx<R>* bhat = static_cast<const cplx<R>*>(t.bhat);
    for (int n0 = 0; n0 < L; n0 += 256) {
        R xv[4];
        cplx<R> wv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + 64 * i + lane, nc = n < N ? n : 0;
            xv[i] = (PE_GEN_ABL & 2) ? R(1) : sample(nc);
            wv[i] = chirp[nc];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + 64 * i + lane;
            if (n < L) Z[n] = n < N ? cplx<R>{xv[i] * wv[i].x, xv[i] * wv[i].y} : cplx<R>{R(0), R(0)};
        }
    }
    group_sync();
    // one radix-2 stage over Z: DIF (a + b, (a - b) w) or DIT with conjugate twiddles (a + b w*, a - b w*)
    auto stage = [&](const int s, const bool dif) {
        const int half = 1 << s, tstep = L >> (s + 1);
#pragma unroll 2
        for (int k = 0; k < NB; ++k) {
            const int jb = lane + 64 * k, pos = jb & (half - 1), i0 = ((jb >> s) << (s + 1)) + pos;
            const cplx<R> w = tw[pos * tstep], av = Z[i0], bv = Z[i0 + half];
            if (dif) {
                const R dr = av.x - bv.x, di = av.y - bv.y;
                Z[i0] = cplx<R>{av.x + bv.x, av.y + bv.y};
                Z[i0 + half] = cplx<R>{dr * w.x - di * w.y, dr * w.y + di * w.x};
            } else {
                const R tr = w.x * bv.x + w.y * bv.y, ti = w.x * bv.y - w.y * bv.x;       // b conj(w)
                Z[i0] = cplx<R>{av.x + tr, av.y + ti};
                Z[i0 + half] = cplx<R>{av.x - tr, av.y - ti};
            }
        }
        group_sync();
    };
    if (!(PE_GEN_ABL & 4)) {
#pragma unroll 1
        for (int s = BITS - 1; s >= 0; --s) stage(s, true);
        for (int p0 = 0; p0 < L; p0 += 256) {
            cplx<R> bh[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) bh[i] = bhat[p0 + 64 * i + lane];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int p = p0 + 64 * i + lane;
                const cplx<R> av = Z[p];
                Z[p] = cplx<R>{av.x * bh[i].x - av.y * bh[i].y, av.x * bh[i].y + av.y * bh[i].x};
            }
        }
        group_sync();
#pragma unroll 1
        for (int s = 0; s < BITS; ++s) stage(s, false);
    }
    // power spectrum: every lane reads its c[k] first (P overlays Z), then the wave synchronises, then P is written
    const R inv_n = R(1) / R(N);
    R psum = R(0);
    constexpr int NSB = (L / 4 + 1 + 63) / 64;          // bins <= N / 2 + 1 <= L / 4 + 1
    cplx<R> cv[NSB];
#pragma unroll
    for (int i = 0; i < NSB; ++i) { const int k = lane + 64 * i; cv[i] = Z[k < bins ? k : 0]; }
    group_sync();
#pragma unroll
    for (int i = 0; i < NSB; ++i) {
        const int k = lane + 64 * i;
        if (k < bins) {
            const R pk = (cv[i].x * cv[i].x + cv[i].y * cv[i].y) * inv_n;
            P[k] = pk;
            // total power as np.sum over the rfft bins does it: every bin once
            psum += pk;
        }
    }
    general_frame_tail<R>(t, P, LM, PART, lane, psum, coeff);
}

// (the transform length is a template parameter of the frame -- unrolled per-lane loops -- and of the kernels around it:
//  the launchers dispatch on log2(n_fft / 2), kernels.hip)
template <class R, int BITS, class Point>
__device__ __forceinline__ void general_frame(const GeneralTables& t, R* S, const int lane, Point point, R (&coeff)[1]) {
    general_frame_t<R, BITS>(t, S, lane, point, coeff);
}

// ---- streaming: one wave per stream, every frame the update completes ---------------------------------------------------
template <class R>
struct GeneralStreamArgs {
    StreamGeom geo;
    GeneralTables tab;
    const int16_t* pcm;         // [n_streams][chunk]
    int chunk;
    int pcm_pairs_ok;           // chunk even and pcm 4-byte aligned: int16 pairs may be loaded as one dword
    const int32_t* ids;         // row v of pcm belongs to stream ids[v] (pe_update_subset), or to stream v (null)
    StreamState st;             // records and leftover PCM, two sides per stream (pe_common.h: StreamRec); carry rows of carry_cap samples
    int carry_cap;
    float* ring;                // [tiles][slots][16 streams][row_floats] -- or, ring_bf16, 16 bf16 per row (row_floats == 16 only)
    int row_floats;
    int ring_bf16;
    // several updates per launch (pe_update_many): chunk u of stream s at pcm + (u * n_streams + s) * chunk
    int n_updates;
    FastDiv div_chunk;          // division by the chunk length
    uint32_t* ke_hist;          // [n_updates][n_padded] emitted-frame counter after every update, or null
    int n_padded;
};

// Two waves per stream: wave `par` takes the due frames kb = first + par, first + par + 2, ... (1024-sample chunks complete one
// or two frames per update: the second frame of a stream no longer waits for its first), wave 0 also moves the leftover
// samples and the counters.  Both read the state before the update; only wave 0 writes the state after it.
// n_updates > 1 (pe_update_many): the frames of the virtual stream carry ++ chunk 0 ++ ... ++ chunk n-1 in ONE launch, the
// emitted-frame counter after every update recorded for the network launch that follows (round 5; it was one launch per update).
template <class R, int BITS, bool BLUE = false>
__device__ __forceinline__ void general_stream(const GeneralStreamArgs<R>& a, R* S, const int s, const int par, const int lane, const int n_par = 2) {
    const StreamGeom& geo = a.geo;
    const int C = a.chunk, hop = geo.hop, flen = geo.frame_len, slots = geo.ring_slots;
    const int sid = a.ids ? a.ids[s] : s;                 // s: position in this launch (PCM row); sid: the stream (state, ring rows)
    const RecPair both = rec_request(a.st.rec, a.st.n_padded, sid);
    const int side = rec_side(both, a.st.call);
    const StreamRec now = rec_pick(both, side);
    const int q = now.q;
    const uint32_t kc = now.kc;
    uint32_t ke = now.ke;
    const int U = a.n_updates;
    const int avail = q + U * C;
    const int nnew = avail >= flen ? 1 + (avail - flen) / hop : 0;
    const size_t carry_side = (size_t)a.st.n_padded * a.carry_cap;
    const int16_t* car = a.st.carry + (size_t)side * carry_side + (size_t)sid * a.carry_cap;
    const int16_t* row0 = a.pcm + (size_t)s * C;
    const size_t update_stride = (size_t)geo.n_streams * C;
    // sample w >= 0 of the call's chunks of this stream, one after the other (pe_update_many: chunk u = w / C, a row of its own)
    auto row = [&](int w) -> const int16_t* {
        if (U == 1) return row0 + w;
        const int u = (int)a.div_chunk.div((uint32_t)w);
        return row0 + (size_t)u * update_stride + (w - u * C);
    };
    auto vsample = [&](int v) -> int { return v < q ? (int)car[v] : (int)*row(v - q); };      // (q < 0: all of it in the chunks)
    // (even, odd) sample pairs as one dword: every quantity that shifts a pair boundary must be even (then a pair never
    // straddles the carry / chunk seam, and every pair address is 4-byte aligned: carry rows are 128-byte aligned)
    const bool pairs = a.pcm_pairs_ok && ((q | hop | flen) & 1) == 0;
    const int tile = sid >> 4, j = sid & 15;
    for (int kb = (nnew > slots ? nnew - slots : 0) + par; kb < nnew; kb += n_par) {
        const int vb = kb * hop;
        R coeff[1];
        // (the lane-dependent invariants of a frame -- bit-reversed addresses, table offsets -- are recomputed per frame: hoisted
        //  out of this one-or-two-trip loop they lived in registers the frame itself needs, and the n_fft = 1024 float64 kernel
        //  spilled 80 bytes per lane to scratch, reloaded with 26 loads per frame)
        int lane_k = lane;
        asm volatile("" : "+v"(lane_k));
        if constexpr (BLUE) {
            general_frame_blue<R, BITS>(a.tab, S, lane_k, [&](int n) -> R { return n < flen ? (R)vsample(vb + n) * RealK<R>::INV_I16 : R(0); }, coeff);
        } else if (pairs) {
            general_frame<R, BITS>(a.tab, S, lane_k, [&](int n) -> cplx<R> {
                const int m = 2 * n < flen ? 2 * n : 0, v = vb + m;           // (beyond the frame: a valid pair, zeroed below)
                const int16_t* p = v < q ? car + v : row(v - q);
                const int w2 = *reinterpret_cast<const int*>(p);
                const bool in = 2 * n < flen;
                return cplx<R>{in ? (R)(int)(short)(w2 & 0xffff) * RealK<R>::INV_I16 : R(0), in ? (R)(w2 >> 16) * RealK<R>::INV_I16 : R(0)};
            }, coeff);
        } else {
            general_frame<R, BITS>(a.tab, S, lane_k, [&](int n) -> cplx<R> {
                return cplx<R>{2 * n < flen ? (R)vsample(vb + 2 * n) * RealK<R>::INV_I16 : R(0),
                               2 * n + 1 < flen ? (R)vsample(vb + 2 * n + 1) * RealK<R>::INV_I16 : R(0)};
            }, coeff);
        }
        const int slot = (int)((kc + (uint32_t)kb) & (uint32_t)(slots - 1));
        const size_t cell = ((size_t)tile * slots + slot) * kTileStreams + j;
        const float xf = lane < geo.n_mfcc ? (float)coeff[0] : 0.0f;
        if (a.ring_bf16) {          // (rounded where it is stored, to nearest even: what the bf16 network does to a float32 row at the load)
            if (lane < kRowFloats) reinterpret_cast<__bf16*>(a.ring)[cell * kRowFloats + lane] = (__bf16)xf;
        } else if (lane < a.row_floats) a.ring[cell * a.row_floats + lane] = xf;
    }
    if (par != 0) return;
    // leftover samples, counters (the arithmetic of mfcc_book_tile)
    const int qn = avail - nnew * hop;
    int16_t* carw = a.st.carry + (size_t)(side ^ 1) * carry_side + (size_t)sid * a.carry_cap;
    if (!(PE_GEN_ABL & 1)) {
        if (pairs) {
            // qn is even here (q, C, hop even): dword loads, four in flight, dword stores
            const int nd = qn > 0 ? qn >> 1 : 0, vb = nnew * hop;
            for (int d0 = 0; d0 < nd; d0 += 256) {
                int w2[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int d = d0 + 64 * i + lane, dc = d < nd ? d : 0, v = vb + 2 * dc;
                    w2[i] = *reinterpret_cast<const int*>(v < q ? car + v : row(v - q));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int d = d0 + 64 * i + lane;
                    if (d < nd) reinterpret_cast<int*>(carw)[d] = w2[i];
                }
            }
        } else {
            for (int m = lane; m < qn; m += 64) carw[m] = (int16_t)vsample(nnew * hop + m);
        }
    }
    if (lane == 0) {
        int qu = q;
        uint32_t kcu = kc;
        for (int u = 0; u < U; ++u) {                   // the counters update by update, as single updates move them (mfcc_book_tile)
            const int av = qu + C;
            const int nn = av >= flen ? 1 + (av - flen) / hop : 0;
            qu = av - nn * hop;
            kcu += (uint32_t)nn;
            const int mm = qu + hop * (int)(kcu - ke);
            if (mm >= geo.window) ke += 1u + (uint32_t)((mm - geo.window) / hop);
            if (a.ke_hist) a.ke_hist[(size_t)u * a.n_padded + sid] = ke;
        }
        a.st.rec[rec_at(a.st.n_padded, sid, side ^ 1)] = StreamRec{qu, kcu, ke, a.st.call};
    }
}

// ---- stateless whole-buffer form (vectorize_raw / pe_evaluate): one frame per wave, float64 samples in -----------------
template <class R>
struct GeneralOfflineArgs {
    StreamGeom geo;
    GeneralTables tab;
    const double* audio;
    long long n_frames;
    double* out;                // [n_frames][n_mfcc] float64, may be null
    float* out_rows;            // [n_frames][row_floats] float32 rows, may be null
    double* out_mels;           // [n_frames][n_filt] log-mel energies, may be null
    int row_floats;
};

template <class R, int BITS, bool BLUE = false>
__device__ __forceinline__ void general_offline(const GeneralOfflineArgs<R>& a, R* S, const long long first, const long long stride, const int lane) {
    const StreamGeom& geo = a.geo;
    const int M = BLUE ? (1 << BITS) : (a.tab.n_fft >> 1);          // (BLUE: reals before LM = 2 L)
    for (long long fr = first; fr < a.n_frames; fr += stride) {
        const double* x = a.audio + fr * geo.hop;
        R coeff[1];
        const int flen = geo.frame_len;
        int lane_k = lane;                                  // (per-frame recomputation of the lane invariants: general_stream)
        asm volatile("" : "+v"(lane_k));
        if constexpr (BLUE) general_frame_blue<R, BITS>(a.tab, S, lane_k, [&](int n) -> R { return n < flen ? (R)x[n] : R(0); }, coeff);
        else general_frame<R, BITS>(a.tab, S, lane_k, [&](int n) -> cplx<R> {
            const int m0 = 2 * n, m1 = 2 * n + 1;                             // (clamped indices: plain loads, selected afterwards)
            const double x0 = x[m0 < flen ? m0 : 0], x1 = x[m1 < flen ? m1 : 0];
            return cplx<R>{m0 < flen ? (R)x0 : R(0), m1 < flen ? (R)x1 : R(0)};
        }, coeff);
        if (a.out && lane < geo.n_mfcc) a.out[fr * geo.n_mfcc + lane] = (double)coeff[0];
        if (a.out_rows && lane < a.row_floats) a.out_rows[fr * a.row_floats + lane] = lane < geo.n_mfcc ? (float)coeff[0] : 0.0f;
        if (a.out_mels) {
            const R* LM = S + 2 * M + ((BLUE || general_overlay(a.tab.n_fft)) ? 0 : M + 1);
            for (int f = lane; f < geo.n_filt; f += 64) a.out_mels[fr * geo.n_filt + f] = (double)LM[f];
        }
        group_sync();
    }
}

// launchers (kernels.hip)
hipError_t launch_general_stream_f64(const GeneralStreamArgs<double>& a, hipStream_t s);
hipError_t launch_general_stream_f32(const GeneralStreamArgs<float>& a, hipStream_t s);
hipError_t launch_general_offline_f64(const GeneralOfflineArgs<double>& a, int n_cus, hipStream_t s);
hipError_t launch_general_offline_f32(const GeneralOfflineArgs<float>& a, int n_cus, hipStream_t s);

}  // namespace pe
