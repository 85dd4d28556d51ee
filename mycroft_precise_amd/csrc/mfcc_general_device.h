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

struct GeneralTables {          // device pointers, built by engine.hip (build_general_tables)
    const void* tw;             // [M / 2] complex R: W_M^k = exp(-2 pi i k / M)
    const void* wn;             // [M / 2 + 1] complex R: W_N^k = exp(-2 pi i k / N)
    const int* mel_ptr;         // [n_filt + 1] CSR rows of the filterbank
    const int* mel_bin;         // [nnz]
    const void* mel_w;          // [nnz] R
    const void* dct;            // [n_mfcc][n_filt] R, DCT-II with norm='ortho'
    int n_fft, log2m, n_filt, n_mfcc, log_mode;
};

constexpr int kGeneralMaxFft = 2048;
constexpr int kGeneralMaxFilt = 128;
constexpr int kGeneralMaxMfcc = 32;

// LDS of one wave, in reals: Z[M] complex, P[M + 1], LM[n_filt + 1]
__host__ __device__ inline size_t general_lds_bytes(int real_size, int n_fft, int n_filt) {
    const int M = n_fft / 2;
    return (size_t)real_size * (2 * M + (M + 1) + (n_filt + 1) + 3);
}

__device__ __forceinline__ int bit_reverse(int v, int bits) { return (int)(__brev((unsigned)v) >> (32 - bits)); }

// One frame.  sample(v) returns sample v of the frame (0 <= v < flen) as R, already scaled; samples beyond flen are 0.
// After the call: coefficient c (c < n_mfcc) is returned in lane c % 64 slot c / 64 of `coeff`; LM[f] holds the log-mel
// energy of filter f.
template <class R, class Sample>
__device__ __forceinline__ void general_frame(const GeneralTables& t, R* S, const int lane, const int flen, Sample sample, R (&coeff)[1]) {
    using K = RealK<R>;
    const int M = t.n_fft >> 1, bits = t.log2m;
    cplx<R>* Z = reinterpret_cast<cplx<R>*>(S);
    R* P = S + 2 * M;
    R* LM = P + (M + 1);
    const cplx<R>* tw = static_cast<const cplx<R>*>(t.tw);
    const cplx<R>* wn = static_cast<const cplx<R>*>(t.wn);
    // packed points, bit-reversed store
    for (int n = lane; n < M; n += 64) {
        const R re = 2 * n < flen ? sample(2 * n) : R(0);
        const R im = 2 * n + 1 < flen ? sample(2 * n + 1) : R(0);
        Z[bit_reverse(n, bits)] = cplx<R>{re, im};
    }
    group_sync();
    for (int s = 0; s < bits; ++s) {
        const int half = 1 << s, tstep = M >> (s + 1);
        for (int jb = lane; jb < (M >> 1); jb += 64) {
            const int pos = jb & (half - 1), i0 = ((jb >> s) << (s + 1)) + pos, i1 = i0 + half;
            const cplx<R> w = tw[pos * tstep], a = Z[i0], b = Z[i1];
            const R tr = w.x * b.x - w.y * b.y, ti = w.x * b.y + w.y * b.x;
            Z[i0] = cplx<R>{a.x + tr, a.y + ti};
            Z[i1] = cplx<R>{a.x - tr, a.y - ti};
        }
        group_sync();
    }
    // real split + power spectrum; the total power as per-lane partial sums in bin order, then one wave reduction
    const R inv_n = R(1) / R(t.n_fft);
    R psum = R(0);
    for (int k = lane; k <= (M >> 1); k += 64) {
        const int km = (M - k) & (M - 1);
        const cplx<R> zk = Z[k & (M - 1)], zm = Z[km];
        // X[k] = E + W_N^k O, X[M - k] = conj(E) - conj(W_N^k O) with E = (zk + conj zm) / 2, O = -i (zk - conj zm) / 2
        const R er = R(0.5) * (zk.x + zm.x), ei = R(0.5) * (zk.y - zm.y);
        const R orr = R(0.5) * (zk.y + zm.y), oi = R(-0.5) * (zk.x - zm.x);
        const cplx<R> w = wn[k];
        const R wr = w.x * orr - w.y * oi, wi = w.x * oi + w.y * orr;
        const R ar = er + wr, ai = ei + wi;                 // X[k]
        const R br = er - wr, bi = -ei + wi;                // X[M - k]
        const R pa = (ar * ar + ai * ai) * inv_n, pb = (br * br + bi * bi) * inv_n;
        P[k] = pa;
        psum += pa;
        if (k != M - k) { P[M - k] = pb; psum += pb; }
    }
    group_sync();
    psum = wave_sum(psum);
    auto vlog = [&](R x) -> R {
        // sonopy clips at eps (safe_log); speechpy replaces exact zeros only (zero_handling): 0 < x < eps stays x
        return real_log(t.log_mode == 0 ? (x > K::EPS ? x : K::EPS) : (x == R(0) ? K::EPS : x));
    };
    const R* mw = static_cast<const R*>(t.mel_w);
    for (int f = lane; f < t.n_filt; f += 64) {
        R acc = R(0);
        for (int i = t.mel_ptr[f]; i < t.mel_ptr[f + 1]; ++i) acc = real_fma(mw[i], P[t.mel_bin[i]], acc);
        LM[f] = vlog(acc);
    }
    if (lane == 0) LM[t.n_filt] = vlog(psum);
    group_sync();
    const R* dct = static_cast<const R*>(t.dct);
    R c = R(0);
    if (lane < t.n_mfcc) {
        if (lane == 0) c = LM[t.n_filt];                       // coefficient 0 := log of the total power
        else for (int f = 0; f < t.n_filt; ++f) c = real_fma(dct[lane * t.n_filt + f], LM[f], c);
    }
    coeff[0] = c;
    group_sync();
}

// ---- streaming: one wave per stream, every frame the update completes ---------------------------------------------------
template <class R>
struct GeneralStreamArgs {
    StreamGeom geo;
    GeneralTables tab;
    const int16_t* pcm;         // [n_streams][chunk]
    int chunk;
    const int16_t* carry;       // [n_streams_padded][carry_cap] leftover before the update
    int16_t* carry_next;        // ... and after it (must not alias)
    int carry_cap;
    const int32_t* st_q; const uint32_t* st_kc; const uint32_t* st_ke;
    int32_t* st_q_next; uint32_t* st_kc_next; uint32_t* st_ke_next;
    float* ring;                // [tiles][slots][16 streams][row_floats]
    int row_floats;
};

template <class R>
__device__ __forceinline__ void general_stream(const GeneralStreamArgs<R>& a, R* S, const int s, const int lane) {
    const StreamGeom& geo = a.geo;
    const int C = a.chunk, hop = geo.hop, flen = geo.frame_len, slots = geo.ring_slots;
    const int q = a.st_q[s];
    const uint32_t kc = a.st_kc[s];
    uint32_t ke = a.st_ke[s];
    const int avail = q + C;
    const int nnew = avail >= flen ? 1 + (avail - flen) / hop : 0;
    const int16_t* car = a.carry + (size_t)s * a.carry_cap;
    const int16_t* row = a.pcm + (size_t)s * C;
    auto vsample = [&](int v) -> int { return v < q ? (int)car[v] : (int)row[v - q]; };      // (q < 0: all of it in the chunk)
    const int tile = s >> 4, j = s & 15;
    for (int kb = nnew > slots ? nnew - slots : 0; kb < nnew; ++kb) {
        const int vb = kb * hop;
        R coeff[1];
        general_frame<R>(a.tab, S, lane, flen, [&](int m) -> R { return (R)vsample(vb + m) * RealK<R>::INV_I16; }, coeff);
        const int slot = (int)((kc + (uint32_t)kb) & (uint32_t)(slots - 1));
        float* out = a.ring + (((size_t)tile * slots + slot) * kTileStreams + j) * a.row_floats;
        if (lane < a.row_floats) out[lane] = lane < geo.n_mfcc ? (float)coeff[0] : 0.0f;
    }
    // leftover samples, counters (the arithmetic of mfcc_book_tile)
    const int qn = avail - nnew * hop;
    int16_t* carw = a.carry_next + (size_t)s * a.carry_cap;
    for (int m = lane; m < qn; m += 64) carw[m] = (int16_t)vsample(nnew * hop + m);
    if (lane == 0) {
        const uint32_t kcn = kc + (uint32_t)nnew;
        const int mm = qn + hop * (int)(kcn - ke);
        if (mm >= geo.window) ke += 1u + (uint32_t)((mm - geo.window) / hop);
        a.st_q_next[s] = qn; a.st_kc_next[s] = kcn; a.st_ke_next[s] = ke;
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

template <class R>
__device__ __forceinline__ void general_offline(const GeneralOfflineArgs<R>& a, R* S, const long long first, const long long stride, const int lane) {
    const StreamGeom& geo = a.geo;
    const int M = a.tab.n_fft >> 1;
    for (long long fr = first; fr < a.n_frames; fr += stride) {
        const double* x = a.audio + fr * geo.hop;
        R coeff[1];
        general_frame<R>(a.tab, S, lane, geo.frame_len, [&](int m) -> R { return (R)x[m]; }, coeff);
        if (a.out && lane < geo.n_mfcc) a.out[fr * geo.n_mfcc + lane] = (double)coeff[0];
        if (a.out_rows && lane < a.row_floats) a.out_rows[fr * a.row_floats + lane] = lane < geo.n_mfcc ? (float)coeff[0] : 0.0f;
        if (a.out_mels) {
            const R* LM = S + 2 * M + (M + 1);
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
