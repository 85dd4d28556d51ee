// MFCC front end for gfx950 (MI355X): int16 PCM -> 13 coefficients per frame.
//
// What it computes is the reference's Vectorizer.mfccs entry
// (/root/reference/precise/vectorization.py:36-39 -> third-party sonopy.mfcc_spec) as driven by
// Listener.update_vectors (/root/reference/precise/network_runner.py:125-146):
//   frame = first n_fft(512) samples of each 1600-sample window (numpy's rfft(n=512) crop),
//   512-point real FFT -> power/512 -> 20 triangular mel filters -> log(clip(eps)) -> DCT-II
//   ortho, keep 13 -> coefficient 0 replaced by log(clip(sum power)).
//
// Mapping to the machine (64-lane waves, LDS, no MFMA here: this stage is HBM/VALU work):
//   * one 16-lane group per stream, four streams per wave, sixteen streams (= one GRU tile) per
//     256-thread workgroup, so a workgroup owns one [16 streams] feature-ring tile;
//   * the 512 real samples are packed as 256 complex points z[n] = x[2n] + i x[2n+1];
//     256 = 16 x 16: every lane runs a 16-point FFT entirely in registers, one padded LDS
//     transpose, a second in-register 16-point FFT, then the real-FFT split with the mirror
//     lane's upper half exchanged through LDS;
//   * twiddles, mel weights (sparse, 455 non-zeros) and the 13x20 DCT live in LDS;
//   * PCM is read straight from the caller's chunk (64-byte runs per group and instruction,
//     adjacent instructions complete the 128-byte lines); only the <= 511 samples of a frame
//     that straddles two chunks are carried in HBM, and the 36 % of every window that the crop
//     makes dead is never read;
//   * arithmetic type R is double (what the reference computes in) or float.
#pragma once
#include "pe_common.h"

namespace pe {

// Section timers for the tuning harness (tools/mfcc_sections.py builds a -DPE_SECTION_TIMERS copy of
// the library); compiled out of the product.
#ifdef PE_SECTION_TIMERS
__device__ unsigned long long pe_dbg_timers[32];
#define PE_T(i) do { if (threadIdx.x == 0 && blockIdx.x + 1 == gridDim.x) pe_dbg_timers[i] = clock64(); } while (0)
#else
#define PE_T(i) do { } while (0)
#endif

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
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// (a + ib) *= W16^M,  W16 = exp(-2 pi i / 16);  only the exponents a 4x4 split needs.
template <int M, class R>
__device__ __forceinline__ void mul_w16(R& a, R& b) {
    using K = RealK<R>;
    if constexpr (M == 0) {
    } else if constexpr (M == 4) {          // -i
        R t = a; a = b; b = -t;
    } else if constexpr (M == 2) {          // H - iH
        R t = (a + b) * K::H; b = (b - a) * K::H; a = t;
    } else if constexpr (M == 6) {          // -H - iH
        R t = (b - a) * K::H; b = -(a + b) * K::H; a = t;
    } else if constexpr (M == 1) {          // C1 - iS1
        R t = a * K::C1 + b * K::S1; b = b * K::C1 - a * K::S1; a = t;
    } else if constexpr (M == 3) {          // S1 - iC1
        R t = a * K::S1 + b * K::C1; b = b * K::S1 - a * K::C1; a = t;
    } else {                                // M == 9: -C1 + iS1
        static_assert(M == 9, "unexpected W16 exponent");
        R t = -(a * K::C1) - b * K::S1; b = a * K::S1 - b * K::C1; a = t;
    }
}

// y_k = sum_n x_n (-i)^(n k), in place
template <class R>
__device__ __forceinline__ void radix4(R& r0, R& i0, R& r1, R& i1, R& r2, R& i2, R& r3, R& i3) {
    const R t0r = r0 + r2, t0i = i0 + i2, t1r = r0 - r2, t1i = i0 - i2;
    const R t2r = r1 + r3, t2i = i1 + i3, t3r = r1 - r3, t3i = i1 - i3;
    r0 = t0r + t2r; i0 = t0i + t2i;
    r2 = t0r - t2r; i2 = t0i - t2i;
    r1 = t1r + t3i; i1 = t1i - t3r;
    r3 = t1r - t3i; i3 = t1i + t3r;
}

// 16-point forward DFT of (re, im), natural order in, natural order out, all in registers.
// n = 4 n1 + n2, k = k1 + 4 k2:  X[k] = sum_n2 W4^(n2 k2) W16^(n2 k1) sum_n1 x[4 n1 + n2] W4^(n1 k1)
template <class R>
__device__ __forceinline__ void fft16(R (&re)[16], R (&im)[16]) {
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2)
        radix4(re[n2], im[n2], re[4 + n2], im[4 + n2], re[8 + n2], im[8 + n2], re[12 + n2], im[12 + n2]);
    // element 4*k1 + n2 now holds A[n2][k1]; twiddle by W16^(n2*k1)
    mul_w16<1>(re[5], im[5]);   mul_w16<2>(re[9], im[9]);   mul_w16<3>(re[13], im[13]);
    mul_w16<2>(re[6], im[6]);   mul_w16<4>(re[10], im[10]); mul_w16<6>(re[14], im[14]);
    mul_w16<3>(re[7], im[7]);   mul_w16<6>(re[11], im[11]); mul_w16<9>(re[15], im[15]);
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1)
        radix4(re[4 * k1], im[4 * k1], re[4 * k1 + 1], im[4 * k1 + 1], re[4 * k1 + 2], im[4 * k1 + 2],
               re[4 * k1 + 3], im[4 * k1 + 3]);
    // element 4*k1 + k2 holds X[k1 + 4*k2]: rename registers into natural order
    R tr[16], ti[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) { tr[(e >> 2) + 4 * (e & 3)] = re[e]; ti[(e >> 2) + 4 * (e & 3)] = im[e]; }
#pragma unroll
    for (int e = 0; e < 16; ++e) { re[e] = tr[e]; im[e] = ti[e]; }
}

// Per-workgroup LDS image of the constant tables.
template <class R>
struct LdsTab {
    const cplx<R>* tw256;
    const cplx<R>* w512;
    const R* dct;
    const R* mel_w;         // [2][17][16]
    const int* mel_flush;   // [17][16]
    const int* mel_pstart;  // [n_filt + 1]
    int spare;              // partial-sum slot nobody reads
    int group_reals;        // LDS reals per 16-lane group
};

// Copy the blob with 16-byte loads, in two halves so that a kernel can put the loads at its very top (they
// then return first: vmcnt retires in order) and the LDS stores after its other, longer-latency loads have
// been issued.  The caller issues __syncthreads() when it needs the tables.
struct TabRegs { uint4 v0, v1, v2, v3; };

template <class R>
__device__ __forceinline__ TabRegs lds_issue(const MfccTables<R>& g) {
    const int n16 = g.blob_bytes >> 4;
    const uint4* src = reinterpret_cast<const uint4*>(g.blob);
    const int tid = threadIdx.x;
    TabRegs t;
    t.v0 = t.v1 = t.v2 = t.v3 = uint4{0, 0, 0, 0};
    if (tid < n16) t.v0 = src[tid];
    if (tid + 256 < n16) t.v1 = src[tid + 256];
    if (tid + 512 < n16) t.v2 = src[tid + 512];
    if (tid + 768 < n16) t.v3 = src[tid + 768];
    return t;
}

template <class R>
__device__ __forceinline__ R* lds_commit(unsigned char* smem, const MfccTables<R>& g, const TabRegs& v, int n_filt, int n_mfcc,
                                         LdsTab<R>& t) {
    const int n16 = g.blob_bytes >> 4;
    const uint4* src = reinterpret_cast<const uint4*>(g.blob);
    uint4* dst = reinterpret_cast<uint4*>(smem);
    const int tid = threadIdx.x;
    if (tid < n16) dst[tid] = v.v0;
    if (tid + 256 < n16) dst[tid + 256] = v.v1;
    if (tid + 512 < n16) dst[tid + 512] = v.v2;
    if (tid + 768 < n16) dst[tid + 768] = v.v3;
    for (int i = tid + 1024; i < n16; i += 256) dst[i] = src[i];
    cplx<R>* tw = reinterpret_cast<cplx<R>*>(smem);
    cplx<R>* w5 = tw + 256;
    R* dct = reinterpret_cast<R*>(w5 + 130);
    R* mw = dct + n_mfcc * n_filt;
    size_t off = (size_t)((unsigned char*)(mw + 2 * kMelSteps * 16) - smem);
    off = (off + 15) & ~(size_t)15;
    int* fl = reinterpret_cast<int*>(smem + off);
    t.spare = g.mel_parts; t.group_reals = group_scratch_reals(g.mel_parts);
    t.tw256 = tw; t.w512 = w5; t.dct = dct; t.mel_w = mw; t.mel_flush = fl; t.mel_pstart = fl + kMelSteps * 16;
    return reinterpret_cast<R*>(smem + g.blob_bytes);
}

template <class R>
__device__ __forceinline__ R* lds_setup(unsigned char* smem, const MfccTables<R>& g, int n_filt, int n_mfcc, LdsTab<R>& t) {
    const TabRegs v = lds_issue<R>(g);
    return lds_commit<R>(smem, g, v, n_filt, n_mfcc, t);
}

// One frame on one 16-lane group.  `load(c, xr, xi)` returns samples 32c+2r and 32c+2r+1 of the
// frame (already scaled to [-1,1), zero beyond frame_len; typically from registers loaded earlier).
// Returns coefficient r in lane r (lanes >= n_mfcc return 0).  S: this group's LDS scratch.
template <class R, class Load>
__device__ __forceinline__ R mfcc_frame(const LdsTab<R>& t, R* S, int r, int n_filt, int n_mfcc, Load load,
                                        const R pscale = RealK<R>::INV_FFT, const int log_mode = 0) {
    using K = RealK<R>;
    R re[16], im[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) load(c, re[c], im[c]);
    PE_T(2);

    // pass 1: lane r transforms z[16c + r] over c -> Y_r[k1]; twiddle by W256^(r k1)
#ifndef PE_ABL_FFT
    fft16(re, im);
#endif
#ifndef PE_ABL_TWIDDLE
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) {
        const cplx<R> w = t.tw256[k1 * 16 + r];
        const R a = re[k1], b = im[k1];
        re[k1] = a * w.x - b * w.y;
        im[k1] = a * w.y + b * w.x;
    }
#endif
    PE_T(3);
    // 16x16 transpose through LDS (row stride 17 reals: conflict-free both ways), re then im
#ifndef PE_ABL_TRANSPOSE
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) S[k1 * kTrStride + r] = re[k1];
    group_sync();
#pragma unroll
    for (int c = 0; c < 16; ++c) re[c] = S[r * kTrStride + c];
    group_sync();
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) S[k1 * kTrStride + r] = im[k1];
    group_sync();
#pragma unroll
    for (int c = 0; c < 16; ++c) im[c] = S[r * kTrStride + c];
    group_sync();
#endif

    PE_T(4);
    // pass 2: lane k1 (= r) transforms over the former lane index -> Z[k1 + 16 k2] in element k2
#ifndef PE_ABL_FFT
    fft16(re, im);
#endif
    PE_T(5);

    // real-FFT split needs Z[256 - p]: it lives in lane (16 - r) & 15, upper half of its registers
#ifndef PE_ABL_EXCHANGE
#pragma unroll
    for (int u = 0; u < 8; ++u) { S[u * 16 + r] = re[8 + u]; S[128 + u * 16 + r] = im[8 + u]; }
    group_sync();
#endif
    const int pl = (16 - r) & 15;
    R qre[8], qim[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        // r != 0: partner element 15 - m -> upper index 7 - m;  r == 0: element 16 - m -> 8 - m (m >= 1)
        int u = (r == 0) ? (8 - m) : (7 - m);
        u = (u > 7) ? 7 : u;                      // r == 0, m == 0 handled below (own Z[0])
#ifndef PE_ABL_EXCHANGE
        qre[m] = S[u * 16 + pl];
        qim[m] = S[128 + u * 16 + pl];
#else
        qre[m] = re[u + 8] + (R)pl; qim[m] = im[u + 8];
#endif
    }
    if (r == 0) { qre[0] = re[0]; qim[0] = im[0]; }
    group_sync();

    PE_T(6);
    // X[p] = E + W512^p O, X[256-p] = conj(E - W512^p O);  power = |X|^2 / 512
    R psum = R(0);
    // every LDS read of a stage is issued before its first LDS write: the compiler cannot prove that the
    // table reads and the scratch writes never alias, so a read placed after a write waits for it -- one
    // LDS round trip per loop iteration (8 here, 17 in the mel pass: 1.5 us per frame before this was done)
    cplx<R> wv[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) wv[m] = t.w512[r + 16 * m];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int p = r + 16 * m;
        const R a = re[m], b = im[m], c = qre[m], d = qim[m];
        const R er = R(0.5) * (a + c), ei = R(0.5) * (b - d);
        const R orr = R(0.5) * (b + d), oi = R(-0.5) * (a - c);
        const cplx<R> w = wv[m];
        const R tr = orr * w.x - oi * w.y, ti = orr * w.y + oi * w.x;
        const R x1r = er + tr, x1i = ei + ti, x2r = er - tr, x2i = ei - ti;
#ifndef PE_ABL_POWER
        const R p1 = (x1r * x1r + x1i * x1i) * pscale;
        const R p2 = (x2r * x2r + x2i * x2i) * pscale;
#else
        const R p1 = a + c + w.x, p2 = b + d;
#endif
        const int pm = 256 - p;
        S[p + (p >> 4)] = p1;
        S[pm + (pm >> 4)] = p2;
        psum += p1 + p2;
    }
    if (r == 0) {
        const R p128 = (re[8] * re[8] + im[8] * im[8]) * pscale;
        S[128 + 8] = p128;
        psum += p128;
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) psum += __shfl_xor(psum, o, 16);
    group_sync();

    PE_T(7);
    // Sparse mel filterbank.  Every bin feeds at most two (neighbouring, triangular) filters, so the
    // pass runs over BINS: lane r walks bins 16r .. 16r+16 (stride-17 layout: no bank conflicts),
    // keeps one running sum per "stream" (1st / 2nd filter of the bin) and drops it into a partial
    // slot whenever the table says the filter under that stream changes.  Slots are numbered filter
    // by filter, so the second pass adds a contiguous range in a fixed order.
    R* LM = S;                    // log-mel energies: over the head of the power spectrum, dead by then
    R* PART = S + kPowerPad;
#ifndef PE_ABL_MEL
    {
        R acc0 = R(0), acc1 = R(0);
        R pwv[kMelSteps], w0v[kMelSteps], w1v[kMelSteps];
        int flv[kMelSteps];
#pragma unroll
        for (int i = 0; i < kMelSteps; ++i) {                  // all reads first (see the power stage)
            pwv[i] = S[kTrStride * r + i + (i >> 4)];          // padded index of bin 16 r + i
            w0v[i] = t.mel_w[i * 16 + r];
            w1v[i] = t.mel_w[(kMelSteps + i) * 16 + r];
            flv[i] = t.mel_flush[i * 16 + r];
        }
#pragma unroll
        for (int i = 0; i < kMelSteps; ++i) {
            acc0 = real_fma(w0v[i], pwv[i], acc0);
            acc1 = real_fma(w1v[i], pwv[i], acc1);
            // no branch: a step that ends no run stores to the spare slot and keeps its sum
            const int fl = flv[i];
            const int s0 = fl & 0xffff, s1 = (fl >> 16) & 0xffff;
            PART[s0 != 0xffff ? s0 : t.spare] = acc0;
            PART[s1 != 0xffff ? s1 : t.spare] = acc1;
            acc0 = s0 != 0xffff ? R(0) : acc0;
            acc1 = s1 != 0xffff ? R(0) : acc1;
        }
    }
#else
    PART[r] = S[r];
#endif
    group_sync();
    for (int f = r; f < n_filt; f += 16) {
        const int p0 = t.mel_pstart[f], np = t.mel_pstart[f + 1] - p0;
        R acc = R(0);
        for (int i0 = 0; i0 < np; i0 += 4) {
            R v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = (i0 + u < np) ? PART[p0 + i0 + u] : R(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += v[u];
        }
        LM[f] = acc;
    }
    PE_T(18);
    // one more "filter" rides along: the lane that would take filter n_filt takes the total power instead,
    // so the log that replaces coefficient 0 is evaluated in the same pass as the filter logs
    for (int f = r; f <= n_filt; f += 16) {
        const R acc = f < n_filt ? LM[f] : psum;
#ifndef PE_ABL_LOG
        // sonopy clips at eps (safe_log); speechpy replaces exact zeros only (zero_handling): 0 < x < eps stays x
        LM[f] = real_log(log_mode == 0 ? (acc > K::EPS ? acc : K::EPS) : (acc == R(0) ? K::EPS : acc));
#else
        LM[f] = acc + K::EPS;
#endif
    }
    group_sync();

    PE_T(8);
    // DCT-II (ortho) rows 0..n_mfcc-1; row 0 replaced by log total power
    R coeff = R(0);
    if (r < n_mfcc) {
        const R* drow = t.dct + r * n_filt;
#ifdef PE_ABL_DCT
        coeff = LM[r] + drow[0];
#else
        constexpr int CH = 10;                                  // reads of a chunk first, then its FMA chain
        for (int j0 = 0; j0 < n_filt; j0 += CH) {
            R dv[CH], lv[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int jf = j0 + u;
                const int jc = jf < n_filt ? jf : n_filt - 1;
                dv[u] = drow[jc];
                lv[u] = LM[jc];
                if (jf >= n_filt) dv[u] = R(0);
            }
#pragma unroll
            for (int u = 0; u < CH; ++u) coeff = real_fma(dv[u], lv[u], coeff);
        }
#endif
        if (r == 0) coeff = LM[n_filt];
    }
    group_sync();
    PE_T(9);
    return coeff;
}

// ---- streaming kernel -------------------------------------------------------------------
struct PcmView {
    const int16_t* row;     // this stream's chunk
    const int16_t* car;     // this stream's carry
    int q;                  // carry holds virtual samples [0, q); the chunk starts at virtual q
    bool pairs;             // dword loads of (even, odd) sample pairs are legal

    __device__ __forceinline__ int sample(int v) const { return v < q ? (int)car[v] : (int)row[v - q]; }
    // low half = sample v, high half = sample v+1 (0 when !second)
    __device__ __forceinline__ int pair(int v, bool second) const {
        if (pairs && second && ((v & 1) == 0)) {
            const int16_t* p = (v < q) ? (car + v) : (row + (v - q));
            return *reinterpret_cast<const int*>(p);
        }
        const int lo = sample(v) & 0xffff;
        const int hi = second ? sample(v + 1) : 0;
        return lo | (hi << 16);
    }
};

// One workgroup (256 threads) = one tile of 16 streams.  Reads the stream state of this update
// (st_*), writes the state after it (st_*_next): the two may alias only when no other role reads
// the old state concurrently.
//
// Latency plan (this stage is a dependent chain per workgroup, so global-memory round trips are
// what is worth hiding): counters -> PCM of the first frame and of the leftover are requested
// before the 12 KB table image is copied to LDS; the next frame's PCM is requested while the
// current frame is transformed.
// fsel / nsel: the new frames of a tile may be split over nsel workgroups (workgroup fsel takes
// frames fsel, fsel + nsel, ...), so an update that completes two frames per stream is not twice
// as long as one that completes one; only workgroup 0 moves the leftover and the counters.
template <class R>
__device__ __forceinline__ void mfcc_stream_tile(const MfccStreamArgs<R>& a, const int tile, unsigned char* smem,
                                                 const int fsel = 0, const int nsel = 1) {
    using K = RealK<R>;
    const StreamGeom& geo = a.geo;
    PE_T(0);
#ifndef PE_ABL_TABLES
    const TabRegs tab_regs = lds_issue<R>(a.tab);           // first in the queue: back before the PCM is
#endif
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 4, r = lane & 15;
    const int j = wave * 4 + grp;                           // stream within the tile
    const long long s = (long long)tile * kTileStreams + j;
    const bool active = s < geo.n_streams;
    const long long sc = active ? s : 0;                    // padded lanes shadow stream 0 (loads only)

    const int C = a.chunk, hop = geo.hop, flen = geo.frame_len;
    const int q = a.st_q[sc];
    const uint32_t kc = a.st_kc[sc];
    uint32_t ke = a.st_ke[sc];
    // Touch this stream's chunk (one dword per 128-byte line) and carry while the counters are on
    // their way: the counter-dependent sample loads below then merge with / hit these lines instead
    // of starting their own HBM round trip.  The values are never used.
    {
        const int16_t* row0 = a.pcm + (size_t)sc * C;
        const int16_t* car0 = a.carry + (size_t)sc * kCarryCap;
#ifndef PE_ABL_TOUCH
        for (int line = r * 64; line < C; line += 16 * 64) (void)*reinterpret_cast<const volatile int16_t*>(row0 + line);
        if (r < 8) (void)*reinterpret_cast<const volatile int16_t*>(car0 + r * 64);
#endif
    }
    const int avail = q + C;                                 // virtual samples now available
    const int nnew = (active && avail >= flen) ? 1 + (avail - flen) / hop : 0;
    const int qn = avail - nnew * hop;                       // leftover after this update (< flen)

    PcmView pv;
    pv.row = a.pcm + (size_t)sc * C;
    pv.car = a.carry + (size_t)sc * kCarryCap;
    pv.q = q;
    pv.pairs = a.pcm_pairs_ok && ((q & 1) == 0);

    // samples [vb, vb + limit) of the virtual stream as int16 pairs, 32c + 2r per lane
    auto fetch = [&](int vb, int limit, int (&dst)[16]) {
        if (pv.pairs && ((vb | limit) & 1) == 0) {
            // aligned case: sixteen unconditional dword loads (addresses clamped, results masked)
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const int n = 32 * c + 2 * r;
                const int v = vb + (n < limit ? n : 0);
                const int16_t* p = (v < q) ? (pv.car + v) : (pv.row + (v - q));
                const int val = *reinterpret_cast<const int*>(p);
                dst[c] = n < limit ? val : 0;
            }
        } else {
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const int n = 32 * c + 2 * r;
                dst[c] = (n < limit) ? pv.pair(vb + n, n + 1 < limit) : 0;
            }
        }
    };
    int cur[16], left[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) { cur[c] = 0; left[c] = 0; }
    const int slots = geo.ring_slots;
    // frames that would be overwritten before anyone reads them (huge chunks) are skipped
    const int f_first = (nnew > slots ? nnew - slots : 0) + fsel;
    const bool owner = fsel == 0;                            // moves the leftover and the counters
    if (fsel > 0 && !__syncthreads_or(f_first < nnew)) return;   // no stream of the tile has a frame for this workgroup
#ifndef PE_ABL_PCM
    if (f_first < nnew) fetch(f_first * hop, flen, cur);
    if (owner && active && qn > 0) fetch(nnew * hop, qn, left);   // read before any carry store
#else
#pragma unroll
    for (int c = 0; c < 16; ++c) { cur[c] = q + c * 977 + r; left[c] = c; }
#endif

    LdsTab<R> tab;
#ifdef PE_ABL_TABLES
    MfccTables<R> none = a.tab;
    none.blob_bytes = 0;
    R* scratch = lds_setup<R>(smem, none, geo.n_filt, geo.n_mfcc, tab) + a.tab.blob_bytes / sizeof(R);
#else
    R* scratch = lds_commit<R>(smem, a.tab, tab_regs, geo.n_filt, geo.n_mfcc, tab);
#endif
    __syncthreads();
    PE_T(1);
    if (!active) return;
    R* S = scratch + (wave * 4 + grp) * tab.group_reals;
    float* ring_rows = a.ring + ((size_t)tile * slots * kTileStreams + j) * kRowFloats;

    float last_row = 0.0f;                                   // the final frame's ring store is issued
    int last_slot = -1;                                      // after the carry stores (see below)
    for (int f = f_first; f < nnew; f += nsel) {
        auto load = [&](int c, R& xr, R& xi) {
            xr = (R)(int)(short)(cur[c] & 0xffff);
            xi = (R)(cur[c] >> 16);
        };
        const R coeff = mfcc_frame<R>(tab, S, r, geo.n_filt, geo.n_mfcc, load, K::PSCALE_I16, geo.log_mode);
        const uint32_t k = kc + (uint32_t)f;
        const int slot = (int)(k & (uint32_t)(slots - 1));
        const float row = (r < geo.n_mfcc) ? (float)coeff : 0.0f;
        if (f + nsel < nnew) {
            ring_rows[(size_t)slot * kTileStreams * kRowFloats + r] = row;
            fetch((f + nsel) * hop, flen, cur);              // L2-resident: the whole chunk was touched
        } else {
            last_row = row;
            last_slot = slot;
        }
    }

    PE_T(10);
    // leftover: virtual samples [nnew*hop, avail) become the new carry
    if (owner && qn > 0) {
        int16_t* carw = a.carry + (size_t)s * kCarryCap;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // every read of the old carry has landed
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int n = 32 * c + 2 * r;
            if (n + 1 < qn) *reinterpret_cast<int*>(carw + n) = left[c];
            else if (n < qn) carw[n] = (int16_t)(left[c] & 0xffff);
        }
    }
    if (last_slot >= 0) ring_rows[(size_t)last_slot * kTileStreams * kRowFloats + r] = last_row;
    if (owner && r == 0) {
        const uint32_t kcn = kc + (uint32_t)nnew;
        // frame k becomes visible once a whole window [k*hop, k*hop + window) has arrived:
        // Listener.update_vectors only vectorizes when len(window_audio) >= window_samples
        const int m = qn + hop * (int)(kcn - ke);
        if (m >= geo.window) ke += 1u + (uint32_t)((m - geo.window) / hop);
        a.st_q_next[s] = qn;
        a.st_kc_next[s] = kcn;
        a.st_ke_next[s] = ke;
    }
    PE_T(11);
}

// n_updates consecutive updates of every stream in ONE launch (pe_update_many).  Which samples form
// which frame is closed-form integer arithmetic over the virtual stream
//     [carry (q samples)] ++ chunk 0 ++ chunk 1 ++ ... ++ chunk n_updates-1,
// so the frames of a call are independent tasks: the 16-lane group of task (tile, kb, stream) transforms
// frames kb, kb + n_kb, ... of its stream (n_kb rows share a tile), and a second, small launch does the
// bookkeeping -- leftover samples to carry_next, counters to st_*_next, and the emitted-frame counter
// after EVERY update (ke_hist) that tells the network launch which window each update saw.
// carry_next must not alias carry: other rows still read the old carry.
template <class R, bool BOOK>
__device__ __forceinline__ void mfcc_many_tile(const MfccStreamArgs<R>& a, unsigned char* smem) {
    using K = RealK<R>;
    const StreamGeom& geo = a.geo;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 4, r = lane & 15;
    // Two launches share this body (so that each gets its own register allocation: the frame rows need
    // ~130 VGPRs, with the bookkeeping code compiled in the allocation was 232 and nothing else fitted on a
    // SIMD next to two of these workgroups):
    //   BOOK = false: task = (tile, row kb < n_kb, stream of the tile) transforms frames kb, kb + n_kb, ...
    //   BOOK = true:  task = (tile, stream) keeps the books of the call
    const int n_kb = a.n_frame_rows;
    const long long task = (long long)blockIdx.x * (blockDim.x >> 4) + wave * 4 + grp;
    const int j = (int)(task & 15);
    const int kb = BOOK ? n_kb : (int)((task >> 4) % n_kb);
    const int tile = BOOK ? (int)(task >> 4) : (int)((task >> 4) / n_kb);
    const long long s = (long long)tile * kTileStreams + j;
    const bool active = s < geo.n_streams;
    constexpr bool book = BOOK;
    LdsTab<R> tab;
    R* scratch = nullptr;
    if (!BOOK) {
        scratch = lds_setup<R>(smem, a.tab, geo.n_filt, geo.n_mfcc, tab);
        __syncthreads();
    }
    if (!active) return;
    const int C = a.chunk, hop = geo.hop, flen = geo.frame_len, slots = geo.ring_slots, U = a.n_updates;
    const size_t update_stride = (size_t)geo.n_streams * C;
    const int q = a.st_q[s];
    const uint32_t kc = a.st_kc[s];
    const int avail = q + U * C;
    const int nnew = avail >= flen ? 1 + (avail - flen) / hop : 0;
    const int16_t* car = a.carry + (size_t)s * kCarryCap;
    const int16_t* base = a.pcm + (size_t)s * C;

    // virtual sample v (0 <= v < avail): carry below q, chunk (v - q) / C above
    auto vsample = [&](int v) -> int {
        if (v < q) return (int)car[v];
        const int w = v - q, u = w / C;
        return (int)base[(size_t)u * update_stride + (w - u * C)];
    };
    // dword loads of (even, odd) pairs: every quantity that shifts a pair boundary must be even, and a
    // frame may cross at most one chunk boundary
    const bool fast = a.pcm_pairs_ok && ((q | hop | C) & 1) == 0 && C >= flen;
    auto fetch = [&](int vb, int limit, int (&dst)[16]) {
        if (fast && ((vb | limit) & 1) == 0) {
            const int w0 = vb - q;                              // < 0: the span starts inside the carry
            int u0 = 0, off0 = w0;
            if (w0 >= 0) { u0 = w0 / C; off0 = w0 - u0 * C; }
            const int16_t* rowu = base + (size_t)u0 * update_stride;
            const ptrdiff_t wrap = (ptrdiff_t)update_stride - C;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const int n = 32 * c + 2 * r;
                const int nn = n < limit ? n : 0;
                const int v = vb + nn, off = off0 + nn;
                const int16_t* p = (v < q) ? (car + v) : (rowu + off + (off >= C ? wrap : 0));
                const int val = *reinterpret_cast<const int*>(p);
                dst[c] = n < limit ? val : 0;
            }
        } else {
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const int n = 32 * c + 2 * r;
                const int lo = n < limit ? (vsample(vb + n) & 0xffff) : 0;
                const int hi = n + 1 < limit ? vsample(vb + n + 1) : 0;
                dst[c] = lo | (hi << 16);
            }
        }
    };

    if (!book) {
        R* S = scratch + (wave * 4 + grp) * tab.group_reals;
        float* ring_rows = a.ring + ((size_t)tile * slots * kTileStreams + j) * kRowFloats;
        const int f_first = nnew > slots ? nnew - slots : 0;   // older frames would be overwritten anyway
        const int k = kb;                                      // one frame per task: the launch has a row per frame
        if (k < f_first || k >= nnew) return;
        int cur[16];
        fetch(k * hop, flen, cur);
        auto load = [&](int c, R& xr, R& xi) {
            xr = (R)(int)(short)(cur[c] & 0xffff);
            xi = (R)(cur[c] >> 16);
        };
        const R coeff = mfcc_frame<R>(tab, S, r, geo.n_filt, geo.n_mfcc, load, K::PSCALE_I16, geo.log_mode);
        const int slot = (int)((kc + (uint32_t)k) & (uint32_t)(slots - 1));
        ring_rows[(size_t)slot * kTileStreams * kRowFloats + r] = (r < geo.n_mfcc) ? (float)coeff : 0.0f;
        return;
    }

    // ---- bookkeeping row ---------------------------------------------------------------------------------
    const int qn = avail - nnew * hop;
    if (qn > 0) {
        int left[16];
        fetch(nnew * hop, qn, left);
        int16_t* carw = a.carry_next + (size_t)s * kCarryCap;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int n = 32 * c + 2 * r;
            if (n + 1 < qn) *reinterpret_cast<int*>(carw + n) = left[c];
            else if (n < qn) carw[n] = (int16_t)(left[c] & 0xffff);
        }
    }
    if (r == 0) {
        int qu = q;
        uint32_t kcu = kc, ke = a.st_ke[s];
        for (int u = 0; u < U; ++u) {                          // the counters update by update, as pe_update moves them
            const int av = qu + C;
            const int nn = av >= flen ? 1 + (av - flen) / hop : 0;
            qu = av - nn * hop;
            kcu += (uint32_t)nn;
            const int m = qu + hop * (int)(kcu - ke);
            if (m >= geo.window) ke += 1u + (uint32_t)((m - geo.window) / hop);
            a.ke_hist[(size_t)u * a.n_padded + s] = ke;
        }
        a.st_q_next[s] = qu;
        a.st_kc_next[s] = kcu;
        a.st_ke_next[s] = ke;
    }
}

// ---- stateless whole-buffer form (vectorize_raw) ----------------------------------------------
template <class R>
__device__ __forceinline__ void mfcc_offline_block(const MfccOfflineArgs<R>& a, unsigned char* smem) {
    const StreamGeom& geo = a.geo;
    LdsTab<R> tab;
    R* scratch = lds_setup<R>(smem, a.tab, geo.n_filt, geo.n_mfcc, tab);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 4, r = lane & 15;
    const long long fr = (long long)blockIdx.x * (blockDim.x >> 4) + wave * 4 + grp;
    if (fr >= a.n_frames) return;
    R* S = scratch + (wave * 4 + grp) * tab.group_reals;
    const double* x = a.audio + fr * geo.hop;
    const int flen = geo.frame_len;
    auto load = [&](int c, R& xr, R& xi) {
        const int n = 32 * c + 2 * r;
        xr = (n < flen) ? (R)x[n] : R(0);
        xi = (n + 1 < flen) ? (R)x[n + 1] : R(0);
    };
    const R coeff = mfcc_frame<R>(tab, S, r, geo.n_filt, geo.n_mfcc, load, RealK<R>::INV_FFT, geo.log_mode);
    if (a.out && r < geo.n_mfcc) a.out[fr * geo.n_mfcc + r] = (double)coeff;
    if (a.out_mels)            // the log-mel energies are still in the group's scratch (lane f % 16 wrote entry f)
        for (int f = r; f < geo.n_filt; f += 16) a.out_mels[fr * geo.n_filt + f] = (double)S[f];
    if (a.out_rows) a.out_rows[fr * kRowFloats + r] = (r < geo.n_mfcc) ? (float)coeff : 0.0f;
}

}  // namespace pe
