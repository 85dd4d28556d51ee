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

template <class R> struct RealK;
template <> struct RealK<double> {
    static constexpr double C1 = 0.92387953251128673848;   // cos(pi/8)
    static constexpr double S1 = 0.38268343236508978178;   // sin(pi/8)
    static constexpr double H = 0.70710678118654752440;    // sqrt(1/2)
    static constexpr double EPS = 2.220446049250313e-16;   // np.finfo(float).eps (sonopy safe_log)
    static constexpr double INV_FFT = 1.0 / 512.0;
    static constexpr double INV_I16 = 1.0 / 32768.0;       // util.py:37
};
template <> struct RealK<float> {
    static constexpr float C1 = 0.92387953251128673848f;
    static constexpr float S1 = 0.38268343236508978178f;
    static constexpr float H = 0.70710678118654752440f;
    static constexpr float EPS = 2.220446049250313e-16f;
    static constexpr float INV_FFT = 1.0f / 512.0f;
    static constexpr float INV_I16 = 1.0f / 32768.0f;
};

__device__ __forceinline__ double real_log(double x) { return log(x); }
__device__ __forceinline__ float real_log(float x) { return logf(x); }
__device__ __forceinline__ double real_fma(double a, double b, double c) { return fma(a, b, c); }
__device__ __forceinline__ float real_fma(float a, float b, float c) { return fmaf(a, b, c); }

// LDS traffic inside one wave is executed in program order by the hardware; these fences only
// keep the compiler from moving DS operations across the hand-off and make it wait for them.
__device__ __forceinline__ void group_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
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
    const R* mel_w;
    const int* mel_start;
    const int* mel_off;
};

constexpr int kTrStride = 17;                       // padded row of the 16x16 LDS transpose
constexpr int kGroupScratch = 16 * kTrStride + kMaxFilt;   // transpose / power buffer + log-mels

__host__ __device__ inline size_t lds_layout_bytes(int real_size, int n_filt, int n_mfcc, int mel_nnz) {
    size_t b = 0;
    b += 256 * 2 * (size_t)real_size;               // tw256
    b += 130 * 2 * (size_t)real_size;               // w512 (129 used)
    b += (size_t)n_mfcc * n_filt * real_size;       // dct
    b += (size_t)mel_nnz * real_size;               // mel_w
    b = (b + 15) & ~(size_t)15;
    b += ((size_t)(2 * n_filt + 1) * sizeof(int) + 15) & ~(size_t)15;
    b += (size_t)16 * kGroupScratch * real_size;    // 16 groups per workgroup
    return b;
}

template <class R>
__device__ __forceinline__ R* lds_setup(unsigned char* smem, const MfccTables<R>& g, int n_filt, int n_mfcc,
                                        LdsTab<R>& t) {
    cplx<R>* tw = reinterpret_cast<cplx<R>*>(smem);
    cplx<R>* w5 = tw + 256;
    R* dct = reinterpret_cast<R*>(w5 + 130);
    R* mw = dct + n_mfcc * n_filt;
    size_t off = (size_t)((unsigned char*)(mw + g.mel_nnz) - smem);
    off = (off + 15) & ~(size_t)15;
    int* ms = reinterpret_cast<int*>(smem + off);
    int* mo = ms + n_filt;
    off += ((size_t)(2 * n_filt + 1) * sizeof(int) + 15) & ~(size_t)15;
    R* scratch = reinterpret_cast<R*>(smem + off);
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int i = tid; i < 256; i += nt) tw[i] = g.tw256[i];
    for (int i = tid; i < 129; i += nt) w5[i] = g.w512[i];
    for (int i = tid; i < n_mfcc * n_filt; i += nt) dct[i] = g.dct[i];
    for (int i = tid; i < g.mel_nnz; i += nt) mw[i] = g.mel_w[i];
    for (int i = tid; i < n_filt; i += nt) ms[i] = g.mel_start[i];
    for (int i = tid; i <= n_filt; i += nt) mo[i] = g.mel_off[i];
    t.tw256 = tw; t.w512 = w5; t.dct = dct; t.mel_w = mw; t.mel_start = ms; t.mel_off = mo;
    __syncthreads();
    return scratch;
}

// One frame on one 16-lane group.  `load(c, xr, xi)` returns samples 32c+2r and 32c+2r+1 of the
// frame (already scaled to [-1,1), zero beyond frame_len).  Returns coefficient r in lane r
// (lanes >= n_mfcc return 0).  S: this group's LDS scratch (kGroupScratch reals).
template <class R, class Load>
__device__ __forceinline__ R mfcc_frame(const LdsTab<R>& t, R* S, int r, int n_filt, int n_mfcc, Load load) {
    using K = RealK<R>;
    R re[16], im[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) load(c, re[c], im[c]);

    // pass 1: lane r transforms z[16c + r] over c -> Y_r[k1]; twiddle by W256^(r k1)
    fft16(re, im);
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) {
        const cplx<R> w = t.tw256[k1 * 16 + r];
        const R a = re[k1], b = im[k1];
        re[k1] = a * w.x - b * w.y;
        im[k1] = a * w.y + b * w.x;
    }
    // 16x16 transpose through LDS (row stride 17 reals: conflict-free both ways), re then im
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

    // pass 2: lane k1 (= r) transforms over the former lane index -> Z[k1 + 16 k2] in element k2
    fft16(re, im);

    // real-FFT split needs Z[256 - p]: it lives in lane (16 - r) & 15, upper half of its registers
#pragma unroll
    for (int u = 0; u < 8; ++u) { S[u * 16 + r] = re[8 + u]; S[128 + u * 16 + r] = im[8 + u]; }
    group_sync();
    const int pl = (16 - r) & 15;
    R qre[8], qim[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        // r != 0: partner element 15 - m -> upper index 7 - m;  r == 0: element 16 - m -> 8 - m (m >= 1)
        int u = (r == 0) ? (8 - m) : (7 - m);
        u = (u > 7) ? 7 : u;                      // r == 0, m == 0 handled below (own Z[0])
        qre[m] = S[u * 16 + pl];
        qim[m] = S[128 + u * 16 + pl];
    }
    if (r == 0) { qre[0] = re[0]; qim[0] = im[0]; }
    group_sync();

    // X[p] = E + W512^p O, X[256-p] = conj(E - W512^p O);  power = |X|^2 / 512
    R psum = R(0);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int p = r + 16 * m;
        const R a = re[m], b = im[m], c = qre[m], d = qim[m];
        const R er = R(0.5) * (a + c), ei = R(0.5) * (b - d);
        const R orr = R(0.5) * (b + d), oi = R(-0.5) * (a - c);
        const cplx<R> w = t.w512[p];
        const R tr = orr * w.x - oi * w.y, ti = orr * w.y + oi * w.x;
        const R x1r = er + tr, x1i = ei + ti, x2r = er - tr, x2i = ei - ti;
        const R p1 = (x1r * x1r + x1i * x1i) * K::INV_FFT;
        const R p2 = (x2r * x2r + x2i * x2i) * K::INV_FFT;
        S[p] = p1;
        S[256 - p] = p2;
        psum += p1 + p2;
    }
    if (r == 0) {
        const R p128 = (re[8] * re[8] + im[8] * im[8]) * K::INV_FFT;
        S[128] = p128;
        psum += p128;
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) psum += __shfl_xor(psum, o, 16);
    group_sync();

    // sparse triangular mel filters, then log(clip)
    R* LM = S + 16 * kTrStride;
    for (int f = r; f < n_filt; f += 16) {
        const int start = t.mel_start[f], o0 = t.mel_off[f], len = t.mel_off[f + 1] - o0;
        R acc = R(0);
        for (int i = 0; i < len; ++i) acc = real_fma(t.mel_w[o0 + i], S[start + i], acc);
        LM[f] = real_log(acc > K::EPS ? acc : K::EPS);
    }
    group_sync();

    // DCT-II (ortho) rows 0..n_mfcc-1; row 0 replaced by log total power
    R coeff = R(0);
    if (r < n_mfcc) {
        const R* drow = t.dct + r * n_filt;
        for (int jf = 0; jf < n_filt; ++jf) coeff = real_fma(drow[jf], LM[jf], coeff);
        if (r == 0) coeff = real_log(psum > K::EPS ? psum : K::EPS);
    }
    group_sync();
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
template <class R>
__device__ __forceinline__ void mfcc_stream_tile(const MfccStreamArgs<R>& a, const int tile, unsigned char* smem) {
    using K = RealK<R>;
    const StreamGeom& geo = a.geo;
    LdsTab<R> tab;
    R* scratch = lds_setup<R>(smem, a.tab, geo.n_filt, geo.n_mfcc, tab);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 4, r = lane & 15;
    const int j = wave * 4 + grp;                           // stream within the tile
    const long long s = (long long)tile * kTileStreams + j;
    if (s >= geo.n_streams) return;
    R* S = scratch + (wave * 4 + grp) * kGroupScratch;

    const int q = a.st_q[s];
    const uint32_t kc = a.st_kc[s];
    uint32_t ke = a.st_ke[s];
    const int C = a.chunk, hop = geo.hop, flen = geo.frame_len;
    const int avail = q + C;                                 // virtual samples now available
    const int nnew = avail >= flen ? 1 + (avail - flen) / hop : 0;

    PcmView pv;
    pv.row = a.pcm + (size_t)s * C;
    pv.car = a.carry + (size_t)s * kCarryCap;
    pv.q = q;
    pv.pairs = a.pcm_pairs_ok && ((q & 1) == 0);

    const int slots = geo.ring_slots;
    float* ring_rows = a.ring + ((size_t)tile * slots * kTileStreams + j) * kRowFloats;

    for (int f = 0; f < nnew; ++f) {
        if (nnew - f > slots) continue;                      // would be overwritten before anyone reads it
        const int vb = f * hop;
        auto load = [&](int c, R& xr, R& xi) {
            const int n = 32 * c + 2 * r;
            int pr = 0;
            if (n < flen) pr = pv.pair(vb + n, n + 1 < flen);
            xr = (R)(int)(short)(pr & 0xffff) * K::INV_I16;
            xi = (R)(pr >> 16) * K::INV_I16;
        };
        const R coeff = mfcc_frame<R>(tab, S, r, geo.n_filt, geo.n_mfcc, load);
        const uint32_t k = kc + (uint32_t)f;
        const int slot = (int)(k & (uint32_t)(slots - 1));
        ring_rows[(size_t)slot * kTileStreams * kRowFloats + r] = (r < geo.n_mfcc) ? (float)coeff : 0.0f;
    }

    // leftover: virtual samples [nnew*hop, avail) become the new carry (< frame_len of them)
    const int qn = avail - nnew * hop;
    if (qn > 0) {
        const int vs = nnew * hop;
        int16_t* carw = a.carry + (size_t)s * kCarryCap;
        int buf[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int n = 32 * c + 2 * r;
            buf[c] = (n < qn) ? pv.pair(vs + n, n + 1 < qn) : 0;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // every read of the old carry has landed
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int n = 32 * c + 2 * r;
            if (n + 1 < qn) *reinterpret_cast<int*>(carw + n) = buf[c];
            else if (n < qn) carw[n] = (int16_t)(buf[c] & 0xffff);
        }
    }
    if (r == 0) {
        const uint32_t kcn = kc + (uint32_t)nnew;
        // frame k becomes visible once a whole window [k*hop, k*hop + window) has arrived:
        // Listener.update_vectors only vectorizes when len(window_audio) >= window_samples
        const int m = qn + hop * (int)(kcn - ke);
        if (m >= geo.window) ke += 1u + (uint32_t)((m - geo.window) / hop);
        a.st_q_next[s] = qn;
        a.st_kc_next[s] = kcn;
        a.st_ke_next[s] = ke;
    }
}

// ---- stateless whole-buffer form (vectorize_raw) ----------------------------------------------
template <class R>
__device__ __forceinline__ void mfcc_offline_block(const MfccOfflineArgs<R>& a, unsigned char* smem) {
    const StreamGeom& geo = a.geo;
    LdsTab<R> tab;
    R* scratch = lds_setup<R>(smem, a.tab, geo.n_filt, geo.n_mfcc, tab);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 4, r = lane & 15;
    const long long fr = (long long)blockIdx.x * 16 + wave * 4 + grp;
    if (fr >= a.n_frames) return;
    R* S = scratch + (wave * 4 + grp) * kGroupScratch;
    const double* x = a.audio + fr * geo.hop;
    const int flen = geo.frame_len;
    auto load = [&](int c, R& xr, R& xi) {
        const int n = 32 * c + 2 * r;
        xr = (n < flen) ? (R)x[n] : R(0);
        xi = (n + 1 < flen) ? (R)x[n + 1] : R(0);
    };
    const R coeff = mfcc_frame<R>(tab, S, r, geo.n_filt, geo.n_mfcc, load);
    if (r < geo.n_mfcc) a.out[fr * geo.n_mfcc + r] = (double)coeff;
}

}  // namespace pe
