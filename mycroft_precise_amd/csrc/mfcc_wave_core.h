// Per-lane arithmetic of the one-frame-per-wave MFCC (see mfcc_wave_tables.h for the work split).
// Every function here touches only ONE lane's registers plus the tables / scratch it is handed; whatever crosses
// lanes (the three digit exchanges, the mirror exchange, the reductions) is done by the caller between the calls:
// mfcc_wave_device.h does it with v_permlane*_swap, LDS and DPP on the GPU, tools/emulate_mfcc_wave.cpp with plain
// loops over 64 "lanes" on the CPU -- the same arithmetic, the same tables, the same order of operations.
#pragma once
#include "mfcc_wave_tables.h"

#if defined(__HIPCC__)
#define PE_HD __host__ __device__ __forceinline__
#else
#define PE_HD inline
#endif

namespace pe_wave {

template <class R> struct cx { R x, y; };

template <class R> struct Regs { R re[4], im[4]; };

template <class R>
struct Tab {            // pointers into one image of the blob (LDS on the GPU)
    const cx<R>* tw1;   // [3][64]
    const cx<R>* tw2;   // [3][16]
    const cx<R>* tw3;   // [3][4]
    const cx<R>* w512;  // [2][64]
    const R* mel_w;     // [mel_len][64]
    const R* dct_w;     // [dct_len][64]
    const int* mel_start;   // [64]
    const int* pstart;      // [65]
    const int* partner;     // [64]
    int mel_len, dct_len, np_max;
};

template <class R>
PE_HD Tab<R> bind(const unsigned char* image, const Layout& L) {
    Tab<R> t;
    t.tw1 = reinterpret_cast<const cx<R>*>(image + L.tw1);
    t.tw2 = reinterpret_cast<const cx<R>*>(image + L.tw2);
    t.tw3 = reinterpret_cast<const cx<R>*>(image + L.tw3);
    t.w512 = reinterpret_cast<const cx<R>*>(image + L.w512);
    t.mel_w = reinterpret_cast<const R*>(image + L.mel_w);
    t.dct_w = reinterpret_cast<const R*>(image + L.dct_w);
    t.mel_start = reinterpret_cast<const int*>(image + L.mel_start);
    t.pstart = reinterpret_cast<const int*>(image + L.pstart);
    t.partner = reinterpret_cast<const int*>(image + L.partner);
    t.mel_len = L.mel_len; t.dct_len = L.dct_len; t.np_max = L.np_max;
    return t;
}

// y_k = sum_n x_n (-i)^(n k), in place
template <class R>
PE_HD void radix4(Regs<R>& v) {
    const R t0r = v.re[0] + v.re[2], t0i = v.im[0] + v.im[2], t1r = v.re[0] - v.re[2], t1i = v.im[0] - v.im[2];
    const R t2r = v.re[1] + v.re[3], t2i = v.im[1] + v.im[3], t3r = v.re[1] - v.re[3], t3i = v.im[1] - v.im[3];
    v.re[0] = t0r + t2r; v.im[0] = t0i + t2i;
    v.re[2] = t0r - t2r; v.im[2] = t0i - t2i;
    v.re[1] = t1r + t3i; v.im[1] = t1i - t3r;
    v.re[3] = t1r - t3i; v.im[3] = t1i + t3r;
}

template <class R>
PE_HD void twiddle3(Regs<R>& v, const cx<R>& w1, const cx<R>& w2, const cx<R>& w3) {
    const cx<R> w[3] = {w1, w2, w3};
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int k = 1; k < 4; ++k) {
        const R a = v.re[k], b = v.im[k];
        v.re[k] = a * w[k - 1].x - b * w[k - 1].y;
        v.im[k] = a * w[k - 1].y + b * w[k - 1].x;
    }
}

// the four radix-4 passes; between them the caller transposes (register index) x (lane digit)
template <class R> PE_HD void pass_a(Regs<R>& v, int l, const Tab<R>& t) { radix4(v); twiddle3(v, t.tw1[l], t.tw1[64 + l], t.tw1[128 + l]); }
template <class R> PE_HD void pass_b(Regs<R>& v, int l, const Tab<R>& t) { radix4(v); const int m = l & 15; twiddle3(v, t.tw2[m], t.tw2[16 + m], t.tw2[32 + m]); }
template <class R> PE_HD void pass_c(Regs<R>& v, int l, const Tab<R>& t) { radix4(v); const int d = l & 3; twiddle3(v, t.tw3[d], t.tw3[4 + d], t.tw3[8 + d]); }
template <class R> PE_HD void pass_d(Regs<R>& v) { radix4(v); }

// source of register r' of lane l in the exchange of lane digit `shift` (4, 2 or 0): (lane, register) it comes from
PE_HD int xchg_src_lane(int l, int shift, int rp) { return (l & ~(3 << shift)) | (rp << shift); }
PE_HD int xchg_src_reg(int l, int shift) { return (l >> shift) & 3; }
PE_HD int xchg_index(int lane, int reg) { return lane * kXchgStride + reg; }     // complex element in the scratch

// Real-FFT split of the pairs of registers 0 and 1 with the mirror bins (zq0 <-> register 0, zq1 <-> register 1):
//   X[p] = E + W512^p O,  X[256 - p] = conj(E - W512^p O),  E = (Z[p] + conj Z[256-p]) / 2,  O = (Z[p] - conj Z[256-p]) / 2i
// (the two halvings are folded into pscale4 = pscale / 4).  pw[0], pw[1] = power of bins kbase, kbase + 64;
// pw[2], pw[3] = power of bins 256 - kbase, 192 - kbase.
template <class R>
PE_HD void split_power(const Regs<R>& v, const cx<R>& zq0, const cx<R>& zq1, const cx<R>& w0, const cx<R>& w1, R pscale4, R (&pw)[4]) {
    const cx<R> zq[2] = {zq0, zq1};
    const cx<R> w[2] = {w0, w1};
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int j = 0; j < 2; ++j) {
        const R a = v.re[j], b = v.im[j], c = zq[j].x, d = zq[j].y;
        const R er = a + c, ei = b - d;             // 2 E
        const R orr = b + d, oi = c - a;            // 2 O = -i (Z[p] - conj Z[q])
        const R tr = orr * w[j].x - oi * w[j].y, ti = orr * w[j].y + oi * w[j].x;
        const R x1r = er + tr, x1i = ei + ti, x2r = er - tr, x2i = ei - ti;
        pw[j] = (x1r * x1r + x1i * x1i) * pscale4;
        pw[2 + j] = (x2r * x2r + x2i * x2i) * pscale4;
    }
}

// bins of the four powers of lane l
PE_HD void power_bins(int l, int (&bins)[4]) {
    const int kb = kbase_of(l);
    bins[0] = kb; bins[1] = kb + 64; bins[2] = 256 - kb; bins[3] = 192 - kb;
}

template <class R>
PE_HD R mel_run(const Tab<R>& t, const R* P, int l) {
    const int s = t.mel_start[l];
    R acc = R(0);
    for (int i = 0; i < t.mel_len; ++i) acc += t.mel_w[i * 64 + l] * P[s + i];
    return acc;
}

template <class R>
PE_HD R filter_sum(const Tab<R>& t, const R* PART, int f) {
    const int p0 = t.pstart[f], np = t.pstart[f + 1] - p0;
    R acc = R(0);
    for (int i = 0; i < t.np_max; ++i) acc += (i < np) ? PART[p0 + i] : R(0);
    return acc;
}

template <class R>
PE_HD R dct_run(const Tab<R>& t, const R* LM, int l, int n_filt) {
    const int q = l & 3;
    R acc = R(0);
    for (int i = 0; i < t.dct_len; ++i) {
        const int n = t.dct_len * q + i;
        acc += t.dct_w[i * 64 + l] * LM[n < n_filt ? n : n_filt - 1];
    }
    return acc;
}

}  // namespace pe_wave
