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

template <class R> struct alignas(2 * sizeof(R)) cx { R x, y; };      // one 16 / 8-byte LDS access

template <class R> struct Regs { R re[4], im[4]; };

template <class R>
struct Tab {            // pointers into one image of the blob (LDS on the GPU)
    const cx<R>* tw1;   // [3][64]
    const cx<R>* tw2;   // [3][16]
    const cx<R>* tw3;   // [3][4]
    const cx<R>* w512;  // [2][64]
    const cx<R>* logtab;    // [128] {1 / c_i, log c_i} (float64 only)
    const R* mel_w;     // [mel_len][64]
    const R* dct_w;     // [dct_len][64]
    const int* mel_start;   // [64]
    const int* pstart;      // [65]
    const int* partner;     // [64]
    const float* proj_w;    // [proj_rows][64] float32, may be absent (proj_rows == 0)
    const float* proj_b;    // [64]
    int mel_len, dct_len, np_max, proj_rows;
    int mel_pad;        // rows of mel_w (>= mel_len, zero beyond)
};

// image: where byte `skip` of the blob sits (the GPU's LDS image leaves out the leading twiddle sections, which
// every wave reads once, straight from global memory)
template <class R>
PE_HD Tab<R> bind(const unsigned char* image0, const Layout& L, int skip = 0) {
    const unsigned char* image = image0 - skip;
    Tab<R> t;
    t.tw1 = reinterpret_cast<const cx<R>*>(image + L.tw1);
    t.tw2 = reinterpret_cast<const cx<R>*>(image + L.tw2);
    t.tw3 = reinterpret_cast<const cx<R>*>(image + L.tw3);
    t.w512 = reinterpret_cast<const cx<R>*>(image + L.w512);
    t.logtab = reinterpret_cast<const cx<R>*>(image + L.logtab);
    t.mel_w = reinterpret_cast<const R*>(image + L.mel_w);
    t.dct_w = reinterpret_cast<const R*>(image + L.dct_w);
    t.mel_start = reinterpret_cast<const int*>(image + L.mel_start);
    t.pstart = reinterpret_cast<const int*>(image + L.pstart);
    t.partner = reinterpret_cast<const int*>(image + L.partner);
    t.proj_w = reinterpret_cast<const float*>(image + L.proj_w);
    t.proj_b = reinterpret_cast<const float*>(image + L.proj_b);
    t.mel_len = L.mel_len; t.dct_len = L.dct_len; t.np_max = L.np_max; t.proj_rows = L.proj_rows;
    t.mel_pad = L.mel_pad;
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

// The twiddles a lane needs are the same for every frame: loaded once per wave and kept in registers.
template <class R>
struct LaneConsts {
    cx<R> tw1[3], tw2[3], tw3[3];   // W256^(l k), W64^((l & 15) k), W16^((l & 3) k), k = 1..3
    cx<R> w512[2];                  // W512^(kbase(l) + 64 j)
    int partner;                    // lane holding the mirror bins
};

template <class R>
PE_HD LaneConsts<R> lane_consts(const Tab<R>& t, int l) {
    LaneConsts<R> c;
    for (int k = 0; k < 3; ++k) { c.tw1[k] = t.tw1[k * 64 + l]; c.tw2[k] = t.tw2[k * 16 + (l & 15)]; c.tw3[k] = t.tw3[k * 4 + (l & 3)]; }
    c.w512[0] = t.w512[l]; c.w512[1] = t.w512[64 + l];
    c.partner = t.partner[l];
    return c;
}

// the four radix-4 passes; between them the caller transposes (register index) x (lane digit)
template <class R> PE_HD void pass_a(Regs<R>& v, const LaneConsts<R>& c) { radix4(v); twiddle3(v, c.tw1[0], c.tw1[1], c.tw1[2]); }
template <class R> PE_HD void pass_b(Regs<R>& v, const LaneConsts<R>& c) { radix4(v); twiddle3(v, c.tw2[0], c.tw2[1], c.tw2[2]); }
template <class R> PE_HD void pass_c(Regs<R>& v, const LaneConsts<R>& c) { radix4(v); twiddle3(v, c.tw3[0], c.tw3[1], c.tw3[2]); }
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

// scratch slots (ppos) of the four powers of lane l: bins kbase, kbase + 64, 256 - kbase, 192 - kbase
template <class R>
PE_HD void power_bins(int l, int (&bins)[4]) {
    const int kb = kbase_of(l);
    bins[0] = ppos<R>(kb); bins[1] = ppos<R>(kb + 64); bins[2] = ppos<R>(256 - kb); bins[3] = ppos<R>(192 - kb);
}

// log(x), x > 0.  float64: table-driven -- x = m 2^e, m in [0.5, 1) falls in one of 128 intervals with centre c;
// log x = e ln 2 + log c + log1p(m / c - 1), |m / c - 1| <= 2^-8, seven terms of the series (truncation < 2^-66);
// absolute error a few 1e-16 plus |e| ulp(ln 2) -- about 20 float64 operations instead of libm's ~80.
PE_HD int mant_index7(double m) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (__double2hiint(m) >> 13) & 127;
#else
    unsigned long long u; std::memcpy(&u, &m, 8); return (int)((u >> 45) & 127);
#endif
}
PE_HD double wave_log(double x, const cx<double>* logtab) {
    int e;
    const double m = frexp(x, &e);
    const cx<double> t = logtab[mant_index7(m)];
    const double r = fma(m, t.x, -1.0);
    double p = fma(r, 1.0 / 7.0, -1.0 / 6.0);
    p = fma(p, r, 1.0 / 5.0);
    p = fma(p, r, -1.0 / 4.0);
    p = fma(p, r, 1.0 / 3.0);
    p = fma(p, r, -1.0 / 2.0);
    p = fma(p, r, 1.0);
    return fma((double)e, 0.6931471805599453094, fma(p, r, t.y));
}
PE_HD float wave_log(float x, const cx<float>*) {
#if defined(__HIP_DEVICE_COMPILE__)
    // v_log_f32 (log2, 1 ulp) times ln 2: two instructions where logf() is thirteen (denormal scaling, an extended-precision
    // product, infinity checks).  The argument is >= 2^-52 (safe_log's clip) or a power of int16 samples: never denormal.
    // Absolute error <= |ln x| 2^-23 + 1 ulp, i.e. a few 1e-6 at ln 2^-52 = -36: inside the float32 front end's own rounding
    // (tests: every mfcc_precision = 'f32' case against the float64 oracle).
    return __builtin_amdgcn_logf(x) * 0.6931471805599453094f;
#else
    return std::log(x);
#endif
}

template <class R>
PE_HD R mel_run(const Tab<R>& t, const R* P, int l) {
    const int s = t.mel_start[l];
    R acc = R(0);
    for (int i = 0; i < t.mel_pad; ++i) acc += t.mel_w[i * 64 + l] * P[s + i];
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
