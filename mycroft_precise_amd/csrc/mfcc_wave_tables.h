// Constant tables of the one-frame-per-wave MFCC kernel (mfcc_wave_device.h), built on the host.
// Pure C++ (no HIP): engine.hip uploads the blob, tools/emulate_mfcc_wave.cpp replays the kernel's data flow
// lane by lane on the CPU from the SAME blob, so that every index convention is checked without a GPU.
//
// Work split of one 512-sample frame over the 64 lanes of a wave (z[n] = x[2n] + i x[2n+1], 256 complex points,
// n = 64 a + 16 b + 4 c + d):
//   lane l = 16 b + 4 c + d starts with register a = z[l + 64 a] (four coalesced 256-byte loads per wave);
//   four radix-4 passes (over a, b, c, d), between them a 4x4 transpose of (register index) x (one lane digit):
//   digit b through v_permlane32/16_swap, digits c and d through the wave's LDS scratch;
//   after the last pass lane l = 16 k1 + 4 k2 + k3 holds Z[kbase(l) + 64 k4] in register k4,
//   kbase(l) = k1 + 4 k2 + 16 k3 (a digit reversal of l);
//   the real-FFT split pairs bin p with 256 - p, which sits in lane partner(l) (kbase' = 64 - kbase), register
//   3 - k4: every lane forms the pairs of its registers 0 and 1 and so owns the power of the four bins
//   {kbase, kbase + 64, 256 - kbase, 192 - kbase} (lane 0: 0, 64, 192, 256 and the self-paired bin 128);
//   mel filterbank: the non-zero weights of every filter are cut into runs of at most mel_len bins, one run per
//   lane (filter by filter, so a filter's partial sums sit in consecutive lanes and are added in lane order);
//   DCT: lane 4 c + q adds terms dct_len q .. dct_len q + dct_len - 1 of coefficient c, then the quad is reduced.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace pe_wave {

constexpr int kLanes = 64;
constexpr int kBins = 257;
constexpr int kMaxMelLen = 16;      // bins per lane in the mel pass
constexpr int kMaxDctLen = 16;      // terms per lane in the DCT (n_filt <= 64)
constexpr int kMaxFilt = 64;
constexpr int kLogTab = 128;        // intervals of the mantissa in the table-driven float64 logarithm

// per-wave LDS scratch, in reals: the exchange area [64 lanes][5] complex (stride 5: 80 / 40 bytes per lane, no
// bank conflicts for 16 / 8-byte accesses), reused after the FFT for the power spectrum, the per-lane partial filter
// sums and the log-mel energies
constexpr int kXchgStride = 5;                        // complex elements per lane
constexpr int kScratchReals = kLanes * kXchgStride * 2;
constexpr int kPowerOff = 0;                          // P[ppos(0) .. ppos(256)]: the power spectrum, one idle slot after every 16 bins
constexpr int kPowerSlots = 257 + 16;                 // upper bound of power_slots<R>()
constexpr int kPartOff = 288;                         // PART[0..63]
constexpr int kLogMelOff = 352;                       // LM[0..n_filt] (n_filt <= 64)
static_assert(kLogMelOff + kMaxFilt + 1 <= kScratchReals, "scratch layout");

#if defined(__HIPCC__)
#define PE_WAVE_HD __host__ __device__
#else
#define PE_WAVE_HD
#endif
PE_WAVE_HD inline int kbase_of(int l) { return (l >> 4) + 4 * ((l >> 2) & 3) + 16 * (l & 3); }
// Slot of bin k in the scratch.  The lanes of a 16-lane store group own bins 4 c + 16 d (+ const): packed densely, the
// four d's of a c would fall on the same LDS banks (a 4-way conflict on every power store); skewed by one slot per 16
// bins they do not.  The filterbank runs are laid out in slot space; an idle slot inside a run carries weight 0.
// (float32: the 4-byte stores of a 32-lane group cover 32 different banks as they are -- no skew, one register less)
template <class R> PE_WAVE_HD inline int ppos(int k) { return sizeof(R) == 8 ? k + (k >> 4) : k; }
template <class R> PE_WAVE_HD inline int bin_of_slot(int p) { return sizeof(R) == 8 ? ((p % 17 == 16) ? -1 : 16 * (p / 17) + p % 17) : p; }
template <class R> PE_WAVE_HD inline int power_slots() { return ppos<R>(256) + 1; }

struct Layout {          // byte offsets into the blob (16-byte aligned sections)
    int tw1, tw2, tw3, w512, logtab, mel_w, dct_w, mel_start, pstart, partner, proj_w, proj_b, total;
    int proj_rows;       // 0: no input-projection epilogue; else n_mfcc (rows of proj_w)
    int mel_pad, dct_pad, np_pad;     // loop bounds the kernel is compiled for (tables zero-padded up to them)
    int mel_len, dct_len, np_max;
};

constexpr int align16(int v) { return (v + 15) & ~15; }

constexpr int kProjRow = 64;        // floats per input-projection row (4 MFMA output tiles x 16 rows)

// (constexpr: for a table SHAPE -- mel_len, dct_len as compiled into a kernel -- every section offset but `total` is a
//  compile-time constant there, and the kernel's LDS addresses fold into instruction offsets)
constexpr Layout layout(int real_size, int mel_len, int dct_len, int np_max, int proj_rows = 0) {
    Layout L{};
    int off = 0;
    L.tw1 = off; off += 3 * 64 * 2 * real_size;
    L.tw2 = off; off += 3 * 16 * 2 * real_size;
    L.tw3 = off; off += 3 * 4 * 2 * real_size;
    L.w512 = off; off += 2 * 64 * 2 * real_size;
    L.logtab = off; off += (real_size == 8 ? kLogTab : 0) * 2 * real_size;       // float64 only (float uses logf)
    L.mel_w = off; off += mel_len * 64 * real_size;
    L.dct_w = off; off += dct_len * 64 * real_size;
    off = align16(off);
    L.mel_start = off; off += 64 * 4;
    L.pstart = off; off += (kMaxFilt + 1) * 4; off = align16(off);
    L.partner = off; off += 64 * 4;
    L.proj_w = off; off += proj_rows * kProjRow * 4;      // float32 [n_mfcc][64]: the network's input kernel in MFMA slot order
    L.proj_b = off; off += (proj_rows ? kProjRow * 4 : 0);
    L.proj_rows = proj_rows;
    L.total = align16(off);
    L.mel_len = mel_len; L.dct_len = dct_len; L.np_max = np_max;
    return L;
}

// mel_filters: [n_filt][257] row-major; dct-II ortho rows built here.  Returns "" or an error message.
// proj_w / proj_b (optional): [n_mfcc][64] / [64] float32, appended to the image for the input-projection epilogue
template <class R>
std::string build(const double* mel_filters, int n_filt, int n_mfcc, std::vector<unsigned char>& blob, Layout& L,
                  const float* proj_w = nullptr, const float* proj_b = nullptr) {
    const double PI = 3.14159265358979323846;
    if (n_filt < 1 || n_filt > kMaxFilt || n_mfcc < 1 || n_mfcc > 16 || n_mfcc > n_filt) return "need 1 <= n_mfcc <= 16, n_mfcc <= n_filt <= 64";
    // support of every filter: one contiguous run of non-zero weights
    std::vector<int> lo(n_filt, 0), hi(n_filt, 0);
    for (int f = 0; f < n_filt; ++f) {
        int a = -1, b = -1;
        for (int k = 0; k < kBins; ++k)
            if (mel_filters[(size_t)f * kBins + k] != 0.0) { if (a < 0) a = k; b = k + 1; }
        if (a < 0) { a = 0; b = 0; }                     // an empty filter: its energy is 0 -> log(eps)
        lo[f] = a; hi[f] = b;
    }
    // runs of a filter's support, cut so that a run spans at most `len` SLOTS (ppos) of the scratch
    auto cut = [&](int len, std::vector<int>* f_of, std::vector<int>* lo_of, std::vector<int>* n_of, std::vector<int>* first_of, int* parts_max) {
        int runs = 0, pmax = 1;
        for (int f = 0; f < n_filt; ++f) {
            if (first_of) (*first_of)[f] = runs;
            int parts = 0;
            if (hi[f] == lo[f]) {
                if (f_of && runs < 64) { (*f_of)[runs] = f; (*lo_of)[runs] = lo[f]; (*n_of)[runs] = 0; }
                ++runs; parts = 1;
            }
            for (int k = lo[f]; k < hi[f];) {
                int e = k;
                while (e + 1 < hi[f] && ppos<R>(e + 1) - ppos<R>(k) + 1 <= len) ++e;
                if (f_of && runs < 64) { (*f_of)[runs] = f; (*lo_of)[runs] = k; (*n_of)[runs] = e - k + 1; }
                ++runs; ++parts;
                k = e + 1;
            }
            if (parts > pmax) pmax = parts;
        }
        if (parts_max) *parts_max = pmax;
        return runs;
    };
    // the kernel's run loop has a compile-time trip count (10 slots in the stock shape, 16 in the wide one), so a run
    // may as well use all of it: the stock length first, the wide one if that does not fit 64 lanes / 8 runs per filter
    const int dct_len = (n_filt + 3) / 4;
    int mel_len = 0;
    {
        int pm = 0;
        if (dct_len <= 5 && cut(10, nullptr, nullptr, nullptr, nullptr, &pm) <= kLanes && pm <= 8) mel_len = 10;
        else if (cut(kMaxMelLen, nullptr, nullptr, nullptr, nullptr, &pm) <= kLanes) mel_len = kMaxMelLen;
    }
    if (!mel_len) return "mel filterbank too wide for one wave: the non-zero runs need more than 64 lanes of 16 slots";
    // the kernel's table-driven loops have compile-time bounds: "stock" (10 / 5 / 8) when everything fits, else 16 / 16 / 16
    std::vector<int> pstart(kMaxFilt + 1, 0), mel_start(64, 0), seg_lo(64, 0), seg_n(64, 0), seg_f(64, -1);
    int np_needed = 1;
    int lane = cut(mel_len, &seg_f, &seg_lo, &seg_n, &pstart, &np_needed);
    const int np_max = np_needed;
    const bool stock = mel_len <= 10 && dct_len <= 5 && np_needed <= 8;
    const int mel_pad = stock ? 10 : 16, dct_pad = stock ? 5 : 16, np_pad = stock ? 8 : 16;
    if (np_needed > np_pad) return "mel filterbank: a filter is spread over more than 16 lanes";
    for (int f = n_filt; f <= kMaxFilt; ++f) pstart[f] = lane;
    L = layout((int)sizeof(R), mel_pad, dct_pad, np_max, proj_w ? n_mfcc : 0);
    L.mel_len = mel_len; L.dct_len = dct_len;
    L.mel_pad = mel_pad; L.dct_pad = dct_pad; L.np_pad = np_pad;
    blob.assign((size_t)L.total, 0);
    if (proj_w) {
        std::memcpy(blob.data() + L.proj_w, proj_w, (size_t)n_mfcc * kProjRow * 4);
        std::memcpy(blob.data() + L.proj_b, proj_b, (size_t)kProjRow * 4);
    }
    auto put_c = [&](int off, int idx, double ang) {
        R v[2] = {(R)std::cos(ang), (R)std::sin(ang)};
        std::memcpy(blob.data() + off + (size_t)idx * 2 * sizeof(R), v, sizeof v);
    };
    auto put_r = [&](int off, int idx, double val) { R v = (R)val; std::memcpy(blob.data() + off + (size_t)idx * sizeof(R), &v, sizeof v); };
    auto put_i = [&](int off, int idx, int val) { std::memcpy(blob.data() + off + (size_t)idx * 4, &val, 4); };
    for (int k = 1; k <= 3; ++k) {
        for (int l = 0; l < 64; ++l) put_c(L.tw1, (k - 1) * 64 + l, -2.0 * PI * (double)(l * k) / 256.0);
        for (int m = 0; m < 16; ++m) put_c(L.tw2, (k - 1) * 16 + m, -2.0 * PI * (double)(m * k) / 64.0);
        for (int d = 0; d < 4; ++d) put_c(L.tw3, (k - 1) * 4 + d, -2.0 * PI * (double)(d * k) / 16.0);
    }
    std::vector<int> lane_of_kbase(64);
    for (int l = 0; l < 64; ++l) lane_of_kbase[kbase_of(l)] = l;
    for (int l = 0; l < 64; ++l) {
        const int kb = kbase_of(l);
        for (int j = 0; j < 2; ++j) put_c(L.w512, j * 64 + l, -2.0 * PI * (double)(kb + 64 * j) / 512.0);
        put_i(L.partner, l, kb == 0 ? 0 : lane_of_kbase[64 - kb]);
    }
    // table-driven logarithm (float64): interval i of the mantissa m in [0.5, 1) has centre c_i = (128.5 + i) / 256;
    // the table holds 1 / c_i (rounded) and -log of THAT rounded value, so log m = logc_i + log1p(m / c_i - 1) exactly
    if (sizeof(R) == 8)
        for (int i = 0; i < kLogTab; ++i) {
            const double inv_c = 256.0 / (128.5 + (double)i);
            const long double logc = -std::log((long double)inv_c);
            R v[2] = {(R)inv_c, (R)logc};
            std::memcpy(blob.data() + L.logtab + (size_t)i * 2 * sizeof(R), v, sizeof v);
        }
    // mel runs: lane reads slots [start, start + mel_pad) of the power spectrum, fully inside [0, kPowerSlots)
    for (int l = 0; l < 64; ++l) {
        int start = ppos<R>(seg_lo[l]);
        if (start + mel_pad > power_slots<R>()) start = power_slots<R>() - mel_pad;
        if (seg_f[l] < 0) start = 0;
        put_i(L.mel_start, l, start);
        for (int i = 0; i < mel_pad; ++i) {
            const int k = bin_of_slot<R>(start + i);
            double w = 0.0;
            if (seg_f[l] >= 0 && k >= seg_lo[l] && k < seg_lo[l] + seg_n[l]) w = mel_filters[(size_t)seg_f[l] * kBins + k];
            put_r(L.mel_w, i * 64 + l, w);
        }
    }
    for (int f = 0; f <= kMaxFilt; ++f) put_i(L.pstart, f, pstart[f]);
    // scipy.fftpack.dct(type=2, norm='ortho') as a matrix: y[c] = s_c sum_n x[n] cos(pi c (2n+1) / (2N))
    for (int l = 0; l < 64; ++l) {
        const int c = l >> 2, q = l & 3;
        for (int i = 0; i < dct_len; ++i) {
            const int n = dct_len * q + i;
            double w = 0.0;
            if (c < n_mfcc && n < n_filt) {
                const double sc = (c == 0) ? std::sqrt(1.0 / n_filt) : std::sqrt(2.0 / n_filt);
                w = sc * std::cos(PI * c * (2 * n + 1) / (2.0 * n_filt));
            }
            put_r(L.dct_w, i * 64 + l, w);
        }
    }
    return "";
}

}  // namespace pe_wave
