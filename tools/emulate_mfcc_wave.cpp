// CPU replay of the one-frame-per-wave MFCC kernel (mycroft_precise_amd/csrc/mfcc_wave_device.h): the SAME
// table blob (mfcc_wave_tables.h) and the SAME per-lane arithmetic (mfcc_wave_core.h), with the cross-lane steps
// (digit exchanges, mirror exchange, reductions) written as loops over 64 "lanes".  It exists so that every index
// convention of the kernel is checked in the build container, which has no GPU (tests/test_wave_emulator.py).
//
//   emulate_mfcc_wave <f64|f32> <n_filt> <n_mfcc> <log_mode> filters.bin frames.bin out.bin
//     filters.bin  [n_filt][257] float64        frames.bin  [N][512] int16 (the cropped frames)
//     out.bin      [N][n_mfcc + n_filt] float64 : MFCC coefficients, then the log-mel energies
//
// g++ -O2 -std=c++17 -I mycroft_precise_amd/csrc tools/emulate_mfcc_wave.cpp -o emulate_mfcc_wave
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "mfcc_wave_core.h"

using namespace pe_wave;

template <class R>
static int run(int n_filt, int n_mfcc, int log_mode, const char* ffilt, const char* fframes, const char* fout) {
    std::ifstream a(ffilt, std::ios::binary), b(fframes, std::ios::binary);
    std::vector<double> filt((size_t)n_filt * kBins);
    a.read(reinterpret_cast<char*>(filt.data()), (std::streamsize)(filt.size() * 8));
    if (!a) { std::fprintf(stderr, "short filter file\n"); return 2; }
    std::vector<int16_t> frames((std::istreambuf_iterator<char>(b)), {});       // bytes, fixed below
    b.clear(); b.seekg(0, std::ios::end);
    const size_t n_bytes = (size_t)b.tellg();
    b.seekg(0);
    frames.assign(n_bytes / 2, 0);
    b.read(reinterpret_cast<char*>(frames.data()), (std::streamsize)n_bytes);
    const size_t N = frames.size() / 512;

    std::vector<unsigned char> blob;
    Layout L;
    const std::string err = build<R>(filt.data(), n_filt, n_mfcc, blob, L);
    if (!err.empty()) { std::fprintf(stderr, "table build failed: %s\n", err.c_str()); return 3; }
    const Tab<R> t = bind<R>(blob.data(), L);
    std::fprintf(stderr, "tables: %d bytes, mel_len %d, dct_len %d, np_max %d\n", L.total, L.mel_len, L.dct_len, L.np_max);

    const R EPS = (R)2.220446049250313e-16;
    const R pscale = (R)((1.0 / 512.0) / 1073741824.0);       // int16 samples enter unscaled: 2^-30 folded in here
    std::vector<double> out(N * (size_t)(n_mfcc + n_filt));
    std::vector<R> S(kScratchReals);
    cx<R>* X = reinterpret_cast<cx<R>*>(S.data());
    for (size_t fr = 0; fr < N; ++fr) {
        const int16_t* x = frames.data() + fr * 512;
        Regs<R> v[64];
        for (int l = 0; l < 64; ++l)
            for (int a4 = 0; a4 < 4; ++a4) { const int n = l + 64 * a4; v[l].re[a4] = (R)x[2 * n]; v[l].im[a4] = (R)x[2 * n + 1]; }
        auto exchange = [&](int shift) {        // 4x4 transpose of (register) x (lane digit at `shift`)
            Regs<R> o[64];
            for (int l = 0; l < 64; ++l)
                for (int rp = 0; rp < 4; ++rp) {
                    const int sl = xchg_src_lane(l, shift, rp), sr = xchg_src_reg(l, shift);
                    o[l].re[rp] = v[sl].re[sr]; o[l].im[rp] = v[sl].im[sr];
                }
            for (int l = 0; l < 64; ++l) v[l] = o[l];
        };
        LaneConsts<R> lc[64];
        for (int l = 0; l < 64; ++l) lc[l] = lane_consts(t, l);
        for (int l = 0; l < 64; ++l) pass_a(v[l], lc[l]);
        exchange(4);
        for (int l = 0; l < 64; ++l) pass_b(v[l], lc[l]);
        exchange(2);
        for (int l = 0; l < 64; ++l) pass_c(v[l], lc[l]);
        exchange(0);
        for (int l = 0; l < 64; ++l) pass_d(v[l]);
        // mirror exchange through the scratch, as the kernel does it
        for (int l = 0; l < 64; ++l) { X[xchg_index(l, 0)] = {v[l].re[2], v[l].im[2]}; X[xchg_index(l, 1)] = {v[l].re[3], v[l].im[3]}; }
        cx<R> zq0[64], zq1[64];
        for (int l = 0; l < 64; ++l) {
            const int pl = lc[l].partner;
            zq0[l] = X[xchg_index(pl, 1)]; zq1[l] = X[xchg_index(pl, 0)];
            if (kbase_of(l) == 0) { zq0[l] = {v[l].re[0], v[l].im[0]}; zq1[l] = {v[l].re[3], v[l].im[3]}; }
        }
        R* P = S.data() + kPowerOff; R* PART = S.data() + kPartOff; R* LM = S.data() + kLogMelOff;
        R lane_sum[64];
        for (int l = 0; l < 64; ++l) {
            R pw[4]; int bins[4];
            split_power(v[l], zq0[l], zq1[l], lc[l].w512[0], lc[l].w512[1], pscale * (R)0.25, pw);
            power_bins<R>(l, bins);
            for (int j = 0; j < 4; ++j) P[bins[j]] = pw[j];
            lane_sum[l] = (pw[0] + pw[1]) + (pw[2] + pw[3]);
            if (kbase_of(l) == 0) { const R p128 = (v[l].re[2] * v[l].re[2] + v[l].im[2] * v[l].im[2]) * pscale; P[ppos<R>(128)] = p128; lane_sum[l] += p128; }
        }
        // xor butterfly over the wave (1, 2, 4, ..., 32), what __shfl_xor does
        for (int o = 1; o < 64; o <<= 1) { R nx[64]; for (int l = 0; l < 64; ++l) nx[l] = lane_sum[l] + lane_sum[l ^ o]; for (int l = 0; l < 64; ++l) lane_sum[l] = nx[l]; }
        for (int l = 0; l < 64; ++l) PART[l] = mel_run(t, P, l);
        R lm[65];
        auto safe = [&](R x) { return log_mode == 0 ? (x > EPS ? x : EPS) : (x == (R)0 ? EPS : x); };
        for (int f = 0; f < n_filt; ++f) lm[f] = wave_log(safe(filter_sum(t, PART, f)), t.logtab);
        lm[n_filt] = wave_log(safe(lane_sum[0]), t.logtab);
        for (int f = 0; f <= n_filt; ++f) LM[f] = lm[f];
        R part[64];
        for (int l = 0; l < 64; ++l) part[l] = dct_run(t, LM, l, n_filt);
        for (int c = 0; c < n_mfcc; ++c) {
            const R s = (part[4 * c] + part[4 * c + 1]) + (part[4 * c + 2] + part[4 * c + 3]);
            out[fr * (size_t)(n_mfcc + n_filt) + c] = (double)(c == 0 ? LM[n_filt] : s);
        }
        for (int f = 0; f < n_filt; ++f) out[fr * (size_t)(n_mfcc + n_filt) + n_mfcc + f] = (double)LM[f];
    }
    std::ofstream o(fout, std::ios::binary);
    o.write(reinterpret_cast<const char*>(out.data()), (std::streamsize)(out.size() * 8));
    return 0;
}

int main(int argc, char** argv) {
    if (argc != 8) { std::fprintf(stderr, "usage: %s <f64|f32> n_filt n_mfcc log_mode filters.bin frames.bin out.bin\n", argv[0]); return 1; }
    const int n_filt = std::atoi(argv[2]), n_mfcc = std::atoi(argv[3]), log_mode = std::atoi(argv[4]);
    if (std::string(argv[1]) == "f64") return run<double>(n_filt, n_mfcc, log_mode, argv[5], argv[6], argv[7]);
    return run<float>(n_filt, n_mfcc, log_mode, argv[5], argv[6], argv[7]);
}
