// MFCC front end for gfx950 (MI355X), FOUR frames per wave: int16 PCM -> 13 coefficients (round 5).
//
// What it computes is what mfcc_wave_device.h computes -- the reference's Vectorizer.mfccs entry
// (/root/reference/precise/vectorization.py:36-39 -> sonopy.mfcc_spec; log_mode 1: the legacy speechpy entry, :40-42) as driven
// by Listener.update_vectors (/root/reference/precise/network_runner.py:125-146) -- with another mapping to the machine
// (profiles/round5/r5_mfcc_f64_budget.txt is its instruction budget: one frame per wave spends 340 vector and 79 LDS
// instructions per frame at 64 lanes, a third of them with most lanes idle or on the cross-lane exchanges):
//   * a frame is the task of a 16-LANE GROUP; a wave works on four frames at once.  Lane j of a group holds the 16 complex points
//     z[16 n1 + j] (z[n] = x[2n] + i x[2n+1]) -- 16 coalesced 64-byte loads per frame;
//   * 256 = 16 x 16:  X[k1 + 16 k2] = sum_n2 W16^(n2 k2) . W256^(n2 k1) . sum_n1 W16^(n1 k1) z[16 n1 + n2]
//     = a 16-point transform in REGISTERS (two radix-4 stages, constants for the inner twiddles), 15 table twiddles, ONE 16 x 16
//     transpose of (register) x (lane) through the wave's LDS scratch, and a second 16-point transform in registers: no
//     permlane traffic, one exchange instead of three;
//   * real-FFT split: bin p = j + 16 k2 < 128 pairs with 256 - p = lane (16 - j) % 16, register 15 - k2: every lane fetches
//     the upper half of its partner's registers (8 values) and owns the power of 16 bins;
//   * power spectrum to LDS in the slot order the filterbank run tables of mfcc_wave_tables.h expect (the SAME tables as the
//     one-frame-per-wave kernel: four runs per lane instead of one), log on 16 lanes per round, DCT as one coefficient per lane.
#pragma once
#include "../../mycroft_precise_amd/csrc/mfcc_wave_device.h"

namespace pe {

// quad twiddle tables, built on the host (engine.hip): W256^(k1 j) [16 k1][16 j], then W512^p [128]
constexpr int kQuadTwElems = 16 * 16 + 128;
// per-wave LDS scratch.  Three phases reuse it: the transpose (per frame group 272 elements of 8 bytes: group offset = 544 dwords
// = 32 mod 64, so the two frame groups of a 32-lane ds_read_b64 group hit disjoint banks), the mirror exchange (per group 144
// complex: float64 at the stride of the last phase = 0 mod 64 dwords, as the 16-lane groups of ds_read_b128 want; float32 at
// 288 dwords = 32 mod 64) and power | partial sums (288 + 64 reals per group; the log-mel energies then overwrite the power slots).
constexpr int kQuadGroupReals = 368;
constexpr int kQuadPartOff = 288, kQuadLogMelOff = 0;      // (the log-mel energies overwrite the power spectrum, dead by then)
template <class R> constexpr int kQuadScratchBytes = sizeof(R) == 8 ? 4 * kQuadGroupReals * 8 : 4 * 272 * 8;
static_assert(4 * 272 * 8 <= 4 * kQuadGroupReals * 8 && 4 * 144 * 8 <= 4 * 272 * 8 && 4 * kQuadGroupReals * 4 <= 4 * 272 * 8, "scratch phases");

template <class R> struct QuadK;
template <> struct QuadK<double> { static constexpr double C1 = RealK<double>::C1, S1 = RealK<double>::S1, H = RealK<double>::H; };
template <> struct QuadK<float> { static constexpr float C1 = RealK<float>::C1, S1 = RealK<float>::S1, H = RealK<float>::H; };

// 16-point DFT of (re, im)[0..15] in place, natural order in and out: U[ka + 4 kb] = sum_nb W4^(nb kb) W16^(nb ka) sum_na W4^(na ka) u[4 na + nb]
template <class R>
__device__ __forceinline__ void quad_dft16(R (&re)[16], R (&im)[16]) {
    using K = QuadK<R>;
    using pe_wave::Regs;
    R tr[4][4], ti[4][4];           // [ka][nb]
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        Regs<R> t;
#pragma unroll
        for (int na = 0; na < 4; ++na) { t.re[na] = re[4 * na + nb]; t.im[na] = im[4 * na + nb]; }
        pe_wave::radix4(t);
#pragma unroll
        for (int ka = 0; ka < 4; ++ka) { tr[ka][nb] = t.re[ka]; ti[ka][nb] = t.im[ka]; }
    }
    // inner twiddles W16^(nb ka): m = 1 (C1, -S1), 2 (H, -H), 3 (S1, -C1), 4 (0, -1), 6 (-H, -H), 9 (-C1, S1)
    auto mul = [](R& a, R& b, const int m) {
        const R x = a, y = b;
        if (m == 1) { a = real_fma(x, K::C1, y * K::S1); b = real_fma(y, K::C1, -(x * K::S1)); }
        else if (m == 2) { a = (x + y) * K::H; b = (y - x) * K::H; }
        else if (m == 3) { a = real_fma(x, K::S1, y * K::C1); b = real_fma(y, K::S1, -(x * K::C1)); }
        else if (m == 4) { a = y; b = -x; }
        else if (m == 6) { a = (y - x) * K::H; b = -((x + y) * K::H); }
        else if (m == 9) { a = -real_fma(x, K::C1, y * K::S1); b = real_fma(x, K::S1, -(y * K::C1)); }
    };
    mul(tr[1][1], ti[1][1], 1); mul(tr[1][2], ti[1][2], 2); mul(tr[1][3], ti[1][3], 3);
    mul(tr[2][1], ti[2][1], 2); mul(tr[2][2], ti[2][2], 4); mul(tr[2][3], ti[2][3], 6);
    mul(tr[3][1], ti[3][1], 3); mul(tr[3][2], ti[3][2], 6); mul(tr[3][3], ti[3][3], 9);
#pragma unroll
    for (int ka = 0; ka < 4; ++ka) {
        Regs<R> s;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) { s.re[nb] = tr[ka][nb]; s.im[nb] = ti[ka][nb]; }
        pe_wave::radix4(s);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) { re[ka + 4 * kb] = s.re[kb]; im[ka + 4 * kb] = s.im[kb]; }
    }
}

// sum over the 16 lanes of a row (DPP), the same value in every lane of the row; fixed order
template <class R> __device__ __forceinline__ R quad_row_sum(R x) {
    x += dpp_mov<0xB1>(x);          // quad_perm:[1,0,3,2]
    x += dpp_mov<0x4E>(x);          // quad_perm:[2,3,0,1]
    x += dpp_mov<0x141>(x);         // row_half_mirror
    x += dpp_mov<0x140>(x);         // row_mirror
    return x;
}

// table entries a lane needs in every pass, read once per wave: the starts of its four filterbank runs, and for its two filters
// (j and j + 16) the first partial sum and their number
struct QuadRuns { int mel_start[4], p0[2], np[2]; };
template <class R>
__device__ __forceinline__ QuadRuns quad_runs(const pe_wave::Tab<R>& t, const int j, const int n_filt) {
    QuadRuns r;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) r.mel_start[rr] = t.mel_start[j + 16 * rr];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int f = j + 16 * u;
        r.p0[u] = f < n_filt ? t.pstart[f] : 0;
        r.np[u] = f < n_filt ? t.pstart[f + 1] - r.p0[u] : 0;
    }
    return r;
}

// One pass of a wave over FOUR frames.  raw[n1] = the int16 pair (samples 2 n, 2 n + 1 in the low / high half) of point
// n = 16 n1 + j of this lane's frame, zero beyond the frame length.  tab: the table image of the one-frame-per-wave kernel (its
// log table, filterbank runs, DCT weights); qtw / qw512: the quad twiddles (LDS); S: this wave's scratch.
// Returns coefficient c of the lane's frame in lane j == c (c < n_mfcc), 0 elsewhere.
template <class R, class SH>
__device__ __forceinline__ R mfcc_quad_frames(const pe_wave::Tab<R>& tab, const pe_wave::cx<R>* qtw, const pe_wave::cx<R>* qw512, unsigned char* S, const int lane,
                                              const int n_filt, const int n_mfcc, const QuadRuns& qr, const int (&raw)[16], const int n_pairs, const R pscale, const int log_mode) {
    using K = RealK<R>;
    using pe_wave::cx;
    const int j = lane & 15, G = lane >> 4;
    R re[16], im[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
        const int v = 16 * n1 + j < n_pairs ? raw[n1] : 0;        // (n_pairs: the frame's sample pairs, 0 for a lane group without a task)
        re[n1] = (R)(int)(short)(v & 0xffff);
        im[n1] = (R)(v >> 16);
    }
    PE_T(3);
    // ---- 256-point transform: 16 x 16 ------------------------------------------------------------------------------------
    quad_dft16<R>(re, im);
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) {
        const cx<R> w = qtw[k1 * 16 + j];
        const R a = re[k1], b = im[k1];
        re[k1] = a * w.x - b * w.y;
        im[k1] = a * w.y + b * w.x;
    }
    PE_T(4);
    if constexpr (sizeof(R) == 8) {
        R* T = reinterpret_cast<R*>(S) + G * 272;
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) T[k1 * 17 + j] = re[k1];
        group_sync();
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2) re[n2] = lds_read(&T[j * 17 + n2]);
        group_sync();
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) T[k1 * 17 + j] = im[k1];
        group_sync();
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2) im[n2] = lds_read(&T[j * 17 + n2]);
        group_sync();
    } else {
        cx<R>* T = reinterpret_cast<cx<R>*>(S) + G * 272;
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) T[k1 * 17 + j] = cx<R>{re[k1], im[k1]};
        group_sync();
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2) { const cx<R> v = lds_read(&T[j * 17 + n2]); re[n2] = v.x; im[n2] = v.y; }
        group_sync();
    }
    PE_T(5);
    quad_dft16<R>(re, im);             // lane j now holds Z[j + 16 k2] in register k2
    PE_T(6);
    // ---- mirror exchange: the upper half of the partner lane's registers -------------------------------------------------------
    cx<R>* MX = sizeof(R) == 8 ? reinterpret_cast<cx<R>*>(reinterpret_cast<R*>(S) + G * kQuadGroupReals) : reinterpret_cast<cx<R>*>(S) + G * 144;
#pragma unroll
    for (int s = 0; s < 8; ++s) MX[j * 9 + s] = cx<R>{re[8 + s], im[8 + s]};
    if (j == 0) {                      // lane 0 pairs with itself one register further on (256 - 16 k2 = 16 (16 - k2))
#pragma unroll
        for (int s = 0; s < 8; ++s) MX[s] = cx<R>{re[(9 + s) & 15], im[(9 + s) & 15]};
    }
    group_sync();
    const int pj = (16 - j) & 15;
    // ---- real split + power: bins p = j + 16 k2 and 256 - p (partner values and W512 twiddles fetched four pairs at a time) ------
    R* P = reinterpret_cast<R*>(S) + G * kQuadGroupReals;
    R* PART = P + kQuadPartOff;     // (273 power slots, rounded up)
    R* LM = P + kQuadLogMelOff;
    const R ps4 = pscale * R(0.25);
    R psum = R(0);
    R pp[8], pq[8];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        cx<R> zq[4], w5[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int k2 = 4 * half + u; zq[u] = lds_read(&MX[pj * 9 + (7 - k2)]); w5[u] = qw512[j + 16 * k2]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k2 = 4 * half + u;
            const R a = re[k2], b = im[k2], c = zq[u].x, d = zq[u].y;
            const R er = a + c, ei = b - d;             // 2 E
            const R orr = b + d, oi = c - a;            // 2 O = -i (Z[p] - conj Z[q])
            const R tr = orr * w5[u].x - oi * w5[u].y, ti = orr * w5[u].y + oi * w5[u].x;
            const R x1r = er + tr, x1i = ei + ti, x2r = er - tr, x2i = ei - ti;
            pp[k2] = (x1r * x1r + x1i * x1i) * ps4;
            pq[k2] = (x2r * x2r + x2i * x2i) * ps4;
            psum += pp[k2] + pq[k2];
        }
    }
    const R p128 = (re[8] * re[8] + im[8] * im[8]) * pscale;
    group_sync();                      // every lane has read its partner's values: the exchange area becomes the power spectrum
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) {
        const int p = j + 16 * k2, q = 256 - p;
        P[pe_wave::ppos<R>(p)] = pp[k2];
        P[pe_wave::ppos<R>(q)] = pq[k2];
    }
    if (j == 0) {
        P[pe_wave::ppos<R>(128)] = p128;
        psum += p128;
    }
    group_sync();
    PE_T(7);
    psum = quad_row_sum(psum);
    // ---- mel filterbank: the run tables of the one-frame-per-wave kernel, four runs per lane; every load of two runs is issued
    //      before the first multiply-add that needs one (two waves per SIMD hide no LDS round trip on their own) ---------------------
#pragma unroll
    for (int rp = 0; rp < 2; ++rp) {
        R wv[2][SH::MEL], pv[2][SH::MEL];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int l = j + 16 * (2 * rp + u);
            const int s = qr.mel_start[2 * rp + u];
#pragma unroll
            for (int i = 0; i < SH::MEL; ++i) { wv[u][i] = lds_read(&tab.mel_w[i * 64 + l]); pv[u][i] = lds_read(&P[s + i]); }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            R acc = R(0);
#pragma unroll
            for (int i = 0; i < SH::MEL; ++i) acc = real_fma(wv[u][i], pv[u][i], acc);
            PART[j + 16 * (2 * rp + u)] = acc;
        }
    }
    group_sync();
    PE_T(8);
    // ---- filter energies (the partial sums of a filter added in run order), total power, log: filters j and j + 16, side by side ----
    {
        R pv[2][SH::NP];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int i = 0; i < SH::NP; ++i) pv[u][i] = lds_read(&PART[qr.p0[u] + i]);     // (past a filter's last run: some finite slot of the group's area, not added)
        }
        R y[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            R x = R(0);
#pragma unroll
            for (int i = 0; i < SH::NP; ++i) x += i < qr.np[u] ? pv[u][i] : R(0);
            if (j + 16 * u == n_filt) x = psum;
            if (j + 16 * u > n_filt) x = R(1);
            y[u] = pe_wave::wave_log(log_mode == 0 ? (x > K::EPS ? x : K::EPS) : (x == R(0) ? K::EPS : x), tab.logtab);
        }
        group_sync();                    // (the log-mel energies overwrite the power slots: every partial sum has been read)
#pragma unroll
        for (int u = 0; u < 2; ++u) if (j + 16 * u <= n_filt) LM[j + 16 * u] = y[u];
    }
    group_sync();
    PE_T(9);
    // ---- DCT-II (ortho): lane c adds the terms of coefficient c in four quarters (the quarters and their order are those of the
    //      one-frame-per-wave kernel's lanes 4c .. 4c+3 and its quad reduction); c0 := log total power ---------------------------------
    R coeff;
    {
        const int dl = tab.dct_len;
        R lv[4][SH::DCT], dv[4][SH::DCT];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int i = 0; i < SH::DCT; ++i) {
                lv[q][i] = lds_read(&LM[dl * q + i]);           // (terms past n_filt: finite leftovers of the scratch times a zero weight)
                dv[q][i] = lds_read(&tab.dct_w[i * 64 + 4 * j + q]);
            }
        }
        const R c0 = lds_read(&LM[n_filt]);
        R part[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            part[q] = R(0);
#pragma unroll
            for (int i = 0; i < SH::DCT; ++i) part[q] = real_fma(dv[q][i], lv[q][i], part[q]);
        }
        coeff = (part[0] + part[1]) + (part[2] + part[3]);
        if (j == 0) coeff = c0;
    }
    PE_T(10);
    group_sync();                        // the scratch may be rewritten by the next pass
    return coeff;
}

// ---- frame tasks of the streaming engine, ONE update per call, sample pairs as dwords (the shape of every pe_update of 1024-sample
//      chunks): four (stream, frame row) tasks per pass.  Counters of up to 64 streams sit in two registers (lane i <-> stream
//      base + i) as in mfcc_frame_tasks; the streams of a pass are the next four set bits of the due mask. ------------------------
template <class R, class SH>
__device__ __forceinline__ void mfcc_quad_tasks(const MfccStreamArgs<R>& a, const pe_wave::Tab<R>& tab, const pe_wave::cx<R>* qtw, const pe_wave::cx<R>* qw512,
                                                unsigned char* S, const int first_wave, const int n_waves) {
    using K = RealK<R>;
    const StreamGeom& geo = a.geo;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, G = lane >> 4;
    const int C = a.chunk, hop = geo.hop, flen = geo.frame_len, slots = geo.ring_slots;
    const int per_wave = (geo.n_streams + n_waves - 1) / n_waves;
    const int s_begin = (first_wave + wave) * per_wave;
    const int s_end = s_begin + per_wave < geo.n_streams ? s_begin + per_wave : geo.n_streams;
    const QuadRuns qr = quad_runs<R>(tab, j, geo.n_filt);
    // task generator: streams in batches of 64 (lane = stream), frame index kb ascending, four due streams per pass.  The loads of a
    // pass are issued one pass ahead of the arithmetic that consumes them.
    int base = s_begin - 64, kb = 0, vq = 0, vkc = 0, nnew = 0, v_first = 0;
    bool in_batch = false;
    unsigned long long due = 0ull;
    // (fetch and copy_issue issue the SAME number of loads on every call, from clamped positions once their generator has run dry:
    //  the compiler's s_waitcnt counts are the minimum over all paths to a use, and one path that skips the loads -- "no next
    //  pass" -- makes every pass wait for the prefetch it has just issued)
    auto fetch = [&](int (&raw)[16], long long& cell) -> bool {
        bool any = true;
        while (!due) {
            if (in_batch) ++kb;
            else {
                base += 64;
                if (base >= s_end) { base = s_end; any = false; break; }
                const int s = base + lane;
                const int sc = s < s_end ? s : 0;
                vq = a.st_q[sc];
                vkc = (int)a.st_kc[sc];
                const int avail = vq + C;
                nnew = (s < s_end && avail >= flen) ? 1 + (int)a.div_hop.div((uint32_t)(avail - flen)) : 0;
                v_first = nnew > slots ? nnew - slots : 0;
                if (s < s_end) {                    // the counters of the batch, one stream per lane, as mfcc_book_tile advances them for one update
                    uint32_t ke = a.st_ke[s];
                    const int qu = avail - nnew * hop;
                    const uint32_t kcu = (uint32_t)vkc + (uint32_t)nnew;
                    const int m = qu + hop * (int)(kcu - ke);
                    if (m >= geo.window) ke += 1u + a.div_hop.div((uint32_t)(m - geo.window));
                    if (a.ke_hist) a.ke_hist[s] = ke;
                    a.st_q_next[s] = qu;
                    a.st_kc_next[s] = kcu;
                    a.st_ke_next[s] = ke;
                }
                __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0) HERE, on the rare path: otherwise every call waits for "maybe just loaded" counters
                kb = 0;
                in_batch = true;
            }
            if (__ballot(kb < nnew) == 0ull) { in_batch = false; continue; }
            due = __ballot(kb >= v_first && kb < nnew);
        }
        int pick[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            pick[g] = due ? __builtin_ctzll(due) : -1;
            if (due) due &= due - 1;
        }
        const int mine = G == 0 ? pick[0] : G == 1 ? pick[1] : G == 2 ? pick[2] : pick[3];
        const bool live = any && mine >= 0;
        const int src = live ? mine : 0;
        const int q = __builtin_amdgcn_ds_bpermute(src * 4, vq);
        const uint32_t kc = (uint32_t)__builtin_amdgcn_ds_bpermute(src * 4, vkc);
        const int st = live ? base + src : 0;
        // sample m of the frame (0 <= m < flen): m < qa from the carry (its sample vb + m), else from the chunk (off0 + m)
        const int vb = kb * hop, qa = q - vb, off0 = vb - q;
        const int16_t* car = a.carry + (size_t)st * kCarryCap + vb;
        const int16_t* row = a.pcm + (size_t)st * C + off0;
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            const int m = 2 * (16 * n1 + j);
            const int mm = m < flen ? m : 0;
            const int16_t* p = (live ? (mm < qa ? car : row) : a.pcm) + mm;
            raw[n1] = *reinterpret_cast<const int*>(p);           // (unconditional, from a clamped position, masked when consumed: the load stays in flight)
        }
        const int tile = st >> 4, js = st & 15;
        const int slot = (int)((kc + (uint32_t)kb) & (uint32_t)(slots - 1));
        cell = live ? (long long)(((size_t)tile * slots + slot) * kTileStreams + js) : -1ll;
        return any;
    };
    auto work = [&](const int (&raw)[16], const long long cell) {
        PE_T(2);
        const R coeff = mfcc_quad_frames<R, SH>(tab, qtw, qw512, S, lane, geo.n_filt, geo.n_mfcc, qr, raw, cell >= 0 ? flen >> 1 : 0, K::PSCALE_I16, geo.log_mode);
        if (cell >= 0) {
            const float xf = j < geo.n_mfcc ? (float)coeff : 0.0f;
            if (a.ring_bf16) reinterpret_cast<__bf16*>(a.ring)[(size_t)cell * kRowFloats + j] = (__bf16)xf;
            else a.ring[(size_t)cell * kRowFloats + j] = xf;
        }
        PE_T(11);
    };
    // two sample buffers in turn (no register copies), every generator call on every path
    auto passes = [&]() {
        int rawA[16], rawB[16];
        long long cellA = -1, cellB = -1;
        bool have = fetch(rawA, cellA);
        while (have) {
            const bool more = fetch(rawB, cellB);
            work(rawA, cellA);
            have = fetch(rawA, cellA);
            if (more) work(rawB, cellB);
        }
    };
    // (every other workgroup moves its leftovers BEFORE its passes: done by all waves at the same moment the role is a memory-bound
    //  phase of the whole machine with the vector pipes idle)
    const bool copy_first = (blockIdx.x & 1) != 0;
    if (!copy_first) passes();
    // leftover samples of the wave's own streams -> the other carry buffer, AFTER its passes: four streams per round (lane group =
    // stream), eight samples per lane and load as in mfcc_book_tile's 16-byte form, eight rounds' loads in flight before the first
    // store (the transform's registers are dead here).  The launch requires chunk >= frame length, which puts every leftover inside
    // the chunk (nnew hop > q + chunk - frame length >= q; no frame completed: q < frame length - chunk <= 0).  Bookkeeping
    // workgroups of their own would each reserve this kernel's LDS and run behind the frames instead of beside them; interleaved
    // with the passes the role costs 9 us at 65536 streams (profiles/round5/r5h_mfcc_quad.log).
    struct __attribute__((packed, aligned(4))) Pcm8 { int d[4]; };
    constexpr int kRounds = 8;
    for (int cb = s_begin; cb < s_end; cb += 64) {
        const int sl = cb + lane;
        const int cq = a.st_q[sl < s_end ? sl : 0];
        for (int r0 = 0; r0 < 16 && cb + 4 * r0 < s_end; r0 += kRounds) {
            Pcm8 v[kRounds][4];
            int tail[kRounds], full8[kRounds], rest[kRounds], strm[kRounds];
#pragma unroll
            for (int k = 0; k < kRounds; ++k) {
                const int src = 4 * (r0 + k) + G;
                const int st = cb + src;
                const bool valid = st < s_end;
                const int q = __builtin_amdgcn_ds_bpermute(src * 4, cq);
                const int avail = q + C;
                const int nnew_s = avail >= flen ? 1 + (int)a.div_hop.div((uint32_t)(avail - flen)) : 0;
                const int qn = valid ? avail - nnew_s * hop : 0;
                strm[k] = valid && qn > 0 ? st : -1;
                full8[k] = qn > 0 ? (qn & ~7) : 0;
                rest[k] = qn > 0 ? ((qn & 7) >> 1) : 0;
                const int16_t* row0 = a.pcm + (size_t)(valid ? st : 0) * C;
                const int16_t* srcp = row0 + (nnew_s * hop - q);           // sample 0 of the leftover (dereferenced only where qn > 0)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int n = 128 * c + 8 * j;
                    v[k][c] = *reinterpret_cast<const Pcm8*>(n < full8[k] ? srcp + n : row0);      // (unconditional, from a clamped position)
                }
                tail[k] = *reinterpret_cast<const int*>(j < rest[k] ? srcp + full8[k] + 2 * j : row0);
            }
#pragma unroll
            for (int k = 0; k < kRounds; ++k) {
                if (strm[k] >= 0) {
                    int16_t* const carw = a.carry_next + (size_t)strm[k] * kCarryCap;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int n = 128 * c + 8 * j;
                        if (n < full8[k]) *reinterpret_cast<int4*>(carw + n) = int4{v[k][c].d[0], v[k][c].d[1], v[k][c].d[2], v[k][c].d[3]};
                    }
                    if (j < rest[k]) *reinterpret_cast<int*>(carw + full8[k] + 2 * j) = tail[k];
                }
            }
        }
    }
    if (copy_first) passes();
}

}  // namespace pe
