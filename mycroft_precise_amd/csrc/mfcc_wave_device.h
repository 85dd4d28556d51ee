// MFCC front end for gfx950 (MI355X), one FRAME per WAVE: int16 PCM -> 13 coefficients.
//
// What it computes is the reference's Vectorizer.mfccs entry (/root/reference/precise/vectorization.py:36-39 ->
// third-party sonopy.mfcc_spec) -- or, with log_mode 1 and that library's filterbank, the legacy speechpy entry
// (:40-42) -- as driven by Listener.update_vectors (/root/reference/precise/network_runner.py:125-146):
//   frame = first n_fft(512) samples of each 1600-sample window (numpy's rfft(n=512) crop),
//   512-point real FFT -> power/512 -> triangular mel filters -> log -> DCT-II ortho -> coefficient 0 := log power.
//
// Mapping to the machine (the arithmetic per lane and the tables: mfcc_wave_core.h / mfcc_wave_tables.h, shared
// with the CPU replay in tools/emulate_mfcc_wave.cpp):
//   * a frame is a TASK of one wave: 4 complex points per lane (16 data VGPRs in float64), four radix-4 passes,
//     the (register x lane digit) transposes between them by v_permlane32/16_swap (digit b) and through 5 KB of
//     wave-private LDS (digits c, d), real-FFT split against the mirror lane, power spectrum to LDS, the sparse
//     mel filterbank as <= mel_len products per lane with the partial sums added in lane order, log on the lanes
//     that own a filter, DCT as dct_len products per lane + a quad reduction;
//   * one wave-instruction loads 256 contiguous bytes of a stream's PCM (4 per frame);
//   * nothing in a frame waits for another wave: LDS hand-offs are inside the wave (DS instructions of a wave
//     execute in order; only the compiler has to be kept from moving them), so a SIMD hides one frame's LDS
//     latency behind the other frames it holds (<= 128 VGPRs: four waves per SIMD);
//   * no MFMA here: byte shuffling and a small FFT, not a GEMM.
#pragma once
#include "pe_common.h"
#include "mfcc_wave_core.h"
#include "mfcc_device.h"        // RealK, real_log, group_sync, PcmView helpers shared with the bookkeeping kernel

namespace pe {

#ifndef PE_TW_LDS
#define PE_TW_LDS 2             // 1: the W64 / W16 twiddles (16 / 4 distinct values) are read from LDS per frame instead of
#endif                          //    living in 24 registers (measured: 109 vs 126 us per update at 65536 streams -- the registers
                                //    buy a fourth wave per SIMD); 2: so are the per-lane W256 / W512 ones (20 more registers
                                //    in float64: what keeps the frame loop free of scratch spills under the 128-register cap)
#ifndef PE_XCHG_B_LDS
#define PE_XCHG_B_LDS 0         // 1: digit b also goes through LDS (debug / cross-check of the permlane path)
#endif

// One 8-byte LDS element per instruction.  Two adjacent 8-byte reads merged into ds_read2_b64 cost 8 LDS cycles against
// 2 + 2 for two ds_read_b64 (MI355X_MICROARCH: read2_b64 is serviced as 2 x 4 groups of 16 lanes, b64 as 2 groups of
// 32), and the compiler merges whenever it can: a volatile access is the one thing it leaves alone (measured, float64
// front end at 65536 streams: 90.8 vs 99.2 us).  4-byte elements are read plainly: ds_read2_b32 costs what two
// ds_read_b32 cost, and the volatile order only gets in the scheduler's way (61.0 vs 59.0 us).
#ifndef PE_LDS_NO_MERGE
#define PE_LDS_NO_MERGE 1
#endif
template <class T> __device__ __forceinline__ T lds_read(const T* p) {
    if constexpr (PE_LDS_NO_MERGE && sizeof(T) == 8) {
        typedef const volatile __attribute__((address_space(3))) unsigned long long* lds_ptr;   // (a volatile GENERIC access would become a flat load)
        const unsigned long long bits = *(lds_ptr)p;
        return __builtin_bit_cast(T, bits);
    } else {
        return *p;
    }
}

constexpr int kWaveScratchReals = pe_wave::kScratchReals;

// The LDS image of a workgroup: the blob from the logarithm table on (mel / DCT weights, run starts, ...); the twiddle
// sections before it are read once per wave straight from global memory into registers (LaneConsts).
__host__ __device__ inline int wave_lds_skip(const pe_wave::Layout& L) { return PE_TW_LDS == 2 ? L.tw1 : PE_TW_LDS ? L.tw2 : L.logtab; }
__host__ __device__ inline size_t wave_lds_bytes(int real_size, const pe_wave::Layout& L, int waves) {
    return (size_t)(L.total - wave_lds_skip(L)) + (size_t)waves * kWaveScratchReals * real_size;
}
// LDS of a frame workgroup: [scratch of wave 0 .. 3][table image].  The scratch comes first so that neither base depends
// on the image size, and the section offsets of the image are compile-time constants of the table shape SH (the host builds
// the blob with the same constexpr pe_wave::layout; launch_* refuse a blob that disagrees): every table address in the frame
// loop is an instruction offset -- with runtime offsets the twelve section pointers lived in SGPRs, spilled to VGPR lanes
// and were fetched back by v_readlane in every frame (round 3: 60 of the 223 instructions between two transforms).
template <class R> constexpr int kWaveImageBase = kFrameWaves * kWaveScratchReals * (int)sizeof(R);
template <class R, class SH> constexpr pe_wave::Layout shape_layout() { return pe_wave::layout((int)sizeof(R), SH::MEL, SH::DCT, 0, 0); }
template <class R, class SH>
__host__ __device__ inline bool shape_layout_matches(const pe_wave::Layout& L) {
    constexpr pe_wave::Layout K = shape_layout<R, SH>();
    return L.tw1 == K.tw1 && L.tw2 == K.tw2 && L.tw3 == K.tw3 && L.w512 == K.w512 && L.logtab == K.logtab && L.mel_w == K.mel_w &&
           L.dct_w == K.dct_w && L.mel_start == K.mel_start && L.pstart == K.pstart && L.partner == K.partner && L.proj_w == K.proj_w;
}
template <class R, class SH>
__device__ __forceinline__ pe_wave::Tab<R> wave_bind(const unsigned char* image, const pe_wave::Layout& L) {
    pe_wave::Layout K = shape_layout<R, SH>();
    K.proj_b = L.proj_b; K.proj_rows = L.proj_rows; K.total = L.total;
    K.mel_pad = L.mel_pad; K.dct_pad = L.dct_pad; K.np_pad = L.np_pad; K.mel_len = L.mel_len; K.dct_len = L.dct_len; K.np_max = L.np_max;
    return pe_wave::bind<R>(image, K, wave_lds_skip(K));
}

// workgroup-wide copy of the table image into LDS in two steps, so that the global loads can be issued at the very
// top of a kernel and the LDS stores + barrier placed where the tables are first needed
// 16-byte pieces per thread held in flight (256 threads: 20 / 12 / 8 KB; the stock float64 image is 16.6 KB, the float32 one 8.3 KB)
template <class R> constexpr int kTabRegs = PE_TW_LDS == 2 ? (sizeof(R) == 8 ? 5 : 3) : 2;
struct TabRegs { uint4 v0, v1, v2, v3, v4; };       // (named members, not an array: the array form ended up in scratch memory)

// UNCONDITIONAL loads from clamped indices, unconditional stores to clamped indices: a load inside an exec-masked block
// makes the compiler wait for every outstanding load (s_waitcnt vmcnt(0)) before the next such block -- the five pieces,
// the stream counters and the first frame's samples were seven round trips in series at the top of every frame wave
// (ISA, round 3).  The surplus threads hold piece 0 and store it where it belongs.
template <class R>
__device__ __forceinline__ TabRegs wave_tables_issue(const WaveTables<R>& g) {
    const int skip = wave_lds_skip(g.L);
    const int n16 = (g.L.total - skip) >> 4;
    const uint4* src = reinterpret_cast<const uint4*>(static_cast<const unsigned char*>(g.blob) + skip);
    auto piece = [&](int k) -> uint4 {
        const int i = threadIdx.x + k * blockDim.x;
        return src[i < n16 ? i : 0];
    };
    TabRegs t;
    t.v0 = piece(0); t.v1 = piece(1);
    if constexpr (kTabRegs<R> > 2) t.v2 = piece(2);
    if constexpr (kTabRegs<R> > 3) { t.v3 = piece(3); t.v4 = piece(4); }
    return t;
}

template <class R>
__device__ __forceinline__ void wave_tables_commit(unsigned char* smem, const WaveTables<R>& g, const TabRegs& t) {
    const int skip = wave_lds_skip(g.L);
    const int n16 = (g.L.total - skip) >> 4;
    const uint4* src = reinterpret_cast<const uint4*>(static_cast<const unsigned char*>(g.blob) + skip);
    uint4* dst = reinterpret_cast<uint4*>(smem + kWaveImageBase<R>);
    auto put = [&](int k, const uint4& v) {
        const int i = threadIdx.x + k * blockDim.x;
        dst[i < n16 ? i : 0] = v;
    };
    put(0, t.v0); put(1, t.v1);
    if constexpr (kTabRegs<R> > 2) put(2, t.v2);
    if constexpr (kTabRegs<R> > 3) { put(3, t.v3); put(4, t.v4); }
    for (int i = threadIdx.x + kTabRegs<R> * blockDim.x; i < n16; i += blockDim.x) dst[i] = src[i];     // (larger filterbanks)
    __syncthreads();
}

__device__ __forceinline__ void pl32_swap(unsigned& a, unsigned& b) {
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);      // a[32..63] <-> b[0..31]
    a = r[0]; b = r[1];
}
__device__ __forceinline__ void pl16_swap(unsigned& a, unsigned& b) {
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);      // a[16..31] <-> b[0..15], a[48..63] <-> b[32..47]
    a = r[0]; b = r[1];
}
__device__ __forceinline__ void swap32(double& a, double& b) {
    unsigned al = (unsigned)__double2loint(a), ah = (unsigned)__double2hiint(a), bl = (unsigned)__double2loint(b), bh = (unsigned)__double2hiint(b);
    pl32_swap(al, bl); pl32_swap(ah, bh);
    a = __hiloint2double((int)ah, (int)al); b = __hiloint2double((int)bh, (int)bl);
}
__device__ __forceinline__ void swap16(double& a, double& b) {
    unsigned al = (unsigned)__double2loint(a), ah = (unsigned)__double2hiint(a), bl = (unsigned)__double2loint(b), bh = (unsigned)__double2hiint(b);
    pl16_swap(al, bl); pl16_swap(ah, bh);
    a = __hiloint2double((int)ah, (int)al); b = __hiloint2double((int)bh, (int)bl);
}
__device__ __forceinline__ void swap32(float& a, float& b) {
    unsigned x = __float_as_uint(a), y = __float_as_uint(b);
    pl32_swap(x, y);
    a = __uint_as_float(x); b = __uint_as_float(y);
}
__device__ __forceinline__ void swap16(float& a, float& b) {
    unsigned x = __float_as_uint(a), y = __float_as_uint(b);
    pl16_swap(x, y);
    a = __uint_as_float(x); b = __uint_as_float(y);
}

// 4x4 transpose of (register index) x (lane digit b = lane bits 5:4) without LDS:
// bit 1 of the register index <-> lane bit 5, then bit 0 <-> lane bit 4
template <class R>
__device__ __forceinline__ void exchange_b(pe_wave::Regs<R>& v) {
    swap32(v.re[0], v.re[2]); swap32(v.im[0], v.im[2]);
    swap32(v.re[1], v.re[3]); swap32(v.im[1], v.im[3]);
    swap16(v.re[0], v.re[1]); swap16(v.im[0], v.im[1]);
    swap16(v.re[2], v.re[3]); swap16(v.im[2], v.im[3]);
}

// the same transpose for any lane digit through the wave's scratch (X: [64][5] complex)
template <class R>
__device__ __forceinline__ void exchange_lds(pe_wave::Regs<R>& v, pe_wave::cx<R>* X, int lane, int shift) {
#pragma unroll
    for (int r = 0; r < 4; ++r) X[pe_wave::xchg_index(lane, r)] = pe_wave::cx<R>{v.re[r], v.im[r]};
    group_sync();
    const int sr = pe_wave::xchg_src_reg(lane, shift);
#pragma unroll
    for (int rp = 0; rp < 4; ++rp) {
        const pe_wave::cx<R> z = lds_read(&X[pe_wave::xchg_index(pe_wave::xchg_src_lane(lane, shift, rp), sr)]);
        v.re[rp] = z.x; v.im[rp] = z.y;
    }
    group_sync();
}

// Sum over the 64 lanes, result in every lane, without LDS: xor-butterfly inside the quads (quad_perm), mirrored halves
// and rows (row_half_mirror, row_mirror), then the two row-pair steps with v_permlane16/32_swap.  The order of the
// additions is fixed (deterministic).  A float64 shuffle through ds_bpermute costs an LDS round trip per step.
template <int CTRL> __device__ __forceinline__ double dpp_mov(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <class R> __device__ __forceinline__ R wave_sum(R x) {
    x += dpp_mov<0xB1>(x);          // quad_perm:[1,0,3,2]
    x += dpp_mov<0x4E>(x);          // quad_perm:[2,3,0,1]
    x += dpp_mov<0x141>(x);         // row_half_mirror
    x += dpp_mov<0x140>(x);         // row_mirror: every lane of a row now holds the row's sum
    R a = x, b = x;
    swap16(a, b);                   // a: rows (0, 0, 2, 2), b: rows (1, 1, 3, 3)
    x = a + b;
    a = x; b = x;
    swap32(a, b);                   // a: lower half everywhere, b: upper half everywhere
    return a + b;
}

// buffer descriptor over `bytes` bytes at p; every input is forced into SGPRs (wave-uniform by construction)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wave_rsrc(const void* p, int bytes) {
    const uintptr_t u = reinterpret_cast<uintptr_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uintptr_t)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// The exchange area has one complex slot per lane that no exchange ever writes (the stride-5 padding).  The tail of a
// frame may READ such a slot as a term with weight zero (DCT terms past n_filt in the wide table shape), so it must
// hold a finite number: zeroed once per wave -- left alone it is whatever the previous workgroup left in LDS.
template <class R>
__device__ __forceinline__ void wave_scratch_init(R* S, int lane) {
    reinterpret_cast<pe_wave::cx<R>*>(S)[pe_wave::xchg_index(lane, pe_wave::kXchgStride - 1)] = pe_wave::cx<R>{R(0), R(0)};
    group_sync();
}

// One frame on one wave.  pcm[a] = the int16 pair (samples 2n, 2n+1 in the low / high half) of point n = lane + 64 a,
// already zero beyond the frame length -- or, FROM_REAL, re[]/im[] hold the samples as reals (offline form).
// Returns coefficient c in the four lanes 4c..4c+3 (c < n_mfcc).  S: this wave's scratch; after the call
// S[kLogMelOff + f] holds the log-mel energy of filter f.
// SH: compile-time bounds of the three table-driven loops (the tables are zero-padded up to them on the host)
struct ShapeStock { static constexpr int MEL = 10, DCT = 5, NP = 8, WPE = 4; };      // 20 filters (sonopy 9 / 5 / 8, speechpy 4 / 5 / 7)
struct ShapeAny { static constexpr int MEL = 16, DCT = 16, NP = 16, WPE = 3; };       // (its longer table loops want more than 128 registers)

// table entries a lane needs in every frame, read once per wave (the compiler cannot keep an LDS read across the
// scratch writes of a frame on its own): each saves a dependent LDS round trip per frame
struct LaneRuns { int mel_start, p0, np, partner; };
template <class R>
__device__ __forceinline__ LaneRuns lane_runs(const pe_wave::Tab<R>& t, int lane, int n_filt) {
    LaneRuns r;
    r.mel_start = t.mel_start[lane];
    const int f = lane < n_filt ? lane : 0;
    r.p0 = t.pstart[f];
    r.np = t.pstart[f + 1] - r.p0;
    r.partner = t.partner[lane];
    return r;
}

template <class R, class SH>
__device__ __forceinline__ R mfcc_wave_frame(const pe_wave::Tab<R>& t, const pe_wave::LaneConsts<R>& lc, const LaneRuns& lr, R* S, const int lane,
                                             const int n_filt, const int n_mfcc, pe_wave::Regs<R>& v, const R pscale, const int log_mode) {
    using K = RealK<R>;
    using namespace pe_wave;
    cx<R>* X = reinterpret_cast<cx<R>*>(S);
#if PE_TW_LDS == 2
    { radix4(v); twiddle3(v, lds_read(&t.tw1[lane]), lds_read(&t.tw1[64 + lane]), lds_read(&t.tw1[128 + lane])); }
#else
    pass_a(v, lc);
#endif
#if PE_XCHG_B_LDS
    exchange_lds(v, X, lane, 4);
#else
    exchange_b(v);
#endif
#if PE_TW_LDS
    { radix4(v); const int m = lane & 15; twiddle3(v, lds_read(&t.tw2[m]), lds_read(&t.tw2[16 + m]), lds_read(&t.tw2[32 + m])); }
    exchange_lds(v, X, lane, 2);
    { radix4(v); const int d = lane & 3; twiddle3(v, lds_read(&t.tw3[d]), lds_read(&t.tw3[4 + d]), lds_read(&t.tw3[8 + d])); }
#else
    pass_b(v, lc);
    exchange_lds(v, X, lane, 2);
    pass_c(v, lc);
#endif
    exchange_lds(v, X, lane, 0);
    pass_d(v);
    // mirror exchange: bins 256 - p of this lane's registers 0 / 1 are registers 3 / 2 of the partner lane
    X[xchg_index(lane, 0)] = cx<R>{v.re[2], v.im[2]};
    X[xchg_index(lane, 1)] = cx<R>{v.re[3], v.im[3]};
    group_sync();
    const int pl = lr.partner;
    cx<R> zq0 = lds_read(&X[xchg_index(pl, 1)]), zq1 = lds_read(&X[xchg_index(pl, 0)]);
#if PE_TW_LDS == 2
    const cx<R> w0 = lds_read(&t.w512[lane]), w1 = lds_read(&t.w512[64 + lane]);
#else
    const cx<R> w0 = lc.w512[0], w1 = lc.w512[1];
#endif
    group_sync();
    const bool lane0 = kbase_of(lane) == 0;
    if (lane0) { zq0 = cx<R>{v.re[0], v.im[0]}; zq1 = cx<R>{v.re[3], v.im[3]}; }
    R pw[4];
    split_power(v, zq0, zq1, w0, w1, pscale * R(0.25), pw);
    R* P = S + kPowerOff;
    R* PART = S + kPartOff;
    R* LM = S + kLogMelOff;
    int bins[4];
    power_bins<R>(lane, bins);
#pragma unroll
    for (int j = 0; j < 4; ++j) P[bins[j]] = pw[j];
    R psum = (pw[0] + pw[1]) + (pw[2] + pw[3]);
    if (lane0) {
        const R p128 = (v.re[2] * v.re[2] + v.im[2] * v.im[2]) * pscale;
        P[ppos<R>(128)] = p128;
        psum += p128;
    }
    group_sync();
    // mel filterbank: this lane's run of one filter (all table reads first, then the FMA chain)
    {
        const int s = lr.mel_start;
        constexpr int HALF = (SH::MEL + 1) / 2;        // two batches of reads: half the registers in flight
        R acc = R(0);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            R pv[HALF], wv[HALF];
#pragma unroll
            for (int u = 0; u < HALF; ++u) {
                const int i = h * HALF + u < SH::MEL ? h * HALF + u : SH::MEL - 1;
                pv[u] = lds_read(&P[s + i]);
                wv[u] = h * HALF + u < SH::MEL ? lds_read(&t.mel_w[i * 64 + lane]) : R(0);
            }
            if (h == 0) psum = wave_sum(psum);          // (rides in the shadow of the LDS reads)
#pragma unroll
            for (int u = 0; u < HALF; ++u) acc = real_fma(wv[u], pv[u], acc);
        }
        group_sync();
        PART[lane] = acc;
    }
    group_sync();
    // filter energies (partial sums of a filter sit in consecutive lanes: added in lane order), total power on
    // the last lane, one log pass for both
    {
        const bool has_filter = lane < n_filt;
        const bool takes_total = lane == 63 && n_filt < 64;
        R x = R(1);
        if (has_filter) {
            const int p0 = lr.p0, np = lr.np;
            R pv[SH::NP];
            // (past a filter's last run the read goes to a slot that holds 0 -- the exchange padding of lane 63, which
            //  nothing overwrites after wave_scratch_init -- instead of selecting on the 8-byte value afterwards: one
            //  32-bit select per term instead of two)
            constexpr int kZero = 2 * (63 * pe_wave::kXchgStride + pe_wave::kXchgStride - 1);
            static_assert(kZero >= pe_wave::kLogMelOff + pe_wave::kMaxFilt + 1 && kZero + 1 < kWaveScratchReals, "the zero slot lies behind the log-mel area");
#pragma unroll
            for (int i = 0; i < SH::NP; ++i) pv[i] = lds_read(i < np ? &PART[p0 + i] : &S[kZero]);
            x = R(0);
#pragma unroll
            for (int i = 0; i < SH::NP; ++i) x += pv[i];
        }
        if (takes_total) x = psum;
        if (has_filter || takes_total) {
            // sonopy clips at eps (safe_log); speechpy replaces exact zeros only (zero_handling): 0 < x < eps stays x
            const R y = wave_log(log_mode == 0 ? (x > K::EPS ? x : K::EPS) : (x == R(0) ? K::EPS : x), t.logtab);
            LM[takes_total ? n_filt : lane] = y;
        }
        if (n_filt == 64 && lane == 0) {    // no spare lane: a second log for the total
            LM[64] = wave_log(log_mode == 0 ? (psum > K::EPS ? psum : K::EPS) : (psum == R(0) ? K::EPS : psum), t.logtab);
        }
    }
    group_sync();
    // DCT-II (ortho): lane 4c + q adds its dct_len terms of coefficient c; quad reduction; c0 := log total power
    R part = R(0);
    {
        const int q = lane & 3;
        R lv[SH::DCT], dv[SH::DCT];
#pragma unroll
        for (int i = 0; i < SH::DCT; ++i) {
            lv[i] = lds_read(&LM[t.dct_len * q + i]);           // (terms past n_filt: finite leftovers of the scratch times a zero weight)
            dv[i] = lds_read(&t.dct_w[i * 64 + lane]);
        }
#pragma unroll
        for (int i = 0; i < SH::DCT; ++i) part = real_fma(dv[i], lv[i], part);
    }
    const R c0 = LM[n_filt];
    part += dpp_mov<0xB1>(part);            // quad_perm:[1,0,3,2]
    part += dpp_mov<0x4E>(part);            // quad_perm:[2,3,0,1]: all four lanes of a quad hold the coefficient
    group_sync();                           // the scratch may be rewritten by the next frame
    return lane < 4 ? c0 : part;
}

// ---- frame tasks of the streaming engine -------------------------------------------------------------------
// Which samples form which frame is closed-form integer arithmetic over the virtual stream
//     [carry (q samples)] ++ chunk 0 ++ chunk 1 ++ ... ++ chunk n_updates-1,
// so the frames a call completes are independent tasks (row kb, stream): frame kb of that stream, if the call
// completes that many.  Every wave owns a contiguous run of streams and works through their due frames; the
// bookkeeping (leftover samples, counters, per-update emitted-frame history) is a separate small role
// (mfcc_book_tile), which writes the OTHER carry buffer.
//
// Latency plan: a wave keeps the NEXT due frame's PCM (4-8 dwords per lane) in flight while it transforms the
// current frame, and nothing between the request and the conversion of those samples waits on the vector-memory
// counter (the task generator reads its counters from registers), so only the first frame of a wave waits for HBM.
template <class R>
struct FrameTask {          // wave-uniform description of one due frame
    const int16_t* car;     // this stream's carry
    const int16_t* row;     // chunk that holds the frame's first new sample (u0)
    float* ring_row;        // where the coefficients go
    float* proj_row;        // where the input projection of the frame goes (may be null)
    int vb, q, off0;        // first virtual sample; carry length; offset of vb in chunk u0 (negative: inside the carry)
};

// a frame's int16 sample pairs in flight: up to two bounds-checked parts per register (OR-ed when converted)
struct PcmRegs { int a0, a1, a2, a3, b0, b1, b2, b3; bool two; };

// the rare PCM path (odd chunk lengths, unaligned buffers, chunks shorter than a frame): one sample at a time
struct RawFrame { int v[4]; };

// (plain arguments: a FrameTask passed by value reserved 48 bytes of private memory per lane in the calling kernels)
__device__ __attribute__((noinline)) RawFrame fetch_frame_slow(const int16_t* car, const int16_t* row, int vb, int q, int off0,
                                                               int lane, int flen, int C, size_t update_stride) {
    RawFrame out;
    auto vsample = [&](int vv) -> int {
        if (vv < q) return (int)car[vv];
        int w = off0 + (vv - vb);
        const int16_t* r = row;
        while (w >= C) { w -= C; r += update_stride; }
        return (int)r[w];
    };
    for (int a4 = 0; a4 < 4; ++a4) {
        const int n = 2 * (lane + 64 * a4);
        const int lo = n < flen ? (vsample(vb + n) & 0xffff) : 0;
        const int hi = n + 1 < flen ? vsample(vb + n + 1) : 0;
        out.v[a4] = lo | (hi << 16);
    }
    return out;
}

// SIMD of the compute unit this wave runs on (HW_ID.SIMD_ID: s_getreg_b32 hwreg(HW_REG_HW_ID, 4, 2))
__device__ __forceinline__ int wave_simd_id() { return (int)(__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4) & 3); }

// by_simd (fused launch at one network tile per compute unit): the network workgroup of the compute unit has its four
// roles on SIMDs 0..3 in a fixed order (fused_update_kernel), and a frame wave's float64 multiply-adds wait while the
// matrix pipe of its SIMD runs the MFMAs of that role (R 37 % of the time, Z1 33 %, P 29 %, Z2 20 %: measured frame
// waves run 1.1x to 2x longer depending on the SIMD).  So the four waves of a frame workgroup split the workgroup's
// slots 3 : 4 : 5 : 4 by the SIMD they sit on instead of evenly.
// SINGLE: the call is one update (n_updates == 1: every fused launch) and, NOPROJ, stores no projection rows: both known at
// compile time there, which removes the several-updates arithmetic (chunk index of a frame, a frame's tail in the next
// update's row) and the projection block from the frame loop.
template <class R, class SH, bool SINGLE = false, bool NOPROJ = false>
__device__ __forceinline__ void mfcc_frame_tasks(const MfccStreamArgs<R>& a, const WaveTables<R>& wt, unsigned char* smem,
                                                 const int first_task, const int task_stride, const bool by_simd = false) {
    using K = RealK<R>;
    const StreamGeom& geo = a.geo;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_kb = a.n_frame_rows;
    const int C = a.chunk, hop = geo.hop, flen = geo.frame_len, slots = geo.ring_slots, U = SINGLE ? 1 : a.n_updates;
    const size_t update_stride = (size_t)geo.n_streams * C;
    // dword loads of (even, odd) sample pairs need every quantity that shifts a pair boundary to be even; a frame
    // may cross at most one chunk boundary (always true for a single update: its samples are carry ++ one chunk)
    const bool pairs = a.pcm_pairs_ok && ((hop | C | flen) & 1) == 0 && (C >= flen || U == 1);

    // ---- task stream of this wave: a contiguous run of (stream, row parity) SLOTS sigma = 2 s + (kb & 1), i.e. of a
    //      stream either all of its due frame rows or only the even / odd ones -- an update in which every stream
    //      completes two frames then spreads evenly over 1.5 x as many waves as there are streams (lock-step batches:
    //      22.4 -> us for the two-frame updates at 4096 streams), and large batches still get whole streams.  The
    //      counters of up to 64 streams sit in two registers (lane i <-> stream base + i), how many frames each
    //      completes is computed per lane, and the due (row, stream) pairs of a row are the set bits of one ballot:
    //      picking the next task costs a few scalar instructions and no memory access ------------------------------
    const int n_waves = task_stride;
    const long long n_slots = 2LL * geo.n_streams;
    const int per_wave = (int)((n_slots + n_waves - 1) / n_waves);
    long long sg_begin = (long long)(first_task + wave) * per_wave;
    long long sg_end = sg_begin + per_wave < n_slots ? sg_begin + per_wave : n_slots;
    if (by_simd) {
        // one int per wave at the base of its (not yet used) scratch: which SIMD each wave of this workgroup sits on
        int* const slot = reinterpret_cast<int*>(smem);
        constexpr int kStride = kWaveScratchReals * (int)sizeof(R) / 4;
        const int simd = wave_simd_id();
        if (lane == 0) slot[wave * kStride] = simd;
        __syncthreads();
        int seen = 0;
#pragma unroll
        for (int w = 0; w < kFrameWaves; ++w) seen |= 1 << slot[w * kStride];
        __syncthreads();                    // (the slots are scratch again)
        if (seen == 15) {                   // four waves on four SIMDs (always, as far as observed; else: the even split)
            const long long wg_begin = (long long)first_task * per_wave;
            long long wg_end = wg_begin + (long long)kFrameWaves * per_wave;
            wg_end = wg_end < n_slots ? wg_end : n_slots;
            const long long len = wg_end > wg_begin ? wg_end - wg_begin : 0;
            // cumulative shares in sixteenths by SIMD: 3 | 4 | 5 | 4 (the wave on SIMD 0 sits beside the critical wave)
#ifndef PE_SIMD_SPLIT_A
#define PE_SIMD_SPLIT_A 3       // cumulative sixteenths: A | B | C | 16 (tuning knob; measured again in round 4: r4p)
#define PE_SIMD_SPLIT_B 7
#define PE_SIMD_SPLIT_C 12
#endif
            const int lo16 = simd == 0 ? 0 : simd == 1 ? PE_SIMD_SPLIT_A : simd == 2 ? PE_SIMD_SPLIT_B : PE_SIMD_SPLIT_C;
            const int hi16 = simd == 0 ? PE_SIMD_SPLIT_A : simd == 1 ? PE_SIMD_SPLIT_B : simd == 2 ? PE_SIMD_SPLIT_C : 16;
            sg_begin = wg_begin + (len * lo16 + 8) / 16;
            sg_end = wg_begin + (len * hi16 + 8) / 16;
        }
    }
    const int s_begin = (int)(sg_begin >> 1);
    const int s_end = sg_end > sg_begin ? (int)((sg_end + 1) >> 1) : s_begin;
    int base = s_begin, kb_next = -1;
    int vq = 0, vkc = 0, v_first = 0, v_nnew = 0;           // per lane: counters, first frame row worth computing, frames completed
    int vss = 0;                                            // per lane: (stream << 1) | current side of its state
    unsigned long long due = 0;
    // lane i <-> position base + i of this launch; its stream is ids[base + i] (pe_update_subset) or the position itself
    auto load_counters = [&]() {
        const int s = base + lane;
        const int sc = s < s_end ? s : 0;                       // (unconditional loads: see wave_tables_issue)
        // (two requests, one per branch: a stream id that is either loaded or computed would be waited for at the join of the
        //  branches -- together with everything else in flight, the table image included -- before the records could be requested)
        int sid;
        RecPair both;
        if (a.ids) { sid = a.ids[sc]; both = rec_request(a.st.rec, a.st.n_padded, sid); }
        else { sid = sc; both = rec_request(a.st.rec, a.st.n_padded, sc); asm volatile("; streams in order"); }   // (the marker keeps the compiler from merging the two requests back into one behind the join)
        const int side = rec_side(both, a.st.call);
        vq = side ? both.r1.q : both.r0.q; vkc = (int)(side ? both.r1.kc : both.r0.kc);
        vq = s < s_end ? vq : 0; vkc = s < s_end ? vkc : 0;
        vss = (sid << 1) | side;
    };
    auto lane_frames = [&]() {                                // (after the counters arrived)
        const int avail = vq + U * C;
        v_nnew = (base + lane < s_end && avail >= flen) ? 1 + (int)a.div_hop.div((uint32_t)(avail - flen)) : 0;
        v_first = v_nnew > slots ? v_nnew - slots : 0;       // older frames would be overwritten anyway
    };
    // advance to the next DUE frame of the current batch of (up to 64) streams; false when the batch is exhausted
    // (no memory access in here: the frame loop must not wait on the vector-memory counter between a PCM request and
    //  the conversion of those samples)
    auto next_frame = [&](FrameTask<R>& f) -> bool {
        while (due == 0) {
            ++kb_next;
            if (kb_next >= n_kb) return false;
            const long long sg = 2LL * (base + lane) + (kb_next & 1);
            due = __ballot(kb_next >= v_first && kb_next < v_nnew && sg >= sg_begin && sg < sg_end);
        }
        const int i = __builtin_ctzll(due);
        due &= due - 1;
        const int kb = kb_next, s = base + i;                 // s: position in this launch (PCM row)
        const int q = __builtin_amdgcn_readlane(vq, i);
        const uint32_t kc = (uint32_t)__builtin_amdgcn_readlane(vkc, i);
        const int ss = __builtin_amdgcn_readlane(vss, i);
        const int sid = ss >> 1;                              // the stream: leftover PCM, ring rows
        const int tile = sid >> 4, j = sid & 15;
        // the q samples in front of the chunk: the stream's current carry side -- or the tail of its row of the previous
        // call's chunks where those were kept (car[v], v < q, never dereferenced when q <= 0)
        f.car = a.head ? a.head + (size_t)s * a.head_chunk + (a.head_chunk - q)
                       : a.st.carry + ((size_t)(ss & 1) * a.st.n_padded + (size_t)sid) * kCarryCap;
        f.vb = kb * hop; f.q = q;
        const int w0 = f.vb - q;
        int u0 = 0;
        f.off0 = w0;
        if (w0 >= 0 && U > 1) { u0 = (int)a.div_chunk.div((uint32_t)w0); f.off0 = w0 - u0 * C; }
        f.row = a.pcm + (size_t)s * C + (size_t)u0 * update_stride;
        const int slot = (int)((kc + (uint32_t)kb) & (uint32_t)(slots - 1));
        const size_t cell = ((size_t)tile * slots + slot) * kTileStreams + j;
        f.ring_row = a.ring_bf16 ? reinterpret_cast<float*>(reinterpret_cast<__bf16*>(a.ring) + cell * kRowFloats)
                                 : a.ring + cell * kRowFloats;
        // projection rows: [tile][slot] blocks of 4 KB laid out [output tile][stream][g][q] (gru_device.h: proj_base)
        f.proj_row = (!NOPROJ && a.proj_ring) ? a.proj_ring + ((size_t)tile * slots + slot) * kTileStreams * kProjRow + (size_t)j * 16 : nullptr;
        return true;
    };
    // The frame's samples as int16 pairs: point n = lane + 64 a4 <-> samples 2n, 2n+1 (zero beyond the frame length).
    // A frame is cut from up to three places (sample m of the frame, 0 <= m < flen):
    //     [0, qa)      the carry, from its sample vb on          qa = q - vb  (> 0 only for the first frame of a call)
    //     [qa, nb)     the chunk row, from off0 + m on           nb = min(flen, C - off0)
    //     [nb, flen)   the same stream's row of the NEXT update  (several updates per call only; never together with
    //                  a carry part: that would take a chunk shorter than a frame, which a multi-update call sends
    //                  down the sample-by-sample path)
    // Each part is ONE bounds-checked buffer load per a4 with the part's start folded into the lane offset: lanes
    // before the part wrap to a huge unsigned offset, lanes behind it exceed num_records, and both read 0 -- no
    // per-lane pointer selection, no masks; the parts are OR-ed when the samples are converted.  Nothing here
    // consumes a loaded value, so the loads stay in flight while the current frame is transformed.
    // (the wave-uniform part of an offset is made opaque: folded into the instruction's immediate offset it would be
    //  added AFTER the unsigned wrap the scheme relies on)
    auto part_offset = [](int shift, int a4) -> int {
        int sh = shift + 256 * a4;
        asm volatile("" : "+s"(sh));
        return sh;
    };
    auto request_pcm = [&](const FrameTask<R>& f) -> PcmRegs {
        PcmRegs r;                  // (b0..b3 stay unset unless `two`: touching them here would wait for the loads)
        r.two = false;
        if (pairs && (f.q & 1) == 0) {
            const int qa = f.q - f.vb;
            const int over = SINGLE ? 0 : f.off0 + flen - C;      // (a single update's frames end inside its chunk)
            const int in_row = f.off0 + flen < C ? f.off0 + flen : C;
            const __amdgpu_buffer_rsrc_t rb = wave_rsrc(f.row, 2 * (in_row > 0 ? in_row : 0));
            const int shift = 2 * f.off0;
            r.a0 = __builtin_amdgcn_raw_buffer_load_b32(rb, 4 * lane + part_offset(shift, 0), 0, 0);
            r.a1 = __builtin_amdgcn_raw_buffer_load_b32(rb, 4 * lane + part_offset(shift, 1), 0, 0);
            r.a2 = __builtin_amdgcn_raw_buffer_load_b32(rb, 4 * lane + part_offset(shift, 2), 0, 0);
            r.a3 = __builtin_amdgcn_raw_buffer_load_b32(rb, 4 * lane + part_offset(shift, 3), 0, 0);
            if (qa > 0 || over > 0) {
                const bool head = qa > 0;
                const __amdgpu_buffer_rsrc_t r2 = head ? wave_rsrc(f.car + f.vb, 2 * (qa < flen ? qa : flen))
                                                       : wave_rsrc(f.row + update_stride, 2 * over);
                const int shift2 = head ? 0 : 2 * (f.off0 - C);
                r.b0 = __builtin_amdgcn_raw_buffer_load_b32(r2, 4 * lane + part_offset(shift2, 0), 0, 0);
                r.b1 = __builtin_amdgcn_raw_buffer_load_b32(r2, 4 * lane + part_offset(shift2, 1), 0, 0);
                r.b2 = __builtin_amdgcn_raw_buffer_load_b32(r2, 4 * lane + part_offset(shift2, 2), 0, 0);
                r.b3 = __builtin_amdgcn_raw_buffer_load_b32(r2, 4 * lane + part_offset(shift2, 3), 0, 0);
                r.two = true;
            }
            return r;
        }
        const RawFrame sl = fetch_frame_slow(f.car, f.row, f.vb, f.q, f.off0, lane, flen, C, update_stride);
        r.a0 = sl.v[0]; r.a1 = sl.v[1]; r.a2 = sl.v[2]; r.a3 = sl.v[3];
        return r;
    };

    // int16 pairs -> the complex points of the packed real FFT (the first consumer of a requested frame)
    auto convert = [&](PcmRegs& p, pe_wave::Regs<R>& v) {
        if (p.two) { p.a0 |= p.b0; p.a1 |= p.b1; p.a2 |= p.b2; p.a3 |= p.b3; }
        v.re[0] = (R)(int)(short)(p.a0 & 0xffff); v.im[0] = (R)(p.a0 >> 16);
        v.re[1] = (R)(int)(short)(p.a1 & 0xffff); v.im[1] = (R)(p.a1 >> 16);
        v.re[2] = (R)(int)(short)(p.a2 & 0xffff); v.im[2] = (R)(p.a2 >> 16);
        v.re[3] = (R)(int)(short)(p.a3 & 0xffff); v.im[3] = (R)(p.a3 >> 16);
    };

    // ---- kernel top: everything whose address is known without a dependent load goes out first ------------------
    // the table image for LDS goes out first: loads return in order, and these (L2 hits after a compute unit's first
    // workgroup) must not queue behind the stream counters and samples, which come from farther away
    const TabRegs tab_regs = wave_tables_issue<R>(wt);
    load_counters();
    // this lane's twiddles, straight from the global image (once per wave)
#if PE_TW_LDS == 2
    const pe_wave::LaneConsts<R> lc{};
#else
    const pe_wave::LaneConsts<R> lc = pe_wave::lane_consts(pe_wave::bind<R>(static_cast<const unsigned char*>(wt.blob), wt.L), lane);
#endif
    const pe_wave::Tab<R> tab = wave_bind<R, SH>(smem + kWaveImageBase<R>, wt.L);
    R* const S = reinterpret_cast<R*>(smem) + (size_t)wave * kWaveScratchReals;
    // tables -> LDS first, then the counters (both on their way since the top; measured 16.40 vs 16.48 us per fused update
    // against requesting the first frame before the image is stored: the workgroup barrier in the commit is passed
    // earlier by every wave), then the first frame's samples
    wave_tables_commit<R>(smem, wt, tab_regs);
    lane_frames();
    FrameTask<R> cur;
    bool have = next_frame(cur);
    PcmRegs pcm;
    if (have) pcm = request_pcm(cur);
    const LaneRuns lr = lane_runs(tab, lane, geo.n_filt);
    wave_scratch_init(S, lane);
    for (;;) {                                          // batches of up to 64 streams (one, unless a wave owns more)
        if (have) {
        // the row of a frame is stored one frame late, right after the wait for the next frame's samples: at that wait
        // only loads are outstanding, and nothing ever waits for a store
        float xf_prev = 0.0f;
        float* row_prev = nullptr;
        auto store_row = [&]() {
            if (lane < kRowFloats) {
                if (a.ring_bf16) reinterpret_cast<__bf16*>(row_prev)[lane] = (__bf16)xf_prev;    // round to nearest even where the row is stored
                else row_prev[lane] = xf_prev;
            }
        };
        for (;;) {
            pe_wave::Regs<R> v;
            convert(pcm, v);                                            // the wait for this frame's samples
            asm volatile("" : "+v"(v.re[0]), "+v"(v.re[3]) : : "memory");  // (the store below must stay below that wait)
            __builtin_amdgcn_sched_barrier(0);
            if (row_prev) store_row();
            FrameTask<R> nxt;
            const bool have_next = next_frame(nxt);
            if (have_next) pcm = request_pcm(nxt);                      // lands while the current frame is transformed
            const R coeff = mfcc_wave_frame<R, SH>(tab, lc, lr, S, lane, geo.n_filt, geo.n_mfcc, v, K::PSCALE_I16, geo.log_mode);
            // coefficient c sits in lanes 4c .. 4c+3: lane c fetches it, and the first 16 lanes store the row as ONE
            // contiguous 64-byte (bf16: 32-byte) write -- 13 coefficients + zero padding, as the clear kernel left it
            const float mine = (lane >> 2) < geo.n_mfcc ? (float)coeff : 0.0f;
            const float xf = __shfl(mine, (lane & 15) * 4, 64);
            xf_prev = xf; row_prev = cur.ring_row;
            if (!NOPROJ && cur.proj_row) {
                // input projection of this frame, once, for every window it will appear in: row[o] = b[o] + sum_c x[c] W[c][o]
                // (o in MFMA slot order); the rounded float32 features are what the network would have read
                float* XF = reinterpret_cast<float*>(S);
                if (lane < kRowFloats) XF[lane] = xf;
                group_sync();
                float acc = tab.proj_b[lane];
                for (int cc = 0; cc < tab.proj_rows; ++cc) acc = fmaf(XF[cc], tab.proj_w[cc * kProjRow + lane], acc);
                cur.proj_row[(size_t)((lane >> 2) & 3) * (kTileStreams * 16) + (lane >> 4) * 4 + (lane & 3)] = acc;   // o = 16 g + 4 tl + q
                group_sync();
            }
            if (!have_next) break;
            cur = nxt;
        }
        store_row();
        }
        base += 64;
        if (base >= s_end) break;
        load_counters();
        lane_frames();
        kb_next = -1; due = 0;
        have = next_frame(cur);
        if (have) pcm = request_pcm(cur);
    }
}

// ---- stateless whole-buffer form (vectorize_raw): one frame per wave, float64 samples in -------------------------
template <class R, class SH>
__device__ __forceinline__ void mfcc_offline_frames(const MfccOfflineArgs<R>& a, const WaveTables<R>& wt, unsigned char* smem) {
    const StreamGeom& geo = a.geo;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const TabRegs tab_regs = wave_tables_issue<R>(wt);
    wave_tables_commit<R>(smem, wt, tab_regs);
    const pe_wave::Tab<R> tab = wave_bind<R, SH>(smem + kWaveImageBase<R>, wt.L);
    R* S = reinterpret_cast<R*>(smem) + (size_t)wave * kWaveScratchReals;
#if PE_TW_LDS == 2
    const pe_wave::LaneConsts<R> lc{};
#else
    const pe_wave::LaneConsts<R> lc = pe_wave::lane_consts(pe_wave::bind<R>(static_cast<const unsigned char*>(wt.blob), wt.L), lane);
#endif
    const LaneRuns lr = lane_runs(tab, lane, geo.n_filt);
    wave_scratch_init(S, lane);
    const int flen = geo.frame_len;
    for (long long fr = (long long)blockIdx.x * waves + wave; fr < a.n_frames; fr += (long long)gridDim.x * waves) {
        const double* x = a.audio + fr * geo.hop;
        pe_wave::Regs<R> v;
#pragma unroll
        for (int a4 = 0; a4 < 4; ++a4) {
            const int n = 2 * (lane + 64 * a4);
            v.re[a4] = n < flen ? (R)x[n] : R(0);
            v.im[a4] = n + 1 < flen ? (R)x[n + 1] : R(0);
        }
        const R coeff = mfcc_wave_frame<R, SH>(tab, lc, lr, S, lane, geo.n_filt, geo.n_mfcc, v, RealK<R>::INV_FFT, geo.log_mode);
        const int c = lane >> 2;
        if ((lane & 3) == 0) {
            if (a.out && c < geo.n_mfcc) a.out[fr * geo.n_mfcc + c] = (double)coeff;
            if (a.out_rows) a.out_rows[fr * kRowFloats + c] = c < geo.n_mfcc ? (float)coeff : 0.0f;
        }
        if (a.out_mels)             // the log-mel energies are still in this wave's scratch
            for (int f = lane; f < geo.n_filt; f += 64) a.out_mels[fr * geo.n_filt + f] = (double)S[pe_wave::kLogMelOff + f];
        group_sync();
    }
}

}  // namespace pe
