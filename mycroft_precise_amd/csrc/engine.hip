// C ABI of libprecise_engine.so (declared in include/precise_engine.h): host-side state,
// constant tables, weight packing and kernel sequencing for the MI355X wake-word hot path.
// The arithmetic lives in kernels.hip (mfcc_wave_device.h, gru_*_device.h); nothing here computes on the CPU
// beyond building constant tables once per engine.
#include "../../include/precise_engine.h"
#include "pe_common.h"
#include "gru_cw_pack.h"
#include "mfcc_general_device.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace pe;

namespace {

thread_local std::string g_global_error;

struct DeviceBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

}  // namespace

struct pe_engine {
    pe_params prm{};
    int device = 0;
    int n_streams = 0, n_tiles = 0, n_padded = 0;
    int units = 0, n_in = 0, n_layers = 0;
    int ring_slots = 0;
    int mel_nnz = 0;
    float dense_bias = 0.f;
    std::string err;
    std::vector<void*> allocs;
    int64_t device_bytes = 0;
    // streaming state: per stream two sides of (16-byte record, leftover PCM); a call that advances a stream reads its
    // current side and writes the other (pe_common.h: StreamRec) -- streams that take no part in a call are not touched
    StreamRec* rec = nullptr;                // [n_padded][2 sides]
    int16_t* carry = nullptr;                // [2][n_padded][carry_cap]
    // leftovers kept in place (pe_update_device_keep / pe_update_async): non-null = the chunks of the last call,
    // [n_streams][kept_chunk], still hold every stream's leftover and the carry was NOT written; any call of another style
    // first copies them into the carry (flush_kept).  call_head / call_keep: what the launches of the call being built take.
    const int16_t* kept = nullptr; int kept_chunk = 0;
    const int16_t* call_head = nullptr; int call_head_chunk = 0; bool call_keep = false;
    uint32_t call_no = 1;                    // number of the last call that wrote records (readers of later launches pass call_no + 1)
    uint32_t renumber_at = 0x7fff0000u;      // call number at which every record is renumbered and the count restarts (pe_set_renumber_at: tests)
    bool fused = true;      // MFCC || GRU in one launch when the chunk size allows it
    int gru_waves = 0;      // 0 = auto (4 waves per tile while tiles <= CUs, else 1), or forced 1 / 4
    // general front end (mfcc_general_device.h): any n_fft / n_filt / n_mfcc the stock-shape wave kernel has no tables for
    bool general = false;
    int row_floats = kRowFloats;   // floats per feature row: 32 when a frame has 17..32 coefficients
    int carry_cap = kCarryCap;     // int16 samples of leftover PCM kept per stream (>= frame length)
    GeneralTables gtab{};
    int gru_tiling = -1;    // -1 = auto (stock width re-tiled while tiles <= 2 CUs; XDL form above 4 tiles per CU), 0 = classic, 1 = re-tiled (gru_cw_device.h), 2 = XDL form (gru_x3_device.h)
    float* cw_blob = nullptr;
    int n_cus = 256;        // compute units of the device (MI355X: 256)
    float* ring = nullptr;
    // input projections x.W + b of every frame beside its feature row (stock-width float32 network, <= kProjMaxTiles
    // tiles): written once by the MFCC stage, read by the network instead of 16 of its 41 MFMAs per timestep
    float* proj_ring = nullptr;
    bool proj_ok = false, proj_on = false;
    std::vector<float> proj_w_host, proj_b_host;
    // several updates per call (pe_reserve_updates / pe_update_many*)
    int max_updates = 1;
    uint32_t* ke_hist = nullptr;
    // constant tables of the MFCC frame kernel, laid out as they sit in LDS (mfcc_wave_tables.h)
    unsigned char* table_blob = nullptr;
    pe_wave::Layout table_layout{};
    // packed network
    float* wxd = nullptr;
    float* wx = nullptr; float* wr1 = nullptr; float* wr2 = nullptr; float* bias = nullptr; float* wd = nullptr;
    // wide / stacked network (units 64..256, 1-2 layers): weight streams in MFMA A-operand order
    bool wide = false;
    float* wide_buf[2][6] = {{nullptr}};     // per layer: wx1, wr1, wx2, wr2, b1, b2
    int wide_kx4[2] = {1, 1};
    float* wide_wd = nullptr;
    // the same network packed for gru_wide_x3_device.h (row i of tile tau = unit 16 tau + i; k-slot 8 gk + e = source unit 16 kappa + 4 gk + e)
    float* wide3_buf[2][6] = {{nullptr}};
    float* wide3_wd = nullptr;
    // bf16-operand network (pe_params.gru_precision = 1)
    uint16_t* wx_bf16 = nullptr; uint16_t* wr_bf16 = nullptr; float* wd_bf16 = nullptr;
    uint32_t* b20_blob = nullptr;     // the same network in the layout of gru_b20_device.h (<= 20 units, <= 14 features)
    // float32 network as three bf16 pieces per operand on the XDL pipe (gru_x3_device.h; tiling 2): packed for every
    // float32 network of <= 20 units and <= 15 inputs without delta features
    uint32_t* x3_blob = nullptr;
    // on-device ThresholdDecoder / TriggerDetector (pe_set_decoder / pe_set_trigger)
    double* cd = nullptr; int cd_len = 0, dec_min_out = 0, dec_out_range = 0; double dec_center = 0.5;
    int32_t* activation = nullptr; double trig_threshold = 0.5; int trig_level = 3, trig_rearm = -8; bool trig_on = false;
    // staging for the host entry points (grown on demand)
    DeviceBuf st_pcm, st_out, st_feats, st_mask, st_audio, st_mfcc, st_conf, st_fired, st_ids;
    std::vector<uint8_t> seen_ids;          // pe_update_subset: duplicate check of the host entry point
    // timing
    bool timing = false;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    bool ev_valid = false, ev_has_gru = false;
    // host-fed pipeline (pe_update_async / pe_wait): a ring of kAsyncDepth updates in flight, each with its own device
    // buffers and pinned staging; the chunk of update u + 1 crosses PCIe (copy stream) while update u runs (compute stream)
    static constexpr int kAsyncDepth = 3;
    struct AsyncSlot {
        float* pin_out = nullptr; size_t pin_out_bytes = 0;     // pinned landing zone for callers whose output array is pageable
        DeviceBuf dev_in, dev_out;
        hipEvent_t copied = nullptr, done = nullptr;
        float* user_out = nullptr; size_t out_bytes = 0;
        bool direct_out = false, busy = false;
    } aslot[kAsyncDepth];
    hipStream_t s_copy = nullptr, s_compute = nullptr;
    // the stream of the last *_device call that moved the streams' state: pe_update_async runs on the engine's own
    // (non-blocking) streams, which nothing orders behind that work -- not even the NULL stream -- so it waits for it
    // by event, lazily (recorded only when the caller switches styles: the *_device loop itself pays nothing)
    hipStream_t last_user_stream = nullptr;
    bool user_dirty = false;
    hipEvent_t ev_user = nullptr;
    unsigned async_next = 0;
    int async_inflight = 0;
    std::vector<std::pair<char*, size_t>> pinned;               // pe_host_alloc'ed ranges (zero-copy sources / destinations)
};

namespace {

int fail(pe_engine* e, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (e) e->err = buf; else g_global_error = buf;
    return code;
}

#define PE_HIP(e, call)                                                                      \
    do {                                                                                     \
        hipError_t _err = (call);                                                            \
        if (_err != hipSuccess)                                                              \
            return fail((e), PE_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_err), \
                        __FILE__, __LINE__);                                                 \
    } while (0)

int drain_async(pe_engine* e);
// every entry point that reads or moves the streams' state first lets the updates of pe_update_async finish (they run on the
// engine's own streams, which no other entry point is ordered against)
#define PE_DRAIN(e) do { int _drc = drain_async(e); if (_drc) return _drc; } while (0)

template <class T>
int dev_alloc(pe_engine* e, T** out, size_t count) {
    void* p = nullptr;
    const size_t bytes = (count ? count : 1) * sizeof(T);
    hipError_t err = hipMalloc(&p, bytes);
    if (err != hipSuccess) return fail(e, PE_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(err));
    e->allocs.push_back(p);
    e->device_bytes += (int64_t)bytes;
    *out = static_cast<T*>(p);
    return PE_OK;
}

template <class T>
int dev_upload(pe_engine* e, T** out, const std::vector<T>& host) {
    int rc = dev_alloc(e, out, host.size());
    if (rc) return rc;
    if (!host.empty()) PE_HIP(e, hipMemcpy(*out, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    return PE_OK;
}

// release one dev_alloc'ed buffer before pe_destroy (tables that are replaced: decoder LUT, ring, ke_hist)
void dev_free(pe_engine* e, void* p, size_t bytes) {
    if (!p) return;
    for (auto it = e->allocs.begin(); it != e->allocs.end(); ++it)
        if (*it == p) { e->allocs.erase(it); break; }
    (void)hipFree(p);
    e->device_bytes -= (int64_t)bytes;
}

int ensure(pe_engine* e, DeviceBuf& b, size_t bytes) {
    if (b.bytes >= bytes) return PE_OK;
    if (b.p) { (void)hipFree(b.p); e->device_bytes -= (int64_t)b.bytes; b.p = nullptr; b.bytes = 0; }
    hipError_t err = hipMalloc(&b.p, bytes);
    if (err != hipSuccess) return fail(e, PE_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(err));
    b.bytes = bytes;
    e->device_bytes += (int64_t)bytes;
    return PE_OK;
}

int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

// floats of HBM behind the feature ring: 16 floats per row, or 16 bf16 (half of it) with ring_precision = 1
size_t ring_floats(const pe_engine* e) {
    const size_t rows = (size_t)e->n_tiles * e->ring_slots * kTileStreams;
    return e->prm.ring_precision == 1 ? rows * kRowFloats / 2 : rows * (size_t)e->row_floats;
}



template <class R>
int build_tables(pe_engine* e, const double* mel_filters) {
    std::vector<unsigned char> blob;
    const std::string err = pe_wave::build<R>(mel_filters, e->prm.n_filt, e->prm.n_mfcc, blob, e->table_layout,
                                              e->proj_ok ? e->proj_w_host.data() : nullptr, e->proj_ok ? e->proj_b_host.data() : nullptr);
    if (!err.empty()) return fail(e, PE_ERR_UNSUPPORTED, "%s", err.c_str());
    return dev_upload(e, &e->table_blob, blob);
}

// Tables of the general front end: twiddles of the packed real FFT, the filterbank as CSR, the DCT-II (ortho) matrix.
template <class R>
int build_general_tables(pe_engine* e, const double* mel_filters) {
    const int N = e->prm.n_fft, bins = N / 2 + 1, nf = e->prm.n_filt, nc = e->prm.n_mfcc;
    const long double PI = 3.14159265358979323846264338327950288L;
    const bool pow2 = general_is_pow2(N);
    // power of two: packed real transform of M = N / 2 points;  otherwise Bluestein over L = 2^log2m >= 2 N - 1 points
    const int log2m_blue = general_blue_log2(N);
    const int M = pow2 ? N / 2 : (1 << log2m_blue);           // length of the complex radix-2 transform in LDS
    std::vector<R> tw((size_t)M), wn((size_t)(pow2 ? (M / 2 + 1) * 2 : 2));       // tw: M / 2 complex = M reals
    for (int k = 0; k < M / 2; ++k) {
        tw[2 * k] = (R)cosl(-2.0L * PI * k / M);
        tw[2 * k + 1] = (R)sinl(-2.0L * PI * k / M);
    }
    if (pow2)
        for (int k = 0; k <= M / 2; ++k) {
            wn[2 * k] = (R)cosl(-2.0L * PI * k / N);
            wn[2 * k + 1] = (R)sinl(-2.0L * PI * k / N);
        }
    std::vector<R> chirp, bhat;
    if (!pow2) {
        // w[n] = exp(-i pi n^2 / N): the angle from n^2 mod 2 N (exact integers), so large n lose nothing
        auto wl = [&](long long n, long double& re, long double& im) {
            const long long r = (n * n) % (2LL * N);
            const long double a = -PI * (long double)r / (long double)N;
            re = cosl(a); im = sinl(a);
        };
        chirp.resize((size_t)2 * N);
        for (int n = 0; n < N; ++n) { long double re, im; wl(n, re, im); chirp[2 * n] = (R)re; chirp[2 * n + 1] = (R)im; }
        // b[m mod L] = conj(w[m]) for |m| < N, zero elsewhere; B = DFT_L(b) (direct, long double, roots from an exact table)
        const int L = M;
        std::vector<long double> br((size_t)L, 0.0L), bi((size_t)L, 0.0L), cr((size_t)L), ci((size_t)L);
        for (int m = -(N - 1); m <= N - 1; ++m) { long double re, im; wl(m, re, im); br[(m + L) % L] = re; bi[(m + L) % L] = -im; }
        for (int j = 0; j < L; ++j) { cr[j] = cosl(-2.0L * PI * j / L); ci[j] = sinl(-2.0L * PI * j / L); }
        bhat.resize((size_t)2 * L);
        int bits = 0;
        while ((1 << bits) < L) ++bits;
        for (int k = 0; k < L; ++k) {
            long double sr = 0.0L, si = 0.0L;
            for (int m = 0; m < L; ++m) {
                if (br[m] == 0.0L && bi[m] == 0.0L) continue;
                const int j = (int)(((long long)k * m) % L);
                sr += br[m] * cr[j] - bi[m] * ci[j];
                si += br[m] * ci[j] + bi[m] * cr[j];
            }
            unsigned rev = 0;
            for (int b = 0; b < bits; ++b) rev |= ((unsigned)(k >> b) & 1u) << (bits - 1 - b);
            bhat[2 * (size_t)rev] = (R)(sr / L);               // bit-reversed position, 1 / L of the inverse transform folded in
            bhat[2 * (size_t)rev + 1] = (R)(si / L);
        }
    }
    // filterbank as lane runs (mfcc_general_device.h: GeneralTables): every filter's non-zeros, in bin order, in runs of
    // <= kGeneralRun entries; run r <-> lane r % 64 of round r / 64
    std::vector<int> run_ptr(nf + 1, 0);
    std::vector<std::pair<int, R>> entries;
    std::vector<int> run_first;            // first entry of every run
    std::vector<int> run_len;
    for (int f = 0; f < nf; ++f) {
        int in_run = 0;
        for (int k = 0; k < bins; ++k) {
            const double v = mel_filters[(size_t)f * bins + k];
            if (v == 0.0) continue;
            if (in_run == 0) { run_first.push_back((int)entries.size()); run_len.push_back(0); }
            entries.emplace_back(k, (R)v);
            ++run_len.back();
            if (++in_run == kGeneralRun) in_run = 0;
        }
        run_ptr[f + 1] = (int)run_first.size();
    }
    const int n_runs = (int)run_first.size();
    const int n_rounds = n_runs > 0 ? (n_runs + 63) / 64 : 1;
    std::vector<R> run_w((size_t)n_rounds * kGeneralRun * 64, R(0));
    std::vector<int> run_bin((size_t)n_rounds * kGeneralRun * 64, 0);
    for (int r = 0; r < n_runs; ++r)
        for (int i = 0; i < run_len[r]; ++i) {
            const size_t at = ((size_t)(r / 64) * kGeneralRun + i) * 64 + (r % 64);
            run_bin[at] = entries[run_first[r] + i].first;
            run_w[at] = entries[run_first[r] + i].second;
        }
    std::vector<R> dct_t((size_t)nf * kGeneralDctCols, R(0));
    for (int c = 0; c < nc; ++c)           // scipy.fftpack.dct(type 2, norm='ortho'): y[c] = 2 f(c) sum_n x[n] cos(pi c (2 n + 1) / (2 N))
        for (int f = 0; f < nf; ++f) {
            const long double scale = c == 0 ? sqrtl(1.0L / (4.0L * nf)) : sqrtl(1.0L / (2.0L * nf));
            dct_t[(size_t)f * kGeneralDctCols + c] = (R)(2.0L * scale * cosl(PI * c * (2 * f + 1) / (2.0L * nf)));
        }
    R* d_tw = nullptr; R* d_wn = nullptr; R* d_w = nullptr; R* d_dct = nullptr; int* d_ptr = nullptr; int* d_bin = nullptr;
    int rc;
    if ((rc = dev_upload(e, &d_tw, tw))) return rc;
    if ((rc = dev_upload(e, &d_wn, wn))) return rc;
    if ((rc = dev_upload(e, &d_w, run_w))) return rc;
    if ((rc = dev_upload(e, &d_dct, dct_t))) return rc;
    if ((rc = dev_upload(e, &d_ptr, run_ptr))) return rc;
    if ((rc = dev_upload(e, &d_bin, run_bin))) return rc;
    int log2m = 0;
    while ((1 << log2m) < M) ++log2m;
    R* d_chirp = nullptr; R* d_bhat = nullptr;
    if (!pow2) {
        if ((rc = dev_upload(e, &d_chirp, chirp))) return rc;
        if ((rc = dev_upload(e, &d_bhat, bhat))) return rc;
    }
    e->gtab = GeneralTables{d_tw, d_wn, d_w, d_bin, d_ptr, d_dct, N, log2m, nf, nc, e->prm.vectorizer == 3 ? 1 : 0, n_rounds, d_chirp, d_bhat};
    return PE_OK;
}

// Arrange the Keras matrices as MFMA A-operands (see the layout comment in gru_kernels.hip).
int pack_gru_weights(pe_engine* e, const pe_gru_layer& L, const float* dense_kernel) {
    const int H = L.units, F = e->n_in;          // F = base features; with use_delta the kernel has 2F rows
    const bool delta = e->prm.use_delta != 0;
    const int R = gru_small_regs(H), NT = gru_small_tiles(H);
    const int KX = e->row_floats / kRowFloats;      // 16-feature groups of a feature row (2: 17..32 coefficients per frame)
    std::vector<float> wx((size_t)KX * NT * 4 * 64, 0.f), wxd((size_t)NT * 4 * 64, 0.f), wr1((size_t)NT * R * 64, 0.f), wr2((size_t)NT * R * 64, 0.f);
    std::vector<float> bias((size_t)NT * 4 * 64, 0.f), wd((size_t)R * 64, 0.f);
    for (int tile = 0; tile < NT; ++tile)
        for (int lane = 0; lane < 64; ++lane) {
            const int i = lane & 15, g = lane >> 4;
            {   // A operand: row i of the tile, k-slot g
                const int reg = i & 3, gout = i >> 2;
                const int slot = 4 * tile + reg;
                const int gate = slot / R, rho = slot % R, u = 4 * rho + gout;
                if (slot < 3 * R && u < H) {
                    const int col = gate * H + u;
                    for (int kx = 0; kx < KX; ++kx)
                        for (int kk = 0; kk < 4; ++kk) {
                            const int phi = 16 * kx + 4 * g + kk;
                            if (phi < F) wx[(((size_t)kx * NT + tile) * 4 + kk) * 64 + lane] = L.kernel[(size_t)phi * 3 * H + col];
                            if (kx == 0 && phi < F && delta) wxd[((size_t)tile * 4 + kk) * 64 + lane] = L.kernel[(size_t)(F + phi) * 3 * H + col];
                        }
                    for (int rs = 0; rs < R; ++rs) {
                        const int usrc = 4 * rs + g;
                        if (usrc >= H) continue;
                        const float w = L.recurrent_kernel[(size_t)usrc * 3 * H + col];
                        (gate < 2 ? wr1 : wr2)[((size_t)tile * R + rs) * 64 + lane] = w;
                    }
                }
            }
            for (int q = 0; q < 4; ++q) {   // C operand: this lane's output rows 4g + q
                const int slot = 4 * tile + q;
                const int gate = slot / R, rho = slot % R, u = 4 * rho + g;
                if (slot < 3 * R && u < H) bias[((size_t)tile * 4 + q) * 64 + lane] = L.bias[gate * H + u];
            }
        }
    for (int rho = 0; rho < R; ++rho)
        for (int lane = 0; lane < 64; ++lane) {
            const int u = 4 * rho + (lane >> 4);
            if (u < H) wd[(size_t)rho * 64 + lane] = dense_kernel[u];
        }
    // input-projection rows (mfcc_wave_device.h epilogue): element o = 16 g + 4 tile + q of a row is accumulator q of
    // output tile `tile` in lane group g, i.e. slot 4 tile + q, unit 4 rho + g
    if (NT <= 4 && !delta && KX == 1) {
        e->proj_w_host.assign((size_t)F * kProjRow, 0.f);
        e->proj_b_host.assign(kProjRow, 0.f);
        for (int o = 0; o < kProjRow; ++o) {
            const int g = o >> 4, tile = (o >> 2) & 3, q = o & 3;
            const int slot = 4 * tile + q;
            const int gate = slot / R, rho = slot % R, u = 4 * rho + g;
            if (tile >= NT || slot >= 3 * R || u >= H) continue;
            const int col = gate * H + u;
            e->proj_b_host[o] = L.bias[col];
            for (int c = 0; c < F; ++c) e->proj_w_host[(size_t)c * kProjRow + o] = L.kernel[(size_t)c * 3 * H + col];
        }
    }
    int rc;
    if ((rc = dev_upload(e, &e->wx, wx))) return rc;
    if ((rc = dev_upload(e, &e->wxd, wxd))) return rc;
    if ((rc = dev_upload(e, &e->wr1, wr1))) return rc;
    if ((rc = dev_upload(e, &e->wr2, wr2))) return rc;
    if ((rc = dev_upload(e, &e->bias, bias))) return rc;
    if ((rc = dev_upload(e, &e->wd, wd))) return rc;
    if (R == 5 && KX == 1) {                    // the stock width also in its re-tiled form (gru_cw_device.h)
        const std::vector<float> blob = pack_gru_cw(L.kernel, L.recurrent_kernel, L.bias, F, H, delta);
        if ((rc = dev_upload(e, &e->cw_blob, blob))) return rc;
    }
    return PE_OK;
}

// bf16 network operands (gru_bf16_device.h): unit u = 8 g + i; tile (gate, t): row 4 gout + q <-> unit
// 8 gout + 4 t + q; A operand of lane (row i, k-group g) = 8 consecutive k (features / source units).
uint16_t to_bf16(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);     // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);                                                 // round to nearest even
    return (uint16_t)(u >> 16);
}

int pack_gru_weights_bf16(pe_engine* e, const pe_gru_layer& L, const float* dense_kernel) {
    // with use_delta the layer has 2 F inputs: features in k = 0..15, their first differences in k = 16..31
    const bool delta = e->prm.use_delta != 0;
    const int H = L.units, F = delta ? L.n_in / 2 : L.n_in;
    if (delta && F > 14) return fail(e, PE_ERR_UNSUPPORTED, "bf16 network with use_delta takes n_mfcc <= 14 (k slots 30, 31 of the input contraction carry the biases)");
    std::vector<uint16_t> wx((size_t)6 * 64 * 8, 0), wr((size_t)6 * 64 * 8, 0);
    std::vector<float> wd((size_t)8 * 64, 0.f);
    for (int tl = 0; tl < 6; ++tl) {
        const int gate = tl >> 1, t = tl & 1;
        for (int lane = 0; lane < 64; ++lane) {
            const int i = lane & 15, g = lane >> 4;
            const int gout = i >> 2, reg = i & 3;
            const int u = 8 * gout + 4 * t + reg;                    // A row i of this tile
            if (u < H) {
                const int col = gate * H + u;
                for (int ek = 0; ek < 8; ++ek) {
                    const int k = 8 * g + ek;
                    if (k < F) wx[((size_t)tl * 64 + lane) * 8 + ek] = to_bf16(L.kernel[(size_t)k * 3 * H + col]);
                    if (delta && k >= 16 && k - 16 < F) wx[((size_t)tl * 64 + lane) * 8 + ek] = to_bf16(L.kernel[(size_t)(F + k - 16) * 3 * H + col]);
                    if (k < H) wr[((size_t)tl * 64 + lane) * 8 + ek] = to_bf16(L.recurrent_kernel[(size_t)k * 3 * H + col]);
                }
                if (g == 3) {                                        // k = 30, 31: the bias as hi + lo against x = 1.0
                    const float b = L.bias[col];
                    const uint16_t hi = to_bf16(b);
                    uint32_t hb = (uint32_t)hi << 16;
                    float hif;
                    std::memcpy(&hif, &hb, 4);
                    wx[((size_t)tl * 64 + lane) * 8 + 6] = hi;
                    wx[((size_t)tl * 64 + lane) * 8 + 7] = to_bf16(b - hif);
                }
            }
        }
    }
    for (int i = 0; i < 8; ++i)
        for (int lane = 0; lane < 64; ++lane) {
            const int u = 8 * (lane >> 4) + i;
            if (u < H) wd[(size_t)i * 64 + lane] = dense_kernel[u];
        }
    int rc;
    if ((rc = dev_upload(e, &e->wx_bf16, wx))) return rc;
    if ((rc = dev_upload(e, &e->wr_bf16, wr))) return rc;
    if ((rc = dev_upload(e, &e->wd_bf16, wd))) return rc;
    return PE_OK;
}

// Float32 network on the XDL pipe (gru_x3_device.h): every weight as three bf16 pieces (round to nearest even of the
// running remainder; hi + mid + lo == the float32 weight exactly), arranged as the A operands of
// v_mfma_f32_16x16x32_bf16: lane (row i = lane & 15, k-group gk = lane >> 4) holds k = 8 gk .. 8 gk + 7.
// Output tiles 0..2 = z / r / candidate of units 0..15 (row i <-> unit i), tile 3 row 4 g + q = gate q of unit 16 + g.
// k-group gk carries the source units 4 gk .. 4 gk + 3 (operands 0..2) and 16 + gk (operand 3) -- the units whose
// values lane group gk owns -- and, on the input side, features 4 gk .. 4 gk + 3 with the bias as pseudo-feature F.
void split3_bf16(float v, uint16_t (&piece)[3]) {
    for (int i = 0; i < 3; ++i) {
        piece[i] = to_bf16(v);
        const uint32_t b = (uint32_t)piece[i] << 16;
        float f;
        std::memcpy(&f, &b, 4);
        v -= f;                      // exact
    }
}

// (<= 20 units, <= 15 inputs, no use_delta: the caller checks x3_eligible)
bool x3_eligible(const pe_params& p, const pe_gru_layer& L) { return p.gru_precision == 0 && !p.use_delta && L.units <= 20 && L.n_in <= 15 && p.n_mfcc <= kRowFloats; }

int pack_gru_weights_x3(pe_engine* e, const pe_gru_layer& L, const float* dense_kernel) {
    const int H = L.units, F = L.n_in;
    std::vector<uint32_t> blob((size_t)kX3BlobBytes / 4, 0u);
    uint16_t* const half = reinterpret_cast<uint16_t*>(blob.data());
    auto gate_col = [&](int tile, int i, int* col) -> bool {     // A row i of an output tile -> column of the Keras matrices
        int gate, unit;
        if (tile < 3) { gate = tile; unit = i; }
        else { gate = i & 3; unit = 16 + (i >> 2); if (gate == 3) return false; }
        if (unit >= H) return false;
        *col = gate * H + unit;
        return true;
    };
    static const int kPiece3[8] = {0, 0, 1, 1, 0, 2, -1, -1};     // operand 3: [hi, hi, mid, mid, hi, lo, -, -]
    for (int tile = 0; tile < kX3Tiles; ++tile)
        for (int lane = 0; lane < 64; ++lane) {
            const int i = lane & 15, gk = lane >> 4;
            int col;
            if (!gate_col(tile, i, &col)) continue;
            for (int m = 0; m < kX3RecOps; ++m)
                for (int ek = 0; ek < 8; ++ek) {
                    const int src = m < 3 ? 4 * gk + (ek & 3) : 16 + gk;
                    const int piece = m == 0 ? 0 : m == 1 ? 1 : m == 2 ? (ek < 4 ? 0 : 2) : kPiece3[ek];
                    if (src >= H || piece < 0) continue;
                    uint16_t pc[3];
                    split3_bf16(L.recurrent_kernel[(size_t)src * 3 * H + col], pc);
                    half[((size_t)(kX3ArOff + (tile * kX3RecOps + m) * 64 + lane)) * 8 + ek] = pc[piece];
                }
            for (int m = 0; m < kX3InOps; ++m)
                for (int ek = 0; ek < 8; ++ek) {
                    const int f = 4 * gk + (ek & 3);
                    const int piece = m == 0 ? 0 : m == 1 ? 1 : (ek < 4 ? 0 : 2);
                    if (f > F) continue;
                    uint16_t pc[3];
                    split3_bf16(f < F ? L.kernel[(size_t)f * 3 * H + col] : L.bias[col], pc);
                    half[((size_t)(kX3AxOff + (tile * kX3InOps + m) * 64 + lane)) * 8 + ek] = pc[piece];
                }
        }
    float* const wd = reinterpret_cast<float*>(blob.data()) + (size_t)kX3WdOff * 4;
    for (int o = 0; o < 5; ++o)
        for (int lane = 0; lane < 64; ++lane) {
            const int u = o < 4 ? 4 * (lane >> 4) + o : 16 + (lane >> 4);
            if (u < H) wd[o * 64 + lane] = dense_kernel[u];
        }
    return dev_upload(e, &e->x3_blob, blob);
}

// bf16 network of <= 20 units in the five-values-per-lane layout (gru_b20_device.h): output tiles 0..2 = z / r / candidate
// of units 0..15 (row i <-> unit i), tile 3 row 4 g + q = gate q of unit 16 + g; recurrent k-slot 8 gk + e <-> source unit
// 4 gk + e (e < 4) / 16 + gk (e = 4); input k-slot 8 gk + e <-> feature 4 gk + e (e < 4; pseudo-features F, F + 1 = bias hi,
// lo), and 8 gk + 4 + e <-> the first difference of that feature (use_delta: kernel rows F .. 2 F - 1).
bool b20_eligible(const pe_params& p, const pe_gru_layer& L) {
    const int F = p.use_delta ? L.n_in / 2 : L.n_in;
    return p.gru_precision == 1 && L.units <= 20 && F <= 14 && p.n_mfcc <= kRowFloats;
}

int pack_gru_weights_b20(pe_engine* e, const pe_gru_layer& L, const float* dense_kernel) {
    const bool delta = e->prm.use_delta != 0;
    const int H = L.units, F = delta ? L.n_in / 2 : L.n_in;
    std::vector<uint32_t> blob((size_t)kB20BlobBytes / 4, 0u);
    uint16_t* const half = reinterpret_cast<uint16_t*>(blob.data());
    for (int tile = 0; tile < kB20Tiles; ++tile)
        for (int lane = 0; lane < 64; ++lane) {
            const int i = lane & 15, gk = lane >> 4;
            int gate, unit;
            if (tile < 3) { gate = tile; unit = i; }
            else { gate = i & 3; unit = 16 + (i >> 2); if (gate == 3) continue; }
            if (unit >= H) continue;
            const int col = gate * H + unit;
            for (int ek = 0; ek < 5; ++ek) {
                const int src = ek < 4 ? 4 * gk + ek : 16 + gk;
                if (src < H) half[((size_t)(kB20ArOff + tile * 64 + lane)) * 8 + ek] = to_bf16(L.recurrent_kernel[(size_t)src * 3 * H + col]);
            }
            for (int ek = 0; ek < 8; ++ek) {
                const int f = 4 * gk + (ek & 3);
                uint16_t v = 0;
                if (ek < 4) {
                    if (f < F) v = to_bf16(L.kernel[(size_t)f * 3 * H + col]);
                    else if (f == F || f == F + 1) {
                        const float b = L.bias[col];
                        const uint16_t hi = to_bf16(b);
                        const uint32_t hb = (uint32_t)hi << 16;
                        float hif;
                        std::memcpy(&hif, &hb, 4);
                        v = f == F ? hi : to_bf16(b - hif);
                    }
                } else if (delta && f < F) v = to_bf16(L.kernel[(size_t)(F + f) * 3 * H + col]);
                half[((size_t)(kB20AxOff + tile * 64 + lane)) * 8 + ek] = v;
            }
        }
    float* const wd = reinterpret_cast<float*>(blob.data()) + (size_t)kB20WdOff * 4;
    for (int o = 0; o < 5; ++o)
        for (int lane = 0; lane < 64; ++lane) {
            const int u = o < 4 ? 4 * (lane >> 4) + o : 16 + (lane >> 4);
            if (u < H) wd[o * 64 + lane] = dense_kernel[u];
        }
    return dev_upload(e, &e->b20_blob, blob);
}

// Wide / stacked network (gru_wide_device.h): wave w owns output tiles tau = w TPW + t of every gate;
// row i of a tile <-> unit 16 tau + 4 (i & 3) + (i >> 2); k-step rho, k-slot gk <-> source unit 4 rho + gk
// (layer 0 input: k-step kk <-> feature 4 gk + kk).  Streams: [wave][k-group][tile][lane] float4.
int pack_gru_weights_wide(pe_engine* e, const pe_weights* w) {
    const int H = w->layers[0].units, WV = gru_wide_waves(H), TPW = H / (16 * WV), H16 = H / 16;
    for (int l = 0; l < w->n_layers; ++l) {
        const pe_gru_layer& L = w->layers[l];
        const int kx4 = l == 0 ? 1 : H16;
        const int F = L.n_in;
        e->wide_kx4[l] = kx4;
        auto in_weight = [&](int rho, int gk, int col) -> float {      // input part, k-step rho, k-slot gk
            if (l == 0) { const int phi = 4 * gk + rho; return phi < F ? L.kernel[(size_t)phi * 3 * H + col] : 0.f; }
            return L.kernel[(size_t)(4 * rho + gk) * 3 * H + col];
        };
        for (int phase = 0; phase < 2; ++phase) {
            const int NT = phase == 0 ? 2 * TPW : TPW;
            std::vector<float> wx((size_t)WV * kx4 * NT * 64 * 4, 0.f), wr((size_t)WV * H16 * NT * 64 * 4, 0.f);
            std::vector<float> bias((size_t)WV * NT * 4 * 64, 0.f);
            for (int wv = 0; wv < WV; ++wv)
                for (int tl = 0; tl < NT; ++tl) {
                    const int gate = phase == 0 ? (tl < TPW ? 0 : 1) : 2;
                    const int tau = wv * TPW + (tl % TPW);
                    for (int lane = 0; lane < 64; ++lane) {
                        const int i = lane & 15, gk = lane >> 4;
                        const int col = gate * H + 16 * tau + 4 * (i & 3) + (i >> 2);
                        for (int r4 = 0; r4 < kx4; ++r4)
                            for (int q = 0; q < 4; ++q)
                                wx[((((size_t)wv * kx4 + r4) * NT + tl) * 64 + lane) * 4 + q] = in_weight(4 * r4 + q, gk, col);
                        for (int r4 = 0; r4 < H16; ++r4)
                            for (int q = 0; q < 4; ++q)
                                wr[((((size_t)wv * H16 + r4) * NT + tl) * 64 + lane) * 4 + q] =
                                    L.recurrent_kernel[(size_t)(4 * (4 * r4 + q) + gk) * 3 * H + col];
                        for (int q = 0; q < 4; ++q)      // C operand: this lane's output rows 4 gk + q
                            bias[(((size_t)wv * NT + tl) * 4 + q) * 64 + lane] = L.bias[gate * H + 16 * tau + 4 * q + gk];
                    }
                }
            int rc;
            if ((rc = dev_upload(e, &e->wide_buf[l][phase == 0 ? 0 : 2], wx))) return rc;
            if ((rc = dev_upload(e, &e->wide_buf[l][phase == 0 ? 1 : 3], wr))) return rc;
            if ((rc = dev_upload(e, &e->wide_buf[l][phase == 0 ? 4 : 5], bias))) return rc;
        }
    }
    std::vector<float> wd((size_t)WV * TPW * 4 * 64, 0.f);
    for (int wv = 0; wv < WV; ++wv)
        for (int tp = 0; tp < TPW; ++tp)
            for (int q = 0; q < 4; ++q)
                for (int lane = 0; lane < 64; ++lane)
                    wd[(((size_t)wv * TPW + tp) * 4 + q) * 64 + lane] = w->dense_kernel[16 * (wv * TPW + tp) + 4 * q + (lane >> 4)];
    return dev_upload(e, &e->wide_wd, wd);
}

// The same network for gru_wide_x3_device.h: float32 weights (split into bf16 pieces in registers, every timestep), wave w
// owns output tiles tau = w TPW + t of every gate; row i of a tile <-> unit 16 tau + i; k-group kappa, lane (row i, k-group
// slot gk) <-> source units 16 kappa + 4 gk + e, e = 0..3 (layer 0 input: feature 4 gk + e).  Streams: [wave][kappa][tile][lane] float4.
int pack_gru_weights_wide_x3(pe_engine* e, const pe_weights* w) {
    const int H = w->layers[0].units, WV = 4, TPW = H / (16 * WV), H16 = H / 16;
    for (int l = 0; l < w->n_layers; ++l) {
        const pe_gru_layer& L = w->layers[l];
        const int kin = l == 0 ? 1 : H16;
        const int F = L.n_in;
        for (int phase = 0; phase < 2; ++phase) {
            const int NT = phase == 0 ? 2 * TPW : TPW;
            std::vector<float> wx((size_t)WV * kin * NT * 64 * 4, 0.f), wr((size_t)WV * H16 * NT * 64 * 4, 0.f);
            std::vector<float> bias((size_t)WV * NT * 4 * 64, 0.f);
            for (int wv = 0; wv < WV; ++wv)
                for (int tl = 0; tl < NT; ++tl) {
                    const int gate = phase == 0 ? (tl < TPW ? 0 : 1) : 2;
                    const int tau = wv * TPW + (tl % TPW);
                    for (int lane = 0; lane < 64; ++lane) {
                        const int i = lane & 15, gk = lane >> 4;
                        const int col = gate * H + 16 * tau + i;
                        // gate-major streams: [gate within the phase][wave][k-group][tile of the gate][lane] float4 (the kernel walks
                        // one gate's TPW tiles at a time: wide_x3_accumulate)
                        const int gsel = phase == 0 ? tl / TPW : 0, tp = tl % TPW;
                        for (int kap = 0; kap < kin; ++kap)
                            for (int q = 0; q < 4; ++q) {
                                const int src = 16 * kap + 4 * gk + q;
                                wx[(((((size_t)gsel * WV + wv) * kin + kap) * TPW + tp) * 64 + lane) * 4 + q] = src < F ? L.kernel[(size_t)src * 3 * H + col] : 0.f;
                            }
                        for (int kap = 0; kap < H16; ++kap)
                            for (int q = 0; q < 4; ++q)
                                wr[(((((size_t)gsel * WV + wv) * H16 + kap) * TPW + tp) * 64 + lane) * 4 + q] =
                                    L.recurrent_kernel[(size_t)(16 * kap + 4 * gk + q) * 3 * H + col];
                        for (int q = 0; q < 4; ++q)      // C operand: this lane's output rows 4 gk + q
                            bias[(((size_t)wv * NT + tl) * 4 + q) * 64 + lane] = L.bias[gate * H + 16 * tau + 4 * gk + q];
                    }
                }
            int rc;
            if ((rc = dev_upload(e, &e->wide3_buf[l][phase == 0 ? 0 : 2], wx))) return rc;
            if ((rc = dev_upload(e, &e->wide3_buf[l][phase == 0 ? 1 : 3], wr))) return rc;
            if ((rc = dev_upload(e, &e->wide3_buf[l][phase == 0 ? 4 : 5], bias))) return rc;
        }
    }
    std::vector<float> wd((size_t)WV * TPW * 4 * 64, 0.f);
    for (int wv = 0; wv < WV; ++wv)
        for (int tp = 0; tp < TPW; ++tp)
            for (int q = 0; q < 4; ++q)
                for (int lane = 0; lane < 64; ++lane)
                    wd[(((size_t)wv * TPW + tp) * 4 + q) * 64 + lane] = w->dense_kernel[16 * (wv * TPW + tp) + 4 * (lane >> 4) + q];
    return dev_upload(e, &e->wide3_wd, wd);
}

// Samples of the virtual stream that must have arrived, counted from a frame's first sample, before the
// reference's Listener has that frame in its window: the whole analysis window (sonopy emits one frame per
// full window), plus one hop for the legacy speechpy front end, whose stack_frames returns
// floor((len - window) / hop) frames -- one fewer.
int emit_window(const pe_params& p) { return p.window_samples + (p.vectorizer == 3 ? p.hop_samples : 0); }
int frame_len_of(const pe_params& p) { return p.window_samples < p.n_fft ? p.window_samples : p.n_fft; }
// frames computed (first frame_len samples arrived) but not yet emitted: at most this many exist at any time
int pending_frames(const pe_params& p) { return (emit_window(p) - frame_len_of(p) + p.hop_samples - 1) / p.hop_samples; }
// vectorize_raw on a whole buffer (vectorization.py:46-50): frames the vectorizer returns for n samples
int64_t frames_of_buffer(const pe_params& p, int64_t n) {
    const int64_t ew = emit_window(p);
    return n >= ew ? 1 + (n - ew) / p.hop_samples : 0;
}

StreamGeom geom(const pe_engine* e) {
    StreamGeom g;
    g.n_streams = e->n_streams;
    g.log_mode = e->prm.vectorizer == 3 ? 1 : 0;
    g.window = emit_window(e->prm);
    g.hop = e->prm.hop_samples;
    g.frame_len = frame_len_of(e->prm);
    g.n_filt = e->prm.n_filt;
    g.n_mfcc = e->prm.n_mfcc;
    g.n_features = e->prm.n_features;
    g.ring_slots = e->ring_slots;
    return g;
}

template <class R>
WaveTables<R> tables(const pe_engine* e) {
    WaveTables<R> t;
    t.blob = e->table_blob;
    t.L = e->table_layout;
    return t;
}

// the most frames one stream can complete in a call of n_updates chunks (a full carry plus the new samples)
int max_frames_per_call(const pe_engine* e, long long new_samples) {
    const long long fmax = ((long long)(e->carry_cap - 1) + new_samples - frame_len_of(e->prm)) / e->prm.hop_samples + 1;
    return (int)(fmax < 1 ? 1 : fmax);
}

StreamState state_of(const pe_engine* e, uint32_t call) {
    StreamState st;
    st.rec = e->rec; st.carry = e->carry; st.n_padded = (uint32_t)e->n_padded; st.call = call;
    return st;
}

// Number of a call that writes records.  Call numbers are 32 bits; only the order of a stream's two sides and "written by
// this very call" are ever read from them, so long before the count wraps every record is renumbered (current side 2, other
// side 1; on the stream this call launches on, in front of its launches) and the count restarts.
int begin_state_call(pe_engine* e, hipStream_t s, uint32_t* call) {
    if (e->call_no >= e->renumber_at) {
        PE_HIP(e, launch_renumber(state_of(e, e->call_no + 1u), e->n_padded, s));
        e->call_no = 2u;
    }
    *call = ++e->call_no;
    return PE_OK;
}

// ids / n_active: the streams of this launch (pe_update_subset; device pointer), or null / 0 = all of them in order
template <class R>
MfccStreamArgs<R> mfcc_args(const pe_engine* e, const int16_t* pcm_dev, int chunk, uint32_t call, const int32_t* ids = nullptr, int n_active = 0) {
    MfccStreamArgs<R> a;
    a.geo = geom(e);
    if (ids) a.geo.n_streams = n_active;
    a.pcm = pcm_dev;
    a.chunk = chunk;
    a.pcm_pairs_ok = ((chunk & 1) == 0) && ((reinterpret_cast<uintptr_t>(pcm_dev) & 3u) == 0);
    a.ids = ids;
    a.st = state_of(e, call);
    a.head = ids ? nullptr : e->call_head; a.head_chunk = e->call_head_chunk; a.keep = (!ids && e->call_keep) ? 1 : 0;
    a.ring = e->ring; a.ring_bf16 = e->prm.ring_precision;
    a.proj_ring = e->proj_on ? e->proj_ring : nullptr;
    a.n_updates = 1; a.ke_hist = nullptr; a.n_padded = e->n_padded;
    a.n_frame_rows = max_frames_per_call(e, chunk);
    a.div_hop = FastDiv::make((uint32_t)e->prm.hop_samples);
    a.div_chunk = FastDiv::make((uint32_t)chunk);
    return a;
}

template <class R>
GeneralStreamArgs<R> general_args(const pe_engine* e, const int16_t* pcm_dev, int chunk, uint32_t call, const int32_t* ids = nullptr, int n_active = 0) {
    GeneralStreamArgs<R> a{};
    a.geo = geom(e);
    if (ids) a.geo.n_streams = n_active;
    a.tab = e->gtab;
    a.pcm = pcm_dev; a.chunk = chunk;
    a.pcm_pairs_ok = ((chunk & 1) == 0) && ((reinterpret_cast<uintptr_t>(pcm_dev) & 3u) == 0);
    a.ids = ids;
    a.st = state_of(e, call); a.carry_cap = e->carry_cap;
    a.ring = e->ring; a.row_floats = e->row_floats;
    a.ring_bf16 = e->prm.ring_precision == 1;
    a.n_updates = 1; a.div_chunk = FastDiv::make((uint32_t)chunk); a.ke_hist = nullptr; a.n_padded = e->n_padded;
    return a;
}

// MFCC alone: every stream of the launch goes from its current side to the other one (records stamped `call`).
int launch_mfcc(pe_engine* e, const int16_t* pcm_dev, int chunk, hipStream_t s, uint32_t call, const int32_t* ids = nullptr, int n_active = 0) {
    // (no cap on the frames ONE update may complete: Listener.update takes a chunk of any length,
    //  network_runner.py:125-146; the frame tasks skip every row that the ring would overwrite again -- v_first in
    //  mfcc_frame_tasks -- so a long chunk costs its last ring_slots frames plus a scalar walk over the row indices)
    if (e->general) {
        if (e->prm.mfcc_precision == 0) PE_HIP(e, launch_general_stream_f64(general_args<double>(e, pcm_dev, chunk, call, ids, n_active), s));
        else PE_HIP(e, launch_general_stream_f32(general_args<float>(e, pcm_dev, chunk, call, ids, n_active), s));
    } else if (e->prm.mfcc_precision == 0) PE_HIP(e, launch_mfcc_f64(mfcc_args<double>(e, pcm_dev, chunk, call, ids, n_active), tables<double>(e), e->n_cus, s));
    else PE_HIP(e, launch_mfcc_f32(mfcc_args<float>(e, pcm_dev, chunk, call, ids, n_active), tables<float>(e), e->n_cus, s));
    return PE_OK;
}

GruArgs gru_args(const pe_engine* e) {
    GruArgs a{};
    a.n_streams = e->n_streams;
    a.n_features = e->prm.n_features;
    a.n_in = e->n_in;
    a.units = e->units;
    a.wxd = e->wxd; a.use_delta = e->prm.use_delta;
    a.wx = e->wx; a.wr1 = e->wr1; a.wr2 = e->wr2; a.bias = e->bias; a.wd = e->wd;
    a.dense_bias = e->dense_bias;
    a.ring = e->ring; a.ring_slots = e->ring_slots; a.ring_bf16 = e->prm.ring_precision;
    a.ids = nullptr; a.rec = e->rec; a.n_padded = (uint32_t)e->n_padded; a.ke_plain = nullptr;
    a.call = e->call_no + 1u;           // a reader behind every call so far (the network role of a fused launch: that call's own number)
    a.proj_ring = e->proj_on ? e->proj_ring : nullptr;
    a.predict_ke = 0;
    a.chunk = 0;
    a.window = emit_window(e->prm); a.hop = e->prm.hop_samples;
    a.frame_len = frame_len_of(e->prm);
    a.bf16 = e->prm.gru_precision == 1;
    a.wx_bf16 = e->wx_bf16; a.wr_bf16 = e->wr_bf16; a.wd_bf16 = e->wd_bf16;
    a.b20 = e->gru_tiling == 0 ? nullptr : e->b20_blob;      // bf16 network: five values per lane where it fits (pe_set_gru_tiling(e, 0): eight)
    a.feats = nullptr; a.out = nullptr; a.row_stride = 0;
    a.row_floats = e->row_floats;
    // Four waves per tile cut the latency of a tile's chain; that only pays while every tile gets a CU of its
    // own next to one MFCC workgroup.  Measured (fused, f64 front end; tools/gpu_policy.py): 4096 streams
    // 21.9 us (4 waves) vs 32.0 (1); 8192: 41.5 vs 33.5; 16384: 76.0 vs 55.3; 65536: 284 vs 199.
    // Stock width: the re-tiled shapes (three full tiles + partial sums; gru_cw_device.h) cut the four-wave kernel's
    // timestep (14.2 vs 16.3 us per window chain, stand-alone) but cost the one-wave kernel 5 % in the throughput
    // regime (two-pass MFMAs + reductions: 81.0 vs 77.3 us at 65 536 streams), so an engine takes ONE tiling for all
    // of its launches -- every shape of a tiling agrees bit for bit -- by its size.  The critical-wave kernel still
    // wins with two tiles per compute unit (8192 streams, fused: 272 vs 254 M windows/s against one wave per tile).
    // Engines with more stream tiles than the machine has SIMDs (> 4 tiles per compute unit: more than 16 384 streams on
    // MI355X) take the XDL form of the float32 network (gru_x3_device.h: every operand as three bf16 pieces): an f32-input
    // MFMA keeps its whole SIMD from issuing for its 8 passes, the bf16 MFMAs cost half the cycles for the same products and
    // do not.  Measured per update, two launches against the fused classic tiling: 54 vs 69 us at 20 480 streams, 59 vs 69 at
    // 24 576, 69 vs 79 at 32 768, 128 vs 154 us at 65 536 -- and 48 vs 43 us at 16 384 (one tile per SIMD: the classic network
    // still runs in one round of waves, and the missing fused launch costs more than the cheaper network saves).
    const bool x3_ok = e->x3_blob && !a.bf16 && !e->wide && !a.proj_ring && e->row_floats == kRowFloats;
    // ... and so do engines reserved for several updates per call (pe_reserve_updates) whose batched network launch has more
    // (update, tile) windows than the machine has SIMDs: ONE form per engine (every launch of a form agrees bit for bit), chosen
    // for the launch the engine was reserved for -- its single updates then run the one-wave XDL kernel in two launches.
    // Measured at 4096 streams x 8 updates per call: 61.6 us for the batched network on the re-tiled f32-input form against
    // the XDL form's rate of ~26 us for the same 32 768 windows (profiles/round5/r5zz_kernel_stats.csv; round 6: profiles/round6).
    const bool many_windows = e->max_updates > 1 && (long long)e->max_updates * e->n_tiles > 4LL * e->n_cus;
    a.x3 = x3_ok && (e->gru_tiling == 2 || (e->gru_tiling < 0 && (e->n_tiles > 4 * e->n_cus || many_windows))) ? e->x3_blob : nullptr;
    const bool cw_ok = e->cw_blob && e->row_floats == kRowFloats && !a.proj_ring && !a.bf16 && !a.x3 && !e->wide;
    const bool retile = cw_ok && (e->gru_tiling == 1 || (e->gru_tiling < 0 && e->n_tiles <= 2 * e->n_cus));
    const int auto_waves = retile ? (e->n_tiles <= 2 * e->n_cus ? 4 : 1) : (e->n_tiles <= e->n_cus ? 4 : 1);
    a.waves_per_tile = a.x3 ? 1 : e->gru_waves ? e->gru_waves : auto_waves;
    if (e->prm.use_delta && !retile) a.waves_per_tile = 1;       // (classic tiling: only the one-wave kernel carries the delta inputs)
    if (e->row_floats != kRowFloats && !(gru_small_regs(e->units) == 5 && !e->prm.use_delta && a.waves_per_tile == 4))
        a.waves_per_tile = 1;       // ... and the 32-float feature rows (stock width: four waves per tile exist, gru_tile_mw5<.., 2>)
    a.cw = retile ? e->cw_blob : nullptr;
    return a;
}

// which form a wide engine's launches take: the XDL form when asked for (pe_set_gru_tiling(e, 2)); the default stays the
// f32-input MFMA kernel -- measured equal (1.26 ms per launch at 256 x 2, profiles/round5/r5f_wide_forms.log) and float32-exact
bool wide_uses_x3(const pe_engine* e) { return e->wide && e->gru_tiling == 2; }

// Network launch for any input mode (0 explicit batch, 1 ring, 2 row sequence)
int launch_network(pe_engine* e, const GruArgs& g, int mode, hipStream_t s) {
    if (e->wide) {
        // two forms of the streamed-weight network (pe_set_gru_tiling): 0 = f32-input MFMAs (gru_wide_device.h), 2 = float32
        // products on the bf16 pipe with the float32 weights split in registers (gru_wide_x3_device.h)
        const bool x3 = wide_uses_x3(e);
        float* const (*buf)[6] = x3 ? e->wide3_buf : e->wide_buf;
        WideArgs wa{};
        wa.base = g;
        wa.n_layers = e->n_layers;
        wa.units = e->units;
        wa.wd = x3 ? e->wide3_wd : e->wide_wd;
        for (int l = 0; l < e->n_layers; ++l) {
            wa.layer[l].wx1 = reinterpret_cast<const float4*>(buf[l][0]);
            wa.layer[l].wr1 = reinterpret_cast<const float4*>(buf[l][1]);
            wa.layer[l].wx2 = reinterpret_cast<const float4*>(buf[l][2]);
            wa.layer[l].wr2 = reinterpret_cast<const float4*>(buf[l][3]);
            wa.layer[l].b1 = buf[l][4];
            wa.layer[l].b2 = buf[l][5];
            wa.layer[l].kx4 = e->wide_kx4[l];
        }
        if (x3) PE_HIP(e, launch_gru_wide_x3(wa, mode, s));
        else PE_HIP(e, launch_gru_wide(wa, mode, s));
        return PE_OK;
    }
    PE_HIP(e, launch_gru_small(g, mode, s));
    return PE_OK;
}

int launch_gru_ring(pe_engine* e, float* out_dev, hipStream_t s, const int32_t* ids = nullptr, int n_active = 0) {
    GruArgs a = gru_args(e);
    a.out = out_dev;
    if (ids) { a.ids = ids; a.n_streams = n_active; }
    return launch_network(e, a, 1, s);
}

// ---- host-fed pipeline -------------------------------------------------------------------------------------------
bool is_pinned(const pe_engine* e, const void* p, size_t bytes) {
    const char* c = static_cast<const char*>(p);
    for (const auto& r : e->pinned)
        if (c >= r.first && c + bytes <= r.first + r.second) return true;
    return false;
}

int async_init(pe_engine* e) {
    if (e->s_compute) return PE_OK;
    PE_HIP(e, hipStreamCreateWithFlags(&e->s_copy, hipStreamNonBlocking));
    PE_HIP(e, hipStreamCreateWithFlags(&e->s_compute, hipStreamNonBlocking));
    for (auto& sl : e->aslot) {
        PE_HIP(e, hipEventCreateWithFlags(&sl.copied, hipEventDisableTiming));
        PE_HIP(e, hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    }
    PE_HIP(e, hipEventCreateWithFlags(&e->ev_user, hipEventDisableTiming));
    return PE_OK;
}

int ensure_pinned(pe_engine* e, void** p, size_t* have, size_t bytes) {
    if (*have >= bytes) return PE_OK;
    if (*p) { (void)hipHostFree(*p); *p = nullptr; *have = 0; }
    hipError_t err = hipHostMalloc(p, bytes, hipHostMallocDefault);
    if (err != hipSuccess) return fail(e, PE_ERR_NOMEM, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(err));
    *have = bytes;
    return PE_OK;
}

// the update in this slot has finished: hand its probabilities to the caller
int finish_slot(pe_engine* e, pe_engine::AsyncSlot& sl) {
    if (!sl.busy) return PE_OK;
    const hipError_t err = hipEventSynchronize(sl.done);
    sl.busy = false;                                  // (also when the wait failed: a slot that stays busy would fail every later drain)
    --e->async_inflight;
    if (err != hipSuccess) return fail(e, PE_ERR_HIP, "hipEventSynchronize(update in flight) failed: %s", hipGetErrorString(err));
    if (!sl.direct_out) std::memcpy(sl.user_out, sl.pin_out, sl.out_bytes);
    return PE_OK;
}

int drain_async(pe_engine* e) {
    if (!e || e->async_inflight == 0) return PE_OK;
    int first_rc = PE_OK;
    for (int i = 0; i < pe_engine::kAsyncDepth; ++i) {                 // oldest first; a failed slot does not keep the others in flight
        const int rc = finish_slot(e, e->aslot[(e->async_next + i) % pe_engine::kAsyncDepth]);
        if (rc && !first_rc) first_rc = rc;
    }
    return first_rc;
}

// a *_device entry point is about to enqueue state-moving work on the caller's stream (see pe_engine::last_user_stream)
void note_user_stream(pe_engine* e, void* stream) {
    e->last_user_stream = static_cast<hipStream_t>(stream);
    e->user_dirty = true;
}

int check_chunk(pe_engine* e, const void* pcm, int chunk) {
    if (!e) return PE_ERR_INVALID;
    if (chunk == 0) return fail(e, PE_ERR_EOF, "empty chunk");
    if (chunk < 0 || !pcm) return fail(e, PE_ERR_INVALID, "bad pcm buffer / chunk_samples=%d", chunk);
    return PE_OK;
}

// Leaving the keep style: the leftovers that still lie in the last call's chunks go to the carry (one small launch on the
// stream of the call that needs them there), and the engine forgets the chunks.
int flush_kept(pe_engine* e, hipStream_t s) {
    if (!e->kept) return PE_OK;
    PE_HIP(e, launch_materialize_carry(state_of(e, e->call_no + 1u), e->kept, e->kept_chunk, e->n_streams, s));
    e->kept = nullptr; e->kept_chunk = 0;
    return PE_OK;
}

// May the leftover of an update of these chunks stay in them?  The stock front end (the general one keeps its own carry
// arithmetic), whole sample pairs (the frame role's dword loads), and a chunk that holds any leftover (< frame_len samples).
bool keep_eligible(const pe_engine* e, const int16_t* pcm_dev, int chunk) {
    return !e->general && (chunk & 1) == 0 && (reinterpret_cast<uintptr_t>(pcm_dev) & 3u) == 0 && chunk >= frame_len_of(e->prm) - 1
        && chunk >= e->carry_cap - 1;
}

// True when no frame computed by an update of `chunk` samples can become visible in that same
// update (it needs window - frame_len more samples), so the network does not depend on it.
bool can_fuse(const pe_engine* e, int chunk) {
    // (the fused kernels exist for the stock table shape: filterbanks whose runs need the wide loop bounds take two launches;
    //  so do networks of 21..32 units on the one-wave kernel: their register-resident weights do not fit beside the
    //  frame role's 128-register budget -- fused, that shape spilled 96-240 bytes per lane into the time loop)
    if (!(e->fused && !e->general && !e->wide && e->table_layout.mel_pad == 10 && chunk <= emit_window(e->prm) - frame_len_of(e->prm))) return false;
    // (gru_x3_device.h: ~220 registers per lane against the frame role's ~100, and one kernel has one budget.  The same
    //  concurrency as TWO kernels -- the network on a side stream between two events -- was built and measured: 142 vs
    //  134 us per update at 65 536 streams, 40 vs 24 us at 4096: the event hand-offs cost more than the overlap buys)
    if (gru_args(e).x3) return false;
    if (e->prm.gru_precision == 0 && gru_small_regs(e->units) >= 6 && gru_args(e).waves_per_tile != 4) return false;
    if (e->prm.use_delta && e->prm.gru_precision == 0) {        // re-tiled one-wave shape with delta inputs: no fused instantiation
        // (waves_per_tile == 4 is not enough: the four-wave shape needs the 32-slot ring it stages in LDS -- after
        //  pe_reserve_updates, or with a longer feature window, the launcher falls back to the one-wave shape, whose
        //  fused instantiation has no delta inputs)
        const GruArgs g = gru_args(e);
        if (g.cw && (g.waves_per_tile != 4 || e->ring_slots != kCwSlots || e->prm.n_features > kCwSlots)) return false;
    }
    return true;
}

// ids / n_active: the streams that take part (pe_update_subset; device pointer, PCM rows and outputs in that order), or
// null / 0 = every stream
// keep: the caller promises that pcm_dev stays alive and unchanged until the NEXT state-moving call's work has completed
// (pe_update_device_keep; pe_update_async: the engine's own ring of device buffers) -- the leftover then stays in the chunk
int do_update(pe_engine* e, const int16_t* pcm_dev, int chunk, float* raw_out_dev, float* feats_out_dev,
              hipStream_t s, const int32_t* ids = nullptr, int n_active = 0, bool keep = false) {
    int rc;
    const bool t = e->timing;
    const bool keep_call = keep && !ids && keep_eligible(e, pcm_dev, chunk);
    if (keep_call && e->kept) {
        // the previous call's chunks must still hold its leftovers: a caller that refills ONE buffer for every call has broken
        // the promise of pe_update_device_keep, and the samples are gone -- refuse instead of computing frames from the wrong audio
        const char* a0 = reinterpret_cast<const char*>(e->kept);
        const char* a1 = a0 + (size_t)e->n_streams * e->kept_chunk * sizeof(int16_t);
        const char* b0 = reinterpret_cast<const char*>(pcm_dev);
        const char* b1 = b0 + (size_t)e->n_streams * chunk * sizeof(int16_t);
        if (b0 < a1 && a0 < b1)
            return fail(e, PE_ERR_INVALID, "pe_update_device_keep: this call's chunks overlap the previous call's (%p), which still hold the "
                        "streams' leftover samples -- alternate between at least two buffers, or use pe_update_device", (const void*)e->kept);
    }
    if (!keep_call && (rc = flush_kept(e, s))) return rc;
    e->call_head = keep_call ? e->kept : nullptr; e->call_head_chunk = e->kept_chunk; e->call_keep = keep_call;
    // (whatever happens below, the launches of later calls must not inherit this call's style)
    struct Reset { pe_engine* e; ~Reset() { e->call_head = nullptr; e->call_head_chunk = 0; e->call_keep = false; } } reset{e};
    e->kept = keep_call ? pcm_dev : nullptr; e->kept_chunk = keep_call ? chunk : 0;
    uint32_t call = 0;
    if ((rc = begin_state_call(e, s, &call))) return rc;
    if (t) { PE_HIP(e, hipEventRecord(e->ev[0], s)); }
    if (raw_out_dev && can_fuse(e, chunk)) {
        GruArgs g = gru_args(e);             // the records as they stand BEFORE the update; the waves predict ke
        g.call = call;                       // (records this launch writes are not read by it)
        g.predict_ke = 1;
        g.chunk = chunk;
        g.out = raw_out_dev;
        if (ids) { g.ids = ids; g.n_streams = n_active; }
        if (e->prm.mfcc_precision == 0) PE_HIP(e, launch_fused_f64(mfcc_args<double>(e, pcm_dev, chunk, call, ids, n_active), tables<double>(e), g, e->n_cus, s));
        else PE_HIP(e, launch_fused_f32(mfcc_args<float>(e, pcm_dev, chunk, call, ids, n_active), tables<float>(e), g, e->n_cus, s));
        if (t) { PE_HIP(e, hipEventRecord(e->ev[1], s)); PE_HIP(e, hipEventRecord(e->ev[2], s)); e->ev_valid = true; e->ev_has_gru = false; }
    } else {
        if ((rc = launch_mfcc(e, pcm_dev, chunk, s, call, ids, n_active))) return rc;
        if (t) { PE_HIP(e, hipEventRecord(e->ev[1], s)); }
        if (raw_out_dev) {
            if ((rc = launch_gru_ring(e, raw_out_dev, s, ids, n_active))) return rc;
        }
        if (t) { PE_HIP(e, hipEventRecord(e->ev[2], s)); e->ev_valid = true; e->ev_has_gru = raw_out_dev != nullptr; }
    }
    if (feats_out_dev) {
        GatherArgs g{e->n_streams, e->prm.n_features, e->prm.n_mfcc, e->ring_slots, e->prm.ring_precision, e->ring, state_of(e, e->call_no + 1u), feats_out_dev, e->row_floats};
        PE_HIP(e, launch_gather(g, s));
    }
    return PE_OK;
}

}  // namespace

extern "C" {

int pe_abi_version(void) { return PE_ABI_VERSION; }
const char* pe_last_global_error(void) { return g_global_error.c_str(); }
const char* pe_last_error(const pe_engine* e) { return e ? e->err.c_str() : g_global_error.c_str(); }

int pe_create(const pe_params* p, const double* mel_filters, const pe_weights* w, int32_t n_streams,
              int32_t device, pe_engine** out) {
    if (!p || !mel_filters || !w || !out) return fail(nullptr, PE_ERR_INVALID, "null argument to pe_create");
    *out = nullptr;
    if (n_streams <= 0) return fail(nullptr, PE_ERR_INVALID, "n_streams must be positive, got %d", n_streams);
    // ListenerParams (params.py:28-118): the stock shape (n_fft = 512, <= 64 filters, <= 16 coefficients) runs on the
    // one-frame-per-wave kernel, every other shape on the general front end (mfcc_general_device.h)
    // ... any n_fft from 16 on: powers of two up to 2048 as a packed real transform, every other length up to 1024 (numpy's
    // rfft takes any n) through Bluestein's chirp-z form over the next power of two >= 2 n_fft - 1
    // (general_is_pow2: powers of two from 64 on; 16 and 32 run as Bluestein transforms over 128 points like every other short length)
    if (p->n_fft < 16 || p->n_fft > kGeneralMaxFft || (!general_is_pow2(p->n_fft) && p->n_fft > kGeneralMaxBlueFft))
        return fail(nullptr, PE_ERR_UNSUPPORTED, "n_fft must be any length in 16..%d or a power of two up to %d (got %d)", kGeneralMaxBlueFft, kGeneralMaxFft, p->n_fft);
    if (p->n_mfcc < 1 || p->n_mfcc > kGeneralMaxMfcc) return fail(nullptr, PE_ERR_UNSUPPORTED, "n_mfcc must be in 1..%d (got %d)", kGeneralMaxMfcc, p->n_mfcc);
    if (p->n_filt < 1 || p->n_filt > kGeneralMaxFilt || p->n_mfcc > p->n_filt)
        return fail(nullptr, PE_ERR_UNSUPPORTED, "need 1 <= n_mfcc <= n_filt <= %d (got n_filt=%d n_mfcc=%d)", kGeneralMaxFilt, p->n_filt, p->n_mfcc);
    bool general = p->n_fft != kNfft || p->n_filt > kMaxFilt || p->n_mfcc > kRowFloats;
    if (p->hop_samples < 1 || p->window_samples < 1 || p->n_features < 1)
        return fail(nullptr, PE_ERR_INVALID, "window/hop/n_features must be positive");
    if (p->mfcc_precision != 0 && p->mfcc_precision != 1) return fail(nullptr, PE_ERR_INVALID, "mfcc_precision must be 0 (f64) or 1 (f32)");
    if (p->gru_precision != 0 && p->gru_precision != 1) return fail(nullptr, PE_ERR_INVALID, "gru_precision must be 0 (f32) or 1 (bf16 operands)");
    if (p->vectorizer != 0 && p->vectorizer != 2 && p->vectorizer != 3)
        return fail(nullptr, PE_ERR_UNSUPPORTED, "vectorizer must be 2 (mfccs) or 3 (speechpy_mfccs); Vectorizer.mels (1) exists in the offline form only (pe_vectorize_mels), got %d", p->vectorizer);
    if (p->ring_precision != 0 && p->ring_precision != 1) return fail(nullptr, PE_ERR_INVALID, "ring_precision must be 0 (f32 rows) or 1 (bf16 rows)");
    if (p->ring_precision == 1 && p->gru_precision != 1) return fail(nullptr, PE_ERR_UNSUPPORTED, "bf16 feature rows (ring_precision = 1) feed the bf16-operand network only (gru_precision = 1)");
    if (w->n_layers < 1 || w->n_layers > 2 || !w->layers) return fail(nullptr, PE_ERR_UNSUPPORTED, "networks of 1 or 2 GRU layers have kernels (got %d layers)", w->n_layers);
    const pe_gru_layer& L = w->layers[0];
    const bool wide = w->n_layers == 2 || L.units > 32;
    if (!wide && L.units < 1) return fail(nullptr, PE_ERR_INVALID, "units must be positive");
    int wide_units = 0;          // width the streamed-weight kernel runs at: the layers' widths padded to a multiple of 64
    if (wide) {
        int hmax = L.units;
        if (w->n_layers == 2) {
            const pe_gru_layer& L2 = w->layers[1];
            if (!L2.kernel || !L2.recurrent_kernel || !L2.bias) return fail(nullptr, PE_ERR_INVALID, "null weight pointer");
            if (L2.n_in != L.units || L2.units < 1) return fail(nullptr, PE_ERR_INVALID, "layer 2 (n_in=%d, units=%d) does not follow layer 1 (units=%d)", L2.n_in, L2.units, L.units);
            if (L2.units > hmax) hmax = L2.units;
        }
        if (hmax > 256) return fail(nullptr, PE_ERR_UNSUPPORTED, "the streamed-weight GRU kernel holds up to 256 units per layer (got %d)", hmax);
        if (p->use_delta || p->gru_precision != 0) return fail(nullptr, PE_ERR_UNSUPPORTED, "use_delta / bf16 have no wide-GRU kernel");
        wide_units = (hmax + 63) / 64 * 64;
    }
    const int feature_size = p->use_delta ? 2 * p->n_mfcc : p->n_mfcc;          // params.py:99-109
    if (L.n_in != feature_size) return fail(nullptr, PE_ERR_INVALID, "layer n_in=%d does not match feature_size=%d", L.n_in, feature_size);
    if (!L.kernel || !L.recurrent_kernel || !L.bias || !w->dense_kernel) return fail(nullptr, PE_ERR_INVALID, "null weight pointer");

    hipError_t herr = hipSetDevice(device);
    if (herr != hipSuccess) return fail(nullptr, PE_ERR_HIP, "hipSetDevice(%d) failed: %s", device, hipGetErrorString(herr));

    {   // the library holds gfx950 code objects only, and every launch-shape rule in here (tiles per compute unit at which the
        // network changes form, frame workgroups per compute unit) was measured on MI355X: refuse anything else by name
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return fail(nullptr, PE_ERR_UNSUPPORTED, "device %d is %s: libprecise_engine.so is built for gfx950 (MI355X) only", device, prop.gcnArchName);
    }
    if (p->n_mfcc > kRowFloats && (wide || p->use_delta || p->gru_precision != 0))
        return fail(nullptr, PE_ERR_UNSUPPORTED, "more than 16 coefficients per frame feed the float32 network of <= 32 units without delta features only");
    // (bf16 operands / bf16 rows behind the general front end: the same network kernels, fed from rows of <= 16 coefficients --
    //  more than 16 were refused above)

    pe_engine* e = new pe_engine();
    e->prm = *p;
    e->device = device;
    if (!general) {
        // a stock-shape filterbank whose runs need more than the 64 lanes of the wave kernel also takes the general path
        std::vector<unsigned char> probe;
        pe_wave::Layout lprobe{};
        const std::string perr = p->mfcc_precision == 0 ? pe_wave::build<double>(mel_filters, p->n_filt, p->n_mfcc, probe, lprobe)
                                                        : pe_wave::build<float>(mel_filters, p->n_filt, p->n_mfcc, probe, lprobe);
        if (!perr.empty()) general = true;
    }
    e->general = general;
    e->row_floats = p->n_mfcc > kRowFloats ? 2 * kRowFloats : kRowFloats;
    {
        const int fl = frame_len_of(*p);
        e->carry_cap = general ? ((fl + 63) / 64 * 64 > kCarryCap ? (fl + 63) / 64 * 64 : kCarryCap) : kCarryCap;
    }
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) e->n_cus = cus;
    }
    e->n_streams = n_streams;
    e->n_tiles = (n_streams + kTileStreams - 1) / kTileStreams;
    e->n_padded = e->n_tiles * kTileStreams;
    e->units = L.units; e->n_in = p->n_mfcc; e->n_layers = w->n_layers; e->wide = wide;
    e->dense_bias = w->dense_bias;
    if (e->prm.vectorizer == 0) e->prm.vectorizer = 2;
    // frames computed (first flen samples arrived) but not yet emitted (whole window arrived):
    // at most ceil((window - flen) / hop) of them exist at any time; T + that many rows are live
    e->ring_slots = next_pow2(p->n_features + pending_frames(e->prm));

    int rc = PE_OK;
    do {
        if ((rc = dev_alloc(e, &e->carry, (size_t)2 * e->n_padded * e->carry_cap))) break;
        if ((rc = dev_alloc(e, &e->rec, (size_t)2 * e->n_padded))) break;
        if ((rc = dev_alloc(e, &e->ring, ring_floats(e)))) break;
        if (wide) {
            // Any width 33..256 (and stacked layers of different widths) runs on the streamed-weight kernel at the next
            // multiple of 64: a padded unit has zero weights and zero bias everywhere, so z = r = 1/2, candidate = 0 and
            // its state stays exactly 0 -- it neither receives nor contributes anything.
            const int Hp = wide_units;
            std::vector<std::vector<float>> kp(w->n_layers), rp(w->n_layers), bp(w->n_layers);
            std::vector<pe_gru_layer> lp(w->n_layers);
            for (int l = 0; l < w->n_layers; ++l) {
                const pe_gru_layer& Ls = w->layers[l];
                const int Hs = Ls.units, Fin = Ls.n_in, Fp = l == 0 ? Fin : Hp;
                kp[l].assign((size_t)Fp * 3 * Hp, 0.f); rp[l].assign((size_t)Hp * 3 * Hp, 0.f); bp[l].assign((size_t)3 * Hp, 0.f);
                for (int gate = 0; gate < 3; ++gate)
                    for (int u = 0; u < Hs; ++u) {
                        for (int k = 0; k < Fin; ++k) kp[l][(size_t)k * 3 * Hp + gate * Hp + u] = Ls.kernel[(size_t)k * 3 * Hs + gate * Hs + u];
                        for (int k = 0; k < Hs; ++k) rp[l][(size_t)k * 3 * Hp + gate * Hp + u] = Ls.recurrent_kernel[(size_t)k * 3 * Hs + gate * Hs + u];
                        bp[l][(size_t)gate * Hp + u] = Ls.bias[gate * Hs + u];
                    }
                lp[l] = pe_gru_layer{Fp, Hp, kp[l].data(), rp[l].data(), bp[l].data()};
            }
            std::vector<float> dp(Hp, 0.f);
            const int Hlast = w->layers[w->n_layers - 1].units;
            for (int u = 0; u < Hlast; ++u) dp[u] = w->dense_kernel[u];
            pe_weights wp{w->n_layers, lp.data(), dp.data(), w->dense_bias};
            e->units = Hp;
            if ((rc = pack_gru_weights_wide(e, &wp))) break;
            if ((rc = pack_gru_weights_wide_x3(e, &wp))) break;
        }
        else if ((rc = pack_gru_weights(e, L, w->dense_kernel))) break;
        if (p->gru_precision == 1 && (rc = pack_gru_weights_bf16(e, L, w->dense_kernel))) break;
        if (!wide && b20_eligible(*p, L) && (rc = pack_gru_weights_b20(e, L, w->dense_kernel))) break;
        if (!wide && x3_eligible(*p, L) && (rc = pack_gru_weights_x3(e, L, w->dense_kernel))) break;
        // the projection rows exist for the stock-width float32 network (3 R <= 16 slots: 4 output tiles, R = 5) fed
        // from the ring; they pay while the ring stays cache-resident (256 B per frame and stream)
        e->proj_ok = !wide && p->gru_precision == 0 && !p->use_delta && gru_small_regs(L.units) == 5 && !e->proj_w_host.empty();
        e->proj_on = false;          // opt-in (pe_set_input_projection): measured no gain for the MFMA kernels at <= 4096 streams
        if (e->general) rc = (p->mfcc_precision == 0) ? build_general_tables<double>(e, mel_filters) : build_general_tables<float>(e, mel_filters);
        else rc = (p->mfcc_precision == 0) ? build_tables<double>(e, mel_filters) : build_tables<float>(e, mel_filters);
        if (rc) break;
        if (e->proj_on && (rc = dev_alloc(e, &e->proj_ring, (size_t)e->n_tiles * e->ring_slots * kTileStreams * kProjRow))) break;
        for (auto& ev : e->ev)
            if (hipEventCreate(&ev) != hipSuccess) { rc = fail(e, PE_ERR_HIP, "hipEventCreate failed"); break; }
        if (rc) break;
        const size_t lds = e->general ? general_lds_bytes(p->mfcc_precision == 0 ? 8 : 4, p->n_fft, p->n_filt, e->gtab.n_rounds)
                                      : (size_t)e->table_layout.total + (size_t)kFrameWaves * pe_wave::kScratchReals * (p->mfcc_precision == 0 ? 8 : 4);
        if (lds > 64 * 1024) { rc = fail(e, PE_ERR_UNSUPPORTED, "MFCC kernel would need %zu bytes of LDS per workgroup", lds); break; }
        if ((rc = pe_clear(e, nullptr))) break;
        hipError_t se = hipDeviceSynchronize();
        if (se != hipSuccess) { rc = fail(e, PE_ERR_HIP, "device sync after init failed: %s", hipGetErrorString(se)); break; }
    } while (false);
    if (rc) {
        g_global_error = e->err;
        pe_destroy(e);
        return rc;
    }
    *out = e;
    return PE_OK;
}

int pe_destroy(pe_engine* e) {
    if (!e) return PE_OK;
    (void)hipSetDevice(e->device);
    (void)drain_async(e);
    for (void* p : e->allocs) (void)hipFree(p);
    for (DeviceBuf* b : {&e->st_pcm, &e->st_out, &e->st_feats, &e->st_mask, &e->st_audio, &e->st_mfcc, &e->st_conf, &e->st_fired, &e->st_ids})
        if (b->p) (void)hipFree(b->p);
    for (auto& ev : e->ev) if (ev) (void)hipEventDestroy(ev);
    if (e->s_compute) (void)hipStreamSynchronize(e->s_compute);
    if (e->s_copy) (void)hipStreamSynchronize(e->s_copy);
    for (auto& sl : e->aslot) {
        if (sl.pin_out) (void)hipHostFree(sl.pin_out);
        if (sl.dev_in.p) (void)hipFree(sl.dev_in.p);
        if (sl.dev_out.p) (void)hipFree(sl.dev_out.p);
        if (sl.copied) (void)hipEventDestroy(sl.copied);
        if (sl.done) (void)hipEventDestroy(sl.done);
    }
    if (e->ev_user) (void)hipEventDestroy(e->ev_user);
    if (e->s_copy) (void)hipStreamDestroy(e->s_copy);
    if (e->s_compute) (void)hipStreamDestroy(e->s_compute);
    for (auto& r : e->pinned) (void)hipHostFree(r.first);
    delete e;
    return PE_OK;
}

int pe_clear(pe_engine* e, const uint8_t* mask_host) {
    if (!e) return PE_ERR_INVALID;
    PE_HIP(e, hipSetDevice(e->device));
    PE_DRAIN(e);
    const uint8_t* mask_dev = nullptr;
    if (mask_host) {
        int rc = ensure(e, e->st_mask, (size_t)e->n_streams);
        if (rc) return rc;
        PE_HIP(e, hipMemcpy(e->st_mask.p, mask_host, (size_t)e->n_streams, hipMemcpyHostToDevice));
        mask_dev = static_cast<const uint8_t*>(e->st_mask.p);
    }
    // (a masked clear leaves the other streams' leftovers where they are: out of kept chunks first; a full clear empties them)
    if (mask_dev) { int frc = flush_kept(e, nullptr); if (frc) return frc; }
    else { e->kept = nullptr; e->kept_chunk = 0; }
    uint32_t call = 0;
    { int crc = begin_state_call(e, nullptr, &call); if (crc) return crc; }
    ClearArgs a{e->n_padded, e->ring_slots, mask_dev, state_of(e, call), e->ring, e->prm.ring_precision, e->activation,
                e->proj_on ? e->proj_ring : nullptr,
                e->proj_on ? reinterpret_cast<const float*>(e->table_blob + e->table_layout.proj_b) : nullptr, e->row_floats};
    if (mask_dev) a.n_streams = e->n_streams;
    PE_HIP(e, launch_clear(a, nullptr));
    PE_HIP(e, hipStreamSynchronize(nullptr));
    return PE_OK;
}

int pe_update_device(pe_engine* e, const int16_t* pcm_dev, int32_t chunk, float* raw_out_dev, void* stream) {
    int rc = check_chunk(e, pcm_dev, chunk);
    if (rc) return rc;
    PE_HIP(e, hipSetDevice(e->device));
    PE_DRAIN(e);
    if (!raw_out_dev) return fail(e, PE_ERR_INVALID, "raw_out_dev is null");
    note_user_stream(e, stream);
    return do_update(e, pcm_dev, chunk, raw_out_dev, nullptr, static_cast<hipStream_t>(stream));
}

// The same update for a caller whose chunks outlive the call (a ring of resident PCM slabs, a capture buffer written ahead):
// the samples left over toward the next frame stay in pcm_dev instead of being copied to the engine's carry, and the next
// call reads the head of its first frame from there.  Any other entry point may follow: it finds the leftovers (one small
// copy launch moves them to the carry first).  Falls back to pe_update_device for chunks that cannot hold a leftover.
int pe_update_device_keep(pe_engine* e, const int16_t* pcm_dev, int32_t chunk, float* raw_out_dev, void* stream) {
    int rc = check_chunk(e, pcm_dev, chunk);
    if (rc) return rc;
    PE_HIP(e, hipSetDevice(e->device));
    PE_DRAIN(e);
    if (!raw_out_dev) return fail(e, PE_ERR_INVALID, "raw_out_dev is null");
    note_user_stream(e, stream);
    return do_update(e, pcm_dev, chunk, raw_out_dev, nullptr, static_cast<hipStream_t>(stream), nullptr, 0, true);
}

int pe_update_vectors_device(pe_engine* e, const int16_t* pcm_dev, int32_t chunk, float* feats_out_dev, void* stream) {
    int rc = check_chunk(e, pcm_dev, chunk);
    if (rc) return rc;
    PE_HIP(e, hipSetDevice(e->device));
    PE_DRAIN(e);
    note_user_stream(e, stream);
    return do_update(e, pcm_dev, chunk, nullptr, feats_out_dev, static_cast<hipStream_t>(stream));
}

int pe_run_device(pe_engine* e, float* raw_out_dev, void* stream) {
    if (!e || !raw_out_dev) return fail(e, PE_ERR_INVALID, "null argument to pe_run_device");
    PE_HIP(e, hipSetDevice(e->device));
    PE_DRAIN(e);
    note_user_stream(e, stream);
    return launch_gru_ring(e, raw_out_dev, static_cast<hipStream_t>(stream));
}

int pe_update(pe_engine* e, const int16_t* pcm_host, int32_t chunk, float* raw_out_host) {
    int rc = check_chunk(e, pcm_host, chunk);
    if (rc) return rc;
    if (!raw_out_host) return fail(e, PE_ERR_INVALID, "raw_out_host is null");
    PE_HIP(e, hipSetDevice(e->device));
    PE_DRAIN(e);
    const size_t pcm_bytes = (size_t)e->n_streams * chunk * sizeof(int16_t);
    if ((rc = ensure(e, e->st_pcm, pcm_bytes))) return rc;
    if ((rc = ensure(e, e->st_out, (size_t)e->n_streams * sizeof(float)))) return rc;
    PE_HIP(e, hipMemcpy(e->st_pcm.p, pcm_host, pcm_bytes, hipMemcpyHostToDevice));
    if ((rc = do_update(e, static_cast<const int16_t*>(e->st_pcm.p), chunk, static_cast<float*>(e->st_out.p), nullptr, nullptr))) return rc;
    PE_HIP(e, hipMemcpy(raw_out_host, e->st_out.p, (size_t)e->n_streams * sizeof(float), hipMemcpyDeviceToHost));
    return PE_OK;
}

// Streams that advance independently (network_runner.py:125-146: every Listener takes chunks at its own pace; one engine
// process per client, runner/precise_runner/runner.py:54-67): the n_active streams named in stream_ids take one chunk each,
// every other stream keeps its leftover samples, counters and feature window untouched and gets no output.
int pe_update_subset_device(pe_engine* e, const int32_t* stream_ids_dev, int32_t n_active, const int16_t* pcm_dev, int32_t chunk,
                            float* raw_out_dev, void* stream) {
    if (!e) return PE_ERR_INVALID;
    if (n_active < 0 || n_active > e->n_streams) return fail(e, PE_ERR_INVALID, "n_active=%d outside 0..%d", n_active, e->n_streams);
    if (n_active == 0) return PE_OK;                 // nobody has audio: nothing moves
    int rc = check_chunk(e, pcm_dev, chunk);
    if (rc) return rc;
    if (!stream_ids_dev || !raw_out_dev) return fail(e, PE_ERR_INVALID, "null argument to pe_update_subset_device");
    PE_HIP(e, hipSetDevice(e->device));
    PE_DRAIN(e);
    note_user_stream(e, stream);
    return do_update(e, pcm_dev, chunk, raw_out_dev, nullptr, static_cast<hipStream_t>(stream), stream_ids_dev, n_active);
}

int pe_update_subset(pe_engine* e, const int32_t* stream_ids_host, int32_t n_active, const int16_t* pcm_host, int32_t chunk, float* raw_out_host) {
    if (!e) return PE_ERR_INVALID;
    if (n_active < 0 || n_active > e->n_streams) return fail(e, PE_ERR_INVALID, "n_active=%d outside 0..%d", n_active, e->n_streams);
    if (n_active == 0) return PE_OK;
    int rc = check_chunk(e, pcm_host, chunk);
    if (rc) return rc;
    if (!stream_ids_host || !raw_out_host) return fail(e, PE_ERR_INVALID, "null argument to pe_update_subset");
    // the host entry point checks what the device one cannot: every id in range, none twice (two rows for one stream in one
    // launch would race on its record)
    e->seen_ids.assign((size_t)e->n_streams, 0);
    for (int i = 0; i < n_active; ++i) {
        const int32_t id = stream_ids_host[i];
        if (id < 0 || id >= e->n_streams) return fail(e, PE_ERR_INVALID, "stream_ids[%d]=%d outside 0..%d", i, id, e->n_streams - 1);
        if (e->seen_ids[(size_t)id]) return fail(e, PE_ERR_INVALID, "stream %d is named twice (stream_ids[%d])", id, i);
        e->seen_ids[(size_t)id] = 1;
    }
    PE_HIP(e, hipSetDevice(e->device));
    PE_DRAIN(e);
    const size_t pcm_bytes = (size_t)n_active * chunk * sizeof(int16_t), out_bytes = (size_t)n_active * sizeof(float), id_bytes = (size_t)n_active * sizeof(int32_t);
    if ((rc = ensure(e, e->st_pcm, pcm_bytes))) return rc;
    if ((rc = ensure(e, e->st_out, out_bytes))) return rc;
    if ((rc = ensure(e, e->st_ids, id_bytes))) return rc;
    PE_HIP(e, hipMemcpy(e->st_ids.p, stream_ids_host, id_bytes, hipMemcpyHostToDevice));
    PE_HIP(e, hipMemcpy(e->st_pcm.p, pcm_host, pcm_bytes, hipMemcpyHostToDevice));
    if ((rc = do_update(e, static_cast<const int16_t*>(e->st_pcm.p), chunk, static_cast<float*>(e->st_out.p), nullptr, nullptr,
                        static_cast<const int32_t*>(e->st_ids.p), n_active))) return rc;
    PE_HIP(e, hipMemcpy(raw_out_host, e->st_out.p, out_bytes, hipMemcpyDeviceToHost));
    return PE_OK;
}

int pe_host_alloc(pe_engine* e, size_t bytes, void** out) {
    if (!e || !out || bytes == 0) return fail(e, PE_ERR_INVALID, "bad arguments to pe_host_alloc");
    PE_HIP(e, hipSetDevice(e->device));
    void* p = nullptr;
    hipError_t err = hipHostMalloc(&p, bytes, hipHostMallocDefault);
    if (err != hipSuccess) return fail(e, PE_ERR_NOMEM, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(err));
    e->pinned.emplace_back(static_cast<char*>(p), bytes);
    *out = p;
    return PE_OK;
}

int pe_host_free(pe_engine* e, void* p) {
    if (!e || !p) return PE_ERR_INVALID;
    PE_HIP(e, hipSetDevice(e->device));
    PE_DRAIN(e);                                      // an update in flight may still read from / write to it
    for (auto it = e->pinned.begin(); it != e->pinned.end(); ++it)
        if (it->first == static_cast<char*>(p)) {
            e->pinned.erase(it);
            PE_HIP(e, hipHostFree(p));
            return PE_OK;
        }
    return fail(e, PE_ERR_INVALID, "pe_host_free: not a pe_host_alloc'ed pointer of this engine");
}

int pe_update_async(pe_engine* e, const int16_t* pcm_host, int32_t chunk, float* raw_out_host) {
    int rc = check_chunk(e, pcm_host, chunk);
    if (rc) return rc;
    if (!raw_out_host) return fail(e, PE_ERR_INVALID, "raw_out_host is null");
    PE_HIP(e, hipSetDevice(e->device));
    if ((rc = async_init(e))) return rc;
    pe_engine::AsyncSlot& sl = e->aslot[e->async_next % pe_engine::kAsyncDepth];
    if ((rc = finish_slot(e, sl))) return rc;          // the ring is full: the oldest update is delivered first
    const size_t pcm_bytes = (size_t)e->n_streams * chunk * sizeof(int16_t), out_bytes = (size_t)e->n_streams * sizeof(float);
    if ((rc = ensure(e, sl.dev_in, pcm_bytes))) return rc;
    if ((rc = ensure(e, sl.dev_out, out_bytes))) return rc;
    // (pageable memory: hipMemcpyAsync stages it through the runtime's own pinned buffers and returns once the last piece is
    //  staged -- the caller gets its buffer back at the return of this call; measured: 207 us per 8.4 MB update that
    //  way against 295 us with a memcpy into a pinned ring of the engine's own, 165 us zero-copy from pe_host_alloc'ed memory.
    //  Memory the runtime knows as pinned -- pe_host_alloc, but also hipHostRegister / torch pin_memory -- is read by the DMA
    //  AFTER this call returns: such a buffer must stay untouched until the update is delivered)
    sl.direct_out = is_pinned(e, raw_out_host, out_bytes);
    if (!sl.direct_out) {
        void* po = sl.pin_out;
        if ((rc = ensure_pinned(e, &po, &sl.pin_out_bytes, out_bytes))) return rc;
        sl.pin_out = static_cast<float*>(po);
    }
    // state-moving work the caller queued through a *_device entry point (on the NULL stream or its own) comes first
    if (e->user_dirty) {
        PE_HIP(e, hipEventRecord(e->ev_user, e->last_user_stream));
        PE_HIP(e, hipStreamWaitEvent(e->s_compute, e->ev_user, 0));
        e->user_dirty = false;
    }
    // From the first enqueue on, a failure must not leave half an update behind: both streams are drained, the slot stays
    // free, and the error goes to the caller (the streams' state may then be one update ahead of what was delivered --
    // the message says so; pe_clear restarts the streams).
    auto enqueue = [&]() -> int {
        // this slot's device chunk was the `head` of the update AFTER the one the slot last carried (the leftovers stay in the
        // chunks: do_update(..., keep)): that update must be through before the copy overwrites it
        pe_engine::AsyncSlot& reader = e->aslot[(e->async_next + 1) % pe_engine::kAsyncDepth];
        if (reader.busy) PE_HIP(e, hipStreamWaitEvent(e->s_copy, reader.done, 0));
        PE_HIP(e, hipMemcpyAsync(sl.dev_in.p, pcm_host, pcm_bytes, hipMemcpyHostToDevice, e->s_copy));
        PE_HIP(e, hipEventRecord(sl.copied, e->s_copy));
        PE_HIP(e, hipStreamWaitEvent(e->s_compute, sl.copied, 0));
        int urc = do_update(e, static_cast<const int16_t*>(sl.dev_in.p), chunk, static_cast<float*>(sl.dev_out.p), nullptr, e->s_compute, nullptr, 0, true);
        if (urc) return urc;
        PE_HIP(e, hipMemcpyAsync(sl.direct_out ? raw_out_host : sl.pin_out, sl.dev_out.p, out_bytes, hipMemcpyDeviceToHost, e->s_compute));
        PE_HIP(e, hipEventRecord(sl.done, e->s_compute));
        return PE_OK;
    };
    if ((rc = enqueue())) {
        const std::string why = e->err;
        (void)hipStreamSynchronize(e->s_copy);
        (void)hipStreamSynchronize(e->s_compute);
        return fail(e, rc, "pe_update_async: %s (this update was not delivered; the engine's streams were drained)", why.c_str());
    }
    sl.user_out = raw_out_host; sl.out_bytes = out_bytes; sl.busy = true;
    ++e->async_inflight;
    ++e->async_next;
    return PE_OK;
}

int pe_wait(pe_engine* e) {
    if (!e) return PE_ERR_INVALID;
    PE_HIP(e, hipSetDevice(e->device));
    return drain_async(e);
}

int pe_update_vectors(pe_engine* e, const int16_t* pcm_host, int32_t chunk, float* feats_out_host) {
    int rc = check_chunk(e, pcm_host, chunk);
    if (rc) return rc;
    PE_HIP(e, hipSetDevice(e->device));
    PE_DRAIN(e);
    const size_t pcm_bytes = (size_t)e->n_streams * chunk * sizeof(int16_t);
    const size_t feat_bytes = (size_t)e->n_streams * e->prm.n_features * e->prm.n_mfcc * sizeof(float);
    if ((rc = ensure(e, e->st_pcm, pcm_bytes))) return rc;
    if (feats_out_host && (rc = ensure(e, e->st_feats, feat_bytes))) return rc;
    PE_HIP(e, hipMemcpy(e->st_pcm.p, pcm_host, pcm_bytes, hipMemcpyHostToDevice));
    if ((rc = do_update(e, static_cast<const int16_t*>(e->st_pcm.p), chunk, nullptr,
                        feats_out_host ? static_cast<float*>(e->st_feats.p) : nullptr, nullptr))) return rc;
    if (feats_out_host) PE_HIP(e, hipMemcpy(feats_out_host, e->st_feats.p, feat_bytes, hipMemcpyDeviceToHost));
    else PE_HIP(e, hipStreamSynchronize(nullptr));
    return PE_OK;
}

int pe_get_vectors(pe_engine* e, float* feats_out_host) {
    if (!e || !feats_out_host) return fail(e, PE_ERR_INVALID, "null argument to pe_get_vectors");
    PE_HIP(e, hipSetDevice(e->device));
    PE_DRAIN(e);
    int rc;
    const size_t feat_bytes = (size_t)e->n_streams * e->prm.n_features * e->prm.n_mfcc * sizeof(float);
    if ((rc = ensure(e, e->st_feats, feat_bytes))) return rc;
    GatherArgs g{e->n_streams, e->prm.n_features, e->prm.n_mfcc, e->ring_slots, e->prm.ring_precision, e->ring, state_of(e, e->call_no + 1u), static_cast<float*>(e->st_feats.p), e->row_floats};
    PE_HIP(e, launch_gather(g, nullptr));
    PE_HIP(e, hipMemcpy(feats_out_host, e->st_feats.p, feat_bytes, hipMemcpyDeviceToHost));
    return PE_OK;
}

int pe_set_vectors(pe_engine* e, const float* feats_host) {
    if (!e || !feats_host) return fail(e, PE_ERR_INVALID, "null argument to pe_set_vectors");
    PE_HIP(e, hipSetDevice(e->device));
    PE_DRAIN(e);
    int rc;
    if ((rc = pe_clear(e, nullptr))) return rc;
    const size_t feat_bytes = (size_t)e->n_streams * e->prm.n_features * e->prm.n_mfcc * sizeof(float);
    if ((rc = ensure(e, e->st_feats, feat_bytes))) return rc;
    PE_HIP(e, hipMemcpy(e->st_feats.p, feats_host, feat_bytes, hipMemcpyHostToDevice));
    uint32_t call = 0;
    if ((rc = begin_state_call(e, nullptr, &call))) return rc;
    GatherArgs g{e->n_streams, e->prm.n_features, e->prm.n_mfcc, e->ring_slots, e->prm.ring_precision, e->ring, state_of(e, call), static_cast<float*>(e->st_feats.p), e->row_floats};
    PE_HIP(e, launch_scatter(g, nullptr));
    if (e->proj_on)
        PE_HIP(e, launch_project_rows(e->ring, e->proj_ring, reinterpret_cast<const float*>(e->table_blob + e->table_layout.proj_w),
                                      reinterpret_cast<const float*>(e->table_blob + e->table_layout.proj_b), e->prm.n_mfcc,
                                      (long long)e->n_tiles * e->ring_slots * kTileStreams, nullptr));
    PE_HIP(e, hipStreamSynchronize(nullptr));
    return PE_OK;
}

int pe_predict_device(pe_engine* e, const float* feats_dev, int32_t n, float* out_dev, void* stream) {
    if (!e) return PE_ERR_INVALID;
    if (n < 0 || (n > 0 && (!feats_dev || !out_dev))) return fail(e, PE_ERR_INVALID, "bad arguments to pe_predict_device");
    if (n == 0) return PE_OK;
    PE_HIP(e, hipSetDevice(e->device));
    GruArgs a = gru_args(e);
    a.n_streams = n;
    a.waves_per_tile = 1;
    a.feats = feats_dev;
    a.out = out_dev;
    { int nrc = launch_network(e, a, 0, static_cast<hipStream_t>(stream)); if (nrc) return nrc; }
    return PE_OK;
}

int pe_predict(pe_engine* e, const float* feats_host, int32_t n, float* out_host) {
    if (!e) return PE_ERR_INVALID;
    if (n < 0 || (n > 0 && (!feats_host || !out_host))) return fail(e, PE_ERR_INVALID, "bad arguments to pe_predict");
    if (n == 0) return PE_OK;
    PE_HIP(e, hipSetDevice(e->device));
    int rc;
    const size_t fb = (size_t)n * e->prm.n_features * (e->prm.use_delta ? 2 * e->n_in : e->n_in) * sizeof(float);
    if ((rc = ensure(e, e->st_feats, fb))) return rc;
    if ((rc = ensure(e, e->st_out, (size_t)n * sizeof(float)))) return rc;
    PE_HIP(e, hipMemcpy(e->st_feats.p, feats_host, fb, hipMemcpyHostToDevice));
    if ((rc = pe_predict_device(e, static_cast<const float*>(e->st_feats.p), n, static_cast<float*>(e->st_out.p), nullptr))) return rc;
    PE_HIP(e, hipMemcpy(out_host, e->st_out.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return PE_OK;
}

namespace {
int vectorize_buffer(pe_engine* e, const double* audio_host, int64_t n_samples, double* feats_out_host,
                     int64_t max_frames, int64_t* n_frames_out, bool mels) {
    if (!e || !n_frames_out) return fail(e, PE_ERR_INVALID, "null argument to pe_vectorize_*");
    if (n_samples < 0 || (n_samples > 0 && !audio_host)) return fail(e, PE_ERR_INVALID, "bad audio buffer");
    const int64_t n_frames = frames_of_buffer(e->prm, n_samples);
    *n_frames_out = n_frames;
    if (n_frames == 0) return PE_OK;
    if (!feats_out_host || max_frames < n_frames) return fail(e, PE_ERR_INVALID, "output holds %lld frames, need %lld", (long long)max_frames, (long long)n_frames);
    PE_HIP(e, hipSetDevice(e->device));
    int rc;
    const int width = mels ? e->prm.n_filt : e->prm.n_mfcc;
    const size_t ab = (size_t)n_samples * sizeof(double), fb = (size_t)n_frames * width * sizeof(double);
    if ((rc = ensure(e, e->st_audio, ab))) return rc;
    if ((rc = ensure(e, e->st_mfcc, fb))) return rc;
    PE_HIP(e, hipMemcpy(e->st_audio.p, audio_host, ab, hipMemcpyHostToDevice));
    double* dev_out = static_cast<double*>(e->st_mfcc.p);
    if (e->general) {
        if (e->prm.mfcc_precision == 0) {
            GeneralOfflineArgs<double> a{geom(e), e->gtab, static_cast<const double*>(e->st_audio.p), n_frames, mels ? nullptr : dev_out, nullptr, mels ? dev_out : nullptr, e->row_floats};
            PE_HIP(e, launch_general_offline_f64(a, e->n_cus, nullptr));
        } else {
            GeneralOfflineArgs<float> a{geom(e), e->gtab, static_cast<const double*>(e->st_audio.p), n_frames, mels ? nullptr : dev_out, nullptr, mels ? dev_out : nullptr, e->row_floats};
            PE_HIP(e, launch_general_offline_f32(a, e->n_cus, nullptr));
        }
    } else if (e->prm.mfcc_precision == 0) {
        MfccOfflineArgs<double> a{geom(e), static_cast<const double*>(e->st_audio.p), n_samples, n_frames,
                                  mels ? nullptr : dev_out, nullptr, mels ? dev_out : nullptr};
        PE_HIP(e, launch_mfcc_offline_f64(a, tables<double>(e), e->n_cus, nullptr));
    } else {
        MfccOfflineArgs<float> a{geom(e), static_cast<const double*>(e->st_audio.p), n_samples, n_frames,
                                 mels ? nullptr : dev_out, nullptr, mels ? dev_out : nullptr};
        PE_HIP(e, launch_mfcc_offline_f32(a, tables<float>(e), e->n_cus, nullptr));
    }
    PE_HIP(e, hipMemcpy(feats_out_host, e->st_mfcc.p, fb, hipMemcpyDeviceToHost));
    return PE_OK;
}
}  // namespace

int pe_vectorize_raw(pe_engine* e, const double* audio_host, int64_t n_samples, double* feats_out_host,
                     int64_t max_frames, int64_t* n_frames_out) {
    return vectorize_buffer(e, audio_host, n_samples, feats_out_host, max_frames, n_frames_out, false);
}

int pe_vectorize_mels(pe_engine* e, const double* audio_host, int64_t n_samples, double* mels_out_host,
                      int64_t max_frames, int64_t* n_frames_out) {
    return vectorize_buffer(e, audio_host, n_samples, mels_out_host, max_frames, n_frames_out, true);
}

int pe_evaluate(pe_engine* e, const double* audio_host, int64_t n_samples, int32_t hop_frames, float* out_host,
                int64_t max_windows, int64_t* n_windows_out) {
    if (!e || !n_windows_out) return fail(e, PE_ERR_INVALID, "null argument to pe_evaluate");
    if (hop_frames < 1) return fail(e, PE_ERR_INVALID, "hop_frames must be >= 1 (chunk_size // hop_samples)");
    if (n_samples < 0 || (n_samples > 0 && !audio_host)) return fail(e, PE_ERR_INVALID, "bad audio buffer");
    const int64_t T = e->prm.n_features;
    const int64_t n_frames = frames_of_buffer(e->prm, n_samples);
    // simulate.py:96-99: windows end at frame i for i in range(T, n_frames, hop_frames)
    const int64_t n_windows = n_frames > T ? (n_frames - T + hop_frames - 1) / hop_frames : 0;
    *n_windows_out = n_windows;
    if (n_windows == 0) return PE_OK;
    if (!out_host || max_windows < n_windows) return fail(e, PE_ERR_INVALID, "output holds %lld windows, need %lld", (long long)max_windows, (long long)n_windows);
    if (n_windows > 0x7fffffff) return fail(e, PE_ERR_INVALID, "too many windows");
    PE_HIP(e, hipSetDevice(e->device));
    int rc;
    const size_t ab = (size_t)n_samples * sizeof(double), rb = (size_t)n_frames * e->row_floats * sizeof(float);
    if ((rc = ensure(e, e->st_audio, ab))) return rc;
    if ((rc = ensure(e, e->st_feats, rb))) return rc;
    if ((rc = ensure(e, e->st_out, (size_t)n_windows * sizeof(float)))) return rc;
    PE_HIP(e, hipMemcpy(e->st_audio.p, audio_host, ab, hipMemcpyHostToDevice));
    if (e->general) {
        if (e->prm.mfcc_precision == 0) {
            GeneralOfflineArgs<double> a{geom(e), e->gtab, static_cast<const double*>(e->st_audio.p), n_frames, nullptr, static_cast<float*>(e->st_feats.p), nullptr, e->row_floats};
            PE_HIP(e, launch_general_offline_f64(a, e->n_cus, nullptr));
        } else {
            GeneralOfflineArgs<float> a{geom(e), e->gtab, static_cast<const double*>(e->st_audio.p), n_frames, nullptr, static_cast<float*>(e->st_feats.p), nullptr, e->row_floats};
            PE_HIP(e, launch_general_offline_f32(a, e->n_cus, nullptr));
        }
    } else if (e->prm.mfcc_precision == 0) {
        MfccOfflineArgs<double> a{geom(e), static_cast<const double*>(e->st_audio.p), n_samples, n_frames, nullptr, static_cast<float*>(e->st_feats.p), nullptr};
        PE_HIP(e, launch_mfcc_offline_f64(a, tables<double>(e), e->n_cus, nullptr));
    } else {
        MfccOfflineArgs<float> a{geom(e), static_cast<const double*>(e->st_audio.p), n_samples, n_frames, nullptr, static_cast<float*>(e->st_feats.p), nullptr};
        PE_HIP(e, launch_mfcc_offline_f32(a, tables<float>(e), e->n_cus, nullptr));
    }
    GruArgs g = gru_args(e);
    g.n_streams = (int)n_windows;
    g.feats = static_cast<const float*>(e->st_feats.p);
    g.row_stride = hop_frames;
    g.out = static_cast<float*>(e->st_out.p);
    g.waves_per_tile = 1;
    if ((rc = launch_network(e, g, 2, nullptr))) return rc;
    PE_HIP(e, hipMemcpy(out_host, e->st_out.p, (size_t)n_windows * sizeof(float), hipMemcpyDeviceToHost));
    return PE_OK;
}

int pe_set_decoder(pe_engine* e, const double* cd, int32_t cd_len, int32_t min_out, int32_t out_range, double center) {
    if (!e || cd_len < 0 || (cd_len > 0 && !cd)) return fail(e, PE_ERR_INVALID, "bad decoder table");
    if (out_range != 0 && cd_len < 1) return fail(e, PE_ERR_INVALID, "decoder needs a non-empty table when out_range != 0");
    PE_HIP(e, hipSetDevice(e->device));
    PE_HIP(e, hipDeviceSynchronize());                    // no decode launch may still read the old table
    dev_free(e, e->cd, (size_t)(e->cd_len ? e->cd_len : 1) * sizeof(double));
    e->cd = nullptr; e->cd_len = 0;
    std::vector<double> host(cd, cd + cd_len);
    int rc = dev_upload(e, &e->cd, host);
    if (rc) return rc;
    e->cd_len = cd_len; e->dec_min_out = min_out; e->dec_out_range = out_range; e->dec_center = center;
    return PE_OK;
}

int pe_set_trigger(pe_engine* e, int32_t chunk_size_bytes, double sensitivity, int32_t trigger_level) {
    if (!e || chunk_size_bytes <= 0) return fail(e, PE_ERR_INVALID, "bad trigger parameters");
    PE_HIP(e, hipSetDevice(e->device));
    if (!e->activation) {
        int rc = dev_alloc(e, &e->activation, (size_t)e->n_padded);
        if (rc) return rc;
    }
    PE_HIP(e, hipMemset(e->activation, 0, (size_t)e->n_padded * sizeof(int32_t)));
    e->trig_threshold = 1.0 - sensitivity;
    e->trig_level = trigger_level;
    const int n = 8 * 2048;                                // -(8 * 2048) // chunk_size, floor division
    e->trig_rearm = -((n + chunk_size_bytes - 1) / chunk_size_bytes);
    e->trig_on = true;
    return PE_OK;
}

int pe_decode_device(pe_engine* e, const float* raw_dev, double* conf_out_dev, unsigned char* fired_out_dev, void* stream) {
    if (!e || !raw_dev) return fail(e, PE_ERR_INVALID, "null argument to pe_decode_device");
    if (!e->cd && e->dec_out_range != 0) return fail(e, PE_ERR_INVALID, "pe_set_decoder has not been called");
    if (!e->cd_len && !e->cd) return fail(e, PE_ERR_INVALID, "pe_set_decoder has not been called");
    PE_HIP(e, hipSetDevice(e->device));
    DecodeArgs a{};
    a.n_streams = e->n_streams; a.raw = raw_dev; a.cd = e->cd; a.cd_len = e->cd_len;
    a.min_out = e->dec_min_out; a.out_range = e->dec_out_range; a.center = e->dec_center;
    a.conf_out = conf_out_dev;
    a.activation = e->trig_on ? e->activation : nullptr;
    a.fired_out = fired_out_dev;
    a.threshold = e->trig_threshold; a.trigger_level = e->trig_level; a.rearm = e->trig_rearm;
    PE_HIP(e, launch_decode(a, static_cast<hipStream_t>(stream)));
    return PE_OK;
}

int pe_decode(pe_engine* e, const float* raw_host, double* conf_out_host, unsigned char* fired_out_host) {
    if (!e || !raw_host) return fail(e, PE_ERR_INVALID, "null argument to pe_decode");
    PE_HIP(e, hipSetDevice(e->device));
    int rc;
    const size_t n = (size_t)e->n_streams;
    if ((rc = ensure(e, e->st_out, n * sizeof(float)))) return rc;
    if ((rc = ensure(e, e->st_conf, n * sizeof(double)))) return rc;
    if ((rc = ensure(e, e->st_fired, n))) return rc;
    PE_HIP(e, hipMemcpy(e->st_out.p, raw_host, n * sizeof(float), hipMemcpyHostToDevice));
    PE_HIP(e, hipMemset(e->st_fired.p, 0, n));
    if ((rc = pe_decode_device(e, static_cast<const float*>(e->st_out.p), static_cast<double*>(e->st_conf.p),
                               static_cast<unsigned char*>(e->st_fired.p), nullptr))) return rc;
    if (conf_out_host) PE_HIP(e, hipMemcpy(conf_out_host, e->st_conf.p, n * sizeof(double), hipMemcpyDeviceToHost));
    if (fired_out_host) PE_HIP(e, hipMemcpy(fired_out_host, e->st_fired.p, n, hipMemcpyDeviceToHost));
    if (!conf_out_host && !fired_out_host) PE_HIP(e, hipStreamSynchronize(nullptr));
    return PE_OK;
}

int pe_reserve_updates(pe_engine* e, int32_t max_updates, int32_t max_chunk_samples) {
    if (!e || max_updates < 1 || max_chunk_samples < 1) return fail(e, PE_ERR_INVALID, "bad arguments to pe_reserve_updates");
    PE_HIP(e, hipSetDevice(e->device));
    PE_DRAIN(e);
    PE_HIP(e, hipDeviceSynchronize());
    const int pending = pending_frames(e->prm);
    const long long frames = ((long long)max_updates * max_chunk_samples + e->prm.hop_samples - 1) / e->prm.hop_samples + 1;
    const int slots = next_pow2((int)(e->prm.n_features + pending + frames));
    if (slots != e->ring_slots) {
        dev_free(e, e->ring, ring_floats(e) * sizeof(float));
        e->ring = nullptr;
        dev_free(e, e->proj_ring, (size_t)e->n_tiles * e->ring_slots * kTileStreams * kProjRow * sizeof(float));
        e->proj_ring = nullptr;
        e->ring_slots = slots;
        int rc = dev_alloc(e, &e->ring, ring_floats(e));
        if (rc) return rc;
        if (e->proj_on && (rc = dev_alloc(e, &e->proj_ring, (size_t)e->n_tiles * e->ring_slots * kTileStreams * kProjRow))) return rc;
    }
    if (!e->ke_hist || max_updates > e->max_updates) {
        if (e->ke_hist) {
            dev_free(e, e->ke_hist, (size_t)e->max_updates * e->n_padded * sizeof(uint32_t));
            e->ke_hist = nullptr;
        }
        int rc = dev_alloc(e, &e->ke_hist, (size_t)max_updates * e->n_padded);
        if (rc) return rc;
    }
    e->max_updates = max_updates;
    return pe_clear(e, nullptr);          // the ring was re-laid out: streams restart
}

int pe_update_many_device(pe_engine* e, const int16_t* pcm_dev, int32_t chunk, int32_t n_updates, float* raw_out_dev, void* stream) {
    int rc = check_chunk(e, pcm_dev, chunk);
    if (rc) return rc;
    if (!raw_out_dev || n_updates < 1) return fail(e, PE_ERR_INVALID, "bad arguments to pe_update_many_device");
    if (n_updates > e->max_updates || !e->ke_hist) return fail(e, PE_ERR_INVALID, "call pe_reserve_updates(e, >= %d, >= %d) first", n_updates, chunk);
    PE_HIP(e, hipSetDevice(e->device));
    PE_DRAIN(e);
    const int flen = frame_len_of(e->prm);
    const int pending = pending_frames(e->prm);
    const long long frames = ((long long)n_updates * chunk + e->prm.hop_samples - 1) / e->prm.hop_samples + 1;
    if ((long long)n_updates * chunk >= (1ll << 30)) return fail(e, PE_ERR_INVALID, "n_updates * chunk_samples must stay below 2^30");
    if (e->prm.n_features + pending + frames > e->ring_slots) return fail(e, PE_ERR_INVALID, "reserved ring too small for %d updates of %d samples", n_updates, chunk);
    hipStream_t s = static_cast<hipStream_t>(stream);
    note_user_stream(e, stream);
    if ((rc = flush_kept(e, s))) return rc;
    uint32_t call = 0;
    if ((rc = begin_state_call(e, s, &call))) return rc;
    if (e->general) {
        // the general front end: every frame the call completes in ONE launch (mfcc_general_device.h: general_stream with n_updates),
        // then the batched network launch over 16-float rows, or one network launch per update over 32-float rows
        if (e->prm.mfcc_precision == 0) {
            GeneralStreamArgs<double> a = general_args<double>(e, pcm_dev, chunk, call);
            a.n_updates = n_updates; a.ke_hist = e->ke_hist;
            PE_HIP(e, launch_general_stream_f64(a, s));
        } else {
            GeneralStreamArgs<float> a = general_args<float>(e, pcm_dev, chunk, call);
            a.n_updates = n_updates; a.ke_hist = e->ke_hist;
            PE_HIP(e, launch_general_stream_f32(a, s));
        }
        GruArgs g = gru_args(e);
        g.ke_plain = e->ke_hist;
        g.out = raw_out_dev;
        if (e->wide || e->row_floats != kRowFloats) {
            for (int u = 0; u < n_updates; ++u) {
                GruArgs gu = g;
                gu.ke_plain = e->ke_hist + (size_t)u * e->n_padded;
                gu.out = raw_out_dev + (size_t)u * e->n_streams;
                int nrc = launch_network(e, gu, 1, s);
                if (nrc) return nrc;
            }
            return PE_OK;
        }
        g.waves_per_tile = 1;
        PE_HIP(e, launch_gru_many(g, n_updates, e->n_padded, s));
        return PE_OK;
    }
    // frames one stream can complete in this call: one task row per frame, at most kMaxFrameRows rows
    const int rows = max_frames_per_call(e, (long long)n_updates * chunk);
    if (rows > kMaxFrameRows) return fail(e, PE_ERR_INVALID, "a call may complete at most %d frames per stream (%d updates of %d samples: %d)", kMaxFrameRows, n_updates, chunk, rows);
    (void)flen;
    if (e->prm.mfcc_precision == 0) {
        MfccStreamArgs<double> a = mfcc_args<double>(e, pcm_dev, chunk, call);
        a.n_updates = n_updates; a.ke_hist = e->ke_hist; a.n_frame_rows = rows;
        PE_HIP(e, launch_mfcc_f64(a, tables<double>(e), e->n_cus, s));
    } else {
        MfccStreamArgs<float> a = mfcc_args<float>(e, pcm_dev, chunk, call);
        a.n_updates = n_updates; a.ke_hist = e->ke_hist; a.n_frame_rows = rows;
        PE_HIP(e, launch_mfcc_f32(a, tables<float>(e), e->n_cus, s));
    }
    GruArgs g = gru_args(e);
    g.ke_plain = e->ke_hist;
    g.out = raw_out_dev;
    if (e->wide) {
        // the streamed-weight network: one launch per update of the call (each fills the machine on its own), the
        // window of update u found through its row of the emitted-frame history
        for (int u = 0; u < n_updates; ++u) {
            GruArgs gu = g;
            gu.ke_plain = e->ke_hist + (size_t)u * e->n_padded;
            gu.out = raw_out_dev + (size_t)u * e->n_streams;
            int nrc = launch_network(e, gu, 1, s);
            if (nrc) return nrc;
        }
        return PE_OK;
    }
    g.waves_per_tile = 1;                                 // (the launcher picks one or four waves per window itself)
    PE_HIP(e, launch_gru_many(g, n_updates, e->n_padded, s));
    return PE_OK;
}

int pe_update_many(pe_engine* e, const int16_t* pcm_host, int32_t chunk, int32_t n_updates, float* raw_out_host) {
    int rc = check_chunk(e, pcm_host, chunk);
    if (rc) return rc;
    if (!raw_out_host || n_updates < 1) return fail(e, PE_ERR_INVALID, "bad arguments to pe_update_many");
    PE_HIP(e, hipSetDevice(e->device));
    const size_t pcm_bytes = (size_t)n_updates * e->n_streams * chunk * sizeof(int16_t);
    const size_t out_bytes = (size_t)n_updates * e->n_streams * sizeof(float);
    if ((rc = ensure(e, e->st_pcm, pcm_bytes))) return rc;
    if ((rc = ensure(e, e->st_out, out_bytes))) return rc;
    PE_HIP(e, hipMemcpy(e->st_pcm.p, pcm_host, pcm_bytes, hipMemcpyHostToDevice));
    if ((rc = pe_update_many_device(e, static_cast<const int16_t*>(e->st_pcm.p), chunk, n_updates, static_cast<float*>(e->st_out.p), nullptr))) return rc;
    PE_HIP(e, hipMemcpy(raw_out_host, e->st_out.p, out_bytes, hipMemcpyDeviceToHost));
    return PE_OK;
}

int pe_get_info(const pe_engine* e, pe_info* out) {
    if (!e || !out) return PE_ERR_INVALID;
    out->n_streams = e->n_streams;
    out->n_features = e->prm.n_features;
    out->n_mfcc = e->prm.n_mfcc;
    out->units = e->units;
    out->n_layers = e->n_layers;
    out->ring_slots = e->ring_slots;
    out->carry_capacity = e->carry_cap;
    out->mfcc_precision = e->prm.mfcc_precision;
    out->gru_precision = e->prm.gru_precision;
    out->device_bytes = e->device_bytes;
    return PE_OK;
}

int pe_get_stream_state(pe_engine* e, int32_t* q_out, uint32_t* computed_out, uint32_t* emitted_out) {
    if (!e) return PE_ERR_INVALID;
    PE_HIP(e, hipSetDevice(e->device));
    PE_DRAIN(e);
    PE_HIP(e, hipDeviceSynchronize());
    // both sides of every record; the current one is picked here exactly as the kernels pick it (pe_common.h: rec_side)
    std::vector<StreamRec> host((size_t)2 * e->n_padded);
    PE_HIP(e, hipMemcpy(host.data(), e->rec, host.size() * sizeof(StreamRec), hipMemcpyDeviceToHost));
    const uint32_t call = e->call_no + 1u;
    for (int s = 0; s < e->n_streams; ++s) {
        const StreamRec& r0 = host[rec_at((uint32_t)e->n_padded, s, 0)];
        const StreamRec& r1 = host[rec_at((uint32_t)e->n_padded, s, 1)];
        const StreamRec& c = (call - r1.wcall) - 1u < (call - r0.wcall) - 1u ? r1 : r0;
        if (q_out) q_out[s] = c.q;
        if (computed_out) computed_out[s] = c.kc;
        if (emitted_out) emitted_out[s] = c.ke;
    }
    return PE_OK;
}

int pe_set_renumber_at(pe_engine* e, uint32_t call_number) {
    if (!e) return PE_ERR_INVALID;
    if (call_number < 8u || call_number > 0x7fff0000u) return fail(e, PE_ERR_INVALID, "renumbering threshold must be in 8..0x7fff0000");
    e->renumber_at = call_number;
    return PE_OK;
}

int pe_set_fused(pe_engine* e, int32_t enabled) {
    if (!e) return PE_ERR_INVALID;
    PE_DRAIN(e);
    e->fused = enabled != 0;
    return PE_OK;
}

int pe_set_input_projection(pe_engine* e, int32_t enabled) {
    if (!e) return PE_ERR_INVALID;
    if (enabled != 0 && enabled != 1) return fail(e, PE_ERR_INVALID, "input projection must be 0 or 1");
    if (enabled && !e->proj_ok) return fail(e, PE_ERR_UNSUPPORTED, "input-projection rows exist for the float32 network of 17..20 units without delta features only");
    if ((enabled != 0) == e->proj_on) return PE_OK;
    PE_HIP(e, hipSetDevice(e->device));
    PE_DRAIN(e);
    PE_HIP(e, hipDeviceSynchronize());
    e->proj_on = enabled != 0;
    if (e->proj_on && !e->proj_ring) {
        int rc = dev_alloc(e, &e->proj_ring, (size_t)e->n_tiles * e->ring_slots * kTileStreams * kProjRow);
        if (rc) { e->proj_on = false; return rc; }
    }
    return pe_clear(e, nullptr);          // rows written so far lack (or no longer need) their projections: streams restart
}

int pe_set_gru_waves(pe_engine* e, int32_t waves) {
    if (!e) return PE_ERR_INVALID;
    if (waves != 0 && waves != 1 && waves != 4) return fail(e, PE_ERR_INVALID, "gru kernel shape must be 0 (auto), 1 or 4 (waves per tile)");
    PE_DRAIN(e);
    e->gru_waves = waves;
    return PE_OK;
}

int pe_set_gru_tiling(pe_engine* e, int32_t tiling) {
    if (!e) return PE_ERR_INVALID;
    if (tiling < -1 || tiling > 2) return fail(e, PE_ERR_INVALID, "gru tiling must be -1 (auto), 0 (classic), 1 (re-tiled) or 2 (XDL form)");
    if (e->wide) {
        if (tiling == 1) return fail(e, PE_ERR_UNSUPPORTED, "wide networks have two forms: 0 (f32-input MFMAs) and 2 (float32 products on the bf16 pipe)");
        PE_DRAIN(e);
        e->gru_tiling = tiling;
        return PE_OK;
    }
    if (tiling == 2 && !e->x3_blob)
        return fail(e, PE_ERR_UNSUPPORTED, "the XDL form of the float32 network (tiling 2) takes <= 20 units, <= 15 inputs, float32 operands, no use_delta");
    if (tiling == 1 && e->prm.gru_precision == 1 && !e->b20_blob)
        return fail(e, PE_ERR_UNSUPPORTED, "the five-values layout of the bf16 network (tiling 1) takes <= 20 units and <= 14 features");
    PE_DRAIN(e);
    e->gru_tiling = tiling;
    return PE_OK;
}

int pe_get_gru_tiling(const pe_engine* e) {
    if (!e) return -2;               // (not PE_ERR_INVALID = 1, which is a valid answer)
    if (e->wide) return wide_uses_x3(e) ? 2 : 0;
    if (e->prm.gru_precision != 0) return gru_args(e).b20 ? 1 : 0;
    const GruArgs a = gru_args(e);
    return a.x3 ? 2 : a.cw ? 1 : 0;
}

int pe_set_timing(pe_engine* e, int32_t enabled) {
    if (!e) return PE_ERR_INVALID;
    e->timing = enabled != 0;
    e->ev_valid = false;
    return PE_OK;
}

int pe_get_last_timing(pe_engine* e, float* mfcc_ms, float* gru_ms) {
    if (!e) return PE_ERR_INVALID;
    if (!e->ev_valid) return fail(e, PE_ERR_INVALID, "no timed update recorded (call pe_set_timing(e, 1) first)");
    PE_HIP(e, hipEventSynchronize(e->ev[2]));
    float a = 0.f, b = 0.f;
    PE_HIP(e, hipEventElapsedTime(&a, e->ev[0], e->ev[1]));
    PE_HIP(e, hipEventElapsedTime(&b, e->ev[1], e->ev[2]));
    if (mfcc_ms) *mfcc_ms = a;      // fused launch: the whole update; else the MFCC kernel
    if (gru_ms) *gru_ms = e->ev_has_gru ? b : 0.f;
    return PE_OK;
}

}  // extern "C"
