// Shared declarations between the C-ABI host code (engine.hip) and the gfx950 kernels.
// Everything here is MI355X-only: 64-lane wavefronts, 16x16x4 f32 MFMA tiles, one 512-point FFT per
// wave.  No portability layer on purpose.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mfcc_wave_tables.h"

namespace pe {

constexpr int kTileStreams = 16;    // streams per GRU tile == MFMA N dimension
constexpr int kRowFloats = 16;      // floats per feature-ring row (n_mfcc <= 16, rest zero)
constexpr int kCarryCap = 512;      // int16 samples of leftover PCM kept per stream (< n_fft)
constexpr int kNfft = 512;          // the only FFT size with a kernel
constexpr int kBins = kNfft / 2 + 1;
constexpr int kMaxFilt = 64;
constexpr int kCwSlots = 32;        // ring slots the four-wave critical-wave GRU kernel stages in LDS (gru_cw_device.h): the engine's ring for T <= 29 + pending

template <class R> struct cplx { R x, y; };

constexpr int kMaxFrameRows = 4096;                 // pe_update_many: frames one stream may complete per call (= task rows per tile)
constexpr int kFrameWaves = 4;                      // waves per workgroup of the MFCC frame role (one frame task per wave at a time)

// Constant tables of the one-frame-per-wave MFCC (mfcc_wave_tables.h builds the blob on the host)
template <class R>
struct WaveTables {
    const void* blob;           // device image, laid out as it sits in LDS (pe_wave::Layout)
    pe_wave::Layout L;
};

// x / d for 0 <= x < 2^31 and a divisor known on the host: one multiply-high and a shift
struct FastDiv {
    uint32_t mul; int shift;
    __host__ static FastDiv make(uint32_t d) {
        FastDiv f{0u, 0};
        if (d <= 1) { f.mul = 0u; f.shift = -1; return f; }            // x / 1 = x
        int l = 0;
        while ((1u << l) < d) ++l;                                       // l = ceil(log2 d) >= 1
        f.mul = (uint32_t)((((uint64_t)1 << (31 + l)) + d - 1) / d);     // ceil(2^(31+l) / d) < 2^32 since d > 2^(l-1)
        f.shift = l - 1;
        return f;
    }
    __host__ __device__ uint32_t div(uint32_t x) const {
        if (shift < 0) return x;
#if defined(__HIP_DEVICE_COMPILE__)
        return __umulhi(x, mul) >> shift;
#else
        return (uint32_t)(((uint64_t)x * mul) >> 32) >> shift;
#endif
    }
};

// ---- per-stream streaming state ------------------------------------------------------------------------------------
// One 16-byte record per stream and SIDE: q = samples held toward the next frame to compute (negative inside the dead zone
// between windows), kc = frames computed, ke = frames emitted (visible to the network; both mod 2^32), wcall = the number
// of the engine call that wrote the record.  Every stream has two sides (records AND leftover-PCM buffers): a call that
// advances a stream reads its CURRENT side and writes the other one, so the roles of one launch that read the state before
// the update (frame tasks, the network role of a fused launch) never see a half-written record -- and a stream that takes no
// part in a call keeps its current side untouched (pe_update_subset: streams advance independently, Listener.update per
// client, network_runner.py:125-146).  Current side = the one written most recently BEFORE this call: records stamped with
// this call's own number are being written right now and are ignored by its readers.  (Call numbers wrap after 2^32 calls;
// the host renumbers all records long before: engine.hip renormalize.)
struct __attribute__((aligned(16))) StreamRec { int32_t q; uint32_t kc, ke, wcall; };

// where side `side` of stream s sits: the two sides of a stream are neighbours (32 bytes, one cache line: a reader's two loads
// miss together).  Measured against the sides n_padded records apart: no difference at 4096 streams (profiles/round6/r6e_*)
__host__ __device__ __forceinline__ size_t rec_at(const uint32_t, const long long s, const int side) { return 2 * (size_t)s + (size_t)side; }

struct StreamState {
    StreamRec* rec;         // [n_padded][2 sides]
    int16_t* carry;         // [2][n_padded][carry_cap] leftover PCM of the one frame that straddles two calls
    uint32_t n_padded;      // records per side
    uint32_t call;          // number of this engine call (readers: records with wcall == call are not yet valid)
};

struct RecPair { StreamRec r0, r1; };
__device__ __forceinline__ RecPair rec_request(const StreamRec* rec, const uint32_t n_padded, const long long s) {
    RecPair p;              // two independent 16-byte loads: one round trip
    p.r0 = rec[rec_at(n_padded, s, 0)];
    p.r1 = rec[rec_at(n_padded, s, 1)];
    return p;
}
// side whose record is current for a reader of call `call`: smallest non-zero distance call - wcall
__device__ __forceinline__ int rec_side(const RecPair& p, const uint32_t call) {
    return (call - p.r1.wcall) - 1u < (call - p.r0.wcall) - 1u ? 1 : 0;
}
__device__ __forceinline__ StreamRec rec_pick(const RecPair& p, const int side) {
    StreamRec r;
    r.q = side ? p.r1.q : p.r0.q; r.kc = side ? p.r1.kc : p.r0.kc; r.ke = side ? p.r1.ke : p.r0.ke; r.wcall = side ? p.r1.wcall : p.r0.wcall;
    return r;
}

struct StreamGeom {
    int n_streams;
    int window;       // samples past a frame's start that make it visible: window_samples (+ hop for speechpy)
    int log_mode;     // 0: log(max(x, eps)) (sonopy safe_log);  1: log(x == 0 ? eps : x) (speechpy zero_handling)
    int hop;          // hop_samples
    int frame_len;    // min(window, n_fft): samples of a window that reach the FFT
    int n_filt;
    int n_mfcc;
    int n_features;   // T
    int ring_slots;   // power of two >= T + pending + 1
};

template <class R>
struct MfccStreamArgs {
    StreamGeom geo;
    const int16_t* pcm;     // [n_streams][chunk]
    int chunk;
    int pcm_pairs_ok;       // chunk even and pcm 4-byte aligned: int16 pairs may be loaded as one dword
    // geo.n_streams counts the streams of THIS launch; row v of pcm belongs to stream ids[v] (pe_update_subset) or, ids == null,
    // to stream v.  State, leftover PCM and ring rows are addressed by the stream, PCM rows and outputs by v.
    const int32_t* ids;
    StreamState st;         // read: each stream's current side; written: its other side (see StreamRec)
    // Leftover samples that stay where they lie (pe_update_device_keep, pe_update_async; full single updates only, position ==
    // stream).  head != null: the previous call's chunks [n_streams][head_chunk] are still alive, and stream s's q leftover
    // samples are the LAST q samples of its row there -- the carry is not read.  keep != 0: this call's own chunks will be the
    // next call's `head`, so the bookkeeping role moves no samples and only publishes the record (the leftover of a call is
    // always shorter than a frame, hence inside a chunk of >= frame_len - 1 samples).
    const int16_t* head;
    int head_chunk;
    int keep;
    float* ring;            // [n_tiles][ring_slots][16 streams][16 floats] -- or, ring_bf16, 16 bf16 per row (32 bytes)
    int ring_bf16;
    float* proj_ring;       // [n_tiles][ring_slots][16 streams][64 floats] x.W + b of every frame, or null
    // several updates per launch (pe_update_many): chunk u of stream s at pcm + (u*n_streams + s)*chunk
    int n_updates;
    int n_frame_rows;       // frame tasks per stream: the most frames one stream can complete in this call
    FastDiv div_hop, div_chunk;   // division by hop_samples / by the chunk length
    uint32_t* ke_hist;      // [n_updates][n_padded] emitted-frame counter after every update
    int n_padded;
};

template <class R>
struct MfccOfflineArgs {
    StreamGeom geo;
    const double* audio;    // [n_samples] float64 samples
    long long n_samples;
    long long n_frames;
    double* out;            // [n_frames][n_mfcc] float64, may be null
    float* out_rows;        // [n_frames][16] float32 rows (coefficients + zero padding), may be null
    double* out_mels;       // [n_frames][n_filt] float64 log-mel energies (Vectorizer.mels), may be null
};

// ---------------------------------------------------------------------------------------
// GRU + Dense head (model.py:76-82), register-resident weights, one wave per 16-stream tile
// ---------------------------------------------------------------------------------------
// gru_x3_device.h: packed operands of the float32 network on the XDL pipe (pe_set_gru_tiling(e, 2))
constexpr int kX3Tiles = 4;                 // TZ, TR, TC, TQ
constexpr int kX3RecOps = 4, kX3InOps = 3;   // A operands per output tile: recurrent, input
// blob (uint4 = 8 bf16 per lane): [AR: tile][m][lane] | [AX: tile][m][lane] | float wd[5][lane]
constexpr int kX3ArOff = 0;
constexpr int kX3AxOff = kX3Tiles * kX3RecOps * 64;                  // in uint4
constexpr int kX3WdOff = kX3AxOff + kX3Tiles * kX3InOps * 64;        // in uint4 (floats follow)
constexpr int kX3BlobBytes = kX3WdOff * 16 + 5 * 64 * 4;

// gru_b20_device.h: packed operands of the bf16-operand network of <= 20 units (pe_params.gru_precision = 1)
constexpr int kB20Tiles = 4;                // TZ, TR, TC, TQ
// blob (uint4 = 8 bf16 per lane): [AR: tile][lane] | [AX: tile][lane] | float wd[5][lane]
constexpr int kB20ArOff = 0;
constexpr int kB20AxOff = kB20Tiles * 64;
constexpr int kB20WdOff = 2 * kB20Tiles * 64;
constexpr int kB20BlobBytes = kB20WdOff * 16 + 5 * 64 * 4;

struct GruArgs {
    int n_streams;
    int n_features;         // T
    int n_in;               // F
    int units;              // H
    // packed MFMA A-operands, one float per lane (see pack_gru_weights in engine.hip)
    const float* wx;        // [NT][4][64]      input kernel, k-step kk <-> feature 4g+kk
    const float* wxd;       // [NT][4][64]      input kernel rows of the delta features (use_delta)
    int use_delta;          // inputs are [x_t, x_t - x_(t-1)] (first timestep's delta is 0); n_in stays F
    const float* wr1;       // [NT][R][64]      recurrent kernel rows of z/r slots (phase 1)
    const float* wr2;       // [NT][R][64]      recurrent kernel rows of candidate slots (phase 2)
    const float* bias;      // [NT][4][64]      accumulator init per output register
    const float* wd;        // [R][64]          dense kernel for unit 4*rho+g
    float dense_bias;
    // stock width (17 <= H <= 20) re-tiled as three full tiles + partial sums for units 16..19 (gru_cw_device.h,
    // CwPack): non-null = the launchers take the re-tiled shapes (gru_tile_cw / gru_tile_v), which agree with each
    // other bit for bit and with the classic tiling to float32 summation order
    const float* cw;
    // bf16-operand variant (gru_bf16_device.h): unit 8 g + i, tiles z0 z1 r0 r1 c0 c1
    int bf16;               // != 0: run the bf16 kernel
    const void* wx_bf16;    // [6][64] x 8 bf16    input kernel, k = 8 g + e <-> feature; k = 30, 31: bias hi, lo
    const void* wr_bf16;    // [6][64] x 8 bf16    recurrent kernel, k = 8 g + e <-> unit
    const float* wd_bf16;   // [8][64]
    // ... and in the five-values-per-lane layout of networks of <= 20 units (gru_b20_device.h): non-null = the bf16 launchers take it
    const void* b20;
    // float32 network on the XDL pipe, operands as three bf16 pieces (gru_x3_device.h; pe_set_gru_tiling(e, 2)):
    // non-null = the launchers take gru_tile_x3 for every input mode
    const void* x3;         // [4 tiles][4][64] + [4 tiles][3][64] uint4 of 8 bf16, then float wd[5][64]
    const void* reserved_;  // (null; keeps the layout of the argument segment the kernels' scalar loads were measured with)
    // input: either the feature ring (+ each stream's record: its emitted-frame counter) ...
    // n_streams counts the windows of THIS launch; window v belongs to stream ids[v] (pe_update_subset) or, ids == null, to
    // stream v: records and ring rows are addressed by the stream, out[] by v
    const int32_t* ids;
    const StreamRec* rec;   // [n_padded][2 sides]
    uint32_t n_padded, call;
    const uint32_t* ke_plain;   // non-null: the emitted-frame counter of stream s is ke_plain[s] (pe_update_many: one row of the
                                // per-update history) and the records are not read
    const float* ring;
    int ring_bf16;          // rows hold 16 bf16 (32 bytes) instead of 16 floats: bf16-operand kernel only
    // ... with, when the MFCC stage wrote it, the input projection x.W + b of every frame beside it, in MFMA slot
    // order [tile][slot][stream][g][output tile][q]: the network then starts every timestep from that accumulator
    // instead of recomputing the projection in each of the n_features windows the frame appears in
    const float* proj_ring;
    int ring_slots;
    // predict_ke != 0: the records hold the state BEFORE the update whose chunk is `chunk`
    // samples; the wave derives the post-update emitted count itself (fused MFCC || GRU launch)
    int predict_ke;
    int chunk, window, hop, frame_len;
    // ... or an explicit [n][T][F] float32 batch (Runner.predict), or -- row_stride > 0 -- one
    // [n_frames][16] float32 row sequence from which window w takes rows [w*row_stride, +T)
    const float* feats;
    int row_stride;
    float* out;             // [n_streams]
    int row_floats;         // floats per feature row: 16, or 32 for 17..32 coefficients per frame (one-wave kernel, KX = 2)
    int waves_per_tile;     // 1: one wave per tile (gru_tile);  4: four waves share a tile (gru_tile_mw)
};

// ---- which stream a lane of a network tile serves, and where its window ends ------------------------------------------
// v = position in this launch (tile * 16 + j); padded lanes shadow position 0's stream (their results are never stored)
__device__ __forceinline__ long long gru_stream_of(const GruArgs& a, const long long v, const bool valid) {
    return a.ids ? (long long)a.ids[valid ? v : 0] : v;
}
// first ring cell (in rows) of a stream: [tile][slot][16 streams] rows, slot 0
__device__ __forceinline__ size_t gru_ring_cell(const GruArgs& a, const long long sid) {
    return (size_t)(sid >> 4) * a.ring_slots * kTileStreams + (size_t)(sid & 15);
}
// the loads behind a window's end (issued early, consumed by gru_ke_resolve: nothing here waits)
struct KeRequest { RecPair p; uint32_t plain; };
__device__ __forceinline__ KeRequest gru_ke_request(const GruArgs& a, const long long sid) {
    // (the records are requested whether or not they will be used: loaded values that meet constants at a join of two branches
    //  cost a wait for them right there -- in front of the weight loads of the network's prologue, 0.36 us per launch of the
    //  stand-alone network, profiles/round6)
    KeRequest r;
    r.p = rec_request(a.rec, a.n_padded, sid);
    r.plain = 0u;
    if (a.ke_plain) r.plain = a.ke_plain[sid];
    return r;
}
// emitted-frame count the window of this launch ends at: the record's own, or -- predict_ke, the network role of a fused
// launch -- what this update will make of it (same arithmetic as mfcc_book_tile)
__device__ __forceinline__ uint32_t gru_ke_resolve(const GruArgs& a, const KeRequest& r) {
    if (a.ke_plain) return r.plain;
    const StreamRec c = rec_pick(r.p, rec_side(r.p, a.call));
    uint32_t ke = c.ke;
    if (a.predict_ke) {
        const int avail = c.q + a.chunk;
        const int nnew = avail >= a.frame_len ? 1 + (avail - a.frame_len) / a.hop : 0;
        const int qn = avail - nnew * a.hop;
        const int m = qn + a.hop * (int)(c.kc + (uint32_t)nnew - ke);
        if (m >= a.window) ke += 1u + (uint32_t)((m - a.window) / a.hop);
    }
    return ke;
}
__device__ __forceinline__ uint32_t gru_window_end(const GruArgs& a, const long long sid) { return gru_ke_resolve(a, gru_ke_request(a, sid)); }

struct WideLayerArgs {
    const float4* wx1;      // phase 1, input part     [wave][kx4][2 TPW][64] float4
    const float4* wr1;      // phase 1, recurrent part [wave][H/16][2 TPW][64] float4
    const float4* wx2;      // phase 2, input part     [wave][kx4][TPW][64] float4
    const float4* wr2;      // phase 2, recurrent part [wave][H/16][TPW][64] float4
    const float* b1;        // [wave][2 TPW][4][64]
    const float* b2;        // [wave][TPW][4][64]
    int kx4;                // k-groups of the input part: 1 for layer 0 (<= 16 features), H/16 above
};

struct WideArgs {
    GruArgs base;          // streams, T, ring / feats / rows addressing, predict_ke, out
    WideLayerArgs layer[2];
    int n_layers;
    int units;              // H
    const float* wd;        // [wave][TPW][4][64]
};

struct GatherArgs {         // ring -> [n][T][F] time-ordered features (update_vectors result)
    int n_streams, n_features, n_mfcc, ring_slots, ring_bf16;
    const float* ring;
    StreamState st;         // gather: read (call = a number no record carries); scatter: both sides rewritten
    float* out;
    int row_floats = kRowFloats;    // floats per feature row (32 when a frame has 17..32 coefficients)
};

struct ClearArgs {
    int n_streams, ring_slots;
    const uint8_t* mask;
    StreamState st;         // both sides rewritten: side 0 current (wcall = call), side 1 older
    float* ring;
    int ring_bf16;
    int32_t* activation;    // per-stream trigger state, may be null
    float* proj_ring;       // input-projection rows, may be null: a cleared row is the projection of a zero frame = bias
    const float* proj_b;    // [kProjRow]
    int row_floats = kRowFloats;
};

// ThresholdDecoder.decode + TriggerDetector.update for every stream (threshold_decoder.py:45-57,
// runner/precise_runner/runner.py:127-142)
struct DecodeArgs {
    int n_streams;
    const float* raw;           // [n] raw network outputs
    const double* cd;           // cumulative distribution LUT
    int cd_len, min_out, out_range;
    double center;
    double* conf_out;           // [n] decoded confidence, may be null
    // trigger (enabled when activation != null)
    int32_t* activation;        // [n] per-stream TriggerDetector.activation
    unsigned char* fired_out;   // [n] 1 where this update caused an activation, may be null
    double threshold;           // 1 - sensitivity
    int trigger_level, rearm;   // rearm = -(8 * 2048) // chunk_size
};
hipError_t launch_decode(const DecodeArgs& a, hipStream_t s);

// launchers implemented in kernels.hip
// MFCC of one call (n_updates chunks per stream): every frame the call completes as a task of one wave, then the
// per-stream bookkeeping (leftover samples to carry_next, counters to st_*_next, ke_hist)
constexpr int kProjRow = pe_wave::kProjRow;
hipError_t launch_mfcc_f64(const MfccStreamArgs<double>& a, const WaveTables<double>& t, int n_cus, hipStream_t s);
hipError_t launch_mfcc_f32(const MfccStreamArgs<float>& a, const WaveTables<float>& t, int n_cus, hipStream_t s);
// network for n_updates x n_streams windows, emitted counters from ke_hist, out[u][stream]
hipError_t launch_gru_many(const GruArgs& a, int n_updates, int n_padded, hipStream_t s);
// one launch, three roles: GRU waves read the feature windows as they will be after this update while MFCC waves
// compute this update's frames and the bookkeeping groups move the leftover (legal when chunk <= window -
// frame_len: no frame computed now becomes visible now)
hipError_t launch_fused_f64(const MfccStreamArgs<double>& m, const WaveTables<double>& t, const GruArgs& g, int n_cus, hipStream_t s);
hipError_t launch_fused_f32(const MfccStreamArgs<float>& m, const WaveTables<float>& t, const GruArgs& g, int n_cus, hipStream_t s);
hipError_t launch_mfcc_offline_f64(const MfccOfflineArgs<double>& a, const WaveTables<double>& t, int n_cus, hipStream_t s);
hipError_t launch_mfcc_offline_f32(const MfccOfflineArgs<float>& a, const WaveTables<float>& t, int n_cus, hipStream_t s);
hipError_t launch_gru_small(const GruArgs& a, int input_mode, hipStream_t s);   // units <= 32; 0 feats, 1 ring, 2 rows
int gru_small_regs(int units);                  // R = ceil(units/4)
int gru_wide_waves(int units);                  // waves per workgroup of the wide kernel (the weight packing follows it)
int gru_small_tiles(int units);                 // NT = ceil(3R/4)
hipError_t launch_gru_wide(const WideArgs& a, int input_mode, hipStream_t s);      // units 64..256, 1-2 layers
hipError_t launch_gru_wide_x3(const WideArgs& a, int input_mode, hipStream_t s);   // the same, float32 products on the bf16 pipe (its own weight packing)
hipError_t launch_gather(const GatherArgs& a, hipStream_t s);
hipError_t launch_scatter(const GatherArgs& a, hipStream_t s);   // a.out is read
// every record's wcall rewritten to 2 (current side) / 1 (other side): the host then continues counting calls from 3
hipError_t launch_renumber(const StreamState& st, int n_padded, hipStream_t s);
// leftovers that were kept in the previous call's chunks (MfccStreamArgs::head) are copied into each stream's CURRENT carry
// side, records untouched: after it every kernel finds the state where it always was (st.call: a reader's number)
hipError_t launch_materialize_carry(const StreamState& st, const int16_t* head, int head_chunk, int n_streams, hipStream_t s);
hipError_t launch_clear(const ClearArgs& a, hipStream_t s);
// proj[row][o] = b[o] + sum_c ring[row][c] w[c][o] for n_rows feature rows (after the ring was written from outside)
hipError_t launch_project_rows(const float* ring, float* proj, const float* w, const float* b, int n_mfcc, long long n_rows, hipStream_t s);

}  // namespace pe
