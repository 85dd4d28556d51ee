// Shared declarations between the C-ABI host code (engine.hip) and the gfx950 kernels.
// Everything here is MI355X-only: 64-lane wavefronts, 16x16x4 f32 MFMA tiles, LDS-staged
// 256-point FFTs.  No portability layer on purpose.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pe {

constexpr int kTileStreams = 16;    // streams per GRU tile == MFMA N dimension
constexpr int kRowFloats = 16;      // floats per feature-ring row (n_mfcc <= 16, rest zero)
constexpr int kCarryCap = 512;      // int16 samples of leftover PCM kept per stream (< n_fft)
constexpr int kNfft = 512;          // the only FFT size with a kernel
constexpr int kBins = kNfft / 2 + 1;
constexpr int kMaxFilt = 64;
constexpr int kMaxMelNnz = 1024;

template <class R> struct cplx { R x, y; };

// ---------------------------------------------------------------------------------------
// MFCC front end (vectorization.py:36-39 -> sonopy.mfcc_spec), streaming form
// ---------------------------------------------------------------------------------------
template <class R>
struct MfccTables {
    const cplx<R>* tw256;   // [16 k1][16 r]  exp(-2 pi i r k1 / 256)
    const cplx<R>* w512;    // [129]          exp(-2 pi i p / 512)
    const R* mel_w;         // [nnz]   filter weights, filter-major, zero weights trimmed
    const int* mel_start;   // [n_filt] first bin of each filter's support
    const int* mel_off;     // [n_filt+1] offsets into mel_w
    const R* dct;           // [n_mfcc][n_filt]  DCT-II, norm='ortho'
    int mel_nnz;
};

struct StreamGeom {
    int n_streams;
    int window;       // window_samples
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
    MfccTables<R> tab;
    const int16_t* pcm;     // [n_streams][chunk]
    int chunk;
    int pcm_pairs_ok;       // chunk even and pcm 4-byte aligned: int16 pairs may be loaded as one dword
    int16_t* carry;         // [n_streams_padded][kCarryCap]
    // stream state BEFORE this update (read) ...
    const int32_t* st_q;    // samples held toward the next frame to compute (may be < 0)
    const uint32_t* st_kc;  // frames computed so far (mod 2^32)
    const uint32_t* st_ke;  // frames emitted so far, i.e. visible to the network (mod 2^32)
    // ... and AFTER it (written).  Ping-pong buffers: the GRU role of a fused launch reads the
    // old state while MFCC workgroups are already publishing the new one.
    int32_t* st_q_next;
    uint32_t* st_kc_next;
    uint32_t* st_ke_next;
    float* ring;            // [n_tiles][ring_slots][16 streams][16 floats]
};

template <class R>
struct MfccOfflineArgs {
    StreamGeom geo;
    MfccTables<R> tab;
    const double* audio;    // [n_samples] float64 samples
    long long n_samples;
    long long n_frames;
    double* out;            // [n_frames][n_mfcc]
};

// ---------------------------------------------------------------------------------------
// GRU + Dense head (model.py:76-82), register-resident weights, one wave per 16-stream tile
// ---------------------------------------------------------------------------------------
struct GruArgs {
    int n_streams;
    int n_features;         // T
    int n_in;               // F
    int units;              // H
    // packed MFMA A-operands, one float per lane (see pack_gru_weights in engine.hip)
    const float* wx;        // [NT][4][64]      input kernel, k-step kk <-> feature 4g+kk
    const float* wr1;       // [NT][R][64]      recurrent kernel rows of z/r slots (phase 1)
    const float* wr2;       // [NT][R][64]      recurrent kernel rows of candidate slots (phase 2)
    const float* bias;      // [NT][4][64]      accumulator init per output register
    const float* wd;        // [R][64]          dense kernel for unit 4*rho+g
    float dense_bias;
    // input: either the feature ring (+ per-stream emitted-frame counters) ...
    const float* ring;
    const uint32_t* st_ke;
    int ring_slots;
    // predict_ke != 0: st_q/st_kc/st_ke hold the state BEFORE the update whose chunk is `chunk`
    // samples; the wave derives the post-update emitted count itself (fused MFCC || GRU launch)
    int predict_ke;
    const int32_t* st_q;
    const uint32_t* st_kc;
    int chunk, window, hop, frame_len;
    // ... or an explicit [n][T][F] float32 batch (Runner.predict)
    const float* feats;
    float* out;             // [n_streams]
};

struct GatherArgs {         // ring -> [n][T][F] time-ordered features (update_vectors result)
    int n_streams, n_features, n_mfcc, ring_slots;
    const float* ring;
    const uint32_t* st_ke;
    float* out;
};

struct ClearArgs {
    int n_streams, ring_slots;
    const uint8_t* mask;
    int32_t* st_q; uint32_t* st_kc; uint32_t* st_ke;
    float* ring;
};

// launchers implemented in kernels.hip
hipError_t launch_mfcc_stream_f64(const MfccStreamArgs<double>& a, hipStream_t s);
hipError_t launch_mfcc_stream_f32(const MfccStreamArgs<float>& a, hipStream_t s);
// one launch, two roles: GRU waves read the feature windows as they will be after this update
// while MFCC workgroups compute this update's frames (legal when chunk <= window - frame_len:
// no frame computed now becomes visible now)
hipError_t launch_fused_f64(const MfccStreamArgs<double>& m, const GruArgs& g, hipStream_t s);
hipError_t launch_fused_f32(const MfccStreamArgs<float>& m, const GruArgs& g, hipStream_t s);
hipError_t launch_mfcc_offline_f64(const MfccOfflineArgs<double>& a, hipStream_t s);
hipError_t launch_mfcc_offline_f32(const MfccOfflineArgs<float>& a, hipStream_t s);
size_t mfcc_lds_bytes(int real_size, int n_filt, int n_mfcc, int mel_nnz);
hipError_t launch_gru_small(const GruArgs& a, bool from_ring, hipStream_t s);   // units <= 32
int gru_small_regs(int units);                  // R = ceil(units/4)
int gru_small_tiles(int units);                 // NT = ceil(3R/4)
hipError_t launch_gather(const GatherArgs& a, hipStream_t s);
hipError_t launch_clear(const ClearArgs& a, hipStream_t s);

}  // namespace pe
