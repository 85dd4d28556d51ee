/*
 * precise_engine.h -- C ABI of the MI355X-native wake-word hot path
 * (libprecise_engine.so, built from mycroft_precise_amd/csrc by hipcc for gfx950).
 *
 * The reference (MycroftAI/mycroft-precise) is pure Python, so there is no existing FFI to
 * mirror; each entry point below replaces one Python-level seam of the reference and cites it
 * (paths relative to /root/reference).  A maintainer binds this library with ctypes exactly as
 * mycroft_precise_amd/_lib.py does -- see INTEGRATION.md.
 *
 * Conventions
 *   - every function returns a pe_status (0 = ok); nothing throws across the ABI;
 *     pe_last_error() returns a human-readable message for the last failure on that engine
 *     (pe_last_global_error() for failures of pe_create itself);
 *   - the caller owns every buffer it passes; the engine copies what it keeps;
 *   - an engine owns the state of n_streams independent audio streams (leftover PCM + the
 *     [n_features x n_mfcc] feature window of network_runner.py:102-104) on ONE device; it is
 *     not thread-safe (the reference's Listener is not either, network_runner.py:98-153);
 *   - "host" entry points take host pointers and synchronise; "*_device" entry points take
 *     device pointers of the engine's device plus a hipStream_t (as void*) and are asynchronous;
 *   - PCM is little-endian int16 mono (util.py:35-37), laid out [n_streams][chunk_samples].
 */
#ifndef PRECISE_ENGINE_H
#define PRECISE_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PE_ABI_VERSION 7

typedef enum pe_status {
    PE_OK = 0,
    PE_ERR_INVALID = 1,      /* bad argument (maps to ValueError)                         */
    PE_ERR_HIP = 2,          /* a HIP runtime call failed                                 */
    PE_ERR_UNSUPPORTED = 3,  /* parameter combination this build has no kernel for        */
    PE_ERR_NOMEM = 4,
    PE_ERR_EOF = 5           /* empty chunk (maps to EOFError, network_runner.py:133-134) */
} pe_status;

/* ListenerParams (precise/params.py:28-118): the derived sizes the hot path reads. */
typedef struct pe_params {
    int32_t sample_rate;     /* 16000                                        params.py:142 */
    int32_t window_samples;  /* 1600   int(sample_rate*window_t+0.5)         params.py:84  */
    int32_t hop_samples;     /* 800    int(sample_rate*hop_t+0.5)            params.py:89  */
    int32_t n_fft;           /* 512    ANY length in 16..1024 (np.fft.rfft takes any n),
                                       or the power of two 2048                       params.py:142 */
    int32_t n_filt;          /* 20     mel filters, 1..128                   params.py:142 */
    int32_t n_mfcc;          /* 13     coefficients kept, 1..32, <= n_filt   params.py:142
                                Front-end kernels: the stock shape (n_fft = 512, <= 64 filters whose runs fit the 64
                                lanes of a wave, <= 16 coefficients) runs on the one-frame-per-wave kernel and the fused
                                launch; EVERY OTHER shape in the ranges above runs on the general front end (same
                                results contract, two launches per update and per pe_update_many call -- one network
                                launch per update of the call for rows of 17..32 coefficients).  17..32 coefficients feed the float32 network of <= 32 units without
                                use_delta only; bf16 operands / rows take <= 16 coefficients on either front end.  An n_fft that is
                                not a power of two >= 64 (16 and 32 included) runs as Bluestein's chirp-z transform over
                                the next power of two >= max(128, 2 n_fft - 1) (one wave's LDS holds it up to n_fft =
                                1024).  Outside the ranges (n_fft > 2048, not a power of two and > 1024, < 16, ...):
                                PE_ERR_UNSUPPORTED.                                                                    */
    int32_t n_features;      /* 29     T, timesteps per network input        params.py:79  */
    int32_t use_delta;       /* 0      1: network inputs are [x_t, x_t - x_(t-1)]
                                       (vectorization.py:53-59), layer n_in = 2 n_mfcc   params.py:143 */
    int32_t mfcc_precision;  /* 0 = float64 front end (what the reference computes in,
                                network_runner.py:102,137);  1 = float32 front end        */
    int32_t gru_precision;   /* 0 = float32 matrix cores (reference precision, tol 1e-4);
                                1 = bf16 operands / float32 accumulate (BASELINE configs[4],
                                tol 1e-2)                                                    */
    int32_t vectorizer;      /* params.py:121-132: 2 = mfccs (sonopy, the default; also serves the
                                offline mels entry), 3 = speechpy_mfccs (legacy .params files without
                                a `vectorizer` key, params.py:147,155): one frame fewer per buffer
                                (a frame is emitted one hop later), exact zeros -- not small values --
                                replaced by eps before the log; the caller supplies that library's
                                filterbank as mel_filters.  0 is read as 2.                     */
    int32_t ring_precision;  /* 0 = float32 feature rows (64 B per frame and stream);
                                1 = bf16 feature rows (32 B), rounded to nearest even where the row is
                                stored -- needs gru_precision = 1 (BASELINE configs[4])         */
} pe_params;

/* One Keras GRU layer (precise/model.py:77-81), Keras weight layout, gate order z|r|h. */
typedef struct pe_gru_layer {
    int32_t n_in;                    /* F (13) for layer 0, units of the previous layer after */
    int32_t units;                   /* H (20): 1..32 register-resident kernels; 33..256 the streamed-weight
                                        kernel (zero-padded to a multiple of 64 inside)           */
    const float* kernel;             /* [n_in][3*units] row-major                              */
    const float* recurrent_kernel;   /* [units][3*units]                                       */
    const float* bias;               /* [3*units]                                              */
} pe_gru_layer;

/* Sequential([GRU..., Dense(1, sigmoid)])   (precise/model.py:76-82) */
typedef struct pe_weights {
    int32_t n_layers;                /* 1 in the reference; 2 = GRU(H1, return_sequences) -> GRU(H2), widths
                                        33..256 (BASELINE configs[3]: 256, 256)                   */
    const pe_gru_layer* layers;
    const float* dense_kernel;       /* [units_last]                                           */
    float dense_bias;
} pe_weights;

typedef struct pe_engine pe_engine;

int pe_abi_version(void);
const char* pe_last_global_error(void);

/* Replaces Listener.__init__ (network_runner.py:101-108) for n_streams streams at once.
 * mel_filters: [n_filt][n_fft/2+1] float64 row-major triangular filterbank, built by the host
 * exactly as the vectorizer the reference calls does (vectorization.py:36-39 -> sonopy). */
int pe_create(const pe_params* params, const double* mel_filters, const pe_weights* weights,
              int32_t n_streams, int32_t device, pe_engine** out);
int pe_destroy(pe_engine* e);
const char* pe_last_error(const pe_engine* e);

/* Listener.clear (network_runner.py:121-123).  mask: n_streams bytes, non-zero = clear that
 * stream; NULL = clear all.  Runs on the NULL stream and synchronises: a caller that drives the
 * *_device entry points on a non-blocking stream must synchronise that stream first (the same holds for
 * pe_get_vectors / pe_set_vectors / pe_get_stream_state). */
int pe_clear(pe_engine* e, const uint8_t* mask);

/* Listener.update up to, not including, ThresholdDecoder.decode (network_runner.py:148-152):
 * append one chunk per stream, emit any new MFCC frames into the feature window, run the
 * network on the window.  raw_out[n_streams] = raw sigmoid output (float32).
 * chunk_samples == 0 -> PE_ERR_EOF. */
int pe_update(pe_engine* e, const int16_t* pcm_host, int32_t chunk_samples, float* raw_out_host);
int pe_update_device(pe_engine* e, const int16_t* pcm_dev, int32_t chunk_samples,
                     float* raw_out_dev, void* hip_stream);

/* The same update for a caller whose device chunks OUTLIVE the call (a ring of resident PCM slabs, a capture buffer that is
 * written ahead).  Listener.update_vectors keeps the samples that do not yet fill a frame (network_runner.py:127-131: the
 * `leftover` of chop_array); pe_update_device copies them into the engine's own carry, every call, for every stream.  Here they
 * stay where they lie -- in pcm_dev -- and the NEXT call cuts the head of its first frame from there.  The promise: pcm_dev
 * stays allocated and unchanged until the work of the next call that advances or clears streams on this engine has completed
 * on its stream (or the engine is destroyed).  Any entry point may follow (a call of another style first moves the leftovers
 * to the carry in one small launch); results are bit-identical to pe_update_device.  Chunks that cannot hold a leftover (odd
 * length, fewer than frame_len - 1 samples, an address that is not 4-byte aligned, a non-stock front end) are taken exactly
 * as pe_update_device takes them.  A call whose chunks overlap the previous keep call's is refused (PE_ERR_INVALID: the leftovers
 * would be gone; the streams' state is untouched).  pe_update_async works this way by itself: its device chunks are the engine's own. */
int pe_update_device_keep(pe_engine* e, const int16_t* pcm_dev, int32_t chunk_samples,
                          float* raw_out_dev, void* hip_stream);

/* Streams that advance independently.  Every reference Listener consumes chunks at its own pace (network_runner.py:125-146;
 * one engine process per client, runner/precise_runner/runner.py:54-67, :232-243); a server that multiplexes many clients on
 * one engine has audio for SOME of them at any moment.  pe_update_subset: stream stream_ids[i] (0 <= id < n_streams, each at
 * most once) takes chunk i of pcm[n_active][chunk_samples] and gets raw_out[i]; every other stream keeps its leftover
 * samples, counters and feature window untouched.  Same launches as pe_update (the fused one where pe_update fuses), sized by
 * n_active: cost follows the active streams, not the engine.  Results are those of a private Listener per stream, bit-identical
 * to pe_update whenever the same streams get the same chunks.  n_active == 0 is a no-op (PE_OK); chunk_samples == 0 with
 * n_active > 0 -> PE_ERR_EOF.  Callers with several chunk lengths at once issue one call per length.
 * The host entry point validates the ids (PE_ERR_INVALID: out of range / named twice); the device entry point cannot: ids
 * out of range or repeated are undefined behaviour there. */
int pe_update_subset(pe_engine* e, const int32_t* stream_ids_host, int32_t n_active, const int16_t* pcm_host,
                     int32_t chunk_samples, float* raw_out_host);
int pe_update_subset_device(pe_engine* e, const int32_t* stream_ids_dev, int32_t n_active, const int16_t* pcm_dev,
                            int32_t chunk_samples, float* raw_out_dev, void* hip_stream);

/* Host-fed pipeline: the reference's engine is handed HOST bytes per chunk (precise/scripts/engine.py:60-63,
 * runner/precise_runner/runner.py:62-67), and pe_update above is copy -> launch -> copy, one after the other.
 * pe_update_async enqueues the same update and returns: the chunk of update u + 1 crosses PCIe (a copy stream of the
 * engine's own) while update u runs (a compute stream of the engine's own), the probabilities come back behind the
 * launch; up to 3 updates are in flight, a 4th call first delivers the oldest.  raw_out_host[n_streams] is valid after
 * pe_wait (or once 3 more updates have been enqueued).  Results are bit-identical to pe_update / pe_update_device.
 *   - pageable pcm_host: staged by the HIP runtime at the call (the caller's PCM buffer is free again when the call
 *     returns); pageable raw_out_host: filled from the engine's pinned ring when the update is delivered;
 *   - buffers from pe_host_alloc (pinned, device-visible; freed by pe_host_free or pe_destroy): ZERO-COPY -- the DMA
 *     reads / writes them directly, which is what reaches PCIe line rate (a CPU memcpy of 8 MB per update does not);
 *     such a PCM buffer must stay untouched until pe_wait or until 3 more updates have been enqueued.
 * Every other entry point that reads or moves the streams' state (pe_update*, pe_clear, pe_get_vectors, ...) first
 * waits for the updates in flight, so the two styles may be mixed; callers that drive the *_device entry points on
 * their own non-blocking stream synchronise that stream before switching to pe_update_async (as for pe_clear). */
int pe_host_alloc(pe_engine* e, size_t bytes, void** out);
int pe_host_free(pe_engine* e, void* p);
int pe_update_async(pe_engine* e, const int16_t* pcm_host, int32_t chunk_samples, float* raw_out_host);
int pe_wait(pe_engine* e);

/* n_updates consecutive pe_update calls in two launches (results bit-identical): chunk u of stream s at
 * pcm[(u * n_streams + s) * chunk_samples], raw_out[u * n_streams + s].  First one launch computes every
 * MFCC frame the call completes (frame-parallel: which samples form which frame is closed-form over
 * leftover ++ chunk 0 ++ ... ++ chunk n-1), then one launch runs the network for all
 * n_updates x n_streams windows -- for callers that can buffer a few chunks (catch-up, bulk
 * replay, latency-tolerant servers) this fills the machine where a single update of a few thousand
 * streams cannot.  pe_reserve_updates sizes the feature ring, the second leftover buffer and the
 * per-update counters for it (and restarts all streams); n_updates * chunk_samples < 2^30.
 * Engines on the general front end (see pe_params) take one front-end launch per call as well, then the batched network
 * launch (rows of 17..32 coefficients: one network launch per update of the call); same bits.  The enlarged ring stays: later single pe_update calls on a reserved engine are unchanged in their
 * results but, up to 8192 streams, use the one-wave network shape (the critical-wave shape stages exactly 32 ring slots in
 * LDS) -- reserve only on engines that use pe_update_many.
 * Network form of a reserved engine (pe_set_gru_tiling -1, the default): when max_updates x stream tiles exceeds four per
 * compute unit (e.g. 4096 streams x 8 updates), the float32 network takes form 2 (float32 products on the bf16 pipe) for ALL
 * launches of the engine -- the batched launch is what the engine was reserved for (402 vs 296 M windows/s at that size), and
 * one form per engine keeps pe_update == pe_update_many bit for bit.  Single updates on such an engine take two launches.  An
 * unreserved engine of the same size runs form 1: the two agree to float32 summation order (<= 1e-6), not bit for bit. */
int pe_reserve_updates(pe_engine* e, int32_t max_updates, int32_t max_chunk_samples);
int pe_update_many(pe_engine* e, const int16_t* pcm_host, int32_t chunk_samples, int32_t n_updates, float* raw_out_host);
int pe_update_many_device(pe_engine* e, const int16_t* pcm_dev, int32_t chunk_samples, int32_t n_updates,
                          float* raw_out_dev, void* hip_stream);

/* Listener.update_vectors (network_runner.py:125-146): as pe_update without the network;
 * feats_out[n_streams][n_features][n_mfcc] float32, oldest row first (may be NULL). */
int pe_update_vectors(pe_engine* e, const int16_t* pcm_host, int32_t chunk_samples,
                      float* feats_out_host);
int pe_update_vectors_device(pe_engine* e, const int16_t* pcm_dev, int32_t chunk_samples,
                             float* feats_out_dev, void* hip_stream);

/* Listener.mfccs (network_runner.py:104,144): copy out the current feature windows,
 * feats_out[n_streams][n_features][n_mfcc] float32, oldest row first; consumes no audio. */
int pe_get_vectors(pe_engine* e, float* feats_out_host);

/* Assignment to Listener.mfccs / a Listener whose runner is replaced after construction
 * (scripts/train_incremental.py:87-88): every stream restarts (pe_clear) with the given feature window
 * already emitted and no leftover audio; feats[n_streams][n_features][n_mfcc] float32, oldest row
 * first.  Feed the leftover samples back with pe_update_vectors to restore a whole Listener state. */
int pe_set_vectors(pe_engine* e, const float* feats_host);

/* Run the network on the current feature windows without consuming audio. */
int pe_run_device(pe_engine* e, float* raw_out_dev, void* hip_stream);

/* Runner.predict (network_runner.py:35-38): feats[n][n_features][feature_size] float32 -> out[n]
 * (feature_size = n_mfcc, or 2 n_mfcc with use_delta: the batch then carries its delta columns). */
int pe_predict(pe_engine* e, const float* feats_host, int32_t n, float* out_host);
int pe_predict_device(pe_engine* e, const float* feats_dev, int32_t n, float* out_dev,
                      void* hip_stream);

/* vectorize_raw (vectorization.py:46-50) for the Vectorizer.mfccs entry (:36-39): stateless
 * MFCC of one whole buffer.  audio: float64 samples in [-1,1) (what the reference's vectorizer
 * receives).  feats_out[max_frames][n_mfcc] float64; *n_frames_out = 1+(n-window)//hop or 0. */
int pe_vectorize_raw(pe_engine* e, const double* audio_host, int64_t n_samples,
                     double* feats_out_host, int64_t max_frames, int64_t* n_frames_out);

/* The same buffer through the Vectorizer.mels entry (vectorization.py:32-35 -> sonopy.mel_spec): log of the
 * mel filterbank energies, no DCT.  mels_out[max_frames][n_filt] float64.  Offline form only, as in the reference:
 * its Listener allocates the feature window n_mfcc wide (network_runner.py:104,123), so mel rows of n_filt columns
 * cannot stream through it either; pe_create refuses vectorizer = 1. */
int pe_vectorize_mels(pe_engine* e, const double* audio_host, int64_t n_samples,
                      double* mels_out_host, int64_t max_frames, int64_t* n_frames_out);

/* The reference's batched offline evaluation (precise/scripts/simulate.py:92-104, also
 * annoyance_estimator.py:114-130): MFCC of one whole recording, one network input per hop_frames
 * (= chunk_size // hop_samples) frames -- windows ending at frame i for i in range(n_features,
 * n_frames, hop_frames) -- all predicted in one batch.  out[n_windows] raw outputs; nothing of the
 * [n_windows][T][F] batch is materialised: the network reads overlapping windows of one row
 * sequence.  Stateless (the engine's streams are untouched). */
int pe_evaluate(pe_engine* e, const double* audio_host, int64_t n_samples, int32_t hop_frames,
                float* out_host, int64_t max_windows, int64_t* n_windows_out);

/* ThresholdDecoder.decode (threshold_decoder.py:45-57) and TriggerDetector.update
 * (runner/precise_runner/runner.py:127-142) for every stream, on the device.
 * pe_set_decoder: cd = the decoder's cumulative table (np.cumsum of the summed normal pdfs,
 * threshold_decoder.py:42,68-70), min_out / out_range / center as the Python object holds them.
 * pe_set_trigger: (re)arms one TriggerDetector per stream (chunk_size in BYTES as in runner.py:122).
 * pe_decode*: raw[n_streams] float32 -> conf[n_streams] float64 (may be NULL) and, when a trigger is
 * set, fired[n_streams] (1 = this prediction caused an activation; may be NULL).  The logit follows the
 * reference's evaluation on the runner's float32 scalar (functions.py:99-101: `1 / x - 1` rounds twice in
 * float32, the logarithm is taken in float64), so the table bin -- and with it the decoded value -- is the
 * one Listener.update returns, not the one a float64 logit would give. */
int pe_set_decoder(pe_engine* e, const double* cd, int32_t cd_len, int32_t min_out, int32_t out_range, double center);
int pe_set_trigger(pe_engine* e, int32_t chunk_size_bytes, double sensitivity, int32_t trigger_level);
int pe_decode_device(pe_engine* e, const float* raw_dev, double* conf_out_dev, unsigned char* fired_out_dev, void* hip_stream);
int pe_decode(pe_engine* e, const float* raw_host, double* conf_out_host, unsigned char* fired_out_host);

/* Introspection used by tests and the bench. */
typedef struct pe_info {
    int32_t n_streams, n_features, n_mfcc, units, n_layers, ring_slots, carry_capacity;
    int32_t mfcc_precision, gru_precision;
    int64_t device_bytes;            /* HBM held by this engine                                */
} pe_info;
int pe_get_info(const pe_engine* e, pe_info* out);

/* Per-stream streaming state, for tests: q = samples held toward the next frame (may be
 * negative inside the dead zone between windows), frames computed / emitted so far (mod 2^32). */
int pe_get_stream_state(pe_engine* e, int32_t* q_out, uint32_t* computed_out, uint32_t* emitted_out);

/* Test aid.  Each stream's state is a pair of 16-byte records stamped with the number of the engine call that wrote them
 * (csrc/pe_common.h: StreamRec); the 32-bit call count is renumbered in place long before it wraps (default: at 0x7fff0000
 * calls).  This moves the threshold (8..0x7fff0000) so that a test can cross it. */
int pe_set_renumber_at(pe_engine* e, uint32_t call_number);

/* Kernel sequencing of pe_update*: 1 (default) = when chunk_samples <= window - min(window,
 * n_fft) -- no frame computed by an update can become visible in the same update -- the MFCC
 * and network roles run concurrently inside ONE launch (networks with a fused instantiation:
 * gru_precision 0 / 1 on the stock front-end shape); 0 = always two dependent launches. */
int pe_set_fused(pe_engine* e, int32_t enabled);

/* Input projections: 1 = the MFCC stage stores x.W + b of every frame beside its feature row (256 bytes per frame and
 * stream) and the network starts each timestep from that row instead of recomputing the projection in each of the
 * n_features windows a frame appears in (16 of its 41 MFMAs per timestep); 0 = recompute (the default: measured, the
 * rows cost more to load every timestep than the MFMAs they save, profiles/DESIGN_notebook_r1-r5.md 4.6).  Available for the float32
 * network of 17..20 units without delta features.  Both settings agree to float32 rounding (different summation
 * order), each is deterministic; changing the setting restarts all streams. */
int pe_set_input_projection(pe_engine* e, int32_t enabled);

/* Network kernel shape: 0 (default) = automatic -- four waves share each 16-stream tile while the engine has few
 * tiles (stock width 17..20 units, re-tiled: up to 2 tiles per compute unit = 8192 streams on MI355X, the
 * critical-wave kernel, use_delta included; other widths: up to 1 tile per compute unit, and use_delta on the
 * one-wave kernel), one wave per tile beyond -- 1 / 4 = forced: results are bit-identical. */
int pe_set_gru_waves(pe_engine* e, int32_t waves_per_tile);

/* Form of the float32 network (model.py:76-82), -1 (default) = automatic by engine size:
 *   0 = the classic four output tiles on v_mfma_f32_16x16x4_f32;
 *   1 = stock width (17..20 units) re-tiled: three full MFMA tiles + partial sums for units 16..19 (csrc/gru_cw_device.h:
 *       shortens the four-wave kernel's timestep); automatic while the engine has no more than two tiles per compute unit;
 *   2 = float32 products on the bf16 matrix pipe (csrc/gru_x3_device.h): every operand as three bf16 pieces that add up
 *       to the float32 value exactly, six piece products per multiplication, float32 accumulate / gates / state -- the
 *       float32 tolerance, at the float32 kernels' distance to a float64 evaluation.  <= 20 units, <= 15 inputs, no
 *       use_delta (PE_ERR_UNSUPPORTED otherwise).  On gfx950 an f32-input MFMA keeps its whole SIMD from issuing while
 *       it runs and a bf16 MFMA does not; automatic for engines with more stream tiles than the machine has SIMDs (more
 *       than four per compute unit: above 16 384 streams on MI355X), where it takes two launches per update instead of
 *       the fused one and is still faster.
 * Every kernel shape of ONE form agrees bit for bit (pe_update / pe_update_many / pe_predict / pe_evaluate, one or four
 * waves, fused or not); the forms agree to float32 summation order (<= 1e-6 on the probability).  Ignored with projection
 * rows; use_delta models of the stock width follow 0 / 1.
 * Wide / stacked networks (33..256 units) have two forms: 0 (and -1, the default) = the streamed-weight kernel on f32-input MFMAs
 * (csrc/gru_wide_device.h), 2 = float32 products on the bf16 pipe with the float32 weight stream split into three bf16 pieces in
 * registers, on the matrix pipe, every timestep (csrc/gru_wide_x3_device.h): same tolerance, measured EQUAL in time at 256 x 2
 * units (five 4-pass MFMAs + twelve vector instructions against four 8-pass MFMAs per tile and 16 source units), kept as the
 * form whose matrix time would shrink with a narrower weight stream; 1 is refused.
 * bf16-operand networks (gru_precision = 1) have two layouts of the same arithmetic contract (tolerance 1e-2, each bit-stable
 * across pe_update / pe_update_many / pe_predict): 1 (and -1, the default, where it fits: <= 20 units, <= 14 features) = five
 * gate values per lane (csrc/gru_b20_device.h: 9 MFMAs per timestep), 0 = eight values per lane (csrc/gru_bf16_device.h: 12
 * MFMAs, every width up to 32); 2 is refused.
 * pe_get_gru_tiling: the form this engine's launches take now (0 / 1 / 2; -2 for a null engine). */
int pe_set_gru_tiling(pe_engine* e, int32_t tiling);
int pe_get_gru_tiling(const pe_engine* e);

/* HIP-event timing of the kernels launched by the last *_device/host update on this engine
 * (milliseconds; measured on the stream the kernels ran on).  Enabled with pe_set_timing(e,1).
 * With a fused launch mfcc_ms is the whole update and gru_ms is 0. */
int pe_set_timing(pe_engine* e, int32_t enabled);
int pe_get_last_timing(pe_engine* e, float* mfcc_ms, float* gru_ms);

#ifdef __cplusplus
}
#endif
#endif /* PRECISE_ENGINE_H */
