// Every __global__ entry point of libprecise_engine.so and its launcher.  The device code lives
// in mfcc_device.h (MFCC front end) and gru_device.h (GRU + Dense on the f32 matrix cores).
#include "mfcc_device.h"
#include "gru_device.h"
#include "gru_bf16_device.h"
#include "gru_wide_device.h"

namespace pe {

// ---- MFCC, streaming: one workgroup per 16-stream tile ---------------------------------------
// Grid = n_tiles * nsel workgroups; workgroup b serves tile b / nsel, frame subset b % nsel.
template <class R>
__global__ __launch_bounds__(256) void mfcc_stream_kernel(const MfccStreamArgs<R> a, const int nsel) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    mfcc_stream_tile<R>(a, blockIdx.x / nsel, smem, blockIdx.x % nsel, nsel);
}

// (waves_per_eu(3): the frame rows need 134 VGPRs in float64; left to itself the allocator takes 231 and two
// of these workgroups then fill a SIMD's register file, so that no network wave of a concurrent launch fits)
template <class R>
__global__ __launch_bounds__(16 * kThroughputGroups) __attribute__((amdgpu_waves_per_eu(3))) void mfcc_many_kernel(const MfccStreamArgs<R> a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    mfcc_many_tile<R, false>(a, smem);
}

template <class R>
__global__ __launch_bounds__(256) void mfcc_many_book_kernel(const MfccStreamArgs<R> a) {
    mfcc_many_tile<R, true>(a, nullptr);
}

// network for a whole batch of updates: workgroup (one wave) b serves update b / n_tiles, tile b % n_tiles
template <int R>
__global__ __launch_bounds__(64) void gru_many_kernel(const GruArgs a, const int n_tiles, const int n_padded) {
    const int u = blockIdx.x / n_tiles, tile = blockIdx.x % n_tiles;
    GruArgs b = a;
    b.st_ke = a.st_ke + (size_t)u * n_padded;
    b.out = a.out + (size_t)u * a.n_streams;
    b.predict_ke = 0;
    gru_tile<R, kRing>(b, tile, threadIdx.x);
}

// same, four waves sharing each (update, tile) -- few tiles per SIMD: latency matters more than issue slots
template <int R>
__global__ __launch_bounds__(256) void gru_many_mw_kernel(const GruArgs a, const int n_tiles, const int n_padded) {
    __shared__ __attribute__((aligned(16))) float S[3 * R * 64 + 256];
    const int u = blockIdx.x / n_tiles, tile = blockIdx.x % n_tiles;
    GruArgs b = a;
    b.st_ke = a.st_ke + (size_t)u * n_padded;
    b.out = a.out + (size_t)u * a.n_streams;
    b.predict_ke = 0;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    gru_tile_mw_any<R>(b, tile, wave, threadIdx.x & 63, S);
}

__global__ __launch_bounds__(64) void gru_many_bf16_kernel(const GruArgs a, const int n_tiles, const int n_padded) {
    const int u = blockIdx.x / n_tiles, tile = blockIdx.x % n_tiles;
    GruArgs b = a;
    b.st_ke = a.st_ke + (size_t)u * n_padded;
    b.out = a.out + (size_t)u * a.n_streams;
    b.predict_ke = 0;
    gru_tile_bf16<kRing>(b, tile, threadIdx.x);
}

template <class R>
__global__ __launch_bounds__(16 * kThroughputGroups) void mfcc_offline_kernel(const MfccOfflineArgs<R> a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    mfcc_offline_block<R>(a, smem);
}

// ---- GRU: one wave per 16-stream tile ----------------------------------------------------------
template <int R, int MODE>
__global__ __launch_bounds__(64) void gru_small_kernel(const GruArgs a) {
    gru_tile<R, MODE>(a, blockIdx.x, threadIdx.x);
}

// ---- wide / stacked GRU: one workgroup per 16-stream tile, weights streamed from L2 -------------------
template <int TPW, int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void gru_wide_kernel(const WideArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef PE_WIDE_STAGGER
    for (int i = (blockIdx.x >> 3) & 31; i > 0; --i) __builtin_amdgcn_s_sleep(PE_WIDE_STAGGER);
#endif
    gru_wide_tile<TPW, MODE, WAVES>(a, blockIdx.x, wave, threadIdx.x & 63, reinterpret_cast<float*>(smem));
}

template <int TPW, int WAVES>
static hipError_t launch_wide_t(const WideArgs& a, int mode, hipStream_t s) {
    const int tiles = (a.base.n_streams + kTileStreams - 1) / kTileStreams;
    if (tiles == 0) return hipSuccess;
    const size_t lds = (size_t)4 * (TPW * WAVES) * 256 * sizeof(float);      // 2 layers x {h, r*h}
    if (mode == kRing) hipLaunchKernelGGL((gru_wide_kernel<TPW, kRing, WAVES>), dim3(tiles), dim3(64 * WAVES), lds, s, a);
    else if (mode == kRows) hipLaunchKernelGGL((gru_wide_kernel<TPW, kRows, WAVES>), dim3(tiles), dim3(64 * WAVES), lds, s, a);
    else hipLaunchKernelGGL((gru_wide_kernel<TPW, kFeats, WAVES>), dim3(tiles), dim3(64 * WAVES), lds, s, a);
    return hipGetLastError();
}

// 8 waves per workgroup (two per SIMD) for H = 128 / 256 was measured and is NOT faster: 256 x 2 layers, 4096
// streams: 4 waves 1.254 ms per launch, 8 waves 1.325 ms (tools/gpu_wide.py) -- kept as a build switch only.
#ifndef PE_WIDE8
#define PE_WIDE8 0
#endif
int gru_wide_waves(int units) { return (PE_WIDE8 && units % 128 == 0) ? 8 : 4; }

hipError_t launch_gru_wide(const WideArgs& a, int mode, hipStream_t s) {
    switch (a.units / 64) {
        case 1: return launch_wide_t<1, 4>(a, mode, s);
        case 2: return PE_WIDE8 ? launch_wide_t<1, 8>(a, mode, s) : launch_wide_t<2, 4>(a, mode, s);
        case 3: return launch_wide_t<3, 4>(a, mode, s);
        case 4: return PE_WIDE8 ? launch_wide_t<2, 8>(a, mode, s) : launch_wide_t<4, 4>(a, mode, s);
        default: return hipErrorInvalidValue;
    }
}

// ---- GRU, bf16 operands: one wave per 16-stream tile --------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(64) void gru_bf16_kernel(const GruArgs a) {
    gru_tile_bf16<MODE>(a, blockIdx.x, threadIdx.x);
}

// fused update with the bf16 network role (four tiles per GRU workgroup, one wave each)
template <class R>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void fused_update_bf16_kernel(const MfccStreamArgs<R> m, const GruArgs g,
                                                                const int n_gru_blocks, const int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x;
    if (b < n_gru_blocks) {
        const int tile = b * 4 + (threadIdx.x >> 6);
        if (tile < n_tiles) gru_tile_bf16<kRing>(g, tile, threadIdx.x & 63);
    } else {
        mfcc_stream_tile<R>(m, b - n_gru_blocks, smem);
    }
}

// ---- GRU: four waves per 16-stream tile (few tiles: fills all four SIMDs of a CU) -----------------
template <int R>
__global__ __launch_bounds__(256) void gru_mw_kernel(const GruArgs a) {
    __shared__ __attribute__((aligned(16))) float S[3 * R * 64 + 256];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    gru_tile_mw_any<R>(a, blockIdx.x, wave, threadIdx.x & 63, S);
}

// ---- fused update: GRU role || MFCC role in ONE launch ------------------------------------------
// Workgroups [0, n_gru_blocks) run the network on the feature windows as they will stand after this
// update; workgroups after them compute this update's MFCC frames.  The GRU workgroups are
// dispatched first: they are the long pole.  MW = true: one GRU workgroup per tile, its four waves
// share the tile (gru_tile_mw); MW = false: four tiles per GRU workgroup, one wave each.
// (waves_per_eu(2): a GRU and an MFCC workgroup must fit one CU together, i.e. <= 256 registers per lane;
// without the bound an innocent change to the MFCC code once pushed the allocation to 267 and the
// launch from 21 to 30 us.)
template <class R, int RG, bool MW>
#ifndef PE_FUSED_WPE
#define PE_FUSED_WPE 2
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PE_FUSED_WPE))) void fused_update_kernel(const MfccStreamArgs<R> m, const GruArgs g,
                                                           const int n_gru_blocks, const int n_tiles, const int nsel) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x;
    if (b < n_gru_blocks) {
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        if (MW) {
            gru_tile_mw_any<RG>(g, b, wave, threadIdx.x & 63, reinterpret_cast<float*>(smem));
        } else {
            const int tile = b * 4 + wave;
            if (tile < n_tiles) gru_tile<RG, kRing>(g, tile, threadIdx.x & 63);
        }
    } else {
        const int mb = b - n_gru_blocks;
        mfcc_stream_tile<R>(m, mb / nsel, smem, mb % nsel, nsel);
    }
}

// Workgroups per tile for the MFCC role.  Splitting the frames of one update over two workgroups per
// tile was measured twice (4096 streams, MI355X): 31.6 us vs 22.9 us with a 73 KB LDS image, 31.5 us vs
// 22.1 us with the present 57 KB one (33 us with the kernel held to 3 waves per SIMD) -- hence 1.  The
// kernels keep the (fsel, nsel) parameters for larger-chunk use; PE_FRAME_SPLIT is the build switch.
#ifndef PE_FRAME_SPLIT
#define PE_FRAME_SPLIT 1
#endif
static int frame_split(const StreamGeom&, int, int) { return PE_FRAME_SPLIT; }

template <class R>
static hipError_t launch_stream(const MfccStreamArgs<R>& a, hipStream_t s) {
    const int tiles = (a.geo.n_streams + kTileStreams - 1) / kTileStreams;
    const size_t lds = lds_layout_bytes(sizeof(R), a.geo.n_filt, a.geo.n_mfcc, a.tab.mel_parts, 16);
    const int nsel = frame_split(a.geo, a.chunk, tiles);
    hipLaunchKernelGGL(mfcc_stream_kernel<R>, dim3(tiles * nsel), dim3(256), lds, s, a, nsel);
    return hipGetLastError();
}

template <class R>
static hipError_t launch_offline(const MfccOfflineArgs<R>& a, hipStream_t s) {
    if (a.n_frames <= 0) return hipSuccess;
    const long long blocks = (a.n_frames + kThroughputGroups - 1) / kThroughputGroups;
    const size_t lds = lds_layout_bytes(sizeof(R), a.geo.n_filt, a.geo.n_mfcc, a.tab.mel_parts, kThroughputGroups);
    hipLaunchKernelGGL(mfcc_offline_kernel<R>, dim3((unsigned)blocks), dim3(16 * kThroughputGroups), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_mfcc_stream_f64(const MfccStreamArgs<double>& a, hipStream_t s) { return launch_stream<double>(a, s); }
hipError_t launch_mfcc_stream_f32(const MfccStreamArgs<float>& a, hipStream_t s) { return launch_stream<float>(a, s); }
template <class R>
static hipError_t launch_many(const MfccStreamArgs<R>& a, hipStream_t s) {
    const int tiles = (a.geo.n_streams + kTileStreams - 1) / kTileStreams;
    const size_t lds = lds_layout_bytes(sizeof(R), a.geo.n_filt, a.geo.n_mfcc, a.tab.mel_parts, kThroughputGroups);
    const long long groups = (long long)tiles * a.n_frame_rows * kTileStreams;
    hipLaunchKernelGGL(mfcc_many_kernel<R>, dim3((unsigned)((groups + kThroughputGroups - 1) / kThroughputGroups)),
                       dim3(16 * kThroughputGroups), lds, s, a);
    hipLaunchKernelGGL(mfcc_many_book_kernel<R>, dim3(tiles), dim3(256), 0, s, a);     // reads what the rows read, writes elsewhere
    return hipGetLastError();
}
hipError_t launch_mfcc_many_f64(const MfccStreamArgs<double>& a, hipStream_t s) { return launch_many<double>(a, s); }
hipError_t launch_mfcc_many_f32(const MfccStreamArgs<float>& a, hipStream_t s) { return launch_many<float>(a, s); }

hipError_t launch_mfcc_offline_f64(const MfccOfflineArgs<double>& a, hipStream_t s) { return launch_offline<double>(a, s); }
hipError_t launch_mfcc_offline_f32(const MfccOfflineArgs<float>& a, hipStream_t s) { return launch_offline<float>(a, s); }

int gru_small_regs(int units) { return (units + 3) / 4; }
int gru_small_tiles(int units) { return (3 * gru_small_regs(units) + 3) / 4; }

template <int R>
static hipError_t launch_r(const GruArgs& a, int mode, hipStream_t s) {
    const int tiles = (a.n_streams + kTileStreams - 1) / kTileStreams;
    if (tiles == 0) return hipSuccess;
    if (mode == kRing && a.waves_per_tile == 4) hipLaunchKernelGGL((gru_mw_kernel<R>), dim3(tiles), dim3(256), 0, s, a);
    else if (mode == kRing) hipLaunchKernelGGL((gru_small_kernel<R, kRing>), dim3(tiles), dim3(64), 0, s, a);
    else if (mode == kRows) hipLaunchKernelGGL((gru_small_kernel<R, kRows>), dim3(tiles), dim3(64), 0, s, a);
    else hipLaunchKernelGGL((gru_small_kernel<R, kFeats>), dim3(tiles), dim3(64), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_gru_small(const GruArgs& a, int from_ring, hipStream_t s) {
    if (a.bf16) {
        const int tiles = (a.n_streams + kTileStreams - 1) / kTileStreams;
        if (tiles == 0) return hipSuccess;
        if (from_ring == kRing) hipLaunchKernelGGL((gru_bf16_kernel<kRing>), dim3(tiles), dim3(64), 0, s, a);
        else if (from_ring == kRows) hipLaunchKernelGGL((gru_bf16_kernel<kRows>), dim3(tiles), dim3(64), 0, s, a);
        else hipLaunchKernelGGL((gru_bf16_kernel<kFeats>), dim3(tiles), dim3(64), 0, s, a);
        return hipGetLastError();
    }
    switch (gru_small_regs(a.units)) {
        case 1: return launch_r<1>(a, from_ring, s);
        case 2: return launch_r<2>(a, from_ring, s);
        case 3: return launch_r<3>(a, from_ring, s);
        case 4: return launch_r<4>(a, from_ring, s);
        case 5: return launch_r<5>(a, from_ring, s);
        case 6: return launch_r<6>(a, from_ring, s);
        case 7: return launch_r<7>(a, from_ring, s);
        case 8: return launch_r<8>(a, from_ring, s);
        default: return hipErrorInvalidValue;
    }
}

template <int R>
static hipError_t launch_many_r(const GruArgs& a, int n_updates, int n_padded, hipStream_t s) {
    const int tiles = (a.n_streams + kTileStreams - 1) / kTileStreams;
    // up to ~1.5 windows per SIMD the four-wave kernel wins (4096 streams x 4 updates: 18.1 vs 20.1 us per
    // update), from 2 per SIMD on the one-wave kernel does (x 16: 12.5 vs 14.0)
    if ((long long)tiles * n_updates <= 1536 && !a.use_delta)      // (the delta inputs: one-wave kernel only)
        hipLaunchKernelGGL((gru_many_mw_kernel<R>), dim3(tiles * n_updates), dim3(256), 0, s, a, tiles, n_padded);
    else
        hipLaunchKernelGGL((gru_many_kernel<R>), dim3(tiles * n_updates), dim3(64), 0, s, a, tiles, n_padded);
    return hipGetLastError();
}

hipError_t launch_gru_many(const GruArgs& a, int n_updates, int n_padded, hipStream_t s) {
    const int tiles = (a.n_streams + kTileStreams - 1) / kTileStreams;
    if (tiles == 0 || n_updates == 0) return hipSuccess;
    if (a.bf16) {
        hipLaunchKernelGGL(gru_many_bf16_kernel, dim3(tiles * n_updates), dim3(64), 0, s, a, tiles, n_padded);
        return hipGetLastError();
    }
    switch (gru_small_regs(a.units)) {
        case 1: return launch_many_r<1>(a, n_updates, n_padded, s);
        case 2: return launch_many_r<2>(a, n_updates, n_padded, s);
        case 3: return launch_many_r<3>(a, n_updates, n_padded, s);
        case 4: return launch_many_r<4>(a, n_updates, n_padded, s);
        case 5: return launch_many_r<5>(a, n_updates, n_padded, s);
        case 6: return launch_many_r<6>(a, n_updates, n_padded, s);
        case 7: return launch_many_r<7>(a, n_updates, n_padded, s);
        case 8: return launch_many_r<8>(a, n_updates, n_padded, s);
        default: return hipErrorInvalidValue;
    }
}

template <class R, int RG>
static hipError_t launch_fused_rg(const MfccStreamArgs<R>& m, const GruArgs& g, hipStream_t s) {
    const int tiles = (m.geo.n_streams + kTileStreams - 1) / kTileStreams;
    const size_t lds = lds_layout_bytes(sizeof(R), m.geo.n_filt, m.geo.n_mfcc, m.tab.mel_parts, 16);
    const int nsel = frame_split(m.geo, m.chunk, tiles);
    if (g.waves_per_tile == 4) {
        hipLaunchKernelGGL((fused_update_kernel<R, RG, true>), dim3(tiles + tiles * nsel), dim3(256), lds, s, m, g, tiles, tiles, nsel);
    } else {
        const int gru_blocks = (tiles + 3) / 4;
        hipLaunchKernelGGL((fused_update_kernel<R, RG, false>), dim3(gru_blocks + tiles * nsel), dim3(256), lds, s, m, g, gru_blocks, tiles, nsel);
    }
    return hipGetLastError();
}

template <class R>
static hipError_t launch_fused(const MfccStreamArgs<R>& m, const GruArgs& g, hipStream_t s) {
    if (g.bf16) {
        const int tiles = (m.geo.n_streams + kTileStreams - 1) / kTileStreams;
        const int gru_blocks = (tiles + 3) / 4;
        const size_t lds = lds_layout_bytes(sizeof(R), m.geo.n_filt, m.geo.n_mfcc, m.tab.mel_parts, 16);
        hipLaunchKernelGGL((fused_update_bf16_kernel<R>), dim3(gru_blocks + tiles), dim3(256), lds, s, m, g, gru_blocks, tiles);
        return hipGetLastError();
    }
    switch (gru_small_regs(g.units)) {
        case 1: return launch_fused_rg<R, 1>(m, g, s);
        case 2: return launch_fused_rg<R, 2>(m, g, s);
        case 3: return launch_fused_rg<R, 3>(m, g, s);
        case 4: return launch_fused_rg<R, 4>(m, g, s);
        case 5: return launch_fused_rg<R, 5>(m, g, s);
        case 6: return launch_fused_rg<R, 6>(m, g, s);
        case 7: return launch_fused_rg<R, 7>(m, g, s);
        case 8: return launch_fused_rg<R, 8>(m, g, s);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_fused_f64(const MfccStreamArgs<double>& m, const GruArgs& g, hipStream_t s) { return launch_fused<double>(m, g, s); }
hipError_t launch_fused_f32(const MfccStreamArgs<float>& m, const GruArgs& g, hipStream_t s) { return launch_fused<float>(m, g, s); }

// ---- small utility kernels ---------------------------------------------------------------------
__global__ void gather_kernel(const GatherArgs a) {
    // out[s][t][f] = ring row of frame (ke - T + t) of stream s      (Listener.mfccs, oldest first)
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)a.n_streams * a.n_features * a.n_mfcc;
    if (idx >= total) return;
    const int f = (int)(idx % a.n_mfcc);
    const int t = (int)((idx / a.n_mfcc) % a.n_features);
    const long long s = idx / ((long long)a.n_mfcc * a.n_features);
    const uint32_t slot = (a.st_ke[s] - (uint32_t)a.n_features + (uint32_t)t) & (uint32_t)(a.ring_slots - 1);
    const long long tile = s / kTileStreams;
    const int j = (int)(s % kTileStreams);
    a.out[idx] = a.ring[(((size_t)tile * a.ring_slots + slot) * kTileStreams + j) * kRowFloats + f];
}

__global__ void scatter_kernel(const GatherArgs a, int32_t* st_q, uint32_t* st_kc) {
    // inverse of gather_kernel: the stream restarts with the given [T][F] window already emitted
    // (frames 0..T-1 in slots 0..T-1, nothing held toward the next frame)
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)a.n_streams * a.n_features * kRowFloats;
    if (idx >= total) return;
    const int f = (int)(idx % kRowFloats);
    const int t = (int)((idx / kRowFloats) % a.n_features);
    const long long s = idx / ((long long)kRowFloats * a.n_features);
    const long long tile = s / kTileStreams;
    const int j = (int)(s % kTileStreams);
    const float v = f < a.n_mfcc ? a.out[(s * a.n_features + t) * a.n_mfcc + f] : 0.0f;
    const_cast<float*>(a.ring)[(((size_t)tile * a.ring_slots + t) * kTileStreams + j) * kRowFloats + f] = v;
    if (f == 0 && t == 0) {
        st_q[s] = 0;
        st_kc[s] = (uint32_t)a.n_features;
        const_cast<uint32_t*>(a.st_ke)[s] = (uint32_t)a.n_features;
    }
}

hipError_t launch_scatter(const GatherArgs& a, int32_t* st_q, uint32_t* st_kc, hipStream_t s) {
    const long long total = (long long)a.n_streams * a.n_features * kRowFloats;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(scatter_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a, st_q, st_kc);
    return hipGetLastError();
}

__global__ void clear_kernel(const ClearArgs a) {
    // one workgroup per stream: zero its counters and every ring row
    const long long s = blockIdx.x;
    if (s >= a.n_streams) return;
    if (a.mask && !a.mask[s]) return;
    if (threadIdx.x == 0) { a.st_q[s] = 0; a.st_kc[s] = 0u; a.st_ke[s] = 0u; if (a.activation) a.activation[s] = 0; }
    const long long tile = s / kTileStreams;
    const int j = (int)(s % kTileStreams);
    for (int i = threadIdx.x; i < a.ring_slots * kRowFloats; i += blockDim.x) {
        const int slot = i / kRowFloats, f = i % kRowFloats;
        a.ring[(((size_t)tile * a.ring_slots + slot) * kTileStreams + j) * kRowFloats + f] = 0.0f;
    }
}

__global__ void decode_kernel(const DecodeArgs a) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= a.n_streams) return;
    const float rawf = a.raw[s];
    const double raw = (double)rawf;
    double conf = raw;
    if (raw != 1.0 && raw != 0.0) {                       // saturated sigmoid passes through (:46-47)
        double cp;
        if (a.out_range == 0) {
            cp = raw > (double)a.min_out ? 1.0 : 0.0;
        } else {
            // asigmoid (functions.py:99-101) on the runner's float32 scalar: numpy evaluates `1 / x - 1` in
            // float32 (two correctly rounded operations), math.log then takes that value as a double
            const float odds = __fsub_rn(__fdiv_rn(1.0f, rawf), 1.0f);
            double ratio = (-log((double)odds) - (double)a.min_out) / (double)a.out_range;
            ratio = fmin(fmax(ratio, 0.0), 1.0);
            cp = a.cd[(int)(ratio * (double)(a.cd_len - 1) + 0.5)];
        }
        conf = cp < a.center ? 0.5 * cp / a.center : 0.5 + 0.5 * (cp - a.center) / (1.0 - a.center);
    }
    if (a.conf_out) a.conf_out[s] = conf;
    if (a.activation) {
        int act = a.activation[s];
        const bool hot = conf > a.threshold;
        bool fired = false;
        if (!hot && act >= 0) {
            if (act > 0) act -= 1;
        } else {
            act += 1;
            fired = act > a.trigger_level;
            if (fired || (hot && act < 0)) act = a.rearm;
        }
        a.activation[s] = act;
        if (a.fired_out) a.fired_out[s] = fired ? 1 : 0;
    }
}

hipError_t launch_decode(const DecodeArgs& a, hipStream_t s) {
    if (a.n_streams == 0) return hipSuccess;
    hipLaunchKernelGGL(decode_kernel, dim3((a.n_streams + 255) / 256), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_gather(const GatherArgs& a, hipStream_t s) {
    const long long total = (long long)a.n_streams * a.n_features * a.n_mfcc;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_clear(const ClearArgs& a, hipStream_t s) {
    if (a.n_streams == 0) return hipSuccess;
    hipLaunchKernelGGL(clear_kernel, dim3(a.n_streams), dim3(64), 0, s, a);
    return hipGetLastError();
}

#ifdef PE_SECTION_TIMERS
extern "C" int pe_debug_read_timers(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pe_dbg_timers), sizeof(unsigned long long) * (n < 32 ? n : 32));
}
#endif

}  // namespace pe
