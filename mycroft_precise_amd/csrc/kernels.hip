// Every __global__ entry point of libprecise_engine.so and its launcher.  The device code lives
// in mfcc_wave_device.h / mfcc_device.h (MFCC front end: frames / bookkeeping) and gru_*_device.h (GRU + Dense
// on the matrix cores).
#include <cstdlib>
#include <type_traits>
#include "mfcc_device.h"
#include "mfcc_wave_device.h"
#include "gru_device.h"
#include "gru_cw_device.h"
#include "gru_bf16_device.h"
#include "gru_b20_device.h"
#include "gru_x3_device.h"
#include "gru_wide_device.h"
#include "gru_wide_x3_device.h"
#include "mfcc_general_device.h"

namespace pe {

// ---- MFCC: one frame task per wave (mfcc_wave_device.h), bookkeeping per stream (mfcc_device.h) ------------------
#ifndef PE_FRAME_WPE
#define PE_FRAME_WPE 4          // waves per SIMD the frame role is compiled for (<= 128 VGPRs)
#endif
// workgroups [0, n_frame_blocks): frame tasks; the rest: one bookkeeping workgroup per tile (they read what the
// frame tasks read and write elsewhere, so the two roles share a launch)
// Every 64-byte line of a kernel's argument segment, requested by the first instructions of the kernel.  The role
// selection reads the trailing integers, waits, branches, and only then do the role's own pointers get loaded -- from
// other cache lines, after a second miss of the scalar cache (two memory round trips in series before the first vector
// load can go out; ISA, round 3).  With every line on its way behind ONE wait the later scalar loads hit.
template <int BYTES>
__device__ __forceinline__ void touch_kernel_arguments() {
    const int* ka = (const int*)__builtin_amdgcn_kernarg_segment_ptr();
    static_assert(BYTES <= 12 * 64, "more lines than this helper requests");
    auto line = [&](int i) -> int { return i * 64 < BYTES ? ka[i * 16] : 0; };
    const int v0 = line(0), v1 = line(1), v2 = line(2), v3 = line(3), v4 = line(4), v5 = line(5), v6 = line(6), v7 = line(7),
              v8 = line(8), v9 = line(9), v10 = line(10), v11 = line(11);
    asm volatile("" :: "s"(v0), "s"(v1), "s"(v2), "s"(v3), "s"(v4), "s"(v5), "s"(v6), "s"(v7), "s"(v8), "s"(v9), "s"(v10), "s"(v11));
}

// Every kernel that hosts the FLOAT32 frame role exists twice: `name` (R = double) and `name_nopk` (R = float), the second
// compiled without packed float32 instructions (v_pk_add/mul/fma_f32).  Why: round 4 found ~0.7 % of the float32 frames of
// a fused launch slightly wrong, timing-dependent, beside the five-values bf16 network role (gru_b20_device.h;
// profiles/round4/r4v_b20_fused_corruption.log); the one change that cured it in every run (4 / 4 in round 4, 2 / 2 and the
// 1e8-frame soaks in round 5) is this one, and it is also FASTER: hipcc's SLP pass packs the butterflies' additions, and on
// gfx950 packed float32 shares the XDL datapath (tools/micro/pipe_overlap.hip; MI355X_MICROARCH: "an anti-lever beside MFMAs")
// -- MFCC launch 51.3 vs 52.9 us at 65 536 streams, fused bf16 update 73.1 vs 75.6 us (profiles/round5/r5a_*).  The attribute
// applies to the whole kernel (code generation is per function), which is why it cannot sit on the frame function itself.
#if defined(__HIP_DEVICE_COMPILE__)
#define PE_NO_PK_F32 __attribute__((target("no-packed-fp32-ops")))
#else
#define PE_NO_PK_F32            // (the host pass only sees the launch stubs)
#endif
#define PE_UNPAREN(...) __VA_ARGS__
// launch KERNEL<R, TARGS...> -- its _nopk twin when R is float
#define PE_LAUNCH_R(R, KERNEL, TARGS, ...)                                                                   \
    do {                                                                                                     \
        if constexpr (std::is_same<R, float>::value) hipLaunchKernelGGL((KERNEL##_nopk<R, PE_UNPAREN TARGS>), __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<R, PE_UNPAREN TARGS>), __VA_ARGS__);                                 \
    } while (0)

// SINGLE: one update, no projection rows (the launcher knows): the frame loop without the several-updates arithmetic
template <class R, class SH, bool SINGLE>
__device__ __forceinline__ void mfcc_kernel_body(const MfccStreamArgs<R>& a, const WaveTables<R>& t, const int n_frame_blocks) {
    touch_kernel_arguments<(int)(sizeof(MfccStreamArgs<R>) + sizeof(WaveTables<R>) + 4)>();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if ((int)blockIdx.x < n_frame_blocks) mfcc_frame_tasks<R, SH, SINGLE, SINGLE>(a, t, smem, (int)blockIdx.x * kFrameWaves, n_frame_blocks * kFrameWaves);
    else mfcc_book_tile<R>(a, (int)blockIdx.x - n_frame_blocks);
}
template <class R, class SH, bool SINGLE = false>
__global__ __launch_bounds__(64 * kFrameWaves) __attribute__((amdgpu_waves_per_eu(SH::WPE))) void mfcc_kernel(const MfccStreamArgs<R> a, const WaveTables<R> t, const int n_frame_blocks) {
    mfcc_kernel_body<R, SH, SINGLE>(a, t, n_frame_blocks);
}
template <class R, class SH, bool SINGLE = false>
__global__ __launch_bounds__(64 * kFrameWaves) __attribute__((amdgpu_waves_per_eu(SH::WPE))) PE_NO_PK_F32 void mfcc_kernel_nopk(const MfccStreamArgs<R> a, const WaveTables<R> t, const int n_frame_blocks) {
    mfcc_kernel_body<R, SH, SINGLE>(a, t, n_frame_blocks);
}


// network for a whole batch of updates: workgroup (one wave) b serves update b / n_tiles, tile b % n_tiles
template <int R, bool PROJ>
__global__ __launch_bounds__(64) void gru_many_kernel(const GruArgs a, const int n_tiles, const int n_padded) {
    const int u = blockIdx.x / n_tiles, tile = blockIdx.x % n_tiles;
    GruArgs b = a;
    b.ke_plain = a.ke_plain + (size_t)u * n_padded;      // row u of the emitted-frame history
    b.out = a.out + (size_t)u * a.n_streams;
    b.predict_ke = 0;
    gru_tile<R, kRing, PROJ>(b, tile, threadIdx.x);
}

// same, four waves sharing each (update, tile) -- few tiles per SIMD: latency matters more than issue slots
template <int R, bool PROJ>
__global__ __launch_bounds__(256) void gru_many_mw_kernel(const GruArgs a, const int n_tiles, const int n_padded) {
    __shared__ __attribute__((aligned(16))) float S[3 * R * 64 + 256];
    const int u = blockIdx.x / n_tiles, tile = blockIdx.x % n_tiles;
    GruArgs b = a;
    b.ke_plain = a.ke_plain + (size_t)u * n_padded;      // row u of the emitted-frame history
    b.out = a.out + (size_t)u * a.n_streams;
    b.predict_ke = 0;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    gru_tile_mw_any<R, PROJ>(b, tile, wave, threadIdx.x & 63, S);
}

// the same two for the re-tiled stock width (gru_cw_device.h)
template <bool DELTA>
__global__ __launch_bounds__(64) void gru_many_v_kernel(const GruArgs a, const int n_tiles, const int n_padded) {
    const int u = blockIdx.x / n_tiles, tile = blockIdx.x % n_tiles;
    GruArgs b = a;
    b.ke_plain = a.ke_plain + (size_t)u * n_padded;      // row u of the emitted-frame history
    b.out = a.out + (size_t)u * a.n_streams;
    b.predict_ke = 0;
    gru_tile_v<kRing, DELTA>(b, tile, threadIdx.x);
}
__global__ __launch_bounds__(256) void gru_many_cw_kernel(const GruArgs a, const int n_tiles, const int n_padded) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int u = blockIdx.x / n_tiles, tile = blockIdx.x % n_tiles;
    GruArgs b = a;
    b.ke_plain = a.ke_plain + (size_t)u * n_padded;      // row u of the emitted-frame history
    b.out = a.out + (size_t)u * a.n_streams;
    b.predict_ke = 0;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    gru_tile_cw<false>(b, tile, wave, threadIdx.x & 63, reinterpret_cast<float*>(smem));
}

template <bool DELTA, bool RB>
__global__ __launch_bounds__(64) void gru_many_bf16_kernel(const GruArgs a, const int n_tiles, const int n_padded) {
    const int u = blockIdx.x / n_tiles, tile = blockIdx.x % n_tiles;
    GruArgs b = a;
    b.ke_plain = a.ke_plain + (size_t)u * n_padded;      // row u of the emitted-frame history
    b.out = a.out + (size_t)u * a.n_streams;
    b.predict_ke = 0;
    if (b.b20) { gru_tile_b20<kRing, DELTA, RB>(b, tile, threadIdx.x); return; }      // <= 20 units: five values per lane
    gru_tile_bf16<kRing, DELTA, RB>(b, tile, threadIdx.x);
}

template <class R, class SH>
__global__ __launch_bounds__(64 * kFrameWaves) __attribute__((amdgpu_waves_per_eu(SH::WPE))) void mfcc_offline_kernel(const MfccOfflineArgs<R> a, const WaveTables<R> t) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    mfcc_offline_frames<R, SH>(a, t, smem);
}
template <class R, class SH>
__global__ __launch_bounds__(64 * kFrameWaves) __attribute__((amdgpu_waves_per_eu(SH::WPE))) PE_NO_PK_F32 void mfcc_offline_kernel_nopk(const MfccOfflineArgs<R> a, const WaveTables<R> t) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    mfcc_offline_frames<R, SH>(a, t, smem);
}

// ---- GRU: one wave per 16-stream tile ----------------------------------------------------------
template <int R, int MODE, bool PROJ = false, int KX = 1>
__global__ __launch_bounds__(64) void gru_small_kernel(const GruArgs a) {
    touch_kernel_arguments<(int)sizeof(GruArgs)>();
    gru_tile<R, MODE, PROJ, KX>(a, blockIdx.x, threadIdx.x);
}

// ---- GRU, stock width re-tiled (gru_cw_device.h): one wave per tile / four waves per tile ------------------------
template <int MODE, bool DELTA>
__global__ __launch_bounds__(64) void gru_v_kernel(const GruArgs a) {
    touch_kernel_arguments<(int)sizeof(GruArgs)>();
    gru_tile_v<MODE, DELTA>(a, blockIdx.x, threadIdx.x);
}
__global__ __launch_bounds__(256) void gru_cw_kernel(const GruArgs a) {
    touch_kernel_arguments<(int)sizeof(GruArgs)>();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    gru_tile_cw<false>(a, blockIdx.x, wave, threadIdx.x & 63, reinterpret_cast<float*>(smem));
}
// the four-wave shape stages the tile's whole ring in LDS and walks at most 32 timesteps of it
static bool cw_four_waves_ok(const GruArgs& a) { return a.ring_slots == kCwSlots && a.n_features <= kCwSlots; }

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

// ---- the same network with the float32 products on the bf16 matrix pipe (gru_wide_x3_device.h; pe_set_gru_tiling(e, 2)) ----
#ifndef PE_WIDE_X3_KS
#define PE_WIDE_X3_KS 2         // 2: eight waves per workgroup, the k-groups of every contraction cut in two (widths that are multiples of 128)
#endif
template <int TPW, int MODE, int KS>
__global__ __launch_bounds__(256 * KS) void gru_wide_x3_kernel(const WideArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    gru_wide_x3_tile<TPW, MODE, KS>(a, blockIdx.x, wave, threadIdx.x & 63, smem);
}

template <int TPW>
static hipError_t launch_wide_x3_t(const WideArgs& a, int mode, hipStream_t s) {
    const int tiles = (a.base.n_streams + kTileStreams - 1) / kTileStreams;
    if (tiles == 0) return hipSuccess;
    constexpr int KS = (PE_WIDE_X3_KS == 2 && TPW % 2 == 0) ? 2 : 1;
    const size_t lds = wide_x3_lds_bytes<TPW>();          // three state vectors + partial sums / z / float32 state of the lead waves
    if (mode == kRing) hipLaunchKernelGGL((gru_wide_x3_kernel<TPW, kRing, KS>), dim3(tiles), dim3(256 * KS), lds, s, a);
    else if (mode == kRows) hipLaunchKernelGGL((gru_wide_x3_kernel<TPW, kRows, KS>), dim3(tiles), dim3(256 * KS), lds, s, a);
    else hipLaunchKernelGGL((gru_wide_x3_kernel<TPW, kFeats, KS>), dim3(tiles), dim3(256 * KS), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_gru_wide_x3(const WideArgs& a, int mode, hipStream_t s) {
    switch (a.units / 64) {
        case 1: return launch_wide_x3_t<1>(a, mode, s);
        case 2: return launch_wide_x3_t<2>(a, mode, s);
        case 3: return launch_wide_x3_t<3>(a, mode, s);
        case 4: return launch_wide_x3_t<4>(a, mode, s);
        default: return hipErrorInvalidValue;
    }
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
template <int MODE, bool DELTA, bool RB = false>
__global__ __launch_bounds__(64) void gru_bf16_kernel(const GruArgs a) {
    touch_kernel_arguments<(int)sizeof(GruArgs)>();
    if (a.b20) { gru_tile_b20<MODE, DELTA, RB>(a, blockIdx.x, threadIdx.x); return; }      // <= 20 units: five values per lane (gru_b20_device.h)
    gru_tile_bf16<MODE, DELTA, RB>(a, blockIdx.x, threadIdx.x);
}

// ---- GRU, float32 as three bf16 pieces per operand on the XDL pipe: one wave per 16-stream tile ------------------
template <int MODE>
__global__ __launch_bounds__(64) void gru_x3_kernel(const GruArgs a) {
    touch_kernel_arguments<(int)sizeof(GruArgs)>();
    gru_tile_x3<MODE>(a, blockIdx.x, threadIdx.x);
}
__global__ __launch_bounds__(64) void gru_many_x3_kernel(const GruArgs a, const int n_tiles, const int n_padded) {
    const int u = blockIdx.x / n_tiles, tile = blockIdx.x % n_tiles;
    GruArgs b = a;
    b.ke_plain = a.ke_plain + (size_t)u * n_padded;      // row u of the emitted-frame history
    b.out = a.out + (size_t)u * a.n_streams;
    b.predict_ke = 0;
    gru_tile_x3<kRing>(b, tile, threadIdx.x);
}

// Dispatch order of the roles of a fused launch.  Workgroups are handed to the CUs in blockIdx order; `frames_first`
// puts the (few, long-lived, VALU-bound) frame workgroups in front of the (many, MFMA-bound) network workgroups so
// that at large batches the two kinds are resident TOGETHER -- with the network first, its workgroups fill every
// wave slot and the frame role only starts when they drain (the launch then costs the SUM of the two roles).
// Returns the index in the canonical order [network | frames | bookkeeping].
constexpr int kFramesFirst = 1, kBySimd = 2;       // launch flags of fused_update_kernel (its last argument)
constexpr int kCwRoleSlot = 1984;                  // four ints of the GRU workgroup's LDS between the mailboxes and the staged ring
static_assert(CwBox::END <= kCwRoleSlot && kCwRoleSlot + 4 <= CwLds::XR, "role slots overlap");
__device__ __forceinline__ int role_block(const int b, const int n_gru, const int n_frames, const int frames_first) {
    if (!frames_first || b >= n_gru + n_frames) return b;
    return b < n_frames ? n_gru + b : b - n_frames;
}

// fused update with the bf16 network role (four tiles per GRU workgroup, one wave each)
template <class R, class SH, bool DELTA, bool RB>
__device__ __forceinline__ void fused_update_bf16_body(const MfccStreamArgs<R>& m, const WaveTables<R>& t, const GruArgs& g,
                                                       const int n_gru_blocks, const int n_frame_blocks, const int n_tiles, const int frames_first) {
    touch_kernel_arguments<(int)(sizeof(MfccStreamArgs<R>) + sizeof(WaveTables<R>) + sizeof(GruArgs) + 16)>();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // frames_first: bit 0 = frame workgroups dispatched first; bits 8.. = network tiles per workgroup (1, 2 or 4: with few
    // tiles, one or two network waves on EVERY compute unit disturb the frame waves less than four on every second one)
    const int tpw = frames_first >> 8;
    const int b = role_block(blockIdx.x, n_gru_blocks, n_frame_blocks, frames_first & 1);
    if (b < n_gru_blocks) {
        const int wave = threadIdx.x >> 6;
        const int tile = b * tpw + wave;
        if (wave < tpw && tile < n_tiles) {
            if (g.b20) { gru_tile_b20<kRing, DELTA, RB>(g, tile, threadIdx.x & 63); return; }
            gru_tile_bf16<kRing, DELTA, RB>(g, tile, threadIdx.x & 63);
        }
    } else if (b < n_gru_blocks + n_frame_blocks) {
        mfcc_frame_tasks<R, SH, true>(m, t, smem, (b - n_gru_blocks) * kFrameWaves, n_frame_blocks * kFrameWaves);
    } else {
        mfcc_book_tile<R>(m, b - n_gru_blocks - n_frame_blocks);
    }
}
template <class R, class SH, bool DELTA, bool RB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PE_FRAME_WPE))) void fused_update_bf16_kernel(const MfccStreamArgs<R> m, const WaveTables<R> t, const GruArgs g,
                                                                const int n_gru_blocks, const int n_frame_blocks, const int n_tiles, const int frames_first) {
    fused_update_bf16_body<R, SH, DELTA, RB>(m, t, g, n_gru_blocks, n_frame_blocks, n_tiles, frames_first);
}
template <class R, class SH, bool DELTA, bool RB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PE_FRAME_WPE))) PE_NO_PK_F32 void fused_update_bf16_kernel_nopk(const MfccStreamArgs<R> m, const WaveTables<R> t, const GruArgs g,
                                                                const int n_gru_blocks, const int n_frame_blocks, const int n_tiles, const int frames_first) {
    fused_update_bf16_body<R, SH, DELTA, RB>(m, t, g, n_gru_blocks, n_frame_blocks, n_tiles, frames_first);
}

// ---- GRU: four waves per 16-stream tile (few tiles: fills all four SIMDs of a CU) -----------------
template <int R, bool PROJ = false, int KX = 1>
__global__ __launch_bounds__(256) void gru_mw_kernel(const GruArgs a) {
    __shared__ __attribute__((aligned(16))) float S[3 * R * 64 + 256];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    gru_tile_mw_any<R, PROJ, KX>(a, blockIdx.x, wave, threadIdx.x & 63, S);
}


// ---- fused update: GRU role || MFCC frame role || bookkeeping role in ONE launch -------------------------------
// Workgroups [0, n_gru_blocks) run the network on the feature windows as they will stand after this update (they
// are dispatched first: the long pole); the next n_frame_blocks compute this update's MFCC frames, one frame task
// per wave; the last n_tiles move the leftover samples and the counters.  MW = true: one GRU workgroup per tile,
// its four waves share the tile (gru_tile_mw); MW = false: four tiles per GRU workgroup, one wave each.
template <class R, class SH, int RG, bool MW, bool PROJ, bool CW>
__device__ __forceinline__ void fused_update_body(const MfccStreamArgs<R>& m, const WaveTables<R>& t, const GruArgs& g,
                                                  const int n_gru_blocks, const int n_frame_blocks, const int n_tiles, const int frames_first) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    touch_kernel_arguments<(int)(sizeof(MfccStreamArgs<R>) + sizeof(WaveTables<R>) + sizeof(GruArgs) + 16)>();
    const int b = role_block(blockIdx.x, n_gru_blocks, n_frame_blocks, frames_first & kFramesFirst);
    if (b < n_gru_blocks) {
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#if defined(PE_PRIO_R)
        if (wave == 0) __builtin_amdgcn_s_setprio(PE_PRIO_R); else __builtin_amdgcn_s_setprio(PE_PRIO_H);
#else
        __builtin_amdgcn_s_setprio(3);          // the network role is the long pole: it wins every issue arbitration
#endif
        if constexpr (CW) {                     // stock width, re-tiled (gru_cw_device.h)
            static_assert(RG == 5 && !PROJ, "the re-tiled shapes exist for the stock width, without projection rows");
            if (MW) {
                // kBySimd: the four roles sit on SIMDs 0..3 in a fixed order (R on 0, Z1, Z2, P) so that the frame waves of
                // this compute unit know what runs beside them (mfcc_frame_tasks(..., by_simd)).  Roles follow the SIMDs only
                // if the four waves sit on four different ones (they do: a workgroup's waves are spread round-robin;
                // checked, not assumed).
                int role = wave;
                if (frames_first & kBySimd) {
                    int* const slot = reinterpret_cast<int*>(smem) + kCwRoleSlot;
                    const int simd = wave_simd_id();
                    if ((threadIdx.x & 63) == 0) slot[wave] = simd;
                    __syncthreads();
                    if (((1 << slot[0]) | (1 << slot[1]) | (1 << slot[2]) | (1 << slot[3])) == 15) role = simd;
                }
                gru_tile_cw<false>(g, b, role, threadIdx.x & 63, reinterpret_cast<float*>(smem));
            } else {
                const int tile = b * 4 + wave;
                if (tile < n_tiles) gru_tile_v<kRing, false>(g, tile, threadIdx.x & 63);       // (use_delta on this shape: two launches, engine.hip can_fuse)
            }
        } else if (MW) {
            gru_tile_mw_any<RG, PROJ>(g, b, wave, threadIdx.x & 63, reinterpret_cast<float*>(smem));
        } else {
            const int tile = b * 4 + wave;
            if (tile < n_tiles) gru_tile<RG, kRing, PROJ>(g, tile, threadIdx.x & 63);
        }
    } else if (b < n_gru_blocks + n_frame_blocks) {
        mfcc_frame_tasks<R, SH, true, !PROJ>(m, t, smem, (b - n_gru_blocks) * kFrameWaves, n_frame_blocks * kFrameWaves, CW && MW && (frames_first & kBySimd));
    } else {
        mfcc_book_tile<R>(m, b - n_gru_blocks - n_frame_blocks);
    }
}
template <class R, class SH, int RG, bool MW, bool PROJ, bool CW = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PE_FRAME_WPE))) void fused_update_kernel(const MfccStreamArgs<R> m, const WaveTables<R> t, const GruArgs g,
                                                           const int n_gru_blocks, const int n_frame_blocks, const int n_tiles, const int frames_first) {
    fused_update_body<R, SH, RG, MW, PROJ, CW>(m, t, g, n_gru_blocks, n_frame_blocks, n_tiles, frames_first);
}
// (R = float: the float32 network role inside loses its packed gate arithmetic too -- 16.2 vs 15.7 us per fused update at 4096
//  streams for the float32 front end + float32 network, which is no BASELINE configuration; the headline kernel is R = double)
template <class R, class SH, int RG, bool MW, bool PROJ, bool CW = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PE_FRAME_WPE))) PE_NO_PK_F32 void fused_update_kernel_nopk(const MfccStreamArgs<R> m, const WaveTables<R> t, const GruArgs g,
                                                           const int n_gru_blocks, const int n_frame_blocks, const int n_tiles, const int frames_first) {
    fused_update_body<R, SH, RG, MW, PROJ, CW>(m, t, g, n_gru_blocks, n_frame_blocks, n_tiles, frames_first);
}


// Workgroups of the frame role: one wave per task while that fits the machine (4 workgroups of 4 waves per compute
// unit are resident: LDS and a 128-register budget), more tasks per wave beyond.
static int frame_blocks(long long n_tasks, int n_cus, int per_cu = 4) {      // per_cu: resident frame workgroups per compute unit
    const long long need = (n_tasks + kFrameWaves - 1) / kFrameWaves;
    const long long cap = (long long)n_cus * per_cu;
    return (int)(need < cap ? (need < 1 ? 1 : need) : cap);
}

// frame workgroups of a streaming launch: every wave owns a contiguous run of (stream, row parity) slots
// (mfcc_frame_tasks), as many waves as the resident cap allows, the runs as even as they can be
static int stream_frame_blocks(int n_streams, int n_cus, int per_cu_default = 4) {
    const long long slots = 2LL * n_streams;
    const long long cap_waves = (long long)frame_blocks(slots, n_cus, per_cu_default) * kFrameWaves;
    const long long per_wave = (slots + cap_waves - 1) / cap_waves;
    const long long waves = (slots + per_wave - 1) / per_wave;
    return (int)((waves + kFrameWaves - 1) / kFrameWaves);
}

template <class R>
static size_t frame_lds(const WaveTables<R>& t) { return wave_lds_bytes(sizeof(R), t.L, kFrameWaves); }
// the kernels address the table image with the compile-time section offsets of their shape (mfcc_wave_device.h: wave_bind)
template <class R>
static bool blob_matches_shape(const WaveTables<R>& t) {
    return t.L.mel_pad == ShapeStock::MEL ? shape_layout_matches<R, ShapeStock>(t.L) : shape_layout_matches<R, ShapeAny>(t.L);
}

template <class R>
static hipError_t launch_mfcc(const MfccStreamArgs<R>& a, const WaveTables<R>& t, int n_cus, hipStream_t s) {
    const int tiles = (a.geo.n_streams + kTileStreams - 1) / kTileStreams;
    const int fb = stream_frame_blocks(a.geo.n_streams, n_cus);
    if (!blob_matches_shape(t)) return hipErrorInvalidValue;
    const bool single = a.n_updates == 1 && !a.proj_ring;
    if (t.L.mel_pad == ShapeStock::MEL) {
        if (single) PE_LAUNCH_R(R, mfcc_kernel, (ShapeStock, true), dim3(fb + tiles), dim3(64 * kFrameWaves), frame_lds(t), s, a, t, fb);
        else PE_LAUNCH_R(R, mfcc_kernel, (ShapeStock), dim3(fb + tiles), dim3(64 * kFrameWaves), frame_lds(t), s, a, t, fb);
    } else {
        if (single) PE_LAUNCH_R(R, mfcc_kernel, (ShapeAny, true), dim3(fb + tiles), dim3(64 * kFrameWaves), frame_lds(t), s, a, t, fb);
        else PE_LAUNCH_R(R, mfcc_kernel, (ShapeAny), dim3(fb + tiles), dim3(64 * kFrameWaves), frame_lds(t), s, a, t, fb);
    }
    return hipGetLastError();
}
// four frames per wave: the frame role of ONE update of a stock-shape engine whose geometry allows dword sample pairs
hipError_t launch_mfcc_f64(const MfccStreamArgs<double>& a, const WaveTables<double>& t, int n_cus, hipStream_t s) { return launch_mfcc<double>(a, t, n_cus, s); }
hipError_t launch_mfcc_f32(const MfccStreamArgs<float>& a, const WaveTables<float>& t, int n_cus, hipStream_t s) { return launch_mfcc<float>(a, t, n_cus, s); }

template <class R>
static hipError_t launch_offline(const MfccOfflineArgs<R>& a, const WaveTables<R>& t, int n_cus, hipStream_t s) {
    if (a.n_frames <= 0) return hipSuccess;
    if (!blob_matches_shape(t)) return hipErrorInvalidValue;
    if (t.L.mel_pad == ShapeStock::MEL) PE_LAUNCH_R(R, mfcc_offline_kernel, (ShapeStock), dim3(frame_blocks(a.n_frames, n_cus)), dim3(64 * kFrameWaves), frame_lds(t), s, a, t);
    else PE_LAUNCH_R(R, mfcc_offline_kernel, (ShapeAny), dim3(frame_blocks(a.n_frames, n_cus)), dim3(64 * kFrameWaves), frame_lds(t), s, a, t);
    return hipGetLastError();
}
hipError_t launch_mfcc_offline_f64(const MfccOfflineArgs<double>& a, const WaveTables<double>& t, int n_cus, hipStream_t s) { return launch_offline<double>(a, t, n_cus, s); }
hipError_t launch_mfcc_offline_f32(const MfccOfflineArgs<float>& a, const WaveTables<float>& t, int n_cus, hipStream_t s) { return launch_offline<float>(a, t, n_cus, s); }

int gru_small_regs(int units) { return (units + 3) / 4; }
int gru_small_tiles(int units) { return (3 * gru_small_regs(units) + 3) / 4; }

template <int R>
static hipError_t launch_r(const GruArgs& a, int mode, hipStream_t s) {
    const int tiles = (a.n_streams + kTileStreams - 1) / kTileStreams;
    if (tiles == 0) return hipSuccess;
    if (a.row_floats == 2 * kRowFloats) {           // 17..32 coefficients per frame: 32-float rows
        if constexpr (R == 5) {                     // (stock width, few tiles: four waves per tile, as the 16-float rows get)
            if (mode == kRing && a.waves_per_tile == 4) {
                hipLaunchKernelGGL((gru_mw_kernel<R, false, 2>), dim3(tiles), dim3(256), 0, s, a);
                return hipGetLastError();
            }
        }
        if (mode == kRing) hipLaunchKernelGGL((gru_small_kernel<R, kRing, false, 2>), dim3(tiles), dim3(64), 0, s, a);
        else if (mode == kRows) hipLaunchKernelGGL((gru_small_kernel<R, kRows, false, 2>), dim3(tiles), dim3(64), 0, s, a);
        else hipLaunchKernelGGL((gru_small_kernel<R, kFeats, false, 2>), dim3(tiles), dim3(64), 0, s, a);
        return hipGetLastError();
    }
    if constexpr (R == 5) {
        if (a.cw) {
            if (mode == kRing && a.waves_per_tile == 4 && cw_four_waves_ok(a)) hipLaunchKernelGGL(gru_cw_kernel, dim3(tiles), dim3(256), kCwLdsBytes, s, a);
            else if (a.use_delta) {
                if (mode == kRing) hipLaunchKernelGGL((gru_v_kernel<kRing, true>), dim3(tiles), dim3(64), 0, s, a);
                else if (mode == kRows) hipLaunchKernelGGL((gru_v_kernel<kRows, true>), dim3(tiles), dim3(64), 0, s, a);
                else hipLaunchKernelGGL((gru_v_kernel<kFeats, true>), dim3(tiles), dim3(64), 0, s, a);
            }
            else if (mode == kRing) hipLaunchKernelGGL((gru_v_kernel<kRing, false>), dim3(tiles), dim3(64), 0, s, a);
            else if (mode == kRows) hipLaunchKernelGGL((gru_v_kernel<kRows, false>), dim3(tiles), dim3(64), 0, s, a);
            else hipLaunchKernelGGL((gru_v_kernel<kFeats, false>), dim3(tiles), dim3(64), 0, s, a);
            return hipGetLastError();
        }
        if (mode == kRing && a.proj_ring) {
            if (a.waves_per_tile == 4) hipLaunchKernelGGL((gru_mw_kernel<R, true>), dim3(tiles), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((gru_small_kernel<R, kRing, true>), dim3(tiles), dim3(64), 0, s, a);
            return hipGetLastError();
        }
    }
    if (mode == kRing && a.waves_per_tile == 4) hipLaunchKernelGGL((gru_mw_kernel<R>), dim3(tiles), dim3(256), 0, s, a);
    else if (mode == kRing) hipLaunchKernelGGL((gru_small_kernel<R, kRing>), dim3(tiles), dim3(64), 0, s, a);
    else if (mode == kRows) hipLaunchKernelGGL((gru_small_kernel<R, kRows>), dim3(tiles), dim3(64), 0, s, a);
    else hipLaunchKernelGGL((gru_small_kernel<R, kFeats>), dim3(tiles), dim3(64), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_gru_small(const GruArgs& a, int from_ring, hipStream_t s) {
    if (a.x3) {
        const int tiles = (a.n_streams + kTileStreams - 1) / kTileStreams;
        if (tiles == 0) return hipSuccess;
        if (from_ring == kRing) hipLaunchKernelGGL(gru_x3_kernel<kRing>, dim3(tiles), dim3(64), 0, s, a);
        else if (from_ring == kRows) hipLaunchKernelGGL(gru_x3_kernel<kRows>, dim3(tiles), dim3(64), 0, s, a);
        else hipLaunchKernelGGL(gru_x3_kernel<kFeats>, dim3(tiles), dim3(64), 0, s, a);
        return hipGetLastError();
    }
    if (a.bf16) {
        const int tiles = (a.n_streams + kTileStreams - 1) / kTileStreams;
        if (tiles == 0) return hipSuccess;
        if (a.use_delta) {
            if (from_ring == kRing && a.ring_bf16) hipLaunchKernelGGL((gru_bf16_kernel<kRing, true, true>), dim3(tiles), dim3(64), 0, s, a);
            else if (from_ring == kRing) hipLaunchKernelGGL((gru_bf16_kernel<kRing, true>), dim3(tiles), dim3(64), 0, s, a);
            else if (from_ring == kRows) hipLaunchKernelGGL((gru_bf16_kernel<kRows, true>), dim3(tiles), dim3(64), 0, s, a);
            else hipLaunchKernelGGL((gru_bf16_kernel<kFeats, true>), dim3(tiles), dim3(64), 0, s, a);
        } else {
            if (from_ring == kRing && a.ring_bf16) hipLaunchKernelGGL((gru_bf16_kernel<kRing, false, true>), dim3(tiles), dim3(64), 0, s, a);
            else if (from_ring == kRing) hipLaunchKernelGGL((gru_bf16_kernel<kRing, false>), dim3(tiles), dim3(64), 0, s, a);
            else if (from_ring == kRows) hipLaunchKernelGGL((gru_bf16_kernel<kRows, false>), dim3(tiles), dim3(64), 0, s, a);
            else hipLaunchKernelGGL((gru_bf16_kernel<kFeats, false>), dim3(tiles), dim3(64), 0, s, a);
        }
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
    const bool few = (long long)tiles * n_updates <= 1536;
    const bool mw = few && !a.use_delta;       // (classic tiling: the delta inputs are on the one-wave kernel only)
    if constexpr (R == 5) {
        if (a.cw) {
            if (few && cw_four_waves_ok(a)) hipLaunchKernelGGL(gru_many_cw_kernel, dim3(tiles * n_updates), dim3(256), kCwLdsBytes, s, a, tiles, n_padded);
            else if (a.use_delta) hipLaunchKernelGGL(gru_many_v_kernel<true>, dim3(tiles * n_updates), dim3(64), 0, s, a, tiles, n_padded);
            else hipLaunchKernelGGL(gru_many_v_kernel<false>, dim3(tiles * n_updates), dim3(64), 0, s, a, tiles, n_padded);
            return hipGetLastError();
        }
        if (a.proj_ring) {
            if (mw) hipLaunchKernelGGL((gru_many_mw_kernel<R, true>), dim3(tiles * n_updates), dim3(256), 0, s, a, tiles, n_padded);
            else hipLaunchKernelGGL((gru_many_kernel<R, true>), dim3(tiles * n_updates), dim3(64), 0, s, a, tiles, n_padded);
            return hipGetLastError();
        }
    }
    if (mw) hipLaunchKernelGGL((gru_many_mw_kernel<R, false>), dim3(tiles * n_updates), dim3(256), 0, s, a, tiles, n_padded);
    else hipLaunchKernelGGL((gru_many_kernel<R, false>), dim3(tiles * n_updates), dim3(64), 0, s, a, tiles, n_padded);
    return hipGetLastError();
}

hipError_t launch_gru_many(const GruArgs& a, int n_updates, int n_padded, hipStream_t s) {
    const int tiles = (a.n_streams + kTileStreams - 1) / kTileStreams;
    if (tiles == 0 || n_updates == 0) return hipSuccess;
    if (a.x3) {
        hipLaunchKernelGGL(gru_many_x3_kernel, dim3(tiles * n_updates), dim3(64), 0, s, a, tiles, n_padded);
        return hipGetLastError();
    }
    if (a.bf16) {
        if (a.use_delta && a.ring_bf16) hipLaunchKernelGGL((gru_many_bf16_kernel<true, true>), dim3(tiles * n_updates), dim3(64), 0, s, a, tiles, n_padded);
        else if (a.use_delta) hipLaunchKernelGGL((gru_many_bf16_kernel<true, false>), dim3(tiles * n_updates), dim3(64), 0, s, a, tiles, n_padded);
        else if (a.ring_bf16) hipLaunchKernelGGL((gru_many_bf16_kernel<false, true>), dim3(tiles * n_updates), dim3(64), 0, s, a, tiles, n_padded);
        else hipLaunchKernelGGL((gru_many_bf16_kernel<false, false>), dim3(tiles * n_updates), dim3(64), 0, s, a, tiles, n_padded);
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
static hipError_t launch_fused_rg(const MfccStreamArgs<R>& m, const WaveTables<R>& t, const GruArgs& g, int n_cus, hipStream_t s) {
    const int tiles = (m.geo.n_streams + kTileStreams - 1) / kTileStreams;
    const size_t lds = frame_lds(t);
    // more network workgroups than the machine holds at once: frames first, three frame workgroups per CU, the network
    // streams through the remaining wave slots (measured at 16384 / 65536 streams, bf16 network + float32 front end:
    // 31.8 / 103.8 us against 35.5 / 109.0 us network-first; the float64 front end gains nothing either way -- its
    // FP64 multiply-adds and the MFMAs do not overlap on a SIMD)
    // one network tile per compute unit on the critical-wave kernel: roles by SIMD, frame slots split by SIMD load
    const bool by_simd = g.cw && g.waves_per_tile == 4 && cw_four_waves_ok(g) && tiles <= n_cus;
    const int frames_first = ((g.waves_per_tile != 4 && tiles >= 4 * n_cus) ? 1 : 0) | (by_simd ? kBySimd : 0);
    // resident frame workgroups per compute unit: at one network tile per compute unit the launch lasts as long as the
    // network's dependent chain, and two frame workgroups (two streams per wave, the second one's samples prefetched)
    // disturb that chain less than four (measured, 4096 streams: 20.6 vs 20.9 us in phase, 21.1 vs 22.5 us with
    // desynchronised streams); larger batches want every wave slot
    const int fb = stream_frame_blocks(m.geo.n_streams, n_cus, tiles <= n_cus ? 2 : (frames_first & kFramesFirst) ? 3 : 4);
    const int gru_blocks = g.waves_per_tile == 4 ? tiles : (tiles + 3) / 4;
    const int fb_ = fb, book = tiles;
    const dim3 grid(gru_blocks + fb_ + book);
    if constexpr (RG == 5) {
        if (g.cw) {                          // stock width, re-tiled: the four-wave shape wants its LDS (mailboxes + staged ring)
            if (g.waves_per_tile == 4 && cw_four_waves_ok(g)) {
                PE_LAUNCH_R(R, fused_update_kernel, (ShapeStock, RG, true, false, true), grid, dim3(256), lds > kCwLdsBytes ? lds : kCwLdsBytes, s, m, t, g, gru_blocks, fb_, tiles, frames_first);
            } else {
                const int gb = (tiles + 3) / 4;
                PE_LAUNCH_R(R, fused_update_kernel, (ShapeStock, RG, false, false, true), dim3(gb + fb_ + book), dim3(256), lds, s, m, t, g, gb, fb_, tiles, frames_first);
            }
            return hipGetLastError();
        }
    }
    if constexpr (RG == 5) {                 // (projection rows exist for the stock width only)
        if (g.proj_ring) {
            if (g.waves_per_tile == 4) PE_LAUNCH_R(R, fused_update_kernel, (ShapeStock, RG, true, true), grid, dim3(256), lds, s, m, t, g, gru_blocks, fb_, tiles, frames_first);
            else PE_LAUNCH_R(R, fused_update_kernel, (ShapeStock, RG, false, true), grid, dim3(256), lds, s, m, t, g, gru_blocks, fb_, tiles, frames_first);
            return hipGetLastError();
        }
    }
    if (g.waves_per_tile == 4) PE_LAUNCH_R(R, fused_update_kernel, (ShapeStock, RG, true, false), grid, dim3(256), lds, s, m, t, g, gru_blocks, fb_, tiles, frames_first);
    else if constexpr (RG <= 5) PE_LAUNCH_R(R, fused_update_kernel, (ShapeStock, RG, false, false), grid, dim3(256), lds, s, m, t, g, gru_blocks, fb_, tiles, frames_first);
    else return hipErrorInvalidValue;        // (21..32 units on the one-wave kernel: engine.hip takes two launches, can_fuse)
    return hipGetLastError();
}

// (the fused kernels are built for the stock table shape only: engine.hip falls back to two launches otherwise)
template <class R>
static hipError_t launch_fused(const MfccStreamArgs<R>& m, const WaveTables<R>& t, const GruArgs& g, int n_cus, hipStream_t s) {
    if (t.L.mel_pad != ShapeStock::MEL || !blob_matches_shape(t)) return hipErrorInvalidValue;
    if (g.bf16) {
        const int tiles = (m.geo.n_streams + kTileStreams - 1) / kTileStreams;
        const int tpw = 4;                     // network tiles per network workgroup
        const int gru_blocks = (tiles + tpw - 1) / tpw;
        const int ff = tiles >= 4 * n_cus ? 1 : 0;
        const int frames_first = (ff & 1) | (tpw << 8);
        const int fb = stream_frame_blocks(m.geo.n_streams, n_cus, ff ? 3 : 4);
        const int book = tiles;
        const dim3 grid(gru_blocks + fb + book);
        if (g.use_delta && g.ring_bf16) PE_LAUNCH_R(R, fused_update_bf16_kernel, (ShapeStock, true, true), grid, dim3(256), frame_lds(t), s, m, t, g, gru_blocks, fb, tiles, frames_first);
        else if (g.use_delta) PE_LAUNCH_R(R, fused_update_bf16_kernel, (ShapeStock, true, false), grid, dim3(256), frame_lds(t), s, m, t, g, gru_blocks, fb, tiles, frames_first);
        else if (g.ring_bf16) PE_LAUNCH_R(R, fused_update_bf16_kernel, (ShapeStock, false, true), grid, dim3(256), frame_lds(t), s, m, t, g, gru_blocks, fb, tiles, frames_first);
        else PE_LAUNCH_R(R, fused_update_bf16_kernel, (ShapeStock, false, false), grid, dim3(256), frame_lds(t), s, m, t, g, gru_blocks, fb, tiles, frames_first);
        return hipGetLastError();
    }
    switch (gru_small_regs(g.units)) {
        case 1: return launch_fused_rg<R, 1>(m, t, g, n_cus, s);
        case 2: return launch_fused_rg<R, 2>(m, t, g, n_cus, s);
        case 3: return launch_fused_rg<R, 3>(m, t, g, n_cus, s);
        case 4: return launch_fused_rg<R, 4>(m, t, g, n_cus, s);
        case 5: return launch_fused_rg<R, 5>(m, t, g, n_cus, s);
        case 6: return launch_fused_rg<R, 6>(m, t, g, n_cus, s);
        case 7: return launch_fused_rg<R, 7>(m, t, g, n_cus, s);
        case 8: return launch_fused_rg<R, 8>(m, t, g, n_cus, s);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_fused_f64(const MfccStreamArgs<double>& m, const WaveTables<double>& t, const GruArgs& g, int n_cus, hipStream_t s) { return launch_fused<double>(m, t, g, n_cus, s); }
hipError_t launch_fused_f32(const MfccStreamArgs<float>& m, const WaveTables<float>& t, const GruArgs& g, int n_cus, hipStream_t s) { return launch_fused<float>(m, t, g, n_cus, s); }

// ---- general front end (mfcc_general_device.h): one wave per stream / per frame -------------------------------------
// (<= 128 registers: four waves per SIMD -- a wave walks the LDS round trips of one frame at a time, the others hide them;
//  BITS = log2(n_fft / 2): the per-lane loops of a frame are unrolled for the transform length)
// (n_fft = 2048: 25 KB of LDS per wave in float64 leave six waves per compute unit anyway: no register cap there)
template <class R, int BITS, bool BLUE = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(BITS >= 10 ? 2 : 4))) void mfcc_general_stream_kernel(const GeneralStreamArgs<R> a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifndef PE_GEN_TWO_WAVES
#define PE_GEN_TWO_WAVES 0      // two waves per stream (one per frame-row parity): measured SLOWER (62.6 vs 47.7 us per update at 4096 streams:
                                // the launch is bound by rounds of resident waves, and this doubles the waves)
#endif
    const int s = PE_GEN_TWO_WAVES ? blockIdx.x >> 1 : blockIdx.x;
    if (s < a.geo.n_streams) general_stream<R, BITS, BLUE>(a, reinterpret_cast<R*>(smem), s, PE_GEN_TWO_WAVES ? blockIdx.x & 1 : 0, threadIdx.x, PE_GEN_TWO_WAVES ? 2 : 1);
}
template <class R, int BITS, bool BLUE = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(BITS >= 10 ? 2 : 4))) void mfcc_general_offline_kernel(const GeneralOfflineArgs<R> a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    general_offline<R, BITS, BLUE>(a, reinterpret_cast<R*>(smem), blockIdx.x, gridDim.x, threadIdx.x);
}
// float32 butterflies without packed float32 here too (round 6, advisor r5): the general front end may run beside the
// five-values bf16 network of ANOTHER engine on the same compute unit, the combination whose stock-shape twin went wrong
// with packed instructions (see PE_NO_PK_F32 above); tests/test_gpu_parity.py soaks it
template <class R, int BITS, bool BLUE = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(BITS >= 10 ? 2 : 4))) PE_NO_PK_F32 void mfcc_general_stream_kernel_nopk(const GeneralStreamArgs<R> a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int s = PE_GEN_TWO_WAVES ? blockIdx.x >> 1 : blockIdx.x;
    if (s < a.geo.n_streams) general_stream<R, BITS, BLUE>(a, reinterpret_cast<R*>(smem), s, PE_GEN_TWO_WAVES ? blockIdx.x & 1 : 0, threadIdx.x, PE_GEN_TWO_WAVES ? 2 : 1);
}
template <class R, int BITS, bool BLUE = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(BITS >= 10 ? 2 : 4))) PE_NO_PK_F32 void mfcc_general_offline_kernel_nopk(const GeneralOfflineArgs<R> a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    general_offline<R, BITS, BLUE>(a, reinterpret_cast<R*>(smem), blockIdx.x, gridDim.x, threadIdx.x);
}
template <class R, int BITS, bool BLUE = false>
static void launch_general_stream_b(const GeneralStreamArgs<R>& a, hipStream_t s) {
    PE_LAUNCH_R(R, mfcc_general_stream_kernel, (BITS, BLUE), dim3((PE_GEN_TWO_WAVES ? 2 : 1) * (unsigned)a.geo.n_streams), dim3(64), general_lds_bytes(sizeof(R), a.tab.n_fft, a.tab.n_filt, a.tab.n_rounds), s, a);
}
template <class R, int BITS, bool BLUE = false>
static void launch_general_offline_b(const GeneralOfflineArgs<R>& a, unsigned blocks, hipStream_t s) {
    PE_LAUNCH_R(R, mfcc_general_offline_kernel, (BITS, BLUE), dim3(blocks), dim3(64), general_lds_bytes(sizeof(R), a.tab.n_fft, a.tab.n_filt, a.tab.n_rounds), s, a);
}
template <class R>
static hipError_t launch_general_stream_t(const GeneralStreamArgs<R>& a, hipStream_t s) {
    if (a.geo.n_streams == 0) return hipSuccess;
    if (a.tab.chirp) {                      // n_fft not a power of two: Bluestein over L = 2^log2m points
        switch (a.tab.log2m) {
            case 7: launch_general_stream_b<R, 7, true>(a, s); break;
            case 8: launch_general_stream_b<R, 8, true>(a, s); break;
            case 9: launch_general_stream_b<R, 9, true>(a, s); break;
            case 10: launch_general_stream_b<R, 10, true>(a, s); break;
            case 11: launch_general_stream_b<R, 11, true>(a, s); break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    switch (a.tab.log2m) {
        case 5: launch_general_stream_b<R, 5>(a, s); break;
        case 6: launch_general_stream_b<R, 6>(a, s); break;
        case 7: launch_general_stream_b<R, 7>(a, s); break;
        case 8: launch_general_stream_b<R, 8>(a, s); break;
        case 9: launch_general_stream_b<R, 9>(a, s); break;
        case 10: launch_general_stream_b<R, 10>(a, s); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
template <class R>
static hipError_t launch_general_offline_t(const GeneralOfflineArgs<R>& a, int n_cus, hipStream_t s) {
    if (a.n_frames <= 0) return hipSuccess;
    const long long cap = (long long)n_cus * 16;
    const unsigned blocks = (unsigned)(a.n_frames < cap ? a.n_frames : cap);
    if (a.tab.chirp) {
        switch (a.tab.log2m) {
            case 7: launch_general_offline_b<R, 7, true>(a, blocks, s); break;
            case 8: launch_general_offline_b<R, 8, true>(a, blocks, s); break;
            case 9: launch_general_offline_b<R, 9, true>(a, blocks, s); break;
            case 10: launch_general_offline_b<R, 10, true>(a, blocks, s); break;
            case 11: launch_general_offline_b<R, 11, true>(a, blocks, s); break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    switch (a.tab.log2m) {
        case 5: launch_general_offline_b<R, 5>(a, blocks, s); break;
        case 6: launch_general_offline_b<R, 6>(a, blocks, s); break;
        case 7: launch_general_offline_b<R, 7>(a, blocks, s); break;
        case 8: launch_general_offline_b<R, 8>(a, blocks, s); break;
        case 9: launch_general_offline_b<R, 9>(a, blocks, s); break;
        case 10: launch_general_offline_b<R, 10>(a, blocks, s); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
hipError_t launch_general_stream_f64(const GeneralStreamArgs<double>& a, hipStream_t s) { return launch_general_stream_t<double>(a, s); }
hipError_t launch_general_stream_f32(const GeneralStreamArgs<float>& a, hipStream_t s) { return launch_general_stream_t<float>(a, s); }
hipError_t launch_general_offline_f64(const GeneralOfflineArgs<double>& a, int n_cus, hipStream_t s) { return launch_general_offline_t<double>(a, n_cus, s); }
hipError_t launch_general_offline_f32(const GeneralOfflineArgs<float>& a, int n_cus, hipStream_t s) { return launch_general_offline_t<float>(a, n_cus, s); }

// ---- small utility kernels ---------------------------------------------------------------------
__global__ void gather_kernel(const GatherArgs a) {
    // out[s][t][f] = ring row of frame (ke - T + t) of stream s      (Listener.mfccs, oldest first)
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)a.n_streams * a.n_features * a.n_mfcc;
    if (idx >= total) return;
    const int f = (int)(idx % a.n_mfcc);
    const int t = (int)((idx / a.n_mfcc) % a.n_features);
    const long long s = idx / ((long long)a.n_mfcc * a.n_features);
    const RecPair both = rec_request(a.st.rec, a.st.n_padded, s);
    const uint32_t ke = rec_pick(both, rec_side(both, a.st.call)).ke;
    const uint32_t slot = (ke - (uint32_t)a.n_features + (uint32_t)t) & (uint32_t)(a.ring_slots - 1);
    const long long tile = s / kTileStreams;
    const int j = (int)(s % kTileStreams);
    const size_t at = (((size_t)tile * a.ring_slots + slot) * kTileStreams + j) * a.row_floats + f;
    a.out[idx] = a.ring_bf16 ? (float)reinterpret_cast<const __bf16*>(a.ring)[at] : a.ring[at];
}

__global__ void scatter_kernel(const GatherArgs a) {
    // inverse of gather_kernel: the stream restarts with the given [T][F] window already emitted
    // (frames 0..T-1 in slots 0..T-1, nothing held toward the next frame)
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int RF = a.row_floats;
    const long long total = (long long)a.n_streams * a.n_features * RF;
    if (idx >= total) return;
    const int f = (int)(idx % RF);
    const int t = (int)((idx / RF) % a.n_features);
    const long long s = idx / ((long long)RF * a.n_features);
    const long long tile = s / kTileStreams;
    const int j = (int)(s % kTileStreams);
    const float v = f < a.n_mfcc ? a.out[(s * a.n_features + t) * a.n_mfcc + f] : 0.0f;
    const size_t at = (((size_t)tile * a.ring_slots + t) * kTileStreams + j) * RF + f;
    if (a.ring_bf16) reinterpret_cast<__bf16*>(const_cast<float*>(a.ring))[at] = (__bf16)v;
    else const_cast<float*>(a.ring)[at] = v;
    if (f == 0 && t == 0) {       // side 0 becomes the current one (stamped with this call), side 1 the older
        a.st.rec[rec_at(a.st.n_padded, s, 0)] = StreamRec{0, (uint32_t)a.n_features, (uint32_t)a.n_features, a.st.call};
        a.st.rec[rec_at(a.st.n_padded, s, 1)] = StreamRec{0, (uint32_t)a.n_features, (uint32_t)a.n_features, a.st.call - 1u};
    }
}

hipError_t launch_scatter(const GatherArgs& a, hipStream_t s) {
    const long long total = (long long)a.n_streams * a.n_features * a.row_floats;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(scatter_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

// call numbers wrap after 2^32 calls: long before, every record is renumbered (current side 2, other side 1) and the host
// restarts its counter at 3 -- only the ORDER of a stream's two sides and "not this call" are ever read
__global__ void renumber_kernel(const StreamState st, const int n_padded) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_padded) return;
    const RecPair both = rec_request(st.rec, st.n_padded, s);
    const int side = rec_side(both, st.call);
    st.rec[rec_at(st.n_padded, s, side)].wcall = 2u;
    st.rec[rec_at(st.n_padded, s, side ^ 1)].wcall = 1u;
}
// leaving the keep style (MfccStreamArgs::head): 16 lanes per stream copy its leftover -- the last q samples of its row of the
// kept chunks -- into its current carry side (a rare launch: one per switch of calling styles)
__global__ __launch_bounds__(256) void materialize_carry_kernel(const StreamState st, const int16_t* head, const int head_chunk, const int n_streams) {
    const int s = blockIdx.x * 16 + (threadIdx.x >> 4), r = threadIdx.x & 15;
    if (s >= n_streams) return;
    const RecPair both = rec_request(st.rec, st.n_padded, s);
    const int side = rec_side(both, st.call);
    const int q = side ? both.r1.q : both.r0.q;
    const int16_t* const src = head + (size_t)s * head_chunk + (head_chunk - q);
    int16_t* const dst = st.carry + ((size_t)side * st.n_padded + (size_t)s) * kCarryCap;
    for (int i = r; i < q; i += 16) dst[i] = src[i];
}
hipError_t launch_materialize_carry(const StreamState& st, const int16_t* head, int head_chunk, int n_streams, hipStream_t s) {
    hipLaunchKernelGGL(materialize_carry_kernel, dim3((n_streams + 15) / 16), dim3(256), 0, s, st, head, head_chunk, n_streams);
    return hipGetLastError();
}

hipError_t launch_renumber(const StreamState& st, int n_padded, hipStream_t s) {
    hipLaunchKernelGGL(renumber_kernel, dim3((n_padded + 255) / 256), dim3(256), 0, s, st, n_padded);
    return hipGetLastError();
}

__global__ void clear_kernel(const ClearArgs a) {
    // one workgroup per stream: zero its counters and every ring row
    const long long s = blockIdx.x;
    if (s >= a.n_streams) return;
    if (a.mask && !a.mask[s]) return;
    if (threadIdx.x == 0) {         // side 0 becomes the current one (stamped with this call), side 1 the older
        a.st.rec[rec_at(a.st.n_padded, s, 0)] = StreamRec{0, 0u, 0u, a.st.call};
        a.st.rec[rec_at(a.st.n_padded, s, 1)] = StreamRec{0, 0u, 0u, a.st.call - 1u};
        if (a.activation) a.activation[s] = 0;
    }
    const long long tile = s / kTileStreams;
    const int j = (int)(s % kTileStreams);
    for (int i = threadIdx.x; i < a.ring_slots * a.row_floats; i += blockDim.x) {
        const int slot = i / a.row_floats, f = i % a.row_floats;
        const size_t at = (((size_t)tile * a.ring_slots + slot) * kTileStreams + j) * a.row_floats + f;
        if (a.ring_bf16) reinterpret_cast<__bf16*>(a.ring)[at] = (__bf16)0.0f;
        else a.ring[at] = 0.0f;
    }
    if (a.proj_ring)            // the projection of an all-zero frame is the bias row
        for (int i = threadIdx.x; i < a.ring_slots * kProjRow; i += blockDim.x) {
            const int slot = i / kProjRow, o = i % kProjRow;
            // element o = 16 g + 4 tl + q of stream j sits at [tl][j][g][q] inside the (tile, slot) block
            a.proj_ring[((size_t)tile * a.ring_slots + slot) * kTileStreams * kProjRow + (size_t)((o >> 2) & 3) * (kTileStreams * 16) + j * 16 + (o >> 4) * 4 + (o & 3)] = a.proj_b[o];
        }
}

__global__ void project_rows_kernel(const float* ring, float* proj, const float* w, const float* b, const int n_mfcc, const long long n_rows) {
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int o = threadIdx.x & 63;
    if (row >= n_rows) return;
    float acc = b[o];
    for (int c = 0; c < n_mfcc; ++c) acc = fmaf(ring[row * kRowFloats + c], w[c * kProjRow + o], acc);
    const long long block = row / kTileStreams;          // (tile, slot) block; row % 16 = stream j
    const int j = (int)(row % kTileStreams);
    proj[block * kTileStreams * kProjRow + (size_t)((o >> 2) & 3) * (kTileStreams * 16) + j * 16 + (o >> 4) * 4 + (o & 3)] = acc;
}

hipError_t launch_project_rows(const float* ring, float* proj, const float* w, const float* b, int n_mfcc, long long n_rows, hipStream_t s) {
    if (n_rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(project_rows_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, s, ring, proj, w, b, n_mfcc, n_rows);
    return hipGetLastError();
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



}  // namespace pe

