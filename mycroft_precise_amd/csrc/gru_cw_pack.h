// Layout and host-side packing of the Keras GRU matrices for the shapes of gru_cw_device.h (stock width, 17 <= H <= 20).
// Keras layout (model.py:76-82 -> GRU weights): kernel [F][3H], recurrent_kernel [H][3H], bias [3H], gate order z | r | h.
#pragma once
#include <vector>
#include <cstddef>

namespace pe {

// packed operands of the re-tiled shapes (GruArgs::cw), offsets in floats
struct CwPack {
    static constexpr int WX = 0;                        // [TZ, TX, TC, TV][4 kk][64]  input kernel, A operands
    static constexpr int BIAS = WX + 4 * 4 * 64;        // [TZ, TX, TC, TV][4 q][64]   accumulator inits
    static constexpr int WR = BIAS + 4 * 4 * 64;        // [TZ, TX, TC][5 rho][64]     recurrent kernel, A operands
    static constexpr int WV = WR + 3 * 5 * 64;          // [z, r, c][5 rho][64]        4x4x1 A operands of units 16..19
    static constexpr int WF = WV + 3 * 5 * 64;          // [z, r, c][4 a][5 rho][64]   the same weights per target a (VALU form)
    static constexpr int WXD = WF + 3 * 4 * 5 * 64;     // [TZ, TX, TC, TV][4 kk][64]  input kernel rows F .. 2F-1 (use_delta), else 0
    static constexpr int SIZE = WXD + 4 * 4 * 64;
};
enum { kTZ = 0, kTX = 1, kTC = 2, kTV = 3 };


// Tile tau in {TZ, TX, TC}: output register q <-> gate tau, units 4 q + g.  TV: q = 0, 1, 2 <-> gate q, units 16 + g.
inline void cw_slot(int tile, int q, int& gate, int& rho) {
    if (tile == kTV) { gate = q; rho = 4; }
    else { gate = tile; rho = q; }
}

// delta: the kernel has 2 F rows, rows F .. 2F-1 multiply x_t - x_(t-1) (vectorization.py:53-59)
inline std::vector<float> pack_gru_cw(const float* kernel, const float* recurrent, const float* bias, int F, int H, bool delta = false) {
    std::vector<float> blob(CwPack::SIZE, 0.f);
    for (int tile = 0; tile < 4; ++tile)
        for (int lane = 0; lane < 64; ++lane) {
            const int i = lane & 15, g = lane >> 4;
            {   // A operand: row i of the tile, k-slot g
                const int q = i & 3, gout = i >> 2;
                int gate, rho;
                cw_slot(tile, q, gate, rho);
                const int u = 4 * rho + gout;
                if (!(tile == kTV && q == 3) && u < H) {
                    const int col = gate * H + u;
                    for (int kk = 0; kk < 4; ++kk) {
                        const int phi = 4 * g + kk;
                        if (phi < F) blob[CwPack::WX + (tile * 4 + kk) * 64 + lane] = kernel[(size_t)phi * 3 * H + col];
                        if (phi < F && delta) blob[CwPack::WXD + (tile * 4 + kk) * 64 + lane] = kernel[(size_t)(F + phi) * 3 * H + col];
                    }
                    if (tile != kTV)
                        for (int rs = 0; rs < 5; ++rs) {
                            const int usrc = 4 * rs + g;
                            if (usrc < H) blob[CwPack::WR + (tile * 5 + rs) * 64 + lane] = recurrent[(size_t)usrc * 3 * H + col];
                        }
                }
            }
            for (int q = 0; q < 4; ++q) {   // C operand: this lane's output rows 4 g + q
                int gate, rho;
                cw_slot(tile, q, gate, rho);
                const int u = 4 * rho + g;
                if (!(tile == kTV && q == 3) && u < H) blob[CwPack::BIAS + (tile * 4 + q) * 64 + lane] = bias[gate * H + u];
            }
        }
    // 4x4x1 A operands: lane (g, stream) supplies U[source unit 4 rho + g][target unit 16 + (lane & 3)]
    for (int gate = 0; gate < 3; ++gate)
        for (int rho = 0; rho < 5; ++rho)
            for (int lane = 0; lane < 64; ++lane) {
                const int g = lane >> 4, tgt = 16 + (lane & 3), usrc = 4 * rho + g;
                if (tgt < H && usrc < H) blob[CwPack::WV + (gate * 5 + rho) * 64 + lane] = recurrent[(size_t)usrc * 3 * H + gate * H + tgt];
                for (int a = 0; a < 4; ++a)     // the same weights, one array per target (VALU form)
                    if (16 + a < H && usrc < H) blob[CwPack::WF + ((gate * 4 + a) * 5 + rho) * 64 + lane] = recurrent[(size_t)usrc * 3 * H + gate * H + 16 + a];
            }
    return blob;
}

}  // namespace pe
