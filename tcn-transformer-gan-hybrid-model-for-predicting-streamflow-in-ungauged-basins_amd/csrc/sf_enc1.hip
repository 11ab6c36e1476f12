// First encoder layer without its pre-activation tensor: Conv2d(C<=16, K, 3, padding=1, bias=False)
// -> BatchNorm2d -> ReLU -> MaxPool2d(2) (src/model/cnn_encoder.py:42-50, layer 0).
//
// With at most 16 (padded) input channels the conv is one MFMA reduction step per tap: 9 MFMAs
// per 32 pixels x 32 output channels.  Its K-channel output at full resolution, however, is the
// largest tensor of the whole model (51 MB per sample in bf16) and the reference formulation
// walks it six times (write, statistics, normalise+pool, two backward passes, weight gradient).
// Here the conv is recomputed from the 16-channel input wherever its output is needed and the
// output never leaves the registers:
//   sf_enc1_stats : conv -> round to bf16 -> per-channel sum / sum of squares
//   sf_enc1_fwd   : conv -> round to bf16 -> scale/shift -> ReLU -> 2x2 max -> pooled map
//   sf_enc1_bwd   : conv -> ... -> routed pooled gradient dzm; accumulates the correlations
//                   A1 = dzm^T xp and A2 = z^T xp (xp = input patches), the patch sums S and the
//                   BatchNorm sums, from which a finalize kernel forms the weight / gamma / beta
//                   gradients (the input needs no gradient).
// The MFMA runs with D[pixel][channel]: a lane owns one output channel, its 16 accumulator
// registers are 16 pixels ordered so that four consecutive registers are one 2x2 pooling window
// -- channel sums, the window maximum and the arg-max routing are all in-lane.
#include "sf_common.h"
#include "../../include/sfhip.h"
#include <type_traits>

namespace {

constexpr int E_NT = 256;                    // 4 waves
constexpr int E_TH = 16, E_TW = 16;          // output pixels per tile (8 x 8 pooling windows)
constexpr int E_PW = E_TW + 2, E_PP = (E_TH + 2) * E_PW;   // halo patch
constexpr int E_ROWB = 32;                   // bytes per LDS row: 16 bf16 channels
constexpr int E_PCHUNKS = E_PP * 2;          // 16-byte chunks of a patch
constexpr int E_PLOADS = (E_PCHUNKS + E_NT - 1) / E_NT;

enum { E_STATS = 0, E_FWD = 1 };
constexpr int E_SUMS_GRID = 1024;          // workgroups (and slabs) of enc1_input_sums_kernel

struct E1Params {
    const bf16* x;       // [N,H,W,16]
    const bf16* w;       // packed [K][9][16]
    int N, H, W, K;
    int tiles_x, tiles_y, ntiles;
    double* ws;          // stats: [2K] sums, sums of squares
    const float* scale;  // fwd
    const float* shift;
    bf16* y;             // fwd: [N,H/2,W/2,K]
    unsigned short* codes;   // fwd, optional: the routing of every pooling window (layout: code_index)
};

// Routing codes.  For every pooling window and channel the forward pass can leave 3 bits: the
// position of the window's maximum (0..3 = (0,0), (0,1), (1,0), (1,1); the first one in that order
// on ties, as torch's max-pool backward picks it) and whether it passed the ReLU.  A lane of the
// forward kernel owns one channel and, per window row of a tile, the four windows 2 g + h of the
// tile's eight: their codes are packed into one 16-bit word (code g in bits 3 g .. 3 g + 2), stored
// at [(n, pooled row wy)][tile column][h][channel].  The backward pass then routes the pooled
// gradient with a few bit operations instead of recomputing conv + BatchNorm + arg-max.
__host__ __device__ inline size_t code_index(int n, int wy, int tx, int h, int k, int PH, int tiles_x, int K) {
    return ((((size_t)n * PH + wy) * tiles_x + tx) * 2 + h) * K + k;
}

__device__ __forceinline__ bf16x8 to_bf16x8(const float* f) {
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (bf16)f[i];
    return v;
}

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
    typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 t;
    t[0] = (bf16)a; t[1] = (bf16)b;
    return *reinterpret_cast<const unsigned*>(&t);
}

struct TilePos { int n, y0, x0; };
__device__ __forceinline__ TilePos tile_pos(const E1Params& p, int t) {
    TilePos tp;
    const int tx = t % p.tiles_x, r = t / p.tiles_x;
    tp.x0 = tx * E_TW;
    tp.y0 = (r % p.tiles_y) * E_TH;
    tp.n = r / p.tiles_y;
    return tp;
}

// the workgroup's copy of the weights: LDS [tap][k][16 channels]
__device__ __forceinline__ void stage_weights(const E1Params& p, char* sW) {
    const int chunks = p.K * 9 * 2;
    for (int i = threadIdx.x; i < chunks; i += E_NT) {
        const int half = i & 1, row = i >> 1;        // row = k * 9 + tap in the packed tensor
        const int k = row / 9, tap = row - k * 9;
        const uint4 v = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(p.w) + (size_t)i * 16);
        // the two 16-byte halves of a row are swapped on every second group of 8 rows: 16 lanes
        // reading 16 consecutive rows then cover all 64 banks once
        *reinterpret_cast<uint4*>(sW + ((tap * p.K + k) * 2 + (half ^ ((k >> 3) & 1))) * 16) = v;
    }
}

// halo patch of a tile -> registers (buffer loads: an offset past the end returns zeros, which is
// how the padding ring and tiles hanging over the image edge are filled)
__device__ __forceinline__ void load_patch(const E1Params& p, __amdgpu_buffer_rsrc_t rs, const TilePos& tp, bool live,
                                           uint4* regs) {
#pragma unroll
    for (int j = 0; j < E_PLOADS; ++j) {
        const int c = threadIdx.x + j * E_NT;
        const int half = c & 1, row = c >> 1;
        const int py = row / E_PW, px = row - py * E_PW;
        const int gy = tp.y0 - 1 + py, gx = tp.x0 - 1 + px;
        const bool ok = live && c < E_PCHUNKS && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        const int off = ok ? (((tp.n * p.H + gy) * p.W + gx) * 2 + half) * 16 : (int)0x80000000;
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
        regs[j] = *reinterpret_cast<const uint4*>(&v);
    }
}
// SWZ: the halves of a patch row are swapped on odd patch lines, which separates the two lanes of
// a 16-lane read group that sit one line apart (the 2x2 window ordering of the MFMA rows)
template <bool SWZ>
__device__ __forceinline__ void store_patch(char* sP, const uint4* regs) {
#pragma unroll
    for (int j = 0; j < E_PLOADS; ++j) {
        const int c = threadIdx.x + j * E_NT;
        const int row = c >> 1, py = row / E_PW;
        const int cs = SWZ ? (c ^ (py & 1)) : c;
        if (c < E_PCHUNKS) *reinterpret_cast<uint4*>(sP + cs * 16) = regs[j];
    }
}

template <int MODE, int NKT>
__global__ __launch_bounds__(E_NT, 3) void enc1_kernel(E1Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sW = smem;                               // 9 * K * 32 bytes
    char* sP = smem + 9 * p.K * E_ROWB;            // E_PP * 32 bytes
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    stage_weights(p, sW);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.x, 0, (unsigned)((size_t)p.N * p.H * p.W * E_ROWB), 0x00020000);

    // lane constants.  MFMA row r of a row tile = window r >> 2 (along x), position r & 3 in it.
    const int a_px = 2 * (r >> 2) + (r & 1), a_py = (r >> 1) & 1;
    // + (2 rt + ty) * PW * 32 + tx * 32; the half depends on the parity of the patch line a_py + ty
    const int a_off0 = (a_py * E_PW + a_px) * E_ROWB + 16 * (h ^ a_py);
    const int a_off1 = (a_py * E_PW + a_px) * E_ROWB + 16 * (h ^ a_py ^ 1);
    const int b_off = r * E_ROWB + 16 * (h ^ ((r >> 3) & 1));     // + (tap * K + kt * 32) * 32

    float ssum[NKT], ssq[NKT], sc[NKT], sf[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        ssum[kt] = ssq[kt] = 0.f;
        if (MODE == E_FWD) { sc[kt] = p.scale[kt * 32 + r]; sf[kt] = p.shift[kt * 32 + r]; }
    }
    const int PH = p.H / 2, PWd = p.W / 2;

    uint4 pre[E_PLOADS];
    int t = blockIdx.x;
    TilePos tp = tile_pos(p, t < p.ntiles ? t : 0);
    load_patch(p, rs, tp, t < p.ntiles, pre);
    for (; t < p.ntiles; t += gridDim.x) {
        __syncthreads();                 // everyone is done with the previous patch (and sW is staged)
        store_patch<true>(sP, pre);
        __syncthreads();
        const TilePos cur = tp;
        const int tn = t + gridDim.x;
        tp = tile_pos(p, tn < p.ntiles ? tn : 0);
        load_patch(p, rs, tp, tn < p.ntiles, pre);   // next tile's patch flies under this tile's MFMAs
        const bool full = cur.y0 + E_TH <= p.H && cur.x0 + E_TW <= p.W;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int rt = wave * 2 + q;             // window row of the tile
            bf16x8 fa[9];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
                fa[tap] = *reinterpret_cast<const bf16x8*>(sP + (tap / 3 == 1 ? a_off1 : a_off0) + ((2 * rt + tap / 3) * E_PW + tap % 3) * E_ROWB);
            // validity of the lane's 16 pixels: register i -> window 2 (i >> 2) + h, position i & 3
            const int gy0 = cur.y0 + 2 * rt;
            // two channel tiles at a time: their MFMA chains (nine dependent MFMAs each) alternate, so that
            // one chain's result latency is covered by the other's issue
            auto finish_tile = [&](int kt, const f32x16& acc) __attribute__((always_inline)) {
                // tiles inside the image skip the validity tests (wave-uniform branch)
                if (MODE == E_STATS) {
                    float s = 0.f, q2 = 0.f;
                    if (full) {
#pragma unroll
                        for (int i = 0; i < 16; i += 2) {
                            const unsigned pk = pack_bf16x2(acc[i], acc[i + 1]);
                            const float z0 = __uint_as_float(pk << 16), z1 = __uint_as_float(pk & 0xffff0000u);
                            s += z0 + z1;
                            q2 += z0 * z0 + z1 * z1;
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int gy = gy0 + ((i >> 1) & 1), gx = cur.x0 + 2 * (2 * (i >> 2) + h) + (i & 1);
                            const float z = (float)(bf16)acc[i];
                            const bool ok = gy < p.H && gx < p.W;
                            s += ok ? z : 0.f;
                            q2 += ok ? z * z : 0.f;
                        }
                    }
                    ssum[kt] += s;
                    ssq[kt] += q2;
                } else {
                    const int wy = (cur.y0 >> 1) + rt;
                    bf16* yrow = p.y + (((size_t)cur.n * PH + wy) * PWd + (cur.x0 >> 1) + h) * p.K + kt * 32 + r;
                    unsigned cw = 0;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const unsigned p01 = pack_bf16x2(acc[4 * g], acc[4 * g + 1]), p23 = pack_bf16x2(acc[4 * g + 2], acc[4 * g + 3]);
                        const float v0 = __uint_as_float(p01 << 16) * sc[kt] + sf[kt], v1 = __uint_as_float(p01 & 0xffff0000u) * sc[kt] + sf[kt];
                        const float v2 = __uint_as_float(p23 << 16) * sc[kt] + sf[kt], v3 = __uint_as_float(p23 & 0xffff0000u) * sc[kt] + sf[kt];
                        // first maximum in the order 0,1,2,3 (strict >), as a two-level tournament
                        const bool c1 = v1 > v0, c3 = v3 > v2;
                        const float b01 = c1 ? v1 : v0, b23 = c3 ? v3 : v2;
                        const bool cf = b23 > b01;
                        const float best = cf ? b23 : b01;
                        float mx = fmaxf(best, 0.f);                                  // ReLU of the maximum
                        if ((v0 != v0) | (v1 != v1) | (v2 != v2) | (v3 != v3)) mx = __builtin_nanf("");   // torch keeps a NaN
                        cw |= ((cf ? 2u + (c3 ? 1u : 0u) : (c1 ? 1u : 0u)) | (best > 0.f ? 4u : 0u)) << (3 * g);
                        const int wx = (cur.x0 >> 1) + 2 * g + h;
                        if (full || (wy < PH && wx < PWd)) yrow[(size_t)2 * g * p.K] = (bf16)mx;
                    }
                    if (p.codes && (full || wy < PH))
                        p.codes[code_index(cur.n, wy, cur.x0 / E_TW, h, kt * 32 + r, PH, p.tiles_x, p.K)] = (unsigned short)cw;
                }
            };
#pragma unroll
            for (int kt = 0; kt < NKT; kt += 2) {
                // keeps the weight reads of later channel tiles from being hoisted up here (registers)
                __builtin_amdgcn_sched_barrier(0);
                f32x16 acc0, acc1;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.f;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const bf16x8 fb0 = *reinterpret_cast<const bf16x8*>(sW + b_off + (tap * p.K + kt * 32) * E_ROWB);
                    const bf16x8 fb1 = *reinterpret_cast<const bf16x8*>(sW + b_off + (tap * p.K + (kt + 1) * 32) * E_ROWB);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tap], fb0, acc0, 0, 0, 0);
                    if (kt + 1 < NKT) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tap], fb1, acc1, 0, 0, 0);
                }
                finish_tile(kt, acc0);
                if (kt + 1 < NKT) finish_tile(kt + 1, acc1);
            }
        }
    }
    if (MODE == E_STATS) {
        // lanes l and l + 32 hold the same channel; then the four waves; one atomic per channel
        __syncthreads();
        float* red = reinterpret_cast<float*>(sP);   // [2][4 waves][K]
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            const float s = ssum[kt] + __shfl_xor(ssum[kt], 32, 64);
            const float q2 = ssq[kt] + __shfl_xor(ssq[kt], 32, 64);
            if (h == 0) {
                red[wave * p.K + kt * 32 + r] = s;
                red[(4 + wave) * p.K + kt * 32 + r] = q2;
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * p.K; i += E_NT) {
            const int which = i / p.K, k = i - which * p.K;
            double v = 0.0;
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) v += (double)red[(which * 4 + w2) * p.K + k];
            unsafeAtomicAdd(p.ws + i, v);
        }
    }
}

// ---- backward ------------------------------------------------------------------------------
// Wave w of a workgroup owns output-channel tile kt = 4 * blockIdx.y + w (its nine weight
// fragments stay in registers) and walks all 8 window rows of every tile of the workgroup:
// conv -> z (bf16-rounded, zero outside the image) -> BatchNorm -> arg-max routing of the pooled
// gradient -> dzm.  Register i of a lane is reduction index 8h + (i & 7) of MFMA step i >> 3, so
// z and dzm are A operands as they stand; the B operand xp[pixel][(tap, c)] is gathered from the
// LDS patch with 2-byte reads (one column (tap, c) per lane, c < Creal only: 9 * Creal columns).
// Accumulated per wave: A1 = dzm^T xp, A2 = z^T xp (rows = channels) and the two BatchNorm sums.
// The tile's pooled gradients go through LDS (16-byte loads, prefetched a tile ahead like the
// patch).  Every workgroup stores its partial results to its own slab; the patch sums
// S[(tap, c)] = sum_p xp[p][(tap, c)] come from enc1_input_sums_kernel (whole-image, border-row,
// border-column and corner sums of the input); enc1_bwd_finalize_kernel adds the slabs in a fixed
// order and forms the gradients.
struct E1Bwd {
    E1Params c;
    const bf16* dy;       // [N,H/2,W/2,K] gradient wrt the pooled map
    const float* mean; const float* invstd;
    float* slab;          // [gridDim.x][slab_floats]
    int Creal, ncols;     // ncols = 9 * Creal
};
// slab layout (floats): A1 [K][NCT*32], A2 [K][NCT*32], m1 [K], m2 [K]
__host__ __device__ inline size_t slab_floats(int K, int nct) { return (size_t)2 * K * nct * 32 + 2 * K; }
constexpr int E_DCH = 64 * 16;                       // 16-byte chunks of a tile's pooled gradient (128 channels)
constexpr int E_DLOADS = E_DCH / E_NT;

template <int NCT>
__global__ __launch_bounds__(E_NT, 2) void enc1_bwd_kernel(E1Bwd a) {
    const E1Params& p = a.c;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sP = smem;
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kt = blockIdx.y * 4 + wave;
    const bool active = kt * 32 < p.K;
    const int k = active ? kt * 32 + r : 0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.x, 0, (unsigned)((size_t)p.N * p.H * p.W * E_ROWB), 0x00020000);
    const int PH = p.H / 2, PWd = p.W / 2;
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.dy, 0, (unsigned)((size_t)p.N * PH * PWd * p.K * 2), 0x00020000);
    // pooled gradient of a tile (8 x 8 windows x this workgroup's 128 channels) -> registers -> LDS
    auto load_dyt = [&](const TilePos& tq, bool live, uint4* regs) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < E_DLOADS; ++j) {
            const int c = threadIdx.x + j * E_NT;
            const int win = c >> 4, part = c & 15;
            const int wy = (tq.y0 >> 1) + (win >> 3), wx = (tq.x0 >> 1) + (win & 7);
            const bool ok = live && wy < PH && wx < PWd && blockIdx.y * 128 + part * 8 < p.K;
            const int off = ok ? (((tq.n * PH + wy) * PWd + wx) * p.K + blockIdx.y * 128 + part * 8) * 2 : (int)0x80000000;
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(rdy, off, 0, 0);
            regs[j] = *reinterpret_cast<const uint4*>(&v);
        }
    };

    // conv weights of the workgroup's 128 channels: LDS [tap][128][16] (B operand of
    // D[pixel][channel]; in registers they would cost 36 VGPRs the accumulators need)
    char* sWb = smem + E_PP * E_ROWB;
    char* sD = sWb + 9 * 128 * E_ROWB;                    // [64 windows][128 channels] bf16
    const int d_off = ((h * 128) + wave * 32 + r) * 2;    // + (rt * 8 + 2 q) * 256
    for (int i = threadIdx.x; i < 128 * 9 * 2; i += E_NT) {
        const int half = i & 1, row = i >> 1;
        const int kk = row / 9, tap = row - kk * 9;
        const int kg = blockIdx.y * 128 + kk;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (kg < p.K) v = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(p.w) + ((size_t)kg * 9 + tap) * 32 + half * 16);
        *reinterpret_cast<uint4*>(sWb + ((tap * 128 + kk) * 2 + (half ^ ((kk >> 3) & 1))) * 16) = v;
    }
    const int w_off = (wave * 32 + r) * E_ROWB + 16 * (h ^ ((r >> 3) & 1));   // + tap * 128 * 32 (halves swizzled)
    const float sc = p.scale[k], sf = p.shift[k], mu = a.mean[k], is = a.invstd[k];

    const int a_px = 2 * (r >> 2) + (r & 1), a_py = (r >> 1) & 1;
    const int a_off = (a_py * E_PW + a_px) * E_ROWB + 16 * h;
    // gather base of the lane's column (tap, c) in each column tile, + 2 window columns for h = 1
    int g_off[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        int col = ct * 32 + r;
        if (col >= a.ncols) col = 0;
        const int tap = col / a.Creal, ch = col - tap * a.Creal;
        g_off[ct] = ((tap / 3) * E_PW + tap % 3) * E_ROWB + 2 * ch + 2 * h * E_ROWB;
    }

    f32x16 accA1[NCT], accA2[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int i = 0; i < 16; ++i) accA1[ct][i] = accA2[ct][i] = 0.f;
    float m1 = 0.f, m2 = 0.f;

    uint4 pre[E_PLOADS], pred[E_DLOADS];
    int t = blockIdx.x;
    TilePos tp = tile_pos(p, t < p.ntiles ? t : 0);
    load_patch(p, rs, tp, t < p.ntiles, pre);
    load_dyt(tp, t < p.ntiles, pred);
    for (; t < p.ntiles; t += gridDim.x) {
        __syncthreads();
        store_patch<false>(sP, pre);
#pragma unroll
        for (int j = 0; j < E_DLOADS; ++j) *reinterpret_cast<uint4*>(sD + (threadIdx.x + j * E_NT) * 16) = pred[j];
        __syncthreads();
        const TilePos cur = tp;
        const int tn = t + gridDim.x;
        tp = tile_pos(p, tn < p.ntiles ? tn : 0);
        load_patch(p, rs, tp, tn < p.ntiles, pre);
        load_dyt(tp, tn < p.ntiles, pred);
        if (!active) continue;
        // tiles inside the image (all but the last tile row / column) skip every validity test
        const bool full = cur.y0 + E_TH <= p.H && cur.x0 + E_TW <= p.W;
        // one window row: conv, routing, correlations
        auto process = [&](int rt, auto FULL) __attribute__((always_inline)) {
            const char* prow = sP + 2 * rt * E_PW * E_ROWB;
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const bf16x8 fa = *reinterpret_cast<const bf16x8*>(prow + a_off + ((tap / 3) * E_PW + tap % 3) * E_ROWB);
                const bf16x8 fw = *reinterpret_cast<const bf16x8*>(sWb + w_off + tap * 128 * E_ROWB);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fw, acc, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // Routing, window by window, on packed bf16 pairs: zp = (z0,z1) / (z2,z3) are A-operand
            // dwords as they stand, and so are the routed gradients dp (the pooled gradient's 16
            // bits in the arg-max position, zero elsewhere).
            unsigned zp[8], dp[8];
            const int gy0 = cur.y0 + 2 * rt;
            unsigned short gcur[4];     // pooled gradient of the lane's 4 windows (zero outside the pooled map)
#pragma unroll
            for (int q = 0; q < 4; ++q) gcur[q] = *reinterpret_cast<const unsigned short*>(sD + d_off + (rt * 8 + 2 * q) * 256);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                unsigned p01 = pack_bf16x2(acc[4 * g], acc[4 * g + 1]), p23 = pack_bf16x2(acc[4 * g + 2], acc[4 * g + 3]);
                if (!decltype(FULL)::value) {
                    const int gx = cur.x0 + 2 * (2 * g + h);
                    const unsigned mx = (gx < p.W ? 0xffffu : 0u) | (gx + 1 < p.W ? 0xffff0000u : 0u);
                    const unsigned m01 = gy0 < p.H ? mx : 0u, m23 = gy0 + 1 < p.H ? mx : 0u;
                    p01 &= m01; p23 &= m23;
                }
                zp[2 * g] = p01; zp[2 * g + 1] = p23;
                const float z0 = __uint_as_float(p01 << 16), z1 = __uint_as_float(p01 & 0xffff0000u);
                const float z2 = __uint_as_float(p23 << 16), z3 = __uint_as_float(p23 & 0xffff0000u);
                const float v0 = z0 * sc + sf, v1 = z1 * sc + sf, v2 = z2 * sc + sf, v3 = z3 * sc + sf;
                // first maximum in the order 0,1,2,3 (strict >), as a two-level tournament
                const bool c1 = v1 > v0, c3 = v3 > v2;
                const float b01 = c1 ? v1 : v0, zb01 = c1 ? z1 : z0;
                const float b23 = c3 ? v3 : v2, zb23 = c3 ? z3 : z2;
                const bool cf = b23 > b01;
                const float best = cf ? b23 : b01, zbest = cf ? zb23 : zb01;
                const unsigned g16 = best > 0.f ? (unsigned)gcur[g] : 0u;   // ReLU: no gradient through max <= 0
                const float gv = __uint_as_float(g16 << 16);
                m1 += gv;
                m2 += gv * zbest;                    // sum dzm * z; centred and scaled at the end
                const unsigned ghi = g16 << 16;
                const unsigned s1 = c1 ? ghi : g16, s3 = c3 ? ghi : g16;
                dp[2 * g] = cf ? 0u : s1;
                dp[2 * g + 1] = cf ? s3 : 0u;
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 fz = *reinterpret_cast<const bf16x8*>(zp + 4 * s2);
                const bf16x8 fd = *reinterpret_cast<const bf16x8*>(dp + 4 * s2);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    __builtin_amdgcn_sched_barrier(0);
                    // xp of the 8 pixels (2 windows) of this half-step for the lane's column
                    unsigned short xs[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int win = 4 * s2 + 2 * (j >> 2), e = j & 3;     // + h via g_off
                        const int off = ((e >> 1) * E_PW + 2 * win + (e & 1)) * E_ROWB;
                        xs[j] = *reinterpret_cast<const unsigned short*>(prow + g_off[ct] + off);
                    }
                    unsigned xw[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) xw[q] = (unsigned)xs[2 * q] | ((unsigned)xs[2 * q + 1] << 16);
                    const bf16x8 fx = *reinterpret_cast<const bf16x8*>(xw);
                    accA1[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fd, fx, accA1[ct], 0, 0, 0);
                    accA2[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fz, fx, accA2[ct], 0, 0, 0);
                }
            }
        };
        // two window rows per trip; not unrolled further: a fully unrolled tile needs more registers
        // than two waves per SIMD have
        auto run_tile = [&](auto FULL) __attribute__((always_inline)) {
#pragma unroll 1
            for (int rp = 0; rp < 4; ++rp) {
                process(2 * rp, FULL);
                process(2 * rp + 1, FULL);
            }
        };
        if (full) run_tile(std::true_type{});
        else run_tile(std::false_type{});
    }
    // sum dzm * xhat = invstd * (sum dzm * z - mean * sum dzm)
    m2 = is * (m2 - mu * m1);
    // ---- slab: every element written (zeros from idle waves), no atomics
    float* slab = a.slab + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * slab_floats(p.K, NCT);
    const int ncp = NCT * 32;
    if (active) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = kt * 32 + 8 * (i >> 2) + 4 * h + (i & 3);
                slab[(size_t)row * ncp + ct * 32 + r] = accA1[ct][i];
                slab[(size_t)(p.K + row) * ncp + ct * 32 + r] = accA2[ct][i];
            }
        const float t1 = m1 + __shfl_xor(m1, 32, 64), t2 = m2 + __shfl_xor(m2, 32, 64);
        if (h == 0) {
            slab[(size_t)2 * p.K * ncp + k] = t1;
            slab[(size_t)2 * p.K * ncp + p.K + k] = t2;
        }
    }
}

// ---- backward, routed by the forward pass's codes ------------------------------------------------
// With the routing codes of the forward pass (code_index) the backward needs neither the conv nor
// BatchNorm nor the arg-max again, and not the pre-activation z either: both z-dependent sums are
// linear in quantities that do not involve the output channel,
//     sum_p dzm[p][k] z[p][k]      = sum_col W[k][col] A1[k][col]          (A1 = dzm^T xp)
//     A2[k][col] = sum_p z[p][k] xp[p][col] = sum_col' W[k][col'] G[col'][col]   (G = xp^T xp, the Gram
//                                                                          matrix of the input patches)
// so the kernel accumulates A1 (per channel tile), sum dzm and -- once, not per channel -- G; the
// finalize kernel applies the weights.  The B operand xp[pixel][(tap, c)] of all these products is
// the same for every channel tile: the four waves of a workgroup gather it ONCE per tile into LDS
// as ready-made MFMA fragments (48 x 1 KiB for three column tiles) and read it back with one
// 16-byte load per MFMA, instead of 8 two-byte gathers per MFMA and wave.
struct E1Bwd2 {
    E1Params c;
    const bf16* dy;                 // [N,H/2,W/2,K] gradient wrt the pooled map
    const unsigned short* codes;    // routing codes of the forward pass
    float* slab;                    // [gridDim.y][gridDim.x][slab2_floats]: A1 [128][NCT*32], sum dzm [128]
    float* gslab;                   // [gridDim.x][NP][32][32] partial Gram matrices (channel group 0 only)
    int Creal, ncols, gram;
};
__host__ __device__ inline size_t slab2_floats(int nct) { return (size_t)128 * nct * 32 + 128; }
__host__ __device__ inline int gram_pairs(int nct) { return nct * (nct + 1) / 2; }
__host__ __device__ inline int gram_pair(int a, int b, int nct) { return a * nct - a * (a - 1) / 2 + (b - a); }   // a <= b

// MODE 0: A1 and sum dzm only (the Gram matrix is at hand already); 1: both; 2: the Gram matrix only
// (a pass over the input alone: no gradients, no codes).
template <int NCT, int MODE>
__global__ __launch_bounds__(E_NT, 2) void enc1_bwd2_kernel(E1Bwd2 a) {
    constexpr bool GRAM = MODE != 0, WANT_A1 = MODE != 2;
    constexpr int NP = GRAM ? NCT * (NCT + 1) / 2 : 1;            // column-tile pairs accumulated here
    const E1Params& p = a.c;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sP = smem;                                   // halo patch [E_PP][16 channels]
    char* sD = sP + E_PP * E_ROWB;                     // pooled gradients [64 windows][128 channels] bf16
    char* sC = sD + E_DCH * 16;                        // codes [8 window rows][2][128 channels] u16
    char* sX = sC + 8 * 2 * 128 * 2;                   // fragments [16 steps][NCT][64 lanes][16 bytes]
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kt = blockIdx.y * 4 + wave;
    const bool active = WANT_A1 && kt * 32 < p.K;
    const bool gram = GRAM && blockIdx.y == 0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.x, 0, (unsigned)((size_t)p.N * p.H * p.W * E_ROWB), 0x00020000);
    const int PH = p.H / 2, PWd = p.W / 2;
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.dy, 0, (unsigned)((size_t)p.N * PH * PWd * p.K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rcd = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.codes, 0, (unsigned)((size_t)p.N * PH * p.tiles_x * 2 * p.K * 2), 0x00020000);
    auto load_dyt = [&](const TilePos& tq, bool live, uint4* regs) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < E_DLOADS; ++j) {
            const int c = threadIdx.x + j * E_NT;
            const int win = c >> 4, part = c & 15;
            const int wy = (tq.y0 >> 1) + (win >> 3), wx = (tq.x0 >> 1) + (win & 7);
            const bool ok = live && wy < PH && wx < PWd && blockIdx.y * 128 + part * 8 < p.K;
            const int off = ok ? (((tq.n * PH + wy) * PWd + wx) * p.K + blockIdx.y * 128 + part * 8) * 2 : (int)0x80000000;
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(rdy, off, 0, 0);
            regs[j] = *reinterpret_cast<const uint4*>(&v);
        }
    };
    // the tile's codes: 8 window rows x 2 window parities x 128 channels, one 16-byte chunk per thread
    auto load_codes = [&](const TilePos& tq, bool live) __attribute__((always_inline)) {
        const int c = threadIdx.x, rt = c >> 5, hh = (c >> 4) & 1, part = c & 15;
        const int wy = (tq.y0 >> 1) + rt, kk = blockIdx.y * 128 + part * 8;
        const bool ok = live && wy < PH && kk < p.K;
        const int off = ok ? (int)(code_index(tq.n, wy, tq.x0 / E_TW, hh, kk, PH, p.tiles_x, p.K) * 2) : (int)0x80000000;
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rcd, off, 0, 0);
        return *reinterpret_cast<const uint4*>(&v);
    };
    const int d_off = ((h * 128) + wave * 32 + r) * 2;    // + (rt * 8 + 2 q) * 256
    const int c_off = (h * 128 + wave * 32 + r) * 2;      // + rt * 512
    // gather base of the lane's column (tap, c) in each column tile, + 2 window columns for h = 1
    int g_off[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        int col = ct * 32 + r;
        if (col >= a.ncols) col = 0;
        const int tap = col / a.Creal, ch = col - tap * a.Creal;
        g_off[ct] = ((tap / 3) * E_PW + tap % 3) * E_ROWB + 2 * ch + 2 * h * E_ROWB;
    }

    f32x16 accA1[NCT], accG[NP];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int i = 0; i < 16; ++i) accA1[ct][i] = 0.f;
#pragma unroll
    for (int q = 0; q < NP; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) accG[q][i] = 0.f;
    float m1 = 0.f;

    uint4 pre[E_PLOADS], pred[E_DLOADS], prec;
    int t = blockIdx.x;
    TilePos tp = tile_pos(p, t < p.ntiles ? t : 0);
    load_patch(p, rs, tp, t < p.ntiles, pre);
    if constexpr (WANT_A1) {
        load_dyt(tp, t < p.ntiles, pred);
        prec = load_codes(tp, t < p.ntiles);
    }
    for (; t < p.ntiles; t += gridDim.x) {
        __syncthreads();                 // everyone is done with the previous tile's patch, gradients, codes, fragments
        store_patch<false>(sP, pre);
        if constexpr (WANT_A1) {
#pragma unroll
            for (int j = 0; j < E_DLOADS; ++j) *reinterpret_cast<uint4*>(sD + (threadIdx.x + j * E_NT) * 16) = pred[j];
            *reinterpret_cast<uint4*>(sC + threadIdx.x * 16) = prec;
        }
        __syncthreads();
        const TilePos cur = tp;
        const int tn = t + gridDim.x;
        tp = tile_pos(p, tn < p.ntiles ? tn : 0);
        load_patch(p, rs, tp, tn < p.ntiles, pre);
        if constexpr (WANT_A1) {
            load_dyt(tp, tn < p.ntiles, pred);
            prec = load_codes(tp, tn < p.ntiles);
        }
        const bool full = cur.y0 + E_TH <= p.H && cur.x0 + E_TW <= p.W;
        // ---- fragments of the tile: step = 2 rt + s2 (16 pixels: windows 4 s2 + {0, 2} + h of window
        // row rt), column tile ct; this wave builds the fragments of steps wave, wave + 4, wave + 8, wave + 12
        auto build = [&](auto FULL) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int step = wave + 4 * i;
                const int rt = step >> 1, s2 = step & 1;
                const char* prow = sP + 2 * rt * E_PW * E_ROWB;
                unsigned keep[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    // output pixels beyond the image do not exist: their patches must not enter the Gram matrix
                    const int win = 4 * s2 + 2 * (j >> 2), e = j & 3;
                    const int gy = cur.y0 + 2 * rt + (e >> 1), gx = cur.x0 + 2 * (win + h) + (e & 1);
                    keep[j] = (decltype(FULL)::value || (gy < p.H && gx < p.W)) ? 0xffffu : 0u;
                }
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    unsigned xs[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int win = 4 * s2 + 2 * (j >> 2), e = j & 3;     // + h via g_off
                        const int off = ((e >> 1) * E_PW + 2 * win + (e & 1)) * E_ROWB;
                        xs[j] = (unsigned)*reinterpret_cast<const unsigned short*>(prow + g_off[ct] + off) & keep[j];
                    }
                    uint4 xw;
                    xw.x = xs[0] | (xs[1] << 16); xw.y = xs[2] | (xs[3] << 16);
                    xw.z = xs[4] | (xs[5] << 16); xw.w = xs[6] | (xs[7] << 16);
                    *reinterpret_cast<uint4*>(sX + (step * NCT + ct) * 1024 + lane * 16) = xw;
                }
            }
        };
        if (full) build(std::true_type{});
        else build(std::false_type{});
        __syncthreads();
        // ---- A1 += dzm^T xp for this wave's channel tile
        if (active) {
#pragma unroll 2
            for (int rt = 0; rt < 8; ++rt) {
                const unsigned cw = *reinterpret_cast<const unsigned short*>(sC + c_off + rt * 512);
                unsigned dp[8];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const unsigned gc = *reinterpret_cast<const unsigned short*>(sD + d_off + (rt * 8 + 2 * g) * 256);
                    const unsigned code = (cw >> (3 * g)) & 7u, pos = code & 3u;
                    const unsigned g16 = (code & 4u) ? gc : 0u;          // ReLU: no gradient through a maximum <= 0
                    m1 += __uint_as_float(g16 << 16);
                    const unsigned ghi = g16 << 16;
                    dp[2 * g] = pos == 0u ? g16 : (pos == 1u ? ghi : 0u);
                    dp[2 * g + 1] = pos == 2u ? g16 : (pos == 3u ? ghi : 0u);
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x8 fd = *reinterpret_cast<const bf16x8*>(dp + 4 * s2);
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) {
                        const bf16x8 fx = *reinterpret_cast<const bf16x8*>(sX + ((2 * rt + s2) * NCT + ct) * 1024 + lane * 16);
                        accA1[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fd, fx, accA1[ct], 0, 0, 0);
                    }
                }
            }
        }
        // ---- G += xp^T xp: this wave takes steps 4 wave .. 4 wave + 3
        if constexpr (GRAM) if (gram) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int step = wave * 4 + q;
                bf16x8 fr[NCT];
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) fr[ct] = *reinterpret_cast<const bf16x8*>(sX + (step * NCT + ct) * 1024 + lane * 16);
#pragma unroll
                for (int ca = 0; ca < NCT; ++ca)
#pragma unroll
                    for (int cb = ca; cb < NCT; ++cb)
                        accG[gram_pair(ca, cb, NCT)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[ca], fr[cb], accG[gram_pair(ca, cb, NCT)], 0, 0, 0);
            }
        }
    }
    // ---- slab: every element written (zeros from idle waves), no atomics
    if constexpr (WANT_A1) {
        float* slab = a.slab + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * slab2_floats(NCT);
        constexpr int ncp = NCT * 32;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = wave * 32 + 8 * (i >> 2) + 4 * h + (i & 3);
                slab[(size_t)row * ncp + ct * 32 + r] = accA1[ct][i];
            }
        const float t1 = m1 + __shfl_xor(m1, 32, 64);
        if (h == 0) slab[(size_t)128 * ncp + wave * 32 + r] = t1;
    }
    // ---- Gram matrix of the workgroup: the four waves' parts added in wave order through LDS
    if constexpr (GRAM) if (gram) {
        float* red = reinterpret_cast<float*>(smem);          // [NP][16][64]
        for (int w2 = 0; w2 < 4; ++w2) {
            __syncthreads();
            if (wave == w2) {
#pragma unroll
                for (int q = 0; q < NP; ++q)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        float* e = red + (q * 16 + i) * 64 + lane;
                        *e = w2 == 0 ? accG[q][i] : *e + accG[q][i];
                    }
            }
        }
        __syncthreads();
        float* gs = a.gslab + (size_t)blockIdx.x * NP * 1024;
        for (int idx = threadIdx.x; idx < NP * 1024; idx += E_NT) {
            const int q = idx >> 10, i = (idx >> 6) & 15, l = idx & 63;
            const int m = 8 * (i >> 2) + 4 * (l >> 5) + (i & 3), n = l & 31;
            gs[q * 1024 + m * 32 + n] = red[idx];
        }
    }
}

// Sum of the workgroups' partial Gram matrices and of the input-sum slabs, in a fixed order
// (reproducible), in double.  out: [n Gram entries][144 input sums] = the "Gram record" that the
// statistics and the gradient finalize kernels read (SF_ENC1_GRAM_DOUBLES doubles at most).
// A workgroup folds 32 entries: eight threads per entry take an eighth of the slabs each (one
// thread per entry walked several hundred slabs with eight loads in flight: 58 us of latency
// for 13 MB), and the eight partial sums are added in order.
__global__ __launch_bounds__(256) void enc1_gram_fold_kernel(const float* __restrict__ gslab, int nslabs, int n,
                                                             const float* __restrict__ insums, double* __restrict__ out) {
    __shared__ double part[8][32];
    const int el = threadIdx.x & 31, pt = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + el;
    const bool live = e < n + 144;
    const int ee = live ? e : 0;
    const float* src = ee < n ? gslab + ee : insums + (ee - n);
    const int cnt = ee < n ? nslabs : E_SUMS_GRID;
    const size_t stride = ee < n ? (size_t)n : 144;
    const int per = (cnt + 7) / 8, lo = pt * per, hi = lo + per < cnt ? lo + per : cnt;
    double acc = 0.0;
    for (int i0 = lo; i0 < hi; i0 += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(i0 + u < hi ? i0 + u : i0) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += i0 + u < hi ? (double)v[u] : 0.0;
    }
    part[pt][el] = acc;
    __syncthreads();
    if (pt == 0 && live) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += part[k][el];
        out[e] = t;
    }
}

// S[(tap, c)] = sum over the output pixels of the input value under tap (ty, tx): every pixel of
// the image except the border row / column the tap never reaches (tap row 0 never reads the last
// image row, tap row 2 never the first; columns alike), from the folded input sums
// [9][16] = all, first row, last row, first column, last column, four corners.
__device__ __forceinline__ double patch_sum(const double* __restrict__ sums, int tap, int c) {
    const int ty = tap / 3, tx = tap % 3;
    const int rowq = ty == 0 ? 2 : 1, colq = tx == 0 ? 4 : 3;
    const int corq = ty == 0 ? (tx == 0 ? 8 : 7) : (tx == 0 ? 6 : 5);
    const double fr = ty != 1 ? 1.0 : 0.0, fc = tx != 1 ? 1.0 : 0.0, fk = (ty != 1 && tx != 1) ? 1.0 : 0.0;
    return sums[c] - fr * sums[rowq * 16 + c] - fc * sums[colq * 16 + c] + fk * sums[corq * 16 + c];
}
__device__ __forceinline__ double gram_at(const double* __restrict__ gram, int qa_col, int qb_col, int nct) {
    const int qa = qa_col >> 5, qm = qa_col & 31, cb = qb_col >> 5, cn = qb_col & 31;
    return qa <= cb ? gram[(size_t)gram_pair(qa, cb, nct) * 1024 + qm * 32 + cn]
                    : gram[(size_t)gram_pair(cb, qa, nct) * 1024 + cn * 32 + qm];
}

// BatchNorm statistics of the conv output without a pass over it: with z[p][k] = sum_col W[k][col] xp[p][col],
//     sum_p z = W[k] . S        and        sum_p z^2 = W[k] G W[k]^T.
// ws[k] = sum, ws[K + k] = sum of squares (the record sf_enc1_stats writes; here of the unrounded z).
__global__ __launch_bounds__(128) void enc1_stats_gram_kernel(const double* __restrict__ rec, const bf16* __restrict__ w, int K,
                                                              int nct, int Creal, double* __restrict__ ws) {
    __shared__ float wrow[128];
    __shared__ double red[2][128];
    const int ncols = 9 * Creal, k = blockIdx.x, col = threadIdx.x;
    const bool live = col < ncols;
    const int cc = live ? col : 0, tap = cc / Creal, c = cc - tap * Creal;
    const double* gram = rec;
    const double* sums = rec + (size_t)gram_pairs(nct) * 1024;
    const float wv = live ? (float)w[((size_t)k * 9 + tap) * 16 + c] : 0.f;
    wrow[col] = wv;
    __syncthreads();
    double t = 0.0;
    if (live)
        for (int q = 0; q < ncols; ++q) t += (double)wrow[q] * gram_at(gram, q, col, nct);
    red[0][col] = live ? (double)wv * patch_sum(sums, tap, c) : 0.0;
    red[1][col] = live ? (double)wv * t : 0.0;
    __syncthreads();
    if (col == 0) {
        double s0 = 0.0, s1 = 0.0;
        for (int q = 0; q < ncols; ++q) { s0 += red[0][q]; s1 += red[1][q]; }
        ws[k] = s0;
        ws[K + k] = s1 > 0.0 ? s1 : 0.0;
    }
}

// Sums of the input the patch sums S are made of: per 8-channel half, over all images, the sum of
// every pixel (0), of the first / last row (1, 2), of the first / last column (3, 4) and of the
// four corners (5: (0,0), 6: (0,W-1), 7: (H-1,0), 8: (H-1,W-1)).  One slab [9][16] per workgroup.
__global__ __launch_bounds__(E_NT) void enc1_input_sums_kernel(const bf16* __restrict__ x, float* __restrict__ out, int N,
                                                               int H, int W) {
    __shared__ float red[9][16][E_NT / 64];
    const int half = threadIdx.x & 1, pl = threadIdx.x >> 1;     // 128 pixels per pass
    float acc[9][8];
#pragma unroll
    for (int q = 0; q < 9; ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[q][e] = 0.f;
    const int rows = N * H;
    constexpr int RU = 4;                            // rows per trip, all loads issued first
    for (int row0 = blockIdx.x * RU; row0 < rows; row0 += gridDim.x * RU) {
        for (int px = pl; px < W; px += E_NT / 2) {
            bf16x8 v[RU];
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int row = row0 + u < rows ? row0 + u : rows - 1;
                v[u] = *reinterpret_cast<const bf16x8*>(x + ((size_t)row * W + px) * 16 + half * 8);
            }
            const bool left = px == 0, right = px == W - 1;
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                if (row0 + u >= rows) break;
                const int y = (row0 + u) % H;
                const bool top = y == 0, bot = y == H - 1;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[0][e] += (float)v[u][e];
                if (top || bot || left || right) {       // 2 % of the pixels
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float f = (float)v[u][e];
                        acc[1][e] += top ? f : 0.f;
                        acc[2][e] += bot ? f : 0.f;
                        acc[3][e] += left ? f : 0.f;
                        acc[4][e] += right ? f : 0.f;
                        acc[5][e] += top && left ? f : 0.f;
                        acc[6][e] += top && right ? f : 0.f;
                        acc[7][e] += bot && left ? f : 0.f;
                        acc[8][e] += bot && right ? f : 0.f;
                    }
                }
            }
        }
    }
    // lanes of equal parity hold the same channel half: reduce over 32 lanes, then over the waves
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 9; ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = acc[q][e];
#pragma unroll
            for (int o = 32; o >= 2; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane < 2) red[q][lane * 8 + e][wave] = v;
        }
    __syncthreads();
    for (int i = threadIdx.x; i < 9 * 16; i += E_NT) {
        float v = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < E_NT / 64; ++w2) v += red[i / 16][i % 16][w2];
        out[(size_t)blockIdx.x * 144 + i] = v;
    }
}

// dW[k][c][tap] = scale[k] * (A1 - m1/M * S - m2/M * invstd[k] * (A2 - mean[k] * S))   (training)
//              = scale[k] * A1                                                          (eval)
// dgamma = m2, dbeta = m1.  Slabs are added in a fixed order (reproducible).  S[(tap, c)] = all - excluded border row - excluded border
// column + their common corner (tap row 0 never reaches the last image row, tap row 2 never the
// first; columns alike).
__global__ __launch_bounds__(1024) void enc1_bwd_finalize_kernel(const float* __restrict__ slabs, int nslabs,
                                                                const float* __restrict__ insums, int K, int nct,
                                                                int Creal, double inv_m, int training,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, float* __restrict__ dw,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta) {
    // one workgroup per channel k; thread = (column, one of 8 slab parts): coalesced slab rows
    constexpr int PARTS = 8;
    __shared__ double sh[PARTS][5][128];
    const int ncp = nct * 32, ncols = 9 * Creal;
    const size_t sf_ = slab_floats(K, nct);
    const int k = blockIdx.x, col = threadIdx.x & 127, part = threadIdx.x >> 7;
    const int grp = k / 128;                 // the slab group (blockIdx.y) that owns this channel
    const bool live = col < ncols;
    const int cc = live ? col : 0;
    const int tap = cc / Creal, c = cc - tap * Creal, ty = tap / 3, tx = tap % 3;
    double a1 = 0.0, a2 = 0.0, m1 = 0.0, m2 = 0.0, s = 0.0;
    const float* base = slabs + (size_t)grp * nslabs * sf_;
    // batches of 8 slabs: all loads of a batch are issued before the first add
    for (int i0 = part; i0 < nslabs; i0 += PARTS * 8) {
        float v1[8], v2[8], v3[8], v4[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * PARTS;
            const float* sl = base + (size_t)(i < nslabs ? i : i0) * sf_;
            v1[u] = sl[(size_t)k * ncp + cc];
            v2[u] = sl[(size_t)(K + k) * ncp + cc];
            v3[u] = sl[(size_t)2 * K * ncp + k];
            v4[u] = sl[(size_t)2 * K * ncp + K + k];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool ok = i0 + u * PARTS < nslabs;
            a1 += ok ? (double)v1[u] : 0.0;
            a2 += ok ? (double)v2[u] : 0.0;
            m1 += ok ? (double)v3[u] : 0.0;
            m2 += ok ? (double)v4[u] : 0.0;
        }
    }
    // S: the excluded row / column / corner enter with a 0 / 1 factor (unconditional loads)
    const int rowq = ty == 0 ? 2 : 1, colq = tx == 0 ? 4 : 3;
    const int corq = ty == 0 ? (tx == 0 ? 8 : 7) : (tx == 0 ? 6 : 5);
    const double fr = ty != 1 ? 1.0 : 0.0, fc = tx != 1 ? 1.0 : 0.0, fk = (ty != 1 && tx != 1) ? 1.0 : 0.0;
    for (int i0 = part; i0 < E_SUMS_GRID; i0 += PARTS * 4) {
        float q0[4], q1[4], q2[4], q3[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * PARTS;
            const float* q = insums + (size_t)(i < E_SUMS_GRID ? i : i0) * 144 + c;
            q0[u] = q[0]; q1[u] = q[rowq * 16]; q2[u] = q[colq * 16]; q3[u] = q[corq * 16];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u * PARTS < E_SUMS_GRID) s += (double)q0[u] - fr * (double)q1[u] - fc * (double)q2[u] + fk * (double)q3[u];
    }
    sh[part][0][col] = a1; sh[part][1][col] = a2; sh[part][2][col] = m1; sh[part][3][col] = m2; sh[part][4][col] = s;
    __syncthreads();
    if (part != 0 || !live) return;
#pragma unroll
    for (int q = 1; q < PARTS; ++q) {
        a1 += sh[q][0][col]; a2 += sh[q][1][col]; m1 += sh[q][2][col]; m2 += sh[q][3][col]; s += sh[q][4][col];
    }
    double g = a1;
    if (training) g = a1 - m1 * inv_m * s - m2 * inv_m * (double)invstd[k] * (a2 - (double)mean[k] * s);
    dw[((size_t)k * Creal + c) * 9 + tap] = (float)((double)scale[k] * g);
    if (col == 0) { dgamma[k] = (float)m2; dbeta[k] = (float)m1; }
}

// The same gradients from the code-routed kernel's sums (enc1_bwd2_kernel): A1 and sum dzm from the
// slabs, S from the input sums, and through the layer's own weights
//     sum dzm * z = sum_col W[k][col] A1[k][col],     A2[k][col] = sum_col' W[k][col'] G[col'][col].
__global__ __launch_bounds__(1024) void enc1_bwd2_finalize_kernel(const float* __restrict__ slabs, int nslabs,
                                                                 const double* __restrict__ rec, const bf16* __restrict__ w,
                                                                 int K, int nct, int Creal, double inv_m, int training,
                                                                 const float* __restrict__ scale,
                                                                 const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd, float* __restrict__ dw,
                                                                 float* __restrict__ dgamma, float* __restrict__ dbeta) {
    constexpr int PARTS = 8;
    __shared__ double sh[PARTS][2][128];
    __shared__ double prod[128];
    __shared__ float wrow[128];
    const int ncp = nct * 32, ncols = 9 * Creal;
    const size_t sf_ = slab2_floats(nct);
    const int k = blockIdx.x, col = threadIdx.x & 127, part = threadIdx.x >> 7;
    const int grp = k / 128, kk = k - grp * 128;
    const bool live = col < ncols;
    const int cc = live ? col : 0;
    const int tap = cc / Creal, c = cc - tap * Creal;
    const double* gram = rec;
    const double* sums = rec + (size_t)gram_pairs(nct) * 1024;
    double a1 = 0.0, m1 = 0.0;
    const float* base = slabs + (size_t)grp * nslabs * sf_;
    for (int i0 = part; i0 < nslabs; i0 += PARTS * 8) {
        float v1[8], v3[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * PARTS;
            const float* sl = base + (size_t)(i < nslabs ? i : i0) * sf_;
            v1[u] = sl[(size_t)kk * ncp + cc];
            v3[u] = sl[(size_t)128 * ncp + kk];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool ok = i0 + u * PARTS < nslabs;
            a1 += ok ? (double)v1[u] : 0.0;
            m1 += ok ? (double)v3[u] : 0.0;
        }
    }
    sh[part][0][col] = a1; sh[part][1][col] = m1;
    __syncthreads();
    if (part == 0) {
#pragma unroll
        for (int q = 1; q < PARTS; ++q) { a1 += sh[q][0][col]; m1 += sh[q][1][col]; }
        const float wv = live ? (float)w[((size_t)k * 9 + tap) * 16 + c] : 0.f;
        wrow[col] = wv;
        prod[col] = live ? (double)wv * a1 : 0.0;
    }
    __syncthreads();
    if (part != 0 || !live) return;
    double m2 = 0.0;
    for (int q = 0; q < ncols; ++q) m2 += prod[q];                  // sum dzm * z, fixed order
    m2 = (double)invstd[k] * (m2 - (double)mean[k] * m1);           // sum dzm * xhat
    double g = a1;
    if (training) {
        const double s = patch_sum(sums, tap, c);
        double a2 = 0.0;
        for (int q = 0; q < ncols; ++q) a2 += (double)wrow[q] * gram_at(gram, q, col, nct);
        g = a1 - m1 * inv_m * s - m2 * inv_m * (double)invstd[k] * (a2 - (double)mean[k] * s);
    }
    dw[((size_t)k * Creal + c) * 9 + tap] = (float)((double)scale[k] * g);
    if (col == 0) { dgamma[k] = (float)m2; dbeta[k] = (float)m1; }
}

int fill(E1Params& p, const char* who, const void* x, const void* w, int N, int H, int W, int K) {
    if (!x || !w || N <= 0 || H < 2 || W < 2 || (K != 128 && K != 256) ||
        (long long)N * H * W * E_ROWB >= 0x7fffffffLL) {
        sf_set_error("%s: unsupported shape N=%d H=%d W=%d K=%d (K must be 128 or 256)", who, N, H, W, K);
        return SF_EINVAL;
    }
    p.x = (const bf16*)x; p.w = (const bf16*)w;
    p.N = N; p.H = H; p.W = W; p.K = K;
    p.tiles_x = (W + E_TW - 1) / E_TW;
    p.tiles_y = (H + E_TH - 1) / E_TH;
    p.ntiles = N * p.tiles_x * p.tiles_y;
    p.ws = nullptr; p.scale = p.shift = nullptr; p.y = nullptr; p.codes = nullptr;
    return SF_OK;
}

template <int MODE>
int launch(const E1Params& p, hipStream_t s, const char* who) {
    const size_t lds = (size_t)9 * p.K * E_ROWB + (size_t)E_PP * E_ROWB + (MODE == E_STATS ? 0 : 0);
    int grid = p.ntiles < 768 ? p.ntiles : 768;
    void (*kern)(E1Params) = nullptr;
    kern = p.K == 128 ? enc1_kernel<MODE, 4> : enc1_kernel<MODE, 8>;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        sf_set_error("%s: cannot raise the dynamic LDS limit to %zu bytes", who, lds);
        return SF_ELAUNCH;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(E_NT), lds, s, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { sf_set_error("%s: launch failed: %s", who, hipGetErrorString(e)); return SF_ELAUNCH; }
    return SF_OK;
}

}  // namespace

extern "C" int sf_enc1_stats(const void* x, const void* w, double* ws, int N, int H, int W, int K, void* stream) {
    E1Params p;
    if (int rc = fill(p, "sf_enc1_stats", x, w, N, H, W, K)) return rc;
    SF_REQUIRE(ws, "sf_enc1_stats: null workspace");
    SF_REQUIRE(8 * K * sizeof(float) <= (size_t)E_PP * E_ROWB, "sf_enc1_stats: K too large for the reduction buffer");
    p.ws = ws;
    if (hipMemsetAsync(ws, 0, sizeof(double) * 2 * K, (hipStream_t)stream) != hipSuccess) {
        sf_set_error("sf_enc1_stats: memset failed");
        return SF_ELAUNCH;
    }
    return launch<E_STATS>(p, (hipStream_t)stream, "sf_enc1_stats");
}

extern "C" size_t sf_enc1_codes_bytes(int N, int H, int W, int K) {
    return (size_t)N * (H / 2) * ((W + E_TW - 1) / E_TW) * 2 * K * sizeof(unsigned short);
}

extern "C" int sf_enc1_fwd(const void* x, const void* w, const float* scale, const float* shift, void* y, void* codes,
                           int N, int H, int W, int K, void* stream) {
    E1Params p;
    if (int rc = fill(p, "sf_enc1_fwd", x, w, N, H, W, K)) return rc;
    SF_REQUIRE(scale && shift && y, "sf_enc1_fwd: null argument");
    SF_REQUIRE(!codes || sf_enc1_codes_bytes(N, H, W, K) < 0x7fffffffULL, "sf_enc1_fwd: the routing codes exceed 2 GiB");
    p.scale = scale; p.shift = shift; p.y = (bf16*)y; p.codes = (unsigned short*)codes;
    return launch<E_FWD>(p, (hipStream_t)stream, "sf_enc1_fwd");
}

extern "C" size_t sf_enc1_bwd_workspace_bytes(int N, int H, int W, int K, int Creal) {
    const int nct = (9 * Creal + 31) / 32;
    const long long ntiles = (long long)N * ((W + E_TW - 1) / E_TW) * ((H + E_TH - 1) / E_TH);
    const long long gx = ntiles < 512 ? ntiles : 512;
    const size_t gy = (size_t)((K + 127) / 128);
    // recomputing kernel: slabs + input sums; code-routed kernel: slabs + Gram slabs + folded Gram matrix (doubles) + input sums
    const size_t a = slab_floats(K, nct) * (size_t)gx * gy + (size_t)E_SUMS_GRID * 144;
    const size_t b = slab2_floats(nct) * (size_t)gx * gy + (size_t)gram_pairs(nct) * 1024 * (size_t)gx + 2 * (size_t)SF_ENC1_GRAM_DOUBLES +
                     (size_t)E_SUMS_GRID * 144;
    return sizeof(float) * (a > b ? a : b);
}

// launch of enc1_bwd2_kernel<nct, MODE> (its LDS limit raised once per device and variant)
template <int MODE>
static int launch_bwd2(const E1Bwd2& a, int nct, int gx, int gy, hipStream_t s, const char* who) {
    const size_t lds = (size_t)E_PP * E_ROWB + (size_t)E_DCH * 16 + 8 * 2 * 128 * 2 + (size_t)16 * nct * 1024;
    void (*kern)(E1Bwd2) = nct == 1 ? enc1_bwd2_kernel<1, MODE> : nct == 2 ? enc1_bwd2_kernel<2, MODE> : enc1_bwd2_kernel<3, MODE>;
    static DeviceOnce attr_set[3];
    DeviceOnce& once = attr_set[nct - 1];
    if (!once.done()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            sf_set_error("%s: cannot raise the dynamic LDS limit to %zu bytes", who, lds);
            return SF_ELAUNCH;
        }
        once.mark();
    }
    hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(E_NT), lds, s, a);
    return SF_OK;
}

// Gram record of the input patches: partial matrices per workgroup, the input sums, both folded
static int gram_record(const E1Params& c, int Creal, bool accumulate_here, float* gslab, float* insums, double* rec, int gx,
                       hipStream_t s, const char* who) {
    const int nct = (9 * Creal + 31) / 32, np = gram_pairs(nct);
    if (accumulate_here) {
        E1Bwd2 a;
        a.c = c; a.dy = nullptr; a.codes = nullptr; a.slab = nullptr; a.gslab = gslab;
        a.Creal = Creal; a.ncols = 9 * Creal; a.gram = 1;
        if (int rc = launch_bwd2<2>(a, nct, gx, 1, s, who)) return rc;
    }
    hipLaunchKernelGGL(enc1_input_sums_kernel, dim3(E_SUMS_GRID), dim3(E_NT), 0, s, c.x, insums, c.N, c.H, c.W);
    hipLaunchKernelGGL(enc1_gram_fold_kernel, dim3((np * 1024 + 144 + 31) / 32), dim3(256), 0, s, gslab, gx, np * 1024, insums, rec);
    return SF_OK;
}

static int gram_shape(E1Params& p, const char* who, const void* x, int N, int H, int W, int Creal) {
    if (!x || N <= 0 || H < 2 || W < 2 || Creal < 1 || 9 * Creal > 96 || (long long)N * H * W * E_ROWB >= 0x7fffffffLL) {
        sf_set_error("%s: unsupported shape N=%d H=%d W=%d with %d real channels (1..10)", who, N, H, W, Creal);
        return SF_EINVAL;
    }
    p.x = (const bf16*)x; p.w = nullptr;
    p.N = N; p.H = H; p.W = W; p.K = 128;
    p.tiles_x = (W + E_TW - 1) / E_TW;
    p.tiles_y = (H + E_TH - 1) / E_TH;
    p.ntiles = N * p.tiles_x * p.tiles_y;
    p.ws = nullptr; p.scale = p.shift = nullptr; p.y = nullptr; p.codes = nullptr;
    return SF_OK;
}

extern "C" size_t sf_enc1_gram_workspace_bytes(int N, int H, int W, int Creal) {
    const int nct = (9 * Creal + 31) / 32;
    const long long ntiles = (long long)N * ((W + E_TW - 1) / E_TW) * ((H + E_TH - 1) / E_TH);
    const long long gx = ntiles < 512 ? ntiles : 512;
    return sizeof(float) * ((size_t)gram_pairs(nct) * 1024 * (size_t)gx + (size_t)E_SUMS_GRID * 144);
}

extern "C" int sf_enc1_gram(const void* x, int N, int H, int W, int Creal, double* rec, float* ws, size_t ws_bytes,
                            void* stream) {
    E1Params p;
    if (int rc = gram_shape(p, "sf_enc1_gram", x, N, H, W, Creal)) return rc;
    SF_REQUIRE(rec && ws, "sf_enc1_gram: null argument");
    SF_REQUIRE(ws_bytes >= sf_enc1_gram_workspace_bytes(N, H, W, Creal), "sf_enc1_gram: workspace too small");
    const int nct = (9 * Creal + 31) / 32, gx = p.ntiles < 512 ? p.ntiles : 512;
    float* insums = ws + (size_t)gram_pairs(nct) * 1024 * gx;
    if (int rc = gram_record(p, Creal, true, ws, insums, rec, gx, (hipStream_t)stream, "sf_enc1_gram")) return rc;
    SF_LAUNCH_CHECK("sf_enc1_gram");
    return SF_OK;
}

extern "C" int sf_enc1_stats_gram(const void* w, const double* rec, int K, int Creal, double* ws, void* stream) {
    SF_REQUIRE(w && rec && ws && (K == 128 || K == 256) && Creal >= 1 && 9 * Creal <= 96, "sf_enc1_stats_gram: bad arguments");
    hipLaunchKernelGGL(enc1_stats_gram_kernel, dim3(K), dim3(128), 0, (hipStream_t)stream, rec, (const bf16*)w, K,
                       (9 * Creal + 31) / 32, Creal, ws);
    SF_LAUNCH_CHECK("sf_enc1_stats_gram");
    return SF_OK;
}

// the code-routed path (enc1_bwd2_kernel): up to three column tiles (at most 10 real input channels)
static int enc1_bwd_routed(const E1Params& c, const void* dy, const void* codes, const double* rec, const float* scale,
                           const float* mean, const float* invstd, int training, float* dw, float* dgamma, float* dbeta,
                           float* ws, int Creal, hipStream_t s) {
    E1Bwd2 a;
    a.c = c;
    a.dy = (const bf16*)dy; a.codes = (const unsigned short*)codes;
    a.Creal = Creal; a.ncols = 9 * Creal;
    const int nct = (a.ncols + 31) / 32, np = gram_pairs(nct);
    const int gx = c.ntiles < 512 ? c.ntiles : 512, gy = (c.K + 127) / 128;
    const bool here = training && !rec;             // the Gram record has to be made in this call
    a.gram = here ? 1 : 0;
    a.slab = ws;
    a.gslab = ws + slab2_floats(nct) * (size_t)gx * (size_t)gy;
    double* own = reinterpret_cast<double*>(a.gslab + (size_t)gx * np * 1024);      // (offsets are multiples of 128 floats: aligned)
    float* insums = reinterpret_cast<float*>(own + SF_ENC1_GRAM_DOUBLES);
    if (int rc = here ? launch_bwd2<1>(a, nct, gx, gy, s, "sf_enc1_bwd") : launch_bwd2<0>(a, nct, gx, gy, s, "sf_enc1_bwd")) return rc;
    if (here) {
        if (int rc = gram_record(c, Creal, false, a.gslab, insums, own, gx, s, "sf_enc1_bwd")) return rc;
        rec = own;
    }
    hipLaunchKernelGGL(enc1_bwd2_finalize_kernel, dim3(c.K), dim3(1024), 0, s, ws, gx, rec, c.w, c.K, nct, Creal,
                       1.0 / ((double)c.N * c.H * c.W), training, scale, mean, invstd, dw, dgamma, dbeta);
    SF_LAUNCH_CHECK("sf_enc1_bwd");
    return SF_OK;
}

extern "C" int sf_enc1_bwd(const void* x, const void* w, const void* dy, const void* codes, const double* gram,
                           const float* scale, const float* shift, const float* mean, const float* invstd, int training,
                           float* dw, float* dgamma, float* dbeta, float* ws, size_t ws_bytes, int N, int H, int W, int K,
                           int Creal, void* stream) {
    E1Bwd a;
    if (int rc = fill(a.c, "sf_enc1_bwd", x, w, N, H, W, K)) return rc;
    SF_REQUIRE(dy && scale && shift && mean && invstd && dw && dgamma && dbeta && ws, "sf_enc1_bwd: null argument");
    SF_REQUIRE(Creal >= 1 && Creal <= 14, "sf_enc1_bwd: 1..14 real input channels supported (got %d)", Creal);
    SF_REQUIRE((long long)N * (H / 2) * (W / 2) * K * 2 < 0x7fffffffLL, "sf_enc1_bwd: pooled gradient exceeds 2 GiB");
    const size_t need = sf_enc1_bwd_workspace_bytes(N, H, W, K, Creal);
    SF_REQUIRE(ws_bytes >= need, "sf_enc1_bwd: workspace too small (%zu < %zu)", ws_bytes, need);
    SF_REQUIRE(((uintptr_t)ws % 8) == 0, "sf_enc1_bwd: the workspace must be 8-byte aligned");
    if (codes && 9 * Creal <= 96 && sf_enc1_codes_bytes(N, H, W, K) < 0x7fffffffULL) {
        a.c.scale = scale; a.c.shift = shift;
        return enc1_bwd_routed(a.c, dy, codes, gram, scale, mean, invstd, training, dw, dgamma, dbeta, ws, Creal, (hipStream_t)stream);
    }
    a.c.scale = scale; a.c.shift = shift;
    a.dy = (const bf16*)dy; a.mean = mean; a.invstd = invstd; a.slab = ws;
    a.Creal = Creal; a.ncols = 9 * Creal;
    const int nct = (a.ncols + 31) / 32;
    const int gx = a.c.ntiles < 512 ? a.c.ntiles : 512, gy = (K + 127) / 128;
    float* insums = ws + slab_floats(K, nct) * (size_t)gx * (size_t)gy;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = (size_t)E_PP * E_ROWB + (size_t)9 * 128 * E_ROWB + (size_t)E_DCH * 16;
    void (*kern)(E1Bwd) = nct == 1 ? enc1_bwd_kernel<1> : nct == 2 ? enc1_bwd_kernel<2> : nct == 3 ? enc1_bwd_kernel<3> : enc1_bwd_kernel<4>;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        sf_set_error("sf_enc1_bwd: cannot raise the dynamic LDS limit to %zu bytes", lds);
        return SF_ELAUNCH;
    }
    hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(E_NT), lds, s, a);
    hipLaunchKernelGGL(enc1_input_sums_kernel, dim3(E_SUMS_GRID), dim3(E_NT), 0, s, (const bf16*)x, insums, N, H, W);
    const int total = K * a.ncols;
    (void)total;
    hipLaunchKernelGGL(enc1_bwd_finalize_kernel, dim3(K), dim3(1024), 0, s, ws, gx, insums, K, nct, Creal,
                       1.0 / ((double)N * H * W), training, scale, mean, invstd, dw, dgamma, dbeta);
    SF_LAUNCH_CHECK("sf_enc1_bwd");
    return SF_OK;
}
