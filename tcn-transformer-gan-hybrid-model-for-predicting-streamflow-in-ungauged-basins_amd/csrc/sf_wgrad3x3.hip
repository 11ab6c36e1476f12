// Convolution weight gradients, bf16 NHWC, for the two filter geometries of the model:
// 3x3 stride 1 pad 1 (encoder, decoder) and 4x4 stride 2 pad 1 (PatchGAN).
//
//   dW[k][t][c] = sum_pixels dOut[pixel][k] * In[STRIDE*pixel + tap_t - 1][c]
//
// A workgroup (8 waves) owns a 64 (output channels) x 64 (input channels) block of dW for ALL
// taps and streams a slice of the spatial tiles: per tile it stages the dOut tile and the halo
// patch of the input in LDS once; each wave keeps one 32x32 accumulator tile per tap it owns in
// registers across the whole slice, so nothing is accumulated in memory until the end.  Both
// MFMA operands are reduction-major in LDS ([pixel][channel]) and are read transposed with
// ds_read_b64_tr_b16.  LDS rows are padded so that the four pixel rows of one transposed read
// fall into the four 64-byte quarters of the 256-byte bank row (192-byte rows for consecutive
// pixels, 160-byte rows for the stride-2 patch): conflict-free, and a tap is a constant byte
// offset from the lane's base address.  The next tile's global loads are in flight (in
// registers) while the current tile's MFMAs run; two waves per SIMD hide the LDS latency.
// Per-slice partial results go to slabs; a second kernel sums the slabs in a fixed order
// (bitwise reproducible, no atomics).
//   3x3: waves = (k half, c half) x (even / odd reduction steps), 9 accumulator tiles each.
//   4x4: waves = (k half, c half) x (taps 0-7 / 8-15), 8 accumulator tiles each.
#include "sf_common.h"
#include "../../include/sfhip.h"
#include <atomic>

namespace {

constexpr int NTW = 512;
constexpr int SA = 192;  // bytes per dOut row in LDS

template <int TAPS, int STRIDE> struct WCfg;
template <> struct WCfg<16, 2> {
    static constexpr int R = 4, TPMAX = 128, PPMAX = 864, SB = 160, TPW = 8, KSPLIT = 1;
};

struct WPParams {
    const bf16* dout;
    const bf16* x;
    float* ws;
    int N, IH, IW, OH, OW, Kd, Cx;
    int TH, TW, tiles_x, tiles_y, ntiles, tiles_per_slice;
    int kt_tiles, ct_tiles, K64, C64, nslices;
    int qslot;      // slot of the list counters when tiles can be stolen, -1: fixed slices
    int shirk;      // self-test of the stealing: odd workers leave their whole list to the others
};

// Tile scheduling of wgrad3_dma_kernel.  0 (default): every workgroup streams a fixed slice of the
// tiles: the slab sums are bit-reproducible.  1: the slice is only the workgroup's own list; a
// workgroup that has finished its list takes tiles from the lists of the others (same XCD, same
// weight block), so a workgroup that starts late or shares its CU with another kernel (RCCL's, in
// data-parallel training) gets help instead of stretching the launch to twice its time.  Undisturbed,
// nobody steals and the partition is the fixed one; otherwise the partition of the sum varies from
// launch to launch (last-bit differences).
std::atomic<int> g_dynamic_tiles{0};
constexpr int STEAL_WORKERS = 512;      // workgroups of one launch at most
struct StealQueue {
    int next[STEAL_WORKERS];            // per workgroup: the next untaken position of its list
    int done;                           // workgroups that have left
    int pad[3];
};
__device__ StealQueue g_wqueues[SF_QUEUE_SLOTS];    // zero between launches: the last workgroup to leave resets its slot
StreamSlots g_wslots;

typedef bf16x4 __attribute__((address_space(3))) * lds_tr_p;

__device__ __forceinline__ bf16x8 tr_pair(const char* a0, const char* a1) {
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_tr_p)a0);
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_tr_p)a1);
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

template <int TAPS, int STRIDE>
__global__ __launch_bounds__(NTW, 2) void wgrad_patch_kernel(WPParams p) {
    using Cf = WCfg<TAPS, STRIDE>;
    constexpr int R = Cf::R, SB = Cf::SB, TPW = Cf::TPW, KSPLIT = Cf::KSPLIT;
    constexpr int ND = Cf::TPMAX * 8 / NTW;
    constexpr int NX = (Cf::PPMAX * 8 + NTW - 1) / NTW;
    constexpr int NSTEPS = Cf::TPMAX / 16;
    extern __shared__ __attribute__((aligned(16))) char smw[];
    char* sD = smw;                       // [TPMAX][SA]
    char* sX = smw + Cf::TPMAX * SA;      // [PPMAX][SB]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kb = wave & 1, cb = (wave >> 1) & 1, grp = wave >> 2;
    const int r = lane & 31, h = lane >> 5;
    const int qq4 = (lane & 15) >> 2, csel = (lane >> 4) & 1, p4 = lane & 3;

    int b = blockIdx.x;
    const int ct = b % p.ct_tiles; b /= p.ct_tiles;
    const int kt = b % p.kt_tiles;
    const int slice = b / p.kt_tiles;
    const int PW = (p.TW - 1) * STRIDE + R, PH = (p.TH - 1) * STRIDE + R;
    const int PP = PH * PW, TP = p.TH * p.TW;

    // ---- tile-invariant staging slots (16-byte chunk cc of pixel row (tid + i*NTW) >> 3)
    const int cc = tid & 7;
    int d_rel[ND], d_lyx[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        const int pix = (tid + i * NTW) >> 3;
        const int ly = pix / p.TW, lx = pix - ly * p.TW;
        d_rel[i] = ly * p.OW + lx;
        d_lyx[i] = pix < TP ? ((ly << 16) | lx) : -1;
    }
    int x_rel[NX], x_pyx[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int pp = (tid + i * NTW) >> 3;
        const int py = pp / PW, px = pp - py * PW;
        x_rel[i] = py * p.IW + px;
        x_pyx[i] = pp < PP ? ((py << 16) | px) : -1;
    }
    const bool d_col_ok = kt * 64 + cc * 8 < p.Kd;
    const bool x_col_ok = ct * 64 + cc * 8 < p.Cx;

    uint4 rd[ND], rx[NX];
    auto load_tile = [&](int tile) {
        const int txi = tile % p.tiles_x;
        const int rest = tile / p.tiles_x;
        const int tyi = rest % p.tiles_y;
        const int n = rest / p.tiles_y;
        const int oy0 = tyi * p.TH, ox0 = txi * p.TW;
        const int iy0 = oy0 * STRIDE - 1, ix0 = ox0 * STRIDE - 1;
        const long long based = ((long long)n * p.OH + oy0) * p.OW + ox0;
        const long long basex = ((long long)n * p.IH + iy0) * p.IW + ix0;
        // buffer loads: 32-bit byte offsets into the tile's image, out-of-range offsets return
        // zeros (padding, partial tiles) -- no branch around a load
        const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.dout + (size_t)n * p.OH * p.OW * p.Kd), 0, (unsigned)(p.OH * p.OW * p.Kd * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.x + (size_t)n * p.IH * p.IW * p.Cx), 0, (unsigned)(p.IH * p.IW * p.Cx * 2), 0x00020000);
        const int doff = ((oy0 * p.OW + ox0) * p.Kd + kt * 64 + cc * 8) * 2;
        const int xoff = ((iy0 * p.IW + ix0) * p.Cx + ct * 64 + cc * 8) * 2;
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int ly = d_lyx[i] >> 16, lx = d_lyx[i] & 0xffff;
            const bool ok = d_lyx[i] >= 0 && d_col_ok && oy0 + ly < p.OH && ox0 + lx < p.OW;
            const v4u v = __builtin_amdgcn_raw_buffer_load_b128(drs, ok ? doff + d_rel[i] * p.Kd * 2 : (int)0x80000000, 0, 0);
            rd[i] = make_uint4(v[0], v[1], v[2], v[3]);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int gy = iy0 + (x_pyx[i] >> 16), gx = ix0 + (x_pyx[i] & 0xffff);
            const bool ok = x_pyx[i] >= 0 && x_col_ok && gy >= 0 && gy < p.IH && gx >= 0 && gx < p.IW;
            const v4u v = __builtin_amdgcn_raw_buffer_load_b128(xrs, ok ? xoff + x_rel[i] * p.Cx * 2 : (int)0x80000000, 0, 0);
            rx[i] = make_uint4(v[0], v[1], v[2], v[3]);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int pix = (tid + i * NTW) >> 3;
            *reinterpret_cast<uint4*>(sD + pix * SA + cc * 16) = rd[i];
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int pp = (tid + i * NTW) >> 3;
            if (pp < Cf::PPMAX) *reinterpret_cast<uint4*>(sX + pp * SB + cc * 16) = rx[i];
        }
    };

    // byte offset of tap j of this wave (wave-uniform)
    int toffB[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int t = (KSPLIT == 1 ? grp * TPW : 0) + j;
        toffB[j] = ((t / R) * PW + (t % R)) * SB;
    }

    f32x16 acc[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    const int tile_begin = slice * p.tiles_per_slice;
    int tile_end = tile_begin + p.tiles_per_slice;
    if (tile_end > p.ntiles) tile_end = p.ntiles;
    const int lane_col = csel * 32 + p4 * 8;  // byte offset inside a 64-byte block of 32 channels
    const char* aBase = sD + kb * 64 + lane_col;
    const char* bBase = sX + cb * 64 + lane_col;
    const int s_first = KSPLIT == 2 ? grp : 0;

    if (tile_begin < tile_end) load_tile(tile_begin);
    for (int tile = tile_begin; tile < tile_end; ++tile) {
        if (tile > tile_begin) __syncthreads();
        store_tile();
        __syncthreads();
        if (tile + 1 < tile_end) load_tile(tile + 1);
        // rows this lane addresses at reduction step s: q = 16 s + 8 h + 4 half + qq4
        int q0 = 16 * s_first + 8 * h + qq4, q1 = q0 + 4;
        int ly0 = q0 / p.TW, lx0 = q0 - ly0 * p.TW;
        int ly1 = q1 / p.TW, lx1 = q1 - ly1 * p.TW;
#pragma unroll 2
        for (int s = s_first; s < NSTEPS; s += KSPLIT) {
            const bf16x8 fa = tr_pair(aBase + q0 * SA, aBase + q1 * SA);
            const int pp0 = q0 < TP ? (ly0 * STRIDE) * PW + lx0 * STRIDE : 0;
            const int pp1 = q1 < TP ? (ly1 * STRIDE) * PW + lx1 * STRIDE : 0;
            const char* b0 = bBase + pp0 * SB;
            const char* b1 = bBase + pp1 * SB;
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                const bf16x8 fb = tr_pair(b0 + toffB[j], b1 + toffB[j]);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[j], 0, 0, 0);
            }
            q0 += 16 * KSPLIT; q1 += 16 * KSPLIT;
            lx0 += 16 * KSPLIT; while (lx0 >= p.TW) { lx0 -= p.TW; ++ly0; }
            lx1 += 16 * KSPLIT; while (lx1 >= p.TW) { lx1 -= p.TW; ++ly1; }
        }
    }

    // ---- slab store: ws[slab][k][t][c]; rows = kout (registers), column = lane
    const int slab_id = KSPLIT == 2 ? slice * 2 + grp : slice;
    float* slab = p.ws + (size_t)slab_id * p.K64 * TAPS * p.C64;
    const int c = ct * 64 + cb * 32 + r;
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int t = (KSPLIT == 1 ? grp * TPW : 0) + j;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int k = kt * 64 + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            slab[((size_t)k * TAPS + t) * p.C64 + c] = acc[j][e];
        }
    }
}

// ------------------------------------------------------------------------------------------
// 3x3 stride 1: tiles go from global memory straight into LDS (global_load_lds_dwordx4), into
// two buffers, so the loads of the next tile run under the MFMAs of the current one and cost
// no registers.  LDS image of a buffer: [dOut k-half 0][dOut k-half 1][In c-half 0][In c-half 1],
// each a dense array of 64-byte rows (32 channels of one pixel).  A transposed read takes four
// consecutive pixels of one half = 256 contiguous bytes = all 64 banks once; a filter tap is a
// constant byte offset.  One DMA wave-instruction fills 16 rows: lane l writes chunk l%4 of
// row l/4 and fetches it from that pixel (buffer load; out-of-range offsets write zeros for
// padding and pixels outside the tile).
// Waves = (k half, c half) x (taps 0-4 / taps 5-8); 5 accumulator tiles.
constexpr int G_TP = 256, G_PP = 384;                // pixels of a tile / of its halo patch (max)
constexpr int G_DIMG = G_TP * 64, G_XIMG = G_PP * 64;
constexpr int G_BUF = 2 * G_DIMG + 2 * G_XIMG;       // 81,920 bytes; two buffers = all of LDS
constexpr int G_NSTEPS = G_TP / 16;


typedef __attribute__((address_space(3))) void* lptr_t;

template <int N> struct IntC { static constexpr int value = N; };

template <bool STEAL>
__global__ __launch_bounds__(NTW, 2) void wgrad3_dma_kernel(WPParams p) {
    extern __shared__ __attribute__((aligned(16))) char smw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kb = wave & 1, cb = (wave >> 1) & 1, grp = wave >> 2;
    const int r = lane & 31, h = lane >> 5;
    const int qq4 = (lane & 15) >> 2, csel = (lane >> 4) & 1, p4 = lane & 3;

    // Slice s takes tiles s, s + nslices, ...: at any moment the workgroups read one band of the
    // images.  Workgroups of one slice (same tiles, different weight blocks) get ids that are
    // equal modulo 8, which the dispatcher places on one XCD: shared tiles are L2 hits.
    const int pairs = p.kt_tiles * p.ct_tiles;
    int pair, slice;
    if ((p.nslices & 7) == 0) {
        const int j = blockIdx.x >> 3;
        pair = j % pairs;
        slice = (blockIdx.x & 7) + 8 * (j / pairs);
    } else {
        pair = blockIdx.x % pairs;
        slice = blockIdx.x / pairs;
    }
    const int ct = pair % p.ct_tiles, kt = pair / p.ct_tiles;
    const int PW = p.TW + 2, PH = p.TH + 2;
    const int PP = PH * PW, TP = p.TH * p.TW;

    // ---- DMA jobs of this wave (tile-invariant part).  Job = 16 rows of one image half.
    // dOut halves: 16 row groups each, this wave fills half (wave & 1), groups (wave >> 1) + 4 i.
    // (issued by waves 4-7 alone, which lose the matrix pipe to waves 0-3 at the start of a tile anyway:
    // half jw & 1, row groups (jw >> 1) + 2 i; SF_WGRAD3_NO_LOADERS: all eight waves, groups (wave >> 1) + 4 i)
#ifdef SF_WGRAD3_NO_LOADERS
    constexpr int GSTEP = 4;
    const int jw = wave;
#else
    constexpr int GSTEP = 2;
    const int jw = wave - 4;
#endif
    constexpr int NDJ = G_TP / 16 / GSTEP, NXJ = G_PP / 16 / GSTEP;
    const int jhalf = jw & 1, jg0 = (jw < 0 ? 0 : jw) >> 1;
    const int jrow = lane >> 2, jcol = jhalf * 32 + (lane & 3) * 8;  // row in the group, channel
    int d_lyx[NDJ], d_rel[NDJ], x_pyx[NXJ], x_rel[NXJ];
#pragma unroll
    for (int i = 0; i < NDJ; ++i) {
        const int pix = 16 * (jg0 + GSTEP * i) + jrow;
        const int ly = pix / p.TW, lx = pix - ly * p.TW;
        d_lyx[i] = pix < TP ? ((ly << 16) | lx) : -1;
        d_rel[i] = ((ly * p.OW + lx) * p.Kd + kt * 64 + jcol) * 2;   // bytes
    }
#pragma unroll
    for (int i = 0; i < NXJ; ++i) {
        const int pp = 16 * (jg0 + GSTEP * i) + jrow;
        const int py = pp / PW, px = pp - py * PW;
        x_pyx[i] = pp < PP ? ((py << 16) | px) : -1;
        x_rel[i] = ((py * p.IW + px) * p.Cx + ct * 64 + jcol) * 2;   // bytes
    }
    const bool d_col_ok = kt * 64 + jcol < p.Kd;
    const bool x_col_ok = ct * 64 + jcol < p.Cx;
    const int lds_d0 = jhalf * G_DIMG + jg0 * 1024;               // + i * 4096 (+ buffer)
    const int lds_x0 = 2 * G_DIMG + jhalf * G_XIMG + jg0 * 1024;

    // The copies are buffer loads with an LDS destination; the buffer is the tile's image, lane
    // offsets are 32-bit bytes, and an offset past the buffer's end writes zeros (padding,
    // pixels outside the tile, channels beyond the tensor).
    auto issue_tile = [&](int tile, char* lbase) __attribute__((always_inline)) {
        if (jw < 0) return;
        const int txi = tile % p.tiles_x;
        const int rest = tile / p.tiles_x;
        const int tyi = rest % p.tiles_y;
        const int n = rest / p.tiles_y;
        const int oy0 = tyi * p.TH, ox0 = txi * p.TW;
        const int iy0 = oy0 - 1, ix0 = ox0 - 1;
        const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.dout + (size_t)n * p.OH * p.OW * p.Kd), 0, (unsigned)(p.OH * p.OW * p.Kd * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.x + (size_t)n * p.IH * p.IW * p.Cx), 0, (unsigned)(p.IH * p.IW * p.Cx * 2), 0x00020000);
        const int doff = (oy0 * p.OW + ox0) * p.Kd * 2;
        const int xoff = (iy0 * p.IW + ix0) * p.Cx * 2;
#pragma unroll
        for (int i = 0; i < NDJ; ++i) {
            const int ly = d_lyx[i] >> 16, lx = d_lyx[i] & 0xffff;
            const bool ok = d_lyx[i] >= 0 && d_col_ok && oy0 + ly < p.OH && ox0 + lx < p.OW;
            const int voff = ok ? doff + d_rel[i] : (int)0x80000000;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(drs, (lptr_t)(lbase + lds_d0 + i * GSTEP * 1024), 16, voff, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NXJ; ++i) {
            const int py = x_pyx[i] >> 16, px = x_pyx[i] & 0xffff;
            const int gy = iy0 + py, gx = ix0 + px;
            const bool ok = x_pyx[i] >= 0 && x_col_ok && gy >= 0 && gy < p.IH && gx >= 0 && gx < p.IW;
            const int voff = ok ? xoff + x_rel[i] : (int)0x80000000;
            // patch rows 368 .. 383 lie beyond the patch when tiles can be stolen, and carry the four words of the hand-over
            if (STEAL && jg0 + GSTEP * i == G_PP / 16 - 1) continue;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lptr_t)(lbase + lds_x0 + i * GSTEP * 1024), 16, voff, 0, 0, 0);
        }
    };

    // taps this wave owns (wave-uniform): 0-4 or 5-8
    constexpr int TPW = 5, TAPS = 9;
    const int tap_first = grp * TPW;
    int toffB[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int t = tap_first + j;
        toffB[j] = t < TAPS ? ((t / 3) * PW + (t % 3)) * 64 : 0;
    }
    // patch rows (as byte offsets / 16) of the two transposed reads of reduction step s:
    // tile pixel q = 16 s + 8 h + 4 half + qq4
    int ppk[G_NSTEPS];
#pragma unroll
    for (int s = 0; s < G_NSTEPS; ++s) {
        const int q0 = 16 * s + 8 * h + qq4, q1 = q0 + 4;
        const int ly0 = q0 / p.TW, lx0 = q0 - ly0 * p.TW;
        const int ly1 = q1 / p.TW, lx1 = q1 - ly1 * p.TW;
        const int pp0 = q0 < TP ? ly0 * PW + lx0 : 0;
        const int pp1 = q1 < TP ? ly1 * PW + lx1 : 0;
        ppk[s] = (pp0 * 4) | ((pp1 * 4) << 16);
    }

    f32x16 acc[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    const int lane_col = csel * 32 + p4 * 8;  // byte offset inside a 64-byte row
    const int aOff = kb * G_DIMG + (8 * h + qq4) * 64 + lane_col;
    const int bOff = 2 * G_DIMG + cb * G_XIMG + lane_col;

    // Software pipeline over the flattened (step, tap) sequence: the transposed reads of position
    // i + 2 are issued before the MFMA of position i, in exactly this order, so every MFMA waits
    // only for its own operands (a counted lgkmcnt) while later reads stay in flight.
    auto run_steps = [&](auto ntc, const char* aLane, const char* bBase) __attribute__((always_inline)) {
        constexpr int NT = decltype(ntc)::value;
        constexpr int TOT = G_NSTEPS * NT, DEPTH = 2;
        bf16x8 fa[2], fb[DEPTH + 1];
        const char* b0[2];
        const char* b1[2];
        auto issue = [&](int i) {
            const int s = i / NT, j = i % NT;
            if (j == 0) {
                int pk = ppk[s];
                asm volatile("" : "+v"(pk));  // keep the address arithmetic inside the loop
                b0[s & 1] = bBase + ((pk & 0xffff) << 4);
                b1[s & 1] = bBase + ((pk >> 16) << 4);
                fa[s & 1] = tr_pair(aLane + s * 1024, aLane + s * 1024 + 256);
            }
            fb[i % (DEPTH + 1)] = tr_pair(b0[s & 1] + toffB[j], b1[s & 1] + toffB[j]);
        };
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) issue(i);
#pragma unroll
        for (int i = 0; i < TOT; ++i) {
            if (i + DEPTH < TOT) issue(i + DEPTH);
            __builtin_amdgcn_sched_barrier(0);
            acc[i % NT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[(i / NT) & 1], fb[i % (DEPTH + 1)], acc[i % NT], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    const bool wave_idle = kt * 64 + kb * 32 >= p.Kd || ct * 64 + cb * 32 >= p.Cx;
    // One tile: start the DMA of the next tile into `fill`, run the MFMAs on `cur`.  The two
    // buffers are handed over as restrict pointers so that the compiler knows the reads of
    // `cur` cannot touch the DMA's destination (otherwise it drains the DMA before every read).
    auto tile_body = [&](const char* __restrict__ cur, char* __restrict__ fill, bool compute, int next_tile) __attribute__((always_inline)) {
        if (next_tile < p.ntiles) issue_tile(next_tile, fill);
        if (wave_idle || !compute) return;  // this wave's 32 x 32 block lies entirely in the channel padding
        if (grp == 0) run_steps(IntC<5>{}, cur + aOff, cur + bOff);
        else run_steps(IntC<4>{}, cur + aOff, cur + bOff);
    };

    if constexpr (!STEAL) {
        const int tile_step = p.nslices;
        if (slice < p.ntiles) issue_tile(slice, smw);
        for (int tile = slice; tile < p.ntiles; tile += 2 * tile_step) {
            // the tile's DMA has landed (each wave waits for its own, then the barrier) and every wave
            // is done reading the other buffer
            __syncthreads();
            tile_body(smw, smw + G_BUF, true, tile + tile_step);
            if (tile + tile_step >= p.ntiles) break;
            __syncthreads();
            tile_body(smw + G_BUF, smw, true, tile + 2 * tile_step);
        }
    } else {
        // Every workgroup owns the list of tiles its fixed slice would stream (position m of worker w:
        // tile x + 8 w + nslices m) and draws positions from the list's counter; when its own list is
        // used up, it draws from the lists of the other workers of its XCD and weight block that have
        // positions left.  Wave 0 does the drawing, four tiles ahead of the MFMAs so that nobody waits
        // for an answer: the ticket for t(i+4) is asked for while tile t(i) is computed, turned into
        // a tile index and parked in LDS after the next barrier, read into a register by all waves an
        // iteration later and used for the copy that starts an iteration after that.  The four words
        // of LDS lie in the last 16 patch rows of buffer 0, which no copy touches (the host turns
        // stealing on only when the patch leaves them free).  A position past the end of a list is a
        // HOLE (an iteration without work); DONE says that no list had anything left.
        constexpr int HOLE = 0x7ffffffe, DONE = 0x7fffffff;
        StealQueue* const wq = g_wqueues + p.qslot;
        const int x8 = blockIdx.x & 7, s8 = p.nslices >> 3, me = (blockIdx.x >> 3) / pairs;
        int* const s_next = reinterpret_cast<int*>(smw + 2 * G_DIMG + 23 * 1024);     // slot of t(j): j % 3; t(0): 3
        auto counter = [&](int w) { return &wq->next[x8 + 8 * (w * pairs + pair)]; };
        auto tile_at = [&](int w, int m) {
            const long long t = x8 + 8 * w + (long long)m * p.nslices;
            return t < p.ntiles ? (int)t : HOLE;
        };
        // wave 0 only (all of it wave-uniform, `ticket` valid in lane 0)
        bool stealing = false, have = false;
        int from = me, ticket = 0;
        auto take = [&]() {
            from = me;
            if (stealing) {     // a worker with positions left, the nearest after this one
                from = -1;
                int c = 0x7fffffff, len = 0;
                if (lane < s8) {
                    c = __hip_atomic_load(counter(lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int first = x8 + 8 * lane;
                    len = first < p.ntiles ? (p.ntiles - first + p.nslices - 1) / p.nslices : 0;
                }
                const unsigned long long left = __ballot(c < len);
                if (left) {
                    const int sh = (me + 1) & 63;
                    const unsigned long long rot = sh ? (left >> sh) | (left << (64 - sh)) : left;
                    from = (__builtin_ctzll(rot) + sh) & 63;
                }
            }
            // the ticket stays untouched until it is parked, so lane 0 does not wait for it here
            if (from >= 0 && lane == 0)
                ticket = __hip_atomic_fetch_add(counter(from), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            have = true;
        };
        auto resolve = [&]() {      // the tile of the outstanding ticket
            have = false;
            if (from < 0) return DONE;
            const int t = tile_at(from, __builtin_amdgcn_readfirstlane(ticket));
            if (t == HOLE && from == me) stealing = true;
            return t;
        };
        if (wave == 0) {
            int base = 0;
            if (p.shirk && (me & 1)) base = 0x100000;
            else if (lane == 0) base = __hip_atomic_fetch_add(counter(me), 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            base = __builtin_amdgcn_readfirstlane(base);
            const int t3 = tile_at(me, base + 3);
            stealing = t3 == HOLE;
            if (lane == 0) {
                s_next[3] = tile_at(me, base);
                s_next[1] = tile_at(me, base + 1);
                s_next[2] = tile_at(me, base + 2);
                s_next[0] = t3;
            }
        }
        __syncthreads();
        int cur_tile = __builtin_amdgcn_readfirstlane(s_next[3]);         // t(0)
        int next_tile = __builtin_amdgcn_readfirstlane(s_next[1]);        // t(1)
        if (cur_tile < p.ntiles) issue_tile(cur_tile, smw);
        auto stage = [&](int it, const char* __restrict__ cur, char* __restrict__ fill) __attribute__((always_inline)) {
            __syncthreads();
            if (next_tile < p.ntiles) issue_tile(next_tile, fill);        // first of all: the copy needs the time
            if (wave == 0 && have) {
                const int t = resolve();                                   // t(it + 3)
                if (lane == 0) s_next[it % 3] = t;
            }
            const int after_next = __builtin_amdgcn_readfirstlane(s_next[(it + 2) % 3]);   // t(it + 2)
            if (wave == 0) take();                                         // t(it + 4)
            tile_body(cur, fill, cur_tile < p.ntiles, DONE);
            cur_tile = next_tile;
            next_tile = after_next;
        };
        for (int it = 0; cur_tile != DONE; it += 2) {
            stage(it, smw, smw + G_BUF);
            if (cur_tile == DONE) break;
            stage(it + 1, smw + G_BUF, smw);
        }
        if (wave == 0) {
            if (have) asm volatile("" ::"v"(ticket));     // the last answer has arrived before the workgroup leaves
            int last = 0;
            if (lane == 0)
                last = __hip_atomic_fetch_add(&wq->done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
            if (__builtin_amdgcn_readfirstlane(last)) {     // everyone else has left: the slot is ready for its next launch
                for (int i = lane; i < (int)gridDim.x; i += 64)
                    __hip_atomic_store(&wq->next[i], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (lane == 0) __hip_atomic_store(&wq->done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }

    // ---- slab store: ws[slice][k][t][c]; rows = kout (registers), column = lane
    float* slab = p.ws + (size_t)slice * p.K64 * TAPS * p.C64;
    const int c = ct * 64 + cb * 32 + r;
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int t = tap_first + j;
        if (t >= TAPS) break;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int k = kt * 64 + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            slab[((size_t)k * TAPS + t) * p.C64 + c] = acc[j][e];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Blocks of 128 output channels x 64 input channels on the same scheme (wgrad_wide_kernel), for
// layers with more than 64 output channels: per tile pixel a workgroup fetches 256 B of dOut and
// the patch's share of 128 B of input, for twice the FLOP of a 64 x 64 block.
//   3x3 stride 1: all nine taps; waves = (k quarter, c half), nine accumulator tiles each: one dOut
//     fragment and nine input fragments feed nine MFMAs per reduction step (3.0 B per KFLOP against
//     3.9 for the 64 x 64 blocks, and the input is fetched once instead of once per 64 channels).
//   4x4 stride 2: the EIGHT taps of one filter-row parity (rows ty = rp and rp + 2, all four
//     columns).  Those taps read input rows of one parity only, so the patch of a TH x TW tile is
//     (TH + 1) rows of 2 TW + 2 pixels: 4.3 B per KFLOP (all sixteen taps on a 64 x 64 block: 5.9).
//     A patch row is stored as its even columns followed by its odd columns (the copy's lanes pick
//     their pixels, so this costs nothing): tap column tx of output pixel lx is entry lx + (tx >> 1)
//     of plane tx & 1, consecutive output pixels read consecutive rows, and a tap is again a
//     constant byte offset.  Waves = (k-quarter pair, c half) x (row ty = rp / rp + 2): two dOut
//     fragments and four input fragments feed eight MFMAs per reduction step.
// LDS image of a buffer: [dOut k-quarter 0..3][In c-half 0][In c-half 1], dense 64-byte rows as above.
// The buffers form a ring of NB: the copies of tile m + NB - 1 start when tile m is computed, and a
// wave waits for its own copies of tile m only (a counted vmcnt: the copies of the later tiles stay
// in flight).  Every wave issues the same number of copies per tile, tiles past the end as zero
// fills that fetch nothing, which keeps that count exact.
#ifndef SF_WIDE_DEPTH
#define SF_WIDE_DEPTH 2         // fragment reads in flight ahead of the MFMA that consumes them
#endif
template <bool S2_, int TP_, int PP_, int NB_> struct WideCfg {
    static constexpr bool S2 = S2_;
    static constexpr int TP = TP_, PP = PP_, NB = NB_;
    static constexpr int TAPS = S2 ? 16 : 9;
    static constexpr int NA = S2 ? 2 : 1, NT = S2 ? 4 : 9;        // dOut fragments and taps per wave
    static constexpr int DQ = TP * 64, XH = PP * 64;
    static constexpr int BUF = 4 * DQ + 2 * XH;
    static constexpr int NSTEPS = TP / 16;
    static constexpr int ND = 4 * TP / 16, NX = 2 * PP / 16;     // copy jobs (1 KB each) per tile
#ifdef SF_WIDE_NO_LOADERS
    static constexpr int JWV = 8;                                  // waves that issue the copies
#else
    static constexpr int JWV = 4;                                  // waves 4-7 (see the kernel)
#endif
    static constexpr int J = (ND + NX) / JWV;                      // per copying wave
    static_assert((ND + NX) % JWV == 0, "copy jobs must divide over the copying waves");
    static_assert(NB * BUF + 16 <= 160 * 1024, "ring exceeds LDS");
    static_assert((NB - 1) * J < 64, "vmcnt counts 63 at most");
};

// Tiles of the wide kernel when other kernels share the CUs (sf_set_wgrad_scheduling): the
// workgroups of one XCD and block draw their tiles from one list (positions x + 8 m of the launch's
// tile order) instead of streaming fixed slices, so a workgroup that starts late leaves its share to
// the others.  Thread 0 asks for the tile two phases ahead while the current one is computed and parks
// the answer in LDS before the next barrier.
__device__ WorkQueue g_widequeues[SF_QUEUE_SLOTS];
StreamSlots g_wideslots;

template <class G, bool QUEUE>
__global__ __launch_bounds__(NTW, 2) void wgrad_wide_kernel(WPParams p) {
    static_assert(!QUEUE || G::NB == 2, "the tile queue assumes two buffers");
    constexpr int NB = G::NB, J = G::J, NA = G::NA, NT = G::NT;
    constexpr bool S2 = G::S2;
    extern __shared__ __attribute__((aligned(16))) char smw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // S2: wave = (k-quarter pair, c half, filter row of the parity); S1: wave = (k quarter, c half)
    const int kq0 = S2 ? 2 * (wave & 1) : (wave & 3);
    const int cb = S2 ? (wave >> 1) & 1 : wave >> 2;
    const int grp = S2 ? wave >> 2 : 0;
    const int r = lane & 31, h = lane >> 5;
    const int qq4 = (lane & 15) >> 2, csel = (lane >> 4) & 1, p4 = lane & 3;

    // blocks = (k tile, c tile) [x row parity]; the workgroups of one slice share their tiles and
    // get ids that are equal modulo 8 (one XCD, one L2)
    const int blocks = (S2 ? 2 : 1) * p.kt_tiles * p.ct_tiles;
    int blk, slice;
    if ((p.nslices & 7) == 0) {
        const int j = blockIdx.x >> 3;
        blk = j % blocks;
        slice = (blockIdx.x & 7) + 8 * (j / blocks);
    } else {
        blk = blockIdx.x % blocks;
        slice = blockIdx.x / blocks;
    }
    const int head = (blockIdx.x & 7) + 8 * blk;      // tile list of this XCD and block
    const int rp = S2 ? blk & 1 : 0;
    if (S2) blk >>= 1;
    const int ct = blk % p.ct_tiles, kt = blk / p.ct_tiles;
    const int PWh = p.TW + 1, PW = S2 ? 2 * PWh : p.TW + 2, PH = S2 ? p.TH + 1 : p.TH + 2;
    const int PP = PH * PW, TP = p.TH * p.TW;
    constexpr int RM = S2 ? 2 : 1;          // input rows between two patch rows

    // ---- copy jobs of this wave (tile-invariant part).  The copies are issued by waves 4-7 alone: they
    // lose the matrix pipe to their older partners on the SIMD (waves 0-3) at the start of a tile
    // anyway, so the ~100 cycles a copy instruction holds its wave cost them nothing there, and waves
    // 0-3 start their MFMAs at once.  Job j = jw + JWV i: the first ND jobs fill dOut (quarter j & 3,
    // row group j >> 2), the others the input (half, row group).
    constexpr int JWV = G::JWV;
    const int jw = JWV == 4 ? wave - 4 : wave;           // index among the copying waves (negative: none)
    const int jrow = lane >> 2, jch = (lane & 3) * 8;
    int j_pyx[J], j_rel[J], j_dst[J];
    bool j_isx[J];
#pragma unroll
    for (int i = 0; i < J; ++i) {
        const int j = (jw < 0 ? 0 : jw) + JWV * i;
        j_isx[i] = j >= G::ND;
        if (!j_isx[i]) {
            const int dq = j & 3, g = j >> 2;
            const int pix = 16 * g + jrow;
            const int ly = pix / p.TW, lx = pix - ly * p.TW;
            const bool ok = pix < TP && kt * 128 + dq * 32 + jch < p.Kd;
            j_pyx[i] = ok ? ((ly << 16) | lx) : -1;
            j_rel[i] = ((ly * p.OW + lx) * p.Kd + kt * 128 + dq * 32 + jch) * 2;          // bytes
            j_dst[i] = dq * G::DQ + g * 1024;
        } else {
            const int jj = j - G::ND, xh = jj & 1, g = jj >> 1;
            const int pp = 16 * g + jrow;
            const int py = pp / PW, c = pp - py * PW;
            int px = c;                                                                   // column of the patch
            if (S2) { const int plane = c >= PWh; px = 2 * (c - plane * PWh) + plane; }
            const bool ok = pp < PP && ct * 64 + xh * 32 + jch < p.Cx;
            j_pyx[i] = ok ? ((py << 16) | px) : -1;
            j_rel[i] = ((RM * py * p.IW + px) * p.Cx + ct * 64 + xh * 32 + jch) * 2;      // bytes
            j_dst[i] = 4 * G::DQ + xh * G::XH + g * 1024;
        }
    }

    auto issue_tile = [&](int tile, char* lbase) __attribute__((always_inline)) {
        if (jw < 0) return;
        const bool live = tile < p.ntiles;
        if (!live) tile = 0;
        const int txi = tile % p.tiles_x;
        const int rest = tile / p.tiles_x;
        const int tyi = rest % p.tiles_y;
        const int n = rest / p.tiles_y;
        const int oy0 = tyi * p.TH, ox0 = txi * p.TW;
        const int iy0 = S2 ? 2 * oy0 + rp - 1 : oy0 - 1, ix0 = S2 ? 2 * ox0 - 1 : ox0 - 1;
        const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.dout + (size_t)n * p.OH * p.OW * p.Kd), 0, (unsigned)(p.OH * p.OW * p.Kd * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.x + (size_t)n * p.IH * p.IW * p.Cx), 0, (unsigned)(p.IH * p.IW * p.Cx * 2), 0x00020000);
        const int doff = (oy0 * p.OW + ox0) * p.Kd * 2;
        const int xoff = (iy0 * p.IW + ix0) * p.Cx * 2;
#pragma unroll
        for (int i = 0; i < J; ++i) {
            const int a = j_pyx[i] >> 16, b = j_pyx[i] & 0xffff;
            if (!j_isx[i]) {
                const bool ok = live && j_pyx[i] >= 0 && oy0 + a < p.OH && ox0 + b < p.OW;
                const int voff = ok ? doff + j_rel[i] : (int)0x80000000;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(drs, (lptr_t)(lbase + j_dst[i]), 16, voff, 0, 0, 0);
            } else {
                const int gy = iy0 + RM * a, gx = ix0 + b;
                const bool ok = live && j_pyx[i] >= 0 && gy >= 0 && gy < p.IH && gx >= 0 && gx < p.IW;
                const int voff = ok ? xoff + j_rel[i] : (int)0x80000000;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lptr_t)(lbase + j_dst[i]), 16, voff, 0, 0, 0);
            }
        }
    };

    // byte offsets of this wave's taps: S2 filter row rp + 2 grp, columns 0..3; S1 all nine
    int toffB[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) toffB[j] = S2 ? (grp * PW + (j & 1) * PWh + (j >> 1)) * 64 : ((j / 3) * PW + j % 3) * 64;
    int ppk[G::NSTEPS];
#pragma unroll
    for (int s = 0; s < G::NSTEPS; ++s) {
        const int q0 = 16 * s + 8 * h + qq4, q1 = q0 + 4;
        const int ly0 = q0 / p.TW, lx0 = q0 - ly0 * p.TW;
        const int ly1 = q1 / p.TW, lx1 = q1 - ly1 * p.TW;
        const int pp0 = q0 < TP ? ly0 * PW + lx0 : 0;
        const int pp1 = q1 < TP ? ly1 * PW + lx1 : 0;
        ppk[s] = (pp0 * 4) | ((pp1 * 4) << 16);
    }

    f32x16 acc[NA][NT];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][j][e] = 0.f;

    const int lane_col = csel * 32 + p4 * 8;
    const int aOff = kq0 * G::DQ + (8 * h + qq4) * 64 + lane_col;
    const int bOff = 4 * G::DQ + cb * G::XH + lane_col;

    // Positions (step, tap): the reads of position i + 2 are issued before the MFMAs of position i
    auto run_steps = [&](const char* aLane, const char* bBase) __attribute__((always_inline)) {
        constexpr int TOT = G::NSTEPS * NT, DEPTH = SF_WIDE_DEPTH;
        bf16x8 fa[2][NA], fb[DEPTH + 1];
        const char* b0[2];
        const char* b1[2];
        auto issue = [&](int i) {
            const int s = i / NT, j = i % NT;
            if (j == 0) {
                int pk = ppk[s];
                asm volatile("" : "+v"(pk));
                b0[s & 1] = bBase + ((pk & 0xffff) << 4);
                b1[s & 1] = bBase + ((pk >> 16) << 4);
#pragma unroll
                for (int a = 0; a < NA; ++a)
                    fa[s & 1][a] = tr_pair(aLane + a * G::DQ + s * 1024, aLane + a * G::DQ + s * 1024 + 256);
            }
            fb[i % (DEPTH + 1)] = tr_pair(b0[s & 1] + toffB[j], b1[s & 1] + toffB[j]);
        };
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) issue(i);
#pragma unroll
        for (int i = 0; i < TOT; ++i) {
            if (i + DEPTH < TOT) issue(i + DEPTH);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < NA; ++a)
                acc[a][i % NT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[(i / NT) & 1][a], fb[i % (DEPTH + 1)], acc[a][i % NT], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    const bool wave_idle = kt * 128 + kq0 * 32 >= p.Kd || ct * 64 + cb * 32 >= p.Cx;
    const int tile_step = p.nslices;
    // one tile: its copies have landed in `cur` (own count, then the barrier, which also says that
    // every wave is done with the buffer `fill`); start the copies NB - 1 tiles ahead, compute
    auto phase = [&](const char* __restrict__ cur, char* __restrict__ fill, int tile) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((NB - 2) * J) : "memory");
#if defined(SF_WIDE_DIAG_NODMA)       // timing only: tiles past the first ring are zero fills
        issue_tile(0x3fffffff, fill);
#elif defined(SF_WIDE_DIAG_SAMETILE)    // timing only: every copy fetches the slice's first tile (L2 hits)
        issue_tile(slice, fill);
#else
        issue_tile(tile + (NB - 1) * tile_step, fill);
#endif
#ifndef SF_WIDE_DIAG_NOMFMA     // timing only: copies and barriers without the arithmetic
        if (!wave_idle) run_steps(cur + aOff, cur + bOff);
#endif
    };
    if constexpr (!QUEUE) {
#pragma unroll
        for (int m = 0; m < NB - 1; ++m) issue_tile(slice + m * tile_step, smw + m * G::BUF);
        for (int tile = slice; tile < p.ntiles;) {
#pragma unroll
            for (int ph = 0; ph < NB; ++ph) {
                phase(smw + ph * G::BUF, smw + ((ph + NB - 1) % NB) * G::BUF, tile);
                tile += tile_step;
                if (tile >= p.ntiles) break;
            }
        }
    } else {
        constexpr int DONE = 0x7fffffff;
        WorkQueue* const wq = g_widequeues + p.qslot;
        int* const s_next = reinterpret_cast<int*>(smw + NB * G::BUF);         // tile of phase k: slot k & 1
        auto tile_at = [&](int m) {
            const long long t = (blockIdx.x & 7) + 8LL * m;
            return t < p.ntiles ? (int)t : DONE;
        };
        if (tid == 0) {
            const int m0 = queue_take(wq, head), m1 = queue_take(wq, head);
            s_next[0] = tile_at(m0);
            s_next[1] = tile_at(m1);
        }
        __syncthreads();
        int cur_tile = __builtin_amdgcn_readfirstlane(s_next[0]);
        issue_tile(cur_tile, smw);
        int pending = 0;
        auto qphase = [&](int k, const char* __restrict__ cur, char* __restrict__ fill) __attribute__((always_inline)) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const int next_tile = __builtin_amdgcn_readfirstlane(s_next[(k + 1) & 1]);
            if (tid == 0) pending = queue_take_async(wq, head);             // the tile of phase k + 2
            issue_tile(next_tile, fill);
            if (!wave_idle) run_steps(cur + aOff, cur + bOff);
            if (tid == 0) s_next[k & 1] = tile_at(queue_ticket_wait(pending));
            cur_tile = next_tile;
        };
        while (cur_tile != DONE) {
            qphase(0, smw, smw + G::BUF);
            if (cur_tile == DONE) break;
            qphase(1, smw + G::BUF, smw);
        }
        if (tid == 0) queue_leave(wq, 8 * blocks, gridDim.x);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the zero fills of the tail

    // ---- slab store: ws[slice][k][t][c]
    float* slab = p.ws + (size_t)slice * p.K64 * G::TAPS * p.C64;
    const int c = ct * 64 + cb * 32 + r;
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int t = S2 ? (rp + 2 * grp) * 4 + j : j;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int k = kt * 128 + (kq0 + a) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                slab[((size_t)k * G::TAPS + t) * p.C64 + c] = acc[a][j][e];
            }
        }
}

// dw[k][c][t] (OIHW f32) = sum_slabs ws[slab][k][t][c]
__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nslabs, int K64,
                                    int C64, int K, int Cin, int taps) {
    const long long total = (long long)K * taps * Cin;
    const size_t stride = (size_t)K64 * taps * C64;
    for (long long o = blockIdx.x * 256LL + threadIdx.x; o < total; o += (long long)gridDim.x * 256) {
        const int c = (int)(o % Cin);
        const long long kt_ = o / Cin;
        const int t = (int)(kt_ % taps);
        const int k = (int)(kt_ / taps);
        const float* src = ws + ((size_t)k * taps + t) * C64 + c;
        // slabs are added in index order (fixed rounding), eight loads in flight at a time
        float a = 0.f;
        int s = 0;
        for (; s + 8 <= nslabs; s += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(s + u) * stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) a += v[u];
        }
        for (; s < nslabs; ++s) a += src[(size_t)s * stride];
        dw[((size_t)k * Cin + c) * taps + t] = a;
    }
}

template <int TAPS, int STRIDE> void pick_tile_w(int OH, int OW, int* th, int* tw) {
    using Cf = WCfg<TAPS, STRIDE>;
    long long best = -1;
    int bth = 1, btw = 16, bpp = 0;
    for (int w = 8; w <= Cf::TPMAX; ++w) {
        const int hh = Cf::TPMAX / w;
        if (hh < 1) continue;
        const int pp = ((hh - 1) * STRIDE + Cf::R) * ((w - 1) * STRIDE + Cf::R);
        if (pp > Cf::PPMAX) continue;
        const long long tiles = (long long)((OW + w - 1) / w) * ((OH + hh - 1) / hh);
        if (best < 0 || tiles < best || (tiles == best && pp < bpp)) { best = tiles; bth = hh; btw = w; bpp = pp; }
    }
    *th = bth; *tw = btw;
}

// Workgroups per weight block: one per CU share (256 / blocks)
int slices_for(int K, int Cin) {
    const int pairs = ((K + 63) / 64) * ((Cin + 63) / 64);
    int n = 256 / pairs;
    return n < 1 ? 1 : n;
}


template <int TAPS, int STRIDE>
int run_wgrad(const void* x, const void* dout, float* dw, float* ws, size_t ws_bytes, size_t need, int N, int IH,
              int IW, int OH, int OW, int Cx, int Kd, int K, int Cin, hipStream_t s, const char* who) {
    using Cf = WCfg<TAPS, STRIDE>;
    if (ws_bytes < need) { sf_set_error("%s: workspace too small (%zu < %zu)", who, ws_bytes, need); return SF_EINVAL; }
    WPParams p;
    p.dout = (const bf16*)dout; p.x = (const bf16*)x; p.ws = ws;
    p.N = N; p.IH = IH; p.IW = IW; p.OH = OH; p.OW = OW; p.Kd = Kd; p.Cx = Cx;
    pick_tile_w<TAPS, STRIDE>(OH, OW, &p.TH, &p.TW);
    p.tiles_x = (OW + p.TW - 1) / p.TW;
    p.tiles_y = (OH + p.TH - 1) / p.TH;
    const long long ntiles = (long long)N * p.tiles_x * p.tiles_y;
    if (ntiles >= 0x7fffffffLL) { sf_set_error("%s: too many tiles", who); return SF_EINVAL; }
    p.ntiles = (int)ntiles;
    p.kt_tiles = (K + 63) / 64;
    p.ct_tiles = (Cin + 63) / 64;
    p.K64 = p.kt_tiles * 64;
    p.C64 = p.ct_tiles * 64;
    const int nslices = slices_for(K, Cin);
    p.nslices = nslices;
    p.tiles_per_slice = (p.ntiles + nslices - 1) / nslices;
    const size_t lds = (size_t)Cf::TPMAX * SA + (size_t)Cf::PPMAX * Cf::SB;
    static DeviceOnce attr_set;
    if (!attr_set.done()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_patch_kernel<TAPS, STRIDE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            sf_set_error("%s: cannot raise the dynamic LDS limit to %zu bytes", who, lds);
            return SF_ELAUNCH;
        }
        attr_set.mark();
    }
    hipLaunchKernelGGL((wgrad_patch_kernel<TAPS, STRIDE>), dim3((unsigned)(nslices * p.kt_tiles * p.ct_tiles)),
                       dim3(NTW), lds, s, p);
    const long long total = (long long)K * TAPS * Cin;
    int grid = (int)((total + 255) / 256); if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid), dim3(256), 0, s, ws, dw, nslices * Cf::KSPLIT, p.K64, p.C64, K,
                       Cin, TAPS);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { sf_set_error("%s: launch failed: %s", who, hipGetErrorString(e)); return SF_ELAUNCH; }
    return SF_OK;
}

// tile of the DMA kernel: at most G_TP pixels, patch at most G_PP; fewest tiles, then least halo
void pick_tile_dma(int OH, int OW, int* th, int* tw) {
    long long best = -1;
    int bth = 1, btw = 8, bpp = 0;
    for (int w = 4; w <= G_TP; ++w) {
        int hh = G_TP / w;
        if (hh > OH) hh = OH;
        if (hh < 1) continue;
        const int pp = (hh + 2) * (w + 2);
        if (pp > G_PP) continue;
        const long long tiles = (long long)((OW + w - 1) / w) * ((OH + hh - 1) / hh);
        if (best < 0 || tiles < best || (tiles == best && pp < bpp)) { best = tiles; bth = hh; btw = w; bpp = pp; }
    }
    *th = bth; *tw = btw;
}

int run_wgrad3_dma(const void* x, const void* dout, float* dw, float* ws, size_t ws_bytes, size_t need, int N, int H,
                   int W, int Cx, int Kd, int K, int Cin, hipStream_t s) {
    const char* who = "sf_wgrad3x3";
    if (ws_bytes < need) { sf_set_error("%s: workspace too small (%zu < %zu)", who, ws_bytes, need); return SF_EINVAL; }
    WPParams p;
    p.dout = (const bf16*)dout; p.x = (const bf16*)x; p.ws = ws;
    p.N = N; p.IH = H; p.IW = W; p.OH = H; p.OW = W; p.Kd = Kd; p.Cx = Cx;
    pick_tile_dma(H, W, &p.TH, &p.TW);
    p.tiles_x = (W + p.TW - 1) / p.TW;
    p.tiles_y = (H + p.TH - 1) / p.TH;
    const long long ntiles = (long long)N * p.tiles_x * p.tiles_y;
    if (ntiles >= 0x7fffffffLL) { sf_set_error("%s: too many tiles", who); return SF_EINVAL; }
    // 32-bit element offsets inside a tile
    if (2LL * H * W * Cx >= 0x7fffffffLL || 2LL * H * W * Kd >= 0x7fffffffLL) {
        sf_set_error("%s: one image exceeds 2 GiB", who); return SF_EINVAL;
    }
    p.ntiles = (int)ntiles;
    p.kt_tiles = (K + 63) / 64;
    p.ct_tiles = (Cin + 63) / 64;
    p.K64 = p.kt_tiles * 64;
    p.C64 = p.ct_tiles * 64;
    int nslices = slices_for(K, Cin);
    if (nslices > p.ntiles) nslices = p.ntiles;
    p.nslices = nslices;
    p.tiles_per_slice = 0;
    const size_t lds = 2 * (size_t)G_BUF;
    static DeviceOnce attr_set;
    if (!attr_set.done()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3_dma_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3_dma_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            sf_set_error("%s: cannot raise the dynamic LDS limit to %zu bytes", who, lds);
            return SF_ELAUNCH;
        }
        attr_set.mark();
    }
    const int pairs = p.kt_tiles * p.ct_tiles;
    const bool queue_ok = g_dynamic_tiles.load() && (nslices & 7) == 0 && nslices <= 512 &&
                          nslices * pairs <= STEAL_WORKERS && (p.TH + 2) * (p.TW + 2) <= 368 && p.ntiles > nslices;
    p.qslot = queue_ok ? g_wslots.slot_of(s) : -1;
    p.shirk = g_dynamic_tiles.load() == 2;
    if (p.qslot >= 0) hipLaunchKernelGGL(wgrad3_dma_kernel<true>, dim3((unsigned)(nslices * pairs)), dim3(NTW), lds, s, p);
    else hipLaunchKernelGGL(wgrad3_dma_kernel<false>, dim3((unsigned)(nslices * pairs)), dim3(NTW), lds, s, p);
    const long long total = (long long)K * 9 * Cin;
    int grid = (int)((total + 255) / 256); if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid), dim3(256), 0, s, ws, dw, nslices, p.K64, p.C64, K, Cin, 9);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { sf_set_error("%s: launch failed: %s", who, hipGetErrorString(e)); return SF_ELAUNCH; }
    return SF_OK;
}

// tile of the wide-block kernel: at most G::TP pixels, patch (th + 2) x (tw + 2) [stride 2: (th + 1) x
// (2 tw + 2)] at most G::PP; fewest tiles, then least halo
template <class G> void pick_tile_wide(int OH, int OW, int* th, int* tw) {
    long long best = -1;
    int bth = 1, btw = 4, bpp = 0;
    auto patch = [](int hh, int w) { return G::S2 ? (hh + 1) * (2 * w + 2) : (hh + 2) * (w + 2); };
    for (int w = 4; w <= G::TP; ++w) {
        int hh = G::TP / w;
        if (hh > OH) hh = OH;
        while (hh >= 1 && patch(hh, w) > G::PP) --hh;
        if (hh < 1) continue;
        const int pp = patch(hh, w);
        const long long tiles = (long long)((OW + w - 1) / w) * ((OH + hh - 1) / hh);
        if (best < 0 || tiles < best || (tiles == best && pp < bpp)) { best = tiles; bth = hh; btw = w; bpp = pp; }
    }
    *th = bth; *tw = btw;
}

int slices_for_wide(int K, int Cin, bool s2) {
    const int blocks = (s2 ? 2 : 1) * ((K + 127) / 128) * ((Cin + 63) / 64);
    int n = 256 / blocks;
    return n < 1 ? 1 : n;
}

template <class G>
int run_wgrad_wide(const char* who, const void* x, const void* dout, float* dw, float* ws, size_t ws_bytes, size_t need,
                   int N, int IH, int IW, int OH, int OW, int Cx, int Kd, int K, int Cin, hipStream_t s) {
    if (ws_bytes < need) { sf_set_error("%s: workspace too small (%zu < %zu)", who, ws_bytes, need); return SF_EINVAL; }
    WPParams p;
    p.dout = (const bf16*)dout; p.x = (const bf16*)x; p.ws = ws;
    p.N = N; p.IH = IH; p.IW = IW; p.OH = OH; p.OW = OW; p.Kd = Kd; p.Cx = Cx;
    pick_tile_wide<G>(OH, OW, &p.TH, &p.TW);
    p.tiles_x = (OW + p.TW - 1) / p.TW;
    p.tiles_y = (OH + p.TH - 1) / p.TH;
    const long long ntiles = (long long)N * p.tiles_x * p.tiles_y;
    if (ntiles >= 0x7fffffffLL / 2) { sf_set_error("%s: too many tiles", who); return SF_EINVAL; }
    // 32-bit byte offsets inside one image (a tile's patch may reach a few rows past the image)
    if (2LL * (IH + 4) * IW * Cx >= 0x7fffffffLL || 2LL * (OH + 1) * OW * Kd >= 0x7fffffffLL) {
        sf_set_error("%s: one image exceeds 2 GiB", who); return SF_EINVAL;
    }
    p.ntiles = (int)ntiles;
    p.kt_tiles = (K + 127) / 128;
    p.ct_tiles = (Cin + 63) / 64;
    p.K64 = p.kt_tiles * 128;
    p.C64 = p.ct_tiles * 64;
    int nslices = slices_for_wide(K, Cin, G::S2);
    p.nslices = nslices;
    p.tiles_per_slice = 0;
    const int blocks = (G::S2 ? 2 : 1) * p.kt_tiles * p.ct_tiles;
    const bool queue_ok = G::NB == 2 && g_dynamic_tiles.load() && (nslices & 7) == 0 && 8 * blocks <= SF_QUEUE_HEADS &&
                          p.ntiles > nslices;
    p.qslot = queue_ok ? g_wideslots.slot_of(s) : -1;
    p.shirk = 0;
    const size_t lds = (size_t)G::NB * G::BUF + 16;
    static DeviceOnce attr_set;
    if (!attr_set.done()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_wide_kernel<G, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_wide_kernel<G, G::NB == 2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            sf_set_error("%s: cannot raise the dynamic LDS limit to %zu bytes", who, lds);
            return SF_ELAUNCH;
        }
        attr_set.mark();
    }
    if (p.qslot >= 0) hipLaunchKernelGGL((wgrad_wide_kernel<G, G::NB == 2>), dim3((unsigned)(nslices * blocks)), dim3(NTW), lds, s, p);
    else hipLaunchKernelGGL((wgrad_wide_kernel<G, false>), dim3((unsigned)(nslices * blocks)), dim3(NTW), lds, s, p);
    const long long total = (long long)K * G::TAPS * Cin;
    int grid = (int)((total + 255) / 256); if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid), dim3(256), 0, s, ws, dw, nslices, p.K64, p.C64, K, Cin, G::TAPS);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { sf_set_error("%s: launch failed: %s", who, hipGetErrorString(e)); return SF_ELAUNCH; }
    return SF_OK;
}

// Measured (batch 32, launch + slab fold, probe data).  PatchGAN layers: 264 / 231 / 209 us on the
// register-staged 64 x 64 kernel, 102 / 101 / 113 us here with two buffers of 128-pixel tiles; rings of
// three 80-pixel or four 64-pixel buffers 99 / 105 / 121 and 107 / 111 / 127.  Encoder 3x3 layer (160 maps of
// 200 x 200, 64 -> 128): 0.97 ms on 64 x 64 blocks, 0.85 ms here with two buffers of 160-pixel tiles, 0.87 with
// 128-pixel tiles, three 96-pixel or four 64-pixel buffers.  So a longer landing window buys nothing and the
// smaller tiles fit the maps worse; the ring stays at two.  With the copies taken out (SF_WIDE_DIAG_NODMA,
// zero operands) the encoder launch takes 0.50 ms, with the MFMAs taken out (SF_WIDE_DIAG_NOMFMA) 0.43 ms
// (6.5 TB/s), with every copy an L2 hit (SF_WIDE_DIAG_SAMETILE) 0.70 ms.

// 3x3 layers with more than 64 output channels: SF_WGRAD3_WIDE=0 keeps the 64 x 64 blocks for comparisons
bool wgrad3_wide() {
    static const bool v = [] { const char* e = getenv("SF_WGRAD3_WIDE"); return !(e && e[0] == '0'); }();
    return v;
}

// SF_WGRAD4_OLD=1 keeps the register-staged kernel (sixteen taps on 64 x 64 blocks) for comparisons
bool wgrad4_old() {
    static const bool v = [] { const char* e = getenv("SF_WGRAD4_OLD"); return e && e[0] == '1'; }();
    return v;
}

}  // namespace

extern "C" int sf_set_wgrad_scheduling(int dynamic) {
    g_dynamic_tiles.store(dynamic < 0 ? 0 : dynamic > 2 ? 2 : dynamic);
    return SF_OK;
}

// wide blocks pay when the layer has a second 64-channel block of outputs
static bool wgrad3_use_wide(int K) { return K > 64 && wgrad3_wide(); }

extern "C" size_t sf_wgrad3x3_workspace_bytes(int N, int H, int W, int K, int Cin) {
    (void)N; (void)H; (void)W;
    const size_t wide = sizeof(float) * (size_t)slices_for_wide(K, Cin, false) * ((K + 127) / 128 * 128) * 9 * ((Cin + 63) / 64 * 64);
    const size_t narrow = sizeof(float) * (size_t)slices_for(K, Cin) * ((K + 63) / 64 * 64) * 9 * ((Cin + 63) / 64 * 64);
    if (K > 64) return wide > narrow ? wide : narrow;
    return sizeof(float) * (size_t)slices_for(K, Cin) * ((K + 63) / 64 * 64) * 9 * ((Cin + 63) / 64 * 64);
}

extern "C" size_t sf_wgrad4x4s2_workspace_bytes(int N, int IH, int IW, int K, int Cin) {
    (void)N; (void)IH; (void)IW;
    if (!wgrad4_old())
        return sizeof(float) * (size_t)slices_for_wide(K, Cin, true) * ((K + 127) / 128 * 128) * 16 * ((Cin + 63) / 64 * 64);
    return sizeof(float) * (size_t)slices_for(K, Cin) * ((K + 63) / 64 * 64) * 16 * ((Cin + 63) / 64 * 64);
}

static int wgrad_common_checks(const char* who, const void* x, const void* dout, float* dw, float* ws, int N, int H,
                               int W, int Cx, int Kd, int K, int Cin) {
    SF_REQUIRE(x && dout && dw && ws, "%s: null argument", who);
    SF_REQUIRE(N > 0 && H > 0 && W > 0 && K > 0 && Cin > 0, "%s: bad shape", who);
    SF_REQUIRE(Cx % 8 == 0 && Kd % 8 == 0 && Cin <= Cx && K <= Kd, "%s: channel strides must be multiples of 8 and cover the real counts", who);
    SF_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)dout % 16) == 0, "%s: operands must be 16-byte aligned", who);
    return SF_OK;
}

extern "C" int sf_wgrad3x3(const void* x, const void* dout, float* dw, float* ws, size_t ws_bytes, int N, int H, int W,
                           int Cx, int Kd, int K, int Cin, void* stream) {
    if (int rc = wgrad_common_checks("sf_wgrad3x3", x, dout, dw, ws, N, H, W, Cx, Kd, K, Cin)) return rc;
    const size_t need = sf_wgrad3x3_workspace_bytes(N, H, W, K, Cin);
    if (wgrad3_use_wide(K))
        return run_wgrad_wide<WideCfg<false, 160, 256, 2>>("sf_wgrad3x3", x, dout, dw, ws, ws_bytes, need, N, H, W, H, W, Cx, Kd,
                                                            K, Cin, (hipStream_t)stream);
    return run_wgrad3_dma(x, dout, dw, ws, ws_bytes, need, N, H, W, Cx, Kd, K, Cin, (hipStream_t)stream);
}

extern "C" int sf_wgrad4x4s2(const void* x, const void* dout, float* dw, float* ws, size_t ws_bytes, int N, int IH,
                             int IW, int Cx, int Kd, int K, int Cin, void* stream) {
    if (int rc = wgrad_common_checks("sf_wgrad4x4s2", x, dout, dw, ws, N, IH, IW, Cx, Kd, K, Cin)) return rc;
    SF_REQUIRE(IH >= 2 && IW >= 2, "sf_wgrad4x4s2: input smaller than the filter footprint");
    const int OH = (IH - 2) / 2 + 1, OW = (IW - 2) / 2 + 1;
    if (!wgrad4_old())
        return run_wgrad_wide<WideCfg<true, 128, 320, 2>>("sf_wgrad4x4s2", x, dout, dw, ws, ws_bytes,
                                                           sf_wgrad4x4s2_workspace_bytes(N, IH, IW, K, Cin), N, IH, IW, OH, OW, Cx,
                                                           Kd, K, Cin, (hipStream_t)stream);
    return run_wgrad<16, 2>(x, dout, dw, ws, ws_bytes, sf_wgrad4x4s2_workspace_bytes(N, IH, IW, K, Cin), N, IH, IW, OH,
                            OW, Cx, Kd, K, Cin, (hipStream_t)stream, "sf_wgrad4x4s2");
}
