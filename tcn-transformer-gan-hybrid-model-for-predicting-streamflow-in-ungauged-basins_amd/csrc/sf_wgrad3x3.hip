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

namespace {

constexpr int NTW = 512;
constexpr int SA = 192;  // bytes per dOut row in LDS

template <int TAPS, int STRIDE> struct WCfg;
template <> struct WCfg<9, 1> {
    static constexpr int R = 3, TPMAX = 256, PPMAX = 400, SB = 192, TPW = 9, KSPLIT = 2;
};
template <> struct WCfg<16, 2> {
    static constexpr int R = 4, TPMAX = 128, PPMAX = 864, SB = 160, TPW = 8, KSPLIT = 1;
};

struct WPParams {
    const bf16* dout;
    const bf16* x;
    float* ws;
    int N, IH, IW, OH, OW, Kd, Cx;
    int TH, TW, tiles_x, tiles_y, ntiles, tiles_per_slice;
    int kt_tiles, ct_tiles, K64, C64;
};

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
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (d_lyx[i] >= 0 && d_col_ok) {
                const int ly = d_lyx[i] >> 16, lx = d_lyx[i] & 0xffff;
                if (oy0 + ly < p.OH && ox0 + lx < p.OW)
                    v = *reinterpret_cast<const uint4*>(p.dout + (based + d_rel[i]) * p.Kd + kt * 64 + cc * 8);
            }
            rd[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (x_pyx[i] >= 0 && x_col_ok) {
                const int gy = iy0 + (x_pyx[i] >> 16), gx = ix0 + (x_pyx[i] & 0xffff);
                if (gy >= 0 && gy < p.IH && gx >= 0 && gx < p.IW)
                    v = *reinterpret_cast<const uint4*>(p.x + (basex + x_rel[i]) * p.Cx + ct * 64 + cc * 8);
            }
            rx[i] = v;
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
        float a = 0.f;
        for (int s = 0; s < nslabs; ++s) a += src[s * stride];
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
    p.tiles_per_slice = (p.ntiles + nslices - 1) / nslices;
    const size_t lds = (size_t)Cf::TPMAX * SA + (size_t)Cf::PPMAX * Cf::SB;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_patch_kernel<TAPS, STRIDE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            sf_set_error("%s: cannot raise the dynamic LDS limit to %zu bytes", who, lds);
            return SF_ELAUNCH;
        }
        attr_set = true;
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

}  // namespace

extern "C" size_t sf_wgrad3x3_workspace_bytes(int N, int H, int W, int K, int Cin) {
    (void)N; (void)H; (void)W;
    return sizeof(float) * (size_t)slices_for(K, Cin) * 2 * ((K + 63) / 64 * 64) * 9 * ((Cin + 63) / 64 * 64);
}

extern "C" size_t sf_wgrad4x4s2_workspace_bytes(int N, int IH, int IW, int K, int Cin) {
    (void)N; (void)IH; (void)IW;
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
    return run_wgrad<9, 1>(x, dout, dw, ws, ws_bytes, sf_wgrad3x3_workspace_bytes(N, H, W, K, Cin), N, H, W, H, W, Cx,
                           Kd, K, Cin, (hipStream_t)stream, "sf_wgrad3x3");
}

extern "C" int sf_wgrad4x4s2(const void* x, const void* dout, float* dw, float* ws, size_t ws_bytes, int N, int IH,
                             int IW, int Cx, int Kd, int K, int Cin, void* stream) {
    if (int rc = wgrad_common_checks("sf_wgrad4x4s2", x, dout, dw, ws, N, IH, IW, Cx, Kd, K, Cin)) return rc;
    SF_REQUIRE(IH >= 2 && IW >= 2, "sf_wgrad4x4s2: input smaller than the filter footprint");
    const int OH = (IH - 2) / 2 + 1, OW = (IW - 2) / 2 + 1;
    return run_wgrad<16, 2>(x, dout, dw, ws, ws_bytes, sf_wgrad4x4s2_workspace_bytes(N, IH, IW, K, Cin), N, IH, IW, OH,
                            OW, Cx, Kd, K, Cin, (hipStream_t)stream, "sf_wgrad4x4s2");
}
