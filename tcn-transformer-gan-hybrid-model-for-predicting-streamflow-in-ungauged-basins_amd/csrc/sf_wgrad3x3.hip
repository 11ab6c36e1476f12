// Weight gradient of a 3x3 stride-1 convolution, bf16 NHWC.
//
//   dW[k][t][c] = sum_pixels dOut[pixel][k] * In[pixel + tap_t][c]
//
// A workgroup owns a 64 (output channels) x 64 (input channels) block of dW for ALL nine taps
// and streams a slice of the spatial tiles: per tile it stages the dOut tile and the halo patch
// of the input in LDS once; each wave keeps nine 32x32 accumulator tiles (one per tap) in
// registers across the whole slice, so nothing is accumulated in memory until the end.  Both
// MFMA operands are reduction-major in LDS ([pixel][channel]) and are read transposed with
// ds_read_b64_tr_b16; rows are 128 bytes with the two 64-byte halves swapped on pixels whose
// index has bit 1 set, which makes every transposed read bank-conflict-free.  The next tile's
// global loads are in flight (in registers) while the current tile's 144 MFMAs per wave run.
// Per-slice partial results go to a slab; a second kernel sums the slabs in a fixed order
// (bitwise reproducible, no atomics).
#include "sf_common.h"
#include "../../include/sfhip.h"

namespace {

constexpr int NTW = 256;
constexpr int MAX_PPW = 400;
constexpr int ND = 8;    // dOut staging slots per thread: 256 pixels x 8 chunks / 256 threads
constexpr int NX = 13;   // patch staging slots per thread: ceil(400 x 8 / 256)

struct W3Params {
    const bf16* dout;
    const bf16* x;
    float* ws;
    int N, H, W, Kd, Cx;
    int TH, TW, tiles_x, tiles_y, ntiles, tiles_per_slice;
    int kt_tiles, ct_tiles, K64, C64;
    int dy[9], dx[9];
};

typedef bf16x4 __attribute__((address_space(3))) * lds_tr_p;

__device__ __forceinline__ bf16x8 tr_pair(const char* base, int off0, int off1) {
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_tr_p)(base + off0));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_tr_p)(base + off1));
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

// byte offset of 16-byte chunk cc of row `row` (128-byte rows, halves swapped when row bit 1 is set)
__device__ __forceinline__ int row_chunk_off(int row, int cc) {
    return row * 128 + ((((cc >> 2) ^ (row >> 1)) & 1) << 6) + (cc & 3) * 16;
}

__global__ __launch_bounds__(NTW, 1) void wgrad3x3_kernel(W3Params p) {
    extern __shared__ __attribute__((aligned(16))) char smw[];
    char* sD = smw;                 // [256][128 B]
    char* sX = smw + 256 * 128;     // [MAX_PPW][128 B]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kb = wave & 1, cb = wave >> 1;
    const int r = lane & 31, h = lane >> 5;
    const int qq4 = (lane & 15) >> 2, csel = (lane >> 4) & 1, p4 = lane & 3;

    int b = blockIdx.x;
    const int ct = b % p.ct_tiles; b /= p.ct_tiles;
    const int kt = b % p.kt_tiles;
    const int slice = b / p.kt_tiles;
    const int PW = p.TW + 2, PP = (p.TH + 2) * PW, TP = p.TH * p.TW;

    // ---- tile-invariant staging slots
    int d_rel[ND], d_lyx[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        const int pix = (tid + i * NTW) >> 3;
        const int ly = pix / p.TW, lx = pix - ly * p.TW;
        d_rel[i] = ly * p.W + lx;
        d_lyx[i] = pix < TP ? ((ly << 16) | lx) : -1;
    }
    int x_rel[NX], x_pyx[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int pp = (tid + i * NTW) >> 3;
        const int py = pp / PW, px = pp - py * PW;
        x_rel[i] = py * p.W + px;
        x_pyx[i] = pp < PP ? ((py << 16) | px) : -1;
    }
    const int cc = tid & 7;  // (tid + i*256) & 7
    const bool d_col_ok = kt * 64 + cc * 8 < p.Kd;
    const bool x_col_ok = ct * 64 + cc * 8 < p.Cx;

    uint4 rd[ND], rx[NX];
    auto load_tile = [&](int tile) {
        const int txi = tile % p.tiles_x;
        const int rest = tile / p.tiles_x;
        const int tyi = rest % p.tiles_y;
        const int n = rest / p.tiles_y;
        const int y0 = tyi * p.TH, x0 = txi * p.TW;
        const long long based = ((long long)n * p.H + y0) * p.W + x0;
        const long long basex = ((long long)n * p.H + (y0 - 1)) * p.W + (x0 - 1);
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (d_lyx[i] >= 0 && d_col_ok) {
                const int ly = d_lyx[i] >> 16, lx = d_lyx[i] & 0xffff;
                if (y0 + ly < p.H && x0 + lx < p.W)
                    v = *reinterpret_cast<const uint4*>(p.dout + (based + d_rel[i]) * p.Kd + kt * 64 + cc * 8);
            }
            rd[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (x_pyx[i] >= 0 && x_col_ok) {
                const int gy = y0 - 1 + (x_pyx[i] >> 16), gx = x0 - 1 + (x_pyx[i] & 0xffff);
                if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)
                    v = *reinterpret_cast<const uint4*>(p.x + (basex + x_rel[i]) * p.Cx + ct * 64 + cc * 8);
            }
            rx[i] = v;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int pix = (tid + i * NTW) >> 3;
            *reinterpret_cast<uint4*>(sD + row_chunk_off(pix, cc)) = rd[i];
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int pp = (tid + i * NTW) >> 3;
            if (pp < MAX_PPW) *reinterpret_cast<uint4*>(sX + row_chunk_off(pp, cc)) = rx[i];
        }
    };

    int toff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) toff[t] = p.dy[t] * PW + p.dx[t];

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    const int tile_begin = slice * p.tiles_per_slice;
    int tile_end = tile_begin + p.tiles_per_slice;
    if (tile_end > p.ntiles) tile_end = p.ntiles;
    const int lane_col = csel * 32 + p4 * 8;  // byte offset inside a 64-byte half

    if (tile_begin < tile_end) load_tile(tile_begin);
    for (int tile = tile_begin; tile < tile_end; ++tile) {
        if (tile > tile_begin) __syncthreads();
        store_tile();
        __syncthreads();
        if (tile + 1 < tile_end) load_tile(tile + 1);
        // rows this lane addresses at k-step s: q = 16 s + 8 h + 4 half + qq4
        int q0 = 8 * h + qq4, q1 = q0 + 4;
        int ly0 = q0 / p.TW, lx0 = q0 - ly0 * p.TW;
        int ly1 = q1 / p.TW, lx1 = q1 - ly1 * p.TW;
        for (int s = 0; s < 16; ++s) {
            const int a0 = q0 * 128 + (((kb ^ (q0 >> 1)) & 1) << 6) + lane_col;
            const int a1 = q1 * 128 + (((kb ^ (q1 >> 1)) & 1) << 6) + lane_col;
            const bf16x8 fa = tr_pair(sD, a0, a1);
            const int pp0 = q0 < TP ? (ly0 + 1) * PW + lx0 + 1 : PW + 1;
            const int pp1 = q1 < TP ? (ly1 + 1) * PW + lx1 + 1 : PW + 1;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int r0 = pp0 + toff[t], r1 = pp1 + toff[t];
                const int b0 = r0 * 128 + (((cb ^ (r0 >> 1)) & 1) << 6) + lane_col;
                const int b1 = r1 * 128 + (((cb ^ (r1 >> 1)) & 1) << 6) + lane_col;
                const bf16x8 fb = tr_pair(sX, b0, b1);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[t], 0, 0, 0);
            }
            q0 += 16; q1 += 16;
            lx0 += 16; while (lx0 >= p.TW) { lx0 -= p.TW; ++ly0; }
            lx1 += 16; while (lx1 >= p.TW) { lx1 -= p.TW; ++ly1; }
        }
    }

    // ---- slab store: ws[slice][k][t][c], rows = kout (registers), column = lane
    float* slab = p.ws + (size_t)slice * p.K64 * 9 * p.C64;
    const int c = ct * 64 + cb * 32 + r;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int k = kt * 64 + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            slab[((size_t)k * 9 + t) * p.C64 + c] = acc[t][e];
        }
}

// dw[k][c][rs] (OIHW f32) = sum_slices ws[slice][k][t][c]
__global__ void wgrad3x3_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nslices, int K64,
                                       int C64, int K, int Cin, int accumulate) {
    const long long total = (long long)K * 9 * Cin;
    for (long long o = blockIdx.x * (long long)NTW + threadIdx.x; o < total; o += (long long)gridDim.x * NTW) {
        const int c = (int)(o % Cin);
        const long long kt_ = o / Cin;
        const int t = (int)(kt_ % 9);
        const int k = (int)(kt_ / 9);
        float a = 0.f;
        const size_t stride = (size_t)K64 * 9 * C64;
        const float* src = ws + ((size_t)k * 9 + t) * C64 + c;
        for (int s = 0; s < nslices; ++s) a += src[s * stride];
        float* dst = dw + ((size_t)k * Cin + c) * 9 + t;
        *dst = accumulate ? *dst + a : a;
    }
}

void pick_tile_w(int H, int W, int* th, int* tw) {
    long long best = -1;
    int bth = 8, btw = 32, bpp = 0;
    for (int w = 16; w <= 128; ++w) {
        const int hh = 256 / w;
        if (hh < 1) continue;
        const int pp = (hh + 2) * (w + 2);
        if (pp > MAX_PPW) continue;
        const long long tiles = (long long)((W + w - 1) / w) * ((H + hh - 1) / hh);
        if (best < 0 || tiles < best || (tiles == best && pp < bpp)) { best = tiles; bth = hh; btw = w; bpp = pp; }
    }
    *th = bth; *tw = btw;
}

}  // namespace

extern "C" size_t sf_wgrad3x3_workspace_bytes(int N, int H, int W, int K, int Cin) {
    const int kt = (K + 63) / 64, ctl = (Cin + 63) / 64;
    int nslices = 256 / (kt * ctl);
    if (nslices < 1) nslices = 1;
    return sizeof(float) * (size_t)nslices * kt * 64 * 9 * ctl * 64;
}

extern "C" int sf_wgrad3x3(const void* x, const void* dout, float* dw, float* ws, size_t ws_bytes, int N, int H, int W,
                           int Cx, int Kd, int K, int Cin, const int* dy, const int* dx, void* stream) {
    SF_REQUIRE(x && dout && dw && ws && dy && dx, "sf_wgrad3x3: null argument");
    SF_REQUIRE(N > 0 && H > 0 && W > 0 && K > 0 && Cin > 0, "sf_wgrad3x3: bad shape");
    SF_REQUIRE(Cx % 8 == 0 && Kd % 8 == 0 && Cin <= Cx && K <= Kd, "sf_wgrad3x3: channel strides must be multiples of 8 and cover the real counts");
    SF_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)dout % 16) == 0, "sf_wgrad3x3: operands must be 16-byte aligned");
    SF_REQUIRE(ws_bytes >= sf_wgrad3x3_workspace_bytes(N, H, W, K, Cin), "sf_wgrad3x3: workspace too small");
    W3Params p;
    p.dout = (const bf16*)dout; p.x = (const bf16*)x; p.ws = ws;
    p.N = N; p.H = H; p.W = W; p.Kd = Kd; p.Cx = Cx;
    for (int t = 0; t < 9; ++t) {
        SF_REQUIRE(dy[t] >= -1 && dy[t] <= 1 && dx[t] >= -1 && dx[t] <= 1, "sf_wgrad3x3: tap offsets must lie in [-1, 1]");
        p.dy[t] = dy[t]; p.dx[t] = dx[t];
    }
    pick_tile_w(H, W, &p.TH, &p.TW);
    p.tiles_x = (W + p.TW - 1) / p.TW;
    p.tiles_y = (H + p.TH - 1) / p.TH;
    const long long ntiles = (long long)N * p.tiles_x * p.tiles_y;
    SF_REQUIRE(ntiles < 0x7fffffffLL, "sf_wgrad3x3: too many tiles");
    p.ntiles = (int)ntiles;
    p.kt_tiles = (K + 63) / 64;
    p.ct_tiles = (Cin + 63) / 64;
    p.K64 = p.kt_tiles * 64;
    p.C64 = p.ct_tiles * 64;
    int nslices = 256 / (p.kt_tiles * p.ct_tiles);
    if (nslices < 1) nslices = 1;
    p.tiles_per_slice = (p.ntiles + nslices - 1) / nslices;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = (size_t)(256 + MAX_PPW) * 128;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3x3_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            sf_set_error("sf_wgrad3x3: cannot raise the dynamic LDS limit to %zu bytes", lds);
            return SF_ELAUNCH;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(wgrad3x3_kernel, dim3((unsigned)(nslices * p.kt_tiles * p.ct_tiles)), dim3(NTW), lds, s, p);
    const long long total = (long long)K * 9 * Cin;
    int grid = (int)((total + NTW - 1) / NTW); if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(wgrad3x3_reduce_kernel, dim3(grid), dim3(NTW), 0, s, ws, dw, nslices, p.K64, p.C64, K, Cin, 0);
    SF_LAUNCH_CHECK("sf_wgrad3x3");
    return SF_OK;
}
