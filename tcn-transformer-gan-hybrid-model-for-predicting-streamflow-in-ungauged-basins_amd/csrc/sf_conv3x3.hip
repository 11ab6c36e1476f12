// 3x3 stride-1 convolution (forward and data gradient) for bf16 NHWC tensors.
//
// A workgroup owns a TH x TW tile of output pixels (TH*TW <= 256) and a tile of KT output
// channels.  Per chunk of CK input channels it stages, once, the (TH+2) x (TW+2) halo patch of
// the input and the weights of all nine taps in LDS; the nine taps then read their B operand
// from the same patch at shifted pixel positions, so every input element crosses L2 -> LDS
// once per channel chunk instead of nine times, and one barrier pair covers 72 MFMAs per wave.
// Global loads of the next chunk are issued into registers before the MFMAs of the current
// chunk (latency hidden behind compute; a second resident workgroup covers the LDS store
// burst).  D is kept as [kout][pixel] so a lane stores 4 consecutive channels per access.
#include "sf_common.h"
#include "../../include/sfhip.h"

namespace {

constexpr int NT3 = 256;
constexpr int MAX_PP = 400;  // patch pixels the LDS budget allows

struct C3Params {
    const bf16* in;
    const bf16* w;
    const float* bias;
    const bf16* aux;
    void* out;
    int N, H, W, C, K, ldo;
    int TH, TW, tiles_x, tiles_y, ktiles;
    int act, aux_mode, out_dtype;
    int dy[9], dx[9];
};

template <int KT, int CK>
__global__ __launch_bounds__(NT3, 2) void conv3x3_kernel(C3Params p) {
    constexpr int LD = CK + 8;                 // padded row (elements): conflict-free b128 reads
    constexpr int CPR = CK / 8;                // 16-byte chunks per row
    constexpr int W_CHUNKS = 9 * KT * CPR;
    constexpr int NWS = (W_CHUNKS + NT3 - 1) / NT3;
    constexpr int NPS = (MAX_PP * CPR + NT3 - 1) / NT3;
    constexpr int MT = KT / 32;
    constexpr int KS = CK / 16;
    extern __shared__ __attribute__((aligned(16))) bf16 smem3[];
    bf16* sW = smem3;                  // [9][KT][LD]
    bf16* sP = smem3 + 9 * KT * LD;    // [PP][LD]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    int b = blockIdx.x;
    const int kt = b % p.ktiles; b /= p.ktiles;
    const int tx = b % p.tiles_x; b /= p.tiles_x;
    const int ty = b % p.tiles_y;
    const int n = b / p.tiles_y;
    const int y0 = ty * p.TH, x0 = tx * p.TW, k0 = kt * KT;
    const int PW = p.TW + 2, PP = (p.TH + 2) * PW;

    // ---- staging slots: patch
    unsigned poff[NPS];    // offset (in 16-byte units) of the slot's chunk at channel 0, or ~0u
    int plds[NPS];
#pragma unroll
    for (int i = 0; i < NPS; ++i) {
        const int q = tid + i * NT3;
        const int pp = q / CPR, cc = q % CPR;
        poff[i] = ~0u;
        plds[i] = -1;
        if (pp < PP) {
            const int py = pp / PW, px = pp - py * PW;
            const int gy = y0 - 1 + py, gx = x0 - 1 + px;
            plds[i] = pp * LD + cc * 8;
            if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)
                poff[i] = (unsigned)(((((long long)n * p.H + gy) * p.W + gx) * p.C) / 8 + cc);
        }
    }
    // ---- staging slots: weights (rows k*9+t in memory -> LDS row t*KT+k)
    int woff[NWS], wlds[NWS];
#pragma unroll
    for (int i = 0; i < NWS; ++i) {
        const int q = tid + i * NT3;
        woff[i] = -1;
        wlds[i] = -1;
        if (q < W_CHUNKS) {
            const int rw = q / CPR, cc = q % CPR;
            const int k = rw / 9, t = rw - k * 9;
            wlds[i] = (t * KT + k) * LD + cc * 8;
            if (k0 + k < p.K) woff[i] = ((k0 + k) * 9 + t) * p.C + cc * 8;
        }
    }

    uint4 rp[NPS], rw_[NWS];
    auto load_chunk = [&](int c0) {
#pragma unroll
        for (int i = 0; i < NPS; ++i) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (poff[i] != ~0u) v = reinterpret_cast<const uint4*>(p.in)[(size_t)poff[i] + (c0 >> 3)];
            rp[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NWS; ++i) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (woff[i] >= 0) v = *reinterpret_cast<const uint4*>(p.w + woff[i] + c0);
            rw_[i] = v;
        }
    };
    auto store_chunk_lds = [&]() {
#pragma unroll
        for (int i = 0; i < NPS; ++i)
            if (plds[i] >= 0) *reinterpret_cast<uint4*>(sP + plds[i]) = rp[i];
#pragma unroll
        for (int i = 0; i < NWS; ++i)
            if (wlds[i] >= 0) *reinterpret_cast<uint4*>(sW + wlds[i]) = rw_[i];
    };

    // ---- this lane's two output pixels (one per 32-pixel MFMA column tile)
    int pbase[2];
    int gpy[2], gpx[2];
    bool pok[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int q = wave * 64 + nt * 32 + r;
        const bool in_tile = q < p.TH * p.TW;
        const int qq = in_tile ? q : 0;
        const int ly = qq / p.TW, lx = qq - ly * p.TW;
        pbase[nt] = ((ly + 1) * PW + lx + 1) * LD + 8 * h;
        gpy[nt] = y0 + ly;
        gpx[nt] = x0 + lx;
        pok[nt] = in_tile && gpy[nt] < p.H && gpx[nt] < p.W;
    }
    int tapoff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) tapoff[t] = (p.dy[t] * PW + p.dx[t]) * LD;

    f32x16 acc[MT][2];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][c][e] = 0.f;

    const int nchunks = p.C / CK;
    load_chunk(0);
    for (int ci = 0; ci < nchunks; ++ci) {
        if (ci > 0) __syncthreads();
        store_chunk_lds();
        __syncthreads();
        if (ci + 1 < nchunks) load_chunk((ci + 1) * CK);
        const bf16* wrow = sW + r * LD + 8 * h;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8 fa[MT], fb[2];
#pragma unroll
                for (int a = 0; a < MT; ++a)
                    fa[a] = *reinterpret_cast<const bf16x8*>(wrow + (t * KT + a * 32) * LD + ks * 16);
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    fb[c] = *reinterpret_cast<const bf16x8*>(sP + pbase[c] + tapoff[t] + ks * 16);
#pragma unroll
                for (int a = 0; a < MT; ++a)
#pragma unroll
                    for (int c = 0; c < 2; ++c)
                        acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a], fb[c], acc[a][c], 0, 0, 0);
            }
        }
    }

    // ---- epilogue
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        if (!pok[c]) continue;
        const long long pix = ((long long)n * p.H + gpy[c]) * p.W + gpx[c];
#pragma unroll
        for (int a = 0; a < MT; ++a) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int k = k0 + a * 32 + 8 * g + 4 * h;
                if (k >= p.K) continue;
                const long long off = pix * p.ldo + k;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[a][c][4 * g + e];
                if (p.bias) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += p.bias[k + e];
                }
                if (p.act == SF_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                } else if (p.act == SF_ACT_LRELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.2f * v[e];
                }
                if (p.aux_mode) {
                    Vec4<bf16> ax;
                    ax.load(p.aux + off);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool pos = ax.get(e) > 0.f;
                        v[e] = pos ? v[e] : (p.aux_mode == 1 ? 0.f : 0.2f * v[e]);
                    }
                }
                if (p.out_dtype == SF_F32) {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + off) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    Vec4<bf16> o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o.set(e, v[e]);
                    o.store(reinterpret_cast<bf16*>(p.out) + off);
                }
            }
        }
    }
}

// tile shape with the fewest tiles (TH*TW <= 256 lanes, patch <= MAX_PP pixels)
void pick_tile(int H, int W, int* th, int* tw) {
    long long best = -1;
    int bth = 8, btw = 32, bpp = 0;
    for (int w = 8; w <= 128; ++w) {
        const int hh = 256 / w;
        if (hh < 1) continue;
        const int pp = (hh + 2) * (w + 2);
        if (pp > MAX_PP) continue;
        const long long tiles = (long long)((W + w - 1) / w) * ((H + hh - 1) / hh);
        if (best < 0 || tiles < best || (tiles == best && pp < bpp)) { best = tiles; bth = hh; btw = w; bpp = pp; }
    }
    *th = bth; *tw = btw;
}

template <int KT, int CK> int launch3(const C3Params& p, hipStream_t s) {
    constexpr int LD = CK + 8;
    const size_t lds = sizeof(bf16) * ((size_t)9 * KT * LD + (size_t)MAX_PP * LD);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_kernel<KT, CK>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            sf_set_error("sf_conv3x3: cannot raise the dynamic LDS limit to %zu bytes", lds);
            return SF_ELAUNCH;
        }
        attr_set = true;
    }
    const long long nblk = (long long)p.N * p.tiles_y * p.tiles_x * p.ktiles;
    if (nblk <= 0 || nblk > 0x7fffffffLL) { sf_set_error("sf_conv3x3: bad grid"); return SF_EINVAL; }
    hipLaunchKernelGGL((conv3x3_kernel<KT, CK>), dim3((unsigned)nblk), dim3(NT3), lds, s, p);
    SF_LAUNCH_CHECK("sf_conv3x3");
    return SF_OK;
}

}  // namespace

extern "C" int sf_conv3x3(const void* in, const void* wp, const float* bias, const void* aux, void* out, int N, int H,
                          int W, int C, int K, int ldo, const int* dy, const int* dx, int act, int aux_mode,
                          int out_dtype, void* stream) {
    SF_REQUIRE(in && wp && out && dy && dx, "sf_conv3x3: null argument");
    SF_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && K > 0, "sf_conv3x3: bad shape");
    SF_REQUIRE(C % 16 == 0, "sf_conv3x3: C=%d must be a multiple of 16", C);
    SF_REQUIRE(K % 4 == 0 && ldo >= K && ldo % 4 == 0, "sf_conv3x3: K=%d / ldo=%d must be multiples of 4", K, ldo);
    SF_REQUIRE(!aux_mode || aux, "sf_conv3x3: aux_mode without aux");
    SF_REQUIRE((long long)N * H * W * C / 8 < 0xFFFFFFFFLL, "sf_conv3x3: input larger than 64 GiB");
    SF_REQUIRE(((uintptr_t)in % 16) == 0 && ((uintptr_t)wp % 16) == 0, "sf_conv3x3: operands must be 16-byte aligned");
    C3Params p;
    p.in = (const bf16*)in; p.w = (const bf16*)wp; p.bias = bias; p.aux = (const bf16*)aux; p.out = out;
    p.N = N; p.H = H; p.W = W; p.C = C; p.K = K; p.ldo = ldo;
    p.act = act; p.aux_mode = aux_mode; p.out_dtype = out_dtype;
    for (int t = 0; t < 9; ++t) {
        SF_REQUIRE(dy[t] >= -1 && dy[t] <= 1 && dx[t] >= -1 && dx[t] <= 1, "sf_conv3x3: tap offsets must lie in [-1, 1]");
        p.dy[t] = dy[t]; p.dx[t] = dx[t];
    }
    pick_tile(H, W, &p.TH, &p.TW);
    p.tiles_x = (W + p.TW - 1) / p.TW;
    p.tiles_y = (H + p.TH - 1) / p.TH;
    hipStream_t s = (hipStream_t)stream;
    if (K <= 64 && C % 32 == 0) {
        p.ktiles = (K + 63) / 64;
        return launch3<64, 32>(p, s);
    }
    p.ktiles = (K + 127) / 128;
    return launch3<128, 16>(p, s);
}
