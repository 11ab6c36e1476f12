// BatchNorm passes over NHWC tensors: batch statistics, bias-gradient sums, BN + activation
// (+ pooling) forward, and the two-pass BN backward fused with the backward of ReLU + maxpool,
// ReLU + adaptive average pool or LeakyReLU.  All of them are HBM-bound streams.
//
// Thread mapping (every kernel here): a thread owns one 16-byte channel chunk `cg` for the
// whole kernel (per-channel constants live in registers) and walks rows phase, phase + rpp, ...
// of the workgroup's contiguous row range.  Rows are processed in batches of U: the U loads
// are issued back to back and without branches (out-of-range rows read a zero chunk), so a
// thread keeps U (or 2U, 5U) 16-byte loads in flight.  Index arithmetic is 32-bit.
// Reductions: f32 inside a batch, f64 across batches, shuffles + LDS inside the workgroup in a
// fixed order, one hardware f64 atomic per channel and workgroup at the end.
#include "sf_common.h"
#include "../../include/sfhip.h"

namespace {

constexpr int BT = 512;           // threads per workgroup
constexpr int MAX_AXIS = 1024;    // avgpool backward: largest H / W the LDS bin tables hold
constexpr int MAX_BINS = 256;

__device__ __attribute__((aligned(16))) uint4 g_zero16;  // never written: stays zero

template <typename T> struct Chunk;
template <> struct Chunk<float> {
    static constexpr int EPC = 4;
    static __device__ __forceinline__ void decode(const uint4& r, float* f) {
        f[0] = __uint_as_float(r.x); f[1] = __uint_as_float(r.y);
        f[2] = __uint_as_float(r.z); f[3] = __uint_as_float(r.w);
    }
    static __device__ __forceinline__ uint4 encode(const float* f) {
        return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
    }
};
template <> struct Chunk<bf16> {
    static constexpr int EPC = 8;
    static __device__ __forceinline__ void decode(const uint4& r, float* f) {
        f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
        f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
        f[4] = __uint_as_float(r.z << 16); f[5] = __uint_as_float(r.z & 0xffff0000u);
        f[6] = __uint_as_float(r.w << 16); f[7] = __uint_as_float(r.w & 0xffff0000u);
    }
    static __device__ __forceinline__ uint4 encode(const float* f) {
        bf16x8 v;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (bf16)f[i];
        return *reinterpret_cast<const uint4*>(&v);
    }
};

__device__ __forceinline__ uint4 ld16(const void* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ const void* zsrc() { return &g_zero16; }

struct RowMap {
    int G, rpp, cg, phase;
    bool active;
    __device__ __forceinline__ RowMap(int C, int epc) {
        G = C / epc;
        rpp = BT / G;
        cg = threadIdx.x % G;
        phase = threadIdx.x / G;
        active = phase < rpp;
    }
};

// Sum of v over the threads of the workgroup that own the same channel chunk; valid in
// threads tid < G (for chunk tid).  Fixed summation order.
__device__ __forceinline__ double group_sum(double v, const RowMap& m, double* sm /* BT doubles */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int nrows;
    if ((m.G & (m.G - 1)) == 0 && m.G < 64) {
        for (int o = 32; o >= m.G; o >>= 1) v += __shfl_xor(v, o, 64);
        nrows = BT / 64;
        __syncthreads();
        if (lane < m.G) sm[wave * m.G + lane] = v;
    } else {
        nrows = m.rpp;
        __syncthreads();
        if (m.active) sm[m.phase * m.G + m.cg] = v;
    }
    __syncthreads();
    double t = 0.0;
    if ((int)threadIdx.x < m.G)
        for (int i = 0; i < nrows; ++i) t += sm[i * m.G + threadIdx.x];
    return t;
}

// ------------------------------------------------------------------------------ column sums
template <typename T, bool SQ>
__global__ __launch_bounds__(BT) void colsum_kernel(const T* __restrict__ x, unsigned M, unsigned per, int C,
                                                    double* __restrict__ ws) {
    constexpr int EPC = Chunk<T>::EPC, U = 8;
    const RowMap m(C, EPC);
    __shared__ double sm[BT];
    double s[EPC], q[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) s[e] = q[e] = 0.0;
    const unsigned r0 = blockIdx.x * per;
    const unsigned r1 = r0 + per < M ? r0 + per : M;
    if (m.active) {
        const T* base = x + m.cg * EPC;
        for (unsigned r = r0 + m.phase; r < r1; r += m.rpp * U) {
            uint4 raw[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned rr = r + u * m.rpp;
                raw[u] = ld16(rr < r1 ? (const void*)(base + (size_t)rr * C) : zsrc());
            }
            float ps[EPC], pq[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) ps[e] = pq[e] = 0.f;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float f[EPC];
                Chunk<T>::decode(raw[u], f);
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    ps[e] += f[e];
                    if (SQ) pq[e] += f[e] * f[e];
                }
            }
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                s[e] += (double)ps[e];
                if (SQ) q[e] += (double)pq[e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        // this block's row of the partial-sum slab (row 0 of ws is the folded result)
        double* row = ws + (size_t)(1 + blockIdx.x) * ((SQ ? 2 : 1) * C);
        const double ts = group_sum(s[e], m, sm);
        if ((int)threadIdx.x < m.G) row[threadIdx.x * EPC + e] = ts;
        if (SQ) {
            const double tq = group_sum(q[e], m, sm);
            if ((int)threadIdx.x < m.G) row[C + threadIdx.x * EPC + e] = tq;
        }
    }
}

// ws[j] = sum over the `parts` slab rows ws[(1 + p) * n + j], j < n: 16 columns x 16 row groups per
// block, eight loads in flight, fixed order (the sums are bit-reproducible).
__global__ __launch_bounds__(256) void ws_fold_kernel(double* __restrict__ ws, int n, int parts) {
    __shared__ double sm[16][17];
    const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int j = blockIdx.x * 16 + cl;
    const int jj = j < n ? j : n - 1;
    const double* slab = ws + n;
    double a = 0.0;
    for (int p = rg; p < parts; p += 16 * 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int pp = p + 16 * u;
            v[u] = slab[(size_t)(pp < parts ? pp : parts - 1) * n + jj];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) a += (p + 16 * u < parts) ? v[u] : 0.0;
    }
    sm[rg][cl] = a;
    __syncthreads();
    if (rg == 0 && j < n) {
        double t = 0.0;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += sm[g][cl];
        ws[j] = t;
    }
}

__global__ void ws_to_float_kernel(const double* ws, float* out, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) out[i] = (float)ws[i];
}

// ------------------------------------------------------------------------------ forward
// y = maxpool2x2(relu(x*scale+shift)); item = one pooling window
template <typename T>
__global__ __launch_bounds__(BT) void bn_relu_maxpool_fwd_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, T* __restrict__ y,
                                                                 unsigned items, unsigned per, int H, int W, int C) {
    constexpr int EPC = Chunk<T>::EPC, U = 2;
    const RowMap m(C, EPC);
    if (!m.active) return;
    const unsigned PH = H / 2, PW = W / 2;
    float sc[EPC], sf[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) { sc[e] = scale[m.cg * EPC + e]; sf[e] = shift[m.cg * EPC + e]; }
    const unsigned r0 = blockIdx.x * per;
    const unsigned r1 = r0 + per < items ? r0 + per : items;
    const T* base = x + m.cg * EPC;
    for (unsigned r = r0 + m.phase; r < r1; r += m.rpp * U) {
        uint4 raw[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned rr = r + u * m.rpp;
            const bool ok = rr < r1;
            const unsigned t = rr / PW, pw = rr - t * PW;
            const unsigned n = t / PH, ph = t - n * PH;
            const size_t pix = ((size_t)n * H + 2 * ph) * W + 2 * pw;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                raw[u][k] = ld16(ok ? (const void*)(base + (pix + (k >> 1) * W + (k & 1)) * C) : zsrc());
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned rr = r + u * m.rpp;
            if (rr >= r1) break;
            float mx[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) mx[e] = 0.f;  // max with 0 == ReLU then max
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float f[EPC];
                Chunk<T>::decode(raw[u][k], f);
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const float v = f[e] * sc[e] + sf[e];
                    mx[e] = (v > mx[e] || v != v) ? v : mx[e];   // a NaN wins and stays, as in torch max_pool2d
                }
            }
            *reinterpret_cast<uint4*>(y + (size_t)rr * C + m.cg * EPC) = Chunk<T>::encode(mx);
        }
    }
}

// y = leaky_relu(x*scale+shift, slope) (slope 0: ReLU); item = one row
template <typename T>
__global__ __launch_bounds__(BT) void bn_lrelu_fwd_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, T* __restrict__ y,
                                                          unsigned M, unsigned per, int C, float slope) {
    constexpr int EPC = Chunk<T>::EPC, U = 4;
    const RowMap m(C, EPC);
    if (!m.active) return;
    float sc[EPC], sf[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) { sc[e] = scale[m.cg * EPC + e]; sf[e] = shift[m.cg * EPC + e]; }
    const unsigned r0 = blockIdx.x * per;
    const unsigned r1 = r0 + per < M ? r0 + per : M;
    for (unsigned r = r0 + m.phase; r < r1; r += m.rpp * U) {
        uint4 raw[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned rr = r + u * m.rpp;
            raw[u] = ld16(rr < r1 ? (const void*)(x + (size_t)rr * C + m.cg * EPC) : zsrc());
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned rr = r + u * m.rpp;
            if (rr >= r1) break;
            float f[EPC];
            Chunk<T>::decode(raw[u], f);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const float v = f[e] * sc[e] + sf[e];
                f[e] = v > 0.f ? v : slope * v;
            }
            *reinterpret_cast<uint4*>(y + (size_t)rr * C + m.cg * EPC) = Chunk<T>::encode(f);
        }
    }
}

// ------------------------------------------------------------------------------ backward
struct BnBwd {
    const void* x; const void* dy;
    const float* scale; const float* shift; const float* mean; const float* invstd; const float* gamma;
    int N, H, W, C, Creal, OH, OW, training;
    unsigned items, per;
    float slope;   // negative-side slope of the activation in mode 2 (LeakyReLU 0.2) / 3 (ReLU: 0)
};

// adaptive average pool bins (ATen): bin i covers [floor(i*in/out), ceil((i+1)*in/out))
__device__ __forceinline__ int bin_start(int i, int in, int out) { return (i * in) / out; }
__device__ __forceinline__ int bin_end(int i, int in, int out) { return ((i + 1) * in + out - 1) / out; }

// LDS tables of the avgpool backward: for every input row / column the range of bins that
// contain it, and for every bin the reciprocal of its length.
struct PoolTables {
    int hb[MAX_AXIS], wb[MAX_AXIS];
    float ih[MAX_BINS], iw[MAX_BINS];
};

__device__ __forceinline__ void build_axis(int* tab, float* inv, int in, int out) {
    for (int i = threadIdx.x; i < out; i += BT) inv[i] = 1.0f / (float)(bin_end(i, in, out) - bin_start(i, in, out));
    for (int p = threadIdx.x; p < in; p += BT) {
        int lo = (p * out) / in - 1; if (lo < 0) lo = 0;
        int hi = ((p + 1) * out + in - 1) / in; if (hi > out - 1) hi = out - 1;
        while (lo < hi && !(p >= bin_start(lo, in, out) && p < bin_end(lo, in, out))) ++lo;
        while (hi > lo && !(p >= bin_start(hi, in, out) && p < bin_end(hi, in, out))) --hi;
        tab[p] = lo | (hi << 16);
    }
}

// Per-thread channel constants
template <int EPC> struct Chan {
    float sc[EPC], sf[EPC], mu[EPC], is[EPC];
    __device__ __forceinline__ void load(const BnBwd& a, int cg) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const int c = cg * EPC + e;
            sc[e] = a.scale[c]; sf[e] = a.shift[c]; mu[e] = a.mean[c]; is[e] = a.invstd[c];
        }
    }
};

// One batch of U items for the backward kernels: loads first, then dz (gradient wrt the BN
// output) and x per pre-BN pixel.  MODE 0 items are 2x2 pooling windows (4 pixels each, the
// window grid covers odd trailing rows / columns too), MODE 1 / 2 items are pixels.
template <typename T, int MODE, int UU = 0> struct BwdBatch {
    static constexpr int EPC = Chunk<T>::EPC;
    static constexpr int U = UU ? UU : (MODE == 0 || MODE == 4) ? 2 : 4;
    static constexpr int PIX = (MODE == 0 || MODE == 4) ? 4 : 1;
    float xf[U][PIX][EPC], dz[U][PIX][EPC];
    unsigned opix[U][PIX];  // pre-BN pixel index (row of x / dx), ~0u when invalid

    __device__ __forceinline__ void run(const BnBwd& a, const RowMap& m, const Chan<EPC>& ch, unsigned r, unsigned r1,
                                        const PoolTables* tabs) {
        const T* xb = reinterpret_cast<const T*>(a.x) + m.cg * EPC;
        const T* gb = reinterpret_cast<const T*>(a.dy) + m.cg * EPC;
        if constexpr (MODE == 4) {
            // avgpool backward over groups of four consecutive pixels of one image row: the
            // group lies in one or two row bins and (checked by the launcher) in at most three
            // consecutive column bins, so 2 x 3 pooled-gradient chunks serve all four pixels.
            const unsigned GW = (a.W + 3) / 4;
            uint4 rx[U][4], rg[U][2][3];
            float wh[U][2], wc[U][4][3];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned rr = r + u * m.rpp;
                const bool ok = rr < r1;
                const unsigned rc = ok ? rr : r;
                const unsigned t = rc / GW, gw = rc - t * GW;
                const unsigned n = t / (unsigned)a.H, hh = t - n * a.H;
                const int hbin = tabs->hb[hh];
                const int i0 = hbin & 0xffff, i1 = hbin >> 16;
                wh[u][0] = tabs->ih[i0];
                wh[u][1] = i1 > i0 ? tabs->ih[i1] : 0.f;
                const int jlo = tabs->wb[4 * gw] & 0xffff;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned w = 4 * gw + k;
                    const bool v = ok && w < (unsigned)a.W;
                    const unsigned pix = (n * a.H + hh) * a.W + w;
                    opix[u][k] = v ? pix : ~0u;
                    rx[u][k] = ld16(v ? (const void*)(xb + (size_t)pix * a.C) : zsrc());
                    const int wbin = tabs->wb[v ? w : 4 * gw];
                    const int j0 = wbin & 0xffff, j1 = wbin >> 16;
                    const float f0 = tabs->iw[j0], f1 = j1 > j0 ? tabs->iw[j1] : 0.f;
#pragma unroll
                    for (int b = 0; b < 3; ++b)
                        wc[u][k][b] = (jlo + b == j0 ? f0 : 0.f) + (jlo + b == j1 ? f1 : 0.f);
                }
                const T* gn = gb + (size_t)n * a.OH * a.OW * a.C;
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    const int j = jlo + b < a.OW ? jlo + b : a.OW - 1;   // weight is zero past the last bin
                    rg[u][0][b] = ld16(gn + (size_t)(i0 * a.OW + j) * a.C);
                    rg[u][1][b] = ld16(gn + (size_t)(i1 * a.OW + j) * a.C);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float G[3][EPC];
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    float g0[EPC], g1[EPC];
                    Chunk<T>::decode(rg[u][0][b], g0);
                    Chunk<T>::decode(rg[u][1][b], g1);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) G[b][e] = g0[e] * wh[u][0] + g1[e] * wh[u][1];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    Chunk<T>::decode(rx[u][k], xf[u][k]);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        const float acc = G[0][e] * wc[u][k][0] + G[1][e] * wc[u][k][1] + G[2][e] * wc[u][k][2];
                        const float y = xf[u][k][e] * ch.sc[e] + ch.sf[e];
                        dz[u][k][e] = y > 0.f ? acc : 0.f;
                    }
                }
            }
        } else if constexpr (MODE == 2) {
            uint4 rx[U], rg[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned rr = r + u * m.rpp;
                const bool ok = rr < r1;
                opix[u][0] = ok ? rr : ~0u;
                rx[u] = ld16(ok ? (const void*)(xb + (size_t)rr * a.C) : zsrc());
                rg[u] = ld16(ok ? (const void*)(gb + (size_t)rr * a.C) : zsrc());
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float g[EPC];
                Chunk<T>::decode(rx[u], xf[u][0]);
                Chunk<T>::decode(rg[u], g);
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const float y = xf[u][0][e] * ch.sc[e] + ch.sf[e];
                    dz[u][0][e] = y > 0.f ? g[e] : a.slope * g[e];
                }
            }
        } else if constexpr (MODE == 1) {
            uint4 rx[U];
            unsigned nn[U], hh[U], ww[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned rr = r + u * m.rpp;
                const bool ok = rr < r1;
                opix[u][0] = ok ? rr : ~0u;
                const unsigned rc = ok ? rr : r;
                const unsigned t = rc / (unsigned)a.W;
                ww[u] = rc - t * a.W;
                nn[u] = t / (unsigned)a.H;
                hh[u] = t - nn[u] * a.H;
                rx[u] = ld16(ok ? (const void*)(xb + (size_t)rr * a.C) : zsrc());
            }
            if (a.H >= a.OH && a.W >= a.OW) {
                // bins overlap by at most one element: a pixel lies in one or two bins per axis.
                // Always fetch the 2 x 2 candidates (a missing second bin repeats the first with
                // weight zero) so that the loads are unconditional and issued together.
                uint4 rg[U][4];
                float wg[U][4];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int hbin = tabs->hb[hh[u]], wbin = tabs->wb[ww[u]];
                    const int i0 = hbin & 0xffff, i1 = hbin >> 16, j0 = wbin & 0xffff, j1 = wbin >> 16;
                    const float wh0 = tabs->ih[i0], wh1 = i1 > i0 ? tabs->ih[i1] : 0.f;
                    const float ww0 = tabs->iw[j0], ww1 = j1 > j0 ? tabs->iw[j1] : 0.f;
                    const T* gn = gb + (size_t)nn[u] * a.OH * a.OW * a.C;
                    rg[u][0] = ld16(gn + (size_t)(i0 * a.OW + j0) * a.C);
                    rg[u][1] = ld16(gn + (size_t)(i0 * a.OW + j1) * a.C);
                    rg[u][2] = ld16(gn + (size_t)(i1 * a.OW + j0) * a.C);
                    rg[u][3] = ld16(gn + (size_t)(i1 * a.OW + j1) * a.C);
                    wg[u][0] = wh0 * ww0; wg[u][1] = wh0 * ww1; wg[u][2] = wh1 * ww0; wg[u][3] = wh1 * ww1;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    Chunk<T>::decode(rx[u], xf[u][0]);
                    float acc[EPC];
#pragma unroll
                    for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float g[EPC];
                        Chunk<T>::decode(rg[u][k], g);
#pragma unroll
                        for (int e = 0; e < EPC; ++e) acc[e] += g[e] * wg[u][k];
                    }
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        const float y = xf[u][0][e] * ch.sc[e] + ch.sf[e];
                        dz[u][0][e] = y > 0.f ? acc[e] : 0.f;
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    Chunk<T>::decode(rx[u], xf[u][0]);
                    float acc[EPC];
#pragma unroll
                    for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
                    const int hbin = tabs->hb[hh[u]], wbin = tabs->wb[ww[u]];
                    for (int i = hbin & 0xffff; i <= (hbin >> 16); ++i)
                        for (int j = wbin & 0xffff; j <= (wbin >> 16); ++j) {
                            float g[EPC];
                            Chunk<T>::decode(ld16(gb + (((size_t)nn[u] * a.OH + i) * a.OW + j) * a.C), g);
                            const float wgt = tabs->ih[i] * tabs->iw[j];
#pragma unroll
                            for (int e = 0; e < EPC; ++e) acc[e] += g[e] * wgt;
                        }
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        const float y = xf[u][0][e] * ch.sc[e] + ch.sf[e];
                        dz[u][0][e] = y > 0.f ? acc[e] : 0.f;
                    }
                }
            }
        } else {
            const unsigned WH = (a.H + 1) / 2, WW = (a.W + 1) / 2, PH = a.H / 2, PW = a.W / 2;
            uint4 rx[U][4], rg[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned rr = r + u * m.rpp;
                const bool ok = rr < r1;
                const unsigned rc = ok ? rr : r;
                const unsigned t = rc / WW, pw = rc - t * WW;
                const unsigned n = t / WH, ph = t - n * WH;
                const bool pooled = ok && ph < PH && pw < PW;
                rg[u] = ld16(pooled ? (const void*)(gb + (((size_t)n * PH + ph) * PW + pw) * a.C) : zsrc());
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned h = 2 * ph + (k >> 1), w = 2 * pw + (k & 1);
                    const bool v = ok && h < (unsigned)a.H && w < (unsigned)a.W;
                    const unsigned pix = (n * a.H + h) * a.W + w;
                    opix[u][k] = v ? pix : ~0u;
                    rx[u][k] = ld16(v ? (const void*)(xb + (size_t)pix * a.C) : zsrc());
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float g[EPC];
                Chunk<T>::decode(rg[u], g);
#pragma unroll
                for (int k = 0; k < 4; ++k) Chunk<T>::decode(rx[u][k], xf[u][k]);
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    float best = xf[u][0][e] * ch.sc[e] + ch.sf[e];
                    int arg = 0;
#pragma unroll
                    for (int k = 1; k < 4; ++k) {
                        const float v = xf[u][k][e] * ch.sc[e] + ch.sf[e];
                        if (v > best) { best = v; arg = k; }
                    }
                    // ReLU precedes the pool: a window whose maximum is <= 0 passes no gradient.
                    // g is zero for windows outside the pooled map (odd trailing row / column).
                    const float gv = best > 0.f ? g[e] : 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) dz[u][k][e] = (k == arg) ? gv : 0.f;
                }
            }
        }
    }
};

template <typename T, int MODE>
__global__ __launch_bounds__(BT) void bn_bwd_reduce_kernel(BnBwd a, double* __restrict__ ws) {
    // mode 4: one group per batch (two would spill next to the 2 * EPC f64 sums)
    using Batch = BwdBatch<T, MODE, MODE == 4 ? 1 : 0>;
    constexpr int EPC = Batch::EPC, U = Batch::U, PIX = Batch::PIX;
    const RowMap m(a.C, EPC);
    __shared__ double sm[BT];
    PoolTables* tabs = nullptr;
    if constexpr (MODE == 1 || MODE == 4) {
        __shared__ PoolTables tabs_s;
        tabs = &tabs_s;
        build_axis(tabs->hb, tabs->ih, a.H, a.OH);
        build_axis(tabs->wb, tabs->iw, a.W, a.OW);
        __syncthreads();
    }
    double s1[EPC], s2[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) s1[e] = s2[e] = 0.0;
    const unsigned r0 = blockIdx.x * a.per;
    const unsigned r1 = r0 + a.per < a.items ? r0 + a.per : a.items;
    if (m.active) {
        Chan<EPC> ch;
        ch.load(a, m.cg);
        for (unsigned r = r0 + m.phase; r < r1; r += m.rpp * U) {
            Batch b;
            b.run(a, m, ch, r, r1, tabs);
            float p1[EPC], p2[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) p1[e] = p2[e] = 0.f;
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int k = 0; k < PIX; ++k)
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        // invalid pixels carry dz = 0 (zero chunk / zero gradient)
                        const float d = b.opix[u][k] != ~0u ? b.dz[u][k][e] : 0.f;
                        p1[e] += d;
                        p2[e] += d * ((b.xf[u][k][e] - ch.mu[e]) * ch.is[e]);
                    }
#pragma unroll
            for (int e = 0; e < EPC; ++e) { s1[e] += (double)p1[e]; s2[e] += (double)p2[e]; }
        }
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        double* row = ws + (size_t)(1 + blockIdx.x) * (2 * a.C);   // this block's row of the partial-sum slab
        const double t1 = group_sum(s1[e], m, sm);
        if ((int)threadIdx.x < m.G) row[threadIdx.x * EPC + e] = t1;
        const double t2 = group_sum(s2[e], m, sm);
        if ((int)threadIdx.x < m.G) row[a.C + threadIdx.x * EPC + e] = t2;
    }
}

template <typename T, int MODE>
__global__ __launch_bounds__(BT) void bn_bwd_apply_kernel(BnBwd a, const double* __restrict__ ws, T* __restrict__ dx,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta) {
    using Batch = BwdBatch<T, MODE>;
    constexpr int EPC = Batch::EPC, U = Batch::U, PIX = Batch::PIX;
    const RowMap m(a.C, EPC);
    PoolTables* tabs = nullptr;
    if constexpr (MODE == 1 || MODE == 4) {
        __shared__ PoolTables tabs_s;
        tabs = &tabs_s;
        build_axis(tabs->hb, tabs->ih, a.H, a.OH);
        build_axis(tabs->wb, tabs->iw, a.W, a.OW);
        __syncthreads();
    }
    if (blockIdx.x == 0) {
        for (int c = threadIdx.x; c < a.Creal; c += BT) {
            if (dgamma) dgamma[c] = (float)ws[a.C + c];
            if (dbeta) dbeta[c] = (float)ws[c];
        }
    }
    if (!m.active) return;
    Chan<EPC> ch;
    ch.load(a, m.cg);
    float m1[EPC], m2[EPC];
    const double inv_m = 1.0 / ((double)a.N * a.H * a.W);
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        const int c = m.cg * EPC + e;
        m1[e] = a.training ? (float)(ws[c] * inv_m) : 0.f;
        m2[e] = a.training ? (float)(ws[a.C + c] * inv_m) : 0.f;
    }
    const unsigned r0 = blockIdx.x * a.per;
    const unsigned r1 = r0 + a.per < a.items ? r0 + a.per : a.items;
    for (unsigned r = r0 + m.phase; r < r1; r += m.rpp * U) {
        Batch b;
        b.run(a, m, ch, r, r1, tabs);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int k = 0; k < PIX; ++k) {
                if (b.opix[u][k] == ~0u) continue;
                float out[EPC];
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const float xh = (b.xf[u][k][e] - ch.mu[e]) * ch.is[e];
                    out[e] = ch.sc[e] * (b.dz[u][k][e] - m1[e] - xh * m2[e]);
                }
                *reinterpret_cast<uint4*>(dx + (size_t)b.opix[u][k] * a.C + m.cg * EPC) = Chunk<T>::encode(out);
            }
    }
}

// ------------------------------------------------------------------------------ avgpool with bin moments
// The two channel sums of the BatchNorm backward behind ReLU + adaptive average pool are linear
// in the pooled gradient g:  sum_p dz = sum_bin g * (positives of the bin / bin size) and
// sum_p dz*xhat = sum_bin g * (sum of xhat over the bin's positives / bin size).  The forward pass
// already walks every bin, so it leaves those two pooled-size factors behind (pos: count of
// pixels with relu input > 0, u16; xh: mean over the bin of [y > 0] * xhat, compute dtype, both
// NHWC [N,OH,OW,C]) and the backward reduction reads three pooled maps instead of the pre-BN map.
template <int EPC> struct PosChunk;
template <> struct PosChunk<8> {
    using Raw = uint4;
    static __device__ __forceinline__ Raw pack(const int* c) {
        auto two = [](int lo, int hi) { return (unsigned)lo | ((unsigned)hi << 16); };     // counts are at most 65535
        return make_uint4(two(c[0], c[1]), two(c[2], c[3]), two(c[4], c[5]), two(c[6], c[7]));
    }
    static __device__ __forceinline__ void unpack(const Raw& r, float* f) {
        f[0] = (float)(r.x & 0xffffu); f[1] = (float)(r.x >> 16); f[2] = (float)(r.y & 0xffffu); f[3] = (float)(r.y >> 16);
        f[4] = (float)(r.z & 0xffffu); f[5] = (float)(r.z >> 16); f[6] = (float)(r.w & 0xffffu); f[7] = (float)(r.w >> 16);
    }
};
template <> struct PosChunk<4> {
    using Raw = uint2;
    static __device__ __forceinline__ Raw pack(const int* c) {
        return make_uint2((unsigned)c[0] | ((unsigned)c[1] << 16), (unsigned)c[2] | ((unsigned)c[3] << 16));
    }
    static __device__ __forceinline__ void unpack(const Raw& r, float* f) {
        f[0] = (float)(r.x & 0xffffu); f[1] = (float)(r.x >> 16); f[2] = (float)(r.y & 0xffffu); f[3] = (float)(r.y >> 16);
    }
};

constexpr int POOL_CS = 256;   // channels per LDS slab of the NCHW transpose
constexpr int POOL_BT = 256;   // threads of the pooling forward (its register count allows three such workgroups per CU)

// One workgroup per (image, output row): a thread owns (channel chunk, output column), walks the
// bin two rows x four columns at a time (eight unconditional loads in flight), and the pooled
// activations go through an LDS tile so that y leaves as NCHW rows.  Per element the loop keeps
// three sums (relu(v), [v > 0] * (x - mean), [v > 0]); invstd and 1 / |bin| are applied per bin.
// FULL: every bin is exactly 4 x 4 (no predicates).
template <typename T, typename TO, bool FULL>
__global__ __launch_bounds__(POOL_BT) void bn_relu_avgpool_mom_fwd_kernel(
    const T* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ mean, const float* __restrict__ invstd, TO* __restrict__ y,
    unsigned short* __restrict__ pos, T* __restrict__ xh, int H, int W, int C, int Creal, int OH, int OW) {
    constexpr int EPC = Chunk<T>::EPC;
    extern __shared__ float pool_tile[];   // [POOL_CS][OW + 1]
    const int n = blockIdx.x / OH, oh = blockIdx.x % OH;
    const int h0 = bin_start(oh, H, OH), h1 = bin_end(oh, H, OH);
    const int ldt = OW + 1;
    const T* img = x + (size_t)n * H * W * C;
    for (int c0 = 0; c0 < C; c0 += POOL_CS) {
        const int cs = (C - c0) < POOL_CS ? (C - c0) : POOL_CS;
        const int gs = cs / EPC;
        __syncthreads();
        for (int it = threadIdx.x; it < gs * OW; it += POOL_BT) {
            const int cg = it % gs, ow = it / gs;
            const int w0 = bin_start(ow, W, OW), w1 = bin_end(ow, W, OW);
            const int cb = c0 + cg * EPC;
            float sc[EPC], sf[EPC], mu[EPC], acc[EPC], accx[EPC];
            int cnt[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                sc[e] = scale[cb + e]; sf[e] = shift[cb + e]; mu[e] = mean[cb + e];
                acc[e] = accx[e] = 0.f; cnt[e] = 0;
            }
            const T* base = img + cb;
            for (int h = h0; h < h1; h += 2)
                for (int wq = w0; wq < w1; wq += 4) {
                    uint4 raw[2][4];
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const bool in = FULL || (h + j < h1 && wq + k < w1);
                            raw[j][k] = ld16(in ? (const void*)(base + ((size_t)(h + j) * W + wq + k) * C) : zsrc());
                        }
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const bool in = FULL || (h + j < h1 && wq + k < w1);
                            float f[EPC];
                            Chunk<T>::decode(raw[j][k], f);
#pragma unroll
                            for (int e = 0; e < EPC; ++e) {
                                const float v = f[e] * sc[e] + sf[e];
                                const bool p = in && !(v <= 0.f);   // one compare; true for NaN, so the ReLU keeps it
                                acc[e] += p ? v : 0.f;
                                accx[e] += p ? f[e] - mu[e] : 0.f;   // centred: no cancellation against the mean
                                cnt[e] += p;
                            }
                        }
                }
            const float inv = 1.0f / (float)((h1 - h0) * (w1 - w0));
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                pool_tile[(cg * EPC + e) * ldt + ow] = acc[e] * inv;
                accx[e] *= invstd[cb + e] * inv;
            }
            const size_t row = ((size_t)n * OH + oh) * OW + ow;
            *reinterpret_cast<typename PosChunk<EPC>::Raw*>(pos + row * C + cb) = PosChunk<EPC>::pack(cnt);
            *reinterpret_cast<uint4*>(xh + row * C + cb) = Chunk<T>::encode(accx);
        }
        __syncthreads();
        for (int it = threadIdx.x; it < cs * OW; it += POOL_BT) {
            const int ow = it % OW, c = it / OW;
            if (c0 + c < Creal) y[(((size_t)n * Creal + c0 + c) * OH + oh) * OW + ow] = from_f32<TO>(pool_tile[c * ldt + ow]);
        }
    }
}

// ws slab row of this workgroup: [0,C) sum_rows g * pos / bin size, [C,2C) sum_rows g * xh
template <typename T>
__global__ __launch_bounds__(BT) void bn_bwd_reduce_pooled_kernel(const T* __restrict__ dy,
                                                                  const unsigned short* __restrict__ pos,
                                                                  const T* __restrict__ xh, unsigned M, unsigned per,
                                                                  int C, int H, int W, int OH, int OW,
                                                                  double* __restrict__ ws) {
    constexpr int EPC = Chunk<T>::EPC, U = 4;
    using PC = PosChunk<EPC>;
    const RowMap m(C, EPC);
    __shared__ double sm[BT];
    __shared__ float ih[MAX_BINS], iw[MAX_BINS];
    for (int i = threadIdx.x; i < OH; i += BT) ih[i] = 1.0f / (float)(bin_end(i, H, OH) - bin_start(i, H, OH));
    for (int i = threadIdx.x; i < OW; i += BT) iw[i] = 1.0f / (float)(bin_end(i, W, OW) - bin_start(i, W, OW));
    __syncthreads();
    double s1[EPC], s2[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) s1[e] = s2[e] = 0.0;
    const unsigned r0 = blockIdx.x * per;
    const unsigned r1 = r0 + per < M ? r0 + per : M;
    if (m.active) {
        const size_t co = (size_t)m.cg * EPC;
        for (unsigned r = r0 + m.phase; r < r1; r += m.rpp * U) {
            uint4 rg[U], rx[U];
            typename PC::Raw rp[U];
            float wgt[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned rr = r + u * m.rpp;
                const bool ok = rr < r1;
                const unsigned rc = ok ? rr : r;
                rg[u] = ld16(ok ? (const void*)(dy + (size_t)rr * C + co) : zsrc());
                rx[u] = ld16(ok ? (const void*)(xh + (size_t)rr * C + co) : zsrc());
                rp[u] = *reinterpret_cast<const typename PC::Raw*>(ok ? (const void*)(pos + (size_t)rr * C + co) : zsrc());
                const unsigned t = rc / (unsigned)OW;
                wgt[u] = ih[t % (unsigned)OH] * iw[rc - t * OW];
            }
            float p1[EPC], p2[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) p1[e] = p2[e] = 0.f;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float g[EPC], xf[EPC], cf[EPC];
                Chunk<T>::decode(rg[u], g);
                Chunk<T>::decode(rx[u], xf);
                PC::unpack(rp[u], cf);
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    p1[e] += g[e] * (cf[e] * wgt[u]);
                    p2[e] += g[e] * xf[e];
                }
            }
#pragma unroll
            for (int e = 0; e < EPC; ++e) { s1[e] += (double)p1[e]; s2[e] += (double)p2[e]; }
        }
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        double* row = ws + (size_t)(1 + blockIdx.x) * (2 * C);
        const double t1 = group_sum(s1[e], m, sm);
        if ((int)threadIdx.x < m.G) row[threadIdx.x * EPC + e] = t1;
        const double t2 = group_sum(s2[e], m, sm);
        if ((int)threadIdx.x < m.G) row[C + threadIdx.x * EPC + e] = t2;
    }
}

// rows per workgroup for `items` rows on at most `cap` workgroups, a multiple of `step`
struct Split { unsigned per; int grid; };
// min_batches > 1 (reductions): a workgroup gets at least that many batches, so that the
// block-level fold and the slab row it writes are amortised over enough streamed rows.
inline Split split_rows(long long items, int step, int cap, int min_batches = 1) {
    long long grid = (items + (long long)step * min_batches - 1) / ((long long)step * min_batches);
    if (grid > cap) grid = cap;
    if (grid < 1) grid = 1;
    long long per = (items + grid - 1) / grid;
    per = (per + step - 1) / step * step;
    grid = (items + per - 1) / per;
    Split s;
    s.per = (unsigned)per;
    s.grid = (int)grid;
    return s;
}

int check_chunked(const char* who, int dtype, int C) {
    const int epc = dtype == SF_BF16 ? 8 : 4;
    if (C <= 0 || C % epc != 0) {
        sf_set_error("%s: channel count %d must be a positive multiple of %d", who, C, epc);
        return SF_EINVAL;
    }
    if (C / epc > BT) {
        sf_set_error("%s: channel count %d too wide", who, C);
        return SF_EINVAL;
    }
    return SF_OK;
}

}  // namespace

#define DISPATCH_T(dtype, CALL)                                       \
    if ((dtype) == SF_BF16) { using T = bf16; CALL; }                 \
    else if ((dtype) == SF_F32) { using T = float; CALL; }            \
    else { sf_set_error("bad dtype %d", (int)(dtype)); return SF_EINVAL; }

static int colsum_launch(const char* who, bool sq, const void* x, int dtype, long long M, int C, double* ws,
                         hipStream_t s) {
    SF_REQUIRE(M < 0x7fffffffLL, "%s: too many rows", who);
    const int epc = dtype == SF_BF16 ? 8 : 4;
    const int rpp = BT / (C / epc);
    const Split sp = split_rows(M, rpp * 8, SF_BN_WS_PARTS, 4);
    if (sq) {
        DISPATCH_T(dtype, hipLaunchKernelGGL((colsum_kernel<T, true>), dim3(sp.grid), dim3(BT), 0, s, (const T*)x,
                                             (unsigned)M, sp.per, C, ws));
    } else {
        DISPATCH_T(dtype, hipLaunchKernelGGL((colsum_kernel<T, false>), dim3(sp.grid), dim3(BT), 0, s, (const T*)x,
                                             (unsigned)M, sp.per, C, ws));
    }
    const int n = (sq ? 2 : 1) * C;
    hipLaunchKernelGGL(ws_fold_kernel, dim3((n + 15) / 16), dim3(256), 0, s, ws, n, sp.grid);
    return SF_OK;
}

extern "C" int sf_bn_stats(const void* x, int dtype, long long M, int C, double* ws, void* stream) {
    SF_REQUIRE(x && ws && M > 0, "sf_bn_stats: bad arguments");
    if (int rc = check_chunked("sf_bn_stats", dtype, C)) return rc;
    if (int rc = colsum_launch("sf_bn_stats", true, x, dtype, M, C, ws, (hipStream_t)stream)) return rc;
    SF_LAUNCH_CHECK("sf_bn_stats");
    return SF_OK;
}

extern "C" int sf_channel_sum(const void* x, int dtype, long long M, int C, int Creal, float* out, double* ws,
                              void* stream) {
    SF_REQUIRE(x && out && ws && M > 0 && Creal <= C, "sf_channel_sum: bad arguments");
    if (int rc = check_chunked("sf_channel_sum", dtype, C)) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (int rc = colsum_launch("sf_channel_sum", false, x, dtype, M, C, ws, s)) return rc;
    hipLaunchKernelGGL(ws_to_float_kernel, dim3(1), dim3(256), 0, s, ws, out, Creal);
    SF_LAUNCH_CHECK("sf_channel_sum");
    return SF_OK;
}

extern "C" int sf_bn_relu_maxpool_fwd(const void* x, int dtype, const float* scale, const float* shift, void* y,
                                      int N, int H, int W, int C, void* stream) {
    SF_REQUIRE(x && y && scale && shift && N > 0 && H >= 2 && W >= 2, "sf_bn_relu_maxpool_fwd: bad arguments");
    if (int rc = check_chunked("sf_bn_relu_maxpool_fwd", dtype, C)) return rc;
    SF_REQUIRE((long long)N * H * W < 0x7fffffffLL, "sf_bn_relu_maxpool_fwd: too many pixels");
    const int epc = dtype == SF_BF16 ? 8 : 4;
    const long long items = (long long)N * (H / 2) * (W / 2);
    const Split sp = split_rows(items, (BT / (C / epc)) * 2, 4096);
    DISPATCH_T(dtype, hipLaunchKernelGGL(bn_relu_maxpool_fwd_kernel<T>, dim3(sp.grid), dim3(BT), 0, (hipStream_t)stream,
                                         (const T*)x, scale, shift, (T*)y, (unsigned)items, sp.per, H, W, C));
    SF_LAUNCH_CHECK("sf_bn_relu_maxpool_fwd");
    return SF_OK;
}

static int bn_act_fwd(const char* who, const void* x, int dtype, const float* scale, const float* shift, void* y,
                      long long M, int C, float slope, void* stream) {
    SF_REQUIRE(x && y && scale && shift && M > 0 && M < 0x7fffffffLL, "%s: bad arguments", who);
    if (int rc = check_chunked(who, dtype, C)) return rc;
    const int epc = dtype == SF_BF16 ? 8 : 4;
    const Split sp = split_rows(M, (BT / (C / epc)) * 4, 4096);
    DISPATCH_T(dtype, hipLaunchKernelGGL(bn_lrelu_fwd_kernel<T>, dim3(sp.grid), dim3(BT), 0, (hipStream_t)stream,
                                         (const T*)x, scale, shift, (T*)y, (unsigned)M, sp.per, C, slope));
    SF_LAUNCH_CHECK(who);
    return SF_OK;
}

extern "C" int sf_bn_lrelu_fwd(const void* x, int dtype, const float* scale, const float* shift, void* y,
                               long long M, int C, void* stream) {
    return bn_act_fwd("sf_bn_lrelu_fwd", x, dtype, scale, shift, y, M, C, 0.2f, stream);
}

extern "C" int sf_bn_relu_fwd(const void* x, int dtype, const float* scale, const float* shift, void* y, long long M,
                              int C, void* stream) {
    return bn_act_fwd("sf_bn_relu_fwd", x, dtype, scale, shift, y, M, C, 0.f, stream);
}

// Largest number of consecutive column bins an aligned group of four pixels touches (in >= out:
// a pixel lies in one or two bins).
static int max_group_bins(int in, int out) {
    int lo[MAX_AXIS], hi[MAX_AXIS];
    for (int p = 0; p < in; ++p) { lo[p] = out; hi[p] = -1; }
    for (int i = 0; i < out; ++i) {
        const int b = (i * in) / out, e = ((i + 1) * in + out - 1) / out;
        for (int p = b; p < e && p < in; ++p) { if (i < lo[p]) lo[p] = i; if (i > hi[p]) hi[p] = i; }
    }
    int worst = 0;
    for (int g = 0; 4 * g < in; ++g) {
        const int last = 4 * g + 3 < in ? 4 * g + 3 : in - 1;
        const int span = hi[last] - lo[4 * g] + 1;
        if (span > worst) worst = span;
    }
    return worst;
}

// Fills the kernel arguments; returns the grid (negative: error) and in *kmode the kernel variant:
// the mode itself, or 4 = avgpool backward by groups of four pixels.
static int fill_bnbwd(BnBwd& a, int mode, int* kmode, const void* x, const void* dy, const float* scale, const float* shift,
                      const float* mean, const float* invstd, const float* gamma, int N, int H, int W, int C,
                      int Creal, int OH, int OW, int training, int dtype, int dy_dtype, int cap) {
    SF_REQUIRE(mode >= 0 && mode <= 3, "bn_bwd: bad mode %d", mode);
    SF_REQUIRE(x && dy && scale && shift && mean && invstd, "bn_bwd: null argument");
    SF_REQUIRE(dy_dtype == dtype, "bn_bwd: dy dtype must equal x dtype");
    SF_REQUIRE(N > 0 && H > 0 && W > 0 && Creal <= C, "bn_bwd: bad shape");
    SF_REQUIRE((long long)N * H * W < 0x7fffffffLL, "bn_bwd: too many pixels");
    if (mode == 1) {
        SF_REQUIRE(OH > 0 && OW > 0, "bn_bwd: avgpool mode needs OH, OW");
        SF_REQUIRE(H <= MAX_AXIS && W <= MAX_AXIS && OH <= MAX_BINS && OW <= MAX_BINS,
                   "bn_bwd: avgpool mode handles maps up to %d x %d and %d x %d bins", MAX_AXIS, MAX_AXIS, MAX_BINS, MAX_BINS);
    }
    if (int rc = check_chunked("bn_bwd", dtype, C)) return rc;
    a.x = x; a.dy = dy; a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd; a.gamma = gamma;
    a.N = N; a.H = H; a.W = W; a.C = C; a.Creal = Creal; a.OH = OH; a.OW = OW; a.training = training;
    a.slope = mode == 3 ? 0.f : 0.2f;
    const int epc = dtype == SF_BF16 ? 8 : 4;
    *kmode = mode == 3 ? 2 : mode;   // modes 2 and 3 differ in a.slope only
    if (mode == 1 && H >= OH && W >= OW && W >= 8 && max_group_bins(W, OW) <= 3) *kmode = 4;
    const long long items = mode == 0    ? (long long)N * ((H + 1) / 2) * ((W + 1) / 2)
                            : *kmode == 4 ? (long long)N * H * ((W + 3) / 4)
                                          : (long long)N * H * W;
    const Split sp = split_rows(items, (BT / (C / epc)) * ((mode == 0 || *kmode == 4) ? 2 : 4), cap, cap <= 512 ? 32 : 1);
    a.items = (unsigned)items;
    a.per = sp.per;
    return sp.grid;
}

extern "C" int sf_bn_bwd_reduce(int mode, const void* x, int dtype, const void* dy, int dy_dtype, const float* scale,
                                const float* shift, const float* mean, const float* invstd, double* ws, int N, int H,
                                int W, int C, int Creal, int OH, int OW, void* stream) {
    BnBwd a;
    int kmode = 0;
    const int grid = fill_bnbwd(a, mode, &kmode, x, dy, scale, shift, mean, invstd, scale, N, H, W, C, Creal, OH, OW, 1, dtype,
                                dy_dtype, SF_BN_WS_PARTS);
    if (grid < 0) return grid;
    SF_REQUIRE(ws, "sf_bn_bwd_reduce: null workspace");
    hipStream_t s = (hipStream_t)stream;
#define RED(MODE) DISPATCH_T(dtype, hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, MODE>), dim3(grid), dim3(BT), 0, s, a, ws))
    if (kmode == 0) { RED(0); } else if (kmode == 1) { RED(1); } else if (kmode == 4) { RED(4); } else { RED(2); }
#undef RED
    hipLaunchKernelGGL(ws_fold_kernel, dim3((2 * C + 15) / 16), dim3(256), 0, s, ws, 2 * C, grid);
    SF_LAUNCH_CHECK("sf_bn_bwd_reduce");
    return SF_OK;
}

static int pooled_shape_ok(const char* who, int dtype, int N, int H, int W, int C, int OH, int OW) {
    SF_REQUIRE(N > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, "%s: bad shape", who);
    SF_REQUIRE(OH <= MAX_BINS && OW <= MAX_BINS, "%s: at most %d x %d bins", who, MAX_BINS, MAX_BINS);
    if (int rc = check_chunked(who, dtype, C)) return rc;
    const long long bh = (H + OH - 1) / OH + 1, bw = (W + OW - 1) / OW + 1;
    SF_REQUIRE(bh * bw <= 65535, "%s: bins of up to %lld x %lld pixels overflow the 16-bit counts", who, bh, bw);
    SF_REQUIRE((long long)N * OH * OW < 0x7fffffffLL, "%s: too many pooled pixels", who);
    return SF_OK;
}

extern "C" int sf_bn_relu_avgpool_fwd_moments(const void* x, int dtype, const float* scale, const float* shift,
                                              const float* mean, const float* invstd, void* y, int out_dtype,
                                              void* pos, void* xh, int N, int H, int W, int C, int Creal, int OH,
                                              int OW, void* stream) {
    SF_REQUIRE(x && y && pos && xh && scale && shift && mean && invstd && Creal <= C,
               "sf_bn_relu_avgpool_fwd_moments: bad arguments");
    if (int rc = pooled_shape_ok("sf_bn_relu_avgpool_fwd_moments", dtype, N, H, W, C, OH, OW)) return rc;
    const size_t lds = sizeof(float) * POOL_CS * (OW + 1);
    SF_REQUIRE(lds <= 65536, "sf_bn_relu_avgpool_fwd_moments: OW=%d > 63", OW);
    dim3 grid((unsigned)(N * OH)), block(POOL_BT);
    hipStream_t s = (hipStream_t)stream;
    bool full = true;   // every bin exactly 4 x 4: the walk needs no predicates
    for (int i = 0; i < OH; ++i) full = full && ((i + 1) * H + OH - 1) / OH - (i * H) / OH == 4;
    for (int i = 0; i < OW; ++i) full = full && ((i + 1) * W + OW - 1) / OW - (i * W) / OW == 4;
#define POOLF(T, TO)                                                                                                    \
    if (full) hipLaunchKernelGGL((bn_relu_avgpool_mom_fwd_kernel<T, TO, true>), grid, block, lds, s, (const T*)x, scale, \
                                 shift, mean, invstd, (TO*)y, (unsigned short*)pos, (T*)xh, H, W, C, Creal, OH, OW);    \
    else hipLaunchKernelGGL((bn_relu_avgpool_mom_fwd_kernel<T, TO, false>), grid, block, lds, s, (const T*)x, scale,    \
                            shift, mean, invstd, (TO*)y, (unsigned short*)pos, (T*)xh, H, W, C, Creal, OH, OW)
    if (dtype == SF_BF16 && out_dtype == SF_BF16) { POOLF(bf16, bf16); }
    else if (dtype == SF_BF16 && out_dtype == SF_F32) { POOLF(bf16, float); }
    else if (dtype == SF_F32 && out_dtype == SF_F32) { POOLF(float, float); }
    else { sf_set_error("sf_bn_relu_avgpool_fwd_moments: unsupported dtype pair %d -> %d", dtype, out_dtype); return SF_EINVAL; }
#undef POOLF
    SF_LAUNCH_CHECK("sf_bn_relu_avgpool_fwd_moments");
    return SF_OK;
}

extern "C" int sf_bn_bwd_reduce_pooled(const void* dy, int dtype, const void* pos, const void* xh, double* ws, int N,
                                       int H, int W, int C, int OH, int OW, void* stream) {
    SF_REQUIRE(dy && pos && xh && ws, "sf_bn_bwd_reduce_pooled: null argument");
    if (int rc = pooled_shape_ok("sf_bn_bwd_reduce_pooled", dtype, N, H, W, C, OH, OW)) return rc;
    const int epc = dtype == SF_BF16 ? 8 : 4;
    const long long M = (long long)N * OH * OW;
    const Split sp = split_rows(M, (BT / (C / epc)) * 4, SF_BN_WS_PARTS, 4);
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_T(dtype, hipLaunchKernelGGL((bn_bwd_reduce_pooled_kernel<T>), dim3(sp.grid), dim3(BT), 0, s, (const T*)dy,
                                         (const unsigned short*)pos, (const T*)xh, (unsigned)M, sp.per, C, H, W, OH, OW, ws));
    hipLaunchKernelGGL(ws_fold_kernel, dim3((2 * C + 15) / 16), dim3(256), 0, s, ws, 2 * C, sp.grid);
    SF_LAUNCH_CHECK("sf_bn_bwd_reduce_pooled");
    return SF_OK;
}

extern "C" int sf_bn_bwd_apply(int mode,const void* x, int dtype, const void* dy, int dy_dtype, const float* scale,
                               const float* shift, const float* mean, const float* invstd, const float* gamma,
                               const double* ws, int training, void* dx, float* dgamma, float* dbeta, int N, int H,
                               int W, int C, int Creal, int OH, int OW, void* stream) {
    BnBwd a;
    int kmode = 0;
    const int grid = fill_bnbwd(a, mode, &kmode, x, dy, scale, shift, mean, invstd, gamma ? gamma : scale, N, H, W, C, Creal, OH,
                                OW, training, dtype, dy_dtype, 4096);
    if (grid < 0) return grid;
    SF_REQUIRE(ws && dx, "sf_bn_bwd_apply: null argument");
    hipStream_t s = (hipStream_t)stream;
#define APP(MODE) DISPATCH_T(dtype, hipLaunchKernelGGL((bn_bwd_apply_kernel<T, MODE>), dim3(grid), dim3(BT), 0, s, a, ws, (T*)dx, dgamma, dbeta))
    if (kmode == 0) { APP(0); } else if (kmode == 1) { APP(1); } else if (kmode == 4) { APP(4); } else { APP(2); }
#undef APP
    SF_LAUNCH_CHECK("sf_bn_bwd_apply");
    return SF_OK;
}
