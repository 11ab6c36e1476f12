// Convolutions whose shapes do not fill MFMA tiles: the PatchGAN's first layer (one input
// channel) and single-output-channel convs (PatchGAN head, decoder 1x1).  They are HBM / L2
// bound; lanes run over 16-byte channel chunks and reduce with wave shuffles.
#include "sf_common.h"
#include "../../include/sfhip.h"

namespace {

constexpr int NT = 256;

template <typename T> __device__ __forceinline__ void load_chunk(const T* p, float* f);
template <> __device__ __forceinline__ void load_chunk<float>(const float* p, float* f) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
}
template <> __device__ __forceinline__ void load_chunk<bf16>(const bf16* p, float* f) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
}
template <typename T> __device__ __forceinline__ void store_chunk(T* p, const float* f);
template <> __device__ __forceinline__ void store_chunk<float>(float* p, const float* f) {
    *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]);
}
template <> __device__ __forceinline__ void store_chunk<bf16>(bf16* p, const float* f) {
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (bf16)f[i];
    *reinterpret_cast<bf16x8*>(p) = v;
}

inline int grid_for(long long items, int cap = 8192) {
    long long b = (items + NT - 1) / NT;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

// Sum `vals[nv]` over the row phases of each channel group (thread = phase*G + cg) and add the
// totals atomically: out_index(cg, v) gives the destination of value v of group cg (or -1).
template <int NV, typename F>
__device__ __forceinline__ void reduce_phases_atomic(const float* vals, int G, int rpp, float* sm /*[8][NT]*/,
                                                     float* out, F out_index) {
    const int cg = threadIdx.x % G, phase = threadIdx.x / G;
#pragma unroll
    for (int v0 = 0; v0 < NV; v0 += 8) {
        __syncthreads();
#pragma unroll
        for (int v = 0; v < 8; ++v)
            if (v0 + v < NV) sm[v * NT + threadIdx.x] = vals[v0 + v];
        __syncthreads();
        if (phase == 0) {
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                if (v0 + v < NV) {
                    float t = 0.f;
                    for (int ph = 0; ph < rpp; ++ph) t += sm[v * NT + ph * G + cg];
                    const long long idx = out_index(cg, v0 + v);
                    if (idx >= 0) unsafeAtomicAdd(out + idx, t);
                }
            }
        }
    }
}

// ------------------------------------------------------------------ first PatchGAN layer
template <typename T>
__global__ void dconv1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                  const float* __restrict__ b, T* __restrict__ y, int N, int H, int W, int K,
                                  int OH, int OW) {
    constexpr int EPC = ElemTraits<T>::EPC;
    extern __shared__ float sw[];  // [16][K] + [K]
    for (int i = threadIdx.x; i < 16 * K; i += NT) {
        const int t = i / K, k = i % K;
        sw[i] = w[k * 16 + t];
    }
    for (int i = threadIdx.x; i < K; i += NT) sw[16 * K + i] = b[i];
    __syncthreads();
    const int G = K / EPC;
    const long long total = (long long)N * OH * OW * G;
    for (long long o = blockIdx.x * (long long)NT + threadIdx.x; o < total; o += (long long)gridDim.x * NT) {
        const int cg = (int)(o % G);
        long long r = o / G;
        const int ow = (int)(r % OW); r /= OW;
        const int oh = (int)(r % OH);
        const long long n = r / OH;
        float acc[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] = sw[16 * K + cg * EPC + e];
        // all 16 taps are loaded first (clamped address, masked value) so the loads pipeline
        float xv[16];
        const float* xin = x + n * H * (long long)W;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int iy = oh * 2 - 1 + (t >> 2), ix = ow * 2 - 1 + (t & 3);
            const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
            const int cy = min(max(iy, 0), H - 1), cx = min(max(ix, 0), W - 1);
            const float v = xin[cy * W + cx];
            xv[t] = ok ? v : 0.f;
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[e] += xv[t] * sw[t * K + cg * EPC + e];
        }
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] = acc[e] > 0.f ? acc[e] : 0.2f * acc[e];
        store_chunk<T>(y + o * EPC, acc);
    }
}

// one input pixel per group of G lanes (G = K / chunk, a power of two <= 64): each lane handles
// one 16-byte channel chunk of the (at most four) contributing output pixels, then the group
// reduces with shuffles.
template <typename T>
__global__ void dconv1_bwd_data_kernel(const T* __restrict__ dz, const float* __restrict__ w,
                                       float* __restrict__ dx, int N, int H, int W, int K, int OH, int OW) {
    constexpr int EPC = ElemTraits<T>::EPC;
    extern __shared__ float sw[];  // [16][K]
    for (int i = threadIdx.x; i < 16 * K; i += NT) {
        const int t = i / K, k = i % K;
        sw[i] = w[k * 16 + t];
    }
    __syncthreads();
    const int G = K / EPC;
    const int cg = threadIdx.x % G;
    const long long gid = (blockIdx.x * (long long)NT + threadIdx.x) / G;
    const long long ngroups = ((long long)gridDim.x * NT) / G;
    const long long total = (long long)N * H * W;
    const long long rounds = (total + ngroups - 1) / ngroups;
    for (long long it = 0; it < rounds; ++it) {
        const long long o = gid + it * ngroups;
        const bool live = o < total;
        float acc = 0.f;
        if (live) {
            const int ix = (int)(o % W);
            const long long r = o / W;
            const int iy = (int)(r % H);
            const long long n = r / H;
            // rows r with (iy + 1 - r) even: r = (iy+1)&1 and r + 2; same for columns
            const int r0 = (iy + 1) & 1, s0 = (ix + 1) & 1;
            float f[4][EPC];
            bool ok[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rr = r0 + 2 * (q >> 1), ss = s0 + 2 * (q & 1);
                const int oh = (iy + 1 - rr) >> 1, ow = (ix + 1 - ss) >> 1;
                ok[q] = (iy + 1 - rr) >= 0 && oh < OH && (ix + 1 - ss) >= 0 && ow < OW;
                const int ch = min(max(oh, 0), OH - 1), cw = min(max(ow, 0), OW - 1);
                load_chunk<T>(dz + ((n * OH + ch) * OW + cw) * K + cg * EPC, f[q]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rr = r0 + 2 * (q >> 1), ss = s0 + 2 * (q & 1);
                const float* wt = sw + (rr * 4 + ss) * K + cg * EPC;
                float a = 0.f;
#pragma unroll
                for (int e = 0; e < EPC; ++e) a += f[q][e] * wt[e];
                acc += ok[q] ? a : 0.f;
            }
        }
        for (int off = G >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (live && cg == 0) dx[o] = acc;
    }
}

// dw[k][t] = sum_m dz[m][k] * x[pixel(m) + tap t], db[k] = sum_m dz[m][k].  A thread owns one
// 16-byte chunk of output channels and one half of the filter rows (8 taps) for the whole
// kernel and walks the rows of the workgroup's contiguous range in batches of U (all loads of
// a batch are issued before its arithmetic, clamped addresses, masked values).
template <typename T>
__global__ __launch_bounds__(NT) void dconv1_bwd_weight_kernel(const float* __restrict__ x, const T* __restrict__ dz,
                                                               float* __restrict__ dw, float* __restrict__ db,
                                                               unsigned M, unsigned per, int H, int W, int K, int OH,
                                                               int OW) {
    constexpr int EPC = ElemTraits<T>::EPC;
    constexpr int U = 4;
    constexpr int NV = 9 * EPC;  // 8 taps + bias
    __shared__ float sm[8 * NT];
    const int G = K / EPC, G2 = G * 2, rpp = NT / G2;
    const int cg = threadIdx.x % G, th = (threadIdx.x / G) & 1, phase = threadIdx.x / G2;
    float acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.f;
    const unsigned r0 = blockIdx.x * per;
    const unsigned r1 = r0 + per < M ? r0 + per : M;
    if (phase < rpp) {
        for (unsigned r = r0 + phase; r < r1; r += rpp * U) {
            float f[U][EPC], xv[U][8];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned rr = r + u * rpp;
                const bool live = rr < r1;
                const unsigned m = live ? rr : r;
                const unsigned q = m / (unsigned)OW, ow = m - q * OW;
                const unsigned n = q / (unsigned)OH, oh = q - n * OH;
                load_chunk<T>(dz + (size_t)m * K + cg * EPC, f[u]);
                if (!live) {
#pragma unroll
                    for (int e = 0; e < EPC; ++e) f[u][e] = 0.f;
                }
                const float* xin = x + (size_t)n * H * W;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int iy = (int)oh * 2 - 1 + 2 * th + (t >> 2), ix = (int)ow * 2 - 1 + (t & 3);
                    const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
                    const int cy = min(max(iy, 0), H - 1), cx = min(max(ix, 0), W - 1);
                    const float v = xin[cy * W + cx];
                    xv[u][t] = ok ? v : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int e = 0; e < EPC; ++e) acc[t * EPC + e] += xv[u][t] * f[u][e];
#pragma unroll
                for (int e = 0; e < EPC; ++e) acc[8 * EPC + e] += f[u][e];
            }
        }
    }
    // thread = phase * G2 + (th * G + cg): reduce over phases with "groups" of G2
    reduce_phases_atomic<NV>(acc, G2, rpp, sm, dw, [&](int g, int v) -> long long {
        const int t = v / EPC, e = v % EPC;
        const int gcg = g % G, gth = g / G;
        if (t < 8) return (long long)(gcg * EPC + e) * 16 + gth * 8 + t;
        return -1;
    });
    reduce_phases_atomic<EPC>(acc + 8 * EPC, G2, rpp, sm, db, [&](int g, int v) -> long long {
        return g < G ? (long long)g * EPC + v : -1;
    });
}

// ---- MFMA forms of the first layer for bf16 tensors with K == 64 (the 16 taps are exactly the
// reduction depth of one 32x32x16 MFMA; x is rounded to bf16 like a conv under autocast).
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

// D[channel][pixel] = W[channel][tap] * patch[tap][pixel]: a wave owns the weights (two fragments)
// and the bias for the whole kernel and walks 32-pixel groups; lane (p, h) gathers filter rows
// 2h, 2h+1 of pixel p (8 clamped, masked loads) and ends up with channels 8g+4h+e of that pixel,
// which one permlane32 swap per dword pair regroups into 16-byte stores.
__global__ __launch_bounds__(NT) void dconv1_fwd_mfma_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ w,
                                                             const float* __restrict__ b, bf16* __restrict__ y,
                                                             unsigned M, int H, int W, int OH, int OW) {
    constexpr int U = 2;
    const int lane = threadIdx.x & 63, p = lane & 31, h = lane >> 5;
    const unsigned wave = blockIdx.x * (NT / 64) + (threadIdx.x >> 6), nwaves = gridDim.x * (NT / 64);
    bf16x8 wa[2];
    float bias[2][16];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = w[(a * 32 + p) * 16 + 8 * h + j];
        wa[a] = to_bf16x8(f);
#pragma unroll
        for (int i = 0; i < 16; ++i) bias[a][i] = b[a * 32 + 8 * (i >> 2) + 4 * h + (i & 3)];
    }
    const unsigned ngroups = (M + 31) >> 5;
    for (unsigned g = wave * U; g < ngroups; g += nwaves * U) {
        float xv[U][8];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            unsigned m = (g + u) * 32 + p;
            m = m < M ? m : M - 1;
            const unsigned q = m / (unsigned)OW, ow = m - q * OW;
            const unsigned n = q / (unsigned)OH, oh = q - n * OH;
            const float* xin = x + (size_t)n * H * W;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int iy = (int)oh * 2 - 1 + 2 * h + (j >> 2), ix = (int)ow * 2 - 1 + (j & 3);
                const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
                const int cy = min(max(iy, 0), H - 1), cx = min(max(ix, 0), W - 1);
                const float v = xin[cy * W + cx];
                xv[u][j] = ok ? v : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bf16x8 fb = to_bf16x8(xv[u]);
            const unsigned m = (g + u) * 32 + p;
            const bool live = m < M;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                f32x16 acc;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = bias[a][i];
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[a], fb, acc, 0, 0, 0);
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float t = acc[8 * gp + e];
                        v[e] = t > 0.f ? t : 0.2f * t;
                    }
                    const unsigned A0 = pack_bf16x2(v[0], v[1]), A1 = pack_bf16x2(v[2], v[3]);
                    const unsigned B0 = pack_bf16x2(v[4], v[5]), B1 = pack_bf16x2(v[6], v[7]);
                    const auto s0 = __builtin_amdgcn_permlane32_swap(A0, B0, false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(A1, B1, false, false);
                    if (live)
                        *reinterpret_cast<uint4*>(y + (size_t)m * 64 + a * 32 + 8 * (2 * gp + h)) =
                            make_uint4(s0[0], s1[0], s0[1], s1[1]);
                }
            }
        }
    }
}

// [dw | db][tap or bias row][channel] = A[row][pixel] * dz[pixel][channel] with the pixels as the
// reduction: rows 0..15 of A are the taps of the pixel's patch, row 16 is all ones (bias
// gradient), the rest zero.  Lane (r, h) covers the run of 8 consecutive pixels 8h..8h+7 of a
// 16-pixel group: 8 x loads for tap r, and 8 dword loads of channels (2r, 2r+1) whose low / high
// halves feed two MFMAs (columns = even / odd channels).  CROSS = false: OW % 8 == 0, a run stays
// inside one output row.  CROSS = true (OW >= 8): a run may cross into the next output row or
// image (at most once) and the last run may end before its eighth pixel; a pixel past the end
// reads pixel M - 1 against a zero column of A.
// Results: with `slab` every workgroup stores its 17 x 64 sums to row blockIdx.x of it and
// dconv1_wfold_kernel adds the rows in order (the 256 x 1088 atomics of the other form all land
// on 70 cache lines at the same moment and took longer than the pass itself); without, atomics
// into the zeroed dw / db.
constexpr int WG_NT = 512;
template <bool CROSS>
__global__ __launch_bounds__(WG_NT) void dconv1_bwd_weight_mfma_kernel(const float* __restrict__ x,
                                                                    const bf16* __restrict__ dz,
                                                                    float* __restrict__ dw, float* __restrict__ db,
                                                                    float* __restrict__ slab,
                                                                    unsigned M, int H, int W, int OH, int OW) {
    constexpr int U = 2;
    __shared__ float sm[WG_NT / 64][17][64];
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5, wv = threadIdx.x >> 6;
    const unsigned wave = blockIdx.x * (WG_NT / 64) + wv, nwaves = gridDim.x * (WG_NT / 64);
    const int t = r & 15, ty = t >> 2, tx = t & 3;
    const unsigned nruns = (M + 7) >> 3, ngroups = (nruns + 1) >> 1;
    const int nimg = (int)(M / ((unsigned)OH * OW));
    f32x16 acc[2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    for (unsigned g = wave * U; g < ngroups; g += nwaves * U) {
        float xv[U][8];
        unsigned d[U][8];
        unsigned xok[U] = {};      // bit j: the tap of pixel j lies inside the image; bit 8 + j: pixel j exists
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned run = (g + u) * 2 + h;
            const bool rlive = run < nruns;
            const unsigned m0 = rlive ? run * 8 : 0;
            const unsigned q = m0 / (unsigned)OW, ow0 = m0 - q * OW;
            const unsigned n0 = q / (unsigned)OH, oh0 = q - n0 * OH;
            if (!CROSS) {
                const int iy = (int)oh0 * 2 - 1 + ty;
                const bool rok = rlive && iy >= 0 && iy < H;
                const float* xrow = x + ((size_t)n0 * H + min(max(iy, 0), H - 1)) * W;
                const bf16* dzp = dz + (size_t)m0 * 64 + 2 * r;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int ix = ((int)ow0 + j) * 2 - 1 + tx;
                    xv[u][j] = xrow[min(max(ix, 0), W - 1)];
                    xok[u] |= (unsigned)(rok && ix >= 0 && ix < W) << j;
                    d[u][j] = *reinterpret_cast<const unsigned*>(dzp + j * 64);
                }
                xok[u] |= rlive ? 0xff00u : 0u;
            } else {
                // the row the run starts in and the one it may wrap to
                const int ohw = oh0 + 1 < (unsigned)OH ? (int)oh0 + 1 : 0;
                const int nw = oh0 + 1 < (unsigned)OH ? (int)n0 : min((int)n0 + 1, nimg - 1);
                const int iy0 = (int)oh0 * 2 - 1 + ty, iy1 = ohw * 2 - 1 + ty;
                const float* row0 = x + ((size_t)n0 * H + min(max(iy0, 0), H - 1)) * W;
                const float* row1 = x + ((size_t)nw * H + min(max(iy1, 0), H - 1)) * W;
                const bool ok0 = iy0 >= 0 && iy0 < H, ok1 = iy1 >= 0 && iy1 < H;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned m = m0 + j;
                    const bool live = rlive && m < M;
                    const bool wrap = (int)ow0 + j >= OW;
                    const int ix = ((int)ow0 + j - (wrap ? OW : 0)) * 2 - 1 + tx;
                    xv[u][j] = (wrap ? row1 : row0)[min(max(ix, 0), W - 1)];
                    xok[u] |= (unsigned)(live && (wrap ? ok1 : ok0) && ix >= 0 && ix < W) << j;
                    xok[u] |= (unsigned)live << (8 + j);
                    d[u][j] = *reinterpret_cast<const unsigned*>(dz + (size_t)(m < M ? m : M - 1) * 64 + 2 * r);
                }
            }
        }
        // the loaded values pass through an opaque statement before they are masked, so that the
        // compiler cannot move a load under its mask's branch (which would serialise the loads)
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                asm volatile("" : "+v"(xv[u][j]));
                xv[u][j] = (xok[u] >> j & 1u) ? xv[u][j] : 0.f;
            }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float av[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) av[j] = !(xok[u] >> (8 + j) & 1u) ? 0.f : (r < 16 ? xv[u][j] : (r == 16 ? 1.f : 0.f));
            const bf16x8 fa = to_bf16x8(av);
            unsigned lo[4], hi[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                lo[i] = (d[u][2 * i] & 0xffffu) | (d[u][2 * i + 1] << 16);
                hi[i] = (d[u][2 * i] >> 16) | (d[u][2 * i + 1] & 0xffff0000u);
            }
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, *reinterpret_cast<const bf16x8*>(lo), acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, *reinterpret_cast<const bf16x8*>(hi), acc[1], 0, 0, 0);
        }
    }
    // acc[a][i]: row 8*(i>>2) + 4h + (i&3), channel 2r + a
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = 8 * (i >> 2) + 4 * h + (i & 3);
            if (row < 17) sm[wv][row][2 * r + a] = acc[a][i];
        }
    __syncthreads();
    for (int i = threadIdx.x; i < 17 * 64; i += WG_NT) {
        const int row = i >> 6, ch = i & 63;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < WG_NT / 64; ++k) s += sm[k][row][ch];
        if (slab) slab[(size_t)blockIdx.x * (17 * 64) + i] = s;
        else unsafeAtomicAdd(row < 16 ? dw + ch * 16 + row : db + ch, s);
    }
}
// dw / db = the slab rows added in order
__global__ __launch_bounds__(64) void dconv1_wfold_kernel(const float* __restrict__ slab, int rows, float* __restrict__ dw,
                                                          float* __restrict__ db) {
    const int i = blockIdx.x * 64 + threadIdx.x;      // 17 blocks: i = row * 64 + ch
    const int row = i >> 6, ch = i & 63;
    float s = 0.f;
    for (int k0 = 0; k0 < rows; k0 += 8) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = slab[(size_t)min(k0 + k, rows - 1) * (17 * 64) + i];
#pragma unroll
        for (int k = 0; k < 8; ++k) s += k0 + k < rows ? v[k] : 0.f;
    }
    if (row < 16) dw[ch * 16 + row] = s;
    else db[ch] = s;
}

// C[tap][pixel] = W^T[tap][channel] * dz[pixel][channel] with two 16x16x32 MFMAs (lane (col, q)
// ends up with the four taps of filter row q of output pixel col), then col2im in registers:
// input column 2ow-1 / 2ow sums taps tx = 0 / 1 of pixel ow and tx = 2 / 3 of pixel ow-1 (lane
// col-1 of the same 16-lane row), and input row 2oh-1 / 2oh sums filter rows 0 / 1 of output row
// oh and 2 / 3 of row oh-1 (lanes +32 of the previous iteration).  A wave marches down RB output
// rows of a 16-column strip (15 new columns, one of overlap; one row of overlap on top); output
// rows / columns OH / OW are virtual (zero dz) and close the last input row / column.
__global__ __launch_bounds__(NT) void dconv1_bwd_data_mfma_kernel(const bf16* __restrict__ dz,
                                                                  const float* __restrict__ w,
                                                                  float* __restrict__ dx, int N, int H, int W, int OH,
                                                                  int OW, int nstrips, int nchunks, int RB) {
    constexpr int U = 4;
    const int lane = threadIdx.x & 63, col = lane & 15, q = lane >> 4;
    const unsigned wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (NT / 64) + (threadIdx.x >> 6));
    if (wave >= (unsigned)N * nstrips * nchunks) return;
    const int strip = wave % nstrips;
    const unsigned rest = wave / nstrips;
    const int chunk = rest % nchunks, n = rest / nchunks;
    bf16x8 wa[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = w[(32 * s + 8 * q + j) * 16 + col];
        wa[s] = to_bf16x8(f);
    }
    const int ow = strip * 15 - 1 + col;
    const bool cok = ow >= 0 && ow < OW;
    const int ra = chunk * RB, rb = min(ra + RB, OH + 1);
    const bf16* base = dz + ((size_t)n * OH * OW + min(max(ow, 0), OW - 1)) * 64 + 8 * q;
    float prev0 = 0.f, prev1 = 0.f;
    for (int oh0 = ra - 1; oh0 < rb; oh0 += U) {
        uint4 d[U][2];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int oh = oh0 + u;
            const bool ok = cok && oh >= 0 && oh < OH;
            const bf16* p = base + (size_t)min(max(oh, 0), OH - 1) * OW * 64;
            const uint4 v0 = *reinterpret_cast<const uint4*>(p), v1 = *reinterpret_cast<const uint4*>(p + 32);
            d[u][0] = ok ? v0 : make_uint4(0, 0, 0, 0);
            d[u][1] = ok ? v1 : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int oh = oh0 + u;
            if (oh >= rb) break;
            f32x4 c = {0.f, 0.f, 0.f, 0.f};
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[0], *reinterpret_cast<const bf16x8*>(&d[u][0]), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[1], *reinterpret_cast<const bf16x8*>(&d[u][1]), c, 0, 0, 0);
            const float h0 = c[0] + __shfl_up(c[2], 1, 16), h1 = c[1] + __shfl_up(c[3], 1, 16);
            const unsigned p0 = __float_as_uint(prev0), p1 = __float_as_uint(prev1);
            const float pv0 = __uint_as_float(__builtin_amdgcn_permlane32_swap(p0, p0, false, false)[1]);
            const float pv1 = __uint_as_float(__builtin_amdgcn_permlane32_swap(p1, p1, false, false)[1]);
            const int iy = 2 * oh - 1 + q, ix = 2 * ow - 1;
            if (oh >= ra && col >= 1 && q < 2 && iy >= 0 && iy < H) {
                float* o = dx + ((size_t)n * H + iy) * W;
                if (ix >= 0 && ix < W) o[ix] = h0 + pv0;
                if (ix + 1 < W) o[ix + 1] = h1 + pv1;
            }
            prev0 = h0;
            prev1 = h1;
        }
    }
}

// ------------------------------------------------------------------ single-output-channel conv
// a group of LPP lanes handles one output pixel; lanes run over channel chunks.
template <typename T>
__global__ void conv_k1_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                   float* __restrict__ y, int N, int H, int W, int C, int Creal, int R, int pad,
                                   int OH, int OW, int LPP) {
    constexpr int EPC = ElemTraits<T>::EPC;
    const int G = C / EPC, RS = R * R;
    const int lane_in = threadIdx.x % LPP;
    const long long gid = (blockIdx.x * (long long)NT + threadIdx.x) / LPP;
    const long long ngroups = ((long long)gridDim.x * NT) / LPP;
    const long long total = (long long)N * OH * OW;
    const long long rounds = (total + ngroups - 1) / ngroups;
    for (long long it = 0; it < rounds; ++it) {
        const long long o = gid + it * ngroups;
        const bool ok = o < total;
        float acc = 0.f;
        if (ok) {
            const int ow = (int)(o % OW);
            const long long r = o / OW;
            const int oh = (int)(r % OH);
            const long long n = r / OH;
            for (int t = 0; t < RS; ++t) {
                const int iy = oh - pad + t / R, ix = ow - pad + t % R;
                if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                const T* src = x + ((n * H + iy) * W + ix) * C;
                for (int cg = lane_in; cg < G; cg += LPP) {
                    float f[EPC];
                    load_chunk<T>(src + cg * EPC, f);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        const int c = cg * EPC + e;
                        if (c < Creal) acc += f[e] * w[c * RS + t];
                    }
                }
            }
        }
        for (int off = LPP >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (ok && lane_in == 0) y[o] = acc + b[0];
    }
}

template <typename T>
__global__ void conv_k1_bwd_data_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                        const T* __restrict__ aux, T* __restrict__ dx, int N, int H, int W, int C,
                                        int Creal, int R, int pad, int OH, int OW) {
    constexpr int EPC = ElemTraits<T>::EPC;
    const int G = C / EPC, RS = R * R;
    const long long total = (long long)N * H * W * G;
    for (long long o = blockIdx.x * (long long)NT + threadIdx.x; o < total; o += (long long)gridDim.x * NT) {
        const int cg = (int)(o % G);
        long long r = o / G;
        const int ix = (int)(r % W); r /= W;
        const int iy = (int)(r % H);
        const long long n = r / H;
        float acc[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
        for (int t = 0; t < RS; ++t) {
            const int oh = iy + pad - t / R, ow = ix + pad - t % R;
            if (oh < 0 || oh >= OH || ow < 0 || ow >= OW) continue;
            const float g = dy[(n * OH + oh) * OW + ow];
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const int c = cg * EPC + e;
                if (c < Creal) acc[e] += g * w[c * RS + t];
            }
        }
        if (aux) {
            float a[EPC];
            load_chunk<T>(aux + o * EPC, a);
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[e] = a[e] > 0.f ? acc[e] : 0.f;
        }
        store_chunk<T>(dx + o * EPC, acc);
    }
}

template <typename T, int RS>
__global__ void conv_k1_bwd_weight_kernel(const T* __restrict__ x, const float* __restrict__ dy,
                                          float* __restrict__ dw, float* __restrict__ db, int N, int H, int W, int C,
                                          int Creal, int R, int pad, int OH, int OW) {
    constexpr int EPC = ElemTraits<T>::EPC;
    constexpr int NV = RS * EPC;
    __shared__ float sm[8 * NT];
    const int G = C / EPC, rpp = NT / G;
    const int cg = threadIdx.x % G, phase = threadIdx.x / G;
    float acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.f;
    float bsum = 0.f;
    const long long M = (long long)N * OH * OW;
    if (phase < rpp) {
        for (long long m = (long long)blockIdx.x * rpp + phase; m < M; m += (long long)gridDim.x * rpp) {
            const int ow = (int)(m % OW);
            const long long r = m / OW;
            const int oh = (int)(r % OH);
            const long long n = r / OH;
            const float g = dy[m];
            if (cg == 0) bsum += g;
#pragma unroll
            for (int t = 0; t < RS; ++t) {
                const int iy = oh - pad + t / R, ix = ow - pad + t % R;
                if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                float f[EPC];
                load_chunk<T>(x + ((n * H + iy) * W + ix) * C + cg * EPC, f);
#pragma unroll
                for (int e = 0; e < EPC; ++e) acc[t * EPC + e] += g * f[e];
            }
        }
    }
    reduce_phases_atomic<NV>(acc, G, rpp, sm, dw, [&](int g, int v) -> long long {
        const int t = v / EPC, e = v % EPC;
        const int c = g * EPC + e;
        return c < Creal ? (long long)c * RS + t : -1;
    });
    reduce_phases_atomic<1>(&bsum, G, rpp, sm, db, [&](int g, int v) -> long long { return g == 0 ? 0 : -1; });
}

// ------------------------------------------------------------------ 1x1 conv to one channel
// Streaming versions for R = 1, pad = 0 (the decoder's output layer): a thread owns one 16-byte
// channel chunk for the whole kernel, a row (pixel) is spread over G = C / chunk lanes, rows are
// taken in batches of U with all loads issued first.  32-bit row indices.
__device__ __attribute__((aligned(16))) uint4 g_zero16s;  // never written: stays zero

template <typename T> __device__ __forceinline__ void decode16(const uint4& r, float* f);
template <> __device__ __forceinline__ void decode16<float>(const uint4& r, float* f) {
    f[0] = __uint_as_float(r.x); f[1] = __uint_as_float(r.y); f[2] = __uint_as_float(r.z); f[3] = __uint_as_float(r.w);
}
template <> __device__ __forceinline__ void decode16<bf16>(const uint4& r, float* f) {
    f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
    f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
    f[4] = __uint_as_float(r.z << 16); f[5] = __uint_as_float(r.z & 0xffff0000u);
    f[6] = __uint_as_float(r.w << 16); f[7] = __uint_as_float(r.w & 0xffff0000u);
}

template <typename T>
__global__ __launch_bounds__(NT) void pw1_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ b, float* __restrict__ y, unsigned M,
                                                     unsigned per, int C, int Creal) {
    constexpr int EPC = ElemTraits<T>::EPC, U = 8;
    const int G = C / EPC, rpp = NT / G;
    const int cg = threadIdx.x % G, phase = threadIdx.x / G;
    float wv[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) wv[e] = cg * EPC + e < Creal ? w[cg * EPC + e] : 0.f;
    const float bias = b[0];
    const unsigned r0 = blockIdx.x * per;
    const unsigned r1 = r0 + per < M ? r0 + per : M;
    for (unsigned r = r0 + phase; r < r1 + phase; r += rpp * U) {   // uniform trip count per group
        uint4 raw[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned rr = r + u * rpp;
            raw[u] = *reinterpret_cast<const uint4*>(rr < r1 ? (const void*)(x + (size_t)rr * C + cg * EPC) : (const void*)&g_zero16s);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float f[EPC];
            decode16<T>(raw[u], f);
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < EPC; ++e) a += f[e] * wv[e];
            for (int off = G >> 1; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
            const unsigned rr = r + u * rpp;
            if (cg == 0 && rr < r1) y[rr] = a + bias;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(NT) void pw1_bwd_data_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                          const T* __restrict__ aux, T* __restrict__ dx, unsigned M,
                                                          unsigned per, int C, int Creal) {
    constexpr int EPC = ElemTraits<T>::EPC, U = 8;
    const int G = C / EPC, rpp = NT / G;
    const int cg = threadIdx.x % G, phase = threadIdx.x / G;
    float wv[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) wv[e] = cg * EPC + e < Creal ? w[cg * EPC + e] : 0.f;
    const unsigned r0 = blockIdx.x * per;
    const unsigned r1 = r0 + per < M ? r0 + per : M;
    for (unsigned r = r0 + phase; r < r1; r += rpp * U) {
        uint4 raw[U];
        float g[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned rr = r + u * rpp;
            const bool ok = rr < r1;
            g[u] = dy[ok ? rr : r];
            if (aux) raw[u] = *reinterpret_cast<const uint4*>(ok ? (const void*)(aux + (size_t)rr * C + cg * EPC) : (const void*)&g_zero16s);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned rr = r + u * rpp;
            if (rr >= r1) break;
            float o[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) o[e] = g[u] * wv[e];
            if (aux) {
                float a[EPC];
                decode16<T>(raw[u], a);
#pragma unroll
                for (int e = 0; e < EPC; ++e) o[e] = a[e] > 0.f ? o[e] : 0.f;
            }
            store_chunk<T>(dx + (size_t)rr * C + cg * EPC, o);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(NT) void pw1_bwd_weight_kernel(const T* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ dw, float* __restrict__ db, unsigned M,
                                                            unsigned per, int C, int Creal) {
    constexpr int EPC = ElemTraits<T>::EPC, U = 8;
    __shared__ float sm[8 * NT];
    const int G = C / EPC, rpp = NT / G;
    const int cg = threadIdx.x % G, phase = threadIdx.x / G;
    float acc[EPC + 1];
#pragma unroll
    for (int e = 0; e <= EPC; ++e) acc[e] = 0.f;
    const unsigned r0 = blockIdx.x * per;
    const unsigned r1 = r0 + per < M ? r0 + per : M;
    for (unsigned r = r0 + phase; r < r1; r += rpp * U) {
        uint4 raw[U];
        float g[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned rr = r + u * rpp;
            const bool ok = rr < r1;
            const float gv = dy[ok ? rr : r];
            g[u] = ok ? gv : 0.f;
            raw[u] = *reinterpret_cast<const uint4*>(ok ? (const void*)(x + (size_t)rr * C + cg * EPC) : (const void*)&g_zero16s);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float f[EPC];
            decode16<T>(raw[u], f);
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[e] += g[u] * f[e];
            acc[EPC] += g[u];
        }
    }
    reduce_phases_atomic<EPC>(acc, G, rpp, sm, dw, [&](int gq, int v) -> long long {
        const int c = gq * EPC + v;
        return c < Creal ? (long long)c : -1;
    });
    reduce_phases_atomic<1>(acc + EPC, G, rpp, sm, db, [&](int gq, int v) -> long long { return gq == 0 ? 0 : -1; });
}

// Everything the backward of ReLU -> Conv2d(C,1,1) needs, in one pass over x (the ReLU output)
// and dy: dx[p][c] = dy[p] * w[c] * (x[p][c] > 0), dw[c] = sum_p dy[p] * x[p][c], db = sum_p dy[p]
// and dsum[c] = sum_p dx[p][c] (of the values as stored, i.e. rounded to T): the bias gradient
// of the conv that produced x.
template <typename T>
__global__ __launch_bounds__(NT) void pw1_bwd_fused_kernel(const T* __restrict__ x, const float* __restrict__ dy,
                                                           const float* __restrict__ w, T* __restrict__ dx,
                                                           float* __restrict__ dw, float* __restrict__ db,
                                                           float* __restrict__ dsum, unsigned M, unsigned per, int C,
                                                           int Creal) {
    constexpr int EPC = ElemTraits<T>::EPC, U = 8;
    __shared__ float sm[8 * NT];
    const int G = C / EPC, rpp = NT / G;
    const int cg = threadIdx.x % G, phase = threadIdx.x / G;
    float wv[EPC], accw[EPC + 1], accs[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        wv[e] = cg * EPC + e < Creal ? w[cg * EPC + e] : 0.f;
        accw[e] = accs[e] = 0.f;
    }
    accw[EPC] = 0.f;
    const unsigned r0 = blockIdx.x * per;
    const unsigned r1 = r0 + per < M ? r0 + per : M;
    for (unsigned r = r0 + phase; r < r1; r += rpp * U) {
        uint4 raw[U];
        float g[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned rr = r + u * rpp;
            const bool ok = rr < r1;
            const float gv = dy[ok ? rr : r];
            g[u] = ok ? gv : 0.f;
            raw[u] = *reinterpret_cast<const uint4*>(ok ? (const void*)(x + (size_t)rr * C + cg * EPC) : (const void*)&g_zero16s);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned rr = r + u * rpp;
            float f[EPC], o[EPC];
            decode16<T>(raw[u], f);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                accw[e] += g[u] * f[e];
                o[e] = from_f32<T>(f[e] > 0.f ? g[u] * wv[e] : 0.f);
                accs[e] += to_f32(from_f32<T>(o[e]));
            }
            accw[EPC] += g[u];
            if (rr < r1) store_chunk<T>(dx + (size_t)rr * C + cg * EPC, o);
        }
    }
    reduce_phases_atomic<EPC>(accw, G, rpp, sm, dw, [&](int gq, int v) -> long long {
        const int c = gq * EPC + v;
        return c < Creal ? (long long)c : -1;
    });
    reduce_phases_atomic<1>(accw + EPC, G, rpp, sm, db, [&](int gq, int v) -> long long { return gq == 0 ? 0 : -1; });
    reduce_phases_atomic<EPC>(accs, G, rpp, sm, dsum, [&](int gq, int v) -> long long {
        const int c = gq * EPC + v;
        return c < Creal ? (long long)c : -1;
    });
}

// ------------------------------------------------------------------ RxR conv to one channel
// (PatchGAN head, R = 4, stride 1).  Weights sit in LDS as [tap][channel]; loads of a work item
// are issued together with clamped / zero-chunk addresses; 32-bit indices.
constexpr int HEAD_MAX_W = 16 * 512;  // taps x channels the LDS weight copy holds
#ifndef SF_HEAD_GRID
#define SF_HEAD_GRID 768
#endif
constexpr int HEAD_GRID_MAX = SF_HEAD_GRID;   // workgroups of the head kernels (three per CU)

__device__ __forceinline__ void head_stage_weights(float* sw, const float* __restrict__ w, int C, int Creal, int RS) {
    if (RS == 16 && (reinterpret_cast<uintptr_t>(w) & 15) == 0) {
        // a thread takes one channel's 16 taps (64 contiguous bytes) and scatters them over the tap rows
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            float4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                v[q] = c < Creal ? *reinterpret_cast<const float4*>(w + (size_t)c * 16 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                sw[(4 * q + 0) * C + c] = v[q].x; sw[(4 * q + 1) * C + c] = v[q].y;
                sw[(4 * q + 2) * C + c] = v[q].z; sw[(4 * q + 3) * C + c] = v[q].w;
            }
        }
    } else {
        for (int i = threadIdx.x; i < RS * C; i += blockDim.x) {
            const int t = i / C, c = i - t * C;
            sw[i] = c < Creal ? w[c * RS + t] : 0.f;
        }
    }
    __syncthreads();
}

// a group of G lanes per output pixel
template <typename T>
__global__ __launch_bounds__(NT) void head_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ b, float* __restrict__ y, unsigned total,
                                                      int H, int W, int C, int Creal, int R, int pad, int OH, int OW) {
    constexpr int EPC = ElemTraits<T>::EPC;
    __shared__ float sw[HEAD_MAX_W];
    const int RS = R * R;
    head_stage_weights(sw, w, C, Creal, RS);
    const int G = C / EPC, ppb = NT / G;
    const int cg = threadIdx.x % G, slot = threadIdx.x / G;
    const float bias = b[0];
    for (unsigned o0 = blockIdx.x * ppb; o0 < total; o0 += gridDim.x * ppb) {
        const unsigned o = o0 + slot;
        const bool live = o < total;
        const unsigned oc = live ? o : o0;
        const unsigned q = oc / (unsigned)OW, ow = oc - q * OW;
        const unsigned n = q / (unsigned)OH, oh = q - n * OH;
        float acc = 0.f;
        for (int t0 = 0; t0 < RS; t0 += 8) {
            uint4 raw[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int t = t0 + k;
                const int ty = t / R, tx = t - ty * R;
                const int iy = (int)oh - pad + ty, ix = (int)ow - pad + tx;
                const bool ok = live && t < RS && iy >= 0 && iy < H && ix >= 0 && ix < W;
                raw[k] = *reinterpret_cast<const uint4*>(ok ? (const void*)(x + (((size_t)n * H + iy) * W + ix) * C + cg * EPC)
                                                            : (const void*)&g_zero16s);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int t = t0 + k < RS ? t0 + k : 0;
                float f[EPC];
                decode16<T>(raw[k], f);
                const float* wt = sw + t * C + cg * EPC;
#pragma unroll
                for (int e = 0; e < EPC; ++e) acc += f[e] * wt[e];
            }
        }
        for (int off = G >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (live && cg == 0) y[o] = acc + bias;
    }
}

// one thread per (input pixel, channel chunk)
template <typename T>
__global__ __launch_bounds__(NT) void head_bwd_data_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                           const T* __restrict__ aux, T* __restrict__ dx, unsigned total,
                                                           int H, int W, int C, int Creal, int R, int pad, int OH,
                                                           int OW) {
    constexpr int EPC = ElemTraits<T>::EPC;
    __shared__ float sw[HEAD_MAX_W];
    const int RS = R * R;
    head_stage_weights(sw, w, C, Creal, RS);
    const int G = C / EPC, ppb = NT / G;
    const int cg = threadIdx.x % G, slot = threadIdx.x / G;
    for (unsigned o0 = blockIdx.x * ppb; o0 < total; o0 += gridDim.x * ppb) {
        const unsigned o = o0 + slot;
        if (o >= total) continue;
        const unsigned q = o / (unsigned)W, ix = o - q * W;
        const unsigned n = q / (unsigned)H, iy = q - n * H;
        float g[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int ty = t / R, tx = t - ty * R;
            const int oh = (int)iy + pad - ty, ow = (int)ix + pad - tx;
            const bool ok = t < RS && oh >= 0 && oh < OH && ow >= 0 && ow < OW;
            const float v = dy[ok ? ((size_t)n * OH + oh) * OW + ow : 0];
            g[t] = ok ? v : 0.f;
        }
        float acc[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const float* wt = sw + (t < RS ? t : 0) * C + cg * EPC;
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[e] += g[t] * wt[e];
        }
        if (aux) {
            float a[EPC];
            load_chunk<T>(aux + (size_t)o * C + cg * EPC, a);
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[e] = a[e] > 0.f ? acc[e] : 0.f;
        }
        store_chunk<T>(dx + (size_t)o * C + cg * EPC, acc);
    }
}

// ---- MFMA forms of the 4x4 head for bf16 (C a multiple of 32, at most 512).
// Forward in two steps: P[pixel][tap] = sum_c x[pixel][c] * w[c][tap] (a [M x C] x [C x 16] product
// in which x crosses the memory system once, where the direct form re-reads every input pixel for
// each of the 16 outputs it feeds), then y[o] = b + sum_taps P[o + tap - pad][tap].
// Lane (n = lane & 15, q = lane >> 4) of v_mfma_f32_16x16x32_bf16: A row n, B column n, reduction
// elements 8q .. 8q+7; D rows 4q .. 4q+3, column n.
constexpr int HT_LD = 520;      // LDS row of one tap: 512 channels + 8 (16-lane reads spread over the banks)
__global__ __launch_bounds__(NT) void head_taps_mfma_kernel(const bf16* __restrict__ x, const float* __restrict__ w,
                                                            float* __restrict__ P, unsigned M, int C, int Creal) {
    __shared__ __attribute__((aligned(16))) bf16 sw[16 * HT_LD];
    for (int c = threadIdx.x; c < C; c += NT) {
        float f[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) f[t] = c < Creal ? w[(size_t)c * 16 + t] : 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) sw[t * HT_LD + c] = (bf16)f[t];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4;
    const unsigned wave = blockIdx.x * (NT / 64) + (threadIdx.x >> 6), nwaves = gridDim.x * (NT / 64);
    const unsigned ngroups = (M + 15) / 16;
    const int ksteps = C / 32;
    for (unsigned g = wave; g < ngroups; g += nwaves) {
        const unsigned pix = g * 16 + n;
        const bf16* xr = x + (size_t)(pix < M ? pix : 0) * C + 8 * q;
        uint4 a[16];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
            a[ks] = (ks < ksteps && pix < M) ? *reinterpret_cast<const uint4*>(xr + ks * 32) : make_uint4(0, 0, 0, 0);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            if (ks < ksteps) {      // (wave-uniform)
                const bf16x8 fb = *reinterpret_cast<const bf16x8*>(sw + n * HT_LD + ks * 32 + 8 * q);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a[ks]), fb, acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned row = g * 16 + 4 * q + i;
            if (row < M) P[(size_t)row * 16 + n] = acc[i];
        }
    }
}

__global__ __launch_bounds__(NT) void head_gather_kernel(const float* __restrict__ P, const float* __restrict__ b,
                                                         float* __restrict__ y, unsigned total, int H, int W, int OH,
                                                         int OW, int pad) {
    const unsigned o = blockIdx.x * NT + threadIdx.x;
    if (o >= total) return;
    const unsigned q = o / (unsigned)OW, ow = o - q * OW;
    const unsigned n = q / (unsigned)OH, oh = q - n * OH;
    float v[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int iy = (int)oh - pad + (t >> 2), ix = (int)ow - pad + (t & 3);
        const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
        const float pv = P[ok ? (((size_t)n * H + iy) * W + ix) * 16 + t : 0];
        v[t] = ok ? pv : 0.f;
    }
    // taps added in filter order, like the direct kernel's loop
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) acc += v[t];
    y[o] = acc + b[0];
}

// Data gradient: dx[pixel][c] = sum_taps dy[pixel + pad - tap] * w[c][tap] as one 32x32x16 MFMA per
// (32 pixels, 32 channels): A = weights (rows = channels), B = the 16 gathered output gradients of a
// pixel (columns = pixels), so that a lane ends up with 4-channel runs of one pixel, regrouped into
// 16-byte stores as in the 3x3 kernel's epilogue.
__global__ __launch_bounds__(NT) void head_bwd_data_mfma_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                                const bf16* __restrict__ aux, bf16* __restrict__ dx,
                                                                unsigned M, int H, int W, int C, int Creal, int pad,
                                                                int OH, int OW) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const unsigned wave = blockIdx.x * (NT / 64) + (threadIdx.x >> 6), nwaves = gridDim.x * (NT / 64);
    const unsigned ngroups = (M + 31) / 32;
    const int nblk = C / 32;
    for (unsigned g = wave; g < ngroups; g += nwaves) {
        const unsigned pix = g * 32 + r;
        const bool live = pix < M;
        const unsigned pc = live ? pix : 0;
        const unsigned qq = pc / (unsigned)W, ix = pc - qq * W;
        const unsigned n = qq / (unsigned)H, iy = qq - n * H;
        float gv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t = 8 * h + j;
            const int oh = (int)iy + pad - (t >> 2), ow = (int)ix + pad - (t & 3);
            const bool ok = live && oh >= 0 && oh < OH && ow >= 0 && ow < OW;
            const float v = dy[ok ? ((size_t)n * OH + oh) * OW + ow : 0];
            gv[j] = ok ? v : 0.f;
        }
        const bf16x8 fb = to_bf16x8(gv);
        for (int blk = 0; blk < nblk; ++blk) {
            const int c = blk * 32 + r;
            float wf[8];
            const float4 w0 = *reinterpret_cast<const float4*>(w + (size_t)(c < Creal ? c : 0) * 16 + 8 * h);
            const float4 w1 = *reinterpret_cast<const float4*>(w + (size_t)(c < Creal ? c : 0) * 16 + 8 * h + 4);
            wf[0] = w0.x; wf[1] = w0.y; wf[2] = w0.z; wf[3] = w0.w; wf[4] = w1.x; wf[5] = w1.y; wf[6] = w1.z; wf[7] = w1.w;
            if (c >= Creal)
#pragma unroll
                for (int j = 0; j < 8; ++j) wf[j] = 0.f;
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(to_bf16x8(wf), fb, acc, 0, 0, 0);
            // acc[4 g4 + e]: channel blk * 32 + 8 g4 + 4 h + e of pixel r
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                const unsigned A0 = pack_bf16x2(acc[8 * gp], acc[8 * gp + 1]), A1 = pack_bf16x2(acc[8 * gp + 2], acc[8 * gp + 3]);
                const unsigned B0 = pack_bf16x2(acc[8 * gp + 4], acc[8 * gp + 5]), B1 = pack_bf16x2(acc[8 * gp + 6], acc[8 * gp + 7]);
                const auto s0 = __builtin_amdgcn_permlane32_swap(A0, B0, false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(A1, B1, false, false);
                uint4 o = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                const size_t off = (size_t)pc * C + blk * 32 + 8 * (2 * gp + h);
                if (aux) {
                    const uint4 m = *reinterpret_cast<const uint4*>(aux + off);
                    // bf16 > 0  <=>  sign clear and not zero
                    auto keep = [](unsigned v, unsigned mk) {
                        const unsigned lo = (!(mk & 0x8000u) && (mk & 0x7fffu)) ? 0xffffu : 0u;
                        const unsigned hi = (!(mk & 0x80000000u) && (mk & 0x7fff0000u)) ? 0xffff0000u : 0u;
                        return v & (lo | hi);
                    };
                    o.x = keep(o.x, m.x); o.y = keep(o.y, m.y); o.z = keep(o.z, m.z); o.w = keep(o.w, m.w);
                }
                if (live) *reinterpret_cast<uint4*>(dx + off) = o;
            }
        }
    }
}

// dw[c][t] = sum over INPUT pixels of x[pixel][c] * dy[pixel + pad - tap t]: a thread owns one
// channel chunk, loads each input chunk once and applies all RS taps to it (the dy values are
// scalars from a tiny tensor), so x crosses the memory system once.
template <typename T>
__global__ __launch_bounds__(NT) void head_bwd_weight_kernel(const T* __restrict__ x, const float* __restrict__ dy,
                                                             float* __restrict__ dw, float* __restrict__ db,
                                                             unsigned M, unsigned per, unsigned Mout, int H, int W,
                                                             int C, int Creal, int R, int pad, int OH, int OW) {
    constexpr int EPC = ElemTraits<T>::EPC, U = 2;
    constexpr int NV = 16 * EPC;
    __shared__ float sm[8 * NT];
    const int G = C / EPC, rpp = NT / G, RS = R * R;
    const int cg = threadIdx.x % G, phase = threadIdx.x / G;
    float acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.f;
    const unsigned r0 = blockIdx.x * per;
    const unsigned r1 = r0 + per < M ? r0 + per : M;
    for (unsigned r = r0 + phase; r < r1; r += rpp * U) {
        uint4 raw[U];
        float g[U][16];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned rr = r + u * rpp;
            const bool live = rr < r1;
            const unsigned mc = live ? rr : r;
            const unsigned q = mc / (unsigned)W, ix = mc - q * W;
            const unsigned n = q / (unsigned)H, iy = q - n * H;
            raw[u] = *reinterpret_cast<const uint4*>(live ? (const void*)(x + (size_t)mc * C + cg * EPC) : (const void*)&g_zero16s);
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int ty = t / R, tx = t - ty * R;
                const int oh = (int)iy + pad - ty, ow = (int)ix + pad - tx;
                const bool ok = live && t < RS && oh >= 0 && oh < OH && ow >= 0 && ow < OW;
                const float v = dy[ok ? ((size_t)n * OH + oh) * OW + ow : 0];
                g[u][t] = ok ? v : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float f[EPC];
            decode16<T>(raw[u], f);
#pragma unroll
            for (int t = 0; t < 16; ++t)
#pragma unroll
                for (int e = 0; e < EPC; ++e) acc[t * EPC + e] += g[u][t] * f[e];
        }
    }
    reduce_phases_atomic<NV>(acc, G, rpp, sm, dw, [&](int gq, int v) -> long long {
        const int t = v / EPC, c = gq * EPC + v % EPC;
        return (t < RS && c < Creal) ? (long long)c * RS + t : -1;
    });
    // db = sum(dy): the first workgroup adds it
    if (blockIdx.x == 0) {
        float bs = 0.f;
        for (unsigned i = threadIdx.x; i < Mout; i += NT) bs += dy[i];
        bs = block_sum_256(bs, sm);
        if (threadIdx.x == 0) unsafeAtomicAdd(db, bs);
    }
}

// MFMA form for bf16 and 4x4 filters: D[channel][tap] = sum over input pixels of
// x[pixel][channel] * dy[pixel + pad - tap].  Lane (r, h) covers pixels 8h..8h+7 of a 16-pixel
// group: 8 dword loads of channels (2r, 2r+1) whose low / high halves feed two MFMAs (rows =
// even / odd channels of a 64-channel block), and 8 dy gathers for tap r (columns >= 16 unused).
// A workgroup owns one 64-channel block and a pixel range; its 4 waves interleave the groups.
__global__ __launch_bounds__(NT) void head_bwd_weight_mfma_kernel(const bf16* __restrict__ x,
                                                                  const float* __restrict__ dy,
                                                                  float* __restrict__ dw, float* __restrict__ db,
                                                                  unsigned M, unsigned per, unsigned Mout, int H, int W,
                                                                  int C, int Creal, int pad, int OH, int OW,
                                                                  int nsplit) {
    __shared__ float sm[NT / 64][16][64];
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5, wv = threadIdx.x >> 6;
    const int cb = blockIdx.x / nsplit, sp = blockIdx.x % nsplit;
    const unsigned m_lo = sp * per, m_hi = m_lo + per < M ? m_lo + per : M;
    const int ty = (r & 15) >> 2, tx = r & 3;
    f32x16 acc[2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    const bf16* xb = x + cb * 64 + 2 * r;
    for (unsigned g0 = m_lo + wv * 16; g0 < m_hi; g0 += (NT / 64) * 16) {
        unsigned d[8];
        float gy[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned m = g0 + 8 * h + j;
            const bool live = m < m_hi;
            const unsigned mc = live ? m : m_lo;
            d[j] = *reinterpret_cast<const unsigned*>(xb + (size_t)mc * C);
            const unsigned q = mc / (unsigned)W, ix = mc - q * W;
            const unsigned n = q / (unsigned)H, iy = q - n * H;
            const int oh = (int)iy + pad - ty, ow = (int)ix + pad - tx;
            const bool ok = live && r < 16 && oh >= 0 && oh < OH && ow >= 0 && ow < OW;
            gy[j] = dy[ok ? ((size_t)n * OH + oh) * OW + ow : 0];
            gy[j] = ok ? gy[j] : 0.f;  // a dead pixel's (valid) x meets a zero dy column
        }
        unsigned lo[4], hi[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            lo[i] = (d[2 * i] & 0xffffu) | (d[2 * i + 1] << 16);
            hi[i] = (d[2 * i] >> 16) | (d[2 * i + 1] & 0xffff0000u);
        }
        const bf16x8 fb = to_bf16x8(gy);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(lo), fb, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(hi), fb, acc[1], 0, 0, 0);
    }
    // acc[a][i]: channel cb*64 + 2*(8*(i>>2) + 4h + (i&3)) + a, tap r
    if (r < 16) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int i = 0; i < 16; ++i) sm[wv][r][2 * (8 * (i >> 2) + 4 * h + (i & 3)) + a] = acc[a][i];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 16 * 64; i += NT) {
        const int t = i >> 6, c = cb * 64 + (i & 63);
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < NT / 64; ++k) v += sm[k][t][i & 63];
        if (c < Creal) unsafeAtomicAdd(dw + (size_t)c * 16 + t, v);
    }
    if (blockIdx.x == 0) {
        float bs = 0.f;
        for (unsigned i = threadIdx.x; i < Mout; i += NT) bs += dy[i];
        __syncthreads();
        bs = block_sum_256(bs, &sm[0][0][0]);
        if (threadIdx.x == 0) unsafeAtomicAdd(db, bs);
    }
}

inline bool head_ok(int dtype, int C, int R, long long items) {
    const int g = C / (dtype == SF_BF16 ? 8 : 4);
    return R >= 1 && R <= 4 && g >= 1 && g <= 64 && (g & (g - 1)) == 0 && R * R * C <= HEAD_MAX_W && g * R * R <= 1024 &&
           (g * R * R) % 64 == 0 && items < 0x7fffffffLL;
}

// the MFMA forms of the 4x4 head: bf16, channel count a multiple of 32 up to 512
inline bool head_mfma_ok(int dtype, int C, int R, long long items) {
    return dtype == SF_BF16 && R == 4 && C % 32 == 0 && C >= 32 && C <= 512 && items < 0x7fffffffLL / 16;
}

struct RowSplit { unsigned per; int grid; };
inline RowSplit split_rows_small(long long items, int step, int cap) {
    long long grid = (items + step - 1) / step;
    if (grid > cap) grid = cap;
    if (grid < 1) grid = 1;
    long long per = (items + grid - 1) / grid;
    per = (per + step - 1) / step * step;
    grid = (items + per - 1) / per;
    RowSplit s;
    s.per = (unsigned)per; s.grid = (int)grid;
    return s;
}
inline bool pw1_ok(int dtype, int C, int R, int pad, long long M) {
    const int g = C / (dtype == SF_BF16 ? 8 : 4);
    return R == 1 && pad == 0 && g >= 1 && g <= 64 && (g & (g - 1)) == 0 && M < 0x7fffffffLL;
}

}  // namespace

#define DISPATCH_T(dtype, CALL)                                       \
    if ((dtype) == SF_BF16) { using T = bf16; CALL; }                 \
    else if ((dtype) == SF_F32) { using T = float; CALL; }            \
    else { sf_set_error("bad dtype %d", (int)(dtype)); return SF_EINVAL; }

static int dconv1_dims(const char* who, int dtype, int N, int H, int W, int K, int* OH, int* OW) {
    const int epc = dtype == SF_BF16 ? 8 : 4;
    if (N <= 0 || H < 2 || W < 2 || K <= 0 || K % epc != 0 || K / epc > NT || NT % (K / epc) != 0) {
        sf_set_error("%s: unsupported shape N=%d H=%d W=%d K=%d", who, N, H, W, K);
        return SF_EINVAL;
    }
    *OH = (H + 2 - 4) / 2 + 1;
    *OW = (W + 2 - 4) / 2 + 1;
    return SF_OK;
}

extern "C" int sf_dconv1_fwd(const float* x, const float* w, const float* b, void* y, int dtype, int N, int H, int W,
                             int K, void* stream) {
    SF_REQUIRE(x && w && b && y, "sf_dconv1_fwd: null argument");
    int OH, OW;
    if (int rc = dconv1_dims("sf_dconv1_fwd", dtype, N, H, W, K, &OH, &OW)) return rc;
    const int epc = dtype == SF_BF16 ? 8 : 4;
    const long long total = (long long)N * OH * OW * (K / epc);
    if (dtype == SF_BF16 && K == 64 && (long long)N * OH * OW < 0x7fffffffLL) {
        const long long M = (long long)N * OH * OW;
        long long grid = (M + 32 * 2 * (NT / 64) - 1) / (32 * 2 * (NT / 64));
        if (grid > 1024) grid = 1024;
        hipLaunchKernelGGL(dconv1_fwd_mfma_kernel, dim3((unsigned)grid), dim3(NT), 0, (hipStream_t)stream, x, w, b,
                           (bf16*)y, (unsigned)M, H, W, OH, OW);
        SF_LAUNCH_CHECK("sf_dconv1_fwd");
        return SF_OK;
    }
    DISPATCH_T(dtype, hipLaunchKernelGGL(dconv1_fwd_kernel<T>, dim3(grid_for(total)), dim3(NT), sizeof(float) * 17 * K,
                                         (hipStream_t)stream, x, w, b, (T*)y, N, H, W, K, OH, OW));
    SF_LAUNCH_CHECK("sf_dconv1_fwd");
    return SF_OK;
}

extern "C" int sf_dconv1_bwd_data(const void* dz, int dtype, const float* w, float* dx, int N, int H, int W, int K,
                                  void* stream) {
    SF_REQUIRE(dz && w && dx, "sf_dconv1_bwd_data: null argument");
    int OH, OW;
    if (int rc = dconv1_dims("sf_dconv1_bwd_data", dtype, N, H, W, K, &OH, &OW)) return rc;
    const int epc_ = dtype == SF_BF16 ? 8 : 4;
    SF_REQUIRE(((K / epc_) & (K / epc_ - 1)) == 0 && K / epc_ <= 64, "sf_dconv1_bwd_data: K/chunk must be a power of two <= 64");
    const long long total = (long long)N * H * W * (K / epc_);
    if (dtype == SF_BF16 && K == 64) {
        const int RB = 25, nstrips = (OW + 1 + 14) / 15, nchunks = (OH + 1 + RB - 1) / RB;
        const long long waves = (long long)N * nstrips * nchunks;
        SF_REQUIRE(waves < 0x7fffffffLL, "sf_dconv1_bwd_data: too many strips");
        hipLaunchKernelGGL(dconv1_bwd_data_mfma_kernel, dim3((unsigned)((waves + NT / 64 - 1) / (NT / 64))), dim3(NT), 0,
                           (hipStream_t)stream, (const bf16*)dz, w, dx, N, H, W, OH, OW, nstrips, nchunks, RB);
        SF_LAUNCH_CHECK("sf_dconv1_bwd_data");
        return SF_OK;
    }
    DISPATCH_T(dtype, hipLaunchKernelGGL(dconv1_bwd_data_kernel<T>, dim3(grid_for(total)), dim3(NT),
                                         sizeof(float) * 16 * K, (hipStream_t)stream, (const T*)dz, w, dx, N, H, W, K,
                                         OH, OW));
    SF_LAUNCH_CHECK("sf_dconv1_bwd_data");
    return SF_OK;
}

static long long dconv1_wgrad_grid(long long M) {
    long long g = ((M + 15) / 16 + 2 * (WG_NT / 64) - 1) / (2 * (WG_NT / 64));
    return g > 256 ? 256 : (g < 1 ? 1 : g);
}

extern "C" size_t sf_dconv1_bwd_weight_workspace_bytes(int N, int H, int W, int K) {
    if (N <= 0 || H < 2 || W < 2 || K != 64) return 0;
    const int OH = (H + 2 - 4) / 2 + 1, OW = (W + 2 - 4) / 2 + 1;
    return (size_t)dconv1_wgrad_grid((long long)N * OH * OW) * 17 * 64 * sizeof(float);
}

extern "C" int sf_dconv1_bwd_weight_ws(const float* x, const void* dz, int dtype, float* dw, float* db, float* ws,
                                       size_t ws_bytes, int N, int H, int W, int K, void* stream) {
    SF_REQUIRE(x && dz && dw && db, "sf_dconv1_bwd_weight: null argument");
    int OH, OW;
    if (int rc = dconv1_dims("sf_dconv1_bwd_weight", dtype, N, H, W, K, &OH, &OW)) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int epc = dtype == SF_BF16 ? 8 : 4;
    SF_REQUIRE(NT % (2 * (K / epc)) == 0, "sf_dconv1_bwd_weight: K=%d unsupported", K);
    const int rpp = NT / (2 * (K / epc));
    const long long M = (long long)N * OH * OW;
    SF_REQUIRE(M < 0x7fffffffLL, "sf_dconv1_bwd_weight: too many pixels");
    const bool mfma = dtype == SF_BF16 && K == 64 && OW >= 8;
    const bool slabs = mfma && ws != nullptr;
    if (slabs) SF_REQUIRE(ws_bytes >= sf_dconv1_bwd_weight_workspace_bytes(N, H, W, K),
                          "sf_dconv1_bwd_weight_ws: workspace of %zu bytes, %zu needed", ws_bytes,
                          sf_dconv1_bwd_weight_workspace_bytes(N, H, W, K));
    if (!slabs && (hipMemsetAsync(dw, 0, sizeof(float) * 16 * K, s) != hipSuccess ||
                   hipMemsetAsync(db, 0, sizeof(float) * K, s) != hipSuccess)) {
        sf_set_error("sf_dconv1_bwd_weight: memset failed");
        return SF_ELAUNCH;
    }
    if (mfma) {
        const long long g = dconv1_wgrad_grid(M);
        if (OW % 8 == 0)
            hipLaunchKernelGGL(dconv1_bwd_weight_mfma_kernel<false>, dim3((unsigned)g), dim3(WG_NT), 0, s, x, (const bf16*)dz,
                               dw, db, slabs ? ws : nullptr, (unsigned)M, H, W, OH, OW);
        else
            hipLaunchKernelGGL(dconv1_bwd_weight_mfma_kernel<true>, dim3((unsigned)g), dim3(WG_NT), 0, s, x, (const bf16*)dz,
                               dw, db, slabs ? ws : nullptr, (unsigned)M, H, W, OH, OW);
        SF_LAUNCH_CHECK("sf_dconv1_bwd_weight");
        if (slabs) {
            hipLaunchKernelGGL(dconv1_wfold_kernel, dim3(17), dim3(64), 0, s, ws, (int)g, dw, db);
            SF_LAUNCH_CHECK("sf_dconv1_bwd_weight (fold)");
        }
        return SF_OK;
    }
    const int step = rpp * 4;
    long long grid = (M + step - 1) / step; if (grid > 256) grid = 256; if (grid < 1) grid = 1;
    long long per = ((M + grid - 1) / grid + step - 1) / step * step;
    grid = (M + per - 1) / per;
    DISPATCH_T(dtype, hipLaunchKernelGGL(dconv1_bwd_weight_kernel<T>, dim3((unsigned)grid), dim3(NT), 0, s, x,
                                         (const T*)dz, dw, db, (unsigned)M, (unsigned)per, H, W, K, OH, OW));
    SF_LAUNCH_CHECK("sf_dconv1_bwd_weight");
    return SF_OK;
}

extern "C" int sf_dconv1_bwd_weight(const float* x, const void* dz, int dtype, float* dw, float* db, int N, int H,
                                    int W, int K, void* stream) {
    return sf_dconv1_bwd_weight_ws(x, dz, dtype, dw, db, nullptr, 0, N, H, W, K, stream);
}

static int k1_dims(const char* who, int dtype, int N, int H, int W, int C, int Creal, int R, int pad, int* OH,
                   int* OW) {
    const int epc = dtype == SF_BF16 ? 8 : 4;
    if (N <= 0 || C <= 0 || C % epc != 0 || Creal > C || Creal <= 0 || (R != 1 && R != 4) || pad < 0 ||
        C / epc > NT || NT % (C / epc) != 0) {
        sf_set_error("%s: unsupported shape N=%d H=%d W=%d C=%d R=%d", who, N, H, W, C, R);
        return SF_EINVAL;
    }
    *OH = H + 2 * pad - R + 1;
    *OW = W + 2 * pad - R + 1;
    if (*OH <= 0 || *OW <= 0) {
        sf_set_error("%s: empty output", who);
        return SF_EINVAL;
    }
    return SF_OK;
}

extern "C" int sf_conv_k1_fwd(const void* x, int dtype, const float* w, const float* b, float* y, int N, int H, int W,
                              int C, int Creal, int R, int pad, void* stream) {
    SF_REQUIRE(x && w && b && y, "sf_conv_k1_fwd: null argument");
    int OH, OW;
    if (int rc = k1_dims("sf_conv_k1_fwd", dtype, N, H, W, C, Creal, R, pad, &OH, &OW)) return rc;
    const int epc = dtype == SF_BF16 ? 8 : 4;
    int lpp = 1;
    while (lpp < C / epc && lpp < 64) lpp <<= 1;
    const long long total = (long long)N * OH * OW;
    if (pw1_ok(dtype, C, R, pad, total)) {
        const RowSplit sp = split_rows_small(total, (NT / (C / epc)) * 8, 4096);
        DISPATCH_T(dtype, hipLaunchKernelGGL(pw1_fwd_kernel<T>, dim3(sp.grid), dim3(NT), 0, (hipStream_t)stream,
                                             (const T*)x, w, b, y, (unsigned)total, sp.per, C, Creal));
        SF_LAUNCH_CHECK("sf_conv_k1_fwd");
        return SF_OK;
    }
    if (head_ok(dtype, C, R, total)) {
        // every workgroup stages the 32 KB of weights first: few, looping workgroups amortise that
        const int ppb = NT / (C / epc);
        long long grid = (total + ppb - 1) / ppb; if (grid > HEAD_GRID_MAX) grid = HEAD_GRID_MAX;
        DISPATCH_T(dtype, hipLaunchKernelGGL(head_fwd_kernel<T>, dim3((unsigned)grid), dim3(NT), 0, (hipStream_t)stream,
                                             (const T*)x, w, b, y, (unsigned)total, H, W, C, Creal, R, pad, OH, OW));
        SF_LAUNCH_CHECK("sf_conv_k1_fwd");
        return SF_OK;
    }
    DISPATCH_T(dtype, hipLaunchKernelGGL(conv_k1_fwd_kernel<T>, dim3(grid_for(total * lpp)), dim3(NT), 0,
                                         (hipStream_t)stream, (const T*)x, w, b, y, N, H, W, C, Creal, R, pad, OH, OW,
                                         lpp));
    SF_LAUNCH_CHECK("sf_conv_k1_fwd");
    return SF_OK;
}

extern "C" size_t sf_conv_k1_fwd_workspace_bytes(int dtype, int N, int H, int W, int C, int R) {
    const long long M = (long long)N * H * W;
    return head_mfma_ok(dtype, C, R, M) ? (size_t)M * 16 * sizeof(float) : 0;
}

extern "C" int sf_conv_k1_fwd_ws(const void* x, int dtype, const float* w, const float* b, float* y, void* ws,
                                 size_t ws_bytes, int N, int H, int W, int C, int Creal, int R, int pad, void* stream) {
    SF_REQUIRE(x && w && b && y, "sf_conv_k1_fwd_ws: null argument");
    int OH, OW;
    if (int rc = k1_dims("sf_conv_k1_fwd_ws", dtype, N, H, W, C, Creal, R, pad, &OH, &OW)) return rc;
    const size_t need = sf_conv_k1_fwd_workspace_bytes(dtype, N, H, W, C, R);
    if (need == 0) return sf_conv_k1_fwd(x, dtype, w, b, y, N, H, W, C, Creal, R, pad, stream);
    SF_REQUIRE(ws && ws_bytes >= need, "sf_conv_k1_fwd_ws: workspace of %zu bytes, %zu needed", ws_bytes, need);
    const unsigned M = (unsigned)((long long)N * H * W);
    const long long total = (long long)N * OH * OW;
    long long grid = (((long long)M + 15) / 16 + NT / 64 - 1) / (NT / 64); if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(head_taps_mfma_kernel, dim3((unsigned)grid), dim3(NT), 0, (hipStream_t)stream, (const bf16*)x, w,
                       (float*)ws, M, C, Creal);
    hipLaunchKernelGGL(head_gather_kernel, dim3((unsigned)((total + NT - 1) / NT)), dim3(NT), 0, (hipStream_t)stream,
                       (const float*)ws, b, y, (unsigned)total, H, W, OH, OW, pad);
    SF_LAUNCH_CHECK("sf_conv_k1_fwd_ws");
    return SF_OK;
}

extern "C" int sf_conv_k1_bwd_data(const float* dy, const float* w, const void* aux, void* dx, int dtype, int N, int H,
                                   int W, int C, int Creal, int R, int pad, void* stream) {
    SF_REQUIRE(dy && w && dx, "sf_conv_k1_bwd_data: null argument");
    int OH, OW;
    if (int rc = k1_dims("sf_conv_k1_bwd_data", dtype, N, H, W, C, Creal, R, pad, &OH, &OW)) return rc;
    const int epc = dtype == SF_BF16 ? 8 : 4;
    if (pw1_ok(dtype, C, R, pad, (long long)N * H * W)) {
        const long long M = (long long)N * H * W;
        const RowSplit sp = split_rows_small(M, (NT / (C / epc)) * 8, 4096);
        DISPATCH_T(dtype, hipLaunchKernelGGL(pw1_bwd_data_kernel<T>, dim3(sp.grid), dim3(NT), 0, (hipStream_t)stream, dy,
                                             w, (const T*)aux, (T*)dx, (unsigned)M, sp.per, C, Creal));
        SF_LAUNCH_CHECK("sf_conv_k1_bwd_data");
        return SF_OK;
    }
    if (head_mfma_ok(dtype, C, R, (long long)N * H * W)) {
        const long long items = (long long)N * H * W;
        long long grid = ((items + 31) / 32 + NT / 64 - 1) / (NT / 64); if (grid > 2048) grid = 2048;
        hipLaunchKernelGGL(head_bwd_data_mfma_kernel, dim3((unsigned)grid), dim3(NT), 0, (hipStream_t)stream, dy, w,
                           (const bf16*)aux, (bf16*)dx, (unsigned)items, H, W, C, Creal, pad, OH, OW);
        SF_LAUNCH_CHECK("sf_conv_k1_bwd_data");
        return SF_OK;
    }
    if (head_ok(dtype, C, R, (long long)N * H * W)) {
        const long long items = (long long)N * H * W;
        const int ppb = NT / (C / epc);
        long long grid = (items + ppb - 1) / ppb; if (grid > HEAD_GRID_MAX) grid = HEAD_GRID_MAX;
        DISPATCH_T(dtype, hipLaunchKernelGGL(head_bwd_data_kernel<T>, dim3((unsigned)grid), dim3(NT), 0,
                                             (hipStream_t)stream, dy, w, (const T*)aux, (T*)dx, (unsigned)items, H, W, C,
                                             Creal, R, pad, OH, OW));
        SF_LAUNCH_CHECK("sf_conv_k1_bwd_data");
        return SF_OK;
    }
    const long long total = (long long)N * H * W * (C / epc);
    DISPATCH_T(dtype, hipLaunchKernelGGL(conv_k1_bwd_data_kernel<T>, dim3(grid_for(total)), dim3(NT), 0,
                                         (hipStream_t)stream, dy, w, (const T*)aux, (T*)dx, N, H, W, C, Creal, R, pad,
                                         OH, OW));
    SF_LAUNCH_CHECK("sf_conv_k1_bwd_data");
    return SF_OK;
}

extern "C" int sf_conv_k1_bwd_weight(const void* x, int dtype, const float* dy, float* dw, float* db, int N, int H,
                                     int W, int C, int Creal, int R, int pad, void* stream) {
    SF_REQUIRE(x && dy && dw && db, "sf_conv_k1_bwd_weight: null argument");
    int OH, OW;
    if (int rc = k1_dims("sf_conv_k1_bwd_weight", dtype, N, H, W, C, Creal, R, pad, &OH, &OW)) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(dw, 0, sizeof(float) * Creal * R * R, s) != hipSuccess || hipMemsetAsync(db, 0, sizeof(float), s) != hipSuccess) {
        sf_set_error("sf_conv_k1_bwd_weight: memset failed");
        return SF_ELAUNCH;
    }
    const int epc = dtype == SF_BF16 ? 8 : 4;
    const int rpp = NT / (C / epc);
    const long long M = (long long)N * OH * OW;
    int grid = (int)((M + rpp * 16 - 1) / (rpp * 16)); if (grid > 256) grid = 256; if (grid < 1) grid = 1;
    if (pw1_ok(dtype, C, R, pad, M)) {
        const RowSplit sp = split_rows_small(M, rpp * 8, 256);
        DISPATCH_T(dtype, hipLaunchKernelGGL(pw1_bwd_weight_kernel<T>, dim3(sp.grid), dim3(NT), 0, s, (const T*)x, dy, dw,
                                             db, (unsigned)M, sp.per, C, Creal));
        SF_LAUNCH_CHECK("sf_conv_k1_bwd_weight");
        return SF_OK;
    }
    if (dtype == SF_BF16 && R == 4 && C % 64 == 0 && (long long)N * H * W < 0x7fffffffLL) {
        const long long Min = (long long)N * H * W;
        long long nsplit = (Min + 1023) / 1024;  // >= 16 groups of 16 pixels per wave
        if (nsplit > 32) nsplit = 32;
        long long per = ((Min + nsplit - 1) / nsplit + 15) / 16 * 16;
        nsplit = (Min + per - 1) / per;
        hipLaunchKernelGGL(head_bwd_weight_mfma_kernel, dim3((unsigned)((C / 64) * nsplit)), dim3(NT), 0, s,
                           (const bf16*)x, dy, dw, db, (unsigned)Min, (unsigned)per, (unsigned)M, H, W, C, Creal, pad, OH,
                           OW, (int)nsplit);
        SF_LAUNCH_CHECK("sf_conv_k1_bwd_weight");
        return SF_OK;
    }
    if (head_ok(dtype, C, R, (long long)N * H * W)) {
        const long long Min = (long long)N * H * W;
        const RowSplit sp = split_rows_small(Min, (NT / (C / epc)) * 2, 256);
        DISPATCH_T(dtype, hipLaunchKernelGGL(head_bwd_weight_kernel<T>, dim3(sp.grid), dim3(NT), 0, s, (const T*)x, dy, dw,
                                             db, (unsigned)Min, sp.per, (unsigned)M, H, W, C, Creal, R, pad, OH, OW));
        SF_LAUNCH_CHECK("sf_conv_k1_bwd_weight");
        return SF_OK;
    }
    if (R == 1) {
        DISPATCH_T(dtype, hipLaunchKernelGGL((conv_k1_bwd_weight_kernel<T, 1>), dim3(grid), dim3(NT), 0, s, (const T*)x,
                                             dy, dw, db, N, H, W, C, Creal, R, pad, OH, OW));
    } else {
        DISPATCH_T(dtype, hipLaunchKernelGGL((conv_k1_bwd_weight_kernel<T, 16>), dim3(grid), dim3(NT), 0, s,
                                             (const T*)x, dy, dw, db, N, H, W, C, Creal, R, pad, OH, OW));
    }
    SF_LAUNCH_CHECK("sf_conv_k1_bwd_weight");
    return SF_OK;
}

extern "C" int sf_conv_k1_bwd_fused(const void* x, int dtype, const float* dy, const float* w, void* dx, float* dw,
                                    float* db, float* dsum, long long M, int C, int Creal, void* stream) {
    SF_REQUIRE(x && dy && w && dx && dw && db && dsum, "sf_conv_k1_bwd_fused: null argument");
    SF_REQUIRE(dtype == SF_BF16 || dtype == SF_F32, "sf_conv_k1_bwd_fused: bad dtype %d", dtype);
    SF_REQUIRE(M > 0 && Creal > 0 && Creal <= C && pw1_ok(dtype, C, 1, 0, M),
               "sf_conv_k1_bwd_fused: C / chunk must be a power of two <= 64 (C=%d)", C);
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(dw, 0, sizeof(float) * Creal, s) != hipSuccess || hipMemsetAsync(db, 0, sizeof(float), s) != hipSuccess ||
        hipMemsetAsync(dsum, 0, sizeof(float) * Creal, s) != hipSuccess) {
        sf_set_error("sf_conv_k1_bwd_fused: memset failed");
        return SF_ELAUNCH;
    }
    const int epc = dtype == SF_BF16 ? 8 : 4;
    const RowSplit sp = split_rows_small(M, (NT / (C / epc)) * 8, 512);
    DISPATCH_T(dtype, hipLaunchKernelGGL(pw1_bwd_fused_kernel<T>, dim3(sp.grid), dim3(NT), 0, s, (const T*)x, dy, w, (T*)dx,
                                         dw, db, dsum, (unsigned)M, sp.per, C, Creal));
    SF_LAUNCH_CHECK("sf_conv_k1_bwd_fused");
    return SF_OK;
}
