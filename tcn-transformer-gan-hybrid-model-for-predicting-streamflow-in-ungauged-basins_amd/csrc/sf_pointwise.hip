// HBM-bound passes: layout changes, weight packing, BatchNorm finalisation, BN + ReLU + adaptive
// average pool forward, bilinear resize.
// Every kernel moves 16-byte channel chunks of NHWC tensors per lane.
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

inline int grid_for(long long items, int cap = 4096) {
    long long b = (items + NT - 1) / NT;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

// ------------------------------------------------------------------------------ layout
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, T* __restrict__ out, long long npix_total,
                                    int C, int HW, int Cpad) {
    constexpr int EPC = ElemTraits<T>::EPC;
    for (long long pix = blockIdx.x * (long long)NT + threadIdx.x; pix < npix_total;
         pix += (long long)gridDim.x * NT) {
        const long long n = pix / HW;
        const int hw = (int)(pix - n * HW);
        const float* src = in + n * C * (long long)HW + hw;
        for (int c0 = 0; c0 < Cpad; c0 += EPC) {
            float f[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) f[e] = (c0 + e < C) ? src[(long long)(c0 + e) * HW] : 0.f;
            store_chunk<T>(out + pix * Cpad + c0, f);
        }
    }
}

// Few channels (Cpad <= 16, bf16): a thread converts 4 consecutive pixels -- one float4 load per
// channel (a wave reads 1 KB runs of each channel plane), 4 x Cpad contiguous bf16 out.  The
// output of a wave is one contiguous run (the NHWC offset is linear in the global pixel index), but
// a thread's own 16-byte pieces sit 4 x Cpad x 2 bytes from its neighbour's: stored directly, every
// store instruction touches 64 different lines.  The pieces therefore pass through a per-wave LDS
// slice (9 slots per thread: the odd stride spreads the 16-byte writes over the banks) and leave
// as whole kilobytes, lane l writing piece l of each.
// CR > 0: the real channel count at compile time (only those planes are loaded); CR = 0: C at run
// time, padding channels re-read plane 0 and discard it.
template <int CP, int CR>
__global__ __launch_bounds__(NT) void nchw_to_nhwc_px4_kernel(const float* __restrict__ in, bf16* __restrict__ out,
                                                              unsigned nquads, int Crun, unsigned HW, int* __restrict__ flag) {
    // flag (may be null): *flag |= 1 when a loaded element is Inf or NaN -- the input check of
    // src/train.py:193 rides on the layout pass instead of reading the images a second time
    constexpr int PPT = 4 * CP / 8;          // 16-byte pieces per thread
    __shared__ uint4 sm[NT / 64][64 * 9];
    static_assert(PPT <= 8, "nine slots per thread hold at most eight pieces");
    const int C = CR ? CR : Crun;
    const unsigned q0 = blockIdx.x * NT + threadIdx.x;
    const unsigned q = q0 < nquads ? q0 : nquads - 1;      // lanes past the end load the last quad and store nothing
    const unsigned pix = q * 4, n = pix / HW, hw = pix - n * HW;
    const float* src = in + (size_t)n * C * HW + hw;
    float4 v[CP];
#pragma unroll
    for (int c = 0; c < CP; ++c) {
        if (CR && c >= CR) v[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        else v[c] = *reinterpret_cast<const float4*>(src + (size_t)(c < C ? c : 0) * HW);
    }
    if (flag) {
        unsigned top = 0;       // largest |x| as bits: exponent all ones <=> Inf or NaN
#pragma unroll
        for (int c = 0; c < CP; ++c) {
            if (CR && c >= CR) continue;
            top = max(top, __float_as_uint(v[c].x) & 0x7fffffffu);
            top = max(top, __float_as_uint(v[c].y) & 0x7fffffffu);
            top = max(top, __float_as_uint(v[c].z) & 0x7fffffffu);
            top = max(top, __float_as_uint(v[c].w) & 0x7fffffffu);
        }
        if (__any(top >= 0x7f800000u) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint4* mine = &sm[wave][0];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int c0 = 0; c0 < CP; c0 += 8) {
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = c0 + e < C ? (&v[c0 + e].x)[k] : 0.f;
            store_chunk<bf16>(reinterpret_cast<bf16*>(mine + lane * 9 + k * (CP / 8) + c0 / 8), f);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // the wave's run of pieces starts at its first quad; piece idx of the run belongs to thread idx / PPT
    const size_t first = (size_t)(blockIdx.x * NT + wave * 64) * PPT, total = (size_t)nquads * PPT;
    uint4* o16 = reinterpret_cast<uint4*>(out);
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int idx = j * 64 + lane;
        const uint4 piece = mine[(idx / PPT) * 9 + idx % PPT];
        if (first + idx < total) o16[first + idx] = piece;
    }
}

// Many channels: a workgroup transposes a tile of 64 pixels x 64 channels through LDS, so that
// both the channel-plane reads (256-byte runs) and the pixel-row writes (128-byte runs) coalesce.
template <typename T, typename TI = float>
__global__ __launch_bounds__(NT) void nchw_to_nhwc_tile_kernel(const TI* __restrict__ in, T* __restrict__ out, int C,
                                                               int HW, int Cpad, int ptiles, int ctiles) {
    constexpr int EPC = ElemTraits<T>::EPC;
    __shared__ float sm[64][65];
    const int ct = blockIdx.x % ctiles, rest = blockIdx.x / ctiles;
    const int pt = rest % ptiles, n = rest / ptiles;
    const int c0 = ct * 64, p0 = pt * 64;
    {
        const int px = threadIdx.x & 63, cq = threadIdx.x >> 6;
        const bool pok = p0 + px < HW;
        const TI* src = in + ((size_t)n * C + c0) * HW + p0 + (pok ? px : 0);
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ch = cq * 16 + i;
            v[i] = (float)src[(size_t)(c0 + ch < C ? ch : 0) * HW];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ch = cq * 16 + i;
            sm[ch][px] = (pok && c0 + ch < C) ? v[i] : 0.f;
        }
    }
    __syncthreads();
    constexpr int CHUNKS = 64 / EPC;            // chunks per pixel row of the tile
    for (int i = threadIdx.x; i < 64 * CHUNKS; i += NT) {
        const int chunk = i % CHUNKS, px = i / CHUNKS;
        if (p0 + px >= HW || c0 + chunk * EPC >= Cpad) continue;
        float f[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) f[e] = sm[chunk * EPC + e][px];
        store_chunk<T>(out + ((size_t)n * HW + p0 + px) * Cpad + c0 + chunk * EPC, f);
    }
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ in, float* __restrict__ out, long long total, int C,
                                    int HW, int Cpad) {
    for (long long o = blockIdx.x * (long long)NT + threadIdx.x; o < total; o += (long long)gridDim.x * NT) {
        const int hw = (int)(o % HW);
        const long long nc = o / HW;
        const int c = (int)(nc % C);
        const long long n = nc / C;
        out[o] = to_f32(in[(n * HW + hw) * Cpad + c]);
    }
}

// Many channels: the mirror of nchw_to_nhwc_tile_kernel.  A workgroup moves 64 pixels x 64 channels
// through LDS: pixel rows are read in 256-byte (f32) / 128-byte (bf16) runs, channel planes written
// in 256-byte runs (read directly, a wave touches 64 lines per load).
template <typename T>
__global__ __launch_bounds__(NT) void nhwc_to_nchw_tile_kernel(const T* __restrict__ in, float* __restrict__ out, int C,
                                                               int HW, int Cpad, int ptiles, int ctiles) {
    __shared__ float sm[64][65];                 // [pixel][channel]
    const int ct = blockIdx.x % ctiles, rest = blockIdx.x / ctiles;
    const int pt = rest % ptiles, n = rest / ptiles;
    const int c0 = ct * 64, p0 = pt * 64;
    {
        const int ch = threadIdx.x & 63, pq = threadIdx.x >> 6;
        const bool cok = c0 + ch < C;
        const T* src = in + ((size_t)n * HW + p0) * Cpad + c0 + (cok ? ch : 0);
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int px = pq * 16 + i;
            v[i] = to_f32(src[(size_t)(p0 + px < HW ? px : 0) * Cpad]);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) sm[pq * 16 + i][ch] = v[i];
    }
    __syncthreads();
    const int px = threadIdx.x & 63, cq = threadIdx.x >> 6;
    if (p0 + px < HW) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ch = cq * 16 + i;
            if (c0 + ch < C) out[((size_t)n * C + c0 + ch) * HW + p0 + px] = sm[px][ch];
        }
    }
}

struct PackArgs {
    int O, I, RS, transpose, ntaps, rows_pad, cols_pad;
    int tap_rs[16];
};
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ w, T* __restrict__ out, PackArgs a) {
    const long long total = (long long)a.rows_pad * a.ntaps * a.cols_pad;
    for (long long o = blockIdx.x * (long long)NT + threadIdx.x; o < total; o += (long long)gridDim.x * NT) {
        const int col = (int)(o % a.cols_pad);
        const long long rt = o / a.cols_pad;
        const int t = (int)(rt % a.ntaps);
        const int row = (int)(rt / a.ntaps);
        const int oc = a.transpose ? col : row;
        const int ic = a.transpose ? row : col;
        float v = 0.f;
        if (oc < a.O && ic < a.I) v = w[((long long)oc * a.I + ic) * a.RS + a.tap_rs[t]];
        out[o] = from_f32<T>(v);
    }
}

template <typename T> __global__ void cast_kernel(const float* __restrict__ in, T* __restrict__ out, long long n) {
    for (long long i = blockIdx.x * (long long)NT + threadIdx.x; i < n; i += (long long)gridDim.x * NT)
        out[i] = from_f32<T>(in[i]);
}

// ------------------------------------------------------------------------------ BN finalize
__global__ void bn_finalize_kernel(const double* __restrict__ ws, long long M, int C, int Creal,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ rm, float* __restrict__ rv, float momentum, float eps,
                                   int training, float* __restrict__ scale, float* __restrict__ shift,
                                   float* __restrict__ mean_o, float* __restrict__ invstd_o) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        if (c >= Creal) {
            scale[c] = 0.f; shift[c] = 0.f; mean_o[c] = 0.f; invstd_o[c] = 0.f;
            continue;
        }
        float mean, var;
        if (training) {
            const double m = ws[c] / (double)M;
            double v = ws[C + c] / (double)M - m * m;
            if (v < 0.0) v = 0.0;
            mean = (float)m;
            var = (float)v;
            const double unb = M > 1 ? v * ((double)M / (double)(M - 1)) : v;
            rm[c] = (1.f - momentum) * rm[c] + momentum * mean;
            rv[c] = (1.f - momentum) * rv[c] + momentum * (float)unb;
        } else {
            mean = rm[c];
            var = rv[c];
        }
        const float inv = 1.0f / sqrtf(var + eps);
        const float sc = gamma[c] * inv;
        scale[c] = sc;
        shift[c] = beta[c] - mean * sc;
        mean_o[c] = mean;
        invstd_o[c] = inv;
    }
}

// ------------------------------------------------------------------------------ BN + ReLU + adaptive avgpool -> NCHW
__device__ __forceinline__ int bin_start(int i, int in, int out) { return (i * in) / out; }
__device__ __forceinline__ int bin_end(int i, int in, int out) { return ((i + 1) * in + out - 1) / out; }

template <typename T, typename TO>
__global__ void bn_relu_avgpool_fwd_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                           const float* __restrict__ shift, TO* __restrict__ y, int N, int H,
                                           int W, int C, int Creal, int OH, int OW) {
    constexpr int EPC = ElemTraits<T>::EPC;
    constexpr int CS = 128;  // channel slab staged through LDS
    extern __shared__ float tile[];  // [CS][OW+1]
    const int n = blockIdx.x / OH, oh = blockIdx.x % OH;
    const int h0 = bin_start(oh, H, OH), h1 = bin_end(oh, H, OH);
    const int ldt = OW + 1;
    for (int c0 = 0; c0 < Creal; c0 += CS) {
        const int cs = (Creal - c0) < CS ? (Creal - c0) : CS;
        const int gs = (cs + EPC - 1) / EPC;
        __syncthreads();
        for (int it = threadIdx.x; it < gs * OW; it += NT) {
            const int cg = it % gs, ow = it / gs;
            const int w0 = bin_start(ow, W, OW), w1 = bin_end(ow, W, OW);
            float sc[EPC], sf[EPC], acc[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                sc[e] = scale[c0 + cg * EPC + e]; sf[e] = shift[c0 + cg * EPC + e]; acc[e] = 0.f;
            }
            for (int h = h0; h < h1; ++h)
                for (int w = w0; w < w1; ++w) {
                    float f[EPC];
                    load_chunk<T>(x + (((long long)n * H + h) * W + w) * C + c0 + cg * EPC, f);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        const float v = f[e] * sc[e] + sf[e];
                        acc[e] += v < 0.f ? 0.f : v;   // ReLU that keeps NaN
                    }
                }
            const float inv = 1.0f / (float)((h1 - h0) * (w1 - w0));
#pragma unroll
            for (int e = 0; e < EPC; ++e) tile[(cg * EPC + e) * ldt + ow] = acc[e] * inv;
        }
        __syncthreads();
        for (int it = threadIdx.x; it < cs * OW; it += NT) {
            const int ow = it % OW, c = it / OW;
            y[(((long long)n * Creal + c0 + c) * OH + oh) * OW + ow] = from_f32<TO>(tile[c * ldt + ow]);
        }
    }
}

// ------------------------------------------------------------------------------ bilinear
// src NHWC f32 [B,h,w,C] -> dst NHWC T [B,OH,OW,C]
template <typename T>
__global__ void bilinear_fwd_kernel(const float* __restrict__ src, T* __restrict__ dst, int B, int C, int h,
                                    int w, int OH, int OW) {
    constexpr int EPC = ElemTraits<T>::EPC;
    const int G = C / EPC;
    const long long total = (long long)B * OH * OW * G;
    for (long long o = blockIdx.x * (long long)NT + threadIdx.x; o < total; o += (long long)gridDim.x * NT) {
        const int cg = (int)(o % G);
        long long r = o / G;
        const int ox = (int)(r % OW); r /= OW;
        const int oy = (int)(r % OH);
        const long long b = r / OH;
        const AxisW ay = axis_w(oy, h, OH), ax = axis_w(ox, w, OW);
        const float* base = src + b * h * (long long)w * C + cg * EPC;
        float out[EPC];
#pragma unroll
        for (int q = 0; q < EPC; q += 4) {
            const float4 p00 = *reinterpret_cast<const float4*>(base + ((long long)ay.i0 * w + ax.i0) * C + q);
            const float4 p01 = *reinterpret_cast<const float4*>(base + ((long long)ay.i0 * w + ax.i1) * C + q);
            const float4 p10 = *reinterpret_cast<const float4*>(base + ((long long)ay.i1 * w + ax.i0) * C + q);
            const float4 p11 = *reinterpret_cast<const float4*>(base + ((long long)ay.i1 * w + ax.i1) * C + q);
            const float* a00 = &p00.x; const float* a01 = &p01.x; const float* a10 = &p10.x; const float* a11 = &p11.x;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                out[q + e] = ay.l0 * (ax.l0 * a00[e] + ax.l1 * a01[e]) + ay.l1 * (ax.l0 * a10[e] + ax.l1 * a11[e]);
        }
        store_chunk<T>(dst + o * EPC, out);
    }
}

// Upsampling along x (OW >= w: the left source column of consecutive outputs advances by 0 or
// 1): a thread walks a segment of one output row for one 16-byte channel group and keeps the
// two source rows of the current column pair in registers (plus the next column, loaded one
// advance ahead), so every output costs 6 flops per element and one store.  Same association
// as the generic kernel.  Lanes of a wave share the segment, so the advance test is uniform.
template <typename T>
__global__ __launch_bounds__(NT) void bilinear_up_walk_kernel(const float* __restrict__ src, T* __restrict__ dst,
                                                              unsigned rows /* B*OH */, int C, int h, int w, int OH,
                                                              int OW, int nseg) {
    constexpr int EPC = ElemTraits<T>::EPC;
    const unsigned G = C / EPC;
    const unsigned o = blockIdx.x * NT + threadIdx.x;
    if (o >= rows * G * nseg) return;
    const unsigned cg = o % G, r = (o / G) % rows, seg = o / (G * rows);
    const unsigned oy = r % OH, b = r / OH;
    const int x0 = (int)((long long)seg * OW / nseg), x1 = (int)((long long)(seg + 1) * OW / nseg);
    const AxisW ay = axis_w(oy, h, OH);
    const float* top = src + ((size_t)(b * h + ay.i0) * w) * C + cg * EPC;
    const float* bot = src + ((size_t)(b * h + ay.i1) * w) * C + cg * EPC;
    float a0[EPC], a1[EPC], b0[EPC], b1[EPC], c0[EPC], c1[EPC];
    auto load_col = [&](int col, float* t, float* u) {
        const int cc = col < w ? col : w - 1;
#pragma unroll
        for (int q = 0; q < EPC; q += 4) {
            const float4 v = *reinterpret_cast<const float4*>(top + (size_t)cc * C + q);
            const float4 z = *reinterpret_cast<const float4*>(bot + (size_t)cc * C + q);
            t[q] = v.x; t[q + 1] = v.y; t[q + 2] = v.z; t[q + 3] = v.w;
            u[q] = z.x; u[q + 1] = z.y; u[q + 2] = z.z; u[q + 3] = z.w;
        }
    };
    int cur = axis_w(x0, w, OW).i0;
    load_col(cur, a0, a1);
    load_col(cur + 1, b0, b1);
    load_col(cur + 2, c0, c1);
    T* out_row = dst + (size_t)r * OW * C + cg * EPC;
    for (int ox = x0; ox < x1; ++ox) {
        const AxisW ax = axis_w(ox, w, OW);
        if (ax.i0 != cur) {
            cur = ax.i0;
#pragma unroll
            for (int e = 0; e < EPC; ++e) { a0[e] = b0[e]; a1[e] = b1[e]; b0[e] = c0[e]; b1[e] = c1[e]; }
            load_col(cur + 2, c0, c1);
        }
        // ax.i1 is cur + 1, or cur at the right edge, where column cur + 1 is clamped to cur too
        float out[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e)
            out[e] = ay.l0 * (ax.l0 * a0[e] + ax.l1 * b0[e]) + ay.l1 * (ax.l0 * a1[e] + ax.l1 * b1[e]);
        store_chunk<T>(out_row + (size_t)ox * C, out);
    }
}

// Backward of the same along x (OW >= w): a thread walks one whole gradient row for one channel
// group, reads every chunk once (batches of 8 unconditional loads) and adds it into the sums of
// its left and right source columns; a column's sum is stored when the walk leaves it.
template <typename T>
__global__ __launch_bounds__(NT) void bilinear_bwd_x_walk_kernel(const T* __restrict__ ddst, float* __restrict__ tmp,
                                                                 unsigned rows /* B*OH */, int C, int w, int OW) {
    constexpr int EPC = ElemTraits<T>::EPC;
    constexpr int U = 8;
    const unsigned G = C / EPC;
    const unsigned o = blockIdx.x * NT + threadIdx.x;
    if (o >= rows * G) return;
    const unsigned cg = o % G, r = o / G;
    const T* row = ddst + (size_t)r * OW * C + cg * EPC;
    float* trow = tmp + (size_t)r * w * C + cg * EPC;
    float acc0[EPC], acc1[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc0[e] = acc1[e] = 0.f;
    auto flush = [&](int col, const float* v) {
#pragma unroll
        for (int q = 0; q < EPC; q += 4)
            *reinterpret_cast<float4*>(trow + (size_t)col * C + q) = make_float4(v[q], v[q + 1], v[q + 2], v[q + 3]);
    };
    int cur = 0;
    for (int ox0 = 0; ox0 < OW; ox0 += U) {
        float f[U][EPC];
#pragma unroll
        for (int u = 0; u < U; ++u) load_chunk<T>(row + (size_t)(ox0 + u < OW ? ox0 + u : OW - 1) * C, f[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (ox0 + u >= OW) break;
            const AxisW a = axis_w(ox0 + u, w, OW);
            if (a.i0 != cur) {
                flush(cur, acc0);
                cur = a.i0;
#pragma unroll
                for (int e = 0; e < EPC; ++e) { acc0[e] = acc1[e]; acc1[e] = 0.f; }
            }
            if (a.i1 != a.i0) {
#pragma unroll
                for (int e = 0; e < EPC; ++e) { acc0[e] += a.l0 * f[u][e]; acc1[e] += a.l1 * f[u][e]; }
            } else {
                const float wv = a.l0 + a.l1;
#pragma unroll
                for (int e = 0; e < EPC; ++e) acc0[e] += wv * f[u][e];
            }
        }
    }
    flush(cur, acc0);
    if (cur + 1 < w) flush(cur + 1, acc1);
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc1[e] = 0.f;
    for (int col = cur + 2; col < w; ++col) flush(col, acc1);
}

// pass 1: tmp[b][oy][sx][c] = sum_ox wx(ox -> sx) * ddst[b][oy][ox][c]
template <typename T>
__global__ void bilinear_bwd_x_kernel(const T* __restrict__ ddst, float* __restrict__ tmp, int B, int C, int w,
                                      int OH, int OW) {
    constexpr int EPC = ElemTraits<T>::EPC;
    const int G = C / EPC;
    const long long total = (long long)B * OH * w * G;
    const float inv_scale = (float)OW / (float)w;
    for (long long o = blockIdx.x * (long long)NT + threadIdx.x; o < total; o += (long long)gridDim.x * NT) {
        const int cg = (int)(o % G);
        long long r = o / G;
        const int sx = (int)(r % w); r /= w;  // r = b*OH + oy
        int lo = (int)floorf(((float)sx - 0.5f) * inv_scale - 0.5f) - 2; if (lo < 0) lo = 0;
        int hi = (int)ceilf(((float)sx + 1.5f) * inv_scale - 0.5f) + 2; if (hi > OW - 1) hi = OW - 1;
        if (sx == 0) lo = 0;
        if (sx == w - 1) hi = OW - 1;
        float acc[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
        // batches of 4 unconditional loads (clamped column, weight 0 past the end)
        for (int ox = lo; ox <= hi; ox += 4) {
            float f[4][EPC], wt[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int oc = ox + u <= hi ? ox + u : hi;
                const AxisW a = axis_w(oc, w, OW);
                const float wv = (a.i0 == sx ? a.l0 : 0.f) + (a.i1 == sx ? a.l1 : 0.f);
                wt[u] = ox + u <= hi ? wv : 0.f;
                load_chunk<T>(ddst + (r * OW + oc) * C + cg * EPC, f[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < EPC; ++e) acc[e] += wt[u] * f[u][e];
        }
#pragma unroll
        for (int q = 0; q < EPC; q += 4)
            *reinterpret_cast<float4*>(tmp + (r * w + sx) * C + cg * EPC + q) =
                make_float4(acc[q], acc[q + 1], acc[q + 2], acc[q + 3]);
    }
}

// pass 2: dsrc[b][sy][sx][c] (NHWC f32) = sum_oy wy(oy -> sy) * tmp[b][oy][sx][c]
__global__ void bilinear_bwd_y_kernel(const float* __restrict__ tmp, float* __restrict__ dsrc, int B, int C,
                                      int h, int w, int OH) {
    const int G = C / 4;
    const long long total = (long long)B * h * w * G;
    const float inv_scale = (float)OH / (float)h;
    for (long long o = blockIdx.x * (long long)NT + threadIdx.x; o < total; o += (long long)gridDim.x * NT) {
        const int cg = (int)(o % G);
        long long r = o / G;
        const int sx = (int)(r % w); r /= w;
        const int sy = (int)(r % h);
        const long long b = r / h;
        int lo = (int)floorf(((float)sy - 0.5f) * inv_scale - 0.5f) - 2; if (lo < 0) lo = 0;
        int hi = (int)ceilf(((float)sy + 1.5f) * inv_scale - 0.5f) + 2; if (hi > OH - 1) hi = OH - 1;
        if (sy == 0) lo = 0;
        if (sy == h - 1) hi = OH - 1;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int oy = lo; oy <= hi; ++oy) {
            const AxisW a = axis_w(oy, h, OH);
            const float wt = (a.i0 == sy ? a.l0 : 0.f) + (a.i1 == sy ? a.l1 : 0.f);
            if (wt == 0.f) continue;
            const float4 f = *reinterpret_cast<const float4*>(tmp + ((b * OH + oy) * w + sx) * (long long)C + cg * 4);
            acc.x += wt * f.x; acc.y += wt * f.y; acc.z += wt * f.z; acc.w += wt * f.w;
        }
        *reinterpret_cast<float4*>(dsrc + o * 4) = acc;
    }
}

}  // namespace

#define DISPATCH_T(dtype, CALL)                                       \
    if ((dtype) == SF_BF16) { using T = bf16; CALL; }                 \
    else if ((dtype) == SF_F32) { using T = float; CALL; }            \
    else { sf_set_error("bad dtype %d", (int)(dtype)); return SF_EINVAL; }

extern "C" int sf_nchw_bf16_to_nhwc(const void* in, void* out, int N, int C, int H, int W, int Cpad, void* stream) {
    SF_REQUIRE(in && out && N > 0 && C > 0 && H > 0 && W > 0 && Cpad >= C, "sf_nchw_bf16_to_nhwc: bad arguments");
    SF_REQUIRE(Cpad % 8 == 0 && Cpad >= 64, "sf_nchw_bf16_to_nhwc: Cpad=%d must be a multiple of 8, at least 64", Cpad);
    const int HWi = H * W;
    const int ptiles = (HWi + 63) / 64, ctiles = (Cpad + 63) / 64;
    const long long blocks = (long long)N * ptiles * ctiles;
    SF_REQUIRE(blocks < 0x7fffffffLL, "sf_nchw_bf16_to_nhwc: tensor too large");
    hipLaunchKernelGGL((nchw_to_nhwc_tile_kernel<bf16, bf16>), dim3((unsigned)blocks), dim3(NT), 0, (hipStream_t)stream,
                       (const bf16*)in, (bf16*)out, C, HWi, Cpad, ptiles, ctiles);
    SF_LAUNCH_CHECK("sf_nchw_bf16_to_nhwc");
    return SF_OK;
}

extern "C" int sf_nan_inf_flag(const float* x, long long n, int* flag, void* stream);

static int nchw_to_nhwc_impl(const float* in, void* out, int dtype, int N, int C, int H, int W, int Cpad, int* flag,
                             void* stream) {
    SF_REQUIRE(in && out && N > 0 && C > 0 && H > 0 && W > 0 && Cpad >= C, "sf_nchw_to_nhwc: bad arguments");
    SF_REQUIRE(Cpad % (dtype == SF_BF16 ? 8 : 4) == 0, "sf_nchw_to_nhwc: Cpad=%d not a chunk multiple", Cpad);
    const long long npix = (long long)N * H * W;
    const int HWi = H * W;
    if (dtype == SF_BF16 && (Cpad == 8 || Cpad == 16) && HWi % 4 == 0 && npix < 0x7fffffffLL && ((uintptr_t)in & 15) == 0) {
        const unsigned nquads = (unsigned)(npix / 4);
        const unsigned grid = (nquads + NT - 1) / NT;
        if (Cpad == 8)
            hipLaunchKernelGGL((nchw_to_nhwc_px4_kernel<8, 0>), dim3(grid), dim3(NT), 0, (hipStream_t)stream, in, (bf16*)out,
                               nquads, C, (unsigned)HWi, flag);
        else if (C == 10)     // the reference's ten input bands
            hipLaunchKernelGGL((nchw_to_nhwc_px4_kernel<16, 10>), dim3(grid), dim3(NT), 0, (hipStream_t)stream, in, (bf16*)out,
                               nquads, C, (unsigned)HWi, flag);
        else
            hipLaunchKernelGGL((nchw_to_nhwc_px4_kernel<16, 0>), dim3(grid), dim3(NT), 0, (hipStream_t)stream, in, (bf16*)out,
                               nquads, C, (unsigned)HWi, flag);
        SF_LAUNCH_CHECK("sf_nchw_to_nhwc");
        return SF_OK;
    }
    // the other layout kernels do not check: a pass of its own over the input
    if (flag) {
        if (int rc = sf_nan_inf_flag(in, (long long)N * C * H * W, flag, stream)) return rc;
    }
    if (Cpad >= 64) {
        const int ptiles = (HWi + 63) / 64, ctiles = (Cpad + 63) / 64;
        const long long blocks = (long long)N * ptiles * ctiles;
        if (blocks < 0x7fffffffLL) {
            DISPATCH_T(dtype, hipLaunchKernelGGL(nchw_to_nhwc_tile_kernel<T>, dim3((unsigned)blocks), dim3(NT), 0,
                                                 (hipStream_t)stream, in, (T*)out, C, HWi, Cpad, ptiles, ctiles));
            SF_LAUNCH_CHECK("sf_nchw_to_nhwc");
            return SF_OK;
        }
    }
    DISPATCH_T(dtype, hipLaunchKernelGGL(nchw_to_nhwc_kernel<T>, dim3(grid_for(npix)), dim3(NT), 0,
                                         (hipStream_t)stream, in, (T*)out, npix, C, H * W, Cpad));
    SF_LAUNCH_CHECK("sf_nchw_to_nhwc");
    return SF_OK;
}

extern "C" int sf_nchw_to_nhwc(const float* in, void* out, int dtype, int N, int C, int H, int W, int Cpad,
                               void* stream) {
    return nchw_to_nhwc_impl(in, out, dtype, N, C, H, W, Cpad, nullptr, stream);
}

extern "C" int sf_nchw_to_nhwc_checked(const float* in, void* out, int dtype, int N, int C, int H, int W, int Cpad,
                                       int* flag, void* stream) {
    SF_REQUIRE(flag, "sf_nchw_to_nhwc_checked: null flag");
    return nchw_to_nhwc_impl(in, out, dtype, N, C, H, W, Cpad, flag, stream);
}

extern "C" int sf_nhwc_to_nchw(const void* in, int dtype, float* out, int N, int C, int H, int W, int Cpad,
                               void* stream) {
    SF_REQUIRE(in && out && N > 0 && C > 0 && Cpad >= C, "sf_nhwc_to_nchw: bad arguments");
    const long long total = (long long)N * C * H * W;
    if (C >= 32 && NT == 256) {
        const int HWi = H * W, ptiles = (HWi + 63) / 64, ctiles = (C + 63) / 64;
        const long long blocks = (long long)N * ptiles * ctiles;
        if (blocks < 0x7fffffffLL) {
            DISPATCH_T(dtype, hipLaunchKernelGGL(nhwc_to_nchw_tile_kernel<T>, dim3((unsigned)blocks), dim3(NT), 0,
                                                 (hipStream_t)stream, (const T*)in, out, C, HWi, Cpad, ptiles, ctiles));
            SF_LAUNCH_CHECK("sf_nhwc_to_nchw");
            return SF_OK;
        }
    }
    DISPATCH_T(dtype, hipLaunchKernelGGL(nhwc_to_nchw_kernel<T>, dim3(grid_for(total)), dim3(NT), 0,
                                         (hipStream_t)stream, (const T*)in, out, total, C, H * W, Cpad));
    SF_LAUNCH_CHECK("sf_nhwc_to_nchw");
    return SF_OK;
}

extern "C" int sf_pack_conv_weight(const float* w, void* out, int dtype, int O, int I, int RS, int transpose,
                                   int ntaps, const int* tap_rs, int rows_pad, int cols_pad, void* stream) {
    SF_REQUIRE(w && out && tap_rs && ntaps >= 1 && ntaps <= 16, "sf_pack_conv_weight: bad arguments");
    SF_REQUIRE(rows_pad >= (transpose ? I : O) && cols_pad >= (transpose ? O : I), "sf_pack_conv_weight: padding smaller than the tensor");
    PackArgs a;
    a.O = O; a.I = I; a.RS = RS; a.transpose = transpose; a.ntaps = ntaps; a.rows_pad = rows_pad; a.cols_pad = cols_pad;
    for (int t = 0; t < 16; ++t) a.tap_rs[t] = t < ntaps ? tap_rs[t] : 0;
    for (int t = 0; t < ntaps; ++t) SF_REQUIRE(tap_rs[t] >= 0 && tap_rs[t] < RS, "sf_pack_conv_weight: tap outside the filter");
    const long long total = (long long)rows_pad * ntaps * cols_pad;
    DISPATCH_T(dtype, hipLaunchKernelGGL(pack_weight_kernel<T>, dim3(grid_for(total)), dim3(NT), 0,
                                         (hipStream_t)stream, w, (T*)out, a));
    SF_LAUNCH_CHECK("sf_pack_conv_weight");
    return SF_OK;
}

extern "C" int sf_cast(const float* in, void* out, int dtype, long long n, void* stream) {
    SF_REQUIRE(in && out && n > 0, "sf_cast: bad arguments");
    DISPATCH_T(dtype, hipLaunchKernelGGL(cast_kernel<T>, dim3(grid_for(n)), dim3(NT), 0, (hipStream_t)stream, in,
                                         (T*)out, n));
    SF_LAUNCH_CHECK("sf_cast");
    return SF_OK;
}

static int check_chunked(const char* who, int dtype, int C) {
    const int epc = dtype == SF_BF16 ? 8 : 4;
    if (C <= 0 || C % epc != 0) {
        sf_set_error("%s: channel count %d must be a positive multiple of %d", who, C, epc);
        return SF_EINVAL;
    }
    if (C / epc > NT) {
        sf_set_error("%s: channel count %d too wide", who, C);
        return SF_EINVAL;
    }
    return SF_OK;
}

extern "C" int sf_bn_finalize(const double* ws, long long M, int C, int Creal, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, float momentum, float eps, int training,
                              float* scale, float* shift, float* mean, float* invstd, void* stream) {
    SF_REQUIRE(gamma && beta && running_mean && running_var && scale && shift && mean && invstd, "sf_bn_finalize: null argument");
    SF_REQUIRE(!training || (ws && M > 0), "sf_bn_finalize: training needs statistics");
    SF_REQUIRE(Creal <= C && C > 0, "sf_bn_finalize: bad channel counts");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream, ws, M, C, Creal, gamma, beta,
                       running_mean, running_var, momentum, eps, training, scale, shift, mean, invstd);
    SF_LAUNCH_CHECK("sf_bn_finalize");
    return SF_OK;
}

extern "C" int sf_bn_relu_avgpool_fwd(const void* x, int dtype, const float* scale, const float* shift, void* y,
                                      int out_dtype, int N, int H, int W, int C, int Creal, int OH, int OW,
                                      void* stream) {
    SF_REQUIRE(x && y && scale && shift && N > 0 && OH > 0 && OW > 0 && Creal <= C, "sf_bn_relu_avgpool_fwd: bad arguments");
    if (int rc = check_chunked("sf_bn_relu_avgpool_fwd", dtype, C)) return rc;
    SF_REQUIRE(OW <= 128, "sf_bn_relu_avgpool_fwd: OW=%d > 128", OW);
    const size_t lds = sizeof(float) * 128 * (OW + 1);
    dim3 grid((unsigned)(N * OH)), block(NT);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SF_BF16 && out_dtype == SF_BF16)
        hipLaunchKernelGGL((bn_relu_avgpool_fwd_kernel<bf16, bf16>), grid, block, lds, s, (const bf16*)x, scale, shift, (bf16*)y, N, H, W, C, Creal, OH, OW);
    else if (dtype == SF_BF16 && out_dtype == SF_F32)
        hipLaunchKernelGGL((bn_relu_avgpool_fwd_kernel<bf16, float>), grid, block, lds, s, (const bf16*)x, scale, shift, (float*)y, N, H, W, C, Creal, OH, OW);
    else if (dtype == SF_F32 && out_dtype == SF_F32)
        hipLaunchKernelGGL((bn_relu_avgpool_fwd_kernel<float, float>), grid, block, lds, s, (const float*)x, scale, shift, (float*)y, N, H, W, C, Creal, OH, OW);
    else { sf_set_error("sf_bn_relu_avgpool_fwd: unsupported dtype pair %d -> %d", dtype, out_dtype); return SF_EINVAL; }
    SF_LAUNCH_CHECK("sf_bn_relu_avgpool_fwd");
    return SF_OK;
}

extern "C" int sf_bilinear_fwd(const void* src, int src_dtype, void* dst, int dtype, int B, int C, int h, int w,
                               int OH, int OW, int Cpad, void* stream) {
    SF_REQUIRE(src && dst && B > 0 && h > 0 && w > 0 && OH > 0 && OW > 0, "sf_bilinear_fwd: bad arguments");
    SF_REQUIRE(src_dtype == SF_F32, "sf_bilinear_fwd: source must be NHWC f32");
    SF_REQUIRE(C == Cpad && Cpad % (dtype == SF_BF16 ? 8 : 4) == 0, "sf_bilinear_fwd: C=%d must equal Cpad and be a chunk multiple", C);
    const int epc = dtype == SF_BF16 ? 8 : 4;
    if (OW >= w && OW >= 16 && (long long)B * OH * (C / epc) * 64 < 0x7fffffffLL) {
        // enough threads to fill the chip, segments of at least 16 outputs
        const long long per_seg = (long long)B * OH * (C / epc);
        long long nseg = (400000 + per_seg - 1) / per_seg;
        if (nseg > OW / 16) nseg = OW / 16;
        if (nseg > 64) nseg = 64;
        if (nseg < 1) nseg = 1;
        const long long items = per_seg * nseg;
        DISPATCH_T(dtype, hipLaunchKernelGGL(bilinear_up_walk_kernel<T>, dim3((unsigned)((items + NT - 1) / NT)), dim3(NT),
                                             0, (hipStream_t)stream, (const float*)src, (T*)dst, (unsigned)(B * OH), C,
                                             h, w, OH, OW, (int)nseg));
        SF_LAUNCH_CHECK("sf_bilinear_fwd");
        return SF_OK;
    }
    const long long total = (long long)B * OH * OW * (C / epc);
    DISPATCH_T(dtype, hipLaunchKernelGGL(bilinear_fwd_kernel<T>, dim3(grid_for(total, 16384)), dim3(NT), 0,
                                         (hipStream_t)stream, (const float*)src, (T*)dst, B, C, h, w, OH, OW));
    SF_LAUNCH_CHECK("sf_bilinear_fwd");
    return SF_OK;
}

extern "C" int sf_bilinear_bwd(const void* ddst, int dtype, float* dsrc, float* tmp, int B, int C, int h, int w,
                               int OH, int OW, int Cpad, void* stream) {
    SF_REQUIRE(ddst && dsrc && tmp && B > 0, "sf_bilinear_bwd: bad arguments");
    SF_REQUIRE(C == Cpad && Cpad % (dtype == SF_BF16 ? 8 : 4) == 0, "sf_bilinear_bwd: C must equal Cpad and be a chunk multiple");
    const int epc = dtype == SF_BF16 ? 8 : 4;
    hipStream_t s = (hipStream_t)stream;
    const long long t1 = (long long)B * OH * w * (C / epc);
    if (OW >= w && (long long)B * OH * (C / epc) < 0x7fffffffLL) {
        const long long items = (long long)B * OH * (C / epc);
        DISPATCH_T(dtype, hipLaunchKernelGGL(bilinear_bwd_x_walk_kernel<T>, dim3((unsigned)((items + NT - 1) / NT)),
                                             dim3(NT), 0, s, (const T*)ddst, tmp, (unsigned)(B * OH), C, w, OW));
    } else {
        DISPATCH_T(dtype, hipLaunchKernelGGL(bilinear_bwd_x_kernel<T>, dim3(grid_for(t1, 16384)), dim3(NT), 0, s,
                                             (const T*)ddst, tmp, B, C, w, OH, OW));
    }
    const long long t2 = (long long)B * h * w * (C / 4);
    hipLaunchKernelGGL(bilinear_bwd_y_kernel, dim3(grid_for(t2, 16384)), dim3(NT), 0, s, tmp, dsrc, B, C, h, w, OH);
    SF_LAUNCH_CHECK("sf_bilinear_bwd");
    return SF_OK;
}

// ------------------------------------------------------------------------------ ReLU + dropout
// y = dropout(relu(x), p): the Linear -> ReLU -> Dropout steps of PureBPModel's MLP
// (src/model/baseline_models.py:181-190).  Kept elements are scaled by 1 / (1 - p); the mask of
// element i is rand_u(seed, i) >= p.  The backward needs no seed: y > 0 exactly where the
// element passed both the ReLU and the mask.
__global__ __launch_bounds__(256) void relu_dropout_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                               long long n, float p, unsigned long long seed) {
    const float keep = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float v = x[i];
        const float m = (p > 0.f && rand_u(seed, (unsigned long long)i) < p) ? 0.f : keep;
        y[i] = v < 0.f ? 0.f : v * m;   // ReLU that keeps NaN
    }
}
__global__ __launch_bounds__(256) void relu_dropout_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                               float* __restrict__ dx, long long n, float p) {
    const float keep = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        dx[i] = y[i] > 0.f ? dy[i] * keep : 0.f;
}

extern "C" int sf_relu_dropout_fwd(const float* x, float* y, long long n, float p, unsigned long long seed,
                                   void* stream) {
    SF_REQUIRE(x && y && n > 0 && p >= 0.f && p < 1.f, "sf_relu_dropout_fwd: bad arguments");
    long long grid = (n + 255) / 256; if (grid > 4096) grid = 4096;
    relu_dropout_fwd_kernel<<<dim3((unsigned)grid), 256, 0, (hipStream_t)stream>>>(x, y, n, p, seed);
    SF_LAUNCH_CHECK("sf_relu_dropout_fwd");
    return SF_OK;
}

extern "C" int sf_relu_dropout_bwd(const float* y, const float* dy, float* dx, long long n, float p, void* stream) {
    SF_REQUIRE(y && dy && dx && n > 0 && p >= 0.f && p < 1.f, "sf_relu_dropout_bwd: bad arguments");
    long long grid = (n + 255) / 256; if (grid > 4096) grid = 4096;
    relu_dropout_bwd_kernel<<<dim3((unsigned)grid), 256, 0, (hipStream_t)stream>>>(y, dy, dx, n, p);
    SF_LAUNCH_CHECK("sf_relu_dropout_bwd");
    return SF_OK;
}
