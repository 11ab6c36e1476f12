// 3x3 convolution of a bilinearly resized map without materialising the resized map
// (decoder: F.interpolate(x, (OH,OW), bilinear) followed by Conv2d(C,K,3,padding=1),
// src/model/decoder.py:165-173).
//
// Both steps are linear and the resize acts on pixels only, so the channel mixing commutes with
// it:   conv(up(x))[p][k] = bias[k] + sum_tap up(Y_tap)[p + tap][k],   Y_tap[s][k] = sum_c W[k][c][tap] x[s][c]
// with up(.) evaluated as zero outside the resized image (the conv's zero padding).  Y is a small
// GEMM on the source pixels (9K outputs per source pixel); what remains per output pixel is
// 9 taps x 4 corners of K-vectors instead of 9*C*K multiply-adds, and the [B,OH,OW,C] resized
// map is never written or read.  The kernel makes the sum separable: per output row it first
// folds the three filter rows and the vertical interpolation into R[tx][source column][k] in
// LDS, then every output pixel is three horizontal interpolations of R.
//
// The backward runs the adjoint: G_tap[s][k] = sum_p lambda(p + tap -> s) dz[p][k] (horizontal
// pass into a temporary, vertical pass into G), after which the weight gradient (G^T x), the
// input gradient (G W) and the bias gradient (column sums of the centre tap) are small GEMMs.
#include "sf_common.h"
#include "../../include/sfhip.h"

namespace {

constexpr int UT = 256;

template <typename T> __device__ __forceinline__ void load8(const T* p, float* f);
template <> __device__ __forceinline__ void load8<float>(const float* p, float* f) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
template <> __device__ __forceinline__ void load8<bf16>(const bf16* p, float* f) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float* f);
template <> __device__ __forceinline__ void store8<float>(float* p, const float* f) {
    *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(f[4], f[5], f[6], f[7]);
}
template <> __device__ __forceinline__ void store8<bf16>(bf16* p, const float* f) {
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (bf16)f[i];
    *reinterpret_cast<bf16x8*>(p) = v;
}

// axis weights of resized coordinate q, all zero outside [0, n_out) (conv zero padding)
__device__ __forceinline__ AxisW padded_axis(int q, int n_in, int n_out) {
    const bool ok = q >= 0 && q < n_out;
    AxisW a = axis_w(ok ? q : 0, n_in, n_out);
    if (!ok) a.l0 = a.l1 = 0.f;
    return a;
}

// One workgroup per output row.  Workgroups equal mod 8 share an L2; an image's Y (9K floats per
// source pixel) stays in one of them.
template <typename T>
__global__ __launch_bounds__(UT) void upconv3_fwd_kernel(const float* __restrict__ Y, const float* __restrict__ bias,
                                                         T* __restrict__ out, int B, int h, int w, int OH, int OW,
                                                         int K, int act, int rstride) {
    extern __shared__ float R[];  // [w][3][K], rstride floats per source column
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int b = (slot / OH) * 8 + xcd, oy = slot % OH;
    if (b >= B) return;
    AxisW ay[3];
#pragma unroll
    for (int ty = 0; ty < 3; ++ty) ay[ty] = padded_axis(oy + ty - 1, h, OH);
    const int K4 = K / 4, units = w * 3 * K4;
    const float* Yb = Y + (size_t)b * h * w * 9 * K;
    for (int u = threadIdx.x; u < units; u += UT) {
        const int k4 = u % K4, t = u / K4, tx = t % 3, j = t / 3;
        float4 v[3][2];
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) {
            v[ty][0] = *reinterpret_cast<const float4*>(Yb + ((size_t)(ay[ty].i0 * w + j) * 9 + ty * 3 + tx) * K + k4 * 4);
            v[ty][1] = *reinterpret_cast<const float4*>(Yb + ((size_t)(ay[ty].i1 * w + j) * 9 + ty * 3 + tx) * K + k4 * 4);
        }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) {
            acc.x += ay[ty].l0 * v[ty][0].x + ay[ty].l1 * v[ty][1].x;
            acc.y += ay[ty].l0 * v[ty][0].y + ay[ty].l1 * v[ty][1].y;
            acc.z += ay[ty].l0 * v[ty][0].z + ay[ty].l1 * v[ty][1].z;
            acc.w += ay[ty].l0 * v[ty][0].w + ay[ty].l1 * v[ty][1].w;
        }
        *reinterpret_cast<float4*>(R + j * rstride + tx * K + k4 * 4) = acc;
    }
    __syncthreads();
    const int G8 = K / 8, nseg = UT / G8;
    const int cg = threadIdx.x % G8, seg = threadIdx.x / G8;
    if (seg >= nseg) return;
    const int x0 = (int)((long long)seg * OW / nseg), x1 = (int)((long long)(seg + 1) * OW / nseg);
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = bias ? bias[cg * 8 + e] : 0.f;
    T* orow = out + ((size_t)(b * OH + oy) * OW) * K + cg * 8;
    const float* Rc = R + cg * 8;
    AxisW a0 = padded_axis(x0 - 1, w, OW), a1 = padded_axis(x0, w, OW);
    for (int ox = x0; ox < x1; ++ox) {
        const AxisW a2 = padded_axis(ox + 1, w, OW);
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = bv[e];
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) {
            const AxisW a = tx == 0 ? a0 : tx == 1 ? a1 : a2;
            float r0[8], r1[8];
            load8<float>(Rc + a.i0 * rstride + tx * K, r0);
            load8<float>(Rc + a.i1 * rstride + tx * K, r1);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += a.l0 * r0[e] + a.l1 * r1[e];
        }
        if (act == SF_ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = o[e] < 0.f ? 0.f : o[e];   // keeps NaN, like torch.relu
        }
        store8<T>(orow + (size_t)ox * K, o);
        a0 = a1;
        a1 = a2;
    }
}

// Running sums of the adjoint interpolation along one axis: the walk visits resized coordinates
// in ascending order, a value goes to its left source index (weight l0) and its right one (l1);
// a source index is final -- and handed to `emit(index, sums)` -- once the left index has moved
// past it.  Source indices that receive nothing are emitted as zeros, so every index in
// [0, n_in) is emitted exactly once, whatever the scale.
template <int NV> struct AdjointCursor {  // sums kept as 4-wide vectors (NV / 4 of them per index)
    static constexpr int NQ = NV / 4;
    f32x4 s0[NQ], s1[NQ];
    int cur;
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int q = 0; q < NQ; ++q) s0[q] = s1[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        cur = -1;
    }
    // emit(cur) and slide the window one index to the right; emit ignores negative indices
    template <typename F> __device__ __forceinline__ void shift(F emit) {
        emit(cur, s0);
#pragma unroll
        for (int q = 0; q < NQ; ++q) { s0[q] = s1[q]; s1[q] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        ++cur;
    }
    template <typename F> __device__ __forceinline__ void add(const AxisW& a, const f32x4* f, F emit) {
        while (cur != a.i0) shift(emit);  // one step per advance when upsampling
        if (a.i1 != a.i0) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) { s0[q] += a.l0 * f[q]; s1[q] += a.l1 * f[q]; }
        } else {
            const float wv = a.l0 + a.l1;
#pragma unroll
            for (int q = 0; q < NQ; ++q) s0[q] += wv * f[q];
        }
    }
    template <typename F> __device__ __forceinline__ void finish(int n_in, F emit) {
        while (cur < n_in) shift(emit);
    }
};

// horizontal pass: tm[row][tx][j][k] = sum_ox lambda_x(ox + tx - 1 -> j) dz[row][ox][k].  A thread
// owns one gradient row and 8 channels, reads every chunk once (batches of U loads) and feeds
// the three column shifts.  The pass is bound by instruction issue (about 200 per output column:
// three axis evaluations, six cursor updates with their data-dependent branches), not by load
// latency: splitting a row over four threads with margins ran twice as long.
template <typename T>
__global__ __launch_bounds__(UT) void upconv3_bwd_x_kernel(const T* __restrict__ dz, float* __restrict__ tm,
                                                           unsigned rows, int w, int OW, int K) {
    constexpr int U = 8;
    const unsigned G8 = K / 8;
    const unsigned o = blockIdx.x * UT + threadIdx.x;
    if (o >= rows * G8) return;
    const unsigned cg = o % G8, r = o / G8;
    const T* row = dz + (size_t)r * OW * K + cg * 8;
    float* trow = tm + (size_t)r * 3 * w * K + cg * 8;
    // six 4-channel cursors (3 column shifts x 2 halves): arrays of sums end up in scratch
    AdjointCursor<4> c0a, c0b, c1a, c1b, c2a, c2b;
    c0a.init(); c0b.init(); c1a.init(); c1b.init(); c2a.init(); c2b.init();
    // the column shifts tx = 0, 1, 2 of output column ox use the axis weights of ox - 1, ox, ox + 1:
    // one evaluation per column, kept in a sliding window
    AxisW ax_prev = axis_w(0, w, OW), ax_cur = ax_prev;
    auto feed = [&](AdjointCursor<4>& c, int tx, int half, const AxisW& ax, const f32x4& fv) __attribute__((always_inline)) {
        float* dst = trow + (size_t)tx * w * K + half * 4;
        c.add(ax, &fv, [&](int col, const f32x4* v) __attribute__((always_inline)) {
            if (col >= 0) *reinterpret_cast<f32x4*>(dst + (size_t)col * K) = v[0];
        });
    };
    // two batches of U loads in flight: the next batch is issued before the current one is fed
    float f[2][U][8];
    auto fetch = [&](int ox0, float (*dst)[8]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < U; ++u) load8<T>(row + (size_t)(ox0 + u < OW ? ox0 + u : OW - 1) * K, dst[u]);
    };
    auto consume = [&](int ox0, const float (*src)[8]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ox = ox0 + u;
            if (ox < OW) {
                const f32x4 fa = {src[u][0], src[u][1], src[u][2], src[u][3]}, fb = {src[u][4], src[u][5], src[u][6], src[u][7]};
                const AxisW ax_next = axis_w(ox + 1 < OW ? ox + 1 : OW - 1, w, OW);
                if (ox > 0) { feed(c0a, 0, 0, ax_prev, fa); feed(c0b, 0, 1, ax_prev, fb); }
                feed(c1a, 1, 0, ax_cur, fa); feed(c1b, 1, 1, ax_cur, fb);
                if (ox + 1 < OW) { feed(c2a, 2, 0, ax_next, fa); feed(c2b, 2, 1, ax_next, fb); }
                ax_prev = ax_cur;
                ax_cur = ax_next;
            }
        }
    };
    fetch(0, f[0]);
    for (int ox0 = 0; ox0 < OW; ox0 += 2 * U) {
        fetch(ox0 + U < OW ? ox0 + U : 0, f[1]);
        consume(ox0, f[0]);
        fetch(ox0 + 2 * U < OW ? ox0 + 2 * U : 0, f[0]);
        consume(ox0 + U, f[1]);
    }
    auto done = [&](AdjointCursor<4>& c, int tx, int half) __attribute__((always_inline)) {
        float* dst = trow + (size_t)tx * w * K + half * 4;
        c.finish(w, [&](int col, const f32x4* v) __attribute__((always_inline)) {
            if (col >= 0) *reinterpret_cast<f32x4*>(dst + (size_t)col * K) = v[0];
        });
    };
    done(c0a, 0, 0); done(c0b, 0, 1); done(c1a, 1, 0); done(c1b, 1, 1); done(c2a, 2, 0); done(c2b, 2, 1);
}

// vertical pass: g[b][i][j][ty*3+tx][k] = sum_oy lambda_y(oy + ty - 1 -> i) tm[b][oy][tx][j][k].
// A thread owns (image, source-row segment, tx, j, 4 channels) and walks the output rows that
// can reach its segment (plus a margin whose partial sums are dropped).
template <typename TG>
__global__ __launch_bounds__(UT) void upconv3_bwd_y_kernel(const float* __restrict__ tm, TG* __restrict__ g, int B,
                                                           int h, int w, int OH, int K, int nseg) {
    constexpr int U = 8;
    const unsigned K4 = K / 4, per_img = 3 * w * K4;
    const unsigned o = blockIdx.x * UT + threadIdx.x;
    if (o >= (unsigned)B * nseg * per_img) return;
    const unsigned k4 = o % K4, j = (o / K4) % w, tx = (o / (K4 * w)) % 3;
    const unsigned seg = (o / per_img) % nseg, b = o / (per_img * nseg);
    const int ia = (int)((long long)seg * h / nseg), ib = (int)((long long)(seg + 1) * h / nseg);
    int lo = 0, hi = OH - 1;
    if (nseg > 1) {
        const float f = (float)OH / (float)h;  // >= 1 when nseg > 1
        lo = (int)floorf((float)(ia - 2) * f) - 3;
        hi = (int)ceilf((float)(ib + 1) * f) + 3;
        if (lo < 0 || ia == 0) lo = 0;
        if (hi > OH - 1 || ib == h) hi = OH - 1;
    }
    const float* col = tm + (((size_t)b * OH * 3 + tx) * w + j) * K + k4 * 4;  // + oy * 3*w*K
    const size_t ostride = (size_t)3 * w * K;
    TG* gb = g + (((size_t)b * h * w + j) * 9 + tx) * K + k4 * 4;  // + i * w*9*K + ty * 3*K
    AdjointCursor<4> c0, c1, c2;
    c0.init(); c1.init(); c2.init();
    auto emit_to = [&](int ty) __attribute__((always_inline)) {
        return [=](int i, const f32x4* v) __attribute__((always_inline)) {
            if (i >= ia && i < ib) {
                Vec4<TG> q;
#pragma unroll
                for (int e = 0; e < 4; ++e) q.set(e, v[0][e]);
                q.store(gb + (size_t)i * w * 9 * K + (size_t)ty * 3 * K);
            }
        };
    };
    auto feed = [&](AdjointCursor<4>& c, int ty, int oy, const f32x4* fv) __attribute__((always_inline)) {
        const int qy = oy + ty - 1;
        if (qy < 0 || qy >= OH) return;
        c.add(axis_w(qy, h, OH), fv, emit_to(ty));
    };
    for (int oy0 = lo; oy0 <= hi; oy0 += U) {
        float4 f[U];
#pragma unroll
        for (int u = 0; u < U; ++u) f[u] = *reinterpret_cast<const float4*>(col + (size_t)(oy0 + u <= hi ? oy0 + u : hi) * ostride);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int oy = oy0 + u;
            if (oy <= hi) {
                const f32x4 fv[1] = {f32x4{f[u].x, f[u].y, f[u].z, f[u].w}};
                feed(c0, 0, oy, fv);
                feed(c1, 1, oy, fv);
                feed(c2, 2, oy, fv);
            }
        }
    }
    c0.finish(h, emit_to(0));
    c1.finish(h, emit_to(1));
    c2.finish(h, emit_to(2));
}

int check_dims(const char* who, int B, int h, int w, int OH, int OW, int K) {
    if (B <= 0 || h <= 0 || w <= 0 || OH <= 0 || OW <= 0 || K <= 0 || K % 8 != 0 || K / 8 > UT ||
        (long long)B * OH * OW * K >= (1LL << 40) || (long long)B * OH * (K / 8) >= 0x7fffffffLL ||
        (long long)B * 8 * 3 * w * (K / 4) >= 0x7fffffffLL) {
        sf_set_error("%s: unsupported shape B=%d h=%d w=%d OH=%d OW=%d K=%d", who, B, h, w, OH, OW, K);
        return SF_EINVAL;
    }
    return SF_OK;
}

}  // namespace

extern "C" int sf_upconv3_fwd(const float* y9, const float* bias, void* out, int dtype, int B, int h, int w, int OH,
                              int OW, int K, int act, void* stream) {
    SF_REQUIRE(y9 && out, "sf_upconv3_fwd: null argument");
    SF_REQUIRE(act == SF_ACT_NONE || act == SF_ACT_RELU, "sf_upconv3_fwd: act must be none or relu");
    if (int rc = check_dims("sf_upconv3_fwd", B, h, w, OH, OW, K)) return rc;
    const int rstride = 3 * K + 4;
    const size_t lds = sizeof(float) * (size_t)w * rstride;
    SF_REQUIRE(lds <= 64 * 1024, "sf_upconv3_fwd: w*K too large for the row buffer (%zu bytes of LDS)", lds);
    const long long grid = (long long)((B + 7) / 8) * 8 * OH;
    SF_REQUIRE(grid < 0x7fffffffLL, "sf_upconv3_fwd: too many rows");
    if (dtype == SF_BF16)
        hipLaunchKernelGGL(upconv3_fwd_kernel<bf16>, dim3((unsigned)grid), dim3(UT), lds, (hipStream_t)stream, y9, bias,
                           (bf16*)out, B, h, w, OH, OW, K, act, rstride);
    else if (dtype == SF_F32)
        hipLaunchKernelGGL(upconv3_fwd_kernel<float>, dim3((unsigned)grid), dim3(UT), lds, (hipStream_t)stream, y9, bias,
                           (float*)out, B, h, w, OH, OW, K, act, rstride);
    else { sf_set_error("sf_upconv3_fwd: bad dtype %d", dtype); return SF_EINVAL; }
    SF_LAUNCH_CHECK("sf_upconv3_fwd");
    return SF_OK;
}

extern "C" int sf_upconv3_bwd(const void* dz, int dtype, float* tmp, void* g, int g_dtype, int B, int h, int w, int OH,
                              int OW, int K, void* stream) {
    SF_REQUIRE(dz && tmp && g, "sf_upconv3_bwd: null argument");
    if (int rc = check_dims("sf_upconv3_bwd", B, h, w, OH, OW, K)) return rc;
    hipStream_t s = (hipStream_t)stream;
    const unsigned rows = (unsigned)B * OH;
    const long long t1 = (long long)rows * (K / 8);
    if (dtype == SF_BF16)
        hipLaunchKernelGGL(upconv3_bwd_x_kernel<bf16>, dim3((unsigned)((t1 + UT - 1) / UT)), dim3(UT), 0, s,
                           (const bf16*)dz, tmp, rows, w, OW, K);
    else if (dtype == SF_F32)
        hipLaunchKernelGGL(upconv3_bwd_x_kernel<float>, dim3((unsigned)((t1 + UT - 1) / UT)), dim3(UT), 0, s,
                           (const float*)dz, tmp, rows, w, OW, K);
    else { sf_set_error("sf_upconv3_bwd: bad dtype %d", dtype); return SF_EINVAL; }
    SF_LAUNCH_CHECK("sf_upconv3_bwd");
    int nseg = 1;
    if (OH >= h) { nseg = h / 8; if (nseg > 8) nseg = 8; if (nseg < 1) nseg = 1; }
    const long long t2 = (long long)B * nseg * 3 * w * (K / 4);
    if (g_dtype == SF_BF16)
        hipLaunchKernelGGL(upconv3_bwd_y_kernel<bf16>, dim3((unsigned)((t2 + UT - 1) / UT)), dim3(UT), 0, s, tmp,
                           (bf16*)g, B, h, w, OH, K, nseg);
    else if (g_dtype == SF_F32)
        hipLaunchKernelGGL(upconv3_bwd_y_kernel<float>, dim3((unsigned)((t2 + UT - 1) / UT)), dim3(UT), 0, s, tmp,
                           (float*)g, B, h, w, OH, K, nseg);
    else { sf_set_error("sf_upconv3_bwd: bad g_dtype %d", g_dtype); return SF_EINVAL; }
    SF_LAUNCH_CHECK("sf_upconv3_bwd");
    return SF_OK;
}
