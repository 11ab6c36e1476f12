// Losses, optimiser and small utilities.
//  * image + station loss: forward and backward in one stream-ordered sequence with no host
//    synchronisation; the station term is a gather-reduce with one wavefront per station
//    (lanes over the k x k window, shuffle reduction) and a scatter of its gradient.
//  * BCE-with-logits against a constant label (adversarial terms).
//  * fused Adam, NaN/Inf flag, lon/lat -> pixel map.
#include "sf_common.h"
#include "../../include/sfhip.h"

namespace {

constexpr int NT = 256;

struct AxisW { int i0, i1; float l0, l1; };
__device__ __forceinline__ AxisW axis_w(int dst, int in, int out) {
    const float scale = (float)in / (float)out;
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    AxisW a;
    a.i0 = (int)src;
    if (a.i0 > in - 1) a.i0 = in - 1;
    a.i1 = a.i0 + (a.i0 < in - 1 ? 1 : 0);
    a.l1 = src - (float)a.i0;
    a.l0 = 1.f - a.l1;
    return a;
}

struct LossArgs {
    const float* pred; const float* target; const int* pos; const float* runoff; const int* offsets;
    int S, B, H, W, TH, TW, image_metric, ksize, station_metric;
    float nonzero_weight, zero_threshold, image_weight, station_weight;
};

__device__ __forceinline__ float target_at(const LossArgs& a, int b, int y, int x) {
    const float* t = a.target + (size_t)b * a.TH * a.TW;
    if (a.TH == a.H && a.TW == a.W) return t[y * a.W + x];
    const AxisW ay = axis_w(y, a.TH, a.H), ax = axis_w(x, a.TW, a.W);
    return ay.l0 * (ax.l0 * t[ay.i0 * a.TW + ax.i0] + ax.l1 * t[ay.i0 * a.TW + ax.i1]) +
           ay.l1 * (ax.l0 * t[ay.i1 * a.TW + ax.i0] + ax.l1 * t[ay.i1 * a.TW + ax.i1]);
}

// window mean of station s (wave-cooperative); returns count of window pixels in n_out
__device__ __forceinline__ float station_mean(const LossArgs& a, int b, int px, int py, int lane, int* n_out,
                                              int* x0o, int* y0o, int* wwo) {
    const int half = a.ksize / 2;
    const int y0 = max(0, py - half), y1 = min(a.H, py + half + 1);
    const int x0 = max(0, px - half), x1 = min(a.W, px + half + 1);
    const int wh = y1 - y0, ww = x1 - x0;
    const int n = (wh > 0 && ww > 0) ? wh * ww : 0;
    float s = 0.f;
    for (int i = lane; i < n; i += 64) s += a.pred[((size_t)b * a.H + y0 + i / ww) * a.W + x0 + i % ww];
    s = wave_sum(s);
    *n_out = n; *x0o = x0; *y0o = y0; *wwo = ww;
    return n > 0 ? s / (float)n : 0.f;
}

__device__ __forceinline__ int sample_of(const int* offsets, int B, int s) {
    int b = 0;
    while (b + 1 < B && offsets[b + 1] <= s) ++b;
    return b;
}

// pass 1: ws[0] = sum w, ws[1] = sum w*pixel_loss, ws[2] = sum station loss terms, ws[3] = #stations
__global__ void loss_reduce_kernel(LossArgs a, double* __restrict__ ws, int pixel_blocks) {
    __shared__ double smd[8];
    if ((int)blockIdx.x < pixel_blocks) {
        const long long total = (long long)a.B * a.H * a.W;
        double sw = 0.0, sl = 0.0;
        for (long long o = blockIdx.x * (long long)NT + threadIdx.x; o < total; o += (long long)pixel_blocks * NT) {
            const int x = (int)(o % a.W);
            const long long r = o / a.W;
            const int y = (int)(r % a.H);
            const int b = (int)(r / a.H);
            const float t = target_at(a, b, y, x);
            const float w = 1.0f + (a.nonzero_weight - 1.0f) * (fabsf(t) > a.zero_threshold ? 1.0f : 0.0f);
            const float d = a.pred[o] - t;
            const float pl = a.image_metric == 0 ? d * d : fabsf(d);
            sw += (double)w;
            sl += (double)(pl * w);
        }
        sw = block_sum_256_d(sw, smd);
        sl = block_sum_256_d(sl, smd + 4);
        if (threadIdx.x == 0) { unsafeAtomicAdd(&ws[0], sw); unsafeAtomicAdd(&ws[1], sl); }
    } else {
        const int lane = threadIdx.x & 63;
        const int wave = ((int)blockIdx.x - pixel_blocks) * (NT / 64) + (threadIdx.x >> 6);
        const int nwaves = ((int)gridDim.x - pixel_blocks) * (NT / 64);
        double tl = 0.0, tc = 0.0;
        for (int s = wave; s < a.S; s += nwaves) {
            const int b = sample_of(a.offsets, a.B, s);
            int n, x0, y0, ww;
            const float m = station_mean(a, b, a.pos[2 * s], a.pos[2 * s + 1], lane, &n, &x0, &y0, &ww);
            if (n > 0) {
                const float d = m - a.runoff[s];
                tl += (double)(a.station_metric == 0 ? d * d : fabsf(d));
                tc += 1.0;
            }
        }
        if (lane == 0 && tc > 0.0) { unsafeAtomicAdd(&ws[2], tl); unsafeAtomicAdd(&ws[3], tc); }
    }
}

// pass 2: dpred = image_weight * d(image loss); thread 0 of block 0 writes the three losses
__global__ void loss_image_grad_kernel(LossArgs a, const double* __restrict__ ws, float* __restrict__ losses,
                                       float* __restrict__ dpred) {
    const double sw = ws[0];
    const float inv_sw = (float)(1.0 / sw);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const float img = (float)(ws[1] / sw);
        const float stn = ws[3] > 0.0 ? (float)(ws[2] / ws[3]) : 0.f;
        losses[1] = img;
        losses[2] = stn;
        losses[0] = a.image_weight * img + a.station_weight * stn;
    }
    const long long total = (long long)a.B * a.H * a.W;
    for (long long o = blockIdx.x * (long long)NT + threadIdx.x; o < total; o += (long long)gridDim.x * NT) {
        const int x = (int)(o % a.W);
        const long long r = o / a.W;
        const int y = (int)(r % a.H);
        const int b = (int)(r / a.H);
        const float t = target_at(a, b, y, x);
        const float w = 1.0f + (a.nonzero_weight - 1.0f) * (fabsf(t) > a.zero_threshold ? 1.0f : 0.0f);
        const float d = a.pred[o] - t;
        const float g = a.image_metric == 0 ? 2.0f * d : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
        dpred[o] = a.image_weight * g * w * inv_sw;
    }
}

// pass 3: scatter the station gradient (windows may overlap -> atomics)
__global__ void loss_station_grad_kernel(LossArgs a, const double* __restrict__ ws, float* __restrict__ dpred) {
    const int lane = threadIdx.x & 63;
    const int wave = (int)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    const int nwaves = (int)gridDim.x * (NT / 64);
    const float cnt = (float)ws[3];
    if (cnt <= 0.f) return;
    for (int s = wave; s < a.S; s += nwaves) {
        const int b = sample_of(a.offsets, a.B, s);
        int n, x0, y0, ww;
        const float m = station_mean(a, b, a.pos[2 * s], a.pos[2 * s + 1], lane, &n, &x0, &y0, &ww);
        if (n <= 0) continue;
        const float d = m - a.runoff[s];
        const float g = a.station_metric == 0 ? 2.0f * d : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
        const float coef = a.station_weight * g / ((float)n * cnt);
        for (int i = lane; i < n; i += 64)
            unsafeAtomicAdd(&dpred[((size_t)b * a.H + y0 + i / ww) * a.W + x0 + i % ww], coef);
    }
}

__global__ void bce_reduce_kernel(const float* __restrict__ x, long long n, float label_c,
                                  const float* __restrict__ labels, double* __restrict__ ws,
                                  float* __restrict__ dx) {
    __shared__ double smd[4];
    double s = 0.0;
    const float inv_n = 1.0f / (float)n;
    for (long long i = blockIdx.x * (long long)NT + threadIdx.x; i < n; i += (long long)gridDim.x * NT) {
        const float v = x[i];
        const float label = labels ? labels[i] : label_c;
        s += (double)(fmaxf(v, 0.f) - v * label + log1pf(expf(-fabsf(v))));
        if (dx) dx[i] = (1.0f / (1.0f + expf(-v)) - label) * inv_n;
    }
    s = block_sum_256_d(s, smd);
    if (threadIdx.x == 0) unsafeAtomicAdd(&ws[0], s);
}
__global__ void bce_finish_kernel(const double* ws, long long n, float* loss) { loss[0] = (float)(ws[0] / (double)n); }

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long long n, float lr, float b1, float b2, float eps, float bc1,
                            float bc2, float gs) {
    const float step = lr / bc1;
    const float inv_sqrt_bc2 = 1.0f / sqrtf(bc2);
    for (long long i = blockIdx.x * (long long)NT + threadIdx.x; i < n; i += (long long)gridDim.x * NT) {
        const float gi = g[i] * gs;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= step * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
    }
}

__global__ void colsum_f32_kernel(const float* __restrict__ x, long long M, long long C, float* __restrict__ out) {
    for (long long c = blockIdx.x * (long long)NT + threadIdx.x; c < C; c += (long long)gridDim.x * NT) {
        float a = 0.f;
        for (long long m = 0; m < M; ++m) a += x[m * C + c];
        out[c] = a;
    }
}

__global__ void nan_inf_kernel(const float* __restrict__ x, long long n, int* flag) {
    bool bad = false;
    for (long long i = blockIdx.x * (long long)NT + threadIdx.x; i < n; i += (long long)gridDim.x * NT) {
        const float v = x[i];
        bad |= !(fabsf(v) <= 3.402823466e38f);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

__global__ void lonlat_kernel(const double* __restrict__ lon, const double* __restrict__ lat, int n, double lon_min,
                              double lon_max, double lat_min, double lat_max, int W, int H, int* __restrict__ px,
                              int* __restrict__ py) {
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= n) return;
    // same operations, in the same order, as the reference; rint() rounds half to even
    const double lon_span = fmax(1e-9, lon_max - lon_min);
    const double lat_span = fmax(1e-9, lat_max - lat_min);
    const double xr = (lon[i] - lon_min) / lon_span;
    const double yr = (lat_max - lat[i]) / lat_span;
    const double fx = rint(xr * (double)(W - 1));
    const double fy = rint(yr * (double)(H - 1));
    int ix = fx < -2147483000.0 ? -2147483000 : (fx > 2147483000.0 ? 2147483000 : (int)fx);
    int iy = fy < -2147483000.0 ? -2147483000 : (fy > 2147483000.0 ? 2147483000 : (int)fy);
    px[i] = max(0, min(ix, W - 1));
    py[i] = max(0, min(iy, H - 1));
}

}  // namespace

extern "C" int sf_image_station_loss(const float* pred, const float* target, const int* pos, const float* runoff,
                                     const int* offsets, int S, int B, int H, int W, int TH, int TW, int image_metric,
                                     float nonzero_weight, float zero_threshold, int ksize, int station_metric,
                                     float image_weight, float station_weight, float* losses, float* dpred, double* ws,
                                     void* stream) {
    SF_REQUIRE(pred && target && losses && dpred && ws, "sf_image_station_loss: null argument");
    SF_REQUIRE(B > 0 && H > 0 && W > 0 && TH > 0 && TW > 0 && S >= 0, "sf_image_station_loss: bad shape");
    SF_REQUIRE(S == 0 || (pos && runoff && offsets), "sf_image_station_loss: stations without arrays");
    SF_REQUIRE(ksize >= 1, "sf_image_station_loss: kernel_size %d", ksize);
    SF_REQUIRE((image_metric == 0 || image_metric == 1) && (station_metric == 0 || station_metric == 1), "sf_image_station_loss: metric");
    hipStream_t s = (hipStream_t)stream;
    LossArgs a;
    a.pred = pred; a.target = target; a.pos = pos; a.runoff = runoff; a.offsets = offsets;
    a.S = S; a.B = B; a.H = H; a.W = W; a.TH = TH; a.TW = TW; a.image_metric = image_metric; a.ksize = ksize;
    a.station_metric = station_metric; a.nonzero_weight = nonzero_weight; a.zero_threshold = zero_threshold;
    a.image_weight = image_weight; a.station_weight = station_weight;
    if (hipMemsetAsync(ws, 0, sizeof(double) * 4, s) != hipSuccess) { sf_set_error("sf_image_station_loss: memset failed"); return SF_ELAUNCH; }
    const long long total = (long long)B * H * W;
    int pblocks = (int)((total + NT * 8 - 1) / (NT * 8)); if (pblocks > 2048) pblocks = 2048; if (pblocks < 1) pblocks = 1;
    int sblocks = S > 0 ? (S + 3) / 4 : 0; if (sblocks > 1024) sblocks = 1024;
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(pblocks + sblocks), dim3(NT), 0, s, a, ws, pblocks);
    hipLaunchKernelGGL(loss_image_grad_kernel, dim3(pblocks), dim3(NT), 0, s, a, ws, losses, dpred);
    if (sblocks > 0) hipLaunchKernelGGL(loss_station_grad_kernel, dim3(sblocks), dim3(NT), 0, s, a, ws, dpred);
    SF_LAUNCH_CHECK("sf_image_station_loss");
    return SF_OK;
}

extern "C" int sf_bce_logits(const float* x, long long n, float label, const float* labels, float* loss, float* dx,
                             double* ws, void* stream) {
    SF_REQUIRE(x && loss && ws && n > 0, "sf_bce_logits: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(ws, 0, sizeof(double), s) != hipSuccess) { sf_set_error("sf_bce_logits: memset failed"); return SF_ELAUNCH; }
    int grid = (int)((n + NT - 1) / NT); if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(bce_reduce_kernel, dim3(grid), dim3(NT), 0, s, x, n, label, labels, ws, dx);
    hipLaunchKernelGGL(bce_finish_kernel, dim3(1), dim3(1), 0, s, ws, n, loss);
    SF_LAUNCH_CHECK("sf_bce_logits");
    return SF_OK;
}

extern "C" int sf_adam(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                       float eps, float bc1, float bc2, float grad_scale, void* stream) {
    SF_REQUIRE(p && g && m && v && n > 0, "sf_adam: bad arguments");
    SF_REQUIRE(bc1 > 0.f && bc2 > 0.f, "sf_adam: bias corrections must be positive");
    int grid = (int)((n + NT - 1) / NT); if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(NT), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps,
                       bc1, bc2, grad_scale);
    SF_LAUNCH_CHECK("sf_adam");
    return SF_OK;
}

extern "C" int sf_colsum_f32(const float* x, long long M, long long C, float* out, void* stream) {
    SF_REQUIRE(x && out && M > 0 && C > 0, "sf_colsum_f32: bad arguments");
    long long grid = (C + NT - 1) / NT; if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(colsum_f32_kernel, dim3((unsigned)grid), dim3(NT), 0, (hipStream_t)stream, x, M, C, out);
    SF_LAUNCH_CHECK("sf_colsum_f32");
    return SF_OK;
}

extern "C" int sf_nan_inf_flag(const float* x, long long n, int* flag, void* stream) {
    SF_REQUIRE(x && flag && n > 0, "sf_nan_inf_flag: bad arguments");
    int grid = (int)((n + NT - 1) / NT); if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(nan_inf_kernel, dim3(grid), dim3(NT), 0, (hipStream_t)stream, x, n, flag);
    SF_LAUNCH_CHECK("sf_nan_inf_flag");
    return SF_OK;
}

extern "C" int sf_lonlat_to_pixel(const double* lon, const double* lat, int n, double lon_min, double lon_max,
                                  double lat_min, double lat_max, int W, int H, int* px, int* py, void* stream) {
    SF_REQUIRE(lon && lat && px && py && n > 0 && W > 0 && H > 0, "sf_lonlat_to_pixel: bad arguments");
    hipLaunchKernelGGL(lonlat_kernel, dim3((n + NT - 1) / NT), dim3(NT), 0, (hipStream_t)stream, lon, lat, n, lon_min,
                       lon_max, lat_min, lat_max, W, H, px, py);
    SF_LAUNCH_CHECK("sf_lonlat_to_pixel");
    return SF_OK;
}
