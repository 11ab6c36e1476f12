// Device-resident dataset: the per-date arithmetic of StreamflowDataset (src/dataset.py) done once
// over all frames held in HBM ([D,C,H*W] float32), and the window gather that serves batches.
//
//   sf_dataset_plane_stats : per (date, channel) sum / sum of squares / min / max over the selected
//                            pixels (src/dataset.py:219-264), one block per plane
//   sf_dataset_transform   : mask, NaN/Inf -> 0, flow threshold on channel 0, normalisation in
//                            float64 (src/dataset.py:412-491), in place
//   sf_gather_frames       : dst[i] = src[rows[i]] for whole frames (the stacking of
//                            src/dataset.py:438 and of the collate function, src/train.py:381-382)
//
// All three are streaming passes bounded by HBM bandwidth.
#include "sf_common.h"
#include <float.h>

#define DS_NT 256

__device__ __forceinline__ double block_min_d(double v, double* sm) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmin(fmin(sm[0], sm[1]), fmin(sm[2], sm[3]));
}

constexpr int GRID_YZ_MAX = 65535;   // blocks in a grid dimension other than x

// grid (C, D).  out[d][c] = {sum, sumsq, min, max} over the selected pixels of plane (d, c);
// count[d] = number of selected pixels; bad0[d] = non-finite values in channel 0 (all pixels).
// Selection: mask > 0.5 where the date has a mask (mask_row[d] >= 0), every pixel otherwise.
// Values pass through numpy's nan_to_num defaults first (NaN -> 0, +-Inf -> +-FLT_MAX), the square
// is taken in float32 like `vals ** 2` on a float32 array, the accumulation is float64.
__global__ __launch_bounds__(DS_NT) void dataset_plane_stats_kernel(
    const float* __restrict__ frames, const float* __restrict__ masks, const int* __restrict__ mask_row, int C,
    long long HW, double* __restrict__ out, long long* __restrict__ count, int* __restrict__ bad0) {
    __shared__ double sm[4];
    const int c = blockIdx.x, d = blockIdx.y;
    const float* x = frames + ((long long)d * C + c) * HW;
    const int mr = mask_row ? mask_row[d] : -1;
    const float* m = mr >= 0 ? masks + (long long)mr * HW : nullptr;
    double s = 0.0, ss = 0.0, mn = INFINITY, mx = -INFINITY, n = 0.0, bad = 0.0;
    for (long long i = threadIdx.x; i < HW; i += DS_NT) {
        float v = x[i];
        const bool sel = m ? m[i] > 0.5f : true;
        if (!isfinite(v)) {
            bad += 1.0;
            v = isnan(v) ? 0.f : (v > 0.f ? FLT_MAX : -FLT_MAX);
        }
        if (sel) {
            const float sq = v * v;
            s += (double)v;
            ss += (double)sq;
            mn = fmin(mn, (double)v);
            mx = fmax(mx, (double)v);
            n += 1.0;
        }
    }
    s = block_sum_256_d(s, sm);
    ss = block_sum_256_d(ss, sm);
    n = block_sum_256_d(n, sm);
    bad = block_sum_256_d(bad, sm);
    mn = block_min_d(mn, sm);
    mx = -block_min_d(-mx, sm);
    if (threadIdx.x == 0) {
        double* o = out + ((long long)d * C + c) * 4;
        o[0] = s;
        o[1] = ss;
        o[2] = mn;
        o[3] = mx;
        if (c == 0) {
            count[d] = (long long)n;
            bad0[d] = (int)bad;
        }
    }
}

// One thread per pixel of a plane, grid (ceil(HW / DS_NT), C, D).  In place on frames; when
// `target` is given, target[d] receives channel 0 processed the way the reference processes its
// output image: mask and threshold but no NaN replacement (src/dataset.py:441-472).
// normalize: 0 none, 1 (v - a[c]) / b[c] in float64, rounded to float32 once.
__global__ __launch_bounds__(DS_NT) void dataset_transform_kernel(
    float* __restrict__ frames, float* __restrict__ target, const float* __restrict__ masks,
    const int* __restrict__ mask_row, const double* __restrict__ a, const double* __restrict__ b, int normalize,
    float thr, int C, long long HW) {
    const long long i = (long long)blockIdx.x * DS_NT + threadIdx.x;
    if (i >= HW) return;
    const int c = blockIdx.y, d = blockIdx.z;
    float* x = frames + ((long long)d * C + c) * HW;
    const int mr = mask_row ? mask_row[d] : -1;
    float v = x[i];
    if (mr >= 0) v *= masks[(long long)mr * HW + i];
    float t = v;
    if (!isfinite(v)) v = 0.f;
    if (c == 0 && fabsf(v) < thr) v = 0.f;
    if (normalize) v = (float)(((double)v - a[c]) / b[c]);
    x[i] = v;
    if (c == 0 && target) {
        if (fabsf(t) < thr) t = 0.f;
        if (normalize) t = (float)(((double)t - a[0]) / b[0]);
        target[(long long)d * HW + i] = t;
    }
}

// grid (chunks, n): frame i of dst (n_elems floats, contiguous) = frame rows[i] of src (frames
// src_stride floats apart).  V = float4 when frame sizes and bases allow 16-byte accesses, float
// otherwise; n_v = elements of V per frame.
template <typename V>
__global__ __launch_bounds__(DS_NT) void gather_frames_kernel(const float* __restrict__ src, long long src_stride,
                                                              long long n_src, const long long* __restrict__ rows,
                                                              long long n_elems, long long n_v,
                                                              float* __restrict__ dst) {
    const long long r = rows[blockIdx.y];
    if (r < 0 || r >= n_src) return;   // a row outside the source leaves its frame untouched
    const V* s = reinterpret_cast<const V*>(src + r * src_stride);
    V* o = reinterpret_cast<V*>(dst + (long long)blockIdx.y * n_elems);
    for (long long i = (long long)blockIdx.x * DS_NT + threadIdx.x; i < n_v; i += (long long)gridDim.x * DS_NT)
        o[i] = s[i];
}

extern "C" int sf_dataset_plane_stats(const float* frames, const float* masks, const int* mask_row, int D, int C,
                                      long long HW, double* out, long long* count, int* bad0, void* stream) {
    SF_REQUIRE(frames && out && count && bad0, "sf_dataset_plane_stats: null buffer");
    SF_REQUIRE(D >= 1 && C >= 1 && HW >= 1, "sf_dataset_plane_stats: bad shape D=%d C=%d HW=%lld", D, C, HW);
    SF_REQUIRE((masks != nullptr) == (mask_row != nullptr), "sf_dataset_plane_stats: masks and mask_row go together");
    // a grid dimension other than x holds at most 65535 blocks: dates go in launches of that many
    for (int d0 = 0; d0 < D; d0 += GRID_YZ_MAX) {
        const int nd = D - d0 < GRID_YZ_MAX ? D - d0 : GRID_YZ_MAX;
        dataset_plane_stats_kernel<<<dim3(C, nd), DS_NT, 0, (hipStream_t)stream>>>(
            frames + (long long)d0 * C * HW, masks, mask_row ? mask_row + d0 : nullptr, C, HW,
            out + (long long)d0 * C * 4, count + d0, bad0 + d0);
        SF_LAUNCH_CHECK("sf_dataset_plane_stats");
    }
    return SF_OK;
}

extern "C" int sf_dataset_transform(float* frames, float* target, const float* masks, const int* mask_row,
                                    const double* a, const double* b, int normalize, float flow_threshold, int D,
                                    int C, long long HW, void* stream) {
    SF_REQUIRE(frames, "sf_dataset_transform: null frames");
    SF_REQUIRE(D >= 1 && C >= 1 && C <= GRID_YZ_MAX && HW >= 1, "sf_dataset_transform: bad shape D=%d C=%d HW=%lld",
               D, C, HW);
    SF_REQUIRE(!normalize || (a && b), "sf_dataset_transform: normalisation needs both coefficient arrays");
    SF_REQUIRE((masks != nullptr) == (mask_row != nullptr), "sf_dataset_transform: masks and mask_row go together");
    const long long gx = (HW + DS_NT - 1) / DS_NT;
    SF_REQUIRE(gx <= 0x7fffffffLL, "sf_dataset_transform: plane too large");
    for (int d0 = 0; d0 < D; d0 += GRID_YZ_MAX) {
        const int nd = D - d0 < GRID_YZ_MAX ? D - d0 : GRID_YZ_MAX;
        dataset_transform_kernel<<<dim3((unsigned)gx, C, nd), DS_NT, 0, (hipStream_t)stream>>>(
            frames + (long long)d0 * C * HW, target ? target + (long long)d0 * HW : nullptr, masks,
            mask_row ? mask_row + d0 : nullptr, a, b, normalize, flow_threshold, C, HW);
        SF_LAUNCH_CHECK("sf_dataset_transform");
    }
    return SF_OK;
}

extern "C" int sf_gather_frames(const float* src, long long src_stride, long long n_src, const long long* rows, int n,
                                long long n_elems, float* dst, void* stream) {
    SF_REQUIRE(src && rows && dst, "sf_gather_frames: null buffer");
    SF_REQUIRE(n >= 1, "sf_gather_frames: n=%d out of range", n);
    SF_REQUIRE(n_elems >= 1 && n_elems <= src_stride && n_src >= 1,
               "sf_gather_frames: frame of %lld floats, stride %lld, %lld source frames", n_elems, src_stride, n_src);
    const bool vec = n_elems % 4 == 0 && src_stride % 4 == 0 && ((uintptr_t)src | (uintptr_t)dst) % 16 == 0;
    const long long n_v = vec ? n_elems / 4 : n_elems;
    long long chunks = (n_v + DS_NT * 4 - 1) / (DS_NT * 4);   // ~4 accesses per thread
    if (chunks > 1024) chunks = 1024;
    for (int i0 = 0; i0 < n; i0 += GRID_YZ_MAX) {
        const int ni = n - i0 < GRID_YZ_MAX ? n - i0 : GRID_YZ_MAX;
        if (vec)
            gather_frames_kernel<float4><<<dim3((unsigned)chunks, ni), DS_NT, 0, (hipStream_t)stream>>>(
                src, src_stride, n_src, rows + i0, n_elems, n_v, dst + (long long)i0 * n_elems);
        else
            gather_frames_kernel<float><<<dim3((unsigned)chunks, ni), DS_NT, 0, (hipStream_t)stream>>>(
                src, src_stride, n_src, rows + i0, n_elems, n_v, dst + (long long)i0 * n_elems);
        SF_LAUNCH_CHECK("sf_gather_frames");
    }
    return SF_OK;
}
