// Per-date image metrics of the evaluation loop (src/evaluate.py:17-86,205-238) on the device:
// for every date of a batch the error sums, the first and second moments of prediction and
// target (correlation, PSNR), their ranges, and the mean SSIM with the reference's 11 x 11
// uniform window and scipy's 'reflect' border (d c b a | a b c d | d c b a).
//
// Three launches, no host sync: (1) per-date moment partials, (2) SSIM over 32 x 32 tiles (each
// block first folds the partials of its date to get the data range), (3) a fold of all partials
// into out[B][13].  Sums are float64 in a fixed order: the results do not depend on timing.
// Each pass reads both maps once: HBM bound.
#include "sf_common.h"

#define MT_NT 256
#define MT_PARTS 16   // moment partials per date
#define MT_NQ 11      // sse, sae, sum p, sum t, sum pp, sum tt, sum pt, min p, max p, min t, max t
#define MT_TILE 32
#define MT_HALO 5
#define MT_PATCH (MT_TILE + 2 * MT_HALO)
#define MT_OUT 13     // the 11 above, the SSIM sum, the data range

__device__ __forceinline__ double block_fold(double v, double* sm, int op /*0 sum, 1 min, 2 max*/) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double w = __shfl_xor(v, o, 64);
        v = op == 0 ? v + w : (op == 1 ? fmin(v, w) : fmax(v, w));
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    if (op == 0) return (sm[0] + sm[1]) + (sm[2] + sm[3]);
    if (op == 1) return fmin(fmin(sm[0], sm[1]), fmin(sm[2], sm[3]));
    return fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3]));
}

// grid (MT_PARTS, B): partial p of date b covers pixels [p*chunk, (p+1)*chunk)
__global__ __launch_bounds__(MT_NT) void date_moments_kernel(const float* __restrict__ pred,
                                                             const float* __restrict__ target, long long HW,
                                                             double* __restrict__ part) {
    __shared__ double sm[4];
    const int b = blockIdx.y;
    const long long chunk = (HW + MT_PARTS - 1) / MT_PARTS;
    const long long lo = blockIdx.x * chunk, hi = lo + chunk < HW ? lo + chunk : HW;
    const float* p = pred + (long long)b * HW;
    const float* t = target + (long long)b * HW;
    double q[MT_NQ] = {0, 0, 0, 0, 0, 0, 0, INFINITY, -INFINITY, INFINITY, -INFINITY};
    for (long long i = lo + threadIdx.x; i < hi; i += MT_NT) {
        const double x = p[i], y = t[i], e = x - y;
        q[0] += e * e;
        q[1] += fabs(e);
        q[2] += x;
        q[3] += y;
        q[4] += x * x;
        q[5] += y * y;
        q[6] += x * y;
        q[7] = fmin(q[7], x);
        q[8] = fmax(q[8], x);
        q[9] = fmin(q[9], y);
        q[10] = fmax(q[10], y);
    }
    double* o = part + ((long long)b * MT_PARTS + blockIdx.x) * MT_NQ;
#pragma unroll
    for (int k = 0; k < MT_NQ; ++k) {
        const double v = block_fold(q[k], sm, k < 7 ? 0 : ((k & 1) ? 1 : 2));
        if (threadIdx.x == 0) o[k] = v;
    }
}

__device__ __forceinline__ int reflect(int i, int n) {
    // scipy.ndimage 'reflect': the edge sample is repeated; loops for maps narrower than the halo
    while (i < 0 || i >= n) i = i < 0 ? -i - 1 : 2 * n - 1 - i;
    return i;
}

// data range of date b as the reference takes it: max(t.max - t.min, p.max - p.min) on float32
__device__ __forceinline__ float date_range(const double* part_b) {
    double pmin = INFINITY, pmax = -INFINITY, tmin = INFINITY, tmax = -INFINITY;
    for (int k = 0; k < MT_PARTS; ++k) {
        const double* q = part_b + k * MT_NQ;
        pmin = fmin(pmin, q[7]);
        pmax = fmax(pmax, q[8]);
        tmin = fmin(tmin, q[9]);
        tmax = fmax(tmax, q[10]);
    }
    return fmaxf((float)tmax - (float)tmin, (float)pmax - (float)pmin);
}

// grid (tiles_x, tiles_y, B).  ssim_part[b][ty * tiles_x + tx] = sum of the SSIM map over the tile.
__global__ __launch_bounds__(MT_NT) void date_ssim_kernel(const float* __restrict__ pred,
                                                          const float* __restrict__ target, int H, int W,
                                                          const double* __restrict__ part,
                                                          double* __restrict__ ssim_part) {
    __shared__ float sp[MT_PATCH][MT_PATCH + 1], st[MT_PATCH][MT_PATCH + 1];
    __shared__ double hs[5][MT_PATCH][MT_TILE + 1];   // horizontal window sums of p, t, pp, tt, pt
    __shared__ double sm[4];
    const int b = blockIdx.z, x0 = blockIdx.x * MT_TILE, y0 = blockIdx.y * MT_TILE;
    const float* p = pred + (long long)b * H * W;
    const float* t = target + (long long)b * H * W;
    const double range = (double)date_range(part + (long long)b * MT_PARTS * MT_NQ);
    const double c1 = (0.01 * range) * (0.01 * range), c2 = (0.03 * range) * (0.03 * range);
    for (int i = threadIdx.x; i < MT_PATCH * MT_PATCH; i += MT_NT) {
        const int r = i / MT_PATCH, c = i % MT_PATCH;
        const long long src = (long long)reflect(y0 + r - MT_HALO, H) * W + reflect(x0 + c - MT_HALO, W);
        sp[r][c] = p[src];
        st[r][c] = t[src];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < MT_PATCH * MT_TILE; i += MT_NT) {
        const int r = i / MT_TILE, c = i % MT_TILE;
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
#pragma unroll
        for (int k = 0; k < 2 * MT_HALO + 1; ++k) {
            const double x = sp[r][c + k], y = st[r][c + k];
            a0 += x;
            a1 += y;
            a2 += x * x;
            a3 += y * y;
            a4 += x * y;
        }
        hs[0][r][c] = a0;
        hs[1][r][c] = a1;
        hs[2][r][c] = a2;
        hs[3][r][c] = a3;
        hs[4][r][c] = a4;
    }
    __syncthreads();
    double acc = 0.0;
    const double inv = 1.0 / 121.0;
    for (int i = threadIdx.x; i < MT_TILE * MT_TILE; i += MT_NT) {
        const int r = i / MT_TILE, c = i % MT_TILE;
        if (y0 + r >= H || x0 + c >= W) continue;
        double m[5] = {0, 0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 2 * MT_HALO + 1; ++k) {
#pragma unroll
            for (int j = 0; j < 5; ++j) m[j] += hs[j][r + k][c];
        }
        const double mp = m[0] * inv, mt = m[1] * inv;
        const double vp = m[2] * inv - mp * mp, vt = m[3] * inv - mt * mt, cov = m[4] * inv - mp * mt;
        acc += ((2.0 * mp * mt + c1) * (2.0 * cov + c2)) / ((mp * mp + mt * mt + c1) * (vp + vt + c2));
    }
    acc = block_fold(acc, sm, 0);
    if (threadIdx.x == 0)
        ssim_part[((long long)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = acc;
}

// grid (B): out[b] = folded moments, SSIM sum (tiles in index order), data range
__global__ __launch_bounds__(64) void date_metrics_finalize_kernel(const double* __restrict__ part,
                                                                   const double* __restrict__ ssim_part, int ntiles,
                                                                   double* __restrict__ out) {
    const int b = blockIdx.x;
    const double* pb = part + (long long)b * MT_PARTS * MT_NQ;
    if (threadIdx.x < MT_NQ) {
        const int k = threadIdx.x;
        double v = pb[k];
        for (int j = 1; j < MT_PARTS; ++j) {
            const double w = pb[j * MT_NQ + k];
            v = k < 7 ? v + w : ((k & 1) ? fmin(v, w) : fmax(v, w));
        }
        out[(long long)b * MT_OUT + k] = v;
    } else if (threadIdx.x == MT_NQ) {
        double s = 0.0;
        for (int j = 0; j < ntiles; ++j) s += ssim_part[(long long)b * ntiles + j];
        out[(long long)b * MT_OUT + 11] = s;
    } else if (threadIdx.x == MT_NQ + 1) {
        out[(long long)b * MT_OUT + 12] = (double)date_range(pb);
    }
}

static long long metrics_tiles(int H, int W) {
    return (long long)((H + MT_TILE - 1) / MT_TILE) * ((W + MT_TILE - 1) / MT_TILE);
}

extern "C" size_t sf_date_image_metrics_workspace_bytes(int B, int H, int W) {
    if (B < 1 || H < 1 || W < 1) return 0;
    return sizeof(double) * (size_t)B * (size_t)(MT_PARTS * MT_NQ + metrics_tiles(H, W));
}

extern "C" int sf_date_image_metrics(const float* pred, const float* target, int B, int H, int W, double* out, void* ws,
                                     size_t ws_bytes, void* stream) {
    SF_REQUIRE(pred && target && out && ws, "sf_date_image_metrics: null buffer");
    SF_REQUIRE(B >= 1 && B <= 65535 && H >= 1 && W >= 1, "sf_date_image_metrics: bad shape B=%d H=%d W=%d", B, H, W);
    const int tx = (W + MT_TILE - 1) / MT_TILE, ty = (H + MT_TILE - 1) / MT_TILE;
    SF_REQUIRE(ty <= 65535, "sf_date_image_metrics: map too tall");
    SF_REQUIRE(ws_bytes >= sf_date_image_metrics_workspace_bytes(B, H, W), "sf_date_image_metrics: workspace of %zu bytes, need %zu",
               ws_bytes, sf_date_image_metrics_workspace_bytes(B, H, W));
    double* part = (double*)ws;
    double* ssim_part = part + (size_t)B * MT_PARTS * MT_NQ;
    hipStream_t s = (hipStream_t)stream;
    date_moments_kernel<<<dim3(MT_PARTS, B), MT_NT, 0, s>>>(pred, target, (long long)H * W, part);
    SF_LAUNCH_CHECK("sf_date_image_metrics (moments)");
    date_ssim_kernel<<<dim3(tx, ty, B), MT_NT, 0, s>>>(pred, target, H, W, part, ssim_part);
    SF_LAUNCH_CHECK("sf_date_image_metrics (ssim)");
    date_metrics_finalize_kernel<<<B, 64, 0, s>>>(part, ssim_part, tx * ty, out);
    SF_LAUNCH_CHECK("sf_date_image_metrics (finalize)");
    return SF_OK;
}
