// Losses, optimiser and small utilities.
//  * image + station loss: forward and backward in one stream-ordered sequence with no host
//    synchronisation; the station term is a gather-reduce with one wavefront per station
//    (lanes over the k x k window, shuffle reduction) and a scatter of its gradient.
//  * BCE-with-logits against a constant label (adversarial terms).
//  * fused Adam, NaN/Inf flag, lon/lat -> pixel map.
#include "sf_common.h"
#include "../../include/sfhip.h"
#include <mutex>

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

// sample whose station range [offsets[b], offsets[b+1]) holds s (empty samples are skipped)
__device__ __forceinline__ int sample_of(const int* offsets, int B, int s) {
    int lo = 0, hi = B - 1;            // invariant: offsets[lo] <= s, answer in [lo, hi]
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (offsets[mid] <= s) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// Workspace layout (doubles): per-block partial sums, written in full by every call and folded in
// a fixed order by whichever block finishes last, so the losses are bit-reproducible; the folded
// scalars the backward pass needs; the arrival counter (zero between calls).
constexpr int LP = 896;                   // max pixel blocks
constexpr int LS = 896;                   // max station blocks
constexpr int WS_W = 0;                   // [LP] sum of pixel weights
constexpr int WS_L = LP;                  // [LP] sum of weight * pixel loss
constexpr int WS_ST = 2 * LP;             // [LS] station loss terms
constexpr int WS_SC = 2 * LP + LS;        // [LS] stations with a non-empty window
constexpr int WS_SCAL = 2 * LP + 2 * LS;  // two floats: 1 / sum of pixel weights, 1 / stations counted (forward + backward in one call)
// Arrival counters (ints, zero between calls).  A few thousand fetch-adds on ONE address take tens
// of microseconds (they are served one after the other), so blocks arrive in groups of ARRIVE_GROUP
// on a counter of their own 128-byte line, and only the last block of each group goes on to the
// top counter; whoever is last there folds.  Every last arriver puts its counter back to zero.
constexpr int ARRIVE_GROUP = 64;
constexpr int WS_GROUPS = WS_SCAL + 16;   // group g: the int at double WS_GROUPS + 16 g
constexpr int MAX_GROUPS = (LP + LS + ARRIVE_GROUP - 1) / ARRIVE_GROUP;
constexpr int WS_ARRIVED = SF_LOSS_WS_DOUBLES - 1;   // top counter; the one slot a caller's workspace has to hold zero in
static_assert(LS == LP, "the fold walks both kinds of partials with one index");
static_assert(WS_GROUPS + 16 * MAX_GROUPS <= WS_ARRIVED, "loss workspace layout");
static_assert(SF_LOSS_WS_DOUBLES - WS_GROUPS == SF_LOSS_WS_COUNTER_DOUBLES, "the counter region the header promises");

__device__ __forceinline__ float pixel_weight(const LossArgs& a, float t) {
    return 1.0f + (a.nonzero_weight - 1.0f) * (fabsf(t) > a.zero_threshold ? 1.0f : 0.0f);
}

// a partial sum on its way to another workgroup of the same launch: written through at agent scope
__device__ __forceinline__ void publish(double* slot, double v) {
    __hip_atomic_store(slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The pixel passes walk the prediction grid by rows: a wave takes pairs of adjacent rows (b, y),
// pair = wave, wave + nwaves, ... and its lanes take V consecutive pixels each of up to two column
// groups, so the horizontal resize taps of a lane are computed once and kept for all its rows, the
// vertical taps once per row, and all index arithmetic is 32-bit.
template <int V> struct ColTaps { int i0[V], i1[V]; float l0[V], l1[V]; };

template <int V>
__device__ __forceinline__ ColTaps<V> col_taps(const LossArgs& a, int x) {
    ColTaps<V> c;
#pragma unroll
    for (int e = 0; e < V; ++e) {
        const AxisW ax = axis_w(x + e, a.TW, a.W);
        c.i0[e] = ax.i0; c.i1[e] = ax.i1; c.l0[e] = ax.l0; c.l1[e] = ax.l1;
    }
    return c;
}

// per-lane view of one station for the backward pass: window and gradient coefficient
struct StationLane { int x0, x1, y0, y1; float coef; };

// One pass over the prediction grid, forward (BWD = false) or backward (BWD = true).
//  forward: adds the block's part of sum w to accw and of sum w * pixel loss to accl, and notes
//           in `bad` whether a prediction value is Inf / NaN; writes nothing.
//  backward: writes d total / d pred of every pixel, complete: gimg * w * d(pixel loss)/d(pred)
//           plus, for every station whose window covers the pixel, that station's coefficient
//           (added in registers, in station order, before the one store of the pixel: no atomics,
//           no second kernel, bit-reproducible).
// A wave works on two adjacent prediction rows at a time: 4 items (2 rows x 2 column groups)
// whose loads are all issued first, from clamped (always valid) addresses; results of items outside
// the grid are dropped afterwards -- no load sits behind a branch.  When the target is smaller
// than the prediction and its rows fit (TW <= SROW), the wave first copies the four target rows
// the two prediction rows interpolate between into its own LDS slice with coalesced loads and
// takes the resize taps from there: per-lane gathers from global memory cost an instruction per
// tap and bound the pass otherwise.
constexpr int SROW = 512;
template <int V, bool BWD>
__device__ __forceinline__ void pixel_pass(const LossArgs& a, int pixel_blocks, float gimg, float gstn,
                                           const float* __restrict__ sdiff, float* __restrict__ dpred,
                                           float& accw, float& accl, bool& bad) {
    __shared__ float srow[NT / 64][4][SROW];
    const bool same = a.TH == a.H && a.TW == a.W;
    const bool staged = !same && a.TW <= SROW;
    const int lane = threadIdx.x & 63, wl = threadIdx.x >> 6, wave = (int)blockIdx.x * (NT / 64) + wl;
    const int nwaves = pixel_blocks * (NT / 64), rows = a.B * a.H, wv = a.W / V;
    const int npairs = (rows + 1) >> 1;
    const int half = a.ksize / 2;
    for (int xb = 0; xb < wv; xb += 128) {
        int xv[2];
        bool xin[2];
        ColTaps<V> c[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            xv[i] = xb + 64 * i + lane;
            xin[i] = xv[i] < wv;
            if (!xin[i]) xv[i] = wv - 1;
            c[i] = col_taps<V>(a, xv[i] * V);
        }
        for (int pr = wave; pr < npairs; pr += nwaves) {
            float t[4][V], pv[4][V];
            bool ok[4];
            size_t off[4];
            AxisW ay[2];
            const float* tp[2];
            int yy[2], bb[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int rr = 2 * pr + j;
                const int r = rr < rows ? rr : rows - 1;
                const int b = r / a.H;
                bb[j] = b;
                yy[j] = r - b * a.H;
                tp[j] = a.target + (size_t)b * a.TH * a.TW;
                ay[j] = axis_w(yy[j], a.TH, a.H);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int u = 2 * j + i;
                    ok[u] = rr < rows && xin[i];
                    off[u] = (size_t)r * a.W + xv[i] * V;
                    if constexpr (V == 4) {
                        const float4 v = *reinterpret_cast<const float4*>(a.pred + off[u]);
                        pv[u][0] = v.x; pv[u][1] = v.y; pv[u][2] = v.z; pv[u][3] = v.w;
                    } else {
                        pv[u][0] = a.pred[off[u]];
                    }
                }
            }
            // backward: the first 64 stations of each row's sample are requested here, beside the
            // prediction loads, so that their two dependent round trips (offsets, then positions
            // and differences) are over by the time the gradient is formed
            int st_s0[2] = {0, 0}, st_s1[2] = {0, 0}, st_px[2] = {0, 0}, st_py[2] = {0, 0};
            float st_d[2] = {0.f, 0.f};
            if (BWD && a.S > 0 && gstn != 0.f) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int bj = __builtin_amdgcn_readfirstlane(bb[j]);
                    st_s0[j] = a.offsets[bj]; st_s1[j] = a.offsets[bj + 1];
                    const int s = st_s0[j] + lane;
                    const int sc = max(s < st_s1[j] ? s : st_s1[j] - 1, 0);
                    st_px[j] = a.pos[2 * sc]; st_py[j] = a.pos[2 * sc + 1];
                    st_d[j] = sdiff ? sdiff[sc] : 0.f;
                }
            }
            if (staged) {
                // rows ay[j].i0 / ay[j].i1 of the target -> srow[wl][2j], srow[wl][2j+1]
                __builtin_amdgcn_wave_barrier();          // the previous pair's reads are done (same wave, in order)
                for (int x = lane; x < a.TW; x += 64) {
                    float v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = tp[q >> 1][((q & 1) ? ay[q >> 1].i1 : ay[q >> 1].i0) * a.TW + x];
#pragma unroll
                    for (int q = 0; q < 4; ++q) srow[wl][q][x] = v[q];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int u = 2 * j + i;
                    if (same) {
                        if constexpr (V == 4) {
                            const float4 v = *reinterpret_cast<const float4*>(tp[j] + (size_t)yy[j] * a.W + xv[i] * V);
                            t[u][0] = v.x; t[u][1] = v.y; t[u][2] = v.z; t[u][3] = v.w;
                        } else {
                            t[u][0] = tp[j][(size_t)yy[j] * a.W + xv[i]];
                        }
                    } else {
                        // same expression as F.interpolate's bilinear kernel (align_corners=False)
                        auto taps = [&](const auto* q0, const auto* q1) __attribute__((always_inline)) {
#pragma unroll
                            for (int e = 0; e < V; ++e)
                                t[u][e] = ay[j].l0 * (c[i].l0[e] * q0[c[i].i0[e]] + c[i].l1[e] * q0[c[i].i1[e]]) +
                                          ay[j].l1 * (c[i].l0[e] * q1[c[i].i0[e]] + c[i].l1[e] * q1[c[i].i1[e]]);
                        };
                        if (staged) taps(&srow[wl][2 * j][0], &srow[wl][2 * j + 1][0]);   // LDS addresses stay LDS addresses
                        else taps(tp[j] + ay[j].i0 * a.TW, tp[j] + ay[j].i1 * a.TW);
                    }
                }
            float g[4][V];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float pw = 0.f, pl = 0.f;
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const float w = pixel_weight(a, t[u][e]);
                    const float d = pv[u][e] - t[u][e];
                    if (BWD) {
                        g[u][e] = gimg * w * (a.image_metric == 0 ? 2.0f * d : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)));
                    } else {
                        pw += w;
                        pl += (a.image_metric == 0 ? d * d : fabsf(d)) * w;
                        bad |= ok[u] && !(fabsf(pv[u][e]) <= 3.402823466e38f);
                    }
                }
                if (!BWD) {
                    accw += ok[u] ? pw : 0.f;
                    accl += ok[u] ? pl : 0.f;
                }
            }
            if (BWD) {
                // every station of the rows' samples whose window holds one of the two rows is handed
                // round the wave and added, in station order, to the pixels it covers
                if (a.S > 0 && gstn != 0.f) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        for (int base = st_s0[j]; base < st_s1[j]; base += 64) {
                            int px = st_px[j], py = st_py[j];
                            float d = st_d[j];
                            const int s = base + lane;
                            if (base != st_s0[j]) {                       // later chunks (more than 64 stations in the sample)
                                const int sc = s < st_s1[j] ? s : st_s1[j] - 1;
                                px = a.pos[2 * sc]; py = a.pos[2 * sc + 1];
                                d = sdiff ? sdiff[sc] : 0.f;
                            }
                            StationLane st;
                            st.y0 = max(0, py - half); st.y1 = min(a.H, py + half + 1);
                            st.x0 = max(0, px - half); st.x1 = min(a.W, px + half + 1);
                            const int n = (st.y1 > st.y0 && st.x1 > st.x0) ? (st.y1 - st.y0) * (st.x1 - st.x0) : 0;
                            const bool hit = s < st_s1[j] && n > 0 && yy[j] >= st.y0 && yy[j] < st.y1 && 2 * pr + j < rows;
                            unsigned long long m = __ballot(hit);
                            if (!sdiff && m) {
                                // no saved differences: the window means of the hit stations are gathered again
                                unsigned long long mm = m;
                                while (mm) {
                                    const int l = __builtin_ctzll(mm);
                                    mm &= mm - 1;
                                    const int qx0 = __shfl(st.x0, l, 64), qy0 = __shfl(st.y0, l, 64);
                                    const int qx1 = __shfl(st.x1, l, 64), qy1 = __shfl(st.y1, l, 64);
                                    const int ww = qx1 - qx0, nn = ww * (qy1 - qy0);
                                    float sum = 0.f;
                                    for (int i2 = lane; i2 < nn; i2 += 64)
                                        sum += a.pred[((size_t)bb[j] * a.H + qy0 + i2 / ww) * a.W + qx0 + i2 % ww];
                                    sum = wave_sum(sum);
                                    const float dd = sum / (float)nn - a.runoff[base + l];
                                    if (lane == l) d = dd;
                                }
                            }
                            const float gp = a.station_metric == 0 ? 2.0f * d : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
                            st.coef = n > 0 ? gstn * gp / (float)n : 0.f;
                            while (m) {
                                const int l = __builtin_ctzll(m);
                                m &= m - 1;
                                const int qx0 = __shfl(st.x0, l, 64), qx1 = __shfl(st.x1, l, 64);
                                const float qc = __shfl(st.coef, l, 64);
#pragma unroll
                                for (int i = 0; i < 2; ++i)
#pragma unroll
                                    for (int e = 0; e < V; ++e) {
                                        const int x = xv[i] * V + e;
                                        g[2 * j + i][e] += (x >= qx0 && x < qx1) ? qc : 0.f;
                                    }
                            }
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (ok[u]) {
                        if constexpr (V == 4) *reinterpret_cast<float4*>(dpred + off[u]) = make_float4(g[u][0], g[u][1], g[u][2], g[u][3]);
                        else dpred[off[u]] = g[u][0];
                    }
            }
        }
    }
}

// Forward: one read of pred and target.  Pixel blocks reduce the weighted pixel loss (and look for
// Inf / NaN in pred: the training loop's prediction check, src/train.py:198, without a pass of its
// own), station blocks gather the k x k window means (one wavefront per station, lanes over the
// window, shuffle reduction) and keep mean - runoff per station for the backward pass.  Every block
// leaves its partial sums in the workspace; the block that arrives last folds them, in a fixed
// order, into the three losses and the two scalars the backward pass divides by.
template <int V>
__global__ __launch_bounds__(NT) void loss_fwd_kernel(LossArgs a, double* __restrict__ ws, float* losses, float* scal,
                                                      float* __restrict__ sdiff, int* __restrict__ nan_flag,
                                                      int pixel_blocks, int station_blocks) {
    __shared__ double smd[4];
    __shared__ int s_last;
    if ((int)blockIdx.x < pixel_blocks) {
        float accw = 0.f, accl = 0.f;
        bool bad = false;
        pixel_pass<V, false>(a, pixel_blocks, 0.f, 0.f, nullptr, nullptr, accw, accl, bad);
        const double sw = block_sum_256_d((double)accw, smd);
        const double sl = block_sum_256_d((double)accl, smd);
        if (threadIdx.x == 0) { publish(ws + WS_W + blockIdx.x, sw); publish(ws + WS_L + blockIdx.x, sl); }
        if (nan_flag && __any(bad) && (threadIdx.x & 63) == 0) atomicOr(nan_flag, 1);
    } else {
        const int sb = (int)blockIdx.x - pixel_blocks;
        const int lane = threadIdx.x & 63;
        double tl = 0.0, tc = 0.0;
        for (int s = sb * (NT / 64) + (threadIdx.x >> 6); s < a.S; s += station_blocks * (NT / 64)) {
            const int b = sample_of(a.offsets, a.B, s);
            int n, x0, y0, ww;
            const float m = station_mean(a, b, a.pos[2 * s], a.pos[2 * s + 1], lane, &n, &x0, &y0, &ww);
            const float d = n > 0 ? m - a.runoff[s] : 0.f;
            if (lane == 0) {
                if (sdiff) sdiff[s] = d;
                if (n > 0) {
                    tl += (double)(a.station_metric == 0 ? d * d : fabsf(d));
                    tc += 1.0;
                }
            }
        }
        tl = block_sum_256_d(tl, smd);
        tc = block_sum_256_d(tc, smd);
        if (threadIdx.x == 0) { publish(ws + WS_ST + sb, tl); publish(ws + WS_SC + sb, tc); }
    }
    // Arrival.  The partials were written through to device-coherent memory (publish) and are read
    // back the same way (fold_partials), so the hand-over needs no release / acquire fence -- an
    // agent-scope release would write back this XCD's whole L2, once per block: the counters are
    // plain relaxed fetch-adds behind a drain of thread 0's own stores.
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int nblk = (int)gridDim.x, grp = (int)blockIdx.x / ARRIVE_GROUP;
        const int ngrp = (nblk + ARRIVE_GROUP - 1) / ARRIVE_GROUP;
        const int in_grp = min(ARRIVE_GROUP, nblk - grp * ARRIVE_GROUP);
        int* gc = reinterpret_cast<int*>(ws + WS_GROUPS + 16 * grp);
        int last = 0;
        if (__hip_atomic_fetch_add(gc, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == in_grp - 1) {
            __hip_atomic_store(gc, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int* top = reinterpret_cast<int*>(ws + WS_ARRIVED);
            if (__hip_atomic_fetch_add(top, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ngrp - 1) {
                __hip_atomic_store(top, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = 1;
            }
        }
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    // all four partial arrays at once (their loads are independent: one round trip, not four), each
    // summed in a fixed order: thread t takes slots t, t + 256, ..., then the waves, then the block
    __shared__ double sfold[4][NT / 64];
    double f[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < LP; i += NT) {
        const bool ip = i < pixel_blocks, is = i < station_blocks;
        const double a0 = __hip_atomic_load(ws + WS_SC + (is ? i : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double a1 = __hip_atomic_load(ws + WS_W + (ip ? i : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double a2 = __hip_atomic_load(ws + WS_L + (ip ? i : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double a3 = __hip_atomic_load(ws + WS_ST + (is ? i : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        f[0] += is ? a0 : 0.0; f[1] += ip ? a1 : 0.0; f[2] += ip ? a2 : 0.0; f[3] += is ? a3 : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f[q] = wave_sum_d(f[q]);
        if ((threadIdx.x & 63) == 0) sfold[q][threadIdx.x >> 6] = f[q];
    }
    __syncthreads();
    const double cnt = sfold[0][0] + sfold[0][1] + sfold[0][2] + sfold[0][3];
    const double sw = sfold[1][0] + sfold[1][1] + sfold[1][2] + sfold[1][3];
    const double sl = sfold[2][0] + sfold[2][1] + sfold[2][2] + sfold[2][3];
    const double tl = sfold[3][0] + sfold[3][1] + sfold[3][2] + sfold[3][3];
    if (threadIdx.x == 0) {
        const float img = (float)(sl / sw);
        const float stn = cnt > 0.0 ? (float)(tl / cnt) : 0.f;
        losses[1] = img;
        losses[2] = stn;
        losses[0] = a.image_weight * img + a.station_weight * stn;
        if (scal != losses + 3) losses[3] = 1.0f;      // the caller gets the finished gradient (see loss_fused)
        scal[0] = (float)(1.0 / sw);
        scal[1] = cnt > 0.0 ? (float)(1.0 / cnt) : 0.f;
    }
}

// Backward: d total / d pred, complete, in one pass and one store per pixel:
//   upstream * (image_weight / sum w * w * d(pixel loss) + station_weight / count * sum over the
//   stations whose window holds the pixel of d(station loss) / window size).
// scal = {1 / sum w, 1 / count or 0} as the forward pass left them; `upstream` is a device scalar
// (the gradient arriving at the total loss) or null for 1.
template <int V>
__global__ __launch_bounds__(NT) void loss_bwd_kernel(LossArgs a, const float* __restrict__ scal,
                                                      const float* __restrict__ upstream, const float* __restrict__ sdiff,
                                                      float* __restrict__ dpred, int pixel_blocks) {
    const float up = upstream ? upstream[0] : 1.0f;
    const float gimg = up * a.image_weight * scal[0];
    const float gstn = up * a.station_weight * scal[1];
    float accw = 0.f, accl = 0.f;
    bool bad = false;
    pixel_pass<V, true>(a, pixel_blocks, gimg, gstn, sdiff, dpred, accw, accl, bad);
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
                            float* __restrict__ v, long long n, float lr, float b1, float b2, float omb1, float omb2,
                            float eps, float bc1, float bc2, float gs) {
    const float step = lr / bc1;
    const float inv_sqrt_bc2 = 1.0f / sqrtf(bc2);
    for (long long i = blockIdx.x * (long long)NT + threadIdx.x; i < n; i += (long long)gridDim.x * NT) {
        const float gi = g[i] * gs;
        const float mi = b1 * m[i] + omb1 * gi;
        const float vi = b2 * v[i] + omb2 * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= step * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
    }
}

// Adam driven by device-side state: state[0] = number of steps taken, state[1] = steps skipped.
// adam_begin_kernel advances one of the two depending on found_inf; adam_guarded_kernel leaves
// everything untouched when found_inf is set, else forms the bias corrections from state[0] (in
// double, as torch.optim.Adam forms them on the host) and optionally writes a bf16 copy of the
// updated parameters.
__global__ void adam_begin_kernel(int* __restrict__ state, const int* __restrict__ found_inf) {
    if (found_inf && found_inf[0] != 0) state[1] += 1;
    else state[0] += 1;
}

__global__ void adam_guarded_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                    float* __restrict__ v, bf16* __restrict__ shadow, long long n, float lr, double beta1,
                                    double beta2, float eps, float gs, const int* __restrict__ state,
                                    const int* __restrict__ found_inf, int step_offset) {
    if (found_inf && found_inf[0] != 0) return;
    const int t = max(state[0] + step_offset, 1);
    const float bc1 = (float)(1.0 - pow(beta1, (double)t)), bc2 = (float)(1.0 - pow(beta2, (double)t));
    const float b1 = (float)beta1, b2 = (float)beta2, omb1 = (float)(1.0 - beta1), omb2 = (float)(1.0 - beta2);
    const float step = lr / bc1;
    const float inv_sqrt_bc2 = 1.0f / sqrtf(bc2);
    // four elements per thread and iteration when the range allows 16-byte accesses
    const bool vec = ((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16 == 0 && ((uintptr_t)shadow % 8) == 0;
    const long long n4 = vec ? n / 4 : 0;
    for (long long i = blockIdx.x * (long long)NT + threadIdx.x; i < n4; i += (long long)gridDim.x * NT) {
        const float4 g4 = reinterpret_cast<const float4*>(g)[i];
        float4 m4 = reinterpret_cast<float4*>(m)[i], v4 = reinterpret_cast<float4*>(v)[i], p4 = reinterpret_cast<float4*>(p)[i];
        float* gp = const_cast<float*>(&g4.x); float* mp = &m4.x; float* vp = &v4.x; float* pp = &p4.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gi = gp[e] * gs;
            mp[e] = b1 * mp[e] + omb1 * gi;
            vp[e] = b2 * vp[e] + omb2 * gi * gi;
            pp[e] -= step * mp[e] / (sqrtf(vp[e]) * inv_sqrt_bc2 + eps);
        }
        reinterpret_cast<float4*>(m)[i] = m4;
        reinterpret_cast<float4*>(v)[i] = v4;
        reinterpret_cast<float4*>(p)[i] = p4;
        if (shadow) {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16)pp[e];
            reinterpret_cast<bf16x4*>(shadow)[i] = o;
        }
    }
    for (long long i = 4 * n4 + blockIdx.x * (long long)NT + threadIdx.x; i < n; i += (long long)gridDim.x * NT) {
        const float gi = g[i] * gs;
        const float mi = b1 * m[i] + omb1 * gi;
        const float vi = b2 * v[i] + omb2 * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float pi = p[i] - step * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
        p[i] = pi;
        if (shadow) shadow[i] = (bf16)pi;
    }
}

// out[c] = sum_m x[m][c].  A block owns 32 columns; its 8 row groups each take every 8th row,
// four loads in flight (rows past the end read row M-1 with weight 0), and are folded through
// LDS in a fixed order.  Both shapes of the Linear bias gradients are latency bound: [32 x
// 262144] (many blocks, four rows each) and [160 x 128] (four blocks, twenty rows per thread).
__global__ __launch_bounds__(NT) void colsum_f32_kernel(const float* __restrict__ x, long long M, long long C,
                                                        float* __restrict__ out) {
    __shared__ float sm[8][33];
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const long long c = (long long)blockIdx.x * 32 + cl;
    const long long cc = c < C ? c : C - 1;
    float a = 0.f;
    for (long long m = rg; m < M; m += 32) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long mm = m + 8 * u;
            v[u] = x[(mm < M ? mm : M - 1) * C + cc];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) a += (m + 8 * u < M) ? v[u] : 0.f;
    }
    sm[rg][cl] = a;
    __syncthreads();
    if (rg == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) t += sm[g][cl];
        out[c] = t;
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

// Scratch of sf_image_station_loss_static: the per-block partial sums of one loss call.  Each GPU
// the library runs on holds its own copy of this array at its own address; host code never uses
// the array's name as a pointer, only the per-device address hipGetSymbolAddress returned.
static __device__ __align__(256) double scratch[SF_LOSS_WS_DOUBLES];

namespace {
constexpr int MAX_DEVICES = 64;
std::mutex scratch_mutex;
double* scratch_ptr[MAX_DEVICES];     // device address of `scratch` per device, null until first use

// Device address of this device's copy of `scratch` (looked up once per device, with that device current).
int static_scratch(hipStream_t stream, double** out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) {
        sf_set_error("sf_image_station_loss_static: no current device, or device index above %d", MAX_DEVICES - 1);
        return SF_EINVAL;
    }
    // the kernels run on the stream's device: it has to be the current one, whose copy of the array is used
    if (stream) {
        hipDevice_t sdev = -1;
        if (hipStreamGetDevice(stream, &sdev) != hipSuccess) {
            (void)hipGetLastError();
            sf_set_error("sf_image_station_loss_static: not a valid stream");
            return SF_EINVAL;
        }
        if ((int)sdev != dev) {
            sf_set_error("sf_image_station_loss_static: the stream belongs to device %d but device %d is current", (int)sdev, dev);
            return SF_EINVAL;
        }
    }
    std::lock_guard<std::mutex> lock(scratch_mutex);
    if (!scratch_ptr[dev]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            sf_set_error("sf_image_station_loss_static: hipGetSymbolAddress failed: %s", hipGetErrorString(e));
            return SF_ELAUNCH;
        }
        scratch_ptr[dev] = (double*)p;
    }
    *out = scratch_ptr[dev];
    return SF_OK;
}
}  // namespace

namespace {
struct LossPlan { LossArgs a; bool vec; int pb, sb; };

int loss_plan(const char* who, LossPlan& q, const float* pred, const float* target, const int* pos, const float* runoff,
              const int* offsets, int S, int B, int H, int W, int TH, int TW, int image_metric, float nonzero_weight,
              float zero_threshold, int ksize, int station_metric, float image_weight, float station_weight,
              const float* dpred) {
    SF_REQUIRE(pred && target, "%s: null argument", who);
    SF_REQUIRE(B > 0 && H > 0 && W > 0 && TH > 0 && TW > 0 && S >= 0, "%s: bad shape", who);
    SF_REQUIRE(S == 0 || (pos && runoff && offsets), "%s: stations without arrays", who);
    SF_REQUIRE(ksize >= 1, "%s: kernel_size %d", who, ksize);
    SF_REQUIRE((image_metric == 0 || image_metric == 1) && (station_metric == 0 || station_metric == 1), "%s: metric", who);
    LossArgs& a = q.a;
    a.pred = pred; a.target = target; a.pos = pos; a.runoff = runoff; a.offsets = offsets;
    a.S = S; a.B = B; a.H = H; a.W = W; a.TH = TH; a.TW = TW; a.image_metric = image_metric; a.ksize = ksize;
    a.station_metric = station_metric; a.nonzero_weight = nonzero_weight; a.zero_threshold = zero_threshold;
    a.image_weight = image_weight; a.station_weight = station_weight;
    const long long total = (long long)B * H * W;
    q.vec = W % 4 == 0 && ((uintptr_t)pred | (uintptr_t)dpred | (uintptr_t)target) % 16 == 0 &&
            (TW % 4 == 0 || TH != H || TW != W);
    SF_REQUIRE((long long)B * H < 0x7fffffffLL && total < (1LL << 40), "%s: batch too large", who);
    // a wave per pair of rows at a time
    const long long rows = (long long)B * H;
    int pb = (int)((rows + 7) / 8); if (pb > LP) pb = LP; if (pb < 1) pb = 1;
    int sb = (S + 3) / 4; if (sb > LS) sb = LS; if (sb < 1) sb = 1;
    q.pb = pb; q.sb = sb;
    return SF_OK;
}

int loss_fwd_launch(const LossPlan& q, float* losses, float* scal, float* sdiff, int* nan_flag, double* ws, hipStream_t s) {
    if (q.vec) hipLaunchKernelGGL(loss_fwd_kernel<4>, dim3(q.pb + q.sb), dim3(NT), 0, s, q.a, ws, losses, scal, sdiff, nan_flag, q.pb, q.sb);
    else hipLaunchKernelGGL(loss_fwd_kernel<1>, dim3(q.pb + q.sb), dim3(NT), 0, s, q.a, ws, losses, scal, sdiff, nan_flag, q.pb, q.sb);
    SF_LAUNCH_CHECK("sf_image_station_loss_fwd");
    return SF_OK;
}

int loss_bwd_launch(const LossPlan& q, const float* scal, const float* upstream, const float* sdiff, float* dpred,
                    hipStream_t s) {
    if (q.vec) hipLaunchKernelGGL(loss_bwd_kernel<4>, dim3(q.pb), dim3(NT), 0, s, q.a, scal, upstream, sdiff, dpred, q.pb);
    else hipLaunchKernelGGL(loss_bwd_kernel<1>, dim3(q.pb), dim3(NT), 0, s, q.a, scal, upstream, sdiff, dpred, q.pb);
    SF_LAUNCH_CHECK("sf_image_station_loss_bwd");
    return SF_OK;
}
}  // namespace

extern "C" int sf_image_station_loss_fwd(const float* pred, const float* target, const int* pos, const float* runoff,
                                         const int* offsets, int S, int B, int H, int W, int TH, int TW,
                                         int image_metric, float nonzero_weight, float zero_threshold, int ksize,
                                         int station_metric, float image_weight, float station_weight, float* losses,
                                         float* station_diff, int* nan_flag, double* ws, void* stream) {
    SF_REQUIRE(losses && ws, "sf_image_station_loss_fwd: null argument");
    LossPlan q;
    const int rc = loss_plan("sf_image_station_loss_fwd", q, pred, target, pos, runoff, offsets, S, B, H, W, TH, TW,
                             image_metric, nonzero_weight, zero_threshold, ksize, station_metric, image_weight,
                             station_weight, nullptr);
    if (rc != SF_OK) return rc;
    return loss_fwd_launch(q, losses, losses + 3, station_diff, nan_flag, ws, (hipStream_t)stream);
}

extern "C" int sf_image_station_loss_bwd(const float* pred, const float* target, const int* pos, const float* runoff,
                                         const int* offsets, int S, int B, int H, int W, int TH, int TW,
                                         int image_metric, float nonzero_weight, float zero_threshold, int ksize,
                                         int station_metric, float image_weight, float station_weight,
                                         const float* upstream, const float* station_diff,
                                         const float* fwd_losses, float* dpred, void* stream) {
    SF_REQUIRE(dpred && fwd_losses, "sf_image_station_loss_bwd: null argument");
    LossPlan q;
    const int rc = loss_plan("sf_image_station_loss_bwd", q, pred, target, pos, runoff, offsets, S, B, H, W, TH, TW,
                             image_metric, nonzero_weight, zero_threshold, ksize, station_metric, image_weight,
                             station_weight, dpred);
    if (rc != SF_OK) return rc;
    return loss_bwd_launch(q, fwd_losses + 3, upstream, station_diff, dpred, (hipStream_t)stream);
}

// forward + backward back to back with an upstream gradient of 1; `zero_counter`: the workspace is
// the caller's and may hold anything, so its arrival counter is cleared on the stream first
static int loss_fused(const char* who, const float* pred, const float* target, const int* pos, const float* runoff,
                      const int* offsets, int S, int B, int H, int W, int TH, int TW, int image_metric,
                      float nonzero_weight, float zero_threshold, int ksize, int station_metric, float image_weight,
                      float station_weight, float* losses, float* dpred, double* ws, bool zero_counter, hipStream_t s) {
    SF_REQUIRE(losses && dpred && ws, "%s: null argument", who);
    LossPlan q;
    int rc = loss_plan(who, q, pred, target, pos, runoff, offsets, S, B, H, W, TH, TW, image_metric, nonzero_weight,
                       zero_threshold, ksize, station_metric, image_weight, station_weight, dpred);
    if (rc != SF_OK) return rc;
    if (zero_counter && hipMemsetAsync(ws + WS_GROUPS, 0, sizeof(double) * (SF_LOSS_WS_DOUBLES - WS_GROUPS), s) != hipSuccess) {
        sf_set_error("%s: clearing the workspace counter failed", who);
        return SF_ELAUNCH;
    }
    float* scal = reinterpret_cast<float*>(ws + WS_SCAL);      // the two divisors travel through the workspace
    rc = loss_fwd_launch(q, losses, scal, nullptr, nullptr, ws, s);
    if (rc != SF_OK) return rc;
    return loss_bwd_launch(q, scal, nullptr, nullptr, dpred, s);
}

extern "C" int sf_image_station_loss_static(const float* pred, const float* target, const int* pos, const float* runoff,
                                            const int* offsets, int S, int B, int H, int W, int TH, int TW,
                                            int image_metric, float nonzero_weight, float zero_threshold, int ksize,
                                            int station_metric, float image_weight, float station_weight,
                                            float* losses, float* dpred, void* stream) {
    SF_REQUIRE(B > 0 && H > 0 && W > 0 && S >= 0, "sf_image_station_loss_static: bad shape");
    SF_REQUIRE((long long)B * H * W <= SF_LOSS_STATIC_MAX_PIXELS,
               "sf_image_station_loss_static: %lld prediction pixels, the limit is %lld", (long long)B * H * W,
               (long long)SF_LOSS_STATIC_MAX_PIXELS);
    SF_REQUIRE(S <= SF_LOSS_STATIC_MAX_STATIONS, "sf_image_station_loss_static: %d stations, the limit is %d", S,
               SF_LOSS_STATIC_MAX_STATIONS);
    double* ws = nullptr;
    const int rc = static_scratch((hipStream_t)stream, &ws);
    if (rc != SF_OK) return rc;
    // every call writes all the partial-sum slots it later reads, and the arrival counter -- zero
    // when the library is loaded -- is put back to zero by the block that folds the sums, so the
    // previous call's contents of the array are never seen and nothing has to be cleared here
    return loss_fused("sf_image_station_loss_static", pred, target, pos, runoff, offsets, S, B, H, W, TH, TW,
                      image_metric, nonzero_weight, zero_threshold, ksize, station_metric, image_weight, station_weight,
                      losses, dpred, ws, false, (hipStream_t)stream);
}

extern "C" int sf_image_station_loss(const float* pred, const float* target, const int* pos, const float* runoff,
                                     const int* offsets, int S, int B, int H, int W, int TH, int TW, int image_metric,
                                     float nonzero_weight, float zero_threshold, int ksize, int station_metric,
                                     float image_weight, float station_weight, float* losses, float* dpred, double* ws,
                                     void* stream) {
    return loss_fused("sf_image_station_loss", pred, target, pos, runoff, offsets, S, B, H, W, TH, TW, image_metric,
                      nonzero_weight, zero_threshold, ksize, station_metric, image_weight, station_weight, losses, dpred,
                      ws, true, (hipStream_t)stream);
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

extern "C" int sf_adam(float* p, const float* g, float* m, float* v, long long n, float lr, double beta1, double beta2,
                       float eps, float bc1, float bc2, float grad_scale, void* stream) {
    SF_REQUIRE(p && g && m && v && n > 0, "sf_adam: bad arguments");
    SF_REQUIRE(bc1 > 0.f && bc2 > 0.f, "sf_adam: bias corrections must be positive");
    int grid = (int)((n + NT - 1) / NT); if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(NT), 0, (hipStream_t)stream, p, g, m, v, n, lr, (float)beta1,
                       (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), eps, bc1, bc2, grad_scale);
    SF_LAUNCH_CHECK("sf_adam");
    return SF_OK;
}

extern "C" int sf_adam_begin(int* state, const int* found_inf, void* stream) {
    SF_REQUIRE(state, "sf_adam_begin: null state");
    hipLaunchKernelGGL(adam_begin_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, found_inf);
    SF_LAUNCH_CHECK("sf_adam_begin");
    return SF_OK;
}

extern "C" int sf_adam_guarded(float* p, const float* g, float* m, float* v, void* shadow_bf16, long long n, float lr,
                               double beta1, double beta2, float eps, float grad_scale, const int* state,
                               const int* found_inf, int step_offset, void* stream) {
    SF_REQUIRE(p && g && m && v && state && n > 0 && step_offset <= 0, "sf_adam_guarded: bad arguments");
    SF_REQUIRE(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0, "sf_adam_guarded: betas must lie in [0, 1)");
    long long work = (n + 3) / 4;
    int grid = (int)((work + NT - 1) / NT); if (grid > 8192) grid = 8192; if (grid < 1) grid = 1;
    hipLaunchKernelGGL(adam_guarded_kernel, dim3(grid), dim3(NT), 0, (hipStream_t)stream, p, g, m, v, (bf16*)shadow_bf16, n,
                       lr, beta1, beta2, eps, grad_scale, state, found_inf, step_offset);
    SF_LAUNCH_CHECK("sf_adam_guarded");
    return SF_OK;
}

extern "C" int sf_colsum_f32(const float* x, long long M, long long C, float* out, void* stream) {
    SF_REQUIRE(x && out && M > 0 && C > 0, "sf_colsum_f32: bad arguments");
    const long long grid = (C + 31) / 32;
    SF_REQUIRE(grid <= 0x7fffffffLL, "sf_colsum_f32: too many columns");
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
