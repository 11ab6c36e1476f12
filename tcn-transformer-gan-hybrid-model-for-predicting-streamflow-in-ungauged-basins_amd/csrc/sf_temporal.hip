// Temporal modules over the short day window: one workgroup per sample keeps the whole
// [T<=8, D<=128] sequence in LDS, so a Transformer encoder layer (or the LSTM stack) is a
// single launch forward and a single launch backward.  LayerNorm / softmax statistics use
// wavefront reductions.  All arithmetic is f32.
#include "sf_common.h"
#include "../../include/sfhip.h"

namespace {

constexpr int NT = 256;
constexpr int MAXT = 8;

struct Ptrs12 { const float* p[12]; };
struct MPtrs12 { float* p[12]; };
struct Ptrs16 { const float* p[16]; };
struct MPtrs16 { float* p[16]; };

// multiplicative dropout mask value (0 or 1/(1-p)) for element i of site s of sample b
__device__ __forceinline__ float drop_mask(float p, unsigned long long seed, int b, int site, int i) {
    if (p <= 0.f) return 1.f;
    const unsigned long long idx = ((unsigned long long)b * 16ull + (unsigned long long)site) * 65536ull + (unsigned long long)i;
    return rand_u(seed, idx) >= p ? 1.0f / (1.0f - p) : 0.f;
}

// acc[t] = wrow . v[t] for the T tokens (v in LDS, row stride ldv): the weight row is read once,
// sixteen 16-byte loads in flight
__device__ __forceinline__ void dot_row_tokens(const float* __restrict__ wrow, const float* __restrict__ v, int ldv,
                                               int n, int T, float* acc) {
#pragma unroll
    for (int t = 0; t < MAXT; ++t) acc[t] = 0.f;
    constexpr int NL = 16;   // 16-byte loads in flight per thread (the kernel is latency bound)
    for (int d = 0; d < n; d += 4 * NL) {
        float4 w[NL];
#pragma unroll
        for (int k = 0; k < NL; ++k) w[k] = *reinterpret_cast<const float4*>(wrow + (d + 4 * k < n ? d + 4 * k : 0));
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            if (d + 4 * k < n) {
#pragma unroll
                for (int t = 0; t < MAXT; ++t)
                    if (t < T) {
                        const float4 vv = *reinterpret_cast<const float4*>(v + t * ldv + d + 4 * k);   // one 16-byte LDS read
                        acc[t] += w[k].x * vv.x + w[k].y * vv.y + w[k].z * vv.z + w[k].w * vv.w;
                    }
            }
        }
    }
}

// As dot_row_tokens, with the row shared by RP = 8 consecutive lanes: lane q takes the 16-byte
// chunks q, q + 8, ... of the row, so the eight lanes read 128 contiguous bytes per load (a row
// per lane touches 64 cache lines per wave instruction), and the partial sums are folded with
// lane shuffles.  Every lane of the group ends with the full sums.
constexpr int RP = 8;
__device__ __forceinline__ void dot_row_tokens_shared(const float* __restrict__ wrow, const float* __restrict__ v,
                                                      int ldv, int n, int T, int q, float* acc) {
#pragma unroll
    for (int t = 0; t < MAXT; ++t) acc[t] = 0.f;
    constexpr int NL = 8;
    const int nchunks = n >> 2;
    for (int c0 = q; c0 < nchunks; c0 += RP * NL) {
        float4 w[NL];
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int c = c0 + k * RP;
            w[k] = *reinterpret_cast<const float4*>(wrow + 4 * (c < nchunks ? c : 0));
        }
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int c = c0 + k * RP;
            if (c < nchunks) {
#pragma unroll
                for (int t = 0; t < MAXT; ++t)
                    if (t < T) {
                        const float4 vv = *reinterpret_cast<const float4*>(v + t * ldv + 4 * c);
                        acc[t] += w[k].x * vv.x + w[k].y * vv.y + w[k].z * vv.z + w[k].w * vv.w;
                    }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
        if (t < T) {
#pragma unroll
            for (int o = RP / 2; o > 0; o >>= 1) acc[t] += __shfl_xor(acc[t], o, 64);
        }
}

__device__ __forceinline__ float dot_row(const float* __restrict__ wrow, const float* __restrict__ v, int n) {
    float a = 0.f;
    for (int d = 0; d < n; d += 4) {
        const float4 w = *reinterpret_cast<const float4*>(wrow + d);
        const float4 x = *reinterpret_cast<const float4*>(v + d);   // v is 16-byte aligned (LDS rows of H % 4 == 0 floats)
        a += w.x * x.x + w.y * x.y + w.z * x.z + w.w * x.w;
    }
    return a;
}

// save layout: per-sample sizes and offsets; array a of sample b lives at
// save + B*off_a + b*size_a, i.e. every array is one contiguous [B*T, width] matrix
struct TrSave {
    int qkv, probs, ctx, xhat1, rstd1, x1, hd, xhat2, rstd2, total;
    __host__ __device__ TrSave(int T, int D, int F) {
        int o = 0;
        qkv = o; o += T * 3 * D;
        probs = o; o += 8 * T * T;
        ctx = o; o += T * D;
        xhat1 = o; o += T * D;
        rstd1 = o; o += MAXT;
        x1 = o; o += T * D;
        hd = o; o += T * F;
        xhat2 = o; o += T * D;
        rstd2 = o; o += MAXT;
        total = (o + 3) & ~3;
    }
};

// LayerNorm over D for every token of `in` (LDS); wave w handles tokens w, w+4.
__device__ __forceinline__ void layer_norm_tokens(const float* in, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, float* out, float* xhat_g,
                                                  float* rstd_g, int T, int D) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t = wave; t < T; t += (int)(blockDim.x >> 6)) {
        float s = 0.f;
        for (int d = lane; d < D; d += 64) s += in[t * D + d];
        const float mean = wave_sum(s) / (float)D;
        float q = 0.f;
        for (int d = lane; d < D; d += 64) { const float c = in[t * D + d] - mean; q += c * c; }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + 1e-5f);
        for (int d = lane; d < D; d += 64) {
            const float xh = (in[t * D + d] - mean) * rstd;
            if (xhat_g) xhat_g[t * D + d] = xh;
            out[t * D + d] = xh * gamma[d] + beta[d];
        }
        if (rstd_g && lane == 0) rstd_g[t] = rstd;
    }
}

// 512 threads: the row-per-thread matrix products of the forward pass are latency bound and the
// in_proj / FFN ones have 384 / 512 rows (138 -> 113 us per layer at B = 32; the backward pass
// measured five times slower with 512 threads and keeps 256)
constexpr int NT_FWD = 512;
__global__ __launch_bounds__(NT_FWD) void transformer_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                 Ptrs12 W, float* __restrict__ save, int T, int D,
                                                                 int F, int nh, float p, unsigned long long seed) {
    constexpr int NT = NT_FWD;
    extern __shared__ float sm[];
    float* xs = sm;                 // [T][D]
    float* qkv = xs + T * D;        // [T][3D]
    float* pr = qkv + T * 3 * D;    // [nh][T][T]
    float* ctx = pr + 8 * T * T;    // [T][D]
    float* x1 = ctx + T * D;        // [T][D]
    float* hd = x1 + T * D;         // [T][F]
    float* tmp = hd + T * F;        // [T][D]
    const int b = blockIdx.x, tid = threadIdx.x;
    const TrSave L(T, D, F);
    const int B = gridDim.x;
    float* const sv = save;
    float* const s_qkv = save ? save + (size_t)B * L.qkv + (size_t)b * T * 3 * D : nullptr;
    float* const s_probs = save ? save + (size_t)B * L.probs + (size_t)b * 8 * T * T : nullptr;
    float* const s_ctx = save ? save + (size_t)B * L.ctx + (size_t)b * T * D : nullptr;
    float* const s_xhat1 = save ? save + (size_t)B * L.xhat1 + (size_t)b * T * D : nullptr;
    float* const s_rstd1 = save ? save + (size_t)B * L.rstd1 + (size_t)b * MAXT : nullptr;
    float* const s_x1 = save ? save + (size_t)B * L.x1 + (size_t)b * T * D : nullptr;
    float* const s_hd = save ? save + (size_t)B * L.hd + (size_t)b * T * F : nullptr;
    float* const s_xhat2 = save ? save + (size_t)B * L.xhat2 + (size_t)b * T * D : nullptr;
    float* const s_rstd2 = save ? save + (size_t)B * L.rstd2 + (size_t)b * MAXT : nullptr;
    const int hdm = D / nh;
    const float* Win = W.p[0]; const float* bin = W.p[1]; const float* Wo = W.p[2]; const float* bo = W.p[3];
    const float* W1 = W.p[4]; const float* b1 = W.p[5]; const float* W2 = W.p[6]; const float* b2 = W.p[7];

    for (int i = tid; i < T * D; i += NT) xs[i] = x[(size_t)b * T * D + i];
    __syncthreads();
    // the projections with as many rows as threads take a row per thread; the two with 128 rows
    // share each row among RP lanes
    const int rq = tid % RP;
    for (int o = tid; o < 3 * D; o += NT) {
        const float bb = bin[o];
        float dacc[MAXT];
        dot_row_tokens(Win + (size_t)o * D, xs, D, D, T, dacc);
        for (int t = 0; t < T; ++t) {
            const float v = dacc[t] + bb;
            qkv[t * 3 * D + o] = v;
            if (sv) s_qkv[t * 3 * D + o] = v;
        }
    }
    __syncthreads();
    const float scale = 1.0f / sqrtf((float)hdm);
    for (int i = tid; i < nh * T * T; i += NT) {
        const int s = i % T, t = (i / T) % T, h = i / (T * T);
        float a = 0.f;
        for (int j = 0; j < hdm; ++j) a += qkv[t * 3 * D + h * hdm + j] * qkv[s * 3 * D + D + h * hdm + j];
        pr[i] = a * scale;
    }
    __syncthreads();
    for (int i = tid; i < nh * T; i += NT) {  // softmax over s for row (h,t)
        float* row = pr + i * T;
        float m = row[0];
        for (int s = 1; s < T; ++s) m = fmaxf(m, row[s]);
        float den = 0.f;
        for (int s = 0; s < T; ++s) { row[s] = expf(row[s] - m); den += row[s]; }
        const float inv = 1.0f / den;
        for (int s = 0; s < T; ++s) {
            const float pv = row[s] * inv;
            if (sv) s_probs[i * T + s] = pv;
            row[s] = pv * drop_mask(p, seed, b, 0, i * T + s);
        }
    }
    __syncthreads();
    for (int i = tid; i < T * D; i += NT) {
        const int d = i % D, t = i / D, h = d / hdm;
        float a = 0.f;
        for (int s = 0; s < T; ++s) a += pr[(h * T + t) * T + s] * qkv[s * 3 * D + 2 * D + d];
        ctx[i] = a;
        if (sv) s_ctx[i] = a;
    }
    __syncthreads();
    for (int o = tid / RP; o < D; o += NT / RP) {
        const float bb = bo[o];
        float dacc[MAXT];
        dot_row_tokens_shared(Wo + (size_t)o * D, ctx, D, D, T, rq, dacc);
        if (rq == 0) for (int t = 0; t < T; ++t) {
            const float v = dacc[t] + bb;
            tmp[t * D + o] = xs[t * D + o] + v * drop_mask(p, seed, b, 1, t * D + o);
        }
    }
    __syncthreads();
    layer_norm_tokens(tmp, W.p[8], W.p[9], x1, s_xhat1, s_rstd1, T, D);
    __syncthreads();
    if (sv) for (int i = tid; i < T * D; i += NT) s_x1[i] = x1[i];
    for (int f = tid; f < F; f += NT) {
        const float bb = b1[f];
        float dacc[MAXT];
        dot_row_tokens(W1 + (size_t)f * D, x1, D, D, T, dacc);
        for (int t = 0; t < T; ++t) {
            float v = dacc[t] + bb;
            v = v < 0.f ? 0.f : v * drop_mask(p, seed, b, 2, t * F + f);   // ReLU that keeps NaN
            hd[t * F + f] = v;
            if (sv) s_hd[t * F + f] = v;
        }
    }
    __syncthreads();
    for (int o = tid / RP; o < D; o += NT / RP) {
        const float bb = b2[o];
        float dacc[MAXT];
        dot_row_tokens_shared(W2 + (size_t)o * F, hd, F, F, T, rq, dacc);
        if (rq == 0) for (int t = 0; t < T; ++t) {
            const float v = dacc[t] + bb;
            tmp[t * D + o] = x1[t * D + o] + v * drop_mask(p, seed, b, 3, t * D + o);
        }
    }
    __syncthreads();
    layer_norm_tokens(tmp, W.p[10], W.p[11], xs, s_xhat2, s_rstd2, T, D);
    __syncthreads();
    for (int i = tid; i < T * D; i += NT) y[(size_t)b * T * D + i] = xs[i];
}

// LayerNorm backward for every token: g (LDS, gradient wrt LN output) -> out (gradient wrt LN input)
__device__ __forceinline__ void layer_norm_bwd_tokens(const float* g, const float* __restrict__ gamma,
                                                      const float* __restrict__ xhat, const float* __restrict__ rstd,
                                                      float* out, int T, int D) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t = wave; t < T; t += (int)(blockDim.x >> 6)) {
        float s1 = 0.f, s2 = 0.f;
        for (int d = lane; d < D; d += 64) {
            const float gy = g[t * D + d] * gamma[d];
            s1 += gy;
            s2 += gy * xhat[t * D + d];
        }
        const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
        const float r = rstd[t];
        for (int d = lane; d < D; d += 64) {
            const float gy = g[t * D + d] * gamma[d];
            out[t * D + d] = r * (gy - m1 - xhat[t * D + d] * m2);
        }
    }
}

// out[t][j] (+)= sum_o in[t][o] * Wm[o][j] for the T tokens of the sample (in / out in LDS,
// Wm row-major in global memory, NJ % 4 == 0).  A thread owns four consecutive j; when there are
// fewer such groups than threads the reduction range is split over NT / (NJ/4) parts that are
// summed through `scratch` (parts * MAXT * NJ floats).  The weight loads of eight reduction
// steps are issued together.  Ends with a barrier.
__device__ __forceinline__ void tok_matmul(const float* __restrict__ in, int ldin, const float* __restrict__ Wm,
                                           int ldw, int NO, int NJ, int T, float* __restrict__ out, int ldout,
                                           bool add, float* __restrict__ scratch) {
    const int tid = threadIdx.x;
    const int J4 = NJ >> 2;
    const int parts = J4 >= NT ? 1 : NT / J4;
    const int per = (NO + parts - 1) / parts;
    for (int g0 = 0; g0 < J4; g0 += NT) {
        const int j4 = parts == 1 ? g0 + tid : tid % J4;
        const int part = parts == 1 ? 0 : tid / J4;
        const bool live = j4 < J4 && part < parts;
        float acc[MAXT][4];
#pragma unroll
        for (int t = 0; t < MAXT; ++t) acc[t][0] = acc[t][1] = acc[t][2] = acc[t][3] = 0.f;
        if (live) {
            const int o0 = part * per, o1 = o0 + per < NO ? o0 + per : NO;
            const float* wp = Wm + 4 * j4;
            constexpr int NL = 8;    // more in flight measured slower here (fewer, longer dependent add chains win)
            for (int ob = o0; ob < o1; ob += NL) {
                float4 w[NL];
#pragma unroll
                for (int k = 0; k < NL; ++k) {
                    const int o = ob + k < o1 ? ob + k : o0;
                    w[k] = *reinterpret_cast<const float4*>(wp + (size_t)o * ldw);
                }
                // the tokens' gradient values of these NL rows: two 16-byte LDS reads per token when the
                // batch is whole and aligned (scalar reads of the same values bound this loop before)
                float gv[MAXT][NL];
                const bool whole = ob + NL <= o1 && ((ob | ldin) & 3) == 0;
#pragma unroll
                for (int t = 0; t < MAXT; ++t) {
                    if (t < T) {
                        if (whole) {
#pragma unroll
                            for (int k4 = 0; k4 < NL; k4 += 4) {
                                const float4 v = *reinterpret_cast<const float4*>(in + t * ldin + ob + k4);
                                gv[t][k4] = v.x; gv[t][k4 + 1] = v.y; gv[t][k4 + 2] = v.z; gv[t][k4 + 3] = v.w;
                            }
                        } else {
#pragma unroll
                            for (int k = 0; k < NL; ++k) gv[t][k] = ob + k < o1 ? in[t * ldin + ob + k] : 0.f;
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < NL; ++k) {
#pragma unroll
                    for (int t = 0; t < MAXT; ++t)
                        if (t < T) {
                            const float g = gv[t][k];
                            acc[t][0] += g * w[k].x; acc[t][1] += g * w[k].y;
                            acc[t][2] += g * w[k].z; acc[t][3] += g * w[k].w;
                        }
                }
            }
        }
        if (parts == 1) {
            if (live) {
#pragma unroll
                for (int t = 0; t < MAXT; ++t)
                    if (t < T) {
                        float* o4 = out + t * ldout + 4 * j4;
                        if (add) { o4[0] += acc[t][0]; o4[1] += acc[t][1]; o4[2] += acc[t][2]; o4[3] += acc[t][3]; }
                        else { o4[0] = acc[t][0]; o4[1] = acc[t][1]; o4[2] = acc[t][2]; o4[3] = acc[t][3]; }
                    }
            }
        } else {
            if (live) {
#pragma unroll
                for (int t = 0; t < MAXT; ++t)
                    if (t < T) {
                        float* s4 = scratch + ((size_t)part * MAXT + t) * NJ + 4 * j4;
                        s4[0] = acc[t][0]; s4[1] = acc[t][1]; s4[2] = acc[t][2]; s4[3] = acc[t][3];
                    }
            }
            __syncthreads();
            for (int i = tid; i < T * NJ; i += NT) {
                const int t = i / NJ, j = i - t * NJ;
                float a = 0.f;
                for (int q = 0; q < parts; ++q) a += scratch[((size_t)q * MAXT + t) * NJ + j];
                out[t * ldout + j] = add ? out[t * ldout + j] + a : a;
            }
            break;  // NJ/4 < NT: a single pass covered every group
        }
    }
    __syncthreads();
}

// Backward of one encoder layer for one sample.  Gradients of the four weight MATRICES are not
// formed here: the per-token gradient vectors are written to `gtok` (arrays g_qkv [B,T,3D],
// g_sa [B,T,D], g_z [B,T,F], g_ff [B,T,D]) and the caller contracts them with the saved
// activations over all B*T tokens in four small GEMMs.  Bias / LayerNorm gradients (short
// vectors) are accumulated with atomics.
__global__ __launch_bounds__(NT) void transformer_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                             float* __restrict__ dx, Ptrs12 W, MPtrs12 G,
                                                             const float* __restrict__ save, float* __restrict__ gtok,
                                                             int T, int D, int F, int nh, float p,
                                                             unsigned long long seed) {
    extern __shared__ float sm[];
    float* g0 = sm;                 // [T][D]
    float* g1 = g0 + T * D;         // [T][D]
    float* gF = g1 + T * D;         // [T][F]  (also d_ctx)
    float* gq = gF + T * F;         // [T][3D]
    float* gp = gq + T * 3 * D;     // [nh][T][T]
    float* scr = gp + 8 * T * T;    // tok_matmul scratch
    const int b = blockIdx.x, tid = threadIdx.x, B = gridDim.x;
    const TrSave L(T, D, F);
    const float* s_qkv = save + (size_t)B * L.qkv + (size_t)b * T * 3 * D;
    const float* s_probs = save + (size_t)B * L.probs + (size_t)b * 8 * T * T;
    const float* s_xhat1 = save + (size_t)B * L.xhat1 + (size_t)b * T * D;
    const float* s_rstd1 = save + (size_t)B * L.rstd1 + (size_t)b * MAXT;
    const float* s_hd = save + (size_t)B * L.hd + (size_t)b * T * F;
    const float* s_xhat2 = save + (size_t)B * L.xhat2 + (size_t)b * T * D;
    const float* s_rstd2 = save + (size_t)B * L.rstd2 + (size_t)b * MAXT;
    float* t_qkv = gtok + (size_t)b * T * 3 * D;
    float* t_sa = gtok + (size_t)B * T * 3 * D + (size_t)b * T * D;
    float* t_z = gtok + (size_t)B * T * 4 * D + (size_t)b * T * F;
    float* t_ff = gtok + (size_t)B * T * (4 * D + F) + (size_t)b * T * D;
    const int hdm = D / nh;
    const float* Win = W.p[0]; const float* Wo = W.p[2]; const float* W1 = W.p[4]; const float* W2 = W.p[6];

    for (int i = tid; i < T * D; i += NT) g0[i] = dy[(size_t)b * T * D + i];
    __syncthreads();
    // ---- LayerNorm 2
    for (int d = tid; d < D; d += NT) {
        float a = 0.f, c = 0.f;
        for (int t = 0; t < T; ++t) { a += g0[t * D + d] * s_xhat2[t * D + d]; c += g0[t * D + d]; }
        unsafeAtomicAdd(G.p[10] + d, a);
        unsafeAtomicAdd(G.p[11] + d, c);
    }
    layer_norm_bwd_tokens(g0, W.p[10], s_xhat2, s_rstd2, g1, T, D);
    __syncthreads();
    // g1 = grad wrt (x1 + ff); d_ff = g1 * mask2 -> g0 (and out to the token-gradient buffer)
    for (int i = tid; i < T * D; i += NT) {
        const float v = g1[i] * drop_mask(p, seed, b, 3, i);
        g0[i] = v;
        t_ff[i] = v;
    }
    __syncthreads();
    for (int o = tid; o < D; o += NT) {  // linear2 bias
        float a = 0.f;
        for (int t = 0; t < T; ++t) a += g0[t * D + o];
        unsafeAtomicAdd(G.p[7] + o, a);
    }
    // d_hidden -> pre-activation gradient (ReLU and dropout share the saved value's sign)
    tok_matmul(g0, D, W2, F, D, F, T, gF, F, false, scr);
    for (int f = tid; f < F; f += NT) {
        float bsum = 0.f;
#pragma unroll
        for (int t = 0; t < MAXT; ++t)
            if (t < T) {
                const float keep = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
                const float v = s_hd[t * F + f] > 0.f ? gF[t * F + f] * keep : 0.f;
                gF[t * F + f] = v;
                t_z[t * F + f] = v;
                bsum += v;
            }
        unsafeAtomicAdd(G.p[5] + f, bsum);  // linear1 bias
    }
    __syncthreads();
    tok_matmul(gF, F, W1, D, F, D, T, g1, D, true, scr);  // d_x1 += dz . W1
    // ---- LayerNorm 1 (g1 = gradient wrt x1)
    for (int d = tid; d < D; d += NT) {
        float a = 0.f, c = 0.f;
        for (int t = 0; t < T; ++t) { a += g1[t * D + d] * s_xhat1[t * D + d]; c += g1[t * D + d]; }
        unsafeAtomicAdd(G.p[8] + d, a);
        unsafeAtomicAdd(G.p[9] + d, c);
    }
    layer_norm_bwd_tokens(g1, W.p[8], s_xhat1, s_rstd1, g0, T, D);
    __syncthreads();
    // g0 = gradient wrt (x + sa); d_sa = g0 * mask1 -> g1
    for (int i = tid; i < T * D; i += NT) {
        const float v = g0[i] * drop_mask(p, seed, b, 1, i);
        g1[i] = v;
        t_sa[i] = v;
    }
    __syncthreads();
    for (int o = tid; o < D; o += NT) {  // out_proj bias
        float a = 0.f;
        for (int t = 0; t < T; ++t) a += g1[t * D + o];
        unsafeAtomicAdd(G.p[3] + o, a);
    }
    __syncthreads();
    tok_matmul(g1, D, Wo, D, D, D, T, gF, D, false, scr);  // d_ctx -> gF[0 : T*D]
    // ---- attention: d_pused = d_ctx . v
    const float* qkv = s_qkv;
    for (int i = tid; i < nh * T * T; i += NT) {
        const int s = i % T, t = (i / T) % T, h = i / (T * T);
        float a = 0.f;
        for (int j = 0; j < hdm; ++j) a += gF[t * D + h * hdm + j] * qkv[s * 3 * D + 2 * D + h * hdm + j];
        gp[i] = a;
    }
    __syncthreads();
    const float scale = 1.0f / sqrtf((float)hdm);
    for (int i = tid; i < nh * T; i += NT) {  // softmax backward per row (h,t)
        float dp[MAXT];
        float dot = 0.f;
        for (int s = 0; s < T; ++s) {
            dp[s] = gp[i * T + s] * drop_mask(p, seed, b, 0, i * T + s);
            dot += dp[s] * s_probs[i * T + s];
        }
        for (int s = 0; s < T; ++s) gp[i * T + s] = s_probs[i * T + s] * (dp[s] - dot) * scale;
    }
    __syncthreads();
    for (int i = tid; i < T * D; i += NT) {
        const int d = i % D, t = i / D, h = d / hdm;
        float dq = 0.f, dk = 0.f, dv = 0.f;
        for (int s = 0; s < T; ++s) {
            dq += gp[(h * T + t) * T + s] * qkv[s * 3 * D + D + d];
            dk += gp[(h * T + s) * T + t] * qkv[s * 3 * D + d];
            const float pu = s_probs[(h * T + s) * T + t] * drop_mask(p, seed, b, 0, (h * T + s) * T + t);
            dv += pu * gF[s * D + d];
        }
        gq[t * 3 * D + d] = dq;
        gq[t * 3 * D + D + d] = dk;
        gq[t * 3 * D + 2 * D + d] = dv;
        t_qkv[t * 3 * D + d] = dq;
        t_qkv[t * 3 * D + D + d] = dk;
        t_qkv[t * 3 * D + 2 * D + d] = dv;
    }
    __syncthreads();
    for (int o = tid; o < 3 * D; o += NT) {  // in_proj bias
        float a = 0.f;
        for (int t = 0; t < T; ++t) a += gq[t * 3 * D + o];
        unsafeAtomicAdd(G.p[1] + o, a);
    }
    __syncthreads();
    tok_matmul(gq, 3 * D, Win, D, 3 * D, D, T, g0, D, true, scr);  // dx = g0 + gq . Win
    for (int i = tid; i < T * D; i += NT) dx[(size_t)b * T * D + i] = g0[i];
}

__global__ void add_pe_kernel(const float* __restrict__ x, const float* __restrict__ pe, float* __restrict__ y,
                              long long total, int TD) {
    for (long long i = blockIdx.x * (long long)NT + threadIdx.x; i < total; i += (long long)gridDim.x * NT)
        y[i] = x[i] + pe[i % TD];
}
__global__ void mean_t_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int T, int D) {
    for (int i = blockIdx.x * NT + threadIdx.x; i < B * D; i += gridDim.x * NT) {
        const int d = i % D, b = i / D;
        float a = 0.f;
        for (int t = 0; t < T; ++t) a += x[((size_t)b * T + t) * D + d];
        y[i] = a / (float)T;
    }
}
__global__ void mean_t_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int B, int T, int D) {
    for (int i = blockIdx.x * NT + threadIdx.x; i < B * T * D; i += gridDim.x * NT) {
        const int d = i % D, b = i / (T * D);
        dx[i] = dy[b * D + d] / (float)T;
    }
}

// ------------------------------------------------------------------------------ LSTM
__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

__global__ __launch_bounds__(NT) void lstm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, Ptrs16 W,
                                                      float* __restrict__ save, int T, int H, int L, float p,
                                                      unsigned long long seed) {
    extern __shared__ float sm[];
    float* xin = sm;             // [T][H]
    float* hs = xin + T * H;     // [T][H]
    float* h = hs + T * H;       // [H]
    float* c = h + H;            // [H]
    float* gates = c + H;        // [4H]
    const int b = blockIdx.x, tid = threadIdx.x;
    float* sv = save ? save + (size_t)b * T * L * 6 * H : nullptr;
    for (int i = tid; i < T * H; i += NT) xin[i] = x[(size_t)b * T * H + i];
    for (int l = 0; l < L; ++l) {
        const float* Wih = W.p[4 * l]; const float* Whh = W.p[4 * l + 1];
        const float* bih = W.p[4 * l + 2]; const float* bhh = W.p[4 * l + 3];
        for (int i = tid; i < H; i += NT) { h[i] = 0.f; c[i] = 0.f; }
        __syncthreads();
        for (int t = 0; t < T; ++t) {
            for (int r = tid; r < 4 * H; r += NT)
                gates[r] = bih[r] + bhh[r] + dot_row(Wih + (size_t)r * H, xin + t * H, H) + dot_row(Whh + (size_t)r * H, h, H);
            __syncthreads();
            for (int j = tid; j < H; j += NT) {
                const float ig = sigmoidf_(gates[j]), fg = sigmoidf_(gates[H + j]);
                const float gg = tanhf(gates[2 * H + j]), og = sigmoidf_(gates[3 * H + j]);
                const float cn = fg * c[j] + ig * gg;
                const float hn = og * tanhf(cn);
                c[j] = cn; h[j] = hn; hs[t * H + j] = hn;
                if (sv) {
                    float* s = sv + ((size_t)l * T + t) * 6 * H;
                    s[j] = ig; s[H + j] = fg; s[2 * H + j] = gg; s[3 * H + j] = og; s[4 * H + j] = cn; s[5 * H + j] = hn;
                }
            }
            __syncthreads();
        }
        if (l + 1 < L) {
            for (int i = tid; i < T * H; i += NT) xin[i] = hs[i] * drop_mask(p, seed, b, l, i);
            __syncthreads();
        }
    }
    for (int j = tid; j < H; j += NT) y[(size_t)b * H + j] = h[j];
}

__global__ __launch_bounds__(NT) void lstm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                      float* __restrict__ dx, Ptrs16 W, MPtrs16 G,
                                                      const float* __restrict__ save, int T, int H, int L, float p,
                                                      unsigned long long seed) {
    extern __shared__ float sm[];
    float* dext = sm;              // [T][H] external gradient on this layer's outputs
    float* dxin = dext + T * H;    // [T][H] gradient wrt this layer's inputs
    float* dG = dxin + T * H;      // [T][4H] pre-activation gate gradients
    float* dh = dG + T * 4 * H;    // [H]
    float* dc = dh + H;            // [H]
    float* xin = dc + H;           // [T][H] this layer's inputs
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* sv = save + (size_t)b * T * L * 6 * H;
    for (int i = tid; i < T * H; i += NT) dext[i] = (i / H == T - 1) ? dy[(size_t)b * H + i % H] : 0.f;
    __syncthreads();
    for (int l = L - 1; l >= 0; --l) {
        const float* Wih = W.p[4 * l]; const float* Whh = W.p[4 * l + 1];
        for (int i = tid; i < T * H; i += NT) {
            if (l == 0) xin[i] = x[(size_t)b * T * H + i];
            else xin[i] = sv[((size_t)(l - 1) * T + i / H) * 6 * H + 5 * H + i % H] * drop_mask(p, seed, b, l - 1, i);
        }
        for (int i = tid; i < H; i += NT) { dh[i] = 0.f; dc[i] = 0.f; }
        __syncthreads();
        for (int t = T - 1; t >= 0; --t) {
            const float* s = sv + ((size_t)l * T + t) * 6 * H;
            for (int j = tid; j < H; j += NT) {
                const float ig = s[j], fg = s[H + j], gg = s[2 * H + j], og = s[3 * H + j], cn = s[4 * H + j];
                const float cprev = t > 0 ? sv[((size_t)l * T + t - 1) * 6 * H + 4 * H + j] : 0.f;
                const float dht = dext[t * H + j] + dh[j];
                const float tc = tanhf(cn);
                const float dct = dc[j] + dht * og * (1.f - tc * tc);
                dG[t * 4 * H + j] = dct * gg * ig * (1.f - ig);
                dG[t * 4 * H + H + j] = dct * cprev * fg * (1.f - fg);
                dG[t * 4 * H + 2 * H + j] = dct * ig * (1.f - gg * gg);
                dG[t * 4 * H + 3 * H + j] = dht * tc * og * (1.f - og);
                dc[j] = dct * fg;
            }
            __syncthreads();
            for (int j = tid; j < H; j += NT) {
                float a = 0.f, r2 = 0.f;
                for (int r = 0; r < 4 * H; ++r) {
                    const float g = dG[t * 4 * H + r];
                    a += Wih[(size_t)r * H + j] * g;
                    r2 += Whh[(size_t)r * H + j] * g;
                }
                dxin[t * H + j] = a;
                dh[j] = r2;
            }
            __syncthreads();
        }
        // weight gradients of this layer
        for (int i = tid; i < 4 * H * H; i += NT) {
            const int j = i % H, r = i / H;
            float a = 0.f, c2 = 0.f;
            for (int t = 0; t < T; ++t) {
                const float g = dG[t * 4 * H + r];
                a += g * xin[t * H + j];
                if (t > 0) c2 += g * sv[((size_t)l * T + t - 1) * 6 * H + 5 * H + j];
            }
            unsafeAtomicAdd(G.p[4 * l] + i, a);
            unsafeAtomicAdd(G.p[4 * l + 1] + i, c2);
        }
        for (int r = tid; r < 4 * H; r += NT) {
            float a = 0.f;
            for (int t = 0; t < T; ++t) a += dG[t * 4 * H + r];
            unsafeAtomicAdd(G.p[4 * l + 2] + r, a);
            unsafeAtomicAdd(G.p[4 * l + 3] + r, a);
        }
        __syncthreads();
        if (l > 0) {
            for (int i = tid; i < T * H; i += NT) dext[i] = dxin[i] * drop_mask(p, seed, b, l - 1, i);
            __syncthreads();
        } else {
            for (int i = tid; i < T * H; i += NT) dx[(size_t)b * T * H + i] = dxin[i];
        }
    }
}

// ------------------------------------------------------------------------------ token outer products
// The four weight-matrix gradients of an encoder layer, dW[o][j] = sum_n g[n][o] * a[n][j] over the
// n = B*T tokens (g: per-token gradients, a: saved activations, both [n][width] row-major), in
// one launch: a block owns a 64 x 64 tile of one matrix, stages g[:, 64] and a[:, 64] for up to
// 128 tokens at a time in LDS with coalesced loads and accumulates a 4 x 4 micro-tile per thread.
// The reduction is 160 tokens long: as separate GEMM launches these were 25 us apiece, all latency.
struct OuterProb { const float* g; const float* a; float* out; int I, J, tile0; };
struct OuterProbs { OuterProb p[4]; int n; };
constexpr int OT = 64, ONC = 128;

__global__ __launch_bounds__(256) void tok_outer_kernel(OuterProbs P) {
    __shared__ float sg[ONC][OT], sa[ONC][OT];
    int q = 3;
    while (q > 0 && (int)blockIdx.x < P.p[q].tile0) --q;
    const OuterProb pr = P.p[q];
    const int tile = (int)blockIdx.x - pr.tile0, tj = pr.J / OT;
    const int o0 = (tile / tj) * OT, j0 = (tile % tj) * OT;
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int n0 = 0; n0 < P.n; n0 += ONC) {
        const int nc = P.n - n0 < ONC ? P.n - n0 : ONC;
        __syncthreads();
        for (int i = threadIdx.x; i < nc * (OT / 4); i += 256) {
            const int r = i / (OT / 4), c4 = i % (OT / 4);
            *reinterpret_cast<float4*>(&sg[r][4 * c4]) =
                *reinterpret_cast<const float4*>(pr.g + (size_t)(n0 + r) * pr.I + o0 + 4 * c4);
            *reinterpret_cast<float4*>(&sa[r][4 * c4]) =
                *reinterpret_cast<const float4*>(pr.a + (size_t)(n0 + r) * pr.J + j0 + 4 * c4);
        }
        __syncthreads();
        for (int r = 0; r < nc; ++r) {
            const float4 gv = *reinterpret_cast<const float4*>(&sg[r][4 * ty]);
            const float4 av = *reinterpret_cast<const float4*>(&sa[r][4 * tx]);
            const float g4[4] = {gv.x, gv.y, gv.z, gv.w}, a4[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] += g4[i] * a4[j];
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        *reinterpret_cast<float4*>(pr.out + (size_t)(o0 + 4 * ty + i) * pr.J + j0 + 4 * tx) =
            make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
}

}  // namespace

extern "C" size_t sf_transformer_save_floats(int T, int D, int F) { return (size_t)TrSave(T, D, F).total; }
extern "C" int sf_transformer_save_offset(int T, int D, int F, int which) {
    const TrSave L(T, D, F);
    const int offs[9] = {L.qkv, L.probs, L.ctx, L.xhat1, L.rstd1, L.x1, L.hd, L.xhat2, L.rstd2};
    return (which >= 0 && which < 9) ? offs[which] : -1;
}
extern "C" size_t sf_lstm_save_floats(int T, int H, int L) { return (size_t)T * L * 6 * H; }

static int tr_check(const char* who, int B, int T, int D, int F, int nh) {
    if (B <= 0 || T <= 0 || T > MAXT || D <= 0 || D % 4 != 0 || F <= 0 || F % 4 != 0 || nh <= 0 || nh > 8 || D % nh != 0) {
        sf_set_error("%s: unsupported shape B=%d T=%d D=%d F=%d nhead=%d (T<=8, D,F multiples of 4, nhead<=8)", who, B, T, D, F, nh);
        return SF_EINVAL;
    }
    return SF_OK;
}

extern "C" int sf_transformer_layer_fwd(const float* x, float* y, const float* const* w, float* save, int B, int T,
                                        int D, int F, int nhead, float p_drop, unsigned long long seed, void* stream) {
    SF_REQUIRE(x && y && w, "sf_transformer_layer_fwd: null argument");
    if (int rc = tr_check("sf_transformer_layer_fwd", B, T, D, F, nhead)) return rc;
    SF_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "sf_transformer_layer_fwd: bad dropout %f", p_drop);
    Ptrs12 W;
    for (int i = 0; i < 12; ++i) { SF_REQUIRE(w[i], "sf_transformer_layer_fwd: null weight %d", i); W.p[i] = w[i]; }
    const size_t lds = sizeof(float) * ((size_t)T * (4 * D + 3 * D + F) + 8 * T * T);
    SF_REQUIRE(lds <= 64 * 1024, "sf_transformer_layer_fwd: sequence does not fit in LDS");
    hipLaunchKernelGGL(transformer_fwd_kernel, dim3(B), dim3(NT_FWD), lds, (hipStream_t)stream, x, y, W, save, T, D, F,
                       nhead, p_drop, seed);
    SF_LAUNCH_CHECK("sf_transformer_layer_fwd");
    return SF_OK;
}

extern "C" int sf_transformer_layer_bwd(const float* x, const float* dy, float* dx, const float* const* w,
                                        float* const* dw, const float* save, float* gtok, int B, int T, int D, int F,
                                        int nhead, float p_drop, unsigned long long seed, void* stream) {
    SF_REQUIRE(x && dy && dx && w && dw && save && gtok, "sf_transformer_layer_bwd: null argument");
    if (int rc = tr_check("sf_transformer_layer_bwd", B, T, D, F, nhead)) return rc;
    Ptrs12 W; MPtrs12 G;
    for (int i = 0; i < 12; ++i) {
        SF_REQUIRE(w[i] && dw[i], "sf_transformer_layer_bwd: null weight/grad %d", i);
        W.p[i] = w[i]; G.p[i] = dw[i];
    }
    const int j4 = D / 4, parts = j4 >= NT ? 1 : NT / j4;
    const size_t lds = sizeof(float) * ((size_t)T * (2 * D + F + 3 * D) + 8 * T * T + (size_t)parts * MAXT * D);
    SF_REQUIRE(lds <= 160 * 1024 && D % 4 == 0 && F % 4 == 0, "sf_transformer_layer_bwd: sequence does not fit in LDS");
    static DeviceOnce attr_set;
    if (!attr_set.done()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&transformer_bwd_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            sf_set_error("sf_transformer_layer_bwd: cannot raise the dynamic LDS limit");
            return SF_ELAUNCH;
        }
        attr_set.mark();
    }
    hipLaunchKernelGGL(transformer_bwd_kernel, dim3(B), dim3(NT), lds, (hipStream_t)stream, x, dy, dx, W, G, save, gtok,
                       T, D, F, nhead, p_drop, seed);
    SF_LAUNCH_CHECK("sf_transformer_layer_bwd");
    return SF_OK;
}

extern "C" int sf_add_pe(const float* x, const float* pe, float* y, int B, int T, int D, void* stream) {
    SF_REQUIRE(x && pe && y && B > 0 && T > 0 && D > 0, "sf_add_pe: bad arguments");
    const long long total = (long long)B * T * D;
    int grid = (int)((total + NT - 1) / NT); if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(add_pe_kernel, dim3(grid), dim3(NT), 0, (hipStream_t)stream, x, pe, y, total, T * D);
    SF_LAUNCH_CHECK("sf_add_pe");
    return SF_OK;
}
extern "C" int sf_mean_t_fwd(const float* x, float* y, int B, int T, int D, void* stream) {
    SF_REQUIRE(x && y && B > 0 && T > 0 && D > 0, "sf_mean_t_fwd: bad arguments");
    int grid = (B * D + NT - 1) / NT; if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(mean_t_fwd_kernel, dim3(grid), dim3(NT), 0, (hipStream_t)stream, x, y, B, T, D);
    SF_LAUNCH_CHECK("sf_mean_t_fwd");
    return SF_OK;
}
extern "C" int sf_mean_t_bwd(const float* dy, float* dx, int B, int T, int D, void* stream) {
    SF_REQUIRE(dy && dx && B > 0 && T > 0 && D > 0, "sf_mean_t_bwd: bad arguments");
    int grid = (B * T * D + NT - 1) / NT; if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(mean_t_bwd_kernel, dim3(grid), dim3(NT), 0, (hipStream_t)stream, dy, dx, B, T, D);
    SF_LAUNCH_CHECK("sf_mean_t_bwd");
    return SF_OK;
}

static int lstm_check(const char* who, int B, int T, int H, int L) {
    if (B <= 0 || T <= 0 || T > MAXT || H <= 0 || H % 4 != 0 || L <= 0 || L > 4) {
        sf_set_error("%s: unsupported shape B=%d T=%d H=%d L=%d (T<=8, H multiple of 4, L<=4)", who, B, T, H, L);
        return SF_EINVAL;
    }
    return SF_OK;
}

extern "C" int sf_lstm_fwd(const float* x, float* y, const float* const* w, float* save, int B, int T, int H, int L,
                           float p_drop, unsigned long long seed, void* stream) {
    SF_REQUIRE(x && y && w, "sf_lstm_fwd: null argument");
    if (int rc = lstm_check("sf_lstm_fwd", B, T, H, L)) return rc;
    Ptrs16 W;
    for (int i = 0; i < 16; ++i) W.p[i] = i < 4 * L ? w[i] : nullptr;
    for (int i = 0; i < 4 * L; ++i) SF_REQUIRE(w[i], "sf_lstm_fwd: null weight %d", i);
    const size_t lds = sizeof(float) * ((size_t)2 * T * H + 6 * H);
    SF_REQUIRE(lds <= 64 * 1024, "sf_lstm_fwd: sequence does not fit in LDS");
    hipLaunchKernelGGL(lstm_fwd_kernel, dim3(B), dim3(NT), lds, (hipStream_t)stream, x, y, W, save, T, H, L, p_drop, seed);
    SF_LAUNCH_CHECK("sf_lstm_fwd");
    return SF_OK;
}

extern "C" int sf_lstm_bwd(const float* x, const float* dy, float* dx, const float* const* w, float* const* dw,
                           const float* save, int B, int T, int H, int L, float p_drop, unsigned long long seed,
                           void* stream) {
    SF_REQUIRE(x && dy && dx && w && dw && save, "sf_lstm_bwd: null argument");
    if (int rc = lstm_check("sf_lstm_bwd", B, T, H, L)) return rc;
    Ptrs16 W; MPtrs16 G;
    for (int i = 0; i < 16; ++i) { W.p[i] = i < 4 * L ? w[i] : nullptr; G.p[i] = i < 4 * L ? dw[i] : nullptr; }
    for (int i = 0; i < 4 * L; ++i) SF_REQUIRE(w[i] && dw[i], "sf_lstm_bwd: null weight/grad %d", i);
    const size_t lds = sizeof(float) * ((size_t)3 * T * H + (size_t)T * 4 * H + 2 * H);
    SF_REQUIRE(lds <= 64 * 1024, "sf_lstm_bwd: sequence does not fit in LDS");
    hipLaunchKernelGGL(lstm_bwd_kernel, dim3(B), dim3(NT), lds, (hipStream_t)stream, x, dy, dx, W, G, save, T, H, L,
                       p_drop, seed);
    SF_LAUNCH_CHECK("sf_lstm_bwd");
    return SF_OK;
}

extern "C" int sf_token_outer4(const float* const* g, const float* const* a, float* const* out, const int* I,
                               const int* J, int n, void* stream) {
    SF_REQUIRE(g && a && out && I && J && n > 0, "sf_token_outer4: bad arguments");
    OuterProbs P;
    P.n = n;
    int tiles = 0;
    for (int q = 0; q < 4; ++q) {
        SF_REQUIRE(g[q] && a[q] && out[q], "sf_token_outer4: null matrix %d", q);
        SF_REQUIRE(I[q] > 0 && J[q] > 0 && I[q] % OT == 0 && J[q] % OT == 0,
                   "sf_token_outer4: matrix %d is %d x %d, both sizes must be multiples of %d", q, I[q], J[q], OT);
        SF_REQUIRE(((uintptr_t)g[q] | (uintptr_t)a[q] | (uintptr_t)out[q]) % 16 == 0, "sf_token_outer4: matrix %d is not 16-byte aligned", q);
        P.p[q].g = g[q]; P.p[q].a = a[q]; P.p[q].out = out[q]; P.p[q].I = I[q]; P.p[q].J = J[q]; P.p[q].tile0 = tiles;
        tiles += (I[q] / OT) * (J[q] / OT);
    }
    hipLaunchKernelGGL(tok_outer_kernel, dim3(tiles), dim3(256), 0, (hipStream_t)stream, P);
    SF_LAUNCH_CHECK("sf_token_outer4");
    return SF_OK;
}

