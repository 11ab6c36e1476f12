// GRU and temporal attention of the reference's AttentionGRUModel
// (src/model/baseline_models.py:221-247, 289-296, 316-317): nn.GRU(H, H, L, batch_first) returning
// the whole output sequence, and scaled dot-product attention over it averaged over time.
// Like the LSTM / Transformer kernels: one workgroup per sample, the sequence resident in LDS,
// forward and backward each one launch; weight gradients are accumulated with float atomics.
#include "sf_common.h"

namespace {

constexpr int NT = 256;
constexpr int GMAXT = 8;

struct GPtrs16 { const float* p[16]; };
struct GMPtrs16 { float* p[16]; };

__device__ __forceinline__ float sigm(float v) { return 1.0f / (1.0f + expf(-v)); }

__device__ __forceinline__ float dot4(const float* __restrict__ wrow, const float* __restrict__ v, int n) {
    float a = 0.f;
    for (int d = 0; d < n; d += 4) {
        const float4 w = *reinterpret_cast<const float4*>(wrow + d);
        const float4 x = *reinterpret_cast<const float4*>(v + d);   // v is 16-byte aligned (LDS rows of H % 4 == 0 floats)
        a += w.x * x.x + w.y * x.y + w.z * x.z + w.w * x.w;
    }
    return a;
}

// multiplicative dropout mask (0 or 1/(1-p)) of element i of the output of layer `site` of sample b
__device__ __forceinline__ float gru_drop(float p, unsigned long long seed, int b, int site, int i) {
    if (p <= 0.f) return 1.f;
    const unsigned long long idx = ((unsigned long long)b * 16ull + (unsigned long long)site) * 65536ull + (unsigned long long)i;
    return rand_u(seed, idx) >= p ? 1.0f / (1.0f - p) : 0.f;
}

// nn.GRU cell, gate order r, z, n (weight_ih / weight_hh rows [0,H), [H,2H), [2H,3H)):
//   r = s(Wir x + bir + Whr h + bhr), z = s(Wiz x + biz + Whz h + bhz),
//   n = tanh(Win x + bin + r * (Whn h + bhn)), h' = (1 - z) * n + z * h
// save per (layer, step): r, z, n, hl = Whn h + bhn, h'  (5H floats)
__global__ __launch_bounds__(NT) void gru_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, GPtrs16 W,
                                                     float* __restrict__ save, int T, int H, int L, float p,
                                                     unsigned long long seed) {
    extern __shared__ float sm[];
    float* xin = sm;               // [T][H]
    float* hs = xin + T * H;       // [T][H]
    float* h = hs + T * H;         // [H]
    float* gi = h + H;             // [3H] input-side pre-activations
    float* gh = gi + 3 * H;        // [3H] hidden-side pre-activations
    const int b = blockIdx.x, tid = threadIdx.x;
    float* sv = save ? save + (size_t)b * T * L * 5 * H : nullptr;
    for (int i = tid; i < T * H; i += NT) xin[i] = x[(size_t)b * T * H + i];
    for (int l = 0; l < L; ++l) {
        const float* Wih = W.p[4 * l]; const float* Whh = W.p[4 * l + 1];
        const float* bih = W.p[4 * l + 2]; const float* bhh = W.p[4 * l + 3];
        for (int i = tid; i < H; i += NT) h[i] = 0.f;
        __syncthreads();
        for (int t = 0; t < T; ++t) {
            for (int r = tid; r < 3 * H; r += NT) {
                gi[r] = bih[r] + dot4(Wih + (size_t)r * H, xin + t * H, H);
                gh[r] = bhh[r] + dot4(Whh + (size_t)r * H, h, H);
            }
            __syncthreads();
            for (int j = tid; j < H; j += NT) {
                const float rg = sigm(gi[j] + gh[j]), zg = sigm(gi[H + j] + gh[H + j]);
                const float hl = gh[2 * H + j];
                const float ng = tanhf(gi[2 * H + j] + rg * hl);
                const float hn = (1.f - zg) * ng + zg * h[j];
                hs[t * H + j] = hn;
                if (sv) {
                    float* s = sv + ((size_t)l * T + t) * 5 * H;
                    s[j] = rg; s[H + j] = zg; s[2 * H + j] = ng; s[3 * H + j] = hl; s[4 * H + j] = hn;
                }
            }
            __syncthreads();
            for (int j = tid; j < H; j += NT) h[j] = hs[t * H + j];
            __syncthreads();
        }
        if (l + 1 < L) {
            for (int i = tid; i < T * H; i += NT) xin[i] = hs[i] * gru_drop(p, seed, b, l, i);
            __syncthreads();
        }
    }
    for (int i = tid; i < T * H; i += NT) y[(size_t)b * T * H + i] = hs[i];
}

__global__ __launch_bounds__(NT) void gru_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                     float* __restrict__ dx, GPtrs16 W, GMPtrs16 G,
                                                     const float* __restrict__ save, int T, int H, int L, float p,
                                                     unsigned long long seed) {
    extern __shared__ float sm[];
    float* dext = sm;              // [T][H] gradient arriving on this layer's outputs
    float* dxin = dext + T * H;    // [T][H] gradient wrt this layer's inputs
    float* dGi = dxin + T * H;     // [T][3H] input-side pre-activation gradients
    float* dGh = dGi + T * 3 * H;  // [T][3H] hidden-side pre-activation gradients
    float* dh = dGh + T * 3 * H;   // [H] gradient carried to the previous step
    float* dhn = dh + H;           // [H] its next value
    float* xin = dhn + H;          // [T][H] this layer's inputs
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* sv = save + (size_t)b * T * L * 5 * H;
    for (int i = tid; i < T * H; i += NT) dext[i] = dy[(size_t)b * T * H + i];
    __syncthreads();
    for (int l = L - 1; l >= 0; --l) {
        const float* Wih = W.p[4 * l]; const float* Whh = W.p[4 * l + 1];
        for (int i = tid; i < T * H; i += NT) {
            if (l == 0) xin[i] = x[(size_t)b * T * H + i];
            else xin[i] = sv[((size_t)(l - 1) * T + i / H) * 5 * H + 4 * H + i % H] * gru_drop(p, seed, b, l - 1, i);
        }
        for (int i = tid; i < H; i += NT) dh[i] = 0.f;
        __syncthreads();
        for (int t = T - 1; t >= 0; --t) {
            const float* s = sv + ((size_t)l * T + t) * 5 * H;
            for (int j = tid; j < H; j += NT) {
                const float rg = s[j], zg = s[H + j], ng = s[2 * H + j], hl = s[3 * H + j];
                const float hprev = t > 0 ? sv[((size_t)l * T + t - 1) * 5 * H + 4 * H + j] : 0.f;
                const float dht = dext[t * H + j] + dh[j];
                const float dn = dht * (1.f - zg) * (1.f - ng * ng);      // wrt the tanh argument
                const float dz = dht * (hprev - ng) * zg * (1.f - zg);    // wrt the z pre-activation
                const float dr = dn * hl * rg * (1.f - rg);               // wrt the r pre-activation
                dGi[t * 3 * H + j] = dr; dGi[t * 3 * H + H + j] = dz; dGi[t * 3 * H + 2 * H + j] = dn;
                dGh[t * 3 * H + j] = dr; dGh[t * 3 * H + H + j] = dz; dGh[t * 3 * H + 2 * H + j] = dn * rg;
                dhn[j] = dht * zg;
            }
            __syncthreads();
            for (int j = tid; j < H; j += NT) {
                float a = 0.f, r2 = 0.f;
                for (int r = 0; r < 3 * H; ++r) {
                    a += Wih[(size_t)r * H + j] * dGi[t * 3 * H + r];
                    r2 += Whh[(size_t)r * H + j] * dGh[t * 3 * H + r];
                }
                dxin[t * H + j] = a;
                dh[j] = dhn[j] + r2;
            }
            __syncthreads();
        }
        for (int i = tid; i < 3 * H * H; i += NT) {
            const int j = i % H, r = i / H;
            float a = 0.f, c2 = 0.f;
            for (int t = 0; t < T; ++t) {
                a += dGi[t * 3 * H + r] * xin[t * H + j];
                if (t > 0) c2 += dGh[t * 3 * H + r] * sv[((size_t)l * T + t - 1) * 5 * H + 4 * H + j];
            }
            unsafeAtomicAdd(G.p[4 * l] + i, a);
            unsafeAtomicAdd(G.p[4 * l + 1] + i, c2);
        }
        for (int r = tid; r < 3 * H; r += NT) {
            float a = 0.f, c2 = 0.f;
            for (int t = 0; t < T; ++t) { a += dGi[t * 3 * H + r]; c2 += dGh[t * 3 * H + r]; }
            unsafeAtomicAdd(G.p[4 * l + 2] + r, a);
            unsafeAtomicAdd(G.p[4 * l + 3] + r, c2);
        }
        __syncthreads();
        if (l > 0) {
            for (int i = tid; i < T * H; i += NT) dext[i] = dxin[i] * gru_drop(p, seed, b, l - 1, i);
            __syncthreads();
        } else {
            for (int i = tid; i < T * H; i += NT) dx[(size_t)b * T * H + i] = dxin[i];
        }
    }
}

// TemporalAttention.forward: Q, K, V = Linear(g); P = softmax(Q K^T / sqrt(H)); out = mean_t (P V)[t].
// W.p = {Wq, bq, Wk, bk, Wv, bv}.  save per sample: Q, K, V [T][H] each, P [T][T].
__global__ __launch_bounds__(NT) void attn_fwd_kernel(const float* __restrict__ g, float* __restrict__ out, GPtrs16 W,
                                                      float* __restrict__ save, int T, int H) {
    extern __shared__ float sm[];
    float* gs = sm;                 // [T][H]
    float* qkv = gs + T * H;        // [3][T][H]
    float* P = qkv + 3 * T * H;     // [T][T]
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < T * H; i += NT) gs[i] = g[(size_t)b * T * H + i];
    __syncthreads();
    for (int i = tid; i < 3 * T * H; i += NT) {
        const int m = i / (T * H), t = (i / H) % T, o = i % H;
        qkv[i] = W.p[2 * m + 1][o] + dot4(W.p[2 * m] + (size_t)o * H, gs + t * H, H);
    }
    __syncthreads();
    const float inv = 1.0f / sqrtf((float)H);
    for (int i = tid; i < T * T; i += NT) {
        const int t = i / T, s = i % T;
        float a = 0.f;
        for (int d = 0; d < H; ++d) a += qkv[t * H + d] * qkv[T * H + s * H + d];
        P[i] = a * inv;
    }
    __syncthreads();
    if (tid < T) {
        float mx = -INFINITY, den = 0.f;
        for (int s = 0; s < T; ++s) mx = fmaxf(mx, P[tid * T + s]);
        for (int s = 0; s < T; ++s) { const float e = expf(P[tid * T + s] - mx); P[tid * T + s] = e; den += e; }
        for (int s = 0; s < T; ++s) P[tid * T + s] /= den;
    }
    __syncthreads();
    for (int d = tid; d < H; d += NT) {
        float a = 0.f;
        for (int t = 0; t < T; ++t)
            for (int s = 0; s < T; ++s) a += P[t * T + s] * qkv[2 * T * H + s * H + d];
        out[(size_t)b * H + d] = a / (float)T;
    }
    if (save) {
        float* sv = save + (size_t)b * (3 * T * H + T * T);
        for (int i = tid; i < 3 * T * H; i += NT) sv[i] = qkv[i];
        for (int i = tid; i < T * T; i += NT) sv[3 * T * H + i] = P[i];
    }
}

// G.p = {dWq, dbq, dWk, dbk, dWv, dbv} (accumulated with atomics; zeroed by the caller)
__global__ __launch_bounds__(NT) void attn_bwd_kernel(const float* __restrict__ g, const float* __restrict__ dout,
                                                      float* __restrict__ dg, GPtrs16 W, GMPtrs16 G,
                                                      const float* __restrict__ save, int T, int H) {
    extern __shared__ float sm[];
    float* gs = sm;                  // [T][H]
    float* qkv = gs + T * H;         // [3][T][H]
    float* dqkv = qkv + 3 * T * H;   // [3][T][H]
    float* P = dqkv + 3 * T * H;     // [T][T]
    float* dS = P + T * T;           // [T][T]
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* sv = save + (size_t)b * (3 * T * H + T * T);
    for (int i = tid; i < T * H; i += NT) gs[i] = g[(size_t)b * T * H + i];
    for (int i = tid; i < 3 * T * H; i += NT) qkv[i] = sv[i];
    for (int i = tid; i < T * T; i += NT) P[i] = sv[3 * T * H + i];
    __syncthreads();
    const float* Q = qkv; const float* K = qkv + T * H; const float* V = qkv + 2 * T * H;
    float* dQ = dqkv; float* dK = dqkv + T * H; float* dV = dqkv + 2 * T * H;
    const float* dob = dout + (size_t)b * H;
    const float invT = 1.0f / (float)T, inv = 1.0f / sqrtf((float)H);
    // dctx[t] = dout / T for every t
    for (int i = tid; i < T * T; i += NT) {          // dP[t][s] = dctx . V[s]
        const int s = i % T;
        float a = 0.f;
        for (int d = 0; d < H; ++d) a += dob[d] * V[s * H + d];
        dS[i] = a * invT;
    }
    for (int i = tid; i < T * H; i += NT) {          // dV[s] = sum_t P[t][s] dctx
        const int s = i / H, d = i % H;
        float a = 0.f;
        for (int t = 0; t < T; ++t) a += P[t * T + s];
        dV[i] = a * dob[d] * invT;
    }
    __syncthreads();
    if (tid < T) {                                   // softmax backward, then the 1/sqrt(H) scale
        float dot = 0.f;
        for (int s = 0; s < T; ++s) dot += P[tid * T + s] * dS[tid * T + s];
        for (int s = 0; s < T; ++s) dS[tid * T + s] = P[tid * T + s] * (dS[tid * T + s] - dot) * inv;
    }
    __syncthreads();
    for (int i = tid; i < T * H; i += NT) {
        const int t = i / H, d = i % H;
        float a = 0.f, c = 0.f;
        for (int s = 0; s < T; ++s) { a += dS[t * T + s] * K[s * H + d]; c += dS[s * T + t] * Q[s * H + d]; }
        dQ[i] = a;
        dK[i] = c;
    }
    __syncthreads();
    for (int i = tid; i < 3 * H * H; i += NT) {      // dW_m[o][j] = sum_t d(m)[t][o] g[t][j]
        const int m = i / (H * H), o = (i / H) % H, j = i % H;
        float a = 0.f;
        for (int t = 0; t < T; ++t) a += dqkv[m * T * H + t * H + o] * gs[t * H + j];
        unsafeAtomicAdd(G.p[2 * m] + o * H + j, a);
    }
    for (int i = tid; i < 3 * H; i += NT) {
        const int m = i / H, o = i % H;
        float a = 0.f;
        for (int t = 0; t < T; ++t) a += dqkv[m * T * H + t * H + o];
        unsafeAtomicAdd(G.p[2 * m + 1] + o, a);
    }
    for (int i = tid; i < T * H; i += NT) {          // dg[t][j] = sum_m sum_o d(m)[t][o] W_m[o][j]
        const int t = i / H, j = i % H;
        float a = 0.f;
        for (int m = 0; m < 3; ++m)
            for (int o = 0; o < H; ++o) a += dqkv[m * T * H + t * H + o] * W.p[2 * m][(size_t)o * H + j];
        dg[(size_t)b * T * H + i] = a;
    }
}

int shape_check(const char* who, int B, int T, int H, int L) {
    if (B <= 0 || T <= 0 || T > GMAXT || H <= 0 || H % 4 != 0 || L <= 0 || L > 4) {
        sf_set_error("%s: unsupported shape B=%d T=%d H=%d L=%d (T<=8, H multiple of 4, L<=4)", who, B, T, H, L);
        return SF_EINVAL;
    }
    return SF_OK;
}

}  // namespace

extern "C" size_t sf_gru_save_floats(int T, int H, int L) { return (size_t)T * L * 5 * H; }
extern "C" size_t sf_attn_save_floats(int T, int H) { return (size_t)3 * T * H + (size_t)T * T; }

extern "C" int sf_gru_fwd(const float* x, float* y, const float* const* w, float* save, int B, int T, int H, int L,
                          float p_drop, unsigned long long seed, void* stream) {
    SF_REQUIRE(x && y && w, "sf_gru_fwd: null argument");
    if (int rc = shape_check("sf_gru_fwd", B, T, H, L)) return rc;
    GPtrs16 W;
    for (int i = 0; i < 16; ++i) W.p[i] = i < 4 * L ? w[i] : nullptr;
    for (int i = 0; i < 4 * L; ++i) SF_REQUIRE(w[i], "sf_gru_fwd: null weight %d", i);
    const size_t lds = sizeof(float) * ((size_t)2 * T * H + 7 * H);
    SF_REQUIRE(lds <= 64 * 1024, "sf_gru_fwd: sequence does not fit in LDS");
    hipLaunchKernelGGL(gru_fwd_kernel, dim3(B), dim3(NT), lds, (hipStream_t)stream, x, y, W, save, T, H, L, p_drop, seed);
    SF_LAUNCH_CHECK("sf_gru_fwd");
    return SF_OK;
}

extern "C" int sf_gru_bwd(const float* x, const float* dy, float* dx, const float* const* w, float* const* dw,
                          const float* save, int B, int T, int H, int L, float p_drop, unsigned long long seed,
                          void* stream) {
    SF_REQUIRE(x && dy && dx && w && dw && save, "sf_gru_bwd: null argument");
    if (int rc = shape_check("sf_gru_bwd", B, T, H, L)) return rc;
    GPtrs16 W;
    GMPtrs16 G;
    for (int i = 0; i < 16; ++i) { W.p[i] = i < 4 * L ? w[i] : nullptr; G.p[i] = i < 4 * L ? dw[i] : nullptr; }
    for (int i = 0; i < 4 * L; ++i) SF_REQUIRE(w[i] && dw[i], "sf_gru_bwd: null weight/grad %d", i);
    const size_t lds = sizeof(float) * ((size_t)3 * T * H + 6 * T * H + 2 * H);
    SF_REQUIRE(lds <= 64 * 1024, "sf_gru_bwd: sequence does not fit in LDS");
    hipLaunchKernelGGL(gru_bwd_kernel, dim3(B), dim3(NT), lds, (hipStream_t)stream, x, dy, dx, W, G, save, T, H, L,
                       p_drop, seed);
    SF_LAUNCH_CHECK("sf_gru_bwd");
    return SF_OK;
}

extern "C" int sf_attn_fwd(const float* g, float* out, const float* const* w, float* save, int B, int T, int H,
                           void* stream) {
    SF_REQUIRE(g && out && w, "sf_attn_fwd: null argument");
    if (int rc = shape_check("sf_attn_fwd", B, T, H, 1)) return rc;
    GPtrs16 W;
    for (int i = 0; i < 16; ++i) W.p[i] = i < 6 ? w[i] : nullptr;
    for (int i = 0; i < 6; ++i) SF_REQUIRE(w[i], "sf_attn_fwd: null weight %d", i);
    const size_t lds = sizeof(float) * ((size_t)4 * T * H + T * T);
    SF_REQUIRE(lds <= 64 * 1024, "sf_attn_fwd: sequence does not fit in LDS");
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(B), dim3(NT), lds, (hipStream_t)stream, g, out, W, save, T, H);
    SF_LAUNCH_CHECK("sf_attn_fwd");
    return SF_OK;
}

extern "C" int sf_attn_bwd(const float* g, const float* dout, float* dg, const float* const* w, float* const* dw,
                           const float* save, int B, int T, int H, void* stream) {
    SF_REQUIRE(g && dout && dg && w && dw && save, "sf_attn_bwd: null argument");
    if (int rc = shape_check("sf_attn_bwd", B, T, H, 1)) return rc;
    GPtrs16 W;
    GMPtrs16 G;
    for (int i = 0; i < 16; ++i) { W.p[i] = i < 6 ? w[i] : nullptr; G.p[i] = i < 6 ? dw[i] : nullptr; }
    for (int i = 0; i < 6; ++i) SF_REQUIRE(w[i] && dw[i], "sf_attn_bwd: null weight/grad %d", i);
    const size_t lds = sizeof(float) * ((size_t)7 * T * H + 2 * T * T);
    SF_REQUIRE(lds <= 64 * 1024, "sf_attn_bwd: sequence does not fit in LDS");
    hipLaunchKernelGGL(attn_bwd_kernel, dim3(B), dim3(NT), lds, (hipStream_t)stream, g, dout, dg, W, G, save, T, H);
    SF_LAUNCH_CHECK("sf_attn_bwd");
    return SF_OK;
}
