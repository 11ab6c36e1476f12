// MFMA GEMM / implicit-GEMM convolution family (see sf_gemm in include/sfhip.h).
//
// D[i][j] = sum_r A(i,r) * B(j,r).  Block tile TI x 128 (i x j), 4 waves as 2 x 2, each wave
// (TI/2) x 64 built from 32x32 MFMA tiles.  Operand tiles go global -> registers -> LDS
// (double-buffered, one barrier per reduction step); a tile is either "row-major"
// [row][r] (fragments = contiguous r) or "reduction-major" [r][row] (fragments read through
// ds_read_b64_tr_b16 for bf16, plain b32 reads for f32).  The B operand can be a conv gather
// of NHWC pixels (zero padding by predication).
#include "sf_common.h"
#include "../../include/sfhip.h"

namespace {

constexpr int TJ = 128;
constexpr int NTHREADS = 256;

template <typename T> struct Mma;
template <> struct Mma<bf16> {
    static constexpr int KSTEP = 16;
    static constexpr int BK = 32;
    static constexpr int PAD_RM = 8;   // row-major tile row padding (elements)
    static constexpr int PAD_KM = 32;  // reduction-major tile line padding
    using Frag = bf16x8;
    static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    // row-major tile: lane (r=lane&31, h=lane>>5) holds elements r0..r0+7, r0 = ks*16 + 8h
    static __device__ __forceinline__ Frag load_rm(const bf16* t, int ld, int row0, int ks, int lane) {
        return *reinterpret_cast<const bf16x8*>(&t[(row0 + (lane & 31)) * ld + ks * 16 + 8 * (lane >> 5)]);
    }
    // reduction-major tile [line][col]: the lane needs 8 consecutive lines at one column ->
    // two transposed 4x16 block reads.
    static __device__ __forceinline__ Frag load_km(const bf16* t, int ld, int col0, int ks, int lane) {
        const int line = ks * 16 + 8 * (lane >> 5) + ((lane & 15) >> 2);
        const int col = col0 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
        typedef bf16x4 __attribute__((address_space(3))) * lds_p;
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(&t[line * ld + col]));
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(&t[(line + 4) * ld + col]));
        Frag f;
        f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
        f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
        return f;
    }
};
template <> struct Mma<float> {
    static constexpr int KSTEP = 2;
    static constexpr int BK = 16;
    static constexpr int PAD_RM = 1;
    static constexpr int PAD_KM = 4;
    using Frag = float;
    static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ Frag load_rm(const float* t, int ld, int row0, int ks, int lane) {
        return t[(row0 + (lane & 31)) * ld + ks * 2 + (lane >> 5)];
    }
    static __device__ __forceinline__ Frag load_km(const float* t, int ld, int col0, int ks, int lane) {
        return t[(ks * 2 + (lane >> 5)) * ld + col0 + (lane & 31)];
    }
};

struct KParams {
    const void* A;
    const void* B;
    void* out;
    const float* bias;
    const void* aux;
    long long I, J, R, lda, ldb, ldo;
    int niters, iters_per_split, nsplit, nsplit_req, ni_tiles, nj_tiles;
    int j_fast;   // consecutive workgroups walk the j tiles of one i tile (they share its A tile in L2) instead of the i tiles of one j tile
    int b_gather, out_mode, out_dtype, act, aux_mode, atomic;
    long long split_stride;   // > 0: split s stores its partial result at out + s * split_stride (no atomics)
    unsigned long long dy_pack, dx_pack;  // tap offsets + 8, 4 bits per tap
    sf_conv_geom g;
};

template <typename T, bool ALIGNED> __device__ __forceinline__ void lds_store16(T* dst, uint4 v) {
    if constexpr (ALIGNED) {
        *reinterpret_cast<uint4*>(dst) = v;
    } else {
        uint32_t* d = reinterpret_cast<uint32_t*>(dst);
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
}

__device__ __attribute__((aligned(16))) uint4 g_zero_chunk;  // never written: stays zero

// tap offset t of a packed table (4 bits per tap, biased by 8): no memory access
__device__ __forceinline__ int tap_of(unsigned long long pack, int t) { return (int)((pack >> (4 * t)) & 15) - 8; }

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, rem = nwg & 7, x = bid & 7, idx = bid >> 3;
    return (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + idx;
}

template <typename T, bool A_KM, bool B_KM, int TI, int BK>
__device__ __forceinline__ void gemm_body(const KParams& p, const int block_id, const int nwg) {
    using M = Mma<T>;
    constexpr int EPC = ElemTraits<T>::EPC;
    constexpr int LDA = A_KM ? (TI + M::PAD_KM) : (BK + M::PAD_RM);
    constexpr int LDB = B_KM ? (TJ + M::PAD_KM) : (BK + M::PAD_RM);
    constexpr int A_ELEMS = A_KM ? BK * LDA : TI * LDA;
    constexpr int B_ELEMS = B_KM ? BK * LDB : TJ * LDB;
    constexpr int A_CHUNKS = TI * BK / EPC;   // 16-byte chunks per A tile
    constexpr int B_CHUNKS = TJ * BK / EPC;
    constexpr int NA = (A_CHUNKS + NTHREADS - 1) / NTHREADS;
    constexpr int NB = B_CHUNKS / NTHREADS;
    constexpr int A_CPL = A_KM ? TI / EPC : BK / EPC;  // chunks per stored line
    constexpr int B_CPL = B_KM ? TJ / EPC : BK / EPC;
    constexpr bool A_AL = A_KM || (sizeof(T) == 2);
    constexpr bool B_AL = B_KM || (sizeof(T) == 2);
    constexpr int MT = TI / 64;  // 32-row MFMA tiles per wave along i
    constexpr int NT = 2;        // along j
    static_assert(A_CHUNKS % NTHREADS == 0 || A_CHUNKS < NTHREADS, "tile/threads mismatch");

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* smem = reinterpret_cast<T*>(smem_raw);  // 2 * (A_ELEMS + B_ELEMS) elements

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;

    // block -> (i tile, j tile, split)
    int lin = xcd_remap(block_id, nwg);
    int it_i, it_j;
    if (p.j_fast) {
        it_j = lin % p.nj_tiles;
        lin /= p.nj_tiles;
        it_i = lin % p.ni_tiles;
        lin /= p.ni_tiles;
    } else {
        it_i = lin % p.ni_tiles;
        lin /= p.ni_tiles;
        it_j = lin % p.nj_tiles;
        lin /= p.nj_tiles;
    }
    const int split = lin;
    const long long i0 = (long long)it_i * TI, j0 = (long long)it_j * TJ;
    const int iter_begin = split * p.iters_per_split;
    int iter_end = iter_begin + p.iters_per_split;
    if (iter_end > p.niters) iter_end = p.niters;

    const T* __restrict__ Ag = reinterpret_cast<const T*>(p.A);
    const T* __restrict__ Bg = reinterpret_cast<const T*>(p.B);
    const sf_conv_geom& g = p.g;

    // ---- per-thread chunk bookkeeping -------------------------------------------------
    // A chunks
    int a_line[NA], a_cc[NA];
#pragma unroll
    for (int c = 0; c < NA; ++c) {
        const int ch = tid + c * NTHREADS;
        a_line[c] = ch / A_CPL;
        a_cc[c] = ch % A_CPL;
    }
    // B chunks
    int b_line[NB], b_cc[NB];
    // gather state: row-major gather -> fixed pixel per chunk; reduction-major gather -> fixed
    // (tap, channel) per chunk and a pixel that advances by BK every iteration.
    int gp_n[NB], gp_a[NB], gp_b[NB], gp_t[NB], gp_c[NB];
    bool gp_ok[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) {
        const int ch = tid + c * NTHREADS;
        b_line[c] = ch / B_CPL;
        b_cc[c] = ch % B_CPL;
        gp_n[c] = gp_a[c] = gp_b[c] = gp_t[c] = gp_c[c] = 0;
        gp_ok[c] = true;
        if (p.b_gather) {
            if constexpr (!B_KM) {
                const long long j = j0 + b_line[c];
                gp_ok[c] = j < p.J;
                const long long jj = gp_ok[c] ? j : 0;
                const int per = g.MH * g.MW;
                gp_n[c] = (int)(jj / per);
                const int rem = (int)(jj - (long long)gp_n[c] * per);
                gp_a[c] = rem / g.MW;
                gp_b[c] = rem - gp_a[c] * g.MW;
            } else {
                const long long jcol = j0 + (long long)b_cc[c] * EPC;
                gp_ok[c] = jcol < p.J;
                const int jj = gp_ok[c] ? (int)jcol : 0;
                gp_t[c] = jj / g.C;
                gp_c[c] = jj - gp_t[c] * g.C;
                const long long m = (long long)iter_begin * BK + b_line[c];
                const int per = g.MH * g.MW;
                gp_n[c] = (int)(m / per);
                const int rem = (int)(m - (long long)gp_n[c] * per);
                gp_a[c] = rem / g.MW;
                gp_b[c] = rem - gp_a[c] * g.MW;
            }
        }
    }
    const bool c_div_bk = (g.C % BK) == 0;

    uint4 ra[NA], rb[NB];

    // Loads never sit behind a branch (the compiler would make each wait for the previous one):
    // a chunk that is out of range or zero padding reads the zero chunk instead.
    const T* zsrc = reinterpret_cast<const T*>(&g_zero_chunk);
    auto load_tiles = [&](int it) {
        const long long r0 = (long long)it * BK;
        // ---- A
#pragma unroll
        for (int c = 0; c < NA; ++c) {
            const T* src = zsrc;
            if (NA * NTHREADS == A_CHUNKS || tid + c * NTHREADS < A_CHUNKS) {
                if constexpr (!A_KM) {
                    const long long i = i0 + a_line[c];
                    const long long r = r0 + (long long)a_cc[c] * EPC;
                    src = (i < p.I && r < p.R) ? Ag + i * p.lda + r : zsrc;
                } else {
                    const long long r = r0 + a_line[c];
                    const long long i = i0 + (long long)a_cc[c] * EPC;
                    src = (r < p.R && i < p.I) ? Ag + r * p.lda + i : zsrc;
                }
            }
            ra[c] = *reinterpret_cast<const uint4*>(src);
        }
        // ---- B
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            const T* src = zsrc;
            if constexpr (!B_KM) {
                const long long r = r0 + (long long)b_cc[c] * EPC;
                if (!p.b_gather) {
                    const long long j = j0 + b_line[c];
                    src = (j < p.J && r < p.R) ? Bg + j * p.ldb + r : zsrc;
                } else {
                    int t, cc;
                    if (c_div_bk) {
                        t = (int)((unsigned)r0 / (unsigned)g.C);
                        cc = (int)r - t * g.C;
                    } else {
                        t = (int)((unsigned)r / (unsigned)g.C);
                        cc = (int)r - t * g.C;
                    }
                    const int iy = gp_a[c] * g.sh + tap_of(p.dy_pack, t);
                    const int ix = gp_b[c] * g.sw + tap_of(p.dx_pack, t);
                    const bool ok = gp_ok[c] && r < p.R && iy >= 0 && iy < g.IH && ix >= 0 && ix < g.IW;
                    const long long pix = ((long long)gp_n[c] * g.IH + iy) * g.IW + ix;
                    src = ok ? Bg + pix * g.C + cc : zsrc;
                }
            } else {
                const long long r = r0 + b_line[c];
                if (!p.b_gather) {
                    const long long j = j0 + (long long)b_cc[c] * EPC;
                    src = (r < p.R && j < p.J) ? Bg + r * p.ldb + j : zsrc;
                } else {
                    const int iy = gp_a[c] * g.sh + tap_of(p.dy_pack, gp_t[c]);
                    const int ix = gp_b[c] * g.sw + tap_of(p.dx_pack, gp_t[c]);
                    const bool ok = gp_ok[c] && r < p.R && iy >= 0 && iy < g.IH && ix >= 0 && ix < g.IW;
                    const long long pix = ((long long)gp_n[c] * g.IH + iy) * g.IW + ix;
                    src = ok ? Bg + pix * g.C + gp_c[c] : zsrc;
                }
            }
            rb[c] = *reinterpret_cast<const uint4*>(src);
        }
        if constexpr (B_KM) {
            if (p.b_gather) {
                // advance every chunk's pixel by BK grid positions
#pragma unroll
                for (int c = 0; c < NB; ++c) {
                    gp_b[c] += BK;
                    while (gp_b[c] >= g.MW) { gp_b[c] -= g.MW; gp_a[c] += 1; }
                    while (gp_a[c] >= g.MH) { gp_a[c] -= g.MH; gp_n[c] += 1; }
                }
            }
        }
    };

    auto store_tiles = [&](int buf) {
        T* sa = smem + buf * (A_ELEMS + B_ELEMS);
        T* sb = sa + A_ELEMS;
#pragma unroll
        for (int c = 0; c < NA; ++c) {
            if (NA * NTHREADS == A_CHUNKS || tid + c * NTHREADS < A_CHUNKS)
                lds_store16<T, A_AL>(sa + a_line[c] * LDA + a_cc[c] * EPC, ra[c]);
        }
#pragma unroll
        for (int c = 0; c < NB; ++c) lds_store16<T, B_AL>(sb + b_line[c] * LDB + b_cc[c] * EPC, rb[c]);
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    if (iter_begin < iter_end) {
        load_tiles(iter_begin);
        for (int it = iter_begin; it < iter_end; ++it) {
            const int buf = (it - iter_begin) & 1;
            store_tiles(buf);
            __syncthreads();
            if (it + 1 < iter_end) load_tiles(it + 1);
            const T* sa = smem + buf * (A_ELEMS + B_ELEMS);
            const T* sb = sa + A_ELEMS;
#pragma unroll
            for (int ks = 0; ks < BK / M::KSTEP; ++ks) {
                typename M::Frag fa[MT], fb[NT];
#pragma unroll
                for (int a = 0; a < MT; ++a) {
                    const int row0 = wi * (TI / 2) + a * 32;
                    fa[a] = A_KM ? M::load_km(sa, LDA, row0, ks, lane) : M::load_rm(sa, LDA, row0, ks, lane);
                }
#pragma unroll
                for (int b = 0; b < NT; ++b) {
                    const int row0 = wj * 64 + b * 32;
                    fb[b] = B_KM ? M::load_km(sb, LDB, row0, ks, lane) : M::load_rm(sb, LDB, row0, ks, lane);
                }
#pragma unroll
                for (int a = 0; a < MT; ++a)
#pragma unroll
                    for (int b = 0; b < NT; ++b) acc[a][b] = M::mma(fa[a], fb[b], acc[a][b]);
            }
        }
    }

    // ---- epilogue: lane holds column j = lane&31, rows i = 8g + 4h + e ------------------
    const int h = lane >> 5;
    const bool atomic = p.atomic != 0;
    const bool add_bias = p.bias != nullptr && split == 0;
    void* const outp = p.split_stride > 0 ? (void*)(reinterpret_cast<float*>(p.out) + (long long)split * p.split_stride) : p.out;
#pragma unroll
    for (int b = 0; b < NT; ++b) {
        const long long j = j0 + wj * 64 + b * 32 + (lane & 31);
        if (j >= p.J) continue;
        long long jloc = j;  // output "pixel" index
        int tap = 0, cch = 0;
        if (p.out_mode == SF_OUT_NHWC) {
            const int per = g.MH * g.MW;
            const int n = (int)(j / per);
            const int rem = (int)(j - (long long)n * per);
            const int a = rem / g.MW, bb = rem - a * g.MW;
            jloc = ((long long)n * g.OH + (a * g.osh + g.ooy)) * g.OW + (bb * g.osw + g.oox);
        } else if (p.out_mode == SF_OUT_OIHW) {
            tap = (int)(j / g.C);
            cch = (int)(j - (long long)tap * g.C);
            if (cch >= g.cin_real) continue;
        }
#pragma unroll
        for (int a = 0; a < MT; ++a) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const long long i = i0 + wi * (TI / 2) + a * 32 + 8 * gq + 4 * h;
                if (i >= p.I) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[a][b][4 * gq + e];
                if (p.out_mode == SF_OUT_NHWC || p.out_mode == SF_OUT_JI) {
                    const long long off = jloc * p.ldo + i;
                    if (add_bias) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += p.bias[i + e];
                    }
                    if (atomic) {
                        float* o = reinterpret_cast<float*>(outp) + off;
#pragma unroll
                        for (int e = 0; e < 4; ++e) unsafeAtomicAdd(o + e, v[e]);
                    } else {
                        if (p.act == SF_ACT_RELU) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = v[e] < 0.f ? 0.f : v[e];   // keeps NaN, like torch.relu
                        } else if (p.act == SF_ACT_LRELU) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.2f * v[e];
                        }
                        if (p.aux_mode) {
                            Vec4<T> ax;
                            ax.load(reinterpret_cast<const T*>(p.aux) + off);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const bool pos = ax.get(e) > 0.f;
                                v[e] = pos ? v[e] : (p.aux_mode == 1 ? 0.f : 0.2f * v[e]);
                            }
                        }
                        if (p.out_dtype == SF_F32) {
                            *reinterpret_cast<float4*>(reinterpret_cast<float*>(outp) + off) =
                                make_float4(v[0], v[1], v[2], v[3]);
                        } else {
                            Vec4<bf16> o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) o.set(e, v[e]);
                            o.store(reinterpret_cast<bf16*>(outp) + off);
                        }
                    }
                } else if (p.out_mode == SF_OUT_IJ) {
                    float* o = reinterpret_cast<float*>(outp);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const long long off = (i + e) * p.ldo + j;
                        if (atomic) unsafeAtomicAdd(o + off, v[e]); else o[off] = v[e];
                    }
                } else {  // SF_OUT_OIHW (always accumulated atomically)
                    float* o = reinterpret_cast<float*>(outp);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (i + e >= p.ldo) continue;  // ldo = real output-channel count of the OIHW tensor
                        const long long off = ((i + e) * g.cin_real + cch) * g.RS + g.tap_rs[tap];
                        unsafeAtomicAdd(o + off, v[e]);
                    }
                }
            }
        }
    }
}

template <typename T, bool A_KM, bool B_KM, int TI, int BK = Mma<T>::BK>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(KParams p) {
    gemm_body<T, A_KM, B_KM, TI, BK>(p, blockIdx.x, gridDim.x);
}

// Up to four independent problems of one kernel shape in a single launch (the four parity
// classes of a stride-2 data gradient): workgroups [first[c], first[c+1]) work on problem c.
constexpr int MAX_BATCH = 4;
struct KBatch {
    KParams k[MAX_BATCH];
    int first[MAX_BATCH + 1];
    int n;
};

template <typename T, int TI>
__global__ __launch_bounds__(NTHREADS) void gemm_batch_kernel(KBatch b) {
    int c = 0;
    while (c + 1 < b.n && (int)blockIdx.x >= b.first[c + 1]) ++c;
    gemm_body<T, false, false, TI, Mma<T>::BK>(b.k[c], blockIdx.x - b.first[c], b.first[c + 1] - b.first[c]);
}

template <typename T, bool A_KM, bool B_KM, int TI, int BK>
int launch_one(const KParams& q, hipStream_t s) {
    using M = Mma<T>;
    constexpr int LDA = A_KM ? (TI + M::PAD_KM) : (BK + M::PAD_RM);
    constexpr int LDB = B_KM ? (TJ + M::PAD_KM) : (BK + M::PAD_RM);
    constexpr size_t lds = sizeof(T) * 2 * ((A_KM ? BK * LDA : TI * LDA) + (B_KM ? BK * LDB : TJ * LDB));
    if (lds > 64 * 1024) {
        static DeviceOnce attr_set;
        if (!attr_set.done()) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<T, A_KM, B_KM, TI, BK>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
                sf_set_error("sf_gemm: cannot raise the dynamic LDS limit to %zu bytes", lds);
                return SF_ELAUNCH;
            }
            attr_set.mark();
        }
    }
    KParams p = q;
    p.niters = ceil_div(p.R, BK);
    p.nsplit = p.nsplit_req > p.niters ? p.niters : p.nsplit_req;
    p.iters_per_split = ceil_div(p.niters, p.nsplit);
    p.nsplit = ceil_div(p.niters, p.iters_per_split);
    p.ni_tiles = ceil_div(p.I, TI);
    p.nj_tiles = ceil_div(p.J, TJ);
    // the operand with the many tiles is the one worth sharing: a 262144 x 128 weight against 160 rows
    // has 2048 i tiles and two j tiles, and i-fastest order read every weight tile from HBM twice
    p.j_fast = p.nj_tiles < p.ni_tiles ? 1 : 0;
    const long long nblk = (long long)p.ni_tiles * p.nj_tiles * p.nsplit;
    if (nblk <= 0 || nblk > 0x7fffffffLL) {
        sf_set_error("sf_gemm: bad grid (%lld blocks)", nblk);
        return SF_EINVAL;
    }
    if (p.split_stride > 0 && p.nsplit < p.nsplit_req) {
        // fewer slices than asked for have work: the caller sums all nsplit_req of them
        if (hipMemsetAsync(reinterpret_cast<float*>(p.out) + (long long)p.nsplit * p.split_stride, 0,
                           sizeof(float) * (size_t)(p.nsplit_req - p.nsplit) * p.split_stride, s) != hipSuccess) {
            sf_set_error("sf_gemm: memset of the unused split slices failed");
            return SF_ELAUNCH;
        }
    }
    hipLaunchKernelGGL((gemm_kernel<T, A_KM, B_KM, TI, BK>), dim3((unsigned)nblk), dim3(NTHREADS), lds, s, p);
    SF_LAUNCH_CHECK("sf_gemm");
    return SF_OK;
}

template <typename T, bool A_KM, bool B_KM>
int launch(const KParams& p, hipStream_t s) {
    constexpr int BK0 = Mma<T>::BK;
    // (a 64-deep reduction step was measured slower for the PatchGAN shapes: its 72 KB of LDS
    // leave one or two workgroups per CU and the exposed latency outweighs the saved barriers)
    if (p.I <= 64) return launch_one<T, A_KM, B_KM, 64, BK0>(p, s);
    return launch_one<T, A_KM, B_KM, 128, BK0>(p, s);
}

template <typename T> int dispatch(const sf_gemm_desc* d, const KParams& p, hipStream_t s) {
    if (!d->a_kmajor && !d->b_kmajor) return launch<T, false, false>(p, s);
    if (d->a_kmajor && d->b_kmajor) return launch<T, true, true>(p, s);
    if (d->a_kmajor && !d->b_kmajor) return launch<T, true, false>(p, s);
    sf_set_error("sf_gemm: a row-major / b reduction-major is not built");
    return SF_ENOSYS;
}

}  // namespace

static int fill_kparams(const sf_gemm_desc* d, const void* A, const void* B, const float* bias, const void* aux,
                        void* out, KParams& p) {
    SF_REQUIRE(d && A && B && out, "sf_gemm: null argument");
    SF_REQUIRE(d->dtype == SF_F32 || d->dtype == SF_BF16, "sf_gemm: bad dtype %d", d->dtype);
    const int epc = d->dtype == SF_BF16 ? 8 : 4;
    const int bk = d->dtype == SF_BF16 ? 32 : 16;
    SF_REQUIRE(d->I > 0 && d->J > 0 && d->R > 0, "sf_gemm: empty problem");
    SF_REQUIRE(d->I % 4 == 0, "sf_gemm: I=%lld must be a multiple of 4", d->I);
    SF_REQUIRE(d->nsplit >= 1, "sf_gemm: nsplit < 1");
    SF_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, "sf_gemm: operands must be 16-byte aligned");
    if (d->a_kmajor) SF_REQUIRE(d->I % epc == 0 && d->lda % epc == 0, "sf_gemm: reduction-major A needs I, lda %% %d == 0", epc);
    else SF_REQUIRE(d->R % epc == 0 && d->lda % epc == 0, "sf_gemm: row-major A needs R, lda %% %d == 0", epc);
    if (d->b_gather) {
        const sf_conv_geom& g = d->g;
        SF_REQUIRE(g.ntaps >= 1 && g.ntaps <= 16, "sf_gemm: ntaps out of range");
        SF_REQUIRE(g.C % epc == 0, "sf_gemm: gathered channel count %d must be a multiple of %d", g.C, epc);
        SF_REQUIRE(g.N > 0 && g.MH > 0 && g.MW > 0 && g.IH > 0 && g.IW > 0, "sf_gemm: bad geometry");
        const long long rows = (long long)g.N * g.MH * g.MW;
        if (!d->b_kmajor) {
            SF_REQUIRE(d->J == rows, "sf_gemm: J=%lld != N*MH*MW=%lld", d->J, rows);
            SF_REQUIRE(d->R == (long long)g.ntaps * g.C, "sf_gemm: R != ntaps*C");
        } else {
            SF_REQUIRE(d->R == rows, "sf_gemm: R=%lld != N*MH*MW=%lld", d->R, rows);
            SF_REQUIRE(d->J == (long long)g.ntaps * g.C, "sf_gemm: J != ntaps*C");
        }
    } else if (d->b_kmajor) {
        SF_REQUIRE(d->J % epc == 0 && d->ldb % epc == 0, "sf_gemm: reduction-major B needs J, ldb %% %d == 0", epc);
    } else {
        SF_REQUIRE(d->R % epc == 0 && d->ldb % epc == 0, "sf_gemm: row-major B needs R, ldb %% %d == 0", epc);
    }
    if (d->out_mode == SF_OUT_NHWC) {
        const sf_conv_geom& g = d->g;
        SF_REQUIRE(d->b_gather && !d->b_kmajor, "sf_gemm: NHWC output needs a row-major gathered B");
        SF_REQUIRE((g.MH - 1) * g.osh + g.ooy < g.OH && (g.MW - 1) * g.osw + g.oox < g.OW && g.ooy >= 0 && g.oox >= 0,
                   "sf_gemm: output placement outside the output tensor");
        SF_REQUIRE(d->ldo >= d->I && d->ldo % 4 == 0, "sf_gemm: ldo");
    }
    if (d->out_mode == SF_OUT_OIHW) SF_REQUIRE(d->b_gather && d->b_kmajor && d->g.RS >= 1 && d->g.cin_real >= 1 && d->ldo >= 1 && d->ldo <= d->I, "sf_gemm: OIHW output needs a reduction-major gathered B and ldo = real output channels");
    if (d->out_mode == SF_OUT_JI) SF_REQUIRE(d->ldo % 4 == 0, "sf_gemm: ldo %% 4");
    if (d->nsplit > 1) SF_REQUIRE(d->out_dtype == SF_F32 && d->act == 0 && d->aux_mode == 0, "sf_gemm: split reduction needs plain f32 output");
    if (d->out_mode == SF_OUT_IJ || d->out_mode == SF_OUT_OIHW) SF_REQUIRE(d->out_dtype == SF_F32 && d->act == 0 && d->aux_mode == 0 && bias == nullptr, "sf_gemm: IJ/OIHW output is plain f32");
    if (d->aux_mode) SF_REQUIRE(aux != nullptr, "sf_gemm: aux_mode without aux");

    p.A = A; p.B = B; p.out = out; p.bias = bias; p.aux = aux;
    p.I = d->I; p.J = d->J; p.R = d->R; p.lda = d->lda; p.ldb = d->ldb; p.ldo = d->ldo;
    p.niters = p.iters_per_split = p.nsplit = 0;   // set per kernel variant in launch_one
    p.nsplit_req = d->nsplit;
    p.ni_tiles = p.nj_tiles = 0;
    p.j_fast = 0;
    p.b_gather = d->b_gather; p.out_mode = d->out_mode; p.out_dtype = d->out_dtype;
    p.act = d->act; p.aux_mode = d->aux_mode; p.g = d->g;
    p.dy_pack = p.dx_pack = 0;
    if (d->b_gather) {
        for (int t = 0; t < d->g.ntaps; ++t) {
            SF_REQUIRE(d->g.dy[t] >= -8 && d->g.dy[t] <= 7 && d->g.dx[t] >= -8 && d->g.dx[t] <= 7, "sf_gemm: tap offsets must lie in [-8, 7]");
            p.dy_pack |= (unsigned long long)(d->g.dy[t] + 8) << (4 * t);
            p.dx_pack |= (unsigned long long)(d->g.dx[t] + 8) << (4 * t);
        }
    }
    // a requested split keeps its accumulate-into-out semantics even when it degenerates to one slice
    p.split_stride = d->nsplit > 1 ? d->split_stride : 0;
    p.atomic = ((d->nsplit > 1 && p.split_stride == 0) || d->out_mode == SF_OUT_OIHW) ? 1 : 0;
    return SF_OK;
}

extern "C" int sf_gemm(const sf_gemm_desc* d, const void* A, const void* B, const float* bias,
                       const void* aux, void* out, void* stream) {
    KParams p;
    if (int rc = fill_kparams(d, A, B, bias, aux, out, p)) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (d->dtype == SF_BF16) return dispatch<bf16>(d, p, s);
    return dispatch<float>(d, p, s);
}

template <typename T, int TI> static int launch_batch(KBatch& b, hipStream_t s) {
    using M = Mma<T>;
    constexpr int BK = M::BK;
    constexpr size_t lds = sizeof(T) * 2 * (TI * (BK + M::PAD_RM) + TJ * (BK + M::PAD_RM));
    long long total = 0;
    for (int c = 0; c < b.n; ++c) {
        KParams& p = b.k[c];
        p.niters = ceil_div(p.R, BK);
        p.nsplit = 1;
        p.iters_per_split = p.niters;
        p.ni_tiles = ceil_div(p.I, TI);
        p.nj_tiles = ceil_div(p.J, TJ);
        b.first[c] = (int)total;
        total += (long long)p.ni_tiles * p.nj_tiles;
        if (total > 0x3fffffffLL) { sf_set_error("sf_gemm_batch: bad grid"); return SF_EINVAL; }
    }
    for (int c = b.n; c <= MAX_BATCH; ++c) b.first[c] = (int)total;
    hipLaunchKernelGGL((gemm_batch_kernel<T, TI>), dim3((unsigned)total), dim3(NTHREADS), lds, s, b);
    SF_LAUNCH_CHECK("sf_gemm_batch");
    return SF_OK;
}

extern "C" int sf_gemm_batch(int n, const sf_gemm_desc* d, const void* const* A, const void* const* B,
                             const float* const* bias, const void* const* aux, void* const* out, void* stream) {
    SF_REQUIRE(n >= 1 && n <= MAX_BATCH && d && A && B && out, "sf_gemm_batch: 1..%d problems", MAX_BATCH);
    KBatch b;
    b.n = n;
    for (int c = 0; c < n; ++c) {
        SF_REQUIRE(!d[c].a_kmajor && !d[c].b_kmajor && d[c].nsplit == 1 && d[c].dtype == d[0].dtype &&
                   (d[c].I <= 64) == (d[0].I <= 64), "sf_gemm_batch: problems must share dtype, row-major operands, tile class and have nsplit 1");
        if (int rc = fill_kparams(&d[c], A[c], B[c], bias ? bias[c] : nullptr, aux ? aux[c] : nullptr, out[c], b.k[c])) return rc;
    }
    for (int c = n; c < MAX_BATCH; ++c) b.k[c] = b.k[0];
    hipStream_t s = (hipStream_t)stream;
    const bool small = d[0].I <= 64;
    if (d[0].dtype == SF_BF16) return small ? launch_batch<bf16, 64>(b, s) : launch_batch<bf16, 128>(b, s);
    return small ? launch_batch<float, 64>(b, s) : launch_batch<float, 128>(b, s);
}
