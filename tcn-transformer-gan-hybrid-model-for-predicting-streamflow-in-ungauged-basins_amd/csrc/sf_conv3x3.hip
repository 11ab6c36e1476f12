// 3x3 stride-1 convolution (forward and data gradient) for bf16 NHWC tensors.  Two kernels:
// conv3_dma_kernel (persistent workgroups, buffer-load-to-LDS stages; C % 32 == 0 or C == 16)
// and, for other channel counts, the register-staged conv3x3_kernel described first.
//
// conv3x3_kernel: a workgroup owns a TH x TW tile of output pixels (TH*TW <= 256) and a tile of KT output
// channels.  Per chunk of CK input channels it stages, once, the (TH+2) x (TW+2) halo patch of
// the input and the weights of all nine taps in LDS; the nine taps then read their B operand
// from the same patch at shifted pixel positions, so every input element crosses L2 -> LDS
// once per channel chunk instead of nine times, and one barrier pair covers 72 MFMAs per wave.
// Global loads of the next chunk are issued into registers before the MFMAs of the current
// chunk (latency hidden behind compute; a second resident workgroup covers the LDS store
// burst).  D is kept as [kout][pixel] so a lane stores 4 consecutive channels per access.
#include "sf_common.h"
#include <type_traits>
#include "../../include/sfhip.h"
#include <atomic>

namespace {

constexpr int NT3 = 256;
constexpr int MAX_PP = 400;  // patch pixels the LDS budget allows

struct C3Params {
    const bf16* in;
    const bf16* w;
    const float* bias;
    const bf16* aux;
    void* out;
    int N, H, W, C, K, ldo;
    int TH, TW, tiles_x, tiles_y, ktiles;
    int act, aux_mode, out_dtype;
    int dy[9], dx[9];
    int wstaged;
    // optional 1x1 projection of the finished output row to one channel (the resident shape only)
    const float* dot_w = nullptr; const float* dot_b = nullptr; float* dot_out = nullptr;
};

__device__ __attribute__((aligned(16))) uint4 g_zero_chunk;  // never written: stays zero

__device__ WorkQueue g_queues[SF_QUEUE_SLOTS];
StreamSlots g_slots;     // which queue slot the launches of a stream use

#ifdef SF_CONV3_STAMPS
// diagnostic build only (tools/conv_stamps.py): shader-clock stamps of wave 0 of every workgroup
// around the barrier, the stage and the epilogue of its first STAMP_STEPS pipeline steps
constexpr int STAMP_STEPS = 48;
#ifdef SF_CONV3_STAMPS_ALLWAVES     // lane 0 of every wave: [workgroup][step][slot][wave]
__device__ unsigned long long g_stamps[256 * STAMP_STEPS * 8 * 8];
#define SF_STAMP(slot)                                                                              \
    do {                                                                                            \
        if ((tid & 63) == 0 && stamp_step < STAMP_STEPS)                                            \
            g_stamps[((blockIdx.x * STAMP_STEPS + stamp_step) * 8 + (slot)) * 8 + (tid >> 6)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#ifdef SF_CONV3_STAMPS_POS          // + the end of every tap position of a stage: [workgroup][step][wave][position]
__device__ unsigned long long g_pos_stamps[256 * STAMP_STEPS * 8 * 20];
#endif
#else
__device__ unsigned long long g_stamps[256 * STAMP_STEPS * 8];
#define SF_STAMP(slot)                                                                              \
    do {                                                                                            \
        if (tid == 0 && stamp_step < STAMP_STEPS)                                                   \
            g_stamps[(blockIdx.x * STAMP_STEPS + stamp_step) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#endif
#else
#define SF_STAMP(slot) do { } while (0)
#endif

template <int KT, int CK>
__global__ __launch_bounds__(NT3, 2) void conv3x3_kernel(C3Params p) {
    constexpr int LD = CK + 8;                 // padded row (elements): conflict-free b128 reads
    constexpr int CPR = CK / 8;                // 16-byte chunks per row
    constexpr int W_CHUNKS = 9 * KT * CPR;
    constexpr int NWS = (W_CHUNKS + NT3 - 1) / NT3;
    constexpr int NPS = (MAX_PP * CPR + NT3 - 1) / NT3;
    constexpr int MT = KT / 32;
    constexpr int KS = CK / 16;
    extern __shared__ __attribute__((aligned(16))) bf16 smem3[];
    bf16* sW = smem3;                  // [9][KT][LD]
    bf16* sP = smem3 + 9 * KT * LD;    // [PP][LD]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    int b = blockIdx.x;
    const int kt = b % p.ktiles; b /= p.ktiles;
    const int tx = b % p.tiles_x; b /= p.tiles_x;
    const int ty = b % p.tiles_y;
    const int n = b / p.tiles_y;
    const int y0 = ty * p.TH, x0 = tx * p.TW, k0 = kt * KT;
    const int PW = p.TW + 2, PP = (p.TH + 2) * PW;

    // ---- staging slots: patch
    unsigned poff[NPS];    // offset (in 16-byte units) of the slot's chunk at channel 0, or ~0u
    int plds[NPS];
#pragma unroll
    for (int i = 0; i < NPS; ++i) {
        const int q = tid + i * NT3;
        const int pp = q / CPR, cc = q % CPR;
        poff[i] = ~0u;
        plds[i] = -1;
        if (pp < PP) {
            const int py = pp / PW, px = pp - py * PW;
            const int gy = y0 - 1 + py, gx = x0 - 1 + px;
            plds[i] = pp * LD + cc * 8;
            if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)
                poff[i] = (unsigned)(((((long long)n * p.H + gy) * p.W + gx) * p.C) / 8 + cc);
        }
    }
    // ---- staging slots: weights (rows k*9+t in memory -> LDS row t*KT+k)
    int woff[NWS], wlds[NWS];
#pragma unroll
    for (int i = 0; i < NWS; ++i) {
        const int q = tid + i * NT3;
        woff[i] = -1;
        wlds[i] = -1;
        if (q < W_CHUNKS) {
            const int rw = q / CPR, cc = q % CPR;
            const int k = rw / 9, t = rw - k * 9;
            wlds[i] = (t * KT + k) * LD + cc * 8;
            if (k0 + k < p.K) woff[i] = ((k0 + k) * 9 + t) * p.C + cc * 8;
        }
    }

    uint4 rp[NPS], rw_[NWS];
    auto load_chunk = [&](int c0) {
#pragma unroll
        for (int i = 0; i < NPS; ++i) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (poff[i] != ~0u) v = reinterpret_cast<const uint4*>(p.in)[(size_t)poff[i] + (c0 >> 3)];
            rp[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NWS; ++i) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (woff[i] >= 0) v = *reinterpret_cast<const uint4*>(p.w + woff[i] + c0);
            rw_[i] = v;
        }
    };
    auto store_chunk_lds = [&]() {
#pragma unroll
        for (int i = 0; i < NPS; ++i)
            if (plds[i] >= 0) *reinterpret_cast<uint4*>(sP + plds[i]) = rp[i];
#pragma unroll
        for (int i = 0; i < NWS; ++i)
            if (wlds[i] >= 0) *reinterpret_cast<uint4*>(sW + wlds[i]) = rw_[i];
    };

    // ---- this lane's two output pixels (one per 32-pixel MFMA column tile)
    int pbase[2];
    int gpy[2], gpx[2];
    bool pok[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int q = wave * 64 + nt * 32 + r;
        const bool in_tile = q < p.TH * p.TW;
        const int qq = in_tile ? q : 0;
        const int ly = qq / p.TW, lx = qq - ly * p.TW;
        pbase[nt] = ((ly + 1) * PW + lx + 1) * LD + 8 * h;
        gpy[nt] = y0 + ly;
        gpx[nt] = x0 + lx;
        pok[nt] = in_tile && gpy[nt] < p.H && gpx[nt] < p.W;
    }
    int tapoff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) tapoff[t] = (p.dy[t] * PW + p.dx[t]) * LD;

    f32x16 acc[MT][2];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][c][e] = 0.f;

    const int nchunks = p.C / CK;
    load_chunk(0);
    for (int ci = 0; ci < nchunks; ++ci) {
        if (ci > 0) __syncthreads();
        store_chunk_lds();
        __syncthreads();
        if (ci + 1 < nchunks) load_chunk((ci + 1) * CK);
        const bf16* wrow = sW + r * LD + 8 * h;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8 fa[MT], fb[2];
#pragma unroll
                for (int a = 0; a < MT; ++a)
                    fa[a] = *reinterpret_cast<const bf16x8*>(wrow + (t * KT + a * 32) * LD + ks * 16);
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    fb[c] = *reinterpret_cast<const bf16x8*>(sP + pbase[c] + tapoff[t] + ks * 16);
#pragma unroll
                for (int a = 0; a < MT; ++a)
#pragma unroll
                    for (int c = 0; c < 2; ++c)
                        acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a], fb[c], acc[a][c], 0, 0, 0);
            }
        }
    }

    // ---- epilogue
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        if (!pok[c]) continue;
        const long long pix = ((long long)n * p.H + gpy[c]) * p.W + gpx[c];
#pragma unroll
        for (int a = 0; a < MT; ++a) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int k = k0 + a * 32 + 8 * g + 4 * h;
                if (k >= p.K) continue;
                const long long off = pix * p.ldo + k;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[a][c][4 * g + e];
                if (p.bias) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += p.bias[k + e];
                }
                if (p.act == SF_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] < 0.f ? 0.f : v[e];   // keeps NaN, like torch.relu
                } else if (p.act == SF_ACT_LRELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.2f * v[e];
                }
                if (p.aux_mode) {
                    Vec4<bf16> ax;
                    ax.load(p.aux + off);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool pos = ax.get(e) > 0.f;
                        v[e] = pos ? v[e] : (p.aux_mode == 1 ? 0.f : 0.2f * v[e]);
                    }
                }
                if (p.out_dtype == SF_F32) {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + off) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    Vec4<bf16> o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o.set(e, v[e]);
                    o.store(reinterpret_cast<bf16*>(p.out) + off);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Persistent variant for C % 32 == 0.  A workgroup (8 waves) walks a list of work items
// (512-pixel tile x 64 output channels); per item it runs over the input channels in stages of
// 32.  A stage = the nine taps' weights [9][64][32] and the halo patch [PP][32] of that channel
// block, copied from global memory straight into LDS (global_load_lds_dwordx4) into one of two
// buffers: the copy of stage s+1 -- which may already belong to the next item -- runs under the
// 144 MFMAs per SIMD of stage s and costs no registers.  Both images are dense arrays of
// 64-byte rows (32 channels of one pixel / one output channel); the 16-byte slot of a row is
// XORed with (row >> 2) & 3, applied on the DMA's per-lane source address and again on the
// ds_read_b128 address, which spreads the 16 rows of a read group over all 64 banks.
// Items are dealt so that the workgroups of one XCD work on the same few tiles at a time
// (different output-channel blocks): the shared input patches are L2 hits.
//
// Two shapes of the same kernel: <KT = 64 output channels, CK = 32 input channels per stage>
// (64-byte rows, slot ^= (row >> 2) & 3) and, for K >= 128 or C % 32 != 0, <KT = 128, CK = 16>
// (32-byte rows, slot ^= (row >> 3) & 1): the wide shape fetches 57 KB instead of 78 KB per 144
// MFMAs per SIMD and reads each patch for half as many output-channel blocks.
constexpr int D_TP = 512, D_PP = 640, D_NT = 512;
constexpr int D_BIAS_MAX = 1024;                // output channels whose bias fits the LDS copy
template <int KT, int CK> struct DCfg {
    static constexpr int ROWB = CK * 2;                  // bytes per LDS row
    static constexpr int SLOTS = ROWB / 16;              // 16-byte slots per row
    static constexpr int RPP = 1024 / ROWB;              // rows per 1 KiB copy piece
    static constexpr int WIMG = 9 * KT * ROWB;           // 36,864 bytes for both shapes
    static constexpr int PIMG = D_PP * ROWB;             // 40,960 / 20,480 bytes
    static constexpr int BUF = WIMG + PIMG;
    static constexpr int WJOBS = WIMG / 1024, PJOBS = PIMG / 1024;
    static constexpr int NWJ = (WJOBS + 7) / 8, NPJ = (PJOBS + 7) / 8;
    static constexpr int MT = KT / 32, KS = CK / 16, NPOS = 9 * KS;
    static __device__ __forceinline__ int swz(int row) { return SLOTS == 4 ? (row >> 2) & 3 : (row >> 3) & 1; }
};

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct C3DParams {
    const bf16* in;
    const bf16* w;
    const float* bias;
    const bf16* aux;
    void* out;
    int N, H, W, C, K, ldo;
    int TH, TW, tiles_x, tiles_y, ntiles, ktiles, nchunks;
    int act, aux_mode, out_dtype;
    int dy[9], dx[9];
    int wstaged;     // weights given as the stages' LDS images (sf_conv3x3_stage_weights): contiguous 1 KiB copy pieces
    int qslot;       // work-queue slot of this launch, -1: items dealt statically
    // resident shape, K <= 64: dot_out[pixel] = dot_b[0] + sum_k dot_w[k] * out[pixel][k] (the values as stored)
    const float* dot_w; const float* dot_b; float* dot_out;
};

struct ItemPos {
    int q, chunk;          // item index inside the XCD's list, channel stage
    int n, y0, x0, k0;     // decoded item
};

// RES (C == 2 stages, K <= KT): the weights of both stages are copied once and stay in LDS for
// the whole kernel -- [stage 0 weights][stage 1 weights][patch buffer 0][patch buffer 1], the
// same 152 KB -- so an item fetches its two patch stages only: half the bytes per MFMA.
// EPI fixes the epilogue at compile time (the run-time flags cost the generic epilogue a few hundred
// scalar branches per item, a fifth of the kernel): 0 = whatever the parameters say; 1 = plain bf16
// output, no bias / activation / mask, K a multiple of KT and of the 16-byte groups; 2 (RES) = bias +
// ReLU + the fused 1x1 projection, whole-line stores; 3 (RES) = ReLU mask of `aux`, whole-line stores.
// The host picks the variant whose conditions the call meets (launch3_dma).
template <int KT, int CK, bool RES, int EPI>
__device__ __forceinline__ void conv3_dma_body(const C3DParams& p) {
    constexpr bool SP = EPI != 0;
    static_assert(EPI < 2 || RES, "epilogues 2 and 3 belong to the resident shape");
    const bool e_bias = SP ? EPI == 2 : p.bias != nullptr;
    const int e_act = SP ? (EPI == 2 ? SF_ACT_RELU : SF_ACT_NONE) : p.act;
    const int e_aux = SP ? (EPI == 3 ? 1 : 0) : p.aux_mode;
    using Cf = DCfg<KT, CK>;
    constexpr int ROWB = Cf::ROWB, SLOTS = Cf::SLOTS, RPP = Cf::RPP, MT = Cf::MT, KS = Cf::KS, NPOS = Cf::NPOS;
    constexpr int D_WIMG = Cf::WIMG, D_BUF = Cf::BUF, D_WJOBS = Cf::WJOBS, D_PJOBS = Cf::PJOBS;
    // LOADERS (the fixed-epilogue variants): the copies of a stage are issued by waves 4-7 alone, all of them
    // at the start of the stage.  Those waves lose the matrix pipe to their older partners on the SIMD for
    // the first half of a stage anyway (issue arbitration goes by age), so the ~100 cycles each copy
    // instruction holds its wave cost them nothing there, while waves 0-3 run their MFMAs without a
    // single copy in their stream and waves 4-7 find none left when they have the pipe to themselves.
#ifdef SF_CONV3_NO_LOADERS
    constexpr bool LOADERS = false;
#else
    constexpr bool LOADERS = EPI != 0;
#endif
    constexpr int JW = LOADERS ? 4 : 8;          // waves that issue copies
    constexpr int D_NWJ = (D_WJOBS + JW - 1) / JW, D_NPJ = (D_PJOBS + JW - 1) / JW;
    extern __shared__ __attribute__((aligned(16))) char smd[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef SF_CONV3_LOADERS_FIRST_HALF     // comparison build: waves 0-3 copy instead
    const int jw = LOADERS ? (wave < 4 ? wave : -1) : wave;
#else
    const int jw = LOADERS ? wave - 4 : wave;    // index among the copying waves (negative: none)
#endif
    const int r = lane & 31, h = lane >> 5;
    const int PW = p.TW + 2, PP = (p.TH + 2) * PW, TP = p.TH * p.TW;

    // ---- work list of this workgroup: the items of "its" XCD (tiles xcd, xcd + 8, ...  times the
    // output-channel blocks).  Every workgroup starts on item jblk; the following ones are taken
    // from the XCD's queue head as the workgroup gets there, so a workgroup that starts late or runs
    // slowly (a CU shared with a communication kernel) simply takes fewer items.
    const int xcd = blockIdx.x & 7, jblk = blockIdx.x >> 3, nb8 = gridDim.x >> 3;
    const int xtiles = xcd < p.ntiles ? (p.ntiles - xcd + 7) >> 3 : 0;
    const int nitems = xtiles * p.ktiles;
    __shared__ int s_next[2];
    const bool dyn = p.qslot >= 0 && p.nchunks >= 2;
    WorkQueue* const wq = g_queues + (p.qslot >= 0 ? p.qslot : 0);
    int q_par = 0, q_pending = 0;
    bool q_have = false;
    auto decode = [&](ItemPos& it) {
        const int m = it.q / p.ktiles, kt = it.q - m * p.ktiles;
        const int tile = xcd + 8 * m;
        const int tx = tile % p.tiles_x;
        const int rest = tile / p.tiles_x;
        const int ty = rest % p.tiles_y;
        it.n = rest / p.tiles_y;
        it.y0 = ty * p.TH; it.x0 = tx * p.TW; it.k0 = kt * KT;
    };

    // ---- DMA jobs (tile-invariant part).  A job fills RPP consecutive rows: lane l writes slot
    // l % SLOTS of row l / SLOTS and fetches the channel chunk slot ^ swz(row).
    const int jrow = lane / SLOTS, jslot = lane % SLOTS;
    // byte offset of this lane's 16 bytes of weight copy job `job`
    auto w_off = [&](int job) {
        if (p.wstaged) return job * 1024 + lane * 16;           // the image itself, in copy order
        const int row = RPP * job + jrow;                        // t * KT + k
        const int t = row / KT, k = row % KT;
        const int wch = jslot ^ Cf::swz(row);                    // channel chunk of the stage
        return (wch << 3) >= p.C ? -1 : ((k * 9 + t) * p.C + (wch << 3)) * 2;   // C < 32: beyond the tensor -> zeros
    };
    // The streamed shapes keep the offsets of their jobs in registers; the resident shape copies its
    // weights once, by all eight waves, and computes them on the spot; with loader waves the weights
    // are staged (the host sees to it) and the offset is two operations.
    constexpr bool W_TABLE = !RES && !LOADERS;
    int w_rel[W_TABLE ? D_NWJ : 1];
#pragma unroll
    for (int i = 0; i < (W_TABLE ? D_NWJ : 0); ++i) w_rel[i] = w_off(wave + 8 * i);
    int p_rel[D_NPJ], p_pyx[D_NPJ];
#pragma unroll
    for (int i = 0; i < D_NPJ; ++i) {
        const int row = RPP * ((jw < 0 ? 0 : jw) + JW * i) + jrow;         // patch pixel
        const int py = row / PW, px = row - py * PW;
        const int pch = jslot ^ Cf::swz(row);
        p_pyx[i] = (row < PP && (pch << 3) < p.C) ? ((py << 16) | px) : -1;
        p_rel[i] = ((py * p.W + px) * p.C + (pch << 3)) * 2;   // bytes
    }
    const bf16* zsrc = reinterpret_cast<const bf16*>(&g_zero_chunk);

    // The copies are buffer loads with an LDS destination: the buffer is the current image (or
    // the weight rows from k0 on) at channel offset c0, per-lane byte offsets are 32-bit, and an
    // offset past the end of the buffer writes zeros -- which is how padding pixels and output
    // channels beyond K are filled (no branch around a load, no pointer select).
    struct StageSrc {
        __amdgpu_buffer_rsrc_t wrs, irs;
        int ioff, ym1, xm1;
    };
    // `live` false (nothing left to copy) makes both buffers empty: the pieces then write zeros
    // and touch no memory, and the MFMA loop needs no branch around them.
    auto stage_src = [&](const ItemPos& it, bool live) {
        const int c0 = it.chunk * CK;
        const bf16* wbase = p.wstaged ? p.w + ((size_t)(it.k0 / KT) * p.nchunks + it.chunk) * (D_WIMG / 2)
                                      : p.w + (size_t)it.k0 * 9 * p.C + c0;
#ifdef SF_CONV3_DIAG_SAMETILE     // timing only: every patch copy fetches the same pixels (L2 hits)
        const bf16* ibase = p.in + c0;
#else
        const bf16* ibase = p.in + (size_t)it.n * p.H * p.W * p.C + c0;
#endif
        const unsigned wbytes = !live ? 0u : p.wstaged ? (unsigned)D_WIMG : (unsigned)(((p.K - it.k0) * 9 * p.C - c0) * 2);
        const unsigned ibytes = live ? (unsigned)((p.H * p.W * p.C - c0) * 2) : 0u;
        StageSrc ss;
        ss.wrs = __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, wbytes, 0x00020000);
        ss.irs = __builtin_amdgcn_make_buffer_rsrc((void*)ibase, 0, ibytes, 0x00020000);
#ifdef SF_CONV3_DIAG_SAMETILE
        ss.ym1 = -1; ss.xm1 = -1;
#else
        ss.ym1 = it.y0 - 1; ss.xm1 = it.x0 - 1;
#endif
        ss.ioff = ((ss.ym1 * p.W + ss.xm1) * p.C) * 2;
        return ss;
    };
    // piece j of a stage's copy: j < D_NWJ weights, then the patch
    auto issue_wpiece = [&](const StageSrc& ss, char* dst, int j) {
        if (RES) {          // once per kernel: every wave takes part (j < ceil(jobs / 8))
            const int job = wave + 8 * j;
            if (job < D_WJOBS)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ss.wrs, (lptr_t)(dst + job * 1024), 16, w_off(job), 0, 0, 0);
            return;
        }
        if (LOADERS && jw < 0) return;
        const int job = jw + JW * j;
        if (D_WJOBS % JW == 0 || job < D_WJOBS)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ss.wrs, (lptr_t)(dst + job * 1024), 16,
                                                     LOADERS ? job * 1024 + lane * 16 : w_rel[j], 0, 0, 0);
    };
    constexpr int P_OFF = RES ? 0 : D_WIMG;      // patch image inside a stage buffer
    auto issue_piece = [&](const StageSrc& ss, char* buf, int j) {
        if (LOADERS && jw < 0) return;
        if (j < D_NWJ) {
            if (!RES) issue_wpiece(ss, buf, j);
        } else {
            const int i = j - D_NWJ, job = jw + JW * i;
            if (D_PJOBS % JW == 0 || job < D_PJOBS) {
                const int gy = ss.ym1 + (p_pyx[i] >> 16), gx = ss.xm1 + (p_pyx[i] & 0xffff);
                const bool ok = p_pyx[i] >= 0 && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
                const int voff = ok ? ss.ioff + p_rel[i] : (int)0x80000000;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ss.irs, (lptr_t)(buf + P_OFF + job * 1024), 16, voff, 0, 0, 0);
            }
        }
    };

    // ---- compute mapping: wave = 64 pixels x KT output channels (MT x 2 MFMA tiles)
    int prow[2];           // patch row of the lane's pixel (centre tap)
    int lyx[2];
    bool pin[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int q = wave * 64 + c * 32 + r;
        pin[c] = q < TP;
        const int qq = pin[c] ? q : 0;
        const int ly = qq / p.TW, lx = qq - ly * p.TW;
        prow[c] = (ly + 1) * PW + lx + 1;
        lyx[c] = (ly << 16) | lx;
    }
    // whole-line epilogue (resident shape): the pixel whose 16-byte piece this lane stores, per pixel tile and pass
    int st_lyx[2][4];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = wave * 64 + c * 32 + (lane >> 3) + 8 * j;
            const int ly = q / p.TW;
            // (a pixel beyond the tile gets a row no image has, so one comparison rejects it)
            st_lyx[c][j] = (RES || EPI == 1) ? (q < TP ? (ly << 16) | (q - ly * p.TW) : 0x7fff0000) : 0;
        }
    // byte offset of the lane's B fragment in the patch image for tap t, reduction step 0
    // (step 1, 64-byte rows only, is the same offset ^ 32): tile-invariant, so computed once
    int pa[2][9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int row = prow[c] + p.dy[t] * PW + p.dx[t];
            pa[c][t] = row * ROWB + ((Cf::swz(row) ^ h) << 4);
        }
    // weights: row = t*KT + a*32 + r, slot = (h + 2 ks) ^ swz(r); one lane offset per ks
    const int aLane0 = r * ROWB + ((h ^ Cf::swz(r)) << 4);
    const int aLane1 = r * ROWB + (((h + 2) ^ Cf::swz(r)) << 4);   // used by the 64-byte-row shape only

    f32x16 acc[MT][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][c][e] = 0.f;
    };
    zero_acc();

#ifdef SF_CONV3_STAMPS
    int stamp_step = 0;
#endif
    // One stage: 9 taps x KS reduction steps, the MT + 2 fragment reads of position i + 1 issued
    // before the 2 MT MFMAs of position i (pinned order, counted lgkmcnt).
    // One copy piece of the next stage is issued per position (an LDS-DMA instruction holds the
    // issuing wave for ~100 cycles; spread out, the other wave's MFMAs cover it).
    // `first` (a compile-time flag): the stage opens an item, its first MFMAs start from zero instead
    // of the accumulators (which then need no clearing after the epilogue)
    auto run_stage = [&](auto first, const char* __restrict__ cur, char* __restrict__ fill, const StageSrc& ss,
                         const char* __restrict__ wts) {
        constexpr bool FIRST = decltype(first)::value;
        const char* wl0 = wts + aLane0;
        const char* wl1 = wts + aLane1;
        const char* pl = cur + P_OFF;
        // fragment ring: the reads of position i + PF are issued before the MFMAs of position i (PF = 2
        // where a position holds only four MFMAs per wave, the resident-weight shape: one position
        // ahead does not cover the LDS latency there)
        constexpr int PF = RES ? 2 : 1, RING = PF + 1;
        bf16x8 fa[RING][MT], fb[RING][2];
        auto issue = [&](int i) {
            const int t = i / KS, ks = i % KS, s = i % RING;
#pragma unroll
            for (int a = 0; a < MT; ++a)
                fa[s][a] = *reinterpret_cast<const bf16x8*>((ks ? wl1 : wl0) + (t * KT + a * 32) * ROWB);
#pragma unroll
            for (int c = 0; c < 2; ++c)
                fb[s][c] = *reinterpret_cast<const bf16x8*>(pl + (pa[c][t] ^ (ks << 5)));
        };
        // copy pieces of the next stage issued per tap position: all of them early in the stage, so
        // that the last one has most of the stage to land before the barrier waits for it
#ifndef SF_CONV3_PPP
#define SF_CONV3_PPP 3
#endif
        constexpr int PPP = SF_CONV3_PPP;
        static_assert(D_NWJ + D_NPJ <= NPOS * PPP, "copy pieces per position");
#if defined(SF_CONV3_DIAG_NOMFMA) || defined(SF_CONV3_BLOCKED)
        // (the order of the first builds, kept for the timing-only build without MFMAs and for comparisons: all reads and
        // copy pieces of a position, then its MFMAs)
#pragma unroll
        for (int i = 0; i < PF; ++i) issue(i);
#pragma unroll
        for (int i = 0; i < NPOS; ++i) {
            if (i + PF < NPOS) issue(i + PF);
#ifdef SF_CONV3_DIAG_NOCOPY       // timing only: the stage buffers keep what the first copies brought
            if (p.N < 0)
#endif
#pragma unroll
            for (int j = i * PPP; j < (i + 1) * PPP; ++j)
                if (j < D_NWJ + D_NPJ) issue_piece(ss, fill, j);
            __builtin_amdgcn_sched_barrier(0);
#ifndef SF_CONV3_DIAG_NOMFMA    // timing only: copies, fragment reads and epilogue without the arithmetic
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    if (FIRST && i == 0) {
                        f32x16 z;
#pragma unroll
                        for (int e = 0; e < 16; ++e) z[e] = 0.f;
                        acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i % RING][a], fb[i % RING][c], z, 0, 0, 0);
                    } else {
                        acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i % RING][a], fb[i % RING][c], acc[a][c], 0, 0, 0);
                    }
                }
#else
            asm volatile("" ::"v"(fa[i % RING][0]), "v"(fb[i % RING][0]), "v"(fb[i % RING][1]));
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
#else
        // Every fragment read of position i + PF and every copy piece of the position sits BEHIND one
        // of position i's MFMAs (pinned): an in-order wave issues them while that MFMA occupies the
        // matrix pipe.  Read order = order of first use.  (Worth 1 % on the resident shape, nothing on
        // the wide one; neither were a two-set register ring that keeps reads twelve MFMAs ahead, nop
        // padding between the MFMAs, a static priority for waves 4-7 or a start-up skew between the
        // two waves of a SIMD.  What the wide shape loses at its stage barriers -- 20 % by the
        // timing-only build without them -- is the older wave of each SIMD winning the matrix pipe
        // and then idling at the barrier while the younger one runs its stage alone.)
        constexpr int NM = 2 * MT, NR = MT + 2;
        auto read_one = [&](int i, int m) {
            const int t = i / KS, ks = i % KS, sl = i % RING;
            // fa0, fb0, fb1, fa1, fa2, ...
            if (m == 0 || m >= 3) {
                const int a = m == 0 ? 0 : m - 2;
                fa[sl][a] = *reinterpret_cast<const bf16x8*>((ks ? wl1 : wl0) + (t * KT + a * 32) * ROWB);
            } else {
                fb[sl][m - 1] = *reinterpret_cast<const bf16x8*>(pl + (pa[m - 1][t] ^ (ks << 5)));
            }
        };
#pragma unroll
        for (int i = 0; i < PF; ++i) issue(i);
        if (LOADERS) {
#ifdef SF_CONV3_DIAG_NOCOPY
            if (p.N < 0)
#endif
#pragma unroll
            for (int j = 0; j < D_NWJ + D_NPJ; ++j) issue_piece(ss, fill, j);
        }
#pragma unroll
        for (int i = 0; i < NPOS; ++i) {
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const int a = m >> 1, c = m & 1;
                __builtin_amdgcn_sched_barrier(0);
                if (FIRST && i == 0) {
                    f32x16 z;
#pragma unroll
                    for (int e = 0; e < 16; ++e) z[e] = 0.f;
                    acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i % RING][a], fb[i % RING][c], z, 0, 0, 0);
                } else {
                    acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i % RING][a], fb[i % RING][c], acc[a][c], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#ifdef SF_CONV3_STAMPS_POS
                if (m == NM - 1 && (tid & 63) == 0 && stamp_step < STAMP_STEPS && i < 20)
                    g_pos_stamps[((blockIdx.x * STAMP_STEPS + stamp_step) * 8 + (tid >> 6)) * 20 + i] = __builtin_amdgcn_s_memtime();
#endif
                if (i + PF < NPOS && m < NR) read_one(i + PF, m);
                // the copy pieces behind the position's last MFMAs
                if (!LOADERS && m >= NM - PPP) {
                    const int j = i * PPP + (m - (NM - PPP));
#ifdef SF_CONV3_DIAG_NOCOPY       // timing only: the stage buffers keep what the first copies brought
                    if (p.N < 0)
#endif
                    if (j < D_NWJ + D_NPJ) issue_piece(ss, fill, j);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#endif
    };

    // Epilogue of an item.  The ReLU-mask loads are issued together and without branches; the
    // bias comes from LDS (copied there once per workgroup).
    const float* sbias = reinterpret_cast<const float*>(smd + 2 * D_BUF);
    // RES with 64 output channels: a pixel's output row is exactly one 128-byte line.  `lines` (a
    // 4 KB slice per wave of the patch buffer the item's last stage was computed from) then takes the
    // 16-byte pieces of 32 pixels, and the wave stores them back out as whole lines, 8 lanes per
    // pixel, with streaming (non-temporal) stores: the output then neither leaves partial lines for L2
    // to merge nor pushes the input lines of the next stages out of it.
    // The wide plain variant does the same per 64-channel half of its 128 output channels, through a
    // slice region of its own behind the stage buffers (no barrier: the slices are private to a wave).
    constexpr bool LINES = RES || EPI == 1;
    auto flush_lines = [&](const ItemPos& it, char* lines, int c, int half) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int G = lane & 7;
        // buffer stores into the item's image: a pixel outside the tile or the image gets an offset past
        // the buffer's end and is dropped, so every wave issues exactly four stores per call (the
        // stage loop counts them) and the addresses are 32-bit
        const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(reinterpret_cast<bf16*>(p.out) + (size_t)it.n * p.H * p.W * p.ldo), 0, (unsigned)(p.H * p.W * p.ldo * 2), 0x00020000);
        // byte offset = item part (scalar) + ly * row bytes + lx * pixel bytes + the lane's piece; the
        // host guarantees row bytes < 2^24, so the products are single 24-bit multiplies
        const int sbase = ((it.y0 * p.W + it.x0) * p.ldo + it.k0 + half * 64) * 2;
        const int hrem = p.H - it.y0, wrem = p.W - it.x0;
        const unsigned row_b = (unsigned)(p.W * p.ldo * 2), px_b = (unsigned)(p.ldo * 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pr = (lane >> 3) + 8 * j;                   // pixel row of the slice
            const u32x4 v = *reinterpret_cast<const u32x4*>(lines + wave * 4096 + pr * 128 + ((G ^ ((pr >> 1) & 7)) << 4));
            const int ly = st_lyx[c][j] >> 16, lx = st_lyx[c][j] & 0xffff;   // q / TW, q % TW: tile-invariant
            const bool ok = ly < hrem && lx < wrem;
            const int voff = ok ? (int)(__umul24((unsigned)ly, row_b) + __umul24((unsigned)lx, px_b)) + sbase + G * 16
                                : (int)0x80000000;
            __builtin_amdgcn_raw_buffer_store_b128(v, ors, voff, 0, 2);      // aux 2 = nt (streaming)
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                          // the slice is rewritten for the next pieces
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // Mask variant: the mask operand of an item (the 16-byte groups this lane will need) is requested
    // before the item's last stage runs, so that the epilogue finds it in registers instead of
    // waiting out two round trips to memory.
    uint4 axp[2][MT][2];
    auto aux_prefetch = [&](const ItemPos& it) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int gy = it.y0 + (lyx[c] >> 16), gx = it.x0 + (lyx[c] & 0xffff);
            const bool ok = pin[c] && gy < p.H && gx < p.W;
            const long long po = ok ? (((long long)it.n * p.H + gy) * p.W + gx) * p.ldo : 0;
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    const int k = it.k0 + a * 32 + 8 * (2 * gp + h);
                    axp[c][a][gp] = *reinterpret_cast<const uint4*>(ok ? p.aux + po + k : zsrc);
                }
        }
    };
    // (read once: a load inside the epilogue would make every item wait out a memory round trip)
    const float dot_b0 = (RES && (SP ? EPI == 2 : p.dot_out != nullptr)) ? p.dot_b[0] : 0.f;
    auto epilogue = [&](const ItemPos& it, char* lines) {
        long long pixoff[2], pixidx[2];
        bool pv[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int gy = it.y0 + (lyx[c] >> 16), gx = it.x0 + (lyx[c] & 0xffff);
            pv[c] = pin[c] && gy < p.H && gx < p.W;
            pixidx[c] = pv[c] ? ((long long)it.n * p.H + gy) * p.W + gx : 0;
            pixoff[c] = pixidx[c] * p.ldo;
        }
        auto pack2 = [](float x, float y) {
            typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
            bf16x2 t;
            t[0] = (bf16)x; t[1] = (bf16)y;
            return *reinterpret_cast<const unsigned*>(&t);
        };
        // bf16 tensors with K % 8 == 0 use 16-byte accesses: lanes l and l + 32 hold the two
        // 4-channel halves of every 8-channel group of the same pixel; one v_permlane32_swap per
        // dword regroups them so that the low lane owns all 8 channels of the even group and the
        // high lane those of the odd group (stores), and back (mask loads).
        const bool wide = SP || ((p.K & 7) == 0 && (p.ldo & 7) == 0);
        const bool wide_out = SP || (wide && p.out_dtype == SF_BF16);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            // mask operand of this pixel tile, all loads first
            uint2 axr[MT][4];
            if (e_aux && wide) {
                uint4 ax4[MT][2];
#pragma unroll
                for (int a = 0; a < MT; ++a)
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        if (EPI == 3) { ax4[a][gp] = axp[c][a][gp]; continue; }
                        const int k = it.k0 + a * 32 + 8 * (2 * gp + h);
                        const bf16* src = (pv[c] && (SP || k < p.K)) ? p.aux + pixoff[c] + k : zsrc;
                        ax4[a][gp] = *reinterpret_cast<const uint4*>(src);
                    }
#pragma unroll
                for (int a = 0; a < MT; ++a)
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        const uint4 d = ax4[a][gp];
                        const auto s0 = __builtin_amdgcn_permlane32_swap(d.x, d.z, false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(d.y, d.w, false, false);
                        axr[a][2 * gp] = make_uint2(s0[0], s1[0]);
                        axr[a][2 * gp + 1] = make_uint2(s0[1], s1[1]);
                    }
            } else if (e_aux) {
#pragma unroll
                for (int a = 0; a < MT; ++a)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int k = it.k0 + a * 32 + 8 * g + 4 * h;
                        const bf16* src = (pv[c] && (SP || k < p.K)) ? p.aux + pixoff[c] + k : zsrc;
                        axr[a][g] = *reinterpret_cast<const uint2*>(src);
                    }
            }
            // bias / activation / mask for the four channels k .. k+3 this lane holds of group g
            auto finish = [&](int a, int g, float* v) {
                const int k = it.k0 + a * 32 + 8 * g + 4 * h;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[a][c][4 * g + e];
                if (e_bias) {
                    const float4 bv = *reinterpret_cast<const float4*>(sbias + ((SP || k < p.K) ? k : 0));
                    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                }
                if (e_act == SF_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] < 0.f ? 0.f : v[e];   // keeps NaN, like torch.relu
                } else if (e_act == SF_ACT_LRELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.2f * v[e];
                }
                if (e_aux) {
                    const float neg = e_aux == 1 ? 0.f : 0.2f;
                    const unsigned u0 = axr[a][g].x, u1 = axr[a][g].y;
                    // bf16 > 0  <=>  sign clear and not zero
                    const bool p0 = !(u0 & 0x8000u) && (u0 & 0x7fffu), p1 = !(u0 & 0x80000000u) && (u0 & 0x7fff0000u);
                    const bool p2 = !(u1 & 0x8000u) && (u1 & 0x7fffu), p3 = !(u1 & 0x80000000u) && (u1 & 0x7fff0000u);
                    v[0] = p0 ? v[0] : neg * v[0]; v[1] = p1 ? v[1] : neg * v[1];
                    v[2] = p2 ? v[2] : neg * v[2]; v[3] = p3 ? v[3] : neg * v[3];
                }
            };
            const bool dot = SP ? EPI == 2 : (RES && p.dot_out != nullptr);      // (host: bf16 output in 16-byte groups, k0 = 0)
            float dsum = 0.f;
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    float v0[4], v1[4];
                    finish(a, 2 * gp, v0);
                    finish(a, 2 * gp + 1, v1);
                    if (wide_out) {
                        const unsigned A0 = pack2(v0[0], v0[1]), A1 = pack2(v0[2], v0[3]);
                        const unsigned B0 = pack2(v1[0], v1[1]), B1 = pack2(v1[2], v1[3]);
                        if (RES && dot) {
                            // the projection reads what a separate pass would read back: the rounded values
                            const float* dw = sbias + D_BIAS_MAX + a * 32 + 4 * h;
                            const float4 w0 = *reinterpret_cast<const float4*>(dw + 16 * gp);
                            const float4 w1 = *reinterpret_cast<const float4*>(dw + 16 * gp + 8);
                            dsum += w0.x * __uint_as_float(A0 << 16) + w0.y * __uint_as_float(A0 & 0xffff0000u) +
                                    w0.z * __uint_as_float(A1 << 16) + w0.w * __uint_as_float(A1 & 0xffff0000u) +
                                    w1.x * __uint_as_float(B0 << 16) + w1.y * __uint_as_float(B0 & 0xffff0000u) +
                                    w1.z * __uint_as_float(B1 << 16) + w1.w * __uint_as_float(B1 & 0xffff0000u);
                        }
                        const auto s0 = __builtin_amdgcn_permlane32_swap(A0, B0, false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(A1, B1, false, false);
                        const int k = it.k0 + a * 32 + 8 * (2 * gp + h);
                        if (LINES && (SP || lines)) {      // the fixed epilogues always stage whole lines
                            // row r of the wave's slice, 16-byte slot (channel group of the 64-channel half) XOR (r >> 1) & 7
                            const int G = (a & 1) * 4 + 2 * gp + h;
                            *reinterpret_cast<uint4*>(lines + wave * 4096 + r * 128 + ((G ^ ((r >> 1) & 7)) << 4)) =
                                make_uint4(s0[0], s1[0], s0[1], s1[1]);
                            if ((a & 1) && gp == 1) flush_lines(it, lines, c, a >> 1);   // 64 channels = one line per pixel
                        } else if (pv[c] && (SP || k < p.K))
                            *reinterpret_cast<uint4*>(reinterpret_cast<bf16*>(p.out) + pixoff[c] + k) =
                                make_uint4(s0[0], s1[0], s0[1], s1[1]);
                    } else {
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const float* v = q ? v1 : v0;
                            const int k = it.k0 + a * 32 + 8 * (2 * gp + q) + 4 * h;
                            if (pv[c] && k < p.K) {
                                const long long off = pixoff[c] + k;
                                if (p.out_dtype == SF_F32) {
                                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + off) = make_float4(v[0], v[1], v[2], v[3]);
                                } else {
                                    Vec4<bf16> o;
#pragma unroll
                                    for (int e = 0; e < 4; ++e) o.set(e, v[e]);
                                    o.store(reinterpret_cast<bf16*>(p.out) + off);
                                }
                            }
                        }
                    }
                }
            if (RES && dot && wide_out) {
                dsum += __shfl_xor(dsum, 32, 64);        // the other 32 channels of the pixel
                if (SP) {       // one store per wave and pixel tile, whatever the lanes' pixels (counted by the stage loop)
                    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(
                        (void*)(p.dot_out + (size_t)it.n * p.H * p.W), 0, (unsigned)(p.H * p.W * 4), 0x00020000);
                    const int gy = it.y0 + (lyx[c] >> 16), gx = it.x0 + (lyx[c] & 0xffff);
                    const int voff = (h == 0 && pv[c]) ? (gy * p.W + gx) * 4 : (int)0x80000000;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dsum + dot_b0), drs, voff, 0, 0);
                } else if (h == 0 && pv[c]) p.dot_out[pixidx[c]] = dsum + dot_b0;
            }
        }
    };

    // ---- stage loop: cur = stage being computed, nxt = stage being copied
    if (e_bias) {
        float* sb = reinterpret_cast<float*>(smd + 2 * D_BUF);
        for (int i = tid; i < D_BIAS_MAX; i += D_NT) sb[i] = i < p.K ? p.bias[i] : 0.f;
    }
    if (RES && (SP ? EPI == 2 : p.dot_out != nullptr) && tid < 64)
        reinterpret_cast<float*>(smd + 2 * D_BUF)[D_BIAS_MAX + tid] = tid < p.K ? p.dot_w[tid] : 0.f;
    // whole-line streaming stores: bf16 output rows of exactly 64 channels (one 128-byte line each)
    const bool line_stores = EPI >= 2 || (RES && p.K == 64 && p.ldo == 64 && p.out_dtype == SF_BF16 &&
                                          (reinterpret_cast<uintptr_t>(p.out) & 127) == 0);
    ItemPos cur, nxt;
    cur.q = jblk; cur.chunk = 0;
    if (cur.q >= nitems) {
        if (p.qslot >= 0 && tid == 0) queue_leave(wq, 8, gridDim.x);
        return;
    }
    decode(cur);
    char* const buf0 = RES ? smd + 2 * D_WIMG : smd;
    char* const buf1 = buf0 + (RES ? Cf::PIMG : D_BUF);
    if (RES) {   // both weight stages, once
        ItemPos w1 = cur;
        w1.chunk = 1;
        const StageSrc s0 = stage_src(cur, true), s1 = stage_src(w1, true);
#pragma unroll
        for (int j = 0; j < D_NWJ; ++j) { issue_wpiece(s0, smd, j); issue_wpiece(s1, smd + D_WIMG, j); }
    }
    {
        const StageSrc ss = stage_src(cur, true);
#pragma unroll
        for (int j = 0; j < D_NWJ + D_NPJ; ++j) issue_piece(ss, buf0, j);
    }
    nxt = cur;
    if (dyn && tid == 0) s_next[0] = nb8 + queue_take(wq, xcd);     // read after the first barrier at the earliest
    auto advance = [&](ItemPos& it) {
        if (++it.chunk == p.nchunks) {
            it.chunk = 0;
            if (dyn) {
                // the item taken a whole item ago; lane 0 asks for the one after it now and parks
                // the answer in the other slot after the next barrier (it has arrived by then)
                it.q = __builtin_amdgcn_readfirstlane(s_next[q_par]);
                q_par ^= 1;
                if (tid == 0) { q_pending = queue_take_async(wq, xcd); q_have = true; }   // read in park(), behind the next full barrier
            } else {
                it.q += nb8;
            }
            if (it.q < nitems) decode(it);
        }
    };
    advance(nxt);
    // The barrier between two stages.  The copies of a stage are older than the output stores of the
    // epilogue that follows them, and vector-memory operations retire in issue order: behind an
    // epilogue the wave waits until only its NST stores are left, so the stores drain under the next
    // stage instead of holding every wave at the barrier for a write round trip.  (The fixed-epilogue
    // variants issue exactly NST stores per item and wave; the run-time epilogue drains everything.)
    constexpr int NST = 4 * MT + (EPI == 2 ? 2 : 0);
    bool after_epi = false;
    auto stage_sync = [&]() {
#ifdef SF_CONV3_DIAG_NOBARRIER    // timing only: the waves run free of each other
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        after_epi = false;
        return;
#endif
#ifndef SF_CONV3_DIAG_DRAIN
        if (SP && after_epi) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NST) : "memory");
        else
#endif
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        after_epi = false;
    };
    // one step of the pipeline; `cur_buf` holds the landed stage, `fill` takes the next copy
    auto step = [&](const char* __restrict__ cur_buf, char* __restrict__ fill) {
        const StageSrc ss = stage_src(nxt, nxt.q < nitems);
        if (EPI == 3 && cur.chunk == p.nchunks - 1) aux_prefetch(cur);
        if (SP && cur.chunk == 0) run_stage(std::true_type{}, cur_buf, fill, ss, RES ? smd + cur.chunk * D_WIMG : cur_buf);
        else run_stage(std::false_type{}, cur_buf, fill, ss, RES ? smd + cur.chunk * D_WIMG : cur_buf);
        SF_STAMP(2);
        if (cur.chunk == p.nchunks - 1) {
            char* lines = nullptr;
            if (RES && line_stores) {
                // every wave is done reading the patch the pieces are parked in (its LDS reads have
                // returned); the copies of the next stage stay in flight
                if (SP) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                else __syncthreads();
                lines = const_cast<char*>(cur_buf);
            } else if (EPI == 1) {
                lines = smd + 2 * D_BUF + sizeof(float) * D_BIAS_MAX;
            }
            SF_STAMP(4);
#ifdef SF_CONV3_DIAG_NOEPI        // timing only: nothing is stored (the condition is never true, the MFMAs stay)
            if (p.N < 0)
#endif
            epilogue(cur, lines);
            SF_STAMP(5);
            if (!SP) zero_acc();
            after_epi = true;
        }
        cur = nxt;
        advance(nxt);
    };
    auto park = [&]() {
        if (dyn && tid == 0 && q_have) { s_next[q_par] = nb8 + queue_ticket_wait(q_pending); q_have = false; }
    };
    while (true) {
        // the stage's copy has landed (own vmcnt, then the barrier) and every wave is done
        // reading the buffer the next copy goes to
        SF_STAMP(0);
        stage_sync();
        SF_STAMP(1);
        park();
        step(buf0, buf1);
        SF_STAMP(3);
#ifdef SF_CONV3_STAMPS
        ++stamp_step;
#endif
        if (cur.q >= nitems) break;
        SF_STAMP(0);
        stage_sync();
        SF_STAMP(1);
        park();
        step(buf1, buf0);
        SF_STAMP(3);
#ifdef SF_CONV3_STAMPS
        ++stamp_step;
#endif
        if (cur.q >= nitems) break;
    }
    if (p.qslot >= 0 && tid == 0) {
        // the last take has returned before the workgroup counts as gone
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(q_pending) :: "memory");
        queue_leave(wq, 8, gridDim.x);
    }
}

#ifdef SF_CONV3_STAMPS
#ifdef SF_CONV3_STAMPS_POS
extern "C" int sf_conv3x3_read_pos_stamps(unsigned long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pos_stamps), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -1;
}
#endif
extern "C" int sf_conv3x3_read_stamps(unsigned long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -1;
}
#endif

// (plain kernels around the templated body: hipcc does not emit the host stub of a kernel
// template whose body holds these buffer-resource lambdas)
__global__ __launch_bounds__(D_NT, 2) void conv3_dma_kernel(C3DParams p) { conv3_dma_body<64, 32, false, 0>(p); }
__global__ __launch_bounds__(D_NT, 2) void conv3_dma_res_any_kernel(C3DParams p) { conv3_dma_body<64, 32, true, 0>(p); }
__global__ __launch_bounds__(D_NT, 2) void conv3_dma_res_kernel(C3DParams p) { conv3_dma_body<64, 32, true, 2>(p); }
__global__ __launch_bounds__(D_NT, 2) void conv3_dma_res_mask_kernel(C3DParams p) { conv3_dma_body<64, 32, true, 3>(p); }
__global__ __launch_bounds__(D_NT, 2) void conv3_dma_wide_any_kernel(C3DParams p) { conv3_dma_body<128, 16, false, 0>(p); }
__global__ __launch_bounds__(D_NT, 2) void conv3_dma_wide_kernel(C3DParams p) { conv3_dma_body<128, 16, false, 1>(p); }

// tile of the persistent kernel: at most D_TP pixels, patch at most D_PP rows (pick_tile_lds: fewest
// tiles within 3 %, then the fewest LDS cycles of the patch-fragment reads)
TileChoice pick_tile_dma(int H, int W, int rowb) {
    static const int dy[9] = {-1, -1, -1, 0, 0, 0, 1, 1, 1}, dx[9] = {-1, 0, 1, -1, 0, 1, -1, 0, 1};
    return pick_tile_lds(0, H, W, D_TP, D_PP, 2, rowb, 1, dy, dx, 9, "SF_CONV3_TILE");
}

template <int KT, int CK, bool RES = false> int launch3_dma(const C3Params& q, hipStream_t s) {
    using Cf = DCfg<KT, CK>;
    C3DParams p;
    p.in = q.in; p.w = q.w; p.bias = q.bias; p.aux = q.aux; p.out = q.out;
    p.N = q.N; p.H = q.H; p.W = q.W; p.C = q.C; p.K = q.K; p.ldo = q.ldo;
    p.act = q.act; p.aux_mode = q.aux_mode; p.out_dtype = q.out_dtype; p.wstaged = q.wstaged;
    p.dot_w = q.dot_w; p.dot_b = q.dot_b; p.dot_out = q.dot_out;
    for (int t = 0; t < 9; ++t) { p.dy[t] = q.dy[t]; p.dx[t] = q.dx[t]; }
    { const TileChoice tc = pick_tile_dma(p.H, p.W, Cf::ROWB); p.TH = tc.th; p.TW = tc.tw; }
    p.tiles_x = (p.W + p.TW - 1) / p.TW;
    p.tiles_y = (p.H + p.TH - 1) / p.TH;
    const long long ntiles = (long long)p.N * p.tiles_x * p.tiles_y;
    if (ntiles >= 0x7fffffffLL / 8) { sf_set_error("sf_conv3x3: too many tiles"); return SF_EINVAL; }
    if (2LL * p.H * p.W * p.C >= 0x7fffffffLL || 2LL * p.K * 9 * p.C >= 0x7fffffffLL) {
        sf_set_error("sf_conv3x3: one image or the weights exceed 2 GiB");
        return SF_EINVAL;
    }
    p.ntiles = (int)ntiles;
    p.ktiles = (p.K + KT - 1) / KT;
    p.nchunks = (p.C + CK - 1) / CK;
    const size_t lds_base = 2 * (size_t)Cf::BUF + sizeof(float) * D_BIAS_MAX + (RES ? sizeof(float) * 64 : 0);
    // the epilogue variant whose fixed conditions this call meets (see conv3_dma_body)
    const bool bf16_groups = p.out_dtype == SF_BF16 && (p.K & 7) == 0 && (p.ldo & 7) == 0;
    const bool lines = RES && p.K == 64 && p.ldo == 64 && p.out_dtype == SF_BF16 &&
                       (reinterpret_cast<uintptr_t>(p.out) & 127) == 0;
    int epi = 0;
    if (!RES && KT == 128 && bf16_groups && p.K % KT == 0 && !p.bias && p.act == SF_ACT_NONE && !p.aux_mode &&
        p.ldo % 64 == 0 && (reinterpret_cast<uintptr_t>(p.out) & 127) == 0 && p.wstaged)
        epi = 1;
    if (RES && lines && p.bias && p.act == SF_ACT_RELU && !p.aux_mode && p.dot_out) epi = 2;
    if (RES && lines && !p.bias && p.act == SF_ACT_NONE && p.aux_mode == 1 && !p.dot_out) epi = 3;
    // the fixed epilogues address the output with 32-bit byte offsets inside one image, built from 24-bit products
    if (2LL * p.H * p.W * p.ldo >= 0x7fffffffLL || 2LL * p.W * p.ldo >= (1 << 24) || p.H >= 0x7fff) epi = 0;
#ifdef SF_CONV3_GENERIC_EPILOGUE
    epi = 0;     // diagnostic build (A/B timing): every call on the run-time epilogue
#endif
    void (*kern)(C3DParams) = RES ? (epi == 2 ? &conv3_dma_res_kernel : epi == 3 ? &conv3_dma_res_mask_kernel : &conv3_dma_res_any_kernel)
                              : KT == 64 ? &conv3_dma_kernel
                                         : (epi == 1 ? &conv3_dma_wide_kernel : &conv3_dma_wide_any_kernel);
    const size_t lds = lds_base + (epi == 1 ? 8 * 4096 : 0);      // + the eight waves' line slices
    static DeviceOnce attr_set[4];
    DeviceOnce& once = attr_set[epi];
    if (!once.done()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            sf_set_error("sf_conv3x3: cannot raise the dynamic LDS limit to %zu bytes", lds);
            return SF_ELAUNCH;
        }
        once.mark();
    }
    long long items = ntiles * p.ktiles;
    long long grid = (items + 7) / 8 * 8;
    if (grid > 256) grid = 256;
    p.qslot = items > grid ? g_slots.slot_of(s) : -1;    // one item each: nothing to deal
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(D_NT), lds, s, p);
    SF_LAUNCH_CHECK("sf_conv3x3");
    return SF_OK;
}

// tile shape with the fewest tiles (TH*TW <= 256 lanes, patch <= MAX_PP pixels)
void pick_tile(int H, int W, int* th, int* tw) {
    long long best = -1;
    int bth = 8, btw = 32, bpp = 0;
    for (int w = 8; w <= 128; ++w) {
        const int hh = 256 / w;
        if (hh < 1) continue;
        const int pp = (hh + 2) * (w + 2);
        if (pp > MAX_PP) continue;
        const long long tiles = (long long)((W + w - 1) / w) * ((H + hh - 1) / hh);
        if (best < 0 || tiles < best || (tiles == best && pp < bpp)) { best = tiles; bth = hh; btw = w; bpp = pp; }
    }
    *th = bth; *tw = btw;
}

template <int KT, int CK> int launch3(const C3Params& p, hipStream_t s) {
    constexpr int LD = CK + 8;
    const size_t lds = sizeof(bf16) * ((size_t)9 * KT * LD + (size_t)MAX_PP * LD);
    static DeviceOnce attr_set;
    if (!attr_set.done()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_kernel<KT, CK>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            sf_set_error("sf_conv3x3: cannot raise the dynamic LDS limit to %zu bytes", lds);
            return SF_ELAUNCH;
        }
        attr_set.mark();
    }
    const long long nblk = (long long)p.N * p.tiles_y * p.tiles_x * p.ktiles;
    if (nblk <= 0 || nblk > 0x7fffffffLL) { sf_set_error("sf_conv3x3: bad grid"); return SF_EINVAL; }
    hipLaunchKernelGGL((conv3x3_kernel<KT, CK>), dim3((unsigned)nblk), dim3(NT3), lds, s, p);
    SF_LAUNCH_CHECK("sf_conv3x3");
    return SF_OK;
}

// [K][9][C] packed weights -> the LDS images of the wide shape's stages, one 36,864-byte block per
// (128-channel output tile, 16-channel input chunk): rows t * 128 + k of 32 bytes, the 16-byte slot
// already XORed with (row >> 3) & 1, zeros beyond K and C.
__global__ void stage_weights_kernel(const bf16* __restrict__ wp, bf16* __restrict__ out, int K, int C, int nchunks,
                                     long long total) {
    using Cf = DCfg<128, 16>;
    constexpr int PER = Cf::WIMG / 2;     // elements per stage image
    for (long long o = blockIdx.x * 256LL + threadIdx.x; o < total; o += (long long)gridDim.x * 256) {
        const int stage = (int)(o / PER), within = (int)(o % PER);
        const int row = within / 16, e = within % 16;
        const int ch = (((e >> 3) ^ Cf::swz(row)) << 3) + (e & 7);
        const int t = row / 128, k = (stage / nchunks) * 128 + row % 128, c = (stage % nchunks) * 16 + ch;
        out[o] = (k < K && c < C) ? wp[((size_t)k * 9 + t) * C + c] : (bf16)0.f;
    }
}

// the shape sf_conv3x3 runs a problem on: 0 = wide <128,16>, 1 = resident <64,32>, 2 = <64,32>, 3 / 4 = register-staged
int conv3_variant(int C, int K) {
    if (K <= D_BIAS_MAX && C % 16 == 0 && (K > 64 || C % 32 != 0)) return 0;
    if (K <= 64 && C == 64) return 1;
    if (K <= D_BIAS_MAX && C % 32 == 0) return 2;
    if (K <= 64 && C % 32 == 0) return 3;
    return 4;
}

}  // namespace

extern "C" size_t sf_conv3x3_staged_weight_bytes(int C, int K) {
    if (C <= 0 || K <= 0 || C % 16 || conv3_variant(C, K) != 0) return 0;
    return (size_t)((K + 127) / 128) * ((C + 15) / 16) * DCfg<128, 16>::WIMG;
}

extern "C" int sf_conv3x3_stage_weights(const void* wp, void* out, int C, int K, void* stream) {
    SF_REQUIRE(wp && out, "sf_conv3x3_stage_weights: null argument");
    const size_t bytes = sf_conv3x3_staged_weight_bytes(C, K);
    SF_REQUIRE(bytes > 0, "sf_conv3x3_stage_weights: the %d -> %d channel conv does not run on the staged-weight kernel", C, K);
    const long long total = (long long)(bytes / 2);
    long long grid = (total + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(stage_weights_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const bf16*)wp,
                       (bf16*)out, K, C, (C + 15) / 16, total);
    SF_LAUNCH_CHECK("sf_conv3x3_stage_weights");
    return SF_OK;
}

static int conv3x3_call(const void* in, const void* wp, const float* bias, const void* aux, void* out, int N, int H,
                        int W, int C, int K, int ldo, const int* dy, const int* dx, int act, int aux_mode,
                        int out_dtype, int w_staged, const float* dot_w, const float* dot_b, float* dot_out,
                        void* stream);

extern "C" int sf_conv3x3(const void* in, const void* wp, const float* bias, const void* aux, void* out, int N, int H,
                          int W, int C, int K, int ldo, const int* dy, const int* dx, int act, int aux_mode,
                          int out_dtype, int w_staged, void* stream) {
    return conv3x3_call(in, wp, bias, aux, out, N, H, W, C, K, ldo, dy, dx, act, aux_mode, out_dtype, w_staged, nullptr,
                        nullptr, nullptr, stream);
}

extern "C" int sf_conv3x3_tile_shape(int H, int W, int row_bytes, int* th, int* tw, double* read_cycles) {
    SF_REQUIRE(H > 0 && W > 0 && (row_bytes == 32 || row_bytes == 64) && th && tw, "sf_conv3x3_tile_shape: bad arguments");
    const TileChoice tc = pick_tile_dma(H, W, row_bytes);
    *th = tc.th; *tw = tc.tw;
    if (read_cycles) *read_cycles = tc.cycles;
    return SF_OK;
}

extern "C" int sf_conv3x3_dot_ok(int C, int K, int ldo, int out_dtype) {
    return conv3_variant(C, K) == 1 && K % 8 == 0 && ldo % 8 == 0 && out_dtype == SF_BF16;
}

extern "C" int sf_conv3x3_dot(const void* in, const void* wp, const float* bias, const void* aux, void* out, int N,
                              int H, int W, int C, int K, int ldo, const int* dy, const int* dx, int act, int aux_mode,
                              int out_dtype, const float* dot_w, const float* dot_b, float* dot_out, void* stream) {
    SF_REQUIRE(dot_w && dot_b && dot_out, "sf_conv3x3_dot: null projection argument");
    SF_REQUIRE(sf_conv3x3_dot_ok(C, K, ldo, out_dtype), "sf_conv3x3_dot: shape C=%d K=%d ldo=%d does not run on the resident-weight kernel", C, K, ldo);
    return conv3x3_call(in, wp, bias, aux, out, N, H, W, C, K, ldo, dy, dx, act, aux_mode, out_dtype, 0, dot_w, dot_b,
                        dot_out, stream);
}

static int conv3x3_call(const void* in, const void* wp, const float* bias, const void* aux, void* out, int N, int H,
                        int W, int C, int K, int ldo, const int* dy, const int* dx, int act, int aux_mode,
                        int out_dtype, int w_staged, const float* dot_w, const float* dot_b, float* dot_out,
                        void* stream) {
    SF_REQUIRE(in && wp && out && dy && dx, "sf_conv3x3: null argument");
    SF_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && K > 0, "sf_conv3x3: bad shape");
    SF_REQUIRE(C % 16 == 0, "sf_conv3x3: C=%d must be a multiple of 16", C);
    SF_REQUIRE(K % 4 == 0 && ldo >= K && ldo % 4 == 0, "sf_conv3x3: K=%d / ldo=%d must be multiples of 4", K, ldo);
    SF_REQUIRE(!aux_mode || aux, "sf_conv3x3: aux_mode without aux");
    SF_REQUIRE((long long)N * H * W * C / 8 < 0xFFFFFFFFLL, "sf_conv3x3: input larger than 64 GiB");
    SF_REQUIRE(((uintptr_t)in % 16) == 0 && ((uintptr_t)wp % 16) == 0, "sf_conv3x3: operands must be 16-byte aligned");
    C3Params p;
    p.in = (const bf16*)in; p.w = (const bf16*)wp; p.bias = bias; p.aux = (const bf16*)aux; p.out = out;
    p.N = N; p.H = H; p.W = W; p.C = C; p.K = K; p.ldo = ldo;
    p.act = act; p.aux_mode = aux_mode; p.out_dtype = out_dtype; p.wstaged = w_staged ? 1 : 0;
    p.dot_w = dot_w; p.dot_b = dot_b; p.dot_out = dot_out;
    SF_REQUIRE(!w_staged || conv3_variant(C, K) == 0, "sf_conv3x3: staged weights given for a shape that does not take them");
    for (int t = 0; t < 9; ++t) {
        SF_REQUIRE(dy[t] >= -1 && dy[t] <= 1 && dx[t] >= -1 && dx[t] <= 1, "sf_conv3x3: tap offsets must lie in [-1, 1]");
        p.dy[t] = dy[t]; p.dx[t] = dx[t];
    }
    pick_tile(H, W, &p.TH, &p.TW);
    p.tiles_x = (W + p.TW - 1) / p.TW;
    p.tiles_y = (H + p.TH - 1) / p.TH;
    hipStream_t s = (hipStream_t)stream;
    switch (conv3_variant(C, K)) {
        case 0: return launch3_dma<128, 16>(p, s);
        case 1: return launch3_dma<64, 32, true>(p, s);   // both weight stages resident in LDS
        case 2: return launch3_dma<64, 32>(p, s);
        case 3: p.ktiles = (K + 63) / 64; return launch3<64, 32>(p, s);
        default: p.ktiles = (K + 127) / 128; return launch3<128, 16>(p, s);
    }
}
