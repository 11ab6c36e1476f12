// 4x4 stride-2 pad-1 convolution (the PatchGAN blocks, src/model/gan.py:24-34), forward and data
// gradient, bf16 NHWC, on the persistent LDS-DMA scheme of sf_conv3x3.hip.
//
// A stride-2 4x4 filter splits by the parity (py, px) of the input pixel it touches into four
// 2x2 filters, each reading one parity plane of the input with stride 1:
//     out[oy][ox] = sum_{py,px} sum_{a,b in {0,1}} W[2a+py][2b+px] . in[2(oy+a)-1+py][2(ox+b)-1+px]
// and its transpose splits by the parity of the pixel it WRITES into four 2x2 filters over the
// gradient map:
//     din[2gy+py][2gx+px] = sum_{a,b} W[3-py-2a][3-px-2b]^T . dout[gy-1+py+a][gx-1+px+b].
// So both directions are runs of 4-tap stages: a stage copies, global -> LDS by buffer-load DMA, the
// four taps' weights [4][128][16 channels] (16 KB) and a (TH+1) x (TW+1) patch of one 16-channel
// block (<= 20 KB) -- for the forward pass the patch is one parity plane, gathered with a pixel
// stride of 2 by the DMA's per-lane source addresses, so no space-to-depth copy of the tensor
// exists anywhere -- into one of two buffers while the previous stage's 32 MFMAs per wave run.
// Forward: an item (512-pixel tile x 128 output channels) runs 4 classes x C/16 stages; data
// gradient: an item (class, 512-pixel tile of that class's pixels, 128 channels) runs K/16 stages
// and stores with a pixel stride of 2.  256 persistent workgroups of 8 waves; items are dealt so
// that the workgroups of one XCD share tiles (other channel blocks / other classes).
// LDS rows are 32 bytes; the 16-byte slot is XORed with (row >> 3) & 1 on the DMA source side and
// on the ds_read_b128 side (conflict-free 16-lane read groups, as in the 3x3 kernel's wide shape).
#include "sf_common.h"
#include "../../include/sfhip.h"
#include <atomic>

namespace {

constexpr int S_TP = 512, S_PP = 640, S_NT = 512, S_BIAS_MAX = 1024;
constexpr int KT = 128, CK = 16, ROWB = CK * 2, SLOTS = ROWB / 16, RPP = 1024 / ROWB;
constexpr int NTAP = 4, WTOT = 16;                       // taps per stage; taps per packed weight row
constexpr int WIMG = NTAP * KT * ROWB, PIMG = S_PP * ROWB, BUF = WIMG + PIMG;
constexpr int WJOBS = WIMG / 1024, PJOBS = PIMG / 1024;
// The copies of a stage are issued by waves 4-7 alone, all of them at the start of the stage: those waves
// lose the matrix pipe to their older partners on the SIMD for the first half of a stage anyway (issue
// arbitration goes by age), so the ~100 cycles each copy instruction holds its wave cost them nothing
// there, waves 0-3 run their MFMAs without a copy in their stream, and waves 4-7 find none left when they
// have the pipe to themselves.  (SF_CONV_S2_NO_LOADERS: every wave copies, spread over the taps.)
#ifdef SF_CONV_S2_NO_LOADERS
constexpr bool LOADERS = false;
#else
constexpr bool LOADERS = true;
#endif
constexpr int JW = LOADERS ? 4 : 8;                       // waves that issue copies
constexpr int NWJ = (WJOBS + JW - 1) / JW, NPJ = (PJOBS + JW - 1) / JW, NPIECE = NWJ + NPJ;
constexpr int MT = KT / 32, NPOS = NTAP;
constexpr int PPP = (NPIECE + NPOS - 1) / NPOS;          // copy pieces issued per tap position
static_assert(WJOBS % JW == 0, "every copying wave copies the same number of weight pieces");

__device__ __forceinline__ int swz(int row) { return (row >> 3) & 1; }

__device__ WorkQueue g_queues[SF_QUEUE_SLOTS];
StreamSlots g_slots;     // which queue slot the launches of a stream use

typedef __attribute__((address_space(3))) void* lptr_t;

struct S2Params {
    const bf16* in;        // forward: layer input [N,IH,IW,C]; data gradient: output gradient [N,IH,IW,C]
    const bf16* w;         // staged weight images, one per (128-channel output tile, class, 16-channel chunk)
    const float* bias;
    const bf16* aux;
    bf16* out;             // [N,OH,OW,ldo]
    int N, IH, IW, C, OH, OW, K, ldo;
    int GH, GW;            // pixel grid that is tiled (forward: OH x OW; gradient: one parity class of it)
    int TH, TW, tiles_x, tiles_y, ntiles, ktiles, cch;
    int act, aux_mode;
    int qslot;             // work-queue slot of this launch, -1: items dealt statically
};

struct ItemPos {
    int q, chunk;
    int n, y0, x0, k0, cls;
};

// EPI fixes the epilogue's options at compile time (decided by scalar branches per 8-channel group
// they cost the run-time form a third of its time): 0 = as the parameters say; 1 = bias only (the
// PatchGAN forward convs, BatchNorm follows); 2 = nothing (data gradient); 3 = LeakyReLU mask of
// `aux` (data gradient behind the first layer's activation).
// MTU: 32-channel output tiles an item really has (4, or 2 when the layer has at most 64 output
// channels: the MFMAs, fragment reads and epilogue of the zero half are skipped, and so are the copy
// pieces of the weight image's zero rows).
template <bool DGRAD, int EPI, int MTU = MT>
__device__ __forceinline__ void conv_s2_body(const S2Params& p) {
    const bool e_bias = EPI ? EPI == 1 : p.bias != nullptr;
    const int e_act = EPI ? SF_ACT_NONE : p.act;
    const int e_aux = EPI ? (EPI == 3 ? 2 : 0) : p.aux_mode;
    extern __shared__ __attribute__((aligned(16))) char smd[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int PW = p.TW + 1, PP = (p.TH + 1) * PW, TP = p.TH * p.TW;
    constexpr int IS = DGRAD ? 1 : 2;                    // pixel stride of the patch gather
    const int nchunks = DGRAD ? p.cch : 4 * p.cch;       // stages per item

    // ---- work list of this workgroup
    const int xcd = blockIdx.x & 7, jblk = blockIdx.x >> 3, nb8 = gridDim.x >> 3;
    const int xt = xcd < p.ntiles ? (p.ntiles - xcd + 7) >> 3 : 0;
    const int nitems = (DGRAD ? 4 * xt : xt) * p.ktiles;
    // every workgroup starts on item jblk of its XCD's list and takes the following ones from the
    // list's queue head as it gets there (a workgroup that starts late or runs slowly takes fewer)
    __shared__ int s_next[2];
    const bool dyn = p.qslot >= 0 && nchunks >= 2;
    WorkQueue* const wq = g_queues + (p.qslot >= 0 ? p.qslot : 0);
    int q_par = 0, q_pending = 0;
    bool q_have = false;
    auto decode = [&](ItemPos& it) {
        int m = it.q / p.ktiles;
        const int kt = it.q - m * p.ktiles;
        it.cls = 0;
        if (DGRAD) { it.cls = m & 3; m >>= 2; }
        const int tile = xcd + 8 * m;
        const int tx = tile % p.tiles_x;
        const int rest = tile / p.tiles_x;
        const int ty = rest % p.tiles_y;
        it.n = rest / p.tiles_y;
        it.y0 = ty * p.TH; it.x0 = tx * p.TW; it.k0 = kt * KT;
    };

    // ---- DMA jobs (tile-invariant part): a job fills RPP consecutive 32-byte rows; lane l writes slot
    // l % 2 of row l / 2 and fetches the channel chunk slot ^ swz(row)
    const int jrow = lane / SLOTS, jslot = lane % SLOTS;
    // the weights arrive as the stages' LDS images (sf_conv4x4s2_stage_weights): every copy piece
    // reads 1 KiB of contiguous memory
    const int jw = LOADERS ? wave - 4 : wave;            // index among the copying waves (negative: none)
    const int jw0 = jw < 0 ? 0 : jw;
    int p_rel[NPJ], p_pyx[NPJ];
#pragma unroll
    for (int i = 0; i < NPJ; ++i) {
        const int row = RPP * (jw0 + JW * i) + jrow;         // patch pixel
        const int py = row / PW, px = row - py * PW;
        const int pch = jslot ^ swz(row);
        p_pyx[i] = row < PP ? ((py << 16) | px) : -1;
        p_rel[i] = ((py * p.IW + px) * IS * p.C + (pch << 3)) * 2;
    }

    struct StageSrc {
        __amdgpu_buffer_rsrc_t wrs, irs;
        int ioff, oy, ox;
    };
    // `live` false (nothing left to copy) makes both buffers empty: the pieces then write zeros and
    // touch no memory, so the MFMA loop needs no branch around them
    auto stage_src = [&](const ItemPos& it, bool live) {
        const int cls = DGRAD ? it.cls : it.chunk / p.cch;
        const int c0 = (DGRAD ? it.chunk : it.chunk - cls * p.cch) * CK;
        const int py = cls >> 1, px = cls & 1;
        const bf16* wbase = p.w + ((size_t)((it.k0 / KT) * 4 + cls) * p.cch + c0 / CK) * (WIMG / 2);
        const bf16* ibase = p.in + (size_t)it.n * p.IH * p.IW * p.C + c0;
        const unsigned wbytes = live ? (unsigned)WIMG : 0u;
        const unsigned ibytes = live ? (unsigned)((p.IH * p.IW * p.C - c0) * 2) : 0u;
        StageSrc ss;
        ss.wrs = __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, wbytes, 0x00020000);
        ss.irs = __builtin_amdgcn_make_buffer_rsrc((void*)ibase, 0, ibytes, 0x00020000);
        ss.oy = (DGRAD ? it.y0 : 2 * it.y0) - 1 + py;
        ss.ox = (DGRAD ? it.x0 : 2 * it.x0) - 1 + px;
        ss.ioff = ((ss.oy * p.IW + ss.ox) * p.C) * 2;
        return ss;
    };
    auto issue_piece = [&](const StageSrc& ss, char* buf, int j) {
        if (LOADERS && jw < 0) return;
        if (j < NWJ) {
            const int job = jw + JW * j;
            // a piece is 32 rows of the image: rows t * 128 + k, so pieces 4 t + 2, 4 t + 3 hold k >= 64
            if (MTU == MT || (job & 3) < 2)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ss.wrs, (lptr_t)(buf + job * 1024), 16, job * 1024 + lane * 16, 0, 0, 0);
        } else if (j < NPIECE) {
            const int i = j - NWJ, job = jw + JW * i;
            if (PJOBS % JW == 0 || job < PJOBS) {
                const int gy = ss.oy + IS * (p_pyx[i] >> 16), gx = ss.ox + IS * (p_pyx[i] & 0xffff);
                const bool ok = p_pyx[i] >= 0 && gy >= 0 && gy < p.IH && gx >= 0 && gx < p.IW;
                const int voff = ok ? ss.ioff + p_rel[i] : (int)0x80000000;     // out of range -> zeros
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ss.irs, (lptr_t)(buf + WIMG + job * 1024), 16, voff, 0, 0, 0);
            }
        }
    };

    // ---- compute mapping: wave = 64 pixels x 128 output channels (4 x 2 MFMA tiles)
    int lyx[2];
    bool pin[2];
    int pa[2][NTAP];          // byte offset of the lane's B fragment in the patch image, per tap
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int q = wave * 64 + c * 32 + r;
        pin[c] = q < TP;
        const int qq = pin[c] ? q : 0;
        const int ly = qq / p.TW, lx = qq - ly * p.TW;
        lyx[c] = (ly << 16) | lx;
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
            const int row = (ly + (t >> 1)) * PW + lx + (t & 1);
            pa[c][t] = row * ROWB + ((swz(row) ^ h) << 4);
        }
    }
    const int aLane = r * ROWB + ((h ^ swz(r)) << 4);      // weights: row t*KT + a*32 + r (swz depends on r only)

    f32x16 acc[MTU][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int a = 0; a < MTU; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][c][e] = 0.f;
    };
    zero_acc();

    // One stage: 4 taps, the 6 fragment reads of tap i + 1 issued before the 8 MFMAs of tap i (pinned
    // order), the next stage's copy pieces spread over the taps.
    auto run_stage = [&](const char* __restrict__ cur, char* __restrict__ fill, const StageSrc& ss) {
        const char* wl = cur + aLane;
        const char* pl = cur + WIMG;
        bf16x8 fa[2][MTU], fb[2][2];
        auto issue = [&](int i) {
            const int s = i & 1;
#pragma unroll
            for (int a = 0; a < MTU; ++a) fa[s][a] = *reinterpret_cast<const bf16x8*>(wl + (i * KT + a * 32) * ROWB);
#pragma unroll
            for (int c = 0; c < 2; ++c) fb[s][c] = *reinterpret_cast<const bf16x8*>(pl + pa[c][i]);
        };
        issue(0);
        if (LOADERS) {
#pragma unroll
            for (int j = 0; j < NPIECE; ++j) issue_piece(ss, fill, j);
        }
#pragma unroll
        for (int i = 0; i < NPOS; ++i) {
            if (i + 1 < NPOS) issue(i + 1);
            if (!LOADERS) {
#pragma unroll
                for (int j = i * PPP; j < (i + 1) * PPP; ++j) issue_piece(ss, fill, j);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < MTU; ++a)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i & 1][a], fb[i & 1][c], acc[a][c], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // Epilogue of an item: bias, activation, optional mask operand, 16-byte bf16 stores.  Lanes l and
    // l + 32 hold the two 4-channel halves of every 8-channel group of one pixel; a
    // v_permlane32_swap per dword regroups them so that each lane owns 8 consecutive channels.
    const float* sbias = reinterpret_cast<const float*>(smd + 2 * BUF);
    auto epilogue = [&](const ItemPos& it) {
        long long pixoff[2];
        bool pv[2];
        const int py = it.cls >> 1, px = it.cls & 1;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            int gy = it.y0 + (lyx[c] >> 16), gx = it.x0 + (lyx[c] & 0xffff);
            pv[c] = pin[c] && gy < p.GH && gx < p.GW;
            if (DGRAD) { gy = 2 * gy + py; gx = 2 * gx + px; }
            pv[c] = pv[c] && gy < p.OH && gx < p.OW;
            pixoff[c] = pv[c] ? (((long long)it.n * p.OH + gy) * p.OW + gx) * p.ldo : 0;
        }
        auto pack2 = [](float x, float y) {
            typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
            bf16x2 t;
            t[0] = (bf16)x; t[1] = (bf16)y;
            return *reinterpret_cast<const unsigned*>(&t);
        };
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            uint2 axr[MTU][4];
            if (e_aux) {
                uint4 ax4[MTU][2];
#pragma unroll
                for (int a = 0; a < MTU; ++a)
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        const int k = it.k0 + a * 32 + 8 * (2 * gp + h);
                        // clamped, always valid address: masked lanes read the tensor's first chunk
                        const bf16* src = p.aux + ((pv[c] && k < p.K) ? pixoff[c] + k : 0);
                        ax4[a][gp] = *reinterpret_cast<const uint4*>(src);
                    }
#pragma unroll
                for (int a = 0; a < MTU; ++a)
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        const uint4 d = ax4[a][gp];
                        const auto s0 = __builtin_amdgcn_permlane32_swap(d.x, d.z, false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(d.y, d.w, false, false);
                        axr[a][2 * gp] = make_uint2(s0[0], s1[0]);
                        axr[a][2 * gp + 1] = make_uint2(s0[1], s1[1]);
                    }
            }
            auto finish = [&](int a, int g, float* v) {
                const int k = it.k0 + a * 32 + 8 * g + 4 * h;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[a][c][4 * g + e];
                if (e_bias) {
                    const float4 bv = *reinterpret_cast<const float4*>(sbias + (k < p.K ? k : 0));
                    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                }
                if (e_act == SF_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] < 0.f ? 0.f : v[e];
                } else if (e_act == SF_ACT_LRELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.2f * v[e];
                }
                if (e_aux) {
                    const float neg = e_aux == 1 ? 0.f : 0.2f;
                    const unsigned u0 = axr[a][g].x, u1 = axr[a][g].y;
                    const bool p0 = !(u0 & 0x8000u) && (u0 & 0x7fffu), p1 = !(u0 & 0x80000000u) && (u0 & 0x7fff0000u);
                    const bool p2 = !(u1 & 0x8000u) && (u1 & 0x7fffu), p3 = !(u1 & 0x80000000u) && (u1 & 0x7fff0000u);
                    v[0] = p0 ? v[0] : neg * v[0]; v[1] = p1 ? v[1] : neg * v[1];
                    v[2] = p2 ? v[2] : neg * v[2]; v[3] = p3 ? v[3] : neg * v[3];
                }
            };
#pragma unroll
            for (int a = 0; a < MTU; ++a)
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    float v0[4], v1[4];
                    finish(a, 2 * gp, v0);
                    finish(a, 2 * gp + 1, v1);
                    const unsigned A0 = pack2(v0[0], v0[1]), A1 = pack2(v0[2], v0[3]);
                    const unsigned B0 = pack2(v1[0], v1[1]), B1 = pack2(v1[2], v1[3]);
                    const auto s0 = __builtin_amdgcn_permlane32_swap(A0, B0, false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(A1, B1, false, false);
                    const int k = it.k0 + a * 32 + 8 * (2 * gp + h);
                    if (pv[c] && k < p.K)
                        *reinterpret_cast<uint4*>(p.out + pixoff[c] + k) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                }
        }
    };

    // ---- stage loop: cur = stage being computed, nxt = stage being copied
    if (e_bias) {
        float* sb = reinterpret_cast<float*>(smd + 2 * BUF);
        for (int i = tid; i < S_BIAS_MAX; i += S_NT) sb[i] = i < p.K ? p.bias[i] : 0.f;
    }
    ItemPos cur, nxt;
    cur.q = jblk; cur.chunk = 0;
    if (cur.q >= nitems) {
        if (p.qslot >= 0 && tid == 0) queue_leave(wq, 8, gridDim.x);
        return;
    }
    decode(cur);
    char* const buf0 = smd;
    char* const buf1 = smd + BUF;
    {
        const StageSrc ss = stage_src(cur, true);
#pragma unroll
        for (int j = 0; j < NPIECE; ++j) issue_piece(ss, buf0, j);
    }
    nxt = cur;
    if (dyn && tid == 0) s_next[0] = nb8 + queue_take(wq, xcd);     // read after the first barrier at the earliest
    auto advance = [&](ItemPos& it) {
        if (++it.chunk == nchunks) {
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
    auto park = [&]() {
        if (dyn && tid == 0 && q_have) { s_next[q_par] = nb8 + queue_ticket_wait(q_pending); q_have = false; }
    };
    auto step = [&](const char* __restrict__ cur_buf, char* __restrict__ fill) {
        const StageSrc ss = stage_src(nxt, nxt.q < nitems);
        run_stage(cur_buf, fill, ss);
        if (cur.chunk == nchunks - 1) {
            epilogue(cur);
            zero_acc();
        }
        cur = nxt;
        advance(nxt);
    };
    while (true) {
        // the stage's copy has landed (own vmcnt, then the barrier) and every wave is done reading the
        // buffer the next copy goes to
        __syncthreads();
        park();
        step(buf0, buf1);
        if (cur.q >= nitems) break;
        __syncthreads();
        park();
        step(buf1, buf0);
        if (cur.q >= nitems) break;
    }
    if (p.qslot >= 0 && tid == 0) {
        // the last take has returned before the workgroup counts as gone
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(q_pending) :: "memory");
        queue_leave(wq, 8, gridDim.x);
    }
}

__global__ __launch_bounds__(S_NT, 2) void conv_s2_fwd_any_kernel(S2Params p) { conv_s2_body<false, 0>(p); }
__global__ __launch_bounds__(S_NT, 2) void conv_s2_fwd_kernel(S2Params p) { conv_s2_body<false, 1>(p); }
__global__ __launch_bounds__(S_NT, 2) void conv_s2_dgrad_any_kernel(S2Params p) { conv_s2_body<true, 0>(p); }
__global__ __launch_bounds__(S_NT, 2) void conv_s2_dgrad_kernel(S2Params p) { conv_s2_body<true, 2>(p); }
__global__ __launch_bounds__(S_NT, 2) void conv_s2_dgrad_mask_kernel(S2Params p) { conv_s2_body<true, 3>(p); }
__global__ __launch_bounds__(S_NT, 2) void conv_s2_dgrad_mask64_kernel(S2Params p) { conv_s2_body<true, 3, 2>(p); }
__global__ __launch_bounds__(S_NT, 2) void conv_s2_dgrad_any64_kernel(S2Params p) { conv_s2_body<true, 0, 2>(p); }
__global__ __launch_bounds__(S_NT, 2) void conv_s2_fwd_any64_kernel(S2Params p) { conv_s2_body<false, 0, 2>(p); }

// tile of at most S_TP pixels whose (TH+1) x (TW+1) patch fits S_PP rows (pick_tile_lds: fewest tiles
// within 3 %, then the fewest LDS cycles of the patch-fragment reads)
void pick_tile_s2(int H, int W, int* th, int* tw) {
    static const int dy[4] = {0, 0, 1, 1}, dx[4] = {0, 1, 0, 1};
    const TileChoice tc = pick_tile_lds(1, H, W, S_TP, S_PP, 1, ROWB, 0, dy, dx, 4, "SF_CONV_S2_TILE");
    *th = tc.th; *tw = tc.tw;
}

// packed [K][16][C] weights (tap index class * 4 + a * 2 + b) -> the LDS images of the stages, one
// 16,384-byte block per (128-channel output tile, class, 16-channel chunk): rows (a*2+b) * 128 + k
// of 32 bytes, the 16-byte slot already XORed with (row >> 3) & 1, zeros beyond K
__global__ void stage_weights_s2_kernel(const bf16* __restrict__ wp, bf16* __restrict__ out, int K, int C, int cch,
                                        long long total) {
    constexpr int PER = WIMG / 2;
    for (long long o = blockIdx.x * 256LL + threadIdx.x; o < total; o += (long long)gridDim.x * 256) {
        const int stage = (int)(o / PER), within = (int)(o % PER);
        const int row = within / CK, e = within % CK;
        const int ch = (((e >> 3) ^ swz(row)) << 3) + (e & 7);
        const int chunk = stage % cch, cls = (stage / cch) & 3, kt = stage / (4 * cch);
        const int t = row / KT, k = kt * KT + row % KT, c = chunk * CK + ch;
        out[o] = k < K ? wp[((size_t)k * WTOT + cls * 4 + t) * C + c] : (bf16)0.f;
    }
}

}  // namespace

extern "C" size_t sf_conv4x4s2_staged_weight_bytes(int C, int K) {
    if (C <= 0 || K <= 0 || C % CK) return 0;
    return (size_t)((K + KT - 1) / KT) * 4 * (C / CK) * WIMG;
}

extern "C" int sf_conv4x4s2_stage_weights(const void* wp, void* out, int C, int K, void* stream) {
    SF_REQUIRE(wp && out, "sf_conv4x4s2_stage_weights: null argument");
    const size_t bytes = sf_conv4x4s2_staged_weight_bytes(C, K);
    SF_REQUIRE(bytes > 0, "sf_conv4x4s2_stage_weights: %d reduction channels, must be a positive multiple of 16", C);
    const long long total = (long long)(bytes / 2);
    long long grid = (total + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(stage_weights_s2_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const bf16*)wp,
                       (bf16*)out, K, C, C / CK, total);
    SF_LAUNCH_CHECK("sf_conv4x4s2_stage_weights");
    return SF_OK;
}

extern "C" int sf_conv4x4s2(const void* in, const void* wp, const float* bias, const void* aux, void* out, int N,
                            int IH, int IW, int C, int OH, int OW, int K, int ldo, int dgrad, int act, int aux_mode,
                            void* stream) {
    SF_REQUIRE(in && wp && out, "sf_conv4x4s2: null argument");
    SF_REQUIRE(N > 0 && IH > 0 && IW > 0 && C > 0 && OH > 0 && OW > 0 && K > 0, "sf_conv4x4s2: bad shape");
    SF_REQUIRE(C % 16 == 0, "sf_conv4x4s2: %d reduction channels, must be a multiple of 16", C);
    SF_REQUIRE(K % 8 == 0 && ldo >= K && ldo % 8 == 0 && K <= S_BIAS_MAX, "sf_conv4x4s2: K=%d / ldo=%d must be multiples of 8, K <= %d",
               K, ldo, S_BIAS_MAX);
    SF_REQUIRE(!aux_mode || aux, "sf_conv4x4s2: aux_mode without aux");
    SF_REQUIRE(((uintptr_t)in % 16) == 0 && ((uintptr_t)wp % 16) == 0 && ((uintptr_t)out % 16) == 0 &&
               (!aux || ((uintptr_t)aux % 16) == 0), "sf_conv4x4s2: operands must be 16-byte aligned");
    if (dgrad) {
        // in = gradient of the conv output [N,IH,IW,C]; out = gradient of the conv input [N,OH,OW,K]
        SF_REQUIRE(IH == (OH - 2) / 2 + 1 && IW == (OW - 2) / 2 + 1, "sf_conv4x4s2: gradient map %dx%d does not belong to a %dx%d input",
                   IH, IW, OH, OW);
    } else {
        SF_REQUIRE(OH == (IH - 2) / 2 + 1 && OW == (IW - 2) / 2 + 1, "sf_conv4x4s2: output %dx%d does not belong to a %dx%d input", OH,
                   OW, IH, IW);
    }
    SF_REQUIRE(2LL * IH * IW * C < 0x7fffffffLL, "sf_conv4x4s2: one image exceeds 2 GiB");
    S2Params p;
    p.in = (const bf16*)in; p.w = (const bf16*)wp; p.bias = bias; p.aux = (const bf16*)aux; p.out = (bf16*)out;
    p.N = N; p.IH = IH; p.IW = IW; p.C = C; p.OH = OH; p.OW = OW; p.K = K; p.ldo = ldo;
    p.act = act; p.aux_mode = aux_mode;
    p.GH = dgrad ? (OH + 1) / 2 : OH;
    p.GW = dgrad ? (OW + 1) / 2 : OW;
    pick_tile_s2(p.GH, p.GW, &p.TH, &p.TW);
    p.tiles_x = (p.GW + p.TW - 1) / p.TW;
    p.tiles_y = (p.GH + p.TH - 1) / p.TH;
    const long long ntiles = (long long)N * p.tiles_x * p.tiles_y;
    SF_REQUIRE(ntiles < 0x7fffffffLL / 64, "sf_conv4x4s2: too many tiles");
    p.ntiles = (int)ntiles;
    p.ktiles = (K + KT - 1) / KT;
    p.cch = C / CK;
    const size_t lds = 2 * (size_t)BUF + sizeof(float) * S_BIAS_MAX;
    // the epilogue variant whose fixed options this call uses (see conv_s2_body)
    int vi = dgrad ? 2 : 0;
    void (*kern)(S2Params) = dgrad ? &conv_s2_dgrad_any_kernel : &conv_s2_fwd_any_kernel;
    if (!dgrad && bias && act == SF_ACT_NONE && !aux_mode) { kern = &conv_s2_fwd_kernel; vi = 1; }
    if (dgrad && !bias && act == SF_ACT_NONE && !aux_mode) { kern = &conv_s2_dgrad_kernel; vi = 3; }
    if (dgrad && !bias && act == SF_ACT_NONE && aux_mode == 2) { kern = &conv_s2_dgrad_mask_kernel; vi = 4; }
    if (K <= 64) {      // half-width items (conv_s2_body's MTU = 2)
        if (vi == 4) { kern = &conv_s2_dgrad_mask64_kernel; vi = 5; }
        else if (dgrad) { kern = &conv_s2_dgrad_any64_kernel; vi = 6; }
        else { kern = &conv_s2_fwd_any64_kernel; vi = 7; }
    }
    static DeviceOnce attr_set[8];
    DeviceOnce& once = attr_set[vi];
    if (!once.done()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess) {
            sf_set_error("sf_conv4x4s2: cannot raise the dynamic LDS limit to %zu bytes", lds);
            return SF_ELAUNCH;
        }
        once.mark();
    }
    const long long items = ntiles * p.ktiles * (dgrad ? 4 : 1);
    long long grid = (items + 7) / 8 * 8;
    if (grid > 256) grid = 256;
    p.qslot = items > grid ? g_slots.slot_of((hipStream_t)stream) : -1;    // one item each: nothing to deal
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(S_NT), lds, (hipStream_t)stream, p);
    SF_LAUNCH_CHECK("sf_conv4x4s2");
    return SF_OK;
}
