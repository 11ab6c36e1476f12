// Shared device / host helpers for the streamflow HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <map>
#include <mutex>
#include <stdlib.h>
#include <string.h>
#include <utility>
#include <vector>

typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SF_OK 0
#define SF_EINVAL (-22)
#define SF_ELAUNCH (-5)
#define SF_ENOSYS (-38)

#define SF_F32 0
#define SF_BF16 1

#define SF_ACT_NONE 0
#define SF_ACT_RELU 1
#define SF_ACT_LRELU 2

void sf_set_error(const char* fmt, ...);

// Function attributes (dynamic LDS limit) are per device: a call site keeps one of these in a
// function-local static and sets the attribute the first time it runs on each device.
struct DeviceOnce {
    unsigned long long seen[4] = {0, 0, 0, 0};   // devices 0..255
    int dev = 0;
    bool done() {
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        dev &= 255;
        return (seen[dev >> 6] >> (dev & 63)) & 1ull;
    }
    void mark() { seen[dev >> 6] |= 1ull << (dev & 63); }
};

#define SF_REQUIRE(cond, ...)                                   \
    do {                                                        \
        if (!(cond)) {                                          \
            sf_set_error(__VA_ARGS__);                          \
            return SF_EINVAL;                                   \
        }                                                       \
    } while (0)

#define SF_LAUNCH_CHECK(name)                                                          \
    do {                                                                               \
        hipError_t e__ = hipGetLastError();                                            \
        if (e__ != hipSuccess) {                                                       \
            sf_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));       \
            return SF_ELAUNCH;                                                         \
        }                                                                              \
    } while (0)

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
    static constexpr int EPC = 4;  // elements per 16-byte chunk
    static constexpr int CODE = SF_F32;
};
template <> struct ElemTraits<bf16> {
    static constexpr int EPC = 8;
    static constexpr int CODE = SF_BF16;
};

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }

// 4 consecutive elements of T, loaded / stored as one vector access
template <typename T> struct Vec4;
template <> struct Vec4<float> {
    float4 v;
    __device__ __forceinline__ void load(const float* p) { v = *reinterpret_cast<const float4*>(p); }
    __device__ __forceinline__ void store(float* p) const { *reinterpret_cast<float4*>(p) = v; }
    __device__ __forceinline__ float get(int i) const { return (&v.x)[i]; }
    __device__ __forceinline__ void set(int i, float f) { (&v.x)[i] = f; }
};
template <> struct Vec4<bf16> {
    bf16x4 v;
    __device__ __forceinline__ void load(const bf16* p) { v = *reinterpret_cast<const bf16x4*>(p); }
    __device__ __forceinline__ void store(bf16* p) const { *reinterpret_cast<bf16x4*>(p) = v; }
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
    __device__ __forceinline__ void set(int i, float f) { v[i] = (bf16)f; }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// sum over a 256-thread block; result valid in every thread
__device__ __forceinline__ float block_sum_256(float v, float* sm /* >= 4 floats */) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return sm[0] + sm[1] + sm[2] + sm[3];
}
__device__ __forceinline__ double block_sum_256_d(double v, double* sm) {
    v = wave_sum_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return sm[0] + sm[1] + sm[2] + sm[3];
}

// uniform [0,1) from a 64-bit seed and an element index (splitmix64 finaliser): the dropout masks
__device__ __forceinline__ float rand_u(unsigned long long seed, unsigned long long idx) {
    unsigned long long z = seed + (idx + 1ull) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}

// bilinear source index / weights of one axis
struct AxisW { int i0, i1; float l0, l1; };
__device__ __forceinline__ AxisW axis_w(int dst, int in, int out) {
    // ATen area_pixel_compute_source_index, align_corners=False, float arithmetic
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

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// Work queues of the persistent kernels.  A launch owns one slot of a small pool of counter sets
// (a file-scope __device__ array of the kernel's translation unit, zero at load): heads[] are
// bumped by workgroups asking for their next item, `done` counts workgroups that have left; the
// last one to leave zeroes the slot again, so nothing has to be cleared between launches.  Slots
// belong to streams (StreamSlots below): launches on one stream run one after the other, so a slot
// never has two users at a time however many launches are queued, and launches on different
// streams (or from different host threads) never share one.
#define SF_QUEUE_SLOTS 64
#define SF_QUEUE_HEADS 124
struct WorkQueue {
    int heads[SF_QUEUE_HEADS];
    int done;
    int pad[3];
};
// next item of list `h` beyond the `first` items that were dealt out statically (one lane calls this)
__device__ __forceinline__ int queue_take(WorkQueue* q, int h) {
    return __hip_atomic_fetch_add(&q->heads[h], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The same ticket without waiting for it: the compiler turns a fetch-add on a uniform address into
// its wave-aggregated form and reads the result back at once, which parks the calling wave (and,
// at the next barrier, the workgroup) for the atomic's round trip once per item.  Here the atomic
// is issued by inline assembly, which the compiler's wait-count pass does not track: the register
// it returns is NOT valid until the caller has passed it through queue_ticket_wait().  One lane
// calls this.
__device__ __forceinline__ int queue_take_async(WorkQueue* q, int h) {
    int ticket;
    const int one = 1;
    int* addr = &q->heads[h];
    asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(ticket) : "v"(addr), "v"(one) : "memory");
    return ticket;
}
// Makes the ticket of queue_take_async readable: waits for the wave's outstanding vector-memory
// operations and ties the wait to the ticket's register, so that the compiler does not read the
// value before the wait.  Free right behind a barrier that the wave entered with its copies
// landed.  What the constraint cannot forbid is a register COPY or SPILL of the ticket between
// the two calls: the atomic's result then lands in a register the compiler has given to
// something else (a build that spilled in such a kernel faulted with an illegal address).  The
// kernels that draw tickets therefore must build without spills and keep the ticket in a
// variable that lives across nothing but the stage loop; tests/test_build_no_spills.py compiles
// them with the resource remarks on and fails on any spill.
__device__ __forceinline__ int queue_ticket_wait(int ticket) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(ticket));
    return ticket;
}
// called once per workgroup by one lane, after the value of its last queue_take has arrived
__device__ __forceinline__ void queue_leave(WorkQueue* q, int nheads, int nworkgroups) {
    const int old = __hip_atomic_fetch_add(&q->done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == nworkgroups - 1) {       // everyone else has left (and taken its last item) already
        for (int i = 0; i < nheads; ++i) __hip_atomic_store(&q->heads[i], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&q->done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Host side of the slot pools: the slot a launch on `stream` may use, or -1 when it has to run
// without a queue (items dealt statically, which every kernel supports): the stream is being
// captured into a graph (a replay could run beside another user of the slot), or more than
// SF_QUEUE_SLOTS different streams have launched this kernel family in this process.
struct StreamSlots {
    std::mutex mu;
    hipStream_t streams[SF_QUEUE_SLOTS];
    int n = 0;
    int slot_of(hipStream_t s) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return -1; }
        if (st != hipStreamCaptureStatusNone) return -1;
        std::lock_guard<std::mutex> lock(mu);
        for (int i = 0; i < n; ++i)
            if (streams[i] == s) return i;
        if (n == SF_QUEUE_SLOTS) return -1;
        streams[n] = s;
        return n++;
    }
};

// ---- tile shapes of the persistent LDS-DMA conv kernels -----------------------------------------
// LDS cycles of the patch-fragment reads (ds_read_b128) of a TH x TW pixel tile, averaged over the
// waves, pixel groups and taps of an item.  The patch is an array of `rowb`-byte rows, one per patch
// pixel, (TW + halo) pixels wide; lane r of a group of 32 consecutive tile pixels reads the row of
// its pixel shifted by the tap, 16-byte slot XORed with (row >> 3) & 1 (32-byte rows) or
// (row >> 2) & 3 (64-byte rows).  The hardware serves a wave's read in four fixed groups of 16
// lanes, one cycle per group when the lanes touch 16 different slots of the 256-byte bank window
// and one more for every further distinct address on a slot.  32 consecutive pixels of one tile row
// are conflict-free by construction of the swizzle; every wrap to the next tile row inside a group
// shifts the later lanes by `halo` rows and costs cycles, so the tile width decides how often that
// happens (4.0 = never).
inline double lds_patch_read_cycles(int TW, int TH, int halo, int rowb, int row0, const int* tap_dy, const int* tap_dx,
                                    int ntaps) {
    static const int groups[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                      {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
    const int PW = TW + halo, TP = TH * TW, ks_n = rowb / 32;
    long long total = 0, reads = 0;
    for (int wave = 0; wave < 8; ++wave)
        for (int c = 0; c < 2; ++c)
            for (int t = 0; t < ntaps; ++t)
                for (int ks = 0; ks < ks_n; ++ks) {
                    int addr[32];
                    for (int r = 0; r < 32; ++r) {
                        int q = wave * 64 + c * 32 + r;
                        if (q >= TP) q = 0;
                        const int ly = q / TW, lx = q - ly * TW;
                        const int row = row0 + (ly + tap_dy[t]) * PW + lx + tap_dx[t];
                        const int sw = rowb == 64 ? (row >> 2) & 3 : (row >> 3) & 1;
                        addr[r] = (row * rowb + (sw << 4)) ^ (ks << 5);        // h = 0; h = 1 is the same pattern one slot on
                    }
                    for (int g = 0; g < 2; ++g) {
                        int slot_addr[16][16], slot_n[16] = {0};
                        int worst = 1;
                        for (int i = 0; i < 16; ++i) {
                            const int ai = addr[groups[g][i]], sl = (ai >> 4) & 15;
                            bool seen = false;
                            for (int k = 0; k < slot_n[sl]; ++k) seen |= slot_addr[sl][k] == ai;
                            if (!seen) slot_addr[sl][slot_n[sl]++] = ai;
                            if (slot_n[sl] > worst) worst = slot_n[sl];
                        }
                        total += 2 * worst;      // both halves of the wave (h = 0 and h = 1)
                    }
                    ++reads;
                }
    return (double)total / (double)reads;
}

// Tile of at most tp_max pixels whose (TH + halo) x (TW + halo) patch fits pp_max rows.  Among the
// shapes within 3 % of the fewest tiles the one whose patch reads cost the fewest LDS cycles is
// taken (then fewer tiles, then the smaller patch).  Environment variable `env` (experiments):
// "TWxTH" forces a shape, "fewest" restores the plain fewest-tiles / least-halo rule.  The choice is
// remembered per (memo key, image size, row width).
struct TileChoice { int th, tw; double cycles; };
inline TileChoice pick_tile_lds(int kind, int H, int W, int tp_max, int pp_max, int halo, int rowb, int row0_scale,
                                const int* tap_dy, const int* tap_dx, int ntaps, const char* env) {
    static std::mutex mu;
    static std::map<long long, TileChoice> memo;
    const long long key = ((long long)kind << 60) | ((long long)H << 36) | ((long long)W << 8) | rowb;
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = memo.find(key);
        if (it != memo.end()) return it->second;
    }
    // row0 = row0_scale * (PW + 1): the patch row of tile pixel (0, 0) (1 halo row and column in front for the 3x3 taps)
    auto cycles_of = [&](int tw, int th) {
        return lds_patch_read_cycles(tw, th, halo, rowb, row0_scale * (tw + halo + 1), tap_dy, tap_dx, ntaps);
    };
    int fth = 0, ftw = 0;
    bool legacy = false;
    if (const char* force = getenv(env)) {
        legacy = strcmp(force, "fewest") == 0;
        if (sscanf(force, "%dx%d", &ftw, &fth) != 2 || ftw < 4 || fth < 1 || fth * ftw > tp_max ||
            (fth + halo) * (ftw + halo) > pp_max)
            fth = ftw = 0;
    }
    struct Cand { int th, tw, pp; long long tiles; };
    std::vector<Cand> cands;
    long long fewest = -1;
    for (int w = 4; w <= tp_max; ++w) {
        int hh = tp_max / w;
        if (hh > H) hh = H;
        if (hh < 1) continue;
        const int pp = (hh + halo) * (w + halo);
        if (pp > pp_max) continue;
        const long long tiles = (long long)((W + w - 1) / w) * ((H + hh - 1) / hh);
        cands.push_back({hh, w, pp, tiles});
        if (fewest < 0 || tiles < fewest) fewest = tiles;
    }
    TileChoice best{1, 8, 0.0};
    int bpp = 0;
    long long btiles = -1;
    for (const Cand& c : cands) {
        if (c.tiles * 100 > fewest * 103) continue;
        const double cyc = legacy ? 0.0 : cycles_of(c.tw, c.th);
        const bool better = btiles < 0 || cyc < best.cycles - 1e-9 ||
                            (cyc < best.cycles + 1e-9 && (c.tiles < btiles || (c.tiles == btiles && c.pp < bpp)));
        if (better) { btiles = c.tiles; best = {c.th, c.tw, cyc}; bpp = c.pp; }
    }
    if (fth) best = {fth, ftw, 0.0};
    best.cycles = cycles_of(best.tw, best.th);
    std::lock_guard<std::mutex> lock(mu);
    memo[key] = best;
    return best;
}
