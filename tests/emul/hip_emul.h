// hip_emul.h -- TEST-ONLY host emulator for the HIP kernel sources under d3human-code_amd/csrc.
//
// The dev container has no GPU.  This header lets the *unmodified* .hip kernel sources be compiled
// with the host clang++ and executed on the CPU, one fiber (own stack, hand-rolled switch) per GPU thread, 64-lane
// wavefronts in lockstep at every wave-level operation (ballot / shuffle / MFMA), blocks serial.
// It exists to debug kernel logic (index maps, MFMA fragment layouts, compaction order) before a
// GPU round trip.  It is NOT a product path: the package never loads the emulated library, it is
// only built and loaded by tests/test_emul_*.py from tests/emul/, and it is far too slow for
// anything but toy sizes.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <functional>
#include <algorithm>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define __constant__ static

struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
struct uint3_e { unsigned x, y, z; };
#define D3H_EMULATED 1
typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipPeekAtLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emul"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
static inline hipError_t hipMallocAsync(void** p, size_t n, hipStream_t) { *p = malloc(n); return *p ? 0 : 1; }
static inline hipError_t hipFreeAsync(void* p, hipStream_t) { free(p); return 0; }          // (the emulated kernels have completed when their launch returns)
#define hipHostMallocCoherent 0x40000000
#define hipHostMallocMapped 0x2
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = calloc(1, n); return *p ? 0 : 2; }
#define hipMemcpyDeviceToDevice 3
#define hipMemcpyDeviceToHost 2
#define hipMemcpyHostToDevice 1

// ---- vector types --------------------------------------------------------------------------
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct alignas(16) float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int3 { int x, y, z; };
struct alignas(16) int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline float3 make_float3(float x, float y, float z) { return {x, y, z}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return {x, y}; }
static inline int3 make_int3(int x, int y, int z) { return {x, y, z}; }
static inline int4 make_int4(int x, int y, int z, int w) { return {x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return {x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return {x, y, z, w}; }

namespace emul {

enum State { RUN = 0, WAIT_WAVE = 1, WAIT_BLOCK = 2, DONE = 3 };

// Fiber switch: callee-saved registers + stack pointer only (x86-64 SysV).  swapcontext() saves the signal mask with a system call on
// every switch, and an emulated MFMA is four switches per lane -- that syscall was most of the emulator's run time.
__attribute__((naked, noinline)) static void fiber_switch(void** /*save_sp: rdi*/, void* /*load_sp: rsi*/) {
    __asm__ volatile(
        "pushq %rbp\n\t"
        "pushq %rbx\n\t"
        "pushq %r12\n\t"
        "pushq %r13\n\t"
        "pushq %r14\n\t"
        "pushq %r15\n\t"
        "movq %rsp, (%rdi)\n\t"
        "movq %rsi, %rsp\n\t"
        "popq %r15\n\t"
        "popq %r14\n\t"
        "popq %r13\n\t"
        "popq %r12\n\t"
        "popq %rbx\n\t"
        "popq %rbp\n\t"
        "ret\n\t");
}

struct Fiber {
    void* sp;
    State st;
    uint3_e tid;
    int lane, wave;
    char* stack;
};

struct WaveScratch {
    uint64_t slot[64][4];   // up to 32 bytes per lane per exchange
    int pred[64];
};

struct Machine {
    void* sched_sp = nullptr;
    std::vector<Fiber> fib;
    std::vector<WaveScratch> ws;
    Fiber* cur = nullptr;
    uint3_e bid, bdim, gdim;
    std::function<void()> body;
    size_t stack_bytes = 256 * 1024;
};

inline Machine& M() { static Machine m; return m; }
inline Fiber& cur() { return *M().cur; }

inline void yield_to_sched(State s) {
    Machine& m = M();
    Fiber* f = m.cur;
    f->st = s;
    fiber_switch(&f->sp, m.sched_sp);
}

inline void wave_sync() { yield_to_sched(WAIT_WAVE); }
inline void block_sync() { yield_to_sched(WAIT_BLOCK); }

static void fiber_entry() {
    Machine& m = M();
    m.body();
    m.cur->st = DONE;
    fiber_switch(&m.cur->sp, m.sched_sp);
    abort();        // a finished fiber is never resumed
}

// initial frame of a fiber: six callee-saved slots, the entry address `ret` jumps to, and one pad slot so that the entry sees the
// stack alignment a `call` would have left (rsp = 16k + 8)
inline void fiber_prepare(Fiber& f, size_t stack_bytes) {
    uintptr_t top = ((uintptr_t)f.stack + stack_bytes) & ~(uintptr_t)15;
    void** sp = (void**)top;
    *--sp = nullptr;                       // pad
    *--sp = (void*)fiber_entry;            // return address of the first switch
    for (int i = 0; i < 6; i++) *--sp = nullptr;
    f.sp = (void*)sp;
}

inline void run_block(unsigned nthreads) {
    Machine& m = M();
    unsigned nw = (nthreads + 63) / 64;
    if (m.fib.size() < nthreads) {
        size_t old = m.fib.size();
        m.fib.resize(nthreads);
        for (size_t i = old; i < nthreads; i++) m.fib[i].stack = (char*)malloc(m.stack_bytes);
    }
    if (m.ws.size() < nw) m.ws.resize(nw);
    for (unsigned t = 0; t < nthreads; t++) {
        Fiber& f = m.fib[t];
        fiber_prepare(f, m.stack_bytes);
        f.st = RUN;
        f.lane = t & 63;
        f.wave = t >> 6;
        f.tid.x = t % m.bdim.x;
        f.tid.y = (t / m.bdim.x) % m.bdim.y;
        f.tid.z = t / (m.bdim.x * m.bdim.y);
    }
    for (;;) {
        bool progress = false;
        unsigned done = 0, at_block = 0;
        for (unsigned w = 0; w < nw; w++) {
            unsigned lo = w * 64, hi = std::min(nthreads, lo + 64);
            for (;;) {
                bool ran = false;
                for (unsigned t = lo; t < hi; t++) {
                    Fiber& f = m.fib[t];
                    if (f.st == RUN) {
                        m.cur = &f;
                        fiber_switch(&m.sched_sp, f.sp);
                        ran = true;
                        progress = true;
                    }
                }
                // release the wave if every live lane waits at a wave-level op
                unsigned nwait = 0, nlive = 0, nblk = 0;
                for (unsigned t = lo; t < hi; t++) {
                    State s = m.fib[t].st;
                    if (s != DONE) nlive++;
                    if (s == WAIT_WAVE) nwait++;
                    if (s == WAIT_BLOCK) nblk++;
                }
                if (nlive && nwait == nlive) {
                    for (unsigned t = lo; t < hi; t++) if (m.fib[t].st == WAIT_WAVE) m.fib[t].st = RUN;
                    progress = true;
                    continue;
                }
                if (nwait && nblk) {
                    fprintf(stderr, "[hip_emul] divergent wave: %u lanes at a wave op, %u at __syncthreads (block %u)\n", nwait, nblk, m.bid.x);
                    abort();
                }
                if (!ran) break;
            }
        }
        for (unsigned t = 0; t < nthreads; t++) {
            if (m.fib[t].st == DONE) done++;
            if (m.fib[t].st == WAIT_BLOCK) at_block++;
        }
        if (done == nthreads) break;
        if (done + at_block == nthreads) {
            for (unsigned t = 0; t < nthreads; t++) if (m.fib[t].st == WAIT_BLOCK) m.fib[t].st = RUN;
            continue;
        }
        if (!progress) { fprintf(stderr, "[hip_emul] deadlock in block %u\n", m.bid.x); abort(); }
    }
}

template <class F>
inline void launch(dim3 grid, dim3 block, F&& body) {
    Machine& m = M();
    m.body = body;
    m.bdim = {block.x, block.y, block.z};
    m.gdim = {grid.x, grid.y, grid.z};
    unsigned nthreads = block.x * block.y * block.z;
    for (unsigned z = 0; z < grid.z; z++)
        for (unsigned y = 0; y < grid.y; y++)
            for (unsigned x = 0; x < grid.x; x++) {
                m.bid = {x, y, z};
                run_block(nthreads);
            }
}

// ---- wave-level exchange helpers ---------------------------------------------------------------
template <class T>
inline T exchange(T v, int src) {
    static_assert(sizeof(T) <= 32, "exchange payload too large");
    Machine& m = M();
    Fiber& f = cur();
    WaveScratch& w = m.ws[f.wave];
    memcpy(w.slot[f.lane], &v, sizeof(T));
    wave_sync();
    T r;
    memcpy(&r, w.slot[src & 63], sizeof(T));
    wave_sync();
    return r;
}

inline unsigned long long ballot(int p) {
    Machine& m = M();
    Fiber& f = cur();
    WaveScratch& w = m.ws[f.wave];
    w.pred[f.lane] = p ? 1 : 0;
    wave_sync();
    unsigned long long r = 0;
    unsigned base = f.wave * 64;
    unsigned nthreads = m.bdim.x * m.bdim.y * m.bdim.z;
    for (int l = 0; l < 64; l++) {
        unsigned t = base + l;
        if (t < nthreads && m.fib[t].st != DONE && w.pred[l]) r |= 1ull << l;
    }
    wave_sync();
    return r;
}

typedef float f32x16_e __attribute__((ext_vector_type(16)));
typedef float f32x4_e __attribute__((ext_vector_type(4)));

// v_mfma_f32_32x32x2_f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], D: col=l&31, row=(r&3)+8*(r>>2)+4*(l>>5)
inline f32x16_e mfma_32x32x2f32(float a, float b, f32x16_e c, int, int, int) {
    Machine& m = M();
    Fiber& f = cur();
    WaveScratch& w = m.ws[f.wave];
    float ab[2] = {a, b};
    memcpy(w.slot[f.lane], ab, 8);
    wave_sync();
    int j = f.lane & 31, h = f.lane >> 5;
    for (int r = 0; r < 16; r++) {
        int i = (r & 3) + 8 * (r >> 2) + 4 * h;
        float acc = c[r];
        for (int k = 0; k < 2; k++) {
            float av[2], bv[2];
            memcpy(av, w.slot[i + 32 * k], 8);
            memcpy(bv, w.slot[j + 32 * k], 8);
            acc = fmaf(av[0], bv[1], acc);
        }
        c[r] = acc;
    }
    wave_sync();
    return c;
}

// v_mfma_f32_16x16x4_f32: A[l&15][k=l>>4], B[k=l>>4][l&15], D: col=l&15, row=(l>>4)*4+r
inline f32x4_e mfma_16x16x4f32(float a, float b, f32x4_e c, int, int, int) {
    Machine& m = M();
    Fiber& f = cur();
    WaveScratch& w = m.ws[f.wave];
    float ab[2] = {a, b};
    memcpy(w.slot[f.lane], ab, 8);
    wave_sync();
    int j = f.lane & 15, q = f.lane >> 4;
    for (int r = 0; r < 4; r++) {
        int i = q * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 4; k++) {
            float av[2], bv[2];
            memcpy(av, w.slot[i + 16 * k], 8);
            memcpy(bv, w.slot[j + 16 * k], 8);
            acc = fmaf(av[0], bv[1], acc);
        }
        c[r] = acc;
    }
    wave_sync();
    return c;
}

// v_mfma_f32_16x16x32_bf16: lane i + 16 q holds k-steps 8 q .. 8 q + 7 of row i (A) / column i (B), two bf16 per dword (low half first);
// D: col = l & 15, row = (l >> 4) * 4 + r.  Products of two bf16 are exact in fp32; the sums are taken in k order (the hardware's
// internal order is not documented: comparisons with the GPU carry a tolerance).
typedef unsigned int u32x4_e __attribute__((ext_vector_type(4)));
inline f32x4_e mfma_16x16x32bf16(u32x4_e a, u32x4_e b, f32x4_e c) {
    Machine& m = M();
    Fiber& f = cur();
    WaveScratch& w = m.ws[f.wave];
    unsigned ab[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    memcpy(w.slot[f.lane], ab, 32);
    wave_sync();
    int j = f.lane & 15, q = f.lane >> 4;
    auto bf = [](unsigned dw, int half) { unsigned u = half ? (dw & 0xffff0000u) : (dw << 16); float v; memcpy(&v, &u, 4); return v; };
    for (int r = 0; r < 4; r++) {
        int i = q * 4 + r;
        float acc = c[r];
        for (int kq = 0; kq < 4; kq++) {
            unsigned av[8], bv[8];
            memcpy(av, w.slot[i + 16 * kq], 32);
            memcpy(bv, w.slot[j + 16 * kq], 32);
            for (int s = 0; s < 8; s++) acc += bf(av[s >> 1], s & 1) * bf(bv[4 + (s >> 1)], s & 1);
        }
        c[r] = acc;
    }
    wave_sync();
    return c;
}

// IEEE binary16 <-> binary32 on the host (round to nearest even, subnormals kept): what v_cvt_pk_f16_f32 / v_cvt_f32_f16 do
inline unsigned f32_to_f16_bits(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    const unsigned sign = (u >> 16) & 0x8000u;
    const unsigned au = u & 0x7fffffffu;
    if (au > 0x7f800000u) return sign | 0x7e00u;                       // NaN
    if (au >= 0x477ff000u) return sign | 0x7c00u;                      // >= 65520 rounds to inf (inf included)
    if (au < 0x33000001u) return sign;                                 // <= 2^-25: rounds to zero (2^-25 itself ties to even = 0)
    int e = (int)(au >> 23) - 127;
    unsigned m = (au & 0x7fffffu) | 0x800000u;                         // 24-bit significand
    int shift = (e < -14) ? (13 + (-14 - e)) : 13;                     // bits dropped
    unsigned keep = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (keep & 1u))) keep++;
    if (e < -14) return sign | keep;                                   // subnormal (a carry into 0x400 is the smallest normal: still right)
    unsigned he = (unsigned)(e + 15);
    return sign | (((he << 10) + (keep - 0x400u)) & 0x7fffu) | 0;      // keep in [0x400, 0x800]: a carry bumps the exponent
}
inline float f16_bits_to_f32(unsigned h) {
    const unsigned sign = (h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3ffu;
    unsigned u;
    if (e == 31) u = sign | 0x7f800000u | (m << 13);
    else if (e == 0) {
        float v = (float)m * 5.9604644775390625e-08f;                  // m 2^-24, exact
        memcpy(&u, &v, 4);
        u |= sign;
    } else u = sign | ((e + 112u) << 23) | (m << 13);
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// v_mfma_f32_16x16x32_f16: operand layout of the bf16 form above, two fp16 per dword
inline f32x4_e mfma_16x16x32f16(u32x4_e a, u32x4_e b, f32x4_e c) {
    Machine& m = M();
    Fiber& f = cur();
    WaveScratch& w = m.ws[f.wave];
    unsigned ab[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    memcpy(w.slot[f.lane], ab, 32);
    wave_sync();
    int j = f.lane & 15, q = f.lane >> 4;
    auto hf = [](unsigned dw, int half) { return f16_bits_to_f32(half ? (dw >> 16) : (dw & 0xffffu)); };
    for (int r = 0; r < 4; r++) {
        int i = q * 4 + r;
        float acc = c[r];
        for (int kq = 0; kq < 4; kq++) {
            unsigned av[8], bv[8];
            memcpy(av, w.slot[i + 16 * kq], 32);
            memcpy(bv, w.slot[j + 16 * kq], 32);
            for (int s = 0; s < 8; s++) acc += hf(av[s >> 1], s & 1) * hf(bv[4 + (s >> 1)], s & 1);
        }
        c[r] = acc;
    }
    wave_sync();
    return c;
}

// v_mfma_f32_32x32x16_bf16: lane i + 32 h holds k-steps 8 h .. 8 h + 7 of row i (A) / column i (B); D as for 32x32x2: col = l & 31,
// row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)
inline f32x16_e mfma_32x32x16bf16(u32x4_e a, u32x4_e b, f32x16_e c) {
    Machine& m = M();
    Fiber& f = cur();
    WaveScratch& w = m.ws[f.wave];
    unsigned ab[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    memcpy(w.slot[f.lane], ab, 32);
    wave_sync();
    int j = f.lane & 31, h = f.lane >> 5;
    auto bf = [](unsigned dw, int half) { unsigned u = half ? (dw & 0xffff0000u) : (dw << 16); float v; memcpy(&v, &u, 4); return v; };
    for (int r = 0; r < 16; r++) {
        int i = (r & 3) + 8 * (r >> 2) + 4 * h;
        float acc = c[r];
        for (int kh = 0; kh < 2; kh++) {
            unsigned av[8], bv[8];
            memcpy(av, w.slot[i + 32 * kh], 32);
            memcpy(bv, w.slot[j + 32 * kh], 32);
            for (int s = 0; s < 8; s++) acc += bf(av[s >> 1], s & 1) * bf(bv[4 + (s >> 1)], s & 1);
        }
        c[r] = acc;
    }
    wave_sync();
    return c;
}

// v_mfma_f32_32x32x16_f16: operand layout of the bf16 form above, two fp16 per dword
inline f32x16_e mfma_32x32x16f16(u32x4_e a, u32x4_e b, f32x16_e c) {
    Machine& m = M();
    Fiber& f = cur();
    WaveScratch& w = m.ws[f.wave];
    unsigned ab[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    memcpy(w.slot[f.lane], ab, 32);
    wave_sync();
    int j = f.lane & 31, h = f.lane >> 5;
    auto hf = [](unsigned dw, int half) { return f16_bits_to_f32(half ? (dw >> 16) : (dw & 0xffffu)); };
    for (int r = 0; r < 16; r++) {
        int i = (r & 3) + 8 * (r >> 2) + 4 * h;
        float acc = c[r];
        for (int kh = 0; kh < 2; kh++) {
            unsigned av[8], bv[8];
            memcpy(av, w.slot[i + 32 * kh], 32);
            memcpy(bv, w.slot[j + 32 * kh], 32);
            for (int s = 0; s < 8; s++) acc += hf(av[s >> 1], s & 1) * hf(bv[4 + (s >> 1)], s & 1);
        }
        c[r] = acc;
    }
    wave_sync();
    return c;
}

}  // namespace emul

#define threadIdx (emul::cur().tid)
#define blockIdx (emul::M().bid)
#define blockDim (emul::M().bdim)
#define gridDim (emul::M().gdim)
#define warpSize 64

// dynamic shared memory (`extern __shared__`, spelled D3H_DYN_SHARED in the kernels): blocks run one after the other, so one buffer
// sized by the launch's shmem argument serves them all
namespace emul {
inline std::vector<unsigned char>& dyn_shared_buf() { static std::vector<unsigned char> b; return b; }
inline void* dyn_shared(size_t bytes = 0) {
    auto& b = dyn_shared_buf();
    if (bytes > b.size()) b.resize(bytes + 64);
    return (void*)(((uintptr_t)b.data() + 15) & ~(uintptr_t)15);
}
}  // namespace emul
#define D3H_DYN_SHARED(type, name) type* name = (type*)emul::dyn_shared()
// global_load_lds_dwordx4: lane i copies 16 bytes from its own global address to (wave-uniform LDS base) + 16 i
// DPP row shift / readlane of d3h_common.h on the fiber wave: lanes whose source falls outside their row of 16 read 0
#define D3H_ROW_SHR(v, N) (((emul::cur().lane & 15) >= (N)) ? emul::exchange((float)(v), emul::cur().lane - (N)) : (emul::exchange((float)(v), emul::cur().lane), 0.f))
#define D3H_READLANE(v, L) emul::exchange((float)(v), (L))
#define D3H_WAVE_SYNC() emul::wave_sync()
#define D3H_WAVES_PER_EU(n)
#define D3H_SCHED_FENCE()
#define D3H_GLDS16(gsrc, lds_wave_base) memcpy((char*)(lds_wave_base) + 16 * emul::cur().lane, (const void*)(gsrc), 16)

#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
    (emul::dyn_shared((size_t)(shmem)), emul::launch((grid), (block), [=]() { kern(__VA_ARGS__); }))

static inline void __syncthreads() { emul::block_sync(); }
#define __builtin_amdgcn_s_barrier() emul::block_sync()
#define __builtin_amdgcn_mfma_f32_32x32x2f32 emul::mfma_32x32x2f32
#define __builtin_amdgcn_mfma_f32_16x16x4f32 emul::mfma_16x16x4f32
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_nontemporal_store(v, p) (*(p) = (v))
#define __builtin_amdgcn_sched_group_barrier(a, b, c) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)

static inline unsigned long long __ballot(int p) { return emul::ballot(p); }
template <class T> static inline T __shfl(T v, int src, int width = 64) {
    int lane = emul::cur().lane;
    int base = lane & ~(width - 1);
    return emul::exchange(v, base + (src & (width - 1)));
}
template <class T> static inline T __shfl_xor(T v, int mask, int width = 64) { return emul::exchange(v, emul::cur().lane ^ mask); }
template <class T> static inline T __shfl_up(T v, unsigned d, int width = 64) {
    int lane = emul::cur().lane;
    int src = (lane & (width - 1)) >= (int)d ? lane - (int)d : lane;
    return emul::exchange(v, src);
}
template <class T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
    int lane = emul::cur().lane;
    int src = (lane & (width - 1)) + (int)d < width ? lane + (int)d : lane;
    return emul::exchange(v, src);
}
template <class T> static inline T __builtin_amdgcn_readfirstlane_e(T v) {
    unsigned long long m = emul::ballot(1);
    return emul::exchange(v, __builtin_ctzll(m));
}
#define __builtin_amdgcn_readfirstlane(v) __builtin_amdgcn_readfirstlane_e(v)
static inline int __lane_id() { return emul::cur().lane; }

static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
static inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
static inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
static inline float __fmul_rn(float a, float b) { return a * b; }
static inline float __fadd_rn(float a, float b) { return a + b; }
static inline float __fsub_rn(float a, float b) { return a - b; }
static inline float __fdiv_rn(float a, float b) { return a / b; }
static inline float __frcp_rn(float a) { return 1.0f / a; }
static inline float rsqrtf(float a) { return 1.0f / sqrtf(a); }
#define __expf(a) expf(a)
#define __builtin_amdgcn_exp2f(a) exp2f(a)
#define __builtin_amdgcn_logf(a) log2f(a)
#define __logf(a) logf(a)
static inline float __saturatef(float a) { return a < 0.f ? 0.f : (a > 1.f ? 1.f : a); }
using std::min;
using std::max;

template <class T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
static inline float atomicAdd(float* p, double v) { float o = *p; *p = o + (float)v; return o; }
template <class T> static inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> static inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <class T> static inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <class T> static inline T atomicCAS(T* p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }
static inline void __threadfence() {}
static inline void __threadfence_system() {}
static inline void __threadfence_block() {}
