// sdf_mlp_x3.h -- fp32 GEMMs of the SDF network on the bf16 matrix pipe: the "x3" operand split.
//
// gfx950 runs the exact-f32 MFMA (v_mfma_f32_16x16x4_f32) at 64 FLOP/clk/SIMD -- the VALU rate, 1/16 of the bf16 forms
// (MI355X_MICROARCH.md, Matrix cores).  A fp32 product a*b is recovered on the fast pipe by writing each operand as the sum of three
// bf16 numbers, a = a_h + a_m + a_l with a_h = bf16(a), a_m = bf16(a - a_h), a_l = bf16(a - a_h - a_m) (3 x 8 significant bits; both
// subtractions are exact in fp32), and keeping the six partial products down to 2^-16 relative weight:
//       a*b  ~  a_h b_h + (a_h b_m + a_m b_h) + (a_h b_l + a_l b_h + a_m b_m)              dropped: a_m b_l + a_l b_m + a_l b_l  (<= 2^-23 |ab|)
// Every bf16 x bf16 product is exact in the MFMA's fp32 datapath and the accumulation is fp32, so a dot product carries the same kind of
// error as a fp32 FMA chain: measured on the reference's network shape (tests/test_sdf_x3.py, tools/dbg/bf16x3_accuracy.py) the output
// differs from float64 by no more than the plain-fp32 evaluation does (max 7e-7 vs 9e-7 on |sdf| ~ 0.7), signs included.
// Six v_mfma_f32_16x16x32_bf16 (16 cycles each) replace eight v_mfma_f32_16x16x4_f32 (32 cycles each) per 16x16x32 block: 96 vs 256
// matrix-pipe cycles, and -- unlike the f32 MFMA -- the bf16 MFMA leaves half of its issue cycles to the VALU of the SIMD's other wave.
//
// Operand layout of v_mfma_f32_16x16x32_bf16 (one dword = two bf16, low half first): lane = i + 16 q holds, in 4 dwords, elements
// s = 0..7 = k-steps 8 q + s of row i (A) / column i (B); D as for the 16x16x4 form: col = lane & 15, row = 4 (lane >> 4) + r.
// The D registers of two neighbouring 16-feature blocks (rb = 2 kb, 2 kb + 1) of a layer's output ARE the B operand of k-block kb of the
// next layer when k-step (q, s) is paired with feature  32 kb + 16 (s >> 2) + 4 q + (s & 3)  -- the packed A fragments use that order, so
// activations stay in registers between layers exactly as in the f32 kernels (sdf_mlp_layout.h).
//
// wpack3 (dwords), consumed in stream order by sdf_mlp_fwd_x3_kernel:
//   layer 0 (net.0)               2 chunks of [rbl 8][kb EMB_KB][part 3][lane 64][4]          (embedding padded to 32 EMB_KB features)
//   layers 1,2,3,5,6              8 chunks of [rbl 2][kb 8][part 3][lane 64][4]
//   layer 4 (net.8, skip)         8 chunks of [rbl 2][kb 8 + EMB_KB][part 3][lane 64][4]       (kb >= 8: embedding)
//   tail (fp32): bias0..bias6, W7, b7 exactly as in wpack
#pragma once
#include "sdf_mlp_dev.h"

namespace D3H_MLP_NS {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int X3_EMB_KB = (EMB_DIM + 31) / 32;                 // 2
constexpr int X3_FRAG = 256;                                   // dwords per A fragment: 64 lanes x 16 B
constexpr int X3_SKIP_KB = 8 + X3_EMB_KB;                      // 10

// Layout of a forward pack with NP operand planes: NP = 3 is the bf16 x 3 split above (six products per block), NP = 2 the fp16 x 2 split
// of the "h2" section below (three products per block).  Same chunk structure, [part NP] instead of [part 3].
template <int NP>
struct XP {
    static constexpr int L0_CHUNK = 8 * X3_EMB_KB * NP * X3_FRAG;
    static constexpr int HID_CHUNK = 2 * 8 * NP * X3_FRAG;
    static constexpr int SKIP_CHUNK = 2 * X3_SKIP_KB * NP * X3_FRAG;
    static constexpr int CHUNK_MAX = SKIP_CHUNK;
    static constexpr int OFF_L1 = 2 * L0_CHUNK;
    static constexpr int OFF_L2 = OFF_L1 + 8 * HID_CHUNK;
    static constexpr int OFF_L3 = OFF_L2 + 8 * HID_CHUNK;
    static constexpr int OFF_L4 = OFF_L3 + 8 * HID_CHUNK;
    static constexpr int OFF_L5 = OFF_L4 + 8 * SKIP_CHUNK;
    static constexpr int OFF_L6 = OFF_L5 + 8 * HID_CHUNK;
    static constexpr int OFF_TAIL = OFF_L6 + 8 * HID_CHUNK;
    static constexpr int WPACK_DWORDS = OFF_TAIL + BIAS_FLOATS;
    __host__ __device__ static inline int layer_offset(int l) {
        switch (l) {
            case 0: return 0;
            case 1: return OFF_L1;
            case 2: return OFF_L2;
            case 3: return OFF_L3;
            case 4: return OFF_L4;
            case 5: return OFF_L5;
            case 6: return OFF_L6;
            default: return OFF_TAIL;
        }
    }
    __host__ __device__ static inline int layer_of_offset(int idx) {
        if (idx < OFF_L1) return 0;
        if (idx < OFF_L2) return 1;
        if (idx < OFF_L3) return 2;
        if (idx < OFF_L4) return 3;
        if (idx < OFF_L5) return 4;
        if (idx < OFF_L6) return 5;
        return 6;
    }
};
constexpr int X3_L0_CHUNK = XP<3>::L0_CHUNK;                   // 12288 dwords (48 KiB)
constexpr int X3_HID_CHUNK = XP<3>::HID_CHUNK;                 // 12288
constexpr int X3_SKIP_CHUNK = XP<3>::SKIP_CHUNK;               // 15360 (60 KiB)
constexpr int X3_CHUNK_MAX = X3_SKIP_CHUNK;
constexpr int X3_OFF_L1 = XP<3>::OFF_L1;
constexpr int X3_OFF_TAIL = XP<3>::OFF_TAIL;
constexpr int X3_WPACK_DWORDS = XP<3>::WPACK_DWORDS;
constexpr int X3_STAGE_F4 = (X3_CHUNK_MAX / 4 + NTHREADS - 1) / NTHREADS;      // 8 x 16 B per thread per chunk

__host__ __device__ inline int x3_layer_offset(int l) { return XP<3>::layer_offset(l); }
__host__ __device__ inline int x3_layer_of_offset(int idx) { return XP<3>::layer_of_offset(idx); }
// input feature of k-step (q, s) of k-block kb
__host__ __device__ inline int x3_feature(int kb, int q, int s) { return 32 * kb + 16 * (s >> 2) + 4 * q + (s & 3); }

#ifndef D3H_EMULATED
typedef __bf16 d3h_bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 d3h_bf16x8 __attribute__((ext_vector_type(8)));
// two floats -> one dword of two bf16 (round to nearest even; v_cvt_pk_bf16_f32), `lo` in the low half
__device__ __forceinline__ unsigned x3_pk(float lo, float hi) {
    d3h_bf16x2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, v);
}
#define D3H_MFMA_BF16X8(a, b, c) \
    __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(d3h_bf16x8, a), __builtin_bit_cast(d3h_bf16x8, b), c, 0, 0, 0)
#else
__device__ __forceinline__ unsigned x3_bf16_bits(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;      // NaN stays a NaN
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ unsigned x3_pk(float lo, float hi) { return x3_bf16_bits(lo) | (x3_bf16_bits(hi) << 16); }
#define D3H_MFMA_BF16X8(a, b, c) emul::mfma_16x16x32bf16(a, b, c)
#endif
__device__ __forceinline__ float x3_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float x3_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// (a, b) -> the three bf16 pairs h + m + l ~ (a, b)
__device__ __forceinline__ void x3_split_pair(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    h = x3_pk(a, b);
    const float ra = a - x3_lo(h), rb = b - x3_hi(h);
    m = x3_pk(ra, rb);
    const float sa = ra - x3_lo(m), sb = rb - x3_hi(m);
    l = x3_pk(sa, sb);
}
// two D blocks (features 32 kb + 4 q + r and 32 kb + 16 + 4 q + r) -> the B operand triple of k-block kb
__device__ __forceinline__ void x3_split_blocks(const f32x4 v0, const f32x4 v1, u32x4 (&out)[3]) {
#ifdef D3H_X3_PROBE_NOSPLIT       // (timing probe; results are wrong)
    out[0] = u32x4{__float_as_uint(v0[0]), __float_as_uint(v0[1]), __float_as_uint(v0[2]), __float_as_uint(v0[3])};
    out[1] = u32x4{__float_as_uint(v1[0]), __float_as_uint(v1[1]), __float_as_uint(v1[2]), __float_as_uint(v1[3])};
    out[2] = out[0];
    return;
#endif
    unsigned h[4], m[4], l[4];
    x3_split_pair(v0[0], v0[1], h[0], m[0], l[0]);
    x3_split_pair(v0[2], v0[3], h[1], m[1], l[1]);
    x3_split_pair(v1[0], v1[1], h[2], m[2], l[2]);
    x3_split_pair(v1[2], v1[3], h[3], m[3], l[3]);
    out[0] = u32x4{h[0], h[1], h[2], h[3]};
    out[1] = u32x4{m[0], m[1], m[2], m[3]};
    out[2] = u32x4{l[0], l[1], l[2], l[3]};
}

// acc += (W_h + W_m + W_l)(x_h + x_m + x_l) over one k-block, small terms first; a[part], x[part]: 0 = h, 1 = m, 2 = l
__device__ __forceinline__ void x3_mac(f32x4& acc, const u32x4 a0, const u32x4 a1, const u32x4 a2, const u32x4 (&x)[3]) {
    acc = D3H_MFMA_BF16X8(a2, x[0], acc);
    acc = D3H_MFMA_BF16X8(a0, x[2], acc);
    acc = D3H_MFMA_BF16X8(a1, x[1], acc);
    acc = D3H_MFMA_BF16X8(a1, x[0], acc);
    acc = D3H_MFMA_BF16X8(a0, x[1], acc);
    acc = D3H_MFMA_BF16X8(a0, x[0], acc);
}

// ---- "h2": the fp16 x 2 split with a scaled residual plane -- THREE products per block instead of six (round 6) ---------------------------
// a = a_h + 2^-11 a_m' with a_h = fp16(a), a_m' = fp16(2^11 (a - a_h)): a - a_h is exact in fp32 and at most 2^-11 |a|, so the pair carries
// 22 significant bits, |a - a_h - 2^-11 a_m'| <= 2^-22 |a|.  A product keeps  a_h b_h + 2^-11 (a_h b_m' + a_m' b_h)  and drops 2^-22 a_m' b_m'
// (<= 2^-22 |ab|): per-product error <= 3 2^-22 |ab| worst case, random in sign, against the 2^-24 rounding of EVERY partial sum of an fp32
// accumulation over 256 terms -- measured on the fitted network (tools/dbg/fp16x2_accuracy.py): max |error| vs float64 7.1e-7 on |sdf| <= 1.6, the
// plain-fp32 evaluation's own figure, signs equal.  The residual plane is scaled so that it stays a NORMAL fp16 number down to |a - a_h| ~ 3e-8
// (unscaled, every residual of an operand below 0.12 would be subnormal: measured 2 x the error); the two cross products are accumulated
// apart (`lo`) and folded in once per 16-row block: acc = hi + 2^-11 lo.  fp16 has 5 exponent bits: operands above 65 504 overflow -- the
// network's activations and weights are O(1) (softplus of a unit-scale SDF); the pack kernel refuses (NaN-fills) weights above 2^14, and the
// sweeps whose operands have no natural scale (gradients: the data-backward and weight-gradient kernels) stay on the bf16 x 3 split.
constexpr float H2_SCALE = 2048.0f, H2_INV_SCALE = 1.0f / 2048.0f;
#ifndef D3H_EMULATED
typedef _Float16 d3h_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 d3h_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned h2_pk(float lo, float hi) {
    d3h_f16x2 v = {(_Float16)lo, (_Float16)hi};          // v_cvt_pk_f16_f32: round to nearest even
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float h2_lo(unsigned u) { return (float)__builtin_bit_cast(d3h_f16x2, u)[0]; }
__device__ __forceinline__ float h2_hi(unsigned u) { return (float)__builtin_bit_cast(d3h_f16x2, u)[1]; }
#define D3H_MFMA_F16X8(a, b, c) \
    __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(d3h_f16x8, a), __builtin_bit_cast(d3h_f16x8, b), c, 0, 0, 0)
#else
__device__ __forceinline__ unsigned h2_pk(float lo, float hi) { return emul::f32_to_f16_bits(lo) | (emul::f32_to_f16_bits(hi) << 16); }
__device__ __forceinline__ float h2_lo(unsigned u) { return emul::f16_bits_to_f32(u & 0xffffu); }
__device__ __forceinline__ float h2_hi(unsigned u) { return emul::f16_bits_to_f32(u >> 16); }
#define D3H_MFMA_F16X8(a, b, c) emul::mfma_16x16x32f16(a, b, c)
#endif
__device__ __forceinline__ void h2_split_pair(float a, float b, unsigned& h, unsigned& m) {
    h = h2_pk(a, b);
    m = h2_pk((a - h2_lo(h)) * H2_SCALE, (b - h2_hi(h)) * H2_SCALE);
}

// operand planes of two D blocks, any plane count
template <int NP>
__device__ __forceinline__ void xp_split_blocks(const f32x4 v0, const f32x4 v1, u32x4 (&out)[NP]) {
    if constexpr (NP == 3) {
        x3_split_blocks(v0, v1, out);
    } else {
        unsigned h[4], m[4];
        h2_split_pair(v0[0], v0[1], h[0], m[0]);
        h2_split_pair(v0[2], v0[3], h[1], m[1]);
        h2_split_pair(v1[0], v1[1], h[2], m[2]);
        h2_split_pair(v1[2], v1[3], h[3], m[3]);
        out[0] = u32x4{h[0], h[1], h[2], h[3]};
        out[1] = u32x4{m[0], m[1], m[2], m[3]};
    }
}

// THE CO-RESIDENCY RULE (round 5; reproducers: tools/probe/mfma_pk_hazard.cpp -- self-contained, no library -- and
// tools/probe/coresidency_repro.cpp -- this library's kernels; measurements: profiles/r5_hazard_*.txt).
// On MI355X a wave that executes PACKED-F32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: what hipcc -O3 SLP-packs adjacent scalar
// f32 arithmetic into -- the 3x3 algebra of lbs_bwd_kernel, most of image_ops.hip, any torch elementwise kernel) gets WRONG VALUES IN LANES 48..63
// (the last of the four 16-lane passes) when it shares a SIMD with a wave of ANOTHER kernel that is issuing bf16 MFMAs back to back.  Scalar f32
// arithmetic is never affected, the same kernels are never affected when they are stream-ordered, and f32-input MFMAs did not trigger it.  It is a
// property of the hardware, not of these kernels: a register-only MFMA loop next to a 40-line skinning kernel reproduces it (10 % of the victim
// launches), and it follows the aggressor waves' ends -- a claiming workgroup whose waves finish at different times still lets a foreign wave onto a
// SIMD whose other wave is mid-MFMA.  What makes it impossible, measured (0 wrong launches of 960 where the bare loop gives 110):
//   (i)  the kernel is allocated all 256 VGPRs and runs 512-thread workgroups: its two waves per SIMD own the register file, no foreign wave fits;
//   (ii) every wave passes a workgroup barrier AFTER its last MFMA: no wave leaves its SIMD while a sibling still issues matrix instructions.
// Every bf16-MFMA kernel of the library does both (the sweeps need 256 registers anyway and end every weight chunk with a barrier; the injected
// reverse sweep needs 236, the weight-gradient kernel 180 and a barrier closes each of its tiles); tests/test_mfma_claim.py checks (i) and (ii) in
// the ISA at build time, tests/test_gpu_hazard.py runs the reproducers on the GPU.  -DD3H_DWX_SHARE_SIMDS builds the kernels WITHOUT the claim
// (the build in which main-stream kernels returned wrong gradients in 4-40 ticks of 96 in round 4).
#if !defined(D3H_DWX_SHARE_SIMDS) && !defined(D3H_EMULATED)
#define D3H_X3_CLAIM_SIMD() asm volatile("v_mov_b32 v255, 0" ::: "v255")
#else
#define D3H_X3_CLAIM_SIMD() ((void)0)
#endif

// ---- weight-chunk staging and the k-loop of one 16-row output block (shared by the forward and the data-backward kernels) ----------
__device__ __forceinline__ void x3_issue(const unsigned* __restrict__ src, unsigned* dst, int n4, int tid) {
#ifdef D3H_X3_PROBE_NOSTAGE       // (timing probe: no weight stream; results are wrong)
    return;
#endif
    const int wave_base = tid & ~63;
#pragma unroll
    for (int i = 0; i < X3_STAGE_F4; ++i) {
        const int j = tid + i * NTHREADS;
        if (j < n4) D3H_GLDS16(src + 4 * (size_t)j, dst + 4 * (wave_base + i * NTHREADS));
    }
}

// acc += W[16 rows of one block][32 NKB inputs] x over the NKB k-blocks of xs; wl -> [kb][part 3][lane 64][4].  One set of A fragments:
// the MFMA order retires the l plane after the first product of a k-block and the m plane after the third, and each plane's fragment of
// k-block kb + 1 is requested right after its last use (the h plane, used last, is needed again only at the fourth MFMA of the next
// k-block).  `mid` runs before k-block MID (MID < 0: never).
#ifndef D3H_X3_MAC
#define D3H_X3_MAC 2
#endif
template <int NKB, int MID, class F>
__device__ __forceinline__ void x3_mac_blocks(f32x4& acc, const u32x4 (&xs)[NKB][3], const unsigned* wl, int lane, F&& mid) {
    const unsigned* p = wl + lane * 4;
    u32x4 a0 = *(const u32x4*)(p), a1 = *(const u32x4*)(p + X3_FRAG), a2 = *(const u32x4*)(p + 2 * X3_FRAG);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        if (kb == MID) mid();
        const unsigned* pn = p + (kb + 1) * 3 * X3_FRAG;
#if defined(D3H_X3_PROBE_NOLDS)     // (timing probe: one set of fragments per block; results are wrong)
        x3_mac(acc, a0, a1, a2, xs[kb]);
        (void)pn;
#elif D3H_X3_MAC == 0
        acc = D3H_MFMA_BF16X8(a2, xs[kb][0], acc);
        if (kb + 1 < NKB) a2 = *(const u32x4*)(pn + 2 * X3_FRAG);
        acc = D3H_MFMA_BF16X8(a1, xs[kb][1], acc);
        acc = D3H_MFMA_BF16X8(a1, xs[kb][0], acc);
        if (kb + 1 < NKB) a1 = *(const u32x4*)(pn + X3_FRAG);
        acc = D3H_MFMA_BF16X8(a0, xs[kb][2], acc);
        acc = D3H_MFMA_BF16X8(a0, xs[kb][1], acc);
        acc = D3H_MFMA_BF16X8(a0, xs[kb][0], acc);
        if (kb + 1 < NKB) a0 = *(const u32x4*)(pn);
#elif D3H_X3_MAC == 1      // all three fragments of the next k-block requested before this k-block's MFMAs, pinned
        u32x4 n0 = a0, n1 = a1, n2 = a2;
        if (kb + 1 < NKB) { n0 = *(const u32x4*)(pn); n1 = *(const u32x4*)(pn + X3_FRAG); n2 = *(const u32x4*)(pn + 2 * X3_FRAG); }
        D3H_SCHED_FENCE();
        x3_mac(acc, a0, a1, a2, xs[kb]);
        D3H_SCHED_FENCE();
        a0 = n0; a1 = n1; a2 = n2;
#elif D3H_X3_MAC == 2      // the rotating order, pinned: each plane's next fragment is requested right after its last MFMA
        acc = D3H_MFMA_BF16X8(a2, xs[kb][0], acc);
        D3H_SCHED_FENCE();
        if (kb + 1 < NKB) a2 = *(const u32x4*)(pn + 2 * X3_FRAG);
        D3H_SCHED_FENCE();
        acc = D3H_MFMA_BF16X8(a1, xs[kb][1], acc);
        acc = D3H_MFMA_BF16X8(a1, xs[kb][0], acc);
        D3H_SCHED_FENCE();
        if (kb + 1 < NKB) a1 = *(const u32x4*)(pn + X3_FRAG);
        D3H_SCHED_FENCE();
        acc = D3H_MFMA_BF16X8(a0, xs[kb][2], acc);
        acc = D3H_MFMA_BF16X8(a0, xs[kb][1], acc);
        acc = D3H_MFMA_BF16X8(a0, xs[kb][0], acc);
        D3H_SCHED_FENCE();
        if (kb + 1 < NKB) a0 = *(const u32x4*)(pn);
        D3H_SCHED_FENCE();
#endif
    }
}

struct X3None {
    __device__ __forceinline__ void operator()() const {}
};

// h2 k-loop of one 16-row block: hi += W_h x_h, lo += W_m' x_h + W_h x_m' over the NKB k-blocks; wl -> [kb][part 2][lane 64][4].  The residual
// plane's fragment is requested again right after its only use, the main plane's after its second (the rotating order of D3H_X3_MAC == 2).
template <int NKB, int MID, class F>
__device__ __forceinline__ void h2_mac_blocks(f32x4& hi, f32x4& lo, const u32x4 (&xs)[NKB][2], const unsigned* wl, int lane, F&& mid) {
    const unsigned* p = wl + lane * 4;
    u32x4 a0 = *(const u32x4*)(p), a1 = *(const u32x4*)(p + X3_FRAG);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        if (kb == MID) mid();
        const unsigned* pn = p + (kb + 1) * 2 * X3_FRAG;
        lo = D3H_MFMA_F16X8(a1, xs[kb][0], lo);
        D3H_SCHED_FENCE();
        if (kb + 1 < NKB) a1 = *(const u32x4*)(pn + X3_FRAG);
        D3H_SCHED_FENCE();
        lo = D3H_MFMA_F16X8(a0, xs[kb][1], lo);
        hi = D3H_MFMA_F16X8(a0, xs[kb][0], hi);
        D3H_SCHED_FENCE();
        if (kb + 1 < NKB) a0 = *(const u32x4*)(pn);
        D3H_SCHED_FENCE();
    }
}
// either split: (hi, lo) accumulate one 16-row block over NKB k-blocks; xp_fold gives the block's value
template <int NP, int NKB, int MID, class F>
__device__ __forceinline__ void xp_mac_blocks(f32x4& hi, f32x4& lo, const u32x4 (&xs)[NKB][NP], const unsigned* wl, int lane, F&& mid) {
    if constexpr (NP == 3) x3_mac_blocks<NKB, MID>(hi, xs, wl, lane, mid);
    else h2_mac_blocks<NKB, MID>(hi, lo, xs, wl, lane, mid);
}
template <int NP>
__device__ __forceinline__ f32x4 xp_fold(const f32x4 hi, const f32x4 lo) {
    if constexpr (NP == 3) return hi;
    else {
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = fmaf(lo[r], H2_INV_SCALE, hi[r]);
        return o;
    }
}

// transposed pack (backward data, dH_{l-1}^T = W_l^T dZ_l^T), consumed in the chunk order of wpackT (sdf_mlp_layout.h: L6, L5, L4 with 8 hidden +
// 2 embedding in-chunks, L3, L2, L1, L0 with 2 embedding in-chunks); every chunk [rbl 2][kb 8][part 3][lane 64][4] with
// element s of lane (i, q) = W_l[out = x3_feature(kb, q, s)][in = 16 (2 c + rbl) + i]
constexpr int X3_WPACKT_DWORDS = 52 * X3_HID_CHUNK;
template <int NP>
struct XPT {          // the transposed pack with NP planes (3: bf16 x 3, 2: fp16 x 2 "h2"): 52 chunks of [rbl 2][kb 8][part NP][lane 64][4]
    static constexpr int HID_CHUNK = XP<NP>::HID_CHUNK;
    static constexpr int WPACKT_DWORDS = 52 * HID_CHUNK;
};

// ---- operand scale of the h2 sweeps whose operands are GRADIENTS (data-backward, weight-gradient): a power of two s such that (largest
// |upstream gradient|) * s lands near 2^H2_GRAD_TOP, i.e. with 2^(16 - H2_GRAD_TOP) of head-room below the fp16 maximum for the growth of
// dZ through the layers, and >= 2^-14 / 2^H2_GRAD_TOP = 2^-20 of the largest entry still a NORMAL fp16 number (smaller entries degrade
// gracefully through the subnormals and the scaled residual plane).  Values in memory (dz, e, t, dx) are always true-scaled: a sweep scales its
// inputs as it loads them and un-scales what it stores.  sc[0] = s, sc[1] = 1 / s.
constexpr int H2_GRAD_TOP = 6;
__host__ __device__ inline float h2_grad_scale(float maxabs) {
    if (!(maxabs > 0.f) || !(maxabs < 3.0e38f)) return 1.0f;
    int e;
    (void)frexpf(maxabs, &e);                  // maxabs = m 2^e, m in [0.5, 1)
    int k = H2_GRAD_TOP - e;
    k = k > 100 ? 100 : (k < -100 ? -100 : k);
    return ldexpf(1.0f, k);
}

}  // namespace D3H_MLP_NS
