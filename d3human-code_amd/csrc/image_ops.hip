// image_ops.hip -- mesh normals, shading normal, image-space losses, SSIM and the SDF edge regulariser on gfx950.
//
// Replaces (reference file:line):
//   render/mesh.py:418-446 auto_normals (== geometry/gshell_tets.py:9-33)      area-weighted vertex normals (scatter-add)
//   render/render.py:261-265 face normals                                      safe_normalize(cross(v1-v0, v2-v0))
//   render/renderutils/c_src/normal.cu:98,128 PrepareShadingNormal{Fwd,Bwd}     (python twin: renderutils/bsdf.py:46-51)
//   render/renderutils/c_src/loss.cu:95,137 imgLoss{Fwd,Bwd}                    l1/mse/smape/relmse x none/log_srgb
//   ssim_loss.py:33-63 ssim                                                     11x11 Gaussian (sigma 1.5), zero padding
//   geometry/hmsdf.py:162-170 compute_sdf_reg_loss                              BCE-with-logits on sign-changing grid edges
// All of these are HBM-streaming passes (O(10) flop per element): one coalesced sweep each, block-level reductions
// (wave shuffles + one atomic per workgroup) for the scalar outputs.
#include "d3h_vec.h"
#include "composite.h"

namespace {

// ---- mesh normals ---------------------------------------------------------------------------------
// all mesh-normal kernels take a batch of vertex sets sharing one face list: blockIdx.y = batch item, vertex stride nv*3
__global__ void face_cross_scatter_kernel(const float* __restrict__ v, const int* __restrict__ f, int nf, float* __restrict__ vn_raw, size_t bs) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nf) return;
    v += blockIdx.y * bs; vn_raw += blockIdx.y * bs;
    int i0 = f[3 * (size_t)i], i1 = f[3 * (size_t)i + 1], i2 = f[3 * (size_t)i + 2];
    V3 v0 = ld3(v + 3 * (size_t)i0), v1 = ld3(v + 3 * (size_t)i1), v2 = ld3(v + 3 * (size_t)i2);
    V3 n = cross(v1 - v0, v2 - v0);
    atomic_add3(vn_raw + 3 * (size_t)i0, n);
    atomic_add3(vn_raw + 3 * (size_t)i1, n);
    atomic_add3(vn_raw + 3 * (size_t)i2, n);
}
__global__ void vnormal_finish_kernel(const float* __restrict__ vn_raw, int nv, float* __restrict__ vn) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;          // nv already counts all batch items (flat)
    if (i >= nv) return;
    V3 r = ld3(vn_raw + 3 * (size_t)i);
    if (!(dot(r, r) > 1e-20f)) r = mk(0.f, 0.f, 1.f);      // mesh.py:439
    st3(vn + 3 * (size_t)i, safe_normalize(r));
}
__global__ void vnormal_finish_bwd_kernel(const float* __restrict__ vn_raw, const float* __restrict__ g_vn, int nv, float* __restrict__ g_raw) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    V3 r = ld3(vn_raw + 3 * (size_t)i);
    V3 g = mk(0.f, 0.f, 0.f);
    if (dot(r, r) > 1e-20f) g = safe_normalize_bwd(r, ld3(g_vn + 3 * (size_t)i));
    st3(g_raw + 3 * (size_t)i, g);
}
__global__ void face_cross_scatter_bwd_kernel(const float* __restrict__ v, const int* __restrict__ f, int nf, const float* __restrict__ g_raw,
                                              float* __restrict__ d_v, size_t bs) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nf) return;
    v += blockIdx.y * bs; g_raw += blockIdx.y * bs; d_v += blockIdx.y * bs;
    int i0 = f[3 * (size_t)i], i1 = f[3 * (size_t)i + 1], i2 = f[3 * (size_t)i + 2];
    V3 v0 = ld3(v + 3 * (size_t)i0), v1 = ld3(v + 3 * (size_t)i1), v2 = ld3(v + 3 * (size_t)i2);
    V3 gn = ld3(g_raw + 3 * (size_t)i0) + ld3(g_raw + 3 * (size_t)i1) + ld3(g_raw + 3 * (size_t)i2);
    V3 e1 = v1 - v0, e2 = v2 - v0;
    V3 ge1 = cross(e2, gn), ge2 = cross(gn, e1);
    atomic_add3(d_v + 3 * (size_t)i1, ge1);
    atomic_add3(d_v + 3 * (size_t)i2, ge2);
    atomic_add3(d_v + 3 * (size_t)i0, (ge1 + ge2) * -1.0f);
}
__global__ void face_normals_kernel(const float* __restrict__ v, const int* __restrict__ f, int nf, float* __restrict__ fn, size_t bs) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nf) return;
    v += blockIdx.y * bs; fn += blockIdx.y * (size_t)nf * 3;
    V3 v0 = ld3(v + 3 * (size_t)f[3 * (size_t)i]), v1 = ld3(v + 3 * (size_t)f[3 * (size_t)i + 1]), v2 = ld3(v + 3 * (size_t)f[3 * (size_t)i + 2]);
    st3(fn + 3 * (size_t)i, safe_normalize(cross(v1 - v0, v2 - v0)));
}
__global__ void face_normals_bwd_kernel(const float* __restrict__ v, const int* __restrict__ f, int nf, const float* __restrict__ g_fn,
                                        float* __restrict__ d_v, size_t bs) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nf) return;
    v += blockIdx.y * bs; d_v += blockIdx.y * bs; g_fn += blockIdx.y * (size_t)nf * 3;
    int i0 = f[3 * (size_t)i], i1 = f[3 * (size_t)i + 1], i2 = f[3 * (size_t)i + 2];
    V3 v0 = ld3(v + 3 * (size_t)i0), v1 = ld3(v + 3 * (size_t)i1), v2 = ld3(v + 3 * (size_t)i2);
    V3 e1 = v1 - v0, e2 = v2 - v0;
    V3 gn = safe_normalize_bwd(cross(e1, e2), ld3(g_fn + 3 * (size_t)i));
    V3 ge1 = cross(e2, gn), ge2 = cross(gn, e1);
    atomic_add3(d_v + 3 * (size_t)i1, ge1);
    atomic_add3(d_v + 3 * (size_t)i2, ge2);
    atomic_add3(d_v + 3 * (size_t)i0, (ge1 + ge2) * -1.0f);
}

// ---- prepare_shading_normal -------------------------------------------------------------------------
// inputs are [B,H,W,3] or broadcast along any of B/H/W (stride 0), as c_src/tensor.h:20-92 allows
struct Bc { const float* p; long long sb, sh, sw; };
__device__ __forceinline__ V3 fetch(const Bc& t, int b, int y, int x) { return ld3(t.p + b * t.sb + y * t.sh + x * t.sw); }

constexpr float NORMAL_THRESHOLD = 0.1f;    // bsdf.py:13

__global__ __launch_bounds__(256) void shading_normal_fwd_kernel(Bc pos, Bc view_pos, Bc pert, Bc snrm, Bc stng, Bc gnrm, int B, int H, int W,
                                                                 int two_sided, int opengl, float* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)B * H * W) return;
    int b = (int)(i / ((size_t)H * W)), rem = (int)(i % ((size_t)H * W)), y = rem / W, x = rem % W;
    V3 sn = fnormalize(fetch(snrm, b, y, x)), st = fnormalize(fetch(stng, b, y, x));
    V3 vv = fnormalize(fetch(view_pos, b, y, x) - fetch(pos, b, y, x));
    V3 p = fetch(pert, b, y, x), gn = fetch(gnrm, b, y, x);
    V3 bt = fnormalize(cross(st, sn));
    float sgn = opengl ? -1.f : 1.f;
    V3 sh = st * p.x + bt * (sgn * p.y) + sn * fmaxf(p.z, 0.f);
    V3 shn = fnormalize(sh);
    if (two_sided) {
        float flip = dot(gn, vv) > 0.f ? 1.f : -1.f;
        shn = shn * flip;
        gn = gn * flip;
    }
    float t = fminf(fmaxf(dot(vv, shn) / NORMAL_THRESHOLD, 0.f), 1.f);
    st3(out + 3 * i, gn + (shn - gn) * t);
}

__global__ __launch_bounds__(256) void shading_normal_bwd_kernel(Bc pos, Bc view_pos, Bc pert, Bc snrm, Bc stng, Bc gnrm, int B, int H, int W,
                                                                 int two_sided, int opengl, const float* __restrict__ g_out,
                                                                 float* __restrict__ d_pos, float* __restrict__ d_view, float* __restrict__ d_pert,
                                                                 float* __restrict__ d_snrm, float* __restrict__ d_stng, float* __restrict__ d_gnrm) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)B * H * W) return;
    int b = (int)(i / ((size_t)H * W)), rem = (int)(i % ((size_t)H * W)), y = rem / W, x = rem % W;
    V3 snr = fetch(snrm, b, y, x), str_ = fetch(stng, b, y, x);
    V3 sn = fnormalize(snr), st = fnormalize(str_);
    V3 vd = fetch(view_pos, b, y, x) - fetch(pos, b, y, x);
    V3 vv = fnormalize(vd);
    V3 p = fetch(pert, b, y, x), gn0 = fetch(gnrm, b, y, x);
    V3 c = cross(st, sn);
    V3 bt = fnormalize(c);
    float sgn = opengl ? -1.f : 1.f;
    V3 sh = st * p.x + bt * (sgn * p.y) + sn * fmaxf(p.z, 0.f);
    V3 shn = fnormalize(sh);
    float flip = (two_sided && !(dot(gn0, vv) > 0.f)) ? -1.f : 1.f;
    V3 s2 = shn * flip, g2 = gn0 * flip;
    float dv = dot(vv, s2) / NORMAL_THRESHOLD;
    float t = fminf(fmaxf(dv, 0.f), 1.f);
    V3 go = ld3(g_out + 3 * i);
    V3 g_g2 = go * (1.f - t), g_s2 = go * t, g_vv = mk(0.f, 0.f, 0.f);
    float g_t = dot(go, s2 - g2);
    if (dv > 0.f && dv < 1.f) {
        float gd = g_t / NORMAL_THRESHOLD;
        g_vv = s2 * gd;
        g_s2 = g_s2 + vv * gd;
    }
    V3 g_sh = fnormalize_bwd(sh, g_s2 * flip);
    V3 g_st = g_sh * p.x, g_bt = g_sh * (sgn * p.y), g_sn = g_sh * fmaxf(p.z, 0.f);
    V3 g_p = mk(dot(g_sh, st), sgn * dot(g_sh, bt), p.z > 0.f ? dot(g_sh, sn) : 0.f);
    V3 g_c = fnormalize_bwd(c, g_bt);
    g_st = g_st + cross(sn, g_c);
    g_sn = g_sn + cross(g_c, st);
    V3 g_vd = fnormalize_bwd(vd, g_vv);
    st3(d_snrm + 3 * i, fnormalize_bwd(snr, g_sn));
    st3(d_stng + 3 * i, fnormalize_bwd(str_, g_st));
    st3(d_gnrm + 3 * i, g_g2 * flip);
    st3(d_pert + 3 * i, g_p);
    st3(d_view + 3 * i, g_vd);
    st3(d_pos + 3 * i, g_vd * -1.f);
}

// ---- image loss (loss.cu) -----------------------------------------------------------------------------
__device__ __forceinline__ float fwd_srgb(float x) { return x > 0.0031308f ? powf(fmaxf(x, 0.0031308f), 1.0f / 2.4f) * 1.055f - 0.055f : 12.92f * fmaxf(x, 0.0f); }
__device__ __forceinline__ float bwd_srgb(float x, float d_out) {
    if (x > 0.0031308f) return d_out * 0.439583f / powf(x, 0.583333f);
    if (x > 0.0f) return d_out * 12.92f;
    return 0.f;
}
__device__ __forceinline__ float clamp_hdr(float x) { return fminf(fmaxf(x, 0.f), 65535.f); }
__device__ __forceinline__ float loss_elem(int loss, float a, float t) {
    float d = a - t;
    if (loss == 1) return d * d;                                   // mse
    if (loss == 2) return fabsf(d) / (a + t + 0.01f);              // smape (loss.cu:85)
    if (loss == 3) return d * d / (a * a + t * t + 0.1f);          // relmse (loss.cu:73)
    return fabsf(d);                                               // l1
}
__device__ __forceinline__ void loss_elem_bwd(int loss, float a, float t, float go, float& ga, float& gt) {
    float d = a - t;
    if (loss == 1) { ga = go * 2.f * d; gt = -ga; }
    else if (loss == 2) {
        float den = t + a + 0.01f, s = d == 0.f ? 0.f : (d < 0.f ? -1.f : 1.f);
        ga = go * s * (2.f * t + 0.01f) / (den * den);
        gt = -go * s * (2.f * a + 0.01f) / (den * den);
    } else if (loss == 3) {
        float den = t * t + a * a + 0.1f;
        ga = go * 2.f * d * (t * (t + a) + 0.1f) / (den * den);
        gt = -go * 2.f * d * (a * (t + a) + 0.1f) / (den * den);
    } else {
        float s = d == 0.f ? 0.f : (d < 0.f ? -1.f : 1.f);
        ga = go * s;
        gt = -ga;
    }
}

// out[0] += sum over pixels of mean_c(loss)   (caller divides by the pixel count, ops.py:497)
__global__ __launch_bounds__(256) void image_loss_fwd_kernel(const float* __restrict__ img, const float* __restrict__ tgt, size_t npix, int loss, int tonemap,
                                                             float* __restrict__ out) {
    __shared__ float s4[4];
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (size_t)gridDim.x * 256) {
        float l = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float a = clamp_hdr(img[3 * i + c]), t = clamp_hdr(tgt[3 * i + c]);
            if (tonemap) { a = fwd_srgb(logf(a + 1.0f)); t = fwd_srgb(logf(t + 1.0f)); }
            l += loss_elem(loss, a, t);
        }
        acc += l / 3.0f;
    }
    float tot = block_sum(acc, s4);
    if (threadIdx.x == 0) atomicAdd(out, tot);
}
__global__ __launch_bounds__(256) void image_loss_bwd_kernel(const float* __restrict__ img, const float* __restrict__ tgt, size_t npix, int loss, int tonemap,
                                                             const float* __restrict__ g_scalar, float scale, float* __restrict__ d_img,
                                                             float* __restrict__ d_tgt) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    float go = g_scalar[0] * scale / 3.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float a0 = img[3 * i + c], t0 = tgt[3 * i + c];
        float a = clamp_hdr(a0), t = clamp_hdr(t0);
        float la = a, lt = t;
        if (tonemap) { la = logf(a + 1.0f); lt = logf(t + 1.0f); a = fwd_srgb(la); t = fwd_srgb(lt); }
        float ga, gt;
        loss_elem_bwd(loss, a, t, go, ga, gt);
        if (tonemap) {   // loss.cu:44-62: gradient only strictly inside (0, 65535)
            ga = (a0 > 0.f && a0 < 65535.f) ? bwd_srgb(la, ga) / (a0 + 1.0f) : 0.f;
            gt = (t0 > 0.f && t0 < 65535.f) ? bwd_srgb(lt, gt) / (t0 + 1.0f) : 0.f;
        }
        if (d_img) d_img[3 * i + c] = ga;
        if (d_tgt) d_tgt[3 * i + c] = gt;
    }
}

// ---- surface sampling (kaolin.ops.mesh.sample_points; hmsdf.py:714,750) --------------------------------------------------------
__global__ void face_areas_kernel(const float* __restrict__ v, const int64_t* __restrict__ f, int nf, float* __restrict__ area) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nf) return;
    V3 a = ld3(v + 3 * f[3 * (size_t)i]), b = ld3(v + 3 * f[3 * (size_t)i + 1]), c = ld3(v + 3 * f[3 * (size_t)i + 2]);
    V3 n = cross(b - a, c - a);
    area[i] = 0.5f * sqrtf(dot(n, n));
}
// p = (1 - sqrt(u)) a + sqrt(u) (1 - w) b + sqrt(u) w c for the picked face, (u, w) uniform in [0, 1)
__global__ void sample_faces_kernel(const float* __restrict__ v, const int64_t* __restrict__ f, const int64_t* __restrict__ pick,
                                    const float* __restrict__ uw, int n, float* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t t = pick[i];
    V3 a = ld3(v + 3 * f[3 * t]), b = ld3(v + 3 * f[3 * t + 1]), c = ld3(v + 3 * f[3 * t + 2]);
    float u = sqrtf(uw[2 * (size_t)i]), w = uw[2 * (size_t)i + 1];
    st3(out + 3 * (size_t)i, a * (1.0f - u) + b * (u * (1.0f - w)) + c * (u * w));
}

// The whole sampler in one call of three small kernels (round 5; the torch form -- area sum, weight fix-up, multinomial's normalise / scan / search, a second
// rand -- was 14 launches on the host-bound stretch right after the marching-tets read-back).
// (1) the face areas (face_areas_kernel), then ONE workgroup turns them into inclusive prefix sums cdf[nf] in place (double accumulation,
//     float storage);
// (2) one thread per sample: face = first i with cdf[i] > r0 * total  (zero-area rows -- degenerate faces, the zero padding of a face list at
//     its allocation bound -- have cdf[i] == cdf[i-1] and are never picked, as with Categorical(areas)), then the barycentric map above.
//     total == 0 (no face with an area): the sampler is ill-defined, as in the reference; picks i % nf, the caller discards the samples.
__global__ __launch_bounds__(1024) void sample_cdf_kernel(int nf, float* __restrict__ cdf) {
    // in: cdf[i] = area of face i (face_areas_kernel, all CUs); out: their inclusive prefix sums.  One workgroup: every thread owns a run of
    // consecutive faces (loaded in one burst), the 1024 run totals are scanned in the waves + one LDS hop.  (The first version computed the
    // areas here as well, twice, behind two dependent gathers each: 76 us on ONE CU for 18 k faces.)
    __shared__ double s_wave[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double carry = 0.0;
    constexpr int PER = 16;
    for (int base = 0; base < nf; base += 1024 * PER) {           // one segment up to 16 384 faces; larger meshes loop with a running carry
        const int nbs = nf - base < 1024 * PER ? nf - base : 1024 * PER;
        const int per = (nbs + 1023) / 1024;
        const int lo = base + tid * per, hi = base + nbs;
        float a[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) a[k] = (k < per && lo + k < hi) ? cdf[lo + k] : 0.f;
        double run = 0.0;
#pragma unroll
        for (int k = 0; k < PER; ++k) run += (double)a[k];
        double x = run;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            double y = __shfl_up(x, d);
            if (lane >= d) x += y;
        }
        if (lane == 63) s_wave[wave] = x;
        __syncthreads();
        double off = carry + (x - run), tot = 0.0;
        for (int w = 0; w < 16; ++w) {
            if (w < wave) off += s_wave[w];
            tot += s_wave[w];
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            off += (double)a[k];
            if (k < per && lo + k < hi) cdf[lo + k] = (float)off;
        }
        carry += tot;
        __syncthreads();
    }
}
__global__ void sample_surface_kernel(const float* __restrict__ v, const int64_t* __restrict__ f, const float* __restrict__ cdf, int nf,
                                      const float* __restrict__ rnd, int n, float* __restrict__ out, int64_t* __restrict__ pick) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float total = cdf[nf - 1];
    int64_t t;
    if (total > 0.f) {
        float r = rnd[3 * (size_t)i] * total;
        if (!(r < total)) r = nextafterf(total, 0.f);  // r0 * total may round up to total: the search below must stay below the last step of the cdf
        int lo = 0, hi = nf - 1;                        // invariant: cdf[hi] > r; answer in [lo, hi]
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] > r) hi = mid; else lo = mid + 1;
        }
        t = lo;
    } else {
        t = i % nf;
    }
    V3 a = ld3(v + 3 * f[3 * t]), b = ld3(v + 3 * f[3 * t + 1]), c = ld3(v + 3 * f[3 * t + 2]);
    float u = sqrtf(rnd[3 * (size_t)i + 1]), w = rnd[3 * (size_t)i + 2];
    st3(out + 3 * (size_t)i, a * (1.0f - u) + b * (u * (1.0f - w)) + c * (u * w));
    pick[i] = t;
}

// ---- gradient of "the first cin of cout channels" --------------------------------------------------------------------------------------
// shade() hands the first three of the texture MLP's six channels (kd) on as the shaded colour (render.py:120,169-170); autograd's slice node
// answered with a zero fill of the six-channel image plus a strided copy into it (17 + 42 us at 4 x 1024^2).  One pass instead: out[i] = (g[i], 0...).
__global__ __launch_bounds__(256) void channels_pad_kernel(const float* __restrict__ g, size_t n, int cin, int cout, float* __restrict__ out) {
    const size_t total = n * (size_t)cout;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < total; k += (size_t)gridDim.x * 256) {
        const size_t i = k / cout;
        const int c = (int)(k - i * cout);
        out[k] = c < cin ? g[i * cin + c] : 0.f;
    }
}

// ---- composite of the shaded layer against per-buffer backgrounds (render/render.py:375-382,430-449) --------------------------------
// Every buffer of the reference's single layer is [value channels, alpha = 1]; render_mesh lerps it against its background with
// weight coverage * alpha and antialiases each result separately.  Here all buffers are written, channel-concatenated, by one pass:
// covered pixel -> [src, 1], uncovered -> background (torch.lerp is exact at weights 0 and 1).  kind 0: zero background; 1: image
// background `bg` [Bbg][H][W][3] with alpha 0 ('shaded'); 2: constant 20 in every channel ('depth'); 3: alpha-only source
// ('msdf_image': lerp(0, 1, coverage * value) -> one channel coverage * value).
__global__ __launch_bounds__(256) void composite_fwd_kernel(CompArgs a, const float* __restrict__ rast, size_t npix, size_t hw, float* __restrict__ out) {
    D3H_DYN_SHARED(float, comp_lds);          // 256 * C floats
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < npix) comp_row(a, rast, i, hw, comp_lds + (size_t)threadIdx.x * a.C);
    block_store_rows(out, comp_lds, (size_t)blockIdx.x * 256, npix, a.C);
}
// d(src) = coverage ? d(out)[value channels] : 0, written densely [npix][nch] for every source with a gradient buffer
__global__ __launch_bounds__(256) void composite_bwd_kernel(CompArgs a, const float* __restrict__ rast, size_t npix, const float* __restrict__ g) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    const bool cov = rast[4 * i + 3] > 0.f;
    const float* gi = g + i * a.C;
    for (int k = 0; k < a.n; ++k) {
        const CompSrc& c = a.s[k];
        if (c.d) {
            float* d = c.d + i * c.nch;
            for (int j = 0; j < c.nch; ++j) d[j] = cov ? gi[j] : 0.f;
        }
        gi += c.kind == 3 ? 1 : c.nch + 1;
    }
}

// ---- fused per-pixel loss stack of tick_init / tick_split (hmsdf.py:835-839,895-898 / 969-975,1064-1068) ----------------------------
// One pass over the antialiased, channel-concatenated render output `st` [npix][C] (render_mesh's stacked image) and the two
// references.  Channel offsets: cs = 'shaded' (rgba), cg = 'geometric_normal' (xyz.), cm = 'msdf_image' (1 channel); < 0 = absent.
//   sums[0] += (shaded.a - ref.a)^2                                    mask MSE
//   sums[1] += mean_c loss(tonemap(shaded.rgb * ref.a), tonemap(ref.rgb * ref.a))     ru.image_loss (loss.cu)
//   sums[2] += |max(m, 0) [ref.a == 0]|      sums[3] += |min(m, 0) [ref.a == 1] - 1|   the two msdf L1 terms
//   sums[4] += |n^ - t^|^2                    sums[5] += cos(n^, t^)                    n^ = normalize(gn) * (1,-1,-1), t^ = normalize(nref)
//   sums[6] += mean_c(kd_grad.rgb) * kd_grad.a     sums[7] += sum_c ks_grad.c * ks_grad.a     sums[8] += sum_c normal_grad.c * normal_grad.a
//              the three terms of regularizer.material_smoothness_grad (render/regularizer.py:47-52); ckg / csg / cng = first channel of
//              'kd_grad' / 'ks_grad' / 'normal_grad' (rgba each) or < 0
// with torch's F.normalize (eps 1e-12) / F.cosine_similarity (eps 1e-8) clamping.  Optionally emits the two SSIM operands as
// NCHW planes (ssim_loss.py:33 is fed shaded.rgb * ref.a and ref.rgb * ref.a, permuted).
__device__ __forceinline__ float sgnf0(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }
struct PixLossCfg {
    int C, cs, cg, cm, nref_stride, loss, tonemap, H, W, ckg, csg, cng;
    int prep;                      // `masked` output: 0 = shaded.rgb * ref.a; 1 = ((2 * that - 1) - shift[c]) / scale[c]  (the LPIPS input map)
    float shift[3], scale[3];
};

__global__ __launch_bounds__(256) void pixel_losses_fwd_kernel(PixLossCfg k, const float* __restrict__ st, const float* __restrict__ cref,
                                                               const float* __restrict__ nref, size_t npix, float* __restrict__ sums,
                                                               float* __restrict__ ssim_a, float* __restrict__ ssim_b, float* __restrict__ masked,
                                                               int* __restrict__ ssim_occ, float* __restrict__ partials) {
    __shared__ float s4[4];
    D3H_DYN_SHARED(float, pl_rows);            // 256 * C floats (see block_load_rows)
    float acc[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const size_t hw = (size_t)k.H * k.W;
    for (size_t first = (size_t)blockIdx.x * 256; first < npix; first += (size_t)gridDim.x * 256) {     // (block-uniform trip count)
        const size_t i = first + threadIdx.x;
        // the per-pixel reference loads are issued before the barriers of the row staging, so all three streams are in flight together
        const bool live = i < npix;
        const float4 rf = live ? *(const float4*)(cref + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
        V3 nref_v = {0.f, 0.f, 0.f};
        if (live && k.cg >= 0 && nref) nref_v = ld3(nref + i * k.nref_stride);
        block_load_rows(pl_rows, st, first, npix, k.C);
        if (!live) continue;
        const float* px = pl_rows + threadIdx.x * k.C;
        const float rc[3] = {rf.x, rf.y, rf.z};
        if (k.cs >= 0) {
            float da = px[k.cs + 3] - rf.w;
            acc[0] += da * da;
            float l = 0.f;
            bool nz = false;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float a0 = px[k.cs + c] * rf.w, t0 = rc[c] * rf.w;
                if (ssim_a) {
                    size_t b = i / hw, o = i - b * hw;
                    ssim_a[(b * 3 + c) * hw + o] = a0;
                    ssim_b[(b * 3 + c) * hw + o] = t0;
                    nz = nz || a0 != 0.f || t0 != 0.f;
                }
                // the masked colour as an image of its own, channels-last (the LPIPS input of tick_split; lpips.py: 2 x - 1, ScalingLayer)
                if (masked) masked[3 * i + c] = k.prep ? ((2.0f * a0 - 1.0f) - k.shift[c]) / k.scale[c] : a0;
                if (k.loss >= 0) {
                    float a = clamp_hdr(a0), t = clamp_hdr(t0);
                    if (k.tonemap) { a = fwd_srgb(logf(a + 1.0f)); t = fwd_srgb(logf(t + 1.0f)); }
                    l += loss_elem(k.loss, a, t);
                }
            }
            acc[1] += l / 3.0f;
            if (ssim_occ && nz) {                              // the occupancy cell (32 rows x 64 columns) of this pixel: see SsimOcc
                const size_t b = i / hw, o = i - b * hw;
                const int y = (int)(o / (size_t)k.W), x = (int)(o - (size_t)y * k.W);
                ssim_occ[(b * ((k.H + 31) / 32) + y / 32) * ((k.W + 63) / 64) + x / 64] = 1;      // (every writer stores the same value)
            }
        }
        if (k.cm >= 0) {
            float m = px[k.cm];
            acc[2] += fabsf(fmaxf(m, 0.f) * (rf.w == 0.f ? 1.f : 0.f));
            acc[3] += fabsf(fminf(m, 0.f) * (rf.w == 1.f ? 1.f : 0.f) - 1.0f);
        }
        if (k.cg >= 0 && nref) {
            float no, nt, n1, n2;
            V3 o = normalize_eps(ld3(px + k.cg), 1e-12f, no);
            o.y = -o.y; o.z = -o.z;
            V3 t = normalize_eps(nref_v, 1e-12f, nt);
            V3 d = o - t;
            acc[4] += dot(d, d);
            V3 x1 = normalize_eps(o, 1e-8f, n1), x2 = normalize_eps(t, 1e-8f, n2);
            acc[5] += dot(x1, x2);
        }
        if (k.ckg >= 0) acc[6] += (px[k.ckg] + px[k.ckg + 1] + px[k.ckg + 2]) / 3.0f * px[k.ckg + 3];
        if (k.csg >= 0) acc[7] += (px[k.csg] + px[k.csg + 1] + px[k.csg + 2]) * px[k.csg + 3];
        if (k.cng >= 0) acc[8] += (px[k.cng] + px[k.cng + 1] + px[k.cng + 2]) * px[k.cng + 3];
    }
    // The nine sums: per-workgroup partials, added up in a fixed order by pixel_losses_finish_kernel (next launch on the stream).  The first
    // version ended every workgroup with nine float atomics into ONE 64-byte line (~3 ns each at the memory side), which capped the grid at
    // 1024 workgroups -- 16 dependent trips of the pixel loop each -- and left the pass latency-bound at 150-160 us for 370 MB.  (A single-launch
    // form -- ticket + __threadfence, the last workgroup adds -- was tried: the device-scope release of a workgroup that has just streamed its
    // share of 100 MB of planes writes its XCD's L2 back, 1024 / 4096 workgroups: 299 / 802 us.)
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        float tot = block_sum(acc[q], s4);
        if (threadIdx.x == 0) partials[(size_t)blockIdx.x * 9 + q] = tot;
    }
}

// one workgroup per sum
__global__ __launch_bounds__(256) void pixel_losses_finish_kernel(const float* __restrict__ partials, int nwg, float* __restrict__ sums) {
    __shared__ float s4[4];
    const int q = blockIdx.x;
    float v = 0.f;
    for (int j = threadIdx.x; j < nwg; j += 256) v += partials[(size_t)j * 9 + q];
    float tot = block_sum(v, s4);
    if (threadIdx.x == 0) sums[q] = tot;
}

// d_st [npix][C] is fully written (zeros in the channels these losses do not read); g[6] = dL/d(sums); d_ssim_a (NCHW planes or null) =
// dL/d(ssim operand a), chained through a = shaded.rgb * ref.a
__global__ __launch_bounds__(256) void pixel_losses_bwd_kernel(PixLossCfg k, const float* __restrict__ st, const float* __restrict__ cref,
                                                               const float* __restrict__ nref, size_t npix, const float* __restrict__ g,
                                                               const float* __restrict__ d_ssim_a, const float* __restrict__ d_masked,
                                                               float* __restrict__ d_st) {
    D3H_DYN_SHARED(float, pl_lds);             // 256 * C floats (see block_store_rows)
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < npix) {
    const size_t hw = (size_t)k.H * k.W;
    const float* px = st + i * k.C;
    float* dp = pl_lds + (size_t)threadIdx.x * k.C;
    for (int c = 0; c < k.C; ++c) dp[c] = 0.f;
    const float4 rf = *(const float4*)(cref + 4 * i);
    const float rc[3] = {rf.x, rf.y, rf.z};
    if (k.cs >= 0) {
        dp[k.cs + 3] = g[0] * 2.0f * (px[k.cs + 3] - rf.w);
        const float go = g[1] / 3.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float a0 = px[k.cs + c] * rf.w, t0 = rc[c] * rf.w;
            float ga = 0.f;
            if (k.loss >= 0) {
                float a = clamp_hdr(a0), t = clamp_hdr(t0);
                float la = a, lt = t;
                if (k.tonemap) { la = logf(a + 1.0f); lt = logf(t + 1.0f); a = fwd_srgb(la); t = fwd_srgb(lt); }
                float gt;
                loss_elem_bwd(k.loss, a, t, go, ga, gt);
                if (k.tonemap) ga = (a0 > 0.f && a0 < 65535.f) ? bwd_srgb(la, ga) / (a0 + 1.0f) : 0.f;
            }
            if (d_ssim_a) {
                size_t b = i / hw, o = i - b * hw;
                ga += d_ssim_a[(b * 3 + c) * hw + o];
            }
            if (d_masked) {
                const float gm = d_masked[3 * i + c];
                ga += k.prep ? (gm / k.scale[c]) * 2.0f : gm;
            }
            dp[k.cs + c] = ga * rf.w;
        }
    }
    if (k.cm >= 0) {
        float m = px[k.cm];
        float m0 = rf.w == 0.f ? 1.f : 0.f, m1 = rf.w == 1.f ? 1.f : 0.f;
        float u = fmaxf(m, 0.f) * m0, v = fminf(m, 0.f) * m1 - 1.0f;
        float gm = 0.f;
        if (m >= 0.f) gm += g[2] * sgnf0(u) * m0;          // clamp(min=0) passes the gradient where m >= 0
        if (m <= 0.f) gm += g[3] * sgnf0(v) * m1;          // clamp(max=0): where m <= 0
        dp[k.cm] = gm;
    }
    if (k.cg >= 0 && nref) {
        float no, nt, n1, n2;
        V3 y = normalize_eps(ld3(px + k.cg), 1e-12f, no);
        V3 o = y;
        o.y = -o.y; o.z = -o.z;
        V3 t = normalize_eps(ld3(nref + i * k.nref_stride), 1e-12f, nt);
        V3 x1 = normalize_eps(o, 1e-8f, n1), x2 = normalize_eps(t, 1e-8f, n2);
        V3 go = (o - t) * (2.0f * g[4]) + normalize_eps_bwd(x1, n1, 1e-8f, x2 * g[5]);
        go.y = -go.y; go.z = -go.z;
        V3 gv = normalize_eps_bwd(y, no, 1e-12f, go);
        st3(dp + k.cg, gv);
    }
    if (k.ckg >= 0) {
        const float a = px[k.ckg + 3], l3 = (px[k.ckg] + px[k.ckg + 1] + px[k.ckg + 2]) / 3.0f;
        dp[k.ckg] = dp[k.ckg + 1] = dp[k.ckg + 2] = g[6] * a / 3.0f;
        dp[k.ckg + 3] = g[6] * l3;
    }
    if (k.csg >= 0) {
        const float a = px[k.csg + 3];
        dp[k.csg] = dp[k.csg + 1] = dp[k.csg + 2] = g[7] * a;
        dp[k.csg + 3] = g[7] * (px[k.csg] + px[k.csg + 1] + px[k.csg + 2]);
    }
    if (k.cng >= 0) {
        const float a = px[k.cng + 3];
        dp[k.cng] = dp[k.cng + 1] = dp[k.cng + 2] = g[8] * a;
        dp[k.cng + 3] = g[8] * (px[k.cng] + px[k.cng + 1] + px[k.cng + 2]);
    }
    }
    block_store_rows(d_st, pl_lds, (size_t)blockIdx.x * 256, npix, k.C);
}

// ---- the mask and image terms of tick_seq (geometry/hmsdf.py:787-797,1110-1123) in one pass ----------------------------------------------
// From the stacked render (shaded rgb at channel cs, the antialiased coverage alpha at channel ca) and the per-pixel garment label of the
// mesh_id buffer (no gradient):  m_all = alpha,  m_cloth = label * alpha,  m_body = (1 - label) * alpha;
//   sums[k]     = sum_p (m_k - gt_k.a)^2                                     (all, cloth, body: the three mask MSEs)
//   sums[3 + k] = sum_p image_loss(shaded.rgb * m_k, gt_k.rgb) per pixel     (ru.image_loss: channel mean of the tone-mapped loss)
// As torch ops this is ~45 launches over the image per iteration (three products, three MSEs, three image losses with their mask products,
// the slice gradients of the stacked image); the seq-stage iteration is launch-bound.
struct SeqLossCfg { int C, cs, ca, loss, tonemap; };

__global__ __launch_bounds__(256) void seq_losses_fwd_kernel(SeqLossCfg k, const float* __restrict__ st, const float* __restrict__ label,
                                                             const float* __restrict__ gt_all, const float* __restrict__ gt_cloth,
                                                             const float* __restrict__ gt_body, size_t npix, float* __restrict__ sums) {
    __shared__ float s4[4];
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (size_t)gridDim.x * 256) {
        const float* px = st + i * k.C;
        const float alpha = px[k.ca], id = label[i];
        const float m[3] = {alpha, id * alpha, (1.0f - id) * alpha};
        const float4 gt[3] = {*(const float4*)(gt_all + 4 * i), *(const float4*)(gt_cloth + 4 * i), *(const float4*)(gt_body + 4 * i)};
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float dm = m[q] - gt[q].w;
            acc[q] += dm * dm;
            const float tc[3] = {gt[q].x, gt[q].y, gt[q].z};
            float l = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float a = clamp_hdr(px[k.cs + c] * m[q]), t = clamp_hdr(tc[c]);
                if (k.tonemap) { a = fwd_srgb(logf(a + 1.0f)); t = fwd_srgb(logf(t + 1.0f)); }
                l += loss_elem(k.loss, a, t);
            }
            acc[3 + q] += l / 3.0f;
        }
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        float tot = block_sum(acc[q], s4);
        if (threadIdx.x == 0 && tot != 0.f) atomicAdd(sums + q, tot);
    }
}

// d_st [npix][C] fully written: the shaded rgb channels, the alpha channel, zeros elsewhere.  g[6] = dL/d(sums).
__global__ __launch_bounds__(256) void seq_losses_bwd_kernel(SeqLossCfg k, const float* __restrict__ st, const float* __restrict__ label,
                                                             const float* __restrict__ gt_all, const float* __restrict__ gt_cloth,
                                                             const float* __restrict__ gt_body, size_t npix, const float* __restrict__ g,
                                                             float* __restrict__ d_st) {
    D3H_DYN_SHARED(float, sq_lds);             // 256 * C floats (see block_store_rows)
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < npix) {
        const float* px = st + i * k.C;
        float* dp = sq_lds + (size_t)threadIdx.x * k.C;
        for (int c = 0; c < k.C; ++c) dp[c] = 0.f;
        const float alpha = px[k.ca], id = label[i];
        const float mw[3] = {1.0f, id, 1.0f - id};                 // m_q = mw_q * alpha
        const float4 gt[3] = {*(const float4*)(gt_all + 4 * i), *(const float4*)(gt_cloth + 4 * i), *(const float4*)(gt_body + 4 * i)};
        float d_rgb[3] = {0.f, 0.f, 0.f}, d_alpha = 0.f;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float m = mw[q] * alpha;
            float dm = g[q] * 2.0f * (m - gt[q].w);
            const float tc[3] = {gt[q].x, gt[q].y, gt[q].z};
            const float go = g[3 + q] / 3.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float rgb = px[k.cs + c];
                const float a0 = rgb * m, t0 = tc[c];
                float a = clamp_hdr(a0), t = clamp_hdr(t0);
                float la = a, lt = t;
                if (k.tonemap) { la = logf(a + 1.0f); lt = logf(t + 1.0f); a = fwd_srgb(la); t = fwd_srgb(lt); }
                float ga, gtt;
                loss_elem_bwd(k.loss, a, t, go, ga, gtt);
                if (k.tonemap) ga = (a0 > 0.f && a0 < 65535.f) ? bwd_srgb(la, ga) / (a0 + 1.0f) : 0.f;      // as image_loss_bwd_kernel (loss.cu:44-62)
                d_rgb[c] += ga * m;
                dm += ga * rgb;
            }
            d_alpha += dm * mw[q];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) dp[k.cs + c] = d_rgb[c];
        dp[k.ca] = d_alpha;
    }
    block_store_rows(d_st, sq_lds, (size_t)blockIdx.x * 256, npix, k.C);
}

// ---- SSIM (ssim_loss.py:33-63), NCHW, separable 11-tap Gaussian with zero padding ----------------------------
struct G11 { float w[11]; };

constexpr int SR = 5;       // radius of the 11-tap window

// Sliding-window forward: one wave owns a 64-column x SW_ROWS-row band of a plane and walks down its rows.  Per input row: the 74
// pixels (64 + 2 x 5 halo) of both images go through a per-wave LDS row buffer (the next row's global loads are already in flight),
// every lane forms the 5 horizontal moments of its column (the same fmaf order as the tiled kernel), pushes them into an 11-deep
// register ring, and the vertical 11-tap sum of the ring gives the output row 5 rows back.  No (42+..)^2 halo tiles in LDS (1.2 KB
// per workgroup instead of 42 KB), the vertical pass never touches LDS, and the row loop keeps one global load per lane in flight.
// The first version (32x32 output tiles, (32+10)^2 halo and the horizontal pass of all five moments staged in 42 KB of LDS) spent 63 %
// of its wave cycles parked at its three block-wide phases (rocprofv3 SQ_WAIT_ANY) at 3 workgroups per CU: 229 us forward, 390 us
// backward at 12 x 1024^2; this one: 121 us and 154 us.  HBM traffic: 2 reads + 5 writes per pixel forward, 5 + 2 reads, 2 writes backward.
constexpr int SW_ROWS = 32;       // rows of an occupancy cell (and the rows per wave of the kernels without occupancy cells)
// ---- occupancy cells (round 6) ---------------------------------------------------------------------------------------------------------
// The images of a loss are zero outside the subject (both are multiplied by the target's alpha: ~80-90 % of a frame).  `occ` [images][OH][OW]
// (OH = ceil(H / 32), OW = ceil(W / 64): one cell per band of one wave; int, non-zero = some pixel of the cell is non-zero in a or b; written by
// the producer of the planes, pixel_losses_fwd_kernel) lets a wave whose 3 x 3 cell neighbourhood is empty -- every input within its 5-pixel
// halo is zero -- skip its loads and its filter: the SSIM map is then the constant the formula gives at zero (1), the moment gradients too
// (written only if a band within two cells will read them back in the backward), and the backward of such a band is exactly zero (its own
// pixels are zero, and d/d mu1 vanishes wherever the local means vanish).  Same results as without `occ` (NULL = every band is computed).
struct SsimOcc { const int* occ; int div, OH, OW; };        // div: planes per occupancy image (3 colour planes share one)
// (the 25 cells of the 5 x 5 neighbourhood are read by 25 lanes in ONE load and combined with two ballots: as a loop of 25 dependent
// branch-and-load steps -- what the compiler made of the scalar form -- every wave, skipped or not, began with 25 serial memory latencies)
__device__ __forceinline__ void ssim_occ_near_far(const SsimOcc& o, int plane_idx, int cy, int cx, bool& near, bool& far) {
    near = far = true;
    if (!o.occ) return;
    const int lane = threadIdx.x & 63;
    const int dy = lane / 5 - 2, dx = lane % 5 - 2;          // lanes 0..24
    const int yy = cy + dy, xx = cx + dx;
    const bool in = lane < 25 && yy >= 0 && yy < o.OH && xx >= 0 && xx < o.OW;
    const int* g = o.occ + (size_t)(plane_idx / o.div) * o.OH * o.OW;
    const bool v = in ? g[yy * o.OW + xx] != 0 : false;
    far = __ballot(v) != 0ull;
    near = __ballot(v && dy >= -1 && dy <= 1 && dx >= -1 && dx <= 1) != 0ull;
}
// the SSIM map value and the moment gradients at one pixel from its five filtered moments (ssim_loss.py:47-63)
__device__ __forceinline__ float ssim_point(const float (&m)[5], float& g_mu1, float& g_mu2, float& g_s11, float& g_s22, float& g_s12) {
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    float mu1 = m[0], mu2 = m[1];
    float v11 = m[2] - mu1 * mu1, v22 = m[3] - mu2 * mu2, v12 = m[4] - mu1 * mu2;
    float A1 = 2.f * mu1 * mu2 + C1, A2 = 2.f * v12 + C2, B1 = mu1 * mu1 + mu2 * mu2 + C1, B2 = v11 + v22 + C2;
    float v = (A1 * A2) / (B1 * B2);
    float iB = 1.f / (B1 * B2);
    float dA1 = A2 * iB, dA2 = A1 * iB, dB1 = -v / B1, dB2 = -v / B2;
    g_s12 = 2.f * dA2; g_s11 = dB2; g_s22 = dB2;
    g_mu1 = dA1 * 2.f * mu2 + dB1 * 2.f * mu1 - g_s11 * 2.f * mu1 - g_s12 * mu2;
    g_mu2 = dA1 * 2.f * mu1 + dB1 * 2.f * mu2 - g_s22 * 2.f * mu2 - g_s12 * mu1;
    return v;
}
// NEED_B = false: the second image is a constant (the target of a loss): of the five moment-gradient planes only d/d mu1, d/d E[a a], d/d E[a b]
// (planes 0, 2, 4) are written -- the backward then reads and filters three planes instead of five
// (the row buffers are wave-private: the waves of a workgroup only meet at the final sum)
template <bool NEED_B, int ROWS>
__global__ __launch_bounds__(256) void ssim_fwd_slide_kernel(G11 c_g, const float* __restrict__ a, const float* __restrict__ b, int H, int W,
                                                             float* __restrict__ out, float* __restrict__ gmom /*[5][N][H][W] or null*/,
                                                             size_t n, SsimOcc oc) {
    __shared__ float rowbuf[4][2][80];
    __shared__ float s4[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int x0 = blockIdx.x * 64;
    const int yb = (blockIdx.y * 4 + wave) * ROWS;
    const size_t plane = (size_t)blockIdx.z * H * W;
    float* ra = rowbuf[wave][0];
    float* rb = rowbuf[wave][1];
    float val = 0.f;
    const int gx = x0 + lane;
    bool near, far;
    ssim_occ_near_far(oc, (int)blockIdx.z, yb / SW_ROWS, (int)blockIdx.x, near, far);
    if (!near) {                                               // (wave-uniform) every input of this band is zero
        if (yb < H && gx < W) {
            const float zero[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
            float g_mu1, g_mu2, g_s11, g_s22, g_s12;
            const float v = ssim_point(zero, g_mu1, g_mu2, g_s11, g_s22, g_s12);
            const int rows = (H - yb < ROWS) ? H - yb : ROWS;
            for (int r = 0; r < rows; ++r) val += v;
            if (gmom && far) {
                for (int r = 0; r < rows; ++r) {
                    size_t i = plane + (size_t)(yb + r) * W + gx;
                    gmom[i] = g_mu1; gmom[2 * n + i] = g_s11; gmom[4 * n + i] = g_s12;
                    if (NEED_B) { gmom[n + i] = g_mu2; gmom[3 * n + i] = g_s22; }
                }
            }
        }
    } else {
    float ring[5][11];
#pragma unroll
    for (int q = 0; q < 5; ++q)
#pragma unroll
        for (int j = 0; j < 11; ++j) ring[q][j] = 0.f;
    const int xa = x0 - SR + lane, xb = x0 + 64 - SR + lane;        // lanes 0..63 -> columns x0-5 .. x0+58; lanes 0..9 -> x0+59 .. x0+68
    float na0, nb0, na1, nb1;
    auto loadrow = [&](int y) {
        const bool iny = y >= 0 && y < H && yb < H;
        const size_t base = plane + (size_t)(iny ? y : 0) * W;
        const bool i0 = iny && xa >= 0 && xa < W, i1 = iny && lane < 2 * SR && xb < W;
        na0 = i0 ? a[base + xa] : 0.f; nb0 = i0 ? b[base + xa] : 0.f;      // zero padding (F.conv2d padding=5)
        na1 = i1 ? a[base + xb] : 0.f; nb1 = i1 ? b[base + xb] : 0.f;
    };
    loadrow(yb - SR);
    for (int r = 0; r < ROWS + 2 * SR; ++r) {
        const int y = yb - SR + r;
        ra[lane] = na0; rb[lane] = nb0;
        if (lane < 2 * SR) { ra[64 + lane] = na1; rb[64 + lane] = nb1; }
        if (r + 1 < ROWS + 2 * SR) loadrow(y + 1);           // (a second row in flight was tried: 584 -> 600 us for the four loss kernels)
        D3H_WAVE_SYNC();
        float m1 = 0.f, m2 = 0.f, s11 = 0.f, s22 = 0.f, s12 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            float w = c_g.w[k], p = ra[lane + k], q = rb[lane + k];
            m1 = fmaf(w, p, m1); m2 = fmaf(w, q, m2);
            s11 = fmaf(w, p * p, s11); s22 = fmaf(w, q * q, s22); s12 = fmaf(w, p * q, s12);
        }
        D3H_WAVE_SYNC();                                   // the row buffer is rewritten at the top of the next trip
#pragma unroll
        for (int q = 0; q < 5; ++q)
#pragma unroll
            for (int j = 0; j < 10; ++j) ring[q][j] = ring[q][j + 1];
        ring[0][10] = m1; ring[1][10] = m2; ring[2][10] = s11; ring[3][10] = s22; ring[4][10] = s12;
        const int gy = y - SR;                             // the output row whose 11 input rows are now in the ring
        if (r >= 2 * SR && gy < H && gx < W) {
            float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                float w = c_g.w[k];
#pragma unroll
                for (int q = 0; q < 5; ++q) m[q] = fmaf(w, ring[q][k], m[q]);
            }
            float g_mu1, g_mu2, g_s11, g_s22, g_s12;
            val += ssim_point(m, g_mu1, g_mu2, g_s11, g_s22, g_s12);
            if (gmom) {
                size_t i = plane + (size_t)gy * W + gx;
                gmom[i] = g_mu1; gmom[2 * n + i] = g_s11; gmom[4 * n + i] = g_s12;
                if (NEED_B) { gmom[n + i] = g_mu2; gmom[3 * n + i] = g_s22; }
            }
        }
    }
    }
    float tot = block_sum(val, s4);
    if (tid == 0) atomicAdd(out, tot);
}

// Sliding-window backward (same scheme as ssim_fwd_slide_kernel): the five moment-gradient planes are filtered with the (symmetric,
// separable) Gaussian -- per input row a horizontal 11-tap pass from a per-wave LDS row buffer, an 11-deep register ring, the vertical
// sum -- and chained to the two images at the output pixel.
// NEED_B = false: d_b is not wanted: planes 0, 2, 4 only (the forward wrote nothing else)
template <bool NEED_B, int ROWS>
__global__ __launch_bounds__(256) void ssim_bwd_slide_kernel(G11 c_g, const float* __restrict__ gmom, const float* __restrict__ a,
                                                             const float* __restrict__ b, int H, int W, const float* __restrict__ g_scalar,
                                                             float scale, float* __restrict__ d_a, float* __restrict__ d_b, size_t n, SsimOcc oc) {
    constexpr int NQ = NEED_B ? 5 : 3;                       // planes filtered; slot q reads plane PQ(q)
    auto PQ = [](int q) { return NEED_B ? q : 2 * q; };
    __shared__ float rowbuf[4][NQ][80];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int x0 = blockIdx.x * 64;
    const int yb = (blockIdx.y * 4 + wave) * ROWS;
    const size_t plane = (size_t)blockIdx.z * H * W;
    {
        bool near, far;
        ssim_occ_near_far(oc, (int)blockIdx.z, yb / SW_ROWS, (int)blockIdx.x, near, far);
        if (!near) {                                           // (wave-uniform; see SsimOcc) the gradient of this band is exactly zero
            const int gx0 = x0 + lane;
            const int rows = (H - yb < ROWS) ? H - yb : ROWS;
            if (yb < H && (W & 3) == 0 && x0 + 64 <= W) {      // 16-byte stores, four rows per instruction (one 4-byte store per lane and row: 67 us for 50 MB)
                const int rr = lane >> 4, c4 = (lane & 15) * 4;
                for (int r = rr; r < rows; r += 4) {
                    size_t i = plane + (size_t)(yb + r) * W + x0 + c4;
                    if (d_a) *(float4*)(d_a + i) = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (NEED_B && d_b) *(float4*)(d_b + i) = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            } else if (yb < H && gx0 < W) {
                for (int r = 0; r < rows; ++r) {
                    size_t i = plane + (size_t)(yb + r) * W + gx0;
                    if (d_a) d_a[i] = 0.f;
                    if (NEED_B && d_b) d_b[i] = 0.f;
                }
            }
            return;
        }
    }
    float ring[NQ][11];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int j = 0; j < 11; ++j) ring[q][j] = 0.f;
    const float go = g_scalar[0] * scale;
    const int gx = x0 + lane;
    const int xa = x0 - SR + lane, xb = x0 + 64 - SR + lane;
    float n0[NQ], n1[NQ];
    auto loadrow = [&](int y) {
        const bool iny = y >= 0 && y < H && yb < H;
        const size_t base = plane + (size_t)(iny ? y : 0) * W;
        const bool i0 = iny && xa >= 0 && xa < W, i1 = iny && lane < 2 * SR && xb < W;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            n0[q] = i0 ? gmom[PQ(q) * n + base + xa] : 0.f;
            n1[q] = i1 ? gmom[PQ(q) * n + base + xb] : 0.f;
        }
    };
    loadrow(yb - SR);
    for (int r = 0; r < ROWS + 2 * SR; ++r) {
        const int y = yb - SR + r;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            rowbuf[wave][q][lane] = n0[q];
            if (lane < 2 * SR) rowbuf[wave][q][64 + lane] = n1[q];
        }
        if (r + 1 < ROWS + 2 * SR) loadrow(y + 1);
        D3H_WAVE_SYNC();
        float hq[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) hq[q] = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            float w = c_g.w[k];
#pragma unroll
            for (int q = 0; q < NQ; ++q) hq[q] = fmaf(w, rowbuf[wave][q][lane + k], hq[q]);
        }
        D3H_WAVE_SYNC();
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
            for (int j = 0; j < 10; ++j) ring[q][j] = ring[q][j + 1];
            ring[q][10] = hq[q];
        }
        const int gy = y - SR;
        if (r >= 2 * SR && gy < H && gx < W) {
            float m[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) m[q] = 0.f;
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                float w = c_g.w[k];
#pragma unroll
                for (int q = 0; q < NQ; ++q) m[q] = fmaf(w, ring[q][k], m[q]);
            }
            size_t i = plane + (size_t)gy * W + gx;
            float p = a[i], q = b[i];
            if constexpr (NEED_B) {
                if (d_a) d_a[i] = go * (m[0] + 2.f * p * m[2] + q * m[4]);
                if (d_b) d_b[i] = go * (m[1] + 2.f * q * m[3] + p * m[4]);
            } else {
                if (d_a) d_a[i] = go * (m[0] + 2.f * p * m[1] + q * m[2]);
            }
        }
    }
}

// ---- SDF edge regulariser (hmsdf.py:162-170) ---------------------------------------------------------------
__device__ __forceinline__ float sgnf(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }
__device__ __forceinline__ float bce_logits(float x, float y) { return fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// out[0] += sum of both BCE terms over sign-changing edges, out[1] += their count
// marks (optional, [number of sdf values], zero on entry): 1 at both ends of every sign-changing edge -- the vertices that can receive a gradient
// from this term and from the surface extraction (d3h_sdf_mlp_bwd_prepare)
__global__ __launch_bounds__(256) void sdf_reg_fwd_kernel(const float* __restrict__ sdf, const int* __restrict__ edges, int ne, float* __restrict__ out,
                                                          float* __restrict__ marks) {
    __shared__ float s4[4];
    float acc = 0.f, cnt = 0.f;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < ne; e += gridDim.x * 256) {
        const int i0 = edges[2 * (size_t)e], i1 = edges[2 * (size_t)e + 1];
        float s0 = sdf[i0], s1 = sdf[i1];
        if (sgnf(s0) != sgnf(s1)) {
            acc += bce_logits(s0, s1 > 0.f ? 1.f : 0.f) + bce_logits(s1, s0 > 0.f ? 1.f : 0.f);
            cnt += 1.f;
            if (marks) { marks[i0] = 1.0f; marks[i1] = 1.0f; }          // (every writer stores the same value)
        }
    }
    float ta = block_sum(acc, s4), tc = block_sum(cnt, s4);
    if (threadIdx.x == 0) { atomicAdd(out, ta); atomicAdd(out + 1, tc); }
}
__global__ __launch_bounds__(256) void sdf_reg_bwd_kernel(const float* __restrict__ sdf, const int* __restrict__ edges, int ne, const float* __restrict__ sums,
                                                          const float* __restrict__ g_scalar, float* __restrict__ d_sdf) {
    int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= ne) return;
    int i0 = edges[2 * (size_t)e], i1 = edges[2 * (size_t)e + 1];
    float s0 = sdf[i0], s1 = sdf[i1];
    if (sgnf(s0) == sgnf(s1)) return;
    float g = g_scalar[0] / sums[1];
    atomicAdd(&d_sdf[i0], g * (sigmoidf(s0) - (s1 > 0.f ? 1.f : 0.f)));
    atomicAdd(&d_sdf[i1], g * (sigmoidf(s1) - (s0 > 0.f ? 1.f : 0.f)));
}

// ---- xfm_points (render/renderutils/ops.py:518-537; c_src/mesh.cu xfm_fwd/bwd): out[b][i] = M[b] [p; w] ----------------------------
// The matmul formulation is a [n,4]x[4,4] GEMM per frame, which the BLAS library runs at ~100 us for 3 10^4 points; this is one
// thread per point.  pts may be a broadcast batch of 1 (pts_bstride = 0): its gradient then sums over the frames.
__global__ __launch_bounds__(256) void xfm_points_fwd_kernel(const float* __restrict__ pts, size_t pts_bstride, const float* __restrict__ M, int n,
                                                             float w, float* __restrict__ out) {
    int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (i >= n) return;
    const float* p = pts + (size_t)b * pts_bstride + 3 * (size_t)i;
    const float* m = M + 16 * b;
    float x = p[0], y = p[1], z = p[2];
    float4 o;
    o.x = fmaf(m[0], x, fmaf(m[1], y, fmaf(m[2], z, m[3] * w)));
    o.y = fmaf(m[4], x, fmaf(m[5], y, fmaf(m[6], z, m[7] * w)));
    o.z = fmaf(m[8], x, fmaf(m[9], y, fmaf(m[10], z, m[11] * w)));
    o.w = fmaf(m[12], x, fmaf(m[13], y, fmaf(m[14], z, m[15] * w)));
    *(float4*)(out + 4 * ((size_t)b * n + i)) = o;
}
__global__ __launch_bounds__(256) void xfm_points_bwd_kernel(const float* __restrict__ g, const float* __restrict__ M, int n, int nb, int bcast,
                                                             float* __restrict__ d_pts) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float ax = 0.f, ay = 0.f, az = 0.f;
    for (int b = bcast ? 0 : blockIdx.y; b < (bcast ? nb : blockIdx.y + 1); ++b) {
        float4 q = *(const float4*)(g + 4 * ((size_t)b * n + i));
        const float* m = M + 16 * b;
        float dx = fmaf(q.x, m[0], fmaf(q.y, m[4], fmaf(q.z, m[8], q.w * m[12])));
        float dy = fmaf(q.x, m[1], fmaf(q.y, m[5], fmaf(q.z, m[9], q.w * m[13])));
        float dz = fmaf(q.x, m[2], fmaf(q.y, m[6], fmaf(q.z, m[10], q.w * m[14])));
        if (bcast) { ax += dx; ay += dy; az += dz; } else { ax = dx; ay = dy; az = dz; }
    }
    float* d = d_pts + 3 * ((bcast ? (size_t)0 : (size_t)blockIdx.y * n) + i);
    d[0] = ax; d[1] = ay; d[2] = az;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
static inline int nb256(size_t n) { return (int)((n + 255) / 256); }

// v: [nb][nv][3] (nb vertex sets sharing the face list f); vn_raw: [nb][nv][3] scratch kept for the backward; vn: [nb][nv][3]
extern "C" int d3h_auto_normals_fwd(const float* v, int nb, int nv, const int* f, int nf, float* vn_raw, float* vn, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (nv <= 0 || nb <= 0) return D3H_OK;
    (void)hipMemsetAsync(vn_raw, 0, sizeof(float) * 3 * (size_t)nv * nb, s);
    if (nf > 0) hipLaunchKernelGGL(face_cross_scatter_kernel, dim3(nb256(nf), nb), dim3(256), 0, s, v, f, nf, vn_raw, (size_t)nv * 3);
    hipLaunchKernelGGL(vnormal_finish_kernel, dim3(nb256((size_t)nv * nb)), dim3(256), 0, s, vn_raw, nv * nb, vn);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}
// d_v accumulated (caller zero-fills); g_raw: [nb][nv][3] scratch
extern "C" int d3h_auto_normals_bwd(const float* v, int nb, int nv, const int* f, int nf, const float* vn_raw, const float* g_vn, float* g_raw,
                                    float* d_v, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (nv <= 0 || nf <= 0 || nb <= 0) return D3H_OK;
    hipLaunchKernelGGL(vnormal_finish_bwd_kernel, dim3(nb256((size_t)nv * nb)), dim3(256), 0, s, vn_raw, g_vn, nv * nb, g_raw);
    hipLaunchKernelGGL(face_cross_scatter_bwd_kernel, dim3(nb256(nf), nb), dim3(256), 0, s, v, f, nf, g_raw, d_v, (size_t)nv * 3);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}
// pts [nbp][n][3] with nbp = nb or 1 (broadcast), M [nb][4][4] row-major, w = homogeneous coordinate (1: points, 0: vectors);
// out [nb][n][4]
extern "C" int d3h_xfm_points_fwd(const float* pts, int nbp, const float* M, int nb, int n, float w, float* out, void* stream) {
    if (nb <= 0 || n < 0 || (nbp != nb && nbp != 1)) return D3H_ERR_ARG;
    if (n == 0) return D3H_OK;
    hipLaunchKernelGGL(xfm_points_fwd_kernel, dim3(nb256(n), nb), dim3(256), 0, (hipStream_t)stream, pts, nbp == 1 ? (size_t)0 : (size_t)n * 3, M, n, w,
                       out);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}
// g [nb][n][4] -> d_pts [nbp][n][3] overwritten (summed over the frames when nbp == 1 < nb)
extern "C" int d3h_xfm_points_bwd(const float* g, int nbp, const float* M, int nb, int n, float* d_pts, void* stream) {
    if (nb <= 0 || n < 0 || (nbp != nb && nbp != 1)) return D3H_ERR_ARG;
    if (n == 0) return D3H_OK;
    const int bcast = (nbp == 1 && nb > 1) ? 1 : 0;
    hipLaunchKernelGGL(xfm_points_bwd_kernel, dim3(nb256(n), bcast ? 1 : nb), dim3(256), 0, (hipStream_t)stream, g, M, n, nb, bcast, d_pts);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}
// v: [nb][nv][3]; fn: [nb][nf][3]
extern "C" int d3h_face_normals_fwd(const float* v, int nb, int nv, const int* f, int nf, float* fn, void* stream) {
    if (nf <= 0 || nb <= 0) return D3H_OK;
    hipLaunchKernelGGL(face_normals_kernel, dim3(nb256(nf), nb), dim3(256), 0, (hipStream_t)stream, v, f, nf, fn, (size_t)nv * 3);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}
// d_v accumulated (caller zero-fills)
extern "C" int d3h_face_normals_bwd(const float* v, int nb, int nv, const int* f, int nf, const float* g_fn, float* d_v, void* stream) {
    if (nf <= 0 || nb <= 0) return D3H_OK;
    hipLaunchKernelGGL(face_normals_bwd_kernel, dim3(nb256(nf), nb), dim3(256), 0, (hipStream_t)stream, v, f, nf, g_fn, d_v, (size_t)nv * 3);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}

// strides[6][3]: element strides (b, h, w) of pos, view_pos, perturbed_nrm, smooth_nrm, smooth_tng, geom_nrm (0 = broadcast)
extern "C" int d3h_shading_normal_fwd(const float* pos, const float* view_pos, const float* pert, const float* snrm, const float* stng,
                                      const float* gnrm, const int64_t* strides, int B, int H, int W, int two_sided, int opengl, float* out,
                                      void* stream) {
    const float* ptr[6] = {pos, view_pos, pert, snrm, stng, gnrm};
    Bc t[6];
    for (int k = 0; k < 6; ++k) { t[k].p = ptr[k]; t[k].sb = strides[3 * k]; t[k].sh = strides[3 * k + 1]; t[k].sw = strides[3 * k + 2]; }
    size_t n = (size_t)B * H * W;
    if (n == 0) return D3H_OK;
    hipLaunchKernelGGL(shading_normal_fwd_kernel, dim3(nb256(n)), dim3(256), 0, (hipStream_t)stream, t[0], t[1], t[2], t[3], t[4], t[5], B, H, W,
                       two_sided, opengl, out);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}
// all six gradients are full resolution [B][H][W][3] (the wrapper reduces broadcast inputs, as autograd does for the reference)
extern "C" int d3h_shading_normal_bwd(const float* pos, const float* view_pos, const float* pert, const float* snrm, const float* stng,
                                      const float* gnrm, const int64_t* strides, int B, int H, int W, int two_sided, int opengl,
                                      const float* g_out, float* d_pos, float* d_view, float* d_pert, float* d_snrm, float* d_stng, float* d_gnrm,
                                      void* stream) {
    const float* ptr[6] = {pos, view_pos, pert, snrm, stng, gnrm};
    Bc t[6];
    for (int k = 0; k < 6; ++k) { t[k].p = ptr[k]; t[k].sb = strides[3 * k]; t[k].sh = strides[3 * k + 1]; t[k].sw = strides[3 * k + 2]; }
    size_t n = (size_t)B * H * W;
    if (n == 0) return D3H_OK;
    hipLaunchKernelGGL(shading_normal_bwd_kernel, dim3(nb256(n)), dim3(256), 0, (hipStream_t)stream, t[0], t[1], t[2], t[3], t[4], t[5], B, H, W,
                       two_sided, opengl, g_out, d_pos, d_view, d_pert, d_snrm, d_stng, d_gnrm);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}

// loss: 0 l1, 1 mse, 2 smape, 3 relmse; tonemap: 0 none, 1 log_srgb.  out[0] (zeroed here) = sum over pixels of the channel-mean loss
extern "C" int d3h_image_loss_fwd(const float* img, const float* tgt, int64_t npix, int loss, int tonemap, float* out, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    (void)hipMemsetAsync(out, 0, sizeof(float), s);
    if (npix > 0) hipLaunchKernelGGL(image_loss_fwd_kernel, dim3(d3h_grid(npix, 256)), dim3(256), 0, s, img, tgt, (size_t)npix, loss, tonemap, out);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}
// g_scalar: device scalar dL/d(out * scale)
extern "C" int d3h_image_loss_bwd(const float* img, const float* tgt, int64_t npix, int loss, int tonemap, const float* g_scalar, float scale,
                                  float* d_img, float* d_tgt, void* stream) {
    if (npix <= 0) return D3H_OK;
    hipLaunchKernelGGL(image_loss_bwd_kernel, dim3(nb256(npix)), dim3(256), 0, (hipStream_t)stream, img, tgt, (size_t)npix, loss, tonemap, g_scalar,
                       scale, d_img, d_tgt);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}

// area[nf] of the triangles f[nf][3] (int64 vertex ids, as Mesh.t_pos_idx) over v[.][3]
extern "C" int d3h_face_areas(const float* v, const int64_t* f, int nf, float* area, void* stream) {
    if (nf < 0 || (nf > 0 && (!v || !f || !area))) return D3H_ERR_ARG;
    if (nf > 0) hipLaunchKernelGGL(face_areas_kernel, dim3(nb256(nf)), dim3(256), 0, (hipStream_t)stream, v, f, nf, area);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}
// out[n][3]: one point per (pick[i], uw[i]) -- the barycentric map of kaolin's sample_points
extern "C" int d3h_sample_faces(const float* v, const int64_t* f, const int64_t* pick, const float* uw, int n, float* out, void* stream) {
    if (n < 0 || (n > 0 && (!v || !f || !pick || !uw || !out))) return D3H_ERR_ARG;
    if (n > 0) hipLaunchKernelGGL(sample_faces_kernel, dim3(nb256(n)), dim3(256), 0, (hipStream_t)stream, v, f, pick, uw, n, out);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}

// kaolin.ops.mesh.sample_points in one call: n area-weighted surface samples of the mesh (v, f[nf][3] int64) from rnd[n][3] uniform numbers in
// [0, 1) (face pick, sqrt-barycentric u, w) -> out[n][3], pick[n] (int64 face ids); cdf[nf]: scratch (the inclusive area prefix sums)
extern "C" int d3h_sample_surface(const float* v, const int64_t* f, int nf, const float* rnd, int n, float* cdf, float* out, int64_t* pick,
                                  void* stream) {
    if (n < 0 || nf < 0 || (n > 0 && (nf == 0 || !v || !f || !rnd || !cdf || !out || !pick))) return D3H_ERR_ARG;
    if (n > 0) {
        hipLaunchKernelGGL(face_areas_kernel, dim3(nb256(nf)), dim3(256), 0, (hipStream_t)stream, v, f, nf, cdf);
        hipLaunchKernelGGL(sample_cdf_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, nf, cdf);
        hipLaunchKernelGGL(sample_surface_kernel, dim3(nb256(n)), dim3(256), 0, (hipStream_t)stream, v, f, cdf, nf, rnd, n, out, pick);
    }
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}

// out [n][cout] = (g [n][cin], zeros): the gradient of x[..., :cin] with respect to x [n][cout]
extern "C" int d3h_channels_pad(const float* g, long long n, int cin, int cout, float* out, void* stream) {
    if (n < 0 || cin <= 0 || cout < cin || (n > 0 && (!g || !out))) return D3H_ERR_ARG;
    if (n > 0) {
        const long long blocks = (n * cout + 255) / 256;
        hipLaunchKernelGGL(channels_pad_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, (hipStream_t)stream, g, (size_t)n, cin, cout, out);
    }
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}

// Composite nsrc layer buffers against their backgrounds into out [B*H*W][C], C = sum(nch + 1) (kind 3: 1).  src[k]: first value channel
// of buffer k at pixel 0, `stride[k]` floats between pixels (so slices of wider tensors need no copy); rast: [B][H][W][4] (coverage =
// triangle id > 0); bg[k]: [Bbg][H][W][3] for kind 1 (bg_batched[k] = Bbg > 1).  Host arrays are read before the call returns.
extern "C" int d3h_composite_fwd(int nsrc, const float* const* src, const int* stride, const int* nch, const int* kind, const float* const* bg,
                                 const int* bg_batched, const float* rast, int B, int H, int W, float* out, void* stream) {
    CompArgs a;
    int rc = comp_args(a, nsrc, src, nullptr, stride, nch, kind, bg, bg_batched, true);
    if (rc != D3H_OK || !src || !rast || !out || B < 0 || H <= 0 || W <= 0) return D3H_ERR_ARG;
    for (int k = 0; k < nsrc; ++k) if (!src[k]) return D3H_ERR_ARG;
    size_t npix = (size_t)B * H * W;
    const int kt_ = d3h_ktime_begin(D3H_KT_COMPOSITE_FWD, (long long)(npix * a.C), (hipStream_t)(stream));
    if (npix > 0) hipLaunchKernelGGL(composite_fwd_kernel, dim3(nb256(npix)), dim3(256), (size_t)256 * a.C * sizeof(float), (hipStream_t)stream, a, rast, npix,
                                     (size_t)H * W, out);
    d3h_ktime_end(kt_, (hipStream_t)(stream));
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}
// g: d(out) [B*H*W][C]; dsrc[k]: dense [B*H*W][nch[k]] gradient of buffer k's value channels, or NULL to skip it
extern "C" int d3h_composite_bwd(int nsrc, float* const* dsrc, const int* nch, const int* kind, const float* rast, int B, int H, int W,
                                 const float* g, void* stream) {
    CompArgs a;
    int rc = comp_args(a, nsrc, nullptr, dsrc, nch /* strides unused */, nch, kind, nullptr, nullptr, false);
    if (rc != D3H_OK || !dsrc || !rast || !g || B < 0 || H <= 0 || W <= 0) return D3H_ERR_ARG;
    size_t npix = (size_t)B * H * W;
    const int kt_ = d3h_ktime_begin(D3H_KT_COMPOSITE_BWD, (long long)(npix * a.C), (hipStream_t)(stream));
    if (npix > 0) hipLaunchKernelGGL(composite_bwd_kernel, dim3(nb256(npix)), dim3(256), 0, (hipStream_t)stream, a, rast, npix, g);
    d3h_ktime_end(kt_, (hipStream_t)(stream));
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}

// Fused per-pixel losses (see pixel_losses_fwd_kernel).  st: [npix][C] with npix = B*H*W; cref: [npix][4]; nref: [npix][nref_stride] or
// NULL; loss < 0 skips the image-loss term; sums[9] is zeroed here and receives raw SUMS (the caller applies the mean factors);
// ssim_a / ssim_b: [B][3][H][W] outputs or NULL.  masked: [npix][3] output or NULL = shaded.rgb * ref.a, mapped by ((2 x - 1) - shift) / scale
// per channel when prep (HOST: shift[3], scale[3]) is given (third_parties/lpips/lpips.py: normalize + ScalingLayer).
extern "C" int d3h_pixel_losses_fwd(const float* st, int C, int cs, int cg, int cm, int ckg, int csg, int cng, const float* cref, const float* nref,
                                    int nref_stride, int B, int H, int W, int loss, int tonemap, float* sums, float* ssim_a, float* ssim_b,
                                    float* masked, const float* prep, int* ssim_occ, void* stream) {
    // ssim_occ: NULL, or [B][ceil(H / 32)][ceil(W / 64)] ints, ZERO on entry: set to 1 where a cell holds a non-zero pixel of the SSIM operands
    // (the occupancy cells of d3h_ssim_fwd / d3h_ssim_bwd)
    if (!st || !cref || !sums || C <= 0 || B < 0 || H <= 0 || W <= 0 || (ssim_a && !ssim_b) || (masked && cs < 0) || (ssim_occ && !ssim_a)) return D3H_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int kt_ = d3h_ktime_begin(D3H_KT_PIXLOSS_FWD, (long long)((size_t)B * H * W * C), (hipStream_t)(stream));
    size_t npix = (size_t)B * H * W;
    PixLossCfg k{C, cs, cg, cm, nref_stride, loss, tonemap, H, W, ckg, csg, cng, prep ? 1 : 0, {0.f, 0.f, 0.f}, {1.f, 1.f, 1.f}};
    if (prep)
        for (int c = 0; c < 3; ++c) { k.shift[c] = prep[c]; k.scale[c] = prep[3 + c]; }
    if (npix == 0) {
        (void)hipMemsetAsync(sums, 0, 9 * sizeof(float), s);
    } else {
        // workgroups: 2048 (eight trips of the pixel loop each at 4 x 1024^2; the sums no longer serialise them -- see the kernel).  Measured
        // at 4 x 1024^2 x 9 channels with the SSIM planes, kernel + finish: 1024 / 2048 / 4096 / 16384 workgroups 143 + 9 / 124 + 13 / 116 + 22 /
        // 125 + 77 us with a one-workgroup finish (now one workgroup per sum); the atomic version: 163.  D3H_PIXLOSS_WGS overrides (A/B)
        static const int env_wgs = [] { const char* e = getenv("D3H_PIXLOSS_WGS"); return e ? atoi(e) : 0; }();
        const size_t cap = env_wgs > 0 ? (size_t)env_wgs : 2048;
        const int pgrid = (int)((npix + 255) / 256 < cap ? (npix + 255) / 256 : cap);
        float* scratch = nullptr;              // [pgrid][9] partial sums
        if (hipMallocAsync((void**)&scratch, (size_t)pgrid * 9 * sizeof(float), s) != hipSuccess) return D3H_ERR_ARG;
        hipLaunchKernelGGL(pixel_losses_fwd_kernel, dim3(pgrid), dim3(256), (size_t)256 * C * sizeof(float), s, k, st, cref, nref, npix, sums, ssim_a, ssim_b,
                           masked, ssim_occ, scratch);
        hipLaunchKernelGGL(pixel_losses_finish_kernel, dim3(9), dim3(256), 0, s, (const float*)scratch, pgrid, sums);
        (void)hipFreeAsync(scratch, s);
    }
    d3h_ktime_end(kt_, (hipStream_t)(stream));
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}
// g[9]: device vector dL/d(sums); d_ssim_a: [B][3][H][W] or NULL; d_masked: [npix][3] or NULL, the gradient of the forward's `masked` (same
// prep); d_st [npix][C] is overwritten
extern "C" int d3h_pixel_losses_bwd(const float* st, int C, int cs, int cg, int cm, int ckg, int csg, int cng, const float* cref, const float* nref,
                                    int nref_stride, int B, int H, int W, int loss, int tonemap, const float* g, const float* d_ssim_a,
                                    const float* d_masked, const float* prep, float* d_st, void* stream) {
    if (!st || !cref || !g || !d_st || C <= 0 || B < 0 || H <= 0 || W <= 0 || (d_masked && cs < 0)) return D3H_ERR_ARG;
    size_t npix = (size_t)B * H * W;
    PixLossCfg k{C, cs, cg, cm, nref_stride, loss, tonemap, H, W, ckg, csg, cng, prep ? 1 : 0, {0.f, 0.f, 0.f}, {1.f, 1.f, 1.f}};
    if (prep)
        for (int c = 0; c < 3; ++c) { k.shift[c] = prep[c]; k.scale[c] = prep[3 + c]; }
    const int kt_ = d3h_ktime_begin(D3H_KT_PIXLOSS_BWD, (long long)(npix * C), (hipStream_t)(stream));
    if (npix > 0) hipLaunchKernelGGL(pixel_losses_bwd_kernel, dim3(nb256(npix)), dim3(256), (size_t)256 * C * sizeof(float), (hipStream_t)stream, k, st, cref,
                                     nref, npix, g, d_ssim_a, d_masked, d_st);
    d3h_ktime_end(kt_, (hipStream_t)(stream));
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}

// The mask and image terms of tick_seq (see seq_losses_fwd_kernel).  st: [npix][C], npix = B*H*W; cs: first channel of shaded.rgb; ca: the
// channel of the antialiased coverage (geometric_normal's alpha); label: [npix] garment label of the mesh_id buffer; gt_*: [npix][4].
// sums[6] (zeroed here) receives raw SUMS: (m_all - gt_all.a)^2, (m_cloth - ..)^2, (m_body - ..)^2, then the three per-pixel image losses.
extern "C" int d3h_seq_losses_fwd(const float* st, int C, int cs, int ca, const float* label, const float* gt_all, const float* gt_cloth,
                                  const float* gt_body, int64_t npix, int loss, int tonemap, float* sums, void* stream) {
    if (!st || !label || !gt_all || !gt_cloth || !gt_body || !sums || C <= 0 || cs < 0 || cs + 3 > C || ca < 0 || ca >= C || npix < 0 || loss < 0)
        return D3H_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    (void)hipMemsetAsync(sums, 0, 6 * sizeof(float), s);
    SeqLossCfg k{C, cs, ca, loss, tonemap};
    int grid = (int)((npix + 255) / 256 < 1024 ? (npix + 255) / 256 : 1024);      // as d3h_pixel_losses_fwd: six atomics per workgroup into one line
    if (npix > 0) hipLaunchKernelGGL(seq_losses_fwd_kernel, dim3(grid), dim3(256), 0, s, k, st, label, gt_all, gt_cloth, gt_body, (size_t)npix, sums);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}
// g[6]: device vector dL/d(sums); d_st [npix][C] is overwritten (zeros outside shaded.rgb and the coverage channel)
extern "C" int d3h_seq_losses_bwd(const float* st, int C, int cs, int ca, const float* label, const float* gt_all, const float* gt_cloth,
                                  const float* gt_body, int64_t npix, int loss, int tonemap, const float* g, float* d_st, void* stream) {
    if (!st || !label || !gt_all || !gt_cloth || !gt_body || !g || !d_st || C <= 0 || cs < 0 || cs + 3 > C || ca < 0 || ca >= C || npix < 0 || loss < 0)
        return D3H_ERR_ARG;
    SeqLossCfg k{C, cs, ca, loss, tonemap};
    if (npix > 0) hipLaunchKernelGGL(seq_losses_bwd_kernel, dim3(nb256((size_t)npix)), dim3(256), (size_t)256 * C * sizeof(float), (hipStream_t)stream, k, st, label,
                                     gt_all, gt_cloth, gt_body, (size_t)npix, g, d_st);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}

static G11 ssim_window() {
    G11 g;
    float sum = 0.f;
    for (int i = 0; i < 11; ++i) { g.w[i] = (float)exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5)); sum += g.w[i]; }   // ssim_loss.py:22-24
    for (int i = 0; i < 11; ++i) g.w[i] /= sum;
    return g;
}

static int ssim_rows(bool with_occ) {
    static const int env = [] { const char* e = getenv("D3H_SSIM_ROWS"); return e ? atoi(e) : 0; }();
    if (env == 8 || env == 16 || env == 32) return env;
    (void)with_occ;
    return 32;
}

// a, b: [N][H][W] planes (N = batch*channels); gmom: [5][N][H][W] (NULL when no backward is needed); tmp: unused, may be NULL;
// out[0] (zeroed here) = sum of the SSIM map
// need_b = 0: the caller will ask d3h_ssim_bwd for d_a only (b is a constant): planes 1 and 3 of gmom are left unwritten, and d3h_ssim_bwd must be
// called with need_b = 0 as well
// occ: NULL, or the occupancy cells [N / occ_div][ceil(H / 32)][ceil(W / 64)] (int; non-zero = the cell holds a non-zero pixel of a or b in one of
// the occ_div planes of that image) -- see SsimOcc: bands whose neighbourhood is empty are not computed, with identical results; the same
// cells must be given to d3h_ssim_bwd (the forward leaves the moment gradients of far-away empty bands unwritten)
extern "C" int d3h_ssim_fwd(const float* a, const float* b, int N, int H, int W, float* tmp, float* gmom, int need_b, float* out, const int* occ,
                            int occ_div, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (occ && (occ_div <= 0 || N % occ_div)) return D3H_ERR_ARG;
    // (measured at 4 x 1024^2 x 3 planes, 16 % of the frame occupied: forward 127 -> 103 us, backward 114 -> 58 us; with EVERY band skipped the
    // forward takes 22 us: the bands that are left run as a dependent chain of 42 row steps on CUs with few other waves to hide their loads behind)
    const SsimOcc oc{occ, occ ? occ_div : 1, d3h_cdiv(H, SW_ROWS), d3h_cdiv(W, 64)};
    G11 g = ssim_window();
    (void)hipMemsetAsync(out, 0, sizeof(float), s);
    size_t n = (size_t)N * H * W;
    if (n == 0) return D3H_OK;
    const int kt_ = d3h_ktime_begin(D3H_KT_SSIM_FWD, (long long)n, (hipStream_t)(stream));       // (after the early-out: a begun record must be ended)
    (void)tmp;      // scratch of the former two-pass version; the tiled kernel stages through LDS
    // rows per wave: 32 (1.3 x halo overhead).  Shorter bands for the few bands occupancy cells leave were tried and are slower -- 4 x 1024^2 x 3
    // planes, 16 % occupied, forward / backward: 32 rows 108 / 61 us, 8 rows 148 / 90 us (without cells: 127 / 114) -- D3H_SSIM_ROWS = 8 | 16 | 32
    const int rows = ssim_rows(occ != nullptr);
#define D3H_SSIM_FWD(NB, R) hipLaunchKernelGGL((ssim_fwd_slide_kernel<NB, R>), dim3(d3h_cdiv(W, 64), d3h_cdiv(H, 4 * R), N), dim3(256), 0, s, g, a, b, H, W, out, gmom, n, oc)
    if (need_b) { if (rows == 8) D3H_SSIM_FWD(true, 8); else if (rows == 16) D3H_SSIM_FWD(true, 16); else D3H_SSIM_FWD(true, 32); }
    else { if (rows == 8) D3H_SSIM_FWD(false, 8); else if (rows == 16) D3H_SSIM_FWD(false, 16); else D3H_SSIM_FWD(false, 32); }
#undef D3H_SSIM_FWD
    d3h_ktime_end(kt_, (hipStream_t)(stream));
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}
// need_b: as given to d3h_ssim_fwd (0: gmom holds planes 0, 2, 4 only and d_b must be NULL)
extern "C" int d3h_ssim_bwd(const float* a, const float* b, int N, int H, int W, const float* gmom, int need_b, float* tmp, const float* g_scalar, float scale,
                            float* d_a, float* d_b, const int* occ, int occ_div, void* stream) {
    if (!need_b && d_b) return D3H_ERR_ARG;
    if (occ && (occ_div <= 0 || N % occ_div)) return D3H_ERR_ARG;
    const SsimOcc oc{occ, occ ? occ_div : 1, d3h_cdiv(H, SW_ROWS), d3h_cdiv(W, 64)};
    hipStream_t s = (hipStream_t)stream;
    size_t n = (size_t)N * H * W;
    if (n == 0) return D3H_OK;
    G11 g = ssim_window();
    (void)tmp;
    const int kt_ = d3h_ktime_begin(D3H_KT_SSIM_BWD, (long long)(n), (hipStream_t)(stream));
    const int rows = ssim_rows(occ != nullptr);
#define D3H_SSIM_BWD(NB, R) hipLaunchKernelGGL((ssim_bwd_slide_kernel<NB, R>), dim3(d3h_cdiv(W, 64), d3h_cdiv(H, 4 * R), N), dim3(256), 0, s, g, gmom, a, b, H, W, g_scalar, scale, d_a, d_b, n, oc)
    if (need_b) { if (rows == 8) D3H_SSIM_BWD(true, 8); else if (rows == 16) D3H_SSIM_BWD(true, 16); else D3H_SSIM_BWD(true, 32); }
    else { if (rows == 8) D3H_SSIM_BWD(false, 8); else if (rows == 16) D3H_SSIM_BWD(false, 16); else D3H_SSIM_BWD(false, 32); }
#undef D3H_SSIM_BWD
    d3h_ktime_end(kt_, (hipStream_t)(stream));
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}

// sums[0] = sum of both BCE terms, sums[1] = number of sign-changing edges (zeroed here); marks: NULL, or one float per sdf value, zero on entry:
// set to 1 at both ends of every sign-changing edge (the points d3h_sdf_mlp_bwd_prepare wants)
extern "C" int d3h_sdf_reg_fwd(const float* sdf, const int* edges, int ne, float* sums, float* marks, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    (void)hipMemsetAsync(sums, 0, 2 * sizeof(float), s);
    // 512 workgroups (two same-line atomics each): 2048 took 58 us for 1.8 10^6 edges, 512: 26 us
    if (ne > 0) hipLaunchKernelGGL(sdf_reg_fwd_kernel, dim3(nb256(ne) < 512 ? nb256(ne) : 512), dim3(256), 0, s, sdf, edges, ne, sums, marks);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}
// d_sdf accumulated (caller zero-fills)
extern "C" int d3h_sdf_reg_bwd(const float* sdf, const int* edges, int ne, const float* sums, const float* g_scalar, float* d_sdf, void* stream) {
    if (ne <= 0) return D3H_OK;
    hipLaunchKernelGGL(sdf_reg_bwd_kernel, dim3(nb256(ne)), dim3(256), 0, (hipStream_t)stream, sdf, edges, ne, sums, g_scalar, d_sdf);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}
