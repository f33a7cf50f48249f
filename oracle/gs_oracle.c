/*
 * gs_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C (CPU) restatement of the reference's differentiable Gaussian-splat
 * rasterizer, used only by tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py as the checker / reported baseline.  The
 * shipped path (frosting_amd/csrc, HIP) never links, imports or calls this.
 *
 * It follows, stage by stage, the reference at
 *   DGR = /root/reference/gaussian_splatting/submodules/diff-gaussian-rasterization
 *   DGR/cuda_rasterizer/auxiliary.h:41-77,139-164   ndc2Pix, getRect, transforms, in_frustum
 *   DGR/cuda_rasterizer/forward.cu:20-71            SH -> RGB
 *   DGR/cuda_rasterizer/forward.cu:74-113           EWA cov2D
 *   DGR/cuda_rasterizer/forward.cu:118-152          cov3D from scale / quaternion
 *   DGR/cuda_rasterizer/forward.cu:155-256          per-Gaussian preprocess
 *   DGR/cuda_rasterizer/rasterizer_impl.cu:70-138   key build, tile ranges
 *   DGR/cuda_rasterizer/rasterizer_impl.cu:277-308  inclusive scan, stable (tile,depth) sort
 *   DGR/cuda_rasterizer/forward.cu:261-374          front-to-back blend
 *   DGR/cuda_rasterizer/backward.cu:399-557         blend backward
 *   DGR/cuda_rasterizer/backward.cu:144-274         cov2D backward
 *   DGR/cuda_rasterizer/backward.cu:20-139,278-396  SH / cov3D / projection backward
 *
 * Arithmetic contract: IEEE binary32, one rounding per written operation, in
 * the reference's source association order (compile with -ffp-contract=off;
 * the Makefile does).  With that contract the preprocess stage is bit-identical
 * to the reference built with the same contraction mode (oracle/_ref
 * "exact" build) -- this is what pins radii / tiles_touched / sort keys.  The
 * blend stages differ from a GPU run only through libm's expf.
 *
 * Pinning: see oracle/README.md -- checked against tests/golden/ fixtures that
 * were produced by the reference itself (oracle/_ref) on an MI355X.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TILE 16
#define TILE_PIX 256

static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

int gso_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void gso_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* float -> int with the GPU's conversion semantics (v_cvt_i32_f32: NaN -> 0,
 * saturating), so that out-of-range screen coordinates clamp the same way. */
static inline int f2i(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)v;
}
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* auxiliary.h:58-77 (matrices are the row-vector convention, read column-major) */
static inline void xform4x3(const float* p, const float* m, float* o)
{
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static inline void xform4x4(const float* p, const float* m, float* o)
{
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

/* auxiliary.h:41-44 -- evaluated in double, rounded once to float */
static inline float ndc2pix(float v, int S) { return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5); }

/* auxiliary.h:46-56 */
static inline void get_rect(float px, float py, int max_radius, int gx, int gy, int* rmin, int* rmax)
{
    rmin[0] = imin(gx, imax(0, f2i((px - (float)max_radius) / (float)TILE)));
    rmin[1] = imin(gy, imax(0, f2i((py - (float)max_radius) / (float)TILE)));
    rmax[0] = imin(gx, imax(0, f2i((px + (float)max_radius + (float)TILE - 1.0f) / (float)TILE)));
    rmax[1] = imin(gy, imax(0, f2i((py + (float)max_radius + (float)TILE - 1.0f) / (float)TILE)));
}

/* forward.cu:118-152.  M[c][r] = s_r * R[c][r];  Sigma[c][r] = sum_k M[r][k]*M[c][k] */
static void cov3d_from_scale_rot(const float* scale, float mod, const float* q, float* cov3D)
{
    float s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
    float r = q[0], x = q[1], y = q[2], z = q[3];
    float R[3][3]; /* R[c][r], glm column-major */
    R[0][0] = 1.f - 2.f * (y * y + z * z); R[0][1] = 2.f * (x * y - r * z); R[0][2] = 2.f * (x * z + r * y);
    R[1][0] = 2.f * (x * y + r * z); R[1][1] = 1.f - 2.f * (x * x + z * z); R[1][2] = 2.f * (y * z - r * x);
    R[2][0] = 2.f * (x * z - r * y); R[2][1] = 2.f * (y * z + r * x); R[2][2] = 1.f - 2.f * (x * x + y * y);
    float M[3][3];
    for (int c = 0; c < 3; c++)
        for (int rr = 0; rr < 3; rr++) M[c][rr] = s[rr] * R[c][rr];
#define SIG(c, rr) (M[rr][0] * M[c][0] + M[rr][1] * M[c][1] + M[rr][2] * M[c][2])
    cov3D[0] = SIG(0, 0); cov3D[1] = SIG(0, 1); cov3D[2] = SIG(0, 2);
    cov3D[3] = SIG(1, 1); cov3D[4] = SIG(1, 2); cov3D[5] = SIG(2, 2);
#undef SIG
}

/* Shared by forward.cu:74-113 and backward.cu:144-196: view-space point with the
 * 1.3*tanfov clamp, T = W*J (T[c][r]; column 2 is zero) */
typedef struct { float t[3]; float T[2][3]; float xmul, ymul; } ewa_t;
static void ewa_setup(const float* mean, float fx, float fy, float tan_fovx, float tan_fovy, const float* vm, ewa_t* e)
{
    float t[3];
    xform4x3(mean, vm, t);
    const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
    const float txtz = t[0] / t[2], tytz = t[1] / t[2];
    t[0] = fminf(limx, fmaxf(-limx, txtz)) * t[2];
    t[1] = fminf(limy, fmaxf(-limy, tytz)) * t[2];
    e->xmul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
    e->ymul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
    const float J00 = fx / t[2], J02 = -(fx * t[0]) / (t[2] * t[2]);
    const float J11 = fy / t[2], J12 = -(fy * t[1]) / (t[2] * t[2]);
    /* W[k][r] = vm[4*r + k] */
    for (int rr = 0; rr < 3; rr++) {
        float W0 = vm[4 * rr + 0], W1 = vm[4 * rr + 1], W2 = vm[4 * rr + 2];
        e->T[0][rr] = W0 * J00 + W1 * 0.0f + W2 * J02;
        e->T[1][rr] = W0 * 0.0f + W1 * J11 + W2 * J12;
    }
    e->t[0] = t[0]; e->t[1] = t[1]; e->t[2] = t[2];
}
/* cov = T^t * Vrk^t * T, entries (0,0), (0,1), (1,1), before the +0.3 */
static void ewa_cov2d(const ewa_t* e, const float* c3, float* a, float* b, float* c)
{
    const float V[3][3] = {{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}};
    float X[3][2]; /* X[k][r] = sum_j T[r][j] * V[j][k] */
    for (int k = 0; k < 3; k++)
        for (int rr = 0; rr < 2; rr++) X[k][rr] = e->T[rr][0] * V[0][k] + e->T[rr][1] * V[1][k] + e->T[rr][2] * V[2][k];
    *a = X[0][0] * e->T[0][0] + X[1][0] * e->T[0][1] + X[2][0] * e->T[0][2];
    *b = X[0][1] * e->T[0][0] + X[1][1] * e->T[0][1] + X[2][1] * e->T[0][2];
    *c = X[0][1] * e->T[1][0] + X[1][1] * e->T[1][1] + X[2][1] * e->T[1][2];
}

/* forward.cu:20-71.  sh is [M][3] for this Gaussian. */
static void sh_to_rgb(int deg, const float* pos, const float* campos, const float* sh, float* rgb, uint8_t* clamped)
{
    float d[3] = {pos[0] - campos[0], pos[1] - campos[1], pos[2] - campos[2]};
    float len = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    float x = d[0] / len, y = d[1] / len, z = d[2] / len;
    for (int ch = 0; ch < 3; ch++) {
#define S(i) sh[(i) * 3 + ch]
        float res = SH_C0 * S(0);
        if (deg > 0) {
            res = res - SH_C1 * y * S(1) + SH_C1 * z * S(2) - SH_C1 * x * S(3);
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                res = res + SH_C2[0] * xy * S(4) + SH_C2[1] * yz * S(5) + SH_C2[2] * (2.0f * zz - xx - yy) * S(6) +
                      SH_C2[3] * xz * S(7) + SH_C2[4] * (xx - yy) * S(8);
                if (deg > 2) {
                    res = res + SH_C3[0] * y * (3.0f * xx - yy) * S(9) + SH_C3[1] * xy * z * S(10) +
                          SH_C3[2] * y * (4.0f * zz - xx - yy) * S(11) +
                          SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * S(12) +
                          SH_C3[4] * x * (4.0f * zz - xx - yy) * S(13) + SH_C3[5] * z * (xx - yy) * S(14) +
                          SH_C3[6] * x * (xx - 3.0f * yy) * S(15);
                }
            }
        }
#undef S
        res += 0.5f;
        clamped[ch] = (res < 0);
        rgb[ch] = fmaxf(res, 0.0f);
    }
}

/* ------------------------------------------------------------------------- */
/* Stage 1: per-Gaussian preprocess + inclusive scan.  Returns num_rendered.  */
/* Optional inputs are NULL when absent (the reference tests data_ptr==null). */
int gso_preprocess(int P, int D, int M, const float* means3D, const float* scales, float scale_modifier,
                   const float* rotations, const float* opacities, const float* shs, const float* cov3D_precomp,
                   const float* colors_precomp, const float* viewmatrix, const float* projmatrix,
                   const float* campos, int W, int H, float tan_fovx, float tan_fovy,
                   /* out, all caller-allocated, P rows */
                   int* radii, float* means2D /*[P,2]*/, float* depths, float* cov3D /*[P,6]*/, float* rgb /*[P,3]*/,
                   float* conic_opacity /*[P,4]*/, uint8_t* clamped /*[P,3]*/, uint32_t* tiles_touched,
                   uint32_t* point_offsets)
{
    const float focal_y = H / (2.0f * tan_fovy);
    const float focal_x = W / (2.0f * tan_fovx);
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        radii[idx] = 0;
        tiles_touched[idx] = 0;
        const float* p = means3D + 3 * idx;
        float p_view[3], p_hom[4];
        xform4x4(p, projmatrix, p_hom);
        xform4x3(p, viewmatrix, p_view);
        if (p_view[2] <= 0.2f) continue; /* auxiliary.h:154: near cull only */
        float p_w = 1.0f / (p_hom[3] + 0.0000001f);
        float p_proj[2] = {p_hom[0] * p_w, p_hom[1] * p_w};
        const float* c3;
        if (cov3D_precomp) c3 = cov3D_precomp + 6 * idx;
        else {
            cov3d_from_scale_rot(scales + 3 * idx, scale_modifier, rotations + 4 * idx, cov3D + 6 * idx);
            c3 = cov3D + 6 * idx;
        }
        ewa_t e;
        ewa_setup(p, focal_x, focal_y, tan_fovx, tan_fovy, viewmatrix, &e);
        float ca, cb, cc;
        ewa_cov2d(&e, c3, &ca, &cb, &cc);
        ca += 0.3f; cc += 0.3f;
        float det = ca * cc - cb * cb;
        if (det == 0.0f) continue;
        float det_inv = 1.f / det;
        float conic[3] = {cc * det_inv, -cb * det_inv, ca * det_inv};
        float mid = 0.5f * (ca + cc);
        float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
        float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
        float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
        float pix[2] = {ndc2pix(p_proj[0], W), ndc2pix(p_proj[1], H)};
        int rmin[2], rmax[2];
        get_rect(pix[0], pix[1], f2i(my_radius), gx, gy, rmin, rmax);
        if ((uint32_t)(rmax[0] - rmin[0]) * (uint32_t)(rmax[1] - rmin[1]) == 0) continue;
        if (!colors_precomp) sh_to_rgb(D, p, campos, shs + (size_t)idx * M * 3, rgb + 3 * idx, clamped + 3 * idx);
        depths[idx] = p_view[2];
        radii[idx] = f2i(my_radius);
        means2D[2 * idx] = pix[0]; means2D[2 * idx + 1] = pix[1];
        conic_opacity[4 * idx] = conic[0]; conic_opacity[4 * idx + 1] = conic[1];
        conic_opacity[4 * idx + 2] = conic[2]; conic_opacity[4 * idx + 3] = opacities[idx];
        tiles_touched[idx] = (uint32_t)(rmax[1] - rmin[1]) * (uint32_t)(rmax[0] - rmin[0]);
    }
    /* rasterizer_impl.cu:277: inclusive sum */
    uint32_t acc = 0;
    for (int i = 0; i < P; i++) { acc += tiles_touched[i]; point_offsets[i] = acc; }
    return (int)acc;
}

/* rasterizer_impl.cu:54-66 / auxiliary.h:139-164 */
void gso_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present)
{
    (void)projmatrix;
    for (int i = 0; i < P; i++) {
        float pv[3];
        xform4x3(means3D + 3 * i, viewmatrix, pv);
        present[i] = !(pv[2] <= 0.2f);
    }
}

/* ------------------------------------------------------------------------- */
/* Stage 2: duplicate with keys, stable sort on (tile, depth), tile ranges.    */
static void radix_sort_pairs(uint64_t* k, uint32_t* v, uint64_t* k2, uint32_t* v2, size_t n, int bits)
{
    /* LSD radix, 8 bits/pass: stable like cub::DeviceRadixSort::SortPairs */
    int passes = (bits + 7) / 8;
    size_t* cnt = (size_t*)malloc(257 * sizeof(size_t));
    for (int p = 0; p < passes; p++) {
        int sh = 8 * p;
        memset(cnt, 0, 257 * sizeof(size_t));
        for (size_t i = 0; i < n; i++) cnt[((k[i] >> sh) & 255) + 1]++;
        for (int d = 0; d < 256; d++) cnt[d + 1] += cnt[d];
        for (size_t i = 0; i < n; i++) {
            size_t pos = cnt[(k[i] >> sh) & 255]++;
            k2[pos] = k[i]; v2[pos] = v[i];
        }
        uint64_t* tk = k; k = k2; k2 = tk;
        uint32_t* tv = v; v = v2; v2 = tv;
    }
    free(cnt);
    if (passes & 1) { memcpy(k2, k, n * sizeof(uint64_t)); memcpy(v2, v, n * sizeof(uint32_t)); }
}

static uint32_t higher_msb(uint32_t n) /* rasterizer_impl.cu:35-50 */
{
    uint32_t msb = sizeof(n) * 4, step = msb;
    while (step > 1) { step /= 2; if (n >> msb) msb += step; else msb -= step; }
    if (n >> msb) msb++;
    return msb;
}

/* keys_sorted/point_list: R entries; ranges: [gx*gy,2] (zero-filled here, empty tiles stay (0,0)) */
void gso_bin_sort(int P, int R, int W, int H, const int* radii, const float* means2D, const float* depths,
                  const uint32_t* point_offsets, uint64_t* keys_sorted, uint32_t* point_list, uint32_t* ranges)
{
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    uint64_t* k2 = (uint64_t*)malloc((size_t)(R > 0 ? R : 1) * sizeof(uint64_t));
    uint32_t* v2 = (uint32_t*)malloc((size_t)(R > 0 ? R : 1) * sizeof(uint32_t));
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        if (radii[idx] > 0) {
            uint32_t off = idx == 0 ? 0 : point_offsets[idx - 1];
            int rmin[2], rmax[2];
            get_rect(means2D[2 * idx], means2D[2 * idx + 1], radii[idx], gx, gy, rmin, rmax);
            uint32_t dbits;
            memcpy(&dbits, &depths[idx], 4);
            for (int y = rmin[1]; y < rmax[1]; y++)
                for (int x = rmin[0]; x < rmax[0]; x++) {
                    uint64_t key = (uint64_t)(uint32_t)(y * gx + x);
                    key <<= 32; key |= dbits;
                    keys_sorted[off] = key; point_list[off] = (uint32_t)idx; off++;
                }
        }
    }
    int bit = (int)higher_msb((uint32_t)(gx * gy));
    radix_sort_pairs(keys_sorted, point_list, k2, v2, (size_t)R, 32 + bit);
    free(k2); free(v2);
    memset(ranges, 0, (size_t)gx * gy * 2 * sizeof(uint32_t));
    for (int i = 0; i < R; i++) { /* rasterizer_impl.cu:116-138 */
        uint32_t cur = (uint32_t)(keys_sorted[i] >> 32);
        if (i == 0) ranges[2 * cur] = 0;
        else {
            uint32_t prev = (uint32_t)(keys_sorted[i - 1] >> 32);
            if (cur != prev) { ranges[2 * prev + 1] = (uint32_t)i; ranges[2 * cur] = (uint32_t)i; }
        }
        if (i == R - 1) ranges[2 * cur + 1] = (uint32_t)R;
    }
}

/* ------------------------------------------------------------------------- */
/* Stage 3: forward blend, forward.cu:261-374 (one pixel at a time)           */
void gso_render(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const float* means2D,
                const float* features /*[P,3]*/, const float* conic_opacity, const float* bg,
                float* final_T, uint32_t* n_contrib, float* out_color /*[3,H,W]*/)
{
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < gx * gy; tile++) {
        const int tx = tile % gx, ty = tile / gx;
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        for (int ly = 0; ly < TILE; ly++)
            for (int lx = 0; lx < TILE; lx++) {
                const int px = tx * TILE + lx, py = ty * TILE + ly;
                if (px >= W || py >= H) continue;
                const float pxf = (float)px, pyf = (float)py;
                float T = 1.0f, C[3] = {0, 0, 0};
                uint32_t contributor = 0, last_contributor = 0;
                for (uint32_t i = r0; i < r1; i++) {
                    contributor++;
                    const uint32_t id = point_list[i];
                    const float dx = means2D[2 * id] - pxf, dy = means2D[2 * id + 1] - pyf;
                    const float* co = conic_opacity + 4 * id;
                    const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > 0.0f) continue;
                    const float alpha = fminf(0.99f, co[3] * expf(power));
                    if (alpha < 1.0f / 255.0f) continue;
                    const float test_T = T * (1 - alpha);
                    if (test_T < 0.0001f) break; /* done */
                    for (int ch = 0; ch < 3; ch++) C[ch] += features[3 * id + ch] * alpha * T;
                    T = test_T;
                    last_contributor = contributor;
                }
                const size_t pid = (size_t)py * W + px;
                final_T[pid] = T;
                n_contrib[pid] = last_contributor;
                for (int ch = 0; ch < 3; ch++) out_color[(size_t)ch * H * W + pid] = C[ch] + T * bg[ch];
            }
    }
}

/* ------------------------------------------------------------------------- */
/* Stage 4: blend backward, backward.cu:399-557.  The reference accumulates    */
/* with float atomics in an unspecified order; here each (tile, Gaussian)      */
/* instance is summed pixel-row-major into a private slot, then slots are      */
/* added to the per-Gaussian rows in sorted-list order (deterministic).        */
void gso_render_backward(int P, int R, int W, int H, const uint32_t* ranges, const uint32_t* point_list,
                         const float* bg, const float* means2D, const float* conic_opacity, const float* colors,
                         const float* final_T, const uint32_t* n_contrib, const float* dL_dpix /*[3,H,W]*/,
                         float* dL_dmean2D /*[P,3]*/, float* dL_dconic /*[P,4]*/, float* dL_dopacity /*[P]*/,
                         float* dL_dcolors /*[P,3]*/)
{
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    float* part = (float*)calloc((size_t)(R > 0 ? R : 1) * 9, sizeof(float));
    const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < gx * gy; tile++) {
        const int tx = tile % gx, ty = tile / gx;
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        const uint32_t todo = r1 - r0;
        for (int ly = 0; ly < TILE; ly++)
            for (int lx = 0; lx < TILE; lx++) {
                const int px = tx * TILE + lx, py = ty * TILE + ly;
                if (px >= W || py >= H) continue;
                const size_t pid = (size_t)py * W + px;
                const float pxf = (float)px, pyf = (float)py;
                const float T_final = final_T[pid];
                float T = T_final;
                uint32_t contributor = todo;
                const uint32_t last_contributor = n_contrib[pid];
                float accum_rec[3] = {0, 0, 0}, last_color[3] = {0, 0, 0}, last_alpha = 0;
                float dLp[3];
                for (int ch = 0; ch < 3; ch++) dLp[ch] = dL_dpix[(size_t)ch * H * W + pid];
                for (uint32_t j = 0; j < todo; j++) {
                    contributor--;
                    if (contributor >= last_contributor) continue;
                    const uint32_t s = r1 - 1 - j;
                    const uint32_t id = point_list[s];
                    const float dx = means2D[2 * id] - pxf, dy = means2D[2 * id + 1] - pyf;
                    const float* co = conic_opacity + 4 * id;
                    const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > 0.0f) continue;
                    const float G = expf(power);
                    const float alpha = fminf(0.99f, co[3] * G);
                    if (alpha < 1.0f / 255.0f) continue;
                    T = T / (1.f - alpha);
                    const float dchannel_dcolor = alpha * T;
                    float dL_dalpha = 0.0f;
                    float* ps = part + (size_t)s * 9;
                    for (int ch = 0; ch < 3; ch++) {
                        const float c = colors[3 * id + ch];
                        accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                        last_color[ch] = c;
                        const float dL_dchannel = dLp[ch];
                        dL_dalpha += (c - accum_rec[ch]) * dL_dchannel;
                        ps[ch] += dchannel_dcolor * dL_dchannel;
                    }
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    float bg_dot_dpixel = 0;
                    for (int ch = 0; ch < 3; ch++) bg_dot_dpixel += bg[ch] * dLp[ch];
                    dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;
                    const float dL_dG = co[3] * dL_dalpha;
                    const float gdx = G * dx, gdy = G * dy;
                    const float dG_ddelx = -gdx * co[0] - gdy * co[1];
                    const float dG_ddely = -gdy * co[2] - gdx * co[1];
                    ps[3] += dL_dG * dG_ddelx * ddelx_dx;
                    ps[4] += dL_dG * dG_ddely * ddely_dy;
                    ps[5] += -0.5f * gdx * dx * dL_dG;
                    ps[6] += -0.5f * gdx * dy * dL_dG;
                    ps[7] += -0.5f * gdy * dy * dL_dG;
                    ps[8] += G * dL_dalpha;
                }
            }
    }
    memset(dL_dmean2D, 0, (size_t)P * 3 * sizeof(float));
    memset(dL_dconic, 0, (size_t)P * 4 * sizeof(float));
    memset(dL_dopacity, 0, (size_t)P * sizeof(float));
    memset(dL_dcolors, 0, (size_t)P * 3 * sizeof(float));
    for (int s = 0; s < R; s++) {
        const uint32_t id = point_list[s];
        const float* ps = part + (size_t)s * 9;
        dL_dcolors[3 * id] += ps[0]; dL_dcolors[3 * id + 1] += ps[1]; dL_dcolors[3 * id + 2] += ps[2];
        dL_dmean2D[3 * id] += ps[3]; dL_dmean2D[3 * id + 1] += ps[4];
        dL_dconic[4 * id] += ps[5]; dL_dconic[4 * id + 1] += ps[6]; dL_dconic[4 * id + 3] += ps[7];
        dL_dopacity[id] += ps[8];
    }
    free(part);
}

/* ------------------------------------------------------------------------- */
/* Stage 5: per-Gaussian backward.  backward.cu:144-274 (cov2D), :346-396      */
/* (projection), :20-139 (SH), :278-341 (cov3D).  Gradient arrays are fully    */
/* written (zeros for culled Gaussians), as the reference's zero-initialised   */
/* outputs would read (rasterize_points.cu:151-159).                           */
static void sh_backward(int deg, int M, const float* pos, const float* campos, const float* sh,
                        const uint8_t* clamped, const float* dL_dcolor, float* dL_dmean, float* dL_dsh)
{
    float dir_orig[3] = {pos[0] - campos[0], pos[1] - campos[1], pos[2] - campos[2]};
    float len = sqrtf(dir_orig[0] * dir_orig[0] + dir_orig[1] * dir_orig[1] + dir_orig[2] * dir_orig[2]);
    float x = dir_orig[0] / len, y = dir_orig[1] / len, z = dir_orig[2] / len;
    float dL_dRGB[3];
    for (int ch = 0; ch < 3; ch++) dL_dRGB[ch] = dL_dcolor[ch] * (clamped[ch] ? 0.f : 1.f);
    float dRGBdx[3] = {0, 0, 0}, dRGBdy[3] = {0, 0, 0}, dRGBdz[3] = {0, 0, 0};
    (void)M;
#define SH(i, ch) sh[(i) * 3 + (ch)]
#define OUT(i, w) for (int ch = 0; ch < 3; ch++) dL_dsh[(i) * 3 + ch] = (w) * dL_dRGB[ch]
    OUT(0, SH_C0);
    if (deg > 0) {
        float w1 = -SH_C1 * y, w2 = SH_C1 * z, w3 = -SH_C1 * x;
        OUT(1, w1); OUT(2, w2); OUT(3, w3);
        for (int ch = 0; ch < 3; ch++) {
            dRGBdx[ch] = -SH_C1 * SH(3, ch); dRGBdy[ch] = -SH_C1 * SH(1, ch); dRGBdz[ch] = SH_C1 * SH(2, ch);
        }
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            float w4 = SH_C2[0] * xy, w5 = SH_C2[1] * yz, w6 = SH_C2[2] * (2.f * zz - xx - yy);
            float w7 = SH_C2[3] * xz, w8 = SH_C2[4] * (xx - yy);
            OUT(4, w4); OUT(5, w5); OUT(6, w6); OUT(7, w7); OUT(8, w8);
            for (int ch = 0; ch < 3; ch++) {
                dRGBdx[ch] += SH_C2[0] * y * SH(4, ch) + SH_C2[2] * 2.f * -x * SH(6, ch) + SH_C2[3] * z * SH(7, ch) +
                              SH_C2[4] * 2.f * x * SH(8, ch);
                dRGBdy[ch] += SH_C2[0] * x * SH(4, ch) + SH_C2[1] * z * SH(5, ch) + SH_C2[2] * 2.f * -y * SH(6, ch) +
                              SH_C2[4] * 2.f * -y * SH(8, ch);
                dRGBdz[ch] += SH_C2[1] * y * SH(5, ch) + SH_C2[2] * 2.f * 2.f * z * SH(6, ch) + SH_C2[3] * x * SH(7, ch);
            }
            if (deg > 2) {
                float w9 = SH_C3[0] * y * (3.f * xx - yy), w10 = SH_C3[1] * xy * z;
                float w11 = SH_C3[2] * y * (4.f * zz - xx - yy), w12 = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
                float w13 = SH_C3[4] * x * (4.f * zz - xx - yy), w14 = SH_C3[5] * z * (xx - yy);
                float w15 = SH_C3[6] * x * (xx - 3.f * yy);
                OUT(9, w9); OUT(10, w10); OUT(11, w11); OUT(12, w12); OUT(13, w13); OUT(14, w14); OUT(15, w15);
                for (int ch = 0; ch < 3; ch++) {
                    dRGBdx[ch] += (SH_C3[0] * SH(9, ch) * 3.f * 2.f * xy + SH_C3[1] * SH(10, ch) * yz +
                                   SH_C3[2] * SH(11, ch) * -2.f * xy + SH_C3[3] * SH(12, ch) * -3.f * 2.f * xz +
                                   SH_C3[4] * SH(13, ch) * (-3.f * xx + 4.f * zz - yy) + SH_C3[5] * SH(14, ch) * 2.f * xz +
                                   SH_C3[6] * SH(15, ch) * 3.f * (xx - yy));
                    dRGBdy[ch] += (SH_C3[0] * SH(9, ch) * 3.f * (xx - yy) + SH_C3[1] * SH(10, ch) * xz +
                                   SH_C3[2] * SH(11, ch) * (-3.f * yy + 4.f * zz - xx) +
                                   SH_C3[3] * SH(12, ch) * -3.f * 2.f * yz + SH_C3[4] * SH(13, ch) * -2.f * xy +
                                   SH_C3[5] * SH(14, ch) * -2.f * yz + SH_C3[6] * SH(15, ch) * -3.f * 2.f * xy);
                    dRGBdz[ch] += (SH_C3[1] * SH(10, ch) * xy + SH_C3[2] * SH(11, ch) * 4.f * 2.f * yz +
                                   SH_C3[3] * SH(12, ch) * 3.f * (2.f * zz - xx - yy) +
                                   SH_C3[4] * SH(13, ch) * 4.f * 2.f * xz + SH_C3[5] * SH(14, ch) * (xx - yy));
                }
            }
        }
    }
#undef SH
#undef OUT
    float dL_ddir[3] = {dRGBdx[0] * dL_dRGB[0] + dRGBdx[1] * dL_dRGB[1] + dRGBdx[2] * dL_dRGB[2],
                        dRGBdy[0] * dL_dRGB[0] + dRGBdy[1] * dL_dRGB[1] + dRGBdy[2] * dL_dRGB[2],
                        dRGBdz[0] * dL_dRGB[0] + dRGBdz[1] * dL_dRGB[1] + dRGBdz[2] * dL_dRGB[2]};
    /* auxiliary.h:107-117 dnormvdv */
    const float* v = dir_orig; const float* dv = dL_ddir;
    float sum2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    dL_dmean[0] += ((+sum2 - v[0] * v[0]) * dv[0] - v[1] * v[0] * dv[1] - v[2] * v[0] * dv[2]) * invsum32;
    dL_dmean[1] += (-v[0] * v[1] * dv[0] + (sum2 - v[1] * v[1]) * dv[1] - v[2] * v[1] * dv[2]) * invsum32;
    dL_dmean[2] += (-v[0] * v[2] * dv[0] - v[1] * v[2] * dv[1] + (sum2 - v[2] * v[2]) * dv[2]) * invsum32;
}

static void cov3d_backward(const float* scale, float mod, const float* q, const float* dL_dcov3D, float* dL_dscale, float* dL_drot)
{
    float r = q[0], x = q[1], y = q[2], z = q[3];
    float R[3][3];
    R[0][0] = 1.f - 2.f * (y * y + z * z); R[0][1] = 2.f * (x * y - r * z); R[0][2] = 2.f * (x * z + r * y);
    R[1][0] = 2.f * (x * y + r * z); R[1][1] = 1.f - 2.f * (x * x + z * z); R[1][2] = 2.f * (y * z - r * x);
    R[2][0] = 2.f * (x * z - r * y); R[2][1] = 2.f * (y * z + r * x); R[2][2] = 1.f - 2.f * (x * x + y * y);
    float s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
    float M[3][3];
    for (int c = 0; c < 3; c++) for (int rr = 0; rr < 3; rr++) M[c][rr] = s[rr] * R[c][rr];
    const float* d = dL_dcov3D;
    float dS[3][3] = {{d[0], 0.5f * d[1], 0.5f * d[2]}, {0.5f * d[1], d[3], 0.5f * d[4]}, {0.5f * d[2], 0.5f * d[4], d[5]}};
    /* dL_dM = 2 * M * dL_dSigma ; (2*M)[k][r] = M[k][r]*2 ; result[c][r] = sum_k (2M)[k][r] * dS[c][k] */
    float dM[3][3];
    for (int c = 0; c < 3; c++)
        for (int rr = 0; rr < 3; rr++)
            dM[c][rr] = (M[0][rr] * 2.0f) * dS[c][0] + (M[1][rr] * 2.0f) * dS[c][1] + (M[2][rr] * 2.0f) * dS[c][2];
    /* Rt[c][r] = R[r][c]; dMt[c][r] = dM[r][c] */
    float dMt[3][3];
    for (int c = 0; c < 3; c++) for (int rr = 0; rr < 3; rr++) dMt[c][rr] = dM[rr][c];
    for (int c = 0; c < 3; c++) dL_dscale[c] = R[0][c] * dMt[c][0] + R[1][c] * dMt[c][1] + R[2][c] * dMt[c][2];
    for (int c = 0; c < 3; c++) for (int rr = 0; rr < 3; rr++) dMt[c][rr] *= s[c];
    dL_drot[0] = 2 * z * (dMt[0][1] - dMt[1][0]) + 2 * y * (dMt[2][0] - dMt[0][2]) + 2 * x * (dMt[1][2] - dMt[2][1]);
    dL_drot[1] = 2 * y * (dMt[1][0] + dMt[0][1]) + 2 * z * (dMt[2][0] + dMt[0][2]) + 2 * r * (dMt[1][2] - dMt[2][1]) - 4 * x * (dMt[2][2] + dMt[1][1]);
    dL_drot[2] = 2 * x * (dMt[1][0] + dMt[0][1]) + 2 * r * (dMt[2][0] - dMt[0][2]) + 2 * z * (dMt[1][2] + dMt[2][1]) - 4 * y * (dMt[2][2] + dMt[0][0]);
    dL_drot[3] = 2 * r * (dMt[0][1] - dMt[1][0]) + 2 * x * (dMt[2][0] + dMt[0][2]) + 2 * y * (dMt[1][2] + dMt[2][1]) - 4 * z * (dMt[1][1] + dMt[0][0]);
}

void gso_preprocess_backward(int P, int D, int M, const float* means3D, const int* radii, const float* shs,
                             const uint8_t* clamped, const float* scales, const float* rotations,
                             float scale_modifier, const float* cov3Ds /* precomp or computed */,
                             const float* viewmatrix, const float* projmatrix, int W, int H, float tan_fovx,
                             float tan_fovy, const float* campos, const float* dL_dmean2D /*[P,3]*/,
                             const float* dL_dconic /*[P,4]*/, const float* dL_dcolor /*[P,3]*/,
                             float* dL_dmeans3D /*[P,3]*/, float* dL_dcov3D /*[P,6]*/, float* dL_dsh /*[P,M,3]*/,
                             float* dL_dscale /*[P,3]*/, float* dL_drot /*[P,4]*/)
{
    const float h_y = H / (2.0f * tan_fovy), h_x = W / (2.0f * tan_fovx);
    memset(dL_dmeans3D, 0, (size_t)P * 3 * sizeof(float));
    memset(dL_dcov3D, 0, (size_t)P * 6 * sizeof(float));
    if (shs) memset(dL_dsh, 0, (size_t)P * M * 3 * sizeof(float));
    if (scales) { memset(dL_dscale, 0, (size_t)P * 3 * sizeof(float)); memset(dL_drot, 0, (size_t)P * 4 * sizeof(float)); }
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        if (!(radii[idx] > 0)) continue;
        const float* mean = means3D + 3 * idx;
        const float* c3 = cov3Ds + 6 * idx;
        /* ---- computeCov2DCUDA ---- */
        const float dLc[3] = {dL_dconic[4 * idx], dL_dconic[4 * idx + 1], dL_dconic[4 * idx + 3]};
        ewa_t e;
        ewa_setup(mean, h_x, h_y, tan_fovx, tan_fovy, viewmatrix, &e);
        float a, b, c;
        ewa_cov2d(&e, c3, &a, &b, &c);
        a += 0.3f; c += 0.3f;
        const float(*T)[3] = e.T;
        float denom = a * c - b * b;
        float dL_da = 0, dL_db = 0, dL_dc = 0;
        float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        float* dcov = dL_dcov3D + 6 * idx;
        if (denom2inv != 0) {
            dL_da = denom2inv * (-c * c * dLc[0] + 2 * b * c * dLc[1] + (denom - a * c) * dLc[2]);
            dL_dc = denom2inv * (-a * a * dLc[2] + 2 * a * b * dLc[1] + (denom - a * c) * dLc[0]);
            dL_db = denom2inv * 2 * (b * c * dLc[0] - (denom + 2 * b * b) * dLc[1] + a * b * dLc[2]);
            dcov[0] = (T[0][0] * T[0][0] * dL_da + T[0][0] * T[1][0] * dL_db + T[1][0] * T[1][0] * dL_dc);
            dcov[3] = (T[0][1] * T[0][1] * dL_da + T[0][1] * T[1][1] * dL_db + T[1][1] * T[1][1] * dL_dc);
            dcov[5] = (T[0][2] * T[0][2] * dL_da + T[0][2] * T[1][2] * dL_db + T[1][2] * T[1][2] * dL_dc);
            dcov[1] = 2 * T[0][0] * T[0][1] * dL_da + (T[0][0] * T[1][1] + T[0][1] * T[1][0]) * dL_db + 2 * T[1][0] * T[1][1] * dL_dc;
            dcov[2] = 2 * T[0][0] * T[0][2] * dL_da + (T[0][0] * T[1][2] + T[0][2] * T[1][0]) * dL_db + 2 * T[1][0] * T[1][2] * dL_dc;
            dcov[4] = 2 * T[0][2] * T[0][1] * dL_da + (T[0][1] * T[1][2] + T[0][2] * T[1][1]) * dL_db + 2 * T[1][1] * T[1][2] * dL_dc;
        }
        const float V[3][3] = {{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}};
#define TV(rw, k) (T[rw][0] * V[k][0] + T[rw][1] * V[k][1] + T[rw][2] * V[k][2])
        float dL_dT00 = 2 * TV(0, 0) * dL_da + TV(1, 0) * dL_db;
        float dL_dT01 = 2 * TV(0, 1) * dL_da + TV(1, 1) * dL_db;
        float dL_dT02 = 2 * TV(0, 2) * dL_da + TV(1, 2) * dL_db;
        float dL_dT10 = 2 * TV(1, 0) * dL_dc + TV(0, 0) * dL_db;
        float dL_dT11 = 2 * TV(1, 1) * dL_dc + TV(0, 1) * dL_db;
        float dL_dT12 = 2 * TV(1, 2) * dL_dc + TV(0, 2) * dL_db;
#undef TV
        /* W[k][r] = vm[4*r+k] */
#define Wm(k, rr) viewmatrix[4 * (rr) + (k)]
        float dL_dJ00 = Wm(0, 0) * dL_dT00 + Wm(0, 1) * dL_dT01 + Wm(0, 2) * dL_dT02;
        float dL_dJ02 = Wm(2, 0) * dL_dT00 + Wm(2, 1) * dL_dT01 + Wm(2, 2) * dL_dT02;
        float dL_dJ11 = Wm(1, 0) * dL_dT10 + Wm(1, 1) * dL_dT11 + Wm(1, 2) * dL_dT12;
        float dL_dJ12 = Wm(2, 0) * dL_dT10 + Wm(2, 1) * dL_dT11 + Wm(2, 2) * dL_dT12;
#undef Wm
        float tz = 1.f / e.t[2], tz2 = tz * tz, tz3 = tz2 * tz;
        float dL_dtx = e.xmul * -h_x * tz2 * dL_dJ02;
        float dL_dty = e.ymul * -h_y * tz2 * dL_dJ12;
        float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * e.t[0]) * tz3 * dL_dJ02 + (2 * h_y * e.t[1]) * tz3 * dL_dJ12;
        float* dm = dL_dmeans3D + 3 * idx;
        const float* vm = viewmatrix;
        dm[0] = vm[0] * dL_dtx + vm[1] * dL_dty + vm[2] * dL_dtz;
        dm[1] = vm[4] * dL_dtx + vm[5] * dL_dty + vm[6] * dL_dtz;
        dm[2] = vm[8] * dL_dtx + vm[9] * dL_dty + vm[10] * dL_dtz;
        /* ---- preprocessCUDA (bwd): projection path ---- */
        const float* proj = projmatrix;
        float m_hom[4];
        xform4x4(mean, proj, m_hom);
        float m_w = 1.0f / (m_hom[3] + 0.0000001f);
        float mul1 = (proj[0] * mean[0] + proj[4] * mean[1] + proj[8] * mean[2] + proj[12]) * m_w * m_w;
        float mul2 = (proj[1] * mean[0] + proj[5] * mean[1] + proj[9] * mean[2] + proj[13]) * m_w * m_w;
        const float gx2 = dL_dmean2D[3 * idx], gy2 = dL_dmean2D[3 * idx + 1];
        float d0 = (proj[0] * m_w - proj[3] * mul1) * gx2 + (proj[1] * m_w - proj[3] * mul2) * gy2;
        float d1 = (proj[4] * m_w - proj[7] * mul1) * gx2 + (proj[5] * m_w - proj[7] * mul2) * gy2;
        float d2 = (proj[8] * m_w - proj[11] * mul1) * gx2 + (proj[9] * m_w - proj[11] * mul2) * gy2;
        dm[0] += d0; dm[1] += d1; dm[2] += d2;
        if (shs)
            sh_backward(D, M, mean, campos, shs + (size_t)idx * M * 3, clamped + 3 * idx, dL_dcolor + 3 * idx, dm,
                        dL_dsh + (size_t)idx * M * 3);
        if (scales)
            cov3d_backward(scales + 3 * idx, scale_modifier, rotations + 4 * idx, dcov, dL_dscale + 3 * idx, dL_drot + 4 * idx);
    }
}
