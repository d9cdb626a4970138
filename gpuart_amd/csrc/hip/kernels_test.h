// kernels_test.h — kernels behind the gpuart_hip_test_* hooks: each runs one device function over arrays so that the
// parity tests can compare it with the golden vectors. Included by gpuart_hip.hip only.
#pragma once
#include "kernels_pipeline.h"

namespace {

// ---- test-hook kernels ---------------------------------------------------------------------------
__global__ void k_test_random(const float4 *in, int n, float4 *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 v = in[i];
    out[i] = make_float4(random1(v.x), random2(v.x, v.y), random3(f3(v.x, v.y, v.z)), random4(v));
}
__global__ void k_test_math(const float4 *in, int n, float4 *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s, c;
    sincos_lp(in[i].x, s, c);
    out[i] = make_float4(s, c, pow_lp(in[i].y, 16.0f), sqrtf(in[i].y));
}
__global__ void k_test_hemisphere(const float4 *v, const float4 *ri, int n, float4 *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    F3 r = random_hemisphere_direction(xyz(v[i]), xyz(ri[i]));
    out[i] = make_float4(r.x, r.y, r.z, 0);
}
__global__ void k_test_inside_cone(const float4 *v, const float4 *nrm, const float4 *ri, float ha, int n, float4 *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    F3 r = random_direction_inside_cone(xyz(v[i]), xyz(nrm[i]), ha, xyz(ri[i]));
    out[i] = make_float4(r.x, r.y, r.z, 0);
}
struct Float4Arg { float v[4]; };
__global__ void k_test_sky(const float4 *dir, Float4Arg sda, int n, float4 *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    F3 r = sky_color(xyz(dir[i]), sda.v);
    out[i] = make_float4(r.x, r.y, r.z, 0);
}
/// recs: n device-layout primitive records (3 quads each)
__global__ void k_test_intersect(const float4 *rs, const float4 *rd, const float4 *recs, int n, float4 *o0, float4 *o1) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Ray r; r.o = xyz(rs[i]); r.d = xyz(rd[i]);
    float pos; F3 p = f3(0, 0, 0), nn = f3(0, 0, 0); int t;
    prim_hit(r, recs[3 * i], recs[3 * i + 1], recs[3 * i + 2], pos, p, nn, t);
    if (pos > 0) { o0[i] = make_float4(pos, p.x, p.y, p.z); o1[i] = make_float4(nn.x, nn.y, nn.z, 0); }
    else { o0[i] = make_float4(pos, 0, 0, 0); o1[i] = make_float4(0, 0, 0, 0); }
}
__global__ void k_test_aabb(const float4 *rs, const float4 *rd, const float4 *bmin, const float4 *bmax, int n, float4 *out, int quick) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Ray r; r.o = xyz(rs[i]); r.d = xyz(rd[i]);
    float pos;
    const F3 lo = xyz(bmin[i]), hi = xyz(bmax[i]), rdiv = f3(1 / r.d.x, 1 / r.d.y, 1 / r.d.z);
    // what the uploader decides per tree (converter.h): ordered AND finite bounds take the fast form
    const float big = 3.4028234663852886e+38f;
    const bool regular = (lo.x <= hi.x) & (lo.y <= hi.y) & (lo.z <= hi.z) & (fabsf(lo.x) <= big) & (fabsf(lo.y) <= big) & (fabsf(lo.z) <= big) &
                         (fabsf(hi.x) <= big) & (fabsf(hi.y) <= big) & (fabsf(hi.z) <= big);
    bool h;
    if (regular) {
        // as trav_step_box tests a box of a regular tree: the quick answer (box_quick.h, the slack sized by this box's own planes —
        // the tightest a tree could have), and the six face tests where it is withdrawn
        const float pmax = fmaxf(fmaxf(fmaxf(fabsf(lo.x), fabsf(lo.y)), fmaxf(fabsf(lo.z), fabsf(hi.x))), fmaxf(fabsf(hi.y), fabsf(hi.z)));
        const bool sub = !(gq_plane_ok(lo.x) & gq_plane_ok(lo.y) & gq_plane_ok(lo.z) & gq_plane_ok(hi.x) & gq_plane_ok(hi.y) & gq_plane_ok(hi.z));  // (as converter.h)
        const float cs = gq_ray_slack((GD_QUICK_BOXES && quick && !sub) ? gq_slack_of_tree(pmax) : __builtin_inff(), r.o.x, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z, rdiv.x, rdiv.y, rdiv.z);
        if (!box_quick(r, rdiv, lo, hi, cs, pos, h)) h = aabb_entry(r, rdiv, lo, hi, pos);
    } else
        h = aabb_entry<true>(r, rdiv, lo, hi, pos);
    out[i] = make_float4(h ? 1.0f : 0.0f, h ? pos : 0.0f, 0, 0);
}
template <bool ANY, bool NEAREST = false>
__global__ void __launch_bounds__(BLOCK) k_test_traverse(Scene sc, const float4 *rs, const float4 *rd, Float4Arg us, int n,
                                                         float4 *o0, float4 *o1, uint4 *spill) {
    __shared__ uint2 ring_a[GD_RING * BLOCK];
    __shared__ float ring_b[GD_RING * BLOCK];
    TravStack st = make_stack(ring_a, ring_b, spill, gridDim.x * BLOCK);
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        Ray r; r.o = xyz(rs[i]); r.d = xyz(rd[i]);
        float closest; uint32_t prim;
        traverse<ANY, false, NEAREST>(sc, r, st, closest, prim, nullptr);
        if (ANY) {
            o0[i] = make_float4(prim != GD_NO_PRIM ? 1.0f : 0.0f, 0, 0, 0);
            o1[i] = make_float4(0, 0, 0, 0);
            continue;
        }
        Surface h; h.p = f3(0, 0, 0); h.n = f3(0, 0, 0);
        bool ush;
        resolve_hit(sc, r, closest, prim, us.v, h, ush);
        if (h.ptype >= 0) {
            o0[i] = make_float4(h.pos, h.p.x, h.p.y, h.p.z);
            o1[i] = make_float4(h.n.x, h.n.y, h.n.z, (float)h.ptype + (ush ? 0.5f : 0.0f));
        } else {
            o0[i] = make_float4(-1, 0, 0, 0);
            o1[i] = make_float4(0, 0, 0, -1);
        }
    }
}
__global__ void k_test_cam_rays(Frame f, float4 *rstart, float4 *rdir) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= f.tw * f.th) return;
    uint32_t lx = i % f.tw, ly = i / f.tw;
    F3 s, d;
    camera_ray(f, f.x0 + lx, frame_y(f, ly), s, d);
    rstart[i] = make_float4(s.x, s.y, s.z, 0);
    rdir[i] = make_float4(d.x, d.y, d.z, 0);
}

}  // namespace

