// device_shade.h — camera rays, sky, direction samplers and the two per-pixel radiance loops.
#pragma once
#include "device_scene.h"
#include "gpuart_hip.h"

namespace gd {

/// Everything a render kernel needs besides the scene (passed by value as a kernel argument).
struct Frame {
    uint32_t W, H;             ///< full frame
    uint32_t x0, y0, tw, th;   ///< this context's tile: tw x th local pixels starting at (x0, y0)
    uint32_t band_rows, band_stride;  ///< local row ly is frame row y0 + (ly / band_rows) * band_stride + ly % band_rows
                                      ///< (band_rows = band_stride = th: a plain rectangle; otherwise row bands interleaved
                                      ///<  with other contexts for load balance)
    float cam_pos[3], bottom_left[3], delta_horz[3], delta_vert[3];
    float uv_coef[12];         ///< llvmpipe plane equations of the quad's UV (A.u A.v B.u B.v)
    const uint32_t *tile_order;  ///< null, or a permutation of the tile's 8x8 pixel blocks: path slots [64 k, 64 k + 64) hold block
                                 ///< tile_order[k] (kernels_pipeline.h slot_pixel) — the order paths are BORN in; no pixel depends on it
    uint32_t *tile_cost;         ///< null, or per 8x8 block (row-major number) a counter: this run adds one per shaded path segment
                                 ///< (kernels_pipeline.h tile_cost_add; k_tile_order turns the counts into the next tile_order)
};

/// Frame row of local row `ly` of the tile.
GD_FN uint32_t frame_y(const Frame &f, uint32_t ly) { return f.y0 + (ly / f.band_rows) * f.band_stride + ly % f.band_rows; }

// ---- reference shaders/vertex.glsl:29-37 as rasterised by llvmpipe (DESIGN.md "UV") -----------
GD_FN void pixel_uv(const Frame &f, uint32_t x, uint32_t y, float &u, float &v) {
    long long side = (long long)(2 * y + 1) * f.W - (long long)(2 * x + 1) * f.H;
    const float *c = (side <= 0) ? f.uv_coef : f.uv_coef + 6;
    float X = (float)x, Y = (float)y;
    u = fmaf(c[2], Y, fmaf(c[1], X, c[0]));
    v = fmaf(c[5], Y, fmaf(c[4], X, c[3]));
}

// ---- reference shaders/cam_init.glsl:45-50 ---------------------------------------------------
GD_FN void camera_ray(const Frame &f, uint32_t x, uint32_t y, F3 &rstart, F3 &rdir) {
    float u, v;
    pixel_uv(f, x, y, u, v);
    rstart = f3((f.bottom_left[0] + f.delta_horz[0] * u) + f.delta_vert[0] * v,
                (f.bottom_left[1] + f.delta_horz[1] * u) + f.delta_vert[1] * v,
                (f.bottom_left[2] + f.delta_horz[2] * u) + f.delta_vert[2] * v);
    rdir = normalize3(rstart - f3(f.cam_pos[0], f.cam_pos[1], f.cam_pos[2]));
}

// ---- reference shaders/common.glsl:40-46 -----------------------------------------------------
// As Mesa compiles it: all(lessThan(a, b)) becomes !any(a >= b) (NIR pushes the negation into the comparisons), so a NaN
// component counts as "small"; the z component of normalize(vec3(v.y, -v.x, 0)) is the constant 0, not 0 * rsq
// (tests/golden/hemisphere_wild.npz).
GD_FN F3 get_orthogonal(F3 v) {
    if (!(fabsf(v.x) >= 1.0e-6f || fabsf(v.y) >= 1.0e-6f)) return f3(1, 0, 0);
    const F3 w = f3(v.y, -v.x, 0.0f);
    const float inv = 1.0f / sqrtf(dot3(w, w));
    return f3(v.y * inv, -v.x * inv, 0.0f);
}

// ---- reference shaders/common.glsl:49-66 -----------------------------------------------------
GD_FN F3 random_hemisphere_direction(F3 v, F3 ri) {
    const float PIDBL = 3.1415926f * 2;
    float a = PIDBL * random3(ri);
    float r2 = random3(f3(ri.z, ri.x, ri.y));
    float sr2 = sqrtf(1.0f - r2);
    float s, c;
    sincos_lp(a, s, c);
    float x = c * sr2, y = s * sr2, z = sqrtf(r2);
    F3 t = get_orthogonal(v);
    // tangent.z is the constant 0 in both branches of GetOrthogonal: Mesa folds its products away (0 * NaN and 0 * inf included)
    F3 ct = f3(-(v.z * t.y), v.z * t.x, v.x * t.y - v.y * t.x);
    return f3((t.x * x + ct.x * y) + v.x * z, (t.y * x + ct.y * y) + v.y * z, ct.z * y + v.z * z);
}

// ---- reference shaders/common.glsl:69-76 (GLSL mat3 is column-major) --------------------------
GD_FN F3 rotate3(F3 v, F3 a, float sine, float cosine) {
    float omc = 1 - cosine;
    F3 c0 = f3((a.x * a.x + (1 - a.x * a.x) * cosine), (a.x * a.y * omc - a.z * sine), (a.x * a.z * omc + a.y * sine));
    F3 c1 = f3((a.x * a.y * omc + a.z * sine), (a.y * a.y + (1 - a.y * a.y) * cosine), (a.y * a.z * omc - a.x * sine));
    F3 c2 = f3((a.x * a.z * omc - a.y * sine), (a.y * a.z * omc + a.x * sine), (a.z * a.z + (1 - a.z * a.z) * cosine));
    return f3((c0.x * v.x + c1.x * v.y) + c2.x * v.z, (c0.y * v.x + c1.y * v.y) + c2.y * v.z,
              (c0.z * v.x + c1.z * v.y) + c2.z * v.z);
}

// ---- reference shaders/common.glsl:81-106 (with its `sina = 1 - cosa*cosa`) -------------------
GD_FN F3 random_direction_inside_cone(F3 v, F3 normal, float halfAngle, F3 ri) {
    float a = random2(ri.x, ri.y) * halfAngle;
    float b = random2(ri.y, ri.z) * 2 * 3.14159f;
    float sina, cosa, sinb, cosb;
    sincos_lp(a, sina, cosa);
    sincos_lp(b, sinb, cosb);
    float sinc = length3(cross3(-v, normal)) / length3(v);
    if (cosa < sinc) {
        cosa = sinc;
        sina = 1 - cosa * cosa;
    }
    F3 vo = get_orthogonal(v);
    F3 w = normalize3(rotate3(v, vo, sina, cosa));
    return rotate3(w, v, sinb, cosb);
}

// ---- reference shaders/sky.glsl:34-60 --------------------------------------------------------
GD_FN F3 sky_color(F3 dir, const float sda[4]) {
    F3 nd = normalize3(dir);
    F3 hp = f3(nd.x, nd.y, 0.0f);  // cross(cross((0,0,1), nd), (0,0,1))
    // hp.z is the constant 0: Mesa folds 0 * rsq(...) and nd.z * 0 away (also where rsq is inf: |hp| underflows for a direction
    // close to the zenith), so the z term never makes a NaN; what is left of dot() is its (y + x) part (tests/golden/sky_wild.npz)
    float weight = 1.0f;
    if (dir.z >= 0) {
        const float inv = 1.0f / sqrtf(dot3(hp, hp));
        weight = nd.y * (hp.y * inv) + nd.x * (hp.x * inv);
    }
    float sw = 1.0f - sda[3] / (3.1415926f / 2);
    F3 cz = f3(mixf(0.2f, 0.0f, sw), mixf(0.6f, 0.2f, sw), mixf(1.0f, 0.5f, sw));
    F3 ch = f3(1.0f, mixf(1.0f, 0.647f, sw), mixf(1.0f, 0.367f, sw));
    float pw = pow_lp(weight, 16.0f);
    // red: mix(x, 1.0, t) is lowered by NIR as x*(1-t) + t
    return f3(cz.x * (1.0f - pw) + pw, mixf(cz.y, ch.y, pw), mixf(cz.z, ch.z, pw));
}

GD_FN F3 primitive_color(int ptype) {
    // reference shaders/path_tracing.glsl:123-126 and direct_lighting.glsl:84-87
    if (ptype == P_SPHERE) return f3(0.65f, 0.4f, 0.35f);
    if (ptype == P_DISC) return f3(0.1f, 0.2f, 0.1f);
    return f3(0.3f, 0.3f, 0.3f);
}

GD_FN F3 lambert(F3 lightDir, F3 normal, F3 diffuse) {
    float dotp = dot3(lightDir, normal);
    if (dotp > 0) return diffuse * dotp;  // lightIntensity == 1
    return f3(0, 0, 0);
}

/// Whether the Sun is visible from a shadow query's result: nothing in the BVH was hit and the user
/// sphere does not occlude either (reference CheckIntersectionInclUserSphere, intersection.glsl:98).
GD_FN bool sun_visible(const gpuart_params &P, F3 origin, F3 sun, uint32_t shadow_prim) {
    if (shadow_prim != GD_NO_PRIM) return false;
    Ray sr; sr.o = origin; sr.d = sun;
    float usPos; F3 a, b;
    sphere_hit(sr, f3(P.userSphere[0], P.userSphere[1], P.userSphere[2]), P.userSphere[3], usPos, a, b);
    return !(usPos > GD_VISIBILITY_OFFSET);
}

// ---- reference shaders/direct_lighting.glsl:134-207 (whole pixel in one thread) ----------------------
template <bool REFWORK>
GD_FN F3 direct_lighting_pixel(const Scene &sc, const gpuart_params &P, F3 rstart, F3 rdir, TravStack &st,
                               WorkCounters *wc) {
    const float AMBIENT = 0.15f;
    F3 sun = f3(P.sunDirAlt[0], P.sunDirAlt[1], P.sunDirAlt[2]);
    F3 cw = f3(1, 1, 1), out = f3(0, 0, 0);
    for (int i = 0; i <= 1; i++) {
        Ray r; r.o = rstart; r.d = rdir;
        float closest; uint32_t prim; Surface h; bool ush;
        traverse<false, REFWORK>(sc, r, st, closest, prim, wc);
        resolve_hit(sc, r, closest, prim, P.userSphere, h, ush);
        if ((P.userSphereFlags & 2u) && ush) {
            rstart = h.p;
            rdir = reflect3(rdir, h.n);
            cw = cw * primitive_color(P_SPHERE);
        } else if ((P.userSphereFlags & 1u) && ush) {
            out = f3(1, 1, 1);
        } else {
            if (h.ptype != -1) {
                F3 diffuse = primitive_color(h.ptype) * cw;
                if (P.sunEnabled == 1) {
                    Ray sr; sr.o = h.p; sr.d = sun;
                    float sc_closest; uint32_t sprim;
                    traverse<!REFWORK, REFWORK>(sc, sr, st, sc_closest, sprim, wc);
                    if (sun_visible(P, h.p, sun, sprim)) out = out + lambert(sun, h.n, diffuse);
                }
                if (P.userSphereFlags & 1u) {
                    F3 dts = f3(P.userSphere[0], P.userSphere[1], P.userSphere[2]) - h.p;
                    float dist = length3(dts);
                    Ray er; er.o = h.p; er.d = f3(dts.x / dist, dts.y / dist, dts.z / dist);
                    float ec; uint32_t eprim;
                    traverse<false, REFWORK>(sc, er, st, ec, eprim, wc);
                    if (eprim == GD_NO_PRIM || ec > dist) {
                        F3 l = lambert(er.d, h.n, diffuse);
                        float d2 = dot3(dts, dts);  // dist*dist: NIR folds sqrt(a)*sqrt(a) to |a|
                        out = out + f3(l.x / d2, l.y / d2, l.z / d2);
                    }
                }
                out = out + AMBIENT * diffuse;
            } else {
                out = cw * sky_color(rdir, P.sunDirAlt);
            }
            break;
        }
    }
    return out;
}

// ---- reference shaders/path_tracing.glsl:133-256, split at its two BVH queries ---------------------------
// The per-path state between queries (what the GLSL keeps in registers across loop iterations):
//   ray (rstart, rdir) of the next/current segment, colorWeight, pathColor, segment counter i.
// path_begin  : lines 154-175  (jitter, first ray)
// path_shade  : lines 182-233  (after the closest-hit query of a segment)
// path_sun    : lines 234-245  (after the Sun shadow query)
// path_finish : lines 247-252  (after the segment loop)

/// First ray of path j of a pixel (path_tracing.glsl:141-170).
GD_FN void path_begin(const gpuart_params &P, float4 seed, int j, F3 rstart0, F3 rdir0, F3 &rstart, F3 &rdir) {
    F3 o1;
    if (fabsf(rdir0.x) > 1.0e-5f || fabsf(rdir0.y) > 1.0e-5f) o1 = normalize3(f3(rdir0.y, -rdir0.x, 0));
    else o1 = normalize3(f3(0, -rdir0.z, rdir0.y));
    F3 o2 = cross3(normalize3(rdir0), o1);
    float rand1 = random1(seed.x + (float)j);
    float rand2 = random1(seed.y + (float)j);
    rstart = (rstart0 + ((rand1 - 0.5f) * o1) * P.pixelSize) + ((rand2 - 0.5f) * o2) * P.pixelSize;
    rdir = rstart - f3(P.cameraPos[0], P.cameraPos[1], P.cameraPos[2]);
}

enum { PATH_ENDED = 0, PATH_CONTINUES = 1 };

/// Outcome of shading one segment.
struct ShadeResult {
    int next;          ///< PATH_ENDED / PATH_CONTINUES (another segment follows)
    bool broke;        ///< the GLSL loop was left by `break` (sky or emissive sphere)
    bool want_shadow;  ///< the reference runs a Sun shadow query from `rstart` here
    bool sun_matters;  ///< ... and its result can change pathColor (dot(sun, n) > 0)
    F3 sun_term;       ///< dot(sun, n) * albedo, to be added to pathColor if the Sun is visible
    bool ush, specular;
};

/// path_tracing.glsl:182-233 for segment i (0-based), given the finished closest-hit query of ray r.
GD_FN ShadeResult path_shade(const Scene &sc, const gpuart_params &P, float4 seed, int i, const Ray &r, float closest,
                             uint32_t prim, F3 &rstart, F3 &rdir, F3 &cw, F3 &pathColor) {
    const float FUZZY_ANGLE = 10 * 3.14159f / 180;
    ShadeResult o;
    o.next = PATH_ENDED; o.broke = false; o.want_shadow = false; o.sun_matters = false; o.sun_term = f3(0, 0, 0); o.specular = false;
    Surface h;
    resolve_hit(sc, r, closest, prim, P.userSphere, h, o.ush);
    int ptype = h.ptype;
    if (o.ush) {
        if (P.userSphereFlags & 1u) {
            pathColor = pathColor + f3(P.userSphereEm[0], P.userSphereEm[1], P.userSphereEm[2]) * cw;
            o.broke = true;
            return o;
        }
        ptype = P_SPHERE;
    } else if (ptype == -1) {
        pathColor = pathColor + (2.0f * sky_color(r.d, P.sunDirAlt)) * cw;
        o.broke = true;
        return o;
    }
    F3 albedo = primitive_color(ptype);
    cw = cw * albedo;
    rstart = h.p;
    F3 seed3 = f3(seed.x, seed.y, seed.z);
    if (o.ush && (P.userSphereFlags & 2u)) {
        if (!(P.userSphereFlags & 4u)) rdir = reflect3(r.d, h.n);
        else rdir = random_direction_inside_cone(reflect3(r.d, h.n), h.n, FUZZY_ANGLE, h.p + seed3);
        o.specular = true;
    } else {
        rdir = random_hemisphere_direction(h.n, h.p + seed3);
    }
    if (P.sunEnabled == 1 && !o.specular) {
        float dotp = dot3(f3(P.sunDirAlt[0], P.sunDirAlt[1], P.sunDirAlt[2]), h.n);
        // The reference issues the shadow query unconditionally; when dotp <= 0 its result cannot
        // change pathColor, so the fast mode skips it (sun_matters); reference-work mode still runs it.
        o.want_shadow = true;
        o.sun_matters = dotp > 0;
        o.sun_term = o.sun_matters ? dotp * albedo : f3(0, 0, 0);
    }
    // loop header of the next iteration (path_tracing.glsl:177)
    if (i + 1 < P.maxSegments && (cw.x > P.minWeight && cw.y > P.minWeight && cw.z > P.minWeight)) o.next = PATH_CONTINUES;
    return o;
}

/// path_tracing.glsl:247-252: value added to the pixel's colour when a path ends.
/// `i` = value of the GLSL loop counter when the loop was left.
GD_FN F3 path_finish(const gpuart_params &P, F3 rdir0, int i, bool ush, bool specular, F3 pathColor) {
    if (i == 0 && !ush) return sky_color(rdir0, P.sunDirAlt);
    if (i == 0 && ush && !specular) return f3(1, 1, 1);
    return pathColor;
}

// ---- the same loop in one thread (megakernel form; kept as an independent cross-check / ablation) ----
template <bool REFWORK>
GD_FN F3 path_tracing_pixel(const Scene &sc, const gpuart_params &P, float4 seed, int npaths, F3 rstart0, F3 rdir0,
                            TravStack &st, WorkCounters *wc, uint32_t &segments) {
    F3 sun = f3(P.sunDirAlt[0], P.sunDirAlt[1], P.sunDirAlt[2]);
    F3 color = f3(0, 0, 0);
    for (int j = 0; j < npaths; j++) {
        F3 rstart, rdir;
        path_begin(P, seed, j, rstart0, rdir0, rstart, rdir);
        F3 pathColor = f3(0, 0, 0), cw = f3(1, 1, 1);
        bool ush = false, specular = false;
        int i = 0;
        if (P.maxSegments > 0 && 1.0f > P.minWeight)
            for (;;) {
                Ray r; r.o = rstart; r.d = rdir;
                float closest; uint32_t prim;
                traverse<false, REFWORK>(sc, r, st, closest, prim, wc);
                segments++;
                ShadeResult s = path_shade(sc, P, seed, i, r, closest, prim, rstart, rdir, cw, pathColor);
                ush = s.ush; specular = s.specular;
                if (s.broke) break;
                if (s.want_shadow && (REFWORK || s.sun_matters)) {
                    Ray sr; sr.o = rstart; sr.d = sun;
                    float sclosest; uint32_t sprim;
                    traverse<!REFWORK, REFWORK>(sc, sr, st, sclosest, sprim, wc);
                    if (sun_visible(P, rstart, sun, sprim)) pathColor = pathColor + s.sun_term;
                }
                i++;
                if (s.next != PATH_CONTINUES) break;
            }
        color = color + path_finish(P, rdir0, i, ush, specular, pathColor);
    }
    return color;
}

}  // namespace gd
