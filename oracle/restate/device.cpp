// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md). Never linked into the product.
// CPU restatement of the reference GLSL; see device.h. Build with -ffp-contract=off.
#include "device.h"

namespace orc {

static const float VISIBILITY_OFFSET = 1.0e-4f;

static inline V3 q3(const float *tree, int addr) { return V3{tree[4 * addr], tree[4 * addr + 1], tree[4 * addr + 2]}; }
static inline V4 q4(const float *tree, int addr) {
    return V4{tree[4 * addr], tree[4 * addr + 1], tree[4 * addr + 2], tree[4 * addr + 3]};
}
static inline V3 xyz(V4 v) { return V3{v.x, v.y, v.z}; }

// ---- noise.glsl --------------------------------------------------------------------------
// noise.glsl:13-22 — one round of Jenkins one-at-a-time.
uint32_t hash1(uint32_t x) {
    x += (x << 10);
    x ^= (x >> 6);
    x += (x << 3);
    x ^= (x >> 11);
    x += (x << 15);
    return x;
}
// noise.glsl:31-38
static inline float floatConstruct(uint32_t m) { return u2f((m & 0x007FFFFFu) | 0x3F800000u) - 1.0f; }
// noise.glsl:43-46 with the compound hashes of :25-27
float random1(float x) { return floatConstruct(hash1(f2u(x))); }
float random2(float x, float y) { return floatConstruct(hash1(f2u(x) ^ hash1(f2u(y)))); }
float random3(V3 v) { return floatConstruct(hash1(f2u(v.x) ^ hash1(f2u(v.y)) ^ hash1(f2u(v.z)))); }
float random4(V4 v) {
    return floatConstruct(hash1(f2u(v.x) ^ hash1(f2u(v.y)) ^ hash1(f2u(v.z)) ^ hash1(f2u(v.w))));
}

// ---- common.glsl -------------------------------------------------------------------------
// common.glsl:40-46
// As Mesa compiles it: all(lessThan(a, b)) becomes !any(a >= b) (NIR pushes the negation into the comparison), so a NaN
// component counts as "small"; the z component of normalize(vec3(v.y, -v.x, 0)) is the constant 0, not 0 * rsq.
V3 GetOrthogonal(V3 v) {
    if (!(fabsf(v.x) >= 1.0e-6f || fabsf(v.y) >= 1.0e-6f)) return V3{1, 0, 0};
    const float inv = rsq(dot3(V3{v.y, -v.x, 0.0f}, V3{v.y, -v.x, 0.0f}));
    return V3{v.y * inv, -v.x * inv, 0.0f};
}

// common.glsl:49-66
V3 GetRandomHemisphereDirection(V3 v, V3 ri) {
    const float PIDBL = 3.1415926f * 2;
    float _2pr1 = PIDBL * random3(ri);
    float r2 = random3(V3{ri.z, ri.x, ri.y});
    float sr2 = sqrtf(1.0f - r2);
    float s, c;
    sincos_lp(_2pr1, &s, &c);
    float x = c * sr2, y = s * sr2, z = sqrtf(r2);
    V3 t = GetOrthogonal(v);
    // tangent.z is the constant 0 in both branches of GetOrthogonal: Mesa folds its products away (0 * NaN and 0 * inf included)
    V3 ct{-(v.z * t.y), v.z * t.x, v.x * t.y - v.y * t.x};
    return V3{(t.x * x + ct.x * y) + v.x * z, (t.y * x + ct.y * y) + v.y * z, ct.z * y + v.z * z};
}

// common.glsl:69-76 — GLSL mat3 constructor is column-major; M*v = (col0*v.x + col1*v.y) + col2*v.z.
V3 rotate(V3 v, V3 a, float sine, float cosine) {
    float omc = 1 - cosine;
    V3 c0{(a.x * a.x + (1 - a.x * a.x) * cosine), (a.x * a.y * omc - a.z * sine), (a.x * a.z * omc + a.y * sine)};
    V3 c1{(a.x * a.y * omc + a.z * sine), (a.y * a.y + (1 - a.y * a.y) * cosine), (a.y * a.z * omc - a.x * sine)};
    V3 c2{(a.x * a.z * omc - a.y * sine), (a.y * a.z * omc + a.x * sine), (a.z * a.z + (1 - a.z * a.z) * cosine)};
    return V3{(c0.x * v.x + c1.x * v.y) + c2.x * v.z, (c0.y * v.x + c1.y * v.y) + c2.y * v.z,
              (c0.z * v.x + c1.z * v.y) + c2.z * v.z};
}

// common.glsl:81-106 (including the reference's `sina = 1-cosa*cosa`)
V3 GetRandomDirectionInsideCone(V3 v, V3 normal, float halfAngle, V3 ri) {
    float a = random2(ri.x, ri.y) * halfAngle;
    float b = random2(ri.y, ri.z) * 2 * 3.14159f;
    float sina, cosa, sinb, cosb;
    sincos_lp(a, &sina, &cosa);
    sincos_lp(b, &sinb, &cosb);
    float sinc = length3(cross3(-v, normal)) / length3(v);
    if (cosa < sinc) {
        cosa = sinc;
        sina = 1 - cosa * cosa;
    }
    V3 vo = GetOrthogonal(v);
    V3 w = normalize3(rotate(v, vo, sina, cosa));
    return rotate(w, v, sinb, cosb);
}

// ---- sphere.glsl:31-70 -------------------------------------------------------------------
void SphereIntersection(V3 rs, V3 rd, V3 center, float radius, float &pos, V3 &p, V3 &n) {
    V3 m = rs - center;
    float a = dot3(rd, rd);
    float b = 2 * dot3(rd, m);
    float c = dot3(m, m) - radius * radius;
    float delta = b * b - 4 * a * c;
    if (delta >= 0) {
        float sd = sqrtf(delta);
        float k1 = (-b + sd) / (a + a);
        float k2 = (-b - sd) / (a + a);
        if (k1 < VISIBILITY_OFFSET)
            pos = k2;
        else if (k2 < VISIBILITY_OFFSET)
            pos = k1;
        else
            pos = (k1 < k2 ? k1 : k2);
        p = rs + rd * pos;
        n = normalize3(p - center);
        if (dot3(rs - p, n) < 0) n = -n;
    } else
        pos = -1;
}

// ---- disc.glsl:30-72 ---------------------------------------------------------------------
void DiscIntersection(V3 rs, V3 rd, V3 center, float radius, V3 dn, float &pos, V3 &p, V3 &n) {
    float tmp = dot3(rd, dn);
    if (fabsf(tmp) < 1.0e-8f) { pos = -1; return; }
    float k = dot3(dn, center - rs) / tmp;
    if (k <= 0) { pos = -1; return; }
    V3 q = k * rd + rs;
    V3 d = q - center;
    if (dot3(d, d) <= radius * radius) {
        pos = k;
        p = rs + k * rd;
        if (dot3(rs - center, dn) > 0) n = dn; else n = -dn;
    } else
        pos = -1;
}

// ---- triangle.glsl:33-82 -----------------------------------------------------------------
void TriangleIntersection(V3 rs, V3 rd, V3 v0, V3 v1, V3 v2, float &pos, V3 &p, V3 &n) {
    V3 edge1 = v1 - v0, edge2 = v2 - v0;
    V3 pvec = cross3(rd, edge2);
    float det = dot3(edge1, pvec);
    if (fabsf(det) < 1.0e-10f) { pos = -1; return; }
    float invDet = 1 / det;
    V3 tvec = rs - v0;
    float u = dot3(tvec, pvec) * invDet;
    if (u < 0 || u > 1) { pos = -1; return; }
    V3 qvec = cross3(tvec, edge1);
    float v = dot3(rd, qvec) * invDet;
    if (v < 0 || (dot3(tvec, pvec) + dot3(rd, qvec)) * invDet > 1)
        pos = -1;
    else {
        pos = dot3(edge2, qvec) * invDet;
        p = rs + rd * pos;
        n = normalize3(cross3(edge1, edge2));
        if (dot3(rs - p, n) < 0) n = -n;
    }
}

// ---- cone.glsl:30-135 --------------------------------------------------------------------
void ConeIntersection(V3 rs, V3 rd, V4 cr1, V4 cr2, V4 axL, float widthCoeff, float cosB, float dotAxC1,
                      float &pos, V3 &p, V3 &n) {
    (void)cr2;
    const float CONE_TOLERANCE = 1.0e-7f;
    V3 ax = xyz(axL), c1 = xyz(cr1);
    float axd = dot3(ax, rd), axs = dot3(ax, rs);
    V3 D = axd * ax;
    V3 E = -rd;
    V3 F = ((c1 + axs * ax) - dotAxC1 * ax) - rs;
    float G = widthCoeff * axd;
    float H = (widthCoeff * axs + cr1.w) - widthCoeff * dotAxC1;  // llvmpipe/NIR evaluation order (verified on goldens)
    float A = ((dot3(D, D) + dot3(E, E)) + 2 * dot3(D, E)) - G * G;
    float B = 2 * dot3(F, D + E) - 2 * G * H;
    float C = dot3(F, F) - H * H;
    if (fabsf(A) < CONE_TOLERANCE) { pos = -1; return; }
    float delta = B * B - 4 * A * C;
    if (delta < CONE_TOLERANCE) { pos = -1; return; }
    float sq = sqrtf(delta);
    float k1 = (-B + sq) / (A + A);
    float k2 = (-B - sq) / (A + A);
    V3 p1 = rs + k1 * rd, p2 = rs + k2 * rd;
    float t1 = dot3(ax, p1 - c1), t2 = dot3(ax, p2 - c1);
    bool on1 = t1 >= 0 && t1 <= axL.w;
    bool on2 = t2 >= 0 && t2 <= axL.w;
    if (k1 < VISIBILITY_OFFSET && on2) { pos = k2; p = p2; }
    else if (k2 < VISIBILITY_OFFSET && on1) { pos = k1; p = p1; }
    else {
        if ((k1 < k2 && on1 && on2) || (on1 && !on2)) { pos = k1; p = p1; }
        else if ((k2 < k1 && on1 && on2) || (!on1 && on2)) { pos = k2; p = p2; }
        else { pos = -1; return; }
    }
    if (pos > 0) {
        V3 proj = c1 + dot3(ax, p - c1) * ax;
        V3 n1 = normalize3(p - proj);
        float u = cosB - dot3(n1, ax);
        n = normalize3(u * ax + n1);
        if (dot3(n, rd) > 0) n = -n;
    }
}

// ---- bvh_intersection.glsl:125-223 -------------------------------------------------------
static int CheckBVHPrimitiveIntersection(V3 rs, V3 rd, int ptype, const float *tree, int addr, float &pos, V3 &p,
                                         V3 &n) {
    if (ptype == P_CONE) {
        V4 params = q4(tree, addr + 3);
        ConeIntersection(rs, rd, q4(tree, addr), q4(tree, addr + 1), q4(tree, addr + 2), params.x, params.y, params.z,
                         pos, p, n);
        if (pos < VISIBILITY_OFFSET) pos = -1;
        return addr + 4;
    } else if (ptype == P_SPHERE) {
        V4 s = q4(tree, addr);
        SphereIntersection(rs, rd, xyz(s), s.w, pos, p, n);
        if (pos < VISIBILITY_OFFSET) pos = -1;
        return addr + 1;
    } else if (ptype == P_DISC) {
        V4 d = q4(tree, addr);
        DiscIntersection(rs, rd, xyz(d), d.w, q3(tree, addr + 1), pos, p, n);
        if (pos < VISIBILITY_OFFSET) pos = -1;
        return addr + 2;
    } else if (ptype == P_TRIANGLE) {
        TriangleIntersection(rs, rd, q3(tree, addr), q3(tree, addr + 1), q3(tree, addr + 2), pos, p, n);
        if (pos < VISIBILITY_OFFSET) pos = -1;
        return addr + 3;
    }
    pos = -1;  // undefined in the reference (falls off the end of a non-void function)
    return addr;
}

// ---- bvh_intersection.glsl:229-354 -------------------------------------------------------
bool IntersectsAABB(V3 rs, V3 rd, V3 rdiv, const float *tree, int addr, float &pos) {
    V3 bmin = q3(tree, addr), bmax = q3(tree, addr + 1);
    if (rs.x >= bmin.x && rs.y >= bmin.y && rs.z >= bmin.z && rs.x <= bmax.x && rs.y <= bmax.y && rs.z <= bmax.z) {
        pos = -1;
        return true;
    }
    bool hit = false;
    pos = 1.0e+19f;
    float k, x, y, z;
    if (rd.x != 0) {
        k = (bmin.x - rs.x) * rdiv.x;
        if (k >= 0) {
            y = rs.y + k * rd.y; z = rs.z + k * rd.z;
            if (y >= bmin.y && y <= bmax.y && z >= bmin.z && z <= bmax.z) { hit = true; if (k < pos) pos = k; }
        }
        k = (bmax.x - rs.x) * rdiv.x;
        if (k >= 0) {
            y = rs.y + k * rd.y; z = rs.z + k * rd.z;
            if (y >= bmin.y && y <= bmax.y && z >= bmin.z && z <= bmax.z) { hit = true; if (k < pos) pos = k; }
        }
    }
    if (rd.y != 0) {
        k = (bmin.y - rs.y) * rdiv.y;
        if (k >= 0) {
            x = rs.x + k * rd.x; z = rs.z + k * rd.z;
            if (x >= bmin.x && x <= bmax.x && z >= bmin.z && z <= bmax.z) { hit = true; if (k < pos) pos = k; }
        }
        k = (bmax.y - rs.y) * rdiv.y;
        if (k >= 0) {
            x = rs.x + k * rd.x; z = rs.z + k * rd.z;
            if (x >= bmin.x && x <= bmax.x && z >= bmin.z && z <= bmax.z) { hit = true; if (k < pos) pos = k; }
        }
    }
    if (rd.z != 0) {
        k = (bmin.z - rs.z) * rdiv.z;
        if (k >= 0) {
            x = rs.x + k * rd.x; y = rs.y + k * rd.y;
            if (x >= bmin.x && x <= bmax.x && y >= bmin.y && y <= bmax.y) { hit = true; if (k < pos) pos = k; }
        }
        k = (bmax.z - rs.z) * rdiv.z;
        if (k >= 0) {
            x = rs.x + k * rd.x; y = rs.y + k * rd.y;
            if (x >= bmin.x && x <= bmax.x && y >= bmin.y && y <= bmax.y) { hit = true; if (k < pos) pos = k; }
        }
    }
    return hit;
}

// ---- bvh_intersection.glsl:360-457 — stackless DFS with parent pointers -------------------
void CheckBVHIntersection(V3 rs, V3 rd, const float *tree, Hit &h, TravStats *st) {
    enum { FROM_NONE = 0, FROM_LO = 1, FROM_HI = 2 };
    int bvhIdx = 0;
    bool recursionReturn = false;
    int returningFrom = FROM_NONE;
    V3 rdiv{1 / rd.x, 1 / rd.y, 1 / rd.z};
    h.pos = -1;
    h.ptype = -1;
    float closestPos = 1e+19f;
    if (st) st->rays++;
    for (;;) {
        const float *info = tree + 4 * (bvhIdx + 2);
        uint32_t flags = f2u(info[0]);
        if (recursionReturn && returningFrom == FROM_HI && (flags & BVH_IS_ROOT)) break;
        if (st) { st->iterations++; if (!recursionReturn) st->nodes++; }
        float boxPos;
        if (IntersectsAABB(rs, rd, rdiv, tree, bvhIdx, boxPos)) {
            if (boxPos > closestPos)
                recursionReturn = true;
            else if (flags & BVH_LEAF) {
                uint32_t nprims = flags & ~BVH_FLAGS_MASK;
                int primAddr = bvhIdx + 3;
                for (uint32_t i = 0; i < nprims; i++) {
                    float cpos; V3 cp{0, 0, 0}, cn{0, 0, 0};
                    int ptype = (int)f2u(tree[4 * primAddr]);
                    if (st && ptype >= 0 && ptype < 4) st->prim_tests[ptype]++;
                    primAddr = CheckBVHPrimitiveIntersection(rs, rd, ptype, tree, primAddr + 1, cpos, cp, cn);
                    if (cpos > 0 && cpos < closestPos) {
                        closestPos = h.pos = cpos;
                        h.p = cp; h.n = cn; h.ptype = ptype;
                    }
                }
                recursionReturn = true;
            } else {
                recursionReturn = false;
                if (returningFrom == FROM_NONE)
                    bvhIdx = (int)f2u(info[1]);
                else if (returningFrom == FROM_LO) {
                    returningFrom = FROM_NONE;
                    bvhIdx = (int)f2u(info[2]);
                } else
                    recursionReturn = true;
            }
        } else
            recursionReturn = true;
        if (recursionReturn) {
            returningFrom = (flags & BVH_IS_LOWER) ? FROM_LO : FROM_HI;
            bvhIdx = (int)f2u(info[3]);
        }
    }
}

// ---- intersection.glsl:71-111 ------------------------------------------------------------
void CheckIntersectionInclUserSphere(V3 rs, V3 rd, const float *tree, const float us[4], Hit &h, bool &userSphereHit,
                                     TravStats *st) {
    CheckBVHIntersection(rs, rd, tree, h, st);
    float usPos; V3 usP{0, 0, 0}, usN{0, 0, 0};
    SphereIntersection(rs, rd, V3{us[0], us[1], us[2]}, us[3], usPos, usP, usN);
    if (usPos > VISIBILITY_OFFSET && (h.pos < 0 || usPos < h.pos)) {
        userSphereHit = true;
        h.ptype = P_SPHERE; h.pos = usPos; h.p = usP; h.n = usN;
    } else
        userSphereHit = false;
}

// ---- sky.glsl:34-60 ----------------------------------------------------------------------
V3 GetSkyColor(V3 dir, const float sda[4]) {
    V3 nd = normalize3(dir);
    // cross(cross((0,0,1), nd), (0,0,1)) = (nd.x, nd.y, 0)
    V3 hp{nd.x, nd.y, 0.0f};
    float weight;
    if (dir.z >= 0) {
        // hp.z is the constant 0: Mesa folds 0 * rsq(...) and nd.z * 0 away (also where rsq is inf: |hp| underflows for a direction
        // close to the zenith), so the z term never makes a NaN; what is left of dot() is its (y + x) part
        const float inv = rsq(dot3(hp, hp));
        weight = nd.y * (hp.y * inv) + nd.x * (hp.x * inv);
    } else
        weight = 1;
    float sunWeight = 1.0f - sda[3] / (3.1415926f / 2);
    const V3 ZH{0.2f, 0.6f, 1}, ZL{0, 0.2f, 0.5f}, HH{1, 1, 1}, HL{1, 0.647f, 0.367f};
    V3 cz{mixf(ZH.x, ZL.x, sunWeight), mixf(ZH.y, ZL.y, sunWeight), mixf(ZH.z, ZL.z, sunWeight)};
    V3 ch{mixf(HH.x, HL.x, sunWeight), mixf(HH.y, HL.y, sunWeight), mixf(HH.z, HL.z, sunWeight)};
    float pw = pow_lp(weight, 16.0f);
    // ch.x folds to the constant 1.0, and NIR lowers flrp(x, 1.0, t) as x*(1-t) + t (verified on goldens).
    (void)ch.x;
    return V3{cz.x * (1.0f - pw) + pw, mixf(cz.y, ch.y, pw), mixf(cz.z, ch.z, pw)};
}

// ---- vertex.glsl:29-37 + GL::Utils::DrawFullscreenQuad (src/gl_utils.cpp:225-254) ------------
// The interpolated UV of a fragment is NOT exactly ((x+.5)/W, (y+.5)/H) on llvmpipe unless W, H are
// powers of two. It is the plane equation llvmpipe's triangle setup builds for each of the two fan
// triangles, evaluated with two fused multiply-adds per channel (verified bit-for-bit on sizes
// from 1x1 to 3840x2160 by tests/golden/make_golden.py `uv`):
//   quad corners V0=(0,0) uv(0,0), V1=(W,0) uv(1,0), V2=(W,H) uv(1,1), V3=(0,H) uv(0,1);
//   triangle A is set up in vertex order (V1,V0,V2), triangle B in order (V2,V0,V3);
//   a pixel centre on or below the V0-V2 diagonal belongs to A.
static void PlaneCoef(float x0, float y0, float x1, float y1, float x2, float y2, float a0, float a1, float a2,
                      float c[3]) {
    float x0c = x0 - 0.5f, y0c = y0 - 0.5f;
    float dx01 = x0 - x1, dy01 = y0 - y1, dx20 = x2 - x0, dy20 = y2 - y0;
    float e = dx01 * dy20, f = dy01 * dx20;
    float ooa = 1.0f / (e - f);
    float dy20o = dy20 * ooa, dy01o = dy01 * ooa, dx20o = dx20 * ooa, dx01o = dx01 * ooa;
    float da01 = a0 - a1, da20 = a2 - a0;
    float dadx = da01 * dy20o - da20 * dy01o;
    float dady = da20 * dx01o - da01 * dx20o;
    c[0] = a0 - (dadx * x0c + dady * y0c);
    c[1] = dadx;
    c[2] = dady;
}

void QuadUVCoefs(int W, int H, float coef[12]) {
    float w = (float)W, h = (float)H;
    PlaneCoef(w, 0, 0, 0, w, h, 1, 0, 1, coef + 0);   // A.u  (V1,V0,V2)
    PlaneCoef(w, 0, 0, 0, w, h, 0, 0, 1, coef + 3);   // A.v
    PlaneCoef(w, h, 0, 0, 0, h, 1, 0, 0, coef + 6);   // B.u  (V2,V0,V3)
    PlaneCoef(w, h, 0, 0, 0, h, 1, 0, 1, coef + 9);   // B.v
}

void PixelUV(int x, int y, int W, int H, const float coef[12], float &u, float &v) {
    const float *c = ((long long)(2 * y + 1) * W - (long long)(2 * x + 1) * H <= 0) ? coef : coef + 6;
    float X = (float)x, Y = (float)y;
    u = fmaf(c[2], Y, fmaf(c[1], X, c[0]));
    v = fmaf(c[5], Y, fmaf(c[4], X, c[3]));
}

// ---- cam_init.glsl:45-50 -------------------------------------------------------------------
void CamInitPixel(int x, int y, int W, int H, const float pos[3], const float bl[3], const float dh[3],
                  const float dv[3], V3 &rstart, V3 &rdir) {
    float coef[12], u, v;
    QuadUVCoefs(W, H, coef);
    PixelUV(x, y, W, H, coef, u, v);
    V3 start{(bl[0] + dh[0] * u) + dv[0] * v, (bl[1] + dh[1] * u) + dv[1] * v, (bl[2] + dh[2] * u) + dv[2] * v};
    rstart = start;
    rdir = normalize3(start - V3{pos[0], pos[1], pos[2]});
}

static const V3 PRIMITIVE_COLOR[4] = {{0.65f, 0.4f, 0.35f}, {0.1f, 0.2f, 0.1f}, {0.3f, 0.3f, 0.3f}, {0.3f, 0.3f, 0.3f}};

static inline V3 reflect3(V3 I, V3 N) { return I - N * (2 * dot3(N, I)); }

// direct_lighting.glsl:89-100
static V3 Lambert(V3 lightDir, V3 normal, V3 diffuse, float intensity) {
    float dotp = dot3(lightDir, normal);
    if (dotp > 0) return (diffuse * intensity) * dotp;
    return V3{0, 0, 0};
}

// ---- direct_lighting.glsl:134-207 --------------------------------------------------------
V3 DirectLightingPixel(V3 rstart, V3 rdir, const float *tree, const Params &P, TravStats *st) {
    const float AMBIENT = 0.15f;
    V3 sun{P.sunDirAlt[0], P.sunDirAlt[1], P.sunDirAlt[2]};
    V3 usc{P.userSphere[0], P.userSphere[1], P.userSphere[2]};
    V3 cw{1, 1, 1}, out{0, 0, 0};
    for (int i = 0; i <= 1; i++) {
        Hit h; bool ush;
        CheckIntersectionInclUserSphere(rstart, rdir, tree, P.userSphere, h, ush, st);
        if ((P.userSphereFlags & USPH_SPECULAR) && ush) {
            rstart = h.p;
            rdir = reflect3(rdir, h.n);
            cw = cw * PRIMITIVE_COLOR[P_SPHERE];
        } else if ((P.userSphereFlags & USPH_EM_NONZERO) && ush) {
            out = V3{1, 1, 1};
        } else {
            if (h.ptype != -1) {
                V3 diffuse = PRIMITIVE_COLOR[h.ptype] * cw;
                if (P.sunEnabled == 1) {
                    Hit sh; bool d;
                    CheckIntersectionInclUserSphere(h.p, sun, tree, P.userSphere, sh, d, st);
                    if (sh.ptype == -1) out = out + Lambert(sun, h.n, diffuse, 1.0f);
                }
                if (P.userSphereFlags & USPH_EM_NONZERO) {
                    V3 dts = usc - h.p;
                    float dist = length3(dts);
                    V3 dn{dts.x / dist, dts.y / dist, dts.z / dist};
                    Hit sh;
                    CheckBVHIntersection(h.p, dn, tree, sh, st);
                    if (sh.ptype == -1 || sh.pos > dist) {
                        V3 l = Lambert(dn, h.n, diffuse, 1);
                        float d2 = dot3(dts, dts);  // NIR folds sqrt(a)*sqrt(a) -> |a| (verified on goldens)
                        out = out + V3{l.x / d2, l.y / d2, l.z / d2};
                    }
                }
                out = out + AMBIENT * diffuse;
            } else {
                out = cw * GetSkyColor(rdir, P.sunDirAlt);
            }
            break;
        }
    }
    return out;
}

// ---- path_tracing.glsl:133-256 -----------------------------------------------------------
// path_tracing.glsl:141-175: the jittered ray of path j's first segment (also exported for the tests that aim a camera ray: capi.cpp)
void FirstSegmentRay(V3 rstart0, V3 rdir0, const Params &P, const float rsd[4], int j, V3 &rstart, V3 &rdir) {
    V3 camPos{P.cameraPos[0], P.cameraPos[1], P.cameraPos[2]};
    V3 o1;
    if (fabsf(rdir0.x) > 1.0e-5f || fabsf(rdir0.y) > 1.0e-5f)
        o1 = normalize3(V3{rdir0.y, -rdir0.x, 0});
    else
        o1 = normalize3(V3{0, -rdir0.z, rdir0.y});
    V3 o2 = cross3(normalize3(rdir0), o1);
    float rand1 = random1(rsd[0] + (float)j);
    float rand2 = random1(rsd[1] + (float)j);
    rstart = (rstart0 + ((rand1 - 0.5f) * o1) * P.pixelSize) + ((rand2 - 0.5f) * o2) * P.pixelSize;
    rdir = rstart - camPos;
}

V3 PathTracingPixel(V3 rstart0, V3 rdir0, const float *tree, const Params &P, const float rsd[4], int npaths,
                    TravStats *st, uint64_t *nseg) {
    const float FUZZY_ANGLE = 10 * 3.14159f / 180;
    V3 sun{P.sunDirAlt[0], P.sunDirAlt[1], P.sunDirAlt[2]};
    V3 seed{rsd[0], rsd[1], rsd[2]};
    V3 color{0, 0, 0};
    for (int j = 0; j < npaths; j++) {
        V3 rstart, rdir;
        FirstSegmentRay(rstart0, rdir0, P, rsd, j, rstart, rdir);
        V3 pathColor{0, 0, 0}, cw{1, 1, 1};
        bool ush = false, specular = false;
        int i;
        for (i = 0; i < P.maxSegments && (cw.x > P.minWeight && cw.y > P.minWeight && cw.z > P.minWeight); i++) {
            Hit h;
            CheckIntersectionInclUserSphere(rstart, rdir, tree, P.userSphere, h, ush, st);
            if (nseg) (*nseg)++;
            int ptype = h.ptype;
            if (ush) {
                if (P.userSphereFlags & USPH_EM_NONZERO) {
                    V3 em{P.userSphereEm[0], P.userSphereEm[1], P.userSphereEm[2]};
                    pathColor = pathColor + em * cw;
                    break;
                } else
                    ptype = P_SPHERE;
            } else if (ptype == -1) {
                V3 sky = GetSkyColor(rdir, P.sunDirAlt);
                pathColor = pathColor + (2.0f * sky) * cw;
                break;
            }
            cw = cw * PRIMITIVE_COLOR[ptype];
            rstart = h.p;
            if (ush && (P.userSphereFlags & USPH_SPECULAR)) {
                if (!(P.userSphereFlags & USPH_FUZZY))
                    rdir = reflect3(rdir, h.n);
                else
                    rdir = GetRandomDirectionInsideCone(reflect3(rdir, h.n), h.n, FUZZY_ANGLE, h.p + seed);
                specular = true;
            } else {
                rdir = GetRandomHemisphereDirection(h.n, h.p + seed);
                specular = false;
            }
            if (P.sunEnabled == 1 && !specular) {
                Hit sh;
                CheckIntersectionInclUserSphere(h.p, sun, tree, P.userSphere, sh, ush, st);
                if (sh.ptype == -1) {
                    float dotp = dot3(sun, h.n);
                    if (dotp > 0) pathColor = pathColor + dotp * PRIMITIVE_COLOR[ptype];
                }
            }
        }
        if (i == 0 && !ush)
            pathColor = GetSkyColor(rdir0, P.sunDirAlt);
        else if (i == 0 && ush && !specular)
            pathColor = V3{1, 1, 1};
        color = color + pathColor;
    }
    return color;
}

}  // namespace orc
