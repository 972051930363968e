// librt_amd: the per-pixel render path as one CDNA4 (gfx950) kernel, behind the C ABI of
// include/rt_abi.h.
//
// Execution model
//   * one lane = one pixel (or one caller-supplied ray), one wavefront = one 8x8 pixel tile, one
//     wavefront per workgroup; no barriers, no matrix cores: the path has no dense contraction, it
//     is scalar f64 arithmetic per ray.  LDS only holds finished pixels until the tile is stored.
//   * scene traversal is wave-synchronous: the wavefront walks an array of 128-byte traversal
//     records (TRec, the node table depth first) with ONE uniform index.  Records are therefore
//     fetched with scalar loads (constant address space) into SGPRs and cost no vector registers or
//     vector memory traffic.  Each lane only keeps the index up to which it is masked off
//     (`skip_until`), set when its ray misses a group's bounding box; when a wave-wide ballot
//     shows that no lane is inside a group the whole subtree is skipped with the skip link.
//   * what cannot change the result is not computed: leaves behind padded boxes the ray misses,
//     subtrees whose box times cannot count, shadow rays whose outcome the colour does not depend
//     on, the container bookkeeping for hits that are not transparent.
//   * recursion (reflection, refraction) is an explicit per-lane frame stack in scratch memory;
//     results are combined in the reference's summation order so colours are bit-identical.
//   * tiles are rendered expensive ones first (tile_class_kernel), so the frame does not end on
//     a few long wavefronts.
//
// Arithmetic contract: IEEE f64, reference operation order, no FMA contraction (the file is
// compiled with -ffp-contract=off and carries the pragma below), correctly rounded / and sqrt.
// pow and cos come from the device math library and may differ from the host libm in the last ulp.
#include <hip/hip_runtime.h>

#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "rt_abi.h"
#include "rt_download.h"

#pragma clang fp contract(off)

#define RT_CONST __attribute__((address_space(4)))
#define DI __device__ __forceinline__
#define RT_EPS 0.00001  // src/lib.rs:22
#define RT_MAX_FRAMES RT_MAX_DEPTH
#define RT_PAT_STACK (RT_MAX_PATTERN_NEST + 1)
#define COUNTER_SHARDS 1024u  // each shard is 16 u64 (one 128-byte line)
#define COUNTER_WORDS (COUNTER_SHARDS * 16u)
// One more line per set: tile counts per cost class for the dispatch order (words 0..2).
#define COUNTER_SET_WORDS (COUNTER_WORDS + 64u)

namespace {

// Traversal record: everything one step of the scene walk needs, in one 128-byte line, so that a
// step costs a single scalar-load round trip.  Records follow the depth-first order of the node
// table, except that the boxed children of one group may be sorted in space (the walk breaks ties
// between equal times by node index, so their order does not show); a node whose transform is not
// the identity AND which carries its own payload (a transformed group or triangle) is preceded by
// an extra T_XFORM record that holds its inverse.
// T_CULL records hold a padded box in the space the ray arrives in, in front of a finite sphere,
// cube or cylinder leaf or of a set of siblings (`skip` = the first record after what the box
// covers).  Lanes whose ray cannot reach the box skip that part; this never changes a result, it
// only avoids arithmetic whose outcome is "no hit".
// A CSG shape (csg.rs) is bracketed by T_CSG_BEGIN (its box, tested like a group's; `skip` = the
// record after the matching end record) and T_CSG_END (`node` = the CSG node, `skip` = the NODE
// index where its right operand starts, the operation in bits 12-13).  Between them the leaves
// collect their intersections per lane instead of competing for the nearest hit; the end record
// sorts and filters them.
enum { T_XFORM = 8, T_CULL = 9, T_CSG_BEGIN = 10, T_CSG_END = 11, T_HAS_INV = 0x100, T_CLOSED = 0x200,
       T_OWN_XFORM = 0x400, T_LINE_ONLY = 0x800, T_CSG_OP_SHIFT = 12, T_DIAG = 0x4000, T_LEVELS_SHIFT = 16 };
struct alignas(16) TRec {
    int kind;    // low byte: RT_* type or T_XFORM; T_HAS_INV: p holds rows 0..2 of the inverse; T_CLOSED;
                 // T_DIAG: that inverse is a scale and a translation (its six off-diagonal entries are zeros);
                 // T_OWN_XFORM: the record before this one is this node's T_XFORM record;
                 // T_LINE_ONLY (groups): hits below can lie outside the box, test the ray's line only;
                 // bits 16-17: how many untransformed levels (0, 1, 2 or more) lie between the space the walk keeps
                 // the ray in and the space this record tests or transforms it in (see through_identity)
    int skip;    // first record after this node's subtree
    int ctx;     // record holding the inverse of the nearest enclosing transformed group, or -1
    int node;    // index into the node table (what a hit reports)
    double r0, r1;  // cylinder / cone minimum, maximum
    double p[12];   // inverse rows | group bounds (6) | triangle p1, e1, e2 (9)
                    // boxes (T_CULL, groups, CSG begin): p[6..8] = (centre, half width) per axis as float pairs,
                    // p[9] = two float factors of the margin (see slabs_f32 and fill_box_f32)
};
static_assert(sizeof(TRec) == 128, "traversal record must be 128 bytes");

// Shading record: everything the shading of a hit on one leaf needs, gathered so that a hit costs
// one round of (per-lane) loads instead of a chain of dependent ones through the node, transform,
// material and pattern tables.  Indexed by node.
enum { S_HAS_XFORM = 1, S_CASTS_SHADOW = 2, S_SOLID = 4 };
struct alignas(16) SRec {
    int type, flags, pattern, n_levels;  // n_levels: the leaf plus all its ancestors
    int anc_xf[RT_MAX_XFORM_NEST];                       // transforms of the transformed ancestors, nearest first, -1 padded
    int anc_level[RT_MAX_XFORM_NEST];                    // how many levels above the leaf each of them is
    int tri, n_anc, pad1, pad2;          // triangle normals index; how many entries of anc_xf are in use
    double inv[12];                      // the leaf's own inverse rows / transposed 3x3 (S_HAS_XFORM)
    double inv_t[9];
    double ka, kd, ks, shin, refl, transp, ri;  // Phong coefficients (phong.rs:10-19)
    double colour[3];                    // S_SOLID: the pattern is one colour
    double r0, r1;                       // cylinder / cone range
    double pad3;
};
static_assert(sizeof(SRec) == 304 + 8 * RT_MAX_XFORM_NEST && sizeof(SRec) % 16 == 0, "shading record layout");

struct DevScene {
    const RT_CONST SRec* shade;
    const RT_CONST TRec* trav;
    int n_trav;
    const RT_CONST rt_xform* xforms;  // ancestors' transforms for shading (SRec::anc_xf)
    const RT_CONST rt_tri* tris;
    const RT_CONST rt_tri_normals* trin;
    const RT_CONST rt_pattern* pats;
    const RT_CONST double* patx;
    const RT_CONST rt_light* lights;
    int n_lights;
    int all_cast_shadow;
    float prune_f32;    // 2.02e-9 * (1 + the largest |coordinate| of any group box), rounded up; infinite when
                        // a box is (no subtree is then pruned by its box times): see slabs_f32
    // Cubes (see the note at slabs_f32).  Scenes whose cubes are all axis-aligned in every space their
    // boxes are tested in: cube_kappa = 0 and rays with a direction component below RT_EPS * stretch
    // (ym above ym_limit) are not tame.  Other scenes: cube_kappa = 2 * 1.4143e-5 * stretch pads the boxes
    // per ray, and rays with ymin above ymin_limit (cube_kappa * ymin > 1/2) are not tame.
    float cube_kappa, ym_limit, ymin_limit;
    float extent_f32;   // largest |plane| of any box record
};

struct DevJob {
    // camera (src/scene/camera.rs)
    int width, height;
    double half_width, half_height, pixel_size;
    double inv[12];
    // partition
    int rank, world_size, tiles_x, local_tile_rows;
    int tile_row0;    // this launch renders the rank's tile rows tile_row0 .. tile_row0 + local_tile_rows - 1 (a band of its part)
    int max_depth;
    int wide_stores;  // f64 output: base 16-byte aligned and even width; u8 output: base 4-byte aligned and width % 4 == 0
    int out_u8;       // the output is 3 bytes per pixel, quantised like Canvas::clamp, instead of 3 doubles
    const int* tile_order;  // three arrays of tiles_x * local_tile_rows tile indices, one per cost class, or null
    const unsigned long long* tile_class_count;  // how many tiles each class holds
    int split;              // class-2 tiles (transparent hits) are rendered in `split` horizontal strips, one wavefront each: 1, 2 or 4
    long long extra_blocks; // workgroups launched beyond one per tile, for those strips
    // ray-list mode (rt_color_at)
    long long n_rays;
    const double* ray_o;
    const double* ray_d;
};

// ---------------------------------------------------------------------------------------------
// 3-vector algebra in the reference's operation order (src/tuples/tuple.rs).  The w lane is
// dropped: it is exactly 1 for points and 0 for vectors everywhere on this path, and the terms
// it contributes are exact zeros.
// ---------------------------------------------------------------------------------------------
struct V3 {
    double x, y, z;
};
DI V3 mk(double x, double y, double z) { return V3{x, y, z}; }
DI V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
DI V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
DI V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
DI V3 operator*(V3 a, double s) { return mk(a.x * s, a.y * s, a.z * s); }
DI V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
DI V3 operator/(V3 a, double s) { return mk(a.x / s, a.y / s, a.z / s); }
DI double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // tuple.rs:75-80
DI V3 cross(V3 a, V3 b) {                                                 // tuple.rs:83-92
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
DI double magnitude(V3 a) { return sqrt(dot(a, a)); }  // tuple.rs:49-53
DI V3 reflect(V3 in, V3 n) {                            // tuple.rs:94-99
    double d = dot(in, n);
    return in - (n * 2.0) * d;
}

// Matrix x tuple (src/matrices/matrix.rs:344-363) on rows 0..2 of an affine inverse.
template <class P>
DI V3 xpoint(P m, V3 p) {
    return mk(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
              m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
}
template <class P>
DI V3 xvec(P m, V3 v) {
    return mk(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z,
              m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
template <class P>
DI V3 xnormal(P t, V3 n) {
    return mk(t[0] * n.x + t[1] * n.y + t[2] * n.z, t[3] * n.x + t[4] * n.y + t[5] * n.z,
              t[6] * n.x + t[7] * n.y + t[8] * n.z);
}

// Applies, outermost first, the inverse transforms of the non-identity groups that enclose a node
// whose `ctx` link is `ctx` (Shape::intersect applied level by level, src/geometry/shape.rs:23-29,
// and Shape::world_to_object, shape.rs:65-79).  Identity levels are exact no-ops and are skipped.
// XF: the scene has transformed groups at all (without them no leaf has a transformed ancestor).
// Loops over a leaf's transformed ancestors: rolled or unrolled as measured (kernel ms, csg 3840x2160 /
// csg 1920x1080 / smooth teapot 3840x2160, all unrolled: 2.05 / 0.80 / 0.677).  The kernels for scenes with
// transformed groups are large enough for the instruction cache to matter (1.1 % of the CSG kernel's
// instruction fetches miss it), but a rolled loop inside the walk can also cost registers:
//   chain_ray rolled          1.97 / 0.77 / 0.797   (not kept: no faster any more in the kernels for CSG scenes
//                                                   once the other loops were rolled, and see chain_ray)
//   to_object rolled          2.01 / 0.78 / 0.663   (kept)
//   smooth-triangle chain     2.09 / 0.78 / 0.689   (left unrolled)
//   normal_to_world rolled    2.06 / 0.79 / 0.669   (kept)
#ifndef RT_ROLL_OBJ
#define RT_ROLL_OBJ 1
#endif
#ifndef RT_ROLL_SMOOTH
#define RT_ROLL_SMOOTH 0
#endif
#ifndef RT_ROLL_NORMAL
#define RT_ROLL_NORMAL 1
#endif
template <bool XF>
DI V3 to_object(const DevScene& S, const RT_CONST SRec* r, V3 p) {
    if (XF) {
        // (a loop, not eight copies: the kernels for scenes with transformed groups are large enough for
        // the instruction cache to matter)
#if RT_ROLL_OBJ
#pragma unroll 1
        for (int k = r->n_anc - 1; k >= 0; --k) p = xpoint(S.xforms[r->anc_xf[k]].inv, p);
#else
#pragma unroll
        for (int k = RT_MAX_XFORM_NEST - 1; k >= 0; --k)
            if (r->anc_xf[k] >= 0) p = xpoint(S.xforms[r->anc_xf[k]].inv, p);
#endif
    }
    if (r->flags & S_HAS_XFORM) p = xpoint(r->inv, p);
    return p;
}
// The reference multiplies a ray by the inverse of EVERY level's transform on its way down
// (Shape::intersect, shape.rs:23-29), the identity matrices of untransformed groups and shapes
// included.  For finite rays that changes nothing (at most the sign of a zero, which no test can
// tell) and the walk skips it; a ray with an infinite or NaN component comes out different: 0 * inf
// is NaN, so after one level every OTHER component is NaN too and after two all are.  Walks of
// wavefronts that carry such a ray apply the levels (`times` saturates at 2: a third changes nothing).
DI void through_identity(int times, V3& o, V3& d) {
    for (int k = 0; k < times; ++k) {
        o = mk((o.x + 0.0 * o.y) + 0.0 * o.z, (0.0 * o.x + o.y) + 0.0 * o.z, (0.0 * o.x + 0.0 * o.y) + o.z);
        d = mk((d.x + 0.0 * d.y) + 0.0 * d.z, (0.0 * d.x + d.y) + 0.0 * d.z, (0.0 * d.x + 0.0 * d.y) + d.z);
    }
}
// The same for a ray during traversal, on the traversal records (`ctx` is a record index there).
// wild: see through_identity.
DI void chain_ray(const DevScene& S, int ctx, bool wild, V3& o, V3& d) {
    // (unrolled: the rolled form, which looks the level-th ancestor up again for every level, was as fast
    // in the kernels for CSG scenes but made the compiler keep the ray in scratch memory around it:
    // 426 MB written per 1920x1080 csg frame against 245 MB)
    int c[RT_MAX_XFORM_NEST];
    c[0] = ctx;
#pragma unroll
    for (int k = 1; k < RT_MAX_XFORM_NEST; ++k) c[k] = c[k - 1] >= 0 ? S.trav[c[k - 1]].ctx : -1;
#pragma unroll
    for (int k = RT_MAX_XFORM_NEST - 1; k >= 0; --k)
        if (c[k] >= 0) {
            const RT_CONST double* m = S.trav[c[k]].p;
            if (wild) through_identity((S.trav[c[k]].kind >> T_LEVELS_SHIFT) & 3, o, d);
            o = xpoint(m, o);
            d = xvec(m, d);
        }
}

// BoundingBox::check_axis / Cube::check_axis (bounding_box.rs:102-124, cube.rs:47-69).
DI void check_axis(double origin, double direction, double mn, double mx, double& tmin, double& tmax) {
    double a = mn - origin;
    double b = mx - origin;
    bool finite = fabs(direction) >= RT_EPS;
    double t0 = finite ? a / direction : a * INFINITY;
    double t1 = finite ? b / direction : b * INFINITY;
    bool swap = t0 > t1;
    tmin = swap ? t1 : t0;
    tmax = swap ? t0 : t1;
}
// BoundingBox::intersects / Cube::local_intersect slab test: returns tmin, tmax.
DI void slabs(V3 o, V3 d, double x0, double y0, double z0, double x1, double y1, double z1, double& tmin,
              double& tmax) {
    double xa, xb, ya, yb, za, zb;
    check_axis(o.x, d.x, x0, x1, xa, xb);
    check_axis(o.y, d.y, y0, y1, ya, yb);
    check_axis(o.z, d.z, z0, z1, za, zb);
    tmin = fmax(fmax(xa, ya), za);
    tmax = fmin(fmin(xb, yb), zb);
}

// Exact division by a value whose correctly rounded reciprocal y = 1.0 / d is already known:
// q0 = a*y is within 2 ulp of a/d, r = a - q0*d is exact in an FMA, and rounding q0 + r*y once
// gives the correctly rounded quotient -- the same final step the hardware's own f64 division
// expansion ends with.  It returns the IEEE quotient bit for bit for finite a and normal d, q
// (rt_selftest_division checks this on the device); it is NOT valid for infinite a.
DI double div_by(double a, double d, double y) {
    double q0 = a * y;
    double r = __builtin_fma(-q0, d, a);
    return __builtin_fma(r, y, q0);
}
// The slab test for lanes whose direction components all satisfy |d| >= RT_EPS and whose box is
// finite: the `* INFINITY` branch of check_axis is not taken and no NaN can arise, so the ordered
// pair is (min, max) of the two quotients.
DI void slabs_fast(V3 o, V3 d, V3 y, double x0, double y0, double z0, double x1, double y1, double z1,
                   double& tmin, double& tmax) {
    double ax = div_by(x0 - o.x, d.x, y.x), bx = div_by(x1 - o.x, d.x, y.x);
    double ay = div_by(y0 - o.y, d.y, y.y), by = div_by(y1 - o.y, d.y, y.y);
    double az = div_by(z0 - o.z, d.z, y.z), bz = div_by(z1 - o.z, d.z, y.z);
    tmin = fmax(fmax(fmin(ax, bx), fmin(ay, by)), fmin(az, bz));
    tmax = fmin(fmin(fmax(ax, bx), fmax(ay, by)), fmax(az, bz));
}
DI bool all_axes_finite(V3 d) { return fabs(d.x) >= RT_EPS && fabs(d.y) >= RT_EPS && fabs(d.z) >= RT_EPS; }
// Correctly rounded 1/d for normal-range d (the callers only use it where |d| >= RT_EPS and d is a
// ray direction component): hardware estimate, two Newton steps, then the residual correction
// that makes the result the IEEE quotient 1.0 / d.  This is the compiler's own f64 division
// expansion without the operand scaling and special-case fix-up that only matter near the ends
// of the exponent range (rt_selftest_division compares it with 1.0 / d on the device).
DI double rcp_exact(double d) {
    double y = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(y, e, y);
    double r = __builtin_fma(-d, y, 1.0);
    return __builtin_fma(r, y, y);
}
// Components the reference treats as parallel to a slab (|d| < RT_EPS, bounding_box.rs:107-113)
// get NaN: the fast slab test is only used when no lane has one, and the cull test then leaves
// that axis unconstrained.
DI double reciprocal1(double d) { return fabs(d) >= RT_EPS ? rcp_exact(d) : __builtin_nan(""); }
DI V3 reciprocal(V3 d) { return mk(reciprocal1(d.x), reciprocal1(d.y), reciprocal1(d.z)); }

// ---- box tests in single precision ------------------------------------------------------------
// Whether a ray can reach a box only decides what the walk skips, never a value that is stored, so
// the decision may come from approximate slab times as long as their error is bounded and every
// comparison leaves that much room.  Per axis t = plane * y + n with y = 1 / d and n = -o * y; in f32,
// with y from the hardware reciprocal (1 ulp) of the rounded d, each time differs from the real
// quotient by less than 6 u |y| (|plane| + |o|), u = 2^-24: the roundings of plane, d, o, the
// reciprocal, the product o * y and the fused multiply-add (the reference's own f64 roundings are
// nine orders of magnitude below that).  The margin uses 16 u:
//     G = 2 * 16 u * ym * (E + so)  +  pad  +  1e-30
// ym = largest |y| over the axes that count, E = largest |plane| of the box, so = |ox| + |oy| + |oz|;
// pad is the allowance the f64 tests had: for T_CULL boxes (r0 + r1 |o|^2) ym (a leaf's own
// quadratic can report a hit slightly outside its box: see fill_cull), for group boxes
// 2e-9 (base + |o|^2) ym (a child's time and its group's box times round differently); 1e-30
// covers values that f32 flushes to zero.  The walk takes these shortcuts only for rays whose
// components are all below 1e15 in magnitude (`tame`; directions that count are at least RT_EPS), so
// that nothing overflows or leaves the normal range of f32; other rays get an infinite margin, which
// decides nothing.
// Axes the reference treats as parallel to the slab (|d| < RT_EPS, bounding_box.rs:107-113) carry
// NaN and drop out of min / max: the box test then ignores that axis, which can only keep more.
// A group's bounds test must agree with the reference's (group.rs:133-154) exactly; lanes whose
// tmin and tmax are closer than the margin, or that have a parallel axis, repeat it in f64.
// Cubes: for an axis whose object-space direction component is below RT_EPS the reference only asks
// whether the ORIGIN lies between the two faces (cube.rs:47-69), so the hit it reports at time t can
// lie outside the cube by up to RT_EPS |t| along that axis: at most sqrt(2) RT_EPS |t| in object
// space (with three such axes the times are infinite), kappa |t| in the space of a box around the
// cube, kappa = 1.4143e-5 * stretch.  Two ways to stay on the safe side, chosen per scene:
//   * every cube is axis-aligned in every space its boxes are tested in (the chain of transforms
//     above it only scales and translates): the branch is only taken by rays with a direction
//     component below RT_EPS * stretch in that space, and those are simply not tame;
//   * otherwise the boxes are grown per ray by rho >= kappa |t|: a hit at time t lies within
//     E + rho + so of the origin along every axis, so |t| <= (E + so + rho) ymin (ymin = smallest |y|)
//     and rho = 2 kappa ymin (E + so) does while kappa ymin <= 1/2 (E: the largest |plane| of any box
//     of the scene).  Growing a box by rho is one addition to the half width of each slab.  Group
//     bounds tests see the grown box too: what it misses the group's own box misses, and a lane only
//     counts as inside without the f64 test when tmax - tmin exceeds the margin by 4 rho ym more.
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr float BOX_F32_2C = 2.86102294921875e-6f;  // 2 * 24 u
// A slab is stored as (centre, half width) c, w: its two times are (c y + n) -+ w |y|, near first,
// whatever the sign of y, so no minimum / maximum per axis is needed (each costs as much as an f64
// multiply-add): three single-precision multiply-adds per axis, |y| as an operand modifier.  The
// half width is rounded up by the rounding of the centre, so the f32 slab contains the f64 one.
// (The final three-way min / max are written as instructions: the compiler cannot see that these
// values are never signalling NaNs and would canonicalise each one.)
DI float min_f32(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
DI float max3_f32(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
DI float min3_f32(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
DI float min3_abs_f32(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, |%1|, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
DI float max3_abs_f32(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// The ray as the box tests see it: 1/d and -o/d per axis (NaN on axes that count as parallel), and
// by how much the boxes are grown for this ray (see the note on cubes).
struct BoxRay {
    float yx, yy, yz;
    float nx, ny, nz;
    float rho;
};
// px, py, pz: the box's (centre, half width) per axis as two floats in one 8-byte slot of the record.
template <bool GROW>
DI void slabs_f32(double px, double py, double pz, const BoxRay& b, float& tmin, float& tmax) {
    const f2 cx = __builtin_bit_cast(f2, px), cy = __builtin_bit_cast(f2, py), cz = __builtin_bit_cast(f2, pz);
    const float mx = __builtin_fmaf(cx.x, b.yx, b.nx), my = __builtin_fmaf(cy.x, b.yy, b.ny), mz = __builtin_fmaf(cz.x, b.yz, b.nz);
    const float wx = GROW ? cx.y + b.rho : cx.y, wy = GROW ? cy.y + b.rho : cy.y, wz = GROW ? cz.y + b.rho : cz.y;
    tmin = max3_f32(__builtin_fmaf(-wx, fabsf(b.yx), mx), __builtin_fmaf(-wy, fabsf(b.yy), my), __builtin_fmaf(-wz, fabsf(b.yz), mz));
    tmax = min3_f32(__builtin_fmaf(wx, fabsf(b.yx), mx), __builtin_fmaf(wy, fabsf(b.yy), my), __builtin_fmaf(wz, fabsf(b.yz), mz));
}
// An f32 that is not below the positive f64 x (or within 1e-38 of it when x is that small).
DI float up_f32(double x) { return (float)x * 1.00000012f; }

// Per-leaf accumulator: nearest positive time (Intersection::hit, intersection.rs:32-41) and,
// for the n1/n2 bookkeeping, the parity and the latest of the leaf's non-positive times.
struct LeafAcc {
    double best;  // 0.0: none yet (the times that count are > 0)
    int par;
    double negmax;
};
// Per-lane list of the intersections collected inside the outermost CSG shape being traversed.
// (The list's length is a variable of its own in the walk: as a member of this struct, whose arrays are
// indexed per lane and therefore live in scratch memory, it was loaded and stored there with every entry.)
struct CsgList {
    double t[RT_MAX_CSG_HITS];
    int id[RT_MAX_CSG_HITS];
    int start[RT_MAX_CSG_NEST];  // list length when each open CSG shape began
};
// f64::total_cmp as an integer key (intersection.rs:60-64); the map is its own inverse.
DI long long total_key(double v) {
    long long b = __double_as_longlong(v);
    b ^= (long long)((unsigned long long)(b >> 63) >> 1);
    return b;
}
DI double key_time(long long k) {
    k ^= (long long)((unsigned long long)(k >> 63) >> 1);
    return __longlong_as_double(k);
}
constexpr int CSG_FAST = 8;  // list parts up to this length are sorted in registers
// CONTAINERS: only the times <= 0 are of interest (see MODE_CONTAINERS); otherwise only the nearest
// positive time.
template <bool CONTAINERS>
DI void leaf_count(LeafAcc& a, double t) {
    if (!CONTAINERS) {
        if (t > 0.0 && (a.best == 0.0 || t < a.best)) a.best = t;
    } else if (t <= 0.0) {
        a.par ^= 1;
        a.negmax = fmax(a.negmax, t);
    }
}

struct Hit {
    bool found;
    double t;
    int node;
    // World::prepare_computations container walk (world.rs:99-132), reduced to what decides
    // n1/n2: whether the hit leaf has an odd number of earlier intersections, and the two
    // leaves with an odd count whose last earlier intersection comes latest in sorted order.
    int par;
    double a_t, b_t;
    int a_id, b_id;
#ifdef RT_PROFILE
    int visits;  // records visited by the wavefront (diagnostic build only)
    unsigned long long t_end, t_leaf;  // ticks spent in CSG end records and in leaf records
#endif
#ifdef RT_PROFILE_MIX
    // diagnostic build: records visited by kind and how many lanes took part in each
    int n_cull, a_cull, n_group, a_group, n_leaf, a_leaf, n_leaf_idle;
#endif
};

// Lanes of the wavefront for which the predicate holds, straight from the comparison's lane mask
// (the library's __ballot takes an int, which costs a select and a second comparison).
DI unsigned long long wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// The same value, as far as the compiler can tell a new one: what is derived from it is computed where
// it is used instead of once, far ahead, and kept (or spilled) until then.
template <typename T>
DI T opaque(T x) {
    asm volatile("" : "+v"(x));
    return x;
}
// The lane number, computed on the spot (two instructions that read no register), for code that runs
// after a long loop which would otherwise keep it, or spill it, for that use alone.
DI int fresh_lane() {
    int x;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(x));
    return x;
}

DI int wave_min(int v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        int o = __shfl_xor(v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}

enum { MODE_CLOSEST = 0, MODE_CONTAINERS = 1, MODE_SHADOW = 2 };

// World::intersect_world + Intersection::hit for all lanes of the wavefront at once
// (world.rs:55-68, group.rs:133-154 and the leaf local_intersect functions).  Must be called from
// wave-uniform control flow; lanes without a ray pass active = false.
// MODE_CONTAINERS: the second pass for a ray whose nearest hit (already in `h`) is transparent: what
// World::prepare_computations' container walk (world.rs:99-132) needs, i.e. for every leaf the
// parity and the latest of its times <= 0.  Positive times are ignored, so everything whose box
// lies wholly in front of the origin is skipped.
// MODE_SHADOW: `limit` is the distance to the light; when every leaf casts a shadow a lane stops
// at its first intersection in (0, limit) (any such intersection implies the nearest one is too).
// XF: the scene has transformed groups or triangles (T_XFORM records); without them the ray never
// changes space between records and the world-space copy is not kept.
// CULL: the scene has T_CULL records.  CSGM: it has CSG shapes.
#ifndef RT_INNER_CULL
#define RT_INNER_CULL (!CSGM)  // cull boxes in a loop of their own (see trace); the kernels for CSG scenes are 6 % slower with it
#endif
#ifndef RT_WALK_PRIO
#define RT_WALK_PRIO 2  // s_setprio level of a wavefront inside a traversal walk (0 elsewhere)
#endif
template <int MODE, bool XF, bool CULL, bool CSGM, bool GROW>
DI void trace(const DevScene& S, bool active, V3 wo, V3 wd, double limit, Hit& h) {
    constexpr bool TRACK = MODE == MODE_CONTAINERS;
    if (!TRACK) {
        h.found = false;  // inside the walk "found" is h.node >= 0; the flag is set at the end
        h.t = 0.0;
        h.node = -1;
    }
    h.par = 0;
    h.a_t = 0.0;
    h.b_t = 0.0;
    h.a_id = -1;
    h.b_id = -1;
#ifdef RT_PROFILE
    h.visits = 0;
    h.t_end = h.t_leaf = 0;
#endif
#ifdef RT_PROFILE_MIX
    h.n_cull = h.a_cull = h.n_group = h.a_group = h.n_leaf = h.a_leaf = h.n_leaf_idle = 0;
#endif
    if (wave_ballot(active) == 0ull) return;
    // A walk is a chain of dependent scalar fetches with a few vector instructions in between: a
    // wavefront in it runs at a higher priority than the ones that shade, which have long runs of
    // independent arithmetic to fill the gaps with (2 % on large frames, 6 % on 1920x1080; the
    // other way round costs 4 %).
    __builtin_amdgcn_s_setprio(RT_WALK_PRIO);

    int skip_until = active ? 0 : INT_MAX;
    const int n = S.n_trav;
    int i = 0;
    int cur_ctx = -1;
    V3 co = wo, cd = wd;  // the ray in the space of the transformed group `cur_ctx`
    // The ray as the single-precision box tests see it (see slabs_f32): per axis (1 / d, -o / d), the
    // largest |1 / d|, a bound on |o|^2 times that, and the parts of the margins that do not depend
    // on the box.  hlim: no time beyond it counts (the light's distance for a shadow ray, then the
    // nearest hit so far; for the container pass only times <= 0 count).
    BoxRay box;
    float ym, osq_ym, c_cull, c_group, grow4 = 0.0f;
    float hlim = TRACK ? 0.0f : (MODE == MODE_SHADOW ? up_f32(limit) : INFINITY);
    bool cfin;   // no axis of this lane's ray counts as parallel to a slab
    bool plain;  // wave-uniform: every active lane's ray has finite components below 1e15 (see T_DIAG)
    auto ray_changed = [&]() {
        const bool fx = fabs(cd.x) >= RT_EPS, fy = fabs(cd.y) >= RT_EPS, fz = fabs(cd.z) >= RT_EPS;
        cfin = fx && fy && fz;
        const float dx = (float)cd.x, dy = (float)cd.y, dz = (float)cd.z;
        const float ox = (float)co.x, oy = (float)co.y, oz = (float)co.z;
        const float yx = fx ? __builtin_amdgcn_rcpf(dx) : __builtin_nanf("");
        const float yy = fy ? __builtin_amdgcn_rcpf(dy) : __builtin_nanf("");
        const float yzz = fz ? __builtin_amdgcn_rcpf(dz) : __builtin_nanf("");
        ym = max3_abs_f32(yx, yy, yzz);
        const float so = (fabsf(ox) + fabsf(oy)) + fabsf(oz);
        const float sd = (fabsf(dx) + fabsf(dy)) + fabsf(dz);
        const bool bounded = (so + sd) <= 1e15f;  // false for NaN and infinite components too
        const float mx = -(ox * yx), my = -(oy * yy), mz = -(oz * yzz);
        box.yx = yx;
        box.yy = yy;
        box.yz = yzz;
        bool tame = bounded && ym <= S.ym_limit;
        box.nx = mx;
        box.ny = my;
        box.nz = mz;
        if (GROW) {  // scenes with cubes that are not axis-aligned
            const float ymin = min3_abs_f32(yx, yy, yzz);
            box.rho = (S.cube_kappa * ymin) * (S.extent_f32 + so);
            tame = tame && ymin <= S.ymin_limit;
            grow4 = (4.0f * box.rho) * ym;
        }
        const float osq = so * so;             // >= |o|^2
        const float cc = __builtin_fmaf(ym * BOX_F32_2C, so, 1e-30f);
        const float cg = __builtin_fmaf(ym, __builtin_fmaf(2.02e-9f, osq, S.prune_f32), cc);
        c_cull = tame ? cc : INFINITY;
        c_group = tame ? cg : INFINITY;
        osq_ym = osq * ym;
        plain = wave_ballot(active && !bounded) == 0ull;
    };
    ray_changed();
    const bool wild = !plain;  // wave-uniform: some lane's ray (as given) has an infinite, NaN or huge component: see through_identity
    const bool early = (MODE == MODE_SHADOW) && S.all_cast_shadow;
    int csg_depth = 0;  // open CSG shapes around the current record (wave-uniform)
    CsgList cl;
    int cl_n = 0;  // entries in the list; counts on beyond RT_MAX_CSG_HITS, entries beyond are dropped
    LeafAcc acc;
    int acc_node = -1;
    // Intersection::hit and the container walk, one leaf at a time in depth-first order.  Stable
    // sorts keep that order among equal times, so a later leaf only wins with a strictly smaller
    // time (group.rs:147, world.rs:65).
    auto begin_leaf = [&](int node) {
        acc.best = 0.0;
        acc.par = 0;
        acc.negmax = -INFINITY;
        acc_node = node;
    };
    auto leaf_add = [&](double t) {
        if (CSGM && csg_depth > 0) {
            if (cl_n < RT_MAX_CSG_HITS) {
                cl.t[cl_n] = t;
                cl.id[cl_n] = acc_node;
            }
            ++cl_n;
        } else {
            leaf_count<TRACK>(acc, t);
        }
    };
    // Written as selects, not branches: the hit state then stays in place across the loop instead of
    // being copied around the join points of every leaf visit.
    auto commit_leaf = [&]() {
        if (!TRACK) {
            // equal times: the leaf that comes first in the reference's order (= node index) wins,
            // whatever order the records are visited in
            const bool better = acc.best != 0.0 && (h.node < 0 || acc.best < h.t || (acc.best == h.t && acc_node < h.node));
            h.t = better ? acc.best : h.t;
            h.node = better ? acc_node : h.node;
            if (!early) hlim = better ? min_f32(hlim, up_f32(acc.best)) : hlim;  // with `early` the lane is decided below
        } else {
            h.par = acc_node == h.node ? acc.par : h.par;  // parity of the hit leaf's own earlier times
            const bool odd = acc.par != 0;
            // latest time first; among equal times the leaf later in the reference's order
            const bool first = odd && (h.a_id < 0 || acc.negmax > h.a_t || (acc.negmax == h.a_t && acc_node > h.a_id));
            const bool second = odd && !first && (h.b_id < 0 || acc.negmax > h.b_t || (acc.negmax == h.b_t && acc_node > h.b_id));
            h.b_t = first ? h.a_t : (second ? acc.negmax : h.b_t);
            h.b_id = first ? h.a_id : (second ? acc_node : h.b_id);
            h.a_t = first ? acc.negmax : h.a_t;
            h.a_id = first ? acc_node : h.a_id;
        }
        if (early && h.node >= 0 && h.t < limit) skip_until = INT_MAX;  // this lane is decided
    };

#ifdef RT_PROFILE
    h.visits = 0;
#endif
    // The record at `i`, fetched whole with one scalar-load round trip (base + 32-bit byte offset: the
    // loads take the offset in a register and no 64-bit address is computed per record).  The empty
    // asm statements make every field live at once, so the loads are issued together instead of one
    // dependent load per use.
    int kind, skip, ctx, rnode;
    double r0, r1;
    double q[12];
    auto fetch = [&]() {
        static_assert(sizeof(TRec) == 128, "the record offset below is i * 128");
        const RT_CONST TRec* rec = reinterpret_cast<const RT_CONST TRec*>(
            reinterpret_cast<const RT_CONST char*>(S.trav) + ((unsigned)i << 7));
        kind = rec->kind;
        skip = rec->skip;
        ctx = rec->ctx;
        rnode = rec->node;
        r0 = rec->r0;
        r1 = rec->r1;
#pragma unroll
        for (int k = 0; k < 12; ++k) q[k] = rec->p[k];
        asm volatile("" ::"s"(kind), "s"(skip), "s"(ctx), "s"(rnode), "s"(r0), "s"(r1));
        asm volatile("" ::"s"(q[0]), "s"(q[1]), "s"(q[2]), "s"(q[3]), "s"(q[4]), "s"(q[5]), "s"(q[6]), "s"(q[7]),
                     "s"(q[8]), "s"(q[9]), "s"(q[10]), "s"(q[11]));
#ifdef RT_PROFILE
        ++h.visits;
#endif
    };
    // Padded box in front of a leaf or a run of siblings, tested in single precision (see slabs_f32):
    // p[6..8] hold the planes as floats, p[9] the margin's two box-dependent factors.  Returns the next
    // record: the one after the box's subtree when no lane needs it.
    auto cull_box = [&]() -> int {
        const bool act = i >= skip_until;
#ifdef RT_PROFILE_MIX
        ++h.n_cull;
        h.a_cull += __builtin_popcountll(wave_ballot(act));
#endif
        float tmin, tmax;
        slabs_f32<GROW>(q[6], q[7], q[8], box, tmin, tmax);
        const f2 kk = __builtin_bit_cast(f2, q[9]);
        const float G = __builtin_fmaf(kk.y, osq_ym, __builtin_fmaf(kk.x, ym, c_cull));
        // (lane masks of the single comparisons, combined as scalars: a ballot of the combined
        // condition would go through a vector register)
        const bool c_miss = (tmin - tmax) > G;
        bool keep = !c_miss;
        unsigned long long drop = wave_ballot(c_miss);
        // T_LINE_ONLY (a box in front of a transformed group or CSG shape with a cone below it): hits below
        // can lie outside the box, only the ray's line decides, as in the reference's own bounds test.
        if (!(XF && (kind & T_LINE_ONLY))) {
            // Inside a CSG shape the filter needs every intersection up to the times that count: what lies
            // behind the origin decides whether the ray starts inside an operand.  What lies beyond every
            // time that counts (the light, the nearest hit so far; anything positive in the container pass)
            // cannot change the verdict on an earlier intersection, so it may go there as well.
            if (!TRACK && (!CSGM || csg_depth == 0)) {
                const bool c_behind = tmax < -G;  // only positive times count
                keep = keep && !c_behind;
                drop |= wave_ballot(c_behind);
            }
            const bool c_beyond = (tmin - G) > hlim;  // beyond the light or the nearest hit so far; container pass: > 0
            keep = keep && !c_beyond;
            drop |= wave_ballot(c_beyond);
        }
        if (act && !keep) skip_until = skip;
        return ((wave_ballot(act) & ~drop) == 0ull) ? skip : i + 1;
    };
    while (i < n) {
        i = __builtin_amdgcn_readfirstlane(i);
        fetch();
        if (CULL && RT_INNER_CULL) {
            // Runs of cull boxes in the space the ray is in already (half of all records) go round a
            // loop of their own: it has only wave-uniform branches, so its back edge is a plain jump
            // and the hit state is not touched, where the big loop's body, which has divergent parts,
            // is threaded through join blocks.  (Group boxes in this loop too, with or without their
            // exact fallback, made every workload slower.)
            while ((kind & 0xff) == T_CULL && (!XF || ctx == cur_ctx)) {
                i = cull_box();
                if (__builtin_expect(i >= n, 0)) break;
                fetch();
            }
            if (i >= n) break;
        }
        const int type = kind & 0xff;
        if (XF && ctx != cur_ctx) {
            co = wo;
            cd = wd;
            chain_ray(S, ctx, wild, co, cd);
            ray_changed();
            cur_ctx = ctx;
        }
        const bool act = i >= skip_until;
        if (XF && type == T_XFORM) {
            // The inverse of a transformed group (or triangle): the records after it, up to the
            // end of the node's subtree, see the ray in that node's space.
            if (wild) through_identity((kind >> T_LEVELS_SHIFT) & 3, co, cd);
            V3 to = xpoint(q, co), td = xvec(q, cd);
            co = to;
            cd = td;
            ray_changed();
            cur_ctx = i;
            i = i + 1;
            continue;
        }
#ifdef RT_PROFILE_MIX
        {
            const int lanes = __builtin_popcountll(wave_ballot(act));
            if (type == RT_GROUP || type == T_CSG_BEGIN) { ++h.n_group; h.a_group += lanes; }
            else if (type < T_XFORM) { ++h.n_leaf; h.a_leaf += lanes; if (lanes == 0) ++h.n_leaf_idle; }
        }
#endif
        if ((XF || !RT_INNER_CULL) && CULL && type == T_CULL) {  // (the first box in a transformed group's space; the others take the loop above)
            i = cull_box();
            continue;
        }
        if (CSGM && type == T_CSG_END) {
            // CSG::local_intersect after its operands (csg.rs:150-155): sort this shape's part of
            // the list by time (stable, total order) and keep what filter_intersections keeps.
#ifdef RT_PROFILE
            const unsigned long long pe0 = __builtin_amdgcn_s_memtime();
#endif
            --csg_depth;
            const int first = cl.start[csg_depth];
            const int last = cl_n < RT_MAX_CSG_HITS ? cl_n : RT_MAX_CSG_HITS;
            // (nothing collected by any lane: nothing to sort, filter or hand over)
            if (wave_ballot(last > first) == 0ull) {
                cl_n = first;
                i = i + 1;
#ifdef RT_PROFILE
                h.t_end += __builtin_amdgcn_s_memtime() - pe0;
#endif
                continue;
            }
            // Short lists (every lane's part has at most CSG_FAST entries: the usual case) are sorted and
            // filtered in registers.  The list lives in scratch memory, where every step of an insertion sort
            // is a load that depends on the one before it.  `bound` (2, 4 or 8, by the longest part in the
            // wavefront) is wave-uniform: entries from there on are padding in every lane, and the loads, the
            // comparators and the filter steps that would only touch padding are jumped over (one copy of
            // the code: three separate networks made the kernel larger and slower).
            auto end_fast = [&]() {
                constexpr int N = CSG_FAST;
                static_assert(N == 8, "the networks below are for 2, 4 and 8 inputs");
                const int m = last - first;
                const int bound = wave_ballot(m > 4) != 0ull ? 8 : (wave_ballot(m > 2) != 0ull ? 4 : 2);
                long long key[N];
                int id[N], pos[N];
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    key[k] = LLONG_MAX;  // the padding sorts last
                    id[k] = 0;
                    pos[k] = k;  // arrival order: what a stable sort keeps among equal keys
                    if (k < bound) {
                        const bool in = k < m;
                        const int at = in ? first + k : 0;
                        const double t = cl.t[at];
                        id[k] = cl.id[at];
                        key[k] = in ? total_key(t) : LLONG_MAX;
                    }
                }
                // (key, arrival) pairs are all different, so the result is the stable order.
                auto order = [&](int a, int b) {
                    const bool swap = key[b] < key[a] || (key[b] == key[a] && pos[b] < pos[a]);
                    const long long ka = key[a], kb = key[b];
                    const int ia = id[a], ib = id[b], pa = pos[a], pb = pos[b];
                    key[a] = swap ? kb : ka;
                    key[b] = swap ? ka : kb;
                    id[a] = swap ? ib : ia;
                    id[b] = swap ? ia : ib;
                    pos[a] = swap ? pb : pa;
                    pos[b] = swap ? pa : pb;
                };
                if (bound == 2) {
                    order(0, 1);
                } else if (bound == 4) {
                    order(0, 1); order(2, 3);
                    order(0, 2); order(1, 3);
                    order(1, 2);
                } else {
                    // 19 comparators
                    order(0, 1); order(2, 3); order(4, 5); order(6, 7);
                    order(0, 2); order(1, 3); order(4, 6); order(5, 7);
                    order(1, 2); order(5, 6); order(0, 4); order(3, 7);
                    order(1, 5); order(2, 6);
                    order(1, 4); order(3, 6);
                    order(2, 4); order(3, 5);
                    order(3, 4);
                }
                const int op = (kind >> T_CSG_OP_SHIFT) & 3;
                bool inl = false, inr = false;
                bool keep[N];
#pragma unroll
                for (int a = 0; a < N; ++a) {
                    keep[a] = false;
                    if (a < bound) {
                        const bool in = a < m;  // the padding sorts last
                        const bool lhit = id[a] > rnode && id[a] < skip;  // the left operand's nodes (csg.rs:91)
                        bool allowed;                                     // csg.rs:113-140
                        if (op == 0)
                            allowed = (lhit && !inr) || (!lhit && inl);
                        else if (op == 1)
                            allowed = (lhit && inr) || (!lhit && inl);
                        else
                            allowed = (lhit && !inr) || (!lhit && !inl);
                        keep[a] = in && allowed;
                        inl = inl != (in && lhit);
                        inr = inr != (in && !lhit);
                    }
                }
                if (csg_depth == 0) {
                    // hand over leaf by leaf in node order, as below
                    int done = -1;
                    for (;;) {
                        int next = INT_MAX;
#pragma unroll
                        for (int a = 0; a < N; ++a)
                            if (a < bound && keep[a] && id[a] > done && id[a] < next) next = id[a];
                        if (wave_ballot(next != INT_MAX) == 0ull) break;
                        begin_leaf(next);
#pragma unroll
                        for (int a = 0; a < N; ++a)
                            if (a < bound && keep[a] && id[a] == next) leaf_count<TRACK>(acc, key_time(key[a]));
                        commit_leaf();
                        done = next;
                    }
                    cl_n = 0;
                } else {
                    int kept = first;
#pragma unroll
                    for (int a = 0; a < N; ++a)
                        if (a < bound && keep[a]) {
                            cl.t[kept] = key_time(key[a]);
                            cl.id[kept] = id[a];
                            ++kept;
                        }
                    cl_n = kept;
                }
            };
            if (wave_ballot(last - first > CSG_FAST) == 0ull) {
                end_fast();
                i = i + 1;
                if (early && wave_ballot(i >= skip_until) == 0ull) i = wave_min(skip_until);
#ifdef RT_PROFILE
                h.t_end += __builtin_amdgcn_s_memtime() - pe0;
#endif
                continue;
            }
            for (int a = first + 1; a < last; ++a) {
                const double ta = cl.t[a];
                const int ia = cl.id[a];
                const long long ka = total_key(ta);
                int b = a - 1;
                while (b >= first && total_key(cl.t[b]) > ka) {
                    cl.t[b + 1] = cl.t[b];
                    cl.id[b + 1] = cl.id[b];
                    --b;
                }
                cl.t[b + 1] = ta;
                cl.id[b + 1] = ia;
            }
            const int op = (kind >> T_CSG_OP_SHIFT) & 3;
            bool inl = false, inr = false;
            int kept = first;
            for (int a = first; a < last; ++a) {
                const bool lhit = cl.id[a] > rnode && cl.id[a] < skip;  // the left operand's nodes (csg.rs:91)
                bool allowed;                                           // csg.rs:113-140
                if (op == 0)
                    allowed = (lhit && !inr) || (!lhit && inl);
                else if (op == 1)
                    allowed = (lhit && inr) || (!lhit && inl);
                else
                    allowed = (lhit && !inr) || (!lhit && !inl);
                if (allowed) {
                    cl.t[kept] = cl.t[a];
                    cl.id[kept] = cl.id[a];
                    ++kept;
                }
                if (lhit)
                    inl = !inl;
                else
                    inr = !inr;
            }
            cl_n = kept;
            if (csg_depth == 0) {
                // The survivors join the enclosing list: hand them over leaf by leaf in depth-first
                // (= node) order, exactly as if each leaf had produced only these intersections.
                int done = -1;
                for (;;) {
                    int next = INT_MAX;
                    for (int a = 0; a < cl_n; ++a)
                        if (cl.id[a] > done && cl.id[a] < next) next = cl.id[a];
                    if (next == INT_MAX) break;
                    begin_leaf(next);
                    for (int a = 0; a < cl_n; ++a)
                        if (cl.id[a] == next) leaf_count<TRACK>(acc, cl.t[a]);
                    commit_leaf();
                    done = next;
                }
                cl_n = 0;
            }
            i = i + 1;
            if (early && wave_ballot(i >= skip_until) == 0ull) i = wave_min(skip_until);
#ifdef RT_PROFILE
            h.t_end += __builtin_amdgcn_s_memtime() - pe0;
#endif
            continue;
        }
        if (type == RT_GROUP || (CSGM && type == T_CSG_BEGIN)) {
            // Group::local_intersect (group.rs:133-154) and CSG::local_intersect (csg.rs:145):
            // bounds test in the shape's own space.
            // Single precision decides every lane whose tmin and tmax are further apart than the
            // margin; the others (and lanes with an axis that counts as parallel, which the f32 test
            // ignores but the reference does not) repeat the test in f64 exactly as the reference does.
            float tmin, tmax;
            slabs_f32<GROW>(q[6], q[7], q[8], box, tmin, tmax);
            const float G = __builtin_fmaf(__builtin_bit_cast(f2, q[9]).x, ym, c_group);
            const float gap = tmin - tmax;
            bool miss = gap > G;
            const bool unsure = act && (!(miss || (GROW ? gap + grow4 : gap) < -G) || !cfin);
            if (wave_ballot(unsure) != 0ull) {
                double e0, e1;
                V3 go = co, gd = cd;
                if (wild) through_identity((kind >> T_LEVELS_SHIFT) & 3, go, gd);
                slabs(go, gd, q[0], q[1], q[2], q[3], q[4], q[5], e0, e1);
                miss = unsure ? (e0 > e1) : miss;
            }
            bool inside = act && !miss;
            // The reference enters every group whose box the ray's LINE meets.  Subtrees that can
            // only yield times that do not count are skipped: behind the origin (unless the n1/n2
            // bookkeeping needs the times <= 0), beyond the light, beyond the nearest hit so far.
            if ((!CSGM || csg_depth == 0) && !(kind & T_LINE_ONLY)) {  // inside a CSG shape its filter needs every intersection
                if (!TRACK) inside = inside && !(tmax < -G);
                inside = inside && !((tmin - G) > hlim);
            }
            if (act && !inside) skip_until = skip;
            const bool enter = wave_ballot(inside) != 0ull;
            if (CSGM && type == T_CSG_BEGIN && enter) {
                cl.start[csg_depth] = cl_n;
                ++csg_depth;
            }
            i = enter ? i + 1 : skip;
            continue;
        }
#ifdef RT_PROFILE
        const unsigned long long pl0 = __builtin_amdgcn_s_memtime();
#endif
        V3 lo = co, ld = cd;
        if (wild) through_identity((kind >> T_LEVELS_SHIFT) & 3, lo, ld);
        if ((kind & T_DIAG) && plain) {
            // The same products and sums with the terms left out that are exact zeros: 0 * finite is
            // +-0, and x + (+-0) is x, except that an exact zero may come out with the other sign; no
            // leaf test can tell (times that are +-0 do not count as hits, directions that are +-0 take
            // the `* INFINITY` branches, squares and differences from +-1 lose the sign).
            lo = mk(q[0] * co.x + q[3], q[5] * co.y + q[7], q[10] * co.z + q[11]);
            ld = mk(q[0] * cd.x, q[5] * cd.y, q[10] * cd.z);
        } else if (kind & T_HAS_INV) {  // Shape::intersect: world_ray.transform(inverse) (shape.rs:24-26, ray.rs:30-35)
            const V3 po = lo, pd = ld;
            lo = xpoint(q, po);
            ld = xvec(q, pd);
        }

        begin_leaf(rnode);
        if (act) {
            switch (type) {
                case RT_SPHERE: {  // sphere.rs:52-78
                    double a = dot(ld, ld);
                    double b = 2.0 * dot(ld, lo);
                    double c = dot(lo, lo) - 1.0;
                    double disc = (b * b) - (4.0 * a * c);
                    if (!(disc < 0.0)) {
                        double sq = sqrt(disc);
                        double den = a * 2.0;
                        leaf_add((-b + sq) / den);
                        leaf_add((-b - sq) / den);
                    }
                    break;
                }
                case RT_PLANE: {  // plane.rs:51-74
                    if (!(fabs(ld.y) < RT_EPS)) leaf_add(-lo.y / ld.y);
                    break;
                }
                case RT_CUBE: {  // cube.rs:77-98
                    double tmin, tmax;
                    if (__builtin_expect(wave_ballot(!all_axes_finite(ld)) == 0ull, 1))
                        slabs_fast(lo, ld, reciprocal(ld), -1.0, -1.0, -1.0, 1.0, 1.0, 1.0, tmin, tmax);
                    else
                        slabs(lo, ld, -1.0, -1.0, -1.0, 1.0, 1.0, 1.0, tmin, tmax);
                    if (!(tmin > tmax)) {
                        leaf_add(tmin);
                        leaf_add(tmax);
                    }
                    break;
                }
                case RT_CYLINDER:
                case RT_CONE: {  // cylinder.rs:62-153, cone.rs:63-163
                    const bool cone = type == RT_CONE;
                    const double mn = r0;
                    const double mx = r1;
                    const bool closed = (kind & T_CLOSED) != 0;
                    double a, b, c;
                    bool walls, done = false;
                    if (cone) {
                        a = (ld.x * ld.x) - (ld.y * ld.y) + (ld.z * ld.z);
                        b = (2.0 * lo.x * ld.x) - (2.0 * lo.y * ld.y) + (2.0 * lo.z * ld.z);
                        c = (lo.x * lo.x) - (lo.y * lo.y) + (lo.z * lo.z);
                        walls = !(fabs(a) < RT_EPS);
                        if (!walls) {
                            if (fabs(b) < RT_EPS) {
                                done = true;  // misses walls and caps (cone.rs:119-121)
                            } else {
                                leaf_add(-c / (2.0 * b));
                            }
                        }
                    } else {
                        a = ld.x * ld.x + ld.z * ld.z;
                        b = (2.0 * lo.x * ld.x) + (2.0 * lo.z * ld.z);
                        c = (lo.x * lo.x) + (lo.z * lo.z) - 1.0;
                        walls = !(a < RT_EPS);
                    }
                    if (!done) {
                        if (walls) {
                            double disc = (b * b) - (4.0 * a * c);
                            if (disc >= 0.0) {
                                double sq = sqrt(disc);
                                double t0 = (-b - sq) / (2.0 * a);
                                double t1 = (-b + sq) / (2.0 * a);
                                if (t0 > t1) {
                                    double tmp = t0;
                                    t0 = t1;
                                    t1 = tmp;
                                }
                                double y0 = lo.y + t0 * ld.y;
                                if (mn < y0 && y0 < mx) leaf_add(t0);
                                double y1 = lo.y + t1 * ld.y;
                                if (mn < y1 && y1 < mx) leaf_add(t1);
                            }
                        }
                        if (closed && !(fabs(ld.y) < RT_EPS)) {  // intersect_caps
                            double lim0 = cone ? fabs(mn) : 1.0;
                            double lim1 = cone ? fabs(mx) : 1.0;
                            double t0 = (mn - lo.y) / ld.y;
                            double x = lo.x + t0 * ld.x, z = lo.z + t0 * ld.z;
                            if (((x * x) + (z * z)) <= lim0) leaf_add(t0);
                            double t1 = (mx - lo.y) / ld.y;
                            x = lo.x + t1 * ld.x;
                            z = lo.z + t1 * ld.z;
                            if (((x * x) + (z * z)) <= lim1) leaf_add(t1);
                        }
                    }
                    break;
                }
                default: {  // RT_TRIANGLE, RT_SMOOTH_TRIANGLE: triangle.rs:83-111, smooth_triangle.rs:101-129
                    V3 p1 = mk(q[0], q[1], q[2]);
                    V3 e1 = mk(q[3], q[4], q[5]);
                    V3 e2 = mk(q[6], q[7], q[8]);
                    V3 dir_cross_e2 = cross(ld, e2);
                    double det = dot(e1, dir_cross_e2);
                    if (!(fabs(det) < RT_EPS)) {
                        double f = 1.0 / det;
                        V3 p1_to_origin = lo - p1;
                        double u = f * dot(p1_to_origin, dir_cross_e2);
                        if (!(u < 0.0 || u > 1.0)) {
                            V3 origin_cross_e1 = cross(p1_to_origin, e1);
                            double v = f * dot(ld, origin_cross_e1);
                            if (!(v < 0.0 || (u + v) > 1.0)) {
                                double t = f * dot(e2, origin_cross_e1);
                                leaf_add(t);  // u and v are recomputed when the hit is shaded
                            }
                        }
                    }
                    break;
                }
            }
        }
        // outside the divergent region on purpose: lanes that skipped the leaf have an empty
        // accumulator, and the hit state is not duplicated around the branch
        if (!CSGM || csg_depth == 0) commit_leaf();
        i = i + 1;
        if (early && wave_ballot(i >= skip_until) == 0ull) {
            i = wave_min(skip_until);  // INT_MAX when every lane is decided: the loop ends
        }
#ifdef RT_PROFILE
        h.t_leaf += __builtin_amdgcn_s_memtime() - pl0;
#endif
    }
    __builtin_amdgcn_s_setprio(0);
    if (!TRACK) h.found = h.node >= 0;
}

// Shape::normal_to_world (shape.rs:83-98): per level inverse-transpose, w := 0, normalise, then
// the parent.  Identity levels skip the product but not the normalisation; once the magnitude is
// exactly 1.0 further identity levels cannot change the vector, so the walk jumps to the next
// enclosing non-identity group.
// m = |n|, n = n / m for a vector that has been normalised before.  Its squared length is then
// very often exactly 1.0 (always for the axis-parallel normals of cubes and planes), the root of
// 1.0 is 1.0 and x / 1.0 is x: when that holds for every lane that takes part, the root and the three
// divisions are left out.
DI void renormalise(V3& n, double& m) {
    const double dd = dot(n, n);
    if (wave_ballot(dd != 1.0) != 0ull) {
        m = sqrt(dd);
        n = n / m;
    } else {
        m = 1.0;
    }
}
template <bool XF>
DI V3 normal_to_world(const DevScene& S, const RT_CONST SRec* r, V3 n) {
    if (r->flags & S_HAS_XFORM) n = xnormal(r->inv_t, n);
    double m = magnitude(n);
    n = n / m;
    const int n_levels = r->n_levels;
    int level = 1;
    if (!XF) {  // only identity ancestors: normalise until the length is exactly 1
        while (level < n_levels && m != 1.0) {
            renormalise(n, m);
            ++level;
        }
        return n;
    }
#if RT_ROLL_NORMAL
#pragma unroll 1
#else
#pragma unroll
#endif
    for (int k = 0; k <= RT_MAX_XFORM_NEST; ++k) {
        const int xf = k < RT_MAX_XFORM_NEST ? r->anc_xf[k] : -1;
        const int stop = xf >= 0 ? r->anc_level[k < RT_MAX_XFORM_NEST ? k : 0] : n_levels;
        while (level < stop && m != 1.0) {  // identity ancestors: normalise only
            renormalise(n, m);
            ++level;
        }
        if (xf < 0) break;
        n = xnormal(S.xforms[xf].inv_t, n);
        m = magnitude(n);
        n = n / m;
        level = stop + 1;
    }
    return n;
}

// Shape::normal_at (shape.rs:52-59) for the point where the ray (ro, rd) hits the leaf `sr`: the point
// to object space, the leaf's local_normal_at, and the normal back to world space.
template <bool XF>
DI V3 normal_at(const DevScene& S, const RT_CONST SRec* sr, V3 ro, V3 rd, V3 point) {
    const int type = sr->type;
    const int sflags = sr->flags;
    const int data = sr->tri;
    V3 lp = to_object<XF>(S, sr, point);
    V3 ln;
    switch (type) {
        case RT_SPHERE:  // sphere.rs:109-111
            ln = lp;
            break;
        case RT_PLANE:  // plane.rs:105-107: the plane's own inverse times (0, 1, 0)
            ln = (sflags & S_HAS_XFORM) ? mk(sr->inv[1], sr->inv[5], sr->inv[9]) : mk(0.0, 1.0, 0.0);
            break;
        case RT_CUBE: {  // cube.rs:129-142
            double ax = fabs(lp.x), ay = fabs(lp.y), az = fabs(lp.z);
            double maxc = fmax(fmax(ax, ay), az);
            if (maxc == ax)
                ln = mk(lp.x, 0.0, 0.0);
            else if (maxc == ay)
                ln = mk(0.0, lp.y, 0.0);
            else
                ln = mk(0.0, 0.0, lp.z);
            break;
        }
        case RT_CYLINDER:
        case RT_CONE: {  // cylinder.rs:184-197, cone.rs:194-213
            double mn = sr->r0, mx = sr->r1;
            double dist = lp.x * lp.x + lp.z * lp.z;
            if (dist < 1.0 && lp.y >= mx - RT_EPS) {
                ln = mk(0.0, 1.0, 0.0);
            } else if (dist < 1.0 && lp.y <= mn + RT_EPS) {
                ln = mk(0.0, -1.0, 0.0);
            } else if (type == RT_CYLINDER) {
                ln = mk(lp.x, 0.0, lp.z);
            } else {
                double y = sqrt(dist);
                ln = mk(lp.x, lp.y > 0.0 ? -y : y, lp.z);
            }
            break;
        }
        case RT_TRIANGLE:  // triangle.rs:142-144
            ln = mk(S.trin[data].n1[0], S.trin[data].n1[1], S.trin[data].n1[2]);
            break;
        default: {  // smooth_triangle.rs:160-162
            // The hit's u and v (smooth_triangle.rs:101-129) are not carried through the
            // traversal: the same operations on the same ray give the same bits here.
            V3 lo = ro, ld = rd;
            if (XF) {
#if RT_ROLL_SMOOTH
#pragma unroll 1
                for (int k = sr->n_anc - 1; k >= 0; --k) {
                    const RT_CONST double* m = S.xforms[sr->anc_xf[k]].inv;
                    lo = xpoint(m, lo);
                    ld = xvec(m, ld);
                }
#else
#pragma unroll
                for (int k = RT_MAX_XFORM_NEST - 1; k >= 0; --k)
                    if (sr->anc_xf[k] >= 0) {
                        const RT_CONST double* m = S.xforms[sr->anc_xf[k]].inv;
                        lo = xpoint(m, lo);
                        ld = xvec(m, ld);
                    }
#endif
            }
            if (sflags & S_HAS_XFORM) {
                lo = xpoint(sr->inv, lo);
                ld = xvec(sr->inv, ld);
            }
            V3 tp1 = mk(S.tris[data].p1[0], S.tris[data].p1[1], S.tris[data].p1[2]);
            V3 te1 = mk(S.tris[data].e1[0], S.tris[data].e1[1], S.tris[data].e1[2]);
            V3 te2 = mk(S.tris[data].e2[0], S.tris[data].e2[1], S.tris[data].e2[2]);
            V3 dir_cross_e2 = cross(ld, te2);
            double tf = 1.0 / dot(te1, dir_cross_e2);
            V3 p1_to_origin = lo - tp1;
            double hit_u = tf * dot(p1_to_origin, dir_cross_e2);
            double hit_v = tf * dot(ld, cross(p1_to_origin, te1));
            V3 a1 = mk(S.trin[data].n1[0], S.trin[data].n1[1], S.trin[data].n1[2]);
            V3 a2 = mk(S.trin[data].n2[0], S.trin[data].n2[1], S.trin[data].n2[2]);
            V3 a3 = mk(S.trin[data].n3[0], S.trin[data].n3[1], S.trin[data].n3[2]);
            ln = (a2 * hit_u) + (a3 * hit_v) + (a1 * (1.0 - hit_u - hit_v));
            break;
        }
    }
    return normal_to_world<XF>(S, sr, ln);
}

// Improved-noise permutation (Ken Perlin, 2002), as used by patterns/perturbed.rs:8-34.
__constant__ unsigned char PERLIN_PERM[256] = {
    151, 160, 137, 91,  90,  15,  131, 13,  201, 95,  96,  53,  194, 233, 7,   225, 140, 36,  103,
    30,  69,  142, 8,   99,  37,  240, 21,  10,  23,  190, 6,   148, 247, 120, 234, 75,  0,   26,
    197, 62,  94,  252, 219, 203, 117, 35,  11,  32,  57,  177, 33,  88,  237, 149, 56,  87,  174,
    20,  125, 136, 171, 168, 68,  175, 74,  165, 71,  134, 139, 48,  27,  166, 77,  146, 158, 231,
    83,  111, 229, 122, 60,  211, 133, 230, 220, 105, 92,  41,  55,  46,  245, 40,  244, 102, 143,
    54,  65,  25,  63,  161, 1,   216, 80,  73,  209, 76,  132, 187, 208, 89,  18,  169, 200, 196,
    135, 130, 116, 188, 159, 86,  164, 100, 109, 198, 173, 186, 3,   64,  52,  217, 226, 250, 124,
    123, 5,   202, 38,  147, 118, 126, 255, 82,  85,  212, 207, 206, 59,  227, 47,  16,  58,  17,
    182, 189, 28,  42,  223, 183, 170, 213, 119, 248, 152, 2,   44,  154, 163, 70,  221, 153, 101,
    155, 167, 43,  172, 9,   129, 22,  39,  253, 19,  98,  108, 110, 79,  113, 224, 232, 178, 185,
    112, 104, 218, 246, 97,  228, 251, 34,  242, 193, 238, 210, 144, 12,  191, 179, 162, 241, 81,
    51,  145, 235, 249, 14,  239, 107, 49,  192, 214, 31,  181, 199, 106, 157, 184, 84,  204, 176,
    115, 121, 50,  45,  127, 4,   150, 254, 138, 236, 205, 93,  222, 114, 67,  29,  24,  72,  243,
    141, 128, 195, 78,  66,  215, 61,  156, 180};
DI long long perm(long long i) { return (long long)PERLIN_PERM[i & 255]; }

// Rust `as i64` on finite moderate values; NaN -> 0.
DI long long to_i64(double v) { return (v != v) ? 0ll : (long long)v; }

DI double fade(double t) { return t * t * t * (t * (t * 6.0 - 15.0) + 10.0); }  // perturbed.rs:59-61
DI double lerp(double t, double a, double b) { return a + t * (b - a); }        // perturbed.rs:63-65
DI double grad(long long hash, double x, double y, double z) {                  // perturbed.rs:68-82
    long long hh = hash & 15;
    double u = hh < 8 ? x : y;
    double v = hh < 4 ? y : ((hh == 12 || hh == 14) ? x : z);
    return ((hh & 1) == 0 ? u : -u) + ((hh & 2) == 0 ? v : -v);
}
DI double perlin(V3 p) {  // perturbed.rs:85-139
    double fx = floor(p.x), fy = floor(p.y), fz = floor(p.z);
    long long X = to_i64(fx) & 255, Y = to_i64(fy) & 255, Z = to_i64(fz) & 255;
    double x = p.x - fx, y = p.y - fy, z = p.z - fz;
    double u = fade(x), v = fade(y), w = fade(z);
    long long A = perm(X) + Y, AA = perm(A) + Z, AB = perm(A + 1) + Z;
    long long B = perm(X + 1) + Y, BA = perm(B) + Z, BB = perm(B + 1) + Z;
    return lerp(w,
                lerp(v, lerp(u, grad(perm(AA), x, y, z), grad(perm(BA), x - 1.0, y, z)),
                     lerp(u, grad(perm(AB), x, y - 1.0, z), grad(perm(BB), x - 1.0, y - 1.0, z))),
                lerp(v, lerp(u, grad(perm(AA + 1), x, y, z - 1.0), grad(perm(BA + 1), x - 1.0, y, z - 1.0)),
                     lerp(u, grad(perm(AB + 1), x, y - 1.0, z - 1.0),
                          grad(perm(BB + 1), x - 1.0, y - 1.0, z - 1.0))));
}

// Pattern::local_pattern_at on the pattern tree (pattern.rs:16-22 and the pattern_at
// implementations), as an explicit-stack walk: selecting patterns (striped, ring, checker,
// perturbed) continue into one child; gradient and blended evaluate both children.
DI V3 pattern_eval(const DevScene& S, int root, V3 p) {
    int st_node[RT_PAT_STACK];
    int st_state[RT_PAT_STACK];
    V3 st_p[RT_PAT_STACK];
    V3 st_c[RT_PAT_STACK];
    int sp = 0;
    int cur = root;
    V3 result = mk(0.0, 0.0, 0.0);
    for (;;) {
        // descend
        for (;;) {
            int type = S.pats[cur].type;
            int xf = S.pats[cur].xform;
            if (xf >= 0) p = xpoint(S.patx + (long long)xf * 12, p);
            if (type == RT_PAT_SOLID) {
                result = mk(S.pats[cur].color[0], S.pats[cur].color[1], S.pats[cur].color[2]);
                break;
            } else if (type == RT_PAT_TEST) {
                result = p;
                break;
            } else if (type == RT_PAT_STRIPED) {  // striped.rs:36-45
                cur = (to_i64(floor(p.x)) % 2 == 0) ? S.pats[cur].a : S.pats[cur].b;
            } else if (type == RT_PAT_RING) {  // ring.rs:32-44
                double xs = p.x * p.x, zs = p.z * p.z;
                cur = (to_i64(sqrt(xs + zs)) % 2 == 0) ? S.pats[cur].a : S.pats[cur].b;
            } else if (type == RT_PAT_CHECKER) {  // checker.rs:36-49
                double s = floor(p.x) + floor(p.y) + floor(p.z);
                cur = (to_i64(s) % 2 == 0) ? S.pats[cur].a : S.pats[cur].b;
            } else if (type == RT_PAT_PERTURBED) {  // perturbed.rs:142-158
                double nz = perlin(p) * S.pats[cur].scale;
                p = mk(p.x + nz, p.y + nz, p.z + nz);
                cur = S.pats[cur].a;
            } else {  // gradient / blended: former first
                if (sp >= RT_PAT_STACK) {  // cannot happen: nesting is validated on the host
                    result = mk(0.0, 0.0, 0.0);
                    break;
                }
                st_node[sp] = cur;
                st_state[sp] = 0;
                st_p[sp] = p;
                ++sp;
                cur = S.pats[cur].a;
            }
        }
        // ascend
        for (;;) {
            if (sp == 0) return result;
            int f = sp - 1;
            if (st_state[f] == 0) {
                st_c[f] = result;
                st_state[f] = 1;
                cur = S.pats[st_node[f]].b;
                p = st_p[f];
                break;  // descend into the latter pattern
            }
            V3 fc = st_c[f], lc = result;
            if (S.pats[st_node[f]].type == RT_PAT_GRADIENT) {  // gradient.rs:36-47
                V3 distance = lc - fc;
                double fraction = st_p[f].x - floor(st_p[f].x);
                result = fc + (distance * fraction);
            } else {  // blended.rs:36-44
                result = (fc + lc) / 2.0;
            }
            --sp;
        }
    }
}

// World::schlick (world.rs:279-302), including its use of cos() where the textbook has sqrt().
DI double schlick(double cosv, double n1, double n2) {
    if (n1 > n2) {
        double n = n1 / n2;
        double sin2_t = (n * n) * (1.0 - (cosv * cosv));
        if (sin2_t > 1.0) return 1.0;
        cosv = cos(1.0 - sin2_t);
    }
    double q = (n1 - n2) / (n1 + n2);
    double r0 = q * q;
    double m = 1.0 - cosv;
    double m2 = m * m;
    double m5 = m * (m2 * m2);
    return r0 + (1.0 - r0) * m5;
}

// Canvas::clamp (canvas.rs:50-60): 255 * c, saturated to [0, 255], rounded up; NaN gives 0 like
// Rust's saturating `as u8`.
DI unsigned clamp_u8(double c) {
    double x = 255.0 * c;
    if (x > 255.0) return 255u;
    if (!(x >= 0.0)) return 0u;
    return (unsigned)ceil(x);
}

// One suspended color_at call that is waiting for its reflected / refracted colour
// (World::shade_hit, world.rs:162-196).
// The surface colours of the frame (one per light, in light order) are kept beside it, packed by the
// scene's number of lights: see `surfs` in rt_kernel.
struct Frame {
    double refl[3];                 // reflected colour * reflective (written once it is known)
    double rray[6];                 // pending refraction ray (under_point, direction); only with a refraction.  Once the
                                    // refraction has been handed to another lane, rray[0..2] receives its colour.
    double k;                       // Schlick reflectance; only when state bit2 is set
    int node;                       // the hit leaf: its reflective / transparency are re-read on return
    int state;                      // F_* bits
};
enum {
    F_WAIT_REFR = 1,   // waiting for the refracted colour (else for the reflected one)
    F_REFR_TODO = 2,   // there is a refraction ray that has not been traced to the end yet
    F_SCHLICK = 4,     // combine with the Schlick factor k
    F_HANDED = 8,      // the refraction subtree is being traced by another lane of the wavefront
    F_ARRIVED = 16     // ... and its colour has arrived in rray[0..2]
};
// Work sharing inside a wavefront.  A pixel whose hits are both reflective and transparent is a
// binary tree of up to 63 color_at calls; a lane walks it depth first, one call per round of the
// wavefront, so such a lane keeps its wavefront alive long after the other lanes have finished.
// Lanes without work (their pixel is done, or they never had one: see DevJob::split) therefore take
// over refraction subtrees of busy lanes.  Per lane, packed in one word:
//   bits 0-2   depth of the subtree's root call below the primary ray (remaining = max_depth - depth - sp)
//   bits 3-8   lane that handed the subtree over, bits 9-11 the frame of that lane which waits for its colour
//   bit 12     this lane traces a handed-over subtree (else its own pixel)
//   bits 13-14 L_WAITING: the top frame waits for a handed-over subtree; L_HOLDING: a finished subtree's colour
//              waits (in frames[0].refl) to be delivered
//   bits 16-20 frames of this lane whose refraction ray has neither been traced nor handed over
// A subtree's colour is computed by the same operations on the same values whichever lane does
// it, and the combination stays with the frame, so the result is bit for bit the depth-first one.
enum { L_DEPTH = 7, L_FROM_SHIFT = 3, L_FRAME_SHIFT = 9, L_HELPER = 1 << 12, L_WAITING = 1 << 13, L_HOLDING = 1 << 14,
       L_TODO_SHIFT = 16 };

// ((acc + surface) + reflected [* k]) + refracted [* (1 - k)] per light, world.rs:183-192.
DI V3 combine(const Frame& f, const double* surf, int n_lights, V3 refl, V3 refr) {
    V3 acc = mk(0.0, 0.0, 0.0);
    const bool use_k = (f.state & 4) != 0;
    V3 a = use_k ? refl * f.k : refl;
    V3 b = use_k ? refr * (1.0 - f.k) : refr;
    for (int l = 0; l < n_lights; ++l) {
        V3 s = mk(surf[l * 3], surf[l * 3 + 1], surf[l * 3 + 2]);
        acc = acc + s + a + b;
    }
    return acc;
}

// Wavefronts per SIMD the register allocation aims for.  Five (96 VGPRs) fit once the shading state
// waits in LDS during the shadow walks (three spilled registers, saved before the bounce loop and
// restored after it); the variants for scenes with CSG shapes would spill forty and stay at four.
#ifndef RT_MIN_WAVES
#define RT_MIN_WAVES 5
#endif
#ifndef RT_MIN_WAVES_CSG
#define RT_MIN_WAVES_CSG 4
#endif
#ifndef RT_BLOCK_WAVES
#define RT_BLOCK_WAVES 1  // wavefronts (8x8 tiles) per workgroup; one gives the dispatcher the finest grain
#endif
#define RT_BLOCK (RT_BLOCK_WAVES * 64)
// Camera::ray_for_pixel (camera.rs:47-66)
DI void primary_ray(const DevJob& J, int px, int py, V3& ro, V3& rd) {
    double x_offset = ((double)px + 0.5) * J.pixel_size;
    double y_offset = ((double)py + 0.5) * J.pixel_size;
    double world_x = J.half_width - x_offset;
    double world_y = J.half_height - y_offset;
    V3 pixel = xpoint(J.inv, mk(world_x, world_y, -1.0));
    V3 origin = mk(J.inv[3], J.inv[7], J.inv[11]);
    V3 dirv = pixel - origin;
    double m = magnitude(dirv);
    ro = origin;
    rd = dirv / m;
}

// Dispatch order.  Workgroups start in index order and the frame ends when the last wavefront
// does; if the expensive tiles (deep reflection / refraction recursion) come last the GPU drains
// while a few long wavefronts finish.  A pre-pass traces the centre ray of every tile and puts
// the tile in a cost class by what that ray hits (2 transparent, 1 reflective, 0 neither or
// nothing); a tile then takes the highest class among itself and its eight neighbours (the centre
// of a tile at the edge of a mirror sees the background, but some of its pixels do not: such a tile
// started among the last ones is what the whole frame then waits for).  The render kernel takes the
// tiles class 2 first.  This only permutes the order in which tiles are rendered.
template <bool XF, bool CULL, bool CSGM, bool GROW = false>
__global__ void __launch_bounds__(64) tile_class_kernel(DevScene S, DevJob J, long long n_tiles, int probe_bounces,
                                                        int* __restrict__ tile_class) {
    const int lane = threadIdx.x;
    const long long tile = (long long)blockIdx.x * 64 + lane;
    const bool active = tile < n_tiles;
    const long long tt = active ? tile : 0;
    const int tx = (int)(tt % J.tiles_x), lty = (int)(tt / J.tiles_x);
    int px = tx * RT_TILE + RT_TILE / 2, py = ((lty + J.tile_row0) * J.world_size + J.rank) * RT_TILE + RT_TILE / 2;
    px = px < J.width ? px : J.width - 1;
    py = py < J.height ? py : J.height - 1;
    V3 ro, rd;
    primary_ray(J, px, py, ro, rd);
    Hit h;
    trace<MODE_CLOSEST, XF, CULL, CSGM, GROW>(S, active, ro, rd, 0.0, h);
    int cls = 0;
    if (active && h.found) {
        const RT_CONST SRec* sr = S.shade + h.node;
        cls = sr->transp != 0.0 ? 2 : (sr->refl != 0.0 ? 1 : 0);
    }
    // A mirror that shows a transparent surface is as expensive as the surface itself (the pixel's
    // tree of color_at calls forks one level further down): follow the reflection of the centre ray.
    // Only a classification: it decides when and in how many strips a tile is rendered, not a pixel.
    for (int bounce = 0; bounce < probe_bounces; ++bounce) {
        const bool follow = active && cls == 1 && h.found;
        if (wave_ballot(follow) == 0ull) break;
        if (follow) {
            const RT_CONST SRec* sr = S.shade + h.node;
            const V3 point = ro + (rd * h.t);
            V3 n = normal_at<XF>(S, sr, ro, rd, point);
            if (dot(n, -rd) < 0.0) n = -n;
            ro = point + (n * RT_EPS);
            rd = reflect(rd, n);
        }
        trace<MODE_CLOSEST, XF, CULL, CSGM, GROW>(S, follow, ro, rd, 0.0, h);
        if (follow) {
            if (!h.found)
                h.node = -1;
            else if (S.shade[h.node].transp != 0.0)
                cls = 2;
            else if (S.shade[h.node].refl == 0.0)
                h.found = false;  // the chain of mirrors ends here
        }
    }
    if (active) tile_class[tile] = cls;
}

// The cheap classifier for scenes whose rays are too slow to trace in a pre-pass (meshes): class 1 when
// the line of the tile's centre ray meets the box of one of the world's finite objects, else 0.  With the
// neighbourhood rule of tile_order_kernel the tiles that show an object, its silhouette or the ground
// right next to it are rendered first, and the frame ends on tiles that only see the background, whose
// wavefronts are short: a frame's last wavefronts decide how long the GPU drains (about 80 us per
// frame on the dragon, measured by overlapping consecutive frames).  Only a dispatch order.
constexpr int MAX_TOP_BOXES = 64;
struct TopBoxes {
    double b[MAX_TOP_BOXES][6];
};
__global__ void __launch_bounds__(64) tile_box_kernel(DevJob J, long long n_tiles, const TopBoxes* __restrict__ boxes, int n_boxes,
                                                      int* __restrict__ tile_class) {
    const long long tile = (long long)blockIdx.x * 64 + threadIdx.x;
    if (tile >= n_tiles) return;
    const int tx = (int)(tile % J.tiles_x), lty = (int)(tile / J.tiles_x);
    int px = tx * RT_TILE + RT_TILE / 2, py = ((lty + J.tile_row0) * J.world_size + J.rank) * RT_TILE + RT_TILE / 2;
    px = px < J.width ? px : J.width - 1;
    py = py < J.height ? py : J.height - 1;
    V3 ro, rd;
    primary_ray(J, px, py, ro, rd);
    const double ix = 1.0 / rd.x, iy = 1.0 / rd.y, iz = 1.0 / rd.z;  // (an infinite reciprocal is fine here: min / max drop the NaN of 0 * inf)
    int cls = 0;
    for (int k = 0; k < n_boxes; ++k) {
        const double* b = boxes->b[k];
        const double x0 = (b[0] - ro.x) * ix, x1 = (b[3] - ro.x) * ix, y0 = (b[1] - ro.y) * iy, y1 = (b[4] - ro.y) * iy,
                     z0 = (b[2] - ro.z) * iz, z1 = (b[5] - ro.z) * iz;
        const double tmin = fmax(fmax(fmin(x0, x1), fmin(y0, y1)), fmin(z0, z1));
        const double tmax = fmin(fmin(fmax(x0, x1), fmax(y0, y1)), fmax(z0, z1));
        if (!(tmin > tmax) && !(tmax < 0.0)) cls = 1;
    }
    tile_class[tile] = cls;
}

// One tile per thread, ORDER_BLOCK tiles per workgroup.  The workgroup adds up its tiles per class
// and takes its share of each class's array with one atomic per class (one atomic per wavefront
// serialised two thousand of them on three words and took longer than the classification).
constexpr int ORDER_BLOCK = 1024;
__global__ void __launch_bounds__(ORDER_BLOCK) tile_order_kernel(int tiles_x, int tile_rows, long long n_tiles,
                                                                 const int* __restrict__ tile_class, int* __restrict__ order,
                                                                 unsigned long long* __restrict__ class_count) {
    __shared__ unsigned long long base[ORDER_BLOCK / 64][3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long tile = (long long)blockIdx.x * ORDER_BLOCK + threadIdx.x;
    const bool active = tile < n_tiles;
    int cls = 0;
    if (active) {
        const int tx = (int)(tile % tiles_x), ty = (int)(tile / tiles_x);
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int x = tx + dx, y = ty + dy;
                if (x >= 0 && x < tiles_x && y >= 0 && y < tile_rows) {
                    const int c = tile_class[(long long)y * tiles_x + x];
                    cls = c > cls ? c : cls;
                }
            }
        cls = cls < 0 ? 0 : (cls > 2 ? 2 : cls);
    }
    unsigned long long mine = 0ull;  // lanes of this wavefront in this lane's class
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const unsigned long long mask = wave_ballot(active && cls == c);
        if (lane == 0) base[wave][c] = (unsigned long long)__builtin_popcountll(mask);
        if (cls == c) mine = mask;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int c = threadIdx.x;
        unsigned long long total = 0ull;
        for (int w = 0; w < ORDER_BLOCK / 64; ++w) total += base[w][c];
        unsigned long long at = total ? atomicAdd(&class_count[c], total) : 0ull;
        for (int w = 0; w < ORDER_BLOCK / 64; ++w) {  // counts become starting positions
            const unsigned long long n = base[w][c];
            base[w][c] = at;
            at += n;
        }
    }
    __syncthreads();
    if (active) {
        const long long pos = (long long)base[wave][cls] + __builtin_popcountll(mine & ((1ull << lane) - 1ull));
        if (pos >= 0 && pos < n_tiles) order[(long long)cls * n_tiles + pos] = (int)tile;
    }
}

// Pixel of this lane: column, row within this rank's compact buffer, and local tile row.
// Workgroups are dealt round-robin to the 8 XCDs, so consecutive tiles land on different XCDs and an
// expensive region of the frame is spread over all of them (giving each XCD a contiguous band of
// tiles instead puts the default scene's glass sphere on one XCD and costs 60 % more time).
// Cheap enough to evaluate twice instead of holding the result in registers.
// rows: how many of the tile's rows this wavefront renders (8, or 8 / split for a strip of a class-2
// tile); lanes whose row is beyond that have no pixel.  origin: index of the strip's first pixel in
// the rank's compact buffer (wave-uniform).
DI void lane_pixel(const DevJob& J, int wave, int lane, int& px, int& row_in_tile, int& lty, int& rows, long long& origin) {
    const long long b = (long long)blockIdx.x * RT_BLOCK_WAVES + wave;
    const long long n_tiles = (long long)J.tiles_x * J.local_tile_rows;
    long long tile = b < n_tiles ? b : n_tiles;
    int strip = 0, split = 1;
    rows = RT_TILE;
    if (J.tile_order) {
        // expensive classes first: class 2, then 1, then 0
        long long n2 = (long long)J.tile_class_count[2], n1 = (long long)J.tile_class_count[1];
        n2 = n2 < n_tiles ? n2 : n_tiles;  // the three counts add up to n_tiles; clamped so that no index below can leave its array
        n1 = n1 < n_tiles - n2 ? n1 : n_tiles - n2;
        // as many strips as the extra workgroups allow: J.split, else half of that, ..., else whole tiles
        for (int sp = J.split; sp > 1; sp >>= 1)
            if (n2 * (sp - 1) <= J.extra_blocks) {
                split = sp;
                break;
            }
        const long long n2s = n2 * split;
        if (b < n2s) {
            strip = (int)(b % split);
            rows = RT_TILE / split;
            tile = J.tile_order[2 * n_tiles + b / split];
        } else {
            const long long t = b - n2s + n2;
            tile = t >= n_tiles ? n_tiles : (t < n2 + n1 ? J.tile_order[n_tiles + (t - n2)] : J.tile_order[t - n2 - n1]);
        }
        if (tile < 0 || tile >= n_tiles) tile = n_tiles;  // never trust a table for an address: renders nothing
    }
    const int tx = (int)(tile % J.tiles_x);
    lty = (int)(tile / J.tiles_x);
    px = tx * RT_TILE + (lane & 7);
    row_in_tile = strip * rows + (lane >> 3);
    origin = ((long long)lty * RT_TILE + strip * rows) * J.width + (long long)tx * RT_TILE;
}

// SHARE: lanes without work take over refraction subtrees of busy lanes (see L_*).  Used for frames
// small enough that their time is that of their longest wavefront; on large frames the GPU is busy
// anyway and the bookkeeping only costs.
// GROW: the scene has cubes that are not axis-aligned in the spaces their boxes are tested in (see the
// note on cubes at slabs_f32); only instantiated for the most general combinations of the others.
template <bool TRACK, bool XF, bool CULL, bool CSGM, bool SHARE = false, bool GROW = false>
__global__ void __launch_bounds__(RT_BLOCK, CSGM ? RT_MIN_WAVES_CSG : RT_MIN_WAVES) rt_kernel(DevScene S, DevJob J, double* __restrict__ out,
                                                 unsigned long long* __restrict__ counters,
                                                 unsigned long long* __restrict__ clear_counters) {
    const int lane = threadIdx.x & 63;
    const int wave = RT_BLOCK_WAVES == 1 ? 0 : (int)(threadIdx.x >> 6);
    // The scene keeps two sets of ray counters; this launch clears the set the next launch will
    // use, which saves a memset (and its launch gap) per frame.
    {
        const unsigned word = blockIdx.x * RT_BLOCK + threadIdx.x;
        if (clear_counters && word < COUNTER_SET_WORDS) clear_counters[word] = 0ull;
    }
    // A lane parks its finished pixel in LDS instead of carrying it through the rest of the
    // wavefront's rounds; the tile is written from there at the end.
    __shared__ double tile_lds[RT_BLOCK_WAVES][64 * 3];
    // The slot's address is derived again from the lane number where it is used: computed once up here it
    // (and the other lane-derived addresses of the epilogue) would be spilled across the bounce loop,
    // 1.3 KB of scratch written and read back per wavefront.
    auto pixel_slot = [&]() -> double* { return tile_lds[wave] + fresh_lane() * 3; };
    double* my_pixel = pixel_slot();
    // What is only needed again after a shadow walk waits in LDS meanwhile ([value][lane]: no bank
    // conflicts), which keeps it out of the registers the walk needs: 127 -> 113 VGPRs at four
    // wavefronts per SIMD, and five fit.  6 KB per wavefront; 20 wavefronts fill a CU's 160 KB.
    __shared__ double park_lds[RT_BLOCK_WAVES][12][64];
    double (*const park)[64] = park_lds[wave];
    my_pixel[0] = 0.0;
    my_pixel[1] = 0.0;
    my_pixel[2] = 0.0;

    bool pending;  // this lane has a ray to trace in the coming round
    bool valid;    // this lane owns a pixel (or a caller-supplied ray) and stores its colour
    V3 ro, rd;
    if (J.n_rays > 0) {
        // World::color_at for caller-supplied rays.
        long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
        pending = r < J.n_rays;
        long long rr = pending ? r : 0;
        ro = mk(J.ray_o[rr * 3], J.ray_o[rr * 3 + 1], J.ray_o[rr * 3 + 2]);
        rd = mk(J.ray_d[rr * 3], J.ray_d[rr * 3 + 1], J.ray_d[rr * 3 + 2]);
    } else {
        int px, row_in_tile, lty, rows;  // rows of the tile this wavefront renders (fewer for a strip of a class-2 tile)
        long long origin;
        lane_pixel(J, wave, lane, px, row_in_tile, lty, rows, origin);
        const int py = ((lty + J.tile_row0) * J.world_size + J.rank) * RT_TILE + row_in_tile;
        pending = (lane >> 3) < rows && lty < J.local_tile_rows && px < J.width && py < J.height;
        primary_ray(J, px, py, ro, rd);
    }
    valid = pending;
    int job = 0;  // see L_*: a lane starts on its own pixel, or with nothing

    unsigned n_primary = pending ? 1u : 0u, n_shadow = 0u, n_reflect = 0u, n_refract = 0u;
    Frame frames[RT_MAX_FRAMES];
    // Surface colours of the suspended calls: frame i, light l at (i * L + l) * 3.  Packed by the
    // scene's light count, so that a two-light scene touches 48 bytes per frame whatever the limit is.
    double surfs[RT_MAX_FRAMES * RT_MAX_LIGHTS * 3];
    int sp = 0;  // suspended calls of this lane's subtree; the ray being traced has remaining = max_depth - depth - sp
    const int L = S.n_lights;

#ifdef RT_PROFILE
    unsigned long long prof_t0 = __builtin_amdgcn_s_memtime();
    unsigned long long prof_closest = 0, prof_shadow = 0, prof_visits = 0, prof_traces = 0, prof_rounds = 0;
    unsigned long long prof_prepare = 0, prof_light = 0, prof_return = 0, prof_end = 0, prof_leaf = 0;
#endif
#ifdef RT_PROFILE_MIX
    unsigned long long mix[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define RT_MIX_ADD(hh) do { mix[0] += (hh).n_cull; mix[1] += (hh).a_cull; mix[2] += (hh).n_group; mix[3] += (hh).a_group; \
                            mix[4] += (hh).n_leaf; mix[5] += (hh).a_leaf; mix[6] += (hh).n_leaf_idle; mix[7] += 1; } while (0)
#else
#define RT_MIX_ADD(hh) do { } while (0)
#endif
#ifdef RT_PROFILE_MIX
    unsigned long long mix_rounds = 0, mix_handoffs = 0, mix_lane_rounds = 0;
#endif
    while (wave_ballot(pending || (SHARE && (job & (L_WAITING | L_HOLDING)) != 0)) != 0ull) {
        Hit h;
#ifdef RT_PROFILE_MIX
        ++mix_rounds;
        mix_lane_rounds += __builtin_popcountll(wave_ballot(pending));
#endif
#ifdef RT_PROFILE
        unsigned long long pt = __builtin_amdgcn_s_memtime();
#endif
        trace<MODE_CLOSEST, XF, CULL, CSGM, GROW>(S, pending, ro, rd, 0.0, h);
        RT_MIX_ADD(h);
        if (TRACK) {
            // n1/n2 (world.rs:99-132) are only read when the hit material is transparent
            // (world.rs:185, 243-246): only those lanes run the container pass.
            const bool need = pending && h.found && S.shade[h.node].transp != 0.0;
            if (wave_ballot(need) != 0ull) {
                trace<MODE_CONTAINERS, XF, CULL, CSGM, GROW>(S, need, ro, rd, 0.0, h);
                RT_MIX_ADD(h);
            }
        }
#ifdef RT_PROFILE
        prof_closest += __builtin_amdgcn_s_memtime() - pt;
        prof_visits += h.visits;
        prof_end += h.t_end;
        prof_leaf += h.t_leaf;
        prof_traces += 1;
        prof_rounds += 1;
#endif
        const bool shade = pending && h.found;
        const int node = shade ? h.node : 0;
        const RT_CONST SRec* sr = S.shade + node;

        // ---- World::prepare_computations (world.rs:70-147) --------------------------------
        // Only what the light loop needs stays in registers: over_point, normalv and the
        // pattern colour.  Everything about the secondary rays goes into the lane's frame now.
        V3 over = mk(0.0, 0.0, 0.0), normalv = over, colour = over;
        bool want_refl = false, want_refr = false;
        if (shade) {
            const int sflags = sr->flags;
            V3 point = ro + (rd * h.t);  // Ray::position (ray.rs:24-26)
            V3 eyev = -rd;
            normalv = normal_at<XF>(S, sr, ro, rd, point);  // Shape::normal_at (shape.rs:52-59)
            if (dot(normalv, eyev) < 0.0) normalv = -normalv;
            over = point + (normalv * RT_EPS);

            // ---- which secondary rays this hit needs (world.rs:214-275) ----------------------
            const int remaining = J.max_depth - (SHARE ? (job & L_DEPTH) : 0) - sp;
            const double refl_c = sr->refl, transp = sr->transp;
            want_refl = remaining > 0 && !(fabs(refl_c - 0.0) < RT_EPS);
            V3 refr_dir = mk(0.0, 0.0, 0.0);
            double kk = 0.0;
            bool use_k = false;
            if (TRACK) {
                // Refractive indices of the last container before / after the hit (world.rs:103-132).
                double n1, n2;
                const int x_id = (h.a_id != node) ? h.a_id : h.b_id;  // latest container other than the hit leaf
                const double ri_x = x_id >= 0 ? S.shade[x_id].ri : 1.0;
                if (h.par) {  // the hit leaf is itself an open container: this is an exit
                    n1 = S.shade[h.a_id].ri;
                    n2 = ri_x;
                } else {
                    n1 = ri_x;
                    n2 = sr->ri;
                }
                const double cos_i = dot(eyev, normalv);
                if (remaining > 0 && !(fabs(transp - 0.0) < RT_EPS)) {
                    double n_ratio = n1 / n2;
                    double sin2_t = (n_ratio * n_ratio) * (1.0 - (cos_i * cos_i));
                    if (!(sin2_t > 1.0)) {
                        double cos_t = sqrt(1.0 - sin2_t);
                        refr_dir = normalv * (n_ratio * cos_i - cos_t) - eyev * n_ratio;
                        want_refr = true;
                    }
                }
                use_k = refl_c > 0.0 && transp > 0.0;  // world.rs:185
                if (use_k && (want_refl || want_refr)) kk = schlick(cos_i, n1, n2);
            }
            if (want_refl || want_refr) {
                // Only what the return path cannot recompute is written: most frames belong to
                // plain reflections and hold the surface colours, the node and the state.
                Frame& f = frames[sp];
                f.node = node;
                f.state = (want_refl ? (want_refr ? F_REFR_TODO : 0) : F_WAIT_REFR) | (use_k ? F_SCHLICK : 0);
                if (SHARE && want_refl && want_refr) job |= 1 << (L_TODO_SHIFT + sp);  // a subtree another lane may take
                if (use_k) f.k = kk;
                if (want_refr) {
                    V3 under = point - (normalv * RT_EPS);
                    f.rray[0] = under.x;
                    f.rray[1] = under.y;
                    f.rray[2] = under.z;
                    f.rray[3] = refr_dir.x;
                    f.rray[4] = refr_dir.y;
                    f.rray[5] = refr_dir.z;
                    if (!want_refl) {
                        f.refl[0] = 0.0;
                        f.refl[1] = 0.0;
                        f.refl[2] = 0.0;
                    }
                }
            }
            // Pattern::pattern_at_shape at over_point (phong.rs:68, pattern.rs:8-14)
            if (sflags & S_SOLID) {
                colour = mk(sr->colour[0], sr->colour[1], sr->colour[2]);
            } else {
                colour = pattern_eval(S, sr->pattern, to_object<XF>(S, sr, over));
            }
        }
        const bool has_children = want_refl || want_refr;
#ifdef RT_PROFILE
        prof_prepare += __builtin_amdgcn_s_memtime() - pt;  // includes the closest-hit trace; subtract closest_cycles
        unsigned long long pl0 = __builtin_amdgcn_s_memtime();
#endif

        // ---- per light: World::is_shadowed + Phong::lighting (world.rs:304-326, phong.rs:59-116)
        // no_direct: the material has neither a diffuse nor a specular term and the colour is
        // bounded, so both terms are exact zeros whatever the light does (0 * inf and 0 * NaN
        // would not be).  Read through a pointer the compiler cannot match with the later loads,
        // so that the coefficients are not kept in registers across the shadow traces.
        bool no_direct = false;
        if (shade) {
            const RT_CONST SRec* q = sr;
            asm volatile("" : "+v"(q));
            // A zero specular coefficient makes the specular term an exact zero only while the power is
            // finite (0 * inf is NaN): shininess in [0, 400] and a base of at most 5.4 keep it so.
            // |reflectv . eyev| <= 3 |eyev| for unit light and normal vectors: with the direction's 1-norm at
            // most 1.8 the base stays below 5.4.
            const double shin = q->shin;
            no_direct = q->kd == 0.0 && q->ks == 0.0 && shin >= 0.0 && shin <= 400.0 &&
                        (fabs(rd.x) + fabs(rd.y)) + fabs(rd.z) <= 1.8 && fabs(colour.x) < 1e100 &&
                        fabs(colour.y) < 1e100 && fabs(colour.z) < 1e100;
        }
        V3 acc = mk(0.0, 0.0, 0.0);
        for (int l = 0; l < L; ++l) {
            V3 lpos = mk(S.lights[l].position[0], S.lights[l].position[1], S.lights[l].position[2]);
            V3 lv = lpos - over;
            double distance = magnitude(lv);
            V3 lightv = lv / distance;
            // World::is_shadowed only chooses between `ambient` and `ambient + diffuse + specular`
            // (phong.rs:79-115).  Where the two are the same colour the shadow ray is not traced
            // (it still counts as a ray of the frame): the light is behind the surface, so diffuse
            // and specular are black (phong.rs:90-93), or the material has no diffuse and no
            // specular term, so both are zero.  A zero's sign cannot show: the sums the colour
            // enters start from +0.
            const double ldn_before = dot(lightv, normalv);
            const double big = 1e100;
            const bool light_bounded = fabs(S.lights[l].intensity[0]) < big && fabs(S.lights[l].intensity[1]) < big &&
                                       fabs(S.lights[l].intensity[2]) < big;
            const bool shadow_matters =
                shade && !(ldn_before < 0.0) && !(no_direct && light_bounded && ldn_before == ldn_before);
            Hit sh;
#ifdef RT_PROFILE
            unsigned long long ps = __builtin_amdgcn_s_memtime();
#endif
            park[0][lane] = normalv.x;
            park[1][lane] = normalv.y;
            park[2][lane] = normalv.z;
            park[3][lane] = colour.x;
            park[4][lane] = colour.y;
            park[5][lane] = colour.z;
            park[6][lane] = rd.x;
            park[7][lane] = rd.y;
            park[8][lane] = rd.z;
            park[9][lane] = acc.x;
            park[10][lane] = acc.y;
            park[11][lane] = acc.z;
            asm volatile("" ::: "memory");  // the values must really leave the registers ...
            trace<MODE_SHADOW, XF, CULL, CSGM, GROW>(S, shadow_matters, over, lightv, distance, sh);
            RT_MIX_ADD(sh);
            asm volatile("" ::: "memory");  // ... and come back from LDS, not from a copy the compiler kept
            normalv = mk(park[0][lane], park[1][lane], park[2][lane]);
            colour = mk(park[3][lane], park[4][lane], park[5][lane]);
            rd = mk(park[6][lane], park[7][lane], park[8][lane]);
            acc = mk(park[9][lane], park[10][lane], park[11][lane]);
#ifdef RT_PROFILE
            prof_shadow += __builtin_amdgcn_s_memtime() - ps;
            prof_visits += sh.visits;
            prof_end += sh.t_end;
            prof_leaf += sh.t_leaf;
            prof_traces += (wave_ballot(shadow_matters) != 0ull) ? 1 : 0;
#endif
            if (shade) {
                ++n_shadow;
                bool in_shadow = sh.found && sh.t < distance &&
                                 (S.all_cast_shadow || (S.shade[sh.node].flags & S_CASTS_SHADOW));
                V3 lint = mk(S.lights[l].intensity[0], S.lights[l].intensity[1], S.lights[l].intensity[2]);
                V3 effective = colour * lint;
                V3 ambient = effective * sr->ka;
                V3 surf = ambient;
                if (!in_shadow) {
                    V3 diffuse = mk(0.0, 0.0, 0.0), specular = diffuse;
                    // the cosine is computed again rather than kept in registers across the trace
                    // (the empty asm hides that it is the same value)
                    double nx = normalv.x;
                    asm volatile("" : "+v"(nx));
                    const double ldn = (lightv.x * nx + lightv.y * normalv.y) + lightv.z * normalv.z;
                    if (!(ldn < 0.0)) {
                        diffuse = effective * sr->kd * ldn;
                        V3 rv = reflect(-lightv, normalv);
                        double rde = dot(rv, -rd);
                        if (!(rde <= 0.0)) {
                            // a zero specular coefficient makes the term an exact zero while the power is
                            // finite (see no_direct); skip pow then
                            const double ks = sr->ks, shin = sr->shin;
                            double factor = (ks == 0.0 && shin >= 0.0 && shin <= 400.0 && rde <= 5.4) ? 1.0 : pow(rde, shin);
                            specular = lint * ks * factor;
                        }
                    }
                    surf = ambient + diffuse + specular;
                }
                if (has_children) {
                    double* const fs = surfs + (sp * L + l) * 3;
                    fs[0] = surf.x;
                    fs[1] = surf.y;
                    fs[2] = surf.z;
                } else {
                    acc = acc + surf;  // + black + black: exact no-ops (world.rs:191)
                }
            }
        }

#ifdef RT_PROFILE
        prof_light += __builtin_amdgcn_s_memtime() - pl0;  // includes the shadow traces; subtract shadow_cycles
        unsigned long long pr0 = __builtin_amdgcn_s_memtime();
#endif
        // ---- suspend, or return the colour up the stack ----------------------------------------
        if constexpr (!SHARE) {
            if (pending) {
                V3 value = acc;  // (0,0,0) when nothing was hit (world.rs:211)
                bool returning = true;
                if (has_children) {
                    const Frame& f = frames[sp];
                    if (want_refl) {
                        rd = reflect(rd, normalv);  // reflectv (world.rs:92)
                        ro = over;
                        ++n_reflect;
                    } else {
                        ro = mk(f.rray[0], f.rray[1], f.rray[2]);
                        rd = mk(f.rray[3], f.rray[4], f.rray[5]);
                        ++n_refract;
                    }
                    ++sp;
                    returning = false;
                }
                while (returning) {
                    if (sp == 0) {
                        my_pixel = pixel_slot();
                        my_pixel[0] = value.x;
                        my_pixel[1] = value.y;
                        my_pixel[2] = value.z;
                        pending = false;
                        break;
                    }
                    Frame& f = frames[sp - 1];
                    if ((f.state & F_WAIT_REFR) == 0) {
                        // the reflected colour arrived (world.rs:228-230)
                        V3 refl = value * S.shade[f.node].refl;
                        if (f.state & F_REFR_TODO) {
                            f.refl[0] = refl.x;
                            f.refl[1] = refl.y;
                            f.refl[2] = refl.z;
                            f.state = (f.state & F_SCHLICK) | F_WAIT_REFR;
                            ro = mk(f.rray[0], f.rray[1], f.rray[2]);
                            rd = mk(f.rray[3], f.rray[4], f.rray[5]);
                            ++n_refract;
                            break;  // trace the refraction ray next
                        }
                        value = combine(f, surfs + (sp - 1) * L * 3, L, refl, mk(0.0, 0.0, 0.0));
                    } else {
                        // the refracted colour arrived (world.rs:274)
                        V3 refr = value * S.shade[f.node].transp;
                        value = combine(f, surfs + (sp - 1) * L * 3, L, mk(f.refl[0], f.refl[1], f.refl[2]), refr);
                    }
                    --sp;
                }
            }
        } else {
            V3 value = acc;  // (0,0,0) when nothing was hit (world.rs:211)
            bool returning = pending;
            if (pending && has_children) {
                const Frame& f = frames[sp];
                if (want_refl) {
                    rd = reflect(rd, normalv);  // reflectv (world.rs:92)
                    ro = over;
                    ++n_reflect;
                } else {
                    ro = mk(f.rray[0], f.rray[1], f.rray[2]);
                    rd = mk(f.rray[3], f.rray[4], f.rray[5]);
                    ++n_refract;
                }
                ++sp;
                returning = false;
            }
            for (;;) {
                // Pops frames while their colours are complete.  A lane ends the loop with a ray to trace
                // (pending), with its subtree finished (pixel parked, or L_HOLDING), or blocked on a
                // subtree another lane traces (L_WAITING).
                while (returning) {
                    if (sp == 0) {
                        if (SHARE && (job & L_HELPER)) {
                            frames[0].refl[0] = value.x;  // free now: the colour waits here for its delivery
                            frames[0].refl[1] = value.y;
                            frames[0].refl[2] = value.z;
                            job |= L_HOLDING;
                        } else {
                            my_pixel = pixel_slot();
                            my_pixel[0] = value.x;
                            my_pixel[1] = value.y;
                            my_pixel[2] = value.z;
                        }
                        pending = false;
                        break;
                    }
                    Frame& f = frames[sp - 1];
                    if ((f.state & F_WAIT_REFR) == 0) {
                        // the reflected colour arrived (world.rs:228-230)
                        V3 refl = value * S.shade[f.node].refl;
                        if (f.state & F_REFR_TODO) {
                            f.refl[0] = refl.x;
                            f.refl[1] = refl.y;
                            f.refl[2] = refl.z;
                            if (SHARE && (f.state & F_HANDED)) {
                                if (f.state & F_ARRIVED) {
                                    V3 refr = mk(f.rray[0], f.rray[1], f.rray[2]) * S.shade[f.node].transp;  // world.rs:274
                                    value = combine(f, surfs + (sp - 1) * L * 3, L, refl, refr);
                                    --sp;
                                    continue;
                                }
                                f.state |= F_WAIT_REFR;
                                job |= L_WAITING;  // nothing to trace until the other lane delivers
                                pending = false;
                                break;
                            }
                            f.state = (f.state & F_SCHLICK) | F_WAIT_REFR;
                            if (SHARE) job &= ~(1 << (L_TODO_SHIFT + sp - 1));
                            ro = mk(f.rray[0], f.rray[1], f.rray[2]);
                            rd = mk(f.rray[3], f.rray[4], f.rray[5]);
                            ++n_refract;
                            pending = true;
                            break;  // trace the refraction ray next
                        }
                        value = combine(f, surfs + (sp - 1) * L * 3, L, refl, mk(0.0, 0.0, 0.0));
                    } else {
                        // the refracted colour arrived (world.rs:274)
                        V3 refr = value * S.shade[f.node].transp;
                        value = combine(f, surfs + (sp - 1) * L * 3, L, mk(f.refl[0], f.refl[1], f.refl[2]), refr);
                    }
                    --sp;
                }
                returning = false;
                    // ---- finished subtrees go back to the lanes that handed them over -------------------
                const unsigned long long holders = wave_ballot((job & L_HOLDING) != 0);
                if (holders == 0ull) break;
                // The park area is free between the shadow walks; its rows serve as mail slots, one per
                // receiving lane.  One colour per receiver and pass: when two subtrees of one lane finish
                // together, one of them waits for the next pass.
                int* const slot = reinterpret_cast<int*>(park[0]);
                slot[lane] = -1;
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
                const int to = (job >> L_FROM_SHIFT) & 63;
                const int ticket = lane | (((job >> L_FRAME_SHIFT) & 7) << 8);
                if (job & L_HOLDING) slot[to] = ticket;  // several writers: one of them lands
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
                if ((job & L_HOLDING) && slot[to] == ticket) {
                    park[1][to] = frames[0].refl[0];
                    park[2][to] = frames[0].refl[1];
                    park[3][to] = frames[0].refl[2];
                    job = 0;  // delivered: this lane is free again
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
                const int got = slot[lane];
                if (got >= 0) {
                    Frame& f = frames[got >> 8];
                    f.rray[0] = park[1][lane];
                    f.rray[1] = park[2][lane];
                    f.rray[2] = park[3][lane];
                    f.state |= F_ARRIVED;
                    if ((job & L_WAITING) && (got >> 8) == sp - 1) {
                        // this lane was waiting for exactly this colour: go on popping
                        job &= ~L_WAITING;
                        V3 refr = mk(f.rray[0], f.rray[1], f.rray[2]) * S.shade[f.node].transp;  // world.rs:274
                        value = combine(f, surfs + (sp - 1) * L * 3, L, mk(f.refl[0], f.refl[1], f.refl[2]), refr);
                        --sp;
                        returning = true;
                    }
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
                if (wave_ballot(returning) == 0ull) break;
            }
        }
        if (SHARE) {
            // ---- lanes without work take over refraction subtrees of busy lanes --------------------
            const bool offers = (job >> L_TODO_SHIFT) != 0;
            const bool idle = !pending && (job & (L_WAITING | L_HOLDING)) == 0;
            const unsigned long long offer_mask = wave_ballot(offers), idle_mask = wave_ballot(idle);
            if (offer_mask != 0ull && idle_mask != 0ull) {
                const int n_offer = __builtin_popcountll(offer_mask), n_idle = __builtin_popcountll(idle_mask);
                const int n = n_offer < n_idle ? n_offer : n_idle;
#ifdef RT_PROFILE_MIX
                mix_handoffs += n;
#endif
                const unsigned long long below = (1ull << lane) - 1ull;
                const int my_offer = __builtin_popcountll(offer_mask & below), my_idle = __builtin_popcountll(idle_mask & below);
                int* const slot = reinterpret_cast<int*>(park[0]);
                if (idle && my_idle < n) slot[my_idle] = lane;  // the k-th idle lane serves the k-th offer
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
                if (offers && my_offer < n) {
                    const int helper = slot[my_offer] & 63;
                    // the oldest open frame: the largest subtree
                    const int fi = __builtin_ctz((unsigned)(job >> L_TODO_SHIFT));
                    Frame& f = frames[fi];
#pragma unroll
                    for (int k = 0; k < 6; ++k) park[1 + k][helper] = f.rray[k];
                    reinterpret_cast<int*>(park[7])[helper] = ((job & L_DEPTH) + fi + 1) | (lane << L_FROM_SHIFT) | (fi << L_FRAME_SHIFT) | L_HELPER;
                    f.state |= F_HANDED;
                    job &= ~(1 << (L_TODO_SHIFT + fi));
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
                if (idle && my_idle < n) {
                    ro = mk(park[1][lane], park[2][lane], park[3][lane]);
                    rd = mk(park[4][lane], park[5][lane], park[6][lane]);
                    job = reinterpret_cast<int*>(park[7])[lane];
                    sp = 0;
                    pending = true;
                    ++n_refract;
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
            }
        }
#ifdef RT_PROFILE
        prof_return += __builtin_amdgcn_s_memtime() - pr0;
#endif
    }

    // The frame is written once and not read again by this launch: the stores are marked non-temporal so
    // that its lines do not push the wavefronts' recursion stacks out of L2.
    typedef double dv2 __attribute__((ext_vector_type(2)));
    // ---- write the tile ------------------------------------------------------------------------
    // A tile row is 8 pixels = 192 contiguous bytes.  The pixels sit in LDS in lane order, so every
    // lane stores 16-byte pieces and each store instruction covers whole rows (three 8-byte
    // stores per lane with a 24-byte stride would write every line three times).
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): LDS writes of this wave are done
    __builtin_amdgcn_wave_barrier();
    const int lane_w = fresh_lane();  // see pixel_slot
    my_pixel = pixel_slot();
    long long out_index;
    int rows = RT_TILE;    // rows of the tile this wavefront stores
    long long origin = 0;  // first pixel of the tile or strip in the output
    if (J.n_rays > 0) {
        out_index = (long long)blockIdx.x * blockDim.x + (wave * 64 + lane_w);
        origin = (long long)blockIdx.x * blockDim.x + wave * 64;
    } else {
        int px, row_in_tile, lty;
        lane_pixel(J, wave, lane_w, px, row_in_tile, lty, rows, origin);  // again: cheaper than keeping it in registers
        out_index = origin + (long long)(lane_w >> 3) * J.width + (lane_w & 7);
    }
    // every pixel of the tile (or strip) is inside the frame: whole rows can be stored
    const unsigned long long all_lanes = rows >= RT_TILE ? ~0ull : ((1ull << (rows * 8)) - 1ull);
    const bool whole_tile = J.wide_stores && J.n_rays == 0 && wave_ballot(valid) == all_lanes;
    if (J.out_u8) {
        // 24 bytes per tile row: the lanes' bytes are packed in LDS and 6 lanes per row store one dword each.
        __shared__ unsigned tile8_lds[RT_BLOCK_WAVES][48];
        unsigned char* const t8 = reinterpret_cast<unsigned char*>(tile8_lds[wave]);
        unsigned char* const out8 = reinterpret_cast<unsigned char*>(out);
        const unsigned r = clamp_u8(my_pixel[0]), g = clamp_u8(my_pixel[1]), b = clamp_u8(my_pixel[2]);
        if (whole_tile) {
            t8[lane_w * 3] = (unsigned char)r;
            t8[lane_w * 3 + 1] = (unsigned char)g;
            t8[lane_w * 3 + 2] = (unsigned char)b;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            if (lane_w < rows * 6) {
                const int row = lane_w / 6, piece = lane_w % 6;
                __builtin_nontemporal_store(tile8_lds[wave][lane_w],
                                            reinterpret_cast<unsigned*>(out8 + (origin + (long long)row * J.width) * 3 + piece * 4));
            }
        } else if (valid) {
            out8[out_index * 3] = (unsigned char)r;
            out8[out_index * 3 + 1] = (unsigned char)g;
            out8[out_index * 3 + 2] = (unsigned char)b;
        }
    } else if (whole_tile) {
        const double* t = tile_lds[wave];
        // rows of the tile are 8 pixels apart in the lane_w order, so the LDS image is already
        // [row][192 bytes]; piece c (16 bytes) belongs to row c / 12, byte offset (c % 12) * 16
        const long long row_stride = (long long)J.width * 3;  // doubles
        const long long base = origin * 3;
        if (rows == RT_TILE) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {  // 96 pieces: every lane has one, the first 32 lanes a second one
                const int c = lane_w + k * 64;
                if (c < RT_TILE * 12) {
                    const int row = c / 12, piece = c % 12;
                    const dv2 v = *reinterpret_cast<const dv2*>(t + row * 24 + piece * 2);
                    __builtin_nontemporal_store(v, reinterpret_cast<dv2*>(out + base + row * row_stride + piece * 2));
                }
            }
        } else if (lane_w < rows * 12) {  // a strip: at most four rows, 48 pieces
            const int row = lane_w / 12, piece = lane_w % 12;
            const dv2 v = *reinterpret_cast<const dv2*>(t + row * 24 + piece * 2);
            __builtin_nontemporal_store(v, reinterpret_cast<dv2*>(out + base + row * row_stride + piece * 2));
        }
    } else if (valid) {
        out[out_index * 3] = my_pixel[0];
        out[out_index * 3 + 1] = my_pixel[1];
        out[out_index * 3 + 2] = my_pixel[2];
    }
    // Ray statistics: one atomic per wavefront and counter, spread over COUNTER_SHARDS copies so
    // that the 10^5 wavefronts of a frame do not serialise on four words.
    counters += 16 * ((blockIdx.x * (unsigned)RT_BLOCK_WAVES + (unsigned)wave) % COUNTER_SHARDS);
#ifdef RT_PROFILE
    if (lane_w == 0) {
        atomicAdd(&counters[4], __builtin_amdgcn_s_memtime() - prof_t0);
        atomicAdd(&counters[5], prof_closest);
        atomicAdd(&counters[6], prof_shadow);
        atomicAdd(&counters[7], prof_visits);
        atomicAdd(&counters[8], prof_traces);
        atomicAdd(&counters[9], prof_rounds);
        atomicAdd(&counters[10], 1ull);
        atomicAdd(&counters[11], prof_prepare);
        atomicAdd(&counters[12], prof_light);
        atomicAdd(&counters[13], prof_return);
        atomicAdd(&counters[14], prof_end);
        atomicAdd(&counters[15], prof_leaf);
    }
#endif
#ifdef RT_PROFILE_MIX
    if (lane_w == 0) {
        for (int k = 0; k < 8; ++k) atomicAdd(&counters[4 + k], mix[k]);
        atomicAdd(&counters[12], mix_rounds);
        atomicAdd(&counters[13], mix_handoffs);
        atomicAdd(&counters[14], mix_lane_rounds);
        atomicMax(&counters[15], mix_rounds);
    }
#endif
    unsigned c[4] = {n_primary, n_shadow, n_reflect, n_refract};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        unsigned v = c[k];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane_w == 0 && v) atomicAdd(&counters[k], (unsigned long long)v);
    }
}


// ---------------------------------------------------------------------------------------------
// Canvas::to_ppm (src/window/canvas.rs:62-126) on the device, from clamped bytes.
//
// The reference's writer is a state machine per image row (canvas.rs:96-126): a value goes on a new line
// when it would bring the line to 70 characters, a line that reaches 69 is ended at once, a row ends its
// last line.  Two facts make it parallel.  (1) Every value is followed by exactly ONE separator, a space
// or a newline (a line ended after its last value leaves nothing to put before the next one), so a
// value's place in the text is the sum of (digits + 1) over the values before it, whatever the line
// breaks are, and a row's length is that sum over the row.  (2) With c(k) the sum of (digits + 1) over
// the values s .. k of a line that starts at value s, the line holds exactly the values with c(k) <= 70:
// content so far + 1 + digits < 70 (canvas.rs:99) is c(k) - 1 < 70, and a line ended early at 69
// characters (canvas.rs:118) is c(k) = 70, after which no value fits.  A line has 17 to 35 values, so one
// wavefront finds a line's end with one prefix sum over the next 64 values, and its lanes write that
// line's digits and separators; only the walk from line to line is sequential.  One wavefront per row.
// ---------------------------------------------------------------------------------------------
constexpr int PPM_SEG = 8192;  // values of a row staged in LDS at a time (plus the 64 a line's search may look ahead)
struct PpmHeader {
    char text[48];  // "P3\n<width> <height>\n255\n" (canvas.rs:66-72)
    int n;
};
DI unsigned ppm_cost(unsigned v) { return 2u + (v >= 10u ? 1u : 0u) + (v >= 100u ? 1u : 0u); }  // digits + separator
__global__ void __launch_bounds__(64) ppm_measure_kernel(const uint8_t* __restrict__ rgb8, long long n_values, int height,
                                                          unsigned long long* __restrict__ row_bytes) {
    const int y = blockIdx.x, lane = threadIdx.x;
    if (y >= height) return;
    const uint8_t* __restrict__ row = rgb8 + (size_t)y * (size_t)n_values;
    unsigned long long sum = 0;
    for (long long k = lane; k < n_values; k += 64) sum += ppm_cost(row[k]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if (lane == 0) row_bytes[y] = sum;
}
// row_bytes[y] becomes the offset of row y; row_bytes[height] the total.  One wavefront: every lane sums a
// run of rows, the runs' sums are scanned across the lanes, every lane writes its run's offsets.
__global__ void __launch_bounds__(64) ppm_scan_kernel(int height, unsigned long long first, unsigned long long* __restrict__ row_bytes) {
    if (blockIdx.x != 0) return;
    const int lane = threadIdx.x;
    const int per = (height + 63) / 64;
    const int y0 = lane * per < height ? lane * per : height, y1 = y0 + per < height ? y0 + per : height;
    unsigned long long mine = 0;
    for (int y = y0; y < y1; ++y) mine += row_bytes[y];
    unsigned long long incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    unsigned long long at = first + (incl - mine);
    for (int y = y0; y < y1; ++y) {
        const unsigned long long n = row_bytes[y];
        row_bytes[y] = at;
        at += n;
    }
    if (lane == 63) row_bytes[height] = first + incl;
}
__global__ void __launch_bounds__(64) ppm_write_kernel(const uint8_t* __restrict__ rgb8, long long n_values, int height,
                                                        const unsigned long long* __restrict__ row_offset, unsigned long long capacity,
                                                        PpmHeader header, char* __restrict__ text) {
    __shared__ uint8_t win[PPM_SEG + 64];
    const int y = blockIdx.x, lane = threadIdx.x;
    if (y >= height || row_offset[height] > capacity) return;
    if (y == 0 && lane < header.n) text[lane] = header.text[lane];
    const uint8_t* __restrict__ row = rgb8 + (size_t)y * (size_t)n_values;
    char* __restrict__ out = text + row_offset[y];
    long long s = 0;              // first value of the current line (wave-uniform)
    unsigned long long base = 0;  // where that value's text starts within the row's text
    while (s < n_values) {
        const long long seg0 = s;
        const long long left = n_values - seg0;
        const int n_in = (int)(left < PPM_SEG + 64 ? left : PPM_SEG + 64);
        __builtin_amdgcn_s_waitcnt(0xc07f);  // the last line's reads of the window are done
        __builtin_amdgcn_wave_barrier();
        for (int k = lane; k < n_in; k += 64) win[k] = row[seg0 + k];
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        const long long seg_end = left < PPM_SEG ? n_values : seg0 + PPM_SEG;
        while (s < seg_end) {
            const bool in = s + lane < n_values;  // (s - seg0 < PPM_SEG, so the window holds s .. s + 63)
            const unsigned v = in ? win[(int)(s - seg0) + lane] : 0u;
            const unsigned cost = in ? ppm_cost(v) : 1000u;  // nothing beyond the row's end fits a line
            unsigned c = cost;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned o = __shfl_up(c, off, 64);
                if (lane >= off) c += o;
            }
            const int fit = __builtin_popcountll(wave_ballot(c <= 70u));  // >= 1: a value alone always fits
            if (lane < fit) {
                char* at = out + base + (c - cost);
                const unsigned hundreds = v / 100u, tens = (v / 10u) % 10u;
                if (v >= 100u) *at++ = (char)('0' + hundreds);
                if (v >= 10u) *at++ = (char)('0' + tens);
                *at++ = (char)('0' + v % 10u);
                *at = lane == fit - 1 ? '\n' : ' ';
            }
            base += (unsigned)__builtin_amdgcn_readlane((int)c, fit - 1);  // the line's length with its newline
            s += fit;
        }
    }
}

// Device self-test of rcp_exact and div_by: out_fast[i] = div_by(a[i], d[i], rcp_exact(d[i])), or NaN when
// rcp_exact(d[i]) != 1.0 / d[i]; out_ieee[i] = a[i] / d[i].
__global__ void division_selftest_kernel(long long n, const double* a, const double* d, double* out_fast,
                                         double* out_ieee) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        double y = rcp_exact(d[i]);
        out_fast[i] = (y == 1.0 / d[i]) ? div_by(a[i], d[i], y) : __builtin_nan("");
        out_ieee[i] = a[i] / d[i];
    }
}

// ---------------------------------------------------------------------------------------------
// Host side of the C ABI
// ---------------------------------------------------------------------------------------------
thread_local std::string g_error;

int fail(int code, const std::string& msg) {
    g_error = msg;
    return code;
}
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(RT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));            \
    } while (0)

constexpr int TIMING_RING = 256;

// Scratch of rt_render_device_static: one set of ray counters followed by the tile dispatch order
// (three order arrays and the tile classes).  Every GPU the library runs on holds its own copy at its
// own address until the process ends; host code never uses the array's name as an address, only
// the per-device pointer hipGetSymbolAddress returned (static_scratch_for).
constexpr size_t STATIC_COUNTER_BYTES = ((COUNTER_SET_WORDS * sizeof(unsigned long long)) + 255) & ~(size_t)255;
constexpr size_t STATIC_ORDER_BYTES = 4 * (size_t)RT_STATIC_MAX_TILES * sizeof(int);
constexpr size_t STATIC_SCRATCH_BYTES = STATIC_COUNTER_BYTES + STATIC_ORDER_BYTES;
static __device__ __align__(256) unsigned char scratch[STATIC_SCRATCH_BYTES];

constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mutex;
unsigned char* g_scratch_ptr[MAX_DEVICES] = {};

// Address of `scratch` on `device`, which must be the current device.  Looked up once per device.
int static_scratch_for(int device, unsigned char** out) {
    if (device < 0 || device >= MAX_DEVICES) return fail(RT_ERR_INVALID, "device ordinal out of range");
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return fail(RT_ERR_HIP, "hipGetSymbolAddress returned a null address");
        g_scratch_ptr[device] = (unsigned char*)p;
    }
    *out = g_scratch_ptr[device];
    return RT_OK;
}

}  // namespace

struct rt_scene {
    int device = 0;
    DevScene dev{};
    bool any_transparent = false;
    bool any_xform = false;  // the traversal records include T_XFORM entries
    bool any_cull = false;   // ... and T_CULL entries
    bool any_csg = false;    // ... and T_CSG_BEGIN / T_CSG_END pairs
    // The dispatch-order pre-pass pays when tiles differ in recursion depth (materials of more than
    // one cost class) and its own rays are cheap (a short record array; its few hundred wavefronts
    // of incoherent rays would crawl through a mesh).
    bool order_useful = false;
    bool mirror_plane = false;  // some plane is reflective: a large part of a frame can show the scene a second time
    // ... else, when the world has a few finite objects: their boxes in world space, padded, for tile_box_kernel
    int n_top_boxes = 0;
    TopBoxes* d_top_boxes = nullptr;
    std::vector<void*> allocations;
    unsigned long long* d_counters = nullptr;  // two sets of COUNTER_SET_WORDS, used by alternate launches
    int* d_order = nullptr;                     // tiles of the current frame by cost class (3 arrays)
    long long order_cap = 0;
    bool set_clean[2] = {true, true};
    int next_set = 0;
    unsigned long long* last_counters = nullptr;  // the counter set the most recent launch wrote
    unsigned char* static_scratch = nullptr;       // this device's copy of `scratch` (rt_render_device_static)
    double* d_frame = nullptr;  // staging for the synchronous entry points
    size_t frame_bytes = 0;
    double* d_rays = nullptr;
    size_t rays_bytes = 0;
    hipEvent_t ev_start[TIMING_RING] = {};
    hipEvent_t ev_stop[TIMING_RING] = {};
    long long launches = 0;
    hipStream_t last_stream = nullptr;
    rt_build_info build{};         // what rt_scene_create spent and produced
    rtdl::Downloader dl;           // frames to caller memory (the synchronous entry points)
    bool dl_ready = false;
    hipEvent_t frame_done = nullptr;  // the staging frame is complete
    // frames rendered in bands (render_host): two streams that take the bands in turn, one event per band
    hipStream_t band_stream[2] = {nullptr, nullptr};
    hipEvent_t band_done[5] = {};
    hipEvent_t band_clear = nullptr;
};

namespace {

template <class T>
int upload(rt_scene* s, const T* src, size_t count, const RT_CONST T** dst) {
    void* p = nullptr;
    size_t bytes = (count ? count : 1) * sizeof(T);
    HIP_TRY(hipMalloc(&p, bytes));
    s->allocations.push_back(p);
    if (count) HIP_TRY(hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice));
    s->build.table_bytes += count * sizeof(T);
    *dst = (const RT_CONST T*)p;
    return RT_OK;
}

int validate(const rt_scene_desc* d, bool& all_cast, bool& any_transparent) {
    if (!d) return fail(RT_ERR_INVALID, "null scene description");
    if (d->n_nodes < 0 || d->n_xforms < 0 || d->n_aabbs < 0 || d->n_ranges < 0 || d->n_tris < 0 ||
        d->n_materials < 0 || d->n_patterns < 0 || d->n_pattern_xforms < 0 || d->n_lights < 0)
        return fail(RT_ERR_INVALID, "negative table size");
    if ((d->n_nodes && !d->nodes) || (d->n_xforms && !d->xforms) || (d->n_aabbs && !d->aabbs) ||
        (d->n_ranges && !d->ranges) || (d->n_tris && (!d->tris || !d->tri_normals)) ||
        (d->n_materials && !d->materials) || (d->n_patterns && !d->patterns) ||
        (d->n_pattern_xforms && !d->pattern_xforms) || (d->n_lights && !d->lights))
        return fail(RT_ERR_INVALID, "null table with non-zero size");
    if (d->n_lights > RT_MAX_LIGHTS) return fail(RT_ERR_LIMIT, "more than RT_MAX_LIGHTS lights");
    all_cast = true;
    std::vector<int> open;  // the containers around node i, outermost first (nodes are in pre-order)
    for (int i = 0; i < d->n_nodes; ++i) {
        const rt_node& n = d->nodes[i];
        auto bad = [&](const char* what) {
            return fail(RT_ERR_INVALID, "node " + std::to_string(i) + ": " + what);
        };
        if (n.type < RT_GROUP || n.type > RT_CSG) return bad("unknown type");
        while (!open.empty() && d->nodes[open.back()].skip <= i) open.pop_back();
        if (n.parent != (open.empty() ? -1 : open.back())) return bad("parent is not the nearest enclosing group or CSG shape");
        const bool container = n.type == RT_GROUP || n.type == RT_CSG;
        if (n.parent < -1 || n.parent >= i) return bad("parent must precede the node");
        if (n.parent >= 0 && d->nodes[n.parent].type != RT_GROUP && d->nodes[n.parent].type != RT_CSG)
            return bad("parent is not a group or CSG shape");
        if (n.skip <= i || n.skip > d->n_nodes) return bad("bad skip link");
        if (n.parent >= 0 && n.skip > d->nodes[n.parent].skip) return bad("subtree leaves its parent");
        if (!container && n.skip != i + 1) return bad("leaf with a subtree");
        if (n.type == RT_CSG) {
            // exactly two operands: the next node and the one after its subtree (csg.rs:20-21)
            if (i + 1 >= n.skip) return bad("CSG shape without operands");
            const int right = d->nodes[i + 1].skip;
            if (right <= i + 1 || right >= n.skip || right > d->n_nodes) return bad("CSG shape without a right operand");
            if (d->nodes[right].skip != n.skip) return bad("CSG shape with more than two operands");
            if (d->nodes[i + 1].parent != i || d->nodes[right].parent != i) return bad("CSG operand with a wrong parent");
            int nest = 1;
            for (int p = n.parent; p >= 0; p = d->nodes[p].parent)
                if (d->nodes[p].type == RT_CSG) ++nest;
            if (nest > RT_MAX_CSG_NEST) return fail(RT_ERR_LIMIT, "CSG shapes nested too deeply");
            if (nest == 1) {  // outermost: every intersection of its leaves must fit the per-lane list
                int most = 0;
                for (int k = i + 1; k < n.skip && k < d->n_nodes; ++k) {
                    const int t = d->nodes[k].type;
                    most += (t == RT_SPHERE || t == RT_CUBE) ? 2 : (t == RT_CYLINDER || t == RT_CONE) ? 4
                            : (t == RT_PLANE || t == RT_TRIANGLE || t == RT_SMOOTH_TRIANGLE) ? 1 : 0;
                }
                if (most > RT_MAX_CSG_HITS)
                    return fail(RT_ERR_LIMIT, "a CSG shape can produce more than RT_MAX_CSG_HITS intersections per ray");
            }
        }
        if (n.xform < -1 || n.xform >= d->n_xforms) return bad("xform out of range");
        if (n.ctx < -1 || n.ctx >= i) return bad("ctx must precede the node");
        // ctx must be the nearest enclosing group with a transform
        int expect = -1;
        for (int p = n.parent; p >= 0; p = d->nodes[p].parent)
            if (d->nodes[p].xform >= 0) {
                expect = p;
                break;
            }
        if (n.ctx != expect) return bad("ctx is not the nearest transformed ancestor");
        int depth = 0;
        for (int c = n.ctx; c >= 0; c = d->nodes[c].ctx) ++depth;
        if (depth > RT_MAX_XFORM_NEST) return fail(RT_ERR_LIMIT, "transformed groups nested too deeply");
        switch (n.type) {
            case RT_GROUP:
            case RT_CSG:
                if (n.data < 0 || n.data >= d->n_aabbs) return bad("aabb out of range");
                break;
            case RT_CYLINDER:
            case RT_CONE:
                if (n.data < 0 || n.data >= d->n_ranges) return bad("range out of range");
                break;
            case RT_TRIANGLE:
            case RT_SMOOTH_TRIANGLE:
                if (n.data < 0 || n.data >= d->n_tris) return bad("triangle out of range");
                break;
            default:
                break;
        }
        if (!container) {
            if (n.material < 0 || n.material >= d->n_materials) return bad("material out of range");
            if (!(n.flags & RT_NODE_CASTS_SHADOW)) all_cast = false;
        } else {
            open.push_back(i);
        }
    }
    any_transparent = false;
    for (int i = 0; i < d->n_materials; ++i) {
        const rt_material& m = d->materials[i];
        if (m.pattern < 0 || m.pattern >= d->n_patterns)
            return fail(RT_ERR_INVALID, "material " + std::to_string(i) + ": pattern out of range");
        if (m.transparency != 0.0) any_transparent = true;
    }
    for (int i = 0; i < d->n_patterns; ++i) {
        const rt_pattern& p = d->patterns[i];
        auto bad = [&](const char* what) {
            return fail(RT_ERR_INVALID, "pattern " + std::to_string(i) + ": " + what);
        };
        if (p.type < RT_PAT_SOLID || p.type > RT_PAT_TEST) return bad("unknown type");
        if (p.xform < -1 || p.xform >= d->n_pattern_xforms) return bad("xform out of range");
        const bool binary = p.type >= RT_PAT_STRIPED && p.type <= RT_PAT_BLENDED;
        // children must follow their parent, which rules out cycles
        if (binary || p.type == RT_PAT_PERTURBED)
            if (p.a <= i || p.a >= d->n_patterns) return bad("child a must follow the pattern");
        if (binary)
            if (p.b <= i || p.b >= d->n_patterns) return bad("child b must follow the pattern");
    }
    // nesting of two-child evaluating patterns (children have larger indices: go backwards)
    {
        std::vector<int> nest(d->n_patterns, 0);
        for (int i = d->n_patterns - 1; i >= 0; --i) {
            const rt_pattern& p = d->patterns[i];
            int below = 0;
            if (p.type != RT_PAT_SOLID && p.type != RT_PAT_TEST) {
                below = nest[p.a];
                if (p.type != RT_PAT_PERTURBED && nest[p.b] > below) below = nest[p.b];
            }
            nest[i] = below + ((p.type == RT_PAT_GRADIENT || p.type == RT_PAT_BLENDED) ? 1 : 0);
            if (nest[i] > RT_MAX_PATTERN_NEST)
                return fail(RT_ERR_LIMIT, "gradient/blended patterns nested too deeply");
        }
    }
    return RT_OK;
}

// Forward matrix (rows 0..2) of an affine map given rows 0..2 of its inverse.  Only used for
// conservative boxes, so ordinary rounding is fine.
bool affine_forward(const double inv[12], double fwd[12]) {
    const double a = inv[0], b = inv[1], c = inv[2], e = inv[4], f = inv[5], g = inv[6], i = inv[8], j = inv[9],
                 k = inv[10];
    const double det = a * (f * k - g * j) - b * (e * k - g * i) + c * (e * j - f * i);
    if (!(std::fabs(det) > 0.0) || !std::isfinite(det)) return false;
    const double r = 1.0 / det;
    fwd[0] = (f * k - g * j) * r;
    fwd[1] = (c * j - b * k) * r;
    fwd[2] = (b * g - c * f) * r;
    fwd[4] = (g * i - e * k) * r;
    fwd[5] = (a * k - c * i) * r;
    fwd[6] = (c * e - a * g) * r;
    fwd[8] = (e * j - f * i) * r;
    fwd[9] = (b * i - a * j) * r;
    fwd[10] = (a * f - b * e) * r;
    for (int row = 0; row < 3; ++row)
        fwd[row * 4 + 3] = -(fwd[row * 4] * inv[3] + fwd[row * 4 + 1] * inv[7] + fwd[row * 4 + 2] * inv[11]);
    for (int n = 0; n < 12; ++n)
        if (!std::isfinite(fwd[n])) return false;
    return true;
}

// Smallest float that is not below v (v >= 0).
float round_up_f32(double v) {
    float f = (float)v;
    if ((double)f < v) f = std::nextafterf(f, INFINITY);
    return f;
}
// The single-precision image of a box for the device's slabs_f32: per axis the centre rounded to
// nearest and the half width rounded up by at least the centre's rounding, so that the f32 box
// contains the f64 one; and the two factors of the margin that depend on the box: k0 = 2 * 24 u *
// (largest |plane|) + pad0 (times the largest |1 / d| of the ray) and k1 = pad1 (times a bound on
// |o|^2 / |d|).  An infinite box gets an infinite k0: nothing is then decided in single precision.
void fill_box_f32(TRec& r, const double bmin[3], const double bmax[3], double pad0, double pad1) {
    float cw[6], kk[2];
    double reach = 0.0;
    for (int k = 0; k < 3; ++k) {
        const double half = 0.5 * (bmax[k] - bmin[k]), centre = bmin[k] + half;
        float c = (float)centre;
        float w = round_up_f32(std::fabs(half) + std::fabs((double)c - centre));
        if (!std::isfinite(c) || !std::isfinite(w)) {  // unbounded along this axis
            c = 0.0f;
            w = INFINITY;
        }
        cw[2 * k] = c;
        cw[2 * k + 1] = w;
        reach = std::fmax(reach, std::fmax(std::fabs(bmin[k]), std::fabs(bmax[k])));
    }
    const double k0 = (2.86102294921875e-6 * reach + pad0) * (1.0 + 1e-6);
    kk[0] = std::isfinite(k0) && reach <= 1e15 ? round_up_f32(k0) : INFINITY;  // NaN planes too
    kk[1] = round_up_f32(pad1 * (1.0 + 1e-6));
    std::memcpy(&r.p[6], cw, sizeof cw);
    std::memcpy(&r.p[9], kk, sizeof kk);
}

// Fills a T_CULL record for the box [bmin, bmax].  far_term: see NodeBox::far.
bool fill_cull(const double bmin[3], const double bmax[3], double far_term, TRec& out) {
    double size = 0.0, reach = 0.0;
    for (int k = 0; k < 3; ++k) {
        size = std::fmax(size, bmax[k] - bmin[k]);
        reach = std::fmax(reach, std::fmax(std::fabs(bmin[k]), std::fabs(bmax[k])));
    }
    if (!std::isfinite(size) || !std::isfinite(reach) || !std::isfinite(far_term)) return false;
    out.kind = T_CULL;
    out.r0 = 1e-6 * (size + reach) + 2.0 * far_term * reach * reach;  // |o - c|^2 <= 2 |o|^2 + 2 reach^2
    out.r1 = 2.0 * far_term;
    if (!std::isfinite(out.r0) || !std::isfinite(out.r1)) return false;
    for (int k = 0; k < 3; ++k) {
        out.p[k] = bmin[k];
        out.p[3 + k] = bmax[k];
    }
    fill_box_f32(out, bmin, bmax, out.r0, out.r1);
    return true;
}

// A node's box in its parent's space, for T_CULL records.
struct NodeBox {
    double mn[3], mx[3];
    // A ray origin at distance D is at most D / smin from a leaf in its object space (smin, smax:
    // smallest / largest stretch of the object-to-here map), where the quadratic's discriminant
    // carries an error of about 10 eps (D / smin)^2; a line that misses the unit shape by less
    // than half of that can still be reported as a hit, which is up to smax times as far in this
    // space.  far = 1e-9 smax / smin^2 (the largest over the leaves below a group) leaves six
    // orders of magnitude over eps.  (The triangle test has no such cancellation; the same padding
    // is simply generous for it.)
    double far;
    // Largest stretch from the object space of a cube at or below this node to the parent's space (0
    // without cubes): a cube's reported hit can lie outside the cube by RT_EPS |t| per axis in object
    // space (cube.rs:47-69 treats small direction components as parallel), this much further here.
    double drift;
    bool ok;  // false: unbounded, or a transform too degenerate to bound
    bool cubes_aligned;  // every cube at or below this node keeps its axes parallel to the parent's axes
};

// Boxes of all nodes, and which nodes are or contain a cone.
void node_boxes(const rt_scene_desc* d, std::vector<NodeBox>& box, std::vector<char>& cone_below) {
    const int N = d->n_nodes;
    box.assign(N, NodeBox{});
    cone_below.assign(N, 0);
    std::vector<double> far_below(N, 0.0);  // per container: largest `far` of its children
    std::vector<double> drift_below(N, 0.0);  // ... and largest `drift`
    std::vector<char> aligned_below(N, 1);    // ... and whether all of them have cubes_aligned
    for (int i = N - 1; i >= 0; --i) {      // children come after their parent
        const rt_node& n = d->nodes[i];
        NodeBox& b = box[i];
        double lo[3] = {-1.0, -1.0, -1.0}, hi[3] = {1.0, 1.0, 1.0};
        double below = 1e-9;
        b.ok = true;
        if (n.type == RT_CONE) cone_below[i] = 1;
        switch (n.type) {
            case RT_SPHERE:
            case RT_CUBE:
                break;
            case RT_CYLINDER: {
                const double mn = d->ranges[n.data].minimum, mx = d->ranges[n.data].maximum;
                lo[1] = std::fmin(mn, mx);
                hi[1] = std::fmax(mn, mx);
                break;
            }
            case RT_TRIANGLE:
            case RT_SMOOTH_TRIANGLE: {
                const rt_tri& t = d->tris[n.data];
                for (int k = 0; k < 3; ++k) {
                    const double p = t.p1[k], q = t.p1[k] + t.e1[k], r = t.p1[k] + t.e2[k];  // p1, p2, p3 up to an ulp
                    lo[k] = std::fmin(p, std::fmin(q, r));
                    hi[k] = std::fmax(p, std::fmax(q, r));
                }
                break;
            }
            case RT_GROUP:
            case RT_CSG:
                std::memcpy(lo, d->aabbs[n.data].min, sizeof lo);
                std::memcpy(hi, d->aabbs[n.data].max, sizeof hi);
                below = std::fmax(far_below[i], 1e-9);
                break;
            default:
                // Planes are unbounded.  Cones are left out: for a ray parallel to one half of the
                // cone the reference reports the single wall intersection without the
                // minimum < y < maximum check (cone.rs:119-127), so a truncated cone can be hit
                // outside any box around it.
                b.ok = false;
        }
        for (int k = 0; k < 3; ++k)
            if (!std::isfinite(lo[k]) || !std::isfinite(hi[k]) || lo[k] > hi[k]) b.ok = false;
        double fwd[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
        bool invertible = true;  // wanted for `drift` even where the node has no box
        if (n.xform >= 0 && !affine_forward(d->xforms[n.xform].inv, fwd)) {
            b.ok = false;
            invertible = false;
        }
        // Bounds on the stretch of the object-to-here map that hold for any affine map (shear, scale
        // after rotation): its largest singular value is at most the Frobenius norm of its 3x3, its
        // smallest at least 1 / the Frobenius norm of the inverse's 3x3.  (Column norms are the
        // singular values only for a rotation times a scale.)
        double fro_fwd = 0.0, fro_inv = 0.0;
        for (int row = 0; row < 3; ++row)
            for (int k = 0; k < 3; ++k) {
                fro_fwd += fwd[row * 4 + k] * fwd[row * 4 + k];
                const double v = n.xform >= 0 ? d->xforms[n.xform].inv[row * 4 + k] : (row == k ? 1.0 : 0.0);
                fro_inv += v * v;
            }
        // an untransformed node stretches nothing (the Frobenius bounds would say sqrt(3) per level)
        const double smax = n.xform >= 0 ? std::sqrt(fro_fwd) : 1.0, smin = n.xform >= 0 ? 1.0 / std::sqrt(fro_inv) : 1.0;
        b.far = below * smax / (smin * smin);
        b.drift = (n.type == RT_CUBE ? 1.0 : drift_below[i]) * smax;
        if (!invertible || !(b.drift >= 0.0)) b.drift = INFINITY;  // also NaN from a degenerate transform
        bool diagonal = true;  // this node's own transform only scales and translates
        if (n.xform >= 0) {
            const double* m = d->xforms[n.xform].inv;
            diagonal = m[1] == 0.0 && m[2] == 0.0 && m[4] == 0.0 && m[6] == 0.0 && m[8] == 0.0 && m[9] == 0.0;
        }
        b.cubes_aligned = b.drift == 0.0 || (aligned_below[i] && diagonal);  // no cube below: nothing to align
        if (!(smin > 0.0) || !std::isfinite(b.far)) b.ok = false;
        if (b.ok) {
            for (int k = 0; k < 3; ++k) {
                b.mn[k] = INFINITY;
                b.mx[k] = -INFINITY;
            }
            for (int corner = 0; corner < 8; ++corner) {
                const double x = (corner & 1) ? hi[0] : lo[0], y = (corner & 2) ? hi[1] : lo[1],
                             z = (corner & 4) ? hi[2] : lo[2];
                for (int row = 0; row < 3; ++row) {
                    const double v = fwd[row * 4] * x + fwd[row * 4 + 1] * y + fwd[row * 4 + 2] * z + fwd[row * 4 + 3];
                    b.mn[row] = std::fmin(b.mn[row], v);
                    b.mx[row] = std::fmax(b.mx[row], v);
                }
            }
        }
        if (n.parent >= 0) {
            drift_below[n.parent] = std::fmax(drift_below[n.parent], b.drift);
            if (!b.cubes_aligned) aligned_below[n.parent] = 0;
            far_below[n.parent] = std::fmax(far_below[n.parent], std::isfinite(b.far) ? b.far : 0.0);
            if (cone_below[i]) cone_below[n.parent] = 1;
        }
    }
}

// Lays the node table out as traversal records (see TRec), depth first in node order.
//
// T_CULL records go in front of (a) a finite sphere, cube or cylinder leaf and (b) runs of at least
// CLUSTER_MIN consecutive siblings that all have finite boxes (leaves, groups, CSG shapes), and of
// the halves of such runs, recursively: a wavefront whose rays all miss a box skips that part of the
// run.  Long runs come from scenes built without groups and from Group::divide (group.rs:79-105),
// which leaves every child that straddles a split plane in the parent.  A run is cut where the two
// parts' boxes are cheapest to visit (surface area times count, the usual estimate of how many rays
// reach a box).  The order of the leaves, which breaks ties between equal times, is untouched.
// A box that is not clearly smaller than the one the ray has just passed (the enclosing group's or
// run's) is not worth a record and is left out.
struct RecordBuilder {
    static constexpr int CLUSTER_MIN = 2;
    // Tuned on the default scene, a scene of loose quadrics and three divided meshes (870 k
    // triangles the largest); the frame times move by a few percent over 1..4, 1..3, 1.5..5, 0.3..0.9.
    // (A T_CULL test became cheaper relative to the others when runs of them got a loop of their own
    // in the kernel: triangles at 2.5..5 instead of 1.5 took 3-5 % off the mesh frames.)
    double need = 2.0;        // expected saved visits a run's box must reach
    double cost_quad = 1.0;   // visit costs in units of one T_CULL test: a leaf's own T_CULL record,
    double cost_tri = 3.5;    // a triangle test,
    double cost_group = 3.5;  // a group's or CSG shape's exact bounds test
    double worth = 0.5;       // a leaf's box must have at most this share of the enclosing box's area
    double xform_worth = 0.8; // ... and a transformed group's or CSG shape's box in front of its T_XFORM record this share

    const rt_scene_desc* d;
    bool no_cull;
    bool reorder = true;       // siblings with boxes may be visited in any order outside CSG shapes
    // a sorted cut must cost at most this share of the best cut in the given order (measured: 0.9 and
    // 1.0 are alike, 0.5 gives up a third of the gain on the dragon mesh)
    static constexpr double keep_order = 0.9;
    std::vector<TRec> out;
    std::vector<int> rec;  // first record of each node (its T_XFORM record when it has one)
    std::vector<NodeBox> box;
    std::vector<char> cone_below;
    std::vector<double> suffix_area;

    static double area(const double mn[3], const double mx[3]) {
        const double x = mx[0] - mn[0], y = mx[1] - mn[1], z = mx[2] - mn[2];
        return x * y + y * z + z * x;
    }
    static bool own_payload(const rt_node& n) {
        return n.type == RT_GROUP || n.type == RT_CSG || n.type == RT_TRIANGLE || n.type == RT_SMOOTH_TRIANGLE;
    }
    static bool container(const rt_node& n) { return n.type == RT_GROUP || n.type == RT_CSG; }
    int ctx_record(const rt_node& n) const { return n.ctx >= 0 ? rec[n.ctx] : -1; }  // ancestors are already emitted
    // Time-based pruning assumes every hit below a box lies inside it; the cone's unchecked wall
    // intersection does not, so boxes over cones only test the ray's line, as the reference does.
    bool boxable(int i) const { return box[i].ok && !(container(d->nodes[i]) && cone_below[i]); }
    static void grow(double mn[3], double mx[3], const NodeBox& b) {
        for (int k = 0; k < 3; ++k) {
            mn[k] = std::fmin(mn[k], b.mn[k]);
            mx[k] = std::fmax(mx[k], b.mx[k]);
        }
    }

    // Outside CSG shapes the members of a part may be visited in any order (the scene walk breaks
    // ties between equal times by node index, not by arrival): the part is sorted along the axis
    // whose best cut is cheapest.
    int spatial_split(std::vector<int>& sib, int lo, int hi) {
        const int n = hi - lo;
        auto centre = [&](int i, int axis) { return box[i].mn[axis] + box[i].mx[axis]; };
        auto sort_by = [&](int axis) {
            std::stable_sort(sib.begin() + lo, sib.begin() + hi,
                             [&](int a, int b) { return centre(a, axis) < centre(b, axis); });
        };
        if (n > 4096) {  // very long: the widest axis only
            double cmn[3] = {INFINITY, INFINITY, INFINITY}, cmx[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (int k = lo; k < hi; ++k)
                for (int a = 0; a < 3; ++a) {
                    cmn[a] = std::fmin(cmn[a], centre(sib[k], a));
                    cmx[a] = std::fmax(cmx[a], centre(sib[k], a));
                }
            int axis = 0;
            for (int a = 1; a < 3; ++a)
                if (cmx[a] - cmn[a] > cmx[axis] - cmn[axis]) axis = a;
            sort_by(axis);
            return best_split(sib, lo, hi);
        }
        // the order the members came in is kept unless sorting along an axis gives a clearly cheaper cut
        const std::vector<int> given(sib.begin() + lo, sib.begin() + hi);
        int best_axis = -1;
        double best;
        const int given_cut = best_split(sib, lo, hi, &best);
        best *= keep_order;
        for (int axis = 0; axis < 3; ++axis) {
            sort_by(axis);
            double cost;
            best_split(sib, lo, hi, &cost);
            if (cost < best) {
                best = cost;
                best_axis = axis;
            }
        }
        if (best_axis < 0) {
            std::copy(given.begin(), given.end(), sib.begin() + lo);
            return given_cut;
        }
        std::copy(given.begin(), given.end(), sib.begin() + lo);
        sort_by(best_axis);
        return best_split(sib, lo, hi);
    }

    int best_split(const std::vector<int>& sib, int lo, int hi, double* cost_out = nullptr) {
        const int n = hi - lo;
        if (cost_out) *cost_out = 0.0;
        if (n <= 2 && !cost_out) return lo + n / 2;
        // Very long runs (an undivided mesh) are cut in their middle half, which bounds the work
        // at n log n; short ones may peel off a single member.
        const int edge = n > 1024 ? n / 4 : 1;
        suffix_area.assign(n + 1, 0.0);
        double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int k = n - 1; k >= 1; --k) {
            grow(mn, mx, box[sib[lo + k]]);
            suffix_area[k] = area(mn, mx);
        }
        double pmn[3] = {INFINITY, INFINITY, INFINITY}, pmx[3] = {-INFINITY, -INFINITY, -INFINITY};
        int best = lo + n / 2;
        double best_cost = INFINITY;
        for (int k = 1; k < n; ++k) {
            grow(pmn, pmx, box[sib[lo + k - 1]]);
            const double cost = area(pmn, pmx) * k + suffix_area[k] * (n - k);
            if (k >= edge && k <= n - edge && cost < best_cost) {
                best_cost = cost;
                best = lo + k;
            }
        }
        if (cost_out) *cost_out = best_cost;
        return best;
    }

    // Rough cost of visiting a member of a run, in units of one T_CULL test.
    double visit_cost(int i) const {
        const rt_node& n = d->nodes[i];
        if (container(n)) return cost_group;
        if (n.type == RT_TRIANGLE || n.type == RT_SMOOTH_TRIANGLE) return cost_tri;
        return cost_quad;  // the leaf's own T_CULL record
    }

    // `enclosing`: area of the box the ray has already been tested against on the way here.
    // Written with an explicit stack: the cheapest cut of a long run can be very uneven, and the
    // depth of a recursion would then grow with the length of the run.
    void emit_run(std::vector<int>& sib, int lo0, int hi0, double enclosing0, bool spatial, bool in_csg) {
        struct Item {
            int lo, hi;        // a part of the run, or (lo < 0) "set the skip link of record hi"
            double enclosing;
        };
        std::vector<Item> todo{{lo0, hi0, enclosing0}};
        while (!todo.empty()) {
            const Item it = todo.back();
            todo.pop_back();
            if (it.lo < 0) {
                out[it.hi].skip = (int)out.size();
                continue;
            }
            if (it.hi - it.lo == 1) {
                emit_subtree(sib[it.lo], it.enclosing, in_csg);
                continue;
            }
            double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
            double far = 0.0, cost = 0.0;
            for (int k = it.lo; k < it.hi; ++k) {
                grow(mn, mx, box[sib[k]]);
                far = std::fmax(far, box[sib[k]].far);
                cost += visit_cost(sib[k]);
            }
            double enclosing = it.enclosing;
            const double a = area(mn, mx);
            const double share = std::isfinite(enclosing) ? (enclosing > 0.0 ? a / enclosing : 1.0) : 0.0;
            TRec c{};
            // worth a record when the visits it is expected to save outweigh its own test
            if (cost * (1.0 - share) >= need && fill_cull(mn, mx, far, c)) {
                c.ctx = ctx_record(d->nodes[sib[it.lo]]);  // siblings share it
                c.node = sib[it.lo];
                todo.push_back({-1, (int)out.size(), 0.0});
                out.push_back(c);
                enclosing = a;
            }
            const int mid = spatial ? spatial_split(sib, it.lo, it.hi) : best_split(sib, it.lo, it.hi);
            todo.push_back({mid, it.hi, enclosing});
            todo.push_back({it.lo, mid, enclosing});
        }
    }

    // in_csg: between a CSG shape's begin and end records the leaves' intersections are collected in
    // arrival order, which the stable sort at the end record relies on: no reordering there.
    void emit_children(int first, int end, double enclosing, bool in_csg) {
        std::vector<int> sib;
        for (int c = first; c < end; c = d->nodes[c].skip) sib.push_back(c);
        const int n = (int)sib.size();
        if (reorder && !in_csg && !no_cull) {
            // every child with a box goes into one set, whatever lies between them
            std::vector<int> set;
            for (int c : sib) {
                if (boxable(c))
                    set.push_back(c);
                else
                    emit_subtree(c, enclosing, in_csg);
            }
            if (!set.empty()) emit_run(set, 0, (int)set.size(), enclosing, true, false);
            return;
        }
        for (int a = 0; a < n;) {
            int b = a;
            if (!no_cull)
                while (b < n && boxable(sib[b])) ++b;
            if (b - a >= CLUSTER_MIN) {
                emit_run(sib, a, b, enclosing, false, in_csg);
                a = b;
            } else {
                emit_subtree(sib[a], enclosing, in_csg);
                ++a;
            }
        }
    }

    void emit_subtree(int i, double enclosing, bool in_csg) {
        const rt_node& n = d->nodes[i];
        TRec r{};
        r.ctx = ctx_record(n);
        r.node = i;
        int flags = (n.flags & RT_NODE_CLOSED) ? T_CLOSED : 0;
        rec[i] = (int)out.size();
        int extra = -1;  // this node's T_XFORM or T_CULL record
        // untransformed levels between this node and the space the walk keeps the ray in (see through_identity)
        int levels = 0;
        for (int a = n.parent; a >= 0 && a != n.ctx && levels < 2; a = d->nodes[a].parent) ++levels;
        int front = -1;  // a box in the parent's space in front of this node's T_XFORM record
        if (n.xform >= 0 && container(n) && !no_cull && box[i].ok && area(box[i].mn, box[i].mx) <= xform_worth * enclosing) {
            // Taking the ray into a transformed group's space and back costs several box tests (33 f64
            // operations and the single-precision set-up, twice): rays that miss the group's box in the
            // parent's space skip that.  With a cone below, the box only tests the ray's line.
            TRec c = r;
            if (fill_cull(box[i].mn, box[i].mx, box[i].far, c)) {
                if (cone_below[i]) c.kind |= T_LINE_ONLY;
                front = (int)out.size();
                out.push_back(c);
                rec[i] = (int)out.size();  // what the nodes below refer to as their space is the T_XFORM record
            }
        }
        if (n.xform >= 0 && own_payload(n)) {
            TRec x = r;
            x.kind = T_XFORM | T_HAS_INV | (levels << T_LEVELS_SHIFT);
            levels = -1;  // the node itself is then tested in its own space
            std::memcpy(x.p, d->xforms[n.xform].inv, sizeof x.p);
            extra = (int)out.size();
            out.push_back(x);
            flags |= T_OWN_XFORM;
            r.ctx = rec[i];  // the node itself is tested in its own space
        } else if (!own_payload(n)) {
            TRec c = r;
            const NodeBox& b = box[i];
            if (!no_cull && b.ok && area(b.mn, b.mx) <= worth * enclosing && fill_cull(b.mn, b.mx, b.far, c)) {
                extra = (int)out.size();
                out.push_back(c);
            }
            if (n.xform >= 0) {
                flags |= T_HAS_INV;
                std::memcpy(r.p, d->xforms[n.xform].inv, sizeof r.p);
                const double* m = d->xforms[n.xform].inv;
                // not inside CSG shapes: their intersection lists are sorted by a total order that
                // tells -0 from +0
                static const bool diag_enabled = !(getenv("RT_DIAG") && atoi(getenv("RT_DIAG")) == 0);  // diagnostic switch
                if (diag_enabled && !in_csg && m[1] == 0.0 && m[2] == 0.0 && m[4] == 0.0 && m[6] == 0.0 && m[8] == 0.0 && m[9] == 0.0)
                    flags |= T_DIAG;
            }
        }
        if (n.xform < 0) ++levels;  // its own identity matrix
        r.kind = (n.type == RT_CSG ? (int)T_CSG_BEGIN : n.type) | flags | ((levels < 0 ? 0 : (levels > 2 ? 2 : levels)) << T_LEVELS_SHIFT);
        double inner = INFINITY;
        if (container(n)) {
            const double* mn = d->aabbs[n.data].min;
            const double* mx = d->aabbs[n.data].max;
            std::memcpy(r.p, mn, 3 * sizeof(double));
            std::memcpy(r.p + 3, mx, 3 * sizeof(double));
            fill_box_f32(r, mn, mx, 0.0, 0.0);
            if (cone_below[i]) {
                r.kind |= T_LINE_ONLY;
            } else {
                inner = area(mn, mx);
                if (!std::isfinite(inner)) inner = INFINITY;  // NaN from inf - inf
            }
        } else if (n.type == RT_CYLINDER || n.type == RT_CONE) {
            r.r0 = d->ranges[n.data].minimum;
            r.r1 = d->ranges[n.data].maximum;
        } else if (n.type == RT_TRIANGLE || n.type == RT_SMOOTH_TRIANGLE) {
            std::memcpy(r.p, d->tris[n.data].p1, 3 * sizeof(double));
            std::memcpy(r.p + 3, d->tris[n.data].e1, 3 * sizeof(double));
            std::memcpy(r.p + 6, d->tris[n.data].e2, 3 * sizeof(double));
        }
        const int at = (int)out.size();
        out.push_back(r);
        if (container(n)) {
            emit_children(i + 1, n.skip, inner, in_csg || n.type == RT_CSG);
            if (n.type == RT_CSG) {
                // The end record follows the last descendant; a skip over a node inside this shape
                // lands before it, a skip over the shape itself after it.
                TRec e{};
                e.kind = T_CSG_END | (((n.flags & RT_NODE_CSG_OP_MASK) >> 4) << T_CSG_OP_SHIFT);
                e.node = i;
                e.skip = d->nodes[i + 1].skip;  // NODE index where the right operand starts
                e.ctx = n.xform >= 0 ? rec[i] : ctx_record(n);
                out.push_back(e);
            }
        }
        const int after = (int)out.size();
        out[at].skip = after;
        if (extra >= 0) out[extra].skip = after;
        if (front >= 0) out[front].skip = after;
    }
};

// What the device needs to know about the scene's cubes and boxes (see the note on cubes at slabs_f32).
struct BoxFacts {
    double drift_max = 0.0;     // the largest NodeBox::drift
    bool cubes_aligned = true;  // NodeBox::cubes_aligned of every node
    double extent = 0.0;        // the largest finite |plane| of any box record
};
std::vector<TRec> build_records(const rt_scene_desc* d, BoxFacts* facts = nullptr) {
    static const bool no_cull = getenv("RT_NO_CULL") && atoi(getenv("RT_NO_CULL")) != 0;
    RecordBuilder b{};
    b.d = d;
    b.no_cull = no_cull;
    b.rec.assign(d->n_nodes + 1, -1);
    node_boxes(d, b.box, b.cone_below);
    b.reorder = !(getenv("RT_REORDER") && atoi(getenv("RT_REORDER")) == 0);  // RT_REORDER=0: diagnostic
    if (const char* tune = getenv("RT_TUNE"))  // diagnostic: "need,cost_quad,cost_tri,cost_group,worth"
        sscanf(tune, "%lf,%lf,%lf,%lf,%lf", &b.need, &b.cost_quad, &b.cost_tri, &b.cost_group, &b.worth);
    b.emit_children(0, d->n_nodes, INFINITY, false);
    if (facts) {
        for (const NodeBox& nb : b.box) {
            facts->drift_max = std::fmax(facts->drift_max, nb.drift);
            if (!nb.cubes_aligned) facts->cubes_aligned = false;
        }
        for (const TRec& r : b.out) {
            const int type = r.kind & 0xff;
            if (type != T_CULL && type != RT_GROUP && type != T_CSG_BEGIN) continue;
            for (int k = 0; k < 6; ++k)
                if (std::isfinite(r.p[k])) facts->extent = std::fmax(facts->extent, std::fabs(r.p[k]));
        }
    }
    return std::move(b.out);
}

// Gathers per-leaf shading data (see SRec).
std::vector<SRec> build_shade_records(const rt_scene_desc* d) {
    std::vector<SRec> out(d->n_nodes);
    for (int i = 0; i < d->n_nodes; ++i) {
        const rt_node& n = d->nodes[i];
        SRec r{};
        r.type = n.type;
        r.flags = (n.flags & RT_NODE_CASTS_SHADOW) ? S_CASTS_SHADOW : 0;
        r.pattern = -1;
        r.tri = (n.type == RT_TRIANGLE || n.type == RT_SMOOTH_TRIANGLE) ? n.data : -1;
        for (int k = 0; k < RT_MAX_XFORM_NEST; ++k) {
            r.anc_xf[k] = -1;
            r.anc_level[k] = 0;
        }
        int levels = 1, found = 0;
        for (int p = n.parent; p >= 0; p = d->nodes[p].parent) {
            if (d->nodes[p].xform >= 0 && found < RT_MAX_XFORM_NEST) {
                r.anc_xf[found] = d->nodes[p].xform;
                r.anc_level[found] = levels;
                ++found;
            }
            ++levels;
        }
        r.n_levels = levels;
        r.n_anc = found;
        if (n.xform >= 0) {
            r.flags |= S_HAS_XFORM;
            std::memcpy(r.inv, d->xforms[n.xform].inv, sizeof r.inv);
            std::memcpy(r.inv_t, d->xforms[n.xform].inv_t, sizeof r.inv_t);
        }
        if (n.type != RT_GROUP && n.type != RT_CSG) {
            const rt_material& m = d->materials[n.material];
            r.ka = m.ambient;
            r.kd = m.diffuse;
            r.ks = m.specular;
            r.shin = m.shininess;
            r.refl = m.reflective;
            r.transp = m.transparency;
            r.ri = m.refractive_index;
            r.pattern = m.pattern;
            const rt_pattern& p = d->patterns[m.pattern];
            if (p.type == RT_PAT_SOLID) {
                r.flags |= S_SOLID;
                std::memcpy(r.colour, p.color, sizeof r.colour);
            }
        }
        if (n.type == RT_CYLINDER || n.type == RT_CONE) {
            r.r0 = d->ranges[n.data].minimum;
            r.r1 = d->ranges[n.data].maximum;
        }
        out[i] = r;
    }
    return out;
}

int check_params(const rt_params* p) {
    if (!p) return fail(RT_ERR_INVALID, "null params");
    if (p->world_size < 1 || p->rank < 0 || p->rank >= p->world_size)
        return fail(RT_ERR_INVALID, "rank/world_size out of range");
    if (p->max_depth < 0 || p->max_depth > RT_MAX_DEPTH)
        return fail(RT_ERR_LIMIT, "max_depth outside 0..RT_MAX_DEPTH");
    return RT_OK;
}

int local_tile_rows(int height, const rt_params* p) {
    int tile_rows = (height + RT_TILE - 1) / RT_TILE;
    if (p->rank >= tile_rows) return 0;
    return (tile_rows - p->rank + p->world_size - 1) / p->world_size;
}

// Frames with at least this many tiles are rendered expensive tiles first (see tile_class_kernel).
constexpr long long ORDER_MIN_TILES = 4096;

// One band of a frame that is rendered in several launches (render_host: band k's rows cross PCIe while
// band k + 1 renders).  The bands of a frame share one counter set, which the caller clears before the
// first band; every band has its own slice of the order arrays and its own tile counts per class.
struct BandCtx {
    long long order_offset;  // ints into the scene's order arrays
    int slot;                // which three words after the counter shards hold this band's tiles per class
    long long frame_tiles;   // tiles of the whole frame: what decides strips and the classifier's depth, not the band's own size
};
constexpr int MAX_BANDS = 16;  // COUNTER_SET_WORDS leaves 64 words after the shards: four per band
static_assert(COUNTER_SET_WORDS - COUNTER_WORDS >= 4 * MAX_BANDS, "tile counts per class of every band");

// fixed: null, or this device's static scratch (rt_render_device_static): the launch then uses the
// counter set and the order arrays in it and allocates nothing.
int launch(rt_scene* s, DevJob job, unsigned blocks, double* d_out, hipStream_t stream, unsigned char* fixed = nullptr,
           const BandCtx* band = nullptr) {
    // blocks: one per tile (or per 64 caller-supplied rays); strips of class-2 tiles are added below
    int slot = (int)(s->launches % TIMING_RING);
    // Launches on one scene must be ordered (one stream at a time): each clears the other counter set.
    const int a = s->next_set, b = a ^ 1;
    unsigned long long* mine = s->d_counters + (size_t)a * COUNTER_SET_WORDS;
    unsigned long long* other = s->d_counters + (size_t)b * COUNTER_SET_WORDS;
    int* order = s->d_order;
    const long long tiles = job.n_rays > 0 ? 0 : (long long)job.tiles_x * job.local_tile_rows;
    unsigned long long* class_count = mine + COUNTER_WORDS;
    if (band) {
        other = nullptr;  // (the caller cleared `mine`; nothing is cleared by the kernels of a banded frame)
        if (order) order += band->order_offset;
        class_count += 4 * band->slot;
    } else if (fixed) {
        // The array keeps the last call's contents: the counters (ray counts, tiles per class) must
        // start at zero; the order arrays are rewritten by the pre-pass before they are read.
        mine = reinterpret_cast<unsigned long long*>(fixed);
        other = nullptr;
        order = reinterpret_cast<int*>(fixed + STATIC_COUNTER_BYTES);
        class_count = mine + COUNTER_WORDS;
        HIP_TRY(hipMemsetAsync(mine, 0, COUNTER_SET_WORDS * sizeof(unsigned long long), stream));
    } else if (!s->set_clean[a]) {
        HIP_TRY(hipMemsetAsync(mine, 0, COUNTER_SET_WORDS * sizeof(unsigned long long), stream));
    }
    const bool clears_other = other && blocks >= (COUNTER_SET_WORDS + RT_BLOCK - 1) / RT_BLOCK;  // enough threads to reach every word
    static const bool order_enabled = !(getenv("RT_TILE_ORDER") && atoi(getenv("RT_TILE_ORDER")) == 0);
    job.tile_order = nullptr;
    static const bool box_order_enabled = !(getenv("RT_BOX_ORDER") && atoi(getenv("RT_BOX_ORDER")) == 0);  // diagnostic switch
    const bool by_boxes = !s->order_useful && box_order_enabled && s->n_top_boxes > 0 && job.n_rays == 0;
    if (order_enabled && (s->order_useful || by_boxes) && tiles >= ORDER_MIN_TILES) {
        if (band && !order) return fail(RT_ERR_INVALID, "banded launch without order arrays");
        if (!fixed && !band && tiles > s->order_cap) {
            if (s->d_order) HIP_TRY(hipFree(s->d_order));
            s->d_order = nullptr;
            s->order_cap = 0;
            HIP_TRY(hipMalloc((void**)&s->d_order, 4 * (size_t)tiles * sizeof(int)));  // three order arrays and the classes
            s->order_cap = tiles;
            order = s->d_order;
        }
        job.tile_order = order;
        job.tile_class_count = class_count;
    }
    // Tiles whose centre ray meets a transparent surface (class 2) are rendered in strips, one
    // wavefront each, so that most lanes of those wavefronts start without a pixel and take over
    // refraction subtrees (see L_*): the frame's longest wavefront gets shorter.  More strips mean
    // more lanes to share with and more wavefronts in total: four for small frames, whose time is
    // that of their longest wavefront, two for large ones.  RT_SPLIT=1, 2 or 4 overrides (diagnostic).
    job.split = 1;
    job.extra_blocks = 0;
    bool share = false;
    const bool grow = s->dev.cube_kappa != 0.0f;
    static const bool csg_share = !(getenv("RT_CSG_SHARE") && atoi(getenv("RT_CSG_SHARE")) == 0);  // diagnostic switch
    if (job.tile_order && s->any_transparent && (!s->any_csg || csg_share)) {
        static const int split_env = getenv("RT_SPLIT") ? atoi(getenv("RT_SPLIT")) : 0;
        // measured on the default scene: 8,100 tiles 0.42 -> 0.25 ms with four strips, 32,400 tiles
        // 0.47 -> 0.35 ms with two, 64,800 tiles (an eighth of 7680x4320) 0.47 -> 0.49 ms, and at 129,600
        // tiles and beyond strips and sharing only add work (+13 % at 7680x4320).  Running the strips as
        // a launch of their own beside a plain launch for the other tiles was tried and is worse at
        // every size: the other tiles lose the sharing, which is most of the gain on small frames.
        const long long size = band ? band->frame_tiles : tiles;
        int split = split_env ? split_env : (size >= 49152 ? 1 : (size >= 16384 ? 2 : 4));
        if (split != 1 && split != 2 && split != 4 && split != 8) split = 1;
        if (split > 1) {
            const long long cap = tiles / 8 > 1024 ? tiles / 8 : 1024;  // beyond that many extra strips: whole tiles
            job.split = split;
            job.extra_blocks = tiles * (split - 1) < cap ? tiles * (split - 1) : cap;
            if ((long long)blocks + job.extra_blocks > 0x7fffffffLL) job.split = 1, job.extra_blocks = 0;
            blocks += (unsigned)job.extra_blocks;
            share = job.split > 1;
        }
    }
    if (!band) HIP_TRY(hipEventRecord(s->ev_start[slot], stream));
    if (job.tile_order && by_boxes) {
        int* classes = order + 3 * tiles;
        hipLaunchKernelGGL(tile_box_kernel, dim3((unsigned)((tiles + 63) / 64)), dim3(64), 0, stream, job, tiles,
                           (const TopBoxes*)s->d_top_boxes, s->n_top_boxes, classes);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(tile_order_kernel, dim3((unsigned)((tiles + ORDER_BLOCK - 1) / ORDER_BLOCK)), dim3(ORDER_BLOCK), 0, stream, job.tiles_x,
                           job.local_tile_rows, tiles, (const int*)classes, order, class_count);
        HIP_TRY(hipGetLastError());
    } else if (job.tile_order) {
        DevJob pre = job;
        pre.tile_order = nullptr;
        using Pre = void (*)(DevScene, DevJob, long long, int, int*);
        static const Pre pres[4] = {tile_class_kernel<false, false, false>, tile_class_kernel<false, true, false>,
                                    tile_class_kernel<true, false, false>, tile_class_kernel<true, true, false>};
        Pre prek = s->any_csg ? tile_class_kernel<true, true, true, true> : pres[(s->any_xform ? 2 : 0) | (s->any_cull ? 1 : 0)];
        if (grow && !s->any_csg) prek = s->any_xform ? tile_class_kernel<true, true, false, true> : tile_class_kernel<false, true, false, true>;
        int* classes = order + 3 * tiles;
        // How many mirror bounces the classifier follows.  Strips need it (a tile that shows the glass in a
        // mirror is as deep as the glass itself); without strips it still sorts those tiles to the front,
        // which is worth more than the longer pre-pass up to about 10^5 tiles: measured with the plain
        // kernel, default scene, 0 / 2 bounces: 64,800 tiles (an eighth of 7680x4320) 0.448 / 0.423 ms,
        // 90,000 tiles 0.610 / 0.580, 129,600 tiles 0.776 / 0.796, 518,400 tiles 2.886 / 2.903.
        // (The work-sharing kernel without strips gains nothing at those sizes: 0.454, 0.626, 0.789, 2.910.)
        static const int probe_env = getenv("RT_PROBE") ? atoi(getenv("RT_PROBE")) : -1;  // diagnostic
        // Beyond that size one bounce is still followed when a PLANE is a mirror: its tiles are a large part of
        // the frame, and the ones that show a transparent object are as deep as the object itself; rendered in
        // row-major order among the other reflective tiles they were what the frame ended on (csg scene,
        // 3840x2160: 1.86 ms without the bounce, 1.34 with one, 1.36 with two; a frame every 1.31 ms with two frames
        // in flight, i.e. without any tail).  The glass scene, whose floor is a mirror too, pays 1 % for it.
        const int probe = probe_env >= 0 ? probe_env
                          : ((share || (s->any_transparent && (band ? band->frame_tiles : tiles) < 106496)) ? 2
                             : ((s->any_transparent && s->mirror_plane) ? 1 : 0));
        hipLaunchKernelGGL(prek, dim3((unsigned)((tiles + 63) / 64)), dim3(64), 0, stream, s->dev, pre, tiles, probe, classes);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(tile_order_kernel, dim3((unsigned)((tiles + ORDER_BLOCK - 1) / ORDER_BLOCK)), dim3(ORDER_BLOCK), 0, stream, job.tiles_x,
                           job.local_tile_rows, tiles, (const int*)classes, order, class_count);
        HIP_TRY(hipGetLastError());
    }
    if (blocks > 0) {
        using Kernel = void (*)(DevScene, DevJob, double*, unsigned long long*, unsigned long long*);
        static const Kernel kernels[8] = {
            rt_kernel<false, false, false, false>, rt_kernel<false, false, true, false>,
            rt_kernel<false, true, false, false>,  rt_kernel<false, true, true, false>,
            rt_kernel<true, false, false, false>,  rt_kernel<true, false, true, false>,
            rt_kernel<true, true, false, false>,   rt_kernel<true, true, true, false>};
        Kernel kernel = kernels[(s->any_transparent ? 4 : 0) | (s->any_xform ? 2 : 0) | (s->any_cull ? 1 : 0)];
        static const Kernel sharing[4] = {rt_kernel<true, false, false, false, true>, rt_kernel<true, false, true, false, true>,
                                          rt_kernel<true, true, false, false, true>, rt_kernel<true, true, true, false, true>};
        if (share) kernel = sharing[(s->any_xform ? 2 : 0) | (s->any_cull ? 1 : 0)];
        // scenes with CSG shapes are rare: they share the two most general variants
        if (grow) {  // boxes grown per ray (always with T_CULL support: cubes have boxes)
            static const Kernel growing[4] = {rt_kernel<false, false, true, false, false, true>, rt_kernel<false, true, true, false, false, true>,
                                              rt_kernel<true, false, true, false, false, true>, rt_kernel<true, true, true, false, false, true>};
            kernel = growing[(s->any_transparent ? 2 : 0) | (s->any_xform ? 1 : 0)];
            if (share) kernel = s->any_xform ? rt_kernel<true, true, true, false, true, true> : rt_kernel<true, false, true, false, true, true>;
        }
        if (s->any_csg) {
            kernel = s->any_transparent ? rt_kernel<true, true, true, true, false, true> : rt_kernel<false, true, true, true, false, true>;
            if (share) kernel = rt_kernel<true, true, true, true, true, true>;
        }
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(RT_BLOCK), 0, stream, s->dev, job, d_out, mine,
                           clears_other ? other : nullptr);
        HIP_TRY(hipGetLastError());
    }
    if (band) return RT_OK;  // the caller does the frame's bookkeeping
    HIP_TRY(hipEventRecord(s->ev_stop[slot], stream));
    if (!fixed) {
        s->set_clean[a] = blocks == 0;
        if (clears_other) s->set_clean[b] = true;
        s->next_set = b;
    }
    s->last_counters = mine;
    s->launches++;
    s->last_stream = stream;
    return RT_OK;
}

int fill_stats(rt_scene* s, rt_stats* st) {
    if (!st) return RT_OK;
    std::vector<unsigned long long> shards(COUNTER_SHARDS * 16);
    HIP_TRY(hipMemcpy(shards.data(), s->last_counters ? s->last_counters : s->d_counters,
                      shards.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long c[16] = {};
    for (unsigned k = 0; k < COUNTER_SHARDS; ++k)
        for (int j = 0; j < 16; ++j) c[j] += shards[k * 16 + j];
    st->rays_primary = c[0];
    st->rays_shadow = c[1];
    st->rays_reflect = c[2];
    st->rays_refract = c[3];
    st->kernel_ms = 0.0f;
    st->reserved = 0.0f;
#ifdef RT_PROFILE_MIX
    {
        unsigned long long cls[3] = {0, 0, 0};
        (void)hipMemcpy(cls, (s->last_counters ? s->last_counters : s->d_counters) + COUNTER_WORDS, sizeof cls, hipMemcpyDeviceToHost);
        fprintf(stderr, "RT_PROFILE_MIX tiles per class: %llu %llu %llu\n", cls[0], cls[1], cls[2]);
        if (s->d_order && s->order_cap > 0) {
            std::vector<int> raw((size_t)s->order_cap);
            (void)hipMemcpy(raw.data(), s->d_order + 3 * s->order_cap, raw.size() * sizeof(int), hipMemcpyDeviceToHost);
            long long hist[4] = {0, 0, 0, 0};
            for (int v : raw) ++hist[v < 0 || v > 2 ? 3 : v];
            fprintf(stderr, "RT_PROFILE_MIX centre-ray classes: %lld %lld %lld (other %lld)\n", hist[0], hist[1], hist[2], hist[3]);
        }
        unsigned long long most = 0;
        for (unsigned k = 0; k < COUNTER_SHARDS; ++k) most = shards[k * 16 + 15] > most ? shards[k * 16 + 15] : most;
        fprintf(stderr, "RT_PROFILE_MIX cull=%llu cull_lanes=%llu group=%llu group_lanes=%llu leaf=%llu leaf_lanes=%llu leaf_idle=%llu walks=%llu rounds=%llu handoffs=%llu lane_rounds=%llu longest_wave_rounds=%llu\n",
                c[4], c[5], c[6], c[7], c[8], c[9], c[10], c[11], c[12], c[13], c[14], most);
    }
#endif
#ifdef RT_PROFILE
    {
        const unsigned long long* pc = c;
        fprintf(stderr, "RT_PROFILE waves=%llu wave_cycles=%llu closest_cycles=%llu shadow_cycles=%llu visits=%llu traces=%llu rounds=%llu prepare_cycles=%llu lighting_cycles=%llu return_cycles=%llu csg_end_cycles=%llu leaf_cycles=%llu\n",
                pc[10], pc[4], pc[5], pc[6], pc[7], pc[8], pc[9], pc[11], pc[12], pc[13], pc[14], pc[15]);
    }
#endif
    if (s->launches > 0) {
        int slot = (int)((s->launches - 1) % TIMING_RING);
        HIP_TRY(hipEventSynchronize(s->ev_stop[slot]));
        HIP_TRY(hipEventElapsedTime(&st->kernel_ms, s->ev_start[slot], s->ev_stop[slot]));
    }
    return RT_OK;
}

}  // namespace

extern "C" {

const char* rt_last_error(void) { return g_error.c_str(); }
// For the other source files of this library (not part of the ABI).
void rt_internal_set_error(const char* message) { g_error = message ? message : ""; }

int rt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int rt_scene_create(const rt_scene_desc* d, int device, rt_scene** out) {
    if (!out) return fail(RT_ERR_INVALID, "null out pointer");
    *out = nullptr;
    bool all_cast = true, any_transparent = false;
    const double t_start = rtdl::now_ms();
    int rc = validate(d, all_cast, any_transparent);
    if (rc != RT_OK) return rc;
    const double t_validated = rtdl::now_ms();
    if (getenv("RT_DUMP_RECORDS")) {  // diagnostic: the traversal records as text, before any device is needed
        BoxFacts facts;
        const std::vector<TRec> recs = build_records(d, &facts);
        fprintf(stderr, "largest drift %g, cubes aligned %d, extent %g\n", facts.drift_max, (int)facts.cubes_aligned, facts.extent);
        for (size_t k = 0; k < recs.size(); ++k) {
            const TRec& r = recs[k];
            fprintf(stderr, "rec %3zu type %3d flags %#6x node %3d skip %3d ctx %3d  [%g %g %g | %g %g %g] r0 %g r1 %g\n", k, r.kind & 0xff,
                    r.kind & ~0xff, r.node, r.skip, r.ctx, r.p[0], r.p[1], r.p[2], r.p[3], r.p[4], r.p[5], r.r0, r.r1);
        }
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(RT_ERR_NO_DEVICE, "no HIP device is visible");
    if (device < 0 || device >= ndev) return fail(RT_ERR_INVALID, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device));
    unsigned char* static_scratch = nullptr;
    if ((rc = static_scratch_for(device, &static_scratch)) != RT_OK) return rc;
    rt_scene* s = new rt_scene();
    s->device = device;
    s->static_scratch = static_scratch;
    s->any_transparent = any_transparent;
    auto cleanup = [&](int code) {
        rt_scene_destroy(s);
        return code;
    };
    double records_ms = 0.0;
    {
        const double t0 = rtdl::now_ms();
        std::vector<SRec> srecs = build_shade_records(d);
        records_ms += rtdl::now_ms() - t0;
        if ((rc = upload(s, srecs.data(), srecs.size(), &s->dev.shade)) != RT_OK) return cleanup(rc);
    }
    {
        const double t0 = rtdl::now_ms();
        BoxFacts facts;
        std::vector<TRec> recs = build_records(d, &facts);
        records_ms += rtdl::now_ms() - t0;
        // see the note on cubes at slabs_f32
        s->dev.extent_f32 = round_up_f32(facts.extent);
        s->dev.cube_kappa = 0.0f;
        s->dev.ym_limit = INFINITY;
        s->dev.ymin_limit = INFINITY;
        if (!std::isfinite(facts.drift_max)) {
            s->dev.ym_limit = 0.0f;  // a degenerate transform above a cube: nothing is decided in single precision
        } else if (facts.drift_max > 0.0 && facts.cubes_aligned) {
            s->dev.ym_limit = (float)(0.99 / (RT_EPS * facts.drift_max));
        } else if (facts.drift_max > 0.0) {
            const double kappa = 2.0 * 1.4143e-5 * facts.drift_max * 1.001;
            s->dev.cube_kappa = round_up_f32(kappa);
            s->dev.ymin_limit = (float)(0.49 / kappa);
        }
        // the kernel addresses records by a 32-bit byte offset
        if (recs.size() >= (size_t(1) << 25)) {
            return cleanup(fail(RT_ERR_LIMIT, "more than 2^25 traversal records"));
        }
        s->build.n_records = (int32_t)recs.size();
        for (const TRec& r : recs)
            if ((r.kind & 0xff) == T_CULL) ++s->build.n_cull_records;
        if ((rc = upload(s, recs.data(), recs.size(), &s->dev.trav)) != RT_OK) return cleanup(rc);
        s->dev.n_trav = (int)recs.size();
        s->build.walk_bytes = recs.size() * sizeof(TRec);
        bool cls[3] = {false, false, false};
        for (int i = 0; i < d->n_nodes; ++i)
            if (d->nodes[i].type != RT_GROUP && d->nodes[i].type != RT_CSG) {
                const rt_material& m = d->materials[d->nodes[i].material];
                cls[m.transparency != 0.0 ? 2 : (m.reflective != 0.0 ? 1 : 0)] = true;
            }
        s->order_useful = recs.size() <= 1024 && (int)cls[0] + (int)cls[1] + (int)cls[2] > 1;
        for (int i = 0; i < d->n_nodes; ++i)
            if (d->nodes[i].type == RT_PLANE && d->materials[d->nodes[i].material].reflective != 0.0) s->mirror_plane = true;
        if (!s->order_useful && recs.size() > 1024) {
            // the world's finite objects (nodes without a parent), in world space, a little padded
            std::vector<NodeBox> nb;
            std::vector<char> cone;
            node_boxes(d, nb, cone);
            TopBoxes top{};
            int n_top = 0;
            bool fits = true;
            for (int i = 0; i < d->n_nodes && fits; i = d->nodes[i].skip) {
                if (!nb[i].ok) continue;  // planes and the like: everywhere
                if (n_top == MAX_TOP_BOXES) {
                    fits = false;
                    break;
                }
                for (int k = 0; k < 3; ++k) {
                    const double pad = 0.01 * (nb[i].mx[k] - nb[i].mn[k]) + 1e-9;
                    top.b[n_top][k] = nb[i].mn[k] - pad;
                    top.b[n_top][3 + k] = nb[i].mx[k] + pad;
                }
                ++n_top;
            }
            if (fits && n_top > 0) {
                void* p = nullptr;
                if (hipMalloc(&p, sizeof top) != hipSuccess || hipMemcpy(p, &top, sizeof top, hipMemcpyHostToDevice) != hipSuccess) {
                    if (p) (void)hipFree(p);
                    return cleanup(fail(RT_ERR_HIP, "upload of the world's object boxes failed"));
                }
                s->d_top_boxes = (TopBoxes*)p;
                s->n_top_boxes = n_top;
            }
        }
        for (const TRec& r : recs)
            if ((r.kind & 0xff) == T_XFORM) s->any_xform = true;
            else if ((r.kind & 0xff) == T_CULL) s->any_cull = true;
            else if ((r.kind & 0xff) == T_CSG_BEGIN) s->any_csg = true;
    }
    if ((rc = upload(s, d->xforms, d->n_xforms, &s->dev.xforms)) != RT_OK) return cleanup(rc);
    if ((rc = upload(s, d->tris, d->n_tris, &s->dev.tris)) != RT_OK) return cleanup(rc);
    if ((rc = upload(s, d->tri_normals, d->n_tris, &s->dev.trin)) != RT_OK) return cleanup(rc);
    if ((rc = upload(s, d->patterns, d->n_patterns, &s->dev.pats)) != RT_OK) return cleanup(rc);
    if ((rc = upload(s, d->pattern_xforms, (size_t)d->n_pattern_xforms * 12, &s->dev.patx)) != RT_OK)
        return cleanup(rc);
    if ((rc = upload(s, d->lights, d->n_lights, &s->dev.lights)) != RT_OK) return cleanup(rc);
    s->dev.n_lights = d->n_lights;
    s->dev.all_cast_shadow = all_cast ? 1 : 0;
    bool finite_bounds = true;
    double extent = 0.0;
    for (int i = 0; i < d->n_aabbs; ++i)
        for (int k = 0; k < 3; ++k)
        {
            if (!std::isfinite(d->aabbs[i].min[k]) || !std::isfinite(d->aabbs[i].max[k])) finite_bounds = false;
            extent = std::fmax(extent, std::fmax(std::fabs(d->aabbs[i].min[k]), std::fabs(d->aabbs[i].max[k])));
        }
    s->dev.prune_f32 = finite_bounds ? round_up_f32(2.02e-9 * (1.0 + extent)) : INFINITY;
    void* p = nullptr;
    if (hipMalloc(&p, 2 * COUNTER_SET_WORDS * sizeof(unsigned long long)) != hipSuccess)
        return cleanup(fail(RT_ERR_HIP, "hipMalloc(counters) failed"));
    s->d_counters = (unsigned long long*)p;
    if (hipMemset(p, 0, 2 * COUNTER_SET_WORDS * sizeof(unsigned long long)) != hipSuccess)
        return cleanup(fail(RT_ERR_HIP, "hipMemset(counters) failed"));
    for (int i = 0; i < TIMING_RING; ++i) {
        if (hipEventCreate(&s->ev_start[i]) != hipSuccess || hipEventCreate(&s->ev_stop[i]) != hipSuccess)
            return cleanup(fail(RT_ERR_HIP, "hipEventCreate failed"));
    }
    if (hipDeviceSynchronize() != hipSuccess) return cleanup(fail(RT_ERR_HIP, "hipDeviceSynchronize after the uploads failed"));
    s->build.validate_ms = (float)(t_validated - t_start);
    s->build.records_ms = (float)records_ms;
    s->build.upload_ms = (float)(rtdl::now_ms() - t_validated - records_ms);  // allocations, copies, events
    s->build.n_nodes = d->n_nodes;
    *out = s;
    return RT_OK;
}

int rt_scene_build_info(const rt_scene* s, rt_build_info* out) {
    if (!s || !out) return fail(RT_ERR_INVALID, "bad arguments");
    *out = s->build;
    return RT_OK;
}

void rt_scene_destroy(rt_scene* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    (void)hipDeviceSynchronize();
    for (void* p : s->allocations) (void)hipFree(p);
    if (s->d_counters) (void)hipFree(s->d_counters);
    if (s->d_order) (void)hipFree(s->d_order);
    if (s->d_top_boxes) (void)hipFree(s->d_top_boxes);
    if (s->d_frame) (void)hipFree(s->d_frame);
    if (s->d_rays) (void)hipFree(s->d_rays);
    if (s->dl_ready) s->dl.destroy();
    if (s->frame_done) (void)hipEventDestroy(s->frame_done);
    for (hipStream_t st : s->band_stream)
        if (st) (void)hipStreamDestroy(st);
    for (hipEvent_t e : s->band_done)
        if (e) (void)hipEventDestroy(e);
    if (s->band_clear) (void)hipEventDestroy(s->band_clear);
    for (int i = 0; i < TIMING_RING; ++i) {
        if (s->ev_start[i]) (void)hipEventDestroy(s->ev_start[i]);
        if (s->ev_stop[i]) (void)hipEventDestroy(s->ev_stop[i]);
    }
    delete s;
}

int64_t rt_local_rows(int32_t height, const rt_params* p) {
    if (check_params(p) != RT_OK || height < 0) return -1;
    int ltr = local_tile_rows(height, p);
    if (ltr == 0) return 0;
    // every local tile is full except possibly the one that holds the frame's last rows
    int64_t rows = (int64_t)ltr * RT_TILE;
    int last_global_tile = (ltr - 1) * p->world_size + p->rank;
    int64_t end = (int64_t)(last_global_tile + 1) * RT_TILE;
    if (end > height) rows -= end - height;
    return rows;
}

int64_t rt_global_row(int64_t local_row, const rt_params* p) {
    if (check_params(p) != RT_OK || local_row < 0) return -1;
    int64_t lt = local_row / RT_TILE;
    return (lt * p->world_size + p->rank) * RT_TILE + local_row % RT_TILE;
}

static int render_device(rt_scene* s, const rt_camera* cam, const rt_params* prm, void* d_rgb, bool out_u8,
                         void* hip_stream, bool use_static = false) {
    if (!s || !cam) return fail(RT_ERR_INVALID, "null scene or camera");
    int rc = check_params(prm);
    if (rc != RT_OK) return rc;
    if (cam->width <= 0 || cam->height <= 0) return fail(RT_ERR_INVALID, "empty canvas");
    HIP_TRY(hipSetDevice(s->device));
    DevJob job{};
    job.width = cam->width;
    job.height = cam->height;
    job.half_width = cam->half_width;
    job.half_height = cam->half_height;
    job.pixel_size = cam->pixel_size;
    std::memcpy(job.inv, cam->inv, sizeof job.inv);
    job.rank = prm->rank;
    job.world_size = prm->world_size;
    job.tiles_x = (cam->width + RT_TILE - 1) / RT_TILE;
    job.local_tile_rows = local_tile_rows(cam->height, prm);
    job.max_depth = prm->max_depth;
    job.n_rays = 0;
    job.out_u8 = out_u8 ? 1 : 0;
    if (out_u8)
        job.wide_stores = ((reinterpret_cast<uintptr_t>(d_rgb) & 3) == 0 && (cam->width & 3) == 0) ? 1 : 0;
    else
        job.wide_stores = ((reinterpret_cast<uintptr_t>(d_rgb) & 15) == 0 && (cam->width & 1) == 0) ? 1 : 0;
    long long tiles = (long long)job.tiles_x * job.local_tile_rows;
    if (tiles > 0 && !d_rgb) return fail(RT_ERR_INVALID, "null output buffer");
    long long blocks = (tiles + RT_BLOCK_WAVES - 1) / RT_BLOCK_WAVES;
    if (blocks > 0x7fffffffLL) return fail(RT_ERR_LIMIT, "frame too large for one launch");
    if (use_static) {
        if (tiles > RT_STATIC_MAX_TILES)
            return fail(RT_ERR_LIMIT, "rt_render_device_static: this rank's part of the frame has more than "
                                      "RT_STATIC_MAX_TILES 8x8 tiles");
        if (!s->static_scratch) return fail(RT_ERR_HIP, "the static scratch array has no address on this device");
    }
    return launch(s, job, (unsigned)blocks, (double*)d_rgb, (hipStream_t)hip_stream,
                  use_static ? s->static_scratch : nullptr);
}

int rt_render_device(rt_scene* s, const rt_camera* cam, const rt_params* prm, double* d_rgb, void* hip_stream) {
    return render_device(s, cam, prm, d_rgb, false, hip_stream);
}

int rt_render_device_u8(rt_scene* s, const rt_camera* cam, const rt_params* prm, uint8_t* d_rgb8, void* hip_stream) {
    return render_device(s, cam, prm, d_rgb8, true, hip_stream);
}

int rt_render_device_static(rt_scene* s, const rt_camera* cam, const rt_params* prm, double* d_rgb, void* hip_stream) {
    return render_device(s, cam, prm, d_rgb, false, hip_stream, true);
}

// Grows the scene's staging frame to `bytes` (device memory).
static int reserve_frame(rt_scene* s, size_t bytes) {
    if (bytes <= s->frame_bytes) return RT_OK;
    if (s->d_frame) (void)hipFree(s->d_frame);
    s->d_frame = nullptr;
    s->frame_bytes = 0;
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, bytes));
    s->d_frame = (double*)p;
    s->frame_bytes = bytes;
    return RT_OK;
}
// The copy stream and the "frame complete" event of the synchronous entry points, made on first use.
static int ready_downloader(rt_scene* s) {
    if (!s->dl_ready) {
        HIP_TRY(s->dl.init(s->device));
        s->dl_ready = true;
    }
    if (!s->frame_done) HIP_TRY(hipEventCreateWithFlags(&s->frame_done, hipEventDisableTiming));
    return RT_OK;
}

// A frame on its way to HOST memory is rendered in bands of tile rows, on two streams in turn, and
// each band's rows start across PCIe as soon as the band is done, so the copy of band k overlaps the
// render of band k + 1.  Whichever of the two takes longer, the other is hidden except for a small
// band at one end: the first band is small so that the copy engine starts early when the copy is the
// longer part (an f64 frame: 3.5 ms against 0.7 ms of rendering at 3840x2160), the last band is small so
// that little is left to copy when the render is (clamped bytes at 7680x4320: 1.8 ms against 2.6 ms).
constexpr size_t BAND_MIN_BYTES = (size_t)8 << 20;
constexpr int BAND_MIN_TILE_ROWS = 64;
static const int BAND_SIXTEENTHS[5] = {1, 3, 8, 3, 1};
static_assert(5 <= MAX_BANDS, "one slot of tile counts per band");

static int render_host_banded(rt_scene* s, const rt_camera* cam, const rt_params* prm, char* out, bool out_u8,
                              size_t row_bytes, size_t bytes) {
    if (cam->width <= 0 || cam->height <= 0) return fail(RT_ERR_INVALID, "empty canvas");
    for (hipStream_t& st : s->band_stream)
        if (!st) HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int k = 0; k < 5; ++k)
        if (!s->band_done[k]) HIP_TRY(hipEventCreateWithFlags(&s->band_done[k], hipEventDisableTiming));
    if (!s->band_clear) HIP_TRY(hipEventCreateWithFlags(&s->band_clear, hipEventDisableTiming));
    DevJob job{};
    job.width = cam->width;
    job.height = cam->height;
    job.half_width = cam->half_width;
    job.half_height = cam->half_height;
    job.pixel_size = cam->pixel_size;
    std::memcpy(job.inv, cam->inv, sizeof job.inv);
    job.rank = prm->rank;
    job.world_size = prm->world_size;
    job.tiles_x = (cam->width + RT_TILE - 1) / RT_TILE;
    job.max_depth = prm->max_depth;
    job.out_u8 = out_u8 ? 1 : 0;
    const int ltr = local_tile_rows(cam->height, prm);
    const long long frame_tiles = (long long)job.tiles_x * ltr;
    if (frame_tiles > 0x7fffffffLL) return fail(RT_ERR_LIMIT, "frame too large for one launch");
    // the order arrays for the whole frame, before anything is in flight (an allocation synchronises the device)
    if ((s->order_useful || s->n_top_boxes > 0) && frame_tiles > s->order_cap) {
        if (s->d_order) HIP_TRY(hipFree(s->d_order));
        s->d_order = nullptr;
        s->order_cap = 0;
        HIP_TRY(hipMalloc((void**)&s->d_order, 4 * (size_t)frame_tiles * sizeof(int)));
        s->order_cap = frame_tiles;
    }
    const int slot = (int)(s->launches % TIMING_RING);
    const int set = s->next_set;
    unsigned long long* mine = s->d_counters + (size_t)set * COUNTER_SET_WORDS;
    hipStream_t first_stream = s->band_stream[0];
    HIP_TRY(hipEventRecord(s->ev_start[slot], first_stream));
    HIP_TRY(hipMemsetAsync(mine, 0, COUNTER_SET_WORDS * sizeof(unsigned long long), first_stream));
    HIP_TRY(hipEventRecord(s->band_clear, first_stream));
    HIP_TRY(hipStreamWaitEvent(s->band_stream[1], s->band_clear, 0));
    const bool pinned = rtdl::is_pinned(out);
    int row0[5], rows_of[5];
    int at = 0;
    for (int k = 0; k < 5; ++k) {
        int n = k == 4 ? ltr - at : (int)((long long)ltr * BAND_SIXTEENTHS[k] / 16);
        if (n < 1) n = 1;
        if (n > ltr - at) n = ltr - at;
        row0[k] = at;
        rows_of[k] = n;
        at += n;
    }
    long long order_offset = 0;
    size_t off[5], len[5];
    for (int k = 0; k < 5; ++k) {
        hipStream_t st = s->band_stream[k & 1];
        off[k] = (size_t)row0[k] * RT_TILE * row_bytes;
        len[k] = off[k] >= bytes ? 0 : std::min((size_t)rows_of[k] * RT_TILE * row_bytes, bytes - off[k]);  // the last tile row may be short
        if (rows_of[k] > 0) {
            job.tile_row0 = row0[k];
            job.local_tile_rows = rows_of[k];
            char* d_band = reinterpret_cast<char*>(s->d_frame) + off[k];
            if (out_u8)
                job.wide_stores = ((reinterpret_cast<uintptr_t>(d_band) & 3) == 0 && (cam->width & 3) == 0) ? 1 : 0;
            else
                job.wide_stores = ((reinterpret_cast<uintptr_t>(d_band) & 15) == 0 && (cam->width & 1) == 0) ? 1 : 0;
            const long long tiles = (long long)job.tiles_x * rows_of[k];
            BandCtx ctx{order_offset, k, frame_tiles};
            int rc = launch(s, job, (unsigned)((tiles + RT_BLOCK_WAVES - 1) / RT_BLOCK_WAVES), reinterpret_cast<double*>(d_band), st, nullptr, &ctx);
            if (rc != RT_OK) {
                (void)hipDeviceSynchronize();
                return rc;
            }
            order_offset += 4 * tiles;
        }
        HIP_TRY(hipEventRecord(s->band_done[k], st));
        if (pinned && len[k]) HIP_TRY(s->dl.enqueue(reinterpret_cast<const char*>(s->d_frame) + off[k], len[k], out + off[k], s->band_done[k]));
    }
    // the frame's device time: from the first band's start to the end of the later of the last two bands
    hipStream_t last_stream = s->band_stream[4 & 1];
    HIP_TRY(hipStreamWaitEvent(last_stream, s->band_done[3], 0));
    HIP_TRY(hipEventRecord(s->ev_stop[slot], last_stream));
    s->set_clean[set] = false;
    s->last_counters = mine;
    s->launches++;
    s->last_stream = last_stream;
    if (pinned) {
        HIP_TRY(s->dl.finish());
    } else {
        for (int k = 0; k < 5; ++k)
            if (len[k])
                HIP_TRY(s->dl.download(reinterpret_cast<const char*>(s->d_frame) + off[k], len[k], 0, 0, 1, out + off[k], s->band_done[k]));
    }
    HIP_TRY(hipStreamSynchronize(last_stream));
    return RT_OK;
}

static int render_host(rt_scene* s, const rt_camera* cam, const rt_params* prm, void* out, bool out_u8,
                       rt_stats* stats) {
    if (!s || !cam) return fail(RT_ERR_INVALID, "null scene or camera");
    int rc = check_params(prm);
    if (rc != RT_OK) return rc;
    if (cam->width <= 0 || cam->height <= 0) return fail(RT_ERR_INVALID, "empty canvas");
    const int64_t rows = rt_local_rows(cam->height, prm);
    if (rows < 0) return fail(RT_ERR_INVALID, "bad canvas height");
    const size_t pixel = out_u8 ? 3 : 3 * sizeof(double);
    const size_t row_bytes = (size_t)cam->width * pixel;
    if (rows > 0 && row_bytes > SIZE_MAX / (size_t)rows) return fail(RT_ERR_LIMIT, "frame size overflows");
    const size_t bytes = (size_t)rows * row_bytes;
    if (bytes && !out) return fail(RT_ERR_INVALID, "null output buffer");
    HIP_TRY(hipSetDevice(s->device));
    if ((rc = reserve_frame(s, bytes)) != RT_OK) return rc;
    if ((rc = ready_downloader(s)) != RT_OK) return rc;
    static const bool bands_enabled = !(getenv("RT_BANDS") && atoi(getenv("RT_BANDS")) == 0);  // diagnostic switch
    if (bands_enabled && bytes >= BAND_MIN_BYTES && local_tile_rows(cam->height, prm) >= BAND_MIN_TILE_ROWS) {
        rc = render_host_banded(s, cam, prm, static_cast<char*>(out), out_u8, row_bytes, bytes);
        return rc != RT_OK ? rc : fill_stats(s, stats);
    }
    rc = render_device(s, cam, prm, s->d_frame, out_u8, nullptr);
    if (rc != RT_OK) return rc;
    HIP_TRY(hipEventRecord(s->frame_done, nullptr));
    // The frame crosses PCIe on the copy stream: straight into page-locked memory, or through the
    // page-locked ring when the caller's buffer is ordinary memory (rt_download.h).
    HIP_TRY(s->dl.download(reinterpret_cast<const char*>(s->d_frame), bytes, 0, 0, 1, static_cast<char*>(out),
                           s->frame_done));
    return fill_stats(s, stats);
}

int rt_render_u8(rt_scene* s, const rt_camera* cam, const rt_params* prm, uint8_t* rgb8_out, rt_stats* stats) {
    return render_host(s, cam, prm, rgb8_out, true, stats);
}

int rt_render(rt_scene* s, const rt_camera* cam, const rt_params* prm, double* rgb_out, rt_stats* stats) {
    return render_host(s, cam, prm, rgb_out, false, stats);
}

int rt_color_at(rt_scene* s, int64_t n, const double* origins, const double* directions, int32_t remaining,
                double* rgb_out, rt_stats* stats) {
    if (!s) return fail(RT_ERR_INVALID, "null scene");
    if (n < 0) return fail(RT_ERR_INVALID, "negative ray count");
    if (remaining < 0 || remaining > RT_MAX_DEPTH) return fail(RT_ERR_LIMIT, "remaining outside 0..RT_MAX_DEPTH");
    if (n == 0) return RT_OK;
    if (!origins || !directions || !rgb_out) return fail(RT_ERR_INVALID, "null ray or output buffer");
    HIP_TRY(hipSetDevice(s->device));
    size_t bytes = (size_t)n * 3 * sizeof(double);
    if (3 * bytes > s->rays_bytes) {
        if (s->d_rays) HIP_TRY(hipFree(s->d_rays));
        s->d_rays = nullptr;
        s->rays_bytes = 0;
        void* p = nullptr;
        HIP_TRY(hipMalloc(&p, 3 * bytes));
        s->d_rays = (double*)p;
        s->rays_bytes = 3 * bytes;
    }
    double* d_o = s->d_rays;
    double* d_d = s->d_rays + (size_t)n * 3;
    double* d_c = s->d_rays + (size_t)n * 6;
    HIP_TRY(hipMemcpy(d_o, origins, bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_d, directions, bytes, hipMemcpyHostToDevice));
    DevJob job{};
    job.max_depth = remaining;
    job.n_rays = n;
    job.ray_o = d_o;
    job.ray_d = d_d;
    job.world_size = 1;
    long long blocks = (n + RT_BLOCK - 1) / RT_BLOCK;
    if (blocks > 0x7fffffffLL) return fail(RT_ERR_LIMIT, "too many rays for one launch");
    int rc = launch(s, job, (unsigned)blocks, d_c, nullptr);
    if (rc != RT_OK) return rc;
    HIP_TRY(hipStreamSynchronize(nullptr));
    HIP_TRY(hipMemcpy(rgb_out, d_c, bytes, hipMemcpyDeviceToHost));
    return fill_stats(s, stats);
}

int rt_selftest_division(int device, int64_t n, const double* a, const double* d, double* out_fast,
                         double* out_ieee) {
    if (n < 0 || (n && (!a || !d || !out_fast || !out_ieee))) return fail(RT_ERR_INVALID, "bad arguments");
    if (n == 0) return RT_OK;
    HIP_TRY(hipSetDevice(device));
    size_t bytes = (size_t)n * sizeof(double);
    double* buf = nullptr;
    HIP_TRY(hipMalloc((void**)&buf, 4 * bytes));
    int rc = RT_OK;
    if (hipMemcpy(buf, a, bytes, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(buf + n, d, bytes, hipMemcpyHostToDevice) != hipSuccess) {
        rc = fail(RT_ERR_HIP, "hipMemcpy to device failed");
    } else {
        hipLaunchKernelGGL(division_selftest_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr,
                           (long long)n, buf, buf + n, buf + 2 * n, buf + 3 * n);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
            hipMemcpy(out_fast, buf + 2 * n, bytes, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(out_ieee, buf + 3 * n, bytes, hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(RT_ERR_HIP, "division self-test kernel failed");
    }
    (void)hipFree(buf);
    return rc;
}

// Formats device bytes as PPM text into device memory; *length gets the text size.  d_work holds
// height + 1 counters.  Synchronises the stream (the caller needs the length).
static int ppm_encode_device(const uint8_t* d_rgb8, int width, int height, char* d_text, unsigned long long capacity,
                             unsigned long long* d_work, unsigned long long* length, hipStream_t stream) {
    char header[64];
    const int hn = std::snprintf(header, sizeof header, "P3\n%d %d\n255\n", width, height);  // canvas.rs:66-72
    PpmHeader head;
    std::memset(&head, 0, sizeof head);
    if (hn <= 0 || hn > (int)sizeof head.text) return fail(RT_ERR_INVALID, "PPM header does not fit");
    std::memcpy(head.text, header, (size_t)hn);
    head.n = hn;
    const long long n_values = (long long)width * 3;
    // one wavefront per row; the writer puts the header in front (nothing is written when the text does not fit)
    hipLaunchKernelGGL(ppm_measure_kernel, dim3((unsigned)height), dim3(64), 0, stream, d_rgb8, n_values, height, d_work);
    hipLaunchKernelGGL(ppm_scan_kernel, dim3(1), dim3(64), 0, stream, height, (unsigned long long)hn, d_work);
    hipLaunchKernelGGL(ppm_write_kernel, dim3((unsigned)height), dim3(64), 0, stream, d_rgb8, n_values, height, d_work, capacity, head, d_text);
    HIP_TRY(hipGetLastError());
    unsigned long long total = 0;
    HIP_TRY(hipMemcpyAsync(&total, d_work + height, sizeof total, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    *length = total;
    if (total > capacity) return fail(RT_ERR_LIMIT, "PPM text does not fit the buffer; *length holds the size needed");
    return RT_OK;
}

int rt_ppm_encode(int device, const uint8_t* rgb8, int32_t width, int32_t height, char* text_out, uint64_t capacity,
                  uint64_t* length) {
    if (width <= 0 || height <= 0 || !rgb8 || !length || (capacity && !text_out))
        return fail(RT_ERR_INVALID, "bad arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(RT_ERR_NO_DEVICE, "no HIP device is visible");
    if (device < 0 || device >= ndev) return fail(RT_ERR_INVALID, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device));
    const size_t in_bytes = (size_t)width * height * 3;
    uint8_t* d_in = nullptr;
    char* d_text = nullptr;
    unsigned long long* d_work = nullptr;
    int rc = RT_OK;
    if (hipMalloc((void**)&d_in, in_bytes) != hipSuccess || hipMalloc((void**)&d_text, capacity ? capacity : 1) != hipSuccess ||
        hipMalloc((void**)&d_work, ((size_t)height + 1) * sizeof(unsigned long long)) != hipSuccess ||
        hipMemcpy(d_in, rgb8, in_bytes, hipMemcpyHostToDevice) != hipSuccess) {
        rc = fail(RT_ERR_HIP, "device allocation or copy failed");
    } else {
        unsigned long long total = 0;
        rc = ppm_encode_device(d_in, width, height, d_text, capacity, d_work, &total, nullptr);
        *length = total;
        if (rc == RT_OK && hipMemcpy(text_out, d_text, (size_t)total, hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(RT_ERR_HIP, "hipMemcpy of the PPM text failed");
    }
    if (d_in) (void)hipFree(d_in);
    if (d_text) (void)hipFree(d_text);
    if (d_work) (void)hipFree(d_work);
    return rc;
}

int rt_render_ppm(rt_scene* s, const rt_camera* cam, const rt_params* prm, char* text_out, uint64_t capacity,
                  uint64_t* length, rt_stats* stats) {
    if (!s || !cam || !length || (capacity && !text_out)) return fail(RT_ERR_INVALID, "bad arguments");
    int rc = check_params(prm);
    if (rc != RT_OK) return rc;
    if (prm->world_size != 1) return fail(RT_ERR_INVALID, "rt_render_ppm renders whole frames: world_size must be 1");
    if (cam->width <= 0 || cam->height <= 0) return fail(RT_ERR_INVALID, "empty canvas");
    HIP_TRY(hipSetDevice(s->device));
    // staging: bytes of the frame, row counters, text
    const size_t frame = (size_t)cam->width * cam->height * 3;
    const size_t work = ((size_t)cam->height + 1) * sizeof(unsigned long long);
    const size_t need = ((frame + 255) & ~(size_t)255) + ((work + 255) & ~(size_t)255) + (capacity ? capacity : 1);
    if ((rc = reserve_frame(s, need)) != RT_OK) return rc;
    if ((rc = ready_downloader(s)) != RT_OK) return rc;
    uint8_t* d_rgb8 = reinterpret_cast<uint8_t*>(s->d_frame);
    unsigned long long* d_work = reinterpret_cast<unsigned long long*>(d_rgb8 + ((frame + 255) & ~(size_t)255));
    char* d_text = reinterpret_cast<char*>(d_work) + ((work + 255) & ~(size_t)255);
    rc = render_device(s, cam, prm, d_rgb8, true, nullptr);
    if (rc != RT_OK) return rc;
    unsigned long long total = 0;
    rc = ppm_encode_device(d_rgb8, cam->width, cam->height, d_text, capacity, d_work, &total, nullptr);
    *length = total;
    if (rc != RT_OK) return rc;
    HIP_TRY(hipEventRecord(s->frame_done, nullptr));
    HIP_TRY(s->dl.download(d_text, (size_t)total, 0, 0, 1, text_out, s->frame_done));
    return fill_stats(s, stats);
}

void* rt_host_alloc(uint64_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? (size_t)bytes : 1, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        g_error = "hipHostMalloc failed";
        return nullptr;
    }
    return p;
}

void rt_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

int rt_scene_launch_times(rt_scene* s, float* ms, int cap) {
    if (!s || !ms || cap < 0) return fail(RT_ERR_INVALID, "bad arguments");
    long long have = s->launches < TIMING_RING ? s->launches : TIMING_RING;
    if (have > cap) have = cap;
    for (long long k = 0; k < have; ++k) {
        long long idx = s->launches - have + k;
        int slot = (int)(idx % TIMING_RING);
        HIP_TRY(hipEventSynchronize(s->ev_stop[slot]));
        HIP_TRY(hipEventElapsedTime(&ms[k], s->ev_start[slot], s->ev_stop[slot]));
    }
    return (int)have;
}

int rt_scene_ray_counts(rt_scene* s, rt_stats* st) {
    if (!s || !st) return fail(RT_ERR_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->last_stream));
    return fill_stats(s, st);
}

}  // extern "C"
