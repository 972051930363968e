// CPU ORACLE -- TEST INFRASTRUCTURE ONLY.
//
// A C++17 restatement of the per-pixel render path of lydiasamuel/raytracer, written from the
// reference's Rust sources (cited per function as file:line below, relative to the reference
// checkout).  It exists so that tests, the smoke check and the benchmark's CPU-baseline leg can
// compare the HIP path against the reference's arithmetic.  Nothing in the shipped library may
// include, link or call this file.
//
// Pinning: this restatement is checked against the reference's own known-answer tests
// (transcribed in tests/test_oracle_kat_*.py) and against the reference's golden render
// example_output.png (tests/test_oracle_golden.py).
//
// Arithmetic contract: every f64 expression is evaluated in the reference's order, with no FMA
// contraction (build with -ffp-contract=off) and no fast-math; libm pow/cos/sqrt/floor/tan/sin are
// the host's glibc, which is what the Rust standard library calls on the same platform.
//
// Two cost modes share the same arithmetic and give bit-identical colours:
//   fast     : inverse matrices cached per object, reflect/refract traced once per hit
//   faithful : inverse recomputed by heap-backed cofactor expansion on every use and
//              reflect/refract re-traced once per light, like the reference does
//              (shape.rs:24, world.rs:180-181); used for the CPU-baseline timing only.

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <limits>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

namespace {

const double EPSILON = 0.00001;  // lib.rs:22
const double INF = std::numeric_limits<double>::infinity();

thread_local bool g_faithful = false;

struct Counters {
    uint64_t rays_primary = 0, rays_shadow = 0, rays_reflect = 0, rays_refract = 0;
    uint64_t aabb_test = 0, sphere_test = 0, plane_test = 0, cube_test = 0, cyl_test = 0,
             cone_test = 0, tri_test = 0, ray_xform = 0, shaded_hits = 0, light_evals = 0,
             pattern_nodes = 0, inverses = 0;
    // Algorithmic f64 operations of the deduplicated path (one per add/sub/mul/div/sqrt/min/max/
    // abs/compare/pow/cos on 3-component vectors and 3x4 affine matrices), accumulated with the
    // per-event constants written next to each `g_cnt.flops +=`; and the non-identity ray transforms.
    uint64_t flops = 0, xform_nonident = 0;
    uint64_t smooth_hits = 0;  // shaded hits on smooth triangles (their three vertex normals are read)
    void add(const Counters& o) {
        const uint64_t* s = reinterpret_cast<const uint64_t*>(&o);
        uint64_t* d = reinterpret_cast<uint64_t*>(this);
        for (size_t i = 0; i < sizeof(Counters) / sizeof(uint64_t); ++i) d[i] += s[i];
    }
};
thread_local Counters g_cnt;

// ---------------------------------------------------------------------------------------------
// Tuple (tuples/tuple.rs:24-181): four f64 components, w carried through every operation.
// ---------------------------------------------------------------------------------------------
struct Tuple {
    double x, y, z, w;
};
inline Tuple point(double x, double y, double z) { return {x, y, z, 1.0}; }
inline Tuple vec(double x, double y, double z) { return {x, y, z, 0.0}; }
inline Tuple operator+(Tuple a, Tuple b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline Tuple operator-(Tuple a, Tuple b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
inline Tuple operator-(Tuple a) { return {-a.x, -a.y, -a.z, -a.w}; }
inline Tuple operator*(Tuple a, double s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
inline Tuple operator/(Tuple a, double s) { return {a.x / s, a.y / s, a.z / s, a.w / s}; }
// tuple.rs:49-53
inline double magnitude(Tuple a) {
    return std::sqrt((a.x * a.x) + (a.y * a.y) + (a.z * a.z) + (a.w * a.w));
}
// tuple.rs:60-66
inline Tuple normalize(Tuple a) {
    double m = magnitude(a);
    return {a.x / m, a.y / m, a.z / m, a.w / m};
}
// tuple.rs:75-80
inline double dot(Tuple a, Tuple b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
// tuple.rs:83-92
inline Tuple cross(Tuple a, Tuple b) {
    return vec(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// tuple.rs:94-99
inline Tuple reflect(Tuple in, Tuple n) { return in - (n * 2.0 * dot(in, n)); }

// Color (tuples/color.rs:52-122)
struct Color {
    double r, g, b;
};
inline Color operator+(Color a, Color b) { return {a.r + b.r, a.g + b.g, a.b + b.b}; }
inline Color operator-(Color a, Color b) { return {a.r - b.r, a.g - b.g, a.b - b.b}; }
inline Color operator*(Color a, Color b) { return {a.r * b.r, a.g * b.g, a.b * b.b}; }
inline Color operator*(Color a, double s) { return {a.r * s, a.g * s, a.b * s}; }
inline Color operator/(Color a, double s) { return {a.r / s, a.g / s, a.b / s}; }
const Color BLACK{0.0, 0.0, 0.0};

// ---------------------------------------------------------------------------------------------
// Matrix (matrices/matrix.rs).  Fixed 4x4 storage; the arithmetic follows the dynamic original.
// ---------------------------------------------------------------------------------------------
struct M4 {
    double m[4][4];
};
M4 identity() {
    M4 r{};
    for (int i = 0; i < 4; ++i) r.m[i][i] = 1.0;
    return r;
}
M4 from_rows(const double* p) {
    M4 r;
    std::memcpy(r.m, p, sizeof(r.m));
    return r;
}
// matrix.rs:56-65
M4 translation(double x, double y, double z) {
    M4 r = identity();
    r.m[0][3] = x;
    r.m[1][3] = y;
    r.m[2][3] = z;
    return r;
}
// matrix.rs:80-89
M4 scaling(double x, double y, double z) {
    M4 r = identity();
    r.m[0][0] = x;
    r.m[1][1] = y;
    r.m[2][2] = z;
    return r;
}
// matrix.rs:91-122
M4 rotation_x(double a) {
    M4 r = identity();
    r.m[1][1] = std::cos(a);
    r.m[1][2] = -std::sin(a);
    r.m[2][1] = std::sin(a);
    r.m[2][2] = std::cos(a);
    return r;
}
M4 rotation_y(double a) {
    M4 r = identity();
    r.m[0][0] = std::cos(a);
    r.m[0][2] = std::sin(a);
    r.m[2][0] = -std::sin(a);
    r.m[2][2] = std::cos(a);
    return r;
}
M4 rotation_z(double a) {
    M4 r = identity();
    r.m[0][0] = std::cos(a);
    r.m[0][1] = -std::sin(a);
    r.m[1][0] = std::sin(a);
    r.m[1][1] = std::cos(a);
    return r;
}
// matrix.rs:128-137
M4 shearing(double xy, double xz, double yx, double yz, double zx, double zy) {
    M4 r = identity();
    r.m[0][1] = xy;
    r.m[0][2] = xz;
    r.m[1][0] = yx;
    r.m[1][2] = yz;
    r.m[2][0] = zx;
    r.m[2][1] = zy;
    return r;
}
// matrix.rs:367-401: sum starts at 0.0 and accumulates k = 0..3 in order.
M4 mul(const M4& a, const M4& b) {
    M4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double sum = 0.0;
            for (int k = 0; k < 4; ++k) sum = sum + (a.m[i][k] * b.m[k][j]);
            r.m[i][j] = sum;
        }
    return r;
}
// matrix.rs:344-363
inline Tuple mul(const M4& a, Tuple t) {
    double r[4];
    for (int i = 0; i < 4; ++i)
        r[i] = a.m[i][0] * t.x + a.m[i][1] * t.y + a.m[i][2] * t.z + a.m[i][3] * t.w;
    return {r[0], r[1], r[2], r[3]};
}
// matrix.rs:175-192
M4 transpose(const M4& a) {
    M4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) r.m[j][i] = a.m[i][j];
    return r;
}

// Determinant / cofactor chain, matrix.rs:233-325, on stack storage.
inline double det2(const double s[2][2]) { return s[0][0] * s[1][1] - s[0][1] * s[1][0]; }
double det3(const double s[3][3]) {
    double d = 0.0;
    for (int c = 0; c < 3; ++c) {
        double q[2][2];
        int qr = 0;
        for (int y = 0; y < 3; ++y) {
            if (y == 0) continue;
            int qc = 0;
            for (int x = 0; x < 3; ++x) {
                if (x == c) continue;
                q[qr][qc++] = s[y][x];
            }
            ++qr;
        }
        double minor = det2(q);
        double cof = ((0 + c) % 2 != 0) ? -minor : minor;  // matrix.rs:317-321
        d += s[0][c] * cof;                                  // matrix.rs:246
    }
    return d;
}
double cofactor4(const M4& a, int row, int col) {
    double s[3][3];
    int sr = 0;
    for (int y = 0; y < 4; ++y) {
        if (y == row) continue;
        int sc = 0;
        for (int x = 0; x < 4; ++x) {
            if (x == col) continue;
            s[sr][sc++] = a.m[y][x];
        }
        ++sr;
    }
    // matrix.rs:307-313: (-1.0).powf(row + col) * det; the power is exactly +1 or -1.
    double sign = ((row + col) % 2 == 0) ? 1.0 : -1.0;
    return sign * det3(s);
}
double det4(const M4& a) {
    double d = 0.0;
    for (int c = 0; c < 4; ++c) d += a.m[0][c] * cofactor4(a, 0, c);
    return d;
}
// matrix.rs:201-226.  Returns false when |det| < EPSILON (the reference returns Err and callers
// unwrap, i.e. panic).
bool inverse_fast(const M4& a, M4& out) {
    double d = det4(a);
    if ((std::fabs(d) - 0.0) < EPSILON) return false;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) out.m[c][r] = cofactor4(a, r, c) / d;
    return true;
}

// The same computation on heap-backed dynamic matrices, allocating a fresh matrix for every
// submatrix and calling pow for the sign, to reproduce the reference's cost structure
// (matrix.rs:253-325).  Results are bit-identical to inverse_fast.
struct DynM {
    int n;
    std::vector<double> v;
    DynM(int n_) : n(n_), v(size_t(n_) * n_, 0.0) {}
    double& at(int r, int c) { return v[size_t(r) * n + c]; }
    double at(int r, int c) const { return v[size_t(r) * n + c]; }
};
DynM dyn_sub(const DynM& a, int row, int col) {
    DynM r(a.n - 1);
    int ry = 0;
    for (int y = 0; y < a.n; ++y) {
        if (y == row) continue;
        int rx = 0;
        for (int x = 0; x < a.n; ++x) {
            if (x == col) continue;
            r.at(ry, rx++) = a.at(y, x);
        }
        ++ry;
    }
    return r;
}
double dyn_det(const DynM& a);
double dyn_cofactor(const DynM& a, int row, int col) {
    if (a.n == 4) {
        DynM s = dyn_sub(a, row, col);
        return std::pow(-1.0, double(row + col)) * dyn_det(s);
    }
    DynM s = dyn_sub(a, row, col);
    double minor = dyn_det(s);
    return ((row + col) % 2 != 0) ? -minor : minor;
}
double dyn_det(const DynM& a) {
    if (a.n == 2) return a.at(0, 0) * a.at(1, 1) - a.at(0, 1) * a.at(1, 0);
    double d = 0.0;
    for (int c = 0; c < a.n; ++c) d += a.at(0, c) * dyn_cofactor(a, 0, c);
    return d;
}
bool inverse_dyn(const M4& a, M4& out) {
    DynM m(4);
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) m.at(r, c) = a.m[r][c];
    double d = dyn_det(m);
    if ((std::fabs(d) - 0.0) < EPSILON) return false;
    DynM res(4);
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) res.at(c, r) = dyn_cofactor(m, r, c) / d;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) out.m[r][c] = res.at(r, c);
    return true;
}

// A transform together with its cached inverse.  inv() recomputes in faithful mode.
struct Xform {
    M4 fwd;
    M4 inv_cached;
    bool invertible;
    bool ident;
    explicit Xform(const M4& f) : fwd(f) {
        invertible = inverse_fast(f, inv_cached);
        ident = true;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
                if (f.m[i][j] != (i == j ? 1.0 : 0.0)) ident = false;
    }
    M4 inv() const {
        if (g_faithful) {
            ++g_cnt.inverses;
            M4 r;
            inverse_dyn(fwd, r);
            return r;
        }
        return inv_cached;
    }
};

// matrix.rs:145-173
M4 view_transform(Tuple from, Tuple to, Tuple up) {
    Tuple forward = normalize(to - from);
    Tuple upn = normalize(up);
    Tuple left = cross(forward, upn);
    Tuple true_up = cross(left, forward);
    M4 o = identity();
    o.m[0][0] = left.x;
    o.m[0][1] = left.y;
    o.m[0][2] = left.z;
    o.m[1][0] = true_up.x;
    o.m[1][1] = true_up.y;
    o.m[1][2] = true_up.z;
    o.m[2][0] = -forward.x;
    o.m[2][1] = -forward.y;
    o.m[2][2] = -forward.z;
    return mul(o, translation(-from.x, -from.y, -from.z));
}

// ---------------------------------------------------------------------------------------------
// Ray (tuples/ray.rs:24-35)
// ---------------------------------------------------------------------------------------------
struct Ray {
    Tuple o, d;
};
inline Tuple position(const Ray& r, double t) { return r.o + (r.d * t); }
inline Ray xform_ray(const Ray& r, const M4& m) { return {mul(m, r.o), mul(m, r.d)}; }

// ---------------------------------------------------------------------------------------------
// BoundingBox (tuples/bounding_box.rs)
// ---------------------------------------------------------------------------------------------
struct BBox {
    Tuple mn, mx;
};
BBox bbox_empty() { return {point(INF, INF, INF), point(-INF, -INF, -INF)}; }
// bounding_box.rs:25-39; f64::min/max return the non-NaN operand, as fmin/fmax do.
BBox bbox_add_point(BBox b, Tuple p) {
    b.mn = point(std::fmin(b.mn.x, p.x), std::fmin(b.mn.y, p.y), std::fmin(b.mn.z, p.z));
    b.mx = point(std::fmax(b.mx.x, p.x), std::fmax(b.mx.y, p.y), std::fmax(b.mx.z, p.z));
    return b;
}
BBox bbox_add(BBox a, BBox b) { return bbox_add_point(bbox_add_point(a, b.mn), b.mx); }  // :172-178
// bounding_box.rs:50-57
bool bbox_contains_box(const BBox& s, const BBox& o) {
    return o.mn.x >= s.mn.x && o.mx.x <= s.mx.x && o.mn.y >= s.mn.y && o.mx.y <= s.mx.y &&
           o.mn.z >= s.mn.z && o.mx.z <= s.mx.z;
}
bool bbox_contains_point(const BBox& s, Tuple p) {
    return p.x >= s.mn.x && p.x <= s.mx.x && p.y >= s.mn.y && p.y <= s.mx.y && p.z >= s.mn.z &&
           p.z <= s.mx.z;
}
// bounding_box.rs:59-78
BBox bbox_transform(const BBox& b, const M4& t) {
    Tuple p[8] = {b.mn,
                  point(b.mn.x, b.mn.y, b.mx.z),
                  point(b.mn.x, b.mx.y, b.mn.z),
                  point(b.mn.x, b.mx.y, b.mx.z),
                  point(b.mx.x, b.mn.y, b.mn.z),
                  point(b.mx.x, b.mn.y, b.mx.z),
                  point(b.mx.x, b.mx.y, b.mn.z),
                  b.mx};
    BBox r = bbox_empty();
    for (auto& q : p) r = bbox_add_point(r, mul(t, q));
    return r;
}
// bounding_box.rs:102-124 (also cube.rs:47-69 with min=-1, max=1)
inline void check_axis(double origin, double direction, double mn, double mx, double& tmin,
                       double& tmax) {
    double tmin_n = mn - origin;
    double tmax_n = mx - origin;
    double a, b;
    if (std::fabs(direction) >= EPSILON) {
        a = tmin_n / direction;
        b = tmax_n / direction;
    } else {
        a = tmin_n * INF;
        b = tmax_n * INF;
    }
    if (a > b) {
        tmin = b;
        tmax = a;
    } else {
        tmin = a;
        tmax = b;
    }
}
// bounding_box.rs:80-98
bool bbox_intersects(const BBox& b, const Ray& r) {
    ++g_cnt.aabb_test;
    g_cnt.flops += 26;
    double x0, x1, y0, y1, z0, z1;
    check_axis(r.o.x, r.d.x, b.mn.x, b.mx.x, x0, x1);
    check_axis(r.o.y, r.d.y, b.mn.y, b.mx.y, y0, y1);
    check_axis(r.o.z, r.d.z, b.mn.z, b.mx.z, z0, z1);
    double tmin = std::fmax(std::fmax(x0, y0), z0);
    double tmax = std::fmin(std::fmin(x1, y1), z1);
    return !(tmin > tmax);
}
// bounding_box.rs:126-161
void bbox_split(const BBox& b, BBox& left, BBox& right) {
    double dx = b.mx.x - b.mn.x, dy = b.mx.y - b.mn.y, dz = b.mx.z - b.mn.z;
    double greatest = std::fmax(dx, std::fmax(dy, dz));
    double x0 = b.mn.x, y0 = b.mn.y, z0 = b.mn.z, x1 = b.mx.x, y1 = b.mx.y, z1 = b.mx.z;
    if (std::fabs(greatest - dx) < EPSILON) {
        x0 += dx / 2.0;
        x1 = x0;
    } else if (std::fabs(greatest - dy) < EPSILON) {
        y0 += dy / 2.0;
        y1 = y0;
    } else {
        z0 += dz / 2.0;
        z1 = z0;
    }
    left = {b.mn, point(x1, y1, z1)};
    right = {point(x0, y0, z0), b.mx};
}

// ---------------------------------------------------------------------------------------------
// Patterns (patterns/*.rs)
// ---------------------------------------------------------------------------------------------
enum PatKind { P_SOLID = 0, P_STRIPED, P_GRADIENT, P_RING, P_CHECKER, P_BLENDED, P_PERTURBED, P_TEST };

// Ken Perlin's reference permutation (Improved Noise, 2002), doubled; perturbed.rs:8-34.
const int PERM_BASE[256] = {
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
inline int64_t perm(int64_t i) { return PERM_BASE[i & 255]; }  // indices stay below 512

// Rust `as i64`: saturating, NaN -> 0.
inline int64_t to_i64(double v) {
    if (v != v) return 0;
    if (v >= 9223372036854775807.0) return INT64_MAX;
    if (v <= -9223372036854775808.0) return INT64_MIN;
    return (int64_t)v;
}

struct Pattern {
    PatKind kind;
    Color color{1, 1, 1};
    std::shared_ptr<Pattern> a, b;
    double scale = 0.0;
    Xform xf{identity()};

    // perturbed.rs:59-82
    static double fade(double t) { return t * t * t * (t * (t * 6.0 - 15.0) + 10.0); }
    static double lerp(double t, double a, double b) { return a + t * (b - a); }
    static double grad(int64_t hash, double x, double y, double z) {
        int64_t h = hash & 15;
        double u = h < 8 ? x : y;
        double v = h < 4 ? y : ((h == 12 || h == 14) ? x : z);
        return ((h & 1) == 0 ? u : -u) + ((h & 2) == 0 ? v : -v);
    }
    // perturbed.rs:85-139
    static double noise(Tuple p) {
        int64_t X = to_i64(std::floor(p.x)) & 255;
        int64_t Y = to_i64(std::floor(p.y)) & 255;
        int64_t Z = to_i64(std::floor(p.z)) & 255;
        double x = p.x - std::floor(p.x);
        double y = p.y - std::floor(p.y);
        double z = p.z - std::floor(p.z);
        double u = fade(x), v = fade(y), w = fade(z);
        int64_t A = perm(X) + Y;
        int64_t AA = perm(A) + Z;
        int64_t AB = perm(A + 1) + Z;
        int64_t B = perm(X + 1) + Y;
        int64_t BA = perm(B) + Z;
        int64_t BB = perm(B + 1) + Z;
        return lerp(
            w,
            lerp(v, lerp(u, grad(perm(AA), x, y, z), grad(perm(BA), x - 1.0, y, z)),
                 lerp(u, grad(perm(AB), x, y - 1.0, z), grad(perm(BB), x - 1.0, y - 1.0, z))),
            lerp(v,
                 lerp(u, grad(perm(AA + 1), x, y, z - 1.0), grad(perm(BA + 1), x - 1.0, y, z - 1.0)),
                 lerp(u, grad(perm(AB + 1), x, y - 1.0, z - 1.0),
                      grad(perm(BB + 1), x - 1.0, y - 1.0, z - 1.0))));
    }

    // pattern.rs:16-22
    Color local_pattern_at(Tuple object_point) const {
        ++g_cnt.pattern_nodes;
        static const int extra[8] = {0, 2, 12, 6, 6, 6, 110, 0};
        g_cnt.flops += (xf.ident ? 0 : 18) + extra[kind];
        Tuple pp = mul(xf.inv(), object_point);
        return pattern_at(pp);
    }
    Color pattern_at(Tuple p) const {
        switch (kind) {
            case P_SOLID:  // solid.rs:21-24
                return color;
            case P_STRIPED:  // striped.rs:36-45
                return (to_i64(std::floor(p.x)) % 2 == 0) ? a->local_pattern_at(p)
                                                          : b->local_pattern_at(p);
            case P_GRADIENT: {  // gradient.rs:36-47
                Color fc = a->local_pattern_at(p);
                Color lc = b->local_pattern_at(p);
                Color distance = lc - fc;
                double fraction = p.x - std::floor(p.x);
                return fc + (distance * fraction);
            }
            case P_RING: {  // ring.rs:32-44
                double xs = p.x * p.x, zs = p.z * p.z;
                return (to_i64(std::sqrt(xs + zs)) % 2 == 0) ? a->local_pattern_at(p)
                                                             : b->local_pattern_at(p);
            }
            case P_CHECKER: {  // checker.rs:36-49
                double s = std::floor(p.x) + std::floor(p.y) + std::floor(p.z);
                return (to_i64(s) % 2 == 0) ? a->local_pattern_at(p) : b->local_pattern_at(p);
            }
            case P_BLENDED: {  // blended.rs:36-44
                Color fc = a->local_pattern_at(p);
                Color lc = b->local_pattern_at(p);
                return (fc + lc) / 2.0;
            }
            case P_PERTURBED: {  // perturbed.rs:142-158
                double n = noise(p) * scale;
                return a->local_pattern_at(point(p.x + n, p.y + n, p.z + n));
            }
            case P_TEST:  // test_pattern.rs:23-26
                return {p.x, p.y, p.z};
        }
        return BLACK;
    }
};

// ---------------------------------------------------------------------------------------------
// Material (materials/phong.rs)
// ---------------------------------------------------------------------------------------------
struct Light {
    Tuple position;
    Color intensity;
};
struct Shape;
struct Material {
    std::shared_ptr<Pattern> pattern;
    double ambient, diffuse, specular, shininess, reflective, transparency, refractive_index;
    Color lighting(const Shape* object, const Light& light, Tuple world_point, Tuple eyev,
                   Tuple normalv, bool in_shadow) const;
};

// ---------------------------------------------------------------------------------------------
// Shapes (geometry/*.rs)
// ---------------------------------------------------------------------------------------------
enum ShapeKind { S_GROUP = 0, S_SPHERE, S_PLANE, S_CUBE, S_CYLINDER, S_CONE, S_TRIANGLE, S_SMOOTH, S_CSG };
enum CsgOp { CSG_DIFFERENCE = 0, CSG_INTERSECTION = 1, CSG_UNION = 2 };  // csg.rs:27-31

struct Hit {
    double t;
    const Shape* obj;
    double u, v;
};

// f64::total_cmp (intersection.rs:60-64)
inline int64_t total_key(double v) {
    int64_t b;
    std::memcpy(&b, &v, 8);
    b ^= (int64_t)((uint64_t)(b >> 63) >> 1);
    return b;
}

struct Shape {
    ShapeKind kind;
    Xform xf{identity()};
    std::shared_ptr<Material> material;
    Shape* parent = nullptr;
    bool casts_shadow = true;
    // cylinder / cone
    double minimum = -INF, maximum = INF;
    bool closed = false;
    // triangles
    Tuple p1{}, p2{}, p3{}, e1{}, e2{}, normal{}, n1{}, n2{}, n3{};
    // group; a CSG shape keeps its left and right operands as children[0] and children[1]
    int op = CSG_UNION;
    std::vector<std::shared_ptr<Shape>> children;
    bool bounds_valid = false;
    BBox cached_bounds{};

    // shape.rs:23-29
    void intersect(const Ray& ray, std::vector<Hit>& out) const {
        M4 inv = xf.inv();
        ++g_cnt.ray_xform;
        if (!xf.ident) {
            ++g_cnt.xform_nonident;
            g_cnt.flops += 33;
        }
        Ray local = xform_ray(ray, inv);
        local_intersect(local, out);
    }

    void local_intersect(const Ray& r, std::vector<Hit>& out) const {
        switch (kind) {
            case S_GROUP: {  // group.rs:133-154
                if (bbox_intersects(const_cast<Shape*>(this)->bounds(), r)) {
                    size_t first = out.size();
                    for (auto& c : children) c->intersect(r, out);
                    if (children.size() > 1)
                        std::stable_sort(out.begin() + first, out.end(), [](const Hit& a, const Hit& b) {
                            return total_key(a.t) < total_key(b.t);
                        });
                }
                return;
            }
            case S_CSG: {  // csg.rs:144-159
                if (bbox_intersects(const_cast<Shape*>(this)->bounds(), r)) {
                    size_t first = out.size();
                    children[0]->intersect(r, out);
                    children[1]->intersect(r, out);
                    std::stable_sort(out.begin() + first, out.end(), [](const Hit& a, const Hit& b) {
                        return total_key(a.t) < total_key(b.t);
                    });
                    filter_intersections(out, first);
                }
                return;
            }
            case S_SPHERE: {  // sphere.rs:52-78
                ++g_cnt.sphere_test;
                Tuple sphere_to_ray = r.o - point(0, 0, 0);
                double a = dot(r.d, r.d);
                double b = 2.0 * dot(r.d, sphere_to_ray);
                double c = dot(sphere_to_ray, sphere_to_ray) - 1.0;
                double disc = (b * b) - (4.0 * a * c);
                g_cnt.flops += 24;
                if (disc < 0.0) return;
                g_cnt.flops += 10;
                double t1 = (-b + std::sqrt(disc)) / (a * 2.0);
                double t2 = (-b - std::sqrt(disc)) / (a * 2.0);
                if (t1 < t2) {
                    out.push_back({t1, this, 0, 0});
                    out.push_back({t2, this, 0, 0});
                } else {
                    out.push_back({t2, this, 0, 0});
                    out.push_back({t1, this, 0, 0});
                }
                return;
            }
            case S_PLANE: {  // plane.rs:51-74
                ++g_cnt.plane_test;
                g_cnt.flops += 4;
                if (std::fabs(r.d.y) < EPSILON) return;
                out.push_back({-r.o.y / r.d.y, this, 0, 0});
                return;
            }
            case S_CUBE: {  // cube.rs:77-98
                ++g_cnt.cube_test;
                g_cnt.flops += 26;
                double x0, x1, y0, y1, z0, z1;
                check_axis(r.o.x, r.d.x, -1.0, 1.0, x0, x1);
                check_axis(r.o.y, r.d.y, -1.0, 1.0, y0, y1);
                check_axis(r.o.z, r.d.z, -1.0, 1.0, z0, z1);
                double tmin = std::fmax(std::fmax(x0, y0), z0);
                double tmax = std::fmin(std::fmin(x1, y1), z1);
                if (tmin > tmax) return;
                out.push_back({tmin, this, 0, 0});
                out.push_back({tmax, this, 0, 0});
                return;
            }
            case S_CYLINDER: {  // cylinder.rs:103-153
                ++g_cnt.cyl_test;
                double a = r.d.x * r.d.x + r.d.z * r.d.z;
                g_cnt.flops += (a < EPSILON) ? 4 : 32;
                if (a < EPSILON) {
                    caps(r, out, false);
                    return;
                }
                double b = (2.0 * r.o.x * r.d.x) + (2.0 * r.o.z * r.d.z);
                double c = (r.o.x * r.o.x) + (r.o.z * r.o.z) - 1.0;
                double disc = (b * b) - (4.0 * a * c);
                if (disc >= 0.0) {
                    double sq = std::sqrt(disc);
                    double t0 = (-b - sq) / (2.0 * a);
                    double t1 = (-b + sq) / (2.0 * a);
                    if (t0 > t1) std::swap(t0, t1);
                    double y0 = r.o.y + t0 * r.d.y;
                    if (minimum < y0 && y0 < maximum) out.push_back({t0, this, 0, 0});
                    double y1 = r.o.y + t1 * r.d.y;
                    if (minimum < y1 && y1 < maximum) out.push_back({t1, this, 0, 0});
                }
                caps(r, out, false);
                return;
            }
            case S_CONE: {  // cone.rs:104-163
                ++g_cnt.cone_test;
                double a = (r.d.x * r.d.x) - (r.d.y * r.d.y) + (r.d.z * r.d.z);
                double b = (2.0 * r.o.x * r.d.x) - (2.0 * r.o.y * r.d.y) + (2.0 * r.o.z * r.d.z);
                double c = (r.o.x * r.o.x) - (r.o.y * r.o.y) + (r.o.z * r.o.z);
                g_cnt.flops += (std::fabs(a) < EPSILON) ? 24 : 40;
                if (std::fabs(a) < EPSILON) {
                    if (std::fabs(b) < EPSILON) return;
                    double t = -c / (2.0 * b);
                    out.push_back({t, this, 0, 0});
                    caps(r, out, true);
                    return;
                }
                double disc = (b * b) - (4.0 * a * c);
                if (disc >= 0.0) {
                    double sq = std::sqrt(disc);
                    double t0 = (-b - sq) / (2.0 * a);
                    double t1 = (-b + sq) / (2.0 * a);
                    if (t0 > t1) std::swap(t0, t1);
                    double y0 = r.o.y + t0 * r.d.y;
                    if (minimum < y0 && y0 < maximum) out.push_back({t0, this, 0, 0});
                    double y1 = r.o.y + t1 * r.d.y;
                    if (minimum < y1 && y1 < maximum) out.push_back({t1, this, 0, 0});
                }
                caps(r, out, true);
                return;
            }
            case S_TRIANGLE:
            case S_SMOOTH: {  // triangle.rs:83-111, smooth_triangle.rs:101-129
                ++g_cnt.tri_test;
                Tuple dir_cross_e2 = cross(r.d, e2);
                double det = dot(e1, dir_cross_e2);
                g_cnt.flops += 16;
                if (std::fabs(det) < EPSILON) return;
                double f = 1.0 / det;
                Tuple p1_to_origin = r.o - p1;
                double u = f * dot(p1_to_origin, dir_cross_e2);
                g_cnt.flops += 12;
                if (u < 0.0 || u > 1.0) return;
                Tuple origin_cross_e1 = cross(p1_to_origin, e1);
                double v = f * dot(r.d, origin_cross_e1);
                g_cnt.flops += 18;
                if (v < 0.0 || (u + v) > 1.0) return;
                double t = f * dot(e2, origin_cross_e1);
                g_cnt.flops += 6;
                if (kind == S_SMOOTH)
                    out.push_back({t, this, u, v});
                else
                    out.push_back({t, this, 0, 0});
                return;
            }
        }
    }

    // cylinder.rs:62-95 / cone.rs:63-96 (cone caps compare against |y|, not y*y)
    void caps(const Ray& r, std::vector<Hit>& out, bool cone) const {
        if (!closed || std::fabs(r.d.y) < EPSILON) return;
        g_cnt.flops += 22;
        double t0 = (minimum - r.o.y) / r.d.y;
        if (cap_ok(r, t0, cone ? std::fabs(minimum) : 1.0)) out.push_back({t0, this, 0, 0});
        double t1 = (maximum - r.o.y) / r.d.y;
        if (cap_ok(r, t1, cone ? std::fabs(maximum) : 1.0)) out.push_back({t1, this, 0, 0});
    }
    static bool cap_ok(const Ray& r, double t, double limit) {
        double x = r.o.x + t * r.d.x;
        double z = r.o.z + t * r.d.z;
        return ((x * x) + (z * z)) <= limit;
    }

    Tuple local_normal_at(Tuple p, const Hit& hit) const {
        switch (kind) {
            case S_SPHERE:  // sphere.rs:109-111
                return p - point(0, 0, 0);
            case S_PLANE:  // plane.rs:105-107
                return mul(xf.inv(), vec(0.0, 1.0, 0.0));
            case S_CUBE: {  // cube.rs:129-142
                double maxc = std::fmax(std::fmax(std::fabs(p.x), std::fabs(p.y)), std::fabs(p.z));
                if (maxc == std::fabs(p.x)) return vec(p.x, 0.0, 0.0);
                if (maxc == std::fabs(p.y)) return vec(0.0, p.y, 0.0);
                return vec(0.0, 0.0, p.z);
            }
            case S_CYLINDER: {  // cylinder.rs:184-197
                double dist = p.x * p.x + p.z * p.z;
                if (dist < 1.0 && p.y >= maximum - EPSILON) return vec(0.0, 1.0, 0.0);
                if (dist < 1.0 && p.y <= minimum + EPSILON) return vec(0.0, -1.0, 0.0);
                return vec(p.x, 0.0, p.z);
            }
            case S_CONE: {  // cone.rs:194-213
                double dist = p.x * p.x + p.z * p.z;
                if (dist < 1.0 && p.y >= maximum - EPSILON) return vec(0.0, 1.0, 0.0);
                if (dist < 1.0 && p.y <= minimum + EPSILON) return vec(0.0, -1.0, 0.0);
                double y = std::sqrt(dist);
                if (p.y > 0.0) return vec(p.x, -y, p.z);
                return vec(p.x, y, p.z);
            }
            case S_TRIANGLE:  // triangle.rs:142-144
                return normal;
            case S_SMOOTH:  // smooth_triangle.rs:160-162
                return (n2 * hit.u) + (n3 * hit.v) + (n1 * (1.0 - hit.u - hit.v));
            case S_GROUP:
            case S_CSG:  // csg.rs:181-183 panics; intersections never carry a CSG shape as their object
                break;
        }
        return vec(0, 0, 0);
    }

    // Shape::includes: leaves compare identity, groups and CSG shapes ask their children
    // (group.rs:172-182, csg.rs:177-179).
    bool includes(const Shape* other) const {
        if (kind != S_GROUP && kind != S_CSG) return this == other;
        for (auto& c : children)
            if (c->includes(other)) return true;
        return false;
    }
    // csg.rs:113-140
    static bool intersection_allowed(int op, bool lhit, bool inl, bool inr) {
        switch (op) {
            case CSG_DIFFERENCE:
                return (lhit && !inr) || (!lhit && inl);
            case CSG_INTERSECTION:
                return (lhit && inr) || (!lhit && inl);
            default:
                return (lhit && !inr) || (!lhit && !inl);
        }
    }
    // csg.rs:80-107, on the tail of `xs` that starts at `first`
    void filter_intersections(std::vector<Hit>& xs, size_t first) const {
        bool inl = false, inr = false;
        size_t keep = first;
        for (size_t i = first; i < xs.size(); ++i) {
            bool lhit = children[0]->includes(xs[i].obj);
            if (intersection_allowed(op, lhit, inl, inr)) xs[keep++] = xs[i];
            if (lhit)
                inl = !inl;
            else
                inr = !inr;
        }
        xs.resize(keep);
    }

    // shape.rs:65-79
    Tuple world_to_object(Tuple p) const {
        Tuple op = parent ? parent->world_to_object(p) : p;
        return mul(xf.inv(), op);
    }
    // shape.rs:83-98
    Tuple normal_to_world(Tuple n) const {
        M4 inv = xf.inv();
        Tuple r = mul(transpose(inv), n);
        r.w = 0.0;
        r = normalize(r);
        return parent ? parent->normal_to_world(r) : r;
    }
    // shape.rs:52-59
    Tuple normal_at(Tuple world_point, const Hit& hit) const {
        Tuple lp = world_to_object(world_point);
        Tuple ln = local_normal_at(lp, hit);
        return normal_to_world(ln);
    }

    BBox bounds() {
        switch (kind) {
            case S_SPHERE:
            case S_CUBE:  // sphere.rs:113-115, cube.rs:144-146
                return {point(-1, -1, -1), point(1, 1, 1)};
            case S_PLANE:  // plane.rs:109-114
                return {point(-INF, 0.0, -INF), point(INF, 0.0, INF)};
            case S_CYLINDER:  // cylinder.rs:199-204
                return {point(-1.0, minimum, -1.0), point(1.0, maximum, 1.0)};
            case S_CONE: {  // cone.rs:215-224
                double limit = std::fmax(std::fabs(minimum), std::fabs(maximum));
                return {point(-limit, minimum, -limit), point(limit, maximum, limit)};
            }
            case S_TRIANGLE:
            case S_SMOOTH:  // triangle.rs:146-153
                return bbox_add_point(bbox_add_point(bbox_add_point(bbox_empty(), p1), p2), p3);
            case S_CSG:      // csg.rs:74-78, 185-202: the same fold over the two operands
            case S_GROUP: {  // group.rs:196-208, 66-77
                if (!bounds_valid) {
                    BBox r = bbox_empty();
                    for (auto& c : children) r = bbox_add(r, c->parent_space_bounds());
                    cached_bounds = r;
                    bounds_valid = true;
                }
                return cached_bounds;
            }
        }
        return bbox_empty();
    }
    BBox parent_space_bounds() { return bbox_transform(bounds(), xf.fwd); }  // shape.rs:103-105

    // group.rs:56-64
    void add_child(const std::shared_ptr<Shape>& c) {
        c->parent = this;
        children.push_back(c);
        bounds_valid = false;
    }
    // group.rs:224-244, 79-115
    void divide(size_t threshold) {
        if (kind == S_CSG) {  // csg.rs:216-219
            children[0]->divide(threshold);
            children[1]->divide(threshold);
            return;
        }
        if (kind != S_GROUP) return;
        if (threshold <= children.size()) {
            BBox lb, rb;
            bbox_split(bounds(), lb, rb);
            std::vector<std::shared_ptr<Shape>> left, right, keep;
            for (auto& c : children) {
                BBox cb = c->parent_space_bounds();
                if (bbox_contains_box(lb, cb))
                    left.push_back(c);
                else if (bbox_contains_box(rb, cb))
                    right.push_back(c);
                else
                    keep.push_back(c);
            }
            children = keep;
            if (!left.empty()) make_subgroup(left);
            if (!right.empty()) make_subgroup(right);
        }
        for (auto& c : children) c->divide(threshold);
    }
    void make_subgroup(const std::vector<std::shared_ptr<Shape>>& list) {
        auto g = std::make_shared<Shape>();
        g->kind = S_GROUP;
        g->material = nullptr;
        for (auto& c : list) g->add_child(c);
        add_child(g);
    }
};

// phong.rs:59-116
Color Material::lighting(const Shape* object, const Light& light, Tuple world_point, Tuple eyev,
                         Tuple normalv, bool in_shadow) const {
    ++g_cnt.light_evals;
    g_cnt.flops += in_shadow ? 30 : 75;
    if (pattern->kind != P_SOLID)
        for (const Shape* s = object; s; s = s->parent)
            if (!s->xf.ident) g_cnt.flops += 18;
    Tuple object_point = object->world_to_object(world_point);  // pattern.rs:8-14
    Color color = pattern->local_pattern_at(object_point);
    Color effective = color * light.intensity;
    Tuple lightv = normalize(light.position - world_point);
    Color amb = effective * ambient;
    if (in_shadow) return amb;
    Color diff, spec;
    double ldn = dot(lightv, normalv);
    if (ldn < 0.0) {
        diff = BLACK;
        spec = BLACK;
    } else {
        diff = effective * diffuse * ldn;
        Tuple reflectv = reflect(-lightv, normalv);
        double rde = dot(reflectv, eyev);
        if (rde <= 0.0) {
            spec = BLACK;
        } else {
            double factor = std::pow(rde, shininess);
            spec = light.intensity * specular * factor;
        }
    }
    return amb + diff + spec;
}

// ---------------------------------------------------------------------------------------------
// World (scene/world.rs) and Computations (scene/computations.rs)
// ---------------------------------------------------------------------------------------------
struct Comps {
    double t;
    const Shape* object;
    Tuple point, over_point, under_point;
    double n1, n2;
    Tuple eyev, normalv, reflectv;
    bool inside;
};

struct World {
    std::vector<std::shared_ptr<Shape>> objects;
    std::vector<Light> lights;

    // world.rs:55-68
    void intersect_world(const Ray& ray, std::vector<Hit>& xs) const {
        xs.clear();
        for (auto& o : objects) o->intersect(ray, xs);
        std::stable_sort(xs.begin(), xs.end(), [](const Hit& a, const Hit& b) { return a.t < b.t; });
    }
    // intersection.rs:32-41
    static int hit_index(const std::vector<Hit>& xs) {
        for (size_t i = 0; i < xs.size(); ++i)
            if (xs[i].t > 0.0) return (int)i;
        return -1;
    }
    // world.rs:70-147
    static Comps prepare(int hit_idx, const Ray& ray, const std::vector<Hit>& xs) {
        ++g_cnt.shaded_hits;
        const Hit& h = xs[hit_idx];
        if (h.obj->kind == S_SMOOTH) ++g_cnt.smooth_hits;
        {
            static const int normal_cost[8] = {0, 3, 15, 8, 8, 8, 0, 17};
            uint64_t f = 60 + normal_cost[h.obj->kind];
            for (const Shape* s = h.obj; s; s = s->parent) f += 9 + (s->xf.ident ? 0 : 33);
            if (h.obj->material->transparency != 0.0) f += 2 * uint64_t(hit_idx + 1);
            g_cnt.flops += f;
        }
        Comps c;
        c.t = h.t;
        c.object = h.obj;
        c.point = position(ray, h.t);
        c.eyev = -ray.d;
        c.normalv = h.obj->normal_at(c.point, h);
        c.inside = false;
        if (dot(c.normalv, c.eyev) < 0.0) {
            c.inside = true;
            c.normalv = -c.normalv;
        }
        c.reflectv = reflect(ray.d, c.normalv);
        c.over_point = c.point + (c.normalv * EPSILON);
        c.under_point = c.point - (c.normalv * EPSILON);
        std::vector<const Shape*> containers;
        c.n1 = 0.0;
        c.n2 = 0.0;
        for (int i = 0; i < (int)xs.size(); ++i) {
            if (i == hit_idx)
                c.n1 = containers.empty() ? 1.0 : containers.back()->material->refractive_index;
            auto it = std::find(containers.begin(), containers.end(), xs[i].obj);
            if (it != containers.end())
                containers.erase(it);
            else
                containers.push_back(xs[i].obj);
            if (i == hit_idx) {
                c.n2 = containers.empty() ? 1.0 : containers.back()->material->refractive_index;
                break;
            }
        }
        return c;
    }
    // world.rs:304-326
    bool is_shadowed(Tuple p, const Light& light) const {
        ++g_cnt.rays_shadow;
        g_cnt.flops += 20;
        Tuple v = light.position - p;
        double distance = magnitude(v);
        Tuple direction = normalize(v);
        Ray r{p, direction};
        std::vector<Hit> xs;
        intersect_world(r, xs);
        int h = hit_index(xs);
        if (h >= 0) {
            if (xs[h].obj->casts_shadow) return xs[h].t < distance;
            return false;
        }
        return false;
    }
    // world.rs:279-302
    static double schlick(const Comps& c) {
        double cosv = dot(c.eyev, c.normalv);
        if (c.n1 > c.n2) {
            double n = c.n1 / c.n2;
            double sin2_t = (n * n) * (1.0 - (cosv * cosv));
            if (sin2_t > 1.0) return 1.0;
            double cos_t = std::cos(1.0 - sin2_t);  // sic: cos, not sqrt (world.rs:293)
            cosv = cos_t;
        }
        double q = (c.n1 - c.n2) / (c.n1 + c.n2);
        double r0 = q * q;
        double m = 1.0 - cosv;
        double m2 = m * m;
        double m5 = m * (m2 * m2);  // powi(5)
        return r0 + (1.0 - r0) * m5;
    }
    // world.rs:214-231
    Color reflected_color(const Comps& c, int remaining) const {
        if (remaining == 0) return BLACK;
        double reflective = c.object->material->reflective;
        if (std::fabs(reflective - 0.0) < EPSILON) return BLACK;
        ++g_cnt.rays_reflect;
        g_cnt.flops += 6;
        Ray r{c.over_point, c.reflectv};
        return color_at(r, remaining - 1) * reflective;
    }
    // world.rs:233-275
    Color refracted_color(const Comps& c, int remaining) const {
        if (remaining == 0) return BLACK;
        double transparency = c.object->material->transparency;
        if (std::fabs(transparency - 0.0) < EPSILON) return BLACK;
        double n_ratio = c.n1 / c.n2;
        double cos_i = dot(c.eyev, c.normalv);
        double sin2_t = (n_ratio * n_ratio) * (1.0 - (cos_i * cos_i));
        g_cnt.flops += 32;
        if (sin2_t > 1.0) return BLACK;
        double cos_t = std::sqrt(1.0 - sin2_t);
        Tuple direction = c.normalv * (n_ratio * cos_i - cos_t) - c.eyev * n_ratio;
        ++g_cnt.rays_refract;
        Ray r{c.under_point, direction};
        return color_at(r, remaining - 1) * transparency;
    }
    // world.rs:162-196
    Color shade_hit(const Comps& c, int remaining) const {
        Color result{0.0, 0.0, 0.0};
        const Material& m = *c.object->material;
        Color reflected{}, refracted{};
        bool have = false;
        for (size_t i = 0; i < lights.size(); ++i) {
            const Light& light = lights[i];
            bool in_shadow = is_shadowed(c.over_point, light);
            Color surface = m.lighting(c.object, light, c.over_point, c.eyev, c.normalv, in_shadow);
            if (g_faithful || !have) {
                reflected = reflected_color(c, remaining);
                refracted = refracted_color(c, remaining);
                have = true;
            }
            if (m.reflective > 0.0 && m.transparency > 0.0) {
                g_cnt.flops += 18 + ((g_faithful || i == 0) ? 22 : 0);
                double reflectance = schlick(c);
                result = result + surface + (reflected * reflectance) + (refracted * (1.0 - reflectance));
            } else {
                g_cnt.flops += 12;
                result = result + surface + reflected + refracted;
            }
        }
        return result;
    }
    // world.rs:198-212
    Color color_at(const Ray& ray, int remaining) const {
        std::vector<Hit> xs;
        intersect_world(ray, xs);
        int h = hit_index(xs);
        if (h >= 0) {
            Comps c = prepare(h, ray, xs);
            return shade_hit(c, remaining);
        }
        return BLACK;
    }
};

// ---------------------------------------------------------------------------------------------
// Camera (scene/camera.rs)
// ---------------------------------------------------------------------------------------------
struct Camera {
    size_t hsize, vsize;
    double fov;
    Xform xf;
    double half_height, half_width, pixel_size;
    // camera.rs:17-43
    Camera(size_t h, size_t v, double f, const M4& t) : hsize(h), vsize(v), fov(f), xf(t) {
        double half_view = std::tan(f / 2.0);
        double aspect = double(h) / double(v);
        if (aspect >= 1.0) {
            half_width = half_view;
            half_height = half_view / aspect;
        } else {
            half_height = half_view;
            half_width = half_view * aspect;
        }
        pixel_size = (half_width * 2.0) / double(h);
    }
    // camera.rs:47-66
    Ray ray_for_pixel(size_t px, size_t py) const {
        double x_offset = (double(px) + 0.5) * pixel_size;
        double y_offset = (double(py) + 0.5) * pixel_size;
        double world_x = half_width - x_offset;
        double world_y = half_height - y_offset;
        M4 inv = xf.inv();
        Tuple pixel = mul(inv, point(world_x, world_y, -1.0));
        Tuple origin = mul(inv, point(0.0, 0.0, 0.0));
        Tuple direction = normalize(pixel - origin);
        return {origin, direction};
    }
    size_t height() const { return hsize; }  // camera.rs:68-70
    size_t width() const { return vsize; }   // camera.rs:72-74
};

// ---------------------------------------------------------------------------------------------
// OBJ parser (scene/obj_file_parser.rs)
// ---------------------------------------------------------------------------------------------
struct ObjParser {
    std::vector<Tuple> vertices, normals;
    std::vector<std::string> group_order;  // insertion order (the reference iterates a HashMap)
    std::map<std::string, std::shared_ptr<Shape>> groups;
    std::string current = "\x01" "default";
    M4 transform = identity();
    std::shared_ptr<Material> material;
    bool casts_shadow = true;
    bool ok = true;

    static bool parse_f64(const std::string& s, double& v) {
        char* end = nullptr;
        v = std::strtod(s.c_str(), &end);
        return end && *end == '\0' && !s.empty();
    }
    static bool parse_usize(const std::string& s, size_t& v) {
        if (s.empty()) return false;
        for (char c : s)
            if (c < '0' || c > '9') return false;
        v = std::strtoull(s.c_str(), nullptr, 10);
        return true;
    }
    // obj_file_parser.rs:40-86
    bool parse(const std::string& path) {
        std::ifstream f(path, std::ios::binary);
        if (!f) return false;
        std::string line;
        while (std::getline(f, line)) {
            if (!line.empty() && line.back() == '\r') line.pop_back();
            std::string trimmed = line;
            size_t a = trimmed.find_first_not_of(" \t\r\n\f\v");
            if (a == std::string::npos) continue;
            std::vector<std::string> p;
            size_t i = 0;
            while (i <= line.size()) {
                size_t j = line.find(' ', i);
                if (j == std::string::npos) j = line.size();
                if (j > i) p.push_back(line.substr(i, j - i));
                i = j + 1;
            }
            if (p.empty()) continue;
            std::string cmd = p[0];
            for (auto& ch : cmd) ch = (char)std::tolower((unsigned char)ch);
            if (cmd == "v") {
                double x, y, z;
                if (p.size() < 4 || !parse_f64(p[1], x) || !parse_f64(p[2], y) || !parse_f64(p[3], z))
                    return false;
                vertices.push_back(point(x, y, z));
            } else if (cmd == "f") {
                if (!face(p)) return false;
            } else if (cmd == "g") {
                if (p.size() < 2) return false;
                current = p[1];
            } else if (cmd == "vn") {
                double x, y, z;
                if (p.size() < 4 || !parse_f64(p[1], x) || !parse_f64(p[2], y) || !parse_f64(p[3], z))
                    return false;
                normals.push_back(vec(x, y, z));
            }
        }
        return true;
    }
    // obj_file_parser.rs:98-136, 156-206
    bool face(const std::vector<std::string>& p) {
        std::vector<size_t> vi, ni;
        std::vector<bool> has_n;
        for (size_t i = 1; i < p.size(); ++i) {
            std::vector<std::string> parts;
            size_t s = 0;
            while (true) {
                size_t j = p[i].find('/', s);
                if (j == std::string::npos) {
                    parts.push_back(p[i].substr(s));
                    break;
                }
                parts.push_back(p[i].substr(s, j - s));
                s = j + 1;
            }
            size_t a = 0, b = 0;
            if (parts.size() == 1) {
                if (!parse_usize(p[i], a)) return false;
                vi.push_back(a);
                ni.push_back(0);
                has_n.push_back(false);
            } else {
                if (parts.size() < 3 || !parse_usize(parts[0], a) || !parse_usize(parts[2], b))
                    return false;
                vi.push_back(a);
                ni.push_back(b);
                has_n.push_back(true);
            }
        }
        std::vector<std::shared_ptr<Shape>> tris;
        for (size_t k = 1; k + 1 < vi.size(); ++k) {
            auto s = std::make_shared<Shape>();
            s->xf = Xform(transform);
            s->material = material;
            s->casts_shadow = casts_shadow;
            s->p1 = vertices.at(vi[0] - 1);
            s->p2 = vertices.at(vi[k] - 1);
            s->p3 = vertices.at(vi[k + 1] - 1);
            s->e1 = s->p2 - s->p1;
            s->e2 = s->p3 - s->p1;
            if (!has_n[0]) {
                s->kind = S_TRIANGLE;
                s->normal = normalize(cross(s->e2, s->e1));  // triangle.rs:46-48
            } else {
                s->kind = S_SMOOTH;
                s->n1 = normals.at(ni[0] - 1);
                s->n2 = normals.at(ni[k] - 1);
                s->n3 = normals.at(ni[k + 1] - 1);
            }
            tris.push_back(s);
        }
        auto it = groups.find(current);
        std::shared_ptr<Shape> g;
        if (it == groups.end()) {
            g = std::make_shared<Shape>();
            g->kind = S_GROUP;
            groups[current] = g;
            group_order.push_back(current);
        } else {
            g = it->second;
        }
        for (auto& t : tris) g->add_child(t);
        return true;
    }
    // obj_file_parser.rs:226-234
    std::shared_ptr<Shape> to_group(const M4& t) {
        auto r = std::make_shared<Shape>();
        r->kind = S_GROUP;
        r->xf = Xform(t);
        for (auto& name : group_order) r->add_child(groups[name]);
        return r;
    }
};

// ---------------------------------------------------------------------------------------------
// Canvas quantiser and P3 writer (window/canvas.rs:50-126)
// ---------------------------------------------------------------------------------------------
inline int clamp_u8(double value) {
    double x = 255.0 * value;
    if (x > 255.0) return 255;
    if (x < 0.0) return 0;
    double c = std::ceil(x);
    if (c != c) return 0;  // NaN as u8 -> 0
    return (int)c;
}
std::string to_ppm(const double* rgb, size_t w, size_t h) {
    std::string out;
    out.reserve(w * h * 12 + 32);
    out += "P3\n" + std::to_string(w) + " " + std::to_string(h) + "\n255\n";
    char buf[8];
    for (size_t y = 0; y < h; ++y) {
        size_t len = 0;
        for (size_t x = 0; x < w; ++x)
            for (int ch = 0; ch < 3; ++ch) {
                int v = clamp_u8(rgb[(y * w + x) * 3 + ch]);
                int n = std::snprintf(buf, sizeof buf, "%d", v);
                if (len + (size_t(n) + 1) >= 70) {
                    out += '\n';
                    len = 0;
                }
                if (len > 0) {
                    out += ' ';
                    len += 1;
                }
                out.append(buf, n);
                len += n;
                if (len + 1 == 70) {
                    out += '\n';
                    len = 0;
                }
            }
        if (len > 0) out += '\n';
    }
    return out;
}

// ---------------------------------------------------------------------------------------------
// Handle registry for the C interface
// ---------------------------------------------------------------------------------------------
struct Registry {
    std::vector<std::shared_ptr<Pattern>> patterns;
    std::vector<std::shared_ptr<Material>> materials;
    std::vector<std::shared_ptr<Shape>> shapes;
    std::vector<std::shared_ptr<World>> worlds;
    std::vector<std::shared_ptr<Camera>> cameras;
    std::vector<std::shared_ptr<ObjParser>> parsers;
    std::map<const Shape*, int64_t> shape_ids;
    int64_t add_shape(const std::shared_ptr<Shape>& s) {
        auto it = shape_ids.find(s.get());
        if (it != shape_ids.end()) return it->second;
        shapes.push_back(s);
        shape_ids[s.get()] = (int64_t)shapes.size() - 1;
        return (int64_t)shapes.size() - 1;
    }
};
Registry g_reg;
std::string g_ppm;

Tuple t4(const double* p) { return {p[0], p[1], p[2], p[3]}; }
void put4(Tuple t, double* p) {
    p[0] = t.x;
    p[1] = t.y;
    p[2] = t.z;
    p[3] = t.w;
}
void putm(const M4& m, double* p) { std::memcpy(p, m.m, sizeof(m.m)); }

}  // namespace

extern "C" {

void orc_reset(void) {
    g_reg = Registry();
    g_ppm.clear();
}
void orc_set_faithful(int on) { g_faithful = on != 0; }

// ---- matrices (row-major double[16]) ----
void orc_mat_identity(double* o) { putm(identity(), o); }
void orc_mat_translation(double x, double y, double z, double* o) { putm(translation(x, y, z), o); }
void orc_mat_scaling(double x, double y, double z, double* o) { putm(scaling(x, y, z), o); }
void orc_mat_rotation_x(double a, double* o) { putm(rotation_x(a), o); }
void orc_mat_rotation_y(double a, double* o) { putm(rotation_y(a), o); }
void orc_mat_rotation_z(double a, double* o) { putm(rotation_z(a), o); }
void orc_mat_shearing(double a, double b, double c, double d, double e, double f, double* o) {
    putm(shearing(a, b, c, d, e, f), o);
}
void orc_mat_view_transform(const double* from, const double* to, const double* up, double* o) {
    putm(view_transform(point(from[0], from[1], from[2]), point(to[0], to[1], to[2]),
                        vec(up[0], up[1], up[2])),
         o);
}
void orc_mat_mul(const double* a, const double* b, double* o) { putm(mul(from_rows(a), from_rows(b)), o); }
int orc_mat_inverse(const double* a, double* o) {
    M4 r;
    bool ok = g_faithful ? inverse_dyn(from_rows(a), r) : inverse_fast(from_rows(a), r);
    if (!ok) return -1;
    putm(r, o);
    return 0;
}
void orc_mat_transpose(const double* a, double* o) { putm(transpose(from_rows(a)), o); }
void orc_mat_mul_tuple(const double* a, const double* t, double* o) { put4(mul(from_rows(a), t4(t)), o); }
double orc_mat_determinant(const double* a) { return det4(from_rows(a)); }
double orc_mat_cofactor(const double* a, int r, int c) { return cofactor4(from_rows(a), r, c); }

// ---- tuples ----
void orc_tuple_normalize(const double* t, double* o) { put4(normalize(t4(t)), o); }
double orc_tuple_magnitude(const double* t) { return magnitude(t4(t)); }
double orc_tuple_dot(const double* a, const double* b) { return dot(t4(a), t4(b)); }
void orc_tuple_cross(const double* a, const double* b, double* o) { put4(cross(t4(a), t4(b)), o); }
void orc_tuple_reflect(const double* a, const double* n, double* o) { put4(reflect(t4(a), t4(n)), o); }

// ---- patterns / materials ----
int64_t orc_pattern_solid(double r, double g, double b) {
    auto p = std::make_shared<Pattern>();
    p->kind = P_SOLID;
    p->color = {r, g, b};
    g_reg.patterns.push_back(p);
    return (int64_t)g_reg.patterns.size() - 1;
}
int64_t orc_pattern_binary(int kind, int64_t a, int64_t b, const double* m) {
    auto p = std::make_shared<Pattern>();
    p->kind = (PatKind)kind;
    p->a = g_reg.patterns.at(a);
    p->b = g_reg.patterns.at(b);
    p->xf = Xform(from_rows(m));
    g_reg.patterns.push_back(p);
    return (int64_t)g_reg.patterns.size() - 1;
}
int64_t orc_pattern_perturbed(int64_t delegate, double scale, const double* m) {
    auto p = std::make_shared<Pattern>();
    p->kind = P_PERTURBED;
    p->a = g_reg.patterns.at(delegate);
    p->scale = scale;
    p->xf = Xform(from_rows(m));
    g_reg.patterns.push_back(p);
    return (int64_t)g_reg.patterns.size() - 1;
}
int64_t orc_pattern_test(const double* m) {
    auto p = std::make_shared<Pattern>();
    p->kind = P_TEST;
    p->xf = Xform(from_rows(m));
    g_reg.patterns.push_back(p);
    return (int64_t)g_reg.patterns.size() - 1;
}
int64_t orc_phong(int64_t pattern, double ambient, double diffuse, double specular, double shininess,
                  double reflective, double transparency, double refractive_index) {
    auto m = std::make_shared<Material>();
    m->pattern = g_reg.patterns.at(pattern);
    m->ambient = ambient;
    m->diffuse = diffuse;
    m->specular = specular;
    m->shininess = shininess;
    m->reflective = reflective;
    m->transparency = transparency;
    m->refractive_index = refractive_index;
    g_reg.materials.push_back(m);
    return (int64_t)g_reg.materials.size() - 1;
}

// ---- shapes ----
int64_t orc_shape(int kind, const double* m, int64_t material, int casts_shadow, double minimum,
                  double maximum, int closed) {
    auto s = std::make_shared<Shape>();
    s->kind = (ShapeKind)kind;
    s->xf = Xform(from_rows(m));
    s->material = g_reg.materials.at(material);
    s->casts_shadow = casts_shadow != 0;
    s->minimum = minimum;
    s->maximum = maximum;
    s->closed = closed != 0;
    return g_reg.add_shape(s);
}
int64_t orc_triangle(const double* p, const double* m, int64_t material, int casts_shadow) {
    auto s = std::make_shared<Shape>();
    s->kind = S_TRIANGLE;
    s->xf = Xform(from_rows(m));
    s->material = g_reg.materials.at(material);
    s->casts_shadow = casts_shadow != 0;
    s->p1 = point(p[0], p[1], p[2]);
    s->p2 = point(p[3], p[4], p[5]);
    s->p3 = point(p[6], p[7], p[8]);
    s->e1 = s->p2 - s->p1;
    s->e2 = s->p3 - s->p1;
    s->normal = normalize(cross(s->e2, s->e1));
    return g_reg.add_shape(s);
}
int64_t orc_smooth_triangle(const double* p, const double* n, const double* m, int64_t material,
                            int casts_shadow) {
    auto s = std::make_shared<Shape>();
    s->kind = S_SMOOTH;
    s->xf = Xform(from_rows(m));
    s->material = g_reg.materials.at(material);
    s->casts_shadow = casts_shadow != 0;
    s->p1 = point(p[0], p[1], p[2]);
    s->p2 = point(p[3], p[4], p[5]);
    s->p3 = point(p[6], p[7], p[8]);
    s->e1 = s->p2 - s->p1;
    s->e2 = s->p3 - s->p1;
    s->n1 = vec(n[0], n[1], n[2]);
    s->n2 = vec(n[3], n[4], n[5]);
    s->n3 = vec(n[6], n[7], n[8]);
    return g_reg.add_shape(s);
}
int64_t orc_group(const double* m) {
    auto s = std::make_shared<Shape>();
    s->kind = S_GROUP;
    s->xf = Xform(from_rows(m));
    return g_reg.add_shape(s);
}
// CSG::new (csg.rs:34-59): casts_shadow is false, both operands get the new shape as parent.
int64_t orc_csg(int op, int64_t left, int64_t right, const double* m, int64_t material) {
    auto s = std::make_shared<Shape>();
    s->kind = S_CSG;
    s->op = op;
    s->xf = Xform(from_rows(m));
    s->material = g_reg.materials.at(material);
    s->casts_shadow = false;
    s->add_child(g_reg.shapes.at(left));
    s->add_child(g_reg.shapes.at(right));
    return g_reg.add_shape(s);
}
int orc_csg_allowed(int op, int lhit, int inl, int inr) {
    return Shape::intersection_allowed(op, lhit != 0, inl != 0, inr != 0) ? 1 : 0;
}
// CSG::filter_intersections on a caller-supplied list; returns how many survive, keeps them in place.
int orc_csg_filter(int64_t h, int n, double* ts, int64_t* objs) {
    std::vector<Hit> xs;
    for (int i = 0; i < n; ++i) xs.push_back({ts[i], g_reg.shapes.at(objs[i]).get(), 0, 0});
    g_reg.shapes.at(h)->filter_intersections(xs, 0);
    for (size_t i = 0; i < xs.size(); ++i) {
        ts[i] = xs[i].t;
        objs[i] = g_reg.shape_ids.at(xs[i].obj);
    }
    return (int)xs.size();
}
int orc_group_add_child(int64_t g, int64_t c) {
    auto& grp = g_reg.shapes.at(g);
    if (grp->kind != S_GROUP) return -1;
    grp->add_child(g_reg.shapes.at(c));
    return 0;
}
int orc_shape_kind(int64_t h) { return (int)g_reg.shapes.at(h)->kind; }
int orc_group_num_children(int64_t g) { return (int)g_reg.shapes.at(g)->children.size(); }
int64_t orc_group_child(int64_t g, int i) { return g_reg.add_shape(g_reg.shapes.at(g)->children.at(i)); }
int64_t orc_shape_parent(int64_t h) {
    Shape* p = g_reg.shapes.at(h)->parent;
    if (!p) return -1;
    auto it = g_reg.shape_ids.find(p);
    return it == g_reg.shape_ids.end() ? -2 : it->second;
}
void orc_divide(int64_t h, int threshold) { g_reg.shapes.at(h)->divide((size_t)threshold); }
void orc_bounds(int64_t h, double* o) {
    BBox b = g_reg.shapes.at(h)->bounds();
    o[0] = b.mn.x; o[1] = b.mn.y; o[2] = b.mn.z; o[3] = b.mx.x; o[4] = b.mx.y; o[5] = b.mx.z;
}
void orc_parent_space_bounds(int64_t h, double* o) {
    BBox b = g_reg.shapes.at(h)->parent_space_bounds();
    o[0] = b.mn.x; o[1] = b.mn.y; o[2] = b.mn.z; o[3] = b.mx.x; o[4] = b.mx.y; o[5] = b.mx.z;
}
void orc_triangle_data(int64_t h, double* o) {  // p1,p2,p3,e1,e2,normal,n1,n2,n3 (27 doubles)
    auto& s = g_reg.shapes.at(h);
    Tuple v[9] = {s->p1, s->p2, s->p3, s->e1, s->e2, s->normal, s->n1, s->n2, s->n3};
    for (int i = 0; i < 9; ++i) {
        o[i * 3] = v[i].x; o[i * 3 + 1] = v[i].y; o[i * 3 + 2] = v[i].z;
    }
}

// ---- bounding boxes (standalone, for the bounding_box.rs KATs) ----
int orc_bbox_intersects(const double* b, const double* o, const double* d) {
    BBox bb{point(b[0], b[1], b[2]), point(b[3], b[4], b[5])};
    return bbox_intersects(bb, Ray{t4(o), t4(d)}) ? 1 : 0;
}
int orc_bbox_contains_point(const double* b, const double* p) {
    BBox bb{point(b[0], b[1], b[2]), point(b[3], b[4], b[5])};
    return bbox_contains_point(bb, point(p[0], p[1], p[2])) ? 1 : 0;
}
int orc_bbox_contains_box(const double* b, const double* o) {
    BBox bb{point(b[0], b[1], b[2]), point(b[3], b[4], b[5])};
    BBox ob{point(o[0], o[1], o[2]), point(o[3], o[4], o[5])};
    return bbox_contains_box(bb, ob) ? 1 : 0;
}
void orc_bbox_transform(const double* b, const double* m, double* o) {
    BBox bb{point(b[0], b[1], b[2]), point(b[3], b[4], b[5])};
    BBox r = bbox_transform(bb, from_rows(m));
    o[0] = r.mn.x; o[1] = r.mn.y; o[2] = r.mn.z; o[3] = r.mx.x; o[4] = r.mx.y; o[5] = r.mx.z;
}
void orc_bbox_split(const double* b, double* l, double* r) {
    BBox bb{point(b[0], b[1], b[2]), point(b[3], b[4], b[5])}, lb, rb;
    bbox_split(bb, lb, rb);
    l[0] = lb.mn.x; l[1] = lb.mn.y; l[2] = lb.mn.z; l[3] = lb.mx.x; l[4] = lb.mx.y; l[5] = lb.mx.z;
    r[0] = rb.mn.x; r[1] = rb.mn.y; r[2] = rb.mn.z; r[3] = rb.mx.x; r[4] = rb.mx.y; r[5] = rb.mx.z;
}

// ---- shape evaluation ----
// Shape::intersect (shape.rs:23-29).  Writes up to `cap` (t, object handle, u, v) records.
int orc_intersect(int64_t h, const double* o, const double* d, double* ts, int64_t* objs, double* uv,
                  int cap) {
    std::vector<Hit> xs;
    g_reg.shapes.at(h)->intersect(Ray{t4(o), t4(d)}, xs);
    for (int i = 0; i < (int)xs.size() && i < cap; ++i) {
        ts[i] = xs[i].t;
        auto it = g_reg.shape_ids.find(xs[i].obj);
        objs[i] = it == g_reg.shape_ids.end() ? -2 : it->second;
        if (uv) {
            uv[2 * i] = xs[i].u;
            uv[2 * i + 1] = xs[i].v;
        }
    }
    return (int)xs.size();
}
void orc_normal_at(int64_t h, const double* p, double u, double v, double* o) {
    Hit hit{0.0, g_reg.shapes.at(h).get(), u, v};
    put4(g_reg.shapes.at(h)->normal_at(t4(p), hit), o);
}
void orc_local_normal_at(int64_t h, const double* p, double u, double v, double* o) {
    Hit hit{0.0, g_reg.shapes.at(h).get(), u, v};
    put4(g_reg.shapes.at(h)->local_normal_at(t4(p), hit), o);
}
void orc_world_to_object(int64_t h, const double* p, double* o) {
    put4(g_reg.shapes.at(h)->world_to_object(t4(p)), o);
}
void orc_normal_to_world(int64_t h, const double* n, double* o) {
    put4(g_reg.shapes.at(h)->normal_to_world(t4(n)), o);
}
void orc_pattern_at_shape(int64_t pat, int64_t shape, const double* p, double* o) {
    Tuple op = g_reg.shapes.at(shape)->world_to_object(t4(p));
    Color c = g_reg.patterns.at(pat)->local_pattern_at(op);
    o[0] = c.r; o[1] = c.g; o[2] = c.b;
}
void orc_pattern_at(int64_t pat, const double* p, double* o) {
    Color c = g_reg.patterns.at(pat)->pattern_at(t4(p));
    o[0] = c.r; o[1] = c.g; o[2] = c.b;
}
double orc_perlin_noise(const double* p) { return Pattern::noise(t4(p)); }
void orc_lighting(int64_t material, int64_t shape, const double* lpos, const double* lcol,
                  const double* p, const double* eyev, const double* normalv, int in_shadow, double* o) {
    Light l{point(lpos[0], lpos[1], lpos[2]), {lcol[0], lcol[1], lcol[2]}};
    Color c = g_reg.materials.at(material)->lighting(g_reg.shapes.at(shape).get(), l, t4(p), t4(eyev),
                                                     t4(normalv), in_shadow != 0);
    o[0] = c.r; o[1] = c.g; o[2] = c.b;
}

// ---- OBJ ----
int64_t orc_obj_parse(const char* path, const double* m, int64_t material, int casts_shadow) {
    auto p = std::make_shared<ObjParser>();
    p->transform = from_rows(m);
    p->material = g_reg.materials.at(material);
    p->casts_shadow = casts_shadow != 0;
    if (!p->parse(path)) return -1;
    g_reg.parsers.push_back(p);
    return (int64_t)g_reg.parsers.size() - 1;
}
int orc_obj_num_vertices(int64_t h) { return (int)g_reg.parsers.at(h)->vertices.size(); }
int orc_obj_num_normals(int64_t h) { return (int)g_reg.parsers.at(h)->normals.size(); }
void orc_obj_vertex(int64_t h, int i, double* o) { put4(g_reg.parsers.at(h)->vertices.at(i), o); }
void orc_obj_normal(int64_t h, int i, double* o) { put4(g_reg.parsers.at(h)->normals.at(i), o); }
int orc_obj_num_groups(int64_t h) { return (int)g_reg.parsers.at(h)->group_order.size(); }
int64_t orc_obj_group(int64_t h, const char* name) {  // name == NULL: the current group
    auto& p = g_reg.parsers.at(h);
    auto it = p->groups.find(name ? std::string(name) : p->current);
    if (it == p->groups.end()) return -1;
    return g_reg.add_shape(it->second);
}
int64_t orc_obj_to_group(int64_t h, const double* m) {
    return g_reg.add_shape(g_reg.parsers.at(h)->to_group(from_rows(m)));
}

// ---- world / camera ----
int64_t orc_world(void) {
    g_reg.worlds.push_back(std::make_shared<World>());
    return (int64_t)g_reg.worlds.size() - 1;
}
void orc_world_add_object(int64_t w, int64_t s) { g_reg.worlds.at(w)->objects.push_back(g_reg.shapes.at(s)); }
void orc_world_add_light(int64_t w, const double* pos, const double* col) {
    g_reg.worlds.at(w)->lights.push_back({point(pos[0], pos[1], pos[2]), {col[0], col[1], col[2]}});
}
int64_t orc_camera(int64_t hsize, int64_t vsize, double fov, const double* m) {
    g_reg.cameras.push_back(std::make_shared<Camera>((size_t)hsize, (size_t)vsize, fov, from_rows(m)));
    return (int64_t)g_reg.cameras.size() - 1;
}
void orc_camera_info(int64_t c, double* o) {  // half_width, half_height, pixel_size, width(), height()
    auto& cam = g_reg.cameras.at(c);
    o[0] = cam->half_width; o[1] = cam->half_height; o[2] = cam->pixel_size;
    o[3] = (double)cam->width(); o[4] = (double)cam->height();
}
void orc_ray_for_pixel(int64_t c, int64_t px, int64_t py, double* o, double* d) {
    Ray r = g_reg.cameras.at(c)->ray_for_pixel((size_t)px, (size_t)py);
    put4(r.o, o);
    put4(r.d, d);
}
int orc_world_intersect(int64_t w, const double* o, const double* d, double* ts, int64_t* objs, int cap) {
    std::vector<Hit> xs;
    g_reg.worlds.at(w)->intersect_world(Ray{t4(o), t4(d)}, xs);
    for (int i = 0; i < (int)xs.size() && i < cap; ++i) {
        ts[i] = xs[i].t;
        auto it = g_reg.shape_ids.find(xs[i].obj);
        objs[i] = it == g_reg.shape_ids.end() ? -2 : it->second;
    }
    return (int)xs.size();
}
// prepare_computations for the hit of `ray` in the world (world.rs:70-147).  Returns 0 when the
// ray hits nothing.  out: t, point[4], over[4], under[4], eyev[4], normalv[4], reflectv[4], n1, n2,
// inside, object handle  (30 doubles)
int orc_prepare(int64_t w, const double* o, const double* d, double* out) {
    std::vector<Hit> xs;
    Ray r{t4(o), t4(d)};
    g_reg.worlds.at(w)->intersect_world(r, xs);
    int h = World::hit_index(xs);
    if (h < 0) return 0;
    Comps c = World::prepare(h, r, xs);
    out[0] = c.t;
    put4(c.point, out + 1);
    put4(c.over_point, out + 5);
    put4(c.under_point, out + 9);
    put4(c.eyev, out + 13);
    put4(c.normalv, out + 17);
    put4(c.reflectv, out + 21);
    out[25] = c.n1;
    out[26] = c.n2;
    out[27] = c.inside ? 1.0 : 0.0;
    auto it = g_reg.shape_ids.find(c.object);
    out[28] = it == g_reg.shape_ids.end() ? -2.0 : (double)it->second;
    out[29] = World::schlick(c);
    return 1;
}
// prepare_computations on a caller-supplied intersection list (hit_index into ts/objs, optional
// uv pairs), then one of shade_hit (0) / reflected_color (1) / refracted_color (2) /
// (schlick, n1, n2) (3) / the Computations themselves (4: same 30-double layout as orc_prepare).
// World handle may be -1 for 3 and 4.
int orc_comps_eval(int64_t w, const double* o, const double* d, const double* ts, const int64_t* objs,
                   const double* uv, int n, int hit_index, int which, int remaining, double* out) {
    std::vector<Hit> xs;
    for (int i = 0; i < n; ++i)
        xs.push_back({ts[i], g_reg.shapes.at(objs[i]).get(), uv ? uv[2 * i] : 0.0, uv ? uv[2 * i + 1] : 0.0});
    Ray r{t4(o), t4(d)};
    Comps c = World::prepare(hit_index, r, xs);
    Color col{0, 0, 0};
    switch (which) {
        case 0: col = g_reg.worlds.at(w)->shade_hit(c, remaining); break;
        case 1: col = g_reg.worlds.at(w)->reflected_color(c, remaining); break;
        case 2: col = g_reg.worlds.at(w)->refracted_color(c, remaining); break;
        case 3: col = {World::schlick(c), c.n1, c.n2}; break;
        case 4: {
            out[0] = c.t;
            put4(c.point, out + 1);
            put4(c.over_point, out + 5);
            put4(c.under_point, out + 9);
            put4(c.eyev, out + 13);
            put4(c.normalv, out + 17);
            put4(c.reflectv, out + 21);
            out[25] = c.n1;
            out[26] = c.n2;
            out[27] = c.inside ? 1.0 : 0.0;
            auto it = g_reg.shape_ids.find(c.object);
            out[28] = it == g_reg.shape_ids.end() ? -2.0 : (double)it->second;
            out[29] = World::schlick(c);
            return 0;
        }
        default: return -1;
    }
    out[0] = col.r; out[1] = col.g; out[2] = col.b;
    return 0;
}
void orc_color_at(int64_t w, const double* o, const double* d, int remaining, double* out) {
    Color c = g_reg.worlds.at(w)->color_at(Ray{t4(o), t4(d)}, remaining);
    out[0] = c.r; out[1] = c.g; out[2] = c.b;
}
// color_at for n rays given as packed xyz origins / directions (w = 1 / 0).
void orc_color_at_batch(int64_t w, int64_t n, const double* o3, const double* d3, int remaining,
                        double* out) {
    auto& world = g_reg.worlds.at(w);
    for (int64_t i = 0; i < n; ++i) {
        Ray r{point(o3[3 * i], o3[3 * i + 1], o3[3 * i + 2]), vec(d3[3 * i], d3[3 * i + 1], d3[3 * i + 2])};
        Color c = world->color_at(r, remaining);
        out[3 * i] = c.r; out[3 * i + 1] = c.g; out[3 * i + 2] = c.b;
    }
}
int orc_is_shadowed(int64_t w, const double* p, const double* lpos, const double* lcol) {
    Light l{point(lpos[0], lpos[1], lpos[2]), {lcol[0], lcol[1], lcol[2]}};
    return g_reg.worlds.at(w)->is_shadowed(t4(p), l) ? 1 : 0;
}

// render (lib.rs:77-133): rows [row0, row1) of the camera.width() x camera.height() canvas into
// rgb (full-frame layout, (y*W + x)*3).  Work is split over `threads` workers striding the linear
// pixel index like lib.rs:94-111.  counters (orc_num_counters() x u64) may be NULL.
static void warm_bounds(Shape* s) {
    if (s->kind != S_GROUP && s->kind != S_CSG) return;
    s->bounds();
    for (auto& c : s->children) warm_bounds(c.get());
}
void orc_render_rows(int64_t w, int64_t c, int remaining, int64_t row0, int64_t row1, int threads,
                     double* rgb, uint64_t* counters) {
    auto world = g_reg.worlds.at(w);
    auto cam = g_reg.cameras.at(c);
    for (auto& o : world->objects) warm_bounds(o.get());  // lazy caches are filled before threads start
    int64_t W = (int64_t)cam->width();
    int64_t first = row0 * W, last = row1 * W;
    bool faithful = g_faithful;
    if (threads < 1) threads = 1;
    std::vector<Counters> per(threads);
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([&, t]() {
            g_faithful = faithful;
            g_cnt = Counters();
            for (int64_t i = first + t; i < last; i += threads) {
                int64_t x = i % W, y = i / W;
                Ray r = cam->ray_for_pixel((size_t)x, (size_t)y);
                ++g_cnt.rays_primary;
                g_cnt.flops += 36;
                Color col = world->color_at(r, remaining);
                rgb[i * 3] = col.r;
                rgb[i * 3 + 1] = col.g;
                rgb[i * 3 + 2] = col.b;
            }
            per[t] = g_cnt;
        });
    for (auto& th : pool) th.join();
    if (counters) {
        Counters total;
        for (auto& p : per) total.add(p);
        std::memcpy(counters, &total, sizeof(Counters));
    }
}
int orc_num_counters(void) { return (int)(sizeof(Counters) / sizeof(uint64_t)); }

// ---- canvas ----
int orc_clamp_u8(double v) { return clamp_u8(v); }
int64_t orc_to_ppm(const double* rgb, int64_t w, int64_t h) {
    g_ppm = to_ppm(rgb, (size_t)w, (size_t)h);
    return (int64_t)g_ppm.size();
}
void orc_ppm_copy(char* dst) { std::memcpy(dst, g_ppm.data(), g_ppm.size()); }

}  // extern "C"
