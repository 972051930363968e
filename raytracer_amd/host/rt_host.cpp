// librt_host: the host side above the C ABI of include/rt_abi.h.
//
// It stands where the reference's Rust crate stands: it offers the reference's scene-construction
// interface (matrices, patterns, Phong, shapes, groups with `divide`, the OBJ loader, World,
// Camera, the PPM canvas writer), flattens a World into the tables rt_scene_create takes --
// computing every inverse with the reference's cofactor algorithm, so the device sees the same
// bits the reference would compute per ray -- and implements `render` by calling the HIP library.
// There is no CPU rendering here: without librt_amd and a GPU, render fails with an error.
//
// Reference citations are file:line in the reference checkout.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <limits>
#include <map>
#include <stdexcept>
#include <memory>
#include <string>
#include <vector>

#include "rt_abi.h"

namespace {

const double EPSILON = 0.00001;  // src/lib.rs:22
const double INF = std::numeric_limits<double>::infinity();

thread_local std::string g_error;

// ---- tuples ---------------------------------------------------------------------------------
struct Tuple {
    double x, y, z, w;
};
Tuple point(double x, double y, double z) { return {x, y, z, 1.0}; }
Tuple vec(double x, double y, double z) { return {x, y, z, 0.0}; }
Tuple sub(Tuple a, Tuple b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
Tuple cross(Tuple a, Tuple b) {  // tuple.rs:83-92
    return vec(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
Tuple normalize(Tuple a) {  // tuple.rs:49-66
    double m = std::sqrt((a.x * a.x) + (a.y * a.y) + (a.z * a.z) + (a.w * a.w));
    return {a.x / m, a.y / m, a.z / m, a.w / m};
}

// ---- Matrix (matrices/matrix.rs) --------------------------------------------------------------
struct Matrix {
    double m[4][4];
    static Matrix identity() {
        Matrix r{};
        for (int i = 0; i < 4; ++i) r.m[i][i] = 1.0;
        return r;
    }
    static Matrix rows(const double* p) {
        Matrix r;
        std::memcpy(r.m, p, sizeof r.m);
        return r;
    }
    void store(double* p) const { std::memcpy(p, m, sizeof m); }
    bool is_identity() const {
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
                if (m[i][j] != (i == j ? 1.0 : 0.0)) return false;
        return true;
    }
};
Matrix translation(double x, double y, double z) {  // matrix.rs:56-65
    Matrix r = Matrix::identity();
    r.m[0][3] = x;
    r.m[1][3] = y;
    r.m[2][3] = z;
    return r;
}
Matrix scaling(double x, double y, double z) {  // matrix.rs:80-89
    Matrix r = Matrix::identity();
    r.m[0][0] = x;
    r.m[1][1] = y;
    r.m[2][2] = z;
    return r;
}
Matrix rotation(int axis, double a) {  // matrix.rs:91-122
    Matrix r = Matrix::identity();
    double c = std::cos(a), s = std::sin(a);
    if (axis == 0) {
        r.m[1][1] = c; r.m[1][2] = -s; r.m[2][1] = s; r.m[2][2] = c;
    } else if (axis == 1) {
        r.m[0][0] = c; r.m[0][2] = s; r.m[2][0] = -s; r.m[2][2] = c;
    } else {
        r.m[0][0] = c; r.m[0][1] = -s; r.m[1][0] = s; r.m[1][1] = c;
    }
    return r;
}
Matrix shearing(double xy, double xz, double yx, double yz, double zx, double zy) {  // matrix.rs:128-137
    Matrix r = Matrix::identity();
    r.m[0][1] = xy; r.m[0][2] = xz; r.m[1][0] = yx; r.m[1][2] = yz; r.m[2][0] = zx; r.m[2][1] = zy;
    return r;
}
Matrix multiply(const Matrix& a, const Matrix& b) {  // matrix.rs:367-401
    Matrix r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double sum = 0.0;
            for (int k = 0; k < 4; ++k) sum = sum + (a.m[i][k] * b.m[k][j]);
            r.m[i][j] = sum;
        }
    return r;
}
Tuple multiply(const Matrix& a, Tuple t) {  // matrix.rs:344-363
    double r[4];
    for (int i = 0; i < 4; ++i) r[i] = a.m[i][0] * t.x + a.m[i][1] * t.y + a.m[i][2] * t.z + a.m[i][3] * t.w;
    return {r[0], r[1], r[2], r[3]};
}
Matrix transposed(const Matrix& a) {  // matrix.rs:175-192
    Matrix r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) r.m[j][i] = a.m[i][j];
    return r;
}
// Cofactor expansion exactly as matrix.rs:233-325 orders it: determinants accumulate
// `d += element * cofactor` along row 0 starting from 0.0; 3x3 cofactors negate on odd
// (row + column); 4x4 cofactors multiply by (-1)^(row + column).
double minor2(const double s[3][3], int skip_col) {
    double q[4];
    int n = 0;
    for (int y = 1; y < 3; ++y)
        for (int x = 0; x < 3; ++x)
            if (x != skip_col) q[n++] = s[y][x];
    return q[0] * q[3] - q[1] * q[2];
}
double determinant3(const double s[3][3]) {
    double d = 0.0;
    for (int c = 0; c < 3; ++c) {
        double minor = minor2(s, c);
        double cof = (c % 2 != 0) ? -minor : minor;
        d += s[0][c] * cof;
    }
    return d;
}
double cofactor(const Matrix& a, int row, int col) {
    double s[3][3];
    int sr = 0;
    for (int y = 0; y < 4; ++y) {
        if (y == row) continue;
        int sc = 0;
        for (int x = 0; x < 4; ++x)
            if (x != col) s[sr][sc++] = a.m[y][x];
        ++sr;
    }
    double sign = ((row + col) % 2 == 0) ? 1.0 : -1.0;
    return sign * determinant3(s);
}
double determinant(const Matrix& a) {
    double d = 0.0;
    for (int c = 0; c < 4; ++c) d += a.m[0][c] * cofactor(a, 0, c);
    return d;
}
bool inverse(const Matrix& a, Matrix& out) {  // matrix.rs:201-226
    double d = determinant(a);
    if ((std::fabs(d) - 0.0) < EPSILON) return false;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) out.m[c][r] = cofactor(a, r, c) / d;
    return true;
}
Matrix view_transform(Tuple from, Tuple to, Tuple up) {  // matrix.rs:145-173
    Tuple forward = normalize(sub(to, from));
    Tuple upn = normalize(up);
    Tuple left = cross(forward, upn);
    Tuple true_up = cross(left, forward);
    Matrix o = Matrix::identity();
    o.m[0][0] = left.x; o.m[0][1] = left.y; o.m[0][2] = left.z;
    o.m[1][0] = true_up.x; o.m[1][1] = true_up.y; o.m[1][2] = true_up.z;
    o.m[2][0] = -forward.x; o.m[2][1] = -forward.y; o.m[2][2] = -forward.z;
    return multiply(o, translation(-from.x, -from.y, -from.z));
}

// ---- BoundingBox (tuples/bounding_box.rs) ---------------------------------------------------------
struct BoundingBox {
    Tuple min, max;
    static BoundingBox empty() { return {point(INF, INF, INF), point(-INF, -INF, -INF)}; }
    BoundingBox add_point(Tuple p) const {  // :25-39 (f64::min/max ignore NaN like fmin/fmax)
        return {point(std::fmin(min.x, p.x), std::fmin(min.y, p.y), std::fmin(min.z, p.z)),
                point(std::fmax(max.x, p.x), std::fmax(max.y, p.y), std::fmax(max.z, p.z))};
    }
    BoundingBox merged(const BoundingBox& o) const { return add_point(o.min).add_point(o.max); }  // :172-178
    bool contains_box(const BoundingBox& o) const {                                            // :50-57
        return o.min.x >= min.x && o.max.x <= max.x && o.min.y >= min.y && o.max.y <= max.y &&
               o.min.z >= min.z && o.max.z <= max.z;
    }
    bool finite_and_ordered() const {
        return std::isfinite(min.x) && std::isfinite(min.y) && std::isfinite(min.z) && std::isfinite(max.x) &&
               std::isfinite(max.y) && std::isfinite(max.z) && min.x <= max.x && min.y <= max.y && min.z <= max.z;
    }
    BoundingBox plus_zero() const {  // what the identity matrix makes of the corners: -0 becomes +0
        return {point(min.x + 0.0, min.y + 0.0, min.z + 0.0), point(max.x + 0.0, max.y + 0.0, max.z + 0.0)};
    }
    BoundingBox transform(const Matrix& t) const {  // :59-78
        Tuple p[8] = {min,
                      point(min.x, min.y, max.z),
                      point(min.x, max.y, min.z),
                      point(min.x, max.y, max.z),
                      point(max.x, min.y, min.z),
                      point(max.x, min.y, max.z),
                      point(max.x, max.y, min.z),
                      max};
        BoundingBox r = empty();
        for (auto& q : p) r = r.add_point(multiply(t, q));
        return r;
    }
    void split_bounds(BoundingBox& left, BoundingBox& right) const {  // :126-161
        double dx = max.x - min.x, dy = max.y - min.y, dz = max.z - min.z;
        double greatest = std::fmax(dx, std::fmax(dy, dz));
        double x0 = min.x, y0 = min.y, z0 = min.z, x1 = max.x, y1 = max.y, z1 = max.z;
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
        left = {min, point(x1, y1, z1)};
        right = {point(x0, y0, z0), max};
    }
};

// ---- patterns, material ---------------------------------------------------------------------------
struct Pattern {
    int kind = RT_PAT_SOLID;
    double color[3] = {1.0, 1.0, 1.0};
    std::shared_ptr<Pattern> a, b;
    double scale = 0.0;
    Matrix transform = Matrix::identity();
};
struct Phong {  // materials/phong.rs:10-19
    std::shared_ptr<Pattern> pattern;
    double ambient, diffuse, specular, shininess, reflective, transparency, refractive_index;
};

// ---- shapes ---------------------------------------------------------------------------------------
struct Shape {
    int kind = RT_GROUP;
    // Boxes are kept once computed (a leaf's never change, a group's until its children do), next to
    // `kind` at the start of the object: `divide` asks every child for its box at every level, and
    // with tens of thousands of triangles each object touched is a cache miss.
    bool bounds_valid = false, ps_valid = false;
    BoundingBox cached{}, ps_cached{};
    Matrix transform = Matrix::identity();
    std::shared_ptr<Phong> material;
    Shape* parent = nullptr;
    bool casts_shadow = true;
    double minimum = -INF, maximum = INF;
    bool closed = false;
    Tuple p1{}, p2{}, p3{}, e1{}, e2{}, normal{}, n1{}, n2{}, n3{};
    std::vector<std::shared_ptr<Shape>> children;  // RT_CSG: left and right operand
    int op = 2;                                     // RT_CSG: 0 difference, 1 intersection, 2 union (csg.rs:27-31)

    // Shape::includes (group.rs:172-182, csg.rs:177-179; leaves compare identity)
    bool includes(const Shape* other) const {
        if (kind != RT_GROUP && kind != RT_CSG) return this == other;
        for (auto& c : children)
            if (c->includes(other)) return true;
        return false;
    }
    static bool intersection_allowed(int op, bool lhit, bool inl, bool inr) {  // csg.rs:113-140
        if (op == 0) return (lhit && !inr) || (!lhit && inl);
        if (op == 1) return (lhit && inr) || (!lhit && inl);
        return (lhit && !inr) || (!lhit && !inl);
    }

    BoundingBox bounds() {
        switch (kind) {
            case RT_SPHERE:
            case RT_CUBE:  // sphere.rs:113-115, cube.rs:144-146
                return {point(-1, -1, -1), point(1, 1, 1)};
            case RT_PLANE:  // plane.rs:109-114
                return {point(-INF, 0.0, -INF), point(INF, 0.0, INF)};
            case RT_CYLINDER:  // cylinder.rs:199-204
                return {point(-1.0, minimum, -1.0), point(1.0, maximum, 1.0)};
            case RT_CONE: {  // cone.rs:215-224
                double limit = std::fmax(std::fabs(minimum), std::fabs(maximum));
                return {point(-limit, minimum, -limit), point(limit, maximum, limit)};
            }
            case RT_TRIANGLE:
            case RT_SMOOTH_TRIANGLE:  // triangle.rs:146-153 (the corners never change: computed once)
                if (!bounds_valid) {
                    cached = BoundingBox::empty().add_point(p1).add_point(p2).add_point(p3);
                    bounds_valid = true;
                }
                return cached;
            default:  // group.rs:196-208, 66-77; csg.rs:74-78 folds its two operands the same way
                if (!bounds_valid) {
                    BoundingBox r = BoundingBox::empty();
                    for (auto& c : children) r = r.merged(c->parent_space_bounds_of());
                    cached = r;
                    bounds_valid = true;
                }
                return cached;
        }
    }
    // This group's children changed.  Only its own stored box is dropped, as in the reference
    // (group.rs:56-64): a group further out that has been asked for its box before keeps that box.
    void boxes_changed() { bounds_valid = ps_valid = false; }
    BoundingBox parent_space_bounds_of() {  // shape.rs:103-105
        if (ps_valid) return ps_cached;
        ps_cached = parent_space_bounds_uncached();
        ps_valid = true;
        return ps_cached;
    }
    BoundingBox parent_space_bounds_uncached() {
        BoundingBox b = bounds();
        // The identity maps the eight corners of a finite, ordered box to themselves (1 * x + 0 * y +
        // 0 * z + 0 * 1 is x, with +0 for either zero), so the box of the products is the box itself:
        // `divide` asks for every child's box at every level, mostly of untransformed triangles.
        if (b.finite_and_ordered() && transform.is_identity()) return b.plus_zero();
        return b.transform(transform);
    }
    void add_child(std::shared_ptr<Shape> c) {  // group.rs:56-64
        c->parent = this;
        children.push_back(std::move(c));
        boxes_changed();
    }
    void make_subgroup(std::vector<std::shared_ptr<Shape>>& list) {  // group.rs:107-115 (empties `list`)
        auto g = std::make_shared<Shape>();
        g->kind = RT_GROUP;
        g->children.reserve(list.size());
        for (auto& c : list) g->add_child(std::move(c));
        list.clear();
        add_child(std::move(g));
    }
    void divide(size_t threshold) {  // group.rs:224-244, 79-105
        if (kind == RT_CSG) {        // csg.rs:216-219
            children[0]->divide(threshold);
            children[1]->divide(threshold);
            return;
        }
        if (kind != RT_GROUP) return;
        if (threshold <= children.size()) {
            BoundingBox lb, rb;
            bounds().split_bounds(lb, rb);
            std::vector<std::shared_ptr<Shape>> left, right, keep;
            for (auto& c : children) {
                BoundingBox cb = c->parent_space_bounds_of();
                if (lb.contains_box(cb))
                    left.push_back(std::move(c));
                else if (rb.contains_box(cb))
                    right.push_back(std::move(c));
                else
                    keep.push_back(std::move(c));
            }
            children = std::move(keep);
            if (!left.empty()) make_subgroup(left);
            if (!right.empty()) make_subgroup(right);
        }
        for (auto& c : children) c->divide(threshold);
    }
};

std::shared_ptr<Shape> make_triangle(const double* p, const double* n, const Matrix& t,
                                     const std::shared_ptr<Phong>& mat, bool cs) {
    auto s = std::make_shared<Shape>();
    s->kind = n ? RT_SMOOTH_TRIANGLE : RT_TRIANGLE;
    s->transform = t;
    s->material = mat;
    s->casts_shadow = cs;
    s->p1 = point(p[0], p[1], p[2]);
    s->p2 = point(p[3], p[4], p[5]);
    s->p3 = point(p[6], p[7], p[8]);
    s->e1 = sub(s->p2, s->p1);  // triangle.rs:46-48, smooth_triangle.rs:52-53
    s->e2 = sub(s->p3, s->p1);
    if (n) {
        s->n1 = vec(n[0], n[1], n[2]);
        s->n2 = vec(n[3], n[4], n[5]);
        s->n3 = vec(n[6], n[7], n[8]);
    } else {
        s->normal = normalize(cross(s->e2, s->e1));
    }
    return s;
}

// ---- OBJ loader (scene/obj_file_parser.rs) ----------------------------------------------------------
struct ObjFileParser {
    std::vector<Tuple> vertices, vertex_normals;
    std::vector<std::string> order;  // groups in first-use order (the reference iterates a HashMap)
    std::map<std::string, std::shared_ptr<Shape>> groups;
    std::string current_group = std::string(1, '\x01') + "default";
    Matrix transform = Matrix::identity();
    std::shared_ptr<Phong> material;
    bool casts_shadow = true;

    static bool number(const std::string& s, double& v) {
        if (s.empty()) return false;
        char* end = nullptr;
        v = std::strtod(s.c_str(), &end);
        return end && *end == '\0';
    }
    static bool index(const std::string& s, size_t& v) {
        if (s.empty()) return false;
        for (char c : s)
            if (c < '0' || c > '9') return false;
        v = std::strtoull(s.c_str(), nullptr, 10);
        return true;
    }
    static std::vector<std::string> split(const std::string& s, char sep, bool keep_empty) {
        std::vector<std::string> out;
        size_t i = 0;
        while (i <= s.size()) {
            size_t j = s.find(sep, i);
            if (j == std::string::npos) j = s.size();
            if (keep_empty || j > i) out.push_back(s.substr(i, j - i));
            i = j + 1;
        }
        return out;
    }
    bool parse_obj_file(const std::string& path) {  // :40-86
        std::ifstream f(path, std::ios::binary);
        if (!f) return false;
        std::string line;
        while (std::getline(f, line)) {
            if (!line.empty() && line.back() == '\r') line.pop_back();
            if (line.find_first_not_of(" \t\r\n\f\v") == std::string::npos) continue;
            std::vector<std::string> p = split(line, ' ', false);
            if (p.empty()) continue;
            std::string cmd = p[0];
            for (auto& ch : cmd) ch = (char)std::tolower((unsigned char)ch);
            if (cmd == "v" || cmd == "vn") {  // :88-96, :144-152
                double x, y, z;
                if (p.size() < 4 || !number(p[1], x) || !number(p[2], y) || !number(p[3], z)) return false;
                if (cmd == "v")
                    vertices.push_back(point(x, y, z));
                else
                    vertex_normals.push_back(vec(x, y, z));
            } else if (cmd == "f") {
                if (!face(p)) return false;
            } else if (cmd == "g") {  // :138-142
                if (p.size() < 2) return false;
                current_group = p[1];
            }
        }
        return true;
    }
    bool face(const std::vector<std::string>& p) {  // :98-136 and fan_triangulation :156-206
        std::vector<size_t> vi, ni;
        bool smooth = false;
        for (size_t i = 1; i < p.size(); ++i) {
            std::vector<std::string> parts = split(p[i], '/', true);
            size_t a = 0, b = 0;
            if (parts.size() == 1) {
                if (!index(p[i], a)) return false;
                if (i == 1) smooth = false;
            } else {
                if (parts.size() < 3 || !index(parts[0], a) || !index(parts[2], b)) return false;
                if (i == 1) smooth = true;
            }
            vi.push_back(a);
            ni.push_back(b);
        }
        std::vector<std::shared_ptr<Shape>> tris;
        for (size_t k = 1; k + 1 < vi.size(); ++k) {
            size_t ids[3] = {vi[0], vi[k], vi[k + 1]};
            size_t nds[3] = {ni[0], ni[k], ni[k + 1]};
            double pts[9], nrm[9];
            for (int c = 0; c < 3; ++c) {
                if (ids[c] < 1 || ids[c] > vertices.size()) return false;
                const Tuple& v = vertices[ids[c] - 1];
                pts[c * 3] = v.x; pts[c * 3 + 1] = v.y; pts[c * 3 + 2] = v.z;
                if (smooth) {
                    if (nds[c] < 1 || nds[c] > vertex_normals.size()) return false;
                    const Tuple& n = vertex_normals[nds[c] - 1];
                    nrm[c * 3] = n.x; nrm[c * 3 + 1] = n.y; nrm[c * 3 + 2] = n.z;
                }
            }
            tris.push_back(make_triangle(pts, smooth ? nrm : nullptr, transform, material, casts_shadow));
        }
        auto it = groups.find(current_group);
        std::shared_ptr<Shape> g;
        if (it == groups.end()) {
            g = std::make_shared<Shape>();
            groups[current_group] = g;
            order.push_back(current_group);
        } else {
            g = it->second;
        }
        for (auto& t : tris) g->add_child(t);
        return true;
    }
    std::shared_ptr<Shape> obj_to_group(const Matrix& t) {  // :226-234
        auto r = std::make_shared<Shape>();
        r->transform = t;
        for (auto& name : order) r->add_child(groups[name]);
        return r;
    }
};

// ---- World, Camera ----------------------------------------------------------------------------------
struct Light {
    double position[3];
    double intensity[3];
};
struct Flat {
    std::vector<rt_node> nodes;
    std::vector<rt_xform> xforms;
    std::vector<rt_aabb> aabbs;
    std::vector<rt_range> ranges;
    std::vector<rt_tri> tris;
    std::vector<rt_tri_normals> trin;
    std::vector<rt_material> materials;
    std::vector<rt_pattern> patterns;
    std::vector<double> patx;
    std::vector<rt_light> lights;
};
struct World {
    std::vector<std::shared_ptr<Shape>> objects;
    std::vector<Light> lights;
    rt_scene* scene = nullptr;  // device copy, created on first render
    int device = -1;            // the GPU `scene` lives on
    bool exported = false;      // rth_world_scene handed `scene` out as a raw handle
    // Scenes whose raw handle a caller may still hold: they stay valid (describing the world as it
    // was) until the world goes away, instead of dangling when the world changes or moves to
    // another GPU.
    std::vector<rt_scene*> retired;
    Flat flat;
    bool flattened = false;
    void drop_scene() {
        if (!scene) return;
        if (exported)
            retired.push_back(scene);
        else
            rt_scene_destroy(scene);
        scene = nullptr;
        exported = false;
        device = -1;
    }
    ~World() {
        if (scene) rt_scene_destroy(scene);
        for (rt_scene* r : retired) rt_scene_destroy(r);
    }
};
struct Camera {  // scene/camera.rs:17-43
    size_t hsize, vsize;
    double field_of_view;
    Matrix transform;
    double half_height, half_width, pixel_size;
    Camera(size_t h, size_t v, double fov, const Matrix& t) : hsize(h), vsize(v), field_of_view(fov), transform(t) {
        double half_view = std::tan(fov / 2.0);
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
    size_t height() const { return hsize; }  // camera.rs:68-70
    size_t width() const { return vsize; }   // camera.rs:72-74
};

// ---- flatten: World -> rt_scene_desc tables --------------------------------------------------------
struct Flattener {
    Flat& f;
    std::map<const Phong*, int> material_index;
    std::string error;

    explicit Flattener(Flat& out) : f(out) {}

    // Rows 0..2 of the cofactor inverse and the 3x3 of its transpose.  Rejects what the reference
    // would panic on (|det| < EPSILON) and anything whose inverse is not affine.
    bool make_xform(const Matrix& t, rt_xform& x) {
        Matrix inv;
        if (!inverse(t, inv)) {
            error = "Error: Matrix is not invertible due to a zero determinant";
            return false;
        }
        if (inv.m[3][0] != 0.0 || inv.m[3][1] != 0.0 || inv.m[3][2] != 0.0 || inv.m[3][3] != 1.0) {
            error = "transform is not affine (bottom row of the inverse is not 0 0 0 1)";
            return false;
        }
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c) x.inv[r * 4 + c] = inv.m[r][c];
        for (int i = 0; i < 3; ++i)
            for (int k = 0; k < 3; ++k) x.inv_t[i * 3 + k] = inv.m[k][i];
        x.reserved[0] = x.reserved[1] = x.reserved[2] = 0.0;
        return true;
    }
    int add_pattern(const Pattern* p) {
        int idx = (int)f.patterns.size();
        f.patterns.push_back(rt_pattern{});
        rt_pattern rp{};
        rp.type = p->kind;
        rp.a = rp.b = -1;
        rp.xform = -1;
        rp.color[0] = p->color[0];
        rp.color[1] = p->color[1];
        rp.color[2] = p->color[2];
        rp.scale = p->scale;
        if (p->kind != RT_PAT_SOLID && !p->transform.is_identity()) {  // Solid's transform is the identity (solid.rs:26-28)
            rt_xform x;
            if (!make_xform(p->transform, x)) return -1;
            rp.xform = (int)(f.patx.size() / 12);
            f.patx.insert(f.patx.end(), x.inv, x.inv + 12);
        }
        if (p->a) {
            rp.a = add_pattern(p->a.get());
            if (rp.a < 0) return -1;
        }
        if (p->b) {
            rp.b = add_pattern(p->b.get());
            if (rp.b < 0) return -1;
        }
        f.patterns[idx] = rp;
        return idx;
    }
    int add_material(const Phong* m) {
        auto it = material_index.find(m);
        if (it != material_index.end()) return it->second;
        rt_material rm{};
        rm.ambient = m->ambient;
        rm.diffuse = m->diffuse;
        rm.specular = m->specular;
        rm.shininess = m->shininess;
        rm.reflective = m->reflective;
        rm.transparency = m->transparency;
        rm.refractive_index = m->refractive_index;
        rm.pattern = add_pattern(m->pattern.get());
        if (rm.pattern < 0) return -1;
        f.materials.push_back(rm);
        material_index[m] = (int)f.materials.size() - 1;
        return (int)f.materials.size() - 1;
    }
    bool add_shape(Shape* s, int parent, int ctx) {
        int idx = (int)f.nodes.size();
        f.nodes.push_back(rt_node{});
        rt_node n{};
        n.type = s->kind;
        n.flags = (s->casts_shadow ? RT_NODE_CASTS_SHADOW : 0) | (s->closed ? RT_NODE_CLOSED : 0);
        n.parent = parent;
        n.material = -1;
        n.xform = -1;
        n.data = -1;
        n.ctx = ctx;
        if (!s->transform.is_identity()) {
            rt_xform x;
            if (!make_xform(s->transform, x)) return false;
            n.xform = (int)f.xforms.size();
            f.xforms.push_back(x);
        }
        if (s->kind == RT_CSG) n.flags = (s->casts_shadow ? RT_NODE_CASTS_SHADOW : 0) | (s->op << 4);
        if (s->kind == RT_GROUP || s->kind == RT_CSG) {
            BoundingBox b = s->bounds();
            rt_aabb a{{b.min.x, b.min.y, b.min.z}, {b.max.x, b.max.y, b.max.z}};
            n.data = (int)f.aabbs.size();
            f.aabbs.push_back(a);
            f.nodes[idx] = n;
            int child_ctx = n.xform >= 0 ? idx : ctx;
            for (auto& c : s->children)
                if (!add_shape(c.get(), idx, child_ctx)) return false;
            f.nodes[idx].skip = (int)f.nodes.size();
            return true;
        }
        if (!s->material) {
            error = "leaf shape without a material";
            return false;
        }
        n.material = add_material(s->material.get());
        if (n.material < 0) return false;
        if (s->kind == RT_CYLINDER || s->kind == RT_CONE) {
            n.data = (int)f.ranges.size();
            f.ranges.push_back(rt_range{s->minimum, s->maximum});
        } else if (s->kind == RT_TRIANGLE || s->kind == RT_SMOOTH_TRIANGLE) {
            n.data = (int)f.tris.size();
            rt_tri t{{s->p1.x, s->p1.y, s->p1.z}, {s->e1.x, s->e1.y, s->e1.z}, {s->e2.x, s->e2.y, s->e2.z}};
            rt_tri_normals tn{};
            if (s->kind == RT_TRIANGLE) {
                tn.n1[0] = s->normal.x; tn.n1[1] = s->normal.y; tn.n1[2] = s->normal.z;
            } else {
                tn.n1[0] = s->n1.x; tn.n1[1] = s->n1.y; tn.n1[2] = s->n1.z;
                tn.n2[0] = s->n2.x; tn.n2[1] = s->n2.y; tn.n2[2] = s->n2.z;
                tn.n3[0] = s->n3.x; tn.n3[1] = s->n3.y; tn.n3[2] = s->n3.z;
            }
            f.tris.push_back(t);
            f.trin.push_back(tn);
        }
        n.skip = idx + 1;
        f.nodes[idx] = n;
        return true;
    }
    bool run(World& w) {
        for (auto& o : w.objects)
            if (!add_shape(o.get(), -1, -1)) return false;
        for (auto& l : w.lights) {
            rt_light rl;
            std::memcpy(rl.position, l.position, sizeof rl.position);
            std::memcpy(rl.intensity, l.intensity, sizeof rl.intensity);
            f.lights.push_back(rl);
        }
        return true;
    }
};

bool flatten(World& w) {
    if (w.flattened) return true;
    w.flat = Flat();
    Flattener fl(w.flat);
    if (!fl.run(w)) {
        g_error = fl.error;
        return false;
    }
    w.flattened = true;
    return true;
}

rt_scene_desc describe(const Flat& f) {
    rt_scene_desc d{};
    d.n_nodes = (int)f.nodes.size();
    d.nodes = f.nodes.data();
    d.n_xforms = (int)f.xforms.size();
    d.xforms = f.xforms.data();
    d.n_aabbs = (int)f.aabbs.size();
    d.aabbs = f.aabbs.data();
    d.n_ranges = (int)f.ranges.size();
    d.ranges = f.ranges.data();
    d.n_tris = (int)f.tris.size();
    d.tris = f.tris.data();
    d.tri_normals = f.trin.data();
    d.n_materials = (int)f.materials.size();
    d.materials = f.materials.data();
    d.n_patterns = (int)f.patterns.size();
    d.patterns = f.patterns.data();
    d.n_pattern_xforms = (int)(f.patx.size() / 12);
    d.pattern_xforms = f.patx.data();
    d.n_lights = (int)f.lights.size();
    d.lights = f.lights.data();
    return d;
}

bool camera_block(const Camera& c, rt_camera& out) {
    Matrix inv;
    if (!inverse(c.transform, inv)) {  // Camera::ray_for_pixel unwraps the inverse (camera.rs:60)
        g_error = "Error: Matrix is not invertible due to a zero determinant";
        return false;
    }
    out.width = (int32_t)c.width();
    out.height = (int32_t)c.height();
    out.half_width = c.half_width;
    out.half_height = c.half_height;
    out.pixel_size = c.pixel_size;
    for (int r = 0; r < 3; ++r)
        for (int k = 0; k < 4; ++k) out.inv[r * 4 + k] = inv.m[r][k];
    return true;
}

// ---- Canvas quantiser and P3 writer (window/canvas.rs:50-126) -------------------------------------
int clamp(double value) {
    double x = 255.0 * value;
    if (x > 255.0) return 255;
    if (x < 0.0) return 0;
    double c = std::ceil(x);
    return (c != c) ? 0 : (int)c;
}
// Canvas::to_ppm (canvas.rs:62-126) over values produced by `value(index)`, already clamped.
template <class F>
std::string ppm_text(size_t w, size_t h, F value) {
    std::string out;
    out.reserve(w * h * 12 + 32);
    out += "P3\n" + std::to_string(w) + " " + std::to_string(h) + "\n255\n";
    char buf[8];
    for (size_t y = 0; y < h; ++y) {
        size_t len = 0;
        for (size_t i = 0; i < w * 3; ++i) {
            int n = std::snprintf(buf, sizeof buf, "%d", value(y * w * 3 + i));
            if (len + (size_t(n) + 1) >= 70) {
                out += '\n';
                len = 0;
            }
            if (len > 0) {
                out += ' ';
                ++len;
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
std::string to_ppm(const double* rgb, size_t w, size_t h) {
    return ppm_text(w, h, [rgb](size_t i) { return clamp(rgb[i]); });
}
std::string to_ppm_u8(const uint8_t* rgb8, size_t w, size_t h) {
    return ppm_text(w, h, [rgb8](size_t i) { return (int)rgb8[i]; });
}

// ---- handle registry ----------------------------------------------------------------------------------
struct Registry {
    std::vector<std::shared_ptr<Pattern>> patterns;
    std::vector<std::shared_ptr<Phong>> materials;
    std::vector<std::shared_ptr<Shape>> shapes;
    std::vector<std::shared_ptr<World>> worlds;
    std::vector<std::shared_ptr<Camera>> cameras;
    std::vector<std::shared_ptr<ObjFileParser>> parsers;
    std::map<const Shape*, int64_t> ids;
    int64_t add(const std::shared_ptr<Shape>& s) {
        auto it = ids.find(s.get());
        if (it != ids.end()) return it->second;
        shapes.push_back(s);
        ids[s.get()] = (int64_t)shapes.size() - 1;
        return (int64_t)shapes.size() - 1;
    }
};
Registry g_reg;
std::string g_ppm;

void put_box(const BoundingBox& b, double* o) {
    o[0] = b.min.x; o[1] = b.min.y; o[2] = b.min.z; o[3] = b.max.x; o[4] = b.max.y; o[5] = b.max.z;
}
void put4(Tuple t, double* o) {
    o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
}
bool contains_shape(const Shape* root, const Shape* target) {
    if (root == target) return true;
    for (auto& c : root->children)
        if (contains_shape(c.get(), target)) return true;
    return false;
}
// A shape changed (children added, divided, wrapped in a CSG shape): the worlds that hold it must be
// flattened and uploaded again; the others keep their device copies.
void invalidate_worlds_with(const Shape* changed) {
    for (auto& w : g_reg.worlds) {
        bool holds = false;
        for (auto& o : w->objects)
            if (contains_shape(o.get(), changed)) {
                holds = true;
                break;
            }
        if (!holds) continue;
        w->drop_scene();
        w->flattened = false;
    }
}

}  // namespace

extern "C" {

const char* rth_last_error(void) { return g_error.empty() ? rt_last_error() : g_error.c_str(); }
void rth_reset(void) { try {
    g_reg = Registry();
    g_ppm.clear();
} catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); } }

// ---- matrices ----
void rth_mat_identity(double* o) { try { Matrix::identity().store(o); } catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); } }
void rth_mat_translation(double x, double y, double z, double* o) { try { translation(x, y, z).store(o); } catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); } }
void rth_mat_scaling(double x, double y, double z, double* o) { try { scaling(x, y, z).store(o); } catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); } }
void rth_mat_rotation_x(double a, double* o) { try { rotation(0, a).store(o); } catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); } }
void rth_mat_rotation_y(double a, double* o) { try { rotation(1, a).store(o); } catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); } }
void rth_mat_rotation_z(double a, double* o) { try { rotation(2, a).store(o); } catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); } }
void rth_mat_shearing(double a, double b, double c, double d, double e, double f, double* o) { try {
    shearing(a, b, c, d, e, f).store(o);
} catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); } }
void rth_mat_view_transform(const double* from, const double* to, const double* up, double* o) { try {
    view_transform(point(from[0], from[1], from[2]), point(to[0], to[1], to[2]), vec(up[0], up[1], up[2])).store(o);
} catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); } }
void rth_mat_mul(const double* a, const double* b, double* o) { try { multiply(Matrix::rows(a), Matrix::rows(b)).store(o); } catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); } }
int rth_mat_inverse(const double* a, double* o) { try {
    Matrix r;
    if (!inverse(Matrix::rows(a), r)) return -1;
    r.store(o);
    return 0;
} catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); return -1; } }
void rth_mat_transpose(const double* a, double* o) { try { transposed(Matrix::rows(a)).store(o); } catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); } }
void rth_mat_mul_tuple(const double* a, const double* t, double* o) { try {
    put4(multiply(Matrix::rows(a), Tuple{t[0], t[1], t[2], t[3]}), o);
} catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); } }

// ---- patterns / materials ----
static int64_t push_pattern(const std::shared_ptr<Pattern>& p) {
    g_reg.patterns.push_back(p);
    return (int64_t)g_reg.patterns.size() - 1;
}
int64_t rth_pattern_solid(double r, double g, double b) { try {
    auto p = std::make_shared<Pattern>();
    p->kind = RT_PAT_SOLID;
    p->color[0] = r; p->color[1] = g; p->color[2] = b;
    return push_pattern(p);
} catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); return -1; } }
int64_t rth_pattern_binary(int kind, int64_t a, int64_t b, const double* m) { try {
    if (kind < RT_PAT_STRIPED || kind > RT_PAT_BLENDED) return -1;
    auto p = std::make_shared<Pattern>();
    p->kind = kind;
    p->a = g_reg.patterns.at(a);
    p->b = g_reg.patterns.at(b);
    p->transform = Matrix::rows(m);
    return push_pattern(p);
} catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); return -1; } }
int64_t rth_pattern_perturbed(int64_t delegate, double scale, const double* m) { try {
    auto p = std::make_shared<Pattern>();
    p->kind = RT_PAT_PERTURBED;
    p->a = g_reg.patterns.at(delegate);
    p->scale = scale;
    p->transform = Matrix::rows(m);
    return push_pattern(p);
} catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); return -1; } }
int64_t rth_pattern_test(const double* m) { try {
    auto p = std::make_shared<Pattern>();
    p->kind = RT_PAT_TEST;
    p->transform = Matrix::rows(m);
    return push_pattern(p);
} catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); return -1; } }
int64_t rth_phong(int64_t pattern, double ambient, double diffuse, double specular, double shininess,
                  double reflective, double transparency, double refractive_index) { try {
    auto m = std::make_shared<Phong>();
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
} catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); return -1; } }

// ---- shapes ----
int64_t rth_shape(int kind, const double* m, int64_t material, int casts_shadow, double minimum, double maximum,
                  int closed) { try {
    if (kind < RT_SPHERE || kind > RT_CONE) return -1;
    auto s = std::make_shared<Shape>();
    s->kind = kind;
    s->transform = Matrix::rows(m);
    s->material = g_reg.materials.at(material);
    s->casts_shadow = casts_shadow != 0;
    s->minimum = minimum;
    s->maximum = maximum;
    s->closed = closed != 0;
    return g_reg.add(s);
} catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); return -1; } }
int64_t rth_triangle(const double* p, const double* m, int64_t material, int casts_shadow) { try {
    return g_reg.add(make_triangle(p, nullptr, Matrix::rows(m), g_reg.materials.at(material), casts_shadow != 0));
} catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); return -1; } }
int64_t rth_smooth_triangle(const double* p, const double* n, const double* m, int64_t material, int casts_shadow) { try {
    return g_reg.add(make_triangle(p, n, Matrix::rows(m), g_reg.materials.at(material), casts_shadow != 0));
} catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); return -1; } }
int64_t rth_group(const double* m) { try {
    auto s = std::make_shared<Shape>();
    s->transform = Matrix::rows(m);
    return g_reg.add(s);
} catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); return -1; } }
// CSG::new (csg.rs:34-59): casts_shadow is false, both operands get the new shape as parent.
int64_t rth_csg(int op, int64_t left, int64_t right, const double* m, int64_t material) { try {
    auto s = std::make_shared<Shape>();
    s->kind = RT_CSG;
    s->op = op;
    s->transform = Matrix::rows(m);
    s->material = g_reg.materials.at(material);
    s->casts_shadow = false;
    s->add_child(g_reg.shapes.at(left));
    s->add_child(g_reg.shapes.at(right));
    invalidate_worlds_with(s->children[0].get());
    invalidate_worlds_with(s->children[1].get());
    return g_reg.add(s);
} catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); return -1; } }
int rth_csg_allowed(int op, int lhit, int inl, int inr) { try {
    return Shape::intersection_allowed(op, lhit != 0, inl != 0, inr != 0) ? 1 : 0;
} catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); return -1; } }
// CSG::filter_intersections (csg.rs:80-107) on a caller-supplied list, survivors kept in place.
int rth_csg_filter(int64_t h, int n, double* ts, int64_t* objs) { try {
    auto& csg = g_reg.shapes.at(h);
    bool inl = false, inr = false;
    int keep = 0;
    for (int i = 0; i < n; ++i) {
        bool lhit = csg->children.at(0)->includes(g_reg.shapes.at(objs[i]).get());
        if (Shape::intersection_allowed(csg->op, lhit, inl, inr)) {
            ts[keep] = ts[i];
            objs[keep] = objs[i];
            ++keep;
        }
        if (lhit)
            inl = !inl;
        else
            inr = !inr;
    }
    return keep;
} catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); return -1; } }
int rth_group_add_child(int64_t g, int64_t c) { try {
    auto& grp = g_reg.shapes.at(g);
    if (grp->kind != RT_GROUP) return -1;
    grp->add_child(g_reg.shapes.at(c));
    invalidate_worlds_with(grp.get());
    return 0;
} catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); return -1; } }
int rth_shape_kind(int64_t h) { try { return g_reg.shapes.at(h)->kind; } catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); return -1; } }
int rth_group_num_children(int64_t g) { try { return (int)g_reg.shapes.at(g)->children.size(); } catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); return -1; } }
int64_t rth_group_child(int64_t g, int i) { try { return g_reg.add(g_reg.shapes.at(g)->children.at(i)); } catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); return -1; } }
void rth_divide(int64_t h, int threshold) { try {
    g_reg.shapes.at(h)->divide((size_t)threshold);
    invalidate_worlds_with(g_reg.shapes.at(h).get());
} catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); } }
void rth_bounds(int64_t h, double* o) { try { put_box(g_reg.shapes.at(h)->bounds(), o); } catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); } }
void rth_parent_space_bounds(int64_t h, double* o) { try { put_box(g_reg.shapes.at(h)->parent_space_bounds_of(), o); } catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); } }
void rth_triangle_data(int64_t h, double* o) { try {
    auto& s = g_reg.shapes.at(h);
    Tuple v[9] = {s->p1, s->p2, s->p3, s->e1, s->e2, s->normal, s->n1, s->n2, s->n3};
    for (int i = 0; i < 9; ++i) {
        o[i * 3] = v[i].x; o[i * 3 + 1] = v[i].y; o[i * 3 + 2] = v[i].z;
    }
} catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); } }

// ---- OBJ ----
int64_t rth_obj_parse(const char* path, const double* m, int64_t material, int casts_shadow) { try {
    auto p = std::make_shared<ObjFileParser>();
    p->transform = Matrix::rows(m);
    p->material = g_reg.materials.at(material);
    p->casts_shadow = casts_shadow != 0;
    if (!p->parse_obj_file(path)) {
        g_error = std::string("Error: Could not read lines from: ") + path;
        return -1;
    }
    g_reg.parsers.push_back(p);
    return (int64_t)g_reg.parsers.size() - 1;
} catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); return -1; } }
int rth_obj_num_vertices(int64_t h) { try { return (int)g_reg.parsers.at(h)->vertices.size(); } catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); return -1; } }
int rth_obj_num_normals(int64_t h) { try { return (int)g_reg.parsers.at(h)->vertex_normals.size(); } catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); return -1; } }
void rth_obj_vertex(int64_t h, int i, double* o) { try { put4(g_reg.parsers.at(h)->vertices.at(i), o); } catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); } }
void rth_obj_normal(int64_t h, int i, double* o) { try { put4(g_reg.parsers.at(h)->vertex_normals.at(i), o); } catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); } }
int rth_obj_num_groups(int64_t h) { try { return (int)g_reg.parsers.at(h)->order.size(); } catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); return -1; } }
int64_t rth_obj_group(int64_t h, const char* name) { try {
    auto& p = g_reg.parsers.at(h);
    auto it = p->groups.find(name ? std::string(name) : p->current_group);
    if (it == p->groups.end()) return -1;
    return g_reg.add(it->second);
} catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); return -1; } }
int64_t rth_obj_to_group(int64_t h, const double* m) { try {
    return g_reg.add(g_reg.parsers.at(h)->obj_to_group(Matrix::rows(m)));
} catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); return -1; } }

// ---- world / camera ----
int64_t rth_world(void) { try {
    g_reg.worlds.push_back(std::make_shared<World>());
    return (int64_t)g_reg.worlds.size() - 1;
} catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); return -1; } }
void rth_world_add_object(int64_t w, int64_t s) { try {
    auto& world = g_reg.worlds.at(w);
    world->objects.push_back(g_reg.shapes.at(s));
    world->drop_scene();
    world->flattened = false;
} catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); } }
void rth_world_add_light(int64_t w, const double* pos, const double* col) { try {
    Light l;
    std::memcpy(l.position, pos, sizeof l.position);
    std::memcpy(l.intensity, col, sizeof l.intensity);
    auto& world = g_reg.worlds.at(w);
    world->lights.push_back(l);
    world->drop_scene();
    world->flattened = false;
} catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); } }
int64_t rth_camera(int64_t hsize, int64_t vsize, double fov, const double* m) { try {
    g_reg.cameras.push_back(std::make_shared<Camera>((size_t)hsize, (size_t)vsize, fov, Matrix::rows(m)));
    return (int64_t)g_reg.cameras.size() - 1;
} catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); return -1; } }
void rth_camera_info(int64_t c, double* o) { try {
    auto& cam = g_reg.cameras.at(c);
    o[0] = cam->half_width; o[1] = cam->half_height; o[2] = cam->pixel_size;
    o[3] = (double)cam->width(); o[4] = (double)cam->height();
} catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); } }
int rth_camera_block(int64_t c, rt_camera* out) { try { return camera_block(*g_reg.cameras.at(c), *out) ? 0 : RT_ERR_INVALID; } catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); return -1; } }

// ---- flatten (inspection) ----
// counts: nodes, xforms, aabbs, ranges, tris, materials, patterns, pattern_xforms, lights
int rth_flatten(int64_t w, int32_t* counts) { try {
    g_error.clear();
    auto& world = g_reg.worlds.at(w);
    if (!flatten(*world)) return RT_ERR_INVALID;
    rt_scene_desc d = describe(world->flat);
    int32_t c[9] = {d.n_nodes, d.n_xforms, d.n_aabbs, d.n_ranges, d.n_tris, d.n_materials, d.n_patterns,
                    d.n_pattern_xforms, d.n_lights};
    if (counts) std::memcpy(counts, c, sizeof c);
    return 0;
} catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); return -1; } }
int rth_flat_nodes(int64_t w, rt_node* out) { try {
    auto& world = g_reg.worlds.at(w);
    if (!flatten(*world)) return RT_ERR_INVALID;
    std::memcpy(out, world->flat.nodes.data(), world->flat.nodes.size() * sizeof(rt_node));
    return 0;
} catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); return -1; } }
int rth_flat_aabbs(int64_t w, rt_aabb* out) { try {
    auto& world = g_reg.worlds.at(w);
    if (!flatten(*world)) return RT_ERR_INVALID;
    std::memcpy(out, world->flat.aabbs.data(), world->flat.aabbs.size() * sizeof(rt_aabb));
    return 0;
} catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); return -1; } }
int rth_flat_xforms(int64_t w, rt_xform* out) { try {
    auto& world = g_reg.worlds.at(w);
    if (!flatten(*world)) return RT_ERR_INVALID;
    std::memcpy(out, world->flat.xforms.data(), world->flat.xforms.size() * sizeof(rt_xform));
    return 0;
} catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); return -1; } }
// The scene description itself (pointers stay valid until the world changes).
int rth_scene_desc(int64_t w, rt_scene_desc* out) { try {
    auto& world = g_reg.worlds.at(w);
    if (!flatten(*world)) return RT_ERR_INVALID;
    *out = describe(world->flat);
    return 0;
} catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); return -1; } }

// ---- render: the replacement of src/lib.rs:77-133 ----
static int ensure_scene(World& world, int device) {
    g_error.clear();
    if (world.scene && world.flattened && world.device == device) return RT_OK;
    world.drop_scene();  // the world changed, or it is wanted on another GPU
    if (!flatten(world)) return RT_ERR_INVALID;
    rt_scene_desc d = describe(world.flat);
    int rc = rt_scene_create(&d, device, &world.scene);
    if (rc == RT_OK) world.device = device;
    return rc;
}
// Uploads the world to `device` (idempotent) and returns the scene handle for direct ABI use.  The
// handle describes the world as it is now and stays valid until rth_reset (or the end of the
// process), also when the world is changed or rendered on another GPU afterwards.
int rth_world_scene(int64_t w, int device, rt_scene** out) { try {
    auto& world = g_reg.worlds.at(w);
    int rc = ensure_scene(*world, device);
    if (rc == RT_OK && out) {
        *out = world->scene;
        world->exported = true;
    }
    return rc;
} catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); return -1; } }
int rth_render(int64_t w, int64_t c, int device, const rt_params* params, double* rgb_out, rt_stats* stats) { try {
    auto& world = g_reg.worlds.at(w);
    int rc = ensure_scene(*world, device);
    if (rc != RT_OK) return rc;
    rt_camera cam;
    if (!camera_block(*g_reg.cameras.at(c), cam)) return RT_ERR_INVALID;
    return rt_render(world->scene, &cam, params, rgb_out, stats);
} catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); return -1; } }
int rth_render_device(int64_t w, int64_t c, int device, const rt_params* params, double* d_rgb, void* stream) { try {
    auto& world = g_reg.worlds.at(w);
    int rc = ensure_scene(*world, device);
    if (rc != RT_OK) return rc;
    rt_camera cam;
    if (!camera_block(*g_reg.cameras.at(c), cam)) return RT_ERR_INVALID;
    return rt_render_device(world->scene, &cam, params, d_rgb, stream);
} catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); return -1; } }
// The same with Canvas::clamp applied on the device: 3 bytes per pixel (canvas.rs:50-60).
int rth_render_u8(int64_t w, int64_t c, int device, const rt_params* params, uint8_t* rgb8_out, rt_stats* stats) { try {
    auto& world = g_reg.worlds.at(w);
    int rc = ensure_scene(*world, device);
    if (rc != RT_OK) return rc;
    rt_camera cam;
    if (!camera_block(*g_reg.cameras.at(c), cam)) return RT_ERR_INVALID;
    return rt_render_u8(world->scene, &cam, params, rgb8_out, stats);
} catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); return -1; } }
int rth_render_device_u8(int64_t w, int64_t c, int device, const rt_params* params, uint8_t* d_rgb8, void* stream) { try {
    auto& world = g_reg.worlds.at(w);
    int rc = ensure_scene(*world, device);
    if (rc != RT_OK) return rc;
    rt_camera cam;
    if (!camera_block(*g_reg.cameras.at(c), cam)) return RT_ERR_INVALID;
    return rt_render_device_u8(world->scene, &cam, params, d_rgb8, stream);
} catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); return -1; } }
int rth_color_at(int64_t w, int device, int64_t n, const double* origins, const double* directions, int remaining,
                 double* rgb_out, rt_stats* stats) { try {
    auto& world = g_reg.worlds.at(w);
    int rc = ensure_scene(*world, device);
    if (rc != RT_OK) return rc;
    return rt_color_at(world->scene, n, origins, directions, remaining, rgb_out, stats);
} catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); return -1; } }

// ---- canvas ----
int rth_clamp_u8(double v) { try { return clamp(v); } catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); return -1; } }
int64_t rth_to_ppm(const double* rgb, int64_t w, int64_t h) { try {
    g_ppm = to_ppm(rgb, (size_t)w, (size_t)h);
    return (int64_t)g_ppm.size();
} catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); return -1; } }
int64_t rth_to_ppm_u8(const uint8_t* rgb8, int64_t w, int64_t h) { try {
    g_ppm = to_ppm_u8(rgb8, (size_t)w, (size_t)h);
    return (int64_t)g_ppm.size();
} catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); return -1; } }
void rth_ppm_copy(char* dst) { try { std::memcpy(dst, g_ppm.data(), g_ppm.size()); } catch (const std::exception& e) { g_error = std::string("invalid handle or argument: ") + e.what(); } }

}  // extern "C"
