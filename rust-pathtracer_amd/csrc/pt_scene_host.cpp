// pt_scene_host.cpp — host side of pt_scene_create: validates a pt_scene_desc, builds the two BVH levels and
// packs everything into the word blob of pt_blob.h.  Runs once per scene, outside the timed render loop
// (the reference's equivalent work is construct_world + Mesh::init + Accelerator::new,
// src/parsing/mod.rs:145-563, src/geometry/mesh.rs:283-305, src/accelerator/mod.rs:31-43).
#include "pt_scene_host.h"
#include "pt_device.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace pth {

namespace {

struct Box { float mn[3], mx[3]; };

Box box_empty() { Box b; for (int i = 0; i < 3; ++i) { b.mn[i] = INFINITY; b.mx[i] = -INFINITY; } return b; }
void box_grow(Box& b, const float* p) { for (int i = 0; i < 3; ++i) { b.mn[i] = std::fmin(b.mn[i], p[i]); b.mx[i] = std::fmax(b.mx[i], p[i]); } }
void box_expand(Box& b, const Box& o) { for (int i = 0; i < 3; ++i) { b.mn[i] = std::fmin(b.mn[i], o.mn[i]); b.mx[i] = std::fmax(b.mx[i], o.mx[i]); } }
Box box_of_corners(const float* a, const float* b) {  // AABB::new, src/aabb.rs:16-21
    Box r; for (int i = 0; i < 3; ++i) { r.mn[i] = std::fmin(a[i], b[i]); r.mx[i] = std::fmax(a[i], b[i]); } return r;
}
void box_center(const Box& b, float* c) { for (int i = 0; i < 3; ++i) c[i] = b.mn[i] + (b.mx[i] - b.mn[i]) / 2.0f; }
float box_area(const Box& b) {  // src/aabb.rs:97-100
    float sx = b.mx[0] - b.mn[0], sy = b.mx[1] - b.mn[1], sz = b.mx[2] - b.mn[2];
    return 2.0f * (sx * sy + sx * sz + sy * sz);
}

uint32_t fbits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
float bits_f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

// Skip-link BVH in pre-order, SAH with 6 buckets on the widest centroid axis, median split when the centroids
// coincide, one shape per leaf: the reference's build (src/accelerator/bvh.rs:299-457) emitted directly in
// the flattened order of src/accelerator/lbvh.rs:47-163, with each leaf merged into its navigator node.
struct BvhBuilder {
    const std::vector<Box>& shapes;
    std::vector<uint32_t>& out;  // PT_NODE_WORDS per node
    uint32_t base;               // first node's index (0)
    std::vector<uint32_t> leaf_of_shape;  // node index of each shape's leaf
    std::vector<char> no_cull;            // per shape (empty = none): its computed hit distance may fall short of its box (spheres): nodes holding one are never culled
    std::vector<float> sphere_radius;     // per shape (empty = none): > 0 an untransformed sphere's radius, +inf a sphere that must never be culled (transformed), 0 no sphere
    std::vector<float> node_margin;       // out, per node (filled when sphere_radius is given): pt_blob.h PT_HDR_TOP_MARGIN
    explicit BvhBuilder(const std::vector<Box>& s, std::vector<uint32_t>& o) : shapes(s), out(o), base(0), leaf_of_shape(s.size(), 0) {}

    uint32_t node_count() const { return (uint32_t)(out.size() / PT_NODE_WORDS); }

    void split(const std::vector<uint32_t>& idx, std::vector<uint32_t>& li, Box& lb, std::vector<uint32_t>& ri, Box& rb) {
        Box bounds = box_empty(), cbounds = box_empty();
        for (uint32_t i : idx) { float c[3]; box_center(shapes[i], c); box_expand(bounds, shapes[i]); box_grow(cbounds, c); }
        float size[3] = {cbounds.mx[0] - cbounds.mn[0], cbounds.mx[1] - cbounds.mn[1], cbounds.mx[2] - cbounds.mn[2]};
        float widest = std::fmax(std::fmax(size[0], size[1]), std::fmax(size[2], 0.0f));
        int axis = 0;
        for (int a = 0; a < 3; ++a) if (size[a] >= widest) axis = a;  // largest lane index among ties (bvh.rs:348-353)
        float extent = (0.0f >= widest) ? 0.0f : cbounds.mx[axis] - cbounds.mn[axis];
        if (extent < 0.00001f) {
            size_t half = idx.size() / 2;
            li.assign(idx.begin(), idx.begin() + half); ri.assign(idx.begin() + half, idx.end());
            lb = box_empty(); for (uint32_t i : li) box_expand(lb, shapes[i]);
            rb = box_empty(); for (uint32_t i : ri) box_expand(rb, shapes[i]);
            return;
        }
        const int NB = 6;
        size_t count[NB] = {0, 0, 0, 0, 0, 0}; Box bb[NB]; std::vector<uint32_t> members[NB];
        for (int b = 0; b < NB; ++b) bb[b] = box_empty();
        for (uint32_t i : idx) {
            float c[3]; box_center(shapes[i], c);
            float rel = (c[axis] - cbounds.mn[axis]) / extent;
            float fb = rel * ((float)NB - 0.01f);
            int b = fb >= 0.0f ? (int)fb : 0; if (b > NB - 1) b = NB - 1;
            count[b]++; box_expand(bb[b], shapes[i]); members[b].push_back(i);
        }
        int best = 0; float best_cost = INFINITY; lb = box_empty(); rb = box_empty();
        for (int k = 0; k < NB - 1; ++k) {
            size_t nl = 0, nr = 0; Box l = box_empty(), r = box_empty();
            for (int j = 0; j <= k; ++j) { nl += count[j]; box_expand(l, bb[j]); }
            for (int j = k + 1; j < NB; ++j) { nr += count[j]; box_expand(r, bb[j]); }
            float cost = ((float)nl * box_area(l) + (float)nr * box_area(r)) / box_area(bounds);
            if (cost < best_cost) { best = k; best_cost = cost; lb = l; rb = r; }
        }
        li.clear(); ri.clear();
        for (int j = 0; j <= best; ++j) li.insert(li.end(), members[j].begin(), members[j].end());
        for (int j = best + 1; j < NB; ++j) ri.insert(ri.end(), members[j].begin(), members[j].end());
    }

    void emit(const std::vector<uint32_t>& idx, const Box& box) {
        size_t at = out.size();
        out.resize(at + PT_NODE_WORDS, 0);
        bool leaf = idx.size() == 1;
        const Box& b = leaf ? shapes[idx[0]] : box;
        if (!leaf) { std::vector<uint32_t> li, ri; Box lb, rb; split(idx, li, lb, ri, rb); emit(li, lb); emit(ri, rb); }
        // [3]: the node to continue with when this subtree is done or skipped; bit 31 (PT_NODE_FLAT) marks a box of zero thickness,
        // which the filtered slab test must take axis by axis (aabb_classify)
        const bool flat = b.mn[0] == b.mx[0] || b.mn[1] == b.mx[1] || b.mn[2] == b.mx[2];
        bool keep = false;
        if (!no_cull.empty()) for (uint32_t i : idx) keep = keep || no_cull[i] != 0;
        const int fx = b.mn[0] == b.mx[0], fy = b.mn[1] == b.mx[1], fz = b.mn[2] == b.mx[2];
        const uint32_t code = fx + fy + fz == 0 ? 0u : (fx + fy + fz > 1 ? 4u : (fx ? 1u : (fy ? 2u : 3u)));   // aabb_classify_by (pt_device.h)
        out[at + 0] = fbits(b.mn[0]); out[at + 1] = fbits(b.mn[1]); out[at + 2] = fbits(b.mn[2]); out[at + 3] = node_count() | (flat ? PT_NODE_FLAT : 0u) | (keep ? PT_NODE_NO_CULL : 0u) | code << 27;
        out[at + 4] = fbits(b.mx[0]); out[at + 5] = fbits(b.mx[1]); out[at + 6] = fbits(b.mx[2]); out[at + 7] = leaf ? idx[0] : PT_NODE_INNER;
        if (leaf) leaf_of_shape[idx[0]] = (uint32_t)(at / PT_NODE_WORDS);
        if (!sphere_radius.empty()) {
            float rmax = 0.0f;
            for (uint32_t i : idx) rmax = std::fmax(rmax, sphere_radius[i]);
            const double dx = (double)b.mx[0] - b.mn[0], dy = (double)b.mx[1] - b.mn[1], dz = (double)b.mx[2] - b.mn[2];
            const double m0 = 3.5 * rmax + PT_SPHERE_CULL_K * (std::sqrt(dx * dx + dy * dy + dz * dz) + rmax);
            if (node_margin.size() < at / PT_NODE_WORDS + 1) node_margin.resize(at / PT_NODE_WORDS + 1, 0.0f);
            node_margin[at / PT_NODE_WORDS] = rmax > 0.0f ? std::nextafterf((float)m0, INFINITY) : 0.0f;   // (inf stays inf)
        }
    }

    void build() {
        if (shapes.empty()) return;
        std::vector<uint32_t> idx(shapes.size());
        for (size_t i = 0; i < idx.size(); ++i) idx[i] = (uint32_t)i;
        if (idx.size() == 1) { emit(idx, shapes[0]); return; }
        std::vector<uint32_t> li, ri; Box lb, rb;
        split(idx, li, lb, ri, rb);  // the root itself has no node (lbvh.rs:134-140)
        emit(li, lb); emit(ri, rb);
    }
};


// ---- a closed mesh's inner sphere (pt_blob.h PT_MESH_INNER_*), all in f64 -------------------------------------------------------------
// Closed = every undirected edge (vertices compared by POSITION: OBJ files repeat vertices along seams) belongs to exactly two triangles.  The centre is the point
// of a 9 x 9 x 9 grid over the bounding box that is farthest from the surface among those at odd crossing parity along three axis rays (all three must agree);
// the radius 0.98 x its distance to the nearest triangle.  Any failure — an open edge, a degenerate triangle, rays that disagree, a ball smaller than 2 % of the
// box — and the mesh simply has no ball.
struct D3 { double x, y, z; };
static inline D3 d3(const float* p) { return D3{p[0], p[1], p[2]}; }
static inline D3 dsub(D3 a, D3 b) { return D3{a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline double ddot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline D3 dcross(D3 a, D3 b) { return D3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
// squared distance from p to triangle abc (Ericson, Real-Time Collision Detection 5.1.5)
static double point_triangle_dist2(D3 p, D3 a, D3 b, D3 c) {
    const D3 ab = dsub(b, a), ac = dsub(c, a), ap = dsub(p, a);
    const double d1 = ddot(ab, ap), d2 = ddot(ac, ap);
    auto len2 = [](D3 v) { return ddot(v, v); };
    if (d1 <= 0 && d2 <= 0) return len2(ap);
    const D3 bp = dsub(p, b); const double d3_ = ddot(ab, bp), d4 = ddot(ac, bp);
    if (d3_ >= 0 && d4 <= d3_) return len2(bp);
    const double vc = d1 * d4 - d3_ * d2;
    if (vc <= 0 && d1 >= 0 && d3_ <= 0) { const double v = d1 / (d1 - d3_); return len2(dsub(ap, D3{ab.x * v, ab.y * v, ab.z * v})); }
    const D3 cp = dsub(p, c); const double d5 = ddot(ab, cp), d6 = ddot(ac, cp);
    if (d6 >= 0 && d5 <= d6) return len2(cp);
    const double vb = d5 * d2 - d1 * d6;
    if (vb <= 0 && d2 >= 0 && d6 <= 0) { const double w = d2 / (d2 - d6); return len2(dsub(ap, D3{ac.x * w, ac.y * w, ac.z * w})); }
    const double va = d3_ * d6 - d5 * d4;
    if (va <= 0 && (d4 - d3_) >= 0 && (d5 - d6) >= 0) { const double w = (d4 - d3_) / ((d4 - d3_) + (d5 - d6)); const D3 bc = dsub(c, b); return len2(dsub(bp, D3{bc.x * w, bc.y * w, bc.z * w})); }
    const double denom = 1.0 / (va + vb + vc), v = vb * denom, w = vc * denom;
    return len2(dsub(ap, D3{ab.x * v + ac.x * w, ab.y * v + ac.y * w, ab.z * v + ac.z * w}));
}
// crossings of the ray p + t e_axis, t > 0, with the triangles; -1 if the ray passes within 1e-9 (relative) of an edge or vertex (the caller drops the point)
static int axis_crossings(D3 p, int axis, const float* V, const uint32_t* ix, uint32_t faces, double scale) {
    const int u = (axis + 1) % 3, v = (axis + 2) % 3;
    auto comp = [](D3 q, int k) { return k == 0 ? q.x : (k == 1 ? q.y : q.z); };
    int n = 0;
    for (uint32_t f = 0; f < faces; ++f) {
        const D3 a = d3(V + 3 * ix[3 * f]), b = d3(V + 3 * ix[3 * f + 1]), c = d3(V + 3 * ix[3 * f + 2]);
        const double pu = comp(p, u), pv = comp(p, v);
        const double au = comp(a, u) - pu, av = comp(a, v) - pv, bu_ = comp(b, u) - pu, bv = comp(b, v) - pv, cu = comp(c, u) - pu, cv = comp(c, v) - pv;
        const double e0 = bu_ * cv - bv * cu, e1 = cu * av - cv * au, e2 = au * bv - av * bu_;   // 2D edge functions of the projection
        const double eps = 1e-9 * scale * scale;
        const int pos = (e0 > eps) + (e1 > eps) + (e2 > eps), neg = (e0 < -eps) + (e1 < -eps) + (e2 < -eps);
        if (pos == 3 || neg == 3) {                       // strictly inside the projected triangle: the ray's line crosses the triangle's plane there
            const double det = e0 + e1 + e2;
            const double t = (e0 * (comp(a, axis) - comp(p, axis)) + e1 * (comp(b, axis) - comp(p, axis)) + e2 * (comp(c, axis) - comp(p, axis))) / det;
            if (std::fabs(t) < 1e-9 * scale) return -1;
            if (t > 0) ++n;
        } else if (!(pos > 0 && neg > 0)) return -1;      // on, or too close to, an edge, a vertex, or the line of a triangle seen edge-on: this point is not used
    }
    return n;
}
// Closed = every undirected edge (end points compared by the BIT PATTERNS of their positions) belongs to exactly two triangles, and no triangle is degenerate:
// not an edge of length 0 and — round-5 advisor — not a sliver either.  A T-junction filled with a zero-area triangle (three distinct collinear vertices) still pairs
// every edge twice, but the watertight triangle test is watertight only across edges computed from the SAME vertex pair (mesh.rs:67-198): through such a junction a ray
// can meet the det == 0 triangle alone and pass.  "Closed" is what the shortcuts below argue from, so such a mesh is not closed.
static bool mesh_is_closed(const float* V, uint32_t vertex_count, const uint32_t* ix, uint32_t faces, const Box& mb) {
    struct Key { uint32_t a[3], b[3]; bool operator<(const Key& o) const { return std::memcmp(this, &o, sizeof(Key)) < 0; } };
    auto pos = [&](uint32_t v, uint32_t* out) { for (int k = 0; k < 3; ++k) { float x = V[3 * v + k]; if (x == 0.0f) x = 0.0f; std::memcpy(&out[k], &x, 4); } };   // (-0 = +0)
    const double ex = (double)mb.mx[0] - mb.mn[0], ey = (double)mb.mx[1] - mb.mn[1], ez = (double)mb.mx[2] - mb.mn[2];
    const double extent = std::fmax(ex, std::fmax(ey, ez));
    std::vector<Key> edges; edges.reserve((size_t)faces * 3);
    for (uint32_t f = 0; f < faces; ++f) {
        for (int k = 0; k < 3; ++k) if (ix[3 * f + k] >= vertex_count) return false;
        const D3 a = d3(V + 3 * ix[3 * f]), b = d3(V + 3 * ix[3 * f + 1]), c = d3(V + 3 * ix[3 * f + 2]);
        const D3 n = dcross(dsub(b, a), dsub(c, a));
        if (!(std::sqrt(ddot(n, n)) > 1e-10 * extent * extent)) return false;     // twice the area: a sliver (or NaN)
        for (int k = 0; k < 3; ++k) {
            const uint32_t v0 = ix[3 * f + k], v1 = ix[3 * f + (k + 1) % 3];
            Key e; pos(v0, e.a); pos(v1, e.b);
            if (std::memcmp(e.a, e.b, 12) == 0) return false;                    // a degenerate edge
            if (std::memcmp(e.a, e.b, 12) > 0) { uint32_t t[3]; std::memcpy(t, e.a, 12); std::memcpy(e.a, e.b, 12); std::memcpy(e.b, t, 12); }
            edges.push_back(e);
        }
    }
    std::sort(edges.begin(), edges.end());
    for (size_t i = 0; i < edges.size();) {
        size_t j = i; while (j < edges.size() && std::memcmp(&edges[j], &edges[i], sizeof(Key)) == 0) ++j;
        if (j - i != 2) return false;
        i = j;
    }
    return true;
}

// ---- a CONVEX closed mesh instance (pt_blob.h PT_INST_CONVEX_*), certified in f64 in WORLD space ---------------------------------------------------------------------
// What stage_shade argues from when it marks a light-sample ray that leaves such an instance (outward: the instance cannot be hit again; inward: nothing but this
// instance's own surface — or something inside it — can be the closest hit, and no light is inside).  Every number the argument uses is checked here, for the vertices
// as the engine sees them (f32 positions through the instance's f32 forward matrix) and the normals as hit_record makes them (vertex normals, or the face normal, through
// the transposed reverse matrix):
//   closed (above); at most 4096 faces; world coordinates within 64 (one f32 ulp there is 4e-6: a hundred times below the margins);
//   every face's hit normals lie on its outward side; per face: the largest angle delta between the face's own normal and its hit normals (0 for a flat-shaded face), faces
//   with cos(delta) < 0.5 taking no claim;
//   CONVEX: no vertex lies more than 2e-4 above any face's plane (the gem's facets are planar to 1e-4) and some vertex lies 1e-2 below it;
//   OUT, face by face: the threshold 0.02 + sin(delta) a direction's cosine to the hit normal must exceed (PT_TRI_OUT_SHIFT);
//   IN, face by face: the face's three corners, moved 1e-3 against the face normal (where pt.rs:176 puts an inward ray's origin), lie at least 1e-4 below EVERY face plane —
//   so does every point of the face (the planes' half-spaces are convex); such a face's triangle carries PT_TRI_IN_SAFE and hit_record hands it on in the hit's instance word;
//   and the world box of every light-tagged instance is disjoint from this instance's (both widened by 1e-3).
static uint32_t convex_certificate(const float* V, uint32_t vertex_count, const uint32_t* ix, uint32_t faces, const float* N, const pt_instance& in, const Box& wbox,
                                   const std::vector<Box>& light_boxes, std::vector<uint32_t>* face_flags) {
    if (faces < 4 || faces > 4096 || vertex_count > 3 * 4096) return 0u;
    auto world = [&](const float* p) {
        if (!in.has_transform) return D3{p[0], p[1], p[2]};
        const float* m = in.forward;
        return D3{(double)m[0] * p[0] + (double)m[1] * p[1] + (double)m[2] * p[2] + m[3], (double)m[4] * p[0] + (double)m[5] * p[1] + (double)m[6] * p[2] + m[7],
                  (double)m[8] * p[0] + (double)m[9] * p[1] + (double)m[10] * p[2] + m[11]};
    };
    auto world_normal = [&](D3 n) {   // hit_record: normalize(reverse^T n)
        if (in.has_transform) { const float* r = in.reverse; n = D3{r[0] * n.x + r[4] * n.y + r[8] * n.z, r[1] * n.x + r[5] * n.y + r[9] * n.z, r[2] * n.x + r[6] * n.y + r[10] * n.z}; }
        const double l = std::sqrt(ddot(n, n));
        return D3{n.x / l, n.y / l, n.z / l};
    };
    std::vector<D3> W(vertex_count);
    for (uint32_t v = 0; v < vertex_count; ++v) {
        W[v] = world(V + 3 * v);
        if (!(std::fabs(W[v].x) <= 64.0 && std::fabs(W[v].y) <= 64.0 && std::fabs(W[v].z) <= 64.0)) return 0u;
    }
    const double kSlack = 2e-4, kThick = 1e-2, kOffset = 1e-3, kInside = 1e-4, kMinCos = 0.5;
    std::vector<D3> fn(faces); std::vector<double> fd(faces), fsin(faces), fcos(faces);   // per face: the plane, and the sine / cosine of the largest angle between its normal and its hit normals
    for (uint32_t f = 0; f < faces; ++f) {
        const D3 a = W[ix[3 * f]], b = W[ix[3 * f + 1]], c = W[ix[3 * f + 2]];
        D3 g = dcross(dsub(b, a), dsub(c, a));
        const double l = std::sqrt(ddot(g, g));
        if (!(l > 0)) return 0u;
        g = D3{g.x / l, g.y / l, g.z / l};
        // the hit normals of this face, as hit_record makes them (local: the vertex normals, or cross(p0 - p2, p1 - p2))
        D3 local[3]; int count = 0;
        if (N != nullptr) for (int k = 0; k < 3; ++k) local[count++] = d3(N + 3 * ix[3 * f + k]);
        else { const D3 p0 = d3(V + 3 * ix[3 * f]), p1 = d3(V + 3 * ix[3 * f + 1]), p2 = d3(V + 3 * ix[3 * f + 2]); local[count++] = dcross(dsub(p0, p2), dsub(p1, p2)); }
        const D3 first = world_normal(local[0]);
        if (ddot(first, g) < 0) g = D3{-g.x, -g.y, -g.z};   // (the face's plane normal on the side the hit normals point to: it must turn out to be the OUTWARD side, below)
        // a hit's normal is normalize(sum b_i n_i), b_i >= 0: no farther from the face's normal than the farthest n_i (a flat-shaded face: the face's own, rounding apart)
        double cmin = 1.0;
        for (int k = 0; k < count; ++k) cmin = std::fmin(cmin, ddot(world_normal(local[k]), g));
        if (!(cmin > 0.0)) return 0u;                        // a hit normal on the other side of its own face
        fn[f] = g; fd[f] = ddot(g, a); fcos[f] = cmin; fsin[f] = std::sqrt(std::fmax(0.0, 1.0 - cmin * cmin));
    }
    for (uint32_t f = 0; f < faces; ++f) {
        double hi = -INFINITY, lo = INFINITY;
        for (uint32_t v = 0; v < vertex_count; ++v) { const double sd = ddot(fn[f], W[v]) - fd[f]; hi = std::fmax(hi, sd); lo = std::fmin(lo, sd); }
        if (!(hi <= kSlack && lo <= -kThick)) return 0u;   // not convex (or the normals point inward, or the body is a sliver)
    }
    // OUT, face by face: the origin p + 1e-3 n (n a hit normal) lies 1e-3 cos(delta) above the face's plane — at least 5e-4: above the slack — and a direction with
    // n . d > sin(delta) has angle(d, face normal) <= angle(d, n) + delta < 90 degrees: it moves away from the plane.  The threshold rides in the triangle's flag word in
    // units of 2^-15, rounded up, + 0.02 (rounding of n . d in f32, of a normal's length: 1e-6).
    face_flags->assign(faces, 0u);
    bool any_out = false;
    for (uint32_t f = 0; f < faces; ++f) {
        if (!(fcos[f] >= kMinCos)) continue;
        const uint32_t tq = (uint32_t)std::ceil((fsin[f] + 0.02) * 32768.0);
        if (tq == 0u || tq > 32767u) continue;
        (*face_flags)[f] = tq << PT_TRI_OUT_SHIFT;
        any_out = true;
    }
    if (!any_out) return 0u;
    uint32_t flags = PT_INST_CONVEX_OUT;
    // IN is a property of a FACE (a sharp edge — the brilliant cut has 64 faces at edges of 97 degrees — puts the corner of one face, moved inward, OUTSIDE the next
    // face's plane): `in_safe[f]`, and the instance takes the flag when any of its faces is safe
    // (the origin is p - 1e-3 n with n within delta of the face's normal: it lies within 1e-3 * 2 sin(delta / 2) of p - 1e-3 n_f, so that point must clear every plane by that much more)
    bool inside_ok = false;
    std::vector<char> in_safe(faces, 0);   // (0 none, 1 the whole face, 2 its inside only)
    for (uint32_t f = 0; f < faces; ++f) {
        if (!(fcos[f] >= kMinCos)) continue;
        const double need = kInside + kOffset * std::sqrt(std::fmax(0.0, 2.0 - 2.0 * fcos[f]));   // 2 sin(delta / 2) = sqrt(2 - 2 cos delta)
        bool ok = true;
        for (int k = 0; k < 3 && ok; ++k) {
            const D3 p = W[ix[3 * f + k]], q = D3{p.x - kOffset * fn[f].x, p.y - kOffset * fn[f].y, p.z - kOffset * fn[f].z};
            for (uint32_t g = 0; g < faces; ++g) if (!(ddot(fn[g], q) - fd[g] <= -need)) { ok = false; break; }
        }
        if (!ok) {   // the face without a strip of PT_TRI_INNER_BARY along its edges (hit_record tests the hit's barycentric coordinates): the corners of that inner triangle
            ok = true;
            const double e = PT_TRI_INNER_BARY * 0.98;   // (the device compares f32 barycentrics: a margin for their rounding)
            for (int k = 0; k < 3 && ok; ++k) {
                const D3 a = W[ix[3 * f + k]], b = W[ix[3 * f + (k + 1) % 3]], c = W[ix[3 * f + (k + 2) % 3]];
                const D3 p = D3{(1 - 2 * e) * a.x + e * b.x + e * c.x, (1 - 2 * e) * a.y + e * b.y + e * c.y, (1 - 2 * e) * a.z + e * b.z + e * c.z};
                const D3 q = D3{p.x - kOffset * fn[f].x, p.y - kOffset * fn[f].y, p.z - kOffset * fn[f].z};
                for (uint32_t g = 0; g < faces; ++g) if (!(ddot(fn[g], q) - fd[g] <= -need)) { ok = false; break; }
            }
            in_safe[f] = ok ? 2 : 0;
        } else in_safe[f] = 1;
        inside_ok = inside_ok || ok;
    }
    bool lights_clear = true;
    for (const Box& lb : light_boxes) {
        bool apart = false;
        for (int k = 0; k < 3; ++k) apart = apart || (double)lb.mn[k] - 1e-3 > (double)wbox.mx[k] + 1e-3 || (double)lb.mx[k] + 1e-3 < (double)wbox.mn[k] - 1e-3;
        lights_clear = lights_clear && apart;
    }
    if (inside_ok && lights_clear) {
        flags |= PT_INST_CONVEX_IN;
        for (uint32_t f = 0; f < faces; ++f) (*face_flags)[f] |= in_safe[f] == 1 ? PT_TRI_IN_SAFE : (in_safe[f] == 2 ? PT_TRI_IN_SAFE_INNER : 0u);
    }
    return flags;
}

static bool inner_sphere(const float* V, uint32_t vertex_count, const uint32_t* ix, uint32_t faces, const Box& mb, float* centre, float* radius, std::vector<float>* more = nullptr) {
    if (!mesh_is_closed(V, vertex_count, ix, faces, mb)) return false;
    const double sx = (double)mb.mx[0] - mb.mn[0], sy = (double)mb.mx[1] - mb.mn[1], sz = (double)mb.mx[2] - mb.mn[2];
    const double scale = std::fmax(sx, std::fmax(sy, sz));
    if (!(scale > 0) || !(sx > 0 && sy > 0 && sz > 0)) return false;
    const int G = 9;
    double best_r2 = 0; D3 best{0, 0, 0};
    struct Cand { D3 p; double r2; };
    std::vector<Cand> inside_points;   // (every inside grid point with its distance: the candidates of the further balls, below)
    const bool want_more = more != nullptr && faces <= 20000;
    for (int gx = 0; gx < G; ++gx) for (int gy = 0; gy < G; ++gy) for (int gz = 0; gz < G; ++gz) {
        // (grid points at irrational-ish offsets, so that axis rays do not run along the seams of symmetric models)
        const D3 p{mb.mn[0] + sx * (gx + 0.5137) / G, mb.mn[1] + sy * (gy + 0.4871) / G, mb.mn[2] + sz * (gz + 0.5063) / G};
        double r2 = INFINITY;
        const double stop_below = want_more ? 0.0 : best_r2;
        for (uint32_t f = 0; f < faces && r2 > stop_below; ++f)
            r2 = std::fmin(r2, point_triangle_dist2(p, d3(V + 3 * ix[3 * f]), d3(V + 3 * ix[3 * f + 1]), d3(V + 3 * ix[3 * f + 2])));
        if (!(r2 > (want_more ? 1e-6 * scale * scale : best_r2))) continue;
        bool inside = true;
        for (int axis = 0; axis < 3 && inside; ++axis) { const int n = axis_crossings(p, axis, V, ix, faces, scale); inside = n > 0 && (n & 1) == 1; }
        if (!inside) continue;
        if (want_more) inside_points.push_back(Cand{p, r2});
        if (r2 > best_r2) { best_r2 = r2; best = p; }
    }
    if (!(best_r2 > 0)) return false;
    // refine the centre: a pattern search from the best grid point (the distance field has no other structure to use), steps from a grid cell down to 1e-4 of the box.
    // A step is taken only to a point that is farther from the surface; it cannot leave the inside: the segment to it stays within the old ball.
    auto dist2 = [&](D3 p, double stop_below) {
        double r2 = INFINITY;
        for (uint32_t f = 0; f < faces && r2 > stop_below; ++f) r2 = std::fmin(r2, point_triangle_dist2(p, d3(V + 3 * ix[3 * f]), d3(V + 3 * ix[3 * f + 1]), d3(V + 3 * ix[3 * f + 2])));
        return r2;
    };
    for (double step = scale / G; step > 1e-4 * scale; step *= 0.5) {
        for (int tries = 0; tries < 64; ++tries) {
            bool moved = false;
            for (int k = 0; k < 6; ++k) {
                const double s1 = (k & 1) ? -step : step;
                if (!(step < std::sqrt(best_r2))) break;      // (the new centre must lie inside the present ball)
                D3 q = best; if (k / 2 == 0) q.x += s1; else if (k / 2 == 1) q.y += s1; else q.z += s1;
                const double r2 = dist2(q, best_r2);
                if (r2 > best_r2) { best_r2 = r2; best = q; moved = true; }
            }
            if (!moved) break;
        }
    }
    const double r = 0.98 * std::sqrt(best_r2);
    if (!(r > 0.02 * scale)) return false;
    centre[0] = (float)best.x; centre[1] = (float)best.y; centre[2] = (float)best.z; *radius = (float)(r * 0.9999);
    // Further balls (at most PT_MESH_MORE_BALLS): greedily the inside grid point farthest from the surface whose CENTRE lies outside every ball taken so far — a
    // ball somewhere else in the body — as long as it is at least a quarter of the first one's radius.  (What they buy was priced on C3's rays before it was built:
    // the first ball catches 24 % of the blocked rays that enter the gem's box, eight balls 58 %.)
    if (want_more) {
        std::vector<D3> centres{best}; std::vector<double> radii{r};
        while (centres.size() < 1 + PT_MESH_MORE_BALLS) {
            const Cand* pick = nullptr;
            for (const Cand& c : inside_points) {
                bool outside_all = true;
                for (size_t k = 0; k < centres.size() && outside_all; ++k) { const D3 dd = dsub(c.p, centres[k]); outside_all = ddot(dd, dd) > radii[k] * radii[k]; }
                if (outside_all && (pick == nullptr || c.r2 > pick->r2)) pick = &c;
            }
            if (pick == nullptr || !(0.98 * std::sqrt(pick->r2) >= PT_MESH_MORE_MIN * r)) break;
            const double rr = 0.98 * std::sqrt(pick->r2);
            centres.push_back(pick->p); radii.push_back(rr);
            more->push_back((float)pick->p.x); more->push_back((float)pick->p.y); more->push_back((float)pick->p.z); more->push_back((float)(rr * 0.9999));
        }
    }
    return true;
}

void pad16(std::vector<uint32_t>& w) { while (w.size() % 4) w.push_back(0); }

void xf_point(const float* m, const float* p, float* o) {
    for (int r = 0; r < 3; ++r) o[r] = m[4 * r] * p[0] + m[4 * r + 1] * p[1] + m[4 * r + 2] * p[2] + m[4 * r + 3];
}

}  // namespace

static bool build_host_scene_with(const pt_scene_desc& d, HostScene* hs, std::string* err, bool curve_tables, uint32_t* curve_table_words);
// The curves' cell tables are an accelerator, and a scene that fits the LDS whole without them and not with them is better off without (the kernels of a scene staged
// whole never touch global memory for scene data: G2F at 24.3 KB): built again without the tables then.
bool build_host_scene(const pt_scene_desc& d, HostScene* hs, std::string* err) {
    uint32_t table_words = 0;
    if (!build_host_scene_with(d, hs, err, true, &table_words)) return false;
    const size_t bytes = hs->blob.size() * 4;
    if (table_words != 0 && bytes > PT_BLOB_LDS_ALL_BYTES && bytes - 4u * table_words <= PT_BLOB_LDS_ALL_BYTES) {
        *hs = HostScene();
        return build_host_scene_with(d, hs, err, false, &table_words);
    }
    return true;
}
static bool build_host_scene_with(const pt_scene_desc& d, HostScene* hs, std::string* err, bool curve_tables, uint32_t* curve_table_words) {
    auto fail = [&](const char* m) { *err = m; return false; };
    *curve_table_words = 0;
    if (d.material_count == 0 || !d.materials) return fail("scene needs at least the error material (index 0)");
    if (d.camera_count == 0 || !d.cameras) return fail("scene has no camera");
    if (d.environment.kind != PT_ENV_CONSTANT && d.environment.kind != PT_ENV_SUN && d.environment.kind != PT_ENV_HDR) return fail("unknown environment kind");
    if (d.environment.kind == PT_ENV_HDR) {
        if (d.environment.texstack < 0 || (uint32_t)d.environment.texstack >= d.texstack_count) return fail("environment texstack out of range");
        if (d.environment.importance_width < 0 || d.environment.importance_height < 0 || d.environment.importance_width > 16384 || d.environment.importance_height > 16384) return fail("bad importance map size");
        if (d.environment.importance_luminance_curve >= (int32_t)d.curve_count) return fail("importance luminance curve out of range");
    } else if (d.environment.curve < 0 || (uint32_t)d.environment.curve >= d.curve_count) return fail("environment curve index out of range");
    auto curve_ok = [&](int32_t c) { return c >= 0 && (uint32_t)c < d.curve_count; };
    for (uint32_t i = 0; i < d.curve_count; ++i) {
        const pt_curve& c = d.curves[i];
        uint32_t per = c.kind == PT_CURVE_TABULATED ? 2 : ((c.kind == PT_CURVE_EXPONENTIAL || c.kind == PT_CURVE_INV_EXPONENTIAL) ? 4 : 1);
        bool needs_data = c.kind == PT_CURVE_LINEAR || c.kind == PT_CURVE_TABULATED || c.kind == PT_CURVE_EXPONENTIAL || c.kind == PT_CURVE_INV_EXPONENTIAL;
        if (c.kind < 0 || c.kind > PT_CURVE_CONST) return fail("unknown curve kind");
        if (needs_data && ((size_t)c.data_offset + (size_t)c.data_count * per > d.curve_data_count)) return fail("curve data out of range");
        if ((c.kind == PT_CURVE_LINEAR || c.kind == PT_CURVE_TABULATED) && c.data_count == 0) return fail("empty curve table");
    }
    for (uint32_t i = 0; i < d.material_count; ++i) {
        const pt_material& m = d.materials[i];
        switch (m.kind) {
            case PT_MATERIAL_LAMBERTIAN: if (m.texstack < 0 || (uint32_t)m.texstack >= d.texstack_count) return fail("lambertian texstack out of range"); break;
            case PT_MATERIAL_GGX: if (!curve_ok(m.curve_eta) || !curve_ok(m.curve_eta_o) || !curve_ok(m.curve_kappa)) return fail("ggx curve out of range");
                if (!(m.alpha > 0.0f)) return fail("ggx alpha must be positive"); break;
            case PT_MATERIAL_DIFFUSE_LIGHT: case PT_MATERIAL_SHARP_LIGHT: if (!curve_ok(m.curve_emit) || !curve_ok(m.curve_bounce)) return fail("light curve out of range"); break;
            case PT_MATERIAL_PASSTHROUGH: if (!curve_ok(m.curve_bounce)) return fail("passthrough colour curve out of range"); break;
            default: return fail("unknown material kind");
        }
        if (m.outer_medium < 0 || m.inner_medium < 0 || (uint32_t)m.outer_medium > d.medium_count || (uint32_t)m.inner_medium > d.medium_count) return fail("material medium id out of range");
    }
    if (d.medium_count > 255) return fail("more than 255 mediums");   // MediumId is a u8 (src/prelude.rs)
    if (d.medium_count && !d.mediums) return fail("mediums missing");
    for (uint32_t i = 0; i < d.medium_count; ++i) {
        const pt_medium& m = d.mediums[i];
        if (m.kind == PT_MEDIUM_HG) { if (!curve_ok(m.curve_g) || !curve_ok(m.curve_sigma_a) || !curve_ok(m.curve_sigma_s)) return fail("HG medium curve out of range"); }
        else if (m.kind == PT_MEDIUM_RAYLEIGH) { if (!curve_ok(m.curve_ior)) return fail("Rayleigh medium curve out of range"); }
        else return fail("unknown medium kind");
    }
    for (uint32_t i = 0; i < d.texstack_count; ++i) {
        const pt_texstack& t = d.texstacks[i];
        if (t.first_layer < 0 || t.layer_count < 0 || (uint32_t)(t.first_layer + t.layer_count) > d.layer_count) return fail("texstack layers out of range");
        for (int l = 0; l < t.layer_count; ++l) {
            const pt_texture_layer& L = d.layers[t.first_layer + l];
            if (L.kind != PT_TEXTURE1 && L.kind != PT_TEXTURE4) return fail("unknown texture layer kind");
            if (L.width <= 0 || L.height <= 0) return fail("empty texture");
            size_t n = (size_t)L.width * L.height * (L.kind == PT_TEXTURE4 ? 4 : 1);
            if (L.data_offset + n > d.texture_data_count) return fail("texture data out of range");
            for (int k = 0; k < (L.kind == PT_TEXTURE4 ? 4 : 1); ++k) if (!curve_ok(L.curves[k])) return fail("texture curve out of range");
            if (L.data_offset + n > 0xffffffffull) return fail("texture data too large");
        }
    }
    auto material_ok = [&](uint32_t id) { return id == PT_MATERIAL_NONE || PT_MATERIAL_INDEX(id) < d.material_count; };

    std::vector<uint32_t>& w = hs->blob;
    w.assign(PT_HDR_WORDS, 0);
    hs->tex.assign(d.texture_data, d.texture_data + d.texture_data_count);
    if (hs->tex.empty()) hs->tex.push_back(0.0f);
    hs->cameras.assign(d.cameras, d.cameras + d.camera_count);

    // curves
    pad16(w);
    std::vector<uint32_t> curve_off(d.curve_count);
    for (uint32_t i = 0; i < d.curve_count; ++i) { curve_off[i] = (uint32_t)w.size(); w.resize(w.size() + PT_CURVE_WORDS, 0); }
    for (uint32_t i = 0; i < d.curve_count; ++i) {
        const pt_curve& c = d.curves[i];
        uint32_t per = c.kind == PT_CURVE_TABULATED ? 2 : ((c.kind == PT_CURVE_EXPONENTIAL || c.kind == PT_CURVE_INV_EXPONENTIAL) ? 4 : 1);
        uint32_t off = (uint32_t)w.size();
        bool has = c.kind == PT_CURVE_LINEAR || c.kind == PT_CURVE_TABULATED || c.kind == PT_CURVE_EXPONENTIAL || c.kind == PT_CURVE_INV_EXPONENTIAL;
        if (has) for (uint32_t k = 0; k < c.data_count * per; ++k) w.push_back(fbits(d.curve_data[c.data_offset + k]));
        // A tabulated curve's cell table (round 5; pt_blob.h PT_CURVE_GRID): the knot range in G equal cells, per cell the number of knots in lower cells — a lower
        // bound of the binary search's answer for every wavelength of the cell (cell() is monotone), from which curve_eval walks up: the same index, one or two knot
        // reads instead of log2 n.  cell() here is curve_grid_cell's arithmetic (pt_device.h), operation for operation.
        uint32_t grid_word = 0, inv_bits = 0;
        if (curve_tables && c.kind == PT_CURVE_TABULATED && c.data_count >= PT_CURVE_GRID_MIN_KNOTS && c.data_count <= 255u) {
            const float* kd = d.curve_data + c.data_offset;
            const uint32_t n = c.data_count;
            bool sorted = true;
            for (uint32_t k = 0; k < n; ++k) sorted = sorted && std::isfinite(kd[2 * k]) && (k == 0 || kd[2 * k] >= kd[2 * (k - 1)]);
            uint32_t cells = 16; while (cells < n) cells *= 2;
            const float x0 = kd[0], width = kd[2 * (n - 1)] - x0;
            const float inv = (float)cells / width;
            if (sorted && width > 0.0f && std::isfinite(inv) && inv > 0.0f && w.size() < (1u << 24)) {   // (the table's word offset rides in 24 bits of the record's grid word: a core section beyond 64 MB keeps the binary search)
                const float top = (float)(cells - 1);
                std::vector<uint32_t> below(cells + 1, 0);   // below[g] = knots in cells < g
                for (uint32_t k = 0; k < n; ++k) {
                    const float fi = (kd[2 * k] - x0) * inv;
                    const uint32_t g = fi >= 0.0f ? (uint32_t)(fi < top ? fi : top) : 0u;
                    below[g + 1] += 1;
                }
                for (uint32_t g = 0; g < cells; ++g) below[g + 1] += below[g];
                const uint32_t toff = (uint32_t)w.size();
                for (uint32_t g = 0; g < cells; g += 4) w.push_back(below[g] | below[g + 1] << 8 | below[g + 2] << 16 | below[g + 3] << 24);
                grid_word = toff | (cells - 1) << 24; inv_bits = fbits(inv); *curve_table_words += cells / 4;
            }
        }
        uint32_t* r = &w[curve_off[i]];
        r[0] = (uint32_t)c.kind; r[1] = (uint32_t)c.mode; r[2] = fbits(c.p0); r[3] = fbits(c.p1); r[4] = off; r[5] = has ? c.data_count : 0;
        r[PT_CURVE_GRID] = grid_word; r[PT_CURVE_GRID_INV] = inv_bits;
    }
    w[PT_HDR_CURVE_OFF] = d.curve_count ? curve_off[0] : 0; w[PT_HDR_CURVE_COUNT] = d.curve_count;
    hs->curve_offsets = curve_off;

    // texstacks
    pad16(w);
    std::vector<uint32_t> ts_off(d.texstack_count);
    for (uint32_t i = 0; i < d.texstack_count; ++i) {
        const pt_texstack& t = d.texstacks[i];
        ts_off[i] = (uint32_t)w.size();
        w.push_back((uint32_t)t.layer_count);
        for (int l = 0; l < t.layer_count; ++l) {
            const pt_texture_layer& L = d.layers[t.first_layer + l];
            w.push_back((uint32_t)L.kind);
            for (int k = 0; k < 4; ++k) w.push_back((L.kind == PT_TEXTURE4 || k == 0) ? curve_off[L.curves[k]] : 0);
            w.push_back((uint32_t)L.width); w.push_back((uint32_t)L.height); w.push_back((uint32_t)L.data_offset);
        }
    }

    // materials
    pad16(w);
    w[PT_HDR_MATERIAL_OFF] = (uint32_t)w.size(); w[PT_HDR_MATERIAL_COUNT] = d.material_count;
    for (uint32_t i = 0; i < d.material_count; ++i) {
        const pt_material& m = d.materials[i];
        uint32_t r[PT_MAT_WORDS] = {0};
        r[PT_MAT_KIND] = (uint32_t)m.kind;
        r[PT_MAT_ALPHA] = fbits(m.alpha);
        r[PT_MAT_SHARPNESS] = fbits(1.0f + std::fabs(m.sharpness));  // SharpLight::new, sharp_light.rs:26
        r[PT_MAT_SIDEDNESS] = (uint32_t)m.sidedness;
        if (m.kind == PT_MATERIAL_LAMBERTIAN) r[PT_MAT_TEXSTACK] = ts_off[m.texstack];
        if (m.kind == PT_MATERIAL_GGX) {
            r[PT_MAT_ETA] = curve_off[m.curve_eta]; r[PT_MAT_ETA_O] = curve_off[m.curve_eta_o]; r[PT_MAT_KAPPA] = curve_off[m.curve_kappa];
            // GGX::new: metallic = kappa.evaluate_integral(BOUNDED_VISIBLE_RANGE, 100, false) > 0 (ggx.rs:205)
            float sum = 0.0f, step = (750.0f - 380.0f) / 100.0f;
            ptd::SceneView view{w.data(), hs->tex.data()};
            for (int k = 0; k < 100; ++k) sum += ptd::curve_eval(view, curve_off[m.curve_kappa], 380.0f + (float)k * step) * step;
            r[PT_MAT_METALLIC] = sum > 0.0f ? 1u : 0u;
        }
        if (m.kind == PT_MATERIAL_DIFFUSE_LIGHT || m.kind == PT_MATERIAL_SHARP_LIGHT) { r[PT_MAT_EMIT] = curve_off[m.curve_emit]; r[PT_MAT_BOUNCE] = curve_off[m.curve_bounce]; }
        if (m.kind == PT_MATERIAL_PASSTHROUGH) r[PT_MAT_BOUNCE] = curve_off[m.curve_bounce];
        r[PT_MAT_MEDIUMS] = (uint32_t)m.outer_medium | (uint32_t)m.inner_medium << 8;
        w.insert(w.end(), r, r + PT_MAT_WORDS);
    }
    // mediums (World::mediums): read by the medium-aware walk only
    if (d.medium_count) {
        pad16(w);
        w[PT_HDR_MEDIUM_OFF] = (uint32_t)w.size(); w[PT_HDR_MEDIUM_COUNT] = d.medium_count;
        for (uint32_t i = 0; i < d.medium_count; ++i) {
            const pt_medium& m = d.mediums[i];
            uint32_t r[PT_MEDIUM_WORDS] = {0};
            r[PT_MED_KIND] = (uint32_t)m.kind;
            if (m.kind == PT_MEDIUM_HG) { r[PT_MED_G] = curve_off[m.curve_g]; r[PT_MED_SIGMA_A] = curve_off[m.curve_sigma_a]; r[PT_MED_SIGMA_S] = curve_off[m.curve_sigma_s]; }
            else r[PT_MED_IOR] = curve_off[m.curve_ior];
            r[PT_MED_CORRECTIVE] = fbits(m.corrective_factor);
            w.insert(w.end(), r, r + PT_MEDIUM_WORDS);
        }
    }

    // meshes: per-mesh BVH over triangles (Mesh::init, mesh.rs:283-305) + gathered triangle records
    // mesh data (BVH nodes, triangles, normals, leaf lists) goes into its own section behind the core blob; its offsets
    // are relative to the section (SceneView::m), so the core alone can be staged in LDS when the whole scene does not fit
    std::vector<uint32_t> md(4, 0u);
    std::vector<uint32_t> mesh_perm0(d.mesh_count, 0), mesh_perm1(d.mesh_count, 0);
    std::vector<uint32_t> mesh_off(d.mesh_count), mesh_node_off(d.mesh_count), mesh_node_count(d.mesh_count);
    std::vector<Box> mesh_box(d.mesh_count);
    hs->mesh_has_light.assign(d.mesh_count, 0);
    std::vector<uint32_t> mesh_light_faces(d.mesh_count, 0);
    for (uint32_t mi = 0; mi < d.mesh_count; ++mi) {
        const pt_mesh& m = d.meshes[mi];
        if ((size_t)m.vertex_offset + m.vertex_count > d.vertex_count) return fail("mesh vertices out of range");
        if ((size_t)m.index_offset + 3 * (size_t)m.face_count > d.index_count) return fail("mesh indices out of range");
        if (m.normal_offset >= 0 && (size_t)m.normal_offset + m.vertex_count > d.normal_count) return fail("mesh normals out of range");
        if (m.face_material_offset >= 0 && (size_t)m.face_material_offset + m.face_count > d.face_material_count) return fail("mesh face materials out of range");
        if (m.face_count == 0) return fail("mesh without faces");
        const float* V = d.vertices + 3 * (size_t)m.vertex_offset;
        Box mb = box_empty();
        for (uint32_t v = 0; v < m.vertex_count; ++v) box_grow(mb, V + 3 * v);  // Mesh::new bounding box over all vertices
        mesh_box[mi] = mb;
        std::vector<Box> tb(m.face_count);
        for (uint32_t f = 0; f < m.face_count; ++f) {
            const uint32_t* ix = d.indices + m.index_offset + 3 * (size_t)f;
            for (int k = 0; k < 3; ++k) if (ix[k] >= m.vertex_count) return fail("mesh index out of range");
            Box b = box_of_corners(V + 3 * ix[0], V + 3 * ix[1]); box_grow(b, V + 3 * ix[2]);  // mesh.rs:57-64
            tb[f] = b;
        }
        pad16(md);
        uint32_t node_off = (uint32_t)md.size();
        // (a node's exit link is 27 bits of a word that also holds the form of its box test, pt_blob.h: 2 n - 1 nodes must fit)
        if (2ull * tb.size() >= (1ull << 27)) { *err = "mesh " + std::to_string(mi) + ": " + std::to_string(tb.size()) + " triangles make a BVH of 2^27 or more nodes (the skip-link field is 27 bits)"; return false; }
        std::vector<uint32_t> nodes; BvhBuilder bb(tb, nodes); bb.build();
        md.insert(md.end(), nodes.begin(), nodes.end());
        mesh_node_off[mi] = node_off; mesh_node_count[mi] = (uint32_t)(nodes.size() / PT_NODE_WORDS);
        uint32_t tri_off = (uint32_t)md.size();
        for (uint32_t f = 0; f < m.face_count; ++f) {
            const uint32_t* ix = d.indices + m.index_offset + 3 * (size_t)f;
            uint32_t mat = m.face_material_offset >= 0 ? d.face_materials[m.face_material_offset + f] : PT_MATERIAL_ID(PT_TAG_MATERIAL, 0);
            if (!material_ok(mat) || mat == PT_MATERIAL_NONE) return fail("mesh face material out of range");
            if (PT_MATERIAL_TAG(mat) == PT_TAG_LIGHT) mesh_light_faces[mi]++;
            for (int k = 0; k < 3; ++k) {
                const float* p = V + 3 * ix[k];
                md.push_back(fbits(p[0])); md.push_back(fbits(p[1])); md.push_back(fbits(p[2])); md.push_back(k == 0 ? mat : 0u);
            }
        }
        // Without vertex normals a hit's normal is the face's: normalize(cross(p0 - p2, p1 - p2)) (mesh.rs:176-181), normalised once more by
        // HitRecord::new (hittable.rs:30-39).  Both depend on the triangle alone, so they are computed here, with the very operations
        // hit_record would use (same header, no contraction, IEEE division and square root on both sides), and the second vertex's spare
        // word points at the result: two square roots, six divisions and a cross product less per closest hit.
        if (m.normal_offset < 0) {
            pad16(md);
            const uint32_t fn_off = (uint32_t)md.size();
            for (uint32_t f = 0; f < m.face_count; ++f) {
                const size_t at = tri_off + (size_t)f * PT_TRI_WORDS;
                auto vert = [&](int k) { return ptd::f3(pt_u2f(md[at + 4 * k]), pt_u2f(md[at + 4 * k + 1]), pt_u2f(md[at + 4 * k + 2])); };
                const ptd::F3 p0 = vert(0), p1 = vert(1), p2 = vert(2);
                const ptd::F3 n = ptd::normalize(ptd::normalize(ptd::cross(ptd::sub(p0, p2), ptd::sub(p1, p2))));
                md[at + 7] = fn_off + 4 * f;
                md.push_back(fbits(n.x)); md.push_back(fbits(n.y)); md.push_back(fbits(n.z)); md.push_back(0u);
            }
        }
        uint32_t normal_off = 0;
        if (m.normal_offset >= 0) {
            const float* N = d.normals + 3 * (size_t)m.normal_offset;
            normal_off = (uint32_t)md.size();
            for (uint32_t f = 0; f < m.face_count; ++f) {
                const uint32_t* ix = d.indices + m.index_offset + 3 * (size_t)f;
                for (int k = 0; k < 3; ++k) { const float* p = N + 3 * ix[k]; md.push_back(fbits(p[0])); md.push_back(fbits(p[1])); md.push_back(fbits(p[2])); md.push_back(0u); }
            }
        }
        // permuted copies of the triangles of a mesh small enough to be taken into the sweep table (pt_blob.h)
        if (m.face_count < PT_SWEEP_MAX_BITS) {
            for (int variant = 0; variant < 2; ++variant) {
                (variant == 0 ? mesh_perm0 : mesh_perm1)[mi] = (uint32_t)md.size() - tri_off;
                for (uint32_t f = 0; f < m.face_count; ++f)
                    for (int k = 0; k < 3; ++k) {
                        const size_t at = tri_off + f * PT_TRI_WORDS + 4 * k;
                        const uint32_t q[4] = {md[at], md[at + 1], md[at + 2], md[at + 3]};  // (copied: push_back reallocates)
                        if (variant == 0) { md.push_back(q[1]); md.push_back(q[2]); md.push_back(q[0]); }   // (y, z, x)
                        else { md.push_back(q[2]); md.push_back(q[0]); md.push_back(q[1]); }               // (z, x, y)
                        md.push_back(q[3]);
                    }
            }
        }
        // leaf list for mesh_sweep: the leaf boxes alone, in pre-order (a dense wave of rays inside a mesh of a few hundred
        // triangles tests them all, 64 at a time, instead of walking the tree lane by lane)
        uint32_t leaf_off = 0, leaf_count = 0;
        if (m.face_count <= PT_MESH_SWEEP_MAX) {
            pad16(md);
            leaf_off = (uint32_t)md.size();
            for (size_t k = 0; k < nodes.size() / PT_NODE_WORDS; ++k) {
                const uint32_t* nd = &nodes[k * PT_NODE_WORDS];
                if (nd[7] == PT_NODE_INNER) continue;
                uint32_t rec[8] = {nd[0], nd[1], nd[2], tri_off + nd[7] * PT_TRI_WORDS, nd[4], nd[5], nd[6], PT_NODE_CODE(nd[3])};   // [7]: the form of the box test
                md.insert(md.end(), rec, rec + 8);
                ++leaf_count;
            }
        }
        // group boxes for mesh_sweep: a ray tests the group boxes first and then only the leaves of the groups it enters (a box that
        // holds a leaf's box passes AABB::hit whenever the leaf's does, like a BVH ancestor)
        uint32_t group_off = 0;
        if (leaf_count > PT_MESH_GROUP_MIN) {
            pad16(md);
            group_off = (uint32_t)md.size();
            for (uint32_t first = 0; first < leaf_count; first += PT_MESH_GROUP) {
                float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
                for (uint32_t t = first; t < leaf_count && t < first + PT_MESH_GROUP; ++t) {
                    const uint32_t* lf = &md[leaf_off + t * 8u];
                    for (int k = 0; k < 3; ++k) { mn[k] = std::fmin(mn[k], bits_f(lf[k])); mx[k] = std::fmax(mx[k], bits_f(lf[4 + k])); }
                }
                const uint32_t flat = (mn[0] == mx[0] || mn[1] == mx[1] || mn[2] == mx[2]) ? 1u : 0u;
                uint32_t rec[8] = {fbits(mn[0]), fbits(mn[1]), fbits(mn[2]), 0u, fbits(mx[0]), fbits(mx[1]), fbits(mx[2]), flat};
                md.insert(md.end(), rec, rec + 8);
            }
        }
        mesh_off[mi] = (uint32_t)w.size();
        uint32_t rec[PT_MESH_WORDS] = {node_off, (uint32_t)(nodes.size() / PT_NODE_WORDS), tri_off, normal_off, m.face_count, leaf_off, leaf_count, group_off};
        for (int k = 8; k < PT_MESH_WORDS; ++k) rec[k] = 0u;
        {   // the inner sphere of a closed mesh without light faces (pt_blob.h PT_MESH_INNER_*)
            float c[3] = {0.0f, 0.0f, 0.0f}, r = 0.0f;
            std::vector<float> more;
            if (mesh_light_faces[mi] == 0 && m.face_count >= 4 && m.face_count <= 200000 && inner_sphere(V, m.vertex_count, d.indices + m.index_offset, m.face_count, mb, c, &r, &more)) {
                rec[PT_MESH_INNER_C] = fbits(c[0]); rec[PT_MESH_INNER_C + 1] = fbits(c[1]); rec[PT_MESH_INNER_C + 2] = fbits(c[2]); rec[PT_MESH_INNER_R] = fbits(r);
                if (!more.empty()) {   // (in the core section, in front of the record: four words a ball)
                    pad16(w);
                    rec[PT_MESH_MORE_OFF] = (uint32_t)w.size(); rec[PT_MESH_MORE_COUNT] = (uint32_t)(more.size() / 4);
                    for (float x : more) w.push_back(fbits(x));
                    mesh_off[mi] = (uint32_t)w.size();
                }
            }
            if (getenv("PT_AMD_HOST_VERBOSE")) fprintf(stderr, "mesh %u: %u faces, inner ball centre (%g, %g, %g) radius %g (0 = none)\n", mi, m.face_count, c[0], c[1], c[2], r);
            const double dx = (double)mb.mx[0] - mb.mn[0], dy = (double)mb.mx[1] - mb.mn[1], dz = (double)mb.mx[2] - mb.mn[2];
            rec[PT_MESH_REACH] = fbits((float)(std::sqrt(dx * dx + dy * dy + dz * dz) * 1.0001));
            {   // the slabs of the 26-DOP beyond the box (pt_blob.h PT_MESH_DOP_*): every vertex, f64, widened by 1e-3 of the slab's width and rounded outward
                static const int dirs[PT_MESH_DOP_DIRS][3] = {{1, 1, 0}, {1, -1, 0}, {1, 0, 1}, {1, 0, -1}, {0, 1, 1}, {0, 1, -1}, {1, 1, 1}, {1, 1, -1}, {1, -1, 1}, {1, -1, -1}};
                pad16(w);
                const uint32_t dop_off = (uint32_t)w.size();
                bool centred = true;   // (no slab lies more than 500 of its own widths from the mesh's origin: mesh_surely_missed's error budget)
                for (int k = 0; k < PT_MESH_DOP_DIRS; ++k) {
                    double lo = INFINITY, hi = -INFINITY;
                    for (uint32_t v = 0; v < m.vertex_count; ++v) {
                        const double x = (double)dirs[k][0] * V[3 * v] + (double)dirs[k][1] * V[3 * v + 1] + (double)dirs[k][2] * V[3 * v + 2];
                        lo = std::fmin(lo, x); hi = std::fmax(hi, x);
                    }
                    const double margin = 1e-3 * (hi - lo);
                    const float flo = std::nextafterf((float)(lo - margin), -INFINITY), fhi = std::nextafterf((float)(hi + margin), INFINITY);
                    w.push_back(fbits(flo)); w.push_back(fbits(fhi));
                    centred = centred && std::isfinite(lo) && std::isfinite(hi) && hi > lo && std::fmax(std::fabs(lo), std::fabs(hi)) <= 500.0 * (hi - lo);
                }
                w.push_back(0u);
                const bool finite = centred;
                rec[PT_MESH_DOP_OFF] = finite ? dop_off : 0u;
                pad16(w);
                mesh_off[mi] = (uint32_t)w.size();
            }
        }
        w.insert(w.end(), rec, rec + PT_MESH_WORDS);
    }

    // instances + their world boxes (Instance::aabb, instance.rs:65-72; Matrix4x4 * AABB, aabb.rs:116-138)
    pad16(w);
    w[PT_HDR_INSTANCE_OFF] = (uint32_t)w.size(); w[PT_HDR_INSTANCE_COUNT] = d.instance_count;
    std::vector<Box> ibox(d.instance_count);
    std::vector<uint32_t> lights;
    for (uint32_t i = 0; i < d.instance_count; ++i) {
        const pt_instance& in = d.instances[i];
        if (!material_ok(in.material)) return fail("instance material out of range");
        uint32_t r[PT_INST_WORDS] = {0};
        r[PT_INST_KIND] = (uint32_t)in.kind;
        r[PT_INST_FLAGS] = (in.has_transform ? 1u : 0u) | (in.two_sided ? 2u : 0u) | (((uint32_t)in.axis & 3u) << 2);
        r[PT_INST_MATERIAL] = in.material;
        for (int k = 0; k < 3; ++k) r[PT_INST_ORIGIN + k] = fbits(in.origin[k]);
        r[PT_INST_RADIUS] = fbits(in.radius);
        r[PT_INST_SIZE] = fbits(in.size[0]); r[PT_INST_SIZE + 1] = fbits(in.size[1]);
        for (int k = 0; k < 12; ++k) { r[PT_INST_FORWARD + k] = fbits(in.forward[k]); r[PT_INST_REVERSE + k] = fbits(in.reverse[k]); }
        Box b;
        switch (in.kind) {
            case PT_SHAPE_RECT: {  // rect.rs:58-66
                if (!(in.size[0] > 0.0f && in.size[1] > 0.0f) || in.axis < 0 || in.axis > 2) return fail("bad rect");
                float hx = in.size[0] / 2.0f, hy = in.size[1] / 2.0f, v[3];
                if (in.axis == PT_AXIS_X) { v[0] = 0.0001f; v[1] = hy; v[2] = hx; }
                else if (in.axis == PT_AXIS_Y) { v[0] = hx; v[1] = 0.0001f; v[2] = hy; }
                else { v[0] = hx; v[1] = hy; v[2] = 0.0001f; }
                float lo[3], hi[3]; for (int k = 0; k < 3; ++k) { lo[k] = in.origin[k] - v[k]; hi[k] = in.origin[k] + v[k]; }
                b = box_of_corners(lo, hi); break;
            }
            case PT_SHAPE_SPHERE: {  // sphere.rs:24-31
                if (!(in.radius > 0.0f)) return fail("bad sphere");
                float lo[3], hi[3]; for (int k = 0; k < 3; ++k) { lo[k] = in.origin[k] - in.radius; hi[k] = in.origin[k] + in.radius; }
                b = box_of_corners(lo, hi); break;
            }
            case PT_SHAPE_DISK: {  // disk.rs:23-28 (radius / 2: reference quirk kept)
                if (!(in.radius > 0.0f)) return fail("bad disk");
                float v[3] = {in.radius / 2.0f, in.radius / 2.0f, 0.001f}, lo[3], hi[3];
                for (int k = 0; k < 3; ++k) { lo[k] = in.origin[k] - v[k]; hi[k] = in.origin[k] + v[k]; }
                b = box_of_corners(lo, hi); break;
            }
            case PT_SHAPE_MESH:
                if (in.mesh < 0 || (uint32_t)in.mesh >= d.mesh_count) return fail("instance mesh out of range");
                r[PT_INST_MESH] = mesh_off[in.mesh]; b = mesh_box[in.mesh]; break;
            default: return fail("unknown shape kind");
        }
        if (in.has_transform) {
            Box t = box_empty();
            for (int c = 0; c < 8; ++c) {
                float p[3] = {(c & 1) == 0 ? b.mn[0] : b.mx[0], ((c >> 1) & 1) == 0 ? b.mn[1] : b.mx[1], ((c >> 2) & 1) == 0 ? b.mn[2] : b.mx[2]}, q[3];
                xf_point(in.forward, p, q); box_grow(t, q);
            }
            b = t;
        }
        ibox[i] = b;
        w.insert(w.end(), r, r + PT_INST_WORDS);
        // World::new light list (world/mod.rs:42-66)
        if (in.kind == PT_SHAPE_MESH) { for (uint32_t k = 0; k < mesh_light_faces[in.mesh]; ++k) lights.push_back(i); }
        else {
            uint32_t mid = in.material == PT_MATERIAL_NONE ? PT_MATERIAL_ID(PT_TAG_MATERIAL, 0) : in.material;
            if (PT_MATERIAL_TAG(mid) == PT_TAG_LIGHT) lights.push_back(i);
        }
    }
    for (uint32_t l : lights) if (d.instances[l].kind == PT_SHAPE_MESH) return fail("mesh lights cannot be sampled (todo!() in the reference, mesh.rs:213-232)");
    {   // convex closed mesh instances (pt_blob.h PT_INST_CONVEX_*): certified once per instance, in world space
        auto lightish = [&](uint32_t i) {   // can a hit on instance i carry a Light tag
            const pt_instance& in = d.instances[i];
            if (in.material != PT_MATERIAL_NONE) return PT_MATERIAL_TAG(in.material) == PT_TAG_LIGHT;
            return in.kind == PT_SHAPE_MESH && mesh_light_faces[in.mesh] != 0;
        };
        // (a Disk's box in the tree is the reference's, half the radius — disk.rs:24-28, a kept quirk — and does not hold the disk: the certificate's "no light near the body"
        // is about where the light IS.  Found by the soak, seed 197995: a lit disk whose reference box cleared a glass cube's while its rim reached into the cube.)
        auto true_box = [&](uint32_t i) {
            const pt_instance& in = d.instances[i];
            if (in.kind != PT_SHAPE_DISK) return ibox[i];
            float lo[3], hi[3];
            const float v[3] = {in.radius, in.radius, 0.001f};
            for (int k = 0; k < 3; ++k) { lo[k] = in.origin[k] - v[k]; hi[k] = in.origin[k] + v[k]; }
            Box b = box_of_corners(lo, hi);
            if (in.has_transform) {
                Box t = box_empty();
                for (int c = 0; c < 8; ++c) {
                    float p[3] = {(c & 1) == 0 ? b.mn[0] : b.mx[0], ((c >> 1) & 1) == 0 ? b.mn[1] : b.mx[1], ((c >> 2) & 1) == 0 ? b.mn[2] : b.mx[2]}, q[3];
                    xf_point(in.forward, p, q); box_grow(t, q);
                }
                b = t;
            }
            return b;
        };
        std::vector<Box> light_boxes;
        for (uint32_t i = 0; i < d.instance_count; ++i) if (lightish(i)) light_boxes.push_back(true_box(i));
        std::vector<int> mesh_closed(d.mesh_count, -1);
        double work_left = 4e8;   // (the certificate is O(faces x vertices + faces^2) per INSTANCE: a scene of thousands of mesh instances certifies the first of them — about a second — and leaves the rest alone)
        std::vector<std::vector<uint32_t>> mesh_face_flags(d.mesh_count);
        bool any = false;
        for (uint32_t i = 0; i < d.instance_count && d.instance_count <= 65536u; ++i) {   // (a hit's instance word and the mark a ray carries name the instance in 16 bits: a scene of more instances takes no certificate)
            const pt_instance& in = d.instances[i];
            if (in.kind != PT_SHAPE_MESH || lightish(i)) continue;
            const pt_mesh& m = d.meshes[in.mesh];
            const float* V = d.vertices + 3 * (size_t)m.vertex_offset;
            const uint32_t* ix = d.indices + m.index_offset;
            if (m.face_count > 4096) continue;
            if (mesh_closed[in.mesh] < 0) mesh_closed[in.mesh] = mesh_is_closed(V, m.vertex_count, ix, m.face_count, mesh_box[in.mesh]) ? 1 : 0;
            if (!mesh_closed[in.mesh]) continue;
            const double work = (double)m.face_count * m.vertex_count + 4.0 * (double)m.face_count * m.face_count;
            if (work > work_left) continue;
            work_left -= work;
            std::vector<uint32_t> face_flags;
            uint32_t cf = convex_certificate(V, m.vertex_count, ix, m.face_count, m.normal_offset >= 0 ? d.normals + 3 * (size_t)m.normal_offset : nullptr, in, ibox[i], light_boxes, &face_flags);
            if (cf != 0u) {
                // The faces' flag words ride in the MESH's triangle records: with several certified instances of one mesh a face keeps the weaker of their claims (the larger
                // outward threshold — 0, "none", being the largest —, the weaker inward flag: whole face > inside only > none).
                std::vector<uint32_t>& mf = mesh_face_flags[in.mesh];
                if (mf.empty()) mf = face_flags;
                else for (uint32_t f = 0; f < m.face_count; ++f) {
                    const uint32_t ta = mf[f] >> PT_TRI_OUT_SHIFT, tb = face_flags[f] >> PT_TRI_OUT_SHIFT, t = (ta == 0u || tb == 0u) ? 0u : (ta > tb ? ta : tb);
                    const uint32_t ia = mf[f] & 3u, ib = face_flags[f] & 3u;
                    const uint32_t in_bits = (ia == 0u || ib == 0u) ? 0u : ((ia == PT_TRI_IN_SAFE && ib == PT_TRI_IN_SAFE) ? PT_TRI_IN_SAFE : PT_TRI_IN_SAFE_INNER);
                    mf[f] = t << PT_TRI_OUT_SHIFT | in_bits;
                }
            }
            if (getenv("PT_AMD_HOST_VERBOSE")) fprintf(stderr, "instance %u: mesh %d, convex certificate %s%s\n", i, in.mesh, (cf & PT_INST_CONVEX_OUT) ? "OUT " : "none", (cf & PT_INST_CONVEX_IN) ? "IN" : "");
            w[w[PT_HDR_INSTANCE_OFF] + i * PT_INST_WORDS + PT_INST_FLAGS] |= cf;
            any = any || cf != 0u;
        }
        if (any) w[PT_HDR_FLAGS] |= PT_FLAG_CONVEX;
        {   // PT_HDR_CONVEX_INST: the only OUT-certified instance, if there is exactly one
            uint32_t count = 0, which = 0;
            for (uint32_t i = 0; i < d.instance_count; ++i) if (w[w[PT_HDR_INSTANCE_OFF] + i * PT_INST_WORDS + PT_INST_FLAGS] & PT_INST_CONVEX_OUT) { ++count; which = i; }
            w[PT_HDR_CONVEX_INST] = count == 1 ? which + 1u : 0u;
        }
        for (uint32_t mi = 0; mi < d.mesh_count; ++mi) {   // the faces' flag words (PT_TRI_FLAGS) into the triangle records, and into their permuted copies
            if (mesh_face_flags[mi].empty()) continue;
            const uint32_t tri_off = w[mesh_off[mi] + PT_MESH_TRI_OFF];
            for (uint32_t f = 0; f < d.meshes[mi].face_count; ++f) {
                const uint32_t word = mesh_face_flags[mi][f];
                md[tri_off + (size_t)f * PT_TRI_WORDS + PT_TRI_FLAGS] = word;
                if (mesh_perm0[mi] != 0u) { md[tri_off + mesh_perm0[mi] + (size_t)f * PT_TRI_WORDS + PT_TRI_FLAGS] = word; md[tri_off + mesh_perm1[mi] + (size_t)f * PT_TRI_WORDS + PT_TRI_FLAGS] = word; }
            }
        }
    }

    // top-level BVH over instances
    pad16(w);
    {
        if (2ull * d.instance_count >= (1ull << 27)) { *err = std::to_string(d.instance_count) + " instances make a top-level BVH of 2^27 or more nodes (the skip-link field is 27 bits)"; return false; }
        std::vector<uint32_t> nodes; BvhBuilder bb(ibox, nodes);
        bb.no_cull.assign(d.instance_count, 0);
        for (uint32_t i = 0; i < d.instance_count; ++i) bb.no_cull[i] = d.instances[i].kind == PT_SHAPE_SPHERE;
        bb.sphere_radius.assign(d.instance_count, 0.0f);
        for (uint32_t i = 0; i < d.instance_count; ++i)
            if (d.instances[i].kind == PT_SHAPE_SPHERE) bb.sphere_radius[i] = d.instances[i].has_transform ? INFINITY : d.instances[i].radius;
        bb.build();
        w[PT_HDR_TOP_NODE_OFF] = (uint32_t)w.size(); w[PT_HDR_TOP_NODE_COUNT] = (uint32_t)(nodes.size() / PT_NODE_WORDS);
        uint32_t top = (uint32_t)w.size();
        w.insert(w.end(), nodes.begin(), nodes.end());
        pad16(w);
        {   // PT_HDR_TOP_MARGIN: the sphere nodes' culling margins, one float per top-level node
            bool any = false;
            for (float m : bb.node_margin) any = any || m != 0.0f;
            if (any) {
                bb.node_margin.resize(nodes.size() / PT_NODE_WORDS, 0.0f);
                w[PT_HDR_TOP_MARGIN] = (uint32_t)w.size();
                for (float m : bb.node_margin) w.push_back(fbits(m));
                pad16(w);
            }
        }
        w[PT_HDR_LIGHT_OFF] = (uint32_t)w.size(); w[PT_HDR_LIGHT_COUNT] = (uint32_t)lights.size();
        w.insert(w.end(), lights.begin(), lights.end());
        pad16(w);
        w[PT_HDR_LIGHT_NODE_OFF] = (uint32_t)w.size();
        for (uint32_t l : lights) w.push_back(top + bb.leaf_of_shape[l] * PT_NODE_WORDS);
        // ---- leaf sweep table (world_hit_sweep): the leaf boxes of both levels in traversal pre-order, as far as 64 mask bits
        // go.  Every instance takes a bit; the triangle leaves of mesh instances are taken in ("inlined") smallest mesh first
        // while they fit; the remaining mesh instances keep only their own bit and their BVH is walked when that bit is hit.
        std::vector<char> walked(d.instance_count, 0);
        size_t sweep_bits = d.instance_count;
        {
            std::vector<uint32_t> mesh_instances;
            for (uint32_t i = 0; i < d.instance_count; ++i) if (d.instances[i].kind == PT_SHAPE_MESH) mesh_instances.push_back(i);
            std::stable_sort(mesh_instances.begin(), mesh_instances.end(), [&](uint32_t a, uint32_t b) {
                return d.meshes[d.instances[a].mesh].face_count < d.meshes[d.instances[b].mesh].face_count; });
            for (uint32_t i : mesh_instances) {
                size_t faces = d.meshes[d.instances[i].mesh].face_count;
                if (sweep_bits + faces <= PT_SWEEP_MAX_BITS) sweep_bits += faces; else walked[i] = 1;
            }
        }
        if (d.instance_count > 0 && d.instance_count <= PT_SWEEP_MAX_BITS) {
            auto is_flat = [](const uint32_t* node) { return node[0] == node[4] || node[1] == node[5] || node[2] == node[6]; };
            // which form of the filtered slab test a box takes (pt_device.h aabb_classify_code): 0 = it has thickness on every axis, 1 / 2 / 3 =
            // flat along x / y / z only, 4 = flat along more than one axis
            auto flat_code = [](const uint32_t* node) {
                const int fx = node[0] == node[4], fy = node[1] == node[5], fz = node[2] == node[6];
                return fx + fy + fz == 0 ? 0u : (fx + fy + fz > 1 ? 4u : (fx ? 1u : (fy ? 2u : 3u)));
            };
            pad16(w);
            std::vector<uint32_t> order;  // top-level leaf nodes in pre-order
            for (size_t k = 0; k < nodes.size() / PT_NODE_WORDS; ++k) if (nodes[k * PT_NODE_WORDS + 7] != PT_NODE_INNER) order.push_back((uint32_t)k);
            uint32_t sweep_off = (uint32_t)w.size();
            w.resize(w.size() + order.size() * PT_SWEEP_INST_WORDS, 0);
            std::vector<uint32_t> bits;
            std::vector<int> root_of;  // per bit: the bit whose box test it copies (itself if none)
            uint32_t bit = 0;
            uint64_t mesh_mask = 0, owner_mask = 0;
            bool any_walked = false;
            for (size_t j = 0; j < order.size(); ++j) {
                const uint32_t* nd = &nodes[order[j] * PT_NODE_WORDS];
                uint32_t inst = nd[7];
                const pt_instance& in = d.instances[inst];
                uint32_t e = sweep_off + (uint32_t)j * PT_SWEEP_INST_WORDS;
                uint32_t rec_off = w[PT_HDR_INSTANCE_OFF] + inst * PT_INST_WORDS;
                uint32_t kf = (uint32_t)in.kind | (is_flat(nd) ? 1u << 8 : 0u) | (in.has_transform ? 1u << 9 : 0u) | (walked[inst] ? PT_SWEEP_WALKED : 0u) | flat_code(nd) << 11;
                uint32_t tri_list = 0, tri_count = 0, first_bit = bit++;
                uint32_t own[PT_SWEEP_BIT_WORDS] = {rec_off, 0u, e + 4, kf | inst << 16, 0u, 0u, 0u, 0u};
                bits.insert(bits.end(), own, own + PT_SWEEP_BIT_WORDS);
                root_of.push_back((int)first_bit);
                if (in.kind == PT_SHAPE_MESH && walked[inst]) any_walked = true;
                if (in.kind != PT_SHAPE_MESH) owner_mask |= 1ull << first_bit;
                std::vector<uint32_t> leaders;  // word offsets of the triangle-leaf records that keep their own box test
                std::vector<uint32_t> leader_code;
                if (in.kind == PT_SHAPE_MESH && !walked[inst]) {
                    mesh_mask |= 1ull << first_bit;
                    pad16(w);
                    tri_list = (uint32_t)w.size();
                    uint32_t tri_base = w[mesh_off[in.mesh] + PT_MESH_TRI_OFF];
                    std::vector<std::array<uint32_t, 6>> seen;  // boxes of this instance's bits, index = bit - first_bit
                    seen.push_back({nd[0], nd[1], nd[2], nd[4], nd[5], nd[6]});
                    for (uint32_t k = 0; k < mesh_node_count[in.mesh]; ++k) {
                        uint32_t mn[PT_NODE_WORDS];
                        for (int q = 0; q < PT_NODE_WORDS; ++q) mn[q] = md[mesh_node_off[in.mesh] + k * PT_NODE_WORDS + q];
                        if (mn[7] == PT_NODE_INNER) continue;
                        std::array<uint32_t, 6> box = {mn[0], mn[1], mn[2], mn[4], mn[5], mn[6]};
                        // the instance's own box is tested against the world ray: only an untransformed instance may share it
                        uint32_t alias = 0;
                        for (size_t q = in.has_transform ? 1 : 0; q < seen.size() && !alias; ++q) if (seen[q] == box) alias = first_bit + (uint32_t)q + 1;
                        seen.push_back(box);
                        root_of.push_back(alias ? root_of[alias - 1] : (int)bit);
                        uint32_t triw = tri_base + mn[7] * PT_TRI_WORDS, flat = is_flat(mn) ? 1u : 0u;
                        uint32_t box_words = e + 4;  // a copy of the instance's own box: settled with it, never looked up
                        if (!alias) {
                            leader_code.push_back(flat_code(mn));
                            leaders.push_back((uint32_t)w.size());
                            uint32_t rec[PT_SWEEP_TRI_WORDS] = {mn[0], mn[1], mn[2], 0u, mn[4], mn[5], mn[6], 0u};  // [3], [7]: the mask, below
                            box_words = (uint32_t)w.size();
                            w.insert(w.end(), rec, rec + PT_SWEEP_TRI_WORDS);
                        } else if (root_of.back() != (int)first_bit) box_words = bits[(size_t)root_of.back() * PT_SWEEP_BIT_WORDS + 2];
                        uint32_t tb[PT_SWEEP_BIT_WORDS] = {rec_off, triw, box_words, (kf & ~0x100u) | flat << 8 | inst << 16, 0u, 0u, mesh_perm0[in.mesh], mesh_perm1[in.mesh]};
                        bits.insert(bits.end(), tb, tb + PT_SWEEP_BIT_WORDS);
                        if (in.has_transform) owner_mask |= 1ull << bit;
                        ++tri_count; ++bit;
                    }
                }
                // the mask a box test sets: its own bit and the bits of the later leaves of this instance whose box is the same
                auto mask_of = [&](uint32_t leader_bit) { uint64_t m = 0; for (uint32_t b2 = first_bit; b2 < bit; ++b2) if (root_of[b2] == (int)leader_bit) m |= 1ull << b2; return m; };
                {
                    uint32_t b2 = first_bit + 1;
                    for (uint32_t off : leaders) {
                        while (root_of[b2] != (int)b2) ++b2;
                        uint64_t m = mask_of(b2++);
                        w[off + 3] = (uint32_t)m; w[off + 7] = (uint32_t)(m >> 32);
                    }
                }
                // The order of the box tests is free (the masks carry the leaf order): the records are grouped by the form of the test, so
                // that the sweep runs one specialised loop per form; [12] holds the sizes of the groups 0..3, the rest is group 4.
                uint32_t group_sizes = 0;
                if (!leaders.empty()) {
                    std::vector<size_t> order(leaders.size());
                    for (size_t k = 0; k < order.size(); ++k) order[k] = k;
                    std::stable_sort(order.begin(), order.end(), [&](size_t a2, size_t b2) { return leader_code[a2] < leader_code[b2]; });
                    std::vector<uint32_t> block(leaders.size() * PT_SWEEP_TRI_WORDS);
                    std::vector<uint32_t> moved(leaders.size());   // new word offset of leader k
                    for (size_t pos = 0; pos < order.size(); ++pos) {
                        const size_t k = order[pos];
                        for (int q = 0; q < PT_SWEEP_TRI_WORDS; ++q) block[pos * PT_SWEEP_TRI_WORDS + q] = w[leaders[k] + q];
                        moved[k] = tri_list + (uint32_t)pos * PT_SWEEP_TRI_WORDS;
                        if (leader_code[k] < 4) group_sizes += 1u << (8 * leader_code[k]);
                    }
                    for (uint32_t b2 = first_bit + 1; b2 < bit; ++b2) {   // the bit table's box references follow the records
                        uint32_t& bw = bits[(size_t)b2 * PT_SWEEP_BIT_WORDS + 2];
                        for (size_t k = 0; k < leaders.size(); ++k) if (bw == leaders[k]) { bw = moved[k]; break; }
                    }
                    for (size_t q = 0; q < block.size(); ++q) w[tri_list + q] = block[q];
                }
                uint64_t own_mask = mask_of(first_bit);
                {   // the instance's bits, all of them, in its own record (PT_INST_SWEEP_MASK: what a ray that may skip the instance drops from its leaf mask)
                    const uint64_t all = (bit >= 64u ? ~0ull : (1ull << bit) - 1ull) & ~((1ull << first_bit) - 1ull);
                    w[rec_off + PT_INST_SWEEP_MASK] = (uint32_t)all; w[rec_off + PT_INST_SWEEP_MASK + 1] = (uint32_t)(all >> 32);
                }
                uint32_t* r = &w[e];
                r[0] = rec_off; r[1] = kf | first_bit << 16 | (uint32_t)leaders.size() << 24; r[2] = (uint32_t)own_mask; r[3] = (uint32_t)(own_mask >> 32);
                r[4] = nd[0]; r[5] = nd[1]; r[6] = nd[2]; r[7] = tri_list; r[8] = nd[4]; r[9] = nd[5]; r[10] = nd[6]; r[11] = tri_count;
                r[12] = group_sizes; r[13] = 0u; r[14] = inst; r[15] = 0u;
            }
            // The instance records too may stand in any order (their masks carry the leaf order).  Those whose box test is all there is to do — no
            // triangle leaf with a test of its own: analytic shapes, the walls whose two triangles ride in the instance's mask, walked meshes — come
            // first, grouped by the form of the test: the sweep runs them through tight loops (pt_device.h sweep_masks), the others through the
            // general one.  A sphere is never culled (pt_device.h beyond) and stays with the general loop.
            {
                std::vector<size_t> perm(order.size());
                for (size_t j = 0; j < perm.size(); ++j) perm[j] = j;
                auto simple_code = [&](size_t j) -> uint32_t {   // 0..4: simple with that test form; 5: general
                    const uint32_t kf = w[sweep_off + j * PT_SWEEP_INST_WORDS + 1];
                    return ((kf >> 24) == 0u && (kf & 0xffu) != PT_SHAPE_SPHERE) ? ((kf >> 11) & 7u) : 5u;
                };
                std::stable_sort(perm.begin(), perm.end(), [&](size_t a2, size_t b2) { return simple_code(a2) < simple_code(b2); });
                std::vector<uint32_t> block(order.size() * PT_SWEEP_INST_WORDS);
                std::vector<uint32_t> new_box(order.size());   // old record j -> word offset of its box in the new order
                uint32_t counts[6] = {0, 0, 0, 0, 0, 0};
                for (size_t pos = 0; pos < perm.size(); ++pos) {
                    const size_t j = perm[pos];
                    for (int q = 0; q < PT_SWEEP_INST_WORDS; ++q) block[pos * PT_SWEEP_INST_WORDS + q] = w[sweep_off + j * PT_SWEEP_INST_WORDS + q];
                    new_box[j] = sweep_off + (uint32_t)pos * PT_SWEEP_INST_WORDS + 4;
                    counts[simple_code(j)] += 1;
                }
                for (size_t k = 0; k < bits.size() / PT_SWEEP_BIT_WORDS; ++k) {   // the bit table's box references follow the records
                    uint32_t& bw = bits[k * PT_SWEEP_BIT_WORDS + 2];
                    if (bw >= sweep_off && bw < sweep_off + order.size() * PT_SWEEP_INST_WORDS) bw = new_box[(bw - sweep_off) / PT_SWEEP_INST_WORDS];
                }
                for (size_t q = 0; q < block.size(); ++q) w[sweep_off + q] = block[q];
                w[PT_HDR_SWEEP_SIMPLE] = counts[0] | counts[1] << 8 | counts[2] << 16 | counts[3] << 24;   // (at most 64 instances)
                w[PT_HDR_SWEEP_SIMPLE + 1] = counts[4];
            }
            for (size_t k = 0; k < root_of.size(); ++k)
                if (root_of[k] != (int)k) {
                    uint64_t m = 1ull << k;
                    bits[(size_t)root_of[k] * PT_SWEEP_BIT_WORDS + 4] |= (uint32_t)m; bits[(size_t)root_of[k] * PT_SWEEP_BIT_WORDS + 5] |= (uint32_t)(m >> 32);
                }
            pad16(w);
            w[PT_HDR_SWEEP_BITS_OFF] = (uint32_t)w.size();
            w.insert(w.end(), bits.begin(), bits.end());
            w[PT_HDR_SWEEP_MESH_MASK] = (uint32_t)mesh_mask; w[PT_HDR_SWEEP_MESH_MASK + 1] = (uint32_t)(mesh_mask >> 32);
            w[PT_HDR_SWEEP_OWNER_MASK] = (uint32_t)owner_mask; w[PT_HDR_SWEEP_OWNER_MASK + 1] = (uint32_t)(owner_mask >> 32);
            w[PT_HDR_SWEEP_OFF] = sweep_off; w[PT_HDR_SWEEP_COUNT] = (uint32_t)order.size();
            if (any_walked) w[PT_HDR_FLAGS] |= PT_FLAG_SWEEP_WALKS;
        }
    }
    pad16(w);

    // environment + world radius (World::new, world/mod.rs:69-81)
    w[PT_HDR_ENV_KIND] = (uint32_t)d.environment.kind;
    w[PT_HDR_ENV_STRENGTH] = fbits(d.environment.strength);
    w[PT_HDR_ENV_CURVE] = d.environment.kind == PT_ENV_HDR ? 0u : curve_off[d.environment.curve];
    if (d.environment.kind == PT_ENV_HDR) {
        const pt_environment& e = d.environment;
        w[PT_HDR_ENV_TEXSTACK] = ts_off[e.texstack];
        for (int k = 0; k < 12; ++k) { w[PT_HDR_ENV_FORWARD + k] = fbits(e.rotation_forward[k]); w[PT_HDR_ENV_REVERSE + k] = fbits(e.rotation_reverse[k]); }
        if (e.importance_width > 0 && e.importance_height > 0 && e.strength > 0.0f) {
            // ImportanceMap::bake_raw (src/world/importance_map.rs:78-253) over BOUNDED_VISIBLE_RANGE (parsing/environment.rs:140):
            // texel luminance = sum of 100 left-Riemann samples of luminance(lambda) * texel spectrum(lambda); per-row pdf and
            // cumulative mass over the columns, marginal over the rows.  Tables go to texture memory (HBM).
            const uint32_t V = (uint32_t)e.importance_height, H = (uint32_t)e.importance_width;
            const int N = 100;
            float lum[N], lam[N];
            const float step = (750.0f - 380.0f) / (float)N;
            ptd::SceneView view{w.data(), hs->tex.data()};
            for (int i = 0; i < N; ++i) {
                lam[i] = 380.0f + (float)i * step;
                if (e.importance_luminance_curve >= 0) lum[i] = ptd::curve_eval(view, curve_off[e.importance_luminance_curve], lam[i]);
                else { float xb, yb, zb; ptd::xyz_bar(lam[i] * 10.0f, &xb, &yb, &zb); lum[i] = yb; }
            }
            std::vector<float> row_pdf((size_t)V * H), row_cmf((size_t)V * H), mpdf(V), mcmf(V);
            // the spectral curves of the environment texture at the 100 wavelengths, evaluated once per channel instead of once
            // per texel (TexStack::eval_at re-evaluates them; same values)
            const pt_texstack& ets = d.texstacks[e.texstack];
            std::vector<float> cv((size_t)ets.layer_count * 4 * N, 0.0f);
            for (int li = 0; li < ets.layer_count; ++li) {
                const pt_texture_layer& L = d.layers[ets.first_layer + li];
                for (int c = 0; c < (L.kind == PT_TEXTURE4 ? 4 : 1); ++c)
                    for (int i = 0; i < N; ++i) cv[((size_t)li * 4 + c) * N + i] = ptd::curve_eval(view, curve_off[L.curves[c]], lam[i]);
            }
            float total = 0.0f;
            for (uint32_t row = 0; row < V; ++row) {
                float row_luminance = 0.0f;
                float* pdf = row_pdf.data() + (size_t)row * H; float* cmf = row_cmf.data() + (size_t)row * H;
                for (uint32_t col = 0; col < H; ++col) {
                    float u = (float)row / (float)V, v = (float)col / (float)H;
                    float cu = pt_clamp(u, 0.0f, 1.0f - PT_F32_EPSILON), cvv = pt_clamp(v, 0.0f, 1.0f - PT_F32_EPSILON);
                    float texel = 0.0f;
                    for (int i = 0; i < N; ++i) {
                        float energy = 0.0f;
                        for (int li = 0; li < ets.layer_count; ++li) {
                            const pt_texture_layer& L = d.layers[ets.first_layer + li];
                            const float* data = d.texture_data + L.data_offset;
                            size_t idx = (size_t)(cvv * (float)L.height) * (size_t)L.width + (size_t)(cu * (float)L.width);
                            const float* c = &cv[(size_t)li * 4 * N];
                            if (L.kind == PT_TEXTURE1) energy += c[i] * data[idx];
                            else { const float* t = data + 4 * idx; energy += (c[i] * t[0] + c[N + i] * t[1]) + (c[2 * N + i] * t[2] + c[3 * N + i] * t[3]); }
                        }
                        texel += lum[i] * energy * step;
                    }
                    row_luminance += texel;
                    pdf[col] = texel; cmf[col] = row_luminance;
                }
                for (uint32_t col = 0; col < H; ++col) { pdf[col] /= row_luminance; cmf[col] /= row_luminance; }
                total += row_luminance;
                mpdf[row] = row_luminance;
            }
            float run = 0.0f;
            for (uint32_t row = 0; row < V; ++row) { mpdf[row] /= total; run += mpdf[row]; mcmf[row] = run; }
            if (hs->tex.size() + 2 * row_pdf.size() + 2 * (size_t)V + 64 > 0xffffffffull) return fail("importance map too large");
            w[PT_HDR_IMAP_ROWS] = V; w[PT_HDR_IMAP_COLS] = H;
            // pdf and cmf of a table interleaved, (cmf[k], pdf[k]): what a sample reads at its end lies on one line (pt_blob.h PT_HDR_IMAP_STRIDE);
            // the tables start on 128-byte lines (the texel array in front of them has any length)
            auto interleaved = [&](const std::vector<float>& cmf, const std::vector<float>& pdf, int cmf_hdr, int pdf_hdr) {
                while (hs->tex.size() % 32) hs->tex.push_back(0.0f);
                w[cmf_hdr] = (uint32_t)hs->tex.size(); w[pdf_hdr] = (uint32_t)hs->tex.size() + 1u;
                for (size_t k = 0; k < cmf.size(); ++k) { hs->tex.push_back(cmf[k]); hs->tex.push_back(pdf[k]); }
            };
            w[PT_HDR_IMAP_STRIDE] = 2u;
            interleaved(row_cmf, row_pdf, PT_HDR_IMAP_ROW_CMF, PT_HDR_IMAP_ROW_PDF);
            interleaved(mcmf, mpdf, PT_HDR_IMAP_MARG_CMF, PT_HDR_IMAP_MARG_PDF);
            // guide tables: entry j = lower bound of j / n in the cmf, so that a search starts in a bracket of a few entries
            // instead of at the whole table (ten dependent loads from L2/HBM per search otherwise)
            auto guide = [&](const float* cmf, uint32_t n) {
                uint32_t k = 0;
                for (uint32_t j = 0; j < n + 3; ++j) {
                    if (j > n) k = n;
                    else { float t = (float)j / (float)n; while (k < n && cmf[k] < t) ++k; }
                    float bits; uint32_t kk = k; memcpy(&bits, &kk, 4);
                    hs->tex.push_back(bits);
                }
            };
            w[PT_HDR_IMAP_MARG_GUIDE] = (uint32_t)hs->tex.size(); guide(mcmf.data(), V);
            w[PT_HDR_IMAP_ROW_GUIDE] = (uint32_t)hs->tex.size();
            for (uint32_t r = 0; r < V; ++r) guide(row_cmf.data() + (size_t)r * H, H);
        }
    }
    w[PT_HDR_ENV_ANGULAR] = fbits(d.environment.angular_diameter);
    for (int k = 0; k < 3; ++k) w[PT_HDR_ENV_SUN_DIR + k] = fbits(d.environment.sun_direction[k]);
    float env_p = lights.empty() ? 1.0f : d.env_sampling_probability;
    w[PT_HDR_ENV_PROB] = fbits(env_p);
    float radius = 0.0f;
    if (d.instance_count) {
        Box wb = ibox[0]; for (uint32_t i = 1; i < d.instance_count; ++i) box_expand(wb, ibox[i]);
        float sx = wb.mx[0] - wb.mn[0], sy = wb.mx[1] - wb.mn[1], sz = wb.mx[2] - wb.mn[2];
        radius = std::sqrt(sx * sx + sy * sy + sz * sz) / 2.0f;
    }
    w[PT_HDR_WORLD_RADIUS] = fbits(radius);
    {
        uint32_t flags = w[PT_HDR_FLAGS] & (PT_FLAG_SWEEP_WALKS | PT_FLAG_CONVEX);
        for (uint32_t i = 0; i < d.instance_count; ++i) {
            const pt_instance& in = d.instances[i];
            if (in.kind == PT_SHAPE_DISK) flags |= PT_FLAG_NO_TOP_CULL;
            // a mesh whose hits can carry a Light tag is not in the light list (world/mod.rs:45-54 only looks at analytic
            // instances' own ids and mesh face ids, and mesh face lights are rejected above) but would pass pt.rs:178
            if (in.kind == PT_SHAPE_MESH && in.material != PT_MATERIAL_NONE && PT_MATERIAL_TAG(in.material) == PT_TAG_LIGHT) flags |= PT_FLAG_NO_SHADOW_BOUND;
        }
        w[PT_HDR_FLAGS] = flags;
    }
    pad16(w);
    w[PT_HDR_CORE_WORDS] = (uint32_t)w.size();
    w.insert(w.end(), md.begin(), md.end());
    pad16(w);
    w[PT_HDR_MAGIC] = PT_BLOB_MAGIC; w[PT_HDR_TOTAL_WORDS] = (uint32_t)w.size();
    hs->light_count = (uint32_t)lights.size();
    hs->material_count = d.material_count;
    hs->curve_count = d.curve_count;
    return true;
}

}  // namespace pth
