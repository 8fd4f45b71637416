// Host-only check of the box hierarchies wrt_scene_commit builds (no GPU needed): every index a walk
// can follow stays inside the array, every triangle (sphere) sits in exactly one leaf (or, for a
// sphere much larger than the rest, in the group's outlier list), every box holds its primitives,
// and no path from the root needs more stack entries than the walk has.
// build: hipcc --offload-arch=gfx950 -O1 -ffp-contract=off -std=c++17 tests/native/check_tree.hip -o <exe>
#include "../../raytracing-one-weekend_amd/csrc/wrt.hip"

#include <cstdio>
#include <random>

static int failures = 0;
#define EXPECT(cond, ...) do { if (!(cond)) { ++failures; std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); } } while (0)

static int as_int(float f) { int v; std::memcpy(&v, &f, 4); return v; }

struct Walk {
    const wrt_ctx& ctx;
    const GroupRec& g;
    const f32x4* tree;
    size_t tree_len;
    int n_nodes;
    std::vector<int> seen;
    int deepest = 0;

    void visit(int code, const float lo[3], const float hi[3], float kappa, int pushes) {
        if (code < 0) {
            EXPECT(code != TREE_DONE, "child code collides with the walk's end marker");
            const int leaf = -1 - code, n = (leaf & 7) + 1, offset = leaf >> 3;
            if (g.type == GROUP_SPHERES) {
                EXPECT(offset >= n_nodes * 4 && (offset - n_nodes * 4) % 2 == 0, "leaf offset %d not on a record", offset);
                for (int j = 0; j < n; ++j) {
                    const size_t at = (size_t)offset + 2 * (size_t)j;
                    EXPECT(at + 1 < tree_len, "leaf record %zu outside the tree (%zu)", at, tree_len);
                    if (at + 1 >= tree_len) return;
                    const f32x4 a = tree[at];
                    const int prim = as_int(tree[at + 1].x);
                    EXPECT(prim >= 0 && prim < g.count, "sphere index %d of %d", prim, g.count);
                    if (prim < 0 || prim >= g.count) continue;
                    ++seen[(size_t)prim];
                    const f32x4 rec = ctx.h_sph[(size_t)g.first + prim];
                    EXPECT(std::memcmp(&a, &rec, 16) == 0, "leaf record of sphere %d differs from the group's arrays", prim);
                    const double r = std::sqrt((double)rec.w);
                    for (int ax = 0; ax < 3; ++ax) {
                        const double rel = (double)rec[ax] - (double)g.c[ax];
                        EXPECT(rel - r >= lo[ax] && rel + r <= hi[ax], "sphere %d outside its leaf box (axis %d)", prim, ax);
                    }
                    EXPECT((double)kappa * r >= 1.0, "kappa %g below 1 / r = %g", kappa, 1.0 / r);
                    EXPECT(std::sqrt((rec.x - g.c[0]) * (double)(rec.x - g.c[0]) + (rec.y - g.c[1]) * (double)(rec.y - g.c[1]) +
                                     (rec.z - g.c[2]) * (double)(rec.z - g.c[2])) <= g.reach2 && r <= g.tree_rmax,
                           "sphere %d beyond the group's reach or largest radius", prim);
                }
                return;
            }
            EXPECT(offset >= n_nodes * 4 && (offset - n_nodes * 4) % 3 == 0, "leaf offset %d not on a record", offset);
            for (int j = 0; j < n; ++j) {
                const size_t at = (size_t)offset + 3 * (size_t)j;
                EXPECT(at + 2 < tree_len, "leaf record %zu outside the tree (%zu)", at, tree_len);
                if (at + 2 >= tree_len) return;
                const f32x4 a = tree[at], b = tree[at + 1], c = tree[at + 2];
                const int prim = as_int(c.y);
                EXPECT(prim >= 0 && prim < g.count, "triangle index %d of %d", prim, g.count);
                if (prim < 0 || prim >= g.count) continue;
                ++seen[(size_t)prim];
                const size_t i = (size_t)g.first + prim;
                const f32x4 r0 = ctx.h_tri[2 * i], r1 = ctx.h_tri[2 * i + 1];
                EXPECT(std::memcmp(&a, &r0, 16) == 0 && std::memcmp(&b, &r1, 16) == 0 && c.x == ctx.h_tri_bz[i],
                       "leaf record of triangle %d differs from the group's arrays", prim);
                const double P[3] = {r0.x, r0.y, r0.z}, A[3] = {r0.w, r1.x, r1.y}, B[3] = {r1.z, r1.w, ctx.h_tri_bz[i]};
                for (int ax = 0; ax < 3; ++ax)
                    for (double v : {P[ax], P[ax] + A[ax], P[ax] + B[ax]}) {
                        const double rel = v - (double)g.c[ax];
                        EXPECT(rel >= lo[ax] && rel <= hi[ax], "corner outside its leaf box (triangle %d axis %d)", prim, ax);
                    }
                const double la = std::sqrt(A[0] * A[0] + A[1] * A[1] + A[2] * A[2]), lb = std::sqrt(B[0] * B[0] + B[1] * B[1] + B[2] * B[2]);
                EXPECT(kappa >= 0.17 * la * lb, "kappa %g below 0.17 |A||B| = %g", kappa, 0.17 * la * lb);
            }
            return;
        }
        EXPECT(code % 4 == 0 && code / 4 < n_nodes, "node offset %d outside the node array", code);
        if (code % 4 != 0 || code / 4 >= n_nodes) return;
        const size_t at = (size_t)code;
        deepest = std::max(deepest, pushes + 1);          // this node may push one entry
        const f32x4 q0 = tree[at], q1 = tree[at + 1], q2 = tree[at + 2], q3 = tree[at + 3];
        const float lo0[3] = {q0.x, q0.y, q0.z}, hi0[3] = {q0.w, q1.x, q1.y}, lo1[3] = {q1.z, q1.w, q2.x}, hi1[3] = {q2.y, q2.z, q2.w};
        for (int ax = 0; ax < 3; ++ax) {
            EXPECT(lo0[ax] >= lo[ax] && hi0[ax] <= hi[ax] && lo1[ax] >= lo[ax] && hi1[ax] <= hi[ax], "child box outside its parent's (node %d)", code);
        }
        EXPECT(q3.x <= kappa && q3.y <= kappa, "child kappa above its parent's (node %d)", code);
        visit(as_int(q3.z), lo0, hi0, q3.x, pushes + 1);
        visit(as_int(q3.w), lo1, hi1, q3.y, pushes + 1);
    }
};

static void add_group(wrt_ctx& ctx, int count, std::mt19937& rng, int shape) {
    GroupRec g;
    std::memset(&g, 0, sizeof(g));
    g.type = GROUP_TRIANGLES; g.first = (int)ctx.h_tri_bz.size(); g.count = count;
    std::uniform_real_distribution<float> u(-1.0f, 1.0f);
    g.c[0] = 3.0f; g.c[1] = -2.0f; g.c[2] = 0.5f;
    for (int j = 0; j < count; ++j) {
        float p[3], a[3], b[3];
        for (int ax = 0; ax < 3; ++ax) {
            // shape 0: a cloud; 1: all centroids on one line (splits along one axis only); 2: all triangles the same
            p[ax] = g.c[ax] + (shape == 2 ? 0.25f : (shape == 1 && ax > 0 ? 0.0f : 2.0f * u(rng)));
            a[ax] = 0.1f * u(rng); b[ax] = 0.1f * u(rng);
            if (shape == 2) { a[ax] = 0.1f * (ax == 0); b[ax] = 0.1f * (ax == 1); }
        }
        ctx.h_tri.push_back(f32x4{p[0], p[1], p[2], a[0]});
        ctx.h_tri.push_back(f32x4{a[1], a[2], b[0], b[1]});
        ctx.h_tri_bz.push_back(b[2]);
    }
    ctx.h_groups.push_back(g);
}

static void add_sphere_group(wrt_ctx& ctx, int count, std::mt19937& rng, int huge) {
    GroupRec g;
    std::memset(&g, 0, sizeof(g));
    g.type = GROUP_SPHERES; g.first = (int)ctx.h_sph.size(); g.count = count;
    std::uniform_real_distribution<float> u(-1.0f, 1.0f);
    g.c[0] = 1.0f; g.c[1] = 0.25f; g.c[2] = -2.0f;          // the loops' reference point (a median centre in the library)
    for (int j = 0; j < count; ++j) {
        float r = 0.2f + 0.05f * u(rng);
        f32x4 s = {g.c[0] + 11.0f * u(rng), g.c[1] + 0.2f * u(rng), g.c[2] + 11.0f * u(rng), 0.0f};
        if (j < huge) { r = 1000.0f + j; s.y = -r; }
        s.w = r * r;
        ctx.h_sph.push_back(s);
        ctx.h_sph_radius.push_back(r);
        ctx.h_sph_material.push_back(0);
    }
    ctx.h_groups.push_back(g);
}

int main() {
    std::mt19937 rng(7);
    wrt_ctx ctx;
    const int counts[] = {16, 17, 64, 644, 5804, 16384, 16385, 32768, 15, 40000};
    for (int shape = 0; shape < 3; ++shape)
        for (int count : counts) add_group(ctx, count, rng, shape);
    // sphere groups: plain fields, one and four spheres much larger than the rest, five (no tree), too few left
    const int sphere_counts[][2] = {{16, 0}, {484, 1}, {191, 0}, {5000, 4}, {300, 5}, {18, 4}, {15, 0}};
    for (const auto& sc : sphere_counts) add_sphere_group(ctx, sc[0], rng, sc[1]);
    build_trees(&ctx);
    for (const GroupRec& g : ctx.h_groups) {
        if (g.type == GROUP_SPHERES) {
            const int huge = g.count == 484 ? 1 : (g.count == 5000 || g.count == 18 ? 4 : (g.count == 300 ? 5 : 0));
            const bool expected = g.count >= TREE_MIN_TRIANGLES && huge <= SPHERE_TREE_OUTLIERS && g.count - huge >= TREE_MIN_TRIANGLES;
            EXPECT((g.tree_base >= 0) == expected, "group of %d spheres (%d huge): tree_base %d", g.count, huge, g.tree_base);
            if (g.tree_base < 0) continue;
            EXPECT(g.n_outliers == huge, "group of %d spheres: %d outliers, expected %d", g.count, g.n_outliers, huge);
            const f32x4* tree = ctx.h_tree.data() + g.tree_base;
            size_t end = ctx.h_tree.size() - 1;
            for (const GroupRec& other : ctx.h_groups)
                if (other.tree_base > g.tree_base) end = std::min(end, (size_t)other.tree_base);
            const size_t len = end - (size_t)g.tree_base, in_tree = (size_t)(g.count - g.n_outliers);
            EXPECT(len > 2 * in_tree && (len - 2 * in_tree) % 4 == 0, "tree of %zu f32x4 for %zu spheres", len, in_tree);
            Walk w{ctx, g, tree, len, (int)((len - 2 * in_tree) / 4), std::vector<int>((size_t)g.count, 0)};
            for (int i = 0; i < g.n_outliers; ++i) {
                EXPECT(g.outlier[i] >= 0 && g.outlier[i] < g.count, "outlier index %d", g.outlier[i]);
                if (g.outlier[i] >= 0 && g.outlier[i] < g.count) ++w.seen[(size_t)g.outlier[i]];
            }
            const float lo[3] = {-INFINITY, -INFINITY, -INFINITY}, hi[3] = {INFINITY, INFINITY, INFINITY};
            w.visit(0, lo, hi, INFINITY, 0);
            for (int j = 0; j < g.count; ++j)
                if (w.seen[(size_t)j] != 1) { EXPECT(false, "group of %d: sphere %d is held %d times", g.count, j, w.seen[(size_t)j]); break; }
            EXPECT(w.deepest <= TREE_HEIGHT, "group of %d spheres: a walk may need %d stack entries", g.count, w.deepest);
            std::printf("group of %5d spheres:   %d stack entries at most, %d outside the tree\n", g.count, w.deepest, g.n_outliers);
            continue;
        }
        const bool expected = g.count >= TREE_MIN_TRIANGLES && g.count <= TREE_MAX_TRIANGLES;
        EXPECT((g.tree_base >= 0) == expected, "group of %d triangles: tree_base %d", g.count, g.tree_base);
        if (g.tree_base < 0) continue;
        const f32x4* tree = ctx.h_tree.data() + g.tree_base;
        // the nodes come first, then 3 x f32x4 per triangle, then the next group's tree (or the array's last entry)
        size_t end = ctx.h_tree.size() - 1;
        for (const GroupRec& other : ctx.h_groups)
            if (other.tree_base > g.tree_base) end = std::min(end, (size_t)other.tree_base);
        const size_t len = end - (size_t)g.tree_base;
        EXPECT(len > 3 * (size_t)g.count && (len - 3 * (size_t)g.count) % 4 == 0, "tree of %zu f32x4 for %d triangles", len, g.count);
        Walk w{ctx, g, tree, len, (int)((len - 3 * (size_t)g.count) / 4), std::vector<int>((size_t)g.count, 0)};
        const float lo[3] = {-INFINITY, -INFINITY, -INFINITY}, hi[3] = {INFINITY, INFINITY, INFINITY};
        w.visit(0, lo, hi, INFINITY, 0);
        for (int j = 0; j < g.count; ++j)
            if (w.seen[(size_t)j] != 1) { EXPECT(false, "group of %d: triangle %d is in %d leaves", g.count, j, w.seen[(size_t)j]); break; }
        EXPECT(w.deepest <= TREE_HEIGHT, "group of %d: a walk may need %d stack entries", g.count, w.deepest);
        std::printf("group of %5d triangles: %d stack entries at most\n", g.count, w.deepest);
    }
    std::printf(failures ? "%d FAILURES\n" : "tree check passed\n", failures);
    return failures ? 1 : 0;
}
