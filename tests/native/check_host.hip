// Host-only run of the library's scene staging and packing (no GPU needed), meant to be built with the
// address and undefined-behaviour sanitizers on the host side:
//     hipcc --offload-arch=gfx950 -O1 -g -ffp-contract=off -std=c++17 -Xarch_host -fsanitize=address
//           -Xarch_host -fsanitize=undefined -Xarch_host -fno-sanitize-recover=undefined
//           tests/native/check_host.hip -o <exe>
//     <exe> scene.bin ...
// Each scene file (written by tests/test_host_sanitizers.py from the package's flat scenes) goes
// through the host halves of wrt_add_* and of wrt_scene_commit, without and with the hierarchies;
// the record fetches of the three group loops are then replayed index by index and every one of them,
// the fetch ahead past a group's last record included, has to lie inside its array, and every region
// of the blob inside the blob.  Synthetic scenes follow: an empty one, odd counts, 33 large groups.
#include "../../raytracing-one-weekend_amd/csrc/wrt.hip"

#include <cstdio>
#include <random>

static int failures = 0;
#define EXPECT(cond, ...) do { if (!(cond)) { ++failures; std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); } } while (0)

struct HostGroup {
    int type = 0, count = 0, material = 0;
    std::vector<float> f[9];           // spheres: centres, radii; triangles: the nine arrays; disk: centre, normal, U, V
    std::vector<uint8_t> sphere_materials;
    float bounds_centre[3] = {0, 0, 0}, bounds_radius = 0;
    double radius = 0;
};

static bool read_exact(std::FILE* fh, void* dst, size_t bytes) { return bytes == 0 || std::fread(dst, 1, bytes, fh) == bytes; }

static bool read_scene(const char* path, std::vector<HostGroup>& groups) {
    std::FILE* fh = std::fopen(path, "rb");
    if (!fh) return false;
    int header[2];
    bool ok = read_exact(fh, header, sizeof(header)) && header[0] == 0x53545257 && header[1] >= 0 && header[1] <= WRT_MAX_GROUPS;
    for (int i = 0; ok && i < header[1]; ++i) {
        HostGroup g;
        int head[3];
        ok = read_exact(fh, head, sizeof(head)) && head[1] > 0 && head[1] <= (1 << 22);
        if (!ok) break;
        g.type = head[0]; g.count = head[1]; g.material = head[2];
        const size_t n = (size_t)g.count;
        if (g.type == GROUP_SPHERES) {
            g.f[0].resize(3 * n); g.f[1].resize(n); g.sphere_materials.resize(n);
            ok = read_exact(fh, g.f[0].data(), 12 * n) && read_exact(fh, g.f[1].data(), 4 * n) &&
                 read_exact(fh, g.sphere_materials.data(), n);
        } else if (g.type == GROUP_TRIANGLES) {
            for (int k = 0; ok && k < 9; ++k) {
                g.f[k].resize((k < 6 ? 3 : 2) * n);
                ok = read_exact(fh, g.f[k].data(), 4 * g.f[k].size());
            }
            ok = ok && read_exact(fh, g.bounds_centre, 12) && read_exact(fh, &g.bounds_radius, 4);
        } else if (g.type == GROUP_DISK) {
            for (int k = 0; ok && k < 4; ++k) { g.f[k].resize(3); ok = read_exact(fh, g.f[k].data(), 12); }
            ok = ok && read_exact(fh, &g.radius, 8);
        } else {
            ok = false;
        }
        if (ok) groups.push_back(std::move(g));
    }
    std::fclose(fh);
    return ok;
}

// the fetch of `n` f32x4 at `at` (f32x4 units from the array's start) must lie inside an array of `size`
static void fetch_inside(const char* what, int group, size_t at, size_t n, size_t size) {
    EXPECT(at + n <= size, "%s of group %d fetches [%zu, %zu) of %zu f32x4", what, group, at, at + n, size);
}

// loop_spheres' fetches: a quad (4 f32x4) at a time, one quad ahead
static void replay_sphere_loop(const wrt_ctx& ctx, int gi) {
    const GroupRec& g = ctx.h_groups[(size_t)gi];
    const size_t size = ctx.h_pairs.size();
    const int pairs = (g.count + 1) >> 1;
    size_t r = (size_t)g.pair_base;
    fetch_inside("sphere loop", gi, r, 4, size);
    int j = 0;
    for (; j + 3 < pairs; j += 4) {
        fetch_inside("sphere loop", gi, r + 4, 4, size);
        fetch_inside("sphere loop", gi, r + 8, 4, size);
        r += 8;
    }
    if (j < pairs) fetch_inside("sphere loop", gi, r + 4, 4, size);
}

// loop_triangles' fetches: a pair record is 16 floats and 2 more (5 f32x4 reserved), one record ahead
static void replay_triangle_loop(const wrt_ctx& ctx, int gi) {
    const GroupRec& g = ctx.h_groups[(size_t)gi];
    const size_t size = ctx.h_pairs.size();
    const int pairs = (g.count + 1) >> 1;
    size_t r = (size_t)g.pair_base;
    fetch_inside("triangle loop", gi, r, 5, size);
    int j = 0;
    for (; j + 1 < pairs; j += 2) {
        fetch_inside("triangle loop", gi, r + 5, 5, size);
        fetch_inside("triangle loop", gi, r + 10, 5, size);
        r += 10;
    }
}

// loop_triangles_rejecting's fetches: 4 f32x4 per two triangles, whole blocks of 16 triangles, one record ahead
static void replay_reject_loop(const wrt_ctx& ctx, int gi) {
    const GroupRec& g = ctx.h_groups[(size_t)gi];
    const size_t size = ctx.h_bounds.size();
    const int blocks = (g.count + 15) >> 4;
    size_t r = (size_t)g.bound_base;
    fetch_inside("reject loop", gi, r, 4, size);
    for (int block = 0; block < blocks; ++block)
        for (int quarter = 0; quarter < 4; ++quarter) {
            fetch_inside("reject loop", gi, r + 4, 4, size);
            fetch_inside("reject loop", gi, r + 8, 4, size);
            r += 8;
        }
}

static void check_scene(const char* name, const std::vector<HostGroup>& groups, int hierarchy, bool report = true) {
    wrt_ctx ctx;
    ctx.hierarchy = hierarchy;
    std::memset(ctx.h_textures, 0, sizeof(ctx.h_textures));
    reset_scene(&ctx);
    for (int i = 0; i < WRT_MAX_MATERIALS; ++i) ctx.h_materials[i].kind = WRT_MAT_DIFFUSE;
    for (const HostGroup& g : groups) {
        int id = -1, rc;
        if (g.type == GROUP_SPHERES) {
            rc = append_sphere_group(&ctx, g.f[0].data(), g.f[1].data(), g.sphere_materials.data(), g.count, &id);
        } else if (g.type == GROUP_TRIANGLES) {
            const float* const v[9] = {g.f[0].data(), g.f[1].data(), g.f[2].data(), g.f[3].data(), g.f[4].data(),
                                       g.f[5].data(), g.f[6].data(), g.f[7].data(), g.f[8].data()};
            rc = append_triangle_group(&ctx, v, g.count, g.material, g.bounds_centre, g.bounds_radius, &id);
        } else {
            rc = append_disk(&ctx, g.f[0].data(), g.f[1].data(), g.f[2].data(), g.f[3].data(), g.radius, g.material, &id);
        }
        EXPECT(rc == WRT_OK && id == (int)ctx.h_groups.size() - 1, "%s: adding a group failed: %s", name, ctx.err.c_str());
    }
    EXPECT(pack_scene(&ctx) == WRT_OK, "%s: pack_scene: %s", name, ctx.err.c_str());
    EXPECT(fetch_margins_hold(&ctx), "%s: fetch margins", name);
    size_t total = 0;
    EXPECT(lay_out_blob(&ctx, nullptr, nullptr, total) == WRT_OK && total > 0 && total % 256 == 0, "%s: blob size %zu", name, total);
    std::vector<unsigned char> host(total);
    unsigned char* const base = reinterpret_cast<unsigned char*>((uintptr_t)1 << 40);      // stands for a device address
    EXPECT(lay_out_blob(&ctx, base, host.data(), total) == WRT_OK, "%s: lay_out_blob: %s", name, ctx.err.c_str());
    // every region inside the blob, 256-byte aligned, in the order they were laid out
    const SceneRec& S = ctx.scene;
    const void* regions[] = {S.sph, S.sph_radius, S.sph_material, S.tri, S.tri_bz, S.tri_shade, S.groups, S.materials,
                             S.textures, S.cam, S.pairs, S.bounds, S.slot_group, S.tree};
    uintptr_t before = (uintptr_t)base;
    for (const void* p : regions) {
        const uintptr_t at = (uintptr_t)p;
        EXPECT(at >= before && at < (uintptr_t)base + total && at % 256 == 0, "%s: a blob region lies at %zu of %zu", name,
               (size_t)(at - (uintptr_t)base), total);
        before = at;
    }
    // the arrays' ends, with the 16 bytes a 16-byte scalar load of the last element's neighbour may touch
    EXPECT((uintptr_t)S.pairs + ctx.h_pairs.size() * 16 + 16 <= (uintptr_t)S.bounds, "%s: pair records run into the next region", name);
    EXPECT((uintptr_t)S.bounds + ctx.h_bounds.size() * 16 + 16 <= (uintptr_t)S.slot_group, "%s: bounding records run into the next region", name);
    EXPECT((uintptr_t)S.tree + ctx.h_tree.size() * 16 <= (uintptr_t)base + total, "%s: the hierarchies run past the blob", name);
    // the blob holds what was staged (spot check: the group records, the last pair record)
    if (!ctx.h_groups.empty())
        EXPECT(std::memcmp(host.data() + ((uintptr_t)S.groups - (uintptr_t)base), ctx.h_groups.data(),
                           ctx.h_groups.size() * sizeof(GroupRec)) == 0, "%s: group records differ in the blob", name);
    int slots = 0, large = 0;
    for (size_t i = 0; i < ctx.h_groups.size(); ++i) {
        const GroupRec& g = ctx.h_groups[i];
        if (g.type == GROUP_DISK) { EXPECT(g.slot < 0, "%s: a disk has a slot", name); continue; }
        // (k_render parks lanes at large groups without a slot and runs the same loops for them)
        if (g.type == GROUP_SPHERES) replay_sphere_loop(ctx, (int)i);
        if (g.type == GROUP_TRIANGLES) { replay_triangle_loop(ctx, (int)i); replay_reject_loop(ctx, (int)i); }
        if (g.count > CHEAP_GROUP_MAX) ++large;
        if (g.slot >= 0) {
            EXPECT(g.slot == slots && g.count > CHEAP_GROUP_MAX && ctx.h_slot_group[(size_t)g.slot] == (int)i, "%s: slot %d", name, g.slot);
            ++slots;
        }
        EXPECT(g.takers >= 8 && g.takers <= 64, "%s: group %zu waits for %d takers", name, i, g.takers);
        // the gathers of the exact evaluation and of shading stay inside the primitive arrays
        if (g.type == GROUP_SPHERES) EXPECT((size_t)g.first + g.count <= ctx.h_sph.size(), "%s: sphere range", name);
        if (g.type == GROUP_TRIANGLES)
            EXPECT(2 * ((size_t)g.first + g.count) <= ctx.h_tri.size() && 4 * ((size_t)g.first + g.count) <= ctx.h_tri_shade.size() &&
                   (size_t)g.first + g.count <= ctx.h_tri_bz.size(), "%s: triangle range", name);
        if (g.tree_base >= 0) EXPECT((size_t)g.tree_base < ctx.h_tree.size(), "%s: tree base", name);
    }
    EXPECT(slots == (large < 32 ? large : 32) && S.n_slots == slots, "%s: %d slots for %d large groups", name, slots, large);
    EXPECT(S.n_groups == (int)groups.size(), "%s: %d groups in the scene record", name, S.n_groups);
    if (report) std::printf("%-28s hierarchy %2d: %4zu groups, %2d slots, %8zu pair / %8zu bounding / %8zu tree f32x4, blob %zu bytes\n", name,
                hierarchy, groups.size(), slots, ctx.h_pairs.size(), ctx.h_bounds.size(), ctx.h_tree.size(), total);
}

static HostGroup random_spheres(int count, std::mt19937& rng) {
    HostGroup g;
    g.type = GROUP_SPHERES; g.count = count;
    std::uniform_real_distribution<float> u(-1.0f, 1.0f);
    for (int j = 0; j < count; ++j) {
        for (int a = 0; a < 3; ++a) g.f[0].push_back(5.0f * u(rng));
        g.f[1].push_back(0.2f + 0.1f * u(rng));
        g.sphere_materials.push_back((uint8_t)(j & 255));
    }
    return g;
}

static HostGroup random_triangles(int count, std::mt19937& rng) {
    HostGroup g;
    g.type = GROUP_TRIANGLES; g.count = count; g.material = 3;
    std::uniform_real_distribution<float> u(-1.0f, 1.0f);
    for (int j = 0; j < count; ++j) {
        for (int a = 0; a < 3; ++a) {
            g.f[0].push_back(2.0f * u(rng)); g.f[1].push_back(0.2f * u(rng)); g.f[2].push_back(0.2f * u(rng));
            for (int k = 3; k < 6; ++k) g.f[k].push_back(a == 1 ? 1.0f : 0.0f);
        }
        for (int k = 6; k < 9; ++k) { g.f[k].push_back(0.5f); g.f[k].push_back(0.25f); }
    }
    g.bounds_radius = 4.0f;
    return g;
}

int main(int argc, char** argv) {
    for (int i = 1; i < argc; ++i) {
        std::vector<HostGroup> groups;
        const bool ok = read_scene(argv[i], groups);
        EXPECT(ok, "cannot read %s", argv[i]);
        if (!ok) continue;
        const char* slash = std::strrchr(argv[i], '/');
        for (int hierarchy : {0, 16}) check_scene(slash ? slash + 1 : argv[i], groups, hierarchy);
    }
    std::mt19937 rng(11);
    for (int hierarchy : {0, 16}) {
        check_scene("empty scene", {}, hierarchy);
        // every remainder of the loops' unrolling: pairs mod 4 for spheres, triangles mod 16 and pairs mod 2
        for (int count = 1; count <= 40; ++count) {
            char name[64];
            std::snprintf(name, sizeof(name), "one group of %d", count);
            check_scene(name, {random_spheres(count, rng)}, hierarchy, false);
            check_scene(name, {random_triangles(count, rng)}, hierarchy, false);
            // the group whose fetch ahead has only the array's padding behind it, after another group
            check_scene(name, {random_triangles(33, rng), random_spheres(count, rng)}, hierarchy, false);
            check_scene(name, {random_spheres(33, rng), random_triangles(count, rng)}, hierarchy, false);
        }
        std::vector<HostGroup> many;
        for (int k = 0; k < 33; ++k) many.push_back(k % 3 == 0 ? random_triangles(9 + k, rng) : random_spheres(9 + k, rng));
        check_scene("33 large groups", many, hierarchy);
        check_scene("odd counts", {random_triangles(645, rng), random_spheres(17, rng), random_triangles(1025, rng)}, hierarchy);
    }
    std::printf(failures ? "%d FAILURES\n" : "host check passed\n", failures);
    return failures ? 1 : 0;
}
