// Device-side building blocks of the render path (gfx950).
//
// Arithmetic contract: this translation unit is compiled with
// -ffp-contract=off and every expression below is written in the reference's
// own association order, so float32 results are the ones numpy produces
// (numpy's einsum sums three float32 products left to right without fused
// multiply-adds; its float64 contiguous kernel sums (p0 + p2) + p1).  Fast
// approximations (v_rcp_f32) are used only as conservative pre-filters that
// decide whether the exact evaluation has to run for a wavefront.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Wave-uniform read-only data is fetched through the constant address space
// so that hipcc selects scalar (s_load) instructions and the values arrive
// in SGPRs, usable directly as VALU operands.
#define WRT_CONST __attribute__((address_space(4)))
template <typename T>
__device__ __forceinline__ const WRT_CONST T* as_uniform(const T* p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    return (const WRT_CONST T*)p;
#pragma clang diagnostic pop
}

enum { GROUP_SPHERES = 0, GROUP_TRIANGLES = 1, GROUP_DISK = 2 };
enum { PURPOSE_JITTER = 0, PURPOSE_LENS = 1, PURPOSE_UNITVEC = 2, PURPOSE_SCHLICK = 3 };

struct GroupRec {            // 80 bytes, uniform
    int type, first, count, material;
    float c[4];              // triangles: bounds centre, bounds_radius^2 | disk: centre, radius^2
    float n[4];              // disk: unit normal
    float u[4];              // disk: U / radius
    float v[4];              // disk: V / radius
};

struct MaterialRec {         // per material index, gathered per lane
    int kind, texture;
    float colour_a[3];
    float colour_b[3];
    double scale[3];
    double offset[3];
    float fuzz;              // float32(fuzziness): multiplies a float32 vector (materials.py:274)
    int fuzzy;               // fuzziness > 0.0001 evaluated on the Python float
    float ior;               // numpy.full(n, ior, dtype=single) (materials.py:319)
    int pad;
};

struct TextureRec {
    const float* rgb;        // [h][w][3]
    int w, h;
    double side_minus_one;   // min(w, h) - 1.0 (materials.py:92,126)
};

struct CameraRec {
    double pos[3], U[3], V[3], bl[3], horiz[3], vert[3];
    double lens_radius;
};

struct SceneRec {
    const f32x4* sph;          // (cx, cy, cz, r*r) per sphere, all sphere groups concatenated
    const float* sph_radius;
    const uint8_t* sph_material;
    const f32x4* tri;          // 3 x f32x4 per triangle: (P0, A.x) (A.yz, B.xy) (B.z, 0, 0, 0)
    const f32x4* tri_shade;    // 4 x f32x4 per triangle: n0 n1 n2 (9), uv0 uv1 uv2 (6), pad
    const GroupRec* groups;
    const MaterialRec* materials;
    const TextureRec* textures;   // 16 slots
    const CameraRec* cam;
    float horizon[3], zenith[3];
    int n_groups;
};

// ------------------------------------------------------------------ RNG
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

struct PathKey { uint32_t x, y, s, k0, k1; };

__device__ __forceinline__ void keyed_words(const PathKey& k, uint32_t purpose, uint32_t bounce,
                                            uint32_t attempt, uint32_t out[4]) {
    philox4x32_10(k.x, k.y, k.s, purpose | (bounce << 4) | (attempt << 12), k.k0, k.k1, out);
}

__device__ __forceinline__ double unit_interval(uint32_t r) { return (double)r * 2.3283064365386963e-10; }

// float64 three-term dot in numpy's contiguous einsum order
__device__ __forceinline__ double dot64(double a0, double a1, double a2, double b0, double b1, double b2) {
    return (a0 * b0 + a2 * b2) + a1 * b1;
}

// numpy_random_unit_vecs (materials.py:401-431) for one path
__device__ __forceinline__ void keyed_unit_vector(const PathKey& k, uint32_t bounce, float& vx, float& vy, float& vz) {
    uint32_t r[4];
    float x, y, z, len2;
    uint32_t attempt = 0;
    while (true) {
        keyed_words(k, PURPOSE_UNITVEC, bounce, attempt, r);
        x = (float)(-1.0 + 2.0 * unit_interval(r[0]));
        y = (float)(-1.0 + 2.0 * unit_interval(r[1]));
        z = (float)(-1.0 + 2.0 * unit_interval(r[2]));
        len2 = (x * x + y * y) + z * z;
        if (!(len2 > 1.0f || len2 < 0.00001f)) break;
        ++attempt;
    }
    const float len = sqrtf(len2);
    vx = x / len; vy = y / len; vz = z / len;
}

// ------------------------------------------------------------------ camera
// Camera.get_ray_components (camera.py:108-190) for one (x, y, sample)
__device__ __forceinline__ void camera_ray(const CameraRec& C, int width, int height, const PathKey& k,
                                           float o[3], float d[3]) {
    uint32_t r[4];
    keyed_words(k, PURPOSE_JITTER, 0, 0, r);
    const float vu = (float)((double)(float)k.x + unit_interval(r[0])) / (float)width;
    const float vv = (float)((double)(float)k.y + unit_interval(r[1])) / (float)height;
    double lx = 0.0, ly = 0.0;
    if (C.lens_radius != 0.0) {
        uint32_t attempt = 0;
        while (true) {
            keyed_words(k, PURPOSE_LENS, 0, attempt, r);
            lx = -1.0 + 2.0 * unit_interval(r[0]);
            ly = -1.0 + 2.0 * unit_interval(r[1]);
            if (!(lx * lx + ly * ly > 1.0)) break;
            ++attempt;
        }
    }
    const double offx = C.lens_radius * lx, offy = C.lens_radius * ly;
    double O[3], D[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double oriented = offx * C.U[a] + offy * C.V[a];
        const double on_plane = (C.bl[a] + C.horiz[a] * (double)vu) + C.vert[a] * (double)vv;
        O[a] = oriented + C.pos[a];
        D[a] = on_plane - O[a];
    }
    const double len = sqrt((D[0] * D[0] + D[2] * D[2]) + D[1] * D[1]);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        o[a] = (float)O[a];
        d[a] = (float)(D[a] / len);
    }
}

// ------------------------------------------------------------------ closest hit
struct Closest {
    float t;      // running best (starts at the reference's sentinel t_max + 1)
    int group;    // group index of the best candidate, -1 = none
    int prim;     // primitive index inside the concatenated sphere / triangle arrays
};

struct PairCounts { unsigned long long sphere, triangle, cull, disk; };

// One ray against groups [g_begin, g_end).  Every lane of the wavefront must
// call this (wave-level votes inside); `live` masks lanes without a ray.
// Within a group the lowest primitive index wins equal t (numpy.argmin,
// sphere_group.py:213, triangle_group.py:399); across groups the earlier
// group wins (strict <, main.py:144).
__device__ __forceinline__ void closest_hit(const SceneRec& S, int g_begin, int g_end, bool live,
                                            float ox, float oy, float oz, float dx, float dy, float dz,
                                            float t_min, float t_max, Closest& best, PairCounts& pc) {
    const float sentinel = t_max + 1.0f;
    const float t_lo = t_min * 0.99999f, t_hi = t_max * 1.00001f;
    best.t = sentinel; best.group = -1; best.prim = -1;
    const WRT_CONST GroupRec* groups = as_uniform(S.groups);
    const WRT_CONST f32x4* sph = as_uniform(S.sph);
    const WRT_CONST f32x4* tri = as_uniform(S.tri);
    const unsigned long long live_mask = __ballot(live);
    if (live_mask == 0ull) return;
    const unsigned n_live = (unsigned)__popcll(live_mask);
    for (int g = g_begin; g < g_end; ++g) {
        const int type = groups[g].type, first = groups[g].first, count = groups[g].count;
        if (type == GROUP_SPHERES) {
            // sphere_group.py:137-205
            pc.sphere += (unsigned long long)n_live * (unsigned)count;
            for (int j = 0; j < count; ++j) {
                const f32x4 s = sph[first + j];
                const float cox = ox - s.x, coy = oy - s.y, coz = oz - s.z;
                const float h = (dx * cox + dy * coy) + dz * coz;
                const float c = ((cox * cox + coy * coy) + coz * coz) - s.w;
                const float disc = h * h - c;
                const bool ok = live && (disc > 0.00001f);
                if (__ballot(ok) != 0ull) {
                    const float root = sqrtf(disc);
                    float t_near = -h - root, t_far = -h + root;
                    t_near = (t_near < t_min) ? sentinel : t_near;
                    t_far = (t_far < t_min) ? sentinel : t_far;
                    const float t = fminf(t_near, t_far);
                    if (ok && t < best.t) { best.t = t; best.group = g; best.prim = first + j; }
                }
            }
        } else if (type == GROUP_TRIANGLES) {
            // bounding-sphere pre-cull, triangle_group.py:158-166
            const float bx = groups[g].c[0], by = groups[g].c[1], bz = groups[g].c[2], br2 = groups[g].c[3];
            const float cox = ox - bx, coy = oy - by, coz = oz - bz;
            const float hh = (dx * cox + dy * coy) + dz * coz;
            const float cc = ((cox * cox + coy * coy) + coz * coz) - br2;
            const bool pass = live && ((hh * hh - cc) > 0.0001f);
            const unsigned long long pass_mask = __ballot(pass);
            pc.cull += n_live;
            if (pass_mask == 0ull) continue;
            pc.triangle += (unsigned long long)__popcll(pass_mask) * (unsigned)count;
            // Moller-Trumbore exactly as triangle_group.py:289-395 evaluates it
            float t_cap = fminf(t_hi, best.t * 1.00001f);
            for (int j = 0; j < count; ++j) {
                const int base = 3 * (first + j);
                const f32x4 r0 = tri[base], r1 = tri[base + 1], r2 = tri[base + 2];
                const float P0x = r0.x, P0y = r0.y, P0z = r0.z;
                const float Ax = r0.w, Ay = r1.x, Az = r1.y;
                const float Bx = r1.z, By = r1.w, Bz = r2.x;
                const float p0 = dy * Bz - dz * By, p1 = dz * Bx - dx * Bz, p2 = dx * By - dy * Bx;
                const float det = (Ax * p0 + Ay * p1) + Az * p2;
                const float tx = ox - P0x, ty = oy - P0y, tz = oz - P0z;
                const float un = (tx * p0 + ty * p1) + tz * p2;
                const float q0 = ty * Az - tz * Ay, q1 = tz * Ax - tx * Az, q2 = tx * Ay - ty * Ax;
                const float vn = (dx * q0 + dy * q1) + dz * q2;
                const float tn = (Bx * q0 + By * q1) + Bz * q2;
                // conservative filter: 1-ulp reciprocal, widened bounds
                const float ri = __builtin_amdgcn_rcpf(det);
                const float ua = un * ri, va = vn * ri, ta = tn * ri;
                const float inside = fminf(fminf(ua, va), 1.00001f - (ua + va));
                const bool maybe = pass && (inside >= -0.00001f) && (ta >= t_lo) && (ta <= t_cap);
                if (__ballot(maybe) != 0ull) {
                    const bool parallel = fabsf(det) < 0.00001f;
                    const float inv = 1.0f / (parallel ? 1.0f : det);
                    const float u = un * inv, v = vn * inv, t = tn * inv;
                    const bool miss = parallel | (u > 1.0f) | (u < 0.0f) | (v < 0.0f) | ((u + v) > 1.0f)
                                      | (t < t_min) | (t > t_max);
                    if (maybe && !miss && t < best.t) {
                        best.t = t; best.group = g; best.prim = first + j;
                        t_cap = fminf(t_hi, t * 1.00001f);
                    }
                }
            }
        } else {
            // disk.py:48-69
            pc.disk += n_live;
            const float cx = groups[g].c[0], cy = groups[g].c[1], cz = groups[g].c[2], r2 = groups[g].c[3];
            const float nx = groups[g].n[0], ny = groups[g].n[1], nz = groups[g].n[2];
            const float ddn = (dx * nx + dy * ny) + dz * nz;
            const float num = ((cx - ox) * nx + (cy - oy) * ny) + (cz - oz) * nz;
            bool ok = live && (fabsf(ddn) > 0.0001f);
            const float t = ok ? num / ddn : sentinel;
            ok = ok && (t > t_min) && (t < t_max);
            const float rx = (ox + dx * t) - cx, ry = (oy + dy * t) - cy, rz = (oz + dz * t) - cz;
            ok = ok && (((rx * rx + ry * ry) + rz * rz) < r2);
            if (ok && t < best.t) { best.t = t; best.group = g; best.prim = 0; }
        }
    }
}

struct HitRecord {
    float p[3], n[3], uv[2];
    int material;
    bool back;
};

// Hit point, shading normal, uv, facing and material of the winning
// primitive (sphere_group.py:225-259, triangle_group.py:414-450, disk.py:65-92).
__device__ __forceinline__ void shade_record(const SceneRec& S, const Closest& best,
                                             float ox, float oy, float oz, float dx, float dy, float dz,
                                             HitRecord& h) {
    const GroupRec& G = S.groups[best.group];
    const float t = best.t;
    h.p[0] = ox + dx * t; h.p[1] = oy + dy * t; h.p[2] = oz + dz * t;
    if (G.type == GROUP_SPHERES) {
        const f32x4 s = S.sph[best.prim];
        const float r = S.sph_radius[best.prim];
        float nx = (h.p[0] - s.x) / r, ny = (h.p[1] - s.y) / r, nz = (h.p[2] - s.z) / r;
        h.back = ((nx * dx + ny * dy) + nz * dz) > 0.0f;
        if (h.back) { nx *= -1.0f; ny *= -1.0f; nz *= -1.0f; }
        h.n[0] = nx; h.n[1] = ny; h.n[2] = nz;
        h.uv[0] = 0.0f; h.uv[1] = 0.0f;
        h.material = S.sph_material[best.prim];
    } else if (G.type == GROUP_TRIANGLES) {
        const int base = 3 * best.prim;
        const f32x4 r0 = S.tri[base], r1 = S.tri[base + 1], r2 = S.tri[base + 2];
        const float Ax = r0.w, Ay = r1.x, Az = r1.y, Bx = r1.z, By = r1.w, Bz = r2.x;
        const float p0 = dy * Bz - dz * By, p1 = dz * Bx - dx * Bz, p2 = dx * By - dy * Bx;
        const float det = (Ax * p0 + Ay * p1) + Az * p2;
        const float inv = 1.0f / det;
        const float tx = ox - r0.x, ty = oy - r0.y, tz = oz - r0.z;
        const float u = ((tx * p0 + ty * p1) + tz * p2) * inv;
        const float q0 = ty * Az - tz * Ay, q1 = tz * Ax - tx * Az, q2 = tx * Ay - ty * Ax;
        const float v = ((dx * q0 + dy * q1) + dz * q2) * inv;
        const float w = (1.0f - u) - v;
        const f32x4 s0 = S.tri_shade[4 * best.prim], s1 = S.tri_shade[4 * best.prim + 1],
                    s2 = S.tri_shade[4 * best.prim + 2], s3 = S.tri_shade[4 * best.prim + 3];
        // n0 = s0.xyz, n1 = (s0.w, s1.x, s1.y), n2 = (s1.z, s1.w, s2.x)
        float nx = (s0.x * w + s0.w * u) + s1.z * v;
        float ny = (s0.y * w + s1.x * u) + s1.w * v;
        float nz = (s0.z * w + s1.y * u) + s2.x * v;
        const float len = sqrtf((nx * nx + ny * ny) + nz * nz);
        nx /= len; ny /= len; nz /= len;
        // uv0 = (s2.y, s2.z), uv1 = (s2.w, s3.x), uv2 = (s3.y, s3.z)
        h.uv[0] = (s2.y * w + s2.w * u) + s3.y * v;
        h.uv[1] = (s2.z * w + s3.x * u) + s3.z * v;
        h.back = det < 0.0f;
        if (h.back) { nx *= -1.0f; ny *= -1.0f; nz *= -1.0f; }
        h.n[0] = nx; h.n[1] = ny; h.n[2] = nz;
        h.material = G.material;
    } else {
        float nx = G.n[0], ny = G.n[1], nz = G.n[2];
        const float rx = h.p[0] - G.c[0], ry = h.p[1] - G.c[1], rz = h.p[2] - G.c[2];
        h.uv[0] = (((rx * G.u[0] + ry * G.u[1]) + rz * G.u[2]) + 1.0f) / 2.0f;
        h.uv[1] = (((rx * G.v[0] + ry * G.v[1]) + rz * G.v[2]) + 1.0f) / 2.0f;
        h.back = ((dx * nx + dy * ny) + dz * nz) > 0.0f;
        if (h.back) { nx *= -1.0f; ny *= -1.0f; nz *= -1.0f; }
        h.n[0] = nx; h.n[1] = ny; h.n[2] = nz;
        h.material = G.material;
    }
}

// ------------------------------------------------------------------ materials
struct Scattered {
    float d[3];       // new direction, as stored back into the float32 ray array (main.py:198)
    float col[3];     // colour of this bounce (main.py:199)
    bool absorbed;
};

__device__ __forceinline__ bool odd_cell(double value) {
    // numpy.remainder(numpy.fabs(numpy.floor(value)), 2) != 0 (materials.py:171-174)
    const double f = fabs(floor(value));
    return (f - 2.0 * floor(f * 0.5)) != 0.0;
}

// <Material>.scatter (materials.py) for one hit, float64 where main.render()'s
// float64 hit arrays make the reference evaluate in float64.
__device__ __forceinline__ void scatter_one(const SceneRec& S, const MaterialRec& M, const PathKey& key,
                                            uint32_t bounce, float dx, float dy, float dz,
                                            const HitRecord& h, Scattered& out) {
    const double nx = h.n[0], ny = h.n[1], nz = h.n[2];
    out.absorbed = false;
    if (M.kind >= 1 && M.kind <= 5) {
        float vx, vy, vz;
        keyed_unit_vector(key, bounce, vx, vy, vz);
        if (dot64(vx, vy, vz, nx, ny, nz) < 0.0) { vx *= -1.0f; vy *= -1.0f; vz *= -1.0f; }
        out.d[0] = vx; out.d[1] = vy; out.d[2] = vz;
        if (M.kind == 1) {
            out.col[0] = M.colour_a[0]; out.col[1] = M.colour_a[1]; out.col[2] = M.colour_a[2];
        } else if (M.kind == 2) {
            const TextureRec T = S.textures[M.texture];
            const double cu = fmin(fmax((double)h.uv[0], 0.0), 1.0) * T.side_minus_one;
            const double cv = fmin(fmax((double)h.uv[1], 0.0), 1.0) * T.side_minus_one;
            const int iu = (int)cu, iv = (int)cv;
            const float* px = T.rgb + ((size_t)iv * (size_t)T.w + (size_t)iu) * 3;
            out.col[0] = px[0]; out.col[1] = px[1]; out.col[2] = px[2];
        } else if (M.kind == 3) {
            const bool X = odd_cell((double)h.p[0] * M.scale[0] + M.offset[0]);
            const bool Y = odd_cell((double)h.p[1] * M.scale[1] + M.offset[1]);
            const bool Z = odd_cell((double)h.p[2] * M.scale[2] + M.offset[2]);
            const bool pick = X != (Y != Z);
#pragma unroll
            for (int c = 0; c < 3; ++c) out.col[c] = pick ? M.colour_a[c] : M.colour_b[c];
        } else if (M.kind == 4) {
            out.col[0] = (float)((nx + 1.0) * 0.5);
            out.col[1] = (float)((ny + 1.0) * 0.5);
            out.col[2] = (float)((nz + 1.0) * 0.5);
        } else {
            float r = 0.4f, g = 0.4f, b = 0.4f;                 // AXIS_COLOUR_PAIRS order, materials.py:14-32
            if (nx > 0.8) { r = 1.0f; g = 0.0f; b = 0.0f; }
            if (ny > 0.8) { r = 0.0f; g = 1.0f; b = 0.0f; }
            if (nz > 0.8) { r = 0.0f; g = 0.0f; b = 1.0f; }
            if (-nx > 0.8) { r = 1.0f; g = 0.0f; b = 1.0f; }
            if (-ny > 0.8) { r = 1.0f; g = 1.0f; b = 0.0f; }
            if (-nz > 0.8) { r = 0.0f; g = 1.0f; b = 1.0f; }
            out.col[0] = r; out.col[1] = g; out.col[2] = b;
        }
        return;
    }
    if (M.kind == 6) {
        // numpy_reflect (materials.py:457): d - (n * 2.0 * (d . n))
        const double dn = dot64(dx, dy, dz, nx, ny, nz);
        double rx = (double)dx - (nx * 2.0) * dn, ry = (double)dy - (ny * 2.0) * dn, rz = (double)dz - (nz * 2.0) * dn;
        out.col[0] = M.colour_a[0]; out.col[1] = M.colour_a[1]; out.col[2] = M.colour_a[2];
        if (M.fuzzy) {
            float vx, vy, vz;
            keyed_unit_vector(key, bounce, vx, vy, vz);
            rx += (double)(vx * M.fuzz); ry += (double)(vy * M.fuzz); rz += (double)(vz * M.fuzz);
            const double len = sqrt(dot64(rx, ry, rz, rx, ry, rz));
            rx /= len; ry /= len; rz /= len;
            if (dot64(rx, ry, rz, nx, ny, nz) < 0.00001) {
                out.col[0] = 0.0f; out.col[1] = 0.0f; out.col[2] = 0.0f;
                out.absorbed = true;
            }
        }
        out.d[0] = (float)rx; out.d[1] = (float)ry; out.d[2] = (float)rz;
        return;
    }
    if (M.kind == 7) {
        // Dielectric.scatter / reflectance / refract (materials.py:314-398)
        const float ratio = h.back ? M.ior : 1.0f / M.ior;
        const double cos_t = fmin(dot64(-1.0 * dx, -1.0 * dy, -1.0 * dz, nx, ny, nz), 1.0);
        const double sin_t = sqrt(1.0 - cos_t * cos_t);
        const bool cannot = ((double)ratio * sin_t) > 1.0;
        float r0 = (1.0f - ratio) / (1.0f + ratio);
        r0 = r0 * r0;
        const double m = 1.0 - cos_t, m2 = m * m;
        const double reflectance = (double)r0 + (double)(1.0f - r0) * (m2 * m2 * m);
        uint32_t r[4];
        keyed_words(key, PURPOSE_SCHLICK, bounce, 0, r);
        const bool reflect = cannot || (reflectance > unit_interval(r[0]));
        double ox_, oy_, oz_;
        if (reflect) {
            const double dn = dot64(dx, dy, dz, nx, ny, nz);
            ox_ = (double)dx - (nx * 2.0) * dn; oy_ = (double)dy - (ny * 2.0) * dn; oz_ = (double)dz - (nz * 2.0) * dn;
        } else {
            const double px = (double)ratio * ((double)dx + cos_t * nx);
            const double py = (double)ratio * ((double)dy + cos_t * ny);
            const double pz = (double)ratio * ((double)dz + cos_t * nz);
            const double par = -1.0 * sqrt(fabs(1.0 - dot64(px, py, pz, px, py, pz)));
            ox_ = px + par * nx; oy_ = py + par * ny; oz_ = pz + par * nz;
        }
        out.d[0] = (float)ox_; out.d[1] = (float)oy_; out.d[2] = (float)oz_;
        out.col[0] = 1.0f; out.col[1] = 1.0f; out.col[2] = 1.0f;
        return;
    }
    // WRT_MAT_NONE: handled by the caller (ray continues unchanged)
    out.d[0] = dx; out.d[1] = dy; out.d[2] = dz;
    out.col[0] = 1.0f; out.col[1] = 1.0f; out.col[2] = 1.0f;
}
