// Device-side building blocks of the render path (gfx950).
//
// Arithmetic contract: this translation unit is compiled with
// -ffp-contract=off and every expression below is written in the reference's
// own association order, so float32 results are the ones numpy produces
// (numpy's einsum sums three float32 products left to right without fused
// multiply-adds; its float64 contiguous kernel sums (p0 + p2) + p1).  Those
// functions (sphere_discriminant, sphere_root, triangle_terms, triangle_exact,
// cull_pass, disk_hit, shade_record, scatter_one) produce every number that
// reaches a result.  Fused multiply-adds and fast approximations (v_rcp_f32)
// appear only in the conservative tests of the group loops, which decide
// which (ray, primitive) pairs those functions have to look at.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

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

struct GroupRec {            // 144 bytes, uniform
    int type, first, count, material;
    int pair_base;           // first f32x4 of this group's component-interleaved pair records
    int bound_base;          // triangles: first f32x4 of the per-triangle bounding records (two triangles per 4 x f32x4)
    float edge_max;          // triangles: largest |component| of any edge vector A, B of the group
    float reach;             // triangles: largest |coordinate - c| of any corner or bounding-sphere centre
    int slot;                // large groups: bit of the render kernel's pending mask, -1 = none
    float reject_eps;        // triangles: eps of the reject test, as used for the bounding records' k
    int tree_base;           // triangles: first f32x4 of the group's box hierarchy, -1 = none
    float reach2;            // triangles with a hierarchy: largest distance of any corner from c
    int takers;              // large groups: lanes the render kernel waits for before it runs the group's loop
    float tree_rmax;         // spheres with a hierarchy: largest radius in the tree
    int n_outliers;          // spheres with a hierarchy: spheres kept out of the tree (much larger than the rest)
    int tree_leaf_shift;     // groups with a hierarchy: a leaf holds up to 1 << this many primitives (2 or 3)
    int outlier[4];          // their indices inside the group
    float c[4];              // triangles: bounds centre, bounds_radius^2 | disk: centre, radius^2 | spheres: median centre
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
    const f32x4* tri;          // 2 x f32x4 per triangle: (P0, A.x) (A.yz, B.xy)
    const float* tri_bz;       // B.z per triangle (kept apart so that no loaded register is unused)
    const f32x4* tri_shade;    // 4 x f32x4 per triangle: n0 n1 n2 (9), uv0 uv1 uv2 (6), pad
    const GroupRec* groups;
    const MaterialRec* materials;
    const TextureRec* textures;   // 16 slots
    const CameraRec* cam;
    const f32x4* pairs;           // sphere and triangle groups again, two primitives per record (group loops)
    const f32x4* bounds;          // per triangle: bounding sphere and B x A, two triangles per record
    int prefilter;                // 1: triangle loops reject by bounding sphere first
    const f32x4* tree;            // box hierarchies of the large triangle groups: nodes, then leaf triangles
    int hierarchy;                // n > 0: triangle groups of n or more triangles that have a hierarchy are walked through it
    const int* slot_group;        // [0, 32): group index of each slot, [32, 64): takers its loop waits for,
                                  // [64, 96): the same with the hierarchy, [96, 128): count if it has one, else -1
    int n_slots;
    float horizon[3], zenith[3];
    int n_groups;
};

// The scene record is read where it lies, in the kernel-argument segment,
// through scalar loads at the point of use: a persistent kernel that kept
// its two dozen pointers in SGPRs from start to end would spill most of them.
typedef const WRT_CONST SceneRec& SceneArg;

// ------------------------------------------------------------------ RNG
// Philox-4x32 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) with
// seven rounds, the fewest that paper's authors found to pass the whole of BigCrush (ten is their
// default, with a safety margin this renderer's few draws per path do not need): three rounds fewer
// are 30 % fewer of the generator's instructions, which are about a tenth of a sphere scene's frame.
#ifndef WRT_PHILOX_ROUNDS
#define WRT_PHILOX_ROUNDS 7        // (another number is for measurements only: the fixtures and the oracle use seven)
#endif
enum { PHILOX_ROUNDS = WRT_PHILOX_ROUNDS };
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                           uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int i = 0; i < PHILOX_ROUNDS; ++i) {
        // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a high and a low one
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

struct PathKey { uint32_t x, y, s, k0, k1; };

__device__ __forceinline__ void keyed_words(const PathKey& k, uint32_t purpose, uint32_t bounce,
                                            uint32_t attempt, uint32_t out[4]) {
    philox4x32(k.x, k.y, k.s, purpose | (bounce << 4) | (attempt << 12), k.k0, k.k1, out);
}

__device__ __forceinline__ double unit_interval(uint32_t r) { return (double)r * 2.3283064365386963e-10; }

// float64 three-term dot in numpy's contiguous einsum order
__device__ __forceinline__ double dot64(double a0, double a1, double a2, double b0, double b1, double b2) {
    return (a0 * b0 + a2 * b2) + a1 * b1;
}

// Candidates of the two rejection loops.  A wavefront pays for its unluckiest lane in such a loop
// (about 6.5 rounds for the unit vector where a lane alone needs 1.9), so a round is made cheap:
// one generator call yields two candidates.
//
// Unit-sphere candidate number a (a = 0, 1, ...) comes from call a / 2 of the path's UNITVEC
// stream: three 21-bit integers v, each coordinate -1 + v * 2^-20 (exact in float32, which is what
// the reference's float64 draw cast to float32 gives for such a value, materials.py:408).  The
// even candidate takes the top 21 bits of w0, w1, w2; the odd one the low 11 bits of each, extended
// by 10 bits of w3 (bits 22.., 12.., 2..).
__device__ __forceinline__ void unit_candidates(const uint32_t w[4], float even[3], float odd[3]) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const uint32_t hi = w[a] >> 11;
        const uint32_t lo = __builtin_amdgcn_alignbit(w[a], w[3] << (10 * a), 22) & 0x1fffffu;
        even[a] = __builtin_fmaf((float)hi, 9.5367431640625e-07f, -1.0f);
        odd[a] = __builtin_fmaf((float)lo, 9.5367431640625e-07f, -1.0f);
    }
}

// The first acceptable of the two candidates of one UNITVEC call (materials.py:417-420: rejected
// when the squared length is above 1 or below 1e-5); -> accepted
__device__ __forceinline__ bool accept_unit_candidate(const uint32_t w[4], float& x, float& y, float& z, float& len2) {
    float even[3], odd[3];
    unit_candidates(w, even, odd);
    const float le = (even[0] * even[0] + even[1] * even[1]) + even[2] * even[2];
    const float lo = (odd[0] * odd[0] + odd[1] * odd[1]) + odd[2] * odd[2];
    const bool ok_even = !(le > 1.0f || le < 0.00001f), ok_odd = !(lo > 1.0f || lo < 0.00001f);
    x = ok_even ? even[0] : odd[0]; y = ok_even ? even[1] : odd[1]; z = ok_even ? even[2] : odd[2];
    len2 = ok_even ? le : lo;
    return ok_even || ok_odd;
}

// numpy_random_unit_vecs (materials.py:401-431) for the paths of a wavefront that want a vector.
// Must be called by the whole wavefront; lanes with `want` false get an unspecified vector.
//
// A path takes the first acceptable candidate of its own stream, whichever lane computes it.  Round
// 0: every lane looks at call 0 of its own path (candidates 0 and 1; about a quarter of the lanes
// fail both).  After that the lanes still without a vector are few, and a round in which each of them
// tried one more call alone would cost the whole wavefront a generator call for a handful of lanes.
// Instead the wavefront's lanes are dealt out among them, `per` lanes (a power of two) to each:
// helper j of a pending path computes call next + j of THAT path's stream (the path's x, y, sample
// and bounce come over by ds_bpermute), and the path takes the result of the lowest call that has
// an acceptable candidate - the same candidate its own loop would have stopped at.  With four
// helpers a path is still pending after the round once in 350 times.
__device__ __forceinline__ void wave_unit_vectors(const PathKey& k, uint32_t bounce, bool want, float& vx, float& vy,
                                                  float& vz) {
    float x = 0.0f, y = 0.0f, z = 1.0f, len2 = 1.0f;
    bool pending = false;
    if (want) {
        uint32_t w[4];
        keyed_words(k, PURPOSE_UNITVEC, bounce, 0, w);
        pending = !accept_unit_candidate(w, x, y, z, len2);
    }
    const unsigned lane = threadIdx.x & 63u;
    uint32_t next_call = 1;                                   // the first call no pending path has looked at
    for (unsigned long long open = __ballot(pending); open != 0ull; open = __ballot(pending)) {
        const int n_open = (int)__popcll(open);
        const int shift = n_open > 32 ? 0 : n_open > 16 ? 1 : n_open > 8 ? 2 : n_open > 4 ? 3 : 4;   // per <= 16
        const unsigned per_mask = (1u << shift) - 1u;
        // the pending lane of rank r tells lane r << shift who it is (all others send to lane 63, which is
        // no rank's first helper unless every lane is pending, and then nobody else sends); that lane's
        // group reads it from there
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(open >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)open, 0u));
        const int told = __builtin_amdgcn_ds_permute((pending ? (rank << shift) : 63) * 4, (int)lane);
        const int owner4 = __builtin_amdgcn_ds_bpermute((int)(lane & ~per_mask) * 4, told) * 4;
        const bool helping = (int)(lane >> shift) < n_open;
        PathKey other = k;                                    // (the key words are the same for every path of a launch)
        other.x = (uint32_t)__builtin_amdgcn_ds_bpermute(owner4, (int)k.x);
        other.y = (uint32_t)__builtin_amdgcn_ds_bpermute(owner4, (int)k.y);
        other.s = (uint32_t)__builtin_amdgcn_ds_bpermute(owner4, (int)k.s);
        const uint32_t other_bounce = (uint32_t)__builtin_amdgcn_ds_bpermute(owner4, (int)bounce);
        uint32_t w[4];
        keyed_words(other, PURPOSE_UNITVEC, other_bounce, next_call + (lane & per_mask), w);
        float hx, hy, hz, hl;
        const bool found = accept_unit_candidate(w, hx, hy, hz, hl) && helping;
        // each pending path takes the lowest of its helpers that found one
        const unsigned long long finds = __ballot(found);
        const unsigned long long mine = (finds >> (rank << shift)) & ((2ull << per_mask) - 1ull);
        const bool got = pending && mine != 0ull;
        const int from4 = (got ? (rank << shift) + (int)__builtin_ctzll(mine) : (int)lane) * 4;
        const float tx = __int_as_float(__builtin_amdgcn_ds_bpermute(from4, __float_as_int(hx)));
        const float ty = __int_as_float(__builtin_amdgcn_ds_bpermute(from4, __float_as_int(hy)));
        const float tz = __int_as_float(__builtin_amdgcn_ds_bpermute(from4, __float_as_int(hz)));
        const float tl = __int_as_float(__builtin_amdgcn_ds_bpermute(from4, __float_as_int(hl)));
        if (got) { x = tx; y = ty; z = tz; len2 = tl; pending = false; }
        next_call += per_mask + 1u;
    }
    const float len = sqrtf(len2);
    vx = x / len; vy = y / len; vz = z / len;
}

// ------------------------------------------------------------------ camera
// Camera.get_ray_components (camera.py:108-190) for one (x, y, sample).  The path's JITTER call
// gives the sub-pixel offset (w0, w1) and lens-disk candidate 0 (w2, w3); candidates 1 + 2c and
// 2 + 2c are (w0, w1) and (w2, w3) of call c of its LENS stream.
__device__ __forceinline__ void camera_ray(const CameraRec& C, int width, int height, const PathKey& k,
                                           float o[3], float d[3]) {
    uint32_t r[4];
    keyed_words(k, PURPOSE_JITTER, 0, 0, r);
    const float vu = (float)((double)(float)k.x + unit_interval(r[0])) / (float)width;
    const float vv = (float)((double)(float)k.y + unit_interval(r[1])) / (float)height;
    double lx = 0.0, ly = 0.0;
    if (C.lens_radius != 0.0) {
        lx = -1.0 + 2.0 * unit_interval(r[2]);
        ly = -1.0 + 2.0 * unit_interval(r[3]);
        uint32_t call = 0;
        while (lx * lx + ly * ly > 1.0) {
            keyed_words(k, PURPOSE_LENS, 0, call, r);
            lx = -1.0 + 2.0 * unit_interval(r[0]);
            ly = -1.0 + 2.0 * unit_interval(r[1]);
            if (!(lx * lx + ly * ly > 1.0)) break;
            lx = -1.0 + 2.0 * unit_interval(r[2]);
            ly = -1.0 + 2.0 * unit_interval(r[3]);
            ++call;
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

struct Ray { float ox, oy, oz, dx, dy, dz; };

// Does a hit at t in group g replace `best`?  main.py:144 replaces on a strictly smaller t while
// it goes through the groups in order, so of equal t the earlier group stays; written with the
// group indices, the rule holds whatever order the groups are gone through in.  (Inside a group
// the candidates come in index order and the first of equal t stays, numpy.argmin.)
__device__ __forceinline__ bool closer(float t, int g, const Closest& best) {
    return t < best.t || (t == best.t && g < best.group);
}

// What the reference would have evaluated, for the roofline accounting
struct PairCounts { unsigned long long sphere, triangle, cull, disk, sphere_slots, triangle_slots, triangle_exact, sphere_exact; };

// ---- one (ray, primitive) pair, exactly as the reference evaluates it ------
// sphere_group.py:137-205.  `s` = (centre, radius^2).
__device__ __forceinline__ float sphere_discriminant(const Ray& r, float sx, float sy, float sz, float sr2, float& h) {
    const float cox = r.ox - sx, coy = r.oy - sy, coz = r.oz - sz;
    h = (r.dx * cox + r.dy * coy) + r.dz * coz;
    const float c = ((cox * cox + coy * coy) + coz * coz) - sr2;
    return h * h - c;
}

__device__ __forceinline__ float sphere_root(float h, float disc, float t_min, float sentinel) {
    const float root = sqrtf(disc);
    float t_near = -h - root, t_far = -h + root;
    t_near = (t_near < t_min) ? sentinel : t_near;
    t_far = (t_far < t_min) ? sentinel : t_far;
    return fminf(t_near, t_far);
}

// triangle_group.py:289-365: numerators of u, v, t and the determinant
struct TriTerms { float det, un, vn, tn; };
__device__ __forceinline__ TriTerms triangle_terms(const Ray& r, float P0x, float P0y, float P0z, float Ax, float Ay,
                                                   float Az, float Bx, float By, float Bz) {
    TriTerms k;
    const float p0 = r.dy * Bz - r.dz * By, p1 = r.dz * Bx - r.dx * Bz, p2 = r.dx * By - r.dy * Bx;
    k.det = (Ax * p0 + Ay * p1) + Az * p2;
    const float tx = r.ox - P0x, ty = r.oy - P0y, tz = r.oz - P0z;
    k.un = (tx * p0 + ty * p1) + tz * p2;
    const float q0 = ty * Az - tz * Ay, q1 = tz * Ax - tx * Az, q2 = tx * Ay - ty * Ax;
    k.vn = (r.dx * q0 + r.dy * q1) + r.dz * q2;
    k.tn = (Bx * q0 + By * q1) + Bz * q2;
    return k;
}

// triangle_group.py:302-305,315,356,365,372-395: exact u, v, t and the miss predicate
__device__ __forceinline__ bool triangle_exact(const TriTerms& k, float t_min, float t_max, float& t) {
    const bool parallel = fabsf(k.det) < 0.00001f;
    const float inv = 1.0f / (parallel ? 1.0f : k.det);
    const float u = k.un * inv, v = k.vn * inv;
    t = k.tn * inv;
    const bool miss = parallel | (u > 1.0f) | (u < 0.0f) | (v < 0.0f) | ((u + v) > 1.0f) | (t < t_min) | (t > t_max);
    return !miss;
}

// bounding-sphere pre-cull, triangle_group.py:158-166 (no t test)
__device__ __forceinline__ bool cull_pass(const Ray& r, float bx, float by, float bz, float br2) {
    const float cox = r.ox - bx, coy = r.oy - by, coz = r.oz - bz;
    const float hh = (r.dx * cox + r.dy * coy) + r.dz * coz;
    const float cc = ((cox * cox + coy * coy) + coz * coz) - br2;
    return (hh * hh - cc) > 0.0001f;
}

// disk.py:48-69; -> hit, t
__device__ __forceinline__ bool disk_hit(const Ray& r, const float c[4], const float n[4], float t_min, float t_max,
                                         float sentinel, float& t) {
    const float ddn = (r.dx * n[0] + r.dy * n[1]) + r.dz * n[2];
    const float num = ((c[0] - r.ox) * n[0] + (c[1] - r.oy) * n[1]) + (c[2] - r.oz) * n[2];
    bool ok = fabsf(ddn) > 0.0001f;
    t = ok ? num / ddn : sentinel;
    ok = ok && (t > t_min) && (t < t_max);
    const float rx = (r.ox + r.dx * t) - c[0], ry = (r.oy + r.dy * t) - c[1], rz = (r.oz + r.dz * t) - c[2];
    return ok && (((rx * rx + ry * ry) + rz * rz) < c[3]);
}

// ---- wave-uniform loops over one group --------------------------------------
// All 64 lanes walk the same primitives, two per step: the records of two
// neighbouring primitives are stored component-interleaved, arrive in SGPR
// pairs through scalar loads issued one step ahead of their use, and feed
// packed FP32 instructions (v_pk_mul_f32 / v_pk_add_f32) whose two halves
// belong to the two primitives.  A packed instruction takes its SGPR-pair
// operand at no extra issue cost, where a plain VALU instruction with an SGPR
// operand issues at little more than half the rate of an all-VGPR one.  Each
// half is an ordinary IEEE multiply or add, so the results are the ones of
// the one-primitive functions above.  Lanes with `active` false ride along.
// Lowest index wins equal t (numpy.argmin, sphere_group.py:213,
// triangle_group.py:399); an earlier group wins equal t (main.py:144, see
// closer()).  A group with an odd count ends in a pair whose second half
// repeats the first.
struct SpherePair { f32x2 cx, cy, cz, r2; };                          // 2 x f32x4
struct TriPair { f32x2 p0x, p0y, p0z, ax, ay, az, bx, by, bz; };      // 5 x f32x4 (last two floats unused)

// Scalar loads return out of order, so the only usable wait is "all of them".
// The loops below therefore keep exactly one record fetch in flight: they wait
// for everything, issue the fetch of the next record and then spend a whole
// step of arithmetic on the record that has just arrived.  Two register sets
// alternate, so no record is ever copied (a scalar move costs a SIMD as much
// issue time as a packed multiply, and the scalar unit is shared by the four
// SIMDs of a compute unit).  The fetch one past a group's last record lands in
// the next group's records or in the padding that ends the array.
typedef float f32x16a __attribute__((ext_vector_type(16), aligned(16)));
typedef float f32x2a __attribute__((ext_vector_type(2), aligned(8)));

// wait for every scalar load in flight; nothing may be scheduled across
#define WRT_SCALAR_LOADS_ARRIVED()                 \
    do {                                           \
        __builtin_amdgcn_sched_barrier(0);         \
        __builtin_amdgcn_s_waitcnt(0xc07f);        \
        __builtin_amdgcn_sched_barrier(0);         \
    } while (0)
#define WRT_PIN() __builtin_amdgcn_sched_barrier(0)

// (volatile: the fetch stays one wide load issued where it is written; otherwise the compiler narrows it
// to the numbers used, splits it into two or three loads and sinks a piece of it into the step that uses
// it, where that step then waits for the piece and for the next record's fetch with it)
__device__ __forceinline__ TriPair fetch_triangle_pair(const WRT_CONST f32x4* p) {
    const f32x16a lo = *reinterpret_cast<const volatile WRT_CONST f32x16a*>(p);
    const f32x2a hi = *reinterpret_cast<const volatile WRT_CONST f32x2a*>(p + 4);
    return TriPair{{lo[0], lo[1]},   {lo[2], lo[3]},   {lo[4], lo[5]},   {lo[6], lo[7]},  {lo[8], lo[9]},
                   {lo[10], lo[11]}, {lo[12], lo[13]}, {lo[14], lo[15]}, {hi[0], hi[1]}};
}
struct SphereQuad { SpherePair a, b; };
__device__ __forceinline__ SphereQuad fetch_sphere_quad(const WRT_CONST f32x4* p) {
    const f32x16a v = *reinterpret_cast<const volatile WRT_CONST f32x16a*>(p);
    return SphereQuad{SpherePair{{v[0], v[1]}, {v[2], v[3]}, {v[4], v[5]}, {v[6], v[7]}},
                      SpherePair{{v[8], v[9]}, {v[10], v[11]}, {v[12], v[13]}, {v[14], v[15]}}};
}

struct Ray2 { f32x2 ox, oy, oz, dx, dy, dz; };                        // each component in both halves
__device__ __forceinline__ Ray2 both_halves(const Ray& r) {
    return Ray2{{r.ox, r.ox}, {r.oy, r.oy}, {r.oz, r.oz}, {r.dx, r.dx}, {r.dy, r.dy}, {r.dz, r.dz}};
}

// An upper bound of the reference's discriminant (sphere_group.py:137-190) for two spheres in 7
// packed instructions.  It only has to be safe, so it is computed in expanded form about a point
// g of the group (the median sphere centre): with o' = o - g per lane and c' = c - g per sphere,
//     (d.co)^2 - |co|^2 + r^2  =  (d.c')^2 - (|c'|^2 - r^2) + w.c' + L,
//     w = 2 (o' - (d.o') d),   L = (d.o')^2 - |o'|^2,
// a dot product with d, its square less one per-sphere number k (the pair records carry c' and k),
// and a fused chain of three with the per-lane vector w.  L is the same for every sphere of the
// group, so it is not added at all: the lane compares with its own threshold 1e-5 - L instead.
// Error budget, u = 2^-24, S = |o'| + |c'|:  the reference's float32 value exceeds the exact
// expression of its rounded co by at most (8.1 |d|^2 + 5) u S^2 + 2u r^2; the rounded co, o' and c'
// move the exact expression by at most 4u (|d|^2 + 1) S^2; of the evaluation below, d.c' errs by
// 3u |d||c'|, its square less k by 7u |d|^2 |c'|^2 + u (|c'|^2 + r^2), the three roundings of the
// chain by 3u ((|d|^2 + 1) S^2 + r^2), the lane's float32 w by (2.5 |d|^2 + 0.5) u S^2 in w.c', and
// the lane's L with the subtraction from the threshold by (9 |d|^2 + 7) u |o'|^2.  With
// S^2 <= 2 (|o'|^2 + |c'|^2) the sum is below
//     (58.2 |d|^2 + 34) u |o'|^2 + (49.2 |d|^2 + 27) u |c'|^2 + 6u r^2.
// The lane shrinks its |o'|^2 by (1 - kappa), kappa = 3.7e-6 |d|^2 + 2.7e-6 (a margin on the
// first term); the record's k = |c'|^2 (1 - 250u) - r^2 (1 + 2e-6) holds the other two for
// |d|^2 <= 4 (lanes with longer directions, or with |o'|^2 beyond 1e30, look at every sphere).
struct SphereBoundRay { f32x2 wx, wy, wz, dx, dy, dz; };   // each number in both halves
__device__ __forceinline__ f32x2 sphere_discriminant_bound2(const SphereBoundRay& r, const SpherePair& s) {
    const f32x2 g = __builtin_elementwise_fma(r.dz, s.cz, __builtin_elementwise_fma(r.dy, s.cy, r.dx * s.cx));
    const f32x2 a = __builtin_elementwise_fma(g, g, -s.r2);                            // s.r2 holds k here
    return __builtin_elementwise_fma(r.wz, s.cz, __builtin_elementwise_fma(r.wy, s.cy,
               __builtin_elementwise_fma(r.wx, s.cx, a)));
}
// the lane's half of the bound: -> its threshold (the bound of a sphere it has to look at is above it)
__device__ __forceinline__ float sphere_bound_setup(const Ray& r, bool active, float gx, float gy, float gz,
                                                    SphereBoundRay& r2) {
    const float ox = r.ox - gx, oy = r.oy - gy, oz = r.oz - gz;
    const float dd = (r.dx * r.dx + r.dy * r.dy) + r.dz * r.dz;
    const float d_o = (r.dx * ox + r.dy * oy) + r.dz * oz, oo = (ox * ox + oy * oy) + oz * oz;
    const float keep = 1.0f - (3.7e-6f * dd + 2.7e-6f);          // 1 - kappa, rounded down
    const float L = d_o * d_o - oo * keep;
    const float wx = 2.0f * (ox - d_o * r.dx), wy = 2.0f * (oy - d_o * r.dy), wz = 2.0f * (oz - d_o * r.dz);
    r2 = SphereBoundRay{{wx, wx}, {wy, wy}, {wz, wz}, {r.dx, r.dx}, {r.dy, r.dy}, {r.dz, r.dz}};
    // lanes that ride along can never pass; lanes outside the bound's assumptions (written so that a
    // NaN is outside) pass always; the threshold leaves room for the rounding of its own subtraction
    return !active ? __builtin_inff() : ((dd <= 4.0f && oo <= 1.0e30f) ? 0.00000999999f - L : -__builtin_inff());
}

// Candidates of the vote are not evaluated on the spot: a hit is rare for one
// lane but common for some lane of 64, and the exact evaluation (IEEE square
// root or division, compares) costs a wavefront more than the arithmetic of
// the step itself.  Each lane instead notes the primitives it has to look at
// in a short queue in LDS, and the wavefront works the queues off together
// afterwards, the lanes sharing the noted pairs out among themselves
// (drain_spheres, drain_triangle_blocks): a few rounds per group instead of
// one per candidate.
#ifndef WRT_QUEUE_SLOTS
#define WRT_QUEUE_SLOTS 12
#endif
// TREE_HEIGHT: what the hierarchy builder keeps its trees within (the walk's stack is a ring in the queue slots)
// MESH_QUEUE_SLOTS: the slots of the kernel instances that run the triangle loops without the walk (sixteen measure
// 0.6 % faster there than twelve: fewer, larger drains; the instance with the walk would lose a wavefront per SIMD to them)
#ifndef WRT_MESH_QUEUE_SLOTS
#define WRT_MESH_QUEUE_SLOTS 16
#endif
enum { QUEUE_SLOTS = WRT_QUEUE_SLOTS, TREE_HEIGHT = 12, MESH_QUEUE_SLOTS = WRT_MESH_QUEUE_SLOTS };
static_assert(TREE_HEIGHT <= QUEUE_SLOTS, "the walk's stack lives in the queue slots");
static_assert(MESH_QUEUE_SLOTS >= QUEUE_SLOTS && MESH_QUEUE_SLOTS <= 28, "the pooled drain's entry masks are 32 bits");
struct LaneQueue {
    unsigned* slot;     // slot[k * 64]: k-th entry of this lane (a primitive's index inside the group, or a block)
    int n;
    int room;           // entries a lane can hold (QUEUE_SLOTS or MESH_QUEUE_SLOTS: what the kernel laid out)
    long long drain_ticks;   // builds with WRT_TIMING: time spent working queues off
    unsigned* share_list;            // the wave's SHARE_ITEMS (owner lane << 24 | triangle) work items
    unsigned long long* share_best;  // the wave's 64 (ordered t << 32 | triangle) minima, one per owner lane
    int noted;                       // primitives in this lane's block entries
#if defined(WRT_DRAIN_COUNT)
    long long count_trips, count_items, count_rounds, count_longest;   // measurement builds: the pooled drain's work
#endif
};
#ifndef WRT_SHARE_ITEMS
#define WRT_SHARE_ITEMS 384
#endif
enum { SHARE_ITEMS = WRT_SHARE_ITEMS };
#ifndef WRT_TIMING
#define WRT_TIMING 0
#endif
#if defined(WRT_DRAIN_COUNT)
#define WRT_DRAIN_BEGIN(q) ((void)0)
#define WRT_DRAIN_END(q) ((q).drain_ticks += 1)
#elif WRT_TIMING
#define WRT_DRAIN_BEGIN(q) const long long drain_began = (long long)__builtin_readcyclecounter()
#define WRT_DRAIN_END(q) ((q).drain_ticks += (long long)__builtin_readcyclecounter() - drain_began)
#else
#define WRT_DRAIN_BEGIN(q) ((void)0)
#define WRT_DRAIN_END(q) ((void)0)
#endif
__device__ __forceinline__ void push(LaneQueue& q, bool take, int index) {
    if (take) { q.slot[q.n * 64] = (unsigned)index; ++q.n; }
}
// true when some lane's queue cannot take `entries` more: time to work the queues off
__device__ __forceinline__ bool short_of_room(const LaneQueue& q, int entries) {
    return __ballot(q.n > q.room - entries) != 0ull;
}

// Block entries: the triangle reject loop looks at 16 triangles between two visits to the queue.
// Each test leaves its verdict in a sign bit, v_alignbit shifts the sign bits of a block
// into one mask (first primitive in bit 15), and a lane with any bit set appends
// (block << 16 | mask).  Every lane writes its next slot each time; only a lane with a
// bit set moves on to the slot after it.
__device__ __forceinline__ unsigned shift_in_signs(unsigned mask, f32x2 v) {
    mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(v.x), 31);
    return __builtin_amdgcn_alignbit(mask, __float_as_uint(v.y), 31);
}
__device__ __forceinline__ void note_block(LaneQueue& q, int block, unsigned mask) {
    q.slot[q.n * 64] = ((unsigned)block << 16) | mask;
    q.n += (mask != 0u) ? 1 : 0;
    q.noted += __popc(mask);
}
__device__ __forceinline__ bool queue_full(const LaneQueue& q) { return __ballot(q.n >= q.room) != 0ull; }

// float -> unsigned with the same order (and back)
__device__ __forceinline__ unsigned ordered_bits(float t) {
    const unsigned b = __float_as_uint(t);
    return b ^ ((b >> 31) != 0u ? 0xffffffffu : 0x80000000u);
}
__device__ __forceinline__ float from_ordered_bits(unsigned o) {
    return __uint_as_float(o ^ ((o >> 31) != 0u ? 0x80000000u : 0xffffffffu));
}

// sphere_group.py:190-213 for the queued spheres of the wavefront's lanes.  The lanes' lists differ in length (five
// entries in the longest against 1.7 on average in a field of 484 spheres), so the wavefront pools them as it does the
// triangles' (drain_triangle_blocks below): entry k of every lane that has one goes into a list in LDS, at consecutive
// places (v_mbcnt of the ballot); whenever the list holds 64 items the lanes take one each, borrow the owner's ray
// (ds_bpermute), evaluate the pair as the reference does and lower the owner's (t, index) word in `share_best` with one
// 64-bit minimum: smaller t wins, equal t goes to the smaller index (numpy.argmin), and the owner's word starts at
// (best.t, 0) so that an equal t of this group never replaces a hit of an earlier group or drain (strict <,
// main.py:144), at (best.t, all ones) if the hit so far is a later group's.  What is left of the list moves to its
// front; the last, partial round comes after the last entries.
enum { SPHERE_SHARE_ITEMS = 128 };     // 63 items left over + 64 of one more entry per lane
__device__ __forceinline__ void sphere_round(SceneArg S, LaneQueue& q, int first, int items, const Ray& r, float t_min,
                                             float sentinel) {
    const unsigned lane = threadIdx.x & 63u;
    const bool have = (int)lane < items;
    const unsigned item = have ? q.share_list[lane] : 0u;
    const int owner4 = (int)(item >> 24) * 4;
    Ray borrowed;
    borrowed.ox = __int_as_float(__builtin_amdgcn_ds_bpermute(owner4, __float_as_int(r.ox)));
    borrowed.oy = __int_as_float(__builtin_amdgcn_ds_bpermute(owner4, __float_as_int(r.oy)));
    borrowed.oz = __int_as_float(__builtin_amdgcn_ds_bpermute(owner4, __float_as_int(r.oz)));
    borrowed.dx = __int_as_float(__builtin_amdgcn_ds_bpermute(owner4, __float_as_int(r.dx)));
    borrowed.dy = __int_as_float(__builtin_amdgcn_ds_bpermute(owner4, __float_as_int(r.dy)));
    borrowed.dz = __int_as_float(__builtin_amdgcn_ds_bpermute(owner4, __float_as_int(r.dz)));
    if (have) {
        const int prim = first + (int)(item & 0xffffffu);
        const f32x4 s = S.sph[prim];
        float h;
        const float disc = sphere_discriminant(borrowed, s.x, s.y, s.z, s.w, h);
        const float t = sphere_root(h, disc, t_min, sentinel);
        // (t == sentinel: both roots before t_min; it can replace nothing)
        if (disc > 0.00001f && t < sentinel)
            atomicMin(&q.share_best[item >> 24],
                      ((unsigned long long)ordered_bits(t) << 32) | (unsigned long long)(unsigned)prim);
    }
}
__device__ __forceinline__ void drain_spheres(SceneArg S, LaneQueue& q, int g, int first, const Ray& r,
                                              float t_min, float sentinel, Closest& best) {
    WRT_DRAIN_BEGIN(q);
#if defined(WRT_DRAIN_COUNT)
    {
        int longest = q.n, sum = q.n;
        for (int d = 32; d > 0; d >>= 1) {
            const int o = __shfl_xor(longest, d, 64); longest = o > longest ? o : longest;
            sum += __shfl_xor(sum, d, 64);
        }
        q.count_trips += longest; q.count_items += sum; q.count_longest += longest; q.count_rounds += (sum + 63) / 64;
    }
#endif
    const unsigned lane = threadIdx.x & 63u;
    const unsigned long long start = ((unsigned long long)ordered_bits(best.t) << 32) |
                                     (best.group > g ? 0xffffffffull : 0ull);
    q.share_best[lane] = start;
    int listed = 0;
    for (int k = 0;; ++k) {
        const unsigned long long writers = __ballot(k < q.n);
        if (writers == 0ull) break;
        if (k < q.n) {
            const unsigned at = __builtin_amdgcn_mbcnt_hi((unsigned)(writers >> 32),
                                                          __builtin_amdgcn_mbcnt_lo((unsigned)writers, (unsigned)listed));
            q.share_list[at] = (lane << 24) | q.slot[k * 64];
        }
        listed += __popcll(writers);
        if (listed >= 64) {
            sphere_round(S, q, first, 64, r, t_min, sentinel);
            listed -= 64;
            const unsigned kept = (int)lane < listed ? q.share_list[64 + lane] : 0u;
            q.share_list[lane] = kept;
        }
    }
    if (listed > 0) sphere_round(S, q, first, listed, r, t_min, sentinel);
    const unsigned long long won = q.share_best[lane];
    if (won != start) {
        best.t = from_ordered_bits((unsigned)(won >> 32)); best.group = g; best.prim = (int)(unsigned)won;
    }
    q.n = 0;
    q.noted = 0;
    WRT_DRAIN_END(q);
}

// one vote for two spheres; the lanes that have to look at one note it down
// (index + 1 == count: the second half repeats the group's last sphere and is left out)
// (the queues are looked at for room only after a step that added to them)
__device__ __forceinline__ void sphere_step(SceneArg S, const SpherePair& s, int index, int count, float thr,
                                            const SphereBoundRay& r2, LaneQueue& q, int g, int first, const Ray& r,
                                            float t_min, float sentinel, Closest& best) {
    const f32x2 d = sphere_discriminant_bound2(r2, s);
    if (__ballot(fmaxf(d.x, d.y) > thr) != 0ull) {
        push(q, d.x > thr, index);
        push(q, d.y > thr && index + 1 < count, index + 1);
        if (short_of_room(q, 2)) drain_spheres(S, q, g, first, r, t_min, sentinel, best);
    }
}

// the bounds of four spheres (the two chains written in turn: a packed instruction that reads the result of the
// one before it costs a wait state)
__device__ __forceinline__ void sphere_bounds4(const SphereBoundRay& r2, const SpherePair& s, const SpherePair& s2,
                                               f32x2& d, f32x2& e) {
    d = r2.dx * s.cx; e = r2.dx * s2.cx;
    d = __builtin_elementwise_fma(r2.dy, s.cy, d); e = __builtin_elementwise_fma(r2.dy, s2.cy, e);
    d = __builtin_elementwise_fma(r2.dz, s.cz, d); e = __builtin_elementwise_fma(r2.dz, s2.cz, e);
    d = __builtin_elementwise_fma(d, d, -s.r2); e = __builtin_elementwise_fma(e, e, -s2.r2);
    d = __builtin_elementwise_fma(r2.wx, s.cx, d); e = __builtin_elementwise_fma(r2.wx, s2.cx, e);
    d = __builtin_elementwise_fma(r2.wy, s.cy, d); e = __builtin_elementwise_fma(r2.wy, s2.cy, e);
    d = __builtin_elementwise_fma(r2.wz, s.cz, d); e = __builtin_elementwise_fma(r2.wz, s2.cz, e);
}

// two steps with one vote: four spheres, one v_max3_f32, one v_max_f32 and one compare (a vote that
// comes out true is rare enough for the four notes it then costs not to matter)
__device__ __forceinline__ void sphere_step2(SceneArg S, const SpherePair& s, const SpherePair& s2, int index, int count,
                                             float thr, const SphereBoundRay& r2, LaneQueue& q, int g, int first,
                                             const Ray& r, float t_min, float sentinel, Closest& best) {
    f32x2 d, e;
    sphere_bounds4(r2, s, s2, d, e);
    if (__ballot(fmaxf(fmaxf(fmaxf(d.x, d.y), e.x), e.y) > thr) != 0ull) {
        push(q, d.x > thr, index);
        push(q, d.y > thr && index + 1 < count, index + 1);
        push(q, e.x > thr && index + 2 < count, index + 2);
        push(q, e.y > thr && index + 3 < count, index + 3);
        if (short_of_room(q, 4)) drain_spheres(S, q, g, first, r, t_min, sentinel, best);
    }
}

// `records`: the group's sphere pairs, 2 x f32x4 each: (c - g, k) of two spheres;
// (gx, gy, gz) = g
__device__ __forceinline__ void loop_spheres(SceneArg S, const WRT_CONST f32x4* records, int g, int first,
                                             int count, bool active, const Ray& r, float gx, float gy, float gz,
                                             float t_min, float sentinel, LaneQueue& q, Closest& best) {
    SphereBoundRay r2;
    const float thr = sphere_bound_setup(r, active, gx, gy, gz, r2);
    const int pairs = (count + 1) >> 1;
    SphereQuad A = fetch_sphere_quad(records), B;
    int j = 0;
#define WRT_STEP(rec, at) sphere_step(S, rec, at, count, thr, r2, q, g, first, r, t_min, sentinel, best)
    for (; j + 3 < pairs; j += 4) {
        WRT_SCALAR_LOADS_ARRIVED();
        B = fetch_sphere_quad(records + 4);
        WRT_PIN();
        sphere_step2(S, A.a, A.b, 2 * j, count, thr, r2, q, g, first, r, t_min, sentinel, best);
        WRT_SCALAR_LOADS_ARRIVED();
        A = fetch_sphere_quad(records + 8);
        WRT_PIN();
        sphere_step2(S, B.a, B.b, 2 * j + 4, count, thr, r2, q, g, first, r, t_min, sentinel, best);
        records += 8;
    }
    WRT_SCALAR_LOADS_ARRIVED();
    if (j < pairs) {                       // up to three pairs left: A holds two of them
        B = fetch_sphere_quad(records + 4);
        WRT_STEP(A.a, 2 * j);
        if (j + 1 < pairs) WRT_STEP(A.b, 2 * j + 2);
        if (j + 2 < pairs) WRT_STEP(B.a, 2 * j + 4);
    }
#undef WRT_STEP
    drain_spheres(S, q, g, first, r, t_min, sentinel, best);
}

// Measurement variant (-DWRT_STAGE_SPHERES_IN_LDS=1), not the shipped path: the same loop with the
// records staged through LDS, 32 pairs (1 KB) per wave at a time, and read back by broadcast
// ds_read_b128 into vector registers.  Kept so that the choice of scalar loads over LDS staging can
// be measured again on any box (tools/ab.sh).
#ifndef WRT_STAGE_SPHERES_IN_LDS
#define WRT_STAGE_SPHERES_IN_LDS 0
#endif
__device__ __forceinline__ void loop_spheres_staged(SceneArg S, const f32x4* records, f32x4* tile, int g, int first,
                                                    int count, bool active, const Ray& r, float gx, float gy,
                                                    float gz, float t_min, float sentinel, LaneQueue& q,
                                                    Closest& best) {
    const unsigned lane = threadIdx.x & 63u;
    SphereBoundRay r2;
    const float thr = sphere_bound_setup(r, active, gx, gy, gz, r2);
    const int pairs = (count + 1) >> 1;
    for (int base = 0; base < pairs; base += 32) {
        if (2 * base + (int)lane < 2 * pairs) tile[lane] = records[2 * base + (int)lane];
        const int n = pairs - base < 32 ? pairs - base : 32;
        for (int j = 0; j < n; ++j) {
            const f32x4 a = tile[2 * j], b = tile[2 * j + 1];
            const SpherePair s = {{a.x, a.y}, {a.z, a.w}, {b.x, b.y}, {b.z, b.w}};
            sphere_step(S, s, 2 * (base + j), count, thr, r2, q, g, first, r, t_min, sentinel, best);
        }
    }
    drain_spheres(S, q, g, first, r, t_min, sentinel, best);
}

struct TriTerms2 { f32x2 det, un, vn, tn; };
__device__ __forceinline__ TriTerms2 triangle_terms2(const Ray2& r, const TriPair& k) {
    TriTerms2 o;
    const f32x2 p0 = r.dy * k.bz - r.dz * k.by, p1 = r.dz * k.bx - r.dx * k.bz, p2 = r.dx * k.by - r.dy * k.bx;
    o.det = (k.ax * p0 + k.ay * p1) + k.az * p2;
    const f32x2 tx = r.ox - k.p0x, ty = r.oy - k.p0y, tz = r.oz - k.p0z;
    o.un = (tx * p0 + ty * p1) + tz * p2;
    const f32x2 q0 = ty * k.az - tz * k.ay, q1 = tz * k.ax - tx * k.az, q2 = tx * k.ay - ty * k.ax;
    o.vn = (r.dx * q0 + r.dy * q1) + r.dz * q2;
    o.tn = (k.bx * q0 + k.by * q1) + k.bz * q2;
    return o;
}

// Conservative filter for two triangles (1-ulp reciprocal, widened bounds):
// a slack is negative only when the exact evaluation cannot produce a hit
// closer than t_cap.  One number per pair of primitives, so a single compare
// feeds the wave-level vote.
__device__ __forceinline__ f32x2 triangle_slack2(const TriTerms2& k, float t_lo, float t_cap) {
    const f32x2 ri = {__builtin_amdgcn_rcpf(k.det.x), __builtin_amdgcn_rcpf(k.det.y)};
    const f32x2 ua = k.un * ri, va = k.vn * ri, ta = k.tn * ri;
    const f32x2 w = 1.00001f - (ua + va);
    const f32x2 inside = f32x2{fminf(fminf(ua.x, va.x), w.x), fminf(fminf(ua.y, va.y), w.y)} + 0.00001f;
    const f32x2 lo = ta - t_lo, hi = t_cap - ta;
    return f32x2{fminf(fminf(inside.x, lo.x), hi.x), fminf(fminf(inside.y, lo.y), hi.y)};
}

struct TriangleWalk { float t_min, t_max, t_lo, t_hi, t_cap; };

// triangle_group.py:302-399 for the queued triangles of each lane
__device__ __forceinline__ void drain_triangles(SceneArg S, LaneQueue& q, int g, int first, bool active,
                                                const Ray& r, TriangleWalk& w, Closest& best) {
    WRT_DRAIN_BEGIN(q);
    for (int k = 0; __ballot(k < q.n) != 0ull; ++k) {
        if (k < q.n) {
            const int prim = first + (int)q.slot[k * 64];
            const f32x4 a = S.tri[2 * prim], b = S.tri[2 * prim + 1];
            const TriTerms terms = triangle_terms(r, a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, S.tri_bz[prim]);
            float t;
            if (triangle_exact(terms, w.t_min, w.t_max, t) && closer(t, g, best)) { best.t = t; best.group = g; best.prim = prim; }
        }
    }
    q.n = 0;
    // inactive lanes keep an unreachable cap
    w.t_cap = active ? fminf(w.t_hi, best.t * 1.00001f) : -__builtin_inff();
    WRT_DRAIN_END(q);
}

// one vote for two triangles; the lanes that have to look at one note it down
// (index + 1 == count: the second half repeats the group's last triangle and is left out)
__device__ __forceinline__ void triangle_step(const TriPair& rec, int index, int count, const Ray2& r2,
                                              const TriangleWalk& w, LaneQueue& q) {
    const TriTerms2 k = triangle_terms2(r2, rec);
    const f32x2 slack = triangle_slack2(k, w.t_lo, w.t_cap);
    if (__ballot(fmaxf(slack.x, slack.y) >= 0.0f) != 0ull) {
        push(q, slack.x >= 0.0f, index);
        push(q, slack.y >= 0.0f && index + 1 < count, index + 1);
    }
}

// `records`: the group's triangle pairs, 5 x f32x4 each
__device__ __forceinline__ void loop_triangles(SceneArg S, const WRT_CONST f32x4* records, int g, int first,
                                               int count, bool active, const Ray& r, float t_min, float t_max,
                                               LaneQueue& q, Closest& best) {
    TriangleWalk w = {t_min, t_max, t_min * 0.99999f, t_max * 1.00001f, 0.0f};
    w.t_cap = active ? fminf(w.t_hi, best.t * 1.00001f) : -__builtin_inff();
    const Ray2 r2 = both_halves(r);
    const int pairs = (count + 1) >> 1;
    TriPair A = fetch_triangle_pair(records), B;
    int j = 0;
    for (; j + 1 < pairs; j += 2) {
        WRT_SCALAR_LOADS_ARRIVED();
        if (short_of_room(q, 2)) drain_triangles(S, q, g, first, active, r, w, best);
        B = fetch_triangle_pair(records + 5);
        WRT_PIN();
        triangle_step(A, 2 * j, count, r2, w, q);
        WRT_SCALAR_LOADS_ARRIVED();
        if (short_of_room(q, 2)) drain_triangles(S, q, g, first, active, r, w, best);
        A = fetch_triangle_pair(records + 10);
        WRT_PIN();
        triangle_step(B, 2 * j + 2, count, r2, w, q);
        records += 10;
    }
    WRT_SCALAR_LOADS_ARRIVED();
    if (j < pairs) {
        if (short_of_room(q, 2)) drain_triangles(S, q, g, first, active, r, w, best);
        triangle_step(A, 2 * j, count, r2, w, q);
    }
    drain_triangles(S, q, g, first, active, r, w, best);
}

// ---- triangle loop with a bounding-sphere reject in front ---------------------
// Möller-Trumbore on every pair costs the wavefront ~29 packed instructions
// per triangle; a ray misses nearly all triangles of a mesh by a wide margin.
// Each triangle therefore carries the smallest sphere around its three
// corners, and the loop first asks whether the ray's line comes near that
// sphere (~11 packed instructions per triangle, fused multiply-adds allowed
// because the test only has to be safe, not identical).  Lanes queue the
// triangles that survive and the exact evaluation above runs on those alone.
//
// Safety: the test must keep every pair for which the reference's float32
// arithmetic reports a hit, rounding errors included.  With u = 2^-24,
// D = max|d_i|, T >= max|o_i - P0_i| (from the ray origin's offset to the
// group's bounding-sphere centre and the group's extent about it, so that it
// does not grow with the scene's distance from the origin), M = largest edge
// component, the
// reference's determinant and its u, v numerators differ from their exact
// values by at most E_det = 32u D M^2 and E = 40u D T M (three-term float32 dot
// products of two-term cross products).  The reference reports a hit only if
// |det| >= 1e-5 and u >= 0, v >= 0, u + v <= 1 as computed, so for
// E_det <= 5e-6 the exact intersection point of the line with the triangle's
// plane has barycentric coordinates u* >= -E/|det*|, v* >= -E/|det*|,
// u* + v* <= 1 + (3E + E_det)/|det*| + 5u and lies within
// rho = 208u D M^2 (T + M) / |det*| + 9e-7 M of the triangle (edge lengths
// are at most sqrt(3) M).  The line thus passes within r + rho of the sphere's
// centre.  |det*| is bounded from below with det' = d . (B x A) from the
// precomputed normal (error at most 24u D M^2 <= |det'|/2 when |det'| >= 1e-5):
// |det*| >= max(|det'|, 1e-5) / 2.  The squared-distance test itself is
// evaluated in float32 with fused multiply-adds, error at most
// 40u |d|^2 |o - c|^2, covered by `slack`.  A lane whose ray is too long or
// whose group has edges too large for E_det <= 5e-6 makes its wavefront fall
// back to the loop above.
//
// Like the sphere loops' bound, the test is evaluated in expanded form about the group's
// bounding-sphere centre g: with o' = o - g per lane and c' = c - g per triangle,
//     |d|^2 ((r + rho)^2 - |co|^2) + (d.co)^2
//         = (d.o' - d.c')^2 + |d|^2 ((2 o'.c' - |o'|^2) - (|c'|^2 - r^2) + rho (2 r + rho)),
// which, multiplied out once more with d.c' as the only product taken of the triangle's numbers
// with d, is
//     Q - slack = (d.c')^2 + c'.v + s - |d|^2 k,   v = 2 (|d|^2 o' - (d.o') d),  s = (d.o')^2 - |d|^2 |o'|^2,
// a square, one fused chain of four from a per-lane start, and a per-triangle constant k.  No reciprocal is taken:
// with rho = widen / D, D = max(|det'|, 1e-5),
//     rho (2 r + rho) <= eps r^2 + (1 + 1/eps) rho^2        (2 r rho <= eps r^2 + rho^2 / eps),
//     1 / D^2 <= 1 / E,  E = det'^2                          (for |det'| < 1e-5 the floor only makes D larger),
// so the line is near enough whenever  Q E + W >= 0  with
//     Q = (d.o' - d.c')^2 + |d|^2 ((2 o'.c' - |o'|^2) - k) + slack,   k = |c'|^2 - (1 + eps) r^2,
//     W = (1 + 1/eps) |d|^2 widen^2,
// thirteen packed instructions for two triangles.  eps is chosen per group by the host (any positive
// value is safe): 0.05 lets 5 % more squared radius through and pays for rays close to a triangle's
// plane with 21 rho^2 instead of rho (2 r + rho).  The float32 error of Q (roundings of v: 8.3u, of s: 12u, of the
// square: 7u, of the chain and k: 11u, of o - g: 4u, each times |d|^2 S^2) is at most 43u |d|^2 S^2 + u |d|^2 r^2
// with S = |o'| + |c'| <= sqrt(3) T: below `slack` = 65u |d|^2 (3 T^2); E and the last operation err by a few u relatively, W is
// taken 1e-5 too large for them.
struct BoundPair { f32x2 cx, cy, cz, k, nx, ny, nz; };                // 4 x f32x4 (last two floats unused)
__device__ __forceinline__ BoundPair fetch_bound_pair(const WRT_CONST f32x4* p) {
    const f32x16a v = *reinterpret_cast<const volatile WRT_CONST f32x16a*>(p);
    return BoundPair{{v[0], v[1]}, {v[2], v[3]}, {v[4], v[5]}, {v[6], v[7]}, {v[8], v[9]}, {v[10], v[11]}, {v[12], v[13]}};
}

struct SphereReject {
    f32x2 dx, dy, dz;                  // d, each number in both halves
    f32x2 vx, vy, vz;                  // 2 (|d|^2 o' - (d.o') d)
    f32x2 s0;                          // (d.o')^2 - |d|^2 |o'|^2 + slack; -inf for lanes that ride along
    f32x2 minus_dd;                    // -|d|^2
    f32x2 W;                           // (1 + 1/eps) |d|^2 widen^2, widen = 2 * 208u D M^2 (T + M), rounded up
    bool usable;     // the bounds of the error analysis hold for this lane
    bool active;     // the lane's ray is at this group
};
__device__ __forceinline__ SphereReject sphere_reject_setup(const Ray& r, bool active, float edge_max, float reach,
                                                            float eps, float gx, float gy, float gz) {
    SphereReject k;
    k.active = active;
    const float D = fmaxf(fmaxf(fabsf(r.dx), fabsf(r.dy)), fabsf(r.dz));
    // every corner and sphere centre of the group lies within `reach` of (gx, gy, gz) in each
    // coordinate, so T bounds |o_i - P0_i| and |o_i - c_i| wherever the scene sits
    const float ox = r.ox - gx, oy = r.oy - gy, oz = r.oz - gz;
    const float O = fmaxf(fmaxf(fabsf(ox), fabsf(oy)), fabsf(oz));
    const float T = (O + reach) * 1.000001f;
    const float dd = (r.dx * r.dx + r.dy * r.dy) + r.dz * r.dz;
    const float d_o = (r.dx * ox + r.dy * oy) + r.dz * oz, oo = (ox * ox + oy * oy) + oz * oz;
    const float widen = 2.6e-5f * D * edge_max * edge_max * (T + edge_max);
    const float W = (1.00002f * (1.0f + 1.0f / eps)) * dd * (widen * widen);
    const float slack = active ? 3.9e-6f * dd * (3.0f * T * T) + 1.0e-30f : -__builtin_inff();
    const float vx = 2.0f * (dd * ox - d_o * r.dx), vy = 2.0f * (dd * oy - d_o * r.dy), vz = 2.0f * (dd * oz - d_o * r.dz);
    const float s0 = (d_o * d_o - dd * oo) + slack;
    k.dx = f32x2{r.dx, r.dx}; k.dy = f32x2{r.dy, r.dy}; k.dz = f32x2{r.dz, r.dz};
    k.vx = f32x2{vx, vx}; k.vy = f32x2{vy, vy}; k.vz = f32x2{vz, vz};
    k.s0 = f32x2{s0, s0}; k.minus_dd = f32x2{-dd, -dd};
    k.W = f32x2{W, W};
    // written so that a NaN or infinite ray is not usable
    k.usable = !active || (D * edge_max * edge_max <= 2.5f && T <= 1.0e15f && dd * (T * T) <= 1.0e30f);
    return k;
}

// sign bit set in a half of the result: the exact evaluation may report a hit
__device__ __forceinline__ f32x2 sphere_reject2(const SphereReject& r, const BoundPair& b) {
    // (the three chains written in turn: a packed instruction that reads the result of the one before it
    // costs a wait state)
    f32x2 g = r.dx * b.cx, det = r.dx * b.nx, t = __builtin_elementwise_fma(r.vx, b.cx, r.s0);
    g = __builtin_elementwise_fma(r.dy, b.cy, g); det = __builtin_elementwise_fma(r.dy, b.ny, det);
    t = __builtin_elementwise_fma(r.vy, b.cy, t);
    g = __builtin_elementwise_fma(r.dz, b.cz, g); det = __builtin_elementwise_fma(r.dz, b.nz, det);
    t = __builtin_elementwise_fma(r.vz, b.cz, t);
    t = __builtin_elementwise_fma(r.minus_dd, b.k, t);
    const f32x2 E = det * det;
    const f32x2 Q = __builtin_elementwise_fma(g, g, t);
    return __builtin_elementwise_fma(-Q, E, -r.W);                        // sign set: the line comes near enough
}

// triangle_group.py:302-399 for the triangles each lane noted
// The lanes' lists differ a lot in length (a ray skimming the mesh passes through dozens of
// bounding spheres, most pass through a few), so the wave pools them: the lanes write the
// triangles of all entries into one list, then take 64 items at a time, each borrowing the owner's
// ray (ds_bpermute), evaluating the pair exactly as above and lowering the owner's entry of
// `share_best` with one 64-bit minimum: smaller t wins, equal t goes to the smaller index
// (numpy.argmin, triangle_group.py:399), and an entry starts at (best.t, 0) so that an equal
// t of this group never replaces a hit of an earlier group or drain (strict <, main.py:144).
// The minimum does not depend on the order of the items, so the list is written in whatever order
// is quickest: one triangle per lane and trip, and the entries dealt out so that no lane has a long
// list to itself: entry k of lane c is expanded by lane (c - 16 (k & 3)) mod 64, which spreads the
// twelve entries of a skimming ray over four lanes (trips per drain: the longest share instead of
// the longest list).  A trip's items go to consecutive places (v_mbcnt of the ballot), and when
// fewer than 64 places are left the list's whole rounds of 64 are worked off, the rest moves to its
// front and the trips go on where they stopped.
__device__ __forceinline__ void drain_triangle_blocks(SceneArg S, LaneQueue& q, int g, int first, int count,
                                                      const Ray& r, const TriangleWalk& w, Closest& best,
                                                      unsigned long long& evaluated) {
    static_assert(SHARE_ITEMS >= 128, "room for a trip's 64 items");
    WRT_DRAIN_BEGIN(q);
    const unsigned lane = threadIdx.x & 63u;
    int total = q.noted;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) total += __shfl_xor(total, d, 64);
    total = __builtin_amdgcn_readfirstlane(total);
    // equal t: an earlier group's hit stays (index field 0 cannot be undercut), a later group's goes
    const unsigned long long start = ((unsigned long long)ordered_bits(best.t) << 32) |
                                     (best.group > g ? 0xffffffffull : 0ull);
    q.share_best[lane] = start;
    // the entries this lane expands: rows j, j + 4, j + 8 of the lane 16 j places on, as far as that lane has entries
    unsigned todo = 0x11111111u & ((1u << q.n) - 1u);
#pragma unroll
    for (int j = 1; j < 4; ++j) {
        const int there = __builtin_amdgcn_ds_bpermute((int)((lane + 16u * (unsigned)j) & 63u) * 4, q.n);
        todo |= (0x11111111u << j) & ((1u << there) - 1u);
    }
#if defined(WRT_DRAIN_COUNT)
    {
        int longest = q.n;
        for (int d = 32; d > 0; d >>= 1) { const int o = __shfl_xor(longest, d, 64); longest = o > longest ? o : longest; }
        q.count_longest += longest;
    }
#endif
    const unsigned* const row0 = q.slot - lane;        // the wave's entries: row0[k * 64 + c] is entry k of lane c
    unsigned open = 0u, tagged_first = 0u;
    int listed = 0;
    for (int done = 0; done < total;) {
        bool exhausted = false;
        while (listed <= SHARE_ITEMS - 64) {
            if (open == 0u && todo != 0u) {
                const unsigned k = (unsigned)__builtin_ctz(todo);
                todo &= todo - 1u;
                const unsigned c = (lane + 16u * (k & 3u)) & 63u;
                const unsigned entry = row0[k * 64u + c];
                open = entry & 0xffffu;
                tagged_first = (c << 24) | (16u * (entry >> 16) + 15u);
            }
            const unsigned long long writers = __ballot(open != 0u);
            if (writers == 0ull) { exhausted = true; break; }
#if defined(WRT_DRAIN_COUNT)
            q.count_trips += 1;
#endif
            if (open != 0u) {
                const unsigned bit = 31u - (unsigned)__builtin_clz(open);
                open &= ~(1u << bit);
                const unsigned at = __builtin_amdgcn_mbcnt_hi((unsigned)(writers >> 32),
                                                              __builtin_amdgcn_mbcnt_lo((unsigned)writers, (unsigned)listed));
                q.share_list[at] = tagged_first - bit;
            }
            listed += __popcll(writers);
        }
        // whole rounds of 64 while more is to come (the rest moves to the front of the list), everything at the end
        const int items = exhausted ? listed : listed & ~63;
        evaluated += (unsigned)items;
        done = exhausted ? total : done + items;
#if defined(WRT_DRAIN_COUNT)
        q.count_items += items;
        q.count_rounds += (items + 63) / 64;
#endif
        for (int base = 0; base < items; base += 64) {
            const unsigned item = base + (int)lane < items ? q.share_list[base + (int)lane] : 0xffffffu;
            // (an index outside the group - only a damaged list could hold one - is left out, not fetched)
            const bool have = (item & 0xffffffu) < (unsigned)count;
            const int owner4 = (int)(item >> 24) * 4;
            Ray borrowed;
            borrowed.ox = __int_as_float(__builtin_amdgcn_ds_bpermute(owner4, __float_as_int(r.ox)));
            borrowed.oy = __int_as_float(__builtin_amdgcn_ds_bpermute(owner4, __float_as_int(r.oy)));
            borrowed.oz = __int_as_float(__builtin_amdgcn_ds_bpermute(owner4, __float_as_int(r.oz)));
            borrowed.dx = __int_as_float(__builtin_amdgcn_ds_bpermute(owner4, __float_as_int(r.dx)));
            borrowed.dy = __int_as_float(__builtin_amdgcn_ds_bpermute(owner4, __float_as_int(r.dy)));
            borrowed.dz = __int_as_float(__builtin_amdgcn_ds_bpermute(owner4, __float_as_int(r.dz)));
            if (have) {
                const int prim = first + (int)(item & 0xffffffu);
                const f32x4 a = S.tri[2 * prim], b = S.tri[2 * prim + 1];
                const TriTerms terms = triangle_terms(borrowed, a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, S.tri_bz[prim]);
                float t;
                if (triangle_exact(terms, w.t_min, w.t_max, t))
                    atomicMin(&q.share_best[item >> 24],
                              ((unsigned long long)ordered_bits(t) << 32) | (unsigned long long)(unsigned)prim);
            }
        }
        if (items < listed) {
            const unsigned kept = (int)lane < listed - items ? q.share_list[items + (int)lane] : 0u;
            q.share_list[lane] = kept;
        }
        listed -= items;
    }
    const unsigned long long won = q.share_best[lane];
    if (won != start) {
        best.t = from_ordered_bits((unsigned)(won >> 32)); best.group = g; best.prim = (int)(unsigned)won;
    }
    q.n = 0;
    q.noted = 0;
    WRT_DRAIN_END(q);
}

// `records`: the group's bounding records, 4 x f32x4 per two triangles, whole blocks of 16 triangles
__device__ __forceinline__ void loop_triangles_rejecting(SceneArg S, const WRT_CONST f32x4* records, int g,
                                                         int first, int count, const SphereReject& k, const Ray& r,
                                                         float t_min, float t_max, LaneQueue& q, Closest& best,
                                                         unsigned long long& evaluated) {
    const TriangleWalk w = {t_min, t_max, 0.0f, 0.0f, 0.0f};
    const int blocks = (count + 15) >> 4;
    BoundPair A = fetch_bound_pair(records), B;
    for (int block = 0; block < blocks; ++block) {
        unsigned mask = 0u;
#pragma unroll
        for (int quarter = 0; quarter < 4; ++quarter) {
            WRT_SCALAR_LOADS_ARRIVED();
            B = fetch_bound_pair(records + 4);
            WRT_PIN();
            mask = shift_in_signs(mask, sphere_reject2(k, A));
            WRT_SCALAR_LOADS_ARRIVED();
            A = fetch_bound_pair(records + 8);
            WRT_PIN();
            mask = shift_in_signs(mask, sphere_reject2(k, B));
            records += 8;
        }
        // triangles past the group's end repeat its last one and are left out
        const int left = count - 16 * block;
        if (left < 16) mask &= 0xffffu << (16 - left);
        note_block(q, block, k.active ? mask : 0u);         // (a NaN ray's verdicts are arbitrary)
        if (queue_full(q)) drain_triangle_blocks(S, q, g, first, count, r, w, best, evaluated);
    }
    WRT_SCALAR_LOADS_ARRIVED();
    drain_triangle_blocks(S, q, g, first, count, r, w, best, evaluated);
}

// ---- optional: a group's box hierarchy, walked by the wavefront's lanes together ------
// Not the reference's algorithm (it looks at every pair): an optional mode that evaluates
// triangle_terms / triangle_exact only for the triangles of the leaves a ray's line comes near,
// with the same closest hit.  What makes it the same hit:
//
// With exact tvec = o - P0 and Cramer's identity det tvec = un A + vn B - tn d for the exact
// numerators, the float32 evaluation (round to nearest, u = 2^-24, no fused operations, the
// association of triangle_terms) returns det_c, un_c, vn_c, tn_c with
//     |det_c - det| <= 5.83u |d||A||B|,   |un_c - un|, |vn_c - vn|, |tn_c - tn| <= 6.83u |tvec| x
//     (|d||B|, |d||A|, |A||B|)
// (a 3-term dot product errs by at most 3u |x||y|, a cross product by 2.83u |a||b| in norm,
// o - P0 by u |tvec|).  For a reported hit (|det_c| >= 1e-5, 0 <= u_c, v_c, u_c + v_c <= 1 + u,
// two more roundings each in u_c, v_c, t_c) the residual of the reported point,
//     R = (o + t_c d) - (P0 + u_c A + v_c B),
// obeys |R| <= 26.4u |tvec||d||A||B| / 1e-5 + 2.1u (|tvec| + 2|A| + 2|B| + |R|)
//          <= 0.158 |A||B| |d| T2 + 2e-6 T2        with T2 >= |o - P0|, |A|, |B| <= 2 T2.
// So o + t_c d lies in any box around the triangle's corners that is grown by that much, and
// t_c inside the box's parameter interval.  A node stores kappa = 0.17 max |A||B| over its
// triangles; the ray brings s >= |d| T2 and a = 3e-6 T2 + 1e-30 (which also covers the
// rounding of o - c, of the box corners' subtractions, and of kappa s).  The interval's ends
// carry at most 4 roundings (v_rcp_f32: 1 ulp), allowed for by the factors 1 -+ 1e-6.
// A node is left out only when its interval misses [t_min, min(best t, t_max)]; equal t still
// gets in, and the tie rule below is the brute-force loops' rule (lowest index inside the
// group, earlier group across groups) in a form that does not depend on the order of evaluation.
struct TreeRay {
    float ox, oy, oz;      // o - c
    float ix, iy, iz;      // 1 / d (v_rcp_f32)
    float mx, my, mz;      // -(o - c) / d
    float s, a;            // box growth = kappa * s + a
    bool usable;           // the bounds above hold (finite, no overflow) for this lane
};
__device__ __forceinline__ TreeRay tree_ray_setup(const Ray& r, bool active, float reach2, float gx, float gy,
                                                  float gz) {
    TreeRay k;
    k.ox = r.ox - gx; k.oy = r.oy - gy; k.oz = r.oz - gz;
    // A direction component below 1e-18 of the largest is replaced by that (same sign): the point at
    // parameter t moves by less than 1e-17 T2 for every t that can be a hit, and 1 / d stays finite.
    const float d_max = fmaxf(fmaxf(fabsf(r.dx), fabsf(r.dy)), fabsf(r.dz)), d_floor = 1.0e-18f * d_max;
    const float dx = fabsf(r.dx) < d_floor ? copysignf(d_floor, r.dx) : r.dx;
    const float dy = fabsf(r.dy) < d_floor ? copysignf(d_floor, r.dy) : r.dy;
    const float dz = fabsf(r.dz) < d_floor ? copysignf(d_floor, r.dz) : r.dz;
    k.ix = __builtin_amdgcn_rcpf(dx); k.iy = __builtin_amdgcn_rcpf(dy); k.iz = __builtin_amdgcn_rcpf(dz);
    k.mx = -k.ox * k.ix; k.my = -k.oy * k.iy; k.mz = -k.oz * k.iz;
    const float oo = (k.ox * k.ox + k.oy * k.oy) + k.oz * k.oz, dd = (r.dx * r.dx + r.dy * r.dy) + r.dz * r.dz;
    const float T2 = (sqrtf(oo) + reach2) * 1.000001f;
    k.s = 1.000001f * sqrtf(dd) * T2;
    k.a = 3.0e-6f * T2 + 1.0e-30f;
    // written so that a NaN or infinite ray is not usable; within these limits no product of the box
    // test is infinity times zero or infinity minus infinity
    k.usable = !active || (T2 <= 1.0e9f && d_max >= 1.0e-9f && d_max <= 1.0e9f);
    return k;
}

// parameter interval of the ray inside the grown box; true when it meets [t_min, upper]
__device__ __forceinline__ bool tree_box(const TreeRay& k, float lox, float loy, float loz, float hix, float hiy,
                                         float hiz, float kappa, float t_min, float upper, float& t_near) {
    const float grow = __builtin_fmaf(kappa, k.s, k.a);
    // (plane - o) / d as plane * (1/d) - o * (1/d): the rounding of the second product moves the
    // plane by at most u |o|, inside `a`
    const float x1 = __builtin_fmaf(lox - grow, k.ix, k.mx), x2 = __builtin_fmaf(hix + grow, k.ix, k.mx);
    const float y1 = __builtin_fmaf(loy - grow, k.iy, k.my), y2 = __builtin_fmaf(hiy + grow, k.iy, k.my);
    const float z1 = __builtin_fmaf(loz - grow, k.iz, k.mz), z2 = __builtin_fmaf(hiz + grow, k.iz, k.mz);
    t_near = fmaxf(fmaxf(fminf(x1, x2), fminf(y1, y2)), fminf(z1, z2));
    const float t_far = fminf(fminf(fmaxf(x1, x2), fmaxf(y1, y2)), fmaxf(z1, z2));
    // the ends moved outwards by their relative error, whatever their sign (t_min may be negative)
    return fmaxf(__builtin_fmaf(-1.0e-6f, fabsf(t_near), t_near), t_min) <=
           fminf(__builtin_fmaf(1.0e-6f, fabsf(t_far), t_far), upper);
}

// A node is 4 x f32x4: (lo0.xyz, hi0.x) (hi0.yz, lo1.xy) (lo1.z, hi1.xyz) (kappa0, kappa1, child0, child1),
// boxes relative to the group's bounds centre.  child >= 0: f32x4 offset of an inner node from the
// group's root; child < 0: -1 - (offset * 8 + n - 1), n <= 8 leaf triangles of 3 x f32x4 each,
// (P0, A.x) (A.yz, B.xy) (B.z, index inside the group, -, -), the numbers of S.tri / S.tri_bz.
// The lane's stack lives in its queue slots (QUEUE_SLOTS entries: the builder keeps trees that low).
enum { TREE_DONE = -2147483647 - 1 };

// ---- the same kind of hierarchy over a sphere group ---------------------------------------------
// A leaf holds up to 8 spheres as 2 x f32x4 each: (centre, float32(r * r)) as in S.sph, (index inside
// the group, -, -, -); a node's kappa is 1 / (the smallest sqrt(float32(r * r)) below it), rounded up.
// What makes the closest hit the brute-force one (sphere_group.py:137-213 evaluated in float32, round
// to nearest, no fused operations, u = 2^-24):
//
// Let co_f = fl(o - c), a = |co_f|, H = d . co_f, r2 = fl(r r), |d|^2 = 1 + e with |e| <= delta, and
// f(t) = |co_f + t d|^2 - r2 = a^2 + 2 t H + t^2 |d|^2 - r2, the squared distance of the ray's point at
// t from the sphere, less r2.  The evaluation computes h_c = fl(d . co_f) with |h_c - H| <= 3.01u |d| a,
// disc_c = fl(fl(h_c h_c) - fl(fl(co_f . co_f) - r2)), which differs from h_c^2 - (a^2 - r2) by
// zeta, |zeta| <= 7.1u a^2 + 2.1u r2, then root_c = fl(sqrt(disc_c)) and the reported
// t_c = fl(-h_c -+ root_c).  With tau = -h_c -+ root_c: tau^2 + 2 tau h_c = root_c^2 - h_c^2, hence
//     f(tau) = zeta + (root_c^2 - disc_c) + 2 tau (H - h_c) + tau^2 e,
// and with |tau| <= 2.43 a + sqrt(r2), disc_c <= 2.01 a^2 + r2:
//     |f(tau)| <= 29u a^2 + 7.3u r2 + (11.8 a^2 + 2 r2) delta;
// the rounding of tau to t_c moves f by at most 22.7u a^2 + 7.9u r2.  So the reported point lies
// within (52u + 11.8 delta) a^2 / r' + (15.2u + 2 delta) r2 / r' of the sphere of radius r' = sqrt(r2)
// around c (| |p| - r' | = |f| / (|p| + r') <= |f| / r'), and the exact o - c differs from co_f by at
// most u a more.  With T >= a for every sphere in the tree (the ray origin's distance from the group's
// reference point plus the tree's reach) and R >= every r', a box around a sphere of the tree grown by
//     kappa s + a',   s = (52u + 11.8 delta) T^2 + (15.2u + 2 delta) R^2,   a' = 2u (T + R),
// holds every point the evaluation can report for it, so t_c lies in that box's parameter interval,
// and in those of the boxes above it (whose kappa is at least as large).  The rest is the triangle
// walk's: intervals widened by their own roundings, equal t still enters, ties go to the lower index
// inside the group and to the earlier group.  Rays with delta > 1e-3 (a refracted direction the
// reference did not normalise) or outside the finite range make the wavefront use the uniform loop.
__device__ __forceinline__ TreeRay sphere_tree_ray_setup(const Ray& r, bool active, float reach, float r_max,
                                                         float gx, float gy, float gz) {
    TreeRay k;
    k.ox = r.ox - gx; k.oy = r.oy - gy; k.oz = r.oz - gz;
    const float d_max = fmaxf(fmaxf(fabsf(r.dx), fabsf(r.dy)), fabsf(r.dz)), d_floor = 1.0e-18f * d_max;
    const float dx = fabsf(r.dx) < d_floor ? copysignf(d_floor, r.dx) : r.dx;
    const float dy = fabsf(r.dy) < d_floor ? copysignf(d_floor, r.dy) : r.dy;
    const float dz = fabsf(r.dz) < d_floor ? copysignf(d_floor, r.dz) : r.dz;
    k.ix = __builtin_amdgcn_rcpf(dx); k.iy = __builtin_amdgcn_rcpf(dy); k.iz = __builtin_amdgcn_rcpf(dz);
    k.mx = -k.ox * k.ix; k.my = -k.oy * k.iy; k.mz = -k.oz * k.iz;
    const float oo = (k.ox * k.ox + k.oy * k.oy) + k.oz * k.oz, dd = (r.dx * r.dx + r.dy * r.dy) + r.dz * r.dz;
    const float T = (sqrtf(oo) + reach) * 1.000001f;
    const float delta = fabsf(dd - 1.0f) + 2.4e-7f;                 // dd itself carries four roundings
    k.s = 1.00001f * ((3.1e-6f + 11.8f * delta) * (T * T) + (9.1e-7f + 2.0f * delta) * (r_max * r_max));
    k.a = 1.2e-7f * (T + r_max) + 1.0e-30f;
    k.usable = !active || (T <= 1.0e9f && d_max >= 1.0e-9f && delta <= 1.0e-3f);
    return k;
}

// The same walk with the wavefront's lanes sharing the work.  Walks differ widely in length (a ray
// that leaves the outermost boxes is done after a few nodes, one that skims the mesh visits a
// hundred), and a wavefront pays for every step of its longest walk.  So a lane without work
// (its walk is over, or its ray is not at this group at all) takes the oldest entry off the stack
// of a lane that has one - the far child nearest the root, the largest piece of work that lane has
// put off - together with the numbers derived from that entry's ray, and walks it as its own.
//
// Leaves are not evaluated by the lane that meets them.  A child box that is met and is a leaf goes
// into the wavefront's pool at once, (ray owner << 26 | leaf), so that a lane's node and its stack
// entries are always inner nodes and every step is a step down for every busy lane; whenever the
// pool holds what all 64 lanes can take (16 leaves of up to 4 primitives, or 8 of up to 8; not
// waiting for all of it measured 2 to 4 % slower), the wavefront evaluates 64 lanes' worth of
// pooled leaves at once, lane i the (i mod 4)-th primitive of the (i / 4)-th leaf for the ray of that
// leaf's owner (ds_bpermute).  Every hit goes to the ray owner's entry of `share_best` through one
// 64-bit minimum (ordered t << 32 | primitive), which is also where every lane working on a ray reads
// how far that ray still has to look; the rule for equal t (lowest index inside the group, earlier
// group across groups) is the pooled drain's.  A primitive is evaluated with its ray's own numbers
// whichever lane does it, and a later evaluation only means that boxes are judged against a bound
// that is not yet as low as it could be, so the closest hit is the same bit for bit.
//
// The stacks are rings of QUEUE_SLOTS entries in LDS (`top` = position of the next push, `held`
// entries below it); a steal round pairs the i-th idle lane with the i-th lane holding entries and
// runs as soon as one pair can be made (two for a sphere group, whose walks are short).
#ifndef WRT_STEAL_MIN
#define WRT_STEAL_MIN 1
#endif
#ifndef WRT_STEAL_MIN_SPHERES
#define WRT_STEAL_MIN_SPHERES 2
#endif
#ifndef WRT_POOL_EAGER
#define WRT_POOL_EAGER 0
#endif
#ifndef WRT_POOL_EAGER_SPHERES
#define WRT_POOL_EAGER_SPHERES 0
#endif
enum { POOL_FIRST = 32, POOL_ROOM = SHARE_ITEMS - POOL_FIRST };
static_assert(POOL_ROOM >= 16 + 128, "a step adds at most 128 leaves to fewer than a round's worth");

// `count` pooled leaves from pool[base], count << leaf_shift <= 64: each lane evaluates one primitive
// of one leaf for the ray of the leaf's owner
template <int KIND>
__device__ __forceinline__ void pooled_leaf_round(const f32x4* __restrict__ tree, const unsigned* pool, int base,
                                                  int count, int leaf_shift, int first, const Ray& r, float t_min,
                                                  float t_max, LaneQueue& q, unsigned long long& evaluated) {
    WRT_DRAIN_BEGIN(q);
    const unsigned lane = threadIdx.x & 63u;
    const int which = (int)(lane >> leaf_shift), j = (int)lane & ((1 << leaf_shift) - 1);
    const unsigned entry = which < count ? pool[base + which] : 0u;
    const unsigned owner = entry >> 26;
    const int code = (int)(entry & 0x3ffffffu);
    const bool have = which < count && j <= (code & 7);
    evaluated += (unsigned)__popcll(__ballot(have));
    const int owner4 = (int)owner * 4;
    Ray b;
    b.ox = __int_as_float(__builtin_amdgcn_ds_bpermute(owner4, __float_as_int(r.ox)));
    b.oy = __int_as_float(__builtin_amdgcn_ds_bpermute(owner4, __float_as_int(r.oy)));
    b.oz = __int_as_float(__builtin_amdgcn_ds_bpermute(owner4, __float_as_int(r.oz)));
    b.dx = __int_as_float(__builtin_amdgcn_ds_bpermute(owner4, __float_as_int(r.dx)));
    b.dy = __int_as_float(__builtin_amdgcn_ds_bpermute(owner4, __float_as_int(r.dy)));
    b.dz = __int_as_float(__builtin_amdgcn_ds_bpermute(owner4, __float_as_int(r.dz)));
    if (have) {
        const f32x4* rec = tree + ((code >> 3) + j * (KIND == GROUP_SPHERES ? 2 : 3));
        if (KIND == GROUP_SPHERES) {
            const f32x4 sph = rec[0];
            float h;
            const float disc = sphere_discriminant(b, sph.x, sph.y, sph.z, sph.w, h);
            if (disc > 0.00001f) {
                // (a root beyond t_max but below the sentinel still lowers the group's t, as in the
                // reference: sphere_group.py:192-205)
                const float t = sphere_root(h, disc, t_min, t_max + 1.0f);
                atomicMin(&q.share_best[owner], ((unsigned long long)ordered_bits(t) << 32) |
                                                    (unsigned long long)(unsigned)(first + (int)__float_as_uint(rec[1].x)));
            }
        } else {
            const f32x4 a = rec[0], e = rec[1], w = rec[2];
            const TriTerms terms = triangle_terms(b, a.x, a.y, a.z, a.w, e.x, e.y, e.z, e.w, w.x);
            float t;
            if (triangle_exact(terms, t_min, t_max, t))
                atomicMin(&q.share_best[owner], ((unsigned long long)ordered_bits(t) << 32) |
                                                    (unsigned long long)(unsigned)(first + (int)__float_as_uint(w.y)));
        }
    }
    WRT_DRAIN_END(q);
}

template <int KIND = GROUP_TRIANGLES>
__device__ __forceinline__ void walk_tree_together(const f32x4* __restrict__ tree, int leaf_shift, int g, int first,
                                                   const TreeRay& own_k, const Ray& r,
                                                   bool active, float t_min, float t_max, LaneQueue& q,
                                                   Closest& best, unsigned long long& evaluated) {
    const unsigned lane = threadIdx.x & 63u;
    unsigned* const stacks = q.slot - lane;                    // the wave's [QUEUE_SLOTS][64] entries
    unsigned short* const donors_by_rank = reinterpret_cast<unsigned short*>(q.share_list);   // scratch of a steal round
    unsigned* const pool = q.share_list + POOL_FIRST;
    constexpr int steal_min = KIND == GROUP_SPHERES ? WRT_STEAL_MIN_SPHERES : WRT_STEAL_MIN;
    const int round_leaves = 64 >> leaf_shift;
    const int drain_at = round_leaves >> (KIND == GROUP_SPHERES ? WRT_POOL_EAGER_SPHERES : WRT_POOL_EAGER);
    // equal t: an earlier group's hit stays (index field 0 cannot be undercut), a later group's goes, and a
    // hit of this very group (a sphere group's outliers are looked at before the walk) is judged by its index
    const unsigned long long start = ((unsigned long long)ordered_bits(best.t) << 32) |
                                     (best.group > g ? 0xffffffffull : (best.group == g ? (unsigned long long)(unsigned)best.prim : 0ull));
    q.share_best[lane] = start;
    TreeRay k = own_k;                   // of the ray this lane is walking for (its own to begin with)
    unsigned owner = lane;
    int node = active ? 0 : TREE_DONE;
    int top = 0, held = 0;
    int pooled = 0;                      // leaves in the pool
    for (;;) {
        const unsigned long long busy = __ballot(node != TREE_DONE);
        if (busy == 0ull) break;                               // an idle lane holds no entries
        const unsigned long long idle = ~busy;
        if ((int)__popcll(idle) >= steal_min) {
            const unsigned long long givers = __ballot(held > 0);
            const int pairs = (int)__popcll(idle) < (int)__popcll(givers) ? (int)__popcll(idle) : (int)__popcll(givers);
            if (pairs >= steal_min) {
                const int rank_give = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(givers >> 32),
                                           __builtin_amdgcn_mbcnt_lo((unsigned)givers, 0u));
                const int rank_idle = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(idle >> 32),
                                           __builtin_amdgcn_mbcnt_lo((unsigned)idle, 0u));
                const bool gives = held > 0 && rank_give < pairs;
                const bool takes = node == TREE_DONE && rank_idle < pairs;
                int oldest = top - held;
                oldest += oldest < 0 ? QUEUE_SLOTS : 0;
                if (gives) {
                    donors_by_rank[rank_give] = (unsigned short)(lane | ((unsigned)oldest << 8));
                    --held;
                }
                unsigned from = lane;
                if (takes) {
                    const unsigned giver = donors_by_rank[rank_idle];
                    from = giver & 63u;
                    node = (int)stacks[(giver >> 8) * 64u + from];
                }
                const unsigned whose = (unsigned)__builtin_amdgcn_ds_bpermute((int)from * 4, (int)owner);
                if (takes) owner = whose;
                const int owner4 = (int)owner * 4;
                // the numbers the owner derived from its ray for this group, out of its registers
                // (a lane that took nothing reads again what it has: k is always its owner's)
#define WRT_BORROW(x) __int_as_float(__builtin_amdgcn_ds_bpermute(owner4, __float_as_int(x)))
                k.ox = WRT_BORROW(own_k.ox); k.oy = WRT_BORROW(own_k.oy); k.oz = WRT_BORROW(own_k.oz);
                k.ix = WRT_BORROW(own_k.ix); k.iy = WRT_BORROW(own_k.iy); k.iz = WRT_BORROW(own_k.iz);
                k.mx = WRT_BORROW(own_k.mx); k.my = WRT_BORROW(own_k.my); k.mz = WRT_BORROW(own_k.mz);
                k.s = WRT_BORROW(own_k.s); k.a = WRT_BORROW(own_k.a);
#undef WRT_BORROW
                continue;
            }
        }
        // two boxes; a child that is a leaf and is met goes to the pool at once, so that a lane's
        // node and its stack entries are always inner nodes
        bool in0 = false, in1 = false;
        float near0 = 0.0f, near1 = 0.0f;
        int c0 = 0, c1 = 0;
        if (node >= 0) {
            // (an unsigned 32-bit byte offset from the uniform base: one shift, no 64-bit address arithmetic)
            const f32x4* n = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(tree) + (unsigned)node * 16u);
            const f32x4 q0 = n[0], q1 = n[1], q2 = n[2], q3 = n[3];
            // the nearer of the ray's best hit so far and the end of its range (v_med3_f32 with -inf: a
            // minimum that takes the two numbers as they are)
            const float upper = __builtin_amdgcn_fmed3f(from_ordered_bits((unsigned)(q.share_best[owner] >> 32)),
                                                        KIND == GROUP_SPHERES ? t_max + 1.0f : t_max, -__builtin_inff());
            in0 = tree_box(k, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q3.x, t_min, upper, near0);
            in1 = tree_box(k, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.y, t_min, upper, near1);
            c0 = (int)__float_as_uint(q3.z); c1 = (int)__float_as_uint(q3.w);
        }
        const bool leaf0 = in0 && c0 < 0, leaf1 = in1 && c1 < 0;
        const unsigned long long leaves0 = __ballot(leaf0), leaves1 = __ballot(leaf1);
        if ((leaves0 | leaves1) != 0ull) {
            if (leaf0) {
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(leaves0 >> 32),
                                      __builtin_amdgcn_mbcnt_lo((unsigned)leaves0, 0u));
                pool[pooled + rank] = (owner << 26) | (unsigned)(-1 - c0);
            }
            pooled += (int)__popcll(leaves0);
            if (leaf1) {
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(leaves1 >> 32),
                                      __builtin_amdgcn_mbcnt_lo((unsigned)leaves1, 0u));
                pool[pooled + rank] = (owner << 26) | (unsigned)(-1 - c1);
            }
            pooled += (int)__popcll(leaves1);
        }
        if (node >= 0) {
            in0 = in0 && c0 >= 0; in1 = in1 && c1 >= 0;
            if (in0 && in1) {
                if (near1 < near0) { const int s = c0; c0 = c1; c1 = s; }
                q.slot[top * 64] = (unsigned)c1;
                top = top + 1 == QUEUE_SLOTS ? 0 : top + 1;
                ++held;
                node = c0;
            } else if (in0) {
                node = c0;
            } else if (in1) {
                node = c1;
            } else if (held > 0) {
                top = top == 0 ? QUEUE_SLOTS - 1 : top - 1;
                --held;
                node = (int)q.slot[top * 64];
            } else {
                node = TREE_DONE;
            }
        }
        while (pooled >= drain_at) {
            const int count = pooled < round_leaves ? pooled : round_leaves;
            pooled -= count;
            pooled_leaf_round<KIND>(tree, pool, pooled, count, leaf_shift, first, r, t_min, t_max, q, evaluated);
        }
    }
    // what is left in the pool
    if (pooled > 0) pooled_leaf_round<KIND>(tree, pool, 0, pooled, leaf_shift, first, r, t_min, t_max, q, evaluated);
    const unsigned long long won = q.share_best[lane];
    if (active && won != start) {
        best.t = from_ordered_bits((unsigned)(won >> 32)); best.group = g; best.prim = (int)(unsigned)won;
    }
}

// A sphere group through its hierarchy: the few spheres kept out of the tree first, each lane on its
// own (they bound how far the walk has to look), then the shared walk.  -> false (nothing done)
// when a lane's ray is outside the range the boxes' growth is proven for.
__device__ __forceinline__ bool walk_spheres(SceneArg S, const WRT_CONST GroupRec& G, int g, bool active, const Ray& r,
                                             float t_min, float t_max, LaneQueue& q, Closest& best,
                                             unsigned long long& evaluated) {
    const TreeRay w = sphere_tree_ray_setup(r, active, G.reach2, G.tree_rmax, G.c[0], G.c[1], G.c[2]);
    if (__ballot(!w.usable) != 0ull) return false;
    const int first = G.first, n_out = G.n_outliers;
    const float sentinel = t_max + 1.0f;
    for (int i = 0; i < n_out; ++i) {
        const int prim = first + G.outlier[i];
        const f32x4 s = as_uniform(S.sph)[prim];
        float h;
        const float disc = sphere_discriminant(r, s.x, s.y, s.z, s.w, h);
        const float t = sphere_root(h, disc, t_min, sentinel);
        if (active && disc > 0.00001f && (closer(t, g, best) || (t == best.t && g == best.group && prim < best.prim))) {
            best.t = t; best.group = g; best.prim = prim;
        }
    }
    evaluated += (unsigned long long)__popcll(__ballot(active)) * (unsigned)n_out;
    walk_tree_together<GROUP_SPHERES>(S.tree + G.tree_base, G.tree_leaf_shift, g, first, w, r, active, t_min, t_max, q,
                                      best, evaluated);
    return true;
}

// ---- the same pairs for one lane on its own (small groups, divergent code) ---
__device__ __forceinline__ void lane_spheres(SceneArg S, int g, int first, int count, const Ray& r,
                                             float t_min, float sentinel, Closest& best) {
    for (int j = 0; j < count; ++j) {
        const f32x4 s = S.sph[first + j];
        float h;
        const float disc = sphere_discriminant(r, s.x, s.y, s.z, s.w, h);
        if (disc > 0.00001f) {
            const float t = sphere_root(h, disc, t_min, sentinel);
            if (t < best.t) { best.t = t; best.group = g; best.prim = first + j; }
        }
    }
}

__device__ __forceinline__ void lane_triangles(SceneArg S, int g, int first, int count, const Ray& r,
                                               float t_min, float t_max, Closest& best) {
    for (int j = 0; j < count; ++j) {
        const f32x4 a = S.tri[2 * (first + j)], b = S.tri[2 * (first + j) + 1];
        const TriTerms k = triangle_terms(r, a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, S.tri_bz[first + j]);
        float t;
        if (triangle_exact(k, t_min, t_max, t) && t < best.t) { best.t = t; best.group = g; best.prim = first + j; }
    }
}

// The same for the lanes of a wavefront that are at the same small group: `first` and `count` are wave-uniform and the
// primitives come through scalar loads.
__device__ __forceinline__ void uniform_spheres(SceneArg S, int g, int first, int count, const Ray& r,
                                                float t_min, float sentinel, Closest& best) {
    const WRT_CONST f32x4* sph = as_uniform(S.sph) + first;
    for (int j = 0; j < count; ++j) {
        const f32x4 s = sph[j];
        float h;
        const float disc = sphere_discriminant(r, s.x, s.y, s.z, s.w, h);
        if (disc > 0.00001f) {
            const float t = sphere_root(h, disc, t_min, sentinel);
            if (t < best.t) { best.t = t; best.group = g; best.prim = first + j; }
        }
    }
}

__device__ __forceinline__ void uniform_triangles(SceneArg S, int g, int first, int count, const Ray& r,
                                                  float t_min, float t_max, Closest& best) {
    const WRT_CONST f32x4* tri = as_uniform(S.tri) + 2 * first;
    const WRT_CONST float* bz = as_uniform(S.tri_bz) + first;
    for (int j = 0; j < count; ++j) {
        const f32x4 a = tri[2 * j], b = tri[2 * j + 1];
        const TriTerms k = triangle_terms(r, a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, bz[j]);
        float t;
        if (triangle_exact(k, t_min, t_max, t) && t < best.t) { best.t = t; best.group = g; best.prim = first + j; }
    }
}

// One ray per lane against groups [g_begin, g_end), every lane walking the
// groups together (stage entry point wrt_intersect).  Every lane of the
// wavefront must call this; `live` masks lanes without a ray.
__device__ __forceinline__ void closest_hit(SceneArg S, int g_begin, int g_end, bool live, const Ray& r,
                                            float t_min, float t_max, LaneQueue& q, Closest& best, PairCounts& pc) {
    const float sentinel = t_max + 1.0f;
    best.t = sentinel; best.group = -1; best.prim = -1;
    const WRT_CONST GroupRec* groups = as_uniform(S.groups);
    const unsigned long long live_mask = __ballot(live);
    if (live_mask == 0ull) return;
    const unsigned n_live = (unsigned)__popcll(live_mask);
    for (int g = g_begin; g < g_end; ++g) {
        const int type = groups[g].type, first = groups[g].first, count = groups[g].count;
        if (type == GROUP_SPHERES) {
            pc.sphere += (unsigned long long)n_live * (unsigned)count;
            pc.sphere_slots += 64ull * (unsigned)count;
            if (!(S.hierarchy != 0 && groups[g].tree_base >= 0 && count >= S.hierarchy &&
                  walk_spheres(S, groups[g], g, live, r, t_min, t_max, q, best, pc.sphere_exact)))
                loop_spheres(S, as_uniform(S.pairs) + groups[g].pair_base, g, first, count, live, r, groups[g].c[0],
                             groups[g].c[1], groups[g].c[2], t_min, sentinel, q, best);
        } else if (type == GROUP_TRIANGLES) {
            const bool pass = live && cull_pass(r, groups[g].c[0], groups[g].c[1], groups[g].c[2], groups[g].c[3]);
            const unsigned long long pass_mask = __ballot(pass);
            pc.cull += n_live;
            if (pass_mask == 0ull) continue;
            pc.triangle += (unsigned long long)__popcll(pass_mask) * (unsigned)count;
            pc.triangle_slots += 64ull * (unsigned)count;
            const SphereReject k = sphere_reject_setup(r, pass, groups[g].edge_max, groups[g].reach,
                                                       groups[g].reject_eps, groups[g].c[0], groups[g].c[1],
                                                       groups[g].c[2]);
            bool walked = false;
            if (S.hierarchy != 0 && groups[g].tree_base >= 0 && count >= S.hierarchy) {
                const TreeRay w = tree_ray_setup(r, pass, groups[g].reach2, groups[g].c[0], groups[g].c[1],
                                                 groups[g].c[2]);
                if (__ballot(!w.usable) == 0ull) {
                    walk_tree_together(S.tree + groups[g].tree_base, groups[g].tree_leaf_shift, g, first, w, r, pass,
                                       t_min, t_max, q, best, pc.triangle_exact);
                    walked = true;
                }
            }
            if (walked) {
            } else if (S.prefilter && __ballot(!k.usable) == 0ull)
                loop_triangles_rejecting(S, as_uniform(S.bounds) + groups[g].bound_base, g, first, count, k, r, t_min,
                                         t_max, q, best, pc.triangle_exact);
            else
                loop_triangles(S, as_uniform(S.pairs) + groups[g].pair_base, g, first, count, pass, r, t_min, t_max, q,
                               best);
        } else {
            pc.disk += n_live;
            float c[4], n[4];
            for (int a = 0; a < 4; ++a) { c[a] = groups[g].c[a]; n[a] = groups[g].n[a]; }
            float t;
            if (disk_hit(r, c, n, t_min, t_max, sentinel, t) && live && t < best.t) {
                best.t = t; best.group = g; best.prim = 0;
            }
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
template <bool TRIANGLES = true>
__device__ __forceinline__ void shade_record(SceneArg S, const Closest& best, const Ray& r, HitRecord& h) {
    const float ox = r.ox, oy = r.oy, oz = r.oz, dx = r.dx, dy = r.dy, dz = r.dz;
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
    } else if (TRIANGLES && G.type == GROUP_TRIANGLES) {
        const f32x4 r0 = S.tri[2 * best.prim], r1 = S.tri[2 * best.prim + 1];
        const float Ax = r0.w, Ay = r1.x, Az = r1.y, Bx = r1.z, By = r1.w, Bz = S.tri_bz[best.prim];
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

// x ** 5 as numpy evaluates it for a float64 array (materials.py:369): pow(x, 5.0) of the C library
// numpy is built against.  Here the product is carried in two doubles (error-free products through
// fma, relative error below 2^-100) and rounded once: the correctly rounded fifth power.  The C
// library's pow is not correctly rounded for every argument (glibc documents an error below 0.52 ulp;
// tests/test_fifth_power.py counts the arguments on which it differs from this function, by one
// unit in the last place), and a last-place difference of the reflectance changes a path only if the
// path's uniform draw falls between the two values, a chance of about 2^-52 per refracting hit.
// (__host__ as well, so that the test can run the function itself on the CPU.)
__host__ __device__ __forceinline__ double fifth_power(double x) {
    const double h2 = x * x, l2 = __builtin_fma(x, x, -h2);
    const double h4 = h2 * h2, l4 = __builtin_fma(h2, h2, -h4) + 2.0 * (h2 * l2);
    const double h5 = h4 * x, l5 = __builtin_fma(h4, x, -h5) + l4 * x;
    return h5 + l5;
}

// does <Material>.scatter draw a random unit vector (numpy_random_unit_vecs, materials.py:62,274)?
__device__ __forceinline__ bool draws_unit_vector(int kind, int fuzzy) {
    return (kind >= 1 && kind <= 5) || (kind == 6 && fuzzy != 0);
}

// <Material>.scatter (materials.py) for one hit, float64 where main.render()'s
// float64 hit arrays make the reference evaluate in float64.  (vx, vy, vz): the path's random unit
// vector of this bounce (wave_unit_vectors), read only by materials for which draws_unit_vector() holds.
__device__ __forceinline__ void scatter_one(SceneArg S, const MaterialRec& M, const PathKey& key,
                                            uint32_t bounce, float dx, float dy, float dz,
                                            const HitRecord& h, float vx, float vy, float vz, Scattered& out) {
    const double nx = h.n[0], ny = h.n[1], nz = h.n[2];
    out.absorbed = false;
    if (M.kind >= 1 && M.kind <= 5) {
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
        const double reflectance = (double)r0 + (double)(1.0f - r0) * fifth_power(1.0 - cos_t);
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
