/*
 * wrt.h - C ABI of libwrt.so, the MI355X (gfx950) implementation of the
 * vectorised render path of ninezerozeronine/raytracing-one-weekend
 * (reference package src/weekend_raytracer).
 *
 * The reference is pure Python + numpy and has no FFI layer of its own; each
 * entry point below names the reference function whose work it takes over.
 * The Python package `weekend_raytracer` in this repository binds these
 * symbols with ctypes and keeps the reference's class API on top of them.
 *
 * Conventions
 *   - every function returns 0 on success and a negative WRT_E_* code on
 *     failure; wrt_last_error() gives the text of the last failure of that
 *     context (or of wrt_create when passed NULL);
 *   - one context per GPU, used by one host thread at a time; a call makes the
 *     context's device current while it runs and puts the caller's back;
 *   - "device pointer" = memory of the context's GPU owned by the caller
 *     (e.g. torch.Tensor.data_ptr()); "host-or-device pointer" = either,
 *     the library copies out of it before returning;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *     launches are asynchronous on it, nothing is synchronised unless said.
 *     A context's launches share its scene and workspaces: a launch on another
 *     stream than the context's previous one is ordered behind it on the
 *     device, and wrt_scene_commit, wrt_set_texture and wrt_destroy wait on
 *     the host for the context's last launch before they overwrite or free
 *     what it reads.  A call on a stream that is being captured into a HIP
 *     graph is left out of that bookkeeping (the context's event is neither
 *     recorded nor waited for inside the capture): the caller orders the
 *     graph's replays against wrt_scene_commit, wrt_set_texture and
 *     wrt_destroy, e.g. by synchronising the replay stream first;
 *   - arithmetic follows the reference operation by operation: float32
 *     intersection without fused multiply-adds, float64 scatter arithmetic
 *     on float32 values, same thresholds and tie-breaks.
 */
#ifndef WRT_H
#define WRT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wrt_ctx wrt_ctx;

enum {
    WRT_OK = 0,
    WRT_E_ARG = -1,      /* bad argument (null pointer, size, index out of range) */
    WRT_E_HIP = -2,      /* a HIP runtime call failed */
    WRT_E_STATE = -3,    /* call order (e.g. render before scene commit) */
    WRT_E_LIMIT = -4     /* a documented limit was exceeded */
};

enum {
    WRT_MAX_BOUNCES = 62,     /* per-bounce statistics slots */
    WRT_MAX_MATERIALS = 256,  /* uint8 material indices (sphere_group.py:30) */
    WRT_MAX_TEXTURES = 16,
    WRT_MAX_GROUPS = 4096,
    WRT_MAX_GROUP_TRIANGLES = 1 << 20,   /* per triangle group (split larger meshes into several groups) */
    WRT_MAX_GROUP_SPHERES = 1 << 24,     /* per sphere group (likewise) */
    WRT_MAX_SCENE_PRIMITIVES = 1 << 26,  /* spheres, and triangles, over all groups of a scene */
    WRT_STATIC_MAX_PIXELS = 4 << 20      /* width * height limit of wrt_render_static (64 MiB of path records) */
};

/* materials.py: one entry per class */
enum {
    WRT_MAT_NONE = 0,             /* index absent from the material map (main.py:186-200: never scattered) */
    WRT_MAT_DIFFUSE = 1,          /* Diffuse                    materials.py:35-78   */
    WRT_MAT_TEXTURED = 2,         /* TexturedDiffuse            materials.py:81-143  */
    WRT_MAT_CHECKER = 3,          /* CheckerboardDiffuse        materials.py:146-183 */
    WRT_MAT_NORMAL_RGB = 4,       /* NormalToRGBDiffuse         materials.py:186-207 */
    WRT_MAT_NORMAL_DISCRETE = 5,  /* NormalToDiscreteRGBDiffuse materials.py:210-234 */
    WRT_MAT_METAL = 6,            /* Metal                      materials.py:237-295 */
    WRT_MAT_DIELECTRIC = 7        /* Dielectric                 materials.py:298-398 */
};

typedef struct wrt_material {
    int32_t kind;          /* WRT_MAT_* */
    int32_t texture;       /* TEXTURED: slot given to wrt_set_texture */
    float colour_a[3];     /* DIFFUSE / METAL colour; CHECKER colour_a */
    float colour_b[3];     /* CHECKER colour_b */
    double scale[3];       /* CHECKER */
    double offset[3];      /* CHECKER */
    double fuzziness;      /* METAL */
    double ior;            /* DIELECTRIC index_of_refraction */
} wrt_material;

typedef struct wrt_stats {
    /* [b] = rays that entered bounce b, b = 0..max_bounces: the numbers the
       reference prints at main.py:111.  Mrays/s counts sum over b < max_bounces. */
    uint64_t rays_per_bounce[WRT_MAX_BOUNCES + 2];
    uint64_t paths;            /* primary rays traced */
    uint64_t sphere_pairs;     /* (ray, sphere) pairs the reference would evaluate  */
    uint64_t triangle_pairs;   /* (ray, triangle) pairs after the bounding-sphere cull */
    uint64_t cull_tests;       /* (ray, triangle group) bounding-sphere tests */
    uint64_t disk_pairs;
    /* lane slots spent in the wave-uniform group loops (64 x primitives per run; none for a
       group walked through its hierarchy); pairs evaluated there / slots = lane efficiency of
       the dominant loops */
    uint64_t sphere_lane_slots;
    uint64_t triangle_lane_slots;
    /* clock tallies of the bounce kernel in shader clock cycles: [0] sum of the wavefronts'
       lifetimes, [1] longest lifetime (over the kernel's duration: the clock it ran at); and,
       zero unless the library was built with -DWRT_TIMING=1: [2] in triangle loops, [3] in
       sphere loops, [4] shading, [5] refilling lanes, [6] per-lane group walk, [7] working
       candidate queues off (part of [2] and [3]) */
    uint64_t clock_ticks[8];
    /* (ray, triangle) pairs of the large groups that went through the full evaluation;
       the others were rejected by the bounding-sphere test (see wrt_set_triangle_reject)
       or never looked at by a hierarchy walk (see wrt_set_hierarchy) */
    uint64_t triangle_evaluations;
    /* (ray, sphere) pairs a hierarchy walk of a sphere group evaluated (zero while sphere groups go
       through the uniform loop, which looks at every pair) */
    uint64_t sphere_evaluations;
    /* wavefronts of the bounce kernel's launches whose lifetime was measured, the count behind
       clock_ticks[0] (a wavefront whose clock reads came out of order is left out of both) */
    uint64_t wavefronts;
    /* times a path was put on a lane's shelf (see wrt_set_path_shelf); zero for renders without it */
    uint64_t shelved_paths;
    /* paths the bounce kernel ended because their state named a primitive, pixel or sample that does not exist
       (the kernel checks before it follows a hit into the scene's arrays or writes a path's record): always
       zero unless something has damaged a path; a render with a non-zero count is not to be trusted */
    uint64_t damaged_paths;
} wrt_stats;

/* ---- context ---------------------------------------------------------- */
int wrt_create(int device, wrt_ctx** out);
int wrt_destroy(wrt_ctx* ctx);
const char* wrt_last_error(const wrt_ctx* ctx);
/* number of compute units of the context's device (for sizing workloads) */
int wrt_device_cus(const wrt_ctx* ctx);

/* ---- scene (host side of L2/L3 stays in Python; these take its arrays) -- */
int wrt_scene_clear(wrt_ctx* ctx);

/* SphereGroup (sphere_group.py:14-45): centres [K*3], radii [K], material
   index per sphere [K]; host-or-device pointers. Groups are intersected in
   the order they are added (main.py:124, first group wins ties). */
int wrt_add_sphere_group(wrt_ctx* ctx, const float* centres, const float* radii,
                         const uint8_t* materials, int count, int* group_id);

/* TriangleGroup (triangle_group.py:31-119): pt0, A = pt1-pt0, B = pt2-pt0,
   per-vertex normals [F*3] and uvs [F*2], one material per group, bounding
   sphere as update_sphere_bounds computes it; host-or-device pointers. */
int wrt_add_triangle_group(wrt_ctx* ctx, const float* pt0, const float* A, const float* B,
                           const float* n0, const float* n1, const float* n2,
                           const float* uv0, const float* uv1, const float* uv2,
                           int count, int material_id,
                           const float bounds_centre[3], float bounds_radius, int* group_id);

/* Disk (disk.py:13-46): unit normal and the U, V axes the constructor derives. */
int wrt_add_disk(wrt_ctx* ctx, const float centre[3], const float normal[3],
                 const float U[3], const float V[3], double radius, int material_index,
                 int* group_id);

/* material_map (scenes.py): table[i] describes material index i, i < count <= 256 */
int wrt_set_materials(wrt_ctx* ctx, const wrt_material* table, int count);

/* TexturedDiffuse.__init__ (materials.py:82-109) result: float32 RGB in [0,1],
   [height][width][3], already flipped vertically; host-or-device pointer */
int wrt_set_texture(wrt_ctx* ctx, int slot, const float* rgb, int width, int height);

/* Camera.__init__ (camera.py:13-79) results */
int wrt_set_camera(wrt_ctx* ctx, const double pos[3], const double U[3], const double V[3],
                   const double bottom_left[3], const double horiz[3], const double vert[3],
                   double lens_radius);

/* HORIZON_COLOUR / SKY_COLOUR (main.py:21-22) */
int wrt_set_sky(wrt_ctx* ctx, const float horizon[3], const float zenith[3]);

/* uploads the scene; required after any change and before the calls below */
int wrt_scene_commit(wrt_ctx* ctx);

/* ---- the hot path: main.render() bounce loop (main.py:93-220) ----------- */
/* Traces samples [s_begin, s_end) of every pixel with keyed random streams
   (counter = pixel x, y, global sample index, purpose/bounce/attempt; key =
   seed) and ADDS to accum[(x*height + y)*3 + c] (device pointer, float32, the
   reference's x-major pixel order, y = 0 at the bottom), one sample after the
   other in sample order, each sample's product of the per-bounce colours:
   consecutive calls for [a, b) and [b, c) leave the same bits as one for [a, c).
   path_colours: optional device pointer [width*height*(s_end-s_begin)][4]
   receiving (r, g, b, segments traced [+100 when the path was still alive at
   the bounce limit]) per path in ray-id order
   r = (x*height + y)*S + s (main.py:102-104); NULL = library workspace.
   Statistics accumulate in the context (wrt_stats_read). */
int wrt_render(wrt_ctx* ctx, int width, int height, int s_begin, int s_end, int max_bounces,
               float t_min, float t_max, uint64_t seed, float* accum, float* path_colours,
               void* stream);

/* wrt_render for callers that cannot allocate device memory.  It has no counterpart in the
   reference; it is for code under HIP graph capture, tight serving loops and harnesses that
   forbid hipMalloc.  Same arguments (without path_colours), same arithmetic and the same accum
   bit for bit, but the per-launch workspace (16 bytes per path and the batch counter) is a
   static device array of the library: 256 + 16 * WRT_STATIC_MAX_PIXELS bytes (64 MiB) that every
   GPU the library has a context on holds until the process ends.  The call allocates and frees
   nothing, on the device or through any library routine; it renders as many samples per launch as
   fit the array.  width * height > WRT_STATIC_MAX_PIXELS returns WRT_E_LIMIT.  All contexts of a
   GPU share that GPU's array: calls that overlap on one GPU (two streams, two contexts, two
   threads) race; calls on one stream, or ordered by the caller, do not.  Kernel timing
   (wrt_profile_enable) does not cover this entry point.  Under stream capture the call records
   its launches (one memset node and two kernel nodes per batch of samples) and nothing else; a
   replay reads the scene that is committed when it runs, see the note on captured streams above. */
int wrt_render_static(wrt_ctx* ctx, int width, int height, int s_begin, int s_end, int max_bounces,
                      float t_min, float t_max, uint64_t seed, float* accum, void* stream);

/* image = sqrt(accum / spp_total)  (main.py:219-220); n_values = width*height*3 */
int wrt_finalize(wrt_ctx* ctx, const float* accum, int64_t n_values, int spp_total,
                 float* image, void* stream);

/* generate_image_from_data (main.py:45-74) without the per-pixel Python loop: image is the
   [width][height][3] float32 result of wrt_finalize (y = 0 at the bottom), rgb receives
   [height][width][3] bytes, row 0 at the top, each int(255 * v) as a float32 product truncated
   towards zero and clipped to 0..255 (what PIL stores for the reference's fill tuple); NaN, on
   which the reference raises, is stored as 0.  Device pointers. */
int wrt_image_bytes(wrt_ctx* ctx, const float* image, int width, int height, uint8_t* rgb, void* stream);

/* No counterpart in the reference, which evaluates triangle_group.py:289-395 in full
   for every (ray, triangle) pair.  on = 1 (default): in groups of more than 8 triangles
   every pair first meets a conservative bounding-sphere test of that one triangle and
   only the survivors get the full evaluation; results are the same bit for bit.
   on = 0: the full evaluation for every pair.  Takes effect with the next render. */
int wrt_set_triangle_reject(wrt_ctx* ctx, int on);

/* No counterpart in the reference (a scheduling choice of the bounce kernel, main.py:110-212 has
   no such thing): with the shelf, a lane of wrt_render's bounce kernel keeps a second path in device
   memory.  A path that waits for a large group's loop while its wavefront has other work goes
   there and the lane starts a fresh one, so that a loop finds a taker in more lanes when it runs.
   mode = -1 (default): used for scenes with two or more large triangle groups, where loops compete
   for a wavefront's lanes (measured +7 % on two meshes of about 6 000 triangles; with one mesh the
   shelf costs 2 to 3 %); 0: never; 1: for every scene with a triangle group.  A path is the same
   whichever lane traces it and whenever, so the results are the same bit for bit in every mode.
   wrt_render_static has no shelf.  Takes effect with the next render. */
int wrt_set_path_shelf(wrt_ctx* ctx, int mode);

/* No counterpart in the reference either (its README lists a spatial structure as future
   work).  min_primitives = 0 (default): every (ray, primitive) pair of a group is examined (for a
   triangle group: of a group that passes its bounding-sphere cull, triangle_group.py:158-166), as
   the reference does.  min_primitives = n > 0: triangle groups and sphere groups of n or more
   primitives (at least 16, at most 32768) are walked through a box hierarchy built by
   wrt_scene_commit; only the primitives of the leaves a ray comes near get the evaluation of
   triangle_group.py:289-395 / sphere_group.py:137-205 (up to four spheres much larger than a
   group's median stay outside its hierarchy and are looked at by every ray).  The boxes are grown
   by a bound on that evaluation's float32 error, so the closest hit (t, primitive, tie order) is
   the same bit for bit; the work done is not the reference's, and throughput in this mode is to
   be reported apart from the brute-force figure.  On one MI355X the walk overtakes the
   brute-force loops at about 600 triangles or 400 spheres per group.  Takes effect with the
   next render or intersect. */
int wrt_set_hierarchy(wrt_ctx* ctx, int min_primitives);

int wrt_stats_reset(wrt_ctx* ctx, void* stream);
/* synchronises `stream`, then copies the statistics out */
int wrt_stats_read(wrt_ctx* ctx, wrt_stats* out, void* stream);

/* Measurement aid (the reference times render() with perf_counter, main.py:93,229):
   while enabled, every launch of the bounce kernel inside wrt_render is
   bracketed by HIP events on the launch stream; wrt_profile_read waits for
   them, returns the summed kernel time and launch count since the last read,
   and clears them. */
int wrt_profile_enable(wrt_ctx* ctx, int on);
int wrt_profile_read(wrt_ctx* ctx, double* kernel_ms, int* launches);

/* ---- stage entry points (same device code as wrt_render) ---------------- */
/* Camera.get_ray_components (camera.py:108-190): origins, dirs: device
   pointers [width*height*(s_end-s_begin)][3] float32 in ray-id order */
int wrt_generate_rays(wrt_ctx* ctx, int width, int height, int s_begin, int s_end,
                      uint64_t seed, float* origins, float* dirs, void* stream);

/* SphereGroup/TriangleGroup/Disk.get_hits (sphere_group.py:72, triangle_group.py:150,
   disk.py:48) for group >= 0, or the merge over all groups of main.py:116-173
   for group = -1.  n rays; origins/dirs device [n][3].  Outputs are device
   pointers, any may be NULL: hit [n] u8, t [n], p [n][3], normal [n][3],
   uv [n][2], material [n] u8, back [n] u8, hit_group [n] i32, prim [n] i32
   (index inside its group; -1 when no hit).  For rays without a hit the
   record is zero and t is the sentinel the reference leaves (t_max+1 for one
   group, 5001 for the merge). */
int wrt_intersect(wrt_ctx* ctx, int group, const float* origins, const float* dirs, int64_t n,
                  float t_min, float t_max, uint8_t* hit, float* t, float* p, float* normal,
                  float* uv, uint8_t* material, uint8_t* back, int32_t* hit_group,
                  int32_t* prim, void* stream);

/* <Material>.scatter (materials.py) with the dtypes main.render() passes:
   incoming directions, hit points, normals [n][3], uvs [n][2] (float32 values;
   evaluated in float64 where the reference does), back [n] u8, and the path
   identity px, py, ps [n] i32 + bounce that key the random draws.
   Outputs: new directions [n][3], colours [n][3], absorbed [n] u8.  The new
   ray origin is the hit point itself. All device pointers. */
int wrt_scatter(wrt_ctx* ctx, int material_index, const float* dirs, const float* points,
                const float* normals, const float* uvs, const uint8_t* back,
                const int32_t* px, const int32_t* py, const int32_t* ps, int bounce,
                uint64_t seed, int64_t n, float* out_dirs, float* out_colours,
                uint8_t* absorbed, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WRT_H */
