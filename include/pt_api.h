/*
 * pt_api.h - C ABI of libptcore, the MI355X render core.
 *
 * The reference application has no callable render API: a render is started, polled and aborted
 * by three code blocks inside its frame function that operate on fields of one big AppData
 * struct.  Every entry point below names the reference block it replaces.  All structs are plain
 * data (no pointers owned by the caller survive a call, the library deep-copies its inputs), all
 * entry points return 0 on success or a negative PT_ERR_* code and never exit or throw.
 */
#ifndef PT_API_H
#define PT_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Geometry types, numbered as the reference's GEOM_TYPE_* enum (code/raycast.cpp:5-13). */
enum {
	PT_GEOM_SPHERE = 0,   /* ellipsoid; a true sphere when the three dims are equal */
	PT_GEOM_BOX = 1,
	PT_GEOM_CYLINDER = 2,
	PT_GEOM_CONE = 3,
	PT_GEOM_PLANE = 4,    /* the infinite plane through pos with normal rot * (0,0,1) (code/raycast.cpp:885-889); dim is ignored */
};

enum {
	PT_OK = 0,
	PT_ERR_ARG = -1,       /* invalid argument / unsupported value */
	PT_ERR_DEVICE = -2,    /* HIP runtime error, see pt_last_error */
	PT_ERR_BUSY = -3,      /* a render is already active on this context */
	PT_ERR_STATE = -4,     /* call made in the wrong state (e.g. trace before prepare) */
	PT_ERR_NOMEM = -5,
	PT_ERR_LIMIT = -6,     /* size beyond a documented maximum */
};

/* Seed policies.  The reference reseeds from the cycle counter per tile job and so defines no
 * reproducible sequence (code/raycast.cpp:994); these are the two deterministic replacements. */
enum {
	PT_SEED_PER_JOB = 1,    /* one PCG stream per tile job, pixels and samples consumed serially
	                           inside the job - the reference's structure; CPU oracle only */
	PT_SEED_PER_SAMPLE = 3, /* one PCG stream per pixel (lens table start) and one per camera
	                           sample (all bounce draws); device path and CPU oracle */
};

/* Authoring fields of one entity: Object / Material / Geometry of code/raycast.cpp:23-78 without
 * the precalculated members, which the library derives itself (code/raycast.cpp:2064-2108). */
typedef struct pt_object {
	float pos[3];
	float rot[4];            /* quaternion, order w,x,y,z (code/rt_math.cpp:2789-2801) */
	float dim[3];            /* sphere: diameters; box: edge lengths; cylinder/cone: x,y diameters, z height */
	float color[3];          /* gamma-2.2 space */
	int32_t type;            /* PT_GEOM_* */
	float emit_color[3];     /* gamma-2.2 space */
	float reflection_mod;    /* 0 diffuse .. 1 mirror */
	float refractive_index;  /* 1 = opaque */
	float roughness;         /* Oren-Nayar, 0 = off */
} pt_object;

/* Camera of code/raycast.cpp:148-156; image-plane height, plane distance and the orientation
 * vectors are derived (code/app.cpp:1077-1090). */
typedef struct pt_camera {
	float pos[3];
	float rot[3];            /* x = yaw about +Z, y = pitch, z unused by the reference */
	float fov;               /* degrees */
	float plane_width;       /* Camera::dim.w, image-plane width in world units */
} pt_camera;

/* World of code/raycast.cpp:245-269 without the object array and the derived members. */
typedef struct pt_world {
	pt_camera camera;
	float background_color[3];   /* gamma space */
	float ambient_color[3];
	float light_color[3];
	int32_t light_rot[2];        /* degrees, World::globalLightRot */
	float focal_distance;
	float aperture;              /* 0 = pinhole */
} pt_world;

/* RaytraceSettings of code/raycast.cpp:320-358 (authoring members) plus the constants the
 * reference hard-codes, with the reference value noted. */
typedef struct pt_settings {
	int32_t width, height;       /* texDim; reference UI clamps to 256..10000 */
	int32_t spp_wanted;          /* sampleCountWanted (code/app.cpp:641) */
	int32_t sample_grid_width;   /* blue-noise tile width in pixels, 3 (code/app.cpp:642) */
	int32_t max_bounces;         /* rayBouncesMaxWanted */
	int32_t grid_hint;           /* cells per axis asked of spatialGridInit, 10 (code/app.cpp:1276) */
	int32_t brute_force_below;   /* object count under which every object is tested, 10 (code/raycast.cpp:1100) */
	int32_t seed_policy;         /* PT_SEED_* */
	uint64_t seed;               /* tables + blue noise are drawn from PCG stream (0, seed);
	                                path streams use it as their initstate */
	int32_t part_index;          /* this context renders the image tiles t with t % part_count == part_index */
	int32_t part_count;          /* 1 = whole image */
	int32_t frame_index;         /* progressive mode: >= 0 traces ONE sample per pixel, sample
	                                (frame_index mod count(tile)), and adds it into the accumulation
	                                buffer; -1 = full render of every sample */
	int32_t sample_ranges;       /* a pixel's samples are summed in this many contiguous ranges (range r of n samples is
	                                [r*n/ranges, (r+1)*n/ranges), each summed in sample order; the range sums are then added
	                                in range order).  A power of two up to 16, or 0 = chosen from spp_wanted by pt_sample_ranges().  The
	                                reference adds all samples in one sequence (= 1 range, code/raycast.cpp:1263); more
	                                ranges only change the rounding of that sum. */
} pt_settings;

typedef struct pt_progress {
	int32_t done;                /* 1 once the frame is complete (or aborted) */
	int32_t aborted;
	float fraction;              /* 0..1, share of the frame's work groups (8x8 tile x sample range) handed out so far;
	                                after an abort: the share of pixels that were finished */
	double seconds;              /* since pt_render_start, the reference's processTime (code/app.cpp:1440) */
	uint64_t paths_done;         /* while the frame is in flight: an estimate, the frame's camera samples times `fraction`;
	                                once done (aborted or not): the exact count of the finished pixels' camera samples */
} pt_progress;

typedef struct pt_stats {
	uint64_t paths;              /* camera samples in this context's partition (exact) */
	uint64_t rays;               /* ray casts issued on the device */
	uint64_t pixels;             /* pixels in this context's partition */
	uint64_t zero_sample_pixels; /* pixels whose blue-noise tile drew no sample (written as 0) */
	int32_t samples_total;       /* entries in the blue-noise sample table */
	int32_t objects;
	int32_t cells[3];
	int32_t cell_size[3];
	int32_t cell_items;          /* entries in the CSR item array */
	int32_t grid_blocks;         /* workgroups of the last trace launch (a persistent grid sized to the device) */
	float trace_ms;              /* mean device time per pt_trace since the last synchronisation (HIP events; of the launches
	                                since then the first four and every eighth are timed, PT_EVENT_EVERY=1 times all) */
	float dominant_kernel_ms;    /* summed device time of the extend/shade kernel in the last trace */
	float prepare_ms;            /* host flatten + grid + tables + upload */
	float trace_ms_min;          /* fastest and slowest of the launches folded into trace_ms */
	float trace_ms_max;
	int32_t trace_launches;
} pt_stats;

typedef struct pt_ctx pt_ctx;

/* Library lifetime.  `device` is a HIP device ordinal. */
pt_ctx* pt_create(int device);
void pt_destroy(pt_ctx* ctx);
const char* pt_last_error(const pt_ctx* ctx);   /* ctx may be NULL: error of a failed pt_create */
void pt_default_settings(pt_settings* s);

/* The demo scene of code/raycast.cpp:1294-1356 (ground box + mirror sphere). objects must hold 2. */
int pt_default_scene(pt_world* world, pt_object* objects, int capacity, int* count);

/*
 * Replaces the render-start block code/app.cpp:1140-1423: validates, deep-copies the inputs,
 * runs per-object precalc, builds sample tables and the uniform grid, uploads, launches the
 * tracer on a worker thread and returns immediately.  `rgb` (width*height*3 floats, linear,
 * clamped to [0,1], row 0 = top) is caller-owned and filled on completion, like AppData::buffer
 * (code/app.cpp:1156-1159, code/raycast.cpp:1270).
 */
int pt_render_start(pt_ctx* ctx, const pt_world* world, const pt_object* objects, int n_objects,
                    const pt_settings* settings, float* rgb);

/* Optional replacement of the image allocation code/app.cpp:1156-1159: width*height*3 floats of page-locked host
 * memory.  Any host buffer works as `rgb`; into one from here the finished image is copied by the copy engine at
 * the speed of the bus (0.5 ms for 1080p), into pageable memory through a staging image and host threads. */
float* pt_image_alloc(pt_ctx* ctx, int width, int height);
void pt_image_free(pt_ctx* ctx, float* image);

/* Replaces the per-frame poll code/app.cpp:1436-1440 (threadQueueFinished + processTime). */
int pt_render_progress(pt_ctx* ctx, pt_progress* info);

/* Replaces the Escape handler code/app.cpp:1425-1428.  The trace kernel stops handing out work groups
 * (one wavefront in 64 group fetches looks at the flag); pixels already in flight finish.  As in the
 * reference, the caller's buffer is partial afterwards: pixels that were finished hold their final
 * value, every other pixel keeps what the caller had there before pt_render_start. */
int pt_render_abort(pt_ctx* ctx);

/* Replaces the per-frame re-upload of the half-written buffer (code/app.cpp:1456-1461): copies the pixels
 * finished so far into `rgb` (width*height*3 floats, normally the buffer given to pt_render_start; pixels
 * not finished yet are left untouched) and reports how many there are.  Callable while a render is
 * active, from the polling thread; after completion it copies nothing new.  Pixels become visible as
 * the device writes their sums through to memory, so the count trails the render by a little. */
int pt_render_snapshot(pt_ctx* ctx, float* rgb, uint64_t* pixels_done);

/* Blocks until the active render is done (headless harness). Returns the render's status. */
int pt_render_wait(pt_ctx* ctx);

/*
 * The same work split in three, for callers that keep the scene resident (benchmarks,
 * progressive display, torch interop):
 *   pt_prepare  - host flatten + tables + grid + upload  (code/app.cpp:1161-1357)
 *   pt_trace    - all device passes into a device framebuffer (code/raycast.cpp:991-1289)
 *   pt_download - device framebuffer -> host
 * d_rgb may be NULL (library-owned framebuffer) or a device pointer to width*height*3 floats;
 * hip_stream is a hipStream_t, or NULL for the context's own stream - a non-blocking stream, NOT the
 * legacy null stream: work queued on the null stream (torch's default stream has handle 0) is not
 * ordered against it, so a caller that orders other work against the trace passes its own stream.
 * pt_trace is synchronous with respect to the host only when `sync` is non-zero.  It clears a pending
 * abort request left by an aborted pt_render_start.
 */
int pt_prepare(pt_ctx* ctx, const pt_world* world, const pt_object* objects, int n_objects,
               const pt_settings* settings);
int pt_trace(pt_ctx* ctx, float* d_rgb, void* hip_stream, int sync);
int pt_sync(pt_ctx* ctx);                                /* waits for queued pt_trace calls, updates the statistics */
int pt_set_frame(pt_ctx* ctx, int frame_index);          /* progressive mode frame counter */
int pt_download(pt_ctx* ctx, float* rgb);
int pt_get_stats(const pt_ctx* ctx, pt_stats* stats);

/*
 * Debug outputs the parity contract needs and the reference lacks: for camera sample 0 of every
 * pixel, the 1-based id of the entity the primary ray hits (0 = miss) and a hash of the grid
 * cells the ray visited (0 when the brute-force path was taken).  Runs its own small pass over
 * the prepared scene.
 */
int pt_debug_primary(pt_ctx* ctx, int32_t* hit_id, uint64_t* cell_hash);
/* Metrics the reference's statistics panel lacks and a headless run wants (the harness's bounce histogram): histogram[k],
 * 1 <= k < bins, = camera samples of this context's partition whose path casts k rays in the frame the prepared scene
 * renders (k <= max_bounces; the last bin also takes every longer path; 2 <= bins <= 65).  The bins add up to
 * pt_stats.paths, and k * histogram[k] to pt_stats.rays.  A pass of its own over every sample with the tracer's own
 * device functions and random streams (on a grid scene it takes several times a frame's time). */
int pt_cast_histogram(pt_ctx* ctx, uint64_t* histogram, int bins);
/* Diagnostic builds of the library only (-DPT_COOP_STATS / -DPT_COOP_TIMING): copies out and clears 16
 * event or cycle counters of the trace kernel; returns 16, or 0 in a normal build (nothing written).
 * No counterpart in the reference. */
int pt_debug_coop_stats(pt_ctx* ctx, unsigned long long* counters16);

/* Debug read-back of the cost estimate the prepared scene's 8x8 tiles are dealt by (dearest first): ray casts of sixteen
 * probe paths per owned tile, in the partition's tile order (local tile t is image tile t * part_count + part_index),
 * capped at 1023.  No counterpart in the reference (its tile jobs go out in a fixed spiral, code/app.cpp:1359-1419); tests
 * use it to place their comparisons where the frame is dearest.  tiles receives the count (0 when the partition owns fewer
 * than two tiles); cost may be NULL. */
int pt_debug_tile_cost(pt_ctx* ctx, uint16_t* cost, int capacity, int* tiles);

/* Debug read-back of the uniform grid the prepared scene uses (the reference's per-cell ObjectLists,
 * code/app.cpp:1278-1354, in CSR form; built on the device for scenes of 4096 entities or more, on
 * the host otherwise; the environment variable PT_GRID_BUILD=host|device overrides the choice).
 * cell_start: cells + 1 words; cell_items: the lists, ascending entity id per cell; obj_range: two
 * words per entity (inclusive cell range, 10 bits per axis, kind in the top bits).  Any pointer may
 * be NULL; the totals are always reported. */
int pt_debug_grid(pt_ctx* ctx, uint32_t* cell_start, int cell_capacity, uint32_t* cell_items, int item_capacity,
                  uint32_t* obj_range, int* cell_total, int* item_total, int* built_on_device);

/* Ray queries through the prepared scene; hit_id receives the 1-based entity id or 0, hit_pos / hit_normal
 * (optional) the hit.  pt_cast_rays is the tracer's own cast (code/raycast.cpp:1100-1107): every entity
 * tested under the brute-force threshold, the grid walk from it on.  pt_pick_rays is the editor's picking
 * ray (code/app.cpp:1634-1637) and focal-distance picker (code/app.cpp:3233-3247): castRayAll with
 * `outside` false, i.e. every entity tested whatever the scene's size, the grid not consulted. */
int pt_cast_rays(pt_ctx* ctx, const float* origins, const float* dirs, int n,
                 int32_t* hit_id, float* hit_pos, float* hit_normal);
int pt_pick_rays(pt_ctx* ctx, const float* origins, const float* dirs, int n,
                 int32_t* hit_id, float* hit_pos, float* hit_normal);

/* pt_cast_rays for callers that cannot allocate device memory (tight serving loops, harnesses that forbid hipMalloc,
 * processes that have handed all free memory to something else).  It has no counterpart in the reference.  Same
 * arguments, same results; the rays and their results are staged in an array that is part of the library's device
 * image (208 KiB per GPU, held until the process ends) instead of a buffer the context allocates and grows, so the
 * call contains no hipMalloc / hipFree.  At most PT_NOALLOC_MAX_RAYS rays a call (PT_ERR_LIMIT beyond).  All contexts
 * on one GPU share that GPU's array: the library serialises their calls with one lock per GPU, and refuses the call
 * with PT_ERR_BUSY while the context's own asynchronous render is active; calls one after another need nothing
 * cleared in between (a call reads back only words it wrote itself).  The call is synchronous: it copies from and to
 * the caller's (pageable) host memory and waits for the device, so it cannot be recorded into a HIP graph. */
enum { PT_NOALLOC_MAX_RAYS = 4096 };
int pt_cast_rays_noalloc(pt_ctx* ctx, const float* origins, const float* dirs, int n,
                         int32_t* hit_id, float* hit_pos, float* hit_normal);
/* For tests: free and total memory of the context's device (hipMemGetInfo) and the capacity of the context's own
 * growing scratch buffer, so that a test can assert that a call allocated nothing. */
int pt_debug_device_memory(pt_ctx* ctx, uint64_t* free_bytes, uint64_t* total_bytes, uint64_t* scratch_capacity);

/* Arithmetic probes, for tests.  The device computes the reference's divisions and square roots
 * (normVec3 code/rt_math.cpp:2166-2168, lenVec3 :2130-2164, Vec3 / float :2093) with shorter instruction
 * sequences than the compiler's whenever the operands allow it; every result must still be the IEEE
 * one.  op 0: out[i] = sqrt(a[i]); op 1: out[3i..] = a[3i..] / b[i]; op 2: out[3i..] = a[3i..] /
 * length(a[3i..]); op 3: out[3i..] = a[3i..] / (float)pi; op 4: out[3i..] = 1 / a[3i..] (the slab test's
 * reciprocals, code/rt_math.cpp lineBoxIntersection).  Host pointers.  pt_debug_arith_sweep runs
 * `count` (at most 2^32) consecutive bit patterns from first_bits through the same op on the device,
 * with derived companions, and counts results whose bits differ from the compiler's own sequences. */
int pt_debug_arith(pt_ctx* ctx, int op, const float* a, const float* b, int n, float* out);
int pt_debug_arith_sweep(pt_ctx* ctx, int op, uint32_t first_bits, uint64_t count, uint64_t* mismatches);

/* Screenshot stage code/app.cpp:1536-1552: linear float -> sRGB OETF -> 8-bit, on the device
 * framebuffer of the last trace. rgb8 holds width*height*3 bytes.  The bytes are the reference
 * expression's (floatToSrgb with a float pow, then colorFloatToInt; code/rt_math.cpp:3634-3640, 351-353):
 * the encoder is monotone, so the device compares against the 255 inputs at which its output steps up,
 * which the host finds with the C library's powf (pt_host_srgb8_thresholds). */
int pt_present_srgb8(pt_ctx* ctx, uint8_t* rgb8);
/* The same encoder on caller-supplied values (host pointers), and over `count` (at most 2^32) consecutive
 * float bit patterns from first_bits: histogram256[k] = inputs encoded to k; mismatches = inputs whose
 * code differs from a plain count of the thresholds at or below them.  For tests. */
int pt_debug_srgb8(pt_ctx* ctx, const float* linear, size_t n, uint8_t* out);
int pt_debug_srgb8_sweep(pt_ctx* ctx, uint32_t first_bits, uint64_t count, uint64_t* mismatches, uint64_t* histogram256);

/* The range count a render uses: settings->sample_ranges, or for 0 the largest power of two not
 * above spp_wanted / 16, within 1..16.  Progressive frames trace one sample per pixel: always 1.
 * Negative on a bad value.  Pure arithmetic. */
int pt_sample_ranges(const pt_settings* settings);

/*
 * Multi-GPU: a context prepared with part_count > 1 renders the 8x8 image tiles t with
 * t % part_count == part_index into a tile-major buffer of pt_partition_floats() floats
 * ([local tile][64 pixels][3]).  After the partitions' buffers have been gathered on one device
 * (the only inter-GPU step; the reference's counterpart is its workers writing one shared
 * AppData::buffer, code/raycast.cpp:1270), pt_assemble lays them out as the row-major image.
 * d_parts: part_count buffers, stride_floats apart; all pointers are device pointers.
 */
int pt_partition_floats(const pt_ctx* ctx, size_t* floats);
/* Number of 8x8 tiles partition `part_index` of `part_count` owns in a width x height image
 * (its buffer holds that many times 192 floats); negative on bad arguments. Pure arithmetic. */
int pt_partition_tiles(int width, int height, int part_index, int part_count);
int pt_assemble(pt_ctx* ctx, const float* d_parts, size_t stride_floats, int part_count, int width,
                int height, float* d_rgb, void* hip_stream);

/*
 * The same frame on several GPUs driven by one process (the reference's counterpart is its pool of worker
 * threads filling one AppData::buffer, code/app.cpp:1359-1419, code/raycast.cpp:1270).  A group holds one
 * context per entry of `devices` (an ordinal may repeat: several partitions on one GPU); partition p renders
 * the tiles t with t % n == p.  When the partitions are done their tile buffers are copied device to device
 * into the first device's gathered buffer (hipMemcpyPeerAsync: over xGMI between different GPUs, every peer on
 * its own link), laid out there as the image (pt_assemble's kernel) and copied to `rgb` once.  The image is
 * bit-identical to the one-GPU image for any n.  start / progress / abort / wait behave like their
 * pt_render_* namesakes; settings->part_index / part_count are ignored.  pt_group_ctx gives the context of a
 * partition for pt_get_stats / pt_last_error (not for rendering while the group is active).
 */
typedef struct pt_group pt_group;
pt_group* pt_group_create(const int* devices, int n);
void pt_group_destroy(pt_group* group);
int pt_group_size(const pt_group* group);
pt_ctx* pt_group_ctx(pt_group* group, int part);
const char* pt_group_last_error(const pt_group* group);
int pt_group_render_start(pt_group* group, const pt_world* world, const pt_object* objects, int n_objects,
                          const pt_settings* settings, float* rgb);
int pt_group_render_progress(pt_group* group, pt_progress* info);
int pt_group_render_snapshot(pt_group* group, float* rgb, uint64_t* pixels_done);
int pt_group_render_abort(pt_group* group);
int pt_group_render_wait(pt_group* group);

/* The same work split in three for a group, for callers that keep the scene resident on every GPU and trace frame after frame
 * (progressive display, benchmarks): the per-frame loop of the application (code/app.cpp:1436-1461) on several GPUs, driven
 * from C.  pt_group_prepare runs every partition's pt_prepare side by side.  pt_group_trace queues one frame: every partition
 * traces into the next of its three tile buffers and copies it, on its own stream, into the next of the root's three gathered
 * buffers (device to device); the root lays the image out on a side stream once the copies' events have fired.  No host thread
 * waits unless `sync` is set, so the gather and the assemble of a frame overlap the next frames' traces.  pt_group_set_frame
 * moves every partition's progressive frame counter (a scene prepared with frame_index >= 0).  pt_group_download waits for
 * everything queued and copies the latest frame's image to the caller. */
int pt_group_prepare(pt_group* group, const pt_world* world, const pt_object* objects, int n_objects, const pt_settings* settings);
int pt_group_set_frame(pt_group* group, int frame_index);
int pt_group_trace(pt_group* group, int sync);
int pt_group_sync(pt_group* group);
int pt_group_download(pt_group* group, float* rgb);

/* Scene files in the reference's own format (saveScene / loadScene, code/raycast.cpp:1578-1604):
 * a 224-byte World image followed by 172-byte Object images.  pt_scene_load with objects == NULL
 * only reports the object count. */
int pt_scene_load(const char* path, pt_world* world, pt_object* objects, int capacity, int* count);
int pt_scene_save(const char* path, const pt_world* world, const pt_object* objects, int count);

/* ---- host stages, exported so they can be checked without a GPU -------------------------- */

/* The 255 smallest floats in [0, 1] that the screenshot encoder maps to 1 .. 255, ascending. */
int pt_host_srgb8_thresholds(float* out255);

/* Flattened per-object record: the members of Object the tracer reads, after
 * preCalcObjects (code/raycast.cpp:2064-2108). 36 floats per object in the order below. */
enum {
	PT_F_POS = 0, PT_F_ROT = 3, PT_F_ROTINV = 7, PT_F_DIM = 11, PT_F_SCALE = 14, PT_F_SCALEINV = 17,
	PT_F_COLOR = 20, PT_F_EMIT = 23, PT_F_BBOX = 26, PT_F_BRADIUS = 29, PT_F_REFL = 30, PT_F_IOR = 31,
	PT_F_ROUGH = 32, PT_F_TYPE = 33 /* int bits */, PT_F_FLAGS = 34 /* bit0 isRotated, bit1 evenDim */,
	PT_F_PAD = 35, PT_F_STRIDE = 36,
};
int pt_host_flatten(const pt_object* objects, int n, float* records /* n*PT_F_STRIDE */);

/* Derived camera/world constants: 27 floats:
 * dim_h, near_dist, dir[3], up[3], right[3], top_left[3], light_dir[3], bg[3], ambient[3], light[3], pad */
enum { PT_W_DIMH = 0, PT_W_NEAR = 1, PT_W_DIR = 2, PT_W_UP = 5, PT_W_RIGHT = 8, PT_W_TOPLEFT = 11,
       PT_W_LIGHTDIR = 14, PT_W_BG = 17, PT_W_AMBIENT = 20, PT_W_LIGHT = 23, PT_W_COUNT = 27 };
int pt_host_world(const pt_world* world, int width, int height, float* out /* PT_W_COUNT */);

/* Random tables of code/app.cpp:669-727 followed by the blue-noise sample table of
 * code/app.cpp:1161-1225, all drawn from PCG stream (0, seed) in that order.
 * dirs: 20000*3, disc: 20000*2, samples: capacity*2, offsets: grid_width^2+1. */
enum { PT_TABLE_COUNT = 20000 };
int pt_host_tables(uint64_t seed, int spp_wanted, int grid_width, float* dirs, float* disc,
                   float* samples, int sample_capacity, int32_t* offsets, int* sample_count);

/* Uniform grid of code/raycast.cpp:595-617 + code/app.cpp:1247-1357 in CSR form, with the
 * intended append semantics (ascending object id per cell).  grid_ints: coordMin[3], cellCount[3],
 * cellSize[3].  Call with cell_start == NULL to size the arrays first. */
int pt_host_grid(const float* records, int n, int hint, int32_t* grid_ints /* 9 */,
                 uint32_t* cell_start, int cell_capacity, uint32_t* cell_items, int item_capacity,
                 int* cell_total, int* item_total);

/* The two device-side conventions of the grid that have no counterpart in the reference, exposed for tests.
 * pt_host_list_entry: the 16-byte list entry of the cooperative cast for an entity with bounding sphere (centre, radius),
 * tag = intersection kind | listed-in-neighbour bits << 3 (9 bits) and index id (< 2^23); *id_back = the id read back out
 * of the entry.  pt_host_cell_order: the strides (in cells) of x, y, z in the device's cell arrays for a grid of
 * cell_count[3] cells of cell_size[3] units. */
int pt_host_list_entry(const float centre[3], float radius, uint32_t tag, uint32_t id, uint32_t entry[4], uint32_t* id_back);
int pt_host_cell_order(const int32_t cell_count[3], const int32_t cell_size[3], int32_t stride[3]);

/* What the cooperative grid cast's candidate filter knows of an entity, from its record (pt_host_flatten), exposed for tests;
 * neither has a counterpart in the reference, which tests every listed entity exactly (code/raycast.cpp:957-983).
 * pt_host_filter_sphere: a sphere that encloses the entity (centre xyz, radius; about the entity's position except for a cone,
 * whose smallest enclosing sphere sits towards its base; an infinite radius for a plane).
 * pt_host_cone_slopes: for a cone record, the eight numbers of the direction test that keeps a cone the ray stays clear of when
 * the ray is parallel to one of the cone's generators (there the reference's intersector, code/raycast.cpp:352-388, divides by a
 * rounded zero and reports a hit on the base plane): out[0..2] the cone's local x axis in the world, out[3] = (A - B) / C,
 * out[4..6] its axis, out[7] = B / C with A = 4/dx^2, B = 4/dy^2, C = 1/dz^2; for a direction d,
 * a = out[3] (d.x')^2 + out[7] (|d|^2 - (d.z')^2) - (d.z')^2 is zero exactly for the generators' directions.  Eight NaNs (every
 * ray keeps the cone) for a record the formula does not fit. */
int pt_host_filter_sphere(const float* record, float centre_radius[4]);
int pt_host_cone_slopes(const float* record, float out[8]);

#ifdef __cplusplus
}
#endif
#endif /* PT_API_H */
