/*
 * TEST INFRASTRUCTURE - C-callable wrappers around the reference's own render core.
 *
 * This file contains no reference code.  oracle/ref_build.sh cuts the line ranges of
 * /root/reference/code/{rt_misc,rt_math,raycast,app}.cpp that hold the render core into a
 * temporary directory and this file #includes them from there, so that tests can call the
 * reference's functions one by one (intersectors, grid, PCG, blue noise, full tile jobs) and pin
 * the CPU restatement in oracle/pt_oracle.c against them.  Built only where /root/reference exists.
 *
 * Seed injection: the reference reseeds its thread-local PCG from __rdtsc() at the top of every
 * tile job (code/raycast.cpp:994); here __rdtsc() is a hook returning a value chosen per job.
 */
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <assert.h>
#include <float.h>
#include <stdint.h>
#include <inttypes.h>
#include <pthread.h>
#include <math.h>
#include <limits>

#include "pt_api.h"

/* rt_types.cpp:8 typedefs `ushort` as unsigned char; glibc's sys/types.h already owns the name. */
#define ushort ref_ushort

static __thread uint64_t g_ref_job_seed = 0;
static inline uint64_t ref_rdtsc_hook() { return g_ref_job_seed; }
#define __rdtsc() ref_rdtsc_hook()
#define getCycleStamp() 0LL      /* only reached from the compiled-out timing macros */

#include "rt_types.cpp"
#include "misc_macros.inc"
#include "misc_memcpy.inc"
#include "misc_pcg.inc"
#include "math_patched.inc"
#include "misc_darray.inc"
#include "rc_core.inc"
#include "rc_defobj.inc"
#include "rc_precalc.inc"
struct EntityUI;                 /* loadScene takes one and never touches it */
#include "rc_scenefile.inc"

struct RefAppData {
	RaytraceSettings settings;
	World world;
	bool preprocessStage;
	Vec3* buffer;
	ProcessPixelsData threadData[256];
	int threadCount;
};
static RefAppData adS;
static RefAppData* ad = &adS;
static int ref_grid_hint = 10;
static void* globalThreadQueue = 0;
static int g_jobs_captured = 0;
static void ref_capture_job(ProcessPixelsData*) { g_jobs_captured++; }
#define threadQueueAdd(q, fn, d) ref_capture_job(d)

/* Upper bound on the ints the reference's per-cell lists can take from the shared buffer:
 * a list of c items has reserved 5+10+20+... <= 4c+5 ints. */
static int ref_grid_buffer_bound(World* world, SpatialGrid* grid) {
	long long cells = (long long)grid->cellCount.x * grid->cellCount.y * grid->cellCount.z;
	long long total = cells * 5 + 64;
	for(int i = 0; i < world->objects.count; i++) {
		Object* obj = world->objects.data + i;
		Vec3 bMin = obj->pos - obj->geometry.boundingBox/2;
		Vec3 bMax = obj->pos + obj->geometry.boundingBox/2;
		long long span = 1;
		for(int k = 0; k < 3; k++) {
			long long lo = (long long)(bMin.e[k]/grid->cellSize.e[k]) - grid->coordMin.e[k] - 1;
			long long hi = (long long)(bMax.e[k]/grid->cellSize.e[k]) - grid->coordMin.e[k];
			if(lo < 0) lo = 0;
			if(hi > grid->cellCount.e[k]-1) hi = grid->cellCount.e[k]-1;
			span *= (hi - lo + 1) > 0 ? (hi - lo + 1) : 1;
		}
		total += span * 4;
	}
	if(total > 0x7fffff00LL) total = 0x7fffff00LL;
	return (int)total;
}

static void ref_tables_block() {
	#include "app_tables.inc"
}
static void ref_camera_block() {
	#include "app_camera.inc"
}
static void ref_setup_verbatim_block() {
	RaytraceSettings* settings = &ad->settings;
	#include "app_setup_verbatim.inc"
}
static void ref_setup_fixed_block() {
	RaytraceSettings* settings = &ad->settings;
	#include "app_setup_fixed.inc"
}
static void ref_jobs_block() {
	RaytraceSettings* settings = &ad->settings;
	#include "app_jobs.inc"
}

static Vec3 v3(const float* p) { return vec3(p[0], p[1], p[2]); }
static void put3(float* o, Vec3 v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }

extern "C" {

void ref_pcg_seed(uint64_t state, uint64_t seq) { pcg32_srandom(state, seq); }
void ref_pcg_get(uint64_t* state, uint64_t* inc) { *state = pcg32_global.state; *inc = pcg32_global.inc; }
void ref_pcg_u32(int n, uint32_t* out) { for(int i = 0; i < n; i++) out[i] = pcg32_random(); }
void ref_pcg_int(int from, int to, int n, int32_t* out) { for(int i = 0; i < n; i++) out[i] = randomIntPCG(from, to); }
void ref_pcg_float(float from, float to, float prec, int n, float* out) { for(int i = 0; i < n; i++) out[i] = randomFloatPCG(from, to, prec); }

/* code/app.cpp:669-727 on the current global PCG state. */
void ref_tables(float* dirs, float* disc) {
	if(ad->settings.randomDirections) free(ad->settings.randomDirections);
	if(ad->settings.randomUnitCirclePoints) free(ad->settings.randomUnitCirclePoints);
	ref_tables_block();
	if(dirs) memcpy(dirs, ad->settings.randomDirections, sizeof(Vec3)*ad->settings.randomDirectionCount);
	if(disc) memcpy(disc, ad->settings.randomUnitCirclePoints, sizeof(Vec2)*ad->settings.randomUnitCirclePointCount);
}

/* Loads a scene, then runs the camera update (code/app.cpp:1075-1090) and the render-start
 * setup (code/app.cpp:1161-1357: blue noise on the current global PCG state, precalc, grid).
 * fixed != 0 selects the variant with the cell-list relocation repaired. Returns the sample count. */
int ref_scene_setup(const pt_world* w, const pt_object* objs, int n, const pt_settings* s, int fixed) {
	World* world = &ad->world;
	DArray<Object> keep = world->objects;
	*world = {};
	world->objects = keep;
	world->objects.clear();
	if(!world->objects.data) world->objects.init();

	Camera* cam = &world->camera;
	cam->pos = v3(w->camera.pos);
	cam->rot = v3(w->camera.rot);
	cam->fov = w->camera.fov;
	cam->dim.w = w->camera.plane_width;
	cam->farDist = 10000;
	world->backgroundColor = v3(w->background_color);
	world->ambientLightColor = v3(w->ambient_color);
	world->globalLightColor = v3(w->light_color);
	world->globalLightRot = vec2i(w->light_rot[0], w->light_rot[1]);
	world->focalPointDistance = w->focal_distance;
	world->apertureSize = w->aperture;

	for(int i = 0; i < n; i++) {
		Object obj = {};
		obj.id = i;
		obj.pos = v3(objs[i].pos);
		obj.rot = quat(objs[i].rot[0], objs[i].rot[1], objs[i].rot[2], objs[i].rot[3]);
		obj.dim = v3(objs[i].dim);
		obj.color = v3(objs[i].color);
		obj.geometry.type = objs[i].type;
		obj.material.emitColor = v3(objs[i].emit_color);
		obj.material.reflectionMod = objs[i].reflection_mod;
		obj.material.refractiveIndex = objs[i].refractive_index;
		obj.material.roughness = objs[i].roughness;
		world->objects.push(obj);
	}

	ad->settings.texDim = vec2i(s->width, s->height);
	ad->settings.sampleCountWanted = s->spp_wanted;
	ad->settings.sampleGridWidth = s->sample_grid_width;
	ad->settings.rayBouncesMaxWanted = s->max_bounces;
	ad->preprocessStage = false;
	ad->settings.darken = false;
	ref_grid_hint = s->grid_hint;

	ref_camera_block();

	if(ad->settings.grid) { free(ad->settings.grid); ad->settings.grid = 0; }
	if(fixed) ref_setup_fixed_block(); else ref_setup_verbatim_block();

	int pixelCount = s->width*s->height;
	ad->buffer = (Vec3*)realloc(ad->buffer, sizeof(Vec3)*pixelCount);
	for(int i = 0; i < pixelCount; i++) ad->buffer[i] = vec3(-1,-1,-1);
	return ad->settings.sampleCount;
}

void ref_get_samples(float* samples, int32_t* offsets) {
	RaytraceSettings* st = &ad->settings;
	if(samples) memcpy(samples, st->samples, sizeof(Vec2)*st->sampleCount);
	if(offsets) memcpy(offsets, st->sampleGridOffsets, sizeof(int)*st->sampleGridCount);
}

/* Per-object members after preCalcObjects, in the PT_F_* record layout of pt_api.h. */
void ref_get_records(float* rec) {
	World* world = &ad->world;
	for(int i = 0; i < world->objects.count; i++) {
		Object* o = world->objects.data + i;
		float* r = rec + i*PT_F_STRIDE;
		memset(r, 0, sizeof(float)*PT_F_STRIDE);
		put3(r+PT_F_POS, o->pos);
		r[PT_F_ROT] = o->rot.w; r[PT_F_ROT+1] = o->rot.x; r[PT_F_ROT+2] = o->rot.y; r[PT_F_ROT+3] = o->rot.z;
		r[PT_F_ROTINV] = o->rotInverse.w; r[PT_F_ROTINV+1] = o->rotInverse.x; r[PT_F_ROTINV+2] = o->rotInverse.y; r[PT_F_ROTINV+3] = o->rotInverse.z;
		put3(r+PT_F_DIM, o->dim);
		put3(r+PT_F_SCALE, o->dimScale);
		put3(r+PT_F_SCALEINV, o->dimScaleInverse);
		put3(r+PT_F_COLOR, o->colorLinear);
		put3(r+PT_F_EMIT, o->material.emitColorLinear);
		put3(r+PT_F_BBOX, o->geometry.boundingBox);
		r[PT_F_BRADIUS] = o->geometry.boundingSphereRadius;
		r[PT_F_REFL] = o->material.reflectionMod;
		r[PT_F_IOR] = o->material.refractiveIndex;
		r[PT_F_ROUGH] = o->material.roughness;
		int32_t type = o->geometry.type, flags = (o->isRotated ? 1 : 0) | (o->evenDim ? 2 : 0);
		memcpy(r+PT_F_TYPE, &type, 4);
		memcpy(r+PT_F_FLAGS, &flags, 4);
	}
}

void ref_get_world(float* out) {
	World* world = &ad->world;
	Camera* cam = &world->camera;
	memset(out, 0, sizeof(float)*PT_W_COUNT);
	out[PT_W_DIMH] = cam->dim.h;
	out[PT_W_NEAR] = cam->nearDist;
	put3(out+PT_W_DIR, cam->ovecs.dir);
	put3(out+PT_W_UP, cam->ovecs.up);
	put3(out+PT_W_RIGHT, cam->ovecs.right);
	put3(out+PT_W_TOPLEFT, ad->settings.camTopLeft);
	put3(out+PT_W_LIGHTDIR, world->globalLightDir);
	put3(out+PT_W_BG, world->backgroundColorLinear);
	put3(out+PT_W_AMBIENT, world->ambientLightColorLinear);
	put3(out+PT_W_LIGHT, world->globalLightColorLinear);
}

/* coordMin[3], cellCount[3], cellSize[3] of the grid built by ref_scene_setup. */
void ref_get_grid(int32_t* ints9) {
	SpatialGrid* g = &ad->settings.spatialGrid;
	for(int i = 0; i < 3; i++) { ints9[i] = g->coordMin.e[i]; ints9[3+i] = g->cellCount.e[i]; ints9[6+i] = g->cellSize.e[i]; }
}
int ref_get_cell(int index, int32_t* items, int cap) {
	ObjectList* l = &ad->settings.grid[index];
	if(!l->data) return 0;
	for(int i = 0; i < l->count && i < cap; i++) items[i] = l->data[i];
	return l->count;
}

/* Intersectors on raw arguments (code/raycast.cpp:390-585, code/rt_math.cpp:2343-2373). */
void ref_line_sphere(int n, const float* lp, const float* ld, const float* sp, const float* sr, float* dist, float* pos, float* nrm) {
	for(int i = 0; i < n; i++) {
		Vec3 p = vec3(0,0,0), q = vec3(0,0,0);
		dist[i] = lineSphereIntersectionX(v3(lp+3*i), v3(ld+3*i), v3(sp+3*i), sr[i], &p, &q);
		put3(pos+3*i, p); put3(nrm+3*i, q);
	}
}
void ref_line_box(int n, const float* lp, const float* ld, const float* bp, const float* bd, float* dist, float* pos, float* nrm) {
	for(int i = 0; i < n; i++) {
		Vec3 p = vec3(0,0,0), q = vec3(0,0,0);
		dist[i] = lineBoxIntersection(v3(lp+3*i), v3(ld+3*i), v3(bp+3*i), v3(bd+3*i), &p, &q);
		put3(pos+3*i, p); put3(nrm+3*i, q);
	}
}
void ref_line_cylinder(int n, const float* lp, const float* ld, float* dist, float* pos, float* nrm) {
	for(int i = 0; i < n; i++) {
		Vec3 p = vec3(0,0,0), q = vec3(0,0,0);
		dist[i] = lineCylinderIntersection(v3(lp+3*i), v3(ld+3*i), &p, &q);
		put3(pos+3*i, p); put3(nrm+3*i, q);
	}
}
void ref_line_cone(int n, const float* lp, const float* ld, float* dist, float* pos, float* nrm) {
	for(int i = 0; i < n; i++) {
		Vec3 p = vec3(0,0,0), q = vec3(0,0,0);
		dist[i] = lineConeIntersection(v3(lp+3*i), v3(ld+3*i), &p, &q);
		put3(pos+3*i, p); put3(nrm+3*i, q);
	}
}
void ref_sphere_collision(int n, const float* lp, const float* ld, const float* sp, const float* sr, int32_t* hit) {
	for(int i = 0; i < n; i++) hit[i] = lineSphereCollision(v3(lp+3*i), v3(ld+3*i), v3(sp+3*i), sr[i]) ? 1 : 0;
}
void ref_optics(int n, const float* inc, const float* nrm, const float* ior, float* kr, float* refr, float* refl) {
	for(int i = 0; i < n; i++) {
		kr[i] = fresnel(v3(inc+3*i), v3(nrm+3*i), ior[i]);
		put3(refr+3*i, refract(v3(inc+3*i), v3(nrm+3*i), ior[i]));
		put3(refl+3*i, reflectVector(v3(inc+3*i), v3(nrm+3*i)));
	}
}

/* getIntersection (code/raycast.cpp:813-893) of object `obj` of the loaded scene. */
void ref_get_intersection(int n, const int32_t* obj, const float* rp, const float* rd, float* dist, float* pos, float* nrm) {
	for(int i = 0; i < n; i++) {
		Vec3 p = vec3(0,0,0), q = vec3(0,0,0);
		dist[i] = getIntersection(ad->world.objects.data + obj[i], v3(rp+3*i), v3(rd+3*i), &p, &q);
		put3(pos+3*i, p); put3(nrm+3*i, q);
	}
}

/* Cell sequence of one ray: spatialGridTraverseInit / Next (code/raycast.cpp:619-741). */
int ref_traverse(const float* rp, const float* rd, int cap, int32_t* cells, int32_t* outside) {
	SpatialGrid g = ad->settings.spatialGrid;
	g.outside = false;
	spatialGridTraverseInit(&g, v3(rp), v3(rd));
	*outside = g.outside ? 1 : 0;
	if(g.outside) return 0;
	int count = 0;
	do {
		if(count < cap) { cells[3*count] = g.coord.x; cells[3*count+1] = g.coord.y; cells[3*count+2] = g.coord.z; }
		count++;
	} while(spatialGridTraverseNext(&g));
	return count;
}

/* castRay (grid, code/raycast.cpp:928-986) or castRayAll (code/raycast.cpp:895-926, with the
 * traverse-init `outside` flag as at code/raycast.cpp:1100-1103). */
void ref_cast(int n, const float* rp, const float* rd, int use_grid, int32_t* id, float* pos, float* nrm) {
	RaytraceSettings settings = ad->settings;
	for(int i = 0; i < n; i++) {
		Vec3 p = vec3(0,0,0), q = vec3(0,0,0);
		settings.spatialGrid.outside = false;
		if(use_grid == 2) {
			id[i] = castRayAll(v3(rp+3*i), v3(rd+3*i), false, &ad->world.objects, &p, &q);   /* as the editor's pick calls it */
		} else if(use_grid) {
			id[i] = castRay(v3(rp+3*i), v3(rd+3*i), &settings, &ad->world.objects, &p, &q);
		} else {
			spatialGridTraverseInit(&settings.spatialGrid, v3(rp+3*i), v3(rd+3*i));
			id[i] = castRayAll(v3(rp+3*i), v3(rd+3*i), settings.spatialGrid.outside, &ad->world.objects, &p, &q);
		}
		put3(pos+3*i, p); put3(nrm+3*i, q);
	}
}

/* Tile jobs in the reference's spiral order (code/app.cpp:1359-1419): x, y, w, h per job. */
int ref_jobs(int32_t* xywh) {
	g_jobs_captured = 0;
	ref_jobs_block();
	for(int i = 0; i < ad->threadCount && xywh; i++) {
		xywh[4*i] = ad->threadData[i].pixelPos.x; xywh[4*i+1] = ad->threadData[i].pixelPos.y;
		xywh[4*i+2] = ad->threadData[i].pixelDim.x; xywh[4*i+3] = ad->threadData[i].pixelDim.y;
	}
	return ad->threadCount;
}

/* Runs every tile job serially through processPixelsThreaded (code/raycast.cpp:991-1289);
 * job i is seeded with pcg32_srandom(0, seed_base + i). Returns the job count. */
int ref_render(uint64_t seed_base, float* rgb) {
	int jobs = ref_jobs(0);
	for(int i = 0; i < jobs; i++) {
		g_ref_job_seed = seed_base + (uint64_t)i;
		ad->threadData[i].stopProcessing = false;
		processPixelsThreaded(ad->threadData + i);
	}
	int pixelCount = ad->settings.texDim.w*ad->settings.texDim.h;
	if(rgb) memcpy(rgb, ad->buffer, sizeof(Vec3)*pixelCount);
	return jobs;
}

/* The same jobs handed out in order to `threads` workers - the reference's own scheduling
 * (code/rt_misc_win32.cpp:45-74) with pthreads in place of the Win32 queue. */
struct RefWorker { int jobs; volatile int* next; uint64_t seed_base; };
static void* ref_worker_main(void* arg) {
	RefWorker* w = (RefWorker*)arg;
	for(;;) {
		int i = __sync_fetch_and_add(w->next, 1);
		if(i >= w->jobs) break;
		g_ref_job_seed = w->seed_base + (uint64_t)i;
		processPixelsThreaded(ad->threadData + i);
	}
	return 0;
}
int ref_render_mt(uint64_t seed_base, int threads, float* rgb) {
	int jobs = ref_jobs(0);
	if(threads < 1) threads = 1;
	if(threads > 256) threads = 256;
	for(int i = 0; i < jobs; i++) ad->threadData[i].stopProcessing = false;
	volatile int next = 0;
	RefWorker w = { jobs, &next, seed_base };
	pthread_t tid[256];
	for(int t = 0; t < threads; t++) pthread_create(&tid[t], 0, ref_worker_main, &w);
	for(int t = 0; t < threads; t++) pthread_join(tid[t], 0);
	int pixelCount = ad->settings.texDim.w*ad->settings.texDim.h;
	if(rgb) memcpy(rgb, ad->buffer, sizeof(Vec3)*pixelCount);
	return jobs;
}

/* One job covering an arbitrary rectangle, seeded with pcg32_srandom(0, seed). */
void ref_render_rect(uint64_t seed, int x, int y, int w, int h, float* rgb) {
	ProcessPixelsData d = {};
	d.world = &ad->world; d.settings = &ad->settings; d.buffer = ad->buffer;
	d.pixelPos = vec2i(x, y); d.pixelDim = vec2i(w, h);
	g_ref_job_seed = seed;
	processPixelsThreaded(&d);
	int pixelCount = ad->settings.texDim.w*ad->settings.texDim.h;
	if(rgb) memcpy(rgb, ad->buffer, sizeof(Vec3)*pixelCount);
}

/* Entity hit by the primary ray of camera sample 0 of every pixel (pinhole camera only):
 * ray construction as code/raycast.cpp:1040-1066, cast as code/raycast.cpp:1100-1107. */
void ref_primary_hits(int32_t* ids) {
	World world = ad->world;
	RaytraceSettings settings = ad->settings;
	Camera camera = world.camera;
	Vec2i texDim = settings.texDim;
	Vec2 texDimFloat = vec2(settings.texDim);
	Vec3 camOvecsRight = camera.ovecs.right;
	Vec3 camOvecsDown = -camera.ovecs.up;
	Vec3 camRightWidth = camera.ovecs.right * camera.dim.w;
	Vec3 camDownHeight = camOvecsDown * camera.dim.h;
	Vec3 camRightOnePixel = camOvecsRight * camera.dim.w * (1.0f/texDim.w);
	Vec3 camDownOnePixel = camOvecsDown * camera.dim.h * (1.0f/texDim.h);
	for(int y = 0; y < texDim.h; y++) for(int x = 0; x < texDim.w; x++) {
		Vec2 percent = vec2(x/texDimFloat.w, y/texDimFloat.h);
		Vec3 camPixelPos = settings.camTopLeft + camRightWidth*percent.w + camDownHeight*percent.h;
		int index = (y%settings.sampleGridWidth)*settings.sampleGridWidth + (x%settings.sampleGridWidth);
		int offset = settings.sampleGridOffsets[index];
		int sampleCount = settings.sampleGridOffsets[index+1] - offset;
		if(sampleCount == 0) { ids[y*texDim.w + x] = -1; continue; }
		Vec2 sample = settings.samples[offset];
		Vec3 rayPos = camPixelPos + camRightOnePixel * sample.x + camDownOnePixel * sample.y;
		Vec3 rayDir = normVec3(rayPos - camera.pos);
		Vec3 p, q;
		int id;
		if(world.objects.count < 10) {
			spatialGridTraverseInit(&settings.spatialGrid, rayPos, rayDir);
			id = castRayAll(rayPos, rayDir, settings.spatialGrid.outside, &world.objects, &p, &q);
		} else {
			id = castRay(rayPos, rayDir, &settings, &world.objects, &p, &q);
		}
		ids[y*texDim.w + x] = id;
	}
}

/* Screenshot encoding of one value (code/app.cpp:1541-1547, code/rt_math.cpp:3634-3640). */
void ref_srgb8(int n, const float* lin, uint8_t* out) {
	for(int i = 0; i < n; i++) out[i] = (uint8_t)(char)colorFloatToInt(floatToSrgb(lin[i]));
}

/* saveScene / loadScene (code/raycast.cpp:1578-1604) on the loaded scene. */
void ref_save_scene(const char* path) { saveScene(&ad->world, (char*)path); }
int ref_load_scene(const char* path, pt_world* w, pt_object* objs, int cap) {
	loadScene(&ad->world, (char*)path, 0);
	World* world = &ad->world;
	memset(w, 0, sizeof *w);
	put3(w->camera.pos, world->camera.pos); put3(w->camera.rot, world->camera.rot);
	w->camera.fov = world->camera.fov; w->camera.plane_width = world->camera.dim.w;
	put3(w->background_color, world->backgroundColor); put3(w->ambient_color, world->ambientLightColor);
	put3(w->light_color, world->globalLightColor);
	w->light_rot[0] = world->globalLightRot.x; w->light_rot[1] = world->globalLightRot.y;
	w->focal_distance = world->focalPointDistance; w->aperture = world->apertureSize;
	for(int i = 0; i < world->objects.count && i < cap; i++) {
		Object* o = world->objects.data + i;
		memset(objs + i, 0, sizeof(pt_object));
		put3(objs[i].pos, o->pos); put3(objs[i].dim, o->dim); put3(objs[i].color, o->color);
		objs[i].rot[0] = o->rot.w; objs[i].rot[1] = o->rot.x; objs[i].rot[2] = o->rot.y; objs[i].rot[3] = o->rot.z;
		objs[i].type = o->geometry.type;
		put3(objs[i].emit_color, o->material.emitColor);
		objs[i].reflection_mod = o->material.reflectionMod;
		objs[i].refractive_index = o->material.refractiveIndex;
		objs[i].roughness = o->material.roughness;
	}
	return world->objects.count;
}

int ref_sizeof(int which) {
	switch(which) {
		case 0: return sizeof(Object);
		case 1: return sizeof(World);
		case 2: return sizeof(Camera);
		case 3: return sizeof(SpatialGrid);
		case 4: return sizeof(RaytraceSettings);
	}
	return 0;
}

} /* extern "C" */
