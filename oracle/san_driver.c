/* TEST INFRASTRUCTURE - drives the CPU restatement (pt_oracle.c) through its whole surface so that it can run under the
 * compiler's sanitizers: built twice by `make san`, once with -fsanitize=address,undefined and once with -fsanitize=thread.
 * Two scenes (the demo scene; forty entities of every kind and material on a ground box, some of them glass and rough),
 * both seed policies, the threaded renderer with four workers, primary-hit pass, ray casts of all three kinds, progressive
 * frames.  Prints a checksum; the sanitizers report on stderr and through the exit code. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pt_api.h"

typedef struct pto_scene pto_scene;
pto_scene* pto_scene_create(const pt_world* w, const pt_object* objs, int n, const pt_settings* st);
void pto_scene_destroy(pto_scene* s);
int pto_render(pto_scene* s, uint64_t seed_base, int policy, float* rgb);
int pto_render_mt(pto_scene* s, uint64_t seed_base, int policy, int threads, float* rgb);
void pto_primary(pto_scene* s, int32_t* ids, uint64_t* hashes);
void pto_cast(pto_scene* s, int n, const float* rp, const float* rd, int use_grid, int32_t* id, float* pos, float* nrm);
void pto_progressive_frame(pto_scene* s, int frame, float* acc);
uint64_t pto_path_count(const pto_scene* s);
void pto_srgb8(int n, const float* lin, uint8_t* out);

static uint32_t g_state = 12345u;
static float unit(void) { g_state = g_state * 1664525u + 1013904223u; return (float)(g_state >> 8) / 16777216.0f; }
static float between(float lo, float hi) { return lo + (hi - lo) * unit(); }

static void demo(pt_world* w, pt_object* o) {
	memset(w, 0, sizeof *w);
	memset(o, 0, 2 * sizeof *o);
	w->camera.pos[1] = -30; w->camera.pos[2] = 7; w->camera.fov = 90; w->camera.plane_width = 10;
	for(int k = 0; k < 3; k++) { w->background_color[k] = 1; w->ambient_color[k] = 1; w->light_color[k] = 1; }
	w->light_rot[1] = 90; w->focal_distance = 30;
	o[0].pos[2] = 0.1f; o[0].rot[0] = 1; o[0].dim[0] = 50; o[0].dim[1] = 50; o[0].dim[2] = 0.01f;
	o[0].color[0] = o[0].color[1] = o[0].color[2] = 0.5f; o[0].type = PT_GEOM_BOX; o[0].refractive_index = 1;
	o[1].pos[2] = 7; o[1].rot[0] = 1; o[1].dim[0] = o[1].dim[1] = o[1].dim[2] = 8;
	o[1].color[0] = o[1].color[1] = o[1].color[2] = 0.7f; o[1].type = PT_GEOM_SPHERE; o[1].reflection_mod = 1; o[1].refractive_index = 1;
}

static double run(const pt_world* w, const pt_object* o, int n, int width, int height, int spp, int bounces) {
	pt_settings st;
	memset(&st, 0, sizeof st);
	st.width = width; st.height = height; st.spp_wanted = spp; st.sample_grid_width = 3; st.max_bounces = bounces;
	st.grid_hint = 10; st.brute_force_below = 10; st.seed_policy = PT_SEED_PER_SAMPLE; st.seed = 4242;
	st.part_count = 1; st.frame_index = -1;
	pto_scene* s = pto_scene_create(w, o, n, &st);
	if(!s) { fprintf(stderr, "scene creation failed\n"); exit(3); }
	const size_t px = (size_t)width * height;
	float* a = (float*)calloc(px * 3, 4), *b = (float*)calloc(px * 3, 4);
	int32_t* ids = (int32_t*)calloc(px, 4);
	uint64_t* hashes = (uint64_t*)calloc(px, 8);
	double sum = 0;
	pto_render_mt(s, 7, PT_SEED_PER_JOB, 4, a);
	pto_render(s, 7, PT_SEED_PER_JOB, b);
	if(memcmp(a, b, px * 12) != 0) { fprintf(stderr, "threaded and serial renders differ\n"); exit(4); }
	pto_render_mt(s, 0, PT_SEED_PER_SAMPLE, 4, b);
	for(size_t i = 0; i < px * 3; i++) sum += a[i] + b[i];
	pto_primary(s, ids, hashes);
	for(size_t i = 0; i < px; i++) sum += ids[i] + (double)(hashes[i] & 1023u);
	enum { RAYS = 2000 };
	static float ro[RAYS * 3], rd[RAYS * 3], rp[RAYS * 3], rn[RAYS * 3];
	static int32_t rid[RAYS];
	for(int i = 0; i < RAYS; i++) {
		for(int k = 0; k < 3; k++) { ro[3 * i + k] = between(-30, 30); rd[3 * i + k] = between(-1, 1); }
		if(i % 7 == 0) rd[3 * i + i % 3] = 0;
	}
	for(int mode = 0; mode < 3; mode++) {
		pto_cast(s, RAYS, ro, rd, mode, rid, rp, rn);
		for(int i = 0; i < RAYS; i++) sum += rid[i];
	}
	memset(a, 0, px * 12);
	for(int f = 0; f < 3; f++) pto_progressive_frame(s, f, a);
	uint8_t* bytes = (uint8_t*)malloc(px * 3);
	pto_srgb8((int)(px * 3), b, bytes);
	for(size_t i = 0; i < px * 3; i++) sum += bytes[i];
	sum += (double)pto_path_count(s);
	free(bytes); free(hashes); free(ids); free(b); free(a);
	pto_scene_destroy(s);
	return sum;
}

int main(void) {
	pt_world w;
	static pt_object o[64];
	demo(&w, o);
	double sum = run(&w, o, 2, 96, 54, 16, 8);
	/* forty entities over the ground box */
	int n = 1;
	for(int i = 0; i < 40; i++, n++) {
		pt_object* e = o + n;
		memset(e, 0, sizeof *e);
		e->type = i % 4;
		e->pos[0] = between(-18, 18); e->pos[1] = between(-18, 18); e->pos[2] = between(1, 12);
		const float d0 = between(1.5f, 5);
		const int even = i % 5 == 0;
		e->dim[0] = d0; e->dim[1] = even ? d0 : between(1.5f, 5); e->dim[2] = even ? d0 : between(1.5f, 5);
		if(i % 3 == 0) e->rot[0] = 1;
		else { float q[4] = { between(-1, 1), between(-1, 1), between(-1, 1), between(-1, 1) }; float l = 0; for(int k = 0; k < 4; k++) l += q[k] * q[k]; l = l > 0 ? 1.0f / __builtin_sqrtf(l) : 1; for(int k = 0; k < 4; k++) e->rot[k] = q[k] * l; }
		for(int k = 0; k < 3; k++) e->color[k] = between(0.4f, 0.95f);
		e->refractive_index = i % 4 == 1 ? between(1.2f, 2.4f) : 1.0f;
		e->reflection_mod = i % 6 == 2 ? 1.0f : i % 6 == 3 ? 0.5f : 0.0f;
		e->roughness = i % 5 == 4 ? 0.5f : 0.0f;
		if(i == 17) e->emit_color[0] = 1;
	}
	w.aperture = 0.8f;
	sum += run(&w, o, n, 80, 45, 16, 16);
	printf("checksum %.3f\n", sum);
	return 0;
}
