/*
 * pt_oracle.c - TEST INFRASTRUCTURE: a CPU restatement of the reference's render core.
 *
 * This is the checker, never the product: only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  It restates, in plain C and scalar IEEE fp32 with the
 * reference's operation order (build with -O2 -ffp-contract=off, never -ffast-math), the path
 *
 *   sample tables -> per-object precalc -> uniform grid -> tile jobs -> per-pixel path loop
 *
 * of /root/reference/code/{raycast.cpp, app.cpp, rt_math.cpp, rt_misc.cpp}.  Every function cites
 * the reference lines it follows.  Parity pin: tests/test_oracle_vs_ref.py compares each stage
 * bit-for-bit with the reference's own code compiled as oracle/_ref/libptref.so, and
 * tests/golden/ holds vectors generated from that library for machines without it.
 *
 * Deliberate, documented departures from the reference (none reachable in the pinned cases):
 *   - pixels whose blue-noise tile holds no sample are written as 0, not 0/0 = NaN
 *     (code/raycast.cpp:1270);
 *   - the per-cell lists are built with the intended append semantics; the reference relocates a
 *     grown list to the base of the shared buffer (code/app.cpp:1331), which corrupts cells with
 *     five or more objects;
 *   - values the reference reads uninitialised (focal point when the focal plane is missed,
 *     code/raycast.cpp:1068-1072; box normal when no face matches, :453-458) are zero here.
 */
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <math.h>
#include <float.h>
#include <pthread.h>

#include "pt_api.h"

typedef struct { float x, y, z; } v3;
typedef struct { float x, y; } v2;
typedef struct { float w, x, y, z; } quat;
typedef struct { int x, y, z; } v3i;

/* ---- scalar helpers: code/rt_math.cpp:60-74,106-125,222-226,323-325 ------------------------ */
static inline float fmin2(float a, float b) { return a <= b ? a : b; }
static inline float fmax2(float a, float b) { return a >= b ? a : b; }
static inline float fmin3(float a, float b, float c) { return fmin2(fmin2(a, b), fmin2(b, c)); }
static inline float fmax3(float a, float b, float c) { return fmax2(fmax2(a, b), fmax2(b, c)); }
static inline float clamp_min(float a, float lo) { return a < lo ? lo : a; }
static inline float clamp_max(float a, float hi) { return a > hi ? hi : a; }
static inline float lerp_check(float t, float a, float b) {
	if(t == 0) return a;
	else if(t == 1) return b;
	else return a + t * (b - a);
}
static inline int round_int(float f) { return (int)floorf(f + 0.5f); }
/* clampInt routes through the float clampMax (code/rt_math.cpp:179-181). */
static inline int clamp_int(int n, int lo, int hi) {
	int a = n < lo ? lo : n;
	float r = clamp_max((float)a, (float)hi);
	return (int)r;
}

/* ---- Vec3: code/rt_math.cpp:1979-2168 ------------------------------------------------------ */
static inline v3 V3(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 add(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 scale(v3 a, float b) { return V3(a.x * b, a.y * b, a.z * b); }
static inline v3 divs(v3 a, float b) { return V3(a.x / b, a.y / b, a.z / b); }
static inline v3 divv(v3 a, v3 b) { return V3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline v3 sdiv(float b, v3 a) { return V3(b / a.x, b / a.y, b / a.z); }
static inline v3 neg(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline int eq3(v3 a, v3 b) { return (a.x == b.x) && (a.y == b.y) && (a.z == b.z); }
static inline float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 cross(v3 a, v3 b) {
	return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float len3(v3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
static inline v3 norm3(v3 a) { return divs(a, len3(a)); }
static inline v3 clamp_max3(v3 a, float hi) { return V3(clamp_max(a.x, hi), clamp_max(a.y, hi), clamp_max(a.z, hi)); }
static inline v3 lerp_check3(float t, v3 a, v3 b) {
	return V3(lerp_check(t, a.x, b.x), lerp_check(t, a.y, b.y), lerp_check(t, a.z, b.z));
}
/* code/rt_math.cpp:2405-2408 */
static inline v3 reflect_vector(v3 dir, v3 normal) {
	float k = 2 * dot(dir, normal);
	return sub(dir, scale(normal, k));
}
/* code/rt_math.cpp:3340-3349 */
static inline int point_in_box(v3 p, v3 lo, v3 hi) {
	return p.x >= lo.x && p.x <= hi.x && p.y >= lo.y && p.y <= hi.y && p.z >= lo.z && p.z <= hi.z;
}

/* ---- Quat: code/rt_math.cpp:2789-2916 ------------------------------------------------------ */
static inline quat quat_axis(float a, v3 axis) {
	quat r;
	r.w = cosf(a * 0.5f);
	r.x = axis.x * sinf(a * 0.5f);
	r.y = axis.y * sinf(a * 0.5f);
	r.z = axis.z * sinf(a * 0.5f);
	return r;
}
static inline quat quat_mul(quat a, quat b) {
	quat r;
	r.w = (a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z);
	r.x = (a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y);
	r.y = (a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x);
	r.z = (a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w);
	return r;
}
static inline quat quat_inverse(quat q) {
	float s = (1 / (q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z));
	quat r = { q.w * s, -q.x * s, -q.y * s, -q.z * s };
	return r;
}
/* code/rt_math.cpp:2902-2907 */
static inline v3 quat_rot(quat q, v3 v) {
	v3 qv = V3(q.x, q.y, q.z);
	v3 c = cross(qv, v);
	v3 t = V3(c.x * 2, c.y * 2, c.z * 2);
	v3 qt = V3(t.x * q.w, t.y * q.w, t.z * q.w);
	return add(add(v, qt), cross(qv, t));
}

/* ---- PCG32: code/rt_misc.cpp:1152-1251 ------------------------------------------------------ */
typedef struct { uint64_t state, inc; } pcg32;
static inline uint32_t pcg_next(pcg32* r) {
	uint64_t old = r->state;
	r->state = old * 6364136223846793005ULL + r->inc;
	uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
	uint32_t rot = (uint32_t)(old >> 59u);
	return (xorshifted >> rot) | (xorshifted << ((-rot) & 31));
}
static inline void pcg_seed(pcg32* r, uint64_t initstate, uint64_t initseq) {
	r->state = 0U;
	r->inc = (initseq << 1u) | 1u;
	pcg_next(r);
	r->state += initstate;
	pcg_next(r);
}
static inline uint32_t pcg_bounded(pcg32* r, uint32_t bound) {
	uint32_t threshold = -bound % bound;
	for(;;) {
		uint32_t v = pcg_next(r);
		if(v >= threshold) return v % bound;
	}
}
static inline int random_int(pcg32* r, int from, int to) { return (int)(pcg_bounded(r, (uint32_t)(to - from + 1)) + (uint32_t)from); }
static inline float random_float(pcg32* r, float from, float to, float prec) {
	return random_int(r, (int)(from / prec), (int)(to / prec)) * prec;
}

/* ---- scene ----------------------------------------------------------------------------------- */
typedef struct {
	v3 pos; quat rot; v3 dim; v3 color; int type;
	v3 emit; float refl, ior, rough;
	/* precalc: code/raycast.cpp:2064-2108 */
	v3 color_lin, emit_lin; int is_rotated, even_dim; quat rot_inv; v3 dim_scale, dim_scale_inv;
	float bradius; v3 bbox;
} obj_t;

typedef struct { v3i coord_min, coord_max, bmin, bmax, cell_count, cell_size; } grid_t;

/* Per-ray traversal state: the "Grid Traversal" half of SpatialGrid (code/raycast.cpp:304-317). */
typedef struct { v3i coord, step; v3 tmax, tdelta; int outside; v3 box_min, box_max; } walk_t;

#define PTO_HIST_BINS 65
typedef struct { uint64_t casts, dda_steps, bsphere_tests, isect[5], rotated_box, uneven_sphere,
                 shade_opaque, shade_oren, shade_glass, paths, lens, emit;
                 uint64_t path_casts[PTO_HIST_BINS]; /* [k]: camera samples whose path cast k rays (the last bin takes the rest) */ } counters_t;

typedef struct pto_scene {
	pt_world win; pt_settings st;
	int n; obj_t* obj;
	/* camera: code/raycast.cpp:148-156, code/app.cpp:1077-1090,1234 */
	v3 cam_pos, cam_dir, cam_up, cam_right, cam_top_left; float dim_w, dim_h, near_dist;
	v3 light_dir, bg_lin, ambient_lin, light_lin;
	float focal_distance, aperture;
	/* tables: code/app.cpp:669-727 */
	v3* dirs; v2* disc;
	/* samples: code/app.cpp:1161-1225 */
	int sample_count; v2* samples; int* offsets; int grid_w;
	grid_t grid; int cells; uint32_t* cell_start; uint32_t* cell_items;
	counters_t cnt;
	/* The reference keeps the traverse-init's `outside` flag in the tile job's copy of the grid and only rewrites it for rays
	 * that start outside the grid's box (code/raycast.cpp:619-638): a ray that starts inside inherits the flag of the last ray
	 * that started outside.  Under PT_SEED_PER_JOB (the reference's structure) the oracle does the same, from `false` at the
	 * start of every job; under PT_SEED_PER_SAMPLE, whose paths are independent by definition, every ray starts from `false`. */
	int sticky_outside, outside_state;
	int n_planes; int* plane_ids;     /* the entities of type plane, ascending: what a ray that misses the grid tests */
} pto_scene;

/* ---- intersectors: code/raycast.cpp:390-585, code/rt_math.cpp:2343-2373 -------------------- */
static float line_sphere(v3 lp, v3 ld, v3 sp, float sr, v3* pos, v3* nrm) {
	v3 old = lp;
	lp = sub(lp, sp);
	float d = dot(ld, lp);
	float l = len3(lp);
	float b = d * d - l * l + sr * sr;
	if(b < 0) return -1;
	float distance = -(dot(ld, lp));
	float sq = sqrtf(b);
	if(distance - sq < 0) {
		if(distance + sq < 0) return -1;
		else distance += sq;
	} else {
		distance -= sq;
	}
	*pos = add(old, scale(ld, distance));
	*nrm = norm3(sub(*pos, sp));
	return distance;
}

static int sphere_collision(v3 lp, v3 ld, v3 sp, float sr) {
	lp = sub(lp, sp);
	float d = dot(ld, lp);
	float l = len3(lp);
	float b = d * d - l * l + sr * sr;
	if(b < 0) return 0;
	float distance = -(dot(ld, lp)) + sqrtf(b);
	if(distance < 0) return 0;
	return 1;
}

static float line_box(v3 lp, v3 ld, v3 bpos, v3 bdim, v3* pos, v3* nrm) {
	v3 half = divs(bdim, 2);
	v3 bmin = sub(bpos, half);
	v3 bmax = add(bpos, half);
	v3 df = sdiv(1.0f, ld);
	float t1 = (bmin.x - lp.x) * df.x;
	float t2 = (bmax.x - lp.x) * df.x;
	float t3 = (bmin.y - lp.y) * df.y;
	float t4 = (bmax.y - lp.y) * df.y;
	float t5 = (bmin.z - lp.z) * df.z;
	float t6 = (bmax.z - lp.z) * df.z;
	float tmin = fmax2(fmax2(fmin2(t1, t2), fmin2(t3, t4)), fmin2(t5, t6));
	float tmax = fmin2(fmin2(fmax2(t1, t2), fmax2(t3, t4)), fmax2(t5, t6));
	if(tmax < 0) return -1;
	if(tmin > tmax) return -1;
	float distance;
	if(tmin < 0) distance = tmax;
	else distance = tmin;
	*pos = add(lp, scale(ld, distance));
	if(distance == t1) *nrm = V3(-1, 0, 0);
	else if(distance == t2) *nrm = V3(1, 0, 0);
	else if(distance == t3) *nrm = V3(0, -1, 0);
	else if(distance == t4) *nrm = V3(0, 1, 0);
	else if(distance == t5) *nrm = V3(0, 0, -1);
	else if(distance == t6) *nrm = V3(0, 0, 1);
	return distance;
}

static float line_cylinder(v3 p, v3 d, v3* pos, v3* nrm) {
	float a = d.x * d.x + d.y * d.y;
	float b = 2 * (p.x * d.x + p.y * d.y);
	float c = (p.x * p.x + p.y * p.y) - 1;
	float disc = b * b - 4 * a * c;
	if(disc < 0) return -1;
	float t1 = (-b + sqrtf(disc)) / (2 * a);
	float t2 = (-b - sqrtf(disc)) / (2 * a);
	float z1 = p.z + d.z * t1;
	float z2 = p.z + d.z * t2;
	int type1 = 0, type2 = 0;
	if(z1 > 1) { if(z2 <= 1) { type1 = 1; t1 = (1 - p.z) / d.z; } else type1 = 3; }
	else if(z1 < -1) { if(z2 >= -1) { type1 = 2; t1 = (-1 - p.z) / d.z; } else type1 = 3; }
	if(z2 > 1) { if(z1 <= 1) { type2 = 1; t2 = (1 - p.z) / d.z; } else type2 = 3; }
	else if(z2 < -1) { if(z1 >= -1) { type2 = 2; t2 = (-1 - p.z) / d.z; } else type2 = 3; }
	if(type1 == 3 && type2 == 3) return -1;
	if(t1 < 0 && t2 < 0) return -1;
	float distance = fmin2(t1, t2);
	if(distance < 0) distance = fmax2(t1, t2);
	*pos = add(p, scale(d, distance));
	{
		int type = distance == t1 ? type1 : type2;
		if(type == 0) {
			if(pos->x == 0 && pos->y == 0) *nrm = V3(1, 0, 0);
			else *nrm = norm3(V3(pos->x, pos->y, 0));
		}
		else if(type == 1) *nrm = V3(0, 0, 1);
		else *nrm = V3(0, 0, -1);
	}
	return distance;
}

static float line_cone(v3 p, v3 d, v3* pos, v3* nrm) {
	p.z -= 0.5f;
	float a = d.x * d.x + d.y * d.y - d.z * d.z;
	float b = 2 * p.x * d.x + 2 * p.y * d.y - 2 * p.z * d.z;
	float c = p.x * p.x + p.y * p.y - p.z * p.z;
	float disc = b * b - 4 * a * c;
	if(disc < 0) return -1;
	float t1 = (-b + sqrtf(disc)) / (2 * a);
	float t2 = (-b - sqrtf(disc)) / (2 * a);
	float z1 = p.z + d.z * t1;
	float z2 = p.z + d.z * t2;
	int type1 = 0, type2 = 0;
	if(z1 > 0 || z1 < -1) type1 = 2;
	if(z2 > 0 || z2 < -1) type2 = 2;
	if(type1 == 2 && type2 == 2) return -1;
	if(type1 == 2 || type2 == 2) {
		if(type1 == 2) t1 = (-1 - p.z) / d.z;
		else t2 = (-1 - p.z) / d.z;
	}
	if(t1 < 0 && t2 < 0) return -1;
	float distance = fmin2(t1, t2);
	if(distance < 0) distance = fmax2(t1, t2);
	p.z += 0.5f;
	*pos = add(p, scale(d, distance));
	{
		int type = distance == t1 ? type1 : type2;
		if(type == 0) {
			if(pos->x == 0 && pos->y == 0) {
				*nrm = V3(0, 0, 1);
			} else {
				v3 n = norm3(V3(pos->x, pos->y, 0));
				n = norm3(V3(n.x, n.y, 1));
				*nrm = n;
			}
		}
		else *nrm = V3(0, 0, -1);
	}
	return distance;
}

static float line_plane(v3 lp, v3 ld, v3 pp, v3 pn, v3* pos, v3* nrm) {
	float a = dot(pn, ld);
	if(a == 0) return -1;
	float distance = dot(pn, sub(pp, lp)) / a;
	if(distance >= 0) {
		if(pos) *pos = add(lp, scale(ld, distance));
		if(nrm) *nrm = a < 0 ? pn : neg(pn);
		return distance;
	}
	return -1;
}

/* code/raycast.cpp:750-770 */
static void to_unit_space(const obj_t* o, v3 pos, v3 dir, v3* pt, v3* dt) {
	pos = sub(pos, o->pos);
	if(o->is_rotated) pos = quat_rot(o->rot_inv, pos);
	pos = mul(pos, o->dim_scale_inv);
	*pt = pos;
	if(o->is_rotated) dir = quat_rot(o->rot_inv, dir);
	dir = mul(dir, o->dim_scale_inv);
	dir = norm3(dir);
	*dt = dir;
}
static void to_global_space(const obj_t* o, v3* pos, v3* dir) {
	*pos = mul(*pos, o->dim_scale);
	if(o->is_rotated) *pos = quat_rot(o->rot, *pos);
	*pos = add(*pos, o->pos);
	*dir = mul(*dir, o->dim_scale_inv);
	if(o->is_rotated) *dir = quat_rot(o->rot, *dir);
	*dir = norm3(*dir);
}

/* code/raycast.cpp:813-893 */
static float get_intersection(const obj_t* o, v3 rp, v3 rd, v3* pos, v3* nrm, counters_t* c) {
	float distance = -1;
	c->isect[o->type]++;
	switch(o->type) {
		case PT_GEOM_SPHERE: {
			if(o->even_dim) {
				distance = line_sphere(rp, rd, o->pos, o->dim.x * 0.5f, pos, nrm);
			} else {
				v3 lp, ld;
				c->uneven_sphere++;
				to_unit_space(o, rp, rd, &lp, &ld);
				distance = line_sphere(lp, ld, V3(0, 0, 0), 0.5f, pos, nrm);
				if(distance != -1) {
					to_global_space(o, pos, nrm);
					distance = len3(sub(rp, *pos));
				}
			}
		} break;
		case PT_GEOM_BOX: {
			if(!o->is_rotated) {
				distance = line_box(rp, rd, o->pos, o->dim, pos, nrm);
			} else {
				v3 lp, ld;
				c->rotated_box++;
				lp = sub(rp, o->pos);
				ld = add(lp, rd);
				lp = quat_rot(o->rot_inv, lp);
				ld = quat_rot(o->rot_inv, ld);
				ld = norm3(sub(ld, lp));
				distance = line_box(lp, ld, V3(0, 0, 0), o->dim, pos, nrm);
				if(distance != -1) {
					*pos = quat_rot(o->rot, *pos);
					*pos = add(*pos, o->pos);
					*nrm = quat_rot(o->rot, *nrm);
					distance = len3(sub(rp, *pos));
				}
			}
		} break;
		case PT_GEOM_CYLINDER: {
			v3 lp, ld;
			to_unit_space(o, rp, rd, &lp, &ld);
			distance = line_cylinder(lp, ld, pos, nrm);
			if(distance != -1) {
				to_global_space(o, pos, nrm);
				distance = len3(sub(rp, *pos));
			}
		} break;
		case PT_GEOM_CONE: {
			v3 lp, ld;
			to_unit_space(o, rp, rd, &lp, &ld);
			distance = line_cone(lp, ld, pos, nrm);
			if(distance != -1) {
				to_global_space(o, pos, nrm);
				distance = len3(sub(rp, *pos));
			}
		} break;
		case PT_GEOM_PLANE: {
			v3 pn = quat_rot(o->rot, V3(0, 0, 1));
			distance = line_plane(rp, rd, o->pos, pn, pos, nrm);
		} break;
	}
	return distance;
}

/* ---- optics: code/raycast.cpp:194-242 -------------------------------------------------------
 * clamp(-1, 1, x) binds as clamp(n=-1, min=1, max=x) = clampMax(clampMin(-1, 1), x) = (1 > x ? x : 1). */
static inline float clamp_m11(float x) { return clamp_max(clamp_min(-1.0f, 1.0f), x); }
static v3 refract_dir(v3 incident, v3 normal, float ior) {
	float cosi = clamp_m11(dot(incident, normal));
	float etai = 1;
	float etat = ior;
	v3 n = normal;
	if(cosi < 0) cosi = -cosi;
	else {
		float t = etai; etai = etat; etat = t;
		n = neg(normal);
	}
	float eta = etai / etat;
	float k = 1 - eta * eta * (1 - cosi * cosi);
	if(k < 0) {
		return V3(0, 0, 0);
	} else {
		return add(scale(incident, eta), scale(n, (eta * cosi - sqrtf(k))));
	}
}
static float fresnel(v3 incident, v3 normal, float ior) {
	float cosi = clamp_m11(dot(incident, normal));
	float etai = 1;
	float etat = ior;
	if(cosi > 0) { float t = etai; etai = etat; etat = t; }
	float sint = etai / etat * sqrtf(fmax2(0.0f, 1 - cosi * cosi));
	float kr;
	if(sint >= 1) {
		kr = 1;
	} else {
		float cost = sqrtf(fmax2(0.0f, 1 - sint * sint));
		cosi = fabsf(cosi);
		float Rs = ((etat * cosi) - (etai * cost)) / ((etat * cosi) + (etai * cost));
		float Rp = ((etai * cosi) - (etat * cost)) / ((etai * cosi) + (etat * cost));
		kr = (Rs * Rs + Rp * Rp) / 2;
	}
	return kr;
}

/* ---- precalc: code/raycast.cpp:80-139,2064-2108 --------------------------------------------- */
static void precalc_object(obj_t* o) {
	float r = 0;
	switch(o->type) {
		case PT_GEOM_BOX: r = len3(o->dim) * 0.5f; break;
		case PT_GEOM_SPHERE: r = fmax3(o->dim.x, o->dim.y, o->dim.z) * 0.5f; break;
		case PT_GEOM_CYLINDER: r = fmax2(o->dim.x, o->dim.y) * 0.5f + o->dim.z * 0.5f; break;
		case PT_GEOM_CONE: r = fmax2(o->dim.x, o->dim.y) * 0.5f + o->dim.z * 0.5f; break;
	}
	o->bradius = r;

	v3 dim = o->dim;
	if(o->type == PT_GEOM_PLANE) dim = V3(1000000, 1000000, 0);
	static const float corners[8][3] = {
		{ 0.5f, 0.5f, 0.5f }, { 0.5f, 0.5f, -0.5f }, { 0.5f, -0.5f, 0.5f }, { 0.5f, -0.5f, -0.5f },
		{ -0.5f, 0.5f, 0.5f }, { -0.5f, 0.5f, -0.5f }, { -0.5f, -0.5f, 0.5f }, { -0.5f, -0.5f, -0.5f },
	};
	v3 mn = V3(FLT_MAX, FLT_MAX, FLT_MAX), mx = V3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
	for(int i = 0; i < 8; i++) {
		v3 p = mul(dim, V3(corners[i][0], corners[i][1], corners[i][2]));
		p = quat_rot(o->rot, p);
		mn.x = fmin2(mn.x, p.x); mn.y = fmin2(mn.y, p.y); mn.z = fmin2(mn.z, p.z);
		mx.x = fmax2(mx.x, p.x); mx.y = fmax2(mx.y, p.y); mx.z = fmax2(mx.z, p.z);
	}
	o->bbox = V3(mx.x * 2, mx.y * 2, mx.z * 2);

	o->color_lin = V3(powf(o->color.x, 2.2f), powf(o->color.y, 2.2f), powf(o->color.z, 2.2f));
	o->is_rotated = !(o->rot.w == 1 && o->rot.x == 0 && o->rot.y == 0 && o->rot.z == 0);
	o->even_dim = (o->dim.x == o->dim.y && o->dim.x == o->dim.z);
	o->emit_lin = V3(powf(o->emit.x, 2.2f), powf(o->emit.y, 2.2f), powf(o->emit.z, 2.2f));
	o->rot_inv = quat_inverse(o->rot);
	v3 sc = V3(1, 1, 1);
	switch(o->type) {
		case PT_GEOM_SPHERE: case PT_GEOM_BOX: sc = V3(1, 1, 1); break;
		case PT_GEOM_CYLINDER: sc = V3(2, 2, 2); break;
		case PT_GEOM_CONE: sc = V3(2, 2, 1); break;
		default: o->dim_scale = V3(0, 0, 0); o->dim_scale_inv = V3(0, 0, 0); return;
	}
	o->dim_scale = divv(o->dim, sc);
	o->dim_scale_inv = divv(sc, o->dim);
}

/* code/rt_math.cpp:258-260,340-347; code/raycast.cpp:173-191,271-277; code/app.cpp:1077-1090,1230-1241 */
static void precalc_world(pto_scene* s) {
	const pt_world* w = &s->win;
	s->cam_pos = V3(w->camera.pos[0], w->camera.pos[1], w->camera.pos[2]);
	s->dim_w = w->camera.plane_width;
	float aspect = (float)s->st.width / s->st.height;
	s->dim_h = s->dim_w * (1 / aspect);
	{
		float angle = w->camera.fov * ((float)3.1415926535897932384626433832795 / 180);
		float side_angle = (float)((3.1415926535897932384626433832795 - angle) / 2.0f);
		float side = s->dim_w / sinf(angle) * sinf(side_angle);
		s->near_dist = side * sinf(side_angle);
	}
	{
		v3 bdir = V3(0, 1, 0), bup = V3(0, 0, 1);
		v3 dir = bdir;
		dir = quat_rot(quat_axis(w->camera.rot[0], bup), dir);
		dir = quat_rot(quat_axis(w->camera.rot[1], norm3(cross(bup, dir))), dir);
		s->cam_dir = dir;
		s->cam_up = norm3(cross(dir, norm3(cross(bup, dir))));
		s->cam_right = neg(norm3(cross(s->cam_up, dir)));
	}
	s->cam_top_left = add(add(add(s->cam_pos, scale(s->cam_dir, s->near_dist)),
	                          scale(scale(s->cam_right, -1), (s->dim_w / 2.0f))),
	                      scale(s->cam_up, (s->dim_h / 2.0f)));
	{
		float axy = (float)w->light_rot[0] * ((float)3.1415926535897932384626433832795 / 180);
		float az = (float)w->light_rot[1] * ((float)3.1415926535897932384626433832795 / 180);
		quat q = quat_mul(quat_axis(axy, V3(0, 0, -1)), quat_axis(-az, V3(0, 1, 0)));
		s->light_dir = quat_rot(q, V3(1, 0, 0));
	}
	s->bg_lin = V3(powf(w->background_color[0], 2.2f), powf(w->background_color[1], 2.2f), powf(w->background_color[2], 2.2f));
	s->ambient_lin = V3(powf(w->ambient_color[0], 2.2f), powf(w->ambient_color[1], 2.2f), powf(w->ambient_color[2], 2.2f));
	s->light_lin = V3(powf(w->light_color[0], 2.2f), powf(w->light_color[1], 2.2f), powf(w->light_color[2], 2.2f));
	s->focal_distance = w->focal_distance;
	s->aperture = w->aperture;
}

/* ---- tables: code/app.cpp:669-727 ----------------------------------------------------------- */
static void make_tables(pcg32* rng, v3* dirs, v2* disc) {
	float precision = 0.001f;
	for(int i = 0; i < PT_TABLE_COUNT; i++) {
		v3 d;
		do {
			/* the three draws are call arguments of vec3(); the reference build (g++, as MSVC)
			 * evaluates them right to left, so z is drawn first */
			d.z = random_float(rng, -1, 1, precision);
			d.y = random_float(rng, -1, 1, precision);
			d.x = random_float(rng, -1, 1, precision);
		} while(len3(d) > 1);
		d = norm3(d);
		if(d.x == 0 && d.y == 0 && d.z == 0) d = V3(1, 0, 0);
		dirs[i] = d;
	}
	float range = 0.5f;
	for(int i = 0; i < PT_TABLE_COUNT; i++) {
		v2 p;
		do {
			p.y = random_float(rng, -range, range, precision);   /* right-to-left, as above */
			p.x = random_float(rng, -range, range, precision);
		} while(sqrtf(p.x * p.x + p.y * p.y) > range);
		if(p.x == 0 && p.y == 0) { p.x = range; p.y = 0; }
		disc[i] = p;
	}
}

/* ---- blue noise: code/rt_math.cpp:3488-3585 (region = [0,w]x[0,w]) ------------------------- */
static inline int imod(int a, int b) { int r = a % b; if(r < 0) r += b; return r; }
static inline float round_down_float(float i) { return i < 0 ? (float)(((int)i) - 1) : (float)((int)i); }
static int blue_noise(pcg32* rng, float width, float radius, v2** out) {
	v2 dim = { width, width };
	float cs = (float)(radius / 1.41421356237309504880);
	int gw = (int)ceilf(dim.x / cs), gh = (int)ceilf(dim.y / cs);
	int grid_size = gw * gh;
	v2* samples = (v2*)malloc(sizeof(v2) * (size_t)grid_size);
	int sample_count = 0;
	float prec = 0.0001f;
	int test_count = 30;
	int* grid = (int*)malloc(sizeof(int) * (size_t)(grid_size + 1));   /* +1: see the guard below */
	for(int i = 0; i < grid_size; i++) grid[i] = -1;
	int* active = (int*)malloc(sizeof(int) * (size_t)grid_size);
	int active_count = 0;

	samples[sample_count].y = random_float(rng, 0, dim.y, prec);   /* right-to-left, as above */
	samples[sample_count].x = random_float(rng, 0, dim.x, prec);
	sample_count++;
	active[active_count++] = 0;
	{
		v2 pos = samples[0];
		int gx = (int)(pos.x / cs), gy = (int)(pos.y / cs);
		grid[gy * gw + gx] = 0;
	}
	v2 region_dim = dim;
	while(active_count > 0) {
		int active_index = random_int(rng, 0, active_count - 1);
		int sample_index = active[active_index];
		v2 as = samples[sample_index];
		for(int i = 0; i < test_count; ++i) {
			float angle = random_float(rng, 0, (float)6.2831853071795864769252867665590, prec);
			float distance = random_float(rng, radius, radius * 2, prec);
			v2 ns;
			ns.x = as.x + cosf(angle) * distance;
			ns.y = as.y + sinf(angle) * distance;
			if(!(ns.x >= 0 && ns.x <= dim.x && ns.y >= 0 && ns.y <= dim.y)) continue;

			int minx = (int)round_down_float((ns.x - radius * 2.0f) / cs);
			int miny = (int)round_down_float((ns.y - radius * 2.0f) / cs);
			int maxx = (int)round_down_float((ns.x + radius * 2.0f) / cs);
			int maxy = (int)round_down_float((ns.y + radius * 2.0f) / cs);
			int valid = 1;
			for(int y = miny; y <= maxy; ++y) {
				for(int x = minx; x <= maxx; ++x) {
					int mx = imod(x, gw);
					int my = imod(y, gh);
					int index = grid[my * gw + mx];
					if(index > -1) {
						v2 sp = samples[index];
						if(mx != x) sp.x += mx > x ? -region_dim.x : region_dim.x;
						if(my != y) sp.y += my > y ? -region_dim.y : region_dim.y;
						float dx = sp.x - ns.x, dy = sp.y - ns.y;
						float dist = sqrtf(dx * dx + dy * dy);
						if(dist < radius) { valid = 0; break; }
					}
				}
				if(!valid) break;
			}
			if(valid) {
				/* The reference indexes the cell grid with the truncated position even when the
				 * sample sits exactly on the far border (index one past the row / the array);
				 * the slot is clamped into the allocation here. */
				int gx = (int)(ns.x / cs), gy = (int)(ns.y / cs);
				int gi = gy * gw + gx;
				if(gi > grid_size) gi = grid_size;
				if(sample_count >= grid_size) break;
				samples[sample_count] = ns;
				active[active_count++] = sample_count;
				grid[gi] = sample_count;
				sample_count++;
			}
		}
		active[active_index] = active[active_count - 1];
		active_count--;
	}
	free(grid);
	free(active);
	*out = samples;
	return sample_count;
}

/* code/app.cpp:1161-1225 */
static void make_samples(pto_scene* s, pcg32* rng) {
	int gw = s->st.sample_grid_width;
	int count_grid = (int)sqrt((double)s->st.spp_wanted);
	float scg = count_grid * 1.3f;
	v2* bn;
	int n = blue_noise(rng, (float)gw, 1 / (float)scg, &bn);
	s->sample_count = n;
	s->grid_w = gw;
	s->samples = (v2*)malloc(sizeof(v2) * (size_t)(n > 0 ? n : 1));
	int cells = gw * gw + 1;
	s->offsets = (int*)calloc((size_t)cells, sizeof(int));
	for(int i = 0; i < n; i++) {
		int tx = (int)bn[i].x, ty = (int)bn[i].y;
		if(tx > gw - 1) tx = gw - 1;
		if(ty > gw - 1) ty = gw - 1;
		s->offsets[1 + ty * gw + tx] += 1;
	}
	for(int i = 0; i < cells - 1; i++) s->offsets[i + 1] += s->offsets[i];
	int* counter = (int*)calloc((size_t)cells, sizeof(int));
	for(int i = 0; i < n; i++) {
		int tx = (int)bn[i].x, ty = (int)bn[i].y;
		if(tx > gw - 1) tx = gw - 1;
		if(ty > gw - 1) ty = gw - 1;
		int gi = ty * gw + tx;
		int index = s->offsets[gi] + counter[gi];
		counter[gi]++;
		s->samples[index].x = bn[i].x - (float)tx;
		s->samples[index].y = bn[i].y - (float)ty;
	}
	free(counter);
	free(bn);
}

/* ---- grid build: code/raycast.cpp:595-617, code/app.cpp:1247-1357 --------------------------- */
static void object_cell_range(const grid_t* g, const obj_t* o, v3i* lo, v3i* hi) {
	v3 half = divs(o->bbox, 2);
	v3 bmin = sub(o->pos, half);
	v3 bmax = add(o->pos, half);
	const float mn[3] = { bmin.x, bmin.y, bmin.z }, mx[3] = { bmax.x, bmax.y, bmax.z };
	const int cs[3] = { g->cell_size.x, g->cell_size.y, g->cell_size.z };
	const int cm[3] = { g->coord_min.x, g->coord_min.y, g->coord_min.z };
	const int cc[3] = { g->cell_count.x, g->cell_count.y, g->cell_count.z };
	int l[3], h[3];
	for(int i = 0; i < 3; i++) {
		l[i] = (int)(mn[i] / cs[i]) - cm[i];
		if(mn[i] < 0) l[i]--;
		l[i] = clamp_int(l[i], 0, cc[i] - 1);
		h[i] = (int)(mx[i] / cs[i]) - cm[i];
		if(mx[i] < 0) h[i]--;
		h[i] = clamp_int(h[i], 0, cc[i] - 1);
	}
	lo->x = l[0]; lo->y = l[1]; lo->z = l[2];
	hi->x = h[0]; hi->y = h[1]; hi->z = h[2];
}

static void build_grid(pto_scene* s) {
	v3 smin = V3(FLT_MAX, FLT_MAX, FLT_MAX), smax = V3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
	for(int i = 0; i < s->n; i++) {
		obj_t* o = s->obj + i;
		if(o->type == PT_GEOM_PLANE) continue;
		v3 half = divs(o->bbox, 2);
		v3 bmin = sub(o->pos, half);
		v3 bmax = add(o->pos, half);
		smin.x = fmin2(smin.x, bmin.x); smin.y = fmin2(smin.y, bmin.y); smin.z = fmin2(smin.z, bmin.z);
		smax.x = fmax2(smax.x, bmax.x); smax.y = fmax2(smax.y, bmax.y); smax.z = fmax2(smax.z, bmax.z);
	}
	grid_t* g = &s->grid;
	int hint = s->st.grid_hint;
	const float bmn[3] = { smin.x, smin.y, smin.z }, bmx[3] = { smax.x, smax.y, smax.z };
	int cs[3], cmin[3], cmax[3];
	for(int i = 0; i < 3; i++) {
		float dim = bmx[i] - bmn[i];
		cs[i] = (int)((float)dim / hint);
		if(cs[i] == 0) cs[i] = 1;
	}
	for(int i = 0; i < 3; i++) cmin[i] = round_int(bmn[i] / cs[i] - 0.5f);
	for(int i = 0; i < 3; i++) cmax[i] = round_int(bmx[i] / cs[i] + 0.5f);
	g->cell_size.x = cs[0]; g->cell_size.y = cs[1]; g->cell_size.z = cs[2];
	g->coord_min.x = cmin[0]; g->coord_min.y = cmin[1]; g->coord_min.z = cmin[2];
	g->coord_max.x = cmax[0]; g->coord_max.y = cmax[1]; g->coord_max.z = cmax[2];
	g->cell_count.x = cmax[0] - cmin[0]; g->cell_count.y = cmax[1] - cmin[1]; g->cell_count.z = cmax[2] - cmin[2];
	g->bmin.x = cmin[0] * cs[0]; g->bmin.y = cmin[1] * cs[1]; g->bmin.z = cmin[2] * cs[2];
	g->bmax.x = cmax[0] * cs[0]; g->bmax.y = cmax[1] * cs[1]; g->bmax.z = cmax[2] * cs[2];

	long long cells = (long long)g->cell_count.x * g->cell_count.y * g->cell_count.z;
	if(cells <= 0 || cells > (1LL << 28)) { s->cells = 0; s->cell_start = (uint32_t*)calloc(1, sizeof(uint32_t)); s->cell_items = 0; return; }
	s->cells = (int)cells;
	s->cell_start = (uint32_t*)calloc((size_t)cells + 1, sizeof(uint32_t));
	for(int pass = 0; pass < 2; pass++) {
		uint32_t* fill = 0;
		if(pass == 1) {
			uint32_t run = 0;
			for(long long c = 0; c < cells; c++) { uint32_t k = s->cell_start[c]; s->cell_start[c] = run; run += k; }
			s->cell_start[cells] = run;
			s->cell_items = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(run > 0 ? run : 1));
			fill = (uint32_t*)calloc((size_t)cells, sizeof(uint32_t));
		}
		for(int i = 0; i < s->n; i++) {
			v3i lo, hi;
			object_cell_range(g, s->obj + i, &lo, &hi);
			for(int z = lo.z; z < hi.z + 1; z++) for(int y = lo.y; y < hi.y + 1; y++) for(int x = lo.x; x < hi.x + 1; x++) {
				long long c = (long long)z * g->cell_count.y * g->cell_count.x + (long long)y * g->cell_count.x + x;
				if(pass == 0) s->cell_start[c]++;
				else s->cell_items[s->cell_start[c] + fill[c]++] = (uint32_t)i;
			}
		}
		free(fill);
	}
}

/* ---- traversal: code/raycast.cpp:619-741 ---------------------------------------------------- */
static void walk_init(const grid_t* g, walk_t* w, v3 rp, v3 rd) {
	v3 bmin = V3((float)g->bmin.x, (float)g->bmin.y, (float)g->bmin.z);
	v3 bmax = V3((float)g->bmax.x, (float)g->bmax.y, (float)g->bmax.z);
	if(!point_in_box(rp, bmin, bmax)) {
		v3 bdim = sub(bmax, bmin);
		v3 bpos = add(bmin, divs(bdim, 2));
		v3 ip = V3(0, 0, 0), in = V3(0, 0, 0);
		float distance = line_box(rp, rd, bpos, bdim, &ip, &in);
		if(distance != -1) {
			rp = add(ip, scale(rd, 0.0001f));
			w->outside = 0;
		} else {
			w->outside = 1;
			return;
		}
	}
	const float p[3] = { rp.x, rp.y, rp.z }, d[3] = { rd.x, rd.y, rd.z };
	const int cs[3] = { g->cell_size.x, g->cell_size.y, g->cell_size.z };
	const int cm[3] = { g->coord_min.x, g->coord_min.y, g->coord_min.z };
	const int cc[3] = { g->cell_count.x, g->cell_count.y, g->cell_count.z };
	int coord[3], step[3]; float tmax[3], tdelta[3];
	for(int i = 0; i < 3; i++) {
		coord[i] = (int)(p[i] / cs[i]) - cm[i];
		if(p[i] < 0) coord[i]--;
		coord[i] = clamp_int(coord[i], 0, cc[i] - 1);
	}
	for(int i = 0; i < 3; i++) {
		if(d[i] == 0) step[i] = 0;
		else step[i] = d[i] > 0 ? 1 : -1;
	}
	for(int i = 0; i < 3; i++) {
		int c = step[i] == -1 ? coord[i] : coord[i] + 1;
		c += cm[i];
		if(step[i] != 0) tmax[i] = ((c * cs[i]) - p[i]) / d[i];
		else tmax[i] = FLT_MAX;
	}
	for(int i = 0; i < 3; i++) {
		tdelta[i] = cs[i] / d[i];
		if(tdelta[i] < 0) tdelta[i] = -tdelta[i];
	}
	w->coord.x = coord[0]; w->coord.y = coord[1]; w->coord.z = coord[2];
	w->step.x = step[0]; w->step.y = step[1]; w->step.z = step[2];
	w->tmax = V3(tmax[0], tmax[1], tmax[2]);
	w->tdelta = V3(tdelta[0], tdelta[1], tdelta[2]);
	w->box_min = V3((float)((coord[0] + cm[0]) * cs[0]), (float)((coord[1] + cm[1]) * cs[1]), (float)((coord[2] + cm[2]) * cs[2]));
	w->box_max = V3((float)((coord[0] + 1 + cm[0]) * cs[0]), (float)((coord[1] + 1 + cm[1]) * cs[1]), (float)((coord[2] + 1 + cm[2]) * cs[2]));
}

#define WALK_AXIS(A) do { \
		w->coord.A = w->coord.A + w->step.A; \
		if(w->coord.A < 0 || w->coord.A >= g->cell_count.A) return 0; \
		w->tmax.A += w->tdelta.A; \
		if(w->step.A < 0) { \
			w->box_max.A = w->box_min.A; \
			w->box_min.A = (float)((w->coord.A + g->coord_min.A) * g->cell_size.A); \
		} else { \
			w->box_min.A = w->box_max.A; \
			w->box_max.A = (float)((w->coord.A + g->coord_min.A + 1) * g->cell_size.A); \
		} } while(0)

static int walk_next(const grid_t* g, walk_t* w) {
	if(w->tmax.x < w->tmax.y) {
		if(w->tmax.x < w->tmax.z) WALK_AXIS(x);
		else WALK_AXIS(z);
	} else {
		if(w->tmax.y < w->tmax.z) WALK_AXIS(y);
		else WALK_AXIS(z);
	}
	return 1;
}

#define HASH_INIT 1469598103934665603ULL
static inline uint64_t hash_cell(uint64_t h, int index) { return (h ^ (uint64_t)(uint32_t)(index + 1)) * 1099511628211ULL; }

/* ---- casts: code/raycast.cpp:895-986 --------------------------------------------------------- */
static int cast_all(pto_scene* s, v3 rp, v3 rd, int outside, v3* ipos, v3* inrm) {
	int id = 0;
	float min_d = FLT_MAX;
	for(int i = 0; i < s->n; i++) {
		const obj_t* o = s->obj + i;
		if(outside && o->type != PT_GEOM_PLANE) continue;
		v3 p = V3(0, 0, 0), q = V3(0, 0, 0);
		float distance = -1;
		int possible;
		if(o->type == PT_GEOM_PLANE) possible = 1;
		else { s->cnt.bsphere_tests++; possible = sphere_collision(rp, rd, o->pos, o->bradius); }
		if(possible) {
			distance = get_intersection(o, rp, rd, &p, &q, &s->cnt);
			if(distance > 0 && distance < min_d) {
				min_d = distance;
				id = i + 1;
				*ipos = p;
				*inrm = q;
			}
		}
	}
	return id;
}

/* Diagnosis: called for every hit the grid cast accepts as its best so far (entity, ray, hit point, distance). */
static void (*g_hit_hook)(int, const float*, const float*, const float*, float) = 0;
void pto_set_hit_hook(void (*hook)(int, const float*, const float*, const float*, float)) { g_hit_hook = hook; }

static int cast_grid(pto_scene* s, v3 rp, v3 rd, v3* ipos, v3* inrm, uint64_t* hash) {
	const grid_t* g = &s->grid;
	walk_t w;
	w.outside = s->sticky_outside ? s->outside_state : 0;
	walk_init(g, &w, rp, rd);
	if(s->sticky_outside) s->outside_state = w.outside;
	int id = 0;
	float min_d = FLT_MAX;
	if(w.outside) {
		/* (the reference walks all objects and picks the planes out, code/raycast.cpp:938-939: the same planes in the same order) */
		for(int k = 0; k < s->n_planes; k++) {
			const int i = s->plane_ids[k];
			{
				v3 p = V3(0, 0, 0), q = V3(0, 0, 0);
				float distance = get_intersection(s->obj + i, rp, rd, &p, &q, &s->cnt);
				if(distance != -1) {
					min_d = distance;
					id = i + 1;
					*ipos = p;
					*inrm = q;
				}
			}
		}
		return id;
	}
	do {
		int index = w.coord.z * g->cell_count.y * g->cell_count.x + w.coord.y * g->cell_count.x + w.coord.x;
		if(hash) *hash = hash_cell(*hash, index);
		s->cnt.dda_steps++;
		for(uint32_t k = s->cell_start[index]; k < s->cell_start[index + 1]; k++) {
			int oi = (int)s->cell_items[k];
			v3 p = V3(0, 0, 0), q = V3(0, 0, 0);
			float distance = get_intersection(s->obj + oi, rp, rd, &p, &q, &s->cnt);
			if(distance > 0 && distance < min_d && point_in_box(p, w.box_min, w.box_max)) {
				min_d = distance;
				id = oi + 1;
				*ipos = p;
				*inrm = q;
				if(g_hit_hook) g_hit_hook(oi, &rp.x, &rd.x, &p.x, distance);
			}
		}
		if(id != 0) break;
	} while(walk_next(g, &w));
	return id;
}

/* code/raycast.cpp:1100-1107 */
static int cast(pto_scene* s, v3 rp, v3 rd, v3* ipos, v3* inrm, uint64_t* hash) {
	s->cnt.casts++;
	if(s->n < s->st.brute_force_below) {
		walk_t w;
		w.outside = s->sticky_outside ? s->outside_state : 0;
		walk_init(&s->grid, &w, rp, rd);
		if(s->sticky_outside) s->outside_state = w.outside;
		return cast_all(s, rp, rd, w.outside, ipos, inrm);
	}
	return cast_grid(s, rp, rd, ipos, inrm, hash);
}

/* ---- the per-pixel loop: code/raycast.cpp:991-1289 ------------------------------------------ */
typedef struct {
	v3 right, down, back, right_width, down_height, right_px, down_px, light_dir;
	int use_aperture;
} frame_t;

static void frame_init(const pto_scene* s, frame_t* f) {
	f->right = s->cam_right;
	f->down = neg(s->cam_up);
	f->back = neg(s->cam_dir);
	f->right_width = scale(s->cam_right, s->dim_w);
	f->down_height = scale(f->down, s->dim_h);
	f->right_px = scale(scale(f->right, s->dim_w), (1.0f / s->st.width));
	f->down_px = scale(scale(f->down, s->dim_h), (1.0f / s->st.height));
	f->light_dir = norm3(s->light_dir);
	f->use_aperture = s->aperture != 0.0f;
}

static inline v3 pixel_pos(const pto_scene* s, const frame_t* f, int x, int y) {
	float px = x / (float)s->st.width, py = y / (float)s->st.height;
	return add(add(s->cam_top_left, scale(f->right_width, px)), scale(f->down_height, py));
}

/* One camera sample: ray generation (code/raycast.cpp:1059-1084) and the bounce loop (:1089-1261).
 * lens_index is the value of randomUnitIndex for this sample. */
static v3 trace_sample(pto_scene* s, const frame_t* f, v3 cam_pixel_pos, v2 smp, int lens_index, pcg32* rng,
                       int* first_hit, uint64_t* first_hash) {
	v3 ray_pos = add(add(cam_pixel_pos, scale(f->right_px, smp.x)), scale(f->down_px, smp.y));
	v3 ray_dir;
	if(!f->use_aperture) {
		ray_dir = norm3(sub(ray_pos, s->cam_pos));
	} else {
		v3 focal_point = V3(0, 0, 0);
		v3 rdir = sub(ray_pos, s->cam_pos);
		v3 focal_plane_pos = add(s->cam_pos, scale(s->cam_dir, s->focal_distance));
		line_plane(ray_pos, rdir, focal_plane_pos, f->back, &focal_point, 0);
		int pi = lens_index % PT_TABLE_COUNT;
		float ox = s->disc[pi].x * s->aperture;
		float oy = s->disc[pi].y * s->aperture;
		ray_pos = add(ray_pos, add(scale(s->cam_right, ox), scale(f->down, oy)));
		ray_dir = norm3(sub(focal_point, ray_pos));
		s->cnt.lens++;
	}
	s->cnt.paths++;

	v3 color = V3(0, 0, 0);
	v3 att = V3(1, 1, 1);
	int path_casts = 0;
	for(int ray = 0; ray < s->st.max_bounces; ray++) {
		v3 ip = V3(0, 0, 0), in = V3(0, 0, 0);
		uint64_t h = HASH_INIT;
		int id = cast(s, ray_pos, ray_dir, &ip, &in, &h);
		path_casts++;
		if(ray == 0) {
			if(first_hit) *first_hit = id;
			if(first_hash) *first_hash = (s->n < s->st.brute_force_below) ? 0 : h;
		}
		if(id != 0) {
			const obj_t* o = s->obj + (id - 1);
			if(!(o->emit_lin.x == 0 && o->emit_lin.y == 0 && o->emit_lin.z == 0)) {
				color = add(color, mul(att, o->emit_lin));
				s->cnt.emit++;
			}
			if(o->ior != 1.0f) {
				s->cnt.shade_glass++;
				float ratio = fresnel(ray_dir, in, o->ior);
				v3 rdir = refract_dir(ray_dir, in, o->ior);
				if(rdir.x == 0 && rdir.y == 0 && rdir.z == 0) ratio = 1;
				v3 bias = scale(in, 0.0001f);
				int outside = dot(ray_dir, in) < 0;
				float rnd = random_float(rng, 0, 1.0f, 0.001f);
				if(rnd < ratio) {
					ray_dir = reflect_vector(ray_dir, in);
					ray_pos = ip;
					ray_pos = add(ray_pos, outside ? bias : neg(bias));
				} else {
					ray_dir = rdir;
					ray_pos = ip;
					ray_pos = add(ray_pos, outside ? neg(bias) : bias);
				}
				att = mul(att, o->color_lin);
			} else {
				s->cnt.shade_opaque++;
				ray_pos = add(ip, scale(in, 0.0001f));
				v3 refl = V3(0, 0, 0);
				if(o->refl != 0.0f) refl = reflect_vector(ray_dir, in);
				v3 rnd = V3(0, 0, 0);
				if(o->refl != 1.0f) {
					int di = random_int(rng, 0, PT_TABLE_COUNT - 1);
					rnd = s->dirs[di];
					float d = dot(rnd, in);
					if(d < 0) rnd = reflect_vector(rnd, in);
				}
				v3 ray_out = neg(ray_dir);
				v3 ray_in = lerp_check3(o->refl, rnd, refl);
				ray_dir = ray_in;
				float diffuse = lerp_check(1 - o->refl, 1, dot(ray_in, in));
				att = scale(divs(mul(att, o->color_lin), (float)3.1415926535897932384626433832795), diffuse);
				if(o->rough != 0.0f) {
					s->cnt.shade_oren++;
					float r2 = o->rough * o->rough;
					float a = 1 - 0.5f * (r2 / (r2 + 0.33f));
					float b = 0.45f * (r2 / (r2 + 0.09f));
					float cos_in = dot(ray_dir, in);
					float cos_out = dot(ray_out, in);
					v3 lp = norm3(sub(ray_dir, scale(in, cos_in)));
					v3 vp = norm3(sub(ray_out, scale(in, cos_out)));
					float cos_phi = dot(lp, vp);
					float th_in = acosf(cos_in);
					float th_out = acosf(cos_out);
					float alpha = fmax2(th_in, th_out);
					float beta = fmin2(th_in, th_out);
					float orna = a + (b * fmax2(0, cos_phi) * sinf(alpha) * tanf(beta));
					att = scale(att, orna);
				}
			}
			if(att.x == 0 && att.y == 0 && att.z == 0) break;
		} else if(ray == 0) {
			color = add(color, s->bg_lin);
			break;
		} else {
			float ld = dot(ray_dir, f->light_dir);
			ld = clamp_min(ld, 0);
			v3 light = scale(s->light_lin, ld);
			color = add(color, mul(att, add(s->ambient_lin, light)));
			break;
		}
	}
	s->cnt.path_casts[path_casts < PTO_HIST_BINS ? path_casts : PTO_HIST_BINS - 1]++;
	return color;
}

#define PT_STREAM_STRIDE 4096ULL

/* Seed policy PT_SEED_PER_SAMPLE: stream ids of one pixel. */
static inline void pixel_stream(const pto_scene* s, int x, int y, pcg32* r) {
	pcg_seed(r, s->st.seed, (uint64_t)y * (uint64_t)s->st.width + (uint64_t)x);
}
static inline void sample_stream(const pto_scene* s, int x, int y, int si, uint64_t pass, pcg32* r) {
	uint64_t wh = (uint64_t)s->st.width * (uint64_t)s->st.height;
	uint64_t pix = (uint64_t)y * (uint64_t)s->st.width + (uint64_t)x;
	pcg_seed(r, s->st.seed + pass, wh + pix * PT_STREAM_STRIDE + (uint64_t)si);
}

/* Range count of a render under PT_SEED_PER_SAMPLE (include/pt_api.h, pt_settings.sample_ranges):
 * the asked count, or for 0 the largest power of two not above spp_wanted / 16, within 1..16.  The
 * reference itself adds all samples of a pixel in one sequence (code/raycast.cpp:1263). */
static int sample_ranges(const pt_settings* st) {
	if(st->frame_index >= 0) return 1;
	if(st->sample_ranges > 0) return st->sample_ranges;
	int ranges = 1;
	while(ranges < 16 && ranges * 2 * 16 <= st->spp_wanted) ranges *= 2;
	return ranges;
}

/* First sample of range r of a pixel that has `count` samples, cut into `ranges` even ranges. */
static inline int range_start(int count, int range, int ranges) { return (int)(((long long)range * count) / ranges); }

/* One pixel, all its samples. rng is the job's generator under PT_SEED_PER_JOB and scratch
 * otherwise. per_sample (optional) receives the unclamped radiance of every sample.
 * Under PT_SEED_PER_JOB the clamped sample colours are added in one sequence, as the reference
 * does; under PT_SEED_PER_SAMPLE they are added range by range (range_start above) and the range
 * sums are then added in range order. */
static v3 trace_pixel(pto_scene* s, const frame_t* f, int x, int y, int policy, pcg32* rng, float* per_sample,
                      int* first_hit, uint64_t* first_hash) {
	v3 cpp = pixel_pos(s, f, x, y);
	int gw = s->grid_w;
	int tile = (y % gw) * gw + (x % gw);
	int offset = s->offsets[tile];
	int count = s->offsets[tile + 1] - offset;
	const v2* smp = s->samples + offset;
	int lens;
	if(policy == PT_SEED_PER_JOB) {
		lens = random_int(rng, 0, PT_TABLE_COUNT - 1);
	} else {
		pcg32 pr;
		pixel_stream(s, x, y, &pr);
		lens = random_int(&pr, 0, PT_TABLE_COUNT - 1);
	}
	if(first_hit) *first_hit = -1;
	if(first_hash) *first_hash = 0;
	v3 pixel = V3(0, 0, 0);
	const int ranges = policy == PT_SEED_PER_JOB ? 1 : sample_ranges(&s->st);
	for(int range = 0; range < ranges; range++) {
		v3 part = V3(0, 0, 0);
		for(int si = range_start(count, range, ranges); si < range_start(count, range + 1, ranges); si++) {
			pcg32 sr;
			pcg32* r = rng;
			if(policy != PT_SEED_PER_JOB) { sample_stream(s, x, y, si, 0, &sr); r = &sr; }
			v3 c = trace_sample(s, f, cpp, smp[si], lens + si, r, si == 0 ? first_hit : 0, si == 0 ? first_hash : 0);
			if(per_sample) { per_sample[3 * si] = c.x; per_sample[3 * si + 1] = c.y; per_sample[3 * si + 2] = c.z; }
			part = add(part, clamp_max3(c, 1));
		}
		pixel = range == 0 ? part : add(pixel, part);
	}
	if(count == 0) return V3(0, 0, 0);
	return clamp_max3(divs(pixel, (float)count), 1);
}

/* Tile jobs in the reference's centre-out spiral: code/app.cpp:1359-1419.  The reference stores
 * them in a fixed array of 256 and overruns it for image sizes that yield more tiles (e.g. 64x36);
 * here the list is sized for the image.  xywh may be NULL to count. */
static int make_jobs(int width, int height, int* xywh, int cap) {
	float dw = (float)width, dh = (float)height;
	int count = (int)(256 * 0.8f);
	float size = sqrtf(((dw * dh) / count));
	int tile = (int)size;
	float rcx = dw / tile, rcy = dh / tile;
	float cenx = rcx / 2, ceny = rcy / 2;
	int six = (int)cenx, siy = (int)ceny;
	int lim_x = (int)ceilf(rcx), lim_y = (int)ceilf(rcy);
	int jobs = 0;
	int max_off = (int)((fmax2(rcx, rcy) / 2) + 1);
	for(int off = 0; off < max_off + 1; off++) {
		int spx = -off, spy = off - 1;
		if(off == 0) { spx = 0; spy = 0; }
		int px = spx, py = spy;
		int dx = 0, dy = -1;
		for(;;) {
			int rx = six + px, ry = siy + py;
			if(rx >= 0 && rx < lim_x && ry >= 0 && ry < lim_y) {
				if(xywh && jobs < cap) {
					xywh[4 * jobs] = rx * tile; xywh[4 * jobs + 1] = ry * tile;
					xywh[4 * jobs + 2] = tile; xywh[4 * jobs + 3] = tile;
				}
				jobs++;
			}
			if(off == 0) break;
			if(abs(px) == off && abs(py) == off) { int t = dx; dx = -dy; dy = t; }
			px += dx; py += dy;
			if(px == spx && py == spy) break;
		}
	}
	return jobs;
}

static void run_job(pto_scene* s, const frame_t* f, const int* job, int job_index, int policy, uint64_t seed_base, float* rgb) {
	pcg32 rng;
	pcg_seed(&rng, 0, seed_base + (uint64_t)job_index);
	s->sticky_outside = policy == PT_SEED_PER_JOB;
	s->outside_state = 0;
	int w = s->st.width, h = s->st.height;
	int pixels = job[2] * job[3];
	for(int pi = 0; pi < pixels; pi++) {
		int x = pi % job[2] + job[0];
		int y = pi / job[2] + job[1];
		if(x < 0 || x >= w || y < 0 || y >= h) continue;
		v3 c = trace_pixel(s, f, x, y, policy, &rng, 0, 0, 0);
		float* o = rgb + 3 * ((size_t)y * w + x);
		o[0] = c.x; o[1] = c.y; o[2] = c.z;
	}
	s->sticky_outside = 0;
}

/* ---- exported API ---------------------------------------------------------------------------- */
static pcg32 g_rng = { 0x853c49e6748fea9bULL, 0xda3e39cb94b95bdbULL };
void pto_pcg_seed(uint64_t state, uint64_t seq) { pcg_seed(&g_rng, state, seq); }
void pto_pcg_u32(int n, uint32_t* out) { for(int i = 0; i < n; i++) out[i] = pcg_next(&g_rng); }
void pto_pcg_int(int from, int to, int n, int32_t* out) { for(int i = 0; i < n; i++) out[i] = random_int(&g_rng, from, to); }
void pto_pcg_float(float from, float to, float prec, int n, float* out) { for(int i = 0; i < n; i++) out[i] = random_float(&g_rng, from, to, prec); }

void pto_scene_destroy(pto_scene* s) {
	if(!s) return;
	free(s->obj); free(s->dirs); free(s->disc); free(s->samples); free(s->offsets);
	free(s->cell_start); free(s->cell_items); free(s->plane_ids);
	free(s);
}

/* Builds everything a render needs. Tables, then blue noise, are drawn from PCG stream
 * (0, settings->seed), the order the application uses (code/app.cpp:669-727 then :1174). */
pto_scene* pto_scene_create(const pt_world* w, const pt_object* objs, int n, const pt_settings* st) {
	pto_scene* s = (pto_scene*)calloc(1, sizeof(pto_scene));
	s->win = *w;
	s->st = *st;
	s->n = n;
	s->obj = (obj_t*)calloc((size_t)(n > 0 ? n : 1), sizeof(obj_t));
	for(int i = 0; i < n; i++) {
		obj_t* o = s->obj + i;
		o->pos = V3(objs[i].pos[0], objs[i].pos[1], objs[i].pos[2]);
		o->rot.w = objs[i].rot[0]; o->rot.x = objs[i].rot[1]; o->rot.y = objs[i].rot[2]; o->rot.z = objs[i].rot[3];
		o->dim = V3(objs[i].dim[0], objs[i].dim[1], objs[i].dim[2]);
		o->color = V3(objs[i].color[0], objs[i].color[1], objs[i].color[2]);
		o->type = objs[i].type;
		o->emit = V3(objs[i].emit_color[0], objs[i].emit_color[1], objs[i].emit_color[2]);
		o->refl = objs[i].reflection_mod; o->ior = objs[i].refractive_index; o->rough = objs[i].roughness;
		precalc_object(o);
	}
	s->plane_ids = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
	for(int i = 0; i < n; i++) if(s->obj[i].type == PT_GEOM_PLANE) s->plane_ids[s->n_planes++] = i;
	precalc_world(s);
	pcg32 rng;
	pcg_seed(&rng, 0, st->seed);
	s->dirs = (v3*)malloc(sizeof(v3) * PT_TABLE_COUNT);
	s->disc = (v2*)malloc(sizeof(v2) * PT_TABLE_COUNT);
	make_tables(&rng, s->dirs, s->disc);
	make_samples(s, &rng);
	build_grid(s);
	return s;
}

void pto_get_tables(const pto_scene* s, float* dirs, float* disc) {
	if(dirs) memcpy(dirs, s->dirs, sizeof(v3) * PT_TABLE_COUNT);
	if(disc) memcpy(disc, s->disc, sizeof(v2) * PT_TABLE_COUNT);
}
int pto_get_samples(const pto_scene* s, float* samples, int32_t* offsets) {
	if(samples) memcpy(samples, s->samples, sizeof(v2) * (size_t)s->sample_count);
	if(offsets) memcpy(offsets, s->offsets, sizeof(int) * (size_t)(s->grid_w * s->grid_w + 1));
	return s->sample_count;
}
static void put3(float* o, v3 v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }
void pto_get_records(const pto_scene* s, float* rec) {
	for(int i = 0; i < s->n; i++) {
		const obj_t* o = s->obj + i;
		float* r = rec + (size_t)i * PT_F_STRIDE;
		memset(r, 0, sizeof(float) * PT_F_STRIDE);
		put3(r + PT_F_POS, o->pos);
		r[PT_F_ROT] = o->rot.w; r[PT_F_ROT + 1] = o->rot.x; r[PT_F_ROT + 2] = o->rot.y; r[PT_F_ROT + 3] = o->rot.z;
		r[PT_F_ROTINV] = o->rot_inv.w; r[PT_F_ROTINV + 1] = o->rot_inv.x; r[PT_F_ROTINV + 2] = o->rot_inv.y; r[PT_F_ROTINV + 3] = o->rot_inv.z;
		put3(r + PT_F_DIM, o->dim);
		put3(r + PT_F_SCALE, o->dim_scale);
		put3(r + PT_F_SCALEINV, o->dim_scale_inv);
		put3(r + PT_F_COLOR, o->color_lin);
		put3(r + PT_F_EMIT, o->emit_lin);
		put3(r + PT_F_BBOX, o->bbox);
		r[PT_F_BRADIUS] = o->bradius;
		r[PT_F_REFL] = o->refl; r[PT_F_IOR] = o->ior; r[PT_F_ROUGH] = o->rough;
		int32_t type = o->type, flags = (o->is_rotated ? 1 : 0) | (o->even_dim ? 2 : 0);
		memcpy(r + PT_F_TYPE, &type, 4);
		memcpy(r + PT_F_FLAGS, &flags, 4);
	}
}
void pto_get_world(const pto_scene* s, float* out) {
	memset(out, 0, sizeof(float) * PT_W_COUNT);
	out[PT_W_DIMH] = s->dim_h; out[PT_W_NEAR] = s->near_dist;
	put3(out + PT_W_DIR, s->cam_dir); put3(out + PT_W_UP, s->cam_up); put3(out + PT_W_RIGHT, s->cam_right);
	put3(out + PT_W_TOPLEFT, s->cam_top_left); put3(out + PT_W_LIGHTDIR, s->light_dir);
	put3(out + PT_W_BG, s->bg_lin); put3(out + PT_W_AMBIENT, s->ambient_lin); put3(out + PT_W_LIGHT, s->light_lin);
}
void pto_get_grid(const pto_scene* s, int32_t* ints9) {
	const grid_t* g = &s->grid;
	ints9[0] = g->coord_min.x; ints9[1] = g->coord_min.y; ints9[2] = g->coord_min.z;
	ints9[3] = g->cell_count.x; ints9[4] = g->cell_count.y; ints9[5] = g->cell_count.z;
	ints9[6] = g->cell_size.x; ints9[7] = g->cell_size.y; ints9[8] = g->cell_size.z;
}
int pto_get_cell(const pto_scene* s, int index, int32_t* items, int cap) {
	int n = (int)(s->cell_start[index + 1] - s->cell_start[index]);
	for(int i = 0; i < n && i < cap; i++) items[i] = (int32_t)s->cell_items[s->cell_start[index] + i];
	return n;
}
int pto_get_csr(const pto_scene* s, uint32_t* start, uint32_t* items) {
	if(start) memcpy(start, s->cell_start, sizeof(uint32_t) * ((size_t)s->cells + 1));
	if(items) memcpy(items, s->cell_items, sizeof(uint32_t) * s->cell_start[s->cells]);
	return (int)s->cell_start[s->cells];
}

static v3 g3(const float* p) { return V3(p[0], p[1], p[2]); }
void pto_line_sphere(int n, const float* lp, const float* ld, const float* sp, const float* sr, float* dist, float* pos, float* nrm) {
	for(int i = 0; i < n; i++) { v3 p = V3(0, 0, 0), q = V3(0, 0, 0); dist[i] = line_sphere(g3(lp + 3 * i), g3(ld + 3 * i), g3(sp + 3 * i), sr[i], &p, &q); put3(pos + 3 * i, p); put3(nrm + 3 * i, q); }
}
void pto_line_box(int n, const float* lp, const float* ld, const float* bp, const float* bd, float* dist, float* pos, float* nrm) {
	for(int i = 0; i < n; i++) { v3 p = V3(0, 0, 0), q = V3(0, 0, 0); dist[i] = line_box(g3(lp + 3 * i), g3(ld + 3 * i), g3(bp + 3 * i), g3(bd + 3 * i), &p, &q); put3(pos + 3 * i, p); put3(nrm + 3 * i, q); }
}
void pto_line_cylinder(int n, const float* lp, const float* ld, float* dist, float* pos, float* nrm) {
	for(int i = 0; i < n; i++) { v3 p = V3(0, 0, 0), q = V3(0, 0, 0); dist[i] = line_cylinder(g3(lp + 3 * i), g3(ld + 3 * i), &p, &q); put3(pos + 3 * i, p); put3(nrm + 3 * i, q); }
}
void pto_line_cone(int n, const float* lp, const float* ld, float* dist, float* pos, float* nrm) {
	for(int i = 0; i < n; i++) { v3 p = V3(0, 0, 0), q = V3(0, 0, 0); dist[i] = line_cone(g3(lp + 3 * i), g3(ld + 3 * i), &p, &q); put3(pos + 3 * i, p); put3(nrm + 3 * i, q); }
}
void pto_sphere_collision(int n, const float* lp, const float* ld, const float* sp, const float* sr, int32_t* hit) {
	for(int i = 0; i < n; i++) hit[i] = sphere_collision(g3(lp + 3 * i), g3(ld + 3 * i), g3(sp + 3 * i), sr[i]);
}
void pto_optics(int n, const float* inc, const float* nrm, const float* ior, float* kr, float* refr, float* refl) {
	for(int i = 0; i < n; i++) {
		kr[i] = fresnel(g3(inc + 3 * i), g3(nrm + 3 * i), ior[i]);
		put3(refr + 3 * i, refract_dir(g3(inc + 3 * i), g3(nrm + 3 * i), ior[i]));
		put3(refl + 3 * i, reflect_vector(g3(inc + 3 * i), g3(nrm + 3 * i)));
	}
}
void pto_get_intersection(pto_scene* s, int n, const int32_t* obj, const float* rp, const float* rd, float* dist, float* pos, float* nrm) {
	for(int i = 0; i < n; i++) { v3 p = V3(0, 0, 0), q = V3(0, 0, 0); dist[i] = get_intersection(s->obj + obj[i], g3(rp + 3 * i), g3(rd + 3 * i), &p, &q, &s->cnt); put3(pos + 3 * i, p); put3(nrm + 3 * i, q); }
}
int pto_traverse(const pto_scene* s, const float* rp, const float* rd, int cap, int32_t* cells, int32_t* outside) {
	walk_t w;
	w.outside = 0;
	walk_init(&s->grid, &w, g3(rp), g3(rd));
	*outside = w.outside;
	if(w.outside) return 0;
	int count = 0;
	do {
		if(count < cap) { cells[3 * count] = w.coord.x; cells[3 * count + 1] = w.coord.y; cells[3 * count + 2] = w.coord.z; }
		count++;
	} while(walk_next(&s->grid, &w));
	return count;
}
void pto_cast(pto_scene* s, int n, const float* rp, const float* rd, int use_grid, int32_t* id, float* pos, float* nrm) {
	for(int i = 0; i < n; i++) {
		v3 p = V3(0, 0, 0), q = V3(0, 0, 0);
		if(use_grid == 2) id[i] = cast_all(s, g3(rp + 3 * i), g3(rd + 3 * i), 0, &p, &q);   /* the editor's pick: castRayAll(pos, dir, false, ...), code/app.cpp:1637 */
		else if(use_grid) id[i] = cast_grid(s, g3(rp + 3 * i), g3(rd + 3 * i), &p, &q, 0);
		else {
			walk_t w;
			w.outside = 0;
			walk_init(&s->grid, &w, g3(rp + 3 * i), g3(rd + 3 * i));
			id[i] = cast_all(s, g3(rp + 3 * i), g3(rd + 3 * i), w.outside, &p, &q);
		}
		put3(pos + 3 * i, p); put3(nrm + 3 * i, q);
	}
}
/* Returns the job count; fills at most `cap` jobs. */
int pto_jobs(const pto_scene* s, int32_t* xywh, int cap) {
	return make_jobs(s->st.width, s->st.height, xywh, cap);
}

/* Single-threaded render of the whole image. policy PT_SEED_PER_JOB: tile jobs in spiral order,
 * job i seeded with (0, seed_base + i).  policy PT_SEED_PER_SAMPLE: seed_base is ignored. */
int pto_render(pto_scene* s, uint64_t seed_base, int policy, float* rgb) {
	frame_t f;
	frame_init(s, &f);
	int n = make_jobs(s->st.width, s->st.height, 0, 0);
	int* jobs = (int*)malloc(sizeof(int) * 4 * (size_t)n);
	make_jobs(s->st.width, s->st.height, jobs, n);
	for(int i = 0; i < n; i++) run_job(s, &f, jobs + 4 * i, i, policy, seed_base, rgb);
	free(jobs);
	return n;
}

/* One job over an arbitrary rectangle, generator seeded with (0, seed). */
void pto_render_rect(pto_scene* s, uint64_t seed, int policy, int x, int y, int w, int h, float* rgb) {
	frame_t f;
	frame_init(s, &f);
	int job[4] = { x, y, w, h };
	run_job(s, &f, job, 0, policy, seed, rgb);
}

/* Multithreaded render scheduled like the reference: the job list above, handed out in order to
 * `threads` workers (the reference runs cores-1 of them, code/main.cpp:39-44). */
typedef struct { pto_scene* s; pto_scene* local; const frame_t* f; const int* jobs; int n; volatile int* next; int policy; uint64_t seed; float* rgb; } worker_t;
static void* worker_main(void* arg) {
	worker_t* w = (worker_t*)arg;
	for(;;) {
		int i = __sync_fetch_and_add(w->next, 1);
		if(i >= w->n) break;
		run_job(w->local, w->f, w->jobs + 4 * i, i, w->policy, w->seed, w->rgb);
	}
	return 0;
}
int pto_render_mt(pto_scene* s, uint64_t seed_base, int policy, int threads, float* rgb) {
	frame_t f;
	frame_init(s, &f);
	int n = make_jobs(s->st.width, s->st.height, 0, 0);
	int* jobs = (int*)malloc(sizeof(int) * 4 * (size_t)n);
	make_jobs(s->st.width, s->st.height, jobs, n);
	if(threads < 1) threads = 1;
	if(threads > 256) threads = 256;
	volatile int next = 0;
	pthread_t tid[256];
	worker_t wk[256];
	pto_scene* locals = (pto_scene*)malloc(sizeof(pto_scene) * (size_t)threads);
	for(int t = 0; t < threads; t++) {
		locals[t] = *s;   /* shallow copy: shared read-only tables, private counters */
		memset(&locals[t].cnt, 0, sizeof(counters_t));
		worker_t w = { s, locals + t, &f, jobs, n, &next, policy, seed_base, rgb };
		wk[t] = w;
		pthread_create(&tid[t], 0, worker_main, &wk[t]);
	}
	for(int t = 0; t < threads; t++) {
		pthread_join(tid[t], 0);
		uint64_t* dst = (uint64_t*)&s->cnt; const uint64_t* src = (const uint64_t*)&locals[t].cnt;
		for(size_t k = 0; k < sizeof(counters_t) / sizeof(uint64_t); k++) dst[k] += src[k];
	}
	free(locals);
	free(jobs);
	return n;
}

/* Entity hit by the primary ray of camera sample 0 of each pixel (-1: the tile has no sample) and
 * the hash of the grid cells that ray visited (0 on the brute-force path). */
void pto_primary(pto_scene* s, int32_t* ids, uint64_t* hashes) {
	frame_t f;
	frame_init(s, &f);
	int keep = s->st.max_bounces;
	s->st.max_bounces = 1;
	for(int y = 0; y < s->st.height; y++) for(int x = 0; x < s->st.width; x++) {
		int id = -1; uint64_t h = 0;
		v3 cpp = pixel_pos(s, &f, x, y);
		int gw = s->grid_w;
		int tile = (y % gw) * gw + (x % gw);
		int offset = s->offsets[tile];
		int count = s->offsets[tile + 1] - offset;
		if(count > 0) {
			pcg32 pr, sr;
			pixel_stream(s, x, y, &pr);
			int lens = random_int(&pr, 0, PT_TABLE_COUNT - 1);
			sample_stream(s, x, y, 0, 0, &sr);
			trace_sample(s, &f, cpp, s->samples[offset], lens, &sr, &id, &h);
		}
		if(ids) ids[(size_t)y * s->st.width + x] = id;
		if(hashes) hashes[(size_t)y * s->st.width + x] = h;
	}
	s->st.max_bounces = keep;
}

/* Unclamped radiance of every sample of pixel (x, y) under PT_SEED_PER_SAMPLE; returns the count. */
int pto_sample_radiance(pto_scene* s, int x, int y, float* out, int cap) {
	frame_t f;
	frame_init(s, &f);
	int gw = s->grid_w;
	int tile = (y % gw) * gw + (x % gw);
	int count = s->offsets[tile + 1] - s->offsets[tile];
	if(count > cap) return -count;
	trace_pixel(s, &f, x, y, PT_SEED_PER_SAMPLE, 0, out, 0, 0);
	return count;
}

/* Progressive frame: sample (frame mod count) of every pixel added into acc (width*height*3). */
void pto_progressive_frame(pto_scene* s, int frame, float* acc) {
	frame_t f;
	frame_init(s, &f);
	for(int y = 0; y < s->st.height; y++) for(int x = 0; x < s->st.width; x++) {
		int gw = s->grid_w;
		int tile = (y % gw) * gw + (x % gw);
		int offset = s->offsets[tile];
		int count = s->offsets[tile + 1] - offset;
		if(count == 0) continue;
		int si = frame % count;
		uint64_t pass = (uint64_t)(frame / count);
		pcg32 pr, sr;
		pixel_stream(s, x, y, &pr);
		int lens = random_int(&pr, 0, PT_TABLE_COUNT - 1);
		sample_stream(s, x, y, si, pass, &sr);
		v3 c = trace_sample(s, &f, pixel_pos(s, &f, x, y), s->samples[offset + si], lens + si, &sr, 0, 0);
		c = clamp_max3(c, 1);
		float* o = acc + 3 * ((size_t)y * s->st.width + x);
		o[0] += c.x; o[1] += c.y; o[2] += c.z;
	}
}

/* Exact number of camera samples of the image (or of the tiles of one partition). */
uint64_t pto_path_count(const pto_scene* s) {
	uint64_t n = 0;
	int gw = s->grid_w;
	for(int y = 0; y < s->st.height; y++) for(int x = 0; x < s->st.width; x++) {
		int tile = (y % gw) * gw + (x % gw);
		n += (uint64_t)(s->offsets[tile + 1] - s->offsets[tile]);
	}
	return n;
}

/* Event counters accumulated by every render since creation (16 values):
 * casts, dda_steps, bsphere_tests, isect[sphere,box,cylinder,cone,plane], rotated_box, uneven_sphere,
 * shade_opaque, shade_oren, shade_glass, paths, lens, emit. */
void pto_counters(const pto_scene* s, uint64_t* out) { memcpy(out, &s->cnt, 16 * sizeof(uint64_t)); }
/* out[k], k < bins: camera samples traced since creation (or the last reset) whose path cast k rays; the last bin
 * also takes every longer path. */
void pto_cast_histogram(const pto_scene* s, uint64_t* out, int bins) {
	for(int k = 0; k < bins; k++) out[k] = 0;
	for(int k = 0; k < PTO_HIST_BINS; k++) if(bins > 0) out[k < bins ? k : bins - 1] += s->cnt.path_casts[k];
}
void pto_counters_reset(pto_scene* s) { memset(&s->cnt, 0, sizeof(counters_t)); }

/* Screenshot encoding: code/app.cpp:1541-1547, code/rt_math.cpp:351-353,3634-3640. */
/* Over `count` consecutive float bit patterns from first_bits: inputs whose screenshot code (the expression of
 * pto_srgb8 below) differs from the number of thresholds at or below them.  thr255 ascending. */
uint64_t pto_srgb8_check(const float* thr255, uint32_t first_bits, uint64_t count) {
	uint64_t bad = 0;
	uint32_t reached = 0;
	for(uint64_t k = 0; k < count; k++) {
		uint32_t bits = first_bits + (uint32_t)k;
		float v;
		memcpy(&v, &bits, 4);
		float e = v;
		if(e < 0.0031308f) e *= 12.92f;
		else e = 1.055f * powf(e, 1.0f / 2.4f) - 0.055f;
		uint8_t code = (uint8_t)(char)(int)roundf(e * 255);
		while(reached < 255 && thr255[reached] <= v) reached++;      /* inputs ascend, so the count only grows */
		while(reached > 0 && !(thr255[reached - 1] <= v)) reached--;
		if(code != (uint8_t)reached) bad++;
	}
	return bad;
}

void pto_srgb8(int n, const float* lin, uint8_t* out) {
	for(int i = 0; i < n; i++) {
		float v = lin[i];
		if(v < 0.0031308f) v *= 12.92f;
		else v = 1.055f * powf(v, 1.0f / 2.4f) - 0.055f;
		out[i] = (uint8_t)(char)(int)roundf(v * 255);
	}
}
