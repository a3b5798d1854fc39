// Host-side stages of the render core.  Each function names the reference lines whose results it
// reproduces bit for bit (scalar fp32, the reference's operation order, no contraction).
#include "pt_host.h"

#include <cfloat>
#include <cmath>
#include <cstring>

namespace pt {

// ---- small fp32 vector kit (code/rt_math.cpp:1979-2168, 2789-2916) -------------------------------
static inline f3 operator+(f3 a, f3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
static inline f3 operator-(f3 a, f3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
static inline f3 operator*(f3 a, f3 b) { return { a.x * b.x, a.y * b.y, a.z * b.z }; }
static inline f3 operator*(f3 a, float b) { return { a.x * b, a.y * b, a.z * b }; }
static inline f3 operator/(f3 a, float b) { return { a.x / b, a.y / b, a.z / b }; }
static inline f3 operator/(f3 a, f3 b) { return { a.x / b.x, a.y / b.y, a.z / b.z }; }
static inline f3 operator-(f3 a) { return { -a.x, -a.y, -a.z }; }
static inline f3 cross(f3 a, f3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
static inline float length(f3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
static inline f3 normalise(f3 a) { return a / length(a); }
// min/max are ternaries in the reference (code/rt_math.cpp:60-74), not fminf/fmaxf.
static inline float tmin(float a, float b) { return a <= b ? a : b; }
static inline float tmax(float a, float b) { return a >= b ? a : b; }

static inline q4 axis_angle(float a, f3 axis) {
	return { cosf(a * 0.5f), axis.x * sinf(a * 0.5f), axis.y * sinf(a * 0.5f), axis.z * sinf(a * 0.5f) };
}
static inline q4 operator*(q4 a, q4 b) {
	return { (a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z), (a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y),
	         (a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x), (a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w) };
}
static inline f3 rotate(q4 q, f3 v) {
	f3 u = { q.x, q.y, q.z };
	f3 t = cross(u, v) * 2.0f;
	return (v + t * q.w) + cross(u, t);
}
static inline float gamma_decode(float c) { return powf(c, 2.2f); }   // code/rt_math.cpp:3619-3624
static const float kPi = (float)3.1415926535897932384626433832795;

// ---- PCG32 ---------------------------------------------------------------------------------------
uint32_t Pcg::next() {
	uint64_t old = state;
	state = old * 6364136223846793005ULL + inc;
	uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
	uint32_t rot = (uint32_t)(old >> 59u);
	return (xs >> rot) | (xs << ((-rot) & 31));
}
void Pcg::seed(uint64_t initstate, uint64_t initseq) {
	state = 0;
	inc = (initseq << 1u) | 1u;
	next();
	state += initstate;
	next();
}
uint32_t Pcg::bounded(uint32_t bound) {
	uint32_t threshold = (0u - bound) % bound;
	for(;;) {
		uint32_t r = next();
		if(r >= threshold) return r % bound;
	}
}
int Pcg::range(int from, int to) { return (int)(bounded((uint32_t)(to - from + 1)) + (uint32_t)from); }
float Pcg::quantised(float from, float to, float prec) { return range((int)(from / prec), (int)(to / prec)) * prec; }

// ---- per-object precalc: code/raycast.cpp:80-139, 2064-2108 -------------------------------------
static void put(float* r, int at, f3 v) { r[at] = v.x; r[at + 1] = v.y; r[at + 2] = v.z; }

void flatten_objects(const pt_object* objs, int n, float* records) {
	for(int i = 0; i < n; i++) {
		const pt_object& o = objs[i];
		float* r = records + (size_t)i * PT_F_STRIDE;
		memset(r, 0, sizeof(float) * PT_F_STRIDE);
		f3 pos = { o.pos[0], o.pos[1], o.pos[2] };
		f3 dim = { o.dim[0], o.dim[1], o.dim[2] };
		q4 rot = { o.rot[0], o.rot[1], o.rot[2], o.rot[3] };

		float radius = 0;
		switch(o.type) {
			case PT_GEOM_BOX: radius = length(dim) * 0.5f; break;
			case PT_GEOM_SPHERE: radius = tmax(tmax(dim.x, dim.y), tmax(dim.y, dim.z)) * 0.5f; break;
			case PT_GEOM_CYLINDER:
			case PT_GEOM_CONE: radius = tmax(dim.x, dim.y) * 0.5f + dim.z * 0.5f; break;
		}

		// symmetric AABB of the rotated corners (code/raycast.cpp:99-139); a plane counts as a slab of 10^6 x 10^6 x 0
		const f3 extent = o.type == PT_GEOM_PLANE ? f3{ 1000000, 1000000, 0 } : dim;
		f3 lo = { FLT_MAX, FLT_MAX, FLT_MAX }, hi = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
		for(int c = 0; c < 8; c++) {
			f3 corner = { (c & 4) ? -0.5f : 0.5f, (c & 2) ? -0.5f : 0.5f, (c & 1) ? -0.5f : 0.5f };
			f3 p = rotate(rot, extent * corner);
			lo = { tmin(lo.x, p.x), tmin(lo.y, p.y), tmin(lo.z, p.z) };
			hi = { tmax(hi.x, p.x), tmax(hi.y, p.y), tmax(hi.z, p.z) };
		}
		f3 bbox = { hi.x * 2, hi.y * 2, hi.z * 2 };

		float inv_len2 = (1 / (rot.w * rot.w + rot.x * rot.x + rot.y * rot.y + rot.z * rot.z));
		q4 rinv = { rot.w * inv_len2, -rot.x * inv_len2, -rot.y * inv_len2, -rot.z * inv_len2 };
		f3 unit = { 1, 1, 1 };
		if(o.type == PT_GEOM_CYLINDER) unit = { 2, 2, 2 };
		if(o.type == PT_GEOM_CONE) unit = { 2, 2, 1 };
		const bool plane = o.type == PT_GEOM_PLANE;      // preCalcObjects leaves a plane's scales alone (code/raycast.cpp:2081-2105)

		put(r, PT_F_POS, pos);
		r[PT_F_ROT] = rot.w; r[PT_F_ROT + 1] = rot.x; r[PT_F_ROT + 2] = rot.y; r[PT_F_ROT + 3] = rot.z;
		r[PT_F_ROTINV] = rinv.w; r[PT_F_ROTINV + 1] = rinv.x; r[PT_F_ROTINV + 2] = rinv.y; r[PT_F_ROTINV + 3] = rinv.z;
		put(r, PT_F_DIM, dim);
		put(r, PT_F_SCALE, plane ? f3{ 0, 0, 0 } : dim / unit);
		put(r, PT_F_SCALEINV, plane ? f3{ 0, 0, 0 } : unit / dim);
		put(r, PT_F_COLOR, { gamma_decode(o.color[0]), gamma_decode(o.color[1]), gamma_decode(o.color[2]) });
		put(r, PT_F_EMIT, { gamma_decode(o.emit_color[0]), gamma_decode(o.emit_color[1]), gamma_decode(o.emit_color[2]) });
		put(r, PT_F_BBOX, bbox);
		r[PT_F_BRADIUS] = radius;
		r[PT_F_REFL] = o.reflection_mod;
		r[PT_F_IOR] = o.refractive_index;
		r[PT_F_ROUGH] = o.roughness;
		int32_t type = o.type;
		int32_t flags = (!(rot.w == 1 && rot.x == 0 && rot.y == 0 && rot.z == 0) ? 1 : 0) |
		                ((dim.x == dim.y && dim.x == dim.z) ? 2 : 0);
		memcpy(r + PT_F_TYPE, &type, 4);
		memcpy(r + PT_F_FLAGS, &flags, 4);
	}
}

// ---- camera / light / colours: code/app.cpp:1077-1090,1230-1241; code/raycast.cpp:173-191,271-277;
//      code/rt_math.cpp:258-260,340-347 ----------------------------------------------------------
void world_consts(const pt_world& w, int width, int height, WorldConsts* o) {
	o->cam_pos = { w.camera.pos[0], w.camera.pos[1], w.camera.pos[2] };
	o->dim_w = w.camera.plane_width;
	float aspect = (float)width / height;
	o->dim_h = o->dim_w * (1 / aspect);

	float fov = w.camera.fov * (kPi / 180);
	float side_angle = (float)((3.1415926535897932384626433832795 - fov) / 2.0f);
	float side = o->dim_w / sinf(fov) * sinf(side_angle);
	o->near_dist = side * sinf(side_angle);

	const f3 base_up = { 0, 0, 1 };
	f3 dir = { 0, 1, 0 };
	dir = rotate(axis_angle(w.camera.rot[0], base_up), dir);
	dir = rotate(axis_angle(w.camera.rot[1], normalise(cross(base_up, dir))), dir);
	o->cam_dir = dir;
	o->cam_up = normalise(cross(dir, normalise(cross(base_up, dir))));
	o->cam_right = -normalise(cross(o->cam_up, dir));
	o->top_left = ((o->cam_pos + o->cam_dir * o->near_dist) + (o->cam_right * -1.0f) * (o->dim_w / 2.0f)) +
	              o->cam_up * (o->dim_h / 2.0f);

	float axy = (float)w.light_rot[0] * (kPi / 180);
	float az = (float)w.light_rot[1] * (kPi / 180);
	o->light_dir = rotate(axis_angle(axy, { 0, 0, -1 }) * axis_angle(-az, { 0, 1, 0 }), { 1, 0, 0 });
	o->bg = { gamma_decode(w.background_color[0]), gamma_decode(w.background_color[1]), gamma_decode(w.background_color[2]) };
	o->ambient = { gamma_decode(w.ambient_color[0]), gamma_decode(w.ambient_color[1]), gamma_decode(w.ambient_color[2]) };
	o->light = { gamma_decode(w.light_color[0]), gamma_decode(w.light_color[1]), gamma_decode(w.light_color[2]) };
	o->focal_distance = w.focal_distance;
	o->aperture = w.aperture;
}

// ---- random tables: code/app.cpp:669-727 ----------------------------------------------------------
// The draws are call arguments in the reference and are evaluated last-to-first by its compilers,
// so the last component is drawn first.
void make_tables(Pcg& rng, f3* dirs, f2* disc) {
	const float q = 0.001f;
	for(int i = 0; i < PT_TABLE_COUNT; i++) {
		f3 d;
		do {
			d.z = rng.quantised(-1, 1, q);
			d.y = rng.quantised(-1, 1, q);
			d.x = rng.quantised(-1, 1, q);
		} while(length(d) > 1);
		d = normalise(d);
		if(d.x == 0 && d.y == 0 && d.z == 0) d = { 1, 0, 0 };
		dirs[i] = d;
	}
	const float range = 0.5f;
	for(int i = 0; i < PT_TABLE_COUNT; i++) {
		f2 p;
		do {
			p.y = rng.quantised(-range, range, q);
			p.x = rng.quantised(-range, range, q);
		} while(sqrtf(p.x * p.x + p.y * p.y) > range);
		if(p.x == 0 && p.y == 0) p = { range, 0 };
		disc[i] = p;
	}
}

// ---- Poisson-disc ("blue noise") samples on a torus: code/rt_math.cpp:3488-3585 -----------------
static inline int wrap(int a, int b) { int r = a % b; return r < 0 ? r + b : r; }
static inline int floor_toward_neg(float v) { return (int)(v < 0 ? (float)(((int)v) - 1) : (float)((int)v)); }

static std::vector<f2> poisson_disc(Pcg& rng, float width, float radius) {
	const float cell = (float)(radius / 1.41421356237309504880);
	const int gw = (int)ceilf(width / cell), gh = (int)ceilf(width / cell);
	const int slots = gw * gh;
	const float q = 0.0001f;
	std::vector<f2> pts;
	pts.reserve((size_t)slots);
	std::vector<int> owner((size_t)slots + 1, -1);   // one spare slot: a point exactly on the far edge
	std::vector<int> active;

	f2 first;
	first.y = rng.quantised(0, width, q);
	first.x = rng.quantised(0, width, q);
	pts.push_back(first);
	active.push_back(0);
	owner[(size_t)((int)(first.y / cell) * gw + (int)(first.x / cell))] = 0;

	while(!active.empty()) {
		int pick = rng.range(0, (int)active.size() - 1);
		f2 from = pts[(size_t)active[(size_t)pick]];
		for(int attempt = 0; attempt < 30; ++attempt) {
			float angle = rng.quantised(0, (float)6.2831853071795864769252867665590, q);
			float dist = rng.quantised(radius, radius * 2, q);
			f2 cand = { from.x + cosf(angle) * dist, from.y + sinf(angle) * dist };
			if(!(cand.x >= 0 && cand.x <= width && cand.y >= 0 && cand.y <= width)) continue;

			int x0 = floor_toward_neg((cand.x - radius * 2.0f) / cell), y0 = floor_toward_neg((cand.y - radius * 2.0f) / cell);
			int x1 = floor_toward_neg((cand.x + radius * 2.0f) / cell), y1 = floor_toward_neg((cand.y + radius * 2.0f) / cell);
			bool ok = true;
			for(int y = y0; y <= y1 && ok; ++y) {
				for(int x = x0; x <= x1; ++x) {
					int wx = wrap(x, gw), wy = wrap(y, gh);
					int other = owner[(size_t)(wy * gw + wx)];
					if(other < 0) continue;
					f2 s = pts[(size_t)other];
					if(wx != x) s.x += wx > x ? -width : width;
					if(wy != y) s.y += wy > y ? -width : width;
					float dx = s.x - cand.x, dy = s.y - cand.y;
					if(sqrtf(dx * dx + dy * dy) < radius) { ok = false; break; }
				}
			}
			if(ok) {
				if((int)pts.size() >= slots) break;
				int slot = (int)(cand.y / cell) * gw + (int)(cand.x / cell);
				if(slot > slots) slot = slots;
				owner[(size_t)slot] = (int)pts.size();
				active.push_back((int)pts.size());
				pts.push_back(cand);
			}
		}
		active[(size_t)pick] = active.back();
		active.pop_back();
	}
	return pts;
}

// ---- bucket the samples per pixel tile: code/app.cpp:1161-1225 ---------------------------------
void make_samples(Pcg& rng, int spp_wanted, int gw, std::vector<f2>* samples, std::vector<int32_t>* offsets) {
	int per_axis = (int)sqrt((double)spp_wanted);
	float scaled = per_axis * 1.3f;
	std::vector<f2> pts = poisson_disc(rng, (float)gw, 1 / (float)scaled);
	offsets->assign((size_t)(gw * gw + 1), 0);
	auto tile_of = [gw](f2 p, int* tx, int* ty) {
		*tx = (int)p.x; *ty = (int)p.y;
		if(*tx > gw - 1) *tx = gw - 1;
		if(*ty > gw - 1) *ty = gw - 1;
	};
	for(const f2& p : pts) {
		int tx, ty;
		tile_of(p, &tx, &ty);
		(*offsets)[(size_t)(1 + ty * gw + tx)] += 1;
	}
	for(int i = 0; i < gw * gw; i++) (*offsets)[(size_t)i + 1] += (*offsets)[(size_t)i];
	std::vector<int> fill((size_t)(gw * gw), 0);
	samples->assign(pts.size(), f2{ 0, 0 });
	for(const f2& p : pts) {
		int tx, ty;
		tile_of(p, &tx, &ty);
		int t = ty * gw + tx;
		(*samples)[(size_t)((*offsets)[(size_t)t] + fill[(size_t)t]++)] = { p.x - (float)tx, p.y - (float)ty };
	}
}

// ---- uniform grid: code/raycast.cpp:595-617, code/app.cpp:1247-1357 -------------------------------
static inline int clamp_cell(int v, int hi) {   // clampInt goes through the float clampMax (code/rt_math.cpp:179-181)
	int a = v < 0 ? 0 : v;
	float f = (float)a, h = (float)hi;
	return (int)(f > h ? h : f);
}

bool grid_desc(const float* records, int n, int hint, GridDesc* g) {
	float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
	for(int i = 0; i < n; i++) {
		const float* r = records + (size_t)i * PT_F_STRIDE;
		int32_t type;
		memcpy(&type, r + PT_F_TYPE, 4);
		if(type == PT_GEOM_PLANE) continue;
		for(int k = 0; k < 3; k++) {
			float half = r[PT_F_BBOX + k] / 2;
			lo[k] = tmin(lo[k], r[PT_F_POS + k] - half);
			hi[k] = tmax(hi[k], r[PT_F_POS + k] + half);
		}
	}
	if(n == 0 || !(lo[0] <= hi[0])) return false;
	int cmax[3];
	for(int k = 0; k < 3; k++) {
		float extent = hi[k] - lo[k];
		g->cell_size[k] = (int)((float)extent / hint);
		if(g->cell_size[k] == 0) g->cell_size[k] = 1;
	}
	for(int k = 0; k < 3; k++) g->coord_min[k] = (int)floorf((lo[k] / g->cell_size[k] - 0.5f) + 0.5f);
	for(int k = 0; k < 3; k++) cmax[k] = (int)floorf((hi[k] / g->cell_size[k] + 0.5f) + 0.5f);
	for(int k = 0; k < 3; k++) g->cell_count[k] = cmax[k] - g->coord_min[k];
	int64_t cells = g->cells();
	if(g->cell_count[0] <= 0 || g->cell_count[1] <= 0 || g->cell_count[2] <= 0 || cells > (1LL << 28)) return false;
	return true;
}

bool build_grid(const float* records, int n, int hint, GridDesc* g, std::vector<uint32_t>* cell_start,
                std::vector<uint32_t>* cell_items) {
	if(!grid_desc(records, n, hint, g)) return false;
	const int64_t cells = g->cells();

	// inclusive cell range of an object's box (code/app.cpp:1297-1312)
	auto range = [&](const float* r, int* a, int* b) {
		for(int k = 0; k < 3; k++) {
			float half = r[PT_F_BBOX + k] / 2;
			float mn = r[PT_F_POS + k] - half, mx = r[PT_F_POS + k] + half;
			a[k] = (int)(mn / g->cell_size[k]) - g->coord_min[k];
			if(mn < 0) a[k]--;
			a[k] = clamp_cell(a[k], g->cell_count[k] - 1);
			b[k] = (int)(mx / g->cell_size[k]) - g->coord_min[k];
			if(mx < 0) b[k]--;
			b[k] = clamp_cell(b[k], g->cell_count[k] - 1);
		}
	};
	const int64_t cx = g->cell_count[0], cy = g->cell_count[1];
	cell_start->assign((size_t)cells + 1, 0);
	for(int i = 0; i < n; i++) {
		int a[3], b[3];
		range(records + (size_t)i * PT_F_STRIDE, a, b);
		for(int z = a[2]; z <= b[2]; z++) for(int y = a[1]; y <= b[1]; y++) for(int x = a[0]; x <= b[0]; x++)
			(*cell_start)[(size_t)(z * cy * cx + y * cx + x) + 1]++;
	}
	for(int64_t c = 0; c < cells; c++) (*cell_start)[(size_t)c + 1] += (*cell_start)[(size_t)c];
	cell_items->assign((size_t)(*cell_start)[(size_t)cells], 0);
	std::vector<uint32_t> fill((size_t)cells, 0);
	for(int i = 0; i < n; i++) {   // ascending object id inside every cell, the reference's append order
		int a[3], b[3];
		range(records + (size_t)i * PT_F_STRIDE, a, b);
		for(int z = a[2]; z <= b[2]; z++) for(int y = a[1]; y <= b[1]; y++) for(int x = a[0]; x <= b[0]; x++) {
			size_t c = (size_t)(z * cy * cx + y * cx + x);
			(*cell_items)[(size_t)(*cell_start)[c] + fill[c]++] = (uint32_t)i;
		}
	}
	return true;
}

bool object_ranges(const float* records, int n, const GridDesc& g, std::vector<uint32_t>* ranges) {
	ranges->assign((size_t)(n > 0 ? n : 1) * 2, 0);
	if(g.cell_count[0] > 1023 || g.cell_count[1] > 1023 || g.cell_count[2] > 1023 || n >= (1 << 23)) return false;
	for(int i = 0; i < n; i++) {
		const float* r = records + (size_t)i * PT_F_STRIDE;
		uint32_t a[3], b[3];
		for(int k = 0; k < 3; k++) {   // the same range the grid build uses (code/app.cpp:1297-1312)
			float half = r[PT_F_BBOX + k] / 2;
			float mn = r[PT_F_POS + k] - half, mx = r[PT_F_POS + k] + half;
			int lo = (int)(mn / g.cell_size[k]) - g.coord_min[k];
			if(mn < 0) lo--;
			int hi = (int)(mx / g.cell_size[k]) - g.coord_min[k];
			if(mx < 0) hi--;
			a[k] = (uint32_t)clamp_cell(lo, g.cell_count[k] - 1);
			b[k] = (uint32_t)clamp_cell(hi, g.cell_count[k] - 1);
		}
		int32_t type, flags;
		memcpy(&type, r + PT_F_TYPE, 4);
		memcpy(&flags, r + PT_F_FLAGS, 4);
		uint32_t kind = type == PT_GEOM_SPHERE ? ((flags & 2) ? 0u : 3u) : type == PT_GEOM_BOX ? ((flags & 1) ? 2u : 1u) : type == PT_GEOM_CYLINDER ? 4u : type == PT_GEOM_CONE ? 5u : 6u;
		(*ranges)[(size_t)i * 2] = a[0] | (a[1] << 10) | (a[2] << 20) | ((kind & 3u) << 30);
		(*ranges)[(size_t)i * 2 + 1] = b[0] | (b[1] << 10) | (b[2] << 20) | ((kind >> 2) << 30);
	}
	return true;
}

// The normal of a plane: rot * (0, 0, 1), as getIntersection works it out for every test (code/raycast.cpp:887).
void plane_normal(const float* record, float out[3]) {
	const q4 rot = { record[PT_F_ROT], record[PT_F_ROT + 1], record[PT_F_ROT + 2], record[PT_F_ROT + 3] };
	const f3 n = rotate(rot, f3{ 0, 0, 1 });
	out[0] = n.x; out[1] = n.y; out[2] = n.z;
}

// Radius of a sphere about the entity's position that encloses it, for the cooperative cast's conservative filter (no result
// depends on it, only which entities get the exact test).  A plane has none: infinity lets every ray through.
float filter_radius(const float* record) {
	int32_t type;
	memcpy(&type, record + PT_F_TYPE, 4);
	if(type == PT_GEOM_PLANE) return INFINITY;
#ifndef PT_LOOSE_FILTER_RADIUS
	// The reference's bounding radius of a cylinder or cone, max(dx, dy) / 2 + dz / 2 (code/raycast.cpp:87-92), is generous: the
	// point farthest from the centre is on the rim of a cap, sqrt((max(dx, dy) / 2)^2 + (dz / 2)^2) away.  (Rounded up.)
	if(type == PT_GEOM_CYLINDER || type == PT_GEOM_CONE) {
		const double rim = (double)tmax(record[PT_F_DIM], record[PT_F_DIM + 1]) * 0.5, half = (double)record[PT_F_DIM + 2] * 0.5;
		return (float)(sqrt(rim * rim + half * half) * 1.000001);
	}
#endif
	return record[PT_F_BRADIUS];
}

// A sphere that encloses the entity, not necessarily about its position, for the same filter and for leaving entities out of
// the cells they stay clear of.  A cone (apex at +dz/2 on its own z axis, base of semi-axes dx/2, dy/2 at -dz/2:
// code/raycast.cpp:352-388 in unit space) fits a sphere through its apex and the rim of its base - or, when it is flatter
// than wide, the sphere about the base's centre with the rim's radius: for a cone twice as high as its base radius 1.25
// radii about a point a quarter of the height below the middle, instead of 1.41 about the middle.  Every other kind is
// symmetric about its position.
float filter_sphere(const float* record, float centre[3]) {
	centre[0] = record[PT_F_POS]; centre[1] = record[PT_F_POS + 1]; centre[2] = record[PT_F_POS + 2];
	int32_t type;
	memcpy(&type, record + PT_F_TYPE, 4);
#ifndef PT_LOOSE_FILTER_RADIUS
	if(type == PT_GEOM_CONE) {
		const double rim = (double)tmax(record[PT_F_DIM], record[PT_F_DIM + 1]) * 0.5, height = (double)record[PT_F_DIM + 2];
		if(!(rim > 0.0) || !(height > 0.0) || !std::isfinite(rim) || !std::isfinite(height)) return filter_radius(record);
		const double above_base = height > rim ? (height * height - rim * rim) / (2.0 * height) : 0.0;
		const double radius = height > rim ? (height * height + rim * rim) / (2.0 * height) : rim;
		// the cone's axis in the world: the rotation applied to (0, 0, 1) (Quat * Vec3, code/rt_math.cpp:2902-2907)
		const double w = record[PT_F_ROT], x = record[PT_F_ROT + 1], y = record[PT_F_ROT + 2], z = record[PT_F_ROT + 3];
		if(!(fabs(w * w + x * x + y * y + z * z - 1.0) < 1e-4)) return filter_radius(record);      // not a rotation: no better than before
		const double axis[3] = { 2.0 * (x * z + w * y), 2.0 * (y * z - w * x), 1.0 - 2.0 * (x * x + y * y) };
		const double along = above_base - height * 0.5;
		for(int a = 0; a < 3; a++) centre[a] = (float)((double)centre[a] + axis[a] * along);
		// rounded up: the centre's own rounding, a quaternion not quite of unit length, the arithmetic above
		return (float)(radius * 1.00001 + 1e-6 * (fabs((double)centre[0]) + fabs((double)centre[1]) + fabs((double)centre[2])));
	}
#endif
	return filter_radius(record);
}

// The reference's cone intersector divides by a = dx^2 + dy^2 - dz^2 of the ray's normalised direction in the cone's unit space
// (code/raycast.cpp:352-388).  For a ray parallel to one of the cone's generators a rounds to zero or next to it: one root is
// then NaN or garbage, the other is replaced by the ray's crossing of the base plane, and that point - anywhere on the plane -
// comes back as the hit.  No bounding volume of the cone holds such a ray; a cast that filters by bounding volume and wants
// these hits too has to let the cone through by the ray's direction d.  With ex, ey, ez the cone's axes in the world and
// A = 4/dx^2, B = 4/dy^2, C = 1/dz^2, a is (A (d.ex)^2 + B (d.ey)^2 - C (d.ez)^2) / (A (d.ex)^2 + B (d.ey)^2 + C (d.ez)^2), and
// (d.ey)^2 is |d|^2 less the other two squares.  out[0..2] = ex, out[3] = (A - B) / C, out[4..6] = ez, out[7] = B / C; the device
// keeps the cone when |a| < 0.02 (a root goes astray by about 1e-7 t / |a| of the ray's length t; the filter's own margins cover
// a hundred times that at 0.02).  A record the formula does not fit (a rotation not of unit length, an extent of zero, extreme
// proportions) gets NaNs: every ray keeps it.
void cone_slopes(const float* record, float out[8]) {
	const double w = record[PT_F_ROT], x = record[PT_F_ROT + 1], y = record[PT_F_ROT + 2], z = record[PT_F_ROT + 3];
	const double len2 = w * w + x * x + y * y + z * z;
	const double dx = record[PT_F_DIM], dy = record[PT_F_DIM + 1], dz = record[PT_F_DIM + 2];
	const bool fits = fabs(len2 - 1.0) < 1e-3 && std::isfinite(dx) && std::isfinite(dy) && std::isfinite(dz) && dx != 0.0 && dy != 0.0 && dz != 0.0;
	if(!fits) {
		for(int k = 0; k < 8; k++) out[k] = NAN;
		return;
	}
	// rot * (1, 0, 0) and rot * (0, 0, 1) (Quat * Vec3, code/rt_math.cpp:2902-2907)
	const double ex[3] = { 1.0 - 2.0 * (y * y + z * z), 2.0 * (x * y + w * z), 2.0 * (x * z - w * y) };
	const double ez[3] = { 2.0 * (x * z + w * y), 2.0 * (y * z - w * x), 1.0 - 2.0 * (x * x + y * y) };
	const double A = 4.0 / (dx * dx), B = 4.0 / (dy * dy), C = 1.0 / (dz * dz);
	// The device forms (d.ex)^2 + (d.ey)^2 as |d|^2 - (d.ez)^2, which loses about 6e-8 |d|^2: times a ratio of a thousand that is
	// still three hundred times below the tolerance.  A needle of a cone (its height more than fifteen base diameters) gets NaNs too.
	if(!(fabs((A - B) / C) < 1e3 && B / C < 1e3)) {
		for(int k = 0; k < 8; k++) out[k] = NAN;
		return;
	}
	for(int k = 0; k < 3; k++) { out[k] = (float)ex[k]; out[4 + k] = (float)ez[k]; }
	out[3] = (float)((A - B) / C);
	out[7] = (float)(B / C);
}

// ---- screenshot encoding (code/app.cpp:1541-1547) ------------------------------------------------------
uint8_t srgb8_encode(float v) {
	if(v < 0.0031308f) v *= 12.92f;
	else v = 1.055f * powf(v, 1.0f / 2.4f) - 0.055f;
	return (uint8_t)(char)(int)roundf(v * 255);
}

void srgb8_thresholds(float* out255) {
	uint32_t from = 0;                               // bits of 0.0f; every code's first input lies at or above the previous one's
	const uint32_t one = 0x3f800000u;                // bits of 1.0f, which encodes to 255
	for(int code = 1; code <= 255; code++) {
		uint32_t lo = from, hi = one;                // invariant: encode(lo) < code <= encode(hi)
		while(hi - lo > 1) {
			const uint32_t mid = lo + (hi - lo) / 2;
			float f;
			memcpy(&f, &mid, 4);
			if(srgb8_encode(f) >= code) hi = mid; else lo = mid;
		}
		memcpy(out255 + (code - 1), &hi, 4);
		from = lo;
	}
}

}  // namespace pt
